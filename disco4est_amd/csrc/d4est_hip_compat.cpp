// libd4est_hip_compat.so: the reference's own entry points for the hot path (include/d4est_hip_compat.h), implemented on the
// C-ABI of libd4est_hip.so only -- plain C++, no HIP header: everything device-side goes through d4est_hip.h.
//
// Element-level shims: a cached one-element plan per (deg, deg_quad, quadrature type) with persistent pinned staging and device
// buffers (allocated on first use, never per call); operator-level shims: the whole-mesh plan bound to the p4est pointer.
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "../../include/d4est_hip_compat.h"

#define COMPAT_ABORT(...)                                       \
  do {                                                          \
    std::fprintf(stderr, "[D4EST_HIP_ABORT] ");                 \
    std::fprintf(stderr, __VA_ARGS__);                          \
    std::fprintf(stderr, " (%s:%d)\n", __FILE__, __LINE__);     \
    std::abort();                                               \
  } while (0)

namespace {

struct ElemCtx {
  d4est_hip_plan_t* plan = nullptr;
  int N3 = 0, Q3 = 0, N2 = 0, cap = 0;
  double* h = nullptr;       // pinned: [in cap][out cap][geometry 10 Q3]
  double* d_in = nullptr;    // cap
  double* d_out = nullptr;   // 3 cap (dudr-sized outputs are not needed; 1 cap used)
  double* d_geo = nullptr;   // 10 Q3: J | rst_xyz[3i+j]
};

std::map<std::tuple<int, int, int>, ElemCtx> g_elem;
std::map<std::array<int, 10>, d4est_hip_transfer_t*> g_transfer;
std::map<const void*, d4est_hip_plan_t*> g_bound;
// what the bound plan stands for and needs from the mesh data the shims cannot read: the SIPG parameters / boundary-condition type the
// plan was set up with (checked against the caller's flux data), the node coordinates (source terms), the Schwarz smoother handle
struct FluxReg { double prefactor; int bc_type; };
std::map<const void*, FluxReg> g_flux;
struct CoordReg { const double* lob[3]; const double* quad[3]; };
std::map<const void*, CoordReg> g_coord;
struct SchwarzReg { d4est_hip_schwarz_t* sz; int iter; double atol, rtol; };
std::map<const void*, SchwarzReg> g_schwarz;
double* g_tr_h = nullptr;   // pinned staging of the transfer shims
double* g_tr_d = nullptr;
size_t g_tr_cap = 0;

int quad_type_of(const d4est_quadrature_t* q) {
  // d4est_quadrature_t begins with `d4est_quadrature_type_t quad_type` (src/Quadrature/d4est_quadrature.h:117-119)
  if (!q) return D4EST_HIP_QUAD_LEGENDRE;
  int t;
  std::memcpy(&t, q, sizeof(int));
  if (t == 0) return D4EST_HIP_QUAD_LEGENDRE;     // QUAD_TYPE_GAUSS_LEGENDRE
  if (t == 1) return D4EST_HIP_QUAD_LOBATTO;      // QUAD_TYPE_GAUSS_LEGENDRE_LOBATTO
  COMPAT_ABORT("quadrature type %d (compactified rules, disabled in the reference: d4est_quadrature.c:90-105) is not supported", t);
}

ElemCtx& elem_ctx(int deg, int deg_quad, int quad_type) {
  auto key = std::make_tuple(deg, deg_quad, quad_type);
  auto it = g_elem.find(key);
  if (it != g_elem.end()) return it->second;
  if (deg < 1 || deg_quad < 1) COMPAT_ABORT("element shim: deg %d / deg_quad %d", deg, deg_quad);
  ElemCtx c;
  int zero = 0;
  c.plan = d4est_hip_plan_create(1, &deg, &deg_quad, &zero, &zero, quad_type);
  c.N3 = (deg + 1) * (deg + 1) * (deg + 1);
  c.Q3 = (deg_quad + 1) * (deg_quad + 1) * (deg_quad + 1);
  c.N2 = (deg + 1) * (deg + 1);
  c.cap = c.N3 > c.Q3 ? c.N3 : c.Q3;
  c.h = (double*)d4est_hip_host_alloc(sizeof(double) * ((size_t)2 * c.cap + (size_t)10 * c.Q3));
  c.d_in = (double*)d4est_hip_malloc(sizeof(double) * c.cap);
  c.d_out = (double*)d4est_hip_malloc(sizeof(double) * c.cap);
  c.d_geo = (double*)d4est_hip_malloc(sizeof(double) * (size_t)10 * c.Q3);
  return g_elem.emplace(key, c).first->second;
}

// ---- QUAD_OBJECT_MORTAR: the (dim - 1)-dimensional forms of interpolate / apply_mass_matrix / apply_galerkin_integral.  The engine's
// operator path never needs them (mortar integrals live inside the fused face kernels), but the reference's estimators and mesh update
// call these entry points with mortar objects (src/Estimators/d4est_estimator_bi.c:85, src/Mesh/d4est_mortars.c), so a build that links
// this library in front must serve them: a 2-D tensor apply on the host with the engine's own 1-D tables -- compatibility, not speed.
struct Tab2 { std::vector<double> I, w; int N = 0, NQ = 0; };
std::map<std::tuple<int, int, int>, Tab2> g_tab2;
const Tab2& tab2(int deg, int deg_quad, int quad_type) {
  auto key = std::make_tuple(deg, deg_quad, quad_type);
  auto it = g_tab2.find(key);
  if (it != g_tab2.end()) return it->second;
  Tab2 t;
  t.N = deg + 1; t.NQ = deg_quad + 1;
  t.I.resize((size_t)t.NQ * t.N);
  t.w.resize(t.NQ);
  // src/Quadrature/d4est_quadrature_legendre.c:22-93 / d4est_quadrature_lobatto.c:23-93: interpolation to and weights of the rule
  d4est_hip_table(quad_type == D4EST_HIP_QUAD_LOBATTO ? D4EST_HIP_TABLE_P_PROLONG : D4EST_HIP_TABLE_LOBATTO_TO_GAUSS, deg, deg_quad, t.I.data());
  d4est_hip_table(quad_type == D4EST_HIP_QUAD_LOBATTO ? D4EST_HIP_TABLE_LOBATTO_WEIGHTS : D4EST_HIP_TABLE_GAUSS_WEIGHTS, deg_quad, 0, t.w.data());
  return g_tab2.emplace(key, t).first->second;
}
// out[NQ x NQ] = (I (x) I) in[N x N]  (first index fastest, as d4est_kron_A1A2x_nonsqr: d4est_quadrature.c:1002-1005)
void interp2(const Tab2& t, const double* in, double* out) {
  std::vector<double> tmp((size_t)t.NQ * t.N);
  for (int b = 0; b < t.N; ++b)
    for (int aq = 0; aq < t.NQ; ++aq) {
      double s = 0.0;
      for (int a = 0; a < t.N; ++a) s += t.I[(size_t)aq * t.N + a] * in[a + t.N * b];
      tmp[aq + (size_t)t.NQ * b] = s;
    }
  for (int bq = 0; bq < t.NQ; ++bq)
    for (int aq = 0; aq < t.NQ; ++aq) {
      double s = 0.0;
      for (int b = 0; b < t.N; ++b) s += t.I[(size_t)bq * t.N + b] * tmp[aq + (size_t)t.NQ * b];
      out[aq + (size_t)t.NQ * bq] = s;
    }
}
// out[N x N] = (I (x) I)^T (w (x) w . jac . in)[NQ x NQ]   (d4est_quadrature.c:185-209, :441-466 with dim = 2)
void galerkin2(const Tab2& t, const double* in_quad, const double* jac, double* out) {
  std::vector<double> f((size_t)t.NQ * t.NQ), tmp((size_t)t.N * t.NQ);
  for (int bq = 0; bq < t.NQ; ++bq)
    for (int aq = 0; aq < t.NQ; ++aq) f[aq + (size_t)t.NQ * bq] = (t.w[bq] * t.w[aq]) * jac[aq + (size_t)t.NQ * bq] * in_quad[aq + (size_t)t.NQ * bq];
  for (int bq = 0; bq < t.NQ; ++bq)
    for (int a = 0; a < t.N; ++a) {
      double s = 0.0;
      for (int aq = 0; aq < t.NQ; ++aq) s += t.I[(size_t)aq * t.N + a] * f[aq + (size_t)t.NQ * bq];
      tmp[a + (size_t)t.N * bq] = s;
    }
  for (int b = 0; b < t.N; ++b)
    for (int a = 0; a < t.N; ++a) {
      double s = 0.0;
      for (int bq = 0; bq < t.NQ; ++bq) s += t.I[(size_t)bq * t.N + b] * tmp[a + (size_t)t.N * bq];
      out[a + (size_t)t.N * b] = s;
    }
}

// the apply_lhs callback the operator-level shims stand for (d4est_hip_compat_bind_operator); nullptr: not registered, not checked
std::map<const void*, d4est_apply_operator_fcn_t> g_bound_lhs;
void check_fcns(const void* p4est, const d4est_elliptic_eqns_t* fcns, const char* who) {
  auto it = g_bound_lhs.find(p4est);
  if (it == g_bound_lhs.end() || !it->second) return;
  if (!fcns || fcns->apply_lhs != it->second)
    COMPAT_ABORT("%s: fcns->apply_lhs is not the operator registered for this p4est (d4est_hip_compat_bind_operator): the bound plan applies "
                 "the Laplacian + its zeroth-order term and would silently stand in for a different operator", who);
}

void need_dim3(int dim, const char* who) {
  if (dim != 3) COMPAT_ABORT("%s: dim = %d; the engine replaces the DIM = 3 (d8est) volume applies only", who, dim);
}
void need_volume(d4est_quadrature_object_type_t t, const char* who) {
  if (t != QUAD_OBJECT_VOLUME) COMPAT_ABORT("%s: QUAD_OBJECT_MORTAR objects stay inside the fused face kernels; volume objects only", who);
}

void upload(ElemCtx& c, const double* src, int n) {
  std::memcpy(c.h, src, sizeof(double) * n);
  d4est_hip_memcpy_h2d_async(c.plan, c.d_in, c.h, sizeof(double) * n);
}
void download(ElemCtx& c, double* dst, int n) {
  d4est_hip_memcpy_d2h_async(c.plan, c.h + c.cap, c.d_out, sizeof(double) * n);
  d4est_hip_plan_synchronize(c.plan);
  std::memcpy(dst, c.h + c.cap, sizeof(double) * n);
}
void upload_jacobian(ElemCtx& c, const double* jac) {
  double* hg = c.h + 2 * (size_t)c.cap;
  std::memcpy(hg, jac, sizeof(double) * c.Q3);
  d4est_hip_memcpy_h2d_async(c.plan, c.d_geo, hg, sizeof(double) * c.Q3);
  d4est_hip_plan_set_jacobian(c.plan, c.d_geo, 1);
}

d4est_hip_transfer_t* transfer_of(int hrefine, int degH, const int* degh) {
  std::array<int, 10> key{};
  key[0] = hrefine;
  key[1] = degH;
  for (int i = 0; i < 8; ++i) key[2 + i] = hrefine ? degh[i] : (i == 0 ? degh[0] : 0);
  auto it = g_transfer.find(key);
  if (it != g_transfer.end()) return it->second;
  int dh[8];
  for (int i = 0; i < 8; ++i) dh[i] = hrefine ? degh[i] : degh[0];
  d4est_hip_transfer_t* t = d4est_hip_transfer_create(1, &hrefine, &degH, dh);
  g_transfer[key] = t;
  return t;
}

// coarse / fine staging of the transfer shims: one pinned and one device buffer, grown (rarely) to the largest pair seen
void transfer_run(d4est_hip_transfer_t* t, int mode, const double* in, double* out) {
  const size_t nc = (size_t)d4est_hip_transfer_coarse_nodes(t), nf = (size_t)d4est_hip_transfer_fine_nodes(t);
  if (nc + nf > g_tr_cap) {
    if (g_tr_h) { d4est_hip_host_free(g_tr_h); d4est_hip_free(g_tr_d); }
    g_tr_cap = 2 * (nc + nf);
    g_tr_h = (double*)d4est_hip_host_alloc(sizeof(double) * g_tr_cap);
    g_tr_d = (double*)d4est_hip_malloc(sizeof(double) * g_tr_cap);
  }
  double *dc = g_tr_d, *df = g_tr_d + nc, *hc = g_tr_h, *hf = g_tr_h + nc;
  if (mode == 0) {            // prolong: coarse -> fine
    std::memcpy(hc, in, sizeof(double) * nc);
    d4est_hip_memcpy_h2d(dc, hc, sizeof(double) * nc);
    d4est_hip_transfer_prolong(t, dc, df);
    d4est_hip_device_synchronize();
    d4est_hip_memcpy_d2h(hf, df, sizeof(double) * nf);
    std::memcpy(out, hf, sizeof(double) * nf);
  } else {                    // 1: prolong-transpose, 2: L2 projection; fine -> coarse
    std::memcpy(hf, in, sizeof(double) * nf);
    d4est_hip_memcpy_h2d(df, hf, sizeof(double) * nf);
    if (mode == 1) d4est_hip_transfer_restrict(t, df, dc);
    else d4est_hip_transfer_project(t, df, dc);
    d4est_hip_device_synchronize();
    d4est_hip_memcpy_d2h(hc, dc, sizeof(double) * nc);
    std::memcpy(out, hc, sizeof(double) * nc);
  }
}


// ---- (dim - 1)-dimensional transfers on the host: what d4est_mortars_project_side_onto_mortar_space / _mass_mortar_onto_side and the
// estimators ask of apply_p_prolong & co. with dim = 2 (src/Mesh/d4est_mortars.c:510-598).  ops: per direction a row-major (n_out x n_in)
// matrix; out[a + n_out b] = sum op_y[b][B] op_x[a][A] in[A + n_in B]
void tensor2(const double* opx, const double* opy, int n_in, int n_out, const double* in, double* out, bool accumulate) {
  std::vector<double> tmp((size_t)n_out * n_in);
  for (int B = 0; B < n_in; ++B)
    for (int a = 0; a < n_out; ++a) {
      double s = 0.0;
      for (int A = 0; A < n_in; ++A) s += opx[(size_t)a * n_in + A] * in[A + (size_t)n_in * B];
      tmp[a + (size_t)n_out * B] = s;
    }
  for (int b = 0; b < n_out; ++b)
    for (int a = 0; a < n_out; ++a) {
      double s = 0.0;
      for (int B = 0; B < n_in; ++B) s += opy[(size_t)b * n_in + B] * tmp[a + (size_t)n_out * B];
      if (accumulate) out[a + (size_t)n_out * b] += s;
      else out[a + (size_t)n_out * b] = s;
    }
}
std::vector<double> table_of(int id, int deg_a, int deg_b) {
  std::vector<double> t((size_t)d4est_hip_table(id, deg_a, deg_b, nullptr));
  d4est_hip_table(id, deg_a, deg_b, t.data());
  return t;
}
std::vector<double> transposed(const double* m, int rows, int cols) {
  std::vector<double> t((size_t)rows * cols);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = m[(size_t)r * cols + c];
  return t;
}
// mode 0 prolong (coarse -> fine), 1 prolong-transpose, 2 L2 projection (fine -> coarse); children = 1 or 4
void transfer2(int mode, int children, int degH, const int* degh, const double* in, double* out) {
  const int nH = degH + 1;
  size_t stride = 0;
  if (mode != 0) std::fill(out, out + (size_t)nH * nH, 0.0);
  for (int c = 0; c < children; ++c) {
    const int nh = degh[c] + 1;
    if (nh < nH) COMPAT_ABORT("transfer (dim 2): degh %d < degH %d (d4est_operators.c:379)", degh[c], degH);
    const size_t half = (size_t)nh * nH;
    const int hx = children == 1 ? 0 : (c & 1), hy = children == 1 ? 0 : ((c >> 1) & 1);
    if (mode == 2) {
      const std::vector<double> R = table_of(children == 1 ? D4EST_HIP_TABLE_P_RESTRICT : D4EST_HIP_TABLE_HP_RESTRICT, degH, degh[c]);   // (nH x nh) per half
      tensor2(R.data() + hx * half, R.data() + hy * half, nh, nH, in + stride, out, true);
    } else {
      const std::vector<double> P = table_of(children == 1 ? D4EST_HIP_TABLE_P_PROLONG : D4EST_HIP_TABLE_HP_PROLONG, degH, degh[c]);    // (nh x nH) per half
      if (mode == 0) tensor2(P.data() + hx * half, P.data() + hy * half, nH, nh, in, out + stride, false);
      else {
        const std::vector<double> Tx = transposed(P.data() + hx * half, nh, nH), Ty = transposed(P.data() + hy * half, nh, nH);
        tensor2(Tx.data(), Ty.data(), nh, nH, in + stride, out, true);
      }
    }
    stride += (size_t)nh * nh;
  }
}
void need_dim23(int dim, const char* who) {
  if (dim != 3 && dim != 2) COMPAT_ABORT("%s: dim = %d; volume (3) and face (2) objects of the d8est build only", who, dim);
}

// the caller's flux data against what the bound plan was set up with (d4est_hip_compat_bind_flux): a plan that stands in for a
// DIFFERENT operator must not answer silently.  prefactor: first member of d4est_laplacian_flux_sipg_params_t.
void check_flux(const void* p4est, int flux_type, const void* sipg_params, int bc_type, const char* who) {
  auto it = g_flux.find(p4est);
  if (it == g_flux.end()) return;
  if (flux_type != 0) COMPAT_ABORT("%s: flux_type %d; the plan applies FLUX_SIPG", who, flux_type);
  double pre = 0.0;
  if (!sipg_params) COMPAT_ABORT("%s: flux_fcn_data->flux_data is NULL", who);
  std::memcpy(&pre, sipg_params, sizeof(double));
  if (pre != it->second.prefactor)
    COMPAT_ABORT("%s: flux_fcn_data carries sipg_penalty_prefactor %.17g, the bound plan was set up with %.17g (d4est_hip_compat_bind_flux)", who, pre, it->second.prefactor);
  if (bc_type != it->second.bc_type)
    COMPAT_ABORT("%s: flux_fcn_data carries bc_type %d, the bound plan was set up with %d (BC_ROBIN 0 / BC_DIRICHLET 1)", who, bc_type, it->second.bc_type);
}

d4est_hip_plan_t* bound(const void* p4est, const char* who) {
  auto it = g_bound.find(p4est);
  if (it == g_bound.end() || !it->second) COMPAT_ABORT("%s: no plan bound to p4est %p (d4est_hip_compat_bind_mesh)", who, p4est);
  return it->second;
}

}  // namespace

extern "C" {

// ---- src/Quadrature/d4est_quadrature.c:263-382 -----------------------------------------------------------------------------
void d4est_quadrature_apply_stiffness_matrix(d4est_operators_t*, d4est_quadrature_t* d4est_quadrature, d4est_geometry_t*, void*,
                                             d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t, double* in,
                                             int deg_lobatto, double* jac_quad, double* rst_xyz[3][3], int deg_quad, double* out) {
  need_volume(object_type, "d4est_quadrature_apply_stiffness_matrix");
  ElemCtx& c = elem_ctx(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature));
  double* hg = c.h + 2 * (size_t)c.cap;
  std::memcpy(hg, jac_quad, sizeof(double) * c.Q3);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) std::memcpy(hg + (size_t)(1 + 3 * i + j) * c.Q3, rst_xyz[i][j], sizeof(double) * c.Q3);
  d4est_hip_memcpy_h2d_async(c.plan, c.d_geo, hg, sizeof(double) * (size_t)10 * c.Q3);
  upload(c, in, c.N3);
  d4est_hip_plan_set_geometry(c.plan, c.d_geo, c.d_geo + c.Q3, 1);
  d4est_hip_apply_stiffness_matrix(c.plan, c.d_in, c.d_out);
  download(c, out, c.N3);
}

// :385-477
void d4est_quadrature_apply_mass_matrix(d4est_operators_t*, d4est_geometry_t*, d4est_quadrature_t* d4est_quadrature, void*,
                                        d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t, double* in,
                                        int deg_lobatto, double* jac_quad, int deg_quad, double* out) {
  if (object_type == QUAD_OBJECT_MORTAR) {   // dim - 1: V^T W J V in on the face
    const Tab2& t = tab2(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature));
    std::vector<double> q((size_t)t.NQ * t.NQ);
    interp2(t, in, q.data());
    galerkin2(t, q.data(), jac_quad, out);
    return;
  }
  need_volume(object_type, "d4est_quadrature_apply_mass_matrix");
  ElemCtx& c = elem_ctx(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature));
  upload_jacobian(c, jac_quad);
  upload(c, in, c.N3);
  d4est_hip_apply_mass_matrix(c.plan, c.d_in, c.d_out);
  download(c, out, c.N3);
}

// :142-213
void d4est_quadrature_apply_galerkin_integral(d4est_operators_t*, d4est_geometry_t*, d4est_quadrature_t* d4est_quadrature, void*,
                                              d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t,
                                              double* in_quad, int deg_lobatto, double* jac_quad, int deg_quad, double* out) {
  if (object_type == QUAD_OBJECT_MORTAR) {
    galerkin2(tab2(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature)), in_quad, jac_quad, out);
    return;
  }
  need_volume(object_type, "d4est_quadrature_apply_galerkin_integral");
  ElemCtx& c = elem_ctx(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature));
  upload_jacobian(c, jac_quad);
  upload(c, in_quad, c.Q3);
  d4est_hip_apply_galerkin_integral(c.plan, c.d_in, c.d_out);
  download(c, out, c.N3);
}

// :966-1016
void d4est_quadrature_interpolate(d4est_operators_t*, d4est_quadrature_t* d4est_quadrature, d4est_geometry_t*, void*,
                                  d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t, double* u_lobatto_in,
                                  int deg_lobatto, double* u_quad_out, int deg_quad) {
  if (object_type == QUAD_OBJECT_MORTAR) {
    interp2(tab2(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature)), u_lobatto_in, u_quad_out);
    return;
  }
  need_volume(object_type, "d4est_quadrature_interpolate");
  ElemCtx& c = elem_ctx(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature));
  upload(c, u_lobatto_in, c.N3);
  d4est_hip_interpolate(c.plan, c.d_in, c.d_out);
  download(c, u_quad_out, c.Q3);
}

// :1222-1331 (always Gauss-Legendre, deg_Gauss == deg_Lobatto asserted at :1233)
void d4est_quadrature_apply_inverse_mass_matrix(d4est_operators_t*, double* in, int deg_Lobatto, double* jac_Gauss, int deg_Gauss, int dim,
                                                double* out) {
  need_dim3(dim, "d4est_quadrature_apply_inverse_mass_matrix");
  ElemCtx& c = elem_ctx(deg_Lobatto, deg_Gauss, D4EST_HIP_QUAD_LEGENDRE);
  upload_jacobian(c, jac_Gauss);
  upload(c, in, c.N3);
  d4est_hip_apply_inverse_mass_matrix(c.plan, c.d_in, c.d_out);
  download(c, out, c.N3);
}

// ---- the callback-taking mass terms of the nonlinear problems (src/Quadrature/d4est_quadrature.c:593-774, :776-936), which the
// Problem files call directly (e.g. constant_density_star_fcns.h:407, :575).  The user functions are HOST function pointers evaluated
// per node, so that part runs on the host: f(x, u) at the quadrature nodes (u interpolated by the shim above), or at the Lobatto nodes
// and then interpolated (interpolate_f); the integrals are the mass / galerkin shims.  A Newton-Krylov loop should instead evaluate f
// once per Newton step and hand it over with d4est_hip_plan_set_lhs_coefficient: this is the compatibility path.
static double compat_identity(double, double, double, double u, void*) { return u; }   // identity_fcn, src/Mesh/d4est_xyz_functions.c:38-48: f(x, u) = u

// product of the two user functions at n nodes: at the quadrature nodes (fields interpolated first) or, with interpolate_f, at the
// Lobatto nodes and interpolated afterwards; `scale` (may be NULL) multiplies the result node by node (the Jacobian)
static void fofu_fofv(d4est_operators_t* ops, d4est_geometry_t* geom, d4est_quadrature_t* quad, void* object,
                      d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t integrand_type, double* u, double* v,
                      int deg_lobatto, double* xyz_quad[3], int deg_quad, d4est_xyzu_fcn_t fofu_fcn, void* fofu_ctx, d4est_xyzu_fcn_t fofv_fcn,
                      void* fofv_ctx, int interpolate_f, double* xyz_lobatto[3], const double* scale, std::vector<double>& out) {
  const int dim = (object_type == QUAD_OBJECT_MORTAR) ? 2 : 3;
  int nq = 1, nl = 1;
  for (int i = 0; i < dim; ++i) { nq *= deg_quad + 1; nl *= deg_lobatto + 1; }
  const bool use_u = (u != nullptr) || (fofu_fcn != nullptr), use_v = (v != nullptr) || (fofv_fcn != nullptr);
  if (!fofu_fcn) fofu_fcn = compat_identity;
  if (!fofv_fcn) fofv_fcn = compat_identity;
  out.assign(nq, 1.0);
  if (!interpolate_f) {
    std::vector<double> uq, vq;
    if (u) { uq.resize(nq); d4est_quadrature_interpolate(ops, quad, geom, object, object_type, integrand_type, u, deg_lobatto, uq.data(), deg_quad); }
    if (v) { vq.resize(nq); d4est_quadrature_interpolate(ops, quad, geom, object, object_type, integrand_type, v, deg_lobatto, vq.data(), deg_quad); }
    for (int i = 0; i < nq; ++i) {
      const double z = xyz_quad[2][i];   // d8est build: z is always passed (d4est_quadrature.c:660-690 under #if P4EST_DIM==3), mortar objects included
      double f = scale ? scale[i] : 1.0;
      if (use_u) f *= fofu_fcn(xyz_quad[0][i], xyz_quad[1][i], z, u ? uq[i] : 0.0, fofu_ctx);
      if (use_v) f *= fofv_fcn(xyz_quad[0][i], xyz_quad[1][i], z, v ? vq[i] : 0.0, fofv_ctx);
      out[i] = f;
    }
  } else {
    if (!xyz_lobatto) COMPAT_ABORT("interpolate_f == 1, but xyz_lobatto == NULL");
    std::vector<double> fl(nl, 1.0);
    for (int i = 0; i < nl; ++i) {
      const double z = xyz_lobatto[2][i];
      if (use_u) fl[i] *= fofu_fcn(xyz_lobatto[0][i], xyz_lobatto[1][i], z, u ? u[i] : 0.0, fofu_ctx);
      if (use_v) fl[i] *= fofv_fcn(xyz_lobatto[0][i], xyz_lobatto[1][i], z, v ? v[i] : 0.0, fofv_ctx);
    }
    d4est_quadrature_interpolate(ops, quad, geom, object, object_type, integrand_type, fl.data(), deg_lobatto, out.data(), deg_quad);
    if (scale)
      for (int i = 0; i < nq; ++i) out[i] *= scale[i];
  }
}

// :1143-1186: the dense element matrix, column i = the mass apply of the i-th unit vector (d4est_linalg_set_column).  Volume objects: all
// columns of the one-element plan in one device call (d4est_hip_compute_weighted_mass_blocks); mortar objects (dim - 1): on the host
void d4est_quadrature_compute_mass_matrix(d4est_operators_t*, d4est_geometry_t*, d4est_quadrature_t* d4est_quadrature, void*,
                                          d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t, int deg_lobatto,
                                          double* jac_quad, int deg_quad, double* out) {
  if (object_type == QUAD_OBJECT_MORTAR) {
    const Tab2& t = tab2(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature));
    const int n = t.N * t.N;
    std::vector<double> u((size_t)n, 0.0), q((size_t)t.NQ * t.NQ), Mu((size_t)n);
    for (int i = 0; i < n; ++i) {
      u[i] = 1.0;
      interp2(t, u.data(), q.data());
      galerkin2(t, q.data(), jac_quad, Mu.data());
      for (int r = 0; r < n; ++r) out[(size_t)r * n + i] = Mu[r];
      u[i] = 0.0;
    }
    return;
  }
  need_volume(object_type, "d4est_quadrature_compute_mass_matrix");
  ElemCtx& c = elem_ctx(deg_lobatto, deg_quad, quad_type_of(d4est_quadrature));
  upload_jacobian(c, jac_quad);
  const size_t nn = (size_t)c.N3 * c.N3;
  double* d_blk = (double*)d4est_hip_malloc(sizeof(double) * nn);
  d4est_hip_compute_weighted_mass_blocks(c.plan, nullptr, d_blk);
  d4est_hip_memcpy_d2h(out, d_blk, sizeof(double) * nn);
  d4est_hip_free(d_blk);
}

// out = V^T W [J f(x,u) f(x,v)] V vec (QUAD_APPLY_MATRIX), or the dense element matrix of that operator (QUAD_COMPUTE_MATRIX: what
// d4est_solver_multigrid_matrix_setup_fofufofvlilj_operator asks for per element, Solver/d4est_solver_multigrid_matrix_operator.c:215-238)
void d4est_quadrature_apply_fofufofvlilj(d4est_operators_t* d4est_ops, d4est_geometry_t* d4est_geom, d4est_quadrature_t* d4est_quad, void* object,
                                         d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t integrand_type,
                                         double* vec, double* u, double* v, int deg_lobatto, double* xyz_quad[3], double* jac_quad, int deg_quad,
                                         double* out, d4est_xyzu_fcn_t fofu_fcn, void* fofu_ctx, d4est_xyzu_fcn_t fofv_fcn, void* fofv_ctx,
                                         d4est_quadrature_apply_or_compute_matrix_t apply_or_compute_matrix, int interpolate_f,
                                         double* xyz_lobatto[3]) {
  if (apply_or_compute_matrix != QUAD_APPLY_MATRIX && apply_or_compute_matrix != QUAD_COMPUTE_MATRIX)
    COMPAT_ABORT("d4est_quadrature_apply_fofufofvlilj: Not a supported option");   // d4est_quadrature.c:762
  if (apply_or_compute_matrix == QUAD_APPLY_MATRIX && !vec) COMPAT_ABORT("d4est_quadrature_apply_fofufofvlilj: vec == NULL");
  std::vector<double> fj;
  fofu_fofv(d4est_ops, d4est_geom, d4est_quad, object, object_type, integrand_type, u, v, deg_lobatto, xyz_quad, deg_quad, fofu_fcn, fofu_ctx,
            fofv_fcn, fofv_ctx, interpolate_f, xyz_lobatto, jac_quad, fj);
  if (apply_or_compute_matrix == QUAD_APPLY_MATRIX)
    d4est_quadrature_apply_mass_matrix(d4est_ops, d4est_geom, d4est_quad, object, object_type, integrand_type, vec, deg_lobatto, fj.data(), deg_quad, out);
  else
    d4est_quadrature_compute_mass_matrix(d4est_ops, d4est_geom, d4est_quad, object, object_type, integrand_type, deg_lobatto, fj.data(), deg_quad, out);
}

// out = V^T W J [f(x,u) f(x,v)]
void d4est_quadrature_apply_fofufofvlj(d4est_operators_t* d4est_ops, d4est_geometry_t* d4est_geom, d4est_quadrature_t* d4est_quad, void* object,
                                       d4est_quadrature_object_type_t object_type, d4est_quadrature_integrand_type_t integrand_type, double* u,
                                       double* v, int deg_lobatto, double* jac_quad, double* xyz_quad[3], int deg_quad, double* out,
                                       d4est_xyzu_fcn_t fofu_fcn, void* fofu_ctx, d4est_xyzu_fcn_t fofv_fcn, void* fofv_ctx, int interpolate_f,
                                       double* xyz_lobatto[3]) {
  std::vector<double> f;
  fofu_fofv(d4est_ops, d4est_geom, d4est_quad, object, object_type, integrand_type, u, v, deg_lobatto, xyz_quad, deg_quad, fofu_fcn, fofu_ctx,
            fofv_fcn, fofv_ctx, interpolate_f, xyz_lobatto, nullptr, f);
  d4est_quadrature_apply_galerkin_integral(d4est_ops, d4est_geom, d4est_quad, object, object_type, integrand_type, f.data(), deg_lobatto, jac_quad,
                                           deg_quad, out);
}

// ---- src/dGMath/d4est_operators.c -------------------------------------------------------------------------------------------
void d4est_operators_apply_dij(d4est_operators_t*, double* in, int dim, int deg, int dir, double* out) {   // :1385-1410
  need_dim3(dim, "d4est_operators_apply_dij");
  ElemCtx& c = elem_ctx(deg, deg, D4EST_HIP_QUAD_LEGENDRE);
  upload(c, in, c.N3);
  d4est_hip_apply_dij(c.plan, c.d_in, dir, c.d_out);
  download(c, out, c.N3);
}
void d4est_operators_apply_dij_transpose(d4est_operators_t*, double* in, int dim, int deg, int dir, double* out) {   // :2259-2284
  need_dim3(dim, "d4est_operators_apply_dij_transpose");
  ElemCtx& c = elem_ctx(deg, deg, D4EST_HIP_QUAD_LEGENDRE);
  upload(c, in, c.N3);
  d4est_hip_apply_dij_transpose(c.plan, c.d_in, dir, c.d_out);
  download(c, out, c.N3);
}
void d4est_operators_apply_lift(d4est_operators_t*, double* in, int dim, int deg, int face, double* out) {   // :1454-1519
  need_dim3(dim, "d4est_operators_apply_lift");
  ElemCtx& c = elem_ctx(deg, deg, D4EST_HIP_QUAD_LEGENDRE);
  upload(c, in, c.N2);
  d4est_hip_apply_lift(c.plan, c.d_in, face, c.d_out);
  download(c, out, c.N3);
}
void d4est_operators_apply_slicer(d4est_operators_t*, double* in, int dim, int face, int deg, double* out) {   // :1521-1582
  need_dim3(dim, "d4est_operators_apply_slicer");
  ElemCtx& c = elem_ctx(deg, deg, D4EST_HIP_QUAD_LEGENDRE);
  upload(c, in, c.N3);
  d4est_hip_apply_slicer(c.plan, c.d_in, face, c.d_out);
  download(c, out, c.N2);
}
void d4est_operators_apply_mij(d4est_operators_t*, double* in, int dim, int deg, double* out) {   // :891-908
  need_dim3(dim, "d4est_operators_apply_mij");
  ElemCtx& c = elem_ctx(deg, deg, D4EST_HIP_QUAD_LEGENDRE);
  upload(c, in, c.N3);
  d4est_hip_apply_mij(c.plan, c.d_in, c.d_out);
  download(c, out, c.N3);
}
void d4est_operators_apply_invmij(d4est_operators_t*, double* in, int dim, int deg, double* out) {   // :910-928
  need_dim3(dim, "d4est_operators_apply_invmij");
  ElemCtx& c = elem_ctx(deg, deg, D4EST_HIP_QUAD_LEGENDRE);
  upload(c, in, c.N3);
  d4est_hip_apply_invmij(c.plan, c.d_in, c.d_out);
  download(c, out, c.N3);
}
void d4est_operators_apply_p_prolong(d4est_operators_t*, double* in, int degH, int dim, int degh, double* out) {   // :1107-1132
  need_dim23(dim, "d4est_operators_apply_p_prolong");
  if (dim == 2) { transfer2(0, 1, degH, &degh, in, out); return; }
  transfer_run(transfer_of(0, degH, &degh), 0, in, out);
}
void d4est_operators_apply_hp_prolong(d4est_operators_t*, double* in, int degH, int dim, int* degh, double* out) {   // :1091-1105
  need_dim23(dim, "d4est_operators_apply_hp_prolong");
  if (dim == 2) { transfer2(0, 4, degH, degh, in, out); return; }
  transfer_run(transfer_of(1, degH, degh), 0, in, out);
}
void d4est_operators_apply_p_restrict(d4est_operators_t*, double* in, int degh, int dim, int degH, double* out) {   // :1205-1230
  need_dim23(dim, "d4est_operators_apply_p_restrict");
  if (dim == 2) { transfer2(2, 1, degH, &degh, in, out); return; }
  transfer_run(transfer_of(0, degH, &degh), 2, in, out);
}
void d4est_operators_apply_hp_restrict(d4est_operators_t*, double* in, int* degh, int dim, int degH, double* out) {   // :1275-1297
  need_dim23(dim, "d4est_operators_apply_hp_restrict");
  if (dim == 2) { transfer2(2, 4, degH, degh, in, out); return; }
  transfer_run(transfer_of(1, degH, degh), 2, in, out);
}
// :572-605: the dense prolongation (sum_i (degh_i+1)^3) x (degH+1)^3, column i = P e_i
void d4est_operators_compute_prolong_matrix(d4est_operators_t* ops, int degH, int dim, int* degh, int children, double* prolong_mat) {
  need_dim3(dim, "d4est_operators_compute_prolong_matrix");
  if (children != 1 && children != 8) COMPAT_ABORT("d4est_operators_compute_prolong_matrix: children = %d", children);
  const size_t nH = (size_t)(degH + 1) * (degH + 1) * (degH + 1);
  size_t nh = 0;
  for (int i = 0; i < children; ++i) nh += (size_t)(degh[i] + 1) * (degh[i] + 1) * (degh[i] + 1);
  std::vector<double> u(nH, 0.0), Mu(nh);
  for (size_t i = 0; i < nH; ++i) {
    u[i] = 1.0;
    if (children == 8) d4est_operators_apply_hp_prolong(ops, u.data(), degH, dim, degh, Mu.data());
    else d4est_operators_apply_p_prolong(ops, u.data(), degH, dim, degh[0], Mu.data());
    for (size_t r = 0; r < nh; ++r) prolong_mat[r * nH + i] = Mu[r];
    u[i] = 0.0;
  }
}
// :608-667: PT_mat_P = sum_i P_i^T mat_i P_i on the device (d4est_hip_transfer_galerkin_blocks of a one-item transfer).  With eight
// children the reference's own arithmetic reads a window of the transposed stacked prolongation as the left factor (:651), which is not
// P_i^T; D4EST_HIP_REFERENCE_PT_WINDOW=1 in the environment reproduces that to the letter, the default is the Galerkin product
void d4est_operators_compute_PT_mat_P(d4est_operators_t*, double* mat, int degH, int dim, int* degh, int children, double* PT_mat_P) {
  need_dim3(dim, "d4est_operators_compute_PT_mat_P");
  if (children != 1 && children != 8) COMPAT_ABORT("d4est_operators_compute_PT_mat_P: children = %d", children);
  static const bool literal = std::getenv("D4EST_HIP_REFERENCE_PT_WINDOW") != nullptr && std::atoi(std::getenv("D4EST_HIP_REFERENCE_PT_WINDOW")) != 0;
  d4est_hip_transfer_t* t = transfer_of(children == 8 ? 1 : 0, degH, degh);
  const size_t nf = (size_t)d4est_hip_transfer_fine_matrix_nodes(t), nc = (size_t)d4est_hip_transfer_coarse_matrix_nodes(t);
  double* d_f = (double*)d4est_hip_malloc(sizeof(double) * nf);
  double* d_c = (double*)d4est_hip_malloc(sizeof(double) * nc);
  d4est_hip_memcpy_h2d(d_f, mat, sizeof(double) * nf);
  d4est_hip_transfer_galerkin_blocks(t, d_f, d_c, literal ? 1 : 0);
  d4est_hip_device_synchronize();
  d4est_hip_memcpy_d2h(PT_mat_P, d_c, sizeof(double) * nc);
  d4est_hip_free(d_f);
  d4est_hip_free(d_c);
}
void d4est_operators_apply_p_prolong_transpose(d4est_operators_t*, double* in, int degh, int dim, int degH, double* out) {   // :1719-1749
  need_dim23(dim, "d4est_operators_apply_p_prolong_transpose");
  if (dim == 2) { transfer2(1, 1, degH, &degh, in, out); return; }
  transfer_run(transfer_of(0, degH, &degh), 1, in, out);
}
void d4est_operators_apply_hp_prolong_transpose(d4est_operators_t*, double* in, int* degh, int dim, int degH, double* out) {   // :1689-1717
  need_dim23(dim, "d4est_operators_apply_hp_prolong_transpose");
  if (dim == 2) { transfer2(1, 4, degH, degh, in, out); return; }
  transfer_run(transfer_of(1, degH, degh), 1, in, out);
}

// ---- operator / smoother level: the plan bound to the p4est ------------------------------------------------------------------
void d4est_laplacian_apply_stiffness_matrix(p4est_t* p4est, d4est_operators_t*, d4est_geometry_t*, d4est_quadrature_t*, d4est_mesh_data_t*,
                                            double* u, double* Au, int local_nodes, int which_field) {   // dGMath/d4est_laplacian.c:198-234
  d4est_hip_plan_t* plan = bound(p4est, "d4est_laplacian_apply_stiffness_matrix");
  if (local_nodes != d4est_hip_plan_local_nodes(plan)) COMPAT_ABORT("d4est_laplacian_apply_stiffness_matrix: local_nodes %d != plan's %d", local_nodes, d4est_hip_plan_local_nodes(plan));
  d4est_hip_apply_stiffness_matrix_host(plan, u + (size_t)which_field * local_nodes, Au + (size_t)which_field * local_nodes);
}

void d4est_laplacian_apply_aij(p4est_t* p4est, d4est_ghost_t*, d4est_ghost_data_t*, d4est_elliptic_data_t* d, d4est_laplacian_flux_data_t* fd,
                               d4est_operators_t*, d4est_geometry_t*, d4est_quadrature_t*, d4est_mesh_data_t*, int which_field) {   // :318-417
  d4est_hip_plan_t* plan = bound(p4est, "d4est_laplacian_apply_aij");
  if (fd) check_flux(p4est, fd->flux_type, fd->flux_data, fd->bc_type, "d4est_laplacian_apply_aij");
  if (!d || d->local_nodes != d4est_hip_plan_local_nodes(plan)) COMPAT_ABORT("d4est_laplacian_apply_aij: elliptic data does not match the bound plan");
  const size_t off = (size_t)which_field * d->local_nodes;   // :364-366: which_field * local_nodes
  d4est_hip_apply_aij_host(plan, d->u + off, d->Au + off);
}

// the reference's second implementation of the same operator (one visit per face, both sides accumulated): same applies here
void d4est_laplacian_with_opt_apply_stiffness_matrix(p4est_t* p4est, d4est_operators_t* ops, d4est_geometry_t* geom, d4est_quadrature_t* quad,
                                                     d4est_mesh_data_t* factors, double* u, double* Au, int local_nodes, int which_field) {
  d4est_laplacian_apply_stiffness_matrix(p4est, ops, geom, quad, factors, u, Au, local_nodes, which_field);   // d4est_laplacian_with_opt.c:145-209
}

void d4est_laplacian_with_opt_apply_aij(p4est_t* p4est, d4est_ghost_t*, d4est_ghost_data_t*, d4est_elliptic_data_t* d,
                                        d4est_laplacian_with_opt_flux_data_t* fd, d4est_operators_t*, d4est_geometry_t*, d4est_quadrature_t*,
                                        d4est_mesh_data_t*, int which_field) {   // d4est_laplacian_with_opt.c
  d4est_hip_plan_t* plan = bound(p4est, "d4est_laplacian_with_opt_apply_aij");
  if (fd) check_flux(p4est, fd->flux_type, fd->flux_data, fd->bc_type, "d4est_laplacian_with_opt_apply_aij");
  if (!d || d->local_nodes != d4est_hip_plan_local_nodes(plan)) COMPAT_ABORT("d4est_laplacian_with_opt_apply_aij: elliptic data does not match the bound plan");
  const size_t off = (size_t)which_field * d->local_nodes;
  d4est_hip_apply_aij_host(plan, d->u + off, d->Au + off);
}

void d4est_solver_multigrid_smoother_cheby_iterate_aux(p4est_t* p4est, d4est_operators_t*, d4est_geometry_t*, d4est_quadrature_t*,
                                                       d4est_mesh_data_t*, d4est_ghost_t*, d4est_ghost_data_t*, d4est_elliptic_data_t* vecs,
                                                       d4est_elliptic_eqns_t* fcns, double* r, int iter, double lmin, double lmax,
                                                       int /*print_residual_norm*/, int /*mg_level*/, int compute_residual_at_end) {
  d4est_hip_plan_t* plan = bound(p4est, "d4est_solver_multigrid_smoother_cheby_iterate_aux");
  check_fcns(p4est, fcns, "d4est_solver_multigrid_smoother_cheby_iterate_aux");
  if (!vecs || vecs->local_nodes != d4est_hip_plan_local_nodes(plan)) COMPAT_ABORT("cheby_iterate_aux: elliptic data does not match the bound plan");
  d4est_hip_cheby_iterate_host(plan, vecs->u, vecs->rhs, vecs->Au, r, iter, lmin, lmax, compute_residual_at_end);
}

void cg_eigs(p4est_t* p4est, d4est_elliptic_data_t* vecs, d4est_elliptic_eqns_t* fcns, d4est_ghost_t*, d4est_ghost_data_t*, d4est_operators_t*,
             d4est_geometry_t*, d4est_quadrature_t*, d4est_mesh_data_t*, int imax, int /*print_spectral_bound_iterations*/, int use_new,
             double* spectral_bound) {
  d4est_hip_plan_t* plan = bound(p4est, "cg_eigs");
  check_fcns(p4est, fcns, "cg_eigs");
  if (!vecs || vecs->local_nodes != d4est_hip_plan_local_nodes(plan)) COMPAT_ABORT("cg_eigs: elliptic data does not match the bound plan");
  const double b = d4est_hip_cg_eigs_host(plan, vecs->u, vecs->rhs, vecs->Au, imax, use_new, nullptr);
  if (spectral_bound) *spectral_bound = b;
}

void d4est_hip_compat_bind_mesh(const void* p4est, d4est_hip_plan_t* plan) {
  if (plan) g_bound[p4est] = plan;
  else { g_bound.erase(p4est); g_bound_lhs.erase(p4est); g_flux.erase(p4est); g_coord.erase(p4est); g_schwarz.erase(p4est); }
}
void d4est_hip_compat_bind_operator(const void* p4est, d4est_apply_operator_fcn_t apply_lhs) {
  if (apply_lhs) g_bound_lhs[p4est] = apply_lhs;
  else g_bound_lhs.erase(p4est);
}

// rhs = M f - A(0) on the bound plan (d4est_laplacian_build_rhs_with_strong_bc, src/dGMath/d4est_laplacian.c:16-140); the caller
// evaluates the source term with d4est_mesh_init_field and sets the plan's boundary data (see the header)
void d4est_hip_compat_build_rhs_with_strong_bc(const void* p4est, d4est_elliptic_data_t* prob_vecs, double* rhs, const double* f,
                                               int init_option, int which_field) {
  d4est_hip_plan_t* plan = bound(p4est, "d4est_hip_compat_build_rhs_with_strong_bc");
  if (!prob_vecs || prob_vecs->local_nodes != d4est_hip_plan_local_nodes(plan)) COMPAT_ABORT("build_rhs_with_strong_bc: elliptic data does not match the bound plan");
  // d4est_mesh_init_field_option_t (src/Mesh/d4est_mesh.h:19): INIT_FIELD_NOT_SET = 0, INIT_FIELD_ON_LOBATTO = 1, INIT_FIELD_ON_QUAD = 2
  if (init_option != 1 && init_option != 2) COMPAT_ABORT("build_rhs_with_strong_bc: init_option %d is not a supported init option (INIT_FIELD_ON_LOBATTO = 1 / INIT_FIELD_ON_QUAD = 2)", init_option);
  d4est_hip_build_rhs_with_strong_bc_host(plan, f, init_option == 2, rhs + (size_t)which_field * prob_vecs->local_nodes);
}

// ---- registrations that let the reference-named entries below work without reading the reference's mesh structs ----------------
void d4est_hip_compat_bind_flux(const void* p4est, double sipg_penalty_prefactor, int bc_type) {
  g_flux[p4est] = FluxReg{sipg_penalty_prefactor, bc_type};
}
void d4est_hip_compat_bind_coordinates(const void* p4est, double* xyz_lobatto[3], double* xyz_quad[3]) {
  CoordReg c{};
  for (int d = 0; d < 3; ++d) { c.lob[d] = xyz_lobatto ? xyz_lobatto[d] : nullptr; c.quad[d] = xyz_quad ? xyz_quad[d] : nullptr; }
  g_coord[p4est] = c;
}
void d4est_hip_compat_bind_schwarz(const void* p4est, d4est_hip_schwarz_t* sz, int subdomain_iter, double subdomain_atol, double subdomain_rtol) {
  if (sz) g_schwarz[p4est] = SchwarzReg{sz, subdomain_iter, subdomain_atol, subdomain_rtol};
  else g_schwarz.erase(p4est);
}

// src/dGMath/d4est_laplacian.c:16-140 with the reference's own argument list.  The source callback is evaluated here, on the host, at the
// node coordinates registered for this p4est (d4est_factors->xyz / ->xyz_quad: d4est_mesh_init_field, src/Mesh/d4est_mesh.c:2200-2260,
// walks the same arrays), then rhs = M f (or V^T W J f) - A(0) on the bound plan with the boundary data currently set on it.
void d4est_laplacian_build_rhs_with_strong_bc(p4est_t* p4est, d4est_ghost_t*, d4est_ghost_data_t*, d4est_operators_t*, d4est_geometry_t*,
                                              d4est_quadrature_t*, d4est_mesh_data_t*, d4est_elliptic_data_t* prob_vecs,
                                              d4est_laplacian_flux_data_t* flux_fcn_data_for_build_rhs, double* rhs, d4est_xyz_fcn_t problem_rhs_fcn,
                                              d4est_mesh_init_field_option_t init_option, void* ctx, int which_field) {
  d4est_hip_plan_t* plan = bound(p4est, "d4est_laplacian_build_rhs_with_strong_bc");
  if (!prob_vecs || prob_vecs->local_nodes != d4est_hip_plan_local_nodes(plan)) COMPAT_ABORT("d4est_laplacian_build_rhs_with_strong_bc: elliptic data does not match the bound plan");
  if (init_option != INIT_FIELD_ON_LOBATTO && init_option != INIT_FIELD_ON_QUAD) COMPAT_ABORT("d4est_laplacian_build_rhs_with_strong_bc: Not a support init option");   // :46-48
  if (!problem_rhs_fcn) COMPAT_ABORT("d4est_laplacian_build_rhs_with_strong_bc: problem_rhs_fcn == NULL");
  if (flux_fcn_data_for_build_rhs)
    check_flux(p4est, flux_fcn_data_for_build_rhs->flux_type, flux_fcn_data_for_build_rhs->flux_data, flux_fcn_data_for_build_rhs->bc_type,
               "d4est_laplacian_build_rhs_with_strong_bc");
  auto it = g_coord.find(p4est);
  const bool on_quad = init_option == INIT_FIELD_ON_QUAD;
  if (it == g_coord.end() || !(on_quad ? it->second.quad[0] : it->second.lob[0]))
    COMPAT_ABORT("d4est_laplacian_build_rhs_with_strong_bc: no %s node coordinates registered for this p4est (d4est_hip_compat_bind_coordinates)", on_quad ? "quadrature" : "Lobatto");
  const double* const* X = on_quad ? it->second.quad : it->second.lob;
  const int n = on_quad ? d4est_hip_plan_local_nodes_quad(plan) : d4est_hip_plan_local_nodes(plan);
  std::vector<double> f((size_t)n);
  for (int i = 0; i < n; ++i) f[i] = problem_rhs_fcn(X[0][i], X[1][i], X[2][i], ctx);
  d4est_hip_build_rhs_with_strong_bc_host(plan, f.data(), on_quad ? 1 : 0, rhs + (size_t)which_field * prob_vecs->local_nodes);
}

// src/dGMath/d4est_operators.c:1951-1991: reverse the node order of a face array along direction dir (0: the fast index, 1: the slow
// one, 2: both); dim = 1: a line.  Pure index work, done on the host.
void d4est_operators_apply_flip(d4est_operators_t*, double* in, int dim, int deg, int dir, double* out) {
  const int n = deg + 1;
  if (dim == 1) { for (int a = 0; a < n; ++a) out[a] = in[deg - a]; return; }
  if (dim != 2) COMPAT_ABORT("d4est_operators_apply_flip: flip not supported in this dimension atm.");   // :1988
  if (dir < 0 || dir > 2) return;   // (the reference does nothing for another dir)
  for (int b = 0; b < n; ++b)
    for (int a = 0; a < n; ++a)
      out[a + (size_t)n * b] = in[((dir == 0 || dir == 2) ? deg - a : a) + (size_t)n * ((dir == 1 || dir == 2) ? deg - b : b)];
}
// :1993-2087: bring a face array from the (+) side's order into the (-) side's: flip0 / flip1 / transpose from p4est's face transform of
// the pair (lower face, higher face, orientation)
void d4est_operators_reorient_face_data(d4est_operators_t* ops, double* in, int face_dim, int deg, int o, int f_m, int f_p, double* out) {
  const int n = deg + 1;
  if (face_dim == 1) {
    if (o == 1) d4est_operators_apply_flip(ops, in, 1, deg, 0, out);
    else std::memcpy(out, in, sizeof(double) * n);
    return;
  }
  if (face_dim != 2 || o < 0 || o > 3) COMPAT_ABORT("d4est_operators_reorient_face_data: face_dim %d, orientation %d", face_dim, o);
  const int code = d4est_hip_face_reorder_code(f_m, f_p, o);
  for (int b = 0; b < n; ++b)
    for (int a = 0; a < n; ++a) {
      // out = transpose?(flip1?(flip0?(in))): undo in reverse order to find the source entry
      int sa = (code & 4) ? b : a, sb = (code & 4) ? a : b;
      if (code & 2) sb = deg - sb;
      if (code & 1) sa = deg - sa;
      out[a + (size_t)n * b] = in[sa + (size_t)n * sb];
    }
}

// src/Mesh/d4est_mortars.c:550-598 / :510-547: a side's face data onto its mortar space (1 -> 1 p-prolongation, 1 -> 4 hp-prolongation,
// 4 -> 4 face by face) and the mass-weighted way back (the transposes).  (P4EST_DIM) - 1 = 2: the face transfers above.
static void project_faces(bool to_mortar, d4est_operators_t* ops, double* in, int faces_in, int* deg_in, double* out, int faces_out, int* deg_out,
                          const char* who) {
  // to_mortar: in = side, out = mortar; else in = mortar, out = side
  int faces_side = to_mortar ? faces_in : faces_out, faces_mortar = to_mortar ? faces_out : faces_in;
  int* deg_side = to_mortar ? deg_in : deg_out;
  int* deg_mortar = to_mortar ? deg_out : deg_in;
  if (faces_side == 1 && faces_mortar == 1) {
    if (to_mortar) d4est_operators_apply_p_prolong(ops, in, deg_side[0], 2, deg_mortar[0], out);
    else d4est_operators_apply_p_prolong_transpose(ops, in, deg_mortar[0], 2, deg_side[0], out);
  } else if (faces_side == 1 && faces_mortar == 4) {
    if (to_mortar) d4est_operators_apply_hp_prolong(ops, in, deg_side[0], 2, deg_mortar, out);
    else d4est_operators_apply_hp_prolong_transpose(ops, in, deg_mortar, 2, deg_side[0], out);
  } else if (faces_side == 4 && faces_mortar == 4) {
    size_t ss = 0, sm = 0;
    for (int i = 0; i < 4; ++i) {
      if (to_mortar) d4est_operators_apply_p_prolong(ops, in + ss, deg_side[i], 2, deg_mortar[i], out + sm);
      else d4est_operators_apply_p_prolong_transpose(ops, in + sm, deg_mortar[i], 2, deg_side[i], out + ss);
      ss += (size_t)(deg_side[i] + 1) * (deg_side[i] + 1);
      sm += (size_t)(deg_mortar[i] + 1) * (deg_mortar[i] + 1);
    }
  } else COMPAT_ABORT("ERROR: %s", who);
}
void d4est_mortars_project_side_onto_mortar_space(d4est_operators_t* d4est_ops, double* in_side, int faces_side, int* deg_side, double* out_mortar,
                                                  int faces_mortar, int* deg_mortar) {
  project_faces(true, d4est_ops, in_side, faces_side, deg_side, out_mortar, faces_mortar, deg_mortar, "d4est_mortars_project_side_onto_mortar_space");
}
void d4est_mortars_project_mass_mortar_onto_side(d4est_operators_t* dgmath, double* in_mortar, int faces_mortar, int* deg_mortar, double* out_side,
                                                 int faces_side, int* deg_side) {
  project_faces(false, dgmath, in_mortar, faces_mortar, deg_mortar, out_side, faces_side, deg_side, "d4est_mortars_project_mass_mortar_onto_side_space");
}

// ---- additive Schwarz: the reference's metadata (src/Solver/d4est_solver_schwarz_metadata.h:19-90) flattened into the arrays
// d4est_hip_schwarz_create takes -- INTEGRATION.md section 2e as code.  Outputs are caller-allocated: sub_first[num_subdomains + 1],
// sub_elem[num_elements], sub_faces / sub_core_faces[3 num_elements].  Subdomain elements keep the reference's order (sorted by
// (tree, quadid), d4est_solver_schwarz_metadata.c:447-455); `id` must be a LOCAL element id (one rank; on several ranks hand over the
// rank's extended mesh and map ghost ids into it first).
void d4est_hip_compat_flatten_schwarz_metadata(const d4est_solver_schwarz_metadata_t* md, int* sub_first, int* sub_elem, int* sub_faces,
                                               int* sub_core_faces) {
  if (!md || !sub_first || !sub_elem || !sub_faces || !sub_core_faces) COMPAT_ABORT("flatten_schwarz_metadata: NULL argument");
  int k = 0;
  for (int i = 0; i < md->num_subdomains; ++i) {
    const d4est_solver_schwarz_subdomain_metadata_t& sd = md->subdomain_metadata[i];
    sub_first[i] = k;
    for (int j = 0; j < sd.num_elements; ++j, ++k) {
      const d4est_solver_schwarz_element_metadata_t& ed = sd.element_metadata[j];
      if (ed.id < 0) COMPAT_ABORT("flatten_schwarz_metadata: subdomain %d element %d has id %d (a second-layer ghost?)", i, j, ed.id);
      sub_elem[k] = ed.id;
      for (int f = 0; f < 3; ++f) { sub_faces[3 * k + f] = ed.faces[f]; sub_core_faces[3 * k + f] = ed.core_faces[f]; }
    }
  }
  sub_first[md->num_subdomains] = k;
  if (k != md->num_elements) COMPAT_ABORT("flatten_schwarz_metadata: %d subdomain elements counted, metadata says %d", k, md->num_elements);
}

// src/Solver/d4est_solver_schwarz.c:172-285 with the reference's argument list: vecs->u += the Schwarz correction of the residual r, on
// the smoother handle bound to this p4est (d4est_hip_compat_bind_schwarz; the three CG options were given there).  Host vectors in and
// out (one upload of u and r, one download of u); `schwarz` is not read.
void d4est_solver_schwarz_iterate(p4est_t* p4est, d4est_geometry_t*, d4est_quadrature_t*, d4est_mesh_data_t*, d4est_ghost_t*,
                                  d4est_solver_schwarz_t*, d4est_elliptic_data_t* vecs, double* r) {
  auto it = g_schwarz.find(p4est);
  if (it == g_schwarz.end()) COMPAT_ABORT("d4est_solver_schwarz_iterate: no Schwarz smoother bound to p4est %p (d4est_hip_compat_bind_schwarz)", (const void*)p4est);
  if (!vecs || !vecs->u || !r) COMPAT_ABORT("d4est_solver_schwarz_iterate: NULL vector");
  const size_t bytes = sizeof(double) * (size_t)vecs->local_nodes;
  double* d_u = (double*)d4est_hip_malloc(bytes);
  double* d_r = (double*)d4est_hip_malloc(bytes);
  d4est_hip_memcpy_h2d(d_u, vecs->u, bytes);
  d4est_hip_memcpy_h2d(d_r, r, bytes);
  d4est_hip_schwarz_iterate(it->second.sz, d_u, d_r, it->second.iter, it->second.atol, it->second.rtol);
  d4est_hip_device_synchronize();
  d4est_hip_memcpy_d2h(vecs->u, d_u, bytes);
  d4est_hip_free(d_u);
  d4est_hip_free(d_r);
}

d4est_hip_plan_t* d4est_hip_compat_bound_plan(const void* p4est) {
  auto it = g_bound.find(p4est);
  return it == g_bound.end() ? nullptr : it->second;
}
void d4est_hip_compat_release(void) {
  for (auto& kv : g_elem) {
    ElemCtx& c = kv.second;
    d4est_hip_plan_destroy(c.plan);
    d4est_hip_host_free(c.h);
    d4est_hip_free(c.d_in); d4est_hip_free(c.d_out); d4est_hip_free(c.d_geo);
  }
  g_elem.clear();
  for (auto& kv : g_transfer) d4est_hip_transfer_destroy(kv.second);
  g_transfer.clear();
  if (g_tr_h) { d4est_hip_host_free(g_tr_h); d4est_hip_free(g_tr_d); g_tr_h = g_tr_d = nullptr; g_tr_cap = 0; }
}

}  // extern "C"
