// extern "C" entry points of libd4est_hip.so (declared in include/d4est_hip.h):
// tables, device-memory helpers, plan life cycle and the volume applies.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <utility>

#include "d4est_hip_internal.h"
#include "d4est_hip_maps.h"
#include "d4est_hip_tables.h"

using d4est_hip::Bucket;
using d4est_hip::Tables1D;

namespace {

double* upload(const std::vector<double>& v) {
  double* d = nullptr;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(double)));
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
  return d;
}

int* upload_i(const std::vector<int>& v) {
  int* d = nullptr;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(int)));
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
  return d;
}

void check_plan(const d4est_hip_plan_t* plan, const char* fn) {
  if (!plan) D4EST_HIP_ABORT("%s: NULL plan", fn);
}

// any change of the plan's state invalidates a captured cheby_iterate graph (D4EST_HIP_TUNE_GRAPH)
static void drop_graph(d4est_hip_plan_t* plan) {
  if (plan->cheby_graph) { (void)hipGraphExecDestroy(plan->cheby_graph); plan->cheby_graph = nullptr; }
  ++plan->op_generation;   // (every caller of this function changes the plan's state)
  // w J c of the fused operator kernels is derived from J and the coefficient: every setter that can change either comes through here
  // (plan_set_jacobian included), so the cached stream can never outlive its inputs
  plan->lhs_wjc_valid = false;
}

}  // namespace

// Stream mode (d4est_hip_wave.h, with_ld): on when one apply streams more than fits the 256 MB Infinity Cache with room to spare --
// metric 48 B per quadrature node + u and A u 16 B per node > 320 MB (measured crossover: p = 12 at 2048 elements, 288 MB, still gains
// from the cache; p = 13 at 2048 elements, 360 MB, and p = 11 at 4096, 453 MB, gain from streaming) -- or forced by tuning key 12.
static void update_stream_mode(d4est_hip_plan* plan) {
  const int t = plan->tuning[D4EST_HIP_TUNE_STREAM];
  const double bytes = 48.0 * (double)plan->local_nodes_quad + 16.0 * (double)plan->local_nodes;
  // (elements that alias each other's quadrature block re-read the metric: not a stream)
  plan->stream_mode = t < 0 ? ((bytes > 320.0e6 && !plan->quad_aliased) ? 1 : 0) : (t != 0 ? 1 : 0);
}

extern "C" {

const char* d4est_hip_version(void) { return "d4est_hip 0.1 (gfx950)"; }

int d4est_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int d4est_hip_table(int table_id, int deg_a, int deg_b, double* out_host) {
  std::vector<double> t, tmp;
  switch (table_id) {
    case D4EST_HIP_TABLE_LOBATTO_NODES: Tables1D::lobatto(deg_a, t, tmp); break;
    case D4EST_HIP_TABLE_LOBATTO_WEIGHTS: Tables1D::lobatto(deg_a, tmp, t); break;
    case D4EST_HIP_TABLE_GAUSS_NODES: Tables1D::gauss(deg_a, t, tmp); break;
    case D4EST_HIP_TABLE_GAUSS_WEIGHTS: Tables1D::gauss(deg_a, tmp, t); break;
    case D4EST_HIP_TABLE_DIJ: t = Tables1D::dij(deg_a); break;
    case D4EST_HIP_TABLE_MIJ: t = Tables1D::mij(deg_a); break;
    case D4EST_HIP_TABLE_INVMIJ: t = Tables1D::invmij(deg_a); break;
    case D4EST_HIP_TABLE_LOBATTO_TO_GAUSS: t = Tables1D::lobatto_to_gauss(deg_a, deg_b); break;
    case D4EST_HIP_TABLE_P_PROLONG: t = Tables1D::p_prolong(deg_a, deg_b); break;
    case D4EST_HIP_TABLE_HP_PROLONG: t = Tables1D::hp_prolong(deg_a, deg_b); break;
    case D4EST_HIP_TABLE_P_RESTRICT: t = Tables1D::p_restrict(deg_a, deg_b); break;
    case D4EST_HIP_TABLE_HP_RESTRICT: t = Tables1D::hp_restrict(deg_a, deg_b); break;
    default: D4EST_HIP_ABORT("d4est_hip_table: unknown table id %d", table_id);
  }
  if (out_host) std::memcpy(out_host, t.data(), t.size() * sizeof(double));
  return (int)t.size();
}

void* d4est_hip_malloc(size_t bytes) {
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
  return p;
}
void d4est_hip_free(void* p) {
  if (p) HIP_CHECK(hipFree(p));
}
void d4est_hip_memcpy_h2d(void* dst, const void* src, size_t bytes) { HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); }
void d4est_hip_memcpy_d2h(void* dst, const void* src, size_t bytes) { HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); }
void d4est_hip_memset(void* dst, int value, size_t bytes) { HIP_CHECK(hipMemset(dst, value, bytes)); }
void d4est_hip_device_synchronize(void) { HIP_CHECK(hipDeviceSynchronize()); }

d4est_hip_plan_t* d4est_hip_plan_create(int n_elements, const int* deg, const int* deg_quad, const int* nodal_stride,
                                        const int* quad_stride, int quad_type) {
  if (n_elements < 0) D4EST_HIP_ABORT("plan_create: n_elements = %d", n_elements);
  if (n_elements > 0 && (!deg || !deg_quad || !nodal_stride || !quad_stride)) D4EST_HIP_ABORT("plan_create: NULL element array");
  if (quad_type != D4EST_HIP_QUAD_LEGENDRE && quad_type != D4EST_HIP_QUAD_LOBATTO) D4EST_HIP_ABORT("plan_create: unknown quadrature type %d", quad_type);
  d4est_hip_plan_t* plan = new d4est_hip_plan_t();
  // D4EST_HIP_FACE_DIRECT = 0 / 1: the default of tuning key D4EST_HIP_TUNE_FACE_DIRECT for plans created from here on (read per
  // plan: the test suite runs the smoother, forest and Schwarz modules once per face path)
  if (const char* e_ = std::getenv("D4EST_HIP_FACE_DIRECT")) plan->tuning[D4EST_HIP_TUNE_FACE_DIRECT] = std::atoi(e_);
  plan->n_elements = n_elements;
  plan->quad_type = quad_type;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) plan->n_cus = prop.multiProcessorCount;
  }
  plan->deg.assign(deg, deg + n_elements);
  plan->deg_quad.assign(deg_quad, deg_quad + n_elements);
  plan->nodal_stride.assign(nodal_stride, nodal_stride + n_elements);
  plan->quad_stride.assign(quad_stride, quad_stride + n_elements);

  // bucket by (deg, deg_quad); keep element order inside a bucket (Morton order of the caller)
  std::map<std::pair<int, int>, std::vector<int>> groups;
  long long ln = 0, lq = 0, sum_q3 = 0;
  for (int e = 0; e < n_elements; ++e) {
    const int p = deg[e], pq = deg_quad[e];
    if (p < 1 || p > Tables1D::kMaxDeg || pq < 1 || pq > Tables1D::kMaxDeg)
      D4EST_HIP_ABORT("plan_create: element %d has deg %d / deg_quad %d outside [1,%d]", e, p, pq, Tables1D::kMaxDeg);
    groups[{p, pq}].push_back(e);
    const long long n3 = (long long)(p + 1) * (p + 1) * (p + 1), q3 = (long long)(pq + 1) * (pq + 1) * (pq + 1);
    ln = std::max(ln, nodal_stride[e] + n3);
    lq = std::max(lq, quad_stride[e] + q3);
    sum_q3 += q3;
  }
  if (ln > 0x7fffffffLL || lq > 0x7fffffffLL) D4EST_HIP_ABORT("plan_create: local_nodes exceeds 32-bit int (reference strides are int)");
  plan->local_nodes = (int)ln;
  plan->local_nodes_quad = (int)lq;
  plan->quad_aliased = sum_q3 > lq;   // elements share quadrature blocks (the copies of a Schwarz subdomain plan)
  update_stream_mode(plan);

  std::vector<int> ids;
  ids.reserve(n_elements);
  for (auto& kv : groups) {
    Bucket bk;
    bk.deg = kv.first.first;
    bk.deg_quad = kv.first.second;
    bk.N = bk.deg + 1;
    bk.NQ = bk.deg_quad + 1;
    bk.n_elem = (int)kv.second.size();
    bk.elem_offset = (int)ids.size();
    ids.insert(ids.end(), kv.second.begin(), kv.second.end());
    std::vector<double> B = Tables1D::quad_interp(quad_type, bk.deg, bk.deg_quad);
    std::vector<double> D = Tables1D::dij(bk.deg);
    std::vector<double> G = Tables1D::matmul(B, D, bk.NQ, bk.N, bk.N);
    bk.d_B = upload(B);
    bk.d_G = upload(G);
    bk.d_D = upload(D);
    bk.d_BT = upload(Tables1D::transpose(B, bk.NQ, bk.N));
    bk.d_GT = upload(Tables1D::transpose(G, bk.NQ, bk.N));
    bk.d_DT = upload(Tables1D::transpose(D, bk.N, bk.N));
    bk.d_w = upload(Tables1D::quad_weights(quad_type, bk.deg_quad));
    {   // even-odd tables: any parity of N, NQ (Tables1D::eo_table)
      bk.d_EBf = upload(Tables1D::eo_table(B, bk.NQ, bk.N, false));
      bk.d_EGf = upload(Tables1D::eo_table(G, bk.NQ, bk.N, true));
      bk.d_EBb = upload(Tables1D::eo_table(Tables1D::transpose(B, bk.NQ, bk.N), bk.N, bk.NQ, false));
      bk.d_EGb = upload(Tables1D::eo_table(Tables1D::transpose(G, bk.NQ, bk.N), bk.N, bk.NQ, true));
      if (bk.N == bk.NQ) {
        std::vector<double> Dq = Tables1D::quad_diff(quad_type, bk.deg_quad);
        bk.d_EDq = upload(Tables1D::eo_table(Dq, bk.N, bk.N, true));
        bk.d_EDqT = upload(Tables1D::eo_table(Tables1D::transpose(Dq, bk.N, bk.N), bk.N, bk.N, true));
      }
    }
    std::vector<double> M = Tables1D::mij(bk.deg), Minv = Tables1D::invmij(bk.deg);
    bk.d_M = upload(M);
    bk.d_MT = upload(Tables1D::transpose(M, bk.N, bk.N));
    bk.d_Minv = upload(Minv);
    bk.d_MinvT = upload(Tables1D::transpose(Minv, bk.N, bk.N));
    if (bk.N == bk.NQ) {
      std::vector<double> Binv = Tables1D::lobatto_to_gauss(bk.deg, bk.deg);
      if (!Tables1D::invert(Binv, bk.N)) D4EST_HIP_ABORT("plan_create: singular lobatto_to_gauss interpolation at deg %d", bk.deg);
      bk.d_Binv = upload(Binv);
      bk.d_BinvT = upload(Tables1D::transpose(Binv, bk.N, bk.N));
      bk.d_wGL = upload(Tables1D::quad_weights(d4est_hip::QUAD_LEGENDRE, bk.deg));
    }
    plan->buckets.push_back(bk);
  }
  std::vector<int> ns_list(ids.size()), qs_list(ids.size());
  for (size_t i = 0; i < ids.size(); ++i) {
    ns_list[i] = nodal_stride[ids[i]];
    qs_list[i] = quad_stride[ids[i]];
  }
  for (Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    const int* nl = ns_list.data() + bk.elem_offset;
    const int* ql = qs_list.data() + bk.elem_offset;
    const int dn = bk.n_elem > 1 ? nl[1] - nl[0] : bk.N * bk.N * bk.N, dq = bk.n_elem > 1 ? ql[1] - ql[0] : bk.NQ * bk.NQ * bk.NQ;
    bool affine = dn >= 0 && dq >= 0;
    for (int i = 1; i < bk.n_elem && affine; ++i) affine = (nl[i] - nl[i - 1] == dn) && (ql[i] - ql[i - 1] == dq);
    if (affine) {
      bk.ns0 = nl[0]; bk.ns_stride = dn; bk.qs0 = ql[0]; bk.qs_stride = dq;
    }
  }
  plan->elem_ids = ids;
  plan->d_elem_ids = upload_i(ids);
  plan->d_ns_list = upload_i(ns_list);
  plan->d_qs_list = upload_i(qs_list);
  return plan;
}

extern "C" void d4est_hip_rccl_exchange_detach_plan(d4est_hip_plan_t* plan);   // d4est_hip_comm.hip

void d4est_hip_plan_destroy(d4est_hip_plan_t* plan) {
  if (!plan) return;
  d4est_hip_rccl_exchange_detach_plan(plan);   // an RCCL exchange object that outlives the plan must not dereference it
  for (Bucket& bk : plan->buckets) {
    (void)hipFree(bk.d_B);
    (void)hipFree(bk.d_G);
    (void)hipFree(bk.d_D);
    (void)hipFree(bk.d_w);
    (void)hipFree(bk.d_BT);
    (void)hipFree(bk.d_GT);
    (void)hipFree(bk.d_DT);
    (void)hipFree(bk.d_EBf);
    (void)hipFree(bk.d_EGf);
    (void)hipFree(bk.d_EDq);
    (void)hipFree(bk.d_EDqT);
    (void)hipFree(bk.d_EBb);
    (void)hipFree(bk.d_EGb);
    (void)hipFree(bk.d_M);
    (void)hipFree(bk.d_MT);
    (void)hipFree(bk.d_Minv);
    (void)hipFree(bk.d_MinvT);
    (void)hipFree(bk.d_Binv);
    (void)hipFree(bk.d_BinvT);
    (void)hipFree(bk.d_wGL);
  }
  (void)hipFree(plan->d_elem_ids);
  (void)hipFree(plan->d_ns_list);
  (void)hipFree(plan->d_qs_list);
  (void)hipFree(plan->d_J);
  (void)hipFree(plan->d_face_stride);
  (void)hipFree(plan->d_metric);
  (void)hipFree(plan->d_metric_affine);
  (void)hipFree(plan->d_nonaffine);
  (void)hipFree(plan->d_scratch);
  d4est_hip::faces_destroy(plan);
  (void)hipFree(plan->d_work_p); (void)hipFree(plan->d_work_d); (void)hipFree(plan->d_work_r); (void)hipFree(plan->d_work_m); (void)hipFree(plan->d_lhs_c); (void)hipFree(plan->d_lhs_wjc);
  (void)hipFree(plan->d_lhs_block_off);
  d4est_hip::lhs_chain_destroy(plan);
  (void)hipFree(plan->d_reduce); (void)hipFree(plan->d_ghost_trace);
  if (plan->h_stage) (void)hipHostFree(plan->h_stage);
  for (int i = 0; i < 4; ++i) (void)hipFree(plan->d_host[i]);
  if (plan->cheby_graph) (void)hipGraphExecDestroy(plan->cheby_graph);
  if (plan->side_stream) { (void)hipStreamDestroy(plan->side_stream); (void)hipEventDestroy(plan->ev_fork); (void)hipEventDestroy(plan->ev_join); }
  delete plan;
}

void d4est_hip_plan_set_stream(d4est_hip_plan_t* plan, void* hip_stream) {
  check_plan(plan, "plan_set_stream");
  drop_graph(plan);
  plan->stream = reinterpret_cast<hipStream_t>(hip_stream);
}

void d4est_hip_plan_set_tuning(d4est_hip_plan_t* plan, int key, int value) {
  check_plan(plan, "plan_set_tuning");
  drop_graph(plan);
  if (key < 0 || key >= D4EST_HIP_TUNE_COUNT) D4EST_HIP_ABORT("plan_set_tuning: unknown key %d", key);
  plan->tuning[key] = value;
  if (key == D4EST_HIP_TUNE_STREAM) update_stream_mode(plan);
}

const char* d4est_hip_plan_last_kernel(const d4est_hip_plan_t* plan) { check_plan(plan, "plan_last_kernel"); return plan->last_kernel; }
const char* d4est_hip_plan_face_path(const d4est_hip_plan_t* plan) {
  check_plan(plan, "plan_face_path");
  if (!plan->has_faces) D4EST_HIP_ABORT("plan_face_path: call plan_set_faces first");
  if (d4est_hip::hybrid_active(plan)) return d4est_hip::hybrid_path(plan);
  return d4est_hip::direct_active(plan) ? (d4est_hip::direct_fused_ok(plan) ? "direct+volume" : "direct") : "two-phase";
}
int d4est_hip_plan_local_nodes(const d4est_hip_plan_t* plan) { check_plan(plan, "plan_local_nodes"); return plan->local_nodes; }
int d4est_hip_plan_local_nodes_quad(const d4est_hip_plan_t* plan) { check_plan(plan, "plan_local_nodes_quad"); return plan->local_nodes_quad; }
int d4est_hip_plan_stream_mode(const d4est_hip_plan_t* plan) { check_plan(plan, "plan_stream_mode"); return plan->stream_mode; }
int d4est_hip_plan_n_elements(const d4est_hip_plan_t* plan) { check_plan(plan, "plan_n_elements"); return plan->n_elements; }

void d4est_hip_plan_set_geometry(d4est_hip_plan_t* plan, const double* J_quad, const double* rst_xyz_quad, int on_device) {
  check_plan(plan, "plan_set_geometry");
  drop_graph(plan);
  if (!J_quad || !rst_xyz_quad) D4EST_HIP_ABORT("plan_set_geometry: NULL geometry array");
  const size_t nq = (size_t)plan->local_nodes_quad;
  if (!plan->d_J) HIP_CHECK(hipMalloc(&plan->d_J, std::max<size_t>(nq, 1) * sizeof(double)));
  if (!plan->d_metric) HIP_CHECK(hipMalloc(&plan->d_metric, std::max<size_t>(6 * nq, 1) * sizeof(double)));
  const double* d_rst = rst_xyz_quad;
  double* tmp_rst = nullptr;
  if (on_device) {
    HIP_CHECK(hipMemcpyAsync(plan->d_J, J_quad, nq * sizeof(double), hipMemcpyDeviceToDevice, plan->stream));
  } else {
    HIP_CHECK(hipMemcpy(plan->d_J, J_quad, nq * sizeof(double), hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&tmp_rst, std::max<size_t>(9 * nq, 1) * sizeof(double)));
    HIP_CHECK(hipMemcpy(tmp_rst, rst_xyz_quad, 9 * nq * sizeof(double), hipMemcpyHostToDevice));
    d_rst = tmp_rst;
  }
  d4est_hip::launch_metric_precombine(plan, plan->d_J, d_rst);
  if (tmp_rst) {
    HIP_CHECK(hipStreamSynchronize(plan->stream));
    HIP_CHECK(hipFree(tmp_rst));
  }
  plan->has_geometry = true;
  plan->lhs_wjc_valid = false;
}

void d4est_hip_plan_set_geometry_numerical(d4est_hip_plan_t* plan, const double* xyz_lobatto, int on_device) {
  check_plan(plan, "plan_set_geometry_numerical");
  drop_graph(plan);
  if (!xyz_lobatto) D4EST_HIP_ABORT("plan_set_geometry_numerical: NULL coordinate array");
  const size_t ln = (size_t)plan->local_nodes, nq = (size_t)plan->local_nodes_quad;
  if (!plan->d_J) HIP_CHECK(hipMalloc(&plan->d_J, std::max<size_t>(nq, 1) * sizeof(double)));
  if (!plan->d_metric) HIP_CHECK(hipMalloc(&plan->d_metric, std::max<size_t>(6 * nq, 1) * sizeof(double)));
  const double* d_xyz = xyz_lobatto;
  double* tmp = nullptr;
  if (!on_device) {
    HIP_CHECK(hipMalloc(&tmp, std::max<size_t>(3 * ln, 1) * sizeof(double)));
    HIP_CHECK(hipMemcpy(tmp, xyz_lobatto, 3 * ln * sizeof(double), hipMemcpyHostToDevice));
    d_xyz = tmp;
  }
  d4est_hip::launch_numerical_geometry(plan, d_xyz);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  if (tmp) HIP_CHECK(hipFree(tmp));
  plan->has_geometry = true;
  plan->lhs_wjc_valid = false;
}

static int* upload_elem_dq(d4est_hip_plan_t* plan, const int* elem_dq, double root_len, const double* extents, const char* who) {
  if (plan->n_elements > 0 && !elem_dq) D4EST_HIP_ABORT("%s: elem_dq is NULL", who);
  if (!extents || !(root_len > 0.) || !(extents[1] > extents[0]) || !(extents[3] > extents[2]) || !(extents[5] > extents[4]))
    D4EST_HIP_ABORT("%s: bad brick extents / root length", who);
  int* d = nullptr;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>((size_t)plan->n_elements, 1) * sizeof(int)));
  if (plan->n_elements > 0) HIP_CHECK(hipMemcpy(d, elem_dq, (size_t)plan->n_elements * sizeof(int), hipMemcpyHostToDevice));
  return d;
}

void d4est_hip_plan_set_geometry_brick(d4est_hip_plan_t* plan, const int* elem_dq, double root_len, const double* extents) {
  check_plan(plan, "plan_set_geometry_brick");
  drop_graph(plan);
  int* d_dq = upload_elem_dq(plan, elem_dq, root_len, extents, "plan_set_geometry_brick");
  const size_t nq = (size_t)plan->local_nodes_quad;
  if (!plan->d_J) HIP_CHECK(hipMalloc(&plan->d_J, std::max<size_t>(nq, 1) * sizeof(double)));
  if (!plan->d_metric) HIP_CHECK(hipMalloc(&plan->d_metric, std::max<size_t>(6 * nq, 1) * sizeof(double)));
  d4est_hip::launch_brick_geometry(plan, d_dq, root_len, extents);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  HIP_CHECK(hipFree(d_dq));
  plan->has_geometry = true;
  plan->lhs_wjc_valid = false;
}

void d4est_hip_plan_set_mortar_geometry_brick(d4est_hip_plan_t* plan, const int* elem_dq, double root_len, const double* extents) {
  check_plan(plan, "plan_set_mortar_geometry_brick");
  drop_graph(plan);
  if (!plan->has_faces) D4EST_HIP_ABORT("plan_set_mortar_geometry_brick: call plan_set_faces first");
  int* d_dq = upload_elem_dq(plan, elem_dq, root_len, extents, "plan_set_mortar_geometry_brick");
  d4est_hip::faces_set_geometry_brick(plan, d_dq, root_len, extents);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  HIP_CHECK(hipFree(d_dq));
}

static d4est_hip::TreeMapParams analytic_params(int geom_type, const double* params, const char* who) {
  if (geom_type != D4EST_HIP_GEOM_CUBED_SPHERE_7TREE) D4EST_HIP_ABORT("%s: unknown geometry type %d", who, geom_type);
  if (!params || !(params[0] > 0.) || !(params[1] > params[0])) D4EST_HIP_ABORT("%s: cubed_sphere_7tree needs params = {R0, R1 > R0, compactify_inner_shell}", who);
  d4est_hip::TreeMapParams P;
  P.type = geom_type;
  P.R0 = params[0];
  P.R1 = params[1];
  P.compactify = params[2] != 0.;
  P.Clength = P.R0 / std::sqrt(3.0);
  return P;
}

static std::vector<d4est_hip::CellDesc> cells_from(int n, const int* tree, const int* q, const int* dq, int max_tree, const char* who) {
  std::vector<d4est_hip::CellDesc> c((size_t)std::max(n, 0));
  for (int e = 0; e < n; ++e) {
    if (tree[e] < 0 || tree[e] > max_tree || dq[e] <= 0) D4EST_HIP_ABORT("%s: element %d has tree %d / dq %d", who, e, tree[e], dq[e]);
    c[e].tree = tree[e];
    c[e].q[0] = q[3 * e]; c[e].q[1] = q[3 * e + 1]; c[e].q[2] = q[3 * e + 2];
    c[e].dq = dq[e];
    c[e].face = 0;
  }
  return c;
}

void d4est_hip_plan_set_geometry_analytic(d4est_hip_plan_t* plan, int geom_type, const double* params, const int* elem_tree,
                                          const int* elem_q, const int* elem_dq, double root_len) {
  check_plan(plan, "plan_set_geometry_analytic");
  drop_graph(plan);
  const d4est_hip::TreeMapParams P = analytic_params(geom_type, params, "plan_set_geometry_analytic");
  if (plan->n_elements > 0 && (!elem_tree || !elem_q || !elem_dq)) D4EST_HIP_ABORT("plan_set_geometry_analytic: NULL element array");
  if (!(root_len > 0.)) D4EST_HIP_ABORT("plan_set_geometry_analytic: root_len");
  std::vector<d4est_hip::CellDesc> cells = cells_from(plan->n_elements, elem_tree, elem_q, elem_dq, 6, "plan_set_geometry_analytic");
  d4est_hip::CellDesc* d_cells = nullptr;
  HIP_CHECK(hipMalloc(&d_cells, std::max<size_t>(cells.size(), 1) * sizeof(d4est_hip::CellDesc)));
  if (!cells.empty()) HIP_CHECK(hipMemcpy(d_cells, cells.data(), cells.size() * sizeof(d4est_hip::CellDesc), hipMemcpyHostToDevice));
  const size_t nq = (size_t)plan->local_nodes_quad;
  if (!plan->d_J) HIP_CHECK(hipMalloc(&plan->d_J, std::max<size_t>(nq, 1) * sizeof(double)));
  if (!plan->d_metric) HIP_CHECK(hipMalloc(&plan->d_metric, std::max<size_t>(6 * nq, 1) * sizeof(double)));
  d4est_hip::launch_analytic_geometry(plan, P, d_cells, root_len);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  HIP_CHECK(hipFree(d_cells));
  plan->has_geometry = true;
  plan->lhs_wjc_valid = false;
}

void d4est_hip_plan_set_mortar_geometry_analytic(d4est_hip_plan_t* plan, int geom_type, const double* params, const int* elem_tree,
                                                 const int* elem_q, const int* elem_dq, const int* ghost_tree, const int* ghost_q,
                                                 const int* ghost_dq, double root_len) {
  check_plan(plan, "plan_set_mortar_geometry_analytic");
  drop_graph(plan);
  if (!plan->has_faces) D4EST_HIP_ABORT("plan_set_mortar_geometry_analytic: call plan_set_faces first");
  const d4est_hip::TreeMapParams P = analytic_params(geom_type, params, "plan_set_mortar_geometry_analytic");
  if (plan->n_elements > 0 && (!elem_tree || !elem_q || !elem_dq)) D4EST_HIP_ABORT("plan_set_mortar_geometry_analytic: NULL element array");
  if (plan->n_ghost > 0 && (!ghost_tree || !ghost_q || !ghost_dq)) D4EST_HIP_ABORT("plan_set_mortar_geometry_analytic: NULL ghost array");
  if (!(root_len > 0.)) D4EST_HIP_ABORT("plan_set_mortar_geometry_analytic: root_len");
  std::vector<d4est_hip::CellDesc> cells = cells_from(plan->n_elements, elem_tree, elem_q, elem_dq, 6, "plan_set_mortar_geometry_analytic");
  std::vector<d4est_hip::CellDesc> gcells = cells_from(plan->n_ghost, ghost_tree, ghost_q, ghost_dq, 6, "plan_set_mortar_geometry_analytic");
  d4est_hip::faces_set_geometry_analytic(plan, P, cells, gcells, root_len);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
}

void d4est_hip_apply_stiffness_matrix(d4est_hip_plan_t* plan, const double* u_dev, double* Au_dev) {
  check_plan(plan, "apply_stiffness_matrix");
  d4est_hip::launch_stiffness(plan, u_dev, Au_dev);
}

void d4est_hip_apply_mass_matrix(d4est_hip_plan_t* plan, const double* u_dev, double* Mu_dev) {
  check_plan(plan, "apply_mass_matrix");
  d4est_hip::launch_mass_like(plan, 0, u_dev, Mu_dev);
}

void d4est_hip_apply_galerkin_integral(d4est_hip_plan_t* plan, const double* f_quad_dev, double* out_dev) {
  check_plan(plan, "apply_galerkin_integral");
  d4est_hip::launch_mass_like(plan, 1, f_quad_dev, out_dev);
}

void d4est_hip_interpolate(d4est_hip_plan_t* plan, const double* u_dev, double* u_quad_dev) {
  check_plan(plan, "interpolate");
  d4est_hip::launch_mass_like(plan, 2, u_dev, u_quad_dev);
}

void d4est_hip_apply_weighted_mass_matrix(d4est_hip_plan_t* plan, const double* u_dev, const double* coeff_quad_dev, double* out_dev) {
  check_plan(plan, "apply_weighted_mass_matrix");
  if (!coeff_quad_dev) D4EST_HIP_ABORT("apply_weighted_mass_matrix: coeff_quad is NULL");
  d4est_hip::launch_mass_like(plan, 3, u_dev, out_dev, coeff_quad_dev, 0);
}

void d4est_hip_plan_set_lhs_coefficient(d4est_hip_plan_t* plan, const double* coeff_quad_dev) {
  check_plan(plan, "plan_set_lhs_coefficient");
  drop_graph(plan);
  plan->d_lhs_coeff = coeff_quad_dev;
  plan->lhs_wjc_valid = false;
  if (coeff_quad_dev) {   // one form of the zeroth-order term at a time: the coefficient replaces element blocks / a Galerkin chain
    plan->d_lhs_blocks = nullptr;
    d4est_hip::lhs_chain_destroy(plan);
  }
  if (coeff_quad_dev) {   // the values are CAPTURED here (see d4est_hip.h): a copy for the separate mass kernel, w J c for the fused operator kernels
    const size_t nq = std::max<size_t>((size_t)plan->local_nodes_quad, 1);
    if (!plan->d_lhs_c) HIP_CHECK(hipMalloc(&plan->d_lhs_c, nq * sizeof(double)));
    HIP_CHECK(hipMemcpyAsync(plan->d_lhs_c, coeff_quad_dev, (size_t)plan->local_nodes_quad * sizeof(double), hipMemcpyDeviceToDevice, plan->stream));
  }
}

void d4est_hip_apply_inverse_mass_matrix(d4est_hip_plan_t* plan, const double* in_dev, double* out_dev) {
  check_plan(plan, "apply_inverse_mass_matrix");
  d4est_hip::launch_mass_like(plan, 4, in_dev, out_dev, nullptr, 1);
}

void d4est_hip_apply_mij(d4est_hip_plan_t* plan, const double* in_dev, double* out_dev) {
  check_plan(plan, "apply_mij");
  d4est_hip::launch_mass_like(plan, 2, in_dev, out_dev, nullptr, 2);
}

void d4est_hip_apply_invmij(d4est_hip_plan_t* plan, const double* in_dev, double* out_dev) {
  check_plan(plan, "apply_invmij");
  d4est_hip::launch_mass_like(plan, 2, in_dev, out_dev, nullptr, 3);
}

int d4est_hip_plan_face_nodes(const d4est_hip_plan_t* plan) {
  check_plan(plan, "plan_face_nodes");
  long long n = 0;
  for (int e = 0; e < plan->n_elements; ++e) n += (long long)(plan->deg[e] + 1) * (plan->deg[e] + 1);
  return (int)n;
}

void d4est_hip_apply_slicer(d4est_hip_plan_t* plan, const double* in_dev, int face, double* out_face_dev) {
  check_plan(plan, "apply_slicer");
  d4est_hip::launch_slicer_lift(plan, in_dev, out_face_dev, face, 0);
}

void d4est_hip_apply_lift(d4est_hip_plan_t* plan, const double* in_face_dev, int face, double* out_dev) {
  check_plan(plan, "apply_lift");
  d4est_hip::launch_slicer_lift(plan, in_face_dev, out_dev, face, 1);
}

void d4est_hip_apply_dij(d4est_hip_plan_t* plan, const double* in_dev, int dir, double* out_dev) {
  check_plan(plan, "apply_dij");
  d4est_hip::launch_dij(plan, in_dev, out_dev, dir, 0);
}

void d4est_hip_apply_dij_transpose(d4est_hip_plan_t* plan, const double* in_dev, int dir, double* out_dev) {
  check_plan(plan, "apply_dij_transpose");
  d4est_hip::launch_dij(plan, in_dev, out_dev, dir, 1);
}

void d4est_hip_compute_dudr(d4est_hip_plan_t* plan, const double* u_dev, double* dudr0_dev, double* dudr1_dev, double* dudr2_dev) {
  check_plan(plan, "compute_dudr");
  d4est_hip::launch_dudr(plan, u_dev, dudr0_dev, dudr1_dev, dudr2_dev);
}

void d4est_hip_plan_set_faces(d4est_hip_plan_t* plan, const int* side_nbr, const int* side_nbr_face, const int* side_reorder,
                              const int* side_mortar_stride, const int* side_bndry_stride, int total_mortar_nodes,
                              int total_bndry_nodes, int n_ghost, const int* ghost_deg, const int* ghost_deg_quad) {
  check_plan(plan, "plan_set_faces");
  drop_graph(plan);
  if (plan->has_faces) D4EST_HIP_ABORT("plan_set_faces: faces already set (build a new plan per mesh)");
  const size_t ns = 6 * (size_t)plan->n_elements;
  if (ns > 0 && (!side_nbr || !side_nbr_face || !side_reorder || !side_mortar_stride || !side_bndry_stride)) D4EST_HIP_ABORT("plan_set_faces: NULL side array");
  if (n_ghost < 0 || (n_ghost > 0 && (!ghost_deg || !ghost_deg_quad))) D4EST_HIP_ABORT("plan_set_faces: bad ghost arrays");
  plan->side_nbr.assign(side_nbr, side_nbr + ns);
  plan->side_nbr_face.assign(side_nbr_face, side_nbr_face + ns);
  plan->side_reorder.assign(side_reorder, side_reorder + ns);
  plan->side_mortar_stride.assign(side_mortar_stride, side_mortar_stride + ns);
  plan->side_bndry_stride.assign(side_bndry_stride, side_bndry_stride + ns);
  plan->total_mortar_nodes = total_mortar_nodes;
  plan->total_bndry_nodes = total_bndry_nodes;
  plan->n_ghost = n_ghost;
  plan->ghost_deg.assign(ghost_deg, ghost_deg + n_ghost);
  plan->ghost_deg_quad.assign(ghost_deg_quad, ghost_deg_quad + n_ghost);
  for (size_t s = 0; s < ns; ++s) {
    if (side_nbr_face[s] < 0 || side_nbr_face[s] > 5 || side_reorder[s] < 0 || side_reorder[s] > 7) D4EST_HIP_ABORT("plan_set_faces: side %zu has face %d / reorder %d", s, side_nbr_face[s], side_reorder[s]);
  }
  d4est_hip::faces_setup(plan);
}

void d4est_hip_plan_set_hanging(d4est_hip_plan_t* plan, const int* side_hang, const int* side_sub, const int* side_nbr4,
                                const int* side_orientation) {
  check_plan(plan, "plan_set_hanging");
  drop_graph(plan);
  if (plan->has_faces) D4EST_HIP_ABORT("plan_set_hanging: call before d4est_hip_plan_set_faces");
  const size_t ns = 6 * (size_t)plan->n_elements;
  if (ns > 0 && (!side_hang || !side_sub || !side_nbr4 || !side_orientation)) D4EST_HIP_ABORT("plan_set_hanging: NULL array");
  plan->side_hang.assign(side_hang, side_hang + ns);
  plan->side_sub.assign(side_sub, side_sub + ns);
  plan->side_nbr4.assign(side_nbr4, side_nbr4 + 4 * ns);
  plan->side_orientation.assign(side_orientation, side_orientation + ns);
}

void d4est_hip_plan_set_sipg(d4est_hip_plan_t* plan, double penalty_prefactor, int penalty_fcn) {
  check_plan(plan, "plan_set_sipg");
  drop_graph(plan);
  if (penalty_fcn < 0 || penalty_fcn > 3) D4EST_HIP_ABORT("plan_set_sipg: unknown penalty function %d", penalty_fcn);
  if (plan->has_face_geometry) D4EST_HIP_ABORT("plan_set_sipg: call before plan_set_mortar_geometry");
  plan->sipg_prefactor = penalty_prefactor;
  plan->sipg_penalty_fcn = penalty_fcn;
}

void d4est_hip_plan_set_mortar_geometry(d4est_hip_plan_t* plan, const double* sj, const double* n, const double* drst_dxyz_m,
                                        const double* drst_dxyz_p_porder, const double* hm, const double* hp, int on_device) {
  check_plan(plan, "plan_set_mortar_geometry");
  drop_graph(plan);
  if (!plan->has_faces) D4EST_HIP_ABORT("plan_set_mortar_geometry: call plan_set_faces first");
  d4est_hip::faces_set_geometry(plan, sj, n, drst_dxyz_m, drst_dxyz_p_porder, hm, hp, on_device);
}

void d4est_hip_plan_set_dirichlet_values(d4est_hip_plan_t* plan, const double* g_lobatto, int on_device) {
  check_plan(plan, "plan_set_dirichlet_values");
  drop_graph(plan);
  if (!plan->has_faces) D4EST_HIP_ABORT("plan_set_dirichlet_values: call plan_set_faces first");
  d4est_hip::faces_set_dirichlet(plan, g_lobatto, on_device);
}

void d4est_hip_plan_set_robin_values(d4est_hip_plan_t* plan, const double* coeff_quad, const double* rhs_quad, int on_device) {
  check_plan(plan, "plan_set_robin_values");
  drop_graph(plan);
  if (!plan->has_faces) D4EST_HIP_ABORT("plan_set_robin_values: call d4est_hip_plan_set_faces first");
  d4est_hip::faces_set_robin(plan, coeff_quad, rhs_quad, on_device);
}

long long d4est_hip_plan_trace_size(const d4est_hip_plan_t* plan) { check_plan(plan, "plan_trace_size"); return plan->local_trace_doubles; }
long long d4est_hip_plan_ghost_trace_size(const d4est_hip_plan_t* plan) { check_plan(plan, "plan_ghost_trace_size"); return plan->ghost_trace_doubles; }

void d4est_hip_compute_ghost_traces(d4est_hip_plan_t* plan, const double* u_ghost_dev, double* ghost_trace_dev) {
  check_plan(plan, "compute_ghost_traces");
  if (!plan->has_faces) D4EST_HIP_ABORT("compute_ghost_traces: call plan_set_faces first");
  d4est_hip::launch_traces(plan, u_ghost_dev, ghost_trace_dev, true);
}

void d4est_hip_compute_face_traces(d4est_hip_plan_t* plan, const double* u_dev, double* trace_dev) {
  check_plan(plan, "compute_face_traces");
  if (!plan->has_faces) D4EST_HIP_ABORT("compute_face_traces: call plan_set_faces first");
  d4est_hip::launch_traces(plan, u_dev, trace_dev, false);
}

void d4est_hip_apply_flux(d4est_hip_plan_t* plan, const double* trace_dev, const double* ghost_trace_dev, double* Au_dev) {
  check_plan(plan, "apply_flux");
  d4est_hip::launch_flux(plan, trace_dev, ghost_trace_dev, Au_dev);
}

void d4est_hip_apply_aij(d4est_hip_plan_t* plan, const double* u_dev, const double* ghost_trace_dev, double* Au_dev) {
  check_plan(plan, "apply_aij");
  if (!plan->has_faces) D4EST_HIP_ABORT("apply_aij: call plan_set_faces first");
  if (!ghost_trace_dev && plan->ghost_trace_doubles == 0) {
    d4est_hip::apply_operator(plan, u_dev, Au_dev, nullptr, false);  // the Laplacian only (no plan_set_lhs_coefficient term)
    return;
  }
  if (d4est_hip::direct_active(plan)) {   // the caller's ghost traces, the local ones from u inside the kernel
    const bool whole = d4est_hip::direct_fused_ok(plan);
    if (!whole) d4est_hip::launch_stiffness(plan, u_dev, Au_dev);
    d4est_hip::launch_flux_direct(plan, u_dev, ghost_trace_dev, Au_dev, nullptr, whole ? 1 : 0);
    return;
  }
  d4est_hip::launch_stiffness(plan, u_dev, Au_dev);
  d4est_hip::launch_traces(plan, u_dev, plan->d_trace, false);
  d4est_hip::launch_flux(plan, plan->d_trace, ghost_trace_dev, Au_dev);
}

void d4est_hip_plan_set_comm(d4est_hip_plan_t* plan, d4est_hip_exchange_fn exchange, d4est_hip_allreduce_fn allreduce, void* ctx) {
  check_plan(plan, "plan_set_comm");
  drop_graph(plan);
  plan->exchange_fn = exchange;
  plan->allreduce_fn = allreduce;
  plan->comm_ctx = ctx;
}

void d4est_hip_apply_lhs(d4est_hip_plan_t* plan, const double* u_dev, double* Au_dev) {
  check_plan(plan, "apply_lhs");
  d4est_hip::apply_operator(plan, u_dev, Au_dev);
}

void d4est_hip_cheby_iterate(d4est_hip_plan_t* plan, double* u_dev, const double* rhs_dev, double* Au_dev, double* r_dev, int iter,
                             double lmin, double lmax, int compute_residual_at_end) {
  check_plan(plan, "cheby_iterate");
  d4est_hip::cheby_iterate(plan, u_dev, rhs_dev, Au_dev, r_dev, iter, lmin, lmax, compute_residual_at_end);
}

void d4est_hip_cheby_update(d4est_hip_plan_t* plan, int n, const double* rhs_dev, const double* Au_dev, double alpha, double beta,
                            double* r_dev, double* p_dev, double* u_dev) {
  check_plan(plan, "cheby_update");
  d4est_hip::launch_cheby_update(plan, n, rhs_dev, Au_dev, alpha, beta, r_dev, p_dev, u_dev);
}

double d4est_hip_cg_eigs(d4est_hip_plan_t* plan, double* u_dev, const double* rhs_dev, double* Au_dev, int imax, int use_new,
                         double* history_host) {
  check_plan(plan, "cg_eigs");
  return d4est_hip::cg_eigs(plan, u_dev, rhs_dev, Au_dev, imax, use_new, history_host);
}

void d4est_hip_copy_blocks(d4est_hip_plan_t* plan, int n_blocks, const double* src_dev, const long long* src_off_dev,
                           double* dst_dev, const long long* dst_off_dev, const int* len_dev) {
  check_plan(plan, "copy_blocks");
  d4est_hip::launch_copy_blocks(plan->stream, n_blocks, src_dev, src_off_dev, dst_dev, dst_off_dev, len_dev);
}

long long d4est_hip_plan_trace_offset(const d4est_hip_plan_t* plan, int side) {
  check_plan(plan, "plan_trace_offset");
  if (!plan->has_faces || side < 0 || side >= 6 * plan->n_elements) D4EST_HIP_ABORT("plan_trace_offset: side %d", side);
  return plan->trace_offset[side];
}

long long d4est_hip_plan_ghost_trace_offset(const d4est_hip_plan_t* plan, int side) {
  check_plan(plan, "plan_ghost_trace_offset");
  if (!plan->has_faces || side < 0 || side >= 6 * plan->n_elements) D4EST_HIP_ABORT("plan_ghost_trace_offset: side %d", side);
  return plan->ghost_trace_offset[side];
}

int d4est_hip_plan_trace_block_len(const d4est_hip_plan_t* plan, int side) {
  check_plan(plan, "plan_trace_block_len");
  if (!plan->has_faces || side < 0 || side >= 6 * plan->n_elements) D4EST_HIP_ABORT("plan_trace_block_len: side %d", side);
  const long long next = (side + 1 < 6 * plan->n_elements) ? plan->trace_offset[side + 1] : plan->local_trace_doubles;
  return (int)(next - plan->trace_offset[side]);
}

int d4est_hip_reorient_face_order(int f_m, int f_p, int orientation, int i) {
  if (f_m < 0 || f_m > 5 || f_p < 0 || f_p > 5 || orientation < 0 || orientation > 3 || i < 0 || i > 3) D4EST_HIP_ABORT("reorient_face_order: bad argument");
  return d4est_hip::reorient_face_order(f_m, f_p, orientation, i);
}

int d4est_hip_face_reorder_code(int f_m, int f_p, int orientation) {
  if (f_m < 0 || f_m > 5 || f_p < 0 || f_p > 5 || orientation < 0 || orientation > 3) D4EST_HIP_ABORT("face_reorder_code: bad argument");
  return d4est_hip::face_reorder_code(f_m, f_p, orientation);
}

int d4est_hip_plan_side_blocks(const d4est_hip_plan_t* plan, int side) {
  check_plan(plan, "plan_side_blocks");
  if (!plan->has_faces || side < 0 || side >= 6 * plan->n_elements) D4EST_HIP_ABORT("plan_side_blocks: side %d", side);
  if (plan->side_first_rec.empty()) return 1;
  return plan->side_first_rec[side + 1] - plan->side_first_rec[side];
}

static int sub_record(const d4est_hip_plan_t* plan, int side, int sub, const char* who) {
  if (!plan->has_faces || side < 0 || side >= 6 * plan->n_elements) D4EST_HIP_ABORT("%s: side %d", who, side);
  const int nb = plan->side_first_rec.empty() ? 1 : plan->side_first_rec[side + 1] - plan->side_first_rec[side];
  if (sub < 0 || sub >= nb) D4EST_HIP_ABORT("%s: side %d has %d block(s), asked for %d", who, side, nb, sub);
  return plan->side_first_rec.empty() ? -1 : plan->side_first_rec[side] + sub;
}

long long d4est_hip_plan_trace_offset_sub(const d4est_hip_plan_t* plan, int side, int sub) {
  check_plan(plan, "plan_trace_offset_sub");
  const int r = sub_record(plan, side, sub, "plan_trace_offset_sub");
  return r < 0 ? plan->trace_offset[side] : plan->rec_qoff[r];
}

long long d4est_hip_plan_ghost_trace_offset_sub(const d4est_hip_plan_t* plan, int side, int sub) {
  check_plan(plan, "plan_ghost_trace_offset_sub");
  const int r = sub_record(plan, side, sub, "plan_ghost_trace_offset_sub");
  return r < 0 ? plan->ghost_trace_offset[side] : plan->rec_goff[r];
}

int d4est_hip_plan_trace_block_len_sub(const d4est_hip_plan_t* plan, int side, int sub) {
  check_plan(plan, "plan_trace_block_len_sub");
  const int r = sub_record(plan, side, sub, "plan_trace_block_len_sub");
  return r < 0 ? d4est_hip_plan_trace_block_len(plan, side) : plan->rec_len[r];
}

void d4est_hip_vec_dot(d4est_hip_plan_t* plan, int n, const double* x_dev, const double* y_dev, double* result_dev) {
  check_plan(plan, "vec_dot");
  d4est_hip::launch_dot(plan, n, x_dev, y_dev, result_dev);
}

// ---- host-pointer entries: persistent mirrors, upload once / download once -------------------------------------------
static void ensure_host_mirrors(d4est_hip_plan_t* plan) {
  if (plan->h_stage) return;
  const size_t bytes = std::max<size_t>((size_t)plan->local_nodes, 1) * sizeof(double);
  HIP_CHECK(hipHostMalloc((void**)&plan->h_stage, 3 * bytes, hipHostMallocDefault));
  for (int i = 0; i < 4; ++i) HIP_CHECK(hipMalloc(&plan->d_host[i], bytes));
}
// host vector -> pinned slot -> device mirror, asynchronously on the plan's stream
static void stage_in(d4est_hip_plan_t* plan, int slot, const double* host, double* dev) {
  const size_t bytes = (size_t)plan->local_nodes * sizeof(double);
  double* pin = plan->h_stage + (size_t)slot * plan->local_nodes;
  std::memcpy(pin, host, bytes);
  HIP_CHECK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, plan->stream));
}
static void stage_out_begin(d4est_hip_plan_t* plan, int slot, const double* dev) {
  HIP_CHECK(hipMemcpyAsync(plan->h_stage + (size_t)slot * plan->local_nodes, dev, (size_t)plan->local_nodes * sizeof(double),
                           hipMemcpyDeviceToHost, plan->stream));
}
static void stage_out_end(d4est_hip_plan_t* plan, int slot, double* host) {
  std::memcpy(host, plan->h_stage + (size_t)slot * plan->local_nodes, (size_t)plan->local_nodes * sizeof(double));
}

void d4est_hip_apply_stiffness_matrix_host(d4est_hip_plan_t* plan, const double* u_host, double* Au_host) {
  check_plan(plan, "apply_stiffness_matrix_host");
  if (!u_host || !Au_host) D4EST_HIP_ABORT("apply_stiffness_matrix_host: NULL vector");
  ensure_host_mirrors(plan);
  stage_in(plan, 0, u_host, plan->d_host[0]);
  d4est_hip::launch_stiffness(plan, plan->d_host[0], plan->d_host[2]);
  stage_out_begin(plan, 1, plan->d_host[2]);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  stage_out_end(plan, 1, Au_host);
}

void d4est_hip_apply_aij_host(d4est_hip_plan_t* plan, const double* u_host, double* Au_host) {
  check_plan(plan, "apply_aij_host");
  if (!u_host || !Au_host) D4EST_HIP_ABORT("apply_aij_host: NULL vector");
  ensure_host_mirrors(plan);
  stage_in(plan, 0, u_host, plan->d_host[0]);
  d4est_hip::apply_operator(plan, plan->d_host[0], plan->d_host[2], nullptr, false);
  stage_out_begin(plan, 1, plan->d_host[2]);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  stage_out_end(plan, 1, Au_host);
}

void d4est_hip_apply_lhs_host(d4est_hip_plan_t* plan, const double* u_host, double* Au_host) {
  check_plan(plan, "apply_lhs_host");
  if (!u_host || !Au_host) D4EST_HIP_ABORT("apply_lhs_host: NULL vector");
  ensure_host_mirrors(plan);
  stage_in(plan, 0, u_host, plan->d_host[0]);
  d4est_hip::apply_operator(plan, plan->d_host[0], plan->d_host[2]);
  stage_out_begin(plan, 1, plan->d_host[2]);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  stage_out_end(plan, 1, Au_host);
}

void d4est_hip_build_rhs_with_strong_bc(d4est_hip_plan_t* plan, const double* f_dev, int f_on_quad, double* rhs_dev) {
  check_plan(plan, "build_rhs_with_strong_bc");
  if (!f_dev || !rhs_dev) D4EST_HIP_ABORT("build_rhs_with_strong_bc: NULL vector");
  if (!plan->has_faces) D4EST_HIP_ABORT("build_rhs_with_strong_bc: call plan_set_faces first");
  d4est_hip::ensure_solver_workspace(plan);
  const size_t bytes = std::max<size_t>((size_t)plan->local_nodes, 1) * sizeof(double);
  // A(u = 0) with the plan's boundary data: the pure Laplacian, as the reference calls d4est_laplacian_apply_aij (:44)
  HIP_CHECK(hipMemsetAsync(plan->d_work_d, 0, bytes, plan->stream));
  d4est_hip::apply_operator(plan, plan->d_work_d, plan->d_work_r, nullptr, false);
  d4est_hip::launch_mass_like(plan, f_on_quad ? 1 : 0, f_dev, rhs_dev);
  d4est_hip::launch_residual_inplace_sub(plan, plan->local_nodes, plan->d_work_r, rhs_dev);   // rhs -= A(0)   (d4est_linalg_vec_axpy(-1, ...), :136)
}

void d4est_hip_build_rhs_with_strong_bc_host(d4est_hip_plan_t* plan, const double* f_host, int f_on_quad, double* rhs_host) {
  check_plan(plan, "build_rhs_with_strong_bc_host");
  if (!f_host || !rhs_host) D4EST_HIP_ABORT("build_rhs_with_strong_bc_host: NULL vector");
  ensure_host_mirrors(plan);
  const size_t nf = f_on_quad ? (size_t)plan->local_nodes_quad : (size_t)plan->local_nodes;
  double* d_f = nullptr;   // (once per solve: a transient buffer, the quadrature-node form does not fit the nodal mirrors)
  HIP_CHECK(hipMalloc(&d_f, std::max<size_t>(nf, 1) * sizeof(double)));
  HIP_CHECK(hipMemcpyAsync(d_f, f_host, nf * sizeof(double), hipMemcpyHostToDevice, plan->stream));
  d4est_hip_build_rhs_with_strong_bc(plan, d_f, f_on_quad, plan->d_host[2]);
  stage_out_begin(plan, 1, plan->d_host[2]);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  stage_out_end(plan, 1, rhs_host);
  HIP_CHECK(hipFree(d_f));
}

void d4est_hip_cheby_iterate_host(d4est_hip_plan_t* plan, double* u_host, const double* rhs_host, double* Au_host, double* r_host,
                                  int iter, double lmin, double lmax, int compute_residual_at_end) {
  check_plan(plan, "cheby_iterate_host");
  if (!u_host || !rhs_host || !r_host) D4EST_HIP_ABORT("cheby_iterate_host: NULL vector");
  ensure_host_mirrors(plan);
  stage_in(plan, 0, u_host, plan->d_host[0]);
  stage_in(plan, 1, rhs_host, plan->d_host[1]);
  d4est_hip::cheby_iterate(plan, plan->d_host[0], plan->d_host[1], plan->d_host[2], plan->d_host[3], iter, lmin, lmax, compute_residual_at_end);
  stage_out_begin(plan, 0, plan->d_host[0]);
  stage_out_begin(plan, 1, plan->d_host[3]);
  if (Au_host) stage_out_begin(plan, 2, plan->d_host[2]);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  stage_out_end(plan, 0, u_host);
  stage_out_end(plan, 1, r_host);
  if (Au_host) stage_out_end(plan, 2, Au_host);
}

double d4est_hip_cg_eigs_host(d4est_hip_plan_t* plan, double* u_host, const double* rhs_host, double* Au_host, int imax, int use_new,
                              double* history_host) {
  check_plan(plan, "cg_eigs_host");
  if (!u_host || !rhs_host) D4EST_HIP_ABORT("cg_eigs_host: NULL vector");
  ensure_host_mirrors(plan);
  stage_in(plan, 0, u_host, plan->d_host[0]);
  stage_in(plan, 1, rhs_host, plan->d_host[1]);
  const double bound = d4est_hip::cg_eigs(plan, plan->d_host[0], plan->d_host[1], plan->d_host[2], imax, use_new, history_host);
  stage_out_begin(plan, 0, plan->d_host[0]);
  if (Au_host) stage_out_begin(plan, 2, plan->d_host[2]);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  stage_out_end(plan, 0, u_host);
  if (Au_host) stage_out_end(plan, 2, Au_host);
  return bound;
}

void d4est_hip_plan_set_jacobian(d4est_hip_plan_t* plan, const double* J_quad, int on_device) {
  check_plan(plan, "plan_set_jacobian");
  drop_graph(plan);
  if (!J_quad) D4EST_HIP_ABORT("plan_set_jacobian: NULL array");
  const size_t nq = (size_t)plan->local_nodes_quad;
  if (!plan->d_J) HIP_CHECK(hipMalloc(&plan->d_J, std::max<size_t>(nq, 1) * sizeof(double)));
  HIP_CHECK(hipMemcpyAsync(plan->d_J, J_quad, nq * sizeof(double), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, plan->stream));
  if (!on_device) HIP_CHECK(hipStreamSynchronize(plan->stream));
}

void* d4est_hip_host_alloc(size_t bytes) {
  void* p = nullptr;
  HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
  return p;
}
void d4est_hip_host_free(void* ptr_host) {
  if (ptr_host) HIP_CHECK(hipHostFree(ptr_host));
}
void d4est_hip_memcpy_h2d_async(d4est_hip_plan_t* plan, void* dst_dev, const void* src_host, size_t bytes) {
  check_plan(plan, "memcpy_h2d_async");
  HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, plan->stream));
}
void d4est_hip_memcpy_d2h_async(d4est_hip_plan_t* plan, void* dst_host, const void* src_dev, size_t bytes) {
  check_plan(plan, "memcpy_d2h_async");
  HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, plan->stream));
}
void d4est_hip_plan_synchronize(d4est_hip_plan_t* plan) {
  check_plan(plan, "plan_synchronize");
  HIP_CHECK(hipStreamSynchronize(plan->stream));
}

}  // extern "C"
