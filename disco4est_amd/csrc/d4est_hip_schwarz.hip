// Additive Schwarz smoother, batched over all subdomains of a rank.
//
// Reference: d4est_solver_schwarz_iterate (src/Solver/d4est_solver_schwarz.c:172-285) visits the subdomains one after another;
// per subdomain it restricts the residual (src/Solver/d4est_solver_schwarz_helpers.c:62-122), runs a CG on the restricted field
// (src/Solver/d4est_solver_schwarz_subdomain_solver_cg.c:101-249) whose operator is the Laplacian over the subdomain's elements with
// zero_and_skip masks (src/Solver/d4est_solver_schwarz_laplacian_ext.c:167-358), weights the solution with the hat function
// (src/Solver/d4est_solver_schwarz_operators.c:78-105, :334-397) and adds it to u
// (src/Solver/d4est_solver_schwarz_transfer_ghost_data.c:97-120).
//
// Here every subdomain is solved at the same time.  The elements of all subdomains, one subdomain after another, form a second
// ("subdomain") plan: element v of it is a copy of a mesh element, its geometric factors alias the mesh element's (same quad_stride /
// mortar strides, nothing is duplicated), its neighbours are the copies inside the same subdomain, and a face whose neighbour lies
// outside the subdomain is bound to a ghost side whose trace is identically zero -- exactly the reference's zero_and_skip rule
// (u = du/dr = 0 on the outside element, nothing accumulated into it; src/dGMath/d4est_laplacian_flux.c:486-520, :944-962).  One
// d4est_hip_apply_aij on that plan is therefore A restricted to every subdomain at once, and the fused SIPG kernels are reused as they
// are.  A field over the subdomains is stored at full element size (the reference's "field over subdomain", nodal_size per subdomain);
// the restricted field of the reference is that field with the nodes outside the overlap kept at zero (restrict-transpose is implicit),
// so the CG vector kernels below only touch the restricted nodes.
//
// CG: one workgroup per subdomain and iteration does the three vector passes (d.Ad, the u/r update with r.r, the new direction) with
// fixed-order reductions; alpha, beta and the break test (delta_new < atol^2 + delta_0 rtol^2) stay on the device, a subdomain that
// has met its tolerance is frozen exactly where the reference's loop breaks.
#include <algorithm>
#include <cmath>
#include <vector>

#include "d4est_hip_internal.h"
#include "d4est_hip_tables.h"

namespace {

struct VirtDesc {
  int dst;      // offset of the copy in a field over the subdomains
  int src;      // nodal_stride of the mesh element
  int N;        // deg + 1
  int lo[3];    // restricted node range per direction: lo <= index < hi
  int hi[3];
  int woff[3];  // hat weight of node index i in direction d: weights[woff[d] + i]
};

}  // namespace

struct d4est_hip_schwarz {
  d4est_hip_plan_t* plan = nullptr;  // the subdomain plan (not owned)
  int n_sub = 0, n_virtual = 0, n_mesh = 0, overlap = 0;
  long long nodal_size = 0, restricted_nodal_size = 0;
  int mesh_nodes = 0;
  VirtDesc* d_vd = nullptr;
  int* d_sub_first = nullptr;
  int* d_mesh_first = nullptr;    // per mesh element: first entry of its contribution list (n_mesh + 1)
  int* d_mesh_contrib = nullptr;  // subdomain elements that are copies of the mesh element, ascending (= ascending subdomain)
  int* d_mesh_stride = nullptr;
  int* d_mesh_N = nullptr;
  double* d_weights = nullptr;
  // workspace of iterate (lazy)
  double *d_du = nullptr, *d_r = nullptr, *d_d = nullptr, *d_Ad = nullptr, *d_zero_ghost = nullptr;
  double *d_delta = nullptr, *d_tol = nullptr;
  int *d_active = nullptr, *d_final_iter = nullptr, *d_n_active = nullptr;
};

namespace d4est_hip {

__device__ inline bool in_overlap(const VirtDesc& q, int n, int& i, int& j, int& k) {
  i = n % q.N;
  j = (n / q.N) % q.N;
  k = n / (q.N * q.N);
  return i >= q.lo[0] && i < q.hi[0] && j >= q.lo[1] && j < q.hi[1] && k >= q.lo[2] && k < q.hi[2];
}

// n-th restricted node of a copy (x fastest) -> its offset in a field over the subdomains; count = restricted_count(q)
__device__ inline int restricted_count(const VirtDesc& q) { return (q.hi[0] - q.lo[0]) * (q.hi[1] - q.lo[1]) * (q.hi[2] - q.lo[2]); }
__device__ inline size_t restricted_offset(const VirtDesc& q, int n) {
  const int e0 = q.hi[0] - q.lo[0], e1 = q.hi[1] - q.lo[1];
  const int i = q.lo[0] + n % e0, j = q.lo[1] + (n / e0) % e1, k = q.lo[2] + n / (e0 * e1);
  return (size_t)q.dst + i + q.N * (j + q.N * k);
}

// fixed-order sum over the workgroup (same result on every launch)
__device__ inline double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// out (field over the subdomains) = restrictor applied to the mesh field, zero outside the overlap; one workgroup per copy
__global__ __launch_bounds__(256) void schwarz_restrict_kernel(const VirtDesc* __restrict__ vd, int n_virtual,
                                                               const double* __restrict__ field, double* __restrict__ out) {
  for (int v = blockIdx.x; v < n_virtual; v += gridDim.x) {
    const VirtDesc q = vd[v];
    const int n3 = q.N * q.N * q.N;
    for (int n = threadIdx.x; n < n3; n += blockDim.x) {
      int i, j, k;
      out[q.dst + n] = in_overlap(q, n, i, j, k) ? field[q.src + n] : 0.0;
    }
  }
}

// keep only the restricted nodes of a field over the subdomains
__global__ __launch_bounds__(256) void schwarz_mask_kernel(const VirtDesc* __restrict__ vd, int n_virtual, double* __restrict__ x) {
  for (int v = blockIdx.x; v < n_virtual; v += gridDim.x) {
    const VirtDesc q = vd[v];
    const int n3 = q.N * q.N * q.N;
    for (int n = threadIdx.x; n < n3; n += blockDim.x) {
      int i, j, k;
      if (!in_overlap(q, n, i, j, k)) x[q.dst + n] = 0.0;
    }
  }
}

// CG start of every subdomain: du = 0, r = d = restricted residual (A du = 0 exactly), delta_0, the break tolerance
__global__ __launch_bounds__(256) void schwarz_cg_init_kernel(const VirtDesc* __restrict__ vd, const int* __restrict__ sub_first,
                                                              const double* __restrict__ r_mesh, double* __restrict__ du,
                                                              double* __restrict__ r, double* __restrict__ d,
                                                              double* __restrict__ delta, double* __restrict__ tol, int* __restrict__ active,
                                                              int* __restrict__ final_iter, int* __restrict__ n_active, int iter,
                                                              double atol, double rtol) {
  __shared__ double red[256];
  const int s = blockIdx.x;
  double acc = 0.0;
  for (int v = sub_first[s]; v < sub_first[s + 1]; ++v) {
    const VirtDesc q = vd[v];
    const int n3 = q.N * q.N * q.N;
    for (int n = threadIdx.x; n < n3; n += blockDim.x) {
      int i, j, k;
      const double val = in_overlap(q, n, i, j, k) ? r_mesh[q.src + n] : 0.0;
      du[q.dst + n] = 0.0;
      r[q.dst + n] = val;
      d[q.dst + n] = val;
      acc += val * val;
    }
  }
  const double d0 = block_sum(acc, red);
  if (threadIdx.x == 0) {
    delta[s] = d0;
    tol[s] = atol * atol + d0 * rtol * rtol;
    final_iter[s] = iter;
    const int on = d0 > 0.0;  // a zero residual has the zero solution (the reference would divide 0 / 0 here)
    active[s] = on;
    if (on) atomicAdd(n_active, 1);
  }
}

// iteration `it` of the subdomain CG (subdomain_solver_cg.c:170-226)
__global__ __launch_bounds__(256) void schwarz_cg_kernel(const VirtDesc* __restrict__ vd, const int* __restrict__ sub_first,
                                                         double* __restrict__ du, double* __restrict__ r, double* __restrict__ d,
                                                         const double* __restrict__ Ad, double* __restrict__ delta,
                                                         const double* __restrict__ tol, int* __restrict__ active,
                                                         int* __restrict__ final_iter, int* __restrict__ n_active, int it) {
  __shared__ double red[256];
  const int s = blockIdx.x;
  if (!active[s]) return;
  if (threadIdx.x == 0) atomicMax(n_active + 1, it + 1);   // this sweep did work: the reference's loop was still running
  const int v0 = sub_first[s], v1 = sub_first[s + 1];
  double acc = 0.0;
  for (int v = v0; v < v1; ++v) {
    const VirtDesc q = vd[v];
    const int nr = restricted_count(q);
    for (int n = threadIdx.x; n < nr; n += blockDim.x) {
      const size_t o = restricted_offset(q, n);
      acc += d[o] * Ad[o];
    }
  }
  const double d_dot_Ad = block_sum(acc, red);
  const double delta_old = delta[s];
  const double alpha = delta_old / d_dot_Ad;
  acc = 0.0;
  for (int v = v0; v < v1; ++v) {
    const VirtDesc q = vd[v];
    const int nr = restricted_count(q);
    for (int n = threadIdx.x; n < nr; n += blockDim.x) {
      const size_t o = restricted_offset(q, n);
      du[o] += alpha * d[o];
      const double rn = r[o] - alpha * Ad[o];
      r[o] = rn;
      acc += rn * rn;
    }
  }
  const double delta_new = block_sum(acc, red);
  const double beta = delta_new / delta_old;
  for (int v = v0; v < v1; ++v) {
    const VirtDesc q = vd[v];
    const int nr = restricted_count(q);
    for (int n = threadIdx.x; n < nr; n += blockDim.x) {
      const size_t o = restricted_offset(q, n);
      d[o] = r[o] + beta * d[o];
    }
  }
  if (threadIdx.x == 0) {
    delta[s] = delta_new;
    if (delta_new < tol[s]) {
      active[s] = 0;
      final_iter[s] = it;
      atomicSub(n_active, 1);
    }
  }
}

// u += sum over the subdomains containing the element of (hat weights * du), in ascending subdomain order; one workgroup per mesh element
__global__ __launch_bounds__(256) void schwarz_correction_kernel(const VirtDesc* __restrict__ vd, const int* __restrict__ mesh_first,
                                                                 const int* __restrict__ mesh_contrib,
                                                                 const int* __restrict__ mesh_stride, const int* __restrict__ mesh_N,
                                                                 int n_mesh, const double* __restrict__ weights,
                                                                 const double* __restrict__ du, double* __restrict__ u) {
  for (int e = blockIdx.x; e < n_mesh; e += gridDim.x) {
    const int N = mesh_N[e], n3 = N * N * N, c0 = mesh_first[e], c1 = mesh_first[e + 1];
    for (int n = threadIdx.x; n < n3; n += blockDim.x) {
      double acc = u[mesh_stride[e] + n];
      for (int c = c0; c < c1; ++c) {
        const VirtDesc q = vd[mesh_contrib[c]];
        int i, j, k;
        if (in_overlap(q, n, i, j, k)) {
          // ((w_z w_y) w_x) du, rounded like Kron/d4est_kron.h:170-178, then one rounded add (no contraction)
          const double w = __dmul_rn(__dmul_rn(weights[q.woff[2] + k], weights[q.woff[1] + j]), weights[q.woff[0] + i]);
          acc = __dadd_rn(acc, __dmul_rn(w, du[q.dst + n]));
        }
      }
      u[mesh_stride[e] + n] = acc;
    }
  }
}

static double quintic(double r) { return (15. * r - 10. * r * r * r + 3. * r * r * r * r * r) / 8.; }
static double phi(double r) { return (r < -1 || r > 1) ? (double)((r > 0) - (r < 0)) : quintic(r); }
static double hat(double r, double overlap_size) { return .5 * (phi((r + 1) / overlap_size) - phi((r - 1) / overlap_size)); }

template <class T>
static T* upload(const std::vector<T>& v) {
  T* d = nullptr;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static void ensure_workspace(d4est_hip_schwarz* sz) {
  if (sz->d_du) return;
  const size_t n = std::max<size_t>((size_t)sz->nodal_size, 1) * sizeof(double);
  HIP_CHECK(hipMalloc(&sz->d_du, n));
  HIP_CHECK(hipMalloc(&sz->d_r, n));
  HIP_CHECK(hipMalloc(&sz->d_d, n));
  HIP_CHECK(hipMalloc(&sz->d_Ad, n));
  const size_t ns = std::max<size_t>((size_t)sz->n_sub, 1);
  HIP_CHECK(hipMalloc(&sz->d_delta, ns * sizeof(double)));
  HIP_CHECK(hipMalloc(&sz->d_tol, ns * sizeof(double)));
  HIP_CHECK(hipMalloc(&sz->d_active, ns * sizeof(int)));
  HIP_CHECK(hipMalloc(&sz->d_final_iter, ns * sizeof(int)));
  HIP_CHECK(hipMalloc(&sz->d_n_active, 2 * sizeof(int)));   // [0] subdomains still iterating, [1] sweeps in which any did
}

static void ensure_zero_ghost(d4est_hip_schwarz* sz) {
  if (sz->d_zero_ghost) return;
  const size_t g = std::max<size_t>((size_t)sz->plan->ghost_trace_doubles, 1) * sizeof(double);
  HIP_CHECK(hipMalloc(&sz->d_zero_ghost, g));
  HIP_CHECK(hipMemset(sz->d_zero_ghost, 0, g));
}

}  // namespace d4est_hip

using namespace d4est_hip;

extern "C" {

d4est_hip_schwarz_t* d4est_hip_schwarz_create(d4est_hip_plan_t* subdomain_plan, int n_subdomains, const int* sub_first,
                                              const int* sub_elem, const int* sub_faces, const int* sub_core_faces,
                                              int num_nodes_overlap, int n_mesh_elements, const int* mesh_deg,
                                              const int* mesh_nodal_stride) {
  if (!subdomain_plan) D4EST_HIP_ABORT("schwarz_create: NULL subdomain plan");
  if (n_subdomains < 0 || n_mesh_elements < 0 || (n_subdomains > 0 && (!sub_first || !sub_elem || !sub_faces || !sub_core_faces)) ||
      (n_mesh_elements > 0 && (!mesh_deg || !mesh_nodal_stride)))
    D4EST_HIP_ABORT("schwarz_create: NULL / negative argument");
  if (num_nodes_overlap <= 0) D4EST_HIP_ABORT("schwarz_create: num_nodes_overlap <= 0");  // d4est_solver_schwarz_metadata.c:166-168
  if (num_nodes_overlap == 1) D4EST_HIP_ABORT("schwarz_create: num_nodes_overlap = 1 gives a zero-width weight ramp (0/0 weights)");
  const int nv = n_subdomains > 0 ? sub_first[n_subdomains] : 0;
  if (nv != subdomain_plan->n_elements)
    D4EST_HIP_ABORT("schwarz_create: the subdomain plan has %d elements, the metadata lists %d", subdomain_plan->n_elements, nv);
  d4est_hip_schwarz* sz = new d4est_hip_schwarz();
  sz->plan = subdomain_plan;
  sz->n_sub = n_subdomains;
  sz->n_virtual = nv;
  sz->n_mesh = n_mesh_elements;
  sz->overlap = num_nodes_overlap;
  sz->nodal_size = subdomain_plan->local_nodes;
  // hat weights per degree: [left ramp | right ramp | core], d4est_solver_schwarz_operators.c:78-105
  const int rs = num_nodes_overlap;
  std::vector<int> wtab(Tables1D::kMaxDeg + 2, -1);
  std::vector<double> weights;
  std::vector<VirtDesc> vd(nv);
  std::vector<std::vector<int>> contrib(n_mesh_elements);
  long long mesh_nodes = 0;
  for (int e = 0; e < n_mesh_elements; ++e) {
    const long long N = mesh_deg[e] + 1;
    mesh_nodes = std::max(mesh_nodes, mesh_nodal_stride[e] + N * N * N);
  }
  sz->mesh_nodes = (int)mesh_nodes;
  for (int s = 0; s < n_subdomains; ++s) {
    int n_core = 0;
    for (int v = sub_first[s]; v < sub_first[s + 1]; ++v) {
      const int e = sub_elem[v];
      if (e < 0 || e >= n_mesh_elements) D4EST_HIP_ABORT("schwarz_create: subdomain %d lists element %d (single-rank meshes only)", s, e);
      const int deg = mesh_deg[e];
      if (deg != subdomain_plan->deg[v]) D4EST_HIP_ABORT("schwarz_create: degree of subdomain element %d differs from the plan's", v);
      if (rs > deg + 1) D4EST_HIP_ABORT("schwarz_create: num_nodes_overlap %d exceeds deg + 1 = %d", rs, deg + 1);
      if (wtab[deg] < 0) {
        wtab[deg] = (int)weights.size();
        std::vector<double> x, w;
        Tables1D::lobatto(deg, x, w);
        const double overlap_size = 1. - x[deg + 1 - rs];
        for (int i = 0; i < rs; ++i) weights.push_back(hat(x[i + deg + 1 - rs] - 2, overlap_size));
        for (int i = 0; i < rs; ++i) weights.push_back(hat(x[i] + 2, overlap_size));
        for (int i = 0; i <= deg; ++i) weights.push_back(hat(x[i], overlap_size));
      }
      VirtDesc q{};
      q.dst = subdomain_plan->nodal_stride[v];
      q.src = mesh_nodal_stride[e];
      q.N = deg + 1;
      bool core = true;
      for (int d = 0; d < 3; ++d) { q.lo[d] = 0; q.hi[d] = q.N; q.woff[d] = wtab[deg] + 2 * rs; }
      for (int k = 0; k < 3; ++k) {
        const int f = sub_faces[3 * v + k], cf = sub_core_faces[3 * v + k];
        if ((f == -1) != (cf == -1)) D4EST_HIP_ABORT("schwarz_create: faces / core_faces of subdomain element %d disagree", v);
        if (f == -1) continue;
        if (f < 0 || f > 5 || cf != (f ^ 1)) D4EST_HIP_ABORT("schwarz_create: bad face pair (%d, %d) on subdomain element %d", f, cf, v);
        core = false;
        const int dir = f / 2, side = f % 2;
        // restrictor: side 0 keeps the first rs nodes, side 1 the last (d4est_solver_schwarz_operators.c:52-58)
        if (side == 0) { q.lo[dir] = 0; q.hi[dir] = rs; } else { q.lo[dir] = q.N - rs; q.hi[dir] = q.N; }
        // weights: core face side 0 = element left of the core = ramp [0, rs), side 1 = right = ramp [rs, 2 rs) (:377-385)
        const int cside = cf % 2;
        q.woff[dir] = wtab[deg] + cside * rs - q.lo[dir];
      }
      n_core += core;
      sz->restricted_nodal_size += (long long)(q.hi[0] - q.lo[0]) * (q.hi[1] - q.lo[1]) * (q.hi[2] - q.lo[2]);
      vd[v] = q;
      contrib[e].push_back(v);
    }
    if (n_core != 1) D4EST_HIP_ABORT("schwarz_create: subdomain %d has %d core elements", s, n_core);
  }
  std::vector<int> first(n_mesh_elements + 1, 0), flat, mstride(mesh_nodal_stride, mesh_nodal_stride + n_mesh_elements), mN(n_mesh_elements);
  for (int e = 0; e < n_mesh_elements; ++e) {
    first[e + 1] = first[e] + (int)contrib[e].size();
    flat.insert(flat.end(), contrib[e].begin(), contrib[e].end());
    mN[e] = mesh_deg[e] + 1;
  }
  sz->d_vd = upload(vd);
  sz->d_sub_first = upload(std::vector<int>(sub_first, sub_first + n_subdomains + 1));
  sz->d_mesh_first = upload(first);
  sz->d_mesh_contrib = upload(flat);
  sz->d_mesh_stride = upload(mstride);
  sz->d_mesh_N = upload(mN);
  sz->d_weights = upload(weights);
  return sz;
}

void d4est_hip_schwarz_destroy(d4est_hip_schwarz_t* sz) {
  if (!sz) return;
  void* ptrs[] = {sz->d_vd, sz->d_sub_first, sz->d_mesh_first, sz->d_mesh_contrib, sz->d_mesh_stride, sz->d_mesh_N, sz->d_weights,
                  sz->d_du, sz->d_r, sz->d_d, sz->d_Ad, sz->d_zero_ghost, sz->d_delta, sz->d_tol, sz->d_active, sz->d_final_iter,
                  sz->d_n_active};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete sz;
}

static void check_schwarz(const d4est_hip_schwarz_t* sz, const char* who) {
  if (!sz || !sz->plan) D4EST_HIP_ABORT("%s: NULL schwarz handle", who);
}

long long d4est_hip_schwarz_nodal_size(const d4est_hip_schwarz_t* sz) { check_schwarz(sz, "schwarz_nodal_size"); return sz->nodal_size; }
long long d4est_hip_schwarz_restricted_nodal_size(const d4est_hip_schwarz_t* sz) {
  check_schwarz(sz, "schwarz_restricted_nodal_size");
  return sz->restricted_nodal_size;
}

void d4est_hip_schwarz_restrict_field(d4est_hip_schwarz_t* sz, const double* field_dev, double* out_dev) {
  check_schwarz(sz, "schwarz_restrict_field");
  if (sz->n_virtual == 0) return;
  hipLaunchKernelGGL(schwarz_restrict_kernel, dim3(std::min(sz->n_virtual, 65536)), dim3(256), 0, sz->plan->stream, sz->d_vd,
                     sz->n_virtual, field_dev, out_dev);
  HIP_CHECK(hipGetLastError());
}

void d4est_hip_schwarz_apply_over_subdomains(d4est_hip_schwarz_t* sz, const double* in_dev, double* out_dev) {
  check_schwarz(sz, "schwarz_apply_over_subdomains");
  if (sz->n_virtual == 0) return;
  ensure_zero_ghost(sz);
  d4est_hip_apply_aij(sz->plan, in_dev, sz->d_zero_ghost, out_dev);
  add_lhs_mass_term(sz->plan, in_dev, out_dev);
  hipLaunchKernelGGL(schwarz_mask_kernel, dim3(std::min(sz->n_virtual, 65536)), dim3(256), 0, sz->plan->stream, sz->d_vd, sz->n_virtual,
                     out_dev);
  HIP_CHECK(hipGetLastError());
}

void d4est_hip_schwarz_add_correction(d4est_hip_schwarz_t* sz, const double* du_dev, double* u_dev) {
  check_schwarz(sz, "schwarz_add_correction");
  if (sz->n_mesh == 0) return;
  hipLaunchKernelGGL(schwarz_correction_kernel, dim3(std::min(sz->n_mesh, 65536)), dim3(256), 0, sz->plan->stream, sz->d_vd,
                     sz->d_mesh_first, sz->d_mesh_contrib, sz->d_mesh_stride, sz->d_mesh_N, sz->n_mesh, sz->d_weights, du_dev, u_dev);
  HIP_CHECK(hipGetLastError());
}

int d4est_hip_schwarz_iterate(d4est_hip_schwarz_t* sz, double* u_dev, const double* r_dev, int subdomain_iter, double subdomain_atol,
                              double subdomain_rtol) {
  check_schwarz(sz, "schwarz_iterate");
  // d4est_solver_schwarz_subdomain_solver_cg.c:86-92
  if (subdomain_iter <= 0 || !(subdomain_rtol > 0) || !(subdomain_atol > 0)) D4EST_HIP_ABORT("schwarz_iterate: some subdomain solver options are <= 0");
  if (sz->n_sub == 0) return 0;
  ensure_workspace(sz);
  ensure_zero_ghost(sz);
  hipStream_t st = sz->plan->stream;
  HIP_CHECK(hipMemsetAsync(sz->d_n_active, 0, 2 * sizeof(int), st));
  hipLaunchKernelGGL(schwarz_cg_init_kernel, dim3(sz->n_sub), dim3(256), 0, st, sz->d_vd, sz->d_sub_first, r_dev, sz->d_du, sz->d_r,
                     sz->d_d, sz->d_delta, sz->d_tol, sz->d_active, sz->d_final_iter, sz->d_n_active, subdomain_iter, subdomain_atol,
                     subdomain_rtol);
  HIP_CHECK(hipGetLastError());
  // A subdomain that has met its tolerance is frozen on the device (its workgroup returns at once), so sweeps past the point where
  // the reference's loops have all broken are no-ops: the host looks at the "still iterating" counter only before the first sweep
  // and then every 8th, instead of synchronising before every sweep; the number of sweeps that did work is counted on the device.
  for (int it = 0; it < subdomain_iter; ++it) {
    if (it % 8 == 0) {
      int n_active = 0;
      HIP_CHECK(hipMemcpyAsync(&n_active, sz->d_n_active, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (n_active == 0) break;  // every subdomain has left its loop
    }
    d4est_hip_apply_aij(sz->plan, sz->d_d, sz->d_zero_ghost, sz->d_Ad);
    add_lhs_mass_term(sz->plan, sz->d_d, sz->d_Ad);   // zeroth-order term of a linearised problem (plan_set_lhs_coefficient on the subdomain plan)
    hipLaunchKernelGGL(schwarz_cg_kernel, dim3(sz->n_sub), dim3(256), 0, st, sz->d_vd, sz->d_sub_first, sz->d_du, sz->d_r, sz->d_d,
                       sz->d_Ad, sz->d_delta, sz->d_tol, sz->d_active, sz->d_final_iter, sz->d_n_active, it);
    HIP_CHECK(hipGetLastError());
  }
  d4est_hip_schwarz_add_correction(sz, sz->d_du, u_dev);
  int counters[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(counters, sz->d_n_active, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return counters[1];
}

void d4est_hip_schwarz_smooth(d4est_hip_schwarz_t* sz, d4est_hip_plan_t* mesh_plan, double* u_dev, const double* rhs_dev, double* r_dev,
                              int smoother_iterations, int subdomain_iter, double subdomain_atol, double subdomain_rtol) {
  check_schwarz(sz, "schwarz_smooth");
  if (!mesh_plan || !mesh_plan->has_faces) D4EST_HIP_ABORT("schwarz_smooth: the mesh plan needs its faces (plan_set_faces)");
  if (mesh_plan->local_nodes != sz->mesh_nodes) D4EST_HIP_ABORT("schwarz_smooth: mesh plan has %d nodes, the smoother's mesh %d", mesh_plan->local_nodes, sz->mesh_nodes);
  if (mesh_plan->stream != sz->plan->stream) D4EST_HIP_ABORT("schwarz_smooth: the mesh plan and the subdomain plan must use the same stream");
  // d4est_solver_multigrid_smoother_schwarz, src/Solver/d4est_solver_multigrid_smoother_schwarz.c:98-196; r doubles as the Au scratch
  for (int i = 0; i < smoother_iterations; ++i) {
    apply_operator(mesh_plan, u_dev, r_dev);
    launch_residual_inplace(mesh_plan, mesh_plan->local_nodes, rhs_dev, r_dev);
    d4est_hip_schwarz_iterate(sz, u_dev, r_dev, subdomain_iter, subdomain_atol, subdomain_rtol);
  }
  apply_operator(mesh_plan, u_dev, r_dev);
  launch_residual_inplace(mesh_plan, mesh_plan->local_nodes, rhs_dev, r_dev);
}

void d4est_hip_schwarz_get_info(d4est_hip_schwarz_t* sz, int* final_iter_host, double* final_res_host) {
  check_schwarz(sz, "schwarz_get_info");
  if (!sz->d_du) D4EST_HIP_ABORT("schwarz_get_info: call schwarz_iterate first");
  HIP_CHECK(hipStreamSynchronize(sz->plan->stream));
  if (final_iter_host) HIP_CHECK(hipMemcpy(final_iter_host, sz->d_final_iter, (size_t)sz->n_sub * sizeof(int), hipMemcpyDeviceToHost));
  if (final_res_host) {
    HIP_CHECK(hipMemcpy(final_res_host, sz->d_delta, (size_t)sz->n_sub * sizeof(double), hipMemcpyDeviceToHost));
    for (int s = 0; s < sz->n_sub; ++s) final_res_host[s] = std::sqrt(final_res_host[s]);
  }
}

}  // extern "C"
