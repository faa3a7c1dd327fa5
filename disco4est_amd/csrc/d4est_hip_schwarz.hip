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
  // flat list of the restricted nodes of every subdomain (field offsets), for the register-resident CG kernel
  int* d_box_off = nullptr;
  int* d_box_first = nullptr;        // n_sub + 1
  int max_box_nodes = 0;
  // condensed copies (see ensure_condensed): copies with a handful of restricted nodes whose rows of the subdomain operator are kept as
  // small dense blocks instead of being applied matrix-free
  std::vector<VirtDesc> h_vd;        // host copy of the copy descriptors
  int condensed_state = 0;           // 0 not looked at yet, 1 in use, -1 not applicable / switched off
  unsigned long long cond_generation = 0;   // the subdomain plan's op_generation the state above belongs to
  int n_cond = 0;
  void* d_cond = nullptr;            // CondDesc per condensed copy
  double* d_cond_blocks = nullptr;
  int* d_keep_list = nullptr;        // the copies that stay matrix-free
  int n_keep = 0;
  int* d_cond_off = nullptr;         // per condensed copy: field offsets of its inputs (own restricted nodes first, then the neighbours')
  long long cond_block_doubles = 0;
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

// The same iteration with the subdomain's restricted nodes taken from a flat offset list and d, A d, r held in registers between the
// three passes (PT nodes per thread): 4 reads + 3 writes per node instead of 8 + 3, and no index arithmetic.  The partial sums are
// formed over a different split of the nodes than in schwarz_cg_kernel (still a fixed order: same result on every launch).
template <int PT>
__global__ __launch_bounds__(256) void schwarz_cg_flat_kernel(const int* __restrict__ box_first, const int* __restrict__ box_off,
                                                              double* __restrict__ du, double* __restrict__ r, double* __restrict__ d,
                                                              const double* __restrict__ Ad, double* __restrict__ delta,
                                                              const double* __restrict__ tol, int* __restrict__ active,
                                                              int* __restrict__ final_iter, int* __restrict__ n_active, int it) {
  __shared__ double red[256];
  const int s = blockIdx.x;
  if (!active[s]) return;
  if (threadIdx.x == 0) atomicMax(n_active + 1, it + 1);   // this sweep did work: the reference's loop was still running
  const int first = box_first[s], cnt = box_first[s + 1] - first;
  int o[PT];
  double dv[PT], av[PT], rv[PT];
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int t = threadIdx.x + 256 * j;
    o[j] = (t < cnt) ? box_off[first + t] : -1;
    dv[j] = (o[j] >= 0) ? d[o[j]] : 0.0;
    av[j] = (o[j] >= 0) ? Ad[o[j]] : 0.0;
    acc += dv[j] * av[j];
  }
  const double d_dot_Ad = block_sum(acc, red);
  const double delta_old = delta[s];
  const double alpha = delta_old / d_dot_Ad;
  acc = 0.0;
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    rv[j] = 0.0;
    if (o[j] >= 0) {
      du[o[j]] += alpha * dv[j];
      rv[j] = r[o[j]] - alpha * av[j];
      r[o[j]] = rv[j];
    }
    acc += rv[j] * rv[j];
  }
  const double delta_new = block_sum(acc, red);
  const double beta = delta_new / delta_old;
#pragma unroll
  for (int j = 0; j < PT; ++j)
    if (o[j] >= 0) d[o[j]] = rv[j] + beta * dv[j];
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

// ---------------------------------------------------------------------------
// Condensed copies.  On a conforming subdomain the 8 corner copies of the 27 carry rs^3 restricted nodes each (8 of 512 at overlap 2,
// p = 7) and still cost a full element apply -- 30 % of the operator's work for 0.5 % of its unknowns.  The rows of R A R^T that belong
// to such a copy c touch only c itself and the copies across its faces:
//     (A d)|c = S_c d|c + sum_f X_{c,f} d|nbr(c,f)          S_c: L_c x L_c,  X_{c,f}: L_c x L_nbr   (L = restricted nodes of a copy)
// These blocks are read off the matrix-free operator itself, once, by probing it with unit vectors (every copy of a round in
// parallel: round (f, m) sets restricted node m of every nbr(c, f) and c reads its column of X_{c,f}; the set-up refuses a mesh on which
// a second neighbour of such a c would be probed in the same round), so they are the SAME numbers the kernels produce, to rounding.  From then on the operator kernel skips
// the condensed copies (they are still read as neighbours) and one small kernel forms their rows: 6.6 KB per corner copy instead of a
// 4 KB + 45 KB element apply.  Only copies whose blocks stay under 8 KB are condensed (at overlap 3 the blocks would cost more to
// stream than the apply they replace).  D4EST_HIP_SCHWARZ_CONDENSE=0 switches it off.  Probing needs a LINEAR operator: the subdomain
// plan carries homogeneous boundary data (the correction's), as disco4est_amd/schwarz.py sets it.
// ---------------------------------------------------------------------------
struct CondDesc {
  int v;          // the condensed copy
  int L;          // its restricted nodes
  int nn;         // neighbours inside the subdomain (<= 6)
  int nb_v[6];    // their copy ids
  int nb_L[6];    // their restricted node counts
  int total;      // L + sum nb_L: columns of [S_c | X_{c,0} | ...]
  long long blk;  // offset of S_c in the block array; X_{c,k} follow in neighbour order, all column-major with leading dimension L
  long long off;  // offset of the copy's input offsets in the offset array (total ints; the first L are also its output offsets)
};

// x = unit vectors: restricted node m of the listed copies (x zeroed beforehand)
__global__ __launch_bounds__(256) void schwarz_probe_set_kernel(const VirtDesc* __restrict__ vd, const int* __restrict__ copies, int n_copies,
                                                                int m, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_copies) return;
  const VirtDesc q = vd[copies[i]];
  if (m < restricted_count(q)) x[restricted_offset(q, m)] = 1.0;
}

// column m of S_c (slot < 0) or of X_{c,slot}: the operator's response at the restricted nodes of c
__global__ __launch_bounds__(64) void schwarz_probe_get_kernel(const VirtDesc* __restrict__ vd, const CondDesc* __restrict__ cd, int n_cond,
                                                               int face_slot_of_round, const int* __restrict__ slot_of, int m,
                                                               const double* __restrict__ y, double* __restrict__ blocks) {
  const int ci = blockIdx.x;
  if (ci >= n_cond) return;
  const CondDesc c = cd[ci];
  const int slot = face_slot_of_round < 0 ? -1 : slot_of[ci];   // neighbour slot of c probed in this round, -1: none (or the self round)
  if (face_slot_of_round >= 0 && slot < 0) return;
  const int Lcol = slot < 0 ? c.L : c.nb_L[slot];
  if (m >= Lcol) return;
  long long off = c.blk;
  if (slot >= 0) {
    off += (long long)c.L * c.L;
    for (int k = 0; k < slot; ++k) off += (long long)c.L * c.nb_L[k];
  }
  const VirtDesc q = vd[c.v];
  for (int r = threadIdx.x; r < c.L; r += blockDim.x) blocks[off + (long long)m * c.L + r] = y[restricted_offset(q, r)];
}

// (A d)|c for every condensed copy: one wavefront per copy.  S_c and the X_{c,k} are ONE column-major L x total matrix; lane
// t = r + L * part takes row r and the columns m = part, part + P, ... (P = 64 / L parts), so that for every step the wave reads one
// contiguous run of the block; the P partial sums of a row are added in a fixed order through LDS.
__global__ __launch_bounds__(64) void schwarz_condensed_apply_kernel(const CondDesc* __restrict__ cd, int n_cond, const int* __restrict__ offs,
                                                                     const double* __restrict__ blocks, const double* __restrict__ in,
                                                                     double* __restrict__ out) {
  __shared__ double s_in[7 * 128];
  __shared__ double s_part[64];
  const int ci = blockIdx.x;
  if (ci >= n_cond) return;
  const CondDesc c = cd[ci];
  const int* __restrict__ io = offs + c.off;
  for (int m = threadIdx.x; m < c.total; m += blockDim.x) s_in[m] = in[io[m]];
  __syncthreads();
  const int P = 64 / c.L;                       // L <= 27: P >= 2
  const int r = threadIdx.x % c.L, part = threadIdx.x / c.L;
  const double* __restrict__ B = blocks + c.blk + r;
  double acc = 0.0;
  if (part < P) {
#pragma unroll 4
    for (int m = part; m < c.total; m += P) acc = fma(B[(long long)m * c.L], s_in[m], acc);
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < c.L) {
    double sum = 0.0;
    for (int p2 = 0; p2 < P; ++p2) sum += s_part[threadIdx.x + c.L * p2];
    out[io[threadIdx.x]] = sum;
  }
}

static int live_count(const VirtDesc& q) { return (q.hi[0] - q.lo[0]) * (q.hi[1] - q.lo[1]) * (q.hi[2] - q.lo[2]); }

static void ensure_zero_ghost(d4est_hip_schwarz* sz);

static void release_condensed(d4est_hip_schwarz* sz) {
  (void)hipFree(sz->d_cond); (void)hipFree(sz->d_cond_blocks); (void)hipFree(sz->d_cond_off); (void)hipFree(sz->d_keep_list);
  sz->d_cond = nullptr; sz->d_cond_blocks = nullptr; sz->d_cond_off = nullptr; sz->d_keep_list = nullptr;
  sz->n_cond = 0; sz->n_keep = 0; sz->cond_block_doubles = 0;
  sz->condensed_state = 0;
}

static void ensure_condensed(d4est_hip_schwarz* sz) {
  d4est_hip_plan* plan = sz->plan;
  // the blocks are the subdomain operator's own numbers, read off by probing: any later change of that operator (SIPG parameters,
  // geometry, boundary data, the zeroth-order coefficient, tuning) makes them stale -- probe again instead of mixing two operators
  if (sz->condensed_state != 0 && sz->cond_generation != plan->op_generation) release_condensed(sz);
  if (sz->condensed_state != 0) return;
  sz->condensed_state = -1;
  sz->cond_generation = plan->op_generation;
  const char* env = std::getenv("D4EST_HIP_SCHWARZ_CONDENSE");   // (read per smoother object: once, on its first use)
  static const bool dbg = std::getenv("D4EST_HIP_DEBUG_FUSED") != nullptr;
  auto no = [&](const char* why) {
    if (dbg) std::fprintf(stderr, "[d4est_hip] schwarz: no condensed copies: %s\n", why);
  };
  if (env && std::atoi(env) == 0) return no("switched off");
  if (!direct_active(plan) || !plan->has_face_geometry || !plan->has_geometry || sz->n_virtual == 0) return no("no direct face kernel on the subdomain plan");
  if (!plan->side_hang.empty()) return no("hanging faces");
  // probing with unit vectors reads A e_j: with non-zero Dirichlet or Robin data on the subdomain plan the operator is affine and its
  // constant part would land in every probed column (the smoother's subdomain plan carries homogeneous data: schwarz.py)
  if (plan->bc_inhomogeneous) return no("inhomogeneous boundary data on the subdomain plan");
  const int nv = sz->n_virtual;
  const std::vector<VirtDesc>& vd = sz->h_vd;
  if ((int)vd.size() != nv) return no("descriptor count");
  // candidates: copies (not cores) with few restricted nodes, fewest first; a candidate is condensed when none of its neighbours is
  // (condensed copies form an independent set: the rows of one never involve another) and its blocks stay small
  std::vector<int> order;
  for (int v = 0; v < nv; ++v)
    if (live_count(vd[v]) <= 27 && live_count(vd[v]) < vd[v].N * vd[v].N * vd[v].N) order.push_back(v);
  std::stable_sort(order.begin(), order.end(), [&](int a_, int b_) { return live_count(vd[a_]) < live_count(vd[b_]); });
  std::vector<char> taken(nv, 0);
  std::vector<CondDesc> cds;
  std::vector<int> offs;           // field offsets of every condensed copy's inputs
  std::vector<int> face_of_slot;   // per condensed copy and slot: the face of c it sits on
  long long blk = 0;
  for (int v : order) {
    CondDesc c{};
    c.v = v; c.L = live_count(vd[v]); c.nn = 0; c.blk = blk;
    bool ok = true;
    long long doubles = (long long)c.L * c.L;
    int faces[6];
    for (int f = 0; f < 6 && ok; ++f) {
      const int nb = plan->side_nbr[6 * (size_t)v + f];
      if (nb < 0) continue;
      if (taken[nb] || live_count(vd[nb]) > 128) { ok = false; break; }
      faces[c.nn] = f;
      c.nb_v[c.nn] = nb; c.nb_L[c.nn] = live_count(vd[nb]);
      doubles += (long long)c.L * c.nb_L[c.nn];
      ++c.nn;
    }
    if (!ok || doubles * 8 > 8192) continue;
    int total = c.L;
    for (int k = 0; k < c.nn; ++k) total += c.nb_L[k];
    if (total > 7 * 128) continue;
    c.total = total;
    c.off = (long long)offs.size();
    auto push_offsets = [&](const VirtDesc& q) {   // restricted_offset on the host
      const int e0 = q.hi[0] - q.lo[0], e1 = q.hi[1] - q.lo[1], n = live_count(q);
      for (int i = 0; i < n; ++i) offs.push_back(q.dst + (q.lo[0] + i % e0) + q.N * ((q.lo[1] + (i / e0) % e1) + q.N * (q.lo[2] + i / (e0 * e1))));
    };
    push_offsets(vd[v]);
    for (int k = 0; k < c.nn; ++k) push_offsets(vd[c.nb_v[k]]);
    for (int k = 0; k < c.nn; ++k) face_of_slot.push_back(faces[k]);
    for (int k = c.nn; k < 6; ++k) face_of_slot.push_back(-1);
    cds.push_back(c);
    taken[v] = 1;
    blk += doubles;
  }
  const int nc = (int)cds.size();
  if (nc == 0) return no("no copy qualifies");
  // rounds: round f probes nbr(c, f) of every condensed c.  Refuse if a probed copy is ALSO another neighbour of some condensed copy
  // (its response would land in the wrong block).
  std::vector<std::vector<int>> probed(6);
  std::vector<std::vector<int>> slot_of(6, std::vector<int>(nc, -1));
  std::vector<char> mark(nv, 0);
  for (int f = 0; f < 6; ++f) {
    for (int ci = 0; ci < nc; ++ci)
      for (int k = 0; k < cds[ci].nn; ++k)
        if (face_of_slot[6 * (size_t)ci + k] == f) { probed[f].push_back(cds[ci].nb_v[k]); slot_of[f][ci] = k; }
    for (int e : probed[f]) mark[e] = 1;
    for (int ci = 0; ci < nc; ++ci) {
      if (slot_of[f][ci] < 0) continue;   // c takes nothing from this round
      for (int k = 0; k < cds[ci].nn; ++k)
        if (face_of_slot[6 * (size_t)ci + k] != f && mark[cds[ci].nb_v[k]]) return no("two neighbours of one condensed copy fall into the same probing round");   // keep the matrix-free rows
    }
    for (int e : probed[f]) mark[e] = 0;
  }
  // ---- probe
  ensure_zero_ghost(sz);
  hipStream_t st = plan->stream;
  double *x = nullptr, *y = nullptr;
  const size_t nbytes = std::max<size_t>((size_t)sz->nodal_size, 1) * sizeof(double);
  HIP_CHECK(hipMalloc(&x, nbytes));
  HIP_CHECK(hipMalloc(&y, nbytes));
  CondDesc* d_cd = upload(cds);
  HIP_CHECK(hipMalloc(&sz->d_cond_blocks, std::max<long long>(blk, 1) * sizeof(double)));
  std::vector<int> self(nc);
  int maxL = 0;
  for (int ci = 0; ci < nc; ++ci) { self[ci] = cds[ci].v; maxL = std::max(maxL, cds[ci].L); }
  int* d_self = upload(self);
  for (int m = 0; m < maxL; ++m) {
    HIP_CHECK(hipMemsetAsync(x, 0, nbytes, st));
    hipLaunchKernelGGL(schwarz_probe_set_kernel, dim3((nc + 255) / 256), dim3(256), 0, st, sz->d_vd, d_self, nc, m, x);
    d4est_hip_apply_aij(plan, x, sz->d_zero_ghost, y);
    hipLaunchKernelGGL(schwarz_probe_get_kernel, dim3(nc), dim3(64), 0, st, sz->d_vd, d_cd, nc, -1, (const int*)nullptr, m, y, sz->d_cond_blocks);
  }
  for (int f = 0; f < 6; ++f) {
    if (probed[f].empty()) continue;
    int* d_probed = upload(probed[f]);
    int* d_slot = upload(slot_of[f]);
    int maxLe = 0;
    for (int e : probed[f]) maxLe = std::max(maxLe, live_count(vd[e]));
    const int np = (int)probed[f].size();
    for (int m = 0; m < maxLe; ++m) {
      HIP_CHECK(hipMemsetAsync(x, 0, nbytes, st));
      hipLaunchKernelGGL(schwarz_probe_set_kernel, dim3((np + 255) / 256), dim3(256), 0, st, sz->d_vd, d_probed, np, m, x);
      d4est_hip_apply_aij(plan, x, sz->d_zero_ghost, y);
      hipLaunchKernelGGL(schwarz_probe_get_kernel, dim3(nc), dim3(64), 0, st, sz->d_vd, d_cd, nc, f, d_slot, m, y, sz->d_cond_blocks);
    }
    HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(d_probed); (void)hipFree(d_slot);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(d_self);
  std::vector<char> skip(nv, 0);
  for (const CondDesc& c : cds) skip[c.v] = 1;
  std::vector<int> keep;
  for (int v = 0; v < nv; ++v)
    if (!skip[v]) keep.push_back(v);
  sz->d_keep_list = upload(keep);
  sz->n_keep = (int)keep.size();
  sz->d_cond_off = upload(offs);
  sz->d_cond = d_cd;
  sz->n_cond = nc;
  sz->cond_block_doubles = blk;
  sz->condensed_state = 1;
  if (dbg) std::fprintf(stderr, "[d4est_hip] schwarz: %d of %d copies condensed, %.1f MB of blocks\n", nc, nv, blk * 8e-6);
}

// A restricted to every subdomain, applied to a field over the subdomains (restricted input: zero outside the overlap)
static void subdomain_operator(d4est_hip_schwarz* sz, const double* in, double* out) {
  ensure_condensed(sz);
  d4est_hip_plan* plan = sz->plan;
  const bool cond = sz->condensed_state == 1 && direct_active(plan);
  if (cond) direct_set_element_list(plan, sz->d_keep_list, sz->n_keep);
  d4est_hip_apply_aij(plan, in, sz->d_zero_ghost, out);
  if (cond) {
    direct_set_element_list(plan, nullptr, 0);
    hipLaunchKernelGGL(schwarz_condensed_apply_kernel, dim3(sz->n_cond), dim3(64), 0, plan->stream, (const CondDesc*)sz->d_cond, sz->n_cond,
                       sz->d_cond_off, sz->d_cond_blocks, in, out);
    HIP_CHECK(hipGetLastError());
  }
  add_lhs_mass_term(plan, in, out);
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
  sz->h_vd = vd;
  {
    std::vector<int> box_first(n_subdomains + 1, 0), box_off;
    bool fits = true;   // field offsets are ints in the descriptors already (dst)
    for (int s_ = 0; s_ < n_subdomains; ++s_) {
      for (int v = sub_first[s_]; v < sub_first[s_ + 1]; ++v) {
        const VirtDesc& q = vd[v];
        const int e0 = q.hi[0] - q.lo[0], e1 = q.hi[1] - q.lo[1], e2 = q.hi[2] - q.lo[2];
        for (int n = 0; n < e0 * e1 * e2; ++n)
          box_off.push_back(q.dst + (q.lo[0] + n % e0) + q.N * ((q.lo[1] + (n / e0) % e1) + q.N * (q.lo[2] + n / (e0 * e1))));
      }
      if (box_off.size() > (size_t)0x7fffffff) { fits = false; break; }
      box_first[s_ + 1] = (int)box_off.size();
      sz->max_box_nodes = std::max(sz->max_box_nodes, box_first[s_ + 1] - box_first[s_]);
    }
    if (fits) {
      sz->d_box_off = upload(box_off);
      sz->d_box_first = upload(box_first);
    }
  }
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
                  sz->d_n_active, sz->d_cond, sz->d_cond_blocks, sz->d_keep_list, sz->d_cond_off, sz->d_box_off, sz->d_box_first};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete sz;
}

static void check_schwarz(const d4est_hip_schwarz_t* sz, const char* who) {
  if (!sz || !sz->plan) D4EST_HIP_ABORT("%s: NULL schwarz handle", who);
}

int d4est_hip_schwarz_condensed_copies(d4est_hip_schwarz_t* sz) {
  check_schwarz(sz, "schwarz_condensed_copies");
  ensure_condensed(sz);
  return sz->condensed_state == 1 ? sz->n_cond : 0;
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
  subdomain_operator(sz, in_dev, out_dev);
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
    subdomain_operator(sz, sz->d_d, sz->d_Ad);   // (+ the zeroth-order term of a linearised problem: plan_set_lhs_coefficient on the subdomain plan)
    if (sz->d_box_off && sz->max_box_nodes <= 256 * 8)
      hipLaunchKernelGGL(schwarz_cg_flat_kernel<8>, dim3(sz->n_sub), dim3(256), 0, st, sz->d_box_first, sz->d_box_off, sz->d_du, sz->d_r,
                         sz->d_d, sz->d_Ad, sz->d_delta, sz->d_tol, sz->d_active, sz->d_final_iter, sz->d_n_active, it);
    else if (sz->d_box_off && sz->max_box_nodes <= 256 * 16)
      hipLaunchKernelGGL(schwarz_cg_flat_kernel<16>, dim3(sz->n_sub), dim3(256), 0, st, sz->d_box_first, sz->d_box_off, sz->d_du, sz->d_r,
                         sz->d_d, sz->d_Ad, sz->d_delta, sz->d_tol, sz->d_active, sz->d_final_iter, sz->d_n_active, it);
    else
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
