// The multigrid MATRIX OPERATOR on the device (last column; verdict row a14).
//
// A linearised nonlinear problem adds a zeroth-order term  V^T W J f(x, u0) V u  to the Laplacian (d4est_quadrature_apply_fofufofvlilj,
// src/Quadrature/d4est_quadrature.c:593-774).  On the FINEST multigrid level the reference applies it matrix-free; on every coarser
// level its smoother applies the Galerkin-restricted term instead:
//   d4est_solver_multigrid_matrix_setup_fofufofvlilj_operator  (src/Solver/d4est_solver_multigrid_matrix_operator.c:160-245)
//       one dense (deg+1)^3 x (deg+1)^3 block per fine element, QUAD_COMPUTE_MATRIX (d4est_quadrature.c:748-760, :1143-1186)
//   d4est_solver_multigrid_matrix_operator_restriction_callback (:6-48) -> d4est_operators_compute_PT_mat_P
//       (src/dGMath/d4est_operators.c:608-667): per coarse element  sum_children P_c^T M_c P_c
//   constant_density_star_apply_jac_add_nonlinear_term_using_matrix
//       (src/Problems/ConstantDensityStar/constant_density_star_fcns.h:485-527, selected at :806-850 when matrix != matrix_at0):
//       Au += M_e u_e, a dense matvec per element
// Three device forms, all behind d4est_hip_apply_lhs / _cheby_iterate / _cg_eigs / the Schwarz subdomain operator:
//   (1) dense element blocks (d4est_hip_plan_set_lhs_element_blocks): the reference's data structure; HBM-bound block stream,
//       8 (deg+1)^3 bytes per DoF;
//   (2) the Galerkin chain (d4est_hip_plan_set_lhs_galerkin_chain): the same operator applied matrix-free as
//       T_0^T ... T_{k-1}^T (V^T W J c V) T_{k-1} ... T_0 u  through the transfer objects and the fine level's mass kernel -- streams the
//       fine level's coefficient (8 B per fine quadrature node) instead of the blocks;
//   (3) the blocks themselves are built on the device: d4est_hip_compute_weighted_mass_blocks (QUAD_COMPUTE_MATRIX for every element)
//       and d4est_hip_transfer_galerkin_blocks (the restriction callback for every coarse element).
#include <algorithm>
#include <map>
#include <vector>

#include "d4est_hip_internal.h"
#include "d4est_hip_tables.h"
#include "d4est_hip_transfer.h"

namespace d4est_hip {

struct LhsChain {
  std::vector<d4est_hip_transfer*> t;   // t[0]: this plan's level <-> the next finer one, ..., t.back(): <-> the fine plan's level
  d4est_hip_plan* fine = nullptr;
  std::vector<double*> x, y;            // per finer level: the prolonged vector and the term on its way back
  bool fused = false;                   // one transfer, every item within the compile-time sizes: galerkin_fast_kernel (d4est_hip_transfer.hip)
};

// ---- (1) Au_e += M_e u_e --------------------------------------------------------------------------------------------------------
// One workgroup per (element, chunk of rows); a wavefront owns rows r, r + 4, ...: its lanes stride along the row (512 contiguous bytes
// per wave load, non-temporal: every block entry is read exactly once per apply), u_e sits in the LDS, the 64 partial sums are folded
// with cross-lane adds.  Row sums are complete in one wave in a fixed order: deterministic.
__global__ __launch_bounds__(256) void block_matvec_add_kernel(const int* __restrict__ elem_ids, const int* __restrict__ ns_list, int N3,
                                                               const double* __restrict__ blocks, const long long* __restrict__ block_off,
                                                               const double* __restrict__ u, double* __restrict__ Au, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) double su[];
  const int i = blockIdx.x;
  const int e = elem_ids[i];
  const int ns = ns_list[i];
  const double* __restrict__ M = blocks + block_off[e];
  for (int k = threadIdx.x; k < N3; k += blockDim.x) su[k] = u[ns + k];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = blockIdx.y * rows_per_wg;
  const int r1 = min(N3, r0 + rows_per_wg);
  // two rows per trip: twice the loads in flight per wavefront
  for (int r = r0 + wave; r < r1; r += 8) {
    const int rb = r + 4;
    const double* __restrict__ row_a = M + (long long)r * N3;
    const double* __restrict__ row_b = M + (long long)(rb < r1 ? rb : r) * N3;
    double sa = 0.0, sb = 0.0;
    for (int c = lane; c < N3; c += 64) {
      const double x = su[c];
      sa = fma(__builtin_nontemporal_load(row_a + c), x, sa);
      sb = fma(__builtin_nontemporal_load(row_b + c), x, sb);
    }
    for (int o = 32; o > 0; o >>= 1) {
      sa += __shfl_xor(sa, o);
      sb += __shfl_xor(sb, o);
    }
    if (lane == 0) {
      Au[ns + r] = __dadd_rn(Au[ns + r], sa);
      if (rb < r1) Au[ns + rb] = __dadd_rn(Au[ns + rb], sb);
    }
  }
}

void add_lhs_blocks_term(d4est_hip_plan* plan, const double* u, double* Au) {
  if (!plan->d_lhs_blocks || plan->local_nodes == 0) return;
  for (const Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    const int N3 = bk.N * bk.N * bk.N;
    // 64 rows per workgroup on large buckets; fewer (down to one trip of the four wavefronts, 8 rows) where that is needed to put
    // ~16 workgroups on every CU: coarse multigrid levels have few elements
    int rows = 64;
    while (rows > 8 && (long long)bk.n_elem * ((N3 + rows - 1) / rows) < 4096) rows >>= 1;
    const dim3 grid(bk.n_elem, (N3 + rows - 1) / rows);
    hipLaunchKernelGGL(block_matvec_add_kernel, grid, dim3(256), (size_t)N3 * sizeof(double), plan->stream, plan->d_elem_ids + bk.elem_offset,
                       plan->d_ns_list + bk.elem_offset, N3, plan->d_lhs_blocks, plan->d_lhs_block_off, u, Au, rows);
  }
  HIP_CHECK(hipGetLastError());
}

// ---- (2) the Galerkin chain -----------------------------------------------------------------------------------------------------
static LhsChain* chain_of(d4est_hip_plan* plan) { return static_cast<LhsChain*>(plan->lhs_chain); }

// the fine plan's w J c, (re)built on the given stream if its inputs changed
static const double* ensure_lhs_wjc_on(d4est_hip_plan* fine, hipStream_t st) {
  hipStream_t saved = fine->stream;
  fine->stream = st;
  const double* w = ensure_lhs_wjc(fine);
  fine->stream = saved;
  return w;
}

void lhs_chain_destroy(d4est_hip_plan* plan) {
  LhsChain* ch = chain_of(plan);
  if (!ch) return;
  for (double* p : ch->x) (void)hipFree(p);
  for (double* p : ch->y) (void)hipFree(p);
  delete ch;
  plan->lhs_chain = nullptr;
}

__global__ __launch_bounds__(256) void mg_add_kernel(int n, const double* __restrict__ x, double* __restrict__ y) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = __dadd_rn(y[i], x[i]);
}

void add_lhs_chain_term(d4est_hip_plan* plan, const double* u, double* Au) {
  LhsChain* ch = chain_of(plan);
  if (!ch || plan->local_nodes == 0) return;
  d4est_hip_plan* fine = ch->fine;
  if (!fine->d_lhs_coeff) D4EST_HIP_ABORT("Galerkin chain: the fine plan has no coefficient (d4est_hip_plan_set_lhs_coefficient on the fine plan)");
  const int k = (int)ch->t.size();
  if (ch->fused) {   // prolong, interpolate, weigh, and back in ONE kernel: only the fine level's w J c is streamed
    galerkin_fused_apply(ch->t[0], ensure_lhs_wjc_on(fine, plan->stream), u, Au, plan->stream);
    return;
  }
  // everything runs on THIS plan's stream (the transfer objects and the fine plan may have been given other streams by their owner)
  hipStream_t st = plan->stream;
  std::vector<hipStream_t> saved(k);
  for (int i = 0; i < k; ++i) { saved[i] = ch->t[i]->stream; ch->t[i]->stream = st; }
  hipStream_t fine_saved = fine->stream;
  fine->stream = st;
  const double* cur = u;
  for (int i = 0; i < k; ++i) {
    d4est_hip_transfer_prolong(ch->t[i], cur, ch->x[i]);
    cur = ch->x[i];
  }
  launch_mass_like(fine, 3, cur, ch->y[k - 1], fine->d_lhs_c, 0);
  for (int i = k - 1; i >= 1; --i) d4est_hip_transfer_restrict(ch->t[i], ch->y[i], ch->y[i - 1]);
  if (!plan->d_work_m) HIP_CHECK(hipMalloc(&plan->d_work_m, (size_t)plan->local_nodes * sizeof(double)));
  d4est_hip_transfer_restrict(ch->t[0], ch->y[0], plan->d_work_m);
  const int n = plan->local_nodes;
  hipLaunchKernelGGL(mg_add_kernel, dim3(std::max(1, std::min((n + 255) / 256, 4096))), dim3(256), 0, st, n, plan->d_work_m, Au);
  HIP_CHECK(hipGetLastError());
  for (int i = 0; i < k; ++i) ch->t[i]->stream = saved[i];
  fine->stream = fine_saved;
}

// ---- (3a) QUAD_COMPUTE_MATRIX for every element ---------------------------------------------------------------------------------
// Column j of block e is V^T (W J c) V e_j (d4est_quadrature_compute_mass_matrix, d4est_quadrature.c:1143-1186: apply to unit vectors,
// d4est_linalg_set_column).  V e_j is the product of three columns of the 1-D interpolation (the forward passes of a unit vector
// multiply by exact zeros and ones), so each column costs the three transposed passes only.
__global__ __launch_bounds__(256) void mass_blocks_kernel(const int* __restrict__ elem_ids, const int* __restrict__ qs_list, int N, int NQ,
                                                          const double* __restrict__ B, const double* __restrict__ w,
                                                          const double* __restrict__ J, const double* __restrict__ c,
                                                          double* __restrict__ blocks, const long long* __restrict__ block_off, int n3max,
                                                          int cols_per_wg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* a = smem;
  double* b = smem + n3max;
  const int i = blockIdx.x;
  const int e = elem_ids[i];
  const int qs = qs_list[i];
  const int N3 = N * N * N, NQ3 = NQ * NQ * NQ;
  double* __restrict__ M = blocks + block_off[e];
  const int j1 = min(N3, (int)(blockIdx.y + 1) * cols_per_wg);
  for (int j = blockIdx.y * cols_per_wg; j < j1; ++j) {
    const int jx = j % N, jy = (j / N) % N, jz = j / (N * N);
    for (int q = threadIdx.x; q < NQ3; q += blockDim.x) {
      const int qa = q % NQ, qb = (q / NQ) % NQ, qk = q / (NQ * NQ);
      const double v = B[qk * N + jz] * (B[qb * N + jy] * B[qa * N + jx]);
      const double wjc = (w[qk] * (w[qb] * w[qa])) * (c ? J[qs + q] * c[qs + q] : J[qs + q]);
      a[q] = wjc * v;
    }
    __syncthreads();
    tensor3<true>(B, B, B, NQ, N, a, b);
    for (int r = threadIdx.x; r < N3; r += blockDim.x) M[(long long)r * N3 + j] = b[r];
    __syncthreads();
  }
}

// ---- (3b) the restriction of the blocks:  coarse_k = sum_c P_c^T M_c P_c ---------------------------------------------------------
// Sum-factorised in two sweeps.  Rows: T_c = M_c P_c, row r of T_c is P_c^T applied to row r of M_c (three 1-D passes).  Columns:
// column j of the coarse block is sum_c P_c^T (column j of T_c).  (The reference forms both products with dense dgemm; the sums are
// re-associated here, nothing else changes.)
__global__ __launch_bounds__(256) void galerkin_rows_kernel(const double* __restrict__ fine, const int* __restrict__ child,
                                                            const long long* __restrict__ moff, const double* __restrict__ ops,
                                                            double* __restrict__ work, int max_n3, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* a = smem;
  double* b = smem + max_n3;
  const int c = blockIdx.x;
  const int* d = child + 8 * c;
  const int NH = d[1], Nh = d[2];
  const int Nh3 = Nh * Nh * Nh, NH3 = NH * NH * NH;
  const double* __restrict__ M = fine + moff[2 * c];
  double* __restrict__ T = work + moff[2 * c + 1];
  const int r1 = min(Nh3, (int)(blockIdx.y + 1) * rows_per_wg);
  for (int r = blockIdx.y * rows_per_wg; r < r1; ++r) {
    for (int i = threadIdx.x; i < Nh3; i += blockDim.x) a[i] = M[(long long)r * Nh3 + i];
    __syncthreads();
    tensor3<true>(ops + d[3], ops + d[4], ops + d[5], Nh, NH, a, b);
    for (int i = threadIdx.x; i < NH3; i += blockDim.x) T[(long long)r * NH3 + i] = b[i];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void galerkin_cols_kernel(const double* __restrict__ work, const int* __restrict__ child,
                                                            const long long* __restrict__ moff, const long long* __restrict__ coff,
                                                            const int* __restrict__ item_first, const double* __restrict__ ops,
                                                            double* __restrict__ coarse, int max_n3, int cols_per_wg, int acc_in_lds) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* a = smem;
  double* b = smem + max_n3;
  double* acc_lds = smem + 2 * max_n3;
  const int it = blockIdx.x;
  const int c0 = item_first[it], c1 = item_first[it + 1];
  const int NH = child[8 * c0 + 1];
  const int NH3 = NH * NH * NH;
  double* __restrict__ out = coarse + coff[it];
  const int j1 = min(NH3, (int)(blockIdx.y + 1) * cols_per_wg);
  for (int j = blockIdx.y * cols_per_wg; j < j1; ++j) {
    // entry (i, j) is touched by ONE thread only, in program order: the sum over the children may live in the output column itself
    for (int i = threadIdx.x; i < NH3; i += blockDim.x) {
      if (acc_in_lds) acc_lds[i] = 0.0;
      else out[(long long)i * NH3 + j] = 0.0;
    }
    for (int c = c0; c < c1; ++c) {
      const int* d = child + 8 * c;
      const int Nh = d[2];
      const int Nh3 = Nh * Nh * Nh;
      const double* __restrict__ T = work + moff[2 * c + 1];
      for (int i = threadIdx.x; i < Nh3; i += blockDim.x) a[i] = T[(long long)i * NH3 + j];
      __syncthreads();
      tensor3<true>(ops + d[3], ops + d[4], ops + d[5], Nh, NH, a, b);
      for (int i = threadIdx.x; i < NH3; i += blockDim.x) {
        if (acc_in_lds) acc_lds[i] += b[i];
        else out[(long long)i * NH3 + j] += b[i];
      }
      __syncthreads();
    }
    if (acc_in_lds)
      for (int i = threadIdx.x; i < NH3; i += blockDim.x) out[(long long)i * NH3 + j] = acc_lds[i];
    __syncthreads();
  }
}

// the reference's arithmetic to the letter on items with eight children (d4est_operators.c:637, :651): the left factor of child c is a
// window of the transposed STACKED prolongation, not P_c^T; it comes as a dense NH^3 x Nh^3 matrix per (item, child)
__global__ __launch_bounds__(256) void galerkin_cols_window_kernel(const double* __restrict__ work, const int* __restrict__ child,
                                                                   const long long* __restrict__ moff, const long long* __restrict__ coff,
                                                                   const int* __restrict__ item_first, const double* __restrict__ window,
                                                                   const long long* __restrict__ woff, double* __restrict__ coarse,
                                                                   int cols_per_wg) {
  extern __shared__ __attribute__((aligned(16))) double a[];
  const int it = blockIdx.x;
  const int c0 = item_first[it], c1 = item_first[it + 1];
  if (c1 - c0 != 8) return;   // one child: the window IS P_0^T, the sum-factorised kernel has written the block
  const int NH = child[8 * c0 + 1];
  const int NH3 = NH * NH * NH;
  double* __restrict__ out = coarse + coff[it];
  const int j1 = min(NH3, (int)(blockIdx.y + 1) * cols_per_wg);
  for (int j = blockIdx.y * cols_per_wg; j < j1; ++j) {
    for (int i = threadIdx.x; i < NH3; i += blockDim.x) out[(long long)i * NH3 + j] = 0.0;
    for (int c = c0; c < c1; ++c) {
      const int Nh = child[8 * c + 2];
      const int Nh3 = Nh * Nh * Nh;
      const double* __restrict__ T = work + moff[2 * c + 1];
      const double* __restrict__ W = window + woff[c];
      for (int i = threadIdx.x; i < Nh3; i += blockDim.x) a[i] = T[(long long)i * NH3 + j];
      __syncthreads();
      for (int I = threadIdx.x; I < NH3; I += blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < Nh3; ++i) s = fma(W[(long long)I * Nh3 + i], a[i], s);
        out[(long long)I * NH3 + j] += s;
      }
      __syncthreads();
    }
  }
}

// dense left factors of the literal form for every item with eight children (host)
static void build_windows(d4est_hip_transfer* t) {
  if (t->d_window) return;
  std::vector<double> win;
  std::vector<long long> woff((size_t)std::max(t->n_children, 1), 0);
  std::map<std::vector<int>, long long> seen;   // (degH, degh[8]) -> offset of its 8 windows
  int rec = 0;
  for (int it = 0; it < t->n_items; ++it) {
    const int nc = t->h_hrefine[it] == 1 ? 8 : 1;
    if (nc == 1) { woff[rec++] = 0; continue; }
    const int dH = t->h_degH[it];
    const int* dh = &t->h_degh[8 * (size_t)it];
    std::vector<int> key(dh, dh + 8);
    key.push_back(dH);
    const long long nH = dH + 1, nH3 = nH * nH * nH;
    long long nh3[8], row0[8], total = 0;
    for (int c = 0; c < 8; ++c) { nh3[c] = (long long)(dh[c] + 1) * (dh[c] + 1) * (dh[c] + 1); row0[c] = total; total += nh3[c]; }
    auto f = seen.find(key);
    long long base;
    if (f != seen.end()) base = f->second;
    else {
      base = (long long)win.size();
      seen[key] = base;
      // stacked prolongation P (total x nH3, row-major): child c's rows are the Kronecker product of its three 1-D hp operators
      // (d4est_operators.c:394-404: child c = (cx, cy, cz) bits), then the reference's PT[j * total + i] = P[i * nH3 + j] (:637)
      std::vector<double> PT((size_t)(total * nH3));
      for (int c = 0; c < 8; ++c) {
        const int nh = dh[c] + 1;
        const std::vector<double> P2 = Tables1D::hp_prolong(dH, dh[c]);
        const double* Px = P2.data() + (size_t)(c & 1) * nh * nH;
        const double* Py = P2.data() + (size_t)((c >> 1) & 1) * nh * nH;
        const double* Pz = P2.data() + (size_t)((c >> 2) & 1) * nh * nH;
        for (int iz = 0; iz < nh; ++iz)
          for (int iy = 0; iy < nh; ++iy)
            for (int ix = 0; ix < nh; ++ix) {
              const long long i = row0[c] + ((long long)iz * nh + iy) * nh + ix;
              for (int Iz = 0; Iz < nH; ++Iz)
                for (int Iy = 0; Iy < nH; ++Iy)
                  for (int Ix = 0; Ix < nH; ++Ix) {
                    const long long jj = ((long long)Iz * nH + Iy) * nH + Ix;
                    PT[(size_t)(jj * total + i)] = Pz[iz * nH + Iz] * (Py[iy * nH + Iy] * Px[ix * nH + Ix]);
                  }
            }
      }
      // child c's window: nH3 x nh3[c] doubles starting at stride_P = sum_{c' < c} nh3[c'] nH3 (:651, :658)
      long long stride_P = 0;
      for (int c = 0; c < 8; ++c) {
        win.insert(win.end(), PT.begin() + stride_P, PT.begin() + stride_P + nH3 * nh3[c]);
        stride_P += nh3[c] * nH3;
      }
    }
    long long o = base;
    for (int c = 0; c < 8; ++c) { woff[rec++] = o; o += nH3 * nh3[c]; }
  }
  HIP_CHECK(hipMalloc(&t->d_window, std::max<size_t>(win.size(), 1) * sizeof(double)));
  if (!win.empty()) HIP_CHECK(hipMemcpy(t->d_window, win.data(), win.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_CHECK(hipMalloc(&t->d_woff, woff.size() * sizeof(long long)));
  HIP_CHECK(hipMemcpy(t->d_woff, woff.data(), woff.size() * sizeof(long long), hipMemcpyHostToDevice));
}

static void set_lds(const void* fn, size_t lds) {
  if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

}  // namespace d4est_hip

using namespace d4est_hip;

extern "C" {

long long d4est_hip_plan_matrix_nodes(const d4est_hip_plan_t* plan) {
  if (!plan) D4EST_HIP_ABORT("plan_matrix_nodes: NULL plan");
  long long n = 0;
  for (int d : plan->deg) { const long long n3 = (long long)(d + 1) * (d + 1) * (d + 1); n += n3 * n3; }
  return n;
}

void d4est_hip_compute_weighted_mass_blocks(d4est_hip_plan_t* plan, const double* coeff_quad_dev, double* blocks_dev) {
  if (!plan) D4EST_HIP_ABORT("compute_weighted_mass_blocks: NULL plan");
  if (!plan->has_geometry) D4EST_HIP_ABORT("compute_weighted_mass_blocks: d4est_hip_plan_set_geometry was not called");
  if (plan->n_elements == 0) return;
  if (!blocks_dev) D4EST_HIP_ABORT("compute_weighted_mass_blocks: NULL output");
  // consecutive blocks in element order (matrix_nodal_stride, d4est_solver_multigrid_matrix_operator.c:180-243)
  std::vector<long long> off(plan->n_elements);
  long long o = 0;
  for (int e = 0; e < plan->n_elements; ++e) { off[e] = o; const long long n3 = (long long)(plan->deg[e] + 1) * (plan->deg[e] + 1) * (plan->deg[e] + 1); o += n3 * n3; }
  long long* d_off = nullptr;
  HIP_CHECK(hipMalloc(&d_off, off.size() * sizeof(long long)));
  HIP_CHECK(hipMemcpyAsync(d_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice, plan->stream));
  for (const Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    const int nm = std::max(bk.N, bk.NQ), n3max = nm * nm * nm, N3 = bk.N * bk.N * bk.N;
    const size_t lds = (size_t)2 * n3max * sizeof(double);
    if (lds > 160 * 1024) D4EST_HIP_ABORT("compute_weighted_mass_blocks: degree pair (%d, %d) exceeds the kernel's LDS", bk.deg, bk.deg_quad);
    set_lds(reinterpret_cast<const void*>(mass_blocks_kernel), lds);
    // enough workgroups to fill the chip, at most one per column
    int chunks = std::max(1, std::min(N3, (4096 + bk.n_elem - 1) / bk.n_elem));
    const int cols = (N3 + chunks - 1) / chunks;
    chunks = (N3 + cols - 1) / cols;
    hipLaunchKernelGGL(mass_blocks_kernel, dim3(bk.n_elem, chunks), dim3(256), lds, plan->stream, plan->d_elem_ids + bk.elem_offset,
                       plan->d_qs_list + bk.elem_offset, bk.N, bk.NQ, bk.d_B, bk.d_w, plan->d_J, coeff_quad_dev, blocks_dev, d_off, n3max, cols);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(plan->stream));   // the staged offsets go away with this call
  HIP_CHECK(hipFree(d_off));
}

long long d4est_hip_transfer_fine_matrix_nodes(const d4est_hip_transfer_t* t) { return t ? t->fine_matrix_nodes : -1; }
long long d4est_hip_transfer_coarse_matrix_nodes(const d4est_hip_transfer_t* t) { return t ? t->coarse_matrix_nodes : -1; }

void d4est_hip_transfer_galerkin_blocks(d4est_hip_transfer_t* t, const double* fine_blocks_dev, double* coarse_blocks_dev, int literal_window) {
  if (!t) D4EST_HIP_ABORT("transfer_galerkin_blocks: NULL transfer");
  if (t->n_items == 0) return;
  if (!fine_blocks_dev || !coarse_blocks_dev) D4EST_HIP_ABORT("transfer_galerkin_blocks: NULL blocks");
  if (!t->d_work) HIP_CHECK(hipMalloc(&t->d_work, std::max<size_t>((size_t)t->work_doubles, 1) * sizeof(double)));
  const int n3 = t->max_n * t->max_n * t->max_n;
  const size_t lds2 = (size_t)2 * n3 * sizeof(double);
  set_lds(reinterpret_cast<const void*>(galerkin_rows_kernel), lds2);
  const int per = std::max(1, std::min(n3, (4096 + t->n_children - 1) / t->n_children));   // chunks per child / item
  const int rows = (n3 + per - 1) / per;
  hipLaunchKernelGGL(galerkin_rows_kernel, dim3(t->n_children, (n3 + rows - 1) / rows), dim3(256), lds2, t->stream, fine_blocks_dev,
                     t->d_child, t->d_moff, t->d_ops, t->d_work, n3, rows);
  const int acc_in_lds = ((size_t)3 * n3 * sizeof(double) <= 160 * 1024) ? 1 : 0;
  const size_t lds3 = (size_t)(acc_in_lds ? 3 : 2) * n3 * sizeof(double);
  set_lds(reinterpret_cast<const void*>(galerkin_cols_kernel), lds3);
  const int per_i = std::max(1, std::min(n3, (4096 + t->n_items - 1) / t->n_items));
  const int cols = (n3 + per_i - 1) / per_i;
  hipLaunchKernelGGL(galerkin_cols_kernel, dim3(t->n_items, (n3 + cols - 1) / cols), dim3(256), lds3, t->stream, t->d_work, t->d_child,
                     t->d_moff, t->d_coff, t->d_item_first, t->d_ops, coarse_blocks_dev, n3, cols, acc_in_lds);
  if (literal_window) {
    build_windows(t);
    const size_t lds1 = (size_t)n3 * sizeof(double);
    set_lds(reinterpret_cast<const void*>(galerkin_cols_window_kernel), lds1);
    hipLaunchKernelGGL(galerkin_cols_window_kernel, dim3(t->n_items, (n3 + cols - 1) / cols), dim3(256), lds1, t->stream, t->d_work,
                       t->d_child, t->d_moff, t->d_coff, t->d_item_first, t->d_window, t->d_woff, coarse_blocks_dev, cols);
  }
  HIP_CHECK(hipGetLastError());
}

void d4est_hip_plan_set_lhs_element_blocks(d4est_hip_plan_t* plan, const double* blocks_dev, const long long* block_offset_host) {
  if (!plan) D4EST_HIP_ABORT("plan_set_lhs_element_blocks: NULL plan");
  plan->op_generation++;
  if (plan->cheby_graph) { HIP_CHECK(hipGraphExecDestroy(plan->cheby_graph)); plan->cheby_graph = nullptr; }
  plan->d_lhs_blocks = blocks_dev;
  if (!blocks_dev) return;
  // one form of the zeroth-order term at a time
  plan->d_lhs_coeff = nullptr;
  plan->lhs_wjc_valid = false;
  lhs_chain_destroy(plan);
  std::vector<long long> off(std::max(plan->n_elements, 1), 0);
  long long o = 0;
  for (int e = 0; e < plan->n_elements; ++e) {
    const long long n3 = (long long)(plan->deg[e] + 1) * (plan->deg[e] + 1) * (plan->deg[e] + 1);
    if (block_offset_host) {
      if (block_offset_host[e] < 0) D4EST_HIP_ABORT("plan_set_lhs_element_blocks: element %d has block offset %lld", e, block_offset_host[e]);
      off[e] = block_offset_host[e];
    } else {
      off[e] = o;
      o += n3 * n3;
    }
  }
  if (!plan->d_lhs_block_off) HIP_CHECK(hipMalloc(&plan->d_lhs_block_off, off.size() * sizeof(long long)));
  HIP_CHECK(hipMemcpy(plan->d_lhs_block_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice));
}

void d4est_hip_plan_set_lhs_galerkin_chain(d4est_hip_plan_t* plan, int n_transfers, d4est_hip_transfer_t* const* transfers,
                                           d4est_hip_plan_t* fine_plan) {
  if (!plan) D4EST_HIP_ABORT("plan_set_lhs_galerkin_chain: NULL plan");
  plan->op_generation++;
  if (plan->cheby_graph) { HIP_CHECK(hipGraphExecDestroy(plan->cheby_graph)); plan->cheby_graph = nullptr; }
  lhs_chain_destroy(plan);
  if (n_transfers <= 0 || !transfers || !fine_plan) return;   // chain off
  if (fine_plan == plan) D4EST_HIP_ABORT("plan_set_lhs_galerkin_chain: the fine plan is the plan itself (use plan_set_lhs_coefficient on the finest level)");
  if (!fine_plan->has_geometry) D4EST_HIP_ABORT("plan_set_lhs_galerkin_chain: the fine plan has no geometry");
  plan->d_lhs_coeff = nullptr;
  plan->lhs_wjc_valid = false;
  plan->d_lhs_blocks = nullptr;
  LhsChain* ch = new LhsChain();
  ch->fine = fine_plan;
  long long expect = plan->local_nodes;
  for (int i = 0; i < n_transfers; ++i) {
    d4est_hip_transfer* t = transfers[i];
    if (!t) D4EST_HIP_ABORT("plan_set_lhs_galerkin_chain: transfer %d is NULL", i);
    if (t->coarse_nodes != expect)
      D4EST_HIP_ABORT("plan_set_lhs_galerkin_chain: transfer %d has %lld coarse nodes, the level below it has %lld", i, t->coarse_nodes, expect);
    expect = t->fine_nodes;
    ch->t.push_back(t);
    double *x = nullptr, *y = nullptr;
    HIP_CHECK(hipMalloc(&x, std::max<size_t>((size_t)t->fine_nodes, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc(&y, std::max<size_t>((size_t)t->fine_nodes, 1) * sizeof(double)));
    ch->x.push_back(x);
    ch->y.push_back(y);
  }
  if (expect != fine_plan->local_nodes)
    D4EST_HIP_ABORT("plan_set_lhs_galerkin_chain: the last transfer ends at %lld nodes, the fine plan has %d", expect, fine_plan->local_nodes);
  if (!plan->d_work_m) HIP_CHECK(hipMalloc(&plan->d_work_m, std::max<size_t>((size_t)plan->local_nodes, 1) * sizeof(double)));
  const bool no_fused = std::getenv("D4EST_HIP_CHAIN_UNFUSED") != nullptr;   // (read at every call: the tests switch it)
  ch->fused = !no_fused && n_transfers == 1 && galerkin_fused_setup(ch->t[0], fine_plan);
  plan->lhs_chain = ch;
}

}  // extern "C"
