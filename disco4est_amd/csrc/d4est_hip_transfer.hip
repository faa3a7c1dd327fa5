// hp-multigrid inter-grid transfer on the device.
//
// Replaces the per-element walk of the reference's V-cycle transfer callbacks
//   d4est_solver_multigrid_refine_and_apply_prolongation        (src/Solver/d4est_solver_multigrid_callbacks.h:245-330)
//   d4est_solver_multigrid_apply_restriction (coarsen callback)  (same file :100-200)
// which call, per coarse element,
//   d4est_operators_apply_p_prolong / _hp_prolong                     (src/dGMath/d4est_operators.c:1107-1132, :1091-1105)
//   d4est_operators_apply_p_prolong_transpose / _hp_prolong_transpose (:1719-1749, :1689-1717)
// A transfer object is the flat list of coarse elements ("items") in the traversal order of both grids:
//   hrefine == 0: one fine element of degree degh[0] <-> the coarse element of degree degH (p-coarsening; a copy if equal)
//   hrefine == 1: eight children (z-order, degrees degh[0..7]) <-> their parent (h- and p-coarsening at once)
// Vectors are element-ordered and contiguous on both grids, as in the reference (fine_stride / coarse_stride advance).
#include <map>
#include <vector>

#include "d4est_hip_internal.h"
#include "d4est_hip_tables.h"
#include "d4est_hip_transfer.h"
#include "d4est_hip_wave.h"

namespace d4est_hip {

// one workgroup per fine element
__global__ __launch_bounds__(256) void prolong_kernel(const double* __restrict__ xc, double* __restrict__ xf,
                                                      const int* __restrict__ child, const long long* __restrict__ off,
                                                      const double* __restrict__ ops, const int* __restrict__ list, int n_children, int max_n3) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* a = smem;
  double* b = smem + max_n3;
  for (int ci = blockIdx.x; ci < n_children; ci += gridDim.x) {
    const int c = list[ci];
    const int* d = child + 8 * c;
    const int NH = d[1], Nh = d[2];
    const long long co = off[2 * c], fo = off[2 * c + 1];
    for (int i = threadIdx.x; i < NH * NH * NH; i += blockDim.x) a[i] = xc[co + i];
    __syncthreads();
    tensor3<false>(ops + d[3], ops + d[4], ops + d[5], NH, Nh, a, b);
    for (int i = threadIdx.x; i < Nh * Nh * Nh; i += blockDim.x) xf[fo + i] = b[i];
    __syncthreads();
  }
}

// one workgroup per coarse element: sum over its children of P_c^T x_c (TRANS, ops = prolongations) or of R_c x_c (ops = projections)
template <bool TRANS>
__global__ __launch_bounds__(256) void restrict_kernel(const double* __restrict__ xf, double* __restrict__ xc,
                                                       const int* __restrict__ child, const long long* __restrict__ off,
                                                       const int* __restrict__ item_first, const double* __restrict__ ops,
                                                       const int* __restrict__ list, int n_items, int max_n3, int acc_in_lds) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* a = smem;
  double* b = smem + max_n3;
  for (int ii = blockIdx.x; ii < n_items; ii += gridDim.x) {
    const int it = list[ii];
    const int c0 = item_first[it], c1 = item_first[it + 1];
    const int NH = child[8 * c0 + 1];
    // the sum over the children: a third LDS field, or -- p = 18, 19: three fields of 19^3 / 20^3 doubles exceed the 160 KB -- the coarse
    // element's own entries of the output (every entry is touched by ONE thread only, in program order: no race either way)
    double* acc = acc_in_lds ? smem + 2 * max_n3 : xc + off[2 * c0];
    for (int i = threadIdx.x; i < NH * NH * NH; i += blockDim.x) acc[i] = 0.0;
    for (int c = c0; c < c1; ++c) {
      const int* d = child + 8 * c;
      const int Nh = d[2];
      const long long fo = off[2 * c + 1];
      for (int i = threadIdx.x; i < Nh * Nh * Nh; i += blockDim.x) a[i] = xf[fo + i];
      __syncthreads();
      tensor3<TRANS>(ops + d[3], ops + d[4], ops + d[5], Nh, NH, a, b);
      for (int i = threadIdx.x; i < NH * NH * NH; i += blockDim.x) acc[i] += b[i];   // same thread <-> same entries: no race
      __syncthreads();
    }
    if (acc_in_lds) {
      const long long co = off[2 * c0];
      for (int i = threadIdx.x; i < NH * NH * NH; i += blockDim.x) xc[co + i] = acc[i];
    }
    __syncthreads();
  }
}


// ---- the MI355X-shaped transfer kernels: degree pairs as compile-time constants ---------------------------------------------------
// One workgroup per fine element (prolongation) or per coarse element (restriction / projection: its children in a loop, summed in
// registers in child order).  A thread owns a LINE of the element along the contracted direction in registers; the 1-D operator rows are
// wave-uniform and come through scalar loads (contract_n, d4est_hip_wave.h) -- no operand traffic in the FMA loops, no index division in
// them.  Pass order x, y, z: the element arrives with one coalesced copy into a padded LDS image (odd row length: the x-lines are read
// thread-strided without bank conflicts), the hand-offs between passes are conflict-free by construction, and the LAST pass (z) leaves
// each thread with a z-line whose stores are coalesced over the whole (x, y) plane.  NH = coarse nodes per direction; the fine size
// Nh = NH + d, d <= DMAX, is looked up per fine element (a wave-uniform switch), so mixed-degree children share one launch.
constexpr int kFastMaxNH = 16;
constexpr int kFastMaxD = 3;

template <int NH, int DMAX>
struct TransferCfg {
  static constexpr int NHM = NH + DMAX;                                      // largest fine size
  static constexpr int THREADS = ((NHM * NHM + 63) / 64) * 64;
  // prolongation: B [NH][NH][Nh|1], C [NH][Nh][Nh];  restriction (per child group): B [Nh][Nh][NH|1], C [Nh][NH][NH]
  static constexpr int LDS_PROLONG = NH * NH * (NHM | 1) + NH * NHM * NHM;
  static constexpr int LDS_RESTRICT = NHM * NHM * (NH | 1) + NHM * NH * NH;
  // child groups of the restriction: as many of an item's eight children at once as the LDS and 1024 threads hold
  static constexpr int cg_max() {
    int cg = 8;
    while (cg > 1 && ((long long)cg * LDS_RESTRICT * 8 > 150 * 1024 || cg * THREADS > 1024)) cg >>= 1;
    return cg;
  }
  static constexpr int CGMAX = cg_max();
};

// t: the thread's index in its group of TransferCfg::THREADS threads (the whole workgroup for the prolongation)
template <int NH, int Nh>
__device__ __forceinline__ void prolong_body(const double* __restrict__ xc_e, double* __restrict__ xf_e, const double* PxT,
                                             const double* PyT, const double* PzT, double* lds, int t) {
  constexpr int RSB = Nh | 1;
  double* B = lds;
  double* C = B + NH * NH * RSB;
  if (t < NH * NH) {                       // thread (b, k): its x-line straight from memory (the element is one contiguous run)
    double x[NH], y[Nh];
#pragma unroll
    for (int a = 0; a < NH; ++a) x[a] = xc_e[t * NH + a];
    contract_n<NH, Nh>(PxT, x, y);
#pragma unroll
    for (int a = 0; a < Nh; ++a) B[t * RSB + a] = y[a];
  }
  __syncthreads();
  if (t < Nh * NH) {                       // thread (aq, k): the y-line
    const int aq = t % Nh, k = t / Nh;
    double x[NH], y[Nh];
#pragma unroll
    for (int b = 0; b < NH; ++b) x[b] = B[(k * NH + b) * RSB + aq];
    contract_n<NH, Nh>(PyT, x, y);
#pragma unroll
    for (int b = 0; b < Nh; ++b) C[(k * Nh + b) * Nh + aq] = y[b];
  }
  __syncthreads();
  if (t < Nh * Nh) {                       // thread (aq, bq): the z-line, stored coalesced
    double x[NH], y[Nh];
#pragma unroll
    for (int k = 0; k < NH; ++k) x[k] = C[k * Nh * Nh + t];
    contract_n<NH, Nh>(PzT, x, y);
#pragma unroll
    for (int k = 0; k < Nh; ++k) xf_e[k * Nh * Nh + t] = y[k];
  }
}

template <int NH, int DMAX>
__global__ __launch_bounds__((TransferCfg<NH, DMAX>::THREADS)) void prolong_fast_kernel(const double* __restrict__ xc, double* __restrict__ xf,
                                                                                      const int* __restrict__ child,
                                                                                      const long long* __restrict__ off,
                                                                                      const double* __restrict__ opsT,
                                                                                      const int* __restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int c = list[blockIdx.x];
  const int* d = child + 8 * c;
  const int dN = d[2] - NH;
  const double* xc_e = xc + off[2 * c];
  double* xf_e = xf + off[2 * c + 1];
  const double *px = opsT + d[3], *py = opsT + d[4], *pz = opsT + d[5];
  const int t = threadIdx.x;
  if (dN == 0) prolong_body<NH, NH>(xc_e, xf_e, px, py, pz, smem, t);
  if constexpr (DMAX >= 1) { if (dN == 1) prolong_body<NH, NH + 1>(xc_e, xf_e, px, py, pz, smem, t); }
  if constexpr (DMAX >= 2) { if (dN == 2) prolong_body<NH, NH + 2>(xc_e, xf_e, px, py, pz, smem, t); }
  if constexpr (DMAX >= 3) { if (dN == 3) prolong_body<NH, NH + 3>(xc_e, xf_e, px, py, pz, smem, t); }
}

// acc (the thread's z-line of the coarse element) += its part of  op_z (x) op_y (x) op_x  applied to one fine element; op* are Nh x NH
// row-major (the prolongation itself for P^T, the transposed projection operator for the L2 projection)
template <int NH, int Nh>
__device__ __forceinline__ void restrict_body(const double* __restrict__ xf_e, const double* Ox, const double* Oy, const double* Oz,
                                              double* lds, double (&acc)[NH], int t) {
  constexpr int RSB = NH | 1;
  double* B = lds;
  double* C = B + Nh * Nh * RSB;
  if (t < Nh * Nh) {
    double x[Nh], y[NH];
#pragma unroll
    for (int a = 0; a < Nh; ++a) x[a] = xf_e[t * Nh + a];
    contract_n<Nh, NH>(Ox, x, y);
#pragma unroll
    for (int a = 0; a < NH; ++a) B[t * RSB + a] = y[a];
  }
  __syncthreads();
  if (t < NH * Nh) {
    const int a = t % NH, k = t / NH;
    double x[Nh], y[NH];
#pragma unroll
    for (int b = 0; b < Nh; ++b) x[b] = B[(k * Nh + b) * RSB + a];
    contract_n<Nh, NH>(Oy, x, y);
#pragma unroll
    for (int b = 0; b < NH; ++b) C[(k * NH + b) * NH + a] = y[b];
  }
  __syncthreads();
  if (t < NH * NH) {
    double x[Nh], y[NH];
#pragma unroll
    for (int k = 0; k < Nh; ++k) x[k] = C[k * NH * NH + t];
    contract_n<Nh, NH>(Oz, x, y);
#pragma unroll
    for (int k = 0; k < NH; ++k) acc[k] += y[k];
  }
}

// blockDim = THREADS * CG: CG groups of threads work on different children of the coarse element at once (a coarse multigrid level has
// few elements: without this an h-restriction runs one wavefront per EIGHT fine elements); group g takes children g, g + CG, ...; the
// partial sums meet in the LDS and are added in group order by group 0 -- a fixed order, so the result is deterministic
template <int NH, int DMAX>
__global__ __launch_bounds__((TransferCfg<NH, DMAX>::THREADS * TransferCfg<NH, DMAX>::CGMAX)) void restrict_fast_kernel(const double* __restrict__ xf, double* __restrict__ xc,
                                                             const int* __restrict__ child, const long long* __restrict__ off,
                                                             const int* __restrict__ item_first, const double* __restrict__ ops,
                                                             const int* __restrict__ list, int CG) {
  using Cfg = TransferCfg<NH, DMAX>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int it = list[blockIdx.x];
  const int c0 = item_first[it], c1 = item_first[it + 1];
  // a group is whole wavefronts (THREADS is a multiple of 64): its index is wave-uniform, and with it the child and its operator tables
  const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / Cfg::THREADS)), t = threadIdx.x % Cfg::THREADS;
  double* lds = smem + (size_t)g * Cfg::LDS_RESTRICT;
  double acc[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) acc[k] = 0.0;
  // every group runs the same number of trips (and so of barriers); a trip without a child only keeps step
  for (int c = c0 + g; c - g < c1; c += CG) {
    if (c < c1) {
      const int* d = child + 8 * c;
      const int dN = d[2] - NH;
      const double* xf_e = xf + off[2 * c + 1];
      const double *ox = ops + d[3], *oy = ops + d[4], *oz = ops + d[5];
      if (dN == 0) restrict_body<NH, NH>(xf_e, ox, oy, oz, lds, acc, t);
      if constexpr (DMAX >= 1) { if (dN == 1) restrict_body<NH, NH + 1>(xf_e, ox, oy, oz, lds, acc, t); }
      if constexpr (DMAX >= 2) { if (dN == 2) restrict_body<NH, NH + 2>(xf_e, ox, oy, oz, lds, acc, t); }
      if constexpr (DMAX >= 3) { if (dN == 3) restrict_body<NH, NH + 3>(xf_e, ox, oy, oz, lds, acc, t); }
    } else {
      __syncthreads();
      __syncthreads();
    }
    __syncthreads();   // the group's image is rewritten by its next child / by the partial sums below
  }
  if (CG > 1) {
    if (t < NH * NH) {
#pragma unroll
      for (int k = 0; k < NH; ++k) lds[k * NH * NH + t] = acc[k];
    }
    __syncthreads();
    if (g == 0 && t < NH * NH) {
      for (int gg = 1; gg < CG; ++gg) {
        const double* o = smem + (size_t)gg * Cfg::LDS_RESTRICT;
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[k] += o[k * NH * NH + t];
      }
    }
  }
  if (g == 0 && t < NH * NH) {
    double* xc_e = xc + off[2 * c0];
#pragma unroll
    for (int k = 0; k < NH; ++k) xc_e[k * NH * NH + t] = acc[k];
  }
}

// ---- the fused Galerkin term: Au_H += sum_c (B_c P_c)^T (w J c)_c (B_c P_c) u_H per coarse element, the coarse-level zeroth-order term of
// the multigrid matrix operator (d4est_hip_mgmatrix.hip) without a fine-level vector: the prolongation to child c and the interpolation to
// its quadrature nodes are ONE composite operator T = B P per direction (NQ x NH, built on the host), so a child costs three passes up,
// the pointwise product with the fine level's w J c (the only stream: 8 B per fine quadrature node, coalesced over the (x, y) plane),
// three passes down; the children's contributions are summed in registers (child groups as in restrict_fast_kernel).
template <int NH, int DMAX>
struct GalerkinCfg {
  static constexpr int NQM = NH + DMAX;
  static constexpr int THREADS = ((NQM * NQM + 63) / 64) * 64;
  static constexpr int LDS = NH * NH * (NQM | 1) + NH * NQM * NQM;
  static constexpr int cg_max() {
    int cg = 8;
    while (cg > 1 && ((long long)cg * LDS * 8 > 150 * 1024 || cg * THREADS > 1024)) cg >>= 1;
    return cg;
  }
  static constexpr int CGMAX = cg_max();
};

template <int NH, int NQ>
__device__ __forceinline__ void galerkin_body(const double* __restrict__ u_e, const double* __restrict__ wjc, const double* TTx, const double* TTy,
                                              const double* TTz, const double* Tx, const double* Ty, const double* Tz, double* lds,
                                              double (&acc)[NH], int t) {
  constexpr int RSB = NQ | 1;
  double* B = lds;                    // [NH][NH][NQ | 1]
  double* C = B + NH * NH * RSB;      // [NH][NQ][NQ]
  if (t < NH * NH) {                  // up, x: the coarse element's x-lines straight from memory
    double x[NH], y[NQ];
#pragma unroll
    for (int a = 0; a < NH; ++a) x[a] = u_e[t * NH + a];
    contract_n<NH, NQ>(TTx, x, y);
#pragma unroll
    for (int a = 0; a < NQ; ++a) B[t * RSB + a] = y[a];
  }
  __syncthreads();
  if (t < NQ * NH) {                  // up, y
    const int aq = t % NQ, k = t / NQ;
    double x[NH], y[NQ];
#pragma unroll
    for (int b = 0; b < NH; ++b) x[b] = B[(k * NH + b) * RSB + aq];
    contract_n<NH, NQ>(TTy, x, y);
#pragma unroll
    for (int b = 0; b < NQ; ++b) C[(k * NQ + b) * NQ + aq] = y[b];
  }
  __syncthreads();
  double w[NQ];
  if (t < NQ * NQ) {                  // up, z; then the coefficient at the thread's quadrature column
    double x[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) x[k] = C[k * NQ * NQ + t];
    contract_n<NH, NQ>(TTz, x, w);
#pragma unroll
    for (int k = 0; k < NQ; ++k) w[k] *= __builtin_nontemporal_load(wjc + k * NQ * NQ + t);
  }
  __syncthreads();
  if (t < NQ * NQ) {                  // down, z
    double y[NH];
    contract_n<NQ, NH>(Tz, w, y);
#pragma unroll
    for (int k = 0; k < NH; ++k) C[k * NQ * NQ + t] = y[k];
  }
  __syncthreads();
  if (t < NQ * NH) {                  // down, y
    const int aq = t % NQ, k = t / NQ;
    double x[NQ], y[NH];
#pragma unroll
    for (int b = 0; b < NQ; ++b) x[b] = C[k * NQ * NQ + b * NQ + aq];
    contract_n<NQ, NH>(Ty, x, y);
#pragma unroll
    for (int b = 0; b < NH; ++b) B[(k * NH + b) * RSB + aq] = y[b];
  }
  __syncthreads();
  if (t < NH * NH) {                  // down, x: into the thread's x-line of the coarse element
    double x[NQ], y[NH];
#pragma unroll
    for (int a = 0; a < NQ; ++a) x[a] = B[t * RSB + a];
    contract_n<NQ, NH>(Tx, x, y);
#pragma unroll
    for (int a = 0; a < NH; ++a) acc[a] += y[a];
  }
}

template <int NH, int DMAX>
__global__ __launch_bounds__((GalerkinCfg<NH, DMAX>::THREADS * GalerkinCfg<NH, DMAX>::CGMAX)) void galerkin_fast_kernel(
    const double* __restrict__ u, double* __restrict__ Au, const double* __restrict__ wjc, const int* __restrict__ child,
    const long long* __restrict__ off, const int* __restrict__ item_first, const int* __restrict__ gchild, const int* __restrict__ gqs,
    const double* __restrict__ T, const double* __restrict__ TT, const int* __restrict__ list, int CG) {
  using Cfg = GalerkinCfg<NH, DMAX>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int it = list[blockIdx.x];
  const int c0 = item_first[it], c1 = item_first[it + 1];
  const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / Cfg::THREADS)), t = threadIdx.x % Cfg::THREADS;
  double* lds = smem + (size_t)g * Cfg::LDS;
  const double* u_e = u + off[2 * c0];
  double acc[NH];
#pragma unroll
  for (int a = 0; a < NH; ++a) acc[a] = 0.0;
  for (int c = c0 + g; c - g < c1; c += CG) {
    if (c < c1) {
      const int* d = gchild + 4 * c;
      const int dQ = d[0] - NH;
      const double* wj = wjc + gqs[c];
      const double *ttx = TT + d[1], *tty = TT + d[2], *ttz = TT + d[3], *tx = T + d[1], *ty = T + d[2], *tz = T + d[3];
      if (dQ == 0) galerkin_body<NH, NH>(u_e, wj, ttx, tty, ttz, tx, ty, tz, lds, acc, t);
      if constexpr (DMAX >= 1) { if (dQ == 1) galerkin_body<NH, NH + 1>(u_e, wj, ttx, tty, ttz, tx, ty, tz, lds, acc, t); }
      if constexpr (DMAX >= 2) { if (dQ == 2) galerkin_body<NH, NH + 2>(u_e, wj, ttx, tty, ttz, tx, ty, tz, lds, acc, t); }
      if constexpr (DMAX >= 3) { if (dQ == 3) galerkin_body<NH, NH + 3>(u_e, wj, ttx, tty, ttz, tx, ty, tz, lds, acc, t); }
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) __syncthreads();
    }
    __syncthreads();
  }
  if (CG > 1) {
    if (t < NH * NH) {
#pragma unroll
      for (int a = 0; a < NH; ++a) lds[t * NH + a] = acc[a];
    }
    __syncthreads();
    if (g == 0 && t < NH * NH) {
      for (int gg = 1; gg < CG; ++gg) {
        const double* o = smem + (size_t)gg * Cfg::LDS;
#pragma unroll
        for (int a = 0; a < NH; ++a) acc[a] += o[t * NH + a];
      }
    }
  }
  if (g == 0 && t < NH * NH) {
    double* Au_e = Au + off[2 * c0];
#pragma unroll
    for (int a = 0; a < NH; ++a) Au_e[t * NH + a] = __dadd_rn(Au_e[t * NH + a], acc[a]);
  }
}

template <typename K>
static void fast_lds_limit(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// launches of one (NH, DMAX) list
template <int NH, int DMAX>
static void go_prolong(d4est_hip_transfer* t, const double* xc, double* xf, const int* list, int n) {
  using C = TransferCfg<NH, DMAX>;
  const size_t lds = (size_t)C::LDS_PROLONG * sizeof(double);
  fast_lds_limit(prolong_fast_kernel<NH, DMAX>, lds);
  hipLaunchKernelGGL((prolong_fast_kernel<NH, DMAX>), dim3(n), dim3(C::THREADS), lds, t->stream, xc, xf, t->d_child, t->d_off, t->d_opsT, list);
}
template <int NH, int DMAX>
static void go_restrict(d4est_hip_transfer* t, const double* xf, double* xc, const double* ops, const int* list, int n, int n_children) {
  using C = TransferCfg<NH, DMAX>;
  // child groups: as many of the item's children at once as the LDS (and 1024 threads) hold
  // child groups where the list alone does not fill the chip (coarse levels); with thousands of coarse elements one group per element
  // keeps more elements in flight (level 5, p = 3: 61 us against 113 us with groups)
  const int cg = (n_children == 8 && n < 8192) ? C::CGMAX : 1;
  const size_t lds = (size_t)cg * C::LDS_RESTRICT * sizeof(double);
  fast_lds_limit(restrict_fast_kernel<NH, DMAX>, lds);
  hipLaunchKernelGGL((restrict_fast_kernel<NH, DMAX>), dim3(n), dim3(C::THREADS * cg), lds, t->stream, xf, xc, t->d_child, t->d_off,
                     t->d_item_first, ops, list, cg);
}

#define D4EST_HIP_TRANSFER_NH(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

static void launch_fast_prolong(d4est_hip_transfer* t, const double* xc, double* xf, int NH, int dmax, const int* list, int n) {
#define X(N_)                                                         \
  if (NH == N_) {                                                     \
    if (dmax == 0) go_prolong<N_, 0>(t, xc, xf, list, n);             \
    else if (dmax == 1) go_prolong<N_, 1>(t, xc, xf, list, n);        \
    else go_prolong<N_, 3>(t, xc, xf, list, n);                       \
    return;                                                           \
  }
  D4EST_HIP_TRANSFER_NH(X)
#undef X
  D4EST_HIP_ABORT("transfer: no fast prolongation kernel for %d coarse nodes per direction", NH);
}
static void launch_fast_restrict(d4est_hip_transfer* t, const double* xf, double* xc, const double* ops, int NH, int dmax, const int* list, int n, int nc) {
#define X(N_)                                                         \
  if (NH == N_) {                                                     \
    if (dmax == 0) go_restrict<N_, 0>(t, xf, xc, ops, list, n, nc);       \
    else if (dmax == 1) go_restrict<N_, 1>(t, xf, xc, ops, list, n, nc);  \
    else go_restrict<N_, 3>(t, xf, xc, ops, list, n, nc);                 \
    return;                                                           \
  }
  D4EST_HIP_TRANSFER_NH(X)
#undef X
  D4EST_HIP_ABORT("transfer: no fast restriction kernel for %d coarse nodes per direction", NH);
}


template <int NH, int DMAX>
static void go_galerkin(d4est_hip_transfer* t, const double* u, double* Au, const double* wjc, const int* list, int n, int n_children, hipStream_t st) {
  using C = GalerkinCfg<NH, DMAX>;
  const int cg = (n_children == 8 && n < 8192) ? C::CGMAX : 1;
  const size_t lds = (size_t)cg * C::LDS * sizeof(double);
  fast_lds_limit(galerkin_fast_kernel<NH, DMAX>, lds);
  hipLaunchKernelGGL((galerkin_fast_kernel<NH, DMAX>), dim3(n), dim3(C::THREADS * cg), lds, st, u, Au, wjc, t->d_child, t->d_off, t->d_item_first,
                     t->d_gal_child, t->d_gal_qs, t->d_gal_T, t->d_gal_TT, list, cg);
}

// true: the fused kernel serves this transfer with this fine plan (every item within the compile-time sizes); tables and lists are built
bool galerkin_fused_setup(d4est_hip_transfer* t, d4est_hip_plan* fine) {
  if (t->gal_fine == fine && t->d_gal_T) return !t->gal_lists.empty() || t->n_items == 0;
  (void)hipFree(t->d_gal_T); (void)hipFree(t->d_gal_TT); (void)hipFree(t->d_gal_child); (void)hipFree(t->d_gal_qs); (void)hipFree(t->d_gal_lists);
  t->d_gal_T = t->d_gal_TT = nullptr; t->d_gal_child = t->d_gal_qs = t->d_gal_lists = nullptr;
  t->gal_lists.clear();
  t->gal_fine = fine;
  if (std::getenv("D4EST_HIP_TRANSFER_GENERIC") || fine->n_elements != t->n_children || fine->quad_aliased) return false;
  std::vector<double> T, TT;
  std::map<std::tuple<int, int, int, int>, int> index;   // (hp, degH, degh, deg_quad) -> offset of the (two) composite operators
  auto get = [&](int hp, int dH, int dh, int dq) {
    auto key = std::make_tuple(hp, dH, dh, dq);
    auto f = index.find(key);
    if (f != index.end()) return f->second;
    const std::vector<double> B = Tables1D::quad_interp(fine->quad_type, dh, dq);   // (dq+1) x (dh+1)
    const std::vector<double> P = hp ? Tables1D::hp_prolong(dH, dh) : Tables1D::p_prolong(dH, dh);
    const int o = (int)T.size();
    for (int h = 0; h < (hp ? 2 : 1); ++h) {
      const std::vector<double> Ph(P.begin() + (size_t)h * (dh + 1) * (dH + 1), P.begin() + (size_t)(h + 1) * (dh + 1) * (dH + 1));
      const std::vector<double> C = Tables1D::matmul(B, Ph, dq + 1, dh + 1, dH + 1);   // (dq+1) x (dH+1)
      const std::vector<double> Ct = Tables1D::transpose(C, dq + 1, dH + 1);
      T.insert(T.end(), C.begin(), C.end());
      TT.insert(TT.end(), Ct.begin(), Ct.end());
    }
    index[key] = o;
    return o;
  };
  std::vector<int> gchild((size_t)4 * std::max(t->n_children, 1), 0), gqs((size_t)std::max(t->n_children, 1), 0);
  std::map<std::pair<int, int>, std::vector<int>> lists;
  std::map<std::pair<int, int>, int> dmax_of;
  long long fo = 0;
  int rec = 0;
  for (int it = 0; it < t->n_items; ++it) {
    const int nc = t->h_hrefine[it] == 1 ? 8 : 1, dH = t->h_degH[it], NH = dH + 1;
    int dmax = 0;
    for (int c = 0; c < nc; ++c, ++rec) {
      const int dh = t->h_degh[8 * (size_t)it + c], dq = fine->deg_quad[rec];
      if (fine->deg[rec] != dh || fine->nodal_stride[rec] != fo) return false;   // the fine plan is not this transfer's fine level
      fo += (long long)(dh + 1) * (dh + 1) * (dh + 1);
      const int base = get(nc == 8, dH, dh, dq);
      const int half = (dq + 1) * NH;
      gchild[4 * (size_t)rec + 0] = dq + 1;
      gchild[4 * (size_t)rec + 1] = base + (nc == 8 ? (c & 1) * half : 0);
      gchild[4 * (size_t)rec + 2] = base + (nc == 8 ? ((c >> 1) & 1) * half : 0);
      gchild[4 * (size_t)rec + 3] = base + (nc == 8 ? ((c >> 2) & 1) * half : 0);
      gqs[rec] = fine->quad_stride[rec];
      dmax = std::max(dmax, dq + 1 - NH);
      if (dq + 1 < NH) return false;
    }
    if (NH < 2 || NH > kFastMaxNH || dmax > kFastMaxD) return false;
    const auto key = std::make_pair(NH, nc);
    lists[key].push_back(it);
    dmax_of[key] = std::max(dmax_of[key], dmax);
  }
  std::vector<int> all;
  for (auto& kv : lists) {
    t->gal_lists.push_back({kv.first.first, dmax_of[kv.first], (int)all.size(), (int)kv.second.size(), kv.first.second});
    all.insert(all.end(), kv.second.begin(), kv.second.end());
  }
  auto up_i = [](const std::vector<int>& v) { int* d = nullptr; HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(int))); if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice)); return d; };
  auto up_d = [](const std::vector<double>& v) { double* d = nullptr; HIP_CHECK(hipMalloc(&d, (v.size() + 16) * sizeof(double))); HIP_CHECK(hipMemset(d, 0, (v.size() + 16) * sizeof(double))); if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice)); return d; };
  t->d_gal_T = up_d(T); t->d_gal_TT = up_d(TT);
  t->d_gal_child = up_i(gchild); t->d_gal_qs = up_i(gqs); t->d_gal_lists = up_i(all);
  return true;
}

void galerkin_fused_apply(d4est_hip_transfer* t, const double* wjc, const double* u, double* Au, hipStream_t st) {
  for (const d4est_hip_transfer::List& L : t->gal_lists) {
    const int* list = t->d_gal_lists + L.first;
    bool done = false;
#define X(N_)                                                                             \
  if (!done && L.NH == N_) {                                                              \
    if (L.dmax == 0) go_galerkin<N_, 0>(t, u, Au, wjc, list, L.n, L.nc, st);              \
    else if (L.dmax == 1) go_galerkin<N_, 1>(t, u, Au, wjc, list, L.n, L.nc, st);         \
    else go_galerkin<N_, 3>(t, u, Au, wjc, list, L.n, L.nc, st);                          \
    done = true;                                                                          \
  }
    D4EST_HIP_TRANSFER_NH(X)
#undef X
    if (!done) D4EST_HIP_ABORT("fused Galerkin term: no kernel for %d coarse nodes per direction", L.NH);
  }
  HIP_CHECK(hipGetLastError());
}

}  // namespace d4est_hip

using d4est_hip::Tables1D;

extern "C" {

d4est_hip_transfer_t* d4est_hip_transfer_create(int n_items, const int* hrefine, const int* degH, const int* degh) {
  if (n_items < 0 || (n_items > 0 && (!hrefine || !degH || !degh))) D4EST_HIP_ABORT("transfer_create: bad arguments");
  d4est_hip_transfer* t = new d4est_hip_transfer();
  t->n_items = n_items;
  std::vector<double> ops, rops, opsT, ropsT;   // P (Nh x NH), R (NH x Nh) and their transposes at the same offsets
  std::map<std::pair<int, int>, int> p_index, hp_index;
  auto get_p = [&](int dH, int dh) {
    auto key = std::make_pair(dH, dh);
    auto it = p_index.find(key);
    if (it != p_index.end()) return it->second;
    std::vector<double> P = Tables1D::p_prolong(dH, dh);   // identity when dH == dh (d4est_operators.c:1114-1118 copies)
    const int o = (int)ops.size();
    ops.insert(ops.end(), P.begin(), P.end());
    std::vector<double> R = Tables1D::p_restrict(dH, dh);  // (dH+1) x (dh+1), d4est_operators.c:1165-1185
    rops.insert(rops.end(), R.begin(), R.end());
    const std::vector<double> PT = Tables1D::transpose(P, dh + 1, dH + 1), RT = Tables1D::transpose(R, dH + 1, dh + 1);
    opsT.insert(opsT.end(), PT.begin(), PT.end());
    ropsT.insert(ropsT.end(), RT.begin(), RT.end());
    p_index[key] = o;
    return o;
  };
  auto get_hp = [&](int dH, int dh) {
    auto key = std::make_pair(dH, dh);
    auto it = hp_index.find(key);
    if (it != hp_index.end()) return it->second;
    std::vector<double> P = Tables1D::hp_prolong(dH, dh);  // 2 x (dh+1) x (dH+1)
    const int o = (int)ops.size();
    ops.insert(ops.end(), P.begin(), P.end());
    std::vector<double> R = Tables1D::hp_restrict(dH, dh); // 2 x (dH+1) x (dh+1), d4est_operators.c:1232-1259
    rops.insert(rops.end(), R.begin(), R.end());
    const size_t half = (size_t)(dh + 1) * (dH + 1);
    for (int h = 0; h < 2; ++h) {   // each half transposed on its own: the offsets of the halves stay
      const std::vector<double> PT = Tables1D::transpose(std::vector<double>(P.begin() + h * half, P.begin() + (h + 1) * half), dh + 1, dH + 1);
      const std::vector<double> RT = Tables1D::transpose(std::vector<double>(R.begin() + h * half, R.begin() + (h + 1) * half), dH + 1, dh + 1);
      opsT.insert(opsT.end(), PT.begin(), PT.end());
      ropsT.insert(ropsT.end(), RT.begin(), RT.end());
    }
    hp_index[key] = o;
    return o;
  };
  std::vector<int> child, item_first(n_items + 1, 0);
  std::vector<long long> off, moff, coff;
  long long co = 0, fo = 0, fm = 0, cm = 0, wk = 0;
  t->h_hrefine.assign(hrefine, hrefine + n_items);
  t->h_degH.assign(degH, degH + n_items);
  t->h_degh.assign(degh, degh + 8 * (size_t)n_items);
  for (int it = 0; it < n_items; ++it) {
    item_first[it] = (int)(child.size() / 8);
    const int dH = degH[it];
    const int nc = hrefine[it] == 1 ? 8 : 1;
    if (hrefine[it] != 0 && hrefine[it] != 1) D4EST_HIP_ABORT("transfer_create: item %d has hrefine %d (0: p only, 1: eight children)", it, hrefine[it]);
    if (dH < 1 || dH > Tables1D::kMaxDeg) D4EST_HIP_ABORT("transfer_create: item %d has degH %d", it, dH);
    for (int c = 0; c < nc; ++c) {
      const int dh = degh[8 * it + c];
      if (dh < dH || dh > Tables1D::kMaxDeg) D4EST_HIP_ABORT("transfer_create: item %d child %d has degh %d < degH %d (the reference asserts degH <= degh, d4est_operators.c:379)", it, c, dh, dH);
      const int nH = dH + 1, nh = dh + 1;
      int ox, oy, oz;
      if (nc == 1) ox = oy = oz = get_p(dH, dh);
      else {
        const int base = get_hp(dH, dh);
        ox = base + (c & 1) * nh * nH;          // d4est_operators.c:394-404: child c = (cx, cy, cz) bits
        oy = base + ((c >> 1) & 1) * nh * nH;
        oz = base + ((c >> 2) & 1) * nh * nH;
      }
      const int rec[8] = {it, nH, nh, ox, oy, oz, c == 0, nc};
      child.insert(child.end(), rec, rec + 8);
      off.push_back(co);
      off.push_back(fo);
      // dense element blocks of the multigrid matrix operator: consecutive, (deg+1)^3 x (deg+1)^3 each, in the same traversal order
      // (fine_matrix_stride / coarse_matrix_stride of Solver/d4est_solver_multigrid_matrix_operator.c:19-46)
      const long long nh3 = (long long)nh * nh * nh, nH3 = (long long)nH * nH * nH;
      moff.push_back(fm);
      moff.push_back(wk);
      fm += nh3 * nh3;
      wk += nh3 * nH3;
      fo += (long long)nh * nh * nh;
      t->max_n = std::max(t->max_n, std::max(nH, nh));
    }
    coff.push_back(cm);
    cm += (long long)(dH + 1) * (dH + 1) * (dH + 1) * (dH + 1) * (dH + 1) * (dH + 1);
    co += (long long)(dH + 1) * (dH + 1) * (dH + 1);
  }
  t->fine_matrix_nodes = fm;
  t->coarse_matrix_nodes = cm;
  t->work_doubles = wk;
  item_first[n_items] = (int)(child.size() / 8);
  // the kernels keep two fields of max_n^3 doubles in the LDS (the restriction / projection a third one for the sum over the children
  // up to p = 17; above that the sum lives in the output vector): the reference's degree range p <= 19 (d4est_operators.c:1205-1297) fits.
  // Anything larger is refused here, once, with the reason -- not at the first restrict call in the middle of a V-cycle.
  if ((size_t)2 * t->max_n * t->max_n * t->max_n * sizeof(double) > 160 * 1024)
    D4EST_HIP_ABORT("transfer_create: degree %d exceeds the transfer kernels' limit (two %d^3 fields do not fit the 160 KB LDS)",
                    t->max_n - 1, t->max_n);
  t->n_children = item_first[n_items];
  t->coarse_nodes = co;
  t->fine_nodes = fo;
  auto up_i = [](const std::vector<int>& v) { int* d = nullptr; HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(int))); if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice)); return d; };
  t->d_child = up_i(child);
  t->d_item_first = up_i(item_first);
  auto up_l = [](const std::vector<long long>& v) { long long* d = nullptr; HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(long long))); if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(long long), hipMemcpyHostToDevice)); return d; };
  t->d_off = up_l(off);
  t->d_moff = up_l(moff);
  t->d_coff = up_l(coff);
  HIP_CHECK(hipMalloc(&t->d_ops, std::max<size_t>(ops.size(), 1) * sizeof(double)));
  if (!ops.empty()) HIP_CHECK(hipMemcpy(t->d_ops, ops.data(), ops.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_CHECK(hipMalloc(&t->d_rops, std::max<size_t>(rops.size(), 1) * sizeof(double)));
  if (!rops.empty()) HIP_CHECK(hipMemcpy(t->d_rops, rops.data(), rops.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_CHECK(hipMalloc(&t->d_opsT, std::max<size_t>(opsT.size(), 1) * sizeof(double)));
  if (!opsT.empty()) HIP_CHECK(hipMemcpy(t->d_opsT, opsT.data(), opsT.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_CHECK(hipMalloc(&t->d_ropsT, std::max<size_t>(ropsT.size(), 1) * sizeof(double)));
  if (!ropsT.empty()) HIP_CHECK(hipMemcpy(t->d_ropsT, ropsT.data(), ropsT.size() * sizeof(double), hipMemcpyHostToDevice));
  // work lists: fine elements / coarse elements by coarse size NH for the compile-time kernels (NH <= 16, every fine size of the item
  // within NH .. NH + 3), everything else through the generic runtime-size kernels
  {
    const bool no_fast = std::getenv("D4EST_HIP_TRANSFER_GENERIC") != nullptr;
    std::map<int, std::vector<int>> pl;
    std::map<std::pair<int, int>, std::vector<int>> rl;   // restriction: coarse elements with eight children (worked on by child groups) apart
    std::map<int, int> pd;
    std::map<std::pair<int, int>, int> rd;
    std::vector<int> pg, rg;
    for (int it = 0; it < n_items; ++it) {
      const int c0 = item_first[it], c1 = item_first[it + 1];
      const int NH = child[8 * c0 + 1];
      int dmax = 0;
      for (int c = c0; c < c1; ++c) dmax = std::max(dmax, child[8 * c + 2] - NH);
      const bool fast = !no_fast && NH >= 2 && NH <= d4est_hip::kFastMaxNH && dmax <= d4est_hip::kFastMaxD;
      if (fast) {
        const auto rk = std::make_pair(NH, c1 - c0);
        rl[rk].push_back(it);
        rd[rk] = std::max(rd[rk], dmax);
        for (int c = c0; c < c1; ++c) pl[NH].push_back(c);
        pd[NH] = std::max(pd[NH], dmax);
      } else {
        rg.push_back(it);
        for (int c = c0; c < c1; ++c) pg.push_back(c);
      }
    }
    std::vector<int> all;
    auto add = [&](std::vector<d4est_hip_transfer::List>& dst, int NH, int dmax, const std::vector<int>& v, int nc = 1) {
      if (v.empty()) return;
      dst.push_back({NH, dmax, (int)all.size(), (int)v.size(), nc});
      all.insert(all.end(), v.begin(), v.end());
    };
    for (auto& kv : pl) add(t->prolong_lists, kv.first, pd[kv.first], kv.second);
    add(t->prolong_lists, 0, 0, pg);
    for (auto& kv : rl) add(t->restrict_lists, kv.first.first, rd[kv.first], kv.second, kv.first.second);
    add(t->restrict_lists, 0, 0, rg);
    t->d_lists = up_i(all);
  }
  return t;
}

void d4est_hip_transfer_destroy(d4est_hip_transfer_t* t) {
  if (!t) return;
  (void)hipFree(t->d_child); (void)hipFree(t->d_off); (void)hipFree(t->d_item_first); (void)hipFree(t->d_ops); (void)hipFree(t->d_rops);
  (void)hipFree(t->d_opsT); (void)hipFree(t->d_ropsT); (void)hipFree(t->d_lists);
  (void)hipFree(t->d_gal_T); (void)hipFree(t->d_gal_TT); (void)hipFree(t->d_gal_child); (void)hipFree(t->d_gal_qs); (void)hipFree(t->d_gal_lists);
  (void)hipFree(t->d_moff); (void)hipFree(t->d_coff); (void)hipFree(t->d_work); (void)hipFree(t->d_window); (void)hipFree(t->d_woff);
  delete t;
}

void d4est_hip_transfer_set_stream(d4est_hip_transfer_t* t, void* hip_stream) {
  if (!t) D4EST_HIP_ABORT("transfer_set_stream: NULL transfer");
  t->stream = (hipStream_t)hip_stream;
}

long long d4est_hip_transfer_coarse_nodes(const d4est_hip_transfer_t* t) { return t ? t->coarse_nodes : -1; }
long long d4est_hip_transfer_fine_nodes(const d4est_hip_transfer_t* t) { return t ? t->fine_nodes : -1; }

void d4est_hip_transfer_prolong(d4est_hip_transfer_t* t, const double* x_coarse_dev, double* x_fine_dev) {
  if (!t) D4EST_HIP_ABORT("transfer_prolong: NULL transfer");
  if (t->n_children == 0) return;
  for (const d4est_hip_transfer::List& L : t->prolong_lists) {
    if (L.NH > 0) { d4est_hip::launch_fast_prolong(t, x_coarse_dev, x_fine_dev, L.NH, L.dmax, t->d_lists + L.first, L.n); continue; }
    const int n3 = t->max_n * t->max_n * t->max_n;
    const size_t lds = (size_t)2 * n3 * sizeof(double);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(d4est_hip::prolong_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(d4est_hip::prolong_kernel, dim3(std::min(L.n, 65536)), dim3(256), lds, t->stream, x_coarse_dev, x_fine_dev,
                       t->d_child, t->d_off, t->d_ops, t->d_lists + L.first, L.n, n3);
  }
  HIP_CHECK(hipGetLastError());
}

static void launch_restrict(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev, bool project, const char* who) {
  if (!t) D4EST_HIP_ABORT("%s: NULL transfer", who);
  if (t->n_items == 0) return;
  for (const d4est_hip_transfer::List& L : t->restrict_lists) {
    if (L.NH > 0) {   // P^T: the prolongation itself is the (Nh x NH) operand; the projection: its operator transposed
      d4est_hip::launch_fast_restrict(t, x_fine_dev, x_coarse_dev, project ? t->d_ropsT : t->d_ops, L.NH, L.dmax, t->d_lists + L.first, L.n, L.nc);
      continue;
    }
    const int n3 = t->max_n * t->max_n * t->max_n;
    const int acc_in_lds = ((size_t)3 * n3 * sizeof(double) <= 160 * 1024) ? 1 : 0;
    const size_t lds = (size_t)(acc_in_lds ? 3 : 2) * n3 * sizeof(double);
    if (lds > 160 * 1024) D4EST_HIP_ABORT("%s: degree too high for the LDS-resident kernel", who);
    const void* fn = project ? reinterpret_cast<const void*>(d4est_hip::restrict_kernel<false>) : reinterpret_cast<const void*>(d4est_hip::restrict_kernel<true>);
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (project)
      hipLaunchKernelGGL(d4est_hip::restrict_kernel<false>, dim3(std::min(L.n, 65536)), dim3(256), lds, t->stream, x_fine_dev, x_coarse_dev,
                         t->d_child, t->d_off, t->d_item_first, t->d_rops, t->d_lists + L.first, L.n, n3, acc_in_lds);
    else
      hipLaunchKernelGGL(d4est_hip::restrict_kernel<true>, dim3(std::min(L.n, 65536)), dim3(256), lds, t->stream, x_fine_dev, x_coarse_dev,
                         t->d_child, t->d_off, t->d_item_first, t->d_ops, t->d_lists + L.first, L.n, n3, acc_in_lds);
  }
  HIP_CHECK(hipGetLastError());
}

void d4est_hip_transfer_restrict(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev) {
  launch_restrict(t, x_fine_dev, x_coarse_dev, false, "transfer_restrict");
}

void d4est_hip_transfer_project(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev) {
  launch_restrict(t, x_fine_dev, x_coarse_dev, true, "transfer_project");
}

}  // extern "C"
