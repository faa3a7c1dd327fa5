// Shared by d4est_hip_transfer.hip and d4est_hip_mgmatrix.hip: the transfer object and the generic three-pass tensor contraction.
#pragma once
#include "d4est_hip_internal.h"

struct d4est_hip_transfer {
  int n_items = 0, n_children = 0;
  long long coarse_nodes = 0, fine_nodes = 0;
  int max_n = 1;
  int* d_child = nullptr;      // per (item, child): 8 ints {item, NH, Nh, off_x, off_y, off_z, first child of item?, n children of item}
  long long* d_off = nullptr;  // per (item, child): {coarse offset, fine offset}
  int* d_item_first = nullptr; // per item: index of its first child record (n_items + 1)
  double* d_ops = nullptr;
  double* d_rops = nullptr;    // the L2-projection operators (p_restrict / hp_restrict) at the same offsets as the prolongations
  double* d_opsT = nullptr;    // every prolongation transposed (NH x Nh), same offsets: the operand form of the compile-time kernels
  double* d_ropsT = nullptr;   // every projection operator transposed (Nh x NH)
  // work lists (d4est_hip_transfer.hip): fine elements (prolongation) / coarse elements (restriction) by coarse size NH for the
  // compile-time kernels, dmax = the largest Nh - NH in the list; NH = 0: the generic runtime-size kernels
  struct List { int NH, dmax, first, n, nc; };   // nc: children per coarse element of a restriction list (1 or 8)
  std::vector<List> prolong_lists, restrict_lists;
  int* d_lists = nullptr;
  // fused Galerkin term (galerkin_fused_*, d4est_hip_transfer.hip): coarse u -> [prolong, interpolate to the fine quadrature nodes] as ONE
  // composite 1-D operator per direction, times w J c of the fine level, and back -- the multigrid matrix operator's coarse term in one
  // kernel that streams only the fine coefficient.  Built for one fine plan at a time (gal_fine).
  const void* gal_fine = nullptr;
  double* d_gal_T = nullptr;     // composite operators T = B P (NQ x NH) ...
  double* d_gal_TT = nullptr;    // ... and their transposes (NH x NQ), same offsets
  int* d_gal_child = nullptr;    // per (item, child): {NQ, off_x, off_y, off_z}
  int* d_gal_qs = nullptr;       // per (item, child): offset of the fine element's quadrature block
  int* d_gal_lists = nullptr;
  std::vector<List> gal_lists;   // coarse elements by (NH, largest NQ - NH, children)
  hipStream_t stream = nullptr;
  // multigrid matrix operator (d4est_hip_mgmatrix.hip): where every child's / item's dense block sits (doubles), and the workspace of
  // the triple product  sum_c P_c^T M_c P_c  (T_c = M_c P_c, allocated on first use)
  long long* d_moff = nullptr;   // per (item, child): {offset of the child's block in the fine matrix, offset of T_c in the workspace}
  long long* d_coff = nullptr;   // per item: offset of its block in the coarse matrix
  long long fine_matrix_nodes = 0, coarse_matrix_nodes = 0, work_doubles = 0;
  double* d_work = nullptr;
  std::vector<int> h_hrefine, h_degH, h_degh;   // the item list as handed over (literal-window left factors are built from it)
  double* d_window = nullptr;                   // literal mode: the dense left factors, per (item, child) at d_woff
  long long* d_woff = nullptr;
};

namespace d4est_hip {

// out (n_out^3) = (Az (x) Ay (x) Ax) in (n_in^3), A* given as n_out x n_in (TRANS = false) or applied transposed
// (A* is n_in x n_out, TRANS = true).  in/out/tmp are LDS arrays of >= max(n_in, n_out)^3 doubles.
template <bool TRANS>
__device__ inline void tensor3(const double* __restrict__ Ax, const double* __restrict__ Ay, const double* __restrict__ Az, int n_in,
                               int n_out, double* a, double* b) {
  // x: a [n_in z][n_in y][n_in x] -> b [z][y][n_out]
  for (int idx = threadIdx.x; idx < n_in * n_in * n_out; idx += blockDim.x) {
    const int o = idx % n_out, r = idx / n_out;
    double s = 0.0;
    for (int i = 0; i < n_in; ++i) s = fma(TRANS ? Ax[i * n_out + o] : Ax[o * n_in + i], a[r * n_in + i], s);
    b[r * n_out + o] = s;
  }
  __syncthreads();
  // y: b [z][n_in y][n_out x] -> a [z][n_out][n_out]
  for (int idx = threadIdx.x; idx < n_in * n_out * n_out; idx += blockDim.x) {
    const int x = idx % n_out, o = (idx / n_out) % n_out, z = idx / (n_out * n_out);
    double s = 0.0;
    for (int i = 0; i < n_in; ++i) s = fma(TRANS ? Ay[i * n_out + o] : Ay[o * n_in + i], b[(z * n_in + i) * n_out + x], s);
    a[(z * n_out + o) * n_out + x] = s;
  }
  __syncthreads();
  // z: a [n_in z][n_out][n_out] -> b [n_out][n_out][n_out]
  for (int idx = threadIdx.x; idx < n_out * n_out * n_out; idx += blockDim.x) {
    const int xy = idx % (n_out * n_out), o = idx / (n_out * n_out);
    double s = 0.0;
    for (int i = 0; i < n_in; ++i) s = fma(TRANS ? Az[i * n_out + o] : Az[o * n_in + i], a[i * n_out * n_out + xy], s);
    b[idx] = s;
  }
  __syncthreads();
}

// the multigrid matrix operator's coarse term in one kernel (one transfer between the plan's level and the fine plan's)
bool galerkin_fused_setup(d4est_hip_transfer* t, d4est_hip_plan* fine);
void galerkin_fused_apply(d4est_hip_transfer* t, const double* wjc, const double* u, double* Au, hipStream_t st);

}  // namespace d4est_hip
