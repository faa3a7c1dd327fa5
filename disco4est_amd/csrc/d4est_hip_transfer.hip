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

namespace d4est_hip {

// one workgroup per fine element
__global__ __launch_bounds__(256) void prolong_kernel(const double* __restrict__ xc, double* __restrict__ xf,
                                                      const int* __restrict__ child, const long long* __restrict__ off,
                                                      const double* __restrict__ ops, int n_children, int max_n3) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* a = smem;
  double* b = smem + max_n3;
  for (int c = blockIdx.x; c < n_children; c += gridDim.x) {
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
                                                       int n_items, int max_n3, int acc_in_lds) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* a = smem;
  double* b = smem + max_n3;
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
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

}  // namespace d4est_hip

using d4est_hip::Tables1D;

extern "C" {

d4est_hip_transfer_t* d4est_hip_transfer_create(int n_items, const int* hrefine, const int* degH, const int* degh) {
  if (n_items < 0 || (n_items > 0 && (!hrefine || !degH || !degh))) D4EST_HIP_ABORT("transfer_create: bad arguments");
  d4est_hip_transfer* t = new d4est_hip_transfer();
  t->n_items = n_items;
  std::vector<double> ops, rops;
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
  return t;
}

void d4est_hip_transfer_destroy(d4est_hip_transfer_t* t) {
  if (!t) return;
  (void)hipFree(t->d_child); (void)hipFree(t->d_off); (void)hipFree(t->d_item_first); (void)hipFree(t->d_ops); (void)hipFree(t->d_rops);
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
  const int n3 = t->max_n * t->max_n * t->max_n;
  const size_t lds = (size_t)2 * n3 * sizeof(double);
  if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(d4est_hip::prolong_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(d4est_hip::prolong_kernel, dim3(std::min(t->n_children, 65536)), dim3(256), lds, t->stream, x_coarse_dev, x_fine_dev,
                     t->d_child, t->d_off, t->d_ops, t->n_children, n3);
  HIP_CHECK(hipGetLastError());
}

static void launch_restrict(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev, bool project, const char* who) {
  if (!t) D4EST_HIP_ABORT("%s: NULL transfer", who);
  if (t->n_items == 0) return;
  const int n3 = t->max_n * t->max_n * t->max_n;
  const int acc_in_lds = ((size_t)3 * n3 * sizeof(double) <= 160 * 1024) ? 1 : 0;
  const size_t lds = (size_t)(acc_in_lds ? 3 : 2) * n3 * sizeof(double);
  if (lds > 160 * 1024) D4EST_HIP_ABORT("%s: degree too high for the LDS-resident kernel", who);
  const void* fn = project ? reinterpret_cast<const void*>(d4est_hip::restrict_kernel<false>) : reinterpret_cast<const void*>(d4est_hip::restrict_kernel<true>);
  if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (project)
    hipLaunchKernelGGL(d4est_hip::restrict_kernel<false>, dim3(std::min(t->n_items, 65536)), dim3(256), lds, t->stream, x_fine_dev, x_coarse_dev,
                       t->d_child, t->d_off, t->d_item_first, t->d_rops, t->n_items, n3, acc_in_lds);
  else
    hipLaunchKernelGGL(d4est_hip::restrict_kernel<true>, dim3(std::min(t->n_items, 65536)), dim3(256), lds, t->stream, x_fine_dev, x_coarse_dev,
                       t->d_child, t->d_off, t->d_item_first, t->d_ops, t->n_items, n3, acc_in_lds);
  HIP_CHECK(hipGetLastError());
}

void d4est_hip_transfer_restrict(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev) {
  launch_restrict(t, x_fine_dev, x_coarse_dev, false, "transfer_restrict");
}

void d4est_hip_transfer_project(d4est_hip_transfer_t* t, const double* x_fine_dev, double* x_coarse_dev) {
  launch_restrict(t, x_fine_dev, x_coarse_dev, true, "transfer_project");
}

}  // extern "C"
