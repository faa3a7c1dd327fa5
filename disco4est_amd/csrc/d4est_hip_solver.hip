// Device-resident smoother inner loops: Chebyshev iteration and the CG-Lanczos spectral bound.
//
// Replaces d4est_solver_multigrid_smoother_cheby_iterate_aux (src/Solver/d4est_solver_multigrid_smoother_cheby.c:81-176)
// and cg_eigs (src/Solver/d4est_solver_cg_eigs.c:116-275).  The reference runs 5 BLAS-1 sweeps per Chebyshev
// iteration and 3 blocking allreduces per CG iteration on the host; here the Chebyshev update is ONE fused sweep
// (r, p, u updated together: 4 reads + 3 writes per node), the CG scalars (alpha, beta, the dots) never leave the
// device, and the only host synchronisation is the final read-back of the (alpha_i, beta_i) history from which the
// Gershgorin bound is formed.  Multi-rank runs hook an allreduce (RCCL) and a face-trace exchange through callbacks.
#include <algorithm>
#include <cmath>

#include "d4est_hip_internal.h"

namespace d4est_hip {

constexpr int kRedBlocks = 1024;

// r = alpha (rhs - Au); p = r + beta p; u += p   -- the reference's copy/xpby/scale/xpby/axpy sequence
// (smoother_cheby.c:135-153), with each product/sum rounded separately like the BLAS-1 calls.
__global__ __launch_bounds__(256) void cheby_update_kernel(int n, const double* __restrict__ rhs, const double* __restrict__ Au,
                                                           double alpha, double beta, double* __restrict__ r,
                                                           double* __restrict__ p, double* __restrict__ u) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double res = __dadd_rn(rhs[i], __dmul_rn(-1.0, Au[i]));
    const double ri = __dmul_rn(alpha, res);
    const double pi = __dadd_rn(__dmul_rn(beta, p[i]), ri);
    if (r) r[i] = ri;
    p[i] = pi;
    u[i] = __dadd_rn(u[i], pi);
  }
}

__global__ __launch_bounds__(256) void residual_kernel(int n, const double* __restrict__ rhs, const double* __restrict__ Au,
                                                       double* __restrict__ r, double* __restrict__ copy) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = __dadd_rn(rhs[i], __dmul_rn(-1.0, Au[i]));
    r[i] = v;
    if (copy) copy[i] = v;
  }
}

// r = rhs - r, in place (the Schwarz smoother keeps A u in the residual vector): no aliased __restrict__ pair
__global__ __launch_bounds__(256) void residual_inplace_kernel(int n, const double* __restrict__ rhs, double* __restrict__ r) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) r[i] = __dadd_rn(rhs[i], __dmul_rn(-1.0, r[i]));
}

__global__ __launch_bounds__(256) void add_kernel(int n, const double* __restrict__ x, double* __restrict__ y) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = __dadd_rn(y[i], x[i]);
}

// deterministic two-stage dot product: fixed grid, fixed summation tree
__global__ __launch_bounds__(256) void dot_partial_kernel(int n, const double* __restrict__ x, const double* __restrict__ y,
                                                          double* __restrict__ partial) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s = fma(x[i], y[i], s);
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void dot_final_kernel(int nblocks, const double* __restrict__ partial, double* __restrict__ out) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

// scal: [0] delta_new, [1] delta_old, [2] d.Au, [3] alpha, [4] beta ; hist: alpha_i at [i], beta_i at [imax + i]
__global__ void cg_alpha_kernel(double* scal, double* hist, int i) {
  const double a = scal[0] / scal[2];
  scal[3] = a;
  hist[i] = a;
  scal[1] = scal[0];  // delta_old = delta_new (cg_eigs.c:206)
}
__global__ void cg_beta_kernel(double* scal, double* hist, int i, int imax) {
  const double b = scal[0] / scal[1];
  scal[4] = b;
  hist[imax + i] = b;
}
// u += alpha d ; r -= alpha Au  (cg_eigs.c:203-204)
__global__ __launch_bounds__(256) void cg_axpy2_kernel(int n, const double* __restrict__ scal, const double* __restrict__ d,
                                                       const double* __restrict__ Au, double* __restrict__ u, double* __restrict__ r) {
  const double a = scal[3];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    u[i] = __dadd_rn(u[i], __dmul_rn(a, d[i]));
    r[i] = __dadd_rn(r[i], __dmul_rn(-a, Au[i]));
  }
}
// d = r + beta d (cg_eigs.c:222)
__global__ __launch_bounds__(256) void cg_xpby_kernel(int n, const double* __restrict__ scal, const double* __restrict__ r, double* __restrict__ d) {
  const double b = scal[4];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = __dadd_rn(__dmul_rn(b, d[i]), r[i]);
}

// variable-length block copy: dst[dst_off[b] + i] = src[src_off[b] + i], i < len[b]  (pack / unpack of face traces)
__global__ __launch_bounds__(256) void copy_blocks_kernel(int n_blocks, const double* __restrict__ src, const long long* __restrict__ src_off,
                                                          double* __restrict__ dst, const long long* __restrict__ dst_off,
                                                          const int* __restrict__ len) {
  for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    const double* s = src + src_off[b];
    double* d = dst + dst_off[b];
    for (int i = threadIdx.x; i < len[b]; i += blockDim.x) d[i] = s[i];
  }
}

void launch_copy_blocks(hipStream_t stream, int n_blocks, const double* src, const long long* src_off, double* dst,
                        const long long* dst_off, const int* len) {
  if (n_blocks <= 0) return;
  hipLaunchKernelGGL(copy_blocks_kernel, dim3(std::min(n_blocks, 4096)), dim3(256), 0, stream, n_blocks, src, src_off, dst, dst_off, len);
  HIP_CHECK(hipGetLastError());
}

static int grid_for(int n) {
  int g = (n + 255) / 256;
  return std::max(1, std::min(g, 4096));
}

void ensure_solver_workspace(d4est_hip_plan* plan) {
  const size_t n = std::max<size_t>((size_t)plan->local_nodes, 1);
  if (!plan->d_work_p) HIP_CHECK(hipMalloc(&plan->d_work_p, n * sizeof(double)));
  if (!plan->d_work_d) HIP_CHECK(hipMalloc(&plan->d_work_d, n * sizeof(double)));
  if (!plan->d_work_r) HIP_CHECK(hipMalloc(&plan->d_work_r, n * sizeof(double)));
  if (!plan->d_reduce) HIP_CHECK(hipMalloc(&plan->d_reduce, (kRedBlocks + 16) * sizeof(double)));
  if (!plan->d_ghost_trace && plan->ghost_trace_doubles > 0) HIP_CHECK(hipMalloc(&plan->d_ghost_trace, (size_t)plan->ghost_trace_doubles * sizeof(double)));
}

void launch_dot(d4est_hip_plan* plan, int n, const double* x, const double* y, double* out_dev) {
  ensure_solver_workspace(plan);
  const int g = std::max(1, std::min((n + 255) / 256, kRedBlocks));
  hipLaunchKernelGGL(dot_partial_kernel, dim3(g), dim3(256), 0, plan->stream, n, x, y, plan->d_reduce);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, plan->stream, g, plan->d_reduce, out_dev);
  HIP_CHECK(hipGetLastError());
}

// Au = A u with the plan's communication hooks (or without ghosts).
// Fork-join on two streams: the trace kernel (and, through the hooks, the ghost exchange) runs on the plan's side
// stream while the volume kernel runs on the main stream -- they only share the read-only u -- and the flux kernel
// joins them.  At config 2 both kernels are latency-structured (~28 us each), so running them side by side hides one.
void apply_operator(d4est_hip_plan* plan, const double* u, double* Au, const ChebyFuse* cf, bool lhs_term) {
  if (!plan->has_faces) D4EST_HIP_ABORT("smoother: the plan has no faces (plan_set_faces)");
  ensure_solver_workspace(plan);
  const bool has_ghost = plan->ghost_trace_doubles > 0;
  if (has_ghost && !plan->exchange_fn) D4EST_HIP_ABORT("apply_lhs: plan has ghost sides but no exchange callback (plan_set_comm)");
  // measured: the two cross-stream event waits cost more than the overlap buys (config 2: 119 -> 129 us; 512 elements:
  // 16 -> 44 us), so the fork is opt-in (tuning value 1) and the default is one stream
  if (hybrid_active(plan)) {
    // mixed-degree / locally refined plan (d4est_hip_direct.hip, "the hybrid operator"): the clean elements' A u from the one-kernel path of
    // their degree bucket, the dirty elements' from traces (dirty elements + their neighbours) + volume + flux on lists.  Disjoint
    // rows, so the order of the launches is free; the small dirty-path kernels go first.
    const int *dirty, *ring;
    int n_dirty, n_ring;
    hybrid_lists(plan, &dirty, &n_dirty, &ring, &n_ring);
    if (cf && !hybrid_can_fuse_update(plan))
      D4EST_HIP_ABORT("apply_operator: the hybrid operator of this plan does not carry a fused update");
    if (cf && !faces_hp(plan)) {
      // conforming mixed-degree plan: the update in every clean bucket's kernel and in the flux kernels of the dirty list (each element's
      // A u is final in exactly one of them); all of them read u and write the new iterate to cf->u_out
      if (!cf->u_out || cf->u_out == u) D4EST_HIP_ABORT("apply_operator: the hybrid operator needs a second vector for the fused update");
      DirectFuse df;
      df.rhs = cf->rhs; df.p = cf->p; df.u_out = cf->u_out; df.r = cf->r; df.alpha = cf->alpha; df.beta = cf->beta;
      df.skip_Au_store = cf->skip_Au_store ? 1 : 0;
      const double* gt = hybrid_hanging(plan) ? plan->d_trace : plan->d_ghost_trace;   // (mixed-aware sides read the trace array)
      if (n_dirty > 0) launch_traces(plan, u, plan->d_trace, false, ring, n_ring);
      launch_flux_hybrid_clean(plan, u, gt, Au, 0, &df);
      if (n_dirty > 0) {
        launch_hybrid_dirty_stiffness(plan, u, Au);
        ChebyFuse cu = *cf;
        cu.u = const_cast<double*>(u);
        launch_flux(plan, plan->d_trace, plan->d_ghost_trace, Au, &cu, dirty, n_dirty);
      }
      launch_flux_hybrid_clean(plan, u, gt, Au, 1, &df);
      return;
    }
    if (hybrid_hanging(plan) && cf) {
      // the Chebyshev update in the kernels' epilogues: the operator kernel updates every element it finishes, the record flux kernel
      // the elements with a record side (disjoint sets; both read u and write the new iterate to cf->u_out)
      if (!cf->u_out || cf->u_out == u) D4EST_HIP_ABORT("apply_operator: the hybrid operator needs a second vector for the fused update");
      DirectFuse df;
      df.rhs = cf->rhs; df.p = cf->p; df.u_out = cf->u_out; df.r = cf->r; df.alpha = cf->alpha; df.beta = cf->beta;
      df.skip_Au_store = cf->skip_Au_store ? 1 : 0;
      launch_traces(plan, u, plan->d_trace, false, ring, n_ring, 2);
      launch_flux_hybrid_clean(plan, u, plan->d_trace, Au, 1, &df);
      ChebyFuse cu = *cf;
      cu.u = const_cast<double*>(u);
      launch_flux_units(plan, plan->d_trace, plan->d_ghost_trace, Au, &cu);
      return;
    }
    if (hybrid_hanging(plan)) {
      // hanging-aware form (a locally refined plan under the hp split): elements with hanging sides are clean too.  The record trace
      // kernel goes first (the clean kernel reads the big elements' sub-mortar blocks for its small sides), the record flux kernel last
      // (it adds the big sides' terms to rows the clean kernel writes, from small-side blocks that kernel exports).
      // (mixed-aware form on a plan without hanging faces: no record kernels, the two-phase kernels whole)
      const bool rec = faces_hp_split(plan);
      if (rec) launch_traces(plan, u, plan->d_trace, false, ring, n_ring, n_dirty > 0 ? 3 : 2);
      else if (n_dirty > 0) launch_traces(plan, u, plan->d_trace, false, ring, n_ring);
      launch_flux_hybrid_clean(plan, u, plan->d_trace, Au, 0);
      if (n_dirty > 0) {
        launch_hybrid_dirty_stiffness(plan, u, Au);
        launch_flux(plan, plan->d_trace, plan->d_ghost_trace, Au, nullptr, dirty, n_dirty, rec ? 1 : 3);
      }
      launch_flux_hybrid_clean(plan, u, plan->d_trace, Au, 1);
      if (rec) launch_flux(plan, plan->d_trace, plan->d_ghost_trace, Au, nullptr, dirty, n_dirty, 2);
      if (lhs_term) add_lhs_mass_term(plan, u, Au);
      return;
    }
    launch_flux_hybrid_clean(plan, u, plan->d_ghost_trace, Au, 0);   // the clean buckets, each on its own stream beside ...
    if (n_dirty > 0) {                                                // ... the dirty path on the plan's stream
      launch_traces(plan, u, plan->d_trace, false, ring, n_ring);
      launch_hybrid_dirty_stiffness(plan, u, Au);
      launch_flux(plan, plan->d_trace, plan->d_ghost_trace, Au, nullptr, dirty, n_dirty);
    }
    launch_flux_hybrid_clean(plan, u, plan->d_ghost_trace, Au, 1);   // join
    if (lhs_term) add_lhs_mass_term(plan, u, Au);
    return;
  }
  if (direct_active(plan)) {
    // one-kernel face terms (d4est_hip_direct.hip): both sides' traces come from u inside the kernel; with ghost sides the trace
    // kernel still runs, to feed the exchange
    const bool mass = lhs_term && plan->d_lhs_coeff;
    // a zeroth-order term held as dense element blocks or as a Galerkin chain (coarse multigrid levels) is not part of the fused
    // kernels' volume stage: volume kernel, the term, then the face kernel
    const bool fused = direct_fused_ok(plan) && !(lhs_term && lhs_extra_term(plan));
    auto run = [&](int vterm) {
      if (cf) {
        DirectFuse df;
        df.rhs = cf->rhs; df.p = cf->p; df.u_out = cf->u_out; df.r = cf->r; df.alpha = cf->alpha; df.beta = cf->beta;
        df.skip_Au_store = (vterm != 0 && cf->skip_Au_store) ? 1 : 0;
        if (!df.u_out || df.u_out == u) D4EST_HIP_ABORT("apply_operator: the direct face kernel needs a second vector for the fused update");
        launch_flux_direct(plan, u, plan->d_ghost_trace, Au, &df, vterm);
      } else {
        launch_flux_direct(plan, u, plan->d_ghost_trace, Au, nullptr, vterm);
      }
    };
    if (has_ghost && fused && !direct_has_element_list(plan)) {
      // Several ranks, whole operator in the kernel.  Only the elements with a ghost (+) side need exchanged data -- the reference,
      // too, packs just its mirror elements (src/Mesh/d4est_ghost_data.c:143-256): the trace kernel runs over THOSE elements to feed
      // the exchange, the operator kernel over the interior elements runs while the blocks travel, and a second, short launch over
      // the boundary elements follows the unpack.  Every element still gets its A u (and its fused update) from exactly one launch.
      const int *bl, *il;
      int nb, ni;
      direct_ghost_split(plan, &bl, &nb, &il, &ni);
      launch_traces(plan, u, plan->d_trace, false, bl, nb);
      plan->exchange_fn(plan->comm_ctx, 0, plan->d_trace, plan->d_ghost_trace);
      const int vterm = mass ? 2 : 1;
      if (ni > 0) {
        direct_set_element_list(plan, il, ni);
        run(vterm);
      }
      plan->exchange_fn(plan->comm_ctx, 1, plan->d_trace, plan->d_ghost_trace);
      direct_set_element_list(plan, bl, nb);
      run(vterm);
      direct_set_element_list(plan, nullptr, 0);
      return;
    }
    // one-kernel face terms (d4est_hip_direct.hip): both sides' traces come from u inside the kernel; with ghost sides the trace
    // kernel still runs, to feed the exchange
    if (has_ghost) {
      launch_traces(plan, u, plan->d_trace, false);
      plan->exchange_fn(plan->comm_ctx, 0, plan->d_trace, plan->d_ghost_trace);
    }
    // the volume term rides in the same kernel (u in, A u out) unless an exchange is to overlap it; the zeroth-order term of
    // plan_set_lhs_coefficient then sits in that kernel's volume stage (vol_term 2)
    const bool whole = !has_ghost && fused;
    const int vterm = whole ? (mass ? 2 : 1) : 0;
    if (!whole) {
      launch_stiffness(plan, u, Au);
      if (lhs_term) add_lhs_mass_term(plan, u, Au);
    }
    if (has_ghost) plan->exchange_fn(plan->comm_ctx, 1, plan->d_trace, plan->d_ghost_trace);
    run(vterm);
    return;
  }
  const bool fork = plan->tuning[D4EST_HIP_TUNE_OVERLAP_TRACES] > 0 && !has_ghost;
  if (fork) {
    if (!plan->side_stream) {
      HIP_CHECK(hipStreamCreateWithFlags(&plan->side_stream, hipStreamNonBlocking));
      HIP_CHECK(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming));
    }
    hipStream_t main = plan->stream;
    HIP_CHECK(hipEventRecord(plan->ev_fork, main));
    HIP_CHECK(hipStreamWaitEvent(plan->side_stream, plan->ev_fork, 0));
    plan->stream = plan->side_stream;
    launch_traces(plan, u, plan->d_trace, false);
    plan->stream = main;
    HIP_CHECK(hipEventRecord(plan->ev_join, plan->side_stream));
    launch_stiffness(plan, u, Au);
    if (lhs_term) add_lhs_mass_term(plan, u, Au);
    HIP_CHECK(hipStreamWaitEvent(main, plan->ev_join, 0));
    launch_flux(plan, plan->d_trace, plan->d_ghost_trace, Au, cf);
    return;
  }
  // with ghost sides the exchange callbacks enqueue on the plan's (single) stream: traces, post, volume, complete, flux
  launch_traces(plan, u, plan->d_trace, false);
  if (has_ghost) plan->exchange_fn(plan->comm_ctx, 0, plan->d_trace, plan->d_ghost_trace);
  launch_stiffness(plan, u, Au);  // overlaps the exchange: the volume term needs no ghost data
  if (lhs_term) add_lhs_mass_term(plan, u, Au);  // before the flux kernel, so an update fused into its epilogue sees the whole A u
  if (has_ghost) plan->exchange_fn(plan->comm_ctx, 1, plan->d_trace, plan->d_ghost_trace);
  launch_flux(plan, plan->d_trace, plan->d_ghost_trace, Au, cf);
}

// the zeroth-order term of a linearised nonlinear problem (d4est_quadrature_apply_fofufofvlilj per element, added with axpy 1.0:
// e.g. constant_density_star_apply_jac_add_nonlinear_term, src/Problems/ConstantDensityStar/constant_density_star_fcns.h:528-603)
void add_lhs_mass_term(d4est_hip_plan* plan, const double* u, double* Au) {
  // on a coarse multigrid level the reference adds the Galerkin-restricted term instead (constant_density_star_fcns.h:806-850)
  if (plan->d_lhs_blocks) { add_lhs_blocks_term(plan, u, Au); return; }
  if (plan->lhs_chain) { add_lhs_chain_term(plan, u, Au); return; }
  if (!plan->d_lhs_coeff || plan->local_nodes == 0) return;
  const int n = plan->local_nodes;
  if (!plan->d_work_m) HIP_CHECK(hipMalloc(&plan->d_work_m, (size_t)n * sizeof(double)));
  launch_mass_like(plan, 3, u, plan->d_work_m, plan->d_lhs_c, 0);
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, plan->stream, n, plan->d_work_m, Au);
  HIP_CHECK(hipGetLastError());
}

// w J c at the quadrature nodes of the elements of one (deg, deg_quad) bucket (elements that alias each other's quadrature block -- the
// copies of a Schwarz subdomain plan -- write the same values)
__global__ __launch_bounds__(256) void lhs_wjc_kernel(const int* __restrict__ qs_list, int n_bucket, int NQ, const double* __restrict__ w,
                                                      const double* __restrict__ J, const double* __restrict__ c, double* __restrict__ wjc) {
  const int NQ3 = NQ * NQ * NQ;
  for (int i = blockIdx.x; i < n_bucket; i += gridDim.x) {
    const int qs = qs_list[i];
    for (int n = threadIdx.x; n < NQ3; n += blockDim.x) {
      const int a = n % NQ, b = (n / NQ) % NQ, k = n / (NQ * NQ);
      wjc[qs + n] = (w[k] * (w[b] * w[a])) * (J[qs + n] * c[qs + n]);
    }
  }
}

const double* ensure_lhs_wjc(d4est_hip_plan* plan) {
  if (!plan->d_lhs_coeff) D4EST_HIP_ABORT("zeroth-order term: no coefficient is set (plan_set_lhs_coefficient)");
  if (!plan->has_geometry) D4EST_HIP_ABORT("zeroth-order term: d4est_hip_plan_set_geometry was not called");
  if (plan->lhs_wjc_valid) return plan->d_lhs_wjc;
  if (!plan->d_lhs_wjc) HIP_CHECK(hipMalloc(&plan->d_lhs_wjc, std::max<size_t>((size_t)plan->local_nodes_quad, 1) * sizeof(double)));
  for (const Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    hipLaunchKernelGGL(lhs_wjc_kernel, dim3(std::min(bk.n_elem, 8192)), dim3(256), 0, plan->stream, plan->d_qs_list + bk.elem_offset, bk.n_elem,
                       bk.NQ, bk.d_w, plan->d_J, plan->d_lhs_c, plan->d_lhs_wjc);
  }
  HIP_CHECK(hipGetLastError());
  plan->lhs_wjc_valid = true;
  return plan->d_lhs_wjc;
}

void launch_residual(d4est_hip_plan* plan, int n, const double* rhs, const double* Au, double* r) {
  if (n > 0) hipLaunchKernelGGL(residual_kernel, dim3(grid_for(n)), dim3(256), 0, plan->stream, n, rhs, Au, r, (double*)nullptr);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void sub_inplace_kernel(int n, const double* __restrict__ a, double* __restrict__ r) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) r[i] = r[i] - a[i];
}
void launch_residual_inplace_sub(d4est_hip_plan* plan, int n, const double* a, double* r) {
  if (n > 0) hipLaunchKernelGGL(sub_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, plan->stream, n, a, r);
  HIP_CHECK(hipGetLastError());
}

void launch_residual_inplace(d4est_hip_plan* plan, int n, const double* rhs, double* r) {
  if (n > 0) hipLaunchKernelGGL(residual_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, plan->stream, n, rhs, r);
  HIP_CHECK(hipGetLastError());
}

static void cheby_iterate_body(d4est_hip_plan* plan, double* u, const double* rhs, double* Au, double* r, int iter, double lmin,
                               double lmax, int compute_residual_at_end);

void cheby_iterate(d4est_hip_plan* plan, double* u, const double* rhs, double* Au, double* r, int iter, double lmin, double lmax,
                   int compute_residual_at_end) {
  ensure_solver_workspace(plan);
  // launch-bound meshes: replay the whole loop (iter x {traces, volume, flux, update}) as one hipGraph
  const bool graph = plan->tuning[D4EST_HIP_TUNE_GRAPH] > 0 && plan->stream != nullptr && !plan->exchange_fn &&
                     plan->ghost_trace_doubles == 0 && plan->tuning[D4EST_HIP_TUNE_OVERLAP_TRACES] <= 0;
  if (!graph) {
    cheby_iterate_body(plan, u, rhs, Au, r, iter, lmin, lmax, compute_residual_at_end);
    return;
  }
  auto& k = plan->cheby_graph_key;
  const bool same = plan->cheby_graph && k.u == u && k.rhs == rhs && k.Au == Au && k.r == r && k.iter == iter &&
                    k.flag == compute_residual_at_end && k.lmin == lmin && k.lmax == lmax && k.stream == plan->stream;
  if (!same) {
    if (plan->cheby_graph) { HIP_CHECK(hipGraphExecDestroy(plan->cheby_graph)); plan->cheby_graph = nullptr; }
    if (!plan->has_faces) D4EST_HIP_ABORT("smoother: the plan has no faces (plan_set_faces)");
    // every lazy allocation of the operator (scratch of the generic kernels, the work vector of the zeroth-order term, trace buffers)
    // must exist before the capture starts -- hipMalloc / hipFree are illegal inside it.  One throw-away apply on the solver's own
    // work vectors takes exactly the code path the captured loop will take; it happens once per captured argument set.
    HIP_CHECK(hipMemsetAsync(plan->d_work_d, 0, std::max<size_t>((size_t)plan->local_nodes, 1) * sizeof(double), plan->stream));
    apply_operator(plan, plan->d_work_d, plan->d_work_r);
    if (direct_active(plan)) (void)direct_second_vector(plan);   // the second iterate vector of the fused update
    if (hybrid_active(plan) && hybrid_can_fuse_update(plan)) (void)hybrid_second_vector(plan);
    HIP_CHECK(hipStreamSynchronize(plan->stream));
    hipGraph_t g = nullptr;
    HIP_CHECK(hipStreamBeginCapture(plan->stream, hipStreamCaptureModeThreadLocal));
    cheby_iterate_body(plan, u, rhs, Au, r, iter, lmin, lmax, compute_residual_at_end);
    HIP_CHECK(hipStreamEndCapture(plan->stream, &g));
    HIP_CHECK(hipGraphInstantiate(&plan->cheby_graph, g, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(g));
    k.u = u; k.rhs = rhs; k.Au = Au; k.r = r; k.iter = iter; k.flag = compute_residual_at_end; k.lmin = lmin; k.lmax = lmax;
    k.stream = plan->stream;
  }
  HIP_CHECK(hipGraphLaunch(plan->cheby_graph, plan->stream));
}

static void cheby_iterate_body(d4est_hip_plan* plan, double* u, const double* rhs, double* Au, double* r, int iter, double lmin,
                               double lmax, int compute_residual_at_end) {
  const int n = plan->local_nodes;
  const double d = (lmax + lmin) * .5, c = (lmax - lmin) * .5;
  double alpha = 0.0, beta = 0.0;
  HIP_CHECK(hipMemsetAsync(plan->d_work_p, 0, std::max<size_t>((size_t)n, 1) * sizeof(double), plan->stream));
  // (the hybrid operator in its hanging-aware form carries the update in the operator kernel and the record flux kernel: hybrid_can_fuse_update)
  const bool fuse_hy = plan->tuning[D4EST_HIP_TUNE_FUSE_UPDATE] != 0 && hybrid_active(plan) && hybrid_can_fuse_update(plan);
  const bool fuse = fuse_hy || (plan->tuning[D4EST_HIP_TUNE_FUSE_UPDATE] != 0 && flux_can_fuse_update(plan));
  // the direct face kernel reads the neighbours' u, so its fused update writes the new iterate to a second vector: the iterates
  // alternate between the caller's u and a plan-owned one (copied back after an odd number of iterations)
  const bool pingpong = fuse && (fuse_hy || direct_active(plan));
  double* ub = pingpong ? (fuse_hy ? hybrid_second_vector(plan) : direct_second_vector(plan)) : u;
  double* const u_caller = u;
  for (int i = 0; i < iter; ++i) {
    if (i == 0) alpha = 1. / d;
    else if (i == 1) alpha = 2. * d / (2 * d * d - c * c);
    else alpha = 1. / (d - (alpha * c * c / 4.));
    beta = alpha * d - 1.;
    // r = alpha (rhs - A u) is observable only after the LAST iteration, and only when the caller does not ask for the true
    // residual at the end (which overwrites it): every other iteration skips the store (one vector of HBM writes per iteration)
    double* r_out = (i == iter - 1 && compute_residual_at_end != 1) ? r : nullptr;
    if (fuse) {   // the update rides in the flux kernel's epilogue: 3 kernels per iteration instead of 4
      ChebyFuse cf;
      cf.rhs = rhs; cf.p = plan->d_work_p; cf.u = u; cf.r = r_out; cf.alpha = alpha; cf.beta = beta;
      if (pingpong) {
        cf.u_out = ub;
        // A u of an iteration is observable only after the last one (the caller's Au holds A u of the last-but-one iterate, as in the
        // reference, :119-154): the one-kernel operator keeps it in registers in between -- one vector of stores less per iteration
        cf.skip_Au_store = (i < iter - 1) ? 1 : 0;
        apply_operator(plan, u, Au, &cf);
        std::swap(u, ub);
      } else {
        apply_operator(plan, u, Au, &cf);
      }
    } else {
      apply_operator(plan, u, Au);
      if (n > 0) hipLaunchKernelGGL(cheby_update_kernel, dim3(grid_for(n)), dim3(256), 0, plan->stream, n, rhs, Au, alpha, beta, r_out, plan->d_work_p, u);
    }
  }
  if (u != u_caller) {   // odd number of ping-pong iterations: the last iterate sits in the plan's vector
    HIP_CHECK(hipMemcpyAsync(u_caller, u, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, plan->stream));
    u = u_caller;
  }
  if (compute_residual_at_end == 1) {
    apply_operator(plan, u, Au);
    if (n > 0) hipLaunchKernelGGL(residual_kernel, dim3(grid_for(n)), dim3(256), 0, plan->stream, n, rhs, Au, r, (double*)nullptr);
  }
  HIP_CHECK(hipGetLastError());
}

static void gershgorin(int use_new, int i, int local_nodes, double a0, double b0, double a1, double b1, double* mx) {
  // src/Solver/d4est_solver_cg_eigs.c:9-33 (old) and :36-64 (new)
  double diag, off;
  if (!use_new) {
    if (i != 0 && i < local_nodes - 1) { diag = (1. / a1 + b0 / a0); off = std::fabs(std::sqrt(b1) / a1) + std::fabs(std::sqrt(b0) / a0); }
    else if (i == 0) { diag = 1. / a1; off = std::sqrt(b1) / a1; }
    else { diag = 1. / a1 + b0 / a0; off = std::fabs(std::sqrt(b0) / a0); }
  } else {
    if (i != 0) { diag = (1. / a1 + b0 / a0); off = std::fabs(std::sqrt(b0) / a0); }
    else { diag = 1. / a1; off = std::sqrt(b1) / a1; }
  }
  *mx = diag + off;
}

double cg_eigs(d4est_hip_plan* plan, double* u, const double* rhs, double* Au, int imax, int use_new, double* hist_out) {
  ensure_solver_workspace(plan);
  const int n = plan->local_nodes;
  if (imax < 1) D4EST_HIP_ABORT("cg_eigs: imax = %d", imax);
  double* scal = plan->d_reduce + kRedBlocks;  // 16 doubles after the partial sums
  double* hist = nullptr;
  HIP_CHECK(hipMalloc(&hist, 2 * (size_t)imax * sizeof(double)));
  double* d = plan->d_work_d;
  double* r = plan->d_work_r;
  const int g = grid_for(n);
  apply_operator(plan, u, Au);
  if (n > 0) hipLaunchKernelGGL(residual_kernel, dim3(g), dim3(256), 0, plan->stream, n, rhs, Au, r, d);  // r = rhs - Au ; d = r
  launch_dot(plan, n, r, r, &scal[0]);
  if (plan->allreduce_fn) plan->allreduce_fn(plan->comm_ctx, &scal[0], 1);
  for (int i = 0; i < imax; ++i) {
    apply_operator(plan, d, Au);
    launch_dot(plan, n, d, Au, &scal[2]);
    if (plan->allreduce_fn) plan->allreduce_fn(plan->comm_ctx, &scal[2], 1);
    hipLaunchKernelGGL(cg_alpha_kernel, dim3(1), dim3(1), 0, plan->stream, scal, hist, i);
    if (n > 0) hipLaunchKernelGGL(cg_axpy2_kernel, dim3(g), dim3(256), 0, plan->stream, n, scal, d, Au, u, r);
    launch_dot(plan, n, r, r, &scal[0]);
    if (plan->allreduce_fn) plan->allreduce_fn(plan->comm_ctx, &scal[0], 1);
    hipLaunchKernelGGL(cg_beta_kernel, dim3(1), dim3(1), 0, plan->stream, scal, hist, i, imax);
    if (n > 0) hipLaunchKernelGGL(cg_xpby_kernel, dim3(g), dim3(256), 0, plan->stream, n, scal, r, d);
  }
  HIP_CHECK(hipGetLastError());
  std::vector<double> h(2 * (size_t)imax);
  HIP_CHECK(hipMemcpyAsync(h.data(), hist, h.size() * sizeof(double), hipMemcpyDeviceToHost, plan->stream));
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  HIP_CHECK(hipFree(hist));
  double bound = 0.0, a_old = -1., b_old = -1.;
  for (int i = 0; i < imax; ++i) {
    double mx;
    gershgorin(use_new, i, n, a_old, b_old, h[i], h[imax + i], &mx);
    bound = (i > 0) ? std::max(bound, mx) : mx;
    a_old = h[i];
    b_old = h[imax + i];
  }
  if (hist_out) std::copy(h.begin(), h.end(), hist_out);
  return bound;
}

void launch_cheby_update(d4est_hip_plan* plan, int n, const double* rhs, const double* Au, double alpha, double beta, double* r,
                         double* p, double* u) {
  if (n > 0) hipLaunchKernelGGL(cheby_update_kernel, dim3(grid_for(n)), dim3(256), 0, plan->stream, n, rhs, Au, alpha, beta, r, p, u);
  HIP_CHECK(hipGetLastError());
}

}  // namespace d4est_hip
