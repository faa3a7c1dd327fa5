// Volume (element-local) kernels of the MI355X DG engine: stiffness, mass-like
// applies, collocation derivatives and the geometry pre-combination.
//
// Replaces, for all local elements at once:
//   d4est_quadrature_apply_stiffness_matrix  (src/Quadrature/d4est_quadrature.c:263-382)
//   d4est_quadrature_apply_mass_matrix       (:385-477)
//   d4est_quadrature_apply_galerkin_integral (:142-213)
//   d4est_quadrature_interpolate             (:966-1016)
//   d4est_laplacian_compute_dudr             (src/dGMath/d4est_laplacian.c:237-282)
//
// Design (gfx950): the 27-pass reference loop is fused into ONE sum-factorised
// pass  out = sum_lp D_lp^T V^T [ W J G_{lp,l} (V D_l u) ].  One element is owned
// by NQ*NQ threads (one 64-lane wavefront at p = 7); each thread keeps a whole
// tensor COLUMN in registers and contracts it with the 1-D operator, whose entries
// are wave-uniform and come from scalar loads.  Between the three tensor
// directions the columns are transposed through LDS with an odd (padded) leading
// dimension so that every column read is bank-conflict free.  HBM traffic per
// element is the algorithmic minimum: u once, the 6-entry symmetric metric once,
// Au once (64 B/DoF at Nq = N).
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <cstring>

#include "d4est_hip_internal.h"
#include "d4est_hip_maps.h"
#include "d4est_hip_tables.h"
#include "d4est_hip_wave.h"
#include "d4est_hip_mwave.h"

namespace d4est_hip {

// ---------------------------------------------------------------------------
// compile-time configuration per (N, NQ)
// ---------------------------------------------------------------------------
template <int N, int NQ>
struct VolCfg {
  static_assert(NQ >= N, "quadrature degree below polynomial degree is served by the generic path");
  static constexpr int PL = NQ * NQ;                                   // threads per element
  static constexpr int EPB = (PL <= 64) ? (64 / PL) : 1;               // elements per block
  static constexpr int THREADS = (PL <= 64) ? 64 : ((PL + 63) / 64) * 64;
  static constexpr int PN = N | 1;                                     // odd padded column lengths
  static constexpr int PQ = NQ | 1;
  static constexpr int FS = NQ * NQ * PQ;                              // doubles per LDS field
  static constexpr int LDS_PER_ELEM = 3 * FS;
  static constexpr size_t LDS_BYTES = (size_t)EPB * LDS_PER_ELEM * sizeof(double);
};


// ---------------------------------------------------------------------------
// stiffness:  Au_e = sum_{lp,l} D_lp^T V^T [ M_{lp,l} (V D_l u_e) ]
// ---------------------------------------------------------------------------
// PF = true: the thread's 6*NQ metric entries are requested at kernel entry (before the
// forward contractions) so that HBM latency overlaps the S1-S3 arithmetic; costs 12*NQ VGPRs.
// EO = true: the four operator arguments are the even-odd tables (Bop = B^T's, Gop = G^T's, BopT = B's, GopT = G's)
template <int N, int NQ, bool PF, bool EO = false, bool NT = false /* stream mode, d4est_hip_wave.h */>
__global__ __launch_bounds__((VolCfg<N, NQ>::THREADS)) void stiffness_kernel(
    const double* __restrict__ u, double* __restrict__ Au, const double* __restrict__ metric,
    const int* __restrict__ ns_list, const int* __restrict__ qs_list,
    int n_bucket, const double* __restrict__ Bop, const double* __restrict__ Gop,
    const double* __restrict__ BopT, const double* __restrict__ GopT) {
  using C = VolCfg<N, NQ>;
  constexpr int PL = C::PL, PN = C::PN, PQ = C::PQ, FS = C::FS;
  constexpr int N3 = N * N * N, NQ3 = NQ * NQ * NQ;
  extern __shared__ __attribute__((aligned(16))) double smem[];

  const int tid = threadIdx.x;
  const int slot = tid / PL;
  const int te = tid - slot * PL;
  const int a = te % NQ, b = te / NQ;
  const int ei = blockIdx.x * C::EPB + slot;
  const bool active = (slot < C::EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * C::LDS_PER_ELEM;
  double* R1 = R0 + FS;
  double* R2 = R1 + FS;

  int ns = 0, qs = 0;
  if (active) {
    ns = ns_list[ei];
    qs = qs_list[ei];
  }

  // ---- load u_e (coalesced) into R0[k][j][i], i fastest, padded PN
  if (active) {
    load_element_image<N, PL, PN>(R0, u + ns, te);
  }
  // ---- metric prefetch: symmetric (rr,rs,rt,ss,st,tt) at the thread's quadrature column, coalesced along (iq,jq)
  double mreg[PF ? 6 : 1][PF ? NQ : 1];
  if (PF && active) {
    const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq)
#pragma unroll
      for (int c = 0; c < 6; ++c) mreg[c][kq] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * kq]);
  }
  __syncthreads();

  // ---- S1: r-contraction, thread (j=a, k=b): B u and G u along i -> R1,R2 [k][iq][j]
  if (active && a < N && b < N) {
    double x[N], br[NQ], gr[NQ];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = lds_ld(&R0[i + PN * (a + N * b)]);
    fwd<N, NQ, EO, false>(BopT, x, br);
    fwd<N, NQ, EO, true>(GopT, x, gr);
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
      R1[a + PN * (iq + NQ * b)] = br[iq];
      R2[a + PN * (iq + NQ * b)] = gr[iq];
    }
  }
  __syncthreads();

  // ---- S2: s-contraction, thread (iq=a, k=b)
  double t_bg[NQ], t_gb[NQ], t_bb[NQ];  // B_s G_r u | G_s B_r u | B_s B_r u
  if (active && b < N) {
    double x[N];
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = lds_ld(&R1[j + PN * (a + NQ * b)]);
    fwd<N, NQ, EO, false>(BopT, x, t_bb);
    fwd<N, NQ, EO, true>(GopT, x, t_gb);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = lds_ld(&R2[j + PN * (a + NQ * b)]);
    fwd<N, NQ, EO, false>(BopT, x, t_bg);
  }
  __syncthreads();
  if (active && b < N) {
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq) {  // [jq][iq][k], k fastest
      R0[b + PN * (a + NQ * jq)] = t_bg[jq];
      R1[b + PN * (a + NQ * jq)] = t_gb[jq];
      R2[b + PN * (a + NQ * jq)] = t_bb[jq];
    }
  }
  __syncthreads();

  // ---- S3: t-contraction, thread (iq=a, jq=b): reference-space gradient at quadrature nodes
  double gr[NQ], gs[NQ], gt[NQ];
  double ca[N], cb[N], cc[N];
  if (active) {
    double x[N];
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = lds_ld(&R0[k + PN * (a + NQ * b)]);
    fwd<N, NQ, EO, false>(BopT, x, gr);
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = lds_ld(&R1[k + PN * (a + NQ * b)]);
    fwd<N, NQ, EO, false>(BopT, x, gs);
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = lds_ld(&R2[k + PN * (a + NQ * b)]);
    fwd<N, NQ, EO, true>(GopT, x, gt);

    // ---- quadrature-point stage: symmetric metric (rr,rs,rt,ss,st,tt), coalesced along (iq,jq)
    const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {
      const int q = NQ * NQ * kq;
      const double m0 = PF ? mreg[0][kq] : m[q], m1 = PF ? mreg[1][kq] : m[NQ3 + q], m2 = PF ? mreg[2][kq] : m[2 * NQ3 + q];
      const double m3 = PF ? mreg[3][kq] : m[3 * NQ3 + q], m4 = PF ? mreg[4][kq] : m[4 * NQ3 + q], m5 = PF ? mreg[5][kq] : m[5 * NQ3 + q];
      const double r = gr[kq], s = gs[kq], t = gt[kq];
      gr[kq] = m0 * r + m1 * s + m2 * t;
      gs[kq] = m1 * r + m3 * s + m4 * t;
      gt[kq] = m2 * r + m4 * s + m5 * t;
    }

    // ---- S5: t-contraction transposed (registers)
    bwd<NQ, N, EO, false, false>(Bop, gr, ca);
    bwd<NQ, N, EO, false, false>(Bop, gs, cb);
    bwd<NQ, N, EO, true, false>(Gop, gt, cc);
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int k = 0; k < N; ++k) {  // [k][iq][jq], jq fastest
      R0[b + PQ * (a + NQ * k)] = ca[k];
      R1[b + PQ * (a + NQ * k)] = cb[k];
      R2[b + PQ * (a + NQ * k)] = cc[k];
    }
  }
  __syncthreads();

  // ---- S6: s-contraction transposed, thread (iq=a, k=b)
  double ar[N], bs[N];
  if (active && b < N) {
    double x[NQ];
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R0[jq + PQ * (a + NQ * b)]);
    bwd<NQ, N, EO, false, false>(Bop, x, ar);
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R1[jq + PQ * (a + NQ * b)]);
    bwd<NQ, N, EO, true, false>(Gop, x, bs);
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R2[jq + PQ * (a + NQ * b)]);
    bwd<NQ, N, EO, false, true>(Bop, x, bs);
  }
  __syncthreads();
  if (active && b < N) {
#pragma unroll
    for (int j = 0; j < N; ++j) {  // [k][j][iq], iq fastest
      R0[a + PQ * (j + N * b)] = ar[j];
      R1[a + PQ * (j + N * b)] = bs[j];
    }
  }
  __syncthreads();

  // ---- S7: r-contraction transposed, thread (j=a, k=b)
  if (active && a < N && b < N) {
    double x[NQ], o[N];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) x[iq] = lds_ld(&R0[iq + PQ * (a + N * b)]);
    bwd<NQ, N, EO, true, false>(Gop, x, o);
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) x[iq] = lds_ld(&R1[iq + PQ * (a + N * b)]);
    bwd<NQ, N, EO, false, true>(Bop, x, o);
#pragma unroll
    for (int i = 0; i < N; ++i) R2[i + PN * (a + N * b)] = o[i];
  }
  __syncthreads();

  // ---- store Au_e (coalesced)
  if (active) {
    store_element_image<N, PL, PN, NT>(Au + ns, R2, te);
  }
}

// ---------------------------------------------------------------------------
// Single-wavefront variant (NQ*NQ <= 64, i.e. p <= 7): the whole element lives in one
// wave, so LDS hand-offs need no s_barrier (program order + the waitcnt the compiler
// inserts; __syncthreads() in a 64-thread workgroup lowers to exactly that) and only
// TWO LDS fields are live at any time: the three S2 -> S3 (and S5 -> S6) fields are
// passed one after the other through the same buffer.  9.2 KB of LDS per element at
// p = 7 lets 16 waves (4 per SIMD) share a CU, which covers config 2 (4096 elements)
// in a single resident round.
// ---------------------------------------------------------------------------

template <int N, int NQ, bool PF, bool EO = false, bool AFF = false, bool NT = false /* stream mode, d4est_hip_wave.h */>
__global__ __launch_bounds__((WaveCfg<N, NQ>::THREADS), (WaveCfg<N, NQ>::THREADS == 64 ? (PF ? 3 : 4) : ((!AFF && N <= 13) ? D4EST_HIP_MW_WAVES : ((!AFF && N == 14) ? 3 : 1)))) void stiffness_wave_kernel(
    const double* __restrict__ u, double* __restrict__ Au, const double* __restrict__ metric,
    const int* __restrict__ ns_list, const int* __restrict__ qs_list, int n_bucket, const double* __restrict__ Bop,
    const double* __restrict__ Gop, const double* __restrict__ BopT, const double* __restrict__ GopT, int stagger,
    const double* __restrict__ affine = nullptr, const double* __restrict__ wq = nullptr) {
  // AFF: affine bucket -- the metric of node (a, b, kq) is rebuilt as (w_a w_b w_kq) * c[0..5] from the element's six
  // constants (affine + 6 * element) instead of being streamed (16 B/DoF instead of 64: SURVEY.md section 8d "affine path")
  // Phase stagger for single-round grids: when every resident wave starts at once, all waves load u, then all
  // contract, then all stream the metric ... and the memory pipe idles during the arithmetic phases.  Delaying every
  // other resident "row" of workgroups (block id bit 8 = the second batch the dispatcher places on each CU) by about
  // one forward phase lets one half's metric stream run under the other half's contractions.
  using C = WaveCfg<N, NQ>;
  if constexpr (C::THREADS == 64) {
  if (stagger > 0 && ((blockIdx.x >> 8) & 1)) {
    for (int s_ = 0; s_ < stagger; ++s_) __builtin_amdgcn_s_sleep(16);  // 16 * 64 cycles per iteration
  }
  }
  constexpr int PL = C::PL, PN = C::PN, FS = C::FS;
  extern __shared__ __attribute__((aligned(16))) double smem[];

  const int tid = threadIdx.x;
  const int slot = tid / PL;
  const int te = tid - slot * PL;
  const int a = te % NQ, b = te / NQ;
  const int ei = blockIdx.x * C::EPB + slot;
  const bool active = (slot < C::EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * C::LDS_PER_ELEM;
  double* R1 = R0 + FS;

  int ns = 0, qs = 0;
  if (active) {
    ns = ns_list[ei];
    qs = qs_list[ei];
  }
  if constexpr (C::THREADS > 64) {
    // one element per workgroup: the strides are wave-uniform -- scalar registers, so every global access below is
    // SGPR base + one VGPR lane offset (frees the 64-bit address pairs the per-lane form keeps live at the quadrature stage)
    ns = __builtin_amdgcn_readfirstlane(ns);
    qs = __builtin_amdgcn_readfirstlane(qs);
  }

  if (active) {
    load_element_image<N, PL, PN>(R0, u + ns, te);
  }
  if constexpr (kMwCollocated<N, NQ, PF, EO>)   // Gop / GopT are then the tables of the differentiation matrix on the quadrature nodes
    stiffness_mw_element_cg<N, AFF, false, NT>(R0, R1, metric, qs, ei, active, te, a, b, Bop, BopT, Gop, GopT, affine, wq);
  else
    stiffness_mw_element<N, NQ, PF, EO, AFF, false, NT>(R0, R1, metric, qs, ei, active, te, a, b, Bop, Gop, BopT, GopT, affine, wq);
  if (active) {
    store_element_image<N, PL, PN, NT>(Au + ns, R0, te);
  }
}

// ---------------------------------------------------------------------------
// Two-wavefront variant (NQ*NQ <= 64): an element is owned by TWO waves; in every contraction each
// wave produces one half of the OUTPUT indices of every column (from the full input column), so a
// thread's register state is half a column per field.  That halves the VGPR cost of holding the
// metric: the thread's 6 x NQ/2 metric doubles are requested at kernel entry (HBM latency overlaps
// the forward contractions) and the kernel still fits 128 VGPRs = 4 waves/SIMD.  The price is one
// extra LDS exchange (the flux at the quadrature nodes, before the transposed t-contraction) and
// s_barriers between stages.
// ---------------------------------------------------------------------------
// y[0..NH) = rows [o0, o0+NH) of (op x), op given transposed (NI x NO row-major)
template <int NI, int NO, int NH>
__device__ __forceinline__ void contract_n_part(const double* __restrict__ opT, int o0, const double* x, double* y) {
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    sdouble_ptr row = launder(opT + i * NO + o0);
    if (i == 0) {
#pragma unroll
      for (int o = 0; o < NH; ++o) y[o] = row[o] * x[0];
    } else {
#pragma unroll
      for (int o = 0; o < NH; ++o) y[o] = fma(row[o], x[i], y[o]);
    }
  }
}
// y[0..NH) (+)= entries [o0, o0+NH) of (op^T x), op is NI x NO row-major
template <int NI, int NO, int NH, bool ACC>
__device__ __forceinline__ void contract_t_part(const double* __restrict__ op, int o0, const double* x, double* y) {
  if (!ACC) {
#pragma unroll
    for (int o = 0; o < NH; ++o) y[o] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    sdouble_ptr row = launder(op + i * NO + o0);
#pragma unroll
    for (int o = 0; o < NH; ++o) y[o] = fma(row[o], x[i], y[o]);
  }
}

template <int N, int NQ>
__global__ __launch_bounds__(128, 4) void stiffness_pair_kernel(
    const double* __restrict__ u, double* __restrict__ Au, const double* __restrict__ metric,
    const int* __restrict__ ns_list, const int* __restrict__ qs_list, int n_bucket, const double* __restrict__ Bop,
    const double* __restrict__ Gop, const double* __restrict__ BopT, const double* __restrict__ GopT) {
  static_assert(N % 2 == 0 && NQ % 2 == 0, "pair kernel splits the output indices in two equal halves");
  using C = WaveCfg<N, NQ>;
  constexpr int PL = C::PL, PN = C::PN, PQ = C::PQ, FS = C::FS, EPB = C::EPB;
  constexpr int N3 = N * N * N, NQ3 = NQ * NQ * NQ;
  constexpr int HN = N / 2, HQ = NQ / 2;
  constexpr int LDS_PER_ELEM = 3 * FS;
  extern __shared__ __attribute__((aligned(16))) double smem[];

  // which half of the output indices this wave produces; readfirstlane makes the wave-uniformity explicit (SGPR)
  const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int slot = lane / PL;
  const int te = lane - slot * PL;
  const int a = te % NQ, b = te / NQ;
  const int ei = blockIdx.x * EPB + slot;
  const bool active = (slot < EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * LDS_PER_ELEM;
  double* R1 = R0 + FS;
  double* R2 = R1 + FS;
  const int q0 = h * HQ, n0 = h * HN;  // first output index of this wave's half

  int ns = 0, qs = 0;
  if (active) {
    ns = ns_list[ei];
    qs = qs_list[ei];
  }
  // ---- u_e -> LDS: both waves together, 2*PL threads per element
  if (active) {
#pragma unroll
    for (int idx = te + h * PL; idx < N3; idx += 2 * PL) {
      const int i = idx % N, j = (idx / N) % N, k = idx / (N * N);
      R0[i + PN * (j + N * k)] = u[ns + idx];
    }
  }
  // ---- metric prefetch for this thread's HQ quadrature nodes of column (iq=a, jq=b)
  double mreg[6][HQ];
  if (active) {
    const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b) + NQ * NQ * q0;
#pragma unroll
    for (int kq = 0; kq < HQ; ++kq)
#pragma unroll
      for (int c = 0; c < 6; ++c) mreg[c][kq] = m[c * NQ3 + NQ * NQ * kq];
  }
  __syncthreads();

  // ---- S1: thread (j=a, k=b), outputs iq in [q0, q0+HQ): R1 <- B_r u, R2 <- G_r u as [k][iq][j]
  if (active && a < N && b < N) {
    double x[N], br[HQ], gr[HQ];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = lds_ld(&R0[i + PN * (a + N * b)]);
    contract_n_part<N, NQ, HQ>(BopT, q0, x, br);
    contract_n_part<N, NQ, HQ>(GopT, q0, x, gr);
#pragma unroll
    for (int iq = 0; iq < HQ; ++iq) {
      R1[a + PN * (q0 + iq + NQ * b)] = br[iq];
      R2[a + PN * (q0 + iq + NQ * b)] = gr[iq];
    }
  }
  __syncthreads();

  // ---- S2: thread (iq=a, k=b), outputs jq in [q0, q0+HQ)
  {
    double x[N], t_bb[HQ], t_gb[HQ], t_bg[HQ];
    const bool on2 = active && b < N;
    if (on2) {
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = lds_ld(&R1[j + PN * (a + NQ * b)]);
      contract_n_part<N, NQ, HQ>(BopT, q0, x, t_bb);
      contract_n_part<N, NQ, HQ>(GopT, q0, x, t_gb);
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = lds_ld(&R2[j + PN * (a + NQ * b)]);
      contract_n_part<N, NQ, HQ>(BopT, q0, x, t_bg);
    }
    __syncthreads();
    if (on2) {
#pragma unroll
      for (int jq = 0; jq < HQ; ++jq) {  // [jq][iq][k]
        R0[b + PN * (a + NQ * (q0 + jq))] = t_bg[jq];
        R1[b + PN * (a + NQ * (q0 + jq))] = t_gb[jq];
        R2[b + PN * (a + NQ * (q0 + jq))] = t_bb[jq];
      }
    }
  }
  __syncthreads();

  // ---- S3: thread (iq=a, jq=b), outputs kq in [q0, q0+HQ); then the metric multiply on those nodes
  double fr[HQ], fs[HQ], ft[HQ];
  if (active) {
    double x[N];
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = lds_ld(&R0[k + PN * (a + NQ * b)]);
    contract_n_part<N, NQ, HQ>(BopT, q0, x, fr);
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = lds_ld(&R1[k + PN * (a + NQ * b)]);
    contract_n_part<N, NQ, HQ>(BopT, q0, x, fs);
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = lds_ld(&R2[k + PN * (a + NQ * b)]);
    contract_n_part<N, NQ, HQ>(GopT, q0, x, ft);
#pragma unroll
    for (int kq = 0; kq < HQ; ++kq) {
      const double r = fr[kq], s = fs[kq], t = ft[kq];
      fr[kq] = mreg[0][kq] * r + mreg[1][kq] * s + mreg[2][kq] * t;
      fs[kq] = mreg[1][kq] * r + mreg[3][kq] * s + mreg[4][kq] * t;
      ft[kq] = mreg[2][kq] * r + mreg[4][kq] * s + mreg[5][kq] * t;
    }
  }
  __syncthreads();
  // ---- exchange the flux halves through LDS ([jq][iq][kq], kq fastest) so each wave sees the full kq column
  if (active) {
#pragma unroll
    for (int kq = 0; kq < HQ; ++kq) {
      R0[q0 + kq + PQ * (a + NQ * b)] = fr[kq];
      R1[q0 + kq + PQ * (a + NQ * b)] = fs[kq];
      R2[q0 + kq + PQ * (a + NQ * b)] = ft[kq];
    }
  }
  __syncthreads();
  // ---- S5: thread (iq=a, jq=b), outputs k in [n0, n0+HN)
  {
    double x[NQ], ca[HN], cb[HN], cc[HN];
    if (active) {
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) x[kq] = lds_ld(&R0[kq + PQ * (a + NQ * b)]);
      contract_t_part<NQ, N, HN, false>(Bop, n0, x, ca);
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) x[kq] = lds_ld(&R1[kq + PQ * (a + NQ * b)]);
      contract_t_part<NQ, N, HN, false>(Bop, n0, x, cb);
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) x[kq] = lds_ld(&R2[kq + PQ * (a + NQ * b)]);
      contract_t_part<NQ, N, HN, false>(Gop, n0, x, cc);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int k = 0; k < HN; ++k) {  // [k][iq][jq], jq fastest
        R0[b + PQ * (a + NQ * (n0 + k))] = ca[k];
        R1[b + PQ * (a + NQ * (n0 + k))] = cb[k];
        R2[b + PQ * (a + NQ * (n0 + k))] = cc[k];
      }
    }
  }
  __syncthreads();
  // ---- S6: thread (iq=a, k=b), outputs j in [n0, n0+HN)
  {
    double x[NQ], ar[HN], bs[HN];
    const bool on6 = active && b < N;
    if (on6) {
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R0[jq + PQ * (a + NQ * b)]);
      contract_t_part<NQ, N, HN, false>(Bop, n0, x, ar);
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R1[jq + PQ * (a + NQ * b)]);
      contract_t_part<NQ, N, HN, false>(Gop, n0, x, bs);
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R2[jq + PQ * (a + NQ * b)]);
      contract_t_part<NQ, N, HN, true>(Bop, n0, x, bs);
    }
    __syncthreads();
    if (on6) {
#pragma unroll
      for (int j = 0; j < HN; ++j) {  // [k][j][iq]
        R0[a + PQ * (n0 + j + N * b)] = ar[j];
        R1[a + PQ * (n0 + j + N * b)] = bs[j];
      }
    }
  }
  __syncthreads();
  // ---- S7: thread (j=a, k=b), outputs i in [n0, n0+HN)
  if (active && a < N && b < N) {
    double x[NQ], o[HN];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) x[iq] = lds_ld(&R0[iq + PQ * (a + N * b)]);
    contract_t_part<NQ, N, HN, false>(Gop, n0, x, o);
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) x[iq] = lds_ld(&R1[iq + PQ * (a + N * b)]);
    contract_t_part<NQ, N, HN, true>(Bop, n0, x, o);
#pragma unroll
    for (int i = 0; i < HN; ++i) R2[n0 + i + PN * (a + N * b)] = o[i];
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int idx = te + h * PL; idx < N3; idx += 2 * PL) {
      const int i = idx % N, j = (idx / N) % N, k = idx / (N * N);
      Au[ns + idx] = lds_ld(&R2[i + PN * (j + N * k)]);
    }
  }
}

// ---------------------------------------------------------------------------
// Single-wavefront kernel with SOFTWARE-PIPELINED operator loads.  Scalar (SMEM) loads return out of
// order, so the only wait is lgkmcnt(0): if the next operator rows are requested before the current
// rows are waited for, the wait covers both and the load latency is fully exposed (measured: 43 % of
// wave cycles in s_waitcnt, profiles/r01_c_*).  Here every step (two operator rows, 2*NO FMAs) first
// forces the wait for ITS rows (an empty asm that consumes one SGPR of each row), THEN requests the next
// step's rows, THEN runs its FMAs: the request overlaps 16 FP64 FMAs (64 cycles).
// ---------------------------------------------------------------------------

template <int N, int NQ>
__global__ __launch_bounds__(64, 4) void stiffness_wave2_kernel(
    const double* __restrict__ u, double* __restrict__ Au, const double* __restrict__ metric,
    const int* __restrict__ ns_list, const int* __restrict__ qs_list, int n_bucket, const double* __restrict__ Bop,
    const double* __restrict__ Gop, const double* __restrict__ BopT, const double* __restrict__ GopT, int stagger) {
  if (stagger > 0 && ((blockIdx.x >> 8) & 1)) {  // optional phase stagger, see stiffness_wave_kernel
    for (int s_ = 0; s_ < stagger; ++s_) __builtin_amdgcn_s_sleep(16);
  }
  using C = WaveCfg<N, NQ>;
  constexpr int PL = C::PL, PN = C::PN, PQ = C::PQ, FS = C::FS;
  constexpr int N3 = N * N * N, NQ3 = NQ * NQ * NQ;
  extern __shared__ __attribute__((aligned(16))) double smem[];

  const int tid = threadIdx.x;
  const int slot = tid / PL;
  const int te = tid - slot * PL;
  const int a = te % NQ, b = te / NQ;
  const int ei = blockIdx.x * C::EPB + slot;
  const bool active = (slot < C::EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * C::LDS_PER_ELEM;
  double* R1 = R0 + FS;
  int ns = 0, qs = 0;
  if (active) {
    ns = ns_list[ei];
    qs = qs_list[ei];
    load_element_image<N, PL, PN>(R0, u + ns, te);
  }
  __syncthreads();
  stiffness_wave2_element<N, NQ, true>(R0, R1, metric, qs, active, a, b, Bop, Gop, BopT, GopT);
  if (active) {
    store_element_image<N, PL, PN>(Au + ns, R0, te);
  }
}

// ---------------------------------------------------------------------------
// EVEN-ODD single-wavefront kernel.  The 1-D operators on symmetric node sets are centro-symmetric
// (B: B[R-1-r][C-1-c] = B[r][c]) or centro-antisymmetric (G = B D).  With xe = x[c] + x[C-1-c], xo = x[c] - x[C-1-c]
// one apply is two half-size products  ye = Me xe, yo = Mo xo  (Me/Mo = (M[r][c] +- M[r][C-1-c]) / 2, r < R/2, c < C/2)
// and  y[r] = ye + yo, y[R-1-r] = +-(ye - yo):  R*C/2 FMAs + R + C adds instead of R*C FMAs (-25 % FP64 issue slots at
// N = 8, -31 % where B and G share the input) and half the scalar operator traffic.
// EO table of an operator: C/2 rows of R doubles, row c = [first half | second half] with
//   symmetric:      first = Me[0..R/2)[c] (multiplies xe), second = Mo[..][c] (multiplies xo)
//   antisymmetric:  first = Mo[..][c]     (multiplies xo), second = Me[..][c] (multiplies xe)
// so that in both cases  y[r] = a[r] + b[r],  y[R-1-r] = a[r] - b[r]  with (a | b) the accumulated row halves, and
// products of different operators can be summed in (a | b) form before the final butterfly.
// ---------------------------------------------------------------------------
// EBf / EGf: EO tables of y = B x / y = G x (N/2 rows of NQ);  EBb / EGb: of y = B^T x / y = G^T x (NQ/2 rows of N).
// N and NQ even.  Strides: affine (ns_stride >= 0) or from the lists.
// AFF = true: affine bucket -- the metric of node (a, b, kq) is rebuilt as (w_a w_b w_kq) * c[0..5] from the element's six
// constants (affine + 6 * element) instead of being streamed: 16 B/DoF of traffic instead of 64 (SURVEY.md section 8d "affine path")
template <int N, int NQ, bool AFF = false>
__global__ __launch_bounds__(64, 4) void stiffness_wave_eo_kernel(
    const double* __restrict__ u, double* __restrict__ Au, const double* __restrict__ metric,
    const int* __restrict__ ns_list, const int* __restrict__ qs_list, int n_bucket, const double* __restrict__ EBf,
    const double* __restrict__ EGf, const double* __restrict__ EBb, const double* __restrict__ EGb, int ns0, int ns_stride,
    int qs0, int qs_stride, const double* __restrict__ affine = nullptr, const double* __restrict__ wq = nullptr) {
  using C = WaveCfg<N, NQ>;
  constexpr int PL = C::PL, PN = C::PN, PQ = C::PQ, FS = C::FS;
  constexpr int N3 = N * N * N, NQ3 = NQ * NQ * NQ;
  constexpr int HN = N / 2, HQ = NQ / 2;
  extern __shared__ __attribute__((aligned(16))) double smem[];

  const int tid = threadIdx.x;
  const int slot = tid / PL;
  const int te = tid - slot * PL;
  const int a = te % NQ, b = te / NQ;
  const int ei = blockIdx.x * C::EPB + slot;
  const bool active = (slot < C::EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * C::LDS_PER_ELEM;
  double* R1 = R0 + FS;
  int ns = 0, qs = 0;
  if (active) {
    if (ns_stride >= 0) {
      ns = ns0 + ei * ns_stride;
      qs = qs0 + ei * qs_stride;
    } else {
      ns = ns_list[ei];
      qs = qs_list[ei];
    }
    if (C::EPB == 1) {
      ns = __builtin_amdgcn_readfirstlane(ns);
      qs = __builtin_amdgcn_readfirstlane(qs);
    }
    load_element_image<N, PL, PN>(R0, u + ns, te);
  }
  __syncthreads();
  stiffness_wave_eo_element<N, NQ, AFF, true>(R0, R1, metric, qs, ei, active, a, b, EBf, EGf, EBb, EGb, affine, wq);
  if (active) {
    store_element_image<N, PL, PN>(Au + ns, R0, te);
  }
}

// ---------------------------------------------------------------------------
// Mixed-degree plans: ONE launch for all buckets with deg_quad = deg <= 7 (one 64-lane workgroup per wave_eo work unit, whatever the
// degree).  A plan with p = 3 ... 9 scattered over its elements (BASELINE config 4's shape) otherwise pays one launch per bucket, each
// far too small to fill the chip; side-by-side launches on forked streams cost more in event waits than they overlap.
// The workgroup looks its bucket up (wave-uniform) and runs that degree's body -- the same code as stiffness_wave_eo_kernel.
// ---------------------------------------------------------------------------
struct WaveEoMulti {
  static constexpr int MAXB = 7;
  int n = 0;
  int wg_end[MAXB] = {};       // exclusive prefix of the buckets' workgroup counts
  int N[MAXB] = {};
  int n_elem[MAXB] = {};
  int elem_offset[MAXB] = {};  // into the plan's bucket-ordered ns / qs lists (and 6 doubles per element of the affine constants)
  const double* EBf[MAXB] = {};
  const double* EGf[MAXB] = {};
  const double* EBb[MAXB] = {};
  const double* EGb[MAXB] = {};
  const double* wq[MAXB] = {};
};

template <int N, bool AFF>
__device__ __forceinline__ void wave_eo_multi_body(double* smem, int wg, const double* __restrict__ u, double* __restrict__ Au,
                                                   const double* __restrict__ metric, const int* __restrict__ ns_list,
                                                   const int* __restrict__ qs_list, int n_bucket, const double* __restrict__ EBf,
                                                   const double* __restrict__ EGf, const double* __restrict__ EBb,
                                                   const double* __restrict__ EGb, const double* __restrict__ affine,
                                                   const double* __restrict__ wq) {
  using C = WaveCfg<N, N>;
  constexpr int PL = C::PL, PN = C::PN, FS = C::FS;
  const int tid = threadIdx.x;
  const int slot = tid / PL;
  const int te = tid - slot * PL;
  const int a = te % N, b = te / N;
  const int ei = wg * C::EPB + slot;
  const bool active = (slot < C::EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * C::LDS_PER_ELEM;
  double* R1 = R0 + FS;
  int ns = 0, qs = 0;
  if (active) {
    ns = ns_list[ei];
    qs = qs_list[ei];
    if (C::EPB == 1) {
      ns = __builtin_amdgcn_readfirstlane(ns);
      qs = __builtin_amdgcn_readfirstlane(qs);
    }
    load_element_image<N, PL, PN>(R0, u + ns, te);
  }
  __syncthreads();
  stiffness_wave_eo_element<N, N, AFF, true>(R0, R1, metric, qs, ei, active, a, b, EBf, EGf, EBb, EGb, affine, wq);
  if (active) {
    store_element_image<N, PL, PN>(Au + ns, R0, te);
  }
}

template <bool AFF>
__global__ __launch_bounds__(64, 4) void stiffness_wave_eo_multi_kernel(const double* __restrict__ u, double* __restrict__ Au,
                                                                        const double* __restrict__ metric,
                                                                        const int* __restrict__ ns_list_all,
                                                                        const int* __restrict__ qs_list_all,
                                                                        const double* __restrict__ affine_all, WaveEoMulti A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int blk = blockIdx.x;
  int bi = 0;
  while (bi + 1 < A.n && blk >= A.wg_end[bi]) ++bi;   // wave-uniform: a handful of scalar compares
  const int wg = blk - (bi > 0 ? A.wg_end[bi - 1] : 0);
  const int off = A.elem_offset[bi], nb = A.n_elem[bi];
  const double* EBf = A.EBf[bi];
  const double* EGf = A.EGf[bi];
  const double* EBb = A.EBb[bi];
  const double* EGb = A.EGb[bi];
  const double* wq = A.wq[bi];
  const double* aff = AFF ? affine_all + (size_t)6 * off : nullptr;
#define D4EST_CASE(N_)                                                                                                            \
  case N_:                                                                                                                        \
    wave_eo_multi_body<N_, AFF>(smem, wg, u, Au, metric, ns_list_all + off, qs_list_all + off, nb, EBf, EGf, EBb, EGb, aff, wq);   \
    break;
  switch (A.N[bi]) {
    D4EST_CASE(2) D4EST_CASE(3) D4EST_CASE(4) D4EST_CASE(5) D4EST_CASE(6) D4EST_CASE(7) D4EST_CASE(8)
    default: break;
  }
#undef D4EST_CASE
}

// Mixed-degree plans, p = 8 ... 12: the buckets whose multi-wave kernels have the same workgroup size (N = 9, 10, 11: 128 threads -- N = 11
// is 121 lines --; N = 12, 13: 192) in ONE launch, like stiffness_wave_eo_multi_kernel for p <= 7 -- a workgroup looks its bucket up and runs that degree's
// body (stiffness_wave_kernel's: collocated-gradient form, streamed metric).  Small buckets (585 elements each on the p = 3 ... 9 mesh of
// the bench) no longer queue behind each other: 13 + 15 us -> one launch.  WaveEoMulti: EGb / EGf carry the tables of Dq / Dq^T.
template <int N, bool NT>
__device__ __forceinline__ void mw_multi_body(double* smem, int ei, const double* __restrict__ u, double* __restrict__ Au,
                                              const double* __restrict__ metric, const int* __restrict__ ns_list, const int* __restrict__ qs_list,
                                              int n_bucket, const double* __restrict__ EBb, const double* __restrict__ EBf,
                                              const double* __restrict__ EDq, const double* __restrict__ EDqT) {
  using C = WaveCfg<N, N>;
  constexpr int PL = C::PL, PN = C::PN;
  const int te = threadIdx.x;
  const int a = te % N, b = te / N;
  const bool active = (te < PL) && (ei < n_bucket);
  double* R0 = smem;
  double* R1 = R0 + C::FS;
  int ns = 0, qs = 0;
  if (ei < n_bucket) {
    ns = __builtin_amdgcn_readfirstlane(ns_list[ei]);
    qs = __builtin_amdgcn_readfirstlane(qs_list[ei]);
  }
  if (active) load_element_image<N, PL, PN>(R0, u + ns, te);
  stiffness_mw_element_cg<N, false, false, NT>(R0, R1, metric, qs, ei, active, te, a, b, EBb, EBf, EDq, EDqT, nullptr, nullptr);
  if (active) store_element_image<N, PL, PN, NT>(Au + ns, R0, te);
}

template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS, D4EST_HIP_MW_WAVES) void stiffness_mw_multi_kernel(const double* __restrict__ u, double* __restrict__ Au,
                                                                                        const double* __restrict__ metric,
                                                                                        const int* __restrict__ ns_list_all,
                                                                                        const int* __restrict__ qs_list_all, WaveEoMulti A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int blk = blockIdx.x;
  int bi = 0;
  while (bi + 1 < A.n && blk >= A.wg_end[bi]) ++bi;   // wave-uniform
  const int ei = blk - (bi > 0 ? A.wg_end[bi - 1] : 0);
  const int off = A.elem_offset[bi], nb = A.n_elem[bi];
  const double* EBf = A.EBf[bi];
  const double* EDqT = A.EGf[bi];
  const double* EBb = A.EBb[bi];
  const double* EDq = A.EGb[bi];
#define D4EST_CASE(N_)                                                                                                  \
  case N_:                                                                                                              \
    if constexpr (WaveCfg<N_, N_>::THREADS == THREADS)                                                                  \
      mw_multi_body<N_, NT>(smem, ei, u, Au, metric, ns_list_all + off, qs_list_all + off, nb, EBb, EBf, EDq, EDqT);    \
    break;
  switch (A.N[bi]) {
    D4EST_CASE(9) D4EST_CASE(10) D4EST_CASE(11) D4EST_CASE(12) D4EST_CASE(13)
    default: break;
  }
#undef D4EST_CASE
}


// Mixed-degree plans, the two launches above in ONE: the p <= 7 buckets (one wavefront per work unit, two independent units per
// 128-thread workgroup, wave-level hand-offs) and the 128-thread multi-wave buckets (N = 9, 10, 11) -- on config 4's degree range p = 3 ... 9
// every bucket of the plan.  Both kinds are short, latency-structured kernels at these sizes (13 - 18 us each for 585-element buckets): side by
// side in one launch they cost the longer of the two (general path; D4EST_HIP_STIFFNESS_SPLIT_LAUNCH=1 keeps the two launches).
template <bool NT>
__global__ __launch_bounds__(128, D4EST_HIP_MW_WAVES) void stiffness_all_multi_kernel(const double* __restrict__ u, double* __restrict__ Au,
                                                                                    const double* __restrict__ metric,
                                                                                    const int* __restrict__ ns_list_all,
                                                                                    const int* __restrict__ qs_list_all, WaveEoMulti A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int blk = blockIdx.x;
  int bi = 0;
  while (bi + 1 < A.n && blk >= A.wg_end[bi]) ++bi;   // wave-uniform
  const int wgb = blk - (bi > 0 ? A.wg_end[bi - 1] : 0);
  const int off = A.elem_offset[bi], nb = A.n_elem[bi];
  const double* EBf = A.EBf[bi];
  const double* EGf = A.EGf[bi];   // (multi-wave buckets: Dq^T)
  const double* EBb = A.EBb[bi];
  const double* EGb = A.EGb[bi];   // (multi-wave buckets: Dq)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define D4EST_CASE_EO(N_)                                                                                                              \
  case N_: {                                                                                                                           \
    using C = WaveCfg<N_, N_>;                                                                                                         \
    constexpr int PL = C::PL, PN = C::PN, FS = C::FS;                                                                                  \
    const int tid = threadIdx.x & 63, slot = tid / PL, te = tid - slot * PL;                                                           \
    const int ei = (2 * wgb + wv) * C::EPB + slot;                                                                                     \
    const bool active = (slot < C::EPB) && (ei < nb);                                                                                  \
    double* R0 = smem + wv * (C::EPB * C::LDS_PER_ELEM) + (active ? slot : 0) * C::LDS_PER_ELEM;                                       \
    double* R1 = R0 + FS;                                                                                                              \
    int ns = 0, qs = 0;                                                                                                                \
    if (active) {                                                                                                                      \
      ns = ns_list_all[off + ei];                                                                                                      \
      qs = qs_list_all[off + ei];                                                                                                      \
      if (C::EPB == 1) { ns = __builtin_amdgcn_readfirstlane(ns); qs = __builtin_amdgcn_readfirstlane(qs); }                           \
      load_element_image<N_, PL, PN>(R0, u + ns, te);                                                                                  \
    }                                                                                                                                  \
    wave_lds_fence();                                                                                                                  \
    stiffness_wave_eo_element<N_, N_, false, false>(R0, R1, metric, qs, ei, active, te % N_, te / N_, EBf, EGf, EBb, EGb, nullptr, nullptr); \
    if (active) store_element_image<N_, PL, PN>(Au + ns, R0, te);                                                                      \
  } break;
#define D4EST_CASE_MW(N_)                                                                                                              \
  case N_:                                                                                                                             \
    mw_multi_body<N_, NT>(smem, wgb, u, Au, metric, ns_list_all + off, qs_list_all + off, nb, EBb, EBf, EGb, EGf);                     \
    break;
  switch (A.N[bi]) {
    D4EST_CASE_EO(2) D4EST_CASE_EO(3) D4EST_CASE_EO(4) D4EST_CASE_EO(5) D4EST_CASE_EO(6) D4EST_CASE_EO(7) D4EST_CASE_EO(8)
    D4EST_CASE_MW(9) D4EST_CASE_MW(10) D4EST_CASE_MW(11)
    default: break;
  }
#undef D4EST_CASE_EO
#undef D4EST_CASE_MW
}

// ---------------------------------------------------------------------------
// N = NQ = 16 (p = 15) on the FP64 matrix cores (N = 13 ... 15 run too, operators zero-padded to 16, but the padding and the idle
// waves eat the gain -- measured 32 / 47 / 36 GDoF/s against 41 / 46 / 34 of the vector-ALU kernel -- so only p = 15 selects it by itself): the whole sum-factorised apply as chains of v_mfma_f64_16x16x4
// (D[16x16] += A[16x4] B[4x16]; lane l = (q = l >> 4, c = l & 15) holds A[row c][k q], B[k q][col c] and, in register v,
// D[row 4 v + q][col c]).  Register v of a result tile IS the B operand of k-step v of a following product that sums over
// the tile's ROW index, so every second contraction takes its input straight from the accumulators:
//   slab k:  P  = X_k^T B^T      (A = the element's slab, read from global memory as [row j][k i]; B = operator)       rows j
//            Z  = B P            (A = operator, B = P's registers)                                                  rows j'
//   LDS round trip ([k][j'][i'], rows padded to 17 doubles, slabs to 272: conflict-free both ways)
//   t:       T  = B Z[:, column tile]  -> metric multiply in registers (all three gradient components of a node sit in the
//            same lane and register) -> W = B^T F (B = F's registers)                                               rows k
//   LDS round trip, in place (a wave overwrites exactly the column tiles it read)
//   slab k:  E  = W_k B           (A = W read from the LDS as [row j'][k i'], B = operator)                          rows j'
//            R  = B^T E           (B = E's registers)  -> Au[i][j][k] from the accumulators, 128-byte segments
// 1024 MFMAs per element, no scalar operator feed, three workgroup barriers; the 16 operator fragments (B, G, natural and
// transposed) live in 32 VGPRs per lane for the whole kernel.  One element per 512-thread workgroup (8 waves: two slabs and
// two column tiles each), 102 KB of LDS, persistent over the bucket with the next element's slabs requested as soon as the
// current ones are consumed.  Why the matrix cores here and not below p = 15: DESIGN.md section 3.
// ---------------------------------------------------------------------------
typedef double mfma_d4v __attribute__((ext_vector_type(4)));
#define D4_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)

struct Mfma16Cfg {
  static constexpr int JS = 17, KS = 16 * 17, FS = 16 * KS;      // row / slab / field strides of the LDS image (doubles)
  static constexpr size_t LDS_BYTES = (size_t)3 * FS * sizeof(double);
};

// AFF: affine bucket -- the metric of node (i, j, k) is (w_i w_j w_k) * c[0..5] with the element's six constants (affine + 6 * element)
template <int N, bool AFF = false>
__global__ __launch_bounds__(512, 1) void stiffness_mfma16_kernel(const double* __restrict__ u, double* __restrict__ Au,
                                                                   const double* __restrict__ metric, const int* __restrict__ ns_list,
                                                                   const int* __restrict__ qs_list, int n_bucket,
                                                                   const double* __restrict__ Bop, const double* __restrict__ Gop,
                                                                   const double* __restrict__ affine = nullptr,
                                                                   const double* __restrict__ wq = nullptr, int stream = 0) {
  constexpr int JS = Mfma16Cfg::JS, KS = Mfma16Cfg::KS, FS = Mfma16Cfg::FS;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15;
  // operator fragments: natural = Op[c][4 s + q], transposed = Op[4 s + q][c]  (Op row-major [quadrature node][Lobatto node])
  double Bn[4], Gn[4], Bt[4], Gt[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bool in = (c < N) && (4 * s + q < N);     // N < 16: operators zero-padded to 16 x 16
    Bn[s] = in ? Bop[c * N + 4 * s + q] : 0.0;
    Gn[s] = in ? Gop[c * N + 4 * s + q] : 0.0;
    Bt[s] = in ? Bop[(4 * s + q) * N + c] : 0.0;
    Gt[s] = in ? Gop[(4 * s + q) * N + c] : 0.0;
  }
  const mfma_d4v zero = {0.0, 0.0, 0.0, 0.0};
  // the wave's two slabs of u, as A operands [row j = c][k i = 4 s + q]
  double xa[2][4];
  int e = blockIdx.x;
  if (e < n_bucket) {
    const double* ue = u + ns_list[e];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        xa[h][s] = (c < N && 4 * s + q < N && 2 * wave + h < N) ? ue[(4 * s + q) + N * c + N * N * (2 * wave + h)] : 0.0;
  }
  for (; e < n_bucket; e += gridDim.x) {
    const int ns = ns_list[e];
    const double* me = metric + (size_t)6 * (AFF ? 0 : qs_list[e]);
    // metric of the wave's first column tile (j' = 2 wave): requested now, used after the forward slab stage
    double mt[6][4];
    auto tile_metric = [&](int ct, double (*dst)[4]) {
      if constexpr (AFF) {
        const double* __restrict__ cc = affine + (size_t)6 * e;
        const double wcj = (c < N && ct < N) ? wq[c] * wq[ct] : 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double w3 = (4 * v + q < N) ? wcj * wq[4 * v + q] : 0.0;
#pragma unroll
          for (int m = 0; m < 6; ++m) dst[m][v] = w3 * cc[m];
        }
      } else {
        with_ld(stream != 0, [&](auto ld) {
#pragma unroll
          for (int m = 0; m < 6; ++m)
#pragma unroll
            for (int v = 0; v < 4; ++v)
              dst[m][v] = (c < N && ct < N && 4 * v + q < N) ? ld(&me[m * (N * N * N) + c + N * ct + N * N * (4 * v + q)]) : 0.0;
        });
      }
    };
    tile_metric(2 * wave, mt);
    // ---- forward slab stage: two slabs, interleaved chains
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 2 * wave + h;
      if (k >= N) continue;
      mfma_d4v pb = zero, pg = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        pb = D4_MFMA(xa[h][s], Bn[s], pb);
        pg = D4_MFMA(xa[h][s], Gn[s], pg);
      }
      mfma_d4v za = zero, zb = zero, zc = zero;   // za = Bs Br X (-> d/dt), zb = Bs Gr X (-> d/dr), zc = Gs Br X (-> d/ds)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        za = D4_MFMA(Bn[s], pb[s], za);
        zb = D4_MFMA(Bn[s], pg[s], zb);
        zc = D4_MFMA(Gn[s], pb[s], zc);
      }
      double* z = smem + k * KS + c;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        z[JS * (4 * v + q)] = zb[v];             // field 0: r
        z[FS + JS * (4 * v + q)] = zc[v];        // field 1: s
        z[2 * FS + JS * (4 * v + q)] = za[v];    // field 2: t
      }
    }
    // xa is dead from here: the next element's slabs travel during the t and backward stages
    {
      const int e_next = e + (int)gridDim.x;
      if (e_next < n_bucket) {
        const double* un = u + ns_list[e_next];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            xa[h][s] = (c < N && 4 * s + q < N && 2 * wave + h < N) ? un[(4 * s + q) + N * c + N * N * (2 * wave + h)] : 0.0;
      }
    }
    __syncthreads();
    // ---- t stage on the wave's two column tiles (j' = 2 wave, 2 wave + 1): forward, metric, backward, W in place of Z
    double mt2[6][4];   // the second tile's metric: requested before the first tile's products
    tile_metric(2 * wave + 1, mt2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ct = 2 * wave + h;
      if (ct >= N) continue;
      const double* z = smem + JS * ct + c;
      mfma_d4v tr = zero, ts = zero, tt = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int off = (4 * s + q) * KS;
        tr = D4_MFMA(Bn[s], z[off], tr);
        ts = D4_MFMA(Bn[s], z[FS + off], ts);
        tt = D4_MFMA(Gn[s], z[2 * FS + off], tt);
      }
      mfma_d4v fr, fs, ft;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        fr[v] = mt[0][v] * tr[v] + mt[1][v] * ts[v] + mt[2][v] * tt[v];
        fs[v] = mt[1][v] * tr[v] + mt[3][v] * ts[v] + mt[4][v] * tt[v];
        ft[v] = mt[2][v] * tr[v] + mt[4][v] * ts[v] + mt[5][v] * tt[v];
      }
      if (h == 0) {
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int v = 0; v < 4; ++v) mt[m][v] = mt2[m][v];
      }
      mfma_d4v wr = zero, ws = zero, wt = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wr = D4_MFMA(Bt[s], fr[s], wr);
        ws = D4_MFMA(Bt[s], fs[s], ws);
        wt = D4_MFMA(Gt[s], ft[s], wt);
      }
      double* w = smem + JS * ct + c;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        w[(4 * v + q) * KS] = wr[v];
        w[FS + (4 * v + q) * KS] = ws[v];
        w[2 * FS + (4 * v + q) * KS] = wt[v];
      }
    }
    __syncthreads();
    // ---- backward slab stage
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 2 * wave + h;
      if (k >= N) continue;
      const double* w = smem + k * KS + JS * c + q;      // A operand [row j' = c][k i' = 4 s + q]
      mfma_d4v e1 = zero, e2 = zero, e3 = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        e1 = D4_MFMA(w[4 * s], Gt[s], e1);               // r term: G^T over i'
        e2 = D4_MFMA(w[FS + 4 * s], Bt[s], e2);          // s term: B^T over i', then G^T over j'
        e3 = D4_MFMA(w[2 * FS + 4 * s], Bt[s], e3);      // t term: B^T over i'
      }
      mfma_d4v e13 = e1 + e3;
      mfma_d4v r1 = zero, r2 = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        r1 = D4_MFMA(Bt[s], e13[s], r1);
        r2 = D4_MFMA(Gt[s], e2[s], r2);
      }
      double* out = Au + ns + c + N * N * k;
      if (stream) {
        D4EST_HIP_STREAM_FENCE();
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (c < N && 4 * v + q < N) __builtin_nontemporal_store(r1[v] + r2[v], &out[N * (4 * v + q)]);
        D4EST_HIP_STREAM_FENCE();
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (c < N && 4 * v + q < N) out[N * (4 * v + q)] = r1[v] + r2[v];
      }
    }
    __syncthreads();   // the next element's forward stage overwrites the image
  }
}

// ---------------------------------------------------------------------------
// mass-like applies (one field):
//   MODE 0: out = V^T (W J) V in          (mass)            in: nodal, out: nodal
//   MODE 1: out = V^T (W J) in_quad       (galerkin)        in: quad,  out: nodal
//   MODE 2: out_quad = V in               (interpolate)     in: nodal, out: quad  (also any square tensor apply A(x)A(x)A)
//   MODE 3: out = V^T (W J c) V in        (weighted mass: d4est_quadrature_apply_fofufofvlilj with the coefficient
//                                          f(u) f(v) given at the quadrature nodes, d4est_quadrature.c:593-774)
//   MODE 4: out = V^-1 (W J)^-1 V^-T in   (inverse mass, d4est_quadrature.c:1222-1331; Bop = V^-T, BopT = V^-1)
// ---------------------------------------------------------------------------
// EO = true: Bop / BopT are the even-odd tables of B^T / B (the symmetric interpolation), see apply_eo
template <int N, int NQ, int MODE, bool EO>
__device__ __forceinline__ void mass_like_body(
    double* smem, int wg, const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ Jq,
    const int* __restrict__ ns_list, const int* __restrict__ qs_list,
    int n_bucket, const double* __restrict__ Bop, const double* __restrict__ BopT, const double* __restrict__ wq,
    const double* __restrict__ coeff) {
  using C = WaveCfg<N, NQ>;
  constexpr int PL = C::PL, PN = C::PN, PQ = C::PQ;
  constexpr int N3 = N * N * N;

  const int tid = threadIdx.x;
  const int slot = tid / PL;
  const int te = tid - slot * PL;
  const int a = te % NQ, b = te / NQ;
  const int ei = wg * C::EPB + slot;
  const bool active = (slot < C::EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * C::LDS_PER_ELEM;
  double* R1 = R0 + C::FS;

  int ns = 0, qs = 0;
  if (active) {
    ns = ns_list[ei];
    qs = qs_list[ei];
  }

  double g[NQ];  // values at quadrature nodes along kq for thread (iq=a, jq=b)
  if (MODE != 1) {
    if (active) {
      load_element_image<N, PL, PN>(R0, in + ns, te);
    }
    __syncthreads();
    if (active && a < N && b < N) {  // r
      double x[N], y[NQ];
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = lds_ld(&R0[i + PN * (a + N * b)]);
      fwd<N, NQ, EO, false>(BopT, x, y);
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) R1[a + PN * (iq + NQ * b)] = y[iq];
    }
    __syncthreads();
    if (active && b < N) {  // s
      double x[N], y[NQ];
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = lds_ld(&R1[j + PN * (a + NQ * b)]);
      fwd<N, NQ, EO, false>(BopT, x, y);
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) R0[b + PN * (a + NQ * jq)] = y[jq];
    }
    __syncthreads();
    if (active) {  // t
      double x[N];
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] = lds_ld(&R0[k + PN * (a + NQ * b)]);
      fwd<N, NQ, EO, false>(BopT, x, g);
    }
  } else if (active) {
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) g[kq] = in[qs + a + NQ * (b + NQ * kq)];
  }

  if (MODE == 2) {
    if (active) {
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) out[qs + a + NQ * (b + NQ * kq)] = g[kq];
    }
    return;
  }

  double c[N];
  if (active) {
    const double wab = wq[a] * wq[b];
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {
      const int q = qs + a + NQ * (b + NQ * kq);
      double sc = (wq[kq] * wab) * Jq[q];
      if (MODE == 3) sc *= coeff[q];
      if (MODE == 4) g[kq] = (1. / sc) * g[kq];  // d4est_kron_oneover_vec_o_vec_o_vec_dot_oneover_x_dot_y (d4est_kron.h:387-397)
      else g[kq] *= sc;
    }
    bwd<NQ, N, EO, false, false>(Bop, g, c);
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int k = 0; k < N; ++k) R0[b + PQ * (a + NQ * k)] = c[k];
  }
  __syncthreads();
  if (active && b < N) {
    double x[NQ], y[N];
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R0[jq + PQ * (a + NQ * b)]);
    bwd<NQ, N, EO, false, false>(Bop, x, y);
#pragma unroll
    for (int j = 0; j < N; ++j) R1[a + PQ * (j + N * b)] = y[j];
  }
  __syncthreads();
  if (active && a < N && b < N) {
    double x[NQ], o[N];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) x[iq] = lds_ld(&R1[iq + PQ * (a + N * b)]);
    bwd<NQ, N, EO, false, false>(Bop, x, o);
#pragma unroll
    for (int i = 0; i < N; ++i) R0[i + PN * (a + N * b)] = o[i];
  }
  __syncthreads();
  if (active) {
    store_element_image<N, PL, PN>(out + ns, R0, te);
  }
}

// ---------------------------------------------------------------------------
// generic path: any (N, NQ) with runtime sizes, one 256-thread block per element,
// staged through a per-block global scratch (L2 resident).  Correctness fallback
// for degree pairs without a compiled fast kernel.
// ---------------------------------------------------------------------------
struct GenDims { int n[3]; };

// out = (op applied along `dir`) in; op(o,c) = op[o*so + c*sc]; in has dims d (x fastest)
__device__ void gen_apply(const double* __restrict__ op, int rows, int so, int sc, int dir, const double* in, GenDims d,
                          double* out, bool acc) {
  GenDims od = d;
  const int cols = d.n[dir];
  od.n[dir] = rows;
  const int total = od.n[0] * od.n[1] * od.n[2];
  int in_stride = 1;
  for (int q = 0; q < dir; ++q) in_stride *= d.n[q];
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    int c[3];
    c[0] = idx % od.n[0];
    c[1] = (idx / od.n[0]) % od.n[1];
    c[2] = idx / (od.n[0] * od.n[1]);
    const int o = c[dir];
    c[dir] = 0;
    const int base = c[0] + d.n[0] * (c[1] + d.n[1] * c[2]);
    double s = 0.0;
    for (int q = 0; q < cols; ++q) s = fma(op[o * so + q * sc], in[base + q * in_stride], s);
    out[idx] = acc ? out[idx] + s : s;
  }
  __syncthreads();
}

template <int N, int NQ, int MODE, bool EO = false>
__global__ __launch_bounds__((WaveCfg<N, NQ>::THREADS)) void mass_like_kernel(
    const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ Jq,
    const int* __restrict__ ns_list, const int* __restrict__ qs_list,
    int n_bucket, const double* __restrict__ Bop, const double* __restrict__ BopT, const double* __restrict__ wq,
    const double* __restrict__ coeff) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  mass_like_body<N, NQ, MODE, EO>(smem, blockIdx.x, in, out, Jq, ns_list, qs_list, n_bucket, Bop, BopT, wq, coeff);
}

// Mixed-degree plans: the mass (MODE 0) / weighted mass (MODE 3) applies of all buckets with deg_quad = deg <= 7 in ONE launch
// (see stiffness_wave_eo_multi_kernel; WaveEoMulti carries EBb / EBf as the two even-odd tables of B)
template <int MODE>
__global__ __launch_bounds__(64) void mass_like_multi_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                             const double* __restrict__ Jq, const int* __restrict__ ns_list_all,
                                                             const int* __restrict__ qs_list_all, const double* __restrict__ coeff,
                                                             WaveEoMulti A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int blk = blockIdx.x;
  int bi = 0;
  while (bi + 1 < A.n && blk >= A.wg_end[bi]) ++bi;
  const int wg = blk - (bi > 0 ? A.wg_end[bi - 1] : 0);
  const int off = A.elem_offset[bi], nb = A.n_elem[bi];
  const double* EBb = A.EBb[bi];
  const double* EBf = A.EBf[bi];
  const double* wq = A.wq[bi];
#define D4EST_CASE(N_)                                                                                                                 \
  case N_:                                                                                                                             \
    mass_like_body<N_, N_, MODE, true>(smem, wg, in, out, Jq, ns_list_all + off, qs_list_all + off, nb, EBb, EBf, wq, coeff);          \
    break;
  switch (A.N[bi]) {
    D4EST_CASE(2) D4EST_CASE(3) D4EST_CASE(4) D4EST_CASE(5) D4EST_CASE(6) D4EST_CASE(7) D4EST_CASE(8)
    default: break;
  }
#undef D4EST_CASE

}

__global__ __launch_bounds__(256) void generic_volume_kernel(
    int mode /* 0 mass, 1 galerkin, 2 interp / square tensor apply, 3 stiffness, 4 weighted mass, 5 inverse mass */, const double* __restrict__ in, double* __restrict__ out,
    const double* __restrict__ metric, const double* __restrict__ Jq, const int* __restrict__ ns_list,
    const int* __restrict__ qs_list, int n_bucket, int N, int NQ,
    const double* __restrict__ Bop, const double* __restrict__ Gop, const double* __restrict__ wq, double* scratch,
    size_t scratch_per_block, const double* __restrict__ coeff) {
  const int NM = N > NQ ? N : NQ;
  const size_t A = (size_t)NM * NM * NM;
  double* s = scratch + (size_t)blockIdx.x * scratch_per_block;
  double *t0 = s, *t1 = s + A, *t2 = s + 2 * A, *t3 = s + 3 * A, *t4 = s + 4 * A, *t5 = s + 5 * A, *t6 = s + 6 * A, *t7 = s + 7 * A;
  const int NQ3 = NQ * NQ * NQ;
  for (int ei = blockIdx.x; ei < n_bucket; ei += gridDim.x) {
    const int ns = ns_list[ei], qs = qs_list[ei];
    const double* ue = in + ns;
    GenDims dn = {{N, N, N}};
    if (mode == 3) {
      // forward: gr = (B,B,G) u, gs = (B,G,B) u, gt = (G,B,B) u   (order t,s,r | here listed as z,y,x operators)
      gen_apply(Bop, NQ, N, 1, 0, ue, dn, t0, false);          // B_r u
      gen_apply(Gop, NQ, N, 1, 0, ue, dn, t1, false);          // G_r u
      GenDims d1 = {{NQ, N, N}};
      gen_apply(Bop, NQ, N, 1, 1, t0, d1, t2, false);          // B_s B_r u
      gen_apply(Gop, NQ, N, 1, 1, t0, d1, t3, false);          // G_s B_r u
      gen_apply(Bop, NQ, N, 1, 1, t1, d1, t4, false);          // B_s G_r u
      GenDims d2 = {{NQ, NQ, N}};
      gen_apply(Bop, NQ, N, 1, 2, t4, d2, t5, false);          // gr
      gen_apply(Bop, NQ, N, 1, 2, t3, d2, t6, false);          // gs
      gen_apply(Gop, NQ, N, 1, 2, t2, d2, t7, false);          // gt
      const double* m = metric + (size_t)6 * qs;
      for (int q = threadIdx.x; q < NQ3; q += blockDim.x) {
        const double r = t5[q], sv = t6[q], t = t7[q];
        const double m0 = m[q], m1 = m[NQ3 + q], m2 = m[2 * NQ3 + q], m3 = m[3 * NQ3 + q], m4 = m[4 * NQ3 + q], m5 = m[5 * NQ3 + q];
        t5[q] = m0 * r + m1 * sv + m2 * t;
        t6[q] = m1 * r + m3 * sv + m4 * t;
        t7[q] = m2 * r + m4 * sv + m5 * t;
      }
      __syncthreads();
      GenDims dq = {{NQ, NQ, NQ}};
      gen_apply(Bop, N, 1, N, 2, t5, dq, t0, false);           // B_t^T fr
      gen_apply(Bop, N, 1, N, 2, t6, dq, t1, false);           // B_t^T fs
      gen_apply(Gop, N, 1, N, 2, t7, dq, t2, false);           // G_t^T ft
      GenDims d3 = {{NQ, NQ, N}};
      gen_apply(Bop, N, 1, N, 1, t0, d3, t3, false);           // B_s^T .
      gen_apply(Gop, N, 1, N, 1, t1, d3, t4, false);           // G_s^T .
      gen_apply(Bop, N, 1, N, 1, t2, d3, t4, true);            // + B_s^T .
      GenDims d4 = {{NQ, N, N}};
      gen_apply(Gop, N, 1, N, 0, t3, d4, out + ns, false);
      gen_apply(Bop, N, 1, N, 0, t4, d4, out + ns, true);
    } else {
      const double* gq;
      if (mode != 1) {
        gen_apply(Bop, NQ, N, 1, 0, ue, dn, t0, false);
        GenDims d1 = {{NQ, N, N}};
        gen_apply(Bop, NQ, N, 1, 1, t0, d1, t1, false);
        GenDims d2 = {{NQ, NQ, N}};
        if (mode == 2) {
          gen_apply(Bop, NQ, N, 1, 2, t1, d2, out + qs, false);
          continue;
        }
        gen_apply(Bop, NQ, N, 1, 2, t1, d2, t2, false);
        gq = t2;
      } else {
        gq = in + qs;
      }
      for (int q = threadIdx.x; q < NQ3; q += blockDim.x) {
        const int iq = q % NQ, jq = (q / NQ) % NQ, kq = q / (NQ * NQ);
        double sc = (wq[kq] * (wq[iq] * wq[jq])) * Jq[qs + q];
        if (mode == 4) sc *= coeff[qs + q];           // weighted mass
        t3[q] = (mode == 5) ? (1. / sc) * gq[q] : gq[q] * sc;  // 5: inverse mass
      }
      __syncthreads();
      GenDims dq = {{NQ, NQ, NQ}};
      gen_apply(Bop, N, 1, N, 2, t3, dq, t0, false);
      GenDims d3 = {{NQ, NQ, N}};
      gen_apply(Bop, N, 1, N, 1, t0, d3, t1, false);
      GenDims d4 = {{NQ, N, N}};
      gen_apply(Bop, N, 1, N, 0, t1, d4, out + ns, false);
    }
  }
}

// ---------------------------------------------------------------------------
// collocation derivatives  dudr_i = D_i u   (3 outputs), one element per NxN threads
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__((WaveCfg<N, N>::THREADS)) void dudr_kernel(
    const double* __restrict__ u, double* __restrict__ d0, double* __restrict__ d1, double* __restrict__ d2,
    const int* __restrict__ ns_list, int n_bucket, const double* __restrict__ DopT) {
  using C = WaveCfg<N, N>;
  constexpr int PL = C::PL, PN = C::PN, N3 = N * N * N;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int slot = tid / PL;
  const int te = tid - slot * PL;
  const int a = te % N, b = te / N;
  const int ei = blockIdx.x * C::EPB + slot;
  const bool active = (slot < C::EPB) && (ei < n_bucket);
  double* R0 = smem + (active ? slot : 0) * C::LDS_PER_ELEM;
  double* R1 = R0 + C::FS;
  int ns = 0;
  if (active) ns = ns_list[ei];
  if (active) {
    load_element_image<N, PL, PN>(R0, u + ns, te);
  }
  __syncthreads();
  // direction t (dir 2): thread (i=a, j=b), column along k; output coalesced directly
  if (active) {
    double x[N], y[N];
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = lds_ld(&R0[a + PN * (b + N * k)]);
    contract_n<N, N>(DopT, x, y);
#pragma unroll
    for (int k = 0; k < N; ++k) d2[ns + a + N * (b + N * k)] = y[k];
    // direction s (dir 1): thread (i=a, k=b), column along j
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = lds_ld(&R0[a + PN * (j + N * b)]);
    contract_n<N, N>(DopT, x, y);
#pragma unroll
    for (int j = 0; j < N; ++j) d1[ns + a + N * (j + N * b)] = y[j];
    // direction r (dir 0): thread (j=a, k=b), column along i -> stage through LDS for a coalesced store
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = lds_ld(&R0[i + PN * (a + N * b)]);
    contract_n<N, N>(DopT, x, y);
#pragma unroll
    for (int i = 0; i < N; ++i) R1[i + PN * (a + N * b)] = y[i];
  }
  __syncthreads();
  if (active) {
    store_element_image<N, PL, PN>(d0 + ns, R1, te);
  }
}

__global__ __launch_bounds__(256) void generic_dudr_kernel(const double* __restrict__ u, double* __restrict__ d0,
                                                           double* __restrict__ d1, double* __restrict__ d2,
                                                           const int* __restrict__ ns_list, int n_bucket, int N,
                                                           const double* __restrict__ Dop) {
  for (int ei = blockIdx.x; ei < n_bucket; ei += gridDim.x) {
    const int ns = ns_list[ei];
    GenDims dn = {{N, N, N}};
    gen_apply(Dop, N, N, 1, 0, u + ns, dn, d0 + ns, false);
    gen_apply(Dop, N, N, 1, 1, u + ns, dn, d1 + ns, false);
    gen_apply(Dop, N, N, 1, 2, u + ns, dn, d2 + ns, false);
  }
}

// out = D_dir in (transpose = 0) or D_dir^T in (1): d4est_operators_apply_dij / _dij_transpose
// (src/dGMath/d4est_operators.c:1385-1410, :2259-2284), batched; a utility entry (the operator path never forms these)
__global__ __launch_bounds__(256) void generic_dij_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                          const int* __restrict__ ns_list, int n_bucket, int N,
                                                          const double* __restrict__ Dop, int dir, int transpose) {
  for (int ei = blockIdx.x; ei < n_bucket; ei += gridDim.x) {
    const int ns = ns_list[ei];
    GenDims dn = {{N, N, N}};
    gen_apply(Dop, N, transpose ? 1 : N, transpose ? N : 1, dir, in + ns, dn, out + ns, false);
  }
}

// ---------------------------------------------------------------------------
// geometry pre-combination: symmetric metric  M_{ab} = w_i w_j w_k J sum_d r_{a,d} r_{b,d}
// from the reference's SoA arrays (setup-time transform).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void metric_precombine_kernel(const double* __restrict__ J, const double* __restrict__ rst,
                                                                size_t local_nodes_quad, const int* __restrict__ qs_list,
                                                                int n_bucket, int NQ,
                                                                const double* __restrict__ wq, double* __restrict__ metric,
                                                                double* __restrict__ affine /* 6 per bucket element */,
                                                                int* __restrict__ nonaffine_flag) {
  const int NQ3 = NQ * NQ * NQ;
  __shared__ double g0[6];
  for (int ei = blockIdx.x; ei < n_bucket; ei += gridDim.x) {
    const int qs = qs_list[ei];
    double* m = metric + (size_t)6 * qs;
    // the unweighted J (dr/dx)(dr/dx)^T of node 0: an element is AFFINE when every node reproduces it (to 4 ulp)
    if (threadIdx.x < 6) {
      int i = 0, j = 0, c = 0;
      for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) {
          if (c == (int)threadIdx.x) { i = a; j = b; }
          ++c;
        }
      double sum = 0.0;
      for (int d = 0; d < 3; ++d)
        sum += rst[(size_t)(3 * i + d) * local_nodes_quad + qs] * rst[(size_t)(3 * j + d) * local_nodes_quad + qs];
      g0[threadIdx.x] = J[qs] * sum;
      affine[(size_t)6 * ei + threadIdx.x] = g0[threadIdx.x];
    }
    __syncthreads();
    double scale = 0.0;
    for (int c = 0; c < 6; ++c) scale = fmax(scale, fabs(g0[c]));
    bool deviates = false;
    for (int q = threadIdx.x; q < NQ3; q += blockDim.x) {
      const int iq = q % NQ, jq = (q / NQ) % NQ, kq = q / (NQ * NQ);
      const double Jq = J[qs + q];
      const double wj = (wq[kq] * (wq[jq] * wq[iq])) * Jq;
      double r[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[i][j] = rst[(size_t)(3 * i + j) * local_nodes_quad + qs + q];
      int c = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
          const double gsum = r[i][0] * r[j][0] + r[i][1] * r[j][1] + r[i][2] * r[j][2];
          m[(size_t)c * NQ3 + q] = wj * gsum;
          if (fabs(Jq * gsum - g0[c]) > 1e-15 * scale) deviates = true;
          ++c;
        }
    }
    if (deviates) atomicOr(nonaffine_flag, 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
#define D4EST_HIP_FAST_PAIRS(X) \
  X(2, 2) X(3, 3) X(4, 4) X(5, 5) X(6, 6) X(7, 7) X(8, 8) X(9, 9) X(10, 10) X(11, 11) X(12, 12) \
  X(13, 13) X(14, 14) X(15, 15) X(16, 16)                                                       \
  X(2, 3) X(3, 4) X(4, 5) X(8, 9) X(3, 6) X(4, 6) X(8, 10)

// p = 16 .. 19 (the reference's tables stop at 20 Lobatto points): only the two-field multi-wave kernels fit the 160 KB LDS
#define D4EST_HIP_BIG_PAIRS(X) X(17, 17) X(18, 18) X(19, 19) X(20, 20)

// (once per kernel instance and size: a driver call per launch would sit in the hot path of every smoother iteration, and inside a
// hipGraph capture region)
template <typename K>
static void set_lds_limit(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> done;
  const void* key = reinterpret_cast<const void*>(kernel);
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(key);
  if (it != done.end() && it->second >= bytes) return;
  HIP_CHECK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done[key] = bytes;
}

static void ensure_scratch(d4est_hip_plan* plan, size_t doubles) {
  if (plan->scratch_doubles >= doubles) return;
  if (plan->d_scratch) HIP_CHECK(hipFree(plan->d_scratch));
  HIP_CHECK(hipMalloc(&plan->d_scratch, doubles * sizeof(double)));
  plan->scratch_doubles = doubles;
}

static void launch_generic(d4est_hip_plan* plan, const Bucket& bk, int mode, const double* in, double* out,
                           const double* coeff = nullptr, const double* op = nullptr, const int* qs_list = nullptr, int NQe = -1,
                           const double* wts = nullptr) {
  if (!wts) wts = bk.d_w;
  if (!op) op = bk.d_B;
  if (!qs_list) qs_list = plan->d_qs_list + bk.elem_offset;
  if (NQe < 0) NQe = bk.NQ;
  const int NM = bk.N > NQe ? bk.N : NQe;
  const size_t per_block = (size_t)8 * NM * NM * NM;
  const int grid = bk.n_elem < 1024 ? bk.n_elem : 1024;
  ensure_scratch(plan, per_block * grid);
  hipLaunchKernelGGL(generic_volume_kernel, dim3(grid), dim3(256), 0, plan->stream, mode, in, out, plan->d_metric, plan->d_J,
                     plan->d_ns_list + bk.elem_offset, qs_list, bk.n_elem, bk.N, NQe, op, bk.d_G, wts, plan->d_scratch,
                     per_block, coeff);
}

template <int N, int NQ>
static void launch_stiffness_wave(d4est_hip_plan* plan, const Bucket& bk, bool use_pf, const double* u, double* Au) {
  if constexpr (NQ * NQ <= 64 && NQ >= N) {
    using W = WaveCfg<N, NQ>;
    const int grid = (bk.n_elem + W::EPB - 1) / W::EPB;
    const int ts = plan->tuning[D4EST_HIP_TUNE_STIFFNESS_STAGGER];
    const int cus_ = plan->n_cus > 0 ? plan->n_cus : 256;
    // stagger is an experiment knob only: delaying half of the resident workgroups changed config-2 time by +-3 %
    // (noise level) in every grouping tried, so auto = off (profiles/r01_d_notes.txt)
    (void)cus_;
    const int stagger = ts < 0 ? 0 : ts;
    const int tw_ = plan->tuning[D4EST_HIP_TUNE_STIFFNESS_WAVE];
    const bool use_affine = bk.affine && plan->tuning[D4EST_HIP_TUNE_AFFINE] != 0 && bk.d_EBf && (tw_ == 11 || tw_ < 0);
    if (use_affine) {
      std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave_eo_kernel<%d,%d,affine>", N, NQ);
      hipLaunchKernelGGL((stiffness_wave_eo_kernel<N, NQ, true>), dim3(grid), dim3(64), W::LDS_BYTES, plan->stream, u, Au, plan->d_metric,
                           plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_EBf, bk.d_EGf,
                           bk.d_EBb, bk.d_EGb, bk.ns0, bk.ns_stride, bk.qs0, bk.qs_stride,
                           plan->d_metric_affine + (size_t)6 * bk.elem_offset, bk.d_w);
    } else if ((tw_ == 11 || tw_ < 0) && bk.d_EBf) {
      std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave_eo_kernel<%d,%d>", N, NQ);
      hipLaunchKernelGGL((stiffness_wave_eo_kernel<N, NQ>), dim3(grid), dim3(64), W::LDS_BYTES, plan->stream, u, Au, plan->d_metric,
                           plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_EBf, bk.d_EGf,
                           bk.d_EBb, bk.d_EGb, bk.ns0, bk.ns_stride, bk.qs0, bk.qs_stride);
    } else if (tw_ == 3 || tw_ < 0) {
      std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave2_kernel<%d,%d>", N, NQ);
      hipLaunchKernelGGL((stiffness_wave2_kernel<N, NQ>), dim3(grid), dim3(64), W::LDS_BYTES, plan->stream, u, Au, plan->d_metric,
                         plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_B, bk.d_G, bk.d_BT,
                         bk.d_GT, stagger);
    } else if (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_WAVE] == 2 && N % 2 == 0 && NQ % 2 == 0) {
      std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_pair_kernel<%d,%d>", N, NQ);
      if constexpr (N % 2 == 0 && NQ % 2 == 0)
        hipLaunchKernelGGL((stiffness_pair_kernel<N, NQ>), dim3(grid), dim3(128), (W::LDS_BYTES / 2) * 3, plan->stream, u, Au,
                           plan->d_metric, plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem,
                           bk.d_B, bk.d_G, bk.d_BT, bk.d_GT);
    } else if (use_pf) {
      std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave_kernel<%d,%d,true>", N, NQ);
      hipLaunchKernelGGL((stiffness_wave_kernel<N, NQ, true>), dim3(grid), dim3(64), W::LDS_BYTES, plan->stream, u, Au,
                         plan->d_metric, plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem,
                         bk.d_B, bk.d_G, bk.d_BT, bk.d_GT, stagger);
    } else {
      std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave_kernel<%d,%d,false>", N, NQ);
      hipLaunchKernelGGL((stiffness_wave_kernel<N, NQ, false>), dim3(grid), dim3(64), W::LDS_BYTES, plan->stream, u, Au,
                         plan->d_metric, plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem,
                         bk.d_B, bk.d_G, bk.d_BT, bk.d_GT, stagger);
    }
  }
}

static void launch_stiffness_mfma16(d4est_hip_plan* plan, const Bucket& bk, const double* u, double* Au) {
  const bool aff = bk.affine && plan->tuning[D4EST_HIP_TUNE_AFFINE] != 0 && plan->d_metric_affine && bk.N == 16;
  std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_mfma16_kernel<%d%s> (512 threads, v_mfma_f64_16x16x4)", bk.N,
                aff ? ",affine" : "");
  const int cus = plan->n_cus > 0 ? plan->n_cus : 256;
  const int grid = bk.n_elem < cus ? bk.n_elem : cus;   // one 102 KB workgroup per CU, persistent over the bucket
  auto go = [&](auto kern) {
    set_lds_limit(kern, Mfma16Cfg::LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Mfma16Cfg::LDS_BYTES, plan->stream, u, Au, plan->d_metric,
                       plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_B, bk.d_G,
                       aff ? plan->d_metric_affine + (size_t)6 * bk.elem_offset : (const double*)nullptr, bk.d_w, plan->stream_mode);
  };
  if (aff) go(stiffness_mfma16_kernel<16, true>);
  else if (bk.N == 16) go(stiffness_mfma16_kernel<16>);
  else if (bk.N == 15) go(stiffness_mfma16_kernel<15>);
  else if (bk.N == 14) go(stiffness_mfma16_kernel<14>);
  else go(stiffness_mfma16_kernel<13>);
}

// Mixed-degree plans: the buckets the single-wavefront even-odd kernel would take (deg_quad = deg <= 7, automatic kernel choice) go
// into ONE launch per metric form (streamed / affine) when there are at least two of them; returns the buckets it covered.
static unsigned launch_stiffness_multi(d4est_hip_plan* plan, const double* u, double* Au) {
  const int tw = plan->tuning[D4EST_HIP_TUNE_STIFFNESS_WAVE];
  if (!(tw < 0 || tw == 11) || plan->tuning[D4EST_HIP_TUNE_STIFFNESS_PREFETCH] > 0) return 0u;
  unsigned covered = 0u;
  for (int aff = 0; aff < 2; ++aff) {
    WaveEoMulti A;
    size_t lds = 0;
    unsigned mine = 0u;
    int wgs = 0;
    for (size_t i = 0; i < plan->buckets.size() && i < 32; ++i) {
      const Bucket& bk = plan->buckets[i];
      if (bk.n_elem == 0 || bk.N != bk.NQ || bk.N < 2 || bk.N > 8 || !bk.d_EBf || A.n == WaveEoMulti::MAXB) continue;
      const bool use_affine = bk.affine && plan->tuning[D4EST_HIP_TUNE_AFFINE] != 0 && plan->d_metric_affine;
      if ((int)use_affine != aff || !(bk.n_elem <= 8192 || use_affine)) continue;   // (larger buckets: the prefetching kernel, see below)
      size_t l = 0;
      int epb = 1;
#define X(N_) if (bk.N == N_) { l = WaveCfg<N_, N_>::LDS_BYTES; epb = WaveCfg<N_, N_>::EPB; }
      X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
      lds = std::max(lds, l);
      wgs += (bk.n_elem + epb - 1) / epb;
      const int j = A.n++;
      A.wg_end[j] = wgs; A.N[j] = bk.N; A.n_elem[j] = bk.n_elem; A.elem_offset[j] = bk.elem_offset;
      A.EBf[j] = bk.d_EBf; A.EGf[j] = bk.d_EGf; A.EBb[j] = bk.d_EBb; A.EGb[j] = bk.d_EGb; A.wq[j] = bk.d_w;
      mine |= 1u << i;
    }
    if (A.n < 2) continue;
    std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave_eo_multi_kernel<%s> (%d buckets)", aff ? "affine" : "general", A.n);
    if (aff)
      hipLaunchKernelGGL((stiffness_wave_eo_multi_kernel<true>), dim3(wgs), dim3(64), lds, plan->stream, u, Au, plan->d_metric, plan->d_ns_list,
                         plan->d_qs_list, plan->d_metric_affine, A);
    else
      hipLaunchKernelGGL((stiffness_wave_eo_multi_kernel<false>), dim3(wgs), dim3(64), lds, plan->stream, u, Au, plan->d_metric, plan->d_ns_list,
                         plan->d_qs_list, (const double*)nullptr, A);
    covered |= mine;
  }
  return covered;
}

// ... and the p = 8 ... 12 buckets of such a plan by workgroup size (stiffness_mw_multi_kernel): general path, automatic kernel choice
static unsigned launch_stiffness_multi_mw(d4est_hip_plan* plan, const double* u, double* Au) {
  if (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] >= 0 && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] != 1) return 0u;
  if (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] == 0) return 0u;
  unsigned covered = 0u;
  for (int threads = 128; threads <= 192; threads += 64) {
    WaveEoMulti A;
    size_t lds = 0;
    unsigned mine = 0u;
    int wgs = 0;
    for (size_t i = 0; i < plan->buckets.size() && i < 32; ++i) {
      const Bucket& bk = plan->buckets[i];
      if (bk.n_elem == 0 || bk.N != bk.NQ || bk.N < 9 || bk.N > 13 || !bk.d_EDq || A.n == WaveEoMulti::MAXB) continue;
      if (bk.affine && plan->tuning[D4EST_HIP_TUNE_AFFINE] != 0 && plan->d_metric_affine) continue;   // (affine buckets keep their kernel)
      const int th = ((bk.N * bk.N + 63) / 64) * 64;
      if (th != threads) continue;
      lds = std::max(lds, (size_t)2 * bk.N * bk.N * (bk.N | 1) * sizeof(double));
      wgs += bk.n_elem;
      const int j = A.n++;
      A.wg_end[j] = wgs; A.N[j] = bk.N; A.n_elem[j] = bk.n_elem; A.elem_offset[j] = bk.elem_offset;
      A.EBf[j] = bk.d_EBf; A.EGf[j] = bk.d_EDqT; A.EBb[j] = bk.d_EBb; A.EGb[j] = bk.d_EDq; A.wq[j] = bk.d_w;
      mine |= 1u << i;
    }
    if (A.n < 2) continue;
    std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_mw_multi_kernel<%d> (%d buckets)", threads, A.n);
    auto go = [&](auto kern) {
      set_lds_limit(kern, lds);
      hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, plan->stream, u, Au, plan->d_metric, plan->d_ns_list, plan->d_qs_list, A);
    };
    if (threads == 128) { if (plan->stream_mode) go(stiffness_mw_multi_kernel<128, true>); else go(stiffness_mw_multi_kernel<128, false>); }
    else { if (plan->stream_mode) go(stiffness_mw_multi_kernel<192, true>); else go(stiffness_mw_multi_kernel<192, false>); }
    covered |= mine;
  }
  return covered;
}


// ... and both kinds in one launch where a plan has p <= 7 buckets AND 128-thread multi-wave buckets (stiffness_all_multi_kernel)
static unsigned launch_stiffness_all_multi(d4est_hip_plan* plan, const double* u, double* Au) {
  const int tw = plan->tuning[D4EST_HIP_TUNE_STIFFNESS_WAVE];
  if (!(tw < 0 || tw == 11) || plan->tuning[D4EST_HIP_TUNE_STIFFNESS_PREFETCH] > 0) return 0u;
  if (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] >= 0 && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] != 1) return 0u;
  if (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] == 0) return 0u;
  static const bool split = std::getenv("D4EST_HIP_STIFFNESS_SPLIT_LAUNCH") != nullptr;
  if (split) return 0u;
  WaveEoMulti A;
  size_t lds = 0;
  unsigned mine = 0u;
  int wgs = 0, n_eo = 0, n_mw = 0;
  for (size_t i = 0; i < plan->buckets.size() && i < 32; ++i) {
    const Bucket& bk = plan->buckets[i];
    if (bk.n_elem == 0 || bk.N != bk.NQ || A.n == WaveEoMulti::MAXB) continue;
    if (bk.affine && plan->tuning[D4EST_HIP_TUNE_AFFINE] != 0 && plan->d_metric_affine) continue;   // (affine buckets keep their kernels)
    const int j = A.n;
    if (bk.N >= 2 && bk.N <= 8 && bk.d_EBf && bk.n_elem <= 8192) {
      size_t l = 0;
      int epb = 1;
#define X(N_) if (bk.N == N_) { l = WaveCfg<N_, N_>::LDS_BYTES; epb = WaveCfg<N_, N_>::EPB; }
      X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
      lds = std::max(lds, 2 * l);
      const int units = (bk.n_elem + epb - 1) / epb;
      wgs += (units + 1) / 2;
      A.EBf[j] = bk.d_EBf; A.EGf[j] = bk.d_EGf; A.EBb[j] = bk.d_EBb; A.EGb[j] = bk.d_EGb;
      ++n_eo;
    } else if (bk.N >= 9 && bk.N <= 11 && bk.d_EDq) {
      lds = std::max(lds, (size_t)2 * bk.N * bk.N * (bk.N | 1) * sizeof(double));
      wgs += bk.n_elem;
      A.EBf[j] = bk.d_EBf; A.EGf[j] = bk.d_EDqT; A.EBb[j] = bk.d_EBb; A.EGb[j] = bk.d_EDq;
      ++n_mw;
    } else {
      continue;
    }
    A.n = j + 1;
    A.wg_end[j] = wgs; A.N[j] = bk.N; A.n_elem[j] = bk.n_elem; A.elem_offset[j] = bk.elem_offset; A.wq[j] = bk.d_w;
    mine |= 1u << i;
  }
  if (n_eo == 0 || n_mw == 0) return 0u;   // one kind only: the launches above
  std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_all_multi_kernel (%d + %d buckets)", n_eo, n_mw);
  auto go = [&](auto kern) {
    set_lds_limit(kern, lds);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(128), lds, plan->stream, u, Au, plan->d_metric, plan->d_ns_list, plan->d_qs_list, A);
  };
  if (plan->stream_mode) go(stiffness_all_multi_kernel<true>); else go(stiffness_all_multi_kernel<false>);
  return mine;
}

void launch_stiffness(d4est_hip_plan* plan, const double* u, double* Au) {
  if (!plan->has_geometry) D4EST_HIP_ABORT("apply_stiffness_matrix: d4est_hip_plan_set_geometry was not called");
  unsigned covered = launch_stiffness_all_multi(plan, u, Au);
  if (covered == 0u) covered = launch_stiffness_multi(plan, u, Au) | launch_stiffness_multi_mw(plan, u, Au);
  size_t bucket_index = 0;
  for (const Bucket& bk : plan->buckets) {
    const size_t this_bucket = bucket_index++;
    if (bk.n_elem == 0) continue;
    if (this_bucket < 32 && ((covered >> this_bucket) & 1u)) continue;
    bool done = false;
    // auto-tuning (measured on MI355X, p = 7): up to ~2 resident rounds (16 one-wave workgroups per CU)
    // the two-buffer wave kernel wins; for larger buckets the 3-buffer kernel with the metric requested
    // at entry streams HBM best.  profiles/r01_*_ab.txt
    const int tw = plan->tuning[D4EST_HIP_TUNE_STIFFNESS_WAVE], tp = plan->tuning[D4EST_HIP_TUNE_STIFFNESS_PREFETCH];
    const bool affine_ok = bk.affine && plan->tuning[D4EST_HIP_TUNE_AFFINE] != 0;
    // multi-wave kernels (p >= 8): the affine bucket's six constants per element, or nullptr = stream the metric
#define D4EST_AFF_ARGS (affine_ok && plan->d_metric_affine ? plan->d_metric_affine + (size_t)6 * bk.elem_offset : (const double*)nullptr), bk.d_w
    const bool use_wave = (tw < 0) ? (bk.n_elem <= 8192 || affine_ok) : (tw != 0);
    const bool use_pf = (tp < 0) ? !use_wave : (tp != 0);
#define X(N_, NQ_)                                                                                              \
  if (!done && bk.N == N_ && bk.NQ == NQ_) {                                                                    \
    using C = VolCfg<N_, NQ_>;                                                                                  \
    if (C::LDS_BYTES <= 160 * 1024) {                                                                           \
      const int grid = (bk.n_elem + C::EPB - 1) / C::EPB;                                                       \
      constexpr bool kWave = (NQ_ * NQ_ <= 64);                                                                 \
      constexpr bool kEven = true;   /* the even-odd contractions take sizes of either parity */                                                   \
      const bool use_eo = kEven && bk.d_EBf && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] != 0;                  \
      if (kWave && use_wave) {                                               \
        launch_stiffness_wave<N_, (kWave ? NQ_ : N_)>(plan, bk, use_pf, u, Au);                                 \
      } else if (N_ >= 13 && N_ <= 16 && NQ_ == N_ && (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] == 2 || (N_ == 16 && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] < 0))) { \
        launch_stiffness_mfma16(plan, bk, u, Au);                                                               \
      } else if (!kWave && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] != 0) {                                  \
        /* p >= 8: multi-wave workgroup, two LDS fields (sequential field hand-off) -> 1.5x the residency */  \
        using W = WaveCfg<N_, NQ_>;                                                                             \
        std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave_kernel<%d,%d,false,%s%s> (%d threads)", N_, NQ_, use_eo ? "eo" : "plain", (use_eo && affine_ok && plan->d_metric_affine) ? ",affine" : "", W::THREADS); \
        if (use_eo) {                                                                                           \
          if (affine_ok && plan->d_metric_affine) {                                                             \
            set_lds_limit(stiffness_wave_kernel<N_, NQ_, false, kEven, true>, W::LDS_BYTES);                    \
            hipLaunchKernelGGL((stiffness_wave_kernel<N_, NQ_, false, kEven, true>), dim3(bk.n_elem), dim3(W::THREADS), W::LDS_BYTES, \
                               plan->stream, u, Au, plan->d_metric, plan->d_ns_list + bk.elem_offset,           \
                               plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_EBb, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDq : bk.d_EGb), bk.d_EBf, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDqT : bk.d_EGf), 0, D4EST_AFF_ARGS); \
          } else {                                                                                              \
          auto go_mw = [&](auto kern) {   /* NT twin: stream mode (plan->stream_mode, d4est_hip_wave.h) */ \
            set_lds_limit(kern, W::LDS_BYTES); \
            hipLaunchKernelGGL(kern, dim3(bk.n_elem), dim3(W::THREADS), W::LDS_BYTES, \
                             plan->stream, u, Au, plan->d_metric, plan->d_ns_list + bk.elem_offset,             \
                             plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_EBb, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDq : bk.d_EGb), bk.d_EBf, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDqT : bk.d_EGf), (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_STAGGER] < 0 ? 0 : plan->tuning[D4EST_HIP_TUNE_STIFFNESS_STAGGER]), (const double*)nullptr, (const double*)nullptr); \
          }; \
          if (plan->stream_mode) go_mw(stiffness_wave_kernel<N_, NQ_, false, kEven, false, true>); \
          else go_mw(stiffness_wave_kernel<N_, NQ_, false, kEven>); \
          }                                                                                                     \
        } else {                                                                                                \
          set_lds_limit(stiffness_wave_kernel<N_, NQ_, false>, W::LDS_BYTES);                                   \
          hipLaunchKernelGGL((stiffness_wave_kernel<N_, NQ_, false>), dim3(bk.n_elem), dim3(W::THREADS), W::LDS_BYTES, \
                             plan->stream, u, Au, plan->d_metric, plan->d_ns_list + bk.elem_offset,             \
                             plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_B, bk.d_G, bk.d_BT, bk.d_GT, 0); \
        }                                                                                                       \
      } else {                                                                                                  \
        constexpr bool kCanPF = (NQ_ <= 8);                                                                     \
        const bool pf = kCanPF && use_pf;                                                                       \
        std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_kernel<%d,%d,%s,%s>", N_, NQ_, pf ? "true" : "false", use_eo ? "eo" : "plain"); \
        const double* o0 = use_eo ? bk.d_EBb : bk.d_B;                                                          \
        const double* o1 = use_eo ? bk.d_EGb : bk.d_G;                                                          \
        const double* o2 = use_eo ? bk.d_EBf : bk.d_BT;                                                         \
        const double* o3 = use_eo ? bk.d_EGf : bk.d_GT;                                                         \
        auto go = [&](auto kern) {                                                                              \
          set_lds_limit(kern, C::LDS_BYTES);                                                                    \
          hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, plan->stream, u, Au, plan->d_metric, \
                             plan->d_ns_list + bk.elem_offset, plan->d_qs_list + bk.elem_offset, bk.n_elem, o0, o1, o2, o3); \
        };                                                                                                      \
        if (pf && use_eo && plan->stream_mode) go(stiffness_kernel<N_, NQ_, kCanPF, kEven, true>);              \
        else if (pf && use_eo) go(stiffness_kernel<N_, NQ_, kCanPF, kEven>);                                    \
        else if (pf) go(stiffness_kernel<N_, NQ_, kCanPF, false>);                                              \
        else if (use_eo) go(stiffness_kernel<N_, NQ_, false, kEven>);                                           \
        else go(stiffness_kernel<N_, NQ_, false, false>);                                                       \
      }                                                                                                         \
      done = true;                                                                                              \
    }                                                                                                           \
  }
    D4EST_HIP_FAST_PAIRS(X)
#undef X
#define X(N_, NQ_)                                                                                              \
  if (!done && bk.N == N_ && bk.NQ == NQ_ && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_BIGP] != 0) {                \
    using W = WaveCfg<N_, NQ_>;                                                                                 \
    static_assert(W::LDS_BYTES <= 160 * 1024, "two-field kernel does not fit the LDS");                         \
    constexpr bool kEven = true;   /* the even-odd contractions take sizes of either parity */                                                     \
    const bool use_eo = kEven && bk.d_EBf && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] != 0;                    \
    std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::stiffness_wave_kernel<%d,%d,false,%s%s> (%d threads)", N_, NQ_, use_eo ? "eo" : "plain", (use_eo && affine_ok && plan->d_metric_affine) ? ",affine" : "", W::THREADS); \
    if (use_eo) {                                                                                               \
      if (affine_ok && plan->d_metric_affine) {                                                                 \
        set_lds_limit(stiffness_wave_kernel<N_, NQ_, false, kEven, true>, W::LDS_BYTES);                        \
        hipLaunchKernelGGL((stiffness_wave_kernel<N_, NQ_, false, kEven, true>), dim3(bk.n_elem), dim3(W::THREADS), W::LDS_BYTES, \
                           plan->stream, u, Au, plan->d_metric, plan->d_ns_list + bk.elem_offset,               \
                           plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_EBb, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDq : bk.d_EGb), bk.d_EBf, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDqT : bk.d_EGf), 0, D4EST_AFF_ARGS); \
      } else {                                                                                                  \
      auto go_mw = [&](auto kern) {   /* NT twin: stream mode (plan->stream_mode, d4est_hip_wave.h) */ \
        set_lds_limit(kern, W::LDS_BYTES); \
        hipLaunchKernelGGL(kern, dim3(bk.n_elem), dim3(W::THREADS), W::LDS_BYTES, \
                         plan->stream, u, Au, plan->d_metric, plan->d_ns_list + bk.elem_offset,                 \
                         plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_EBb, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDq : bk.d_EGb), bk.d_EBf, (kMwCollocated<N_, NQ_, false, true> ? bk.d_EDqT : bk.d_EGf), (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_STAGGER] < 0 ? 0 : plan->tuning[D4EST_HIP_TUNE_STIFFNESS_STAGGER]), (const double*)nullptr, (const double*)nullptr); \
      }; \
      if (plan->stream_mode) go_mw(stiffness_wave_kernel<N_, NQ_, false, kEven, false, true>); \
      else go_mw(stiffness_wave_kernel<N_, NQ_, false, kEven>); \
      }                                                                                                         \
    } else {                                                                                                    \
      set_lds_limit(stiffness_wave_kernel<N_, NQ_, false, false>, W::LDS_BYTES);                                \
      hipLaunchKernelGGL((stiffness_wave_kernel<N_, NQ_, false, false>), dim3(bk.n_elem), dim3(W::THREADS), W::LDS_BYTES, \
                         plan->stream, u, Au, plan->d_metric, plan->d_ns_list + bk.elem_offset,                 \
                         plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.d_B, bk.d_G, bk.d_BT, bk.d_GT, 0);     \
    }                                                                                                           \
    done = true;                                                                                                \
  }
    D4EST_HIP_BIG_PAIRS(X)
#undef X
#undef D4EST_AFF_ARGS
    if (!done) {
      std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::generic_volume_kernel (N=%d,NQ=%d)", bk.N, bk.NQ);
      launch_generic(plan, bk, 3, u, Au);
    }
  }
  HIP_CHECK(hipGetLastError());
}

// which = 0: the quadrature interpolation B; 1: B^-1 (inverse mass); 2: 1-D mass M; 3: M^-1 (square nodal applies)
template <int MODE>
static void launch_mass_like_mode(d4est_hip_plan* plan, const double* in, double* out, const double* coeff, int which) {
  unsigned covered = 0u;
  if constexpr (MODE == 0 || MODE == 3) {
    if (which == 0 && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] != 0) {   // mixed-degree plans: one launch for the deg_quad = deg <= 7 buckets
      WaveEoMulti A;
      size_t lds = 0;
      int wgs = 0;
      for (size_t i = 0; i < plan->buckets.size() && i < 32; ++i) {
        const Bucket& bk = plan->buckets[i];
        if (bk.n_elem == 0 || bk.N != bk.NQ || bk.N < 2 || bk.N > 8 || !bk.d_EBf || A.n == WaveEoMulti::MAXB) continue;
        size_t l = 0;
        int epb = 1;
#define X(N_) if (bk.N == N_) { l = WaveCfg<N_, N_>::LDS_BYTES; epb = WaveCfg<N_, N_>::EPB; }
        X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
        lds = std::max(lds, l);
        wgs += (bk.n_elem + epb - 1) / epb;
        const int j = A.n++;
        A.wg_end[j] = wgs; A.N[j] = bk.N; A.n_elem[j] = bk.n_elem; A.elem_offset[j] = bk.elem_offset;
        A.EBf[j] = bk.d_EBf; A.EBb[j] = bk.d_EBb; A.wq[j] = bk.d_w;
        covered |= 1u << i;
      }
      if (A.n >= 2)
        hipLaunchKernelGGL((mass_like_multi_kernel<MODE>), dim3(wgs), dim3(64), lds, plan->stream, in, out, plan->d_J, plan->d_ns_list,
                           plan->d_qs_list, coeff, A);
      else
        covered = 0u;
    }
  }
  size_t bucket_index = 0;
  for (const Bucket& bk : plan->buckets) {
    const size_t this_bucket = bucket_index++;
    if (bk.n_elem == 0) continue;
    if (this_bucket < 32 && ((covered >> this_bucket) & 1u)) continue;
    if (which == 1 && bk.N != bk.NQ) D4EST_HIP_ABORT("apply_inverse_mass_matrix needs deg_quad == deg (reference asserts the same, d4est_quadrature.c:1233)");
    const double* op = which == 0 ? bk.d_B : (which == 1 ? bk.d_BinvT : (which == 2 ? bk.d_M : bk.d_Minv));
    const double* opT = which == 0 ? bk.d_BT : (which == 1 ? bk.d_Binv : (which == 2 ? bk.d_MT : bk.d_MinvT));
    // square nodal applies write to the nodal layout
    const int* qs_list = (which >= 2) ? plan->d_ns_list + bk.elem_offset : plan->d_qs_list + bk.elem_offset;
    const int NQe = (which >= 2) ? bk.N : bk.NQ;
    const double* wts = (which == 1) ? bk.d_wGL : bk.d_w;  // the inverse mass is always Gauss-Legendre (d4est_quadrature.c:1239-1241)
    bool done = false;
#define X(N_, NQ_)                                                                                                    \
  if (!done && bk.N == N_ && NQe == NQ_) {                                                                            \
    using C = WaveCfg<N_, NQ_>;                                                                                       \
    if (C::LDS_BYTES <= 160 * 1024) {                                                                                 \
      const int grid = (bk.n_elem + C::EPB - 1) / C::EPB;                                                             \
      constexpr bool kEven = true;   /* the even-odd contractions take sizes of either parity */                                                         \
      if (kEven && which == 0 && bk.d_EBf && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] != 0) {                        \
        set_lds_limit(mass_like_kernel<N_, NQ_, MODE, kEven>, C::LDS_BYTES);                                          \
        hipLaunchKernelGGL((mass_like_kernel<N_, NQ_, MODE, kEven>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, plan->stream, \
                           in, out, plan->d_J, plan->d_ns_list + bk.elem_offset, qs_list, bk.n_elem, bk.d_EBb, bk.d_EBf, wts, \
                           coeff);                                                                                    \
      } else {                                                                                                        \
        set_lds_limit(mass_like_kernel<N_, NQ_, MODE, false>, C::LDS_BYTES);                                          \
        hipLaunchKernelGGL((mass_like_kernel<N_, NQ_, MODE, false>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, plan->stream, \
                           in, out, plan->d_J, plan->d_ns_list + bk.elem_offset, qs_list, bk.n_elem, op, opT, wts,     \
                           coeff);                                                                                    \
      }                                                                                                               \
      done = true;                                                                                                    \
    }                                                                                                                 \
  }
    D4EST_HIP_FAST_PAIRS(X)
    D4EST_HIP_BIG_PAIRS(X)
#undef X
    if (!done) launch_generic(plan, bk, MODE == 3 ? 4 : (MODE == 4 ? 5 : MODE), in, out, coeff, op, qs_list, NQe, wts);
  }
  HIP_CHECK(hipGetLastError());
}

void launch_mass_like(d4est_hip_plan* plan, int mode, const double* in, double* out, const double* coeff, int which) {
  if (mode != 2 && !plan->has_geometry) D4EST_HIP_ABORT("mass/galerkin apply: d4est_hip_plan_set_geometry was not called");
  if (mode == 0) launch_mass_like_mode<0>(plan, in, out, coeff, which);
  else if (mode == 1) launch_mass_like_mode<1>(plan, in, out, coeff, which);
  else if (mode == 2) launch_mass_like_mode<2>(plan, in, out, coeff, which);
  else if (mode == 3) launch_mass_like_mode<3>(plan, in, out, coeff, which);
  else if (mode == 4) launch_mass_like_mode<4>(plan, in, out, coeff, which);
  else D4EST_HIP_ABORT("launch_mass_like: bad mode %d", mode);
}

// The volume term on a sub-list of every bucket (the hybrid operator's dirty elements, d4est_hip_direct.hip): the bucket-ordered lists and
// counts are swapped for the view's while launch_stiffness runs.  Offsets come from the lists (no affine strides), the metric is
// streamed (the per-bucket affine constants are indexed by bucket position).
void launch_stiffness_view(d4est_hip_plan* plan, const double* u, double* Au, int* ns_list_view, int* qs_list_view, const int* view_offset,
                           const int* view_count) {
  struct Saved { int elem_offset, n_elem, ns_stride, qs_stride; bool affine; };
  std::vector<Saved> saved(plan->buckets.size());
  int* ns_saved = plan->d_ns_list;
  int* qs_saved = plan->d_qs_list;
  for (size_t i = 0; i < plan->buckets.size(); ++i) {
    Bucket& bk = plan->buckets[i];
    saved[i] = Saved{bk.elem_offset, bk.n_elem, bk.ns_stride, bk.qs_stride, bk.affine};
    bk.elem_offset = view_offset[i]; bk.n_elem = view_count[i]; bk.ns_stride = -1; bk.qs_stride = -1; bk.affine = false;
  }
  plan->d_ns_list = ns_list_view;
  plan->d_qs_list = qs_list_view;
  char kept[sizeof(plan->last_kernel)];
  std::memcpy(kept, plan->last_kernel, sizeof(kept));
  launch_stiffness(plan, u, Au);
  std::memcpy(plan->last_kernel, kept, sizeof(kept));
  plan->d_ns_list = ns_saved;
  plan->d_qs_list = qs_saved;
  for (size_t i = 0; i < plan->buckets.size(); ++i) {
    Bucket& bk = plan->buckets[i];
    bk.elem_offset = saved[i].elem_offset; bk.n_elem = saved[i].n_elem; bk.ns_stride = saved[i].ns_stride; bk.qs_stride = saved[i].qs_stride;
    bk.affine = saved[i].affine;
  }
}

void launch_dudr(d4est_hip_plan* plan, const double* u, double* d0, double* d1, double* d2) {
  for (const Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    bool done = false;
#define X(N_, NQ_)                                                                                         \
  if (!done && N_ == NQ_ && bk.N == N_) {                                                                  \
    using C = WaveCfg<N_, N_>;                                                                               \
    if (C::LDS_BYTES <= 160 * 1024) {                                                                      \
      set_lds_limit(dudr_kernel<N_>, C::LDS_BYTES);                                                        \
      const int grid = (bk.n_elem + C::EPB - 1) / C::EPB;                                                  \
      hipLaunchKernelGGL((dudr_kernel<N_>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, plan->stream, u,   \
                         d0, d1, d2, plan->d_ns_list + bk.elem_offset, bk.n_elem, bk.d_DT);                                                                          \
      done = true;                                                                                         \
    }                                                                                                      \
  }
    D4EST_HIP_FAST_PAIRS(X)
    D4EST_HIP_BIG_PAIRS(X)
#undef X
    if (!done) {
      const int grid = bk.n_elem < 2048 ? bk.n_elem : 2048;
      hipLaunchKernelGGL(generic_dudr_kernel, dim3(grid), dim3(256), 0, plan->stream, u, d0, d1, d2,
                         plan->d_ns_list + bk.elem_offset, bk.n_elem, bk.N, bk.d_D);
    }
  }
  HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Geometric factors of the reference's `brick` geometry generated on the device (brick only):
// d4est_geometry_brick_DX (src/Geometry/d4est_geometry_brick.c:140-206) is diagonal and constant per element,
//   dx_d/dr_d = (X1_d - X0_d) (dq / P4EST_ROOT_LEN) / 2,   J = prod_d dx_d/dr_d,   dr_d/dx_d = 1 / (dx_d/dr_d),
// so J_quad, the symmetric metric and the six affine constants are written directly: no 96 B/node host arrays, no upload.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void brick_metric_kernel(const int* __restrict__ elem_ids, const int* __restrict__ qs_list,
                                                           int n_bucket, int NQ, const double* __restrict__ wq,
                                                           const int* __restrict__ elem_dq, double root_len, double ex, double ey, double ez,
                                                           double* __restrict__ Jq, double* __restrict__ metric,
                                                           double* __restrict__ affine) {
  const int NQ3 = NQ * NQ * NQ;
  for (int ei = blockIdx.x; ei < n_bucket; ei += gridDim.x) {
    const int qs = qs_list[ei];
    const double half = (double)elem_dq[elem_ids[ei]] / root_len / 2.;
    const double hx = ex * half, hy = ey * half, hz = ez * half;
    const double J = hx * hy * hz;
    const double g[6] = {J * ((1. / hx) * (1. / hx)), 0., 0., J * ((1. / hy) * (1. / hy)), 0., J * ((1. / hz) * (1. / hz))};
    if (threadIdx.x < 6) affine[(size_t)6 * ei + threadIdx.x] = g[threadIdx.x];
    double* m = metric + (size_t)6 * qs;
    for (int q = threadIdx.x; q < NQ3; q += blockDim.x) {
      const int iq = q % NQ, jq = (q / NQ) % NQ, kq = q / (NQ * NQ);
      const double w3 = wq[kq] * (wq[jq] * wq[iq]);
      Jq[qs + q] = J;
#pragma unroll
      for (int c = 0; c < 6; ++c) m[(size_t)c * NQ3 + q] = w3 * g[c];
    }
  }
}

void launch_brick_geometry(d4est_hip_plan* plan, const int* d_elem_dq, double root_len, const double* extents) {
  if (!plan->d_metric_affine) {
    HIP_CHECK(hipMalloc(&plan->d_metric_affine, std::max<size_t>((size_t)6 * plan->n_elements, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc(&plan->d_nonaffine, std::max<size_t>(plan->buckets.size(), 1) * sizeof(int)));
  }
  for (size_t bi = 0; bi < plan->buckets.size(); ++bi) {
    Bucket& bk = plan->buckets[bi];
    if (bk.n_elem == 0) continue;
    const int grid = bk.n_elem < 4096 ? bk.n_elem : 4096;
    hipLaunchKernelGGL(brick_metric_kernel, dim3(grid), dim3(256), 0, plan->stream, plan->d_elem_ids + bk.elem_offset,
                       plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.NQ, bk.d_w, d_elem_dq, root_len, extents[1] - extents[0],
                       extents[3] - extents[2], extents[5] - extents[4], plan->d_J, plan->d_metric,
                       plan->d_metric_affine + (size_t)6 * bk.elem_offset);
    bk.affine = true;
  }
  HIP_CHECK(hipGetLastError());
}

// Volume factors of an analytic tree map (d4est_hip_maps.h), DX_compute_method = GEOM_COMPUTE_ANALYTIC: per quadrature node
// dx/dr from the map's Jacobian, J = det, dr/dx = inverse (src/Mesh/d4est_mesh.c:2637-2680, src/Geometry/d4est_geometry.c:877-976),
// then J and the pre-combined symmetric metric are written directly.
__global__ __launch_bounds__(256) void analytic_metric_kernel(const int* __restrict__ elem_ids, const int* __restrict__ qs_list, int n_bucket,
                                                              int NQ, const double* __restrict__ xq, const double* __restrict__ wq,
                                                              const CellDesc* __restrict__ cells, TreeMapParams P, double root_len,
                                                              double* __restrict__ Jq, double* __restrict__ metric) {
  const int NQ3 = NQ * NQ * NQ;
  for (int ei = blockIdx.x; ei < n_bucket; ei += gridDim.x) {
    const int qs = qs_list[ei];
    const CellDesc cell = cells[elem_ids[ei]];
    double* m = metric + (size_t)6 * qs;
    for (int n = threadIdx.x; n < NQ3; n += blockDim.x) {
      const int iq = n % NQ, jq = (n / NQ) % NQ, kq = n / (NQ * NQ);
      const double r[3] = {xq[iq], xq[jq], xq[kq]};
      double dxdr[3][3], inv[3][3];
      cell_dxdr(P, cell, root_len, r, dxdr);
      const double J = invert3(dxdr, inv);        // inv[i][j] = d r_i / d x_j
      const double w3 = wq[kq] * (wq[jq] * wq[iq]) * J;
      Jq[qs + n] = J;
      int c = 0;
      for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) {
          m[(size_t)c * NQ3 + n] = w3 * (inv[a][0] * inv[b][0] + inv[a][1] * inv[b][1] + inv[a][2] * inv[b][2]);
          ++c;
        }
    }
  }
}

void launch_analytic_geometry(d4est_hip_plan* plan, const TreeMapParams& P, const CellDesc* d_cells, double root_len) {
  if (!plan->d_metric_affine) {
    HIP_CHECK(hipMalloc(&plan->d_metric_affine, std::max<size_t>((size_t)6 * plan->n_elements, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc(&plan->d_nonaffine, std::max<size_t>(plan->buckets.size(), 1) * sizeof(int)));
  }
  for (Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    std::vector<double> x, w;
    if (plan->quad_type == QUAD_LEGENDRE) Tables1D::gauss(bk.deg_quad, x, w);
    else Tables1D::lobatto(bk.deg_quad, x, w);
    double* d_x = nullptr;
    HIP_CHECK(hipMalloc(&d_x, x.size() * sizeof(double)));
    HIP_CHECK(hipMemcpyAsync(d_x, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, plan->stream));
    const int grid = bk.n_elem < 4096 ? bk.n_elem : 4096;
    hipLaunchKernelGGL(analytic_metric_kernel, dim3(grid), dim3(256), 0, plan->stream, plan->d_elem_ids + bk.elem_offset,
                       plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.NQ, d_x, bk.d_w, d_cells, P, root_len, plan->d_J, plan->d_metric);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(plan->stream));
    HIP_CHECK(hipFree(d_x));
    bk.affine = false;      // curved: the metric is streamed
  }
}

// d4est_operators_apply_slicer / _apply_lift (src/dGMath/d4est_operators.c:1521-1582, :1454-1519), batched: the trace of a volume
// field on face f of every element (N^2 values per element, face axes in increasing order, first axis fastest) and its inverse
// scatter (zero elsewhere).  Face vectors are element-ordered with stride sum_{e' < e} N_{e'}^2.
__global__ __launch_bounds__(256) void slicer_lift_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                          const int* __restrict__ elem_ids, const int* __restrict__ ns_list,
                                                          const int* __restrict__ face_stride, int n_bucket, int N, int face, int lift) {
  const int dir = face >> 1, fix = (face & 1) ? N - 1 : 0, N2 = N * N, N3 = N2 * N;
  for (int ei = blockIdx.x; ei < n_bucket; ei += gridDim.x) {
    const int ns = ns_list[ei], fs = face_stride[elem_ids[ei]];
    if (lift) {
      for (int idx = threadIdx.x; idx < N3; idx += blockDim.x) {
        const int i = idx % N, j = (idx / N) % N, k = idx / N2;
        const int pos = dir == 0 ? i : (dir == 1 ? j : k);
        const int a = dir == 0 ? j : i, b = dir == 2 ? j : k;
        out[ns + idx] = (pos == fix) ? in[fs + a + N * b] : 0.0;
      }
    } else {
      for (int ab = threadIdx.x; ab < N2; ab += blockDim.x) {
        const int a = ab % N, b = ab / N;
        const int v = dir == 0 ? fix + N * (a + N * b) : (dir == 1 ? a + N * (fix + N * b) : a + N * (b + N * fix));
        out[fs + ab] = in[ns + v];
      }
    }
  }
}

void launch_slicer_lift(d4est_hip_plan* plan, const double* in, double* out, int face, int lift) {
  if (face < 0 || face > 5) D4EST_HIP_ABORT("apply_slicer / apply_lift: face %d", face);
  if (!plan->d_face_stride) {
    std::vector<int> fs((size_t)plan->n_elements + 1, 0);
    for (int e = 0; e < plan->n_elements; ++e) fs[e + 1] = fs[e] + (plan->deg[e] + 1) * (plan->deg[e] + 1);
    plan->face_nodes = fs[plan->n_elements];
    HIP_CHECK(hipMalloc(&plan->d_face_stride, fs.size() * sizeof(int)));
    HIP_CHECK(hipMemcpy(plan->d_face_stride, fs.data(), fs.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  for (const Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    const int grid = bk.n_elem < 16384 ? bk.n_elem : 16384;
    hipLaunchKernelGGL(slicer_lift_kernel, dim3(grid), dim3(256), 0, plan->stream, in, out, plan->d_elem_ids + bk.elem_offset,
                       plan->d_ns_list + bk.elem_offset, plan->d_face_stride, bk.n_elem, bk.N, face, lift);
  }
  HIP_CHECK(hipGetLastError());
}

void launch_dij(d4est_hip_plan* plan, const double* in, double* out, int dir, int transpose) {
  if (dir < 0 || dir > 2) D4EST_HIP_ABORT("apply_dij: direction %d", dir);
  if (in == out) D4EST_HIP_ABORT("apply_dij: in and out must not alias");
  for (const Bucket& bk : plan->buckets) {
    if (bk.n_elem == 0) continue;
    const int grid = bk.n_elem < 16384 ? bk.n_elem : 16384;
    hipLaunchKernelGGL(generic_dij_kernel, dim3(grid), dim3(256), 0, plan->stream, in, out, plan->d_ns_list + bk.elem_offset,
                       bk.n_elem, bk.N, bk.d_D, dir, transpose);
  }
  HIP_CHECK(hipGetLastError());
}

// J and d(rst)/d(xyz) out of d(xyz)/d(rst) at every quadrature node, in place: a[(3 i + j) nq + n] holds dx_i/dr_j on entry and
// dr_i/dx_j on exit (the reference's rst_xyz_quad layout).  d4est_geometry_compute_jacobian / _drst_dxyz,
// src/Geometry/d4est_geometry.c:877-976, same cofactor expressions.
__global__ __launch_bounds__(256) void jacobian_inverse_kernel(double* __restrict__ a, double* __restrict__ jac, size_t nq) {
  for (size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x; n < nq; n += (size_t)gridDim.x * blockDim.x) {
    const double xr = a[0 * nq + n], xs = a[1 * nq + n], xt = a[2 * nq + n];
    const double yr = a[3 * nq + n], ys = a[4 * nq + n], yt = a[5 * nq + n];
    const double zr = a[6 * nq + n], zs = a[7 * nq + n], zt = a[8 * nq + n];
    const double J = xr * (ys * zt - zs * yt) - yr * (xs * zt - zs * xt) + zr * (xs * yt - ys * xt);
    jac[n] = J;
    a[0 * nq + n] = (ys * zt - zs * yt) / J;    // rx
    a[1 * nq + n] = -(xs * zt - zs * xt) / J;   // ry
    a[2 * nq + n] = (xs * yt - ys * xt) / J;    // rz
    a[3 * nq + n] = -(yr * zt - zr * yt) / J;   // sx
    a[4 * nq + n] = (xr * zt - zr * xt) / J;    // sy
    a[5 * nq + n] = -(xr * yt - yr * xt) / J;   // sz
    a[6 * nq + n] = (yr * zs - zr * ys) / J;    // tx
    a[7 * nq + n] = -(xr * zs - zr * xs) / J;   // ty
    a[8 * nq + n] = (xr * ys - yr * xs) / J;    // tz
  }
}

// GEOM_COMPUTE_NUMERICAL volume factors (src/Mesh/d4est_mesh.c:2637-2671): dx_d/dr_d1 = interpolate(D_d1 x_d) at the quadrature
// nodes, then J and the inverse; d_xyz = x | y | z at the Lobatto nodes (3 local_nodes).  Fills plan->d_J and the metric.
void launch_numerical_geometry(d4est_hip_plan* plan, const double* d_xyz) {
  const size_t ln = (size_t)plan->local_nodes, nq = (size_t)plan->local_nodes_quad;
  double *d_dr = nullptr, *d_a = nullptr;
  HIP_CHECK(hipMalloc(&d_dr, std::max<size_t>(3 * ln, 1) * sizeof(double)));
  HIP_CHECK(hipMalloc(&d_a, std::max<size_t>(9 * nq, 1) * sizeof(double)));
  for (int d = 0; d < 3; ++d) {
    launch_dudr(plan, d_xyz + d * ln, d_dr, d_dr + ln, d_dr + 2 * ln);
    for (int d1 = 0; d1 < 3; ++d1) launch_mass_like(plan, 2, d_dr + d1 * ln, d_a + (size_t)(3 * d + d1) * nq);
  }
  if (nq > 0) {
    const int grid = (int)std::min<size_t>((nq + 255) / 256, 65536);
    hipLaunchKernelGGL(jacobian_inverse_kernel, dim3(grid), dim3(256), 0, plan->stream, d_a, plan->d_J, nq);
    HIP_CHECK(hipGetLastError());
  }
  launch_metric_precombine(plan, plan->d_J, d_a);   // synchronises the stream
  HIP_CHECK(hipFree(d_dr));
  HIP_CHECK(hipFree(d_a));
}

void launch_metric_precombine(d4est_hip_plan* plan, const double* d_J, const double* d_rst) {
  if (!plan->d_metric_affine) {
    HIP_CHECK(hipMalloc(&plan->d_metric_affine, std::max<size_t>((size_t)6 * plan->n_elements, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc(&plan->d_nonaffine, std::max<size_t>(plan->buckets.size(), 1) * sizeof(int)));
  }
  HIP_CHECK(hipMemsetAsync(plan->d_nonaffine, 0, std::max<size_t>(plan->buckets.size(), 1) * sizeof(int), plan->stream));
  for (size_t bi = 0; bi < plan->buckets.size(); ++bi) {
    const Bucket& bk = plan->buckets[bi];
    if (bk.n_elem == 0) continue;
    const int grid = bk.n_elem < 4096 ? bk.n_elem : 4096;
    hipLaunchKernelGGL(metric_precombine_kernel, dim3(grid), dim3(256), 0, plan->stream, d_J, d_rst,
                       (size_t)plan->local_nodes_quad, plan->d_qs_list + bk.elem_offset, bk.n_elem, bk.NQ, bk.d_w,
                       plan->d_metric, plan->d_metric_affine + (size_t)6 * bk.elem_offset, plan->d_nonaffine + bi);
  }
  HIP_CHECK(hipGetLastError());
  // affine buckets (every element has a constant J (dr/dx)(dr/dx)^T): the stiffness kernel can rebuild the metric from
  // 6 numbers per element and the 1-D weights instead of streaming 48 B per node (set-up time: one small read-back)
  std::vector<int> flags(plan->buckets.size(), 1);
  HIP_CHECK(hipStreamSynchronize(plan->stream));
  if (!flags.empty()) HIP_CHECK(hipMemcpy(flags.data(), plan->d_nonaffine, flags.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t b = 0; b < plan->buckets.size(); ++b) plan->buckets[b].affine = (flags[b] == 0) && plan->buckets[b].n_elem > 0;
}

}  // namespace d4est_hip
