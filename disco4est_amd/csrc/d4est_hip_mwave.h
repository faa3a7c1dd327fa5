// The multi-wave sum-factorised stiffness apply of ONE element (NQ*NQ threads of a workgroup, p = 8 ... 19) as a device function:
// the body of stiffness_wave_kernel (d4est_hip_volume.hip), shared with the whole-operator kernel of d4est_hip_direct_mw.hip.
// Replaces d4est_quadrature_apply_stiffness_matrix (src/Quadrature/d4est_quadrature.c:263-382) for one element.
#pragma once
#include "d4est_hip_wave.h"

namespace d4est_hip {

// ---- even-odd form of the same contractions (see stiffness_wave_eo_kernel below for the derivation and table layout):
// tab = EO table of the operator, C/2 rows of R doubles, row = [first half | second half]
// which EO contraction form a (C, R) pair uses: measured on MI355X with tools/sweep_p.py (GDoF/s pipelined | hoisted):
// p=5 75.6|76.8, 7 81.5|83.1, 9 62.8|63.6, 11 51.1|48.7, 13 44.1|40.9, 15 39.9|46.8, 17 34.1|26.9, 19 38.7|18.5
// (odd sizes, round 2: threshold 9 / 12 / 14 -> p = 10: 51.9 / 50.5 / 50.7, p = 12: 53.1 / 52.0 / 48.5 GDoF/s: 12 stays)
template <int C, int R>
constexpr bool kEoPipelined = ((C > R ? C : R) >= 12) && ((C > R ? C : R) != 16);
// full EO rows through one base pointer with immediate offsets (contract_rows_eo_imm): where a row (R doubles) and its successor fit
// the scalar registers
#ifndef D4EST_HIP_EO_ROWS_IMM_MAX
#define D4EST_HIP_EO_ROWS_IMM_MAX 20
#endif
#ifndef D4EST_HIP_EO_ROWS_IMM_MIN
#define D4EST_HIP_EO_ROWS_IMM_MIN 9
#endif
template <int C, int R>
constexpr bool kEoRowsImm = ((C > R ? C : R) <= D4EST_HIP_EO_ROWS_IMM_MAX) && ((C > R ? C : R) >= D4EST_HIP_EO_ROWS_IMM_MIN);


// y (+)= M x for a centro-symmetric (ANTI = false) or centro-antisymmetric (ANTI = true) operator M (R x C, both even).
// Each half of the EO rows is contracted in chunks of <= 8 outputs by the software-pipelined contract_single (two scalar
// rows in flight, sched_barrier per step): without the pipelining barriers the compiler hoists every row load to the top and
// spills 1000+ SGPRs at N >= 16.
template <int C, int R, bool ANTI, bool ACC>
__device__ __forceinline__ void apply_eo(const double* __restrict__ tab, const double* x, double* y) {
  constexpr int HC = (C + 1) / 2, P1 = (R + 1) / 2, P2 = R / 2;   // table rows; outputs of the first / second input combination
  double xe[HC], xo[HC], ab[R];
  eo_pre<C>(x, xe, xo);
  const double* xf = ANTI ? xo : xe;  // first part multiplies xe (symmetric) / xo (antisymmetric)
  const double* xs = ANTI ? xe : xo;
  if constexpr (kEoRowsImm<C, R>) {
    contract_rows_eo_imm<HC, R, false>(tab, xf, xs, ab);
  } else if constexpr (kEoPipelined<C, R>) {
    static_assert(P1 <= 24, "apply_eo: row parts longer than 24 are not supported");
    constexpr int A0 = P1 < 8 ? P1 : 8, A1 = (P1 - 8 > 0) ? (P1 - 8 < 8 ? P1 - 8 : 8) : 0, A2 = (P1 - 16 > 0) ? P1 - 16 : 0;
    constexpr int B0 = P2 < 8 ? P2 : 8, B1 = (P2 - 8 > 0) ? (P2 - 8 < 8 ? P2 - 8 : 8) : 0, B2 = (P2 - 16 > 0) ? P2 - 16 : 0;
    contract_single<HC, A0, false, R>(tab, xf, ab);
    if constexpr (A1 > 0) contract_single<HC, A1, false, R>(tab + 8, xf, ab + 8);
    if constexpr (A2 > 0) contract_single<HC, A2, false, R>(tab + 16, xf, ab + 16);
    if constexpr (B0 > 0) contract_single<HC, B0, false, R>(tab + P1, xs, ab + P1);
    if constexpr (B1 > 0) contract_single<HC, B1, false, R>(tab + P1 + 8, xs, ab + P1 + 8);
    if constexpr (B2 > 0) contract_single<HC, B2, false, R>(tab + P1 + 16, xs, ab + P1 + 16);
  } else {
    // free scheduling: the compiler hoists the row loads (deep memory-level parallelism, at the price of SGPR spills)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const double* xx = half == 0 ? xf : xs;
      const int len = half == 0 ? P1 : P2;
#pragma unroll
      for (int o0 = 0; o0 < len; o0 += 8) {
#pragma unroll
        for (int c = 0; c < HC; ++c) {
          sdouble_ptr row = launder(tab + c * R + half * P1 + o0);
#pragma unroll
          for (int o = 0; o < 8; ++o)
            if (o0 + o < len) ab[half * P1 + o0 + o] = (c == 0) ? row[o] * xx[0] : fma(row[o], xx[c], ab[half * P1 + o0 + o]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R / 2; ++r) {
    const double p = ab[r] + ab[P1 + r], m = ab[r] - ab[P1 + r];
    y[r] = ACC ? y[r] + p : p;
    y[R - 1 - r] = ACC ? y[R - 1 - r] + m : m;
  }
  if constexpr (R % 2 != 0) y[R / 2] = ACC ? y[R / 2] + ab[R / 2] : ab[R / 2];
}
// y = op x (operator given transposed, or as EO table when EO); y (+)= op^T x (operator itself, or EO table of op^T)
template <int NI, int NO, bool EO, bool ANTI>
__device__ __forceinline__ void fwd(const double* __restrict__ tab, const double* x, double* y) {
  if constexpr (EO) apply_eo<NI, NO, ANTI, false>(tab, x, y);
  else contract_n<NI, NO>(tab, x, y);
}
template <int NI, int NO, bool EO, bool ANTI, bool ACC>
__device__ __forceinline__ void bwd(const double* __restrict__ tab, const double* x, double* y) {
  if constexpr (EO) apply_eo<NI, NO, ANTI, ACC>(tab, x, y);
  else contract_t<NI, NO, ACC>(tab, x, y);
}

#ifndef D4EST_HIP_METRIC_DEPTH
#define D4EST_HIP_METRIC_DEPTH 4
#endif
#ifndef D4EST_HIP_METRIC_EARLY
#define D4EST_HIP_METRIC_EARLY 2
#endif
#ifndef D4EST_HIP_MW_COLLOCATED
#define D4EST_HIP_MW_COLLOCATED 1   /* deg_quad = deg: the collocated-gradient form (12 instead of 16 one-dimensional products per thread) */
#endif
#ifndef D4EST_HIP_MW_WAVES
#define D4EST_HIP_MW_WAVES 4
#endif
// On entry the workgroup's threads have written u_e to R0 as [i + PN (j + N k)] (no barrier yet); on exit R0 holds (A u)_e in the same
// layout, behind a barrier.  R0, R1: WaveCfg<N, NQ>::FS doubles each.  te = thread within the element, (a, b) = (te % NQ, te / NQ);
// ns-independent: the caller loads and stores the element.  Operator arguments as stiffness_wave_kernel's.
// MASS: + V^T [ w J c (V u) ], the zeroth-order term of a linearised nonlinear problem (see stiffness_wave_eo_element, d4est_hip_wave.h)
template <int N, int NQ, bool PF, bool EO, bool AFF, bool MASS = false, bool NT = false>
__device__ __forceinline__ void stiffness_mw_element(double* R0, double* R1, const double* __restrict__ metric, int qs, int ei, bool active,
                                                     int te, int a, int b, const double* __restrict__ Bop, const double* __restrict__ Gop,
                                                     const double* __restrict__ BopT, const double* __restrict__ GopT,
                                                     const double* __restrict__ affine, const double* __restrict__ wq,
                                                     const double* __restrict__ cq = nullptr, unsigned long long* stamps = nullptr) {
  // (stamps: diagnostic builds only -- s_memtime at the stage boundaries, tools/stamps_mw.py)
#define MWE_STAMP(k) do { if (stamps && te == 0) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
  using C = WaveCfg<N, NQ>;
  constexpr int PL = C::PL, PN = C::PN, PQ = C::PQ;
  constexpr int NQ3 = NQ * NQ * NQ;
  double mreg[PF ? 6 : 1][PF ? NQ : 1];
  if (PF && active) {
    const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq)
#pragma unroll
      for (int c = 0; c < 6; ++c) mreg[c][kq] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * kq]);
  }
  __syncthreads();

  // ---- S1: r-contraction, thread (j=a, k=b); u column -> registers, then R0 <- B u, R1 <- G u as [k][iq][j]
  {
    double x[N], br[NQ], gr[NQ];
    const bool on = active && a < N && b < N;
    if (on) {
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = lds_ld(&R0[i + PN * (a + N * b)]);
      fwd<N, NQ, EO, false>(BopT, x, br);
      fwd<N, NQ, EO, true>(GopT, x, gr);
    }
    __syncthreads();
    if (on) {
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        R0[a + PN * (iq + NQ * b)] = br[iq];
        R1[a + PN * (iq + NQ * b)] = gr[iq];
      }
    }
  }
  __syncthreads();

  MWE_STAMP(0);   // element image + S1
  // ---- S2 (thread (iq=a, k=b)) interleaved with S3 (thread (iq=a, jq=b)): one field at a time through R0
  double gr[NQ], gs[NQ], gt[NQ];
  double vm[MASS ? NQ : 1];   // MASS: V u, then w J c V u, at the thread's quadrature nodes
  // Multi-wave general path: the thread's 6 NQ metric values are requested MD quadrature planes ahead of their use.  Left to itself
  // the compiler requests one plane's six values, waits for all of them, multiplies, requests the next: NQ serialised memory
  // round trips in the middle of the element (12 at p = 11 -- most of an element's lifetime).  The registers for the planes in
  // flight come from gr and gs, which wait in the (then idle) LDS fields in thread-private slots during this stage.
  constexpr bool kPark = !PF && !AFF && C::THREADS > 64;
  constexpr int MDW = MASS ? 2 : D4EST_HIP_METRIC_DEPTH;   // (the mass term's line takes the registers of two planes in flight)
  constexpr int MD = (MDW < NQ) ? MDW : NQ;
  constexpr int MEW = MASS ? 0 : D4EST_HIP_METRIC_EARLY;
  constexpr int ME = (MEW < MD) ? MEW : MD;   // planes requested before the last forward contraction
  double mw[kPark ? NQ : 1][6];
  {
    double x1[N], x2[N], t[NQ], y[N];
    const bool on2 = active && b < N;
    if (on2) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        x1[j] = lds_ld(&R0[j + PN * (a + NQ * b)]);  // B_r u
        x2[j] = lds_ld(&R1[j + PN * (a + NQ * b)]);  // G_r u
      }
    }
    __syncthreads();
    // field 1: B_s G_r u  -> gr = B_t(.)
    if (on2) {
      fwd<N, NQ, EO, false>(BopT, x2, t);
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) R0[b + PN * (a + NQ * jq)] = t[jq];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int k = 0; k < N; ++k) y[k] = lds_ld(&R0[k + PN * (a + NQ * b)]);
      fwd<N, NQ, EO, false>(BopT, y, gr);
    }
    // field 2: G_s B_r u  -> gs = B_t(.)   (goes through R1 so the two transfers overlap)
    if (on2) {
      fwd<N, NQ, EO, true>(GopT, x1, t);
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) R1[b + PN * (a + NQ * jq)] = t[jq];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int k = 0; k < N; ++k) y[k] = lds_ld(&R1[k + PN * (a + NQ * b)]);
      fwd<N, NQ, EO, false>(BopT, y, gs);
    }
    // field 3: B_s B_r u  -> gt = G_t(.)
    if (on2) {
      fwd<N, NQ, EO, false>(BopT, x1, t);
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) R0[b + PN * (a + NQ * jq)] = t[jq];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int k = 0; k < N; ++k) y[k] = lds_ld(&R0[k + PN * (a + NQ * b)]);
      if constexpr (kPark) {
        // R1 is free from here on (every thread read field 2 before the barrier above): the finished line gr waits there, in the
        // thread's own slots [kq][te] (conflict-free), while the registers it leaves carry metric planes in flight
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) R1[kq * PL + te] = gr[kq];
        const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
        for (int kq = 0; kq < ME; ++kq)
#pragma unroll
          for (int c = 0; c < 6; ++c) mw[kq][c] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * kq]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!(MASS && kPark)) fwd<N, NQ, EO, true>(GopT, y, gt);
      if constexpr (MASS && !kPark) fwd<N, NQ, EO, false>(BopT, y, vm);
    }
    if constexpr (MASS && kPark) {
      // the mass term's line needs the registers of gs as well: every thread has read field 3, so R0 is free and gs waits there
      // BEFORE the last two forward contractions (without the mass term it goes there after them, see the quadrature stage)
      __syncthreads();
      if (active) {
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) R0[kq * PL + te] = gs[kq];
        fwd<N, NQ, EO, true>(GopT, y, gt);
        fwd<N, NQ, EO, false>(BopT, y, vm);
      }
    }
  }

  MWE_STAMP(1);   // S2 / S3
  // ---- quadrature-point stage
  if constexpr (MASS) {
    if (active) {
      const double* __restrict__ cp = cq + qs + (a + NQ * b);   // w J c, pre-combined (ensure_lhs_wjc)
      double cv[NQ];
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) cv[kq] = cp[NQ * NQ * kq];
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) vm[kq] *= cv[kq];
    }
  }
  if constexpr (AFF) {
    if (active) {
    const double* __restrict__ c = affine + (size_t)6 * ei;
    const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
    const double wab = wq[b] * wq[a];
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {
      const double w3 = wq[kq] * wab;
      const double r = w3 * gr[kq], s_ = w3 * gs[kq], t = w3 * gt[kq];
      gr[kq] = c0 * r + c1 * s_ + c2 * t;
      gs[kq] = c1 * r + c3 * s_ + c4 * t;
      gt[kq] = c2 * r + c4 * s_ + c5 * t;
    }
    }
  } else {
    if constexpr (kPark && !MASS) __syncthreads();   // every thread has read field 3: R0 is free as well
    if (active) {
    const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
    if constexpr (kPark) {
      if constexpr (!MASS) {
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) R0[kq * PL + te] = gs[kq];
      }
#pragma unroll
      for (int kq = ME; kq < MD; ++kq)
#pragma unroll
        for (int c = 0; c < 6; ++c) mw[kq][c] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * kq]);
      double rn = lds_ld(&R1[te]), sn = lds_ld(&R0[te]);
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) {
        if (kq + MD < NQ) {
#pragma unroll
          for (int c = 0; c < 6; ++c) mw[kq + MD][c] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * (kq + MD)]);
        }
        const double r = rn, s = sn, t = gt[kq];
        if (kq + 1 < NQ) {
          rn = lds_ld(&R1[(kq + 1) * PL + te]);
          sn = lds_ld(&R0[(kq + 1) * PL + te]);
        }
        __builtin_amdgcn_sched_barrier(0);
        R1[kq * PL + te] = mw[kq][0] * r + mw[kq][1] * s + mw[kq][2] * t;
        R0[kq * PL + te] = mw[kq][1] * r + mw[kq][3] * s + mw[kq][4] * t;
        gt[kq] = mw[kq][2] * r + mw[kq][4] * s + mw[kq][5] * t;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) {
        gr[kq] = lds_ld(&R1[kq * PL + te]);
        gs[kq] = lds_ld(&R0[kq * PL + te]);
      }
    } else {
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
    }
    }
  }

  MWE_STAMP(2);   // quadrature stage (the metric stream)
  // ---- S5 (thread (iq=a, jq=b), registers) interleaved with S6 (thread (iq=a, k=b)) through R0/R1
  double ar[N], bs[N];
  {
    double c[N], x[NQ];
    const bool on6 = active && b < N;
    __syncthreads();
    if (active) {
      bwd<NQ, N, EO, false, false>(Bop, gr, c);
#pragma unroll
      for (int k = 0; k < N; ++k) R0[b + PQ * (a + NQ * k)] = c[k];  // [k][iq][jq]
    }
    __syncthreads();
    if (on6) {
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R0[jq + PQ * (a + NQ * b)]);
      bwd<NQ, N, EO, false, false>(Bop, x, ar);
    }
    if (active) {
      bwd<NQ, N, EO, false, false>(Bop, gs, c);
#pragma unroll
      for (int k = 0; k < N; ++k) R1[b + PQ * (a + NQ * k)] = c[k];
    }
    __syncthreads();
    if (on6) {
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R1[jq + PQ * (a + NQ * b)]);
      bwd<NQ, N, EO, true, false>(Gop, x, bs);
    }
    if (active) {
      bwd<NQ, N, EO, true, false>(Gop, gt, c);
      if constexpr (MASS) bwd<NQ, N, EO, false, true>(Bop, vm, c);   // both continue through B_s^T B_r^T
#pragma unroll
      for (int k = 0; k < N; ++k) R0[b + PQ * (a + NQ * k)] = c[k];
    }
    __syncthreads();
    if (on6) {
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) x[jq] = lds_ld(&R0[jq + PQ * (a + NQ * b)]);
      bwd<NQ, N, EO, false, true>(Bop, x, bs);
    }
    __syncthreads();
    if (on6) {
#pragma unroll
      for (int j = 0; j < N; ++j) {  // [k][j][iq]
        R0[a + PQ * (j + N * b)] = ar[j];
        R1[a + PQ * (j + N * b)] = bs[j];
      }
    }
  }
  __syncthreads();

  MWE_STAMP(3);   // S5 / S6
  // ---- S7: r-contraction transposed, thread (j=a, k=b)
  {
    double x[NQ], y[NQ], o[N];
    const bool on = active && a < N && b < N;
    if (on) {
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        x[iq] = lds_ld(&R0[iq + PQ * (a + N * b)]);
        y[iq] = lds_ld(&R1[iq + PQ * (a + N * b)]);
      }
      bwd<NQ, N, EO, true, false>(Gop, x, o);
      bwd<NQ, N, EO, false, true>(Bop, y, o);
    }
    __syncthreads();
    if (on) {
#pragma unroll
      for (int i = 0; i < N; ++i) R0[i + PN * (a + N * b)] = o[i];
    }
  }
  __syncthreads();
  MWE_STAMP(4);   // S7
#undef MWE_STAMP
}

// ---------------------------------------------------------------------------
// The same apply in the COLLOCATED-GRADIENT form, for deg_quad = deg (N = NQ; round 3).  With as many quadrature nodes as Lobatto nodes
// the interpolation B is square and invertible, so the gradient at the quadrature nodes is the derivative of the interpolant THERE:
//     (B_t (x) B_s (x) G_r) u = (I (x) I (x) Dq) (B_t (x) B_s (x) B_r) u,     G = B D,  Dq = B D B^-1 = the differentiation matrix ON the
// quadrature nodes (Tables1D::quad_diff).  One interpolation of u (3 one-dimensional products per thread), three derivatives of it
// (3), the metric multiply, three transposed derivatives summed into one field (3), one transposed interpolation (3): 12 products per
// thread instead of the 16 of the form above (8 forward: B_r G_r | B_s G_r, G_s B_r, B_s B_r | B_t B_t G_t; 8 backward) -- the same
// operator, re-associated (differences at rounding level, held to the oracle by the same tests).  The two gradient lines that are formed
// by other threads arrive through LDS in exactly the thread-private slots [kq + P te] in which the quadrature stage keeps them while
// the metric planes are in flight, so the parking of the form above costs nothing here.  V u itself passes through registers (w below):
// the zeroth-order term (MASS) needs no contraction of its own in either direction.
// EDq / EDqT: even-odd tables of Dq and of its transpose (centro-antisymmetric); Bop / BopT as above.  On entry R0 holds u_e (no barrier
// yet); on exit R0 holds (A u)_e behind a barrier.
// ---------------------------------------------------------------------------
template <int N, int NQ, bool PF, bool EO>
inline constexpr bool kMwCollocated = D4EST_HIP_MW_COLLOCATED && N == NQ && EO && !PF && (N * N > 64);

template <int N, bool AFF, bool MASS = false, bool NT = false>
__device__ __forceinline__ void stiffness_mw_element_cg(double* R0, double* R1, const double* __restrict__ metric, int qs, int ei, bool active,
                                                        int te, int a, int b, const double* __restrict__ Bop, const double* __restrict__ BopT,
                                                        const double* __restrict__ EDq, const double* __restrict__ EDqT,
                                                        const double* __restrict__ affine, const double* __restrict__ wq,
                                                        const double* __restrict__ cq = nullptr, unsigned long long* stamps = nullptr) {
#define MWE_STAMP(k) do { if (stamps && te == 0) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
  using C = WaveCfg<N, N>;
  constexpr int P = C::PN;
  constexpr int N3 = N * N * N;
  const int line = P * te;   // the thread's private line [kq + P (a + N b)] in either field
  // ---- F1 (r): thread (j = a, k = b):  R1[j + P (iq + N k)] <- B u
  __syncthreads();
  if (active) {
    double x[N], y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = lds_ld(&R0[i + line]);
    fwd<N, N, true, false>(BopT, x, y);
#pragma unroll
    for (int iq = 0; iq < N; ++iq) R1[a + P * (iq + N * b)] = y[iq];
  }
  // ---- F2 (s): thread (iq = a, k = b):  R0[k + P (iq + N jq)] <- B (.)
  __syncthreads();
  if (active) {
    double x[N], y[N];
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = lds_ld(&R1[j + line]);
    fwd<N, N, true, false>(BopT, x, y);
#pragma unroll
    for (int jq = 0; jq < N; ++jq) R0[b + P * (a + N * jq)] = y[jq];
  }
  MWE_STAMP(0);
  // ---- F3 (t): thread (iq = a, jq = b): w = V u along kq (registers); gt = Dq w; w goes out in the layouts of the s- and r-derivative
  __syncthreads();
  double gt[N];
  double vm[MASS ? N : 1];
  {
    double w[N];
    if (active) {
      double x[N];
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] = lds_ld(&R0[k + line]);
      fwd<N, N, true, false>(BopT, x, w);
#pragma unroll
      for (int kq = 0; kq < N; ++kq) R1[a + P * (b + N * kq)] = w[kq];   // r-lines: [iq + P (jq + N kq)]  (R1's readers finished before the barrier above)
      fwd<N, N, true, true>(EDq, w, gt);
      if constexpr (MASS) {
#pragma unroll
        for (int kq = 0; kq < N; ++kq) vm[kq] = w[kq];
      }
    }
    __syncthreads();   // every thread has read its line of R0  (the barrier is outside the branch: a wavefront may hold idle threads)
    if (active) {
#pragma unroll
      for (int kq = 0; kq < N; ++kq) R0[b + P * (a + N * kq)] = w[kq];   // s-lines: [jq + P (iq + N kq)]
    }
  }
  // ---- F4: thread (iq = a, kq = b): gs line = Dq (s-line);  thread (jq = a, kq = b): gr line = Dq (r-line); both go to the private
  // slots [kq + P (iq + N jq)] of the threads (iq, jq)
  __syncthreads();
  {
    double z[N], x[N], ys[N], yr[N];
    if (active) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        z[j] = lds_ld(&R0[j + line]);
        x[j] = lds_ld(&R1[j + line]);
      }
      fwd<N, N, true, true>(EDq, z, ys);
      fwd<N, N, true, true>(EDq, x, yr);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        R0[b + P * (a + N * j)] = ys[j];   // gs(iq = a, jq = j, kq = b)  -> [kq + P (iq + N jq)]
        R1[b + P * (j + N * a)] = yr[j];   // gr(iq = j, jq = a, kq = b)  -> [kq + P (iq + N jq)]
      }
    }
  }
  __syncthreads();
  MWE_STAMP(1);
  // ---- quadrature-point stage: thread (iq = a, jq = b); gs in R0[kq + line], gr in R1[kq + line] (private), gt in registers
  if constexpr (MASS) {
    if (active) {
      const double* __restrict__ cp = cq + qs + (a + N * b);   // w J c, pre-combined (ensure_lhs_wjc)
      double cv[N];
#pragma unroll
      for (int kq = 0; kq < N; ++kq) cv[kq] = cp[N * N * kq];
#pragma unroll
      for (int kq = 0; kq < N; ++kq) vm[kq] *= cv[kq];
    }
  }
  if (active) {
    if constexpr (AFF) {
      const double* __restrict__ c = affine + (size_t)6 * ei;
      const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
      const double wab = wq[b] * wq[a];
#pragma unroll
      for (int kq = 0; kq < N; ++kq) {
        const double w3 = wq[kq] * wab;
        const double r = w3 * lds_ld(&R1[kq + line]), s_ = w3 * lds_ld(&R0[kq + line]), t = w3 * gt[kq];
        R1[kq + line] = c0 * r + c1 * s_ + c2 * t;
        R0[kq + line] = c1 * r + c3 * s_ + c4 * t;
        gt[kq] = c2 * r + c4 * s_ + c5 * t;
      }
    } else {
      constexpr int MDW = MASS ? 2 : D4EST_HIP_METRIC_DEPTH;
      constexpr int MD = (MDW < N) ? MDW : N;
      double mw[N][6];
      const double* __restrict__ m = metric + (size_t)6 * qs + (a + N * b);
#pragma unroll
      for (int kq = 0; kq < MD; ++kq)
#pragma unroll
        for (int c = 0; c < 6; ++c) mw[kq][c] = ld_sel<NT>(&m[c * N3 + N * N * kq]);
      double rn = lds_ld(&R1[line]), sn = lds_ld(&R0[line]);
#pragma unroll
      for (int kq = 0; kq < N; ++kq) {
        if (kq + MD < N) {
#pragma unroll
          for (int c = 0; c < 6; ++c) mw[kq + MD][c] = ld_sel<NT>(&m[c * N3 + N * N * (kq + MD)]);
        }
        const double r = rn, s = sn, t = gt[kq];
        if (kq + 1 < N) {
          rn = lds_ld(&R1[kq + 1 + line]);
          sn = lds_ld(&R0[kq + 1 + line]);
        }
        __builtin_amdgcn_sched_barrier(0);
        R1[kq + line] = mw[kq][0] * r + mw[kq][1] * s + mw[kq][2] * t;
        R0[kq + line] = mw[kq][1] * r + mw[kq][3] * s + mw[kq][4] * t;
        gt[kq] = mw[kq][2] * r + mw[kq][4] * s + mw[kq][5] * t;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  MWE_STAMP(2);
  // ---- B1: the transposed derivatives.  t: registers.  s / r: the threads (iq, kq) / (jq, kq) take their lines out of the flux fields
  // (stride P N) and put the result back IN PLACE (each thread rewrites exactly the entries it read)
  double ft[N];
  if (active) {
    bwd<N, N, true, true, false>(EDqT, gt, ft);
    if constexpr (MASS) {
#pragma unroll
      for (int kq = 0; kq < N; ++kq) ft[kq] += vm[kq];
    }
  }
  __syncthreads();
  {
    double z[N], x[N], ys[N], yr[N];
    if (active) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        z[j] = lds_ld(&R0[b + P * (a + N * j)]);   // flux_s(iq = a, jq = j, kq = b)
        x[j] = lds_ld(&R1[b + P * (j + N * a)]);   // flux_r(iq = j, jq = a, kq = b)
      }
      bwd<N, N, true, true, false>(EDqT, z, ys);
      bwd<N, N, true, true, false>(EDqT, x, yr);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        R0[b + P * (a + N * j)] = ys[j];
        R1[b + P * (j + N * a)] = yr[j];
      }
    }
  }
  __syncthreads();
  // ---- B2 (t^T): thread (iq = a, jq = b): F = ft + (s part) + (r part) along kq;  R?[jq + P (iq + N k)] <- B^T F
  {
    double c[N];
    if (active) {
      double F[N];
#pragma unroll
      for (int kq = 0; kq < N; ++kq) F[kq] = ft[kq] + (lds_ld(&R0[kq + line]) + lds_ld(&R1[kq + line]));
      bwd<N, N, true, false, false>(Bop, F, c);
    }
    __syncthreads();   // every thread has read its private lines
    if (active) {
#pragma unroll
      for (int k = 0; k < N; ++k) R0[b + P * (a + N * k)] = c[k];
    }
  }
  MWE_STAMP(3);
  // ---- B3 (s^T): thread (iq = a, k = b):  R1[iq + P (j + N k)] <- B^T (.)
  __syncthreads();
  if (active) {
    double x[N], y[N];
#pragma unroll
    for (int jq = 0; jq < N; ++jq) x[jq] = lds_ld(&R0[jq + line]);
    bwd<N, N, true, false, false>(Bop, x, y);
#pragma unroll
    for (int j = 0; j < N; ++j) R1[a + P * (j + N * b)] = y[j];
  }
  // ---- B4 (r^T): thread (j = a, k = b):  R0[i + P (j + N k)] <- B^T (.)
  __syncthreads();
  if (active) {
    double x[N], o[N];
#pragma unroll
    for (int iq = 0; iq < N; ++iq) x[iq] = lds_ld(&R1[iq + line]);
    bwd<N, N, true, false, false>(Bop, x, o);
#pragma unroll
    for (int i = 0; i < N; ++i) R0[i + line] = o[i];
  }
  __syncthreads();
  MWE_STAMP(4);
#undef MWE_STAMP
}

}  // namespace d4est_hip
