// "Direct" face kernel: the SIPG face terms of a conforming, uniform-degree plan (deg, deg_quad <= 7) WITHOUT the mortar-node
// trace arrays in between.
//
// Replaces, like d4est_hip_faces.hip, d4est_laplacian_flux_interface / _boundary (src/dGMath/d4est_laplacian_flux.c:23-1014) with
// the SIPG callbacks (src/dGMath/d4est_laplacian_flux_sipg.c:15-942) over d4est_mortars_compute_flux_on_local_elements
// (src/Mesh/d4est_mortars.c:601-840) -- same numbers, different data flow.
//
// The two-phase form (trace kernel -> 4 fields per mortar node -> flux kernel) moves 12 KB of traces per p = 7 element out to HBM
// and 24 KB back in (own + neighbour), 36 % of the operator's traffic, and both kernels are LDS-bound: six waves per element, each
// lane re-reading operator rows and data columns from LDS (45 + 70 KB of LDS traffic per element).  Here ONE wavefront owns an
// element and works in the volume kernel's style:
//   * a lane owns a whole LINE (8 doubles in registers); the 1-D operators are wave-uniform and arrive as scalar operands;
//     LDS is only the transposition network between the passes (~450 b64 accesses per lane and element instead of ~1400);
//   * the (+) side's trace is RECOMPUTED from the neighbour's nodal values u (the normal lines of its face: 4 KB per side, served
//     by L2 / Infinity Cache -- u is 16 MB at config 2), so no trace is written or read;
//   * per reference direction d the two faces 2d, 2d+1 give 8 nodal face fields (trace, normal derivative) x (own, neighbour) x 2
//     = 64 rows = one row per lane:
//       pass 1 (rows over a):     P = C x, R = CD x                                 (C: side nodes -> mortar quadrature nodes)
//       pass 2 (columns over b):  u = C P, du/dt_b = CD P, du/dn = C S, du/dt_a = C R   (32 + 32 lanes, two products each)
//       SIPG terms at the mortar node of the lane (the (+) values through the p4est re-ordering of the face pair)
//       lift pass 1 (rows over a'):  Y = E A  (D^T E for the field that is differentiated along a)
//       lift pass 2 (columns over b'): face-local part + the normal term-2 field, which D^T spreads along the normal line
//     and the element's Au line registers are updated through an LDS accumulator in the three line orientations.
// Ghost (+) sides read their mortar-node block from the exchanged ghost trace buffer exactly as the two-phase flux kernel does, so
// multi-rank plans run the trace kernel only to feed the exchange.
#include <algorithm>
#include <type_traits>

#include "d4est_hip_direct.h"
#include "d4est_hip_internal.h"
#include "d4est_hip_tables.h"
#include "d4est_hip_wave.h"

namespace d4est_hip {

// y = M x, M (NO x NI): tab = M transposed (NI x NO row-major), or the even-odd table of M when EO (NI, NO even; ANTI: M is
// centro-antisymmetric) -- see stiffness_wave_eo_kernel for the table layout.  The EO form feeds its scalar operator rows through
// the software-pipelined contract_single_eo (two rows in flight, a scheduling barrier per step): left to itself the compiler
// hoists every row load of a stage and spills hundreds of SGPRs through v_readlane / v_writelane.
#ifndef D4EST_HIP_DIRECT_IMM_ROWS
#define D4EST_HIP_DIRECT_IMM_ROWS 1   /* 1 (default; config 2 apply_aij -3 %, Schwarz sweep -1 %, 11 % fewer scalar instructions): the operator rows through one laundered base pointer with immediate offsets (contract_rows_eo_imm) */
#endif
#if D4EST_HIP_DIRECT_IMM_ROWS
#define D4EST_DIRECT_ROWS contract_rows_eo_imm
#else
#define D4EST_DIRECT_ROWS contract_rows_eo
#endif
template <int NI, int NO, bool EO, bool ANTI>
__device__ __forceinline__ void prod(const double* __restrict__ tab, const double* x, double* y) {
  if constexpr (EO) {
    constexpr int HC = (NI + 1) / 2;
    double xe[HC], xo[HC], ab[NO];
    eo_pre<NI>(x, xe, xo);
    D4EST_DIRECT_ROWS<HC, NO, false>(tab, ANTI ? xo : xe, ANTI ? xe : xo, ab);   // one scalar row per step: 32 SGPRs in flight (the two-row
                                                                                // form of the volume kernel measures the same here and needs 64)
    eo_post<NO>(ab, y);
  } else {
    contract_n<NI, NO>(tab, x, y);
  }
}

// yS = S x and yA = A x for a centro-symmetric S and a centro-antisymmetric A on the same input (one pass over x)
template <int NI, int NO, bool EO>
__device__ __forceinline__ void prod_pair(const double* __restrict__ tabS, const double* __restrict__ tabA, const double* x, double* yS,
                                          double* yA) {
  if constexpr (EO) {
    constexpr int HC = (NI + 1) / 2;
    double xe[HC], xo[HC], abS[NO], abA[NO];
    eo_pre<NI>(x, xe, xo);
    D4EST_DIRECT_ROWS<HC, NO, false>(tabS, xe, xo, abS);
    D4EST_DIRECT_ROWS<HC, NO, false>(tabA, xo, xe, abA);
    eo_post<NO>(abS, yS);
    eo_post<NO>(abA, yA);
  } else {
    contract_n<NI, NO>(tabS, x, yS);
    contract_n<NI, NO>(tabA, x, yA);
  }
}

template <int N>
__device__ __forceinline__ double row_dot(const double* __restrict__ drow, const double* x) {
  sdouble_ptr row = launder(drow);
  double s = row[0] * x[0];
#pragma unroll
  for (int i = 1; i < N; ++i) s = fma(row[i], x[i], s);
  return s;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// LDS layouts of the transposition buffer (doubles).  Every "one line per lane" read walks rows of ODD length (RS, RQ), so the 32
// lanes of a pass hit 32 different bank pairs; the field strides are = 16 dwords mod 64 banks at N = NQ = 8.
template <int N, int NQ>
struct DirectCfg {
  static_assert(NQ >= N && NQ * NQ <= 64, "one wavefront per element: (deg_quad + 1)^2 <= 64");
  static constexpr int N2 = N * N, N3 = N2 * N, T = NQ * NQ, PN = N | 1;
  static constexpr int RS = N | 1, RQ = NQ | 1;       // padded row lengths: lines over a side index / a mortar index
  static constexpr int GS = NQ * RS, YS = N * RQ;     // field strides of the pass-1 outputs [c][a'][b] and [c8][a][b']
  static constexpr int QS = T + 8, VS = N2 + 8;       // mortar values [block][a' + NQ b'], lifted fields [block][a + N b]
  static constexpr int S_DOUBLES = cmax(cmax(cmax(8 * N * RS, 8 * GS), cmax(8 * QS, 8 * NQ * RQ)), cmax(8 * YS, 6 * VS));
  static constexpr int U_DOUBLES = PN * N2;
  static constexpr int OPSZ = N * NQ;
  static constexpr bool FULL = (N == 8 && NQ == 8);   // every lane is a face node, a row and a column: no guards
};

// kDirectWPB wavefronts = elements per workgroup (each wave works alone; fewer, larger workgroups launch faster: 4096
// one-wave workgroups take ~10 us to get going at config 2)
constexpr int kDirectWPB = 4;

#ifndef D4EST_HIP_DIRECT_NT_MASK
#define D4EST_HIP_DIRECT_NT_MASK 7   /* stream mode (VOL & 8) of the one-wavefront kernel: which streams carry the non-temporal hint -- 1 the mortar factors, 2 the volume metric, 4 the A u stores (kernel experiments: tools/build_variant.sh) */
#endif
#ifndef D4EST_HIP_DIRECT_GEOM_EARLY
#define D4EST_HIP_DIRECT_GEOM_EARLY 1   /* both faces' geometric factors requested: 0 at their use, 1 before the SIPG loop, 2 with the neighbour lines */
#endif
// the seven geometric-factor fields of a side at mortar node k (sj n_l / 2-weighted rows 0..5, penalty row 6), or the Robin coefficient
#ifndef D4EST_HIP_DIRECT_GEOM_NT_RT
#define D4EST_HIP_DIRECT_GEOM_NT_RT 1   /* 1: outside stream mode the mortar factors still take the non-temporal hint where the launch asks for it (DirectVol::stream bit 1, a wave-uniform branch around the batch of loads) */
#endif
template <int T, bool NT = false /* stream mode, d4est_hip_wave.h */, bool HANG = false /* kind 3 exists: a side another kernel serves */>
__device__ __forceinline__ void direct_load_geom(double* gq, int kind, bool on, int k, int sgeom, const double* __restrict__ geom,
                                                 const double* __restrict__ robin_c, bool nt_rt = false) {
#pragma unroll
  for (int c = 0; c < 7; ++c) gq[c] = 0.0;
  if (on && !(HANG && kind == 3)) {
    if (kind == 0 && robin_c) {
      gq[6] = robin_c[sgeom + k];   // am = ap = 0: no term 1 / term 2 on a Robin side
    } else {
      const double* __restrict__ g = geom + (size_t)7 * sgeom + k;
      if (!NT && D4EST_HIP_DIRECT_GEOM_NT_RT && nt_rt) {
        // (the pointer goes through an opaque copy: the two arms load the same addresses, and the compiler otherwise hoists ONE load out of
        // them, dropping the hint)
        const double* gn = g;
        asm volatile("" : "+v"(gn));
#pragma unroll
        for (int c = 0; c < 7; ++c) gq[c] = __builtin_nontemporal_load(&gn[c * T]);
      } else {
#pragma unroll
        for (int c = 0; c < 7; ++c) gq[c] = ld_sel<NT>(&g[c * T]);
      }
    }
  }
}

// VOL & 16 (the hybrid operator on a locally refined plan, "hanging-aware"): side kind 3 = a side the mortar-record kernels serve (a big
// hanging side, or a small one they keep): no contribution here; kind 2 also stands for a SMALL hanging side whose (+) block -- the big
// element's sub-mortar trace, written by trace_hp_mfma16_kernel before this kernel -- sits in the plan's trace array (passed as
// ghost_qtrace), and the side's own mortar-node block (u, du/dr_0..2: what the trace kernels would write) is exported to that
// array at nbr_ns, for the record flux kernel of the big element across it, which runs after this kernel.
template <int N, int NQ, bool EO, bool FUSE, int VOL = 0 /* 0 faces only; + volume term: 1 streamed metric, 2 affine metric; + 4: and the zeroth-order term; + 8: stream mode (non-temporal metric / factor loads and A u stores); + 16: hanging-aware */>
__global__ __launch_bounds__(64 * kDirectWPB, 4) void faces_direct_kernel(const double* __restrict__ u, const double* __restrict__ ghost_qtrace,
                                                             double* __restrict__ Au, const DirectSide* __restrict__ sides,
                                                             const DirectGhostOff* __restrict__ ghost_off,
                                                             const double* __restrict__ ops, const double* __restrict__ geom,
                                                             const double* __restrict__ bndry_q, const double* __restrict__ robin_c,
                                                             const double* __restrict__ robin_r, int n_elem, int ns0, int ns_stride,
                                                             int xcd_chunk, DirectFuse cf, DirectVol vol, const int* __restrict__ elem_list) {
  using C = DirectCfg<N, NQ>;
  constexpr int N2 = C::N2, N3 = C::N3, T = C::T, PN = C::PN, RS = C::RS, RQ = C::RQ, GS = C::GS, QS = C::QS, YS = C::YS, VS = C::VS;
  constexpr bool FULL = C::FULL;
  // per wave: [0, L0) the element's u (line reads in the three directions), later the accumulator of the lifted face terms; [L0, L0 + L1)
  // the transposition buffer of every pass (in place: a wave runs in lockstep).  With VOL the two halves are the volume kernel's two fields.
  constexpr int L0 = cmax(C::U_DOUBLES, VOL ? WaveCfg<N, NQ>::FS : 0), L1 = cmax(C::S_DOUBLES, VOL ? WaveCfg<N, NQ>::FS : 0);
  __shared__ double s_L[kDirectWPB][L0 + L1];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* s_U = s_L[wv];
  double* s_S = s_L[wv] + L0;
  // the five tables are addressed from `ops` at every use (one live pointer instead of five: the kernel is short of SGPRs)
#define tC (ops)
#define tCD (ops + C::OPSZ)
#define tE (ops + 2 * C::OPSZ)
#define tDtE (ops + 3 * C::OPSZ)
#define dr0 (ops + 4 * C::OPSZ) /* D[0][:], then D[N-1][:] */
  const int lane = threadIdx.x & 63;   // (the mask tells the compiler the range: without it every lane-indexed loop grows guards)
  const int a = lane % NQ, b = lane / NQ;
  const bool on_q = FULL || lane < T, on_m = FULL || (on_q && a < N && b < N);
  // XCD-aware element order (workgroups are dealt round-robin to the 8 XCDs, each with its own L2): XCD x walks the x-th contiguous
  // (Morton-local) eighth of the elements, so a neighbour's u is more often in the reader's L2
  const int v = blockIdx.x;
  const int slot = (xcd_chunk > 0 ? (v & 7) * xcd_chunk + (v >> 3) : v) * kDirectWPB + wv;   // xcd_chunk in workgroups
  if (slot >= n_elem) return;
  // an element list leaves out elements whose rows the caller forms another way (condensed Schwarz copies); they are still READ as neighbours
  const int e = elem_list ? __builtin_amdgcn_readfirstlane(elem_list[slot]) : slot;
  // (ns_stride < 0: a mixed-degree or locally refined plan, whose clean elements this kernel serves through a list -- the element's
  // nodal offset then sits in the otherwise unused fourth word of its first side descriptor: one more scalar load)
  const int ns = __builtin_amdgcn_readfirstlane(ns_stride >= 0 ? ns0 + e * ns_stride : direct_kargs()->sides[6 * (size_t)e].pad);

  // ---- the element's u -> LDS (odd padded line length: conflict-free line reads in all three directions); all loads first
  {
    constexpr int UT = (N3 + 63) / 64;
    double uo[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) uo[t] = (N3 % 64 == 0 || lane + 64 * t < N3) ? u[ns + lane + 64 * t] : 0.0;
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const int idx = lane + 64 * t;
      const int i = idx % N, j = (idx / N) % N, k = idx / N2;
      if (N3 % 64 == 0 || idx < N3) s_U[i + PN * (j + N * k)] = uo[t];
    }
  }
  wave_lds_fence();
  // ---- own nodal face fields: trace and normal derivative at face node (a, b) of the six faces
  double own_tr[6], own_nd[6];
  {
    sdouble_ptr r0 = launder(dr0), r1 = launder(dr0 + N);   // the two rows of D once for the three directions
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      double x[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int idx = (d == 0) ? i + PN * (a + N * b) : (d == 1 ? a + PN * (i + N * b) : a + PN * (b + N * i));
        x[i] = on_m ? lds_ld(&s_U[idx]) : 0.0;
      }
      own_tr[2 * d] = x[0];
      own_tr[2 * d + 1] = x[N - 1];
      double s0 = r0[0] * x[0], s1 = r1[0] * x[0];
#pragma unroll
      for (int i = 1; i < N; ++i) {
        s0 = fma(r0[i], x[i], s0);
        s1 = fma(r1[i], x[i], s1);
      }
      own_nd[2 * d] = s0;
      own_nd[2 * d + 1] = s1;
    }
  }
  wave_lds_fence();   // s_U is free from here on: it becomes the accumulator of the lifted face terms

  double facc[N];   // VOL: the element's face terms at (i = a, j = b, k = 0 .. N-1)
#pragma unroll
  for (int i = 0; i < N; ++i) facc[i] = 0.0;
  int any3 = 0;     // hanging-aware form: the element has a side the record kernels serve -- its A u is not final here (wave-uniform)

  auto dir_body = [&](auto dc) {
    constexpr int d = decltype(dc)::value;
    constexpr int t0 = (d == 0) ? 1 : 0, t1d = (d == 2) ? 1 : 2;   // reference directions of the face indices a and b
    // (late argument loads: see DirectKernargs; the descriptor array is read-only for the kernel's lifetime: constant address
    // space, so its fields stay scalar loads)
    typedef const DirectSide __attribute__((address_space(4))) * sside_ptr;
    const sside_ptr sd = (sside_ptr)(unsigned long long)(direct_kargs()->sides + 6 * (size_t)e);
    const int kcf[2] = {sd[2 * d].kcf, sd[2 * d + 1].kcf};
    const int sgeom[2] = {sd[2 * d].geom, sd[2 * d + 1].geom};
    if constexpr ((VOL & 16) != 0 && FUSE) any3 |= ((kcf[0] & 3) == 3) | ((kcf[1] & 3) == 3);
    // ---- nodal fields of the two faces: c = 0..3 trace (own 2d, own 2d+1, nbr 2d, nbr 2d+1), c = 4..7 normal derivative
    double fld[8] = {own_tr[2 * d], own_tr[2 * d + 1], 0.0, 0.0, own_nd[2 * d], own_nd[2 * d + 1], 0.0, 0.0};
    // the normal lines of the two (+) elements' faces at THEIR face node (a, b): both faces' lines (and, below, both faces' geometric
    // factors) are requested before the first of them is used -- one memory round trip per direction instead of one per face and array
    double yy[2][N];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < N; ++i) yy[h][i] = 0.0;
      if ((kcf[h] & 3) == 1 && on_m) {
        const double* __restrict__ up = u + sd[2 * d + h].nbr_ns;
        const int dp = kcf[h] >> 6;
        const int st = (dp == 0) ? 1 : (dp == 1 ? N : N2);
        const int o0 = (dp == 0) ? N * a + N2 * b : (dp == 1 ? a + N2 * b : a + N * b);
#pragma unroll
        for (int i = 0; i < N; ++i) yy[h][i] = up[o0 + st * i];
      }
    }
#if D4EST_HIP_DIRECT_GEOM_EARLY == 2
    double gqa[2][7];
#pragma unroll
    for (int h = 0; h < 2; ++h) direct_load_geom<T, (VOL & 8) != 0 && (D4EST_HIP_DIRECT_NT_MASK & 1) != 0, (VOL & 16) != 0>(gqa[h], kcf[h] & 3, on_q, lane, sgeom[h], direct_kargs()->geom, direct_kargs()->robin_c);
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if ((kcf[h] & 3) == 1) {
        const int hi = (kcf[h] >> 5) & 1;
        fld[2 + h] = hi ? yy[h][N - 1] : yy[h][0];
        fld[6 + h] = row_dot<N>(dr0 + hi * N, yy[h]);
      }
    }
    // ---- pass 1: row (c, b) per lane, contract the face index a
    wave_lds_fence();
    if (on_m) {
#pragma unroll
      for (int c = 0; c < 8; ++c) s_S[(c * N + b) * RS + a] = fld[c];
    }
    wave_lds_fence();
    const bool row_on = FULL || lane < 8 * N;
    const int rc = lane / N, rb = lane % N;
    double P[NQ], R[NQ];
    {
      double x[N];
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = row_on ? lds_ld(&s_S[lane * RS + i]) : 0.0;
      prod_pair<N, NQ, EO>(tC, tCD, x, P, R);   // R = CD x is used for the trace rows (c < 4) only
    }
    // ---- pass 2: column (c, a') per lane, contract the face index b
    wave_lds_fence();
    if (row_on) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) s_S[rc * GS + q * RS + rb] = P[q];
    }
    wave_lds_fence();
    const bool col_on = FULL || lane < 8 * NQ;
    const int cc = lane / NQ, aq = lane % NQ;
    double o1[NQ], o2[NQ];
    {
      double col[N];
#pragma unroll
      for (int i = 0; i < N; ++i) col[i] = col_on ? lds_ld(&s_S[cc * GS + aq * RS + i]) : 0.0;
      wave_lds_fence();
      if (row_on && rc < 4) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_S[rc * GS + q * RS + rb] = R[q];
      }
      wave_lds_fence();
      prod_pair<N, NQ, EO>(tC, tCD, col, o1, o2);   // lanes c < 4: u, du/dt_b        lanes c >= 4: du/dn (o2 replaced below)
      if (cc >= 4) {
        double col2[N];
#pragma unroll
        for (int i = 0; i < N; ++i) col2[i] = col_on ? lds_ld(&s_S[(cc - 4) * GS + aq * RS + i]) : 0.0;
        prod<N, NQ, EO, false>(tC, col2, o2);       //                                 lanes c >= 4: du/dt_a
      }
    }
    // ---- SIPG terms, one face at a time (the mortar values of its two sides go through the buffer)
    double At[2][4];
    const direct_kargs_ptr K = direct_kargs();
    const double* __restrict__ geom = K->geom;
    const double* __restrict__ robin_c = K->robin_c;
#if D4EST_HIP_DIRECT_GEOM_EARLY == 1
    double gqa[2][7];
    const bool geom_nt = (VOL != 0) && (K->vol.stream & 2) != 0;   // (wave-uniform, see D4EST_HIP_DIRECT_GEOM_NT_RT)
#pragma unroll
    for (int h = 0; h < 2; ++h) direct_load_geom<T, (VOL & 8) != 0 && (D4EST_HIP_DIRECT_NT_MASK & 1) != 0, (VOL & 16) != 0>(gqa[h], kcf[h] & 3, on_q, lane, sgeom[h], geom, robin_c, geom_nt);
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      wave_lds_fence();
      if (col_on && (cc & 1) == h) {
        const int mp = (cc >> 1) & 1;
        const int dn = mp ? (kcf[h] >> 6) : d;   // the reference frame of the side that owns the trace
        const int ta = (dn == 0) ? 1 : 0, tb = (dn == 2) ? 1 : 2;
        const int c1 = (cc >= 4) ? 1 + dn : 0, c2 = (cc >= 4) ? 1 + ta : 1 + tb;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          s_S[(mp * 4 + c1) * QS + aq + NQ * q] = o1[q];
          s_S[(mp * 4 + c2) * QS + aq + NQ * q] = o2[q];
        }
      }
      wave_lds_fence();
      double qm[4] = {0, 0, 0, 0}, qp[4] = {0, 0, 0, 0}, gq[7] = {0, 0, 0, 0, 0, 0, 0};
      const int kind = kcf[h] & 3, code = (kcf[h] >> 2) & 7;
      if (on_q) {
        const int k = lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) qm[c] = lds_ld(&s_S[c * QS + k]);
        if (kind == 1) {
          const int kp = reorder_index(code, NQ - 1, a, b);
#pragma unroll
          for (int c = 0; c < 4; ++c) qp[c] = lds_ld(&s_S[(4 + c) * QS + kp]);
        } else if (kind == 2) {
          const double* __restrict__ p = K->ghost_qtrace + K->ghost_off[6 * (size_t)e + 2 * d + h] + reorder_index(code, NQ - 1, a, b);
#pragma unroll
          for (int c = 0; c < 4; ++c) qp[c] = p[c * T];
        } else if ((VOL & 16) != 0 && kind == 3) {
        } else if (robin_c) {
          qp[0] = K->robin_r[sgeom[h] + k];
        } else {
          qp[0] = K->bndry_q[sgeom[h] + k];
        }
#if D4EST_HIP_DIRECT_GEOM_EARLY == 0
        direct_load_geom<T, (VOL & 8) != 0 && (D4EST_HIP_DIRECT_NT_MASK & 1) != 0, (VOL & 16) != 0>(gq, kind, true, k, sgeom[h], geom, robin_c);
#endif
        if constexpr ((VOL & 16) != 0) {
          const int xoff = sd[2 * d + h].nbr_ns;
          if (kind == 2 && xoff >= 0) {   // export the side's own mortar-node block (wave-uniform branch)
            double* __restrict__ tp = const_cast<double*>((const double*)K->ghost_qtrace) + xoff;
#pragma unroll
            for (int c = 0; c < 4; ++c) tp[c * T + k] = qm[c];
          }
        }
      }
#if D4EST_HIP_DIRECT_GEOM_EARLY != 0
#pragma unroll
      for (int c = 0; c < 7; ++c) gq[c] = gqa[h][c];
#endif
      // interface: t1 = -1/2 sj n.(grad u_m + grad u_p), t2_l = -1/2 am_l [u]; boundary: t1 = -sj n.grad u_m, t2_l = -am_l (u - g)
      // (d4est_laplacian_flux_sipg.c:494-942, :15-336); Robin (:339-489): sj (coeff u_m - rhs) only
      double t1 = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) t1 += gq[i] * qm[1 + i] + gq[3 + i] * qp[1 + i];   // gq[3..5] = 0 on boundary sides
      const double jump = qm[0] - qp[0];
      const double w1 = (kind != 0) ? -0.5 : -1.0;
      At[h][0] = (kind == 0 && robin_c) ? gq[6] * qm[0] - qp[0] : w1 * t1 + gq[6] * jump;
#pragma unroll
      for (int l = 0; l < 3; ++l) At[h][1 + l] = w1 * gq[l] * jump;
    }
    // ---- lift pass 1: row (c8 = 4 h + field, b') per lane, contract a'
    wave_lds_fence();
    if (on_q) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) s_S[((4 * h + c) * NQ + b) * RQ + a] = At[h][c];
    }
    wave_lds_fence();
    double Yv[N];
    {
      double row[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) row[q] = col_on ? lds_ld(&s_S[lane * RQ + q]) : 0.0;
      // the term-2 field along a takes D_a^T on the way:  val = E_b E_a A0 + E_b (D^T E)_a A_t0 + (D^T E)_b E_a A_t1
      if ((cc & 3) == 1 + t0) prod<NQ, N, EO, true>(tDtE, row, Yv);
      else prod<NQ, N, EO, false>(tE, row, Yv);
    }
    wave_lds_fence();
    if (col_on) {
#pragma unroll
      for (int i = 0; i < N; ++i) s_S[cc * YS + i * RQ + aq] = Yv[i];
    }
    wave_lds_fence();
    // ---- lift pass 2: column (h, g, a) per lane: g = 0 face-local part through E, 1 term 2 along b through D^T E, 2 normal term 2
    const bool v_on = lane < 6 * N;
    const int vh = lane / (3 * N), vg = (lane / N) % 3, va = lane % N;
    double Vv[N];
    {
      double colq[NQ];
      const int f1 = (vg == 0) ? 0 : (vg == 1 ? 1 + t1d : 1 + d);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        double t = v_on ? lds_ld(&s_S[(4 * vh + f1) * YS + va * RQ + q]) : 0.0;
        if (v_on && vg == 0) t += lds_ld(&s_S[(4 * vh + 1 + t0) * YS + va * RQ + q]);
        colq[q] = t;
      }
      if (vg == 1) prod<NQ, N, EO, true>(tDtE, colq, Vv);
      else prod<NQ, N, EO, false>(tE, colq, Vv);
    }
    wave_lds_fence();
    if (v_on) {
#pragma unroll
      for (int i = 0; i < N; ++i) s_S[(3 * vh + vg) * VS + i * N + va] = Vv[i];
    }
    wave_lds_fence();
    // ---- the element's normal line at face node (a, b): face-local part at its two ends, D^T of the normal term 2 along it
    double acc[N];
    if (on_m) {
      const double val0 = lds_ld(&s_S[0 * VS + b * N + a]) + lds_ld(&s_S[1 * VS + b * N + a]), n0 = lds_ld(&s_S[2 * VS + b * N + a]);
      const double val1 = lds_ld(&s_S[3 * VS + b * N + a]) + lds_ld(&s_S[4 * VS + b * N + a]), n1 = lds_ld(&s_S[5 * VS + b * N + a]);
      sdouble_ptr r0 = launder(dr0), r1 = launder(dr0 + N);
#pragma unroll
      for (int i = 0; i < N; ++i) acc[i] = fma(r0[i], n0, r1[i] * n1);
      acc[0] += val0;
      acc[N - 1] += val1;
      if constexpr (d == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) s_U[i + PN * (a + N * b)] = acc[i];
      } else if constexpr (d == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) s_U[a + PN * (i + N * b)] += acc[i];
      } else if constexpr (VOL != 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) facc[i] = lds_ld(&s_U[a + PN * (b + N * i)]) + acc[i];
      } else {
        DirectFuse cfl;
        if constexpr (FUSE) cfl = direct_load_fuse(direct_kargs());
        double* __restrict__ Au_ = direct_kargs()->Au;
        // every load of the epilogue first (A u, and the smoother's rhs, p, u of the N nodes of this lane): one memory round trip instead
        // of two or three per node -- the stores to p / the new iterate may alias the loads as far as the compiler knows, so a loop that
        // loads inside ran 3 N dependent round trips (config 2: +14 us per Chebyshev iteration)
        double av[N], rh[FUSE ? N : 1], pp[FUSE ? N : 1], uu[FUSE ? N : 1];
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const size_t o = (size_t)ns + a + N * b + N2 * i;
          av[i] = Au_[o];
          if constexpr (FUSE) { rh[i] = cfl.rhs[o]; pp[i] = cfl.p[o]; uu[i] = u[o]; }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const size_t o = (size_t)ns + a + N * b + N2 * i;
          const double au = av[i] + (lds_ld(&s_U[a + PN * (b + N * i)]) + acc[i]);
          Au_[o] = au;
          if constexpr (FUSE) {
            // the Chebyshev update of the node (cheby_update_kernel, same roundings): u is an INPUT of this kernel (the
            // neighbours read it), so the new iterate goes to a second vector
            const double res = __dadd_rn(rh[i], __dmul_rn(-1.0, au));
            const double ri = __dmul_rn(cfl.alpha, res);
            const double pi = __dadd_rn(__dmul_rn(cfl.beta, pp[i]), ri);
            if (cfl.r) cfl.r[o] = ri;
            cfl.p[o] = pi;
            cfl.u_out[o] = __dadd_rn(uu[i], pi);
          }
        }
      }
    }
    wave_lds_fence();
  };
  dir_body(std::integral_constant<int, 0>{});
  dir_body(std::integral_constant<int, 1>{});
  dir_body(std::integral_constant<int, 2>{});
  if constexpr (VOL != 0) {
    static_assert(VOL == 0 || N == NQ, "fused volume term: N = NQ");
    // ---- the volume term: u_e -> R0, the even-odd sum-factorised apply (stiffness_wave_eo_kernel's body), A u = volume + faces
    {
      constexpr int UT = (N3 + 63) / 64;
      double uo[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) uo[t] = (N3 % 64 == 0 || lane + 64 * t < N3) ? u[ns + lane + 64 * t] : 0.0;
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const int idx = lane + 64 * t;
        const int i = idx % N, j = (idx / N) % N, k = idx / N2;
        if (N3 % 64 == 0 || idx < N3) s_U[i + PN * (j + N * k)] = uo[t];
      }
    }
    wave_lds_fence();
    {
      const DirectVol vl = direct_load_vol(direct_kargs());
      const int qs = __builtin_amdgcn_readfirstlane(vl.qs_stride >= 0 ? vl.qs0 + e * vl.qs_stride : vl.qs_list[e]);
      stiffness_wave_eo_element<N, NQ, (VOL & 3) == 2, false, (VOL & 4) != 0, (VOL & 8) != 0 && (D4EST_HIP_DIRECT_NT_MASK & 2) != 0>(s_U, s_S, vl.metric, qs, e, on_q, a, b, vl.EBf, vl.EGf, vl.EBb,
                                                                              vl.EGb, vl.affine, vl.wq, vl.cq);
    }
    if (on_m) {
      DirectFuse cfl;
      if constexpr (FUSE) cfl = direct_load_fuse(direct_kargs());
      double* __restrict__ Au_ = direct_kargs()->Au;
      // hanging-aware form: an element with a side of kind 3 gets the rest of its A u -- and then its update -- from the record flux kernel
      const bool upd = !((VOL & 16) != 0 && any3 != 0);
      double rh[FUSE ? N : 1], pp[FUSE ? N : 1], uu[FUSE ? N : 1];   // the smoother's loads first, all of them: see the faces-only form above
      if constexpr (FUSE) {
        if (upd) {
#pragma unroll
          for (int i = 0; i < N; ++i) {
            const size_t o = (size_t)ns + a + N * b + N2 * i;
            rh[i] = cfl.rhs[o]; pp[i] = cfl.p[o]; uu[i] = u[o];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const size_t o = (size_t)ns + a + N * b + N2 * i;
        const double au = s_U[a + PN * (b + N * i)] + facc[i];
        if constexpr (!FUSE && (VOL & 8) != 0 && (D4EST_HIP_DIRECT_NT_MASK & 4) != 0) __builtin_nontemporal_store(au, &Au_[o]);   // stream mode
        else if (!FUSE || !cfl.skip_Au_store || !upd) Au_[o] = au;
        if (FUSE && upd) {   // the Chebyshev update of the node, as in the faces-only form
          const double res = __dadd_rn(rh[i], __dmul_rn(-1.0, au));
          const double ri = __dmul_rn(cfl.alpha, res);
          const double pi = __dadd_rn(__dmul_rn(cfl.beta, pp[i]), ri);
          if (cfl.r) cfl.r[o] = ri;
          cfl.p[o] = pi;
          cfl.u_out[o] = __dadd_rn(uu[i], pi);
        }
      }
    }
  }
#undef tC
#undef tCD
#undef tE
#undef tDtE
#undef dr0
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static DirectHost* host_of(d4est_hip_plan* plan) { return static_cast<DirectHost*>(plan->direct); }

void direct_destroy(d4est_hip_plan* plan) {
  DirectHost* dh = host_of(plan);
  if (!dh) return;
  (void)hipFree(dh->d_sides);
  (void)hipFree(dh->d_ghost_off);
  (void)hipFree(dh->d_ops);
  (void)hipFree(dh->d_u2);
  (void)hipFree(dh->d_bnd_list); (void)hipFree(dh->d_int_list);
  delete dh;
  plan->direct = nullptr;
}

#define D4EST_HIP_DIRECT_PAIRS(X) X(2, 2) X(3, 3) X(4, 4) X(5, 5) X(6, 6) X(7, 7) X(8, 8) X(2, 3) X(3, 4) X(4, 5) X(3, 6) X(4, 6)

static bool direct_pair_built(int N, int NQ) {
#define X(N_, NQ_) if (N == N_ && NQ == NQ_) return true;
  D4EST_HIP_DIRECT_PAIRS(X)
#undef X
  return direct_mw_built(N, NQ);   // deg = deg_quad = 8 ... 15: the multi-wave kernel (d4est_hip_direct_mw.hip)
}

void direct_setup(d4est_hip_plan* plan, int N, int NQ, int ns0, int ns_stride, const double* Cm, const double* CDm, const double* Em) {
  direct_destroy(plan);
  if (!direct_pair_built(N, NQ)) return;
  const int ne = plan->n_elements;
  DirectHost* dh = new DirectHost;
  dh->N = N; dh->NQ = NQ; dh->ns0 = ns0; dh->ns_stride = ns_stride;
  dh->mw = direct_mw_built(N, NQ);
  dh->eo = true;   // the even-odd products take sizes of either parity
  std::vector<double> Cv(Cm, Cm + (size_t)NQ * N), CDv(CDm, CDm + (size_t)NQ * N), Ev(Em, Em + (size_t)N * NQ);
  std::vector<double> D = Tables1D::dij(N - 1);
  std::vector<double> DtE = Tables1D::matmul(Tables1D::transpose(D, N, N), Ev, N, N, NQ);   // (N x NQ)
  std::vector<double> ops;
  auto put = [&](const std::vector<double>& M, int R, int Cc, bool anti) {
    std::vector<double> t = dh->eo ? Tables1D::eo_table(M, R, Cc, anti) : Tables1D::transpose(M, R, Cc);
    t.resize((size_t)R * Cc, 0.0);
    ops.insert(ops.end(), t.begin(), t.end());
  };
  put(Cv, NQ, N, false);
  put(CDv, NQ, N, true);
  put(Ev, N, NQ, false);
  put(DtE, N, NQ, true);
  for (int i = 0; i < N; ++i) ops.push_back(D[i]);
  for (int i = 0; i < N; ++i) ops.push_back(D[(size_t)(N - 1) * N + i]);
  ops.resize(ops.size() + 16, 0.0);   // slack for whole-row scalar loads
  HIP_CHECK(hipMalloc(&dh->d_ops, ops.size() * sizeof(double)));
  HIP_CHECK(hipMemcpy(dh->d_ops, ops.data(), ops.size() * sizeof(double), hipMemcpyHostToDevice));
  std::vector<DirectSide> sd(6 * (size_t)ne);
  std::vector<DirectGhostOff> goff(6 * (size_t)ne, 0);
  for (int e = 0; e < ne; ++e)
    for (int f = 0; f < 6; ++f) {
      const size_t s = 6 * (size_t)e + f;
      DirectSide d{};
      const int nbr = plan->side_nbr[s];
      const int kind = (nbr == -1) ? 0 : (nbr >= 0 ? 1 : 2);
      const int fp = (kind == 0) ? 0 : plan->side_nbr_face[s];
      d.kcf = kind | ((plan->side_reorder[s] & 7) << 2) | (fp << 5);
      d.nbr_ns = (kind == 1) ? plan->nodal_stride[nbr] : 0;
      d.geom = plan->side_mortar_stride[s];
      goff[s] = (kind == 2) ? plan->ghost_trace_offset[s] : 0;
      sd[s] = d;
    }
  HIP_CHECK(hipMalloc(&dh->d_sides, std::max<size_t>(sd.size(), 1) * sizeof(DirectSide)));
  if (!sd.empty()) HIP_CHECK(hipMemcpy(dh->d_sides, sd.data(), sd.size() * sizeof(DirectSide), hipMemcpyHostToDevice));
  HIP_CHECK(hipMalloc(&dh->d_ghost_off, std::max<size_t>(goff.size(), 1) * sizeof(DirectGhostOff)));
  if (!goff.empty()) HIP_CHECK(hipMemcpy(dh->d_ghost_off, goff.data(), goff.size() * sizeof(DirectGhostOff), hipMemcpyHostToDevice));
  // elements whose rows need ghost data / need none (a rank without ghosts: everything is interior, the lists stay empty)
  std::vector<int> bnd, inn;
  for (int e = 0; e < ne; ++e) {
    bool g = false;
    for (int f = 0; f < 6; ++f) g = g || plan->side_nbr[6 * (size_t)e + f] <= -2;
    (g ? bnd : inn).push_back(e);
  }
  if (!bnd.empty()) {
    dh->n_bnd = (int)bnd.size(); dh->n_int = (int)inn.size();
    HIP_CHECK(hipMalloc(&dh->d_bnd_list, bnd.size() * sizeof(int)));
    HIP_CHECK(hipMemcpy(dh->d_bnd_list, bnd.data(), bnd.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&dh->d_int_list, std::max<size_t>(inn.size(), 1) * sizeof(int)));
    if (!inn.empty()) HIP_CHECK(hipMemcpy(dh->d_int_list, inn.data(), inn.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  plan->direct = dh;
}

void direct_ghost_split(d4est_hip_plan* plan, const int** bnd_list, int* n_bnd, const int** int_list, int* n_int) {
  DirectHost* dh = host_of(plan);
  if (!dh) D4EST_HIP_ABORT("direct_ghost_split: the plan has no direct face kernel");
  *bnd_list = dh->d_bnd_list; *n_bnd = dh->n_bnd; *int_list = dh->d_int_list; *n_int = dh->n_int;
}

// Below this many elements the chip is far from full and the two-phase kernels win: they put 3 + 6 wavefronts on an element, the
// direct kernel ONE, whose serial chain then IS the run time (measured, p = 7: 64 elements 16 us two-phase against 23 us in one kernel,
// 512 elements 25 against 25 -- Chebyshev 30 against 34 --, 4096 elements 83 against 52).  Auto only; tuning values 1 / 2 force it.
constexpr int kDirectMinElements = 768;

bool direct_active(const d4est_hip_plan* plan) {
  const int t = plan->tuning[D4EST_HIP_TUNE_FACE_DIRECT];
  const DirectHost* dh = static_cast<const DirectHost*>(plan->direct);
  // (the multi-wave kernel of p >= 8 puts a whole workgroup on an element and wins at every size measured: 64 elements at p = 11
  // 29.8 us against 37.5 us two-phase, 512 elements 42 against 59, 4096 elements 220 against 315)
  return dh != nullptr && t != 0 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 &&
         (t > 0 || dh->mw || plan->n_elements >= kDirectMinElements);
}

void direct_set_element_list(d4est_hip_plan* plan, const int* list_dev, int n_list) {
  DirectHost* dh = host_of(plan);
  if (!dh) D4EST_HIP_ABORT("direct_set_element_list: the plan has no direct face kernel");
  dh->d_list = list_dev;
  dh->n_list = list_dev ? n_list : 0;
}

bool direct_has_element_list(const d4est_hip_plan* plan) {
  const DirectHost* dh = static_cast<const DirectHost*>(plan->direct);
  return dh && dh->d_list != nullptr;
}

double* direct_second_vector(d4est_hip_plan* plan) {
  DirectHost* dh = host_of(plan);
  if (!dh->d_u2) HIP_CHECK(hipMalloc(&dh->d_u2, std::max<size_t>((size_t)plan->local_nodes, 1) * sizeof(double)));
  return dh->d_u2;
}

// the whole operator in one kernel?  conforming uniform plan with the direct tables, N = NQ in {6, 8}, one bucket in element order
// with affine strides, even-odd volume tables, default volume-kernel tuning
bool direct_fused_ok(const d4est_hip_plan* plan) {
  const DirectHost* dh = static_cast<const DirectHost*>(plan->direct);
  static const bool dbg = std::getenv("D4EST_HIP_DEBUG_FUSED") != nullptr;
  auto no = [&](const char* why) {
    if (dbg) std::fprintf(stderr, "[d4est_hip] whole-operator kernel not used: %s\n", why);
    return false;
  };
  if (!direct_active(plan) || !dh) return no("no direct face kernel on this plan");
  const int t = plan->tuning[D4EST_HIP_TUNE_FACE_DIRECT];
  if (!(t < 0 || t == 2)) return no("tuning key 11");
  if (dh->N != dh->NQ) return no("deg != deg_quad");
  if (!plan->has_geometry) return no("no geometry yet");
  if (plan->buckets.size() != 1) return no("more than one (deg, deg_quad) bucket");
  const Bucket& bk = plan->buckets[0];
  if (bk.N != dh->N || bk.NQ != dh->NQ || !bk.d_EBf) return no("bucket tables");
  // (the nodal offsets are affine: the direct tables exist only on uniform plans; the bucket keeps affine strides only when the
  // quadrature offsets are affine too -- not so on a Schwarz subdomain plan, which reads them from the list)
  if (bk.ns_stride >= 0 && (bk.ns0 != dh->ns0 || bk.ns_stride != dh->ns_stride)) return no("nodal offsets");
  if (bk.n_elem != plan->n_elements) return no("bucket size");
  if (dh->order_ok < 0) {   // (the bucket order is fixed at plan creation: looked at once, not at every apply)
    int ok = 1;
    for (int i = 0; i < plan->n_elements && ok; ++i) ok = (plan->elem_ids[i] == i);
    dh->order_ok = ok;
  }
  if (!dh->order_ok) return no("bucket order");
  if (dh->mw) {
    if (plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] == 0) return no("even-odd contractions switched off");
    return true;
  }
  const int tw = plan->tuning[D4EST_HIP_TUNE_STIFFNESS_WAVE];
  if (!(tw < 0 || tw == 11)) return no("tuning key 1");   // the volume kernel whose body rides along must be the selected one
  return true;
}

// vol_term: 0 the face terms only (Au += ...), 1 the whole operator (Au = volume + faces; direct_fused_ok), 2 the whole operator
// with the plan's zeroth-order term (plan_set_lhs_coefficient) in its volume stage.
// dh / bk: the direct tables and the bucket they belong to -- the plan's own (uniform conforming plans) or one clean-element bucket of
// the hybrid operator (hybrid = true: offsets by element id from the side table / d_qs_by_elem, streamed metric).
static void launch_direct_core(d4est_hip_plan* plan, DirectHost* dh, const Bucket& bk, bool hybrid, const int* d_qs_by_elem, const double* u,
                               const double* ghost_trace, double* Au, const DirectFuse* cf, const double* robin_c, const double* robin_r,
                               int vol_term) {
  if (!plan->has_face_geometry) D4EST_HIP_ABORT("apply flux: plan_set_mortar_geometry was not called");
  const int n = dh->d_list ? dh->n_list : plan->n_elements;
  if (n == 0) return;
  if (plan->ghost_trace_doubles > 0 && !ghost_trace) D4EST_HIP_ABORT("apply flux: plan has ghost sides but no ghost trace buffer was given");
  static const bool no_remap = std::getenv("D4EST_HIP_NO_XCD_REMAP") != nullptr;
  const int wpb = dh->mw ? 1 : kDirectWPB;
  const int n_wg = (n + wpb - 1) / wpb;
  const int chunk = (n % (8 * wpb) == 0 && !no_remap) ? n_wg / 8 : 0;
  DirectVol vol;
  int vmode = 0;
  if (vol_term) {
    if (!hybrid && !direct_fused_ok(plan)) D4EST_HIP_ABORT("direct face kernel: the fused volume term was requested on a plan that cannot take it");
    vol.metric = plan->d_metric;
    vol.EBf = bk.d_EBf; vol.EGf = bk.d_EGf; vol.EBb = bk.d_EBb; vol.EGb = bk.d_EGb;
    vol.EDq = bk.d_EDq; vol.EDqT = bk.d_EDqT;
    vol.stream = plan->stream_mode;
    {
      // one-wavefront kernel outside stream mode: the mortar factors -- read once per apply, 42 B/DoF at p = 7 -- with the non-temporal hint
      // (bit 1), so that the neighbours' u stays in the XCD's L2.  Measured (alternating runs): config 2 (222 MB per apply, fits the Infinity
      // Cache) HBM traffic 275 -> 243 MB = 1.24 -> 1.09 x algorithmic but 51.6 -> 53.2 us; the hybrid operator on the locally refined
      // p = 7 brick (306 MB with its trace array) 82.2 -> 79.4 us.  So: on for the hybrid operator's launches, off on uniform plans
      // (D4EST_HIP_GEOM_NT = 0 never, 2 always; the hint on the metric or on A u alone changes neither traffic nor time)
      static const int geom_nt = [] { const char* e = std::getenv("D4EST_HIP_GEOM_NT"); return e ? std::atoi(e) : 1; }();
      if (!dh->mw && !vol.stream && (geom_nt == 2 || (geom_nt == 1 && hybrid))) vol.stream |= 2;
    }
    // (measured, round 4: forcing the non-temporal hints on at config 2 -- 222 MB per apply, below the 320 MB threshold -- takes the HBM
    // traffic of this kernel from 275 to 242 MB = 1.24 -> 1.09 x algorithmic, the neighbours' u stays in the L2, but the kernel does not
    // get faster: 50.4 - 50.7 us without, 51.9 - 52.4 us with the hints in alternating runs; the threshold stays.
    // profiles/r04_aij_l4p7_stream_traffic.txt)
    if (hybrid && dh->hy_qs_stride > 0) { vol.qs0 = dh->hy_qs0; vol.qs_stride = dh->hy_qs_stride; vol.qs_list = nullptr; }
    else if (hybrid) { vol.qs0 = 0; vol.qs_stride = -1; vol.qs_list = d_qs_by_elem; }
    else { vol.qs0 = bk.qs0; vol.qs_stride = bk.ns_stride >= 0 ? bk.qs_stride : -1; vol.qs_list = plan->d_qs_list + bk.elem_offset; }
    const bool aff = !hybrid && bk.affine && plan->tuning[D4EST_HIP_TUNE_AFFINE] != 0 && plan->d_metric_affine;
    if (aff) { vol.affine = plan->d_metric_affine; vol.wq = bk.d_w; }
    vmode = aff ? 2 : 1;
    if (vol_term == 2) {
      if (!plan->d_lhs_coeff) D4EST_HIP_ABORT("direct face kernel: the zeroth-order term was requested but no coefficient is set");
      vol.cq = ensure_lhs_wjc(plan);
      vmode |= 4;
    }
    const char* sm = (!aff && vol_term != 2 && (vol.stream & 1)) ? ",stream" : "";
    if (dh->mw) std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::operator_mw_kernel<%d,vol%s%s> (stiffness_wave_kernel body + faces)", dh->N, aff ? ",affine" : "", sm);
    else std::snprintf(plan->last_kernel, sizeof(plan->last_kernel), "d4est_hip::faces_direct_kernel<%d,%d,vol%s%s> (faces + stiffness_wave_eo body)", dh->N, dh->NQ, aff ? ",affine" : "", sm);
  }
  if (dh->mw) {
    launch_direct_mw(plan, dh, u, ghost_trace, Au, cf, robin_c, robin_r, vmode, vol, n, chunk);
    return;
  }
  const DirectFuse cfv = cf ? *cf : DirectFuse{};
  bool done = false;
  if (vmode == 1 && (vol.stream & 1)) vmode = 9;   // stream mode (plan->stream_mode): the twin with non-temporal metric / factor loads and A u stores
  if (dh->hang) {
    if (!(vmode == 1 || vmode == 9)) D4EST_HIP_ABORT("direct face kernel: the hanging-aware form exists for the plain whole operator only (vmode %d)", vmode);
    vmode |= 16;
  }
#define D4EST_HIP_DIRECT_GO(N_, NQ_, FUSE_, VOL_)                                                                             \
  hipLaunchKernelGGL((faces_direct_kernel<N_, NQ_, true, FUSE_, VOL_>), dim3(n_wg), dim3(64 * kDirectWPB), 0,     \
                     plan->stream, u, ghost_trace, Au, dh->d_sides, dh->d_ghost_off, dh->d_ops, plan->d_face_geom, plan->d_bndry,            \
                     robin_c, robin_r, n, dh->ns0, dh->ns_stride, chunk, cfv, vol, dh->d_list)
#define X(N_, NQ_)                                                                 \
  if (!done && dh->N == N_ && dh->NQ == NQ_) {                                     \
    if constexpr (N_ == NQ_) {                                                     \
      if (vmode == 1) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 1); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 1); done = true; } \
      if (vmode == 9) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 9); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 9); done = true; } \
      if (vmode == 17) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 17); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 17); done = true; } \
      if (vmode == 25) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 25); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 25); done = true; } \
      if (vmode == 2) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 2); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 2); done = true; } \
      if (vmode == 5) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 5); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 5); done = true; } \
      if (vmode == 6) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 6); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 6); done = true; } \
    }                                                                              \
    if (!done) { if (cf) D4EST_HIP_DIRECT_GO(N_, NQ_, true, 0); else D4EST_HIP_DIRECT_GO(N_, NQ_, false, 0); done = true; }        \
  }
  D4EST_HIP_DIRECT_PAIRS(X)
#undef X
#undef D4EST_HIP_DIRECT_GO
  if (!done) D4EST_HIP_ABORT("direct face kernel: no instance for N = %d, NQ = %d", dh->N, dh->NQ);
  static bool occ_done = false;
  if (!occ_done && std::getenv("D4EST_HIP_DEBUG_OCC")) {
    occ_done = true;
    int nb = -1, nv = -1;
    hipFuncAttributes at{}, av{};
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(faces_direct_kernel<8, 8, true, false, 0>), 64 * kDirectWPB, 0);
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(faces_direct_kernel<8, 8, true, false, 0>));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nv, reinterpret_cast<const void*>(faces_direct_kernel<8, 8, true, false, 1>), 64 * kDirectWPB, 0);
    (void)hipFuncGetAttributes(&av, reinterpret_cast<const void*>(faces_direct_kernel<8, 8, true, false, 1>));
    std::fprintf(stderr, "[d4est_hip] occupancy: faces_direct<8,8> %d workgroups of %d waves per CU (regs %d, lds %zu, scratch %zu); with the volume term %d (regs %d, lds %zu, scratch %zu)\n",
                 nb, kDirectWPB, at.numRegs, at.sharedSizeBytes, at.localSizeBytes, nv, av.numRegs, av.sharedSizeBytes, av.localSizeBytes);
  }
  HIP_CHECK(hipGetLastError());
}

void launch_direct_faces(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf,
                         const double* robin_c, const double* robin_r, int vol_term) {
  DirectHost* dh = host_of(plan);
  if (!dh) D4EST_HIP_ABORT("direct face kernel: the plan has no direct tables");
  launch_direct_core(plan, dh, plan->buckets[0], false, nullptr, u, ghost_trace, Au, cf, robin_c, robin_r, vol_term);
}

// ---------------------------------------------------------------------------
// The hybrid operator: mixed-degree and locally refined plans (BASELINE config 4's mesh class).
// Most elements of an hp-adaptive mesh sit inside a region of one degree and one refinement level: all six of their sides are
// conforming, against a local element of the same degree (or the domain boundary).  Those CLEAN elements get the whole operator from
// the trace-free one-kernel path -- faces_direct_kernel / operator_mw_kernel of their degree bucket over an element list, u in, A u out,
// no trace arrays -- and only the DIRTY rest (a mixed-degree, hanging or ghost side) runs traces + volume + flux through the two-phase
// kernels, on lists: traces of the dirty elements and of their neighbours (the ring), the volume term and the flux of the dirty ones.
// Every element's A u is written by exactly one path.  Reference: d4est_laplacian_apply_aij (src/dGMath/d4est_laplacian.c:318-417) is
// one path for every mesh; this is the same operator, split by where each element's data comes from.
// ---------------------------------------------------------------------------
struct HybridHost {
  std::vector<DirectHost*> dh;        // per plan bucket (nullptr: no clean elements there)
  std::vector<int*> d_clean;          // per plan bucket: its clean elements
  std::vector<int> n_clean;
  DirectSide* d_sides = nullptr;      // shared by the buckets' tables (the element's nodal offset in sides[6 e].pad)
  DirectGhostOff* d_ghost_off = nullptr;
  int* d_qs_by_elem = nullptr;
  int *d_dirty = nullptr, *d_ring = nullptr;
  int n_dirty = 0, n_ring = 0, n_clean_total = 0;
  std::vector<int> h_dirty, h_ring;   // host copies (the face set-up splits them by kernel family, hybrid_host_lists)
  int *d_ns_dirty = nullptr, *d_qs_dirty = nullptr;   // bucket-ordered lists of the dirty elements (the volume kernels' view)
  std::vector<int> dirty_off, dirty_cnt;
  char path[96] = "";
  double* d_u2 = nullptr;             // second iterate vector of the fused Chebyshev update (hybrid_second_vector)
  bool hang = false;                  // hanging-aware form: the clean kernels read record traces and export small sides' blocks
  // the clean buckets' launches are short, latency-structured kernels (one wavefront / workgroup per element, a few hundred elements
  // each): back to back on one stream they cost their serial chains one after another (p = 3 ... 9 graded: 7 launches, 175 us), so each
  // bucket gets its own stream, forked from and joined to the plan's stream by events, beside the dirty path on the plan's stream
  std::vector<hipStream_t> side;
  std::vector<hipEvent_t> done;
  hipEvent_t fork = nullptr;
};
static HybridHost* hybrid_of(const d4est_hip_plan* plan) { return static_cast<HybridHost*>(plan->hybrid); }

void hybrid_destroy(d4est_hip_plan* plan) {
  HybridHost* hh = hybrid_of(plan);
  if (!hh) return;
  for (DirectHost* d : hh->dh)
    if (d) { (void)hipFree(d->d_ops); delete d; }
  for (int* p : hh->d_clean) (void)hipFree(p);
  for (hipStream_t st : hh->side) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : hh->done) (void)hipEventDestroy(ev);
  if (hh->fork) (void)hipEventDestroy(hh->fork);
  (void)hipFree(hh->d_sides); (void)hipFree(hh->d_ghost_off); (void)hipFree(hh->d_qs_by_elem); (void)hipFree(hh->d_dirty); (void)hipFree(hh->d_ring);
  (void)hipFree(hh->d_ns_dirty); (void)hipFree(hh->d_qs_dirty); (void)hipFree(hh->d_u2);
  delete hh;
  plan->hybrid = nullptr;
}

bool hybrid_pair_built(int N, int NQ) { return N == NQ && direct_pair_built(N, NQ); }

template <typename T>
static T* hy_upload(const std::vector<T>& v) {
  T* d = nullptr;
  HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

// clean[e]: 1 = the element takes the one-kernel path.  tables(bucket) -> {C, CD, E} of the bucket's degree (host pointers, NQ x N / N x NQ)
void hybrid_setup(d4est_hip_plan* plan, const std::vector<char>& clean, const std::vector<const double*>& C, const std::vector<const double*>& CD,
                  const std::vector<const double*>& E, const std::vector<HybridSideOverride>* ov, const char* form) {
  hybrid_destroy(plan);
  const int ne = plan->n_elements;
  HybridHost* hh = new HybridHost;
  const size_t nb = plan->buckets.size();
  hh->dh.assign(nb, nullptr); hh->d_clean.assign(nb, nullptr); hh->n_clean.assign(nb, 0);
  hh->dirty_off.assign(nb, 0); hh->dirty_cnt.assign(nb, 0);
  // side table of every element (only clean elements are worked on, everybody is read as a neighbour)
  std::vector<DirectSide> sd(6 * (size_t)ne);
  std::vector<DirectGhostOff> goff(6 * (size_t)ne, 0);
  for (int e = 0; e < ne; ++e)
    for (int f = 0; f < 6; ++f) {
      const size_t s = 6 * (size_t)e + f;
      DirectSide d{};
      const int nbr = plan->side_nbr[s];
      const int kind = (nbr == -1) ? 0 : (nbr >= 0 ? 1 : 2);
      const int fp = (kind == 0) ? 0 : plan->side_nbr_face[s];
      d.kcf = kind | ((plan->side_reorder[s] & 7) << 2) | (fp << 5);
      d.nbr_ns = (kind == 1) ? plan->nodal_stride[nbr] : 0;
      d.geom = plan->side_mortar_stride[s];
      d.pad = (f == 0) ? plan->nodal_stride[e] : 0;
      if (ov && clean[e] && (*ov)[s].kind >= 0) {
        const HybridSideOverride& o = (*ov)[s];
        d.kcf = o.kind | ((plan->side_reorder[s] & 7) << 2);
        if (o.kind == 2) { d.nbr_ns = o.export_off; d.geom = o.geom; goff[s] = o.goff; }
        hh->hang = true;
      }
      sd[s] = d;
    }
  hh->d_sides = hy_upload(sd);
  hh->d_ghost_off = hy_upload(goff);
  hh->d_qs_by_elem = hy_upload(plan->quad_stride);
  // bucket-ordered walk: clean lists per bucket, dirty lists for the volume kernels' view
  std::vector<int> dirty, ns_dirty, qs_dirty;
  for (size_t b = 0; b < nb; ++b) {
    const Bucket& bk = plan->buckets[b];
    std::vector<int> cl;
    hh->dirty_off[b] = (int)ns_dirty.size();
    for (int i = 0; i < bk.n_elem; ++i) {
      const int e = plan->elem_ids[bk.elem_offset + i];
      if (clean[e]) cl.push_back(e);
      else { ns_dirty.push_back(plan->nodal_stride[e]); qs_dirty.push_back(plan->quad_stride[e]); }
    }
    hh->dirty_cnt[b] = (int)ns_dirty.size() - hh->dirty_off[b];
    if (cl.empty()) continue;
    std::sort(cl.begin(), cl.end());   // element (Morton) order: neighbours' u close in the caches
    DirectHost* dh = new DirectHost;
    const int N = bk.N, NQ = bk.NQ;
    dh->N = N; dh->NQ = NQ; dh->ns0 = 0; dh->ns_stride = -1;
    dh->mw = direct_mw_built(N, NQ);
    dh->eo = true;
    dh->hang = hh->hang;
    std::vector<double> Cv(C[b], C[b] + (size_t)NQ * N), CDv(CD[b], CD[b] + (size_t)NQ * N), Ev(E[b], E[b] + (size_t)N * NQ);
    std::vector<double> D = Tables1D::dij(N - 1);
    std::vector<double> DtE = Tables1D::matmul(Tables1D::transpose(D, N, N), Ev, N, N, NQ);
    std::vector<double> ops;
    auto put = [&](const std::vector<double>& Mx, int R, int Cc, bool anti) {
      std::vector<double> t = Tables1D::eo_table(Mx, R, Cc, anti);
      t.resize((size_t)R * Cc, 0.0);
      ops.insert(ops.end(), t.begin(), t.end());
    };
    put(Cv, NQ, N, false);
    put(CDv, NQ, N, true);
    put(Ev, N, NQ, false);
    put(DtE, N, NQ, true);
    for (int i = 0; i < N; ++i) ops.push_back(D[i]);
    for (int i = 0; i < N; ++i) ops.push_back(D[(size_t)(N - 1) * N + i]);
    ops.resize(ops.size() + 16, 0.0);
    dh->d_ops = hy_upload(ops);
    dh->d_sides = hh->d_sides;
    dh->d_ghost_off = hh->d_ghost_off;
    hh->d_clean[b] = hy_upload(cl);
    hh->n_clean[b] = (int)cl.size();
    dh->d_list = hh->d_clean[b];
    dh->n_list = hh->n_clean[b];
    // every element of the plan, in order, at affine offsets (a locally refined mesh of one degree in the hanging-aware form): the
    // arithmetic addressing of a uniform plan -- no list, no per-element offset loads in front of the element's first memory request
    if ((int)cl.size() == ne && ne > 0) {
      bool affine = true;
      const int n3 = N * N * N, q3 = NQ * NQ * NQ;
      for (int e = 0; e < ne && affine; ++e)
        affine = plan->nodal_stride[e] == plan->nodal_stride[0] + e * n3 && plan->quad_stride[e] == plan->quad_stride[0] + e * q3;
      if (affine) {
        dh->d_list = nullptr; dh->n_list = 0;
        dh->ns0 = plan->nodal_stride[0]; dh->ns_stride = n3;
        dh->hy_qs0 = plan->quad_stride[0]; dh->hy_qs_stride = q3;
      }
    }
    hh->dh[b] = dh;
    hh->n_clean_total += (int)cl.size();
  }
  // dirty elements in element order, and the ring: the dirty elements plus every local element across one of their sides (whose traces the
  // dirty flux reads; hanging neighbours through side_nbr4)
  std::vector<char> in_ring(ne, 0);
  for (int e = 0; e < ne; ++e) {
    if (clean[e]) continue;
    dirty.push_back(e);
    in_ring[e] = 1;
    for (int f = 0; f < 6; ++f) {
      const size_t s = 6 * (size_t)e + f;
      if (plan->side_nbr[s] >= 0) in_ring[plan->side_nbr[s]] = 1;
      if (!plan->side_nbr4.empty())
        for (int k = 0; k < 4; ++k)
          if (plan->side_nbr4[4 * s + k] >= 0) in_ring[plan->side_nbr4[4 * s + k]] = 1;
    }
  }
  std::vector<int> ring;
  for (int e = 0; e < ne; ++e)
    if (in_ring[e]) ring.push_back(e);
  hh->d_dirty = hy_upload(dirty); hh->n_dirty = (int)dirty.size();
  hh->d_ring = hy_upload(ring); hh->n_ring = (int)ring.size();
  hh->h_dirty = dirty; hh->h_ring = ring;
  hh->d_ns_dirty = hy_upload(ns_dirty);
  hh->d_qs_dirty = hy_upload(qs_dirty);
  char tag[32] = "";
  if (hh->hang) std::snprintf(tag, sizeof(tag), " (%s)", form);
  std::snprintf(hh->path, sizeof(hh->path), "hybrid%s: direct+volume on %d clean elements, two-phase on %d", tag, hh->n_clean_total, hh->n_dirty);
  plan->hybrid = hh;
}

bool hybrid_active(const d4est_hip_plan* plan) {
  const HybridHost* hh = hybrid_of(plan);
  return hh != nullptr && plan->tuning[D4EST_HIP_TUNE_HYBRID] != 0 && plan->tuning[D4EST_HIP_TUNE_FLUX_FAST] != 0 &&
         plan->tuning[D4EST_HIP_TUNE_FACE_DIRECT] != 0 && plan->tuning[D4EST_HIP_TUNE_STIFFNESS_EO] != 0 && plan->has_geometry;
}
const char* hybrid_path(const d4est_hip_plan* plan) { return hybrid_of(plan)->path; }
// The Chebyshev update can ride in the hybrid operator's kernels where ONE clean launch writes every element's A u and the record flux
// kernel finishes the elements it serves (hanging-aware form, no dirty elements, one single-wave bucket, no zeroth-order term: that
// term is added after the kernels on this path)
bool hybrid_can_fuse_update(const d4est_hip_plan* plan) {
  const HybridHost* hh = hybrid_of(plan);
  if (!hh || !hybrid_active(plan)) return false;
  if (plan->d_lhs_coeff || lhs_extra_term(plan)) return false;
  d4est_hip_plan* p = const_cast<d4est_hip_plan*>(plan);
  // ... or, on a plan WITHOUT hanging faces (mixed degrees), in every clean bucket's kernel and in the flux kernels of the dirty list: each
  // element's A u is final in exactly one of them (with hanging faces a dirty element's rows are finished by the record flux kernel)
  if (!faces_hp(p)) return true;
  if (!hh->hang || hh->n_dirty != 0) return false;
  if (!faces_have_units(p)) return false;
  int n_launch = 0;
  for (const DirectHost* d : hh->dh)
    if (d) { ++n_launch; if (d->mw) return false; }
  return n_launch == 1;
}
double* hybrid_second_vector(d4est_hip_plan* plan) {
  HybridHost* hh = hybrid_of(plan);
  if (!hh->d_u2) HIP_CHECK(hipMalloc(&hh->d_u2, std::max<size_t>((size_t)plan->local_nodes, 1) * sizeof(double)));
  return hh->d_u2;
}
bool hybrid_hanging(const d4est_hip_plan* plan) { return hybrid_of(plan)->hang; }
void hybrid_host_lists(const d4est_hip_plan* plan, const std::vector<int>** dirty, const std::vector<int>** ring) {
  const HybridHost* hh = hybrid_of(plan);
  *dirty = &hh->h_dirty; *ring = &hh->h_ring;
}
void hybrid_lists(const d4est_hip_plan* plan, const int** dirty, int* n_dirty, const int** ring, int* n_ring) {
  const HybridHost* hh = hybrid_of(plan);
  *dirty = hh->d_dirty; *n_dirty = hh->n_dirty; *ring = hh->d_ring; *n_ring = hh->n_ring;
}

// the clean elements: one whole-operator launch per degree bucket (u in, A u out), every bucket on its own stream between a fork event
// on the plan's stream (phase 0, before the dirty path is queued there) and the join (phase 1, after it)
void launch_hybrid_clean(d4est_hip_plan* plan, const double* u, const double* ghost_trace, double* Au, const double* robin_c, const double* robin_r,
                         int phase, const DirectFuse* cf) {
  HybridHost* hh = hybrid_of(plan);
  if (cf && !hybrid_can_fuse_update(plan)) D4EST_HIP_ABORT("hybrid operator: a fused update was requested on a plan that cannot carry it");
  static const bool serial = std::getenv("D4EST_HIP_HYBRID_SERIAL") != nullptr;
  int n_launch = 0;
  for (DirectHost* d : hh->dh) n_launch += d != nullptr;
  const bool forked = !serial && n_launch > 1;
  if (!forked) {   // one bucket: nothing to overlap
    if (phase == 0) return;
    for (size_t b = 0; b < hh->dh.size(); ++b)
      if (hh->dh[b]) launch_direct_core(plan, hh->dh[b], plan->buckets[b], true, hh->d_qs_by_elem, u, ghost_trace, Au, cf, robin_c, robin_r, 1);
    return;
  }
  hipStream_t main = plan->stream;
  if (phase == 0) {
    if (!hh->fork) {
      HIP_CHECK(hipEventCreateWithFlags(&hh->fork, hipEventDisableTiming));
      for (int i = 0; i < n_launch; ++i) {
        hipStream_t st; hipEvent_t ev;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hh->side.push_back(st); hh->done.push_back(ev);
      }
    }
    if (plan->d_lhs_coeff == nullptr) {}   // (the zeroth-order term is added after the join, on the plan's stream)
    HIP_CHECK(hipEventRecord(hh->fork, main));
    int i = 0;
    for (size_t b = 0; b < hh->dh.size(); ++b) {
      if (!hh->dh[b]) continue;
      HIP_CHECK(hipStreamWaitEvent(hh->side[i], hh->fork, 0));
      plan->stream = hh->side[i];
      launch_direct_core(plan, hh->dh[b], plan->buckets[b], true, hh->d_qs_by_elem, u, ghost_trace, Au, cf, robin_c, robin_r, 1);
      plan->stream = main;
      HIP_CHECK(hipEventRecord(hh->done[i], hh->side[i]));
      ++i;
    }
    return;
  }
  for (int i = 0; i < n_launch; ++i) HIP_CHECK(hipStreamWaitEvent(main, hh->done[i], 0));
}
// the dirty elements' volume term
void launch_hybrid_dirty_stiffness(d4est_hip_plan* plan, const double* u, double* Au) {
  HybridHost* hh = hybrid_of(plan);
  if (hh->n_dirty == 0) return;
  launch_stiffness_view(plan, u, Au, hh->d_ns_dirty, hh->d_qs_dirty, hh->dirty_off.data(), hh->dirty_cnt.data());
}

}  // namespace d4est_hip

// Offsets of DirectKernargs' members in parameter order (tests/test_capi.py compares them with the .args offsets the compiler
// recorded for faces_direct_kernel in the code object: the late argument loads rely on the two layouts being the same).
extern "C" __attribute__((visibility("default"))) int d4est_hipi_direct_kernarg_offsets(int* out, int cap) {
  using K = d4est_hip::DirectKernargs;
  const int off[] = {(int)offsetof(K, u), (int)offsetof(K, ghost_qtrace), (int)offsetof(K, Au), (int)offsetof(K, sides),
                     (int)offsetof(K, ghost_off), (int)offsetof(K, ops), (int)offsetof(K, geom), (int)offsetof(K, bndry_q),
                     (int)offsetof(K, robin_c), (int)offsetof(K, robin_r), (int)offsetof(K, n_elem), (int)offsetof(K, ns0),
                     (int)offsetof(K, ns_stride), (int)offsetof(K, xcd_chunk), (int)offsetof(K, cf), (int)offsetof(K, vol),
                     (int)offsetof(K, elem_list)};
  const int n = (int)(sizeof(off) / sizeof(off[0]));
  for (int i = 0; i < n && i < cap; ++i) out[i] = off[i];
  return n;
}

