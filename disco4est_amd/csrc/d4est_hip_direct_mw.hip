// The whole operator A u = (stiffness + SIPG face terms) u of a conforming, uniform-degree plan with deg = deg_quad = 8 ... 15 in ONE
// kernel: u in, A u out -- no mortar-node trace arrays, no read-modify-write of A u.
//
// Replaces d4est_laplacian_apply_aij (src/dGMath/d4est_laplacian.c:318-417) = d4est_laplacian_apply_stiffness_matrix (:198-234) +
// d4est_laplacian_compute_dudr (:237-282) + d4est_laplacian_flux_interface / _boundary (src/dGMath/d4est_laplacian_flux.c:23-1014)
// with the SIPG callbacks (src/dGMath/d4est_laplacian_flux_sipg.c:15-942) for such a plan; same numbers as the two-phase kernels of
// d4est_hip_faces.hip (trace kernel -> 4 fields per mortar node -> flux kernel), which move 131 KB per p = 11 element through HBM for
// their own intermediates.
//
// The design is the one of faces_direct_kernel (d4est_hip_direct.hip, deg_quad <= 7, one wavefront per element) carried over to the
// multi-wave workgroup of the p >= 8 volume kernel (stiffness_wave_kernel, one element per workgroup):
//   * the volume term runs first (stiffness_mw_element, d4est_hip_mwave.h) and leaves (K u)_e in the LDS image R0, which then is the
//     accumulator of the lifted face terms; the second LDS field is the transposition buffer of the face passes -- two fields per
//     workgroup as in the volume kernel, so the occupancy is the volume kernel's;
//   * per reference direction d the two faces 2d, 2d+1: every thread of the face (a, b) loads the normal line of the element and of the
//     two (+) elements at its face node from u (own: just read by this workgroup, neighbours: L2 / Infinity Cache), which gives
//     8 nodal face fields (trace, normal derivative) x (own, neighbour) x 2 faces;
//   * the 1-D products of the four passes (side nodes -> mortar quadrature nodes along a, along b; integrate-and-project back along
//     a', along b') are LINE TASKS: one line per thread, the operator wave-uniform through scalar loads in the even-odd form.  Tasks of
//     a pass are ordered by operator (C before C D, E before D^T E) and the second group starts on a wavefront boundary where the
//     workgroup has the threads for it, so a wavefront issues one product per pass except where it straddles the boundary;
//   * the SIPG terms are formed at the mortar node of the thread, one face at a time; the (+) values come through the p4est re-ordering
//     code of the face pair, ghost (+) sides read the exchanged mortar-node block, Dirichlet / Robin sides their boundary arrays.
#include <algorithm>
#include <type_traits>

#include "d4est_hip_direct.h"
#include "d4est_hip_internal.h"
#include "d4est_hip_mwave.h"
#include "d4est_hip_tables.h"
#include "d4est_hip_wave.h"

namespace d4est_hip {

namespace {

constexpr int imax(int a, int b) { return a > b ? a : b; }
constexpr int iup64(int a) { return (a + 63) / 64 * 64; }

template <int N>
struct MwCfg {
  using W = WaveCfg<N, N>;
  static constexpr int N2 = N * N, N3 = N2 * N, T = N2, PN = N | 1, RS = N | 1;
  static constexpr int THREADS = W::THREADS, PL = W::PL, FS = W::FS;
  static constexpr int GS = N * RS;                 // field stride of the line images [field][line][entry]
  static constexpr int QS = T + 8, VS = N2 + 8;     // mortar values [block][a' + N b'], lifted fields [block][a + N b]
  // transposition buffer: 8 nodal fields in / 12 pass-1 fields out, 8 mortar-value blocks, 8 term fields, 6 lifted fields
  static constexpr int S_DOUBLES = imax(imax(12 * GS, 8 * QS), imax(FS, 6 * VS));
  static constexpr size_t LDS_BYTES = (size_t)(FS + S_DOUBLES) * sizeof(double);
  static constexpr int OPSZ = N * N;
};

// A pass = nA line tasks with operator A followed by nB with operator B.  Group B starts on a wavefront boundary when the workgroup
// has the threads for it in the same round (a wavefront then issues ONE product); otherwise right behind group A.
template <int THREADS, int nA, int nB>
struct PassMap {
  static constexpr int oB = (iup64(nA) + nB <= (nA + nB + THREADS - 1) / THREADS * THREADS) ? iup64(nA) : nA;
  static constexpr int END = oB + nB;
  static constexpr int ROUNDS = (END + THREADS - 1) / THREADS;
  // slot -> task (group A: [0, nA), group B: nA + [0, nB)), or -1
  __device__ static __forceinline__ int task(int slot) { return slot < nA ? slot : ((slot >= oB && slot < END) ? slot - oB + nA : -1); }
};

#ifndef D4EST_HIP_MWD_ABLATE
#define D4EST_HIP_MWD_ABLATE 0   /* timing experiments only (wrong results): 1 no line loads, 2 no products, 4 no volume term, 8 no factor loads, 16 no workgroup barriers in the face part */
#endif
#if (D4EST_HIP_MWD_ABLATE & 16)
#define MW_SYNC() wave_lds_fence()
#else
#define MW_SYNC() __syncthreads()
#endif
#ifndef D4EST_HIP_MWD_WAVES
#define D4EST_HIP_MWD_WAVES 4
#endif
// diagnostic build only (-DD4EST_HIP_MWD_STAMPS=1, tools/stamps_mw.py): wave 0 of every workgroup writes s_memtime at the phase
// boundaries into the buffer handed over as the (otherwise unused) ghost trace argument; never compiled into the product
#ifndef D4EST_HIP_MWD_STAMPS
#define D4EST_HIP_MWD_STAMPS 0
#endif
#if D4EST_HIP_MWD_STAMPS
#define MW_STAMP(k) do { if (ghost_qtrace && threadIdx.x == 0) ((unsigned long long*)ghost_qtrace)[(size_t)blockIdx.x * 40 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MW_STAMP(k) do { } while (0)
#endif

// y = M x for a centro-symmetric / -antisymmetric (ANTI) N x N operator in the even-odd form: full table rows through one base pointer
// (contract_rows_eo_imm: 2-3 scalar instructions per row)
template <int N, bool ANTI>
__device__ __forceinline__ void face_prod(const double* __restrict__ tab, const double* x, double* y) {
  constexpr int HC = (N + 1) / 2;
  double xe[HC], xo[HC], ab[N];
  eo_pre<N>(x, xe, xo);
  contract_rows_eo_imm<HC, N, false>(tab, ANTI ? xo : xe, ANTI ? xe : xo, ab);
  eo_post<N>(ab, y);
}

// y = (A or B) x for the line task of this thread; w0 = first slot of the thread's wavefront in this round (wave-uniform: a wavefront
// that holds tasks of one group issues that product alone, one that straddles the groups issues both and each thread keeps its own)
template <int N, int nA, int oB, int END, bool ANTI_A, bool ANTI_B>
__device__ __forceinline__ void pass_product(const double* __restrict__ tabA, const double* __restrict__ tabB, int w0, bool isB,
                                             const double* x, double* y) {
  if constexpr ((D4EST_HIP_MWD_ABLATE & 2) != 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i];
    return;
  }
  const bool hasA = w0 < nA, hasB = (w0 + 64 > oB) && (w0 < END);
  if (!hasA && !hasB) return;   // no task in this wavefront (its threads' y is never stored)
  double ya[N], yb[N];
  if (hasA && hasB) {
    face_prod<N, ANTI_A>(tabA, x, ya);
    face_prod<N, ANTI_B>(tabB, x, yb);
#pragma unroll
    for (int i = 0; i < N; ++i) ya[i] = isB ? yb[i] : ya[i];
  } else if (hasB) {
    face_prod<N, ANTI_B>(tabB, x, ya);
  } else {
    face_prod<N, ANTI_A>(tabA, x, ya);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = ya[i];
}

template <int N>
__device__ __forceinline__ double mw_row_dot(const double* __restrict__ drow, const double* x) {
  double s = 0.0;
#pragma unroll
  for (int o0 = 0; o0 < N; o0 += 8) {
    sdouble_ptr row = launder(drow + o0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (o0 + i < N) s = (o0 + i == 0) ? row[i] * x[0] : fma(row[i], x[o0 + i], s);
  }
  return s;
}

// task of a thread in a pass, packed: line-in-field | field-or-group index << 8 | group B << 30; -1: none
template <typename PM, int N, int nA>
__device__ __forceinline__ int pack_task(int slot) {
  const int tk = PM::task(slot);
  if (tk < 0) return -1;
  const bool isB = tk >= nA;
  const int l = isB ? tk - nA : tk;
  return (l % N) | ((l / N) << 8) | (isB ? (1 << 30) : 0);
}

// VOL: 0 the face terms only (Au += ...), 1 the whole operator with the streamed metric, 2 with the affine metric; + 4: with the
// zeroth-order term V^T w J c V u of plan_set_lhs_coefficient in the volume stage.
// The parameter list is faces_direct_kernel's (DirectKernargs mirrors it: arguments needed late are read from the kernel-argument
// segment at their use instead of being held -- and spilled -- in scalar registers for the whole kernel).
template <int N, bool FUSE, int VOL>
__global__ __launch_bounds__((MwCfg<N>::THREADS), (N <= 13 ? D4EST_HIP_MWD_WAVES : (N == 14 ? 3 : 2))) void operator_mw_kernel(
    const double* __restrict__ u, const double* __restrict__ ghost_qtrace, double* __restrict__ Au, const DirectSide* __restrict__ sides,
    const DirectGhostOff* __restrict__ ghost_off, const double* __restrict__ ops, const double* __restrict__ geom,
    const double* __restrict__ bndry_q, const double* __restrict__ robin_c, const double* __restrict__ robin_r, int n_elem, int ns0,
    int ns_stride, int xcd_chunk, DirectFuse cf, DirectVol vol, const int* __restrict__ elem_list) {
  using C = MwCfg<N>;
  constexpr int N2 = C::N2, T = C::T, PN = C::PN, RS = C::RS, GS = C::GS, QS = C::QS, VS = C::VS, PL = C::PL, TH = C::THREADS;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* R0 = smem;            // u_e, then (K u)_e, then the accumulator of the lifted face terms: [i + PN (j + N k)]
  double* S = smem + C::FS;     // second field of the volume term, then the transposition buffer of the face passes
#define tC (ops)
#define tCD (ops + C::OPSZ)
#define tE (ops + 2 * C::OPSZ)
#define tDtE (ops + 3 * C::OPSZ)
#define dr0 (ops + 4 * C::OPSZ) /* D[0][:], then D[N-1][:] */

  const int te = threadIdx.x;
  const int a = te % N, b = te / N;
  const bool on = te < PL;   // the thread owns face node (a, b) / mortar node te / the volume term's line
  const int wave0 = __builtin_amdgcn_readfirstlane(te & ~63);
  // XCD-aware element order (workgroups are dealt round-robin to the 8 XCDs, each with its own L2): XCD x walks the x-th contiguous
  // (Morton-local) eighth of the elements, so a neighbour's u is more often in the reader's L2
  const int v = blockIdx.x;
  const int slot = xcd_chunk > 0 ? (v & 7) * xcd_chunk + (v >> 3) : v;
  if (slot >= n_elem) return;
  const int e = elem_list ? __builtin_amdgcn_readfirstlane(elem_list[slot]) : slot;
  const int ns = __builtin_amdgcn_readfirstlane(ns_stride >= 0 ? ns0 + e * ns_stride : sides[6 * (size_t)e].pad);   // (see faces_direct_kernel)
  MW_STAMP(0);

  // the six side descriptors now (scalar loads; two dependent round trips -- kernel-argument segment, then the array -- that would
  // otherwise sit in front of every direction's line loads)
  int kcf6[6], sgeom6[6], nbr6[6];
  {
    typedef const DirectSide __attribute__((address_space(4))) * sside_ptr;
    const sside_ptr sd = (sside_ptr)(unsigned long long)(sides + 6 * (size_t)e);
#pragma unroll
    for (int f = 0; f < 6; ++f) { kcf6[f] = sd[f].kcf; sgeom6[f] = sd[f].geom; nbr6[f] = sd[f].nbr_ns; }
  }
  // ---- R0 <- (K u)_e, or the A u the face terms are added to
  if constexpr (VOL != 0) {
    if (on) load_element_image<N, PL, PN>(R0, u + ns, te);
    const DirectVol vl = direct_load_vol(direct_kargs());
    const int qs = __builtin_amdgcn_readfirstlane(vl.qs_stride >= 0 ? vl.qs0 + e * vl.qs_stride : vl.qs_list[e]);
    unsigned long long* st_ = (D4EST_HIP_MWD_STAMPS && ghost_qtrace) ? (unsigned long long*)ghost_qtrace + (size_t)blockIdx.x * 40 + 34 : nullptr;
    if constexpr ((D4EST_HIP_MWD_ABLATE & 4) == 0) {
#if D4EST_HIP_MW_COLLOCATED
      stiffness_mw_element_cg<N, (VOL & 3) == 2, (VOL & 4) != 0, (VOL & 8) != 0>(R0, S, vl.metric, qs, e, on, te, a, b, vl.EBb, vl.EBf, vl.EDq, vl.EDqT, vl.affine,
                                                                 vl.wq, vl.cq, st_);
#else
      stiffness_mw_element<N, N, false, true, (VOL & 3) == 2, (VOL & 4) != 0, (VOL & 8) != 0>(R0, S, vl.metric, qs, e, on, te, a, b, vl.EBb, vl.EGb, vl.EBf,
                                                                              vl.EGf, vl.affine, vl.wq, vl.cq, st_);
#endif
    }
    else __syncthreads();
  } else {
    if (on) load_element_image<N, PL, PN>(R0, direct_kargs()->Au + ns, te);
    __syncthreads();
  }

  MW_STAMP(1);
  // ---- the thread's line tasks in the four passes (the same in every direction; packed: see pack_task)
  using PM1 = PassMap<TH, 8 * N, 4 * N>;    // C on the 8 nodal fields | C D on the 4 trace fields
  using PM2 = PassMap<TH, 12 * N, 4 * N>;   // C on P_0..7, R_0..3 | C D on P_0..3
  using PM3 = PassMap<TH, 6 * N, 2 * N>;    // E on six term fields | D^T E on the two that are differentiated along a
  using PM4 = PassMap<TH, 4 * N, 2 * N>;    // E: (h, g) = (0,0) (0,2) (1,0) (1,2) | D^T E: (0,1) (1,1)
  int tk1[PM1::ROUNDS], tk2[PM2::ROUNDS], tk3[PM3::ROUNDS], tk4[PM4::ROUNDS];
#pragma unroll
  for (int r = 0; r < PM1::ROUNDS; ++r) tk1[r] = pack_task<PM1, N, 8 * N>(te + r * TH);
#pragma unroll
  for (int r = 0; r < PM2::ROUNDS; ++r) tk2[r] = pack_task<PM2, N, 12 * N>(te + r * TH);
#pragma unroll
  for (int r = 0; r < PM3::ROUNDS; ++r) tk3[r] = pack_task<PM3, N, 6 * N>(te + r * TH);
#pragma unroll
  for (int r = 0; r < PM4::ROUNDS; ++r) tk4[r] = pack_task<PM4, N, 4 * N>(te + r * TH);
  const int ab_rs = b * RS + a;

  auto dir_body = [&](auto dc) {
    constexpr int d = decltype(dc)::value;
    constexpr int t0 = (d == 0) ? 1 : 0, t1d = (d == 2) ? 1 : 2;   // reference directions of the face indices a and b
    const int kcf[2] = {kcf6[2 * d], kcf6[2 * d + 1]};
    const int sgeom[2] = {sgeom6[2 * d], sgeom6[2 * d + 1]};
    // ---- nodal fields of the two faces at face node (a, b): c = 0..3 trace (own 2d, own 2d+1, nbr 2d, nbr 2d+1), c = 4..7 normal
    // derivative.  The normal lines of the element and of the two (+) elements (at THEIR face node (a, b)) are requested together.
    double fld[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (on) {
      double xo[N], yy[2][N];
      const double* __restrict__ up = u + ns;
      constexpr int st = (d == 0) ? 1 : (d == 1 ? N : N2);
      const int o0 = (d == 0) ? N * te : (d == 1 ? a + N2 * b : te);   // N a + N2 b = N te;  a + N b = te
#pragma unroll
      for (int i = 0; i < N; ++i) xo[i] = (D4EST_HIP_MWD_ABLATE & 1) ? 1.0 + i : up[o0 + st * i];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if ((kcf[h] & 3) == 1) {
          const double* __restrict__ upn = u + nbr6[2 * d + h];
          const int dp = kcf[h] >> 6;
          const int stn = (dp == 0) ? 1 : (dp == 1 ? N : N2);
          const int on0 = (dp == 0) ? N * te : (dp == 1 ? a + N2 * b : te);
#pragma unroll
          for (int i = 0; i < N; ++i) yy[h][i] = (D4EST_HIP_MWD_ABLATE & 1) ? 2.0 + i + stn + on0 : upn[on0 + stn * i];
        }
      }
      fld[0] = xo[0];
      fld[1] = xo[N - 1];
      fld[4] = mw_row_dot<N>(dr0, xo);
      fld[5] = mw_row_dot<N>(dr0 + N, xo);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if ((kcf[h] & 3) == 1) {
          const int hi = (kcf[h] >> 5) & 1;
          fld[2 + h] = hi ? yy[h][N - 1] : yy[h][0];
          fld[6 + h] = mw_row_dot<N>(dr0 + hi * N, yy[h]);
        }
      }
    }
    MW_STAMP(2 + 10 * d);   // lines loaded, nodal fields formed
    // ---- pass 1: line (field c, b), contract the face index a:  P_c = C x_c (c = 0..7), R_c = C D x_c (trace fields c = 0..3)
    MW_SYNC();   // (the buffer's last readers: the volume term / the previous direction's line update)
    if (on) {
#pragma unroll
      for (int c = 0; c < 8; ++c) S[c * N * RS + ab_rs] = fld[c];
    }
    MW_SYNC();
    {
      double x[PM1::ROUNDS][N], y[PM1::ROUNDS][N];
#pragma unroll
      for (int r = 0; r < PM1::ROUNDS; ++r) {   // (threads without a task read line 0: their product goes nowhere)
        const int in = tk1[r] < 0 ? 0 : (((tk1[r] >> 8) & 0xff) * N + (tk1[r] & 0xff)) * RS;
#pragma unroll
        for (int i = 0; i < N; ++i) x[r][i] = lds_ld(&S[in + i]);
      }
      MW_SYNC();
      MW_STAMP(3 + 10 * d);   // staged, rows read
#pragma unroll
      for (int r = 0; r < PM1::ROUNDS; ++r) {
        const bool isB = (tk1[r] >> 30) & 1;
        pass_product<N, 8 * N, PM1::oB, PM1::END, false, true>(tC, tCD, wave0 + r * TH, isB, x[r], y[r]);
        if (tk1[r] >= 0) {   // output field 0..7: P_c, 8..11: R_c; line b: [field][a'][b]
          const int out = (((tk1[r] >> 8) & 0xff) + (isB ? 8 : 0)) * GS + (tk1[r] & 0xff);
#pragma unroll
          for (int q = 0; q < N; ++q) S[out + q * RS] = y[r][q];
        }
      }
    }
    MW_SYNC();
    MW_STAMP(4 + 10 * d);   // pass 1 done
    // the two faces' geometric factors (HBM) are requested here: their latency passes under the pass-2 products
    double gqa[2][7];
    {
      const direct_kargs_ptr K = direct_kargs();
      const double* __restrict__ geom_ = K->geom;
      const double* __restrict__ robin_c_ = K->robin_c;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 7; ++c) gqa[h][c] = 0.0;
        if (on) {
          if ((kcf[h] & 3) == 0 && robin_c_) {
            gqa[h][6] = robin_c_[sgeom[h] + te];   // am = ap = 0: no term 1 / term 2 on a Robin side
          } else {
            const double* __restrict__ g = geom_ + (size_t)7 * sgeom[h] + te;
#pragma unroll
            for (int c = 0; c < 7; ++c) gqa[h][c] = (D4EST_HIP_MWD_ABLATE & 8) ? 0.5 + c : ld_sel<(VOL & 8) != 0>(&g[c * T]);   // (stream mode)
          }
        }
      }
    }
    // ---- pass 2: line (field, a'), contract the face index b:  u = C P_c, du/dn = C P_{4+c}, du/dt_a = C R_c | du/dt_b = C D P_c
    double o2[PM2::ROUNDS][N];
    {
      double x[PM2::ROUNDS][N];
#pragma unroll
      for (int r = 0; r < PM2::ROUNDS; ++r) {
        const int in = tk2[r] < 0 ? 0 : ((tk2[r] >> 8) & 0xff) * GS + (tk2[r] & 0xff) * RS;
#pragma unroll
        for (int i = 0; i < N; ++i) x[r][i] = lds_ld(&S[in + i]);
      }
#pragma unroll
      for (int r = 0; r < PM2::ROUNDS; ++r)
        pass_product<N, 12 * N, PM2::oB, PM2::END, false, true>(tC, tCD, wave0 + r * TH, (tk2[r] >> 30) & 1, x[r], o2[r]);
    }
    MW_STAMP(5 + 10 * d);   // pass 2 products done
    // ---- SIPG terms, one face at a time (the mortar values of its two sides go through the buffer)
    double At[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      MW_SYNC();
#pragma unroll
      for (int r = 0; r < PM2::ROUNDS; ++r) {
        if (tk2[r] >= 0) {
          // field index f = 0..11 (group A: P_c, P_{4+c}, R_c) or 0..3 (group B: P_c); c = f & 3: (own h0, own h1, nbr h0, nbr h1)
          const int f = (tk2[r] >> 8) & 0xff, c = f & 3, aq = tk2[r] & 0xff;
          const int g = ((tk2[r] >> 30) & 1) ? 3 : (f >> 2);   // 0 u, 1 du/dn, 2 du/dt_a, 3 du/dt_b
          if ((c & 1) == h) {
            const int mp = c >> 1;
            const int dn = mp ? (kcf[h] >> 6) : d;   // the reference frame of the side that owns the trace
            const int ta = (dn == 0) ? 1 : 0, tb = (dn == 2) ? 1 : 2;
            const int comp = (g == 0) ? 0 : (g == 1 ? 1 + dn : (g == 2 ? 1 + ta : 1 + tb));
            const int out = (mp * 4 + comp) * QS + aq;
#pragma unroll
            for (int q = 0; q < N; ++q) S[out + N * q] = o2[r][q];
          }
        }
      }
      MW_SYNC();
      double qm[4] = {0, 0, 0, 0}, qp[4] = {0, 0, 0, 0};
      const int kind = kcf[h] & 3, code = (kcf[h] >> 2) & 7;
      if (on) {
        const int k = te;
#pragma unroll
        for (int c = 0; c < 4; ++c) qm[c] = lds_ld(&S[c * QS + k]);
        if (kind == 1) {
          const int kp = reorder_index(code, N - 1, a, b);
#pragma unroll
          for (int c = 0; c < 4; ++c) qp[c] = lds_ld(&S[(4 + c) * QS + kp]);
        } else if (kind == 2) {
          const direct_kargs_ptr K = direct_kargs();
          const double* __restrict__ p = K->ghost_qtrace + K->ghost_off[6 * (size_t)e + 2 * d + h] + reorder_index(code, N - 1, a, b);
#pragma unroll
          for (int c = 0; c < 4; ++c) qp[c] = p[c * T];
        } else {
          const direct_kargs_ptr K = direct_kargs();
          qp[0] = K->robin_c ? K->robin_r[sgeom[h] + k] : K->bndry_q[sgeom[h] + k];
        }
      }
      // interface: t1 = -1/2 sj n.(grad u_m + grad u_p), t2_l = -1/2 am_l [u]; boundary: t1 = -sj n.grad u_m, t2_l = -am_l (u - g)
      // (d4est_laplacian_flux_sipg.c:494-942, :15-336); Robin (:339-489): sj (coeff u_m - rhs) only
      double t1 = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) t1 += gqa[h][i] * qm[1 + i] + gqa[h][3 + i] * qp[1 + i];   // gq[3..5] = 0 on boundary sides
      const double jump = qm[0] - qp[0];
      const double w1 = (kind != 0) ? -0.5 : -1.0;
      const bool robin = kind == 0 && direct_kargs()->robin_c;
      At[h][0] = robin ? gqa[h][6] * qm[0] - qp[0] : w1 * t1 + gqa[h][6] * jump;
#pragma unroll
      for (int l = 0; l < 3; ++l) At[h][1 + l] = w1 * gqa[h][l] * jump;
    }
    MW_STAMP(6 + 10 * d);   // SIPG terms of both faces
    // ---- lift pass 1: line (term field 4 h + c, b'), contract a':  E, and D^T E for the term-2 field that is differentiated along a
    // (val = E_b E_a A0 + E_b (D^T E)_a A_t0 + (D^T E)_b E_a A_t1)
    MW_SYNC();
    if (on) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) S[(4 * h + c) * N * RS + ab_rs] = At[h][c];
    }
    MW_SYNC();
    {
      double x[PM3::ROUNDS][N], y[PM3::ROUNDS][N];
      int fl[PM3::ROUNDS];
#pragma unroll
      for (int r = 0; r < PM3::ROUNDS; ++r) {
        // group A: the six fields (h, c != 1 + t0), j = 3 h + rank of c; group B: (h = j, 1 + t0)
        const int j = (tk3[r] >> 8) & 0xff;
        const bool isB = (tk3[r] >> 30) & 1;
        const int hh = isB ? j : (j >= 3 ? 1 : 0), rr = j - 3 * hh;
        const int c = isB ? 1 + t0 : rr + (rr >= 1 + t0 ? 1 : 0);
        fl[r] = tk3[r] < 0 ? 0 : 4 * hh + c;
        const int in = tk3[r] < 0 ? 0 : (fl[r] * N + (tk3[r] & 0xff)) * RS;
#pragma unroll
        for (int i = 0; i < N; ++i) x[r][i] = lds_ld(&S[in + i]);
      }
      MW_SYNC();
#pragma unroll
      for (int r = 0; r < PM3::ROUNDS; ++r) {
        pass_product<N, 6 * N, PM3::oB, PM3::END, false, true>(tE, tDtE, wave0 + r * TH, (tk3[r] >> 30) & 1, x[r], y[r]);
        if (tk3[r] >= 0) {   // [field][a][b']
          const int out = fl[r] * GS + (tk3[r] & 0xff);
#pragma unroll
          for (int i = 0; i < N; ++i) S[out + i * RS] = y[r][i];
        }
      }
    }
    MW_SYNC();
    MW_STAMP(7 + 10 * d);   // lift 1 done
    // ---- lift pass 2: line (h, g, a), contract b': g = 0 face-local part through E, 1 term 2 along b through D^T E, 2 normal term 2
    {
      double x[PM4::ROUNDS][N], y[PM4::ROUNDS][N];
      int blk[PM4::ROUNDS];
#pragma unroll
      for (int r = 0; r < PM4::ROUNDS; ++r) {
        const int j = (tk4[r] >> 8) & 0xff, la = tk4[r] & 0xff;
        const bool isB = (tk4[r] >> 30) & 1;
        const int vh = isB ? j : j >> 1, vg = isB ? 1 : 2 * (j & 1);
        blk[r] = 3 * vh + vg;
        const int f1 = (vg == 0) ? 0 : (vg == 1 ? 1 + t1d : 1 + d);
        const int in = tk4[r] < 0 ? 0 : (4 * vh + f1) * GS + la * RS;
        const int in2 = (4 * vh + 1 + t0) * GS + la * RS;
        const bool two = tk4[r] >= 0 && vg == 0;
#pragma unroll
        for (int q = 0; q < N; ++q) {
          double t = lds_ld(&S[in + q]);
          if (two) t += lds_ld(&S[in2 + q]);
          x[r][q] = t;
        }
      }
      MW_SYNC();
#pragma unroll
      for (int r = 0; r < PM4::ROUNDS; ++r) {
        pass_product<N, 4 * N, PM4::oB, PM4::END, false, true>(tE, tDtE, wave0 + r * TH, (tk4[r] >> 30) & 1, x[r], y[r]);
        if (tk4[r] >= 0) {
          const int out = blk[r] * VS + (tk4[r] & 0xff);
#pragma unroll
          for (int i = 0; i < N; ++i) S[out + i * N] = y[r][i];
        }
      }
    }
    MW_SYNC();
    MW_STAMP(8 + 10 * d);   // lift 2 done
    // ---- the element's normal line at face node (a, b): face-local part at its two ends, D^T of the normal term 2 along it
    if (on) {
      const double val0 = lds_ld(&S[0 * VS + te]) + lds_ld(&S[1 * VS + te]), n0 = lds_ld(&S[2 * VS + te]);
      const double val1 = lds_ld(&S[3 * VS + te]) + lds_ld(&S[4 * VS + te]), n1 = lds_ld(&S[5 * VS + te]);
      double acc[N];
#pragma unroll
      for (int o0 = 0; o0 < N; o0 += 8) {
        sdouble_ptr r0 = launder(dr0 + o0), r1 = launder(dr0 + N + o0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (o0 + i < N) acc[o0 + i] = fma(r0[i], n0, r1[i] * n1);
      }
      acc[0] += val0;
      acc[N - 1] += val1;
      const int base = (d == 0) ? PN * te : (d == 1 ? a + PN * N * b : a + PN * b);
      constexpr int str = (d == 0) ? 1 : (d == 1 ? PN : PN * N);
#pragma unroll
      for (int i = 0; i < N; ++i) R0[base + str * i] += acc[i];
    }
  };
  dir_body(std::integral_constant<int, 0>{});
  dir_body(std::integral_constant<int, 1>{});
  dir_body(std::integral_constant<int, 2>{});
  __syncthreads();
  MW_STAMP(32);

  // ---- A u (and the Chebyshev update of the node: cheby_update_kernel, same roundings; u is an INPUT of this kernel -- the
  // neighbours read it --, so the new iterate goes to a second vector)
  if (on) {
    constexpr int NL = N;   // PL = N^2: one k-plane per pass
    const int ij = (te % N) + PN * (te / N);
    double* __restrict__ Au_ = direct_kargs()->Au;
    if constexpr (FUSE) {
      const DirectFuse cfl = direct_load_fuse(direct_kargs());
      constexpr int BT = 4;   // planes per batch: the smoother vectors of a batch are requested before the first is used
#pragma unroll
      for (int q0 = 0; q0 < NL; q0 += BT) {
        double rh[BT], pp[BT], uu[BT];
#pragma unroll
        for (int q = 0; q < BT; ++q) {
          if (q0 + q < NL) {
            const size_t o = (size_t)ns + te + PL * (q0 + q);
            rh[q] = cfl.rhs[o];
            pp[q] = cfl.p[o];
            uu[q] = u[o];
          }
        }
#pragma unroll
        for (int q = 0; q < BT; ++q) {
          if (q0 + q < NL) {
            const size_t o = (size_t)ns + te + PL * (q0 + q);
            const double au = R0[ij + PN * N * (q0 + q)];
            if (!cfl.skip_Au_store) Au_[o] = au;
            const double res = __dadd_rn(rh[q], __dmul_rn(-1.0, au));
            const double ri = __dmul_rn(cfl.alpha, res);
            const double pi = __dadd_rn(__dmul_rn(cfl.beta, pp[q]), ri);
            if (cfl.r) cfl.r[o] = ri;
            cfl.p[o] = pi;
            cfl.u_out[o] = __dadd_rn(uu[q], pi);
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        if constexpr ((VOL & 8) != 0) __builtin_nontemporal_store(R0[ij + PN * N * q], &Au_[(size_t)ns + te + PL * q]);   // stream mode
        else Au_[(size_t)ns + te + PL * q] = R0[ij + PN * N * q];
      }
    }
  }
  MW_STAMP(33);
#undef tC
#undef tCD
#undef tE
#undef tDtE
#undef dr0
}

}  // namespace

// The instantiations are spread over two translation units so that the build's longest compile halves: this file holds N = 9 ... 12,
// d4est_hip_direct_mw_hi.hip includes it with D4EST_HIP_MW_PART = 1 and holds N = 13 ... 16.
#ifndef D4EST_HIP_MW_PART
#define D4EST_HIP_MW_PART 0
#endif
#ifdef D4EST_HIP_MW_ONLY   /* development builds: one size */
#define D4EST_HIP_DIRECT_MW_SIZES(X) X(D4EST_HIP_MW_ONLY)
#define D4EST_HIP_DIRECT_MW_SIZES_HERE(X) X(D4EST_HIP_MW_ONLY)
#else
#define D4EST_HIP_DIRECT_MW_SIZES(X) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#if D4EST_HIP_MW_PART == 0
#define D4EST_HIP_DIRECT_MW_SIZES_HERE(X) X(9) X(10) X(11) X(12)
#else
#define D4EST_HIP_DIRECT_MW_SIZES_HERE(X) X(13) X(14) X(15) X(16)
#endif
#endif

#if D4EST_HIP_MW_PART == 0
bool direct_mw_built(int N, int NQ) {
  if (N != NQ) return false;
#define X(N_) if (N == N_) return true;
  D4EST_HIP_DIRECT_MW_SIZES(X)
#undef X
  return false;
}
bool launch_direct_mw_hi(d4est_hip_plan* plan, DirectHost* dh, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf,
                         const double* robin_c, const double* robin_r, int vmode, const DirectVol& vol, int n, int chunk);
#define D4EST_HIP_MW_LAUNCHER void launch_direct_mw
#else
#define D4EST_HIP_MW_LAUNCHER bool launch_direct_mw_hi
#endif

// the dynamic-LDS attribute of a kernel instance is set ONCE (a static flag per instantiation of this template), not at every launch:
// the launch sits in the hot path of a Chebyshev iteration / a Schwarz CG sweep, and inside hipGraph capture regions
template <typename K, K Kernel>
static void mw_set_lds_limit_once(size_t bytes) {
  static bool done = false;
  if (done) return;
  if (bytes > 48 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done = true;
}

// vmode: VOL of operator_mw_kernel; + 8 (stream mode, plan->stream_mode) exists for the whole operator with the streamed metric (1 -> 9)
D4EST_HIP_MW_LAUNCHER(d4est_hip_plan* plan, DirectHost* dh, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf,
                      const double* robin_c, const double* robin_r, int vmode, const DirectVol& vol, int n, int chunk) {
  const DirectFuse cfv = cf ? *cf : DirectFuse{};
  bool done = false;
  if (vmode == 1 && vol.stream) vmode = 9;
#define D4EST_HIP_MW_GO(N_, FUSE_, VOL_)                                                                                          \
  do {                                                                                                                             \
    mw_set_lds_limit_once<decltype(&operator_mw_kernel<N_, FUSE_, VOL_>), &operator_mw_kernel<N_, FUSE_, VOL_>>(MwCfg<N_>::LDS_BYTES);  \
    hipLaunchKernelGGL((operator_mw_kernel<N_, FUSE_, VOL_>), dim3(n), dim3(MwCfg<N_>::THREADS), MwCfg<N_>::LDS_BYTES, plan->stream, \
                       u, ghost_trace, Au, dh->d_sides, dh->d_ghost_off, dh->d_ops, plan->d_face_geom, plan->d_bndry, robin_c,     \
                       robin_r, n, dh->ns0, dh->ns_stride, chunk, cfv, vol, dh->d_list);                                           \
  } while (0)
#define X(N_)                                                                                       \
  if (!done && dh->N == N_) {                                                                       \
    if (vmode == 1) { if (cf) D4EST_HIP_MW_GO(N_, true, 1); else D4EST_HIP_MW_GO(N_, false, 1); }   \
    else if (vmode == 9) { if (cf) D4EST_HIP_MW_GO(N_, true, 9); else D4EST_HIP_MW_GO(N_, false, 9); } \
    else if (vmode == 2) { if (cf) D4EST_HIP_MW_GO(N_, true, 2); else D4EST_HIP_MW_GO(N_, false, 2); } \
    else if (vmode == 5) { if (cf) D4EST_HIP_MW_GO(N_, true, 5); else D4EST_HIP_MW_GO(N_, false, 5); } \
    else if (vmode == 6) { if (cf) D4EST_HIP_MW_GO(N_, true, 6); else D4EST_HIP_MW_GO(N_, false, 6); } \
    else { if (cf) D4EST_HIP_MW_GO(N_, true, 0); else D4EST_HIP_MW_GO(N_, false, 0); }              \
    done = true;                                                                                    \
  }
  D4EST_HIP_DIRECT_MW_SIZES_HERE(X)
#undef X
#undef D4EST_HIP_MW_GO
#if D4EST_HIP_MW_PART == 0
#ifndef D4EST_HIP_MW_ONLY
  if (!done) done = launch_direct_mw_hi(plan, dh, u, ghost_trace, Au, cf, robin_c, robin_r, vmode, vol, n, chunk);
#endif
  if (!done) D4EST_HIP_ABORT("multi-wave direct kernel: no instance for N = %d", dh->N);
  HIP_CHECK(hipGetLastError());
#else
  if (done) HIP_CHECK(hipGetLastError());
  return done;
#endif
}

}  // namespace d4est_hip
