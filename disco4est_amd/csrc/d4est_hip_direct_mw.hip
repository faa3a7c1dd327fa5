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

// y = (A or B) x for the line task of this thread; w0 = first slot of the thread's wavefront in this round (wave-uniform)
template <int N, int nA, int oB, int END, bool ANTI_A, bool ANTI_B>
__device__ __forceinline__ void pass_product(const double* __restrict__ tabA, const double* __restrict__ tabB, int w0, bool isB,
                                             const double* x, double* y) {
  const bool hasA = w0 < nA, hasB = (w0 + 64 > oB) && (w0 < END);   // wave-uniform: a wavefront that straddles the groups issues both
  double ya[N], yb[N];
#pragma unroll
  for (int i = 0; i < N; ++i) ya[i] = yb[i] = 0.0;
  if (hasA) fwd<N, N, true, ANTI_A>(tabA, x, ya);
  if (hasB) fwd<N, N, true, ANTI_B>(tabB, x, yb);
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = isB ? yb[i] : ya[i];
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

#ifndef D4EST_HIP_MWD_WAVES
#define D4EST_HIP_MWD_WAVES 4
#endif

// VOL: 0 the face terms only (Au += ...), 1 the whole operator with the streamed metric, 2 with the affine metric
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
  const double* tC = ops;
  const double* tCD = ops + C::OPSZ;
  const double* tE = ops + 2 * C::OPSZ;
  const double* tDtE = ops + 3 * C::OPSZ;
  const double* dr0 = ops + 4 * C::OPSZ;   // D[0][:], then D[N-1][:]

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
  const int ns = __builtin_amdgcn_readfirstlane(ns0 + e * ns_stride);

  // ---- R0 <- (K u)_e, or the A u the face terms are added to
  if constexpr (VOL != 0) {
    if (on) load_element_image<N, PL, PN>(R0, u + ns, te);
    const int qs = __builtin_amdgcn_readfirstlane(vol.qs_stride >= 0 ? vol.qs0 + e * vol.qs_stride : vol.qs_list[e]);
    stiffness_mw_element<N, N, false, true, VOL == 2>(R0, S, vol.metric, qs, e, on, te, a, b, vol.EBb, vol.EGb, vol.EBf, vol.EGf,
                                                      vol.affine, vol.wq);
  } else {
    if (on) load_element_image<N, PL, PN>(R0, Au + ns, te);
    __syncthreads();
  }

  auto dir_body = [&](auto dc) {
    constexpr int d = decltype(dc)::value;
    constexpr int t0 = (d == 0) ? 1 : 0, t1d = (d == 2) ? 1 : 2;   // reference directions of the face indices a and b
    const DirectSide* sd = sides + 6 * (size_t)e + 2 * d;
    int kcf[2], sgeom[2], nbr_ns[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      kcf[h] = __builtin_amdgcn_readfirstlane(sd[h].kcf);
      sgeom[h] = __builtin_amdgcn_readfirstlane(sd[h].geom);
      nbr_ns[h] = __builtin_amdgcn_readfirstlane(sd[h].nbr_ns);
    }
    // ---- nodal fields of the two faces at face node (a, b): c = 0..3 trace (own 2d, own 2d+1, nbr 2d, nbr 2d+1), c = 4..7 normal
    // derivative.  The normal lines of the element and of the two (+) elements (at THEIR face node (a, b)) are requested together.
    double fld[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    {
      double xo[N], yy[2][N];
#pragma unroll
      for (int i = 0; i < N; ++i) xo[i] = yy[0][i] = yy[1][i] = 0.0;
      if (on) {
        const double* __restrict__ up = u + ns;
        constexpr int st = (d == 0) ? 1 : (d == 1 ? N : N2);
        const int o0 = (d == 0) ? N * a + N2 * b : (d == 1 ? a + N2 * b : a + N * b);
#pragma unroll
        for (int i = 0; i < N; ++i) xo[i] = up[o0 + st * i];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if ((kcf[h] & 3) == 1) {
            const double* __restrict__ upn = u + nbr_ns[h];
            const int dp = kcf[h] >> 6;
            const int stn = (dp == 0) ? 1 : (dp == 1 ? N : N2);
            const int on0 = (dp == 0) ? N * a + N2 * b : (dp == 1 ? a + N2 * b : a + N * b);
#pragma unroll
            for (int i = 0; i < N; ++i) yy[h][i] = upn[on0 + stn * i];
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
    }
    // ---- pass 1: line (field c, b), contract the face index a:  P_c = C x_c (c = 0..7), R_c = C D x_c (trace fields c = 0..3)
    __syncthreads();   // (the buffer's last readers: the volume term / the previous direction's line update)
    if (on) {
#pragma unroll
      for (int c = 0; c < 8; ++c) S[(c * N + b) * RS + a] = fld[c];
    }
    __syncthreads();
    {
      using PM = PassMap<TH, 8 * N, 4 * N>;
      double x[PM::ROUNDS][N], y[PM::ROUNDS][N];
      int tk[PM::ROUNDS];
#pragma unroll
      for (int r = 0; r < PM::ROUNDS; ++r) {
        tk[r] = PM::task(te + r * TH);
        const int line = tk[r] < 8 * N ? tk[r] : tk[r] - 8 * N;
#pragma unroll
        for (int i = 0; i < N; ++i) x[r][i] = tk[r] >= 0 ? lds_ld(&S[line * RS + i]) : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PM::ROUNDS; ++r) {
        pass_product<N, 8 * N, PM::oB, PM::END, false, true>(tC, tCD, wave0 + r * TH, tk[r] >= 8 * N, x[r], y[r]);
        if (tk[r] >= 0) {   // output field tk / N (0..7: P_c, 8..11: R_c), line b = tk % N: [field][a'][b]
          const int fo = tk[r] / N, lb = tk[r] % N;
#pragma unroll
          for (int q = 0; q < N; ++q) S[fo * GS + q * RS + lb] = y[r][q];
        }
      }
    }
    __syncthreads();
    // ---- pass 2: line (field, a'), contract the face index b:  u = C P_c, du/dn = C P_{4+c}, du/dt_a = C R_c | du/dt_b = C D P_c
    using PM2 = PassMap<TH, 12 * N, 4 * N>;
    double o2[PM2::ROUNDS][N];
    int tk2[PM2::ROUNDS];
    {
      double x[PM2::ROUNDS][N];
#pragma unroll
      for (int r = 0; r < PM2::ROUNDS; ++r) {
        tk2[r] = PM2::task(te + r * TH);
        const int line = tk2[r] < 12 * N ? tk2[r] : tk2[r] - 12 * N;   // = field * N + a'
#pragma unroll
        for (int i = 0; i < N; ++i) x[r][i] = tk2[r] >= 0 ? lds_ld(&S[(line / N) * GS + (line % N) * RS + i]) : 0.0;
      }
#pragma unroll
      for (int r = 0; r < PM2::ROUNDS; ++r)
        pass_product<N, 12 * N, PM2::oB, PM2::END, false, true>(tC, tCD, wave0 + r * TH, tk2[r] >= 12 * N, x[r], o2[r]);
    }
    // ---- SIPG terms, one face at a time (the mortar values of its two sides go through the buffer)
    double At[2][4];
    double gqa[2][7];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int c = 0; c < 7; ++c) gqa[h][c] = 0.0;
      if (on) {
        const int kind = kcf[h] & 3;
        if (kind == 0 && robin_c) {
          gqa[h][6] = robin_c[sgeom[h] + te];   // am = ap = 0: no term 1 / term 2 on a Robin side
        } else {
          const double* __restrict__ g = geom + (size_t)7 * sgeom[h] + te;
#pragma unroll
          for (int c = 0; c < 7; ++c) gqa[h][c] = g[c * T];
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PM2::ROUNDS; ++r) {
        if (tk2[r] >= 0) {
          const int g = tk2[r] / (4 * N), c = (tk2[r] / N) & 3, aq = tk2[r] % N;   // g: 0 u, 1 du/dn, 2 du/dt_a, 3 du/dt_b
          if ((c & 1) == h) {
            const int mp = c >> 1;
            const int dn = mp ? (kcf[h] >> 6) : d;   // the reference frame of the side that owns the trace
            const int ta = (dn == 0) ? 1 : 0, tb = (dn == 2) ? 1 : 2;
            const int comp = (g == 0) ? 0 : (g == 1 ? 1 + dn : (g == 2 ? 1 + ta : 1 + tb));
#pragma unroll
            for (int q = 0; q < N; ++q) S[(mp * 4 + comp) * QS + aq + N * q] = o2[r][q];
          }
        }
      }
      __syncthreads();
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
          const double* __restrict__ p = ghost_qtrace + ghost_off[6 * (size_t)e + 2 * d + h] + reorder_index(code, N - 1, a, b);
#pragma unroll
          for (int c = 0; c < 4; ++c) qp[c] = p[c * T];
        } else if (robin_c) {
          qp[0] = robin_r[sgeom[h] + k];
        } else {
          qp[0] = bndry_q[sgeom[h] + k];
        }
      }
      // interface: t1 = -1/2 sj n.(grad u_m + grad u_p), t2_l = -1/2 am_l [u]; boundary: t1 = -sj n.grad u_m, t2_l = -am_l (u - g)
      // (d4est_laplacian_flux_sipg.c:494-942, :15-336); Robin (:339-489): sj (coeff u_m - rhs) only
      double t1 = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) t1 += gqa[h][i] * qm[1 + i] + gqa[h][3 + i] * qp[1 + i];   // gq[3..5] = 0 on boundary sides
      const double jump = qm[0] - qp[0];
      const double w1 = (kind != 0) ? -0.5 : -1.0;
      At[h][0] = (kind == 0 && robin_c) ? gqa[h][6] * qm[0] - qp[0] : w1 * t1 + gqa[h][6] * jump;
#pragma unroll
      for (int l = 0; l < 3; ++l) At[h][1 + l] = w1 * gqa[h][l] * jump;
    }
    // ---- lift pass 1: line (term field 4 h + c, b'), contract a':  E, and D^T E for the term-2 field that is differentiated along a
    // (val = E_b E_a A0 + E_b (D^T E)_a A_t0 + (D^T E)_b E_a A_t1)
    __syncthreads();
    if (on) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) S[((4 * h + c) * N + b) * RS + a] = At[h][c];
    }
    __syncthreads();
    {
      using PM = PassMap<TH, 6 * N, 2 * N>;
      double x[PM::ROUNDS][N], y[PM::ROUNDS][N];
      int fl[PM::ROUNDS], lb[PM::ROUNDS];
      bool isB[PM::ROUNDS];
#pragma unroll
      for (int r = 0; r < PM::ROUNDS; ++r) {
        const int tk = PM::task(te + r * TH);
        isB[r] = tk >= 6 * N;
        // group A: the six fields (h, c != 1 + t0); group B: (h, 1 + t0)
        const int j = (isB[r] ? tk - 6 * N : tk) / N;
        const int hh = isB[r] ? j : j / 3, rr = j % 3;
        const int c = isB[r] ? 1 + t0 : rr + (rr >= 1 + t0 ? 1 : 0);
        fl[r] = tk >= 0 ? 4 * hh + c : -1;
        lb[r] = (tk >= 0 ? tk : 0) % N;
#pragma unroll
        for (int i = 0; i < N; ++i) x[r][i] = tk >= 0 ? lds_ld(&S[(fl[r] * N + lb[r]) * RS + i]) : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PM::ROUNDS; ++r) {
        pass_product<N, 6 * N, PM::oB, PM::END, false, true>(tE, tDtE, wave0 + r * TH, isB[r], x[r], y[r]);
        if (fl[r] >= 0) {   // [field][a][b']
#pragma unroll
          for (int i = 0; i < N; ++i) S[fl[r] * GS + i * RS + lb[r]] = y[r][i];
        }
      }
    }
    __syncthreads();
    // ---- lift pass 2: line (h, g, a), contract b': g = 0 face-local part through E, 1 term 2 along b through D^T E, 2 normal term 2
    {
      using PM = PassMap<TH, 4 * N, 2 * N>;
      double x[PM::ROUNDS][N], y[PM::ROUNDS][N];
      int blk[PM::ROUNDS], la[PM::ROUNDS];
      bool isB[PM::ROUNDS];
#pragma unroll
      for (int r = 0; r < PM::ROUNDS; ++r) {
        const int tk = PM::task(te + r * TH);
        isB[r] = tk >= 4 * N;
        const int j = (isB[r] ? tk - 4 * N : tk) / N;   // group A: (h, g) = (0,0) (0,2) (1,0) (1,2); group B: (0,1) (1,1)
        const int vh = isB[r] ? j : j >> 1, vg = isB[r] ? 1 : 2 * (j & 1);
        la[r] = (tk >= 0 ? tk : 0) % N;
        blk[r] = tk >= 0 ? 3 * vh + vg : -1;
        const int f1 = (vg == 0) ? 0 : (vg == 1 ? 1 + t1d : 1 + d);
#pragma unroll
        for (int q = 0; q < N; ++q) {
          double t = tk >= 0 ? lds_ld(&S[(4 * vh + f1) * GS + la[r] * RS + q]) : 0.0;
          if (tk >= 0 && vg == 0) t += lds_ld(&S[(4 * vh + 1 + t0) * GS + la[r] * RS + q]);
          x[r][q] = t;
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PM::ROUNDS; ++r) {
        pass_product<N, 4 * N, PM::oB, PM::END, false, true>(tE, tDtE, wave0 + r * TH, isB[r], x[r], y[r]);
        if (blk[r] >= 0) {
#pragma unroll
          for (int i = 0; i < N; ++i) S[blk[r] * VS + i * N + la[r]] = y[r][i];
        }
      }
    }
    __syncthreads();
    // ---- the element's normal line at face node (a, b): face-local part at its two ends, D^T of the normal term 2 along it
    if (on) {
      const double val0 = lds_ld(&S[0 * VS + b * N + a]) + lds_ld(&S[1 * VS + b * N + a]), n0 = lds_ld(&S[2 * VS + b * N + a]);
      const double val1 = lds_ld(&S[3 * VS + b * N + a]) + lds_ld(&S[4 * VS + b * N + a]), n1 = lds_ld(&S[5 * VS + b * N + a]);
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
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int idx = (d == 0) ? i + PN * (a + N * b) : (d == 1 ? a + PN * (i + N * b) : a + PN * (b + N * i));
        R0[idx] += acc[i];
      }
    }
  };
  dir_body(std::integral_constant<int, 0>{});
  dir_body(std::integral_constant<int, 1>{});
  dir_body(std::integral_constant<int, 2>{});
  __syncthreads();

  // ---- A u (and the Chebyshev update of the node: cheby_update_kernel, same roundings; u is an INPUT of this kernel -- the
  // neighbours read it --, so the new iterate goes to a second vector)
  if (on) {
    constexpr int NL = N;   // PL = N^2: one k-plane per pass
    const int ij = (te % N) + PN * (te / N);
    if constexpr (FUSE) {
      constexpr int BT = 4;   // planes per batch: the smoother vectors of a batch are requested before the first is used
#pragma unroll
      for (int q0 = 0; q0 < NL; q0 += BT) {
        double rh[BT], pp[BT], uu[BT];
#pragma unroll
        for (int q = 0; q < BT; ++q) {
          if (q0 + q < NL) {
            const size_t o = (size_t)ns + te + PL * (q0 + q);
            rh[q] = cf.rhs[o];
            pp[q] = cf.p[o];
            uu[q] = u[o];
          }
        }
#pragma unroll
        for (int q = 0; q < BT; ++q) {
          if (q0 + q < NL) {
            const size_t o = (size_t)ns + te + PL * (q0 + q);
            const double au = R0[ij + PN * N * (q0 + q)];
            if (!cf.skip_Au_store) Au[o] = au;
            const double res = __dadd_rn(rh[q], __dmul_rn(-1.0, au));
            const double ri = __dmul_rn(cf.alpha, res);
            const double pi = __dadd_rn(__dmul_rn(cf.beta, pp[q]), ri);
            if (cf.r) cf.r[o] = ri;
            cf.p[o] = pi;
            cf.u_out[o] = __dadd_rn(uu[q], pi);
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NL; ++q) Au[(size_t)ns + te + PL * q] = R0[ij + PN * N * q];
    }
  }
}

}  // namespace

#ifdef D4EST_HIP_MW_ONLY   /* development builds: one size */
#define D4EST_HIP_DIRECT_MW_SIZES(X) X(D4EST_HIP_MW_ONLY)
#else
#define D4EST_HIP_DIRECT_MW_SIZES(X) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#endif

bool direct_mw_built(int N, int NQ) {
  if (N != NQ) return false;
#define X(N_) if (N == N_) return true;
  D4EST_HIP_DIRECT_MW_SIZES(X)
#undef X
  return false;
}

template <typename K>
static void mw_set_lds_limit(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// vmode: 0 the face terms only (Au += ...), 1 / 2 the whole operator with the streamed / affine metric
void launch_direct_mw(d4est_hip_plan* plan, DirectHost* dh, const double* u, const double* ghost_trace, double* Au, const DirectFuse* cf,
                      const double* robin_c, const double* robin_r, int vmode, const DirectVol& vol, int n, int chunk) {
  const DirectFuse cfv = cf ? *cf : DirectFuse{};
  bool done = false;
#define D4EST_HIP_MW_GO(N_, FUSE_, VOL_)                                                                                          \
  do {                                                                                                                             \
    mw_set_lds_limit(operator_mw_kernel<N_, FUSE_, VOL_>, MwCfg<N_>::LDS_BYTES);                                                  \
    hipLaunchKernelGGL((operator_mw_kernel<N_, FUSE_, VOL_>), dim3(n), dim3(MwCfg<N_>::THREADS), MwCfg<N_>::LDS_BYTES, plan->stream, \
                       u, ghost_trace, Au, dh->d_sides, dh->d_ghost_off, dh->d_ops, plan->d_face_geom, plan->d_bndry, robin_c,     \
                       robin_r, n, dh->ns0, dh->ns_stride, chunk, cfv, vol, dh->d_list);                                           \
  } while (0)
#define X(N_)                                                                                       \
  if (!done && dh->N == N_) {                                                                       \
    if (vmode == 1) { if (cf) D4EST_HIP_MW_GO(N_, true, 1); else D4EST_HIP_MW_GO(N_, false, 1); }   \
    else if (vmode == 2) { if (cf) D4EST_HIP_MW_GO(N_, true, 2); else D4EST_HIP_MW_GO(N_, false, 2); } \
    else { if (cf) D4EST_HIP_MW_GO(N_, true, 0); else D4EST_HIP_MW_GO(N_, false, 0); }              \
    done = true;                                                                                    \
  }
  D4EST_HIP_DIRECT_MW_SIZES(X)
#undef X
#undef D4EST_HIP_MW_GO
  if (!done) D4EST_HIP_ABORT("multi-wave direct kernel: no instance for N = %d", dh->N);
  HIP_CHECK(hipGetLastError());
}

}  // namespace d4est_hip
