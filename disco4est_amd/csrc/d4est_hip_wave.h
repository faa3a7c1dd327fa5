// Device helpers shared by the single-wavefront kernels (volume, faces, direct faces).
#pragma once
#include <hip/hip_runtime.h>

namespace d4est_hip {

// The 1-D operator entries are wave-uniform: they are fetched with scalar loads
// (s_load) and feed v_fma_f64 as SGPR operands.  `launder` hides the pointer's
// provenance from the optimiser once per operator ROW, so identical loads are not
// CSE'd across stages (which would keep 128+ doubles live in SGPRs and spill them
// through v_readlane); each row (<= 16 doubles) lives only for its own FMAs.
// The laundered pointer is re-typed to the constant address space (4) so the
// backend keeps emitting s_load_dwordx* for it.
typedef const double __attribute__((address_space(4))) * sdouble_ptr;
__device__ __forceinline__ sdouble_ptr launder(const double* p) {
  unsigned long long v = reinterpret_cast<unsigned long long>(p);
  asm volatile("" : "+s"(v));
  return (sdouble_ptr)v;
}

// y = op x with op (NO x NI) passed TRANSPOSED: opT is NI x NO row-major.  Loop order i-outer so that one scalar
// row opT[i][0..NO) feeds NO INDEPENDENT FMA chains (a dependent chain per output would serialise on the
// FP64 FMA latency: measured 39 % issue-stall cycles with the o-outer form, profiles/r01_c_*).
template <int NI, int NO>
__device__ __forceinline__ void contract_n(const double* __restrict__ opT, const double* x, double* y) {
  // outputs in chunks of <= 8: one scalar row chunk is <= 16 SGPRs (longer rows spill SGPRs through v_readlane)
#pragma unroll
  for (int o0 = 0; o0 < NO; o0 += 8) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      sdouble_ptr row = launder(opT + i * NO + o0);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        if (o0 + o < NO) y[o0 + o] = (i == 0) ? row[o] * x[0] : fma(row[o], x[i], y[o0 + o]);
      }
    }
  }
}

// y (+)= op^T x, op is NI x NO row-major.  Loop order i-outer so that each scalar
// row op[i][0..NO) is consumed by NO independent FMA chains.
template <int NI, int NO, bool ACC>
__device__ __forceinline__ void contract_t(const double* __restrict__ op, const double* x, double* y) {
#pragma unroll
  for (int o0 = 0; o0 < NO; o0 += 8) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      sdouble_ptr row = launder(op + i * NO + o0);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        if (o0 + o < NO) y[o0 + o] = (i == 0 && !ACC) ? row[o] * x[0] : fma(row[o], x[i], y[o0 + o]);
      }
    }
  }
}

// ---- even-odd form of the contractions (see stiffness_wave_eo_kernel in d4est_hip_volume.hip for the derivation and table layout)
// Sizes of either parity: xe / xo have (C + 1) / 2 entries, for odd C the last one of both is the middle input; the accumulated
// (a | b) row splits at (R + 1) / 2, for odd R its entry R / 2 is the middle output (Tables1D::eo_table).
template <int C>
__device__ __forceinline__ void eo_pre(const double* x, double* xe, double* xo) {
#pragma unroll
  for (int c = 0; c < C / 2; ++c) {
    xe[c] = x[c] + x[C - 1 - c];
    xo[c] = x[c] - x[C - 1 - c];
  }
  if constexpr (C % 2 != 0) {
    xe[C / 2] = x[C / 2];
    xo[C / 2] = x[C / 2];
  }
}
template <int R>
__device__ __forceinline__ void eo_post(const double* ab, double* y) {
  constexpr int SP = (R + 1) / 2;
#pragma unroll
  for (int r = 0; r < R / 2; ++r) {
    y[r] = ab[r] + ab[SP + r];
    y[R - 1 - r] = ab[r] - ab[SP + r];
  }
  if constexpr (R % 2 != 0) y[R / 2] = ab[R / 2];
}

__device__ __forceinline__ void sgpr_touch(double v) { asm volatile("" ::"s"(v)); }
// launder a pointer in an asm that also CONSUMES two already-requested scalars: the loads through the returned
// pointer cannot be issued before the s_waitcnt that makes d0/d1 available.
__device__ __forceinline__ void launder2_after(const double* pa, const double* pb, double d0, double d1, sdouble_ptr& ra,
                                               sdouble_ptr& rb) {
  unsigned long long va = reinterpret_cast<unsigned long long>(pa), vb = reinterpret_cast<unsigned long long>(pb);
  asm volatile("" : "+s"(va), "+s"(vb) : "s"(d0), "s"(d1));
  ra = (sdouble_ptr)va;
  rb = (sdouble_ptr)vb;
}

#ifndef D4EST_HIP_EO_IMM_ROWS
#define D4EST_HIP_EO_IMM_ROWS 1
#endif
// two operators at once, one EO row of each per step: yA (+)= sum_c rowA_c * (xfA[c] | xsA[c]), same for B.  HC = C/2 rows.
template <int HC, int R, bool ACCA, bool ACCB>
__device__ __forceinline__ void contract_pair_eo(const double* __restrict__ opA, const double* xfA, const double* xsA, double* yA,
                                                 const double* __restrict__ opB, const double* xfB, const double* xsB, double* yB) {
  constexpr int HR = (R + 1) / 2;   // outputs [0, HR) take the first input combination, [HR, R) the second
  double ca[R], cb[R], na[R], nb[R];
#if D4EST_HIP_EO_IMM_ROWS   /* the rows through two laundered BASE pointers with immediate offsets (no scalar address arithmetic per row) */
  unsigned long long bA = reinterpret_cast<unsigned long long>(opA), bB = reinterpret_cast<unsigned long long>(opB);
  asm volatile("" : "+s"(bA), "+s"(bB));
  {
    sdouble_ptr ra = (sdouble_ptr)bA, rb = (sdouble_ptr)bB;
#pragma unroll
    for (int o = 0; o < R; ++o) { ca[o] = ra[o]; cb[o] = rb[o]; }
  }
#else
  {
    sdouble_ptr ra = launder(opA), rb = launder(opB);
#pragma unroll
    for (int o = 0; o < R; ++o) { ca[o] = ra[o]; cb[o] = rb[o]; }
  }
#endif
#pragma unroll
  for (int i = 0; i < HC; ++i) {
    if (i + 1 < HC) {
#if D4EST_HIP_EO_IMM_ROWS
      asm volatile("" : "+s"(bA), "+s"(bB) : "s"(ca[0]), "s"(cb[0]));
      sdouble_ptr ra = (sdouble_ptr)bA, rb = (sdouble_ptr)bB;
#pragma unroll
      for (int o = 0; o < R; ++o) { na[o] = ra[(i + 1) * R + o]; nb[o] = rb[(i + 1) * R + o]; }
#else
      sdouble_ptr ra, rb;
      launder2_after(opA + (i + 1) * R, opB + (i + 1) * R, ca[0], cb[0], ra, rb);
#pragma unroll
      for (int o = 0; o < R; ++o) { na[o] = ra[o]; nb[o] = rb[o]; }
#endif
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const double xa = (o < HR) ? xfA[i] : xsA[i], xb = (o < HR) ? xfB[i] : xsB[i];
      yA[o] = (i == 0 && !ACCA) ? ca[o] * xa : fma(ca[o], xa, yA[o]);
      yB[o] = (i == 0 && !ACCB) ? cb[o] * xb : fma(cb[o], xb, yB[o]);
    }
    if (i + 1 < HC) {
#pragma unroll
      for (int o = 0; o < R; ++o) { ca[o] = na[o]; cb[o] = nb[o]; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one operator, two EO rows per step
template <int HC, int R, bool ACC>
__device__ __forceinline__ void contract_single_eo(const double* __restrict__ op, const double* xf, const double* xs, double* y) {
  constexpr int HR = (R + 1) / 2;   // outputs [0, HR) take the first input combination, [HR, R) the second
  constexpr int STEPS = (HC + 1) / 2;
  double c0[R], c1[R], n0[R], n1[R];
#if D4EST_HIP_EO_IMM_ROWS
  unsigned long long bs = reinterpret_cast<unsigned long long>(op);
  asm volatile("" : "+s"(bs));
  {
    sdouble_ptr r0 = (sdouble_ptr)bs;
#pragma unroll
    for (int o = 0; o < R; ++o) c0[o] = r0[o];
    if (HC > 1) {
#pragma unroll
      for (int o = 0; o < R; ++o) c1[o] = r0[R + o];
    }
  }
#else
  {
    sdouble_ptr r0 = launder(op);
#pragma unroll
    for (int o = 0; o < R; ++o) c0[o] = r0[o];
    if (HC > 1) {
      sdouble_ptr r1 = launder(op + R);
#pragma unroll
      for (int o = 0; o < R; ++o) c1[o] = r1[o];
    }
  }
#endif
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int i0 = 2 * st, i1 = 2 * st + 1;
    if (i0 + 2 < HC) {
#if D4EST_HIP_EO_IMM_ROWS
      asm volatile("" : "+s"(bs) : "s"(c0[0]), "s"((i1 < HC) ? c1[0] : c0[0]));
      sdouble_ptr r0 = (sdouble_ptr)bs;
#pragma unroll
      for (int o = 0; o < R; ++o) n0[o] = r0[(i0 + 2) * R + o];
      if (i1 + 2 < HC) {
#pragma unroll
        for (int o = 0; o < R; ++o) n1[o] = r0[(i1 + 2) * R + o];
      }
#else
      sdouble_ptr r0, r1;
      launder2_after(op + (i0 + 2) * R, op + ((i1 + 2 < HC) ? (i1 + 2) : (i0 + 2)) * R, c0[0], (i1 < HC) ? c1[0] : c0[0], r0, r1);
#pragma unroll
      for (int o = 0; o < R; ++o) n0[o] = r0[o];
      if (i1 + 2 < HC) {
#pragma unroll
        for (int o = 0; o < R; ++o) n1[o] = r1[o];
      }
#endif
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const double x0 = (o < HR) ? xf[i0] : xs[i0];
      y[o] = (i0 == 0 && !ACC) ? c0[o] * x0 : fma(c0[o], x0, y[o]);
      if (i1 < HC) {
        const double x1 = (o < HR) ? xf[i1] : xs[i1];
        y[o] = fma(c1[o], x1, y[o]);
      }
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      if (i0 + 2 < HC) c0[o] = n0[o];
      if (i1 + 2 < HC) c1[o] = n1[o];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one operator, ONE EO row per step, the next row requested while the current one is consumed: 2 R SGPR pairs in flight (half of
// contract_single_eo), for kernels that are short of SGPRs
template <int HC, int R, bool ACC>
__device__ __forceinline__ void contract_rows_eo(const double* __restrict__ op, const double* xf, const double* xs, double* y) {
  constexpr int HR = (R + 1) / 2;   // outputs [0, HR) take the first input combination, [HR, R) the second
  double c0[R], n0[R];
  {
    sdouble_ptr r0 = launder(op);
#pragma unroll
    for (int o = 0; o < R; ++o) c0[o] = r0[o];
  }
#pragma unroll
  for (int i = 0; i < HC; ++i) {
    if (i + 1 < HC) {
      sdouble_ptr r0, r1;
      launder2_after(op + (i + 1) * R, op + (i + 1) * R, c0[0], c0[0], r0, r1);
#pragma unroll
      for (int o = 0; o < R; ++o) n0[o] = r0[o];
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const double x0 = (o < HR) ? xf[i] : xs[i];
      y[o] = (i == 0 && !ACC) ? c0[o] * x0 : fma(c0[o], x0, y[o]);
    }
    if (i + 1 < HC) {
#pragma unroll
      for (int o = 0; o < R; ++o) c0[o] = n0[o];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one operator, ONE full EO row per step through ONE laundered base pointer: the row loads carry immediate offsets (no scalar
// address arithmetic per row) and a row of R doubles is one or two wide scalar loads -- per R FMAs 2-3 scalar instructions instead
// of the 9-10 of the half-row forms above (a scalar instruction costs a wavefront the same issue time as a vector one).  Each step
// re-launders the base in an asm that consumes the current row, so the next row's load cannot be hoisted above the wait for this one.
template <int HC, int R, bool ACC>
__device__ __forceinline__ void contract_rows_eo_imm(const double* __restrict__ op, const double* xf, const double* xs, double* y) {
  constexpr int HR = (R + 1) / 2;   // outputs [0, HR) take the first input combination, [HR, R) the second
  double c0[R], n0[R];
  unsigned long long base = reinterpret_cast<unsigned long long>(op);
  asm volatile("" : "+s"(base));
  {
    sdouble_ptr r0 = (sdouble_ptr)base;
#pragma unroll
    for (int o = 0; o < R; ++o) c0[o] = r0[o];
  }
#pragma unroll
  for (int i = 0; i < HC; ++i) {
    if (i + 1 < HC) {
      asm volatile("" : "+s"(base) : "s"(c0[0]));
      sdouble_ptr r0 = (sdouble_ptr)base;
#pragma unroll
      for (int o = 0; o < R; ++o) n0[o] = r0[(i + 1) * R + o];
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const double x0 = (o < HR) ? xf[i] : xs[i];
      y[o] = (i == 0 && !ACC) ? c0[o] * x0 : fma(c0[o], x0, y[o]);
    }
    if (i + 1 < HC) {
#pragma unroll
      for (int o = 0; o < R; ++o) c0[o] = n0[o];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LDS read that the backend must not pair into ds_read2_b64: on gfx950 a ds_read2_b64 takes 8 LDS cycles (banks mod 32), two
// ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS table); volatile accesses are never combined
// Stream mode (plan->stream_mode, d4est_hip_internal.h): on plans whose working set does not fit the 256 MB Infinity Cache the data an
// apply touches exactly once -- the metric and the mortar factors on the way in, A u on the way out -- moves with the non-temporal hint
// (global_load / global_store ... nt), so that it neither displaces u from the XCD's L2 (the neighbours' lines of the whole-operator
// kernels come from there) nor allocates cache lines it will not use: level 5, p = 7 stiffness 191 -> 156 us, config 3's full operator
// +3 %.  On plans that do fit (config 2: 134 MB) the same hint costs 2 ... 19 %, because repeated applies are served from that cache:
// hence two instantiations (template parameter NT, ld_sel / store_element_image) chosen at launch.  A run-time branch around each
// batch of loads (with_ld: kept for the matrix-core kernel, whose loads sit outside its pipelined loops) cost the multi-wave kernels
// 4 % with the switch off -- the branches cut their software-pipelined quadrature loops into basic blocks.
struct LdPlain {
  __device__ __forceinline__ double operator()(const double* p) const { return *p; }
};
struct LdStream {
  __device__ __forceinline__ double operator()(const double* p) const { return __builtin_nontemporal_load(p); }
};
// (the two empty asm statements keep the optimiser from merging the branches back into one batch of plain loads: it treats loads that
// differ only in the hint as identical and hoists / sinks them out of the if, dropping the hint)
#define D4EST_HIP_STREAM_FENCE() asm volatile("; stream mode" ::: "memory")
template <class F>
__device__ __forceinline__ void with_ld(bool nt, F&& f) {
  if (nt) {
    D4EST_HIP_STREAM_FENCE();
    f(LdStream{});
    D4EST_HIP_STREAM_FENCE();
  } else {
    f(LdPlain{});
  }
}
template <bool NT>
__device__ __forceinline__ double ld_sel(const double* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

__device__ __forceinline__ double lds_ld(const double* p) {
  typedef const volatile double __attribute__((address_space(3))) * lds_cvptr;
  return *(lds_cvptr)p;
}

// An element's nodal values <-> its LDS image [i + PN (j + N k)], one coalesced pass by the PL lanes `te` of the element.  Written with
// a compile-time trip count: a `for (idx = te; idx < N3; idx += PL)` loop is not unrolled (its trip count depends on te), and the
// rolled loop waits for each global load before it requests the next -- N3 / PL serialised memory round trips per element (12 at
// p = 11).  Here every load is in flight before the first LDS write.
template <int N, int PL, int PN>
__device__ __forceinline__ void load_element_image(double* R, const double* __restrict__ src, int te) {
  constexpr int N3 = N * N * N, NL = (N3 + PL - 1) / PL;
  double v[NL];
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    const int idx = te + PL * q;
    v[q] = (N3 % PL == 0 || idx < N3) ? src[idx] : 0.0;
  }
  if constexpr (PL == N * N) {   // one k-plane per pass: the lane's (i, j) is the same in every plane (no division per load)
    const int ij = (te % N) + PN * (te / N);
#pragma unroll
    for (int q = 0; q < NL; ++q) R[ij + PN * N * q] = v[q];
    return;
  }
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    const int idx = te + PL * q;
    const int i = idx % N, j = (idx / N) % N, k = idx / (N * N);
    if (N3 % PL == 0 || idx < N3) R[i + PN * (j + N * k)] = v[q];
  }
}
template <int N, int PL, int PN, bool NT = false>
__device__ __forceinline__ void store_element_image(double* __restrict__ dst, const double* R, int te) {
  constexpr int N3 = N * N * N, NL = (N3 + PL - 1) / PL;
  auto put = [](double* p_, double v_) {   // NT: stream mode, see above
    if constexpr (NT) __builtin_nontemporal_store(v_, p_);
    else *p_ = v_;
  };
  if constexpr (PL == N * N) {
    const int ij = (te % N) + PN * (te / N);
#pragma unroll
    for (int q = 0; q < NL; ++q) put(&dst[te + PL * q], R[ij + PN * N * q]);
    return;
  }
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    const int idx = te + PL * q;
    const int i = idx % N, j = (idx / N) % N, k = idx / (N * N);
    if (N3 % PL == 0 || idx < N3) put(&dst[idx], R[i + PN * (j + N * k)]);
  }
}

__host__ __device__ inline int face_fix(int f, int N) { return (f & 1) ? (N - 1) : 0; }

__device__ inline int reorder_index(int code, int deg, int a, int b) {
  // out(a,b) = in(a2,b2) for out = transpose?(flip1?(flip0?(in)))  (dGMath/d4est_operators.c:2044-2081)
  int a1 = (code & 4) ? b : a, b1 = (code & 4) ? a : b;
  if (code & 2) b1 = deg - b1;
  if (code & 1) a1 = deg - a1;
  return a1 + (deg + 1) * b1;
}

// ordering point for LDS traffic that stays inside ONE wavefront (a wave runs in lockstep: no workgroup barrier needed)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// software-pipelined plain row contractions (two rows in flight; see the stiffness kernels in d4est_hip_volume.hip)
template <int NI, int NO, bool ACC, int LD = NO>
__device__ __forceinline__ void contract_single(const double* __restrict__ op, const double* x, double* y);

// yA (+)= sum_i rowA_i * xA[i], yB (+)= sum_i rowB_i * xB[i]; row_i = NO consecutive doubles at op + i*NO
// (op = the operator TRANSPOSED for y = op x, or the operator itself for y = op^T x)
template <int NI, int NO, bool ACCA, bool ACCB>
__device__ __forceinline__ void contract_pair(const double* __restrict__ opA, const double* xA, double* yA,
                                              const double* __restrict__ opB, const double* xB, double* yB) {
  double ca[NO], cb[NO], na[NO], nb[NO];
  {
    sdouble_ptr ra = launder(opA), rb = launder(opB);
#pragma unroll
    for (int o = 0; o < NO; ++o) { ca[o] = ra[o]; cb[o] = rb[o]; }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (i + 1 < NI) {
      sdouble_ptr ra, rb;
      launder2_after(opA + (i + 1) * NO, opB + (i + 1) * NO, ca[0], cb[0], ra, rb);
#pragma unroll
      for (int o = 0; o < NO; ++o) { na[o] = ra[o]; nb[o] = rb[o]; }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      yA[o] = (i == 0 && !ACCA) ? ca[o] * xA[0] : fma(ca[o], xA[i], yA[o]);
      yB[o] = (i == 0 && !ACCB) ? cb[o] * xB[0] : fma(cb[o], xB[i], yB[o]);
    }
    if (i + 1 < NI) {
#pragma unroll
      for (int o = 0; o < NO; ++o) { ca[o] = na[o]; cb[o] = nb[o]; }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the next step's wait/launder asm behind this step's FMAs
  }
}

// y (+)= sum_i row_i * x[i], two rows per step
template <int NI, int NO, bool ACC, int LD>
__device__ __forceinline__ void contract_single(const double* __restrict__ op, const double* x, double* y) {
  constexpr int STEPS = (NI + 1) / 2;
  double c0[NO], c1[NO], n0[NO], n1[NO];
  {
    sdouble_ptr r0 = launder(op);
#pragma unroll
    for (int o = 0; o < NO; ++o) c0[o] = r0[o];
    if (NI > 1) {
      sdouble_ptr r1 = launder(op + LD);
#pragma unroll
      for (int o = 0; o < NO; ++o) c1[o] = r1[o];
    }
  }
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int i0 = 2 * st, i1 = 2 * st + 1;
    if (i0 + 2 < NI) {
      sdouble_ptr r0, r1;
      launder2_after(op + (i0 + 2) * LD, op + ((i1 + 2 < NI) ? (i1 + 2) : (i0 + 2)) * LD, c0[0], (i1 < NI) ? c1[0] : c0[0], r0, r1);
#pragma unroll
      for (int o = 0; o < NO; ++o) n0[o] = r0[o];
      if (i1 + 2 < NI) {
#pragma unroll
        for (int o = 0; o < NO; ++o) n1[o] = r1[o];
      }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      y[o] = (i0 == 0 && !ACC) ? c0[o] * x[0] : fma(c0[o], x[i0], y[o]);
      if (i1 < NI) y[o] = fma(c1[o], x[i1], y[o]);
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      if (i0 + 2 < NI) c0[o] = n0[o];
      if (i1 + 2 < NI) c1[o] = n1[o];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one element per NQ*NQ lanes of a wavefront: LDS image of the single-wavefront volume kernels
template <int N, int NQ>
struct WaveCfg {
  static constexpr int PL = NQ * NQ;
  static constexpr int EPB = (PL <= 64) ? 64 / PL : 1;
  static constexpr int THREADS = (PL <= 64) ? 64 : ((PL + 63) / 64) * 64;  // > 64: one element per multi-wave workgroup
  static constexpr int PN = N | 1, PQ = NQ | 1;
  static constexpr int FS = NQ * NQ * PQ;
  static constexpr int LDS_PER_ELEM = 2 * FS;
  static constexpr size_t LDS_BYTES = (size_t)EPB * LDS_PER_ELEM * sizeof(double);
};

// ---------------------------------------------------------------------------
// The sum-factorised stiffness apply of ONE element by the NQ*NQ lanes `te` of a wavefront in the even-odd form (derivation, table
// layout and the stage list: stiffness_wave_eo_kernel in d4est_hip_volume.hip, which is this function plus the element's loads and
// stores).  On entry R0 holds u_e as [i + PN (j + N k)]; on exit R0 holds (A u)_e in the same layout.  R0, R1: FS doubles each.
// WG_SYNC: the hand-offs through LDS use __syncthreads() (the stand-alone kernel: one wave per workgroup) or the wave-level fence
// (kernels whose workgroup holds several independent waves).
// ---------------------------------------------------------------------------
#ifndef D4EST_HIP_WAVE_EO_DEPTH
#define D4EST_HIP_WAVE_EO_DEPTH 3
#endif
#ifndef D4EST_HIP_WAVE_EO_EARLY
#define D4EST_HIP_WAVE_EO_EARLY 1
#endif
// MASS: the zeroth-order term of a linearised nonlinear problem rides along: + V^T [ w J c (V u) ] (d4est_quadrature_apply_fofufofvlilj,
// src/Quadrature/d4est_quadrature.c:593-774, with the coefficient c = f(x, u0) handed over at the quadrature nodes) -- V u = B_t B_s B_r u
// costs one more t-contraction of the line B_s B_r u the gradient already forms, the weighted value one more transposed t-contraction
// summed into the G_t^T term; cq = w J c at the quadrature nodes (pre-combined when the coefficient is set: one stream of 8 B per node).
template <int N, int NQ, bool AFF, bool WG_SYNC, bool MASS = false, bool NT = false>
__device__ __forceinline__ void stiffness_wave_eo_element(double* R0, double* R1, const double* __restrict__ metric, int qs, int ei,
                                                          bool active, int a, int b, const double* __restrict__ EBf,
                                                          const double* __restrict__ EGf, const double* __restrict__ EBb,
                                                          const double* __restrict__ EGb, const double* __restrict__ affine,
                                                          const double* __restrict__ wq, const double* __restrict__ cq = nullptr) {
  using C = WaveCfg<N, NQ>;
  constexpr int PN = C::PN, PQ = C::PQ;
  constexpr int NQ3 = NQ * NQ * NQ;
  constexpr int HN = (N + 1) / 2, HQ = (NQ + 1) / 2;   // rows of the even-odd tables = entries of xe / xo
  auto SYNC = [] {
    if constexpr (WG_SYNC) __syncthreads();
    else wave_lds_fence();
  };

  // ---- S1: thread (j=a, k=b): B_r u, G_r u
  {
    double x[N], xe[HN], xo[HN], br[NQ], gr[NQ];
    const bool on = active && a < N && b < N;
    if (on) {
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = lds_ld(&R0[i + PN * (a + N * b)]);
      eo_pre<N>(x, xe, xo);
      contract_pair_eo<HN, NQ, false, false>(EBf, xe, xo, br, EGf, xo, xe, gr);
    }
    SYNC();
    if (on) {
      double y[NQ];
      eo_post<NQ>(br, y);
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) R0[a + PN * (iq + NQ * b)] = y[iq];
      eo_post<NQ>(gr, y);
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) R1[a + PN * (iq + NQ * b)] = y[iq];
    }
  }
  SYNC();

  // ---- S2 (thread (iq=a, k=b)) and S3 (thread (iq=a, jq=b))
  double gr[NQ], gs[NQ], gt[NQ];
  // (MASS: V u at the thread's quadrature nodes waits in R1, in the thread's own slots [kq][te] -- R1 is idle between the last read of
  // S3 and the first write of S5 -- and is weighted with w J c where the transposed t-contraction takes it: no registers across stages)
  // General path of the stand-alone kernel: the 6 NQ metric values of the thread are requested MD quadrature planes ahead of their
  // use, the first ME of them before the last forward contraction; the scheduling barriers pin "request plane kq + MD, multiply
  // plane kq" (the compiler's own order drains the queue between small groups: 6-7 dependent memory round trips at p = 7).
  constexpr bool kPipe = !AFF && WG_SYNC && (D4EST_HIP_WAVE_EO_DEPTH > 0);
  constexpr int MD = (D4EST_HIP_WAVE_EO_DEPTH < NQ) ? D4EST_HIP_WAVE_EO_DEPTH : NQ;
  constexpr int ME = (D4EST_HIP_WAVE_EO_EARLY < MD) ? D4EST_HIP_WAVE_EO_EARLY : MD;
  double mw[kPipe ? NQ : 1][6];
  {
    double t3[NQ];
    const bool on2 = active && b < N;
    {
      double x1[N], x2[N], x1e[HN], x1o[HN], x2e[HN], x2o[HN], t1[NQ], t2[NQ];
      if (on2) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          x1[j] = lds_ld(&R0[j + PN * (a + NQ * b)]);  // B_r u
          x2[j] = lds_ld(&R1[j + PN * (a + NQ * b)]);  // G_r u
        }
        eo_pre<N>(x1, x1e, x1o);
        eo_pre<N>(x2, x2e, x2o);
        contract_pair_eo<HN, NQ, false, false>(EBf, x2e, x2o, t1, EGf, x1o, x1e, t2);  // B_s G_r u | G_s B_r u
        contract_single_eo<HN, NQ, false>(EBf, x1e, x1o, t3);                           // B_s B_r u
      }
      SYNC();
      if (on2) {
        double y[NQ];
        eo_post<NQ>(t1, y);
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq) R0[b + PN * (a + NQ * jq)] = y[jq];  // [jq][iq][k]
        eo_post<NQ>(t2, y);
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq) R1[b + PN * (a + NQ * jq)] = y[jq];
      }
    }
    SYNC();
    if (active) {
      double y1[N], y2[N], y1e[HN], y1o[HN], y2e[HN], y2o[HN], ge[NQ], he[NQ];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        y1[k] = lds_ld(&R0[k + PN * (a + NQ * b)]);
        y2[k] = lds_ld(&R1[k + PN * (a + NQ * b)]);
      }
      eo_pre<N>(y1, y1e, y1o);
      eo_pre<N>(y2, y2e, y2o);
      contract_pair_eo<HN, NQ, false, false>(EBf, y1e, y1o, ge, EBf, y2e, y2o, he);
      eo_post<NQ>(ge, gr);
      eo_post<NQ>(he, gs);
    }
    SYNC();
    if (on2) {
      double y[NQ];
      eo_post<NQ>(t3, y);
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) R0[b + PN * (a + NQ * jq)] = y[jq];
    }
    SYNC();
    if (active) {
      double y3[N], y3e[HN], y3o[HN], ge[NQ];
#pragma unroll
      for (int k = 0; k < N; ++k) y3[k] = lds_ld(&R0[k + PN * (a + NQ * b)]);
      if constexpr (kPipe) {
        const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
        for (int kq = 0; kq < ME; ++kq)
#pragma unroll
          for (int c = 0; c < 6; ++c) mw[kq][c] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * kq]);
        __builtin_amdgcn_sched_barrier(0);
      }
      eo_pre<N>(y3, y3e, y3o);
      if constexpr (MASS) {
        double he[NQ], vm[NQ];
        contract_pair_eo<HN, NQ, false, false>(EGf, y3o, y3e, ge, EBf, y3e, y3o, he);
        eo_post<NQ>(he, vm);
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) R1[kq * C::PL + (a + NQ * b)] = vm[kq];
      } else {
        contract_single_eo<HN, NQ, false>(EGf, y3o, y3e, ge);
      }
      eo_post<NQ>(ge, gt);
    }
  }

  // ---- quadrature-point stage
  if (active) {
    if constexpr (AFF) {
      const double* __restrict__ c = affine + (size_t)6 * ei;
      const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
      const double wab = wq[b] * wq[a];
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) {
        const double w3 = wq[kq] * wab;
        const double r = w3 * gr[kq], s = w3 * gs[kq], t = w3 * gt[kq];
        gr[kq] = c0 * r + c1 * s + c2 * t;
        gs[kq] = c1 * r + c3 * s + c4 * t;
        gt[kq] = c2 * r + c4 * s + c5 * t;
      }
    } else if constexpr (kPipe) {
      const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
      for (int kq = ME; kq < MD; ++kq)
#pragma unroll
        for (int c = 0; c < 6; ++c) mw[kq][c] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * kq]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) {
        if (kq + MD < NQ) {
#pragma unroll
          for (int c = 0; c < 6; ++c) mw[kq + MD][c] = ld_sel<NT>(&m[c * NQ3 + NQ * NQ * (kq + MD)]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const double r = gr[kq], s = gs[kq], t = gt[kq];
        gr[kq] = mw[kq][0] * r + mw[kq][1] * s + mw[kq][2] * t;
        gs[kq] = mw[kq][1] * r + mw[kq][3] * s + mw[kq][4] * t;
        gt[kq] = mw[kq][2] * r + mw[kq][4] * s + mw[kq][5] * t;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
      for (int kq = 0; kq < NQ; ++kq) {
        const int q = NQ * NQ * kq;
        const double m0 = m[q], m1 = m[NQ3 + q], m2 = m[2 * NQ3 + q], m3 = m[3 * NQ3 + q], m4 = m[4 * NQ3 + q], m5 = m[5 * NQ3 + q];
        const double r = gr[kq], s = gs[kq], t = gt[kq];
        gr[kq] = m0 * r + m1 * s + m2 * t;
        gs[kq] = m1 * r + m3 * s + m4 * t;
        gt[kq] = m2 * r + m4 * s + m5 * t;
      }
    }
  }

  // ---- S5 (registers) / S6 (thread (iq=a, k=b))
  {
    double cc[N], ar[N], bs[N];
    const bool on6 = active && b < N;
    {
      double ca[N], cb[N];
      if (active) {
        double re[HQ], ro[HQ], se[HQ], so[HQ], te_[HQ], to[HQ];
        eo_pre<NQ>(gr, re, ro);
        eo_pre<NQ>(gs, se, so);
        eo_pre<NQ>(gt, te_, to);
        contract_pair_eo<HQ, N, false, false>(EBb, re, ro, ca, EBb, se, so, cb);
        contract_single_eo<HQ, N, false>(EGb, to, te_, cc);
        if constexpr (MASS) {   // summed in (a | b) form with G_t^T(...): both continue through B_s^T B_r^T
          const double* __restrict__ cp = cq + qs + (a + NQ * b);
          double cv[NQ], vm[NQ], me[HQ], mo[HQ];
#pragma unroll
          for (int kq = 0; kq < NQ; ++kq) cv[kq] = cp[NQ * NQ * kq];
#pragma unroll
          for (int kq = 0; kq < NQ; ++kq) vm[kq] = lds_ld(&R1[kq * C::PL + (a + NQ * b)]) * cv[kq];
          eo_pre<NQ>(vm, me, mo);
          contract_single_eo<HQ, N, true>(EBb, me, mo, cc);
        }
      }
      SYNC();
      if (active) {
        double y[N];
        eo_post<N>(ca, y);
#pragma unroll
        for (int k = 0; k < N; ++k) R0[b + PQ * (a + NQ * k)] = y[k];  // [k][iq][jq]
        eo_post<N>(cb, y);
#pragma unroll
        for (int k = 0; k < N; ++k) R1[b + PQ * (a + NQ * k)] = y[k];
      }
    }
    SYNC();
    if (on6) {
      double x[NQ], y[NQ], xe[HQ], xo[HQ], ye[HQ], yo[HQ];
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) {
        x[jq] = lds_ld(&R0[jq + PQ * (a + NQ * b)]);
        y[jq] = lds_ld(&R1[jq + PQ * (a + NQ * b)]);
      }
      eo_pre<NQ>(x, xe, xo);
      eo_pre<NQ>(y, ye, yo);
      contract_pair_eo<HQ, N, false, false>(EBb, xe, xo, ar, EGb, yo, ye, bs);
    }
    SYNC();
    if (active) {
      double y[N];
      eo_post<N>(cc, y);
#pragma unroll
      for (int k = 0; k < N; ++k) R0[b + PQ * (a + NQ * k)] = y[k];
    }
    SYNC();
    if (on6) {
      double z[NQ], ze[HQ], zo[HQ];
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) z[jq] = lds_ld(&R0[jq + PQ * (a + NQ * b)]);
      eo_pre<NQ>(z, ze, zo);
      contract_single_eo<HQ, N, true>(EBb, ze, zo, bs);  // summed in (a | b) form with G_s^T(...)
    }
    SYNC();
    if (on6) {
      double y[N];
      eo_post<N>(ar, y);
#pragma unroll
      for (int j = 0; j < N; ++j) R0[a + PQ * (j + N * b)] = y[j];  // [k][j][iq]
      eo_post<N>(bs, y);
#pragma unroll
      for (int j = 0; j < N; ++j) R1[a + PQ * (j + N * b)] = y[j];
    }
  }
  SYNC();

  // ---- S7: thread (j=a, k=b)
  {
    double o[N], o2[N];
    const bool on = active && a < N && b < N;
    if (on) {
      double x[NQ], y[NQ], xe[HQ], xo[HQ], ye[HQ], yo[HQ];
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        x[iq] = lds_ld(&R0[iq + PQ * (a + N * b)]);
        y[iq] = lds_ld(&R1[iq + PQ * (a + N * b)]);
      }
      eo_pre<NQ>(x, xe, xo);
      eo_pre<NQ>(y, ye, yo);
      contract_pair_eo<HQ, N, false, false>(EGb, xo, xe, o, EBb, ye, yo, o2);
    }
    SYNC();
    if (on) {
      double y[N];
#pragma unroll
      for (int i = 0; i < N; ++i) o[i] += o2[i];
      eo_post<N>(o, y);
#pragma unroll
      for (int i = 0; i < N; ++i) R0[i + PN * (a + N * b)] = y[i];
    }
  }
  SYNC();
}

// ---------------------------------------------------------------------------
// The same apply with PLAIN contractions (any N <= NQ, NQ * NQ <= 64; the odd degrees): stiffness_wave2_kernel's body.  On entry R0
// holds u_e as [i + PN (j + N k)]; on exit R0 holds (A u)_e in the same layout.  Bop / Gop: B, G row-major (NQ x N); BopT / GopT: their
// transposes.
// ---------------------------------------------------------------------------
template <int N, int NQ, bool WG_SYNC>
__device__ __forceinline__ void stiffness_wave2_element(double* R0, double* R1, const double* __restrict__ metric, int qs, bool active,
                                                        int a, int b, const double* __restrict__ Bop, const double* __restrict__ Gop,
                                                        const double* __restrict__ BopT, const double* __restrict__ GopT) {
  using C = WaveCfg<N, NQ>;
  constexpr int PN = C::PN, PQ = C::PQ;
  constexpr int NQ3 = NQ * NQ * NQ;
  auto SYNC = [] {
    if constexpr (WG_SYNC) __syncthreads();
    else wave_lds_fence();
  };

  // ---- S1: thread (j=a, k=b)
  {
    double x[N], br[NQ], gr[NQ];
    const bool on = active && a < N && b < N;
    if (on) {
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = R0[i + PN * (a + N * b)];
      contract_pair<N, NQ, false, false>(BopT, x, br, GopT, x, gr);
    }
    SYNC();
    if (on) {
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        R0[a + PN * (iq + NQ * b)] = br[iq];
        R1[a + PN * (iq + NQ * b)] = gr[iq];
      }
    }
  }
  SYNC();

  // ---- S2 (thread (iq=a, k=b)) and S3 (thread (iq=a, jq=b))
  double gr[NQ], gs[NQ], gt[NQ];
  {
    double x1[N], x2[N], t1[NQ], t2[NQ], t3[NQ];
    const bool on2 = active && b < N;
    if (on2) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        x1[j] = R0[j + PN * (a + NQ * b)];  // B_r u
        x2[j] = R1[j + PN * (a + NQ * b)];  // G_r u
      }
      contract_pair<N, NQ, false, false>(BopT, x2, t1, GopT, x1, t2);  // B_s G_r u | G_s B_r u
      contract_single<N, NQ, false>(BopT, x1, t3);                      // B_s B_r u
    }
    SYNC();
    if (on2) {
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) {  // [jq][iq][k]
        R0[b + PN * (a + NQ * jq)] = t1[jq];
        R1[b + PN * (a + NQ * jq)] = t2[jq];
      }
    }
    SYNC();
    if (active) {
      double y1[N], y2[N];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        y1[k] = R0[k + PN * (a + NQ * b)];
        y2[k] = R1[k + PN * (a + NQ * b)];
      }
      contract_pair<N, NQ, false, false>(BopT, y1, gr, BopT, y2, gs);
    }
    SYNC();
    if (on2) {
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) R0[b + PN * (a + NQ * jq)] = t3[jq];
    }
    SYNC();
    if (active) {
      double y3[N];
#pragma unroll
      for (int k = 0; k < N; ++k) y3[k] = R0[k + PN * (a + NQ * b)];
      contract_single<N, NQ, false>(GopT, y3, gt);
    }
  }

  // ---- quadrature-point stage
  if (active) {
    const double* __restrict__ m = metric + (size_t)6 * qs + (a + NQ * b);
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {
      const int q = NQ * NQ * kq;
      const double m0 = m[q], m1 = m[NQ3 + q], m2 = m[2 * NQ3 + q], m3 = m[3 * NQ3 + q], m4 = m[4 * NQ3 + q], m5 = m[5 * NQ3 + q];
      const double r = gr[kq], s = gs[kq], t = gt[kq];
      gr[kq] = m0 * r + m1 * s + m2 * t;
      gs[kq] = m1 * r + m3 * s + m4 * t;
      gt[kq] = m2 * r + m4 * s + m5 * t;
    }
  }

  // ---- S5 (registers) / S6 (thread (iq=a, k=b))
  {
    double ca[N], cb[N], cc[N], ar[N], bs[N];
    const bool on6 = active && b < N;
    if (active) {
      contract_pair<NQ, N, false, false>(Bop, gr, ca, Bop, gs, cb);
      contract_single<NQ, N, false>(Gop, gt, cc);
    }
    SYNC();
    if (active) {
#pragma unroll
      for (int k = 0; k < N; ++k) {  // [k][iq][jq]
        R0[b + PQ * (a + NQ * k)] = ca[k];
        R1[b + PQ * (a + NQ * k)] = cb[k];
      }
    }
    SYNC();
    if (on6) {
      double x[NQ], y[NQ];
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) {
        x[jq] = R0[jq + PQ * (a + NQ * b)];
        y[jq] = R1[jq + PQ * (a + NQ * b)];
      }
      contract_pair<NQ, N, false, false>(Bop, x, ar, Gop, y, bs);
    }
    SYNC();
    if (active) {
#pragma unroll
      for (int k = 0; k < N; ++k) R0[b + PQ * (a + NQ * k)] = cc[k];
    }
    SYNC();
    if (on6) {
      double z[NQ];
#pragma unroll
      for (int jq = 0; jq < NQ; ++jq) z[jq] = R0[jq + PQ * (a + NQ * b)];
      contract_single<NQ, N, true>(Bop, z, bs);
    }
    SYNC();
    if (on6) {
#pragma unroll
      for (int j = 0; j < N; ++j) {  // [k][j][iq]
        R0[a + PQ * (j + N * b)] = ar[j];
        R1[a + PQ * (j + N * b)] = bs[j];
      }
    }
  }
  SYNC();

  // ---- S7: thread (j=a, k=b)
  {
    double x[NQ], y[NQ], o[N], o2[N];
    const bool on = active && a < N && b < N;
    if (on) {
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        x[iq] = R0[iq + PQ * (a + N * b)];
        y[iq] = R1[iq + PQ * (a + N * b)];
      }
      contract_pair<NQ, N, false, false>(Gop, x, o, Bop, y, o2);
    }
    SYNC();
    if (on) {
#pragma unroll
      for (int i = 0; i < N; ++i) R0[i + PN * (a + N * b)] = o[i] + o2[i];
    }
  }
  SYNC();
}

}  // namespace d4est_hip
