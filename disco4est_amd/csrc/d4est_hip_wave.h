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
template <int C>
__device__ __forceinline__ void eo_pre(const double* x, double* xe, double* xo) {
#pragma unroll
  for (int c = 0; c < C / 2; ++c) {
    xe[c] = x[c] + x[C - 1 - c];
    xo[c] = x[c] - x[C - 1 - c];
  }
}
template <int R>
__device__ __forceinline__ void eo_post(const double* ab, double* y) {
#pragma unroll
  for (int r = 0; r < R / 2; ++r) {
    y[r] = ab[r] + ab[R / 2 + r];
    y[R - 1 - r] = ab[r] - ab[R / 2 + r];
  }
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

// two operators at once, one EO row of each per step: yA (+)= sum_c rowA_c * (xfA[c] | xsA[c]), same for B.  HC = C/2 rows.
template <int HC, int R, bool ACCA, bool ACCB>
__device__ __forceinline__ void contract_pair_eo(const double* __restrict__ opA, const double* xfA, const double* xsA, double* yA,
                                                 const double* __restrict__ opB, const double* xfB, const double* xsB, double* yB) {
  constexpr int HR = R / 2;
  double ca[R], cb[R], na[R], nb[R];
  {
    sdouble_ptr ra = launder(opA), rb = launder(opB);
#pragma unroll
    for (int o = 0; o < R; ++o) { ca[o] = ra[o]; cb[o] = rb[o]; }
  }
#pragma unroll
  for (int i = 0; i < HC; ++i) {
    if (i + 1 < HC) {
      sdouble_ptr ra, rb;
      launder2_after(opA + (i + 1) * R, opB + (i + 1) * R, ca[0], cb[0], ra, rb);
#pragma unroll
      for (int o = 0; o < R; ++o) { na[o] = ra[o]; nb[o] = rb[o]; }
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
  constexpr int HR = R / 2;
  constexpr int STEPS = (HC + 1) / 2;
  double c0[R], c1[R], n0[R], n1[R];
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
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int i0 = 2 * st, i1 = 2 * st + 1;
    if (i0 + 2 < HC) {
      sdouble_ptr r0, r1;
      launder2_after(op + (i0 + 2) * R, op + ((i1 + 2 < HC) ? (i1 + 2) : (i0 + 2)) * R, c0[0], (i1 < HC) ? c1[0] : c0[0], r0, r1);
#pragma unroll
      for (int o = 0; o < R; ++o) n0[o] = r0[o];
      if (i1 + 2 < HC) {
#pragma unroll
        for (int o = 0; o < R; ++o) n1[o] = r1[o];
      }
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
  constexpr int HR = R / 2;
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

}  // namespace d4est_hip
