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
