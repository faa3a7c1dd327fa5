// Micro-benchmark: fp64 issue rates on gfx950 (v_fma_f64 vs v_mfma_f64_16x16x4_f64, alone and together).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_rates.hip -o gpurun_out/fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: VALU fma, 1: MFMA, 2: both interleaved in one wave
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  double x0 = threadIdx.x, x1 = 1.0, x2 = 2.0, x3 = 3.0, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  d4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 2) {
      x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
      x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
      x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
      x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
    if (MODE == 1 || MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      if (MODE == 1) {
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1] + c2[2] + c3[3];
}

template <int MODE>
void run(const char* name, int waves_per_simd, double flop_per_iter_per_wave) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  double* out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)blocks * 4;
  double tf = flop_per_iter_per_wave * iters * waves / (ms * 1e-3) / 1e12;
  // cycles per iteration per SIMD at 2.4 GHz nominal
  double cyc = ms * 1e-3 * 2.4e9 / iters / waves_per_simd;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  ~%.1f cycles/iter/SIMD(@2.4GHz)\n", name, waves_per_simd, ms, tf, cyc);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("valu fma f64 (16/iter)", w, 16.0 * 64 * 2);
    run<1>("mfma f64 16x16x4 (4/iter)", w, 4.0 * 2 * 16 * 16 * 4);
    run<2>("both (16 fma + 2 mfma)", w, 16.0 * 64 * 2 + 2.0 * 2 * 16 * 16 * 4);
  }
  return 0;
}
