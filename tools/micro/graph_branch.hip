// Micro-benchmark: do parallel branches of a hipGraph overlap small kernels at a lower cost than stream fork / join with events?
// Three latency-bound kernels A, B, C that are independent, then D that needs all three; per "apply":
//   serial : A B C D on one stream
//   events : fork to two side streams with events, join, D                (what the hybrid operator's forked form does)
//   graph  : the event form captured into a graph (parallel branches), replayed
// build: hipcc --offload-arch=gfx950 -O2 -o graph_branch graph_branch.hip ; run: ./graph_branch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// a dependent chain of `hops` loads per thread: latency-bound like the small face kernels (few hundred workgroups)
__global__ void chase(const int* __restrict__ next, double* __restrict__ out, int hops, int n) {
  int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
  double acc = 0.0;
  for (int h = 0; h < hops; ++h) { i = next[i]; acc += i; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const int n = 1 << 22, blocks = 300, threads = 192, hops = 12;
  int* next; double *o1, *o2, *o3, *o4;
  CK(hipMalloc(&next, n * sizeof(int)));
  { int* h = new int[n]; for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 1103515245LL + 12345) % n); CK(hipMemcpy(next, h, n * sizeof(int), hipMemcpyHostToDevice)); delete[] h; }
  CK(hipMalloc(&o1, blocks * threads * 8)); CK(hipMalloc(&o2, blocks * threads * 8)); CK(hipMalloc(&o3, blocks * threads * 8)); CK(hipMalloc(&o4, blocks * threads * 8));
  hipStream_t s0, s1, s2; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t fork, j1, j2; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
  auto serial = [&]() { for (double* o : {o1, o2, o3, o4}) hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, s0, next, o, hops, n); };
  auto events = [&]() {
    hipEventRecord(fork, s0);
    hipStreamWaitEvent(s1, fork, 0); hipStreamWaitEvent(s2, fork, 0);
    hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, s0, next, o1, hops, n);
    hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, s1, next, o2, hops, n);
    hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, s2, next, o3, hops, n);
    hipEventRecord(j1, s1); hipEventRecord(j2, s2);
    hipStreamWaitEvent(s0, j1, 0); hipStreamWaitEvent(s0, j2, 0);
    hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, s0, next, o4, hops, n);
  };
  auto time_it = [&](auto fn, int reps) {
    for (int i = 0; i < 20; ++i) fn();
    hipStreamSynchronize(s0);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) fn();
    hipStreamSynchronize(s0);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  };
  const int reps = 200;
  std::printf("one kernel alone      : %7.1f us\n", time_it([&]() { hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, s0, next, o1, hops, n); }, reps));
  std::printf("serial  (A B C D)     : %7.1f us per apply\n", time_it(serial, reps));
  std::printf("events  (A|B|C then D): %7.1f us per apply\n", time_it(events, reps));
  // graph: capture the event form once (10 applies per graph, like a smoother loop), replay
  for (int per : {1, 10}) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    for (int i = 0; i < per; ++i) events();
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double t = time_it([&]() { hipGraphLaunch(ge, s0); }, reps / per + 5);
    std::printf("graph of the event form, %2d applies per launch: %7.1f us per apply\n", per, t / per);
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    for (int i = 0; i < per; ++i) serial();
    CK(hipStreamEndCapture(s0, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    t = time_it([&]() { hipGraphLaunch(ge2, s0); }, reps / per + 5);
    std::printf("graph of the serial form, %2d applies per launch: %7.1f us per apply\n", per, t / per);
  }
  return 0;
}
