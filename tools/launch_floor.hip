// microbenchmark: cost of back-to-back trivial kernels on one stream (eager vs graph, small vs large kernarg)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { unsigned v[200]; unsigned* p; };
__global__ void k_small(unsigned* p) { if (threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b) { if (threadIdx.x == 9999) *b.p = b.v[3]; }
__global__ void k_touch(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> double timeit(hipStream_t st, int n, F f) {
  for (int i = 0; i < 50; i++) f();
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) f();
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
  unsigned* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  Big b; b.p = d;
  for (int flags = 0; flags < 2; flags++) {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, flags ? hipStreamNonBlocking : hipStreamDefault));
    printf("stream flags=%s\n", flags ? "NonBlocking" : "Default");
    printf("  eager small  x1   : %.2f us/launch\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_small, 1, 64, 0, st, d); }));
    printf("  eager big    x1   : %.2f us/launch\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_big, 1, 64, 0, st, b); }));
    printf("  eager touch  x1   : %.2f us/launch\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_touch, 1, 64, 0, st, d); }));
    printf("  eager small 8192b : %.2f us/launch\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_small, 8192, 256, 0, st, d); }));
    for (int kind = 0; kind < 3; kind++) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 128; i++) {
        if (kind == 0) hipLaunchKernelGGL(k_small, 1, 64, 0, st, d);
        else if (kind == 1) hipLaunchKernelGGL(k_big, 1, 64, 0, st, b);
        else hipLaunchKernelGGL(k_small, 8192, 256, 0, st, d);
      }
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      double us = timeit(st, 100, [&] { hipGraphLaunch(ge, st); });
      printf("  graph of 128 %-11s: %.2f us/kernel (%.1f us/replay)\n", kind == 0 ? "small" : kind == 1 ? "big-arg" : "8192-block", us / 128, us);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    hipStreamDestroy(st);
  }
  return 0;
}
