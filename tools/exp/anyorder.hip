// Does hipExtAnyOrderLaunch let kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is "not supported on AMD GFX9xx
// boards" for the module API; this measures it.)  N one-workgroup kernels that each spin ~1 ms on the 100 MHz clock are launched into
// one stream, first in order, then with the flag; total time ~N ms = serial, ~1 ms = overlapped.  Also: the same N kernels spread
// over S streams (S = 1, 2, 4, 8, 16) to show what a hardware queue per stream buys.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(unsigned long long ticks, int* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int k = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && k < (1 << 26)) ++k;  // bounded: always ends
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = k;
}

static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  const int N = 16;
  int* sink = nullptr;
  if (hipMalloc((void**)&sink, sizeof(int) * 64) != hipSuccess) return 1;
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const unsigned long long ticks = 100000;  // 1 ms at 100 MHz
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 1000ull, sink);
  hipStreamSynchronize(st);
  for (int mode = 0; mode < 2; ++mode) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, ticks, sink);
      else hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
    }
    const hipError_t e = hipStreamSynchronize(st);
    printf("%s: %d kernels of 1 ms in one stream: %.2f ms (%s)\n", mode ? "hipExtAnyOrderLaunch" : "in order", N, ms_since(t0), hipGetErrorString(e));
  }
  for (int S : {1, 2, 4, 8, 16}) {
    std::vector<hipStream_t> ss(S);
    for (auto& s : ss) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto& s : ss) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 1000ull, sink); }
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, ss[i % S], ticks, sink);
    hipDeviceSynchronize();
    printf("%d kernels of 1 ms over %d streams: %.2f ms\n", N, S, ms_since(t0));
    for (auto& s : ss) hipStreamDestroy(s);
  }
  hipFree(sink);
  return 0;
}
