// How many kernels execute at once on gfx950 under this runtime?  S streams, one 1-ms one-workgroup kernel each (spinning on the 100 MHz
// clock), all launched back to back: the total is 1 ms while all S run concurrently and k ms when only S / k do.  Run under different
// GPU_MAX_HW_QUEUES (the runtime reads it at start-up).  Round 5: the pipeline groups' plateau looked like a cap on concurrent kernels.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin_kernel(unsigned long long ticks, int* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int k = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && k < (1 << 26)) { ++k; __builtin_amdgcn_s_sleep(8); }  // bounded: always ends
  if (threadIdx.x == 0 && sink) sink[blockIdx.x & 63] = k;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int* sink = nullptr;
  if (hipMalloc((void**)&sink, sizeof(int) * 64) != hipSuccess) return 1;
  const char* q = getenv("GPU_MAX_HW_QUEUES");
  printf("GPU_MAX_HW_QUEUES=%s\n", q ? q : "(unset: 4)");
  const unsigned long long ticks = 100000;  // 1 ms
  for (int S : {1, 2, 4, 6, 8, 10, 12, 16, 24, 32}) {
    std::vector<hipStream_t> ss(S);
    for (auto& s : ss) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto& s : ss) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 1000ull, sink);  // first use: binds the stream to a queue
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& s : ss) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, sink);
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("  %2d streams, one 1-ms kernel each: %.2f ms  -> %.1f kernels at once\n", S, ms, S / ms);
    for (auto& s : ss) hipStreamDestroy(s);
  }
  hipFree(sink);
  return 0;
}
