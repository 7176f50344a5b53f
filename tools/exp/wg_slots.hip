// How many one-wavefront workgroups does a CU of gfx950 hold at once?  (The stream-batched tracker is one wavefront = one workgroup
// per feature: if the workgroup slots of a CU run out before its wavefront slots or registers do, building the kernel for more
// wavefronts per SIMD cannot raise its occupancy.)  N x 256 workgroups of 64 threads that each spin ~1 ms on the 100 MHz clock:
// total ~1 ms while N workgroups fit a CU at once, ~2 ms from the first N that does not.  Three variants: no LDS, 1.7 KB of LDS per
// workgroup (the tracker's), and two wavefronts per workgroup.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

template <int LDS_BYTES>
__global__ __launch_bounds__(64) void spin1(unsigned long long ticks, int* sink) {
  __shared__ int pad[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int k = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && k < (1 << 26)) { ++k; __builtin_amdgcn_s_sleep(8); }  // bounded: always ends
  if (LDS_BYTES > 0) pad[threadIdx.x % (LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1)] = k;
  if (threadIdx.x == 0 && sink) sink[blockIdx.x & 63] = k + (LDS_BYTES > 0 ? pad[0] : 0);
}
__global__ __launch_bounds__(128) void spin2(unsigned long long ticks, int* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int k = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && k < (1 << 26)) { ++k; __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0 && sink) sink[blockIdx.x & 63] = k;
}

template <typename F>
static double timed(F launch) {
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  launch();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  int* sink = nullptr;
  if (hipMalloc((void**)&sink, sizeof(int) * 64) != hipSuccess) return 1;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const unsigned long long ticks = 100000;  // 1 ms
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin1<0>, 64, 0);
  printf("%d CUs; occupancy query: %d one-wavefront workgroups per CU without LDS", cus, occ);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin1<1728>, 64, 0);
  printf(", %d with 1.7 KB of LDS", occ);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin2, 128, 0);
  printf(", %d two-wavefront workgroups\n", occ);
  timed([&] { hipLaunchKernelGGL(spin1<0>, dim3(cus), dim3(64), 0, 0, 1000ull, sink); });
  for (int n : {8, 16, 20, 24, 32, 40}) {
    const double a = timed([&] { hipLaunchKernelGGL(spin1<0>, dim3(n * cus), dim3(64), 0, 0, ticks, sink); });
    const double b = timed([&] { hipLaunchKernelGGL(spin1<1728>, dim3(n * cus), dim3(64), 0, 0, ticks, sink); });
    const double c = timed([&] { hipLaunchKernelGGL(spin2, dim3(n * cus / 2), dim3(128), 0, 0, ticks, sink); });
    printf("%2d wavefronts per CU asked for: one per workgroup %.2f ms, with 1.7 KB LDS %.2f ms, two per workgroup %.2f ms\n", n, a, b, c);
  }
  hipFree(sink);
  return 0;
}
