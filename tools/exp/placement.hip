// Developer experiment: are the workgroups of concurrent kernels spread over idle CUs or packed onto the same ones?
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/placement tools/exp/placement.hip -lpthread && GPU_MAX_HW_QUEUES=16 /tmp/placement
// T streams each run the same ALU-bound kernel (blocks x 256 threads, one wave per SIMD of a CU) at the same time; the
// kernel's own clock says how long its slowest workgroup took.  128 workgroups = half of the 256 CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <atomic>
#include <thread>
#include <vector>

__global__ void alu_kernel(int iters, double* sink, unsigned long long* ticks, unsigned* cu_hist) {
  const unsigned long long t0 = wall_clock64();
  double a = 1.0 + threadIdx.x * 1e-9;
  for (int k = 0; k < iters; ++k) a = a * 1.0000001 + 1e-9;
  if (a == 1234.5) sink[0] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(ticks, wall_clock64() - t0);
    unsigned hw_id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    unsigned xcc_id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const unsigned cu = (hw_id >> 8) & 15, se = (hw_id >> 13) & 7, xcc = xcc_id & 15;
    atomicAdd(&cu_hist[(xcc * 8 + se) * 16 + cu], 1u);
  }
}

int main() {
  for (int blocks : {64, 128, 183}) for (int T : {1, 2, 4, 8}) {
    std::vector<std::thread> th;
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<double> us(T);
    unsigned* hist; (void)hipMalloc(&hist, 8 * 8 * 16 * 4); (void)hipMemset(hist, 0, 8 * 8 * 16 * 4);
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
      hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      double* sink; (void)hipMalloc(&sink, 8);
      unsigned long long* ticks; (void)hipMalloc(&ticks, 8); (void)hipMemsetAsync(ticks, 0, 8, s);
      alu_kernel<<<blocks, 256, 0, s>>>(100, sink, ticks, hist); (void)hipStreamSynchronize(s);
      (void)hipMemsetAsync(ticks, 0, 8, s); (void)hipMemsetAsync(hist, 0, 8 * 8 * 16 * 4, s); (void)hipStreamSynchronize(s);
      ready++; while (!go.load()) {}
      alu_kernel<<<blocks, 256, 0, s>>>(20000, sink, ticks, hist);
      (void)hipStreamSynchronize(s);
      unsigned long long tk = 0; (void)hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
      us[t] = 0.01 * (double)tk;
    });
    while (ready.load() < T) {}
    go = true;
    for (auto& x : th) x.join();
    std::vector<unsigned> h(8 * 8 * 16);
    (void)hipMemcpy(h.data(), hist, h.size() * 4, hipMemcpyDeviceToHost);
    int used = 0, maxper = 0; for (unsigned v : h) { if (v) ++used; if ((int)v > maxper) maxper = v; }
    double s = 0; for (double v : us) s += v;
    printf("%3d workgroups x %d streams: slowest workgroup %7.1f us on average; %3d distinct CUs used, at most %d workgroups on one CU\n", blocks, T, s / T, used, maxper);
    (void)hipFree(hist);
  }
  return 0;
}
