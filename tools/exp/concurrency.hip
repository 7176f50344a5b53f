// Developer experiment: what makes small kernels of concurrent streams slow each other down?
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/conc tools/exp/concurrency.hip -lpthread && /tmp/conc
// T host threads, one stream each, each runs N round trips of one kernel (host waits on a pinned flag, as the LM loop does).
// Kernel variants: ALU only (a dependent f64 FMA chain, ~8 us), or a dependent chain of L2-missing loads (~8 us).
// Reported: GPU-side duration of the kernel (wall_clock64 inside, workgroup 0) and the host's time per round trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

__global__ void work_kernel(int mode, int iters, const unsigned* ring, unsigned ring_mask, double* sink, volatile unsigned* flag,
                            unsigned seq, unsigned* arrive, unsigned target, unsigned long long* ticks, int fence) {
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { __hip_atomic_store((unsigned*)flag + 16, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); if (fence) __threadfence_system(); }
  double a = threadIdx.x * 1e-9 + 1.0;
  unsigned i = (blockIdx.x * 977u + threadIdx.x * 64u) & ring_mask;
  if (mode == 0) { for (int k = 0; k < iters; ++k) a = a * 1.0000001 + 1e-9; }
  else { for (int k = 0; k < iters; ++k) i = ring[i]; a += i; }
  if (a == 12345.678) sink[0] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == target) {
      ticks[0] += wall_clock64() - t0;  // first-to-last: roughly the kernel's duration
      __hip_atomic_store((unsigned*)flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (fence) __threadfence_system();
    }
  }
}

static void run(int T, int N, int mode, int blocks, int iters, int fence) {
  std::vector<std::thread> th;
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<double> secs(T), gpu_us(T), start_us(T);
  for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const unsigned ring_n = 1u << 22;  // 16 MB per thread: misses L2
    unsigned* ring; (void)hipMalloc(&ring, ring_n * 4);
    std::vector<unsigned> h(ring_n);
    for (unsigned i = 0; i < ring_n; ++i) h[i] = (i * 2654435761u + 12345u) & (ring_n - 1);
    (void)hipMemcpy(ring, h.data(), ring_n * 4, hipMemcpyHostToDevice);
    double* sink; (void)hipMalloc(&sink, 8);
    unsigned* arrive; (void)hipMalloc(&arrive, 4); (void)hipMemset(arrive, 0, 4);
    unsigned long long* ticks; (void)hipMalloc(&ticks, 8); (void)hipMemset(ticks, 0, 8);
    unsigned* flag; (void)hipHostMalloc(&flag, 128, hipHostMallocCoherent); flag[0] = 0; flag[16] = 0;
    double t_start_acc = 0;
    unsigned total = 0;
    auto once = [&](unsigned seq) {
      total += blocks;
      const auto tl = std::chrono::steady_clock::now();
      work_kernel<<<blocks, 64, 0, s>>>(mode, iters, ring, ring_n - 1, sink, flag, seq, arrive, total, ticks, fence);
      while (*(volatile unsigned*)(flag + 16) != seq) {}
      t_start_acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl).count();
      while (*(volatile unsigned*)flag != seq) {}
    };
    for (unsigned w = 1; w <= 50; ++w) once(w);
    (void)hipMemset(ticks, 0, 8); t_start_acc = 0;
    ready++; while (!go.load()) {}
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) once(1000u + i);
    secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long tk = 0; (void)hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
    gpu_us[t] = 0.01 * (double)tk / N; start_us[t] = 1e6 * t_start_acc / N;
    (void)hipFree(ring); (void)hipFree(sink); (void)hipFree(arrive); (void)hipFree(ticks); (void)hipHostFree(flag); (void)hipStreamDestroy(s);
  });
  while (ready.load() < T) {}
  go = true;
  for (auto& x : th) x.join();
  double s = 0, g = 0, st = 0; for (int t = 0; t < T; ++t) { s += secs[t]; g += gpu_us[t]; st += start_us[t]; }
  printf("%-5s %s blocks %4d  threads %2d : launch call -> start word seen %6.1f us, kernel %6.1f us on the GPU, round trip %6.1f us\n", mode ? "loads" : "alu", fence ? "fenced " : "relaxed", blocks, T, st / T, g / T, 1e6 * s / T / N);
  fflush(stdout);
}

int main() {
  for (int mode = 0; mode < 2; ++mode)
    for (int blocks : {8, 128})
      for (int T : {1, 8}) for (int fence : {0, 1}) run(T, 3000, mode, blocks, mode ? 24 : 2400, fence);
  return 0;
}
