// Developer experiment: aggregate small-kernel launch + retire rate against the number of host threads / streams.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_rate tools/exp/launch_rate.hip -lpthread && /tmp/launch_rate
// mode A: T threads, one stream each, every thread launches N dependent 64-wave kernels back to back, one sync at the end
// mode B: the same, but a host round trip (pinned flag written by the kernel, host spins) after every launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

__global__ void small_kernel(double* buf, volatile unsigned* flag, unsigned seq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  buf[i] = buf[i] * 1.0000001 + 1.0;
  if (flag && i == 0) { __threadfence_system(); *flag = seq; }
}

static double run(int T, int N, bool round_trip, int blocks) {
  std::vector<std::thread> th;
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<double> secs(T);
  for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* buf; hipMalloc(&buf, sizeof(double) * 64 * blocks); hipMemset(buf, 0, sizeof(double) * 64 * blocks);
    unsigned* flag; hipHostMalloc(&flag, 64, hipHostMallocDefault); *flag = 0;
    for (int w = 0; w < 200; ++w) small_kernel<<<blocks, 64, 0, s>>>(buf, nullptr, 0);
    hipStreamSynchronize(s);
    ready++; while (!go.load()) {}
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= N; ++i) {
      small_kernel<<<blocks, 64, 0, s>>>(buf, round_trip ? flag : nullptr, (unsigned)i);
      if (round_trip) while (*(volatile unsigned*)flag != (unsigned)i) {}
    }
    hipStreamSynchronize(s);
    secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    hipFree(buf); hipHostFree(flag); hipStreamDestroy(s);
  });
  while (ready.load() < T) {}
  go = true;
  for (auto& x : th) x.join();
  double worst = 0; for (double s : secs) worst = s > worst ? s : worst;
  return worst;
}

int main(int argc, char** argv) {
  const int N = 20000, blocks = argc > 1 ? atoi(argv[1]) : 64;
  for (int rt = 0; rt < 2; ++rt)
    for (int T : {1, 2, 4, 8, 16}) {
      const double s = run(T, N, rt, blocks);
      printf("%s  threads %2d  %7.1f k launches/s aggregate  %6.2f us per launch per thread\n", rt ? "round-trip" : "back-to-back", T,
             1e-3 * T * N / s, 1e6 * s / N);
      fflush(stdout);
    }
  return 0;
}
