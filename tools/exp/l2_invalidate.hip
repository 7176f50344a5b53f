// Developer experiment: do OTHER streams' kernel boundaries cost a running kernel its L2 contents?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/l2inv tools/exp/l2_invalidate.hip -lpthread && /tmp/l2inv
// Victim: one wave chases pointers through a 1 MB ring (fits the 4 MB L2 of its XCD) for many passes inside ONE launch
// and records the clock per pass.  Aggressors: T host threads launching empty kernels back to back on their own streams.
// Aggressor variants: (a) empty kernels, (b) kernels that only execute an agent-scope seq_cst fence per wave
// (buffer_wbl2 + buffer_inv) from 256 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <vector>

__global__ void chase_kernel(const unsigned* ring, int steps, int passes, unsigned long long* ticks, unsigned* sink) {
  unsigned i = threadIdx.x;
  for (int p = 0; p < passes; ++p) {
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = ring[i];
    if (threadIdx.x == 0) ticks[p] = wall_clock64() - t0;
  }
  sink[threadIdx.x] = i;
}
__global__ void empty_kernel() {}
__global__ void fence_kernel(unsigned* x) { __threadfence(); if (x && threadIdx.x == 1234567) *x = 1; }
__global__ void release_kernel(unsigned* x) { if (x) x[blockIdx.x * 64 + threadIdx.x] = 1; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__global__ void acquire_kernel(unsigned* x) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); if (x && threadIdx.x == 1234567) *x = 1; }

int main() {
  const int N = 1 << 18;  // 1 MB of unsigned
  std::vector<unsigned> h(N);
  // a stride permutation with 64 independent chains (one per lane), each visiting N/64 distinct 64-byte-apart entries
  for (int i = 0; i < N; ++i) h[i] = (unsigned)((i + 64 * 17) % N);
  unsigned* ring; hipMalloc(&ring, N * 4); hipMemcpy(ring, h.data(), N * 4, hipMemcpyHostToDevice);
  const int passes = 200, steps = N / 64 / 17 * 4;
  unsigned long long* ticks; hipMalloc(&ticks, passes * 8);
  unsigned* sink; hipMalloc(&sink, 64 * 4);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
  hipStream_t vs; hipStreamCreateWithFlags(&vs, hipStreamNonBlocking);
  unsigned* scratch; hipMalloc(&scratch, 256 * 64 * 4);
  for (int variant = 0; variant < 5; ++variant) {
    for (int T : {0, 1, 4, 8}) {
      if (variant == 0 && T > 0) continue;
      if (variant > 0 && T == 0) continue;
      std::atomic<bool> stop{false};
      std::atomic<long> launches{0};
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, variant] {
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        long n = 0;
        while (!stop.load()) {
          for (int k = 0; k < 32; ++k) {
            if (variant == 1) empty_kernel<<<64, 64, 0, s>>>();
            else if (variant == 2) fence_kernel<<<256, 64, 0, s>>>(nullptr);
            else if (variant == 3) release_kernel<<<256, 64, 0, s>>>(scratch);
            else acquire_kernel<<<256, 64, 0, s>>>(nullptr);
          }
          hipStreamSynchronize(s); n += 32;
        }
        launches += n; hipStreamDestroy(s);
      });
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
      chase_kernel<<<1, 64, 0, vs>>>(ring, steps, passes, ticks, sink);
      hipStreamSynchronize(vs);
      stop = true;
      for (auto& x : th) x.join();
      std::vector<unsigned long long> tk(passes);
      hipMemcpy(tk.data(), ticks, passes * 8, hipMemcpyDeviceToHost);
      double first = tk[0], rest = 0; for (int p = 1; p < passes; ++p) rest += tk[p];
      rest /= (passes - 1);
      printf("%-28s aggressor threads %d: first pass %7.1f ns/load, later passes %7.1f ns/load  (aggressor launches %ld)\n",
             variant == 0 ? "alone" : variant == 1 ? "empty kernels elsewhere" : variant == 2 ? "seq_cst fences elsewhere" : variant == 3 ? "store+release elsewhere" : "acquire fences elsewhere", T,
             1e6 * first / rate / steps, 1e6 * rest / rate / steps, launches.load());
      fflush(stdout);
    }
  }
  return 0;
}
