// Developer experiment: host <-> resident-kernel handshake through pinned memory, against the number of concurrent pairs.
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/pingpong tools/exp/pingpong.hip -lpthread && GPU_MAX_HW_QUEUES=16 /tmp/pingpong
// T host threads, one stream and ONE launch each: the kernel (blocks workgroups) loops N times: workgroup 0 polls a command
// word in pinned host memory, posts it to the other workgroups through device memory, all do `work` us of ALU, the last to
// arrive writes the acknowledgement word to pinned host memory.  The host thread writes command i and spins for ack i.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

__global__ void resident_kernel(volatile unsigned* cmd, volatile unsigned* ack, unsigned* post, unsigned* arrive, int n, int iters, double* sink) {
  __shared__ unsigned sSeq;
  double a = 1.0 + threadIdx.x * 1e-9;
  for (unsigned i = 1; i <= (unsigned)n; ++i) {
    if (threadIdx.x == 0) {
      if (blockIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load((unsigned*)cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != i) { if (++spins > (1u << 22)) break; }
        __hip_atomic_store(post, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        unsigned spins = 0;
        while (__hip_atomic_load(post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != i) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) break; }
      }
      sSeq = i;
    }
    __syncthreads();
    for (int k = 0; k < iters; ++k) a = a * 1.0000001 + 1e-9;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == i * gridDim.x) __hip_atomic_store((unsigned*)ack, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (a == 1234.5) sink[0] = a;
}

static void run(int T, int N, int blocks, int iters) {
  std::vector<std::thread> th;
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<double> secs(T);
  for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned* pin; (void)hipHostMalloc(&pin, 256, hipHostMallocCoherent); pin[0] = 0; pin[32] = 0;
    unsigned* dev; (void)hipMalloc(&dev, 256); (void)hipMemsetAsync(dev, 0, 256, s);
    double* sink; (void)hipMalloc(&sink, 8);
    (void)hipStreamSynchronize(s);  // never a device-wide wait: the other threads' resident kernels only end when THEIR hosts are done
    resident_kernel<<<blocks, 64, 0, s>>>(pin, pin + 32, dev, dev + 32, N, iters, sink);
    ready++; while (!go.load()) {}
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned i = 1; i <= (unsigned)N; ++i) {
      __atomic_store_n(pin, i, __ATOMIC_RELEASE);
      { unsigned long spins = 0; while (__atomic_load_n(pin + 32, __ATOMIC_ACQUIRE) != i) { if (++spins > 2000000000ul) { fprintf(stderr, "host gave up\n"); break; } } }
    }
    secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipStreamSynchronize(s);
    (void)hipFree(dev); (void)hipFree(sink); (void)hipHostFree(pin); (void)hipStreamDestroy(s);
  });
  while (ready.load() < T) {}
  go = true;
  for (auto& x : th) x.join();
  double s = 0; for (double v : secs) s += v;
  printf("resident kernels %2d x %3d workgroups, %s work: %6.2f us per host->device->host round trip\n", T, blocks, iters ? "~8 us of ALU" : "no", 1e6 * s / T / N);
  fflush(stdout);
}

int main() {
  for (int blocks : {1, 64})
    for (int iters : {0, 600})
      for (int T : {1, 8}) run(T, 2000, blocks, iters);
  return 0;
}
