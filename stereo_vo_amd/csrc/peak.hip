// Measurement aid: the machine's own ceilings, measured on the card the bench runs on, for the roofline
// objects of bench.py — f64 vector FMA rate, f64 matrix (v_mfma_f64_16x16x4_f64) rate, and HBM copy bandwidth.
// The local hardware guide has no FP64 row; 78.6 TFLOP/s is AMD's public figure for both pipes.
#include "common.h"

namespace {
constexpr int PEAK_ITERS = 4096;
typedef double peak_d4 __attribute__((ext_vector_type(4)));

// 8 independent fma chains per lane: 2 * 8 * PEAK_ITERS flop per lane
__global__ __launch_bounds__(256) void peak_f64_fma_kernel(double* out, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#pragma unroll 8
  for (int i = 0; i < PEAK_ITERS; ++i) {
    x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
    x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

// the arithmetic the BA kernels actually issue (no FMA contraction: bit-exact parity with the CPU oracle): separate
// multiply and add, 8 chains: 2 * 8 * PEAK_ITERS flop per lane in twice the instructions
__global__ __launch_bounds__(256) void peak_f64_muladd_kernel(double* out, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#pragma unroll 8
  for (int i = 0; i < PEAK_ITERS; ++i) {
    x0 = __dadd_rn(__dmul_rn(x0, a), b); x1 = __dadd_rn(__dmul_rn(x1, a), b); x2 = __dadd_rn(__dmul_rn(x2, a), b); x3 = __dadd_rn(__dmul_rn(x3, a), b);
    x4 = __dadd_rn(__dmul_rn(x4, a), b); x5 = __dadd_rn(__dmul_rn(x5, a), b); x6 = __dadd_rn(__dmul_rn(x6, a), b); x7 = __dadd_rn(__dmul_rn(x7, a), b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

// 4 independent accumulator tiles per wave: 4 * 2048 flop per wave and iteration
__global__ __launch_bounds__(256) void peak_f64_mfma_kernel(double* out, double a, double b) {
  peak_d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double av = a + threadIdx.x * 1e-9, bv = b;
#pragma unroll 4
  for (int i = 0; i < PEAK_ITERS; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (c0[0] + c1[1]) + (c2[2] + c3[3]);
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace

extern "C" int svo_measure_peak(svo_ctx* ctx, const char* what, double* value) {
  if (!ctx || !what || !value) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  hipStream_t st = ctx->stream;
  hipEvent_t e0, e1;
  SVO_HIP_CHECK(ctx, hipEventCreate(&e0));
  SVO_HIP_CHECK(ctx, hipEventCreate(&e1));
  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  const int grid = cus * 8, block = 256;
  float best_ms = 1e30f;
  double work = 0.0;  // flop or bytes per launch
  int rc = SVO_OK;
  const bool copy = !strcmp(what, "hbm_copy");
  if (copy) {
    const size_t bytes = (size_t)512 << 20;  // 512 MiB each way: far beyond the 256 MiB Infinity Cache
    uint4 *src = nullptr, *dst = nullptr;
    SVO_HIP_CHECK(ctx, hipMalloc((void**)&src, bytes));
    if (hipMalloc((void**)&dst, bytes) != hipSuccess) { (void)hipFree(src); ctx->err = "measure_peak: allocation failed"; return SVO_ERR_HIP; }
    (void)hipMemsetAsync(src, 1, bytes, st);
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0, st);
      hipLaunchKernelGGL(peak_copy_kernel, dim3(cus * 16), dim3(block), 0, st, src, dst, bytes / 16);
      (void)hipEventRecord(e1, st);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best_ms) best_ms = ms;
    }
    work = 2.0 * (double)bytes;
    (void)hipFree(src); (void)hipFree(dst);
  } else {
    double* out = nullptr;
    SVO_HIP_CHECK(ctx, hipMalloc((void**)&out, sizeof(double) * (size_t)grid * block));
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0, st);
      if (!strcmp(what, "f64_fma")) hipLaunchKernelGGL(peak_f64_fma_kernel, dim3(grid), dim3(block), 0, st, out, 0.999999, 1e-6);
      else if (!strcmp(what, "f64_muladd")) hipLaunchKernelGGL(peak_f64_muladd_kernel, dim3(grid), dim3(block), 0, st, out, 0.999999, 1e-6);
      else if (!strcmp(what, "f64_mfma")) hipLaunchKernelGGL(peak_f64_mfma_kernel, dim3(grid), dim3(block), 0, st, out, 0.5, 0.25);
      else { rc = SVO_ERR_INVALID; break; }
      (void)hipEventRecord(e1, st);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best_ms) best_ms = ms;
    }
    if (!strcmp(what, "f64_mfma")) work = (double)grid * (block / 64) * 4.0 * 2048.0 * PEAK_ITERS;
    else work = (double)grid * block * 2.0 * 8.0 * PEAK_ITERS;
    (void)hipFree(out);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) { ctx->err = "measure_peak: unknown quantity (f64_fma, f64_muladd, f64_mfma, hbm_copy)"; return rc; }
  *value = work / ((double)best_ms * 1e-3);  // flop/s or bytes/s
  return SVO_OK;
}
