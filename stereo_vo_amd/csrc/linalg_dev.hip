// Parity tap of the device-side reduced-camera-system solve (csrc/lm_device.h): the same function the controller
// workgroup of ba_lm_kernel runs, on caller data, so that tests can compare it bit for bit with svo_cholesky_solve
// (host/linalg.cpp) — both are the dense LLT step of ceres::Solve's DENSE_SCHUR (reference src/bundle_adjuster.cpp:9,140).
#include "common.h"
#include "lm_device.h"

__global__ __launch_bounds__(128) void cholesky_solve_kernel(double* __restrict__ A, double* __restrict__ b, int n, int* __restrict__ ok_out) {
  extern __shared__ double lds[];  // A (n x n) | b (n) | col (6 n + 1: the panel columns of svo_dev_cholesky_solve)
  double* sA = lds;
  double* sb = sA + n * n;
  double* col = sb + n;
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) sA[i] = A[i];
  for (int i = threadIdx.x; i < n; i += blockDim.x) sb[i] = b[i];
  __syncthreads();
  const bool good = svo_dev_cholesky_solve(sA, sb, n, col);
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) A[i] = sA[i];
  for (int i = threadIdx.x; i < n; i += blockDim.x) b[i] = sb[i];
  if (threadIdx.x == 0) *ok_out = good ? 1 : 0;
}

extern "C" int svo_cholesky_solve_dev(svo_ctx* ctx, double* A, double* b, int n) {
  if (!ctx || !A || !b || n < 0) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  if (n == 0) return SVO_OK;
  const size_t lds = sizeof(double) * ((size_t)n * n + 7 * (size_t)n + 1);
  SVO_REQUIRE(ctx, lds <= 150 * 1024, "cholesky_solve_dev: system too large for one workgroup's LDS");
  SvoScratch s(ctx);
  double* dA = s.take<double>((size_t)n * n);
  double* db = s.take<double>(n);
  int* dok = s.take<int>(1);
  if (!dA || !db || !dok) { ctx->err = "cholesky_solve_dev: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dA, A, sizeof(double) * n * n, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(db, b, sizeof(double) * n, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)cholesky_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(cholesky_solve_kernel, dim3(1), dim3(128), lds, st, dA, db, n, dok);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  int ok = 0;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(A, dA, sizeof(double) * n * n, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(b, db, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(&ok, dok, sizeof(int), hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return ok ? SVO_OK : SVO_ERR_NUMERIC;
}
