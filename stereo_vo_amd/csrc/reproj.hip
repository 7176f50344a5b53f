// a11 — batched ReprojectionFactor::Evaluate (reference src/reprojection_factor.cpp:10-88).
// One lane per observation; FP64 VALU.  Algorithmic traffic per observation:
// in 7+3+2 doubles, out 2+14+6 doubles = 272 B for ~250 flops => HBM-bound as a stand-alone call
// (inside the BA kernels the same device function is fused and nothing is written back).
#include "common.h"
#include "reproj_device.h"

__global__ __launch_bounds__(256) void reproj_eval_kernel(int n, const double* __restrict__ pose7,
                                                          const double* __restrict__ point3,
                                                          const double* __restrict__ obs2, double f,
                                                          double cx, double cy, double* __restrict__ r2,
                                                          double* __restrict__ jpose14,
                                                          double* __restrict__ jpoint6) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double q[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) q[k] = pose7[7 * (size_t)i + k];
    const D3 p{point3[3 * (size_t)i], point3[3 * (size_t)i + 1], point3[3 * (size_t)i + 2]};
    double r[2], Jq[14], Jx[6];
    reproj_full(q, p, obs2[2 * (size_t)i], obs2[2 * (size_t)i + 1], f, cx, cy, r, jpose14 ? Jq : nullptr,
                jpoint6 ? Jx : nullptr);
    r2[2 * (size_t)i] = r[0];
    r2[2 * (size_t)i + 1] = r[1];
    if (jpose14) {
#pragma unroll
      for (int k = 0; k < 14; ++k) jpose14[14 * (size_t)i + k] = Jq[k];
    }
    if (jpoint6) {
#pragma unroll
      for (int k = 0; k < 6; ++k) jpoint6[6 * (size_t)i + k] = Jx[k];
    }
  }
}

extern "C" int svo_reproj_eval_dev(svo_ctx* ctx, int n, const double* pose7, const double* point3,
                                   const double* obs2, double focal, double cx, double cy, double* r2,
                                   double* jpose14, double* jpoint6) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || (pose7 && point3 && obs2 && r2)), "reproj_eval: null buffer");
  if (n == 0) return SVO_OK;
  const int block = 256;
  const int grid = svo_div_up(n, block) < 2048 ? svo_div_up(n, block) : 2048;
  hipLaunchKernelGGL(reproj_eval_kernel, dim3(grid), dim3(block), 0, ctx->stream, n, pose7, point3, obs2, focal, cx,
                     cy, r2, jpose14, jpoint6);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

extern "C" int svo_reproj_eval(svo_ctx* ctx, int n, const double* pose7, const double* point3,
                               const double* obs2, double focal, double cx, double cy, double* r2,
                               double* jpose14, double* jpoint6) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || (pose7 && point3 && obs2 && r2)), "reproj_eval: null buffer");
  hipStream_t st = ctx->stream;
  // host batches are streamed through the fixed workspace in chunks (34 doubles per observation)
  const int chunk_max = (int)((ctx->ws_bytes - 4096) / (34 * sizeof(double)));
  for (int base = 0; base < n; base += chunk_max) {
    const int m = n - base < chunk_max ? n - base : chunk_max;
    SvoScratch s(ctx);
    double* d_pose = s.take<double>(7 * (size_t)m);
    double* d_pt = s.take<double>(3 * (size_t)m);
    double* d_obs = s.take<double>(2 * (size_t)m);
    double* d_r = s.take<double>(2 * (size_t)m);
    double* d_jq = jpose14 ? s.take<double>(14 * (size_t)m) : nullptr;
    double* d_jx = jpoint6 ? s.take<double>(6 * (size_t)m) : nullptr;
    if (!d_pose || !d_pt || !d_obs || !d_r || (jpose14 && !d_jq) || (jpoint6 && !d_jx)) {
      ctx->err = "reproj_eval: workspace too small";
      return SVO_ERR_CAPACITY;
    }
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(d_pose, pose7 + 7 * (size_t)base, sizeof(double) * 7 * m, hipMemcpyHostToDevice, st));
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(d_pt, point3 + 3 * (size_t)base, sizeof(double) * 3 * m, hipMemcpyHostToDevice, st));
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(d_obs, obs2 + 2 * (size_t)base, sizeof(double) * 2 * m, hipMemcpyHostToDevice, st));
    int rc = svo_reproj_eval_dev(ctx, m, d_pose, d_pt, d_obs, focal, cx, cy, d_r, d_jq, d_jx);
    if (rc) return rc;
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(r2 + 2 * (size_t)base, d_r, sizeof(double) * 2 * m, hipMemcpyDeviceToHost, st));
    if (jpose14) SVO_HIP_CHECK(ctx, hipMemcpyAsync(jpose14 + 14 * (size_t)base, d_jq, sizeof(double) * 14 * m, hipMemcpyDeviceToHost, st));
    if (jpoint6) SVO_HIP_CHECK(ctx, hipMemcpyAsync(jpoint6 + 6 * (size_t)base, d_jx, sizeof(double) * 6 * m, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  return SVO_OK;
}
