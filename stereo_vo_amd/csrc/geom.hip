// a8 — per-feature stereo reprojection (reference src/image_processor.cpp:178-207) and
// a6 — new-vs-tracked dedup (reference src/image_processor.cpp:113-128).
// Both are "predicate + stable compaction" over at most a few thousand features: one 1024-thread
// workgroup, ballot/popcount prefix inside each wave, LDS prefix across the 16 waves.  Output order is
// the input order, as the reference's push_back loops produce.
#include "kernels.h"
#include "group_kernels.h"
#include "ref_constants.h"
#include "tail_device.h"

namespace {
constexpr int CT = 1024;
}  // namespace

__global__ __launch_bounds__(CT) void triangulate_kernel(const float* __restrict__ xy, const float* __restrict__ disp,
                                                         const int* __restrict__ n_dev, int n_host, SvoMat4 M,
                                                         float* __restrict__ kept_xy, float* __restrict__ xyz,
                                                         int* __restrict__ kept_index, int* __restrict__ n_kept,
                                                         SvoPublish pub) {
  svo_latency_critical();
  __shared__ int sWave[CT / 64];
  const int n = n_dev ? *n_dev : n_host;
  svo_triangulate_block<CT, false>(xy, disp, n, M, kept_xy, xyz, kept_index, n_kept, sWave);
  svo_publish_block(pub);  // the outputs may be pinned host memory (pipeline): the host polls instead of copying
}

// a6 in ONE launch.  One wavefront per detected corner: 64 tracked features are tested per step, any hit drops the corner;
// the last workgroup to arrive compacts the survivors in input order (tail_device.h).
__device__ __forceinline__ void dedup_body(const float* __restrict__ det, const int* __restrict__ n_det_dev,
                                           int n_det_host, const float* __restrict__ trk,
                                           const int* __restrict__ n_trk_dev, int n_trk_host, float min_d,
                                           uint8_t* keep, float* __restrict__ kept_xy, int* __restrict__ n_kept,
                                           unsigned* arrive, unsigned target) {
  svo_latency_critical();
  __shared__ int sWave[4];
  __shared__ int sLast;
  const int nd = n_det_dev ? *n_det_dev : n_det_host;
  const int nt = n_trk_dev ? *n_trk_dev : n_trk_host;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i < nd) {
    const float x = det[2 * i], y = det[2 * i + 1];
    bool hit = false;
    for (int j0 = 0; j0 < nt && !hit; j0 += 64) {
      const int j = j0 + lane;
      bool h = false;
      if (j < nt) {
        const float dx = x - trk[2 * j], dy = y - trk[2 * j + 1];
        h = sqrtf(dx * dx + dy * dy) < min_d;  // src/image_processor.cpp:118-123
      }
      hit = __any(h);
    }
    if (lane == 0) svo_wt_store(&keep[i], (uint8_t)(hit ? 0 : 1));
  }
  if (!svo_last_arrival(arrive, target, &sLast)) return;
  int base = 0;
  for (int c0 = 0; c0 < nd; c0 += 256) {
    const int k = c0 + threadIdx.x;
    const bool kp = k < nd && svo_coherent_load(&keep[k]) != 0;
    const int slot = svo_compact_slot<256>(kp, base, sWave);
    if (slot >= 0) { kept_xy[2 * slot] = det[2 * k]; kept_xy[2 * slot + 1] = det[2 * k + 1]; }
  }
  if (threadIdx.x == 0) *n_kept = base;
}

__global__ __launch_bounds__(256) void dedup_kernel(const float* __restrict__ det, const int* __restrict__ n_det_dev,
                                                    int n_det_host, const float* __restrict__ trk,
                                                    const int* __restrict__ n_trk_dev, int n_trk_host, float min_d,
                                                    uint8_t* keep, float* __restrict__ kept_xy, int* __restrict__ n_kept,
                                                    unsigned* arrive, unsigned target) {
  dedup_body(det, n_det_dev, n_det_host, trk, n_trk_dev, n_trk_host, min_d, keep, kept_xy, n_kept, arrive, target);
}


__global__ void gather_track_kernel(const int* __restrict__ idx, const int* __restrict__ n_dev, int n_host,
                                    const float* __restrict__ init_src, const long long* __restrict__ ids_src,
                                    float* __restrict__ init_dst, long long* __restrict__ ids_dst) {
  svo_latency_critical();
  const int n = n_dev ? *n_dev : n_host;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int s = idx[i];
    init_dst[2 * i] = init_src[2 * s]; init_dst[2 * i + 1] = init_src[2 * s + 1];
    ids_dst[i] = ids_src[s];
  }
}

int svo_k_gather_track(svo_ctx* ctx, const int* idx, const int* n_dev, int n_max, const float* init_src,
                       const long long* ids_src, float* init_dst, long long* ids_dst) {
  if (n_max <= 0) return SVO_OK;
  hipLaunchKernelGGL(gather_track_kernel, dim3(svo_div_up(n_max, 256)), dim3(256), 0, ctx->stream, idx, n_dev, n_max,
                     init_src, ids_src, init_dst, ids_dst);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

__global__ void tracker_init_kernel(const float* __restrict__ h_xy, const float* __restrict__ h_init, const long long* __restrict__ h_ids, int n,
                                    float* __restrict__ d_xy, float* __restrict__ d_init, long long* __restrict__ d_ids,
                                    SvoPublish pub) {
  svo_latency_critical();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float x = h_xy[2 * i], y = h_xy[2 * i + 1];
    d_xy[2 * i] = x; d_xy[2 * i + 1] = y;
    d_init[2 * i] = h_init[2 * i]; d_init[2 * i + 1] = h_init[2 * i + 1];  // = (x, y) unless init() saw duplicate ids
    d_ids[i] = h_ids[i];
  }
  svo_publish_block(pub);
}

int svo_k_tracker_init(svo_ctx* ctx, const float* h_xy, const float* h_init, const long long* h_ids, int n, float* d_xy, float* d_init,
                       long long* d_ids, const SvoPublish* pub) {
  hipLaunchKernelGGL(tracker_init_kernel, dim3(1), dim3(1024), 0, ctx->stream, h_xy, h_init, h_ids, n, d_xy, d_init, d_ids,
                     pub ? *pub : SvoPublish{});
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_k_gather_xy_ids(svo_ctx* ctx, const int* idx, int n, const float* xy_src, const long long* ids_src,
                        float* xy_dst, long long* ids_dst) {
  return svo_k_gather_track(ctx, idx, nullptr, n, xy_src, ids_src, xy_dst, ids_dst);
}

SvoMat4 svo_k_reprojection_matrix(const float* pose16, float focal, float cx, float cy, float baseline) {
  float Q[16] = {0};
  Q[0] = (float)(1.0 / (double)focal);
  Q[5] = (float)(1.0 / (double)focal);
  Q[3] = -cx / focal;
  Q[7] = -cy / focal;
  Q[11] = 1.0f;
  Q[14] = (float)(1.0 / (double)(baseline * focal));
  SvoMat4 M;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += (double)pose16[4 * i + k] * (double)Q[4 * k + j];
      M.m[4 * i + j] = (float)s;
    }
  return M;
}

int svo_k_triangulate(svo_ctx* ctx, const float* xy, const float* disp, const int* n_dev, int n_max,
                      const SvoMat4& M, float* kept_xy, float* xyz, int* kept_index, int* n_kept, const SvoPublish* pub) {
  SvoProfScope prof(ctx, SVO_PROF_TRIANGULATE);
  hipLaunchKernelGGL(triangulate_kernel, dim3(1), dim3(CT), 0, ctx->stream, xy, disp, n_dev, n_max, M, kept_xy, xyz,
                     kept_index, n_kept, pub ? *pub : SvoPublish{});
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_k_dedup(svo_ctx* ctx, const float* det_xy, const int* n_det_dev, int n_det_max, const float* trk_xy,
                const int* n_trk_dev, int n_trk_max, float min_distance, float* kept_xy, int* n_kept) {
  if (n_det_max > ctx->lim.max_batch * ctx->lim.max_candidates) { ctx->err = "dedup: too many detected corners"; return SVO_ERR_CAPACITY; }
  uint8_t* flags = ctx->d_state;  // scratch (the corner-select state array is idle here)
  const int grid = n_det_max > 0 ? svo_div_up(n_det_max, 4) : 1;
  const SvoPublish arr = svo_arrive_next(ctx, grid);
  hipLaunchKernelGGL(dedup_kernel, dim3(grid), dim3(256), 0, ctx->stream, det_xy, n_det_dev, n_det_max, trk_xy, n_trk_dev,
                     n_trk_max, min_distance, flags, kept_xy, n_kept, arr.arrive, arr.target);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

extern "C" int svo_triangulate(svo_ctx* ctx, const float* xy, const float* disp, int n, const float* pose16,
                               float focal, float cx, float cy, float baseline, float* kept_xy, float* xyz,
                               int* kept_index, int* n_kept) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, n >= 0 && pose16 && n_kept && (n == 0 || (xy && disp && kept_xy && xyz)), "triangulate: null buffer");
  *n_kept = 0;
  if (n == 0) return SVO_OK;
  SvoScratch s(ctx);
  float* dxy = s.take<float>(2 * (size_t)n);
  float* dd = s.take<float>(n);
  float* dk = s.take<float>(2 * (size_t)n);
  float* d3 = s.take<float>(3 * (size_t)n);
  int* di = s.take<int>(n);
  int* dn = s.take<int>(1);
  if (!dxy || !dd || !dk || !d3 || !di || !dn) { ctx->err = "triangulate: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dd, disp, sizeof(float) * n, hipMemcpyHostToDevice, st));
  const SvoMat4 M = svo_k_reprojection_matrix(pose16, focal, cx, cy, baseline);
  int rc = svo_k_triangulate(ctx, dxy, dd, nullptr, n, M, dk, d3, di, dn);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(n_kept, dn, sizeof(int), hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const int m = *n_kept;
  if (m > 0) {
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(kept_xy, dk, sizeof(float) * 2 * m, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(xyz, d3, sizeof(float) * 3 * m, hipMemcpyDeviceToHost, st));
    if (kept_index) SVO_HIP_CHECK(ctx, hipMemcpyAsync(kept_index, di, sizeof(int) * m, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  return SVO_OK;
}

extern "C" int svo_dedup(svo_ctx* ctx, const float* detected_xy, int n_detected, const float* tracked_xy,
                         int n_tracked, float min_distance, float* kept_xy, int* n_kept) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, n_detected >= 0 && n_tracked >= 0 && n_kept, "dedup: bad sizes");
  SVO_REQUIRE(ctx, (n_detected == 0 || (detected_xy && kept_xy)) && (n_tracked == 0 || tracked_xy), "dedup: null buffer");
  *n_kept = 0;
  if (n_detected == 0) return SVO_OK;
  SvoScratch s(ctx);
  float* dd = s.take<float>(2 * (size_t)n_detected);
  float* dt = s.take<float>(2 * (size_t)(n_tracked > 0 ? n_tracked : 1));
  float* dk = s.take<float>(2 * (size_t)n_detected);
  int* dn = s.take<int>(1);
  if (!dd || !dt || !dk || !dn) { ctx->err = "dedup: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dd, detected_xy, sizeof(float) * 2 * n_detected, hipMemcpyHostToDevice, st));
  if (n_tracked > 0) SVO_HIP_CHECK(ctx, hipMemcpyAsync(dt, tracked_xy, sizeof(float) * 2 * n_tracked, hipMemcpyHostToDevice, st));
  int rc = svo_k_dedup(ctx, dd, nullptr, n_detected, dt, nullptr, n_tracked, min_distance, dk, dn);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(n_kept, dn, sizeof(int), hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (*n_kept > 0) {
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(kept_xy, dk, sizeof(float) * 2 * (size_t)*n_kept, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  return SVO_OK;
}
