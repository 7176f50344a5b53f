// Internal device-pointer entry points shared between the C-ABI wrappers and the in-library
// pipeline (host/pipeline.cpp).  All asynchronous on ctx->stream.  Counts that are produced on the
// device stay on the device (`const int* n_dev`) so stages chain without host round trips;
// `n_max` bounds the launch.
#ifndef SVO_KERNELS_H_
#define SVO_KERNELS_H_
#include "common.h"

struct SvoMat4 { float m[16]; };

// a8: M = float(pose * Q) formed on the host in the reference's order (src/image_processor.cpp:183-189,202).
SvoMat4 svo_k_reprojection_matrix(const float* pose16, float focal, float cx, float cy, float baseline);
int svo_k_triangulate(svo_ctx* ctx, const float* xy, const float* disp, const int* n_dev, int n_max,
                      const SvoMat4& M, float* kept_xy, float* xyz, int* kept_index, int* n_kept,
                      const SvoPublish* pub = nullptr);
// a7 (sparse) + a8 in one launch: disparities at the n features, triangulation / compaction by the last workgroup; the
// outputs may be pinned host memory, `*pub_out` is the completion word to wait for (svo_wait_word)
int svo_k_stereo_triangulate(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width, int height, int row_stride,
                             int num_disparities, int block_size, const float* xy, const int* n_dev, int n_max, float* disp,
                             const SvoMat4& M, float* kept_xy, float* xyz, int* kept_index, int* n_kept, int word, SvoPublish* pub_out);
// a6
int svo_k_dedup(svo_ctx* ctx, const float* det_xy, const int* n_det_dev, int n_det_max, const float* trk_xy,
                const int* n_trk_dev, int n_trk_max, float min_distance, float* kept_xy, int* n_kept);
// a3: pyramid of `batch` images. levels buffer: per image svo_k_pyramid_bytes(w,h) bytes, level 0 first.
size_t svo_k_pyramid_bytes(int w, int h);
int svo_k_build_pyramid(svo_ctx* ctx, const uint8_t* imgs, int batch, int w, int h, int row_stride,
                        size_t image_stride, uint8_t* pyr, size_t pyr_stride, bool level0_in_place = false);
int svo_k_pyramid_level0(svo_ctx* ctx, const uint8_t* imgs, int batch, int w, int h, int row_stride, size_t image_stride, uint8_t* pyr,
                         size_t pyr_stride, hipStream_t st);
// forward+backward LK and the survivor filter of FeatureTracker::track_features.
// optional extras of the compaction step: per-feature state carried along with the kept features, and pinned
// host words that receive (n_kept, av_parallax) directly
struct SvoTrackCarry {
  const float* init_src = nullptr; const long long* ids_src = nullptr;
  float* init_dst = nullptr; long long* ids_dst = nullptr;
  int* host_n = nullptr; float* host_av = nullptr;           // pinned words for (n_kept, av_parallax)
  float* host_xy = nullptr; long long* host_ids = nullptr;   // pinned mirrors of the kept features / ids
  SvoPublish pub;                                            // completion word published after all of the above
};
int svo_k_track(svo_ctx* ctx, const uint8_t* pyr_prev, const uint8_t* pyr_next, int w, int h, const float* xy,
                const float* initial_xy, const int* n_dev, int n_max, float* fwd_xy, uint8_t* keep_flag,
                float* parallax, float* kept_xy, int* kept_index, int* n_kept, float* av_parallax,
                const SvoTrackCarry* carry = nullptr);
int svo_k_lk(svo_ctx* ctx, const uint8_t* pyr_prev, const uint8_t* pyr_next, int w, int h, const float* xy,
             int n, float* out_xy, uint8_t* status);
// gathers used by the tracker / pipeline: dst[i] = src[idx[i]] for i < n (n on the device or host)
int svo_k_gather_track(svo_ctx* ctx, const int* idx, const int* n_dev, int n_max, const float* init_src,
                       const long long* ids_src, float* init_dst, long long* ids_dst);
// tracker (re)initialisation from pinned host arrays: d_xy = d_init = h_xy, d_ids = h_ids (one launch, no H2D blits)
int svo_k_tracker_init(svo_ctx* ctx, const float* h_xy, const float* h_init, const long long* h_ids, int n, float* d_xy, float* d_init,
                       long long* d_ids, const SvoPublish* pub);
int svo_k_gather_xy_ids(svo_ctx* ctx, const int* idx, int n, const float* xy_src, const long long* ids_src,
                        float* xy_dst, long long* ids_dst);
// a5 (device-pointer form)
int svo_k_pnp(svo_ctx* ctx, SvoScratch& s, const float* d_xyz, const float* d_xy, int n, float focal, float cx, float cy,
              double* rvec3, double* tvec3, int iterations, float reproj_err, double confidence, int* d_inliers,
              int* n_inliers, int* h_inliers = nullptr /* pinned: also receives the inlier list */,
              float* d_inlier_xy = nullptr /* device: xy of the inliers, in list order (the dedup stage's input) */);
#endif
