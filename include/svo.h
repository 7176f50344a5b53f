/*
 * svo.h — C-ABI drop-in boundary of the MI355X-native stereo-VO hot path.
 *
 * Every entry point replaces one call the reference's ImageProcessor /
 * FeatureTracker / BundleAdjuster make into OpenCV / Ceres (reference
 * file:line is cited on each declaration; paths are relative to the
 * reference repository root).  Plain pointers and sizes only: no C++ types,
 * no torch types.  All functions return 0 (SVO_OK) on success and a negative
 * svo_status otherwise; nothing ever throws across this boundary.  The
 * reference's own error convention is "void + early return" (SURVEY §8b), so a
 * C++ adapter maps a non-zero status to "skip this frame".
 *
 * Memory spaces: functions without a suffix take caller-owned HOST pointers
 * (the boundary the reference classes would bind); functions ending in `_dev`
 * take DEVICE pointers (HBM resident; what the in-library pipeline and
 * bench.py use) and are asynchronous on svo_stream(ctx) unless stated.
 *
 * Threading: one caller thread per context (the reference's vo_node is
 * single-threaded, src/vo_node.cpp:139-227).
 */
#ifndef SVO_H_
#define SVO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum svo_status {
  SVO_OK = 0,
  SVO_ERR_INVALID = -1,   /* bad argument / shape */
  SVO_ERR_HIP = -2,       /* a HIP runtime call failed (see svo_last_error) */
  SVO_ERR_CAPACITY = -3,  /* a workspace bound given at svo_create was exceeded */
  SVO_ERR_NO_DEVICE = -4, /* no gfx950 device visible */
  SVO_ERR_NUMERIC = -5    /* singular system / non-finite value */
} svo_status;

/* Mirrors struct CameraInfo, src/camera_info.hpp:4-18 (same field order, so a
 * reference CameraInfo can be reinterpret_cast). k1..p2 are never read. */
typedef struct svo_camera_info {
  double focal, cx, cy;
  double k1, k2, p1, p2;
  double baseline;
} svo_camera_info;

typedef struct svo_ctx svo_ctx;

/* Limits fixed at context creation (workspace is allocated once; no
 * allocation happens on the hot path). */
typedef struct svo_limits {
  int max_width, max_height; /* largest image */
  int max_batch;             /* frames per batched front-end call */
  int max_corners;           /* corners returned per image (ref: 300, src/image_processor.cpp:22) */
  int max_candidates;        /* NMS survivors per image before min-distance selection */
  int max_features;          /* tracked features per frame (ref: 400, src/bundle_adjuster.hpp:75) */
} svo_limits;

/* Environment knobs read at creation (deployment tuning, no effect on results):
 *   SVO_BA_CU_SHARE=n   when several stereo streams share one GPU: window-sized bundle adjusters (svo_ba_create with
 *                       max_observations <= 100000) launch on their own n compute units of every 32 and the
 *                       context's stream on the remaining ones (HIP CU-masked streams).  n = 8 is one shader engine
 *                       per XCD on MI355X; measured +12 % frames/s at 8 streams per GPU, -2 % with a single stream.
 *                       Unset / 0: both use the whole GPU. */
int svo_create(svo_ctx** out, int device, const svo_limits* limits);
void svo_destroy(svo_ctx* ctx);
const char* svo_last_error(const svo_ctx* ctx);
/* hipStream_t the context launches on (as void*). */
void* svo_stream(svo_ctx* ctx);
int svo_sync(svo_ctx* ctx);
/* library build tag; "gfx950" must appear in it. */
const char* svo_version(void);

/* The reference's first-party literals as compiled into the kernels and the host chain (csrc/ref_constants.h), for
 * audit: tests check them against values extracted from the reference's source text (tests/golden/constants_golden.json).
 * All doubles; names follow that fixture.  Reference lines: src/image_processor.cpp:22,23,63,80,174,176,194,
 * src/feature_tracker.cpp:24-26,47,53,81, src/vo_node.cpp:33-36, src/bundle_adjuster.hpp:75, src/bundle_adjuster.cpp:11-12. */
typedef struct svo_reference_constants_t {
  double gftt_max_corners, gftt_quality, min_detected, keyframe_percent_lost;
  double pnp_iterations, pnp_reproj_error, pnp_confidence;
  double stereo_num_disparities, stereo_block_size, stereo_disparity_scale, triangulate_min_disparity_exclusive;
  double lk_win_w, lk_win_h, lk_max_level, lk_max_iterations, lk_epsilon, lk_min_eig_threshold;
  double fb_max_distance, max_parallax, draw_thickness;
  double parallax_thresh, min_feature_distance, sliding_window_size, max_features;
  double ba_max_solver_time_s, ba_num_threads;
} svo_reference_constants_t;
int svo_reference_constants(svo_reference_constants_t* out);

/* Measurement aid (not a reference interface): time every launch of ONE named kernel with HIP events
 * recorded on the context stream.  kernel: "corner_response", "corner_nms", "corner_select", "pyr_down",
 * "lk_fb", "stereo_at", "triangulate", "pnp_hypotheses", "pnp_refine", "ba_linearize", "ba_backsub", "ba_step";
 * NULL/"" disables.  svo_profile_read synchronises the stream and returns the summed duration and the
 * launch count since the last svo_profile_select. */
int svo_profile_select(svo_ctx* ctx, const char* kernel);
/* Measurement aid: the card's own ceilings for the roofline objects (SURVEY 8d asks for a measured FP64 figure).
 * what: "f64_fma" (vector FMA), "f64_muladd" (separate multiply + add: what the parity-exact kernels issue),
 * "f64_mfma" (v_mfma_f64_16x16x4_f64), "hbm_copy" (512 MiB device copy).  *value: flop/s or bytes/s (read + write). */
int svo_measure_peak(svo_ctx* ctx, const char* what, double* value);
int svo_profile_read(svo_ctx* ctx, double* total_ms, int* launches);

/* ------------------------------------------------------------------ a11 --
 * Batched ReprojectionFactor::Evaluate (src/reprojection_factor.cpp:10-88).
 * pose7 = [qw qx qy qz tx ty tz] (src/bundle_adjuster.hpp:50), n of them;
 * point3, obs2 likewise.  r2: n x 2.  jpose14: n x (2x7 row-major) or NULL;
 * jpoint6: n x (2x3 row-major) or NULL (Ceres' null conventions,
 * src/reprojection_factor.cpp:58-59,77).  Entries 5 and 11 of each 2x7 are 0
 * (src/reprojection_factor.cpp:61). */
int svo_reproj_eval(svo_ctx* ctx, int n, const double* pose7, const double* point3,
                    const double* obs2, double focal, double cx, double cy,
                    double* r2, double* jpose14, double* jpoint6);
int svo_reproj_eval_dev(svo_ctx* ctx, int n, const double* pose7, const double* point3,
                        const double* obs2, double focal, double cx, double cy,
                        double* r2, double* jpose14, double* jpoint6);

/* ------------------------------------------------------------------- a1 --
 * cv::goodFeaturesToTrack(img, out, max_corners, quality, min_distance)
 * with defaults blockSize=3, useHarris=false, no mask
 * (call site src/image_processor.cpp:22; semantics SURVEY Appendix A.1).
 * xy: max_corners x 2 floats (integer-valued pixel coords, strongest first);
 * *n receives the count.  Batched form: images are `image_stride` bytes
 * apart, rows `row_stride` bytes apart; xy is batch x max_corners x 2,
 * n is batch ints. */
int svo_corner_detect(svo_ctx* ctx, const uint8_t* img, int width, int height, int row_stride,
                      int max_corners, double quality, double min_distance,
                      float* xy, int* n);
int svo_corner_detect_batch_dev(svo_ctx* ctx, const uint8_t* imgs, int batch, int width, int height,
                                int row_stride, size_t image_stride, int max_corners,
                                double quality, double min_distance, float* xy, int* n);
/* Debug/parity taps of the same path: the min-eigenvalue map (float, width*height)
 * produced by the response kernel. Host pointers. */
int svo_corner_response(svo_ctx* ctx, const uint8_t* img, int width, int height, int row_stride,
                        float* eig);

/* ------------------------------------------------------------------- a7 --
 * cv::StereoBM::create(num_disp, block)->compute(L,R) + convertTo(CV_32F,1/16)
 * (src/image_processor.cpp:173-176; defaults XSOBEL cap 31,
 * minDisparity 0, textureThreshold 10, uniquenessRatio 15).
 * Dense form writes the CV_16S map (4 fractional bits, FILTERED = -16).
 * Sparse form evaluates exactly the same function only at (int)y,(int)x of
 * each point (what src/image_processor.cpp:193 samples) and returns the
 * float disparity (-1.0 = filtered). */
int svo_stereo_bm(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width, int height,
                  int row_stride, int num_disparities, int block_size, int16_t* disp16);
int svo_stereo_disparity_at(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width,
                            int height, int row_stride, int num_disparities, int block_size,
                            const float* xy, int n, float* disp);
int svo_stereo_disparity_at_dev(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width,
                                int height, int row_stride, int num_disparities, int block_size,
                                const float* xy, const int* n_dev, int n_max, float* disp);

/* ------------------------------------------------------------------- a8 --
 * ImageProcessor::triangulate_stereo's reprojection loop
 * (src/image_processor.cpp:178-207): keep i iff disp[i] > 0; X = pose * Q * [x y d 1]^T,
 * de-homogenised.  pose16: row-major 4x4 float camera->world.  Output order =
 * input order (stable).  kept_xy: n x 2, xyz: n x 3, kept_index: n (index into the
 * input list) or NULL. */
int svo_triangulate(svo_ctx* ctx, const float* xy, const float* disp, int n, const float* pose16,
                    float focal, float cx, float cy, float baseline,
                    float* kept_xy, float* xyz, int* kept_index, int* n_kept);

/* ------------------------------------------------------------------- a3 --
 * cv::calcOpticalFlowPyrLK(prev, next, pts, out, status, err, Size(21,21), 3,
 * TermCriteria(COUNT+EPS,30,0.01), 0, 1e-2)
 * (src/feature_tracker.cpp:23-26).  out: n x 2, status: n bytes. */
int svo_lk_track(svo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width, int height,
                 int row_stride, const float* xy, int n, float* out_xy, uint8_t* status);
/* The 4-level pyramid the tracker uses (pyrDown chain), for parity taps.
 * levels: concatenated level images, level l has size ((w+2^l-1)>>l) x ((h+2^l-1)>>l), tight rows. */
int svo_build_pyramid(svo_ctx* ctx, const uint8_t* img, int width, int height, int row_stride,
                      uint8_t* levels, size_t levels_bytes);

/* FeatureTracker::track_features (src/feature_tracker.cpp:18-67) as one call:
 * forward + backward LK and the survivor / parallax filter.
 * initial_xy[i] is the keyframe position of feature i (initial_features.at(id)).
 * Outputs: kept_xy (n x 2), kept_index (n; index into the input list, ascending),
 * *n_kept, *av_parallax (sum over kept / n, src/feature_tracker.cpp:59,63),
 * *percent_lost is left to the caller (needs |initial_features|, :64). */
int svo_track_features(svo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width,
                       int height, int row_stride, const float* xy, const float* initial_xy, int n,
                       float* kept_xy, int* kept_index, int* n_kept, float* av_parallax);

/* ------------------------------------------------------------------- a6 --
 * new-vs-tracked dedup loop (src/image_processor.cpp:113-128): keep detected[i]
 * iff no tracked[j] has sqrt(dx^2+dy^2) < min_distance. Stable order. */
int svo_dedup(svo_ctx* ctx, const float* detected_xy, int n_detected, const float* tracked_xy,
              int n_tracked, float min_distance, float* kept_xy, int* n_kept);

/* ------------------------------------------------------------------- a5 --
 * cv::solvePnPRansac(obj, img, K, 0, rvec, tvec, true, iters, reproj_err, conf, inliers)
 * (src/image_processor.cpp:76-80).  Deterministic restatement:
 * fixed-seed hypothesis sampling, Gauss-Newton minimal solves from the extrinsic
 * guess, inlier test err^2 <= reproj_err^2, refinement on the inliers.
 * rvec3/tvec3 are in/out doubles; inliers: n ints (ascending), *n_inliers. */
int svo_pnp_ransac(svo_ctx* ctx, const float* xyz, const float* xy, int n, float focal, float cx,
                   float cy, double* rvec3, double* tvec3, int iterations, float reproj_err,
                   double confidence, int* inliers, int* n_inliers);

/* ------------------------------------------------------------ a9,a10,a12,a13 --
 * Sliding-window bundle adjustment (src/bundle_adjuster.cpp:5-163).
 * The graph lives on the device; ids are assigned exactly as the reference does
 * (sequential in creation order, SURVEY C-3). */
typedef struct svo_ba svo_ba;

typedef struct svo_ba_options {
  int max_iterations;      /* Ceres default 50 */
  double max_time_s;       /* reference: 0.1 (src/bundle_adjuster.cpp:11); <=0 disables (parity runs) */
  double function_tolerance, gradient_tolerance, parameter_tolerance; /* 1e-6, 1e-10, 1e-8 */
  double initial_radius;   /* 1e4 */
  int max_features;        /* per keyframe; reference 400 (src/bundle_adjuster.hpp:75) */
  int accumulation;        /* how J^T J / the Schur products are summed: SVO_BA_ACC_* (default AUTO) */
} svo_ba_options;

/* AUTO: DETERMINISTIC (per-chunk partial sums in the declared summation order: bit-identical to the oracle) while the
 * partial store (wire elements x chunk groups x 16 bytes) fits 512 MB, else MFMA when eligible (<= 22 poses, one observation per (landmark, pose)), else ATOMICS. */
enum { SVO_BA_ACC_AUTO = 0, SVO_BA_ACC_DETERMINISTIC = 1, SVO_BA_ACC_ATOMICS = 2, SVO_BA_ACC_MFMA = 3 };

typedef struct svo_ba_summary {
  int iterations, successful_steps, termination; /* 0 conv, 1 no-conv(iter/time), 2 failure */
  double initial_cost, final_cost;
  double solve_ms;
} svo_ba_summary;

/* Sums `n_doubles` doubles at DEVICE pointer `dev_ptr` in place over all ranks (in-process emulation of the collective
 * for tests; production ranks hand over an RCCL communicator with svo_ba_set_comm instead). */
typedef int (*svo_allreduce_fn)(void* dev_ptr, size_t n_doubles, void* user);

/* ---- step control of ceres::Solve (src/bundle_adjuster.cpp:140) with pluggable passes ------------------------------
 * The LM loop itself (host/lm.cpp) is one piece of host code shared by every backend and every rank.  A backend
 * provides the two passes over the observations; payloads are host arrays ALREADY SUMMED over all ranks:
 *   payload1 (n*n + 3n + 2 doubles, n = 6 (K-1)): [S (n x n, full) | g_red | g_c | diag U | cost | sum g_p^2]
 *   payload2 (4 doubles): [candidate cost | landmark part of the model change | sum dp^2 | sum p^2]
 * linearize: pass A at the current point with `radius`; `first` != 0 fixes the landmarks' Jacobi scales.
 * step:      pass B at the current point: pose step dc (n) and candidate poses (7K) in, candidate landmarks formed,
 *            payload2 out.  The backend may also produce the NEXT iteration's payload1 in the same call, so that an LM
 *            iteration costs one host round trip (see host/lm.cpp):
 *              ctl->spec_radius > 0: pass A at the candidate with that radius in the same sweep; both payloads are
 *                summed by ONE collective;
 *              else ctl->chain != 0: once payload2 is summed, take Ceres' accept / radius decision where the sums live
 *                (svo_lm_decide, host/lm_decide.h, from ctl->cost, ctl->mcc, radius, ctl->decrease_factor) and run pass
 *                A for it: at the candidate with the new radius if accepted, at the current point with the reduced
 *                radius if not.
 *            *next_radius = the radius that pass A ran with (0: none was produced), *next_at_candidate = where.
 *            The step control re-derives the decision itself and uses payload1_next only if both agree.
 * accept:    the candidate becomes the current point.
 * Callbacks return 0 or an svo_status. */
typedef struct svo_lm_step_ctl {
  double cost, mcc, decrease_factor;
  double spec_radius;
  int chain;
} svo_lm_step_ctl;
typedef struct svo_lm_ops {
  void* user;
  int (*linearize)(void* user, double radius, int first, double* payload1);
  int (*step)(void* user, const double* dc, const double* cand_poses7, double radius, const svo_lm_step_ctl* ctl,
              double* payload2, double* payload1_next, double* next_radius, int* next_at_candidate);
  int (*accept)(void* user);
} svo_lm_ops;
typedef struct svo_lm_stats {
  int linearize_calls;   /* stand-alone pass-A calls (each one exchange) */
  int step_calls;        /* pass-B calls (each one exchange) */
  int speculations;      /* steps that also produced a linearisation for the next iteration (same sweep or chained) */
  int speculation_hits;  /* ... that the step control could use: the iteration cost exactly one host round trip */
  int single_exchange;   /* steps whose pass A rode in the SAME collective as payload2 (saturated-radius prediction) */
  int collectives;       /* all-reduce calls the solve issued (sharded runs; what N ranks would issue: also counted on one rank) */
  int device_control;    /* 1: the step control ran on the device (bulk / sharded path: no host work inside an LM iteration) */
  int fallbacks;         /* device-resident window solves that gave up and were re-run through the host-driven loop */
  double host_us;        /* host time spent inside the LM loop of the last solve (launch calls + waits that were not overlapped), us */
} svo_lm_stats;

void svo_ba_default_options(svo_ba_options* o);
/* The LM loop over caller-provided passes.  poses7: K x 7 current poses, updated in place on every accepted step
 * (pose 0 is constant, src/bundle_adjuster.cpp:130).  opt NULL = defaults.  stats may be NULL. */
int svo_lm_solve(int n_poses, double* poses7, const svo_lm_ops* ops, const svo_ba_options* opt,
                 svo_ba_summary* summary, svo_lm_stats* stats);
/* Ceres' accept / radius rule for one step (host/lm_decide.h; the kernels evaluate the same function): what a backend's
 * `step` uses for ctl->chain when its sums live on the host. */
int svo_lm_decide_step(double cost, double mcc, double radius, double decrease_factor, double cost_new,
                       double model_change_points, int* accept, double* next_radius);
int svo_ba_create(svo_ctx* ctx, svo_ba** out, int window_size, const svo_camera_info* cam,
                  const svo_ba_options* opt, int max_landmarks, int max_observations);
void svo_ba_destroy(svo_ba* ba);
/* forget all keyframes and landmarks, keep every buffer (a fresh BundleAdjuster without re-allocation). */
int svo_ba_reset(svo_ba* ba);
/* BundleAdjuster::add_keyframe (src/bundle_adjuster.cpp:60-135). pose7 from the
 * keyframe's (orientation, position) floats widened to double (:63-70).
 * tracked_ids/tracked_xy: n_tracked observations of existing landmarks;
 * new_xy/new_xyz: n_new fresh landmarks, truncated to max_features-n_tracked
 * (:85-90); new_ids receives the ids assigned, *n_new_out the surviving count. */
int svo_ba_add_keyframe(svo_ba* ba, const double* pose7, const int64_t* tracked_ids,
                        const float* tracked_xy, int n_tracked, const float* new_xy,
                        const float* new_xyz, int n_new, int64_t* new_ids, int* n_new_out);
/* BundleAdjuster::bundle_adjust (src/bundle_adjuster.cpp:137-157): no-op unless a
 * keyframe was added since the last solve. */
int svo_ba_solve(svo_ba* ba, svo_ba_summary* summary);
/* The dense SPD solve of the reduced camera system — what Ceres' DENSE_SCHUR does with Eigen's LLT inside
 * ceres::Solve (src/bundle_adjuster.cpp:156).  Host-only (the system is at most 378 x 378).  A: n x n
 * row-major, lower triangle read, overwritten by L; b: right-hand side, overwritten by the solution.
 * Declared operation order (see host/linalg.cpp): bit-identical to the plain left-looking loop.
 * Returns SVO_ERR_NUMERIC when A is not positive definite. */
int svo_cholesky_solve(double* A, double* b, int n);
/* The same solve by ONE workgroup on the GPU (csrc/lm_device.h): what the controller workgroup of the device-resident
 * solve runs between two passes, exposed for parity tests — bit-identical to svo_cholesky_solve.  Host pointers. */
int svo_cholesky_solve_dev(svo_ctx* ctx, double* A, double* b, int n);
/* pose of window slot k (0 = oldest, -1 = newest). */
int svo_ba_get_pose(svo_ba* ba, int k, double* pose7);
int svo_ba_window_count(svo_ba* ba);
/* BundleAdjuster::get_world_points (src/bundle_adjuster.cpp:159-163): double->float gather. */
int svo_ba_get_points(svo_ba* ba, const int64_t* ids, int n, float* xyz);

/* Bulk problem interface (synthetic BA of BASELINE config 4; also what a
 * sharded rank loads): poses K x 7 (pose 0 constant), points N x 3, observations
 * sorted by landmark: obs_pose/obs_point/obs_uv.  In a sharded run every rank
 * loads all poses and only its own landmarks; the payloads are summed over the ranks by
 * RCCL (svo_ba_set_comm) or by the `allreduce` callback (tests), once per LM iteration
 * when the speculative linearisation hits (host/lm.cpp), twice otherwise. */
int svo_ba_load_problem(svo_ba* ba, int n_poses, const double* poses7, int n_points,
                        const double* points3, int n_obs, const int32_t* obs_pose,
                        const int32_t* obs_point, const double* obs_uv);
int svo_ba_set_allreduce(svo_ba* ba, svo_allreduce_fn fn, void* user);
/* Sharded run over RCCL: `nccl_comm` is the rank's ncclComm_t (as void*).  The library calls
 * ncclAllReduce(sum, f64, in place) on the adjuster's own stream — one call per LM iteration when the speculative
 * linearisation hits (n*n + 3n + 2 + 8 doubles: 107 KB at K = 20), no host code in between.  NULL detaches. */
int svo_ba_set_comm(svo_ba* ba, void* nccl_comm);
/* RCCL communicator helpers for callers that have no other RCCL binding (C++ consumers such as the reference's
 * vo_node; bench.py).  They use the librccl already loaded in the process, else /opt/rocm/lib/librccl.so.1.
 * id128: 128-byte ncclUniqueId created on one rank and distributed by any out-of-band channel. */
int svo_rccl_unique_id(void* id128);
int svo_rccl_comm_create(void** nccl_comm, int n_ranks, int rank, const void* id128, int device);
int svo_rccl_comm_destroy(void* nccl_comm);
/* Where ceres::Solve's step control runs for window-sized, single-rank, deterministic solves (src/bundle_adjuster.cpp:140):
 * mode 1: on the device — the whole solve is ONE launch (csrc/ba.hip ba_lm_kernel), the host only waits for its completion
 * word; mode 0: on the host (host/lm.cpp), 1-3 launches per LM iteration; mode -1 (default): on the device while more
 * than two pipelines are inside svo_pipeline_process_batch* (pipeline groups always solve on the device).  Results are
 * bit-identical either way. */
int svo_ba_set_device_lm(svo_ba* ba, int mode);
/* Which form a device-resident window solve takes (both run src/bundle_adjuster.cpp:140's whole ceres::Solve on the device,
 * bit-identical to each other and to the host-driven loop):
 *   0 wide     ba_lm_kernel: one workgroup per two chunks of 64 observations (~47 for the reference's 5-keyframe window), replicated
 *              step control, lowest latency — a lone stereo stream (the reference's vo_node);
 *   1 compact  ba_lm_compact_kernel: ONE workgroup per solve, its wavefronts take the chunks in turn, step control once, nothing
 *              waits for another workgroup — 1/16 of the wide form's wavefronts and LDS for ~10x its latency (measured: it does not
 *              pay at 48-128 streams on one MI355X); the re-run of a wide solve that gave up, the overflow of the
 *              admission budget (SVO_BA_OVERFLOW=1); takes windows of up to 256 chunks (the wide form: 128);
 *  -1 default  SVO_BA_FORM=wide|compact if set, else wide. */
int svo_ba_set_solve_form(svo_ba* ba, int form);
/* Where the step control runs for BULK / SHARDED solves (hardware-order accumulation, svo_ba_load_problem +
 * svo_ba_solve_problem; the all-reduce of src/bundle_adjuster.cpp:140's normal equations over the ranks): mode 1 / -1 (default):
 * on the device — per LM iteration the host only enqueues [pass B, all-reduce, pass A, all-reduce, control kernel], a fixed
 * number of slots ahead of the control kernel's status records (csrc/ba.hip ba_bulk_control_kernel; reduced camera systems up
 * to n = 128); mode 0: host/lm.cpp drives every iteration (one D2H + host Cholesky + upload per iteration). */
int svo_ba_set_bulk_control(svo_ba* ba, int mode);
/* counters of the last svo_ba_solve / svo_ba_solve_problem */
int svo_ba_last_stats(svo_ba* ba, svo_lm_stats* stats);
int svo_ba_solve_problem(svo_ba* ba, svo_ba_summary* summary);
int svo_ba_read_problem(svo_ba* ba, double* poses7, double* points3);

/* ---------------------------------------------------------------- pipeline --
 * ImageProcessor::process + BundleAdjuster::bundle_adjust for a batch of
 * consecutive frames already resident in HBM (src/image_processor.cpp:18-163,
 * driver loop src/vo_node.cpp:141-148).  Stateless stages (a1, pyramids) are
 * batched over the frames; the sequential chain runs per frame.  Results are
 * identical to frame-by-frame processing. */
typedef struct svo_pipeline svo_pipeline;
typedef struct svo_pipeline_params {
  svo_camera_info cam;
  int width, height;
  int max_corners;            /* 300 */
  double quality;             /* 0.1 */
  float min_feature_distance; /* 30 (src/vo_node.cpp:34) */
  float parallax_thresh;      /* 20 (src/vo_node.cpp:33) */
  int window_size;            /* 5  (src/vo_node.cpp:36) */
  int max_features;           /* 400 */
  int ba_max_iterations;      /* 50 */
  double ba_max_time_s;       /* 0.1; <=0 disables */
} svo_pipeline_params;

typedef struct svo_frame_result {
  int n_detected, n_tracked, n_inliers, n_new;
  int is_keyframe;       /* 1 if this frame became a keyframe */
  float av_parallax, percent_lost;
  double pose7[7];       /* last keyframe pose (world wrt camera) after bundle_adjust */
  int ba_iterations;
} svo_frame_result;

void svo_pipeline_default_params(svo_pipeline_params* p);
int svo_pipeline_create(svo_ctx* ctx, svo_pipeline** out, const svo_pipeline_params* p);
void svo_pipeline_destroy(svo_pipeline* p);
int svo_pipeline_reset(svo_pipeline* p);
/* left/right: batch images, tight rows (row stride = width), image stride = width*height. */
int svo_pipeline_process_batch_dev(svo_pipeline* p, const uint8_t* left, const uint8_t* right,
                                   int batch, svo_frame_result* results);
int svo_pipeline_process_batch(svo_pipeline* p, const uint8_t* left, const uint8_t* right,
                               int batch, svo_frame_result* results);
/* Feature-set taps for parity tests: ids + positions the tracker holds after the last frame. */
int svo_pipeline_get_tracked(svo_pipeline* p, int64_t* ids, float* xy, int capacity, int* n);

/* ------------------------------------------------------------ pipeline group --
 * The same per-frame path (ImageProcessor::process + BundleAdjuster::bundle_adjust, src/image_processor.cpp:18-163,
 * src/vo_node.cpp:141-148) for n_lanes independent stereo streams behind ONE caller thread: each lane keeps its own
 * tracker, graph and poses exactly as an svo_pipeline does, every stage that several lanes reach together is ONE kernel
 * launch (blockIdx.y = lane) and their bundle adjustments are ONE device-resident solve launch; lanes never wait for
 * each other.  Lane results are bit-identical to n_lanes separate svo_pipeline objects.  The context's
 * svo_limits.max_batch bounds n_lanes x frames per call; 1 <= n_lanes <= 64.  (No reference counterpart: the reference is one stream in one
 * process; this is how one GPU serves many of them — and what a rank of the multi-GPU bench runs.) */
typedef struct svo_pipeline_group svo_pipeline_group;
int svo_pipeline_group_create(svo_ctx* ctx, svo_pipeline_group** out, const svo_pipeline_params* p, int n_lanes);
void svo_pipeline_group_destroy(svo_pipeline_group* g);
int svo_pipeline_group_reset(svo_pipeline_group* g);
int svo_pipeline_group_lanes(const svo_pipeline_group* g);
/* left/right: DEVICE pointers; lane l's `batch` images (tight rows, image stride = width*height) start lane_stride bytes
 * after lane l-1's.  results: n_lanes x batch, lane-major. */
int svo_pipeline_group_process_batch_dev(svo_pipeline_group* g, const uint8_t* left, const uint8_t* right, size_t lane_stride,
                                         int batch, svo_frame_result* results);
/* Host-pointer / streaming entry.  The reference hands over HOST images frame by frame (cv::Mat copies made in the image
 * callback, src/vo_node.cpp:70-73, consumed at :141-143): a group takes them through TWO pinned staging slots it owns.
 *   svo_pipeline_group_staging   the slot's buffers: n_lanes x max_batch x height x width bytes each, lane_stride =
 *                                max_batch x width x height.  The caller writes the next batch there in place (the
 *                                callback's copy lands where the DMA engine reads it: no second host copy);
 *   svo_pipeline_group_upload    starts the H2D copy of the slot's first `batch` frames per lane on the group's copy stream
 *                                and returns at once: the upload of batch b+1 overlaps the processing of batch b;
 *   svo_pipeline_group_process_uploaded  waits for the slot's upload, then processes it exactly as
 *                                svo_pipeline_group_process_batch_dev does (bit-identical results);
 *   svo_pipeline_group_process_batch     convenience, nothing overlapped: caller-owned host images are copied into slot 0,
 *                                uploaded and processed (what ImageProcessor::process(const StereoPair&) does, for every lane).
 * Streaming loop: fill(0); upload(0); for b: { fill((b+1)&1); upload((b+1)&1); process_uploaded(b&1); }. */
int svo_pipeline_group_staging(svo_pipeline_group* g, int slot, uint8_t** left, uint8_t** right, size_t* lane_stride);
int svo_pipeline_group_upload(svo_pipeline_group* g, int slot, int batch);
int svo_pipeline_group_process_uploaded(svo_pipeline_group* g, int slot, svo_frame_result* results);
int svo_pipeline_group_process_batch(svo_pipeline_group* g, const uint8_t* left, const uint8_t* right, size_t lane_stride,
                                     int batch, svo_frame_result* results);
int svo_pipeline_group_get_tracked(svo_pipeline_group* g, int lane, int64_t* ids, float* xy, int capacity, int* n);
/* Launch statistics of the last process_batch call, by stage: 0 track (LK + compaction), 1 PnP-RANSAC (hypotheses, bookkeeping and
 * refinement in one launch), 3 dedup / stereo + triangulation, 4 bundle-adjustment solves, 5 corner detection + pyramids;
 * launches6[i] launches carried lanes6[i] lane-stages in total.  Slot 2 (round 5) is the driving thread instead: launches6[2] = microseconds of
 * its loop passes that did something, lanes6[2] = microseconds of the call's main loop. */
int svo_pipeline_group_last_stats(const svo_pipeline_group* g, long* launches6, long* lanes6);
/* Algorithmic work (SURVEY 8(d) per-iteration figures: 466 flops per observation + 50 + 144 L + 216 L (L + 1) / 2 per landmark;
 * 24 B per observation + 48 B per landmark + 56 B per pose) of the bundle adjustments all lanes have finished since the last
 * reset: out4 = [f64 flops, bytes, solves, LM iterations].  Measurement aid (bench.py's roofline of the solve kernel). */
int svo_pipeline_group_solve_work(svo_pipeline_group* g, double* out4, int reset);

/* FeatureTracker::draw_track + get_drawing (src/feature_tracker.cpp:74-91; used by src/vo_node.cpp:137,188):
 * the keyframe image as RGB (3 bytes per pixel, width*height*3 output) with one green arrow of thickness 4 per feature
 * from its keyframe position to its current position.  Host-side visualisation with this repository's own rasteriser
 * (same picture as cv::arrowedLine, not pixel-identical). */
int svo_draw_track(const uint8_t* gray, int width, int height, int row_stride, const float* from_xy,
                   const float* to_xy, int n, uint8_t* rgb);
/* The same for a pipeline's tracker: `keyframe_gray` is the host copy of the image the tracker was (re)initialised
 * on (the reference keeps a clone, :14); arrows come from the tracker's initial / current feature positions. */
int svo_pipeline_draw_track(svo_pipeline* p, const uint8_t* keyframe_gray, int row_stride, uint8_t* rgb);

/* ------------------------------------------------------------- synthetic data --
 * Deterministic KITTI-shaped stereo stream: integer PRNG, ray-cast
 * textured billboards; host buffers; bit-identical on every host. Not part of the
 * reference; test/bench input only. */
typedef struct svo_synth_params {
  uint64_t seed;
  int width, height;
  double focal, cx, cy, baseline;
  double step_z, step_x, yaw_per_frame; /* camera motion per frame */
  int n_billboards;
} svo_synth_params;
void svo_synth_default_params(svo_synth_params* p, int width, int height);
int svo_synth_render(const svo_synth_params* p, int frame, uint8_t* left, uint8_t* right);
/* ground-truth camera-in-world pose of `frame` as 3x4 row-major [R|t] (KITTI poses row format,
 * src/kitti_node.cpp:47-50). */
int svo_synth_pose(const svo_synth_params* p, int frame, double* rt12);

/* ------------------------------------------------ dataset ingestion, driver, ATE --
 * KITTI odometry layout read by the reference's kitti_node (src/kitti_node.cpp:37-68): images
 * <data_path><SS>/image_0|image_1/%06d.png (8-bit gray), poses <data_path>data_odometry_poses/dataset/poses/SS.txt
 * (12 doubles per row, row-major 3x4 [R|t], camera in world).  PNG (zlib only) and PGM are decoded. */
int svo_image_read_gray(const char* path, uint8_t* buf, size_t capacity, int* width, int* height);
int svo_kitti_read_poses(const char* poses_file, double* rt12, int capacity_frames, int* n_frames);
/* RMSE of positions after the best rigid (with_scale: similarity) alignment est -> gt. */
int svo_ate_rmse(const double* est_xyz, const double* gt_xyz, int n, int with_scale, double* rmse);
typedef struct svo_run_stats {
  int frames, keyframes;
  double ate_rmse; /* vs the poses file, evaluated at keyframes, rigid alignment; -1 without ground truth */
  double seconds;
} svo_run_stats;
/* Non-ROS driver with vo_node's loop semantics (src/vo_node.cpp:141-150): process + bundle_adjust per frame,
 * one camera-in-world pose per processed frame written to traj_rt12 (max_frames x 12, may be NULL). */
int svo_kitti_run(svo_ctx* ctx, const svo_pipeline_params* params, const char* data_path, int sequence,
                  int max_frames, double* traj_rt12, svo_run_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* SVO_H_ */
