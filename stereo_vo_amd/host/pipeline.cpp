// Host side of the hot path: the reference's ImageProcessor / FeatureTracker / BundleAdjuster control flow
// re-hosted over the HIP kernels (everything heavy stays in HBM; the host only sees the handful of
// scalars the reference's own branches test) and the svo_pipeline_* C-ABI on top.
// Reference: src/image_processor.cpp:18-208, src/feature_tracker.cpp:3-72, src/bundle_adjuster.cpp:60-163,
// driver rule src/vo_node.cpp:141-148.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>

#include <atomic>
#include <chrono>
#include <unordered_map>

#include "det_trig.h"
#include "kernels.h"
#include "ref_constants.h"
#include "stereo_vo.hpp"

namespace svo {

// SVO_TIMING=1: wall-clock per phase of the host-side chain, summed over every pipeline of the process (stream threads
// and bundle-adjustment workers add concurrently: atomic nanosecond counters) and printed when a pipeline is destroyed.
// Scoped objects only, so an early return cannot leak one; a no-op unless the variable is set.
struct PhaseTimer {
  static std::atomic<uint64_t> acc_ns[8];
  static bool enabled() { static const bool on = getenv("SVO_TIMING") != nullptr; return on; }
  static const char* name(int i) {
    static const char* n[8] = {"prepare_batch", "track", "pnp", "dedup+stereo+triangulate", "add_keyframe+init", "bundle_adjust", "first_keyframe", "other"};
    return n[i];
  }
  int id;
  std::chrono::steady_clock::time_point t0;
  explicit PhaseTimer(int i) : id(i) { if (enabled()) t0 = std::chrono::steady_clock::now(); }
  void next(int i) { stop(); id = i; if (enabled()) t0 = std::chrono::steady_clock::now(); }  // close this phase, open phase i
  void stop() {
    if (id >= 0 && enabled())
      acc_ns[id].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                           std::memory_order_relaxed);
    id = -1;
  }
  ~PhaseTimer() { stop(); }
  PhaseTimer(const PhaseTimer&) = delete;
  PhaseTimer& operator=(const PhaseTimer&) = delete;
};
std::atomic<uint64_t> PhaseTimer::acc_ns[8];

#define SVO_TRY(expr)            \
  do {                           \
    hipError_t e__ = (expr);     \
    if (e__ != hipSuccess) {     \
      ctx_->err = std::string(#expr) + ": " + hipGetErrorString(e__); \
      return;                    \
    }                            \
  } while (0)

// ------------------------------------------------------------------------------------------ ReprojectionFactor
bool ReprojectionFactor::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
  double obs[2] = {ox_, oy_};
  return svo_reproj_eval(ctx_, 1, parameters[0], parameters[1], obs, info_.focal, info_.cx, info_.cy, residuals,
                         jacobians ? jacobians[0] : nullptr, jacobians ? jacobians[1] : nullptr) == SVO_OK;
}

// ------------------------------------------------------------------------------------------ BundleAdjuster
BundleAdjuster::BundleAdjuster(svo_ctx* ctx, size_t window_size, svo_camera_info info, int max_features, int max_iterations,
                               double max_time_s)
    : ctx_(ctx), window_size_(window_size), info_(info), max_features_(max_features), max_iterations_(max_iterations),
      max_time_s_(max_time_s) {
  reset();
}

std::atomic<int> g_pipelines{0};  // pipelines currently INSIDE svo_pipeline_process_batch*: stereo streams running at this moment
void pipeline_count_add(int delta) { g_pipelines.fetch_add(delta, std::memory_order_relaxed); }
unsigned spin_budget() {
  static const char* e = getenv("SVO_SPIN");  // override: pause iterations before sleeping
  if (e && *e) return (unsigned)atol(e);
  return g_pipelines.load(std::memory_order_relaxed) <= 2 ? 4000000u : 300u;  // ~0.1 s (never sleeps in practice) / ~10 us
}

}  // namespace svo
bool svo_throughput_mode() { return svo::g_pipelines.load(std::memory_order_relaxed) > 2; }
namespace svo {

void BundleAdjuster::wait() {
  // the solve in flight takes well under a millisecond: poll first, sleep if it does not show up within the budget
  const unsigned budget = spin_budget();
  unsigned spins = 0;
  while (job_state_.load(std::memory_order_acquire) != 0) {
    __builtin_ia32_pause();
    if (++spins > budget) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [this] { return job_state_.load(std::memory_order_acquire) == 0; });
      return;
    }
  }
}

void BundleAdjuster::worker_loop() {
  (void)hipSetDevice(ctx_->device);
  for (;;) {
    // wait for a job: short spin, then sleep
    const unsigned budget = spin_budget();
    unsigned spins = 0;
    while (job_state_.load(std::memory_order_acquire) != 1 && !quit_.load(std::memory_order_acquire)) {
      __builtin_ia32_pause();
      if (++spins > budget) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return job_state_.load(std::memory_order_acquire) == 1 || quit_.load(std::memory_order_acquire); });
        spins = 0;
      }
    }
    if (quit_.load(std::memory_order_acquire)) return;
    run_bundle_adjust();
    {
      std::lock_guard<std::mutex> lk(mu_);  // pairs with the predicate check of a sleeping waiter
      job_state_.store(0, std::memory_order_release);
    }
    cv_done_.notify_all();
  }
}

void BundleAdjuster::bundle_adjust_async() {
  wait();
  if (!launch_needed_) return;
  launch_needed_ = false;
  if (!worker_.joinable()) worker_ = std::thread([this]() { worker_loop(); });
  {
    std::lock_guard<std::mutex> lk(mu_);  // pairs with the predicate check of a sleeping worker
    job_state_.store(1, std::memory_order_release);
  }
  cv_.notify_one();
}

void BundleAdjuster::reset() {
  wait();
  last_keyframe_.reset();
  last_iterations_ = 0;
  new_frame_added_ = launch_needed_ = false;
  if (ba_) { svo_ba_reset(ba_); return; }  // keep the buffers
  svo_ba_options opt;
  svo_ba_default_options(&opt);
  opt.max_features = max_features_;       // src/bundle_adjuster.hpp:75
  opt.max_iterations = max_iterations_;
  opt.max_time_s = max_time_s_;           // src/bundle_adjuster.cpp:11
  const int max_lm = 1 << 20;             // landmark store grows monotonically (SURVEY C-3)
  const int max_obs = (int)(window_size_ + 1) * max_features_ + 64;
  if (svo_ba_create(ctx_, &ba_, (int)window_size_, &info_, &opt, max_lm > max_obs ? max_obs : max_lm, max_obs)) ba_ = nullptr;
  last_keyframe_.reset();
  last_iterations_ = 0;
}

BundleAdjuster::~BundleAdjuster() {
  wait();
  if (worker_.joinable()) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_.store(true, std::memory_order_release);
    }
    cv_.notify_one();
    worker_.join();
  }
  if (ba_) svo_ba_destroy(ba_);
}

void BundleAdjuster::add_keyframe(std::shared_ptr<Keyframe> kf) {  // src/bundle_adjuster.cpp:60-135
  wait();
  if (!ba_) return;
  double pose7[7] = {kf->orientation.w_, kf->orientation.x_, kf->orientation.y_, kf->orientation.z_,
                     kf->position(0), kf->position(1), kf->position(2)};
  const int nt = (int)kf->tracked_ids.size(), nn = (int)kf->new_features_2d.size();
  std::vector<int64_t> tid(nt), nid(nn > 0 ? nn : 1);
  for (int i = 0; i < nt; ++i) tid[i] = (int64_t)kf->tracked_ids[i];
  int kept = 0;
  const int rc = svo_ba_add_keyframe(ba_, pose7, tid.data(), (const float*)kf->tracked_features_2d.data(), nt,
                                     (const float*)kf->new_features_2d.data(), (const float*)kf->new_features_3d.data(), nn,
                                     nid.data(), &kept);
  if (rc) return;
  kf->new_features_2d.resize(kept);  // :86-90
  kf->new_features_3d.resize(kept);
  kf->new_ids.clear();               // real ids only (SURVEY C-4)
  for (int i = 0; i < kept; ++i) kf->new_ids.push_back((size_t)nid[i]);
  last_keyframe_ = kf;               // :132
  new_frame_added_ = true;           // :134
  launch_needed_ = true;
}

void BundleAdjuster::bundle_adjust() {  // src/bundle_adjuster.cpp:137-157 (synchronous use, caller thread)
  wait();
  launch_needed_ = false;
  run_bundle_adjust();
}

// The solve itself; runs on the caller thread (bundle_adjust) or on the worker (bundle_adjust_async), never both:
// every caller-thread entry point joins the worker first.  Touches no caller-thread flag.
void BundleAdjuster::run_bundle_adjust() {
  last_iterations_ = 0;
  if (!ba_ || !last_keyframe_) return;
  if (new_frame_added_) {  // :138
    PhaseTimer pt(5);
    svo_ba_summary s;
    if (svo_ba_solve(ba_, &s)) return;
    last_iterations_ = s.iterations;
    double p[7];
    if (svo_ba_get_pose(ba_, -1, p)) return;
    last_keyframe_->orientation = Quaternionf{(float)p[0], (float)p[1], (float)p[2], (float)p[3]};  // :146-153
    last_keyframe_->position = Vector3f{{(float)p[4], (float)p[5], (float)p[6]}};
    new_frame_added_ = false;  // :155
  }
  solved_pose_[0] = last_keyframe_->orientation.w_; solved_pose_[1] = last_keyframe_->orientation.x_;
  solved_pose_[2] = last_keyframe_->orientation.y_; solved_pose_[3] = last_keyframe_->orientation.z_;
  for (int i = 0; i < 3; ++i) solved_pose_[4 + i] = last_keyframe_->position(i);
}

void BundleAdjuster::get_world_points(std::vector<Point3f>& world_points, const std::vector<size_t>& ids) {
  wait();  // the landmarks must be the bundle-adjusted ones (src/bundle_adjuster.cpp:159-163)
  const size_t n = ids.size();
  if (!n || !ba_) return;
  std::vector<int64_t> id64(n);
  for (size_t i = 0; i < n; ++i) id64[i] = (int64_t)ids[i];
  const size_t base = world_points.size();
  world_points.resize(base + n);
  svo_ba_get_points(ba_, id64.data(), (int)n, (float*)(world_points.data() + base));  // :159-163
}

void BundleAdjuster::get_world_points_into(float* xyz, const std::vector<size_t>& ids) {
  wait();
  const size_t n = ids.size();
  if (!n || !ba_) return;
  std::vector<int64_t> id64(n);
  for (size_t i = 0; i < n; ++i) id64[i] = (int64_t)ids[i];
  svo_ba_get_points(ba_, id64.data(), (int)n, xyz);  // :159-163
}

// ------------------------------------------------------------------------------------------ FeatureTracker
FeatureTracker::FeatureTracker(svo_ctx* ctx, int max_features, int max_width, int max_height) : ctx_(ctx), cap_(max_features) {
  pyr_cap_ = svo_k_pyramid_bytes(max_width, max_height);
  hipError_t e = hipSetDevice(ctx->device);
  auto dev = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
  for (int b = 0; b < 2; ++b) {
    dev((void**)&d_xy_[b], sizeof(float) * 2 * cap_);
    dev((void**)&d_init_[b], sizeof(float) * 2 * cap_);
    dev((void**)&d_ids_[b], sizeof(long long) * cap_);
  }
  dev((void**)&d_fwd_, sizeof(float) * 2 * cap_);
  dev((void**)&d_par_, sizeof(float) * cap_);
  dev((void**)&d_keep_, cap_);
  dev((void**)&d_kidx_, sizeof(int) * cap_);
  dev((void**)&d_n_, sizeof(int));
  dev((void**)&d_av_, sizeof(float));
  dev((void**)&d_last_pyr_, pyr_cap_);
  // pinned host mirrors of the current feature set: the compaction kernel writes them in place
  const size_t mirror_bytes = (sizeof(float) * 2 + sizeof(long long)) * (size_t)cap_ + 64;
  if (e == hipSuccess) e = hipHostMalloc((void**)&h_mirror_, mirror_bytes, hipHostMallocDefault);
  if (e != hipSuccess) {  // surfaced as SVO_ERR_HIP by svo_pipeline_create / the adapters (ok() == false)
    ctx_->err = std::string("FeatureTracker: allocation failed: ") + hipGetErrorString(e);
    return;
  }
  memset(h_mirror_, 0, mirror_bytes);
  h_ids_ = reinterpret_cast<long long*>(h_mirror_);
  h_xy_ = reinterpret_cast<float*>(h_ids_ + cap_);
  h_n_ = reinterpret_cast<int*>(h_xy_ + 2 * (size_t)cap_);
  h_av_ = reinterpret_cast<float*>(h_n_ + 1);
  alloc_ok_ = true;
}

FeatureTracker::~FeatureTracker() {
  void* ptrs[] = {d_xy_[0], d_xy_[1], d_init_[0], d_init_[1], d_ids_[0], d_ids_[1], d_fwd_, d_par_, d_keep_, d_kidx_, d_n_, d_av_, d_last_pyr_,
                  d_host_img_, d_host_pyr_};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h_mirror_) (void)hipHostFree(h_mirror_);
  if (h_init_dup_) (void)hipHostFree(h_init_dup_);
}

void FeatureTracker::init(const uint8_t* pyramid, int width, int height, const std::vector<Point2f>& features,
                          const std::vector<size_t>& ids) {  // src/feature_tracker.cpp:3-16
  hipStream_t st = ctx_->stream;
  int n = (int)ids.size();
  if (n > cap_) n = cap_;
  // the previous init launch may still be reading the mirrors
  if (pending_init_seq_) {
    SvoPublish prev; prev.word = svo_word(ctx_, SVO_W_INIT); prev.seq = pending_init_seq_;
    if (svo_wait_word(ctx_, prev)) return;
    pending_init_seq_ = 0;
  }
  if (n) memcpy(h_xy_, features.data(), sizeof(float) * 2 * n);
  for (int i = 0; i < n; ++i) h_ids_[i] = (long long)ids[i];
  init_xy_.assign(features.begin(), features.begin() + n);  // initial_features (src/feature_tracker.cpp:9-12), host side
  init_ids_.assign(h_ids_, h_ids_ + n);
  // initial_features is a map (src/feature_tracker.hpp:49): insert() keeps the FIRST feature of an id (:10-12), the parallax
  // of every feature carrying that id is measured from it (:47) and percent_lost counts the map's entries (:64).  The
  // pipeline's ids ascend strictly (tracked survivors in order, then new ids): only a caller of the public init() can
  // hand over duplicates, and only then is the id -> first feature table built.
  bool ascending = true;
  for (int i = 1; i < n && ascending; ++i) ascending = ids[i] > ids[i - 1];
  int n_ids = n;
  const float* init_src = h_xy_;
  if (!ascending) {
    std::unordered_map<size_t, int> first;
    first.reserve((size_t)n * 2);
    std::vector<int> first_of((size_t)n);
    for (int i = 0; i < n; ++i) first_of[i] = first.emplace(ids[i], i).first->second;
    n_ids = (int)first.size();
    if (n_ids != n) {
      if (!h_init_dup_ && hipHostMalloc((void**)&h_init_dup_, sizeof(float) * 2 * (size_t)cap_, hipHostMallocDefault) != hipSuccess) {
        h_init_dup_ = nullptr;
        ctx_->err = "FeatureTracker::init: pinned allocation failed";
        return;
      }
      for (int i = 0; i < n; ++i) { h_init_dup_[2 * i] = features[first_of[i]].x; h_init_dup_[2 * i + 1] = features[first_of[i]].y; }
      init_src = h_init_dup_;
    }
  }
  const SvoPublish pub = svo_publish_next(ctx_, SVO_W_INIT);
  pending_init_seq_ = pub.seq;
  if (svo_k_tracker_init(ctx_, h_xy_, init_src, h_ids_, n, d_xy_[cur_], d_init_[cur_], d_ids_[cur_], &pub)) return;
  remember(pyramid, width, height);  // clone, :14
  if (!borrow_) SVO_TRY(hipStreamSynchronize(st));  // the caller may reuse `pyramid` right away
  n_ = n;
  n_initial_ = n_ids;
  has_image_ = true;
}

// The reference clones the image it will track from (src/feature_tracker.cpp:14,66).  With borrowed pyramids
// (the owner keeps them alive and calls retain() before overwriting them) the clone is deferred to retain().
void FeatureTracker::remember(const uint8_t* pyramid, int width, int height) {
  last_w_ = width; last_h_ = height;
  if (borrow_) { last_pyr_ = pyramid; return; }
  SVO_TRY(hipMemcpyAsync(d_last_pyr_, pyramid, svo_k_pyramid_bytes(width, height), hipMemcpyDeviceToDevice, ctx_->stream));
  last_pyr_ = d_last_pyr_;
}

void FeatureTracker::retain() {
  if (!last_pyr_ || last_pyr_ == d_last_pyr_) return;
  SVO_TRY(hipMemcpyAsync(d_last_pyr_, last_pyr_, svo_k_pyramid_bytes(last_w_, last_h_), hipMemcpyDeviceToDevice, ctx_->stream));
  last_pyr_ = d_last_pyr_;
}

void FeatureTracker::track_features(float& av_parallax, float& percent_lost, const uint8_t* pyramid, int width, int height,
                                    bool /*flow_back: the reference always passes true*/) {  // src/feature_tracker.cpp:18-67
  hipStream_t st = ctx_->stream;
  const int nxt = 1 - cur_;
  // one compaction launch also carries (initial position, id) along with the kept features (C-1: old ids), mirrors
  // the kept features / ids into pinned host memory and publishes (n_kept, av_parallax) + a completion word:
  // no gather launch, no D2H blits, no stream wait
  SvoTrackCarry carry;
  carry.init_src = d_init_[cur_]; carry.ids_src = d_ids_[cur_]; carry.init_dst = d_init_[nxt]; carry.ids_dst = d_ids_[nxt];
  carry.host_n = h_n_; carry.host_av = h_av_; carry.host_xy = h_xy_; carry.host_ids = h_ids_;
  carry.pub = svo_publish_next(ctx_, SVO_W_TRACK);
  if (svo_k_track(ctx_, last_pyr_, pyramid, width, height, d_xy_[cur_], d_init_[cur_], nullptr, n_, d_fwd_, d_keep_, d_par_,
                  d_xy_[nxt], d_kidx_, d_n_, d_av_, &carry)) return;
  remember(pyramid, width, height);  // :66
  if (svo_wait_word(ctx_, carry.pub)) return;
  if (!borrow_) SVO_TRY(hipStreamSynchronize(st));
  pending_init_seq_ = 0;  // stream order: the init launch finished before this track did
  cur_ = nxt;
  n_ = *h_n_;
  av_parallax = *h_av_;                                                                   // :63
  percent_lost = (float)(1.0 - (double)((float)n_ / (float)n_initial_));                   // :64
}

void FeatureTracker::get_tracked_features(std::vector<Point2f>& features, std::vector<size_t>& ids) {  // :69-72
  features.resize(n_);
  ids.resize(n_);
  if (!n_) return;
  memcpy(features.data(), h_xy_, sizeof(float) * 2 * n_);  // pinned mirrors, current since the last init / track
  for (int i = 0; i < n_; ++i) ids[i] = (size_t)h_ids_[i];
}

// draw_track's inputs (src/feature_tracker.cpp:78-82): keyframe position (initial_features.at(id)) and current position of
// every feature.  Host data only: the current set lives in the pinned mirrors, the keyframe set was copied at init().
void FeatureTracker::get_track_arrows(std::vector<Point2f>& initial, std::vector<Point2f>& current) {
  initial.resize(n_);
  current.resize(n_);
  if (!n_) return;
  memcpy(current.data(), h_xy_, sizeof(float) * 2 * n_);
  std::unordered_map<long long, Point2f> at;
  at.reserve(init_ids_.size() * 2);
  for (size_t i = 0; i < init_ids_.size(); ++i) at.emplace(init_ids_[i], init_xy_[i]);  // insert(): the first entry of an id wins (:10)
  for (int i = 0; i < n_; ++i) {
    const auto it = at.find(h_ids_[i]);
    initial[i] = it != at.end() ? it->second : current[i];
  }
}

void FeatureTracker::draw_track() {  // src/feature_tracker.cpp:74-83 (the pixels are drawn by whoever holds the image)
  if (!drawing_) return;
  get_track_arrows(drawn_initial_, drawn_current_);
  ++draw_serial_;
}

// mono8 host image -> HBM staging -> 4-level pyramid (what the device-resident methods take)
int FeatureTracker::upload_pyramid(const uint8_t* image, int width, int height, int stride) {
  if (!alloc_ok_ || !image || width < 1 || height < 1 || stride < width) return SVO_ERR_INVALID;
  if (svo_k_pyramid_bytes(width, height) > pyr_cap_) { ctx_->err = "FeatureTracker: image larger than the tracker's capacity"; return SVO_ERR_CAPACITY; }
  if (!d_host_img_) {
    SVO_HIP_CHECK(ctx_, hipMalloc((void**)&d_host_img_, pyr_cap_));  // level 0 of the largest pyramid bounds the image
    SVO_HIP_CHECK(ctx_, hipMalloc((void**)&d_host_pyr_, pyr_cap_));
  }
  SVO_HIP_CHECK(ctx_, hipMemcpy2DAsync(d_host_img_, (size_t)width, image, (size_t)stride, (size_t)width, (size_t)height,
                                       hipMemcpyHostToDevice, ctx_->stream));
  return svo_k_build_pyramid(ctx_, d_host_img_, 1, width, height, width, (size_t)width * height, d_host_pyr_, svo_k_pyramid_bytes(width, height));
}

int FeatureTracker::init_host(const uint8_t* image, int width, int height, int stride, const std::vector<Point2f>& features,
                              const std::vector<size_t>& ids) {
  const bool b = borrow_;
  borrow_ = false;  // the staging pyramid is overwritten by the next call: clone it now (src/feature_tracker.cpp:14)
  int rc = upload_pyramid(image, width, height, stride);
  if (!rc) init(d_host_pyr_, width, height, features, ids);
  borrow_ = b;
  return rc ? rc : (ctx_->err.empty() ? SVO_OK : SVO_ERR_HIP);
}

int FeatureTracker::track_features_host(float& av_parallax, float& percent_lost, const uint8_t* image, int width, int height,
                                        int stride, bool flow_back) {
  if (!has_image_) return SVO_ERR_INVALID;
  const bool b = borrow_;
  borrow_ = false;
  retain();  // the previous image may be a borrowed pyramid
  int rc = upload_pyramid(image, width, height, stride);
  if (!rc) track_features(av_parallax, percent_lost, d_host_pyr_, width, height, flow_back);
  borrow_ = b;
  return rc ? rc : (ctx_->err.empty() ? SVO_OK : SVO_ERR_HIP);
}

// ------------------------------------------------------------------------------------------ ImageProcessor
namespace {
// cv::Rodrigues on a CV_32F rvec with declared arithmetic (host/det_trig.h): the same bits on the host, on the device and in the oracle
inline void rodrigues_f(const float* rv, float* R9) { svo_det_rodrigues_f(rv, R9); }

// Eigen::Quaternionf(Matrix3f) (src/image_processor.cpp:92), float arithmetic, row-major m.
void quat_from_R(const float* m, float* q /*wxyz*/) {
  float t = m[0] + m[4] + m[8];
  if (t > 0.f) {
    t = sqrtf(t + 1.0f);
    q[0] = 0.5f * t;
    t = 0.5f / t;
    q[1] = (m[7] - m[5]) * t; q[2] = (m[2] - m[6]) * t; q[3] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
    q[1 + i] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
}
}  // namespace

ImageProcessor::ImageProcessor(svo_ctx* ctx, const float K[9], std::shared_ptr<FeatureTracker> tracker,
                               std::shared_ptr<BundleAdjuster> adjuster, float bline, float min_dist, float par_thresh,
                               int max_corners, double quality, int max_batch)
    : ctx_(ctx), feature_tracker(std::move(tracker)), bundle_adjuster(std::move(adjuster)), baseline(bline),
      min_feature_distance(min_dist), parallax_thresh(par_thresh), max_corners_(max_corners), quality_(quality),
      max_batch_(max_batch < 1 ? 1 : max_batch) {
  memcpy(K_, K, sizeof(K_));
  feature_tracker->borrow_pyramids(true);  // the batch pyramids outlive every frame of the batch; retain() in prepare_batch
  pyr_stride_ = svo_k_pyramid_bytes(ctx->lim.max_width, ctx->lim.max_height);
  const size_t mc = (size_t)max_corners_, mf = (size_t)ctx->lim.max_features;
  hipError_t e = hipSetDevice(ctx->device);
  auto dev = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
  dev((void**)&d_corners_, sizeof(float) * 2 * mc * max_batch_);
  dev((void**)&d_ncorners_, sizeof(int) * max_batch_);
  dev((void**)&d_pyr_, pyr_stride_ * max_batch_);
  dev((void**)&d_xyz_, sizeof(float) * 3 * mf);
  dev((void**)&d_trk_xy_, sizeof(float) * 2 * mf);
  dev((void**)&d_trk_ids_, sizeof(long long) * mf);
  dev((void**)&d_inl_, sizeof(int) * mf);
  dev((void**)&d_new_xy_, sizeof(float) * 2 * mc);
  dev((void**)&d_disp_, sizeof(float) * mc);
  dev((void**)&d_kxy_, sizeof(float) * 5 * mc + 64);  // kept xy (2) + xyz (3) contiguous for one copy
  dev((void**)&d_cnt_, sizeof(int) * 4);
  h_ncorners_.assign(max_batch_, 0);
  // pinned host arena: world points for PnP | inlier list | triangulation outputs (count, kept 2-D, 3-D) | batch corner
  // counts + status.  Kernels read / write these in place; the host polls completion words (common.h SvoPublish).
  const size_t bytes = sizeof(float) * 3 * mf + sizeof(int) * mf + 64 + sizeof(float) * 5 * mc + 256 + sizeof(int) * ((size_t)max_batch_ + 16);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h_arena_, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {  // surfaced as SVO_ERR_HIP by svo_pipeline_create / the adapters (ok() == false)
    ctx_->err = std::string("ImageProcessor: allocation failed: ") + hipGetErrorString(e);
    return;
  }
  memset(h_arena_, 0, bytes);
  d_kxyz_ = d_kxy_ + 2 * mc;
  h_xyz_ = reinterpret_cast<float*>(h_arena_);
  h_inl_ = reinterpret_cast<int*>(h_xyz_ + 3 * mf);
  h_tri_cnt_ = h_inl_ + mf;
  h_tri_xy_ = reinterpret_cast<float*>(h_tri_cnt_ + 16);
  h_tri_xyz_ = h_tri_xy_ + 2 * mc;
  h_batch_cnt_ = reinterpret_cast<int*>(h_tri_xyz_ + 3 * mc + 16);
  alloc_ok_ = feature_tracker->ok() && bundle_adjuster->handle() != nullptr;
}

ImageProcessor::~ImageProcessor() {
  void* ptrs[] = {d_corners_, d_ncorners_, d_pyr_, d_xyz_, d_trk_xy_, d_trk_ids_, d_inl_, d_new_xy_, d_disp_, d_kxy_, d_cnt_, d_stage_};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h_arena_) (void)hipHostFree(h_arena_);
}

void ImageProcessor::reset() {
  rvec[0] = rvec[1] = rvec[2] = tvec[0] = tvec[1] = tvec[2] = 0.f;
  batch_ = 0;
  feature_tracker->reset();
  bundle_adjuster->reset();
}

int ImageProcessor::prepare_batch(const uint8_t* left, int batch, int width, int height) {
  PhaseTimer pt(0);
  if (!alloc_ok_) return SVO_ERR_HIP;
  if (batch < 1 || batch > max_batch_ || batch > ctx_->lim.max_batch) { ctx_->err = "prepare_batch: batch outside limits"; return SVO_ERR_INVALID; }
  width_ = width; height_ = height; batch_ = batch;
  const size_t istride = (size_t)width * height;
  // a1 on every frame (the reference detects on every frame, src/image_processor.cpp:22, SURVEY C-7)
  int rc = svo_corner_detect_batch_dev(ctx_, left, batch, width, height, width, istride, max_corners_, quality_,
                                       (double)min_feature_distance, d_corners_, d_ncorners_);
  if (rc) return rc;
  pyr_stride_ = svo_k_pyramid_bytes(width, height);
  feature_tracker->retain();  // the tracker's previous image may live in the buffer overwritten next
  rc = svo_k_build_pyramid(ctx_, left, batch, width, height, width, istride, d_pyr_, pyr_stride_);
  if (rc) return rc;
  int* h = h_batch_cnt_;
  if (hipMemcpyAsync(h, d_ncorners_, sizeof(int) * batch, hipMemcpyDeviceToHost, ctx_->stream) != hipSuccess ||
      hipMemcpyAsync(h + batch, ctx_->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx_->stream) != hipSuccess ||
      hipStreamSynchronize(ctx_->stream) != hipSuccess) {
    ctx_->err = "prepare_batch: readback failed";
    return SVO_ERR_HIP;
  }
  if (h[batch]) {
    (void)hipMemsetAsync(ctx_->d_status, 0, sizeof(int), ctx_->stream);
    ctx_->err = (h[batch] & 8) ? "corner detection: more raw local maxima than the streaming pass's list holds (raw_cap: width x height / 4 per image under a 1 GiB budget)"
                         : "corner detection exceeded a workspace bound (svo_limits.max_candidates)";
    return SVO_ERR_CAPACITY;
  }
  for (int i = 0; i < batch; ++i) h_ncorners_[i] = h[i];
  return SVO_OK;
}

// src/image_processor.cpp:165-208.  StereoBM evaluated only at the feature pixels (exactly equivalent, SURVEY C-8).
void ImageProcessor::triangulate_stereo(std::vector<Point3f>& features_3d, std::vector<Point2f>& valid_features_2d,
                                        const float* d_features, const int* d_n, int n_max, const DeviceImage& left,
                                        const DeviceImage& right, const float camera_pose[16]) {
  if (n_max <= 0) return;
  const SvoMat4 M = svo_k_reprojection_matrix(camera_pose, K_[0], K_[2], K_[5], baseline);  // :178-189
  // ONE launch for :173-176 and :190-207; the tail's outputs are only consumed by the host (keyframe bookkeeping): it
  // writes them into the pinned arena and publishes a completion word — no D2H blits, no stream wait
  SvoPublish pub;
  if (svo_k_stereo_triangulate(ctx_, left.data, right.data, left.width, left.height, left.stride, svo_ref::STEREO_NUM_DISPARITIES,
                               svo_ref::STEREO_BLOCK_SIZE, d_features, d_n, n_max, d_disp_, M, h_tri_xy_, h_tri_xyz_, nullptr, h_tri_cnt_,
                               SVO_W_TRI, &pub)) return;
  if (svo_wait_word(ctx_, pub)) return;
  const int m = *h_tri_cnt_;
  valid_features_2d.resize(m);
  features_3d.resize(m);
  memcpy(valid_features_2d.data(), h_tri_xy_, sizeof(float) * 2 * m);
  memcpy(features_3d.data(), h_tri_xyz_, sizeof(float) * 3 * m);
}

int ImageProcessor::process_host(const uint8_t* left, int left_stride, const uint8_t* right, int right_stride, int width,
                                 int height, double t) {
  if (!alloc_ok_) return SVO_ERR_HIP;
  SVO_REQUIRE(ctx_, left && right && width >= 32 && height >= 32 && left_stride >= width && right_stride >= width &&
                        width <= ctx_->lim.max_width && height <= ctx_->lim.max_height, "process_host: bad image arguments");
  const size_t bytes = (size_t)width * height;
  if (stage_bytes_ < 2 * bytes) {
    if (d_stage_) (void)hipFree(d_stage_);
    d_stage_ = nullptr; stage_bytes_ = 0;
    SVO_HIP_CHECK(ctx_, hipMalloc((void**)&d_stage_, 2 * bytes));
    stage_bytes_ = 2 * bytes;
  }
  SVO_HIP_CHECK(ctx_, hipMemcpy2DAsync(d_stage_, (size_t)width, left, (size_t)left_stride, (size_t)width, (size_t)height,
                                       hipMemcpyHostToDevice, ctx_->stream));
  SVO_HIP_CHECK(ctx_, hipMemcpy2DAsync(d_stage_ + bytes, (size_t)width, right, (size_t)right_stride, (size_t)width, (size_t)height,
                                       hipMemcpyHostToDevice, ctx_->stream));
  ctx_->err.clear();
  batch_ = 0;  // no prepared batch: process() detects and builds the pyramid for this frame
  process(StereoPair(DeviceImage{d_stage_, width, height, width}, DeviceImage{d_stage_ + bytes, width, height, width}, t, -1));
  return ctx_->err.empty() ? SVO_OK : SVO_ERR_HIP;
}

void ImageProcessor::process(const StereoPair& sp) {  // src/image_processor.cpp:18-163
  stats_ = Stats();
  int slot = sp.batch_slot;
  if (slot < 0 || slot >= batch_) {
    if (prepare_batch(sp.left.data, 1, sp.left.width, sp.left.height)) return;
    slot = 0;
  }
  hipStream_t st = ctx_->stream;
  const float* d_det = d_corners_ + (size_t)slot * 2 * max_corners_;
  const uint8_t* pyr = d_pyr_ + (size_t)slot * pyr_stride_;
  const int n_det = h_ncorners_[slot];
  stats_.n_detected = n_det;
  if (n_det < svo_ref::MIN_DETECTED) return;  // :23-25

  std::shared_ptr<Keyframe> last_keyframe = bundle_adjuster->get_last_keyframe();  // :27
  if (last_keyframe == nullptr) {  // :30-58
    std::vector<Point3f> features_3d;
    std::vector<Point2f> valid_features_2d;
    const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    triangulate_stereo(features_3d, valid_features_2d, d_det, nullptr, n_det, sp.left, sp.right, eye);
    auto kf = std::make_shared<Keyframe>(Vector3f{{0, 0, 0}}, Quaternionf{1, 0, 0, 0}, sp.left, std::vector<Point2f>(),
                                         std::vector<size_t>(), valid_features_2d, features_3d);
    bundle_adjuster->add_keyframe(kf);
    if (keyframe_hook_) keyframe_hook_();
    feature_tracker->init(pyr, width_, height_, kf->new_features_2d, kf->new_ids);
    tvec[0] = tvec[1] = tvec[2] = 0.f;
    rvec[0] = rvec[1] = rvec[2] = 0.f;
    stats_.is_keyframe = 1;
    stats_.n_new = (int)kf->new_ids.size();
    return;
  }

  float av_parallax = 0, percent_lost = 0;
  {
    PhaseTimer pt(1);
    feature_tracker->track_features(av_parallax, percent_lost, pyr, width_, height_, true);  // :62
  }
  stats_.n_tracked = feature_tracker->count();
  stats_.av_parallax = av_parallax;
  stats_.percent_lost = percent_lost;
  if (av_parallax <= parallax_thresh && (double)percent_lost < svo_ref::KEYFRAME_PERCENT_LOST) return;  // :63-65

  PhaseTimer phase(2);
  std::vector<Point2f> tracked_features;
  std::vector<size_t> tracked_ids;
  feature_tracker->get_tracked_features(tracked_features, tracked_ids);   // :71 (pinned mirrors: no copy, no wait)
  const int m = (int)tracked_ids.size();
  bundle_adjuster->get_world_points_into(h_xyz_, tracked_ids);           // :72, straight into the pinned arena

  // PnP :74-82 (rvec/tvec are CV_32F in/out; the solver works in double)
  double rv[3] = {rvec[0], rvec[1], rvec[2]}, tv[3] = {tvec[0], tvec[1], tvec[2]};
  int num_inliers = 0;
  const int* inlier_indices = h_inl_;
  if (m > 0) {
    SVO_TRY(hipMemcpyAsync(d_xyz_, h_xyz_, sizeof(float) * 3 * m, hipMemcpyHostToDevice, st));
    SvoScratch scratch(ctx_);
    if (svo_k_pnp(ctx_, scratch, d_xyz_, feature_tracker->device_features(), m, K_[0], K_[2], K_[5], rv, tv, svo_ref::PNP_ITERATIONS, svo_ref::PNP_REPROJ_ERROR,
                  svo_ref::PNP_CONFIDENCE,
                  d_inl_, &num_inliers, h_inl_, d_trk_xy_)) return;  // the refinement also leaves the inliers' features for the dedup below
  }
  for (int i = 0; i < 3; ++i) { rvec[i] = (float)rv[i]; tvec[i] = (float)tv[i]; }
  stats_.n_inliers = num_inliers;

  float rmat[9], q[4];
  rodrigues_f(rvec, rmat);   // :84-85
  quat_from_R(rmat, q);      // :87-92
  auto kf = std::make_shared<Keyframe>(Vector3f{{tvec[0], tvec[1], tvec[2]}}, Quaternionf{q[0], q[1], q[2], q[3]}, sp.left,
                                       std::vector<Point2f>(num_inliers), std::vector<size_t>(num_inliers),
                                       std::vector<Point2f>(), std::vector<Point3f>());
  for (int i = 0; i < num_inliers; ++i) {  // :104-108
    const int idx = inlier_indices[i];
    kf->tracked_ids[i] = tracked_ids[idx];
    kf->tracked_features_2d[i] = tracked_features[idx];
  }

  phase.next(3);
  // dedup :113-128 on the device; the surviving corners stay in HBM for the stereo stage
  if (svo_k_dedup(ctx_, d_det, nullptr, n_det, d_trk_xy_, nullptr, num_inliers, min_feature_distance, d_new_xy_, d_cnt_)) return;

  // hmat = [R^T | -R^T t]  :130-134 (float Mats; the product accumulates in double)
  float hmat[16] = {0};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) hmat[4 * i + j] = rmat[3 * j + i];
    double s = 0.0;
    for (int k = 0; k < 3; ++k) s += (double)(-rmat[3 * k + i]) * (double)tvec[k];
    hmat[4 * i + 3] = (float)s;
  }
  hmat[15] = 1.f;
  triangulate_stereo(kf->new_features_3d, kf->new_features_2d, d_new_xy_, d_cnt_, n_det, sp.left, sp.right, hmat);  // :137-142

  phase.next(4);
  bundle_adjuster->add_keyframe(kf);  // :144
  if (keyframe_hook_) keyframe_hook_();  // the driver may start the solve now: the rest of process() does not touch the graph
  feature_tracker->draw_track();      // :146 (arrow snapshot; a no-op unless an adapter enabled drawing)

  std::vector<Point2f> features_2d_for_tracker(kf->tracked_features_2d);  // :148-162
  features_2d_for_tracker.insert(features_2d_for_tracker.end(), kf->new_features_2d.begin(), kf->new_features_2d.end());
  std::vector<size_t> ids_for_tracker(kf->tracked_ids);
  ids_for_tracker.insert(ids_for_tracker.end(), kf->new_ids.begin(), kf->new_ids.end());
  feature_tracker->init(pyr, width_, height_, features_2d_for_tracker, ids_for_tracker);
  stats_.is_keyframe = 1;
  stats_.n_new = (int)kf->new_ids.size();
}

}  // namespace svo

// ------------------------------------------------------------------------------------------------- C-ABI
struct svo_pipeline {
  svo_ctx* ctx;
  svo_pipeline_params prm;
  std::shared_ptr<svo::FeatureTracker> tracker;
  std::shared_ptr<svo::BundleAdjuster> adjuster;
  std::unique_ptr<svo::ImageProcessor> proc;
  uint8_t* d_imgs = nullptr;  // staging for the host-pointer entry point (left batch | right batch)
  size_t d_imgs_bytes = 0;
};

extern "C" int svo_reference_constants(svo_reference_constants_t* c) {
  if (!c) return SVO_ERR_INVALID;
  c->gftt_max_corners = svo_ref::GFTT_MAX_CORNERS; c->gftt_quality = svo_ref::GFTT_QUALITY; c->min_detected = svo_ref::MIN_DETECTED;
  c->keyframe_percent_lost = svo_ref::KEYFRAME_PERCENT_LOST;
  c->pnp_iterations = svo_ref::PNP_ITERATIONS; c->pnp_reproj_error = svo_ref::PNP_REPROJ_ERROR; c->pnp_confidence = svo_ref::PNP_CONFIDENCE;
  c->stereo_num_disparities = svo_ref::STEREO_NUM_DISPARITIES; c->stereo_block_size = svo_ref::STEREO_BLOCK_SIZE;
  c->stereo_disparity_scale = svo_ref::STEREO_DISPARITY_SCALE; c->triangulate_min_disparity_exclusive = svo_ref::TRIANGULATE_MIN_DISPARITY;
  c->lk_win_w = c->lk_win_h = svo_ref::LK_WIN; c->lk_max_level = svo_ref::LK_MAX_LEVEL; c->lk_max_iterations = svo_ref::LK_MAX_ITERATIONS;
  c->lk_epsilon = svo_ref::LK_EPSILON; c->lk_min_eig_threshold = svo_ref::LK_MIN_EIG_THRESHOLD;
  c->fb_max_distance = svo_ref::FB_MAX_DISTANCE; c->max_parallax = svo_ref::MAX_PARALLAX; c->draw_thickness = svo_ref::DRAW_THICKNESS;
  c->parallax_thresh = svo_ref::PARALLAX_THRESH; c->min_feature_distance = svo_ref::MIN_FEATURE_DISTANCE;
  c->sliding_window_size = svo_ref::SLIDING_WINDOW_SIZE; c->max_features = svo_ref::MAX_FEATURES;
  c->ba_max_solver_time_s = svo_ref::BA_MAX_SOLVER_TIME_S; c->ba_num_threads = svo_ref::BA_NUM_THREADS;
  return SVO_OK;
}

extern "C" void svo_pipeline_default_params(svo_pipeline_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->cam.focal = 718.856; p->cam.cx = 607.1928; p->cam.cy = 185.2157; p->cam.baseline = 0.537165718864418;  // config/kitti00.yaml:1-4
  p->width = 1241; p->height = 376;
  p->max_corners = svo_ref::GFTT_MAX_CORNERS;              // src/image_processor.cpp:22
  p->quality = svo_ref::GFTT_QUALITY;
  p->min_feature_distance = svo_ref::MIN_FEATURE_DISTANCE;  // src/vo_node.cpp:34
  p->parallax_thresh = svo_ref::PARALLAX_THRESH;            // src/vo_node.cpp:33
  p->window_size = svo_ref::SLIDING_WINDOW_SIZE;            // src/vo_node.cpp:36
  p->max_features = svo_ref::MAX_FEATURES;                  // src/bundle_adjuster.hpp:75
  p->ba_max_iterations = 50;                                // Ceres default (Solver::Options::max_num_iterations)
  p->ba_max_time_s = svo_ref::BA_MAX_SOLVER_TIME_S;         // src/bundle_adjuster.cpp:11
}

extern "C" int svo_pipeline_create(svo_ctx* ctx, svo_pipeline** out, const svo_pipeline_params* p) {
  if (!ctx || !out || !p) return SVO_ERR_INVALID;
  SVO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  SVO_REQUIRE(ctx, p->width >= 32 && p->height >= 32 && p->width <= ctx->lim.max_width && p->height <= ctx->lim.max_height,
              "pipeline_create: image size outside the context limits");
  SVO_REQUIRE(ctx, p->max_corners >= 4 && p->max_corners <= ctx->lim.max_corners && p->max_features >= 4 &&
                       p->max_features <= ctx->lim.max_features, "pipeline_create: feature counts outside the context limits");
  SVO_REQUIRE(ctx, p->window_size >= 1 && p->window_size <= 63, "pipeline_create: window size must be 1..63");
  svo_pipeline* pl = new svo_pipeline();
  pl->ctx = ctx;
  pl->prm = *p;
  pl->tracker = std::make_shared<svo::FeatureTracker>(ctx, ctx->lim.max_features, ctx->lim.max_width, ctx->lim.max_height);
  pl->adjuster = std::make_shared<svo::BundleAdjuster>(ctx, (size_t)p->window_size, p->cam, p->max_features,
                                                       p->ba_max_iterations, p->ba_max_time_s);
  // K as the reference builds it (CV_32F 3x3, src/vo_node.cpp:104-108)
  const float K[9] = {(float)p->cam.focal, 0.f, (float)p->cam.cx, 0.f, (float)p->cam.focal, (float)p->cam.cy, 0.f, 0.f, 1.f};
  pl->proc.reset(new svo::ImageProcessor(ctx, K, pl->tracker, pl->adjuster, (float)p->cam.baseline, p->min_feature_distance,
                                         p->parallax_thresh, p->max_corners, p->quality, ctx->lim.max_batch));
  if (!pl->tracker->ok() || !pl->proc->ok()) {  // an allocation failed: report it here, not as a kernel fault later
    delete pl;
    return SVO_ERR_HIP;
  }
  *out = pl;
  return SVO_OK;
}

extern "C" void svo_pipeline_destroy(svo_pipeline* p) {
  if (!p) return;
  if (svo::PhaseTimer::enabled()) {
    for (int i = 0; i < 8; ++i) {
      const uint64_t ns = svo::PhaseTimer::acc_ns[i].load(std::memory_order_relaxed);
      if (ns) fprintf(stderr, "[svo timing] %-28s %10.3f ms\n", svo::PhaseTimer::name(i), 1e-6 * (double)ns);
    }
  }
  if (p->d_imgs) (void)hipFree(p->d_imgs);
  delete p;
}

extern "C" int svo_pipeline_reset(svo_pipeline* p) {
  if (!p) return SVO_ERR_INVALID;
  (void)hipSetDevice(p->ctx->device);
  p->adjuster->wait();
  p->proc->reset();
  return SVO_OK;
}

extern "C" int svo_pipeline_process_batch_dev(svo_pipeline* p, const uint8_t* left, const uint8_t* right, int batch,
                                              svo_frame_result* results) {
  if (!p) return SVO_ERR_INVALID;
  svo_ctx* ctx = p->ctx;
  SVO_REQUIRE(ctx, left && right && results && batch >= 1, "pipeline_process_batch: bad arguments");
  const int W = p->prm.width, H = p->prm.height;
  const size_t istride = (size_t)W * H;
  ctx->err.clear();
  SVO_HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the calling thread may never have selected this GPU
  struct Active { Active() { svo::pipeline_count_add(1); } ~Active() { svo::pipeline_count_add(-1); } } active;  // latency / throughput mode
  int rc = p->proc->prepare_batch(left, batch, W, H);
  if (rc) return rc;
  // The reference runs process() then bundle_adjust() per frame (src/vo_node.cpp:141-148).  Here the solve of
  // keyframe k runs asynchronously (own stream + worker thread) while frames k+1.. are tracked; it is joined
  // before anything reads the graph again (next keyframe's get_world_points / add_keyframe), so every number
  // is the same as in the synchronous order.  Poses of frames [k, next keyframe) are filled at that join.
  int pending_from = -1;
  auto fill_pending = [&](int upto) {
    if (pending_from < 0) return;
    for (int j = pending_from; j < upto; ++j) {
      memcpy(results[j].pose7, p->adjuster->solved_pose(), sizeof(results[j].pose7));
      if (j == pending_from && results[j].is_keyframe) results[j].ba_iterations = p->adjuster->last_iterations();
    }
    pending_from = -1;
  };
  // The solve of a new keyframe starts inside process(), right after add_keyframe (hook below), so that the host-side
  // gather / upload of the problem overlaps the tracker re-initialisation instead of following it.
  int cur = 0;
  p->proc->on_keyframe_added([&]() {
    // process() joined the previous solve before editing the graph: its poses are final now
    fill_pending(cur);
    pending_from = cur;
    p->adjuster->bundle_adjust_async();  // src/vo_node.cpp:147
  });
  for (int i = 0; i < batch; ++i) {
    cur = i;
    svo::DeviceImage L{left + i * istride, W, H, W}, R{right + i * istride, W, H, W};
    p->proc->process(svo::StereoPair(L, R, (double)i, i));  // src/vo_node.cpp:141-144
    svo_frame_result& r = results[i];
    memset(&r, 0, sizeof(r));
    const auto& s = p->proc->stats();
    r.n_detected = s.n_detected; r.n_tracked = s.n_tracked; r.n_inliers = s.n_inliers; r.n_new = s.n_new;
    r.is_keyframe = s.is_keyframe; r.av_parallax = s.av_parallax; r.percent_lost = s.percent_lost;
    if (!ctx->err.empty()) { p->adjuster->wait(); p->proc->on_keyframe_added(nullptr); return SVO_ERR_HIP; }
    if (p->adjuster->get_last_keyframe() == nullptr) continue;  // src/vo_node.cpp:146
    if (p->adjuster->new_keyframe_pending()) {  // (not reached with the hook; kept for a processor without it)
      fill_pending(i);
      pending_from = i;
      p->adjuster->bundle_adjust_async();
    } else if (pending_from < 0) {
      pending_from = i;  // no solve in flight: the pose is the last solved one
    }
  }
  p->proc->on_keyframe_added(nullptr);
  p->adjuster->wait();
  if (pending_from < 0) pending_from = batch;
  fill_pending(batch);
  if (!ctx->err.empty()) return SVO_ERR_HIP;
  return SVO_OK;
}

extern "C" int svo_pipeline_process_batch(svo_pipeline* p, const uint8_t* left, const uint8_t* right, int batch,
                                          svo_frame_result* results) {
  if (!p) return SVO_ERR_INVALID;
  svo_ctx* ctx = p->ctx;
  SVO_REQUIRE(ctx, left && right && results && batch >= 1 && batch <= ctx->lim.max_batch, "pipeline_process_batch: bad arguments");
  const size_t bytes = (size_t)p->prm.width * p->prm.height * batch;
  if (p->d_imgs_bytes < 2 * bytes) {
    if (p->d_imgs) (void)hipFree(p->d_imgs);
    p->d_imgs = nullptr; p->d_imgs_bytes = 0;
    SVO_HIP_CHECK(ctx, hipMalloc((void**)&p->d_imgs, 2 * bytes));
    p->d_imgs_bytes = 2 * bytes;
  }
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(p->d_imgs, left, bytes, hipMemcpyHostToDevice, ctx->stream));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(p->d_imgs + bytes, right, bytes, hipMemcpyHostToDevice, ctx->stream));
  return svo_pipeline_process_batch_dev(p, p->d_imgs, p->d_imgs + bytes, batch, results);
}

extern "C" int svo_pipeline_draw_track(svo_pipeline* p, const uint8_t* keyframe_gray, int row_stride, uint8_t* rgb) {
  if (!p || !keyframe_gray || !rgb) return SVO_ERR_INVALID;
  (void)hipSetDevice(p->ctx->device);
  std::vector<svo::Point2f> cur, init;
  p->tracker->get_track_arrows(init, cur);
  return svo_draw_track(keyframe_gray, p->prm.width, p->prm.height, row_stride, (const float*)init.data(),
                        (const float*)cur.data(), (int)cur.size(), rgb);
}

extern "C" int svo_pipeline_get_tracked(svo_pipeline* p, int64_t* ids, float* xy, int capacity, int* n) {
  if (!p || !n) return SVO_ERR_INVALID;
  (void)hipSetDevice(p->ctx->device);
  p->adjuster->wait();
  std::vector<svo::Point2f> f;
  std::vector<size_t> id;
  p->tracker->get_tracked_features(f, id);
  *n = (int)id.size();
  for (int i = 0; i < *n && i < capacity; ++i) {
    if (ids) ids[i] = (int64_t)id[i];
    if (xy) { xy[2 * i] = f[i].x; xy[2 * i + 1] = f[i].y; }
  }
  return SVO_OK;
}
