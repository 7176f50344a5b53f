// Host-side mirror of the reference's class surfaces, device resident.
//
//   reference (OpenCV/Eigen/Ceres types)                    this header (plain types, HBM-resident images)
//   --------------------------------------------------------------------------------------------------------
//   struct CameraInfo              src/camera_info.hpp:4-18  ->  svo_camera_info (same fields, same order)
//   struct StereoPair              src/image_processor.hpp:9-17 -> svo::StereoPair {left, right, t}
//   struct Keyframe                src/bundle_adjuster.hpp:22-46 -> svo::Keyframe (same 8 fields)
//   class  FeatureTracker          src/feature_tracker.hpp:20-54 -> svo::FeatureTracker (init / track_features /
//                                                                   get_tracked_features)
//   class  BundleAdjuster          src/bundle_adjuster.hpp:86-126 -> svo::BundleAdjuster (get_last_keyframe /
//                                                                   add_keyframe / bundle_adjust / get_world_points)
//   class  ImageProcessor          src/image_processor.hpp:31-46 -> svo::ImageProcessor (ctor, process)
//   class  ReprojectionFactor      src/reprojection_factor.hpp:7-18 -> svo::ReprojectionFactor (Evaluate, Ceres' null
//                                                                   conventions) — host entry to the a11 kernel
// Same method names, argument meaning and error behaviour (void + silent early return).  cv::Mat / Eigen
// types do not exist in this environment; INTEGRATION.md shows the few-line adapters a maintainer with
// OpenCV/Eigen installed adds so that vo_node.cpp compiles unchanged against these classes.
#ifndef SVO_STEREO_VO_HPP_
#define SVO_STEREO_VO_HPP_
#include <cstddef>
#include <cstdint>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "svo.h"

struct svo_ba;

namespace svo {

struct Point2f { float x, y; };
struct Point3f { float x, y, z; };
struct Vector3f { float v[3]; float& operator()(int i) { return v[i]; } float operator()(int i) const { return v[i]; } };
struct Quaternionf { float w_, x_, y_, z_; float& w() { return w_; } float& x() { return x_; } float& y() { return y_; } float& z() { return z_; } };

// mono8 image resident in HBM (what cv::Mat holds on the host in the reference)
struct DeviceImage { const uint8_t* data = nullptr; int width = 0, height = 0, stride = 0; };

struct StereoPair {  // src/image_processor.hpp:9-17
  DeviceImage left, right;
  double t;
  int batch_slot;  // index into the batch prepared with ImageProcessor::prepare_batch (-1: none)
  StereoPair(const DeviceImage& l, const DeviceImage& r, double t_, int slot = -1) : left(l), right(r), t(t_), batch_slot(slot) {}
};

struct Keyframe {  // src/bundle_adjuster.hpp:22-46
  Vector3f position;
  Quaternionf orientation;
  DeviceImage image;
  std::vector<Point2f> tracked_features_2d;
  std::vector<size_t> tracked_ids;
  std::vector<Point2f> new_features_2d;
  std::vector<Point3f> new_features_3d;
  std::vector<size_t> new_ids;
  Keyframe(Vector3f p, Quaternionf q, DeviceImage img, std::vector<Point2f> t2d, std::vector<size_t> tid,
           std::vector<Point2f> n2d, std::vector<Point3f> n3d)
      : position(p), orientation(q), image(img), tracked_features_2d(std::move(t2d)), tracked_ids(std::move(tid)),
        new_features_2d(std::move(n2d)), new_features_3d(std::move(n3d)) {}
};

class ReprojectionFactor {  // src/reprojection_factor.hpp:7-18
 public:
  ReprojectionFactor(svo_ctx* ctx, double ox, double oy, svo_camera_info info) : ctx_(ctx), ox_(ox), oy_(oy), info_(info) {}
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const;
 private:
  svo_ctx* ctx_;
  double ox_, oy_;
  svo_camera_info info_;
};

class BundleAdjuster {  // src/bundle_adjuster.hpp:86-126
 public:
  BundleAdjuster(svo_ctx* ctx, size_t window_size, svo_camera_info info, int max_features = 400,
                 int max_iterations = 50, double max_time_s = 0.1);
  ~BundleAdjuster();
  std::shared_ptr<Keyframe> get_last_keyframe() { return last_keyframe_; }
  void add_keyframe(std::shared_ptr<Keyframe> keyframe);
  void bundle_adjust();
  // Same solve, started on the adjuster's own HIP stream and a worker thread so that it overlaps the
  // tracker's kernels of the following frames; every method that reads or edits the graph joins it first,
  // so results are identical to the synchronous call.
  void bundle_adjust_async();
  void wait();
  bool pending() const { return job_state_.load(std::memory_order_acquire) != 0; }
  bool new_keyframe_pending() const { return launch_needed_; }  // caller-thread flag (the worker never touches it)
  void get_world_points(std::vector<Point3f>& world_points, const std::vector<size_t>& ids);
  void get_world_points_into(float* xyz, const std::vector<size_t>& ids);  // same, into a caller buffer (3 floats per id)
  int last_iterations() const { return last_iterations_; }
  const double* solved_pose() const { return solved_pose_; }
  svo_ba* handle() { return ba_; }
  void reset();
 private:
  void run_bundle_adjust();
  svo_ctx* ctx_;
  svo_ba* ba_ = nullptr;
  size_t window_size_;
  svo_camera_info info_;
  int max_features_, max_iterations_;
  double max_time_s_;
  std::shared_ptr<Keyframe> last_keyframe_;
  int last_iterations_ = 0;
  bool new_frame_added_ = false;   // owned by whoever runs bundle_adjust()
  bool launch_needed_ = false;     // owned by the caller thread
  double solved_pose_[7] = {1, 0, 0, 0, 0, 0, 0};
  // One persistent worker per adjuster (created with the first asynchronous solve): a keyframe costs no thread
  // creation.  job_state_: 0 idle, 1 posted / running.  The worker spins briefly for the next job (keyframes arrive
  // every millisecond or so when a stream runs flat out), then sleeps on the condition variable.
  void worker_loop();
  std::thread worker_;
  std::atomic<int> job_state_{0};
  std::atomic<bool> quit_{false};
  std::mutex mu_;
  std::condition_variable cv_;       // a job was posted (or quit)
  std::condition_variable cv_done_;  // the job finished
};

// How long host threads busy-poll before they go to sleep (pause iterations).  One or two stereo streams running at
// this moment (calls inside svo_pipeline_process_batch*): latency mode, spin (a sleeping thread costs ~15 us to wake, twice per keyframe).  More: throughput mode — many stereo
// streams share the host's CPU quota (16 cores per GPU on the target boxes; two spinning threads per stream exhaust it
// at 8 streams and the scheduler then throttles every thread), so waits that usually last hundreds of microseconds
// (the main thread joining a solve, the worker between solves) spin only briefly.
unsigned spin_budget();
void pipeline_count_add(int delta);

class FeatureTracker {  // src/feature_tracker.hpp:20-54 (draw_track/get_drawing: get_track_arrows + svo_draw_track)
 public:
  FeatureTracker(svo_ctx* ctx, int max_features, int max_width, int max_height);
  ~FeatureTracker();
  // `pyramid` is the device pyramid of `image` (built by ImageProcessor::prepare_batch); it is copied, as the
  // reference clones the image (src/feature_tracker.cpp:14).
  void init(const uint8_t* pyramid, int width, int height, const std::vector<Point2f>& features, const std::vector<size_t>& ids);
  void track_features(float& av_parallax, float& percent_lost, const uint8_t* pyramid, int width, int height, bool flow_back);
  void get_tracked_features(std::vector<Point2f>& features, std::vector<size_t>& ids);
  // inputs of draw_track (src/feature_tracker.cpp:74-83): per feature its keyframe position and its current position;
  // svo_draw_track rasterises them over the keyframe image (draw_track / get_drawing of the reference)
  void get_track_arrows(std::vector<Point2f>& initial, std::vector<Point2f>& current);
  // draw_track as the reference calls it from ImageProcessor::process (src/image_processor.cpp:146): when enabled,
  // snapshots the arrows of the current feature set (host data only; the adapter rasterises them over the keyframe
  // image it keeps).  Disabled by default: the C-ABI pipeline draws on demand (svo_pipeline_draw_track).
  void enable_drawing(bool on) { drawing_ = on; }
  void draw_track();
  unsigned drawing_serial() const { return draw_serial_; }   // bumped by every draw_track() snapshot
  const std::vector<Point2f>& drawn_initial() const { return drawn_initial_; }
  const std::vector<Point2f>& drawn_current() const { return drawn_current_; }
  // Host-image forms of init / track_features (what the reference's cv::Mat overloads bind to): upload the mono8 image
  // (row stride in bytes), build its pyramid on the device, then run the device-resident method.
  int init_host(const uint8_t* image, int width, int height, int stride, const std::vector<Point2f>& features,
                const std::vector<size_t>& ids);
  int track_features_host(float& av_parallax, float& percent_lost, const uint8_t* image, int width, int height, int stride,
                          bool flow_back);
  bool ok() const { return alloc_ok_; }
  // device views of the current feature set (for the in-library pipeline)
  const float* device_features() const { return d_xy_[cur_]; }
  const long long* device_ids() const { return d_ids_[cur_]; }
  int count() const { return n_; }
  void reset() { n_ = 0; n_initial_ = 0; has_image_ = false; last_pyr_ = nullptr; }
  // Borrowed pyramids: the caller keeps every pyramid passed to init()/track_features() alive and calls retain()
  // before overwriting it; the per-frame clone (src/feature_tracker.cpp:14,66) is then made only at retain().
  void borrow_pyramids(bool on) { borrow_ = on; }
  void retain();
 private:
  void remember(const uint8_t* pyramid, int width, int height);
  int upload_pyramid(const uint8_t* image, int width, int height, int stride);
  svo_ctx* ctx_;
  bool alloc_ok_ = false;
  uint8_t* d_host_img_ = nullptr;   // staging of the host-image entry points: image | its pyramid (allocated on first use)
  uint8_t* d_host_pyr_ = nullptr;
  std::vector<long long> init_ids_;  // host copy of the keyframe feature set (initial_features, src/feature_tracker.hpp:49)
  std::vector<Point2f> init_xy_;
  bool drawing_ = false;
  unsigned draw_serial_ = 0;
  std::vector<Point2f> drawn_initial_, drawn_current_;
  int cap_;
  size_t pyr_cap_;
  float* d_xy_[2] = {nullptr, nullptr};
  float* d_init_[2] = {nullptr, nullptr};
  long long* d_ids_[2] = {nullptr, nullptr};
  float* d_fwd_ = nullptr; float* d_par_ = nullptr; uint8_t* d_keep_ = nullptr; int* d_kidx_ = nullptr;
  int* d_n_ = nullptr; float* d_av_ = nullptr;
  uint8_t* d_last_pyr_ = nullptr;
  const uint8_t* last_pyr_ = nullptr;  // d_last_pyr_ or a borrowed pyramid
  uint8_t* h_mirror_ = nullptr;        // pinned: ids | xy | n | av_parallax, written by the kernels in place
  float* h_init_dup_ = nullptr;        // pinned, allocated on first need: initial positions when init() was handed duplicate ids
  long long* h_ids_ = nullptr; float* h_xy_ = nullptr; int* h_n_ = nullptr; float* h_av_ = nullptr;
  int pending_init_seq_ = 0;           // completion word value of an init launch that may still read the mirrors (0: none)
  int last_w_ = 0, last_h_ = 0;
  bool borrow_ = false;
  int cur_ = 0, n_ = 0, n_initial_ = 0;
  bool has_image_ = false;
};

class ImageProcessor {  // src/image_processor.hpp:31-46
 public:
  ImageProcessor(svo_ctx* ctx, const float K[9], std::shared_ptr<FeatureTracker> tracker,
                 std::shared_ptr<BundleAdjuster> adjuster, float bline, float min_feature_distance, float parallax_thresh,
                 int max_corners = 300, double quality = 0.1, int max_batch = 1);
  ~ImageProcessor();
  // Batch the stateless per-frame stages (corner detection on every left image, pyramids) for `batch`
  // consecutive frames resident in HBM.  process() on those frames then only runs the sequential chain.
  int prepare_batch(const uint8_t* left, int batch, int width, int height);
  void process(const StereoPair& stereo_pair);
  // Host-image form (what the reference's StereoPair of cv::Mat binds to): uploads both mono8 images (row strides in
  // bytes) into HBM staging owned by this object, then process().  Returns an svo_status.
  int process_host(const uint8_t* left, int left_stride, const uint8_t* right, int right_stride, int width, int height,
                   double t);
  bool ok() const { return alloc_ok_; }
  // Called by process() right after BundleAdjuster::add_keyframe (before the tracker is re-initialised): lets a driver
  // start the asynchronous solve of the new keyframe while process() finishes.  Unset: nothing happens.
  void on_keyframe_added(std::function<void()> hook) { keyframe_hook_ = std::move(hook); }
  // per-frame diagnostics of the last process() call
  struct Stats { int n_detected = 0, n_tracked = 0, n_inliers = 0, n_new = 0, is_keyframe = 0; float av_parallax = 0, percent_lost = 0; };
  const Stats& stats() const { return stats_; }
  void reset();
 private:
  void triangulate_stereo(std::vector<Point3f>& features_3d, std::vector<Point2f>& valid_features_2d,
                          const float* d_features, const int* d_n, int n_max, const DeviceImage& left,
                          const DeviceImage& right, const float camera_pose[16]);
  svo_ctx* ctx_;
  bool alloc_ok_ = false;
  std::function<void()> keyframe_hook_;
  uint8_t* d_stage_ = nullptr; size_t stage_bytes_ = 0;  // process_host: left | right
  float K_[9];
  std::shared_ptr<FeatureTracker> feature_tracker;
  std::shared_ptr<BundleAdjuster> bundle_adjuster;
  float baseline, min_feature_distance, parallax_thresh;
  int max_corners_; double quality_; int max_batch_;
  float rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};  // CV_32F in the reference (src/image_processor.cpp:54-55)
  // batch state
  float* d_corners_ = nullptr; int* d_ncorners_ = nullptr; uint8_t* d_pyr_ = nullptr; size_t pyr_stride_ = 0;
  std::vector<int> h_ncorners_;
  int batch_ = 0, width_ = 0, height_ = 0;
  // scratch
  float *d_xyz_ = nullptr, *d_trk_xy_ = nullptr, *d_new_xy_ = nullptr, *d_disp_ = nullptr, *d_kxy_ = nullptr, *d_kxyz_ = nullptr;
  long long* d_trk_ids_ = nullptr; int *d_inl_ = nullptr, *d_cnt_ = nullptr;
  // pinned host arena (read / written in place by the kernels)
  uint8_t* h_arena_ = nullptr; float* h_xyz_ = nullptr; int* h_inl_ = nullptr; int* h_tri_cnt_ = nullptr; int* h_batch_cnt_ = nullptr;
  float *h_tri_xy_ = nullptr, *h_tri_xyz_ = nullptr;
  Stats stats_;
};

}  // namespace svo
#endif
