// Drop-in replacement of the reference's src/bundle_adjuster.hpp: same include name, same public names and signatures
// (Keyframe :22-46, max_features :75, BundleAdjuster :77-126), so src/vo_node.cpp and src/kitti_node.cpp compile
// untouched (they use: make_shared<BundleAdjuster>(size_t, CameraInfo) vo_node.cpp:112, get_last_keyframe() :146,148,
// bundle_adjust() :147, Keyframe::orientation / ::position :149-150, plus the two using-directives below).
// The Ceres problem, the PoseVariable / Feature parameter blocks and remove_oldest_pose are gone: the sliding-window
// graph and the LM / Schur solve live in libsvo_hip.so (svo::BundleAdjuster -> svo_ba_* in include/svo.h).
#ifndef BUNDLE_ADJUSTER_H_
#define BUNDLE_ADJUSTER_H_

#include <Eigen/Dense>
#include <opencv2/core.hpp>
#include <cstdlib>
#include <functional>
#include <memory>
#include <queue>
#include <stack>
#include <vector>

#include "camera_info.hpp"
#include "svo_adapter.hpp"

using namespace std;    // the reference header exports both (src/bundle_adjuster.hpp:12-13) and vo_node.cpp relies on it
using namespace Eigen;

// Pose of the WORLD frame w.r.t. the CAMERA frame plus the keyframe's features (src/bundle_adjuster.hpp:15-46).
struct Keyframe {
  Vector3f position;
  Quaternionf orientation;
  cv::Mat image;
  vector<cv::Point2f> tracked_features_2d;
  vector<size_t> tracked_ids;
  vector<cv::Point2f> new_features_2d;
  vector<cv::Point3f> new_features_3d;
  vector<size_t> new_ids;

  Keyframe(Vector3f position, Quaternionf orientation, cv::Mat image, vector<cv::Point2f> tracked_features_2d,
           vector<size_t> tracked_ids, vector<cv::Point2f> new_features_2d, vector<cv::Point3f> new_features_3d)
      : position(position), orientation(orientation), image(image), tracked_features_2d(tracked_features_2d),
        tracked_ids(tracked_ids), new_features_2d(new_features_2d), new_features_3d(new_features_3d) {}
};

static const size_t max_features = 400;  // src/bundle_adjuster.hpp:75

class BundleAdjuster {
 public:
  BundleAdjuster(size_t _window_size, CameraInfo info);  // src/bundle_adjuster.hpp:86
  ~BundleAdjuster();

  // last keyframe passed to add_keyframe (by ImageProcessor::process or directly), pose refreshed by bundle_adjust()
  inline shared_ptr<Keyframe> get_last_keyframe() {
    sync_from_impl();
    return last_keyframe;
  }
  // src/bundle_adjuster.cpp:60-135: ids for the new features, truncation to max_features, window pop.  MUTATES
  // *keyframe exactly as the reference does (new_features_2d/3d truncated :85-90, new_ids filled :115).
  void add_keyframe(shared_ptr<Keyframe> keyframe);
  void bundle_adjust();  // src/bundle_adjuster.cpp:137-157
  void get_world_points(vector<cv::Point3f> &world_points, const vector<size_t> &ids);  // :159-163

  // ---- adapter plumbing (not part of the reference surface)
  const shared_ptr<svo::BundleAdjuster> &impl() const { return impl_; }
  void note_keyframe_image(const cv::Mat &image) { pending_image = image; }  // ImageProcessor: image of the keyframe it just added

 private:
  void sync_from_impl();

  shared_ptr<svo::BundleAdjuster> impl_;
  shared_ptr<Keyframe> last_keyframe;
  shared_ptr<svo::Keyframe> mirrored;  // the library-side keyframe `last_keyframe` mirrors
  cv::Mat pending_image;
  size_t window_size;
  CameraInfo camera_info;
};

#endif
