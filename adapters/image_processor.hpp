// Drop-in replacement of the reference's src/image_processor.hpp (StereoPair :9-17, ImageProcessor :19-83): same include
// name, class names and signatures.  process() is ONE call into libsvo_hip.so (svo::ImageProcessor::process_host):
// corner detection, LK tracking, PnP-RANSAC, dedup, StereoBM at the features, triangulation and the keyframe
// bookkeeping all run on the MI355X; triangulate_stereo is therefore no longer a member.
#ifndef IMAGE_PROCESSOR_H_
#define IMAGE_PROCESSOR_H_

#include "bundle_adjuster.hpp"
#include "feature_tracker.hpp"
#include <opencv2/core.hpp>
#include <Eigen/Dense>

struct StereoPair {  // src/image_processor.hpp:9-17 (src/vo_node.cpp:66,73 use .t and the 3-argument constructor)
  cv::Mat left;
  cv::Mat right;
  double t;

  StereoPair(const cv::Mat &left, const cv::Mat &right, double t) : left(left), right(right), t(t) {}
};

class ImageProcessor {
 public:
  // cam_mat: 3x3 CV_32F [f 0 cx; 0 f cy; 0 0 1] (src/vo_node.cpp:104-108); the other arguments as src/vo_node.cpp:114-119
  ImageProcessor(cv::Mat cam_mat, shared_ptr<FeatureTracker> tracker, shared_ptr<BundleAdjuster> adjuster, float bline,
                 float min_feature_distance, float parallax_thresh);
  ~ImageProcessor();

  void process(const StereoPair &stereo_pair);  // src/image_processor.cpp:18-163

  // ---- adapter plumbing: per-frame counters of the last process() call (tests / logging)
  const svo::ImageProcessor::Stats *last_stats() const { return impl_ ? &impl_->stats() : nullptr; }

 private:
  shared_ptr<BundleAdjuster> bundle_adjuster;
  shared_ptr<FeatureTracker> feature_tracker;
  cv::Mat camera_matrix;
  float baseline;
  float min_feature_distance;
  float parallax_thresh;
  unique_ptr<svo::ImageProcessor> impl_;
};

#endif
