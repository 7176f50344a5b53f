// Drop-in replacement of the reference's src/feature_tracker.hpp:20-54 (same include name, class name, signatures).
// Forward / backward pyramidal LK and the survivor filter run in libsvo_hip.so (svo::FeatureTracker -> lk_fb_kernel);
// the images stay in HBM, this object only keeps the host clone of the keyframe image that get_drawing() needs.
#ifndef FEATURE_TRACKER_H_
#define FEATURE_TRACKER_H_

#include <bundle_adjuster.hpp>
#include <unordered_map>

class FeatureTracker {
 public:
  FeatureTracker();  // src/vo_node.cpp:113 default-constructs it

  void init(const cv::Mat &image, const vector<cv::Point2f> &features, const vector<size_t> &ids);  // :3-16
  void track_features(float &av_parallax, float &percent_lost, const cv::Mat &image, bool flow_back);  // :18-67
  void get_tracked_features(vector<cv::Point2f> &features, vector<size_t> &ids);                       // :69-72
  void draw_track();      // :74-83
  cv::Mat get_drawing();  // :85-91 (src/vo_node.cpp:188)

  // ---- adapter plumbing (not part of the reference surface)
  const shared_ptr<svo::FeatureTracker> &impl() const { return impl_; }
  // ImageProcessor::process re-initialises the tracker inside the library; it reports the new keyframe image here
  // (after the arrows of the previous keyframe have been drawn, as src/image_processor.cpp:146,162 orders it)
  void finish_keyframe(const cv::Mat &keyframe_image);

 private:
  void rasterise();  // snapshot arrows of impl_ over initial_image -> track_drawing

  cv::Mat initial_image;
  cv::Mat track_drawing;
  shared_ptr<svo::FeatureTracker> impl_;
  unsigned drawn_serial = 0;
};

#endif
