// Shared plumbing of the drop-in adapters: one svo_ctx per process (the reference's classes take no context argument,
// src/vo_node.cpp:112-119) and the cv::Mat <-> plain-pointer helpers.  Only `rows/cols/data/step/type()/empty()/
// clone()/create()/at<>()` of cv::Mat are used, i.e. opencv2/core.hpp is all of OpenCV the adapters need.
#ifndef SVO_ADAPTER_HPP_
#define SVO_ADAPTER_HPP_
#include <opencv2/core.hpp>

#include "camera_info.hpp"
#include "stereo_vo.hpp"  // this repository: stereo_vo_amd/host/stereo_vo.hpp
#include "svo.h"          // this repository: include/svo.h

namespace svo_adapter {

// Lazily created process-wide context.  Limits: SVO_ADAPTER_MAX_WIDTH / _MAX_HEIGHT (default 1920 x 1200, covers every
// config/*.yaml of the reference), 300 corners (src/image_processor.cpp:22), 400 features (src/bundle_adjuster.hpp:75),
// device SVO_ADAPTER_DEVICE (default 0).  Returns nullptr (and every adapter method then returns silently, the
// reference's error convention) when no MI355X is visible: there is no CPU fallback.
svo_ctx* context();
// Destroys the process-wide context (tests; a ROS node simply exits).
void shutdown();

// The reference hard-codes Ceres' max_solver_time_in_seconds = 0.1 (src/bundle_adjuster.cpp:11), which makes results
// depend on wall-clock time.  SVO_ADAPTER_BA_MAX_TIME_S overrides it (<= 0: iteration cap only, reproducible).
double ba_max_time_s();

inline svo_camera_info to_svo(const CameraInfo& c) {
  static_assert(sizeof(CameraInfo) == sizeof(svo_camera_info), "CameraInfo and svo_camera_info share one layout");
  svo_camera_info o;
  o.focal = c.focal; o.cx = c.cx; o.cy = c.cy; o.k1 = c.k1; o.k2 = c.k2; o.p1 = c.p1; o.p2 = c.p2; o.baseline = c.baseline;
  return o;
}

// mono8 view of a cv::Mat; false unless CV_8UC1 and non-empty
inline bool mono8(const cv::Mat& m, const uint8_t** data, int* width, int* height, int* stride) {
  if (m.empty() || m.type() != CV_8UC1) return false;
  *data = m.data; *width = m.cols; *height = m.rows; *stride = (int)(size_t)m.step;
  return true;
}

}  // namespace svo_adapter
#endif
