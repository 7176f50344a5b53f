// CameraInfo — the boundary POD of the reference (src/camera_info.hpp:4-18), same fields in the same order so that
// `CameraInfo info = {focal, cx, cy, 0, 0, 0, 0, baseline}` (src/vo_node.cpp:110) keeps compiling and so that it can be
// handed to the C-ABI as an svo_camera_info (include/svo.h) without conversion.
#ifndef CAMERA_INFO_H_
#define CAMERA_INFO_H_

struct CameraInfo {
  double focal, cx, cy;   // intrinsics
  double k1, k2, p1, p2;  // distortion (never read on this path)
  double baseline;
};

#endif
