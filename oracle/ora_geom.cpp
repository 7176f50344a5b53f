// oracle: first-party per-feature loops of the reference.  TEST INFRASTRUCTURE ONLY.
// These follow reference source directly (no third-party algorithm involved).
#include <cmath>

#include "ora_constants.h"
#include "svo_oracle.h"

// ImageProcessor::triangulate_stereo, src/image_processor.cpp:178-207.
//   Q rows: [1/f 0 0 -cx/f], [0 1/f 0 -cy/f], [0 0 0 1], [0 0 1/(b f) 0]      (:183-189)
//   keep iff disp > 0 (:194); world = camera_pose * Q * [x y disp 1]^T (:202); divide by w (:203-205).
// cv::Mat float products accumulate in double and round once per element (OpenCV gemm, float
// data / double work type); (camera_pose*Q) is formed first, then applied to the point.
extern "C" int ora_triangulate(const float* xy, const float* disp, int n, const float* pose16,
                               float focal, float cx, float cy, float baseline, float* kept_xy,
                               float* xyz, int* kept_index) {
  float Q[16] = {0};
  Q[0] = (float)(1.0 / (double)focal);
  Q[5] = (float)(1.0 / (double)focal);
  Q[3] = -cx / focal;
  Q[7] = -cy / focal;
  Q[11] = 1.0f;
  Q[14] = (float)(1.0 / (double)(baseline * focal));
  float M[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += (double)pose16[4 * i + k] * (double)Q[4 * k + j];
      M[4 * i + j] = (float)s;
    }
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float d = disp[i];
    if (!(d > ora_k::kTriangulateMinDisparity)) continue;
    const float v[4] = {xy[2 * i], xy[2 * i + 1], d, 1.0f};
    float wv[4];
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += (double)M[4 * r + k] * (double)v[k];
      wv[r] = (float)s;
    }
    kept_xy[2 * m] = xy[2 * i];
    kept_xy[2 * m + 1] = xy[2 * i + 1];
    xyz[3 * m] = wv[0] / wv[3];
    xyz[3 * m + 1] = wv[1] / wv[3];
    xyz[3 * m + 2] = wv[2] / wv[3];
    if (kept_index) kept_index[m] = i;
    ++m;
  }
  return m;
}

// new-vs-tracked dedup, src/image_processor.cpp:113-128:
//   drop detected[i] if any tracked[j] has sqrt(dx*dx + dy*dy) < min_feature_distance (float).
extern "C" int ora_dedup(const float* det_xy, int n_det, const float* trk_xy, int n_trk,
                         float min_distance, float* kept_xy) {
  int m = 0;
  for (int i = 0; i < n_det; ++i) {
    bool tracked = false;
    for (int j = 0; j < n_trk; ++j) {
      const float dx = det_xy[2 * i] - trk_xy[2 * j];
      const float dy = det_xy[2 * i + 1] - trk_xy[2 * j + 1];
      if (std::sqrt(dx * dx + dy * dy) < min_distance) { tracked = true; break; }
    }
    if (!tracked) {
      kept_xy[2 * m] = det_xy[2 * i];
      kept_xy[2 * m + 1] = det_xy[2 * i + 1];
      ++m;
    }
  }
  return m;
}
