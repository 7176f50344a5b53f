// The pose chain between cv::solvePnPRansac and triangulate_stereo (reference src/image_processor.cpp:84-92,130-134,183-189,202):
// rvec (CV_32F) -> R (cv::Rodrigues) -> hmat = [R^T | -R^T t] -> M = float(hmat * Q), in ONE function that the host (host/group.cpp,
// the stereo + triangulation launch of a lane that did not run PnP) and the device (csrc/pnp.hip: the PnP launch's last workgroup
// leaves M for the stereo launch queued right behind it — round 5, one host turn less per keyframe) both call: the same operations in
// the same order, hence the same bits.  Trigonometry: host/det_trig.h (declared arithmetic).  Q as svo_k_reprojection_matrix
// (csrc/geom.hip) forms it from the reference's six assignments.
#ifndef SVO_CHAIN_MATH_H_
#define SVO_CHAIN_MATH_H_
#include "det_trig.h"

// device record a lane's PnP launch leaves for the stereo + triangulation launch behind it
struct SvoChainRec {
  int best;     // the RANSAC bookkeeping's verdict: >= 0 model found, -1 none, -2 more hypotheses needed (the stereo launch then does nothing)
  int n_inl;    // inliers of the refined model (0 without a model): how many entries of the lane's inlier list the dedup tests against
  int pad[2];
  float M[16];  // reprojection matrix of this keyframe's camera pose
};

// rmat: cv::Rodrigues(rvec) (row-major, float) as the caller holds it; M16 = float(hmat * Q)
SVO_HD inline void svo_chain_matrix_from_R(const float* rmat, const float* tvec, float focal, float cx, float cy, float baseline, float* M16) {
  float pose[16];
  for (int i = 0; i < 16; ++i) pose[i] = 0.f;
  for (int r = 0; r < 3; ++r) {  // hmat = [R^T | -R^T t]  :130-134 (float Mats; the product accumulates in double)
    for (int c = 0; c < 3; ++c) pose[4 * r + c] = rmat[3 * c + r];
    double s = 0.0;
    for (int c = 0; c < 3; ++c) s += (double)(-rmat[3 * c + r]) * (double)tvec[c];
    pose[4 * r + 3] = (float)s;
  }
  pose[15] = 1.f;
  float Q[16];
  for (int i = 0; i < 16; ++i) Q[i] = 0.f;
  Q[0] = (float)(1.0 / (double)focal);
  Q[5] = (float)(1.0 / (double)focal);
  Q[3] = -cx / focal;
  Q[7] = -cy / focal;
  Q[11] = 1.0f;
  Q[14] = (float)(1.0 / (double)(baseline * focal));
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += (double)pose[4 * i + k] * (double)Q[4 * k + j];
      M16[4 * i + j] = (float)s;
    }
}
SVO_HD inline void svo_chain_matrix(const float* rvec, const float* tvec, float focal, float cx, float cy, float baseline, float* M16) {
  float rmat[9];
  svo_det_rodrigues_f(rvec, rmat);
  svo_chain_matrix_from_R(rmat, tvec, focal, cx, cy, baseline, M16);
}
#endif
