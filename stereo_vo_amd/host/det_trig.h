// Trigonometry with DECLARED arithmetic for the pose conversions of the keyframe chain — cv::Rodrigues on rvec
// (reference src/image_processor.cpp:84-92,130-134: rvec -> R -> Quaternionf / hmat) and its inverse (the quaternion the
// PnP refinement returns -> rvec, stored as CV_32F between frames, :54-55) — so that the host, the kernels and the CPU oracle
// (oracle/ora_trig.h restates it) produce the same bits wherever the conversion runs.  glibc's and the device library's
// sin / cos / atan2 differ in the last place; until round 5 these conversions therefore had to stay on the host (libm on both
// sides of the parity check), which kept PnP and the stereo + triangulation launch apart.
//   sin, cos  svo_det_sincos (host/lm_math.h): halve the argument to <= 0.5, Taylor polynomials to x^15 / x^14 in Horner form,
//             double-angle steps back; arguments here are |rvec| and |rvec| / 2, never negative.
//   atan2     first quadrant only (y >= 0, x >= 0, not both 0: the callers flip the quaternion to w >= 0):
//             t = min / max in [0, 1]; t > tan(pi/8): z = (t - 1) / (t + 1), atan t = pi/4 + atan z; |z| <= tan(pi/8):
//             atan z = z (1 - w/3 + w^2/5 - ... - w^23/47), w = z z, Horner from the highest term; y > x: pi/2 - atan(x / y).
// Every operation is rounded separately (no FMA contraction: library, kernels and oracle are built with -ffp-contract=off).
// |det - libm| <= 4 ulp on the tested ranges (tests/test_pnp.py); results are stored as float by every caller.
#ifndef SVO_DET_TRIG_H_
#define SVO_DET_TRIG_H_
#include "lm_math.h"  // SVO_HD, svo_det_sincos

SVO_HD inline double svo_det_atan_small(double z) {  // |z| <= tan(pi/8)
  const double w = z * z;
  double p = 1.0 / 47.0;
  p = 1.0 / 45.0 - w * p; p = 1.0 / 43.0 - w * p; p = 1.0 / 41.0 - w * p; p = 1.0 / 39.0 - w * p; p = 1.0 / 37.0 - w * p;
  p = 1.0 / 35.0 - w * p; p = 1.0 / 33.0 - w * p; p = 1.0 / 31.0 - w * p; p = 1.0 / 29.0 - w * p; p = 1.0 / 27.0 - w * p;
  p = 1.0 / 25.0 - w * p; p = 1.0 / 23.0 - w * p; p = 1.0 / 21.0 - w * p; p = 1.0 / 19.0 - w * p; p = 1.0 / 17.0 - w * p;
  p = 1.0 / 15.0 - w * p; p = 1.0 / 13.0 - w * p; p = 1.0 / 11.0 - w * p; p = 1.0 / 9.0 - w * p;  p = 1.0 / 7.0 - w * p;
  p = 1.0 / 5.0 - w * p;  p = 1.0 / 3.0 - w * p;  p = 1.0 - w * p;
  return z * p;
}

SVO_HD inline double svo_det_atan01(double t) {  // t in [0, 1]
  if (t > 0.41421356237309503) {
    const double z = (t - 1.0) / (t + 1.0);
    return 0.78539816339744828 + svo_det_atan_small(z);
  }
  return svo_det_atan_small(t);
}

// atan2(y, x) for y >= 0, x >= 0, (x, y) != (0, 0)
SVO_HD inline double svo_det_atan2_q1(double y, double x) {
  if (y <= x) return svo_det_atan01(y / x);
  return 1.5707963267948966 - svo_det_atan01(x / y);
}

// cv::Rodrigues, rvec (CV_32F) -> R (row-major, float): evaluated in double, stored as float (src/image_processor.cpp:84,130)
SVO_HD inline void svo_det_rodrigues_f(const float* rv, float* R9) {
  const double rx = rv[0], ry = rv[1], rz = rv[2];
  const double th = sqrt(rx * rx + ry * ry + rz * rz);
  double R[9];
  if (th < 2.220446049250313e-16) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
  } else {
    double s, c;
    svo_det_sincos(th, &s, &c);
    const double c1 = 1.0 - c, it = 1.0 / th;
    const double x = rx * it, y = ry * it, z = rz * it;
    R[0] = c + c1 * x * x; R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y; R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
  }
  for (int i = 0; i < 9; ++i) R9[i] = (float)R[i];
}

// rvec (as doubles) -> the unit quaternion the PnP solver starts from (extrinsic guess, src/image_processor.cpp:76-80)
SVO_HD inline void svo_det_quat_from_rvec(const double* rv, double* q /*wxyz*/) {
  const double th = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  if (th < 1e-12) { q[0] = 1; q[1] = 0.5 * rv[0]; q[2] = 0.5 * rv[1]; q[3] = 0.5 * rv[2]; return; }
  double s, c;
  svo_det_sincos(0.5 * th, &s, &c);
  const double sn = s / th;
  q[0] = c; q[1] = sn * rv[0]; q[2] = sn * rv[1]; q[3] = sn * rv[2];
}

// the quaternion the PnP refinement returns -> rvec (double; every caller stores it as float)
SVO_HD inline void svo_det_rvec_from_quat(const double* q_in /*wxyz*/, double* rv) {
  double q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
  if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double vn = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (vn < 1e-12) { rv[0] = 2 * q[1]; rv[1] = 2 * q[2]; rv[2] = 2 * q[3]; return; }
  const double th = 2.0 * svo_det_atan2_q1(vn, q[0]);
  rv[0] = q[1] / vn * th; rv[1] = q[2] / vn * th; rv[2] = q[3] / vn * th;
}
#endif
