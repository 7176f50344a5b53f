// oracle: trigonometry with DECLARED arithmetic for the pose conversions of the keyframe chain (cv::Rodrigues and its inverse
// around cv::solvePnPRansac, reference src/image_processor.cpp:76-92,130-134).  TEST INFRASTRUCTURE ONLY.
// Restates stereo_vo_amd/host/det_trig.h operation by operation (the product never includes this file, this file never includes
// the product's): the oracle DEFINES these conversions for the unpinned PnP row (parity unpinned, svo_oracle.h) and the HIP path
// must match it bit for bit.  glibc's sin / cos / atan2 are not used: the device library's differ in the last place.
//   sin, cos: argument halved to <= 0.5, Taylor polynomials to x^15 / x^14 (Horner), double-angle steps back (as ora_ba.cpp's
//             det_sincos, which serves the local parameterization's Plus);
//   atan2 in the first quadrant: t = min / max; t > tan(pi/8): z = (t - 1) / (t + 1), atan t = pi/4 + atan z;
//             atan z = z (1 - w/3 + w^2/5 - ... - w^23/47), w = z z, Horner from the highest term; y > x: pi/2 - atan(x / y).
// Built with -ffp-contract=off: every operation is rounded separately.
#ifndef ORA_TRIG_H_
#define ORA_TRIG_H_
#include <cmath>

namespace ora_trig {
inline void sincos_det(double x, double* sn, double* cs) {
  int k = 0;
  while (x > 0.5) { x *= 0.5; ++k; }
  const double x2 = x * x;
  double s = x * (1.0 + x2 * (-1.0 / 6.0 + x2 * (1.0 / 120.0 + x2 * (-1.0 / 5040.0 + x2 * (1.0 / 362880.0 + x2 * (-1.0 / 39916800.0 +
             x2 * (1.0 / 6227020800.0 + x2 * (-1.0 / 1307674368000.0))))))));
  double c = 1.0 + x2 * (-0.5 + x2 * (1.0 / 24.0 + x2 * (-1.0 / 720.0 + x2 * (1.0 / 40320.0 + x2 * (-1.0 / 3628800.0 +
             x2 * (1.0 / 479001600.0 + x2 * (-1.0 / 87178291200.0)))))));
  for (int i = 0; i < k; ++i) {
    const double s2 = 2.0 * s * c;
    c = 1.0 - 2.0 * s * s;
    s = s2;
  }
  *sn = s; *cs = c;
}

inline double atan_small(double z) {
  const double w = z * z;
  double p = 1.0 / 47.0;
  for (int d = 45; d >= 3; d -= 2) p = 1.0 / (double)d - w * p;
  p = 1.0 - w * p;
  return z * p;
}

inline double atan01(double t) {
  if (t > 0.41421356237309503) return 0.78539816339744828 + atan_small((t - 1.0) / (t + 1.0));
  return atan_small(t);
}

inline double atan2_q1(double y, double x) {
  if (y <= x) return atan01(y / x);
  return 1.5707963267948966 - atan01(x / y);
}

// cv::Rodrigues on a CV_32F rvec: computed in double, stored as float
inline void rodrigues_f(const float* rv, float* R9) {
  const double rx = rv[0], ry = rv[1], rz = rv[2];
  const double th = std::sqrt(rx * rx + ry * ry + rz * rz);
  double R[9];
  if (th < 2.220446049250313e-16) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
  } else {
    double s, c;
    sincos_det(th, &s, &c);
    const double c1 = 1.0 - c, it = 1.0 / th;
    const double x = rx * it, y = ry * it, z = rz * it;
    R[0] = c + c1 * x * x; R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y; R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
  }
  for (int i = 0; i < 9; ++i) R9[i] = (float)R[i];
}

inline void quat_from_rvec(const double* rv, double* q /*wxyz*/) {
  const double th = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  if (th < 1e-12) { q[0] = 1; q[1] = 0.5 * rv[0]; q[2] = 0.5 * rv[1]; q[3] = 0.5 * rv[2]; return; }
  double s, c;
  sincos_det(0.5 * th, &s, &c);
  const double sn = s / th;
  q[0] = c; q[1] = sn * rv[0]; q[2] = sn * rv[1]; q[3] = sn * rv[2];
}

inline void rvec_from_quat(const double* q_in /*wxyz*/, double* rv) {
  double q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
  if (q[0] < 0) for (double& v : q) v = -v;
  const double vn = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (vn < 1e-12) { rv[0] = 2 * q[1]; rv[1] = 2 * q[2]; rv[2] = 2 * q[3]; return; }
  const double th = 2.0 * atan2_q1(vn, q[0]);
  rv[0] = q[1] / vn * th; rv[1] = q[2] / vn * th; rv[2] = q[3] / vn * th;
}
}  // namespace ora_trig
#endif
