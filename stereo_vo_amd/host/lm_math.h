// Arithmetic of the LM step control that exists on both sides of the PCIe bus: the host loop (host/lm.cpp) and the
// controller workgroup of the device-resident solve (csrc/ba.hip, ba_lm_kernel) call THESE functions, so a solve that
// never leaves the GPU takes bit for bit the steps of the host-driven one.  Plain IEEE double arithmetic with a written
// operation order, no libm, no FMA contraction (-ffp-contract=off on both compilers).
// Semantics: SURVEY Appendix B (ceres::Solve of src/bundle_adjuster.cpp:140; the local parameterization of
// src/bundle_adjuster.cpp:19-20); the same sequences are restated in oracle/ora_ba.cpp.
#ifndef SVO_LM_MATH_H_
#define SVO_LM_MATH_H_
#include <math.h>

#include "lm_decide.h"  // SVO_HD

// sin/cos with a declared operation sequence: identical on every host and on the device
SVO_HD inline void svo_det_sincos(double x, double* sn, double* cs) {
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

// Plus of ProductParameterization(QuaternionParameterization, Identity(3)) (src/bundle_adjuster.cpp:19-20):
// q+ = [cos|d|, sin|d|/|d| d] (x) q, t+ = t + dt; no renormalisation.
SVO_HD inline void svo_plus_pose(const double* p, const double* d, double* out) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double qd[4];
  if (nd > 0) {
    double sn, cs;
    svo_det_sincos(nd, &sn, &cs);
    const double s = sn / nd;
    qd[0] = cs; qd[1] = s * d[0]; qd[2] = s * d[1]; qd[3] = s * d[2];
  } else { qd[0] = 1; qd[1] = qd[2] = qd[3] = 0; }
  const double* q = p;
  out[0] = qd[0] * q[0] - qd[1] * q[1] - qd[2] * q[2] - qd[3] * q[3];
  out[1] = qd[0] * q[1] + qd[1] * q[0] + qd[2] * q[3] - qd[3] * q[2];
  out[2] = qd[0] * q[2] - qd[1] * q[3] + qd[2] * q[0] + qd[3] * q[1];
  out[3] = qd[0] * q[3] + qd[1] * q[2] - qd[2] * q[1] + qd[3] * q[0];
  out[4] = p[4] + d[3]; out[5] = p[5] + d[4]; out[6] = p[6] + d[5];
}

#endif
