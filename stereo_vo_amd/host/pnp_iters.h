// OpenCV's RANSACUpdateNumIters (the adaptive iteration cap inside cv::solvePnPRansac, reference call site
// src/image_processor.cpp:76-80) with DECLARED arithmetic, so that the host, the kernels and the CPU oracle
// (oracle/ora_pnp.cpp restates it) take the same cut-off bit for bit wherever the bookkeeping runs:
//   (1 - ep)^k   = k - 1 multiplications, left to right;
//   log(x)       = e ln2 + 2 s (1 + z/3 + z^2/5 + ... + z^12/25),  x = m 2^e with m in [sqrt(1/2), sqrt(2)),
//                  s = (m - 1) / (m + 1), z = s s, Horner from the highest term; every operation rounded separately
//                  (no FMA contraction: the library, the kernels and the oracle are all built with -ffp-contract=off).
// |det_log - log| < 5e-16 relative on (0, 1] (tests/test_pnp.py); libm's log / pow differ between glibc and the device
// library in the last place, which is why they are not used here.
#ifndef SVO_PNP_ITERS_H_
#define SVO_PNP_ITERS_H_
#include <float.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SVO_HD __host__ __device__
#else
#define SVO_HD
#endif

SVO_HD inline double svo_det_powi(double x, int k) {
  double r = 1.0;
  for (int i = 0; i < k; ++i) r = r * x;
  return r;
}

// natural logarithm of a positive, finite, normal double
SVO_HD inline double svo_det_log(double x) {
  uint64_t bits;
  memcpy(&bits, &x, 8);
  int e = (int)((bits >> 52) & 0x7ffu) - 1022;  // x = m 2^e, m in [0.5, 1)
  bits = (bits & 0x000fffffffffffffull) | 0x3fe0000000000000ull;
  double m;
  memcpy(&m, &bits, 8);
  if (m < 0.70710678118654757) { m = m * 2.0; e -= 1; }  // exact
  const double s = (m - 1.0) / (m + 1.0), z = s * s;
  double p = 1.0 / 25.0;
  p = p * z + 1.0 / 23.0; p = p * z + 1.0 / 21.0; p = p * z + 1.0 / 19.0; p = p * z + 1.0 / 17.0;
  p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0; p = p * z + 1.0 / 11.0; p = p * z + 1.0 / 9.0;
  p = p * z + 1.0 / 7.0;  p = p * z + 1.0 / 5.0;  p = p * z + 1.0 / 3.0;  p = p * z + 1.0;
  const double t1 = (double)e * 0.6931471805599453, t2 = 2.0 * s, t3 = t2 * p;
  return t1 + t3;
}

// round half to even of a finite double that fits an int (what lrint does in the default rounding mode)
SVO_HD inline int svo_det_lrint(double v) {
  const double f = v < 0 ? -v : v;
  long long i = (long long)f;  // truncation
  const double frac = f - (double)i;
  if (frac > 0.5 || (frac == 0.5 && (i & 1))) ++i;
  return (int)(v < 0 ? -i : i);
}

SVO_HD inline int svo_pnp_update_num_iters_det(double p, double ep, int model_points, int max_iters) {
  p = p > 0.0 ? p : 0.0; p = p < 1.0 ? p : 1.0;
  ep = ep > 0.0 ? ep : 0.0; ep = ep < 1.0 ? ep : 1.0;
  double num = 1.0 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  double denom = 1.0 - svo_det_powi(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = svo_det_log(num);
  denom = svo_det_log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : svo_det_lrint(num / denom);
}
#endif  // SVO_PNP_ITERS_H_
