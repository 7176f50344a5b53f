// oracle: ReprojectionFactor::Evaluate restated in compact vector form.
// TEST INFRASTRUCTURE ONLY (see svo_oracle.h).  PINNED by tests/golden/reproj_golden.json.
//
// Reference: src/reprojection_factor.cpp:10-88.
//   gamma = (v v^T + (w I + [v]x)^2) p / |q|^2 + t          (:24-33)
//   r     = [f 0 cx; 0 f cy] * gamma / gamma_z - obs         (:35-38)
//   jacobians[0] = d r / d [qw qx qy qz tx ty tz], 2x7 row-major, entries 5 and 11 zero (:60-75)
//   jacobians[1] = d r / d p, 2x3 row-major                                         (:77-84)
// The reference spells the derivatives out as MATLAB-generated scalar expressions; here they are
// derived by the chain rule  d r/d x = (d r/d gamma)(d gamma/d x), which is algebraically identical
// (the derivative includes the 1/|q|^2 factor, i.e. it is exact for non-unit q).
#include "svo_oracle.h"

namespace {
struct V3 { double x, y, z; };
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
}  // namespace

extern "C" void ora_reproj_eval(int n, const double* pose7, const double* point3,
                                const double* obs2, double f, double cx, double cy, double* r2,
                                double* jpose14, double* jpoint6) {
  for (int i = 0; i < n; ++i) {
    const double* q = pose7 + 7 * i;
    const double w = q[0];
    const V3 v{q[1], q[2], q[3]};
    const V3 t{q[4], q[5], q[6]};
    const V3 p{point3[3 * i], point3[3 * i + 1], point3[3 * i + 2]};
    const double nn = w * w + dot(v, v);
    const double inv_n = 1.0 / nn;
    const V3 u = add(scale(p, w), cross(v, p));                       // (wI+[v]x) p
    const double vp = dot(v, p);
    const V3 Mp = add(add(scale(v, vp), scale(u, w)), cross(v, u));   // |q|^2 R p
    const V3 g = add(scale(Mp, inv_n), t);
    const double psi = 1.0 / g.z;
    r2[2 * i] = f * g.x * psi + cx - obs2[2 * i];
    r2[2 * i + 1] = f * g.y * psi + cy - obs2[2 * i + 1];
    if (!jpose14 && !jpoint6) continue;
    // d r / d gamma (2x3)
    const double a = f * psi;
    const double bx = -f * g.x * psi * psi;
    const double by = -f * g.y * psi * psi;
    if (jpose14) {
      double* J = jpose14 + 14 * i;
      const double s2 = 2.0 * inv_n * inv_n;
      // d gamma / d w
      V3 dq[4];
      dq[0] = add(scale(u, 2.0 * inv_n), scale(Mp, -s2 * w));
      const V3 e[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
      const double pv[3] = {p.x, p.y, p.z};
      const double vv[3] = {v.x, v.y, v.z};
      for (int k = 0; k < 3; ++k) {
        const V3 ep = cross(e[k], p);
        V3 d = add(scale(v, pv[k]), scale(e[k], vp));
        d = add(d, scale(ep, w));
        d = add(d, cross(e[k], u));
        d = add(d, cross(v, ep));
        dq[k + 1] = add(scale(d, inv_n), scale(Mp, -s2 * vv[k]));
      }
      for (int k = 0; k < 4; ++k) {
        J[k] = a * dq[k].x + bx * dq[k].z;
        J[7 + k] = a * dq[k].y + by * dq[k].z;
      }
      J[4] = a;  J[5] = 0.0; J[6] = bx;
      J[11] = 0.0; J[12] = a; J[13] = by;
    }
    if (jpoint6) {
      double* J = jpoint6 + 6 * i;
      // R = (2 v v^T + (w^2 - |v|^2) I + 2 w [v]x) / |q|^2
      const double dgl = w * w - dot(v, v);
      const double R[3][3] = {
          {(2 * v.x * v.x + dgl) * inv_n, (2 * v.x * v.y - 2 * w * v.z) * inv_n, (2 * v.x * v.z + 2 * w * v.y) * inv_n},
          {(2 * v.y * v.x + 2 * w * v.z) * inv_n, (2 * v.y * v.y + dgl) * inv_n, (2 * v.y * v.z - 2 * w * v.x) * inv_n},
          {(2 * v.z * v.x - 2 * w * v.y) * inv_n, (2 * v.z * v.y + 2 * w * v.x) * inv_n, (2 * v.z * v.z + dgl) * inv_n}};
      for (int k = 0; k < 3; ++k) {
        J[k] = a * R[0][k] + bx * R[2][k];
        J[3 + k] = a * R[1][k] + by * R[2][k];
      }
    }
  }
}
