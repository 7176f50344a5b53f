// The arithmetic of csrc/reproj_device.h (ReprojectionFactor::Evaluate, reference src/reprojection_factor.cpp:10-88), without an
// include guard: it is compiled TWICE — once as is (no FMA contraction: the bits of the oracle, every parity-exact path) and
// once with the suffix _c (SVO_RD) under `#pragma clang fp contract(fast)` for the bulk kernels whose sums are
// hardware-ordered anyway (tolerance-level parity; the separate multiply + add halves the f64 rate of these expressions).
__device__ __forceinline__ D3 SVO_RD(d3_cross)(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ D3 SVO_RD(d3_add)(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 SVO_RD(d3_scale)(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double SVO_RD(d3_dot)(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Residual only.
__device__ __forceinline__ void SVO_RD(reproj_residual)(const double* __restrict__ q, D3 p, double ox, double oy,
                                                double f, double cx, double cy, double& r0, double& r1) {
  const double w = q[0];
  const D3 v{q[1], q[2], q[3]};
  const D3 t{q[4], q[5], q[6]};
  const double inv_n = 1.0 / (w * w + SVO_RD(d3_dot)(v, v));
  const D3 u = SVO_RD(d3_add)(SVO_RD(d3_scale)(p, w), SVO_RD(d3_cross)(v, p));
  const D3 Mp = SVO_RD(d3_add)(SVO_RD(d3_add)(SVO_RD(d3_scale)(v, SVO_RD(d3_dot)(v, p)), SVO_RD(d3_scale)(u, w)), SVO_RD(d3_cross)(v, u));
  const D3 g = SVO_RD(d3_add)(SVO_RD(d3_scale)(Mp, inv_n), t);
  const double psi = 1.0 / g.z;
  r0 = f * g.x * psi + cx - ox;
  r1 = f * g.y * psi + cy - oy;
}

// Residual + optional 2x7 (Jq, may be null) + optional 2x3 (Jx, may be null), all row-major.
__device__ __forceinline__ void SVO_RD(reproj_full)(const double* __restrict__ q, D3 p, double ox, double oy, double f,
                                            double cx, double cy, double* r, double* Jq, double* Jx) {
  const double w = q[0];
  const D3 v{q[1], q[2], q[3]};
  const D3 t{q[4], q[5], q[6]};
  const double nn = w * w + SVO_RD(d3_dot)(v, v);
  const double inv_n = 1.0 / nn;
  const D3 u = SVO_RD(d3_add)(SVO_RD(d3_scale)(p, w), SVO_RD(d3_cross)(v, p));
  const double vp = SVO_RD(d3_dot)(v, p);
  const D3 Mp = SVO_RD(d3_add)(SVO_RD(d3_add)(SVO_RD(d3_scale)(v, vp), SVO_RD(d3_scale)(u, w)), SVO_RD(d3_cross)(v, u));
  const D3 g = SVO_RD(d3_add)(SVO_RD(d3_scale)(Mp, inv_n), t);
  const double psi = 1.0 / g.z;
  r[0] = f * g.x * psi + cx - ox;
  r[1] = f * g.y * psi + cy - oy;
  const double a = f * psi;
  const double bx = -f * g.x * psi * psi;
  const double by = -f * g.y * psi * psi;
  if (Jq) {
    const double s2 = 2.0 * inv_n * inv_n;
    D3 dq = SVO_RD(d3_add)(SVO_RD(d3_scale)(u, 2.0 * inv_n), SVO_RD(d3_scale)(Mp, -s2 * w));
    Jq[0] = a * dq.x + bx * dq.z;
    Jq[7] = a * dq.y + by * dq.z;
    const double pv[3] = {p.x, p.y, p.z};
    const double vv[3] = {v.x, v.y, v.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const D3 e{k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
      const D3 ep = SVO_RD(d3_cross)(e, p);
      D3 d = SVO_RD(d3_add)(SVO_RD(d3_scale)(v, pv[k]), SVO_RD(d3_scale)(e, vp));
      d = SVO_RD(d3_add)(d, SVO_RD(d3_scale)(ep, w));
      d = SVO_RD(d3_add)(d, SVO_RD(d3_cross)(e, u));
      d = SVO_RD(d3_add)(d, SVO_RD(d3_cross)(v, ep));
      dq = SVO_RD(d3_add)(SVO_RD(d3_scale)(d, inv_n), SVO_RD(d3_scale)(Mp, -s2 * vv[k]));
      Jq[1 + k] = a * dq.x + bx * dq.z;
      Jq[8 + k] = a * dq.y + by * dq.z;
    }
    Jq[4] = a;   Jq[5] = 0.0; Jq[6] = bx;
    Jq[11] = 0.0; Jq[12] = a; Jq[13] = by;
  }
  if (Jx) {
    const double dgl = w * w - SVO_RD(d3_dot)(v, v);
    const double R00 = (2 * v.x * v.x + dgl) * inv_n, R01 = (2 * v.x * v.y - 2 * w * v.z) * inv_n, R02 = (2 * v.x * v.z + 2 * w * v.y) * inv_n;
    const double R10 = (2 * v.y * v.x + 2 * w * v.z) * inv_n, R11 = (2 * v.y * v.y + dgl) * inv_n, R12 = (2 * v.y * v.z - 2 * w * v.x) * inv_n;
    const double R20 = (2 * v.z * v.x - 2 * w * v.y) * inv_n, R21 = (2 * v.z * v.y + 2 * w * v.x) * inv_n, R22 = (2 * v.z * v.z + dgl) * inv_n;
    Jx[0] = a * R00 + bx * R20; Jx[1] = a * R01 + bx * R21; Jx[2] = a * R02 + bx * R22;
    Jx[3] = a * R10 + by * R20; Jx[4] = a * R11 + by * R21; Jx[5] = a * R12 + by * R22;
  }
}
