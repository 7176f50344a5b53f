// a9/a10/a12/a13 — sliding-window bundle adjustment: BundleAdjuster (reference src/bundle_adjuster.cpp:5-163)
// whose solve is ceres::Solve with DENSE_SCHUR (:9-12,140) over ReprojectionFactor residuals
// (src/reprojection_factor.cpp:10-88), quaternion (x) identity local parameterization (:19-20,123),
// oldest pose constant (:130).  LM semantics: SURVEY.md Appendix B as restated in oracle/ora_ba.cpp.
//
// Per LM iteration two kernels, each a single pass over the observations (landmark-major CSR):
//   ba_linearize_kernel : lane = observation.  Residual + analytic Jacobians (FP64 VALU, fused device
//       function, nothing written back), per-landmark V / g_p by in-wave segment gathers, 3x3 inverse,
//       Y = W s Vd^-1, and the landmark's Schur contribution -Y_k (W_k' s)^T accumulated into a
//       per-workgroup LDS image of the reduced camera system (ds_add_f64), flushed once per workgroup
//       with global f64 atomics into payload1 = [S | g_red | g_c | diag U | cost | sum g_p^2].
//   ba_backsub_kernel   : recomputes the landmark blocks (cheaper than 144 B/observation of W traffic),
//       back-substitutes the camera step, writes candidate points and evaluates the candidate cost in
//       the same pass -> payload2 = [cost_new | model-change(points) | sum dp^2 | sum p^2].
// Deterministic mode (window-sized problems, i.e. everything the pipeline solves): instead of LDS/global
// atomics the kernels write per-pair 6x6 blocks, per-observation vectors and per-landmark scalars to
// contribution slots, and ba_reduce1/2_kernel sum every destination with the DECLARED order
// "28 consecutive segments of ceil(len/28) entries summed sequentially, then the 28 segment sums added
// sequentially" over its slot list in landmark order (one lane per (segment, element)).  The oracle performs the
// same sums in the same order, so the whole LM trajectory — and therefore every later PnP inlier set — is
// bit-identical between CPU and GPU and independent of grid size.  (Needed because the reference's
// problem has a scale gauge: with one fixed pose and only reprojection factors the iterates slide along
// a flat direction and amplify any summation-order difference; measured 3e-2 pose drift otherwise.)
// Problems whose pair slots would not fit (config 4) keep the atomic path and a tolerance-level result.
// The n x n (n = 6 (K-1) <= 114) Cholesky, step control and termination run on the host from the
// (all-reduced) payloads, so every rank of a sharded run takes identical decisions.
// A rank of a sharded run holds all poses and its own landmarks; `allreduce` sums payload1/2 in place
// on the device (RCCL all-reduce over xGMI) — the only exchange of the path.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <memory>
#include <vector>

#include "kernels.h"
#include "reproj_device.h"

namespace {
constexpr double MIN_DIAG = 1e-6, MAX_DIAG = 1e32, MAX_RADIUS = 1e16, MIN_RADIUS = 1e-32, MIN_REL_DECREASE = 1e-3;

struct BaDev {
  int K = 0, n = 0, M = 0, L = 0, C = 0;
  double* poses = nullptr;       // K x 7 (linearisation point)
  double* cand_poses = nullptr;  // K x 7
  double* dc = nullptr;          // n
  double* points = nullptr;      // Npts x 3
  double* cand_points = nullptr;
  int32_t* obs_pose = nullptr;
  int32_t* obs_point = nullptr;
  double* obs_uv = nullptr;
  int32_t* lm_start = nullptr;   // per landmark index j (dense over [0,Npts]): first obs; lm_start[j+1] end
  int32_t* chunk_start = nullptr;
  double* sp = nullptr;          // Npts x 3 point Jacobi scales
  double* pay1 = nullptr;
  double* pay2 = nullptr;
  double f = 0, cx = 0, cy = 0;
  // deterministic mode: contribution slots + destination lists
  int det = 0;
  // Contributions are stored DESTINATION-ORDERED so the reduce kernels stream contiguous memory:
  int32_t* pair_base = nullptr;   // per observation: first pair slot (pairs (o, t>=o) of its landmark)
  int32_t* pair_pos = nullptr;    // per pair slot: [position in its block list, position in the mirrored list or -1]
  int32_t* obs_pos = nullptr;     // per observation: position in its pose list or -1
  double* pairB = nullptr;        // (sum of block-list lengths) x 36, block lists back to back
  double* obsV = nullptr;         // (free observations) x 18  (g_c | g_red part | diag U), pose lists back to back
  double* lmV = nullptr;          // Npts x 4, by landmark index (zero for landmarks without observations)
  int32_t* list_start = nullptr;  // F*F + F + 1 entries (+1): offsets into pairB / obsV / lmV rows
  double* pay1_out = nullptr;     // where the reduce kernels write (pinned host memory when single-rank)
  double* pay2_out = nullptr;
};

__device__ __forceinline__ bool inv3_sym(const double* V, double* Vi) {
  const double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(fabs(det) > 0)) { for (int i = 0; i < 9; ++i) Vi[i] = 0.0; return false; }
  const double id = 1.0 / det;
  Vi[0] = c00 * id; Vi[1] = c01 * id; Vi[2] = c02 * id;
  Vi[3] = Vi[1]; Vi[4] = (a * f - c * c) * id; Vi[5] = (b * c - a * e) * id;
  Vi[6] = Vi[2]; Vi[7] = Vi[5]; Vi[8] = (a * d - b * b) * id;
  return true;
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }

// residual + tangent Jacobians of one observation
__device__ __forceinline__ void eval_obs(const double* __restrict__ pose, D3 p, double u, double v, double f, double cx,
                                         double cy, bool want_jc, double* r, double* Jc, double* Jp) {
  double Jq[14];
  reproj_full(pose, p, u, v, f, cx, cy, r, want_jc ? Jq : nullptr, Jp);
  if (want_jc) {
    const double w = pose[0], x = pose[1], y = pose[2], z = pose[3];
    const double T[4][3] = {{-x, -y, -z}, {w, z, -y}, {-z, w, x}, {y, -x, w}};
#pragma unroll
    for (int row = 0; row < 2; ++row) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += Jq[7 * row + k] * T[k][c];
        Jc[6 * row + c] = s;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) Jc[6 * row + 3 + c] = Jq[7 * row + 4 + c];
    }
  }
}
}  // namespace

__global__ __launch_bounds__(256) void ba_linearize_kernel(BaDev P, double radius, int first_pass) {
  extern __shared__ double lds[];  // payload1 image: S (n*n) | gred (n) | gc (n) | dU (n) | cost | gp2
  const int n = P.n;
  const int pay1 = n * n + 3 * n + 2;
  double* sS = lds;
  double* sGred = sS + n * n;
  double* sGc = sGred + n;
  double* sDU = sGc + n;
  if (!P.det) {
    for (int i = threadIdx.x; i < pay1; i += blockDim.x) lds[i] = 0.0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double lcost = 0.0, lgp2 = 0.0;
  const int wpb = blockDim.x >> 6;
  for (int chunk = blockIdx.x * wpb + wave; chunk < P.C; chunk += gridDim.x * wpb) {
    const int c0 = P.chunk_start[chunk], c1 = P.chunk_start[chunk + 1];
    const int o = c0 + lane;
    const bool active = o < c1;
    int k = 0, j = 0, first = lane, len = 0;
    double r[2] = {0, 0}, Jc[12], Jp[6];
#pragma unroll
    for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) Jp[i] = 0.0;
    if (active) {
      k = P.obs_pose[o]; j = P.obs_point[o];
      first = P.lm_start[j] - c0; len = P.lm_start[j + 1] - P.lm_start[j];
      const D3 p{P.points[3 * j], P.points[3 * j + 1], P.points[3 * j + 2]};
      eval_obs(P.poses + 7 * k, p, P.obs_uv[2 * o], P.obs_uv[2 * o + 1], P.f, P.cx, P.cy, k > 0, r, Jc, Jp);
      lcost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
    }
    const double my_cost = active ? 0.5 * (r[0] * r[0] + r[1] * r[1]) : 0.0;
    double cost_l = 0.0;  // landmark cost, summed in observation order (deterministic mode)
    int maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
    // landmark sums: every lane of a segment gathers the whole segment in observation order
    double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0};
    for (int t = 0; t < maxlen; ++t) {
      const int src = (first + t) & 63;
      double q[6], rr[2];
#pragma unroll
      for (int i = 0; i < 6; ++i) q[i] = shfl_d(Jp[i], src);
      rr[0] = shfl_d(r[0], src); rr[1] = shfl_d(r[1], src);
      const double ct = shfl_d(my_cost, src);
      if (t < len) {
        cost_l += ct;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          gp[a] += q[a] * rr[0] + q[3 + a] * rr[1];
#pragma unroll
          for (int b = 0; b < 3; ++b) V[3 * a + b] += q[a] * q[b] + q[3 + a] * q[3 + b];
        }
      }
    }
    double s[3] = {1, 1, 1};
    if (active) {
      if (first_pass) {
#pragma unroll
        for (int a = 0; a < 3; ++a) s[a] = 1.0 / (1.0 + sqrt(V[4 * a]));
        if (lane == first) { P.sp[3 * j] = s[0]; P.sp[3 * j + 1] = s[1]; P.sp[3 * j + 2] = s[2]; }
      } else {
        s[0] = P.sp[3 * j]; s[1] = P.sp[3 * j + 1]; s[2] = P.sp[3 * j + 2];
      }
      if (lane == first) {
        lgp2 += gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
        if (P.det) { P.lmV[4 * (size_t)j] = cost_l; P.lmV[4 * (size_t)j + 1] = gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2]; }
      }
    }
    double Vd[9], Vi[9], gps[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      gps[a] = gp[a] * s[a];
#pragma unroll
      for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) Vd[4 * a] += fmin(fmax(Vd[4 * a], MIN_DIAG), MAX_DIAG) / radius;
    inv3_sym(Vd, Vi);
    // W s and Y = (W s) Vd^-1 for free poses
    double Ws[18], Y[18];
    const bool freep = active && k > 0;
    const int base = 6 * (k - 1);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) Ws[3 * a + b] = freep ? (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b]) * s[b] : 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) Y[3 * a + b] = Ws[3 * a] * Vi[b] + Ws[3 * a + 1] * Vi[3 + b] + Ws[3 * a + 2] * Vi[6 + b];
    if (freep && P.det) {
      double* ov = P.obsV + (size_t)P.obs_pos[o] * 18;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        ov[a] = Jc[a] * r[0] + Jc[6 + a] * r[1];
        ov[6 + a] = -(Y[3 * a] * gps[0] + Y[3 * a + 1] * gps[1] + Y[3 * a + 2] * gps[2]);
        ov[12 + a] = Jc[a] * Jc[a] + Jc[6 + a] * Jc[6 + a];
      }
    }
    if (freep && !P.det) {
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        atomicAdd(&sGc[base + a], Jc[a] * r[0] + Jc[6 + a] * r[1]);
        atomicAdd(&sDU[base + a], Jc[a] * Jc[a] + Jc[6 + a] * Jc[6 + a]);
        atomicAdd(&sGred[base + a], -(Y[3 * a] * gps[0] + Y[3 * a + 1] * gps[1] + Y[3 * a + 2] * gps[2]));
#pragma unroll
        for (int b = 0; b < 6; ++b) atomicAdd(&sS[(base + a) * n + base + b], Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
      }
    }
    // Schur pairs: lane (pose k) x every later-or-equal member of its segment; mirrored on the host
    for (int t = 0; t < maxlen; ++t) {
      const int src = (first + t) & 63;
      const int kt = __shfl(k, src);
      double Wt[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) Wt[i] = shfl_d(Ws[i], src);
      if (freep && t < len && kt > 0 && src >= lane) {
        if (P.det) {
          const int slot = P.pair_base[o] + (src - lane);
          const int posA = P.pair_pos[2 * slot], posB = P.pair_pos[2 * slot + 1];
          double* B = P.pairB + (size_t)posA * 36;
          double* Bt = posB >= 0 ? P.pairB + (size_t)posB * 36 : nullptr;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
              const double v = -(Y[3 * a] * Wt[3 * b] + Y[3 * a + 1] * Wt[3 * b + 1] + Y[3 * a + 2] * Wt[3 * b + 2]);
              const double w = src == lane ? (Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]) + v : v;
              B[6 * a + b] = w;
              if (Bt) Bt[6 * b + a] = w;  // the mirrored pose pair receives the transpose
            }
        } else {
          const int bt = 6 * (kt - 1);
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
              atomicAdd(&sS[(base + a) * n + bt + b],
                        -(Y[3 * a] * Wt[3 * b] + Y[3 * a + 1] * Wt[3 * b + 1] + Y[3 * a + 2] * Wt[3 * b + 2]));
        }
      }
    }
  }
  if (P.det) return;  // sums are formed by ba_reduce1_kernel in the declared order
  // block totals of cost / gp2
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { lcost += __shfl_xor(lcost, off); lgp2 += __shfl_xor(lgp2, off); }
  if (lane == 0) { atomicAdd(&lds[pay1 - 2], lcost); atomicAdd(&lds[pay1 - 1], lgp2); }
  __syncthreads();
  for (int i = threadIdx.x; i < pay1; i += blockDim.x) {
    const double v = lds[i];
    if (v != 0.0) atomicAdd(&P.pay1[i], v);
  }
}

__global__ __launch_bounds__(256) void ba_backsub_kernel(BaDev P, double radius) {
  __shared__ double sAcc[4];
  if (threadIdx.x < 4) sAcc[threadIdx.x] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a_cost = 0, a_mc = 0, a_dp2 = 0, a_p2 = 0;
  const int wpb = blockDim.x >> 6;
  for (int chunk = blockIdx.x * wpb + wave; chunk < P.C; chunk += gridDim.x * wpb) {
    const int c0 = P.chunk_start[chunk], c1 = P.chunk_start[chunk + 1];
    const int o = c0 + lane;
    const bool active = o < c1;
    int k = 0, j = 0, first = lane, len = 0;
    double r[2] = {0, 0}, Jc[12], Jp[6], jd[2] = {0, 0}, u = 0, v = 0;
    double det_c = 0.0, det_mc = 0.0, det_dp2 = 0.0, det_p2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) Jp[i] = 0.0;
    D3 p{0, 0, 1};
    if (active) {
      k = P.obs_pose[o]; j = P.obs_point[o];
      first = P.lm_start[j] - c0; len = P.lm_start[j + 1] - P.lm_start[j];
      p = D3{P.points[3 * j], P.points[3 * j + 1], P.points[3 * j + 2]};
      u = P.obs_uv[2 * o]; v = P.obs_uv[2 * o + 1];
      eval_obs(P.poses + 7 * k, p, u, v, P.f, P.cx, P.cy, k > 0, r, Jc, Jp);
      if (k > 0) {
        const double* d = P.dc + 6 * (k - 1);
#pragma unroll
        for (int a = 0; a < 6; ++a) { jd[0] += Jc[a] * d[a]; jd[1] += Jc[6 + a] * d[a]; }
      }
    }
    int maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
    double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, wd[3] = {0, 0, 0};
    for (int t = 0; t < maxlen; ++t) {
      const int src = (first + t) & 63;
      double q[6], rr[2], dd[2];
#pragma unroll
      for (int i = 0; i < 6; ++i) q[i] = shfl_d(Jp[i], src);
      rr[0] = shfl_d(r[0], src); rr[1] = shfl_d(r[1], src);
      dd[0] = shfl_d(jd[0], src); dd[1] = shfl_d(jd[1], src);
      if (t < len) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          gp[a] += q[a] * rr[0] + q[3 + a] * rr[1];
          wd[a] += q[a] * dd[0] + q[3 + a] * dd[1];
#pragma unroll
          for (int b = 0; b < 3; ++b) V[3 * a + b] += q[a] * q[b] + q[3 + a] * q[3 + b];
        }
      }
    }
    if (active) {
      const double s[3] = {P.sp[3 * j], P.sp[3 * j + 1], P.sp[3 * j + 2]};
      double Vd[9], Vi[9], De[3], rh[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        rh[a] = -(gp[a] + wd[a]) * s[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) { De[a] = fmin(fmax(Vd[4 * a], MIN_DIAG), MAX_DIAG) / radius; Vd[4 * a] += De[a]; }
      inv3_sym(Vd, Vi);
      double np[3];
      const double pv[3] = {p.x, p.y, p.z};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double y = Vi[3 * a] * rh[0] + Vi[3 * a + 1] * rh[1] + Vi[3 * a + 2] * rh[2];
        const double d = y * s[a];
        np[a] = pv[a] + d;
        if (lane == first) {
          a_mc += 0.5 * y * (De[a] * y - gp[a] * s[a]);
          a_dp2 += d * d;
          a_p2 += pv[a] * pv[a];
        }
      }
      if (lane == first) { P.cand_points[3 * j] = np[0]; P.cand_points[3 * j + 1] = np[1]; P.cand_points[3 * j + 2] = np[2]; }
      double r0, r1;
      reproj_residual(P.cand_poses + 7 * k, D3{np[0], np[1], np[2]}, u, v, P.f, P.cx, P.cy, r0, r1);
      a_cost += 0.5 * (r0 * r0 + r1 * r1);
      det_c = 0.5 * (r0 * r0 + r1 * r1);
      det_mc = 0.0; det_dp2 = 0.0; det_p2 = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double y = Vi[3 * a] * rh[0] + Vi[3 * a + 1] * rh[1] + Vi[3 * a + 2] * rh[2];
        const double d = y * s[a];
        det_mc += 0.5 * y * (De[a] * y - gp[a] * s[a]);
        det_dp2 += d * d;
        det_p2 += pv[a] * pv[a];
      }
    }
    if (P.det) {  // candidate cost of the landmark in observation order, then the landmark's slot
      double cn = 0.0;
      for (int t = 0; t < maxlen; ++t) {
        const double ct = shfl_d(det_c, (first + t) & 63);
        if (t < len) cn += ct;
      }
      if (active && lane == first) {
        double* lv = P.lmV + 4 * (size_t)j;
        lv[0] = cn; lv[1] = det_mc; lv[2] = det_dp2; lv[3] = det_p2;
      }
    }
  }
  if (P.det) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a_cost += __shfl_xor(a_cost, off); a_mc += __shfl_xor(a_mc, off);
    a_dp2 += __shfl_xor(a_dp2, off); a_p2 += __shfl_xor(a_p2, off);
  }
  if (lane == 0) { atomicAdd(&sAcc[0], a_cost); atomicAdd(&sAcc[1], a_mc); atomicAdd(&sAcc[2], a_dp2); atomicAdd(&sAcc[3], a_p2); }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(&P.pay2[threadIdx.x], sAcc[threadIdx.x]);
}

// R(list): 28 consecutive segments summed sequentially, then the segment sums added sequentially
// (the declared order; see oracle/ora_ba.cpp).  One workgroup per destination: F*F pose-pair blocks
// (36 values), F pose vectors (18 values), 1 scalar pair; lane = (segment, element).
constexpr int RSEG = 28;
__global__ __launch_bounds__(1024) void ba_reduce1_kernel(BaDev P) {
  __shared__ double sP[RSEG][36];
  const int F = P.K - 1, n = P.n, tid = threadIdx.x, d = blockIdx.x;
  const int width = d < F * F ? 36 : (d < F * F + F ? 18 : 2);
  const int stride = d < F * F ? 36 : (d < F * F + F ? 18 : 4);
  const double* base = d < F * F ? P.pairB : (d < F * F + F ? P.obsV : P.lmV);
  const int seg = tid / width, e = tid % width;
  const int nd = F * F + F + 1;
  const int e0 = P.list_start[d], len = P.list_start[nd + 1 + d] - e0;
  const int seglen = (len + RSEG - 1) / RSEG;
  if (seg < RSEG) {
    double acc = 0.0;
    const int b0 = seg * seglen, b1 = min(len, (seg + 1) * seglen);
    const double* src = base + (size_t)e0 * stride + e;
    // 8 independent loads in flight, adds strictly in list order
    for (int q0 = b0; q0 < b1; q0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = q0 + u < b1 ? src[(size_t)(q0 + u) * stride] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (q0 + u < b1) acc += v[u];
    }
    sP[seg][e] = acc;
  }
  __syncthreads();
  if (tid < width) {
    double acc = 0.0;
    for (int sg = 0; sg < RSEG; ++sg) acc += sP[sg][tid];
    double* out = P.pay1_out;
    if (d < F * F) {
      const int ka = d / F, kb = d % F;
      out[(size_t)(6 * ka + tid / 6) * n + 6 * kb + tid % 6] = acc;
    } else if (d < F * F + F) {
      const int k = d - F * F;
      if (tid < 6) out[(size_t)n * n + n + 6 * k + tid] = acc;                  // g_c
      else if (tid < 12) out[(size_t)n * n + 6 * k + (tid - 6)] = acc;          // g_red (the -Y g_p part)
      else out[(size_t)n * n + 2 * n + 6 * k + (tid - 12)] = acc;               // diag U
    } else {
      out[(size_t)n * n + 3 * n + tid] = acc;
    }
  }
}

__global__ __launch_bounds__(128) void ba_reduce2_kernel(BaDev P) {
  __shared__ double sP[RSEG][4];
  const int F = P.K - 1, tid = threadIdx.x;
  const int d = F * F + F;  // the landmark list
  const int seg = tid / 4, e = tid % 4;
  const int e0 = P.list_start[d], len = P.list_start[F * F + F + 1 + 1 + d] - e0;
  const int seglen = (len + RSEG - 1) / RSEG;
  if (seg < RSEG) {
    double acc = 0.0;
    const int b0 = seg * seglen, b1 = min(len, (seg + 1) * seglen);
    const double* src = P.lmV + 4 * (size_t)e0 + e;
    for (int q0 = b0; q0 < b1; q0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = q0 + u < b1 ? src[4 * (size_t)(q0 + u)] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (q0 + u < b1) acc += v[u];
    }
    sP[seg][e] = acc;
  }
  __syncthreads();
  if (tid < 4) {
    double acc = 0.0;
    for (int sg = 0; sg < RSEG; ++sg) acc += sP[sg][tid];
    P.pay2_out[tid] = acc;
  }
}

// ----------------------------------------------------------------------------- host side
namespace {
bool cholesky_solve(std::vector<double>& A, std::vector<double>& b, int n) {
  for (int j = 0; j < n; ++j) {
    double s = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(s > 0)) return false;
    const double l = sqrt(s);
    A[(size_t)j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double v = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / l;
    }
  }
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * b[k];
    b[i] = v / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * b[k];
    b[i] = v / A[(size_t)i * n + i];
  }
  return true;
}

void plus_pose(const double* p, const double* d, double* out) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double qd[4];
  if (nd > 0) {
    const double s = sin(nd) / nd;
    qd[0] = cos(nd); qd[1] = s * d[0]; qd[2] = s * d[1]; qd[3] = s * d[2];
  } else { qd[0] = 1; qd[1] = qd[2] = qd[3] = 0; }
  const double* q = p;
  out[0] = qd[0] * q[0] - qd[1] * q[1] - qd[2] * q[2] - qd[3] * q[3];
  out[1] = qd[0] * q[1] + qd[1] * q[0] + qd[2] * q[3] - qd[3] * q[2];
  out[2] = qd[0] * q[2] - qd[1] * q[3] + qd[2] * q[0] + qd[3] * q[1];
  out[3] = qd[0] * q[3] + qd[1] * q[2] - qd[2] * q[1] + qd[3] * q[0];
  out[4] = p[4] + d[3]; out[5] = p[5] + d[4]; out[6] = p[6] + d[5];
}
}  // namespace

struct svo_ba {
  svo_ctx* ctx = nullptr;
  svo_camera_info cam{};
  svo_ba_options opt{};
  int window_size = 5, max_landmarks = 0, max_obs = 0, max_poses = 0;
  svo_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  // device problem
  BaDev d;
  size_t cap_points = 0, cap_obs = 0, cap_chunks = 0, cap_pay1 = 0, cap_pairs = 0, cap_list = 0;
  // host mirrors of the loaded problem
  std::vector<double> h_poses, h_cand_poses;
  int n_points = 0;
  hipStream_t stream = nullptr;  // BA has its own stream so a solve can overlap the tracker's kernels
  double* step_buf[2] = {nullptr, nullptr};
  double* h_pin = nullptr;  // pinned staging: payload1 / payload2 / dc / poses
  size_t pin_bytes = 0;
  // sliding-window graph (BundleAdjuster state, host side; ids sequential — SURVEY C-3)
  struct Obs { float u, v; int64_t id; };
  struct PoseVar { double pose[7]; std::vector<Obs> obs; };
  std::deque<PoseVar> window;
  std::vector<double> feat_pos;  // 3 per feature id
  bool new_frame_added = false;
  std::vector<int64_t> solve_lm_ids;
  std::vector<int32_t> h_list_begin, h_list_end;
  size_t n_pair_rows = 0;
};

static int ba_alloc(svo_ba* ba) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  const int Kmax = ba->max_poses, nmax = 6 * (Kmax - 1);
  ba->cap_points = ba->max_landmarks; ba->cap_obs = ba->max_obs; ba->cap_chunks = ba->max_obs + 1;
  ba->cap_pay1 = (size_t)nmax * nmax + 3 * (size_t)nmax + 2;
#define A(ptr, T, cnt) SVO_HIP_CHECK(ctx, hipMalloc((void**)&(ptr), sizeof(T) * (size_t)(cnt)))
  // two [dc | poses] step buffers: the candidate of an accepted step becomes the linearisation point by a pointer swap
  A(ba->step_buf[0], double, (nmax > 0 ? nmax : 1) + 7 * Kmax); A(ba->step_buf[1], double, (nmax > 0 ? nmax : 1) + 7 * Kmax);
  A(d.points, double, 3 * ba->cap_points); A(d.cand_points, double, 3 * ba->cap_points);
  A(d.sp, double, 3 * ba->cap_points);
  A(d.obs_pose, int32_t, ba->cap_obs); A(d.obs_point, int32_t, ba->cap_obs); A(d.obs_uv, double, 2 * ba->cap_obs);
  A(d.lm_start, int32_t, ba->cap_points + 1); A(d.chunk_start, int32_t, ba->cap_chunks + 1);
  A(d.pay1, double, ba->cap_pay1); A(d.pay2, double, 4);
  A(d.pair_base, int32_t, ba->cap_obs + 1); A(d.obs_pos, int32_t, ba->cap_obs + 1); A(d.obsV, double, 18 * ba->cap_obs);
  A(d.lmV, double, 4 * ba->cap_points); A(d.list_start, int32_t, 2 * (64 * 64 + 64 + 1) + 8);
#undef A
  SVO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ba->stream, hipStreamNonBlocking));
  ba->pin_bytes = sizeof(double) * (ba->cap_pay1 + 64 + 16 * (size_t)Kmax);
  SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_pin, ba->pin_bytes, hipHostMallocDefault));
  return SVO_OK;
}

extern "C" void svo_ba_default_options(svo_ba_options* o) {
  if (!o) return;
  o->max_iterations = 50;
  o->max_time_s = 0.1;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4;
  o->max_features = 400;
}

extern "C" int svo_ba_create(svo_ctx* ctx, svo_ba** out, int window_size, const svo_camera_info* cam,
                             const svo_ba_options* opt, int max_landmarks, int max_observations) {
  if (!ctx || !out || !cam) return SVO_ERR_INVALID;
  SVO_REQUIRE(ctx, window_size >= 1 && window_size <= 64, "ba_create: window size must be 1..64");
  SVO_REQUIRE(ctx, max_landmarks >= 1 && max_observations >= 1, "ba_create: capacities must be positive");
  svo_ba* ba = new svo_ba();
  ba->ctx = ctx;
  ba->cam = *cam;
  if (opt) ba->opt = *opt; else svo_ba_default_options(&ba->opt);
  ba->window_size = window_size;
  ba->max_poses = window_size + 1 > 2 ? window_size + 1 : 2;
  if (ba->max_poses > 64) ba->max_poses = 64;
  ba->max_landmarks = max_landmarks;
  ba->max_obs = max_observations;
  int rc = ba_alloc(ba);
  if (rc) { svo_ba_destroy(ba); return rc; }
  *out = ba;
  return SVO_OK;
}

extern "C" void svo_ba_destroy(svo_ba* ba) {
  if (!ba) return;
  BaDev& d = ba->d;
  void* ptrs[] = {ba->step_buf[0], ba->step_buf[1], d.points, d.cand_points, d.sp, d.obs_pose, d.obs_point, d.obs_uv,
                  d.lm_start, d.chunk_start, d.pay1, d.pay2, d.pair_base, d.pair_pos, d.obs_pos, d.pairB, d.obsV, d.lmV, d.list_start};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ba->h_pin) (void)hipHostFree(ba->h_pin);
  if (ba->stream) { (void)hipStreamSynchronize(ba->stream); (void)hipStreamDestroy(ba->stream); }
  delete ba;
}

extern "C" int svo_ba_set_allreduce(svo_ba* ba, svo_allreduce_fn fn, void* user) {
  if (!ba) return SVO_ERR_INVALID;
  ba->allreduce = fn;
  ba->allreduce_user = user;
  return SVO_OK;
}

// Upload a landmark-major problem (shared by the bulk API and the sliding-window solve).
static int ba_upload(svo_ba* ba, int K, const double* poses7, int npts, const double* points3, int M,
                     const int32_t* op, const int32_t* oj, const double* uv) {
  svo_ctx* ctx = ba->ctx;
  SVO_REQUIRE(ctx, K >= 1 && K <= ba->max_poses, "ba: pose count outside the window capacity");
  SVO_REQUIRE(ctx, npts >= 0 && (size_t)npts <= ba->cap_points && M >= 0 && (size_t)M <= ba->cap_obs, "ba: problem exceeds capacity");
  BaDev& d = ba->d;
  d.K = K; d.n = 6 * (K - 1); d.M = M; d.f = ba->cam.focal; d.cx = ba->cam.cx; d.cy = ba->cam.cy;
  ba->n_points = npts;
  // CSR over landmark index + wave chunks (<= 64 observations, whole landmarks)
  std::vector<int32_t> lm_start((size_t)npts + 1, 0), chunks;
  for (int o = 0; o < M; ++o) {
    SVO_REQUIRE(ctx, oj[o] >= 0 && oj[o] < npts && op[o] >= 0 && op[o] < K, "ba: observation index out of range");
    SVO_REQUIRE(ctx, o == 0 || oj[o] >= oj[o - 1], "ba: observations must be sorted by landmark");
    lm_start[oj[o] + 1]++;
  }
  for (int j = 0; j < npts; ++j) {
    SVO_REQUIRE(ctx, lm_start[j + 1] <= 64, "ba: a landmark has more than 64 observations");
    lm_start[j + 1] += lm_start[j];
  }
  chunks.push_back(0);
  int cur = 0;
  for (int j = 0; j < npts; ++j) {
    const int len = lm_start[j + 1] - lm_start[j];
    if (len == 0) continue;
    if (cur + len > 64) { chunks.push_back(lm_start[j]); cur = 0; }
    cur += len;
  }
  chunks.push_back(M);
  d.C = (int)chunks.size() - 1;
  d.L = npts;
  hipStream_t st = ba->stream;
  // deterministic mode: pair slots + destination lists (landmark order) if they fit
  {
    const int F = K - 1;
    std::vector<int32_t> pair_base((size_t)M + 1, 0);
    for (int j = 0; j < npts; ++j)
      for (int o = lm_start[j]; o < lm_start[j + 1]; ++o) pair_base[o + 1] = pair_base[o] + (lm_start[j + 1] - o);
    const size_t n_pairs = (size_t)pair_base[M];
    d.det = n_pairs <= ((size_t)1 << 21) ? 1 : 0;  // <= 604 MB of pair blocks
    if (d.det) {
      // destination lists in landmark order; the landmark list is the identity over [0, npts)
      const int nd = F * F + F + 1;
      std::vector<int32_t> cnt(nd, 0), pair_pos(2 * n_pairs + 2, -1), obs_pos((size_t)M + 1, -1);
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
          // block lists are back to back in pairB, pose lists back to back in obsV, the landmark list is
          // rows [0, npts) of lmV
          ba->h_list_begin.assign(nd, 0); ba->h_list_end.assign(nd, 0);
          int32_t acc = 0;
          for (int q = 0; q < F * F; ++q) { ba->h_list_begin[q] = acc; acc += cnt[q]; ba->h_list_end[q] = acc; }
          ba->n_pair_rows = (size_t)acc;
          acc = 0;
          for (int q = F * F; q < F * F + F; ++q) { ba->h_list_begin[q] = acc; acc += cnt[q]; ba->h_list_end[q] = acc; }
          ba->h_list_begin[F * F + F] = 0; ba->h_list_end[F * F + F] = npts;
        }
        std::vector<int32_t> fill(nd, 0);
        for (int j = 0; j < npts; ++j)
          for (int i = lm_start[j]; i < lm_start[j + 1]; ++i) {
            const int ki = op[i] - 1;
            if (ki < 0) continue;
            if (pass == 0) cnt[F * F + ki]++;
            else obs_pos[i] = ba->h_list_begin[F * F + ki] + fill[F * F + ki]++;
            for (int t = i; t < lm_start[j + 1]; ++t) {
              const int kt = op[t] - 1;
              if (kt < 0) continue;
              const int slot = pair_base[i] + (t - i);
              const int da = ki * F + kt, db = kt * F + ki;
              if (pass == 0) { cnt[da]++; if (t != i) cnt[db]++; }
              else {
                pair_pos[2 * slot] = ba->h_list_begin[da] + fill[da]++;
                if (t != i) pair_pos[2 * slot + 1] = ba->h_list_begin[db] + fill[db]++;
              }
            }
          }
      }
      const size_t rows = ba->n_pair_rows;
      if (rows > ba->cap_pairs || 2 * n_pairs + 2 > ba->cap_list) {
        if (d.pairB) (void)hipFree(d.pairB);
        if (d.pair_pos) (void)hipFree(d.pair_pos);
        d.pairB = nullptr; d.pair_pos = nullptr;
        ba->cap_pairs = rows + rows / 4 + 1024;
        ba->cap_list = 2 * n_pairs + n_pairs / 2 + 4096;
        SVO_HIP_CHECK(ctx, hipMalloc((void**)&d.pairB, sizeof(double) * 36 * ba->cap_pairs));
        SVO_HIP_CHECK(ctx, hipMalloc((void**)&d.pair_pos, sizeof(int32_t) * ba->cap_list));
      }
      SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.pair_base, pair_base.data(), sizeof(int32_t) * (M + 1), hipMemcpyHostToDevice, st));
      SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.pair_pos, pair_pos.data(), sizeof(int32_t) * (2 * n_pairs + 2), hipMemcpyHostToDevice, st));
      SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.obs_pos, obs_pos.data(), sizeof(int32_t) * (M + 1), hipMemcpyHostToDevice, st));
      // list_start[q] = begin(q), list_start[nd + 1 + q] = end(q)
      std::vector<int32_t> ls(2 * (size_t)nd + 2, 0);
      for (int q = 0; q < nd; ++q) { ls[q] = ba->h_list_begin[q]; ls[nd + 1 + q] = ba->h_list_end[q]; }
      SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.list_start, ls.data(), sizeof(int32_t) * ls.size(), hipMemcpyHostToDevice, st));
      if (npts) SVO_HIP_CHECK(ctx, hipMemsetAsync(d.lmV, 0, sizeof(double) * 4 * npts, st));
      SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
  }
  ba->h_poses.assign(poses7, poses7 + 7 * (size_t)K);
  ba->h_cand_poses = ba->h_poses;
  d.poses = ba->step_buf[0] + (d.n > 0 ? d.n : 1);
  d.cand_poses = ba->step_buf[1] + (d.n > 0 ? d.n : 1);
  d.dc = ba->step_buf[1];
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.poses, poses7, sizeof(double) * 7 * K, hipMemcpyHostToDevice, st));
  if (npts) SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.points, points3, sizeof(double) * 3 * npts, hipMemcpyHostToDevice, st));
  if (npts) SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.cand_points, points3, sizeof(double) * 3 * npts, hipMemcpyHostToDevice, st));
  if (M) {
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.obs_pose, op, sizeof(int32_t) * M, hipMemcpyHostToDevice, st));
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.obs_point, oj, sizeof(int32_t) * M, hipMemcpyHostToDevice, st));
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.obs_uv, uv, sizeof(double) * 2 * M, hipMemcpyHostToDevice, st));
  }
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.lm_start, lm_start.data(), sizeof(int32_t) * (npts + 1), hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.chunk_start, chunks.data(), sizeof(int32_t) * chunks.size(), hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));  // host vectors go out of scope
  return SVO_OK;
}

// The LM loop (mirrors oracle/ora_ba.cpp step for step).
static int ba_lm(svo_ba* ba, svo_ba_summary* sum) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  hipStream_t st = ba->stream;
  const int n = d.n, K = d.K;
  const size_t pay1 = (size_t)n * n + 3 * (size_t)n + 2;
  const auto t_begin = std::chrono::steady_clock::now();
  double* h_pay1 = ba->h_pin;
  double* h_pay2 = h_pay1 + pay1;
  double* h_dc = h_pay2 + 8;
  double* h_cp = h_dc + (n > 0 ? n : 1);  // contiguous with h_dc: one H2D per iteration
  const int grid = std::max(1, std::min(svo_div_up(d.C, 4), 512));
  const size_t lds_bytes = pay1 * sizeof(double);
  if (!d.det && lds_bytes > 64 * 1024) {
    SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ba_linearize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  std::vector<double> sc(n, 0.0), Sm((size_t)n * n), rhs(n), Df(n), Sfull((size_t)n * n);
  bool have_scale = false;
  double radius = ba->opt.initial_radius, decrease_factor = 2.0;
  double* cur_points = d.points;
  double* cand_points = d.cand_points;
  double* cur_poses = d.poses;
  double* cand_poses = d.cand_poses;

  auto linearize = [&](double rad) -> int {
    d.points = cur_points; d.cand_points = cand_points; d.poses = cur_poses; d.cand_poses = cand_poses;
    if (!d.det) SVO_HIP_CHECK(ctx, hipMemsetAsync(d.pay1, 0, sizeof(double) * pay1, st));
    if (d.C > 0) {
      SvoProfScope prof(ctx, SVO_PROF_BA_LINEARIZE, st);
      if (d.det) hipLaunchKernelGGL(ba_linearize_kernel, dim3(d.C), dim3(64), 64, st, d, rad, have_scale ? 0 : 1);  // one wave per workgroup: spreads the chunks over the CUs
      else hipLaunchKernelGGL(ba_linearize_kernel, dim3(grid), dim3(256), lds_bytes, st, d, rad, have_scale ? 0 : 1);
    }
    // single rank + deterministic mode: the reduce kernel writes the payload straight into pinned host
    // memory (no copy kernel); a sharded run keeps it on the device for the all-reduce
    const bool zero_copy = d.det && !ba->allreduce;
    d.pay1_out = zero_copy ? h_pay1 : d.pay1;
    d.pay2_out = zero_copy ? h_pay2 : d.pay2;
    if (d.det) hipLaunchKernelGGL(ba_reduce1_kernel, dim3((K - 1) * (K - 1) + (K - 1) + 1), dim3(1024), 0, st, d);
    SVO_HIP_CHECK(ctx, hipGetLastError());
    if (ba->allreduce) {
      SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
      if (ba->allreduce(d.pay1, pay1, ba->allreduce_user)) { ctx->err = "ba: allreduce callback failed"; return SVO_ERR_INVALID; }
    }
    if (!zero_copy) SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_pay1, d.pay1, sizeof(double) * pay1, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
    // mirror the upper pair blocks (kernel writes each unordered pose pair once)
    const double* S = h_pay1;
    for (int a = 0; a < K - 1; ++a)
      for (int b = 0; b < K - 1; ++b)
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) {
            const size_t ij = (size_t)(6 * a + i) * n + 6 * b + j, ji = (size_t)(6 * b + j) * n + 6 * a + i;
            Sfull[ij] = (a == b || d.det) ? S[ij] : S[ij] + S[ji];
          }
    return SVO_OK;
  };
  auto gradient_norm = [&]() {
    const double* gc = h_pay1 + (size_t)n * n + n;
    double g2 = h_pay1[pay1 - 1];
    for (int a = 0; a < n; ++a) g2 += gc[a] * gc[a];
    return sqrt(g2);
  };

  int iterations = 0, successful = 0, termination = 1;
  int rc = linearize(radius);
  if (rc) return rc;
  double cost = h_pay1[pay1 - 2];
  const double initial_cost = cost;
  {
    const double* dU = h_pay1 + (size_t)n * n + 2 * (size_t)n;
    for (int a = 0; a < n; ++a) sc[a] = 1.0 / (1.0 + sqrt(dU[a]));
    have_scale = true;
  }
  bool need_linearize = false;
  if (gradient_norm() <= ba->opt.gradient_tolerance) termination = 0;
  else
    while (true) {
      if (iterations >= ba->opt.max_iterations) { termination = 1; break; }
      if (ba->opt.max_time_s > 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() >= ba->opt.max_time_s) {
        termination = 1; break;  // src/bundle_adjuster.cpp:11 (wall clock; disabled for parity runs)
      }
      if (radius <= MIN_RADIUS) { termination = 0; break; }
      ++iterations;
      if (need_linearize) { rc = linearize(radius); if (rc) return rc; need_linearize = false; }
      const double* gred = h_pay1 + (size_t)n * n;
      const double* gc = gred + n;
      const double* dU = gc + n;
      for (int a = 0; a < n; ++a) {
        Df[a] = std::min(std::max(dU[a] * sc[a] * sc[a], MIN_DIAG), MAX_DIAG) / radius;
        for (int b = 0; b < n; ++b) Sm[(size_t)a * n + b] = Sfull[(size_t)a * n + b] * sc[a] * sc[b];
        Sm[(size_t)a * n + a] += Df[a];
        rhs[a] = -(gred[a] + gc[a]) * sc[a];  // kernel accumulates only the -Y g_p part of the reduced gradient
      }
      const bool ok = n == 0 || cholesky_solve(Sm, rhs, n);
      bool step_ok = false;
      double cost_new = 0, model_change = 0, step2 = 0, x2 = 0;
      if (ok) {
        double mcc = 0;
        for (int a = 0; a < n; ++a) {
          mcc += 0.5 * rhs[a] * (Df[a] * rhs[a] - gc[a] * sc[a]);
          h_dc[a] = rhs[a] * sc[a];
        }
        for (int k = 0; k < K; ++k) {
          if (k == 0) memcpy(&ba->h_cand_poses[0], &ba->h_poses[0], 7 * sizeof(double));
          else plus_pose(&ba->h_poses[7 * k], &h_dc[6 * (k - 1)], &ba->h_cand_poses[7 * k]);
        }
        memcpy(h_cp, ba->h_cand_poses.data(), sizeof(double) * 7 * K);
        // one H2D: [dc (n) | candidate poses (7K)] are adjacent both in the pinned buffer and on the device
        d.poses = cur_poses; d.cand_poses = cand_poses; d.dc = cand_poses - (n > 0 ? n : 1);
        SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.dc, h_dc, sizeof(double) * ((n > 0 ? n : 1) + 7 * K), hipMemcpyHostToDevice, st));
        if (!d.det) SVO_HIP_CHECK(ctx, hipMemsetAsync(d.pay2, 0, sizeof(double) * 4, st));
        d.points = cur_points; d.cand_points = cand_points;
        if (d.C > 0) {
          SvoProfScope prof(ctx, SVO_PROF_BA_BACKSUB, st);
          if (d.det) hipLaunchKernelGGL(ba_backsub_kernel, dim3(d.C), dim3(64), 0, st, d, radius);
          else hipLaunchKernelGGL(ba_backsub_kernel, dim3(grid), dim3(256), 0, st, d, radius);
        }
        if (d.det) hipLaunchKernelGGL(ba_reduce2_kernel, dim3(1), dim3(128), 0, st, d);
        SVO_HIP_CHECK(ctx, hipGetLastError());
        if (ba->allreduce) {
          SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
          if (ba->allreduce(d.pay2, 4, ba->allreduce_user)) { ctx->err = "ba: allreduce callback failed"; return SVO_ERR_INVALID; }
        }
        if (!(d.det && !ba->allreduce)) SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_pay2, d.pay2, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
        SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
        cost_new = h_pay2[0];
        model_change = mcc + h_pay2[1];
        step2 = h_pay2[2]; x2 = h_pay2[3];
        for (int k = 1; k < K; ++k)
          for (int a = 0; a < 7; ++a) {
            const double dd = ba->h_cand_poses[7 * k + a] - ba->h_poses[7 * k + a];
            step2 += dd * dd;
            x2 += ba->h_poses[7 * k + a] * ba->h_poses[7 * k + a];
          }
        step_ok = model_change > 0;
      }
      if (!step_ok) { radius /= decrease_factor; decrease_factor *= 2; need_linearize = true; continue; }
      auto accept = [&]() {
        ba->h_poses = ba->h_cand_poses;
        std::swap(cur_points, cand_points);
        std::swap(cur_poses, cand_poses);  // the candidate poses are already on the device
        cost = cost_new;
      };
      if (sqrt(step2) <= ba->opt.parameter_tolerance * (sqrt(x2) + ba->opt.parameter_tolerance)) { termination = 0; break; }
      const double cost_change = cost - cost_new;
      if (fabs(cost_change) <= ba->opt.function_tolerance * cost) {
        if (cost_change > 0) accept();
        termination = 0;
        break;
      }
      const double rho = cost_change / model_change;
      if (getenv("SVO_BA_TRACE"))
        fprintf(stderr, "[hip] it %d cost %.17g new %.17g model %.17g rho %.6g radius %.6g\n", iterations, cost, cost_new, model_change, rho, radius);
      if (rho > MIN_REL_DECREASE) {
        accept();
        ++successful;
        const double t = 2.0 * rho - 1.0;
        radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
        radius = std::min(MAX_RADIUS, radius);
        decrease_factor = 2.0;
        rc = linearize(radius);
        if (rc) return rc;
        if (gradient_norm() <= ba->opt.gradient_tolerance) { termination = 0; break; }
      } else {
        radius /= decrease_factor; decrease_factor *= 2; need_linearize = true;
      }
    }
  // leave the result in d.points / d.poses
  d.points = cur_points; d.cand_points = cand_points; d.poses = cur_poses; d.cand_poses = cand_poses;
  if (sum) {
    sum->iterations = iterations; sum->successful_steps = successful; sum->termination = termination;
    sum->initial_cost = initial_cost; sum->final_cost = cost;
    sum->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return SVO_OK;
}

extern "C" int svo_ba_load_problem(svo_ba* ba, int n_poses, const double* poses7, int n_points, const double* points3,
                                   int n_obs, const int32_t* obs_pose, const int32_t* obs_point, const double* obs_uv) {
  if (!ba) return SVO_ERR_INVALID;
  SVO_REQUIRE(ba->ctx, poses7 && (n_points == 0 || points3) && (n_obs == 0 || (obs_pose && obs_point && obs_uv)),
              "ba_load_problem: null buffer");
  return ba_upload(ba, n_poses, poses7, n_points, points3, n_obs, obs_pose, obs_point, obs_uv);
}

extern "C" int svo_ba_solve_problem(svo_ba* ba, svo_ba_summary* summary) {
  if (!ba) return SVO_ERR_INVALID;
  SVO_REQUIRE(ba->ctx, ba->d.K >= 1, "ba_solve_problem: no problem loaded");
  return ba_lm(ba, summary);
}

extern "C" int svo_ba_read_problem(svo_ba* ba, double* poses7, double* points3) {
  if (!ba) return SVO_ERR_INVALID;
  svo_ctx* ctx = ba->ctx;
  if (poses7) memcpy(poses7, ba->h_poses.data(), sizeof(double) * 7 * (size_t)ba->d.K);
  if (points3 && ba->n_points) {
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(points3, ba->d.points, sizeof(double) * 3 * (size_t)ba->n_points, hipMemcpyDeviceToHost, ba->stream));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
  }
  return SVO_OK;
}

// ---- sliding-window graph (BundleAdjuster::add_keyframe / bundle_adjust / get_world_points)
extern "C" int svo_ba_add_keyframe(svo_ba* ba, const double* pose7, const int64_t* tracked_ids, const float* tracked_xy,
                                   int n_tracked, const float* new_xy, const float* new_xyz, int n_new, int64_t* new_ids,
                                   int* n_new_out) {
  if (!ba) return SVO_ERR_INVALID;
  svo_ctx* ctx = ba->ctx;
  SVO_REQUIRE(ctx, pose7 && n_tracked >= 0 && n_new >= 0 && n_new_out, "ba_add_keyframe: bad arguments");
  SVO_REQUIRE(ctx, (n_tracked == 0 || (tracked_ids && tracked_xy)) && (n_new == 0 || (new_xy && new_xyz && new_ids)),
              "ba_add_keyframe: null buffer");
  svo_ba::PoseVar pv;
  memcpy(pv.pose, pose7, sizeof(pv.pose));  // src/bundle_adjuster.cpp:63-70
  const int64_t nfeat = (int64_t)(ba->feat_pos.size() / 3);
  for (int i = 0; i < n_tracked; ++i) {   // :72-83
    SVO_REQUIRE(ctx, tracked_ids[i] >= 0 && tracked_ids[i] < nfeat, "ba_add_keyframe: unknown feature id");
    pv.obs.push_back({tracked_xy[2 * i], tracked_xy[2 * i + 1], tracked_ids[i]});
  }
  const int maxf = ba->opt.max_features;
  const int max_new = n_tracked > maxf ? 0 : maxf - n_tracked;  // :85-90 with the C-5 guard
  const int keep = n_new > max_new ? max_new : n_new;
  for (int i = 0; i < keep; ++i) {        // :92-122; ids sequential (C-3), new_ids = real ids only (C-4)
    const int64_t id = (int64_t)(ba->feat_pos.size() / 3);
    ba->feat_pos.push_back(new_xyz[3 * i]); ba->feat_pos.push_back(new_xyz[3 * i + 1]); ba->feat_pos.push_back(new_xyz[3 * i + 2]);
    new_ids[i] = id;
    pv.obs.push_back({new_xy[2 * i], new_xy[2 * i + 1], id});
  }
  *n_new_out = keep;
  ba->window.push_back(std::move(pv));
  if ((int)ba->window.size() > ba->window_size) ba->window.pop_front();  // :126-128 (remove_oldest_pose)
  ba->new_frame_added = true;                                            // :134
  return SVO_OK;
}

extern "C" int svo_ba_window_count(svo_ba* ba) { return ba ? (int)ba->window.size() : 0; }

extern "C" int svo_ba_get_pose(svo_ba* ba, int k, double* pose7) {
  if (!ba || !pose7) return SVO_ERR_INVALID;
  const int K = (int)ba->window.size();
  if (k < 0) k += K;
  SVO_REQUIRE(ba->ctx, k >= 0 && k < K, "ba_get_pose: slot out of range");
  memcpy(pose7, ba->window[k].pose, 7 * sizeof(double));
  return SVO_OK;
}

extern "C" int svo_ba_get_points(svo_ba* ba, const int64_t* ids, int n, float* xyz) {
  if (!ba) return SVO_ERR_INVALID;
  SVO_REQUIRE(ba->ctx, n >= 0 && (n == 0 || (ids && xyz)), "ba_get_points: null buffer");
  const int64_t nfeat = (int64_t)(ba->feat_pos.size() / 3);
  for (int i = 0; i < n; ++i) {  // src/bundle_adjuster.cpp:159-163 (double -> float)
    SVO_REQUIRE(ba->ctx, ids[i] >= 0 && ids[i] < nfeat, "ba_get_points: unknown feature id");
    for (int a = 0; a < 3; ++a) xyz[3 * i + a] = (float)ba->feat_pos[3 * ids[i] + a];
  }
  return SVO_OK;
}

extern "C" int svo_ba_solve(svo_ba* ba, svo_ba_summary* summary) {
  if (!ba) return SVO_ERR_INVALID;
  if (summary) memset(summary, 0, sizeof(*summary));
  if (!ba->new_frame_added) return SVO_OK;  // src/bundle_adjuster.cpp:138
  const int K = (int)ba->window.size();
  struct Flat { int k; float u, v; int64_t id; };
  std::vector<Flat> flat;
  for (int k = 0; k < K; ++k)
    for (const auto& o : ba->window[k].obs) flat.push_back({k, o.u, o.v, o.id});
  std::vector<int> perm(flat.size());
  for (size_t i = 0; i < perm.size(); ++i) perm[i] = (int)i;
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return flat[a].id < flat[b].id; });
  std::vector<double> poses(7 * (size_t)K), points, uv;
  std::vector<int32_t> op, oj;
  std::vector<int64_t>& lm_ids = ba->solve_lm_ids;
  lm_ids.clear();
  for (int k = 0; k < K; ++k) memcpy(&poses[7 * k], ba->window[k].pose, 7 * sizeof(double));
  for (int idx : perm) {
    const Flat& f = flat[idx];
    if (lm_ids.empty() || lm_ids.back() != f.id) {
      lm_ids.push_back(f.id);
      for (int a = 0; a < 3; ++a) points.push_back(ba->feat_pos[3 * f.id + a]);
    }
    op.push_back(f.k); oj.push_back((int32_t)lm_ids.size() - 1);
    uv.push_back(f.u); uv.push_back(f.v);
  }
  int rc = ba_upload(ba, K, poses.data(), (int)lm_ids.size(), points.data(), (int)op.size(), op.data(), oj.data(), uv.data());
  if (rc) return rc;
  rc = ba_lm(ba, summary);
  if (rc) return rc;
  std::vector<double> out_pts(points.size());
  rc = svo_ba_read_problem(ba, poses.data(), out_pts.data());
  if (rc) return rc;
  for (int k = 0; k < K; ++k) memcpy(ba->window[k].pose, &poses[7 * k], 7 * sizeof(double));
  for (size_t l = 0; l < lm_ids.size(); ++l)
    for (int a = 0; a < 3; ++a) ba->feat_pos[3 * lm_ids[l] + a] = out_pts[3 * l + a];
  ba->new_frame_added = false;  // :155
  return SVO_OK;
}
