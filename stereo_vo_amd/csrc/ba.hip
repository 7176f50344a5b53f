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
// Problems whose pair slots would not fit (config 4), or that ask for it (svo_ba_options.accumulation), sum in
// hardware order with a tolerance-level result: ba_linearize_mfma_kernel applies each landmark's Schur
// contribution as a rank-3 update of S on the f64 matrix cores (<= 22 poses), the LDS-atomic kernel is the fallback.
// Host <-> device hand-over of the host-driven loop (single rank, deterministic mode): the reduce kernels write the
// payloads into pinned host memory and publish a completion word that the host polls; the step [dc | candidate
// poses] is read by ba_backsub_kernel in place from pinned memory.  Per LM iteration: 4 launches, no copy, no stream
// wait (SVO_BA_FUSE=1 folds reduce2 into the back-substitution launch: last workgroup reduces; measured, not default).
// The n x n (n = 6 (K-1) <= 114) Cholesky, step control and termination run on the host from the
// (all-reduced) payloads, so every rank of a sharded run takes identical decisions.
// A rank of a sharded run holds all poses and its own landmarks; `allreduce` sums payload1/2 in place
// on the device (RCCL all-reduce over xGMI) — the only exchange of the path.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <sched.h>
#include <chrono>
#include <deque>
#include <memory>
#include <vector>

#include "kernels.h"
#include "reproj_device.h"

bool svo_host_cholesky_solve(double* A, double* b, int n);  // host/linalg.cpp

namespace {
constexpr int RSEG = 28;  // segments of the declared reduction order R(list)
constexpr double MIN_DIAG = 1e-6, MAX_DIAG = 1e32, MAX_RADIUS = 1e16, MIN_RADIUS = 1e-32, MIN_REL_DECREASE = 1e-3;

// LM state of the device-resident loop (deterministic, single-rank solves): the host never sees an
// iteration, it enqueues chunks of [linearize, reduce1, solve, backsub, reduce2+decide] and polls `done`.
struct LmDev {
  double radius, decrease_factor, cost, initial_cost, mcc;
  double function_tol, gradient_tol, parameter_tol;
  int max_iterations;
  int iterations, successful, termination, done;
  int cur;          // which (points, step) buffer pair holds the linearisation point
  int have_scale;   // Jacobi scales fixed (first linearisation done)
  int step_valid;   // the solve kernel produced a step for this iteration
  int grad_check;   // previous step accepted: test the gradient of the new linearisation
  unsigned bar;     // grid-barrier arrival counter of the persistent kernel (monotone within a launch)
  int abort;        // a bounded spin gave up (never expected; keeps a bug from hanging the GPU)
  double sc[6 * 63], Df[6 * 63];
};

// Cross-workgroup reads inside the persistent kernel go through agent-scope (sc1) loads: never the scalar
// cache, never a stale L1 line (cdna_hip_programming.md Guideline 16, Pitfall 6).
__device__ __forceinline__ int ldv(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ldv(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct BaDev {
  int K = 0, n = 0, M = 0, L = 0, C = 0;
  LmDev* lm = nullptr;           // non-null: buffers/radius come from the device state
  double* pts[2] = {nullptr, nullptr};
  double* step[2] = {nullptr, nullptr};  // [dc (max(n,1)) | poses (7K)] x 2
  int* done_host = nullptr;      // pinned: ints [done, iterations, successful, termination, cur, pad], then doubles [initial_cost, cost]
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
  const double* step_in = nullptr;  // pinned host [dc | candidate poses]: read in place by ba_backsub (no H2D blit per LM iteration)
  // completion flags in pinned host memory (single-rank deterministic mode): the reduce kernels publish `seq` after
  // their payload, the host loop polls the word instead of paying a stream wait per half-iteration
  int* flag1 = nullptr;
  int* flag2 = nullptr;
  unsigned* arrive = nullptr;   // device counter of finished reduce1 workgroups (monotone; target = total so far)
  unsigned arrive_target = 0;
  unsigned* arrive2 = nullptr;  // fused back-substitution + reduce2: counter of finished backsub workgroups
  unsigned arrive2_target = 0;  // 0: separate ba_reduce2_kernel launch
  int seq = 0;
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

// Landmark sums of the bulk (hardware-order) paths: NV per-observation terms are summed over each landmark's
// lane segment [first, last] by a segmented inclusive scan (log2(maxlen) shuffle steps) and the segment total
// is broadcast back — instead of every lane walking its whole segment.  Tree order: not for the declared-order
// (deterministic) mode.
template <int NV>
__device__ __forceinline__ void segment_totals(double (&v)[NV], int lane, int first, int last, int maxlen) {
  for (int off = 1; off < maxlen; off <<= 1) {
    const bool take = lane - off >= first;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double u = __shfl_up(v[i], off);
      v[i] += take ? u : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = shfl_d(v[i], last);
}

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

// sin/cos with a declared operation sequence (see oracle/ora_ba.cpp): bit-identical on host and device.
__host__ __device__ inline void det_sincos(double x, double* sn, double* cs) {
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

__host__ __device__ inline void plus_pose(const double* p, const double* d, double* out) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double qd[4];
  if (nd > 0) {
    double sn, cs;
    det_sincos(nd, &sn, &cs);
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

__device__ __forceinline__ void ba_linearize_body(const BaDev& P, double radius, int first_pass) {
  const double* poses_ = P.poses;
  const double* points_ = P.points;
  if (P.lm) {
    if (ldv(&P.lm->done)) return;
    const int c = ldv(&P.lm->cur);
    poses_ = P.step[c] + (P.n > 0 ? P.n : 1); points_ = P.pts[c];
    radius = ldv(&P.lm->radius); first_pass = !ldv(&P.lm->have_scale);
  }
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
      const D3 p{points_[3 * j], points_[3 * j + 1], points_[3 * j + 2]};
      eval_obs(poses_ + 7 * k, p, P.obs_uv[2 * o], P.obs_uv[2 * o + 1], P.f, P.cx, P.cy, k > 0, r, Jc, Jp);
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

// ---------------------------------------------------------------------------------------------------
// Bulk (non-deterministic) linearisation with the Schur products on the f64 matrix cores.
//
// For a landmark with V^-1 = Lc Lc^T (3x3 Cholesky) and Z_i = (W_i s) Lc (6x3 per observing pose), its whole
// contribution to the reduced camera matrix is  -Z Z^T  with Z the (6 x poses) x 3 stack: one rank-3 update
// of S.  S (n <= 128) is cut into 16x16 tiles; v_mfma_f64_16x16x4_f64 applies the update to one tile with
// K = 3 of its 4 k-slots used.  A workgroup is 8 waves; the accumulators are shared by the workgroup, not
// per wave: wave w keeps tiles (w, w..w+4 mod 8) — the circulant half of the symmetric matrix, 5 (4 for
// w >= 4) accumulator tiles = 40 VGPRs — in registers across ALL landmarks the workgroup sees and flushes
// them once at the end.
//   phase 1 (wave = one chunk of <= 64 observations, lane = observation; as ba_linearize_body): residual,
//            Jacobians, landmark sums by lane gathers, V^-1, Z -> LDS, plus a pose->lane byte table and the
//            tile-row mask of every landmark; U / g_c / g_red go to a small LDS image (ds_add_f64).
//   phase 2 (wave = tile row): for each of the 8 staged chunks, for each landmark whose mask touches the
//            row: gather the A operand (16 rows x 3) once, the B operands per touched tile, MFMA.
// Algorithmic work per landmark with L observations: 36 L^2 multiply-adds of Schur product — here
// 2048 flop per touched tile on the matrix pipe instead of 36 L(L+1)/2 LDS atomics.
constexpr int MF_WAVES = 8;
constexpr int MF_COPIES = 8;    // private copies of the U / g_c / g_red image (landmark index mod 8): same-pose lanes of a wave rarely share one
constexpr int MF_TBL_ROW = 32;   // bytes per landmark in the pose->lane table: free poses <= 21 (n <= 128)
typedef double mf_d4 __attribute__((ext_vector_type(4)));

static inline size_t ba_mfma_lds_bytes(int n, int F) {
  return sizeof(double) * ((size_t)MF_WAVES * 64 * 18 + MF_COPIES * ((size_t)F * 21 + 2 * (size_t)n) + 2) + (size_t)MF_WAVES * 64 * MF_TBL_ROW +
         sizeof(uint32_t) * MF_WAVES * 64 + sizeof(int) * MF_WAVES;
}

__global__ __launch_bounds__(512) void ba_linearize_mfma_kernel(BaDev P, double radius, int first_pass) {
  extern __shared__ double lds[];
  const int n = P.n, F = P.K - 1;
  double* sZ = lds;                                  // [8][64][18]
  const int img = F * 21 + 2 * n;                    // one image: U upper triangles [F][21] | g_red [n] | g_c [n]
  double* sImg = sZ + MF_WAVES * 64 * 18;            // [MF_COPIES][img]
  double* sAcc = sImg + MF_COPIES * img;             // cost, sum g_p^2
  uint8_t* sTbl = reinterpret_cast<uint8_t*>(sAcc + 2);                         // [8][64][32]
  uint32_t* sMask = reinterpret_cast<uint32_t*>(sTbl + MF_WAVES * 64 * MF_TBL_ROW);  // [8][64]
  int* sNlm = reinterpret_cast<int*>(sMask + MF_WAVES * 64);                    // [8]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < MF_COPIES * img + 2; i += blockDim.x) sImg[i] = 0.0;
  __syncthreads();

  mf_d4 acc[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) acc[d] = mf_d4{0.0, 0.0, 0.0, 0.0};
  // operand coordinates of this lane for the row tile and the 5 column tiles: global row g = 16 t + (lane & 15)
  // -> (pose g / 6, component g % 6); k-slot kk = lane >> 4 (slot 3 is padding)
  const int kk = lane >> 4;
  // invalid lanes (padding rows / k-slot 3) read table byte MF_TBL_ROW-1, which no pose ever writes (-> 255 -> 0)
  int tix[5], go_[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const int g = 16 * ((wave + d) & 7) + (lane & 15);
    const int p = g / 6;
    const bool valid = g < n && kk < 3;
    tix[d] = valid ? p : MF_TBL_ROW - 1;
    go_[d] = valid ? 3 * (g - 6 * p) + kk : 0;
  }

  double lcost = 0.0, lgp2 = 0.0;
  const int groups = (P.C + MF_WAVES - 1) / MF_WAVES;
  // per-lane observation record of the NEXT group, fetched while the current group is in phase 2 (the
  // index -> landmark -> point chain is three dependent HBM/L2 round trips that two waves per SIMD cannot hide)
  struct Fetch { int c0, k, j, first, len; bool active; double u, v; D3 p; };
  auto fetch = [&](int grp) {
    Fetch f{0, 0, 0, lane, 0, false, 0.0, 0.0, D3{0, 0, 1}};
    const int chunk = grp * MF_WAVES + wave;
    if (grp < groups && chunk < P.C) {
      f.c0 = P.chunk_start[chunk];
      const int o = f.c0 + lane;
      f.active = o < P.chunk_start[chunk + 1];
      if (f.active) {
        f.k = P.obs_pose[o]; f.j = P.obs_point[o];
        const int l0 = P.lm_start[f.j];
        f.first = l0 - f.c0; f.len = P.lm_start[f.j + 1] - l0;
        f.p = D3{P.points[3 * f.j], P.points[3 * f.j + 1], P.points[3 * f.j + 2]};
        f.u = P.obs_uv[2 * o]; f.v = P.obs_uv[2 * o + 1];
      }
    }
    return f;
  };
  Fetch nxt = fetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int chunk = grp * MF_WAVES + wave;
    const Fetch cur = nxt;
    // ---- phase 1 -------------------------------------------------------------------------------
    {
      uint64_t* t8 = reinterpret_cast<uint64_t*>(sTbl + wave * 64 * MF_TBL_ROW);
#pragma unroll
      for (int i = 0; i < MF_TBL_ROW / 8; ++i) t8[lane * (MF_TBL_ROW / 8) + i] = ~0ull;
      sMask[wave * 64 + lane] = 0u;
    }
    int nlm = 0;
    if (chunk < P.C) {
      const bool active = cur.active;
      const int k = cur.k, j = cur.j, first = cur.first, len = cur.len;
      double r[2] = {0, 0}, Jc[12], Jp[6];
#pragma unroll
      for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) Jp[i] = 0.0;
      if (active) {
        eval_obs(P.poses + 7 * k, cur.p, cur.u, cur.v, P.f, P.cx, P.cy, k > 0, r, Jc, Jp);
        lcost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
      }
      int maxlen = len;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
      // landmark sums V = sum Jp^T Jp (symmetric: 6 terms), g_p = sum Jp^T r
      double V[9], gp[3];
      {
        double t[9] = {Jp[0] * Jp[0] + Jp[3] * Jp[3], Jp[0] * Jp[1] + Jp[3] * Jp[4], Jp[0] * Jp[2] + Jp[3] * Jp[5],
                       Jp[1] * Jp[1] + Jp[4] * Jp[4], Jp[1] * Jp[2] + Jp[4] * Jp[5], Jp[2] * Jp[2] + Jp[5] * Jp[5],
                       Jp[0] * r[0] + Jp[3] * r[1], Jp[1] * r[0] + Jp[4] * r[1], Jp[2] * r[0] + Jp[5] * r[1]};
        segment_totals<9>(t, lane, first, len > 0 ? first + len - 1 : lane, maxlen);
        V[0] = t[0]; V[1] = V[3] = t[1]; V[2] = V[6] = t[2]; V[4] = t[3]; V[5] = V[7] = t[4]; V[8] = t[5];
        gp[0] = t[6]; gp[1] = t[7]; gp[2] = t[8];
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
        if (lane == first) lgp2 += gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
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
      // V^-1 = Lc Lc^T
      double l00 = 0, l10 = 0, l20 = 0, l11 = 0, l21 = 0, l22 = 0;
      if (Vi[0] > 0) {
        l00 = sqrt(Vi[0]); l10 = Vi[3] / l00; l20 = Vi[6] / l00;
        const double d1 = Vi[4] - l10 * l10;
        if (d1 > 0) {
          l11 = sqrt(d1); l21 = (Vi[7] - l20 * l10) / l11;
          const double d2 = Vi[8] - l20 * l20 - l21 * l21;
          if (d2 > 0) l22 = sqrt(d2);
        }
      }
      const bool freep = active && k > 0;
      const int base = 6 * (k - 1);
      const unsigned long long flags = __ballot(active && lane == first);
      const int lm_local = __popcll(flags & ((2ull << lane) - 1ull)) - 1;
      double* sU = sImg + (lm_local & (MF_COPIES - 1)) * img;
      double* sGred = sU + F * 21;
      double* sGc = sGred + n;
      double Z[18];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double w[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) w[b] = freep ? (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b]) * s[b] : 0.0;
        Z[3 * a] = w[0] * l00 + w[1] * l10 + w[2] * l20;
        Z[3 * a + 1] = w[1] * l11 + w[2] * l21;
        Z[3 * a + 2] = w[2] * l22;
        if (freep) {
          // g_red part: -(W s) V^-1 (g_p s)
          double y = 0;
#pragma unroll
          for (int b = 0; b < 3; ++b) y += (w[0] * Vi[b] + w[1] * Vi[3 + b] + w[2] * Vi[6 + b]) * gps[b];
          atomicAdd(&sGred[base + a], -y);
          atomicAdd(&sGc[base + a], Jc[a] * r[0] + Jc[6 + a] * r[1]);
        }
      }
      if (freep) {
        double* u = sU + (k - 1) * 21;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) atomicAdd(&u[q++], Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
      }
      double* z = sZ + (size_t)(wave * 64 + lane) * 18;
#pragma unroll
      for (int i = 0; i < 18; ++i) z[i] = Z[i];
      nlm = __popcll(flags);
      if (freep) {
        sTbl[(wave * 64 + lm_local) * MF_TBL_ROW + (k - 1)] = (uint8_t)lane;
        atomicOr(&sMask[wave * 64 + lm_local], (1u << (base >> 4)) | (1u << ((base + 5) >> 4)));
      }
    }
    if (lane == 0) sNlm[wave] = nlm;
    __syncthreads();
    nxt = fetch(grp + gridDim.x);
    // ---- phase 2: wave = tile row --------------------------------------------------------------
    for (int c = 0; c < MF_WAVES; ++c) {
      const int nl = sNlm[c];
      const uint8_t* tb = sTbl + c * 64 * MF_TBL_ROW;
      const double* zc = sZ + (size_t)c * 64 * 18;
      // landmarks of this chunk whose poses touch my tile row, as a wave-uniform bit set
      const uint32_t mvec = lane < nl ? sMask[c * 64 + lane] : 0u;
      unsigned long long todo = __ballot((mvec >> wave) & 1u);
      // two-stage pipeline: the pose->lane bytes of the next landmark are in flight while the current
      // landmark's operands are read and multiplied; no lane-divergent control flow in the loop
      int srcN[5];
      int lmi = -1;
      if (todo) {
        lmi = __builtin_ctzll(todo); todo &= todo - 1ull;
        const uint8_t* tl = tb + lmi * MF_TBL_ROW;
#pragma unroll
        for (int d = 0; d < 5; ++d) srcN[d] = tl[tix[d]];
      }
      while (lmi >= 0) {
        const uint32_t mask = __builtin_amdgcn_readlane(mvec, lmi);
        int src[5];
        double op[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) { src[d] = srcN[d]; op[d] = zc[(src[d] & 63) * 18 + go_[d]]; }
        lmi = -1;
        if (todo) {
          lmi = __builtin_ctzll(todo); todo &= todo - 1ull;
          const uint8_t* tl = tb + lmi * MF_TBL_ROW;
#pragma unroll
          for (int d = 0; d < 5; ++d) srcN[d] = tl[tix[d]];
        }
#pragma unroll
        for (int d = 0; d < 5; ++d) op[d] = src[d] != 255 ? op[d] : 0.0;
        const double a_op = -op[0];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
          if (d == 4 && wave >= 4) continue;  // tile (w, w+4) is kept by w < 4 only
          if (!((mask >> ((wave + d) & 7)) & 1u)) continue;
          acc[d] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, op[d], acc[d], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- flush: tiles -> S (each unordered pose pair once; diagonal pose blocks in full) -----------
  double* S = P.pay1;
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (d == 4 && wave >= 4) continue;
    const int cc = (wave + d) & 7;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int row = 16 * wave + (lane >> 4) + 4 * rg, col = 16 * cc + (lane & 15);
      const double v = acc[d][rg];
      if (v == 0.0 || row >= n || col >= n) continue;
      const int pr = row / 6, pc = col / 6;
      if (d == 0) {
        if (pr <= pc) atomicAdd(&S[(size_t)row * n + col], v);
      } else if (pr == pc) {
        atomicAdd(&S[(size_t)row * n + col], v);
        atomicAdd(&S[(size_t)col * n + row], v);
      } else if (pr < pc) {
        atomicAdd(&S[(size_t)row * n + col], v);
      } else {
        atomicAdd(&S[(size_t)col * n + row], v);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { lcost += __shfl_xor(lcost, off); lgp2 += __shfl_xor(lgp2, off); }
  if (lane == 0) { atomicAdd(&sAcc[0], lcost); atomicAdd(&sAcc[1], lgp2); }
  __syncthreads();
  double* gGred = S + (size_t)n * n;
  double* gGc = gGred + n;
  double* gDU = gGc + n;
  for (int i = threadIdx.x; i < F * 21; i += blockDim.x) {
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < MF_COPIES; ++c) v += sImg[c * img + i];
    if (v == 0.0) continue;
    const int p = i / 21;
    int q = i - 21 * p, a = 0;
    while (q >= 6 - a) { q -= 6 - a; ++a; }
    const int b = a + q, ra = 6 * p + a, rb = 6 * p + b;
    atomicAdd(&S[(size_t)ra * n + rb], v);
    if (a != b) atomicAdd(&S[(size_t)rb * n + ra], v);
    else atomicAdd(&gDU[ra], v);
  }
  for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) {  // g_red then g_c: adjacent in the image and in the payload
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < MF_COPIES; ++c) v += sImg[c * img + F * 21 + i];
    if (v != 0.0) atomicAdd(&gGred[i], v);
  }
  if (threadIdx.x < 2) atomicAdd(&gDU[n + threadIdx.x], sAcc[threadIdx.x]);
}

__device__ __forceinline__ void ba_backsub_body(const BaDev& P, double radius) {
  const double* poses_ = P.poses;
  const double* points_ = P.points;
  const double* cand_poses_ = P.cand_poses;
  double* cand_points_ = P.cand_points;
  const double* dc_ = P.dc;
  if (P.lm) {
    if (ldv(&P.lm->done) || !ldv(&P.lm->step_valid)) return;
    const int c = ldv(&P.lm->cur), nn = P.n > 0 ? P.n : 1;
    poses_ = P.step[c] + nn; points_ = P.pts[c];
    dc_ = P.step[1 - c]; cand_poses_ = P.step[1 - c] + nn; cand_points_ = P.pts[1 - c];
    radius = ldv(&P.lm->radius);
  } else if (P.step_in) {
    dc_ = P.step_in;
    cand_poses_ = P.step_in + (P.n > 0 ? P.n : 1);
    // the candidate becomes the linearisation point if the step is accepted: leave a device copy for later launches
    if (blockIdx.x == 0)
      for (int i = threadIdx.x; i < 7 * P.K; i += blockDim.x) P.cand_poses[i] = cand_poses_[i];
  }
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
      p = D3{points_[3 * j], points_[3 * j + 1], points_[3 * j + 2]};
      u = P.obs_uv[2 * o]; v = P.obs_uv[2 * o + 1];
      eval_obs(poses_ + 7 * k, p, u, v, P.f, P.cx, P.cy, k > 0, r, Jc, Jp);
      if (k > 0) {
        const double* d = dc_ + 6 * (k - 1);
#pragma unroll
        for (int a = 0; a < 6; ++a) { jd[0] += Jc[a] * d[a]; jd[1] += Jc[6 + a] * d[a]; }
      }
    }
    int maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
    double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, wd[3] = {0, 0, 0};
    if (!P.det) {
      double t[12] = {Jp[0] * Jp[0] + Jp[3] * Jp[3], Jp[0] * Jp[1] + Jp[3] * Jp[4], Jp[0] * Jp[2] + Jp[3] * Jp[5],
                      Jp[1] * Jp[1] + Jp[4] * Jp[4], Jp[1] * Jp[2] + Jp[4] * Jp[5], Jp[2] * Jp[2] + Jp[5] * Jp[5],
                      Jp[0] * r[0] + Jp[3] * r[1], Jp[1] * r[0] + Jp[4] * r[1], Jp[2] * r[0] + Jp[5] * r[1],
                      Jp[0] * jd[0] + Jp[3] * jd[1], Jp[1] * jd[0] + Jp[4] * jd[1], Jp[2] * jd[0] + Jp[5] * jd[1]};
      segment_totals<12>(t, lane, first, len > 0 ? first + len - 1 : lane, maxlen);
      V[0] = t[0]; V[1] = V[3] = t[1]; V[2] = V[6] = t[2]; V[4] = t[3]; V[5] = V[7] = t[4]; V[8] = t[5];
      gp[0] = t[6]; gp[1] = t[7]; gp[2] = t[8]; wd[0] = t[9]; wd[1] = t[10]; wd[2] = t[11];
    }
    for (int t = 0; t < (P.det ? maxlen : 0); ++t) {
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
      if (lane == first) { cand_points_[3 * j] = np[0]; cand_points_[3 * j + 1] = np[1]; cand_points_[3 * j + 2] = np[2]; }
      double r0, r1;
      reproj_residual(cand_poses_ + 7 * k, D3{np[0], np[1], np[2]}, u, v, P.f, P.cx, P.cy, r0, r1);
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
  if (P.det) {
    if (!P.arrive2_target) return;  // ba_reduce2_kernel follows as its own launch
    // ---- fused reduce2: the workgroup that finishes last forms the declared-order sums over lmV and publishes.
    // Writers: stores, agent-scope fence, arrive.  Reader: sees the final count, fences (acquire), reads lmV —
    // same code and order as ba_reduce2_body, so the result is bit-identical to the two-launch form.
    __shared__ int sLast;
    __shared__ double sP2[RSEG][4];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
      sLast = __hip_atomic_fetch_add(P.arrive2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == P.arrive2_target;
    __syncthreads();
    if (!sLast) return;
    __threadfence();
    const int F = P.K - 1, tid = threadIdx.x;
    const int dl = F * F + F;  // the landmark list
    const int e0 = P.list_start[dl], len = P.list_start[F * F + F + 1 + 1 + dl] - e0;
    const int seglen = (len + RSEG - 1) / RSEG;
    for (int item = tid; item < RSEG * 4; item += (int)blockDim.x) {
      const int seg = item / 4, e = item % 4;
      double acc = 0.0;
      const int b0 = seg * seglen, b1 = min(len, (seg + 1) * seglen);
      const double* src = P.lmV + 4 * (size_t)e0 + e;
      for (int q0 = b0; q0 < b1; q0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = q0 + u < b1 ? src[4 * (size_t)(q0 + u)] : 0.0;  // plain loads: the fence above is the acquire
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (q0 + u < b1) acc += v[u];
      }
      sP2[seg][e] = acc;
    }
    __syncthreads();
    if (tid < 4) {
      double acc = 0.0;
      for (int sg = 0; sg < RSEG; ++sg) acc += sP2[sg][tid];
      P.pay2_out[tid] = acc;
    }
    if (P.flag2 && tid < 64) {  // lanes 0-3 of wave 0 stored the payload; fence, then lane 0 publishes
      __threadfence_system();
      if (tid == 0) __hip_atomic_store(P.flag2, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
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
__device__ __forceinline__ void ba_reduce1_body(const BaDev& P) {
  if (P.lm && ldv(&P.lm->done)) return;
  if ((int)blockIdx.x >= (P.K - 1) * (P.K - 1) + (P.K - 1) + 1) return;
  __shared__ double sP[RSEG][36];
  const int F = P.K - 1, n = P.n, tid = threadIdx.x, d = blockIdx.x;
  const int width = d < F * F ? 36 : (d < F * F + F ? 18 : 2);
  const int stride = d < F * F ? 36 : (d < F * F + F ? 18 : 4);
  const double* base = d < F * F ? P.pairB : (d < F * F + F ? P.obsV : P.lmV);
  const int nd = F * F + F + 1;
  const int e0 = P.list_start[d], len = P.list_start[nd + 1 + d] - e0;
  const int seglen = (len + RSEG - 1) / RSEG;
  for (int item = tid; item < RSEG * width; item += (int)blockDim.x) {  // one pass with 1024 threads, two with 512
    const int seg = item / width, e = item % width;
    double acc = 0.0;
    const int b0 = seg * seglen, b1 = min(len, (seg + 1) * seglen);
    // 16 independent loads in flight, adds strictly in list order.  The row pointer advances by addition: a
    // per-element 64-bit index multiply is a quarter-rate instruction and was most of this loop's ALU time.
    const double* pq = base + ((size_t)e0 + (size_t)b0) * stride + e;
    for (int q0 = b0; q0 < b1; q0 += 16, pq += 16 * stride) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = q0 + u < b1 ? pq[u * stride] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u)
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
  if (P.flag1) {
    // payload (host memory) first, system-scope fence, then arrive; the last workgroup publishes the sequence word
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(P.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == P.arrive_target) {
        __threadfence_system();
        __hip_atomic_store(P.flag1, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

__device__ __forceinline__ void ba_reduce2_body(const BaDev& P) {
  if (P.lm && (ldv(&P.lm->done) || !ldv(&P.lm->step_valid))) return;
  __shared__ double sP[RSEG][4];
  __shared__ double sOut[4];
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
    sOut[tid] = acc;
    if (!P.lm) P.pay2_out[tid] = acc;
  }
  if (!P.lm) {
    if (P.flag2 && tid < 64) {  // lanes 0-3 of this wave stored the payload; fence, then lane 0 publishes
      __threadfence_system();
      if (tid == 0) __hip_atomic_store(P.flag2, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  __syncthreads();
  if (tid != 0) return;
  // ---- step control (same statements, same order as the host loop in ba_lm / oracle/ora_ba.cpp)
  LmDev& S = *P.lm;
  const int K = P.K, nn = P.n > 0 ? P.n : 1;
  const int cur = ldv(&S.cur);
  double radius = ldv(&S.radius), decrease_factor = ldv(&S.decrease_factor), cost = ldv(&S.cost);
  const double* poses = P.step[cur] + nn;
  const double* cand = P.step[1 - cur] + nn;
  const double cost_new = sOut[0];
  const double model_change = ldv(&S.mcc) + sOut[1];
  double step2 = sOut[2], x2 = sOut[3];
  for (int k = 1; k < K; ++k)
    for (int a = 0; a < 7; ++a) {
      const double pv = ldv(&poses[7 * k + a]);
      const double dd = ldv(&cand[7 * k + a]) - pv;
      step2 += dd * dd;
      x2 += pv * pv;
    }
  auto publish = [&](int cur_now, double cost_now) {
    P.done_host[1] = ldv(&S.iterations); P.done_host[2] = ldv(&S.successful); P.done_host[3] = S.termination; P.done_host[4] = cur_now;
    double* dh = reinterpret_cast<double*>(P.done_host + 6);
    dh[0] = ldv(&S.initial_cost); dh[1] = cost_now;
    __threadfence_system();
    P.done_host[0] = 1;
  };
  if (!(model_change > 0)) { S.radius = radius / decrease_factor; S.decrease_factor = decrease_factor * 2; return; }
  if (sqrt(step2) <= S.parameter_tol * (sqrt(x2) + S.parameter_tol)) { S.termination = 0; S.done = 1; publish(cur, cost); return; }
  const double cost_change = cost - cost_new;
  if (fabs(cost_change) <= S.function_tol * cost) {
    int c2 = cur;
    if (cost_change > 0) { c2 = 1 - cur; S.cur = c2; S.cost = cost_new; cost = cost_new; }
    S.termination = 0; S.done = 1; publish(c2, cost);
    return;
  }
  const double rho = cost_change / model_change;
  if (rho > MIN_REL_DECREASE) {
    S.cur = 1 - cur; S.cost = cost_new; S.successful = ldv(&S.successful) + 1;
    const double t = 2.0 * rho - 1.0;
    radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
    S.radius = fmin(MAX_RADIUS, radius);
    S.decrease_factor = 2.0;
    S.grad_check = 1;
  } else {
    S.radius = radius / decrease_factor; S.decrease_factor = decrease_factor * 2;
  }
}

// Reduced camera system: scaling, LM diagonal, Cholesky (column sweep, same operation order as the host
// cholesky_solve), pose step and candidate poses.  One workgroup (WAVE = false) or ONE wavefront of a larger
// workgroup (WAVE = true, the persistent kernel: LDS traffic of a single wave is ordered, so a workgroup-scope
// fence + wave barrier replaces s_barrier and the other 15 wavefronts do not have to take part).
template <bool WAVE>
__device__ __forceinline__ void ba_solve_body(const BaDev& P) {
  LmDev& S = *P.lm;
  auto SYNC = [&]() {
    if (WAVE) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
      __syncthreads();
    }
  };
  if (ldv(&S.done)) return;
  extern __shared__ double sm[];  // Sm (n*n) | b (n) | sc (n) | Df (n)
  const int n = P.n, K = P.K, tid = threadIdx.x, nn = n > 0 ? n : 1;
  const int T = WAVE ? 64 : (int)blockDim.x;
  double* Sm = sm;
  double* sB = sm + (size_t)n * n;
  double* sSc = sB + nn;
  double* sDf = sSc + nn;
  __shared__ int sFail;
  const double* pay = P.pay1;
  const double* gred = pay + (size_t)n * n;
  const double* gc = gred + n;
  const double* dU = gc + n;
  auto publish = [&]() {
    P.done_host[1] = S.iterations; P.done_host[2] = S.successful; P.done_host[3] = S.termination; P.done_host[4] = S.cur;
    double* dh = reinterpret_cast<double*>(P.done_host + 6);
    dh[0] = S.initial_cost; dh[1] = S.cost;
    __threadfence_system();
    P.done_host[0] = 1;
  };
  if (tid == 0) {
    sFail = 0;
    S.step_valid = 0;
    bool check = false;
    if (!ldv(&S.have_scale)) {
      for (int a = 0; a < n; ++a) S.sc[a] = 1.0 / (1.0 + sqrt(ldv(&dU[a])));
      S.have_scale = 1;
      S.cost = ldv(&pay[(size_t)n * n + 3 * n]);
      S.initial_cost = S.cost;
      check = true;
    } else if (ldv(&S.grad_check)) {
      check = true;
    }
    S.grad_check = 0;
    if (check) {
      double g2 = ldv(&pay[(size_t)n * n + 3 * n + 1]);
      for (int a = 0; a < n; ++a) { const double g = ldv(&gc[a]); g2 += g * g; }
      if (sqrt(g2) <= S.gradient_tol) { S.termination = 0; S.done = 1; publish(); sFail = 2; }
    }
    if (!sFail) {
      if (ldv(&S.iterations) >= S.max_iterations) { S.termination = 1; S.done = 1; publish(); sFail = 2; }
      else if (ldv(&S.radius) <= MIN_RADIUS) { S.termination = 0; S.done = 1; publish(); sFail = 2; }
      else S.iterations = ldv(&S.iterations) + 1;
    }
  }
  SYNC();
  if (sFail) return;
  const double radius = ldv(&S.radius);
  // Jacobi scales were written by lane 0 (possibly just now): re-read them through L2, keep copies in LDS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int a = tid; a < n; a += T) {
    const double sca = ldv(&S.sc[a]);
    sSc[a] = sca;
    sDf[a] = fmin(fmax(ldv(&dU[a]) * sca * sca, MIN_DIAG), MAX_DIAG) / radius;
  }
  SYNC();
  for (int i = tid; i < n * n; i += T) {
    const int a = i / n, b = i % n;
    double v = ldv(&pay[i]) * sSc[a] * sSc[b];
    if (a == b) v += sDf[a];
    Sm[i] = v;
  }
  for (int a = tid; a < n; a += T) sB[a] = -(ldv(&gred[a]) + ldv(&gc[a])) * sSc[a];
  SYNC();
  // Cholesky, column by column (row i's dot products run sequentially in k, as on the host).  The right-hand
  // side rides along as an extra row: y_j = (b_j - sum_{k<j} L_jk y_k) / L_jj is exactly the forward
  // substitution, product for product.
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      double sd = Sm[(size_t)j * n + j];
      for (int k = 0; k < j; ++k) sd -= Sm[(size_t)j * n + k] * Sm[(size_t)j * n + k];
      if (!(sd > 0)) sFail = 1; else Sm[(size_t)j * n + j] = sqrt(sd);
    }
    SYNC();
    if (sFail) break;
    const double l = Sm[(size_t)j * n + j];
    for (int i = j + 1 + tid; i <= n; i += T) {
      if (i < n) {
        double v = Sm[(size_t)i * n + j];
        for (int k = 0; k < j; ++k) v -= Sm[(size_t)i * n + k] * Sm[(size_t)j * n + k];
        Sm[(size_t)i * n + j] = v / l;
      } else {  // the augmented row: forward substitution of column j
        double v = sB[j];
        for (int k = 0; k < j; ++k) v -= Sm[(size_t)j * n + k] * sB[k];
        sB[j] = v / l;
      }
    }
    SYNC();
  }
  if (sFail) {  // not positive definite: an invalid step
    if (tid == 0) { S.radius = ldv(&S.radius) / S.decrease_factor; S.decrease_factor *= 2; }
    return;
  }
  // backward substitution (k descending) as a column sweep
  for (int k = n - 1; k >= 0; --k) {
    if (tid == 0) sB[k] = sB[k] / Sm[(size_t)k * n + k];
    SYNC();
    const double bk = sB[k];
    for (int i = tid; i < k; i += T) sB[i] -= Sm[(size_t)k * n + i] * bk;
    SYNC();
  }
  // step, model change (pose part), candidate poses
  const int c = ldv(&S.cur);
  double* dc = P.step[1 - c];
  const double* poses = P.step[c] + nn;
  double* cand = P.step[1 - c] + nn;
  if (tid == 0) {
    double mcc = 0;
    for (int a = 0; a < n; ++a) {
      mcc += 0.5 * sB[a] * (sDf[a] * sB[a] - ldv(&gc[a]) * sSc[a]);
      S.Df[a] = sDf[a];
    }
    S.mcc = mcc;
    S.step_valid = 1;
  }
  for (int a = tid; a < n; a += T) dc[a] = sB[a] * sSc[a];
  for (int k = tid; k < K; k += T) {
    double pk[7], dk[6], out[7];
    for (int a = 0; a < 7; ++a) pk[a] = ldv(&poses[7 * k + a]);
    if (k == 0) { for (int a = 0; a < 7; ++a) cand[a] = pk[a]; }
    else {
      for (int a = 0; a < 6; ++a) dk[a] = sB[6 * (k - 1) + a] * sSc[6 * (k - 1) + a];
      plus_pose(pk, dk, out);
      for (int a = 0; a < 7; ++a) cand[7 * k + a] = out[a];
    }
  }
}

// By-value wrappers (host-driven loop) and by-pointer wrappers (hipGraph replay: the parameters live in
// device memory so the instantiated graph never has to be updated between solves).
__global__ __launch_bounds__(256) void ba_linearize_kernel(BaDev P, double radius, int first_pass) { ba_linearize_body(P, radius, first_pass); }
__global__ __launch_bounds__(256) void ba_backsub_kernel(BaDev P, double radius) { ba_backsub_body(P, radius); }
__global__ __launch_bounds__(1024) void ba_reduce1_kernel(BaDev P) { ba_reduce1_body(P); }
__global__ __launch_bounds__(128) void ba_reduce2_kernel(BaDev P) { ba_reduce2_body(P); }
__global__ __launch_bounds__(128) void ba_solve_kernel(BaDev P) { ba_solve_body<false>(P); }
__global__ __launch_bounds__(64) void ba_linearize_gkernel(const BaDev* __restrict__ Pp) { const BaDev P = *Pp; ba_linearize_body(P, 0.0, 0); }
__global__ __launch_bounds__(64) void ba_backsub_gkernel(const BaDev* __restrict__ Pp) { const BaDev P = *Pp; ba_backsub_body(P, 0.0); }
__global__ __launch_bounds__(1024) void ba_reduce1_gkernel(const BaDev* __restrict__ Pp) { const BaDev P = *Pp; ba_reduce1_body(P); }
__global__ __launch_bounds__(128) void ba_reduce2_gkernel(const BaDev* __restrict__ Pp) { const BaDev P = *Pp; ba_reduce2_body(P); }
__global__ __launch_bounds__(128) void ba_solve_gkernel(const BaDev* __restrict__ Pp) { const BaDev P = *Pp; ba_solve_body<false>(P); }

// ---- persistent single-launch solve: the five phases of an LM iteration separated by grid barriers
// (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "barrier-counter": every storing wave drains,
// workgroup barrier, lane-0 agent release, agent atomic arrive, relaxed sc1 poll with s_sleep, ONE agent acquire,
// workgroup barrier).  All workgroups are trivially co-resident (grid <= 128 x 512 threads on 256 CUs); every spin
// is bounded and raises LmDev::abort instead of hanging.
__device__ __forceinline__ bool grid_barrier(LmDev* lm, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int sAbort;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&lm->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int ab = 0;
    while (__hip_atomic_load(&lm->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 21) || __hip_atomic_load(&lm->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(&lm->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ab = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sAbort = ab;
  }
  __syncthreads();
  return sAbort == 0;
}

__global__ __launch_bounds__(512) void ba_persistent_kernel(BaDev P, int max_loops) {
  unsigned epoch = 0;
  const unsigned G = gridDim.x;
  for (int it = 0; it < max_loops; ++it) {
    if (ldv(&P.lm->done)) break;  // written by workgroup 0 before the last barrier: uniform across the grid
    ba_linearize_body(P, 0.0, 0);
    if (!grid_barrier(P.lm, ++epoch * G)) break;
    ba_reduce1_body(P);
    if (!grid_barrier(P.lm, ++epoch * G)) break;
    if (blockIdx.x == 0 && threadIdx.x < 64) ba_solve_body<true>(P);
    if (!grid_barrier(P.lm, ++epoch * G)) break;
    ba_backsub_body(P, 0.0);
    if (!grid_barrier(P.lm, ++epoch * G)) break;
    if (blockIdx.x == 0) ba_reduce2_body(P);
    if (!grid_barrier(P.lm, ++epoch * G)) break;
  }
}

// ----------------------------------------------------------------------------- host side
namespace {
bool cholesky_solve(std::vector<double>& A, std::vector<double>& b, int n) { return svo_host_cholesky_solve(A.data(), b.data(), n); }
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
  size_t cap_points = 0, cap_obs = 0, cap_chunks = 0, cap_pay1 = 0, cap_pairs = 0;
  // host mirrors of the loaded problem
  std::vector<double> h_poses, h_cand_poses;
  int n_points = 0;
  LmDev* d_lm = nullptr;
  uint8_t* d_arena = nullptr;     // all per-solve inputs in one allocation: one H2D per solve
  uint8_t* h_arena = nullptr;     // pinned staging image of the arena
  size_t arena_cap = 0;
  BaDev* d_params = nullptr;      // device copy of the kernel parameters (graph kernels read it)
  hipGraphExec_t graph_exec = nullptr;
  double t_launch = 0, t_sync = 0, t_upload = 0, t_total = 0; long n_chunks = 0, n_solves = 0;
  double t_lin = 0, t_host = 0, t_back = 0; long n_lin = 0, n_back = 0;  // host-driven loop phases (SVO_TIMING)
  int graph_threads = 0;          // solve-kernel block size baked into the graph
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
  unsigned* d_arrive = nullptr; unsigned arrive_total = 0, arrive2_total = 0; int seq = 0;
  bool upload_pending = false;  // H2D of the problem image enqueued, not yet known complete
  bool mfma_ok = false;   // bulk problem eligible for ba_linearize_mfma_kernel (n <= 128, one observation per (landmark, pose))
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
  A(d.sp, double, 3 * ba->cap_points);
  A(d.pay1, double, ba->cap_pay1); A(d.pay2, double, 4);
  A(ba->d_lm, LmDev, 1); A(ba->d_params, BaDev, 1);
  A(ba->d_arrive, unsigned, 2);
  SVO_HIP_CHECK(ctx, hipMemset(ba->d_arrive, 0, 2 * sizeof(unsigned)));
  A(d.obsV, double, 18 * ba->cap_obs);
  A(d.lmV, double, 4 * ba->cap_points);
#undef A
  {
    // pre-size the per-solve input arena and the pair-block store for window-shaped problems (every landmark seen
    // at most once per pose) so that the hot path never allocates; bulk problems beyond this grow lazily
    const size_t M = ba->cap_obs, Kc = (size_t)Kmax;
    const size_t pairs = M * (Kc + 1) / 2 + 64;
    const size_t est = 16 * 3 * ba->cap_points + 16 * M + 8 * M + 4 * (ba->cap_points + 1) + 4 * (M + 2) * 3 + 8 * pairs + 8 * (Kc * Kc + Kc + 2) + 16 * 256;
    if (est < ((size_t)512 << 20)) {
      ba->arena_cap = est;
      SVO_HIP_CHECK(ctx, hipMalloc((void**)&ba->d_arena, ba->arena_cap));
      SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_arena, ba->arena_cap, hipHostMallocDefault));
    }
    if (2 * pairs * 288 < ((size_t)512 << 20)) {
      ba->cap_pairs = 2 * pairs;
      SVO_HIP_CHECK(ctx, hipMalloc((void**)&d.pairB, sizeof(double) * 36 * ba->cap_pairs));
    }
  }
  {
    // SVO_BA_CU_SHARE=n (n = 8 on MI355X: one shader engine of every XCD; see include/svo.h): window-sized adjusters
    // run on their own n CUs of every 32 and the context's stream on the others, so that the LM loop's small dependent
    // kernels never queue behind other stereo streams' wide LK launches.  Bulk-sized adjusters keep the whole GPU.
    const char* e = getenv("SVO_BA_CU_SHARE");
    const int nres = e ? atoi(e) : 0;
    if (nres > 0 && nres < 32 && ba->max_obs <= 100000) {
      uint32_t mask[8];
      for (int i = 0; i < 8; ++i) mask[i] = (1u << nres) - 1u;
      SVO_HIP_CHECK(ctx, hipExtStreamCreateWithCUMask(&ba->stream, 8, mask));
    } else {
      SVO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ba->stream, hipStreamNonBlocking));
    }
  }
  ba->pin_bytes = sizeof(double) * (ba->cap_pay1 + 1400 + 16 * (size_t)Kmax) + 2 * sizeof(BaDev);  // payloads | flags | LmDev image | poses
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
  o->accumulation = SVO_BA_ACC_AUTO;
}

extern "C" int svo_ba_create(svo_ctx* ctx, svo_ba** out, int window_size, const svo_camera_info* cam,
                             const svo_ba_options* opt, int max_landmarks, int max_observations) {
  if (!ctx || !out || !cam) return SVO_ERR_INVALID;
  svo_use_device(ctx);
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
  if (getenv("SVO_TIMING") && ba->n_lin)
    fprintf(stderr, "[svo ba] host loop: linearize+reduce+D2H %.1f us x %ld, host solve %.1f us, backsub+D2H %.1f us x %ld\n",
            1e3 * ba->t_lin / ba->n_lin, ba->n_lin, ba->n_back ? 1e3 * ba->t_host / ba->n_back : 0.0, ba->n_back ? 1e3 * ba->t_back / ba->n_back : 0.0, ba->n_back);
  if (getenv("SVO_TIMING") && ba->n_chunks)
    fprintf(stderr, "[svo ba] solves %ld chunks %ld graph-launch %.3f ms sync %.3f ms upload %.3f ms total %.3f ms\n", ba->n_solves, ba->n_chunks,
            ba->t_launch, ba->t_sync, ba->t_upload, ba->t_total);
  if (ba->graph_exec) (void)hipGraphExecDestroy(ba->graph_exec);
  void* ptrs[] = {ba->d_arrive, ba->d_lm, ba->d_params, ba->step_buf[0], ba->step_buf[1], d.sp, d.pay1, d.pay2, d.pairB, d.obsV, d.lmV, ba->d_arena};
  if (ba->h_arena) (void)hipHostFree(ba->h_arena);
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
  if (ba->upload_pending) { SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream)); ba->upload_pending = false; }
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
  std::vector<int32_t> pair_base_v, pair_pos_v, obs_pos_v, ls;
  bool has_empty_landmark = false;
  for (int j = 0; j < npts; ++j) has_empty_landmark |= lm_start[j + 1] == lm_start[j];
  // deterministic mode: pair slots + destination lists (landmark order) if they fit
  {
    const int F = K - 1;
    std::vector<int32_t> pair_base((size_t)M + 1, 0);
    for (int j = 0; j < npts; ++j)
      for (int o = lm_start[j]; o < lm_start[j + 1]; ++o) pair_base[o + 1] = pair_base[o] + (lm_start[j + 1] - o);
    const size_t n_pairs = (size_t)pair_base[M];
    d.det = n_pairs <= ((size_t)1 << 21) ? 1 : 0;  // <= 604 MB of pair blocks
    if (ba->opt.accumulation == SVO_BA_ACC_ATOMICS || ba->opt.accumulation == SVO_BA_ACC_MFMA) d.det = 0;
    SVO_REQUIRE(ctx, !(ba->opt.accumulation == SVO_BA_ACC_DETERMINISTIC && !d.det), "ba: problem too large for deterministic accumulation");
    {
      bool dup = false;
      for (int j = 0; j < npts && !dup; ++j) {
        uint64_t seen = 0;
        for (int o = lm_start[j]; o < lm_start[j + 1]; ++o) { const uint64_t bit = 1ull << op[o]; dup |= (seen & bit) != 0; seen |= bit; }
      }
      ba->mfma_ok = !dup && d.n <= 128 && d.n > 0 && ba->opt.accumulation != SVO_BA_ACC_ATOMICS;
      SVO_REQUIRE(ctx, !(ba->opt.accumulation == SVO_BA_ACC_MFMA && !ba->mfma_ok), "ba: MFMA accumulation needs <= 22 poses and one observation per (landmark, pose)");
    }
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
      if (rows > ba->cap_pairs) {
        if (d.pairB) (void)hipFree(d.pairB);
        d.pairB = nullptr;
        ba->cap_pairs = rows + rows / 4 + 1024;
        SVO_HIP_CHECK(ctx, hipMalloc((void**)&d.pairB, sizeof(double) * 36 * ba->cap_pairs));
      }
      ls.assign(2 * (size_t)nd + 2, 0);  // list_start[q] = begin(q), list_start[nd + 1 + q] = end(q)
      for (int q = 0; q < nd; ++q) { ls[q] = ba->h_list_begin[q]; ls[nd + 1 + q] = ba->h_list_end[q]; }
      pair_base_v.swap(pair_base); pair_pos_v.swap(pair_pos); obs_pos_v.swap(obs_pos);
    }
  }
  // ---- one pinned staging image, one H2D: [points | points (candidate copy) | uv | obs_pose | obs_point |
  //      lm_start | chunk_start | pair_base | obs_pos | pair_pos | list_start | poses]
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_pts = off; off = al(off + sizeof(double) * 3 * (size_t)npts);
  const size_t o_cpts = off; off = al(off + sizeof(double) * 3 * (size_t)npts);
  const size_t o_uv = off; off = al(off + sizeof(double) * 2 * (size_t)M);
  const size_t o_op = off; off = al(off + sizeof(int32_t) * (size_t)M);
  const size_t o_oj = off; off = al(off + sizeof(int32_t) * (size_t)M);
  const size_t o_lm = off; off = al(off + sizeof(int32_t) * ((size_t)npts + 1));
  const size_t o_ch = off; off = al(off + sizeof(int32_t) * chunks.size());
  const size_t o_pb = off; off = al(off + sizeof(int32_t) * pair_base_v.size());
  const size_t o_ob = off; off = al(off + sizeof(int32_t) * obs_pos_v.size());
  const size_t o_pp = off; off = al(off + sizeof(int32_t) * pair_pos_v.size());
  const size_t o_ls = off; off = al(off + sizeof(int32_t) * ls.size());
  const size_t total = off;
  if (total > ba->arena_cap) {
    if (ba->d_arena) (void)hipFree(ba->d_arena);
    if (ba->h_arena) (void)hipHostFree(ba->h_arena);
    ba->d_arena = nullptr; ba->h_arena = nullptr;
    ba->arena_cap = total + total / 4 + 4096;
    SVO_HIP_CHECK(ctx, hipMalloc((void**)&ba->d_arena, ba->arena_cap));
    SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_arena, ba->arena_cap, hipHostMallocDefault));
  }
  uint8_t* h = ba->h_arena;
  if (npts) { memcpy(h + o_pts, points3, sizeof(double) * 3 * (size_t)npts); memcpy(h + o_cpts, points3, sizeof(double) * 3 * (size_t)npts); }
  if (M) {
    memcpy(h + o_uv, uv, sizeof(double) * 2 * (size_t)M);
    memcpy(h + o_op, op, sizeof(int32_t) * (size_t)M);
    memcpy(h + o_oj, oj, sizeof(int32_t) * (size_t)M);
  }
  memcpy(h + o_lm, lm_start.data(), sizeof(int32_t) * lm_start.size());
  memcpy(h + o_ch, chunks.data(), sizeof(int32_t) * chunks.size());
  if (!pair_base_v.empty()) memcpy(h + o_pb, pair_base_v.data(), sizeof(int32_t) * pair_base_v.size());
  if (!obs_pos_v.empty()) memcpy(h + o_ob, obs_pos_v.data(), sizeof(int32_t) * obs_pos_v.size());
  if (!pair_pos_v.empty()) memcpy(h + o_pp, pair_pos_v.data(), sizeof(int32_t) * pair_pos_v.size());
  if (!ls.empty()) memcpy(h + o_ls, ls.data(), sizeof(int32_t) * ls.size());
  uint8_t* D = ba->d_arena;
  d.points = (double*)(D + o_pts); d.cand_points = (double*)(D + o_cpts); d.obs_uv = (double*)(D + o_uv);
  d.obs_pose = (int32_t*)(D + o_op); d.obs_point = (int32_t*)(D + o_oj); d.lm_start = (int32_t*)(D + o_lm);
  d.chunk_start = (int32_t*)(D + o_ch); d.pair_base = (int32_t*)(D + o_pb); d.obs_pos = (int32_t*)(D + o_ob);
  d.pair_pos = (int32_t*)(D + o_pp); d.list_start = (int32_t*)(D + o_ls);
  ba->h_poses.assign(poses7, poses7 + 7 * (size_t)K);
  ba->h_cand_poses = ba->h_poses;
  d.poses = ba->step_buf[0] + (d.n > 0 ? d.n : 1);
  d.cand_poses = ba->step_buf[1] + (d.n > 0 ? d.n : 1);
  d.dc = ba->step_buf[1];
  double* h_pose_stage = ba->h_pin;  // payload area is idle during the upload
  memcpy(h_pose_stage, poses7, sizeof(double) * 7 * K);
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(D, h, total, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(d.poses, h_pose_stage, sizeof(double) * 7 * K, hipMemcpyHostToDevice, st));
  if (d.det && has_empty_landmark && npts) SVO_HIP_CHECK(ctx, hipMemsetAsync(d.lmV, 0, sizeof(double) * 4 * npts, st));
  // no wait here: the pinned staging images are next touched by the host after the solve that follows has
  // drained this stream (ba_lm), and the payload area is written by kernels ordered after the pose copy
  ba->upload_pending = true;
  return SVO_OK;
}

// Device-resident LM loop: the host only enqueues iteration chunks and polls a pinned flag.
static int ba_lm_device(svo_ba* ba, svo_ba_summary* sum) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  d.arrive2_target = 0; d.step_in = nullptr; d.flag1 = d.flag2 = nullptr;  // host-loop-only features
  hipStream_t st = ba->stream;
  const int n = d.n, K = d.K, nn = n > 0 ? n : 1;
  const auto t_begin = std::chrono::steady_clock::now();
  int* done_host = reinterpret_cast<int*>(ba->h_pin + ba->cap_pay1 + 16);
  double* h_fin = reinterpret_cast<double*>(done_host + 6);
  LmDev init;
  memset(&init, 0, sizeof(init));
  init.radius = ba->opt.initial_radius; init.decrease_factor = 2.0;
  init.function_tol = ba->opt.function_tolerance; init.gradient_tol = ba->opt.gradient_tolerance;
  init.parameter_tol = ba->opt.parameter_tolerance; init.max_iterations = ba->opt.max_iterations;
  init.termination = 1;
  LmDev* h_init = reinterpret_cast<LmDev*>(ba->h_pin + ba->cap_pay1 + 32);
  *h_init = init;
  memset(done_host, 0, 6 * sizeof(int));
  h_fin[0] = h_fin[1] = 0.0;
  d.lm = ba->d_lm;
  d.pts[0] = d.points; d.pts[1] = d.cand_points;
  d.step[0] = ba->step_buf[0]; d.step[1] = ba->step_buf[1];  // ba_upload put the poses into step_buf[0] + nn
  d.done_host = done_host;
  d.pay1_out = d.pay1; d.pay2_out = d.pay2;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_lm, h_init, sizeof(LmDev), hipMemcpyHostToDevice, st));
  const size_t solve_lds = ((size_t)n * n + n + 8) * sizeof(double);
  // One hipGraph = `chunk` LM iterations x 5 kernels, captured once per adjuster with worst-case grids
  // (extra workgroups exit immediately); replayed with ONE launch call per chunk.
  const int chunk = 3;  // iterations between two polls of the done flag (typical solve: 5-6 iterations)
  const int Kmax = ba->max_poses, nmax = 6 * (Kmax - 1);
  const int nd_max = (Kmax - 1) * (Kmax - 1) + (Kmax - 1) + 1;
  const size_t solve_lds_max = ((size_t)nmax * nmax + 3 * nmax + 8) * sizeof(double);
  const int solve_threads = nmax < 64 ? 64 : 128;
  if (!ba->graph_exec) {
    if (solve_lds_max > 64 * 1024)
      SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ba_solve_gkernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds_max));
    hipGraph_t graph = nullptr;
    SVO_HIP_CHECK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int lin_grid = 256;  // grid-stride over the wave chunks
    for (int it = 0; it < chunk; ++it) {
      hipLaunchKernelGGL(ba_linearize_gkernel, dim3(lin_grid), dim3(64), 64, st, ba->d_params);
      hipLaunchKernelGGL(ba_reduce1_gkernel, dim3(nd_max), dim3(1024), 0, st, ba->d_params);
      hipLaunchKernelGGL(ba_solve_gkernel, dim3(1), dim3(solve_threads), solve_lds_max, st, ba->d_params);
      hipLaunchKernelGGL(ba_backsub_gkernel, dim3(lin_grid), dim3(64), 0, st, ba->d_params);
      hipLaunchKernelGGL(ba_reduce2_gkernel, dim3(1), dim3(128), 0, st, ba->d_params);
    }
    SVO_HIP_CHECK(ctx, hipStreamEndCapture(st, &graph));
    SVO_HIP_CHECK(ctx, hipGraphInstantiate(&ba->graph_exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
  }
  // parameters of this solve -> device (pinned staging, one copy)
  BaDev* h_params = reinterpret_cast<BaDev*>(ba->h_pin + ba->cap_pay1 + 900);
  *h_params = d;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_params, h_params, sizeof(BaDev), hipMemcpyHostToDevice, st));
  (void)solve_lds;
  int enqueued = 0;
  bool done = false;
  while (!done && enqueued < ba->opt.max_iterations + 2) {
    const auto tl0 = std::chrono::steady_clock::now();
    SVO_HIP_CHECK(ctx, hipGraphLaunch(ba->graph_exec, st));
    enqueued += chunk;
    const auto tl1 = std::chrono::steady_clock::now();
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
    const auto tl2 = std::chrono::steady_clock::now();
    ba->t_launch += std::chrono::duration<double, std::milli>(tl1 - tl0).count();
    ba->t_sync += std::chrono::duration<double, std::milli>(tl2 - tl1).count();
    ba->n_chunks++;
    done = *reinterpret_cast<volatile int*>(done_host) != 0;
    if (!done && ba->opt.max_time_s > 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() >= ba->opt.max_time_s) break;
  }
  int cur = 0, iterations = 0, successful = 0, termination = 1;
  double initial_cost = 0, cost = 0;
  if (done) {
    iterations = done_host[1]; successful = done_host[2]; termination = done_host[3]; cur = done_host[4];
    initial_cost = h_fin[0]; cost = h_fin[1];
  } else {  // stopped by the wall clock (src/bundle_adjuster.cpp:11) or the safety cap: read the state back
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_init, ba->d_lm, sizeof(LmDev), hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
    iterations = h_init->iterations; successful = h_init->successful; termination = 1; cur = h_init->cur;
    initial_cost = h_init->initial_cost; cost = h_init->cost;
  }
  d.lm = nullptr;
  d.points = d.pts[cur]; d.cand_points = d.pts[1 - cur];
  d.poses = d.step[cur] + nn; d.cand_poses = d.step[1 - cur] + nn; d.dc = d.step[1 - cur];
  double* h_p = ba->h_pin;  // payload area is free in this mode
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_p, d.poses, sizeof(double) * 7 * K, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  ba->h_poses.assign(h_p, h_p + 7 * (size_t)K);
  ba->h_cand_poses = ba->h_poses;
  if (sum) {
    sum->iterations = iterations; sum->successful_steps = successful; sum->termination = termination;
    sum->initial_cost = initial_cost; sum->final_cost = cost;
    sum->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return SVO_OK;
}

// One launch per solve: the whole LM loop runs in ba_persistent_kernel.
static int ba_lm_persistent(svo_ba* ba, svo_ba_summary* sum) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  d.arrive2_target = 0; d.step_in = nullptr; d.flag1 = d.flag2 = nullptr;  // host-loop-only features
  hipStream_t st = ba->stream;
  const int n = d.n, K = d.K, nn = n > 0 ? n : 1;
  const auto t_begin = std::chrono::steady_clock::now();
  int* done_host = reinterpret_cast<int*>(ba->h_pin + ba->cap_pay1 + 16);
  double* h_fin = reinterpret_cast<double*>(done_host + 6);
  LmDev* h_init = reinterpret_cast<LmDev*>(ba->h_pin + ba->cap_pay1 + 32);
  memset(h_init, 0, sizeof(LmDev));
  h_init->radius = ba->opt.initial_radius; h_init->decrease_factor = 2.0;
  h_init->function_tol = ba->opt.function_tolerance; h_init->gradient_tol = ba->opt.gradient_tolerance;
  h_init->parameter_tol = ba->opt.parameter_tolerance; h_init->max_iterations = ba->opt.max_iterations;
  h_init->termination = 1;
  memset(done_host, 0, 6 * sizeof(int));
  h_fin[0] = h_fin[1] = 0.0;
  d.lm = ba->d_lm;
  d.pts[0] = d.points; d.pts[1] = d.cand_points;
  d.step[0] = ba->step_buf[0]; d.step[1] = ba->step_buf[1];
  d.done_host = done_host;
  d.pay1_out = d.pay1; d.pay2_out = d.pay2;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_lm, h_init, sizeof(LmDev), hipMemcpyHostToDevice, st));
  const int nd = (K - 1) * (K - 1) + (K - 1) + 1;
  int grid = std::max(nd, svo_div_up(d.C, 8));
  if (grid > 128) grid = 128;  // both the chunk loop and ... (reduce1 needs grid >= nd: K <= 11)
  if (grid < nd) { d.lm = nullptr; ctx->err = "ba: window too large for the persistent kernel"; return SVO_ERR_INVALID; }
  const size_t lds = ((size_t)n * n + 3 * (size_t)nn + 8) * sizeof(double);
  hipLaunchKernelGGL(ba_persistent_kernel, dim3(grid), dim3(512), lds, st, d, ba->opt.max_iterations + 2);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  const bool done = done_host[0] != 0;
  int cur = 0, iterations = 0, successful = 0, termination = 1;
  double initial_cost = 0, cost = 0;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_init, ba->d_lm, sizeof(LmDev), hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (h_init->abort) { d.lm = nullptr; ctx->err = "ba: persistent kernel grid barrier timed out"; return SVO_ERR_HIP; }
  if (done) {
    iterations = done_host[1]; successful = done_host[2]; termination = done_host[3]; cur = done_host[4];
    initial_cost = h_fin[0]; cost = h_fin[1];
  } else {
    iterations = h_init->iterations; successful = h_init->successful; termination = 1; cur = h_init->cur;
    initial_cost = h_init->initial_cost; cost = h_init->cost;
  }
  d.lm = nullptr;
  d.points = d.pts[cur]; d.cand_points = d.pts[1 - cur];
  d.poses = d.step[cur] + nn; d.cand_poses = d.step[1 - cur] + nn; d.dc = d.step[1 - cur];
  double* h_p = ba->h_pin;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_p, d.poses, sizeof(double) * 7 * K, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  ba->h_poses.assign(h_p, h_p + 7 * (size_t)K);
  ba->h_cand_poses = ba->h_poses;
  if (sum) {
    sum->iterations = iterations; sum->successful_steps = successful; sum->termination = termination;
    sum->initial_cost = initial_cost; sum->final_cost = cost;
    sum->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return SVO_OK;
}

// The LM loop (mirrors oracle/ora_ba.cpp step for step).
static int ba_lm(svo_ba* ba, svo_ba_summary* sum) {
  if (ba->d.det && !ba->allreduce && getenv("SVO_BA_PERSISTENT") && ba->d.K <= 11 && !(ba->opt.max_time_s > 0)) return ba_lm_persistent(ba, sum);
  // Measured on MI355X (bench workload, ~14 LM iterations per solve): host-driven loop 81 us/iteration,
  // device-resident loop with plain launches 97 us, hipGraph replay 81 us + idle tail iterations (7.6 us of dead
  // time per graph node).  The host-driven loop stays the default; SVO_BA_DEVICE_LM=1 selects the graph path.
  if (ba->d.det && !ba->allreduce && getenv("SVO_BA_DEVICE_LM")) return ba_lm_device(ba, sum);
  ba->d.lm = nullptr;
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
  const bool use_mfma = !d.det && ba->mfma_ok;
  const size_t mfma_lds = ba_mfma_lds_bytes(n, K - 1);
  const int mfma_grid = std::max(1, std::min(svo_div_up(d.C, MF_WAVES), 256));
  if (use_mfma) {
    SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ba_linearize_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mfma_lds));
  } else if (!d.det && lds_bytes > 64 * 1024) {
    SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ba_linearize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  std::vector<double> sc(n, 0.0), Sm((size_t)n * n), rhs(n), Df(n), Sfull((size_t)n * n);
  bool have_scale = false;
  double radius = ba->opt.initial_radius, decrease_factor = 2.0;
  double* cur_points = d.points;
  double* cand_points = d.cand_points;
  double* cur_poses = d.poses;
  double* cand_poses = d.cand_poses;

  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  // Single-rank deterministic solves poll completion words that the reduce kernels publish in pinned memory
  // (after a system-scope fence) instead of a stream wait per half-iteration; SVO_BA_NO_POLL=1 restores the waits.
  const bool poll = d.det && !ba->allreduce && !getenv("SVO_BA_NO_POLL");
  // SVO_BA_FUSE=1: back-substitution + reduce2 in one launch (the last workgroup reduces).  Measured on the bench
  // workload: +1.5 % frames/s at 8 streams, -4 % for a single stream (27 vs 22 us per half-iteration: the agent-scope
  // fences and the 64-thread tail cost more than the saved launch), so the two-launch form stays the default.
  const bool fuse2 = d.det && d.C > 0 && getenv("SVO_BA_FUSE") != nullptr;
  int* h_flag1 = reinterpret_cast<int*>(h_pay2 + 6);
  int* h_flag2 = reinterpret_cast<int*>(h_pay2 + 7);
  d.flag1 = poll ? h_flag1 : nullptr; d.flag2 = poll ? h_flag2 : nullptr; d.arrive = ba->d_arrive;
  int rc_poll = 0;
  auto wait_flag = [&](int* flag, int seq) -> int {
    const auto t0 = now();
    unsigned spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
      __builtin_ia32_pause();
      if (++spins > 4096u && (spins & 63u) == 0) sched_yield();  // long wait: stay polite when threads outnumber cores
      if ((spins & 0xFFFFu) == 0 && ms(t0, now()) > 10000.0) {  // never expected: fall back to the stream wait
        SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { ctx->err = "ba: completion word never arrived"; return SVO_ERR_HIP; }
      }
    }
    return SVO_OK;
  };
  auto linearize = [&](double rad) -> int {
    const auto tp0 = now();
    d.points = cur_points; d.cand_points = cand_points; d.poses = cur_poses; d.cand_poses = cand_poses;
    if (!d.det) SVO_HIP_CHECK(ctx, hipMemsetAsync(d.pay1, 0, sizeof(double) * pay1, st));
    if (d.C > 0) {
      SvoProfScope prof(ctx, SVO_PROF_BA_LINEARIZE, st);
      if (d.det) hipLaunchKernelGGL(ba_linearize_kernel, dim3(d.C), dim3(64), 64, st, d, rad, have_scale ? 0 : 1);  // one wave per workgroup: spreads the chunks over the CUs
      else if (use_mfma) hipLaunchKernelGGL(ba_linearize_mfma_kernel, dim3(mfma_grid), dim3(64 * MF_WAVES), mfma_lds, st, d, rad, have_scale ? 0 : 1);
      else hipLaunchKernelGGL(ba_linearize_kernel, dim3(grid), dim3(256), lds_bytes, st, d, rad, have_scale ? 0 : 1);
    }
    // single rank + deterministic mode: the reduce kernel writes the payload straight into pinned host
    // memory (no copy kernel); a sharded run keeps it on the device for the all-reduce
    const bool zero_copy = d.det && !ba->allreduce;
    d.pay1_out = zero_copy ? h_pay1 : d.pay1;
    d.pay2_out = zero_copy ? h_pay2 : d.pay2;
    const int nred = (K - 1) * (K - 1) + (K - 1) + 1;
    if (poll) { ba->arrive_total += (unsigned)nred; d.arrive_target = ba->arrive_total; d.seq = ++ba->seq; }
    if (d.det) hipLaunchKernelGGL(ba_reduce1_kernel, dim3(nred), dim3(1024), 0, st, d);
    SVO_HIP_CHECK(ctx, hipGetLastError());
    if (poll) { rc_poll = wait_flag(h_flag1, d.seq); if (rc_poll) return rc_poll; }
    if (ba->allreduce) {
      SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
      if (ba->allreduce(d.pay1, pay1, ba->allreduce_user)) { ctx->err = "ba: allreduce callback failed"; return SVO_ERR_INVALID; }
    }
    if (!zero_copy) SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_pay1, d.pay1, sizeof(double) * pay1, hipMemcpyDeviceToHost, st));
    if (!poll) SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
    // mirror the upper pair blocks (kernel writes each unordered pose pair once)
    const double* S = h_pay1;
    for (int a = 0; a < K - 1; ++a)
      for (int b = 0; b < K - 1; ++b)
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) {
            const size_t ij = (size_t)(6 * a + i) * n + 6 * b + j, ji = (size_t)(6 * b + j) * n + 6 * a + i;
            Sfull[ij] = (a == b || d.det) ? S[ij] : S[ij] + S[ji];
          }
    ba->t_lin += ms(tp0, now()); ba->n_lin++;
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
      const auto th0 = now();
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
        const auto tb0 = now();
        ba->t_host += ms(th0, tb0);
        // one H2D: [dc (n) | candidate poses (7K)] are adjacent both in the pinned buffer and on the device
        d.poses = cur_poses; d.cand_poses = cand_poses; d.dc = cand_poses - (n > 0 ? n : 1);
        d.step_in = h_dc;  // zero-copy: the kernel reads the 6(K-1)+7K doubles straight from the pinned buffer
        if (!d.det) SVO_HIP_CHECK(ctx, hipMemsetAsync(d.pay2, 0, sizeof(double) * 4, st));
        d.points = cur_points; d.cand_points = cand_points;
        if (d.C > 0) {
          SvoProfScope prof(ctx, SVO_PROF_BA_BACKSUB, st);
          if (d.det) {
            d.arrive2 = ba->d_arrive + 1;
            d.arrive2_target = 0;
            if (fuse2) { if (poll) d.seq = ++ba->seq; ba->arrive2_total += (unsigned)d.C; d.arrive2_target = ba->arrive2_total; }
            hipLaunchKernelGGL(ba_backsub_kernel, dim3(d.C), dim3(64), 0, st, d, radius);
          }
          else hipLaunchKernelGGL(ba_backsub_kernel, dim3(grid), dim3(256), 0, st, d, radius);
        }
        if (poll && !fuse2) d.seq = ++ba->seq;
        if (d.det && !fuse2) hipLaunchKernelGGL(ba_reduce2_kernel, dim3(1), dim3(128), 0, st, d);
        SVO_HIP_CHECK(ctx, hipGetLastError());
        if (poll) { rc = wait_flag(h_flag2, d.seq); if (rc) return rc; }
        if (ba->allreduce) {
          SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
          if (ba->allreduce(d.pay2, 4, ba->allreduce_user)) { ctx->err = "ba: allreduce callback failed"; return SVO_ERR_INVALID; }
        }
        if (!(d.det && !ba->allreduce)) SVO_HIP_CHECK(ctx, hipMemcpyAsync(h_pay2, d.pay2, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
        if (!poll) SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
        ba->t_back += ms(tb0, now()); ba->n_back++;
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
  if (poll) SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));  // nothing is pending; keeps later users of the stream ordered
  d.flag1 = d.flag2 = nullptr;
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
  svo_use_device(ba->ctx);
  SVO_REQUIRE(ba->ctx, poses7 && (n_points == 0 || points3) && (n_obs == 0 || (obs_pose && obs_point && obs_uv)),
              "ba_load_problem: null buffer");
  return ba_upload(ba, n_poses, poses7, n_points, points3, n_obs, obs_pose, obs_point, obs_uv);
}

extern "C" int svo_ba_solve_problem(svo_ba* ba, svo_ba_summary* summary) {
  if (!ba) return SVO_ERR_INVALID;
  svo_use_device(ba->ctx);
  SVO_REQUIRE(ba->ctx, ba->d.K >= 1, "ba_solve_problem: no problem loaded");
  const int rc = ba_lm(ba, summary);
  if (!rc) ba->upload_pending = false;  // every LM path ends with the stream drained
  return rc;
}

extern "C" int svo_ba_read_problem(svo_ba* ba, double* poses7, double* points3) {
  if (!ba) return SVO_ERR_INVALID;
  svo_ctx* ctx = ba->ctx;
  if (poses7) memcpy(poses7, ba->h_poses.data(), sizeof(double) * 7 * (size_t)ba->d.K);
  if (points3 && ba->n_points) {
    // through the pinned arena (idle once the solve has finished): the runtime's pageable path would stage and wait
    const size_t bytes = sizeof(double) * 3 * (size_t)ba->n_points;
    void* stage = bytes <= ba->arena_cap ? (void*)ba->h_arena : (void*)points3;
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(stage, ba->d.points, bytes, hipMemcpyDeviceToHost, ba->stream));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
    ba->upload_pending = false;
    if (stage != (void*)points3) memcpy(points3, stage, bytes);
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

extern "C" int svo_ba_reset(svo_ba* ba) {
  if (!ba) return SVO_ERR_INVALID;
  ba->window.clear();
  ba->feat_pos.clear();
  ba->new_frame_added = false;
  ba->d.K = 0;
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
  svo_use_device(ba->ctx);
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
  const auto tu0 = std::chrono::steady_clock::now();
  int rc = ba_upload(ba, K, poses.data(), (int)lm_ids.size(), points.data(), (int)op.size(), op.data(), oj.data(), uv.data());
  if (rc) return rc;
  ba->t_upload += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count();
  rc = ba_lm(ba, summary);
  if (!rc) ba->upload_pending = false;
  ba->t_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count();
  ba->n_solves++;
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
