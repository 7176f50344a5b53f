// a5 — cv::solvePnPRansac(obj, img, K, 0, rvec, tvec, true, 100, 8.0, 0.99, inliers), reference call site
// src/image_processor.cpp:76-80.  OpenCV's sample sequence / EPnP are version specific;
// the deterministic RANSAC implemented here is the one DEFINED in oracle/ora_pnp.cpp and DESIGN.md §PnP:
//   pnp_ransac_kernel / pnp_group_kernel : ONE launch per solvePnPRansac call.
//       hypotheses  — one wavefront per hypothesis, four per workgroup (all `iterations` hypotheses in the launch): splitmix64
//                     sampling of 5 distinct points, LM minimal solve from the extrinsic guess with a trig-free quaternion
//                     retraction, then every lane tests its share of the n points and the inlier set is emitted as a 64-bit
//                     ballot mask per 64 points.  The 5-term normal-equation sums are evaluated entry-per-lane in the
//                     oracle's k = 0..4 order => bit-identical to the CPU.
//       bookkeeping — by the last workgroup to arrive: the counts consumed in order h = 0,1,.. with OpenCV's adaptive
//                     iteration cap (RANSACUpdateNumIters, declared arithmetic: host/pnp_iters.h).
//       refinement  — by the same workgroup: LM over the best model's inliers; reductions use the declared order "per-thread
//                     strided partials (stride 256), then binary tree" so the refined pose is bit-identical to the oracle.
//   (Rounds 1-3: two launches with the bookkeeping on the host between them.)
#include <math.h>

#include <algorithm>
#include <cfloat>

#include "det_trig.h"
#include "kernels.h"
#include "group_kernels.h"
#include "pnp_iters.h"
#include "tail_device.h"

namespace {
constexpr int MODEL = 5;
constexpr unsigned long long SEED = 0x5EED0A5ull, STRIDE = 0xD1B54A32D192ED03ull;

struct PnpPose { double q[4]; double t[3]; };

__host__ __device__ inline unsigned long long splitmix(unsigned long long& s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

// Where a tracked feature's world point comes from (get_world_points, src/bundle_adjuster.cpp:159-163, called at
// src/image_processor.cpp:72):
//   XyzArray : a float3 array in feature order (the single pipeline: gathered on the host, uploaded);
//   XyzStore : the pipeline group's DEVICE-RESIDENT landmark store, keyed by feature id — entry (id mod capacity) =
//              {x, y, z, low 32 bits of the id}, written by the solve that last optimised the landmark (ba_lm_kernel's delivery,
//              or the scatter behind a host-driven solve).  An entry under another id is reported (`bad`), never used silently.
struct XyzArray {
  const float* xyz;
  __device__ __forceinline__ void get(int i, float& X, float& Y, float& Z) const { X = xyz[3 * i]; Y = xyz[3 * i + 1]; Z = xyz[3 * i + 2]; }
};
struct XyzStore {
  const long long* ids; const float4* store; unsigned mask; int* bad;
  __device__ __forceinline__ void get(int i, float& X, float& Y, float& Z) const {
    const unsigned key = (unsigned)ids[i];
    const float4 e = store[key & mask];
    if (__float_as_uint(e.w) != key) svo_host_store(bad, 1);
    X = e.x; Y = e.y; Z = e.z;
  }
};
// LDS traffic of ONE wavefront is ordered; the fence only keeps the compiler from moving accesses across it
__device__ __forceinline__ void pnp_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <bool WAVE>
__device__ __forceinline__ void pnp_sync() { if (WAVE) pnp_wave_fence(); else __syncthreads(); }

// residual (and 2x6 Jacobian) of one point; returns ex^2+ey^2
template <typename Xyz>
__device__ __forceinline__ double pnp_term(const double* R, const double* t, const Xyz& src, const float* xy, int i,
                                           double f, double cx, double cy, double* e, double* J /*12 or null*/) {
  float Xf, Yf, Zf;
  src.get(i, Xf, Yf, Zf);
  const double X = Xf, Y = Yf, Z = Zf;
  const double rx = R[0] * X + R[1] * Y + R[2] * Z;
  const double ry = R[3] * X + R[4] * Y + R[5] * Z;
  const double rz = R[6] * X + R[7] * Y + R[8] * Z;
  const double px = rx + t[0], py = ry + t[1], pz = rz + t[2];
  const double iz = 1.0 / pz;
  const double ex = f * px * iz + cx - (double)xy[2 * i];
  const double ey = f * py * iz + cy - (double)xy[2 * i + 1];
  e[0] = ex; e[1] = ey;
  if (J) {
    const double a = f * iz, bx = -f * px * iz * iz, by = -f * py * iz * iz;
    J[0] = bx * ry;           J[1] = a * rz - bx * rx; J[2] = -a * ry;
    J[6] = -a * rz + by * ry; J[7] = -by * rx;         J[8] = a * rx;
    J[3] = a; J[4] = 0; J[5] = bx;
    J[9] = 0; J[10] = a; J[11] = by;
  }
  return ex * ex + ey * ey;
}

// One damped Gauss-Newton step and its retraction, as declared in oracle/ora_pnp.cpp (solve6: lower Cholesky of
// H + lambda diag(H) + 1e-12 I, left-looking, products subtracted one at a time in ascending k, forward then backward
// substitution; retract: q+ = normalise(normalise([1, d/2]) (x) q), t+ = t + d[3:6]) — by ONE WAVEFRONT instead of one lane:
// lane i < 6 owns row i of the factor, columns are finished left to right, every element receives exactly the oracle's
// operations in the oracle's order (one square root per column, one division per element) — same bits —
// but the 15 divisions of the column scalings and the 8 of the two quaternion normalisations run side by side: 28 dependent
// f64 square roots / divisions per LM step instead of 45, and those sequences ARE the step (an f64 division is ~12 dependent
// instructions).  Called by all 64 lanes of the first wavefront; values cross lanes through v_readlane (constant lanes).
__device__ __forceinline__ double bcast_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__device__ bool solve6_retract_wave(const double* Hin, const double* g, double lambda, const PnpPose& P, double* d_out, PnpPose& N) {
  const int lane = threadIdx.x & 63;
  const int ii = lane < 6 ? lane : 5;  // idle lanes shadow row 5 (their results are never used)
  double L[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    L[k] = k <= ii ? Hin[6 * k + ii] : 0.0;
    if (k == ii) L[k] += lambda * Hin[6 * k + k] + 1e-12;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double v = L[j];
#pragma unroll
    for (int k = 0; k < j; ++k) v -= L[k] * bcast_d(L[k], j);
    const double s = bcast_d(v, j);
    if (!(s > 0)) return false;  // wave-uniform
    const double ljj = sqrt(s);
    L[j] = ii == j ? ljj : v / ljj;
  }
  const double mg = -g[ii];
  double y = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = mg;
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[k] * bcast_d(y, k);
    const double yi = v / L[i];
    if (ii == i) y = yi;
  }
  double dd = 0.0;
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = y;
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v -= bcast_d(L[i], k) * bcast_d(dd, k);
    const double di = v / L[i];
    if (ii == i) dd = di;
  }
  double d[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = bcast_d(dd, k);
  if (lane < 6) d_out[lane] = dd;
  // retract(P, d, N): every lane forms the sums, lane k < 4 does the k-th division of each normalisation
  const double dq0 = 1.0, dq1 = 0.5 * d[0], dq2 = 0.5 * d[1], dq3 = 0.5 * d[2];
  const double nn = sqrt(dq0 * dq0 + dq1 * dq1 + dq2 * dq2 + dq3 * dq3);
  const int kk = lane & 3;
  const double mine = (kk == 0 ? dq0 : kk == 1 ? dq1 : kk == 2 ? dq2 : dq3) / nn;
  const double n0 = bcast_d(mine, 0), n1 = bcast_d(mine, 1), n2_ = bcast_d(mine, 2), n3 = bcast_d(mine, 3);
  const double* q = P.q;
  const double r0 = n0 * q[0] - n1 * q[1] - n2_ * q[2] - n3 * q[3];
  const double r1 = n0 * q[1] + n1 * q[0] + n2_ * q[3] - n3 * q[2];
  const double r2 = n0 * q[2] - n1 * q[3] + n2_ * q[0] + n3 * q[1];
  const double r3 = n0 * q[3] + n1 * q[2] - n2_ * q[1] + n3 * q[0];
  const double nr = sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
  const double rq = (kk == 0 ? r0 : kk == 1 ? r1 : kk == 2 ? r2 : r3) / nr;
  if (lane < 4) N.q[lane] = rq;
  if (lane < 3) N.t[lane] = P.t[lane] + d[3 + lane];
  return true;
}

// Shared LM driver.  `acc(pose, H, g)` evaluates the cost — and the normal equations into H / g (LDS) — at `pose` for the
// whole workgroup and returns the cost (uniform).  The candidate is evaluated WITH its normal equations right away: an
// accepted step (the common case) then needs no second sweep over the points; the numbers are the ones a re-evaluation
// at the accepted pose would give, so the iterates are unchanged.
struct LmShared {
  PnpPose cur, cand;
  double H[36], g[6], Hc[36], gc[6], d[6];
  double cost;
  int ok;
};

// WAVE: the owner set is ONE wavefront (a hypothesis), hand-overs are wave-level fences; otherwise the whole workgroup.
template <bool WAVE, typename Acc>
__device__ void lm_solve(LmShared& S, int max_it, Acc acc) {
  const int t = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  double lambda = 1e-3;
  double cost = acc(S.cur, S.H, S.g);
  for (int it = 0; it < max_it; ++it) {
    if (t < 64) {
      const bool ok = solve6_retract_wave(S.H, S.g, lambda, S.cur, S.d, S.cand);
      if (t == 0) S.ok = ok ? 1 : 0;
    }
    pnp_sync<WAVE>();
    const int ok = S.ok;
    if (!ok) { lambda *= 10; pnp_sync<WAVE>(); continue; }
    const double c2 = acc(S.cand, S.Hc, S.gc);
    if (c2 < cost) {
      const double* d = S.d;
      const double step2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
      // CvLevMarq's stop: relative parameter change below FLT_EPSILON, against the pose before the step (oracle/ora_pnp.cpp)
      const double x2 = ((S.cur.t[0] * S.cur.t[0] + S.cur.t[1] * S.cur.t[1]) + S.cur.t[2] * S.cur.t[2]) +
                        4.0 * ((S.cur.q[1] * S.cur.q[1] + S.cur.q[2] * S.cur.q[2]) + S.cur.q[3] * S.cur.q[3]);
      pnp_sync<WAVE>();
      if (t == 0) S.cur = S.cand;
      if (t < 36) S.H[t] = S.Hc[t];
      if (t < 6) S.g[t] = S.gc[t];
      pnp_sync<WAVE>();
      lambda *= 0.1;
      if (lambda < 1e-9) lambda = 1e-9;
      cost = c2;
      if (step2 < 1e-20 || step2 <= 1.4210854715202004e-14 * x2) break;  // FLT_EPSILON^2 = 2^-46
    } else {
      lambda *= 10;
      if (lambda > 1e6) break;
    }
  }
}
}  // namespace

// LDS of one hypothesis (one wavefront)
struct HypLds {
  LmShared S;
  double sJ[MODEL][12], sE[MODEL][2];
  int sIdx[MODEL];
};

// Hypothesis `h` by the calling WAVEFRONT (64 lanes; any workgroup shape).  WT: count / mask / pose are consumed by another
// workgroup of this launch (the bookkeeping of pnp_group_kernel) and are written through.
template <typename Xyz, bool WT>
__device__ __forceinline__ void pnp_hypothesis_wave(const Xyz& src, const float* __restrict__ xy, int n, double f, double cx, double cy,
                                                    const PnpPose& P0, double thr2, int h, HypLds& L, double* __restrict__ hyp_pose,
                                                    int* __restrict__ hyp_count, unsigned long long* __restrict__ hyp_mask, int mask_words,
                                                    int* __restrict__ host_count) {
  const int lane = threadIdx.x & 63;
  LmShared& S = L.S;
  if (lane == 0) {
    unsigned long long s = SEED + (unsigned long long)h * STRIDE;
    for (int k = 0; k < MODEL; ++k) {
      for (;;) {
        const int c = (int)(splitmix(s) % (unsigned long long)n);
        bool dup = false;
        for (int j = 0; j < k; ++j) dup |= L.sIdx[j] == c;
        if (!dup) { L.sIdx[k] = c; break; }
      }
    }
    S.cur = P0;
  }
  pnp_wave_fence();
  auto acc = [&](const PnpPose& P, double* H, double* g) -> double {
    double R[9];
    quat_to_R(P.q, R);
    if (lane < MODEL) pnp_term(R, P.t, src, xy, L.sIdx[lane], f, cx, cy, L.sE[lane], L.sJ[lane]);
    pnp_wave_fence();
    if (lane < 27) {
      if (lane < 21) {  // upper-triangular entry (r,c)
        int r = 0, e = lane;
        while (e >= 6 - r) { e -= 6 - r; ++r; }
        const int c = r + e;
        double s = 0.0;
        for (int k = 0; k < MODEL; ++k) s += L.sJ[k][r] * L.sJ[k][c] + L.sJ[k][6 + r] * L.sJ[k][6 + c];
        H[6 * r + c] = s;
      } else {
        const int r = lane - 21;
        double s = 0.0;
        for (int k = 0; k < MODEL; ++k) s += L.sJ[k][r] * L.sE[k][0] + L.sJ[k][6 + r] * L.sE[k][1];
        g[r] = s;
      }
    }
    double cost = 0.0;
    for (int k = 0; k < MODEL; ++k) cost += L.sE[k][0] * L.sE[k][0] + L.sE[k][1] * L.sE[k][1];
    pnp_wave_fence();
    return cost;
  };
  lm_solve<true>(S, 12, acc);
  pnp_wave_fence();
  const PnpPose P = S.cur;
  double R[9];
  quat_to_R(P.q, R);
  int cnt = 0;
  for (int w = 0; w < mask_words; ++w) {
    const int i = w * 64 + lane;
    bool in = false;
    if (i < n) {
      float Xf, Yf, Zf;
      src.get(i, Xf, Yf, Zf);
      const double X = Xf, Y = Yf, Z = Zf;
      const double px = R[0] * X + R[1] * Y + R[2] * Z + P.t[0];
      const double py = R[3] * X + R[4] * Y + R[5] * Z + P.t[1];
      const double pz = R[6] * X + R[7] * Y + R[8] * Z + P.t[2];
      if (pz > 0) {
        const double iz = 1.0 / pz;
        const double ex = f * px * iz + cx - (double)xy[2 * i];
        const double ey = f * py * iz + cy - (double)xy[2 * i + 1];
        in = ex * ex + ey * ey <= thr2;
      }
    }
    const unsigned long long m = __ballot(in);
    cnt += __popcll(m);
    if (lane == 0) {
      if (WT) svo_wt_store(&hyp_mask[(size_t)h * mask_words + w], m);
      else hyp_mask[(size_t)h * mask_words + w] = m;
    }
  }
  if (lane == 0) {
    if (WT) svo_wt_store(&hyp_count[h], cnt); else hyp_count[h] = cnt;
    if (host_count) svo_host_store(&host_count[h], cnt);  // pinned: the host's RANSAC bookkeeping reads the counts in place
    for (int k = 0; k < 7; ++k) {
      const double v = k < 4 ? P.q[k] : P.t[k - 4];
      if (WT) svo_wt_store(&hyp_pose[7 * h + k], v); else hyp_pose[7 * h + k] = v;
    }
  }
}

// Refinement over the inliers of hypothesis `best` by the calling workgroup (256 threads).  COHERENT: mask and pose of the
// hypothesis were written by other workgroups of this very launch (read at the coherence point).
// What the stereo + triangulation launch queued behind this one needs (SvoChainRec): rvec / tvec exactly as the host stores them
// (float, from the refined pose by the declared conversion; the previous ones when no model was found), then hmat and M by the
// function the host calls for lanes without PnP (host/chain_math.h).  One thread.
__device__ __forceinline__ void pnp_write_chain(const SvoPnpLane& a, int best, int n_inl, const double* q, const double* t) {
  float rvec[3], tvec[3];
  if (best >= 0) {
    double rv[3];
    svo_det_rvec_from_quat(q, rv);
    for (int k = 0; k < 3; ++k) { rvec[k] = (float)rv[k]; tvec[k] = (float)t[k]; }
  } else {
    for (int k = 0; k < 3; ++k) { rvec[k] = a.prev_rvec[k]; tvec[k] = a.prev_tvec[k]; }
  }
  SvoChainRec* c = a.chain;
  svo_chain_matrix(rvec, tvec, a.cam_f, a.cam_cx, a.cam_cy, a.cam_b, c->M);
  c->n_inl = best >= 0 ? n_inl : 0;
  c->best = best;
}

template <typename Xyz, bool COHERENT>
__device__ __forceinline__ void pnp_refine_body(const Xyz& src, const float* __restrict__ xy, int n,
                                                double f, double cx, double cy, const double* hyp_pose,
                                                const unsigned long long* hyp_mask, int mask_words,
                                                int best, double* __restrict__ out_pose, int* __restrict__ inliers,
                                                int* __restrict__ n_inliers, double* __restrict__ host_pose,
                                                int* __restrict__ host_inliers, int* __restrict__ host_nin,
                                                float* __restrict__ inlier_xy, SvoPublish pub, LmShared& S, double (*sPart)[28], int* sBase,
                                                const SvoPnpLane* chain_lane = nullptr) {
  svo_latency_critical();
  const int tid = threadIdx.x;
  // inlier list of the best hypothesis, ascending (one wave builds it)
  if (tid < 64) {
    int base = 0;
    for (int w = 0; w < mask_words; ++w) {
      const unsigned long long m = COHERENT ? svo_coherent_load(&hyp_mask[(size_t)best * mask_words + w]) : hyp_mask[(size_t)best * mask_words + w];
      if ((m >> tid) & 1ull) {
        const int slot = base + __popcll(m & ((1ull << tid) - 1ull));
        inliers[slot] = w * 64 + tid;
        if (host_inliers) host_inliers[slot] = w * 64 + tid;
        if (inlier_xy) { inlier_xy[2 * slot] = xy[2 * (w * 64 + tid)]; inlier_xy[2 * slot + 1] = xy[2 * (w * 64 + tid) + 1]; }  // the dedup stage's input, gathered here
      }
      base += __popcll(m);
    }
    if (tid == 0) {
      *sBase = base;
      for (int k = 0; k < 4; ++k) S.cur.q[k] = COHERENT ? svo_coherent_load(&hyp_pose[7 * best + k]) : hyp_pose[7 * best + k];
      for (int k = 0; k < 3; ++k) S.cur.t[k] = COHERENT ? svo_coherent_load(&hyp_pose[7 * best + 4 + k]) : hyp_pose[7 * best + 4 + k];
    }
  }
  __syncthreads();
  const int m = *sBase;
  auto acc = [&](const PnpPose& P, double* H, double* g) -> double {
    double R[9];
    quat_to_R(P.q, R);
    double v[28];
#pragma unroll
    for (int e = 0; e < 28; ++e) v[e] = 0.0;
    for (int k = tid; k < m; k += 256) {
      double e2[2], J[12];
      const int pi = __hip_atomic_load(&inliers[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // written by wave 0 above
      v[27] += pnp_term(R, P.t, src, xy, pi, f, cx, cy, e2, J);
      int o = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        v[21 + r] += J[r] * e2[0] + J[6 + r] * e2[1];
#pragma unroll
        for (int c = r; c < 6; ++c) { v[o] += J[r] * J[c] + J[6 + r] * J[6 + c]; ++o; }
      }
    }
    // the declared binary tree (partial t += partial t + s for s = 128, 64, .., 1).  The two levels that cross wavefronts go
    // through LDS (128 rows, then 64 of them again), the six inside the first wavefront through lane shifts: t + s is lane
    // t + s of the same wavefront for s <= 32 — the same additions, partial t on the left — and the workgroup needs 28 KB of
    // LDS instead of 57 (a CU that holds three resident solves has 13 KB left: the 57 KB form waited for one of them to end).
    if (tid >= 128) {
#pragma unroll
      for (int e = 0; e < 28; ++e) sPart[tid - 128][e] = v[e];
    }
    __syncthreads();
    if (tid < 128) {
#pragma unroll
      for (int e = 0; e < 28; ++e) v[e] += sPart[tid][e];
    }
    __syncthreads();
    if (tid >= 64 && tid < 128) {
#pragma unroll
      for (int e = 0; e < 28; ++e) sPart[tid - 64][e] = v[e];
    }
    __syncthreads();
    if (tid < 64) {
#pragma unroll
      for (int e = 0; e < 28; ++e) v[e] += sPart[tid][e];
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
        for (int e = 0; e < 28; ++e) v[e] += __shfl_down(v[e], s);  // lanes >= s compute values nobody reads
      }
    }
    __syncthreads();  // every thread is done reading sPart
    if (tid == 0) {
      int o = 0;
      for (int r = 0; r < 6; ++r) {
        g[r] = v[21 + r];
        for (int c = r; c < 6; ++c) H[6 * r + c] = v[o++];
      }
      sPart[0][27] = v[27];
    }
    __syncthreads();
    const double cost = sPart[0][27];
    __syncthreads();
    return cost;
  };
  lm_solve<false>(S, 20, acc);
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 4; ++k) out_pose[k] = S.cur.q[k];
    for (int k = 0; k < 3; ++k) out_pose[4 + k] = S.cur.t[k];
    *n_inliers = m;
    if (host_pose) {
      for (int k = 0; k < 4; ++k) host_pose[k] = S.cur.q[k];
      for (int k = 0; k < 3; ++k) host_pose[4 + k] = S.cur.t[k];
      *host_nin = m;
    }
    if (chain_lane && chain_lane->chain) pnp_write_chain(*chain_lane, best, m, S.cur.q, S.cur.t);
  }
  svo_publish_block(pub);
}

// ---- the WHOLE solvePnPRansac of a stream as one launch.  Every workgroup runs four hypotheses (one per wavefront, wave-level
// hand-overs); the LAST workgroup of the stream to arrive does OpenCV's bookkeeping (hypotheses consumed in order, strictly
// more inliers replaces the best, RANSACUpdateNumIters with the declared arithmetic of host/pnp_iters.h) and refines the
// winner — the operations of oracle/ora_pnp.cpp in its order, hence the same bits (tests/test_pnp.py, tests/test_group.py).
// One launch and one completion word per keyframe where round 3 had a world-point upload kernel (group), two launches and a
// host round trip between them.  Src: where the world points come from (XyzArray / XyzStore above).
template <typename Src>
__device__ __forceinline__ void pnp_fused_body(const SvoPnpLane& a, const Src& src) {
  svo_latency_critical();
  __shared__ double sPart[128][28];  // first the four hypotheses' LDS, then the refinement's partial sums (the tree's two cross-wavefront levels)
  __shared__ LmShared S;
  __shared__ int sBase, sLast, sBest;
  static_assert(4 * sizeof(HypLds) <= sizeof(double) * 128 * 28, "the hypotheses' LDS must fit the refinement's");
  const int tid = threadIdx.x, wave = tid >> 6;
  const int h = blockIdx.x * 4 + wave;
  if ((int)blockIdx.x * 4 >= a.launched) return;  // (a launch carries lanes with different hypothesis counts: not this lane's workgroup)
  if (h < a.launched) {
    PnpPose P0;
    for (int k = 0; k < 4; ++k) P0.q[k] = a.q0[k];
    for (int k = 0; k < 3; ++k) P0.t[k] = a.t0[k];
    HypLds& L = reinterpret_cast<HypLds*>(&sPart[0][0])[wave];
    pnp_hypothesis_wave<Src, true>(src, a.xy, a.n, a.f, a.cx, a.cy, P0, a.thr2, h, L, a.hyp_pose, a.hyp_count, a.hyp_mask, a.mask_words, nullptr);
  }
  if (!svo_last_arrival(a.arrive, a.arrive_target, &sLast)) return;
  if (tid == 0) {
    // OpenCV consumes the hypotheses in order and stops at the adaptive cap: with mostly inliers the cap falls to a handful after
    // the first good model, so a lane's FIRST launch computes only the first few (svo_kg_pnp_first) and the rest — never looked
    // at in that case — only when the cap stays above what was computed (host_best = -2: launch again with all of them)
    int best = -1, best_cnt = 0, niters = a.iterations;
    for (int hh = 0; hh < niters && hh < a.launched; ++hh) {
      const int c = svo_coherent_load(&a.hyp_count[hh]);
      if (c > (best_cnt > MODEL - 1 ? best_cnt : MODEL - 1)) {
        best = hh; best_cnt = c;
        niters = svo_pnp_update_num_iters_det(a.confidence, (double)(a.n - best_cnt) / a.n, MODEL, niters);
      }
    }
    if (niters > a.launched) best = -2;  // hypotheses launched .. niters-1 are still to be consumed
    sBest = best;
    *a.host_best = best;
    if (best < 0) *a.host_nin = 0;
  }
  __syncthreads();
  const int best = sBest;
  SvoPublish pub;
  pub.word = a.word; pub.seq = a.seq;
  if (best < 0) {
    if (tid == 0 && a.chain) pnp_write_chain(a, best, 0, a.q0, a.t0);
    svo_publish_block(pub);
    return;
  }
  pnp_refine_body<Src, true>(src, a.xy, a.n, a.f, a.cx, a.cy, a.hyp_pose, a.hyp_mask, a.mask_words, best, a.out_pose, a.inliers, a.n_inliers,
                             a.host_pose, a.host_inliers, a.host_nin, a.inlier_xy, pub, S, sPart, &sBase, &a);
}

// single stream: the world points as a float3 array in feature order (gathered by the host: get_world_points)
__global__ __launch_bounds__(256) void pnp_ransac_kernel(SvoPnpLane a, const float* __restrict__ xyz) { pnp_fused_body(a, XyzArray{xyz}); }

// stream-batched form (group_kernels.h): blockIdx.y = lane, world points from the lane's device-resident landmark store
__global__ __launch_bounds__(256) void pnp_group_kernel(SvoPnpLanes g) {
  const SvoPnpLane& a = g.lane[blockIdx.y];
  pnp_fused_body(a, XyzStore{a.ids, a.store, a.store_mask, a.host_bad});
}

// ----------------------------------------------------------------------------- host side
static int update_num_iters(double p, double ep, int model_points, int max_iters) {
  return svo_pnp_update_num_iters_det(p, ep, model_points, max_iters);  // OpenCV's RANSACUpdateNumIters, declared arithmetic (host/pnp_iters.h)
}

// Device-pointer form used by the pipeline: xyz/xy on the device, n known on the host.
// Work buffers come from `s`.  On return rvec3/tvec3 are updated (host), d_inliers holds the list.
int svo_k_pnp(svo_ctx* ctx, SvoScratch& s, const float* d_xyz, const float* d_xy, int n, float focal, float cxf, float cyf,
              double* rvec3, double* tvec3, int iterations, float reproj_err, double confidence, int* d_inliers,
              int* n_inliers, int* h_inliers, float* d_inlier_xy) {
  *n_inliers = 0;
  if (n < MODEL || iterations < 1) return SVO_OK;
  PnpPose P0;
  {
    svo_det_quat_from_rvec(rvec3, P0.q);  // declared arithmetic (host/det_trig.h)
    for (int k = 0; k < 3; ++k) P0.t[k] = tvec3[k];
  }
  const int words = svo_div_up(n, 64);
  double* d_pose = s.take<double>(7 * (size_t)iterations);
  int* d_count = s.take<int>(iterations);
  unsigned long long* d_mask = s.take<unsigned long long>((size_t)iterations * words);
  double* d_out = s.take<double>(7);
  int* d_nin = s.take<int>(1);
  if (!d_pose || !d_count || !d_mask || !d_out || !d_nin) { ctx->err = "pnp: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  // ONE launch (pnp_ransac_kernel); what the host needs comes back in pinned memory behind one completion word: no D2H
  // blits, no stream waits
  double* h_out = (double*)((char*)ctx->h_pinned + 4096);
  int* h_nin = (int*)((char*)ctx->h_pinned + 4096 + 64);
  int* h_best = (int*)((char*)ctx->h_pinned + 4096 + 128);
  SvoPnpLane a;
  memset(&a, 0, sizeof(a));
  a.xy = d_xy; a.n = n; a.f = (double)focal; a.cx = (double)cxf; a.cy = (double)cyf;
  for (int k = 0; k < 4; ++k) a.q0[k] = P0.q[k];
  for (int k = 0; k < 3; ++k) a.t0[k] = P0.t[k];
  a.thr2 = (double)reproj_err * (double)reproj_err; a.confidence = confidence; a.iterations = iterations;
  a.hyp_pose = d_pose; a.hyp_count = d_count; a.hyp_mask = d_mask; a.mask_words = words;
  a.out_pose = d_out; a.inliers = d_inliers; a.n_inliers = d_nin; a.inlier_xy = d_inlier_xy;
  a.host_pose = h_out; a.host_inliers = h_inliers; a.host_nin = h_nin; a.host_best = h_best; a.host_bad = nullptr;
  a.chain = nullptr;
  // first the leading hypotheses only; all of them when the bookkeeping's cap stays above what was computed (h_best = -2)
  for (a.launched = svo_kg_pnp_first(iterations);; a.launched = iterations) {
    const int wgs = svo_div_up(a.launched, 4);
    const SvoPublish pub = svo_publish_next(ctx, SVO_W_PNP_REF);   // the word the last workgroup publishes
    const SvoPublish arr = svo_arrive_next(ctx, wgs);               // the arrival counter the workgroups meet at
    a.arrive = arr.arrive; a.arrive_target = arr.target; a.word = pub.word; a.seq = pub.seq;
    {
    SvoProfScope prof(ctx, SVO_PROF_PNP_HYP);
    hipLaunchKernelGGL(pnp_ransac_kernel, dim3(wgs), dim3(256), 0, st, a, d_xyz);
    }
    SVO_HIP_CHECK(ctx, hipGetLastError());
    const int rc = svo_wait_word(ctx, pub);
    if (rc) return rc;
    if (*h_best != -2 || a.launched >= iterations) break;
  }
  if (*h_best < 0) return SVO_OK;
  svo_det_rvec_from_quat(h_out, rvec3);  // declared arithmetic (host/det_trig.h)
  for (int k = 0; k < 3; ++k) tvec3[k] = h_out[4 + k];
  *n_inliers = *h_nin;
  return SVO_OK;
}

extern "C" int svo_pnp_ransac(svo_ctx* ctx, const float* xyz, const float* xy, int n, float focal, float cx, float cy,
                              double* rvec3, double* tvec3, int iterations, float reproj_err, double confidence,
                              int* inliers, int* n_inliers) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, n >= 0 && rvec3 && tvec3 && n_inliers && (n == 0 || (xyz && xy && inliers)), "pnp_ransac: null buffer");
  SVO_REQUIRE(ctx, iterations >= 1 && iterations <= 1024, "pnp_ransac: iterations must be 1..1024");
  *n_inliers = 0;
  if (n < MODEL) return SVO_OK;
  SvoScratch s(ctx);
  float* dxyz = s.take<float>(3 * (size_t)n);
  float* dxy = s.take<float>(2 * (size_t)n);
  int* din = s.take<int>(n);
  if (!dxyz || !dxy || !din) { ctx->err = "pnp_ransac: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dxyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  int rc = svo_k_pnp(ctx, s, dxyz, dxy, n, focal, cx, cy, rvec3, tvec3, iterations, reproj_err, confidence, din, n_inliers, nullptr);
  if (rc) return rc;
  if (*n_inliers > 0) {
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(inliers, din, sizeof(int) * (size_t)*n_inliers, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  return SVO_OK;
}

int svo_pnp_update_num_iters(double p, double ep, int model_points, int max_iters) { return update_num_iters(p, ep, model_points, max_iters); }
int svo_pnp_model_points() { return MODEL; }

int svo_kg_pnp(svo_ctx* ctx, hipStream_t st, const SvoPnpLanes& lanes, int n_lanes) {
  int most = 1;
  for (int i = 0; i < n_lanes; ++i) most = std::max(most, lanes.lane[i].launched);
  SvoProfScope prof(ctx, SVO_PROF_PNP_HYP, st);
  hipLaunchKernelGGL(pnp_group_kernel, dim3(svo_div_up(most, 4), n_lanes), dim3(256), 0, st, lanes);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}
int svo_kg_pnp_workgroups(int launched) { return svo_div_up(launched, 4); }
// 12 hypotheses = three workgroups: enough whenever the best of them has >= 80 % inliers (the cap is log 0.01 / log(1 - w^5):
// 5 at w = 0.9, 12 at w = 0.8, 25 at w = 0.7)
int svo_kg_pnp_first(int iterations) { return std::min(iterations, 12); }
