// oracle: BundleAdjuster::bundle_adjust -> ceres::Solve with DENSE_SCHUR
// (src/bundle_adjuster.cpp:9-12,137-157) over ReprojectionFactor residuals
// (src/reprojection_factor.cpp:10-88) with the quaternion (x) identity local parameterization
// (src/bundle_adjuster.cpp:19-20,123), oldest pose constant (:130), squared loss (:79,118).
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED vs Ceres (not available here); restates SURVEY.md
// Appendix B: trust-region Levenberg-Marquardt, Jacobi column scaling fixed at the first Jacobian,
// LM diagonal clamp(diag(J'^T J'),1e-6,1e32)/radius, Schur elimination of the 3x3 landmark blocks,
// dense Cholesky on the reduced camera system, Ceres' step acceptance / radius update / tolerances.
// Stated differences: (1) the model cost change is evaluated in its algebraically equal closed form
// 1/2 y^T (D^2 y - g'); (2) the gradient test uses the 2-norm (an upper bound of Ceres' max-norm) so
// that every quantity a sharded run decides on is a sum; (3) no wall-clock limit (SURVEY C-10).
// (Round 1 also took the candidate step when the function tolerance fired; Ceres' FunctionToleranceReached()
// returns before the step is accepted, and so does this restatement now.)
//
// Declared summation order (what makes the HIP solver bit-identical to this one, and this one independent of its
// thread count).  Ceres' own order is unspecified (4 threads, src/bundle_adjuster.cpp:12): this restatement DEFINES one,
// chosen (round 4) so that it is cheap on the GPU — sums stay inside a wavefront's registers / LDS as long as possible:
//   observations are landmark-major; CHUNKS are formed greedily from whole landmarks, a chunk closes when the next
//     landmark would take it beyond 64 observations (landmarks without observations are skipped);
//   level 0  every sum over the observations of ONE landmark runs sequentially in observation order (V, g_p, the
//            landmark's cost); a chunk's partial P_c[e] of payload element e is the sequential sum, from +0.0, of the
//            chunk's contributions to e in (landmark, observation i, observation t) order;
//   level 1  GROUPS of G consecutive chunks, G = 1 for C <= 128 chunks, else ceil(C / 128): Q_g[e] = P_c0[e] + P_c1[e] + ...
//            sequentially in chunk order (starting FROM the first partial);
//   level 2  total[e] = Q_0[e] + Q_1[e] + ... sequentially in group order (starting from Q_0[e]).
// Elements e ("wire format", E = 36 F(F+1)/2 + 33 F + 2 with F = K - 1 free poses):
//   Schur part of every UPPER pose-pair block (ka <= kb), 36 each: pair (i, t >= i) of a landmark contributes
//     B = -(Y_i (W_t s)^T) to block (k_i, k_t) when k_i <= k_t, and B^T to block (k_t, k_i) when t != i and k_t <= k_i
//     (two observations of one landmark in one pose: both, the direct entry first);
//   per free pose 33: g_c (6) | the -Y g_p part of the reduced gradient (6) | the upper triangle of U = J_c^T J_c (21),
//     one contribution per observation of that pose;
//   cost | sum g_p^2, one contribution per landmark.
// Assembly (after the totals): S[(k,a),(k,b)] = U_k[min(a,b)][max(a,b)] + Schur_(k,k)[a][b]; S[(ka,a),(kb,b)] =
// Schur_(ka,kb)[a][b] for ka < kb and its exact transpose below the diagonal blocks; diag U[6k+a] = U_k[a][a].
// payload2 (pass B's four scalars per landmark) goes through the same three levels.
// (Rounds 1-3 declared "28 strided segments over destination-ordered per-pair 6x6 slots": every pair block had to be
// written through to memory and read back.  ora_ba_set_order(1) still selects that order — only for the comparison
// tests/test_ba.py::test_declared_orders_agree and profiles/r04_order_comparison.txt.)
//
// Sharded form: a rank holds all poses and a subset of landmarks.  Per LM iteration it sums
//   payload1 = [ U - sum_j W_j Vd_j^-1 W_j^T  (n x n) | g_c - sum_j W_j Vd_j^-1 g_pj (n) | g_c (n) |
//                diag U (n) | cost | sum g_p^2 ]              (pose block unscaled, n = 6 (K-1))
//   payload2 = [ candidate cost | point part of the model change | sum dp^2 | sum p^2 ]
// through `allreduce`, so every rank takes identical decisions.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svo_oracle.h"

namespace {
struct Problem {
  int K, N, M, n;
  double* poses;
  double* points;
  const int32_t *op, *oj;
  const double* uv;
  double f, cx, cy;
  std::vector<int> lm_start;  // CSR over landmarks that have observations
  std::vector<int> lm_id;
};

// residual + tangent Jacobians for one observation. Jc: 2x6 (pose tangent), Jp: 2x3.
inline void eval_obs(const Problem& P, const double* poses, const double* points, int o, double* r,
                     double* Jc, double* Jp) {
  double jp14[14], jx6[6];
  const double* pose = poses + 7 * P.op[o];
  ora_reproj_eval(1, pose, points + 3 * P.oj[o], P.uv + 2 * o, P.f, P.cx, P.cy, r,
                  Jc ? jp14 : nullptr, Jp ? jx6 : nullptr);
  if (Jc) {
    const double w = pose[0], x = pose[1], y = pose[2], z = pose[3];
    // d q / d delta for q+ = [cos|d|, sin|d|/|d| d] (x) q at d=0 : rows (w,x,y,z)
    const double T[4][3] = {{-x, -y, -z}, {w, z, -y}, {-z, w, x}, {y, -x, w}};
    for (int row = 0; row < 2; ++row) {
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += jp14[7 * row + k] * T[k][c];
        Jc[6 * row + c] = s;
      }
      for (int c = 0; c < 3; ++c) Jc[6 * row + 3 + c] = jp14[7 * row + 4 + c];
    }
  }
  if (Jp) std::memcpy(Jp, jx6, sizeof(jx6));
}

inline bool inv3_sym(const double* V, double* Vi) {
  const double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(std::fabs(det) > 0)) return false;
  const double id = 1.0 / det;
  Vi[0] = c00 * id; Vi[1] = c01 * id; Vi[2] = c02 * id;
  Vi[3] = Vi[1]; Vi[4] = (a * f - c * c) * id; Vi[5] = (b * c - a * e) * id;
  Vi[6] = Vi[2]; Vi[7] = Vi[5]; Vi[8] = (a * d - b * b) * id;
  return true;
}

bool cholesky_solve(std::vector<double>& A, std::vector<double>& b, int n) {
  for (int j = 0; j < n; ++j) {
    double s = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(s > 0)) return false;
    const double l = std::sqrt(s);
    A[(size_t)j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double v = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / l;
    }
  }
  // substitutions: the pivots' reciprocals are formed ONCE (r_i = 1 / l_ii) and every row is scaled by multiplication — on the
  // device the 2 n divisions of the plain form were a dependent chain of ~100 cycles each (declared in round 4, together with
  // stereo_vo_amd/host/linalg.cpp and csrc/lm_device.h)
  std::vector<double> r(n > 0 ? n : 1);
  for (int i = 0; i < n; ++i) r[i] = 1.0 / A[(size_t)i * n + i];
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * b[k];
    b[i] = v * r[i];
  }
  for (int i = n - 1; i >= 0; --i) {  // inner index DESCENDING: the order a parallel column sweep produces
    double v = b[i];
    for (int k = n - 1; k > i; --k) v -= A[(size_t)k * n + i] * b[k];
    b[i] = v * r[i];
  }
  return true;
}

// sin/cos with a declared operation sequence (no libm): halve until <= 0.5, 8-term Taylor polynomials in
// Horner form, then double-angle steps.  Bit-identical on CPU and GPU (no FMA contraction).
void det_sincos(double x, double* sn, double* cs) {
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

void plus_pose(const double* p, const double* d, double* out) {
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
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
}  // namespace

namespace {
// R(list): the declared reduction.  The list is cut into 28 consecutive segments of ceil(len/28) entries;
// each segment is summed sequentially, then the 28 segment sums are summed sequentially.
// get(e, out) yields `width` doubles of list entry e.
template <typename Get>
void reduce_list(int count, int width, Get get, double* out) {
  const int S = 28;
  const int seglen = (count + S - 1) / S;
  std::vector<double> part((size_t)S * width, 0.0), v(width);
  for (int sg = 0; sg < S; ++sg) {
    const int e1 = std::min(count, (sg + 1) * seglen);
    for (int e = sg * seglen; e < e1; ++e) {
      get(e, v.data());
      for (int w = 0; w < width; ++w) part[(size_t)sg * width + w] += v[w];
    }
  }
  for (int w = 0; w < width; ++w) {
    double acc = 0.0;
    for (int sg = 0; sg < S; ++sg) acc += part[(size_t)sg * width + w];
    out[w] = acc;
  }
}
}  // namespace


// ---------------------------------------------------------------------------------------------------------------
// Stateful form: one object holds the problem, the current point (poses, points), the candidate and the Jacobi scales.
// ora_ba_solve below drives it with the plain (non-speculative) LM loop; tests also drive it through the PRODUCT's
// step control (svo_lm_solve in libsvo_hip.so) to check that host logic without a GPU.
static int g_order = 2;  // 2: chunk order (declared above); 1: rounds 1-3's 28-segment order (comparison only)
extern "C" void ora_ba_set_order(int order) { g_order = order == 1 ? 1 : 2; }

struct ora_ba_state {
  int order = 2;
  // chunk order: landmarks (indices into P.lm_start) of every chunk, groups of chunks, wire-format sizes
  std::vector<int> chunk_lm;   // chunk c = landmarks [chunk_lm[c], chunk_lm[c + 1])
  int C = 0, G = 1, NG = 0, nU = 0, E = 0;
  Problem P;
  std::vector<double> poses, points;            // current point (owned copies)
  std::vector<double> cand_poses, cand_points;  // candidate
  std::vector<int> pair_base;
  std::vector<std::vector<int>> blk_list, pose_list;
  std::vector<double> pairB, obsV, lmV, sp;
  int L = 0, F = 0, num_threads = 1;
  bool have_scale = false;
  size_t pay1 = 0;
};

extern "C" ora_ba_state* ora_ba_open(int n_poses, const double* poses7, int n_points, const double* points3, int n_obs,
                                     const int32_t* obs_pose, const int32_t* obs_point, const double* obs_uv,
                                     double focal, double cx, double cy, int num_threads) {
  ora_ba_state* S = new ora_ba_state();
  S->poses.assign(poses7, poses7 + 7 * (size_t)n_poses);
  S->points.assign(points3, points3 + 3 * (size_t)n_points);
  S->cand_poses = S->poses;
  S->cand_points = S->points;
  Problem& P = S->P;
  P = Problem{n_poses, n_points, n_obs, 6 * (n_poses - 1), S->poses.data(), S->points.data(), obs_pose, obs_point, obs_uv, focal, cx, cy, {}, {}};
  for (int o = 0; o < n_obs; ++o)
    if (o == 0 || obs_point[o] != obs_point[o - 1]) { P.lm_start.push_back(o); P.lm_id.push_back(obs_point[o]); }
  P.lm_start.push_back(n_obs);
  S->L = (int)P.lm_id.size();
  S->F = n_poses - 1;
  S->num_threads = num_threads < 1 ? 1 : num_threads;
  const int L = S->L, F = S->F, n = P.n;
  S->pay1 = (size_t)n * n + 3 * (size_t)n + 2;
  S->order = g_order;
  {
    // chunks: whole landmarks, greedily, at most 64 observations each
    // (tried in round 4: also at most 32 observations of a chunk in any one pose, so that no sequential list inside a chunk is
    // longer than 32 — an LM iteration alone on the GPU went from 45 to 41 us, but the 12 % more chunks = workgroups per solve
    // cost the loaded machine 4-5 % of its frame rate: 18.0-18.2 k against 18.8-19.1 k frames/s; rejected)
    S->chunk_lm.push_back(0);
    int cur = 0;
    for (int l = 0; l < L; ++l) {
      const int len = P.lm_start[l + 1] - P.lm_start[l];
      if (cur + len > 64) { S->chunk_lm.push_back(l); cur = 0; }
      cur += len;
    }
    if (L > 0) S->chunk_lm.push_back(L);
    S->C = (int)S->chunk_lm.size() - 1;
    S->G = S->C <= 128 ? 1 : (S->C + 127) / 128;
    S->NG = S->C > 0 ? (S->C + S->G - 1) / S->G : 0;
    S->nU = F * (F + 1) / 2;
    S->E = 36 * S->nU + 33 * F + 2;
  }
  if (S->order == 2) return S;
  // ---- order 1 only: contribution slots and destination lists (landmark order)
  S->pair_base.assign((size_t)n_obs + 1, 0);
  for (int l = 0; l < L; ++l)
    for (int o = P.lm_start[l]; o < P.lm_start[l + 1]; ++o) S->pair_base[o + 1] = S->pair_base[o] + (P.lm_start[l + 1] - o);
  S->blk_list.assign((size_t)F * F, {});
  S->pose_list.assign(F, {});
  for (int l = 0; l < L; ++l)
    for (int i = P.lm_start[l]; i < P.lm_start[l + 1]; ++i) {
      const int ki = obs_pose[i] - 1;
      if (ki < 0) continue;
      S->pose_list[ki].push_back(i);
      for (int t = i; t < P.lm_start[l + 1]; ++t) {
        const int kt = obs_pose[t] - 1;
        if (kt < 0) continue;
        const int slot = S->pair_base[i] + (t - i);
        S->blk_list[(size_t)ki * F + kt].push_back(slot * 2);  // entry = slot*2 + transposed
        if (t != i) S->blk_list[(size_t)kt * F + ki].push_back(slot * 2 + 1);
      }
    }
  S->pairB.assign((size_t)S->pair_base[n_obs] * 36, 0.0);
  S->obsV.assign((size_t)n_obs * 18, 0.0);
  S->lmV.assign((size_t)L * 4, 0.0);
  S->sp.assign((size_t)L * 3, 0.0);
  return S;
}

extern "C" void ora_ba_close(ora_ba_state* S) { delete S; }
extern "C" size_t ora_ba_payload1_len(const ora_ba_state* S) { return S->pay1; }

extern "C" void ora_ba_read(const ora_ba_state* S, double* poses7, double* points3) {
  if (poses7) std::memcpy(poses7, S->poses.data(), sizeof(double) * S->poses.size());
  if (points3) std::memcpy(points3, S->points.data(), sizeof(double) * S->points.size());
}


// ---------------------------------------------------------------------------------------------------------------
// Chunk order (the declared order, see the head of this file).
namespace {
inline int upper_index(int ka, int kb, int F) { return ka * F - ka * (ka - 1) / 2 + (kb - ka); }

// levels 1 and 2 over per-chunk partials part[c * width + e] -> out[e]
void sum_partials(const ora_ba_state* S, const std::vector<double>& part, int width, double* out) {
  const int C = S->C, G = S->G, NG = S->NG;
  for (int e = 0; e < width; ++e) {
    double tot = 0.0;
    for (int g = 0; g < NG; ++g) {
      const int c0 = g * G, c1 = std::min(C, c0 + G);
      double q = part[(size_t)c0 * width + e];                                   // level 1 starts FROM the first partial
      for (int c = c0 + 1; c < c1; ++c) q += part[(size_t)c * width + e];
      tot = g == 0 ? q : tot + q;                                                // level 2 starts from Q_0
    }
    out[e] = tot;
  }
}

// pass A in chunk order: the wire-format totals, then the assembled payload1 [S | g_red | g_c | diag U | cost | sum g_p^2]
void linearize_v2(ora_ba_state* S, int at_candidate, double rad, int first, double* pay) {
  const Problem& P = S->P;
  const double* poses = at_candidate ? S->cand_poses.data() : S->poses.data();
  const double* points = at_candidate ? S->cand_points.data() : S->points.data();
  const int F = S->F, n = P.n, nU = S->nU, E = S->E, C = S->C;
  const double min_diag = 1e-6, max_diag = 1e32;
  std::vector<double>& sp = S->sp;
  if (sp.size() < (size_t)S->L * 3) sp.assign((size_t)S->L * 3, 0.0);
  std::vector<double> part((size_t)C * E, 0.0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(S->num_threads)
  for (int c = 0; c < C; ++c) {
    double* Pc = &part[(size_t)c * E];  // level 0: every element starts at +0.0, contributions in (landmark, i, t) order
    double* Pv = Pc + 36 * (size_t)nU;  // per pose 33
    double* Ps = Pv + 33 * (size_t)F;   // cost, sum g_p^2
    for (int l = S->chunk_lm[c]; l < S->chunk_lm[c + 1]; ++l) {
      const int o0 = P.lm_start[l], o1 = P.lm_start[l + 1], len = o1 - o0;
      std::vector<double> R((size_t)len * 2), JC((size_t)len * 12, 0.0), JP((size_t)len * 6), WS((size_t)len * 18), YY((size_t)len * 18);
      double V[9] = {0}, gp[3] = {0}, cost_l = 0.0;
      for (int o = o0; o < o1; ++o) {
        double* r = &R[2 * (o - o0)];
        double* Jc = &JC[12 * (o - o0)];
        double* Jp = &JP[6 * (o - o0)];
        eval_obs(P, poses, points, o, r, P.op[o] > 0 ? Jc : nullptr, Jp);
        cost_l += 0.5 * (r[0] * r[0] + r[1] * r[1]);
        for (int a = 0; a < 3; ++a) {
          gp[a] += Jp[a] * r[0] + Jp[3 + a] * r[1];
          for (int b = 0; b < 3; ++b) V[3 * a + b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
        }
      }
      double* s = &sp[(size_t)l * 3];
      if (first) for (int a = 0; a < 3; ++a) s[a] = 1.0 / (1.0 + std::sqrt(V[4 * a]));
      double Vd[9], gps[3], Vi[9] = {0};
      for (int a = 0; a < 3; ++a) {
        gps[a] = gp[a] * s[a];
        for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
      }
      for (int a = 0; a < 3; ++a) Vd[4 * a] += std::min(std::max(Vd[4 * a], min_diag), max_diag) / rad;
      inv3_sym(Vd, Vi);
      Ps[0] += cost_l;
      Ps[1] += gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
      for (int o = o0; o < o1; ++o) {
        if (P.op[o] <= 0) continue;
        const double* r = &R[2 * (o - o0)];
        const double* Jc = &JC[12 * (o - o0)];
        const double* Jp = &JP[6 * (o - o0)];
        double* Ws = &WS[18 * (o - o0)];
        double* Y = &YY[18 * (o - o0)];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 3; ++b) Ws[3 * a + b] = (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b]) * s[b];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 3; ++b) Y[3 * a + b] = Ws[3 * a] * Vi[b] + Ws[3 * a + 1] * Vi[3 + b] + Ws[3 * a + 2] * Vi[6 + b];
        double* pv = Pv + 33 * (size_t)(P.op[o] - 1);
        for (int a = 0; a < 6; ++a) {
          pv[a] += Jc[a] * r[0] + Jc[6 + a] * r[1];
          pv[6 + a] += -(Y[3 * a] * gps[0] + Y[3 * a + 1] * gps[1] + Y[3 * a + 2] * gps[2]);
        }
        int u = 12;
        for (int a = 0; a < 6; ++a)
          for (int b = a; b < 6; ++b) pv[u++] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
      }
      for (int i = o0; i < o1; ++i) {
        if (P.op[i] <= 0) continue;
        const int ki = P.op[i] - 1;
        const double* Y = &YY[18 * (i - o0)];
        for (int t = i; t < o1; ++t) {
          if (P.op[t] <= 0) continue;
          const int kt = P.op[t] - 1;
          const double* Wt = &WS[18 * (t - o0)];
          double B[36];
          for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) B[6 * a + b] = -(Y[3 * a] * Wt[3 * b] + Y[3 * a + 1] * Wt[3 * b + 1] + Y[3 * a + 2] * Wt[3 * b + 2]);
          if (ki <= kt) {
            double* d = Pc + 36 * (size_t)upper_index(ki, kt, F);
            for (int e = 0; e < 36; ++e) d[e] += B[e];
          }
          if (t != i && kt <= ki) {
            double* d = Pc + 36 * (size_t)upper_index(kt, ki, F);
            for (int a = 0; a < 6; ++a)
              for (int b = 0; b < 6; ++b) d[6 * b + a] += B[6 * a + b];
          }
        }
      }
    }
  }
  if (first) S->have_scale = true;
  std::vector<double> tot(E > 0 ? E : 1, 0.0);
  if (C > 0) sum_partials(S, part, E, tot.data());
  // assembly
  std::fill(pay, pay + S->pay1, 0.0);
  double* Sx = pay;
  double* gred = Sx + (size_t)n * n;
  double* gc = gred + n;
  double* dU = gc + n;
  for (int ka = 0; ka < F; ++ka)
    for (int kb = ka; kb < F; ++kb) {
      const double* blk = &tot[36 * (size_t)upper_index(ka, kb, F)];
      const double* U = &tot[36 * (size_t)nU + 33 * (size_t)ka + 12];
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          double v = blk[6 * a + b];
          if (ka == kb) {
            const int lo = std::min(a, b), hi = std::max(a, b);
            v = U[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)] + v;
          }
          Sx[(size_t)(6 * ka + a) * n + 6 * kb + b] = v;
          if (ka != kb) Sx[(size_t)(6 * kb + b) * n + 6 * ka + a] = v;
        }
    }
  for (int k = 0; k < F; ++k) {
    const double* pv = &tot[36 * (size_t)nU + 33 * (size_t)k];
    for (int a = 0; a < 6; ++a) {
      gc[6 * k + a] = pv[a];
      gred[6 * k + a] = pv[6 + a];
      dU[6 * k + a] = pv[12 + a * 6 - a * (a - 1) / 2];
    }
  }
  pay[S->pay1 - 2] = tot[E - 2];
  pay[S->pay1 - 1] = tot[E - 1];
}

void backsub_v2(ora_ba_state* S, const double* dc, const double* cand_poses7, double rad, double* pay2) {
  const Problem& P = S->P;
  const int C = S->C;
  const double min_diag = 1e-6, max_diag = 1e32;
  std::memcpy(S->cand_poses.data(), cand_poses7, sizeof(double) * S->cand_poses.size());
  S->cand_points = S->points;
  std::vector<double>&sp = S->sp, &cand_points = S->cand_points;
  std::vector<double> part((size_t)C * 4, 0.0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(S->num_threads)
  for (int c = 0; c < C; ++c) {
    double* Pc = &part[(size_t)c * 4];
    for (int l = S->chunk_lm[c]; l < S->chunk_lm[c + 1]; ++l) {
      const int o0 = P.lm_start[l], o1 = P.lm_start[l + 1];
      double V[9] = {0}, gp[3] = {0}, wd[3] = {0};
      for (int o = o0; o < o1; ++o) {
        double r[2], Jc[12], Jp[6];
        const int k = P.op[o];
        eval_obs(P, S->poses.data(), S->points.data(), o, r, k > 0 ? Jc : nullptr, Jp);
        double jd[2] = {0, 0};
        if (k > 0)
          for (int a = 0; a < 6; ++a) { jd[0] += Jc[a] * dc[6 * (k - 1) + a]; jd[1] += Jc[6 + a] * dc[6 * (k - 1) + a]; }
        for (int a = 0; a < 3; ++a) {
          gp[a] += Jp[a] * r[0] + Jp[3 + a] * r[1];
          wd[a] += Jp[a] * jd[0] + Jp[3 + a] * jd[1];
          for (int b = 0; b < 3; ++b) V[3 * a + b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
        }
      }
      const double* s = &sp[(size_t)l * 3];
      double Vd[9], De[3], rh[3], Vi[9] = {0};
      for (int a = 0; a < 3; ++a) {
        rh[a] = -(gp[a] + wd[a]) * s[a];
        for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
      }
      for (int a = 0; a < 3; ++a) { De[a] = std::min(std::max(Vd[4 * a], min_diag), max_diag) / rad; Vd[4 * a] += De[a]; }
      inv3_sym(Vd, Vi);
      const double* p0 = &S->points[3 * (size_t)P.lm_id[l]];
      double* pt = &cand_points[3 * (size_t)P.lm_id[l]];
      double mc = 0, dp2 = 0, p2 = 0;
      for (int a = 0; a < 3; ++a) {
        const double y = Vi[3 * a] * rh[0] + Vi[3 * a + 1] * rh[1] + Vi[3 * a + 2] * rh[2];
        mc += 0.5 * y * (De[a] * y - gp[a] * s[a]);
        const double d = y * s[a];
        dp2 += d * d;
        p2 += p0[a] * p0[a];
        pt[a] = p0[a] + d;
      }
      double cn = 0;
      for (int o = o0; o < o1; ++o) {
        double r[2];
        eval_obs(P, S->cand_poses.data(), cand_points.data(), o, r, nullptr, nullptr);
        cn += 0.5 * (r[0] * r[0] + r[1] * r[1]);
      }
      Pc[0] += cn; Pc[1] += mc; Pc[2] += dp2; Pc[3] += p2;
    }
  }
  pay2[0] = pay2[1] = pay2[2] = pay2[3] = 0.0;
  if (C > 0) sum_partials(S, part, 4, pay2);
}
}  // namespace

// pass A: linearise at the current point (at_candidate = 0) or at the candidate (1), fill the slots, reduce into
// this rank's payload1 for `rad`.  `first` fixes the Jacobi scales of the landmarks from this Jacobian.
extern "C" void ora_ba_linearize(ora_ba_state* S, int at_candidate, double rad, int first, double* pay) {
  if (S->order == 2) { linearize_v2(S, at_candidate, rad, first, pay); return; }
  const Problem& P = S->P;
  const double* poses = at_candidate ? S->cand_poses.data() : S->poses.data();
  const double* points = at_candidate ? S->cand_points.data() : S->points.data();
  const int L = S->L, F = S->F, n = P.n, num_threads = S->num_threads;
  const double min_diag = 1e-6, max_diag = 1e32;
  std::vector<double>&pairB = S->pairB, &obsV = S->obsV, &lmV = S->lmV, &sp = S->sp;
  const std::vector<int>& pair_base = S->pair_base;
#pragma omp parallel for schedule(static) num_threads(num_threads)
  for (int l = 0; l < L; ++l) {
    const int o0 = P.lm_start[l], o1 = P.lm_start[l + 1], len = o1 - o0;
    std::vector<double> R((size_t)len * 2), JC((size_t)len * 12, 0.0), JP((size_t)len * 6), WS((size_t)len * 18), YY((size_t)len * 18);
    double V[9] = {0}, gp[3] = {0}, cost_l = 0.0;
    for (int o = o0; o < o1; ++o) {
      double* r = &R[2 * (o - o0)];
      double* Jc = &JC[12 * (o - o0)];
      double* Jp = &JP[6 * (o - o0)];
      eval_obs(P, poses, points, o, r, P.op[o] > 0 ? Jc : nullptr, Jp);
      cost_l += 0.5 * (r[0] * r[0] + r[1] * r[1]);
      for (int a = 0; a < 3; ++a) {
        gp[a] += Jp[a] * r[0] + Jp[3 + a] * r[1];
        for (int b = 0; b < 3; ++b) V[3 * a + b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
      }
    }
    double* s = &sp[(size_t)l * 3];
    if (first) for (int a = 0; a < 3; ++a) s[a] = 1.0 / (1.0 + std::sqrt(V[4 * a]));
    double Vd[9], gps[3], Vi[9] = {0};
    for (int a = 0; a < 3; ++a) {
      gps[a] = gp[a] * s[a];
      for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
    }
    for (int a = 0; a < 3; ++a) Vd[4 * a] += std::min(std::max(Vd[4 * a], min_diag), max_diag) / rad;
    inv3_sym(Vd, Vi);
    lmV[4 * (size_t)l] = cost_l;
    lmV[4 * (size_t)l + 1] = gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
    for (int o = o0; o < o1; ++o) {
      if (P.op[o] <= 0) continue;
      const double* r = &R[2 * (o - o0)];
      const double* Jc = &JC[12 * (o - o0)];
      const double* Jp = &JP[6 * (o - o0)];
      double* Ws = &WS[18 * (o - o0)];
      double* Y = &YY[18 * (o - o0)];
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) Ws[3 * a + b] = (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b]) * s[b];
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) Y[3 * a + b] = Ws[3 * a] * Vi[b] + Ws[3 * a + 1] * Vi[3 + b] + Ws[3 * a + 2] * Vi[6 + b];
      double* ov = &obsV[(size_t)o * 18];
      for (int a = 0; a < 6; ++a) {
        ov[a] = Jc[a] * r[0] + Jc[6 + a] * r[1];
        ov[6 + a] = -(Y[3 * a] * gps[0] + Y[3 * a + 1] * gps[1] + Y[3 * a + 2] * gps[2]);
        ov[12 + a] = Jc[a] * Jc[a] + Jc[6 + a] * Jc[6 + a];
      }
    }
    for (int i = o0; i < o1; ++i) {
      if (P.op[i] <= 0) continue;
      const double* Y = &YY[18 * (i - o0)];
      const double* Jc = &JC[12 * (i - o0)];
      for (int t = i; t < o1; ++t) {
        if (P.op[t] <= 0) continue;
        const double* Wt = &WS[18 * (t - o0)];
        double* B = &pairB[(size_t)(pair_base[i] + (t - i)) * 36];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) {
            const double v = -(Y[3 * a] * Wt[3 * b] + Y[3 * a + 1] * Wt[3 * b + 1] + Y[3 * a + 2] * Wt[3 * b + 2]);
            B[6 * a + b] = t == i ? (Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]) + v : v;
          }
      }
    }
  }
  if (first) S->have_scale = true;
  std::fill(pay, pay + S->pay1, 0.0);
  double* Sx = pay;
  double* gred = Sx + (size_t)n * n;
  double* gc = gred + n;
  double* dU = gc + n;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
  for (int d = 0; d < F * F + F + 1; ++d) {
    if (d < F * F) {
      const std::vector<int>& lst = S->blk_list[d];
      double B[36];
      reduce_list((int)lst.size(), 36, [&](int e, double* out) {
        const double* src = &pairB[(size_t)(lst[e] >> 1) * 36];
        if (lst[e] & 1) { for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out[6 * a + b] = src[6 * b + a]; }
        else std::memcpy(out, src, 36 * sizeof(double));
      }, B);
      const int ka = d / F, kb = d % F;
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) Sx[(size_t)(6 * ka + a) * n + 6 * kb + b] = B[6 * a + b];
    } else if (d < F * F + F) {
      const int k = d - F * F;
      const std::vector<int>& lst = S->pose_list[k];
      double v[18];
      reduce_list((int)lst.size(), 18, [&](int e, double* out) { std::memcpy(out, &obsV[(size_t)lst[e] * 18], 18 * sizeof(double)); }, v);
      for (int a = 0; a < 6; ++a) { gc[6 * k + a] = v[a]; gred[6 * k + a] = v[6 + a]; dU[6 * k + a] = v[12 + a]; }
    } else {
      double v[2];
      reduce_list(L, 2, [&](int e, double* out) { out[0] = lmV[4 * (size_t)e]; out[1] = lmV[4 * (size_t)e + 1]; }, v);
      pay[S->pay1 - 2] = v[0];
      pay[S->pay1 - 1] = v[1];
    }
  }
}

// pass B: back-substitute at the current point with the pose step dc (unscaled tangent) and the candidate poses the
// caller formed from it; builds the candidate landmarks and this rank's payload2.
extern "C" void ora_ba_backsub(ora_ba_state* S, const double* dc, const double* cand_poses7, double rad, double* pay2) {
  if (S->order == 2) { backsub_v2(S, dc, cand_poses7, rad, pay2); return; }
  const Problem& P = S->P;
  const int L = S->L, num_threads = S->num_threads;
  const double min_diag = 1e-6, max_diag = 1e32;
  std::memcpy(S->cand_poses.data(), cand_poses7, sizeof(double) * S->cand_poses.size());
  S->cand_points = S->points;
  std::vector<double>&lmV = S->lmV, &sp = S->sp, &cand_points = S->cand_points;
#pragma omp parallel for schedule(static) num_threads(num_threads)
  for (int l = 0; l < L; ++l) {
    const int o0 = P.lm_start[l], o1 = P.lm_start[l + 1];
    double V[9] = {0}, gp[3] = {0}, wd[3] = {0};
    for (int o = o0; o < o1; ++o) {
      double r[2], Jc[12], Jp[6];
      const int k = P.op[o];
      eval_obs(P, S->poses.data(), S->points.data(), o, r, k > 0 ? Jc : nullptr, Jp);
      double jd[2] = {0, 0};
      if (k > 0)
        for (int a = 0; a < 6; ++a) { jd[0] += Jc[a] * dc[6 * (k - 1) + a]; jd[1] += Jc[6 + a] * dc[6 * (k - 1) + a]; }
      for (int a = 0; a < 3; ++a) {
        gp[a] += Jp[a] * r[0] + Jp[3 + a] * r[1];
        wd[a] += Jp[a] * jd[0] + Jp[3 + a] * jd[1];
        for (int b = 0; b < 3; ++b) V[3 * a + b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
      }
    }
    const double* s = &sp[(size_t)l * 3];
    double Vd[9], De[3], rh[3], Vi[9] = {0};
    for (int a = 0; a < 3; ++a) {
      rh[a] = -(gp[a] + wd[a]) * s[a];
      for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
    }
    for (int a = 0; a < 3; ++a) { De[a] = std::min(std::max(Vd[4 * a], min_diag), max_diag) / rad; Vd[4 * a] += De[a]; }
    inv3_sym(Vd, Vi);
    const double* p0 = &S->points[3 * (size_t)P.lm_id[l]];
    double* pt = &cand_points[3 * (size_t)P.lm_id[l]];
    double mc = 0, dp2 = 0, p2 = 0;
    for (int a = 0; a < 3; ++a) {
      const double y = Vi[3 * a] * rh[0] + Vi[3 * a + 1] * rh[1] + Vi[3 * a + 2] * rh[2];
      mc += 0.5 * y * (De[a] * y - gp[a] * s[a]);
      const double d = y * s[a];
      dp2 += d * d;
      p2 += p0[a] * p0[a];
      pt[a] = p0[a] + d;
    }
    double cn = 0;
    for (int o = o0; o < o1; ++o) {
      double r[2];
      eval_obs(P, S->cand_poses.data(), cand_points.data(), o, r, nullptr, nullptr);
      cn += 0.5 * (r[0] * r[0] + r[1] * r[1]);
    }
    lmV[4 * (size_t)l] = cn; lmV[4 * (size_t)l + 1] = mc; lmV[4 * (size_t)l + 2] = dp2; lmV[4 * (size_t)l + 3] = p2;
  }
  reduce_list(L, 4, [&](int e, double* out) { std::memcpy(out, &lmV[4 * (size_t)e], 4 * sizeof(double)); }, pay2);
}

// the candidate becomes the current point
extern "C" void ora_ba_accept(ora_ba_state* S) {
  S->poses = S->cand_poses;
  S->points = S->cand_points;
}

extern "C" int ora_ba_solve(int n_poses, double* poses7, int n_points, double* points3, int n_obs,
                            const int32_t* obs_pose, const int32_t* obs_point, const double* obs_uv,
                            double focal, double cx, double cy, int max_iterations,
                            double function_tol, double gradient_tol, double parameter_tol,
                            double initial_radius, int num_threads, ora_allreduce_fn allreduce,
                            void* user, double* summary5) {
  ora_ba_state* S = ora_ba_open(n_poses, poses7, n_points, points3, n_obs, obs_pose, obs_point, obs_uv, focal, cx, cy, num_threads);
  const int n = S->P.n;
  const size_t pay1 = S->pay1;
  std::vector<double> sc(n, 0.0), pay(pay1), Sm((size_t)n * n), rhs(n), dc(n), cand_poses((size_t)7 * n_poses);
  bool have_scale = false;
  double radius = initial_radius, decrease_factor = 2.0;
  const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  const double min_rel_decrease = 1e-3;
  double pay2[4];
  auto linearize = [&](double rad) {
    ora_ba_linearize(S, 0, rad, have_scale ? 0 : 1, pay.data());
    if (allreduce) allreduce(pay.data(), pay1, user);
  };
  auto backsub = [&](double rad) {
    for (int k = 0; k < n_poses; ++k) {
      if (k == 0) std::memcpy(&cand_poses[0], S->poses.data(), 7 * sizeof(double));
      else plus_pose(S->poses.data() + 7 * k, &dc[6 * (k - 1)], &cand_poses[7 * k]);
    }
    ora_ba_backsub(S, dc.data(), cand_poses.data(), rad, pay2);
    if (allreduce) allreduce(pay2, 4, user);
  };

  int iterations = 0, successful = 0, termination = 1;
  linearize(radius);
  double cost = pay[pay1 - 2];
  const double initial_cost = cost;
  bool need_linearize = false;
  {
    const double* gc = pay.data() + (size_t)n * n + n;
    const double* dU = gc + n;
    for (int a = 0; a < n; ++a) sc[a] = 1.0 / (1.0 + std::sqrt(dU[a]));
    have_scale = true;
    double g2 = pay[pay1 - 1];
    for (int a = 0; a < n; ++a) g2 += gc[a] * gc[a];
    if (std::sqrt(g2) <= gradient_tol) { termination = 0; goto done; }
  }
  while (true) {
    if (iterations >= max_iterations) { termination = 1; break; }
    if (radius <= min_radius) { termination = 0; break; }
    ++iterations;
    if (need_linearize) { linearize(radius); need_linearize = false; }
    const double* Sx = pay.data();
    const double* gred = Sx + (size_t)n * n;
    const double* gc = gred + n;
    const double* dU = gc + n;
    std::vector<double> Df(n);
    for (int a = 0; a < n; ++a) {
      Df[a] = std::min(std::max(dU[a] * sc[a] * sc[a], min_diag), max_diag) / radius;
      for (int b = 0; b < n; ++b) Sm[(size_t)a * n + b] = Sx[(size_t)a * n + b] * sc[a] * sc[b];
      Sm[(size_t)a * n + a] += Df[a];
      rhs[a] = -(gred[a] + gc[a]) * sc[a];
    }
    bool ok = n == 0 || cholesky_solve(Sm, rhs, n);
    bool step_ok = false;
    double cost_new = 0, model_change = 0, step2 = 0, x2 = 0;
    if (ok) {
      double mcc = 0;
      for (int a = 0; a < n; ++a) {
        mcc += 0.5 * rhs[a] * (Df[a] * rhs[a] - gc[a] * sc[a]);
        dc[a] = rhs[a] * sc[a];
      }
      backsub(radius);
      cost_new = pay2[0];
      model_change = mcc + pay2[1];
      step2 = pay2[2]; x2 = pay2[3];
      for (int k = 1; k < n_poses; ++k)
        for (int a = 0; a < 7; ++a) {
          const double d = cand_poses[7 * k + a] - S->poses[7 * k + a];
          step2 += d * d;
          x2 += S->poses[7 * k + a] * S->poses[7 * k + a];
        }
      step_ok = model_change > 0;
    }
    if (!step_ok) {  // invalid step
      radius /= decrease_factor; decrease_factor *= 2; need_linearize = true;
      continue;
    }
    if (std::sqrt(step2) <= parameter_tol * (std::sqrt(x2) + parameter_tol)) { termination = 0; break; }
    const double cost_change = cost - cost_new;
    if (std::fabs(cost_change) <= function_tol * cost) {  // Ceres: FunctionToleranceReached() returns BEFORE the step is
      termination = 0;                                    // tested / taken: the candidate is discarded, x stays
      break;
    }
    const double rho = cost_change / model_change;
    if (std::getenv("SVO_BA_TRACE"))
      std::fprintf(stderr, "[ora] it %d cost %.17g new %.17g model %.17g rho %.6g radius %.6g\n", iterations, cost, cost_new, model_change, rho, radius);
    if (rho > min_rel_decrease) {
      ora_ba_accept(S);
      cost = cost_new;
      ++successful;
      const double t = 2.0 * rho - 1.0;
      radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
      radius = std::min(max_radius, radius);
      decrease_factor = 2.0;
      linearize(radius);
      const double* gc2 = pay.data() + (size_t)n * n + n;
      double g2 = pay[pay1 - 1];
      for (int a = 0; a < n; ++a) g2 += gc2[a] * gc2[a];
      if (std::sqrt(g2) <= gradient_tol) { termination = 0; break; }
    } else {
      radius /= decrease_factor; decrease_factor *= 2; need_linearize = true;
    }
  }
done:
  ora_ba_read(S, poses7, points3);
  ora_ba_close(S);
  summary5[0] = iterations; summary5[1] = successful; summary5[2] = termination;
  summary5[3] = initial_cost; summary5[4] = cost;
  return 0;
}
