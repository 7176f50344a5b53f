// Step control of the bundle adjustment: the trust-region Levenberg-Marquardt loop that ceres::Solve runs for
// BundleAdjuster::bundle_adjust (reference src/bundle_adjuster.cpp:137-157, options :9-12; semantics SURVEY.md
// Appendix B) — Jacobi column scaling fixed at the first Jacobian, LM diagonal clamp(diag, 1e-6, 1e32) / radius, dense
// Cholesky of the reduced camera system, Ceres' acceptance test / radius update / tolerances.  Host code, one copy: the
// HIP adjuster (csrc/ba.hip) plugs its kernels in through svo_lm_ops; every rank of a sharded run executes this loop on
// the same all-reduced payloads and therefore takes identical decisions.
//
// One host round trip per LM iteration.  An iteration needs two sums over all observations: the candidate's cost
// (pass B, after the pose step is known) and the next linearisation (pass A, whose radius and point depend on pass B's
// sums through Ceres' accept / radius rule).  The backend is asked to produce both in one `step` call:
//   chained   — the accept / radius decision is a closed-form function of the summed payload2 (svo_lm_decide,
//               lm_decide.h), so the backend evaluates it where the sums live (a device kernel behind the reduction /
//               the tiny all-reduce) and enqueues pass A for the outcome right behind pass B: at the candidate with the
//               new radius, or at the current point with the reduced radius after a rejection.  No host in between.
//   same-sweep — Ceres' update  r / max(1/3, 1 - (2 rho - 1)^3)  is exactly  r / (1/3)  for every rho >= 0.9368 (the
//               saturated regime of a converging solve); when the last accepted step was saturated, pass A at the
//               candidate runs with that radius in the SAME sweep as pass B and both payloads share ONE collective.
// This loop re-derives every decision itself from the returned payload2 and uses the returned linearisation only if
// its (point, radius) is exactly what it needs; otherwise pass A runs again.  The arithmetic that decides anything is
// the same either way, so results do not depend on whether a prediction holds (bit for bit in the deterministic
// accumulation mode).  SVO_LM_NO_SPECULATION=1 turns both mechanisms off (two round trips per iteration).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "lm_decide.h"
#include "lm_math.h"
#include "svo.h"

bool svo_host_cholesky_solve(double* A, double* b, int n);  // host/linalg.cpp

namespace {
constexpr double MIN_DIAG = 1e-6, MAX_DIAG = 1e32, MAX_RADIUS = 1e16, MIN_RADIUS = 1e-32;

}  // namespace

extern "C" int svo_lm_decide_step(double cost, double mcc, double radius, double decrease_factor, double cost_new,
                                  double model_change_points, int* accept, double* next_radius) {
  if (!accept || !next_radius) return SVO_ERR_INVALID;
  const SvoLmDecision d = svo_lm_decide(cost, mcc, radius, decrease_factor, cost_new, model_change_points);
  *accept = d.accept;
  *next_radius = d.next_radius;
  return SVO_OK;
}

extern "C" int svo_lm_solve(int n_poses, double* poses7, const svo_lm_ops* ops, const svo_ba_options* opt_in,
                            svo_ba_summary* sum, svo_lm_stats* stats) {
  if (n_poses < 1 || !poses7 || !ops || !ops->linearize || !ops->step || !ops->accept) return SVO_ERR_INVALID;
  svo_ba_options opt;
  if (opt_in) opt = *opt_in; else svo_ba_default_options(&opt);
  const int K = n_poses, n = 6 * (K - 1);
  const size_t pay1 = (size_t)n * n + 3 * (size_t)n + 2;
  const auto t_begin = std::chrono::steady_clock::now();
  // two payload1 buffers: the linearisation in use and the speculative one of the step in flight
  std::vector<double> bufA(pay1), bufB(pay1), sc(n, 0.0), Sm((size_t)n * n), rhs(n), Df(n), dc(n > 0 ? n : 1),
      cand((size_t)7 * K);
  double* cur = bufA.data();
  double* spec = bufB.data();
  double pay2[4];
  double radius = opt.initial_radius, decrease_factor = 2.0;
  int iterations = 0, successful = 0, termination = 1;
  svo_lm_stats st;
  memset(&st, 0, sizeof(st));
  const bool allow_spec = !getenv("SVO_LM_NO_SPECULATION");
  const bool trace = getenv("SVO_BA_TRACE") != nullptr;
  bool saturated = false;  // the last accepted step had rho >= 0.9368: predict the same for the next one
  int rc = 0;
  auto gradient_norm = [&](const double* pay) {
    const double* gc = pay + (size_t)n * n + n;
    double g2 = pay[pay1 - 1];
    for (int a = 0; a < n; ++a) g2 += gc[a] * gc[a];
    return sqrt(g2);
  };
  auto finish = [&](double initial_cost, double cost) {
    if (sum) {
      sum->iterations = iterations; sum->successful_steps = successful; sum->termination = termination;
      sum->initial_cost = initial_cost; sum->final_cost = cost;
      sum->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    // all-reduces the loop issued (or would issue on N ranks): a stand-alone pass A = 1, a same-sweep step = 1 (both payloads in one
    // buffer), a chained step = 2 (payload2, decision, payload1), a plain step = 1
    st.collectives = st.linearize_calls + st.single_exchange + 2 * (st.speculations - st.single_exchange) + (st.step_calls - st.speculations);
    if (stats) *stats = st;
  };

  ++st.linearize_calls;
  if ((rc = ops->linearize(ops->user, radius, /*first*/ 1, cur))) return rc;
  double cost = cur[pay1 - 2];
  const double initial_cost = cost;
  {
    const double* dU = cur + (size_t)n * n + 2 * (size_t)n;
    for (int a = 0; a < n; ++a) sc[a] = 1.0 / (1.0 + sqrt(dU[a]));
  }
  bool need_linearize = false;
  if (gradient_norm(cur) <= opt.gradient_tolerance) {
    termination = 0;
  } else {
    while (true) {
      if (iterations >= opt.max_iterations) { termination = 1; break; }
      if (opt.max_time_s > 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() >= opt.max_time_s) {
        termination = 1; break;  // src/bundle_adjuster.cpp:11 (wall clock; disabled for reproducible runs)
      }
      if (radius <= MIN_RADIUS) { termination = 0; break; }
      ++iterations;
      if (need_linearize) {
        ++st.linearize_calls;
        if ((rc = ops->linearize(ops->user, radius, 0, cur))) return rc;
        need_linearize = false;
      }
      const double* S = cur;
      const double* gred = S + (size_t)n * n;
      const double* gc = gred + n;
      const double* dU = gc + n;
      for (int a = 0; a < n; ++a) {
        Df[a] = std::min(std::max(dU[a] * sc[a] * sc[a], MIN_DIAG), MAX_DIAG) / radius;
        for (int b = 0; b <= a; ++b) Sm[(size_t)a * n + b] = S[(size_t)a * n + b] * sc[a] * sc[b];  // the factorisation reads the lower triangle only
        Sm[(size_t)a * n + a] += Df[a];
        rhs[a] = -(gred[a] + gc[a]) * sc[a];  // the backend accumulates only the -Y g_p part of the reduced gradient
      }
      const bool ok = n == 0 || svo_host_cholesky_solve(Sm.data(), rhs.data(), n);
      bool step_ok = false;
      double cost_new = 0, model_change = 0, step2 = 0, x2 = 0, next_radius = 0, mcc = 0;
      int next_at_candidate = 0;
      svo_lm_step_ctl ctl;
      memset(&ctl, 0, sizeof(ctl));
      if (ok) {
        for (int a = 0; a < n; ++a) {
          mcc += 0.5 * rhs[a] * (Df[a] * rhs[a] - gc[a] * sc[a]);
          dc[a] = rhs[a] * sc[a];
        }
        for (int k = 0; k < K; ++k) {
          if (k == 0) memcpy(&cand[0], &poses7[0], 7 * sizeof(double));
          else svo_plus_pose(&poses7[7 * k], &dc[6 * (k - 1)], &cand[7 * k]);
        }
        ctl.cost = cost; ctl.mcc = mcc; ctl.decrease_factor = decrease_factor;
        if (allow_spec && iterations < opt.max_iterations) {  // the last iteration cannot use a new linearisation
          // the radius an accepted step with rho >= 0.9368 produces
          if (saturated) ctl.spec_radius = std::min(MAX_RADIUS, radius / (1.0 / 3.0));
          else ctl.chain = 1;
        }
        ++st.step_calls;
        if (ctl.spec_radius > 0 || ctl.chain) ++st.speculations;
        if (ctl.spec_radius > 0) ++st.single_exchange;
        if ((rc = ops->step(ops->user, dc.data(), cand.data(), radius, &ctl, pay2, spec, &next_radius, &next_at_candidate))) return rc;
        cost_new = pay2[0];
        model_change = mcc + pay2[1];
        step2 = pay2[2]; x2 = pay2[3];
        for (int k = 1; k < K; ++k)
          for (int a = 0; a < 7; ++a) {
            const double dd = cand[7 * k + a] - poses7[7 * k + a];
            step2 += dd * dd;
            x2 += poses7[7 * k + a] * poses7[7 * k + a];
          }
        step_ok = model_change > 0;
      }
      // the backend's linearisation is usable iff it was taken at the point and with the radius this loop arrives at
      auto take_next = [&](bool at_candidate, double want_radius) {
        if (!(next_radius > 0) || next_radius != want_radius || (next_at_candidate != 0) != at_candidate) return false;
        std::swap(cur, spec);
        ++st.speculation_hits;
        return true;
      };
      if (!step_ok) {  // invalid step (not positive definite, or no model decrease)
        radius /= decrease_factor; decrease_factor *= 2; saturated = false;
        need_linearize = !(ok && take_next(false, radius));
        continue;
      }
      if (sqrt(step2) <= opt.parameter_tolerance * (sqrt(x2) + opt.parameter_tolerance)) { termination = 0; break; }
      const double cost_change = cost - cost_new;
      if (fabs(cost_change) <= opt.function_tolerance * cost) {  // Ceres returns before the step is taken
        termination = 0;
        break;
      }
      const SvoLmDecision dec = svo_lm_decide(cost, mcc, radius, decrease_factor, cost_new, pay2[1]);
      if (trace)
        fprintf(stderr, "[lm] it %d cost %.17g new %.17g model %.17g rho %.6g radius %.6g mode %s\n", iterations, cost, cost_new,
                model_change, cost_change / model_change, radius, ctl.spec_radius > 0 ? "same-sweep" : ctl.chain ? "chained" : "plain");
      if (dec.accept) {
        if ((rc = ops->accept(ops->user))) return rc;
        memcpy(poses7, cand.data(), sizeof(double) * 7 * K);
        cost = cost_new;
        ++successful;
        saturated = dec.next_radius == std::min(MAX_RADIUS, radius / (1.0 / 3.0));
        radius = dec.next_radius;
        decrease_factor = 2.0;
        if (!take_next(true, radius)) {
          ++st.linearize_calls;
          if ((rc = ops->linearize(ops->user, radius, 0, cur))) return rc;
        }
        if (gradient_norm(cur) <= opt.gradient_tolerance) { termination = 0; break; }
      } else {
        radius = dec.next_radius; decrease_factor *= 2; saturated = false;
        need_linearize = !take_next(false, radius);
      }
    }
  }
  finish(initial_cost, cost);
  return SVO_OK;
}
