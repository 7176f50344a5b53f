// Ceres' step acceptance and trust-region radius update (TrustRegionMinimizer / TrustRegionStepEvaluator
// of ceres::Solve, src/bundle_adjuster.cpp:140) as ONE function of the summed payload2 — shared by the host step control
// (host/lm.cpp) and the kernels that take the decision where the sums live (csrc/ba.hip), so that pass A of the next
// LM iteration can be enqueued behind pass B without a host round trip.  Plain IEEE double arithmetic, no FMA
// contraction on either side: host and device produce the same bits.
#ifndef SVO_LM_DECIDE_H_
#define SVO_LM_DECIDE_H_

#if defined(__HIPCC__)
#define SVO_HD __host__ __device__
#else
#define SVO_HD
#endif

struct SvoLmDecision {
  int accept;          // 1: the candidate becomes the current point; 0: it is rejected (or the step was invalid)
  double next_radius;  // trust-region radius of the next linearisation
};

// cost: cost at the current point; mcc: pose part of the model cost change; cost_new = payload2[0];
// model_change_points = payload2[1].
SVO_HD inline SvoLmDecision svo_lm_decide(double cost, double mcc, double radius, double decrease_factor, double cost_new,
                                          double model_change_points) {
  const double kMaxRadius = 1e16, kMinRelativeDecrease = 1e-3;
  SvoLmDecision d;
  const double model_change = mcc + model_change_points;
  d.accept = 0;
  d.next_radius = radius / decrease_factor;  // invalid step (no model decrease) or rejected step
  if (model_change > 0) {
    const double rho = (cost - cost_new) / model_change;
    if (rho > kMinRelativeDecrease) {
      const double t = 2.0 * rho - 1.0;
      const double shrink = 1.0 - t * t * t;
      double r = radius / (shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0);
      if (r > kMaxRadius) r = kMaxRadius;
      d.accept = 1;
      d.next_radius = r;
    }
  }
  return d;
}

#endif
