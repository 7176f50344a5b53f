// Drop-in replacement of the reference's src/reprojection_factor.hpp:7-18: same class name, constructor and Evaluate
// signature (Ceres' conventions: `jacobians` may be null, each jacobians[i] may be null; 2x7 / 2x3 row-major).
// The arithmetic runs in libsvo_hip.so (svo_reproj_eval -> the a11 kernel).  The reference derives from
// ceres::SizedCostFunction<2, 7, 3>; a build that still links Ceres defines SVO_ADAPTER_WITH_CERES to keep that base.
#ifndef REPROJECTION_FACTOR_H_
#define REPROJECTION_FACTOR_H_

#ifdef SVO_ADAPTER_WITH_CERES
#include <ceres/ceres.h>
#endif
#include "camera_info.hpp"
#include "svo_adapter.hpp"

class ReprojectionFactor
#ifdef SVO_ADAPTER_WITH_CERES
    : public ceres::SizedCostFunction<2, 7, 3>
#endif
{
 public:
  ReprojectionFactor(double ox, double oy, CameraInfo info);
  virtual ~ReprojectionFactor() {}

  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const;

 private:
  double obs[2];
  CameraInfo camera_info;
};

#endif
