// ReprojectionFactor over the C-ABI.  Replaces src/reprojection_factor.cpp of the reference.
#include "reprojection_factor.hpp"

ReprojectionFactor::ReprojectionFactor(double ox, double oy, CameraInfo info) : camera_info(info) {
  obs[0] = ox;
  obs[1] = oy;
}

bool ReprojectionFactor::Evaluate(double const *const *parameters, double *residuals, double **jacobians) const {
  svo_ctx *ctx = svo_adapter::context();
  if (!ctx || !parameters || !residuals) return false;
  // parameters[0] = pose [qw qx qy qz tx ty tz], parameters[1] = landmark xyz (src/reprojection_factor.cpp:13-22)
  return svo_reproj_eval(ctx, 1, parameters[0], parameters[1], obs, camera_info.focal, camera_info.cx, camera_info.cy, residuals,
                         jacobians ? jacobians[0] : nullptr, jacobians ? jacobians[1] : nullptr) == SVO_OK;
}
