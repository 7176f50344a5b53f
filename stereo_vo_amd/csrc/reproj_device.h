// Device-side ReprojectionFactor (reference: src/reprojection_factor.cpp:10-88).
// gamma = R(q) p / |q|^2 + t with |q|^2 R(q) = v v^T + (w I + [v]x)^2  (:24-33);
// r = [f 0 cx; 0 f cy] gamma / gamma_z - obs (:35-38).  Jacobians by the chain rule through gamma
// (algebraically identical to the reference's expanded scalar expressions at :63-83, including the
// 1/|q|^2 factor).  Layouts: 2x7 row-major [qw qx qy qz tx ty tz] with entries 5, 11 zero; 2x3 row-major.
#ifndef SVO_REPROJ_DEVICE_H_
#define SVO_REPROJ_DEVICE_H_
#include <hip/hip_runtime.h>

struct D3 { double x, y, z; };
#define SVO_RD(name) name
#include "reproj_device_body.h"
#undef SVO_RD
#define SVO_RD(name) name##_c
#pragma clang fp contract(fast)
#include "reproj_device_body.h"
#pragma clang fp contract(off)
#undef SVO_RD
#endif
