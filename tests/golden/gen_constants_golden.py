#!/usr/bin/env python3
"""Generate tests/golden/constants_golden.json — every first-party literal the hot path depends on, extracted by
regular expressions from the reference's source TEXT, plus triangulation known answers built from the extracted
Q-matrix assignments.

Runs ONLY in the build container (needs /root/reference as text).  The output holds numbers only (names, values,
inputs and expected outputs) — no reference text.  tests/test_constants.py asserts that the product
(svo_reference_constants / svo_pipeline_default_params), and the CPU oracle agree with it.

Literals and where they sit in the reference:
  src/image_processor.cpp:22,23   goodFeaturesToTrack(.., 300, 0.1, ..); fewer than 4 corners -> skip
  src/image_processor.cpp:63      keyframe gate: percent_lost < 0.4
  src/image_processor.cpp:80      solvePnPRansac(.., true, 100, 8.0, 0.99, ..)
  src/image_processor.cpp:174-176 StereoBM::create(16*3, 21); convertTo(CV_32F, 1.0/16)
  src/image_processor.cpp:184-189 the six Q.at<float>(i, j) = ... assignments
  src/feature_tracker.cpp:24-26   Size(21, 21), 3, TermCriteria(.., 30, 0.01), 0, 1e-2
  src/feature_tracker.cpp:47,53   forward/backward distance < 2; parallax > 200
  src/vo_node.cpp:33-36           parallax_thresh 20, min_feature_distance 30, image_queue_size 5, sliding_window_size 5
  src/bundle_adjuster.hpp:75      max_features 400
  src/bundle_adjuster.cpp:9-12    DENSE_SCHUR, max_solver_time_in_seconds 0.1, num_threads 4
"""
import json
import os
import random
import re
import sys

import numpy as np

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "constants_golden.json")


def one(pattern, text, what, flags=re.S):
    m = re.search(pattern, text, flags)
    if not m:
        sys.exit(f"pattern for {what} not found")
    return m.groups()


def num(s):
    return float(eval(s, {"__builtins__": {}}, {}))  # plain arithmetic literals such as 16*3 or 1.0/16


def main():
    ip = open(os.path.join(REF, "image_processor.cpp")).read()
    ft = open(os.path.join(REF, "feature_tracker.cpp")).read()
    vo = open(os.path.join(REF, "vo_node.cpp")).read()
    bh = open(os.path.join(REF, "bundle_adjuster.hpp")).read()
    bc = open(os.path.join(REF, "bundle_adjuster.cpp")).read()
    c = {}
    g = one(r"goodFeaturesToTrack\(\s*stereo_pair\.left\s*,\s*detected_features\s*,\s*([^,]+),\s*([^,]+),\s*min_feature_distance\s*\)", ip, "gftt")
    c["gftt_max_corners"], c["gftt_quality"] = num(g[0]), num(g[1])
    c["min_detected"] = num(one(r"detected_features\.size\(\)\s*<\s*(\d+)", ip, "min detected")[0])
    c["keyframe_percent_lost"] = num(one(r"percent_lost\s*<\s*([0-9.]+)", ip, "percent lost")[0])
    g = one(r"rvec\s*,\s*tvec\s*,\s*(true|false)\s*,\s*([^,]+),\s*([^,]+),\s*([^,]+),\s*inlier_indices", ip, "pnp")
    c["pnp_use_extrinsic_guess"] = 1.0 if g[0] == "true" else 0.0
    c["pnp_iterations"], c["pnp_reproj_error"], c["pnp_confidence"] = num(g[1]), num(g[2]), num(g[3])
    g = one(r"StereoBM::create\(\s*([^,]+),\s*([^)]+)\)", ip, "stereobm")
    c["stereo_num_disparities"], c["stereo_block_size"] = num(g[0]), num(g[1])
    c["stereo_disparity_scale"] = num(one(r"convertTo\(\s*disparity\s*,\s*CV_32F\s*,\s*([^)]+)\)", ip, "disparity scale")[0])
    c["triangulate_min_disparity_exclusive"] = num(one(r"if\s*\(\s*disp\s*>\s*([0-9.]+)\s*\)", ip, "disp > 0")[0])
    g = one(r"cv::Size\(\s*(\d+)\s*,\s*(\d+)\s*\)\s*,\s*(\d+)\s*,\s*cv::TermCriteria\([^,]+,\s*([^,]+),\s*([^)]+)\)\s*,\s*(\d+)\s*,\s*([^)]+)\)", ft, "LK call")
    c["lk_win_w"], c["lk_win_h"], c["lk_max_level"] = num(g[0]), num(g[1]), num(g[2])
    c["lk_max_iterations"], c["lk_epsilon"], c["lk_flags"], c["lk_min_eig_threshold"] = num(g[3]), num(g[4]), num(g[5]), num(g[6])
    assert len(re.findall(r"calcOpticalFlowPyrLK", ft)) == 2  # forward and backward calls carry the same arguments
    calls = re.findall(r"cv::Size\(\s*\d+\s*,\s*\d+\s*\)\s*,\s*\d+\s*,\s*cv::TermCriteria\([^)]*\)\s*,\s*\d+\s*,\s*[^)]+\)", ft)
    assert len(calls) == 2 and len({re.sub(r"\s+", "", x) for x in calls}) == 1, calls
    c["fb_max_distance"] = num(one(r"reverse_track\[i\]\s*\)\s*<\s*([0-9.]+)", ft, "fb distance")[0])
    c["max_parallax"] = num(one(r"parallax\s*>\s*([0-9.]+)", ft, "max parallax")[0])
    c["draw_thickness"] = num(one(r"CV_RGB\(0,\s*255,\s*0\)\s*,\s*(\d+)\)", ft, "arrow thickness")[0])
    for name in ("parallax_thresh", "min_feature_distance", "image_queue_size", "sliding_window_size"):
        c[name] = num(one(r"static const \w+ " + name + r"\s*=\s*([0-9.]+)\s*;", vo, name)[0])
    c["max_features"] = num(one(r"static const size_t max_features\s*=\s*(\d+)\s*;", bh, "max_features")[0])
    c["ba_max_solver_time_s"] = num(one(r"max_solver_time_in_seconds\s*=\s*([0-9.]+)\s*;", bc, "solver time")[0])
    c["ba_num_threads"] = num(one(r"options\.num_threads\s*=\s*(\d+)\s*;", bc, "threads")[0])
    solver = one(r"linear_solver_type\s*=\s*ceres::(\w+)\s*;", bc, "solver")[0]
    assert solver == "DENSE_SCHUR", solver
    c["ba_linear_solver_is_dense_schur"] = 1.0

    # ---- Q (src/image_processor.cpp:183-189): Q = zeros(4,4,CV_32F) then six assignments, evaluated as the C++ does
    # (double arithmetic on float operands, stored as float)
    qa = re.findall(r"Q\.at<float>\(\s*(\d)\s*,\s*(\d)\s*\)\s*=\s*([^;]+);", ip)
    assert len(qa) == 6, qa

    def q_matrix(focal, cx, cy, baseline):
        env = dict(focal=float(np.float32(focal)), cx=float(np.float32(cx)), cy=float(np.float32(cy)), baseline=float(np.float32(baseline)))
        Q = np.zeros((4, 4), np.float32)
        for i, j, e in qa:
            Q[int(i), int(j)] = np.float32(eval(e, {"__builtins__": {}}, env))
        return Q
    cams = [(718.856, 607.1928, 185.2157, 0.537165718864418), (385.7544860839844, 323.1204833984375, 236.7432098388672, 0.05)]
    rng = random.Random(0x5EED0C05)
    kat = []
    for ci, cam in enumerate(cams):
        Q = q_matrix(*cam)
        for t in range(8):
            # camera -> world pose: identity for the first case, then a rigid motion (src/image_processor.cpp:39,130-134)
            if t == 0:
                P = np.eye(4, dtype=np.float32)
            else:
                a = rng.uniform(-0.3, 0.3)
                P = np.eye(4, dtype=np.float32)
                P[0, 0], P[0, 2], P[2, 0], P[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
                P[:3, 3] = [rng.uniform(-3, 3), rng.uniform(-0.5, 0.5), rng.uniform(-10, 10)]
            x, y = float(rng.randrange(60, 1200)), float(rng.randrange(12, 360))
            d = rng.randrange(1 * 16, 47 * 16) / 16.0  # a StereoBM disparity: multiple of 1/16
            # world_point = camera_pose * Q * [x y d 1]^T, de-homogenised (:195-205); evaluated here in double
            h = P.astype(np.float64) @ (Q.astype(np.float64) @ np.array([x, y, d, 1.0]))
            kat.append(dict(camera=ci, pose16=[float(v) for v in P.reshape(-1)], x=x, y=y, disp=d,
                            xyz=[h[0] / h[3], h[1] / h[3], h[2] / h[3]]))
    out = dict(source="regex extraction from the reference's src/*.cpp|hpp text (see the generator's docstring for file:line)",
               constants=c,
               q=[dict(focal=cam[0], cx=cam[1], cy=cam[2], baseline=cam[3], Q=[float(v) for v in q_matrix(*cam).reshape(-1)]) for cam in cams],
               triangulation=kat)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, len(c), "constants,", len(kat), "triangulation cases")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixture is generated in the build container only")
    main()
