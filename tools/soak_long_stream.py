#!/usr/bin/env python3
"""BASELINE configs[2]-shaped soak (run on the GPU box): a long KITTI-shaped synthetic stereo stream through ONE
pipeline with a 10-keyframe window — ids, the landmark store and every sequence counter keep growing, nothing is reset.
Reports frames/s (rendering excluded), keyframes, live features and the keyframe ATE against the generator's poses."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime for the process)
import stereo_vo_amd as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
B = 16
W, H = 1241, 376
ctx = S.Context(W, H, max_batch=B, max_corners=1500, max_candidates=1 << 16, max_features=2000)
p = S.synth_default(W, H)
pp = S.pipeline_default_params()
pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
pp.width, pp.height = W, H
pp.max_corners, pp.quality, pp.min_feature_distance, pp.max_features, pp.window_size = 1500, 0.02, 10.0, 2000, 10
pp.ba_max_time_s = 0.0
pipe = S.Pipeline(ctx, pp)
t_gpu, n_kf, est, gt, tracked = 0.0, 0, [], [], []
for f0 in range(0, N, B):
    fr = [S.synth_render(p, f0 + i) for i in range(min(B, N - f0))]
    L, R = np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])
    t0 = time.perf_counter()
    res = pipe.process_batch(L, R)
    t_gpu += time.perf_counter() - t0
    for i, r in enumerate(res):
        n_kf += r.is_keyframe
        tracked.append(r.n_tracked)
        if r.is_keyframe and r.pose7[0] != 0:
            w, x, y, z = r.pose7[:4]
            t = np.array(r.pose7[4:])
            Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                           [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                           [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            est.append(-Rm.T @ t)
            gt.append(S.synth_pose(p, f0 + i)[:, 3])
    if (f0 // B) % 10 == 0:
        print(f"frame {f0 + len(res)}/{N} keyframes {n_kf} mean tracked {np.mean(tracked[-160:]):.0f}", flush=True)
ate = S.api.ate_rmse(np.array(est), np.array(gt), False)
path = float(np.linalg.norm(np.diff(np.array(gt), axis=0), axis=1).sum())
print(f"frames {N}  gpu {t_gpu:.2f} s  {N / t_gpu:.0f} frames/s (host-pointer entry, PCIe upload included)  keyframes {n_kf}  "
      f"ATE at keyframes {ate:.3f} m over {path:.0f} m ({100 * ate / max(path, 1e-9):.2f} %)")
