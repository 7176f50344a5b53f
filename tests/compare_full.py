#!/usr/bin/env python3
"""Frame-by-frame comparison of the HIP pipeline and the oracle pipeline on the bench workload (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import oracle_lib as O  # noqa: E402
import stereo_vo_amd as S  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = S.Context(bench.W, bench.H, max_batch=B, max_corners=bench.MAXC, max_candidates=1 << 16, max_features=bench.MAX_FEAT)
p, L, R = bench.render_batch(S, 0x5EED0001, B)
pp = S.pipeline_default_params()
pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
pp.width, pp.height = bench.W, bench.H
pp.max_corners, pp.quality, pp.min_feature_distance = bench.MAXC, bench.QUALITY, bench.MIN_DIST
pp.max_features, pp.window_size, pp.ba_max_iterations, pp.ba_max_time_s = bench.MAX_FEAT, bench.WINDOW, 50, 0.0
g = S.Pipeline(ctx, pp)
o = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=bench.W, height=bench.H, max_corners=bench.MAXC,
               quality=bench.QUALITY, min_feature_distance=bench.MIN_DIST, parallax_thresh=20.0, window_size=bench.WINDOW,
               max_features=bench.MAX_FEAT, ba_max_iterations=50, num_threads=8)
key = lambda x: (x.n_detected, x.n_tracked, x.n_inliers, x.n_new, x.is_keyframe, x.ba_iterations)
for i in range(B):
    rg = g.process_batch(L[i:i + 1], R[i:i + 1])[0]
    ro = o.process(L[i], R[i])
    ig, xg = g.tracked()
    io, xo = o.tracked()
    same_ids = np.array_equal(ig, io) and np.array_equal(xg.view(np.uint32), xo.view(np.uint32))
    dp = np.abs(np.array(list(rg.pose7)) - np.array(list(ro.pose7))).max()
    print(i, key(rg), key(ro), "ids_same" if same_ids else "IDS DIFFER", "pose diff %.2e" % dp, flush=True)
