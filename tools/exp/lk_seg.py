import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import bench
import stereo_vo_amd as S
ctx = S.Context(bench.W, bench.H, max_batch=16, max_corners=bench.MAXC, max_candidates=1 << 16, max_features=bench.MAX_FEAT)
p, L, R = bench.render_batch(S, 0x5EED0001, 16)
pp = S.pipeline_default_params()
pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
pp.width, pp.height = bench.W, bench.H
pp.max_corners, pp.quality, pp.min_feature_distance = bench.MAXC, bench.QUALITY, bench.MIN_DIST
pp.max_features, pp.window_size, pp.ba_max_iterations, pp.ba_max_time_s = bench.MAX_FEAT, bench.WINDOW, 50, 0.0
g = S.Pipeline(ctx, pp)
g.process_batch(L, R)
out = np.zeros(8, np.uint64)
S.api.lib().svo_lk_seg_dump(out.ctypes.data_as(C.c_void_p))
n = max(int(out[0]), 1)
print("iterations", n, "cycles per iteration: setup+weights(+restage) %.0f  lds read+unpack %.0f  pixels %.0f  reductions %.0f  update %.0f" %
      tuple(float(out[i]) / n for i in range(1, 6)))
