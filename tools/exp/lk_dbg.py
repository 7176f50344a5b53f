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
lib = S.api.lib()
for i in range(16):
    r = g.process_batch(L[i:i + 1], R[i:i + 1])[0]
    n = 731
    out = np.zeros(3 * n, np.uint32)
    lib.svo_lk_dbg_dump(out.ctypes.data_as(C.c_void_p), n)
    o = out.reshape(n, 3)
    it, st, tk = o[:, 0], o[:, 1], o[:, 2] * 0.01
    if i == 0: continue
    q = lambda a, x: float(np.percentile(a, x))
    print(f"frame {i}: tracked {r.n_tracked} iterations mean {it.mean():.0f} p50 {q(it,50):.0f} p90 {q(it,90):.0f} max {it.max()}   restages mean {st.mean():.1f} max {st.max()}   "
          f"us mean {tk.mean():.0f} p50 {q(tk,50):.0f} p90 {q(tk,90):.0f} max {tk.max():.0f}   us/iter of the slowest {tk.max()/max(it[tk.argmax()],1):.2f}")
