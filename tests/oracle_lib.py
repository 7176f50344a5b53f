"""ctypes loader for the CPU oracle (oracle/).  TEST INFRASTRUCTURE: only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class OraPipelineParams(C.Structure):
    _fields_ = [("focal", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("baseline", C.c_double),
                ("width", C.c_int), ("height", C.c_int), ("max_corners", C.c_int), ("quality", C.c_double),
                ("min_feature_distance", C.c_float), ("parallax_thresh", C.c_float),
                ("window_size", C.c_int), ("max_features", C.c_int), ("ba_max_iterations", C.c_int),
                ("num_threads", C.c_int)]


class OraFrameResult(C.Structure):
    _fields_ = [("n_detected", C.c_int), ("n_tracked", C.c_int), ("n_inliers", C.c_int),
                ("n_new", C.c_int), ("is_keyframe", C.c_int), ("av_parallax", C.c_float),
                ("percent_lost", C.c_float), ("pose7", C.c_double * 7), ("ba_iterations", C.c_int)]


ORA_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_size_t, C.c_void_p)


def oracle():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "libsvo_oracle.so")
        if not os.path.exists(path):
            import importlib.util
            spec = importlib.util.spec_from_file_location("oracle_build", os.path.join(ROOT, "oracle", "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build_oracle()
        L = C.CDLL(path)
        L.ora_pyramid_bytes.restype = C.c_size_t
        L.ora_pipeline_create.restype = C.c_void_p
        L.ora_pipeline_destroy.argtypes = [C.c_void_p]
        L.ora_pipeline_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_pipeline_get_tracked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def reproj_eval(pose7, point3, obs2, focal, cx, cy, want_jpose=True, want_jpoint=True):
    pose7, point3, obs2 = _f64(pose7), _f64(point3), _f64(obs2)
    n = pose7.shape[0]
    r = np.empty((n, 2))
    jq = np.empty((n, 14)) if want_jpose else None
    jx = np.empty((n, 6)) if want_jpoint else None
    oracle().ora_reproj_eval(n, _p(pose7), _p(point3), _p(obs2), C.c_double(focal), C.c_double(cx),
                             C.c_double(cy), _p(r), _p(jq), _p(jx))
    return r, jq, jx


def corner_response(img):
    img = _u8(img)
    h, w = img.shape
    eig = np.empty((h, w), np.float32)
    oracle().ora_corner_response(_p(img), w, h, w, _p(eig))
    return eig


def corner_detect(img, max_corners=300, quality=0.1, min_distance=30.0, return_ncand=False):
    img = _u8(img)
    h, w = img.shape
    xy = np.empty((max_corners, 2), np.float32)
    nc = C.c_int(0)
    n = oracle().ora_corner_detect(_p(img), w, h, w, max_corners, C.c_double(quality),
                                   C.c_double(min_distance), _p(xy), C.byref(nc))
    if return_ncand:
        return xy[:n].copy(), nc.value
    return xy[:n].copy()


def stereo_prefilter(img, cap=31):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    oracle().ora_stereo_prefilter(_p(img), w, h, w, cap, _p(out))
    return out


def stereo_bm(left, right, ndisp=48, block=21):
    left, right = _u8(left), _u8(right)
    h, w = left.shape
    d = np.empty((h, w), np.int16)
    oracle().ora_stereo_bm(_p(left), _p(right), w, h, w, ndisp, block, _p(d))
    return d


def stereo_disparity_at(left, right, xy, ndisp=48, block=21):
    left, right, xy = _u8(left), _u8(right), _f32(xy)
    h, w = left.shape
    n = xy.shape[0]
    d = np.empty(n, np.float32)
    oracle().ora_stereo_disparity_at(_p(left), _p(right), w, h, w, ndisp, block, _p(xy), n, _p(d))
    return d


def triangulate(xy, disp, pose16, focal, cx, cy, baseline):
    xy, disp, pose16 = _f32(xy), _f32(disp), _f32(pose16)
    n = xy.shape[0]
    kxy = np.empty((n, 2), np.float32)
    xyz = np.empty((n, 3), np.float32)
    kidx = np.empty(n, np.int32)
    m = oracle().ora_triangulate(_p(xy), _p(disp), n, _p(pose16), C.c_float(focal), C.c_float(cx),
                                 C.c_float(cy), C.c_float(baseline), _p(kxy), _p(xyz), _p(kidx))
    return kxy[:m].copy(), xyz[:m].copy(), kidx[:m].copy()


def build_pyramid(img, levels=4):
    img = _u8(img)
    h, w = img.shape
    L = oracle()
    total = L.ora_pyramid_bytes(w, h, levels)
    buf = np.empty(total, np.uint8)
    L.ora_build_pyramid(_p(img), w, h, w, levels, _p(buf))
    out, off = [], 0
    lw, lh = w, h
    for _ in range(levels):
        out.append(buf[off:off + lw * lh].reshape(lh, lw).copy())
        off += lw * lh
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
    return out


def lk_track(prev, nxt, xy):
    prev, nxt, xy = _u8(prev), _u8(nxt), _f32(xy)
    h, w = prev.shape
    n = xy.shape[0]
    out = np.empty((n, 2), np.float32)
    st = np.empty(n, np.uint8)
    oracle().ora_lk_track(_p(prev), _p(nxt), w, h, w, _p(xy), n, _p(out), _p(st))
    return out, st


def track_features(prev, nxt, xy, initial_xy):
    prev, nxt, xy, initial_xy = _u8(prev), _u8(nxt), _f32(xy), _f32(initial_xy)
    h, w = prev.shape
    n = xy.shape[0]
    kxy = np.empty((n, 2), np.float32)
    kidx = np.empty(n, np.int32)
    av = C.c_float(0)
    m = oracle().ora_track_features(_p(prev), _p(nxt), w, h, w, _p(xy), _p(initial_xy), n, _p(kxy),
                                    _p(kidx), C.byref(av))
    return kxy[:m].copy(), kidx[:m].copy(), av.value


def dedup(det, trk, min_distance):
    det, trk = _f32(det), _f32(trk)
    out = np.empty_like(det)
    m = oracle().ora_dedup(_p(det), det.shape[0], _p(trk), trk.shape[0], C.c_float(min_distance), _p(out))
    return out[:m].copy()


def pnp_ransac(xyz, xy, focal, cx, cy, rvec, tvec, iterations=100, reproj_err=8.0, confidence=0.99):
    xyz, xy = _f32(xyz), _f32(xy)
    n = xyz.shape[0]
    rv, tv = _f64(rvec).copy(), _f64(tvec).copy()
    inl = np.empty(max(n, 1), np.int32)
    m = oracle().ora_pnp_ransac(_p(xyz), _p(xy), n, C.c_float(focal), C.c_float(cx), C.c_float(cy),
                                _p(rv), _p(tv), iterations, C.c_float(reproj_err), C.c_double(confidence),
                                _p(inl))
    return rv, tv, inl[:m].copy()


def det_atan2_q1(y, x):
    oracle().ora_det_atan2_q1.restype = C.c_double
    return float(oracle().ora_det_atan2_q1(C.c_double(y), C.c_double(x)))


def det_sincos(x):
    s, c = C.c_double(0), C.c_double(0)
    oracle().ora_det_sincos(C.c_double(x), C.byref(s), C.byref(c))
    return s.value, c.value


def det_rvec_quat_roundtrip(rvec):
    rv = _f64(rvec).copy()
    q, back = np.zeros(4), np.zeros(3)
    oracle().ora_det_rvec_quat_roundtrip(_p(rv), _p(q), _p(back))
    return q, back


def pnp_update_num_iters(p, ep, model_points, max_iters):
    return int(oracle().ora_pnp_update_num_iters(C.c_double(p), C.c_double(ep), int(model_points), int(max_iters)))


def pnp_det_log(x):
    f = oracle().ora_pnp_det_log
    f.restype = C.c_double
    return float(f(C.c_double(x)))


def ba_solve(poses7, points3, obs_pose, obs_point, obs_uv, focal, cx, cy, max_iterations=50,
             function_tol=1e-6, gradient_tol=1e-10, parameter_tol=1e-8, initial_radius=1e4,
             num_threads=1, allreduce=None):
    poses = _f64(poses7).copy()
    pts = _f64(points3).copy()
    op = np.ascontiguousarray(obs_pose, np.int32)
    oj = np.ascontiguousarray(obs_point, np.int32)
    uv = _f64(obs_uv)
    summ = np.zeros(5)
    cb = ORA_ALLREDUCE(allreduce) if allreduce is not None else None
    oracle().ora_ba_solve(poses.shape[0], _p(poses), pts.shape[0], _p(pts), op.shape[0], _p(op), _p(oj),
                          _p(uv), C.c_double(focal), C.c_double(cx), C.c_double(cy), max_iterations,
                          C.c_double(function_tol), C.c_double(gradient_tol), C.c_double(parameter_tol),
                          C.c_double(initial_radius), num_threads, cb, None, _p(summ))
    return poses, pts, dict(iterations=int(summ[0]), successful=int(summ[1]), termination=int(summ[2]),
                            initial_cost=summ[3], final_cost=summ[4])


def ba_set_order(order):
    """2 (default): the declared chunk order; 1: rounds 1-3's 28-segment order (comparison only).  Process-wide: takes effect
    for solvers opened afterwards."""
    oracle().ora_ba_set_order(int(order))


class Pipeline:
    def __init__(self, **kw):
        L = oracle()
        self.p = OraPipelineParams(**kw)
        self.h = C.c_void_p(L.ora_pipeline_create(C.byref(self.p)))

    def process(self, left, right):
        left, right = _u8(left), _u8(right)
        res = OraFrameResult()
        oracle().ora_pipeline_process(self.h, _p(left), _p(right), C.byref(res))
        return res

    def tracked(self, capacity=8192):
        ids = np.empty(capacity, np.int64)
        xy = np.empty((capacity, 2), np.float32)
        n = oracle().ora_pipeline_get_tracked(self.h, _p(ids), _p(xy), capacity)
        return ids[:n].copy(), xy[:n].copy()

    def __del__(self):
        try:
            oracle().ora_pipeline_destroy(self.h)
        except Exception:
            pass


ORACLE_CONSTANT_NAMES = ("min_detected", "keyframe_percent_lost", "pnp_iterations", "pnp_reproj_error", "pnp_confidence",
                         "stereo_num_disparities", "stereo_block_size", "stereo_disparity_scale",
                         "triangulate_min_disparity_exclusive", "lk_win_w", "lk_max_level", "lk_max_iterations", "lk_epsilon",
                         "lk_min_eig_threshold", "fb_max_distance", "max_parallax")


def reference_constants():
    """The first-party literals as the oracle uses them (oracle/ora_constants.h), by the golden fixture's names."""
    buf = np.zeros(64, np.float64)
    n = oracle().ora_reference_constants(_p(buf), 64)
    assert n == len(ORACLE_CONSTANT_NAMES), n
    return dict(zip(ORACLE_CONSTANT_NAMES, buf[:n].tolist()))


class BAState:
    """ora_ba_state: passes A and B of the oracle's solver one at a time (this rank's unsummed payloads)."""

    def __init__(self, poses7, points3, obs_pose, obs_point, obs_uv, focal, cx, cy, num_threads=1):
        L = oracle()
        L.ora_ba_open.restype = C.c_void_p
        L.ora_ba_payload1_len.restype = C.c_size_t
        L.ora_ba_payload1_len.argtypes = [C.c_void_p]
        L.ora_ba_close.argtypes = [C.c_void_p]
        L.ora_ba_accept.argtypes = [C.c_void_p]
        L.ora_ba_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_ba_linearize.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.ora_ba_backsub.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        self.poses0, self.pts0 = _f64(poses7).copy(), _f64(points3).copy()
        self.op = np.ascontiguousarray(obs_pose, np.int32)
        self.oj = np.ascontiguousarray(obs_point, np.int32)
        self.uv = _f64(obs_uv).copy()
        self.K, self.N = self.poses0.shape[0], self.pts0.shape[0]
        self.h = C.c_void_p(L.ora_ba_open(self.K, _p(self.poses0), self.N, _p(self.pts0), self.op.shape[0], _p(self.op), _p(self.oj),
                                          _p(self.uv), C.c_double(focal), C.c_double(cx), C.c_double(cy), num_threads))
        self.pay1 = L.ora_ba_payload1_len(self.h)

    def linearize(self, at_candidate, radius, first):
        out = np.zeros(self.pay1)
        oracle().ora_ba_linearize(self.h, int(at_candidate), radius, int(first), _p(out))
        return out

    def backsub(self, dc, cand_poses, radius):
        dc, cand = _f64(dc), _f64(cand_poses)
        if dc.size == 0:
            dc = np.zeros(1)
        out = np.zeros(4)
        oracle().ora_ba_backsub(self.h, _p(dc), _p(cand), radius, _p(out))
        return out

    def accept(self):
        oracle().ora_ba_accept(self.h)

    def read(self):
        poses, pts = np.empty((self.K, 7)), np.empty((self.N, 3))
        oracle().ora_ba_read(self.h, _p(poses), _p(pts))
        return poses, pts

    def close(self):
        if self.h:
            oracle().ora_ba_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def product_lm_over_oracle_passes(poses7, points3, obs_pose, obs_point, obs_uv, focal, cx, cy, allreduce=None, max_iterations=50,
                                  num_threads=1):
    """The PRODUCT's LM step control (svo_lm_solve in libsvo_hip.so, host code) driving the ORACLE's passes: checks the
    product's host logic — speculation included — without a GPU.  allreduce(array) sums a host array in place over
    the ranks (ONE call per exchange).  Returns (poses, points, BASummary, LmStats, exchanges)."""
    import stereo_vo_amd as S
    st = BAState(poses7, points3, obs_pose, obs_point, obs_uv, focal, cx, cy, num_threads)
    exchanges = [0]

    def exch(buf):
        exchanges[0] += 1
        if allreduce is not None:
            allreduce(buf)
        return buf

    def linearize(radius, first):
        return exch(st.linearize(0, radius, first))

    def step(dc, cand, radius, ctl):
        p2 = st.backsub(dc, cand, radius)
        if ctl.spec_radius > 0:  # same sweep: ONE collective for both payloads
            both = exch(np.concatenate([p2, st.linearize(1, ctl.spec_radius, 0)]))
            return both[:4], both[4:], ctl.spec_radius, True
        p2 = exch(p2)
        if ctl.chain:  # decide from the SUMMED payload2 with the product's own rule, then pass A for the outcome
            acc, nr = S.api.lm_decide_step(ctl.cost, ctl.mcc, radius, ctl.decrease_factor, p2[0], p2[1])
            return p2, exch(st.linearize(1 if acc else 0, nr, 0)), nr, acc
        return p2, None, 0.0, False
    poses, summ, stats = S.lm_solve(poses7, linearize, step, st.accept, max_iterations=max_iterations)
    _, pts = st.read()
    st.close()
    return poses, pts, summ, stats, exchanges[0]
