"""ctypes binding of include/svo.h.  Mirrors the C-ABI one to one; numpy arrays in, numpy arrays out."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SvoError(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "libsvo_hip.so")


class Limits(C.Structure):
    _fields_ = [("max_width", C.c_int), ("max_height", C.c_int), ("max_batch", C.c_int),
                ("max_corners", C.c_int), ("max_candidates", C.c_int), ("max_features", C.c_int)]


class CameraInfo(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("focal", "cx", "cy", "k1", "k2", "p1", "p2", "baseline")]


class BAOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("max_time_s", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_radius", C.c_double),
                ("max_features", C.c_int), ("accumulation", C.c_int)]


BA_ACC = {"auto": 0, "deterministic": 1, "atomics": 2, "mfma": 3}


class BASummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful_steps", C.c_int), ("termination", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("solve_ms", C.c_double)]


class PipelineParams(C.Structure):
    _fields_ = [("cam", CameraInfo), ("width", C.c_int), ("height", C.c_int), ("max_corners", C.c_int),
                ("quality", C.c_double), ("min_feature_distance", C.c_float),
                ("parallax_thresh", C.c_float), ("window_size", C.c_int), ("max_features", C.c_int),
                ("ba_max_iterations", C.c_int), ("ba_max_time_s", C.c_double)]


class FrameResult(C.Structure):
    _fields_ = [("n_detected", C.c_int), ("n_tracked", C.c_int), ("n_inliers", C.c_int),
                ("n_new", C.c_int), ("is_keyframe", C.c_int), ("av_parallax", C.c_float),
                ("percent_lost", C.c_float), ("pose7", C.c_double * 7), ("ba_iterations", C.c_int)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("width", C.c_int), ("height", C.c_int), ("focal", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("baseline", C.c_double), ("step_z", C.c_double),
                ("step_x", C.c_double), ("yaw_per_frame", C.c_double), ("n_billboards", C.c_int)]


REFERENCE_CONSTANT_NAMES = (
    "gftt_max_corners", "gftt_quality", "min_detected", "keyframe_percent_lost", "pnp_iterations", "pnp_reproj_error",
    "pnp_confidence", "stereo_num_disparities", "stereo_block_size", "stereo_disparity_scale",
    "triangulate_min_disparity_exclusive", "lk_win_w", "lk_win_h", "lk_max_level", "lk_max_iterations", "lk_epsilon",
    "lk_min_eig_threshold", "fb_max_distance", "max_parallax", "draw_thickness", "parallax_thresh", "min_feature_distance",
    "sliding_window_size", "max_features", "ba_max_solver_time_s", "ba_num_threads")


class ReferenceConstants(C.Structure):
    _fields_ = [(n, C.c_double) for n in REFERENCE_CONSTANT_NAMES]


def reference_constants():
    """The reference's first-party literals as compiled into the library (svo_reference_constants)."""
    c = ReferenceConstants()
    if lib().svo_reference_constants(C.byref(c)) != 0:
        raise SvoError("svo_reference_constants failed")
    return {n: getattr(c, n) for n in REFERENCE_CONSTANT_NAMES}


class LmStats(C.Structure):
    _fields_ = [("linearize_calls", C.c_int), ("step_calls", C.c_int), ("speculations", C.c_int), ("speculation_hits", C.c_int),
                ("single_exchange", C.c_int), ("collectives", C.c_int), ("device_control", C.c_int), ("fallbacks", C.c_int),
                ("host_us", C.c_double)]


class LmStepCtl(C.Structure):
    _fields_ = [("cost", C.c_double), ("mcc", C.c_double), ("decrease_factor", C.c_double), ("spec_radius", C.c_double),
                ("chain", C.c_int)]


LM_LINEARIZE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_double))
LM_STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(LmStepCtl),
                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int))
LM_ACCEPT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class LmOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("linearize", LM_LINEARIZE_FN), ("step", LM_STEP_FN), ("accept", LM_ACCEPT_FN)]


def lm_decide_step(cost, mcc, radius, decrease_factor, cost_new, model_change_points):
    """svo_lm_decide_step: Ceres' accept / radius rule -> (accept, next_radius)."""
    acc, rad = C.c_int(0), C.c_double(0.0)
    lib().svo_lm_decide_step(C.c_double(cost), C.c_double(mcc), C.c_double(radius), C.c_double(decrease_factor), C.c_double(cost_new),
                             C.c_double(model_change_points), C.byref(acc), C.byref(rad))
    return bool(acc.value), rad.value


def lm_solve(poses7, linearize, step, accept, max_iterations=50, max_time_s=0.0):
    """svo_lm_solve — the product's LM step control (host/lm.cpp) over caller-provided passes (no GPU involved).
    linearize(radius, first) -> payload1 array;
    step(dc, cand_poses, radius, ctl) -> (payload2, payload1_next or None, next_radius, next_at_candidate) with ctl an
    LmStepCtl (cost, mcc, decrease_factor, spec_radius, chain);  accept() -> None.
    Returns (poses7, BASummary, LmStats)."""
    poses = np.ascontiguousarray(poses7, np.float64).copy()
    K = poses.shape[0]
    n = 6 * (K - 1)
    pay1 = n * n + 3 * n + 2

    def _lin(user, radius, first, out):
        np.ctypeslib.as_array(out, shape=(pay1,))[:] = linearize(radius, first)
        return 0

    def _step(user, dc, cand, radius, ctl, out2, out1, out_radius, out_at_cand):
        d = np.ctypeslib.as_array(dc, shape=(max(n, 1),))[:n].copy()
        c = np.ctypeslib.as_array(cand, shape=(K, 7)).copy()
        p2, p1, nr, at_cand = step(d, c, radius, ctl.contents)
        np.ctypeslib.as_array(out2, shape=(4,))[:] = p2
        if p1 is not None:
            np.ctypeslib.as_array(out1, shape=(pay1,))[:] = p1
        out_radius[0] = nr if p1 is not None else 0.0
        out_at_cand[0] = int(bool(at_cand))
        return 0

    def _acc(user):
        accept()
        return 0
    ops = LmOps(None, LM_LINEARIZE_FN(_lin), LM_STEP_FN(_step), LM_ACCEPT_FN(_acc))
    opt = ba_default_options()
    opt.max_iterations, opt.max_time_s = max_iterations, max_time_s
    s, st = BASummary(), LmStats()
    rc = lib().svo_lm_solve(K, _p(poses), C.byref(ops), C.byref(opt), C.byref(s), C.byref(st))
    if rc:
        raise SvoError(f"svo_lm_solve failed ({rc})")
    return poses, s, st


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/svo.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "svo_create", "svo_destroy", "svo_last_error", "svo_stream", "svo_sync", "svo_version", "svo_reference_constants",
    "svo_profile_select", "svo_profile_read", "svo_measure_peak",
    "svo_reproj_eval", "svo_reproj_eval_dev",
    "svo_corner_detect", "svo_corner_detect_batch_dev", "svo_corner_response",
    "svo_stereo_bm", "svo_stereo_disparity_at", "svo_stereo_disparity_at_dev",
    "svo_triangulate", "svo_lk_track", "svo_build_pyramid", "svo_track_features", "svo_dedup",
    "svo_pnp_ransac",
    "svo_ba_default_options", "svo_ba_create", "svo_ba_destroy", "svo_ba_reset", "svo_ba_add_keyframe", "svo_ba_solve",
    "svo_ba_get_pose", "svo_ba_window_count", "svo_ba_get_points", "svo_ba_load_problem",
    "svo_ba_set_allreduce", "svo_ba_set_device_lm", "svo_ba_set_bulk_control", "svo_ba_set_solve_form", "svo_ba_solve_problem", "svo_ba_read_problem", "svo_ba_set_comm", "svo_ba_last_stats", "svo_lm_solve", "svo_lm_decide_step",
    "svo_rccl_unique_id", "svo_rccl_comm_create", "svo_rccl_comm_destroy",
    "svo_pipeline_default_params", "svo_pipeline_create", "svo_pipeline_destroy", "svo_pipeline_reset",
    "svo_pipeline_process_batch_dev", "svo_pipeline_process_batch", "svo_pipeline_get_tracked",
    "svo_pipeline_group_create", "svo_pipeline_group_destroy", "svo_pipeline_group_reset", "svo_pipeline_group_lanes",
    "svo_pipeline_group_process_batch_dev", "svo_pipeline_group_get_tracked", "svo_pipeline_group_last_stats",
    "svo_pipeline_group_solve_work", "svo_pipeline_group_staging", "svo_pipeline_group_upload", "svo_pipeline_group_process_uploaded", "svo_pipeline_group_process_batch",
    "svo_synth_default_params", "svo_synth_render", "svo_synth_pose",
    "svo_image_read_gray", "svo_kitti_read_poses", "svo_ate_rmse", "svo_kitti_run", "svo_cholesky_solve", "svo_cholesky_solve_dev", "svo_draw_track", "svo_pipeline_draw_track",
]


def lib():
    """Load libsvo_hip.so (raises if it has not been built: there is no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise SvoError(f"{p} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(p)
        L.svo_version.restype = C.c_char_p
        L.svo_last_error.restype = C.c_char_p
        L.svo_last_error.argtypes = [C.c_void_p]
        L.svo_stream.restype = C.c_void_p
        L.svo_stream.argtypes = [C.c_void_p]
        L.svo_destroy.argtypes = [C.c_void_p]
        L.svo_destroy.restype = None
        _LIB = L
    return _LIB


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def synth_default(width, height):
    p = SynthParams()
    lib().svo_synth_default_params(C.byref(p), width, height)
    return p


def synth_render(params, frame):
    L = lib()
    left = np.empty((params.height, params.width), np.uint8)
    right = np.empty_like(left)
    rc = L.svo_synth_render(C.byref(params), frame, _p(left), _p(right))
    if rc:
        raise SvoError(f"svo_synth_render rc={rc}")
    return left, right


def synth_pose(params, frame):
    rt = np.empty(12, np.float64)
    rc = lib().svo_synth_pose(C.byref(params), frame, _p(rt))
    if rc:
        raise SvoError(f"svo_synth_pose rc={rc}")
    return rt.reshape(3, 4)


class Context:
    """svo_ctx wrapper.  Host-pointer entry points (numpy in/out)."""

    def __init__(self, max_width, max_height, device=0, max_batch=1, max_corners=2048,
                 max_candidates=65536, max_features=2048):
        self.L = lib()
        self.lim = Limits(max_width, max_height, max_batch, max_corners, max_candidates, max_features)
        self.h = C.c_void_p()
        rc = self.L.svo_create(C.byref(self.h), device, C.byref(self.lim))
        if rc:
            raise SvoError(f"svo_create failed rc={rc} (no GPU / HIP error); the HIP path has no fallback")

    def close(self):
        if self.h:
            self.L.svo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise SvoError(f"{what} rc={rc}: {self.L.svo_last_error(self.h).decode()}")

    def sync(self):
        self._chk(self.L.svo_sync(self.h), "svo_sync")

    def cholesky_solve_dev(self, A, b):
        """svo_cholesky_solve_dev on copies (the device-side solve of the LM controller workgroup)."""
        A = np.array(A, np.float64, order="C")
        b = np.array(b, np.float64)
        self._chk(self.L.svo_cholesky_solve_dev(self.h, _p(A), _p(b), b.shape[0]), "svo_cholesky_solve_dev")
        return b

    def profile_select(self, kernel):
        self._chk(self.L.svo_profile_select(self.h, kernel.encode() if kernel else None), "svo_profile_select")

    def profile_read(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.L.svo_profile_read(self.h, C.byref(ms), C.byref(n)), "svo_profile_read")
        return ms.value, n.value

    @property
    def stream(self):
        return self.L.svo_stream(self.h)

    # ---- a11
    def measure_peak(self, what):
        """flop/s or bytes/s of this card for "f64_fma" / "f64_muladd" / "f64_mfma" / "hbm_copy" (svo_measure_peak)."""
        v = C.c_double(0.0)
        self._chk(self.L.svo_measure_peak(self.h, what.encode(), C.byref(v)), "svo_measure_peak")
        return v.value

    def reproj_eval(self, pose7, point3, obs2, focal, cx, cy, want_jpose=True, want_jpoint=True):
        pose7, point3, obs2 = _f64(pose7), _f64(point3), _f64(obs2)
        n = pose7.shape[0]
        r = np.empty((n, 2))
        jq = np.empty((n, 14)) if want_jpose else None
        jx = np.empty((n, 6)) if want_jpoint else None
        self._chk(self.L.svo_reproj_eval(self.h, n, _p(pose7), _p(point3), _p(obs2), C.c_double(focal),
                                         C.c_double(cx), C.c_double(cy), _p(r), _p(jq), _p(jx)),
                  "svo_reproj_eval")
        return r, jq, jx

    # ---- a1
    def corner_response(self, img):
        img = _u8(img)
        h, w = img.shape
        eig = np.empty((h, w), np.float32)
        self._chk(self.L.svo_corner_response(self.h, _p(img), w, h, w, _p(eig)), "svo_corner_response")
        return eig

    def corner_detect(self, img, max_corners=300, quality=0.1, min_distance=30.0):
        img = _u8(img)
        h, w = img.shape
        xy = np.empty((max_corners, 2), np.float32)
        n = C.c_int(0)
        self._chk(self.L.svo_corner_detect(self.h, _p(img), w, h, w, max_corners, C.c_double(quality),
                                           C.c_double(min_distance), _p(xy), C.byref(n)),
                  "svo_corner_detect")
        return xy[:n.value].copy()

    # ---- a7
    def stereo_bm(self, left, right, ndisp=48, block=21):
        left, right = _u8(left), _u8(right)
        h, w = left.shape
        d = np.empty((h, w), np.int16)
        self._chk(self.L.svo_stereo_bm(self.h, _p(left), _p(right), w, h, w, ndisp, block, _p(d)),
                  "svo_stereo_bm")
        return d

    def stereo_disparity_at(self, left, right, xy, ndisp=48, block=21):
        left, right, xy = _u8(left), _u8(right), _f32(xy)
        h, w = left.shape
        n = xy.shape[0]
        d = np.empty(n, np.float32)
        self._chk(self.L.svo_stereo_disparity_at(self.h, _p(left), _p(right), w, h, w, ndisp, block,
                                                 _p(xy), n, _p(d)), "svo_stereo_disparity_at")
        return d

    # ---- a8
    def triangulate(self, xy, disp, pose16, focal, cx, cy, baseline):
        xy, disp, pose16 = _f32(xy), _f32(disp), _f32(pose16)
        n = xy.shape[0]
        kxy = np.empty((n, 2), np.float32)
        xyz = np.empty((n, 3), np.float32)
        kidx = np.empty(n, np.int32)
        m = C.c_int(0)
        self._chk(self.L.svo_triangulate(self.h, _p(xy), _p(disp), n, _p(pose16), C.c_float(focal),
                                         C.c_float(cx), C.c_float(cy), C.c_float(baseline), _p(kxy),
                                         _p(xyz), _p(kidx), C.byref(m)), "svo_triangulate")
        return kxy[:m.value].copy(), xyz[:m.value].copy(), kidx[:m.value].copy()

    # ---- a3
    def build_pyramid(self, img):
        img = _u8(img)
        h, w = img.shape
        sizes = []
        lw, lh = w, h
        for _ in range(4):
            sizes.append((lh, lw))
            lw, lh = (lw + 1) // 2, (lh + 1) // 2
        total = sum(a * b for a, b in sizes)
        buf = np.empty(total, np.uint8)
        self._chk(self.L.svo_build_pyramid(self.h, _p(img), w, h, w, _p(buf), C.c_size_t(total)),
                  "svo_build_pyramid")
        out, off = [], 0
        for (a, b) in sizes:
            out.append(buf[off:off + a * b].reshape(a, b).copy())
            off += a * b
        return out

    def lk_track(self, prev, nxt, xy):
        prev, nxt, xy = _u8(prev), _u8(nxt), _f32(xy)
        h, w = prev.shape
        n = xy.shape[0]
        out = np.empty((n, 2), np.float32)
        st = np.empty(n, np.uint8)
        self._chk(self.L.svo_lk_track(self.h, _p(prev), _p(nxt), w, h, w, _p(xy), n, _p(out), _p(st)),
                  "svo_lk_track")
        return out, st

    def track_features(self, prev, nxt, xy, initial_xy):
        prev, nxt, xy, initial_xy = _u8(prev), _u8(nxt), _f32(xy), _f32(initial_xy)
        h, w = prev.shape
        n = xy.shape[0]
        kxy = np.empty((n, 2), np.float32)
        kidx = np.empty(n, np.int32)
        m = C.c_int(0)
        av = C.c_float(0)
        self._chk(self.L.svo_track_features(self.h, _p(prev), _p(nxt), w, h, w, _p(xy), _p(initial_xy), n,
                                            _p(kxy), _p(kidx), C.byref(m), C.byref(av)),
                  "svo_track_features")
        return kxy[:m.value].copy(), kidx[:m.value].copy(), av.value

    # ---- a6
    def dedup(self, det, trk, min_distance):
        det, trk = _f32(det), _f32(trk)
        out = np.empty_like(det)
        m = C.c_int(0)
        self._chk(self.L.svo_dedup(self.h, _p(det), det.shape[0], _p(trk), trk.shape[0],
                                   C.c_float(min_distance), _p(out), C.byref(m)), "svo_dedup")
        return out[:m.value].copy()

    # ---- a5
    def pnp_ransac(self, xyz, xy, focal, cx, cy, rvec, tvec, iterations=100, reproj_err=8.0, confidence=0.99):
        xyz, xy = _f32(xyz), _f32(xy)
        n = xyz.shape[0]
        rv, tv = _f64(rvec).copy(), _f64(tvec).copy()
        inl = np.empty(max(n, 1), np.int32)
        m = C.c_int(0)
        self._chk(self.L.svo_pnp_ransac(self.h, _p(xyz), _p(xy), n, C.c_float(focal), C.c_float(cx),
                                        C.c_float(cy), _p(rv), _p(tv), iterations, C.c_float(reproj_err),
                                        C.c_double(confidence), _p(inl), C.byref(m)), "svo_pnp_ransac")
        return rv, tv, inl[:m.value].copy()


class BA:
    """svo_ba wrapper: sliding-window graph (add_keyframe / solve) and the bulk-problem interface."""

    def __init__(self, ctx, window_size, focal, cx, cy, baseline=0.0, max_landmarks=1 << 16,
                 max_observations=1 << 18, max_iterations=50, max_time_s=0.0, max_features=400, accumulation="auto", device_lm=None, bulk_control=None, solve_form=None):
        self.ctx = ctx
        self.L = ctx.L
        cam = CameraInfo(focal, cx, cy, 0, 0, 0, 0, baseline)
        opt = BAOptions()
        self.L.svo_ba_default_options(C.byref(opt))
        opt.max_iterations = max_iterations
        opt.max_time_s = max_time_s
        opt.max_features = max_features
        opt.accumulation = BA_ACC[accumulation]
        self.h = C.c_void_p()
        ctx._chk(self.L.svo_ba_create(ctx.h, C.byref(self.h), window_size, C.byref(cam), C.byref(opt),
                                      max_landmarks, max_observations), "svo_ba_create")
        if device_lm is not None:
            ctx._chk(self.L.svo_ba_set_device_lm(self.h, 1 if device_lm else 0), "svo_ba_set_device_lm")
        if solve_form is not None:  # "wide" / "compact": which device-resident form a window solve takes
            ctx._chk(self.L.svo_ba_set_solve_form(self.h, {"wide": 0, "compact": 1}[solve_form]), "svo_ba_set_solve_form")
        if bulk_control is not None:  # bulk / sharded solves: step control on the device (True) or host-driven (False)
            ctx._chk(self.L.svo_ba_set_bulk_control(self.h, 1 if bulk_control else 0), "svo_ba_set_bulk_control")
        self.L.svo_ba_destroy.argtypes = [C.c_void_p]
        self._cb = None

    def close(self):
        if self.h:
            self.L.svo_ba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_problem(self, poses7, points3, obs_pose, obs_point, obs_uv):
        poses7, points3, obs_uv = _f64(poses7), _f64(points3), _f64(obs_uv)
        op = np.ascontiguousarray(obs_pose, np.int32)
        oj = np.ascontiguousarray(obs_point, np.int32)
        self._shape = (poses7.shape[0], points3.shape[0])
        self.ctx._chk(self.L.svo_ba_load_problem(self.h, poses7.shape[0], _p(poses7), points3.shape[0],
                                                 _p(points3), op.shape[0], _p(op), _p(oj), _p(obs_uv)),
                      "svo_ba_load_problem")

    def set_allreduce(self, fn):
        """fn(dev_ptr:int, n_doubles:int) -> 0; must sum the device buffer in place over all ranks."""
        self._cb = ALLREDUCE_FN(lambda ptr, n, user: int(fn(ptr, n) or 0)) if fn else None
        self.ctx._chk(self.L.svo_ba_set_allreduce(self.h, self._cb if self._cb else C.cast(None, ALLREDUCE_FN), None),
                      "svo_ba_set_allreduce")

    def set_comm(self, nccl_comm):
        """nccl_comm: ncclComm_t as an int / c_void_p (see rccl_comm_create); None detaches."""
        self.ctx._chk(self.L.svo_ba_set_comm(self.h, C.c_void_p(nccl_comm) if nccl_comm else None), "svo_ba_set_comm")

    def solve_problem(self):
        s = BASummary()
        self.ctx._chk(self.L.svo_ba_solve_problem(self.h, C.byref(s)), "svo_ba_solve_problem")
        return s

    def last_stats(self):
        st = LmStats()
        self.ctx._chk(self.L.svo_ba_last_stats(self.h, C.byref(st)), "svo_ba_last_stats")
        return st

    def read_problem(self):
        K, N = self._shape
        poses = np.empty((K, 7))
        pts = np.empty((N, 3))
        self.ctx._chk(self.L.svo_ba_read_problem(self.h, _p(poses), _p(pts)), "svo_ba_read_problem")
        return poses, pts

    def add_keyframe(self, pose7, tracked_ids, tracked_xy, new_xy, new_xyz):
        pose7 = _f64(pose7)
        tid = np.ascontiguousarray(tracked_ids, np.int64)
        txy, nxy, nxyz = _f32(tracked_xy), _f32(new_xy), _f32(new_xyz)
        nn = nxy.shape[0] if nxy.ndim == 2 else 0
        ids = np.empty(max(nn, 1), np.int64)
        m = C.c_int(0)
        self.ctx._chk(self.L.svo_ba_add_keyframe(self.h, _p(pose7), _p(tid), _p(txy), tid.shape[0], _p(nxy),
                                                 _p(nxyz), nn, _p(ids), C.byref(m)), "svo_ba_add_keyframe")
        return ids[:m.value].copy()

    def solve(self):
        s = BASummary()
        self.ctx._chk(self.L.svo_ba_solve(self.h, C.byref(s)), "svo_ba_solve")
        return s

    def get_pose(self, k=-1):
        p = np.empty(7)
        self.ctx._chk(self.L.svo_ba_get_pose(self.h, k, _p(p)), "svo_ba_get_pose")
        return p

    def window_count(self):
        return self.L.svo_ba_window_count(self.h)

    def get_points(self, ids):
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.empty((ids.shape[0], 3), np.float32)
        self.ctx._chk(self.L.svo_ba_get_points(self.h, _p(ids), ids.shape[0], _p(out)), "svo_ba_get_points")
        return out


def rccl_unique_id():
    """128-byte ncclUniqueId (bytes) from the librccl the library binds; distribute it to every rank out of band."""
    buf = C.create_string_buffer(128)
    if lib().svo_rccl_unique_id(buf) != 0:
        raise SvoError("svo_rccl_unique_id failed (librccl not available?)")
    return buf.raw


def rccl_comm_create(n_ranks, rank, id128, device):
    comm = C.c_void_p()
    rc = lib().svo_rccl_comm_create(C.byref(comm), n_ranks, rank, C.c_char_p(id128), device)
    if rc != 0:
        raise SvoError(f"svo_rccl_comm_create failed ({rc})")
    return comm.value


def rccl_comm_destroy(comm):
    lib().svo_rccl_comm_destroy.argtypes = [C.c_void_p]
    lib().svo_rccl_comm_destroy(comm)


def ba_default_options():
    o = BAOptions()
    lib().svo_ba_default_options(C.byref(o))
    return o


def pipeline_default_params():
    p = PipelineParams()
    lib().svo_pipeline_default_params(C.byref(p))
    return p


class Pipeline:
    """svo_pipeline wrapper: ImageProcessor::process + BundleAdjuster::bundle_adjust per frame."""

    def __init__(self, ctx, params):
        self.ctx, self.L, self.prm = ctx, ctx.L, params
        self.h = C.c_void_p()
        ctx._chk(self.L.svo_pipeline_create(ctx.h, C.byref(self.h), C.byref(params)), "svo_pipeline_create")
        self.L.svo_pipeline_destroy.argtypes = [C.c_void_p]

    def close(self):
        if self.h:
            self.L.svo_pipeline_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.ctx._chk(self.L.svo_pipeline_reset(self.h), "svo_pipeline_reset")

    def process_batch(self, left, right):
        """left/right: (B, H, W) uint8 host arrays."""
        left, right = _u8(left), _u8(right)
        b = left.shape[0]
        res = (FrameResult * b)()
        self.ctx._chk(self.L.svo_pipeline_process_batch(self.h, _p(left), _p(right), b, res), "svo_pipeline_process_batch")
        return list(res)

    def process_batch_dev(self, left_ptr, right_ptr, batch):
        """left_ptr/right_ptr: raw device pointers (ints) to (B, H, W) uint8 images resident in HBM."""
        res = (FrameResult * batch)()
        self.ctx._chk(self.L.svo_pipeline_process_batch_dev(self.h, C.c_void_p(left_ptr), C.c_void_p(right_ptr), batch, res),
                      "svo_pipeline_process_batch_dev")
        return list(res)

    def draw_track(self, keyframe_gray):
        """RGB drawing of the tracker state over the (host) keyframe image: FeatureTracker::draw_track + get_drawing."""
        g = np.ascontiguousarray(keyframe_gray, np.uint8)
        out = np.empty(g.shape + (3,), np.uint8)
        self.ctx._chk(self.L.svo_pipeline_draw_track(self.h, _p(g), g.shape[1], _p(out)), "svo_pipeline_draw_track")
        return out

    def tracked(self, capacity=8192):
        ids = np.empty(capacity, np.int64)
        xy = np.empty((capacity, 2), np.float32)
        n = C.c_int(0)
        self.ctx._chk(self.L.svo_pipeline_get_tracked(self.h, _p(ids), _p(xy), capacity, C.byref(n)), "svo_pipeline_get_tracked")
        return ids[:n.value].copy(), xy[:n.value].copy()


class PipelineGroup:
    """svo_pipeline_group wrapper: n_lanes independent stereo streams behind one caller thread (stream-batched launches)."""

    STAGES = ("track", "pnp_ransac", "host_thread_busy_us_of_call_us", "dedup_stereo_triangulate", "bundle_adjust", "corners_pyramids")

    def __init__(self, ctx, params, n_lanes):
        self.ctx, self.L, self.prm, self.n_lanes = ctx, ctx.L, params, n_lanes
        self.h = C.c_void_p()
        self.L.svo_pipeline_group_destroy.argtypes = [C.c_void_p]
        self.L.svo_pipeline_group_destroy.restype = None
        self.L.svo_pipeline_group_process_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        ctx._chk(self.L.svo_pipeline_group_create(ctx.h, C.byref(self.h), C.byref(params), n_lanes), "svo_pipeline_group_create")

    def close(self):
        if self.h:
            self.L.svo_pipeline_group_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.ctx._chk(self.L.svo_pipeline_group_reset(self.h), "svo_pipeline_group_reset")

    def process_batch_dev(self, left_ptr, right_ptr, lane_stride, batch):
        """left_ptr/right_ptr: raw device pointers to (n_lanes, B, H, W) uint8 images (lane_stride bytes between lanes).
        Returns a list of n_lanes lists of FrameResult."""
        res = (FrameResult * (self.n_lanes * batch))()
        self.ctx._chk(self.L.svo_pipeline_group_process_batch_dev(self.h, C.c_void_p(left_ptr), C.c_void_p(right_ptr), C.c_size_t(lane_stride),
                                                                  batch, res), "svo_pipeline_group_process_batch_dev")
        self.last_raw = bytes(res)  # every lane's results as the library wrote them (it zeroes the records first): bench.py's bit-for-bit checks
        return [list(res[l * batch:(l + 1) * batch]) for l in range(self.n_lanes)]

    def staging(self, slot):
        """The slot's pinned staging buffers as numpy views (n_lanes, max_batch, H, W): the caller fills them in place."""
        lp, rp, stride = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint8)(), C.c_size_t(0)
        self.L.svo_pipeline_group_staging.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.ctx._chk(self.L.svo_pipeline_group_staging(self.h, slot, C.byref(lp), C.byref(rp), C.byref(stride)), "svo_pipeline_group_staging")
        h, w = self.prm.height, self.prm.width
        mb = stride.value // (h * w)
        shape = (self.n_lanes, mb, h, w)
        n = self.n_lanes * mb * h * w
        return (np.ctypeslib.as_array(lp, shape=(n,)).reshape(shape), np.ctypeslib.as_array(rp, shape=(n,)).reshape(shape))

    def upload(self, slot, batch):
        self.ctx._chk(self.L.svo_pipeline_group_upload(self.h, slot, batch), "svo_pipeline_group_upload")

    def process_uploaded(self, slot, batch):
        res = (FrameResult * (self.n_lanes * batch))()
        self.L.svo_pipeline_group_process_uploaded.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.ctx._chk(self.L.svo_pipeline_group_process_uploaded(self.h, slot, res), "svo_pipeline_group_process_uploaded")
        self.last_raw = bytes(res)
        return [list(res[l * batch:(l + 1) * batch]) for l in range(self.n_lanes)]

    def process_batch(self, left, right):
        """left/right: (n_lanes, B, H, W) uint8 HOST arrays (the convenience entry: copy into slot 0, upload, process)."""
        left, right = _u8(left), _u8(right)
        batch = left.shape[1]
        res = (FrameResult * (self.n_lanes * batch))()
        self.L.svo_pipeline_group_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.ctx._chk(self.L.svo_pipeline_group_process_batch(self.h, _p(left), _p(right), C.c_size_t(batch * left.shape[2] * left.shape[3]), batch, res),
                      "svo_pipeline_group_process_batch")
        self.last_raw = bytes(res)
        return [list(res[l * batch:(l + 1) * batch]) for l in range(self.n_lanes)]

    def get_tracked(self, lane, capacity=8192):
        ids = np.empty(capacity, np.int64)
        xy = np.empty((capacity, 2), np.float32)
        n = C.c_int(0)
        self.ctx._chk(self.L.svo_pipeline_group_get_tracked(self.h, lane, _p(ids), _p(xy), capacity, C.byref(n)), "svo_pipeline_group_get_tracked")
        return ids[:n.value].copy(), xy[:n.value].copy()

    def solve_work(self, reset=False):
        """Algorithmic [f64 flops, bytes, solves, LM iterations] of the bundle adjustments finished since the last reset."""
        out = (C.c_double * 4)()
        self.ctx._chk(self.L.svo_pipeline_group_solve_work(self.h, out, int(reset)), "svo_pipeline_group_solve_work")
        return list(out)

    def last_stats(self):
        """{stage: (launches, lane-stages carried)} of the last process_batch_dev call; "host_thread_busy_us_of_call_us": (microseconds of the
        driving thread's loop passes that did something, microseconds of the call's main loop)."""
        a, b = (C.c_long * 6)(), (C.c_long * 6)()
        self.ctx._chk(self.L.svo_pipeline_group_last_stats(self.h, a, b), "svo_pipeline_group_last_stats")
        return {s: (a[i], b[i]) for i, s in enumerate(self.STAGES) if s != "unused"}


class RunStats(C.Structure):
    _fields_ = [("frames", C.c_int), ("keyframes", C.c_int), ("ate_rmse", C.c_double), ("seconds", C.c_double)]


def image_read_gray(path, max_pixels=1 << 24):
    buf = np.empty(max_pixels, np.uint8)
    w, h = C.c_int(0), C.c_int(0)
    rc = lib().svo_image_read_gray(path.encode(), _p(buf), C.c_size_t(max_pixels), C.byref(w), C.byref(h))
    if rc:
        raise SvoError(f"svo_image_read_gray({path}) rc={rc}")
    return buf[:w.value * h.value].reshape(h.value, w.value).copy()


def kitti_read_poses(path, max_frames=100000):
    rt = np.empty((max_frames, 12))
    n = C.c_int(0)
    rc = lib().svo_kitti_read_poses(path.encode(), _p(rt), max_frames, C.byref(n))
    if rc:
        raise SvoError(f"svo_kitti_read_poses({path}) rc={rc}")
    return rt[:n.value].reshape(-1, 3, 4).copy()


def ate_rmse(est_xyz, gt_xyz, with_scale=False):
    est_xyz, gt_xyz = _f64(est_xyz), _f64(gt_xyz)
    out = C.c_double(0)
    rc = lib().svo_ate_rmse(_p(est_xyz), _p(gt_xyz), est_xyz.shape[0], int(with_scale), C.byref(out))
    if rc:
        raise SvoError(f"svo_ate_rmse rc={rc}")
    return out.value


def kitti_run(ctx, params, data_path, sequence, max_frames):
    """Non-ROS driver over a KITTI-layout directory (data_path must end with '/')."""
    traj = np.zeros((max_frames, 12))
    st = RunStats()
    ctx._chk(lib().svo_kitti_run(ctx.h, C.byref(params), data_path.encode(), sequence, max_frames, _p(traj), C.byref(st)),
             "svo_kitti_run")
    return traj[:st.frames].reshape(-1, 3, 4).copy(), st


def cholesky_solve(A, b):
    """svo_cholesky_solve on copies: returns x with (L L^T) x = b, or raises SvoError when A is not SPD."""
    A = np.array(A, np.float64, order="C")
    b = np.array(b, np.float64)
    rc = lib().svo_cholesky_solve(_p(A), _p(b), b.shape[0])
    if rc != 0:
        raise SvoError(f"svo_cholesky_solve rc={rc}")
    return b


def draw_track(gray, from_xy, to_xy):
    """svo_draw_track: gray (H, W) uint8 -> RGB (H, W, 3) with a green arrow per (from, to) pair."""
    g = np.ascontiguousarray(gray, np.uint8)
    a = np.ascontiguousarray(from_xy, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(to_xy, np.float32).reshape(-1, 2)
    out = np.empty(g.shape + (3,), np.uint8)
    rc = lib().svo_draw_track(_p(g), g.shape[1], g.shape[0], g.shape[1], _p(a), _p(b), a.shape[0], _p(out))
    if rc:
        raise SvoError(f"svo_draw_track rc={rc}")
    return out
