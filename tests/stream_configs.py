"""The three long synthetic streams whose per-frame results are pinned by oracle-derived fixtures (tests/golden/stream_keys_*.npz):
ONE definition for the generator (tests/golden/gen_stream_keys.py), the tests (tests/test_pipeline.py) and bench.py's kitti_stream /
hd10k legs.  A key = what the reference's caller sees of a frame (src/vo_node.cpp:141-150: the keyframe's pose; the counters
of ImageProcessor::process, src/image_processor.cpp:18-163): n_detected, n_tracked, n_inliers, n_new, is_keyframe, LM iterations, the
bits of av_parallax (f32) and of the 7 pose doubles."""
import numpy as np

STREAMS = {
    # BASELINE configs[2] with the parameters tests/test_pipeline.py uses (configs[1]'s detector settings, 10-keyframe window)
    "kitti_test": dict(width=1241, height=376, frames=4541, max_corners=1500, quality=0.02, min_feature_distance=10.0, max_features=2000, window_size=10,
                       synth={}),
    # BASELINE configs[2] at its stated load (bench.py kitti_stream: ~13 k observations / ~7 k landmarks per window)
    "kitti_bench": dict(width=1241, height=376, frames=4541, max_corners=2800, quality=0.004, min_feature_distance=7.0, max_features=3300, window_size=10,
                        synth={}),
    # BASELINE configs[4] (bench.py hd10k): 1280 x 720, d435i focal, ~10 k features
    "hd10k": dict(width=1280, height=720, frames=64, max_corners=10000, quality=0.001, min_feature_distance=4.0, max_features=10000, window_size=10,
                  synth=dict(focal=385.7545, cx=640.0, cy=360.0, baseline=0.05, step_z=0.25)),
}
KEY_DTYPE = np.dtype([("n_detected", "<i4"), ("n_tracked", "<i4"), ("n_inliers", "<i4"), ("n_new", "<i4"), ("is_keyframe", "<i4"),
                      ("ba_iterations", "<i4"), ("av_parallax_bits", "<u4"), ("pose_bits", "<u8", (7,))])


def synth_params(S, name):
    """The generator's parameters of stream `name` (S = the stereo_vo_amd module: svo_synth_render is host code)."""
    c = STREAMS[name]
    p = S.synth_default(c["width"], c["height"])
    for k, v in c["synth"].items():
        setattr(p, k, v)
    return p


def key_of(r):
    """One frame's key from a FrameResult (product) or an OraFrameResult (oracle): same field names."""
    k = np.zeros((), KEY_DTYPE)
    k["n_detected"], k["n_tracked"], k["n_inliers"], k["n_new"] = r.n_detected, r.n_tracked, r.n_inliers, r.n_new
    k["is_keyframe"], k["ba_iterations"] = r.is_keyframe, r.ba_iterations
    k["av_parallax_bits"] = np.float32(r.av_parallax).view(np.uint32)
    k["pose_bits"] = np.array(list(r.pose7), np.float64).view(np.uint64)
    return k


def load_keys(name):
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"stream_keys_{name}.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    return z["keys"].view(KEY_DTYPE).reshape(-1)


def compare(keys, results, first=0):
    """Indices (at most 5) of frames whose result differs from the fixture; keys = load_keys(...), results = FrameResults of frames first, first + 1, ..."""
    bad = []
    for i, r in enumerate(results):
        if first + i >= len(keys):
            break
        if key_of(r).tobytes() != keys[first + i].tobytes():
            bad.append(first + i)
            if len(bad) >= 5:
                break
    return bad
