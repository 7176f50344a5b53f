"""Generates tests/golden/stream_keys_<name>.npz: the CPU oracle (oracle/ora_pipeline.cpp, the restatement of
ImageProcessor::process + BundleAdjuster::bundle_adjust, reference src/image_processor.cpp:18-163, src/bundle_adjuster.cpp:41-163,
driver rule src/vo_node.cpp:141-148) run over EVERY frame of the long synthetic streams of tests/stream_configs.py, one key per
frame.  Run in the build container (minutes per stream on 8 cores); the GPU tests and bench.py then compare every frame of the
HIP path against these keys — not only the first 48.  The frames come from the in-repo integer-PRNG renderer
(stereo_vo_amd/host/synth.cpp, host code, bit-reproducible on every machine); nothing is read from /root/reference.
    python tests/golden/gen_stream_keys.py [name ...] [--frames N]
The fixture records what it was generated with (oracle source hash, parameters); tests refuse a fixture whose parameters differ."""
import hashlib
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import stream_configs as SC  # noqa: E402
import stereo_vo_amd as S  # noqa: E402


def oracle_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "oracle", "*.cpp")) + glob.glob(os.path.join(ROOT, "oracle", "*.h")) + [os.path.join(ROOT, "stereo_vo_amd", "host", "synth.cpp")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def generate(name, frames=None):
    c = SC.STREAMS[name]
    n = frames or c["frames"]
    p = SC.synth_params(S, name)
    o = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=c["width"], height=c["height"], max_corners=c["max_corners"],
                   quality=c["quality"], min_feature_distance=c["min_feature_distance"], parallax_thresh=20.0, window_size=c["window_size"],
                   max_features=c["max_features"], ba_max_iterations=50, num_threads=min(os.cpu_count() or 1, 16))
    keys = np.zeros(n, SC.KEY_DTYPE)
    t0 = time.time()
    for i in range(n):
        left, right = S.synth_render(p, i)
        keys[i] = SC.key_of(o.process(left, right))
        if i % 200 == 199:
            print(f"{name}: frame {i + 1}/{n}, {time.time() - t0:.0f} s, keyframes so far {int(keys['is_keyframe'][:i + 1].sum())}", flush=True)
    out = os.path.join(ROOT, "tests", "golden", f"stream_keys_{name}.npz")
    meta = dict(name=name, frames_generated=n, oracle_source_sha256_16=oracle_source_hash(), **{k: v for k, v in c.items() if k != "synth"}, synth=str(c["synth"]))
    np.savez_compressed(out, keys=keys.view(np.uint8), meta=np.array(repr(meta)))
    print(f"wrote {out}: {n} frames, {int(keys['is_keyframe'].sum())} keyframes, {os.path.getsize(out)} bytes, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else None
    if frames is not None:
        args = [a for a in args if a != str(frames)]
    for name in (args or list(SC.STREAMS)):
        generate(name, frames)
