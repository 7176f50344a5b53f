"""KITTI-layout ingestion (src/kitti_node.cpp:37-68), the non-ROS driver with vo_node's
loop semantics (src/vo_node.cpp:141-150) and ATE."""
import os

import numpy as np
import pytest

import oracle_lib as O


def _export(tmp, n=8, w=496, h=160, focal=300.0):
    """Write a synthetic sequence in the KITTI odometry layout."""
    from PIL import Image
    import stereo_vo_amd as S
    p = S.synth_default(w, h)
    p.focal = focal
    root = str(tmp) + "/"
    os.makedirs(root + "07/image_0"); os.makedirs(root + "07/image_1")
    os.makedirs(root + "data_odometry_poses/dataset/poses")
    frames = []
    with open(root + "data_odometry_poses/dataset/poses/07.txt", "w") as f:
        for i in range(n):
            L, R = S.synth_render(p, i)
            frames.append((L, R))
            Image.fromarray(L).save(root + "07/image_0/%06d.png" % i)
            Image.fromarray(R).save(root + "07/image_1/%06d.png" % i, compress_level=1)
            f.write(" ".join("%.12e" % v for v in S.synth_pose(p, i).ravel()) + "\n")
    return p, root, frames


def test_png_pgm_decoder_matches_pil(tmp_path):
    from PIL import Image
    import stereo_vo_amd as S
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    smooth = np.clip(np.cumsum(rng.integers(-3, 4, (64, 80)), axis=1) + 120, 0, 255).astype(np.uint8)  # exercises Sub/Up/Paeth
    for k, img in enumerate((g, smooth)):
        for opt in (False, True):
            f = str(tmp_path / f"a{k}{int(opt)}.png")
            Image.fromarray(img).save(f, optimize=opt)
            assert np.array_equal(S.image_read_gray(f), img)
    rgb = rng.integers(0, 256, (20, 31, 3)).astype(np.uint8)
    f = str(tmp_path / "c.png")
    Image.fromarray(rgb).save(f)
    c = rgb.astype(np.int64)
    exp = ((299 * c[..., 0] + 587 * c[..., 1] + 114 * c[..., 2] + 500) // 1000).astype(np.uint8)
    assert np.array_equal(S.image_read_gray(f), exp)
    Image.fromarray((g.astype(np.uint32) * 257).astype(np.uint16)).save(str(tmp_path / "d.png"))  # 16-bit gray
    assert np.array_equal(S.image_read_gray(str(tmp_path / "d.png")), g)
    with open(tmp_path / "e.pgm", "wb") as fh:
        fh.write(b"P5\n# comment\n53 37\n255\n" + g.tobytes())
    assert np.array_equal(S.image_read_gray(str(tmp_path / "e.pgm")), g)
    with pytest.raises(S.SvoError):
        S.image_read_gray(str(tmp_path / "missing.png"))


def test_poses_file_and_ate(tmp_path):
    import stereo_vo_amd as S
    p, root, _ = _export(tmp_path, n=3, w=64, h=48)
    rt = S.kitti_read_poses(root + "data_odometry_poses/dataset/poses/07.txt")
    assert rt.shape == (3, 3, 4) and np.allclose(rt[2], S.synth_pose(p, 2), atol=1e-11)
    rng = np.random.default_rng(1)
    gt = np.cumsum(rng.normal(0, 1, (50, 3)), axis=0)
    a = 0.7
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    est = (gt - [1, 2, 3]) @ Rz          # a rigidly moved copy: ATE must vanish
    assert S.ate_rmse(est, gt) < 1e-9
    assert S.ate_rmse(est * 0.5, gt, with_scale=True) < 1e-9 and S.ate_rmse(est * 0.5, gt) > 0.1
    noisy = est + rng.normal(0, 0.05, est.shape)
    assert 0.05 < S.ate_rmse(noisy, gt) < 0.12


@pytest.mark.gpu
def test_kitti_driver_matches_oracle_and_ground_truth(ctx, tmp_path):
    import stereo_vo_amd as S
    n = 10
    p, root, frames = _export(tmp_path, n=n)
    pp = S.pipeline_default_params()
    pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
    pp.width, pp.height, pp.min_feature_distance, pp.ba_max_time_s = p.width, p.height, 12.0, 0.0
    traj, st = S.kitti_run(ctx, pp, root, 7, n)
    assert st.frames == n and st.keyframes >= 3
    o = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=p.width, height=p.height, max_corners=300,
                   quality=0.1, min_feature_distance=12.0, parallax_thresh=20.0, window_size=5, max_features=400,
                   ba_max_iterations=50, num_threads=4)
    for i, (L, R) in enumerate(frames):
        r = o.process(L, R)
        q = np.array(list(r.pose7), np.float32)
        if not q[:4].any():
            continue
        w, x, y, z = q[0], -q[1], -q[2], -q[3]
        Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                       [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float32)
        pw = Rm @ (-q[4:])
        assert np.allclose(traj[i][:, 3], pw, rtol=0, atol=1e-5), i   # src/vo_node.cpp:149-150
    # ATE is evaluated at keyframes: between keyframes the published pose is the last keyframe's (SURVEY C-12)
    gt = np.array([S.synth_pose(p, i)[:, 3] for i in range(n)])
    assert 0 <= st.ate_rmse < 0.25, st.ate_rmse                       # ~7 m of travel, 496x160 images
    assert S.ate_rmse(traj[:, :, 3], gt) > st.ate_rmse                # all-frames ATE includes the keyframe lag


def test_draw_track_host_rasteriser():
    """f4 — FeatureTracker::draw_track / get_drawing (src/feature_tracker.cpp:74-91): gray -> RGB, a green arrow of
    thickness 4 per feature from its keyframe position to its current one.  Own rasteriser (visualisation only):
    checked structurally, not against OpenCV pixels."""
    import stereo_vo_amd as S
    g = (np.arange(80 * 120).reshape(80, 120) % 251).astype(np.uint8)
    a = np.array([[20.0, 20.0], [100.0, 60.0], [5.0, 70.0]], np.float32)
    b = np.array([[60.0, 20.0], [100.0, 30.0], [5.0, 70.0]], np.float32)   # horizontal, vertical, zero-length
    img = S.api.draw_track(g, a, b)
    assert img.shape == (80, 120, 3) and img.dtype == np.uint8
    green = (img[..., 0] == 0) & (img[..., 1] == 255) & (img[..., 2] == 0)
    # untouched pixels are the gray value replicated (cvtColor GRAY2RGB)
    assert np.array_equal(img[~green], np.repeat(g[~green][:, None], 3, 1))
    # the shafts: every pixel on the centre lines is green, the band is ~4 px wide
    assert green[20, 20:61].all() and green[30:61, 100].all()
    assert 3 <= green[:, 40].sum() <= 5 and 3 <= green[45, :].sum() <= 5
    # arrow head of the horizontal arrow: two strokes of 0.1 x 40 = 4 px behind the tip, above and below the shaft
    assert green[17, 57] or green[17, 58] or green[16, 57]
    assert green[23, 57] or green[23, 58] or green[24, 57]
    assert green[70, 5]                                    # zero-length arrow still marks its point
    # end points far outside the image are clipped, not walked for ever
    far = S.api.draw_track(g, np.array([[10.0, 10.0]], np.float32), np.array([[1e7, -1e7]], np.float32))
    assert far.shape == (80, 120, 3)
    # nothing to draw: plain gray -> RGB
    assert np.array_equal(S.api.draw_track(g, np.zeros((0, 2)), np.zeros((0, 2))), np.repeat(g[..., None], 3, 2))


@pytest.mark.gpu
def test_pipeline_draw_track(ctx, frames):
    """svo_pipeline_draw_track: arrows from the tracker's keyframe positions to its current positions."""
    import stereo_vo_amd as S
    p, fr = frames
    L, R = np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])
    pp = S.pipeline_default_params()
    pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
    pp.width, pp.height = L.shape[2], L.shape[1]
    pp.ba_max_time_s = 0.0
    pipe = S.Pipeline(ctx, pp)
    res = pipe.process_batch(L[:3], R[:3])
    assert res[0].is_keyframe
    ids, xy = pipe.tracked()
    img = pipe.draw_track(L[0])
    assert img.shape == L[0].shape + (3,)
    green = (img[..., 0] == 0) & (img[..., 1] == 255) & (img[..., 2] == 0)
    assert green.sum() >= len(ids)  # every tracked feature leaves at least its brush stamp
    inside = [(int(round(x)), int(round(y))) for x, y in xy if 2 <= x < L.shape[2] - 2 and 2 <= y < L.shape[1] - 2]
    if not res[1].is_keyframe and not res[2].is_keyframe:
        assert all(green[y, x] for x, y in inside)  # arrow tips sit on the current feature positions
    pipe.close()


def test_decoders_reject_malformed_files(tmp_path):
    """The decoders read files from disk: truncated / hostile headers must be an error, never an out-of-bounds read."""
    import struct
    import zlib
    import stereo_vo_amd as S

    def chunk(t, data):
        return struct.pack(">I", len(data)) + t + data + struct.pack(">I", zlib.crc32(t + data) & 0xFFFFFFFF)
    sig = b"\x89PNG\r\n\x1a\n"
    ihdr = struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 0)
    idat = zlib.compress(b"".join(b"\x00" + bytes([10 * r] * 4) for r in range(4)))
    good = sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")
    bad = {
        "short_ihdr.png": sig + chunk(b"IHDR", ihdr[:5]) + chunk(b"IDAT", idat) + chunk(b"IEND", b"") + b"\0" * 16,
        "no_ihdr.png": sig + chunk(b"IDAT", idat) + chunk(b"IEND", b"") + b"\0" * 32,
        "two_ihdr.png": sig + chunk(b"IHDR", ihdr) + chunk(b"IHDR", ihdr) + chunk(b"IDAT", idat) + chunk(b"IEND", b""),
        "zero_size.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 0, 4, 8, 0, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b""),
        "huge.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 70000, 4, 8, 0, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b""),
        "chunk_len.png": sig + struct.pack(">I", 0xFFFFFFF0) + b"IHDR" + ihdr + b"\0" * 8,
        "neg.pgm": b"P5\n-1 -1\n255\n" + b"\0" * 4,
        "overflow.pgm": b"P5\n99999999999999999999 2\n255\n" + b"\0" * 4,
        "zero.pgm": b"P5\n0 7\n255\n",
        "truncated.pgm": b"P5\n8 8\n255\n" + b"\0" * 10,
    }
    f = tmp_path / "good.png"
    f.write_bytes(good)
    assert np.array_equal(S.image_read_gray(str(f)), np.array([[10 * r] * 4 for r in range(4)], np.uint8))
    for name, blob in bad.items():
        f = tmp_path / name
        f.write_bytes(blob)
        with pytest.raises(S.SvoError):
            S.image_read_gray(str(f))


@pytest.mark.gpu
def test_kitti_driver_200_frames_at_kitti_size_matches_oracle(tmp_path):
    """BASELINE configs[0] at its stated size: the first 200 frames of a KITTI-shaped sequence (1241x376, kitti00
    intrinsics, the reference's own constants: 300 corners, quality 0.1, 30 px, 5-keyframe window) through svo_kitti_run,
    compared with the oracle pipeline on ALL 200 frames (camera-in-world positions, src/vo_node.cpp:149-150)."""
    import stereo_vo_amd as S
    n = 200
    p, root, frames = _export(tmp_path, n=n, w=1241, h=376, focal=718.856)
    pp = S.pipeline_default_params()
    pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
    pp.width, pp.height, pp.ba_max_time_s = p.width, p.height, 0.0
    ctx = S.Context(1241, 376, max_batch=1, max_corners=300, max_candidates=1 << 17, max_features=400)
    traj, st = S.kitti_run(ctx, pp, root, 7, n)
    assert st.frames == n and st.keyframes >= 20
    o = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=p.width, height=p.height, max_corners=300,
                   quality=0.1, min_feature_distance=30.0, parallax_thresh=20.0, window_size=5, max_features=400,
                   ba_max_iterations=50, num_threads=8)
    n_kf = 0
    for i, (L, R) in enumerate(frames):
        r = o.process(L, R)
        n_kf += r.is_keyframe
        q = np.array(list(r.pose7), np.float32)
        if not q[:4].any():
            continue
        w, x, y, z = q[0], -q[1], -q[2], -q[3]
        Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                       [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float32)
        pw = Rm @ (-q[4:])
        assert np.allclose(traj[i][:, 3], pw, rtol=0, atol=1e-4), i
    assert n_kf == st.keyframes
    ctx.close()
