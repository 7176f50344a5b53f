"""a7 stereo disparity, a8 triangulation, a6 dedup, a3 pyramid/LK/track_features.
Integer / index outputs must be bit-exact vs the oracle; float positions are produced by the same
declared operation order on both sides and are compared bit-exact as well."""
import numpy as np
import pytest

import oracle_lib as O


def _pair(frames, i=0):
    p, fr = frames
    return p, fr[i][0], fr[i][1]


# ------------------------------------------------------------------ CPU: oracle self-consistency
def test_oracle_sparse_equals_dense(frames):
    _, L, R = _pair(frames)
    d = O.stereo_bm(L, R)
    ys, xs = np.mgrid[0:L.shape[0]:7, 0:L.shape[1]:5]
    xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32) + 0.4  # truncation to int (C-13)
    s = O.stereo_disparity_at(L, R, xy)
    assert np.array_equal(s, d[ys.ravel(), xs.ravel()].astype(np.float32) / 16.0)
    assert (d[:10] == -16).all() and (d[:, :57] == -16).all() and (d[:, -10:] == -16).all()
    assert (d > 0).mean() > 0.2


def test_oracle_disparity_matches_geometry(frames):
    """Known answer from the scene: a textured fronto-parallel plane at depth Z has disparity f*b/Z."""
    h, w = 120, 320
    rng = np.random.default_rng(5)
    tex = np.kron(rng.integers(0, 255, (h // 4 + 1, w // 4 + 30)), np.ones((4, 4))).astype(np.uint8)
    L = tex[:h, 20:20 + w]
    R = tex[:h, 20 + 13:20 + 13 + w]  # right image = left shifted by d = 13 px
    d = O.stereo_bm(L, R)
    valid = d[10:-10, 57:-10]
    # StereoBM's fixed-point rounding ((.. + 15) >> 4) biases the sub-pixel term upward by < 1/16 px
    assert ((valid == 13 * 16) | (valid == 13 * 16 + 1)).mean() > 0.95


def test_oracle_prefilter_rows_and_odd_height():
    img = (np.arange(7 * 9).reshape(7, 9) * 5 % 251).astype(np.uint8)
    p = O.stereo_prefilter(img)
    assert (p[:, 0] == 31).all() and (p[:, -1] == 31).all() and (p[6] == 31).all()


def test_oracle_triangulate_closed_form():
    f, cx, cy, b = 718.856, 607.19, 185.21, 0.537
    xy = np.array([[100.0, 50.0], [700.0, 200.0], [5.0, 5.0]], np.float32)
    disp = np.array([10.0, -1.0, 0.5], np.float32)
    k2, k3, ki = O.triangulate(xy, disp, np.eye(4, dtype=np.float32), f, cx, cy, b)
    assert list(ki) == [0, 2]
    for (x, y), d, X in zip(k2, disp[ki], k3):
        assert np.allclose(X, [(x - cx) * b / d, (y - cy) * b / d, b * f / d], rtol=1e-5)


def test_oracle_dedup():
    det = np.array([[10, 10], [100, 100], [40, 10], [39.9, 10]], np.float32)
    trk = np.array([[10, 10]], np.float32)
    assert np.array_equal(O.dedup(det, trk, 30.0), det[[1, 2]])  # dist 30.0 is NOT < 30 -> kept


def test_oracle_pyramid_and_lk_on_shift():
    rng = np.random.default_rng(2)
    base = rng.integers(0, 255, (40, 60)).astype(np.float32)
    big = np.kron(base, np.ones((6, 6), np.float32))
    k = np.ones(5) / 5
    big = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, big)
    big = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, big)
    A = big[20:200, 20:320].astype(np.uint8)
    B = big[17:197, 15:315].astype(np.uint8)  # content moves +5 px in x, +3 px in y
    pyr = O.build_pyramid(A)
    assert [p.shape for p in pyr] == [(180, 300), (90, 150), (45, 75), (23, 38)]
    pts = np.array([[x, y] for y in range(40, 150, 22) for x in range(40, 260, 31)], np.float32)
    out, st = O.lk_track(A, B, pts)
    assert st.mean() > 0.8
    err = np.abs(out[st > 0] - pts[st > 0] - [5, 3])
    assert np.median(err) < 0.1
    kxy, kidx, av = O.track_features(A, B, pts, pts)
    assert len(kidx) >= 0.7 * len(pts) and abs(av * len(pts) / max(len(kidx), 1) - np.hypot(5, 3)) < 0.2


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
def test_hip_stereo_sparse_and_dense_bit_exact(ctx, frames):
    for i in range(2):
        _, L, R = _pair(frames, i)
        d_o = O.stereo_bm(L, R)
        assert np.array_equal(ctx.stereo_bm(L, R), d_o)
        c = O.corner_detect(L, 300, 0.05, 8.0)
        assert len(c) > 50
        assert np.array_equal(ctx.stereo_disparity_at(L, R, c), O.stereo_disparity_at(L, R, c))
        ys, xs = np.mgrid[0:L.shape[0]:3, 0:L.shape[1]:3]
        xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32) + 0.3
        got = ctx.stereo_disparity_at(L, R, xy)
        assert np.array_equal(got, d_o[ys.ravel(), xs.ravel()].astype(np.float32) / 16.0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,nd,blk", [((75, 131), 32, 11), ((64, 200), 64, 21), ((61, 90), 16, 5)])
def test_hip_stereo_odd_sizes(ctx, shape, nd, blk):
    rng = np.random.default_rng(shape[0])
    tex = np.kron(rng.integers(0, 255, (shape[0] // 3 + 1, shape[1] // 3 + 30)), np.ones((3, 3))).astype(np.uint8)
    L = np.ascontiguousarray(tex[:shape[0], 12:12 + shape[1]])
    R = np.ascontiguousarray(tex[:shape[0], 19:19 + shape[1]])
    assert np.array_equal(ctx.stereo_bm(L, R, nd, blk), O.stereo_bm(L, R, nd, blk))


@pytest.mark.gpu
def test_hip_triangulate_dedup_bit_exact(ctx):
    rng = np.random.default_rng(11)
    n = 3000
    xy = rng.uniform(0, 1241, (n, 2)).astype(np.float32)
    disp = np.where(rng.random(n) < 0.3, -1.0, rng.uniform(0.0625, 47, n)).astype(np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.array([[0.9998, -0.01, 0.02], [0.01, 0.9999, 0.003], [-0.02, -0.003, 0.9998]], np.float32)
    pose[:3, 3] = [0.3, -0.1, 2.5]
    a = ctx.triangulate(xy, disp, pose, 718.856, 607.1928, 185.2157, 0.537165718864418)
    b = O.triangulate(xy, disp, pose, 718.856, 607.1928, 185.2157, 0.537165718864418)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    e = ctx.triangulate(xy[:0], disp[:0], pose, 1.0, 0.0, 0.0, 1.0)
    assert len(e[0]) == 0
    det = np.round(rng.uniform(0, 1241, (2500, 2))).astype(np.float32)
    trk = rng.uniform(0, 1241, (1800, 2)).astype(np.float32)
    trk[:50] = det[:50] + np.array([30.0, 0.0], np.float32)  # exactly-at-threshold cases
    assert np.array_equal(ctx.dedup(det, trk, 30.0), O.dedup(det, trk, 30.0))
    assert np.array_equal(ctx.dedup(det, trk[:0], 30.0), det)


@pytest.mark.gpu
def test_hip_pyramid_bit_exact(ctx, frames):
    _, L, _ = _pair(frames)
    for img in (L, L[:61, :77].copy(), L[:8, :8].copy()):
        for a, b in zip(ctx.build_pyramid(img), O.build_pyramid(img)):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_hip_fused_pyramid_levels_1_to_3_from_one_read_of_level_0():
    """Round 5: pyr_build_kernel forms levels 1-3 of an image from ONE staged read of level 0 (a workgroup = a 128 x 64 tile of level 1,
    the halo of every next level recomputed in LDS, REFLECT_101 at every level's own border).  Bit-exact against the oracle at the
    sizes the configs use (1241 x 376, 1280 x 720) and at sizes that put image borders, odd halves and one-pixel-wide remainders at
    every position relative to the tiling."""
    import stereo_vo_amd as S
    rng = np.random.default_rng(17)
    ctx2 = S.Context(1280, 720, max_batch=2, max_corners=256, max_candidates=1 << 12, max_features=256)
    for (h, w) in ((376, 1241), (720, 1280), (64, 64), (65, 257), (129, 259), (130, 513), (255, 1025), (257, 771), (131, 77), (200, 253), (127, 255)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        img[::7, ::5] = 255; img[3::11, 1::9] = 0
        for lvl, (a, b) in enumerate(zip(ctx2.build_pyramid(img), O.build_pyramid(img))):
            assert np.array_equal(a, b), (h, w, lvl, np.argwhere(a != b)[:4])
    ctx2.close()


@pytest.mark.gpu
def test_hip_lk_and_track_bit_exact(ctx, frames):
    p, fr = frames
    for i in range(3):
        A, B = fr[i][0], fr[i + 1][0]
        pts = O.corner_detect(A, 400, 0.05, 8.0)
        # add border / out-of-image / low-texture points: exercise every status path
        extra = np.array([[0.0, 0.0], [A.shape[1] - 1.0, A.shape[0] - 1.0], [3.5, 77.25], [-30.0, 10.0],
                          [A.shape[1] + 40.0, 20.0], [250.3, 5.1]], np.float32)
        pts = np.concatenate([pts, extra]).astype(np.float32)
        out_g, st_g = ctx.lk_track(A, B, pts)
        out_o, st_o = O.lk_track(A, B, pts)
        assert np.array_equal(st_g, st_o)
        assert np.array_equal(out_g.view(np.uint32), out_o.view(np.uint32))
        init = pts + np.float32(0.5)
        kg = ctx.track_features(A, B, pts, init)
        ko = O.track_features(A, B, pts, init)
        assert np.array_equal(kg[1], ko[1]) and np.array_equal(kg[0].view(np.uint32), ko[0].view(np.uint32))
        assert np.float32(kg[2]).view(np.uint32) == np.float32(ko[2]).view(np.uint32)
        assert 0 < len(ko[1]) < len(pts)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,shift", [(96, 64, (1.6, -0.7)), (51, 37, (-2.3, 1.1)), (200, 31, (6.5, 0.4)), (33, 120, (0.3, -7.2))])
def test_hip_lk_small_images_borders_and_restaging(ctx, w, h, shift):
    """The LK kernel stages its tiles two ways: unaligned dword loads when the tile lies inside the pyramid level, byte
    loads through reflect-101 otherwise.  Small and odd-sized images make the coarse levels smaller than a 24x24 / 32x32
    tile (every tile reflects, some several times), a dense grid of points including the borders and beyond hits the
    x0 + N == w boundary of the fast path at level 0, and a large shift makes windows walk out of the staged region
    (restaging).  Positions and status must equal the oracle's bit for bit."""
    rng = np.random.default_rng(w * 1000 + h)
    big = rng.integers(0, 256, size=(h + 40, w + 40)).astype(np.float32)
    # smooth the noise a little so that LK has a gradient to follow, then cut two shifted views
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16
    for ax in (0, 1):
        big = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, big)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)

    def view(dx, dy):
        x, y = xx + 20 + dx, yy + 20 + dy
        x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
        fx, fy = x - x0, y - y0
        v = (big[y0, x0] * (1 - fx) * (1 - fy) + big[y0, x0 + 1] * fx * (1 - fy) + big[y0 + 1, x0] * (1 - fx) * fy +
             big[y0 + 1, x0 + 1] * fx * fy)
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)

    A, B = view(0.0, 0.0), view(*shift)
    gx, gy = np.meshgrid(np.arange(-2.0, w + 3.0, 6.5), np.arange(-2.0, h + 3.0, 5.25))
    pts = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    out_g, st_g = ctx.lk_track(A, B, pts)  # the session context (1280x720 capacity) takes any smaller image
    out_o, st_o = O.lk_track(A, B, pts)
    assert np.array_equal(st_g, st_o)
    assert np.array_equal(out_g.view(np.uint32), out_o.view(np.uint32))
    assert st_o.sum() > 0
