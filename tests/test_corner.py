"""a1 corner detection — cv::goodFeaturesToTrack at reference src/image_processor.cpp:22.
Index sets must be BIT-EXACT between the HIP path and the oracle (SURVEY Appendix A.1 order)."""
import numpy as np
import pytest

import oracle_lib as O


def _img(seed, h, w, kind="noise"):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2)).astype(np.uint8)
        img = np.kron(base, np.ones((8, 8), np.uint8))[:h, :w]
        return (img // 2 + rng.integers(0, 128, (h, w))).astype(np.uint8)
    if kind == "flat":
        return np.full((h, w), 77, np.uint8)
    if kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        return (((yy // 16 + xx // 16) % 2) * 200 + 20).astype(np.uint8)
    raise ValueError(kind)


def test_oracle_basic_properties():
    img = _img(1, 96, 128)
    c = O.corner_detect(img, 50, 0.1, 10.0)
    assert 4 <= len(c) <= 50
    assert np.all(c == np.round(c)) and c[:, 0].min() >= 1 and c[:, 0].max() <= 126
    d = np.linalg.norm(c[:, None] - c[None], axis=2) + np.eye(len(c)) * 1e9
    assert d.min() >= 10.0
    # strongest first
    e = O.corner_response(img)
    v = e[c[:, 1].astype(int), c[:, 0].astype(int)]
    assert np.all(np.diff(v) <= 0)
    assert len(O.corner_detect(_img(0, 64, 64, "flat"), 10, 0.1, 5.0)) == 0


def test_oracle_checker_corners_on_junctions():
    img = _img(0, 96, 96, "checker")
    c = O.corner_detect(img, 100, 0.1, 8.0)
    assert len(c) > 10
    # every corner sits within 2 px of a 16-px lattice junction
    r = np.minimum(c % 16, 16 - c % 16)
    assert r.max() <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("shape,seed", [((96, 128), 1), ((160, 496), 2), ((61, 67), 3), ((376, 1241), 4)])
def test_hip_response_bit_exact(ctx, shape, seed):
    img = _img(seed, *shape)
    assert np.array_equal(ctx.corner_response(img).view(np.uint32), O.corner_response(img).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,seed,maxc,q,md", [
    ((96, 128), 1, 50, 0.1, 10.0), ((160, 496), 2, 300, 0.1, 30.0), ((61, 67), 3, 1000, 0.01, 3.0),
    ((376, 1241), 4, 300, 0.1, 30.0), ((376, 1241), 5, 1500, 0.02, 12.0), ((720, 1280), 6, 4096, 0.01, 6.0),
    ((120, 160), 7, 7, 0.1, 12.5), ((120, 160), 8, 500, 0.05, 0.0)])
def test_hip_corners_bit_exact(ctx, shape, seed, maxc, q, md):
    img = _img(seed, *shape)
    got = ctx.corner_detect(img, maxc, q, md)
    exp = O.corner_detect(img, maxc, q, md)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.gpu
def test_hip_synthetic_frames_and_edge_cases(ctx, frames):
    _, fr = frames
    for left, right in fr:
        for im in (left, right):
            assert np.array_equal(ctx.corner_detect(im, 300, 0.1, 30.0), O.corner_detect(im, 300, 0.1, 30.0))
    assert len(ctx.corner_detect(_img(0, 64, 64, "flat"), 10, 0.1, 5.0)) == 0
    ck = _img(0, 96, 96, "checker")  # plateaus / exact ties exercise the (value, index) total order
    assert np.array_equal(ctx.corner_detect(ck, 100, 0.1, 8.0), O.corner_detect(ck, 100, 0.1, 8.0))
