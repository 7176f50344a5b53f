"""a5 PnP-RANSAC — cv::solvePnPRansac call at reference src/image_processor.cpp:76-80, restated
deterministically (oracle/ora_pnp.cpp).  Inlier index sets bit-exact; pose bit-exact as well because
every reduction uses the declared order."""
import numpy as np
import pytest

import oracle_lib as O

F, CX, CY = 718.856, 607.1928, 185.2157


def _scene(seed, n, outlier_frac=0.2, noise=0.3):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(6, 40, n)], 1)
    rv = np.array([0.01, -0.03, 0.005]) * rng.uniform(0.5, 1.5)
    tv = np.array([0.05, -0.02, -0.8]) * rng.uniform(0.5, 1.5)
    th = np.linalg.norm(rv); k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    Xc = X @ R.T + tv
    uv = np.stack([F * Xc[:, 0] / Xc[:, 2] + CX, F * Xc[:, 1] / Xc[:, 2] + CY], 1)
    uv += rng.normal(0, noise, uv.shape)
    out = rng.random(n) < outlier_frac
    uv[out] += rng.uniform(20, 80, (out.sum(), 2)) * rng.choice([-1, 1], (out.sum(), 2))
    return X.astype(np.float32), uv.astype(np.float32), rv, tv, ~out


def test_oracle_recovers_pose_and_inliers():
    X, uv, rv, tv, good = _scene(1, 300)
    r, t, inl = O.pnp_ransac(X, uv, F, CX, CY, np.zeros(3), np.zeros(3))
    assert np.linalg.norm(r - rv) < 2e-3 and np.linalg.norm(t - tv) < 2e-2
    mask = np.zeros(len(X), bool); mask[inl] = True
    assert (mask & good).sum() >= 0.97 * good.sum() and (mask & ~good).sum() <= 2
    assert np.all(np.diff(inl) > 0)


def test_oracle_degenerate_inputs():
    X, uv, *_ = _scene(2, 4)
    r, t, inl = O.pnp_ransac(X, uv, F, CX, CY, np.ones(3) * 0.01, np.ones(3))
    assert len(inl) == 0 and np.allclose(r, 0.01) and np.allclose(t, 1.0)  # < 5 points: unchanged
    X, uv, *_ = _scene(3, 50, outlier_frac=1.0)
    r, t, inl = O.pnp_ransac(X, uv, F, CX, CY, np.zeros(3), np.zeros(3))
    assert len(inl) < 12


def test_iteration_cap_with_declared_arithmetic_equals_the_libm_formula():
    """RANSACUpdateNumIters with the declared logarithm / power (so that the kernels can take the cut-off themselves, bit for
    bit) against the libm formula OpenCV uses: the logarithm within 5e-16 relative, the cap identical for every inlier count
    of every set size the pipelines see."""
    import math
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(1e-300, 1.0, 2000), 10.0 ** rng.uniform(-300, 0, 2000), [1.0, 0.5, 0.01, 2.2250738585072014e-308]])
    for x in xs:
        assert abs(O.pnp_det_log(float(x)) - math.log(x)) <= 5e-16 * abs(math.log(x)) + 1e-300

    def libm_cap(p, ep, k, mx):
        p = min(max(p, 0.0), 1.0); ep = min(max(ep, 0.0), 1.0)
        num = max(1.0 - p, 2.2250738585072014e-308)
        den = 1.0 - (1.0 - ep) ** k
        if den < 2.2250738585072014e-308:
            return 0
        num, den = math.log(num), math.log(den)
        return mx if den >= 0 or -num >= mx * (-den) else int(round(num / den))  # Python rounds half to even, like lrint
    for n in (5, 7, 64, 300, 731, 1500, 3000):
        for cnt in range(5, n + 1, max(1, n // 97)):
            for mx in (100, 37, 3):
                assert O.pnp_update_num_iters(0.99, (n - cnt) / n, 5, mx) == libm_cap(0.99, (n - cnt) / n, 5, mx), (n, cnt, mx)
    assert O.pnp_update_num_iters(0.99, 0.0, 5, 100) == 0 and O.pnp_update_num_iters(0.99, 1.0, 5, 100) == 100


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,of", [(1, 300, 0.2), (4, 1500, 0.35), (5, 64, 0.0), (6, 7, 0.1), (7, 2000, 0.6)])
def test_hip_pnp_bit_exact(ctx, seed, n, of):
    X, uv, rv, tv, good = _scene(seed, n, of)
    r0, t0 = np.array([0.002, 0.001, -0.003]), np.array([0.01, 0.0, -0.1])
    rg, tg, ig = ctx.pnp_ransac(X, uv, F, CX, CY, r0, t0)
    ro, to, io = O.pnp_ransac(X, uv, F, CX, CY, r0, t0)
    assert np.array_equal(ig, io)
    assert np.allclose(rg, ro, rtol=0, atol=1e-12) and np.allclose(tg, to, rtol=0, atol=1e-12)
    if len(io) > 20:
        assert np.linalg.norm(rg - rv) < 5e-3


@pytest.mark.gpu
def test_hip_pnp_too_few_points(ctx):
    X, uv, *_ = _scene(2, 4)
    r, t, inl = ctx.pnp_ransac(X, uv, F, CX, CY, np.ones(3) * 0.01, np.ones(3))
    assert len(inl) == 0 and np.allclose(r, 0.01) and np.allclose(t, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("iterations", [1, 3, 4, 5, 37, 100, 101])
def test_hip_pnp_hypothesis_counts_that_do_not_fill_the_last_workgroup(ctx, iterations):
    """The launch runs four hypotheses per workgroup: counts that are not a multiple of four (and a single workgroup) leave
    wavefronts without a hypothesis; bookkeeping and refinement must still be the oracle's."""
    X, uv, *_ = _scene(9, 400, 0.3)
    r0, t0 = np.array([0.002, 0.001, -0.003]), np.array([0.01, 0.0, -0.1])
    rg, tg, ig = ctx.pnp_ransac(X, uv, F, CX, CY, r0, t0, iterations=iterations)
    ro, to, io = O.pnp_ransac(X, uv, F, CX, CY, r0, t0, iterations=iterations)
    assert np.array_equal(ig, io)
    assert np.allclose(rg, ro, rtol=0, atol=1e-12) and np.allclose(tg, to, rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_hip_pnp_without_a_model_leaves_the_pose_alone(ctx):
    """No hypothesis reaches five inliers (pure outliers at a 0.05 px threshold): the oracle returns no inliers and the guess
    unchanged; so must the launch whose last workgroup finds best < 0 and skips the refinement."""
    X, uv, *_ = _scene(3, 60, outlier_frac=1.0)
    r0, t0 = np.array([0.3, -0.2, 0.1]), np.array([1.0, 2.0, 3.0])
    ro, to, io = O.pnp_ransac(X, uv, F, CX, CY, r0, t0, reproj_err=0.05)
    rg, tg, ig = ctx.pnp_ransac(X, uv, F, CX, CY, r0, t0, reproj_err=0.05)
    assert len(io) == 0 and len(ig) == 0
    assert np.array_equal(rg, ro) and np.array_equal(tg, to) and np.array_equal(rg, r0) and np.array_equal(tg, t0)


def test_declared_trigonometry_of_the_pose_conversions_is_within_a_few_ulp_of_libm():
    """Round 5: cv::Rodrigues and its inverse around solvePnPRansac (src/image_processor.cpp:84-92,130-134) use declared
    arithmetic (host/det_trig.h, restated in oracle/ora_trig.h) so that they can run on the device with the host's bits.  Bound
    against libm: atan2 in the first quadrant and sin / cos on [0, pi] within 4 ulp of the result's scale; rvec -> quaternion ->
    rvec reproduces rvec to 1e-15 (every caller stores these as float)."""
    import math
    rng = np.random.default_rng(11)
    pts = [(1.0, 1.0), (0.0, 1.0), (1.0, 0.0), (1e-12, 1.0), (1.0, 1e-12), (0.41421356237309503, 1.0), (0.4142135623730951, 1.0), (1e-300, 1e-300)]
    pts += [tuple(v) for v in rng.uniform(0, 1, (3000, 2))] + [(float(a), float(b)) for a, b in zip(10.0 ** rng.uniform(-12, 0, 500), 10.0 ** rng.uniform(-12, 0, 500))]
    for y, x in pts:
        d = O.det_atan2_q1(y, x)
        assert abs(d - math.atan2(y, x)) <= 4 * np.spacing(max(abs(d), 1e-300)), (y, x, d, math.atan2(y, x))
    for x in list(rng.uniform(0, math.pi, 3000)) + [0.0, 1e-9, 0.5, 0.5000000001, 1.0, math.pi / 2, math.pi]:
        s, c = O.det_sincos(float(x))
        assert abs(s - math.sin(x)) <= 8 * np.spacing(1.0) and abs(c - math.cos(x)) <= 8 * np.spacing(1.0), (x, s, c)
    for _ in range(500):
        rv = rng.normal(size=3) * rng.choice([1e-14, 1e-3, 0.3, 1.0])
        if np.linalg.norm(rv) > 3.0:
            continue
        q, back = O.det_rvec_quat_roundtrip(rv)
        assert abs(np.linalg.norm(q) - 1.0) < 1e-14 and np.allclose(back, rv, rtol=0, atol=1e-15 + 4e-16 * np.linalg.norm(rv))
