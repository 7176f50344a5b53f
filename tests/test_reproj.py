"""a11 ReprojectionFactor::Evaluate — reference src/reprojection_factor.cpp:10-88.

Golden vectors (tests/golden/reproj_golden.json) were produced by evaluating the reference's own
scalar expressions (gen_reproj_golden.py); the oracle is pinned to them on CPU, the HIP kernel is
checked against both on the GPU.  Tolerance: 1e-9 relative to the largest entry of each block
(float64 arithmetic, different but algebraically identical expression trees)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reproj_golden.json")
RTOL = 1e-9


def _cases():
    return json.load(open(GOLD))["cases"]


def _check(c, r, jq, jx):
    checks = [(r, c["r"]), (jq, c["jpose"]), (jx, c["jpoint"])]
    if c.get("r_text") is not None:
        # the reference's own scalar residual text (src/reprojection_factor.cpp:53-54), unit-quaternion cases only
        checks.append((r, c["r_text"]))
    for got, exp in checks:
        exp = np.asarray(exp)
        assert np.max(np.abs(got - exp)) <= RTOL * max(1.0, np.max(np.abs(exp)))


def test_golden_holds_text_evaluated_residuals():
    cs = _cases()
    assert sum(c["r_text"] is not None for c in cs) == 128  # every unit-quaternion case


def test_oracle_matches_reference_golden():
    for c in _cases():
        r, jq, jx = O.reproj_eval([c["pose"]], [c["point"]], [c["obs"]], c["focal"], c["cx"], c["cy"])
        _check(c, r[0], jq[0], jx[0])
        assert jq[0][5] == 0.0 and jq[0][11] == 0.0  # setZero entries, src/reprojection_factor.cpp:61


def test_oracle_jacobian_is_the_derivative():
    rng = np.random.default_rng(7)
    for _ in range(20):
        q = rng.normal(size=4); q *= rng.uniform(0.5, 2.0) / np.linalg.norm(q)
        pose = np.concatenate([q, rng.uniform(-2, 2, 3)])
        pt = rng.uniform(-3, 3, 3) + np.array([0, 0, 12.0])
        obs = rng.uniform(0, 300, 2)
        r0, jq, jx = O.reproj_eval([pose], [pt], [obs], 500.0, 320.0, 240.0)
        if abs(r0).max() > 1e5:
            continue
        for k in range(7):
            h = 1e-6
            pp, pm = pose.copy(), pose.copy(); pp[k] += h; pm[k] -= h
            fd = (O.reproj_eval([pp], [pt], [obs], 500.0, 320.0, 240.0)[0][0] -
                  O.reproj_eval([pm], [pt], [obs], 500.0, 320.0, 240.0)[0][0]) / (2 * h)
            assert np.allclose(fd, jq[0].reshape(2, 7)[:, k], rtol=1e-5, atol=1e-4 * (1 + abs(fd).max()))


def test_null_jacobian_conventions_oracle():
    c = _cases()[0]
    r, jq, jx = O.reproj_eval([c["pose"]], [c["point"]], [c["obs"]], c["focal"], c["cx"], c["cy"], False, False)
    assert jq is None and jx is None and np.allclose(r[0], c["r"], rtol=1e-9)


@pytest.mark.gpu
def test_hip_matches_golden_and_oracle(ctx):
    cs = _cases()
    for cam in {(c["focal"], c["cx"], c["cy"]) for c in cs}:
        sub = [c for c in cs if (c["focal"], c["cx"], c["cy"]) == cam]
        pose = np.array([c["pose"] for c in sub]); pt = np.array([c["point"] for c in sub])
        obs = np.array([c["obs"] for c in sub])
        r, jq, jx = ctx.reproj_eval(pose, pt, obs, *cam)
        ro, jqo, jxo = O.reproj_eval(pose, pt, obs, *cam)
        for i, c in enumerate(sub):
            _check(c, r[i], jq[i], jx[i])
        # same expression tree, no FMA contraction on either side: expect (near) bit equality
        for a, b in ((r, ro), (jq, jqo), (jx, jxo)):
            assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))
        assert np.all(jq[:, 5] == 0.0) and np.all(jq[:, 11] == 0.0)


@pytest.mark.gpu
def test_hip_null_jacobians_and_empty(ctx):
    c = _cases()[3]
    r, jq, jx = ctx.reproj_eval([c["pose"]], [c["point"]], [c["obs"]], c["focal"], c["cx"], c["cy"], False, True)
    assert jq is None and np.allclose(jx[0], c["jpoint"], rtol=1e-9, atol=1e-9)
    r, jq, jx = ctx.reproj_eval(np.zeros((0, 7)), np.zeros((0, 3)), np.zeros((0, 2)), 1.0, 0.0, 0.0)
    assert r.shape == (0, 2)


@pytest.mark.gpu
def test_hip_large_batch_linearity(ctx):
    """Full-size property: 1M observations; residual is affine in obs, Jacobians independent of it."""
    rng = np.random.default_rng(3)
    n = 1_000_000
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    pose = np.concatenate([0.05 * q + np.array([1, 0, 0, 0]), rng.uniform(-1, 1, (n, 3))], axis=1)
    pt = rng.uniform(-5, 5, (n, 3)) + np.array([0, 0, 20.0])
    obs = rng.uniform(0, 1000, (n, 2))
    r1, jq1, jx1 = ctx.reproj_eval(pose, pt, obs, 718.856, 607.1928, 185.2157)
    r2, jq2, jx2 = ctx.reproj_eval(pose, pt, obs + 3.0, 718.856, 607.1928, 185.2157)
    assert np.allclose(r1 - r2, 3.0, rtol=0, atol=1e-9)
    assert np.array_equal(jq1, jq2) and np.array_equal(jx1, jx2)
    idx = rng.integers(0, n, 2000)
    ro, jqo, jxo = O.reproj_eval(pose[idx], pt[idx], obs[idx], 718.856, 607.1928, 185.2157)
    assert np.allclose(r1[idx], ro, rtol=1e-12, atol=1e-9) and np.allclose(jq1[idx], jqo, rtol=1e-11, atol=1e-9)
