"""Pins for the first-party constants and the Q matrix (SURVEY §8c "what IS pinned by first-party source").

tests/golden/constants_golden.json is extracted by regular expressions from the reference's source text
(tests/golden/gen_constants_golden.py: src/image_processor.cpp:22,23,63,80,174-176,184-189,194, src/feature_tracker.cpp:24-26,
47,53,81, src/vo_node.cpp:33-36, src/bundle_adjuster.hpp:75, src/bundle_adjuster.cpp:9-12).  Three independent statements
of the same literals must agree with it: the product's compiled-in constants (svo_reference_constants, used by the
kernels and the host chain), the product's default pipeline parameters, and the CPU oracle's constants."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "constants_golden.json")))


def _same(a, b):
    # literals such as 0.1, 0.99, 1e-2 are doubles in the reference text and float or double in the code: equal after
    # rounding to float where the code holds a float
    return a == b or np.float32(a) == np.float32(b)


def test_product_constants_equal_the_reference_text():
    from stereo_vo_amd import api
    got = api.reference_constants()
    gold = GOLD["constants"]
    for name, val in got.items():
        assert name in gold, name
        assert _same(val, gold[name]), (name, val, gold[name])
    # every extracted literal that the hot path consumes is covered (the rest: ROS queue depth, flags fixed by design)
    not_compiled_in = {"pnp_use_extrinsic_guess", "lk_flags", "image_queue_size", "ba_linear_solver_is_dense_schur"}
    assert set(gold) - set(got) == not_compiled_in
    assert gold["pnp_use_extrinsic_guess"] == 1.0 and gold["lk_flags"] == 0.0 and gold["ba_linear_solver_is_dense_schur"] == 1.0


def test_default_pipeline_params_equal_the_reference_text():
    import stereo_vo_amd as S
    p, g = S.pipeline_default_params(), GOLD["constants"]
    assert p.max_corners == g["gftt_max_corners"] and p.quality == g["gftt_quality"]
    assert p.min_feature_distance == g["min_feature_distance"] and p.parallax_thresh == g["parallax_thresh"]
    assert p.window_size == g["sliding_window_size"] and p.max_features == g["max_features"]
    assert p.ba_max_time_s == g["ba_max_solver_time_s"]
    o = S.api.ba_default_options()
    assert o.max_time_s == g["ba_max_solver_time_s"] and o.max_features == g["max_features"]


def test_oracle_constants_equal_the_reference_text():
    got, gold = O.reference_constants(), GOLD["constants"]
    for name, val in got.items():
        assert _same(val, gold[name]), (name, val, gold[name])


def test_oracle_triangulation_matches_the_extracted_q_matrix():
    """Known answers built from the SIX Q.at<float>(i,j) assignments of src/image_processor.cpp:184-189 (evaluated as
    text by the generator): X = pose * Q * [x y d 1]^T, de-homogenised.  float32 storage of Q / pose / result on the
    oracle side vs float64 evaluation in the fixture: 2e-6 relative."""
    for q in GOLD["q"]:
        f, cx, cy, b = (float(np.float32(q[k])) for k in ("focal", "cx", "cy", "baseline"))
        Q = np.array(q["Q"]).reshape(4, 4)
        assert np.allclose(Q, [[1 / f, 0, 0, -cx / f], [0, 1 / f, 0, -cy / f], [0, 0, 0, 1], [0, 0, 1 / (b * f), 0]], rtol=1e-6)
    for c in GOLD["triangulation"]:
        q = GOLD["q"][c["camera"]]
        kxy, xyz, kidx = O.triangulate([[c["x"], c["y"]]], [c["disp"]], c["pose16"], q["focal"], q["cx"], q["cy"], q["baseline"])
        assert len(kidx) == 1 and np.allclose(xyz[0], c["xyz"], rtol=2e-6, atol=1e-5), (xyz, c["xyz"])
    # disparity 0 and negative are dropped (:194, exclusive bound)
    q = GOLD["q"][0]
    kxy, xyz, kidx = O.triangulate([[100, 100], [200, 100], [300, 100]], [0.0, -1.0, 2.0], np.eye(4).ravel(), q["focal"], q["cx"], q["cy"], q["baseline"])
    assert list(kidx) == [2]


@pytest.mark.gpu
def test_hip_triangulation_matches_the_extracted_q_matrix(ctx):
    for c in GOLD["triangulation"]:
        q = GOLD["q"][c["camera"]]
        kxy, xyz, kidx = ctx.triangulate([[c["x"], c["y"]]], [c["disp"]], c["pose16"], q["focal"], q["cx"], q["cy"], q["baseline"])
        assert len(kidx) == 1 and np.allclose(xyz[0], c["xyz"], rtol=2e-6, atol=1e-5), (xyz, c["xyz"])
