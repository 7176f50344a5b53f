"""The product's LM step control (host/lm.cpp, svo_lm_solve — what every rank of a sharded solve executes) checked
WITHOUT a GPU: plugged onto the oracle's passes it must reproduce the oracle's own plain LM loop bit for bit, although it
linearises speculatively and exchanges once per iteration (reference semantics: ceres::Solve at
src/bundle_adjuster.cpp:140)."""
import os

import numpy as np
import pytest

import ba_problem as BP
import oracle_lib as O


@pytest.mark.parametrize("seed,K,N,dense", [(1, 5, 300, False), (2, 6, 800, False), (5, 12, 400, True), (6, 2, 50, False)])
def test_product_lm_control_equals_the_oracle_loop(seed, K, N, dense):
    p = BP.make_problem(seed, K, N, dense=dense)
    args = (p["poses0"], p["points0"], p["op"], p["oj"], p["uv"], BP.F, BP.CX, BP.CY)
    po, pto, so = O.ba_solve(*args)
    poses, pts, s, st, exchanges = O.product_lm_over_oracle_passes(*args)
    assert (s.iterations, s.successful_steps, s.termination) == (so["iterations"], so["successful"], so["termination"])
    assert s.initial_cost == so["initial_cost"] and s.final_cost == so["final_cost"]
    assert np.array_equal(poses, po) and np.array_equal(pts, pto)  # bit for bit: speculation changes no arithmetic
    # every iteration asked the backend for the next linearisation along with pass B and could use it, unless it was
    # the last one: ONE host round trip per LM iteration, no stand-alone pass A besides the very first
    assert st.step_calls == s.iterations
    misses = st.speculations - st.speculation_hits  # the terminating iteration + mispredicted same-sweep radii
    assert st.speculations == s.iterations and misses <= 3
    assert st.linearize_calls <= 1 + misses
    # collectives: every stand-alone pass A; per step one (same sweep) or two (payload2, decision, payload1)
    chained = st.speculations - st.single_exchange
    assert exchanges == st.linearize_calls + (st.step_calls - chained) + 2 * chained


def test_speculation_can_be_disabled_without_changing_results(monkeypatch):
    p = BP.make_problem(9, 6, 500)
    args = (p["poses0"], p["points0"], p["op"], p["oj"], p["uv"], BP.F, BP.CX, BP.CY)
    a = O.product_lm_over_oracle_passes(*args)
    monkeypatch.setenv("SVO_LM_NO_SPECULATION", "1")
    b = O.product_lm_over_oracle_passes(*args)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2].final_cost == b[2].final_cost
    assert b[3].speculations == 0 and a[3].speculation_hits > 0
    assert b[3].linearize_calls >= b[2].successful_steps and a[3].linearize_calls <= 3


def test_iteration_cap_and_bad_arguments():
    import stereo_vo_amd as S
    p = BP.make_problem(3, 5, 200)
    args = (p["poses0"], p["points0"], p["op"], p["oj"], p["uv"], BP.F, BP.CX, BP.CY)
    po, pto, so = O.ba_solve(*args, max_iterations=2)
    poses, pts, s, st, _ = O.product_lm_over_oracle_passes(*args, max_iterations=2)
    assert s.iterations == so["iterations"] == 2 and s.termination == so["termination"] == 1
    assert np.array_equal(poses, po) and np.array_equal(pts, pto)
    assert S.lib().svo_lm_solve(0, None, None, None, None, None) == -1
