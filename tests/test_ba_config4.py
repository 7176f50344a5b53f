"""BASELINE configs[3] AT ITS STATED SIZE: synthetic 50 k-landmark / 20-keyframe bundle adjustment (43 k landmarks and
444 k observations survive the visibility test) — the bulk accumulation paths of csrc/ba.hip (f64-MFMA rank-3 Schur updates,
LDS atomics), including the 256-workgroup multi-chunk loop of ba_linearize_mfma_kernel that smaller problems never reach,
and the landmark sharding of SURVEY §8e with the payloads summed by the all-reduce callback.
Reference semantics: ceres::Solve with DENSE_SCHUR, src/bundle_adjuster.cpp:9-12,140."""
import os
import sys
import threading

import numpy as np
import pytest

import ba_problem as BP
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def cfg4():
    from tools import bench_ba
    p = bench_ba.make_problem()
    assert len(p["op"]) > 400_000 and len(p["points0"]) > 40_000 and p["poses0"].shape[0] == 20
    return p


@pytest.fixture(scope="module")
def cfg4_oracle_one_iteration(cfg4):
    p = cfg4
    return O.ba_solve(p["poses0"], p["points0"], p["op"], p["oj"], p["uv"], BP.F, BP.CX, BP.CY, max_iterations=1,
                      num_threads=min(os.cpu_count() or 1, 16))


def _solve(ctx, p, mode, iters, parts=None, allreduce=None, bulk_control=None):
    import stereo_vo_amd as S
    pts, op, oj, uv = parts if parts else (p["points0"], p["op"], p["oj"], p["uv"])
    ba = S.api.BA(ctx, 20, BP.F, BP.CX, BP.CY, max_landmarks=len(pts) + 8, max_observations=len(op) + 8, max_iterations=iters,
                  max_time_s=0.0, accumulation=mode, bulk_control=bulk_control)
    if allreduce is not None:
        ba.set_allreduce(allreduce)
    ba.load_problem(p["poses0"], pts, op, oj, uv)
    s = ba.solve_problem()
    poses, points = ba.read_problem()
    st = ba.last_stats()
    ba.close()
    return s, poses, points, st


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mfma", "atomics"])
def test_hip_config4_first_iteration_matches_oracle(ctx, cfg4, cfg4_oracle_one_iteration, mode):
    """One LM iteration from identical inputs: hardware-order sums vs the oracle's declared order, 1e-9."""
    po, pto, so = cfg4_oracle_one_iteration
    s, poses, pts, _ = _solve(ctx, cfg4, mode, 1)
    assert s.iterations == so["iterations"] == 1 and s.successful_steps == so["successful"]
    assert abs(s.initial_cost - so["initial_cost"]) <= 1e-11 * so["initial_cost"]
    assert abs(s.final_cost - so["final_cost"]) <= 1e-9 * so["final_cost"]
    assert np.allclose(poses, po, rtol=0, atol=1e-9)
    assert np.allclose(pts, pto, rtol=1e-8, atol=1e-8)
    assert np.array_equal(poses[0], cfg4["poses0"][0])  # pose 0 constant (src/bundle_adjuster.cpp:130)


@pytest.mark.gpu
def test_hip_config4_cost_is_monotone_and_one_round_trip_per_iteration(ctx, cfg4):
    costs = []
    for it in (1, 2, 4, 7, 10):
        s, _, _, st = _solve(ctx, cfg4, "mfma", it)
        costs.append(s.final_cost)
        assert s.iterations == it
        # every step also carried the next linearisation (chained decision or same-sweep): no stand-alone pass A after
        # the first unless a prediction missed
        assert st.step_calls == it and st.speculations >= it - 1
        assert st.linearize_calls <= 2 + (st.speculations - st.speculation_hits)  # first pass A, the one after the last step
    # accepted steps only ever lower the cost (1e-9: separate runs of the hardware-order sums differ in the last digits)
    assert all(b <= a * (1 + 1e-9) for a, b in zip(costs, costs[1:])), costs
    assert costs[-1] < 0.01 * s.initial_cost


@pytest.mark.gpu
def test_hip_config4_device_step_control_equals_the_host_driven_loop(ctx, cfg4):
    """Round 5: the bulk path's step control on the device (ba_bulk_control_kernel: Cholesky n = 114 in LDS, Ceres' decision,
    pose update; the host only enqueues) against host/lm.cpp driving the same kernels — the same arithmetic on sums that differ
    in their last bits (hardware-order accumulation), so: same iteration / step counts, costs to 1e-9, poses to 1e-8."""
    iters = 8
    sh, ph, xh, sth = _solve(ctx, cfg4, "mfma", iters, bulk_control=False)
    sd, pd, xd, std = _solve(ctx, cfg4, "mfma", iters, bulk_control=True)
    assert sth.device_control == 0 and std.device_control == 1
    assert sd.iterations == sh.iterations == iters and sd.successful_steps == sh.successful_steps
    assert abs(sd.initial_cost - sh.initial_cost) <= 1e-11 * sh.initial_cost
    assert abs(sd.final_cost - sh.final_cost) <= 1e-9 * sh.final_cost
    assert np.allclose(pd, ph, rtol=0, atol=1e-8)
    assert np.allclose(xd, xh, rtol=1e-8, atol=1e-7)
    assert np.array_equal(pd[0], cfg4["poses0"][0])
    assert std.step_calls == iters and std.linearize_calls == 1 and std.speculation_hits >= iters - 1
    assert 0 < std.host_us / iters < 200  # enqueueing five operations per iteration; the GPU never waits for it (run-ahead)


@pytest.mark.gpu
@pytest.mark.parametrize("world,control", [(2, "host"), (4, "host"), (8, "host"), (2, "device"), (4, "device")])
def test_hip_config4_landmark_shards_equal_the_unsharded_solve(ctx, cfg4, world, control):
    """`world` landmark shards (j mod world) solved in lock-step inside one process, the callback summing their device
    payloads (what the RCCL all-reduce does across ranks): every shard must take the same decisions and end at the
    unsharded poses; its own landmarks must match the unsharded ones."""
    import torch
    import stereo_vo_amd as S
    from stereo_vo_amd import sharding
    iters = 3
    s_ref, poses_ref, pts_ref, _ = _solve(ctx, cfg4, "mfma", iters)
    bar = threading.Barrier(world)
    bufs, res, errs = [None] * world, [None] * world, []
    ctxs = [S.Context(64, 64) for _ in range(world)]
    calls = [0] * world

    def run(rank):
        try:
            torch.cuda.set_device(0)

            def cb(ptr, n):
                calls[rank] += 1
                t = torch.as_tensor(sharding._DevBuf(ptr, n), device="cuda:0")
                bufs[rank] = t
                bar.wait()
                tot = torch.stack([b for b in bufs]).sum(0)  # same order on every "rank": identical sums
                torch.cuda.synchronize()
                bar.wait()
                t.copy_(tot)
                torch.cuda.synchronize()
                bar.wait()
                return 0
            parts = sharding.shard_problem(cfg4["points0"], cfg4["op"], cfg4["oj"], cfg4["uv"], rank, world)
            s, poses, pts, st = _solve(ctxs[rank], cfg4, "mfma", iters, parts=parts[:4], allreduce=cb, bulk_control=control == "device")
            res[rank] = (parts[4], s, poses, pts, st)
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for rank in range(world):
        mine, s, poses, pts, st = res[rank]
        assert s.iterations == s_ref.iterations == iters and s.successful_steps == s_ref.successful_steps
        assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * s_ref.final_cost
        assert np.allclose(poses, poses_ref, rtol=0, atol=1e-8)
        assert np.allclose(pts, pts_ref[mine], rtol=1e-8, atol=1e-7)
        assert np.array_equal(poses, res[0][2])  # bit-identical poses on every "rank"
        # collectives per rank: one per stand-alone pass A; per LM iteration ONE when both payloads share it (same sweep) or
        # when no next linearisation is asked for (the last iteration), two when the decision is chained (payload2, payload1)
        # With the step control on the device (opt-in) every sequence slot is chained and the host stays SVO_BA_RUNAHEAD (2) slots
        # ahead of the status records: one collective for the first linearisation, two per enqueued slot — the same count on every
        # rank, because a slot is enqueued as a function of the records alone.
        assert st.device_control == int(control == "device") and (st.host_us > 0) == (control == "device")
        assert calls[rank] == st.collectives == calls[0]
        assert st.collectives <= 1 + 2 * (iters + 2)
    [c.close() for c in ctxs]
