"""N>1 path on CPU: world_size-2 gloo run (landmarks shard across ranks, one all-reduce of the
reduced camera system per LM iteration).  The sharded solve must reproduce the single-rank solve."""
import os
import subprocess
import sys

import numpy as np

import ba_problem as BP
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_problem_partitions_every_landmark_once():
    from stereo_vo_amd import sharding
    p = BP.make_problem(3, 5, 200)
    seen = np.zeros(len(p["points0"]), int)
    nobs = 0
    for r in range(3):
        pts, op, oj, uv, mine = sharding.shard_problem(p["points0"], p["op"], p["oj"], p["uv"], r, 3)
        seen[mine] += 1
        nobs += len(op)
        assert np.all(np.diff(oj) >= 0) and (len(oj) == 0 or oj.max() < len(pts))
        assert np.array_equal(pts, p["points0"][mine])
    assert np.all(seen == 1) and nobs == len(p["op"])


def test_two_rank_gloo_matches_single_rank(tmp_path):
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_multirank_worker.py"), out]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    r = np.load(out)
    assert bool(r["same"]), "ranks disagree on poses / iteration count"
    p = BP.make_problem(11, 6, 500)
    poses, pts, s = O.ba_solve(p["poses0"], p["points0"], p["op"], p["oj"], p["uv"], BP.F, BP.CX, BP.CY)
    dt, ang = BP.pose_error(r["poses"], poses)
    assert dt < 1e-6 and ang < 1e-6 and int(r["iterations"]) == s["iterations"]
    assert np.allclose(r["points"], pts, rtol=1e-7, atol=1e-6)
    assert abs(float(r["final_cost"]) - s["final_cost"]) <= 1e-9 * s["final_cost"]


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["deterministic", "mfma"])
def test_hip_two_rank_rccl_matches_single_rank(tmp_path, ctx, mode):
    """Two processes, one GPU each, the HIP library per rank, ncclAllReduce issued by the library on the adjuster's
    stream (svo_ba_set_comm).  Needs two visible GPUs: skipped on a one-GPU box (RCCL refuses two ranks on one
    device); tools/rehearse_n2.sh rehearses the same code path there through the callback collective."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one per rank)")
    import stereo_vo_amd as S
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "tests", "_rccl_worker.py"), out, mode]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    r = np.load(out)
    assert bool(r["same"]), "ranks disagree on poses / iteration count"
    p = BP.make_problem(31, 8, 4000)
    ref = S.api.BA(ctx, 8, BP.F, BP.CX, BP.CY, max_landmarks=len(p["points0"]) + 8, max_observations=len(p["op"]) + 8,
                   max_time_s=0.0, accumulation=mode)
    ref.load_problem(p["poses0"], p["points0"], p["op"], p["oj"], p["uv"])
    s = ref.solve_problem()
    poses, pts = ref.read_problem()
    ref.close()
    dt, ang = BP.pose_error(r["poses"], poses)
    assert dt < 1e-5 and ang < 1e-5
    assert abs(float(r["final_cost"]) - s.final_cost) <= 1e-6 * s.final_cost
    assert int(r["usable"]) >= int(r["steps"]) - 3  # one host round trip per LM iteration on every rank


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["deterministic", "mfma"])
def test_hip_native_rccl_call_path_on_one_rank(ctx, mode):
    """A world-size-1 communicator exercises everything of the native collective that one GPU can: librccl bound at run
    time (the copy the process already holds), ncclCommInitRank, ncclAllReduce enqueued by the library on the
    adjuster's stream between its kernels (chained decision included), D2H behind it.  The sum over one rank is the
    identity, so the solve must equal the one without a communicator — bit for bit in the deterministic mode."""
    import stereo_vo_amd as S
    from stereo_vo_amd import api
    p = BP.make_problem(41, 6, 1500)
    kw = dict(max_landmarks=len(p["points0"]) + 8, max_observations=len(p["op"]) + 8, max_time_s=0.0, accumulation=mode)

    def solve(comm):
        ba = S.api.BA(ctx, 6, BP.F, BP.CX, BP.CY, **kw)
        if comm:
            ba.set_comm(comm)
        ba.load_problem(p["poses0"], p["points0"], p["op"], p["oj"], p["uv"])
        s = ba.solve_problem()
        out = (s.iterations, s.final_cost) + ba.read_problem() + (ba.last_stats(),)
        ba.close()
        return out
    comm = api.rccl_comm_create(1, 0, api.rccl_unique_id(), 0)
    try:
        a, b = solve(None), solve(comm)
    finally:
        api.rccl_comm_destroy(comm)
    assert a[0] == b[0] and b[4].speculation_hits >= b[0] - 3
    if mode == "deterministic":
        assert a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    else:
        assert abs(a[1] - b[1]) <= 1e-6 * a[1] and np.allclose(a[2], b[2], atol=1e-5)
