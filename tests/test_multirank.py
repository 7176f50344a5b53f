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
