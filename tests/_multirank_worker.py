"""Worker for tests/test_multirank.py: one rank of a world_size-2 gloo run on CPU.
Checks the N>1 path's host logic under a REAL collective: landmark sharding + the PRODUCT's LM step control
(svo_lm_solve in libsvo_hip.so — the code every rank of a sharded GPU run executes, chained / same-sweep
linearisation included) + gloo all-reduce.  The two passes over the observations come from the CPU oracle
(the HIP kernels cannot run without a GPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ba_problem as BP  # noqa: E402
import oracle_lib as O  # noqa: E402
from stereo_vo_amd import sharding  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    p = BP.make_problem(11, 6, 500)
    pts, op, oj, uv, mine = sharding.shard_problem(p["points0"], p["op"], p["oj"], p["uv"], rank, world)
    def allreduce(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
    poses, lpts, summ, stats, exchanges = O.product_lm_over_oracle_passes(p["poses0"], pts, op, oj, uv, BP.F, BP.CX, BP.CY, allreduce=allreduce)
    s = dict(iterations=summ.iterations, final_cost=summ.final_cost)
    assert stats.speculation_hits >= summ.iterations - 3, "one host round trip per LM iteration"
    # every rank must hold identical poses and have taken the same number of iterations
    t = torch.from_numpy(np.concatenate([poses.ravel(), [s["iterations"], s["final_cost"], exchanges]]))
    g = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(g, t)
    same = all(torch.equal(g[0], x) for x in g)
    full = np.zeros_like(p["points0"])
    full[mine] = lpts
    tf = torch.from_numpy(full)
    dist.all_reduce(tf)
    if rank == 0:
        np.savez(out, poses=poses, points=tf.numpy(), iterations=s["iterations"], final_cost=s["final_cost"], same=same)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
