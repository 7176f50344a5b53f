"""Worker for tests/test_multirank.py::test_hip_two_rank_rccl_*: one rank of a 2-process run, ONE GPU PER RANK, the HIP
library per rank, the reduced camera system summed by RCCL called from the library (svo_ba_set_comm).  Rank 0 writes
the result; the test compares it with the single-rank solve."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ba_problem as BP  # noqa: E402


def main():
    out, mode = sys.argv[1], sys.argv[2]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    local = int(os.environ.get("SVO_TEST_FORCE_DEVICE", local))  # probe: two ranks on one device (RCCL is expected to refuse)
    torch.cuda.set_device(local)
    dist.init_process_group("gloo")  # rendezvous + id broadcast only: the data path is the library's own RCCL communicator
    import stereo_vo_amd as S
    from stereo_vo_amd import api, sharding
    p = BP.make_problem(31, 8, 4000)
    pts, op, oj, uv, mine = sharding.shard_problem(p["points0"], p["op"], p["oj"], p["uv"], rank, world)
    box = [api.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, 0)
    comm = api.rccl_comm_create(world, rank, box[0], local)
    ctx = S.Context(64, 64, device=local)
    ba = S.api.BA(ctx, 8, BP.F, BP.CX, BP.CY, max_landmarks=len(pts) + 8, max_observations=len(op) + 8, max_time_s=0.0, accumulation=mode)
    ba.set_comm(comm)
    ba.load_problem(p["poses0"], pts, op, oj, uv)
    s = ba.solve_problem()
    poses, lpts = ba.read_problem()
    st = ba.last_stats()
    t = torch.from_numpy(np.concatenate([poses.ravel(), [s.iterations, s.final_cost]]))
    g = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(g, t)
    same = all(torch.equal(g[0], x) for x in g)
    full = np.zeros_like(p["points0"])
    full[mine] = lpts
    tf = torch.from_numpy(full)
    dist.all_reduce(tf)
    if rank == 0:
        np.savez(out, poses=poses, points=tf.numpy(), iterations=s.iterations, final_cost=s.final_cost, same=same,
                 steps=st.step_calls, usable=st.speculation_hits)
    ba.close()
    api.rccl_comm_destroy(comm)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
