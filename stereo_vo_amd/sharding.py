"""Landmark sharding of a bundle-adjustment problem across ranks.

Poses are replicated; landmark j and all its observations go to rank j % world.  The per-landmark Schur
elimination is local, so the only exchange is one all-reduce of the packed reduced camera system
[S | g_red | g_c | diag U | cost | sum g_p^2] (+ one of 4 scalars) per LM iteration, which every rank
feeds through the same host-side step control => identical decisions on all ranks.
"""
import ctypes as C

import numpy as np


def shard_problem(points3, obs_pose, obs_point, obs_uv, rank, world):
    """Return (local_points, local_obs_pose, local_obs_point, local_obs_uv, global_index_of_local_points).
    Observations stay landmark-major."""
    points3 = np.asarray(points3)
    obs_point = np.asarray(obs_point)
    mine = np.nonzero(np.arange(points3.shape[0]) % world == rank)[0]
    remap = -np.ones(points3.shape[0], np.int64)
    remap[mine] = np.arange(len(mine))
    m = remap[obs_point] >= 0
    return (points3[mine], np.asarray(obs_pose)[m].astype(np.int32), remap[obs_point[m]].astype(np.int32),
            np.asarray(obs_uv)[m], mine)


class _DevBuf:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can wrap it zero-copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def allreduce_device_fn(dist, device):
    """svo_allreduce_fn for the HIP library: RCCL all-reduce (sum) of `n` doubles at device pointer `ptr`."""
    import torch

    views = {}  # (ptr, n) -> zero-copy tensor view of the library's payload buffer (built once: two calls per LM iteration)

    def fn(ptr, n):
        t = views.get((ptr, n))
        if t is None:
            t = views[(ptr, n)] = torch.as_tensor(_DevBuf(ptr, n), device=device)
        dist.all_reduce(t)
        torch.cuda.synchronize(device)
        return 0
    return fn


def allreduce_host_fn(dist):
    """Callback for the CPU oracle (tests): all-reduce (sum) of a host double buffer (gloo)."""
    import torch

    def fn(buf, n, user):
        a = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(a)
        dist.all_reduce(t)
        return 0
    return fn
