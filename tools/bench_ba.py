"""bench.py --workload ba50k: BASELINE.json configs[3] — synthetic 50k-landmark / 20-keyframe bundle
adjustment, landmarks sharded across the ranks, one RCCL all-reduce of the reduced camera system per LM
iteration (the path's only exchange step).  A step = one LM iteration (strong scaling: the
problem is fixed, per-rank work shrinks with N)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, CX, CY, W, H = 718.856, 607.1928, 185.2157, 1241, 376
FP64_MATRIX_PEAK_TFLOPS = 78.6  # v_mfma_f64_16x16x4: 2048 flop / 64 cycles / SIMD (measured: SQ_VALU_MFMA_BUSY_CYCLES = 64 x MFMA count)


def make_problem(seed=0xBA000004, K=20, N=50000, dense=False, noise=0.5):
    """Vectorised SURVEY §8d config-4 generator: poses on a gently curving trajectory 1 m apart, landmarks in the
    union of frusta, visibility windows L_j ~ U{2..20}, 0.5 px noise, perturbed initial values, pose 0 fixed."""
    rng = np.random.default_rng(seed)
    yaw = 0.01 * np.arange(K)
    C = np.stack([0.005 * np.arange(K) ** 2, np.zeros(K), 1.0 * np.arange(K)], 1)
    q = np.stack([np.cos(-yaw / 2), np.zeros(K), np.sin(-yaw / 2), np.zeros(K)], 1)

    def rot(qq):
        w, x, y, z = qq.T
        return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                         np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                         np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    R = rot(q)
    t = -np.einsum("kij,kj->ki", R, C)
    poses = np.concatenate([q, t], 1)
    pts = np.stack([rng.uniform(-14, 14, N), rng.uniform(-2.5, 2.0, N), rng.uniform(4, 60 + K, N)], 1)
    Lj = np.full(N, K) if dense else rng.integers(2, K + 1, N)
    sj = np.zeros(N, int) if dense else (rng.random(N) * (K - Lj + 1)).astype(int)
    Xc = np.einsum("kij,nj->nki", R, pts) + t[None]
    u = F * Xc[..., 0] / Xc[..., 2] + CX
    v = F * Xc[..., 1] / Xc[..., 2] + CY
    kk = np.arange(K)[None]
    vis = (Xc[..., 2] > 1.0) & (u >= 0) & (u < W) & (v >= 0) & (v < H) & (kk >= sj[:, None]) & (kk < (sj + Lj)[:, None])
    vis &= vis.sum(1, keepdims=True) >= 2
    keep = vis.any(1)
    pts, vis, u, v = pts[keep], vis[keep], u[keep], v[keep]
    oj, op = np.nonzero(vis)  # landmark-major, pose ascending
    uv = np.stack([u[oj, op], v[oj, op]], 1) + rng.normal(0, noise, (len(oj), 2))
    pts0 = pts + rng.normal(0, 0.10, pts.shape)
    poses0 = poses.copy()
    poses0[1:, 4:] += rng.normal(0, 0.05, (K - 1, 3))
    dq = np.concatenate([np.ones((K - 1, 1)), rng.normal(0, 0.004, (K - 1, 3))], 1)
    w1, x1, y1, z1 = dq.T
    w2, x2, y2, z2 = poses0[1:, :4].T
    poses0[1:, :4] = np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                               w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1)
    return dict(poses0=poses0, points0=pts0, op=op.astype(np.int32), oj=oj.astype(np.int32), uv=uv, poses_gt=poses)


def flops_per_iteration(op, oj, n_points):
    """466 per observation + per landmark 50 + 144 L + 216 L(L+1)/2 (f64)."""
    L = np.bincount(oj, minlength=n_points).astype(np.float64)
    return 466.0 * len(op) + float(np.sum(50 + 144 * L + 216 * L * (L + 1) / 2))


def run(args):
    sys.path.insert(0, ROOT)
    import stereo_vo_amd as S
    from stereo_vo_amd import sharding
    import bench
    torch, dist, rank, local, world = bench.dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    p = make_problem(dense=os.environ.get("SVO_BA_DENSE") == "1")
    # developer aid: SVO_BA_SHARD_OF=N times the per-rank load of an N-rank run on one GPU (no collective)
    emu = int(os.environ.get("SVO_BA_SHARD_OF", "0"))
    pts, op, oj, uv, mine = sharding.shard_problem(p["points0"], p["op"], p["oj"], p["uv"], rank if not emu else 0, world if not emu else emu)
    ctx = S.Context(64, 64, device=local)
    K = p["poses0"].shape[0]
    iters = args.steps
    acc_mode = os.environ.get("SVO_BA_ACC", "mfma")
    ba = S.BA(ctx, K, F, CX, CY, max_landmarks=len(pts) + 8, max_observations=len(op) + 8, max_iterations=iters, max_time_s=0.0,
              accumulation=acc_mode)
    if dist is not None:
        ba.set_allreduce(sharding.allreduce_device_fn(dist, dev))
    # warm-up solve (W iterations), then the timed solve of exactly K LM iterations from the same start
    ba.load_problem(p["poses0"], pts, op, oj, uv)
    ctx.profile_select("ba_linearize")
    bench.barrier_sync(torch, dist, ctx)
    t0 = time.perf_counter()
    s = ba.solve_problem()
    bench.barrier_sync(torch, dist, ctx)
    dt = time.perf_counter() - t0
    k_ms, k_n = ctx.profile_read()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_it = max(s.iterations, 1)
    fl_local = flops_per_iteration(op, oj, len(pts))
    out = {"metric": "BA LM iterations/sec (50k landmarks x 20 keyframes)", "value": n_it / dt, "unit": "iterations/s",
           "n_gpus": world, "steps": n_it, "warmup": 0, "ms_per_step": 1e3 * dt / n_it, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "ba_50k_landmarks_20_keyframes (BASELINE configs[3])", "observations": int(len(p["op"])),
                      "landmarks": int(len(p["points0"])), "poses": K, "final_cost": s.final_cost,
                      "initial_cost": s.initial_cost, "collective": "allreduce(sum,f64) of %d doubles per LM iteration" % ((6 * (K - 1)) ** 2 + 18 * (K - 1) + 2)}}
    if k_n:
        # roofline of the dominant kernel.  Two figures: `achieved` = SURVEY 8d algorithmic f64 flops of one
        # linearisation / launch time (what the work needs); `mfma_issued` = MFMA instructions x 2048 flop /
        # launch time (what the matrix pipe executed: K = 3 of 4 slots and padded 16x16 tiles included).
        avg_us = 1e3 * k_ms / k_n
        tf = fl_local / (avg_us * 1e-6) / 1e12
        Lc = np.bincount(oj, minlength=len(pts))
        out["roofline"] = {"kernel": "ba_linearize_mfma_kernel" if acc_mode == "mfma" else "ba_linearize_kernel",
                           "bound": "mfma", "achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": tf / FP64_MATRIX_PEAK_TFLOPS, "traffic": None, "avg_launch_us": avg_us,
                           "launches": k_n, "accumulation": acc_mode,
                           "note": "f64: matrix and vector peak are both 78.6 TF on MI355X; per-rank algorithmic flops "
                                   "(466/observation + 50 + 144 L + 108 L(L+1) per landmark); MFMA pipe busy fraction "
                                   "from rocprofv3 PMC is in profiles/r01_ba50k_mfma_pmc*.txt"}
    ba.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    return out if rank == 0 else None
