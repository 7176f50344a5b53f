#!/usr/bin/env python3
"""bench.py — stereo frames/s of the MI355X-native stereo-VO hot path (BASELINE.json metric).

A "step" = one pass of the whole hot path, on every one of `--streams` (default 8) independent stereo streams
that share the GPU (own HIP stream, pipeline and BA worker each; exactly how ranks are used across GPUs), i.e. one pass (ImageProcessor::process + BundleAdjuster::bundle_adjust per
frame: corner detection, pyramids, forward/backward LK + survivor filter, PnP-RANSAC, dedup, stereo
disparity at the features, triangulation, sliding-window bundle adjustment) over one batch of B
consecutive synthetic KITTI-shaped stereo pairs that are already resident in HBM, starting from a reset
pipeline (so every step does identical work).  Workload = BASELINE.json configs[1]:
1241x376, ~1.5k corners per frame (max_corners 1500, quality 0.02, minDistance 10), 5-keyframe BA window.

N GPUs: the frame stream shards across ranks (rank r processes its own sequence chunk; weak scaling, no
data-path collective — SURVEY §8e "front end / frames").  value = frames all ranks processed / max-over-
ranks time.  The BA all-reduce path (config 4) is benchmarked with --workload ba50k.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP
events on the library's stream) and `cpu_baseline` (oracle pipeline on the host cores, rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# HIP runtime knob, read when the runtime initialises: the default of 4 hardware queues makes the 2 HIP streams of
# each stereo stream (tracker + bundle adjuster) share queues and serialise; measured +6 % at 8 streams.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H = 1241, 376
MAXC, QUALITY, MIN_DIST, MAX_FEAT, WINDOW = 1500, 0.02, 10.0, 2000, 5
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 measured copy)
FP64_VEC_PEAK_TFLOPS = 78.6  # public MI355X FP64 vector spec (not in the local guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="stereo pairs per step (per stream)")
    ap.add_argument("--streams", type=int, default=8, help="independent stereo streams processed concurrently per GPU")
    ap.add_argument("--workload", default="kitti_cfg1", choices=["kitti_cfg1", "ba50k"])
    ap.add_argument("--profile-kernel", default="lk_fb", help="kernel timed with HIP events for the roofline object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16)
    return ap.parse_args()


def dist_setup(n):
    """One process per GPU; RCCL via torch.distributed when n > 1."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "SVO_BENCH_FORCE_DEVICE" in os.environ:  # rehearsal of the N>1 code path on a one-GPU box (gloo backend)
        local = int(os.environ["SVO_BENCH_FORCE_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        backend = os.environ.get("SVO_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        return torch, dist, rank, local, world
    return torch, None, rank, local, world


def barrier_sync(torch, dist, ctx):
    ctx.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def render_batch(S, seed, batch):
    p = S.synth_default(W, H)
    p.seed = seed
    fr = [S.synth_render(p, i) for i in range(batch)]
    return p, np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])


def cpu_baseline(p, L, R, frames):
    """Oracle pipeline (CPU restatement of the reference path) on the same frames, host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    pipe = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=W, height=H, max_corners=MAXC,
                      quality=QUALITY, min_feature_distance=MIN_DIST, parallax_thresh=20.0, window_size=WINDOW,
                      max_features=MAX_FEAT, ba_max_iterations=50, num_threads=cores)
    n = min(frames, L.shape[0])
    t0 = time.perf_counter()
    res = [pipe.process(L[i], R[i]) for i in range(n)]
    reps = 1
    while time.perf_counter() - t0 < 8.0 and reps < 12:  # bounded sample of ~10 s: the same frames again, fresh pipeline
        again = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=W, height=H, max_corners=MAXC,
                           quality=QUALITY, min_feature_distance=MIN_DIST, parallax_thresh=20.0, window_size=WINDOW,
                           max_features=MAX_FEAT, ba_max_iterations=50, num_threads=cores)
        for i in range(n):
            again.process(L[i], R[i])
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    # the reference runs Ceres with 4 threads (src/bundle_adjuster.cpp:12): the same sample once more at 4 threads
    os.environ["OMP_NUM_THREADS"] = "4"
    pipe4 = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=W, height=H, max_corners=MAXC,
                       quality=QUALITY, min_feature_distance=MIN_DIST, parallax_thresh=20.0, window_size=WINDOW,
                       max_features=MAX_FEAT, ba_max_iterations=50, num_threads=4)
    t1 = time.perf_counter()
    for i in range(n):
        pipe4.process(L[i], R[i])
    dt4 = time.perf_counter() - t1
    return dict(value=n / dt, unit="frames/s", cores=cores, kind="port",
                sample=f"first {n} frames of the same batch, whole oracle pipeline, {reps} passes of {dt:.1f} s",
                value_4_threads=n / dt4), res


class _Stream:
    """One independent stereo stream: its own svo_ctx (HIP stream), pipeline and BA worker, its own frames."""

    def __init__(self, S, torch, local, seed, B):
        self.ctx = S.Context(W, H, device=local, max_batch=B, max_corners=MAXC, max_candidates=1 << 16, max_features=MAX_FEAT)
        self.p, self.L, self.R = render_batch(S, seed, B)
        dev = torch.device("cuda", local)
        self.dL = torch.from_numpy(self.L).to(dev)
        self.dR = torch.from_numpy(self.R).to(dev)
        pp = S.pipeline_default_params()
        pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = self.p.focal, self.p.cx, self.p.cy, self.p.baseline
        pp.width, pp.height = W, H
        pp.max_corners, pp.quality, pp.min_feature_distance = MAXC, QUALITY, MIN_DIST
        pp.max_features, pp.window_size = MAX_FEAT, WINDOW
        pp.ba_max_iterations, pp.ba_max_time_s = int(os.environ.get("SVO_BENCH_BA_ITERS", "50")), 0.0  # env: sensitivity experiments only
        self.pipe = S.Pipeline(self.ctx, pp)
        self.B = B
        self.res = None

    def step(self):
        self.pipe.reset()
        self.res = self.pipe.process_batch_dev(self.dL.data_ptr(), self.dR.data_ptr(), self.B)

    def close(self):
        self.pipe.close()
        self.ctx.close()


def run_kitti(args):
    import threading
    import stereo_vo_amd as S
    torch, dist, rank, local, world = dist_setup(args.gpus)
    B, NS = args.batch, max(1, args.streams)
    if NS >= 4:  # include/svo.h: adjusters on one shader engine per XCD, trackers on the other three
        os.environ.setdefault("SVO_BA_CU_SHARE", "8")
    streams = [_Stream(S, torch, local, 0x5EED0001 + rank * 64 + i, B) for i in range(NS)]
    torch.cuda.synchronize()
    ctx = streams[0].ctx
    dev = torch.device("cuda", local)

    def run_steps(k):
        """k steps on every stream; streams run concurrently (ctypes releases the GIL inside the library)."""
        if NS == 1:
            for _ in range(k):
                streams[0].step()
            return
        def work(st):
            for _ in range(k):
                st.step()
        th = [threading.Thread(target=work, args=(st,)) for st in streams]
        [t.start() for t in th]
        [t.join() for t in th]

    run_steps(args.warmup)
    # single-stream rate (latency-bound: one sequential VO chain) measured first, in the same run
    single = None
    if NS > 1:
        barrier_sync(torch, dist, ctx)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            streams[0].step()
        streams[0].ctx.sync()
        dts = time.perf_counter() - t0
        single = {"value": B * args.steps / dts, "unit": "frames/s", "ms_per_step": 1e3 * dts / args.steps,
                  "note": "one stream alone on the GPU (per rank)"}
    ctx.profile_select(args.profile_kernel)
    barrier_sync(torch, dist, ctx)
    t0 = time.perf_counter()
    run_steps(args.steps)
    for st in streams:
        st.ctx.sync()
    barrier_sync(torch, dist, ctx)
    dt = time.perf_counter() - t0
    k_ms, k_n = ctx.profile_read()
    ctx.profile_select(None)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = streams[0].res
    frames = world * NS * B * args.steps
    n_kf = sum(r.is_keyframe for r in res)
    n_trk = [r.n_tracked for r in res if r.n_tracked]
    ba_it = sum(r.ba_iterations for r in res)
    out = {
        "metric": "stereo frames/sec on 1241x376 KITTI pairs", "value": frames / dt, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64",
        "data": "synthetic",
        "config": {"workload": "kitti_1241x376_1500corners_5kf_window (BASELINE configs[1])", "batch_per_stream": B,
                   "streams_per_gpu": NS, "max_corners": MAXC, "quality": QUALITY, "min_distance": MIN_DIST, "window": WINDOW,
                   "keyframes_per_step": n_kf, "mean_tracked": float(np.mean(n_trk)) if n_trk else 0.0,
                   "ba_lm_iterations_per_step": ba_it,
                   "sharding": "independent stereo streams: streams_per_gpu per rank, ranks hold different streams; no collective"},
    }
    if single is not None:
        out["single_stream"] = single
    # roofline of the profiled kernel (HIP events on the library stream of stream 0, over the timed region)
    if k_n > 0:
        avg_us = 1e3 * k_ms / k_n
        out["roofline"] = roofline_for(args.profile_kernel, avg_us, res, k_n, args.steps)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, ores = cpu_baseline(streams[0].p, streams[0].L, streams[0].R, args.cpu_frames)
        out["cpu_baseline"] = cb
        m = min(len(ores), len(res))
        same = all((a.n_detected, a.n_tracked, a.n_inliers, a.n_new, a.is_keyframe, a.ba_iterations) ==
                   (b.n_detected, b.n_tracked, b.n_inliers, b.n_new, b.is_keyframe, b.ba_iterations) and
                   list(a.pose7) == list(b.pose7) for a, b in zip(res[:m], ores[:m]))
        # ATE of the GPU trajectory against the CPU path's trajectory on identical inputs (BASELINE metric; m)
        def centres(rs):
            c = []
            for r in rs:
                q, t = np.array(r.pose7[:4]), np.array(r.pose7[4:])
                w, x, y, z = q
                Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                               [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                               [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
                c.append(-Rm.T @ t)  # camera in world (src/vo_node.cpp:149-150)
            return np.array(c)
        ate = float(S.api.ate_rmse(centres(res[:m]), centres(ores[:m]), False)) if m >= 3 else 0.0
        out["parity_vs_cpu"] = {"frames": m, "index_sets_and_poses_identical": bool(same), "ate_rmse_m": ate}
    for st in streams:
        st.close()
    if dist is not None:
        dist.destroy_process_group()
    return out if rank == 0 else None


def pmc_traffic_bytes(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed PMC passes (profiles/r01_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs, KB units, no 2x correction applied because
    the kernel's loads are byte-granular gathers, which the guide calls uncalibrated).  None if absent."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))[kernel_name]
        return (t["fetch_kb_per_launch"] + t["write_kb_per_launch"]) * 1024.0
    except Exception:
        return None


def roofline_for(kernel, avg_us, res, launches, steps):
    """Algorithmic work per launch / measured average launch duration."""
    A = W * H
    if kernel == "lk_fb":
        # one launch per tracked frame: n features x (fwd + bwd) x 4 levels; bytes actually needed from
        # HBM/L2: both pyramids once (2 x 1.33 A); the binding resource is integer VALU/LDS, reported as
        # achieved GB/s against HBM for the contract plus the op rate in `note`.
        n = np.mean([r.n_tracked for r in res if r.n_tracked]) if any(r.n_tracked for r in res) else 0
        byts = 2 * 1.33 * A
        gbs = byts / (avg_us * 1e-6) / 1e9
        return {"kernel": "lk_fb_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes("lk_fb_kernel"), "avg_launch_us": avg_us,
                "launches": launches, "algorithmic_bytes_per_launch": byts,
                "note": f"latency/integer-VALU bound, not HBM bound (DESIGN.md section 4); {n:.0f} features/launch; per feature 2 directions x 4 levels x <=30 iterations x 441-px window"}
    if kernel == "corner_response":
        B = launches and (len(res))
        byts = 5.0 * A * B  # read u8, write f32 response per frame, B frames per launch
        gbs = byts / (avg_us * 1e-6) / 1e9
        return {"kernel": "corner_response_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": avg_us, "launches": launches}
    if kernel in ("ba_linearize", "ba_backsub"):
        return {"kernel": kernel + "_kernel", "bound": "mfma", "achieved": None, "peak": FP64_VEC_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": None, "traffic": None, "avg_launch_us": avg_us, "launches": launches}
    return {"kernel": kernel, "avg_launch_us": avg_us, "launches": launches}


def main():
    args = parse()
    if args.workload == "kitti_cfg1":
        out = run_kitti(args)
    else:
        from tools import bench_ba
        out = bench_ba.run(args)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
