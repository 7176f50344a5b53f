#!/usr/bin/env python3
"""bench.py — stereo frames/s of the MI355X-native stereo-VO hot path (BASELINE.json metric).

Default line = "96 concurrent streams as 3 pipeline groups, inputs resident in HBM": a compute rate.  The same 16 frames per
stream are re-processed every step from a reset pipeline (identical work per step, no upload in the timed region);
`--workload kitti_stream` is the streaming figure (BASELINE configs[2]: 4541 frames through the host-pointer entry,
upload included, 10-keyframe window, nothing reset).  The default line also carries `streaming` (the same lanes fed from host
memory through svo_pipeline_group_staging / _upload / _process_uploaded, upload timed), `single_stream`, `parity_self` (every
lane of every timed step against the first step) and `parity_vs_cpu` (4 lanes of every group against the CPU oracle), and `other_workloads`
(kitti_stream, ba50k sparse and dense, hd_1280x720_10k), each with its own roofline / cpu_baseline / parity record.

A "step" = one pass of the whole hot path on every one of `--streams` (default 96) independent stereo streams that share the
GPU — as `--groups` (default 3) pipeline groups (svo_pipeline_group_*: one host thread per group, one kernel launch per stage
for the lanes that are ready, their bundle adjustments one device-resident launch), or with `--groups 0` as one svo_pipeline
and one host thread per stream (round 2's form) — i.e. one pass (ImageProcessor::process + BundleAdjuster::bundle_adjust per
frame: corner detection, pyramids, forward/backward LK + survivor filter, PnP-RANSAC, dedup, stereo disparity at the features,
triangulation, sliding-window bundle adjustment) over one batch of B consecutive synthetic KITTI-shaped stereo pairs that are
already resident in HBM, starting from a reset pipeline (so every step does identical work).  Workload = BASELINE.json
configs[1]: 1241x376, ~1.5k corners per frame (max_corners 1500, quality 0.02, minDistance 10), 5-keyframe BA window.

N GPUs: the frame stream shards across ranks (rank r processes its own sequence chunk; weak scaling, no
data-path collective — SURVEY §8e "front end / frames").  value = frames all ranks processed / max-over-
ranks time.  The BA all-reduce path (config 4) is benchmarked with --workload ba50k.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the front end's contract figure; the dominant
kernel = ba_lm_kernel with live HIP-event duration and live algorithmic flops / bytes; the tracker on VALU issue) and `cpu_baseline` (oracle pipeline on the host cores, rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

# HIP runtime knob, read when the runtime initialises (profiles/r04_exp_hw_queues.txt: 4 / 8 / 12 / 16; r03_group_sweep.txt also 20 / 24):
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # HIP streams that share a hardware queue serialise (a 3 ms solve in front of a tracking launch), and more hardware queues than the GPU has slots for slow EVERY stream of the process down (round 5, profiles/r05_exp_lanes_groups.txt: 32 queues gave the headline 27-33 k frames/s but halved the single-stream workloads that follow in the same process).  So: 16 queues, and at most 16 streams — see group_lines() below

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H = 1241, 376
MAXC, QUALITY, MIN_DIST, MAX_FEAT, WINDOW = 1500, 0.02, 10.0, 2000, 5
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0        # ... and the copy bandwidth that guide measured
VALU_ISSUE_PEAK_GINSTR = 1228.8  # 256 CUs x 4 SIMDs x 2.4 GHz / 2 (one wave64 VALU instruction per 2 cycles per SIMD)
VALU_CYCLES_TRACKER_MIX = 3.6     # SIMD cycles per VALU instruction of the tracker's iteration (81 half-rate + 51 full-rate, measured rates)
VALU_CYCLES_OTHER_KERNELS = 3.4   # assumed for the other kernels (between the measured 2.5 and 4.3)
FP64_VEC_PEAK_TFLOPS = 78.6  # public MI355X FP64 vector spec (not in the local guide)
PROFILE_SUMMARY = os.path.join(ROOT, "profiles", "r05_summary.json")  # written by tools/prof_round.sh + prof_summary.py


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="stereo pairs per step (per stream)")
    ap.add_argument("--streams", type=int, default=96, help="independent stereo streams processed concurrently per GPU (rounds 3-4: 48 in 2 groups)")
    ap.add_argument("--groups", type=int, default=3, help="pipeline groups (= host driver threads) the streams are split over; 0: one host thread and one svo_pipeline per stream (round 2's shape)")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="group i starts its steps i x this many milliseconds after group 0 (inside the timed region)")
    ap.add_argument("--workload", default="kitti_cfg1", choices=["kitti_cfg1", "kitti_stream", "ba50k", "hd10k"])
    ap.add_argument("--frames", type=int, default=4541, help="kitti_stream: length of the stream (KITTI 00 has 4541 frames)")
    ap.add_argument("--profile-kernel", default="lk_fb", help="kernel timed with HIP events for the roofline object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-stream measurement (profiling runs)")
    ap.add_argument("--no-streaming", action="store_true", help="default line only: skip the host-fed (PCIe upload timed) variant of the same lanes")
    ap.add_argument("--no-other-workloads", action="store_true", help="default line only: skip the driver-timed figures of BASELINE configs[2..4]")
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


RES_KEY = lambda r: (r.n_detected, r.n_tracked, r.n_inliers, r.n_new, r.is_keyframe, r.ba_iterations, list(r.pose7))


def run_threads(fns):
    """One thread per callable; returns when all have ended and RAISES the first exception any of them died of.  (Until the middle of
    round 5 the bench joined its group threads without looking: a group that died on an error shortened the run instead of failing
    it, and the frames it never processed were counted — profiles/r05_exp_reset_fill.txt.)"""
    import threading
    errs = []

    def guarded(fn):
        try:
            fn()
        except BaseException as e:  # noqa: BLE001 — a thread that dies must fail the run, not shorten it
            errs.append(e)
    th = [threading.Thread(target=guarded, args=(fn,)) for fn in fns]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]


def group_self_parity(groups, B):
    """Every lane of every group, every timed step: the SAME frames from a reset group must give the SAME result records, bit for
    bit (a stale or late hand-over inside a solve would show here on whatever lane and step it hits)."""
    lanes = steps = bad = 0
    first_bad = None
    for gi, g in enumerate(groups):
        if not g.raws:
            continue
        rec = len(g.raws[0]) // (g.n * B)
        ref = g.raws[0]
        steps = max(steps, len(g.raws))
        lanes += g.n
        for si, raw in enumerate(g.raws[1:], 1):
            if raw == ref:
                continue
            for l in range(g.n):
                if raw[l * B * rec:(l + 1) * B * rec] != ref[l * B * rec:(l + 1) * B * rec]:
                    bad += 1
                    first_bad = first_bad or {"group": gi, "lane": l, "step": si}
    short = [gi for gi, g in enumerate(groups) if len(g.raws) != steps]
    if short:
        raise RuntimeError(f"groups {short} recorded fewer steps than the others: {[len(g.raws) for g in groups]}")
    return {"lanes_checked": lanes, "steps_checked": steps, "lane_steps_that_differ_from_step_0": bad, "first": first_bad,
            "what": "raw svo_frame_result records of every lane, every timed step against the first timed step"}


def group_parity_vs_cpu(groups, B, lane0_oracle, per_group=4):
    """The CPU oracle on `per_group` lanes of every group (lane 0 of group 0 comes from the cpu_baseline leg): counts, keyframe
    decisions, LM iteration counts and poses must be identical."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    jobs = [(gi, l) for gi, g in enumerate(groups) for l in range(min(per_group, g.n)) if (gi, l) != (0, 0)]

    def run(job):
        gi, l = job
        p, L, R = groups[gi].data[l]
        pipe = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=W, height=H, max_corners=MAXC, quality=QUALITY,
                          min_feature_distance=MIN_DIST, parallax_thresh=20.0, window_size=WINDOW, max_features=MAX_FEAT, ba_max_iterations=50, num_threads=2)
        return [RES_KEY(pipe.process(L[i], R[i])) for i in range(B)]
    with ThreadPoolExecutor(max_workers=8) as ex:
        refs = list(ex.map(run, jobs))
    bad = [job for job, ref in zip(jobs, refs) if [RES_KEY(r) for r in groups[job[0]].all_res[job[1]]] != ref]
    ok0 = [RES_KEY(r) for r in groups[0].all_res[0][:len(lane0_oracle)]] == [RES_KEY(r) for r in lane0_oracle]
    return {"lanes_checked": len(jobs) + 1, "lanes_that_differ": len(bad) + (0 if ok0 else 1), "lanes": [[0, 0]] + [list(j) for j in jobs],
            "frames_per_lane": B}


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


class _Group:
    """`lanes` independent stereo streams behind ONE host thread (svo_pipeline_group): one context, stream-batched launches."""

    def __init__(self, S, torch, local, seeds, B):
        n = len(seeds)
        self.ctx = S.Context(W, H, device=local, max_batch=n * B, max_corners=MAXC, max_candidates=1 << 16, max_features=MAX_FEAT)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 16)) as ex:  # (the renderer is host C++ behind ctypes: the GIL is released)
            data = list(ex.map(lambda sd: render_batch(S, sd, B), seeds))
        self.p, self.L, self.R = data[0]
        dev = torch.device("cuda", local)
        self.dL = torch.from_numpy(np.stack([d[1] for d in data])).to(dev)  # (lanes, B, H, W)
        self.dR = torch.from_numpy(np.stack([d[2] for d in data])).to(dev)
        pp = S.pipeline_default_params()
        pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = self.p.focal, self.p.cx, self.p.cy, self.p.baseline
        pp.width, pp.height = W, H
        pp.max_corners, pp.quality, pp.min_feature_distance = MAXC, QUALITY, MIN_DIST
        pp.max_features, pp.window_size = MAX_FEAT, WINDOW
        pp.ba_max_iterations, pp.ba_max_time_s = int(os.environ.get("SVO_BENCH_BA_ITERS", "50")), 0.0
        self.pipe = S.PipelineGroup(self.ctx, pp, n)
        self.B, self.n = B, n
        self.res = None
        self.all_res = None
        self.data = data             # (params, left, right) per lane: the parity legs render nothing again
        self.raws = []               # per step: every lane's raw result records (identical frames => identical bits, checked after the timed region)
        self.stats_sum = {}          # launches / lane-stages per stage, summed over the steps since clear_counters()

    def clear_counters(self):
        self.raws, self.stats_sum = [], {}

    def step(self):
        self.pipe.reset()
        self.all_res = self.pipe.process_batch_dev(self.dL.data_ptr(), self.dR.data_ptr(), self.B * W * H, self.B)
        self.res = self.all_res[0]
        self.raws.append(self.pipe.last_raw)
        for k, v in self.pipe.last_stats().items():
            a = self.stats_sum.setdefault(k, [0, 0])
            a[0] += v[0]; a[1] += v[1]

    def fill_staging(self):
        """The lanes' frames into BOTH pinned staging slots (what an image callback would do for every batch; the frames
        are the same every step, so it is done once, outside the timed region)."""
        for slot in (0, 1):
            sl, sr = self.pipe.staging(slot)
            for l, (_, Lh, Rh) in enumerate(self.data):
                sl[l, :self.B], sr[l, :self.B] = Lh, Rh
        self.slot = 0
        self.pipe.upload(0, self.B)

    def step_streaming(self):
        """One step fed from HOST memory: the upload of the next step's batch is started first (copy stream), then the batch
        uploaded during the previous step is processed."""
        self.pipe.reset()
        nxt = self.slot ^ 1
        self.pipe.upload(nxt, self.B)
        self.all_res = self.pipe.process_uploaded(self.slot, self.B)
        self.res = self.all_res[0]
        self.raws.append(self.pipe.last_raw)
        self.slot = nxt

    def close(self):
        self.pipe.close()
        self.ctx.close()


def group_lines(n_groups):
    """HIP streams ("lines") per pipeline group so that the process stays within its 16 hardware queues (streams that share a queue
    serialise: 96 lanes in 3 groups gave 20.6-21.7 k frames/s with 5 lines per group and 15.7-18.2 k with 6): 2 groups take the library's
    defaults (1 tracking + 2 chain + 4 solve lines each, round 4's sweep), 3 or more groups 1 + 1 + 2 and ONE compact line — solves the
    admission budget of the wide form refuses leave at once in the one-workgroup form on a line of their own (round 5; without it 96 lanes
    in 3 groups gave 19.8 k, 128 in 4 17.8 k: their lanes wait for admission).  Lanes x groups x lines measured in round 5:
    profiles/r05_exp_lanes_groups_honest.txt — a plateau at 20-22 k frames/s from 48 lanes in 2 groups to 120 in 3; the figures of
    profiles/r05_exp_lanes_groups.txt above that are retracted (threads of dead groups were counted, see its header).
    Environment variables set by the caller win."""
    if n_groups >= 3:
        os.environ.setdefault("SVO_GROUP_CHAIN_LINES", "1")
        os.environ.setdefault("SVO_GROUP_BA_LINES", "2")
        os.environ.setdefault("SVO_GROUP_COMPACT_LINES", "1")


def run_kitti(args):
    import threading
    import stereo_vo_amd as S
    torch, dist, rank, local, world = dist_setup(args.gpus)
    B, NS = args.batch, max(1, args.streams)
    seeds = [0x5EED0001 + rank * 1024 + i for i in range(NS)]
    NG = max(0, min(args.groups, NS))
    group_lines(NG)
    single_pipe = None
    if NG > 0:  # pipeline groups: NG host threads, the streams dealt round-robin
        streams = [_Group(S, torch, local, seeds[gi::NG], B) for gi in range(NG)]
        if NS > 1 and not args.no_single:
            single_pipe = _Stream(S, torch, local, seeds[0], B)
    else:
        streams = [_Stream(S, torch, local, sd, B) for sd in seeds]
    torch.cuda.synchronize()
    ctx = streams[0].ctx
    dev = torch.device("cuda", local)

    def run_steps(k):
        """k steps on every stream; streams run concurrently (ctypes releases the GIL inside the library)."""
        if len(streams) == 1:
            for _ in range(k):
                streams[0].step()
            return
        def work(st, delay):
            if delay > 0:
                time.sleep(delay)  # inside the timed region: the groups' steps start out of phase (a group's step is a dense detection burst followed by a latency-bound tail)
            for _ in range(k):
                st.step()
        run_threads([lambda st=st, d=1e-3 * args.stagger_ms * i: work(st, d) for i, st in enumerate(streams)])

    run_steps(args.warmup)
    # single-stream rate (latency-bound: one sequential VO chain) measured first, in the same run
    single = None
    if NS > 1 and not args.no_single:
        one = single_pipe if single_pipe is not None else streams[0]
        for _ in range(max(1, args.warmup)):
            one.step()
        barrier_sync(torch, dist, ctx)
        n_single = max(args.steps, 100)  # ~1 s of a 9 ms step: a steadier figure than 20 steps give (and the GPU is not cold when the timed region starts)
        t0 = time.perf_counter()
        for _ in range(n_single):
            one.step()
        one.ctx.sync()
        dts = time.perf_counter() - t0
        single = {"value": B * n_single / dts, "unit": "frames/s", "ms_per_step": 1e3 * dts / n_single, "steps": n_single,
                  "note": "one stream alone on the GPU (per rank)"}
    ctx.profile_select(args.profile_kernel)
    if NG > 1:
        streams[1].ctx.profile_select("ba_step")  # the solve launches of the second group, timed live as well (HIP events on its solve lines)
    for st in streams:
        if hasattr(st, "clear_counters"):
            st.clear_counters()
    if NG > 1:
        streams[1].pipe.solve_work(reset=True)
    barrier_sync(torch, dist, ctx)
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    run_steps(args.steps)
    for st in streams:
        st.ctx.sync()
    barrier_sync(torch, dist, ctx)
    dt = time.perf_counter() - t0
    host_cores = (time.process_time() - cpu0) / dt  # CPU time of all threads of this rank over the timed region
    k_ms, k_n = ctx.profile_read()
    ctx.profile_select(None)
    ba_ms, ba_n, ba_work = 0.0, 0, None
    if NG > 1:
        ba_ms, ba_n = streams[1].ctx.profile_read()
        streams[1].ctx.profile_select(None)
        ba_work = streams[1].pipe.solve_work()
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
        "n_gpus": (dist.get_world_size() if dist is not None else 1), "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64",
        "data": "synthetic",
        "config": {"workload": "kitti_1241x376_1500corners_5kf_window (BASELINE configs[1])", "batch_per_stream": B,
                   "streams_per_gpu": NS, "max_corners": MAXC, "quality": QUALITY, "min_distance": MIN_DIST, "window": WINDOW,
                   "keyframes_per_step": n_kf, "mean_tracked": float(np.mean(n_trk)) if n_trk else 0.0,
                   "ba_lm_iterations_per_step": ba_it, "host_cores_busy": round(host_cores, 2),
                   "pipeline_groups": NG, "hardware_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "0")),
                   "sharding": "independent stereo streams: streams_per_gpu per rank, ranks hold different streams; no collective"},
    }
    if single is not None:
        out["single_stream"] = single
    out["config"]["label"] = (f"{NS} concurrent stream(s) per GPU" + (f" as {NG} pipeline group(s), one host thread each" if NG else ", one host thread each") +
                              f", inputs resident in HBM (same {B} frames per step from a reset pipeline)")
    # roofline: (a) the SURVEY 8(d) contract figure of the whole front end, (b) the dominant kernel on the resource
    # that binds it (HIP events on the library's stream over the timed region + the committed SQ counter pass)
    avg_us = 1e3 * k_ms / k_n if k_n > 0 else None
    lanes_per_launch = 1.0
    g0_stats = streams[0].stats_sum if NG else None  # summed over the timed region
    if g0_stats and g0_stats.get("track", [0, 0])[0]:
        lanes_per_launch = g0_stats["track"][1] / g0_stats["track"][0]
    out["roofline"] = front_end_roofline(frames / dt / world, args.profile_kernel, avg_us, res, k_n, lanes_per_launch, bool(NG))
    share = profile_summary()
    if share and share.get("valu_wave_instructions_per_frame"):
        # the resource the whole path comes closest to: wave-level VALU instructions of every kernel per stereo frame (committed
        # SQ_INSTS_VALU pass of this workload, profiles/r05_sq_counters.txt) x this run's frames/s against the chip's VALU issue rate
        vpf = float(share["valu_wave_instructions_per_frame"])
        g = vpf * (frames / dt / world) / 1e9
        byk = share.get("valu_wave_instructions_per_frame_by_kernel") or {}
        lk = float(sum(v for k, v in byk.items() if k.startswith("lk_fb")))
        # SIMD cycles: the tracker's iteration is 81 half-rate (4.3 cycles) + 51 full-rate (2.5) instructions = 3.6 cycles per instruction
        # (tools/exp/issue_rate.hip on this GPU, profiles/r05_exp_issue_rate.txt, and the kernel's ISA); the other kernels at the mid-point 3.4
        simd_cycles_per_frame = lk * VALU_CYCLES_TRACKER_MIX + (vpf - lk) * VALU_CYCLES_OTHER_KERNELS
        ceiling = 256 * 4 * 2.4e9 / simd_cycles_per_frame
        out["roofline"]["chip_valu_issue"] = {
            "bound": "valu_issue", "valu_wave_instructions_per_frame": vpf, "by_kernel_per_frame": byk,
            "achieved_ginstr_s": g, "peak_ginstr_s": VALU_ISSUE_PEAK_GINSTR, "frac_of_full_rate_issue": g / VALU_ISSUE_PEAK_GINSTR,
            "simd_cycles_per_frame": simd_cycles_per_frame, "frames_per_s_if_every_simd_computed_all_the_time": ceiling,
            "frac": (frames / dt / world) / ceiling,
            "note": "per GPU; every kernel of the path together (the tracker ~70 %); full-rate peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 "
                    "instruction, but 60 % of the tracker's instructions are half-rate on gfx950: frac prices them at their measured cost"}
    if share:
        out["kernel_time_share"] = {"source": "profiles/r05_kernel_stats_*.csv (rocprofv3 --kernel-trace --stats of this command)",
                                    "default_percent": share.get("kernel_time_share_default"),
                                    "1_stream_percent": share.get("kernel_time_share_1_stream")}
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
        if NG > 0:
            out["parity_vs_cpu"].update(group_parity_vs_cpu(streams, B, ores))
    if NG > 0:
        out["parity_self"] = group_self_parity(streams, B)
        launches_sum = {k: list(v) for k, v in streams[0].stats_sum.items()}
    if NG > 0 and not args.no_streaming:
        # the same lanes fed from host memory: every step uploads its batch over PCIe (pinned staging slots of the groups,
        # the upload of step s+1 overlapped with the processing of step s); results must equal the resident ones bit for bit
        resident_raw = [g.raws[0] if g.raws else None for g in streams]
        for g in streams:
            g.fill_staging()
            g.clear_counters()

        def swork(st, k):
            for _ in range(k):
                st.step_streaming()
        def run_streaming(k):
            run_threads([lambda st=st: swork(st, k) for st in streams])
        run_streaming(max(1, args.warmup))
        for g in streams:
            g.clear_counters()
        barrier_sync(torch, dist, ctx)
        t1 = time.perf_counter()
        run_streaming(args.steps)
        for st in streams:
            st.ctx.sync()
        barrier_sync(torch, dist, ctx)
        dts = time.perf_counter() - t1
        same = all(r is not None and all(raw == r for raw in g.raws) for g, r in zip(streams, resident_raw))
        out["streaming"] = {"value": NS * B * args.steps / dts, "unit": "frames/s", "ms_per_step": 1e3 * dts / args.steps,
                            "fraction_of_resident": (NS * B * args.steps / dts) / (frames / world / dt),
                            "pcie_gbs": 2.0 * NS * B * W * H * args.steps / dts / 1e9,
                            "bit_identical_to_resident": bool(same),
                            "note": "per rank; svo_pipeline_group_upload / _process_uploaded: host images in the groups' pinned staging slots, H2D of "
                                    "step s+1 on a copy stream while step s is processed"}
    if NG > 0:
        out["config"]["launches_per_step_of_group_0"] = {k: [round(v[0] / args.steps, 1), round(v[1] / args.steps, 1)] for k, v in launches_sum.items()}
        if ba_n and ba_work:
            # the kernel with the largest share of summed kernel time: whole window solves, one launch for the lanes that are ready.
            # achieved = SURVEY 8(d) algorithmic f64 flops (and bytes) of the solves the launches carried / their summed duration
            # (HIP events on the solve lines of the second group); traffic = FETCH_SIZE + WRITE_SIZE per launch from the committed
            # counter pass of this command (profiles/r04_traffic.json)
            fl, by, n_solves, n_its = ba_work
            us = 1e3 * ba_ms / ba_n
            tf = fl / (ba_ms * 1e-3) / 1e12
            prof = profile_summary() or {}
            tr = (prof.get("ba_lm_traffic_bytes_per_launch") or None)  # the wide kernel's own FETCH_SIZE + WRITE_SIZE per launch (committed counter pass)
            dk = {"kernel": "ba_lm_kernel (+ ba_lm_compact_kernel for the solves the admission budget refuses)", "bound": "latency (tagged hand-overs between the workgroups of a solve; FP64 VALU inside a pass)",
                  "avg_launch_us": us, "launches": ba_n, "solves": n_solves, "lm_iterations": n_its,
                  "algorithmic_flops_per_launch": fl / ba_n, "algorithmic_bytes_per_launch": by / ba_n,
                  "achieved": tf, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VEC_PEAK_TFLOPS,
                  "hbm_frac_algorithmic": by / (ba_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr,
                  "traffic_over_algorithmic": (tr / (by / ba_n)) if tr else None,
                  "time_share_percent": {k: v for k, v in (prof.get("kernel_time_share_default") or {}).items() if k.startswith("ba_lm")},
                  "workload_definition": "r05: device-resident solves only in the work count (round 4 counted host-driven ones too)",
                  "measured": "HIP events on the solve lines of the second pipeline group over the timed region; per launch = per "
                              "%.1f solves of %.1f LM iterations" % (n_solves / ba_n, n_its / max(n_solves, 1))}
            out["roofline"]["tracker_kernel"] = out["roofline"].pop("dominant_kernel", None)
            out["roofline"]["dominant_kernel"] = dk
    for st in streams:
        st.close()
    if single_pipe is not None:
        single_pipe.close()
    if dist is not None:
        dist.destroy_process_group()
    return out if rank == 0 else None


def fixture_parity(name, results, rank):
    """EVERY frame of a long stream against the oracle-derived fixture tests/golden/stream_keys_<name>.npz (generated once in the build
    container by tests/golden/gen_stream_keys.py: the CPU oracle over all frames).  Counters, av_parallax bits and
    pose bits per frame.  Rank 0's stream only (the other ranks render another seed)."""
    if rank != 0:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stream_configs as SC
    keys = SC.load_keys(name)
    if keys is None:
        return {"fixture": f"tests/golden/stream_keys_{name}.npz", "frames": 0, "note": "fixture not present"}
    n = min(len(keys), len(results))
    bad = SC.compare(keys, results[:n])
    return {"fixture": f"tests/golden/stream_keys_{name}.npz", "frames": n, "all_frames_identical_to_the_oracle": not bad, "first_differing_frames": bad}


def window_load(kfs, window):
    """Observations and live landmarks of the sliding window behind every keyframe (steady state: full windows only): a keyframe
    adds n_inliers observations of known landmarks and n_new new ones (src/bundle_adjuster.cpp:72-121)."""
    obs = [k.n_inliers + k.n_new for k in kfs]
    o, l = [], []
    for i in range(window - 1, len(kfs)):
        w = range(i - window + 1, i + 1)
        o.append(sum(obs[j] for j in w))
        l.append(kfs[i - window + 1].n_inliers + sum(kfs[j].n_new for j in w))
    return {"mean_observations_per_window": float(np.mean(o)) if o else 0.0, "mean_live_landmarks_per_window": float(np.mean(l)) if l else 0.0,
            "mean_detected": 0.0 if not kfs else float(np.mean([k.n_detected for k in kfs]))}


def run_kitti_stream(args):
    """BASELINE configs[2]: one KITTI-00-shaped stream (default 4541 frames) through ONE pipeline with a 10-keyframe
    window, handed over as HOST buffers batch by batch (svo_pipeline_process_batch: the PCIe upload is inside the timed
    region), nothing reset.  A step = one batch of 16 consecutive frames.  N GPUs: every rank runs its own stream."""
    from concurrent.futures import ThreadPoolExecutor
    import stereo_vo_amd as S
    torch, dist, rank, local, world = dist_setup(args.gpus)
    B, window = args.batch, 10
    n_frames = max(B, args.frames)
    steps = (n_frames + B - 1) // B
    # BASELINE configs[2] asks for ~5k live landmarks in a 10-keyframe window (~15k observations): on the synthetic
    # street that takes 2,500 corners at quality 0.005 / minDistance 8 (measured with the oracle: 7.5k landmarks, 14k observations
    # per window; configs[1]'s 1500 / 0.02 / 10 give 4.5k / 8.6k)
    SMAXC, SQUAL, SMIND, SMAXF = 2800, 0.004, 7.0, 3300
    ctx = S.Context(W, H, device=local, max_batch=B, max_corners=SMAXC, max_candidates=1 << 16, max_features=SMAXF)
    p = S.synth_default(W, H)
    p.seed += rank  # the default scene of the generator (the one tools/soak_long_stream.py and the KITTI driver test use)
    pp = S.pipeline_default_params()
    pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
    pp.width, pp.height = W, H
    pp.max_corners, pp.quality, pp.min_feature_distance, pp.max_features, pp.window_size = SMAXC, SQUAL, SMIND, SMAXF, window
    pp.ba_max_time_s = 0.0
    pipe = S.Pipeline(ctx, pp)
    # the whole stream is rendered up front into host memory (the synthetic renderer is test input, not the path)
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 16)) as ex:
        fr = list(ex.map(lambda i: S.synth_render(p, i), range(n_frames)))  # ctypes releases the GIL
    L = np.stack([f[0] for f in fr])
    R = np.stack([f[1] for f in fr])
    del fr
    # warm-up on a SEPARATE pipeline (code objects, workspaces): the measured stream starts from frame 0, cold state
    wp = S.Pipeline(ctx, pp)
    for _ in range(max(1, args.warmup)):
        wp.reset()
        wp.process_batch(L[:B], R[:B])
    wp.close()
    barrier_sync(torch, dist, ctx)
    t0 = time.perf_counter()
    n_kf, est, gt, res_all = 0, [], [], []
    for f0 in range(0, n_frames, B):
        res = pipe.process_batch(L[f0:f0 + B], R[f0:f0 + B])
        res_all.extend(res)
    ctx.sync()
    barrier_sync(torch, dist, ctx)
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    for i, r in enumerate(res_all):
        n_kf += r.is_keyframe
        if r.is_keyframe and r.pose7[0] != 0:
            w, x, y, z = r.pose7[:4]
            tt = np.array(r.pose7[4:])
            Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                           [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                           [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            est.append(-Rm.T @ tt)  # camera in world (src/vo_node.cpp:149-150)
            gt.append(S.synth_pose(p, i)[:, 3])
    ate = float(S.api.ate_rmse(np.array(est), np.array(gt), False)) if len(est) >= 3 else None
    path = float(np.linalg.norm(np.diff(np.array(gt), axis=0), axis=1).sum()) if len(gt) >= 2 else 0.0
    frames = world * n_frames
    out = {"metric": "stereo frames/sec on 1241x376 KITTI pairs", "value": frames / dt, "unit": "frames/s", "n_gpus": (dist.get_world_size() if dist is not None else 1),
           "steps": steps, "warmup": max(1, args.warmup), "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8/f32/f64", "data": "synthetic",
           "config": {"workload": f"kitti00_shaped_stream_{n_frames}_frames_10kf_window (BASELINE configs[2])",
                      "label": "ONE stream per GPU, host buffers in (PCIe upload timed), nothing reset", "batch": B, "window": window,
                      "max_corners": SMAXC, "quality": SQUAL, "min_distance": SMIND,
                      "keyframes": n_kf, "mean_tracked": float(np.mean([r.n_tracked for r in res_all if r.n_tracked] or [0])),
                      **window_load([r for r in res_all if r.is_keyframe], window),
                      "ba_lm_iterations": int(sum(r.ba_iterations for r in res_all)),
                      "ate_rmse_m_at_keyframes_vs_generator": ate, "path_length_m": path,
                      "pcie_bytes_per_step": 2 * B * W * H}}
    out["roofline"] = front_end_roofline(frames / dt / world, args.profile_kernel, None, res_all, 0)
    if (SMAXC, SQUAL, SMIND, SMAXF, window) == (2800, 0.004, 7.0, 3300, 10):  # = tests/stream_configs.py "kitti_bench"
        out["parity_vs_oracle_fixture"] = fixture_parity("kitti_bench", res_all, rank)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cores = min(os.cpu_count() or 1, 16)
        os.environ["OMP_NUM_THREADS"] = str(cores)
        n = min(args.cpu_frames * 2, n_frames)
        op = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=W, height=H, max_corners=SMAXC, quality=SQUAL,
                        min_feature_distance=SMIND, parallax_thresh=20.0, window_size=window, max_features=SMAXF,
                        ba_max_iterations=50, num_threads=cores)
        t1 = time.perf_counter()
        ores = [op.process(L[i], R[i]) for i in range(n)]
        dtc = time.perf_counter() - t1
        same = all((a.n_detected, a.n_tracked, a.n_inliers, a.n_new, a.is_keyframe, a.ba_iterations) ==
                   (b.n_detected, b.n_tracked, b.n_inliers, b.n_new, b.is_keyframe, b.ba_iterations) and
                   list(a.pose7) == list(b.pose7) for a, b in zip(res_all[:n], ores))
        out["cpu_baseline"] = dict(value=n / dtc, unit="frames/s", cores=cores, kind="port",
                                   sample=f"first {n} frames of the same stream, whole oracle pipeline (10-keyframe window), {dtc:.1f} s")
        out["parity_vs_cpu"] = {"frames": n, "index_sets_and_poses_identical": bool(same)}
    pipe.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    return out if rank == 0 else None


HDW, HDH = 1280, 720


def run_hd10k(args):
    """BASELINE configs[4]: synthetic 1280x720 stereo stream (d435i focal, config/d435i.yaml:1,4), ~10 k features per frame
    (max_corners 10000, quality 0.001, minDistance 4: the cap is reached on the synthetic scene), 10-keyframe window, ONE stream resident in HBM, 64 frames in steps of 16."""
    import stereo_vo_amd as S
    torch, dist, rank, local, world = dist_setup(args.gpus)
    B, n_frames = 16, 64
    ctx = S.Context(HDW, HDH, device=local, max_batch=B, max_corners=10240, max_candidates=1 << 17, max_features=10240)
    p = S.synth_default(HDW, HDH)
    p.focal, p.cx, p.cy, p.baseline = 385.7545, 640.0, 360.0, 0.05
    p.step_z = 0.25  # 60 frames/s x 0.25 m = 15 m/s (the generator's default 0.8 m per frame is KITTI's 10 Hz)
    p.seed += rank
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 16)) as ex:
        fr = list(ex.map(lambda i: S.synth_render(p, i), range(n_frames)))
    L, R = np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])
    pp = S.pipeline_default_params()
    pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
    pp.width, pp.height = HDW, HDH
    pp.max_corners, pp.quality, pp.min_feature_distance, pp.max_features, pp.window_size = 10000, 0.001, 4.0, 10000, 10
    pp.ba_max_time_s = 0.0
    dev = torch.device("cuda", local)
    dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    pipe = S.Pipeline(ctx, pp)
    istride = HDW * HDH

    def one_pass():
        pipe.reset()
        res = []
        for f0 in range(0, n_frames, B):
            res += pipe.process_batch_dev(dL.data_ptr() + f0 * istride, dR.data_ptr() + f0 * istride, B)
        return res
    one_pass()  # warm-up
    barrier_sync(torch, dist, ctx)
    t0 = time.perf_counter()
    res = one_pass()
    ctx.sync()
    barrier_sync(torch, dist, ctx)
    dt = time.perf_counter() - t0
    A = HDW * HDH
    gbs = n_frames / dt * 6.33 * A / 1e9
    out = {"metric": "stereo frames/sec on 1280x720 pairs, ~10k features", "value": world * n_frames / dt, "unit": "frames/s",
           "n_gpus": (dist.get_world_size() if dist is not None else 1), "steps": n_frames // B, "warmup": 1,
           "ms_per_step": 1e3 * dt / (n_frames // B), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8/f32/f64", "data": "synthetic",
           "config": {"workload": "hd_1280x720_10k_features_10kf_window (BASELINE configs[4])", "frames": n_frames, "batch": B,
                      "keyframes": int(sum(r.is_keyframe for r in res)),
                      "mean_tracked": float(np.mean([r.n_tracked for r in res if r.n_tracked] or [0])),
                      "mean_detected": float(np.mean([r.n_detected for r in res])),
                      **{k: v for k, v in window_load([r for r in res if r.is_keyframe], 10).items() if k != "mean_detected"},
                      "ba_lm_iterations": int(sum(r.ba_iterations for r in res)), "fps_vs_60": n_frames / dt / 60.0},
           "roofline": {"bound": "hbm", "kernel": "whole front end per stereo pair (6.33 A bytes, SURVEY 8d)", "achieved": gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_pair": 6.33 * A, "traffic": None}}
    out["parity_vs_oracle_fixture"] = fixture_parity("hd10k", res, rank)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cores = min(os.cpu_count() or 1, 16)
        os.environ["OMP_NUM_THREADS"] = str(cores)
        n = 6
        op = O.Pipeline(focal=p.focal, cx=p.cx, cy=p.cy, baseline=p.baseline, width=HDW, height=HDH, max_corners=10000, quality=0.001,
                        min_feature_distance=4.0, parallax_thresh=20.0, window_size=10, max_features=10000, ba_max_iterations=50, num_threads=cores)
        t1 = time.perf_counter()
        ores = [op.process(L[i], R[i]) for i in range(n)]
        dtc = time.perf_counter() - t1
        same = all((a.n_detected, a.n_tracked, a.n_inliers, a.n_new, a.is_keyframe, a.ba_iterations) ==
                   (b.n_detected, b.n_tracked, b.n_inliers, b.n_new, b.is_keyframe, b.ba_iterations) and
                   list(a.pose7) == list(b.pose7) for a, b in zip(res[:n], ores))
        out["cpu_baseline"] = dict(value=n / dtc, unit="frames/s", cores=cores, kind="port",
                                   sample=f"first {n} frames of the same stream, whole oracle pipeline, {dtc:.1f} s")
        out["parity_vs_cpu"] = {"frames": n, "index_sets_and_poses_identical": bool(same)}
    pipe.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    return out if rank == 0 else None


def other_workloads(args):
    """Driver-timed figures of BASELINE configs[2], [3] (sparse and dense) and [4] inside the default line: each in its own
    sub-object with its own roofline, cpu_baseline sample and parity flag; bounded so that the whole default run stays
    within a few minutes."""
    import copy
    out = {}

    def sub(name, fn):
        t0 = time.perf_counter()
        try:
            r = fn()
            keep = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline", "cpu_baseline", "parity_vs_cpu", "parity_vs_oracle_fixture") if k in r}
            keep["wall_s"] = round(time.perf_counter() - t0, 1)
            out[name] = keep
        except Exception as e:  # a sub-workload must never take the headline down with it
            out[name] = {"error": f"{type(e).__name__}: {e}", "wall_s": round(time.perf_counter() - t0, 1)}

    a = copy.copy(args)
    a.batch, a.cpu_frames = 16, 12
    sub("kitti_stream_4541_frames", lambda: run_kitti_stream(a))
    from tools import bench_ba
    b = copy.copy(args)
    b.steps, b.warmup = 20, 3
    os.environ.pop("SVO_BA_DENSE", None)
    sub("ba50k_sparse", lambda: bench_ba.run(b, cpu_seconds=6.0))
    os.environ["SVO_BA_DENSE"] = "1"
    sub("ba50k_dense", lambda: bench_ba.run(b, cpu_seconds=6.0))
    os.environ.pop("SVO_BA_DENSE", None)
    sub("hd_1280x720_10k", lambda: run_hd10k(args))
    return out


def profile_summary():
    try:
        return json.load(open(PROFILE_SUMMARY))
    except Exception:
        return None


def front_end_roofline(pairs_per_s_per_gpu, kernel, avg_us, res, launches, lanes_per_launch=1.0, grouped=False):
    """one stereo pair needs 6.33 A bytes of compulsory HBM traffic (corner 1 A + stereo 4 A as a dense map
    + LK pyramid 1.33 A); achieved = pairs/s x 6.33 A.  The path is latency / instruction-issue bound, not HBM bound, so
    the dominant kernel is also reported against the resource that binds it (VALU issue)."""
    A = W * H
    contract = 6.33 * A
    gbs = pairs_per_s_per_gpu * contract / 1e9
    prof = profile_summary() or {}
    r = {"bound": "hbm", "kernel": "whole front end per stereo pair (corner 1 A + StereoBM 4 A + LK pyramid 1.33 A)",
         "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "frac_of_measured_copy": gbs / HBM_COPY_GBS, "algorithmic_bytes_per_pair": contract,
         # HBM bytes per pair from the committed FETCH_SIZE / WRITE_SIZE passes of this command (tools/prof_summary.py: every
         # front-end kernel of the trace, matched by base name): `traffic` is the figure the guide prescribes — FETCH_SIZE doubled
         # on the kernels that stream whole images with 16-byte requests — next to the raw counters
         "traffic": prof.get("front_end_hbm_bytes_per_pair_pmc_fetch_x2_on_streaming_kernels"),
         "traffic_raw_counters": prof.get("front_end_hbm_bytes_per_pair_pmc"),
         "traffic_over_algorithmic": (prof.get("front_end_hbm_bytes_per_pair_pmc_fetch_x2_on_streaming_kernels") / contract) if prof.get("front_end_hbm_bytes_per_pair_pmc_fetch_x2_on_streaming_kernels") else None,
         "traffic_by_kernel_raw": prof.get("front_end_hbm_bytes_per_pair_by_kernel"),
         "note": "per GPU; the front end is latency / instruction-issue bound at these sizes (SURVEY 8d 'honest expectation'): "
                 "the contract fraction is reported as asked, the binding resource of the dominant kernel is below"}
    if avg_us:
        n = np.mean([x.n_tracked for x in res if x.n_tracked]) if any(x.n_tracked for x in res) else 0
        dk = {"kernel": kernel + ("_group_kernel" if grouped else "_kernel"), "avg_launch_us": avg_us, "launches": launches,
              "measured": "HIP events on the library's stream around every launch in the timed region"}
        if kernel == "lk_fb":
            # a grouped launch tracks the features of every lane that reached the stage together (one wavefront per feature)
            byts = 2 * 1.33 * A * lanes_per_launch  # both pyramids of every lane once
            dk.update({"lanes_per_launch": lanes_per_launch, "features_per_launch": float(n) * lanes_per_launch, "algorithmic_bytes_per_launch": byts,
                       "hbm_frac": byts / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bound": "valu_issue"})
            per_wave = prof.get("lk_fb_valu_instructions_per_wave")
            vi = per_wave * float(n) * lanes_per_launch if per_wave else prof.get("lk_fb_valu_wave_instructions_per_launch")
            if vi:
                g = vi / (avg_us * 1e-6) / 1e9
                dk.update({"valu_wave_instructions_per_launch": vi, "achieved_ginstr_s": g, "peak_ginstr_s": VALU_ISSUE_PEAK_GINSTR,
                           "frac": g / VALU_ISSUE_PEAK_GINSTR,
                           "note": "VALU instructions per wavefront (= per feature) from the committed SQ pass (profiles/r05_sq_counters.txt) x the "
                                   "features of a launch, duration live; <= 30 iterations x 4 levels x 2 directions of a 441-pixel window per feature"})
        r["dominant_kernel"] = dk
    return r


def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT the HIP runtime (the parent must stay a process that never mapped
    libamdhip64: on this pool an exec / fork+exec from a HIP-initialised process can take the machine down).  Order:
    SVO_BENCH_GPU_COUNT (tests / rehearsals) -> the *_VISIBLE_DEVICES lists the runtime itself honours -> the KFD
    topology in sysfs (nodes with simd_count > 0 are GPUs)."""
    forced = os.environ.get("SVO_BENCH_GPU_COUNT")
    if forced is not None:
        return int(forced)
    kfd = 0
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for node in os.listdir(base):
            try:
                props = open(os.path.join(base, node, "properties")).read()
            except OSError:
                continue
            m = re.search(r"^simd_count\s+(\d+)", props, re.M)
            if m and int(m.group(1)) > 0:
                kfd += 1
    except OSError:
        kfd = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            kfd = min(kfd, listed) if kfd else listed
    return kfd


def hip_runtime_mapped():
    """True when this process has libamdhip64 mapped (importing torch is enough for that)."""
    try:
        return "libamdhip64" in open("/proc/self/maps").read()
    except OSError:
        return False


def spawn_ranks(args):
    """`bench.py --gpus N` without a launcher: this parent makes NO GPU call and never maps the HIP runtime (no torch import:
    never an `exec` after HIP is initialised, never a fork of an initialised process); it counts the GPUs from the
    environment / sysfs, starts N ranks with torch.distributed.run and relays rank 0's JSON line."""
    import subprocess
    have = visible_gpu_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible: refusing to measure fewer ranks than asked", file=sys.stderr, flush=True)
        return 2
    if hip_runtime_mapped():
        print("bench.py: the launching process has the HIP runtime mapped; refusing to fork + exec the ranks from it", file=sys.stderr, flush=True)
        return 2
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if child.returncode != 0 or line is None:
        sys.stdout.write(child.stdout)
        print(f"bench.py: the {args.gpus}-rank run failed (exit code {child.returncode})", file=sys.stderr, flush=True)
        return child.returncode or 1
    print(line, flush=True)
    return 0


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (it spawns the ranks) or under "
              f"torch.distributed.run with --nproc-per-node equal to --gpus", file=sys.stderr, flush=True)
        sys.exit(2)
    if args.workload == "kitti_cfg1":
        out = run_kitti(args)
        if out is not None and world == 1 and not args.no_other_workloads:
            out["other_workloads"] = other_workloads(args)
    elif args.workload == "hd10k":
        out = run_hd10k(args)
    elif args.workload == "kitti_stream":
        out = run_kitti_stream(args)
    else:
        from tools import bench_ba
        out = bench_ba.run(args)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
