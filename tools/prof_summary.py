"""Condense a tools/prof_round.sh output directory into the files that are committed under profiles/:
  <tag>_kernel_stats_8streams.csv / _1stream.csv / _ba50k.csv   rocprofv3 --stats tables (name, calls, total, average, %)
  <tag>_sq_counters.txt      per kernel and launch: instruction mix + parked / issue-stalled / issuing split
  <tag>_traffic.json         HBM KB per launch per kernel (FETCH_SIZE + WRITE_SIZE passes) and bytes per stereo pair
  <tag>_ba50k_mfma_pmc.txt   f64 MFMA counts and matrix-pipe busy cycles per launch
  <tag>_summary.json         what bench.py reads: kernel_time_share (8 streams / 1 stream), lk_fb VALU instructions per
                             launch, front-end HBM bytes per pair
usage: prof_summary.py <prof dir> <out dir> <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)


def kname(full):
    """Kernel name as it is classified below: no namespace, no return type, no template arguments, no parameter list —
    `void stereo_triangulate_group_kernel<64>(...)` and `stereo_triangulate_group_kernel(...)` are the same kernel (round 4's
    summary matched exact names and silently dropped the templated sparse-stereo kernel from the front-end bytes)."""
    n = full.replace("(anonymous namespace)::", "").split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0].strip()


# Every kernel of a trace belongs to exactly one class; a kernel nobody classified stops the summary (a rename can never
# silently drop bytes or time again).  "wide": streams whole images with 16-byte requests (the guide's x2 on FETCH_SIZE applies).
FRONT_END = {"corner_response_kernel", "corner_response_nms_kernel", "corner_threshold_kernel", "corner_nms_kernel", "corner_select_kernel",
             "pyr_copy_kernel", "pyr_down_kernel", "pyr_build_kernel", "lk_fb_kernel", "lk_fb_group_kernel", "track_compact_kernel", "tracker_init_kernel",
             "tracker_init_group_kernel", "stereo_at_kernel", "stereo_triangulate_kernel", "stereo_triangulate_group_kernel", "stereo_prefilter_kernel",
             "stereo_dense_kernel", "dedup_group_kernel", "dedup_kernel", "triangulate_kernel", "gather_track_kernel", "lk_kernel", "chain_group_kernel"}
WIDE = {"corner_response_kernel", "corner_response_nms_kernel", "corner_nms_kernel", "pyr_copy_kernel", "pyr_down_kernel", "pyr_build_kernel"}
SOLVE = {"ba_lm_kernel", "ba_lm_compact_kernel", "ba_step_kernel", "ba_decide_linearize_kernel", "ba_reduce_kernel", "ba_linearize_det_kernel", "ba_linearize_kernel",
         "ba_linearize_mfma_kernel", "ba_backsub_kernel", "ba_decide_kernel", "ba_bulk_control_kernel", "ba_store_scatter_kernel", "pnp_ransac_kernel", "pnp_group_kernel"}
OTHER_PREFIXES = ("__amd_rocclr_", "peak_", "cholesky_solve_kernel", "reproj_", "nccl", "rccl", "msccl", "at::", "void at::", "vectorized_elementwise", "elementwise_kernel")


def classify(k):
    if k in FRONT_END:
        return "front_end"
    if k in SOLVE:
        return "solve"
    if k.startswith(OTHER_PREFIXES) or "at::native" in k or "Generic_" in k:
        return "other"
    return None


def check_classified(names, where):
    unknown = sorted({k for k in names if classify(k) is None})
    if unknown:
        raise SystemExit(f"prof_summary: unclassified kernel(s) in {where}: {unknown} — add them to FRONT_END / SOLVE / OTHER_PREFIXES")


def stats(path):
    rows = list(csv.DictReader(open(path)))
    return [(kname(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])) for r in rows]


summary = {}
stamp = open(os.path.join(src, "STAMP.txt")).read().strip() if os.path.exists(os.path.join(src, "STAMP.txt")) else "unknown"
summary["stamp"] = stamp
for key, sub in (("default", "s8"), ("1_stream", "s1"), ("ba50k", "ba")):
    f = os.path.join(src, sub, "p_kernel_stats.csv")
    if not os.path.exists(f):
        continue
    st = stats(f)
    check_classified([n for n, c, t, a, p in st], f)
    with open(os.path.join(out, f"{tag}_kernel_stats_{key}.csv"), "w") as w:
        w.write("# " + stamp + "\n")
        w.write("kernel,calls,total_ms,average_us,percent\n")
        for n, c, t, a, p in st:
            w.write(f"{n},{c},{t / 1e6:.3f},{a / 1e3:.2f},{p:.2f}\n")
    summary["kernel_time_share_" + key] = {n: round(p, 2) for n, c, t, a, p in st if p >= 0.3}
    summary["kernel_average_us_" + key] = {n: round(a / 1e3, 2) for n, c, t, a, p in st if p >= 0.3}
for name in ("bench_s8.json", "bench_s1.json", "bench_ba.json", "gpu_busy_s8.txt", "STAMP.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(out, f"{tag}_{name}"))


def counters(sub):
    f = glob.glob(os.path.join(src, sub, "*counter_collection.csv"))
    if not f:
        return {}, {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for row in csv.DictReader(open(f[0])):
        k = kname(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            calls[k] += 1
    return agg, calls


agg, calls = counters("sq")
if agg:
    with open(os.path.join(out, f"{tag}_sq_counters.txt"), "w") as w:
        w.write("# " + stamp + "\n")
        for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"])[:12]:
            a, c = agg[k], calls[k]
            wc = max(a["SQ_WAVE_CYCLES"], 1.0)
            w.write("%s calls=%d per launch: waves=%.0f valu=%.0f lds=%.0f salu=%.0f wave_cycles(quad)=%.0f | of wave cycles: parked "
                    "(waitcnt/barrier) %.1f%%, issue-stalled %.1f%%, issuing %.1f%%\n"
                    % (k, c, a["SQ_WAVES"] / c, a["SQ_INSTS_VALU"] / c, a["SQ_INSTS_LDS"] / c, a["SQ_INSTS_SALU"] / c, a["SQ_WAVE_CYCLES"] / c,
                       100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc))
    for lk in ("lk_fb_group_kernel", "lk_fb_kernel"):
        if lk in agg:  # one wavefront per feature (+ idle ones of narrower lanes): instructions per launched wave
            summary["lk_fb_kernel_profiled"] = lk
            summary["lk_fb_valu_wave_instructions_per_launch"] = agg[lk]["SQ_INSTS_VALU"] / calls[lk]
            summary["lk_fb_waves_per_launch"] = agg[lk]["SQ_WAVES"] / calls[lk]
            summary["lk_fb_valu_instructions_per_wave"] = agg[lk]["SQ_INSTS_VALU"] / max(agg[lk]["SQ_WAVES"], 1.0)
            break
    # chip-wide VALU issue: wave-level VALU instructions of EVERY kernel of the pass per stereo frame (the pass ran
    # `bench.py --steps 3 --warmup 1 --streams 32 --groups 1`: frames from its own line) — bench.py multiplies by its frames/s
    sq_frames = 4 * 16 * 32
    try:
        for ln in open(os.path.join(src, "bench_sq.log")):
            if ln.startswith('{"metric"'):
                b = json.loads(ln)
                sq_frames = (b["steps"] + b["warmup"]) * b["config"]["batch_per_stream"] * b["config"]["streams_per_gpu"]
    except Exception:
        pass
    summary["valu_wave_instructions_per_frame"] = sum(agg[k]["SQ_INSTS_VALU"] for k in agg) / sq_frames
    summary["valu_wave_instructions_per_frame_by_kernel"] = {k: round(agg[k]["SQ_INSTS_VALU"] / sq_frames) for k in sorted(agg, key=lambda k: -agg[k]["SQ_INSTS_VALU"])[:8]}
    for sk in ("ba_lm_compact_kernel", "ba_lm_kernel"):
        if sk in agg:
            pre = "ba_lm" if sk == "ba_lm_kernel" else "ba_lm_compact"
            summary[pre + "_valu_wave_instructions_per_launch"] = agg[sk]["SQ_INSTS_VALU"] / calls[sk]
            summary[pre + "_waves_per_launch"] = agg[sk]["SQ_WAVES"] / calls[sk]
            summary[pre + "_parked_fraction"] = agg[sk]["SQ_WAIT_ANY"] / max(agg[sk]["SQ_WAVE_CYCLES"], 1.0)

traffic = collections.defaultdict(dict)
launches = {}
for c, key in (("FETCH_SIZE", "fetch_kb_per_launch"), ("WRITE_SIZE", "write_kb_per_launch")):
    a, n = counters(c)
    for k in a:
        traffic[k][key] = a[k][c] / n[k]
        launches[k] = n[k]
if traffic:
    # the counter passes ran `bench.py --steps 3 --warmup 1 --streams 32 --groups 1 --no-single`: frames = (steps + warmup) x batch x lanes, from the run's own line
    frames = 4 * 16 * 32
    try:
        for ln in open(os.path.join(src, "bench_FETCH_SIZE.log")):
            if ln.startswith('{"metric"'):
                b = json.loads(ln)
                frames = (b["steps"] + b["warmup"]) * b["config"]["batch_per_stream"] * b["config"]["streams_per_gpu"]
    except Exception:
        pass
    check_classified(traffic.keys(), "the FETCH_SIZE / WRITE_SIZE passes")
    front = sorted(k for k in traffic if classify(k) == "front_end")
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; applied to the kernels
    # that stream whole images with 16-byte requests (response, NMS, pyramid), not to the byte-granular gathers
    wide = WIDE
    per_pair_corrected = sum(((2.0 if k in wide else 1.0) * traffic[k].get("fetch_kb_per_launch", 0) + traffic[k].get("write_kb_per_launch", 0)) * 1024.0 * launches[k]
                             for k in front if k in traffic) / frames
    summary["front_end_hbm_bytes_per_pair_pmc_fetch_x2_on_streaming_kernels"] = per_pair_corrected
    per_pair = sum((traffic[k].get("fetch_kb_per_launch", 0) + traffic[k].get("write_kb_per_launch", 0)) * 1024.0 * launches[k]
                   for k in front if k in traffic) / frames
    summary["front_end_hbm_bytes_per_pair_pmc"] = per_pair
    summary["front_end_kernels_counted"] = front
    summary["front_end_hbm_bytes_per_pair_by_kernel"] = {k: round((traffic[k].get("fetch_kb_per_launch", 0) + traffic[k].get("write_kb_per_launch", 0)) * 1024.0 * launches[k] / frames) for k in front}
    for k in traffic:
        traffic[k]["launches"] = launches[k]
    for sk in ("ba_lm_compact_kernel", "ba_lm_kernel"):  # the solve kernel's own HBM traffic per launch (bench.py sets it against the algorithmic bytes)
        if sk in traffic:
            pre = "ba_lm" if sk == "ba_lm_kernel" else "ba_lm_compact"
            summary[pre + "_traffic_bytes_per_launch"] = (traffic[sk].get("fetch_kb_per_launch", 0) + traffic[sk].get("write_kb_per_launch", 0)) * 1024.0
    json.dump(dict(stamp=stamp, unit="KB per launch, raw FETCH_SIZE / WRITE_SIZE (no 2x correction: byte-granular gathers)", frames=frames,
                   front_end_bytes_per_pair=per_pair, kernels=traffic), open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)

agg, calls = counters("ba_pmc")
if agg:
    with open(os.path.join(out, f"{tag}_ba50k_mfma_pmc.txt"), "w") as w:
        w.write("# " + stamp + "\n")
        for k in agg:
            w.write(k + " calls=%d " % calls[k] + " ".join("%s=%.0f" % (c, v / calls[k]) for c, v in sorted(agg[k].items())) + "  (per launch)\n")
json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary)[:600])
