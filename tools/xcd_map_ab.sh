#!/bin/bash
# FETCH_SIZE of the tracking and the sparse-stereo launches with and without the XCD-aware item -> workgroup map, same box, same
# command (one pipeline group of 24 streams, 3 steps); writes gpurun_out/xcd_map_ab.txt.  Counter pass on its own (--kernel-trace only).
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/xcd_map_ab.txt
echo "# $(sha256sum stereo_vo_amd/libsvo_hip.so | cut -c1-16) libsvo_hip.so; commit ${1:-unknown}; FETCH_SIZE in KB per launch (mean over the launches), 24 lanes in 1 group, 3 timed steps" > $OUT
for X in 1 0; do
  export SVO_GROUP_LK_XCD=$X SVO_GROUP_TRI_XCD=$X
  D=gpurun_out/xcd_ab_$X
  rm -rf "$D"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$D" -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-single --no-streaming --streams 24 --groups 1 > "$D.log" 2>&1
  python3 - "$D/p_counter_collection.csv" "$X" >> $OUT <<'PY'
import csv, sys, collections
s = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        s[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
for k in sorted(s):
    if "lk_fb_group" in k or "stereo_triangulate_group" in k or "pnp_group" in k:
        print("xcd map %s: %-60s %10.1f KB x %d launches" % ("ON " if sys.argv[2] == "1" else "OFF", k[:60], s[k] / n[k], n[k]))
PY
  rm -rf "$D"
done
cat $OUT
