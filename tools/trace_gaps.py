#!/usr/bin/env python3
"""Where a lane's time goes inside svo_pipeline_group_process_batch: reads the SVO_GROUP_TRACE=1 lines of a run (stderr) and
sums, over all lanes and calls, the time between consecutive events of a lane, by (event -> next event).  Usage:
  SVO_GROUP_TRACE=1 python bench.py ... 2> trace.txt ; python tools/trace_gaps.py trace.txt"""
import collections
import sys

calls, cur = [], []
for line in open(sys.argv[1], errors="replace"):
    if not line.startswith("[svo group]"):
        continue
    f = line.split()
    if f[3] == "end":
        calls.append((cur, float(f[2]))); cur = []
    elif f[3] == "lane":
        cur.append((float(f[2]), int(f[4]), f[5]))
gap = collections.defaultdict(float); cnt = collections.defaultdict(int)
total = lanes_total = 0.0
for evs, end in calls:
    by = collections.defaultdict(list)
    for t, lane, what in evs:
        by[lane].append((t, what))
    for lane, L in by.items():
        L.sort()
        prev_t, prev = 0.0, "begin"
        for t, what in L:
            gap[(prev, what)] += t - prev_t; cnt[(prev, what)] += 1
            prev_t, prev = t, what
        gap[(prev, "end")] += end - prev_t; cnt[(prev, "end")] += 1
        lanes_total += end
    total += end
print("calls %d, mean call %.1f ms, lane-time %.1f ms" % (len(calls), 1e-3 * total / max(1, len(calls)), 1e-3 * lanes_total))
for k, v in sorted(gap.items(), key=lambda kv: -kv[1]):
    print("%-24s -> %-24s %6.2f %% of lane time, %7d times, mean %8.1f us" % (k[0], k[1], 100 * v / lanes_total, cnt[k], v / cnt[k]))
