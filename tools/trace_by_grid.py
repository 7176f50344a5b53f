"""Developer aid: from a rocprofv3 kernel_trace.csv, per (kernel, grid size): launches, mean / median duration, and the
mean gap to the previous kernel on the same queue."""
import csv
import collections
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_end = {}
gaps = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    key = (name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    by[key].append(e - s)
    q = r.get("Queue_Id", "0")
    if q in last_end:
        gaps[key].append(s - last_end[q])
    last_end[q] = e
tot = sum(sum(v) for v in by.values())
for key, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:30]:
    g = gaps.get(key, [0])
    print("%-28s grid %8d  n %6d  mean %8.1f us  median %8.1f us  share %5.1f %%  gap-before median %7.1f us" %
          (key[0][:28], key[1], len(v), statistics.mean(v) / 1e3, statistics.median(v) / 1e3, 100.0 * sum(v) / tot, statistics.median(g) / 1e3))
