"""Developer aid: from a rocprofv3 kernel_trace.csv, per time window: the fraction of wall time with >= 1
kernel resident (union of [start, end) intervals) and the average number of concurrently resident kernels."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
win = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 50_000_000
t0, t1 = iv[0][0], max(e for _, e in iv)
nw = (t1 - t0) // win + 1
busy = [0] * nw
tot = [0] * nw
cnt = [0] * nw
# union
merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])


def spread(lst, acc):
    for s, e in lst:
        w = (s - t0) // win
        while s < e:
            wend = t0 + (w + 1) * win
            seg = min(e, wend) - s
            acc[w] += seg
            s += seg
            w += 1


spread(merged, busy)
spread(iv, tot)
for s, _ in iv:
    cnt[(s - t0) // win] += 1
for w in range(nw):
    print("t=%6.0f ms  busy %5.1f %%  concurrency %.2f  launches %d" % (w * win / 1e6, 100.0 * busy[w] / win, tot[w] / max(busy[w], 1), cnt[w]))
