"""Developer aid: print the kernel timeline (start offset, duration, gap to the previous kernel on any queue) of a slice of
a rocprofv3 kernel_trace.csv: rows [skip, skip + count) in start order."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip, count = int(sys.argv[2]), int(sys.argv[3])
t0 = int(rows[skip]["Start_Timestamp"])
prev_end = t0
for r in rows[skip:skip + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f  dur %7.1f  q%-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0][:40]))
    prev_end = max(prev_end, e)
