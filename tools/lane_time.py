"""Developer aid: where a lane's wall clock goes inside svo_pipeline_group_process_batch_dev, from the SVO_GROUP_TRACE=1 event
log (stderr lines "[svo group] <us> lane <l> <event> <arg>").  Sums, over all lanes and calls, the time between the events
that bracket each wait.  usage: lane_time.py <log>"""
import collections
import re
import sys

ev = re.compile(r"\[svo group\]\s+([0-9.]+) lane\s+(\d+) (\S+)\s+(-?\d+)")
end = re.compile(r"\[svo group\]\s+([0-9.]+) end")
calls, cur = [], collections.defaultdict(list)
for ln in open(sys.argv[1], errors="replace"):
    m = ev.search(ln)
    if m:
        cur[int(m.group(2))].append((float(m.group(1)), m.group(3), int(m.group(4))))
        continue
    m = end.search(ln)
    if m:
        calls.append((float(m.group(1)), cur))
        cur = collections.defaultdict(list)
tot = collections.Counter()
n_lane_calls = 0
wall = 0.0
pairs = {"track_launch": ("track_done", "tracking launch in flight"), "pnp_launch": ("pnp_done", "PnP-RANSAC launch in flight"),
         "tri_launch": ("tri_done", "stereo + triangulation in flight"),
         "need_solve": ("solve_joined_for_pnp", "keyframe waits for its previous solve")}
for t_end, lanes in calls:
    for l, evs in lanes.items():
        n_lane_calls += 1
        wall += t_end
        evs.sort()
        open_ = {}
        last_t = 0.0
        for t, what, arg in evs:
            for start, (stop, label) in pairs.items():
                if what == start:
                    open_[start] = t
                if what == stop and start in open_:
                    tot[label] += t - open_.pop(start)
            last_t = t
        tot["after the lane's last event (other lanes still running)"] += t_end - last_t
acc = sum(tot.values())
print(f"{len(calls)} calls, {n_lane_calls} lane-calls, mean call {wall / max(n_lane_calls, 1) / 1e3:.2f} ms")
for k, v in tot.most_common():
    print(f"  {k:58s} {v / n_lane_calls / 1e3:8.3f} ms per lane-call  ({100 * v / wall:5.1f} % of the lane's wall clock)")
print(f"  {'not bracketed (host turn-around, queued behind the bus)':58s} {(wall - acc) / n_lane_calls / 1e3:8.3f} ms per lane-call  ({100 * (wall - acc) / wall:5.1f} %)")
