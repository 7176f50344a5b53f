#!/usr/bin/env python3
"""Developer aid: T independent window-sized BA solves running concurrently (one adjuster + one host thread each, nothing
else on the GPU) -- time per LM iteration against T.  Separates adjuster-vs-adjuster interference from interference by the
tracker's wide launches.  Usage: python tools/ba_concurrent.py [T ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import stereo_vo_amd as S
import ba_problem as BP


def main():
    counts = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    ctx = S.api.Context(1241, 376, max_batch=1)
    reps = 30
    for T in counts:
        probs = [BP.make_problem(100 + t, 10, 450, dense=False) for t in range(T)]
        bas = [S.api.BA(ctx, 10, BP.F, BP.CX, BP.CY, max_landmarks=4096, max_observations=16384) for _ in range(T)]
        its = [0] * T
        secs = [0.0] * T
        bar = threading.Barrier(T)

        def run(t):
            p, ba = probs[t], bas[t]
            for r in range(reps + 3):
                ba.load_problem(p["poses0"], p["points0"], p["op"], p["oj"], p["uv"])
                bar.wait()
                t0 = time.perf_counter()
                s = ba.solve_problem()
                dt = time.perf_counter() - t0
                if r >= 3:
                    its[t] += s.iterations
                    secs[t] += dt
        th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        [x.start() for x in th]
        [x.join() for x in th]
        print(f"T={T}: {1e6 * sum(secs) / sum(its):7.1f} us per LM iteration per adjuster  "
              f"({sum(its) / reps / T:.1f} iterations, {len(probs[0]['op'])} observations)", flush=True)
        for b in bas:
            b.close()


if __name__ == "__main__":
    main()
