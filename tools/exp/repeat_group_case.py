"""Developer aid: one group parity case repeated in ONE process (an intermittent failure prints its assertion and stops)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_group as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
case = tuple(float(x) if "." in x else int(x) for x in sys.argv[2:6]) if len(sys.argv) >= 6 else (20, 6, 14.0, 300)
for i in range(n):
    t0 = time.perf_counter()
    try:
        T.test_hip_group_lanes_match_their_oracles(*case)
    except BaseException:  # noqa: BLE001
        print(f"run {i}: FAILED after {time.perf_counter() - t0:.1f} s", flush=True)
        traceback.print_exc()
        sys.exit(1)
    print(f"run {i}: ok {time.perf_counter() - t0:.1f} s", flush=True)
