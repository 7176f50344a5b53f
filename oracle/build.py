"""Build recipe of the CPU oracle (g++, OpenMP).  TEST INFRASTRUCTURE: used by tests/, bench.py's
cpu_baseline leg and __graft_entry__ only."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libsvo_oracle.so")


def build_oracle(force=False):
    srcs = sorted(glob.glob(os.path.join(HERE, "*.cpp")))
    deps = srcs + [os.path.join(HERE, "svo_oracle.h"), os.path.join(HERE, "ora_constants.h")]
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    stale = not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)
    if force or stale:
        cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-Wall",
               "-o", LIB] + srcs
        print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_oracle("--force" in sys.argv)
