"""Build recipes: the gfx950 C-ABI library (hipcc) and, for tests/bench only, the CPU oracle (g++).

Everything is built in-tree (the .so files travel to the GPU box with the snapshot; they are
git-ignored).  No CMake: a handful of translation units compiled directly.
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo_vo_amd", "csrc")
HOST = os.path.join(ROOT, "stereo_vo_amd", "host")
LIB = os.path.join(ROOT, "stereo_vo_amd", "libsvo_hip.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libsvo_oracle.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
             "-I", CSRC, "-I", HOST]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_hip(force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(HOST, "*.cpp")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HOST, "*.hpp")) + \
        [os.path.join(ROOT, "include", "svo.h")]
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + [d for d in deps if d.endswith((".h", ".hpp"))]):
            lang = ["-x", "hip"] if s.endswith(".hip") else []
            cmd = [HIPCC] + HIP_FLAGS + lang + ["-c", s, "-o", o]
            print("+", " ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), s))
    for p, s in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    if force or procs or not os.path.exists(LIB):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"])
    return LIB


def build_oracle(force=False):
    srcs = sorted(glob.glob(os.path.join(ORACLE_DIR, "*.cpp")))
    deps = srcs + [os.path.join(ORACLE_DIR, "svo_oracle.h")]
    os.makedirs(os.path.dirname(ORACLE_LIB), exist_ok=True)
    if force or _newer(ORACLE_LIB, deps):
        _run(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp",
              "-Wall", "-o", ORACLE_LIB] + srcs)
    return ORACLE_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--oracle" in sys.argv or "--all" in sys.argv or len(sys.argv) == 1:
        build_oracle(force)
    if "--hip" in sys.argv or "--all" in sys.argv or len(sys.argv) == 1:
        build_hip(force)
