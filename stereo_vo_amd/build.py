"""Build recipe of the gfx950 C-ABI library (hipcc), in-tree: the .so travels to the GPU box with the
snapshot and is git-ignored.  No CMake: a handful of translation units compiled directly.
(The CPU checker has its own recipe under oracle/build.py; the product never builds or loads it.)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo_vo_amd", "csrc")
HOST = os.path.join(ROOT, "stereo_vo_amd", "host")
LIB = os.path.join(ROOT, "stereo_vo_amd", "libsvo_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
             "-I", CSRC, "-I", HOST] + os.environ.get("SVO_EXTRA_HIPFLAGS", "").split()  # developer experiments (-DSVO_EXP_...): never set for a product build


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
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-lz", "-ldl"])
    return LIB


if __name__ == "__main__":
    build_hip("--force" in sys.argv)
