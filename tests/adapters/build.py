"""Build recipe of the adapter demo (tests only): g++ compiles adapters/*.cpp + tests/adapters/vo_node_calls.cpp against
the syntax-only stub headers in tests/stubs/ (this image has neither OpenCV nor Eigen) and links libsvo_hip.so.
A maintainer with the real libraries drops `-I tests/stubs` and adds `pkg-config --cflags opencv eigen3` instead."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT_DIR = os.path.join(ROOT, "tests", "adapters", "_build")
EXE = os.path.join(OUT_DIR, "vo_node_calls")


def build_adapter_demo(force=False):
    lib = os.path.join(ROOT, "stereo_vo_amd", "libsvo_hip.so")
    srcs = sorted(glob.glob(os.path.join(ROOT, "adapters", "*.cpp"))) + [os.path.join(ROOT, "tests", "adapters", "vo_node_calls.cpp")]
    deps = srcs + glob.glob(os.path.join(ROOT, "adapters", "*.hpp")) + glob.glob(os.path.join(ROOT, "tests", "stubs", "*", "*")) + \
        [lib, os.path.join(ROOT, "include", "svo.h"), os.path.join(ROOT, "stereo_vo_amd", "host", "stereo_vo.hpp")]
    if not force and os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return EXE
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Wno-unused-parameter",
           "-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "adapters"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "stereo_vo_amd", "host")] + srcs + \
          ["-o", EXE, lib, "-Wl,-rpath," + os.path.join(ROOT, "stereo_vo_amd"), "-Wl,-rpath,$ORIGIN/../../../stereo_vo_amd", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined", "-lpthread"]
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return EXE


if __name__ == "__main__":
    print(build_adapter_demo(force=True))
