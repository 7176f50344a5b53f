"""AddressSanitizer + UndefinedBehaviorSanitizer over the GPU-free host code of the product (the step control
host/lm.cpp, the dense Cholesky, the KITTI-layout decoders, the track rasteriser, the synthetic renderer), on the CPU:
GPU sanitizers are not available on the target pool, and these are the pieces that parse files and index host arrays."""
import os
import struct
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chunk(t, data):
    return struct.pack(">I", len(data)) + t + data + struct.pack(">I", zlib.crc32(t + data) & 0xFFFFFFFF)


def _write_inputs(d):
    from PIL import Image
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    Image.fromarray(g).save(os.path.join(d, "ok_gray.png"))
    Image.fromarray(rng.integers(0, 256, (20, 31, 3)).astype(np.uint8)).save(os.path.join(d, "ok_rgb.png"))
    Image.fromarray((g.astype(np.uint32) * 257).astype(np.uint16)).save(os.path.join(d, "ok_gray16.png"))
    open(os.path.join(d, "ok_p5.pgm"), "wb").write(b"P5\n# c\n53 37\n255\n" + g.tobytes())
    sig = b"\x89PNG\r\n\x1a\n"
    ihdr = struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 0)
    idat = zlib.compress(b"".join(b"\x00" + bytes([10 * r] * 4) for r in range(4)))
    bad = {
        "bad_short_ihdr.png": sig + _chunk(b"IHDR", ihdr[:5]) + _chunk(b"IDAT", idat) + _chunk(b"IEND", b"") + b"\0" * 16,
        "bad_no_ihdr.png": sig + _chunk(b"IDAT", idat) + _chunk(b"IEND", b"") + b"\0" * 32,
        "bad_two_ihdr.png": sig + _chunk(b"IHDR", ihdr) + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", idat) + _chunk(b"IEND", b""),
        "bad_zero.png": sig + _chunk(b"IHDR", struct.pack(">IIBBBBB", 0, 4, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", idat) + _chunk(b"IEND", b""),
        "bad_huge.png": sig + _chunk(b"IHDR", struct.pack(">IIBBBBB", 70000, 4, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", idat) + _chunk(b"IEND", b""),
        "bad_len.png": sig + struct.pack(">I", 0xFFFFFFF0) + b"IHDR" + ihdr + b"\0" * 8,
        "bad_short_idat.png": sig + _chunk(b"IHDR", struct.pack(">IIBBBBB", 64, 64, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", idat) + _chunk(b"IEND", b""),
        "bad_filter.png": sig + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(b"".join(b"\x09" + bytes(4) for _ in range(4)))) + _chunk(b"IEND", b""),
        "bad_truncated.png": (sig + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", idat))[:40],
        "bad_neg.pgm": b"P5\n-1 -1\n255\n" + b"\0" * 4,
        "bad_overflow.pgm": b"P5\n99999999999999999999 2\n255\n" + b"\0" * 4,
        "bad_zero.pgm": b"P5\n0 7\n255\n",
        "bad_truncated.pgm": b"P5\n8 8\n255\n" + b"\0" * 10,
        "bad_empty.png": b"",
    }
    for name, blob in bad.items():
        open(os.path.join(d, name), "wb").write(blob)
    rows = ["%e " * 11 % tuple(range(11)) + "1.0" for _ in range(3)] + ["1 2 3"]
    open(os.path.join(d, "poses.txt"), "w").write("\n".join(rows) + "\n")


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    host = os.path.join(ROOT, "stereo_vo_amd", "host")
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(ROOT, "tests", "sanitize", "host_sanitize.cpp")] + \
           [os.path.join(host, f) for f in ("lm.cpp", "linalg.cpp", "kitti_io.cpp", "draw.cpp", "synth.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", host, "-I", os.path.join(ROOT, "stereo_vo_amd", "csrc")] + \
          srcs + ["-o", exe, "-lz", "-lpthread"]
    subprocess.run(cmd, check=True, timeout=600)
    d = tmp_path / "inputs"
    d.mkdir()
    _write_inputs(str(d))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "host sanitize ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_xcd_aware_block_map_serves_every_slot_exactly_once(tmp_path):
    """The item -> workgroup map of the grouped tracking / sparse-stereo launches (csrc/xcd_map.h): the device function's own text
    compiled for the host walks whole grids — every (lane, item) slot once, max(n, 1) workgroups per lane (what the lanes'
    arrival targets count), contiguous runs per XCD label — under ASan + UBSan."""
    exe = str(tmp_path / "xcd_map_test")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "stereo_vo_amd", "csrc"), os.path.join(ROOT, "tests", "sanitize", "xcd_map_test.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "xcd map ok" in r.stdout
