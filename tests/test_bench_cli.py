"""bench.py's launcher contract: `--gpus N` without a launcher starts N ranks itself and must REFUSE (non-zero exit, clear
message) when fewer GPUs are visible — never print a line measured on fewer ranks than asked."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    want = torch.cuda.device_count() + 1
    if want < 2:
        want = 2
    r = _run(["--gpus", str(want), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and '"metric"' not in r.stdout


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE" in r.stderr and '"metric"' not in r.stdout
