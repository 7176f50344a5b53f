import os
import sys

import pytest

# the HIP runtime reads this when it initialises: the GPU tests run with the hardware-queue count bench.py measures with
# (two pipeline groups = 2 x 7 HIP streams; with the default 4 their launches share queues — slower, same results)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

# torch bundles its own libamdhip64; load it BEFORE libsvo_hip.so so that the process holds exactly one HIP
# runtime (same soname => the loader reuses the first one).  Tests that hand torch tensors / RCCL buffers to
# the library need that; the library itself does not depend on torch.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One svo_ctx for the whole GPU session (fails loudly if the HIP library or the GPU is missing)."""
    import stereo_vo_amd as S
    c = S.Context(1280, 720, max_batch=4, max_corners=4096, max_candidates=1 << 17, max_features=4096)
    yield c
    c.close()


@pytest.fixture(scope="session")
def frames():
    """A few small synthetic stereo frames shared by the parity tests (rendered once)."""
    import stereo_vo_amd as S
    from stereo_vo_amd import api
    p = api.synth_default(496, 160)
    p.focal = 300.0
    out = [S.synth_render(p, i) for i in range(4)]
    return p, out
