"""CPU: the C-ABI library loads without a GPU and exports every symbol include/svo.h declares."""
import os
import re

import stereo_vo_amd as S
from stereo_vo_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "svo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svo_[a-z0-9_]+)\s*\(", txt)) - {"svo_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    L = S.lib()
    declared = _declared_symbols()
    assert len(declared) >= 35
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"declared in svo.h but not exported: {missing}"
    assert set(declared) == set(api.SYMBOLS), sorted(set(declared) ^ set(api.SYMBOLS))
    assert b"gfx950" in L.svo_version()


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import ctypes as C
    n = C.c_int(0)
    # svo_create must either succeed (GPU box) or return an error code; never a CPU fallback object
    lim = api.Limits(64, 64, 1, 16, 1024, 16)
    h = C.c_void_p()
    rc = S.lib().svo_create(C.byref(h), 0, C.byref(lim))
    if rc == 0:
        S.lib().svo_destroy(h)
    else:
        assert rc in (-2, -4) and not h.value


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "stereo_vo_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                # comments may cite the oracle files as the arithmetic spec; code may not include/load/call them
                if re.search(r'#include\s*"svo_oracle|import\s+oracle_lib|libsvo_oracle|\bora_[a-z_]+\s*\(', t):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
