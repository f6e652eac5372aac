"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/pumipic_hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi(pp):
    from pumipic_amd import capi as c
    c.build()
    return c


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pumipic_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.lib()
    names = _declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    # and the ctypes table binds exactly the declared set
    assert sorted(capi.SYMBOLS) == names


def test_no_cpu_fallback_without_gpu(capi):
    if capi.lib().pp_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.PPError):
        capi.init(0)
    assert b"no CPU fallback" in capi.lib().pp_last_error()
    # handle constructors fail loudly as well
    with pytest.raises(capi.PPError):
        capi.PS.scs(capi.PARTICLE_PUSH, 1, np.array([0], dtype=np.int32))
    with pytest.raises(capi.PPError):
        capi.Mesh(2, np.zeros((3, 2)), np.array([[0, 1, 2]]))


def test_product_never_touches_the_oracle():
    """the product package may not import, link or reference anything under oracle/"""
    pkg = os.path.join(ROOT, "pumi-pic_amd")
    for dp, _, files in os.walk(pkg):
        if "build" in dp.split(os.sep) or "__pycache__" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"\bppo(_|\b)", src), (dp, f)
                assert "oracle/" not in src and "load_oracle" not in src, (dp, f)
