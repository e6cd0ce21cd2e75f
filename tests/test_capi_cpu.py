"""The C-ABI library without a GPU: it loads, exports every symbol include/*.h declares, validates
arguments, and refuses to compute (no CPU fallback)."""
import ctypes as C
import glob
import os
import re
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, RADIUS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ma_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    L = ma.lib()
    syms = _declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_version_and_device_count():
    assert b"gfx950" in ma.lib().ma_version()
    assert ma.device_count() >= 0


def test_argument_validation_precedes_the_device():
    L = ma.lib()
    assert L.ma_zgesv(-1, None, None, None) == ma.MA_ERR_DIM
    assert L.ma_zgesv(0, None, None, None) == ma.MA_OK            # empty system: nothing to do
    assert L.ma_zgesv(3, None, None, None) == ma.MA_ERR_INVALID
    h = C.c_void_p()
    assert L.ma_lu_plan_create(0, 0, C.byref(h)) == ma.MA_ERR_DIM
    assert L.ma_bem_plan_create(None, 0, C.byref(h)) == ma.MA_ERR_INVALID
    assert b"NULL" in L.ma_last_error_string()
    om = O.icosphere(RADIUS, 0)
    om.bc_len[1] = 7                                               # more boundary values than a panel has slots
    with pytest.raises(ma.MaError) as e:
        ma.BemPlan(to_ma_mesh(om))
    assert e.value.status == ma.MA_ERR_INVALID
    om = O.icosphere(RADIUS, 0)
    om.conn[0, 0] = 999
    with pytest.raises(ma.MaError) as e:
        ma.BemPlan(to_ma_mesh(om))
    assert e.value.status == ma.MA_ERR_INVALID


def test_no_cpu_fallback():
    if ma.device_count() > 0:
        pytest.skip("a GPU is present")
    om = O.icosphere(RADIUS, 0)
    with pytest.raises(ma.MaError) as e:
        ma.assemble_tbem(to_ma_mesh(om), 10.0, 0.4j)
    assert e.value.status == ma.MA_ERR_NO_DEVICE
    with pytest.raises(ma.MaError) as e:
        ma.zgesv(np.eye(2), np.ones(2))
    assert e.value.status == ma.MA_ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """Nothing under math_audio_amd/ may include, import or link oracle/ (tests and bench's cpu_baseline only)."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "math_audio_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                t = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"oracle_lib|libma_oracle|ma_oracle\.h|mao_[a-z]", t):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
