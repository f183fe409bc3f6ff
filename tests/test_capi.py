"""The C-ABI library loads and exports every symbol include/ratatosk_hip.h declares (no compute without a GPU),
and refuses to compute when no device is present (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, SIM_LIB
from ratatosk_amd import api


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ratatosk_hip.h")).read()
    return sorted(set(re.findall(r"\b(rtk_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported_by_product_library():
    L = ctypes.CDLL(api.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libratatosk_hip.so does not export %s" % s


def test_simulator_exports_same_abi():
    L = ctypes.CDLL(SIM_LIB)
    for s in declared_symbols():
        assert hasattr(L, s)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.RtkError) as e:
        api.myers_batch(["ACGT"], ["ACGT"])
    assert "no HIP device" in str(e.value) or "rtk error" in str(e.value)
