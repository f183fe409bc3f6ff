"""Anchor-stage HIP kernels (k_lookup_exact, k_mask, k_inexact, k_finalize) through the C ABI against the oracle's
getSeeds restatement; bit-exact (pos, unitig, dist, strand) for solid and weak anchors."""
import pytest

from test_sim_seeds import _check

pytestmark = pytest.mark.gpu


def test_gpu_seeds_branching(ds_small):
    assert _check(ds_small, 12, None) > 0


def test_gpu_seeds_clean(ds_clean):
    _check(ds_clean, 10, None)


def test_gpu_seeds_variant_enumeration_agrees(ds_small, ds_tandem, ds_k25, monkeypatch):
    assert _check(ds_tandem, 40, None) > 0
    monkeypatch.setenv("RTK_INEXACT_ENUM", "1")
    assert _check(ds_small, 12, None) > 0
    assert _check(ds_tandem, 40, None) > 0
