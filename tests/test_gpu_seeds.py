"""Anchor-stage HIP kernels (k_lookup_exact, k_mask, k_inexact, k_finalize) through the C ABI against the oracle's
getSeeds restatement; bit-exact (pos, unitig, dist, strand) for solid and weak anchors."""
import pytest

from conftest import make_dataset
from test_sim_seeds import _check, _check_gap_reads

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


def test_gpu_seeds_mask_in_segments(ds_small, ds_clean, tmp_path, monkeypatch):
    """Long reads are masked by several waves, one per segment of windows (8192 by default, 4096 = one round of the wave here): reads of 20-60 kb
    through a graph with repeats, so that gaps of every kind (closed, left open by the colour test, at the read's head and tail) lie across
    segment borders; and the 3-5 kb reads of the small sets in two segments each."""
    pre = make_dataset(tmp_path, "long", ["--seed", 91, "--ref-len", 200000, "--het", 0.003, "--repeat-frac", 0.1, "--sr-cov", 30, "--sr-err", 0.01,
                                          "--lr-n", 8, "--lr-len", 30000, "--lr-profile", "ont", "--lr-err", 0.10], ["--global-cov-factor", 1.2])
    assert _check(pre, 8, None) > 0
    monkeypatch.setenv("RTK_MASK_SEG", "4096")
    assert _check(pre, 8, None) > 0
    assert _check(ds_small, 12, None) > 0
    _check(ds_clean, 10, None)


def test_gpu_seeds_mask_gaps_across_segment_borders(ds_clean, ds_small, monkeypatch):
    """the constructed reads of the simulator test, six times as long (45 kb: a dozen segments of 4096 windows, five of 8192)"""
    _check_gap_reads((ds_clean, ds_small), 6, ("4096", "8192"), None, monkeypatch)
