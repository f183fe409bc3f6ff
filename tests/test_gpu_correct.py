"""End-to-end parity of the HIP path (all kernels, through the C ABI) with the oracle: corrected FASTQ records must be
byte-identical (sequence and quality) on seeded synthetic inputs, including the edge cases the reference handles."""
import pytest

from oracle import oracle_py as op
from test_sim_correct import _check

pytestmark = pytest.mark.gpu


def test_gpu_correct_branching(ds_small):
    s0 = op.read_fastq(ds_small + ".lr.fq")[0][1]
    extra = ["ACGT" * 5, "A" * 31, "N" * 200, s0[:500].lower(), s0[:400] + "N" * 40 + s0[440:1200]]
    st, got, seqs = _check(ds_small, 12, None, extra)
    assert st["n_expand"] > 0 and st["ms_correct"] > 0


def test_gpu_correct_clean(ds_clean):
    _check(ds_clean, 10, None)


def test_gpu_scratch_overflow_is_redone_on_device(ds_small, monkeypatch):
    monkeypatch.setenv("RTK_TEST_TINY_SCRATCH", "1")
    st, got, seqs = _check(ds_small, 12, None, counters_must_match=False)
    assert st["n_arena_overflow"] > 0


def test_gpu_correct_other_options(ds_small):
    _check(ds_small, 12, None, opts=dict(insert_sz=300, max_len_weak_region1=300, max_qual=30))


def test_gpu_correct_k25(ds_k25):
    s0 = op.read_fastq(ds_k25 + ".lr.fq")[0][1]
    extra = [s0[:300] + "R" + s0[301:700] + "YN" + s0[702:1500], s0[:25], s0[:26]]
    _check(ds_k25, 6, None, extra, k=25)


def test_gpu_correct_volume(ds_medium):
    """1.3 Mb of ONT-profile reads on a repeat-bearing diploid graph, oracle on all host threads."""
    import os
    st, got, seqs = _check(ds_medium, 160, None, threads=os.cpu_count() or 4)
    assert st["n_regions"] > 5000
