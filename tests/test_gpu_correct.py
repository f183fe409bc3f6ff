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
