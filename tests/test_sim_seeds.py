"""Anchor stage (mask, 1-edit k-mer probes, overlap filter, keep_non_overlap, adjacency check) of the device programs,
executed by the host simulator, against the oracle's getSeeds restatement. Exact equality of both anchor lists."""
from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _check(prefix, n, lib_path):
    fa, rt = prefix + ".index.k31.fasta.gz", prefix + ".index.k31.rtsk"
    og, pg = op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0, lib_path=lib_path)
    reads = op.read_fastq(prefix + ".lr.fq")
    tot_w = 0
    for name, s, q in reads[:n]:
        a, b = pg.seeds(s), og.seeds(s)
        assert a == b, name
        tot_w += len(a[1])
    # edge cases: read of length <= k has no seeds at all (src/Graph.cpp:49); all-N read; read with N inside
    s0 = reads[0][1]
    for s in ["ACGT" * 7 + "ACG", "N" * 100, s0[:300] + "N" + s0[301:900]]:
        assert pg.seeds(s) == og.seeds(s)
    return tot_w


def test_sim_seeds_branching(ds_small):
    assert _check(ds_small, 12, SIM_LIB) > 0  # weak anchors must be exercised


def test_sim_seeds_clean(ds_clean):
    _check(ds_clean, 6, SIM_LIB)


def test_sim_seeds_variant_enumeration_agrees(ds_small, ds_tandem, monkeypatch):
    """The 1-edit search runs on half-k-mer seeds + verification; spelling and probing every variant (RTK_INEXACT_ENUM=1, the first
    implementation) must give the same anchors - also where h-mers repeat (tandem repeats: more than four hits per window)."""
    assert _check(ds_tandem, 10, SIM_LIB) > 0
    monkeypatch.setenv("RTK_INEXACT_ENUM", "1")
    assert _check(ds_small, 6, SIM_LIB) > 0
    assert _check(ds_tandem, 10, SIM_LIB) > 0


def test_sim_seeds_mask_in_segments(ds_small, ds_clean, monkeypatch):
    """k_mask cuts a read into segments of RTK_MASK_SEG windows, one wave each (8192 in production: the launch lasted as long as its longest read).
    With segments of 64 and 192 windows every read of these sets is masked in dozens of pieces: gaps that reach back over a segment border, the read's
    first hit and the last hit before a segment read off the presence bits, head and tail rules applied once. Same anchors as the oracle."""
    for seg in ("64", "192"):
        monkeypatch.setenv("RTK_MASK_SEG", seg)
        assert _check(ds_small, 12, SIM_LIB) > 0
        _check(ds_clean, 6, SIM_LIB)
