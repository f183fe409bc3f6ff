"""Bifrost assumption [A2] as a switch (reference: `dbg.searchSequence(l_s, false, true, true, true, /*or_exclusive_match*/ true)`,
src/Graph.cpp:193; Bifrost itself is not in the reference tree). Three readings are kept alive, in the oracle and on the device:
  exclusive      substitution -> insertion -> deletion, and a window one kind has matched is not searched with the next kind (RTK_A2_XOR unset or
                 =exclusive: the default since round 4, it is what the flag's name asks for)
  exclusive-ids  the same with the kinds in the order of the function's parameters: insertion -> deletion -> substitution
  union          every graph k-mer one substitution / insertion / deletion away from a window is reported (the default of rounds 1-3)
All must give identical anchors and corrected reads on the oracle and the device; the readings themselves must differ where a window has
hits of two kinds (otherwise the switch would test nothing). oracle/oracle_graph.hpp lists the assumptions."""
import pytest

from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _anchors_and_reads(prefix, n, lib_path):
    fa, rt = prefix + ".index.k31.fasta.gz", prefix + ".index.k31.rtsk"
    og, pg = op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0, lib_path=lib_path)
    reads = op.read_fastq(prefix + ".lr.fq")[:n]
    weak = []
    for name, s, q in reads:
        a, b = pg.seeds(s), og.seeds(s)
        assert a == b, name
        weak.append(a[1])
        assert og.inexact(s) is not None
    seqs, quals = [r[1] for r in reads], [r[2] for r in reads]
    got = pg.correct_batch(seqs, quals)
    want, _ = og.correct_batch(seqs, quals)
    assert got == want
    return weak, got, [og.inexact(s)[1] for s in seqs]


def _both(prefix, n, lib_path, monkeypatch, third=False):
    monkeypatch.setenv("RTK_A2_XOR", "union")
    u = _anchors_and_reads(prefix, n, lib_path)
    monkeypatch.delenv("RTK_A2_XOR", raising=False)  # the default: exclusive
    x = _anchors_and_reads(prefix, n, lib_path)
    if third:
        monkeypatch.setenv("RTK_A2_XOR", "exclusive")
        assert _anchors_and_reads(prefix, n, lib_path) == x  # the default by its name
        monkeypatch.setenv("RTK_A2_XOR", "exclusive-ids")
        y = _anchors_and_reads(prefix, n, lib_path)
        monkeypatch.delenv("RTK_A2_XOR", raising=False)
        return u, x, y
    monkeypatch.delenv("RTK_A2_XOR", raising=False)
    return u, x


def test_sim_both_readings_of_a2(ds_small, ds_tandem, monkeypatch):
    (uw, ur, ui), (xw, xr, xi), (yw, yr, yi) = _both(ds_small, 8, SIM_LIB, monkeypatch, third=True)
    # the raw 1-edit hit lists: an exclusive reading is a subset of the union, and a strict one somewhere (windows with hits of two kinds exist);
    # the two orders of the kinds keep different hits somewhere
    n_u = sum(len(v) for v in ui); n_x = sum(len(v) for v in xi); n_y = sum(len(v) for v in yi)
    assert n_x < n_u and n_y < n_u, (n_u, n_x, n_y)
    for a, b, c in zip(ui, xi, yi):
        assert set(map(tuple, b)) <= set(map(tuple, a)) and set(map(tuple, c)) <= set(map(tuple, a))
    assert any(set(map(tuple, b)) != set(map(tuple, c)) for b, c in zip(xi, yi))
    _both(ds_tandem, 6, SIM_LIB, monkeypatch, third=True)


def test_sim_exclusive_reading_with_variant_enumeration(ds_small, monkeypatch):
    """the fallback 1-edit search (every variant spelled and probed) implements the switch as well"""
    monkeypatch.setenv("RTK_INEXACT_ENUM", "1")
    _both(ds_small, 4, SIM_LIB, monkeypatch)


@pytest.mark.gpu
def test_gpu_both_readings_of_a2(ds_small, ds_tandem, ds_medium, monkeypatch):
    (uw, ur, ui), (xw, xr, xi), (yw, yr, yi) = _both(ds_small, 12, None, monkeypatch, third=True)
    assert sum(len(v) for v in xi) < sum(len(v) for v in ui) and sum(len(v) for v in yi) < sum(len(v) for v in ui)
    _both(ds_tandem, 40, None, monkeypatch, third=True)
    (_, ur, _), (_, xr, _) = _both(ds_medium, 60, None, monkeypatch)
    print("reads of ds_medium that differ between union and exclusive: %d of %d" % (sum(1 for a, b in zip(ur, xr) if a != b), len(ur)))
    monkeypatch.setenv("RTK_INEXACT_ENUM", "1")
    _both(ds_small, 12, None, monkeypatch)
