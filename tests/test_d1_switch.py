"""Canonical rule [D1] as a switch. `chooseColors` (src/Correction.cpp:215-429) visits the anchors' colour sets in ascending order of cardinality, but
the list it sorts comes out of an `unordered_map` keyed by POINTERS (src/Correction.cpp:286) and `std::sort` is not stable (:293): the order of sets of
equal cardinality depends on heap addresses in the reference. This build orders such ties by unitig id; both directions are kept alive, in the oracle
(oracle_correct.cpp, RTK_D1_ORDER) and on the device (rtk_colours.h / rtk_region.h, rtk_opts::d1_desc):
  asc   ties by ascending unitig id   (RTK_D1_ORDER unset or =asc)
  desc  ties by descending unitig id  (RTK_D1_ORDER=desc)
Device == oracle under each; the number of reads the rule decides is printed (profiles/r04_d1_count.json holds it for configs[1] and the diploid set)."""
import pytest

from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _corrected(prefix, n, lib_path):
    fa, rt = prefix + ".index.k31.fasta.gz", prefix + ".index.k31.rtsk"
    og, pg = op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0, lib_path=lib_path)
    reads = op.read_fastq(prefix + ".lr.fq")[:n]
    seqs, quals = [r[1] for r in reads], [r[2] for r in reads]
    got = pg.correct_batch(seqs, quals)
    want, _ = og.correct_batch(seqs, quals)
    assert got == want
    return got


def _both(prefix, n, lib_path, monkeypatch):
    monkeypatch.delenv("RTK_D1_ORDER", raising=False)
    a = _corrected(prefix, n, lib_path)
    monkeypatch.setenv("RTK_D1_ORDER", "desc")
    d = _corrected(prefix, n, lib_path)
    monkeypatch.delenv("RTK_D1_ORDER", raising=False)
    return sum(1 for x, y in zip(a, d) if x != y), len(a)


def test_sim_both_orders_of_d1(ds_small, ds_tandem, ds_snps_rich, monkeypatch):
    _both(ds_small, 8, SIM_LIB, monkeypatch)
    _both(ds_tandem, 6, SIM_LIB, monkeypatch)
    _both(ds_snps_rich, 6, SIM_LIB, monkeypatch)


@pytest.mark.gpu
def test_gpu_both_orders_of_d1(ds_small, ds_tandem, ds_medium, ds_snps, monkeypatch):
    _both(ds_small, 12, None, monkeypatch)
    _both(ds_tandem, 40, None, monkeypatch)
    _both(ds_snps, 20, None, monkeypatch)
    d, n = _both(ds_medium, 60, None, monkeypatch)
    print("reads of ds_medium that differ between the two orders of [D1]: %d of %d" % (d, n))
