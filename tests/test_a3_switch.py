"""Bifrost assumption [A3] as a switch: the order in which `getSuccessors()` hands out the neighbours of a unitig end. `exploreSubGraph`
(src/GraphTraversal.cpp:456-587) pushes them in that order, so it decides between candidate paths of equal score. Two readings are kept alive, in the
oracle (oracle/oracle_graph.cpp, Graph::successors) and on the device (rtk_explore_subgraph, rtk_opts::a3_strand_order):
  walk    by the base appended in walk direction, A,C,G,T, on both strands                                   (RTK_A3_ORDER unset or =walk)
  strand  on the reverse strand by the base as the unitig's own strand spells it (T,G,C,A in walk direction)   (RTK_A3_ORDER=strand)
Device == oracle under each; the number of reads the reading decides is printed (DESIGN_HISTORY.md section 4 quotes it for configs[1])."""
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
    monkeypatch.delenv("RTK_A3_ORDER", raising=False)
    w = _corrected(prefix, n, lib_path)
    monkeypatch.setenv("RTK_A3_ORDER", "strand")
    s = _corrected(prefix, n, lib_path)
    monkeypatch.delenv("RTK_A3_ORDER", raising=False)
    return sum(1 for a, b in zip(w, s) if a != b), len(w)


def test_sim_both_readings_of_a3(ds_small, ds_tandem, monkeypatch):
    _both(ds_small, 8, SIM_LIB, monkeypatch)
    _both(ds_tandem, 6, SIM_LIB, monkeypatch)


@pytest.mark.gpu
def test_gpu_both_readings_of_a3(ds_small, ds_tandem, ds_medium, ds_snps, monkeypatch):
    _both(ds_small, 12, None, monkeypatch)
    _both(ds_tandem, 40, None, monkeypatch)
    _both(ds_snps, 20, None, monkeypatch)
    d, n = _both(ds_medium, 60, None, monkeypatch)
    print("reads of ds_medium that differ between the two readings of [A3]: %d of %d" % (d, n))
