"""Lane-per-region kernel (csrc/hip/rtk_region_lane.h, k_regions_lanes; RTK_LANE_MAX_GAP opts in): the gaps of its class are corrected one region per lane,
whatever the lane program does not hold is handed on to the wave kernel, and the corrected reads -- sequence AND quality strings -- are byte-identical to the
oracle's either way. A lane's program has no cross-lane operation, so the 1-lane simulator runs exactly what a lane runs on the device; the gpu tests run the
same program 64 regions per wavefront, beside the wave kernel (second stream, shared queue) and one after the other. The tests also hold the lane route to a
minimum share of its class: a lane kernel that handed everything on would otherwise pass on the wave kernel's results."""
import pytest

from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _run(prefix, n, lib, k=31, opts=None):
    fa, rt = prefix + ".index.k%d.fasta.gz" % k, prefix + ".index.k%d.rtsk" % k
    pg = api.Graph(fa, rt, k, device=0, lib_path=lib)
    reads = op.read_fastq(prefix + ".lr.fq")[:n]
    seqs, quals = [r[1] for r in reads], [r[2] for r in reads]
    b = api.Batch(pg, seqs, quals)
    b.run(pg.opts(**(opts or {})))
    return b.fetch(), b.stats(), seqs, quals


def _oracle(prefix, seqs, quals, k=31, opts=None):
    og = op.Graph(prefix + ".index.k%d.fasta.gz" % k, prefix + ".index.k%d.rtsk" % k, k)
    return og.correct_batch(seqs, quals, opts=og.opts(**(opts or {})), threads=4)[0]


def _check(prefix, n, lib, monkeypatch, k=31, max_gap="256", min_lane_share=0.9, opts=None):
    monkeypatch.setenv("RTK_LANE_MAX_GAP", "0")
    base, st0, seqs, quals = _run(prefix, n, lib, k, opts)
    assert st0["n_lane_regions"] == 0
    want = _oracle(prefix, seqs, quals, k, opts)
    assert base == want
    monkeypatch.setenv("RTK_LANE_MAX_GAP", max_gap)
    got, st, _, _ = _run(prefix, n, lib, k, opts)
    assert got == want, "lane kernel on: %d reads differ from the oracle" % sum(1 for a, b in zip(got, want) if a != b)
    assert st["n_lane_regions"] > 0 and st["n_regions"] == st0["n_regions"]
    done = st["n_lane_regions"] - st["n_lane_handed"]
    assert done >= min_lane_share * st["n_lane_regions"], (st["n_lane_regions"], st["n_lane_handed"])
    return st


def test_sim_lanes_branching(ds_small, monkeypatch):
    st = _check(ds_small, 12, SIM_LIB, monkeypatch)
    assert st["n_lane_regions"] > 100


def test_sim_lanes_clean_and_options(ds_clean, ds_small, monkeypatch):
    _check(ds_clean, 8, SIM_LIB, monkeypatch, min_lane_share=0.5)  # (long unitigs: a handful of regions in the class)
    _check(ds_small, 8, SIM_LIB, monkeypatch, opts=dict(insert_sz=300, max_len_weak_region1=300, max_qual=30))
    _check(ds_small, 8, SIM_LIB, monkeypatch, max_gap="64", min_lane_share=0.7)


def test_sim_lanes_snp_annotations_cycles_and_other_k(ds_snps_rich, ds_snps, ds_k25, ds_k21, monkeypatch):
    """SNP annotations (getAmbiguityVector / fixAmbiguity lane by lane), short cycles (handed on: fixRepeats is the wave kernel's), k = 25 and k = 21."""
    st = _check(ds_snps_rich, 16, SIM_LIB, monkeypatch, min_lane_share=0.8)
    assert st["n_lane_handed"] > 0  # the tandem repeats of this set
    _check(ds_snps, 24, SIM_LIB, monkeypatch)
    _check(ds_snps, 8, SIM_LIB, monkeypatch, opts=dict(min_confidence_snp_corr=0.5, out_qual=3, max_qual=30))
    _check(ds_k25, 6, SIM_LIB, monkeypatch, k=25)
    _check(ds_k21, 8, SIM_LIB, monkeypatch, k=21)


def test_sim_lanes_tiny_capacities_are_handed_on(ds_small, ds_snps, monkeypatch):
    """Capacities far too small (test hook): a large part of the class is handed on to the wave kernel, whose own work areas overflow too and are redone:
    the results do not change."""
    monkeypatch.setenv("RTK_TEST_TINY_SCRATCH", "1")
    monkeypatch.setenv("RTK_LANE_MAX_GAP", "256")
    for prefix, n in ((ds_small, 12), (ds_snps, 10)):
        got, st, seqs, quals = _run(prefix, n, SIM_LIB)
        assert got == _oracle(prefix, seqs, quals)
        assert st["n_lane_regions"] > 0 and st["n_lane_handed"] > st["n_lane_regions"] // 4


@pytest.mark.gpu
def test_gpu_lanes_beside_the_wave_kernel(ds_medium, ds_snps_rich, monkeypatch):
    st = _check(ds_medium, 160, None, monkeypatch)
    assert st["n_lane_regions"] > 2000
    _check(ds_medium, 160, None, monkeypatch, max_gap="128")
    _check(ds_snps_rich, 80, None, monkeypatch, min_lane_share=0.8)


@pytest.mark.gpu
def test_gpu_lanes_serial_small_rounds_and_tiny_capacities(ds_medium, ds_snps, monkeypatch):
    monkeypatch.setenv("RTK_LANE_SERIAL", "1")
    _check(ds_medium, 80, None, monkeypatch)
    monkeypatch.setenv("RTK_LANE_ROUND", "16")
    monkeypatch.setenv("RTK_LANE_WAVES", "64")  # few persistent waves: several rounds each
    _check(ds_medium, 80, None, monkeypatch)
    monkeypatch.delenv("RTK_LANE_SERIAL"); monkeypatch.delenv("RTK_LANE_ROUND")
    monkeypatch.setenv("RTK_TEST_TINY_SCRATCH", "1")
    monkeypatch.setenv("RTK_LANE_MAX_GAP", "256")
    got, st, seqs, quals = _run(ds_snps, 20, None)
    assert got == _oracle(ds_snps, seqs, quals)
    assert st["n_lane_handed"] > st["n_lane_regions"] // 4


@pytest.mark.gpu
def test_gpu_stolen_regions_that_overflow_are_redone(ds_medium, ds_snps, monkeypatch):
    """beside the wave kernel with ONE slow lane wave (rounds of 4), the wave kernel steals nearly the whole lane class once its own lists are drained; with
    tiny work areas (test hook) the stolen regions overflow, and the redo launches -- which walk rorder and horder, not the lane class's lists -- must still find them
    (a stolen region that overflows joins horder). Before the fix such a region was never redone and its read came out with a stale segment, without an error."""
    monkeypatch.setenv("RTK_LANE_MAX_GAP", "256")
    monkeypatch.setenv("RTK_LANE_WAVES", "1"); monkeypatch.setenv("RTK_LANE_ROUND", "4")
    monkeypatch.setenv("RTK_TEST_TINY_SCRATCH", "1")
    for prefix, n in ((ds_medium, 60), (ds_snps, 20)):
        got, st, seqs, quals = _run(prefix, n, None)
        want = _oracle(prefix, seqs, quals)
        assert got == want, "%d reads differ from the oracle" % sum(1 for a, b in zip(got, want) if a != b)
        assert st["n_arena_overflow"] > 0


@pytest.mark.gpu
def test_gpu_big_tickets_take_the_lane_kernel_by_themselves(ds_medium, monkeypatch):
    """RTK_LANE_AUTO_BASES (default 512 Mi bases): a ticket that large runs the lane kernel for gaps under 128 without RTK_LANE_MAX_GAP, one kernel after the other; the threshold is
    lowered here so that a 0.6 Mb ticket crosses it. Same bytes as the oracle; a ticket below the threshold does not touch the lane kernel."""
    monkeypatch.delenv("RTK_LANE_MAX_GAP", raising=False)
    got0, st0, seqs, quals = _run(ds_medium, 80, None)
    assert st0["n_lane_regions"] == 0
    monkeypatch.setenv("RTK_LANE_AUTO_BASES", "1000")
    # (the threshold is read once per process: a fresh process)
    import subprocess, sys, json, os
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_lanes_regions import _run\n"
            "got, st, seqs, quals = _run(%r, 80, None)\n"
            "print(json.dumps({'got': got, 'lane': st['n_lane_regions'], 'handed': st['n_lane_handed']}))") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ds_medium)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, RTK_LANE_AUTO_BASES="1000"), timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert [tuple(x) for x in d["got"]] == got0 == _oracle(ds_medium, seqs, quals)
    assert d["lane"] > 500 and d["handed"] < d["lane"] // 4
