"""End-to-end parity of the HIP path (all kernels, through the C ABI) with the oracle: corrected FASTQ records must be
byte-identical (sequence and quality) on seeded synthetic inputs, including the edge cases the reference handles."""
import os

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
    st, got, seqs = _check(ds_medium, 160, None, threads=os.cpu_count() or 4)
    assert st["n_regions"] > 5000


def test_gpu_overlapped_stages_give_the_same_records(ds_medium):
    """Three batches through api.run_pipelined (seed stage of batch i+1 beside the region stage of batch i, one stream each, shared
    cached scratch slots): every batch must still equal the oracle."""
    from ratatosk_amd import api
    fa, rt = ds_medium + ".index.k31.fasta.gz", ds_medium + ".index.k31.rtsk"
    og, pg = op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0)
    reads = op.read_fastq(ds_medium + ".lr.fq")
    parts = [reads[0:60], reads[60:110], reads[110:160]]
    batches = [api.Batch(pg, [r[1] for r in p], [r[2] for r in p]) for p in parts]
    api.run_pipelined(batches + batches)  # every batch twice: re-running a resident batch is what the bench does
    for p, b in zip(parts, batches):
        want, _ = og.correct_batch([r[1] for r in p], [r[2] for r in p], threads=os.cpu_count() or 4)
        assert b.fetch() == want


def test_gpu_small_tickets_of_many_callers(ds_medium):
    """The reference's way of using the seam: many worker threads, each with a ticket of about a megabase of its own (src/Ratatosk.cpp:727-772). Twelve callers, 36 tickets of
    ~110 kb through rtk_batch_create / run / fetch at the same time (the stages of different tickets overlap on the device, the graph-wide work areas go from ticket to
    ticket behind their locks): every ticket equals the oracle."""
    import threading
    from ratatosk_amd import api
    fa, rt = ds_medium + ".index.k31.fasta.gz", ds_medium + ".index.k31.rtsk"
    og, pg = op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0)
    reads = op.read_fastq(ds_medium + ".lr.fq")[:156]
    parts = [reads[i:i + 13] for i in range(0, 156, 13)] * 3
    out, err, nxt, lock = [None] * len(parts), [], [0], threading.Lock()

    def caller():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(parts):
                return
            try:
                b = api.Batch(pg, [r[1] for r in parts[i]], [r[2] for r in parts[i]]); b.run(pg.opts()); out[i] = b.fetch(); b.close()
            except Exception as e:  # noqa: BLE001
                err.append(repr(e))

    th = [threading.Thread(target=caller) for _ in range(12)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not err, err[:2]
    want, _ = og.correct_batch([r[1] for r in reads], [r[2] for r in reads], threads=os.cpu_count() or 4)
    for i, p_ in enumerate(parts):
        j = (i % 12) * 13
        assert out[i] == want[j:j + len(p_)], "ticket %d differs" % i


def test_gpu_correct_short_cycles(ds_tandem):
    _check(ds_tandem, 40, None)


def test_gpu_correct_snp_annotations(ds_snps):
    """SNP-annotated index: getAmbiguityVector / fixAmbiguity on the GPU equal the oracle's."""
    _check(ds_snps, 40, None)
    _check(ds_snps, 12, None, opts=dict(min_confidence_snp_corr=0.5, out_qual=3, max_qual=30))


def test_gpu_snp_annotations_with_repeats_cycles_and_tiny_scratch(ds_snps_rich, ds_snps, monkeypatch):
    _check(ds_snps_rich, 80, None, threads=os.cpu_count() or 4)
    monkeypatch.setenv("RTK_TEST_TINY_SCRATCH", "1")
    st, _, _ = _check(ds_snps, 20, None, counters_must_match=False)
    assert st["n_arena_overflow"] > 0


def test_gpu_correct_k21(ds_k21):
    _check(ds_k21, 16, None, k=21)
