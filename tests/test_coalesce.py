"""Ticket coalescing inside rtk_correct_batch (include/ratatosk_hip.h, revision 6): the reference's worker model -- `-c` threads, each calling the
per-read loop on a ticket of its own (src/Ratatosk.cpp:727-772, src/Common.hpp:138) -- through the one function SURVEY.md 8(b) spells out. Concurrent
calls are merged into one launch; every caller must still get exactly its own reads, equal to the oracle's, whatever it was merged with."""
import ctypes as C
import os
import threading

import pytest

from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _run_callers(lib_path, ds, n_callers, n_tickets, reads_per_ticket, opts_kw=None, mix_fasta=False):
    fa, rt = ds + ".index.k31.fasta.gz", ds + ".index.k31.rtsk"
    L = api.load_library(lib_path)
    pg = api.Graph(fa, rt, 31, device=0, lib_path=lib_path)
    reads = op.read_fastq(ds + ".lr.fq")
    parts = [reads[(i * reads_per_ticket) % (len(reads) - reads_per_ticket):][:reads_per_ticket] for i in range(n_tickets)]
    kws = [(opts_kw[i % len(opts_kw)] if opts_kw else {}) for i in range(n_tickets)]
    opts = [pg.opts(**kw) for kw in kws]
    out, err, nxt, lock = [None] * n_tickets, [], [0], threading.Lock()
    g0, t0 = C.c_uint64(), C.c_uint64()
    assert L.rtk_coalesce_stats(pg.h, C.byref(g0), C.byref(t0)) == 0
    start = threading.Barrier(n_callers)

    def caller():
        start.wait()
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= n_tickets:
                return
            p = parts[i]; n = len(p)
            sa = (C.c_char_p * n)(*[r[1].encode() for r in p]); la = (C.c_uint32 * n)(*[len(r[1]) for r in p])
            qa = None if (mix_fasta and i % 3 == 0) else (C.c_char_p * n)(*[r[2].encode() for r in p])
            os_, oq, ol = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_uint32 * n)()
            rc = L.rtk_correct_batch(pg.h, C.byref(opts[i]), n, sa, qa, la, os_, oq, ol)
            if rc != 0:
                err.append((rc, L.rtk_last_error().decode())); return
            out[i] = [(C.string_at(os_[j], ol[j]).decode(), C.string_at(oq[j], ol[j]).decode()) for j in range(n)]
            L.rtk_free_many(os_, n); L.rtk_free_many(oq, n)
            assert not os_[0] and not oq[n - 1]

    th = [threading.Thread(target=caller) for _ in range(n_callers)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not err, err[:2]
    g1, t1 = C.c_uint64(), C.c_uint64()
    assert L.rtk_coalesce_stats(pg.h, C.byref(g1), C.byref(t1)) == 0
    og = op.Graph(fa, rt, 31)
    cache = {}
    for i, p in enumerate(parts):
        key = (p[0][0], tuple(sorted(kws[i].items())))
        if key not in cache:
            cache[key], _ = og.correct_batch([r[1] for r in p], [r[2] for r in p], opts=og.opts(**kws[i]), threads=os.cpu_count() or 4)
        assert out[i] == cache[key], "ticket %d differs from the oracle" % i
    return g1.value - g0.value, t1.value - t0.value


def test_sim_concurrent_callers_are_merged_and_get_their_own_reads(ds_snps):
    groups, tickets = _run_callers(SIM_LIB, ds_snps, n_callers=6, n_tickets=18, reads_per_ticket=3)
    assert tickets == 18 and groups < tickets  # some calls shared a launch


def test_sim_tickets_with_other_options_or_input_kind_are_not_mixed(ds_snps):
    """Two option sets and FASTA / FASTQ tickets in flight at once: a group only holds tickets whose rtk_opts bytes and kind of input agree."""
    groups, tickets = _run_callers(SIM_LIB, ds_snps, n_callers=6, n_tickets=18, reads_per_ticket=3, opts_kw=[{}, dict(insert_sz=300, max_qual=30)], mix_fasta=True)
    assert tickets == 18


def test_sim_lone_caller_runs_every_call_on_its_own(ds_snps):
    groups, tickets = _run_callers(SIM_LIB, ds_snps, n_callers=1, n_tickets=4, reads_per_ticket=3)
    assert groups == tickets == 4  # nobody to merge with: no waiting, no merging


def test_sim_a_failed_group_does_not_fail_its_members(ds_snps, monkeypatch):
    """A merged batch that fails (test hook) is not the callers' failure: every member runs its own ticket again, alone, and gets the oracle's reads."""
    monkeypatch.setenv("RTK_TEST_COALESCE_FAIL", "1")
    groups, tickets = _run_callers(SIM_LIB, ds_snps, n_callers=6, n_tickets=12, reads_per_ticket=3)
    assert tickets == 12


@pytest.mark.gpu
def test_gpu_concurrent_callers_are_merged_and_get_their_own_reads(ds_medium):
    groups, tickets = _run_callers(None, ds_medium, n_callers=12, n_tickets=48, reads_per_ticket=9)
    assert tickets == 48 and groups < 40


@pytest.mark.gpu
def test_gpu_mixed_options_and_input_kinds(ds_medium):
    _run_callers(None, ds_medium, n_callers=8, n_tickets=24, reads_per_ticket=7, opts_kw=[{}, dict(insert_sz=300, max_qual=30)], mix_fasta=True)
