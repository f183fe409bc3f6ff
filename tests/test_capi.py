"""The C-ABI library loads and exports every symbol include/ratatosk_hip.h declares (no compute without a GPU),
and refuses to compute when no device is present (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, SIM_LIB
from ratatosk_amd import api


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ratatosk_hip.h")).read()
    return sorted(set(re.findall(r"\b(rtk_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported_by_product_library():
    L = ctypes.CDLL(api.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libratatosk_hip.so does not export %s" % s


def test_simulator_exports_same_abi():
    L = ctypes.CDLL(SIM_LIB)
    for s in declared_symbols():
        assert hasattr(L, s)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.RtkError) as e:
        api.myers_batch(["ACGT"], ["ACGT"])
    assert "no HIP device" in str(e.value) or "rtk error" in str(e.value)


@pytest.mark.gpu
def test_gpu_rtk_correct_batch_through_ctypes(ds_snps):
    """The one function SURVEY.md 8(b) spells out, called as a reference-side binding would call it (plain pointers, outputs malloc'd by the
    library and released with rtk_free) on the product library and a GPU: FASTQ input, FASTA input (qual = NULL), an empty batch, and the
    refusals (an rtk_opts that rtk_opts_default did not fill, null pointers) -- against the oracle."""
    import ctypes as C
    from oracle import oracle_py as op
    fa, rt = ds_snps + ".index.k31.fasta.gz", ds_snps + ".index.k31.rtsk"
    reads = op.read_fastq(ds_snps + ".lr.fq")[:12]
    seqs, quals = [r[1] for r in reads] + ["ACGT" * 5, "N" * 80], [r[2] for r in reads] + ["I" * 20, "#" * 80]
    L = api.load_library()
    assert os.path.samefile(L._name, api.LIB_PATH)  # the product library, not the simulator
    h = C.c_void_p()
    assert L.rtk_graph_load(fa.encode(), rt.encode(), 31, 4, C.byref(h)) == 0
    assert L.rtk_graph_upload(h, 0) == 0
    o = api.RtkOpts(); assert L.rtk_opts_default(h, C.byref(o)) == 0
    og = op.Graph(fa, rt, 31)
    n = len(seqs)
    sa = (C.c_char_p * n)(*[s.encode() for s in seqs]); qa = (C.c_char_p * n)(*[q.encode() for q in quals]); la = (C.c_uint32 * n)(*[len(s) for s in seqs])

    def call(qual_arg, opts=o, count=n):
        os_, oq, ol = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_uint32 * n)()
        rc = L.rtk_correct_batch(h, C.byref(opts), count, sa, qual_arg, la, os_, oq, ol)
        if rc != 0:
            return rc, None
        got = [(C.string_at(os_[i], ol[i]).decode(), C.string_at(oq[i], ol[i]).decode()) for i in range(count)]
        for i in range(count):  # outputs are the caller's: released with rtk_free (NUL-terminated: the length is also strlen)
            assert C.string_at(os_[i]) == got[i][0].encode()
            L.rtk_free(os_[i]); L.rtk_free(oq[i])
        return rc, got

    rc, got = call(qa)
    want, _ = og.correct_batch(seqs, quals)
    assert rc == 0 and got == want
    assert any(g[0] != s for g, s in zip(got, seqs))  # something was corrected
    # FASTA input: qual = NULL (src/Ratatosk.cpp:658,767); the first pass writes synthetic qualities either way
    rc, got_fa = call(None)
    want_fa, _ = og.correct_batch(seqs, None)
    assert rc == 0 and got_fa == want_fa and got_fa == want
    # an empty batch is not an error
    assert call(qa, count=0) == (0, [])
    # refusals: a zero-filled rtk_opts, one from a shorter header, null pointers -- error codes and a message, nothing is run
    z = api.RtkOpts()
    rc, _ = call(qa, opts=z)
    assert rc == -3 and b"rtk_opts_default" in L.rtk_last_error()
    short = api.RtkOpts(); C.memmove(C.byref(short), C.byref(o), C.sizeof(o)); short.struct_size -= 4
    assert call(qa, opts=short)[0] == -3
    os_, oq, ol = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_uint32 * n)()
    assert L.rtk_correct_batch(None, C.byref(o), n, sa, qa, la, os_, oq, ol) == -3
    assert L.rtk_correct_batch(h, C.byref(o), n, sa, qa, la, None, oq, ol) == -3
    # rtk_seeds refuses the same way (every entry that takes an rtk_opts does)
    ns, nw = C.c_uint64(), C.c_uint64(); sol, wk = (C.c_int64 * 64)(), (C.c_int64 * 64)()
    assert L.rtk_seeds(h, C.byref(z), seqs[0].encode(), len(seqs[0]), C.byref(ns), sol, C.byref(nw), wk, 16) == -3
    L.rtk_graph_free(h)
