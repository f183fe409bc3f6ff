"""Whole per-read correction (seeds -> regions -> stitch) of the device programs, executed by the host simulator,
against the oracle: corrected sequence AND quality strings must be byte-identical for every read."""
from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _check(prefix, n, lib_path, extra_reads=(), counters_must_match=True, k=31, opts=None, threads=4):
    """opts: overrides of the Correct_Opt fields (same names in the oracle's and the library's option structs)."""
    fa, rt = prefix + ".index.k%d.fasta.gz" % k, prefix + ".index.k%d.rtsk" % k
    og, pg = op.Graph(fa, rt, k), api.Graph(fa, rt, k, device=0, lib_path=lib_path)
    reads = op.read_fastq(prefix + ".lr.fq")[:n]
    seqs = [r[1] for r in reads] + list(extra_reads)
    quals = [r[2] for r in reads] + ["I" * len(s) for s in extra_reads]
    b = api.Batch(pg, seqs, quals)
    b.run(pg.opts(**(opts or {})))
    got = b.fetch()
    st = b.stats()
    want, cnt = og.correct_batch(seqs, quals, opts=og.opts(**(opts or {})), threads=threads)
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert g_[0] == w_[0], "sequence of read %d differs" % i
        assert g_[1] == w_[1], "quality of read %d differs" % i
    if counters_must_match:  # redone regions are counted twice, so only checked on runs without scratch overflow
        assert 0 < st["n_expand"] <= cnt["n_expand"] and 0 < st["n_colour_elem"] <= cnt["n_colour_elem"]  # the device prunes DFS subtrees that cannot matter (rtk_explore_subgraph): never more events than the reference walk
    return st, got, seqs


def test_sim_correct_branching(ds_small):
    s0 = op.read_fastq(ds_small + ".lr.fq")[0][1]
    # edge cases of correctSequence (src/Correction.cpp:165-171): too short, no solid anchor, lower case, N runs
    extra = ["ACGT" * 5, "A" * 31, "N" * 200, s0[:500].lower(), s0[:400] + "N" * 40 + s0[440:1200]]
    st, got, seqs = _check(ds_small, 12, SIM_LIB, extra)
    assert st["n_expand"] > 0 and st["n_regions"] > len(seqs)
    assert got[len(seqs) - 5][0] == "ACGT" * 5 and set(got[len(seqs) - 5][1]) == {"!"}


def test_sim_correct_clean(ds_clean):
    _check(ds_clean, 8, SIM_LIB)


def test_sim_scratch_overflow_is_redone_on_device(ds_small, monkeypatch):
    """Work areas that are too small must not change results: affected reads / regions are redone with bigger arenas."""
    monkeypatch.setenv("RTK_TEST_TINY_SCRATCH", "1")
    st, got, seqs = _check(ds_small, 12, SIM_LIB, counters_must_match=False)
    assert st["n_arena_overflow"] > 0


def test_sim_correct_other_options(ds_small):
    """-i / -w / -Q away from their defaults (src/Ratatosk.cpp:145-301): shorter insert size changes masking and colour windows,
    a short maximum weak-region length leaves long regions uncorrected (src/Correction.cpp:74), other quality scale."""
    _check(ds_small, 8, SIM_LIB, opts=dict(insert_sz=300, max_len_weak_region1=300, max_qual=30))


def test_sim_correct_k25(ds_k25):
    s0 = op.read_fastq(ds_k25 + ".lr.fq")[0][1]
    # IUPAC codes and N inside reads: windows over them are never looked up, alignments use the 28-pair table (src/Common.hpp:262-276)
    extra = [s0[:300] + "R" + s0[301:700] + "YN" + s0[702:1500], s0[:25], s0[:26]]
    _check(ds_k25, 6, SIM_LIB, extra, k=25)


def test_sim_correct_k21(ds_k21):
    _check(ds_k21, 8, SIM_LIB, k=21)


def test_sim_correct_snp_annotations(ds_snps):
    """Index with SNP annotations (what the reference's `index` step writes unless -F, src/Graph.cpp:484): getAmbiguityVector
    (src/GraphTraversal.cpp:966-1036) and fixAmbiguity (src/Alignment.cpp:527-844) on the device equal the oracle's, also with
    another confidence threshold (-m) and quality range; and they do change results."""
    _, got, seqs = _check(ds_snps, 40, SIM_LIB)
    _check(ds_snps, 12, SIM_LIB, opts=dict(min_confidence_snp_corr=0.5, out_qual=3, max_qual=30))
    plain = api.Graph(ds_snps + "_plain.index.k31.fasta.gz", ds_snps + "_plain.index.k31.rtsk", 31, device=0, lib_path=SIM_LIB).correct_batch(seqs, ["I" * len(x) for x in seqs])
    assert sum(1 for a, b in zip(got, plain) if a != b) >= 5
    assert all(len(a[0]) == len(a[1]) for a in got)


def test_sim_snp_annotations_with_repeats_cycles_and_tiny_scratch(ds_snps_rich, ds_snps, monkeypatch):
    """SNP annotations together with short cycles, repeats and global colour sets; and with work areas that overflow (the sets of
    rtk_ambiguity.h live in the region scratch lists: affected regions are redone with bigger ones)."""
    _check(ds_snps_rich, 16, SIM_LIB)
    monkeypatch.setenv("RTK_TEST_TINY_SCRATCH", "1")
    st, _, _ = _check(ds_snps, 10, SIM_LIB, counters_must_match=False)
    assert st["n_arena_overflow"] > 0


def test_strip_annotations_gives_the_plain_index(ds_snps):
    """rtk_graph_strip_annotations (CLI --strip-annotations) drops the SNP / short-cycle annotations before the upload: same results
    as the index built without them."""
    import ctypes as C
    fa, rt = ds_snps + ".index.k31.fasta.gz", ds_snps + ".index.k31.rtsk"
    reads = op.read_fastq(ds_snps + ".lr.fq")[:6]
    seqs, quals = [r[1] for r in reads], [r[2] for r in reads]
    L = api.load_library(SIM_LIB)
    h = C.c_void_p()
    assert L.rtk_graph_load(fa.encode(), rt.encode(), 31, 1, C.byref(h)) == 0
    assert L.rtk_graph_strip_annotations(h) > 100
    assert L.rtk_graph_upload(h, 0) == 0
    o = api.RtkOpts(); assert L.rtk_opts_default(h, C.byref(o)) == 0
    want = api.Graph(ds_snps + "_plain.index.k31.fasta.gz", ds_snps + "_plain.index.k31.rtsk", 31, device=0, lib_path=SIM_LIB).correct_batch(seqs, quals)
    n = len(seqs)
    sa = (C.c_char_p * n)(*[s.encode() for s in seqs]); qa = (C.c_char_p * n)(*[q.encode() for q in quals]); la = (C.c_uint32 * n)(*[len(s) for s in seqs])
    os_, oq, ol = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_uint32 * n)()
    assert L.rtk_correct_batch(h, C.byref(o), n, sa, qa, la, os_, oq, ol) == 0
    got = [(C.string_at(os_[i], ol[i]).decode(), C.string_at(oq[i], ol[i]).decode()) for i in range(n)]
    assert got == want
    # an rtk_opts the library's rtk_opts_default did not fill (a zero-filled struct, a caller built against an older, shorter header) is refused, not run
    assert o.struct_size == C.sizeof(api.RtkOpts) and L.rtk_api_revision() >= 5
    z = api.RtkOpts()
    assert L.rtk_correct_batch(h, C.byref(z), n, sa, qa, la, os_, oq, ol) == -3 and b"rtk_opts_default" in L.rtk_last_error()
    o.struct_size -= 4
    assert L.rtk_correct_batch(h, C.byref(o), n, sa, qa, la, os_, oq, ol) == -3
    L.rtk_graph_free(h)


def test_sim_correct_short_cycles(ds_tandem):
    """Index with short-cycle annotations (detectShortCycles restated in rtk_build_index): fixRepeats (src/GraphTraversal.cpp:1149-1334)
    on the device equals the oracle's."""
    _check(ds_tandem, 40, SIM_LIB)
