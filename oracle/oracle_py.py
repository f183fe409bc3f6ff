"""ORACLE (test infrastructure): ctypes access to oracle/liboracle.so and oracle/_ref/libedlib_ref.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. The product
(ratatosk_amd/) never does.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcOpts(C.Structure):
    # mirrors struct orc_opts in oracle_capi.cpp (pass-1 fields of the reference's Correct_Opt, src/Common.hpp:101-156)
    _fields_ = [("insert_sz", C.c_uint64), ("min_cov_vertices", C.c_uint64), ("max_len_weak_region1", C.c_uint64),
                ("max_km_cov", C.c_uint64), ("weak_region_len_factor", C.c_double), ("large_k_factor", C.c_double),
                ("min_score", C.c_double), ("max_qual", C.c_int32), ("out_qual", C.c_int32), ("min_confidence_snp_corr", C.c_double),
                ("max_len_weak_region2", C.c_uint64), ("skip_phasing", C.c_int32), ("force_unres_snp_corr", C.c_int32)]


def default_opts(max_km_cov=128):
    return OrcOpts(500, 2, 1000, max_km_cov, 0.25, 1.5, 0.0, 40, 1, 0.9, 5000, 0, 0)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_graph_load.restype = C.c_void_p
        L.orc_graph_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_graph_free.argtypes = [C.c_void_p]
        L.orc_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_unitig.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_global_set.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_neighbours.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_int64)]
        L.orc_unitig_annotations.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64),
                                             C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_myers.restype = C.c_int
        L.orc_myers.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
        L.orc_exact.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_int64)]
        L.orc_seeds.argtypes = [C.c_void_p, C.POINTER(OrcOpts), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int64),
                                C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_uint64]
        L.orc_inexact.argtypes = [C.c_void_p, C.POINTER(OrcOpts), C.c_char_p, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_uint64]
        L.orc_correct_batch.argtypes = [C.c_void_p, C.POINTER(OrcOpts), C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                        C.c_int, C.POINTER(C.c_uint64)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_use_reference_edlib.argtypes = [C.c_char_p]
        L.orc_correct_batch2.argtypes = [C.c_void_p, C.POINTER(OrcOpts), C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int]
        L.orc_fix_snps.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p]
        L.orc_wyhash8.restype = C.c_uint64
        L.orc_wyhash8.argtypes = [C.c_uint64, C.c_uint64]
        _lib = L
    return _lib


def use_reference_edlib(on=True):
    """Route every alignment of the oracle's correction through the REFERENCE's edlib (oracle/_ref); False: back to oracle_myers.cpp."""
    path = os.path.join(_HERE, "_ref", "libedlib_ref.so")
    if on and not os.path.exists(path):
        raise RuntimeError("oracle/_ref/libedlib_ref.so not built")
    if lib().orc_use_reference_edlib(path.encode() if on else None) != 0:
        raise RuntimeError("oracle/_ref/libedlib_ref.so lacks ref_edlib_moves: rebuild it (make -C oracle ref)")


_ref = None


def ref_edlib_lib():
    """The REFERENCE's own edlib (compiled from /root/reference/src/edlib.cpp into oracle/_ref/); None if absent."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libedlib_ref.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_edlib.restype = C.c_int
        L.ref_edlib.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
        _ref = L
    return _ref


def _align(fn, q, t, k, mode, path, iupac=True):
    q = q.encode() if isinstance(q, str) else q
    t = t.encode() if isinstance(t, str) else t
    cap = len(t) + 2
    locs = (C.c_int * cap)()
    nloc = C.c_int(0)
    cig = C.create_string_buffer(4 * (len(q) + len(t)) + 16)
    d = fn(q, len(q), t, len(t), k, mode, 1 if path else 0, 1 if iupac else 0, C.byref(nloc), locs, cap, cig, len(cig))
    return d, [locs[i] for i in range(nloc.value)] if d >= 0 else [], cig.value.decode()


def myers(q, t, k=-1, mode=0, path=False, iupac=True):
    """oracle Myers: (editDistance, endLocations, cigar). mode 0 NW, 1 SHW, 2 HW."""
    return _align(lib().orc_myers, q, t, k, mode, path, iupac)


def ref_edlib(q, t, k=-1, mode=0, path=False, iupac=True):
    L = ref_edlib_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libedlib_ref.so not built")
    return _align(L.ref_edlib, q, t, k, mode, path, iupac)


class Graph:
    def __init__(self, fasta_gz, rtsk, k=31):
        err = C.create_string_buffer(512)
        self.h = lib().orc_graph_load(fasta_gz.encode(), rtsk.encode(), k, err, 512)
        if not self.h:
            raise RuntimeError("oracle graph load failed: " + err.value.decode())
        self.k = k
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().orc_graph_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        self.n_unitigs, self.n_kmers, self.max_km_cov_top = a.value, b.value, c.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_graph_free(self.h)
                self.h = None
        except Exception:
            pass

    def opts(self, **kw):
        o = default_opts(max(self.max_km_cov_top, 128))  # src/Ratatosk.cpp:625
        for k_, v in kw.items():
            setattr(o, k_, v)
        return o

    def unitig(self, u):
        sl, kc, sh, gid, nl = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int64(), C.c_uint64()
        lib().orc_unitig(self.h, u, None, 0, C.byref(sl), C.byref(kc), C.byref(sh), C.byref(gid), None, 0, C.byref(nl))
        seq = C.create_string_buffer(sl.value + 1)
        loc = (C.c_uint32 * max(1, nl.value))()
        lib().orc_unitig(self.h, u, seq, sl.value, C.byref(sl), C.byref(kc), C.byref(sh), C.byref(gid), loc, nl.value, C.byref(nl))
        return dict(seq=seq.raw[:sl.value].decode(), kmcov=kc.value, shared=sh.value, global_id=gid.value, local=[loc[i] for i in range(nl.value)])

    def global_set(self, gid):
        n = C.c_uint64()
        lib().orc_global_set(self.h, gid, None, 0, C.byref(n))
        out = (C.c_uint32 * max(1, n.value))()
        lib().orc_global_set(self.h, gid, out, n.value, C.byref(n))
        return [out[i] for i in range(n.value)]

    def neighbours(self, u, direction):
        out = (C.c_int64 * 4)()
        lib().orc_neighbours(self.h, u, direction, out)
        return [out[i] for i in range(4)]

    def annotations(self, u):
        """What the index holds for unitig u: ([(position, IUPAC code)], [compact cycle strings])."""
        na, nc = C.c_uint64(), C.c_uint64()
        lib().orc_unitig_annotations(self.h, u, None, None, 0, C.byref(na), None, 0, C.byref(nc))
        pos = (C.c_uint32 * max(1, na.value))()
        code = C.create_string_buffer(max(1, na.value))
        cyc = C.create_string_buffer(max(1, nc.value))
        lib().orc_unitig_annotations(self.h, u, pos, code, na.value, C.byref(na), cyc, nc.value, C.byref(nc))
        amb = [(pos[i], code.raw[i:i + 1].decode()) for i in range(na.value)]
        raw = cyc.raw[:nc.value]
        cycles = [c.decode() for c in raw.split(b"\0")[:-1]] if nc.value else []
        return amb, cycles

    def fix_snps(self, seq):
        """fixSNPs() of one read (src/Alignment.cpp:846-965)."""
        s = seq.encode() if isinstance(seq, str) else seq
        out = C.create_string_buffer(len(s) + 1)
        lib().orc_fix_snps(self.h, s, len(s), out)
        return out.raw[:len(s)]

    def exact(self, seq):
        s = seq.encode() if isinstance(seq, str) else seq
        nw = max(0, len(s) - self.k + 1)
        out = (C.c_int64 * max(1, nw))()
        lib().orc_exact(self.h, s, len(s), out)
        return [out[i] for i in range(nw)]

    def seeds(self, seq, opts=None):
        s = seq.encode() if isinstance(seq, str) else seq
        o = opts or self.opts()
        cap = 64 * len(s) + 64
        ns, nw = C.c_uint64(), C.c_uint64()
        so, we = (C.c_int64 * (4 * cap))(), (C.c_int64 * (4 * cap))()
        rc = lib().orc_seeds(self.h, C.byref(o), s, len(s), C.byref(ns), so, C.byref(nw), we, cap)
        assert rc == 0
        f = lambda a, n: [(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]) for i in range(n)]
        return f(so, ns.value), f(we, nw.value)

    def inexact(self, seq, opts=None):
        s = seq.encode() if isinstance(seq, str) else seq
        o = opts or self.opts()
        cap = 64 * len(s) + 64
        n = C.c_uint64()
        hits = (C.c_int64 * (4 * cap))()
        masked = C.create_string_buffer(len(s) + 1)
        rc = lib().orc_inexact(self.h, C.byref(o), s, len(s), masked, C.byref(n), hits, cap)
        assert rc == 0
        return masked.raw[:len(s)].decode(), [(hits[4 * i], hits[4 * i + 1], hits[4 * i + 2], hits[4 * i + 3]) for i in range(n.value)]

    def correct_batch(self, seqs, quals=None, opts=None, threads=1):
        """Returns ([(seq, qual)], counters dict)."""
        n = len(seqs)
        o = opts or self.opts()
        bs = [s.encode() if isinstance(s, str) else s for s in seqs]
        seq_arr = (C.c_char_p * n)(*bs)
        if quals is not None:
            bq = [q.encode() if isinstance(q, str) else q for q in quals]
            qual_arr = (C.c_char_p * n)(*bq)
        else:
            qual_arr = None
        lens = (C.c_uint32 * n)(*[len(b) for b in bs])
        os_, oq = (C.c_void_p * n)(), (C.c_void_p * n)()
        ol = (C.c_uint32 * n)()
        cnt = (C.c_uint64 * 8)()
        lib().orc_correct_batch(self.h, C.byref(o), n, seq_arr, qual_arr, lens, os_, oq, ol, threads, cnt)
        out = []
        for i in range(n):
            out.append((C.string_at(os_[i], ol[i]).decode(), C.string_at(oq[i], ol[i]).decode()))
            lib().orc_free(os_[i]); lib().orc_free(oq[i])
        names = ["n_probe", "n_verify", "n_expand", "n_colour_elem", "n_path_base", "n_align", "n_align_cells", "n_regions"]
        return out, {k_: cnt[i] for i, k_ in enumerate(names)}


def _graph_correct_batch2(self, seqs, quals, raws, opts=None, threads=1):
    """Pass 2 (`correct -2`): seqs / quals = pass-1 corrected reads, raws = the uncorrected reads, same order. Returns [(seq, qual)]."""
    n = len(seqs)
    o = opts or self.opts()
    enc = lambda v: [x.encode() if isinstance(x, str) else x for x in v]
    bs, bq, br = enc(seqs), enc(quals), enc(raws)
    sa, qa, ra = (C.c_char_p * n)(*bs), (C.c_char_p * n)(*bq), (C.c_char_p * n)(*br)
    la, rl = (C.c_uint32 * n)(*[len(b) for b in bs]), (C.c_uint32 * n)(*[len(b) for b in br])
    os_, oq, ol = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_uint32 * n)()
    lib().orc_correct_batch2(self.h, C.byref(o), n, sa, qa, la, ra, rl, os_, oq, ol, threads)
    out = []
    for i in range(n):
        out.append((C.string_at(os_[i], ol[i]).decode(), C.string_at(oq[i], ol[i]).decode()))
        lib().orc_free(os_[i]); lib().orc_free(oq[i])
    return out


Graph.correct_batch2 = _graph_correct_batch2


def read_fastq(path):
    """Tiny FASTQ/FASTA reader for tests (name, seq, qual)."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    recs = []
    with op(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                break
            h = h.rstrip("\n")
            if not h:
                continue
            if h[0] == "@":
                s = f.readline().rstrip("\n"); f.readline(); q = f.readline().rstrip("\n")
                recs.append((h[1:].split()[0], s, q))
            elif h[0] == ">":
                s = f.readline().rstrip("\n")
                recs.append((h[1:].split()[0], s, ""))
    return recs
