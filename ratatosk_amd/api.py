"""Python host side over the C ABI of libratatosk_hip.so (include/ratatosk_hip.h).

Mirrors the reference's per-read seam (reference: src/Ratatosk.cpp:808-864 -- getSeeds + correctSequence on a
shared read-only graph) with the same option names as Correct_Opt (src/Common.hpp:101-156). All compute happens
in the HIP kernels; this module only marshals pointers. There is no CPU fallback: a missing extension or a
missing GPU raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTK_LIB_OVERRIDE") or os.path.join(_HERE, "libratatosk_hip.so")  # the override is for A/B runs of two builds (profiles/scripts/ab_lib.sh)


class RtkError(RuntimeError):
    pass


class RtkOpts(C.Structure):
    _fields_ = [("insert_sz", C.c_uint64), ("min_cov_vertices", C.c_uint64), ("max_len_weak_region1", C.c_uint64),
                ("max_km_cov", C.c_uint64), ("weak_region_len_factor", C.c_double), ("large_k_factor", C.c_double),
                ("min_score", C.c_double), ("max_qual", C.c_int32), ("out_qual", C.c_int32), ("min_confidence_snp_corr", C.c_double),
                ("long_read_correct", C.c_int32), ("force_unres_snp_corr", C.c_int32), ("max_len_weak_region2", C.c_uint64),
                ("a2_exclusive", C.c_int32), ("a3_strand_order", C.c_int32), ("d1_desc", C.c_int32), ("struct_size", C.c_uint32)]


class RtkGraphInfo(C.Structure):
    _fields_ = [("n_unitigs", C.c_uint64), ("n_kmers", C.c_uint64), ("n_bases", C.c_uint64), ("n_colour_ids", C.c_uint64),
                ("n_global_sets", C.c_uint64), ("table_slots", C.c_uint64), ("hbm_bytes", C.c_uint64), ("max_km_cov_top", C.c_uint64),
                ("k", C.c_int32), ("device", C.c_int32)]


class RtkStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_total", "ms_lookup_exact", "ms_mask", "ms_lookup_inexact", "ms_seeds", "ms_regions", "ms_correct", "ms_stitch")] + \
               [(n, C.c_uint64) for n in ("n_windows", "n_probes_exact", "n_probes_inexact", "n_hits_inexact", "n_regions", "n_region_items", "n_arena_overflow",
                                         "n_expand", "n_colour_elem", "n_path_base", "n_align", "n_align_cells", "in_bases", "out_bases",
                                         "cyc_colour", "cyc_paths", "cyc_consensus", "cyc_total", "cyc_myers", "cyc_sets", "cyc_tostring", "cyc_pathqual", "n_slots_exact", "n_slots_inexact", "cyc_walk", "n_moves")] + \
               [("ms_lanes", C.c_double), ("n_lane_regions", C.c_uint64), ("n_lane_handed", C.c_uint64), ("ms_phase", C.c_double), ("n_phase_skipped", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_libs = {}


# kernels of several tickets / streams only run side by side if the HIP runtime may open enough hardware queues (default 4 per process);
# read when the runtime initialises, so it is set before the library is loaded (and only if the caller has not chosen a value)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def load_library(path=None):
    """dlopen the extension; fails loudly when it has not been built (no silent fallback)."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RtkError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (make -C ratatosk_amd/csrc)" % path)
    L = C.CDLL(path)
    L.rtk_last_error.restype = C.c_char_p
    L.rtk_version.restype = C.c_char_p
    L.rtk_graph_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.rtk_graph_load2.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.rtk_graph_move_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.rtk_graph_host_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.rtk_graph_download_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.rtk_graph_upload.argtypes = [C.c_void_p, C.c_int]
    L.rtk_graph_shell.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.rtk_graph_n_buffers.argtypes = [C.c_void_p]
    L.rtk_graph_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.rtk_graph_alloc_buffers.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int, C.POINTER(RtkGraphInfo)]
    L.rtk_graph_adopt_device.argtypes = [C.c_void_p]
    L.rtk_graph_buffer_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    L.rtk_graph_get_info.argtypes = [C.c_void_p, C.POINTER(RtkGraphInfo)]
    L.rtk_graph_free.argtypes = [C.c_void_p]
    L.rtk_opts_default.argtypes = [C.c_void_p, C.POINTER(RtkOpts)]
    L.rtk_correct_batch.argtypes = [C.c_void_p, C.POINTER(RtkOpts), C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    L.rtk_batch_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]
    L.rtk_batch_create2.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]
    L.rtk_batch_run.argtypes = [C.c_void_p, C.POINTER(RtkOpts)]
    L.rtk_graph_strip_annotations.restype = C.c_longlong
    L.rtk_graph_strip_annotations.argtypes = [C.c_void_p]
    L.rtk_batch_run_seeds.argtypes = [C.c_void_p, C.POINTER(RtkOpts)]
    L.rtk_batch_run_regions.argtypes = [C.c_void_p, C.POINTER(RtkOpts)]
    L.rtk_batch_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    L.rtk_batch_fetch_view.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]
    L.rtk_n_devices.restype = C.c_int
    L.rtk_device_memory.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rtk_graph_clone_to_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.rtk_batch_get_stats.argtypes = [C.c_void_p, C.POINTER(RtkStats)]
    L.rtk_batch_free.argtypes = [C.c_void_p]
    L.rtk_lookup_exact.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_int64)]
    L.rtk_fix_snps.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p]
    L.rtk_seeds.argtypes = [C.c_void_p, C.POINTER(RtkOpts), C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_int64),
                            C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.c_uint64]
    L.rtk_myers_batch.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_uint32, C.c_char_p, C.c_uint32]
    L.rtk_myers_batch_waves.argtypes = L.rtk_myers_batch.argtypes + [C.c_int]
    L.rtk_myers_batch_lanes.argtypes = L.rtk_myers_batch.argtypes
    L.rtk_myers_batch_lanes.restype = C.c_int
    L.rtk_coalesce_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rtk_graph_reserve_batches.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64]
    L.rtk_free_many.argtypes = [C.POINTER(C.c_void_p), C.c_uint32]; L.rtk_free_many.restype = None
    L.rtk_free.argtypes = [C.c_void_p]
    _libs[path] = L
    return L


RTK_LOAD_DEVICE_TABLES = 1


def _b(s):
    return s.encode() if isinstance(s, str) else s


class Graph:
    """The compacted coloured de Bruijn graph, loaded from the reference's index files and resident in HBM."""

    def __init__(self, fasta_gz, rtsk, k=31, device=0, lib_path=None, upload=True, n_threads=None, host_tables=None):
        """host_tables: build the lookup structures (k-mer table, half-k-mer index, adjacency) on the host threads instead of in HBM at upload time.
        Default: on the device whenever the graph is uploaded (RTK_LOAD_DEVICE_TABLES); a graph that stays on the host (upload=False) gets the host's."""
        self.L = load_library(lib_path)
        self.h = C.c_void_p()
        if n_threads is None:  # parse + flatten on the host's threads (same image whatever their number)
            n_threads = max(1, min(32, os.cpu_count() or 1))
        if host_tables is None:
            host_tables = not upload or os.environ.get("RTK_HOST_TABLES") == "1"
        self._check(self.L.rtk_graph_load2(_b(fasta_gz), _b(rtsk), k, n_threads, 0 if host_tables else RTK_LOAD_DEVICE_TABLES, C.byref(self.h)))
        self.k = k
        if upload:
            self._check(self.L.rtk_graph_upload(self.h, device))

    def _check(self, rc):
        if rc != 0:
            raise RtkError("rtk error %d: %s" % (rc, self.L.rtk_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.rtk_graph_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = RtkGraphInfo()
        self._check(self.L.rtk_graph_get_info(self.h, C.byref(i)))
        return i

    def opts(self, **kw):
        o = RtkOpts()
        self._check(self.L.rtk_opts_default(self.h, C.byref(o)))
        for k_, v in kw.items():
            setattr(o, k_, v)
        return o

    def lookup_exact(self, seq):
        s = _b(seq)
        nw = max(0, len(s) - self.k + 1)
        out = (C.c_int64 * max(1, nw))()
        self._check(self.L.rtk_lookup_exact(self.h, s, len(s), out))
        return [out[i] for i in range(nw)]

    def fix_snps(self, seq):
        """fixSNPs() of one read (`-f` of the second pass, src/Alignment.cpp:846-965)."""
        s = _b(seq)
        out = C.create_string_buffer(len(s) + 1)
        self._check(self.L.rtk_fix_snps(self.h, s, len(s), out))
        return out.raw[:len(s)]

    def seeds(self, seq, opts=None):
        s = _b(seq)
        o = opts or self.opts()
        cap = 64 * len(s) + 64
        ns, nw = C.c_uint64(), C.c_uint64()
        so, we = (C.c_int64 * (4 * cap))(), (C.c_int64 * (4 * cap))()
        self._check(self.L.rtk_seeds(self.h, C.byref(o), s, len(s), C.byref(ns), so, C.byref(nw), we, cap))
        f = lambda a, n: [(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]) for i in range(n)]
        return f(so, ns.value), f(we, nw.value)

    def correct_batch(self, seqs, quals=None, opts=None, raw=None):
        """[(corrected seq, corrected qual)] for a batch of long reads (one call == one ticket batch).
        Second pass: opts.long_read_correct = 1, quals = the pass-1 qualities, raw = the uncorrected reads (same order)."""
        b = Batch(self, seqs, quals, raw)
        try:
            b.run(opts)
            return b.fetch()
        finally:
            b.close()


class Batch:
    """A batch of long reads resident in HBM: create (H2D) / run (kernels) / fetch (D2H)."""

    def __init__(self, graph, seqs, quals=None, raw=None):
        self.g = graph
        self.L = graph.L
        n = len(seqs)
        bs = [_b(s) for s in seqs]
        seq_arr = (C.c_char_p * n)(*bs)
        qual_arr = None
        if quals is not None:
            bq = [_b(q) for q in quals]
            qual_arr = (C.c_char_p * n)(*bq)
        lens = (C.c_uint32 * n)(*[len(b) for b in bs])
        self.n = n
        self.in_bases = sum(len(b) for b in bs)
        self.h = C.c_void_p()
        if raw is not None:
            br = [_b(s) for s in raw]
            raw_arr = (C.c_char_p * n)(*br)
            raw_lens = (C.c_uint32 * n)(*[len(b) for b in br])
            graph._check(self.L.rtk_batch_create2(graph.h, n, seq_arr, qual_arr, lens, raw_arr, raw_lens, C.byref(self.h)))
        else:
            graph._check(self.L.rtk_batch_create(graph.h, n, seq_arr, qual_arr, lens, C.byref(self.h)))

    def run(self, opts=None):
        o = opts or self.g.opts()
        self.g._check(self.L.rtk_batch_run(self.h, C.byref(o)))

    def run_seeds(self, opts=None):
        """Stage A (anchors). May run on another host thread while run_regions of a different batch is in flight."""
        o = opts or self.g.opts()
        self.g._check(self.L.rtk_batch_run_seeds(self.h, C.byref(o)))

    def run_regions(self, opts=None):
        """Stage B (region correction + stitching); needs run_seeds of this batch."""
        o = opts or self.g.opts()
        self.g._check(self.L.rtk_batch_run_regions(self.h, C.byref(o)))

    def fetch(self):
        n = self.n
        os_, oq = (C.c_void_p * n)(), (C.c_void_p * n)()
        ol = (C.c_uint32 * n)()
        self.g._check(self.L.rtk_batch_fetch(self.h, os_, oq, ol))
        out = []
        for i in range(n):
            out.append((C.string_at(os_[i], ol[i]).decode(), C.string_at(oq[i], ol[i]).decode()))
            self.L.rtk_free(os_[i]); self.L.rtk_free(oq[i])
        return out

    def stats(self):
        s = RtkStats()
        self.g._check(self.L.rtk_batch_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def close(self):
        if getattr(self, "h", None):
            self.L.rtk_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def myers_batch(queries, targets, ks=None, modes=None, want_path=False, use_iupac=True, lib_path=None, waves=0, lanes=False):
    """edlibAlign over a batch on the device: returns [(editDistance, endLocations, cigar)]. lanes=True: one problem per lane (stage entry
    rtk_myers_batch_lanes)."""
    L = load_library(lib_path)
    n = len(queries)
    bq = [_b(q) for q in queries]; bt = [_b(t) for t in targets]
    ks = ks if ks is not None else [-1] * n
    modes = modes if modes is not None else [0] * n
    qa = (C.c_char_p * n)(*bq); ta = (C.c_char_p * n)(*bt)
    ql = (C.c_uint32 * n)(*[len(x) for x in bq]); tl = (C.c_uint32 * n)(*[len(x) for x in bt])
    ka = (C.c_int32 * n)(*ks); ma = (C.c_int32 * n)(*modes)
    cap_locs = max(len(x) for x in bt) + 2 if n else 2
    cap_cig = 4 * (max((len(x) for x in bq), default=0) + max((len(x) for x in bt), default=0)) + 16
    dist = (C.c_int32 * n)(); nloc = (C.c_int32 * n)(); locs = (C.c_int32 * (n * cap_locs))()
    cig = C.create_string_buffer(n * cap_cig) if want_path else None
    if lanes:  # one problem per lane (stage entry; csrc/hip/rtk_myers_lane.h)
        rc = L.rtk_myers_batch_lanes(n, qa, ql, ta, tl, ka, ma, 1 if want_path else 0, 1 if use_iupac else 0, dist, nloc, locs, cap_locs, cig, cap_cig)
    elif waves > 1:  # the multi-wave schedule of the second pass's whole-read alignment (stage entry)
        rc = L.rtk_myers_batch_waves(n, qa, ql, ta, tl, ka, ma, 1 if want_path else 0, 1 if use_iupac else 0, dist, nloc, locs, cap_locs, cig, cap_cig, waves)
    else:
        rc = L.rtk_myers_batch(n, qa, ql, ta, tl, ka, ma, 1 if want_path else 0, 1 if use_iupac else 0, dist, nloc, locs, cap_locs, cig, cap_cig)
    if rc != 0:
        raise RtkError("rtk error %d: %s" % (rc, L.rtk_last_error().decode()))
    out = []
    for i in range(n):
        el = [locs[i * cap_locs + j] for j in range(min(nloc[i], cap_locs))] if dist[i] >= 0 else []
        c = ""
        if want_path:
            c = cig.raw[i * cap_cig:(i + 1) * cap_cig].split(b"\0", 1)[0].decode()
        out.append((dist[i], el, c))
    return out


def myers_lanes_last_routes(lib_path=None):
    """(problems computed one per lane, problems handed on to the wave route) of this thread's last myers_batch(..., lanes=True) call."""
    L = load_library(lib_path)
    a, b = C.c_uint64(), C.c_uint64()
    L.rtk_myers_lanes_last_routes.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]; L.rtk_myers_lanes_last_routes.restype = None
    L.rtk_myers_lanes_last_routes(C.byref(a), C.byref(b))
    return a.value, b.value


def run_pipelined(batches, opts=None):
    """Runs the batches in order with the seed stage of batch i+1 overlapping the region stage of batch i (two host threads,
    one HIP stream per batch). Results are the same as calling run() on each batch."""
    import threading
    prev, err = None, []

    def seeds(b):
        try:
            b.run_seeds(opts)
        except Exception as e:  # surfaced on the calling thread
            err.append(e)

    for b in batches:
        t = threading.Thread(target=seeds, args=(b,))
        t.start()
        if prev is not None:
            prev.run_regions(opts)
        t.join()
        if err:
            raise err[0]
        prev = b
    if prev is not None:
        prev.run_regions(opts)
