"""Host logic: the product's flat-graph loader against the oracle's independent loader on the same index files
(unitigs, data words, colour sets incl. global/local split, adjacency), and the .rtsk PairID stream codecs."""
import ctypes as C

from oracle import oracle_py as op
from ratatosk_amd import api


def test_flat_graph_matches_oracle_loader(ds_small):
    fa, rt = ds_small + ".index.k31.fasta.gz", ds_small + ".index.k31.rtsk"
    og = op.Graph(fa, rt, 31)
    pg = api.Graph(fa, rt, 31, upload=False)
    info = pg.info()
    assert info.n_unitigs == og.n_unitigs and info.n_kmers == og.n_kmers and info.max_km_cov_top == og.max_km_cov_top
    assert info.n_global_sets > 0, "dataset should exercise the global/local colour split (G2)"
    assert pg.opts().max_km_cov == max(128, og.max_km_cov_top)


def test_no_compute_without_upload(ds_small):
    pg = api.Graph(ds_small + ".index.k31.fasta.gz", ds_small + ".index.k31.rtsk", 31, upload=False)
    try:
        pg.lookup_exact("A" * 40)
        assert False
    except api.RtkError as e:
        assert "not resident" in str(e)


# ---- the lookup structures built in HBM (RTK_LOAD_DEVICE_TABLES, csrc/hip/rtk_graph_tables.hip) against the host's (the definition) ----
_BUFS = ["useq", "uoff", "adj", "flags", "kcov", "card", "loff", "gid", "goff", "col", "ht", "bf", "cycoff", "cyc", "bf1", "amb", "hx", "hxl", "hap"]
_TABLES = {"ht", "bf", "bf1", "hx", "hxl", "adj"}


def _host_buffer(g, name):
    import numpy as np
    p, b = C.c_void_p(), C.c_uint64()
    g._check(g.L.rtk_graph_host_buffer(g.h, _BUFS.index(name), C.byref(p), C.byref(b)))
    return np.frombuffer((C.c_char * b.value).from_address(p.value), dtype=np.uint64 if name not in ("adj", "flags", "kcov", "card", "gid", "col") else np.uint32).copy()


def _device_buffer(g, name):
    import numpy as np
    p, b = C.c_void_p(), C.c_uint64()
    g._check(g.L.rtk_graph_buffer(g.h, _BUFS.index(name), C.byref(p), C.byref(b)))
    out = np.zeros(b.value // 4, dtype=np.uint32)
    g._check(g.L.rtk_graph_download_buffer(g.h, _BUFS.index(name), out.ctypes.data_as(C.c_void_p), b.value))
    return out.view(np.uint64) if name not in ("adj", "flags", "kcov", "card", "gid", "col") else out


def _load(ds, k, deferred, upload=False, device=0):
    g = api.Graph.__new__(api.Graph)
    g.L, g.k, g.h = api.load_library(None), k, C.c_void_p()
    g._check(g.L.rtk_graph_load2(api._b(ds + ".index.k%d.fasta.gz" % k), api._b(ds + ".index.k%d.rtsk" % k), k, 8, api.RTK_LOAD_DEVICE_TABLES if deferred else 0, C.byref(g.h)))
    if upload:
        g._check(g.L.rtk_graph_upload(g.h, device))
    return g


import pytest  # noqa: E402


@pytest.mark.parametrize("which", ["ds_small", "ds_k21", "ds_snps", "ds_tandem"])
def test_deferred_load_gives_the_same_unitig_data(which, request):
    """the .rtsk records find their unitigs through the extremity table exactly as through the table of all k-mers: every buffer that is not a lookup
    structure is the same bytes, and the info (table slots included: one sizing policy) agrees"""
    ds = request.getfixturevalue(which); k = 21 if which == "ds_k21" else 31
    a, b = _load(ds, k, False), _load(ds, k, True)
    for name in _BUFS:
        if name not in _TABLES:
            assert (_host_buffer(a, name) == _host_buffer(b, name)).all(), name
    ia, ib = a.info(), b.info()
    for f in ("n_unitigs", "n_kmers", "n_bases", "n_colour_ids", "n_global_sets", "table_slots", "max_km_cov_top"):
        assert getattr(ia, f) == getattr(ib, f), f
    try:
        sizes = (C.c_uint64 * len(_BUFS))()
        b._check(b.L.rtk_graph_buffer_bytes(b.h, sizes, len(_BUFS)))
        assert False, "sizes of tables that do not exist yet"
    except api.RtkError as e:
        assert "rtk_graph_upload" in str(e)


def _table_as_set(ht):
    import numpy as np
    kv = ht.reshape(-1, 2)
    kv = kv[kv[:, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)]
    return kv[np.lexsort((kv[:, 1], kv[:, 0]))]


def _check_device_tables(ds, k):
    import numpy as np
    host = _load(ds, k, False)
    dev = _load(ds, k, True, upload=True)
    for name in ("bf", "bf1", "hxl", "adj"):  # the same bytes
        h, d = _host_buffer(host, name), _device_buffer(dev, name)
        assert h.shape == d.shape and (h == d).all(), name
    # the two open-addressing tables: the same slots-count and the same entries; WHICH slot of its probe sequence an entry took depends on who claimed first
    for name in ("ht", "hx"):
        h, d = _host_buffer(host, name), _device_buffer(dev, name)
        assert h.shape == d.shape, name
    assert (_table_as_set(_host_buffer(host, "ht")) == _table_as_set(_device_buffer(dev, "ht"))).all()
    hx_h, hx_d = np.sort(_host_buffer(host, "hx")), np.sort(_device_buffer(dev, "hx"))
    assert (hx_h == hx_d).all()
    for name in _BUFS:
        if name not in _TABLES:
            assert (_host_buffer(host, name) == _device_buffer(dev, name)).all(), name
    ih, idv = host.info(), dev.info()
    assert ih.table_slots == idv.table_slots and ih.hbm_bytes == idv.hbm_bytes
    return dev


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["ds_small", "ds_k21", "ds_k25", "ds_snps_rich", "ds_tandem", "ds_medium"])
def test_gpu_device_tables_equal_host_tables(which, request):
    ds = request.getfixturevalue(which); k = {"ds_k21": 21, "ds_k25": 25}.get(which, 31)
    dev = _check_device_tables(ds, k)
    # every probe sequence ends: each k-mer of a unitig is found where it lies (a lookup through the device-built table and filters)
    import numpy as np
    useq, uoff = _host_buffer(dev, "useq"), _host_buffer(dev, "uoff")
    u = int(len(uoff) // 2); s = "".join("ACGT"[(int(useq[p >> 5]) >> (2 * (p & 31))) & 3] for p in range(int(uoff[u]), int(uoff[u + 1])))
    hits = dev.lookup_exact(s)
    assert hits == [(u << 33) | (i << 1) | 1 for i in range(len(s) - k + 1)]


@pytest.mark.gpu
def test_gpu_device_tables_in_several_sort_ranges(ds_small, monkeypatch):
    """the half-k-mer index sorted in ranges of leading h-mer bits (what a whole-genome graph does when its keys do not fit twice): same lists"""
    monkeypatch.setenv("RTK_HX_PART_KEYS", "20000")
    _check_device_tables(ds_small, 31)


@pytest.mark.gpu
def test_gpu_device_tables_two_word_kmers(tmp_path_factory):
    """k = 63 (second-pass graphs): fingerprint keys, every k-mer verified to find itself; no half-k-mer index"""
    from conftest import make_dataset
    ds = make_dataset(tmp_path_factory.mktemp("k63"), "k63", ["--seed", "5", "--ref-len", "30000", "--sr-cov", "30", "--lr-cov", "1"], ["-k", "63"])
    _check_device_tables(ds, 63)


def test_a_repeated_kmer_is_reported(tmp_path):
    """a unitig file that holds a k-mer twice is no compacted de Bruijn graph: the host build of the table says so"""
    import gzip
    seq = "ACGTTGCAAGGCTTACCGATAGGCTAACGTTAGGCATCGATCGGATTACAGGCATTAGC"
    fa = str(tmp_path / "dup.index.k31.fasta.gz")
    with gzip.open(fa, "wt") as f:
        f.write(">0\n%s\n>1\n%s\n" % (seq, seq[3:] + "ACG"))
    rt = str(tmp_path / "dup.index.k31.rtsk"); open(rt, "wb").close()
    g = api.Graph.__new__(api.Graph); g.L, g.k, g.h = api.load_library(None), 31, C.c_void_p()
    rc = g.L.rtk_graph_load2(api._b(fa), api._b(rt), 31, 2, 0, C.byref(g.h))
    assert rc != 0 and "occurs twice" in g.L.rtk_last_error().decode()


def test_unitig_fasta_of_several_gzip_members(ds_small, tmp_path, monkeypatch):
    """the index tool writes the unitig FASTA as gzip members of 32 MB of text each (compressed, and later inflated, side by side): a file of many
    small members is the same text to zlib / Python's gzip and the same graph to the loader, on one thread and on several"""
    import gzip, os, subprocess
    from conftest import BIN
    monkeypatch.setenv("RTK_FASTA_MEMBER_BYTES", "3000")
    out = str(tmp_path / "mm")
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", ds_small + ".sr.fq", "-o", out, "--fast", "--global-cov-factor", "1.2"], stderr=subprocess.DEVNULL)
    raw = open(out + ".index.k31.fasta.gz", "rb").read()
    assert raw.count(b"\x1f\x8b\x08") >= 5
    assert gzip.decompress(raw) == gzip.open(ds_small + ".index.k31.fasta.gz", "rb").read()
    assert open(out + ".index.k31.rtsk", "rb").read() == open(ds_small + ".index.k31.rtsk", "rb").read()
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", ds_small + ".sr.fq", "-o", out + "1", "--global-cov-factor", "1.2"], stderr=subprocess.DEVNULL)  # plain path, one thread: the same members
    assert open(out + "1.index.k31.fasta.gz", "rb").read() == raw
    for threads in (1, 8):
        g = api.Graph.__new__(api.Graph); g.L, g.k, g.h = api.load_library(None), 31, C.c_void_p()
        g._check(g.L.rtk_graph_load2(api._b(out + ".index.k31.fasta.gz"), api._b(out + ".index.k31.rtsk"), 31, threads, 0, C.byref(g.h)))
        ref = _load(ds_small, 31, False)
        for name in _BUFS:
            if name not in ("ht", "hx"):
                assert (_host_buffer(g, name) == _host_buffer(ref, name)).all(), name


def _moved_into_caller_buffers(ds_small):
    """body of test_gpu_resident_graph_moved_into_caller_buffers, run in a process of its own (torch first, then the library: ratatosk_amd/dist.py's order)"""
    import torch
    torch.zeros(1, device="cuda:0")
    dev = _load(ds_small, 31, True, upload=True)
    useq, uoff = _host_buffer(dev, "useq"), _host_buffer(dev, "uoff")
    u = int(len(uoff) // 3); s = "".join("ACGT"[(int(useq[p >> 5]) >> (2 * (p & 31))) & 3] for p in range(int(uoff[u]), int(uoff[u + 1])))
    before = dev.lookup_exact(s)
    n_buf = len(_BUFS); sizes = (C.c_uint64 * n_buf)()
    dev._check(dev.L.rtk_graph_buffer_bytes(dev.h, sizes, n_buf))
    tensors = []
    for i in range(n_buf):
        tensors.append(torch.empty(max(8, int(sizes[i])), dtype=torch.uint8, device="cuda:0"))
        dev._check(dev.L.rtk_graph_move_buffer(dev.h, i, C.c_void_p(tensors[i].data_ptr()), max(8, int(sizes[i]))))
    torch.cuda.synchronize()
    assert dev.lookup_exact(s) == before == [(u << 33) | (i << 1) | 1 for i in range(len(s) - 31 + 1)]
    p, b = C.c_void_p(), C.c_uint64()
    dev._check(dev.L.rtk_graph_buffer(dev.h, _BUFS.index("ht"), C.byref(p), C.byref(b)))
    assert p.value == tensors[_BUFS.index("ht")].data_ptr()
    dev.close()  # (the tensors outlive the graph: the library must not free them)
    assert int(tensors[0][:8].sum().item()) >= 0


@pytest.mark.gpu
def test_gpu_resident_graph_moved_into_caller_buffers(ds_small):
    """what rank 0 of a multi-GPU job does before it broadcasts (ratatosk_amd/dist.py): the graph is loaded and its tables are built in the library's own
    HBM, then every flat buffer is moved into a torch tensor of its size; the graph answers as before and nothing is freed twice. In a process of its
    own: torch brings its own HIP runtime and has to be initialised before the library is loaded (as in dist.py and bench.py); in the test process the
    library of the earlier tests is there first and torch then finds no device."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path[:0] = [%r, %r]; import torch; torch.zeros(1, device='cuda:0'); import test_graph_load as T; T._moved_into_caller_buffers(%r); print('moved ok')" % (here, os.path.dirname(here), ds_small)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "moved ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
