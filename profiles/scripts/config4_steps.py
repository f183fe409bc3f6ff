"""Loads the index config4_run.py left under $RTK_C4_DIR/c4_keep and runs N tickets of 64 Mb (one at a time): the command the rocprofv3 passes of
r05_config4.sh wrap. With RTK_C4_ROOFLINE_OUT=file.json it also writes, per kernel of the ticket, the algorithmic bytes (event counters of the library, same
formulas as bench.py), the HIP-event time and the fraction of the 8 TB/s peak at this graph size.
Usage: python profiles/scripts/config4_steps.py [N=3] [THREADS=128]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from ratatosk_amd import api
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 128
pre = os.path.join(os.environ.get("RTK_C4_DIR", "/tmp"), "c4_keep", "c4")
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, n_threads=threads)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 64_000_000)
b = api.Batch(g, seqs, quals)
sts = []
for i in range(n + 1):
    b.run(g.opts())
    if i: sts.append(b.stats())
print("ran", n + 1, "tickets of", b.in_bases, "bases;", {k: round(v, 2) for k, v in b.stats().items() if k.startswith("ms_")})
out_fn = os.environ.get("RTK_C4_ROOFLINE_OUT")
if out_fn:
    S = lambda key: sum(s[key] for s in sts) / len(sts)
    alg = {"k_lookup_exact": 8.0 * S("n_probes_exact") + 16.0 * S("n_slots_exact") + 9.0 * S("in_bases"),
           "k_inexact": 16.0 * S("n_slots_inexact") + 1.0 * S("in_bases") + 16.0 * S("n_hits_inexact"),
           "k_regions": 40.0 * S("n_expand") + 4.0 * S("n_colour_elem") + 0.25 * S("n_path_base") + 4.0 * S("in_bases"),
           "k_mask": 9.0 * S("in_bases"), "k_finalize": 12.0 * S("in_bases") + 16.0 * S("n_hits_inexact"), "k_stitch": 4.0 * S("out_bases")}
    ms = {"k_lookup_exact": S("ms_lookup_exact"), "k_mask": S("ms_mask"), "k_inexact": S("ms_lookup_inexact"), "k_finalize": S("ms_seeds"), "k_regions": S("ms_correct"), "k_stitch": S("ms_stitch")}
    info = g.info()
    fr, tt = C.c_uint64(), C.c_uint64(); g.L.rtk_device_memory(0, C.byref(fr), C.byref(tt))
    json.dump({"what": "configs[4]: per-kernel roofline of one 64 Mb ticket on the resident whole-genome-scale graph (one ticket at a time, HIP events inside the library)",
               "graph": {"unitigs": int(info.n_unitigs), "kmers": int(info.n_kmers), "hbm_gb": round(info.hbm_bytes / 1e9, 2)}, "ticket_bases": int(S("in_bases")), "tickets_averaged": len(sts),
               "hbm_in_use_gb": round((tt.value - fr.value) / 1e9, 1), "peak_GBs": 8000.0,
               "kernels": {k: {"ms": round(ms[k], 3), "alg_bytes": int(alg[k]), "achieved_GBs": round(alg[k] / (ms[k] * 1e-3) / 1e9, 2) if ms[k] > 0 else 0.0, "frac": round(alg[k] / (ms[k] * 1e-3) / 1e9 / 8000.0, 5) if ms[k] > 0 else 0.0} for k in alg},
               "ms_total": round(S("ms_total"), 2), "events_per_ticket": {k: int(S(k)) for k in ("n_regions", "n_expand", "n_colour_elem", "n_path_base", "n_probes_inexact", "n_slots_inexact", "n_hits_inexact", "n_align")},
               "alg_bytes_are": "bench.py's formulas (DESIGN_HISTORY.md section 5); k_inexact: 16 B per index slot visited or candidate checked (two list words), 1 B per read base, 16 B per hit"},
              open(out_fn, "w"), indent=1)
