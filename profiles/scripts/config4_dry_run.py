"""configs[4] dry run (BASELINE.json: whole-genome-scale graph resident in HBM): a synthetic reference of REF_MB megabases (3 % two-copy
repeats) through the repo's own tools -> index files -> rtk_graph_load (threaded) -> rtk_graph_upload -> one batch of long reads.
Reports sizes and times; anything that breaks at this scale (32-bit offsets, single-threaded steps, HBM footprint) shows up here.
Usage (on the GPU box): python profiles/scripts/config4_dry_run.py [REF_MB=500] [SR_COV=12] [THREADS=64]"""
import ctypes as C, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ratatosk_amd import api
ref_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 500
sr_cov = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 64
wd = tempfile.mkdtemp(prefix="rtk_c4_", dir=os.environ.get("RTK_C4_DIR", "/tmp"))
pre = os.path.join(wd, "c4")
bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin")
out = {"ref_mb": ref_mb, "sr_cov": sr_cov, "threads": threads}
def save():  # partial results survive a failure further down
    if os.environ.get("RTK_C4_OUT"):
        json.dump(out, open(os.environ["RTK_C4_OUT"], "w"))
t0 = time.time()
subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre, "--seed", "5", "--ref-len", str(ref_mb * 1000000), "--repeat-frac", "0.03", "--sr-cov", str(sr_cov), "--sr-err", "0.005",
                       "--lr-cov", "%.4f" % (100.0 / ref_mb), "--lr-len", "8000", "--lr-profile", "ont", "--lr-err", "0.07", "--lr-truth"], stderr=subprocess.DEVNULL)
out["simulate_s"] = round(time.time() - t0, 1); out["sr_fastq_gb"] = round(os.path.getsize(pre + ".sr.fq") / 1e9, 2)
t0 = time.time()
r = subprocess.run([os.path.join(bin_dir, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre, "--gpu"], stderr=subprocess.PIPE, text=True, check=True, env=dict(os.environ, RTK_INDEX_TRACE="1"))  # k-mers counted on the device, the other steps on the host threads (same files as the plain tool)
out["build_index_s"] = round(time.time() - t0, 1); out["build_index_log"] = r.stderr.strip().splitlines()
os.remove(pre + ".sr.fq"); save()
out["host_ram_gb"] = round(os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9)
fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
out["index_files_gb"] = {"fasta.gz": round(os.path.getsize(fa) / 1e9, 3), "rtsk": round(os.path.getsize(rt) / 1e9, 3)}
L = api.load_library()
h = C.c_void_p()
t0 = time.time(); rc = L.rtk_graph_load(fa.encode(), rt.encode(), 31, threads, C.byref(h)); out["graph_load_s"] = round(time.time() - t0, 1)
assert rc == 0, L.rtk_last_error()
t0 = time.time(); rc = L.rtk_graph_upload(h, 0); out["graph_upload_s"] = round(time.time() - t0, 1)
assert rc == 0, L.rtk_last_error()
g = api.Graph.__new__(api.Graph); g.L, g.k, g.h = L, 31, h
info = g.info()
sizes = (C.c_uint64 * L.rtk_graph_n_buffers(None))(); L.rtk_graph_buffer_bytes(h, sizes, len(sizes))
names = ["useq", "uoff", "adj", "flags", "kcov", "card", "loff", "gid", "goff", "col", "ht", "bf", "cycoff", "cyc", "bf1", "amb", "hx", "hxl"]
out["graph"] = {"unitigs": int(info.n_unitigs), "kmers": int(info.n_kmers), "colour_ids": int(info.n_colour_ids), "hbm_gb": round(info.hbm_bytes / 1e9, 2),
                "buffers_gb": {n: round(sizes[i] / 1e9, 3) for i, n in enumerate(names)}}
save()
import bench
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 64_000_000)
b = api.Batch(g, seqs, quals)
t0 = time.time(); b.run(); out["first_batch_run_s"] = round(time.time() - t0, 2)
t0 = time.time(); b.run(); out["second_batch_run_s"] = round(time.time() - t0, 3)
st = b.stats()
out["batch"] = {"reads": len(seqs), "bases": st["in_bases"], "kernel_ms": {k_: round(st[k_], 2) for k_ in ("ms_lookup_exact", "ms_mask", "ms_lookup_inexact", "ms_seeds", "ms_correct", "ms_stitch", "ms_total")},
                "regions": st["n_regions"], "bases_per_s_kernels": round(st["in_bases"] / (st["ms_total"] * 1e-3))}
got = b.fetch(); save()
out["batch"]["reads_changed"] = sum(1 for s, g_ in zip(seqs, got) if s != g_[0])
# ---- size-independent checks (the oracle cannot hold a graph of this size): (1) a corrected read is closer to the stretch of the reference it was
# simulated from than the raw read was (edit distances by the device's own banded NW, itself held to the reference's edlib by the test tiers);
# (2) the share of k-mer windows of a read found in the graph goes up
def rc(x):
    return x[::-1].translate(str.maketrans("ACGT", "TGCA"))
ref = {}
with open(pre + ".ref.fa") as f:
    name = None
    for line in f:
        if line.startswith(">"):
            name = len(ref); ref[name] = []
        else:
            ref[name].append(line.strip())
ref = {k_: "".join(v) for k_, v in ref.items()}
truth = [l.split("\t") for l in open(pre + ".lr.truth.tsv").read().splitlines()]
n_chk = min(400, len(seqs))
tr = []
for i in range(n_chk):
    _, hap, start, ln, strand = truth[i]
    t_ = ref[int(hap)][int(start):int(start) + int(ln)]
    tr.append(rc(t_) if strand == "-" else t_)
d_raw = [r_[0] for r_ in api.myers_batch(seqs[:n_chk], tr, [-1] * n_chk, [0] * n_chk)]
d_cor = [r_[0] for r_ in api.myers_batch([g_[0] for g_ in got[:n_chk]], tr, [-1] * n_chk, [0] * n_chk)]
tot_len = sum(len(t_) for t_ in tr)
solid = lambda s_: sum(1 for h_ in g.lookup_exact(s_.upper()) if h_ != -1) / max(1, len(s_) - 30)
sol_raw = sum(solid(s_) for s_ in seqs[:50]) / 50; sol_cor = sum(solid(g_[0]) for g_ in got[:50]) / 50
out["property_checks"] = {"reads_checked": n_chk, "error_rate_raw": round(sum(d_raw) / tot_len, 4), "error_rate_corrected": round(sum(d_cor) / tot_len, 4),
                          "reads_not_closer_to_truth": sum(1 for a_, b_ in zip(d_raw, d_cor) if b_ > a_), "solid_window_share_raw": round(sol_raw, 3), "solid_window_share_corrected": round(sol_cor, 3)}
save()
assert out["property_checks"]["error_rate_corrected"] < 0.5 * out["property_checks"]["error_rate_raw"] and sol_cor > sol_raw
try:
    import torch
    free, total = torch.cuda.mem_get_info(0); out["hbm_in_use_gb_after_batch"] = round((total - free) / 1e9, 1)
except Exception as e:
    out["hbm_in_use_gb_after_batch"] = str(e)
save()
print(json.dumps(out))
