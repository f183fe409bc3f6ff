"""configs[4] of BASELINE.json at its stated size: whole-genome-scale graph (REF_MB megabases, diploid with 0.1 % heterozygous SNPs, 3 % two-copy
repeats), SR_COV x PE150 short reads SAMPLED ON THE FLY inside the index tool (`-s sample:...`: the 180 GB FASTQ of a 3 Gb x 30x set never exists),
index built with `--gpu --snps`, graph loaded and resident in HBM, TICKETS 64 Mb tickets of ONT-profile long reads corrected with three in flight.
Size-independent checks as in round 3 (the oracle cannot hold this graph): corrected reads against the stretches of the reference they were
simulated from (edit distances by the device's banded NW), share of k-mer windows found in the graph.
Usage (GPU box): python profiles/scripts/config4_run.py [REF_MB=3000] [SR_COV=30] [THREADS=128] [TICKETS=6]      (RTK_C4_OUT=file.json, RTK_C4_DIR=/tmp)
Leaves the index under $RTK_C4_DIR/c4_keep/ when RTK_C4_KEEP=1 (the file-to-file run and the profiling passes of r05_config4.sh reuse it)."""
import ctypes as C, json, os, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from ratatosk_amd import api
ref_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sr_cov = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 128
n_tickets = int(sys.argv[4]) if len(sys.argv) > 4 else 6
base = os.environ.get("RTK_C4_DIR", "/tmp")
wd = os.path.join(base, "c4_keep") if os.environ.get("RTK_C4_KEEP") else tempfile.mkdtemp(prefix="rtk_c4_", dir=base)
os.makedirs(wd, exist_ok=True)
pre = os.path.join(wd, "c4")
bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin")
out = {"ref_mb": ref_mb, "sr_cov": sr_cov, "threads": threads, "what": "configs[4]: whole-genome-scale graph resident in HBM; short reads sampled on the fly inside the index tool"}
def save():
    if os.environ.get("RTK_C4_OUT"):
        json.dump(out, open(os.environ["RTK_C4_OUT"], "w"), indent=1)
lr_bases = n_tickets * 64_000_000 + 2_000_000
t0 = time.time()
subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre, "--seed", "5", "--ref-len", str(ref_mb * 1000000), "--het", "0.001", "--repeat-frac", "0.03", "--sr-cov", "0",
                       "--lr-cov", "%.5f" % (lr_bases / (ref_mb * 1e6)), "--lr-len", "8000", "--lr-profile", "ont", "--lr-err", "0.07", "--lr-truth"], stderr=subprocess.DEVNULL)
out["simulate_s"] = round(time.time() - t0, 1); out["ref_fasta_gb"] = round(os.path.getsize(pre + ".ref.fa") / 1e9, 2); save()
assert abs(os.path.getsize(pre + ".ref.fa") - 2 * ref_mb * 1e6) < 0.01 * 2 * ref_mb * 1e6, "reference FASTA of an unexpected size"
spec = "sample:%s.ref.fa?cov=%g&len=150&insert=400&err=0.005&seed=7" % (pre, sr_cov)
out["short_reads"] = {"source": spec.replace(wd, "$WD"), "bases": int(sr_cov * ref_mb * 1e6), "fastq_bytes_never_written": int(2 * sr_cov * ref_mb * 1e6 * (150 + 150 + 12) / 300)}
t0 = time.time()
idx_log = os.environ.get("RTK_C4_INDEX_LOG", os.path.join(wd, "build_index.log"))  # (written as the tool goes: a run that is cut short still shows where it was)
# The container's control group caps host memory (300 GiB on the GPU boxes of this pool, whatever the machine has): its use is sampled while the tool runs, the
# tool is stopped before the limit is (a box that runs out of memory is lost), and the peak goes into the report.
def cgroup(name):
    try:
        return int(open("/sys/fs/cgroup/" + name).read().strip())
    except Exception:
        return None
mem_max = cgroup("memory.max")
snps = os.environ.get("RTK_C4_SNPS", "1" if ref_mb < 2500 else "0") == "1"
if not snps:
    out["snp_annotations"] = "left out at this size: the tool's 1-substitution neighbour index (two sorted views of all oriented k-mers, ~100 GB at 3 Gb) does not fit beside its k-mer table (137 GB) under the container's %s GB host-memory limit; the 300 Mb run of the gpu tier has them" % (round(mem_max / 1e9) if mem_max else "?")
peak = [0]
with open(idx_log, "w") as lf:
    pr = subprocess.Popen([os.path.join(bin_dir, "rtk_build_index"), "-s", spec, "-o", pre, "--gpu"] + (["--snps"] if snps else []), stderr=lf, text=True, env=dict(os.environ, RTK_INDEX_TRACE="1", RTK_INDEX_THREADS=str(threads)))
    deadline = time.time() + float(os.environ.get("RTK_C4_INDEX_TIMEOUT", "3000")); why = None
    while pr.poll() is None:
        time.sleep(1.0)
        cur = None
        try:  # anonymous memory: what cannot be given back (the page cache of the reference and of the output files can)
            cur = [int(l.split()[1]) for l in open("/sys/fs/cgroup/memory.stat") if l.startswith("anon ")][0]
        except Exception:
            cur = cgroup("memory.current")
        if cur: peak[0] = max(peak[0], cur)
        if mem_max and cur and cur > 0.85 * mem_max: why = "host memory at %.0f of %.0f GB" % (cur / 1e9, mem_max / 1e9)
        if time.time() > deadline: why = "time limit"
        if why:
            pr.kill(); pr.wait(); break
out["build_index_s"] = round(time.time() - t0, 1); out["build_index_log"] = open(idx_log).read().strip().splitlines()[-60:]; out["build_index_peak_host_gb"] = round(peak[0] / 1e9, 1); out["host_memory_limit_gb"] = round(mem_max / 1e9, 1) if mem_max else None; save()
assert why is None and pr.returncode == 0, (why, out["build_index_log"][-5:])
out["box"] = {"host_ram_gb": round(os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9), "cpus": os.cpu_count()}
fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
out["index_files_gb"] = {"fasta.gz": round(os.path.getsize(fa) / 1e9, 3), "rtsk": round(os.path.getsize(rt) / 1e9, 3)}
L = api.load_library()
h = C.c_void_p()
host_tables = os.environ.get("RTK_HOST_TABLES") == "1"  # (the round-4 runs before the device builder: everything on the host threads)
os.environ["RTK_LOAD_TRACE"] = "1"
import io, contextlib
err_path = os.path.join(wd, "load_trace.txt"); saved_err = os.dup(2); fd = os.open(err_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC); os.dup2(fd, 2)  # (the library reports its sections on stderr)
try:
    t0 = time.time(); rc = L.rtk_graph_load2(fa.encode(), rt.encode(), 31, threads, 0 if host_tables else api.RTK_LOAD_DEVICE_TABLES, C.byref(h)); out["graph_load_s"] = round(time.time() - t0, 1)
    assert rc == 0, L.rtk_last_error()
    t0 = time.time(); rc = L.rtk_graph_upload(h, 0); out["graph_upload_s"] = round(time.time() - t0, 1)
    assert rc == 0, L.rtk_last_error()
finally:
    os.dup2(saved_err, 2); os.close(fd)
out["graph_tables"] = "host threads" if host_tables else "device (rtk_graph_tables.hip), inside graph_upload_s"
out["graph_load_trace"] = [l.strip() for l in open(err_path).read().splitlines() if l.startswith("[rtk load]")]
g = api.Graph.__new__(api.Graph); g.L, g.k, g.h = L, 31, h
info = g.info()
sizes = (C.c_uint64 * L.rtk_graph_n_buffers(None))(); L.rtk_graph_buffer_bytes(h, sizes, len(sizes))
names = ["useq", "uoff", "adj", "flags", "kcov", "card", "loff", "gid", "goff", "col", "ht", "bf", "cycoff", "cyc", "bf1", "amb", "hx", "hxl", "hap"]
out["graph"] = {"unitigs": int(info.n_unitigs), "kmers": int(info.n_kmers), "colour_ids": int(info.n_colour_ids), "hbm_gb": round(info.hbm_bytes / 1e9, 2),
                "buffers_gb": {(names[i] if i < len(names) else "buf%d" % i): round(sizes[i] / 1e9, 3) for i in range(len(sizes))}}
save()
import bench
seqs, quals = bench.read_long_reads(pre + ".lr.fq", n_tickets * 64_000_000)
tickets, cs, cq, cur = [], [], [], 0
for s_, q_ in zip(seqs, quals):
    cs.append(s_); cq.append(q_); cur += len(s_)
    if cur >= 64_000_000:
        tickets.append((cs, cq)); cs, cq, cur = [], [], 0
opts = g.opts()
batches = [api.Batch(g, *t) for t in tickets]
t0 = time.time(); batches[0].run(opts); out["first_ticket_run_s"] = round(time.time() - t0, 2)
# all tickets, three in flight (the CLI's three workers per GPU): two host threads keep the seed stage of one ticket beside the region stage of another
t0 = time.time()
api.run_pipelined(batches, opts)  # (returns when every ticket's stream has been waited for)
dt = time.time() - t0
tot = sum(b.in_bases for b in batches)
st = [b.stats() for b in batches]
kern = {"k_lookup_exact": "ms_lookup_exact", "k_mask": "ms_mask", "k_inexact": "ms_lookup_inexact", "k_finalize": "ms_seeds", "k_regions": "ms_correct", "k_stitch": "ms_stitch"}
out["tickets"] = {"n": len(batches), "bases": tot, "seconds": round(dt, 3), "bases_per_s": round(tot / dt), "ms_per_ticket": round(1e3 * dt / len(batches), 2),
                  "kernel_ms_per_ticket_overlapped": {k_: round(sum(s_[v] for s_ in st) / len(st), 2) for k_, v in kern.items()}, "regions_per_ticket": int(sum(s_["n_regions"] for s_ in st) / len(st))}
b = batches[0]; b.run(opts); s1 = b.stats()
out["tickets"]["kernel_ms_one_ticket_alone"] = {k_: round(s1[v], 2) for k_, v in kern.items()}
got = b.fetch(); seqs0 = tickets[0][0]; save()
fr, tt = C.c_uint64(), C.c_uint64()
if L.rtk_device_memory(0, C.byref(fr), C.byref(tt)) == 0:
    out["hbm_in_use_gb_with_%d_tickets" % len(batches)] = round((tt.value - fr.value) / 1e9, 1); out["hbm_total_gb"] = round(tt.value / 1e9, 1)
def rcs(x):
    return x[::-1].translate(str.maketrans("ACGT", "TGCA"))
truth = [l.split("\t") for l in open(pre + ".lr.truth.tsv").read().splitlines()][:400]
need = {}
for i, (_, hap, start, ln, strand) in enumerate(truth):
    need.setdefault(int(hap), []).append((int(start), int(ln), strand, i))
tr = [None] * len(truth)
hap_i, pos, want = -1, 0, None
with open(pre + ".ref.fa") as f:  # one pass over the reference (6 GB at 3 Gb diploid): only the truth stretches are kept
    for line in f:
        if line.startswith(">"):
            hap_i += 1; pos = 0; want = sorted(need.get(hap_i, [])); continue
        line = line.rstrip("\n"); n = len(line)
        if want:
            for (s0, ln, strand, i) in want:
                if s0 < pos + n and s0 + ln > pos:
                    a, b_ = max(s0, pos) - pos, min(s0 + ln, pos + n) - pos
                    tr[i] = (tr[i] or "") + line[a:b_]
            want = [w for w in want if w[0] + w[1] > pos + n]
        pos += n
for i, (_, hap, start, ln, strand) in enumerate(truth):
    assert tr[i] is not None and len(tr[i]) == int(ln), (i, ln, len(tr[i] or ""))
    if strand == "-":
        tr[i] = rcs(tr[i])
n_chk = len(truth)
d_raw = [r_[0] for r_ in api.myers_batch(seqs0[:n_chk], tr, [-1] * n_chk, [0] * n_chk)]
d_cor = [r_[0] for r_ in api.myers_batch([g_[0] for g_ in got[:n_chk]], tr, [-1] * n_chk, [0] * n_chk)]
tot_len = sum(len(t_) for t_ in tr)
solid = lambda s_: sum(1 for h_ in g.lookup_exact(s_.upper()) if h_ != -1) / max(1, len(s_) - 30)
sol_raw = sum(solid(s_) for s_ in seqs0[:50]) / 50; sol_cor = sum(solid(g_[0]) for g_ in got[:50]) / 50
out["property_checks"] = {"reads_checked": n_chk, "error_rate_raw": round(sum(d_raw) / tot_len, 4), "error_rate_corrected": round(sum(d_cor) / tot_len, 4),
                          "reads_not_closer_to_truth": sum(1 for a_, b_ in zip(d_raw, d_cor) if b_ > a_), "solid_window_share_raw": round(sol_raw, 3), "solid_window_share_corrected": round(sol_cor, 3)}
save()
assert out["property_checks"]["error_rate_corrected"] < 0.5 * out["property_checks"]["error_rate_raw"] and sol_cor > sol_raw
print(json.dumps(out))
