"""configs[4] of BASELINE.json as one leg of the bench line (bench.py runs this file as a subprocess at N = 1; it also runs on its own):
whole-genome-scale graph (REF_MB megabases diploid, 0.1 % heterozygous SNPs, 3 % two-copy repeats; 30x PE150 short reads SAMPLED ON THE FLY inside the
index tool, `-s sample:...`: the 180 GB of FASTQ of a 3 Gb x 30x set never exist), plain index (no SNP annotations: `index -F`), graph built into HBM by
the device table builder and RESIDENT there, then TICKETS tickets of 64 Mb of DISTINCT ONT-profile long reads (every ticket other reads, simulated from the
same genome: a ticket touches other pages of the 160 GB of tables than the one before it) corrected with the seed stage of one beside the region stage
of the other. Reported: resident GB by buffer, bases/s over the tickets, per-kernel ms / algorithmic bytes / fraction of the 8 TB/s peak (one ticket at a
time, HIP events inside the library, bench.py's formulas), HBM in use, and size-independent checks (the oracle cannot hold this graph): the corrected
reads against the stretches of the reference they were simulated from (edit distances by the device's NW kernel), the share of k-mer windows found in
the graph before / after.

Usage: python bench_config4.py OUT.json [REF_MB=3000] [SR_COV=30] [TICKETS=16] [THREADS=128] [WORKDIR=/tmp/rtk_c4]
Guards (every one ends with {"skipped": reason} in OUT.json, exit code 0): host memory the container may use, free disk, HBM of device 0, time."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
KERN = {"k_lookup_exact": "ms_lookup_exact", "k_mask": "ms_mask", "k_inexact": "ms_lookup_inexact", "k_finalize": "ms_seeds", "k_regions": "ms_correct", "k_stitch": "ms_stitch"}


def cgroup(name):
    try:
        v = open("/sys/fs/cgroup/" + name).read().strip()
        return None if v == "max" else int(v)
    except Exception:
        return None


def host_memory_limit():
    lim = cgroup("memory.max")
    try:
        avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable:")][0]
    except Exception:
        avail = None
    cands = [x for x in (lim, avail) if x]
    return min(cands) if cands else None


def main():
    out_fn = sys.argv[1]
    ref_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    sr_cov = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    n_tickets = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    threads = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    wd = sys.argv[6] if len(sys.argv) > 6 else "/tmp/rtk_c4"
    index_timeout = float(os.environ.get("RTK_C4_INDEX_TIMEOUT", "900"))
    out = {"workload": "configs[4]: %d Mb diploid reference (0.1 %% het SNPs, 3 %% two-copy repeats), %gx PE150 short reads sampled inside the index tool, plain index, graph resident in HBM, "
                       "%d tickets of 64 Mb of distinct ONT-profile long reads" % (ref_mb, sr_cov, n_tickets)}

    def save():
        json.dump(out, open(out_fn, "w"), indent=1)

    def skip(why):
        out["skipped"] = why; save(); print(json.dumps(out)); sys.exit(0)

    # ---- guards (sized from the round-5 run at 3 Gb: 211 GB of host memory at the peak of the index build, 6 GB reference + 12 GB index + 2 GB per Gb of reads on disk, 272 GB of HBM in use)
    scale = ref_mb / 3000.0
    need_host, need_disk, need_hbm = 235e9 * scale + 8e9, (20e9 * scale + 2.2 * n_tickets * 64e6 + 4e9), 150e9 * scale + 60e9
    lim = host_memory_limit()
    if lim is not None and lim < need_host:
        skip("host memory: %.0f GB usable by this container, the index build of a %d Mb reference needs ~%.0f GB" % (lim / 1e9, ref_mb, need_host / 1e9))
    os.makedirs(wd, exist_ok=True)
    st = os.statvfs(wd)
    if st.f_bavail * st.f_frsize < need_disk:
        skip("disk: %.0f GB free under %s, ~%.0f GB needed (reference, index files, long reads)" % (st.f_bavail * st.f_frsize / 1e9, wd, need_disk / 1e9))
    from ratatosk_amd import api
    sim = os.environ.get("RTK_C4_SIM") == "1"  # CPU dry run of this script's own logic on the developer simulator (tests/test_bench_contract.py); never a measurement
    lib_path = os.path.join(ROOT, "tests", "hostsim", "librtk_hostsim.so") if sim else None
    L = api.load_library(lib_path)
    fr, tt = C.c_uint64(), C.c_uint64()
    if L.rtk_device_memory(0, C.byref(fr), C.byref(tt)) != 0:
        skip("no HIP device")
    if fr.value < need_hbm and not sim:
        skip("HBM: %.0f GB free on device 0, ~%.0f GB needed" % (fr.value / 1e9, need_hbm / 1e9))
    out["guards"] = {"host_memory_usable_gb": round(lim / 1e9, 1) if lim else None, "disk_free_gb": round(st.f_bavail * st.f_frsize / 1e9, 1), "hbm_free_gb": round(fr.value / 1e9, 1)}

    pre = os.path.join(wd, "c4")
    bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin")
    lr_bases = n_tickets * 64_000_000 + 2_000_000
    stamp = json.dumps({"ref_mb": ref_mb, "sr_cov": sr_cov, "lr_bases": lr_bases}, sort_keys=True)
    fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
    reused = False
    try:
        reused = open(pre + ".stamp.json").read() == stamp and all(os.path.exists(p) for p in (fa, rt, pre + ".lr.fq", pre + ".ref.fa", pre + ".lr.truth.tsv"))
    except OSError:
        pass
    if not reused:
        t0 = time.time()
        subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre, "--seed", "5", "--ref-len", str(ref_mb * 1000000), "--het", "0.001", "--repeat-frac", "0.03", "--sr-cov", "0",
                               "--lr-cov", "%.6f" % (lr_bases / (ref_mb * 1e6)), "--lr-len", "8000", "--lr-profile", "ont", "--lr-err", "0.07", "--lr-truth"], stderr=subprocess.DEVNULL)
        out["simulate_s"] = round(time.time() - t0, 1)
        spec = "sample:%s.ref.fa?cov=%g&len=150&insert=400&err=0.005&seed=7" % (pre, sr_cov)
        t0 = time.time()
        idx_log = pre + ".build_index.log"
        mem_max = cgroup("memory.max")
        peak, why = 0, None
        with open(idx_log, "w") as lf:
            pr = subprocess.Popen([os.path.join(bin_dir, "rtk_build_index"), "-s", spec, "-o", pre] + ([] if sim else ["--gpu"]), stderr=lf, text=True, env=dict(os.environ, RTK_INDEX_TRACE="1", RTK_INDEX_THREADS=str(threads)))
            deadline = time.time() + index_timeout
            while pr.poll() is None:  # the tool is stopped before the container's memory limit is (a box that runs out of memory is lost)
                time.sleep(1.0)
                try:
                    cur = [int(l.split()[1]) for l in open("/sys/fs/cgroup/memory.stat") if l.startswith("anon ")][0]
                except Exception:
                    cur = cgroup("memory.current")
                if cur:
                    peak = max(peak, cur)
                if mem_max and cur and cur > 0.88 * mem_max:
                    why = "host memory at %.0f of %.0f GB during the index build" % (cur / 1e9, mem_max / 1e9)
                if time.time() > deadline:
                    why = "index build beyond its %d s limit" % index_timeout
                if why:
                    pr.kill(); pr.wait(); break
        out["build_index_s"] = round(time.time() - t0, 1); out["build_index_peak_host_gb"] = round(peak / 1e9, 1)
        if why or pr.returncode != 0:
            out["build_index_log_tail"] = open(idx_log).read().strip().splitlines()[-8:]
            skip(why or ("rtk_build_index failed (rc %d)" % pr.returncode))
        open(pre + ".stamp.json", "w").write(stamp)
    else:
        out["index"] = "reused from " + wd
    out["index_files_gb"] = {"fasta.gz": round(os.path.getsize(fa) / 1e9, 3), "rtsk": round(os.path.getsize(rt) / 1e9, 3)}
    save()

    # ---- graph: parsed on the host threads, lookup structures built on the device, resident in HBM
    h = C.c_void_p()
    t0 = time.time(); rc = L.rtk_graph_load2(fa.encode(), rt.encode(), 31, threads, 0 if sim else api.RTK_LOAD_DEVICE_TABLES, C.byref(h)); out["graph_load_s"] = round(time.time() - t0, 1)
    if rc != 0:
        skip("rtk_graph_load2: " + L.rtk_last_error().decode())
    t0 = time.time(); rc = L.rtk_graph_upload(h, 0); out["graph_upload_and_device_tables_s"] = round(time.time() - t0, 1)
    if rc != 0:
        skip("rtk_graph_upload: " + L.rtk_last_error().decode())
    g = api.Graph.__new__(api.Graph); g.L, g.k, g.h = L, 31, h; g.lib_path = lib_path
    info = g.info()
    sizes = (C.c_uint64 * L.rtk_graph_n_buffers(None))(); L.rtk_graph_buffer_bytes(h, sizes, len(sizes))
    names = ["useq", "uoff", "adj", "flags", "kcov", "card", "loff", "gid", "goff", "col", "ht", "bf", "cycoff", "cyc", "bf1", "amb", "hx", "hxl", "hap"]
    out["graph"] = {"unitigs": int(info.n_unitigs), "kmers": int(info.n_kmers), "colour_ids": int(info.n_colour_ids), "hbm_gb": round(info.hbm_bytes / 1e9, 2),
                    "buffers_gb": {(names[i] if i < len(names) else "buf%d" % i): round(sizes[i] / 1e9, 3) for i in range(len(sizes)) if sizes[i] >= 5e7}}
    save()

    # ---- tickets of distinct reads
    import bench
    seqs, quals = bench.read_long_reads(pre + ".lr.fq", n_tickets * 64_000_000)
    tickets, cs, cq, cur = [], [], [], 0
    for s_, q_ in zip(seqs, quals):
        cs.append(s_); cq.append(q_); cur += len(s_)
        if cur >= 64_000_000:
            tickets.append((cs, cq)); cs, cq, cur = [], [], 0
    opts = g.opts()
    in_flight = 5 if len(tickets) >= 10 else max(1, len(tickets) // 2)  # resident at a time (as the CLI keeps them): a ticket's buffers are ~60 B per base
    t0 = time.time(); b0 = api.Batch(g, *tickets[0]); b0.run(opts); out["first_ticket_s"] = round(time.time() - t0, 2)
    got0 = b0.fetch(); b0.close()
    done_b, dt_all, stats = 0, 0.0, []
    for c0 in range(0, len(tickets), in_flight):
        batches = [api.Batch(g, *t) for t in tickets[c0:c0 + in_flight]]  # (H2D outside the clock: `value` of the bench line is defined that way too)
        t0 = time.time(); api.run_pipelined(batches, opts); dt = time.time() - t0
        if c0 == 0:  # the first group is not timed: its tickets allocate the batch buffers (hipMalloc of hundreds of MB stalls the device) that the later groups take from the library's pool
            out["first_group_s_with_allocations"] = round(dt, 3)
            if L.rtk_device_memory(0, C.byref(fr), C.byref(tt)) == 0:
                out["hbm_in_use_gb_with_%d_tickets_resident" % len(batches)] = round((tt.value - fr.value) / 1e9, 1); out["hbm_total_gb"] = round(tt.value / 1e9, 1)
        else:
            dt_all += dt; done_b += sum(b.in_bases for b in batches); stats += [b.stats() for b in batches]
        for b in batches:
            b.close()
    out["tickets"] = {"n": len(stats), "distinct": True, "bases": int(done_b), "seconds": round(dt_all, 3), "bases_per_s": round(done_b / dt_all) if dt_all > 0 else 0, "ms_per_ticket": round(1e3 * dt_all / max(1, len(stats)), 2),
                      "regions_per_ticket": int(sum(s_["n_regions"] for s_ in stats) / max(1, len(stats))),
                      "how": "groups of %d resident tickets (after one untimed group), every ticket other reads and run ONCE, seed stage of one beside the region stage of the other (api.run_pipelined); kernels only, inputs in HBM (like `value`)" % in_flight}
    # per kernel, one ticket at a time, three DIFFERENT tickets (each runs once: nothing of it is in a cache)
    sts = []
    for t in tickets[1:4]:
        b = api.Batch(g, *t); b.run(opts); sts.append(b.stats()); b.close()
    S = lambda key: sum(s[key] for s in sts) / len(sts)
    alg = {"k_lookup_exact": 8.0 * S("n_probes_exact") + 16.0 * S("n_slots_exact") + 9.0 * S("in_bases"),
           "k_inexact": 16.0 * S("n_slots_inexact") + 1.0 * S("in_bases") + 16.0 * S("n_hits_inexact"),
           "k_regions": 40.0 * S("n_expand") + 4.0 * S("n_colour_elem") + 0.25 * S("n_path_base") + 4.0 * S("in_bases"),
           "k_mask": 9.0 * S("in_bases"), "k_finalize": 12.0 * S("in_bases") + 16.0 * S("n_hits_inexact"), "k_stitch": 4.0 * S("out_bases")}
    out["kernels_one_ticket_alone"] = {k: {"ms": round(S(v), 3), "alg_bytes": int(alg[k]), "achieved_GBs": round(alg[k] / (S(v) * 1e-3) / 1e9, 2) if S(v) > 0 else 0.0,
                                            "frac": round(alg[k] / (S(v) * 1e-3) / 1e9 / 8000.0, 5) if S(v) > 0 else 0.0} for k, v in KERN.items()}
    out["kernels_note"] = "averages over three different tickets, each run once; alg_bytes: bench.py's formulas; frac against the 8 TB/s HBM peak"
    save()

    # ---- size-independent checks on the first ticket
    def rcs(x):
        return x[::-1].translate(str.maketrans("ACGT", "TGCA"))
    n_chk = 200
    truth = [l.split("\t") for l in open(pre + ".lr.truth.tsv").read().splitlines()][:n_chk]
    hap_off, pos = [], 0
    with open(pre + ".ref.fa", "rb") as f:  # every haplotype is ONE line: the truth stretches are read by offset
        while True:
            f.seek(pos); hdr = f.readline()
            if not hdr or not hdr.startswith(b">"):
                break
            hap_off.append(pos + len(hdr)); pos = pos + len(hdr) + ref_mb * 1000000 + 1
        tr = []
        for (_, hap, start, ln, strand) in truth:
            f.seek(hap_off[int(hap)] + int(start)); s_ = f.read(int(ln)).decode()
            tr.append(rcs(s_) if strand.strip() == "-" else s_)
    seqs0 = tickets[0][0]
    d_raw = [r_[0] for r_ in api.myers_batch(seqs0[:n_chk], tr, [-1] * n_chk, [0] * n_chk, lib_path=lib_path)]
    d_cor = [r_[0] for r_ in api.myers_batch([g_[0] for g_ in got0[:n_chk]], tr, [-1] * n_chk, [0] * n_chk, lib_path=lib_path)]
    tot_len = sum(len(t_) for t_ in tr)
    solid = lambda s_: sum(1 for h_ in g.lookup_exact(s_.upper()) if h_ != -1) / max(1, len(s_) - 30)
    sol_raw = sum(solid(s_) for s_ in seqs0[:50]) / 50; sol_cor = sum(solid(g_[0]) for g_ in got0[:50]) / 50
    out["property_checks"] = {"reads_checked": n_chk, "error_rate_raw": round(sum(d_raw) / tot_len, 4), "error_rate_corrected": round(sum(d_cor) / tot_len, 4),
                              "reads_not_closer_to_truth": sum(1 for a_, b_ in zip(d_raw, d_cor) if b_ > a_), "solid_window_share_raw": round(sol_raw, 3), "solid_window_share_corrected": round(sol_cor, 3),
                              "ok": bool(sum(d_cor) < 0.5 * sum(d_raw) and sol_cor > sol_raw)}
    save()
    g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
