#!/usr/bin/env python
"""bench.py -- corrected long-read bases/sec of the MI355X-native `Ratatosk correct -1` hot path.

A "step" is one pass of the whole hot path (exact k-mer scan -> mask -> 1-edit k-mer scan -> anchor filters -> region
enumeration -> colour-guided BFS/DFS + Myers scoring -> stitch) over one batch of synthetic long reads that is already
resident in HBM when the timed region starts (rtk_batch_create is outside, rtk_batch_run inside).

Workload at every N: the HG002-chr20-scale set BASELINE.json's target is written on (configs[2]'s graph: 60 Mb diploid random
reference, 0.1 % heterozygous SNPs, 30x PE 150 bp short reads at 0.5 % substitutions, ONT-R9.4-profile long reads, k = 31 first
pass; it fits one GPU). The index is built by the repo's own index producer (k-mers counted on the device), correction runs on
the GPU. At N = 1 the same run also measures configs[1] (5 Mb haploid reference) as the extra leg `config1`.

N > 1 (torch.distributed.run, one rank per GPU, RCCL): rank 0 loads the flat graph and broadcasts its buffers once; long
reads are sharded by batch ticket (rank r takes batches r, r+N, ...), no collective on the data path -> weak scaling. Before
the sharded run rank 0 measures the one-GPU rate on this very graph in the same process (`config.n1_on_this_graph`).

Prints ONE JSON line on rank 0 (see the driver contract), with `roofline` for the dominant kernel (algorithmic bytes /
HIP-event duration measured here) and `cpu_baseline` (the oracle, multithreaded, on a bounded sample; N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before torch / HIP initialise: streams of several tickets need hardware queues of their own (ROCm default: 4)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--serial", action="store_true", help="(the default since round 4) run the two stages of every step back to back")
    ap.add_argument("--overlap", action="store_true", help="software-pipeline consecutive steps on two streams (stage A of step s+1 beside stage B of step s): the default of rounds 1-3; "
                                                           "measured slower in round 4 (56.8 vs 54.9 ms per step on the 60 Mb set, 38.8 vs 38.0 on configs[1]): the seed kernels of the next step only take "
                                                           "wave slots from the region kernel, whose waves are all busy")
    ap.add_argument("--ref-len", type=int, default=60_000_000, help="reference length of the main workload (default: the 60 Mb chr20-scale set of configs[2], at every N)")
    ap.add_argument("--het", type=float, default=None, help="heterozygous SNP rate of the diploid reference (default 0.001; 0 with --config1-only)")
    ap.add_argument("--config2", action="store_true", help="(accepted for older scripts: the 60 Mb set is the default workload now)")
    ap.add_argument("--config1-only", action="store_true", help="main workload = configs[1] (5 Mb haploid reference): the quick developer line, A/B runs of kernel builds")
    ap.add_argument("--no-config1-leg", action="store_true", help="N = 1: skip the extra configs[1] measurement")
    ap.add_argument("--batch-bases", type=int, default=64_000_000, help="long-read bases per step (per GPU)")
    ap.add_argument("--cpu-sample-bases", type=int, default=16_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the host-inclusive legs (C ABI with host buffers, CLI file to file); N = 1 only anyway")
    ap.add_argument("--host-tickets", type=int, default=9, help="tickets of the host-inclusive C-ABI leg")
    ap.add_argument("--host-callers", type=int, default=3, help="host threads calling create + run + fetch concurrently")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--plain-index", action="store_true", help="index without SNP annotations (like the reference's `index -F`); for A/B measurements only")
    ap.add_argument("--no-config4", action="store_true", help="N = 1: skip the configs[4] leg (3 Gb graph resident in HBM, bench_config4.py: ~7 minutes, most of it the index build)")
    ap.add_argument("--config4-ref-mb", type=int, default=3000, help="reference length of the configs[4] leg in Mb (its stated size: 3000)")
    ap.add_argument("--config4-tickets", type=int, default=16, help="tickets of 64 Mb of distinct long reads of the configs[4] leg")
    ap.add_argument("--sim", action="store_true", help="CPU-only developer simulator + gloo (tests of the N>1 plumbing); never a benchmark")
    a = ap.parse_args()
    a.serial = not a.overlap
    if a.config1_only:
        a.ref_len = 5_000_000 if a.ref_len == 60_000_000 else a.ref_len
        a.het = 0.0 if a.het is None else a.het
        a.no_config1_leg = True
    if a.het is None:
        a.het = 0.001
    return a


def make_dataset(workdir, ref_len, lr_bases, snps=True, het=0.0, fast="--gpu", name="c2"):
    """Seeded synthetic inputs + index in the reference's file formats."""
    bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin")
    pre = os.path.join(workdir, name)
    lr_cov = max(1.0, float(lr_bases) / ref_len)
    # a --workdir that already holds this very set (same generator arguments) is reused: the profiling scripts run this file a dozen times
    stamp, stamp_fn = json.dumps({"ref_len": ref_len, "lr_cov": "%.3f" % lr_cov, "snps": bool(snps), "het": "%g" % het, "name": name}, sort_keys=True), pre + ".stamp.json"
    try:
        if open(stamp_fn).read() == stamp and all(os.path.exists(pre + e) for e in (".index.k31.fasta.gz", ".index.k31.rtsk", ".lr.fq", ".sr.fq")):
            return pre
    except OSError:
        pass
    subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre, "--seed", "2", "--ref-len", str(ref_len), "--sr-cov", "30",
                           "--sr-err", "0.005", "--lr-cov", "%.3f" % lr_cov, "--lr-len", "8000", "--lr-profile", "ont", "--lr-err", "0.07"] + (["--het", "%g" % het] if het > 0 else []), stderr=subprocess.DEVNULL)
    # --snps: SNP annotations like the reference's default `index` step (detectSNPs runs unless -F, src/Ratatosk.cpp:1120-1127)
    # (--gpu: the k-mers of the short reads are counted on the device and the other heavy steps run on the host threads; the files are the plain
    # tool's byte for byte, tests/test_index_build.py. `fast` = "" under the CPU-only simulator, which has no device to count on.)
    cmd = [os.path.join(bin_dir, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre] + (["--snps"] if snps else [])
    r = subprocess.run(cmd + ([fast] if fast else []), stderr=subprocess.PIPE, text=True)
    if r.returncode != 0 and fast:
        sys.stderr.write("rtk_build_index %s failed (%s); building with the plain tool\n" % (fast, r.stderr.strip()[-200:]))
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("rtk_build_index failed: " + r.stderr[-500:])
    for line in r.stderr.splitlines():
        if "SNP annotations" in line:
            sys.stderr.write(line + "\n")
    open(stamp_fn, "w").write(stamp)
    return pre


def read_long_reads(path, max_bases):
    seqs, quals, tot = [], [], 0
    with open(path) as f:
        while tot < max_bases:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip("\n"); f.readline(); q = f.readline().rstrip("\n")
            seqs.append(s); quals.append(q); tot += len(s)
    return seqs, quals


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same workload (profiles/rNN_pmc_summary.json,
    written by profiles/scripts/profile_set.sh): 2 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 B, MI355X_MICROARCH.md)
    + WRITE_SIZE, both reported in KB. None when no profile has been collected for the current code."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")):
        m = re.search(r"r(\d+)_pmc_summary", f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None, None
    try:
        k = json.load(open(best[1]))["kernels"][kernel]
        return int(2.0 * k["fetch_bytes_per_launch_raw"] + k["write_bytes_per_launch_raw"]), os.path.relpath(best[1], ROOT)
    except Exception:
        return None, None


def host_inclusive_leg(a, api, graph, opts, tickets):
    """Host buffers in, host buffers out through the C ABI: rtk_batch_create (pack + H2D) + rtk_batch_run + rtk_batch_fetch_view
    (D2H into pinned memory) + rtk_batch_free per ticket, `host_callers` threads each owning one ticket at a time (the library overlaps
    the stages of different tickets on the device). The read pointers are marshalled before the clock starts (they ARE the host
    buffers); FASTQ parsing / formatting are in the CLI leg. Not `value`: PCIe and host packing are inside."""
    import threading
    L = graph.L
    n_t = max(1, a.host_tickets)
    packed = []
    for i in range(min(n_t, len(tickets))):
        seqs = [s.encode() for s in tickets[i][0]]
        n = len(seqs)
        packed.append((n, (C.c_char_p * n)(*seqs), (C.c_uint32 * n)(*[len(x) for x in seqs]), sum(len(x) for x in seqs), seqs))
    order = [packed[i % len(packed)] for i in range(n_t)]
    nxt, lock, err, done_bases = [0], threading.Lock(), [], [0]

    def caller():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(order) or err:
                return
            n, sa, la, nb, _ = order[i]
            h = C.c_void_p()
            rc = L.rtk_batch_create(graph.h, n, sa, None, la, C.byref(h))
            if rc == 0:
                rc = L.rtk_batch_run(h, C.byref(opts))
            pool, off, ln = C.c_char_p(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
            if rc == 0:
                rc = L.rtk_batch_fetch_view(h, C.byref(pool), C.byref(off), C.byref(ln))
            if h:
                L.rtk_batch_free(h)
            if rc != 0:
                err.append(L.rtk_last_error().decode()); return
            with lock:
                done_bases[0] += nb

    # one untimed ticket first: staging buffers and device pools of this ticket size exist afterwards
    nxt[0] = len(order) - 1; caller(); nxt[0] = 0; done_bases[0] = 0
    th = [threading.Thread(target=caller) for _ in range(max(1, a.host_callers))]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    if err:
        return {"error": err[0]}
    return {"value": done_bases[0] / dt, "unit": "bases/s", "callers": len(th), "tickets": len(order), "ticket_bases": int(done_bases[0] / max(1, len(order))),
            "what": "rtk_batch_create + rtk_batch_run + rtk_batch_fetch_view + rtk_batch_free per ticket, host buffers in / pinned host records out, PCIe inside"}


def correct_batch_leg(api, graph, opts, tickets, n_tickets, callers):
    """`callers` threads, each calling rtk_correct_batch -- the function of SURVEY.md 8(b), what a reference worker thread would call per ticket (src/Ratatosk.cpp:808-864) --
    on tickets of its own: host strings in, malloc'd strings out (released with rtk_free inside the clock). The library merges the tickets of concurrent callers
    into one launch (include/ratatosk_hip.h, revision 6); groups / tickets of the run are reported."""
    import threading
    L = graph.L
    packed = []
    for i in range(min(n_tickets, len(tickets))):
        seqs = [s.encode() for s in tickets[i][0]]
        n = len(seqs)
        packed.append((n, (C.c_char_p * n)(*seqs), (C.c_uint32 * n)(*[len(x) for x in seqs]), sum(len(x) for x in seqs), seqs))
    order = [packed[i % len(packed)] for i in range(n_tickets)]
    nxt, lock, err, done_bases = [0], threading.Lock(), [], [0]

    def caller():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(order) or err:
                return
            n, sa, la, nb, _ = order[i]
            os_, oq, ol = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_uint32 * n)()
            rc = L.rtk_correct_batch(graph.h, C.byref(opts), n, sa, None, la, os_, oq, ol)
            if rc != 0:
                err.append(L.rtk_last_error().decode()); return
            L.rtk_free_many(os_, n); L.rtk_free_many(oq, n)  # (one foreign call each: a Python loop of 2 n ctypes calls costs more than the frees)
            with lock:
                done_bases[0] += nb

    nxt[0] = len(order) - 1; caller(); nxt[0] = 0; done_bases[0] = 0  # one untimed ticket: buffers of this size exist afterwards
    g0, t0_ = C.c_uint64(), C.c_uint64(); L.rtk_coalesce_stats(graph.h, C.byref(g0), C.byref(t0_))
    th = [threading.Thread(target=caller) for _ in range(max(1, callers))]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    if err:
        return {"error": err[0]}
    g1, t1_ = C.c_uint64(), C.c_uint64(); L.rtk_coalesce_stats(graph.h, C.byref(g1), C.byref(t1_))
    return {"value": done_bases[0] / dt, "launch_groups": int(g1.value - g0.value), "tickets": int(t1_.value - t0_.value)}


def ticket_size_leg(a, api, graph, opts, mine):
    """bases/s through the C ABI by ticket size (the reference hands its workers batches of >= 1 MiB of bases, src/Common.hpp:138 / src/Ratatosk.cpp:757-772;
    INTEGRATION.md recommends a bigger buffer_sz): the reads of this rank's tickets cut into tickets of >= 1 / 4 / 16 / 64 Mi bases, each size run with one
    caller and with three (host buffers in, pinned host records out, PCIe inside: the host_inclusive leg at other ticket sizes)."""
    import types
    reads = [(s_, q_) for t in mine for s_, q_ in zip(t[0], t[1])]
    out = {}
    for mib in (1, 4, 16, 64):
        want = mib << 20
        tickets, cs, cq, cur = [], [], [], 0
        for s_, q_ in reads:
            cs.append(s_); cq.append(q_); cur += len(s_)
            if cur >= want:
                tickets.append((cs, cq)); cs, cq, cur = [], [], 0
        if not tickets:
            continue
        n_t = max(3, min(48, (128 << 20) // want))  # ~128 Mi bases per measurement, at least three tickets
        row = {"ticket_bases": int(sum(len(x) for x in tickets[0][0])), "tickets": n_t}
        for callers in ((1, 3, 8, 16) if mib <= 4 else (1, 3)):  # (the reference runs `-c` workers, each with a ticket of its own: many callers is ITS way of using small tickets)
            best = None  # the better of two runs: the first tickets of a new size allocate their buffers (a hipMalloc of GBs stalls the device, DESIGN_HISTORY.md 3.3b), later ones take them from the pool
            for _ in range(2):
                r = host_inclusive_leg(types.SimpleNamespace(host_tickets=n_t, host_callers=callers), api, graph, opts, tickets)
                if "value" not in r:
                    best = r; break
                if best is None or r["value"] > best["value"]:
                    best = r
            row["split_api_callers_%d" % callers] = best.get("value") if "value" in best else best
            # the same tickets through rtk_correct_batch (the seam's own function): concurrent callers share launches
            best = None
            for _ in range(2):
                r = correct_batch_leg(api, graph, opts, tickets, n_t, callers)
                if "value" not in r:
                    best = r; break
                if best is None or r["value"] > best["value"]:
                    best = r
            row["callers_%d" % callers] = best.get("value") if "value" in best else best
            if "value" in best:
                row["callers_%d_tickets_per_launch" % callers] = round(best["tickets"] / max(1, best["launch_groups"]), 2)
        out["%dMi" % mib] = row
    out["what"] = ("callers_N: N threads calling rtk_correct_batch (host strings in, malloc'd strings out, rtk_free inside the clock), tickets of concurrent callers merged into one launch by the library; "
                   "split_api_callers_N: the same tickets through rtk_batch_create + run + fetch_view + free, one launch per ticket (the round-5 figures)")
    return out


def lane_kernel_leg(a, api, graph, opts, mine):
    """The lane-per-region kernel (csrc/hip/rtk_region_lane.h, k_regions_lanes; off by default) on one resident ticket, beside the default path on the same
    ticket: kernel times of the region stage either way, the class's size, the share the lane program handed on to the wave kernel, and whether the corrected
    batch is the same bytes."""
    import hashlib

    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            b = api.Batch(graph, *mine[0])
            best = None
            for _ in range(3):
                b.run(opts); st = b.stats()
                if best is None or st["ms_correct"] < best["ms_correct"]:
                    best = st
            h = hashlib.sha256()
            for g_ in b.fetch():
                h.update(g_[0].encode()); h.update(g_[1].encode())
            b.close()
            return best, h.hexdigest()[:16]
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    try:
        off, d0 = run({"RTK_LANE_MAX_GAP": "0"})
        res = {"default": "off (RTK_LANE_MAX_GAP=0): the wave kernel alone", "region_stage_ms_wave_kernel_alone": round(off["ms_correct"], 3), "settings": {}}
        for name, env in (("gap<128, 1024 lane waves, beside the wave kernel (shared queue)", {"RTK_LANE_MAX_GAP": "128", "RTK_LANE_WAVES": "1024"}),
                          ("gap<128, 1024 lane waves, one kernel after the other", {"RTK_LANE_MAX_GAP": "128", "RTK_LANE_WAVES": "1024", "RTK_LANE_SERIAL": "1"}),
                          ("gap<256, 2048 lane waves, one kernel after the other", {"RTK_LANE_MAX_GAP": "256", "RTK_LANE_WAVES": "2048", "RTK_LANE_SERIAL": "1"})):
            st, d = run(env)
            res["settings"][name] = {"region_stage_ms": round(st["ms_correct"], 3), "k_regions_lanes_ms": round(st["ms_lanes"], 3), "lane_class_regions": int(st["n_lane_regions"]),
                                     "handed_on_to_wave_kernel": int(st["n_lane_handed"]), "handed_on_share": round(st["n_lane_handed"] / max(1, st["n_lane_regions"]), 4),
                                     "regions": int(st["n_regions"]), "same_bytes_as_default": d == d0}
        return res
    except Exception as e:  # never takes the bench line down
        return {"error": str(e)[-300:]}


def config4_leg(a, workdir, t_start):
    """configs[4] (BASELINE.json: whole-genome-scale graph resident in HBM, long reads, roofline report) as its own process once this one has released the device:
    bench_config4.py builds the 3 Gb index (short reads sampled inside the index tool), keeps the graph resident and corrects tickets of DISTINCT reads.
    Guarded: host memory / disk / HBM inside that script, time here -- every refusal is a {"skipped": reason}."""
    # time budget: the leg starts ~4 min into a run and takes ~6.5 min (index build 5.6); it is not started later than 7.5 min in and is cut at 16 min (index build at 10),
    # so that the whole run stays under 25 min whatever the box does
    if time.time() - t_start > 450:
        return {"skipped": "time: this bench run had already used %d s when the leg was due (limit 450 s; run `python bench_config4.py out.json` on its own)" % (time.time() - t_start)}
    out_fn = os.path.join(workdir, "config4.json")
    try:
        os.remove(out_fn)
    except OSError:
        pass
    cmd = [sys.executable, os.path.join(ROOT, "bench_config4.py"), out_fn, str(a.config4_ref_mb), "30", str(a.config4_tickets), "128", os.path.join(workdir, "c4")]
    try:
        import signal
        pr = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True, env=dict(os.environ, RTK_C4_INDEX_TIMEOUT=os.environ.get("RTK_C4_INDEX_TIMEOUT", "600")))
        try:
            _, err = pr.communicate(timeout=960)
        except subprocess.TimeoutExpired:
            os.killpg(pr.pid, signal.SIGKILL)  # (the whole group: the index tool it started must not outlive it)
            pr.communicate()
            d = json.load(open(out_fn)) if os.path.exists(out_fn) else {}
            d["skipped"] = "time: the leg was cut after 960 s"
            return d
        d = json.load(open(out_fn))
        if pr.returncode != 0 and "skipped" not in d:
            d["error"] = (err or "").strip()[-400:]
        return d
    except Exception as e:  # never takes the bench line down
        try:
            d = json.load(open(out_fn)); d["error"] = str(e)[-300:]; return d
        except Exception:
            return {"error": str(e)[-300:]}


def cli_leg(a, pre, fa, rt):
    """The shipped C++ driver, file to file: `Ratatosk correct -1` on the generated long-read FASTQ (parse + pack + H2D + kernels + D2H
    + format + ordered write); its own statistics line gives the wall time of the correction phase (graph load reported apart)."""
    import re
    import shutil
    exe = os.path.join(ROOT, "ratatosk_amd", "bin", "Ratatosk")
    out = os.path.join(os.path.dirname(pre), "cli_out")
    env = dict(os.environ, RTK_CLI_STATS="1")
    cores = min(16, os.cpu_count() or 1)
    try:
        lst = out + ".inputs.fq"  # ONE file of ~7 Gb (the generated FASTQ 48 times, read names repeat): long-read runs come as few big files, and the
        with open(pre + ".lr.fq", "rb") as f:  # pipeline has to reach its steady state (the first tickets pay for the pinned staging buffers, the
            one = f.read()                      # device buffer pool and the first launches: ~0.3 s)
        with open(lst, "wb") as f:
            reps = max(2, min(48, int(7.2e9 // max(1, len(one) // 2))))  # ~7 Gb of bases (a FASTQ record is ~2 bytes per base)
            for _ in range(reps):
                f.write(one)
        del one
        r = subprocess.run([exe, "correct", "-1", "-c", str(cores), "--gpus", "1", "-g", fa, "-d", rt, "-l", lst, "-o", out], capture_output=True, text=True, env=env, timeout=600)
        m = re.search(r"graph load \+ upload ([0-9.]+) s; correction phase ([0-9.]+) s wall, (\d+) bases, ([0-9.e+]+) bases/s on (\d+) GPU\(s\) x (\d+) workers; thread-seconds: parse ([0-9.]+), correct \(pack \+ GPU \+ fetch\) ([0-9.]+), format ([0-9.]+), write ([0-9.]+)", r.stderr + r.stdout)
        if r.returncode != 0 or not m:
            return {"error": (r.stderr or r.stdout)[-300:]}
        res = {"value": int(m.group(3)) / float(m.group(2)), "unit": "bases/s", "bases": int(m.group(3)), "correction_phase_s": float(m.group(2)), "graph_load_upload_s": float(m.group(1)),
               "workers_per_gpu": int(m.group(6)), "thread_seconds": {"parse": float(m.group(7)), "pack+gpu+fetch": float(m.group(8)), "format": float(m.group(9)), "write": float(m.group(10))},
               "what": "Ratatosk correct -1 -c %d --gpus 1, plain FASTQ in (one file: the generated reads repeated up to ~7 Gb, parsed as byte ranges by the -c threads), OUT.2.fastq out (input order, FASTQ blocks formatted and written with pwrite by formatter threads), wall time of the correction phase" % cores}
        for fn in (out + ".2.fastq", lst):
            try:
                os.remove(fn)
            except OSError:
                pass
        return res
    except Exception as e:  # the host legs never take the bench line down
        return {"error": str(e)[-300:]}


def second_pass_leg(a, pre):
    """The next row of the scope table, measured the same way (not part of `value`): `Ratatosk correct -1`, the second index at k2 = 63
    (short-read graph coloured by the pass-1 reads, like src/Ratatosk.cpp:1193,1227), then `Ratatosk correct -2`, file to file."""
    import re
    exe = os.path.join(ROOT, "ratatosk_amd", "bin", "Ratatosk")
    out = os.path.join(os.path.dirname(pre), "p2_out")
    env = dict(os.environ, RTK_CLI_STATS="1")
    cores = min(16, os.cpu_count() or 1)
    pat = r"correction phase ([0-9.]+) s wall, (\d+) bases"
    try:
        r1 = subprocess.run([exe, "correct", "-1", "-c", str(cores), "--gpus", "1", "-g", pre + ".index.k31.fasta.gz", "-d", pre + ".index.k31.rtsk", "-l", pre + ".lr.fq", "-o", out], capture_output=True, text=True, env=env, timeout=300)
        if r1.returncode != 0:
            return {"error": (r1.stderr or r1.stdout)[-300:]}
        t0 = time.time()
        subprocess.run([os.path.join(ROOT, "ratatosk_amd", "bin", "rtk_build_index"), "-s", pre + ".sr.fq", "--colour-reads", out + ".2.fastq", "-k", "63", "-o", out + ".p2"], stderr=subprocess.DEVNULL, check=True, timeout=600)
        t_idx = time.time() - t0
        # like the first-pass CLI leg: list files (the corrected reads and, in step, the uncorrected ones, 18 times each: 2.7 Gb, ~80 tickets for 8 in flight) so that the ticket
        # pipeline runs in steady state; the figure of ONE copy of the files (4 tickets: mostly pipeline fill and drain) is kept next to it
        reps = 18
        with open(out + ".p2in.txt", "w") as f:
            f.write((out + ".2.fastq\n") * reps)
        with open(out + ".p2raw.txt", "w") as f:
            f.write((pre + ".lr.fq\n") * reps)
        res = {}
        for tag, l_in, l_raw in (("steady", out + ".p2in.txt", out + ".p2raw.txt"), ("once", out + ".2.fastq", pre + ".lr.fq")):
            r2 = subprocess.run([exe, "correct", "-2", "-c", str(cores), "--gpus", "1", "-g", out + ".p2.index.k63.fasta.gz", "-d", out + ".p2.index.k63.rtsk", "-l", l_in, "-L", l_raw, "-o", out],
                                capture_output=True, text=True, env=env, timeout=600)
            m = re.search(pat, r2.stderr + r2.stdout)
            if r2.returncode != 0 or not m:
                return {"error": (r2.stderr or r2.stdout)[-300:]}
            res[tag] = (int(m.group(2)), float(m.group(1)))
        try:
            os.remove(out + ".fastq")
        except OSError:
            pass
        return {"value": res["steady"][0] / res["steady"][1], "unit": "bases/s", "bases": res["steady"][0], "correction_phase_s": res["steady"][1], "k2": 63, "second_index_build_s": round(t_idx, 1),
                "one_copy": {"value": res["once"][0] / res["once"][1], "bases": res["once"][0], "correction_phase_s": res["once"][1]},
                "what": "Ratatosk correct -2 -c %d --gpus 1 on the OUT.2.fastq of a `correct -1` run + the uncorrected reads (list files: %d times each, steady state; one_copy: the files once), "
                        "OUT.fastq out; phasing() pre-filter, exact anchors, k2 = 63 (two-word k-mers)" % (cores, reps)}
    except Exception as e:
        return {"error": str(e)[-300:]}


def cpu_baseline_leg(a, api, graph, opts, fa, rt, tickets, whole_alg, out):
    """CPU baseline: the oracle (C++ restatement of the reference path; the reference binary cannot be built: Bifrost absent) with the
    reference's threading model (N threads pulling reads, src/Ratatosk.cpp:727-904) and the REFERENCE's own edlib underneath
    (oracle/_ref) when it is there. Two figures as BASELINE.md 3 promises: 32 threads (the reference's `medium node` sizing) with >= 20
    reads per thread, and all host threads on the sample the parity spot check uses."""
    from oracle import oracle_py as op
    og = op.Graph(fa, rt, 31)
    alignment_layer = "oracle_myers.cpp (unbanded restatement)"
    try:
        op.use_reference_edlib(True); alignment_layer = "reference edlib (oracle/_ref/libedlib_ref.so)"
    except Exception:
        pass
    seqs, quals = tickets[0]

    def sample(max_reads, max_bases):
        ss, qq, tot = [], [], 0
        for s, q in zip(seqs, quals):
            if len(ss) >= max_reads or tot >= max_bases:
                break
            ss.append(s); qq.append(q); tot += len(s)
        return ss, qq, tot

    cores = os.cpu_count() or 1
    quota = None  # the container's CPU quota (cgroup v2 cpu.max "quota period"): the host may show 256 hardware threads of which the control group grants a part
    try:
        q_, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q_ != "max":
            quota = float(q_) / float(p_)
    except Exception:
        pass
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))
    legs = {}
    t32 = min(32, cores)
    ss, qq, tot_b = sample(20 * t32, 1 << 62)
    t1 = time.time(); _, cnt32 = og.correct_batch(ss, qq, threads=t32); dt = time.time() - t1
    legs["threads_min32_quota"] = {"is": "min(32, the CPUs the container is granted) threads, >= 20 reads per thread: the reference's `medium node` sizing where the box allows it", "value": tot_b / dt, "unit": "bases/s", "threads": t32, "reads": len(ss), "bases": tot_b, "per_thread": tot_b / dt / t32, "seconds": round(dt, 2)}
    # the section-8(d) formula of the survey charges every spelled 1-edit variant as a probe: the oracle spells them, so its counters give it
    survey = (8.0 * cnt32["n_probe"] + 8.0 * cnt32["n_verify"] + 40.0 * cnt32["n_expand"] + 4.0 * cnt32["n_colour_elem"] + 0.25 * cnt32["n_path_base"]) / max(1, tot_b) + 4.0
    out["config"]["alg_bytes_per_base_survey_formula"] = round(survey, 1)
    out["config"]["alg_bytes_note"] = "alg_bytes_per_base = 8 N_probe(exact) + 16 N_slot + 40 N_expand + 4 N_colour + 0.25 N_pathbase + 4 L as the HIP path executes it (1-edit search by half-k-mer seeds); *_survey_formula = SURVEY.md 8(d): 8 N_probe + 8 N_verify + ... with every spelled 1-edit variant a probe (counted by the oracle on the 32-thread sample)"
    ss, qq, tot_b = sample(1 << 30, a.cpu_sample_bases)
    t1 = time.time(); want, _ = og.correct_batch(ss, qq, threads=cores); dt_cpu = time.time() - t1
    legs["all_threads"] = {"value": tot_b / dt_cpu, "unit": "bases/s", "threads": cores, "reads": len(ss), "bases": tot_b, "per_thread": tot_b / dt_cpu / cores, "reads_per_thread": round(len(ss) / cores, 1), "seconds": round(dt_cpu, 2)}
    op.use_reference_edlib(False)
    # parity spot check on the same sample (the oracle is the checker here, not the thing measured above)
    chk = api.Batch(graph, ss, qq); chk.run(opts); got = chk.fetch(); chk.close()
    best = max(legs.values(), key=lambda x: x["value"])
    return {"value": best["value"], "unit": "bases/s", "cores": best["threads"], "kind": "port", "alignment_layer": alignment_layer,
            "sample": "%d reads / %d bases of step 0 (best of the two legs below)" % (best["reads"], best["bases"]), "legs": legs, "parity_on_sample": got == want,
            "host": {"hardware_threads": os.cpu_count(), "cgroup_cpu_quota": quota, "threads_of_the_all_threads_leg": cores}}


PMC_NOTE = "from the committed rocprofv3 --pmc passes of this workload (profiles/rNN_pmc_summary.json): counters cannot be collected inside a timed run"
SIMDS, CLOCK_HZ = 1024, 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz; a wave64 VALU instruction issues over 2 cycles


def pmc_issue(kernel, avg_ms):
    """Issue-slot use of `kernel`: (VALU + SALU wave-instructions per launch, from the committed SQ counter pass) / (SIMDs x clock x launch
    time / 2 cycles per wave64 instruction). Says how far "latency-bound" is from "issue-bound"."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")):
        m = re.search(r"r(\d+)_pmc_summary", f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    try:
        sq = json.load(open(best[1]))["kernels"][kernel]["sq"]
        n = float(sq["SQ_INSTS_VALU"]) + float(sq["SQ_INSTS_SALU"])
        return round(n / (SIMDS * CLOCK_HZ * avg_ms * 1e-3 / 2.0), 4), int(n)
    except Exception:
        return None, None


def run_workload(a, ctx, ref_len, het, name, steps, warmup, n1_first=False):
    """One workload through the timed loop: dataset + index (rank 0), graph replicated, batches resident in HBM, W warm-up steps, K timed
    steps bracketed by synchronize + barrier. Returns everything the JSON line is made of."""
    import torch
    import torch.distributed as dist
    from ratatosk_amd import api, dist as rdist
    rank, world, device, lib_path = ctx["rank"], ctx["world"], ctx["device"], ctx["lib_path"]
    n_batches = steps + warmup
    # distinct tickets: two per rank are enough to overlap consecutive steps; at N = 1 up to four (30x of a 60 Mb reference would give 28)
    lr_bases = int(2.3 * a.batch_bases * world) + 200_000 if world > 1 else min(a.batch_bases * n_batches, 30 * ref_len, int(4.3 * a.batch_bases) + 200_000)
    if rank == 0:
        workdir = ctx["workdir"]
        t0 = time.time()
        pre = make_dataset(workdir, ref_len, lr_bases, snps=not a.plain_index, het=het, fast="" if a.sim else "--gpu", name=name)
        t_data = time.time() - t0
    else:
        pre, t_data = None, 0.0
    if world > 1:
        box = [pre]
        dist.broadcast_object_list(box, src=0)
        pre = box[0]
    fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
    t0 = time.time()
    repl = {}
    graph = rdist.load_graph_replicated(fa, rt, 31, rank, world, device, lib_path=lib_path, report=repl)
    t_graph = time.time() - t0
    seqs, quals = read_long_reads(pre + ".lr.fq", lr_bases)
    # batches by ticket: consecutive reads until >= batch_bases; rank r owns tickets r, r+N, ...
    tickets, cur_s, cur_q, cur = [], [], [], 0
    for s_, q_ in zip(seqs, quals):
        cur_s.append(s_); cur_q.append(q_); cur += len(s_)
        if cur >= a.batch_bases:
            tickets.append((cur_s, cur_q)); cur_s, cur_q, cur = [], [], 0
    if cur_s and not tickets:
        tickets.append((cur_s, cur_q))
    mine = [t for i, t in enumerate(tickets) if i % world == rank]
    shared_tickets = False
    if not mine:
        mine = [tickets[rank % len(tickets)]]; shared_tickets = True
    opts = graph.opts()
    # resident in HBM before timing; at least two batch objects so that consecutive steps can overlap (stage A of step s+1 with stage B of step s)
    batches = [api.Batch(graph, *mine[i % len(mine)]) for i in range(max(2, min(n_batches, len(mine))))]

    def sync():
        if not a.sim:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(seq_b, serial):
        t0_ = time.time()
        if serial:
            for b in seq_b:
                b.run(opts)
        else:
            api.run_pipelined(seq_b, opts)
        return t0_

    # a step = one batch through both stages, one step after the other (--overlap: consecutive steps software-pipelined on two HIP streams).
    # All K steps are complete before the clock stops.
    api.run_pipelined([batches[w % len(batches)] for w in range(warmup)], opts)
    sync()
    n1 = None
    if n1_first and world > 1:
        # the N = 1 point of THIS graph, measured in this process before the sharded run: rank 0 runs K steps alone, the other ranks wait
        if rank == 0:
            seq_1 = [batches[(warmup + st) % len(batches)] for st in range(steps)]
            t0 = timed(seq_1, a.serial)
            if not a.sim:
                torch.cuda.synchronize()
            dt1 = time.time() - t0
            b1 = sum(b.in_bases for b in seq_1)
            n1 = {"value": b1 / dt1 if dt1 > 0 else 0.0, "ms_per_step": 1e3 * dt1 / max(1, steps), "steps": steps, "measured": "in this run, on rank 0 alone (the other ranks idle at a barrier), same graph, same tickets, before the sharded steps"}
        sync()
    seq_b = [batches[(warmup + st) % len(batches)] for st in range(steps)]
    t0 = timed(seq_b, a.serial)
    sync()
    dt = time.time() - t0
    done_bases = sum(b.in_bases for b in seq_b)
    stats = [b.stats() for b in seq_b]
    # per-kernel times of a step on an otherwise idle GPU (after the clock has stopped): in the timed region two batches overlap, and the
    # HIP-event span of the seed kernels of one then includes their wait for wave slots behind the other's persistent region kernel
    stats_serial = []
    if not a.serial:
        for b in seq_b[:2]:
            b.run(opts); stats_serial.append(b.stats())
        sync()
    if world > 1:
        t = torch.tensor([dt, float(done_bases)], dtype=torch.float64, device="cpu" if a.sim else "cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_all, bases_all = float(tmax[0]), float(tsum[1])
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "device": device, "bases": int(done_bases), "seconds": round(dt, 4), "distinct_tickets": len(mine), "shared_tickets": shared_tickets})
        assert dist.get_world_size() == world == a.gpus and (a.sim or dist.get_backend() == "nccl"), "bench.py --gpus N must run as N ranks over RCCL"
    else:
        dt_all, bases_all = dt, float(done_bases)
        per_rank = [{"rank": 0, "device": device, "bases": int(done_bases), "seconds": round(dt, 4), "distinct_tickets": len(mine), "shared_tickets": shared_tickets}]
    for b in batches:
        b.close()
    return {"pre": pre, "fa": fa, "rt": rt, "graph": graph, "info": graph.info(), "opts": opts, "mine": mine, "repl": repl, "t_data": t_data, "t_graph": t_graph,
            "dt_all": dt_all, "bases_all": bases_all, "per_rank": per_rank, "stats": stats, "stats_serial": stats_serial, "n1": n1}


KERN = {"k_lookup_exact": "ms_lookup_exact", "k_mask": "ms_mask", "k_inexact": "ms_lookup_inexact", "k_finalize": "ms_seeds", "k_regions": "ms_correct", "k_stitch": "ms_stitch"}


T_START = time.time()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from ratatosk_amd import api

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    lib_path = os.path.join(ROOT, "tests", "hostsim", "librtk_hostsim.so") if a.sim else None
    if not a.sim and not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible (the hot path has no CPU fallback)")
    device = 0 if a.sim else local_rank
    if not a.sim:
        torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group(backend="gloo" if a.sim else "nccl")
    api.load_library(lib_path)
    ctx = {"rank": rank, "world": world, "device": device, "lib_path": lib_path, "workdir": (a.workdir or tempfile.mkdtemp(prefix="rtk_bench_")) if rank == 0 else None}
    if ctx["workdir"]:
        os.makedirs(ctx["workdir"], exist_ok=True)

    diploid = a.het > 0
    w = run_workload(a, ctx, a.ref_len, a.het, "c2" if diploid else "c1", a.steps, a.warmup, n1_first=True)
    stats, stats_serial, info = w["stats"], w["stats_serial"], w["info"]

    if rank == 0:
        # ---- roofline of the dominant kernel, from the HIP-event times of this very run ----
        kern = KERN
        tot = {k_: sum(s[v] for s in stats) for k_, v in kern.items()}
        dom = max(tot, key=tot.get)
        n_l = max(1, len(stats))
        avg_ms = tot[dom] / n_l
        S = lambda key: sum(s[key] for s in stats) / n_l  # per-launch averages
        # algorithmic bytes per launch (SURVEY.md 8d / DESIGN.md): 8-byte pre-filter word per k-mer query + 16-byte table slot per slot visited, 40 B per expanded unitig
        # (8 neighbour slots + flag word), 4 B per colour id streamed, 0.25 B per path base materialised, 4 B per read base in/out.
        alg = {
            "k_lookup_exact": 8.0 * S("n_probes_exact") + 16.0 * S("n_slots_exact") + 1.0 * S("in_bases") + 8.0 * S("in_bases"),
            "k_inexact": 16.0 * S("n_slots_inexact") + 1.0 * S("in_bases") + 16.0 * S("n_hits_inexact"),  # index slots + 2 per checked candidate (rtk_seeded_window)
            "k_regions": 40.0 * S("n_expand") + 4.0 * S("n_colour_elem") + 0.25 * S("n_path_base") + 4.0 * S("in_bases"),
            "k_mask": 9.0 * S("in_bases"), "k_finalize": 12.0 * S("in_bases") + 16.0 * S("n_hits_inexact"), "k_stitch": 4.0 * S("out_bases"),
        }
        achieved = alg[dom] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(dom)
        issue_frac, issue_n = pmc_issue(dom, avg_ms)
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "traffic_source": traffic_src, "traffic_note": PMC_NOTE if traffic is not None else None,
                    "issue_frac": issue_frac, "issue_wave_instructions_per_launch": issue_n, "issue_frac_is": "(SQ_INSTS_VALU + SQ_INSTS_SALU of the committed PMC pass) / (1024 SIMDs x 2.4 GHz x this run's launch time / 2 cycles per wave64 instruction)",
                    "alg_bytes_per_launch": int(alg[dom]), "avg_launch_ms": round(avg_ms, 3),
                    "kernel_ms_per_step": {k_: round(v / n_l, 3) for k_, v in tot.items()},
                    "kernel_ms_per_step_is": ("HIP-event spans inside the timed region, one step at a time (the default): every kernel has the machine to itself" if a.serial else "HIP-event spans inside the timed region with consecutive steps on two streams (--overlap): the spans of k_mask / k_inexact / k_finalize include waiting for wave slots behind the other step's persistent k_regions; kernel_ms_per_step_serial has the same kernels with one step at a time"),
                    "kernel_ms_per_step_serial": ({k_: round(sum(s_[v] for s_ in stats_serial) / len(stats_serial), 3) for k_, v in kern.items()} if stats_serial else {k_: round(v / n_l, 3) for k_, v in tot.items()}), "regions_per_step": int(S("n_regions")), "aligns_per_step": int(S("n_align")), "align_word_columns_per_step": int(S("n_align_cells")), "expansions_per_step": int(S("n_expand")), "colour_ids_per_step": int(S("n_colour_elem")), "path_bases_per_step": int(S("n_path_base")),
                    "index_lookups_inexact_per_step": int(S("n_probes_inexact")), "index_slots_inexact_per_step": int(S("n_slots_inexact")),
                    "k_regions_wave_cycle_share": {k_: round(S(k_) / max(1.0, S("cyc_total")), 3) for k_ in ("cyc_colour", "cyc_paths", "cyc_consensus", "cyc_myers", "cyc_sets", "cyc_tostring", "cyc_pathqual", "cyc_walk")},
                    "alignment_moves_per_step": int(S("n_moves")), "k_regions_wave_ticks_per_step": int(S("cyc_total")), "regions_redone_bigger_arena": int(S("n_arena_overflow"))}
        # the second kernel of the path, the one that is bound by HBM (random 8-byte reads of the half-k-mer index, the unitig pool behind its candidates)
        ki_ms = tot["k_inexact"] / n_l
        ki_traffic, ki_src = pmc_traffic("k_inexact")
        roofline["k_inexact"] = {"bound": "hbm", "achieved": round(alg["k_inexact"] / (ki_ms * 1e-3) / 1e9, 3) if ki_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(alg["k_inexact"] / (ki_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ki_ms > 0 else 0.0, "alg_bytes_per_launch": int(alg["k_inexact"]), "avg_launch_ms": round(ki_ms, 3),
                                 "traffic": ki_traffic, "traffic_source": ki_src,
                                 "alg_bytes_are": "8 B per index slot visited + 8 B per list word + 16 B per checked candidate (its entry holds the bases around the half k-mer: no read of the unitig pool) + 1 B per read base + 16 B per hit written"}
        whole_alg = (8.0 * S("n_probes_exact") + 16.0 * (S("n_slots_exact") + S("n_slots_inexact")) + 40.0 * S("n_expand") + 4.0 * S("n_colour_elem") + 0.25 * S("n_path_base") + 4.0 * S("in_bases")) / max(1.0, S("in_bases"))
        dt_all, bases_all = w["dt_all"], w["bases_all"]
        set_name = ("HG002-chr20-scale set (configs[2]'s graph): k=31 first pass, %.1f Mb diploid random ref (%.2f %% het SNPs)" % (a.ref_len / 1e6, 100 * a.het)) if diploid else ("configs[1]: k=31 first-pass correct, %.1f Mb random ref" % (a.ref_len / 1e6))
        out = {
            "metric": "corrected long-read bases/sec", "value": bases_all / dt_all if dt_all > 0 else 0.0, "unit": "bases/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt_all / max(1, a.steps), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "vs_baseline_note": "null: BASELINE.md holds no published number for this metric and the reference binary cannot be built here (Bifrost, its un-vendored submodule, is absent), so the north_star's '>= 10x the reference CPU' cannot be evaluated against the real binary; cpu_baseline is this repo's port of the path",
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s, 30x PE150 short reads, ONT-R9.4-profile long reads, %d bases/step/GPU%s" % (set_name, a.batch_bases, (", sharded across %d MI355X, >= 2 distinct tickets per GPU" % world) if world > 1 else ", 1 MI355X"),
                       "graph_replication": w["repl"] or None,
                       "n1_on_this_graph": w["n1"],
                       "graph": {"unitigs": int(info.n_unitigs), "kmers": int(info.n_kmers), "hbm_bytes": int(info.hbm_bytes)}, "parallelism": "reads sharded by ticket x%d, graph replicated (one RCCL broadcast per flat buffer)" % world, "per_rank": w["per_rank"],
                       "value_is": "kernel-resident throughput: batches packed and in HBM before the clock starts (the contract's definition); the host-buffer-to-host-buffer rate is host_inclusive, the file-to-file rate cli_file_to_file",
                       "alg_bytes_per_base": round(whole_alg, 1), "setup_s": {"data+index": round(w["t_data"], 1), "graph_load+upload": round(w["t_graph"], 1)}},
            "roofline": roofline,
        }
        if world == 1 and not a.no_host_legs:
            out["host_inclusive"] = host_inclusive_leg(a, api, w["graph"], w["opts"], w["mine"])
            out["by_ticket_size"] = ticket_size_leg(a, api, w["graph"], w["opts"], w["mine"])
            out["lane_kernel"] = lane_kernel_leg(a, api, w["graph"], w["opts"], w["mine"])
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg(a, api, w["graph"], w["opts"], w["fa"], w["rt"], w["mine"], whole_alg, out)
    w["graph"].close()
    if rank == 0 and world == 1 and not a.no_host_legs:
        # The executable's legs run once this process has released the device (its graph image, 83 GB of work areas, pools): `Ratatosk correct -2` sizes its tickets in flight by
        # the device memory that is free, and beside this process it ran with five or six instead of eight (6.1 x 10^8 in the line against 8.1-9.0 x 10^8 on its own, round 6)
        out["cli_file_to_file"] = cli_leg(a, w["pre"], w["fa"], w["rt"])
        # SURVEY.md 8(d) defines the metric over the wall time of the correction phase of the executable: that figure, at the top level beside the kernel-resident `value`
        out["value_correction_phase"] = out["cli_file_to_file"].get("value")
        out["value_correction_phase_is"] = "cli_file_to_file.value: `Ratatosk correct -1` file to file, input bases / wall time of the correction phase (parse + pack + H2D + kernels + D2H + format + ordered write); quote THIS when one number is quoted"
        out["second_pass"] = second_pass_leg(a, w["pre"])
    if world == 1 and not a.no_config1_leg:
        # the metric's other single-GPU configuration, in the same run: configs[1] (5 Mb haploid reference; its graph sits largely in the caches)
        w1 = run_workload(a, ctx, 5_000_000, 0.0, "c1", a.steps, a.warmup)
        n1_ = max(1, len(w1["stats"]))
        out["config1"] = {"workload": "configs[1]: k=31 first-pass correct, 5.0 Mb random ref, 30x PE150 short reads, ONT-R9.4-profile long reads, %d bases/step, same steps / warm-up / overlap as the main line" % a.batch_bases,
                          "value": w1["bases_all"] / w1["dt_all"] if w1["dt_all"] > 0 else 0.0, "unit": "bases/s", "ms_per_step": 1e3 * w1["dt_all"] / max(1, a.steps),
                          "kernel_ms_per_step": {k_: round(sum(s_[v] for s_ in w1["stats"]) / n1_, 3) for k_, v in KERN.items()},
                          "kernel_ms_per_step_serial": ({k_: round(sum(s_[v] for s_ in w1["stats_serial"]) / len(w1["stats_serial"]), 3) for k_, v in KERN.items()} if w1["stats_serial"]
                                                        else {k_: round(sum(s_[v] for s_ in w1["stats"]) / n1_, 3) for k_, v in KERN.items()}),  # (steps one at a time are the default: the spans above ARE serial)
                          "graph": {"unitigs": int(w1["info"].n_unitigs), "kmers": int(w1["info"].n_kmers), "hbm_bytes": int(w1["info"].hbm_bytes)}}
        w1["graph"].close()
    if world == 1 and not (a.no_config4 or a.no_host_legs or a.config1_only or a.sim):  # (the developer runs that leave out the host legs leave this one out too)
        # (last: every graph and work area of this process is released by now -- the leg's own process needs the 288 GB)
        out["config4"] = config4_leg(a, ctx["workdir"], T_START)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
