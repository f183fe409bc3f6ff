"""The bench.py contract on CPU: one JSON line with the fields the driver reads, at N = 1 and as two ranks under torch.distributed.run
(developer simulator + gloo: the plumbing of `--gpus N` -- graph replication, ticket sharding, max-over-ranks timing -- not a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--sim", "--ref-len", "60000", "--batch-bases", "120000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-host-legs"]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check_common(d, n):
    assert d["metric"] == "corrected long-read bases/sec" and d["unit"] == "bases/s" and d["n_gpus"] == n and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["value"] > 0
    assert "workload" in d["config"] and len(d["config"]["per_rank"]) == n
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and "traffic" in r


def test_bench_line_single(tmp_path):
    """N = 1: the main line is the chr20-scale (diploid) set -- here at a reduced --ref-len -- and the same run carries configs[1] as the extra leg."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--workdir", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    _check_common(d, 1)
    assert d["config"]["workload"].startswith("HG002-chr20-scale set (configs[2]'s graph)") and "diploid" in d["config"]["workload"] and d["config"]["n1_on_this_graph"] is None
    assert d["vs_baseline_note"].startswith("null:") and "issue_frac" in d["roofline"]
    c1 = d["config1"]
    assert c1["workload"].startswith("configs[1]") and c1["value"] > 0 and c1["ms_per_step"] > 0 and set(c1["kernel_ms_per_step"]) == set(d["roofline"]["kernel_ms_per_step"])


def test_bench_line_two_ranks(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + ["--workdir", str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    _check_common(d, 2)
    pr = d["config"]["per_rank"]
    assert sorted(p["rank"] for p in pr) == [0, 1] and all(p["bases"] > 0 for p in pr)
    assert abs(d["value"] - sum(p["bases"] for p in pr) / max(p["seconds"] for p in pr)) / d["value"] < 0.05  # whole-job rate: all ranks' bases over the slowest rank's time
    n1 = d["config"]["n1_on_this_graph"]  # the one-GPU point of the same graph, measured in this very run on rank 0 (not read from a file)
    assert n1["value"] > 0 and n1["ms_per_step"] > 0 and n1["steps"] == 1 and "in this run" in n1["measured"] and "config1" not in d


def test_bench_line_eight_ranks_is_the_configs2_run(tmp_path):
    """`--gpus 8` as the driver launches it (8 ranks; here gloo + the simulator, a reduced reference): the line a SCALE record would be made of.
    N > 1 is the configs[2] run (diploid graph, >= 2 distinct tickets per rank, per-rank bases, the broadcast of every flat buffer timed)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29537",
           os.path.join(ROOT, "bench.py"), "--gpus", "8"] + SMALL + ["--workdir", str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    _check_common(d, 8)
    assert d["config"]["workload"].startswith("HG002-chr20-scale set (configs[2]'s graph)") and "diploid" in d["config"]["workload"] and "sharded across 8 MI355X" in d["config"]["workload"]
    pr = d["config"]["per_rank"]
    assert sorted(p["rank"] for p in pr) == list(range(8)) and all(p["bases"] > 0 and p["distinct_tickets"] >= 2 and not p["shared_tickets"] for p in pr)
    rep = d["config"]["graph_replication"]
    assert rep["ranks"] == 8 and len(rep["bytes_per_buffer"]) == len(rep["broadcast_s_per_buffer"]) >= 19 and rep["load_threads"] >= 1
    assert abs(d["value"] - sum(p["bases"] for p in pr) / max(p["seconds"] for p in pr)) / d["value"] < 0.05


def test_default_graph_is_the_chr20_scale_set_at_every_n():
    """Without --ref-len every N takes the configs[2] reference (60 Mb, 0.1 % heterozygous SNPs): the set the target is written on and the one
    graph all points of a scaling series share; --config1-only is the quick developer line on configs[1]."""
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]; a = bench.parse(); assert (a.ref_len, a.het, a.no_config1_leg) == (60_000_000, 0.001, False)
        sys.argv = ["bench.py", "--gpus", "8"]; a = bench.parse(); assert (a.ref_len, a.het) == (60_000_000, 0.001)
        sys.argv = ["bench.py", "--config2"]; a = bench.parse(); assert (a.ref_len, a.het) == (60_000_000, 0.001)
        sys.argv = ["bench.py", "--config1-only"]; a = bench.parse(); assert (a.ref_len, a.het, a.no_config1_leg) == (5_000_000, 0.0, True)
        sys.argv = ["bench.py", "--gpus", "4", "--ref-len", "1000"]; a = bench.parse(); assert a.ref_len == 1000
    finally:
        sys.argv = old


def test_config4_leg_script_on_the_simulator(tmp_path):
    """bench_config4.py (the `config4` leg of the bench line) end to end on the developer simulator at a three-thousandth of its size: a 1 Mb diploid reference, reads sampled inside
    the index tool, two 64 Mb tickets of distinct reads (one untimed group, then timed), per-kernel figures, the truth-based property checks; and its guards: a reference that
    cannot fit the container's memory ends with {"skipped": reason}, exit code 0."""
    out = os.path.join(str(tmp_path), "c4.json")
    env = dict(os.environ, RTK_C4_SIM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_config4.py"), out, "1", "30", "2", "8", os.path.join(str(tmp_path), "c4")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.load(open(out))
    assert "skipped" not in d and d["graph"]["kmers"] > 900_000 and d["tickets"]["n"] >= 1 and d["tickets"]["distinct"] is True and d["tickets"]["bases_per_s"] > 0
    assert set(d["kernels_one_ticket_alone"]) == {"k_lookup_exact", "k_mask", "k_inexact", "k_finalize", "k_regions", "k_stitch"}
    pc = d["property_checks"]
    assert pc["ok"] and pc["reads_checked"] == 200 and pc["error_rate_corrected"] < 0.2 * pc["error_rate_raw"] and pc["solid_window_share_corrected"] > 0.9
    # a second run finds the index it left
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_config4.py"), out, "1", "30", "2", "8", os.path.join(str(tmp_path), "c4")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and json.load(open(out)).get("index", "").startswith("reused")
    # guard: 10 Tb cannot fit
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_config4.py"), out, "10000000", "30", "3", "8", os.path.join(str(tmp_path), "c4big")], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "skipped" in json.load(open(out))
