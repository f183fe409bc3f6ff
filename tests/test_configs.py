"""Parity at the sizes BASELINE.json's configs name, not only on the small seeded sets of the other tests:

  configs[0]  50 kb random ref, 1 000 x 2x150 bp pairs (6x), 100 x 10 kb long reads with 10 % errors -- oracle AND HIP path must
              reproduce the corrected FASTQ frozen in tests/golden/config0.json (byte-identical: sha256 + per-read CRCs);
  configs[1]  5 Mb ref, 30x PE150 at 0.5 %, ONT-profile long reads (>= 16 Mb of them), SNP-annotated index as bench.py builds it;
  configs[2]  graph of the chr20-scale set: 60 Mb diploid reference with 0.1 % heterozygous SNPs, 30x short reads; a bounded
              long-read sample so that the oracle leg stays below two minutes.

The HIP path runs through the C ABI (ratatosk_amd.api -> libratatosk_hip.so); the oracle is the checker."""
import hashlib
import json
import os
import subprocess
import sys
import time
import zlib

import pytest

from conftest import BIN, ROOT, SIM_LIB
from oracle import oracle_py as op

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_config0_golden as g0  # noqa: E402

GOLD0 = json.load(open(os.path.join(ROOT, "tests", "golden", "config0.json")))


@pytest.fixture(scope="module")
def config0(tmp_path_factory):
    pre = g0.make(str(tmp_path_factory.mktemp("config0")))
    assert g0.input_sums(pre) == GOLD0["inputs"], "rtk_simulate / rtk_build_index no longer write the frozen configs[0] inputs: regenerate tests/golden/config0.json deliberately"
    return pre


def _check_config0(recs, names):
    assert len(recs) == GOLD0["n_reads"]
    bad = [i for i, w in enumerate(recs) if zlib.crc32((w[0] + "\n" + w[1]).encode()) != GOLD0["read_crc32"][i]]
    assert not bad, "corrected records differ from the frozen configs[0] output at reads %s" % bad[:10]
    assert hashlib.sha256(g0.fastq_bytes(names, recs)).hexdigest() == GOLD0["fastq_sha256"]


def test_config0_oracle_reproduces_frozen_fastq(config0):
    reads = op.read_fastq(config0 + ".lr.fq")
    og = op.Graph(config0 + ".index.k31.fasta.gz", config0 + ".index.k31.rtsk", 31)
    want, _ = og.correct_batch([r[1] for r in reads], [r[2] for r in reads], threads=4)
    _check_config0(want, [r[0] for r in reads])


def test_config0_union_reading_of_a2_reproduces_the_bytes_frozen_in_rounds_1_to_3(config0, monkeypatch):
    """[A2] became `exclusive` by default in round 4; under RTK_A2_XOR=union oracle and device program still write the FASTQ frozen before."""
    from ratatosk_amd import api
    monkeypatch.setenv("RTK_A2_XOR", "union")
    reads = op.read_fastq(config0 + ".lr.fq")
    og = op.Graph(config0 + ".index.k31.fasta.gz", config0 + ".index.k31.rtsk", 31)
    want, _ = og.correct_batch([r[1] for r in reads], [r[2] for r in reads], threads=4)
    assert hashlib.sha256(g0.fastq_bytes([r[0] for r in reads], want)).hexdigest() == GOLD0["a2_union"]["fastq_sha256"]
    pg = api.Graph(config0 + ".index.k31.fasta.gz", config0 + ".index.k31.rtsk", 31, device=0, lib_path=SIM_LIB)
    got = pg.correct_batch([r[1] for r in reads[:6]], [r[2] for r in reads[:6]])
    for i, w in enumerate(got):
        assert zlib.crc32((w[0] + "\n" + w[1]).encode()) == GOLD0["a2_union"]["read_crc32"][i], "read %d" % i


def test_config0_device_program_on_simulator(config0):
    """The device programs (host simulator build) on the first reads of configs[0]: same records as frozen."""
    from ratatosk_amd import api
    reads = op.read_fastq(config0 + ".lr.fq")[:6]
    pg = api.Graph(config0 + ".index.k31.fasta.gz", config0 + ".index.k31.rtsk", 31, device=0, lib_path=SIM_LIB)
    got = pg.correct_batch([r[1] for r in reads], [r[2] for r in reads])
    for i, w in enumerate(got):
        assert zlib.crc32((w[0] + "\n" + w[1]).encode()) == GOLD0["read_crc32"][i], "read %d" % i


@pytest.mark.parametrize("n_dev", [3, 8])
def test_config0_cli_on_simulator_with_three_replicas(config0, tmp_path, n_dev):
    """The C++ host driver (reader thread, tickets, 3 workers per GPU, ordered writer, ONE index parse + device-to-device replicas)
    linked against the simulator with three / eight pretend GPUs (a node's worth): OUT.2.fastq is the frozen configs[0] file byte for byte."""
    out = str(tmp_path / "out")
    env = dict(os.environ, RTK_SIM_DEVICES=str(n_dev))
    r = subprocess.run([os.path.join(ROOT, "tests", "hostsim", "Ratatosk_sim"), "correct", "-1", "-v", "-c", "2", "--gpus", str(n_dev), "-B", "40000", "-g", config0 + ".index.k31.fasta.gz",
                        "-d", config0 + ".index.k31.rtsk", "-l", config0 + ".lr.fq", "-o", out], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "on %d GPU(s)" % n_dev in r.stdout
    assert hashlib.sha256(open(out + ".2.fastq", "rb").read()).hexdigest() == GOLD0["fastq_sha256"]


@pytest.mark.gpu
def test_gpu_config0_frozen_fastq_through_the_cli(config0, tmp_path):
    """`Ratatosk correct -1 -c 4` at configs[0]'s stated parameters: OUT.2.fastq must be the frozen file byte for byte."""
    out = str(tmp_path / "out")
    r = subprocess.run([os.path.join(BIN, "Ratatosk"), "correct", "-1", "-c", "4", "-g", config0 + ".index.k31.fasta.gz", "-d", config0 + ".index.k31.rtsk",
                        "-l", config0 + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert hashlib.sha256(open(out + ".2.fastq", "rb").read()).hexdigest() == GOLD0["fastq_sha256"]
    got = op.read_fastq(out + ".2.fastq")
    _check_config0([(g[1], g[2]) for g in got], [g[0] for g in got])


def _sized(tmp, name, sim_args, index_args, sample_bases, skip_bases=0, keep_short_reads=False):
    t0 = time.time()
    pre = os.path.join(str(tmp), name)
    subprocess.check_call([os.path.join(BIN, "rtk_simulate"), "--prefix", pre] + [str(a) for a in sim_args], stderr=subprocess.DEVNULL)
    # (these sets are only built in the gpu tier: k-mers counted on the device, the other heavy steps on the host threads -- the files are the plain
    # tool's byte for byte, tests/test_index_build.py)
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre, "--gpu"] + list(index_args), stderr=subprocess.DEVNULL)
    if not keep_short_reads:
        os.remove(pre + ".sr.fq")
    t_data = time.time() - t0
    reads = op.read_fastq(pre + ".lr.fq")
    seqs, quals, tot, skipped = [], [], 0, 0
    for _, s, q in reads:
        if skipped < skip_bases:
            skipped += len(s); continue
        seqs.append(s); quals.append(q); tot += len(s)
        if tot >= sample_bases:
            break
    return pre, seqs, quals, tot, t_data


def _parity(pre, seqs, quals):
    from ratatosk_amd import api
    fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
    pg = api.Graph(fa, rt, 31, device=0)
    t0 = time.time(); got = pg.correct_batch(seqs, quals); t_gpu = time.time() - t0
    og = op.Graph(fa, rt, 31)
    t0 = time.time(); want, _ = og.correct_batch(seqs, quals, threads=os.cpu_count() or 4); t_cpu = time.time() - t0
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, "%d of %d reads differ from the oracle (first: %s)" % (len(bad), len(seqs), bad[:5])
    assert sum(1 for s, w in zip(seqs, want) if s != w[0]) > len(seqs) // 2  # the correction did something
    return pg.info(), t_gpu, t_cpu


@pytest.mark.gpu
def test_gpu_config1_size(tmp_path_factory):
    """configs[1]: 5 Mb reference, 30x PE150 (0.5 % substitutions), >= 16 Mb of ONT-profile long reads, index with SNP annotations."""
    pre, seqs, quals, tot, t_data = _sized(tmp_path_factory.mktemp("config1"), "c1",
                                           ["--seed", 2, "--ref-len", 5_000_000, "--sr-cov", 30, "--sr-err", 0.005, "--lr-cov", 4, "--lr-len", 8000, "--lr-profile", "ont", "--lr-err", 0.07],
                                           ["--snps"], 16_000_000)
    assert tot >= 16_000_000
    info, t_gpu, t_cpu = _parity(pre, seqs, quals)
    assert info.n_kmers > 5_000_000
    print("configs[1]: %d reads / %d bases, data %.0f s, HIP (host-inclusive) %.1f s, oracle %.1f s" % (len(seqs), tot, t_data, t_gpu, t_cpu))


@pytest.fixture(scope="module")
def config2_set(tmp_path_factory):
    """configs[2] / configs[3]: 60 Mb diploid reference (0.1 % heterozygous SNPs), 30x short reads, 8 Mb of its long reads (the short reads are
    kept: the second-pass index of configs[3] is built from them)."""
    return _sized(tmp_path_factory.mktemp("config2"), "c2",
                  ["--seed", 3, "--ref-len", 60_000_000, "--het", 0.001, "--sr-cov", 30, "--sr-err", 0.005, "--lr-cov", 0.15, "--lr-len", 8000, "--lr-profile", "ont", "--lr-err", 0.07],
                  [], 8_000_000, keep_short_reads=True)


@pytest.mark.gpu
def test_gpu_config2_graph_size(config2_set):
    """configs[2]'s graph: 60 Mb diploid reference (0.1 % heterozygous SNPs), 30x short reads; 8 Mb of its long reads."""
    pre, seqs, quals, tot, t_data = config2_set
    info, t_gpu, t_cpu = _parity(pre, seqs, quals)
    assert info.n_kmers > 60_000_000
    print("configs[2] graph: %d reads / %d bases, data %.0f s, HIP (host-inclusive) %.1f s, oracle %.1f s" % (len(seqs), tot, t_data, t_gpu, t_cpu))


@pytest.mark.gpu
def test_gpu_config3_two_passes_on_the_config2_graph(config2_set):
    """configs[3] at its graph size: `-1` on the 60 Mb diploid graph (HIP), the second index at k2 = 63 from the same 30x short reads coloured
    by the pass-1 reads (src/Ratatosk.cpp:1193,1227), then `-2` on >= 4 Mb of them: HIP against the oracle, sequence and quality of every read."""
    from ratatosk_amd import api
    pre, seqs, quals, tot, _ = config2_set
    n, acc = 0, 0
    while n < len(seqs) and acc < 4_200_000:
        acc += len(seqs[n]); n += 1
    seqs, quals = seqs[:n], quals[:n]
    assert acc >= 4_000_000
    pg1 = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0)
    out1 = pg1.correct_batch(seqs, quals)
    pg1.close()
    p1 = pre + ".pass1.fq"
    with open(p1, "w") as f:
        for i, (s, q) in enumerate(out1):
            f.write("@r%d\n%s\n+\n%s\n" % (i, s, q))
    t0 = time.time()
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", pre + ".sr.fq", "--colour-reads", p1, "-k", "63", "-o", pre + ".p2"], stderr=subprocess.DEVNULL)
    t_idx = time.time() - t0
    fa, rt = pre + ".p2.index.k63.fasta.gz", pre + ".p2.index.k63.rtsk"
    pg, og = api.Graph(fa, rt, 63, device=0), op.Graph(fa, rt, 63)
    s1, q1 = [o[0] for o in out1], [o[1] for o in out1]
    t0 = time.time(); got = pg.correct_batch(s1, q1, pg.opts(long_read_correct=1), raw=seqs); t_gpu = time.time() - t0
    t0 = time.time(); want = og.correct_batch2(s1, q1, seqs, og.opts(long_read_correct=1), threads=os.cpu_count() or 4); t_cpu = time.time() - t0
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if tuple(a) != tuple(b)]
    assert not bad, "%d of %d reads differ from the oracle (first: %s)" % (len(bad), len(seqs), bad[:5])
    assert sum(1 for a, b in zip(want, s1) if a[0] != b) > 0  # the second pass did something
    print("configs[3] on the configs[2] graph: %d reads / %d bases, second index %.0f s, HIP -2 (host-inclusive) %.1f s, oracle %.1f s" % (n, acc, t_idx, t_gpu, t_cpu))


@pytest.mark.gpu
def test_gpu_config4_scaled_down_graph_above_2_pow_28_kmers(tmp_path):
    """configs[4] (whole-genome-scale graph resident in HBM) at a tenth of its size, so that the paths only a big graph takes are in this tier: more
    than 2^28 k-mers (the k-mer table with a number of slots that is not a power of two, 64-bit offsets into the unitig pool and the half-k-mer
    lists), short reads sampled on the fly inside the index tool (no FASTQ on disk), index with SNP annotations built with the k-mers counted on
    the device. The oracle cannot hold such a graph: the checks are size-independent -- corrected reads are closer to the stretches of the
    reference they were simulated from than the raw reads (error rate at least halved, no read further away), more k-mer windows in the graph.
    profiles/scripts/config4_run.py is the same program the 3 Gb runs of profiles/r04_config4_dry_run.json and profiles/r05_config4_run.json used."""
    out = str(tmp_path / "c4.json")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "config4_run.py"), "300", "8", "64", "1"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RTK_C4_OUT=out, RTK_C4_DIR=str(tmp_path)))
    assert r.returncode == 0, (r.stderr[-1500:], open(out).read()[-1500:] if os.path.exists(out) else "")
    d = json.load(open(out))
    assert d["graph"]["kmers"] > (1 << 28) and d["graph"]["hbm_gb"] > 10
    pc = d["property_checks"]
    assert pc["reads_checked"] == 400 and pc["error_rate_corrected"] < 0.5 * pc["error_rate_raw"] and pc["reads_not_closer_to_truth"] <= 8
    assert pc["solid_window_share_corrected"] > 0.8 > pc["solid_window_share_raw"]
    print("configs[4] at 300 Mb: %.0f s in all (index %.0f s, load %.0f s), %.1f GB in HBM, %.3g bases/s" % (time.time() - t0, d["build_index_s"], d["graph_load_s"], d["graph"]["hbm_gb"], d["tickets"]["bases_per_s"]))
