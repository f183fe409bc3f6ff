import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIM_LIB = os.path.join(ROOT, "tests", "hostsim", "librtk_hostsim.so")
BIN = os.path.join(ROOT, "ratatosk_amd", "bin")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(BIN, "rtk_simulate"), os.path.join(BIN, "rtk_build_index"), os.path.join(BIN, "rtk_gunzip"), SIM_LIB,
            os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "ratatosk_amd", "libratatosk_hip.so"), os.path.join(BIN, "Ratatosk")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


_ensure_built()


def make_dataset(tmpdir, name, sim_args, index_args=()):
    """Synthetic reference/short/long reads + index, via the repo's own tools. Returns the path prefix."""
    pre = os.path.join(str(tmpdir), name)
    subprocess.check_call([os.path.join(BIN, "rtk_simulate"), "--prefix", pre] + [str(a) for a in sim_args], stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre] + [str(a) for a in index_args], stderr=subprocess.DEVNULL)
    return pre


@pytest.fixture(scope="session")
def ds_small(tmp_path_factory):
    """30 kb diploid reference with two-copy repeats: branching graph, global colour sets, 10 % error long reads."""
    d = tmp_path_factory.mktemp("ds_small")
    return make_dataset(d, "small", ["--seed", 11, "--ref-len", 30000, "--het", 0.004, "--repeat-frac", 0.1, "--sr-cov", 40, "--sr-err", 0.01,
                                     "--lr-n", 12, "--lr-len", 3000, "--lr-profile", "ont", "--lr-err", 0.08], ["--global-cov-factor", 1.2])


@pytest.fixture(scope="session")
def ds_clean(tmp_path_factory):
    """Config-1-like plumbing set: clean haploid graph with long unitigs (same-unitig shortcut, Hirschberg-sized sub-paths)."""
    d = tmp_path_factory.mktemp("ds_clean")
    return make_dataset(d, "clean", ["--seed", 1, "--ref-len", 50000, "--sr-cov", 30, "--sr-err", 0.002, "--lr-n", 10, "--lr-len", 5000, "--lr-err", 0.10])


@pytest.fixture(scope="session")
def ds_k25(tmp_path_factory):
    """k = 25 graph (the reference accepts any odd k1 <= 31 for pass 1, src/Ratatosk.cpp:213)."""
    d = tmp_path_factory.mktemp("ds_k25")
    return make_dataset(d, "k25", ["--seed", 5, "--ref-len", 20000, "--het", 0.003, "--sr-cov", 40, "--sr-err", 0.005, "--lr-n", 6, "--lr-len", 2500, "--lr-profile", "ont", "--lr-err", 0.08], ["-k", 25])


@pytest.fixture(scope="session")
def ds_k21(tmp_path_factory):
    """k = 21: half-k-mers of 10 bases, so the 1-edit search sees many chance candidates per lookup; SNP annotations on."""
    d = tmp_path_factory.mktemp("ds_k21")
    return make_dataset(d, "k21", ["--seed", 9, "--ref-len", 60000, "--het", 0.004, "--repeat-frac", 0.05, "--sr-cov", 40, "--sr-err", 0.005,
                                   "--lr-n", 16, "--lr-len", 3000, "--lr-profile", "ont", "--lr-err", 0.08], ["-k", 21, "--snps"])


@pytest.fixture(scope="session")
def ds_medium(tmp_path_factory):
    """GPU-tier set: 400 kb diploid reference with repeats, 160 ONT-profile reads (~1.3 Mb): every branch of the region program at volume."""
    d = tmp_path_factory.mktemp("ds_medium")
    return make_dataset(d, "medium", ["--seed", 7, "--ref-len", 400000, "--het", 0.002, "--repeat-frac", 0.05, "--sr-cov", 30, "--sr-err", 0.005,
                                      "--lr-n", 160, "--lr-len", 8000, "--lr-profile", "ont", "--lr-err", 0.07], ["--global-cov-factor", 1.5])


@pytest.fixture(scope="session")
def ds_tandem(tmp_path_factory):
    """Tandem repeats (unit 7..45 bp spanning > 2k, one haplotype a unit short): unitigs on short cycles, fixRepeats has work."""
    d = tmp_path_factory.mktemp("ds_tandem")
    return make_dataset(d, "tandem", ["--seed", 21, "--ref-len", 60000, "--het", 0.003, "--tandem", 30, "--sr-cov", 40, "--sr-err", 0.005,
                                      "--lr-n", 40, "--lr-len", 3000, "--lr-profile", "ont", "--lr-err", 0.07])


@pytest.fixture(scope="session")
def ds_snps(tmp_path_factory):
    """Diploid set whose index carries SNP annotations (`rtk_build_index --snps`): getAmbiguityVector / fixAmbiguity have work.
    PREFIX_plain.index.* is the same graph without the annotations."""
    d = tmp_path_factory.mktemp("ds_snps")
    pre = make_dataset(d, "snps", ["--seed", 31, "--ref-len", 60000, "--het", 0.004, "--sr-cov", 40, "--sr-err", 0.005,
                                   "--lr-n", 40, "--lr-len", 3000, "--lr-profile", "ont", "--lr-err", 0.07], ["--snps"])
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre + "_plain"], stderr=subprocess.DEVNULL)
    return pre


@pytest.fixture(scope="session")
def ds_snps_rich(tmp_path_factory):
    """Everything at once: diploid (0.6 % SNPs), two-copy repeats, tandem repeats, global colour sets, index with SNP AND short-cycle
    annotations."""
    d = tmp_path_factory.mktemp("ds_snps_rich")
    return make_dataset(d, "rich", ["--seed", 41, "--ref-len", 200000, "--het", 0.006, "--repeat-frac", 0.05, "--tandem", 10, "--sr-cov", 35, "--sr-err", 0.005,
                                    "--lr-n", 80, "--lr-len", 5000, "--lr-profile", "ont", "--lr-err", 0.07], ["--snps", "--global-cov-factor", 1.5])


def golden_rows():
    path = os.path.join(ROOT, "tests", "golden", "edlib_golden.tsv")
    rows = []
    with open(path) as f:
        for line in f:
            q, t, k, mode, want_path, d, locs, cig = line.rstrip("\n").split("\t")
            rows.append(dict(q="" if q == "-" else q, t="" if t == "-" else t, k=int(k), mode=int(mode), path=bool(int(want_path)), d=int(d),
                             locs=[] if locs == "-" else [int(x) for x in locs.split(",")], cigar="" if cig == "-" else cig))
    return rows
