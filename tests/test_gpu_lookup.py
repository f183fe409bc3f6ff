"""K1 exact k-mer lookup kernel against the oracle's searchSequence restatement ([A1]); bit-exact (unitig, dist, strand)."""
import pytest

from oracle import oracle_py as op
from ratatosk_amd import api

pytestmark = pytest.mark.gpu


def _check(prefix):
    fa, rt = prefix + ".index.k31.fasta.gz", prefix + ".index.k31.rtsk"
    og, pg = op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0)
    reads = op.read_fastq(prefix + ".lr.fq")
    for name, s, q in reads[:6] + [("short", "ACGT" * 5, ""), ("withN", reads[0][1][:200] + "N" + reads[0][1][200:400], "")]:
        assert pg.lookup_exact(s) == og.exact(s), name
    # a unitig itself: every window must hit, on the forward strand, at consecutive offsets
    u = og.unitig(0)
    hits = pg.lookup_exact(u["seq"])
    assert all(h >= 0 for h in hits) and [((h >> 1) & 0xFFFFFFFF) for h in hits] == list(range(len(hits)))


def test_gpu_exact_lookup_branching(ds_small):
    _check(ds_small)


def test_gpu_exact_lookup_clean(ds_clean):
    _check(ds_clean)
