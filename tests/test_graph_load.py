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
