"""Device Myers programs executed by the host simulator (1-lane wave) against the reference golden vectors.
Checks the scalar logic (distance, locations, traceback, Hirschberg split); the wave-parallel lanes are checked on the GPU."""
from conftest import SIM_LIB, golden_rows
from ratatosk_amd import api


def test_sim_myers_golden():
    rows = golden_rows()
    dist = [r for r in rows if not r["path"]]
    res = api.myers_batch([r["q"] for r in dist], [r["t"] for r in dist], [r["k"] for r in dist], [r["mode"] for r in dist], want_path=False, lib_path=SIM_LIB)
    for r, (d, locs, _) in zip(dist, res):
        assert d == r["d"] and locs == r["locs"]
    path = [r for r in rows if r["path"]]
    res = api.myers_batch([r["q"] for r in path], [r["t"] for r in path], [r["k"] for r in path], [r["mode"] for r in path], want_path=True, lib_path=SIM_LIB)
    for r, (d, locs, cig) in zip(path, res):
        assert d == r["d"] and locs == r["locs"] and cig == r["cigar"]
