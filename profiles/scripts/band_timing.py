"""Latency of ONE whole-read NW path alignment (raw read against corrected read, the alignment of phasing(), src/Graph.cpp:975) through the
stage entry, per read length and workgroup size. One problem per launch, so the kernel's duration in the rocprofv3 trace is the latency;
the host-side time printed here includes allocation and copies. Usage: band_timing.py [lengths...]"""
import random, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ratatosk_amd import api
rnd = random.Random(3)
def mutate(q, err):
    out = []
    for c in q:
        r = rnd.random()
        if r < err / 3: continue
        if r < 2 * err / 3: out.append(rnd.choice("ACGT")); out.append(c); continue
        if r < err: out.append(rnd.choice("ACGT")); continue
        out.append(c)
    return "".join(out)
lens = [int(x) for x in sys.argv[1:]] or [10000, 40000, 98000]
api.myers_batch(["ACGT"], ["ACGT"])
for L in lens:
    q = "".join(rnd.choice("ACGT") for _ in range(L)); t = mutate(q, 0.1)
    ref = None
    for waves in (1, 4, 8, 16):
        t0 = time.time(); r = api.myers_batch([q], [t], [-1], [0], want_path=True, waves=waves)[0]; dt = time.time() - t0
        if ref is None: ref = r
        print("len %6d waves %2d: dist %6d host %.1f ms same=%s" % (L, waves, r[0], dt * 1e3, r == ref), flush=True)
