"""Throughput of the two schedules for the small alignments of the region program: one wave per problem (rtk_myers_batch: the route k_regions takes, one 32- or 64-bit
word of the query per lane) against one LANE per problem (rtk_myers_batch_lanes). Problems shaped like those of a 64 Mb step of configs[1] (DESIGN_HISTORY.md section 3.5:
1.16 M alignments of 536 word-columns on average, 80 % of the regions with gaps under 256 bases): queries of 100-320 characters against targets of about the
same length at 10 % divergence, NW and SHW, distances and paths. RTK_MYERS_TIME=1 makes the library print the kernel times (HIP events)."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RTK_MYERS_TIME"] = "1"
from ratatosk_amd import api  # noqa: E402

rnd = random.Random(7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
qs, ts, ks, ms = [], [], [], []
for i in range(n):
    m = rnd.randrange(100, 321)
    q = "".join(rnd.choice("ACGT") for _ in range(m))
    t = "".join((rnd.choice("ACGT") if rnd.random() < 0.1 else c) for c in q if rnd.random() > 0.03)
    qs.append(q); ts.append(t); ks.append(-1); ms.append(i & 1)
words = sum(((len(q) + 63) // 64) * len(t) for q, t in zip(qs, ts))
print("%d problems, %.3g word-columns (%.0f per problem)" % (n, words, words / n), flush=True)
for rep in range(2):
    t0 = time.time(); a = api.myers_batch(qs, ts, ks, ms, use_iupac=False, lanes=True); t1 = time.time()
    b = api.myers_batch(qs, ts, ks, ms, use_iupac=False); t2 = time.time()
    assert [(x[0], x[1]) for x in a] == [(x[0], x[1]) for x in b]
    print("rep %d, distances: same results; wall clock of the calls (pool build, copies, kernel): lanes %.2f s, waves %.2f s" % (rep, t1 - t0, t2 - t1), flush=True)
for rep in range(2):
    t0 = time.time(); a = api.myers_batch(qs, ts, ks, ms, want_path=True, use_iupac=False, lanes=True); t1 = time.time()
    b = api.myers_batch(qs, ts, ks, ms, want_path=True, use_iupac=False); t2 = time.time()
    assert a == b
    print("rep %d, with paths: same results (CIGARs too); wall clock of the calls: lanes %.2f s, waves %.2f s" % (rep, t1 - t0, t2 - t1), flush=True)
