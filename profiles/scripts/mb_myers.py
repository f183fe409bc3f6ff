"""Micro-benchmark of the HIP Myers kernel (k_myers_batch): many equal-sized SHW distance problems; run under rocprofv3 for kernel time."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ratatosk_amd import api
random.seed(1)
def rs(n): return "".join(random.choice("ACGT") for _ in range(n))
for (m, n, cnt) in [(60, 60, 40000), (300, 300, 40000), (1000, 1000, 20000), (300, 1000, 20000)]:
    base = [rs(m) for _ in range(200)]
    Q = [base[i % 200] for i in range(cnt)]
    T = [(base[i % 200] + rs(max(0, n - m)))[:n] for i in range(cnt)]
    api.myers_batch(Q[:100], T[:100], [-1] * 100, [1] * 100)
    t = time.time(); api.myers_batch(Q, T, [-1] * cnt, [1] * cnt); dt = time.time() - t
    print("m=%d n=%d cnt=%d wall=%.3fs" % (m, n, cnt, dt))
