"""Developer micro-benchmark (round 5): the whole-read NW path alignment of phasing() (raw read against corrected read, ~7 % apart) by read length, one wave per problem
(rtk_myers_batch with paths = the route k_phase takes), kernel time by HIP events (RTK_MYERS_TIME) and the cycle shares of the Hirschberg driver (RTK_MYERS_PROF).
Usage: python profiles/scripts/r05_phase_align_micro.py [N_PER_CASE=1024]"""
import ctypes as C, os, random, sys, time
sys.path.insert(0, ".")
os.environ["RTK_MYERS_TIME"] = "1"; os.environ["RTK_MYERS_PROF"] = "1"
from ratatosk_amd import api
L = api.load_library()
random.seed(5)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def mut(s, e):
    out = []
    for ch in s:
        r = random.random()
        if r < e * 0.4: continue
        if r < e * 0.7: out.append(random.choice("ACGT")); out.append(ch); continue
        if r < e: out.append(random.choice("ACGT")); continue
        out.append(ch)
    return "".join(out)
for m, e in ((2000, 0.07), (4000, 0.07), (8000, 0.07), (16000, 0.07), (24000, 0.07), (8000, 0.02), (8000, 0.12)):
    base = ["".join(random.choice("ACGT") for _ in range(m)) for _ in range(32)]
    qs = [mut(b, e).encode() for b in base]; ts = [b.encode() for b in base]
    n = N if m <= 8000 else max(256, N * 8000 // m)
    qa = (C.c_char_p * n)(*[qs[i % 32] for i in range(n)]); ta = (C.c_char_p * n)(*[ts[i % 32] for i in range(n)])
    ql = (C.c_uint32 * n)(*[len(qs[i % 32]) for i in range(n)]); tl = (C.c_uint32 * n)(*[len(ts[i % 32]) for i in range(n)])
    ka = (C.c_int32 * n)(*([-1] * n)); ma = (C.c_int32 * n)(*([0] * n))
    dist = (C.c_int32 * n)(); nloc = (C.c_int32 * n)()
    cap = 8 * m; cig = C.create_string_buffer(n * cap)
    for rep in range(2):
        t0 = time.time()
        rc = L.rtk_myers_batch(n, qa, ql, ta, tl, ka, ma, 1, 1, dist, nloc, None, 0, cig, cap)
        print("m=%d err=%.2f n=%d rep %d: rc=%d %.1f ms wall, dist[0]=%d (%.1f %% of m)" % (m, e, n, rep, rc, 1e3 * (time.time() - t0), dist[0], 100.0 * dist[0] / m), flush=True)
