"""Developer micro-benchmark: wave time of one alignment by size / mode on the route the region program takes (no end-location
list). Run under rocprofv3 --kernel-trace: one k_myers_batch dispatch per case, 1024 waves, N / 1024 alignments per wave."""
import ctypes as C, random, sys, time
sys.path.insert(0, ".")
from ratatosk_amd import api
L = api.load_library()
random.seed(3)
N = 131072
def mut(s, e):
    out = []
    for ch in s:
        r = random.random()
        if r < e / 3: continue
        if r < 2 * e / 3: out.append(random.choice("ACGT")); out.append(ch); continue
        if r < e: out.append(random.choice("ACGT")); continue
        out.append(ch)
    return "".join(out)
base = ["".join(random.choice("ACGT") for _ in range(1300)) for _ in range(64)]
cases = [(40, 1, 0), (80, 1, 0), (150, 1, 0), (300, 1, 0), (600, 1, 0), (1200, 1, 0), (150, 0, 0), (150, 1, 1), (300, 1, 1), (600, 1, 1), (150, 0, 1)]
for m, mode, path in cases:
    qs = [base[i][:m].encode() for i in range(64)]; ts = [mut(base[i][:m], 0.08).encode() for i in range(64)]
    qa = (C.c_char_p * N)(*[qs[i % 64] for i in range(N)]); ta = (C.c_char_p * N)(*[ts[i % 64] for i in range(N)])
    ql = (C.c_uint32 * N)(*[len(qs[i % 64]) for i in range(N)]); tl = (C.c_uint32 * N)(*[len(ts[i % 64]) for i in range(N)])
    ka = (C.c_int32 * N)(*([-1] * N)); ma = (C.c_int32 * N)(*([mode] * N))
    dist = (C.c_int32 * N)(); nloc = (C.c_int32 * N)()
    t0 = time.time()
    rc = L.rtk_myers_batch(N, qa, ql, ta, tl, ka, ma, path, 1, dist, nloc, None, 0, None, 0)
    print("m=%d mode=%d path=%d: rc=%d %.1f ms wall, dist[0]=%d" % (m, mode, path, rc, 1e3 * (time.time() - t0), dist[0]), flush=True)
