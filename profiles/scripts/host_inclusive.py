"""Host-buffers-in / host-buffers-out rate of the boundary call (rtk_batch_create + run + fetch = rtk_correct_batch), i.e. including
the packing, the H2D / D2H copies over PCIe and the unpacking. Two overlapped callers, like the CLI. Not the bench `value`."""
import os, sys, time, tempfile, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ratatosk_amd import api
api.load_library(None)
wd = tempfile.mkdtemp(prefix="rtk_hi_")
pre = bench.make_dataset(wd, 5_000_000, 150_000_000)
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 128_000_000)
tickets, cs, cq, cur = [], [], [], 0
for s, q in zip(seqs, quals):
    cs.append(s.encode()); cq.append(q.encode()); cur += len(s)
    if cur >= 32_000_000:
        tickets.append((cs, cq, cur)); cs, cq, cur = [], [], 0
g.correct_batch(*tickets[0][:2])  # warm-up (scratch allocation)
for workers in (1, 2):
    it = iter(tickets); lock = threading.Lock(); done = [0]
    def work():
        while True:
            with lock:
                t = next(it, None)
            if t is None:
                return
            g.correct_batch(t[0], t[1])
            with lock:
                done[0] += t[2]
    t0 = time.time()
    th = [threading.Thread(target=work) for _ in range(workers)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.time() - t0
    print("host-inclusive, %d caller(s): %.1f M bases/s (%d tickets, %.0f ms per 32 Mb ticket)" % (workers, done[0] / dt / 1e6, len(tickets), 1e3 * dt / len(tickets)))
