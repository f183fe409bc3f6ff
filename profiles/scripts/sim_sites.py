#!/usr/bin/env python
"""Developer statistics on the CPU simulator: alignments of the region stage by call site (calls, word-columns, mean m / n, stored sweeps),
DFS tree nodes, candidates per DFS call. Usage: sim_sites.py PREFIX [max_reads]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ratatosk_amd import api
from oracle import oracle_py as op
lib = os.path.join(ROOT, "tests", "hostsim", "librtk_hostsim.so")
L = api.load_library(lib)
pre = sys.argv[1]; mx = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0, lib_path=lib)
reads = op.read_fastq(pre + ".lr.fq")[:mx]
seqs, quals = [r[1] for r in reads], [r[2] for r in reads]
buf = (C.c_ulonglong * 256)()
L.rtk_sim_site_stats(buf, 1)
b = api.Batch(g, seqs, quals); b.run(g.opts()); st = b.stats(); b.close()
L.rtk_sim_site_stats(buf, 0)
names = {0: "other", 1: "score terminal NW", 2: "score nonterm HW (ref in path)", 3: "score nonterm HW (path in ref)", 4: "path qual SHW path", 5: "explore prefix SHW", 6: "select nt HW", 7: "resize SHW", 8: "fixRepeats NW", 9: "fixRepeats NW k", 10: "final select NW", 11: "partial select SHW (restart)", 12: "partial select SHW (final)", 13: "trim SHW", 14: "consensus fw NW path", 15: "consensus bw NW path", 16: "consensus final NW", 17: "fixAmbiguity SHW path"}
print("bases %d regions %d aligns %d expansions %d" % (st["in_bases"], st["n_regions"], st["n_align"], st["n_expand"]))
tot = sum(buf[8 * i + 1] for i in range(20))
for i in range(20):
    c = buf[8 * i]
    if c:
        print("site %2d %-32s calls %7d cols32 %10d (%.3f) mean m %6.1f n %6.1f stored %7d bounded %7d" % (i, names.get(i, "?"), c, buf[8 * i + 1], buf[8 * i + 1] / max(1, tot), buf[8 * i + 2] / c, buf[8 * i + 3] / c, buf[8 * i + 4], buf[8 * i + 6]))
print("DFS nodes %d, columns added %d; DFS calls %d; second walks %d" % (buf[8 * 20], buf[8 * 20 + 1], buf[8 * 21], buf[8 * 28]))
print("aligns per DFS call histogram:", [buf[8 * (22 + (b >> 3)) + (b & 7)] for b in range(16)])
