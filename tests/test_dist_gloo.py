"""N > 1 path on CPU: two processes (gloo), graph loaded by rank 0 and replicated by per-buffer broadcast, reads sharded by
ticket; every rank must reproduce the oracle on its shard. Uses the host simulator for the device programs (no GPU here)."""
import os
import subprocess
import sys

from conftest import ROOT, SIM_LIB

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from ratatosk_amd import api, dist as rdist
from oracle import oracle_py as op
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
pre = %(pre)r
fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
g = rdist.load_graph_replicated(fa, rt, 31, rank, world, 0, lib_path=%(sim)r)
reads = op.read_fastq(pre + ".lr.fq")[:8]
tickets = [reads[i:i + 2] for i in range(0, len(reads), 2)]
mine = rdist.shard_tickets(len(tickets), rank, world)
assert mine == [i for i in range(len(tickets)) if i %% world == rank]
og = op.Graph(fa, rt, 31)
for t in mine:
    seqs = [r[1] for r in tickets[t]]; quals = [r[2] for r in tickets[t]]
    got = g.correct_batch(seqs, quals)
    want, _ = og.correct_batch(seqs, quals)
    assert got == want, (rank, t)
info = g.info()
assert info.n_unitigs == og.n_unitigs
dist.barrier()
open(os.path.join(%(out)r, "rank%%d.ok" %% rank), "w").write(str(len(mine)))
'''


def test_two_ranks_gloo_replicated_graph_sharded_reads(ds_small, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, pre=ds_small, sim=SIM_LIB, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29511", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").read_text() == "2" and (tmp_path / "rank1.ok").read_text() == "2"
