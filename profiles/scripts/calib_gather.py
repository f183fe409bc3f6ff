"""Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for this path's access pattern (MI355X_MICROARCH.md, HBM section:
"calibrate on a known byte count in your own access pattern"). Two torch kernels with known byte counts:
  1. random 8-byte gather of N int64 out of a table far larger than L2+Infinity Cache  (the k-mer filter / table pattern)
  2. a wide coalesced copy of the same number of elements                              (the streaming pattern)
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` with --kernel-trace; profiles/scripts/summarise.py reads the result."""
import torch

N = 1 << 26          # 64 Mi accesses
T = 1 << 30          # 8 GiB table of int64
torch.manual_seed(1)
tab = torch.arange(T, dtype=torch.int64, device="cuda")
idx = torch.randint(0, T, (N,), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    out = tab[idx]                 # indexing kernel: N random 8-B reads + 8N index bytes read + 8N written
    torch.cuda.synchronize()
for _ in range(3):
    cp = tab[: N].clone()          # copy kernel: 8N read + 8N written, coalesced
    torch.cuda.synchronize()
print("calib done", int(out[0]) == int(idx[0]), N, T)
