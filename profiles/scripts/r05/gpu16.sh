cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in "" wpe5 wpe4; do
  echo "== variant '$V'"
  L=""; [ -n "$V" ] && L=$PWD/ratatosk_amd/variants/libratatosk_hip_$V.so
  RTK_LIB_OVERRIDE=$L python bench.py --no-cpu-baseline --no-host-legs --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 ms_per_step %.2f' % d['ms_per_step'], d['roofline']['kernel_ms_per_step']); print('c1 %.2f' % d['config1']['ms_per_step'], d['config1']['kernel_ms_per_step'])"
done > gpurun_out/r05_inexact_wpe.txt 2>&1
cat gpurun_out/r05_inexact_wpe.txt
