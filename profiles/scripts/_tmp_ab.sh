cd /root/repo
run() { RTK_LIB_OVERRIDE=$1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['kernel_ms_per_step']['k_regions'])"; }
for i in 1 2; do
run $PWD/ratatosk_amd/libratatosk_hip.so
run $PWD/ratatosk_amd/_ab_4calls.so
done
