#!/bin/bash
# batch arenas + buffers reserved ahead: parity (correction, pass 2, CLI, concurrent callers), then the second pass five times and the first-pass CLI / bench line quickly
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_correct.py tests/test_pass2.py tests/test_cli.py tests/test_coalesce.py tests/test_capi.py tests/test_gpu_seeds.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_arena_tests.txt; cat gpurun_out/r06_arena_tests.txt
bash profiles/scripts/r06/pass2_trace.sh > /dev/null 2>&1; cp gpurun_out/r06_pass2_trace.txt gpurun_out/r06_pass2_trace_arena.txt; grep -E "== run|new device" gpurun_out/r06_pass2_trace_arena.txt
timeout 900 python bench.py --workdir /tmp/rtk_wd --no-cpu-baseline --no-config4 --no-config1-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g cli %.4g host_inclusive %.4g second_pass %.4g' % (d['value'], d['cli_file_to_file']['value'], d['host_inclusive']['value'], d['second_pass']['value'])); [print(k, {c: '%.3g' % v[c] for c in v if c.startswith('callers') and isinstance(v[c], float)}) for k, v in d['by_ticket_size'].items() if isinstance(v, dict)]" > gpurun_out/r06_arena_bench.txt 2>&1; cat gpurun_out/r06_arena_bench.txt
