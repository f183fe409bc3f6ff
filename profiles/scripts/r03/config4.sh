#!/bin/bash
# configs[4] dry run on the GPU box: REF_MB megabases, 12x short reads, index with --gpu. Every step under its own timeout.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
df -h /tmp | tail -1 > gpurun_out/c4_env.txt; free -g | head -2 >> gpurun_out/c4_env.txt; nproc >> gpurun_out/c4_env.txt
REF_MB=${1:-1500}
RTK_LOAD_TRACE=1 RTK_INDEX_CAP=4000000000 RTK_C4_OUT=gpurun_out/c4_partial.json timeout 2700 python profiles/scripts/config4_dry_run.py $REF_MB 12 64 > gpurun_out/c4.json 2> gpurun_out/c4.err
echo "rc=$?" >> gpurun_out/c4_env.txt
grep "rtk load" gpurun_out/c4.err; tail -3 gpurun_out/c4.err; cat gpurun_out/c4_env.txt; head -c 3000 gpurun_out/c4.json
