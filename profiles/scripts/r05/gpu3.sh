cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 0 128 256 256:2048 > gpurun_out/r05_lanes_ab2_c1.log 2>&1
grep -E "gap<|Error|error" gpurun_out/r05_lanes_ab2_c1.log | tail -12
