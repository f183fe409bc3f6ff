cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "== shared queue, beside"; timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 0 64:512 128:512 128:1024 192:1024 256:1024 256:1536 2>&1 | grep -E "gap<|Error|error" ) > gpurun_out/r05_lanes_ab10_c1.log 2>&1
cat gpurun_out/r05_lanes_ab10_c1.log
