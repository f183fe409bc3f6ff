cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "== c2 (60 Mb diploid), beside with shared queue"; timeout 2400 python profiles/scripts/r05_lanes_ab.py c2 64000000 0 128:512 128:1024 192:1024 2>&1 | grep -E "gap<|Error|error|batch"
  echo "== c2 serial"; RTK_LANE_SERIAL=1 timeout 1200 python profiles/scripts/r05_lanes_ab.py c2 64000000 128:1024 2>&1 | grep -E "gap<|Error|error" ) > gpurun_out/r05_lanes_ab_c2.log 2>&1
cat gpurun_out/r05_lanes_ab_c2.log
