cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "== contiguous per-lane layout, serial"; RTK_LANE_SERIAL=1 timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 128:1024 256:2048 2>&1 | grep -E "gap<|Error|error"
  echo "== round 1"; RTK_LANE_SERIAL=1 RTK_LANE_ROUND=1 timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 128:4096 2>&1 | grep -E "gap<|Error|error" ) > gpurun_out/r05_lanes_ab9_c1.log 2>&1
cat gpurun_out/r05_lanes_ab9_c1.log
