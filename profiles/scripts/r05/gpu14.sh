cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "== 256 Mb batch, beside (shared queue)"; timeout 1500 python profiles/scripts/r05_lanes_ab.py c1 256000000 0 128:2048 128:4096 192:4096 2>&1 | grep -E "gap<|Error|error|batch"
  echo "== serial"; RTK_LANE_SERIAL=1 timeout 1500 python profiles/scripts/r05_lanes_ab.py c1 256000000 128:4096 256:4096 2>&1 | grep -E "gap<|Error|error" ) > gpurun_out/r05_lanes_ab_256M.log 2>&1
cat gpurun_out/r05_lanes_ab_256M.log
