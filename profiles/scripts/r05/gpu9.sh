cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for R in 1 4 8; do
echo "== RTK_LANE_ROUND=$R serial, 4096 waves"
RTK_LANE_SERIAL=1 RTK_LANE_ROUND=$R timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 128:4096 2>&1 | grep -E "gap<|Error|error" | tail -12
done > gpurun_out/r05_lanes_ab8_c1.log 2>&1
cat gpurun_out/r05_lanes_ab8_c1.log
