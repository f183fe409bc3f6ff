cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for R in 64 32; do
echo "== RTK_LANE_ROUND=$R beside"
RTK_LANE_ROUND=$R timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 0 64:1024 128:1024 128:2048 192:2048 192:4096 2>&1 | grep -E "gap<|Error|error" | tail -12
done > gpurun_out/r05_lanes_ab6_c1.log 2>&1
cat gpurun_out/r05_lanes_ab6_c1.log
