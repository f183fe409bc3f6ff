cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for R in 64 32 16; do
for S in 0 1; do
echo "== RTK_LANE_ROUND=$R RTK_LANE_SERIAL=$S"
RTK_LANE_ROUND=$R RTK_LANE_SERIAL=$S timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 128:2048 192:4096 256:4096 2>&1 | grep -E "gap<|Error|error" | tail -12
done; done > gpurun_out/r05_lanes_ab5_c1.log 2>&1
cat gpurun_out/r05_lanes_ab5_c1.log
