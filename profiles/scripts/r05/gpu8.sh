cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for RW in 3072 2560 2048; do
echo "== RTK_REGION_WAVES=$RW beside"
RTK_REGION_WAVES=$RW timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 0 64:1024 128:1024 192:1536 192:2048 256:2048 2>&1 | grep -E "gap<|Error|error" | tail -12
done > gpurun_out/r05_lanes_ab7_c1.log 2>&1
cat gpurun_out/r05_lanes_ab7_c1.log
