cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 0 64 128 192 256 256:2048 > gpurun_out/r05_lanes_ab4_c1.log 2>&1
grep -E "gap<|Error|error" gpurun_out/r05_lanes_ab4_c1.log | tail -12
RTK_LANE_SERIAL=1 timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 128 256 256:2048 > gpurun_out/r05_lanes_ab4s_c1.log 2>&1
grep -E "gap<|Error|error" gpurun_out/r05_lanes_ab4s_c1.log | tail -12
RTK_LANE_SERIAL=1 RTK_TRACE=1 RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_laneprof.so timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 256 > gpurun_out/r05_lanes_prof_c1.log 2>&1
grep -E "wave-time" gpurun_out/r05_lanes_prof_c1.log | tail -1 | tr '[' '\n'
