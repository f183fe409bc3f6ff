cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RTK_LANE_SERIAL=1 RTK_LANE_ROUND=1 RTK_TRACE=1 RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_laneprof.so timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 128:4096 > gpurun_out/r05_lanes_prof1_c1.log 2>&1
grep -E "lanes lap profile" gpurun_out/r05_lanes_prof1_c1.log | tail -1 | tr '[' '\n'
grep -E "gap<" gpurun_out/r05_lanes_prof1_c1.log | tail -1
