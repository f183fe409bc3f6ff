set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_correct.py -x -q -m gpu > gpurun_out/r05_gpu_correct.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gpu_correct.log
tail -5 gpurun_out/r05_gpu_correct.log
timeout 1200 python profiles/scripts/r05_lanes_ab.py c1 64000000 0 64 128 256 256:512 256:2048 > gpurun_out/r05_lanes_ab_c1.log 2>&1
tail -12 gpurun_out/r05_lanes_ab_c1.log
