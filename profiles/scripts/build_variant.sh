#!/bin/bash
# Developer build of the library with extra compile flags for the main device translation unit:
#   profiles/scripts/build_variant.sh <name> "<flags>"   ->  ratatosk_amd/variants/libratatosk_hip_<name>.so
# Run it with RTK_LIB_OVERRIDE=<that file> (ratatosk_amd/api.py, bench.py).
set -e
NAME=$1; FL=$2; cd "$(dirname "$0")/../../ratatosk_amd/csrc"; mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-result $FL -I../../include"
/opt/rocm/bin/hipcc $FLAGS -c -o ../variants/rtk_device_$NAME.o hip/rtk_device.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libratatosk_hip_$NAME.so host/flat_graph.o ../variants/rtk_device_$NAME.o hip/rtk_phase_long.o hip/rtk_index.o hip/rtk_graph_tables.o -lz -lpthread
rm -f ../variants/rtk_device_$NAME.o; echo "built ratatosk_amd/variants/libratatosk_hip_$NAME.so"
