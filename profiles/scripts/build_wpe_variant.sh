#!/bin/bash
# Developer build: the library with another register budget for the region kernel AND the functions it calls.
#   profiles/scripts/build_wpe_variant.sh <waves per SIMD: 2|3|5|6|7> <out.so> ["extra -D flags"]
# amdgpu_waves_per_eu is a kernel attribute; the non-inlined wave programs keep the default budget (128 VGPRs), so the kernel's own
# attribute changes nothing. This script compiles the device side to LLVM IR, puts "amdgpu-waves-per-eu" (and a work-group size of
# 64) on every function, compiles the IR to a code object and embeds it in the host object (-fcuda-include-gpubinary).
# 5 waves per SIMD also needs 8 KB of LDS per wave: -DRTK_LDS_SET_CAP=1792u -DRTK_SLIM_HDR. Run the result with
# RTK_LIB_OVERRIDE=<out.so> RTK_REGION_WAVES=<256 * 4 * waves per SIMD>.
# Measured (round 2, bench.py --serial, k_regions per 64 Mb step): 4 waves/SIMD (default, 128 VGPRs) 37.8 ms; 5 (96 VGPRs, 5120 waves)
# 43.0 ms; 3 (168 VGPRs, 3072 waves) 46.1 ms; 2 (256 VGPRs, 2048 waves) 48.7 ms; default build at 3072 waves 43.3 ms.
set -e
V=$1; OUT=$(realpath -m $2); T=$(mktemp -d); cd "$(dirname "$0")/../../ratatosk_amd/csrc"
EXTRA=""; [ "$V" = 5 ] && EXTRA="-DRTK_LDS_SET_CAP=1472u -DRTK_CS_MAX_IDS=256u -DRTK_SLIM_HDR"; [ "$V" = 6 ] && EXTRA="-DRTK_LDS_SET_CAP=1152u -DRTK_CS_MAX_IDS=128u -DRTK_SLIM_HDR"; [ "$V" = 7 ] && EXTRA="-DRTK_LDS_SET_CAP=1024u -DRTK_CS_MAX_IDS=128u -DRTK_SLIM_HDR"; EXTRA="$EXTRA $3"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-result -DRTK_REGION_WPE=$V $EXTRA -I../../include"
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -emit-llvm -S -o $T/dev.ll hip/rtk_device.hip 2>/dev/null
python3 - $V $T <<'PY'
import sys
v, t = sys.argv[1], sys.argv[2]
out = []
for line in open(t + "/dev.ll").read().split("\n"):
    if line.startswith("attributes #") and '"target-cpu"' in line:
        line = line.replace('"amdgpu-flat-work-group-size"="1,1024"', '"amdgpu-flat-work-group-size"="1,64"')
        if '"amdgpu-flat-work-group-size"' not in line: line = line.rstrip()[:-1].rstrip() + ' "amdgpu-flat-work-group-size"="1,64" }'
        if '"amdgpu-waves-per-eu"' not in line: line = line.rstrip()[:-1].rstrip() + ' "amdgpu-waves-per-eu"="%s,%s" }' % (v, v)
    out.append(line)
open(t + "/dev_p.ll", "w").write("\n".join(out))
PY
/opt/rocm/lib/llvm/bin/clang -x ir $T/dev_p.ll -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -fPIC -o $T/dev.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.co -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c -o $T/host.o hip/rtk_device.hip 2>/dev/null
/opt/rocm/bin/hipcc -O3 --hip-link -shared -fPIC -o $OUT host/flat_graph.o hip/rtk_phase_long.o hip/rtk_index.o hip/rtk_graph_tables.o $T/host.o -lz -lpthread
rm -rf $T; echo "built $OUT"
