#!/bin/bash
# Registers, scratch, LDS and occupancy of every kernel of the product library as the compiler reports them (-Rpass-analysis=kernel-resource-usage; runs without a GPU).
#   usage: profiles/scripts/resource_usage.sh rNN   ->  profiles/rNN_kernel_resource_usage.txt
set -e
R=${1:-r05}; cd "$(dirname "$0")/../../ratatosk_amd/csrc"; T=$(mktemp -d)
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-result -I../../include"
for f in rtk_device rtk_phase_long rtk_index rtk_graph_tables; do [ -f hip/$f.hip ] && /opt/rocm/bin/hipcc $FLAGS -c -o $T/$f.o hip/$f.hip -Rpass-analysis=kernel-resource-usage 2> $T/$f.txt || true; done
python3 - $T > ../../profiles/${R}_kernel_resource_usage.txt <<'PY'
import re, sys, glob, subprocess
print("kernel resource usage, gfx950, flags of ratatosk_amd/csrc/Makefile (hipcc -Rpass-analysis=kernel-resource-usage; profiles/scripts/resource_usage.sh)")
print("occupancy = waves per SIMD the registers allow (x4 SIMDs per CU); LDS per workgroup (160 KB per CU: 16 one-wave workgroups need <= 10240 B each)\n")
for f in sorted(glob.glob(sys.argv[1] + "/*.txt")):
    txt = open(f).read()
    print("== hip/" + f.split("/")[-1].replace(".txt", ".hip"))
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].split(" ")[0]
        m = re.match(r"_ZN?(?:12_GLOBAL__N_1)?(\d+)", name)  # the kernel's own name out of the mangled one: length-prefixed
        if m: name = name[m.end():m.end() + int(m.group(1))]
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        if not name.startswith("k_"): lib = locals().get("lib", 0) + 1; continue   # (rocPRIM sort / scan / select instances of the index and table builders: counted, not listed)
        print("  %-34s VGPRs %3s  AGPRs %3s  SGPRs %3s  scratch %5s B/lane  occupancy %s waves/SIMD  LDS %6s B" % (name[:34], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
PY
cat ../../profiles/${R}_kernel_resource_usage.txt
