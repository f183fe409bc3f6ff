// phasing() of the LONG reads of a second-pass ticket (reference: src/Graph.cpp:869-1097) on workgroups of several waves.
// The step aligns the whole uncorrected read against the whole corrected one (:975): quadratic in the read length, and with one wave
// per read a launch lasts as long as its longest read (1.2 s for a 100 kb read against 30 ms of work per wave on average). Here wave 0
// of a workgroup runs the same read program as k_phase and the other waves wait for its big alignment passes, whose 4096-row blocks
// they sweep at the same time, one 64-column chunk behind each other (rtk_myers.h, RTK_MULTIWAVE).
// Own translation unit: every device function is compiled a second time with workgroup barriers turned into fence + wave barrier
// (rtk_wave.h), because a program that runs on one wave of a bigger workgroup must never wait for the others at an s_barrier.
#define RTK_MULTIWAVE 1
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/ratatosk_hip.h"
#include "rtk_mem.h"
#include "rtk_myers.h"
#include "rtk_types.h"
#include "rtk_wave.h"
#include "rtk_seeds.h"
#include "rtk_region.h"
#include "rtk_phase_long.h"

__global__ void __launch_bounds__(1024) k_phase_long(const LaunchCtx* L, GraphView g, OptsView o, BatchView bv, PhaseView pv, char* scratch, uint64_t stride, RegionScratchCfg cfg, const uint32_t* list, uint32_t n_list, int only_flagged) {
    const int wave = static_cast<int>(threadIdx.x) >> 6;
    RtkCoop* st = rtk_coop();
    if (threadIdx.x == 0) { st->seq = 0; st->n_done = 0; st->exit_flag = 0; st->n_waves = static_cast<int>(blockDim.x) >> 6; }
    __syncthreads(); // the only workgroup barrier of the kernel: every wave is here
    if (wave != 0) { rtk_myers_coop_helper(wave); return; }
    __shared__ RegionScratch hdr;
    RegionScratch* sc = region_scratch_carve(scratch + static_cast<uint64_t>(blockIdx.x) * stride, cfg, &hdr);
    rtk_sync();
    (void)o;
    RCtx c = {L->g, L->o, L->bv, L->rb, {rtk_opaque(sc)}, {g.k}};
    for (uint32_t ri = blockIdx.x; ri < n_list; ri += gridDim.x) {
        const uint32_t r = list[ri];
        if (only_flagged && bv.status[r] == 0) continue;
        *sc->overflow = 0; sc->top[0] = 0;
        rtk_phase_read(c, pv, r);
        if (rtk_lane() == 0) bv.status[r] = *sc->overflow;
    }
    if (rtk_lane() == 0) { rtk_atomic_add(bv.counters + 60, sc->my.hb_pass); rtk_atomic_add(bv.counters + 61, sc->my.hb_split); rtk_atomic_add(bv.counters + 62, sc->my.hb_leaf); rtk_atomic_add(bv.counters + 63, sc->my.hb_total); rtk_atomic_add(bv.counters + 59, sc->cnt[9]); }
    rtk_coop_st(&st->exit_flag, 1); // the helpers leave
}

void rtk_launch_phase_long(int grid, int waves, rtk_stream_t s, const LaunchCtx* L, const GraphView& g, const OptsView& o, const BatchView& bv, const PhaseView& pv, char* scratch, uint64_t stride,
                           const RegionScratchCfg& cfg, const uint32_t* list, uint32_t n_list, int only_flagged) {
    hipLaunchKernelGGL(k_phase_long, dim3(static_cast<unsigned>(grid)), dim3(static_cast<unsigned>(64 * waves)), 0, s, L, g, o, bv, pv, scratch, stride, cfg, list, n_list, only_flagged);
    rtk_check(hipGetLastError(), "kernel launch (k_phase_long)");
}
