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
    __shared__ RegionScratch hdr;
    __shared__ MyersScratch lsc[16]; // work areas of the waves for the leaf tracebacks of an alignment: [0] = the program wave's own, the helpers' behind the read program's area
    if (threadIdx.x == 0) { st->seq = 0; st->n_done = 0; st->exit_flag = 0; st->next_item = 0; st->leaf_mode = 0; st->n_gangs = 0; st->lsc = lsc; st->n_leaf_waves = (static_cast<int>(blockDim.x) >> 6) < RTK_LEAF_WAVES ? (static_cast<int>(blockDim.x) >> 6) : RTK_LEAF_WAVES; st->n_waves = static_cast<int>(blockDim.x) >> 6; }
    char* const wg_base = scratch + static_cast<uint64_t>(blockIdx.x) * stride;
    if (wave != 0 && wave < RTK_LEAF_WAVES) lsc[wave] = scratch_carve(wg_base + region_scratch_bytes(cfg) + static_cast<uint64_t>(wave - 1) * scratch_bytes(rtk_leaf_cfg()), rtk_leaf_cfg());
    __syncthreads(); // the only workgroup barrier of the kernel: every wave is here
    if (wave != 0) { rtk_myers_coop_helper(wave); return; }
    RegionScratch* sc = region_scratch_carve(wg_base, cfg, &hdr);
    rtk_sync();
    lsc[0] = hdr.my;
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
    // (stride: region_scratch_bytes(cfg) + (waves - 1) * scratch_bytes(rtk_leaf_cfg()), see rtk_phase_long_stride)
    hipLaunchKernelGGL(k_phase_long, dim3(static_cast<unsigned>(grid)), dim3(static_cast<unsigned>(64 * waves)), 0, s, L, g, o, bv, pv, scratch, stride, cfg, list, n_list, only_flagged);
    rtk_check(hipGetLastError(), "kernel launch (k_phase_long)");
}

// ---- stage entry rtk_myers_batch_waves: the problems of rtk_myers_batch on workgroups of several waves (same program on wave 0) ----
RTK_FN void rtk_myers_batch_item(const MyersScratch& sc, const MyersProb& p, uint32_t i, const char* pool, int want_path, int use_iupac, int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs,
                                 uint8_t* moves_out, uint32_t* n_moves_out, uint32_t cap_moves, uint32_t* status) {
    *sc.overflow = 0;
    const char* q = pool + p.q_off; const char* t = pool + p.t_off;
    const MyersResult r = rtk_myers_distance(sc, q, static_cast<int>(p.qlen), t, static_cast<int>(p.tlen), p.k, p.mode, use_iupac != 0, cap_locs ? end_locs + static_cast<uint64_t>(i) * cap_locs : nullptr, static_cast<int>(cap_locs));
    dist[i] = r.dist; n_loc[i] = r.nloc;
    uint32_t nm = 0;
    if (want_path && r.dist >= 0 && p.qlen > 0 && p.tlen > 0) {
        if (p.k < 0 && p.mode != RTK_MODE_HW) {
            const MyersResult r2 = rtk_myers_path(sc, q, static_cast<int>(p.qlen), t, static_cast<int>(p.tlen), p.mode, use_iupac != 0, &nm);
            if (r2.dist != r.dist || r2.first != r.first) *sc.overflow = 3;
        } else
        rtk_myers_alignment(sc, q, static_cast<int>(p.qlen), t, r.first + 1, r.dist, use_iupac != 0, &nm);
        if (nm <= cap_moves) rtk_wcopy(moves_out + static_cast<uint64_t>(i) * cap_moves, sc.moves, nm);
    }
    n_moves_out[i] = nm;
    status[i] = *sc.overflow;
}
__global__ void __launch_bounds__(1024) k_myers_batch_waves(const MyersProb* probs, uint32_t n, const char* pool, int want_path, int use_iupac, char* scratch, uint64_t scratch_stride, ScratchCfg cfg,
                                                            int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, uint8_t* moves_out, uint32_t* n_moves_out, uint32_t cap_moves, uint32_t* status, unsigned long long* prof) {
    const int wave = static_cast<int>(threadIdx.x) >> 6;
    RtkCoop* st = rtk_coop();
    __shared__ MyersScratch sc;
    __shared__ MyersScratch lsc[16];
    if (threadIdx.x == 0) { st->seq = 0; st->n_done = 0; st->exit_flag = 0; st->next_item = 0; st->leaf_mode = 0; st->n_gangs = 0; st->lsc = lsc; st->n_leaf_waves = (static_cast<int>(blockDim.x) >> 6) < RTK_LEAF_WAVES ? (static_cast<int>(blockDim.x) >> 6) : RTK_LEAF_WAVES; st->n_waves = static_cast<int>(blockDim.x) >> 6; }
    char* const wg_base = scratch + static_cast<uint64_t>(blockIdx.x) * scratch_stride;
    if (wave != 0 && wave < RTK_LEAF_WAVES) lsc[wave] = scratch_carve(wg_base + scratch_bytes(cfg) + static_cast<uint64_t>(wave - 1) * scratch_bytes(rtk_leaf_cfg()), rtk_leaf_cfg());
    __syncthreads(); // the only workgroup barrier of the kernel
    if (wave != 0) { rtk_myers_coop_helper(wave); return; }
    sc = scratch_carve(wg_base, cfg); lsc[0] = sc;
    rtk_sync();
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x)
        rtk_myers_batch_item(sc, probs[i], i, pool, want_path, use_iupac, dist, n_loc, end_locs, cap_locs, moves_out, n_moves_out, cap_moves, status);
    if (prof && rtk_lane() == 0) { rtk_atomic_add(prof + 0, sc.hb_total.get()); rtk_atomic_add(prof + 1, sc.hb_pass.get()); rtk_atomic_add(prof + 2, sc.hb_split.get()); rtk_atomic_add(prof + 3, sc.hb_leaf.get()); rtk_atomic_add(prof + 4, sc.walk_cycles.get()); }
    rtk_coop_st(&st->exit_flag, 1);
}
void rtk_launch_myers_batch_waves(int grid, int waves, const MyersProb* probs, uint32_t n, const char* pool, int want_path, int use_iupac, char* scratch, uint64_t scratch_stride, const ScratchCfg& cfg,
                                  int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, uint8_t* moves_out, uint32_t* n_moves_out, uint32_t cap_moves, uint32_t* status, unsigned long long* prof) {
    hipLaunchKernelGGL(k_myers_batch_waves, dim3(static_cast<unsigned>(grid)), dim3(static_cast<unsigned>(64 * waves)), 0, 0, probs, n, pool, want_path, use_iupac, scratch, scratch_stride, cfg,
                       dist, n_loc, end_locs, cap_locs, moves_out, n_moves_out, cap_moves, status, prof);
    rtk_check(hipGetLastError(), "kernel launch (k_myers_batch_waves)");
}
