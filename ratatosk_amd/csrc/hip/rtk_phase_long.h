// Launcher of the multi-wave phasing kernel (rtk_phase_long.hip), declared for rtk_device.hip. Not part of the simulator build.
#ifndef RTK_PHASE_LONG_H
#define RTK_PHASE_LONG_H
#include "rtk_phasing.h"
// reads list[0..n_list) (long reads: their whole-read alignment spans several 4096-row blocks), one workgroup of `waves` waves per read at a
// time, `grid` workgroups, each with its own work area scratch + i * stride. Status as k_phase: bv.status[r] = overflow code.
void rtk_launch_phase_long(int grid, int waves, rtk_stream_t st, const LaunchCtx* L, const GraphView& g, const OptsView& o, const BatchView& bv, const PhaseView& pv, char* scratch, uint64_t stride,
                           const RegionScratchCfg& cfg, const uint32_t* list, uint32_t n_list, int only_flagged);
// bytes of one workgroup's work area: the read program's + one leaf-traceback area per helper wave
inline uint64_t rtk_phase_long_stride(const RegionScratchCfg& cfg, int waves) { return region_scratch_bytes(cfg) + static_cast<uint64_t>(waves > 1 ? (waves < RTK_LEAF_WAVES ? waves : RTK_LEAF_WAVES) - 1 : 0) * scratch_bytes(rtk_leaf_cfg()); }
#endif
