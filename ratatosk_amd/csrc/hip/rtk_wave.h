// Wavefront execution shim.
//
// Execution model of every kernel in this directory: ONE 64-lane wavefront (= one workgroup) owns one work item
// (a k-mer window tile, a long read, a weak region). Control flow is wave-uniform ("scalar program"); lanes only
// diverge inside the bulk primitives (copies, 2-bit decode, k-mer probes, bit-parallel Myers, sorted-set algebra).
// Uniform values are stored by ALL lanes (same address, same value) so that each lane later observes its own store
// in program order; data produced lane-parallel is published by rtk_sync() at the end of the primitive.
//
// RTK_SIM builds the same scalar programs for the host with a 1-lane "wave" (tests/hostsim): a developer
// simulator used by the CPU-only test tier to exercise the device logic without a GPU. It is a separate shared
// object, is never linked into libratatosk_hip.so and is not a fallback: the product fails with RTK_ERR_NO_DEVICE
// when no GPU is present.
#ifndef RTK_WAVE_H
#define RTK_WAVE_H

#include <stdint.h>
#include <string.h>

#include "rtk_types.h"

#ifdef RTK_SIM

#define RTK_DEV inline
#define RTK_FN inline
#define RTK_FN_SEARCH inline
#define RTK_FN_DRIVER inline
#define RTK_FN_REGION inline
#define RTK_FN_LEAF inline
#define RTK_FN_HOT inline
#define RTK_WAVE 1
inline int rtk_lane() { return 0; }
inline uint64_t rtk_ballot(bool p) { return p ? 1ull : 0ull; }
template <class T> inline T rtk_shfl(T v, int) { return v; }
template <class T> inline T rtk_shfl_up1(T v, T lane0_value) { (void)v; return lane0_value; }
inline void rtk_sync() {}
template <class T> inline T* rtk_opaque(T* p) { return p; }
#define RTK_ASSUME_LDS(p) ((void)0)
inline int rtk_popc(uint64_t x) { return __builtin_popcountll(x); }
inline int rtk_ffs(uint64_t x) { return __builtin_ffsll(static_cast<long long>(x)); } // 1-based, 0 if none
template <class T> inline T rtk_atomic_add_raw(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline uint32_t rtk_atomic_or(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
#ifdef RTK_LANE_PROF
#include <x86intrin.h>
inline unsigned long long rtk_clock() { return __rdtsc(); } // developer build of the simulator: the lap profile of the lane program in host cycles
#else
inline unsigned long long rtk_clock() { return 0; }
#endif
inline uint64_t rtk_brev64(uint64_t x) { uint64_t r = 0; for (int i = 0; i < 64; ++i) { r = (r << 1) | (x & 1ull); x >>= 1; } return r; }

#else

#include <hip/hip_runtime.h>

#define RTK_DEV __device__ __forceinline__
#define RTK_FN __device__ __noinline__ // large device functions are real calls: keeps hipcc compile time and code size bounded
// The path search (extractSemiWeakPaths -> explorePathsBFS -> exploreSubGraph) is compiled into its caller: each of these programs
// has ONE call site, and a call between two of them costs tens of 64-lane stack stores and reloads (register saves, by-reference
// arguments) per region / hop / BFS step: k_regions 40.8 -> 38.5 ms per 64 Mb. -DRTK_SEARCH_CALLS restores the calls (A/B).
#ifdef RTK_SEARCH_CALLS
#define RTK_FN_SEARCH RTK_FN
#else
#define RTK_FN_SEARCH __device__ __forceinline__
#endif
#ifdef RTK_INLINE_LEAVES
#define RTK_FN_LEAF __device__ __forceinline__
#else
#define RTK_FN_LEAF RTK_FN
#endif
#ifdef RTK_INLINE_REGION
#define RTK_FN_REGION __device__ __forceinline__
#else
#define RTK_FN_REGION RTK_FN
#endif
#ifdef RTK_INLINE_DRIVER
#define RTK_FN_DRIVER __device__ __forceinline__
#else
#define RTK_FN_DRIVER RTK_FN
#endif
// Thin wrappers and small leaves on the hot path (alignment entry, path scoring, path extension, colour memo, lane copies / fills).
// A real call costs the callee-saved spills of the AMDGPU calling convention (one private-memory store and load of 64 lanes per
// saved register, on every call): these are inlined; -DRTK_HOT_CALLS turns them back into calls for A/B measurements.
#ifdef RTK_HOT_CALLS
#define RTK_FN_HOT RTK_FN
#else
#define RTK_FN_HOT RTK_DEV
#endif
#define RTK_WAVE 64
__device__ __forceinline__ int rtk_lane() { return static_cast<int>(threadIdx.x) & 63; }
__device__ __forceinline__ uint64_t rtk_ballot(bool p) { return __ballot(p ? 1 : 0); }
template <class T> __device__ __forceinline__ T rtk_shfl(T v, int src) { return __shfl(v, src, 64); }
// value of lane-1 (lane 0 receives lane0_value)
template <class T> __device__ __forceinline__ T rtk_shfl_up1(T v, T lane0_value) { const T r = __shfl_up(v, 1, 64); return rtk_lane() == 0 ? lane0_value : r; }
// publishes lane-parallel stores to the other lanes of the (single-wave) workgroup. In the translation unit of the multi-wave kernels
// (RTK_MULTIWAVE: one program wave + helper waves per workgroup, rtk_phase_long.hip) a program runs on ONE wave of a bigger
// workgroup, so the workgroup barrier becomes a fence + wave barrier: a real s_barrier would wait for waves that never come.
#ifdef RTK_MULTIWAVE
#define RTK_WG_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define RTK_WG_SYNC() __syncthreads()
#endif
__device__ __forceinline__ void rtk_sync() { RTK_WG_SYNC(); }
// a wave-uniform pointer the optimiser knows nothing about (address space, constant value): for pointers into LDS that travel through
// generic-pointer code -- the backend folds the null test of the cast back to LDS into an instruction it cannot encode
// the object behind a generic pointer is in LDS: lets the compiler turn the flat accesses through it into ds_ instructions
#define RTK_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)(p)))
template <class T> __device__ __forceinline__ T* rtk_opaque(T* p) { unsigned long long v = reinterpret_cast<unsigned long long>(p); asm volatile("" : "+s"(v)); return reinterpret_cast<T*>(v); }
__device__ __forceinline__ int rtk_popc(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int rtk_ffs(uint64_t x) { return __ffsll(static_cast<unsigned long long>(x)); }
template <class T> __device__ __forceinline__ T rtk_atomic_add_raw(T* p, T v) { return atomicAdd(p, v); }
__device__ __forceinline__ uint32_t rtk_atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
#ifdef RTK_NO_CLOCK // A/B build: what the cycle counters of the wave programs cost (measured in round 4: nothing, 30.93 against 30.90 ms)
__device__ __forceinline__ unsigned long long rtk_clock() { return 0ull; }
#else
__device__ __forceinline__ unsigned long long rtk_clock() { return static_cast<unsigned long long>(clock64()); }
#endif
__device__ __forceinline__ uint64_t rtk_brev64(uint64_t x) { return __builtin_bitreverse64(x); }

#endif

// 64-bit specialisations of shuffles are provided by HIP for (unsigned) long long; int8/bool go through int.

// wave-wide reductions / scans over one value per lane
RTK_DEV int rtk_wave_sum(int v) {
#ifdef RTK_SIM
    return v;
#else
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#endif
}

RTK_DEV int rtk_wave_excl_scan(int v, int* total) { // exclusive prefix sum across lanes
#ifdef RTK_SIM
    *total = v; return 0;
#else
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (rtk_lane() >= o) inc += t; }
    *total = __shfl(inc, 63, 64);
    return inc - v;
#endif
}

template <class T, class V> RTK_DEV T rtk_atomic_add(T* p, V v) { return rtk_atomic_add_raw(p, static_cast<T>(v)); }
template <class T, class V> RTK_DEV T rtk_atomic_add(const U<T*>& p, V v) { return rtk_atomic_add_raw(p.get(), static_cast<T>(v)); }

// uniform load: *p for a p that is the same in every lane
template <class T> RTK_DEV T rtk_ld(const T* p) { return rtk_u(*p); }
template <class T> RTK_DEV T rtk_ld(const U<T>* p) { return p->get(); }
template <class T> RTK_DEV T rtk_ld(const UL<T>* p) { return p->get(); }

// bulk copy / fill (lane-strided); publishes. 16 bytes per lane and step when both sides are 16-byte aligned.
struct alignas(16) RtkV16 { uint64_t a, b; };
RTK_DEV void rtk_copy_lanes(void* dst, const void* src, uint64_t n) {
    char* d = static_cast<char*>(dst); const char* s = static_cast<const char*>(src);
    uint64_t done = 0;
    if (((reinterpret_cast<uint64_t>(d) | reinterpret_cast<uint64_t>(s)) & 15ull) == 0) {
        const uint64_t n16 = n >> 4;
        for (uint64_t i = static_cast<uint64_t>(rtk_lane()); i < n16; i += RTK_WAVE) reinterpret_cast<RtkV16*>(d)[i] = reinterpret_cast<const RtkV16*>(s)[i];
        done = n16 << 4;
    }
    for (uint64_t i = done + static_cast<uint64_t>(rtk_lane()); i < n; i += RTK_WAVE) d[i] = s[i];
}
RTK_FN_HOT void rtk_wcopy(void* dst, const void* src, uint64_t n) {
    rtk_copy_lanes(rtk_u(dst), rtk_u(src), rtk_u(n));
    rtk_sync();
}
// two copies, one publish
RTK_FN_HOT void rtk_wcopy2(void* dst0, const void* src0, uint64_t n0, void* dst1, const void* src1, uint64_t n1) {
    rtk_copy_lanes(rtk_u(dst0), rtk_u(src0), rtk_u(n0));
    rtk_copy_lanes(rtk_u(dst1), rtk_u(src1), rtk_u(n1));
    rtk_sync();
}

RTK_FN_HOT void rtk_wfill(void* dst, int c, uint64_t n) {
    char* d = static_cast<char*>(rtk_u(dst)); n = rtk_u(n); c = rtk_u(c);
    for (uint64_t i = static_cast<uint64_t>(rtk_lane()); i < n; i += RTK_WAVE) d[i] = static_cast<char>(c);
    rtk_sync();
}

// Read that owns base b0 of the concatenated read buffer: the largest r with roff[r] <= b0 (roff[0] = 0). Wave-uniform. 64 pivots per round, one
// per lane (three dependent loads for a ticket of tens of thousands of reads instead of the 13-16 of a binary search: the tile programs of the
// k-mer scans start with it).
RTK_DEV uint32_t rtk_owner_read(const uint64_t* roff_, uint32_t n_reads_, uint64_t b0_) {
    const uint64_t* roff = rtk_u(roff_); const uint32_t n_reads = rtk_u(n_reads_); const uint64_t b0 = rtk_u(b0_);
    uint32_t lo = 0, hi = n_reads;
#ifdef RTK_SIM
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (roff[mid] <= b0) lo = mid; else hi = mid; }
#else
    while (hi - lo > 1) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint64_t p = static_cast<uint64_t>(lo) + static_cast<uint64_t>(rtk_lane() + 1) * step; // pivots lo + step, lo + 2 step, ...: "roff[p] <= b0" holds for a prefix of the lanes
        const bool le = p < hi && roff[p] <= b0;
        const uint32_t c = static_cast<uint32_t>(rtk_popc(rtk_ballot(le)));
        const uint64_t nh = static_cast<uint64_t>(lo) + static_cast<uint64_t>(c + 1) * step;
        lo = rtk_u(lo + c * step); hi = rtk_u(nh < hi ? static_cast<uint32_t>(nh) : hi);
    }
#endif
    return lo;
}

#endif
