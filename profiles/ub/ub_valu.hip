// Micro-benchmarks for the Myers step on gfx950: dependent-issue latency of the instruction kinds the step is made of, and
// candidate step loops. One wave; cycles from s_memtime (constant 100 MHz on CDNA: we use clock64 = s_memrealtime?) -> use wall_clock64 and
// __builtin_readcyclecounter both.  Build: hipcc --offload-arch=gfx950 -O3 -o ub_valu ub_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define N_IT 4096
__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }

__global__ void k_chain_add(unsigned* out, unsigned long long* t, unsigned a) {
    unsigned x = threadIdx.x + a;
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a)); }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_chain_indep2(unsigned* out, unsigned long long* t, unsigned a) { // two independent chains interleaved
    unsigned x = threadIdx.x + a, y = threadIdx.x * 3 + a;
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(a)); }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x + y; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_chain_dpp(unsigned* out, unsigned long long* t, unsigned a) {
    int x = threadIdx.x + a;
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, false); }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_chain_dpp_add(unsigned* out, unsigned long long* t, unsigned a) { // dpp then a VALU op on it
    int x = threadIdx.x + a;
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, false); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a)); }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_chain_readlane(unsigned* out, unsigned long long* t, unsigned a) { // v -> readlane -> s -> v_add
    int x = threadIdx.x + a;
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { int s = __builtin_amdgcn_readlane(x, 5); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "s"(s)); }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_branch(unsigned* out, unsigned long long* t, unsigned a) { // a taken scalar branch per iteration + one valu
    int x = threadIdx.x + a; int s = a;
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT * 16; ++i) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a)); }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x + s; if (threadIdx.x == 0) t[0] = t1 - t0;
}

// ---- the current 32-bit step (steady phase) as the compiler builds it: TRACK = 0 / 1
__device__ __forceinline__ int step32(uint32_t& Pv, uint32_t& Mv, uint32_t Eq, int hin, int bit, uint32_t& Ph_out, uint32_t& Mh_out) {
    const uint32_t pv = Pv, mv = Mv; const uint32_t Xv = Eq | mv; Eq |= static_cast<uint32_t>(hin) >> 31;
    const uint32_t Xh = (((Eq & pv) + pv) ^ pv) | Eq; uint32_t Ph = mv | ~(Xh | pv); uint32_t Mh = pv & Xh; Ph_out = Ph; Mh_out = Mh;
    const int hout = static_cast<int>((Ph >> bit) & 1u) - static_cast<int>((Mh >> bit) & 1u);
    Ph = (Ph << 1) | (hin > 0 ? 1u : 0u); Mh = (Mh << 1) | (static_cast<uint32_t>(hin) >> 31);
    Pv = Mh | ~(Xv | Ph); Mv = Ph & Xv; return hout;
}
template <int TRACK>
__global__ void k_step_cur(unsigned* out, unsigned long long* t, const unsigned char* tp, int n, int W, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3) {
    const int lane = threadIdx.x; const int bit = (lane == W - 1) ? 17 : 31;
    uint32_t eqA = e0 * (lane + 1), eqC = e1 * (lane + 3), eqG = e2 * (lane + 5), eqT = e3 * (lane + 7);
    uint32_t Pv = ~0u, Mv = 0; int hout_prev = 0; uint32_t m1_prev = 0, m2_prev = 0; int score = 32 * W; int best = 0x7fffffff, first = -1, last = -1, cnt = 0;
    unsigned long long t0 = clk();
    for (int c0 = 0; c0 < n; c0 += 64) {
        int my_t = tp[c0 + lane]; asm volatile("" : "+v"(my_t));
        for (int j = 0; j < 64; ++j) {
            const int s = c0 + j;
            const int in_t = __builtin_amdgcn_readlane(my_t, j);
            const int hin = __builtin_amdgcn_update_dpp(1, hout_prev, 0x138, 0xF, 0xF, false);
            const uint32_t m1 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 1, 1), static_cast<int>(m1_prev), 0x138, 0xF, 0xF, false));
            const uint32_t m2 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 2, 1), static_cast<int>(m2_prev), 0x138, 0xF, 0xF, false));
            const uint32_t lo_ = (eqC & m1) | (eqA & ~m1), hi_ = (eqG & m1) | (eqT & ~m1);
            const uint32_t Eq = (hi_ & m2) | (lo_ & ~m2);
            uint32_t nPv = Pv, nMv = Mv, Ph, Mh;
            const int hout = step32(nPv, nMv, Eq, hin, bit, Ph, Mh);
            Pv = nPv; Mv = nMv; hout_prev = hout; score += hout; m1_prev = m1; m2_prev = m2;
            if (TRACK) { const int tcol = s - (W - 1); const int sv = __builtin_amdgcn_readlane(score, W - 1);
                if (sv < best) { best = sv; first = tcol; last = tcol; cnt = 1; } else if (sv == best) { last = tcol; ++cnt; } }
        }
    }
    unsigned long long t1 = clk();
    out[lane] = Pv ^ Mv ^ score ^ best ^ first ^ last ^ cnt; if (lane == 0) t[0] = t1 - t0;
}

// ---- candidate: speculative step. Both outcomes of (hin < 0) are prepared before hin arrives; the lane-to-lane chain is dpp -> cmp -> select.
// hin travels as the two bits (neg, pos) packed with the two character-select bits in one DPP word.
template <int TRACK>
__global__ void k_step_spec(unsigned* out, unsigned long long* t, const unsigned char* tp, int n, int W, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3) {
    const int lane = threadIdx.x; const int bit = (lane == W - 1) ? 17 : 31;
    uint32_t eqA = e0 * (lane + 1), eqC = e1 * (lane + 3), eqG = e2 * (lane + 5), eqT = e3 * (lane + 7);
    uint32_t Pv = ~0u, Mv = 0; int hout_prev = 0; uint32_t m1_prev = 0, m2_prev = 0; int score = 32 * W;
    int best = 0x7fffffff, first = -1, last = -1, cnt = 0; // lane-local tracking (only lane W-1's values are read at the end)
    unsigned long long t0 = clk();
    for (int c0 = 0; c0 < n; c0 += 64) {
        int my_t = tp[c0 + lane]; asm volatile("" : "+v"(my_t));
#pragma unroll 4
        for (int j = 0; j < 64; ++j) {
            const int s = c0 + j;
            const int in_t = __builtin_amdgcn_readlane(my_t, j);
            // character masks first (they do not depend on the lane above's arithmetic, only on its forwarding)
            const uint32_t m1 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 1, 1), static_cast<int>(m1_prev), 0x138, 0xF, 0xF, false));
            const uint32_t m2 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 2, 1), static_cast<int>(m2_prev), 0x138, 0xF, 0xF, false));
            const uint32_t lo_ = (eqC & m1) | (eqA & ~m1), hi_ = (eqG & m1) | (eqT & ~m1);
            const uint32_t Eq = (hi_ & m2) | (lo_ & ~m2);
            const uint32_t pv = Pv, mv = Mv;
            const uint32_t Xv = Eq | mv;
            const uint32_t Eq1 = Eq | 1u;
            const uint32_t s0 = (Eq & pv) + pv, s1 = (Eq1 & pv) + pv;
            const uint32_t Xh0 = (s0 ^ pv) | Eq, Xh1 = (s1 ^ pv) | Eq1;
            const uint32_t Ph0 = mv | ~(Xh0 | pv), Ph1 = mv | ~(Xh1 | pv);
            const uint32_t Mh0 = pv & Xh0, Mh1 = pv & Xh1;
            const int ho0 = static_cast<int>((Ph0 >> bit) & 1u) - static_cast<int>((Mh0 >> bit) & 1u);
            const int ho1 = static_cast<int>((Ph1 >> bit) & 1u) - static_cast<int>((Mh1 >> bit) & 1u);
            const int hin = __builtin_amdgcn_update_dpp(1, hout_prev, 0x138, 0xF, 0xF, false);
            const bool neg = hin < 0;
            const int hout = neg ? ho1 : ho0;
            hout_prev = hout;
            uint32_t Ph = neg ? Ph1 : Ph0, Mh = neg ? Mh1 : Mh0;
            Ph = (Ph << 1) | (hin > 0 ? 1u : 0u); Mh = (Mh << 1) | (neg ? 1u : 0u);
            Pv = Mh | ~(Xv | Ph); Mv = Ph & Xv;
            score += hout; m1_prev = m1; m2_prev = m2;
            if (TRACK) { const int tcol = s - (W - 1); const bool lt = score < best, eq = score == best;
                best = lt ? score : best; first = lt ? tcol : first; last = (lt || eq) ? tcol : last; cnt = lt ? 1 : (cnt + (eq ? 1 : 0)); }
        }
    }
    unsigned long long t1 = clk();
    out[lane] = Pv ^ Mv ^ score ^ best ^ first ^ last ^ cnt; if (lane == 0) t[0] = t1 - t0;
}

// ---- candidate: the current arithmetic, lane-local tracking (no readlane / scalar branches), unrolled
template <int TRACK>
__global__ void k_step_vtrack(unsigned* out, unsigned long long* t, const unsigned char* tp, int n, int W, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3) {
    const int lane = threadIdx.x; const int bit = (lane == W - 1) ? 17 : 31;
    uint32_t eqA = e0 * (lane + 1), eqC = e1 * (lane + 3), eqG = e2 * (lane + 5), eqT = e3 * (lane + 7);
    uint32_t Pv = ~0u, Mv = 0; int hout_prev = 0; uint32_t m1_prev = 0, m2_prev = 0; int score = 32 * W; int best = 0x7fffffff, first = -1, last = -1, cnt = 0;
    unsigned long long t0 = clk();
    for (int c0 = 0; c0 < n; c0 += 64) {
        int my_t = tp[c0 + lane]; asm volatile("" : "+v"(my_t));
#pragma unroll 4
        for (int j = 0; j < 64; ++j) {
            const int s = c0 + j;
            const int in_t = __builtin_amdgcn_readlane(my_t, j);
            const int hin = __builtin_amdgcn_update_dpp(1, hout_prev, 0x138, 0xF, 0xF, false);
            const uint32_t m1 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 1, 1), static_cast<int>(m1_prev), 0x138, 0xF, 0xF, false));
            const uint32_t m2 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 2, 1), static_cast<int>(m2_prev), 0x138, 0xF, 0xF, false));
            const uint32_t lo_ = (eqC & m1) | (eqA & ~m1), hi_ = (eqG & m1) | (eqT & ~m1);
            const uint32_t Eq = (hi_ & m2) | (lo_ & ~m2);
            uint32_t nPv = Pv, nMv = Mv, Ph, Mh;
            const int hout = step32(nPv, nMv, Eq, hin, bit, Ph, Mh);
            Pv = nPv; Mv = nMv; hout_prev = hout; score += hout; m1_prev = m1; m2_prev = m2;
            if (TRACK) { const int tcol = s - (W - 1); const bool lt = score < best, eq = score == best;
                best = lt ? score : best; first = lt ? tcol : first; last = (lt || eq) ? tcol : last; cnt = lt ? 1 : (cnt + (eq ? 1 : 0)); }
        }
    }
    unsigned long long t1 = clk();
    out[lane] = Pv ^ Mv ^ score ^ best ^ first ^ last ^ cnt; if (lane == 0) t[0] = t1 - t0;
}

int main() {
    unsigned* out; unsigned long long* t; unsigned char* tp; const int n = 64 * 256;
    hipMalloc(&out, 256); hipMalloc(&t, 64); hipMalloc(&tp, n + 64);
    std::vector<unsigned char> h(n + 64); for (int i = 0; i < n + 64; ++i) h[i] = "ACGT"[(i * 2654435761u >> 13) & 3];
    hipMemcpy(tp, h.data(), n + 64, hipMemcpyHostToDevice);
    unsigned long long ht = 0; unsigned hout[64];
    auto rep = [&](const char* name, double per) { hipDeviceSynchronize(); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); hipMemcpy(hout, out, 256, hipMemcpyDeviceToHost); printf("%-28s %8.2f cycles per unit (check %u)\n", name, double(ht) / per, hout[3]); };
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k_chain_add, 1, 64, 0, 0, out, t, 3u); rep("dependent v_add_u32", N_IT * 16.0);
        hipLaunchKernelGGL(k_chain_indep2, 1, 64, 0, 0, out, t, 3u); rep("2 independent adds (pair)", N_IT * 8.0);
        hipLaunchKernelGGL(k_chain_dpp, 1, 64, 0, 0, out, t, 3u); rep("dependent dpp mov", N_IT * 16.0);
        hipLaunchKernelGGL(k_chain_dpp_add, 1, 64, 0, 0, out, t, 3u); rep("dpp + add (pair)", N_IT * 8.0);
        hipLaunchKernelGGL(k_chain_readlane, 1, 64, 0, 0, out, t, 3u); rep("readlane + add (pair)", N_IT * 8.0);
        hipLaunchKernelGGL(k_branch, 1, 64, 0, 0, out, t, 3u); rep("add + loop branch", N_IT * 16.0);
        for (int W : {4, 7, 16}) {
            printf("W = %d\n", W);
            hipLaunchKernelGGL(k_step_cur<0>, 1, 64, 0, 0, out, t, tp, n, W, 0x12345678u, 0x9abcdef1u, 0x0f1e2d3cu, 0x4b5a6978u); rep("step current NW", double(n));
            hipLaunchKernelGGL(k_step_cur<1>, 1, 64, 0, 0, out, t, tp, n, W, 0x12345678u, 0x9abcdef1u, 0x0f1e2d3cu, 0x4b5a6978u); rep("step current TRACK", double(n));
            hipLaunchKernelGGL(k_step_vtrack<0>, 1, 64, 0, 0, out, t, tp, n, W, 0x12345678u, 0x9abcdef1u, 0x0f1e2d3cu, 0x4b5a6978u); rep("step unrolled NW", double(n));
            hipLaunchKernelGGL(k_step_vtrack<1>, 1, 64, 0, 0, out, t, tp, n, W, 0x12345678u, 0x9abcdef1u, 0x0f1e2d3cu, 0x4b5a6978u); rep("step unrolled vtrack", double(n));
            hipLaunchKernelGGL(k_step_spec<0>, 1, 64, 0, 0, out, t, tp, n, W, 0x12345678u, 0x9abcdef1u, 0x0f1e2d3cu, 0x4b5a6978u); rep("step speculative NW", double(n));
            hipLaunchKernelGGL(k_step_spec<1>, 1, 64, 0, 0, out, t, tp, n, W, 0x12345678u, 0x9abcdef1u, 0x0f1e2d3cu, 0x4b5a6978u); rep("step speculative vtrack", double(n));
        }
    }
    // clock: cycles of __builtin_readcyclecounter per second
    { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a); hipLaunchKernelGGL(k_chain_add, 1, 64, 0, 0, out, t, 3u); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); printf("counter: %.1f MHz (kernel %.3f ms incl. launch)\n", double(ht) / (ms * 1e3), ms); }
    return 0;
}
