// The lookup structures of the flat graph, built in HBM from the packed unitigs (the loader behind src/Ratatosk.cpp:1087-1089
// dbg.read; what Bifrost's own k-mer index and neighbour iterators are to the reference). Same content as host/flat_graph.cpp builds on the host
// threads -- the host build stays the definition and the CPU-tier / simulator path; tests/test_graph_load.py holds the two to each other --
// but a 3 Gb graph takes seconds here instead of minutes, and the two largest arrays never cross PCIe:
//   k-mer table + filters  one thread per pool word (32 bases): the k-mers that END in its word, rolled from the k - 1 bases in front; slots claimed with a
//                          compare-and-swap on the key word; a k-mer met twice raises the error flag (two-word k-mers: every k-mer has to find ITSELF afterwards).
//                          HBM-bound on random 16-byte slots: ~4 transactions per k-mer.
//   half-k-mer index       (h-mer, start position) of every h-mer start as ONE 64-bit key (h-mer << 34 | position in the pool: unitigs lie in the pool in id
//                          order, so that is the host's order by (h-mer, unitig, offset)), radix-sorted (rocPRIM); the list of an h-mer starts at
//                          (index of its first key) + (rank of the h-mer among the distinct ones) -- the ranks come from a bitmap over all 4^h h-mers
//                          (128 MB for k = 31) and its popcount prefix, known before the sort: the lists come out byte for byte as the host writes them.
//                          Graphs whose keys do not fit twice are sorted in ranges of leading h-mer bits (a histogram of the first pass sizes them).
//   adjacency              one thread per (unitig, strand, base): rtk_find_km of the neighbour k-mer, kept if it opens its unitig in walk direction ([A3]).
// Own translation unit: rocPRIM's templates.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../host/flat_graph.hpp"
#include "rtk_graph_tables.h"
#include "rtk_mem.h"
#include "rtk_types.h"

namespace {

struct Dev { // a device allocation released on scope exit unless taken
    void* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    void alloc(uint64_t bytes) { if (p) (void)hipFree(p); p = nullptr; rtk_check(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc (graph tables)"); }
    void* take() { void* q = p; p = nullptr; return q; }
    template <class T> T* as() { return static_cast<T*>(p); }
};

#define RTK_TB_BLOCK 256
#define RTK_POS_BITS 34
#define RTK_POS_MASK ((1ull << RTK_POS_BITS) - 1ull)

// Thread of pool word w: fn(code, unitig, offset of the window in its unitig, its position in the pool) for every window of L bases that lies inside
// one unitig and ENDS at one of the 32 positions of the word. The code is rolled base by base from the (at most L - 1) bases in front of the word.
template <class F>
__device__ __forceinline__ void windows_ending_in_word(const uint64_t* __restrict__ useq, const uint64_t* __restrict__ uoff, uint32_t n_unitigs, uint64_t n_bases, uint64_t w, int L, F fn) {
    const uint64_t p0 = w * 32ull;
    if (p0 >= n_bases) return;
    const uint64_t p1 = p0 + 32ull < n_bases ? p0 + 32ull : n_bases;
    uint32_t lo = 0, hi = n_unitigs; // uoff[lo] <= p0 < uoff[hi]
    while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (uoff[mid] <= p0) lo = mid; else hi = mid; }
    uint32_t u = lo; uint64_t ub = uoff[u], ue = uoff[u + 1];
    uint64_t p = p0 >= static_cast<uint64_t>(L - 1) ? p0 - static_cast<uint64_t>(L - 1) : 0ull; if (p < ub) p = ub;
    uint32_t run = 0; RtkKm fw = rtk_km_zero();
    uint64_t cur = useq[p >> 5];
    for (; p < p1; ++p) {
        if ((p & 31ull) == 0ull) cur = useq[p >> 5];
        if (p == ue) { ++u; ub = ue; ue = uoff[u + 1]; run = 0; }
        fw = rtk_km_push(fw, (cur >> (2 * (p & 31ull))) & 3ull, L); ++run;
        if (run >= static_cast<uint32_t>(L) && p >= p0) fn(fw, u, static_cast<uint32_t>(p + 1 - L - ub), p + 1 - L);
    }
}

__global__ void k_fill_slots(uint64_t* __restrict__ ht, uint64_t slots) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < slots; i += stride) { ht[2 * i] = RTK_EMPTY_KEY; ht[2 * i + 1] = 0; }
}
// a place word of the half-k-mer index keeps its in-unitig offset in 31 bits (bit 31 and bit 63 are flags): the same refusal as the host builder's (flat_graph.cpp)
__global__ void k_check_unitig_lengths(const uint64_t* __restrict__ uoff, uint32_t n_unitigs, uint32_t* __restrict__ err) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t u = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; u < n_unitigs; u += stride) if ((uoff[u + 1] - uoff[u]) >> 31) atomicOr(err, 2u);
}
__global__ void k_fill_words(uint64_t* __restrict__ a, uint64_t n, uint64_t v) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}

// ---- k-mer table + filters (host/flat_graph.cpp "k-mer table + filters") ----
__global__ void k_tables_insert(const uint64_t* __restrict__ useq, const uint64_t* __restrict__ uoff, uint32_t n_unitigs, uint64_t n_bases, int k,
                                uint64_t* __restrict__ ht, uint64_t slots, uint64_t* __restrict__ bf, uint64_t bf_mask, uint64_t* __restrict__ bf1, uint64_t bf1_mask, int bf1_off, int filters, uint32_t* __restrict__ err) {
    const bool wide = k > 31;
    const uint64_t n_words = (n_bases + 31ull) / 32ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        windows_ending_in_word(useq, uoff, n_unitigs, n_bases, w, k, [&](const RtkKm& fw, uint32_t u, uint32_t off, uint64_t) {
            const RtkKm rc = rtk_km_revcomp(fw, k);
            const bool is_fw = !rtk_km_less(rc, fw);
            const RtkKm can = is_fw ? fw : rc;
            const uint64_t hh = wide ? rtk_km_hash(can) : rtk_hash64(can.lo), key = wide ? rtk_km_fingerprint(can) : can.lo;
            uint64_t s = rtk_ht_slot(hh, slots);
            for (;;) {
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(ht + 2 * s), static_cast<unsigned long long>(RTK_EMPTY_KEY), static_cast<unsigned long long>(key));
                if (old == RTK_EMPTY_KEY) break;
                if (!wide && old == key) { atomicOr(err, 1u); return; } // the k-mer is in the table already: not a compacted de Bruijn graph for this k
                s = rtk_ht_next(s, slots);
            }
            ht[2 * s + 1] = (static_cast<uint64_t>(u) << 32) | (static_cast<uint64_t>(off) << 1) | (is_fw ? 1ull : 0ull);
            if (!filters) return; // (the index build's table: looked up without filters)
            atomicOr(reinterpret_cast<unsigned long long*>(bf + ((hh >> 32) & bf_mask)), static_cast<unsigned long long>((1ull << (hh & 63)) | (1ull << ((hh >> 6) & 63))));
            if (!bf1_off) { const uint64_t b1 = (hh >> 12) & bf1_mask; atomicOr(reinterpret_cast<unsigned long long*>(bf1 + (b1 >> 6)), static_cast<unsigned long long>(1ull << (b1 & 63ull))); }
        });
    }
}

// two-word k-mers: the key words are fingerprints, which cannot tell a repeated k-mer while the table is filled -- every k-mer has to find ITSELF
__global__ void k_tables_verify_wide(GraphView g, uint64_t n_bases, uint32_t* __restrict__ err) {
    const uint64_t* useq = g.useq; const uint64_t* uoff = g.uoff; const uint32_t n = g.n_unitigs; const int k = g.k;
    const uint64_t n_words = (n_bases + 31ull) / 32ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        windows_ending_in_word(useq, uoff, n, n_bases, w, k, [&](const RtkKm& fw, uint32_t u, uint32_t off, uint64_t) {
            if (rtk_find_kmer_wide(g, fw, nullptr) != rtk_pack_hit(u, off, 1u)) atomicOr(err, 1u);
        });
    }
}

// ---- adjacency (host/flat_graph.cpp "adjacency": neighbours of the unitig end in walk direction, A,C,G,T) ----
__device__ __forceinline__ RtkKm km_at(const uint64_t* __restrict__ useq, uint64_t pos, int k) {
    RtkKm x = rtk_km_zero();
    for (int i = 0; i < k; ++i) { const uint64_t p = pos + static_cast<uint64_t>(i); x = rtk_km_push(x, (useq[p >> 5] >> (2 * (p & 31ull))) & 3ull, k); }
    return x;
}
__global__ void k_tables_adjacency(GraphView g, uint32_t* __restrict__ adj) {
    const uint64_t* useq = g.useq; const uint64_t* uoff = g.uoff; const uint32_t n = g.n_unitigs; const int k = g.k;
    const uint64_t total = static_cast<uint64_t>(n) * 8ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint32_t u = static_cast<uint32_t>(t >> 3), d = static_cast<uint32_t>((t >> 2) & 1ull); const uint64_t b = t & 3ull;
        const RtkKm end = d == 0 ? km_at(useq, uoff[u + 1] - static_cast<uint64_t>(k), k) : rtk_km_revcomp(km_at(useq, uoff[u], k), k);
        const uint64_t hit = rtk_find_km(g, rtk_km_push(end, b, k), nullptr);
        uint32_t out = RTK_NONE32;
        if (hit != RTK_NO_HIT) {
            const UMap f = rtk_unpack_hit(hit);
            const uint32_t nk = static_cast<uint32_t>(uoff[f.unitig + 1] - uoff[f.unitig]) - static_cast<uint32_t>(k) + 1u;
            if ((f.strand && f.dist == 0) || (!f.strand && f.dist == nk - 1u)) out = (f.unitig << 1) | f.strand; // find(km, extremities_only = true): the k-mer opens its unitig in walk direction
        }
        adj[t] = out;
    }
}

// ---- half-k-mer index (host/flat_graph.cpp "half-k-mer index") ----
// first pass: which h-mers exist (a bit each) and how many starts fall into each range of leading h-mer bits
__global__ void k_hx_mark(const uint64_t* __restrict__ useq, const uint64_t* __restrict__ uoff, uint32_t n_unitigs, uint64_t n_bases, int h, int bshift, uint32_t n_bins,
                          uint64_t* __restrict__ bitmap, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t lh[4096];
    for (uint32_t i = threadIdx.x; i < n_bins; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const uint64_t n_words = (n_bases + 31ull) / 32ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        windows_ending_in_word(useq, uoff, n_unitigs, n_bases, w, h, [&](const RtkKm& fw, uint32_t, uint32_t, uint64_t) {
            const uint64_t rc_ = rtk_revcomp(fw.lo, h); const uint64_t hm = fw.lo < rc_ ? fw.lo : rc_; // the index is keyed by the canonical h-mer
            atomicOr(reinterpret_cast<unsigned long long*>(bitmap + (hm >> 6)), static_cast<unsigned long long>(1ull << (hm & 63ull)));
            atomicAdd(&lh[hm >> bshift], 1u);
        });
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_bins; i += blockDim.x) if (lh[i]) atomicAdd(hist + i, static_cast<unsigned long long>(lh[i]));
}

// the keys of the bins [bin_lo, bin_hi): every thread counts its windows, the wave takes one range of the key array, the threads write their keys
__global__ void k_hx_keys(const uint64_t* __restrict__ useq, const uint64_t* __restrict__ uoff, uint32_t n_unitigs, uint64_t n_bases, int h, int bshift, uint32_t bin_lo, uint32_t bin_hi,
                          uint64_t* __restrict__ keys, unsigned long long* __restrict__ top) {
    const uint64_t n_words = (n_bases + 31ull) / 32ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t rounds = (n_words + stride - 1) / stride;
    const int lane = threadIdx.x & 63;
    for (uint64_t r = 0; r < rounds; ++r) { // (every lane takes part in the scans of every round)
        const uint64_t w = r * stride + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
        uint32_t mine = 0;
        if (w < n_words) windows_ending_in_word(useq, uoff, n_unitigs, n_bases, w, h, [&](const RtkKm& fw, uint32_t, uint32_t, uint64_t) { const uint64_t rc_ = rtk_revcomp(fw.lo, h); const uint32_t b = static_cast<uint32_t>((fw.lo < rc_ ? fw.lo : rc_) >> bshift); if (b >= bin_lo && b < bin_hi) ++mine; });
        uint32_t incl = mine;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        const uint32_t tot = __shfl(incl, 63, 64);
        if (tot == 0) continue;
        unsigned long long base = 0;
        if (lane == 63) base = atomicAdd(top, static_cast<unsigned long long>(tot));
        base = __shfl(base, 63, 64);
        uint64_t at = base + (incl - mine);
        if (mine) windows_ending_in_word(useq, uoff, n_unitigs, n_bases, w, h, [&](const RtkKm& fw, uint32_t, uint32_t, uint64_t pos) { const uint64_t rc_ = rtk_revcomp(fw.lo, h); const uint64_t cn = fw.lo < rc_ ? fw.lo : rc_; const uint32_t b = static_cast<uint32_t>(cn >> bshift); if (b >= bin_lo && b < bin_hi) keys[at++] = (cn << RTK_POS_BITS) | pos; });
    }
}

struct PopcountOf { const uint64_t* w; __device__ uint32_t operator()(uint64_t i) const { return static_cast<uint32_t>(__popcll(w[i])); } };

// sorted keys -> the lists and the slot table. S: the keys of one range of bins, `base` keys lie in the ranges before it.
__global__ void k_hx_scatter(const uint64_t* __restrict__ S, uint64_t n_s, uint64_t base, const uint64_t* __restrict__ bitmap, const uint32_t* __restrict__ rank, const uint64_t* __restrict__ useq, int h,
                             const uint64_t* __restrict__ uoff, uint32_t n_unitigs, uint64_t* __restrict__ hx, uint64_t hx_mask, uint64_t* __restrict__ hxl) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_s; i += stride) {
        const uint64_t key = S[i], hm = key >> RTK_POS_BITS, pos = key & RTK_POS_MASK;
        const uint64_t r = static_cast<uint64_t>(rank[hm >> 6]) + static_cast<uint64_t>(__popcll(bitmap[hm >> 6] & ((1ull << (hm & 63ull)) - 1ull)));
        uint32_t lo = 0, hi = n_unitigs;
        while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (uoff[mid] <= pos) lo = mid; else hi = mid; }
        const uint64_t gi = base + i;
        { // the two words of the place: the h + 1 bases behind and in front of the h-mer (zeros where the unitig ends), then unitig << 32 | following-bases-exist (a_ok) << 31 | offset
            const uint64_t u0 = uoff[lo], u1 = uoff[lo + 1], nbf = static_cast<uint64_t>(h) + 1ull;
            const bool a_ok = pos + static_cast<uint64_t>(h) + nbf <= u1, b_ok = pos >= u0 + nbf;
            uint64_t after = 0, before = 0;
            if (a_ok) for (uint64_t x = 0; x < nbf; ++x) { const uint64_t q_ = pos + static_cast<uint64_t>(h) + x; after = (after << 2) | ((useq[q_ >> 5] >> (2ull * (q_ & 31ull))) & 3ull); }
            if (b_ok) for (uint64_t x = 0; x < nbf; ++x) { const uint64_t q_ = pos - nbf + x; before = (before << 2) | ((useq[q_ >> 5] >> (2ull * (q_ & 31ull))) & 3ull); }
            hxl[2 * gi + r + 1] = (after << 32) | before;
            uint64_t fwd = 0; for (int x = 0; x < h; ++x) { const uint64_t q_ = pos + static_cast<uint64_t>(x); fwd = (fwd << 2) | ((useq[q_ >> 5] >> (2ull * (q_ & 31ull))) & 3ull); }
            hxl[2 * gi + r + 2] = (fwd != hm ? (1ull << 63) : 0ull) | (static_cast<uint64_t>(lo) << 32) | (a_ok ? (1ull << 31) : 0ull) | (pos - u0); // bit 63: the unitig spells the reverse complement of the (canonical) key
        }
        if (i == 0 || (S[i - 1] >> RTK_POS_BITS) != hm) { // first key of its h-mer: the count word and the slot (a range of bins never splits an h-mer)
            uint64_t a = i, b = i + 1, step = 1;
            while (b < n_s && (S[b] >> RTK_POS_BITS) == hm) { a = b; b = b + step < n_s ? b + step : n_s; step <<= 1; } // gallop, then bisect: (a same, b differs or the end)
            while (b - a > 1) { const uint64_t mid = a + (b - a) / 2; if ((S[mid] >> RTK_POS_BITS) == hm) a = mid; else b = mid; }
            const uint64_t w = 2 * gi + r;
            hxl[w] = b - i;
            const uint64_t word = (hm << RTK_POS_BITS) | w;
            uint64_t q = rtk_hash64(hm) & hx_mask;
            while (atomicCAS(reinterpret_cast<unsigned long long*>(hx + q), static_cast<unsigned long long>(RTK_EMPTY_KEY), static_cast<unsigned long long>(word)) != RTK_EMPTY_KEY) q = (q + 1) & hx_mask;
        }
    }
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int grid_for(uint64_t items) { const uint64_t b = (items + RTK_TB_BLOCK - 1) / RTK_TB_BLOCK; return static_cast<int>(b < 1 ? 1 : (b > 65536 ? 65536 : b)); }
static void sync_check(const char* what) { rtk_check(hipGetLastError(), what); rtk_check(hipDeviceSynchronize(), what); }

} // namespace

namespace rtk {

void device_tables_build(const uint64_t* d_useq, const uint64_t* d_uoff, uint32_t n_unitigs, uint64_t n_bases, uint64_t n_kmers, int k, DeviceTables* out) {
    const bool trace = getenv("RTK_LOAD_TRACE") != nullptr;
    const TableSizes tsz = table_sizes(k, n_kmers, n_bases);
    const uint64_t n_words = (n_bases + 31) / 32;
    const double t0 = now_s();
    Dev d_err; d_err.alloc(4); rtk_check(hipMemset(d_err.p, 0, 4), "hipMemset");
    Dev ht, bf, bf1, hx, hxl, adj;
    // ---- half-k-mer index (first: its sort buffers are gone before the k-mer table is allocated) ----
    uint64_t hx_words = 1, hxl_words = 1;
    if (!tsz.hx) { hx.alloc(8); hxl.alloc(8); hipLaunchKernelGGL(k_fill_words, dim3(1), dim3(64), 0, 0, hx.as<uint64_t>(), 1ull, RTK_EMPTY_KEY); rtk_check(hipMemset(hxl.p, 0, 8), "hipMemset"); }
    else {
        const int h = tsz.h;
        if (n_bases > RTK_POS_MASK) throw std::runtime_error("half-k-mer index: more than 2^34 bases in the unitig pool (set RTK_INEXACT_ENUM=1)");
        if (n_unitigs >= (1u << 31)) throw std::runtime_error("half-k-mer index: more than 2^31 unitigs");
        { hipLaunchKernelGGL(k_check_unitig_lengths, dim3(grid_for(n_unitigs)), dim3(RTK_TB_BLOCK), 0, 0, d_uoff, n_unitigs, d_err.as<uint32_t>()); sync_check("k_check_unitig_lengths");
          uint32_t e = 0; rtk_check(hipMemcpy(&e, d_err.p, 4, hipMemcpyDeviceToHost), "hipMemcpy"); if (e & 2u) throw std::runtime_error("half-k-mer index: a unitig of more than 2^31 bases"); }
        const uint64_t n_pairs = n_bases - static_cast<uint64_t>(n_unitigs) * static_cast<uint64_t>(h - 1);
        const uint64_t bm_words = ((1ull << (2 * h)) + 63) / 64;
        const int nbits = 2 * h < 12 ? 2 * h : 12; const int bshift = 2 * h - nbits; const uint32_t n_bins = 1u << nbits;
        Dev bitmap, hist, rank;
        bitmap.alloc(8 * bm_words); rtk_check(hipMemset(bitmap.p, 0, 8 * bm_words), "hipMemset");
        hist.alloc(8ull * n_bins); rtk_check(hipMemset(hist.p, 0, 8ull * n_bins), "hipMemset");
        hipLaunchKernelGGL(k_hx_mark, dim3(grid_for(n_words)), dim3(RTK_TB_BLOCK), 0, 0, d_useq, d_uoff, n_unitigs, n_bases, h, bshift, n_bins, bitmap.as<uint64_t>(), hist.as<unsigned long long>());
        sync_check("k_hx_mark");
        // rank of an h-mer among the distinct ones = popcount prefix of the bitmap
        rank.alloc(4 * bm_words);
        { PopcountOf pc; pc.w = bitmap.as<uint64_t>();
          auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), pc);
          size_t tb = 0; rtk_check(rocprim::exclusive_scan(nullptr, tb, in, rank.as<uint32_t>(), 0u, static_cast<size_t>(bm_words), rocprim::plus<uint32_t>()), "rocprim::exclusive_scan");
          Dev tmp; tmp.alloc(tb);
          rtk_check(rocprim::exclusive_scan(tmp.p, tb, in, rank.as<uint32_t>(), 0u, static_cast<size_t>(bm_words), rocprim::plus<uint32_t>()), "rocprim::exclusive_scan");
          sync_check("rank of the h-mers"); }
        uint32_t last_rank = 0; uint64_t last_word = 0;
        rtk_check(hipMemcpy(&last_rank, rank.as<uint32_t>() + (bm_words - 1), 4, hipMemcpyDeviceToHost), "hipMemcpy"); rtk_check(hipMemcpy(&last_word, bitmap.as<uint64_t>() + (bm_words - 1), 8, hipMemcpyDeviceToHost), "hipMemcpy");
        const uint64_t uniq = static_cast<uint64_t>(last_rank) + static_cast<uint64_t>(__builtin_popcountll(last_word));
        if (2 * n_pairs + uniq >= (1ull << RTK_POS_BITS)) throw std::runtime_error("half-k-mer index: more than 2^34 list words (set RTK_INEXACT_ENUM=1)");
        uint64_t hslots = 16; while (hslots < 2 * uniq) hslots <<= 1;
        hx_words = hslots; hxl_words = 2 * n_pairs + uniq + 1;
        hx.alloc(8 * hx_words); hxl.alloc(8 * hxl_words);
        hipLaunchKernelGGL(k_fill_words, dim3(grid_for(hx_words)), dim3(RTK_TB_BLOCK), 0, 0, hx.as<uint64_t>(), hx_words, RTK_EMPTY_KEY);
        rtk_check(hipMemset(hxl.as<uint64_t>() + (hxl_words - 1), 0, 8), "hipMemset");
        // ranges of bins whose keys fit (twice: the sort) into the memory left
        std::vector<unsigned long long> hh(n_bins); rtk_check(hipMemcpy(hh.data(), hist.p, 8ull * n_bins, hipMemcpyDeviceToHost), "hipMemcpy");
        { uint64_t s = 0; for (uint32_t b = 0; b < n_bins; ++b) s += hh[b]; if (s != n_pairs) throw std::runtime_error("half-k-mer index: the device counted another number of h-mer starts than the unitig lengths give"); }
        size_t fr = 0, tot = 0; rtk_check(hipMemGetInfo(&fr, &tot), "hipMemGetInfo");
        // (the k-mer table and its filters are allocated after this: leave them their room)
        const uint64_t later = 16 * tsz.ht_slots + 8 * tsz.bf_words + 8 * tsz.bf1_words + 32ull * n_unitigs + (1ull << 30);
        uint64_t cap = fr > later ? (static_cast<uint64_t>(fr) - later) / 2 / 16 : (1ull << 24); // (half of it: a caller may be reserving its work areas on another thread)
        if (cap > (1ull << 31)) cap = 1ull << 31;
        { const char* e = getenv("RTK_HX_PART_KEYS"); if (e) cap = strtoull(e, nullptr, 10); }
        if (cap < 1024) cap = 1024;
        std::vector<std::pair<uint32_t, uint32_t> > parts; uint64_t max_part = 0;
        for (uint32_t b = 0; b < n_bins;) { uint64_t s = hh[b]; uint32_t e = b + 1; while (e < n_bins && s + hh[e] <= cap) { s += hh[e]; ++e; } parts.push_back(std::make_pair(b, e)); if (s > max_part) max_part = s; b = e; }
        Dev keys, alt, top, tmp; keys.alloc(8 * max_part); alt.alloc(8 * max_part); top.alloc(8);
        size_t tb = 0; { rocprim::double_buffer<uint64_t> db(keys.as<uint64_t>(), alt.as<uint64_t>()); rtk_check(rocprim::radix_sort_keys(nullptr, tb, db, static_cast<size_t>(max_part), 0, RTK_POS_BITS + 2 * h), "rocprim::radix_sort_keys"); }
        tmp.alloc(tb);
        uint64_t done = 0;
        for (size_t pi = 0; pi < parts.size(); ++pi) {
            uint64_t n_s = 0; for (uint32_t b = parts[pi].first; b < parts[pi].second; ++b) n_s += hh[b];
            if (n_s == 0) continue;
            rtk_check(hipMemset(top.p, 0, 8), "hipMemset");
            hipLaunchKernelGGL(k_hx_keys, dim3(grid_for(n_words)), dim3(RTK_TB_BLOCK), 0, 0, d_useq, d_uoff, n_unitigs, n_bases, h, bshift, parts[pi].first, parts[pi].second, keys.as<uint64_t>(), top.as<unsigned long long>());
            rtk_check(hipGetLastError(), "k_hx_keys");
            rocprim::double_buffer<uint64_t> db(keys.as<uint64_t>(), alt.as<uint64_t>());
            size_t tb2 = tb; rtk_check(rocprim::radix_sort_keys(tmp.p, tb2, db, static_cast<size_t>(n_s), 0, RTK_POS_BITS + 2 * h), "rocprim::radix_sort_keys");
            hipLaunchKernelGGL(k_hx_scatter, dim3(grid_for(n_s)), dim3(RTK_TB_BLOCK), 0, 0, db.current(), n_s, done, bitmap.as<uint64_t>(), rank.as<uint32_t>(), d_useq, h, d_uoff, n_unitigs, hx.as<uint64_t>(), hx_words - 1, hxl.as<uint64_t>());
            sync_check("half-k-mer index (keys, sort, lists)");
            unsigned long long wrote = 0; rtk_check(hipMemcpy(&wrote, top.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
            if (wrote != n_s) throw std::runtime_error("half-k-mer index: key count of a range differs from its histogram");
            done += n_s;
        }
        if (trace) fprintf(stderr, "[rtk load] device: half-k-mer index: %llu starts, %llu distinct, %zu range(s) of leading bits\n", static_cast<unsigned long long>(n_pairs), static_cast<unsigned long long>(uniq), parts.size());
    }
    const double t1 = now_s();
    // ---- k-mer table + filters ----
    const int bf1_off = tsz.bf1_words == 1 ? 1 : 0;
    ht.alloc(16 * tsz.ht_slots); bf.alloc(8 * tsz.bf_words); bf1.alloc(8 * tsz.bf1_words);
    hipLaunchKernelGGL(k_fill_slots, dim3(grid_for(tsz.ht_slots)), dim3(RTK_TB_BLOCK), 0, 0, ht.as<uint64_t>(), tsz.ht_slots);
    rtk_check(hipMemset(bf.p, 0, 8 * tsz.bf_words), "hipMemset");
    if (bf1_off) hipLaunchKernelGGL(k_fill_words, dim3(1), dim3(64), 0, 0, bf1.as<uint64_t>(), 1ull, ~0ull); else rtk_check(hipMemset(bf1.p, 0, 8 * tsz.bf1_words), "hipMemset");
    hipLaunchKernelGGL(k_tables_insert, dim3(grid_for(n_words)), dim3(RTK_TB_BLOCK), 0, 0, d_useq, d_uoff, n_unitigs, n_bases, k, ht.as<uint64_t>(), tsz.ht_slots, bf.as<uint64_t>(), tsz.bf_words - 1,
                       bf1.as<uint64_t>(), tsz.bf1_words * 64 - 1, bf1_off, 1, d_err.as<uint32_t>());
    sync_check("k_tables_insert");
    GraphView gv; memset(&gv, 0, sizeof(gv));
    gv.k = k; gv.n_unitigs = n_unitigs; gv.n_kmers = n_kmers; gv.ht_slots = tsz.ht_slots; gv.useq = d_useq; gv.uoff = d_uoff; gv.ht = ht.as<uint64_t>();
    gv.bf = bf.as<uint64_t>(); gv.bf_mask = tsz.bf_words - 1; gv.bf1 = bf1.as<uint64_t>(); gv.bf1_mask = tsz.bf1_words * 64 - 1;
    if (k > 31) { hipLaunchKernelGGL(k_tables_verify_wide, dim3(grid_for(n_words)), dim3(RTK_TB_BLOCK), 0, 0, gv, n_bases, d_err.as<uint32_t>()); sync_check("k_tables_verify_wide"); }
    uint32_t err = 0; rtk_check(hipMemcpy(&err, d_err.p, 4, hipMemcpyDeviceToHost), "hipMemcpy");
    if (err) throw std::runtime_error("k-mer occurs twice in the unitig file: not a compacted de Bruijn graph for this k");
    const double t2 = now_s();
    // ---- adjacency ----
    adj.alloc(32ull * n_unitigs);
    hipLaunchKernelGGL(k_tables_adjacency, dim3(grid_for(8ull * n_unitigs)), dim3(RTK_TB_BLOCK), 0, 0, gv, adj.as<uint32_t>());
    sync_check("k_tables_adjacency");
    const double t3 = now_s();
    out->ht = ht.take(); out->ht_bytes = 16 * tsz.ht_slots; out->ht_slots = tsz.ht_slots;
    out->bf = bf.take(); out->bf_bytes = 8 * tsz.bf_words; out->bf1 = bf1.take(); out->bf1_bytes = 8 * tsz.bf1_words;
    out->hx = hx.take(); out->hx_bytes = 8 * hx_words; out->hxl = hxl.take(); out->hxl_bytes = 8 * hxl_words;
    out->adj = adj.take(); out->adj_bytes = 32ull * n_unitigs;
    out->seconds[0] = t2 - t1; out->seconds[1] = t1 - t0; out->seconds[2] = t3 - t2; out->seconds[3] = t3 - t0;
    if (trace) fprintf(stderr, "[rtk load] device tables: half-k-mer index %.2f s, k-mer table + filters %.2f s, adjacency %.2f s\n", t1 - t0, t2 - t1, t3 - t2);
}

// the k-mer table alone (no filters), for the index build's colouring pass (rtk_index.hip): slots at load 0.7, any number of k-mers
void device_kmer_table(const uint64_t* d_useq, const uint64_t* d_uoff, uint32_t n_unitigs, uint64_t n_bases, uint64_t n_kmers, int k, void** ht_out, uint64_t* slots_out) {
    const uint64_t slots = n_kmers + n_kmers * 3 / 7 + 16, n_words = (n_bases + 31) / 32;
    Dev ht, d_err; ht.alloc(16 * slots); d_err.alloc(4); rtk_check(hipMemset(d_err.p, 0, 4), "hipMemset");
    hipLaunchKernelGGL(k_fill_slots, dim3(grid_for(slots)), dim3(RTK_TB_BLOCK), 0, 0, ht.as<uint64_t>(), slots);
    hipLaunchKernelGGL(k_tables_insert, dim3(grid_for(n_words)), dim3(RTK_TB_BLOCK), 0, 0, d_useq, d_uoff, n_unitigs, n_bases, k, ht.as<uint64_t>(), slots, static_cast<uint64_t*>(nullptr), 0ull, static_cast<uint64_t*>(nullptr), 0ull, 1, 0, d_err.as<uint32_t>());
    sync_check("k_tables_insert (index build)");
    uint32_t err = 0; rtk_check(hipMemcpy(&err, d_err.p, 4, hipMemcpyDeviceToHost), "hipMemcpy");
    if (err) throw std::runtime_error("a k-mer occurs twice in the unitigs");
    *ht_out = ht.take(); *slots_out = slots;
}

} // namespace rtk
