// GPU side of the index build (reference: `Ratatosk index`, src/Ratatosk.cpp:1066-1067 Bifrost build + src/Graph.cpp:1561 addCoverage).
// The reference builds its graph with Bifrost on the CPU; the data-parallel step of an index build that touches every base of the 30x short
// reads most often -- counting their k-mers -- is done here on the device, behind the C ABI (include/ratatosk_hip.h), for the index tool
// (csrc/tools/build_index.cpp --gpu):
//   rtk_index_count_kmers   canonical k-mers of the reads seen >= min_count times: every read position spells its k-mer (one lane per position,
//                           the window packed 2 bits per base without branches), the k-mers of the pass are radix-sorted (rocPRIM) and the first
//                           element of every run of >= min_count equal keys is kept. HBM-bound: 1 byte read + 8 bytes written per base, then the sort.
// (Mapping the reads back onto the unitigs for the colour sets and coverages -- what addCoverage computes -- runs on the host threads of the tool,
// by byte ranges of the read files; it is the next candidate for the device.)
// One-word k-mers (k <= 31) only; the tool keeps its CPU path for k = 63 and for gzip input. Own translation unit: rocPRIM's templates.
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../../include/ratatosk_hip.h"
#include "../common/fastx.hpp"
#include "../common/kmer.hpp"
#include "rtk_graph_tables.h"
#include "rtk_mem.h"
#include "rtk_types.h"

int rtk_fail(int code, const std::string& msg); // (rtk_device.hip)

namespace {

#define RTK_IDX_SENTINEL 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t idx_revcomp(uint64_t x, int k) { return rtk_revcomp(x, k); }
__device__ __forceinline__ uint64_t idx_hash(uint64_t x) { return rtk_hash64(x); }

// code of base c (A/a 0, C/c 1, G/g 2, T/t 3) or 4
__device__ __forceinline__ uint32_t idx_code(unsigned char c) {
    const unsigned char u = c & 0xDF; // upper case
    return u == 'A' ? 0u : (u == 'C' ? 1u : (u == 'G' ? 2u : (u == 'T' ? 3u : 4u)));
}

// One lane per character position of the chunk: the canonical k-mer that starts there (all k characters A/C/G/T, the separator between reads is
// not), kept when its hash falls into partition `part` of `n_part`. Survivors are appended to `keys` (wave-level compaction: one atomic per wave).
__global__ void k_index_kmers(const char* __restrict__ chars, uint64_t n, int k, uint32_t part, uint32_t n_part, uint64_t* __restrict__ keys, unsigned long long* __restrict__ top, uint64_t cap) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * blockDim.x; i0 < n; i0 += stride) {
        const uint64_t i = i0 + threadIdx.x;
        uint64_t km = 0; bool ok = i + static_cast<uint64_t>(k) <= n;
        if (ok) {
            for (int j = 0; j < k; ++j) { const uint32_t c = idx_code(static_cast<unsigned char>(chars[i + j])); ok = ok && c < 4u; km = (km << 2) | (c & 3u); }
        }
        uint64_t can = 0;
        if (ok) { const uint64_t rc = idx_revcomp(km, k); can = km <= rc ? km : rc; ok = n_part <= 1u || (idx_hash(can) >> 40) % n_part == part; }
        const uint64_t bal = __ballot(ok ? 1 : 0);
        if (bal) {
            const int lane = threadIdx.x & 63;
            unsigned long long base = 0;
            if (lane == __ffsll(static_cast<unsigned long long>(bal)) - 1) base = atomicAdd(top, static_cast<unsigned long long>(__popcll(bal)));
            base = __shfl(base, __ffsll(static_cast<unsigned long long>(bal)) - 1, 64);
            const uint64_t at = base + static_cast<uint64_t>(__popcll(bal & ((1ull << lane) - 1ull)));
            if (ok && at < cap) keys[at] = can;
        }
    }
}

// first element of every run of >= min_count equal keys of a sorted array
struct SolidHead {
    const uint64_t* keys; uint64_t n; uint32_t min_count;
    __device__ bool operator()(uint64_t i) const {
        const uint64_t x = keys[i];
        if (i != 0 && keys[i - 1] == x) return false;
        return i + min_count - 1 < n && keys[i + min_count - 1] == x;
    }
};
struct KeyAt { const uint64_t* keys; __device__ uint64_t operator()(uint64_t i) const { return keys[i]; } };

struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } void alloc(uint64_t bytes) { if (p) (void)hipFree(p); p = nullptr; rtk_check(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc (index build)"); } };
struct PinBuf { void* p = nullptr; ~PinBuf() { if (p) (void)hipHostFree(p); } void alloc(uint64_t bytes) { rtk_check(hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc (index build)"); } };

// the sequences of the input files, chunk after chunk: a chunk = the read sequences of one byte range of a plain file, separated by '\n'
// (not a base), parsed by `n_threads` threads; handed to `sink(chars, n)` one chunk at a time (calls are serialised).
template <class Sink>
bool for_each_sequence_chunk(const std::vector<std::string>& files, int n_threads, uint64_t chunk_bytes, Sink sink, std::string* err) {
    for (size_t f = 0; f < files.size(); ++f) {
        if (rtk::SampleSource::is_spec(files[f])) { // reads sampled from a reference on the fly (common/sample_source.hpp): pair ranges, generated by the threads
            std::shared_ptr<rtk::SampleSource> ss = rtk::SampleSource::get(files[f], err);
            if (!ss) return false;
            const uint64_t L = ss->read_len(), per = std::max<uint64_t>(1, chunk_bytes / (2 * (L + 1))), n_ch = (ss->n_pairs() + per - 1) / per;
            std::atomic<uint64_t> next(0); std::mutex m_sink;
            std::vector<std::thread> th;
            const int nt = n_threads < 1 ? 1 : n_threads;
            for (int t = 0; t < nt; ++t) th.emplace_back([&]() {
                std::string buf;
                for (;;) {
                    const uint64_t c = next.fetch_add(1);
                    if (c >= n_ch) break;
                    const uint64_t p0 = c * per, p1 = std::min<uint64_t>(ss->n_pairs(), p0 + per);
                    buf.assign(static_cast<size_t>((p1 - p0) * 2 * (L + 1)), '\n');
                    for (uint64_t p = p0; p < p1; ++p) ss->pair(p, &buf[static_cast<size_t>((p - p0) * 2 * (L + 1))], &buf[static_cast<size_t>((p - p0) * 2 * (L + 1) + L + 1)]);
                    std::lock_guard<std::mutex> lk(m_sink);
                    sink(buf.data(), buf.size());
                }
            });
            for (size_t t = 0; t < th.size(); ++t) th[t].join();
            continue;
        }
        if (!rtk::PlainChunks::is_plain(files[f])) { // gzip or unknown: one reader thread
            rtk::FastxReader rd; if (!rd.open(files[f], n_threads < 16 ? n_threads : 16)) { *err = "cannot open " + files[f]; return false; } // (a gzip file of several members is inflated on the threads, common/mgzip.hpp)
            std::string name, seq, qual, buf; buf.reserve(chunk_bytes);
            while (rd.next(name, seq, qual)) { buf += seq; buf.push_back('\n'); if (buf.size() >= chunk_bytes) { sink(buf.data(), buf.size()); buf.clear(); } }
            if (rd.failed()) { *err = files[f] + " ends in a damaged or cut-short gzip stream"; return false; }
            if (!buf.empty()) sink(buf.data(), buf.size());
            continue;
        }
        rtk::PlainChunks pc; if (!pc.open(files[f], chunk_bytes)) { *err = "cannot open " + files[f]; return false; }
        std::atomic<size_t> next(0); std::mutex m_sink; std::atomic<bool> bad(false);
        std::vector<std::thread> th;
        const int nt = n_threads < 1 ? 1 : n_threads;
        for (int t = 0; t < nt; ++t) th.emplace_back([&]() {
            std::string buf;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= pc.n_chunks() || bad) break;
                rtk::PackedReads r(false);
                if (!pc.parse_chunk(i, r)) { bad = true; break; }
                buf.clear(); buf.reserve(r.n_bases() + r.size());
                for (size_t x = 0; x < r.size(); ++x) { buf.append(r.seq(x), r.seq_len(x)); buf.push_back('\n'); }
                std::lock_guard<std::mutex> lk(m_sink);
                sink(buf.data(), buf.size());
            }
        });
        for (size_t t = 0; t < th.size(); ++t) th[t].join();
        if (bad) { *err = "read error on " + files[f]; return false; }
    }
    return true;
}


// ------------------------------------------------------------------------------------------------ unitigs (rtk_index_unitigs)
// The solid k-mers in a table of 16-byte slots {canonical k-mer, value}: value bits 0..3 = which of the four successors (x << 2 | b) of the canonical
// orientation are solid, bits 4..7 = which of the four predecessors (b in front), bit 8 = the k-mer lies on a unitig written here.
#define RTK_UT_CLAIMED 256ull
__device__ __forceinline__ uint64_t ut_find(const uint64_t* __restrict__ T, uint64_t slots, uint64_t can) {
    uint64_t s = __umul64hi(idx_hash(can), slots);
    for (;;) { const uint64_t key = T[2 * s]; if (key == can) return s; if (key == RTK_IDX_SENTINEL) return RTK_IDX_SENTINEL; s = s + 1 == slots ? 0 : s + 1; }
}
__device__ __forceinline__ uint32_t rev4(uint32_t n) { return ((n & 1u) << 3) | ((n & 2u) << 1) | ((n & 4u) >> 1) | ((n & 8u) >> 3); }
// edge bits of an ORIENTED k-mer from those of its canonical form: the successor by base b of the reverse complement is the predecessor by base 3 - b
__device__ __forceinline__ uint32_t ut_omask(uint32_t m, bool is_can) { return is_can ? (m & 255u) : (rev4((m >> 4) & 15u) | (rev4(m & 15u) << 4)); }
__device__ __forceinline__ uint32_t ut_mask_of(const uint64_t* __restrict__ T, uint64_t slots, uint64_t x, int k, uint64_t* slot_out) {
    const uint64_t rc = idx_revcomp(x, k), can = x <= rc ? x : rc;
    const uint64_t s = ut_find(T, slots, can); if (slot_out) *slot_out = s;
    return ut_omask(static_cast<uint32_t>(T[2 * s + 1]), x == can);
}
// the link the construction follows forwards from x (its oriented edge bits mx): the only successor of x, if x is its only predecessor
__device__ __forceinline__ bool ut_next(const uint64_t* __restrict__ T, uint64_t slots, int k, uint64_t kmask, uint64_t x, uint32_t mx, uint64_t* y, uint32_t* my, uint64_t* slot_y) {
    const uint32_t sc = mx & 15u; if (__popc(sc) != 1) return false;
    const uint64_t yy = ((x << 2) | static_cast<uint64_t>(__ffs(sc) - 1)) & kmask;
    const uint32_t m = ut_mask_of(T, slots, yy, k, slot_y); if (__popc(m >> 4) != 1) return false;
    *y = yy; *my = m; return true;
}
__device__ __forceinline__ bool ut_prev(const uint64_t* __restrict__ T, uint64_t slots, int k, uint64_t x, uint32_t mx) {
    const uint32_t pc = mx >> 4; if (__popc(pc) != 1) return false;
    const uint64_t yy = (x >> 2) | (static_cast<uint64_t>(__ffs(pc) - 1) << (2 * (k - 1)));
    return __popc(ut_mask_of(T, slots, yy, k, nullptr) & 15u) == 1;
}

__global__ void k_ut_fill(uint64_t* __restrict__ T, uint64_t slots) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < slots; i += stride) { T[2 * i] = RTK_IDX_SENTINEL; T[2 * i + 1] = 0; }
}
__global__ void k_ut_insert(const uint64_t* __restrict__ solid, uint64_t n, uint64_t* __restrict__ T, uint64_t slots) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t c = solid[i]; uint64_t s = __umul64hi(idx_hash(c), slots);
        while (atomicCAS(reinterpret_cast<unsigned long long*>(T + 2 * s), static_cast<unsigned long long>(RTK_IDX_SENTINEL), static_cast<unsigned long long>(c)) != RTK_IDX_SENTINEL) s = s + 1 == slots ? 0 : s + 1;
    }
}
__global__ void k_ut_edges(const uint64_t* __restrict__ solid, uint64_t n, int k, uint64_t* __restrict__ T, uint64_t slots) {
    const uint64_t kmask = (1ull << (2 * k)) - 1ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t c = solid[i]; uint64_t m = 0;
        for (uint64_t b = 0; b < 4; ++b) {
            const uint64_t y = ((c << 2) | b) & kmask, yr = idx_revcomp(y, k); if (ut_find(T, slots, y <= yr ? y : yr) != RTK_IDX_SENTINEL) m |= 1ull << b;
            const uint64_t z = (c >> 2) | (b << (2 * (k - 1))), zr = idx_revcomp(z, k); if (ut_find(T, slots, z <= zr ? z : zr) != RTK_IDX_SENTINEL) m |= 16ull << b;
        }
        T[2 * ut_find(T, slots, c) + 1] = m;
    }
}
// Every maximal chain of mutually unique links is walked from both of its end k-mers; the end whose canonical k-mer is the smaller one owns it (tools/
// build_index.cpp fast_unitigs: the same rules, so that the two produce the same unitigs). An owner reports the oriented k-mer it starts from, the number of
// k-mers, the smallest canonical k-mer on the chain (its seed: unitigs are numbered by it) and whether that one reads backwards on the walk (the unitig is
// then the reverse complement of the walk). record == nullptr: count the owners only.
__global__ void k_ut_chains(const uint64_t* __restrict__ solid, uint64_t n, int k, const uint64_t* __restrict__ T, uint64_t slots,
                            unsigned long long* __restrict__ n_chains, uint64_t cap, uint64_t* __restrict__ start, uint64_t* __restrict__ seed, uint32_t* __restrict__ len_rev, uint32_t* __restrict__ too_long) {
    const uint64_t kmask = (1ull << (2 * k)) - 1ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t s = solid[i];
        const uint32_t ms = ut_mask_of(T, slots, s, k, nullptr);
        uint64_t y = 0; uint32_t my = 0;
        const bool has_fw = ut_next(T, slots, k, kmask, s, ms, &y, &my, nullptr), has_bw = ut_prev(T, slots, k, s, ms);
        if (has_fw && has_bw) continue; // inside a chain (or on a closed loop)
        uint64_t x = has_bw ? idx_revcomp(s, k) : s; uint32_t mx = has_bw ? ut_omask(ms, false) : ms; // walk inwards from this end
        const uint64_t x0 = x;
        uint64_t len = 1, mc = s; bool m_fw = (x == s);
        while (ut_next(T, slots, k, kmask, x, mx, &y, &my, nullptr)) {
            x = y; mx = my; ++len;
            const uint64_t r = idx_revcomp(x, k), c = x <= r ? x : r;
            if (c < mc) { mc = c; m_fw = (x == c); }
            if (len > n) break;
        }
        const uint64_t xr = idx_revcomp(x, k), end_c = x <= xr ? x : xr;
        if (len > 1 && end_c == s) continue; // the chain comes back to its own first k-mer (hairpin): left to the plain construction
        if (end_c < s) continue;            // the other end owns the chain
        if (len > n) continue;
        if (len >= (1ull << 31)) { atomicOr(too_long, 1u); continue; }
        const unsigned long long at = atomicAdd(n_chains, 1ull);
        if (start && at < cap) { start[at] = x0; seed[at] = mc; len_rev[at] = static_cast<uint32_t>(len) | (m_fw ? 0u : 0x80000000u); }
    }
}
struct ChainBases { const uint32_t* len_rev; const uint32_t* order; int k; __device__ uint64_t operator()(uint64_t j) const { return static_cast<uint64_t>(len_rev[order[j]] & 0x7FFFFFFFu) + static_cast<uint64_t>(k - 1); } };
// chain order[j] written as unitig j at seq_off[j]: walked again from its start, every k-mer claimed (a k-mer claimed twice: *clash)
__global__ void k_ut_write(uint64_t n_ch, const uint32_t* __restrict__ order, const uint64_t* __restrict__ start, const uint32_t* __restrict__ len_rev, const uint64_t* __restrict__ seq_off,
                           int k, uint64_t* __restrict__ T, uint64_t slots, char* __restrict__ pool, uint32_t* __restrict__ clash) {
    const uint64_t kmask = (1ull << (2 * k)) - 1ull, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < n_ch; j += stride) {
        const uint32_t c = order[j]; const uint64_t len = len_rev[c] & 0x7FFFFFFFu; const bool rev = (len_rev[c] >> 31) != 0;
        const uint64_t L = len + static_cast<uint64_t>(k - 1); char* out = pool + seq_off[j];
        auto put = [&](uint64_t pos, uint32_t b) { if (!rev) out[pos] = "ACGT"[b]; else out[L - 1 - pos] = "TGCA"[b]; }; // base b at position pos of the walk
        uint64_t x = start[c], slot = 0; uint32_t mx = ut_mask_of(T, slots, x, k, &slot);
        for (int q = 0; q < k; ++q) put(static_cast<uint64_t>(q), static_cast<uint32_t>((x >> (2 * (k - 1 - q))) & 3ull));
        if (atomicOr(reinterpret_cast<unsigned long long*>(T + 2 * slot + 1), static_cast<unsigned long long>(RTK_UT_CLAIMED)) & RTK_UT_CLAIMED) atomicOr(clash, 1u);
        for (uint64_t q = 1; q < len; ++q) {
            uint64_t y = 0; uint32_t my = 0;
            if (!ut_next(T, slots, k, kmask, x, mx, &y, &my, &slot)) { atomicOr(clash, 2u); break; }
            x = y; mx = my; put(static_cast<uint64_t>(k - 1) + q, static_cast<uint32_t>(x & 3ull));
            if (atomicOr(reinterpret_cast<unsigned long long*>(T + 2 * slot + 1), static_cast<unsigned long long>(RTK_UT_CLAIMED)) & RTK_UT_CLAIMED) atomicOr(clash, 1u);
        }
    }
}
struct Unclaimed { const uint64_t* solid; const uint64_t* T; uint64_t slots; __device__ bool operator()(uint64_t i) const { return !(T[2 * ut_find(T, slots, solid[i]) + 1] & RTK_UT_CLAIMED); } };
struct Iota32 { __device__ uint32_t operator()(uint64_t i) const { return static_cast<uint32_t>(i); } };


// ------------------------------------------------------------------------------------------------ colours and coverage (rtk_index_colour_*)
// unitig sequences (characters, unitig u at off[u]) -> the 2-bit pool of the flat graph (base p at bits 2 (p & 31) of word p >> 5)
__global__ void k_col_pack(const char* __restrict__ pool, uint64_t n_bases, uint64_t* __restrict__ useq, uint64_t n_words) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        uint64_t x = 0;
        for (uint64_t j = 0; j < 32 && 32 * w + j < n_bases; ++j) x |= static_cast<uint64_t>(idx_code(static_cast<unsigned char>(pool[32 * w + j])) & 3u) << (2 * j);
        useq[w] = x;
    }
}
// One lane per character position of a chunk of reads (sequences separated by '\n'): its k-mer looked up in the unitig table. A run of consecutive
// positions of one read on one unitig is one EVENT (unitig << 32 | id of the read) and one addition of its length to the unitig's coverage, made by
// the first lane of the run inside its wave (a run that crosses a wave boundary gives two events: duplicates go when the events are sorted).
__global__ void k_col_map(const char* __restrict__ chars, uint64_t n, int k, const uint64_t* __restrict__ starts, const uint32_t* __restrict__ ids, uint32_t n_reads,
                          const uint64_t* __restrict__ ht, uint64_t slots, unsigned long long* __restrict__ cov, uint64_t* __restrict__ events, unsigned long long* __restrict__ top, uint64_t cap) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * blockDim.x; i0 < n; i0 += stride) { // (whole waves take part in every round)
        const uint64_t i = i0 + threadIdx.x;
        uint64_t km = 0; bool ok = i + static_cast<uint64_t>(k) <= n;
        if (ok) for (int j = 0; j < k; ++j) { const uint32_t c = idx_code(static_cast<unsigned char>(chars[i + j])); ok = ok && c < 4u; km = (km << 2) | (c & 3u); }
        uint32_t u = 0xFFFFFFFFu;
        if (ok) {
            const uint64_t rc = idx_revcomp(km, k), can = km <= rc ? km : rc;
            uint64_t s = __umul64hi(idx_hash(can), slots);
            for (;;) { const uint64_t key = ht[2 * s]; if (key == can) { u = static_cast<uint32_t>(ht[2 * s + 1] >> 32); break; } if (key == RTK_IDX_SENTINEL) break; s = s + 1 == slots ? 0 : s + 1; }
        }
        const bool hit = u != 0xFFFFFFFFu;
        const uint32_t pu = __shfl_up(u, 1, 64); // (position i - 1 with a k-mer on the same unitig: the same read, its k-mer holds no separator)
        const bool head = hit && (lane == 0 || pu != u);
        const uint64_t heads = __ballot(head ? 1 : 0), hits = __ballot(hit ? 1 : 0);
        if (!heads) continue;
        const unsigned long long base = __shfl(lane == 0 ? atomicAdd(top, static_cast<unsigned long long>(__popcll(heads))) : 0ull, 0, 64);
        if (head) {
            const uint64_t above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1); // the next head of the wave, if any
            const int nxt = above ? __ffsll(static_cast<unsigned long long>(above)) - 1 : 64;
            const uint64_t span = (nxt == 64 ? ~0ull : ((1ull << nxt) - 1ull)) & ~((1ull << lane) - 1ull);
            atomicAdd(cov + u, static_cast<unsigned long long>(__popcll(hits & span)));
            uint32_t lo = 0, hi = n_reads; // the read of position i: the last one that starts at or before it
            while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (starts[mid] <= i) lo = mid; else hi = mid; }
            const uint64_t at = base + static_cast<uint64_t>(__popcll(heads & ((1ull << lane) - 1ull)));
            if (at < cap) events[at] = (static_cast<uint64_t>(u) << 32) | ids[lo];
        }
    }
}

struct ColourJob {
    int device = 0, k = 31; uint32_t n_unitigs = 0;
    DevBuf useq, uoff, ht, cov, events, alt, top, tmp;
    uint64_t slots = 0, cap = 0, n_events = 0; // n_events: sorted, distinct events at the front of `events`
    DevBuf d_chars[2], d_starts[2], d_ids[2]; PinBuf h_chars[2], h_starts[2], h_ids[2]; uint64_t chunk_cap = 0, reads_cap = 0;
    hipStream_t st[2] = {nullptr, nullptr}; int slot = 0;
    std::mutex m; uint64_t bases = 0, chunks = 0, compactions = 0; double t_table = 0.0;
    std::chrono::steady_clock::time_point t0;
    ~ColourJob() { if (st[0]) (void)hipStreamDestroy(st[0]); if (st[1]) (void)hipStreamDestroy(st[1]); }
    // the events so far sorted, the distinct ones kept (both streams idle)
    void compact() {
        rtk_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        unsigned long long n = 0; rtk_check(hipMemcpy(&n, top.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
        if (n > cap) throw std::runtime_error("more (unitig, read) events than the device buffer holds between two compactions (RTK_INDEX_EVENTS: number of events to make room for)");
        if (n == n_events) return;
        ++compactions;
        rocprim::double_buffer<uint64_t> db(static_cast<uint64_t*>(events.p), static_cast<uint64_t*>(alt.p));
        size_t tb = 0; rtk_check(rocprim::radix_sort_keys(nullptr, tb, db, static_cast<size_t>(n), 0, 64), "rocprim::radix_sort_keys");
        tmp.alloc(tb); rtk_check(rocprim::radix_sort_keys(tmp.p, tb, db, static_cast<size_t>(n), 0, 64), "rocprim::radix_sort_keys");
        uint64_t* sorted = db.current(); uint64_t* other = db.alternate();
        DevBuf d_n; d_n.alloc(8);
        size_t ub = 0; rtk_check(rocprim::unique(nullptr, ub, sorted, other, static_cast<unsigned long long*>(d_n.p), static_cast<size_t>(n)), "rocprim::unique");
        tmp.alloc(ub); rtk_check(rocprim::unique(tmp.p, ub, sorted, other, static_cast<unsigned long long*>(d_n.p), static_cast<size_t>(n)), "rocprim::unique");
        rtk_check(hipDeviceSynchronize(), "events sorted");
        unsigned long long nu = 0; rtk_check(hipMemcpy(&nu, d_n.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
        if (other != static_cast<uint64_t*>(events.p)) rtk_check(hipMemcpy(events.p, other, 8 * nu, hipMemcpyDeviceToDevice), "hipMemcpy");
        n_events = nu; rtk_check(hipMemcpy(top.p, &nu, 8, hipMemcpyHostToDevice), "hipMemcpy");
    }
};

} // namespace

extern "C" int rtk_index_count_kmers(int device, int k, const char* const* files, int n_files, uint32_t min_count, int n_threads, uint64_t** solid_out, uint64_t* n_solid) {
    if (!files || n_files <= 0 || !solid_out || !n_solid) return rtk_fail(RTK_ERR_ARG, "rtk_index_count_kmers: null argument");
    if (k < 3 || k > 31 || !(k & 1)) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_count_kmers: one-word k-mers only (odd k <= 31)");
    if (min_count < 1) min_count = 1;
    if (rtk_device_count() <= device || device < 0) return rtk_fail(RTK_ERR_NO_DEVICE, "rtk_index_count_kmers: no such HIP device (no CPU fallback)");
    try {
        rtk_check(hipSetDevice(device), "hipSetDevice");
        std::vector<std::string> fl(files, files + n_files);
        // passes: the k-mers of one pass (one hash partition of the k-mer space) must fit twice (radix sort) next to what else lives on the device
        uint64_t total_bytes = 0;
        for (size_t f = 0; f < fl.size(); ++f) { if (rtk::SampleSource::is_spec(fl[f])) { std::string e_; std::shared_ptr<rtk::SampleSource> ss = rtk::SampleSource::get(fl[f], &e_); if (!ss) return rtk_fail(RTK_ERR_IO, "rtk_index_count_kmers: " + e_); total_bytes += 2 * ss->n_bases(); continue; } FILE* fp = fopen(fl[f].c_str(), "rb"); if (!fp) return rtk_fail(RTK_ERR_IO, "rtk_index_count_kmers: cannot open " + fl[f]); fseek(fp, 0, SEEK_END); total_bytes += static_cast<uint64_t>(ftell(fp)); fclose(fp); }
        size_t fr = 0, tot = 0; rtk_check(hipMemGetInfo(&fr, &tot), "hipMemGetInfo");
        if (getenv("RTK_INDEX_TRACE")) fprintf(stderr, "rtk_index_count_kmers: inputs sized (%.1f GB), %.1f GB of device memory free\n", total_bytes / 1e9, fr / 1e9);
        const uint64_t est_kmers = total_bytes / 2 + (1u << 20); // FASTQ: half of the bytes are bases (gzip input: a multiple of it; the capacity test below catches that)
        uint64_t cap = static_cast<uint64_t>(fr) / 10 * 7 / 16; // 70 % of the free memory for keys + their sort buffer (the rest: two chunks of text, the sort's histograms)
        { const char* e = getenv("RTK_INDEX_CAP"); if (e) cap = strtoull(e, nullptr, 10); }
        if (cap < (1u << 20)) cap = 1u << 20;
        uint32_t n_part = static_cast<uint32_t>((est_kmers + cap - 1) / cap); if (n_part < 1) n_part = 1;
        uint64_t chunk_bytes = 256ull << 20;
        { const char* e = getenv("RTK_INDEX_CHUNK"); if (e && strtoull(e, nullptr, 10) >= 1024) chunk_bytes = strtoull(e, nullptr, 10); } // developer / tests: small chunks, so that long records are cut into pieces
        const uint64_t chunk_slack = std::min<uint64_t>(64ull << 20, chunk_bytes / 4);
        std::vector<uint64_t> solid;
        // Several partitions = several passes over the reads. The text of the first pass is kept in host memory when it fits into half of what is free there
        // (a 3 Gb x 30x set: 90 GB of sequences, sampled or parsed ONCE instead of once per partition -- 13 passes at 0.5 Gb/s of host-side sampling were 36 minutes)
        std::vector<std::string> kept; bool keep_text = false, kept_complete = false;
        std::vector<size_t> run_start; // solid[run_start[p] ..): the sorted k-mers of partition p
        const bool trace = getenv("RTK_INDEX_TRACE") != nullptr; const auto t_begin = std::chrono::steady_clock::now();
        auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
        { uint64_t avail = static_cast<uint64_t>(sysconf(_SC_AVPHYS_PAGES)) * static_cast<uint64_t>(sysconf(_SC_PAGE_SIZE));
          { // a container's memory limit is not what sysconf reports: what is left under the control group's limit, if there is one
            unsigned long long lim = 0, cur = 0; bool have = false;
            if (FILE* f1 = fopen("/sys/fs/cgroup/memory.max", "r")) { have = fscanf(f1, "%llu", &lim) == 1; fclose(f1); if (have) { if (FILE* f2 = fopen("/sys/fs/cgroup/memory.current", "r")) { if (fscanf(f2, "%llu", &cur) != 1) cur = 0; fclose(f2); } } }
            if (have && lim > cur && lim - cur < avail) avail = lim - cur; }
          const char* e = getenv("RTK_INDEX_KEEP_TEXT"); keep_text = e ? atoi(e) != 0 : (est_kmers + est_kmers / 8 < avail / 2); // (the caller's own tables come on top of it later: half of what is left, no more)
          if (trace) fprintf(stderr, "rtk_index_count_kmers: %.1f GB of host memory to be had, text %s\n", avail / 1e9, keep_text ? "kept across partitions" : "read again for every partition"); }
        for (bool done = false; !done;) {
            done = true; solid.clear(); run_start.clear(); if (!kept_complete) kept.clear(); // (a restart with more partitions keeps the text of the complete first pass)
            const uint64_t cap_p = n_part == 1 ? std::min<uint64_t>(cap, est_kmers + est_kmers / 8) : cap;
            DevBuf d_keys, d_alt, d_top, d_chars[2], d_sel, d_nsel;
            d_keys.alloc(8 * cap_p); d_alt.alloc(8 * cap_p); d_top.alloc(8); d_nsel.alloc(8);
            d_chars[0].alloc(chunk_bytes + chunk_slack); d_chars[1].alloc(chunk_bytes + chunk_slack);
            PinBuf h_chars[2]; h_chars[0].alloc(chunk_bytes + chunk_slack); h_chars[1].alloc(chunk_bytes + chunk_slack);
            hipStream_t st[2]; rtk_check(hipStreamCreate(&st[0]), "hipStreamCreate"); rtk_check(hipStreamCreate(&st[1]), "hipStreamCreate");
            for (uint32_t part = 0; part < n_part && done; ++part) {
                rtk_check(hipMemset(d_top.p, 0, 8), "hipMemset");
                int slot = 0; std::string err; uint64_t n_sunk = 0, b_sunk = 0;
                if (trace) fprintf(stderr, "rtk_index_count_kmers: partition %u of %u starts at %.1f s (room for %llu k-mers)\n", part + 1, n_part, since(), static_cast<unsigned long long>(cap_p));
                auto sink = [&](const char* chars, size_t n) {
                    if (trace && (++n_sunk & 31u) == 0) fprintf(stderr, "rtk_index_count_kmers:   %llu chunks, %.1f GB of text at %.1f s\n", static_cast<unsigned long long>(n_sunk), b_sunk / 1e9, since());
                    b_sunk += n; // one chunk: pinned copy, H2D and the k-mer kernel on the slot's stream (the other slot's work overlaps the next parse)
                    for (size_t off = 0; off < n;) {
                        const size_t piece = std::min<size_t>(n - off, chunk_bytes + chunk_slack);
                        rtk_check(hipStreamSynchronize(st[slot]), "hipStreamSynchronize");
                        memcpy(h_chars[slot].p, chars + off, piece);
                        rtk_check(hipMemcpyAsync(d_chars[slot].p, h_chars[slot].p, piece, hipMemcpyHostToDevice, st[slot]), "hipMemcpyAsync");
                        hipLaunchKernelGGL(k_index_kmers, dim3(4096), dim3(256), 0, st[slot], static_cast<const char*>(d_chars[slot].p), static_cast<uint64_t>(piece), k, part, n_part,
                                           static_cast<uint64_t*>(d_keys.p), static_cast<unsigned long long*>(d_top.p), cap_p);
                        rtk_check(hipGetLastError(), "kernel launch (k_index_kmers)");
                        slot ^= 1; off += (off + piece < n) ? piece - static_cast<size_t>(k - 1) : piece; // a cut inside a sequence: the next piece starts k - 1 characters back, so that every window is seen once
                    }
                };
                if (kept_complete) { for (size_t c = 0; c < kept.size(); ++c) sink(kept[c].data(), kept[c].size()); }
                else if (keep_text && n_part > 1) {
                    auto sink_keep = [&](const char* chars, size_t n) { kept.push_back(std::string(chars, n)); sink(chars, n); };
                    if (!for_each_sequence_chunk(fl, n_threads, chunk_bytes, sink_keep, &err)) { (void)hipStreamDestroy(st[0]); (void)hipStreamDestroy(st[1]); return rtk_fail(RTK_ERR_IO, "rtk_index_count_kmers: " + err); }
                    kept_complete = true;
                }
                else if (!for_each_sequence_chunk(fl, n_threads, chunk_bytes, sink, &err)) { (void)hipStreamDestroy(st[0]); (void)hipStreamDestroy(st[1]); return rtk_fail(RTK_ERR_IO, "rtk_index_count_kmers: " + err); }
                rtk_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
                unsigned long long n_keys = 0; rtk_check(hipMemcpy(&n_keys, d_top.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
                if (n_keys > cap_p) { // more k-mers than estimated: more partitions, again
                    n_part = static_cast<uint32_t>((n_keys * static_cast<uint64_t>(n_part) + cap - 1) / cap) + 1; done = false; break;
                }
                if (n_keys == 0) continue;
                if (trace) fprintf(stderr, "rtk_index_count_kmers:   %llu k-mers on the device at %.1f s; sorting\n", n_keys, since());
                // sort, then the first key of every run of >= min_count
                rocprim::double_buffer<uint64_t> db(static_cast<uint64_t*>(d_keys.p), static_cast<uint64_t*>(d_alt.p));
                size_t tb = 0; rtk_check(rocprim::radix_sort_keys(nullptr, tb, db, static_cast<size_t>(n_keys), 0, 2 * k), "rocprim::radix_sort_keys");
                DevBuf d_tmp; d_tmp.alloc(tb);
                rtk_check(rocprim::radix_sort_keys(d_tmp.p, tb, db, static_cast<size_t>(n_keys), 0, 2 * k), "rocprim::radix_sort_keys");
                if (trace) { rtk_check(hipDeviceSynchronize(), "radix sort"); fprintf(stderr, "rtk_index_count_kmers:   sorted at %.1f s\n", since()); }
                const uint64_t* sorted = db.current(); uint64_t* other = db.alternate();
                SolidHead pred; pred.keys = sorted; pred.n = n_keys; pred.min_count = min_count;
                KeyAt at; at.keys = sorted;
                auto idx = rocprim::make_counting_iterator<uint64_t>(0);
                auto vals = rocprim::make_transform_iterator(idx, at);
                size_t sb = 0; // select with a flag iterator: the flag of element i is the predicate on its index
                auto flags = rocprim::make_transform_iterator(idx, pred);
                rtk_check(rocprim::select(nullptr, sb, vals, flags, other, static_cast<unsigned long long*>(d_nsel.p), static_cast<size_t>(n_keys)), "rocprim::select");
                d_tmp.alloc(sb);
                rtk_check(rocprim::select(d_tmp.p, sb, vals, flags, other, static_cast<unsigned long long*>(d_nsel.p), static_cast<size_t>(n_keys)), "rocprim::select");
                rtk_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
                unsigned long long n_sel = 0; rtk_check(hipMemcpy(&n_sel, d_nsel.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
                const size_t old = solid.size(); solid.resize(old + n_sel); run_start.push_back(old);
                if (trace) fprintf(stderr, "rtk_index_count_kmers: partition %u of %u: %llu k-mers, %llu solid (at %.1f s%s)\n", part + 1, n_part, n_keys, n_sel, since(), kept_complete ? ", text kept in host memory" : "");
                if (n_sel) rtk_check(hipMemcpy(solid.data() + old, other, 8ull * n_sel, hipMemcpyDeviceToHost), "hipMemcpy");
            }
            (void)hipStreamDestroy(st[0]); (void)hipStreamDestroy(st[1]);
        }
        uint64_t* out = static_cast<uint64_t*>(malloc(8 * (solid.size() ? solid.size() : 1)));
        if (!out) return rtk_fail(RTK_ERR_IO, "rtk_index_count_kmers: out of host memory");
        if (run_start.size() <= 1) { if (!solid.empty()) memcpy(out, solid.data(), 8 * solid.size()); }
        else {
            // the partitions are sorted each and a k-mer lives in one of them: merged by ranges of the key space, one range per thread (every thread finds its
            // stretch of every partition by bisection; the ranges before it give it its place in the output)
            const size_t P = run_start.size(); run_start.push_back(solid.size());
            const int T = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
            std::vector<std::vector<size_t> > cut(static_cast<size_t>(T) + 1, std::vector<size_t>(P));
            const unsigned __int128 span = static_cast<unsigned __int128>(1) << (2 * k);
            for (int t = 0; t <= T; ++t) for (size_t r = 0; r < P; ++r) {
                if (t == T) { cut[t][r] = run_start[r + 1]; continue; }
                const uint64_t v = static_cast<uint64_t>(span * static_cast<unsigned __int128>(t) / static_cast<unsigned __int128>(T));
                cut[t][r] = static_cast<size_t>(std::lower_bound(solid.begin() + static_cast<std::ptrdiff_t>(run_start[r]), solid.begin() + static_cast<std::ptrdiff_t>(run_start[r + 1]), v) - solid.begin());
            }
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
                size_t at = 0; for (size_t r = 0; r < P; ++r) at += cut[t][r] - run_start[r];
                std::vector<size_t> head(cut[t]); const std::vector<size_t>& end = cut[t + 1];
                for (;;) { size_t best = P; uint64_t bv = 0; for (size_t r = 0; r < P; ++r) if (head[r] < end[r] && (best == P || solid[head[r]] < bv)) { best = r; bv = solid[head[r]]; } if (best == P) break; out[at++] = bv; ++head[best]; }
            });
            for (size_t t = 0; t < th.size(); ++t) th[t].join();
        }
        if (trace) fprintf(stderr, "rtk_index_count_kmers: %zu solid k-mers from %u partition(s) in %.1f s\n", solid.size(), n_part, since());
        *solid_out = out; *n_solid = solid.size();
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_index_count_kmers: ") + e.what()); }
    return RTK_OK;
}


// Unitigs of the solid k-mers: every maximal chain of mutually unique links that does not meet itself, oriented so that its smallest canonical k-mer reads
// forwards, in the order of those k-mers -- what tools/build_index.cpp fast_unitigs builds on the host threads (the rules are restated there and here;
// tests/test_index_build.py holds both to the plain construction and to oracle/oracle_index.py). Chains that meet themselves (closed loops, hairpins through
// a reverse complement) are not built: their k-mers come back in *left (sorted) for the caller's plain construction.
extern "C" int rtk_index_unitigs(int device, int k, const uint64_t* solid, uint64_t n_solid, char** seq_pool, uint64_t** seq_off, uint64_t** seeds, uint64_t* n_unitigs, uint64_t** left, uint64_t* n_left) {
    if (!solid || !seq_pool || !seq_off || !seeds || !n_unitigs || !left || !n_left) return rtk_fail(RTK_ERR_ARG, "rtk_index_unitigs: null argument");
    if (k < 3 || k > 31 || !(k & 1)) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_unitigs: one-word k-mers only (odd k <= 31)");
    if (rtk_device_count() <= device || device < 0) return rtk_fail(RTK_ERR_NO_DEVICE, "rtk_index_unitigs: no such HIP device (no CPU fallback)");
    if (n_solid >= (1ull << 32)) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_unitigs: more than 2^32 solid k-mers");
    *seq_pool = nullptr; *seq_off = nullptr; *seeds = nullptr; *left = nullptr; *n_unitigs = 0; *n_left = 0;
    const bool trace = getenv("RTK_INDEX_TRACE") != nullptr;
    try {
        rtk_check(hipSetDevice(device), "hipSetDevice");
        const auto t0 = std::chrono::steady_clock::now();
        auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
        const uint64_t n = n_solid, slots = n + n * 3 / 7 + 16;
        auto grid = [](uint64_t items) { const uint64_t b = (items + 255) / 256; return dim3(static_cast<unsigned>(b < 1 ? 1 : (b > 65536 ? 65536 : b))); };
        DevBuf d_solid, d_T, d_cnt, d_flag;
        d_solid.alloc(8 * n); d_T.alloc(16 * slots); d_cnt.alloc(8); d_flag.alloc(8);
        rtk_check(hipMemcpy(d_solid.p, solid, 8 * n, hipMemcpyHostToDevice), "hipMemcpy");
        rtk_check(hipMemset(d_cnt.p, 0, 8), "hipMemset"); rtk_check(hipMemset(d_flag.p, 0, 8), "hipMemset");
        const uint64_t* ds = static_cast<const uint64_t*>(d_solid.p); uint64_t* T = static_cast<uint64_t*>(d_T.p);
        uint32_t* d_too_long = static_cast<uint32_t*>(d_flag.p); uint32_t* d_clash = d_too_long + 1;
        hipLaunchKernelGGL(k_ut_fill, grid(slots), dim3(256), 0, 0, T, slots);
        hipLaunchKernelGGL(k_ut_insert, grid(n), dim3(256), 0, 0, ds, n, T, slots);
        hipLaunchKernelGGL(k_ut_edges, grid(n), dim3(256), 0, 0, ds, n, k, T, slots);
        rtk_check(hipGetLastError(), "kernel launch (unitig table)"); rtk_check(hipDeviceSynchronize(), "unitig table");
        const double t_table = since();
        // owners counted, then recorded
        hipLaunchKernelGGL(k_ut_chains, grid(n), dim3(256), 0, 0, ds, n, k, static_cast<const uint64_t*>(T), slots, static_cast<unsigned long long*>(d_cnt.p), 0ull, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr), static_cast<uint32_t*>(nullptr), d_too_long);
        rtk_check(hipGetLastError(), "kernel launch (k_ut_chains)"); rtk_check(hipDeviceSynchronize(), "k_ut_chains");
        unsigned long long n_ch = 0; rtk_check(hipMemcpy(&n_ch, d_cnt.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
        uint32_t fl[2] = {0, 0}; rtk_check(hipMemcpy(fl, d_flag.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
        if (fl[0]) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_unitigs: a unitig of more than 2^31 k-mers");
        if (n_ch >= (1ull << 32)) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_unitigs: more than 2^32 unitigs");
        DevBuf d_start, d_seed, d_seed2, d_lr, d_ord, d_ord2, d_off, d_tmp;
        d_start.alloc(8 * n_ch); d_seed.alloc(8 * n_ch); d_seed2.alloc(8 * n_ch); d_lr.alloc(4 * n_ch); d_ord.alloc(4 * n_ch); d_ord2.alloc(4 * n_ch); d_off.alloc(8 * (n_ch + 1));
        rtk_check(hipMemset(d_cnt.p, 0, 8), "hipMemset");
        hipLaunchKernelGGL(k_ut_chains, grid(n), dim3(256), 0, 0, ds, n, k, static_cast<const uint64_t*>(T), slots, static_cast<unsigned long long*>(d_cnt.p), static_cast<uint64_t>(n_ch), static_cast<uint64_t*>(d_start.p), static_cast<uint64_t*>(d_seed.p), static_cast<uint32_t*>(d_lr.p), d_too_long);
        rtk_check(hipGetLastError(), "kernel launch (k_ut_chains)"); rtk_check(hipDeviceSynchronize(), "k_ut_chains");
        const double t_chains = since();
        std::vector<uint64_t> h_off(n_ch + 1, 0), h_seed(n_ch);
        uint64_t total = 0;
        if (n_ch) {
            // the chains in the order of their seeds (one chain per seed: a k-mer lies on one chain)
            auto iota = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), Iota32());
            size_t tb = 0; rtk_check(rocprim::radix_sort_pairs(nullptr, tb, static_cast<uint64_t*>(d_seed.p), static_cast<uint64_t*>(d_seed2.p), iota, static_cast<uint32_t*>(d_ord.p), static_cast<size_t>(n_ch), 0, 2 * k), "rocprim::radix_sort_pairs");
            d_tmp.alloc(tb);
            rtk_check(rocprim::radix_sort_pairs(d_tmp.p, tb, static_cast<uint64_t*>(d_seed.p), static_cast<uint64_t*>(d_seed2.p), iota, static_cast<uint32_t*>(d_ord.p), static_cast<size_t>(n_ch), 0, 2 * k), "rocprim::radix_sort_pairs");
            ChainBases cb; cb.len_rev = static_cast<const uint32_t*>(d_lr.p); cb.order = static_cast<const uint32_t*>(d_ord.p); cb.k = k;
            auto lens = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), cb);
            size_t sb = 0; rtk_check(rocprim::exclusive_scan(nullptr, sb, lens, static_cast<uint64_t*>(d_off.p), 0ull, static_cast<size_t>(n_ch), rocprim::plus<uint64_t>()), "rocprim::exclusive_scan");
            d_tmp.alloc(sb);
            rtk_check(rocprim::exclusive_scan(d_tmp.p, sb, lens, static_cast<uint64_t*>(d_off.p), 0ull, static_cast<size_t>(n_ch), rocprim::plus<uint64_t>()), "rocprim::exclusive_scan");
            rtk_check(hipDeviceSynchronize(), "chain order");
            rtk_check(hipMemcpy(h_off.data(), d_off.p, 8 * n_ch, hipMemcpyDeviceToHost), "hipMemcpy");
            rtk_check(hipMemcpy(h_seed.data(), d_seed2.p, 8 * n_ch, hipMemcpyDeviceToHost), "hipMemcpy");
            uint32_t last_c = 0, last_lr = 0; rtk_check(hipMemcpy(&last_c, static_cast<uint32_t*>(d_ord.p) + (n_ch - 1), 4, hipMemcpyDeviceToHost), "hipMemcpy");
            rtk_check(hipMemcpy(&last_lr, static_cast<uint32_t*>(d_lr.p) + last_c, 4, hipMemcpyDeviceToHost), "hipMemcpy");
            total = h_off[n_ch - 1] + (last_lr & 0x7FFFFFFFu) + static_cast<uint64_t>(k - 1); h_off[n_ch] = total;
        }
        DevBuf d_pool; d_pool.alloc(total ? total : 8);
        if (n_ch) {
            hipLaunchKernelGGL(k_ut_write, grid(n_ch), dim3(256), 0, 0, static_cast<uint64_t>(n_ch), static_cast<const uint32_t*>(d_ord.p), static_cast<const uint64_t*>(d_start.p), static_cast<const uint32_t*>(d_lr.p), static_cast<const uint64_t*>(d_off.p), k, T, slots, static_cast<char*>(d_pool.p), d_clash);
            rtk_check(hipGetLastError(), "kernel launch (k_ut_write)"); rtk_check(hipDeviceSynchronize(), "k_ut_write");
        }
        rtk_check(hipMemcpy(fl, d_flag.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
        if (fl[1]) return rtk_fail(RTK_ERR_FORMAT, "rtk_index_unitigs: a k-mer ended up on two unitigs"); // (the tool then runs its plain construction)
        const double t_write = since();
        // the k-mers on no unitig written here, in sorted order
        DevBuf d_left, d_nleft; d_left.alloc(8 * (n ? n : 1)); d_nleft.alloc(8);
        Unclaimed un; un.solid = ds; un.T = T; un.slots = slots;
        { auto idx = rocprim::make_counting_iterator<uint64_t>(0); auto flags = rocprim::make_transform_iterator(idx, un);
          size_t sb = 0; rtk_check(rocprim::select(nullptr, sb, ds, flags, static_cast<uint64_t*>(d_left.p), static_cast<unsigned long long*>(d_nleft.p), static_cast<size_t>(n)), "rocprim::select");
          d_tmp.alloc(sb);
          rtk_check(rocprim::select(d_tmp.p, sb, ds, flags, static_cast<uint64_t*>(d_left.p), static_cast<unsigned long long*>(d_nleft.p), static_cast<size_t>(n)), "rocprim::select");
          rtk_check(hipDeviceSynchronize(), "left-over k-mers"); }
        unsigned long long nl = 0; rtk_check(hipMemcpy(&nl, d_nleft.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
        char* o_pool = static_cast<char*>(malloc(total ? total : 1)); uint64_t* o_off = static_cast<uint64_t*>(malloc(8 * (n_ch + 1))); uint64_t* o_seed = static_cast<uint64_t*>(malloc(8 * (n_ch ? n_ch : 1))); uint64_t* o_left = static_cast<uint64_t*>(malloc(8 * (nl ? nl : 1)));
        if (!o_pool || !o_off || !o_seed || !o_left) { free(o_pool); free(o_off); free(o_seed); free(o_left); return rtk_fail(RTK_ERR_IO, "rtk_index_unitigs: out of host memory"); }
        if (total) rtk_check(hipMemcpy(o_pool, d_pool.p, total, hipMemcpyDeviceToHost), "hipMemcpy");
        memcpy(o_off, h_off.data(), 8 * (n_ch + 1)); if (n_ch) memcpy(o_seed, h_seed.data(), 8 * n_ch);
        if (nl) rtk_check(hipMemcpy(o_left, d_left.p, 8 * nl, hipMemcpyDeviceToHost), "hipMemcpy");
        *seq_pool = o_pool; *seq_off = o_off; *seeds = o_seed; *n_unitigs = n_ch; *left = o_left; *n_left = nl;
        if (trace) fprintf(stderr, "rtk_index_unitigs: %llu unitigs, %llu bases, %llu k-mers left to the plain construction; table + edge bits %.2f s, chains %.2f s, order + sequences %.2f s, left-overs + copies %.2f s\n",
                           n_ch, static_cast<unsigned long long>(total), nl, t_table, t_chains - t_table, t_write - t_chains, since() - t_write);
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_index_unitigs: ") + e.what()); }
    return RTK_OK;
}


// Colours and coverage of an index build (addCoverage, src/Graph.cpp:1561-1985: every read k-mer mapped onto its unitig) on the device. The caller -- the index
// tool, which owns the numbering of the reads (a pair keeps one id: name changes, or pair numbers of a sampled source) -- opens a job on the unitigs, feeds its
// reads chunk by chunk from any number of threads, and gets back the distinct (unitig, read id) events in sorted order and the k-mer coverage of every unitig.
//   begin  unitig u = seq_pool[seq_off[u] .. seq_off[u + 1]) (characters); the pool is packed to 2 bits and the k-mer table built in HBM (rtk_graph_tables.hip)
//   chunk  chars: sequences separated by '\n' (n_chars characters); read r starts at starts[r] and has the id ids[r]. At most RTK_COLOUR_CHUNK (64 MB) per call.
//   end    *events: n_events words unitig << 32 | id, ascending, distinct; *cov: n_unitigs counts. Freed with rtk_free. The job is gone afterwards (also on error).
extern "C" int rtk_index_colour_begin(int device, int k, const char* seq_pool, const uint64_t* seq_off, uint64_t n_unitigs, void** job_out) {
    if (!seq_pool || !seq_off || !job_out || n_unitigs == 0) return rtk_fail(RTK_ERR_ARG, "rtk_index_colour_begin: null argument");
    if (k < 3 || k > 31 || !(k & 1)) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_colour_begin: one-word k-mers only (odd k <= 31)");
    if (rtk_device_count() <= device || device < 0) return rtk_fail(RTK_ERR_NO_DEVICE, "rtk_index_colour_begin: no such HIP device (no CPU fallback)");
    if (n_unitigs >= 0xFFFFFFFFull) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_colour_begin: more than 2^32 - 1 unitigs");
    *job_out = nullptr;
    std::unique_ptr<ColourJob> J(new ColourJob());
    try {
        rtk_check(hipSetDevice(device), "hipSetDevice");
        J->t0 = std::chrono::steady_clock::now();
        J->device = device; J->k = k; J->n_unitigs = static_cast<uint32_t>(n_unitigs);
        const uint64_t n_bases = seq_off[n_unitigs], n_kmers = n_bases - n_unitigs * static_cast<uint64_t>(k - 1), n_words = (n_bases + 31) / 32;
        { DevBuf d_pool; d_pool.alloc(n_bases); rtk_check(hipMemcpy(d_pool.p, seq_pool, n_bases, hipMemcpyHostToDevice), "hipMemcpy");
          J->useq.alloc(8 * (n_words + 2)); rtk_check(hipMemset(J->useq.p, 0, 8 * (n_words + 2)), "hipMemset");
          hipLaunchKernelGGL(k_col_pack, dim3(4096), dim3(256), 0, 0, static_cast<const char*>(d_pool.p), n_bases, static_cast<uint64_t*>(J->useq.p), n_words);
          rtk_check(hipGetLastError(), "kernel launch (k_col_pack)"); rtk_check(hipDeviceSynchronize(), "k_col_pack"); }
        J->uoff.alloc(8 * (n_unitigs + 1)); rtk_check(hipMemcpy(J->uoff.p, seq_off, 8 * (n_unitigs + 1), hipMemcpyHostToDevice), "hipMemcpy");
        void* ht = nullptr; rtk::device_kmer_table(static_cast<const uint64_t*>(J->useq.p), static_cast<const uint64_t*>(J->uoff.p), J->n_unitigs, n_bases, n_kmers, k, &ht, &J->slots);
        J->ht.p = ht;
        J->cov.alloc(8 * n_unitigs); rtk_check(hipMemset(J->cov.p, 0, 8 * n_unitigs), "hipMemset");
        J->top.alloc(8); rtk_check(hipMemset(J->top.p, 0, 8), "hipMemset");
        J->chunk_cap = 64ull << 20; J->reads_cap = J->chunk_cap / 16; // (a read of fewer than 15 characters per 16 bytes of chunk: the caller splits such chunks)
        for (int i = 0; i < 2; ++i) {
            J->d_chars[i].alloc(J->chunk_cap); J->d_starts[i].alloc(8 * J->reads_cap); J->d_ids[i].alloc(4 * J->reads_cap);
            J->h_chars[i].alloc(J->chunk_cap); J->h_starts[i].alloc(8 * J->reads_cap); J->h_ids[i].alloc(4 * J->reads_cap);
            rtk_check(hipStreamCreate(&J->st[i]), "hipStreamCreate");
        }
        size_t fr = 0, tot = 0; rtk_check(hipMemGetInfo(&fr, &tot), "hipMemGetInfo");
        J->cap = static_cast<uint64_t>(fr) / 10 * 6 / 16; // 60 % of what is left for the events and their sort buffer
        { const char* e = getenv("RTK_INDEX_EVENTS"); if (e) J->cap = strtoull(e, nullptr, 10); }
        if (J->cap < 1024) J->cap = 1024;
        J->events.alloc(8 * J->cap); J->alt.alloc(8 * J->cap);
        J->t_table = std::chrono::duration<double>(std::chrono::steady_clock::now() - J->t0).count();
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_index_colour_begin: ") + e.what()); }
    *job_out = J.release();
    return RTK_OK;
}

extern "C" int rtk_index_colour_chunk(void* job, const char* chars, uint64_t n_chars, const uint64_t* starts, const uint32_t* ids, uint32_t n_reads) {
    ColourJob* J = static_cast<ColourJob*>(job);
    if (!J || !chars || !starts || !ids) return rtk_fail(RTK_ERR_ARG, "rtk_index_colour_chunk: null argument");
    if (n_reads == 0 || n_chars == 0) return RTK_OK;
    if (n_chars > J->chunk_cap || n_reads > J->reads_cap) return rtk_fail(RTK_ERR_ARG, "rtk_index_colour_chunk: chunk larger than 64 MB of characters / 4 M reads");
    try {
        std::lock_guard<std::mutex> lk(J->m);
        rtk_check(hipSetDevice(J->device), "hipSetDevice");
        // room for this chunk's events (at most one per position): sort and keep the distinct ones when the buffer is half full
        if (J->chunks && ((J->chunks & 7u) == 0 || J->cap < (1ull << 30))) { // (looked at every 8th chunk: reading the counter waits for the device)
            rtk_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
            unsigned long long n = 0; rtk_check(hipMemcpy(&n, J->top.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
            if (n > J->cap / 2) J->compact();
        }
        const int sl = J->slot; J->slot ^= 1;
        rtk_check(hipStreamSynchronize(J->st[sl]), "hipStreamSynchronize");
        memcpy(J->h_chars[sl].p, chars, n_chars); memcpy(J->h_starts[sl].p, starts, 8ull * n_reads); memcpy(J->h_ids[sl].p, ids, 4ull * n_reads);
        rtk_check(hipMemcpyAsync(J->d_chars[sl].p, J->h_chars[sl].p, n_chars, hipMemcpyHostToDevice, J->st[sl]), "hipMemcpyAsync");
        rtk_check(hipMemcpyAsync(J->d_starts[sl].p, J->h_starts[sl].p, 8ull * n_reads, hipMemcpyHostToDevice, J->st[sl]), "hipMemcpyAsync");
        rtk_check(hipMemcpyAsync(J->d_ids[sl].p, J->h_ids[sl].p, 4ull * n_reads, hipMemcpyHostToDevice, J->st[sl]), "hipMemcpyAsync");
        hipLaunchKernelGGL(k_col_map, dim3(4096), dim3(256), 0, J->st[sl], static_cast<const char*>(J->d_chars[sl].p), n_chars, J->k, static_cast<const uint64_t*>(J->d_starts[sl].p), static_cast<const uint32_t*>(J->d_ids[sl].p), n_reads,
                           static_cast<const uint64_t*>(J->ht.p), J->slots, static_cast<unsigned long long*>(J->cov.p), static_cast<uint64_t*>(J->events.p), static_cast<unsigned long long*>(J->top.p), J->cap);
        rtk_check(hipGetLastError(), "kernel launch (k_col_map)");
        J->bases += n_chars; ++J->chunks;
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_index_colour_chunk: ") + e.what()); }
    return RTK_OK;
}

extern "C" int rtk_index_colour_end(void* job, uint64_t** events, uint64_t* n_events, uint64_t** cov) {
    std::unique_ptr<ColourJob> J(static_cast<ColourJob*>(job));
    if (!J) return rtk_fail(RTK_ERR_ARG, "rtk_index_colour_end: null job");
    if (!events || !n_events || !cov) return RTK_OK; // (abandoned job: released)
    *events = nullptr; *cov = nullptr; *n_events = 0;
    try {
        rtk_check(hipSetDevice(J->device), "hipSetDevice");
        J->compact();
        uint64_t* ev = static_cast<uint64_t*>(malloc(8 * (J->n_events ? J->n_events : 1))); uint64_t* cv = static_cast<uint64_t*>(malloc(8ull * J->n_unitigs));
        if (!ev || !cv) { free(ev); free(cv); return rtk_fail(RTK_ERR_IO, "rtk_index_colour_end: out of host memory"); }
        if (J->n_events) rtk_check(hipMemcpy(ev, J->events.p, 8 * J->n_events, hipMemcpyDeviceToHost), "hipMemcpy");
        rtk_check(hipMemcpy(cv, J->cov.p, 8ull * J->n_unitigs, hipMemcpyDeviceToHost), "hipMemcpy");
        *events = ev; *n_events = J->n_events; *cov = cv;
        if (getenv("RTK_INDEX_TRACE")) fprintf(stderr, "rtk_index_colour: %llu characters in %llu chunks -> %llu distinct (unitig, read) events (sorted and thinned out %llu times); table %.2f s, all %.2f s\n", static_cast<unsigned long long>(J->bases),
                                               static_cast<unsigned long long>(J->chunks), static_cast<unsigned long long>(J->n_events), static_cast<unsigned long long>(J->compactions), J->t_table, std::chrono::duration<double>(std::chrono::steady_clock::now() - J->t0).count());
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_index_colour_end: ") + e.what()); }
    return RTK_OK;
}
