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

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../../include/ratatosk_hip.h"
#include "../common/fastx.hpp"
#include "../common/kmer.hpp"
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
        const uint64_t est_kmers = total_bytes / 2 + (1u << 20); // FASTQ: half of the bytes are bases (gzip input: a multiple of it; the capacity test below catches that)
        uint64_t cap = static_cast<uint64_t>(fr) / 10 * 4 / 16; // 40 % of the free memory for keys + their sort buffer
        { const char* e = getenv("RTK_INDEX_CAP"); if (e) cap = strtoull(e, nullptr, 10); }
        if (cap < (1u << 20)) cap = 1u << 20;
        uint32_t n_part = static_cast<uint32_t>((est_kmers + cap - 1) / cap); if (n_part < 1) n_part = 1;
        const uint64_t chunk_bytes = 256ull << 20;
        std::vector<uint64_t> solid;
        for (bool done = false; !done;) {
            done = true; solid.clear();
            const uint64_t cap_p = n_part == 1 ? std::min<uint64_t>(cap, est_kmers + est_kmers / 8) : cap;
            DevBuf d_keys, d_alt, d_top, d_chars[2], d_sel, d_nsel;
            d_keys.alloc(8 * cap_p); d_alt.alloc(8 * cap_p); d_top.alloc(8); d_nsel.alloc(8);
            d_chars[0].alloc(chunk_bytes + (64u << 20)); d_chars[1].alloc(chunk_bytes + (64u << 20));
            PinBuf h_chars[2]; h_chars[0].alloc(chunk_bytes + (64u << 20)); h_chars[1].alloc(chunk_bytes + (64u << 20));
            hipStream_t st[2]; rtk_check(hipStreamCreate(&st[0]), "hipStreamCreate"); rtk_check(hipStreamCreate(&st[1]), "hipStreamCreate");
            for (uint32_t part = 0; part < n_part && done; ++part) {
                rtk_check(hipMemset(d_top.p, 0, 8), "hipMemset");
                int slot = 0; std::string err;
                auto sink = [&](const char* chars, size_t n) { // one chunk: pinned copy, H2D and the k-mer kernel on the slot's stream (the other slot's work overlaps the next parse)
                    for (size_t off = 0; off < n;) {
                        const size_t piece = std::min<size_t>(n - off, chunk_bytes + (64u << 20));
                        rtk_check(hipStreamSynchronize(st[slot]), "hipStreamSynchronize");
                        memcpy(h_chars[slot].p, chars + off, piece);
                        rtk_check(hipMemcpyAsync(d_chars[slot].p, h_chars[slot].p, piece, hipMemcpyHostToDevice, st[slot]), "hipMemcpyAsync");
                        hipLaunchKernelGGL(k_index_kmers, dim3(4096), dim3(256), 0, st[slot], static_cast<const char*>(d_chars[slot].p), static_cast<uint64_t>(piece), k, part, n_part,
                                           static_cast<uint64_t*>(d_keys.p), static_cast<unsigned long long*>(d_top.p), cap_p);
                        rtk_check(hipGetLastError(), "kernel launch (k_index_kmers)");
                        slot ^= 1; off += piece;
                    }
                };
                if (!for_each_sequence_chunk(fl, n_threads, chunk_bytes, sink, &err)) { (void)hipStreamDestroy(st[0]); (void)hipStreamDestroy(st[1]); return rtk_fail(RTK_ERR_IO, "rtk_index_count_kmers: " + err); }
                rtk_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
                unsigned long long n_keys = 0; rtk_check(hipMemcpy(&n_keys, d_top.p, 8, hipMemcpyDeviceToHost), "hipMemcpy");
                if (n_keys > cap_p) { // more k-mers than estimated: more partitions, again
                    n_part = static_cast<uint32_t>((n_keys * static_cast<uint64_t>(n_part) + cap - 1) / cap) + 1; done = false; break;
                }
                if (n_keys == 0) continue;
                // sort, then the first key of every run of >= min_count
                rocprim::double_buffer<uint64_t> db(static_cast<uint64_t*>(d_keys.p), static_cast<uint64_t*>(d_alt.p));
                size_t tb = 0; rtk_check(rocprim::radix_sort_keys(nullptr, tb, db, static_cast<size_t>(n_keys), 0, 2 * k), "rocprim::radix_sort_keys");
                DevBuf d_tmp; d_tmp.alloc(tb);
                rtk_check(rocprim::radix_sort_keys(d_tmp.p, tb, db, static_cast<size_t>(n_keys), 0, 2 * k), "rocprim::radix_sort_keys");
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
                const size_t old = solid.size(); solid.resize(old + n_sel);
                if (n_sel) rtk_check(hipMemcpy(solid.data() + old, other, 8ull * n_sel, hipMemcpyDeviceToHost), "hipMemcpy");
            }
            (void)hipStreamDestroy(st[0]); (void)hipStreamDestroy(st[1]);
        }
        if (n_part > 1) std::sort(solid.begin(), solid.end()); // (partitions are sorted each; a k-mer lives in one partition)
        uint64_t* out = static_cast<uint64_t*>(malloc(8 * (solid.size() ? solid.size() : 1)));
        if (!out) return rtk_fail(RTK_ERR_IO, "rtk_index_count_kmers: out of host memory");
        if (!solid.empty()) memcpy(out, solid.data(), 8 * solid.size());
        *solid_out = out; *n_solid = solid.size();
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_index_count_kmers: ") + e.what()); }
    return RTK_OK;
}
