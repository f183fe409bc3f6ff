// Reader/writer for the reference's unitig-data file (`*.rtsk`) records.
//
// Format (reference: src/Graph.cpp:786-801 writer, :722-784 reader; src/UnitigData.hpp:493-553;
// src/SharedPairID.cpp:445-478; src/PairID.cpp:1137-1215):
//   per unitig, no header, no count:
//     Kmer head                     16 bytes (2 x u64 LE, 2 bits/base, first base in MSBs of word 0)
//     u64  kmCov_cardBranches       bit63 branching, bit62 visit, bits31..61 unphased cov, bits0..30 phased cov
//     u64  shared_pids              bit8 short cycle, bits4..7 fw successor-base mask, bits0..3 bw mask
//     PairID global                 (the u64 0x1 when there is no global set)
//     PairID local
//     PairID ambiguity_ids          ids = (pos<<4)+iupacIdx
//     PairID hap_ids
//     u64  n_cycle_bytes  + bytes   NUL separated successor-base strings
//   PairID stream: one u64 w; (w&7)==1 inline bit-vector, ids = set bits of w>>3 (<61);
//     (w&7)==2 single id w>>3; (w&7)==3 -> w>>3 bytes of CRoaring *portable* serialisation follow;
//     (w&7)==0 -> Bifrost TinyBitmap::write payload follows (assumed layout [A8], see tinybitmap_read; inconsistent payloads are rejected).
#ifndef RTK_COMMON_RTSK_IO_HPP
#define RTK_COMMON_RTSK_IO_HPP

#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <streambuf>
#include <string>
#include <vector>

namespace rtk {

struct RtskRecord {
    uint64_t head[2];
    uint64_t kmcov;
    uint64_t shared;
    std::vector<uint32_t> global_ids, local_ids, ambiguity_ids, hap_ids;
    std::string cycles;
};

// ---- CRoaring portable format (format spec: RoaringFormatSpec; cookies 12346 / 12347) ----
inline void roaring_portable_decode(const unsigned char* buf, size_t n, std::vector<uint32_t>& out) {
    auto rd16 = [&](size_t off) -> uint32_t { if (off + 2 > n) throw std::runtime_error("rtsk: truncated roaring"); return buf[off] | (buf[off + 1] << 8); };
    auto rd32 = [&](size_t off) -> uint32_t { if (off + 4 > n) throw std::runtime_error("rtsk: truncated roaring"); uint32_t v; memcpy(&v, buf + off, 4); return v; };
    size_t off = 0;
    const uint32_t cookie = rd32(off); off += 4;
    uint32_t size;
    bool has_run = false;
    const unsigned char* run_bm = nullptr;
    if ((cookie & 0xFFFF) == 12347) {
        has_run = true;
        size = (cookie >> 16) + 1;
        run_bm = buf + off;
        off += (size + 7) / 8;
    } else if (cookie == 12346) {
        size = rd32(off); off += 4;
    } else throw std::runtime_error("rtsk: bad roaring cookie");
    std::vector<uint32_t> keys(size), cards(size);
    for (uint32_t i = 0; i < size; ++i) { keys[i] = rd16(off); cards[i] = rd16(off + 2) + 1; off += 4; }
    if (!has_run || size >= 4) off += 4ULL * size; // offset header
    for (uint32_t i = 0; i < size; ++i) {
        const uint32_t hi = keys[i] << 16;
        const bool is_run = has_run && ((run_bm[i / 8] >> (i % 8)) & 1);
        if (is_run) {
            const uint32_t nr = rd16(off); off += 2;
            for (uint32_t r = 0; r < nr; ++r) {
                const uint32_t s = rd16(off), l = rd16(off + 2); off += 4;
                for (uint32_t v = s; v <= s + l; ++v) out.push_back(hi | v);
            }
        } else if (cards[i] <= 4096) {
            for (uint32_t c = 0; c < cards[i]; ++c) { out.push_back(hi | rd16(off)); off += 2; }
        } else {
            if (off + 8192 > n) throw std::runtime_error("rtsk: truncated roaring bitset");
            for (uint32_t w = 0; w < 1024; ++w) {
                uint64_t x; memcpy(&x, buf + off + 8 * w, 8);
                while (x) { const int b = __builtin_ctzll(x); out.push_back(hi | (w * 64 + b)); x &= x - 1; }
            }
            off += 8192;
        }
    }
}

// ids must be sorted ascending & unique. Emits the no-run-container layout (cookie 12346).
inline void roaring_portable_encode(const std::vector<uint32_t>& ids, std::string& out) {
    std::vector<std::pair<uint32_t, std::pair<size_t, size_t> > > cont; // key -> [begin,end)
    for (size_t i = 0; i < ids.size();) {
        size_t j = i;
        while (j < ids.size() && (ids[j] >> 16) == (ids[i] >> 16)) ++j;
        cont.push_back(std::make_pair(ids[i] >> 16, std::make_pair(i, j)));
        i = j;
    }
    auto put16 = [&](uint32_t v) { out.push_back(static_cast<char>(v & 0xFF)); out.push_back(static_cast<char>((v >> 8) & 0xFF)); };
    auto put32 = [&](uint32_t v) { for (int b = 0; b < 4; ++b) out.push_back(static_cast<char>((v >> (8 * b)) & 0xFF)); };
    const size_t base = out.size();
    put32(12346); put32(static_cast<uint32_t>(cont.size()));
    for (size_t c = 0; c < cont.size(); ++c) { put16(cont[c].first); put16(static_cast<uint32_t>(cont[c].second.second - cont[c].second.first - 1)); }
    uint32_t off = static_cast<uint32_t>(8 + 8 * cont.size());
    for (size_t c = 0; c < cont.size(); ++c) {
        put32(off);
        const size_t card = cont[c].second.second - cont[c].second.first;
        off += (card <= 4096) ? static_cast<uint32_t>(2 * card) : 8192u;
    }
    for (size_t c = 0; c < cont.size(); ++c) {
        const size_t b = cont[c].second.first, e = cont[c].second.second;
        if (e - b <= 4096) { for (size_t i = b; i < e; ++i) put16(ids[i] & 0xFFFF); }
        else {
            uint64_t words[1024]; memset(words, 0, sizeof(words));
            for (size_t i = b; i < e; ++i) { const uint32_t v = ids[i] & 0xFFFF; words[v >> 6] |= 1ULL << (v & 63); }
            out.append(reinterpret_cast<const char*>(words), sizeof(words));
        }
    }
    (void)base;
}

// Bifrost TinyBitmap stream (PairID flag 0; reference: src/PairID.cpp:1158-1167 writes the flag word, then TinyBitmap::write).
// [A8] Layout as published in pmelsted/bifrost src/TinyBitmap.{hpp,cpp} -- NOT verifiable in this build (Bifrost absent, no reference-
// written .rtsk at hand): an array of uint16 words, word 0 = header (size_in_words << 3 | mode | bits), word 1 = cardinality (words
// in use for the list modes), word 2 = offset = the high 16 bits shared by every value, data from word 3:
//   mode 0 bitmap   bit b of word 3 + i  <=>  value (offset << 16) | (16 i + b)
//   mode 2 list     `cardinality` ascending low halves
//   mode 4 RLE list `cardinality` / 2 pairs (first, last), inclusive, ascending
// An empty TinyBitmap is the single word 0. Anything inconsistent with this layout is rejected (never guessed at).
inline void tinybitmap_read(std::istream& in, std::vector<uint32_t>& ids) {
    uint16_t header = 0;
    in.read(reinterpret_cast<char*>(&header), 2);
    if (!in.good()) throw std::runtime_error("rtsk: truncated TinyBitmap header");
    const uint32_t sz = header >> 3, mode = header & 0x6u;
    if (header & 1u) throw std::runtime_error("rtsk: TinyBitmap header has bit 0 set: not the layout assumed in [A8], refusing to guess");
    { // [A8] cannot be checked against a Bifrost build here ("fail loudly until verified"): such a payload is REFUSED unless the caller opts in
        const char* allow = getenv("RTK_ALLOW_TINYBITMAP");
        if (!(allow && allow[0] == '1')) throw std::runtime_error("rtsk: the index holds a Bifrost TinyBitmap colour set (PairID flag 0); its layout is assumption [A8] (oracle/oracle_graph.hpp), not verified against a Bifrost-written index: run `Ratatosk correct` with --allow-tinybitmap (library callers: set RTK_ALLOW_TINYBITMAP=1) to decode it under that assumption");
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "rtsk: note: RTK_ALLOW_TINYBITMAP=1: decoding Bifrost TinyBitmap colour sets with the layout assumed in [A8]; it has not been verified against a Bifrost-written index\n"); }
    }
    if (sz == 0) return;
    if (sz < 3 || sz > 4096 || (mode != 0 && mode != 2 && mode != 4)) throw std::runtime_error("rtsk: TinyBitmap header does not match the assumed Bifrost layout [A8]");
    std::vector<uint16_t> w(sz); w[0] = header;
    in.read(reinterpret_cast<char*>(&w[1]), static_cast<std::streamsize>(2 * (sz - 1)));
    if (!in.good()) throw std::runtime_error("rtsk: truncated TinyBitmap payload");
    const uint32_t card = w[1], hi = static_cast<uint32_t>(w[2]) << 16;
    if (mode == 0) {
        for (uint32_t i = 3; i < sz; ++i) for (uint32_t b = 0; b < 16; ++b) if ((w[i] >> b) & 1u) ids.push_back(hi | (16u * (i - 3) + b));
        if (ids.size() != card) throw std::runtime_error("rtsk: TinyBitmap bitmap cardinality mismatch [A8]");
    } else if (mode == 2) {
        if (3 + card > sz) throw std::runtime_error("rtsk: TinyBitmap list longer than its block [A8]");
        for (uint32_t i = 0; i < card; ++i) { if (i && w[3 + i] <= w[2 + i]) throw std::runtime_error("rtsk: TinyBitmap list not ascending [A8]"); ids.push_back(hi | w[3 + i]); }
    } else {
        if ((card & 1u) || 3 + card > sz) throw std::runtime_error("rtsk: TinyBitmap run list malformed [A8]");
        for (uint32_t i = 0; i < card; i += 2) {
            const uint32_t a = w[3 + i], b = w[4 + i];
            if (b < a || (i && a <= w[2 + i])) throw std::runtime_error("rtsk: TinyBitmap runs not ascending [A8]");
            for (uint32_t v = a; v <= b; ++v) ids.push_back(hi | v);
        }
    }
}

inline void pairid_read(std::istream& in, std::vector<uint32_t>& ids) {
    ids.clear();
    uint64_t w = 0;
    in.read(reinterpret_cast<char*>(&w), 8);
    if (!in.good()) throw std::runtime_error("rtsk: truncated PairID word");
    const uint64_t flag = w & 7ULL;
    if (flag == 1) {
        uint64_t bits = w >> 3;
        while (bits) { ids.push_back(static_cast<uint32_t>(__builtin_ctzll(bits))); bits &= bits - 1; }
    } else if (flag == 2) {
        ids.push_back(static_cast<uint32_t>(w >> 3));
    } else if (flag == 3) {
        const size_t n = static_cast<size_t>(static_cast<uint32_t>(w >> 3));
        std::vector<unsigned char> buf(n);
        in.read(reinterpret_cast<char*>(buf.data()), static_cast<std::streamsize>(n));
        if (!in.good() && n) throw std::runtime_error("rtsk: truncated roaring payload");
        roaring_portable_decode(buf.data(), n, ids);
    } else if (flag == 0) {
        if (w != 0) throw std::runtime_error("rtsk: PairID flag 0 with a non-zero word (src/PairID.cpp:1158-1167 writes the bare flag before the TinyBitmap)");
        tinybitmap_read(in, ids);
    } else throw std::runtime_error("rtsk: unknown PairID flag");
}

inline void pairid_write(std::ostream& out, const std::vector<uint32_t>& ids) {
    uint64_t w;
    if (ids.empty()) { w = 1; out.write(reinterpret_cast<const char*>(&w), 8); return; }
    if (ids.back() < 61) {
        uint64_t bits = 0;
        for (size_t i = 0; i < ids.size(); ++i) bits |= 1ULL << ids[i];
        w = (bits << 3) | 1ULL; out.write(reinterpret_cast<const char*>(&w), 8); return;
    }
    if (ids.size() == 1) { w = (static_cast<uint64_t>(ids[0]) << 3) | 2ULL; out.write(reinterpret_cast<const char*>(&w), 8); return; }
    std::string payload;
    roaring_portable_encode(ids, payload);
    w = (static_cast<uint64_t>(payload.size()) << 3) | 3ULL;
    out.write(reinterpret_cast<const char*>(&w), 8);
    out.write(payload.data(), static_cast<std::streamsize>(payload.size()));
}

inline bool rtsk_read_record(std::istream& in, RtskRecord& r) {
    in.read(reinterpret_cast<char*>(r.head), 16);
    if (in.gcount() == 0) return false; // clean EOF
    if (in.gcount() != 16) throw std::runtime_error("rtsk: truncated head k-mer");
    in.read(reinterpret_cast<char*>(&r.kmcov), 8);
    in.read(reinterpret_cast<char*>(&r.shared), 8);
    if (!in.good()) throw std::runtime_error("rtsk: truncated record");
    pairid_read(in, r.global_ids);
    pairid_read(in, r.local_ids);
    pairid_read(in, r.ambiguity_ids);
    pairid_read(in, r.hap_ids);
    uint64_t n = 0;
    in.read(reinterpret_cast<char*>(&n), 8);
    if (!in.good()) throw std::runtime_error("rtsk: truncated cycle length");
    r.cycles.assign(static_cast<size_t>(n), '\0');
    if (n) { in.read(&r.cycles[0], static_cast<std::streamsize>(n)); if (!in.good()) throw std::runtime_error("rtsk: truncated cycles"); }
    return true;
}

// Length in bytes of the record that starts at p (n bytes left in the file), without decoding its id streams: the loader cuts the file into records with
// it and decodes them on all threads. 0 at a clean end (n == 0); throws on a record that is cut short.
inline size_t rtsk_record_bytes(const unsigned char* p, size_t n) {
    if (n == 0) return 0;
    size_t at = 0;
    auto need = [&](size_t bytes) { if (at + bytes > n) throw std::runtime_error("rtsk: truncated record"); };
    need(32); at += 32; // head k-mer, coverage word, shared word
    for (int s = 0; s < 4; ++s) { // global, local, ambiguity, haplotype ids
        need(8); uint64_t w; memcpy(&w, p + at, 8); at += 8;
        const uint64_t flag = w & 7ULL;
        if (flag == 3) { const size_t len = static_cast<size_t>(static_cast<uint32_t>(w >> 3)); need(len); at += len; }
        else if (flag == 0) { // (the checks of tinybitmap_read on the header, so that a stream that is no TinyBitmap is named here instead of mis-cutting the file)
            if (w != 0) throw std::runtime_error("rtsk: PairID flag 0 with a non-zero word (src/PairID.cpp:1158-1167 writes the bare flag before the TinyBitmap)");
            need(2); uint16_t header; memcpy(&header, p + at, 2); const size_t sz = header >> 3; const uint32_t mode = header & 0x6u;
            if (header & 1u) throw std::runtime_error("rtsk: TinyBitmap header has bit 0 set: not the layout assumed in [A8], refusing to guess");
            if (sz != 0 && (sz < 3 || sz > 4096 || (mode != 0 && mode != 2 && mode != 4))) throw std::runtime_error("rtsk: TinyBitmap header does not match the assumed Bifrost layout [A8]");
            const size_t len = sz == 0 ? 2 : 2 * sz; if (at + len > n) throw std::runtime_error("rtsk: truncated TinyBitmap payload"); at += len; }
        else if (flag != 1 && flag != 2) throw std::runtime_error("rtsk: unknown PairID flag");
    }
    need(8); uint64_t nc; memcpy(&nc, p + at, 8); at += 8;
    if (nc > n - at) throw std::runtime_error("rtsk: truncated cycles");
    return at + static_cast<size_t>(nc);
}

// an istream over a stretch of memory (the records of a file that was read in one piece)
struct MemStreamBuf : std::streambuf { MemStreamBuf() {} MemStreamBuf(const char* b, size_t n) { reset(b, n); } void reset(const char* b, size_t n) { char* p = const_cast<char*>(b); setg(p, p, p + n); } };

inline void rtsk_write_record(std::ostream& out, const RtskRecord& r) {
    out.write(reinterpret_cast<const char*>(r.head), 16);
    out.write(reinterpret_cast<const char*>(&r.kmcov), 8);
    out.write(reinterpret_cast<const char*>(&r.shared), 8);
    pairid_write(out, r.global_ids);
    pairid_write(out, r.local_ids);
    pairid_write(out, r.ambiguity_ids);
    pairid_write(out, r.hap_ids);
    const uint64_t n = r.cycles.size();
    out.write(reinterpret_cast<const char*>(&n), 8);
    if (n) out.write(r.cycles.data(), static_cast<std::streamsize>(n));
}

// On-disk Kmer (2 x u64; MAX_KMER_SIZE=64 build of the reference, CMakeLists.txt:6):
// base i at bits [2*(31 - i%32), +1] of word i/32.
inline void disk_kmer_from_string(const char* s, int k, uint64_t w[2]) {
    w[0] = w[1] = 0;
    for (int i = 0; i < k; ++i) {
        uint64_t b = 0;
        switch (s[i]) { case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break; case 'T': b = 3; break; default: throw std::runtime_error("rtsk: non-ACGT in head k-mer"); }
        w[i / 32] |= b << (2 * (31 - (i % 32)));
    }
}

inline std::string disk_kmer_to_string(const uint64_t w[2], int k) {
    std::string s(k, 'A');
    for (int i = 0; i < k; ++i) s[i] = "ACGT"[(w[i / 32] >> (2 * (31 - (i % 32)))) & 3];
    return s;
}

} // namespace rtk

#endif
