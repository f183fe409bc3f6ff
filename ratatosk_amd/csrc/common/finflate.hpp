// Inflate of ONE gzip member, written for the host reader (common/mgzip.hpp): input is the mapped file (contiguous, so the decoder never
// suspends for input), output goes into the caller's chunks, each of which starts with the 32 KB of history before it.
//
// Why not zlib: zlib 1.2.11's inflate delivers 0.2-0.35 GB/s of FASTQ text per thread, and a gzip file that is ONE member cannot be
// spread over threads, so a single `reads.fastq.gz` caps the whole run an order of magnitude below what one GPU corrects. This decoder
// does what the fast inflate implementations do: a 64-bit bit buffer refilled once per symbol pair without branches, two-level decode
// tables whose entries carry base value, extra-bit count and code length in one word, literals decoded in runs, matches copied eight bytes
// at a time. RFC 1951 (deflate) and RFC 1952 (gzip) are the specification; nothing here is specific to Ratatosk. Every member's CRC-32
// and length are checked at its end (carry-less-multiply CRC when the CPU has it, verified against zlib's at first use), so a decoding
// error of any kind - a damaged file, or a defect of this decoder - ends the run with an error, never with wrong reads. RTK_ZLIB_INFLATE=1
// switches back to zlib.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace rtk {

// ---- CRC-32 (gzip polynomial) -----------------------------------------------------------------------------------------------------
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul_(uint32_t crc, const unsigned char* buf, size_t len) { // len >= 64, multiple of 16
    // folding by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"), bit-reflected form
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00)); x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10));
    x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20)); x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128(static_cast<int>(crc)));
    x0 = k1k2; buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00)); y6 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10));
        y7 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20)); y8 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6); x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = k3k4;
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf));
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    x0 = poly;
    x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    return static_cast<uint32_t>(_mm_extract_epi32(x1, 1));
}
#endif

// crc32(crc, buf, len) with zlib's semantics
inline uint32_t fast_crc32(uint32_t crc, const unsigned char* buf, size_t len) {
#if defined(__x86_64__)
    static const int usable = []() -> int { // the CPU has the instructions AND the routine agrees with zlib on a test pattern
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return 0;
        unsigned char t[1024 + 37]; uint32_t s = 12345; for (size_t i = 0; i < sizeof(t); ++i) { s = s * 1103515245u + 12345u; t[i] = static_cast<unsigned char>(s >> 16); }
        for (size_t n = 64; n <= 1024; n += 16 * 7) {
            const uint32_t want = static_cast<uint32_t>(crc32(0x1234u, t + 5, static_cast<uInt>(n)));
            if ((~crc32_clmul_(~0x1234u, t + 5, n)) != want) return 0;
        }
        return 1;
    }();
    if (usable && len >= 64) {
        const size_t n = len & ~static_cast<size_t>(15);
        crc = ~crc32_clmul_(~crc, buf, n);
        buf += n; len -= n;
    }
#endif
    while (len) { const uInt n = len > (1u << 30) ? (1u << 30) : static_cast<uInt>(len); crc = static_cast<uint32_t>(crc32(crc, buf, n)); buf += n; len -= n; }
    return crc;
}

// ---- inflate ----------------------------------------------------------------------------------------------------------------------
class FastInflate {
public:
    enum { HIST = 32768, SLACK = 320 }; // an output span needs SLACK writable bytes behind its end (matches are copied in 8-byte steps)

    // [in, in_end): the member starts at `in` (gzip header); in_end = end of the FILE (the member's own end is found by decoding)
    bool begin(const unsigned char* in, const unsigned char* in_end) {
        in_ = in; end_ = in_end; bitbuf_ = 0; bitcnt_ = 0; state_ = ST_HEADER; final_ = false; stored_left_ = 0; overrun_ = 0; total_out_ = 0; crc_ = 0; err_ = false; member_end_ = nullptr;
        // RFC 1952 header
        if (end_ - in_ < 18 || in_[0] != 0x1f || in_[1] != 0x8b || in_[2] != 8 || (in_[3] & 0xE0)) return fail_();
        const unsigned flg = in_[3];
        const unsigned char* p = in_ + 10;
        if (flg & 4) { if (end_ - p < 2) return fail_(); const size_t xl = p[0] | (p[1] << 8); p += 2; if (static_cast<size_t>(end_ - p) < xl) return fail_(); p += xl; }
        if (flg & 8) { while (p < end_ && *p) ++p; if (p >= end_) return fail_(); ++p; }
        if (flg & 16) { while (p < end_ && *p) ++p; if (p >= end_) return fail_(); ++p; }
        if (flg & 2) { if (end_ - p < 2) return fail_(); p += 2; }
        in_ = p; state_ = ST_BLOCK_HEADER;
        return true;
    }
    bool failed() const { return err_; }
    bool done() const { return state_ == ST_DONE; }
    const unsigned char* member_end() const { return member_end_; } // valid when done()

    // Decodes into [out, out + cap), with `hist` bytes of earlier output readable right before `out` (hist >= min(32768, bytes produced so
    // far)). Returns the bytes written: 0 with done() or failed() set, otherwise > 0 and possibly up to SLACK MORE than cap (all of them
    // output; the span must have SLACK writable bytes behind it).
    size_t decode(unsigned char* out, size_t cap, size_t hist) {
        if (err_ || state_ == ST_DONE) return 0;
        unsigned char* o = out; unsigned char* const o_end = out + cap;
        unsigned char* summed = out; // [out, summed) is in crc_ / total_out_ already
        const unsigned char* const o_min = out - hist;
        unsigned char* const o_stop = cap > 64 ? o_end - 64 : out; // the code loop checks the span once per refill and may run ~300 bytes past it: into SLACK
        while (o < o_stop && state_ != ST_DONE && !err_) {
            if (state_ == ST_BLOCK_HEADER) {
                refill_(); if (err_) break;
                final_ = (bitbuf_ & 1) != 0; const unsigned type = static_cast<unsigned>((bitbuf_ >> 1) & 3); drop_(3);
                if (type == 0) {
                    if (!to_byte_boundary_() || end_ - in_ < 4) { fail_(); break; }
                    const unsigned len = in_[0] | (in_[1] << 8), nlen = in_[2] | (in_[3] << 8);
                    if ((len ^ 0xFFFFu) != nlen) { fail_(); break; }
                    in_ += 4; stored_left_ = len; state_ = len ? ST_STORED : (final_ ? ST_TRAILER : ST_BLOCK_HEADER);
                } else if (type == 1) { build_fixed_(); state_ = ST_CODES; }
                else if (type == 2) { if (!read_dynamic_() || err_) { fail_(); break; } state_ = ST_CODES; }
                else { fail_(); break; }
            } else if (state_ == ST_STORED) {
                size_t n = stored_left_; if (n > static_cast<size_t>(o_end - o)) n = static_cast<size_t>(o_end - o); // (o < o_stop < o_end here)
                if (static_cast<size_t>(end_ - in_) < n) { fail_(); break; }
                memcpy(o, in_, n); o += n; in_ += n; stored_left_ -= static_cast<uint32_t>(n);
                if (stored_left_ == 0) state_ = final_ ? ST_TRAILER : ST_BLOCK_HEADER;
            } else if (state_ == ST_CODES) {
                o = codes_(o, o_stop, o_min);
            } else if (state_ == ST_TRAILER) {
                if (!to_byte_boundary_() || end_ - in_ < 8) { fail_(); break; }
                crc_ = fast_crc32(crc_, summed, static_cast<size_t>(o - summed)); total_out_ += static_cast<uint64_t>(o - summed); summed = o;
                const uint32_t want_crc = in_[0] | (in_[1] << 8) | (in_[2] << 16) | (static_cast<uint32_t>(in_[3]) << 24);
                const uint32_t want_len = in_[4] | (in_[5] << 8) | (in_[6] << 16) | (static_cast<uint32_t>(in_[7]) << 24);
                if (want_crc != crc_ || want_len != static_cast<uint32_t>(total_out_)) { fail_(); break; }
                member_end_ = in_ + 8; state_ = ST_DONE;
            } else { fail_(); break; }
        }
        if (err_) return 0;
        if (o != summed) { crc_ = fast_crc32(crc_, summed, static_cast<size_t>(o - summed)); total_out_ += static_cast<uint64_t>(o - summed); }
        return static_cast<size_t>(o - out);
    }

private:
    enum State { ST_HEADER, ST_BLOCK_HEADER, ST_STORED, ST_CODES, ST_TRAILER, ST_DONE };
    enum { LIT_PB = 11, DIST_PB = 8, PRE_PB = 7 };
    // table entry: bits 0-3 code length consumed at this level (primary: full length, or PB for a pointer), bits 4-7 kind, bits 8-15 extra bits /
    // sub-table index bits, bits 16-31 value (literal, length base, distance base, sub-table offset)
    // (bit 7 = literal, bits 7 + 6 = TWO literals in one entry of pair_: value = first | second << 8, length = both codes)
    enum { K_LIT = 0x80, K_PAIR = 0xC0, K_BASE = 0x10, K_EOB = 0x20, K_SUB = 0x30, K_BAD = 0 };

    bool fail_() { err_ = true; return false; }
    // drops the bits up to the next byte boundary and hands the whole bytes still in the bit buffer back to the input; false when bits were used
    // that lie behind the end of the file
    bool to_byte_boundary_() {
        drop_(bitcnt_ & 7);
        const unsigned back = bitcnt_ >> 3;
        if (back < overrun_) return false;
        in_ -= back - overrun_; overrun_ = 0; bitbuf_ = 0; bitcnt_ = 0;
        return true;
    }

    void refill_() {
        if (end_ - in_ >= 8) {
            uint64_t w; memcpy(&w, in_, 8);
            bitbuf_ |= w << bitcnt_;
            in_ += (63 - bitcnt_) >> 3;
            bitcnt_ |= 56;
        } else {
            while (bitcnt_ <= 56) { // behind the end of the file the stream reads as zeros; a member that really runs off the end fails its trailer checks
                const uint64_t b = in_ < end_ ? *in_ : 0; if (in_ < end_) ++in_; else ++overrun_;
                bitbuf_ |= b << bitcnt_; bitcnt_ += 8;
                if (overrun_ > 64) { err_ = true; return; }
            }
        }
    }
    void drop_(unsigned n) { bitbuf_ >>= n; bitcnt_ -= n; }

    // canonical Huffman decode table, two levels. lens[0..n): code lengths (0 = unused). kind_of(sym, &value, &extra) describes the symbols.
    template <class Describe>
    bool build_(const uint8_t* lens, unsigned n, unsigned pb, uint32_t* table, unsigned table_cap, bool allow_incomplete, Describe describe) {
        unsigned count[16]; memset(count, 0, sizeof(count));
        for (unsigned i = 0; i < n; ++i) ++count[lens[i]];
        if (count[0] == n) { // no codes at all: every lookup fails (a block may legally have no distance codes when it holds literals only)
            for (unsigned i = 0; i < (1u << pb); ++i) table[i] = K_BAD;
            return allow_incomplete;
        }
        unsigned maxlen = 15; while (maxlen > 1 && count[maxlen] == 0) --maxlen;
        int left = 1;
        for (unsigned l = 1; l <= 15; ++l) { left <<= 1; left -= static_cast<int>(count[l]); if (left < 0) return false; } // over-subscribed
        if (left > 0 && !(allow_incomplete && maxlen == 1)) return false; // incomplete: only the one-code distance tree may be (zlib's rule)
        unsigned next[16]; { unsigned code = 0; count[0] = 0; for (unsigned l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; } }
        for (unsigned i = 0; i < (1u << pb); ++i) table[i] = K_BAD;
        // sub-table sizes: the longest code behind every primary prefix
        unsigned top = 1u << pb;
        if (maxlen > pb) {
            uint8_t need[1 << LIT_PB]; memset(need, 0, 1u << pb);
            unsigned nx[16]; memcpy(nx, next, sizeof(nx));
            for (unsigned s = 0; s < n; ++s) { const unsigned l = lens[s]; if (!l) continue; const unsigned c = nx[l]++; if (l > pb) { const unsigned r = rev_(c, l) & ((1u << pb) - 1); if (l - pb > need[r]) need[r] = static_cast<uint8_t>(l - pb); } }
            for (unsigned r = 0; r < (1u << pb); ++r) if (need[r]) {
                if (top + (1u << need[r]) > table_cap) return false;
                table[r] = (top << 16) | (static_cast<uint32_t>(need[r]) << 8) | K_SUB | pb;
                for (unsigned i = 0; i < (1u << need[r]); ++i) table[top + i] = K_BAD;
                top += 1u << need[r];
            }
        }
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s]; if (!l) continue;
            const unsigned c = rev_(next[l]++, l);
            uint32_t value = 0, extra = 0, kind = 0; describe(s, &value, &extra, &kind);
            if (kind == K_BAD) continue; // a symbol that must not occur (litlen 286/287, dist 30/31): its codes stay invalid
            if (l <= pb) { const uint32_t e = (value << 16) | (extra << 8) | kind | l; for (unsigned i = c; i < (1u << pb); i += 1u << l) table[i] = e; }
            else {
                const uint32_t ptr = table[c & ((1u << pb) - 1)];
                const unsigned sub = ptr >> 16, sb = (ptr >> 8) & 0xFF;
                const uint32_t e = (value << 16) | (extra << 8) | kind | (l - pb);
                for (unsigned i = c >> pb; i < (1u << sb); i += 1u << (l - pb)) table[sub + i] = e;
            }
        }
        return true;
    }
    static unsigned rev_(unsigned c, unsigned l) { unsigned r = 0; for (unsigned i = 0; i < l; ++i) { r = (r << 1) | (c & 1); c >>= 1; } return r; }

    static void describe_litlen_(unsigned s, uint32_t* value, uint32_t* extra, uint32_t* kind) {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t ext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) { *value = s; *extra = 0; *kind = K_LIT; }
        else if (s == 256) { *value = 0; *extra = 0; *kind = K_EOB; }
        else if (s < 286) { *value = base[s - 257]; *extra = ext[s - 257]; *kind = K_BASE; }
        else *kind = K_BAD;
    }
    static void describe_dist_(unsigned s, uint32_t* value, uint32_t* extra, uint32_t* kind) {
        static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t ext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (s < 30) { *value = base[s]; *extra = ext[s]; *kind = K_BASE; } else *kind = K_BAD;
    }
    static void describe_pre_(unsigned s, uint32_t* value, uint32_t* extra, uint32_t* kind) { *value = s; *extra = 0; *kind = K_LIT; }

    void build_fixed_() {
        uint8_t l[288 + 32];
        for (unsigned i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)); // RFC 1951 3.2.6
        build_(l, 288, LIT_PB, lit_, sizeof(lit_) / 4, false, describe_litlen_); build_pairs_();
        for (unsigned i = 0; i < 32; ++i) l[i] = 5;
        build_(l, 32, DIST_PB, dist_, sizeof(dist_) / 4, false, describe_dist_);
    }

    // FASTQ text is mostly literals with codes of 2-4 bits: the primary index of 11 bits usually holds two whole codes, and decoding both
    // with one lookup halves the length of the dependent lookup -> shift -> lookup chain that bounds the literal rate
    void build_pairs_() {
        const unsigned mask = (1u << LIT_PB) - 1;
        for (unsigned i = 0; i <= mask; ++i) {
            const uint32_t e1 = lit_[i];
            uint32_t r = e1;
            if ((e1 & 0xF0) == K_LIT) {
                const unsigned l1 = e1 & 15;
                const uint32_t e2 = lit_[(i >> l1) & mask]; // the bits above 11 - l1 read as zeros: e2 is the right entry exactly when its code is no longer than what is left
                if ((e2 & 0xF0) == K_LIT && (e2 & 15) + l1 <= LIT_PB) r = ((e2 >> 16) << 24) | (((e1 >> 16) & 0xFF) << 16) | K_PAIR | (l1 + (e2 & 15));
            }
            pair_[i] = r;
        }
    }

    bool read_dynamic_() {
        refill_();
        const unsigned hlit = static_cast<unsigned>(bitbuf_ & 31) + 257, hdist = static_cast<unsigned>((bitbuf_ >> 5) & 31) + 1, hclen = static_cast<unsigned>((bitbuf_ >> 10) & 15) + 4; drop_(14);
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t pl[19]; memset(pl, 0, sizeof(pl));
        for (unsigned i = 0; i < hclen; ++i) { if (bitcnt_ < 3) refill_(); pl[order[i]] = static_cast<uint8_t>(bitbuf_ & 7); drop_(3); }
        uint32_t pre[1 << PRE_PB];
        if (!build_(pl, 19, PRE_PB, pre, 1u << PRE_PB, false, describe_pre_)) return false;
        uint8_t lens[286 + 30 + 140]; unsigned i = 0; const unsigned total = hlit + hdist;
        while (i < total) {
            refill_(); if (err_) return false;
            const uint32_t e = pre[bitbuf_ & ((1u << PRE_PB) - 1)];
            if ((e & 0xF0) == K_BAD) return false;
            drop_(e & 15);
            const unsigned s = e >> 16;
            if (s < 16) lens[i++] = static_cast<uint8_t>(s);
            else {
                unsigned rep, v = 0;
                if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + static_cast<unsigned>(bitbuf_ & 3); drop_(2); }
                else if (s == 17) { rep = 3 + static_cast<unsigned>(bitbuf_ & 7); drop_(3); }
                else { rep = 11 + static_cast<unsigned>(bitbuf_ & 127); drop_(7); }
                if (i + rep > total) return false;
                while (rep--) lens[i++] = static_cast<uint8_t>(v);
            }
        }
        if (lens[256] == 0) return false; // no end-of-block code
        if (!build_(lens, hlit, LIT_PB, lit_, sizeof(lit_) / 4, true, describe_litlen_)) return false; // (zlib's rule: incomplete only with a single 1-bit code)
        build_pairs_();
        return build_(lens + hlit, hdist, DIST_PB, dist_, sizeof(dist_) / 4, true, describe_dist_);
    }

    // the symbols of a block until its end or until `o` reaches `o_stop` (checked once per refill: up to ~300 bytes are written behind it)
    unsigned char* codes_(unsigned char* o, unsigned char* o_stop, const unsigned char* o_min) {
        uint64_t bb = bitbuf_; unsigned bc = bitcnt_; const unsigned char* in = in_;
        const uint32_t* const lit = lit_; const uint32_t* const dist = dist_; const uint32_t* const pair = pair_;
        const unsigned char* const in_fast = end_ - 8;
#define RTK_FI_REFILL() do { if (in <= in_fast) { uint64_t w_; memcpy(&w_, in, 8); bb |= w_ << bc; in += (63 - bc) >> 3; bc |= 56; } \
                             else { bitbuf_ = bb; bitcnt_ = bc; in_ = in; refill_(); bb = bitbuf_; bc = bitcnt_; in = in_; } } while (0)
        RTK_FI_REFILL();
        uint32_t e = pair[bb & ((1u << LIT_PB) - 1)]; // the entry of the NEXT symbol is always looked up ahead: its load runs beside the stores of the current one
        while (o < o_stop && !err_) {
            if (e & 0x80) { // one or two literals straight from the primary table
                bb >>= (e & 15); bc -= (e & 15);
                const uint16_t two = static_cast<uint16_t>(e >> 16); memcpy(o, &two, 2); o += 1 + ((e >> 6) & 1); // (the second byte is overwritten when the entry holds one)
                if (bc < 15) { RTK_FI_REFILL(); } // a whole code for the next lookup
                e = pair[bb & ((1u << LIT_PB) - 1)];
                continue;
            }
            if (bc < 48) { RTK_FI_REFILL(); } // a length / distance pair takes up to 15 + 5 + 15 + 13 bits (the refill leaves the bits `e` was read from alone)
            if ((e & 0xF0) == K_SUB) {
                bb >>= LIT_PB; bc -= LIT_PB; e = lit[(e >> 16) + (bb & ((1u << ((e >> 8) & 0xFF)) - 1))];
                if ((e & 0xF0) == K_LIT) { bb >>= (e & 15); bc -= (e & 15); *o++ = static_cast<unsigned char>(e >> 16); if (bc < 15) { RTK_FI_REFILL(); } e = pair[bb & ((1u << LIT_PB) - 1)]; continue; }
            }
            bb >>= (e & 15); bc -= (e & 15);
            if ((e & 0xF0) == K_BASE) {
                unsigned len = e >> 16; const unsigned xb = (e >> 8) & 0xFF;
                len += static_cast<unsigned>(bb & ((1u << xb) - 1)); bb >>= xb; bc -= xb;
                uint32_t d = dist[bb & ((1u << DIST_PB) - 1)];
                if ((d & 0xF0) == K_SUB) { bb >>= DIST_PB; bc -= DIST_PB; d = dist[(d >> 16) + (bb & ((1u << ((d >> 8) & 0xFF)) - 1))]; }
                if ((d & 0xF0) != K_BASE) { err_ = true; break; }
                bb >>= (d & 15); bc -= (d & 15);
                const unsigned db = (d >> 8) & 0xFF;
                const size_t distance = (d >> 16) + static_cast<size_t>(bb & ((1u << db) - 1)); bb >>= db; bc -= db;
                if (distance > static_cast<size_t>(o - o_min)) { err_ = true; break; } // before the start of the history
                RTK_FI_REFILL();
                e = pair[bb & ((1u << LIT_PB) - 1)]; // (looked up before the copy)
                const unsigned char* s = o - distance; unsigned char* const oe = o + len;
                if (distance >= 8) { // sixteen bytes without asking (most matches of FASTQ text are shorter), the rest in steps of eight
                    uint64_t w; memcpy(&w, s, 8); memcpy(o, &w, 8); memcpy(&w, s + 8, 8); memcpy(o + 8, &w, 8);
                    if (len > 16) { s += 16; o += 16; do { memcpy(&w, s, 8); memcpy(o, &w, 8); s += 8; o += 8; } while (o < oe); }
                }
                else if (distance == 1) { memset(o, *s, len); }
                else { do { *o++ = *s++; } while (o < oe); }
                o = oe;
                continue;
            }
            if ((e & 0xF0) == K_EOB) { state_ = final_ ? ST_TRAILER : ST_BLOCK_HEADER; break; }
            err_ = true; break; // an unused code
        }
#undef RTK_FI_REFILL
        bitbuf_ = bb; bitcnt_ = bc; in_ = in;
        return o;
    }

    const unsigned char* in_; const unsigned char* end_; const unsigned char* member_end_;
    uint64_t bitbuf_; unsigned bitcnt_; unsigned overrun_ = 0;
    State state_; bool final_, err_; uint32_t stored_left_;
    uint64_t total_out_; uint32_t crc_;
    uint32_t lit_[(1 << LIT_PB) + 4800];
    uint32_t pair_[1 << LIT_PB];
    uint32_t dist_[(1 << DIST_PB) + 4000];
};

} // namespace rtk
