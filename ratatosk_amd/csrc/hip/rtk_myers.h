// Bit-parallel Myers/Hyyro edit distance on one wavefront, with the observable behaviour of the reference's
// vendored edlib (reference: src/edlib.cpp:141-296 edlibAlign; :547-704 semi-global; :730-931 NW; :945-1144
// traceback; :1164-1216 traceback/Hirschberg switch; :1234-1399 Hirschberg; src/Common.hpp:262-276 equalities).
//
// Layout: one 64-bit word of the query per lane; the carry between vertically adjacent words travels lane->lane+1
// with one __shfl_up per step, so the wave sweeps the DP matrix along anti-diagonals ("step s: lane l works on
// column s-l"). Queries longer than 64 words are processed in row blocks of 64 words; the horizontal deltas leaving
// a block are parked in a per-column int8 array and picked up by lane 0 of the next block.
//
// Band (edlib.cpp:194-212 k doubling, :778-915 firstBlock/lastBlock): the big NW problems -- bounded distances and the half passes of
// the Hirschberg recursion, i.e. the whole-read alignment of phasing() -- only compute the Ukkonen band of their k: the cells (i, j)
// with |j - i| + |(n - m) - (j - i)| <= k, a stripe of k + 1 diagonals, at word granularity. Everything outside is an upper bound
// ("+1 per step away from the band"), so the computed value D' is >= the true D everywhere and equal to it on every cell an optimal
// path of cost <= k visits: the distance, the Hirschberg split rows (left + right == optimum only holds on optimal-path cells, which
// are exact) and the tracebacks are those of the unbanded matrix. edlib's band prunes the same way (only cells that cannot lie on an
// optimal path), which is why its results equal the unbanded oracle's (oracle/oracle_myers.hpp; pinned against the reference build).
// Where edlib doubles k until the score fits, the first split of an unknown-distance problem runs with a guess and is repeated once
// with the (real, reachable) score it found if that score exceeds the guess. Two schedules:
//   * ring (band <= 4000 diagonals, one wave): lane = word & 63; a word enters the band at the bottom, leaves it at the top 64 + band
//     columns later and its lane takes word + 64; a pass is n + W steps whatever the number of row blocks (rtk_myers_ring);
//   * blocks (wider bands): the 4096-row blocks of rtk_myers_block each sweep the columns [4096 b + dlo, 4096 b + 4095 + dhi] only.
// Small problems (one row block) and the stored sweeps of the tracebacks compute whole columns: the wave sweeps them in n + W steps anyway.
// The query profile is indexed by character class (15 IUPAC letters); any other byte is compared on the fly.
#ifndef RTK_MYERS_H
#define RTK_MYERS_H

#include "rtk_wave.h"

#define RTK_MODE_NW 0
#define RTK_MODE_SHW 1
#define RTK_MODE_HW 2

struct MyersScratch {
    U<uint64_t*> peq;       // [15 * w_cap]
    U<uint32_t> w_cap;      // max query words
    U<int8_t*> carry;       // [t_cap]
    U<int32_t*> colscore;   // [t_cap] last-row score of every column
    U<uint32_t> t_cap;      // max target columns
    U<uint64_t*> tb;        // [tb_cap_words] traceback table, 4 words per (column, query word): Pv, Mv, Ph, Mh
    U<uint64_t> tb_cap_words;
    U<int32_t*> rowL;       // [r_cap] Hirschberg: D(left half)[row]
    U<int32_t*> rowR;       // [r_cap]
    U<uint32_t> r_cap;
    U<uint8_t*> moves;      // [mv_cap] alignment moves: 0 match, 1 insert (query only), 2 delete (target only), 3 mismatch
    U<uint8_t*> moves_tmp;  // [mv_cap]
    U<uint32_t> mv_cap;
    U<int32_t*> hstack;     // [5 * 64] explicit Hirschberg stack
    UL<uint32_t*> overflow; // (device memory, or the header's own word in LDS) set to non-zero when a capacity is exceeded (work item is re-run with a bigger arena)
    U<uint32_t> tb_gen;     // bumped by everything that writes the traceback table: a saved sweep (MyersSaved) is only resumed on its own table
    U<unsigned long long> walk_cycles, walk_moves, walk_reloads, walk_scalar, walk_calls, walk_tail_cycles; // profile of the traceback walks
    U<unsigned long long> hb_pass, hb_split, hb_leaf, hb_total; // profile of the Hirschberg driver: half passes, column extraction + split search, leaf tracebacks, all
    // Optional, set by a caller around ONE rtk_myers_path call: a bit per target position, bit i set = the caller looks at the moves that happen at target position i
    // (the moves over target character i and the insertions in front of it). A sub-problem of the Hirschberg recursion none of whose target positions t0 .. t0 + tn
    // (both ends) has its bit set is not solved: its moves are written as tn deletions then qm insertions, which carry the only facts such a caller uses -- how many query
    // and target characters the stretch consumes (phasing(), rtk_phasing.h: the walk of src/Graph.cpp:991-1069 copies the corrected read wherever pos2rm has no bit).
    // Everything the split of a level decides (distance, split rows) is computed as always, so the sub-problems that ARE solved are those of the full recursion.
    U<uint64_t*> need_bm; // (NOT a pointer to const: the bitmap is written by the kernel that reads it; a pointer-to-const field is read through the scalar cache, rtk_types.h)
};
// no position of [lo, hi] (both included) has its bit set
RTK_DEV bool rtk_need_none(const uint64_t* bm, int lo, int hi) {
    for (int w = lo >> 6; w <= (hi >> 6); ++w) {
        uint64_t mk = ~0ull;
        if (w == (lo >> 6)) mk &= ~0ull << (lo & 63);
        if (w == (hi >> 6)) mk &= ~0ull >> (63 - (hi & 63));
        if (bm[w] & mk) return false;
    }
    return true;
}

// traceback table entry of (column, 64-bit query word): word-major, so that the walk's window of 64 consecutive columns of one word is
// one contiguous 2 KB run (16 cache lines) instead of 64 entries a table row apart. `ncols` = number of columns of the sweep that stored it.
#define RTK_TB(col, w, ncols) (static_cast<uint64_t>(w) * static_cast<uint64_t>(ncols) + static_cast<uint64_t>(col))
// An entry is 32 bytes: the low 32-bit halves of {Pv, Mv, Ph, Mh}, then their high halves, so that each of the two lanes that share a
// 64-bit word in the 32-bit sweep writes its four words with ONE 16-byte store.
struct RtkTbHalf { uint32_t pv, mv, ph, mh; };
#if defined(RTK_EXPERIMENT_NO_TB) && !defined(RTK_SIM) // timing experiment only (results are wrong): the sweeps do not write the table at all
#define RTK_TB_ST(p, hv) ((void)(p), (void)(hv))
#elif defined(RTK_TB_NT) && !defined(RTK_SIM) // A/B build: the table is written once and read along one path: streaming stores, so that it does not push the waves' stacks out of the L2
typedef uint32_t rtk_v4u __attribute__((ext_vector_type(4)));
#define RTK_TB_ST(p, hv) __builtin_nontemporal_store(rtk_v4u{(hv).pv, (hv).mv, (hv).ph, (hv).mh}, reinterpret_cast<rtk_v4u*>(p))
#else
#define RTK_TB_ST(p, hv) (*(p) = (hv))
#endif
RTK_DEV void rtk_tb_put(uint64_t* e, uint64_t Pv, uint64_t Mv, uint64_t Ph, uint64_t Mh) {
    RtkTbHalf lo, hi;
    lo.pv = static_cast<uint32_t>(Pv); lo.mv = static_cast<uint32_t>(Mv); lo.ph = static_cast<uint32_t>(Ph); lo.mh = static_cast<uint32_t>(Mh);
    hi.pv = static_cast<uint32_t>(Pv >> 32); hi.mv = static_cast<uint32_t>(Mv >> 32); hi.ph = static_cast<uint32_t>(Ph >> 32); hi.mh = static_cast<uint32_t>(Mh >> 32);
    reinterpret_cast<RtkTbHalf*>(e)[0] = lo; reinterpret_cast<RtkTbHalf*>(e)[1] = hi;
}
RTK_DEV void rtk_tb_get(const uint64_t* e, uint64_t& Pv, uint64_t& Mv, uint64_t& Ph, uint64_t& Mh) {
    const RtkTbHalf lo = reinterpret_cast<const RtkTbHalf*>(e)[0], hi = reinterpret_cast<const RtkTbHalf*>(e)[1];
    Pv = static_cast<uint64_t>(lo.pv) | (static_cast<uint64_t>(hi.pv) << 32); Mv = static_cast<uint64_t>(lo.mv) | (static_cast<uint64_t>(hi.mv) << 32);
    Ph = static_cast<uint64_t>(lo.ph) | (static_cast<uint64_t>(hi.ph) << 32); Mh = static_cast<uint64_t>(lo.mh) | (static_cast<uint64_t>(hi.mh) << 32);
}

struct MySeq { // a character sequence read forwards or backwards (Hirschberg aligns reversed halves, edlib.cpp:1259-1263)
    const char* p; int32_t n; int32_t rev;
};
RTK_DEV MySeq rtk_seq(const char* p, int32_t n, int32_t rev = 0) { MySeq s; s.p = p; s.n = n; s.rev = rev; return s; }
RTK_DEV unsigned char rtk_seq_at(const MySeq& s, int32_t i) { return static_cast<unsigned char>(s.rev ? s.p[s.n - 1 - i] : s.p[i]); }

// character class: 0..3 A C G T, 4..14 M R S V W Y H K D B N, 15 anything else
RTK_DEV int rtk_cls(unsigned char c) {
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        case 'M': return 4; case 'R': return 5; case 'S': return 6; case 'V': return 7; case 'W': return 8; case 'Y': return 9;
        case 'H': return 10; case 'K': return 11; case 'D': return 12; case 'B': return 13; case 'N': return 14;
        default: return 15;
    }
}

// Branch-free 2-bit packing of up to 32 characters (A 0, C 1, G 2, T 3; first character in the high bits of the result, `want`
// characters in total, unaligned source with at least 32 readable bytes). *n_ok = number of leading characters that are A/C/G/T
// (upper case), capped at `want`; the code is only meaningful for those. Bits 1-2 of 'A' 0x41, 'C' 0x43, 'T' 0x54, 'G' 0x47 are a
// 2-bit code; a character is valid iff rebuilding it from that code gives it back. No per-character branches: the per-lane switch
// of rtk_cls costs ~50 scalar instructions per character in exec-mask bookkeeping.
RTK_DEV uint64_t rtk_spread8(uint64_t v) { v = (v | (v << 4)) & 0x0F0Full; v = (v | (v << 2)) & 0x3333ull; return (v | (v << 1)) & 0x5555ull; } // bit i of a byte -> bit 2i
// inv2 (optional): bit 2(want-1-i) set iff character i is not A/C/G/T (same layout as the low code bits)
RTK_DEV uint64_t rtk_pack_acgt(const unsigned char* p, int want, int* n_ok, uint64_t* inv2 = nullptr) {
    uint64_t code = 0, inv = 0; int ok = 0; bool open = true;
    for (int j = 0; j < 4; ++j) {
        uint64_t x; __builtin_memcpy(&x, p + 8 * j, 8);
        const uint64_t b1 = (x >> 1) & 0x0101010101010101ull, b2 = (x >> 2) & 0x0101010101010101ull;
        const uint64_t b12 = b1 & b2, b2n = b2 & ~b1;
        const uint64_t recon = 0x4141414141414141ull + (b1 << 1) + (b12 << 2) + (b2n << 4) + (b2n << 1) + b2n;
        const uint64_t bad = x ^ recon;
        const int good = bad ? (__builtin_ctzll(bad) >> 3) : 8; // leading valid characters of this word
        // gather the bit planes, first character first: byte i -> bit 7 - i
        const uint64_t hi = (b2 * 0x8040201008040201ull) >> 56, lo = ((b1 ^ b2) * 0x8040201008040201ull) >> 56;
        code = (code << 16) | (rtk_spread8(hi) << 1) | rtk_spread8(lo);
        if (inv2) {
            const uint64_t nz = ((((bad & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | bad) >> 7) & 0x0101010101010101ull; // 1 per non-zero byte
            inv = (inv << 16) | rtk_spread8((nz * 0x8040201008040201ull) >> 56);
        }
        if (open) { ok += good; open = good == 8; }
    }
    // 32 characters packed; keep the first `want`
    if (want < 32) { code >>= 2 * (32 - want); inv >>= 2 * (32 - want); }
    if (inv2) *inv2 = inv;
    *n_ok = ok < want ? ok : want;
    return code;
}

// the k-mer code (k <= 63) of the k characters at p; false when one of them is not A/C/G/T. Reads at most max(32, k) bytes from p.
RTK_DEV bool rtk_km_from_text(const unsigned char* p, int k, RtkKm* out) {
    int n_ok;
    if (k <= 32) { out->hi = 0; out->lo = rtk_pack_acgt(p, k, &n_ok); return n_ok == k; }
    out->hi = rtk_pack_acgt(p, k - 32, &n_ok);
    if (n_ok != k - 32) return false;
    out->lo = rtk_pack_acgt(p + (k - 32), 32, &n_ok);
    return n_ok == 32;
}

// 15-bit set of classes a character of class c is equal to (identity + the 28 (code, base) pairs of Common.hpp:262-274)
RTK_DEV uint32_t rtk_eq_classes(int c, bool iupac) {
    if (c >= 15) return 0u;
    uint32_t m = 1u << c;
    if (!iupac) return m;
    // base sets of the codes:           M    R    S    V    W    Y    H     K     D     B     N
    const uint32_t code_set[11] = {0x3, 0x5, 0x6, 0x7, 0x9, 0xA, 0xB, 0xC, 0xD, 0xE, 0xF};
    if (c < 4) { for (int i = 0; i < 11; ++i) if ((code_set[i] >> c) & 1u) m |= 1u << (4 + i); }
    else m |= code_set[c - 4];
    return m;
}

RTK_DEV bool rtk_chars_equal(unsigned char a, unsigned char b, bool iupac) {
    if (a == b) return true;
    const int ca = rtk_cls(a), cb = rtk_cls(b);
    if (ca >= 15 || cb >= 15) return false;
    return (rtk_eq_classes(ca, iupac) >> cb) & 1u;
}

// One Advance_Block step (Myers 1999 / Hyyro 2003). hin, return value in {-1,0,1}; bit = row whose horizontal delta leaves the word.
RTK_DEV int rtk_myers_step(uint64_t& Pv, uint64_t& Mv, uint64_t Eq, int hin, int bit, uint64_t& Ph_out, uint64_t& Mh_out) {
    const uint64_t pv = Pv, mv = Mv;
    const uint64_t Xv = Eq | mv;
    if (hin < 0) Eq |= 1ull;
    const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
    uint64_t Ph = mv | ~(Xh | pv);
    uint64_t Mh = pv & Xh;
    Ph_out = Ph; Mh_out = Mh;
    const int hout = static_cast<int>((Ph >> bit) & 1ull) - static_cast<int>((Mh >> bit) & 1ull);
    Ph <<= 1; Mh <<= 1;
    if (hin > 0) Ph |= 1ull; else if (hin < 0) Mh |= 1ull;
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}

// Query profile: peq[cls * W + w], bit i of word w set iff query[64w+i] equals a character of class cls.
RTK_FN void rtk_myers_build_peq(const MyersScratch& sc, MySeq q, int W, bool iupac) {
    for (int w = rtk_lane(); w < W; w += RTK_WAVE) {
        uint64_t acc[15];
        for (int c = 0; c < 15; ++c) acc[c] = 0;
        const int lim = (q.n - 64 * w) < 64 ? (q.n - 64 * w) : 64;
        for (int i = 0; i < lim; ++i) {
            uint32_t m = rtk_eq_classes(rtk_cls(rtk_seq_at(q, 64 * w + i)), iupac);
            for (int c = 0; c < 15; ++c) acc[c] |= static_cast<uint64_t>((m >> c) & 1u) << i;
        }
        for (int c = 0; c < 15; ++c) sc.peq[static_cast<uint64_t>(c) * W + w] = acc[c];
    }
    rtk_sync();
}

RTK_DEV uint64_t rtk_myers_eq_word(const MyersScratch& sc, const MySeq& q, int W, int w, unsigned char tc) {
    const int cls = rtk_cls(tc);
    if (cls < 15) return sc.peq[static_cast<uint64_t>(cls) * W + w];
    uint64_t e = 0; // a byte outside the IUPAC alphabet only equals itself
    const int lim = (q.n - 64 * w) < 64 ? (q.n - 64 * w) : 64;
    for (int i = 0; i < lim; ++i) e |= static_cast<uint64_t>(rtk_seq_at(q, 64 * w + i) == tc) << i;
    return e;
}

#ifndef RTK_SIM
// Branch-free anti-diagonal sweep for the common case: query <= 64 words (one row block), target made of A/C/G/T only.
// Scalar branches are expensive on CDNA (instruction-fetch restart), so everything per step is predicated with
// v_cndmask; the only branch left is the once-per-64-columns flush of the score buffer.
template <int STORE>
__device__ __forceinline__ void rtk_myers_sweep_acgt(int m, int n, int W, int top_h, int last_bit, const char* __restrict__ tp, int trev,
                                                     uint64_t eqA, uint64_t eqC, uint64_t eqG, uint64_t eqT,
                                                     int32_t* __restrict__ colscore, uint64_t* __restrict__ tb, uint64_t& Pv_out, uint64_t& Mv_out) {
    const int lane = rtk_lane();
    const int w = lane;
    const bool has_word = lane < W;
    const int bit = (w == W - 1) ? last_bit : 63;
    uint64_t Pv = ~0ull, Mv = 0ull;
    int hout_prev = 0; unsigned tc_prev = 0;
    int score = m, sbuf = 0;
    const int steps = n + W - 1;
    for (int c0 = 0; c0 < steps; c0 += 64) {
        const int cj = c0 + lane;
        int my_t = (cj < n) ? static_cast<int>(static_cast<unsigned char>(trev ? tp[n - 1 - cj] : tp[cj])) : 0;
        asm volatile("" : "+v"(my_t));
        const int lim = (steps - c0) < 64 ? (steps - c0) : 64;
        for (int j = 0; j < lim; ++j) {
            const int s = c0 + j;
            const unsigned in_t = static_cast<unsigned>(__builtin_amdgcn_readlane(my_t, j));
            const unsigned mine = static_cast<unsigned>(hout_prev + 1) | (tc_prev << 8);
            const unsigned got = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(top_h + 1) | (in_t << 8)), static_cast<int>(mine), 0x138, 0xF, 0xF, false));
            const int hin = static_cast<int>(got & 0xFFu) - 1;
            const unsigned tc = got >> 8;
            const unsigned sel = (tc >> 1) & 3u; // 'A' -> 0, 'C' -> 1, 'T' -> 2, 'G' -> 3
            const uint64_t Eq = (sel & 2u) ? ((sel & 1u) ? eqG : eqT) : ((sel & 1u) ? eqC : eqA);
            uint64_t nPv = Pv, nMv = Mv, Ph, Mh;
            const int hout = rtk_myers_step(nPv, nMv, Eq, hin, bit, Ph, Mh);
            const int col = s - lane;
            const bool active = has_word && col >= 0 && col < n;
            if (STORE) { if (active) { rtk_tb_put(tb + 4ull * RTK_TB(col, w, n), nPv, nMv, Ph, Mh); } }
            Pv = active ? nPv : Pv; Mv = active ? nMv : Mv;
            hout_prev = active ? hout : hout_prev;
            score += (active && lane == W - 1) ? hout : 0;
            tc_prev = tc;
            const int tcol = s - (W - 1);
            if (tcol >= 0) { // uniform
                const int sv = __builtin_amdgcn_readlane(score, W - 1);
                sbuf = (lane == (tcol & 63)) ? sv : sbuf;
                if ((tcol & 63) == 63 || tcol == n - 1) { const int cc = (tcol & ~63) + lane; if (cc <= tcol) colscore[cc] = sbuf; }
            }
        }
    }
    Pv_out = Pv; Mv_out = Mv;
}
#endif

#ifndef RTK_SIM
// Self-contained fast path for forward sequences with <= 64 query words: profile words built from 8 wide loads per lane,
// branch-free sweep, running minimum kept in scalar registers (no per-column score array), A/C/G/T check folded into the
// per-64-column target load. Returns plain == false (and no result) when the target holds another character.
struct SweepStat { int final_score, best, first, last, cnt; bool plain; };
template <int STORE>
__device__ __forceinline__ SweepStat rtk_myers_fast(const char* __restrict__ qp, int m, const char* __restrict__ tp, int n, int top_h, bool iupac, uint64_t* __restrict__ tb) {
    SweepStat st; st.final_score = m; st.best = 0x7fffffff; st.first = -1; st.last = -1; st.cnt = 0; st.plain = true;
    const int lane = rtk_lane();
    const int W = (m + 63) >> 6, last_bit = (m - 1) & 63;
    const int w = lane;
    const bool has_word = lane < W;
    uint64_t eqA = 0, eqC = 0, eqG = 0, eqT = 0;
    if (has_word) {
        const int lim = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
        uint64_t qw[8];
        for (int j = 0; j < 8; ++j) { uint64_t x = 0; if (8 * j < lim) __builtin_memcpy(&x, qp + 64 * w + 8 * j, 8); qw[j] = x; } // may read up to 7 bytes past the query inside its padded buffer
        for (int i = 0; i < lim; ++i) {
            const unsigned char qc = static_cast<unsigned char>((qw[i >> 3] >> (8 * (i & 7))) & 0xFFull);
            uint32_t bm;
            if (qc == 'A') bm = 1u; else if (qc == 'C') bm = 2u; else if (qc == 'G') bm = 4u; else if (qc == 'T') bm = 8u;
            else bm = rtk_eq_classes(rtk_cls(qc), iupac) & 0xFu;
            eqA |= static_cast<uint64_t>(bm & 1u) << i; eqC |= static_cast<uint64_t>((bm >> 1) & 1u) << i;
            eqG |= static_cast<uint64_t>((bm >> 2) & 1u) << i; eqT |= static_cast<uint64_t>((bm >> 3) & 1u) << i;
        }
    }
    const int bit = (w == W - 1) ? last_bit : 63;
    uint64_t Pv = ~0ull, Mv = 0ull;
    int hout_prev = 0; unsigned tc_prev = 0;
    int score = m;
    int best = 0x7fffffff, first = -1, last = -1, cnt = 0, fin = m;
    const int steps = n + W - 1;
    for (int c0 = 0; c0 < steps; c0 += 64) {
        const int cj = c0 + lane;
        int my_t = 'A';
        if (cj < n) my_t = static_cast<int>(static_cast<unsigned char>(tp[cj]));
        if (rtk_ballot(!(my_t == 'A' || my_t == 'C' || my_t == 'G' || my_t == 'T')) != 0ull) { st.plain = false; return st; }
        asm volatile("" : "+v"(my_t));
        const int lim = (steps - c0) < 64 ? (steps - c0) : 64;
        for (int j = 0; j < lim; ++j) {
            const int s = c0 + j;
            const unsigned in_t = static_cast<unsigned>(__builtin_amdgcn_readlane(my_t, j));
            const unsigned mine = static_cast<unsigned>(hout_prev + 1) | (tc_prev << 8);
            const unsigned got = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(top_h + 1) | (in_t << 8)), static_cast<int>(mine), 0x138, 0xF, 0xF, false));
            const int hin = static_cast<int>(got & 0xFFu) - 1;
            const unsigned tc = got >> 8;
            const unsigned sel = (tc >> 1) & 3u; // 'A' -> 0, 'C' -> 1, 'T' -> 2, 'G' -> 3
            const uint64_t Eq = (sel & 2u) ? ((sel & 1u) ? eqG : eqT) : ((sel & 1u) ? eqC : eqA);
            uint64_t nPv = Pv, nMv = Mv, Ph, Mh;
            const int hout = rtk_myers_step(nPv, nMv, Eq, hin, bit, Ph, Mh);
            const int col = s - lane;
            const bool active = has_word && col >= 0 && col < n;
            if (STORE) { if (active) { rtk_tb_put(tb + 4ull * RTK_TB(col, w, n), nPv, nMv, Ph, Mh); } }
            Pv = active ? nPv : Pv; Mv = active ? nMv : Mv;
            hout_prev = active ? hout : hout_prev;
            score += (active && lane == W - 1) ? hout : 0;
            tc_prev = tc;
            const int tcol = s - (W - 1);
            if (tcol >= 0) { // wave-uniform: last-row score of column tcol, tracked in scalar registers
                const int sv = __builtin_amdgcn_readlane(score, W - 1);
                fin = sv;
                if (sv < best) { best = sv; first = tcol; last = tcol; cnt = 1; }
                else if (sv == best) { last = tcol; ++cnt; }
            }
        }
    }
    st.final_score = fin; st.best = best; st.first = first; st.last = last; st.cnt = cnt;
    return st;
}
#endif
#ifndef RTK_SIM
// The same sweep on 32-bit words (queries up to 2048 characters = 64 lanes): every bit-vector operation is one VALU instruction
// instead of two, and twice as many lanes work per step. The deltas are properties of the DP matrix, not of the word size, so the
// results - and the traceback table, written as the two halves of the 64-bit entries rtk_myers_walk reads - are the same.
RTK_DEV int rtk_myers_step32(uint32_t& Pv, uint32_t& Mv, uint32_t Eq, int hin, int bit, uint32_t& Ph_out, uint32_t& Mh_out) {
    const uint32_t pv = Pv, mv = Mv;
    const uint32_t Xv = Eq | mv;
    Eq |= static_cast<uint32_t>(hin) >> 31; // hin < 0
    const uint32_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
    uint32_t Ph = mv | ~(Xh | pv);
    uint32_t Mh = pv & Xh;
    Ph_out = Ph; Mh_out = Mh;
    const int hout = static_cast<int>((Ph >> bit) & 1u) - static_cast<int>((Mh >> bit) & 1u);
    Ph = (Ph << 1) | (hin > 0 ? 1u : 0u); Mh = (Mh << 1) | (static_cast<uint32_t>(hin) >> 31);
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}

// TRACK = 0: only the final score is wanted (NW); 1: minimum of the last row with its first / last position and count (SHW, HW).
template <int STORE, int TRACK>
__device__ __forceinline__ SweepStat rtk_myers_fast32(const char* __restrict__ qp, int m, const char* __restrict__ tp, int n, int top_h, bool iupac, uint64_t* __restrict__ tb) {
    SweepStat st; st.final_score = m; st.best = 0x7fffffff; st.first = -1; st.last = -1; st.cnt = 0; st.plain = true;
    const int lane = rtk_lane();
    const int W = (m + 31) >> 5, W64 = (m + 63) >> 6, last_bit = (m - 1) & 31;
    const int w = lane;
    const bool has_word = lane < W;
    uint32_t eqA = 0, eqC = 0, eqG = 0, eqT = 0;
    if (has_word) {
        const int lim = (m - 32 * w) < 32 ? (m - 32 * w) : 32;
        uint64_t qw[4];
        for (int j = 0; j < 4; ++j) { uint64_t x = 0; if (8 * j < lim) __builtin_memcpy(&x, qp + 32 * w + 8 * j, 8); qw[j] = x; } // may read up to 7 bytes past the query inside its padded buffer
        // four characters at a time: bits 1 and 2 of 'A' 0x41, 'C' 0x43, 'T' 0x54, 'G' 0x47 are a 2-bit code (0, 1, 2, 3); the two bit
        // planes are gathered into 32-bit masks and the four profile words are their boolean combinations. A word holding anything
        // else than A/C/G/T (IUPAC codes, N, the bytes past the query) is left to the character-by-character loop below.
        uint32_t p1 = 0, p2 = 0, done = 0;
        for (int j = 0; j < 8; ++j) {
            if (4 * j >= lim) break;
            const uint32_t x = static_cast<uint32_t>(qw[j >> 1] >> (32 * (j & 1)));
            const uint32_t b1 = (x >> 1) & 0x01010101u, b2 = (x >> 2) & 0x01010101u;
            const uint32_t b12 = b1 & b2, b2n = b2 & ~b1;
            const uint32_t recon = 0x41414141u + (b1 << 1) + (b12 << 2) + (b2n << 4) + (b2n << 1) + b2n; // 0x41 + 2*b1 + 4*b1*b2 + 0x13*(b2 & !b1), per byte
            if (x != recon || 4 * j + 4 > lim) continue;
            const uint32_t n1 = (b1 & 1u) | ((b1 >> 7) & 2u) | ((b1 >> 14) & 4u) | ((b1 >> 21) & 8u);
            const uint32_t n2 = (b2 & 1u) | ((b2 >> 7) & 2u) | ((b2 >> 14) & 4u) | ((b2 >> 21) & 8u);
            p1 |= n1 << (4 * j); p2 |= n2 << (4 * j); done |= 0xFu << (4 * j);
        }
        eqA = ~p1 & ~p2 & done; eqC = p1 & ~p2 & done; eqT = ~p1 & p2 & done; eqG = p1 & p2 & done;
        if (done != ((lim >= 32) ? ~0u : ((1u << lim) - 1u))) {
            for (int i = 0; i < lim; ++i) {
                if ((done >> i) & 1u) continue;
                const unsigned char qc = static_cast<unsigned char>((qw[i >> 3] >> (8 * (i & 7))) & 0xFFull);
                uint32_t bm;
                if (qc == 'A') bm = 1u; else if (qc == 'C') bm = 2u; else if (qc == 'G') bm = 4u; else if (qc == 'T') bm = 8u;
                else bm = rtk_eq_classes(rtk_cls(qc), iupac) & 0xFu;
                eqA |= (bm & 1u) << i; eqC |= ((bm >> 1) & 1u) << i; eqG |= ((bm >> 2) & 1u) << i; eqT |= ((bm >> 3) & 1u) << i;
            }
        }
    }
    const int bit = (w == W - 1) ? last_bit : 31;
    uint32_t Pv = ~0u, Mv = 0u;
    int hout_prev = 0; uint32_t m1_prev = 0, m2_prev = 0;
    int score = m;
    int vbest = 0x7fffffff, vfirst = -1, vlast = -1, vcnt = 0;
    const int steps = n + W - 1;
    // table entry of (column, 64-bit word) = two RtkTbHalf (low halves, high halves of Pv, Mv, Ph, Mh); this lane owns one of them
    RtkTbHalf* const tbh = reinterpret_cast<RtkTbHalf*>(tb) + 2ull * RTK_TB(0, w >> 1, n) + (w & 1); // this lane's half of the entries of its 64-bit word
    // one step of the anti-diagonal pipeline. MASKED = 1 while the pipeline fills or drains (some words have no column yet / any
    // more); in between every word has one and the activity test, the column range checks and the conditional updates fall away.
#define RTK_STEP32(MASKED)                                                                                                                   \
    {                                                                                                                                        \
        const int s = c0 + j;                                                                                                                \
        const int in_t = __builtin_amdgcn_readlane(my_t, j);                                                                                 \
        /* 'A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54: bit 1 picks C/G over A/T, bit 2 picks T/G over A/C. The two choices travel down the  */ \
        /* lanes as full-width masks next to the horizontal delta (three DPP moves), so that the profile word is three bitwise selects.  */ \
        const int hin = __builtin_amdgcn_update_dpp(top_h, hout_prev, 0x138, 0xF, 0xF, false);                                              \
        const uint32_t m1 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 1, 1), static_cast<int>(m1_prev), 0x138, 0xF, 0xF, false)); \
        const uint32_t m2 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 2, 1), static_cast<int>(m2_prev), 0x138, 0xF, 0xF, false)); \
        const uint32_t lo_ = (eqC & m1) | (eqA & ~m1), hi_ = (eqG & m1) | (eqT & ~m1);                                                       \
        const uint32_t Eq = (hi_ & m2) | (lo_ & ~m2);                                                                                        \
        uint32_t nPv = Pv, nMv = Mv, Ph, Mh;                                                                                                 \
        const int hout = rtk_myers_step32(nPv, nMv, Eq, hin, bit, Ph, Mh);                                                                   \
        const int col = s - lane;                                                                                                            \
        if (MASKED) {                                                                                                                        \
            const bool active = has_word && col >= 0 && col < n;                                                                             \
            if (STORE) { if (active) { RtkTbHalf hv; hv.pv = nPv; hv.mv = nMv; hv.ph = Ph; hv.mh = Mh; RTK_TB_ST(&tbh[2ull * static_cast<uint64_t>(col)], hv); } } \
            Pv = active ? nPv : Pv; Mv = active ? nMv : Mv;                                                                                  \
            hout_prev = active ? hout : hout_prev;                                                                                           \
            score += active ? hout : 0;                                                                                                      \
        } else {                                                                                                                             \
            if (STORE) { if (has_word) { RtkTbHalf hv; hv.pv = nPv; hv.mv = nMv; hv.ph = Ph; hv.mh = Mh; RTK_TB_ST(&tbh[2ull * static_cast<uint64_t>(col)], hv); } } \
            Pv = nPv; Mv = nMv; hout_prev = hout; score += hout;                                                                             \
        }                                                                                                                                    \
        m1_prev = m1; m2_prev = m2;                                                                                                          \
        if (TRACK) {                                                                                                                         \
            const int tcol = s - (W - 1);                                                                                                    \
            if (!(MASKED) || tcol >= 0) { /* every lane follows the minimum of ITS word's last row (selects, no branches); lane W - 1's is the one read at the end */ \
                const bool lt_ = score < vbest, eq_ = score == vbest;                                                                        \
                vbest = lt_ ? score : vbest; vfirst = lt_ ? tcol : vfirst; vlast = (lt_ || eq_) ? tcol : vlast; vcnt = lt_ ? 1 : (vcnt + (eq_ ? 1 : 0)); \
            }                                                                                                                                \
        }                                                                                                                                    \
    }
    int nxt_t = 'A'; // target characters of the NEXT 64 columns: requested one block ahead, so their latency hides behind 64 steps
    if (lane < n) nxt_t = static_cast<int>(static_cast<unsigned char>(tp[lane]));
    for (int c0 = 0; c0 < steps; c0 += 64) {
        int my_t = nxt_t;
        { const int cn = c0 + 64 + lane; nxt_t = 'A'; if (cn < n) nxt_t = static_cast<int>(static_cast<unsigned char>(tp[cn])); }
        if (rtk_ballot(!(my_t == 'A' || my_t == 'C' || my_t == 'G' || my_t == 'T')) != 0ull) { st.plain = false; return st; }
        asm volatile("" : "+v"(my_t));
        const int lim = (steps - c0) < 64 ? (steps - c0) : 64;
        // steps [0, W-1) fill the pipeline, [W-1, n) run it full, [n, n+W-1) drain it
        int j_fill = (W - 1) - c0; j_fill = j_fill < 0 ? 0 : (j_fill > lim ? lim : j_fill);
        int j_full = n - c0; j_full = j_full < j_fill ? j_fill : (j_full > lim ? lim : j_full);
        int j = 0;
        for (; j < j_fill; ++j) RTK_STEP32(1)
#ifdef RTK_MY_UNROLL
        _Pragma("unroll 4")
#endif
        for (; j < j_full; ++j) RTK_STEP32(0)
        for (; j < lim; ++j) RTK_STEP32(1)
    }
#undef RTK_STEP32
    st.final_score = __builtin_amdgcn_readlane(score, W - 1);
    if (TRACK) { st.best = __builtin_amdgcn_readlane(vbest, W - 1); st.first = __builtin_amdgcn_readlane(vfirst, W - 1); st.last = __builtin_amdgcn_readlane(vlast, W - 1); st.cnt = __builtin_amdgcn_readlane(vcnt, W - 1); }
    return st;
}

// dispatcher: 32-bit words up to 2048 query characters, 64-bit words beyond
template <int STORE, int TRACK>
__device__ __forceinline__ SweepStat rtk_myers_fast_any(const char* __restrict__ qp, int m, const char* __restrict__ tp, int n, int top_h, bool iupac, uint64_t* __restrict__ tb) {
    if (m <= 2048) return rtk_myers_fast32<STORE, TRACK>(qp, m, tp, n, top_h, iupac, tb);
    return rtk_myers_fast<STORE>(qp, m, tp, n, top_h, iupac, tb);
}
#endif


// ---- band of a pass: diagonals dlo <= j - i <= dhi (i = query row, j = target column, 0-based) -----------------------------------------------
#define RTK_BAND_FULL (1 << 29)
#define RTK_BAND_INF 0x3fffffff      // score of a row outside the band
#define RTK_RING_MAX_BAND 4000       // widest band (diagonals) of the ring schedule: a lane must be done with word w before word w + 64 starts
struct RtkBand { int dlo, dhi; };
RTK_HD RtkBand rtk_band_full() { RtkBand b; b.dlo = -RTK_BAND_FULL; b.dhi = RTK_BAND_FULL; return b; }
// Ukkonen band of an NW problem of m rows and n columns whose distance is <= k (k is raised to |n - m| if smaller)
RTK_HD RtkBand rtk_band_nw(int m, int n, int k) {
    const int d = n - m, ad = d < 0 ? -d : d;
    const int h = k > ad ? (k - ad) / 2 : 0;
    RtkBand b; b.dlo = (d < 0 ? d : 0) - h; b.dhi = (d > 0 ? d : 0) + h; return b;
}
RTK_HD bool rtk_band_is_full(const RtkBand& b, int m, int n) { return b.dlo <= -m && b.dhi >= n; }
// words computed together share their column range: gran = 1 (ring: every word its own) or 64 (row blocks)
RTK_HD int rtk_band_clo(int w, int dlo, int gran) { const long long c = 64LL * ((w / gran) * gran) + dlo; return c < 0 ? 0 : static_cast<int>(c); }
RTK_HD int rtk_band_chi(int w, int W, int n, int dhi, int gran) { int wl = (w / gran) * gran + gran - 1; if (wl > W - 1) wl = W - 1; const long long c = 64LL * wl + 63 + dhi; return c > n - 1 ? n - 1 : static_cast<int>(c); }
// words that have a column at all: the band reaches row n - 1 - dlo at most
RTK_HD int rtk_band_words(int m, int n, int dlo) { const int W = (m + 63) >> 6; const long long r = static_cast<long long>(n) - 1 - dlo; if (r < 0) return 0; const long long w = (r >> 6) + 1; return w < W ? static_cast<int>(w) : W; }
// schedule of a banded pass: the ring (every word its own column range) when it pays and fits, row blocks otherwise
RTK_HD int rtk_band_gran(int m, int dlo, int dhi) { return (((m + 63) >> 6) > 64 && (dhi - dlo) <= RTK_RING_MAX_BAND) ? 1 : 64; }
// first guess of the distance of an NW problem (the top of a Hirschberg recursion): generous where the ring makes the width free
RTK_HD int rtk_band_guess(int m, int n) {
    const int d = n - m, ad = d < 0 ? -d : d, mn = m < n ? m : n;
    int g = ad + (mn / 8 > 64 ? mn / 8 : 64);
    if (g < RTK_RING_MAX_BAND - 100) g = RTK_RING_MAX_BAND - 100;
    return g < m + n ? g : m + n;
}

struct RtkGangCtx { const char* q; const char* t; char* stage; uint64_t* peq; uint64_t* fin; int32_t* nodes; int n_nodes, iupac; }; // one round of gang sweeps (rtk_myers_lvl.h)
struct RtkCoopJob { const char* qp; const char* tp; int m, n, qrev, trev, top_h, iupac; uint64_t* fin_pv; uint64_t* fin_mv; int8_t* carry; int32_t* colscore; int dlo, dhi, ring; uint64_t* peq4; };
#ifndef RTK_SIM
__device__ __forceinline__ int rtk_coop_ld(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void rtk_coop_st(int* p, int v) { if (rtk_lane() == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// One row block (64 words = 4096 query rows) of a pass over several blocks: same arithmetic as the general loop of rtk_myers_pass, without
// branches in the step when the target holds A/C/G/T only. progress == nullptr: the blocks are swept one after the other by this wave;
// otherwise the counters of the pass (completed columns per block), through which the waves of a workgroup follow each other.
// Band (J.dlo, J.dhi): the block only sweeps the columns [cb_lo, cb_hi] in which one of its words has a row inside the band; it starts from
// the "+1 per row" vertical deltas, takes "+1 per column" from above once the block above is past its own last column, and leaves the
// vertical deltas of column cb_hi in fin_pv / fin_mv (rtk_myers_column knows which rows of them are inside the band at the last column).
__device__ __forceinline__ unsigned char rtk_job_qc(const char* __restrict__ qp, int m, int qrev, int i) { return static_cast<unsigned char>(qrev ? qp[m - 1 - i] : qp[i]); }
__device__ __forceinline__ void rtk_myers_eq4(const char* __restrict__ qp, int m, int qrev, int w, bool iupac, uint64_t& eqA, uint64_t& eqC, uint64_t& eqG, uint64_t& eqT) {
    eqA = 0; eqC = 0; eqG = 0; eqT = 0;
    const int lim = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
    for (int i = 0; i < lim; ++i) {
        const unsigned char qc = rtk_job_qc(qp, m, qrev, 64 * w + i);
        uint32_t bm;
        if (qc == 'A') bm = 1u; else if (qc == 'C') bm = 2u; else if (qc == 'G') bm = 4u; else if (qc == 'T') bm = 8u;
        else bm = rtk_eq_classes(rtk_cls(qc), iupac) & 0xFu; // bases this query character equals
        eqA |= static_cast<uint64_t>(bm & 1u) << i; eqC |= static_cast<uint64_t>((bm >> 1) & 1u) << i;
        eqG |= static_cast<uint64_t>((bm >> 2) & 1u) << i; eqT |= static_cast<uint64_t>((bm >> 3) & 1u) << i;
    }
}
__device__ __noinline__ void rtk_myers_block(const RtkCoopJob& J, int b, int* progress) {
    const char* __restrict__ const qp = rtk_u(J.qp); const char* __restrict__ const tp = rtk_u(J.tp);
    const int m = rtk_u(J.m), n = rtk_u(J.n), qrev = rtk_u(J.qrev), trev = rtk_u(J.trev), top_h = rtk_u(J.top_h); const bool iupac = rtk_u(J.iupac) != 0;
    const int dlo = rtk_u(J.dlo), dhi = rtk_u(J.dhi);
    int8_t* __restrict__ const carry = rtk_u(J.carry); int32_t* __restrict__ const colscore = rtk_u(J.colscore);
    uint64_t* const fin_pv = rtk_u(J.fin_pv); uint64_t* const fin_mv = rtk_u(J.fin_mv);
    const int W = (m + 63) >> 6, last_bit = (m - 1) & 63;
    const int lane = rtk_lane();
    const int w0 = 64 * b;
    const int nw = (W - w0) < 64 ? (W - w0) : 64;
    const int w = w0 + lane;
    const bool has_word = lane < nw;
    const bool is_last_word = has_word && (w == W - 1);
    const bool is_block_tail = (lane == nw - 1);
    const int bit = is_last_word ? last_bit : 63;
    const bool last_block = (w0 + nw >= W);
    const bool banded = !(dlo <= -m && dhi >= n);
    const int cb_lo = rtk_band_clo(w0, dlo, 64), cb_hi = rtk_band_chi(w0, W, n, dhi, 64); // columns of this block
    const int prev_hi = b > 0 ? rtk_band_chi(w0 - 1, W, n, dhi, 64) : -1;               // last column of the block above
    const int above_h = b == 0 ? top_h : 1;                                              // delta entering the block where nothing is above it
    if (cb_lo > cb_hi) { // no column (below the band): nothing to do, nobody waits
        if (progress) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (!last_block) rtk_coop_st(&progress[b], n); } else rtk_sync();
        return;
    }
    const int nc = cb_hi - cb_lo + 1;
    uint64_t eqA = 0, eqC = 0, eqG = 0, eqT = 0;
    if (has_word) rtk_myers_eq4(qp, m, qrev, w, iupac, eqA, eqC, eqG, eqT);
    uint64_t Pv = ~0ull, Mv = 0ull;
    int hout_prev = 0; unsigned tc_prev = 0;
    int score = m;
    const int steps = nc + nw - 1;
    int sbuf = 0;
    const bool track = last_block && !banded; // bottom-row scores of every column: only meaningful when the sweep starts at column 0
    bool plain = true; // target made of A/C/G/T only: the profile word is a select, no branches in the step
    for (int c0 = cb_lo; c0 <= cb_hi && plain; c0 += 64) {
        const int cj = c0 + lane; bool okc = true;
        if (cj <= cb_hi) { const unsigned char ch = static_cast<unsigned char>(trev ? tp[n - 1 - cj] : tp[cj]); okc = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
        plain = (rtk_ballot(!okc) == 0ull);
    }
    for (int c0 = 0; c0 < steps; c0 += 64) {
        const int cj = cb_lo + c0 + lane; // absolute column this lane fetches for the chunk
        if (b > 0 && progress && cb_lo + c0 <= prev_hi) { // the deltas of the chunk's columns must have left the block above (as far as it goes)
            const int need = (cb_lo + c0 + 64) <= prev_hi ? (cb_lo + c0 + 64) : (prev_hi + 1);
            while (rtk_coop_ld(&progress[b - 1]) < need) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        int my_t = (cj <= cb_hi) ? static_cast<int>(static_cast<unsigned char>(trev ? tp[n - 1 - cj] : tp[cj])) : 0;
        int my_c = (b != 0 && cj <= prev_hi) ? static_cast<int>(carry[cj]) : above_h;
        asm volatile("" : "+v"(my_t), "+v"(my_c));
        const int lim = (steps - c0) < 64 ? (steps - c0) : 64;
        if (plain && track) { // the block that holds the last query row of an unbanded pass: its bottom-row scores are parked 64 columns at a time
            for (int j = 0; j < lim; ++j) {
                const int s = c0 + j;
                const unsigned in_t = static_cast<unsigned>(__builtin_amdgcn_readlane(my_t, j));
                const int in_c = __builtin_amdgcn_readlane(my_c, j);
                const unsigned mine = static_cast<unsigned>(hout_prev + 1) | (tc_prev << 8);
                const unsigned got = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(in_c + 1) | (in_t << 8)), static_cast<int>(mine), 0x138, 0xF, 0xF, false));
                const int hin = static_cast<int>(got & 0xFFu) - 1;
                const unsigned tc = got >> 8;
                const unsigned sel = (tc >> 1) & 3u;
                const uint64_t Eq = (sel & 2u) ? ((sel & 1u) ? eqG : eqT) : ((sel & 1u) ? eqC : eqA);
                uint64_t nPv = Pv, nMv = Mv, Ph, Mh;
                const int hout = rtk_myers_step(nPv, nMv, Eq, hin, bit, Ph, Mh);
                const int lc = s - lane; // column index inside the block's range
                const bool active = has_word && lc >= 0 && lc < nc;
                Pv = active ? nPv : Pv; Mv = active ? nMv : Mv; hout_prev = active ? hout : hout_prev;
                score += (active && is_block_tail) ? hout : 0;
                tc_prev = tc;
                const int tcol = s - (nw - 1);
                if (tcol >= 0 && tcol < n) { // uniform
                    const int sv = __builtin_amdgcn_readlane(score, nw - 1);
                    sbuf = (lane == (tcol & 63)) ? sv : sbuf;
                    if ((tcol & 63) == 63 || tcol == n - 1) { const int cc = (tcol & ~63) + lane; if (cc <= tcol) colscore[cc] = sbuf; }
                }
            }
        } else
        if (plain) { // the common case, predicated instead of branched (scalar branches cost an instruction-fetch restart)
            for (int j = 0; j < lim; ++j) {
                const int s = c0 + j;
                const unsigned in_t = static_cast<unsigned>(__builtin_amdgcn_readlane(my_t, j));
                const int in_c = __builtin_amdgcn_readlane(my_c, j);
                const unsigned mine = static_cast<unsigned>(hout_prev + 1) | (tc_prev << 8);
                const unsigned got = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(in_c + 1) | (in_t << 8)), static_cast<int>(mine), 0x138, 0xF, 0xF, false));
                const int hin = static_cast<int>(got & 0xFFu) - 1;
                const unsigned tc = got >> 8;
                const unsigned sel = (tc >> 1) & 3u; // 'A' -> 0, 'C' -> 1, 'T' -> 2, 'G' -> 3
                const uint64_t Eq = (sel & 2u) ? ((sel & 1u) ? eqG : eqT) : ((sel & 1u) ? eqC : eqA);
                uint64_t nPv = Pv, nMv = Mv, Ph, Mh;
                const int hout = rtk_myers_step(nPv, nMv, Eq, hin, bit, Ph, Mh);
                const int lc = s - lane;
                const bool active = has_word && lc >= 0 && lc < nc;
                Pv = active ? nPv : Pv; Mv = active ? nMv : Mv; hout_prev = active ? hout : hout_prev;
                if (active && is_block_tail && !last_block) carry[cb_lo + lc] = static_cast<int8_t>(hout);
                tc_prev = tc;
            }
        } else
        for (int j = 0; j < lim; ++j) {
            const int s = c0 + j;
            const unsigned in_t = static_cast<unsigned>(__builtin_amdgcn_readlane(my_t, j));
            const int in_c = __builtin_amdgcn_readlane(my_c, j);
            const unsigned mine = static_cast<unsigned>(hout_prev + 1) | (tc_prev << 8);
            const unsigned got = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(in_c + 1) | (in_t << 8)), static_cast<int>(mine), 0x138, 0xF, 0xF, false));
            const int hin = static_cast<int>(got & 0xFFu) - 1;
            const unsigned tc = got >> 8;
            const int lc = s - lane;
            const bool active = has_word && lc >= 0 && lc < nc;
            if (active) {
                uint64_t Eq;
                if (tc == 'A') Eq = eqA; else if (tc == 'C') Eq = eqC; else if (tc == 'G') Eq = eqG; else if (tc == 'T') Eq = eqT;
                else {
                    Eq = 0; const int lim2 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                    for (int i = 0; i < lim2; ++i) Eq |= static_cast<uint64_t>(rtk_chars_equal(rtk_job_qc(qp, m, qrev, 64 * w + i), static_cast<unsigned char>(tc), iupac)) << i;
                }
                uint64_t Ph, Mh;
                const int hout = rtk_myers_step(Pv, Mv, Eq, hin, bit, Ph, Mh);
                if (is_block_tail) { if (!last_block) carry[cb_lo + lc] = static_cast<int8_t>(hout); else score += hout; }
                hout_prev = hout;
            }
            tc_prev = tc;
            if (track) {
                const int tcol = s - (nw - 1);
                if (tcol >= 0 && tcol < n) {
                    const int sv = __builtin_amdgcn_readlane(score, nw - 1);
                    if (lane == (tcol & 63)) sbuf = sv;
                    if ((tcol & 63) == 63 || tcol == n - 1) { const int cc = (tcol & ~63) + lane; if (cc <= tcol) colscore[cc] = sbuf; }
                }
            }
        }
        if (!last_block && progress) { // columns whose delta has left this block: the tail lane is nw - 1 columns behind lane 0
            int done = c0 + lim - (nw - 1); done = done < 0 ? 0 : (done > nc ? nc : done);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            rtk_coop_st(&progress[b], cb_lo + done);
        }
    }
    if (fin_pv && has_word) { fin_pv[w] = Pv; fin_mv[w] = Mv; }
    if (progress) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (!last_block) rtk_coop_st(&progress[b], n); }
    else rtk_sync();
}

// ------------------------------------------------------------------------------------------------ banded pass on one wave: the ring
// NW pass (top row +1 per column) over a band of at most RTK_RING_MAX_BAND diagonals, any number of query words. Word w owns lane w & 63
// and is swept over its own columns [64 w + dlo, 64 w + 63 + dhi] (clipped to the matrix) at step column + w, as in the anti-diagonal
// pipeline of the row blocks; the words above and below it at that moment sit in the neighbouring lanes (the ring closes from lane 63 to
// lane 0: DPP wave_ror:1). The word at the top of the band (the "head": nothing above it any more) takes +1 per column from above and the
// target character of its column from the chunk register; every other word takes both from the lane above. A word that is past its last
// column parks its vertical deltas in fin_pv / fin_mv and its lane moves on to word + 64, whose first column lies at least 65 steps later
// because the band is narrower than 4096 - 65 - 31 diagonals: the switch is done lazily, once per chunk of 64 steps. J.peq4: 4 words
// (A, C, G, T profile) per query word, filled here.
template <int PLAIN>
__device__ __forceinline__ void rtk_myers_ring_sweep(const RtkCoopJob& J) {
    const char* __restrict__ const qp = rtk_u(J.qp); const char* __restrict__ const tp = rtk_u(J.tp);
    const int m = rtk_u(J.m), n = rtk_u(J.n), qrev = rtk_u(J.qrev), trev = rtk_u(J.trev); const bool iupac = rtk_u(J.iupac) != 0;
    const int dlo = rtk_u(J.dlo), dhi = rtk_u(J.dhi);
    uint64_t* const fin_pv = rtk_u(J.fin_pv); uint64_t* const fin_mv = rtk_u(J.fin_mv);
    const uint64_t* __restrict__ const peq4 = rtk_u(J.peq4);
    const int W = (m + 63) >> 6, last_bit = (m - 1) & 63;
    const int Wa = rtk_u(rtk_band_words(m, n, dlo));
    const int lane = rtk_lane();
    // this lane's current word
    int w = lane; bool live = w < Wa;
    uint64_t eqA = 0, eqC = 0, eqG = 0, eqT = 0, Pv = ~0ull, Mv = 0ull;
    int clo_w = 0x7fffffff, chi_w = -1, head_from = 0x7fffffff, bit = 63;
#define RTK_RING_LOAD_WORD()                                                                                                                  \
    {                                                                                                                                        \
        eqA = peq4[4ull * w]; eqC = peq4[4ull * w + 1]; eqG = peq4[4ull * w + 2]; eqT = peq4[4ull * w + 3];                                   \
        Pv = ~0ull; Mv = 0ull;                                                                                                               \
        clo_w = rtk_band_clo(w, dlo, 1); chi_w = rtk_band_chi(w, W, n, dhi, 1);                                                               \
        head_from = (w == 0) ? -0x7fffffff : 64 * w + dhi; /* columns at which no word is above this one (dhi < 2^29: no overflow) */       \
        bit = (w == W - 1) ? last_bit : 63;                                                                                                  \
    }
    if (live) RTK_RING_LOAD_WORD()
    int hout_prev = 0; unsigned tc_prev = 0;
    const int steps = rtk_u(n + Wa - 1);
    int s = 0, w_head = 0, head_chi = rtk_u(rtk_band_chi(0, W, n, dhi, 1)), cb = 0;
    int nxt_t = 0;
    { const int cn = lane; if (cn < n) nxt_t = static_cast<int>(static_cast<unsigned char>(trev ? tp[n - 1 - cn] : tp[cn])); }
    int my_t = nxt_t;
    { const int cn = 64 + lane; nxt_t = 0; if (cn < n) nxt_t = static_cast<int>(static_cast<unsigned char>(trev ? tp[n - 1 - cn] : tp[cn])); }
    // The sweep is cut into runs in which nothing wave-level changes: the head of the band stays the same word and its columns stay in
    // the chunk register, so the run is a counted loop with scalar control (column of the head = c0 + j).
    while (s < steps) {
        s = rtk_u(s); w_head = rtk_u(w_head); head_chi = rtk_u(head_chi); cb = rtk_u(cb);
        { // words that are past their last column: park the deltas, take the next word of this lane
            const bool done = live && (s - w > chi_w);
            if (rtk_ballot(done) != 0ull) {
                if (done) {
                    if (fin_pv) { fin_pv[w] = Pv; fin_mv[w] = Mv; }
                    w += 64; live = w < Wa;
                    if (live) RTK_RING_LOAD_WORD() else { clo_w = 0x7fffffff; chi_w = -1; }
                }
            }
        }
        asm volatile("" : "+v"(my_t));
        const int c0 = s - w_head; // column of the head at step s
        int run = steps - s;
        { const int r1 = head_chi - c0 + 1, r2 = cb + 64 - c0; run = run < r1 ? run : r1; run = run < r2 ? run : r2; run = run < 64 ? run : 64; run = rtk_u(run); }
        const int ci = c0 - cb;
        for (int j = 0; j < run; ++j) {
            const unsigned in_t = static_cast<unsigned>(__builtin_amdgcn_readlane(my_t, ci + j));
            const unsigned mine = static_cast<unsigned>(hout_prev + 1) | (tc_prev << 8);
            unsigned got = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mine), 0x13C, 0xF, 0xF, false)); // wave_ror:1
            const int c = s + j - w;
            got = (c >= head_from) ? (2u | (in_t << 8)) : got;
            const int hin = static_cast<int>(got & 0xFFu) - 1;
            const unsigned tc = got >> 8;
            const bool active = c >= clo_w && c <= chi_w;
            if (PLAIN) {
                const unsigned sel = (tc >> 1) & 3u; // 'A' -> 0, 'C' -> 1, 'T' -> 2, 'G' -> 3
                const uint64_t Eq = (sel & 2u) ? ((sel & 1u) ? eqG : eqT) : ((sel & 1u) ? eqC : eqA);
                uint64_t nPv = Pv, nMv = Mv, Ph, Mh;
                const int hout = rtk_myers_step(nPv, nMv, Eq, hin, bit, Ph, Mh);
                Pv = active ? nPv : Pv; Mv = active ? nMv : Mv; hout_prev = active ? hout : hout_prev;
            } else if (active) {
                uint64_t Eq;
                if (tc == 'A') Eq = eqA; else if (tc == 'C') Eq = eqC; else if (tc == 'G') Eq = eqG; else if (tc == 'T') Eq = eqT;
                else { // rare: IUPAC code, N or foreign byte in the target -> compare the 64 query characters of this word directly
                    Eq = 0; const int lim2 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                    for (int i = 0; i < lim2; ++i) Eq |= static_cast<uint64_t>(rtk_chars_equal(rtk_job_qc(qp, m, qrev, 64 * w + i), static_cast<unsigned char>(tc), iupac)) << i;
                }
                uint64_t Ph, Mh;
                hout_prev = rtk_myers_step(Pv, Mv, Eq, hin, bit, Ph, Mh);
            }
            tc_prev = tc;
        }
        s += run;
        if (s - w_head > head_chi && w_head + 1 < Wa) { ++w_head; head_chi = rtk_band_chi(w_head, W, n, dhi, 1); } // the head of the band moves down (one step without a new column)
        if (s - w_head >= cb + 64) {
            cb += 64; my_t = nxt_t;
            const int cn = cb + 64 + lane; nxt_t = 0; if (cn < n) nxt_t = static_cast<int>(static_cast<unsigned char>(trev ? tp[n - 1 - cn] : tp[cn]));
        }
    }
#undef RTK_RING_LOAD_WORD
    if (live && fin_pv) { fin_pv[w] = Pv; fin_mv[w] = Mv; }
}
__device__ __noinline__ void rtk_myers_ring(const RtkCoopJob& J) {
    const char* __restrict__ const qp = rtk_u(J.qp); const char* __restrict__ const tp = rtk_u(J.tp);
    const int m = rtk_u(J.m), n = rtk_u(J.n), qrev = rtk_u(J.qrev), trev = rtk_u(J.trev); const bool iupac = rtk_u(J.iupac) != 0;
    uint64_t* const peq4 = rtk_u(J.peq4);
    const int Wa = rtk_band_words(m, n, rtk_u(J.dlo));
    const int lane = rtk_lane();
    for (int w = lane; w < Wa; w += RTK_WAVE) { // the profile of every word that gets a column
        uint64_t a, c, g, t; rtk_myers_eq4(qp, m, qrev, w, iupac, a, c, g, t);
        peq4[4ull * w] = a; peq4[4ull * w + 1] = c; peq4[4ull * w + 2] = g; peq4[4ull * w + 3] = t;
    }
    bool plain = true;
    for (int c0 = 0; c0 < n && plain; c0 += 64) {
        const int cj = c0 + lane; bool okc = true;
        if (cj < n) { const unsigned char ch = static_cast<unsigned char>(trev ? tp[n - 1 - cj] : tp[cj]); okc = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
        plain = (rtk_ballot(!okc) == 0ull);
    }
    rtk_sync(); // the profile words are read by other lanes than wrote them
    if (plain) rtk_myers_ring_sweep<1>(J); else rtk_myers_ring_sweep<0>(J);
    rtk_sync();
}
// work items of a pass: one for the ring, else its row blocks that have a column
__device__ __forceinline__ bool rtk_pass_is_ring(int m, const RtkBand& band) { return rtk_band_gran(m, band.dlo, band.dhi) == 1; }
__device__ __forceinline__ int rtk_pass_items(int m, int n, const RtkBand& band) { return rtk_pass_is_ring(m, band) ? 1 : ((rtk_band_words(m, n, band.dlo) + 63) >> 6); }
__device__ __forceinline__ RtkCoopJob rtk_make_job(const MyersScratch& sc, const MySeq& q, const MySeq& t, int top_h, bool iupac, const RtkBand& band, uint64_t* fin_pv, uint64_t* fin_mv) {
    RtkCoopJob j; j.qp = q.p; j.tp = t.p; j.m = q.n; j.n = t.n; j.qrev = q.rev; j.trev = t.rev; j.top_h = top_h; j.iupac = iupac ? 1 : 0;
    j.fin_pv = fin_pv; j.fin_mv = fin_mv; j.carry = rtk_u(sc.carry); j.colscore = rtk_u(sc.colscore);
    j.dlo = band.dlo; j.dhi = band.dhi; j.ring = rtk_pass_is_ring(q.n, band) ? 1 : 0; j.peq4 = rtk_u(sc.peq);
    return j;
}
#endif

#if defined(RTK_MULTIWAVE) && !defined(RTK_SIM)
// ------------------------------------------------------------------------------------------------ one pass on several waves
// The row blocks (64 words = 4096 query rows each) of ONE pass run on the waves of the workgroup at the same time: block b consumes
// the horizontal deltas block b-1 leaves at its bottom row (`carry`, one byte per column, in memory) one 64-column chunk behind it, so
// a pass over B blocks takes about n + 128 B steps instead of B n. The waves of a workgroup sit on one CU and share its L1, so
// workgroup-scope fences (no cache maintenance) order the carries in memory against the progress counters in LDS. The program wave (wave 0) publishes the pass in an LDS mailbox, the
// helper waves of the workgroup pick it up, everybody takes the blocks b = wave, wave + NW, ... in ascending order (a block only ever
// waits for a lower-numbered one, and those are started first: no cycle), wave 0 continues when all helpers have reported.
#define RTK_COOP_MAXB 512 // row blocks of all the passes of a round
#define RTK_COOP_MAXJ 64  // passes of a round (two per Hirschberg sub-problem)
struct RtkCoop { RtkCoopJob job[RTK_COOP_MAXJ]; int first[RTK_COOP_MAXJ + 1]; int node[RTK_COOP_MAXJ / 2][8]; int n_jobs, n_items, seq, n_done, exit_flag, n_waves, next_item; int progress[RTK_COOP_MAXB];
                 // a round of leaf tracebacks (rtk_myers_alignment_bfs): every wave of the workgroup takes leaves, each with its own traceback table
                 int n_gangs; RtkGangCtx gctx; // gang sweeps of a round (items 0 .. n_gangs - 1; the row blocks of the jobs follow)
                 int n_leaf_waves; // waves that own a work area for leaf tracebacks (the others sit a leaf round out)
                 int leaf_mode, liupac; const char* lq; const char* lt; int32_t* llist; uint8_t* lmoves; MyersScratch* lsc; uint32_t lnm[16]; const uint64_t* lneed; };
__device__ __noinline__ void rtk_myers_leaf_item(RtkCoop* st, int wave, int x);
__device__ __noinline__ void rtk_myers_gang(const RtkGangCtx& C_, int gang_);
__device__ __forceinline__ RtkCoop* rtk_coop() { __shared__ RtkCoop st; return &st; }

__device__ __forceinline__ void rtk_myers_coop_run(RtkCoop* st, int wave) {
    // A round is a list of passes; pass j owns the row blocks first[j] .. first[j + 1) of the round's block list. Every wave takes its
    // indices in ascending order and a block only waits for index i - 1 (the block above it in its own pass): the lowest unfinished
    // index can always run, so nobody waits for ever.
    // Items are claimed in ascending order from a shared counter (their durations differ: a ring pass is a whole pass, a row block a slice of
    // one): whoever holds item i - 1 is running, so the wait of a block for the one above it always ends.
    const int nit = rtk_coop_ld(&st->n_items);
    const bool leaves = rtk_coop_ld(&st->leaf_mode) != 0;
    if (leaves && wave >= rtk_coop_ld(&st->n_leaf_waves)) return; // no work area for a traceback table
    const int ng = leaves ? 0 : rtk_coop_ld(&st->n_gangs);
    int j = 0;
    for (;;) {
        int i = 0;
        if (rtk_lane() == 0) i = __hip_atomic_fetch_add(&st->next_item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= nit) break;
        if (leaves) { rtk_myers_leaf_item(st, wave, i); continue; }
        if (i < ng) { rtk_myers_gang(st->gctx, i); continue; }
        i -= ng;
        while (i >= rtk_coop_ld(&st->first[j + 1])) ++j;
        const int f = rtk_coop_ld(&st->first[j]);
        if (rtk_coop_ld(&st->job[j].ring)) rtk_myers_ring(st->job[j]); // a banded pass on one wave
        else rtk_myers_block(st->job[j], i - f, st->progress + f);
    }
}

// helper waves of the workgroup: wait for passes until the program wave says it is done
__device__ __forceinline__ void rtk_myers_coop_helper(int wave) {
    RtkCoop* st = rtk_coop();
    int seen = 0;
    while (true) {
        int sq;
        while ((sq = rtk_coop_ld(&st->seq)) == seen) { if (rtk_coop_ld(&st->exit_flag)) return; __builtin_amdgcn_s_sleep(32); }
        seen = sq;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        rtk_myers_coop_run(st, wave);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (rtk_lane() == 0) __hip_atomic_fetch_add(&st->n_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// program wave: publish one pass -- or two passes of the same query length (q2 != nullptr: the second one works on the columns behind
// the first one's in the carry / score arrays) --, take part, wait for the helpers. Returns false when not worth sharing (caller runs them alone).
__device__ __forceinline__ bool rtk_myers_pass_coop(const MyersScratch& sc, const MySeq& q, const MySeq& t, int top_h, bool iupac, const RtkBand& band, uint64_t* fin_pv, uint64_t* fin_mv,
                                                    const MySeq* q2 = nullptr, const MySeq* t2 = nullptr, uint64_t* fin_pv2 = nullptr, uint64_t* fin_mv2 = nullptr) {
    RtkCoop* st = rtk_coop();
    const int nwv = rtk_coop_ld(&st->n_waves);
    const int W = (q.n + 63) >> 6, nj = q2 ? 2 : 1;
    const int it1 = rtk_pass_items(q.n, t.n, band), it2 = q2 ? rtk_pass_items(q2->n, t2->n, band) : 0;
    if (nwv < 2 || it1 + it2 < 2 || it1 + it2 > RTK_COOP_MAXB || (q2 && q2->n != q.n)) return false;
    if (static_cast<uint64_t>(8 * W + 8) > 15ull * sc.w_cap) return false; // two profile tables
    if (rtk_lane() == 0) {
        st->job[0] = rtk_make_job(sc, q, t, top_h, iupac, band, fin_pv, fin_mv);
        if (q2) { RtkCoopJob j = rtk_make_job(sc, *q2, *t2, top_h, iupac, band, fin_pv2, fin_mv2); j.carry += t.n; j.colscore += t.n; j.peq4 += 4ull * W + 4; st->job[1] = j; }
        st->n_jobs = nj; st->n_items = it1 + it2; st->first[0] = 0; st->first[1] = it1; st->first[2] = it1 + it2;
    }
    for (int i = rtk_lane(); i < it1 + it2; i += RTK_WAVE) st->progress[i] = 0;
    if (rtk_lane() == 0) { st->n_done = 0; st->next_item = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the strings of the pass (global memory) and the mailbox
    rtk_coop_st(&st->seq, rtk_coop_ld(&st->seq) + 1);
    rtk_myers_coop_run(st, 0);
    while (rtk_coop_ld(&st->n_done) < nwv - 1) __builtin_amdgcn_s_sleep(8);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

// program wave: the passes in st->job[0 .. n_jobs) with their block ranges st->first[] are ready: run them with the helpers
__device__ __forceinline__ void rtk_myers_round_coop(RtkCoop* st, int n_items) {
    const int nwv = rtk_coop_ld(&st->n_waves);
    for (int i = rtk_lane(); i < n_items; i += RTK_WAVE) st->progress[i] = 0;
    if (rtk_lane() == 0) { st->n_done = 0; st->next_item = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    rtk_coop_st(&st->seq, rtk_coop_ld(&st->seq) + 1);
    rtk_myers_coop_run(st, 0);
    while (rtk_coop_ld(&st->n_done) < nwv - 1) __builtin_amdgcn_s_sleep(8);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#endif

// Full pass of query q over target t. Writes colscore[j] = D[m][j+1] for every column; optionally the traceback
// table (store != 0) and the final vertical delta vectors (fin_pv/fin_mv, W words each) for column extraction.
// top_h: +1 NW/SHW, 0 HW (edlib.cpp:584).
RTK_FN void rtk_myers_pass(const MyersScratch& sc_, MySeq q_, MySeq t_, int top_h_, bool iupac_, int store_, uint64_t* fin_pv_, uint64_t* fin_mv_) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc);
    MySeq q, t; q.p = rtk_u(q_.p); q.n = rtk_u(q_.n); q.rev = rtk_u(q_.rev); t.p = rtk_u(t_.p); t.n = rtk_u(t_.n); t.rev = rtk_u(t_.rev);
    const int top_h = rtk_u(top_h_), store = rtk_u(store_); const bool iupac = rtk_u(iupac_); uint64_t* fin_pv = rtk_u(fin_pv_); uint64_t* fin_mv = rtk_u(fin_mv_);
    const int m = q.n, n = t.n, W = (m + 63) >> 6, last_bit = (m - 1) & 63;
#ifdef RTK_SIM
    rtk_myers_build_peq(sc, q, W, iupac);
    int score = m;
    // the simulator walks the matrix column by column; per-word state lives in fin arrays or a local buffer
    uint64_t* Pv = fin_pv; uint64_t* Mv = fin_mv;
    uint64_t lpv[64], lmv[64]; // only used when the caller does not want the final vectors and W <= 64
    uint64_t* heapPv = nullptr; uint64_t* heapMv = nullptr;
    if (!Pv) { if (W <= 64) { Pv = lpv; Mv = lmv; } else { heapPv = new uint64_t[W]; heapMv = new uint64_t[W]; Pv = heapPv; Mv = heapMv; } }
    for (int w = 0; w < W; ++w) { Pv[w] = ~0ull; Mv[w] = 0; }
    for (int j = 0; j < n; ++j) {
        const unsigned char tc = rtk_seq_at(t, j);
        int hin = top_h;
        for (int w = 0; w < W; ++w) {
            uint64_t Ph, Mh;
            const uint64_t Eq = rtk_myers_eq_word(sc, q, W, w, tc);
            hin = rtk_myers_step(Pv[w], Mv[w], Eq, hin, (w == W - 1) ? last_bit : 63, Ph, Mh);
            if (store) { rtk_tb_put(sc.tb + 4ull * RTK_TB(j, w, n), Pv[w], Mv[w], Ph, Mh); }
        }
        score += hin;
        sc.colscore[j] = score;
    }
    delete[] heapPv; delete[] heapMv;
#else
    const int lane = rtk_lane();
#ifdef RTK_MULTIWAVE
    if (!store && rtk_myers_pass_coop(sc, q, t, top_h, iupac, rtk_band_full(), fin_pv, fin_mv)) { rtk_sync(); return; } // several row blocks: shared with the helper waves of the workgroup
#endif
    if (!store && W > 64) { // several row blocks on this wave: the block sweep without branches in the step
        const RtkCoopJob j = rtk_make_job(sc, q, t, top_h, iupac, rtk_band_full(), fin_pv, fin_mv);
        for (int b = 0; b < ((W + 63) >> 6); ++b) rtk_myers_block(j, b, nullptr);
        rtk_sync();
        return;
    }
    // local copies: the scratch descriptor lives in private memory and its fields could alias the stores below,
    // which would force a (slow) reload of every pointer on every step
    int8_t* __restrict__ const carry = rtk_u(sc.carry); int32_t* __restrict__ const colscore = rtk_u(sc.colscore); uint64_t* __restrict__ const tb = rtk_u(sc.tb);
    const char* __restrict__ const tp = t.p; const int trev = t.rev;
    const char* __restrict__ const qp = q.p; const int qrev = q.rev;
    for (int w0 = 0; w0 < W; w0 += 64) {
        const int nw = (W - w0) < 64 ? (W - w0) : 64;
        const int w = w0 + lane;
        const bool has_word = lane < nw;
        const bool is_last_word = has_word && (w == W - 1);
        const bool is_block_tail = (lane == nw - 1);
        const int bit = is_last_word ? last_bit : 63;
        // this lane's profile words for A, C, G, T stay in registers for the whole pass; other classes are fetched on demand
        uint64_t eqA = 0, eqC = 0, eqG = 0, eqT = 0;
        if (has_word) { // built straight from the query characters of this word (no profile table in memory on the device)
            const int lim = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
            for (int i = 0; i < lim; ++i) {
                const unsigned char qc = static_cast<unsigned char>(qrev ? qp[m - 1 - (64 * w + i)] : qp[64 * w + i]);
                const uint32_t bm = rtk_eq_classes(rtk_cls(qc), iupac) & 0xFu; // bases this query character equals
                eqA |= static_cast<uint64_t>(bm & 1u) << i; eqC |= static_cast<uint64_t>((bm >> 1) & 1u) << i;
                eqG |= static_cast<uint64_t>((bm >> 2) & 1u) << i; eqT |= static_cast<uint64_t>((bm >> 3) & 1u) << i;
            }
        }
        uint64_t Pv = ~0ull, Mv = 0ull;
        if (W <= 64) { // single row block: is the target plain A/C/G/T? then take the branch-free sweep
            bool plain = true;
            for (int c0 = 0; c0 < n && plain; c0 += 64) {
                const int cj = c0 + lane;
                bool okc = true;
                if (cj < n) { const unsigned char ch = static_cast<unsigned char>(trev ? tp[n - 1 - cj] : tp[cj]); okc = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
                plain = (rtk_ballot(!okc) == 0ull);
            }
            if (plain) {
                if (store) rtk_myers_sweep_acgt<1>(m, n, W, top_h, last_bit, tp, trev, eqA, eqC, eqG, eqT, colscore, tb, Pv, Mv);
                else rtk_myers_sweep_acgt<0>(m, n, W, top_h, last_bit, tp, trev, eqA, eqC, eqG, eqT, colscore, tb, Pv, Mv);
                if (fin_pv && has_word) { fin_pv[w] = Pv; fin_mv[w] = Mv; }
                rtk_sync();
                continue;
            }
        }
        int hout_prev = 0; unsigned tc_prev = 0;
        int score = m;
        const int steps = n + nw - 1;
        // The target character of a column enters at lane 0 and then rides down the lanes together with the horizontal
        // delta (one packed __shfl_up per step): no memory access on the step-to-step dependency chain.
        const bool last_block = (w0 + nw >= W);
        int sbuf = 0;       // last-row scores of up to 64 columns, one per lane (lane = column & 63), flushed with one coalesced store
        for (int c0 = 0; c0 < steps; c0 += 64) {
            const int cj = c0 + lane;
            int my_t = (cj < n) ? static_cast<int>(static_cast<unsigned char>(trev ? tp[n - 1 - cj] : tp[cj])) : 0; // lane j holds t[c0 + j] ...
            int my_c = (w0 != 0 && cj < n) ? static_cast<int>(carry[cj]) : top_h; // ... and the delta entering row block w0
            asm volatile("" : "+v"(my_t), "+v"(my_c)); // the loads are complete here, so the step loop below carries no memory wait
            const int lim = (steps - c0) < 64 ? (steps - c0) : 64;
            for (int j = 0; j < lim; ++j) {
                const int s = c0 + j;
                const unsigned in_t = static_cast<unsigned>(__builtin_amdgcn_readlane(my_t, j));
                const int in_c = __builtin_amdgcn_readlane(my_c, j);
                const unsigned mine = static_cast<unsigned>(hout_prev + 1) | (tc_prev << 8);
                // wave_shr:1 (DPP): lane l receives lane l-1's value, lane 0 keeps `old` = the values entering the block
                const unsigned got = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(in_c + 1) | (in_t << 8)), static_cast<int>(mine), 0x138, 0xF, 0xF, false));
                const int hin = static_cast<int>(got & 0xFFu) - 1;
                const unsigned tc = got >> 8;
                const int col = s - lane;
                const bool active = has_word && col >= 0 && col < n;
                if (active) {
                    uint64_t Eq;
                    if (tc == 'A') Eq = eqA; else if (tc == 'C') Eq = eqC; else if (tc == 'G') Eq = eqG; else if (tc == 'T') Eq = eqT;
                    else { // rare: IUPAC code, N or foreign byte in the target -> compare the 64 query characters of this word directly
                        Eq = 0; const int lim2 = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
                        for (int i = 0; i < lim2; ++i) Eq |= static_cast<uint64_t>(rtk_chars_equal(static_cast<unsigned char>(qrev ? qp[m - 1 - (64 * w + i)] : qp[64 * w + i]), static_cast<unsigned char>(tc), iupac)) << i;
                    }
                    uint64_t Ph, Mh;
                    const int hout = rtk_myers_step(Pv, Mv, Eq, hin, bit, Ph, Mh);
                    if (store) { rtk_tb_put(tb + 4ull * RTK_TB(col, w, n), Pv, Mv, Ph, Mh); }
                    if (is_block_tail) { if (!last_block) carry[col] = static_cast<int8_t>(hout); else score += hout; }
                    hout_prev = hout;
                }
                tc_prev = tc;
                if (last_block) { // park the tail lane's score of column (s - nw + 1) in lane (column & 63); store 64 of them at once
                    const int tcol = s - (nw - 1);
                    if (tcol >= 0 && tcol < n) {
                        const int sv = __builtin_amdgcn_readlane(score, nw - 1);
                        if (lane == (tcol & 63)) sbuf = sv;
                        if ((tcol & 63) == 63 || tcol == n - 1) { const int cc = (tcol & ~63) + lane; if (cc <= tcol) colscore[cc] = sbuf; }
                    }
                }
            }
        }
        if (fin_pv && has_word) { fin_pv[w] = Pv; fin_mv[w] = Mv; }
        rtk_sync();
    }
#endif
    rtk_sync();
}

RTK_FN void rtk_myers_column(const uint64_t* fin_pv_, const uint64_t* fin_mv_, int m_, int n_, int32_t* out_, int dlo_ = -RTK_BAND_FULL, int dhi_ = RTK_BAND_FULL, int gran_ = 64);
// NW pass (top row +1 per column) over the band [dlo, dhi] of diagonals only. Leaves the vertical deltas of every word at the word's last
// column in fin_pv / fin_mv (both required) and returns the granularity of the schedule it took (1: ring, 64: row blocks), which
// rtk_myers_column needs to tell the rows that are inside the band at the last column from the others.
RTK_FN int rtk_myers_pass_banded(const MyersScratch& sc_, MySeq q_, MySeq t_, bool iupac_, int dlo_, int dhi_, uint64_t* fin_pv_, uint64_t* fin_mv_) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc);
    MySeq q, t; q.p = rtk_u(q_.p); q.n = rtk_u(q_.n); q.rev = rtk_u(q_.rev); t.p = rtk_u(t_.p); t.n = rtk_u(t_.n); t.rev = rtk_u(t_.rev);
    const bool iupac = rtk_u(iupac_); uint64_t* fin_pv = rtk_u(fin_pv_); uint64_t* fin_mv = rtk_u(fin_mv_);
    RtkBand band; band.dlo = rtk_u(dlo_); band.dhi = rtk_u(dhi_);
    const int m = q.n, n = t.n, W = (m + 63) >> 6;
    const int gran = rtk_band_gran(m, band.dlo, band.dhi);
#ifdef RTK_SIM
    // the simulator walks the band column by column with the same column ranges per word as the device schedules
    const int last_bit = (m - 1) & 63, Wa = rtk_band_words(m, n, band.dlo);
    rtk_myers_build_peq(sc, q, W, iupac);
    for (int w = 0; w < Wa; ++w) { fin_pv[w] = ~0ull; fin_mv[w] = 0; }
    for (int j = 0; j < n; ++j) {
        const unsigned char tc = rtk_seq_at(t, j);
        int hin = 1;
        for (int w = 0; w < Wa; ++w) {
            if (j < rtk_band_clo(w, band.dlo, gran) || j > rtk_band_chi(w, W, n, band.dhi, gran)) continue;
            if (w > 0 && j > rtk_band_chi(w - 1, W, n, band.dhi, gran)) hin = 1; // nothing above any more: the row above the band grows by one per column
            uint64_t Ph, Mh;
            const uint64_t Eq = rtk_myers_eq_word(sc, q, W, w, tc);
            hin = rtk_myers_step(fin_pv[w], fin_mv[w], Eq, hin, (w == W - 1) ? last_bit : 63, Ph, Mh);
        }
    }
#else
#ifdef RTK_MULTIWAVE
    if (rtk_myers_pass_coop(sc, q, t, 1, iupac, band, fin_pv, fin_mv)) { rtk_sync(); return gran; }
#endif
    const RtkCoopJob j = rtk_make_job(sc, q, t, 1, iupac, band, fin_pv, fin_mv);
    if (gran == 1) rtk_myers_ring(j);
    else { const int nb = (rtk_band_words(m, n, band.dlo) + 63) >> 6; for (int b = 0; b < nb; ++b) rtk_myers_block(j, b, nullptr); }
    rtk_sync();
#endif
    return gran;
}

struct MyersResult { int32_t dist, first, last, nloc; };

// edlibAlign(..., TASK_DISTANCE): edit distance (or -1 if above a non-negative k), first and largest end location and
// their number. SHW/HW report target position -1 (score m) when m % 64 != 0, like edlib's padded last block
// (edlib.cpp:658-692). Optionally lists every end location (locs_out, up to cap).
RTK_FN MyersResult rtk_myers_distance(const MyersScratch& sc_, const char* q_, int m_, const char* t_, int n_, int k_, int mode_, bool iupac_,
                                        int32_t* locs_out_ = nullptr, int cap_ = 0) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); const char* q = rtk_u(q_); const char* t = rtk_u(t_); const int m = rtk_u(m_), n = rtk_u(n_), k = rtk_u(k_), mode = rtk_u(mode_), cap = rtk_u(cap_);
    const bool iupac = rtk_u(iupac_); int32_t* locs_out = rtk_u(locs_out_);
    MyersResult r; r.dist = -1; r.first = -1; r.last = -1; r.nloc = 0;
    if (m == 0 || n == 0) { // edlib.cpp:161-179
        if (mode == RTK_MODE_NW) { r.dist = m > n ? m : n; r.first = r.last = n - 1; }
        else { r.dist = m; r.first = r.last = -1; }
        r.nloc = 1;
        if (locs_out && cap > 0) locs_out[0] = r.first;
        return r;
    }
    if (mode == RTK_MODE_NW && k >= 0 && k < (n > m ? n - m : m - n)) return r; // edlib.cpp:744-747
    if (static_cast<uint32_t>((m + 63) >> 6) > sc.w_cap || static_cast<uint32_t>(n) > sc.t_cap) { *sc.overflow = 1; return r; }
#ifndef RTK_SIM
    if (m <= 4096 && !locs_out) {
        const SweepStat st = (mode == RTK_MODE_NW) ? rtk_myers_fast_any<0, 0>(q, m, t, n, 1, iupac, nullptr) : rtk_myers_fast_any<0, 1>(q, m, t, n, mode == RTK_MODE_HW ? 0 : 1, iupac, nullptr);
        if (st.plain) {
            if (mode == RTK_MODE_NW) { if (k >= 0 && st.final_score > k) return r; r.dist = st.final_score; r.first = r.last = n - 1; r.nloc = 1; return r; }
            int best = st.best; const bool pseudo = (m & 63) != 0;
            if (pseudo && m < best) best = m;
            if (k >= 0 && best > k) return r;
            r.dist = best;
            if (pseudo && m == best) { r.first = -1; r.last = (st.best == best) ? st.last : -1; r.nloc = 1 + ((st.best == best) ? st.cnt : 0); }
            else { r.first = st.first; r.last = st.last; r.nloc = st.cnt; }
            return r;
        }
    }
#endif
    if (mode == RTK_MODE_NW && m > 4096 && static_cast<uint32_t>(m) <= sc.r_cap && static_cast<uint64_t>(2 * ((m + 63) >> 6)) <= sc.tb_cap_words) {
        // several row blocks: only the Ukkonen band of k (edlib.cpp:744-775). Unknown distance: a guess instead of edlib's doubling (:199-212) -- a score
        // found inside a band is a real one, so when it exceeds the guess the band of that score holds the optimum and one more pass settles it.
        const int W = (m + 63) >> 6;
        uint64_t* fin = sc.tb; int32_t* col = sc.rowL;
        { MyersScratch& msc = const_cast<MyersScratch&>(sc); msc.tb_gen = rtk_ld(&msc.tb_gen) + 1u; } // the table's memory holds the delta vectors
        int kk = k >= 0 ? k : rtk_band_guess(m, n);
        for (;;) {
            const RtkBand band = rtk_band_nw(m, n, kk);
            const int gran = rtk_myers_pass_banded(sc, rtk_seq(q, m), rtk_seq(t, n), iupac, band.dlo, band.dhi, fin, fin + W);
            rtk_myers_column(fin, fin + W, m, n, col, band.dlo, band.dhi, gran);
            const int d = rtk_ld(col + (m - 1));
            if (d <= kk) { r.dist = d; r.first = r.last = n - 1; r.nloc = 1; if (locs_out && cap > 0) locs_out[0] = n - 1; return r; }
            if (k >= 0) return r; // above k
            kk = d;
        }
    }
    rtk_myers_pass(sc, rtk_seq(q, m), rtk_seq(t, n), mode == RTK_MODE_HW ? 0 : 1, iupac, 0, nullptr, nullptr);
    if (mode == RTK_MODE_NW) {
        const int d = sc.colscore[n - 1];
        if (k >= 0 && d > k) return r;
        r.dist = d; r.first = r.last = n - 1; r.nloc = 1;
        if (locs_out && cap > 0) locs_out[0] = n - 1;
        return r;
    }
    // min over columns, then positions of the minimum (lane-strided, reduced across the wave)
    int best = 0x7fffffff;
    for (int j = rtk_lane(); j < n; j += RTK_WAVE) { const int v = sc.colscore[j]; best = v < best ? v : best; }
#ifndef RTK_SIM
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(best, o, 64); best = v < best ? v : best; }
#endif
    const bool pseudo = (m & 63) != 0;
    if (pseudo && m < best) best = m;
    if (k >= 0 && best > k) return r;
    r.dist = best;
    int cnt = 0, first = 0x7fffffff, last = -2;
    if (pseudo && m == best) { cnt = 1; first = -1; last = -1; if (locs_out && cap > 0) locs_out[0] = -1; }
    for (int j0 = 0; j0 < n; j0 += RTK_WAVE) {
        const int j = j0 + rtk_lane();
        const bool hit = (j < n) && (sc.colscore[j] == best);
        const uint64_t b = rtk_ballot(hit);
        if (b) {
            const int lo = j0 + rtk_ffs(b) - 1;
            const int hi = j0 + 63 - __builtin_clzll(b);
            if (lo < first) first = lo;
            if (hi > last) last = hi;
            if (locs_out) {
                const int my = cnt + rtk_popc(b & ((1ull << rtk_lane()) - 1ull));
                if (hit && my < cap) locs_out[my] = j;
            }
            cnt += rtk_popc(b);
        }
    }
    rtk_sync();
    r.first = first; r.last = last; r.nloc = cnt;
    return r;
}

// Canonical NW traceback over the stored table, preferring up (insert) > left (delete) > diagonal
// (edlib.cpp:1021-1137). Appends the moves (already in forward order) to sc.moves at *n_moves.
// Walks the stored table from cell (m, n) back to the origin; `cur` = D[m][n]. Appends the moves to sc.moves (after *n_moves).
RTK_FN void rtk_myers_walk(const MyersScratch& sc_, int m_, int n_, int ncols_, int cur_, uint32_t* n_moves_) { // ncols: columns of the sweep that stored the table (>= n)
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); uint32_t* n_moves = rtk_u(n_moves_);
    const int m = rtk_u(m_), n = rtk_u(n_), ncols = rtk_u(ncols_);
    int cur = rtk_u(cur_);
    const unsigned long long tw0 = rtk_clock(); unsigned n_rel = 0, n_sc = 0;
    int i = m, j = n;
    uint32_t nt = 0; // moves are produced backwards into moves_tmp, from its end
    uint8_t* tmp = rtk_ld(&sc.moves_tmp);
    const uint32_t cap = rtk_ld(&sc.mv_cap);
    uint64_t* const tbp = rtk_ld(&sc.tb);
#ifndef RTK_SIM
    // The wave keeps, for the current query word, the four delta words of 64 consecutive columns in registers
    // (lane l <-> column c_hi - l); a traceback step is then scalar (v_readlane), with one table reload per ~60 moves.
    const int lane = rtk_lane();
    int w_cur = -1, c_hi = -1;
    uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#define RTK_RL64(v, l) ((static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>((v) >> 32), (l)))) << 32) | static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>((v) & 0xFFFFFFFFull), (l))))
#endif
    while (i > 0 && j > 0) {
        const int r = i - 1, c = j - 1, w = r >> 6, b = r & 63;
#ifdef RTK_SIM
        uint64_t a0, a1, a2, a3; rtk_tb_get(tbp + 4ull * RTK_TB(c, w, ncols), a0, a1, a2, a3);
        uint64_t l0 = 0, l1 = 0;
        if (c > 0) { uint64_t l2, l3; rtk_tb_get(tbp + 4ull * RTK_TB(c - 1, w, ncols), l0, l1, l2, l3); }
#else
        if (w != w_cur || c > c_hi || c_hi - c > 62) {
            c_hi = c; w_cur = w; ++n_rel;
            const int col = c - lane;
            if (col >= 0) rtk_tb_get(tbp + 4ull * RTK_TB(col, w, ncols), e0, e1, e2, e3);
        }
        const int li = c_hi - c;
        { // run of inserts (moves up column c): consecutive rows from r downwards whose vertical delta is +1 = ones of Pv & ~Mv below bit b
            const uint64_t up = RTK_RL64(e0, li) & ~RTK_RL64(e1, li);
            if ((up >> b) & 1ull) {
                const uint64_t nup = ~(up << (63 - b));
                const int run = nup ? __builtin_clzll(nup) : 64; // >= 1, stays inside this 64-row word
                if (lane < run) tmp[cap - (nt + 1u + static_cast<uint32_t>(lane))] = 1;
                i -= run; cur -= run; nt += static_cast<uint32_t>(run);
                continue;
            }
        }
        { // run of deletes (moves left along row r): leading columns whose cell has no +1 vertical delta but a +1 horizontal one
            const int l = lane - li;
            const int vd_ = static_cast<int>((e0 >> b) & 1ull) - static_cast<int>((e1 >> b) & 1ull);
            const int hd_ = static_cast<int>((e2 >> b) & 1ull) - static_cast<int>((e3 >> b) & 1ull);
            const bool isleft = l >= 0 && (c - l) >= 0 && vd_ != 1 && hd_ == 1;
            const uint64_t nleft = ~(rtk_ballot(isleft) >> li);
            const int run = nleft ? __builtin_ctzll(nleft) : 64;
            if (run > 0) {
                if (l >= 0 && l < run) tmp[cap - (nt + 1u + static_cast<uint32_t>(l))] = 2;
                j -= run; cur -= run; nt += static_cast<uint32_t>(run);
                continue;
            }
        }
        { // Runs of diagonal moves, up to 62 at a time: the move out of a cell only depends on the deltas stored around it, so the lane
          // holding column c - l looks at cell (r - l, c - l) of the diagonal through (r, c); the leading lanes that see neither an
          // insert nor a delete form one run of match / mismatch moves, written out together.
            const int l = lane - li;
            const int rl = r - l;
            const bool valid = l >= 0 && lane <= 62 && rl >= 64 * w && (c - l) >= 1;
            const int bb = rl & 63, bn = (bb + 1) & 63;
            const int vd_ = static_cast<int>((e0 >> bb) & 1ull) - static_cast<int>((e1 >> bb) & 1ull);
            const int hd_ = static_cast<int>((e2 >> bb) & 1ull) - static_cast<int>((e3 >> bb) & 1ull);
            const int vdn = static_cast<int>((e0 >> bn) & 1ull) - static_cast<int>((e1 >> bn) & 1ull); // my column, one row further down: what lane - 1 needs
            const int vdl = __shfl_down(vdn, 1, 64);
            const bool isdiag = valid && vd_ != 1 && hd_ != 1;
            const uint64_t dm = rtk_ballot(isdiag) >> li;
            const uint64_t ndm = ~dm;
            const int run = ndm ? __builtin_ctzll(ndm) : 64;
            if (run > 0) {
                const bool mine = l >= 0 && l < run;
                const bool mism = (hd_ + vdl) != 0;
                const uint64_t mm = rtk_ballot(mine && mism);
                if (mine) tmp[cap - (nt + 1u + static_cast<uint32_t>(l))] = mism ? 3 : 0;
                cur -= rtk_popc(mm); i -= run; j -= run; nt += static_cast<uint32_t>(run);
                continue;
            }
        }
        ++n_sc;
        const uint64_t a0 = RTK_RL64(e0, li), a1 = RTK_RL64(e1, li), a2 = RTK_RL64(e2, li), a3 = RTK_RL64(e3, li);
        const uint64_t l0 = RTK_RL64(e0, li + 1), l1 = RTK_RL64(e1, li + 1); // column c-1 (unused when c == 0)
#endif
        const int vd = static_cast<int>((a0 >> b) & 1ull) - static_cast<int>((a1 >> b) & 1ull);
        const int hd = static_cast<int>((a2 >> b) & 1ull) - static_cast<int>((a3 >> b) & 1ull);
        uint8_t mv;
        if (vd == 1) { mv = 1; --i; cur -= 1; }
        else if (hd == 1) { mv = 2; --j; cur -= 1; }
        else {
            const int left = cur - hd;
            int diag;
            if (c == 0) diag = i - 1;
            else diag = left - (static_cast<int>((l0 >> b) & 1ull) - static_cast<int>((l1 >> b) & 1ull));
            mv = (diag == cur) ? 0 : 3;
            --i; --j; cur = diag;
        }
        ++nt; tmp[cap - nt] = mv;
    }
    const unsigned long long tw1 = rtk_clock();
    // whatever is left is a run of inserts (query only) or deletes (target only)
    if (i > 0) { rtk_wfill(tmp + (cap - nt - static_cast<uint32_t>(i)), 1, static_cast<uint64_t>(i)); nt += static_cast<uint32_t>(i); i = 0; }
    if (j > 0) { rtk_wfill(tmp + (cap - nt - static_cast<uint32_t>(j)), 2, static_cast<uint64_t>(j)); nt += static_cast<uint32_t>(j); j = 0; }
    rtk_sync(); // runs were written by their lanes
    rtk_wcopy(rtk_ld(&sc.moves) + *n_moves, tmp + (cap - nt), nt);
    *n_moves += nt;
    { MyersScratch& msc = const_cast<MyersScratch&>(sc); msc.walk_cycles += rtk_clock() - tw0; msc.walk_moves += nt; msc.walk_reloads += n_rel; msc.walk_scalar += n_sc; msc.walk_calls += 1; msc.walk_tail_cycles += rtk_clock() - tw1; }
}

RTK_FN void rtk_myers_traceback(const MyersScratch& sc_, MySeq q_, MySeq t_, bool iupac_, uint32_t* n_moves_) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); uint32_t* n_moves = rtk_u(n_moves_); const bool iupac = rtk_u(iupac_);
    MySeq q, t; q.p = rtk_u(q_.p); q.n = rtk_u(q_.n); q.rev = rtk_u(q_.rev); t.p = rtk_u(t_.p); t.n = rtk_u(t_.n); t.rev = rtk_u(t_.rev);
    const int m = q.n, n = t.n;
    int cur;
#ifndef RTK_SIM
    SweepStat fst; fst.plain = false;
    if (m <= 4096 && !q.rev && !t.rev) fst = rtk_myers_fast_any<1, 0>(q.p, m, t.p, n, 1, iupac, rtk_ld(&sc.tb));
    if (fst.plain) { cur = fst.final_score; rtk_sync(); }
    else
#endif
    { rtk_myers_pass(sc, q, t, 1, iupac, 1, nullptr, nullptr); cur = rtk_ld(rtk_ld(&sc.colscore) + (n - 1)); }
    rtk_myers_walk(sc, m, n, n, cur, n_moves);
}
// Scores of the last column of a pass from the vertical deltas it left behind: out[i] = D[i + 1][n] for the rows of the words that are
// inside the band at the last column, RTK_BAND_INF for the others. A word parks its deltas at ITS last column; the row above the band
// then grows by one per column, so the score at the top of word w in the last column is n + the sum of the column totals of the words
// above it, whatever their last columns were (for a pass without a band: the usual prefix sum).
RTK_FN void rtk_myers_column(const uint64_t* fin_pv_, const uint64_t* fin_mv_, int m_, int n_, int32_t* out_, int dlo_, int dhi_, int gran_) {
    const uint64_t* fin_pv = rtk_u(fin_pv_); const uint64_t* fin_mv = rtk_u(fin_mv_); const int m = rtk_u(m_), n = rtk_u(n_); int32_t* out = rtk_u(out_);
    const int dlo = rtk_u(dlo_), dhi = rtk_u(dhi_), gran = rtk_u(gran_);
    const int W = (m + 63) >> 6, Wa = rtk_band_words(m, n, dlo);
    // word prefix: value at the top of word w = n + sum over previous words of (popc(P) - popc(M)) restricted to valid rows
    for (int w0 = 0, base = n; w0 < W; w0 += RTK_WAVE) {
        const int w = w0 + rtk_lane();
        int delta = 0;
        uint64_t pv = 0, mv = 0;
        const bool have = w < Wa; // the word had a column
        if (have) {
            pv = fin_pv[w]; mv = fin_mv[w];
            const int rows = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
            const uint64_t mask = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
            pv &= mask; mv &= mask;
            delta = rtk_popc(pv) - rtk_popc(mv);
        }
        int total;
        const int excl = rtk_wave_excl_scan(delta, &total);
        if (w < W) {
            const int rows = (m - 64 * w) < 64 ? (m - 64 * w) : 64;
            if (have && rtk_band_chi(w, W, n, dhi, gran) == n - 1) {
                int v = base + excl;
                for (int b = 0; b < rows; ++b) { v += static_cast<int>((pv >> b) & 1ull) - static_cast<int>((mv >> b) & 1ull); out[64 * w + b] = v; }
            } else for (int b = 0; b < rows; ++b) out[64 * w + b] = RTK_BAND_INF;
        }
        base += total;
    }
    rtk_sync();
}

#ifndef RTK_SIM
#include "rtk_myers_lvl.h"
#endif
#if defined(RTK_MULTIWAVE) && !defined(RTK_SIM)
// one leaf traceback of a leaf round, on whichever wave claimed it (its work area: st->lsc[wave]); leaves that do not fit a helper's
// work area are left to the program wave (length stays -1)
__device__ __noinline__ void rtk_myers_leaf_item(RtkCoop* st, int wave, int x) {
    int32_t* const L = reinterpret_cast<int32_t*>(rtk_u(reinterpret_cast<unsigned long long>(st->llist)));
    const char* const q = reinterpret_cast<const char*>(rtk_u(reinterpret_cast<unsigned long long>(st->lq)));
    const char* const t = reinterpret_cast<const char*>(rtk_u(reinterpret_cast<unsigned long long>(st->lt)));
    uint8_t* const mvs = reinterpret_cast<uint8_t*>(rtk_u(reinterpret_cast<unsigned long long>(st->lmoves)));
    MyersScratch* const hs = reinterpret_cast<MyersScratch*>(rtk_u(reinterpret_cast<unsigned long long>(st->lsc))) + wave;
    MyersScratch& h = *hs; RTK_ASSUME_LDS(&h);
    const bool iupac = rtk_coop_ld(&st->liupac) != 0;
    const int q0 = rtk_ld(L + 6 * x), qm = rtk_ld(L + 6 * x + 1), t0 = rtk_ld(L + 6 * x + 2), tn = rtk_ld(L + 6 * x + 3), off = rtk_ld(L + 6 * x + 5);
    int len = -1;
    const uint64_t* const need = reinterpret_cast<const uint64_t*>(rtk_u(reinterpret_cast<unsigned long long>(st->lneed)));
    if (qm == 0 || tn == 0) { rtk_wfill(mvs + off, qm == 0 ? 2 : 1, static_cast<uint64_t>(qm + tn)); len = qm + tn; } // edlib.cpp:1171-1178
    else if (need && rtk_need_none(need, t0, t0 + tn)) { rtk_wfill(mvs + off, 2, static_cast<uint64_t>(tn)); rtk_wfill(mvs + off + tn, 1, static_cast<uint64_t>(qm)); len = qm + tn; } // (MyersScratch::need_bm)
    else {
        const long long W = (qm + 63) >> 6;
        const bool fits = static_cast<uint64_t>(4 * W * tn) <= h.tb_cap_words && static_cast<uint32_t>(qm + tn) + 64u <= h.mv_cap && static_cast<uint32_t>(tn) <= h.t_cap && static_cast<uint32_t>(W) <= h.w_cap;
        if (fits) {
            h.moves = mvs + off;
            uint32_t* const nm = &st->lnm[wave];
            if (rtk_lane() == 0) *nm = 0;
            rtk_sync();
            rtk_myers_traceback(h, rtk_seq(q + q0, qm), rtk_seq(t + t0, tn), iupac, nm);
            len = static_cast<int>(rtk_u(*nm));
            if (rtk_ld(rtk_ld(&h.overflow)) != 0) len = -1; // (cannot happen after the fit test; the program wave does it again)
        }
    }
    if (rtk_lane() == 0) L[6 * x + 4] = len;
    rtk_sync();
}
#endif

// obtainAlignment (edlib.cpp:1164-1216) with the Hirschberg split of edlib.cpp:1234-1399 restated canonically:
// target halved at n/2; the FIRST query row (ascending) whose left + right scores add up to the optimum, then the
// row -1 boundary, then the last row. Iterative (explicit stack), emits moves in order into sc.moves.
// best < 0: the distance is not known yet. A problem that is split learns it from the split itself (the optimum is the smallest sum of a
// left and a right score), one that fits the in-memory traceback does not need it; *best_out (optional) receives it either way
// (-1 when the traceback branch was taken without it). Saves the separate distance pass of a Hirschberg-sized problem.
RTK_FN void rtk_myers_alignment(const MyersScratch& sc_, const char* q_, int m_, const char* t_, int n_, int best_, bool iupac_, uint32_t* n_moves_, int* best_out_ = nullptr) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); const char* q = rtk_u(q_); const char* t = rtk_u(t_); const int m = rtk_u(m_), n = rtk_u(n_), best = rtk_u(best_);
    const bool iupac = rtk_u(iupac_); uint32_t* n_moves = rtk_u(n_moves_); int* best_out = rtk_u(best_out_);
    if (best_out) *best_out = best;
    *n_moves = 0;
    { MyersScratch& msc = const_cast<MyersScratch&>(sc); msc.tb_gen = rtk_ld(&msc.tb_gen) + 1u; }
    if (static_cast<uint32_t>(m + n) > sc.mv_cap || static_cast<uint32_t>((m + 63) >> 6) > sc.w_cap || static_cast<uint32_t>(n) > sc.t_cap || static_cast<uint32_t>(m) > sc.r_cap) { *sc.overflow = 1; return; }
#ifndef RTK_SIM
    if (rtk_myers_alignment_lvl(sc, q, m, t, n, best, iupac, n_moves, best_out)) return; // level by level, the half passes of a level side by side in the lanes (rtk_myers_lvl.h)
#endif
    int32_t* st = sc.hstack;
    int sp = 0;
    MyersScratch& prof = const_cast<MyersScratch&>(sc); const unsigned long long t_all0 = rtk_clock();
    st[0] = 0; st[1] = m; st[2] = 0; st[3] = n; st[4] = best; sp = 1;
    uint64_t* fin = sc.tb; // Hirschberg passes do not store the table, so its memory holds the final delta vectors (2 x 2 x W words)
    while (sp > 0) {
        --sp;
        const int q0 = st[5 * sp], qm = st[5 * sp + 1], t0 = st[5 * sp + 2], tn = st[5 * sp + 3], bs_in = st[5 * sp + 4];
        if (qm == 0 || tn == 0) { // edlib.cpp:1171-1178
            rtk_wfill(sc.moves + *n_moves, qm == 0 ? 2 : 1, static_cast<uint64_t>(qm + tn));
            *n_moves += static_cast<uint32_t>(qm + tn);
            continue;
        }
        if (bs_in >= 0 && sc.need_bm && rtk_need_none(sc.need_bm, t0, t0 + tn)) { // (MyersScratch::need_bm: nobody looks at the moves of this stretch)
            rtk_wfill(sc.moves + *n_moves, 2, static_cast<uint64_t>(tn)); rtk_wfill(sc.moves + *n_moves + tn, 1, static_cast<uint64_t>(qm));
            *n_moves += static_cast<uint32_t>(qm + tn);
            continue;
        }
        const long long W = (qm + 63) >> 6;
        if ((2LL * 8 + 4) * W * tn + 8LL * tn < 1024 * 1024) { // edlib.cpp:1191-1193
            if (static_cast<uint64_t>(4 * W * tn) > sc.tb_cap_words) { *sc.overflow = 1; return; }
            { const unsigned long long t0_ = rtk_clock(); rtk_myers_traceback(sc, rtk_seq(q + q0, qm), rtk_seq(t + t0, tn), iupac, n_moves); prof.hb_leaf += rtk_clock() - t0_; }
            continue;
        }
        const int lh = tn / 2, rh = tn - lh;
        if (lh == 0 || static_cast<uint64_t>(4 * W) > sc.tb_cap_words || sp + 2 > 60) { *sc.overflow = 1; return; }
        // The two half passes only sweep the Ukkonen band of the sub-problem's distance (the right one is the same band seen from the other
        // corner: the band is symmetric under d -> (tn - qm) - d). The distance of the whole problem may be unknown (bs_in < 0): a guess,
        // and a second round with the score the first one found when that score is above the guess.
        int kk = bs_in >= 0 ? bs_in : rtk_band_guess(qm, tn);
        int bs_known = bs_in;
        for (;;) {
            const RtkBand band = rtk_band_nw(qm, tn, kk);
            const int gran = rtk_band_gran(qm, band.dlo, band.dhi);
            bool both = false;
            const unsigned long long t_p0 = rtk_clock();
#if defined(RTK_MULTIWAVE) && !defined(RTK_SIM)
            { const MySeq qa = rtk_seq(q + q0, qm), ta = rtk_seq(t + t0, lh), qb = rtk_seq(q + q0, qm, 1), tb_ = rtk_seq(t + t0 + lh, rh, 1); // the two half passes side by side on the waves of the workgroup
              both = static_cast<uint32_t>(lh + rh) <= sc.t_cap && rtk_myers_pass_coop(sc, qa, ta, 1, iupac, band, fin, fin + W, &qb, &tb_, fin + 2 * W, fin + 3 * W);
              if (both) rtk_sync(); }
#endif
            if (!both) rtk_myers_pass_banded(sc, rtk_seq(q + q0, qm), rtk_seq(t + t0, lh), iupac, band.dlo, band.dhi, fin, fin + W);
            if (!both) rtk_myers_pass_banded(sc, rtk_seq(q + q0, qm, 1), rtk_seq(t + t0 + lh, rh, 1), iupac, band.dlo, band.dhi, fin + 2 * W, fin + 3 * W);
            const unsigned long long t_p1 = rtk_clock(); prof.hb_pass += t_p1 - t_p0;
            rtk_myers_column(fin, fin + W, qm, lh, sc.rowL, band.dlo, band.dhi, gran);
            rtk_myers_column(fin + 2 * W, fin + 3 * W, qm, rh, sc.rowR, band.dlo, band.dhi, gran);
            prof.hb_split += rtk_clock() - t_p1;
            if (bs_in >= 0) break;
            // R(i) = cost of aligning q[i..qm) with the right half = rowR[qm-1-i]
            // the optimum = the smallest left + right sum over every split point (rows 0 .. qm-2, the row -1 boundary, the last row)
            int mn = 0x7fffffff;
            for (int b0 = 0; b0 + 1 < qm; b0 += RTK_WAVE) { const int qi = b0 + rtk_lane(); if (qi + 1 < qm) { const int v = sc.rowL[qi] + sc.rowR[qm - 2 - qi]; mn = v < mn ? v : mn; } }
#ifndef RTK_SIM
            for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(mn, o, 64); mn = v < mn ? v : mn; }
#endif
            mn = rtk_u(mn);
            const int e0 = lh + sc.rowR[qm - 1], e1 = sc.rowL[qm - 1] + rh;
            mn = e0 < mn ? e0 : mn; mn = e1 < mn ? e1 : mn;
            if (mn <= kk) { bs_known = mn; if (best_out) *best_out = mn; break; }
            kk = mn; // a real score, above the guess: its band holds the optimum
        }
        const unsigned long long t_p1 = rtk_clock();
        const int bs = bs_known;
        int split = -2;
        for (int b0 = 0; b0 + 1 < qm && split == -2; b0 += RTK_WAVE) {
            const int qi = b0 + rtk_lane();
            const bool ok = (qi + 1 < qm) && (sc.rowL[qi] + sc.rowR[qm - 2 - qi] == bs);
            const uint64_t bal = rtk_ballot(ok);
            if (bal) split = b0 + rtk_ffs(bal) - 1;
        }
        int ls, rs;
        if (split >= 0) { ls = sc.rowL[split]; rs = sc.rowR[qm - 2 - split]; }
        else if (lh + sc.rowR[qm - 1] == bs) { split = -1; ls = lh; rs = sc.rowR[qm - 1]; }
        else if (sc.rowL[qm - 1] + rh == bs) { split = qm - 1; ls = sc.rowL[qm - 1]; rs = rh; }
        else { *sc.overflow = 2; return; } // inconsistent optimum: cannot happen for a correct distance
        const int ul = split + 1;
        // push right then left so that the left half is emitted first
        st[5 * sp] = q0 + ul; st[5 * sp + 1] = qm - ul; st[5 * sp + 2] = t0 + lh; st[5 * sp + 3] = rh; st[5 * sp + 4] = rs; ++sp;
        st[5 * sp] = q0; st[5 * sp + 1] = ul; st[5 * sp + 2] = t0; st[5 * sp + 3] = lh; st[5 * sp + 4] = ls; ++sp;
        prof.hb_split += rtk_clock() - t_p1;
    }
    prof.hb_total += rtk_clock() - t_all0;
}

#include "variants/rtk_myers_lt.h"

// edlibAlign(..., k = -1, NW or SHW, TASK_PATH): result and moves. When the whole table fits the in-memory traceback branch of
// obtainAlignment (edlib.cpp:1191-1193) ONE stored sweep serves both the distance and the traceback: an SHW matrix restricted to
// columns [0, end] IS the NW matrix of the truncated target edlib re-aligns (same top row, same left column), so the table
// entries, the walk and the moves are the same. Anything else (IUPAC/N in the target, long queries, Hirschberg-sized tables)
// takes the two-pass route.
RTK_FN MyersResult rtk_myers_path(const MyersScratch& sc_, const char* q_, int m_, const char* t_, int n_, int mode_, bool iupac_, uint32_t* n_moves_) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); const char* q = rtk_u(q_); const char* t = rtk_u(t_); const int m = rtk_u(m_), n = rtk_u(n_), mode = rtk_u(mode_);
    const bool iupac = rtk_u(iupac_); uint32_t* n_moves = rtk_u(n_moves_);
    *n_moves = 0;
    { MyersScratch& msc = const_cast<MyersScratch&>(sc); msc.tb_gen = rtk_ld(&msc.tb_gen) + 1u; }
    MyersResult r; bool have = false;
#ifdef RTK_HAVE_LT
    if (rtk_lt_fits(sc, m, n)) { // small problem: the table lives in LDS, a chunk of steps at a time (rtk_myers_lt.h)
        SweepStat st;
        const bool ok = (mode == RTK_MODE_NW) ? rtk_lt_align<0>(sc, q, m, t, n, iupac, RTK_MODE_NW, &st, n_moves) : rtk_lt_align<1>(sc, q, m, t, n, iupac, RTK_MODE_SHW, &st, n_moves);
        rtk_sync();
        if (ok) {
            r.dist = -1; r.first = -1; r.last = -1; r.nloc = 0;
            if (mode == RTK_MODE_NW) { r.dist = st.final_score; r.first = r.last = n - 1; r.nloc = 1; return r; }
            int best = st.best; const bool pseudo = (m & 63) != 0;
            if (pseudo && m < best) best = m;
            r.dist = best;
            if (pseudo && m == best) { r.first = -1; r.last = (st.best == best) ? st.last : -1; r.nloc = 1 + ((st.best == best) ? st.cnt : 0); }
            else { r.first = st.first; r.last = st.last; r.nloc = st.cnt; }
            if (r.first + 1 <= 0) rtk_myers_alignment(sc, q, m, t, 0, r.dist, iupac, n_moves); // empty target prefix: m inserts (edlib.cpp:1171-1178)
            return r;
        }
        *n_moves = 0;
    }
#endif
#ifndef RTK_SIM
    const long long W = (m + 63) >> 6;
    if (m > 0 && n > 0 && m <= 4096 && static_cast<uint64_t>(4 * W * n) <= sc.tb_cap_words && static_cast<uint32_t>(m + n) <= sc.mv_cap && static_cast<uint32_t>(n) <= sc.t_cap &&
        static_cast<uint32_t>(m) <= sc.r_cap && static_cast<uint32_t>(W) <= sc.w_cap) {
        const SweepStat st = (mode == RTK_MODE_NW) ? rtk_myers_fast_any<1, 0>(q, m, t, n, 1, iupac, rtk_ld(&sc.tb)) : rtk_myers_fast_any<1, 1>(q, m, t, n, 1, iupac, rtk_ld(&sc.tb));
        rtk_sync();
        if (st.plain) {
            r.dist = -1; r.first = -1; r.last = -1; r.nloc = 0;
            if (mode == RTK_MODE_NW) { r.dist = st.final_score; r.first = r.last = n - 1; r.nloc = 1; }
            else { // same bookkeeping as rtk_myers_distance
                int best = st.best; const bool pseudo = (m & 63) != 0;
                if (pseudo && m < best) best = m;
                r.dist = best;
                if (pseudo && m == best) { r.first = -1; r.last = (st.best == best) ? st.last : -1; r.nloc = 1 + ((st.best == best) ? st.cnt : 0); }
                else { r.first = st.first; r.last = st.last; r.nloc = st.cnt; }
            }
            have = true;
            const long long tn = (mode == RTK_MODE_NW) ? n : (r.first + 1);
            if (tn > 0 && (2LL * 8 + 4) * W * tn + 8LL * tn < 1024 * 1024) { rtk_myers_walk(sc, m, static_cast<int>(tn), n, r.dist, n_moves); return r; }
        }
    }
#endif
    if (!have && mode == RTK_MODE_NW && m > 0 && n > 0 && (2LL * 8 + 4) * ((m + 63) >> 6) * n + 8LL * n >= 1024 * 1024) {
        // Hirschberg-sized NW problem: the first split yields the distance, no separate distance pass
        int d = -1;
        rtk_myers_alignment(sc, q, m, t, n, -1, iupac, n_moves, &d);
        r.dist = d; r.first = r.last = n - 1; r.nloc = 1;
        return r;
    }
    if (!have) r = rtk_myers_distance(sc, q, m, t, n, -1, mode, iupac);
    if (m > 0 && n > 0) rtk_myers_alignment(sc, q, m, t, (mode == RTK_MODE_NW) ? n : (r.first + 1), r.dist, iupac, n_moves);
    return r;
}


// NW distance and SHW path alignment of the SAME pair from ONE stored sweep. An SHW matrix (top row 0..n, left column 0..m) is the NW
// matrix; its bottom-right cell is the NW distance, the minimum of its last row the SHW result. getScorePath scores a terminal path
// with NW (src/GraphTraversal.cpp:880) and, if it survives, aligns the same two strings again with SHW + path for its quality string
// (:727): rtk_myers_nw_and_save answers the first call and keeps what the second one needs; rtk_myers_path_from_saved then only walks.
struct MyersSaved { uint32_t valid, gen; int32_t m, n, nw_dist; MyersResult shw; uint8_t* stash; uint32_t stash_cap, stash_n; }; // stash (set by the caller, may be null): room for the moves of the alignment, kept from the sweep on (LDS-table route: the table does not outlive the call)
RTK_FN bool rtk_myers_nw_and_save(const MyersScratch& sc_, const char* q_, int m_, const char* t_, int n_, bool iupac_, MyersSaved* out_) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); const char* q = rtk_u(q_); const char* t = rtk_u(t_); const int m = rtk_u(m_), n = rtk_u(n_); const bool iupac = rtk_u(iupac_); MyersSaved* out = rtk_u(out_);
    out->valid = 0;
#ifdef RTK_HAVE_LT
    if (out->stash && rtk_lt_fits(sc, m, n) && static_cast<uint32_t>(m + n) <= out->stash_cap) {
        // the SHW path alignment right away (its table lives in LDS and is gone when this call returns); the moves wait in the caller's stash
        SweepStat st; uint32_t nm = 0;
        MyersScratch& msc = const_cast<MyersScratch&>(sc); const uint32_t gen = rtk_ld(&msc.tb_gen) + 1u; msc.tb_gen = gen;
        const bool ok = rtk_lt_align<1>(sc, q, m, t, n, iupac, RTK_MODE_SHW, &st, &nm);
        rtk_sync();
        if (ok) {
            MyersResult r;
            int best = st.best; const bool pseudo = (m & 63) != 0;
            if (pseudo && m < best) best = m;
            r.dist = best;
            if (pseudo && m == best) { r.first = -1; r.last = (st.best == best) ? st.last : -1; r.nloc = 1 + ((st.best == best) ? st.cnt : 0); }
            else { r.first = st.first; r.last = st.last; r.nloc = st.cnt; }
            out->m = m; out->n = n; out->nw_dist = st.final_score; out->shw = r; out->gen = gen;
            if (r.first + 1 > 0) { nm = rtk_u(nm); rtk_wcopy(out->stash, rtk_ld(&sc.moves), nm); out->stash_n = nm; out->valid = 2u; } // 2: moves in the stash
            return true;
        }
    }
#endif
#ifndef RTK_SIM
    const long long W = (m + 63) >> 6;
    if (!(m > 0 && n > 0 && m <= 4096 && static_cast<uint64_t>(4 * W * n) <= sc.tb_cap_words && static_cast<uint32_t>(m + n) <= sc.mv_cap && static_cast<uint32_t>(n) <= sc.t_cap &&
          static_cast<uint32_t>(m) <= sc.r_cap && static_cast<uint32_t>(W) <= sc.w_cap)) return false;
    MyersScratch& msc = const_cast<MyersScratch&>(sc); const uint32_t gen = rtk_ld(&msc.tb_gen) + 1u; msc.tb_gen = gen;
    const SweepStat st = rtk_myers_fast_any<1, 1>(q, m, t, n, 1, iupac, rtk_ld(&sc.tb));
    rtk_sync();
    if (!st.plain) return false;
    MyersResult r; // same bookkeeping as rtk_myers_distance (SHW)
    int best = st.best; const bool pseudo = (m & 63) != 0;
    if (pseudo && m < best) best = m;
    r.dist = best;
    if (pseudo && m == best) { r.first = -1; r.last = (st.best == best) ? st.last : -1; r.nloc = 1 + ((st.best == best) ? st.cnt : 0); }
    else { r.first = st.first; r.last = st.last; r.nloc = st.cnt; }
    out->m = m; out->n = n; out->nw_dist = st.final_score; out->shw = r; out->gen = gen;
    const long long tn = r.first + 1;
    out->valid = (tn > 0 && (2LL * 8 + 4) * W * tn + 8LL * tn < 1024 * 1024) ? 1u : 0u; // the in-memory traceback branch of obtainAlignment (edlib.cpp:1191-1193)
    return true;
#else
    (void)sc; (void)q; (void)t; (void)m; (void)n; (void)iupac;
    return false;
#endif
}
RTK_FN bool rtk_myers_path_from_saved(const MyersScratch& sc_, const MyersSaved& sv_, uint32_t* n_moves_, MyersResult* r_) {
    const MyersScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); const MyersSaved& sv = *rtk_u(&sv_); uint32_t* n_moves = rtk_u(n_moves_); MyersResult* r = rtk_u(r_);
    *n_moves = 0;
    if (sv.valid == 2u) { *r = sv.shw; rtk_wcopy(rtk_ld(&sc.moves), sv.stash, sv.stash_n); *n_moves = sv.stash_n; return true; } // walked when it was swept
    if (!sv.valid || sv.gen != rtk_ld(&sc.tb_gen)) return false; // the table has been written again since
    *r = sv.shw;
    rtk_myers_walk(sc, sv.m, sv.shw.first + 1, sv.n, sv.shw.dist, n_moves);
    return true;
}

// work area of a helper wave for the leaf tracebacks of a multi-wave alignment: a table of the in-memory traceback branch (< 1 MB by edlib's
// count, edlib.cpp:1191-1193), one row block, the longest target / move list such a leaf can have with that
struct ScratchCfg;
// one problem of the stage entry rtk_myers_batch (offsets into one character pool)
struct MyersProb { uint64_t q_off, t_off; uint32_t qlen, tlen; int32_t k, mode; };

// ------------------------------------------------------------------------------------------------ work area of one wave
struct ScratchCfg { uint32_t w_cap, t_cap, r_cap, mv_cap; uint64_t tb_cap_words; };

#define RTK_LEAF_WAVES 8 // waves of a workgroup that take part in leaf rounds (the program wave + 7 helpers with a work area each)
RTK_HD ScratchCfg rtk_leaf_cfg() { ScratchCfg c; c.w_cap = 64; c.t_cap = 37504; c.r_cap = 64; c.mv_cap = 53248; c.tb_cap_words = 4ull * 52429 + 64; return c; }
RTK_HD uint64_t scratch_bytes(const ScratchCfg& c) {
    uint64_t b = 0;
    b += 8ull * 15 * c.w_cap; b += (c.t_cap + 63) / 64 * 64; b += 4ull * c.t_cap; b += 8ull * c.tb_cap_words; b += 8ull * c.r_cap;
    b += 2ull * ((c.mv_cap + 63) / 64 * 64); b += 4 * 5 * 64; b += 64;
    return (b + 255) / 256 * 256;
}

RTK_HD MyersScratch scratch_carve(char* base, const ScratchCfg& c) {
    MyersScratch s; char* p = base;
    s.peq = reinterpret_cast<uint64_t*>(p); p += 8ull * 15 * c.w_cap; s.w_cap = c.w_cap;
    s.tb = reinterpret_cast<uint64_t*>(p); p += 8ull * c.tb_cap_words; s.tb_cap_words = c.tb_cap_words;
    s.colscore = reinterpret_cast<int32_t*>(p); p += 4ull * c.t_cap; s.t_cap = c.t_cap;
    s.rowL = reinterpret_cast<int32_t*>(p); p += 4ull * c.r_cap; s.rowR = reinterpret_cast<int32_t*>(p); p += 4ull * c.r_cap; s.r_cap = c.r_cap;
    s.hstack = reinterpret_cast<int32_t*>(p); p += 4 * 5 * 64;
    s.overflow = reinterpret_cast<uint32_t*>(p); p += 64;
    s.tb_gen = 0;
    s.walk_cycles = 0; s.walk_moves = 0; s.walk_reloads = 0; s.walk_scalar = 0; s.walk_calls = 0; s.walk_tail_cycles = 0;
    s.hb_pass = 0; s.hb_split = 0; s.hb_leaf = 0; s.hb_total = 0; s.need_bm = nullptr;
    s.carry = reinterpret_cast<int8_t*>(p); p += (c.t_cap + 63) / 64 * 64;
    s.moves = reinterpret_cast<uint8_t*>(p); p += (c.mv_cap + 63) / 64 * 64; s.moves_tmp = reinterpret_cast<uint8_t*>(p); s.mv_cap = c.mv_cap;
    return s;
}


#endif
