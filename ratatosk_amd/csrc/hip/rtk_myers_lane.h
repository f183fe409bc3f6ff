// One alignment per LANE: distance and end locations of edlibAlign (NW / SHW / HW, threshold k) for the small problems of the region program -- queries of a
// few words against targets of a few hundred characters -- computed column by column with the query's delta vectors in the lane's registers (reference:
// src/edlib.cpp:586-677 the block recurrence, :161-179 zero lengths, :744-747 NW threshold, the padded last block's position -1). The wave programs of
// rtk_myers.h give such a problem one wave and keep one WORD per lane busy (3-10 % of the lanes, DESIGN_HISTORY.md section 3.5); here 64 problems share a wave.
// Stage entry rtk_myers_batch_lanes: the building block of the lane-per-region formulation (DESIGN_HISTORY.md section 9), held to the same golden vectors; it is
// not called by the correction path yet. No cross-lane operation: the 1-lane simulator runs exactly the code a lane runs on the device.
#ifndef RTK_MYERS_LANE_H
#define RTK_MYERS_LANE_H

#define RTK_ML_MAXW 8      // words of the query (512 characters)
#define RTK_ML_NSYM 5      // target characters of this route: A C T G N (anything else in the target: wave route)
#define RTK_ML_MAXN 2048   // characters of the target

// bytes of work area per wave in device memory: the last-row score of every column, interleaved by lane (SHW / HW list every column with the best score)
RTK_HD uint64_t rtk_ml_scratch_bytes() { return static_cast<uint64_t>(RTK_WAVE) * 4ull * RTK_ML_MAXN; }
// words of match vectors per wave (LDS on the device: [symbol][word][lane], a lane's words are 64 apart: no bank conflicts)
#define RTK_ML_PEQ_WORDS (RTK_ML_NSYM * RTK_ML_MAXW * RTK_WAVE)

RTK_DEV uint64_t rtk_ml_ld8(const char* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; } // (the texts are read eight characters at a time; the pool is padded by 64 bytes)
// bit j = byte j of x equals c
RTK_DEV uint64_t rtk_ml_eq8(uint64_t x, uint64_t c) {
    const uint64_t y = x ^ (c * 0x0101010101010101ull);
    const uint64_t z = ~(((y & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | y) & 0x8080808080808080ull;
    return ((z >> 7) * 0x0102040810204080ull) >> 56;
}
// 0 A, 1 C, 2 T, 3 G ((c >> 1) & 3), 4 N; 5 = not a character of this route
RTK_DEV int rtk_ml_sym(uint32_t c) { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? static_cast<int>((c >> 1) & 3u) : (c == 'N' ? 4 : 5); }

// returns 0 = done (dist / nloc / locs written), 1 = not a problem for this route (too long, a target character outside ACGTN): the caller takes the wave route.
// iupac: equality of edlibAlign with the IUPAC codes as additional equalities (what the region program aligns with); the target's characters are A C G T N either way.
// peq: this lane's slice of the wave's match vectors (word (s * MAXW + w) at peq[(s * MAXW + w) * RTK_WAVE]); cs: its slice of the column scores.
RTK_DEV uint32_t rtk_myers_lane(const char* q, int m, const char* t, int n, int k, int mode, bool iupac, uint64_t* peq, int32_t* cs, int32_t* dist, int32_t* nloc, int32_t* locs, int cap, int32_t* first) {
    *dist = -1; *nloc = 0; *first = -1;
    if (m == 0 || n == 0) { // edlib.cpp:161-179
        if (mode == RTK_MODE_NW) { *dist = m > n ? m : n; *first = n - 1; if (cap > 0) locs[0] = n - 1; } else { *dist = m; if (cap > 0) locs[0] = -1; }
        *nloc = 1; return 0u;
    }
    if (m > 64 * RTK_ML_MAXW || n > RTK_ML_MAXN) return 1u;
    if (mode == RTK_MODE_NW && k >= 0 && k < (n > m ? n - m : m - n)) return 0u; // edlib.cpp:744-747
    const int W = (m + 63) >> 6, last_bit = (m - 1) & 63;
    // match vectors of the five characters: eight query characters per load, eight bits per character and load
    for (int w = 0; w < W; ++w) {
        uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
        for (int g = 0; g < 8 && 64 * w + 8 * g < m; ++g) {
            const uint64_t x = rtk_ml_ld8(q + 64 * w + 8 * g); const int sh = 8 * g;
            const uint64_t a8 = rtk_ml_eq8(x, 'A'), c8 = rtk_ml_eq8(x, 'C'), t8 = rtk_ml_eq8(x, 'T'), g8 = rtk_ml_eq8(x, 'G'), n8 = rtk_ml_eq8(x, 'N');
            e0 |= a8 << sh; e1 |= c8 << sh; e2 |= t8 << sh; e3 |= g8 << sh; e4 |= n8 << sh;
            if (iupac) { // IUPAC equality (edlib's additional equalities, src/Alignment.cpp: a code equals the bases it stands for): a base of the query equals a target N; a
                         // code of the query -- rare: a merged SNP -- is looked at character by character
                e4 |= (a8 | c8 | t8 | g8) << sh;
                uint64_t other = ~(a8 | c8 | t8 | g8) & 0xFFull;
                const int left = m - (64 * w + 8 * g); if (left < 8) other &= (1ull << left) - 1ull;
                for (; other; other &= other - 1ull) {
                    const int b = __builtin_ctzll(other); const unsigned char qc = static_cast<unsigned char>(x >> (8 * b)); const uint64_t bit = 1ull << (sh + b);
                    if (rtk_chars_equal(qc, 'A', true)) e0 |= bit; if (rtk_chars_equal(qc, 'C', true)) e1 |= bit; if (rtk_chars_equal(qc, 'T', true)) e2 |= bit;
                    if (rtk_chars_equal(qc, 'G', true)) e3 |= bit; if (rtk_chars_equal(qc, 'N', true)) e4 |= bit;
                }
            }
        }
        const uint64_t keep = (w == W - 1 && last_bit < 63) ? ((2ull << last_bit) - 1ull) : ~0ull; // (characters behind the query's end match nothing)
        peq[static_cast<uint64_t>(0 * RTK_ML_MAXW + w) * RTK_WAVE] = e0 & keep; peq[static_cast<uint64_t>(1 * RTK_ML_MAXW + w) * RTK_WAVE] = e1 & keep;
        peq[static_cast<uint64_t>(2 * RTK_ML_MAXW + w) * RTK_WAVE] = e2 & keep; peq[static_cast<uint64_t>(3 * RTK_ML_MAXW + w) * RTK_WAVE] = e3 & keep;
        peq[static_cast<uint64_t>(4 * RTK_ML_MAXW + w) * RTK_WAVE] = e4 & keep;
    }
    uint64_t Pv[RTK_ML_MAXW], Mv[RTK_ML_MAXW];
#pragma unroll
    for (int w = 0; w < RTK_ML_MAXW; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
    int score = m, best = 0x7fffffff;
    const int top_h = (mode == RTK_MODE_HW) ? 0 : 1;
    const bool every_column = mode != RTK_MODE_NW;
    uint64_t tw = 0;
    for (int j = 0; j < n; ++j) {
        if ((j & 7) == 0) tw = rtk_ml_ld8(t + j);
        const int s = rtk_ml_sym(static_cast<uint32_t>(tw & 0xFFull)); tw >>= 8;
        if (s > 4) return 1u;
        const uint64_t* const e = peq + static_cast<uint64_t>(s * RTK_ML_MAXW) * RTK_WAVE;
        int hin = top_h;
#pragma unroll
        for (int w = 0; w < RTK_ML_MAXW; ++w) if (w < W) { // edlib.cpp:586-677, one word
            uint64_t Eq = e[static_cast<uint64_t>(w) * RTK_WAVE];
            const uint64_t pv = Pv[w], mv = Mv[w];
            const uint64_t Xv = Eq | mv;
            if (hin < 0) Eq |= 1ull;
            const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv), Mh = pv & Xh;
            const int bit = (w == W - 1) ? last_bit : 63;
            const int hout = static_cast<int>((Ph >> bit) & 1ull) - static_cast<int>((Mh >> bit) & 1ull);
            Ph <<= 1; Mh <<= 1;
            if (hin > 0) Ph |= 1ull; else if (hin < 0) Mh |= 1ull;
            Pv[w] = Mh | ~(Xv | Ph); Mv[w] = Ph & Xv;
            hin = hout;
        }
        score += hin;
        if (every_column) { cs[static_cast<uint64_t>(j) * RTK_WAVE] = score; best = score < best ? score : best; }
    }
    if (mode == RTK_MODE_NW) {
        if (k >= 0 && score > k) return 0u;
        *dist = score; *nloc = 1; *first = n - 1; if (cap > 0) locs[0] = n - 1;
        return 0u;
    }
    const bool pseudo = (m & 63) != 0; // edlib's padded last block exposes target position -1 with score m
    if (pseudo && m < best) best = m;
    if (k >= 0 && best > k) return 0u;
    *dist = best;
    int nl = 0, f = -2;
    if (pseudo && m == best) { if (nl < cap) locs[nl] = -1; ++nl; f = -1; }
    for (int j = 0; j < n; ++j) if (cs[static_cast<uint64_t>(j) * RTK_WAVE] == best) { if (nl < cap) locs[nl] = j; ++nl; if (f == -2) f = j; }
    *nloc = nl; *first = f;
    return 0u;
}


// ---- the path of a lane's problem (edlib.cpp:271-284 obtainAlignment below its 1 MB threshold: one NW pass of the whole query over target[0 .. n) that keeps the four
// delta vectors of every column, then the walk of edlib.cpp:1021-1137 -- up before left before diagonal). tb: this lane's slice of the wave's table, entry
// ((j * W + w) * 4 + f) * RTK_WAVE with f = 0 Pv, 1 Mv, 2 Ph, 3 Mh (interleaved by lane: the 64 lanes of a wave store 512 contiguous bytes per entry).
// moves: 0 match, 1 insertion (query character alone), 2 deletion, 3 mismatch, written in alignment order; returns 1 when the table or the move list is too small.
#define RTK_ML_TB_WORDCOLS 4096 // word-columns of a lane's table (128 KB per lane)
RTK_HD uint64_t rtk_ml_table_bytes() { return static_cast<uint64_t>(RTK_WAVE) * 32ull * RTK_ML_TB_WORDCOLS; }

RTK_DEV uint32_t rtk_myers_lane_path(int m, const char* t, int n, const uint64_t* peq, uint64_t* tb, uint8_t* moves, uint32_t cap, uint32_t* n_moves) {
    *n_moves = 0;
    const int W = (m + 63) >> 6, last_bit = (m - 1) & 63;
    if (static_cast<uint64_t>(W) * static_cast<uint64_t>(n) > RTK_ML_TB_WORDCOLS || static_cast<uint32_t>(m + n) > cap) return 1u;
    uint64_t Pv[RTK_ML_MAXW], Mv[RTK_ML_MAXW];
#pragma unroll
    for (int w = 0; w < RTK_ML_MAXW; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
    int cur = m;
    uint64_t tw = 0;
    for (int j = 0; j < n; ++j) {
        if ((j & 7) == 0) tw = rtk_ml_ld8(t + j);
        const int s = rtk_ml_sym(static_cast<uint32_t>(tw & 0xFFull)); tw >>= 8;
        if (s > 4) return 1u;
        const uint64_t* const e = peq + static_cast<uint64_t>(s * RTK_ML_MAXW) * RTK_WAVE;
        int hin = 1;
#pragma unroll
        for (int w = 0; w < RTK_ML_MAXW; ++w) if (w < W) {
            uint64_t Eq = e[static_cast<uint64_t>(w) * RTK_WAVE];
            const uint64_t pv = Pv[w], mv = Mv[w];
            const uint64_t Xv = Eq | mv;
            if (hin < 0) Eq |= 1ull;
            const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv), Mh = pv & Xh;
            uint64_t* const ent = tb + static_cast<uint64_t>(j * W + w) * 4ull * RTK_WAVE;
            ent[2 * RTK_WAVE] = Ph; ent[3 * RTK_WAVE] = Mh;
            const int bit = (w == W - 1) ? last_bit : 63;
            const int hout = static_cast<int>((Ph >> bit) & 1ull) - static_cast<int>((Mh >> bit) & 1ull);
            Ph <<= 1; Mh <<= 1;
            if (hin > 0) Ph |= 1ull; else if (hin < 0) Mh |= 1ull;
            Pv[w] = Mh | ~(Xv | Ph); Mv[w] = Ph & Xv;
            ent[0] = Pv[w]; ent[RTK_WAVE] = Mv[w];
            hin = hout;
        }
        cur += hin;
    }
    // the walk, from the last cell; the moves are produced last first, at the end of the list, and moved to its front afterwards
    uint32_t o = cap;
    int i = m, j = n;
    while (i > 0 && j > 0) {
        const int r = i - 1, cc = j - 1, w = r >> 6, b = r & 63;
        const uint64_t* const ent = tb + static_cast<uint64_t>(cc * W + w) * 4ull * RTK_WAVE;
        const int vd = static_cast<int>((ent[0] >> b) & 1ull) - static_cast<int>((ent[RTK_WAVE] >> b) & 1ull);
        const int hd = static_cast<int>((ent[2 * RTK_WAVE] >> b) & 1ull) - static_cast<int>((ent[3 * RTK_WAVE] >> b) & 1ull);
        if (vd == 1) { moves[--o] = 1; --i; cur -= 1; }
        else if (hd == 1) { moves[--o] = 2; --j; cur -= 1; }
        else {
            const int left = cur - hd;
            int diag;
            if (cc == 0) diag = i - 1;
            else { const uint64_t* const el = tb + static_cast<uint64_t>((cc - 1) * W + w) * 4ull * RTK_WAVE; diag = left - (static_cast<int>((el[0] >> b) & 1ull) - static_cast<int>((el[RTK_WAVE] >> b) & 1ull)); }
            moves[--o] = static_cast<uint8_t>(diag == cur ? 0 : 3);
            --i; --j; cur = diag;
        }
    }
    while (i > 0) { moves[--o] = 1; --i; }
    while (j > 0) { moves[--o] = 2; --j; }
    const uint32_t nm = cap - o;
    for (uint32_t x = 0; x < nm; ++x) moves[x] = moves[o + x];
    *n_moves = nm;
    return 0u;
}

#endif
