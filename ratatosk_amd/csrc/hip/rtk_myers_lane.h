// One alignment per LANE: distance and end locations of edlibAlign (NW / SHW / HW, threshold k) for the small problems of the region program -- queries of a
// few words against targets of a few hundred characters -- computed column by column with the query's delta vectors in the lane's registers (reference:
// src/edlib.cpp:586-677 the block recurrence, :161-179 zero lengths, :744-747 NW threshold, the padded last block's position -1). The wave programs of
// rtk_myers.h give such a problem one wave and keep one WORD per lane busy (3-10 % of the lanes); here 64 problems share a wave.
// Stage entry rtk_myers_batch_lanes: the building block of the lane-per-region formulation, held to the same golden vectors; it is
// not called by the correction path yet. No cross-lane operation: the 1-lane simulator runs exactly the code a lane runs on the device.
#ifndef RTK_MYERS_LANE_H
#define RTK_MYERS_LANE_H

#define RTK_ML_MAXW 8      // words of the query (512 characters)
#define RTK_ML_MAXSYM 8    // distinct characters of the target
#define RTK_ML_MAXN 2048   // characters of the target

// bytes of work area per wave: the match vectors of MAXSYM characters and the last-row score of every column, interleaved by lane
RTK_HD uint64_t rtk_ml_scratch_bytes() { return static_cast<uint64_t>(RTK_WAVE) * (8ull * RTK_ML_MAXSYM * RTK_ML_MAXW + 4ull * RTK_ML_MAXN); }

// returns 0 = done (dist / nloc / locs written), 1 = not a problem for this route (too long, too many distinct characters): the caller takes the wave route
RTK_DEV uint32_t rtk_myers_lane(const char* q, int m, const char* t, int n, int k, int mode, uint64_t* peq, int32_t* cs, int32_t* dist, int32_t* nloc, int32_t* locs, int cap) {
    *dist = -1; *nloc = 0;
    if (m == 0 || n == 0) { // edlib.cpp:161-179
        if (mode == RTK_MODE_NW) { *dist = m > n ? m : n; if (cap > 0) locs[0] = n - 1; } else { *dist = m; if (cap > 0) locs[0] = -1; }
        *nloc = 1; return 0u;
    }
    if (m > 64 * RTK_ML_MAXW || n > RTK_ML_MAXN) return 1u;
    if (mode == RTK_MODE_NW && k >= 0 && k < (n > m ? n - m : m - n)) return 0u; // edlib.cpp:744-747
    const int W = (m + 63) >> 6, last_bit = (m - 1) & 63;
    // the distinct characters of the target (packed, one byte each) and their match vectors
    uint64_t syms = 0; int ns = 0;
    for (int j = 0; j < n; ++j) {
        const uint64_t c = static_cast<unsigned char>(t[j]);
        bool seen = false;
        for (int s = 0; s < ns; ++s) seen = seen || ((syms >> (8 * s)) & 0xFFull) == c;
        if (seen) continue;
        if (ns == RTK_ML_MAXSYM) return 1u;
        for (int w = 0; w < W; ++w) {
            uint64_t bits = 0; const int i1 = (64 * w + 64 < m) ? 64 * w + 64 : m;
            for (int i = 64 * w; i < i1; ++i) bits |= static_cast<uint64_t>(static_cast<unsigned char>(q[i]) == c ? 1u : 0u) << (i & 63);
            peq[static_cast<uint64_t>(ns * RTK_ML_MAXW + w) * RTK_WAVE] = bits;
        }
        syms |= c << (8 * ns); ++ns;
    }
    uint64_t Pv[RTK_ML_MAXW], Mv[RTK_ML_MAXW];
#pragma unroll
    for (int w = 0; w < RTK_ML_MAXW; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
    int score = m, best = 0x7fffffff;
    const int top_h = (mode == RTK_MODE_HW) ? 0 : 1;
    for (int j = 0; j < n; ++j) {
        const uint64_t c = static_cast<unsigned char>(t[j]);
        int s = 0; for (int x = 1; x < ns; ++x) if (((syms >> (8 * x)) & 0xFFull) == c) s = x;
        int hin = top_h;
#pragma unroll
        for (int w = 0; w < RTK_ML_MAXW; ++w) if (w < W) { // edlib.cpp:586-677, one word
            uint64_t Eq = peq[static_cast<uint64_t>(s * RTK_ML_MAXW + w) * RTK_WAVE];
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
        cs[static_cast<uint64_t>(j) * RTK_WAVE] = score;
        best = score < best ? score : best;
    }
    if (mode == RTK_MODE_NW) {
        if (k >= 0 && score > k) return 0u;
        *dist = score; *nloc = 1; if (cap > 0) locs[0] = n - 1;
        return 0u;
    }
    const bool pseudo = (m & 63) != 0; // edlib's padded last block exposes target position -1 with score m
    if (pseudo && m < best) best = m;
    if (k >= 0 && best > k) return 0u;
    *dist = best;
    int nl = 0;
    if (pseudo && m == best) { if (nl < cap) locs[nl] = -1; ++nl; }
    for (int j = 0; j < n; ++j) if (cs[static_cast<uint64_t>(j) * RTK_WAVE] == best) { if (nl < cap) locs[nl] = j; ++nl; }
    *nloc = nl;
    return 0u;
}

#endif
