// Anchor ("seed") stage of the per-read correction on the device: the work of getSeeds
// (reference: src/Graph.cpp:3-482) after the exact k-mer scan -- masking (:102-191), 1-edit k-mer search
// (:193, Bifrost searchSequence inexact [A2]), solid/weak split (:201-219), overlap filter (:221-239),
// keep_non_overlap (src/Alignment.cpp:1017-1199) and the adjacent-run consistency check (:329-372).
// One wavefront owns one long read (mask, finalize) or one tile of 64 window positions (inexact probes).
#ifndef RTK_SEEDS_H
#define RTK_SEEDS_H

#include "rtk_myers.h"
#include "rtk_sets.h"
#include "rtk_types.h"
#include "rtk_wave.h"

struct OptsView {
    U<uint32_t> insert_sz, min_cov_vertices, max_len_weak_region1, max_km_cov;
    U<double> weak_region_len_factor, large_k_factor, min_score;
    U<int32_t> max_qual, out_qual;
    U<double> min_confidence_snp_corr;
    // second pass (`correct -2`, long_read_correct in the reference): qualities of pass 1 are carried over, no 1-edit search
    U<int32_t> long_read_correct;
    U<uint32_t> max_len_weak_region2;
    U<uint32_t> d1_desc; // [D1] switch (rtk_opts::d1_desc): anchors of equal colour-set cardinality by unitig id, 0 ascending, 1 descending
    U<uint32_t> a3_strand_order; // [A3] switch (rtk_opts::a3_strand_order): 1 = neighbours of a reverse-strand end visited in the order of the unitig's own strand
    U<uint32_t> a2_exclusive; // [A2] switch (rtk_opts::a2_exclusive): 0 = union; 1, 2 = a window matched by one kind of edit is not searched with the next kind (1: substitution, insertion, deletion; 2: insertion, deletion, substitution)
};

struct BatchView {
    U<uint32_t> n_reads;
    U<uint64_t> n_bases;
    U<const char*> seq;          // upper-cased reads, concatenated
    U<const char*> qual;         // pass 2: the reads' quality strings (same offsets as seq); null in pass 1, which writes its own
    U<const uint64_t*> roff;     // [n_reads+1]
    U<const uint32_t*> order;    // [n_reads] read indices, longest first (per-read kernels start their longest items first)
    U<uint64_t*> hits;           // [n_bases] exact hit of the window starting at each base (packed, RTK_NO_HIT if none)
    U<uint64_t*> hitmap;         // [n_bases/64 + 2] bit b&63 of word b>>6: window b has an exact hit
    U<char*> masked;             // [n_bases] the 'N'-masked copy searched inexactly (src/Graph.cpp:102)
    U<uint64_t*> wdesc;          // [n_bases] group of raw inexact hits of the window: pool offset << 24 | count
    U<uint64_t*> ipool;          // raw inexact hits {k-mer code in read orientation, packed hit}
    U<uint64_t> ipool_cap;       // entries
    U<unsigned long long*> ipool_top;
    U<uint32_t*> s_pos;          // solid anchor positions of read r at [roff[r], roff[r] + n_solid[r])
    U<uint32_t*> n_solid;        // [n_reads]
    U<uint32_t*> wk_pos;         // weak anchors (all reads), read r at [w_off[r], w_off[r] + w_cnt[r])
    U<uint64_t*> wk_hit;
    U<uint64_t> wk_cap;
    U<unsigned long long*> wk_top;
    U<uint64_t*> w_off;          // [n_reads]
    U<uint32_t*> w_cnt;          // [n_reads]
    U<uint32_t*> status;         // [n_reads] non-zero: a scratch capacity was exceeded for this read
    U<unsigned long long*> counters; // [16] event counters (see rtk_pipeline.inc)
};

struct SeedScratch {
    U<uint32_t*> set[6]; U<uint32_t> set_cap;           // sorted-id buffers
    U<uint32_t*> vpos; U<uint64_t*> vcode; U<uint64_t*> vhit; U<uint64_t*> vkey; U<uint64_t*> vidx; U<uint32_t> v_cap; // weak-hit work lists (vkey/vidx hold 2*v_cap)
    U<uint32_t*> gstart; U<uint32_t*> gcnt; U<uint32_t*> gps; U<uint32_t*> gpe; U<uint8_t*> gkeep; U<uint8_t*> vflag; // variant groups
    U<uint8_t*> sflag;                               // per solid candidate
    U<uint32_t*> overflow;
};


struct SeedScratchCfg { uint32_t set_cap, v_cap, s_cap; };

RTK_HD uint64_t seed_scratch_bytes(const SeedScratchCfg& c) {
    uint64_t b = 0;
    b += 6ull * 4 * c.set_cap;                    // set[6]
    b += 4ull * c.v_cap + 8ull * c.v_cap * 2;     // vpos, vcode, vhit
    b += 2ull * 8 * (2ull * c.v_cap + 512);       // vkey, vidx (padded to a power of two for the sort)
    b += 4ull * 4 * c.v_cap + 2ull * c.v_cap;     // gstart, gcnt, gps, gpe, gkeep, vflag
    b += c.s_cap; b += 64;
    return (b + 255) / 256 * 256;
}

RTK_HD SeedScratch seed_scratch_carve(char* base, const SeedScratchCfg& c) {
    SeedScratch s; char* p = base;
    s.vcode = reinterpret_cast<uint64_t*>(p); p += 8ull * c.v_cap;
    s.vhit = reinterpret_cast<uint64_t*>(p); p += 8ull * c.v_cap;
    s.vkey = reinterpret_cast<uint64_t*>(p); p += 8ull * (2ull * c.v_cap + 512);
    s.vidx = reinterpret_cast<uint64_t*>(p); p += 8ull * (2ull * c.v_cap + 512);
    for (int i = 0; i < 6; ++i) { s.set[i] = reinterpret_cast<uint32_t*>(p); p += 4ull * c.set_cap; }
    s.set_cap = c.set_cap;
    s.vpos = reinterpret_cast<uint32_t*>(p); p += 4ull * c.v_cap; s.v_cap = c.v_cap;
    s.gstart = reinterpret_cast<uint32_t*>(p); p += 4ull * c.v_cap; s.gcnt = reinterpret_cast<uint32_t*>(p); p += 4ull * c.v_cap;
    s.gps = reinterpret_cast<uint32_t*>(p); p += 4ull * c.v_cap; s.gpe = reinterpret_cast<uint32_t*>(p); p += 4ull * c.v_cap;
    s.overflow = reinterpret_cast<uint32_t*>(p); p += 64;
    s.gkeep = reinterpret_cast<uint8_t*>(p); p += c.v_cap; s.vflag = reinterpret_cast<uint8_t*>(p); p += c.v_cap;
    s.sflag = reinterpret_cast<uint8_t*>(p);
    return s;
}

#define RTK_CNT_WINDOWS 0
#define RTK_CNT_PROBES_EXACT 1
#define RTK_CNT_PROBES_INEXACT 2
#define RTK_CNT_HITS_INEXACT 3
#define RTK_CNT_REGIONS 4
#define RTK_CNT_ITEMS 5
#define RTK_CNT_OVERFLOW 6
#define RTK_CNT_EXPAND 7
#define RTK_CNT_COLOUR 8
#define RTK_CNT_PATHBASE 9
#define RTK_CNT_ALIGN 10
#define RTK_CNT_CELLS 11
#define RTK_CNT_SLOTS_EXACT 12
#define RTK_CNT_SLOTS_INEXACT 13
#define RTK_CNT_PHASE_SKIPPED 210 // second pass: reads whose whole-read alignment was skipped (rtk_phasing.h)

RTK_DEV uint32_t rtk_hit_unitig(uint64_t h) { return static_cast<uint32_t>(h >> 33); }
RTK_DEV bool rtk_is_branching(const GraphView& g, uint32_t u) { return (g.flags[u] & RTK_F_BRANCHING) != 0; }


// min(|colours(u) & set|, cap)  (getNumberSharedPairID(SharedPairID, PairID), src/Common.cpp:73-83)
RTK_FN uint32_t rtk_shared_with_set(const GraphView& g_, uint32_t u_, const uint32_t* set_, uint32_t n_, uint32_t cap_) {
    const GraphView& g = *rtk_u(&g_); uint32_t u = rtk_u(u_); const uint32_t* set = rtk_u(set_); uint32_t n = rtk_u(n_); uint32_t cap = rtk_u(cap_);
    uint32_t shared = 0;
    const int32_t gi = g.gid[u];
    if (gi >= 0) shared = rtk_set_inter_count(g.col + g.goff[gi], static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi]), set, n, cap);
    if (shared < cap) shared += rtk_set_inter_count(g.col + g.loff[u], static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]), set, n, cap - shared);
    return shared;
}

// The `want` lowest ids of colours(u) & set, ascending, into out; returns how many (< want when the intersection is smaller). This IS
// "(global & set) | (local & set), truncated to its `want` lowest ids" (chooseColors, src/Correction.cpp:372-390) without building
// either intersection: `set` is walked from its smallest id, 64 ids per step, and the walk stops as soon as `want` are found
// (want <= 30, the sets hold hundreds to thousands of ids).
RTK_FN uint32_t rtk_first_shared(const GraphView& g_, uint32_t u_, const uint32_t* set_, uint32_t n_, uint32_t want_, uint32_t* out_) {
    const GraphView& g = *rtk_u(&g_); uint32_t u = rtk_u(u_); const uint32_t* set = rtk_u(set_); uint32_t n = rtk_u(n_); uint32_t want = rtk_u(want_); uint32_t* out = rtk_u(out_);
    if (n == 0 || want == 0) return 0;
    const int32_t gi = g.gid[u];
    const uint32_t* gl = gi >= 0 ? g.col + g.goff[gi] : nullptr; const uint32_t ngl = gi >= 0 ? static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi]) : 0u;
    const uint32_t* lo = g.col + g.loff[u]; const uint32_t nlo = static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]);
    if (ngl + nlo == 0) return 0;
#ifndef RTK_SIM
    uint32_t* const lds = rtk_lds_set_buf();
    const bool staged = (ngl + nlo) <= RTK_LDS_SET_CAP;
    if (staged) { for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < ngl + nlo; i += RTK_WAVE) lds[i] = i < ngl ? gl[i] : lo[i - ngl]; RTK_WG_SYNC(); }
#endif
    uint32_t found = 0;
    for (uint32_t i0 = 0; i0 < n && found < want; i0 += RTK_WAVE) {
        const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
        uint32_t x = 0; bool in = false;
        if (i < n) {
            x = set[i];
#ifndef RTK_SIM
            if (staged) in = rtk_lds_contains(lds, ngl, x) || rtk_lds_contains(lds + ngl, nlo, x); else
#endif
            in = (ngl && rtk_set_contains(gl, ngl, x)) || rtk_set_contains(lo, nlo, x);
        }
        const uint64_t bal = rtk_ballot(in);
        const uint32_t my = found + static_cast<uint32_t>(rtk_popc(bal & ((1ull << rtk_lane()) - 1ull)));
        if (in && my < want) out[my] = x;
        found += static_cast<uint32_t>(rtk_popc(bal));
    }
    rtk_sync();
    return found < want ? found : want;
}

// min(|colours(u) & colours(v)|, cap)  (getNumberSharedPairID(SharedPairID, SharedPairID), src/Common.cpp:51-71);
// global and local parts of one unitig are disjoint, so the four partial intersections add up
RTK_FN uint32_t rtk_shared_unitigs(const GraphView& g_, uint32_t u_, uint32_t v_, uint32_t cap_) {
    const GraphView& g = *rtk_u(&g_); uint32_t u = rtk_u(u_); uint32_t v = rtk_u(v_); uint32_t cap = rtk_u(cap_);
    const int32_t gu = g.gid[u], gv = g.gid[v];
    const uint32_t* lv = g.col + g.loff[v]; const uint32_t nlv = static_cast<uint32_t>(g.loff[v + 1] - g.loff[v]);
    if (gu >= 0 && gu == gv) {
        uint32_t shared = static_cast<uint32_t>(g.goff[gu + 1] - g.goff[gu]);
        if (shared < cap) shared += rtk_set_inter_count(g.col + g.loff[u], static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]), lv, nlv, cap - shared);
        return shared;
    }
    uint32_t shared = 0;
    if (gv >= 0) shared = rtk_shared_with_set(g, u, g.col + g.goff[gv], static_cast<uint32_t>(g.goff[gv + 1] - g.goff[gv]), cap);
    if (shared < cap) shared += rtk_shared_with_set(g, u, lv, nlv, cap - shared);
    return shared;
}

// The same count, one pair of unitigs PER LANE (no wave collectives): the junction filter of getSeeds evaluates its candidates 64 at a time.
// Sorted id arrays; the shorter is searched in the longer, the walk stops at `cap` matches.
RTK_DEV uint32_t rtk_lane_inter_count(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb, uint32_t cap) {
    if (na > nb) { const uint32_t* t = a; a = b; b = t; const uint32_t tn = na; na = nb; nb = tn; }
    uint32_t cnt = 0, lo = 0;
    for (uint32_t i = 0; i < na && cnt < cap && lo < nb; ++i) {
        const uint32_t x = a[i];
        uint32_t l = lo, h = nb;
        while (l < h) { const uint32_t m = (l + h) >> 1; if (b[m] < x) l = m + 1; else h = m; }
        lo = l;
        if (lo < nb && b[lo] == x) { ++cnt; ++lo; }
    }
    return cnt;
}
RTK_DEV uint32_t rtk_lane_shared_unitigs(const GraphView& g, uint32_t u, uint32_t v, uint32_t cap) { // min(|colours(u) & colours(v)|, cap), as rtk_shared_unitigs
    const int32_t gu = g.gid[u], gv = g.gid[v];
    const uint32_t* lu = g.col + g.loff[u]; const uint32_t nlu = static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]);
    const uint32_t* lv = g.col + g.loff[v]; const uint32_t nlv = static_cast<uint32_t>(g.loff[v + 1] - g.loff[v]);
    if (gu >= 0 && gu == gv) {
        const uint64_t ng = g.goff[gu + 1] - g.goff[gu];
        if (ng >= cap) return cap;
        return static_cast<uint32_t>(ng) + rtk_lane_inter_count(lu, nlu, lv, nlv, cap - static_cast<uint32_t>(ng));
    }
    const uint32_t* gul = gu >= 0 ? g.col + g.goff[gu] : nullptr; const uint32_t ngu = gu >= 0 ? static_cast<uint32_t>(g.goff[gu + 1] - g.goff[gu]) : 0u;
    const uint32_t* gvl = gv >= 0 ? g.col + g.goff[gv] : nullptr; const uint32_t ngv = gv >= 0 ? static_cast<uint32_t>(g.goff[gv + 1] - g.goff[gv]) : 0u;
    uint32_t shared = 0; // the four partial intersections add up (global and local part of one unitig are disjoint)
    if (ngu && ngv) shared += rtk_lane_inter_count(gul, ngu, gvl, ngv, cap);
    if (shared < cap && ngv && nlu) shared += rtk_lane_inter_count(lu, nlu, gvl, ngv, cap - shared);
    if (shared < cap && ngu && nlv) shared += rtk_lane_inter_count(gul, ngu, lv, nlv, cap - shared);
    if (shared < cap && nlu && nlv) shared += rtk_lane_inter_count(lu, nlu, lv, nlv, cap - shared);
    return shared;
}

// colours(u) = global | local, merged into the running union held in sc.set[cur]; returns new size (0xFFFFFFFF on overflow)
RTK_FN uint32_t rtk_union_unitig(const GraphView& g_, const SeedScratch& sc_, int& cur_, uint32_t n_cur_, uint32_t u_) {
    const GraphView& g = *rtk_u(&g_); const SeedScratch& sc = *rtk_u(&sc_); RTK_ASSUME_LDS(&sc); int& cur = *rtk_u(&cur_); uint32_t n_cur = rtk_u(n_cur_); uint32_t u = rtk_u(u_);
    const int32_t gi = g.gid[u];
    if (gi >= 0) {
        const uint32_t ng = static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi]);
        if (n_cur + ng > sc.set_cap) { *sc.overflow = 1; return 0xFFFFFFFFu; }
        n_cur = rtk_set_union(sc.set[cur], n_cur, g.col + g.goff[gi], ng, sc.set[cur ^ 1], sc.set[2]);
        cur ^= 1;
    }
    const uint32_t nl = static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]);
    if (nl) {
        if (n_cur + nl > sc.set_cap) { *sc.overflow = 1; return 0xFFFFFFFFu; }
        n_cur = rtk_set_union(sc.set[cur], n_cur, g.col + g.loff[u], nl, sc.set[cur ^ 1], sc.set[2]);
        cur ^= 1;
    }
    return n_cur;
}

// ---------------------------------------------------------------------------------------------- mask (src/Graph.cpp:102-191)
// presence bits of the 64 windows starting at global position g0 (bit j = window g0 + j), limited to the first n_valid of them
RTK_DEV uint64_t rtk_hit_bits(const uint64_t* map, uint64_t g0, uint64_t n_valid) {
    if (n_valid == 0) return 0ull;
    const uint64_t w0 = map[g0 >> 6], sh = g0 & 63ull;
    uint64_t bits = w0 >> sh;
    if (sh) bits |= map[(g0 >> 6) + 1] << (64ull - sh);
    return n_valid >= 64 ? bits : (bits & ((1ull << n_valid) - 1ull));
}

// Visits the exact hits at windows lo .. hi (inclusive, read coordinates) in ascending (dir > 0) or descending order and calls
// fn(unitig) at every change of unitig (src/Graph.cpp:127-183 walks the hits one by one and reacts to unitig changes only).
// 64 windows per step: presence bits from the bitmap, the hits of the set bits fetched by their lanes, changes found with a ballot.
template <class Fn>
RTK_DEV void rtk_scan_hit_runs(const uint64_t* hits, const uint64_t* hmap, uint64_t base, uint32_t nwin, int64_t lo, int64_t hi, int dir, Fn fn) {
    uint32_t prev_u = RTK_NONE32;
    if (hi >= static_cast<int64_t>(nwin)) hi = static_cast<int64_t>(nwin) - 1;
    if (lo < 0) lo = 0;
    for (int64_t done = 0; lo + done <= hi; done += RTK_WAVE) {
        const int64_t x = dir > 0 ? (lo + done + rtk_lane()) : (hi - done - rtk_lane());
        bool valid = x >= lo && x <= hi;
        if (valid) { const uint64_t g = base + static_cast<uint64_t>(x); valid = (hmap[g >> 6] >> (g & 63ull)) & 1ull; }
        uint32_t u = RTK_NONE32;
        if (valid) u = rtk_hit_unitig(hits[x]);
        const uint64_t m = rtk_ballot(valid);
        if (!m) continue;
        const uint64_t pm = m & ((1ull << rtk_lane()) - 1ull);
        const uint32_t up = rtk_shfl(u, pm ? (63 - __builtin_clzll(pm)) : 0);
        const uint32_t u_before = pm ? up : prev_u;
        uint64_t st = rtk_ballot(valid && (u_before == RTK_NONE32 || u != u_before));
        while (st) {
            const int l = rtk_ffs(st) - 1; st &= st - 1ull;
            if (!fn(rtk_u(rtk_shfl(u, l)))) return;
        }
        prev_u = rtk_u(rtk_shfl(u, 63 - __builtin_clzll(m)));
    }
}

// Windows [w_lo, w_hi) of read r (both multiples of the segment length, itself a multiple of 64 * RTK_WAVE; the kernel cuts long reads into such segments, one wave each: the launch lasted as
// long as the longest read of the ticket). The masked copy is all 'N' before the launch (run_seed_stage), so that a segment may open a gap that reaches
// back into the segment before it. A gap [previous hit, p) belongs to the segment that holds its closing hit p; what the walk of src/Graph.cpp:127-183
// carries along -- the read's first hit and the last hit before the segment -- is read off the presence bits. The rule for the read's head is applied
// by the segment that holds the first hit, the rule for its tail by the last segment.
RTK_FN void rtk_mask_read(const GraphView& g, const OptsView& o, const BatchView& bv, const SeedScratch& sc, uint32_t r, uint32_t w_lo, uint32_t w_hi) {
    RTK_ASSUME_LDS(&sc);
    const uint64_t base = bv.roff[r];
    const uint32_t L = static_cast<uint32_t>(bv.roff[r + 1] - base);
    const uint32_t k = static_cast<uint32_t>(g.k);
    if (L <= k) return; // src/Graph.cpp:49
    const uint32_t nwin = L - k + 1;
    if (w_lo >= nwin) return;
    const bool last_seg = w_hi >= nwin;
    if (w_hi > nwin) w_hi = nwin;
    const uint64_t* hits = bv.hits + base;
    int64_t prev = -1, first = -1;
    const uint64_t* hmap = bv.hitmap;
    if (w_lo > 0) { // the first hit of the read and the last one before this segment
        for (uint32_t cc = 0; cc < w_lo && first < 0; cc += 64 * RTK_WAVE) {
            const uint32_t my_c0 = cc + 64u * static_cast<uint32_t>(rtk_lane());
            const uint64_t my_bits = (my_c0 < w_lo) ? rtk_hit_bits(hmap, base + my_c0, 64) : 0ull;
            const uint64_t any = rtk_ballot(my_bits != 0);
            if (any) { const int l = rtk_ffs(any) - 1; first = static_cast<int64_t>(cc) + 64 * l + __builtin_ctzll(rtk_u(rtk_shfl(my_bits, l))); }
        }
        for (uint32_t hi = w_lo; first >= 0 && hi > 0 && prev < 0; hi -= 64 * RTK_WAVE) { // blocks hi - 64, hi - 128, ...: lane 0 holds the nearest one
            const uint32_t my_c0 = hi - 64u * (static_cast<uint32_t>(rtk_lane()) + 1u);
            const uint64_t my_bits = rtk_hit_bits(hmap, base + my_c0, 64);
            const uint64_t any = rtk_ballot(my_bits != 0);
            if (any) { const int l = rtk_ffs(any) - 1; prev = static_cast<int64_t>(hi) - 64 * (l + 1) + 63 - __builtin_clzll(rtk_u(rtk_shfl(my_bits, l))); }
        }
    }
    const bool first_is_mine = first < 0;
    for (uint32_t cc = w_lo; cc < w_hi; cc += 64 * RTK_WAVE) { // every lane fetches the presence bits of one 64-window block, then the blocks are visited in order
        const uint32_t my_c0 = cc + 64u * static_cast<uint32_t>(rtk_lane());
        const uint64_t my_bits = (my_c0 < w_hi) ? rtk_hit_bits(hmap, base + my_c0, w_hi - my_c0) : 0ull;
      for (int bl = 0; bl < RTK_WAVE; ++bl) {
        const uint32_t c0 = cc + 64u * static_cast<uint32_t>(bl);
        if (c0 >= w_hi) break;
        uint64_t bal = rtk_u(rtk_shfl(my_bits, bl));
        if (bal == ~0ull && prev == static_cast<int64_t>(c0) - 1) { if (first < 0) first = c0; prev = c0 + 63; continue; } // inside a run: no gap
        // only the first hit of a run can close a gap: visit run starts, with `prev` = the last hit before each of them
        const uint64_t all_hits = bal; const int64_t prev_in = prev;
        uint64_t starts = bal & ~((bal << 1) | ((prev_in >= 0 && prev_in == static_cast<int64_t>(c0) - 1) ? 1ull : 0ull));
        if (all_hits) prev = static_cast<int64_t>(c0) + 63 - __builtin_clzll(all_hits);
        while (starts) {
            const int bit = rtk_ffs(starts) - 1;
            const uint32_t p = c0 + static_cast<uint32_t>(bit);
            starts &= starts - 1ull;
            const uint64_t below = all_hits & ((1ull << bit) - 1ull);
            const int64_t prev_hit = below ? (static_cast<int64_t>(c0) + 63 - __builtin_clzll(below)) : prev_in;
            if (prev_hit >= 0) {
                const uint32_t pv = static_cast<uint32_t>(prev_hit);
                const uint32_t diff = p - pv;
                bool unmask = false;
                if (diff >= o.insert_sz) unmask = true;
                else if (diff >= o.insert_sz / 2) {
                    const uint32_t ssl = o.insert_sz - diff;
                    const uint32_t min_pos_left = (pv < ssl) ? 0u : (pv - ssl);
                    const uint64_t max_pos_right = static_cast<uint64_t>(p) + ssl;
                    int cur = 0; uint32_t nL = 0, nR = 0; bool ovf = false;
                    // left of the gap: hits at pv, pv-1, ... above min_pos_left, stopping before the read's first hit (G13: index 0 is never visited)
                    const int64_t lb = (first > static_cast<int64_t>(min_pos_left)) ? first : static_cast<int64_t>(min_pos_left);
                    rtk_scan_hit_runs(hits, hmap, base, nwin, lb + 1, static_cast<int64_t>(pv), -1, [&](uint32_t u) {
                        if (!rtk_is_branching(g, u)) { nL = rtk_union_unitig(g, sc, cur, nL, u); if (nL == 0xFFFFFFFFu) { ovf = true; return false; } }
                        return true; });
                    if (!ovf) { // park the left union in set[3]
                        if (nL > sc.set_cap) ovf = true; else rtk_wcopy(sc.set[3], sc.set[cur], 4ull * nL);
                    }
                    cur = 0;
                    if (!ovf) rtk_scan_hit_runs(hits, hmap, base, nwin, static_cast<int64_t>(p), static_cast<int64_t>(max_pos_right) - 1, +1, [&](uint32_t u) {
                        if (!rtk_is_branching(g, u)) { nR = rtk_union_unitig(g, sc, cur, nR, u); if (nR == 0xFFFFFFFFu) { ovf = true; return false; } }
                        return true; });
                    if (!ovf) unmask = rtk_set_inter_count(sc.set[3], nL, sc.set[cur], nR, o.min_cov_vertices) < o.min_cov_vertices;
                }
                if (unmask) rtk_wcopy(bv.masked + base + pv + k, bv.seq + base + pv + k, diff - k);
            }
            if (first < 0) first = p;
        }
      }
    }
    if (first >= 0) {
        if (first_is_mine && static_cast<uint64_t>(first) >= o.insert_sz / 2) rtk_wcopy(bv.masked + base, bv.seq + base, static_cast<uint64_t>(first) + k - 1);
        if (last_seg && L - static_cast<uint64_t>(prev) >= o.insert_sz / 2) rtk_wcopy(bv.masked + base + prev + 1, bv.seq + base + prev + 1, L - static_cast<uint64_t>(prev) - 1);
    }
}

// ---------------------------------------------------------------------------------------------- inexact probes (src/Graph.cpp:193 [A2])
// One tile = 64 consecutive base positions; every candidate window of the tile is expanded by the whole wave into its
// 93 substitution + 124 "insertion" + 29 "deletion" variants (one variant per lane per round), each probed in the k-mer table.
#define RTK_N_VARIANTS 246
// kinds of 1-edit relation between a window and a graph k-mer ([A2] switch: rtk_opts::a2_exclusive)
#define RTK_EDIT_SUB 1u
#define RTK_EDIT_INS 2u
#define RTK_EDIT_DEL 4u

// variant v (0..245) of the window whose first k-1 / k / k+1 characters are (w_k1, w_ck, w_ck1): 2-bit code of the graph k-mer to look for
RTK_DEV bool rtk_variant_code(int v, int k, uint64_t w_k1, uint32_t w_ck, uint32_t w_ck1, uint64_t* code_out) {
    uint64_t code = 0; bool valid = false;
    if (v < 93) { // substitution: needs k characters
        if (w_ck <= 3) {
            const int oo = v / 3, j = v % 3;
            const uint64_t full = (w_k1 << 2) | w_ck;
            const int sh = 2 * (k - 1 - oo);
            const uint32_t orig = static_cast<uint32_t>((full >> sh) & 3ull);
            const uint32_t nb = static_cast<uint32_t>(j) + (static_cast<uint32_t>(j) >= orig ? 1u : 0u);
            code = (full & ~(3ull << sh)) | (static_cast<uint64_t>(nb) << sh); valid = true;
        }
    } else if (v < 217) { // graph k-mer has one extra base: k-1 read characters + an inserted one
        const int vv = v - 93, oo = vv / 4; const uint64_t nb = static_cast<uint64_t>(vv % 4);
        const int rest = k - 1 - oo; // characters after the inserted base
        const uint64_t lo_mask = rest ? ((1ull << (2 * rest)) - 1ull) : 0ull;
        code = ((w_k1 >> (2 * rest)) << (2 * rest + 2)) | (nb << (2 * rest)) | (w_k1 & lo_mask); valid = true;
        // inserting b in front of a b spells the same k-mer as inserting it behind that b: only the last offset of such a run is probed
        // (the hits of a window are reduced to distinct k-mers anyway, src/Graph.cpp:201-216)
        if (rest >= 1 && ((w_k1 >> (2 * (rest - 1))) & 3ull) == nb) valid = false;
    } else if (v < RTK_N_VARIANTS) { // graph k-mer lacks one interior read base: k+1 read characters
        if (w_ck <= 3 && w_ck1 <= 3) {
            const int oo = v - 217 + 1; // deleted offset in [1, k-2]
            const uint64_t full2 = (w_k1 << 4) | (static_cast<uint64_t>(w_ck) << 2) | static_cast<uint64_t>(w_ck1); // k+1 characters (exactly 64 bits for k = 31)
            const int keep_lo = k - oo; // characters after the deleted one
            const uint64_t lo_mask = (1ull << (2 * keep_lo)) - 1ull;
            const uint64_t hi = (2 * (keep_lo + 1) >= 64) ? 0ull : (full2 >> (2 * (keep_lo + 1)));
            code = (hi << (2 * keep_lo)) | (full2 & lo_mask); valid = true;
            // deleting either of two equal neighbours spells the same k-mer: only the last offset of a run is probed
            if (oo <= k - 3 && ((full2 >> (2 * (k - oo))) & 3ull) == ((full2 >> (2 * (k - oo - 1))) & 3ull)) valid = false;
        }
    }
    *code_out = code & ((k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull));
    return valid;
}

struct PoolChunk { unsigned long long base; uint32_t left; }; // wave-private slice of the raw-hit pool (one device atomic per 4096 entries)
#define RTK_POOL_CHUNK 4096u

// [D1] sort key of a candidate anchor of chooseColors: (cardinality, unitig id), the id inverted for the descending reading; and the id back out of a key
RTK_DEV uint64_t rtk_d1_key(uint32_t card, uint32_t u, uint32_t desc) { return (static_cast<uint64_t>(card) << 32) | (desc ? (~u & 0xFFFFFFFFu) : u); }
RTK_DEV uint32_t rtk_d1_unitig(uint64_t key, uint32_t desc) { const uint32_t lo = static_cast<uint32_t>(key & 0xFFFFFFFFull); return desc ? ~lo : lo; }
RTK_DEV uint32_t rtk_variant_kind(int v) { return v < 93 ? RTK_EDIT_SUB : (v < 217 ? RTK_EDIT_INS : RTK_EDIT_DEL); }
// [A2] exclusive readings: of the kinds of edit that matched a window (`any`), the one whose hits are kept. mode 1: substitution -> insertion -> deletion
// (the order of the blocks in Bifrost's searchSequence as we remember it); mode 2: insertion -> deletion -> substitution (the order of its parameters)
RTK_DEV uint32_t rtk_a2_keep(uint32_t any, uint32_t mode) {
    if (mode == 2u) return (any & RTK_EDIT_INS) ? RTK_EDIT_INS : ((any & RTK_EDIT_DEL) ? RTK_EDIT_DEL : (any & RTK_EDIT_SUB));
    return any & (0u - any);
}
RTK_FN void rtk_inexact_tile(const GraphView& g, const BatchView& bv, uint64_t tile, unsigned long long* acc_probes, unsigned long long* acc_slots, unsigned long long* acc_hits, PoolChunk* chunk, uint32_t exclusive) {
    const int k = g.k;
#ifndef RTK_SIM
    const uint64_t b = tile * 64 + static_cast<uint64_t>(rtk_lane());
    // LDS staging of the tile's 64 + k + 1 characters of the masked read: one coalesced load, then every lane slides over its own window
    __shared__ unsigned char tile_chars[128];
    {
        const uint64_t a0 = tile * 64 + static_cast<uint64_t>(rtk_lane()), a1 = a0 + 64;
        tile_chars[rtk_lane()] = (a0 < bv.n_bases) ? static_cast<unsigned char>(bv.masked[a0]) : 'N';
        tile_chars[64 + rtk_lane()] = (a1 < bv.n_bases) ? static_cast<unsigned char>(bv.masked[a1]) : 'N';
        RTK_WG_SYNC();
    }
#endif
#ifdef RTK_SIM
    const int n_sub = 64;
#else
    const int n_sub = 1;
#endif
    const uint64_t* const roff = bv.roff; const uint32_t n_reads = bv.n_reads;
    uint32_t lo_tile = 0; // one scalar search per tile: largest r with roff[r] <= first base of the tile
    lo_tile = rtk_owner_read(roff, n_reads, tile * 64);
    for (int sub = 0; sub < n_sub; ++sub) { // the 1-lane simulator visits the 64 positions of the tile one after the other
#ifdef RTK_SIM
        const uint64_t bb = tile * 64 + static_cast<uint64_t>(sub);
#else
        const uint64_t bb = b;
#endif
        // per-lane: is the window starting at bb a candidate? how many usable characters follow (k-1, k or k+1)?
        bool cand = false; uint64_t c_k1 = 0; uint32_t ck = 4, ck1 = 4;
        if (bb < bv.n_bases) {
            uint32_t lo = lo_tile; // owning read: steps forward from the read of the tile's first base
            while (lo + 1 < n_reads && roff[lo + 1] <= bb) ++lo;
            const uint64_t rend = roff[lo + 1];
            if (bb + static_cast<uint64_t>(k) <= rend && rend - roff[lo] > static_cast<uint64_t>(k)) {
                bool ok = true;
#ifdef RTK_SIM
                const unsigned char* wc = reinterpret_cast<const unsigned char*>(bv.masked.get()) + bb;
#else
                const unsigned char* wc = tile_chars + rtk_lane(); // k + 1 <= 64: the window and its two look-ahead characters are inside the staged 128 bytes
#endif
                for (int i = 0; i < k - 1; ++i) { const int c = rtk_cls(wc[i]); if (c > 3) { ok = false; break; } c_k1 = (c_k1 << 2) | static_cast<uint64_t>(c); }
                if (ok) {
                    cand = true;
                    const int c = rtk_cls(wc[k - 1]);
                    if (c <= 3) { ck = static_cast<uint32_t>(c); if (bb + static_cast<uint64_t>(k) + 1 <= rend) { const int c2 = rtk_cls(wc[k]); if (c2 <= 3) ck1 = static_cast<uint32_t>(c2); } }
                }
            }
        }
        uint64_t bal = rtk_ballot(cand);
        // appends the `total` hits of window w_b (lane-local lists my_code / my_hit, my_n entries) to the raw-hit pool
        auto emit = [&](uint64_t w_b, int total, const uint64_t* my_code, const uint64_t* my_hit, int my_n, uint32_t probes, uint32_t slots) {
            if (total > 0) {
                if (chunk->left < static_cast<uint32_t>(total)) { // refill the wave's private slice (the tail of the old slice is abandoned)
                    unsigned long long nb = 0;
                    if (rtk_lane() == 0) nb = rtk_atomic_add(bv.ipool_top, static_cast<unsigned long long>(RTK_POOL_CHUNK));
                    chunk->base = rtk_shfl(nb, 0); chunk->left = RTK_POOL_CHUNK;
                }
                const unsigned long long pbase = chunk->base;
                chunk->base += static_cast<unsigned long long>(total); chunk->left -= static_cast<uint32_t>(total);
                if (pbase + static_cast<unsigned long long>(total) <= bv.ipool_cap) {
                    int tot2; const int off = rtk_wave_excl_scan(my_n, &tot2);
                    for (int i = 0; i < my_n; ++i) { bv.ipool[2 * (pbase + off + i)] = my_code[i]; bv.ipool[2 * (pbase + off + i) + 1] = my_hit[i]; }
                    if (rtk_lane() == 0) bv.wdesc[w_b] = (static_cast<uint64_t>(pbase) << 24) | static_cast<uint64_t>(total);
                } else if (rtk_lane() == 0) rtk_atomic_add(bv.counters + RTK_CNT_OVERFLOW, 1ull);
            }
            *acc_slots += slots;
            *acc_probes += probes; // per-lane tallies, reduced once per wave at kernel end (a device-wide atomic per window saturates one L2 word)
            if (rtk_lane() == 0) *acc_hits += static_cast<unsigned long long>(total);
        };
#ifdef RTK_SIM
        while (bal) { // simulator: one lane walks all variants of its window
            bal &= bal - 1ull;
            uint64_t sim_code[RTK_N_VARIANTS], sim_hit[RTK_N_VARIANTS]; uint32_t sim_kind[RTK_N_VARIANTS]; int total = 0; uint32_t probes = 0, slots = 0, any = 0;
            for (int v = 0; v < RTK_N_VARIANTS; ++v) {
                uint64_t code; uint64_t hit = RTK_NO_HIT;
                if (rtk_variant_code(v, k, c_k1, ck, ck1, &code)) { uint32_t np; hit = rtk_find_kmer(g, code, &np); probes += 1; slots += np; }
                if (hit != RTK_NO_HIT) { sim_code[total] = code; sim_hit[total] = hit; sim_kind[total] = rtk_variant_kind(v); any |= sim_kind[total]; ++total; }
            }
            if (exclusive && total) { const uint32_t keep = rtk_a2_keep(any, exclusive); int w2 = 0; for (int i = 0; i < total; ++i) if (sim_kind[i] & keep) { sim_code[w2] = sim_code[i]; sim_hit[w2] = sim_hit[i]; ++w2; } total = w2; }
            emit(bb, total, sim_code, sim_hit, total, probes, slots);
        }
#else
        // Two-stage software pipeline over the candidate windows of the tile. Per window: four rounds of 64 variants; all four
        // pre-filter words are requested together, then the first table slot (key + value, one 16-byte read) of every variant that
        // passes. Those slot reads stay in flight while the NEXT window's variants are generated and its filter words requested.
        // (A third stage - filter words of window k+2 behind the slots of window k+1 - measured slower: 21.7 vs 19.6 ms per 32 Mb.)
        // A window in flight is remembered by its three scalar words; the k-mers of the few variants that pass the filter are spelled
        // again when their slot arrives, so that only the slot contents wait in vector registers.
        struct Win { uint64_t w_b, w_k1; uint32_t w_ck, w_ck1; };
        const uint64_t* const ht = g.ht; const uint64_t ht_slots = g.ht_slots; const uint64_t* const bf = g.bf; const uint64_t bf_mask = g.bf_mask; const uint64_t* const bf1 = g.bf1; const uint64_t bf1_mask = g.bf1_mask;
        auto stage_probe = [&](Win& wn, uint32_t& pass, uint64_t* skey, uint64_t* sval) { // takes the next candidate off `bal`
            const int sl = rtk_ffs(bal) - 1;
            bal &= bal - 1ull;
            wn.w_k1 = rtk_u(rtk_shfl(c_k1, sl)); wn.w_ck = rtk_u(rtk_shfl(ck, sl)); wn.w_ck1 = rtk_u(rtk_shfl(ck1, sl));
            wn.w_b = tile * 64 + static_cast<uint64_t>(sl);
            uint64_t hh[4], word[4]; bool valid[4], valid1[4]; uint32_t probes = 0;
            for (int rr = 0; rr < 4; ++rr) {
                uint64_t code, can; uint32_t q;
                valid[rr] = rtk_variant_code(rtk_lane() + 64 * rr, k, wn.w_k1, wn.w_ck, wn.w_ck1, &code);
                rtk_kmer_prepare(code, k, &can, &hh[rr], &q);
                probes += valid[rr] ? 1u : 0u;
            }
            // first-level bit (L2 resident), then the filter word only for the variants whose bit is set
            uint64_t w1[4];
            for (int rr = 0; rr < 4; ++rr) w1[rr] = valid[rr] ? bf1[((hh[rr] >> 12) & bf1_mask) >> 6] : 0ull;
            for (int rr = 0; rr < 4; ++rr) valid1[rr] = valid[rr] && rtk_filter1_pass(w1[rr], hh[rr], bf1_mask);
            for (int rr = 0; rr < 4; ++rr) word[rr] = valid1[rr] ? bf[(hh[rr] >> 32) & bf_mask] : 0ull;
            *acc_probes += probes; // per-lane tallies, reduced once per wave at kernel end (a device-wide atomic per window saturates one L2 word)
            pass = 0;
            for (int rr = 0; rr < 4; ++rr) {
                const bool ps = valid1[rr] && rtk_filter_pass(word[rr], hh[rr]);
                skey[rr] = RTK_EMPTY_KEY; sval[rr] = 0;
                if (ps) { pass |= 1u << rr; const uint64_t* sp = ht + 2 * rtk_ht_slot(hh[rr], ht_slots); skey[rr] = sp[0]; sval[rr] = sp[1]; }
            }
        };
        auto stage_resolve = [&](const Win& wn, uint32_t pass, const uint64_t* skey, const uint64_t* sval) {
            uint64_t my_code[4]; uint64_t my_hit[4]; uint32_t my_kind[4]; int my_n = 0; uint32_t slots = 0; int total = 0; uint32_t any = 0;
            for (int rr = 0; rr < 4; ++rr) {
                uint64_t hit = RTK_NO_HIT; uint64_t code = 0;
                if ((pass >> rr) & 1u) {
                    uint64_t can, hh; uint32_t q;
                    rtk_variant_code(rtk_lane() + 64 * rr, k, wn.w_k1, wn.w_ck, wn.w_ck1, &code);
                    rtk_kmer_prepare(code, k, &can, &hh, &q);
                    slots += 1;
                    if (skey[rr] == can) hit = rtk_pack_hit(static_cast<uint32_t>(sval[rr] >> 32), static_cast<uint32_t>((sval[rr] & 0xFFFFFFFFull) >> 1), (static_cast<uint32_t>(sval[rr] & 1ull) == q) ? 1u : 0u);
                    else if (skey[rr] != RTK_EMPTY_KEY) { uint32_t np; hit = rtk_table_probe_from(g, can, rtk_ht_next(rtk_ht_slot(hh, ht_slots), ht_slots), q, &np); slots += np; } // collision: keep probing from the next slot
                }
                const uint64_t hb = rtk_ballot(hit != RTK_NO_HIT);
                if (hit != RTK_NO_HIT) { my_code[my_n] = code; my_hit[my_n] = hit; my_kind[my_n] = rtk_variant_kind(rtk_lane() + 64 * rr); any |= my_kind[my_n]; ++my_n; }
                total += rtk_popc(hb);
            }
            if (exclusive && total) { // [A2] exclusive: only the hits of the first kind of edit (substitution -> insertion -> deletion) that has one in this window
                for (int o = 32; o > 0; o >>= 1) any |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(any), o, 64));
                const uint32_t keep = rtk_a2_keep(any, exclusive);
                int w2 = 0; for (int i = 0; i < my_n; ++i) if (my_kind[i] & keep) { my_code[w2] = my_code[i]; my_hit[w2] = my_hit[i]; ++w2; }
                my_n = w2; total = rtk_wave_sum(my_n);
            }
            emit(wn.w_b, total, my_code, my_hit, my_n, 0u, slots);
        };
        if (bal) {
            Win cur; uint32_t pass; uint64_t skey[4], sval[4];
            stage_probe(cur, pass, skey, sval);
            while (bal) {
                Win nxt; uint32_t npass; uint64_t nkey[4], nval[4];
                stage_probe(nxt, npass, nkey, nval);   // next window: filter words, then its slots, on their way ...
                stage_resolve(cur, pass, skey, sval);  // ... while this window's slots (requested a window ago) are compared
                cur = nxt; pass = npass;
                for (int rr = 0; rr < 4; ++rr) { skey[rr] = nkey[rr]; sval[rr] = nval[rr]; }
            }
            stage_resolve(cur, pass, skey, sval);
        }
#endif
    }
}

// ---------------------------------------------------------------------------------------------- 1-edit search by half-k-mer seeds ([A2], same hit sets as above)
// A graph k-mer G one edit away from a read window keeps its first h or its last h characters (h = (k-1)/2, the middle character
// belongs to neither half): the edit cannot touch both. So instead of spelling the ~246 variants of a window and probing each, a lane
// looks up the read h-mers that would be G's first half (1 key) or G's last half (3 keys: substitution, graph-has-extra-base, graph-
// lacks-a-base shift the second half by 0 / -1 / +1) in the index of all unitig h-mers (GraphView::hx), in both orientations, and
// checks the k-mers those h-mers belong to with three XOR / count-leading-zero tests. One lane per window.

RTK_DEV uint64_t rtk_rev2(uint64_t x, int n) { // reverses the order of the n 2-bit groups held in the low 2n bits
    uint64_t y = rtk_brev64(x);
    y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
    return (n >= 32) ? y : (y >> (64 - 2 * n));
}
RTK_DEV uint64_t rtk_pool_kmer(const GraphView& g, uint64_t gp, int k) { // 2-bit code (first base in the high bits) of the k bases of the unitig pool at gp
    const uint64_t w0 = g.useq[gp >> 5], w1 = g.useq[(gp >> 5) + 1];
    const int sh = static_cast<int>(2 * (gp & 31ull));
    uint64_t lsb = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
    if (k < 32) lsb &= (1ull << (2 * k)) - 1ull;
    return rtk_rev2(lsb, k);
}
RTK_DEV int rtk_clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
RTK_DEV int rtk_ctz64(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }

// which 1-edit variants of the window whose first k-1 characters are w_k1 (and ck, ck1 the next two, 4 = unusable) is G? (oracle: searchInexact)
// bit 0 substitution, bit 1 graph k-mer has an extra base ("insertion"), bit 2 graph k-mer lacks a read base ("deletion"); 0 = none.
// want = the kinds that matter to the caller: the tests stop at the first of them that holds when only "any" is asked for.
RTK_DEV uint32_t rtk_one_edit(uint64_t G, int k, uint64_t w_k1, uint32_t ck, uint32_t ck1, bool all_kinds) {
    const uint64_t mk = (k < 32) ? ((1ull << (2 * k)) - 1ull) : ~0ull, mk1 = (1ull << (2 * (k - 1))) - 1ull;
    uint32_t kinds = 0;
    if (ck <= 3) { // substitution: exactly one differing character
        const uint64_t x = (G ^ ((w_k1 << 2) | static_cast<uint64_t>(ck))) & mk;
        if (rtk_popc((x | (x >> 1)) & 0x5555555555555555ull) == 1) { kinds |= RTK_EDIT_SUB; if (!all_kinds) return kinds; }
    }
    { // G = the k-1 read characters with one base inserted anywhere: common prefix + common suffix cover the k-1 characters
        const uint64_t xp = ((G >> 2) ^ w_k1) & mk1, xs = (G ^ w_k1) & mk1;
        const int lcp = xp ? (rtk_clz64(xp) - (64 - 2 * (k - 1))) / 2 : k - 1, lcs = xs ? rtk_ctz64(xs) / 2 : k - 1;
        if (lcp + lcs >= k - 1) { kinds |= RTK_EDIT_INS; if (!all_kinds) return kinds; }
    }
    if (ck <= 3 && ck1 <= 3) { // G = the k+1 read characters without one interior character (offset 1..k-2)
        const uint64_t S = (w_k1 << 4) | (static_cast<uint64_t>(ck) << 2) | static_cast<uint64_t>(ck1);
        const uint64_t xp = (G ^ (S >> 2)) & mk, xs = (G ^ S) & mk;
        const int lcp = xp ? (rtk_clz64(xp) - (64 - 2 * k)) / 2 : k, lcs = xs ? rtk_ctz64(xs) / 2 : k;
        const int lo = (k - lcs) > 1 ? (k - lcs) : 1, hi = lcp < (k - 2) ? lcp : (k - 2);
        if (lo <= hi) kinds |= RTK_EDIT_DEL;
    }
    return kinds;
}

// visits (code, hit) of every graph k-mer that is a 1-edit variant of the window and contains one of its seed h-mers; a k-mer reached
// through two seeds is visited twice (the hits of a window are reduced to distinct k-mers downstream, src/Graph.cpp:201-216)
template <class V>
RTK_DEV void rtk_seeded_window(const GraphView& g, int k, uint64_t w_k1, uint32_t ck, uint32_t ck1, uint32_t* n_lookups, uint32_t* n_slots, bool all_kinds, V visit) { // visit(code, hit, kinds)
    const int h = (k - 1) / 2;
    const uint64_t hm = (1ull << (2 * h)) - 1ull;
    const uint64_t* const hx = g.hx; const uint64_t hx_mask = g.hx_mask; const uint64_t* const hxl = g.hxl;
    // the window's characters as one code of k + 1 (ck / ck1 = 0 where the read has none: those keys are not asked for). The four seed h-mers are shifts of it:
    //   q = 0  m[p .. p+h)              first half of G
    //   q = 1  m[p+k-1-h .. p+k-1)      last half when the graph k-mer has an extra base
    //   q = 2  m[p+k-h .. p+k)          last half, substitution              (needs ck)
    //   q = 3  m[p+k+1-h .. p+k+1)      last half, a read base missing in G  (needs ck and ck1)
    // Round 6: the keys are computed where they are used. As `key[4]` / `first_half[4]` indexed by the loop variable they lived in the lane's stack: 36 bytes stored per
    // window = the 3.3 GB a launch wrote whatever the graph (WRITE_SIZE of round 5), and the kernel's 160 bytes of scratch per lane with the hit registers below.
    const uint64_t S_all = (w_k1 << 4) | (static_cast<uint64_t>(ck <= 3 ? ck : 0u) << 2) | static_cast<uint64_t>(ck1 <= 3 ? ck1 : 0u);
    const int nkeys = (ck <= 3) ? ((ck1 <= 3) ? 4 : 3) : 2;
    // (measured in round 5: all eight home slots, then all count words, in flight together -- three rounds of independent loads instead of eight chains -- is SLOWER,
    // 5.8 against 4.7 ms per 64 Mb on the 60 Mb graph: the lookups of six waves per SIMD already overlap, the arrays of the batched form spill)
    // The index is keyed by the CANONICAL h-mer (the smaller of an h-mer and its reverse complement; round 5): one look-up per read h-mer finds the places where it
    // stands in a unitig as it is AND those where its reverse complement does (bit 63 of a place: the unitig holds the reverse complement of the key). Before, the two
    // orientations were two look-ups -- two random lines of the table each. (Also measured in round 5: every read position of a tile looked up once and the list starts shared
    // by the up to four windows that use them, through LDS -- look-ups 47 M -> 15.7 M per 64 Mb, and the kernel SLOWER, 3.44 -> 4.08 ms: the look-up is not what the lanes wait for.)
    for (int q = 0; q < nkeys; ++q) {
        const bool first_half_q = q == 0;
        const uint64_t kq = (S_all >> (q == 0 ? 2 * (k - 1 - h) + 4 : 2 * (3 - q))) & hm, rq = rtk_revcomp(kq, h);
        const uint64_t c = kq < rq ? kq : rq;
        *n_lookups += 1;
        uint64_t i = rtk_hash64(c) & hx_mask, first = 0; bool found = false;
        while (true) { const uint64_t sv = hx[i]; *n_slots += 1; if (sv == RTK_EMPTY_KEY) break; if ((sv >> 34) == c) { first = sv & 0x3FFFFFFFFull; found = true; break; } i = (i + 1) & hx_mask; }
        if (!found) continue;
        const uint32_t cnt = static_cast<uint32_t>(hxl[first]); ++first;
        const bool palin = kq == rq; // (even h only: the h-mer reads the same on both strands, every place is a place of both orientations)
        for (uint32_t e = 0; e < cnt; ++e) {
            // a place is two words: the h + 1 bases behind / in front of the h-mer, then reversed << 63 | unitig << 32 | following-bases-exist << 31 | offset (bit 31 = `a_ok` of the two builders: the h + 1 bases that FOLLOW the h-mer in the forward unitig sequence exist): the candidate k-mer is the
            // h-mer with one of its flanks, the entry alone verifies it (round 5: the unitig's bounds and its sequence were two more dependent cache lines each)
            const uint64_t fl = hxl[first + 2ull * e], ent = hxl[first + 2ull * e + 1ull];
            *n_slots += 2; // a candidate costs its 16-byte list entry
            const uint64_t x = (ent >> 63) ? (c == kq ? rq : kq) : c; // the h-mer as the unitig spells it
            const uint32_t u = static_cast<uint32_t>(ent >> 32) & 0x7FFFFFFFu, pos = static_cast<uint32_t>(ent & 0x7FFFFFFFull);
            for (int ori = (x == kq ? 0 : 1), last = (palin ? 1 : ori); ori <= last; ++ori) { // ori 1: the unitig spells the reverse complement of the read's h-mer
                // the unitig h-mer is the first half of the forward k-mer F starting there, or the last half of the one starting h+1 earlier;
                // read-oriented G = F when the read h-mer itself was found, its reverse complement when the reverse-complemented key was
                const bool at_start = (first_half_q != (ori != 0));
                int64_t t; uint64_t F;
                if (at_start) { if (!((ent >> 31) & 1ull)) continue; t = pos; F = (x << (2 * (h + 1))) | (fl >> 32); }
                else { if (pos < static_cast<uint32_t>(h + 1)) continue; t = static_cast<int64_t>(pos) - (h + 1); F = ((fl & 0xFFFFFFFFull) << (2 * h)) | x; }
                const uint64_t G = ori ? rtk_revcomp(F, k) : F;
                const uint32_t kinds = rtk_one_edit(G, k, w_k1, ck, ck1, all_kinds);
                if (kinds) visit(G, rtk_pack_hit(u, static_cast<uint32_t>(t), ori ? 0u : 1u), kinds);
            }
        }
    }
}

#ifndef RTK_SEED_REGS
#define RTK_SEED_REGS 4 // distinct hits of a window kept in registers; windows with more take the counted slow path
#endif
RTK_DEV void rtk_inexact_tile_seeded(const GraphView& g, const BatchView& bv, uint64_t tile, unsigned long long* acc_probes, unsigned long long* acc_slots, unsigned long long* acc_hits, PoolChunk* chunk, uint32_t exclusive) {
    const int k = g.k;
    const uint64_t* const roff = bv.roff; const uint32_t n_reads = bv.n_reads;
    uint32_t lo_tile = 0; // one scalar search per tile: largest r with roff[r] <= first base of the tile
    lo_tile = rtk_owner_read(roff, n_reads, tile * 64);
#ifdef RTK_SIM
    const int n_sub = 64;
#else
    const int n_sub = 1;
#endif
    for (int sub = 0; sub < n_sub; ++sub) { // the 1-lane simulator visits the 64 positions of the tile one after the other
#ifdef RTK_SIM
        const uint64_t bb = tile * 64 + static_cast<uint64_t>(sub);
#else
        const uint64_t bb = tile * 64 + static_cast<uint64_t>(rtk_lane());
#endif
        bool cand = false; uint64_t c_k1 = 0; uint32_t ck = 4, ck1 = 4;
        if (bb < bv.n_bases) {
            uint32_t lo = lo_tile;
            while (lo + 1 < n_reads && roff[lo + 1] <= bb) ++lo;
            const uint64_t rend = roff[lo + 1];
            if (bb + static_cast<uint64_t>(k) <= rend && rend - roff[lo] > static_cast<uint64_t>(k)) {
                // k + 1 characters packed without per-character branches (the masked copy is padded by 64 bytes)
                int n_ok; const uint64_t S = rtk_pack_acgt(reinterpret_cast<const unsigned char*>(bv.masked.get()) + bb, k + 1, &n_ok);
                if (n_ok >= k - 1) {
                    cand = true; c_k1 = S >> 4;
                    if (n_ok >= k) { ck = static_cast<uint32_t>((S >> 2) & 3ull); if (n_ok >= k + 1 && bb + static_cast<uint64_t>(k) + 1 <= rend) ck1 = static_cast<uint32_t>(S & 3ull); }
                }
            }
        }
        // pass 1: up to four distinct hits of the lane's window are staged. Round 6: in LDS (64 lanes x 4 hits x 16 bytes = 4 KB per wave, word (2 i + w) of lane l at
        // [(2 i + w) * 64 + l]), their kinds of edit packed into one register. As arrays indexed by my_n they lived on the lane's stack; as 20 registers indexed
        // by constants they pushed other values out of the kernel's 80 (six waves per SIMD): a hit is a rare event (3 % of the windows), its staging can be slow.
#ifdef RTK_SIM
        uint64_t st_[2 * RTK_SEED_REGS];
#define RTK_ST(i, w) st_[2 * (i) + (w)]
#else
        __shared__ uint64_t rtk_inexact_stage[2 * RTK_SEED_REGS * RTK_WAVE];
        uint64_t* const st_ = rtk_inexact_stage + rtk_lane();
#define RTK_ST(i, w) st_[(2 * (i) + (w)) * RTK_WAVE]
#endif
        uint32_t kindpack = 0; int my_n = 0; bool more = false; uint32_t lookups = 0, slots = 0; // kinds of hit i: bits [4 i, 4 i + 3)
        // [A2] exclusive: the kinds of edit are searched one after the other and the first kind with a hit is the window's only one (rtk_a2_keep). ONE
        // visit of the window: every hit is kept with the kinds of edit that reach it, the kinds seen anywhere in the window decide which hits stay
        uint32_t keep = 7u, any = 0;
        if (cand) rtk_seeded_window(g, k, c_k1, ck, ck1, &lookups, &slots, exclusive != 0u, [&](uint64_t code, uint64_t hit, uint32_t kinds) {
            any |= kinds;
            for (int i = 0; i < my_n; ++i) if (RTK_ST(i, 1) == hit) { kindpack |= kinds << (4 * i); return; }
            if (my_n < RTK_SEED_REGS) { RTK_ST(my_n, 0) = code; RTK_ST(my_n, 1) = hit; kindpack |= kinds << (4 * my_n); ++my_n; } else more = true;
        });
        uint32_t live = (1u << my_n) - 1u; // the staged hits of a kept kind, in discovery order
        if (exclusive && cand) {
            keep = rtk_a2_keep(any, exclusive);
            if (!more) {
                for (int i = 0; i < RTK_SEED_REGS; ++i) if (!((kindpack >> (4 * i)) & keep)) live &= ~(1u << i);
                my_n = rtk_popc(static_cast<uint64_t>(live));
            }
        }
        *acc_probes += lookups; *acc_slots += slots;
        if (more) { // a window inside a repeat: count every visit, take a private slice of the pool, write them all. (`more` is raised by hits of ANY kind of edit -- the kinds
            // kept are only known once the whole window has been visited -- but the slice is sized from the KEPT hits alone: the pool does not grow with the exclusive reading of [A2])
            uint32_t n_all = 0, l2 = 0, s2 = 0;
            rtk_seeded_window(g, k, c_k1, ck, ck1, &l2, &s2, exclusive, [&](uint64_t, uint64_t, uint32_t kinds) { if (kinds & keep) ++n_all; });
            const unsigned long long pb = rtk_atomic_add(bv.ipool_top, static_cast<unsigned long long>(n_all));
            if (pb + n_all <= bv.ipool_cap && n_all < (1u << 24)) {
                uint32_t w = 0;
                rtk_seeded_window(g, k, c_k1, ck, ck1, &l2, &s2, exclusive, [&](uint64_t code, uint64_t hit, uint32_t kinds) { if (!(kinds & keep)) return; bv.ipool[2 * (pb + w)] = code; bv.ipool[2 * (pb + w) + 1] = hit; ++w; });
                bv.wdesc[bb] = (static_cast<uint64_t>(pb) << 24) | static_cast<uint64_t>(n_all);
                rtk_atomic_add(bv.counters + RTK_CNT_HITS_INEXACT, static_cast<unsigned long long>(n_all));
            } else rtk_atomic_add(bv.counters + RTK_CNT_OVERFLOW, 1ull);
            my_n = 0; live = 0;
        }
        int total; const int off = rtk_wave_excl_scan(my_n, &total);
        if (total > 0) {
            if (chunk->left < static_cast<uint32_t>(total)) { // refill the wave's private slice (the tail of the old slice is abandoned)
                unsigned long long nb = 0;
                if (rtk_lane() == 0) nb = rtk_atomic_add(bv.ipool_top, static_cast<unsigned long long>(RTK_POOL_CHUNK));
                chunk->base = rtk_shfl(nb, 0); chunk->left = RTK_POOL_CHUNK;
            }
            const unsigned long long pbase = chunk->base;
            chunk->base += static_cast<unsigned long long>(total); chunk->left -= static_cast<uint32_t>(total);
            if (pbase + static_cast<unsigned long long>(total) <= bv.ipool_cap) {
                int w_ = 0;
                for (int i = 0; i < RTK_SEED_REGS; ++i) if ((live >> i) & 1u) { bv.ipool[2 * (pbase + off + w_)] = RTK_ST(i, 0); bv.ipool[2 * (pbase + off + w_) + 1] = RTK_ST(i, 1); ++w_; }
                if (my_n) bv.wdesc[bb] = (static_cast<uint64_t>(pbase + off) << 24) | static_cast<uint64_t>(my_n);
            } else if (rtk_lane() == 0) rtk_atomic_add(bv.counters + RTK_CNT_OVERFLOW, 1ull);
            if (rtk_lane() == 0) *acc_hits += static_cast<unsigned long long>(total);
        }
    }
}

// ---------------------------------------------------------------------------------------------- finalize (src/Graph.cpp:201-372)
RTK_DEV bool rtk_char_eq_base(char c, uint32_t b) { return rtk_cls(static_cast<unsigned char>(c)) == static_cast<int>(b); }

// classification of one weak hit (src/Alignment.cpp:1049-1072): returns key or 0 when the hit is dropped
RTK_FN uint64_t rtk_weak_key(const char* ref, uint32_t pos, uint64_t code, int k) {
    // the k read characters as 2-bit codes next to the graph k-mer `code`; a character that is not A/C/G/T equals nothing (inv)
    int n_ok; uint64_t inv;
    const uint64_t R = rtk_pack_acgt(reinterpret_cast<const unsigned char*>(ref) + pos, k, &n_ok, &inv);
    const uint64_t even = 0x5555555555555555ull;
    const uint64_t mk = (k < 32) ? ((1ull << (2 * k)) - 1ull) : ~0ull, mk1 = (1ull << (2 * (k - 1))) - 1ull;
    auto neq = [&](uint64_t x) -> uint64_t { return (x | (x >> 1)) & even; }; // one bit (the low one of its pair) per differing character
    // character i sits at bits [2(k-1-i), 2(k-1-i)+1]
    const uint64_t d0 = (neq(code ^ R) | inv) & mk;                 // read[i] != kmer[i]
    const int l = d0 ? (__builtin_clzll(d0) - (64 - 2 * k)) / 2 : k; // first differing position
    if (l >= k) return 0;
    const uint64_t below = (l + 1 < k) ? ((1ull << (2 * (k - 1 - l))) - 1ull) : 0ull; // characters after position l (k-character layout)
    int type_var = 0; uint32_t mis = 0;
    const uint32_t ql = static_cast<uint32_t>((code >> (2 * (k - 1 - l))) & 3ull);
    if ((d0 & below) == 0) { type_var = 1; mis = 1u << ql; }        // read[i] == kmer[i] for every i > l
    if (!type_var) { // read[l + i] == kmer[l + 1 + i]: the read without the k-mer's character l; compare kmer[1..k) with read[0..k-1) from position l on
        const uint64_t d = (neq((code & mk1) ^ (R >> 2)) | (inv >> 2)) & mk1; // (k-1)-character layout: index j <-> read[j] vs kmer[j+1], bits 2(k-2-j)
        const uint64_t from_l = (l < k - 1) ? ((1ull << (2 * (k - 1 - l))) - 1ull) : 0ull; // indices j >= l
        if ((d & from_l) == 0) { type_var = 2; mis = 1u << ql; }
    }
    if (!type_var) { // read[l + 1 + i] == kmer[l + i]: compare kmer[0..k-1) with read[1..k) from position l on
        const uint64_t d = (neq((code >> 2) ^ (R & mk1)) | (inv & mk1)) & mk1; // index j <-> kmer[j] vs read[j+1], bits 2(k-2-j)
        const uint64_t from_l = (l < k - 1) ? ((1ull << (2 * (k - 1 - l))) - 1ull) : 0ull;
        if ((d & from_l) == 0) type_var = 3;
    }
    if (type_var == 0 || l == 0 || l == k - 1) return 0;
    return (static_cast<uint64_t>(pos + static_cast<uint32_t>(l)) << 16) | (static_cast<uint64_t>(mis) << 8) | static_cast<uint64_t>(type_var);
}

RTK_FN void rtk_finalize_read(const GraphView& g, const OptsView& o, const BatchView& bv, const SeedScratch& sc, uint32_t r) {
    RTK_ASSUME_LDS(&sc);
    const uint64_t base = bv.roff[r];
    const uint32_t L = static_cast<uint32_t>(bv.roff[r + 1] - base);
    const uint32_t k = static_cast<uint32_t>(g.k);
    if (rtk_lane() == 0) { bv.n_solid[r] = 0; bv.w_cnt[r] = 0; bv.w_off[r] = 0; }
    if (L <= k) { rtk_sync(); return; }
    const uint32_t nwin = L - k + 1;
    unsigned long long tph[8]; int iph = 0; const unsigned long long tstart = rtk_clock(); unsigned long long tlast = tstart;
#define RTK_PHASE() { const unsigned long long tn_ = rtk_clock(); tph[iph++] = tn_ - tlast; tlast = tn_; }
    const uint64_t* hits = bv.hits + base;
    uint32_t* s_pos = bv.s_pos + base;
    // ---- solid = exact hits minus runs that overlap the next run by less than k (src/Graph.cpp:221-239) ----
    // hit x (run end e, next hit nx after the gap) is dropped iff nx < x + k.
    // In bits: with b = presence of the k-1 windows after x (bit 0 = x+1), x is dropped iff b has a one above its first zero,
    // i.e. iff b & (b + 1) != 0. One ballot per 64 windows gives the presence bits; each lane looks at its own and the next block's.
    uint32_t n1 = 0;
    {
        const uint64_t* hmap = bv.hitmap;
        const uint32_t lane = static_cast<uint32_t>(rtk_lane());
        const uint64_t kmask = (1ull << (k - 1)) - 1ull;
        for (uint32_t cc = 0; cc < nwin; cc += 64 * RTK_WAVE) { // one 64-window block (and its successor) per lane, fetched together
            const uint32_t my_c0 = cc + 64u * lane;
            const uint64_t my_cur = (my_c0 < nwin) ? rtk_hit_bits(hmap, base + my_c0, nwin - my_c0) : 0ull;
            const uint64_t my_nxt = (my_c0 + 64 < nwin) ? rtk_hit_bits(hmap, base + my_c0 + 64, nwin - my_c0 - 64) : 0ull;
#ifndef RTK_SIM
            { // Every lane judges the 64 windows of ITS block with 128-bit shifts: window x is dropped iff a run of hits STARTS at x + 2 .. x + k - 1 (a one above a
              // zero in the presence bits behind x), i.e. iff the run starts T, OR-ed over a sliding window of k - 2 positions (doubling), have a bit at x + 2.
              // The sixty-four blocks of a step were visited one after the other, one window per lane (two 64-bit shuffles, a ballot and a store per block).
                const uint64_t Tl = my_cur & ~(my_cur << 1), Th = my_nxt & ~((my_nxt << 1) | (my_cur >> 63)); // (a start at bit 0 or 1 of the block is never asked for)
                uint64_t Rl = Tl, Rh = Th; uint32_t have = 1; const uint32_t w = k - 2;
                auto or_shifted = [&](uint32_t sft) { const uint64_t sl = (Rl >> sft) | (Rh << (64u - sft)), sh = Rh >> sft; Rl |= sl; Rh |= sh; }; // 0 < sft < 64
                while (2u * have <= w) { or_shifted(have); have *= 2u; }
                if (have < w) or_shifted(w - have);
                const uint64_t keep = my_cur & ~((Rl >> 2) | (Rh << 62));
                int total = 0; const uint32_t off = static_cast<uint32_t>(rtk_wave_excl_scan(rtk_popc(keep), &total));
                uint32_t o = n1 + off;
                for (uint64_t m = keep; m; m &= m - 1ull) s_pos[o++] = my_c0 + static_cast<uint32_t>(__builtin_ctzll(m));
                n1 += static_cast<uint32_t>(rtk_u(total));
            }
#else
            for (int bl = 0; bl < RTK_WAVE; ++bl) {
                const uint32_t c0 = cc + 64u * static_cast<uint32_t>(bl);
                if (c0 >= nwin) break;
                const uint64_t cur = rtk_u(rtk_shfl(my_cur, bl)), nxt = rtk_u(rtk_shfl(my_nxt, bl));
                if (cur == 0) continue;
#ifdef RTK_SIM
                for (uint32_t j = 0; j < 64; ++j) { // the 1-lane simulator walks the 64 windows of the block
                    if (!((cur >> j) & 1ull)) continue;
                    const uint64_t after = (j == 63) ? nxt : ((cur >> (j + 1)) | (nxt << (63 - j)));
                    const uint64_t b = after & kmask;
                    if ((b & (b + 1ull)) == 0) s_pos[n1++] = c0 + j;
                }
#else
                // bits x+1 .. x+k-1 of the 128-bit presence string (k - 1 <= 62)
                const uint64_t after = (lane == 63) ? nxt : ((cur >> (lane + 1)) | (nxt << (63 - lane)));
                const uint64_t b = after & kmask;
                const bool keep = ((cur >> lane) & 1ull) && ((b & (b + 1ull)) == 0);
                const uint64_t bal = rtk_ballot(keep);
                if (keep) s_pos[n1 + static_cast<uint32_t>(rtk_popc(bal & ((1ull << lane) - 1ull)))] = c0 + lane;
                n1 += static_cast<uint32_t>(rtk_popc(bal));
#endif
            }
#endif
        }
    }
    rtk_sync();
    RTK_PHASE();
    // ---- adjacent solid anchors on different unitigs must be graph neighbours sharing >= min_cov colours (:329-372) ----
    for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < n1; i += RTK_WAVE) sc.sflag[i] = 0; // 1 = emptied
    rtk_sync();
    // A junction = two solid anchors at adjacent positions on different unitigs. The verdict on one only depends on its two unitigs, the
    // emptiness flags it leaves behind carry on to later junctions. This kernel lasts as long as the longest read of the ticket (tens of
    // thousands of solid anchors, hundreds of junctions, one every other 64-anchor chunk), so nothing here may cost a memory round trip per
    // chunk or per junction: (1) the junctions are gathered into a list, four chunks of anchors in flight per step; (2) they are judged 64 at
    // a time, one per lane (adjacency words and colour sets of 64 pairs read side by side); (3) only the invalid ones -- a handful -- are then
    // visited in order to spread their flags (a valid junction changes nothing, whatever the flags around it say).
    {
        uint64_t* const jl = sc.vkey; const uint32_t jcap = 2u * sc.v_cap + 512u; // junction list: index of the right anchor, bit 63 = invalid
        uint32_t nj = 0;
        auto flush = [&]() {
            rtk_sync();
            for (uint32_t j0 = 0; j0 < nj; j0 += RTK_WAVE) { // (2)
                const uint32_t j = j0 + static_cast<uint32_t>(rtk_lane());
                if (j < nj) {
                    const uint32_t ii = static_cast<uint32_t>(jl[j]);
                    const UMap ul = rtk_unpack_hit(hits[s_pos[ii - 1]]), ur = rtk_unpack_hit(hits[s_pos[ii]]);
                    // right unitig must be the successor of the left one in walk direction (tail/head k-1 overlap) ...
                    bool inv = true;
                    const uint32_t* a = g.adj + 8ull * ul.unitig + (ul.strand ? 0 : 4);
                    for (int bb = 0; bb < 4; ++bb) if (a[bb] != RTK_NONE32 && (a[bb] >> 1) == ur.unitig && (a[bb] & 1u) == ur.strand) inv = false;
                    // ... and share enough colours
                    if (!inv) inv = rtk_lane_shared_unitigs(g, ul.unitig, ur.unitig, o.min_cov_vertices) < o.min_cov_vertices;
                    if (inv) jl[j] |= 1ull << 63;
                }
            }
            rtk_sync();
            for (uint32_t j0 = 0; j0 < nj; j0 += RTK_WAVE) { // (3)
                const uint32_t j = j0 + static_cast<uint32_t>(rtk_lane());
                const uint64_t e = j < nj ? jl[j] : 0ull;
                uint64_t bad = rtk_ballot((e >> 63) != 0);
                while (bad) {
                    const int l = rtk_ffs(bad) - 1; bad &= bad - 1ull;
                    const uint32_t ii = static_cast<uint32_t>(rtk_u(rtk_shfl(e, l)));
                    if (sc.sflag[ii - 1] || sc.sflag[ii]) continue;
                    const UMap ul = rtk_unpack_hit(hits[s_pos[ii - 1]]), ur = rtk_unpack_hit(hits[s_pos[ii]]);
                    uint32_t i_l = ii - 1, i_r = ii + 1;
                    i_l -= (i_l != 0) ? 1u : 0u;
                    while (i_l > 0 && s_pos[i_l] == s_pos[i_l + 1] - 1 && rtk_hit_unitig(hits[s_pos[i_l]]) == ul.unitig) { sc.sflag[i_l] = 1; --i_l; }
                    while (i_r < n1 && s_pos[i_r] == s_pos[i_r - 1] + 1 && rtk_hit_unitig(hits[s_pos[i_r]]) == ur.unitig) { sc.sflag[i_r] = 1; ++i_r; }
                    sc.sflag[ii - 1] = 1; sc.sflag[ii] = 1;
                    rtk_sync();
                }
            }
            nj = 0;
        };
        const uint32_t lane = static_cast<uint32_t>(rtk_lane());
        for (uint32_t c0 = 1; c0 < n1; c0 += 4 * RTK_WAVE) { // (1)
            if (nj + 4u * RTK_WAVE > jcap) flush();
            uint32_t p[4], q[4]; uint64_t hp[4], hq[4]; bool cand[4];
            for (int x = 0; x < 4; ++x) { const uint32_t i = c0 + static_cast<uint32_t>(x) * RTK_WAVE + lane; p[x] = q[x] = 0; if (i < n1) { p[x] = s_pos[i]; q[x] = s_pos[i - 1]; } }
            for (int x = 0; x < 4; ++x) { const uint32_t i = c0 + static_cast<uint32_t>(x) * RTK_WAVE + lane; cand[x] = i < n1 && p[x] - q[x] == 1; hp[x] = hq[x] = 0; if (cand[x]) { hp[x] = hits[p[x]]; hq[x] = hits[q[x]]; } }
            for (int x = 0; x < 4; ++x) {
                const bool c = cand[x] && rtk_hit_unitig(hp[x]) != rtk_hit_unitig(hq[x]);
                const uint64_t bal = rtk_ballot(c);
                if (c) jl[nj + static_cast<uint32_t>(rtk_popc(bal & ((1ull << lane) - 1ull)))] = c0 + static_cast<uint32_t>(x) * RTK_WAVE + lane;
                nj += static_cast<uint32_t>(rtk_popc(bal));
            }
        }
        flush();
    }
    rtk_sync();
    uint32_t n2 = 0;
    for (uint32_t c0 = 0; c0 < n1; c0 += RTK_WAVE) { // in-place compaction (write index never passes read index)
        const uint32_t i = c0 + static_cast<uint32_t>(rtk_lane());
        const bool keep = i < n1 && !sc.sflag[i];
        const uint32_t v = keep ? s_pos[i] : 0;
        const uint64_t bal = rtk_ballot(keep);
        rtk_sync();
        if (keep) s_pos[n2 + static_cast<uint32_t>(rtk_popc(bal & ((1ull << rtk_lane()) - 1ull)))] = v;
        n2 += static_cast<uint32_t>(rtk_popc(bal));
    }
    rtk_sync();
    if (rtk_lane() == 0) bv.n_solid[r] = n2;
    RTK_PHASE();
    // ---- weak = raw inexact hits sorted by (pos, mapped k-mer), deduplicated (src/Graph.cpp:201-216) ----
    // 64 windows per step. Windows holding several raw hits are first sorted by k-mer and stripped of duplicates in place (one
    // window at a time, the whole wave on it); then every lane appends the hits of its own window at its prefix-sum offset.
    uint32_t nv = 0; bool ovf = false;
    // This loop runs on the ONE wave that owns the read, 1 500 chunks for a 100 kb read, and the launch lasts as long as its slowest read: nothing in it
    // may wait for memory once per chunk. (1) The descriptors of RTK_GQ chunks are fetched together, those of the next step BEFORE this step's chunks are
    // worked on; a chunk without raw hits -- most of them -- costs nothing more. (2) The hits of the next window with several raw hits are fetched before
    // the current one is ranked and written back. (3) A lane appends the hits of its window four at a time (four 16-byte loads in flight, then the stores).
    // The pointers are read from the launch's views once: behind every store the compiler would fetch them again.
#ifdef RTK_SIM
#define RTK_GQ 2
#define RTK_GH 1
#else
#ifndef RTK_GQ_N
#define RTK_GQ_N 4
#endif
#ifndef RTK_GH_N
#define RTK_GH_N 4
#endif
#define RTK_GQ RTK_GQ_N
#define RTK_GH RTK_GH_N
static_assert(RTK_GQ % RTK_GH == 0, "a step is a whole number of halves");
#endif
    {
      const uint64_t* const wdesc = bv.wdesc.get() + base; uint64_t* const ipool = bv.ipool.get();
      uint32_t* const vpos = sc.vpos.get(); uint64_t* const vcode = sc.vcode.get(); uint64_t* const vhit = sc.vhit.get(); const uint32_t v_cap = sc.v_cap;
      const uint32_t lane_u = static_cast<uint32_t>(rtk_lane());
      uint64_t dn[RTK_GQ];
      for (int q = 0; q < RTK_GQ; ++q) { const uint32_t xx = static_cast<uint32_t>(q) * RTK_WAVE + lane_u; dn[q] = (xx < nwin) ? wdesc[xx] : 0ull; }
      for (uint32_t c4 = 0; c4 < nwin && !ovf; c4 += RTK_GQ * RTK_WAVE) {
        uint64_t d4[RTK_GQ];
        for (int q = 0; q < RTK_GQ; ++q) d4[q] = dn[q];
        { const uint32_t cn = c4 + RTK_GQ * RTK_WAVE; // (1)
          if (cn < nwin) for (int q = 0; q < RTK_GQ; ++q) { const uint32_t xx = cn + static_cast<uint32_t>(q) * RTK_WAVE + lane_u; dn[q] = (xx < nwin) ? wdesc[xx] : 0ull; } }
        // (a) the windows with several raw hits of all chunks of the step: duplicates out, in place
        uint32_t ucnt[RTK_GQ];
#pragma unroll
        for (int x4 = 0; x4 < RTK_GQ; ++x4) ucnt[x4] = (d4[x4] & 0xFFFFFFull) ? 1u : 0u;
#pragma unroll
        for (int x4 = 0; x4 < RTK_GQ; ++x4) {
          if (ovf) break;
          const uint64_t d = d4[x4];
          const uint64_t off = d >> 24; const uint32_t cnt = static_cast<uint32_t>(d & 0xFFFFFFull);
          uint64_t multi = rtk_ballot(cnt >= 2);
          if (!multi) continue;
#ifndef RTK_SIM
          // (2) the group of the next window with several hits, one element per lane when it fits (w_cnt <= 64)
          uint64_t nkey = ~0ull, nval = ~0ull;
          auto fetch_group = [&](uint64_t m) { nkey = ~0ull; nval = ~0ull; if (!m) return; const int s2 = rtk_ffs(m) - 1; const uint64_t o2 = rtk_u(rtk_shfl(off, s2)); const uint32_t n2_ = rtk_u(rtk_shfl(cnt, s2));
                                               if (n2_ <= 64 && lane_u < n2_) { nkey = ipool[2 * (o2 + lane_u)]; nval = ipool[2 * (o2 + lane_u) + 1]; } };
          fetch_group(multi);
#endif
          while (multi) {
            const int sl = rtk_ffs(multi) - 1;
            multi &= multi - 1ull;
            const uint64_t w_off = rtk_u(rtk_shfl(off, sl)); const uint32_t w_cnt = rtk_u(rtk_shfl(cnt, sl));
#ifndef RTK_SIM
            const uint64_t key = nkey, val = nval;
            fetch_group(multi);
            if (w_cnt <= 64) { // the usual case: the group fits one element per lane -> rank by all-pairs comparison in registers, no scratch, no barrier
                const uint32_t li = lane_u;
                bool dup = false;
                for (uint32_t j = 0; j < w_cnt; ++j) { // is an equal k-mer ordered before mine? ((k-mer, hit, index) ascending, like the sort + first-of-run rule)
                    const uint64_t kj = rtk_shfl(key, static_cast<int>(j)), vj = rtk_shfl(val, static_cast<int>(j));
                    dup = dup || (kj == key && (vj < val || (vj == val && j < li)));
                }
                const bool uq = li < w_cnt && !dup;
                uint64_t ub = rtk_ballot(uq);
                const uint32_t nu = static_cast<uint32_t>(rtk_popc(ub));
                uint32_t pos = 0;
                while (ub) { const int j = rtk_ffs(ub) - 1; ub &= ub - 1ull; pos += (rtk_shfl(key, j) < key) ? 1u : 0u; }
                if (uq) { ipool[2 * (w_off + pos)] = key; ipool[2 * (w_off + pos) + 1] = val; }
                if (static_cast<int>(li) == sl) ucnt[x4] = nu;
                continue;
            }
#endif
            if (2ull * w_cnt > 2ull * sc.v_cap + 512ull) { ovf = true; break; } // vkey / vidx hold 2 * v_cap + 512 entries, the sort pads to a power of two
            for (uint32_t i = lane_u; i < w_cnt; i += RTK_WAVE) { sc.vkey[i] = ipool[2 * (w_off + i)]; sc.vidx[i] = ipool[2 * (w_off + i) + 1]; }
            rtk_sync();
            rtk_sort_pairs(sc.vkey, sc.vidx, w_cnt);
            uint32_t nu = 0; // first entry of every run of equal k-mers goes back to the front of the group
            for (uint32_t i0 = 0; i0 < w_cnt; i0 += RTK_WAVE) {
                const uint32_t i = i0 + lane_u;
                const bool uq = i < w_cnt && (i == 0 || sc.vkey[i] != sc.vkey[i - 1]);
                const uint64_t ub = rtk_ballot(uq);
                if (uq) { const uint64_t dst = w_off + nu + static_cast<uint32_t>(rtk_popc(ub & ((1ull << rtk_lane()) - 1ull))); ipool[2 * dst] = sc.vkey[i]; ipool[2 * dst + 1] = sc.vidx[i]; }
                nu += static_cast<uint32_t>(rtk_popc(ub));
            }
            rtk_sync();
            if (rtk_lane() == sl) ucnt[x4] = nu;
          }
        }
        if (ovf) break;
        rtk_sync(); // compacted groups are read back by their window's lane
        // (b) where every window's hits go: chunk after chunk, window after window
        uint32_t my_o[RTK_GQ]; uint32_t nv_step = nv; bool any = false;
#pragma unroll
        for (int x4 = 0; x4 < RTK_GQ; ++x4) { int tot = 0; const uint32_t mo = static_cast<uint32_t>(rtk_wave_excl_scan(static_cast<int>(ucnt[x4]), &tot)); my_o[x4] = nv_step + mo; nv_step += static_cast<uint32_t>(rtk_u(tot)); any = any || tot != 0; }
        if (!any) continue;
        if (nv_step > v_cap) { ovf = true; break; }
        // (c) the first hit of every window of the step: the loads of all chunks in flight, then the stores; (3) the further hits of a window four at a time
#pragma unroll
        for (int h0 = 0; h0 < RTK_GQ; h0 += RTK_GH) { // (half a step at a time: registers)
          uint64_t cc_[RTK_GH], hh_[RTK_GH];
#pragma unroll
          for (int x4 = 0; x4 < RTK_GH; ++x4) if (ucnt[h0 + x4]) { const uint64_t off = d4[h0 + x4] >> 24; cc_[x4] = ipool[2 * off]; hh_[x4] = ipool[2 * off + 1]; }
#pragma unroll
          for (int x4 = 0; x4 < RTK_GH; ++x4) if (ucnt[h0 + x4]) { const uint32_t o = my_o[h0 + x4]; vpos[o] = c4 + static_cast<uint32_t>(h0 + x4) * RTK_WAVE + lane_u; vcode[o] = cc_[x4]; vhit[o] = hh_[x4]; }
        }
#pragma unroll
        for (int x4 = 0; x4 < RTK_GQ; ++x4) {
          if (rtk_ballot(ucnt[x4] > 1) == 0) continue;
          const uint64_t off = d4[x4] >> 24; const uint32_t x = c4 + static_cast<uint32_t>(x4) * RTK_WAVE + lane_u;
          for (uint32_t i = 1; i < ucnt[x4]; i += 4) {
              uint64_t cc_[4], hh_[4];
              for (uint32_t j = 0; j < 4; ++j) if (i + j < ucnt[x4]) { cc_[j] = ipool[2 * (off + i + j)]; hh_[j] = ipool[2 * (off + i + j) + 1]; }
              for (uint32_t j = 0; j < 4; ++j) if (i + j < ucnt[x4]) { const uint32_t o = my_o[x4] + i + j; vpos[o] = x; vcode[o] = cc_[j]; vhit[o] = hh_[j]; }
          }
        }
        nv = nv_step;
      }
    }
#undef RTK_GQ
#undef RTK_GH
    if (ovf) { *sc.overflow = 1; nv = 0; }
    rtk_sync();
    RTK_PHASE();
    // ---- keep_non_overlap (src/Alignment.cpp:1017-1199) ----
    // NB: an inexact hit is never "solid": its mapped k-mer differs from the read window by construction.
    uint32_t n_keep = 0;
    if (nv) {
        const char* ref = bv.seq + base;
        uint32_t nvalid = 0;
#ifndef RTK_SIM
        if (nv > RTK_LDS_SORT_CAP && L < (1u << 26)) {
            // thousands of hits (the read that the launch waits for): the classified ones are compacted in hit order as (32-bit key, hit index) -- variant position,
            // base mask and kind of the 64-bit key side by side -- and sorted by a stable radix sort (three passes for a 100 kb read; the bitonic network on
            // 8 192 pairs was a third of that read's time); the group arrays, not in use yet, are the buffers
            uint32_t* const Ka = sc.gstart.get(); uint32_t* const Ia = sc.gcnt.get(); uint32_t* const Kb = sc.gps.get(); uint32_t* const Ib = sc.gpe.get();
            const uint64_t lt = (1ull << rtk_lane()) - 1ull;
            for (uint32_t c0 = 0; c0 < nv; c0 += RTK_WAVE) {
                const uint32_t i = c0 + static_cast<uint32_t>(rtk_lane());
                uint64_t key = 0;
                if (i < nv) { key = rtk_weak_key(ref, sc.vpos[i], sc.vcode[i], static_cast<int>(k)); sc.vflag[i] = 0; }
                const bool ok = key != 0;
                const uint64_t bal = rtk_ballot(ok);
                if (ok) { const uint32_t j = nvalid + static_cast<uint32_t>(rtk_popc(bal & lt)); Ka[j] = static_cast<uint32_t>(((key >> 16) << 6) | (((key >> 8) & 15ull) << 2) | (key & 3ull)); Ia[j] = i; }
                nvalid += static_cast<uint32_t>(rtk_popc(bal));
            }
            rtk_sync();
            rtk_radix_sort_pairs_u32(Ka, Ia, Kb, Ib, nvalid, (L << 6) | 63u);
            for (uint32_t j = static_cast<uint32_t>(rtk_lane()); j < nvalid; j += RTK_WAVE) {
                const uint32_t K = Ka[j];
                sc.vkey[j] = (static_cast<uint64_t>(K >> 6) << 16) | (static_cast<uint64_t>((K >> 2) & 15u) << 8) | static_cast<uint64_t>(K & 3u); sc.vidx[j] = Ia[j];
            }
            rtk_sync();
        } else
#endif
        {
            for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < nv; i += RTK_WAVE) {
                const uint64_t key = rtk_weak_key(ref, sc.vpos[i], sc.vcode[i], static_cast<int>(k));
                sc.vkey[i] = key ? key : ~0ull; sc.vidx[i] = i; sc.vflag[i] = 0;
            }
            rtk_sync();
            rtk_sort_pairs(sc.vkey, sc.vidx, nv);
            { // number of classified hits = first index with an all-ones key
                uint32_t lo = 0, hi = nv; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sc.vkey[mid] == ~0ull) hi = mid; else lo = mid + 1; } nvalid = lo;
            }
        }
        RTK_PHASE();
        // groups of equal key
        uint32_t ng = 0;
        for (uint32_t i0 = 0; i0 < nvalid; i0 += RTK_WAVE) { // group starts, compacted in order
            const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
            const bool st = i < nvalid && (i == 0 || sc.vkey[i] != sc.vkey[i - 1]);
            const uint64_t sb = rtk_ballot(st);
            if (st) sc.gstart[ng + static_cast<uint32_t>(rtk_popc(sb & ((1ull << rtk_lane()) - 1ull)))] = i;
            ng += static_cast<uint32_t>(rtk_popc(sb));
        }
        rtk_sync();
        // (vcode is free once the hits are classified: it takes one word per group -- the variant's position and, for a group of one hit, the
        // unitig and strand of that hit -- so that the neighbour test below reads a group with independent loads instead of three dependent ones)
        uint64_t* const ginfo = sc.vcode.get();
        for (uint32_t gi = static_cast<uint32_t>(rtk_lane()); gi < ng; gi += RTK_WAVE) { // one group per lane: extent and position span
            const uint32_t a = sc.gstart[gi], e = (gi + 1 < ng) ? sc.gstart[gi + 1] : nvalid;
            uint32_t ps = 0xFFFFFFFFu, pe = 0;
            for (uint32_t j = a; j < e; ++j) { const uint32_t pp = sc.vpos[sc.vidx[j]]; ps = pp < ps ? pp : ps; pe = (pp + k) > pe ? (pp + k) : pe; }
            sc.gcnt[gi] = e - a; sc.gps[gi] = ps; sc.gpe[gi] = pe; sc.gkeep[gi] = 1;
            const uint64_t h0 = sc.vhit[sc.vidx[a]];
            ginfo[gi] = (((static_cast<uint64_t>(rtk_hit_unitig(h0)) << 1) | (h0 & 1ull)) << 32) | (sc.vkey[a] >> 16 & 0xFFFFFFFFull);
        }
        rtk_sync();
        RTK_PHASE();
        // a variant is dropped iff another variant overlaps it within k without sharing a unitig (order independent)
        for (uint32_t gi = static_cast<uint32_t>(rtk_lane()); gi < ng; gi += RTK_WAVE) {
            const uint64_t G1 = ginfo[gi]; const uint32_t p1 = static_cast<uint32_t>(G1 & 0xFFFFFFFFull);
            const uint32_t ps1 = sc.gps[gi], pe1 = sc.gpe[gi], c1 = sc.gcnt[gi];
            const uint32_t lower = (p1 < k - 1) ? 0u : (p1 - k + 1); const uint32_t upper = ((p1 + k) >= L) ? L : (p1 + k);
            bool conflict = false;
            for (int dir = 0; dir < 2 && !conflict; ++dir) {
                int64_t gj = dir ? static_cast<int64_t>(gi) + 1 : static_cast<int64_t>(gi) - 1;
                while (gj >= 0 && gj < static_cast<int64_t>(ng) && !conflict) {
                    const uint64_t G2 = ginfo[gj]; const uint32_t ps2 = sc.gps[gj], pe2 = sc.gpe[gj], c2 = sc.gcnt[gj]; // (one round trip)
                    const uint32_t p2 = static_cast<uint32_t>(G2 & 0xFFFFFFFFull);
                    if (p2 < lower || p2 > upper) break;
                    const bool ov1 = (p1 >= ps2) && (p1 < pe2);
                    const bool ov2 = (p2 >= ps1) && (p2 < pe1);
                    if (ov1 || ov2) {
                        bool same = false;
                        if (c1 == 1 && c2 == 1) same = (G1 >> 32) == (G2 >> 32);
                        else for (uint32_t a = 0; a < c1 && !same; ++a) {
                            const uint64_t ha = sc.vhit[sc.vidx[sc.gstart[gi] + a]];
                            for (uint32_t b2 = 0; b2 < c2 && !same; ++b2) {
                                const uint64_t hb = sc.vhit[sc.vidx[sc.gstart[gj] + b2]];
                                same = (rtk_hit_unitig(ha) == rtk_hit_unitig(hb)) && ((ha & 1ull) == (hb & 1ull));
                            }
                        }
                        if (!same) conflict = true;
                    }
                    gj += dir ? 1 : -1;
                }
            }
            if (conflict) sc.gkeep[gi] = 0;
        }
        rtk_sync();
        for (uint32_t gi = static_cast<uint32_t>(rtk_lane()); gi < ng; gi += RTK_WAVE) if (sc.gkeep[gi]) for (uint32_t a = 0; a < sc.gcnt[gi]; ++a) sc.vflag[sc.vidx[sc.gstart[gi] + a]] = 1;
        rtk_sync();
        RTK_PHASE();
        // count, reserve space in the weak pool, write in original order
        for (uint32_t c0 = 0; c0 < nv; c0 += RTK_WAVE) { const uint32_t i = c0 + static_cast<uint32_t>(rtk_lane()); n_keep += static_cast<uint32_t>(rtk_popc(rtk_ballot(i < nv && sc.vflag[i]))); }
        if (n_keep) {
            unsigned long long wb = 0;
            if (rtk_lane() == 0) wb = rtk_atomic_add(bv.wk_top, static_cast<unsigned long long>(n_keep));
            wb = rtk_shfl(wb, 0);
            if (wb + n_keep > bv.wk_cap) { *sc.overflow = 1; n_keep = 0; }
            else {
                uint32_t w = 0;
                for (uint32_t c0 = 0; c0 < nv; c0 += RTK_WAVE) {
                    const uint32_t i = c0 + static_cast<uint32_t>(rtk_lane());
                    const bool kp = i < nv && sc.vflag[i];
                    const uint64_t bal = rtk_ballot(kp);
                    if (kp) { const uint64_t dst = wb + w + static_cast<uint32_t>(rtk_popc(bal & ((1ull << rtk_lane()) - 1ull))); bv.wk_pos[dst] = sc.vpos[i]; bv.wk_hit[dst] = sc.vhit[i]; }
                    w += static_cast<uint32_t>(rtk_popc(bal));
                }
                if (rtk_lane() == 0) { bv.w_off[r] = wb; bv.w_cnt[r] = n_keep; }
            }
        }
    }
    RTK_PHASE();
    if (rtk_lane() == 0) { // phase profile of the stage (wave cycles), read back under RTK_TRACE
        for (int i = 0; i < iph && i < 7; ++i) rtk_atomic_add(bv.counters + 24 + i, tph[i]);
#ifndef RTK_SIM
        { const unsigned long long mine = rtk_clock() - tstart; const unsigned long long was = atomicMax(bv.counters.get() + 31, mine);
          if (mine > was) for (int i = 0; i < iph && i < 7; ++i) bv.counters[72 + i] = tph[i]; } // (developer trace: the phases of the slowest read so far; races between two record holders are harmless)
#endif
    }
#undef RTK_PHASE
    rtk_sync();
}

#endif
