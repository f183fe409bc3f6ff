// Region stage of the per-read correction on the device: one wavefront owns one weak region of one long read
// (reference: src/Correction.cpp:159-958 correctSequence and its `correct` lambda :431-753, chooseColors :215-429,
// extractSemiWeakPaths :3-157; src/GraphTraversal.cpp explorePathsBFS :3-210, explorePathsBFS2 :212-454,
// exploreSubGraph :456-587, getScorePath :722-772 and :867-909; src/Alignment.cpp selectBest*Alignment :3-147,:967-1015,
// generateConsensus :309-470; src/Path.hpp; src/ResultCorrection.hpp).
//
// The program is wave-uniform: every lane executes the same control flow on the same values; the lanes split up
// only inside the bulk primitives (2-bit decode of unitig substrings, bit-parallel Myers with one query word per
// lane, sorted-set algebra with __ballot compaction, copies). Paths live as immutable records in per-wave bump
// arenas in HBM (three nesting levels: region / BFS call / DFS call); the reference's queue, stack and candidate
// vectors become small handle lists. Canonical tie rules [D1] (see oracle/oracle_correct.hpp) are applied where the
// reference depends on heap addresses. Index annotations that our index producer never emits (short cycles, SNP
// ambiguities) are handled by rtk_fix_repeats and rtk_ambiguity.h.
#ifndef RTK_REGION_H
#define RTK_REGION_H

#include "rtk_myers.h"
#include "rtk_seeds.h"
#include "rtk_sets.h"
#include "rtk_types.h"
#include "rtk_wave.h"

// ------------------------------------------------------------------------------------------------ data
struct RegionDesc { // one entry per output segment of a read, in read order
    uint32_t read;
    uint32_t kind;     // RTK_RG_*
    uint32_t i_solid;  // index of the left solid anchor (interior / tail), unused otherwise
    uint32_t prev_pos; // where the previous segment stopped in the read
    uint64_t seg_off;  // out: offset of the segment in the segment pool (sequence bytes, then quality bytes)
    uint32_t seq_len, qual_len; // out
    uint32_t status;   // out: non-zero = scratch overflow, redo with a bigger arena
    uint32_t pad;
};
#define RTK_RG_WHOLE_MAX 0   // read returned unchanged, qualities all 'I' (every window solid)
#define RTK_RG_WHOLE_MIN 1   // read returned unchanged, qualities all '!' (no solid anchor / too short)
#define RTK_RG_HEAD 2        // before the first solid anchor (reverse-complement correction, src/Correction.cpp:776-797)
#define RTK_RG_GAP 3         // between two consecutive solid anchors that are not adjacent (:803-935)
#define RTK_RG_TAIL 4        // after the last solid anchor, corrected forward (:940-950)
#define RTK_RG_TAIL_COPY 5   // read ends on a solid anchor (:951-955)

struct RegionBatch {
    U<RegionDesc*> regions; U<uint64_t> regions_cap; U<unsigned long long*> n_regions;
    U<uint64_t*> r_first; U<uint32_t*> r_count;   // per read: its slice of `regions`
    U<char*> seq_rc;                           // reverse complement of every read (same offsets as seq)
    U<char*> qual_rev;                         // pass 2: every read's quality string reversed (q_bw of src/Correction.cpp:186,198)
    U<char*> seg_pool; U<uint64_t> seg_cap; U<unsigned long long*> seg_top;
    U<unsigned long long*> next_region;        // dequeue head of the persistent region kernel
    U<uint32_t*> eorder;                       // the regions that need no graph walk (k_regions_easy), in any order; their number is n_heavy[2]
    U<uint32_t*> rorder; U<unsigned long long*> n_heavy; // dequeue order of the region kernel: the heavy regions (long gaps, read heads / tails) from the front, the light ones from the back (k_region_order); [0] heavy, [1] light
    U<unsigned long long*> n_overflow;         // regions that ran out of scratch in the last launch
    U<uint32_t*> horder; // the regions the lane kernel handed on to the wave kernel; their number is n_heavy[4]
    U<uint32_t*> lorder; U<unsigned long long*> next_lane; U<uint32_t> lane_max_gap; // the regions of the lane-per-region kernel (k_regions_lanes): gaps under lane_max_gap bases, by size class; their number is n_heavy[3]; 0: no such class
    U<char*> out_pool; U<uint64_t> out_cap; U<unsigned long long*> out_top;
    U<uint64_t*> out_off; U<uint32_t*> out_seq_len; U<uint32_t*> out_qual_len; // per read
    U<uint64_t*> st_off;                       // per segment: quality bytes << 32 | characters of its read in front of it (k_stitch)
};

struct RegionScratchCfg { ScratchCfg my; uint32_t set_cap, um_cap, str_cap, list_cap, memo_cap, bm_words; uint64_t arena_cap; };

struct WPath { U<UMap*> ums; U<char*> qual; U<uint32_t> n, l, qlen; }; // mutable working path

// anchors of a read in one orientation, side lists of chooseColors, result of one `correct` call (functions further down)
struct Anchors { U<const uint32_t*> pos; U<const uint64_t*> hit; U<const uint64_t*> hits_by_pos; U<uint32_t> n, L; U<int> rev; U<int> k; };
struct SideList { uint32_t* u; uint8_t* nb; uint32_t n, cap; };
struct ResCorr { char* seq; char* qual; uint32_t seq_len, qual_len; uint64_t* bm; uint32_t old_len; bool is_corrected; uint32_t n_all; int all_set; };
// Locals of the region drivers that travel by reference (rtk_correct_region, rtk_generate_consensus, rtk_choose_colors): kept in the
// header (LDS in the kernels) instead of the wave's stack, where every wave-uniform word is a 256-byte row per store and per load
// state of one rtk_correct_region call that its three parts hand on (side lists + colours | path search | assembly + trim)
struct RegionCall { const char* s_read; const char* q_read; uint64_t complete; UMap um1, um2; uint32_t s_len, p1, p2, first_pos, len_weak_region, lw_lo, lw_hi, n_all, n_partial, n_amb, has_end_pt, found_first, lrc; };
struct DriverLocals { Anchors an[4]; ResCorr rc[2]; SideList side[3]; uint32_t len[6]; int best[2]; MyersSaved saved; RegionCall call; };

struct RegionScratch {
    MyersScratch my;
    U<uint32_t*> set[10]; U<uint32_t> set_cap;
    U<char*> arena[3]; U<uint64_t> arena_cap; U<uint64_t> top[3];   // 0 region level, 1 BFS level, 2 DFS level
    WPath wp[4]; U<uint32_t> um_cap;
    U<char*> str[5]; U<uint32_t> str_cap;
    U<char*> rbuf[8];                                       // result strings: fw seq/qual, bw seq/qual, out seq/qual, 2 temporaries
    U<uint64_t*> list[11]; U<uint32_t> list_cap;                // 6..10: SNP-annotation sets (rtk_ambiguity.h)
    U<uint32_t*> memo_u; U<uint8_t*> memo_v; U<uint32_t> memo_cap; U<uint32_t> memo_n;
    U<uint64_t*> bm[3]; U<uint32_t> bm_words;
    UL<uint32_t*> overflow; U<uint32_t> ovf_word; // the flag itself, next to the header (same memory: LDS in the kernels)
    DriverLocals loc;
    U<unsigned long long> cnt[16]; // expand, colour, pathbase, align, cells, then cycles: colour, paths, consensus, total, myers, sets
    U<unsigned long long> fine[16]; // developer cycle counters printed with RTK_TRACE (RTK_FINE names in rtk_pipeline_run.inc)
#ifdef RTK_PROF
    U<unsigned long long> prof[48]; U<unsigned long long> prof_t; // developer build (-DRTK_PROF): lap profile of the region program, every cycle of a wave attributed to one slot (RTK_PL)
#endif
#ifndef RTK_SLIM_HDR
    U<unsigned long long> hist[32]; // region time by size class: [b] cycles, [8 + b] regions, [16 + b] regions that needed the reverse strand too, [24 + b] DFS calls
#endif
};
#ifdef RTK_SLIM_HDR // A/B build: header of 1 KB (20 waves per CU fit next to a 7 KB set buffer); the size-class table is not kept
#define RTK_HIST_ADD(sc, i, v) ((void)0)
#define RTK_HIST_GET(sc, i) 0ull
#else
#define RTK_HIST_ADD(sc, i, v) ((sc).hist[i] += (v))
#define RTK_HIST_GET(sc, i) ((sc).hist[i])
#endif
#ifdef RTK_PROF
#define RTK_PL(sc, i) do { const unsigned long long t_ = rtk_clock(); (sc).prof[i] += t_ - (sc).prof_t; (sc).prof_t = t_; } while (0)
#else
#define RTK_PL(sc, i) ((void)0)
#endif

// The views of a launch, ONE copy in device memory per batch (written by k_set_ctx in front of the kernels that read it). The wave
// programs read them through RCtx: a per-wave copy on the wave's stack costs 64 lanes x the struct in scratch memory (the stack is
// interleaved per lane), 45 KB per wave that every `c.g.x` then fetches a 256-byte row of.
struct LaunchCtx { GraphView g; OptsView o; BatchView bv; RegionBatch rb; };

struct RCtx { // everything a region program needs
    const GraphView& g; const OptsView& o; const BatchView& bv; const RegionBatch& rb; // -> the LaunchCtx of the launch
    UL<RegionScratch*> sc;
    U<int> k;
};

// the header of the wave's work area: in LDS in every kernel that runs the region / read programs (k_regions, k_phase, k_phase_long)
RTK_DEV RegionScratch& rtk_hdr(const RCtx& c) { RegionScratch* p = c.sc; RTK_ASSUME_LDS(p); return *p; }

#ifndef RTK_SIM
RTK_DEV UMap rtk_u(const UMap& m) { UMap r; r.unitig = rtk_u(m.unitig); r.dist = rtk_u(m.dist); r.len = rtk_u(m.len); r.strand = rtk_u(m.strand); return r; }
#endif

// ------------------------------------------------------------------------------------------------ helpers (src/Common.hpp:410-438)
RTK_DEV char rtk_get_qual(double score, uint64_t qv_min, uint64_t qv_max) {
    const char phred_base_std = static_cast<char>(33);
    const char phred_scale_std = static_cast<char>(qv_max);
    const double s = score < 1.0 ? score : 1.0;
    const double qv_score = s * static_cast<double>(static_cast<uint64_t>(phred_scale_std) - qv_min);
    return static_cast<char>(qv_score + static_cast<double>(phred_base_std) + static_cast<double>(qv_min));
}
RTK_DEV void rtk_min_max_len(uint64_t l, double f, uint64_t* mn, uint64_t* mx) {
    const double lf = static_cast<double>(l);
    const double a = lf - (lf * f), b = lf + (lf * f);
    *mn = static_cast<uint64_t>(a > 1.0 ? a : 1.0); *mx = static_cast<uint64_t>(b > 1.0 ? b : 1.0);
}

RTK_DEV char rtk_comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
                 case 'V': return 'B'; case 'B': return 'V'; case 'H': return 'D'; case 'D': return 'H'; default: return c; }
}

// the flag is the header's own ovf_word (s.overflow points at it for the alignment code, which only knows its MyersScratch)
RTK_DEV void rtk_fail_ovf(RegionScratch& s, uint32_t code) { s.ovf_word = code; }
RTK_DEV bool rtk_failed(const RegionScratch& s) { return s.ovf_word != 0; }

// anchors of a read in forward or reverse-complement orientation (src/Correction.cpp:196-213)
RTK_DEV uint32_t rtk_an_pos(const Anchors& a, uint32_t i) { return a.rev ? (a.L - a.pos[a.n - 1 - i] - static_cast<uint32_t>(a.k)) : a.pos[i]; }
RTK_DEV UMap rtk_an_um(const Anchors& a, uint32_t i) {
    const uint32_t j = a.rev ? (a.n - 1 - i) : i;
    UMap u = rtk_unpack_hit(a.hit ? a.hit[j] : a.hits_by_pos[a.pos[j]]);
    if (a.rev) u.strand ^= 1u;
    return u;
}

// positions are ascending in the anchor index: searches replace the reference's linear walks over the lists. A search is a chain of
// dependent memory round trips, so it is 64-ary: every lane probes one pivot per step (two steps for 4096 anchors instead of twelve).
// first x in [lo, hi) with pos(x) >= key (strict: > key), else hi
RTK_DEV uint32_t rtk_an_search(const Anchors& a, uint32_t lo_, uint32_t hi_, uint64_t key_, bool strict_) {
    uint32_t lo = rtk_u(lo_), hi = rtk_u(hi_); const uint64_t key = rtk_u(key_); const bool strict = rtk_u(strict_);
    const uint32_t lane = static_cast<uint32_t>(rtk_lane());
#ifdef RTK_SIM
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; const uint64_t p = rtk_an_pos(a, mid); if (strict ? (p <= key) : (p < key)) lo = mid + 1; else hi = mid; }
    (void)lane; return lo;
#else
    while (lo < hi) {
        const uint32_t span = hi - lo;
        if (span <= RTK_WAVE) { // one probe per candidate
            const uint32_t x = lo + lane; bool t = false;
            if (x < hi) { const uint64_t p = rtk_an_pos(a, x); t = strict ? (p > key) : (p >= key); }
            const uint64_t b = rtk_ballot(t);
            return b ? lo + static_cast<uint32_t>(rtk_ffs(b) - 1) : hi;
        }
        // 64 pivots strictly inside [lo, hi): x_i = lo + (i + 1) * span / 65
        const uint32_t x = lo + static_cast<uint32_t>((static_cast<uint64_t>(lane + 1) * span) / (RTK_WAVE + 1));
        const uint64_t p = rtk_an_pos(a, x);
        const bool t = strict ? (p > key) : (p >= key);
        const uint64_t b = rtk_ballot(t); // monotone: 0..0 1..1
        const int j = b ? rtk_ffs(b) - 1 : RTK_WAVE; // first pivot that satisfies the test
        const uint32_t nlo = (j == 0) ? lo : rtk_u(rtk_shfl(x, j - 1)) + 1u; // the answer is after pivot j-1 ...
        const uint32_t nhi = (j == RTK_WAVE) ? hi : rtk_u(rtk_shfl(x, j));   // ... and not after pivot j
        lo = nlo; hi = nhi;
    }
    return lo;
#endif
}
RTK_DEV uint32_t rtk_an_first_ge(const Anchors& a, uint32_t lo, uint32_t hi, uint64_t key) { return rtk_an_search(a, lo, hi, key, false); }
RTK_DEV uint32_t rtk_an_first_gt(const Anchors& a, uint32_t lo, uint32_t hi, uint64_t key) { return rtk_an_search(a, lo, hi, key, true); }

// ------------------------------------------------------------------------------------------------ arenas and paths (src/Path.hpp)
struct PathHdr { U<uint32_t> n, l, qlen, pad; }; // followed by n UMap and qlen quality bytes

RTK_DEV uint64_t rtk_arena_alloc(RegionScratch& s, int lvl, uint64_t bytes) {
    bytes = (bytes + 15ull) & ~15ull;
    const uint64_t off = rtk_ld(&s.top[lvl]);
    if (off + bytes > rtk_ld(&s.arena_cap)) { rtk_fail_ovf(s, 3); return 0; }
    s.top[lvl] = off + bytes; return off;
}
RTK_DEV PathHdr* rtk_path_hdr(const RegionScratch& s, int lvl, uint64_t h) { return reinterpret_cast<PathHdr*>(rtk_ld(&s.arena[lvl]) + h); }
RTK_DEV UMap* rtk_path_ums(const RegionScratch& s, int lvl, uint64_t h) { return reinterpret_cast<UMap*>(rtk_ld(&s.arena[lvl]) + h + sizeof(PathHdr)); }
RTK_DEV char* rtk_path_qual(const RegionScratch& s, int lvl, uint64_t h) { const PathHdr* p = rtk_path_hdr(s, lvl, h); return reinterpret_cast<char*>(const_cast<PathHdr*>(p)) + sizeof(PathHdr) + sizeof(UMap) * rtk_ld(&p->n); }
// handles carry their level in the top 2 bits
RTK_DEV uint64_t rtk_mk_handle(int lvl, uint64_t off) { return (static_cast<uint64_t>(lvl) << 62) | off; }
RTK_DEV int rtk_h_lvl(uint64_t h) { return static_cast<int>(h >> 62); }
RTK_DEV uint64_t rtk_h_off(uint64_t h) { return h & 0x3FFFFFFFFFFFFFFFull; }

RTK_DEV void rtk_wp_clear(WPath& p) { p.n = 0; p.l = 0; p.qlen = 0; }

RTK_FN_LEAF uint64_t rtk_wp_commit(RegionScratch& s_, const WPath& p_, int lvl_) { // working path -> immutable record
    RegionScratch& s = *rtk_u(&s_); RTK_ASSUME_LDS(&s); const WPath& p = *rtk_u(&p_); RTK_ASSUME_LDS(&p); const int lvl = rtk_u(lvl_);
    const unsigned long long tc0 = rtk_clock();
    const uint32_t pn = rtk_ld(&p.n), pl = rtk_ld(&p.l), pq = rtk_ld(&p.qlen);
    const uint64_t off = rtk_arena_alloc(s, lvl, sizeof(PathHdr) + sizeof(UMap) * pn + pq);
    if (rtk_failed(s)) return 0;
    char* rec = rtk_ld(&s.arena[lvl]) + off;
    PathHdr* h = reinterpret_cast<PathHdr*>(rec);
    h->n = pn; h->l = pl; h->qlen = pq; h->pad = 0;
    rtk_wcopy2(rec + sizeof(PathHdr), rtk_ld(&p.ums), sizeof(UMap) * pn, rec + sizeof(PathHdr) + sizeof(UMap) * pn, rtk_ld(&p.qual), pq);
    s.cnt[11] += rtk_clock() - tc0;
    return rtk_mk_handle(lvl, off);
}

RTK_FN_LEAF void rtk_wp_load(RegionScratch& s_, WPath& p_, uint64_t h_) {
    RegionScratch& s = *rtk_u(&s_); RTK_ASSUME_LDS(&s); WPath& p = *rtk_u(&p_); RTK_ASSUME_LDS(&p); const uint64_t h = rtk_u(h_);
    const int lvl = rtk_h_lvl(h); const uint64_t off = rtk_h_off(h);
    const char* rec = rtk_ld(&s.arena[lvl]) + off;
    const PathHdr* hd = reinterpret_cast<const PathHdr*>(rec);
    const uint32_t hn = rtk_ld(&hd->n), hl = rtk_ld(&hd->l), hq = rtk_ld(&hd->qlen);
    if (hn > rtk_ld(&s.um_cap) || hq > rtk_ld(&s.str_cap)) { rtk_fail_ovf(s, 4); rtk_wp_clear(p); return; }
    const unsigned long long tc0 = rtk_clock();
    p.n = hn; p.l = hl; p.qlen = hq;
    rtk_wcopy2(rtk_ld(&p.ums), rec + sizeof(PathHdr), sizeof(UMap) * hn, rtk_ld(&p.qual), rec + sizeof(PathHdr) + sizeof(UMap) * hn, hq);
    s.cnt[11] += rtk_clock() - tc0;
}

RTK_DEV uint32_t rtk_rec_n(const RegionScratch& s, uint64_t h) { return rtk_ld(&rtk_path_hdr(s, rtk_h_lvl(h), rtk_h_off(h))->n); }
RTK_DEV uint32_t rtk_rec_l(const RegionScratch& s, uint64_t h) { return rtk_ld(&rtk_path_hdr(s, rtk_h_lvl(h), rtk_h_off(h))->l); }
RTK_DEV UMap rtk_rec_back(const RegionScratch& s, uint64_t h) { const int lv = rtk_h_lvl(h); const uint64_t o = rtk_h_off(h); const char* rec = rtk_ld(&s.arena[lv]) + o; return rtk_u(reinterpret_cast<const UMap*>(rec + sizeof(PathHdr))[rtk_ld(&reinterpret_cast<const PathHdr*>(rec)->n) - 1]); }

RTK_DEV uint32_t rtk_nkm_u(const RCtx& c, uint32_t u) { // k-mers of unitig u, uniform
    const uint64_t* uo = c.g.uoff.get() + u;
    return static_cast<uint32_t>(rtk_ld(uo + 1) - rtk_ld(uo)) - static_cast<uint32_t>(rtk_u(c.k)) + 1u;
}
RTK_DEV void rtk_wp_norm_back(const RCtx& c, WPath& p) { // the former end becomes a whole unitig (Path.hpp:319-323)
    const uint32_t pn = rtk_ld(&p.n);
    if (pn >= 2) { UMap* e = rtk_ld(&p.ums) + (pn - 1); e->dist = 0; e->len = rtk_nkm_u(c, rtk_ld(&e->unitig)); }
}

RTK_FN_HOT void rtk_wp_extend(const RCtx& c, WPath& p_, const UMap& um_) { // Path.hpp:308-330
    RegionScratch& s = *rtk_u(c.sc); WPath& p = *rtk_u(&p_); RTK_ASSUME_LDS(&p); const UMap um = rtk_u(um_);
    if (rtk_um_is_empty(um)) return;
    const uint32_t pn = rtk_ld(&p.n);
    if (pn >= rtk_ld(&s.um_cap)) { rtk_fail_ovf(s, 5); return; }
    UMap* ums = rtk_ld(&p.ums);
    if (pn == 0) { ums[0] = um; p.n = 1; p.l = um.len + static_cast<uint32_t>(rtk_u(c.k)) - 1; }
    else { rtk_wp_norm_back(c, p); ums[pn] = um; p.n = pn + 1; p.l = rtk_ld(&p.l) + um.len; }
}

// extend with a quality slice q[0..qn) (Path.hpp:332-363): appended only when its length equals um.len + k - 1
RTK_FN void rtk_wp_extend_q(const RCtx& c_, WPath& p_, UMap um_, const char* q_, uint32_t qn_) {
    const RCtx& c = *rtk_u(&c_); WPath& p = *rtk_u(&p_); RTK_ASSUME_LDS(&p); const UMap um = rtk_u(um_); const char* q = rtk_u(q_); uint32_t qn = rtk_u(qn_);
    RegionScratch& s = rtk_hdr(c);
    if (rtk_um_is_empty(um)) return;
    if (p.n >= s.um_cap) { rtk_fail_ovf(s, 5); return; }
    const uint32_t want = um.len + static_cast<uint32_t>(c.k) - 1;
    if (p.n == 0) {
        p.ums[0] = um; p.n = 1; p.l = want;
        if (qn == want) { if (qn > s.str_cap) { rtk_fail_ovf(s, 6); return; } rtk_wcopy(p.qual, q, qn); p.qlen = qn; }
    } else {
        rtk_wp_norm_back(c, p); p.ums[p.n] = um; ++p.n; p.l += um.len;
        if (qn == want) {
            const uint32_t add = qn - (static_cast<uint32_t>(c.k) - 1);
            if (p.qlen + add > s.str_cap) { rtk_fail_ovf(s, 6); return; }
            rtk_wcopy(p.qual + p.qlen, q + (c.k - 1), add); p.qlen += add;
        }
    }
}

// fills qual with `ch` for a fresh single-unitig path (string(len + k - 1, getQual(1.0)))
RTK_FN_LEAF void rtk_wp_start(const RCtx& c_, WPath& p_, UMap um_, char ch_) {
    const RCtx& c = *rtk_u(&c_); WPath& p = *rtk_u(&p_); RTK_ASSUME_LDS(&p); const UMap um = rtk_u(um_); char ch = rtk_u(ch_);
    RegionScratch& s = rtk_hdr(c);
    rtk_wp_clear(p);
    const uint32_t want = um.len + static_cast<uint32_t>(c.k) - 1;
    if (want > s.str_cap) { rtk_fail_ovf(s, 6); return; }
    p.ums[0] = um; p.n = 1; p.l = want;
    rtk_wfill(p.qual, ch, want); p.qlen = want;
}

// p.merge(o) where o is a committed record (Path.hpp:366-414)
RTK_FN void rtk_wp_merge(const RCtx& c_, WPath& p_, uint64_t ho_) {
    const RCtx& c = *rtk_u(&c_); WPath& p = *rtk_u(&p_); RTK_ASSUME_LDS(&p); uint64_t ho = rtk_u(ho_);
    RegionScratch& s = rtk_hdr(c);
    const int lv = rtk_h_lvl(ho); const uint64_t oo = rtk_h_off(ho);
    const PathHdr* o = rtk_path_hdr(s, lv, oo);
    const UMap* oums = rtk_path_ums(s, lv, oo);
    const char* oq = rtk_path_qual(s, lv, oo);
    if (o->l == 0) return;
    if (p.l == 0) { rtk_wp_load(s, p, ho); return; }
    if ((p.qlen == 0) != (o->qlen == 0)) return;
    const UMap last = p.ums[p.n - 1];
    if (last.unitig != oums[0].unitig || last.strand != oums[0].strand) return;
    if (p.n + o->n > s.um_cap) { rtk_fail_ovf(s, 5); return; }
    if (p.n == 1) {
        UMap& st = p.ums[0];
        if (!st.strand) st.dist = oums[0].dist;
        st.len += oums[0].len - 1;
        for (uint32_t i = 1; i < o->n; ++i) p.ums[p.n++] = oums[i];
    } else {
        UMap& en = p.ums[p.n - 1];
        if (!en.strand) en.dist = oums[0].dist;
        en.len += oums[0].len - 1;
        if (o->n >= 2) { rtk_wp_norm_back(c, p); for (uint32_t i = 1; i < o->n; ++i) p.ums[p.n++] = oums[i]; }
    }
    p.l += o->l - static_cast<uint32_t>(c.k);
    if (o->qlen != 0) {
        const uint32_t kk = static_cast<uint32_t>(c.k);
        const uint32_t add = o->qlen > kk ? o->qlen - kk : 0; // o.qual.substr(k)
        if (p.qlen + add > s.str_cap) { rtk_fail_ovf(s, 6); return; }
        rtk_wcopy(p.qual + p.qlen, oq + kk, add); p.qlen += add;
    }
}

RTK_FN void rtk_wp_prune_prefix(const RCtx& c_, WPath& p_, uint32_t len_) {
    const RCtx& c = *rtk_u(&c_); WPath& p = *rtk_u(&p_); RTK_ASSUME_LDS(&p); uint32_t len = rtk_u(len_); // Path.hpp:487-571
    if (p.n == 0 || p.l == 0 || len >= p.l) return;
    const uint32_t k = static_cast<uint32_t>(c.k);
    UMap& st = p.ums[0];
    if (p.n == 1) { if (!st.strand) st.dist += p.l - len; st.len -= p.l - len; }
    else if (st.len + k - 1 >= len) {
        p.l = st.len + k - 1; p.n = 1;
        if (!st.strand) st.dist += p.l - len;
        st.len -= p.l - len;
    } else if (p.n == 2 || len > (p.l - p.ums[p.n - 1].len)) {
        UMap& en = p.ums[p.n - 1];
        if (!en.strand) en.dist += p.l - len;
        en.len -= p.l - len;
    } else {
        uint32_t acc = st.len + k - 1, w = 1; bool cut = false;
        const UMap old_end = p.ums[p.n - 1];
        for (uint32_t i = 1; i + 1 < p.n; ++i) {
            UMap cur = p.ums[i]; cur.dist = 0; cur.len = rtk_nkm(c.g, cur.unitig);
            acc += cur.len;
            if (acc < len) { p.ums[w++] = cur; }
            else { if (!cur.strand) cur.dist += acc - len; cur.len -= acc - len; p.ums[w++] = cur; cut = true; break; }
        }
        if (!cut) p.ums[w++] = old_end;
        p.n = w;
    }
    p.l = len;
    if (p.qlen != 0 && p.qlen > p.l) p.qlen = p.l;
}

// mappedSequenceToString of one mapping into dst (lane-parallel 2-bit decode, reverse complement on the fly)
RTK_DEV void rtk_um_decode(const RCtx& c, const UMap& um, char* dst, uint32_t skip) {
    const uint32_t n = um.len + static_cast<uint32_t>(rtk_u(c.k)) - 1;
    const uint64_t b0 = rtk_ld(c.g.uoff.get() + um.unitig) + um.dist;
    const uint64_t* useq = c.g.useq.get();
    for (uint32_t i = skip + static_cast<uint32_t>(rtk_lane()); i < n; i += RTK_WAVE) {
        const uint64_t pos = um.strand ? (b0 + i) : (b0 + (n - 1 - i));
        const uint32_t b = static_cast<uint32_t>((useq[pos >> 5] >> (2 * (pos & 31))) & 3ull);
        const uint32_t code = um.strand ? b : (3u - b);
        dst[i - skip] = static_cast<char>((0x54474341u >> (8 * code)) & 0xFFu); // "ACGT"
    }
}

// Path::toString (Path.hpp:449-485) of `n` mappings into dst; returns length (0xFFFFFFFF on overflow)
RTK_FN uint32_t rtk_ums_to_string(const RCtx& c, const UMap* ums_, uint32_t n_, char* dst_) {
    RegionScratch& s = *rtk_u(c.sc); const UMap* ums = rtk_u(ums_); const uint32_t n = rtk_u(n_); char* dst = rtk_u(dst_);
    const unsigned long long tc0 = rtk_clock();
    uint32_t len = 0;
    const uint32_t k1 = static_cast<uint32_t>(rtk_u(c.k)) - 1, str_cap = rtk_ld(&s.str_cap);
    for (uint32_t i = 0; i < n; ++i) {
        const UMap um = rtk_u(ums[i]);
        const uint32_t skip = i ? k1 : 0;
        const uint32_t add = um.len + k1 - skip;
        if (len + add > str_cap) { rtk_fail_ovf(s, 7); return 0xFFFFFFFFu; }
        rtk_um_decode(c, um, dst + len, skip);
        len += add;
    }
    rtk_sync();
    s.cnt[2] += len;
    s.cnt[12] += rtk_clock() - tc0;
    return len;
}
RTK_DEV uint32_t rtk_rec_to_string(const RCtx& c, uint64_t h, char* dst) {
    const RegionScratch& s = *rtk_u(c.sc);
    return rtk_ums_to_string(c, rtk_path_ums(s, rtk_h_lvl(h), rtk_h_off(h)), rtk_rec_n(s, h), dst);
}

// Developer statistics (simulator build only): alignments by call site. RTK_SITE(id) names the site of the calls that follow.
#ifdef RTK_SIM
#include <atomic>
extern thread_local int rtk_sim_site;
extern std::atomic<unsigned long long> rtk_sim_site_stat[32][8]; // calls, 32-bit word-columns, sum m, sum n, stored sweeps, stored word-columns, bounded (k >= 0), m > 2048
#define RTK_SITE(id) (rtk_sim_site = (id))
static inline void rtk_site_note(uint32_t m, uint32_t n, int k, bool stored) {
    std::atomic<unsigned long long>* t = rtk_sim_site_stat[rtk_sim_site & 31];
    const unsigned long long cells = static_cast<unsigned long long>((m + 31) / 32) * n;
    if ((rtk_sim_site & 31) == 2) { int b = 0; while (b < 7 && (256u << b) <= n) ++b; rtk_sim_site_stat[26][b] += 1; rtk_sim_site_stat[27][b] += n; }
    t[0] += 1; t[1] += cells; t[2] += m; t[3] += n; if (stored) { t[4] += 1; t[5] += cells; } if (k >= 0) t[6] += 1; if (m > 2048) t[7] += 1;
}
#else
#define RTK_SITE(id) ((void)0)
#define rtk_site_note(m, n, k, stored) ((void)0)
#endif

RTK_FN_HOT MyersResult rtk_align(const RCtx& c, const char* q_, uint32_t m_, const char* t_, uint32_t n_, int kk_, int mode_, bool iupac_ = true) {
    RegionScratch& s = *rtk_u(c.sc); const char* q = rtk_u(q_); const char* t = rtk_u(t_);
    const uint32_t m = rtk_u(m_), n = rtk_u(n_); const int kk = rtk_u(kk_), mode = rtk_u(mode_); const bool iupac = rtk_u(iupac_);
    s.cnt[3] += 1; s.cnt[4] += static_cast<unsigned long long>((m + 63) / 64) * n;
    const unsigned long long t0 = rtk_clock();
    rtk_site_note(m, n, kk, false);
    const MyersResult r = rtk_myers_distance(s.my, q, static_cast<int>(m), t, static_cast<int>(n), kk, mode, iupac);
    s.cnt[9] += rtk_clock() - t0;
    return r;
}

// alignment with its moves (left in s.my.moves); counted like the distance call + path call pair it replaces
RTK_FN_HOT MyersResult rtk_align_path(const RCtx& c_, const char* q_, uint32_t m_, const char* t_, uint32_t n_, int mode_, uint32_t* n_moves_) {
    const RCtx& c = *rtk_u(&c_); RegionScratch& s = rtk_hdr(c); const char* q = rtk_u(q_); const char* t = rtk_u(t_);
    const uint32_t m = rtk_u(m_), n = rtk_u(n_); const int mode = rtk_u(mode_); uint32_t* n_moves = rtk_u(n_moves_);
    s.cnt[3] += (m > 0 && n > 0) ? 2 : 1; s.cnt[4] += static_cast<unsigned long long>((m + 63) / 64) * n;
    const unsigned long long t0 = rtk_clock();
    rtk_site_note(m, n, -1, true);
    const MyersResult r = rtk_myers_path(s.my, q, static_cast<int>(m), t, static_cast<int>(n), mode, true, n_moves);
    s.cnt[9] += rtk_clock() - t0;
    return r;
}

#include "rtk_ambiguity.h"

// ------------------------------------------------------------------------------------------------ candidate selection (src/Alignment.cpp:3-147, 967-1015)
// handles[] are committed paths; strings are materialised into str[0].
RTK_FN void rtk_select_best(const RCtx& c, const uint64_t* handles_, uint32_t n_, const char* ref_, uint32_t ref_len_, int mode_, double cut_, int* best_id, int* best_end) {
    RegionScratch& s = *rtk_u(c.sc); const uint64_t* handles = rtk_u(handles_); const uint32_t n = rtk_u(n_), ref_len = rtk_u(ref_len_); const char* ref = rtk_u(ref_);
    const int mode = rtk_u(mode_); const double cut = rtk_u(cut_);
    double best = 0.0; int bid = -1, bend = -1;
    char* const str0 = rtk_ld(&s.str[0]);
    for (uint32_t i = 0; i < n && !rtk_failed(s); ++i) {
        const uint32_t sl = rtk_rec_to_string(c, rtk_ld(handles + i), str0);
        if (sl == 0xFFFFFFFFu) break;
        const uint32_t norm = (mode == RTK_MODE_NW) ? (sl > ref_len ? sl : ref_len) : sl;
        if (i == 0) {
            const MyersResult a = rtk_align(c, str0, sl, ref, ref_len, -1, mode);
            best = static_cast<double>(rtk_u(a.dist)) / static_cast<double>(norm); bend = rtk_u(a.first); bid = 0;
        } else {
            const int kk = static_cast<int>(best * static_cast<double>(norm) + 1.0); // G5: double -> int as edlibNewAlignConfig receives it
            const MyersResult a = rtk_align(c, str0, sl, ref, ref_len, kk, mode);
            const int ad = rtk_u(a.dist);
            if (ad >= 0 && (static_cast<double>(ad) / static_cast<double>(norm)) < best) { best = static_cast<double>(ad) / static_cast<double>(norm); bend = rtk_u(a.first); bid = static_cast<int>(i); }
        }
    }
    if (mode != RTK_MODE_NW && cut > 0.0 && best > cut) { bid = -1; bend = -1; }
    *best_id = bid; *best_end = bend;
}

// ------------------------------------------------------------------------------------------------ scoring (src/GraphTraversal.cpp:867-909, 722-772)
// path string must already be in str[1] (length sl)
RTK_FN_HOT double rtk_score_path(const RCtx& c, uint32_t sl_, const char* ref_, uint32_t ref_len_, bool terminal_) {
    RegionScratch& s = *rtk_u(c.sc); const uint32_t sl = rtk_u(sl_), ref_len = rtk_u(ref_len_); const char* ref = rtk_u(ref_); const bool terminal = rtk_u(terminal_);
    double score = 0.0;
    if (sl != 0) {
        const char* const str1 = rtk_ld(&s.str[1]);
        if (terminal) { RTK_SITE(1); const MyersResult a = rtk_align(c, str1, sl, ref, ref_len, -1, RTK_MODE_NW); score = 1.0 - (static_cast<double>(rtk_u(a.dist)) / static_cast<double>(sl)); }
        else if (sl >= ref_len) { RTK_SITE(2); const MyersResult a = rtk_align(c, ref, ref_len, str1, sl, -1, RTK_MODE_HW); score = 1.0 - (static_cast<double>(rtk_u(a.dist)) / static_cast<double>(ref_len)); }
        else {
            const uint64_t cap = static_cast<uint64_t>(static_cast<double>(sl) * (1.0 + rtk_u(c.o.weak_region_len_factor)));
            const uint32_t l_ref_len = ref_len < cap ? ref_len : static_cast<uint32_t>(cap);
            RTK_SITE(3); const MyersResult a = rtk_align(c, str1, sl, ref, l_ref_len, -1, RTK_MODE_HW);
            score = 1.0 - (static_cast<double>(rtk_u(a.dist)) / static_cast<double>(sl));
        }
        score = score > 0.0 ? score : 0.0; score = score < 1.0 ? score : 1.0;
    }
    return score;
}

// quality string of a path (SHW path alignment against ref) written to qout[0..sl); path string in str[1]
RTK_FN void rtk_score_path_qual(const RCtx& c, uint32_t sl_, const char* ref_, uint32_t ref_len_, double score_best_, double score_second_, char* qout_, const MyersSaved* saved_ = nullptr) {
    RegionScratch& s = *rtk_u(c.sc); const uint32_t sl = rtk_u(sl_), ref_len = rtk_u(ref_len_); const char* ref = rtk_u(ref_); char* qout = rtk_u(qout_); const MyersSaved* saved = rtk_u(saved_);
    const double score_best = rtk_u(score_best_), score_second = rtk_u(score_second_);
    const unsigned long long tq0 = rtk_clock();
    const double score_comp = score_best * ((score_best == 0.0) ? 0.0 : (1.0 - (score_second / score_best)));
    const char* const str1 = rtk_ld(&s.str[1]);
    uint32_t nm = 0;
    bool resumed = false;
    if (saved && saved->valid && static_cast<uint32_t>(saved->m) == sl && static_cast<uint32_t>(saved->n) == ref_len) { // the sweep that scored this very path is still in the table
        MyersResult r0; const unsigned long long t0 = rtk_clock();
        resumed = rtk_myers_path_from_saved(s.my, *saved, &nm, &r0);
        s.cnt[9] += rtk_clock() - t0;
    }
    if (!resumed) { RTK_SITE(4); rtk_align_path(c, str1, sl, ref, ref_len, RTK_MODE_SHW, &nm); }
    nm = rtk_u(nm);
    const char c_best = rtk_get_qual(score_best, 0, static_cast<uint64_t>(rtk_u(c.o.max_qual)));
    rtk_wfill(qout, rtk_get_qual(score_comp, static_cast<uint64_t>(rtk_u(c.o.out_qual)), static_cast<uint64_t>(rtk_u(c.o.max_qual))), sl);
    // walk the moves: a base gets the best-score quality when it sits on an identical reference base in an M run.
    // query/reference positions of every move come from a prefix count of the moves (chunked wave scan).
    uint32_t qp = 0, rp = 0;
    const uint8_t* mv = rtk_ld(&s.my.moves);
    for (uint32_t i0 = 0; i0 < nm; i0 += RTK_WAVE) {
        const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
        const uint8_t m = i < nm ? mv[i] : 255;
        const bool isq = (m == 0 || m == 3 || m == 1), isr = (m == 0 || m == 3 || m == 2);
        const uint64_t bq = rtk_ballot(isq), br = rtk_ballot(isr);
        const uint64_t lt = (1ull << rtk_lane()) - 1ull;
        const uint32_t myq = qp + static_cast<uint32_t>(rtk_popc(bq & lt)), myr = rp + static_cast<uint32_t>(rtk_popc(br & lt));
        if ((m == 0 || m == 3) && str1[myq] == ref[myr]) qout[myq] = c_best;
        qp += static_cast<uint32_t>(rtk_popc(bq)); rp += static_cast<uint32_t>(rtk_popc(br));
    }
    rtk_sync();
    s.cnt[13] += rtk_clock() - tq0;
}

// ------------------------------------------------------------------------------------------------ colour memo (src/GraphTraversal.cpp:485-487)
RTK_FN_HOT bool rtk_colour_ok(const RCtx& c, uint32_t u_, const uint32_t* all_pids_, uint32_t n_all_) {
    RegionScratch& s = *rtk_u(c.sc); const unsigned long long tk0 = rtk_clock(); const uint32_t u = rtk_u(u_), n_all = rtk_u(n_all_); const uint32_t* all_pids = rtk_u(all_pids_);
    const uint32_t mn = rtk_ld(&s.memo_n); const uint32_t* mu = rtk_ld(&s.memo_u); uint8_t* mvv = rtk_ld(&s.memo_v);
    for (uint32_t i0 = 0; i0 < mn; i0 += RTK_WAVE) { // 64 memo entries per step
        const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
        const uint64_t hit = rtk_ballot(i < mn && mu[i] == u);
        if (hit) { s.cnt[15] += rtk_clock() - tk0; return rtk_ld(mvv + i0 + static_cast<uint32_t>(rtk_ffs(hit) - 1)) != 0; }
    }
    const uint32_t mcv = static_cast<uint32_t>(rtk_u(c.o.min_cov_vertices));
    const bool ok = (n_all == 0) || (rtk_u(rtk_shared_with_set(c.g, u, all_pids, n_all, mcv)) >= mcv);
    s.cnt[1] += rtk_ld(c.g.card.get() + u) + n_all;
    if (mn < rtk_ld(&s.memo_cap)) { const_cast<uint32_t*>(mu)[mn] = u; mvv[mn] = ok ? 1 : 0; s.memo_n = mn + 1; rtk_sync(); }
    s.cnt[15] += rtk_clock() - tk0;
    return ok;
}

RTK_DEV bool rtk_edge_bit(const GraphView& g, uint32_t u, uint32_t strand, int base) { // UnitigData::getSharedPids (UnitigData.hpp:275-284)
    const uint32_t idx = 1u << base;
    return strand ? ((g.flags[u] & (idx << 4)) != 0) : ((g.flags[u] & idx) != 0);
}
RTK_DEV int rtk_nb_successors(const GraphView& g, const UMap& um) {
    const uint32_t* a = g.adj.get() + 8ull * um.unitig + (um.strand ? 0 : 4);
    int n = 0; for (int b = 0; b < 4; ++b) n += (rtk_ld(a + b) != RTK_NONE32) ? 1 : 0; return n;
}

// ------------------------------------------------------------------------------------------------ DFS (src/GraphTraversal.cpp:456-587)
// Results: handles of terminal / non-terminal paths (level-2 arena) in list[2] / list[3]; returns counts and best scores.
struct DfsOut { uint32_t n_t, n_nt; double t1, nt1, nt2; uint32_t nt_score_deferred, nt_qual_deferred; };
// What the caller needs to finish a non-terminal sub-path later (lazy evaluation, see rtk_explore_subgraph): where its reference
// window starts, its scores (or "not scored yet") and whether its quality string is still to be written.
struct NtPending { uint32_t e; double nt1, nt2; uint32_t score_deferred, qual_deferred; };

RTK_FN_SEARCH DfsOut rtk_explore_subgraph(const RCtx& c, const uint32_t* all_pids_, uint32_t n_all_, const char* ref_, uint32_t ref_len_, uint32_t max_len_path_,
                                    const UMap& um_, const UMap& um_e_, uint32_t level_) {
    // LAZY NON-TERMINAL PATHS. In explorePathsBFS2 a non-terminal sub-path of a DFS call is only used when the queue entry built from
    // it is popped while still shorter than max_len_path (src/GraphTraversal.cpp:364-366, 393-411); its score (HW alignment of the
    // reference window inside a path of four whole unitigs) and its quality string (SHW path alignment + traceback) are consumed by
    // nothing else when it is the ONLY non-terminal candidate of the call: the >= / > bookkeeping of :540-549 has nobody to compare it
    // with, `nt1 < min_score` (:295) cannot hold for min_score <= 0, selectBestSubstringAlignment (:297-300) needs two candidates.
    // So with an end anchor the candidates are collected first; several candidates are scored as the reference does, a single one is
    // handed back unscored, and in both cases the quality string is left to the caller (rtk_explore_paths), which computes score and
    // quality -- same inputs, same values -- only if the path is really extended. Without end anchor (explorePathsBFS) every
    // extension is a candidate at once (:165-172): everything stays eager there.
    RegionScratch& s = *rtk_u(c.sc);
    const uint32_t* all_pids = rtk_u(all_pids_); const char* ref = rtk_u(ref_);
    const uint32_t n_all = rtk_u(n_all_), ref_len = rtk_u(ref_len_), max_len_path = rtk_u(max_len_path_), level = rtk_u(level_);
    const UMap um = rtk_u(um_), um_e = rtk_u(um_e_);
    DfsOut out; out.n_t = 0; out.n_nt = 0; out.t1 = 0.0; out.nt1 = 0.0; out.nt2 = 0.0; out.nt_score_deferred = 0; out.nt_qual_deferred = 0;
    double score_t1 = 0.0, score_nt1 = 0.0, score_t2 = 0.0, score_nt2 = 0.0;
    uint32_t n_t = 0, n_nt = 0;
    s.top[2] = 0;
    uint64_t* T = rtk_ld(&s.list[2]); uint64_t* NT = rtk_ld(&s.list[3]);
    uint64_t* stk = rtk_ld(&s.list[4]); uint32_t sp = 0; // entries: handle (0 = empty path) and level, two words each
    const uint32_t list_cap = rtk_ld(&s.list_cap);
    char* const str1 = rtk_ld(&s.str[1]); char* const str2 = rtk_ld(&s.str[2]);
    const uint32_t* const g_adj = c.g.adj.get(); const uint32_t* const g_flags = c.g.flags.get();
    stk[0] = ~0ull; stk[1] = level; sp = 1;
    WPath& w = s.wp[2];
    const bool has_end = !rtk_um_is_empty(um_e);
    const bool lazy_nt = has_end && !(rtk_u(c.o.min_score) > 0.0);
    const bool lrc = rtk_u(c.o.long_read_correct) != 0;
    const uint32_t max_len_subpath = static_cast<uint32_t>(static_cast<uint64_t>(static_cast<double>(rtk_u(c.k)) * rtk_u(c.o.large_k_factor)));
    uint32_t n_nt_live = 0, n_t_scored = 0;
    MyersSaved& t_saved = s.loc.saved; t_saved.stash = reinterpret_cast<uint8_t*>(rtk_ld(&s.str[3])); t_saved.stash_cap = rtk_ld(&s.str_cap); t_saved.stash_n = 0; t_saved.valid = 0; t_saved.gen = 0; t_saved.m = 0; t_saved.n = 0; t_saved.nw_dist = 0; t_saved.shw.dist = -1; t_saved.shw.first = -1; t_saved.shw.last = -1; t_saved.shw.nloc = 0;
    unsigned long long n_exp = 0;
    const unsigned long long td0 = rtk_clock(); const unsigned long long my0 = s.cnt[9];
#ifdef RTK_SIM
    const unsigned long long dfs_al0 = s.cnt[3];
#endif
    // Walk 0 prunes (lazy mode only): an extension already longer than max_len_path can neither reach a terminal path that passes the
    // length test of :511 nor a non-terminal leaf that would ever be looked at again, so its subtree is skipped -- unless a LIVE
    // non-terminal candidate turns up, in which case the skipped candidates' scores can decide the survivor and walk 1 repeats the
    // reference's full walk for the non-terminal candidates only (terminal ones are complete after walk 0).
    uint32_t n_pruned = 0;
    for (int walk = 0; walk < 2 && !rtk_failed(s); ++walk) {
    const bool prune = lazy_nt && walk == 0, do_terminal = walk == 0;
    if (walk == 1) { if (!(lazy_nt && n_nt_live > 0 && n_pruned > 0)) break;
#ifdef RTK_SIM
        rtk_sim_site_stat[28][0] += 1;
#endif
        n_nt = 0; n_nt_live = 0; stk[0] = ~0ull; stk[1] = level; sp = 1; }
    while (sp > 0 && !rtk_failed(s)) {
        --sp;
        RTK_PL(s, 15);
        const uint64_t hp = rtk_ld(stk + 2 * sp); const uint32_t lvl = static_cast<uint32_t>(rtk_ld(stk + 2 * sp + 1));
        const UMap um_start = (hp == ~0ull) ? um : rtk_rec_back(s, hp);
        const uint32_t* adj = g_adj + 8ull * um_start.unitig + (um_start.strand ? 0 : 4);
        ++n_exp;
        // the four neighbour slots and the edge bits of this unitig, fetched together
        const uint32_t a4[4] = { rtk_ld(adj), rtk_ld(adj + 1), rtk_ld(adj + 2), rtk_ld(adj + 3) };
        const uint32_t eb = (rtk_ld(g_flags + um_start.unitig) >> (um_start.strand ? 4 : 0)) & 0xFu; // UnitigData::getSharedPids (UnitigData.hpp:275-284)
        const bool rev_order = rtk_u(c.o.a3_strand_order) != 0 && !um_start.strand; // [A3] switch: slot = base appended in walk direction (A,C,G,T)
        RTK_PL(s, 8);
        for (int bi = 0; bi < 4 && !rtk_failed(s); ++bi) {
            const int b = rev_order ? 3 - bi : bi;
            const uint32_t ab = a4[b];
            if (ab == RTK_NONE32) continue;
            UMap sc; sc.unitig = ab >> 1; sc.strand = ab & 1u; sc.dist = 0; sc.len = rtk_nkm_u(c, sc.unitig);
            const bool col_ok = rtk_u(rtk_colour_ok(c, sc.unitig, all_pids, n_all));
            RTK_PL(s, 9);
            if (!(((eb >> b) & 1u) && col_ok)) continue;
            if (do_terminal && has_end && sc.unitig == um_e.unitig && um_e.strand == sc.strand) { // terminal
                if (hp == ~0ull) rtk_wp_clear(w); else rtk_wp_load(s, w, hp);
                UMap pref = sc;
                if (pref.strand) { pref.dist = 0; pref.len = um_e.dist + 1; } else { pref.dist = um_e.dist; pref.len = sc.len - um_e.dist; }
                rtk_wp_extend(c, w, pref);
                RTK_PL(s, 10);
                if (rtk_ld(&w.l) <= max_len_path && !rtk_failed(s)) {
                    const uint32_t sl = rtk_u(rtk_ums_to_string(c, rtk_ld(&w.ums), rtk_ld(&w.n), str1));
                    if (sl == 0xFFFFFFFFu) break;
                    RTK_PL(s, 11);
                    // the first terminal candidate of a call -- usually the only one -- is scored by a stored sweep that its quality
                    // string can be read from afterwards (rtk_myers_nw_and_save); further candidates overwrite nothing
                    double sco;
                    ++n_t_scored;
                    if (n_t_scored == 1 && sl != 0 && rtk_myers_nw_and_save(s.my, str1, static_cast<int>(sl), ref, static_cast<int>(ref_len), true, &t_saved)) {
                        s.cnt[3] += 1; s.cnt[4] += static_cast<unsigned long long>((sl + 63) / 64) * ref_len;
                        sco = 1.0 - (static_cast<double>(rtk_u(t_saved.nw_dist)) / static_cast<double>(sl));
                        sco = sco > 0.0 ? sco : 0.0; sco = sco < 1.0 ? sco : 1.0;
                    } else sco = rtk_u(rtk_score_path(c, sl, ref, ref_len, true));
                    RTK_PL(s, 12);
                    if (sco >= score_t1) {
                        if (sco > score_t1) n_t = 0;
                        if (n_t >= list_cap) { rtk_fail_ovf(s, 8); break; }
                        T[n_t++] = rtk_wp_commit(s, w, 2);
                        score_t2 = score_t1; score_t1 = sco;
                    } else if (sco > score_t2) score_t2 = sco;
                    RTK_PL(s, 13);
                }
            }
            { // non-terminal
                if (prune) { // length of the extension (Path::extend, Path.hpp:319-330) before building it
                    const uint32_t l_new = (hp == ~0ull) ? (sc.len + static_cast<uint32_t>(rtk_u(c.k)) - 1u) : (rtk_rec_l(s, hp) + sc.len);
                    if (l_new > max_len_path) { ++n_pruned; continue; }
                }
                if (hp == ~0ull) rtk_wp_clear(w); else rtk_wp_load(s, w, hp);
                rtk_wp_extend(c, w, sc);
                if (rtk_failed(s)) break;
                RTK_PL(s, 14);
#ifdef RTK_SIM
                rtk_sim_site_stat[20][0] += 1; rtk_sim_site_stat[20][1] += sc.len + ((hp == ~0ull) ? static_cast<uint32_t>(rtk_u(c.k)) - 1 : 0); // DFS tree nodes and the columns they add
#endif
                // exploreSubGraph descends `level` unitigs (:531-535), exploreSubGraphLong (pass 2) until the sub-path spans k * large_k_factor (:594, :669-671)
                const bool deeper = lrc ? (rtk_ld(&w.l) < max_len_subpath) : (lvl != 0);
                if (deeper) {
                    if (2 * (sp + 1) > list_cap) { rtk_fail_ovf(s, 8); break; }
                    stk[2 * sp] = rtk_wp_commit(s, w, 2); stk[2 * sp + 1] = lvl ? lvl - 1 : 0; ++sp;
                } else if (rtk_nb_successors(c.g, sc) > 0) {
                    if (lazy_nt) { // candidate kept in discovery order, scored after the walk (or never)
                        if (n_nt >= list_cap) { rtk_fail_ovf(s, 8); break; }
                        NT[n_nt++] = rtk_wp_commit(s, w, 2);
                        // P (+) Q is looked at again only if it is shorter than the caller's max_len_path (:364-366); in terms of this call's
                        // arguments (max_len_path here = the caller's minus the characters of P before its last unitig `um`): l(Q) + um.len < max_len_path
                        if (rtk_ld(&w.l) + um.len < max_len_path) ++n_nt_live;
                    } else {
                        const uint32_t sl = rtk_u(rtk_ums_to_string(c, rtk_ld(&w.ums), rtk_ld(&w.n), str1));
                        if (sl == 0xFFFFFFFFu) break;
                        const double sco = rtk_u(rtk_score_path(c, sl, ref, ref_len, false));
                        if (sco >= score_nt1) {
                            if (sco > score_nt1) n_nt = 0;
                            if (n_nt >= list_cap) { rtk_fail_ovf(s, 8); break; }
                            NT[n_nt++] = rtk_wp_commit(s, w, 2);
                            score_nt2 = score_nt1; score_nt1 = sco;
                        } else if (sco > score_nt2) score_nt2 = sco;
                    }
                }
            }
        }
    }
    } // walk
#ifdef RTK_SIM
    { const unsigned long long na = s.cnt[3] - dfs_al0; const unsigned b = na > 15 ? 15 : static_cast<unsigned>(na); rtk_sim_site_stat[21][0] += 1; rtk_sim_site_stat[22 + (b >> 3)][b & 7] += 1; rtk_sim_site_stat[24 + (b >> 3)][b & 7] += na; }
#endif
    RTK_PL(s, 15);
    s.cnt[0] += n_exp; RTK_HIST_ADD(s, 31, 1);
    s.cnt[14] += (rtk_clock() - td0) - (s.cnt[9] - my0); // DFS bookkeeping: loop time minus the alignments inside it
    bool nt_score_deferred = false;
    if (lazy_nt && !rtk_failed(s)) {
        // whichever candidate survives the scoring is only re-queued; if none of them can pass the length test of the pop, the queue
        // ends empty whatever the scores are: nothing to compute (a mix of short and long candidates still needs every score)
        if (n_nt_live == 0) n_nt = 0;
        if (n_nt == 1) nt_score_deferred = true; // nobody to compare it with: scored by the caller if the path is ever extended
        else if (n_nt > 1) { // the reference's bookkeeping (:540-549) over the candidates in discovery order
            const uint32_t n_cand = n_nt; n_nt = 0;
            for (uint32_t i = 0; i < n_cand && !rtk_failed(s); ++i) {
                const uint64_t hc = rtk_ld(NT + i);
                rtk_wp_load(s, w, hc);
                const uint32_t sl = rtk_u(rtk_ums_to_string(c, rtk_ld(&w.ums), rtk_ld(&w.n), str1));
                if (sl == 0xFFFFFFFFu) break;
                const double sco = rtk_u(rtk_score_path(c, sl, ref, ref_len, false));
                if (sco >= score_nt1) {
                    if (sco > score_nt1) n_nt = 0;
                    NT[n_nt++] = hc; // n_nt <= i: survivors move towards the front
                    score_nt2 = score_nt1; score_nt1 = sco;
                } else if (sco > score_nt2) score_nt2 = sco;
            }
        }
    }
    RTK_PL(s, 16);
    // qualities (:556-584): re-commit every surviving path with its quality string (non-terminal ones: left to the caller when lazy)
    for (int which = 0; which < (lazy_nt ? 1 : 2) && !rtk_failed(s); ++which) {
        uint64_t* L = which ? NT : T; const uint32_t nL = which ? n_nt : n_t;
        for (uint32_t i = 0; i < nL && !rtk_failed(s); ++i) {
            rtk_wp_load(s, w, rtk_ld(L + i));
            const uint32_t sl = rtk_u(rtk_ums_to_string(c, rtk_ld(&w.ums), rtk_ld(&w.n), str1));
            if (sl == 0xFFFFFFFFu || sl > rtk_ld(&s.str_cap)) { rtk_fail_ovf(s, 7); break; }
            RTK_PL(s, 17);
            rtk_score_path_qual(c, sl, ref, ref_len, which ? score_nt1 : score_t1, which ? score_nt2 : score_t2, str2, (which == 0 && n_t_scored == 1) ? &t_saved : nullptr);
            RTK_PL(s, 18);
            if (sl == rtk_ld(&w.l)) { rtk_wcopy(rtk_ld(&w.qual), str2, sl); w.qlen = sl; } // Path::setQuality only accepts q.length() == l
            L[i] = rtk_wp_commit(s, w, 2);
            RTK_PL(s, 19);
        }
    }
    out.n_t = n_t; out.n_nt = n_nt; out.t1 = score_t1; out.nt1 = score_nt1; out.nt2 = score_nt2;
    out.nt_score_deferred = nt_score_deferred ? 1u : 0u; out.nt_qual_deferred = (lazy_nt && n_nt != 0) ? 1u : 0u;
    return out;
}

// explore() (src/GraphTraversal.cpp:41-93, 251-304). p = committed path (level 1). Results stay in list[2]/list[3] (level-2 arena).
RTK_FN_SEARCH void rtk_explore(const RCtx& c_, const uint32_t* all_pids_, uint32_t n_all_, const char* ref_, uint32_t ref_len_, const UMap& um_e_, uint64_t hp_, uint32_t max_len_path_, uint32_t* n_t_, uint32_t* n_nt_, NtPending* pend_) {
    const RCtx& c = *rtk_u(&c_); const uint32_t* all_pids = rtk_u(all_pids_); uint32_t n_all = rtk_u(n_all_); const char* ref = rtk_u(ref_); uint32_t ref_len = rtk_u(ref_len_); const UMap um_e = rtk_u(um_e_); uint64_t hp = rtk_u(hp_); uint32_t max_len_path = rtk_u(max_len_path_); uint32_t* n_t = rtk_u(n_t_); uint32_t* n_nt = rtk_u(n_nt_); NtPending* pend = rtk_u(pend_);
    RegionScratch& s = rtk_hdr(c);
    *n_t = 0; *n_nt = 0; pend->e = 0; pend->nt1 = 0.0; pend->nt2 = 0.0; pend->score_deferred = 0; pend->qual_deferred = 0;
    const UMap um = rtk_rec_back(s, hp);
    const uint32_t path_len = rtk_rec_l(s, hp);
    const uint32_t k = static_cast<uint32_t>(c.k);
    const bool non_empty_path = (path_len > (um.len + k - 1)) && !rtk_um_is_empty(um);
    const uint32_t path_len_prefix = non_empty_path ? (path_len - um.len - k + 1) : 0;
    uint32_t end_pos_ref = 0;
    if (non_empty_path) {
        const uint32_t sl = rtk_rec_to_string(c, hp, s.str[0]);
        if (sl == 0xFFFFFFFFu) return;
        RTK_SITE(5); const MyersResult a = rtk_align(c, s.str[0], path_len_prefix, ref, ref_len, -1, RTK_MODE_SHW);
        end_pos_ref = static_cast<uint32_t>(a.first + 1);
    }
    RTK_PL(s, 7);
    if ((ref_len - end_pos_ref) != 0 && path_len < max_len_path) {
        DfsOut o = rtk_explore_subgraph(c, all_pids, n_all, ref + end_pos_ref, ref_len - end_pos_ref, max_len_path - path_len_prefix, um, um_e, 3);
        if (rtk_failed(s)) return;
        if (o.n_t && o.t1 < c.o.min_score) o.n_t = 0;
        if (o.n_nt && !o.nt_score_deferred && o.nt1 < c.o.min_score) o.n_nt = 0; // a deferred score only exists for min_score <= 0: never below it
        pend->e = end_pos_ref; pend->nt1 = o.nt1; pend->nt2 = o.nt2; pend->score_deferred = o.nt_score_deferred; pend->qual_deferred = o.nt_qual_deferred;
        if (o.n_nt > 1) {
            int bid, bend;
            RTK_SITE(6); rtk_select_best(c, s.list[3], o.n_nt, ref + end_pos_ref, ref_len - end_pos_ref, RTK_MODE_HW, -1.0, &bid, &bend);
            s.list[3][0] = s.list[3][bid]; o.n_nt = 1;
        }
        *n_t = o.n_t; *n_nt = o.n_nt;
    }
}

// P (+) Q: w = copy of p extended by every mapping of sub with its quality slice (src/GraphTraversal.cpp:379-390)
RTK_FN_LEAF void rtk_extend_by(const RCtx& c_, WPath& w_, uint64_t hsub_, uint32_t upto_) {
    const RCtx& c = *rtk_u(&c_); WPath& w = *rtk_u(&w_); RTK_ASSUME_LDS(&w); uint64_t hsub = rtk_u(hsub_); uint32_t upto = rtk_u(upto_);
    RegionScratch& s = rtk_hdr(c);
    const int lv = rtk_h_lvl(hsub); const uint64_t oo = rtk_h_off(hsub);
    const PathHdr* h = rtk_path_hdr(s, lv, oo); const UMap* ums = rtk_path_ums(s, lv, oo); const char* q = rtk_path_qual(s, lv, oo);
    uint32_t j = 0;
    for (uint32_t i = 0; i < h->n && i < upto && !rtk_failed(s); ++i) {
        const uint32_t want = ums[i].len + static_cast<uint32_t>(c.k) - 1;
        uint32_t qn = 0;
        if (j <= h->qlen) qn = (h->qlen - j) < want ? (h->qlen - j) : want; // std::string::substr clamps
        rtk_wp_extend_q(c, w, ums[i], q + j, qn);
        j += ums[i].len;
    }
}

RTK_FN void rtk_resize_to_best(const RCtx& c_, uint64_t* v_, uint32_t* n_, const char* ref_, uint32_t ref_len_) {
    const RCtx& c = *rtk_u(&c_); uint64_t* v = rtk_u(v_); uint32_t* n = rtk_u(n_); const char* ref = rtk_u(ref_); uint32_t ref_len = rtk_u(ref_len_); // resizeVector
    if (*n <= 1) return;
    int bid, bend;
    RTK_SITE(7); rtk_select_best(c, v, *n, ref, ref_len, RTK_MODE_SHW, -1.0, &bid, &bend);
    if (rtk_failed(*c.sc)) return;
    v[0] = v[bid]; *n = 1;
}

RTK_DEV UMap rtk_start_suffix(const RCtx& c, const UMap& um_s) { // src/GraphTraversal.cpp:113-125, 325-338
    UMap t = um_s;
    if (t.strand) { t.dist += t.len - 1; t.len = rtk_nkm(c.g, um_s.unitig) - t.dist; }
    else { t.len = um_s.dist + 1; t.dist = 0; }
    return t;
}

// explorePathsBFS2 / explorePathsBFS. Returns a level-1 handle of the single resulting path, or ~0 if none.
// ------------------------------------------------------------------------------------------------ fixRepeats (src/GraphTraversal.cpp:1149-1334)
// For every unitig of the path that lies on a short cycle (micro / mini-satellite motif) the stored compact cycles are tried as one
// more turn through it: prefix + [unitig to its end, cycle unitigs, unitig from its start] + suffix; a turn is kept when it lowers the
// NW distance to the read window (bounded by the distance so far). Identity when no unitig of the path is flagged.
// is any unitig of the path on a short cycle? (the fast way out of fixRepeats, tested by the caller so that the common case costs no call)
RTK_DEV bool rtk_path_has_short_cycle(const RCtx& c, uint64_t hp) {
    RegionScratch& s = rtk_hdr(c); const GraphView& g = c.g;
    const int lv = rtk_h_lvl(hp); const uint64_t oo = rtk_h_off(hp);
    const UMap* pu = rtk_path_ums(s, lv, oo); const uint32_t pn = rtk_rec_n(s, hp);
    bool any = false;
    for (uint32_t i0 = 0; i0 < pn && !any; i0 += RTK_WAVE) { const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane()); any = rtk_ballot(i < pn && (g.flags[pu[i].unitig] & RTK_F_SHORT_CYCLE)) != 0ull; }
    return any;
}
RTK_FN uint64_t rtk_fix_repeats(const RCtx& c_, uint64_t hp_, const char* ref_, uint32_t ref_len_) {
    const RCtx& c = *rtk_u(&c_); const uint64_t hp = rtk_u(hp_); const char* ref = rtk_u(ref_); const uint32_t ref_len = rtk_u(ref_len_);
    RegionScratch& s = rtk_hdr(c);
    const GraphView& g = c.g;
    const uint32_t k = static_cast<uint32_t>(c.k);
    WPath& P = s.wp[1]; WPath& E = s.wp[2]; UMap* R = s.wp[3].ums;
    rtk_wp_load(s, P, hp);
    if (rtk_failed(s)) return ~0ull;
    const char q_max = rtk_get_qual(1.0, 0, static_cast<uint64_t>(c.o.max_qual));
    int ed;
    { const uint32_t sl = rtk_ums_to_string(c, P.ums, P.n, s.str[0]); if (sl == 0xFFFFFFFFu) return ~0ull; RTK_SITE(8); ed = rtk_u(rtk_align(c, s.str[0], sl, ref, ref_len, -1, RTK_MODE_NW).dist); }
    for (uint32_t i = 0; i < P.n && !rtk_failed(s); ++i) {
        const UMap um_path = rtk_u(P.ums[i]);
        if (!(g.flags[um_path.unitig] & RTK_F_SHORT_CYCLE)) continue;
        uint64_t best_h = ~0ull;
        UMap um_start = um_path, um_end = um_path; // the unitig from the mapped start to its end / from its beginning to the mapped end, both forward (:1213-1224)
        um_start.len = rtk_nkm(g, um_path.unitig) - um_path.dist; um_start.strand = 1;
        um_end.dist = 0; um_end.len = um_path.dist + um_path.len; um_end.strand = 1;
        const char* cyc = g.cyc; const uint64_t c_lo = g.cycoff[um_path.unitig], c_hi = g.cycoff[um_path.unitig + 1];
        for (uint64_t a = c_lo; a < c_hi && !rtk_failed(s);) {
            // Path(um_start, cycle, um_end) (Path.hpp:109-152) as an explicit unitig list R
            uint32_t nR = 0, rep_l = um_start.len + k - 1; bool ok = true;
            if (s.um_cap < 4) { rtk_fail_ovf(s, 5); break; }
            R[nR++] = um_start;
            UMap curr = um_start;
            uint64_t e = a;
            for (; e < c_hi; ++e) {
                const char ch = rtk_ld(cyc + e);
                if (ch == 0) break;
                const uint32_t nb = rtk_ld(g.adj + 8ull * curr.unitig + (curr.strand ? 0 : 4) + (((static_cast<uint32_t>(ch) >> 1) & 3u) ^ (((static_cast<uint32_t>(ch) >> 1) & 3u) >> 1))); // A,C,G,T -> 0..3
                if (nb == RTK_NONE32) { ok = false; continue; }
                if (!ok) continue;
                curr.unitig = nb >> 1; curr.strand = nb & 1u; curr.dist = 0; curr.len = rtk_nkm(g, curr.unitig);
                if (nR + 2 > s.um_cap) { rtk_fail_ovf(s, 5); break; }
                R[nR++] = curr; rep_l += curr.len;
            }
            a = e + 1;
            if (rtk_failed(s)) break;
            if (ok) { R[nR++] = um_end; rep_l += um_end.len; } else { nR = 0; rep_l = 0; }
            rtk_sync();
            if (!um_path.strand) { // rev_comp (Path.hpp:208-262): reversed order, flipped strands
                for (uint32_t x = 0; x < nR / 2; ++x) { const UMap t = rtk_u(R[x]); R[x] = R[nR - 1 - x]; R[nR - 1 - x] = t; }
                rtk_sync();
                for (uint32_t x = static_cast<uint32_t>(rtk_lane()); x < nR; x += RTK_WAVE) R[x].strand ^= 1u;
                rtk_sync();
            }
            // evaluatePath (:1167-1201)
            rtk_wp_clear(E);
            uint32_t len_prefix = 0;
            for (uint32_t j = 0; j < i; ++j) { const UMap u = rtk_u(P.ums[j]); rtk_wp_extend(c, E, u); len_prefix += u.len; }
            for (uint32_t x = 0; x < nR; ++x) { const UMap u = rtk_u(R[x]); rtk_wp_extend(c, E, u); }
            for (uint32_t j = i + 1; j < P.n; ++j) { const UMap u = rtk_u(P.ums[j]); rtk_wp_extend(c, E, u); }
            if (rtk_failed(s)) break;
            const uint32_t qn = P.qlen;
            if (len_prefix > qn) { rtk_fail_ovf(s, 14); break; } // std::string::replace would throw in the reference: a path without qualities never gets here
            const uint32_t cut = (um_path.len + k - 1) < (qn - len_prefix) ? (um_path.len + k - 1) : (qn - len_prefix);
            const uint32_t new_len = qn - cut + rep_l;
            E.qlen = 0;
            if (new_len == E.l) { // Path::setQuality
                if (new_len > s.str_cap) { rtk_fail_ovf(s, 6); break; }
                rtk_wcopy(E.qual, P.qual, len_prefix);
                rtk_wfill(E.qual + len_prefix, q_max, rep_l);
                rtk_wcopy(E.qual + len_prefix + rep_l, P.qual + len_prefix + cut, qn - len_prefix - cut);
                E.qlen = new_len;
            }
            const uint32_t sl = rtk_ums_to_string(c, E.ums, E.n, s.str[0]); if (sl == 0xFFFFFFFFu) break;
            RTK_SITE(9); const int d = rtk_u(rtk_align(c, s.str[0], sl, ref, ref_len, ed, RTK_MODE_NW).dist);
            if (d >= 0 && d < ed) { ed = d; best_h = rtk_wp_commit(s, E, 1); }
        }
        if (rtk_failed(s)) break;
        if (best_h != ~0ull) { // a better aligning path: go on behind the inserted unitigs (:1283-1292)
            const uint32_t diff = rtk_rec_n(s, best_h) - P.n;
            rtk_wp_load(s, P, best_h);
            i += diff - 1;
        } else {
            while (i + 1 < P.n && rtk_u(P.ums[i + 1]).unitig == um_path.unitig) ++i;
        }
    }
    if (rtk_failed(s)) return ~0ull;
    return rtk_wp_commit(s, P, 1);
}

RTK_FN_SEARCH uint64_t rtk_explore_paths(const RCtx& c_, const uint32_t* all_pids_, uint32_t n_all_, const char* ref_, uint32_t ref_len_, const UMap& um_s_, const UMap& um_e_, bool has_end_) {
    const RCtx& c = *rtk_u(&c_); const uint32_t* all_pids = rtk_u(all_pids_); uint32_t n_all = rtk_u(n_all_); const char* ref = rtk_u(ref_); uint32_t ref_len = rtk_u(ref_len_); const UMap um_s = rtk_u(um_s_); const UMap um_e = rtk_u(um_e_); bool has_end = rtk_u(has_end_);
    RegionScratch& s = rtk_hdr(c);
    const uint32_t k = static_cast<uint32_t>(c.k);
    s.top[1] = 0; s.memo_n = 0;
    uint64_t* v = s.list[0]; uint64_t* v_tmp = s.list[1];
    uint32_t nv = 0, nvt = 0;
    const char q_max = rtk_get_qual(1.0, 0, static_cast<uint64_t>(c.o.max_qual));
    const bool ok_start = !rtk_um_is_empty(um_s) && ((c.g.flags[um_s.unitig] & RTK_F_EDGE_MASK) != 0);
    const bool ok_end = !has_end || (!rtk_um_is_empty(um_e) && ((c.g.flags[um_e.unitig] & RTK_F_EDGE_MASK) != 0));
    if (ok_start && ok_end) {
        const uint32_t level = 4;
        const bool lrc = c.o.long_read_correct != 0;
        const uint32_t max_len_subpath = static_cast<uint32_t>(static_cast<uint64_t>(static_cast<double>(c.k) * c.o.large_k_factor));
        uint64_t mn, mx; rtk_min_max_len(ref_len - k, c.o.weak_region_len_factor, &mn, &mx);
        const uint32_t min_len_path = static_cast<uint32_t>(mn) + k;
        const uint32_t max_len_path = static_cast<uint32_t>(mx > 10 ? mx : 10) + k;
        const uint32_t max_paths = 1024;
        WPath& w = s.wp[1];
        const UMap ust = rtk_start_suffix(c, um_s);
        if (has_end) {
            if (um_s.unitig == um_e.unitig && um_s.strand == um_e.strand && ust.dist <= um_e.dist) { // :340-358
                const uint32_t len = (ust.len + k - 1) - (um_e.strand ? (rtk_ulen(c.g, um_e.unitig) - um_e.dist - k) : um_e.dist);
                if (len >= min_len_path && len <= max_len_path) {
                    UMap bt = ust;
                    if (bt.strand) bt.len = um_e.dist - bt.dist + 1; else { bt.dist = um_e.dist; bt.len -= um_e.dist; }
                    rtk_wp_start(c, w, bt, q_max);
                    if (nv < s.list_cap) v[nv++] = rtk_wp_commit(s, w, 1); else rtk_fail_ovf(s, 8);
                }
            }
        } else if ((ust.len + k - 1) >= min_len_path) { // :127-140
            UMap back = ust;
            if ((back.len + k - 1) > max_len_path) { if (!back.strand) back.dist = back.len - (max_len_path - k + 1); back.len = max_len_path - k + 1; }
            rtk_wp_start(c, w, back, q_max);
            v[nv++] = rtk_wp_commit(s, w, 1);
        }
        rtk_wp_start(c, w, ust, q_max);
        uint64_t qh = rtk_wp_commit(s, w, 1); bool q_has = true; // the queue never holds more than one path (each pop pushes <= 1)
        // a queue entry P (+) Q whose non-terminal sub-path Q has not been given its score / quality string yet (see rtk_explore_subgraph)
        bool q_pending = false; uint64_t pend_hp = 0, pend_hq = 0; NtPending pend; pend.e = 0; pend.nt1 = 0.0; pend.nt2 = 0.0; pend.score_deferred = 0; pend.qual_deferred = 0;
        RTK_PL(s, 6);
        while (q_has && !rtk_failed(s)) {
            if (q_pending) { // the pop of src/GraphTraversal.cpp:364-366: only a path shorter than max_len_path is ever looked at again
                q_pending = false;
                const int lv = rtk_h_lvl(pend_hq); const uint64_t oo = rtk_h_off(pend_hq);
                const UMap* qu = rtk_path_ums(s, lv, oo); const uint32_t qn = rtk_rec_n(s, pend_hq);
                uint32_t l_ext = rtk_rec_l(s, pend_hp);
                for (uint32_t i = 0; i < qn; ++i) l_ext += rtk_u(qu[i].len); // Path::extend adds um.len per unitig (Path.hpp:319-330)
                if (!(l_ext < max_len_path)) break;
                WPath& wq = s.wp[2];
                rtk_wp_load(s, wq, pend_hq);
                const uint32_t sl = rtk_u(rtk_ums_to_string(c, rtk_ld(&wq.ums), rtk_ld(&wq.n), s.str[1]));
                if (sl == 0xFFFFFFFFu || sl > rtk_ld(&s.str_cap)) { rtk_fail_ovf(s, 7); break; }
                double nt1 = pend.nt1; const double nt2 = pend.nt2;
                if (pend.score_deferred) nt1 = rtk_u(rtk_score_path(c, sl, ref + pend.e, ref_len - pend.e, false));
                rtk_score_path_qual(c, sl, ref + pend.e, ref_len - pend.e, nt1, nt2, s.str[2]);
                if (sl == rtk_ld(&wq.l)) { rtk_wcopy(rtk_ld(&wq.qual), s.str[2], sl); wq.qlen = sl; } // Path::setQuality only accepts q.length() == l
                const uint64_t hq = rtk_wp_commit(s, wq, 1);
                if (rtk_failed(s)) break;
                rtk_wp_load(s, w, pend_hp); rtk_extend_by(c, w, hq, 0xFFFFFFFFu);
                qh = rtk_wp_commit(s, w, 1);
                if (rtk_failed(s)) break;
            }
            const uint64_t hp = qh; q_has = false;
            if (rtk_rec_l(s, hp) < max_len_path) {
                uint32_t n_t, n_nt;
                RTK_PL(s, 20);
                rtk_explore(c, all_pids, n_all, ref, ref_len, has_end ? um_e : rtk_um_empty(), hp, max_len_path, &n_t, &n_nt, &pend);
                if (rtk_failed(s)) break;
                if (has_end) {
                    for (uint32_t i = 0; i < n_t && !rtk_failed(s); ++i) {
                        rtk_wp_load(s, w, hp); rtk_extend_by(c, w, s.list[2][i], 0xFFFFFFFFu);
                        if (nvt >= s.list_cap) { rtk_fail_ovf(s, 8); break; }
                        v_tmp[nvt++] = rtk_wp_commit(s, w, 1);
                    }
                    for (uint32_t i = 0; i < n_nt && !rtk_failed(s); ++i) {
                        if (lrc ? (rtk_rec_l(s, s.list[3][i]) >= max_len_subpath) : (rtk_rec_n(s, s.list[3][i]) == level)) { // :395
                            if (pend.qual_deferred) { // keep what is needed to finish Q when (if) the entry is popped: its unitigs move to the BFS-level arena
                                rtk_wp_load(s, s.wp[2], s.list[3][i]);
                                pend_hq = rtk_wp_commit(s, s.wp[2], 1); pend_hp = hp; q_pending = true; q_has = true;
                            } else {
                                rtk_wp_load(s, w, hp); rtk_extend_by(c, w, s.list[3][i], 0xFFFFFFFFu);
                                qh = rtk_wp_commit(s, w, 1); q_has = true; // queue size 1 < 512: resizeQueue never fires
                            }
                        }
                    }
                    if (nvt >= max_paths) {
                        for (uint32_t i = 0; i < nvt && !rtk_failed(s); ++i) {
                            const uint32_t l = rtk_rec_l(s, v_tmp[i]);
                            if (l >= min_len_path && l <= max_len_path) { if (nv + 1 >= max_paths) rtk_resize_to_best(c, v, &nv, ref, ref_len); if (nv >= s.list_cap) { rtk_fail_ovf(s, 8); break; } v[nv++] = v_tmp[i]; }
                        }
                        nvt = 0;
                    }
                } else { // BFS without end anchor: every extension inside the length window is a candidate (:158-191)
                    for (uint32_t i = 0; i < n_nt && !rtk_failed(s); ++i) {
                        const uint64_t hs = s.list[3][i];
                        const uint32_t nsub = rtk_rec_n(s, hs);
                        for (uint32_t u = 1; u <= nsub && !rtk_failed(s); ++u) {
                            rtk_wp_load(s, w, hp); rtk_extend_by(c, w, hs, u);
                            if (w.l >= min_len_path && w.l <= max_len_path) { if (nvt >= s.list_cap) { rtk_fail_ovf(s, 8); break; } v_tmp[nvt++] = rtk_wp_commit(s, w, 1); }
                            if (u == nsub && (lrc ? (rtk_rec_l(s, hs) >= max_len_subpath) : (nsub == level))) { qh = rtk_wp_commit(s, w, 1); q_has = true; } // :174
                        }
                    }
                    if (nvt >= max_paths) {
                        for (uint32_t i = 0; i < nvt && !rtk_failed(s); ++i) {
                            rtk_wp_load(s, w, v_tmp[i]); rtk_wp_prune_prefix(c, w, max_len_path);
                            if (nv >= s.list_cap) { rtk_fail_ovf(s, 8); break; }
                            v[nv++] = rtk_wp_commit(s, w, 1);
                        }
                        nvt = 0;
                    }
                }
            }
        }
        if (!rtk_failed(s)) { // final flush
            if (has_end) {
                for (uint32_t i = 0; i < nvt && !rtk_failed(s); ++i) {
                    const uint32_t l = rtk_rec_l(s, v_tmp[i]);
                    if (l >= min_len_path && l <= max_len_path) { if (nv + 1 >= max_paths) rtk_resize_to_best(c, v, &nv, ref, ref_len); if (nv >= s.list_cap) { rtk_fail_ovf(s, 8); break; } v[nv++] = v_tmp[i]; }
                }
            } else {
                for (uint32_t i = 0; i < nvt && !rtk_failed(s); ++i) {
                    rtk_wp_load(s, w, v_tmp[i]); rtk_wp_prune_prefix(c, w, max_len_path);
                    if (nv >= s.list_cap) { rtk_fail_ovf(s, 8); break; }
                    v[nv++] = rtk_wp_commit(s, w, 1);
                }
            }
        }
    }
    RTK_PL(s, 20);
    if (rtk_failed(s) || nv == 0) return ~0ull;
    if (nv > 1) { int bid, bend; RTK_SITE(10); rtk_select_best(c, v, nv, ref, ref_len, RTK_MODE_NW, -1.0, &bid, &bend); if (rtk_failed(s)) return ~0ull; v[0] = v[bid]; }
    { const uint64_t r_ = rtk_path_has_short_cycle(c, v[0]) ? rtk_fix_repeats(c, v[0], ref, ref_len) : v[0]; RTK_PL(s, 21); return r_; }
}

// ------------------------------------------------------------------------------------------------ extractSemiWeakPaths (src/Correction.cpp:3-157)
// BFS results never hold more than one path, so `paths1` is a single running path (level 0). Dead ends are appended to
// `partial` (list[5]). Returns the complete path handle or ~0.
RTK_FN_SEARCH uint64_t rtk_extract_semi_weak(const RCtx& c_, const char* s_read_, uint32_t s_len_, const uint32_t* all_pids_, uint32_t n_all_, uint32_t start_pos_, const UMap& start_um_, uint32_t end_pos_in_, const UMap& end_um_, const Anchors& lvw_, uint32_t lvw_lo_, uint32_t lvw_hi_, uint32_t i_weak_, uint32_t* n_partial_) {
    const RCtx& c = *rtk_u(&c_); const char* s_read = rtk_u(s_read_); uint32_t s_len = rtk_u(s_len_); const uint32_t* all_pids = rtk_u(all_pids_); uint32_t n_all = rtk_u(n_all_); uint32_t start_pos = rtk_u(start_pos_); const UMap start_um = rtk_u(start_um_); uint32_t end_pos_in = rtk_u(end_pos_in_); const UMap end_um = rtk_u(end_um_); const Anchors& lvw = *rtk_u(&lvw_); uint32_t lvw_lo = rtk_u(lvw_lo_); uint32_t lvw_hi = rtk_u(lvw_hi_); uint32_t i_weak = rtk_u(i_weak_); uint32_t* n_partial = rtk_u(n_partial_);
    RegionScratch& s = rtk_hdr(c);
    const uint32_t k = static_cast<uint32_t>(c.k);
    const bool no_end = rtk_um_is_empty(end_um);
    const uint32_t pos2 = no_end ? s_len - k : end_pos_in;
    const uint32_t max_len_weak_region = c.o.long_read_correct ? c.o.max_len_weak_region2 : c.o.max_len_weak_region1; // :23
    uint32_t next_weak_pos = 0;
    bool begin = true, end = false;
    WPath& w0 = s.wp[0];
    rtk_wp_start(c, w0, start_um, rtk_get_qual(1.0, 0, static_cast<uint64_t>(c.o.max_qual)));
    uint64_t cur = rtk_wp_commit(s, w0, 0); uint32_t cur_pos = start_pos; bool have = true;
    const uint32_t nw = lvw_hi - lvw_lo; // weak anchors of the region are lvw[lvw_lo + i], i in [0, nw)
    if (i_weak < nw) i_weak = rtk_an_first_ge(lvw, lvw_lo + i_weak, lvw_lo + nw, start_pos) - lvw_lo; // the reference's forward walks over the weak anchors, as searches
    if (i_weak < nw) { const uint32_t wp = rtk_u(rtk_an_pos(lvw, lvw_lo + i_weak)); next_weak_pos = wp > start_pos + k ? wp : start_pos + k; }
    while (have && !end && !rtk_failed(s)) {
        if (i_weak < nw) { const uint64_t lim_a = static_cast<uint64_t>(pos2 - k), lim_b = next_weak_pos; i_weak = rtk_an_first_ge(lvw, lvw_lo + i_weak, lvw_lo + nw, lim_a < lim_b ? lim_a : lim_b) - lvw_lo; }
        else i_weak = nw;
        end = (i_weak == nw) || (static_cast<uint64_t>(rtk_u(rtk_an_pos(lvw, lvw_lo + i_weak))) >= static_cast<uint64_t>(pos2 - k));
        const uint32_t target_pos = end ? pos2 : rtk_u(rtk_an_pos(lvw, lvw_lo + i_weak));
        const uint32_t l_len = (target_pos - cur_pos) + k;
        const UMap um_start = begin ? start_um : rtk_rec_back(s, cur);
        uint64_t res = ~0ull; bool called = false;
        { // one call site for the three cases: to the end of the read (:61-72), to the right solid anchor (:74-78), to the next weak anchor (:110-114)
            UMap um_to = rtk_um_empty(); bool with_end = false;
            if (end) { if (no_end) called = l_len <= (max_len_weak_region / 2); else { called = l_len <= max_len_weak_region; um_to = end_um; with_end = true; } }
            else if (l_len <= max_len_weak_region) { called = true; um_to = rtk_u(rtk_an_um(lvw, lvw_lo + i_weak)); with_end = true; }
            RTK_PL(s, 5);
            if (called) res = rtk_explore_paths(c, all_pids, n_all, s_read + cur_pos, l_len, um_start, um_to, with_end);
        }
        if (rtk_failed(s)) break;
        if (called && res != ~0ull) {
            rtk_wp_load(s, w0, cur); rtk_wp_merge(c, w0, res);
            cur = rtk_wp_commit(s, w0, 0); cur_pos = target_pos;
        } else {
            if (*n_partial >= s.list_cap) { rtk_fail_ovf(s, 8); break; }
            s.list[5][(*n_partial)++] = cur; have = false;
        }
        if (!end) next_weak_pos = rtk_u(rtk_an_pos(lvw, lvw_lo + i_weak)) + k;
        begin = false;
    }
    RTK_PL(s, 22);
    return (have && !rtk_failed(s)) ? cur : ~0ull;
}

// ------------------------------------------------------------------------------------------------ chooseColors (src/Correction.cpp:215-429)
// anchors of the three sides are given as small (unitig, non-branching) lists, first insertion wins (unordered_map::insert).
RTK_DEV bool rtk_side_insert(SideList& l, uint32_t u, bool nonbranching) { // returns true when unseen
    const uint32_t n = rtk_u(l.n); const uint32_t* lu = rtk_u(l.u);
    for (uint32_t i0 = 0; i0 < n; i0 += RTK_WAVE) { // 64 entries per step
        const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
        if (rtk_ballot(i < n && lu[i] == u)) return false;
    }
    if (n < rtk_u(l.cap)) { l.u[n] = u; l.nb[n] = nonbranching ? 1 : 0; l.n = n + 1; rtk_sync(); }
    return true;
}

// set-buffer helpers on RegionScratch: buffers are addressed by index; sizes kept by the caller
RTK_DEV uint32_t rtk_rs_union(RegionScratch& s, int a, uint32_t na, const uint32_t* b, uint32_t nb, int out) {
    if (na + nb > s.set_cap) { rtk_fail_ovf(s, 9); return 0; }
    return rtk_set_union(s.set[a], na, b, nb, s.set[out], s.set[9]);
}

#include "rtk_colours.h"

// Computes all_pids into set[0]; returns its size. Uses set[1..9] as temporaries.
RTK_FN uint32_t rtk_choose_colors_general(const RCtx& c_, const SideList& side_s_, const SideList& side_e_, const SideList& side_w_);
// chooseColors: the two register / bit-vector programs of rtk_colours.h first (nearly every region), the general program below otherwise.
// Compiled into its caller: the dispatcher itself as a function would save 17 register rows on every region for a path it almost never takes.
RTK_DEV uint32_t rtk_choose_colors(const RCtx& c, const SideList& side_s, const SideList& side_e, const SideList& side_w) {
    RegionScratch& s = rtk_hdr(c);
    const unsigned long long tf = rtk_clock();
#ifndef RTK_SIM
    { const uint32_t r0 = rtk_u(rtk_choose_colors_small(c, side_s, side_e, side_w));
      if (r0 != RTK_NONE32) { const unsigned long long d_ = rtk_clock() - tf; s.fine[0] += d_; s.fine[12] += d_; s.fine[14] += 1; return rtk_failed(s) ? 0 : r0; } }
#endif
    const uint32_t r = rtk_u(rtk_choose_colors_bits(c, side_s, side_e, side_w));
    if (r != RTK_NONE32) { const unsigned long long d_ = rtk_clock() - tf; s.fine[0] += d_; s.fine[13] += d_; s.fine[15] += 1; return rtk_failed(s) ? 0 : r; }
    return rtk_u(rtk_choose_colors_general(c, side_s, side_e, side_w));
}
RTK_FN uint32_t rtk_choose_colors_general(const RCtx& c_, const SideList& side_s_, const SideList& side_e_, const SideList& side_w_) {
    const RCtx& c = *rtk_u(&c_); const SideList& side_s = *rtk_u(&side_s_); const SideList& side_e = *rtk_u(&side_e_); const SideList& side_w = *rtk_u(&side_w_); RTK_ASSUME_LDS(&side_s); RTK_ASSUME_LDS(&side_e); RTK_ASSUME_LDS(&side_w);
    RegionScratch& s = rtk_hdr(c);
    const GraphView& g = c.g;
    unsigned long long tf = rtk_clock();
    // a_pid[shift], shift = side index (0 middle, 1 right, 2 left) + 3 * nonbranching: built one after the other into the arena (level 2 is free here)
    s.top[2] = 0;
    tf = rtk_clock();
#define RTK_FINE_LAP(i) { const unsigned long long tn_ = rtk_clock(); s.fine[i] += tn_ - tf; tf = tn_; }
    const SideList* sides[3] = {&side_w, &side_e, &side_s};
    const uint32_t* a_ptr[6]; uint32_t a_n[6];
    for (int sh = 0; sh < 6 && !rtk_failed(s); ++sh) {
        const SideList& sl = *sides[sh % 3]; const uint8_t want_nb = sh >= 3 ? 1 : 0;
        int cur = 1; uint32_t n = 0, n_src = 0; const uint32_t* one = nullptr;
        for (uint32_t i = 0; i < sl.n && !rtk_failed(s); ++i) {
            if (sl.nb[i] != want_nb) continue;
            const uint32_t u = sl.u[i];
            const int32_t gi = g.gid[u]; // G2: only the global set when there is one
            const uint32_t* src = gi >= 0 ? g.col + g.goff[gi] : g.col + g.loff[u];
            const uint32_t ns = gi >= 0 ? static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi]) : static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]);
            s.cnt[1] += ns;
            if (ns == 0) continue;
            // a class fed by ONE anchor set (the usual case: a region is flanked by a unitig or two) is that set: used where it lies in
            // the graph's colour pool, neither merged nor copied
            if (n_src == 0) { one = src; n = ns; n_src = 1; continue; }
            if (n_src == 1) { if (n > s.set_cap) { rtk_fail_ovf(s, 9); break; } rtk_wcopy(s.set[cur], one, 4ull * n); rtk_sync(); }
            n = rtk_rs_union(s, cur, n, src, ns, cur ^ 3); cur ^= 3; ++n_src; // ping-pong between set[1] and set[2]
#ifdef RTK_SIM
            rtk_sim_site_stat[29][4] += 1;
#endif
        }
        a_n[sh] = n;
        if (n_src <= 1) a_ptr[sh] = n_src ? one : reinterpret_cast<const uint32_t*>(s.arena[2].get());
        else {
            const uint64_t off = rtk_arena_alloc(s, 2, 4ull * n + 4);
            if (!rtk_failed(s)) rtk_wcopy(s.arena[2] + off, s.set[cur], 4ull * n);
            a_ptr[sh] = reinterpret_cast<const uint32_t*>(s.arena[2] + off);
        }
    }
    if (rtk_failed(s)) return 0;
#ifdef RTK_SIM
    { std::atomic<unsigned long long>* t = rtk_sim_site_stat[29]; t[0] += 1; t[1] += side_s.n; t[2] += side_e.n; t[3] += side_w.n; for (int i = 0; i < 6; ++i) rtk_sim_site_stat[30][i] += a_n[i];
      unsigned long long tot = 0; for (int i = 0; i < 6; ++i) tot += a_n[i]; int b = 0; while (b < 7 && (256ull << b) <= tot) ++b; rtk_sim_site_stat[31][b] += 1; }
#endif
    RTK_FINE_LAP(0)
    auto A = [&](int i) -> const uint32_t* { return a_ptr[i]; };
    // candidate anchors: cardinality >= min_cov_vertices, ordered by (cardinality, unitig id) [D1]
    uint64_t* keys = s.list[4]; uint64_t* vals = s.list[3];
    uint32_t nsp = 0;
    for (int sd = 0; sd < 3; ++sd) for (uint32_t i = 0; i < sides[sd]->n; ++i) {
        const uint32_t u = sides[sd]->u[i];
        if (g.card[u] < c.o.min_cov_vertices) continue;
        bool dup = false; for (uint32_t j0 = 0; j0 < nsp && !dup; j0 += RTK_WAVE) { const uint32_t j = j0 + static_cast<uint32_t>(rtk_lane()); dup = rtk_ballot(j < nsp && rtk_d1_unitig(keys[j], c.o.d1_desc) == u) != 0ull; }
        if (dup) continue;
        if (2 * (nsp + 1) > s.list_cap) { rtk_fail_ovf(s, 8); return 0; }
        keys[nsp] = rtk_d1_key(g.card[u], u, c.o.d1_desc); vals[nsp] = 0; ++nsp; rtk_sync();
    }
    rtk_sort_pairs(keys, vals, nsp);
    RTK_FINE_LAP(1)
    const uint32_t cov = 30;
    for (uint32_t j = 0; j < nsp; ++j) { const uint32_t cd = static_cast<uint32_t>(keys[j] >> 32); vals[j] = cd < cov ? cd : cov; } // remaining quota (p_spid.second)
    // Set expressions of src/Correction.cpp:233-275,300-352 on immutable operands: a result is either one of its operands (union with
    // / difference by the empty set -- the usual case: most regions have no weak anchor, so the whole "middle" side is empty) or a
    // fresh slice of the DFS-level arena; nothing is copied to be kept, and a class only computes what it reads.
    struct SetRef { const uint32_t* p; uint32_t n; };
    const SetRef EMPTY = { reinterpret_cast<const uint32_t*>(s.arena[2].get()), 0u };
    auto alloc = [&](uint32_t n) -> uint32_t* { const uint64_t off = rtk_arena_alloc(s, 2, 4ull * n + 4); return rtk_failed(s) ? nullptr : reinterpret_cast<uint32_t*>(s.arena[2] + off); };
    auto Un = [&](SetRef a, SetRef b) -> SetRef {
        if (!a.n) return b; if (!b.n) return a;
        if (a.n + b.n > s.set_cap) { rtk_fail_ovf(s, 9); return EMPTY; } // set[9] holds b \ a
        uint32_t* o = alloc(a.n + b.n); if (!o) return EMPTY;
        SetRef r; r.p = o; r.n = rtk_set_union(a.p, a.n, b.p, b.n, o, s.set[9]); return r; };
    auto In = [&](SetRef a, SetRef b) -> SetRef {
        if (!a.n || !b.n) return EMPTY;
        if (a.n > b.n) { const SetRef t = a; a = b; b = t; } // walk the smaller set, search the larger one
        uint32_t* o = alloc(a.n); if (!o) return EMPTY;
        SetRef r; r.p = o; r.n = rtk_set_inter(a.p, a.n, b.p, b.n, o); return r; };
    auto Di = [&](SetRef a, SetRef b) -> SetRef {
        if (!a.n) return EMPTY; if (!b.n) return a;
        uint32_t* o = alloc(a.n); if (!o) return EMPTY;
        SetRef r; r.p = o; r.n = rtk_set_diff(a.p, a.n, b.p, b.n, o); return r; };
    SetRef a[6]; for (int i = 0; i < 6; ++i) { a[i].p = a_ptr[i]; a[i].n = a_n[i]; }
    const SetRef pos0 = Un(a[0], a[3]), pos1 = Un(a[1], a[4]), pos2 = Un(a[2], a[5]);
    const SetRef a01 = In(pos0, pos1), a12 = In(pos1, pos2), a02 = In(pos0, pos2);
    const SetRef nobranch_all = Un(Un(a[3], a[4]), a[5]);
    if (rtk_failed(s)) return 0;
    RTK_FINE_LAP(2)
    uint32_t n_all = 0; int allb = 0; // all_pids lives in set[0] (while it is being built: in set[allb])
    uint32_t nb_unselected = nsp;
    SetRef nobranch = nobranch_all, branching = EMPTY, i3 = EMPTY, i2 = EMPTY, prev2 = EMPTY; // prev2: a_pid2 of the previous class
    bool have_i3 = false, have_i2 = false;
    for (int i = 5; i >= 0 && !rtk_failed(s); --i) {
        if (nb_unselected == 0) break;
        if ((i == 5 || i == 2) && !have_i3) { i3 = In(a01, a12); have_i3 = true; }
        if ((i == 4 || i == 1) && !have_i2) { i2 = Un(Un(a01, a12), a02); have_i2 = true; }
        SetRef a2 = EMPTY; // a_pid2[i]
        if (i == 5) a2 = In(nobranch, i3);
        else if (i == 4) { nobranch = Di(nobranch, prev2); a2 = In(nobranch, i2); }
        else if (i == 3) { nobranch = Di(nobranch, prev2); a2 = nobranch; }
        else if (i == 2) { branching = Di(Un(Un(a[0], a[1]), a[2]), nobranch_all); a2 = In(branching, i3); }
        else if (i == 1) { branching = Di(branching, prev2); a2 = In(branching, i2); }
        else { branching = Di(branching, prev2); a2 = branching; }
        const uint32_t n2 = a2.n;
        if (rtk_failed(s)) break;
        prev2 = a2; // a_pid2[i] is needed by the next class
        RTK_FINE_LAP(3)
        if (n2 != 0) {
            nb_unselected = 0;
            const uint32_t* cur_p = a2.p; uint32_t ncur = n2; int curb = 8; // curr_pid: a_pid2[i] itself until the first selection, then set[7] / set[8] (ping-pong)
            for (uint32_t j = 0; j < nsp && !rtk_failed(s); ++j) {
                const uint32_t u = rtk_d1_unitig(keys[j], c.o.d1_desc);
                int quota = static_cast<int>(vals[j]);
                bool touch = false;
                if (quota > 0) { touch = (i == 0 || rtk_shared_with_set(g, u, cur_p, ncur, 1) >= 1); RTK_FINE_LAP(4) }
#ifdef RTK_SIM
                rtk_sim_site_stat[29][5] += 1; if (touch) rtk_sim_site_stat[29][6] += 1;
#endif
                if (touch) {
                    const uint32_t min_cov = g.card[u] < cov ? g.card[u] : cov;
                    const uint32_t sh = rtk_shared_with_set(g, u, s.set[allb], n_all, min_cov);
                    RTK_FINE_LAP(5)
                    quota = static_cast<int>(min_cov - (sh < min_cov ? sh : min_cov));
                    if (quota > 0) {
                        const uint32_t all_card = n_all;
                        // pid = (global & curr) | (local & curr), truncated to its `quota` lowest ids
                        const uint32_t npid = rtk_first_shared(g, u, cur_p, ncur, static_cast<uint32_t>(quota), s.set[4]);
                        if (n_all + npid > s.set_cap) { rtk_fail_ovf(s, 9); break; }
                        const uint32_t nn = rtk_set_union(s.set[allb], n_all, s.set[4], npid, s.set[allb ^ 3], s.set[9]);
                        allb ^= 3; n_all = nn; // all_pids alternates between set[0] and set[3]; it is moved to set[0] once, at the end
                        const int nb2 = curb == 8 ? 7 : 8;
                        if (ncur > s.set_cap) { rtk_fail_ovf(s, 9); break; }
                        ncur = rtk_set_diff(cur_p, ncur, s.set[4], npid, s.set[nb2]); curb = nb2; cur_p = s.set[nb2];
#ifdef RTK_SIM
                        rtk_sim_site_stat[29][7] += 1; rtk_sim_site_stat[30][6] += ncur; rtk_sim_site_stat[30][7] += n_all;
#endif
                        const int gained = static_cast<int>(n_all - all_card);
                        quota -= gained < quota ? gained : quota;
                        RTK_FINE_LAP(6)
                    }
                }
                vals[j] = static_cast<uint64_t>(quota);
                nb_unselected += quota > 0 ? 1u : 0u;
            }
        }
    }
    if (allb != 0 && !rtk_failed(s)) rtk_wcopy(s.set[0], s.set[3], 4ull * n_all);
    return rtk_failed(s) ? 0 : n_all;
}

// ------------------------------------------------------------------------------------------------ ResultCorrection (src/ResultCorrection.hpp)

// position bitmaps (ResultCorrection's Roaring set of corrected old positions): word-wise, one word per lane
RTK_DEV void rtk_bm_add_range(uint64_t* bm, uint32_t a, uint32_t b) { // [a, b)
    if (b <= a) return;
    const uint32_t w0 = a >> 6, w1 = (b - 1) >> 6;
    for (uint32_t w = w0 + static_cast<uint32_t>(rtk_lane()); w <= w1; w += RTK_WAVE) {
        const uint32_t lo = (w == w0) ? (a & 63u) : 0u, hi = (w == w1) ? ((b - 1) & 63u) : 63u;
        const uint64_t mask = ((hi == 63u) ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
        bm[w] |= mask;
    }
    rtk_sync();
}
RTK_DEV uint32_t rtk_bm_card(const uint64_t* bm, uint32_t n) {
    int c = 0;
    for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < (n + 63) / 64; w += RTK_WAVE) c += rtk_popc(bm[w]);
    return static_cast<uint32_t>(rtk_u(rtk_wave_sum(c)));
}
RTK_DEV bool rtk_bm_get(const uint64_t* bm, uint32_t i) { return (bm[i >> 6] >> (i & 63)) & 1ull; }
// first position >= p (capped at n) whose bit equals `want`
RTK_DEV uint32_t rtk_bm_next(const uint64_t* bm, uint32_t n, uint32_t p, bool want) {
    if (p >= n) return n;
    const uint32_t words = (n + 63) / 64;
    for (uint32_t w0 = p >> 6; w0 < words; w0 += RTK_WAVE) {
        const uint32_t w = w0 + static_cast<uint32_t>(rtk_lane());
        uint64_t x = 0;
        if (w < words) { x = want ? bm[w] : ~bm[w]; if (w == (p >> 6)) x &= ~((1ull << (p & 63u)) - 1ull); }
        const uint64_t bal = rtk_ballot(x != 0);
        if (bal) {
            const int l = rtk_ffs(bal) - 1;
            const uint32_t pos = 64u * (w0 + static_cast<uint32_t>(l)) + static_cast<uint32_t>(rtk_ffs(rtk_shfl(x, l)) - 1);
            return rtk_u(pos < n ? pos : n);
        }
    }
    return n;
}
RTK_DEV uint32_t rtk_rc_len_corrected(const ResCorr& r, uint32_t p) { return rtk_bm_next(r.bm, r.old_len, p, false) - (p < r.old_len ? p : r.old_len); } // :117-128
RTK_DEV uint32_t rtk_rc_len_uncorrected(const ResCorr& r, uint32_t p) { return rtk_bm_next(r.bm, r.old_len, p, true) - (p < r.old_len ? p : r.old_len); } // :130-142
// 64 bits of the bitmap starting at bit `lo` (may be negative; bits outside the words read as 0)
RTK_DEV uint64_t rtk_bm_window(const uint64_t* bm, uint32_t words, int64_t lo) {
    if (lo <= -64) return 0ull;
    if (lo < 0) return words ? (bm[0] << static_cast<uint32_t>(-lo)) : 0ull;
    const uint32_t w = static_cast<uint32_t>(lo >> 6), sh = static_cast<uint32_t>(lo & 63);
    uint64_t x = 0;
    if (w < words) x = bm[w] >> sh;
    if (sh && w + 1 < words) x |= bm[w + 1] << (64u - sh);
    return x;
}

RTK_FN void rtk_rc_reverse_complement(RegionScratch& s_, ResCorr& r_, uint64_t* tmp_bm_, char* tmp_) {
    RegionScratch& s = *rtk_u(&s_); RTK_ASSUME_LDS(&s); ResCorr& r = *rtk_u(&r_); uint64_t* tmp_bm = rtk_u(tmp_bm_); char* tmp = rtk_u(tmp_); // :72-88
    if (r.seq_len == 0) return;
    const uint32_t words = (r.old_len + 63) / 64;
    // new bit j = old bit old_len-1-j: output word ow is the bit reversal of the 64 old bits ending at old_len-1-64*ow
    for (uint32_t ow = static_cast<uint32_t>(rtk_lane()); ow < words; ow += RTK_WAVE)
        tmp_bm[ow] = rtk_brev64(rtk_bm_window(r.bm, words, static_cast<int64_t>(r.old_len) - 1 - 64ll * ow - 63));
    rtk_sync();
    for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < words; w += RTK_WAVE) r.bm[w] = tmp_bm[w];
    rtk_sync();
    for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < r.seq_len; i += RTK_WAVE) tmp[i] = rtk_comp(r.seq[r.seq_len - 1 - i]);
    rtk_sync(); rtk_wcopy(r.seq, tmp, r.seq_len);
    for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < r.qual_len; i += RTK_WAVE) tmp[i] = r.qual[r.qual_len - 1 - i];
    rtk_sync(); rtk_wcopy(r.qual, tmp, r.qual_len);
    (void)s;
}

// appenders for the growing corrected strings
RTK_FN_LEAF void rtk_app(RegionScratch& s_, char* dst_, uint32_t* len_, const char* src_, uint32_t n_) {
    RegionScratch& s = *rtk_u(&s_); RTK_ASSUME_LDS(&s); char* dst = rtk_u(dst_); uint32_t* len = rtk_u(len_); const char* src = rtk_u(src_); uint32_t n = rtk_u(n_); if (*len + n > s.str_cap) { rtk_fail_ovf(s, 7); return; } rtk_wcopy(dst + *len, src, n); *len += n; }
RTK_FN_LEAF void rtk_app_fill(RegionScratch& s_, char* dst_, uint32_t* len_, char ch_, uint32_t n_) {
    RegionScratch& s = *rtk_u(&s_); RTK_ASSUME_LDS(&s); char* dst = rtk_u(dst_); uint32_t* len = rtk_u(len_); char ch = rtk_u(ch_); uint32_t n = rtk_u(n_); if (*len + n > s.str_cap) { rtk_fail_ovf(s, 7); return; } rtk_wfill(dst + *len, ch, n); *len += n; }

// Bifrost Kmer(const char*) 2-bit code of any character (end k-mer test, src/Correction.cpp:720-724)
RTK_DEV int rtk_bifrost_code(char ch) { const int x = (ch & 4) >> 1; return x + ((x ^ (ch & 2)) >> 1); }


// Visits anchors x = start, start+step, ... while `in_range(pos)` holds (positions are sorted, so the condition is a prefix
// property), calling fn(um) once per RUN of consecutive anchors on the same unitig. Repeated visits of one unitig are no-ops for
// the side lists (first insertion wins, the branching quota only grows), so skipping them is exact. Lanes fetch 64 anchors at a time.
template <class Cond, class Fn>
RTK_DEV void rtk_scan_anchor_runs(const Anchors& a, int64_t start, int step, Cond in_range, Fn fn) {
    uint32_t prev_unitig = RTK_NONE32; bool first = true;
    for (int64_t b = 0;; b += RTK_WAVE) {
        const int64_t x = start + static_cast<int64_t>(step) * (b + rtk_lane());
        bool ok = false; UMap um = rtk_um_empty();
        if (x >= 0 && x < static_cast<int64_t>(a.n)) { ok = in_range(rtk_an_pos(a, static_cast<uint32_t>(x))); if (ok) um = rtk_an_um(a, static_cast<uint32_t>(x)); }
        const uint64_t okm = rtk_ballot(ok);
        const int lead = (~okm == 0ull) ? RTK_WAVE : (rtk_ffs(~okm) - 1); // anchors of this chunk that are visited
        if (lead == 0) break;
        uint32_t left_unitig = rtk_shfl_up1(um.unitig, prev_unitig);
        const bool run_start = ok && rtk_lane() < lead && (um.unitig != left_unitig || (first && rtk_lane() == 0));
        uint64_t rs = rtk_ballot(run_start);
        while (rs) {
            const int l = rtk_ffs(rs) - 1; rs &= rs - 1ull;
            UMap u; u.unitig = rtk_shfl(um.unitig, l); u.dist = rtk_shfl(um.dist, l); u.len = 1; u.strand = rtk_shfl(um.strand, l);
            fn(u);
        }
        prev_unitig = rtk_shfl(um.unitig, lead - 1); first = false;
        if (lead < RTK_WAVE) break;
    }
}

// ------------------------------------------------------------------------------------------------ the `correct` lambda (src/Correction.cpp:431-753)
// s_read: read in the orientation of this call; v_s / v_w: anchors in that orientation. Result strings go to res.seq / res.qual.
// q_read: pass 2 only, the quality string that goes with s_read in this call (the reference passes q_fw, q_bw or -- for the head region --
// q_fw next to the reverse-complemented read, :787 G17); uncorrected stretches keep their qualities instead of getting q_min.
#ifdef RTK_REGION_SPLIT
#include "variants/rtk_region_split.h" // (measured and rejected: the call in three programs; not part of the default build)
#endif

RTK_FN_REGION void rtk_correct_region(const RCtx& c_, const char* s_read_, uint32_t s_len_, const Anchors& v_s_, const Anchors& v_w_, uint32_t i_s_, uint32_t i_w_, const ResCorr* rc_, ResCorr& res_, const char* q_read_ = nullptr) {
    const RCtx& c = *rtk_u(&c_); const char* s_read = rtk_u(s_read_); const char* q_read = rtk_u(q_read_);
    const bool lrc = rtk_u(c.o.long_read_correct) != 0 && q_read != nullptr; uint32_t s_len = rtk_u(s_len_); const Anchors& v_s = *rtk_u(&v_s_); const Anchors& v_w = *rtk_u(&v_w_); RTK_ASSUME_LDS(&v_s); RTK_ASSUME_LDS(&v_w); uint32_t i_s = rtk_u(i_s_); uint32_t i_w = rtk_u(i_w_); const ResCorr* rc = rtk_u(rc_); ResCorr& res = *rtk_u(&res_); RTK_ASSUME_LDS(&res);
    RegionScratch& s = rtk_hdr(c);
    const uint32_t k = static_cast<uint32_t>(c.k);
    const GraphView& g = c.g;
    const bool has_end_pt = (i_s + 1) < v_s.n;
    // (values that live across the calls below are made scalar on purpose: a wave-uniform value in a vector register costs a 256-byte row
    // every time it is saved around a call, in a scalar register it is one lane of a spill register)
    uint32_t p1 = rtk_u(rtk_an_pos(v_s, i_s)); UMap um1 = rtk_u(rtk_an_um(v_s, i_s));
    const uint32_t p2 = rtk_u(has_end_pt ? rtk_an_pos(v_s, i_s + 1) : (s_len - k));
    const UMap um2 = rtk_u(has_end_pt ? rtk_an_um(v_s, i_s + 1) : rtk_um_empty());
    const uint32_t first_pos = p1;
    uint32_t len_weak_region = p2 - p1 + k;
    const uint64_t u_min_start = static_cast<uint64_t>(p1) - static_cast<uint64_t>(c.o.insert_sz); // wraps below insert_sz (G1)
    const uint64_t u_min_end = static_cast<uint64_t>(p2) + static_cast<uint64_t>(c.o.insert_sz);
    res.old_len = len_weak_region; res.is_corrected = false; res.seq_len = 0; res.qual_len = 0; res.n_all = 0;
    if ((len_weak_region + 63) / 64 + 1 > s.bm_words) { rtk_fail_ovf(s, 10); return; }
    for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < (len_weak_region + 63) / 64 + 1; w += RTK_WAVE) res.bm[w] = 0;
    rtk_sync();
    const char q_min = rtk_get_qual(0.0, 0, static_cast<uint64_t>(c.o.max_qual));
    const uint32_t max_len_weak_anchors = c.o.long_read_correct ? c.o.max_len_weak_region2 : c.o.max_len_weak_region1; // :177
    // weak anchors inside the region: l_v_w = v_w[lw_lo .. lw_hi)
    uint32_t lw_lo = 0, lw_hi = 0;
    {
        const uint32_t v_w_sz = v_w.n;
        if (v_w_sz) {
            const uint32_t pos_end = has_end_pt ? p2 : s_len;
            const uint32_t x = i_w - (((i_w != 0) && (i_w >= v_w_sz)) ? 1u : 0u);
            lw_lo = rtk_u(rtk_an_first_ge(v_w, x, v_w_sz, first_pos));
            lw_hi = rtk_u(rtk_an_first_ge(v_w, lw_lo, v_w_sz, pos_end));
        }
    }
#ifdef RTK_REGION_SPLIT
    {
        RegionCall& st = s.loc.call;
        st.s_read = s_read; st.q_read = q_read; st.s_len = s_len; st.p1 = p1; st.p2 = p2; st.um1 = um1; st.um2 = um2; st.first_pos = first_pos; st.len_weak_region = len_weak_region;
        st.lw_lo = lw_lo; st.lw_hi = lw_hi; st.has_end_pt = has_end_pt ? 1u : 0u; st.lrc = lrc ? 1u : 0u; (void)u_min_start; (void)u_min_end; (void)q_min; (void)max_len_weak_anchors; (void)g;
        uint32_t n_all_ = 0;
        if (rc == nullptr) { n_all_ = rtk_u(rtk_region_colours(c, v_s, v_w, i_s, i_w)); if (rtk_failed(s)) return; } else n_all_ = rtk_u(rc->n_all);
        st.n_all = n_all_; res.n_all = n_all_;
        rtk_region_search(c, v_w, res);
        if (rtk_failed(s)) return;
        rtk_region_assemble(c, res);
    }
}
#else
    uint32_t n_all = 0;
    RTK_PL(s, 2);
    if (rc == nullptr) {
        const unsigned long long t_side0 = rtk_clock();
        // side lists live in list[0..2] memory (u32 unitig + flag bytes)
        SideList& sl = s.loc.side[0]; SideList& sr = s.loc.side[1]; SideList& sm = s.loc.side[2];
        const uint32_t cap = s.list_cap;
        sl.u = reinterpret_cast<uint32_t*>(s.list[0].get()); sl.nb = reinterpret_cast<uint8_t*>(s.list[0].get() + cap / 2); sl.n = 0; sl.cap = cap;
        sr.u = reinterpret_cast<uint32_t*>(s.list[1].get()); sr.nb = reinterpret_cast<uint8_t*>(s.list[1].get() + cap / 2); sr.n = 0; sr.cap = cap;
        sm.u = reinterpret_cast<uint32_t*>(s.list[2].get()); sm.nb = reinterpret_cast<uint8_t*>(s.list[2].get() + cap / 2); sm.n = 0; sm.cap = cap;
        auto consider = [&](SideList& m, const UMap& um, uint32_t& nb_branching) {
            const uint32_t u = um.unitig; const bool br = rtk_is_branching(g, u);
            if (g.kcov[u] < c.o.max_km_cov && (!br || nb_branching < 5)) { const bool unseen = rtk_side_insert(m, u, !br); nb_branching += (unseen && br) ? 1u : 0u; }
        };
        { // left (:476-516)
            uint32_t nbb = 0;
            rtk_scan_anchor_runs(v_s, static_cast<int64_t>(i_s), -1, [&](uint32_t p) { return static_cast<uint64_t>(p) > u_min_start; }, [&](const UMap& um) { consider(sl, um, nbb); });
            const uint32_t v_w_sz = v_w.n;
            if (v_w_sz) {
                const uint32_t x0 = i_w - (((i_w != 0) && (i_w >= v_w_sz)) ? 1u : 0u);
                // the reference walks back while pos > u_min_start and index > 0: it lands on the last anchor at or below u_min_start (or on 0)
                const uint32_t f = rtk_an_first_gt(v_w, 0, x0 + 1, u_min_start);
                const uint32_t x = f > 0 ? f - 1 : 0;
                rtk_scan_anchor_runs(v_w, static_cast<int64_t>(x), +1, [&](uint32_t p) { return p < first_pos; }, [&](const UMap& um) { consider(sl, um, nbb); });
            }
        }
        if (has_end_pt) { // right (:518-561)
            uint32_t nbb = 0;
            rtk_scan_anchor_runs(v_s, static_cast<int64_t>(i_s) + 1, +1, [&](uint32_t p) { return static_cast<uint64_t>(p) < u_min_end; }, [&](const UMap& um) { consider(sr, um, nbb); });
            const uint32_t v_w_sz = v_w.n;
            if (v_w_sz) {
                const uint32_t x0 = i_w - (((i_w != 0) && (i_w >= v_w_sz)) ? 1u : 0u);
                const uint32_t x = rtk_an_first_ge(v_w, x0, v_w_sz, p2);
                rtk_scan_anchor_runs(v_w, static_cast<int64_t>(x), +1, [&](uint32_t p) { return static_cast<uint64_t>(p) < u_min_end; }, [&](const UMap& um) { consider(sr, um, nbb); });
            }
        }
        if (lw_hi > lw_lo) { // middle (:563-585)
            const uint32_t pos_end_m = has_end_pt ? p2 : s_len;
            rtk_scan_anchor_runs(v_w, static_cast<int64_t>(lw_lo), +1, [&](uint32_t p) { return p < pos_end_m; }, [&](const UMap& um) { const uint32_t u = um.unitig; if (g.kcov[u] < c.o.max_km_cov) rtk_side_insert(sm, u, !rtk_is_branching(g, u)); });
        }
        if (sl.n >= cap / 2 || sr.n >= cap / 2 || sm.n >= cap / 2) { rtk_fail_ovf(s, 8); return; }
        s.fine[7] += rtk_clock() - t_side0;
        RTK_PL(s, 3);
        { const unsigned long long t0 = rtk_clock(); n_all = rtk_u(rtk_choose_colors(c, sl, sr, sm)); s.cnt[5] += rtk_clock() - t0; }
        RTK_PL(s, 4);
        if (rtk_failed(s)) return;
        // keep all_pids for the reverse-complement call (rc = &fw): set[0] is preserved by everything below
    } else n_all = rtk_u(rc->n_all);
    res.n_all = n_all;
    const uint32_t* all_pids = s.set[0];
    // ---- paths ----
    s.top[0] = 0;
    uint32_t n_partial = 0, n_amb = 0; // n_amb: size of v_ambiguity (list[RTK_L_AMB])
    uint64_t complete = ~0ull;
    char* s_corr = res.seq; char* q_corr = res.qual; uint32_t& sl_ = s.loc.len[4]; uint32_t& ql_ = s.loc.len[5]; sl_ = 0; ql_ = 0; // (lengths that rtk_app updates through a pointer: LDS words)
    const Anchors& lvw = v_w;
    const uint32_t nlw = lw_hi - lw_lo;
    auto clamp_len = [&](uint32_t pos, uint32_t len) -> uint32_t { return (pos + len <= s_len) ? len : (pos < s_len ? s_len - pos : 0); }; // std::string::substr
    auto add_uncorrected = [&](uint32_t pos, uint32_t len, char q) { rtk_app(s, s_corr, &sl_, s_read + pos, clamp_len(pos, len));
        if (lrc) rtk_app(s, q_corr, &ql_, q_read + pos, clamp_len(pos, len)); else rtk_app_fill(s, q_corr, &ql_, q, len_weak_region); }; // :459-469
    // extractSemiWeakPaths from the left solid anchor (:613), then again from a weak anchor behind the best partial path as long as
    // there is one (:619-651): ONE call site, so that the whole search can be compiled into this function
    bool first_call = true, found_first = false, do_call = n_all >= c.o.min_cov_vertices;
    uint32_t i_w_s = 0;
    for (;;) {
        if (do_call) { const unsigned long long t0 = rtk_clock(); complete = rtk_u(rtk_extract_semi_weak(c, s_read, s_len, all_pids, n_all, p1, um1, p2, um2, lvw, lw_lo, lw_hi, first_call ? 0u : i_w_s, &n_partial)); n_partial = rtk_u(n_partial); s.cnt[6] += rtk_clock() - t0; }
        if (rtk_failed(s)) return;
        RTK_PL(s, 22);
        if (first_call && complete != ~0ull) found_first = true;
        first_call = false;
        if (!(complete == ~0ull && n_partial != 0 && nlw != 0 && n_all >= c.o.min_cov_vertices)) break;
        { // :619-651
            int& aid = s.loc.best[0]; int& aend = s.loc.best[1];
            RTK_SITE(11); rtk_select_best(c, s.list[5], n_partial, s_read + p1, len_weak_region, RTK_MODE_SHW, c.o.weak_region_len_factor, &aid, &aend);
            if (rtk_failed(s) || aid == -1) break;
            {
                const uint32_t next_pos = p1 + static_cast<uint32_t>(aend) + k;
                while (i_w_s < nlw && rtk_an_pos(lvw, lw_lo + i_w_s) < next_pos) ++i_w_s;
                if (i_w_s >= nlw || static_cast<uint64_t>(rtk_an_pos(lvw, lw_lo + i_w_s)) >= static_cast<uint64_t>(p2) - k || (rtk_an_pos(lvw, lw_lo + i_w_s) - p1) >= max_len_weak_anchors) break;
            }
            const uint64_t hb = rtk_u(s.list[5][aid]);
            const uint32_t wpos = rtk_u(rtk_an_pos(lvw, lw_lo + i_w_s));
            const uint32_t pl = rtk_rec_to_string(c, hb, s.str[0]); if (pl == 0xFFFFFFFFu) break;
            n_amb = rtk_amb_collect(c, hb, sl_, n_amb);
            rtk_app(s, s_corr, &sl_, s.str[0], pl);
            rtk_app(s, s_corr, &sl_, s_read + p1 + aend + 1, wpos - p1 - static_cast<uint32_t>(aend) - 1);
            rtk_app(s, q_corr, &ql_, rtk_path_qual(s, rtk_h_lvl(hb), rtk_h_off(hb)), rtk_path_hdr(s, rtk_h_lvl(hb), rtk_h_off(hb))->qlen);
            if (lrc) rtk_app(s, q_corr, &ql_, q_read + p1 + aend + 1, clamp_len(p1 + static_cast<uint32_t>(aend) + 1, wpos - p1 - static_cast<uint32_t>(aend) - 1)); // :642
            else rtk_app_fill(s, q_corr, &ql_, q_min, wpos - p1 - static_cast<uint32_t>(aend) - 1);
            rtk_bm_add_range(res.bm, p1 - first_pos, p1 + static_cast<uint32_t>(aend) + 1 - first_pos);
            p1 = wpos; um1 = rtk_u(rtk_an_um(lvw, lw_lo + i_w_s));
            len_weak_region = p2 - p1 + k;
            s.top[0] = 0; n_partial = 0; // paths of the previous attempt are dead
            do_call = true;
        }
    }
    if (rtk_failed(s)) return;
    RTK_PL(s, 23);
    if (!found_first) {
        if (complete != ~0ull) {
            const uint32_t pl = rtk_rec_to_string(c, complete, s.str[0]); if (pl == 0xFFFFFFFFu) return;
            n_amb = rtk_amb_collect(c, complete, sl_, n_amb);
            rtk_app(s, s_corr, &sl_, s.str[0], pl);
            rtk_app(s, q_corr, &ql_, rtk_path_qual(s, rtk_h_lvl(complete), rtk_h_off(complete)), rtk_path_hdr(s, rtk_h_lvl(complete), rtk_h_off(complete))->qlen);
            rtk_bm_add_range(res.bm, p1 - first_pos, p2 - first_pos + k);
        } else if (n_partial != 0) {
            int& aid = s.loc.best[0]; int& aend = s.loc.best[1];
            RTK_SITE(12); rtk_select_best(c, s.list[5], n_partial, s_read + p1, len_weak_region, RTK_MODE_SHW, c.o.weak_region_len_factor, &aid, &aend);
            if (rtk_failed(s)) return;
            if (aid == -1) add_uncorrected(p1, len_weak_region, q_min);
            else {
                const uint64_t hb = s.list[5][aid];
                const uint32_t pl = rtk_rec_to_string(c, hb, s.str[0]); if (pl == 0xFFFFFFFFu) return;
                n_amb = rtk_amb_collect(c, hb, sl_, n_amb);
                rtk_app(s, s_corr, &sl_, s.str[0], pl);
                const uint32_t rest = len_weak_region - static_cast<uint32_t>(aend) - 1;
                rtk_app(s, s_corr, &sl_, s_read + p1 + aend + 1, rest);
                rtk_app(s, q_corr, &ql_, rtk_path_qual(s, rtk_h_lvl(hb), rtk_h_off(hb)), rtk_path_hdr(s, rtk_h_lvl(hb), rtk_h_off(hb))->qlen);
                if (lrc) rtk_app(s, q_corr, &ql_, q_read + p1 + aend + 1, clamp_len(p1 + static_cast<uint32_t>(aend) + 1, rest)); // :684
                else rtk_app_fill(s, q_corr, &ql_, q_min, rest);
                rtk_bm_add_range(res.bm, p1 - first_pos, p1 + static_cast<uint32_t>(aend) + 1 - first_pos);
            }
        } else if (sl_ != 0) add_uncorrected(p1, len_weak_region, q_min);
        else { sl_ = 0; ql_ = 0; add_uncorrected(first_pos, len_weak_region, q_min); } // setUncorrected
    } else {
        const uint32_t pl = rtk_rec_to_string(c, complete, s.str[0]); if (pl == 0xFFFFFFFFu) return;
        sl_ = 0; ql_ = 0;
        n_amb = rtk_amb_collect(c, complete, 0, n_amb);
        rtk_app(s, s_corr, &sl_, s.str[0], pl);
        rtk_app(s, q_corr, &ql_, rtk_path_qual(s, rtk_h_lvl(complete), rtk_h_off(complete)), rtk_path_hdr(s, rtk_h_lvl(complete), rtk_h_off(complete))->qlen);
        rtk_bm_add_range(res.bm, 0, len_weak_region);
    }
    if (rtk_failed(s)) return;
    RTK_PL(s, 24);
    if (n_amb != 0) { const unsigned long long ta0 = rtk_clock(); rtk_fix_ambiguity(c, s_corr, sl_, q_corr, ql_, s_read + first_pos, res.old_len, n_amb); s.fine[9] += rtk_clock() - ta0; if (rtk_failed(s)) return; } // :716
    RTK_PL(s, 25);
    if (rtk_bm_card(res.bm, res.old_len) == res.old_len) { // :718-725 (G20): last k-mer of the WHOLE read vs last k-mer of the corrected region
        bool same = sl_ >= k && s_len >= k;
        for (uint32_t i = 0; same && i < k; ++i) same = rtk_bifrost_code(s_read[s_len - k + i]) == rtk_bifrost_code(s_corr[sl_ - k + i]);
        if (same) res.is_corrected = true;
    }
    if (!res.is_corrected) { // :727-747 trim the corrected string to the largest SHW end location of the raw region
        const unsigned long long tt0 = rtk_clock();
        RTK_SITE(13); const MyersResult a = rtk_align(c, s_read + first_pos, p2 - first_pos + k, s_corr, sl_, -1, RTK_MODE_SHW);
        s.fine[8] += rtk_clock() - tt0;
        if (a.dist >= 0) {
            const uint32_t keep = (a.first == -1) ? 0u : static_cast<uint32_t>(a.last + 1); // endLocations[0] == -1 wraps to SIZE_MAX in the reference
            if (keep < sl_) sl_ = keep;
            if (keep < ql_) ql_ = keep;
        }
    }
    res.seq_len = sl_; res.qual_len = ql_;
    RTK_PL(s, 26);
}
#endif

// ------------------------------------------------------------------------------------------------ generateConsensus (src/Alignment.cpp:309-470)
struct CigCur { const uint8_t* mv; uint32_t n, idx, qpos, rpos; }; // op-granular cursor over an alignment (moves 0/3 = M, 1 = I, 2 = D)
RTK_DEV char rtk_mv_op(uint8_t m) { return (m == 1) ? 'I' : (m == 2 ? 'D' : 'M'); }
RTK_DEV uint32_t rtk_op_len(const CigCur& cc) { // length of the run of equal ops at the cursor; 64 moves per step
    const uint8_t* mv = rtk_u(cc.mv); const uint32_t n = rtk_u(cc.n), idx = rtk_u(cc.idx);
    const char op = rtk_mv_op(rtk_ld(mv + idx));
    for (uint32_t j0 = idx; j0 < n; j0 += RTK_WAVE) {
        const uint32_t j = j0 + static_cast<uint32_t>(rtk_lane());
        const uint64_t diff = rtk_ballot(j < n && rtk_mv_op(mv[j]) != op);
        if (diff) return j0 + static_cast<uint32_t>(rtk_ffs(diff) - 1) - idx;
    }
    return n - idx;
}

RTK_FN void rtk_move_into_cigar(uint32_t start_, uint32_t end_, CigCur& cc_, uint32_t* rs_, uint32_t* re_, uint32_t* ref_out_) {
    uint32_t start = rtk_u(start_); uint32_t end = rtk_u(end_); CigCur& cc = *rtk_u(&cc_); uint32_t* rs = rtk_u(rs_); uint32_t* re = rtk_u(re_); uint32_t* ref_out = rtk_u(ref_out_); // moveIntoCIGAR (:354-411)
    uint32_t read_pos_start = cc.qpos, read_pos_end;
    while (cc.idx != cc.n && cc.rpos < start) {
        const uint32_t l = rtk_op_len(cc); const char op = rtk_mv_op(cc.mv[cc.idx]);
        if (op == 'M') { if (cc.rpos + l > start) { read_pos_start = cc.qpos + (start - cc.rpos); break; } cc.qpos += l; cc.rpos += l; }
        else if (op == 'I') cc.qpos += l; else cc.rpos += l;
        cc.idx += l; read_pos_start = cc.qpos;
    }
    read_pos_end = read_pos_start;
    while (cc.idx != cc.n && cc.rpos < end) {
        const uint32_t l = rtk_op_len(cc); const char op = rtk_mv_op(cc.mv[cc.idx]);
        if (op == 'M') { if (cc.rpos + l > end) { *rs = read_pos_start; *re = cc.qpos + (end - cc.rpos); *ref_out = end; return; } cc.qpos += l; cc.rpos += l; }
        else if (op == 'I') cc.qpos += l; else cc.rpos += l;
        cc.idx += l; read_pos_end = cc.qpos;
    }
    *rs = read_pos_start; *re = read_pos_end; *ref_out = cc.rpos;
}

// writes the consensus into out_s/out_q; returns false when the result is "empty" (caller falls back to the raw region)
// lane-parallel string predicates (wave-uniform results)
RTK_DEV bool rtk_str_equal(const char* a, const char* b, uint32_t n) {
    for (uint32_t i0 = 0; i0 < n; i0 += RTK_WAVE) { const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane()); if (rtk_ballot(i < n && a[i] != b[i]) != 0ull) return false; }
    return true;
}
RTK_DEV bool rtk_all_acgt(const char* p, uint32_t n) {
    for (uint32_t i0 = 0; i0 < n; i0 += RTK_WAVE) { const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane()); const char ch = i < n ? p[i] : 'A'; if (rtk_ballot(!(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T')) != 0ull) return false; }
    return true;
}

RTK_FN bool rtk_generate_consensus(const RCtx& c_, const ResCorr* fw_, const ResCorr* bw_, const char* ref_, uint32_t ref_len_, double max_norm_, char* out_s_, uint32_t* out_sl_, char* out_q_, uint32_t* out_ql_) {
    const RCtx& c = *rtk_u(&c_); const ResCorr* fw = rtk_u(fw_); const ResCorr* bw = rtk_u(bw_); RTK_ASSUME_LDS(fw); RTK_ASSUME_LDS(bw); const char* ref = rtk_u(ref_); uint32_t ref_len = rtk_u(ref_len_); double max_norm = rtk_u(max_norm_); char* out_s = rtk_u(out_s_); uint32_t* out_sl = rtk_u(out_sl_); char* out_q = rtk_u(out_q_); uint32_t* out_ql = rtk_u(out_ql_);
    RegionScratch& s = rtk_hdr(c);
    *out_sl = 0; *out_ql = 0;
    RTK_PL(s, 28);
    const uint32_t nfw = rtk_bm_card(fw->bm, fw->old_len), nbw = rtk_bm_card(bw->bm, bw->old_len);
    auto take = [&](const ResCorr* r) { rtk_app(s, out_s, out_sl, r->seq, r->seq_len); rtk_app(s, out_q, out_ql, r->qual, r->qual_len); return true; };
    if (nbw == 0 && nfw != 0) return take(fw);
    else if (nfw == 0 && nbw != 0) return take(bw);
    else if (nfw + nbw == 0) return false;
    if (nbw > nfw) { const ResCorr* t = fw; fw = bw; bw = t; }
    // NW path alignments of both corrections against the raw region; the moves are parked in str[3] (fw) and str[4] (bw)
    uint32_t nm_fw = 0, nm_bw = 0;
    RTK_SITE(14); const MyersResult afw = rtk_align_path(c, fw->seq, fw->seq_len, ref, ref_len, RTK_MODE_NW, &nm_fw);
    if (rtk_failed(s) || nm_fw > s.str_cap) { rtk_fail_ovf(s, 7); return false; }
    rtk_wcopy(s.str[3], s.my.moves, nm_fw);
    RTK_PL(s, 29);
    // Both directions usually arrive at the same corrected string: its alignment against the raw region is then the one just computed
    const bool same_strings = bw->seq_len == fw->seq_len && rtk_str_equal(bw->seq, fw->seq, fw->seq_len);
    MyersResult abw = afw;
    if (same_strings) { nm_bw = nm_fw; rtk_wcopy(s.str[4], s.str[3], nm_fw); }
    else {
        RTK_SITE(15); abw = rtk_align_path(c, bw->seq, bw->seq_len, ref, ref_len, RTK_MODE_NW, &nm_bw);
        if (rtk_failed(s) || nm_bw > s.str_cap) { rtk_fail_ovf(s, 7); return false; }
        rtk_wcopy(s.str[4], s.my.moves, nm_bw);
    }
    const double n_fw = static_cast<double>(afw.dist) / static_cast<double>(fw->seq_len > ref_len ? fw->seq_len : ref_len);
    const double n_bw = static_cast<double>(abw.dist) / static_cast<double>(bw->seq_len > ref_len ? bw->seq_len : ref_len);
    if (max_norm > 0.0 && (n_fw > max_norm || n_bw > max_norm)) {
        if (n_fw > max_norm && n_bw > max_norm) return false;
        if (n_fw > max_norm) return take(bw);
        return take(fw);
    }
    RTK_PL(s, 30);
    CigCur cf, cb;
    cf.mv = reinterpret_cast<const uint8_t*>(s.str[3].get()); cf.n = nm_fw; cf.idx = 0; cf.qpos = 0; cf.rpos = 0;
    cb.mv = reinterpret_cast<const uint8_t*>(s.str[4].get()); cb.n = nm_bw; cb.idx = 0; cb.qpos = 0; cb.rpos = 0;
    uint32_t i = 0;
    while (i < ref_len && !rtk_failed(s)) {
        int64_t len_fw = rtk_rc_len_corrected(*fw, i), len_bw = rtk_rc_len_corrected(*bw, i);
        if ((len_fw + len_bw) <= 0) {
            len_fw = rtk_rc_len_uncorrected(*fw, i); len_bw = rtk_rc_len_uncorrected(*bw, i);
            if (len_fw > len_bw || len_fw <= 0) len_fw = -1; else len_bw = -1;
        }
        uint32_t rs, re, rout;
        if (len_fw >= len_bw) {
            rtk_move_into_cigar(i, static_cast<uint32_t>(static_cast<int64_t>(i) + len_fw), cf, &rs, &re, &rout);
            if (re > rs) { rtk_app(s, out_s, out_sl, fw->seq + rs, (rs < fw->seq_len) ? ((re - rs) < (fw->seq_len - rs) ? (re - rs) : (fw->seq_len - rs)) : 0);
                           rtk_app(s, out_q, out_ql, fw->qual + rs, (rs < fw->qual_len) ? ((re - rs) < (fw->qual_len - rs) ? (re - rs) : (fw->qual_len - rs)) : 0); }
        } else {
            rtk_move_into_cigar(i, static_cast<uint32_t>(static_cast<int64_t>(i) + len_bw), cb, &rs, &re, &rout);
            if (re > rs) { rtk_app(s, out_s, out_sl, bw->seq + rs, (rs < bw->seq_len) ? ((re - rs) < (bw->seq_len - rs) ? (re - rs) : (bw->seq_len - rs)) : 0);
                           rtk_app(s, out_q, out_ql, bw->qual + rs, (rs < bw->qual_len) ? ((re - rs) < (bw->qual_len - rs) ? (re - rs) : (bw->qual_len - rs)) : 0); }
        }
        if (rout == i) { rtk_fail_ovf(s, 11); return false; } // no progress: would loop forever in the reference as well
        i = rout;
    }
    RTK_PL(s, 31);
    if (max_norm > 0.0 && !rtk_failed(s)) {
        // The merged string is very often one of the two inputs again. Its distance to the raw region is then the one computed above --
        // provided the plain configuration of this last call (edlibDefaultAlignConfig, :460: no IUPAC equalities) cannot tell the two
        // apart, i.e. both strings hold A/C/G/T only -- and that distance already passed the max_norm test above: nothing to compute.
        const bool is_fw = *out_sl == fw->seq_len && rtk_str_equal(out_s, fw->seq, fw->seq_len);
        const bool is_bw = !is_fw && *out_sl == bw->seq_len && rtk_str_equal(out_s, bw->seq, bw->seq_len);
        if ((is_fw || is_bw) && rtk_all_acgt(out_s, *out_sl) && rtk_all_acgt(ref, ref_len)) return true;
        RTK_SITE(16); const MyersResult a = rtk_align(c, out_s, *out_sl, ref, ref_len, -1, RTK_MODE_NW, /*iupac=*/false); // edlibDefaultAlignConfig (:460)
        const double n = static_cast<double>(a.dist) / static_cast<double>(*out_sl > ref_len ? *out_sl : ref_len);
        if (n > max_norm) { *out_sl = 0; *out_ql = 0; return take(fw); }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ scratch layout
RTK_HD uint64_t region_scratch_bytes(const RegionScratchCfg& c) {
    uint64_t b = scratch_bytes(c.my);
    b += 10ull * 4 * c.set_cap + 3ull * c.arena_cap + 4ull * (sizeof(UMap) * c.um_cap + c.str_cap) + (5ull + 8ull) * c.str_cap;
    b += 11ull * 8 * c.list_cap + 5ull * c.memo_cap + 3ull * 8 * c.bm_words + sizeof(RegionScratch) + 1024;
    return (b + 255) / 256 * 256;
}

// The RegionScratch header (pointers into the slab + the mutable control words: arena tops, working-path lengths, overflow flag,
// counters) is read on every step of the wave-level programs. The kernels keep it in LDS (`hdr` = a __shared__ object of the
// one-wave workgroup): a control-word read is an LDS access instead of an L2 / HBM round trip. hdr == nullptr: at the start of the slab.
RTK_DEV RegionScratch* region_scratch_carve(char* base, const RegionScratchCfg& c, RegionScratch* hdr = nullptr) {
    RegionScratch* s = hdr ? hdr : reinterpret_cast<RegionScratch*>(base);
    char* p = base + ((sizeof(RegionScratch) + 255) / 256 * 256);
    RegionScratch t;
    t.my = scratch_carve(p, c.my); p += scratch_bytes(c.my);
    for (int i = 0; i < 3; ++i) { t.arena[i] = p; p += c.arena_cap; t.top[i] = 0; }
    t.arena_cap = c.arena_cap;
    for (int i = 0; i < 11; ++i) { t.list[i] = reinterpret_cast<uint64_t*>(p); p += 8ull * c.list_cap; }
    t.list_cap = c.list_cap;
    for (int i = 0; i < 3; ++i) { t.bm[i] = reinterpret_cast<uint64_t*>(p); p += 8ull * c.bm_words; }
    t.bm_words = c.bm_words;
    for (int i = 0; i < 4; ++i) { t.wp[i].ums = reinterpret_cast<UMap*>(p); p += sizeof(UMap) * c.um_cap; t.wp[i].n = 0; t.wp[i].l = 0; t.wp[i].qlen = 0; }
    t.um_cap = c.um_cap;
    for (int i = 0; i < 10; ++i) { t.set[i] = reinterpret_cast<uint32_t*>(p); p += 4ull * c.set_cap; }
    t.set_cap = c.set_cap;
    t.memo_u = reinterpret_cast<uint32_t*>(p); p += 4ull * c.memo_cap; t.memo_cap = c.memo_cap; t.memo_n = 0;
    for (int i = 0; i < 4; ++i) { t.wp[i].qual = p; p += c.str_cap; }
    for (int i = 0; i < 5; ++i) { t.str[i] = p; p += c.str_cap; }
    for (int i = 0; i < 8; ++i) { t.rbuf[i] = p; p += c.str_cap; }
    t.str_cap = c.str_cap;
    t.memo_v = reinterpret_cast<uint8_t*>(p); p += c.memo_cap;
    t.ovf_word = 0; t.overflow = reinterpret_cast<uint32_t*>(&s->ovf_word); t.my.overflow = t.overflow;
    for (int i = 0; i < 16; ++i) { t.cnt[i] = 0; t.fine[i] = 0; }
#ifdef RTK_PROF
    for (int i = 0; i < 48; ++i) t.prof[i] = 0;
    t.prof_t = rtk_clock();
#endif
#ifndef RTK_SLIM_HDR
    for (int i = 0; i < 32; ++i) t.hist[i] = 0;
#endif
    *s = t; // every lane stores the same header
    return s;
}

// ------------------------------------------------------------------------------------------------ region driver (src/Correction.cpp:776-957)
RTK_FN void rtk_emit_segment(const RCtx& c_, RegionDesc* rd_, const char* sq_, uint32_t sl_, const char* ql_, uint32_t qll_) {
    const RCtx& c = *rtk_u(&c_); RegionDesc* rd = rtk_u(rd_); const char* sq = rtk_u(sq_); uint32_t sl = rtk_u(sl_); const char* ql = rtk_u(ql_); uint32_t qll = rtk_u(qll_);
    unsigned long long off = 0;
    if (rtk_lane() == 0) off = rtk_atomic_add(c.rb.seg_top, static_cast<unsigned long long>(sl) + qll);
    off = rtk_shfl(off, 0);
    if (off + sl + qll > c.rb.seg_cap) { rtk_fail_ovf(*c.sc, 12); return; }
    rtk_wcopy(c.rb.seg_pool + off, sq, sl);
    rtk_wcopy(c.rb.seg_pool + off + sl, ql, qll);
    rd->seg_off = off; rd->seq_len = sl; rd->qual_len = qll;
}

RTK_FN_DRIVER void rtk_region_program(const RCtx& c_, RegionDesc* rd_) {
    const RCtx& c = *rtk_u(&c_); RegionDesc* rd = rtk_u(rd_);
    RegionScratch& s = rtk_hdr(c);
    const uint32_t r = rtk_u(rd->read), k = static_cast<uint32_t>(c.k);
    const uint64_t base = rtk_u(c.bv.roff[r]);
    const uint32_t L = rtk_u(static_cast<uint32_t>(c.bv.roff[r + 1] - base));
    const char* s_fw = c.bv.seq + base; const char* s_bw = c.rb.seq_rc + base;
    const char q_min = rtk_get_qual(0.0, 0, static_cast<uint64_t>(c.o.max_qual)), q_max = rtk_get_qual(1.0, 0, static_cast<uint64_t>(c.o.max_qual));
    char* out_s = s.rbuf[4]; char* out_q = s.rbuf[5]; uint32_t& osl = s.loc.len[0]; uint32_t& oql = s.loc.len[1]; osl = 0; oql = 0;
    Anchors& so = s.loc.an[0]; Anchors& we = s.loc.an[1]; Anchors& so_r = s.loc.an[2]; Anchors& we_r = s.loc.an[3];
    so.pos = c.bv.s_pos + base; so.hit = nullptr; so.hits_by_pos = c.bv.hits + base; so.n = c.bv.n_solid[r]; so.L = L; so.rev = 0; so.k = c.k;
    we.pos = c.bv.wk_pos + c.bv.w_off[r]; we.hit = c.bv.wk_hit + c.bv.w_off[r]; we.hits_by_pos = nullptr; we.n = c.bv.w_cnt[r]; we.L = L; we.rev = 0; we.k = c.k;
    so_r = so; so_r.rev = 1; we_r = we; we_r.rev = 1;
    rtk_sync();
    ResCorr& fw = s.loc.rc[0]; ResCorr& bw = s.loc.rc[1];
    fw.seq = s.rbuf[0]; fw.qual = s.rbuf[1]; fw.bm = s.bm[0]; bw.seq = s.rbuf[2]; bw.qual = s.rbuf[3]; bw.bm = s.bm[1];
    if (L + 64 > s.str_cap) { rtk_fail_ovf(s, 7); return; }
    // pass 2 (long_read_correct): the read's own qualities are carried wherever pass 1 writes q_max / q_min, and a stretch whose bases all
    // have the maximum quality already is left alone (hasMinQual, src/Correction.hpp:45-52; :779, :808, :941)
    const bool lrc = c.o.long_read_correct != 0 && c.bv.qual.get() != nullptr;
    const char* q_fw = lrc ? c.bv.qual + base : nullptr; const char* q_bw = lrc ? c.rb.qual_rev + base : nullptr;
    auto has_min_qual = [&](uint32_t start, uint32_t end) -> bool {
        for (uint32_t i0 = start; i0 < end; i0 += RTK_WAVE) {
            const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
            bool bad = false;
            if (i < end) { const char ch = s_fw[i]; bad = (q_fw[i] < q_max) && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
            if (rtk_ballot(bad) != 0ull) return false;
        }
        return true;
    };
    auto app_q = [&](uint32_t pos, uint32_t n, char fill) { if (lrc) rtk_app(s, out_q, &oql, q_fw + pos, n); else rtk_app_fill(s, out_q, &oql, fill, n); }; // q_fw.substr(pos, n) | string(n, fill)
    const uint32_t kind = rtk_u(rd->kind);
    RTK_PL(s, 34);
    if (kind == RTK_RG_WHOLE_MAX || kind == RTK_RG_WHOLE_MIN) { // :165-171
        rtk_app(s, out_s, &osl, s_fw, L); app_q(0, L, kind == RTK_RG_WHOLE_MAX ? q_max : q_min);
    } else if (kind == RTK_RG_HEAD) { // :776-797
        if (!lrc || !has_min_qual(0, so.pos[0] + k)) {
            const uint32_t i_solid_rev = so.n - 1;
            uint32_t i_weak_rev = we.n;
            i_weak_rev = rtk_an_first_gt(we_r, 0, i_weak_rev, rtk_an_pos(so_r, i_solid_rev)); // the reference steps back while the previous weak anchor lies after the solid one
            rtk_correct_region(c, s_bw, L, so_r, we_r, i_solid_rev, i_weak_rev, nullptr, bw, q_fw); // q_fw next to s_bw: as the reference writes it (:787, G17)
            if (rtk_failed(s)) return;
            rtk_rc_reverse_complement(s, bw, s.bm[2], s.rbuf[6]);
            rtk_app(s, out_s, &osl, bw.seq, bw.seq_len >= k ? bw.seq_len - k : bw.seq_len); // substr(0, length - k): wraps to "everything" below k
            rtk_app(s, out_q, &oql, bw.qual, bw.qual_len >= k ? bw.qual_len - k : bw.qual_len);
        } else { rtk_app(s, out_s, &osl, s_fw, so.pos[0]); app_q(0, so.pos[0], q_min); }
    } else if (kind == RTK_RG_GAP) { // :803-935
        const uint32_t i = rtk_u(rd->i_solid), prev_pos = rtk_u(rd->prev_pos);
        const uint32_t pa = rtk_u(so.pos[i]), pb = rtk_u(so.pos[i + 1]);
        const UMap ua = rtk_u(rtk_an_um(so, i)), ub = rtk_u(rtk_an_um(so, i + 1));
        const uint32_t i_weak = rtk_u(rtk_an_first_ge(we, 0, we.n, pa)); // first weak anchor at or after the left solid anchor (:801)
        bool isUncorrected = false;
        bool sameUnitig = (ua.unitig == ub.unitig) && (ua.strand == ub.strand);
        if (lrc && has_min_qual(pa, pb + k)) isUncorrected = true; // :808
        else if (sameUnitig && !(c.g.flags[ua.unitig] & RTK_F_SHORT_CYCLE)) { // same-unitig shortcut (:814-858)
            const uint32_t min_pos = ua.dist < ub.dist ? ua.dist : ub.dist, max_pos = ua.dist < ub.dist ? ub.dist : ua.dist;
            const uint32_t len_query_km = pb - pa, len_unitig_km = max_pos - min_pos;
            uint64_t mn, mx; rtk_min_max_len(len_unitig_km, c.o.weak_region_len_factor, &mn, &mx);
            sameUnitig = sameUnitig && ((ua.strand && (ua.dist < ub.dist)) || (!ua.strand && (ua.dist > ub.dist)));
            sameUnitig = sameUnitig && (len_query_km >= mn) && (len_query_km <= mx);
            RTK_PL(s, 0);
            if (sameUnitig) {
                UMap sub = ua; sub.dist = min_pos; sub.len = len_unitig_km + 1;
                const uint32_t sl = rtk_ums_to_string(c, &sub, 1, s.str[0]); if (sl == 0xFFFFFFFFu) return;
                rtk_app(s, out_s, &osl, s_fw + prev_pos, pa - prev_pos);
                rtk_app(s, out_s, &osl, s.str[0], sl >= k ? sl - k : sl);
                if (lrc) { // :847-853
                    const uint32_t buff = (sl >= 2 * k) ? k : (sl - k);
                    rtk_app(s, out_q, &oql, q_fw + prev_pos, pa - prev_pos + buff);
                    if (sl - buff - k > 0) rtk_app_fill(s, out_q, &oql, q_max, sl - buff - k);
                } else rtk_app_fill(s, out_q, &oql, q_max, (pa - prev_pos) + (sl - k));
                RTK_PL(s, 1);
            } else isUncorrected = true;
        } else if (pb >= pa + k) {
            RTK_PL(s, 0);
            rtk_correct_region(c, s_fw, L, so, we, i, i_weak, nullptr, fw, q_fw);
            if (rtk_failed(s)) return;
            const uint32_t l_solid = pa - prev_pos;
            auto emit_minus_k = [&](const char* seq, uint32_t sl, const char* q, uint32_t ql) { // (prefix + x).substr(0, len - k)
                const uint32_t ts = l_solid + sl, tq = l_solid + ql;
                const uint32_t ks = ts >= k ? ts - k : ts, kq = tq >= k ? tq - k : tq;
                rtk_app(s, out_s, &osl, s_fw + prev_pos, ks < l_solid ? ks : l_solid); if (ks > l_solid) rtk_app(s, out_s, &osl, seq, ks - l_solid);
                app_q(prev_pos, kq < l_solid ? kq : l_solid, q_max); if (kq > l_solid) rtk_app(s, out_q, &oql, q, kq - l_solid);
            };
            if (fw.is_corrected) emit_minus_k(fw.seq, fw.seq_len, fw.qual, fw.qual_len);
            else {
                const uint32_t i_solid_bw = so.n - i - 2;
                uint32_t i_weak_bw = we.n - i_weak;
                i_weak_bw = rtk_an_first_gt(we_r, 0, i_weak_bw, rtk_an_pos(so_r, i_solid_bw));
                RTK_PL(s, 27);
{ const uint32_t gl_ = pb - pa; RTK_HIST_ADD(s, 16 + (gl_ < 40 ? 0 : gl_ < 64 ? 1 : gl_ < 128 ? 2 : gl_ < 256 ? 3 : gl_ < 512 ? 4 : gl_ < 1024 ? 5 : 6), 1); }
                rtk_correct_region(c, s_bw, L, so_r, we_r, i_solid_bw, i_weak_bw, &fw, bw, q_bw);
                if (rtk_failed(s)) return;
                RTK_PL(s, 27);
                rtk_rc_reverse_complement(s, bw, s.bm[2], s.rbuf[6]);
                if (bw.is_corrected) {
                    // l_solid = (|s_bw| - rev_pos(i_solid_bw + 1) - k) - prev_pos == pa - prev_pos
                    emit_minus_k(bw.seq, bw.seq_len, bw.qual, bw.qual_len);
                } else {
                    const uint32_t ref_len = pb - pa + k;
                    uint32_t& csl = s.loc.len[2]; uint32_t& cql = s.loc.len[3]; csl = 0; cql = 0;
                    const unsigned long long tc0 = rtk_clock();
                    const bool ok = rtk_generate_consensus(c, &fw, &bw, s_fw + pa, ref_len, c.o.weak_region_len_factor, s.rbuf[6], &csl, s.rbuf[7], &cql);
                    s.cnt[7] += rtk_clock() - tc0;
                    RTK_PL(s, 32);
                    if (rtk_failed(s)) return;
                    if (!ok || csl == 0) { // raw region, k solid qualities then minimum quality (:898-904)
                        csl = 0; cql = 0;
                        rtk_app(s, s.rbuf[6], &csl, s_fw + pa, ref_len);
                        if (lrc) rtk_app(s, s.rbuf[7], &cql, q_fw + pa, ref_len); // :902
                        else { rtk_app_fill(s, s.rbuf[7], &cql, q_max, k); rtk_app_fill(s, s.rbuf[7], &cql, q_min, pb - pa); }
                    }
                    emit_minus_k(s.rbuf[6], csl, s.rbuf[7], cql);
                }
            }
        } else isUncorrected = true;
        if (isUncorrected) { // :920-932
            rtk_app(s, out_s, &osl, s_fw + prev_pos, pb - prev_pos);
            if (lrc) rtk_app(s, out_q, &oql, q_fw + prev_pos, pb - prev_pos); // :924
            else {
                rtk_app_fill(s, out_q, &oql, q_max, pa - prev_pos);
                if (pb < pa + k) rtk_app_fill(s, out_q, &oql, q_max, pb - pa);
                else { rtk_app_fill(s, out_q, &oql, q_max, k); rtk_app_fill(s, out_q, &oql, q_min, pb - pa - k); }
            }
        }
    } else if (kind == RTK_RG_TAIL) { // :940-950
        const uint32_t i = rd->i_solid, prev_pos = rd->prev_pos;
        const uint32_t pa = so.pos[i];
        const uint32_t i_weak = rtk_an_first_ge(we, 0, we.n, pa);
        if (lrc && has_min_qual(pa, L)) { // :941: nothing to do, the else branch of :951-955
            rtk_app(s, out_s, &osl, s_fw + prev_pos, L - prev_pos); rtk_app(s, out_q, &oql, q_fw + prev_pos, L - prev_pos);
        } else {
            rtk_correct_region(c, s_fw, L, so, we, i, i_weak, nullptr, fw, q_fw);
            if (rtk_failed(s)) return;
            const uint32_t l_solid = pa - prev_pos;
            rtk_app(s, out_s, &osl, s_fw + prev_pos, l_solid); rtk_app(s, out_s, &osl, fw.seq, fw.seq_len);
            app_q(prev_pos, l_solid, q_max); rtk_app(s, out_q, &oql, fw.qual, fw.qual_len);
        }
    } else { // RTK_RG_TAIL_COPY (:951-955)
        const uint32_t i = rd->i_solid, prev_pos = rd->prev_pos;
        const uint32_t pa = so.pos[i];
        rtk_app(s, out_s, &osl, s_fw + prev_pos, L - prev_pos);
        if (lrc) rtk_app(s, out_q, &oql, q_fw + prev_pos, L - prev_pos);
        else { rtk_app_fill(s, out_q, &oql, q_max, pa - prev_pos + k); rtk_app_fill(s, out_q, &oql, q_min, L - pa - k); }
    }
    if (rtk_failed(s)) return;
    RTK_PL(s, 33);
    rtk_emit_segment(c, rd, out_s, osl, out_q, oql);
    RTK_PL(s, 35);
}

// ------------------------------------------------------------------------------------------------ region enumeration (one wave per read)
// dst[i] = tab[src[n - 1 - i]] (tab == nullptr: the characters as they are). One wave; four characters per lane and access (the reverse complement of a 64 Mb
// step was 0.7 of k_enum's 0.8 ms as byte loads, a twelve-way switch per character and byte stores: `tab` is the complement as a 256-byte table in LDS, one
// entry per bank), eight such words per lane in flight. The words are not aligned (a read starts anywhere): global accesses need not be.
RTK_DEV uint32_t rtk_ld_u32(const char* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
RTK_DEV void rtk_st_u32(char* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
RTK_FN void rtk_reverse_copy(char* __restrict__ dst_, const char* __restrict__ src_, uint32_t n_, const unsigned char* tab_) {
    char* __restrict__ const dst = rtk_gp(rtk_u(dst_)); const char* __restrict__ const src = rtk_gp(rtk_u(src_)); const uint32_t n = rtk_u(n_); const unsigned char* const tab = rtk_u(tab_);
    const uint32_t n4 = n & ~3u;
    constexpr uint32_t RW = 8;
    for (uint32_t i0 = 0; i0 < n4; i0 += 4u * RW * RTK_WAVE) {
        uint32_t w[RW];
#pragma unroll
        for (uint32_t u = 0; u < RW; ++u) { const uint32_t i = i0 + 4u * (u * RTK_WAVE + static_cast<uint32_t>(rtk_lane())); w[u] = i < n4 ? rtk_ld_u32(src + (n - 4u - i)) : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < RW; ++u) {
            const uint32_t i = i0 + 4u * (u * RTK_WAVE + static_cast<uint32_t>(rtk_lane()));
            uint32_t b0 = w[u] >> 24, b1 = (w[u] >> 16) & 0xFFu, b2 = (w[u] >> 8) & 0xFFu, b3 = w[u] & 0xFFu; // the last character of the word comes first
            if (tab) { RTK_ASSUME_LDS(tab); b0 = tab[b0]; b1 = tab[b1]; b2 = tab[b2]; b3 = tab[b3]; }
            if (i < n4) rtk_st_u32(dst + i, b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
        }
    }
    for (uint32_t i = n4 + static_cast<uint32_t>(rtk_lane()); i < n; i += RTK_WAVE) { { const unsigned char c = static_cast<unsigned char>(src[n - 1u - i]); unsigned char o = c; if (tab) { RTK_ASSUME_LDS(tab); o = tab[c]; } dst[i] = static_cast<char>(o); } }
}

RTK_FN void rtk_enum_regions(const GraphView& g, const BatchView& bv, const RegionBatch& rb, uint32_t r, const unsigned char* comp_tab) {
    const uint32_t k = static_cast<uint32_t>(g.k);
    const uint64_t base = bv.roff[r];
    const uint32_t L = static_cast<uint32_t>(bv.roff[r + 1] - base);
    const uint32_t* sp = bv.s_pos + base; const uint32_t ns = bv.n_solid[r];
    // reverse complement of the read (used by the head and backward corrections, src/Correction.cpp:175)
#ifndef RTK_AB_ENUM_RC_REPS // (developer A/B builds: what the reverse complement / the anchor loops cost, by doing them several times)
#define RTK_AB_ENUM_RC_REPS 1
#endif
#ifndef RTK_AB_ENUM_GAP_REPS
#define RTK_AB_ENUM_GAP_REPS 1
#endif
    // reverse complement of the read (used by the head and backward corrections, src/Correction.cpp:175); pass 2: the quality string reversed beside it
    for (int rep_ = 0; rep_ < RTK_AB_ENUM_RC_REPS; ++rep_) {
        rtk_reverse_copy(rb.seq_rc.get() + base, bv.seq.get() + base, L, comp_tab);
        if (bv.qual.get() != nullptr && rb.qual_rev.get() != nullptr) rtk_reverse_copy(rb.qual_rev.get() + base, bv.qual.get() + base, L, nullptr);
    }
    uint32_t n_gaps = 0;
    const bool whole = (L <= k) || ns == 0 || (ns == L - k + 1);
    // (this program runs on ONE wave per read and the launch lasts as long as its longest read -- tens of thousands of solid anchors: the anchors are
    // read sixteen chunks of 64 at a time, and what the descriptors need from a neighbouring anchor comes out of the lanes' registers, not from memory)
    constexpr uint32_t EU = 16;
    for (int rep_ = 0; rep_ < RTK_AB_ENUM_GAP_REPS; ++rep_) { n_gaps = 0;
    if (!whole) for (uint32_t c0 = 0; c0 + 1 < ns; c0 += EU * RTK_WAVE) {
        uint32_t a[EU], b2[EU];
        for (uint32_t u = 0; u < EU; ++u) { const uint32_t i = c0 + u * RTK_WAVE + static_cast<uint32_t>(rtk_lane()); const bool in = i + 1 < ns; a[u] = in ? sp[i] : 0u; b2[u] = in ? sp[i + 1] : 1u; }
        for (uint32_t u = 0; u < EU; ++u) n_gaps += static_cast<uint32_t>(rtk_popc(rtk_ballot(a[u] != b2[u] - 1u)));
    }
    }
    const uint32_t total = whole ? 1u : ((sp[0] != 0 ? 1u : 0u) + n_gaps + 1u);
    unsigned long long first = 0;
    if (rtk_lane() == 0) first = rtk_atomic_add(rb.n_regions, static_cast<unsigned long long>(total));
    first = rtk_shfl(first, 0);
    rb.r_first[r] = first; rb.r_count[r] = total;
    if (first + total > rb.regions_cap) return; // host notices n_regions > cap and retries with a bigger list
    RegionDesc* out = rb.regions + first;
    uint32_t w = 0;
    auto put = [&](uint32_t kind, uint32_t i_solid, uint32_t prev_pos) {
        RegionDesc d; d.read = r; d.kind = kind; d.i_solid = i_solid; d.prev_pos = prev_pos; d.seg_off = 0; d.seq_len = 0; d.qual_len = 0; d.status = 0; d.pad = 0;
        out[w++] = d;
    };
    if (whole) { put((L > k && ns != 0 && ns == L - k + 1) ? RTK_RG_WHOLE_MAX : RTK_RG_WHOLE_MIN, 0, 0); rtk_sync(); return; }
    if (sp[0] != 0) put(RTK_RG_HEAD, 0, 0);
    uint32_t prev_pos = sp[0];
    // the gaps between runs of consecutive solid anchors, 64 anchors at a time: every lane that sees a gap writes its descriptor. The
    // segment before it stopped at the anchor behind the PREVIOUS gap (prev_pos = sp[previous gap + 1], sp[0] for the first one)
    for (uint32_t c0 = 0; c0 + 1 < ns; c0 += EU * RTK_WAVE) {
        uint32_t a[EU], b2[EU]; // a = sp[i], b2 = sp[i + 1] (out of range: a pair without a gap)
        for (uint32_t u = 0; u < EU; ++u) { const uint32_t i = c0 + u * RTK_WAVE + static_cast<uint32_t>(rtk_lane()); const bool in = i + 1 < ns; a[u] = in ? sp[i] : 0u; b2[u] = in ? sp[i + 1] : 1u; }
        for (uint32_t u = 0; u < EU; ++u) {
            const uint32_t i = c0 + u * RTK_WAVE + static_cast<uint32_t>(rtk_lane());
            const bool gap = a[u] != b2[u] - 1u;
            const uint64_t bal = rtk_ballot(gap);
            if (bal == 0ull) continue;
            const uint64_t below = bal & ((1ull << rtk_lane()) - 1ull); // gaps of this chunk in front of this lane's
            const uint32_t behind_prev = rtk_shfl(b2[u], below ? (63 - __builtin_clzll(below)) : 0); // the anchor behind the previous gap of the chunk: sp[that gap + 1]
            if (gap) {
                RegionDesc d; d.read = r; d.kind = RTK_RG_GAP; d.i_solid = i; d.prev_pos = below ? behind_prev : prev_pos; d.seg_off = 0; d.seq_len = 0; d.qual_len = 0; d.status = 0; d.pad = 0;
                out[w + static_cast<uint32_t>(rtk_popc(below))] = d;
            }
            w += static_cast<uint32_t>(rtk_popc(bal));
            prev_pos = rtk_u(rtk_shfl(b2[u], 63 - __builtin_clzll(bal)));
        }
    }
    put(sp[ns - 1] < L - k ? RTK_RG_TAIL : RTK_RG_TAIL_COPY, ns - 1, prev_pos);
    rtk_sync();
}

// ------------------------------------------------------------------------------------------------ stitch
// Two steps (round 4; one wave per read copied a 100 kb read's thousand segments one after the other while the machine idled):
// (1) one wave per read adds up the lengths of its segments, reserves the read's place in the output pool and leaves every segment's
// place inside the read (st_off: characters / quality bytes in front of it); (2) the segments of ALL reads are copied 64 per wave.
RTK_FN void rtk_stitch_offsets(const BatchView& bv, const RegionBatch& rb, uint32_t r) {
    const uint64_t f0 = rb.r_first[r];
    const RegionDesc* rg = rb.regions + f0; uint64_t* st = rb.st_off.get() + f0; const uint32_t n = rb.r_count[r];
    // lengths of the read's segments, 64 at a time (the loads of a chunk are independent: one round trip per chunk, not per segment)
    uint64_t ts = 0, tq = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += RTK_WAVE) {
        const uint32_t i = c0 + static_cast<uint32_t>(rtk_lane());
        int a = 0, b = 0; if (i < n) { a = static_cast<int>(rg[i].seq_len); b = static_cast<int>(rg[i].qual_len); }
        int ta, tb; const int pa = rtk_wave_excl_scan(a, &ta), pb = rtk_wave_excl_scan(b, &tb);
        if (i < n) st[i] = ((tq + static_cast<uint64_t>(pb)) << 32) | (ts + static_cast<uint64_t>(pa));
        ts += static_cast<uint64_t>(rtk_u(ta)); tq += static_cast<uint64_t>(rtk_u(tb));
    }
    unsigned long long off = 0;
    if (rtk_lane() == 0) off = rtk_atomic_add(rb.out_top, static_cast<unsigned long long>(ts + tq));
    off = rtk_shfl(off, 0);
    rb.out_off[r] = off; rb.out_seq_len[r] = static_cast<uint32_t>(ts); rb.out_qual_len[r] = static_cast<uint32_t>(tq);
    (void)bv;
}

// segments c0 .. c0 + 63 of the flat list: their descriptors and their reads' places one per lane, then the copies back to back
RTK_FN void rtk_stitch_copy(const RegionBatch& rb, uint64_t c0, uint64_t n_regions) {
    const uint64_t i = c0 + static_cast<uint64_t>(rtk_lane());
    uint32_t sl = 0, ql = 0; uint64_t so = 0, ws = 0, wq = 0;
    if (i < n_regions) {
        const RegionDesc* rd = rb.regions.get() + i;
        const uint32_t r = rd->read; const uint64_t st = rb.st_off[i];
        const uint64_t off = rb.out_off[r], ts = rb.out_seq_len[r], tq = rb.out_qual_len[r];
        if (off + ts + tq <= rb.out_cap) { sl = rd->seq_len; ql = rd->qual_len; so = rd->seg_off; ws = off + (st & 0xFFFFFFFFull); wq = off + ts + (st >> 32); } // (a read beyond the pool's end is not written: the host sees out_top > out_cap)
    }
    const uint32_t m = (n_regions - c0) < static_cast<uint64_t>(RTK_WAVE) ? static_cast<uint32_t>(n_regions - c0) : static_cast<uint32_t>(RTK_WAVE);
    for (uint32_t j = 0; j < m; ++j) {
        const uint32_t jsl = rtk_shfl(sl, static_cast<int>(j)), jql = rtk_shfl(ql, static_cast<int>(j)); const uint64_t jso = rtk_shfl(so, static_cast<int>(j)), jws = rtk_shfl(ws, static_cast<int>(j)), jwq = rtk_shfl(wq, static_cast<int>(j));
        rtk_wcopy2(rb.out_pool + jws, rb.seg_pool + jso, jsl, rb.out_pool + jwq, rb.seg_pool + jso + jsl, jql);
    }
}

#endif
