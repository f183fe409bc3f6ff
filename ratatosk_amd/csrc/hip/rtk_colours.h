// chooseColors (reference: src/Correction.cpp:215-429) on bit vectors, for the regions that make up nearly all of a batch: the colour
// sets of the anchors around one weak region hold a few hundred distinct pair ids together (measured on configs[1]: < 512 in 90 % of
// the calls). Those ids are sorted once into a small universe kept in LDS; every set of the algorithm -- the six anchor classes, their
// unions / intersections / differences, curr_pid, all_pids -- is then ONE 64-bit word per lane (4096 bits), the whole class loop runs
// in registers (OR / AND / ANDN, popcount + wave sum, "the quota lowest ids" = a prefix count), and all_pids is expanded back into a
// sorted id list at the end. Same selections as the general sorted-array version in rtk_region.h, which stays the fallback for
// larger universes (returns RTK_NONE32 then). Bit order = id order, so "lowest ids first" is "lowest bits first".
#ifndef RTK_COLOURS_H
#define RTK_COLOURS_H

#define RTK_CB_MAX_IDS (RTK_LDS_SET_CAP - 384u)   // ids gathered from all anchors (with repeats); universe (u32), 256 radix counters and 64 scatter words share the 8 KB LDS buffer
#define RTK_CB_MAX_SLOTS 24u   // side-list entries whose bit vectors are kept

#ifndef RTK_SIM
// Least-significant-digit radix sort of n 32-bit keys by one wave, 8 bits per pass: the bitonic network it replaces costs 45-66 stages of
// LDS compare-exchanges (1 400 LDS operations per lane for 512 keys), this costs two passes over the keys per digit. `a` holds the keys
// (LDS) and the result; `b` is the other buffer (LDS or global memory, n entries); `bins` = 256 counters in LDS. A pass is stable: the 64
// keys of a chunk find their equals by eight ballots (one per digit bit), rank themselves among them, and chunks are taken in order.
RTK_DEV void rtk_radix_sort_u32(uint32_t* a, uint32_t* b, uint32_t n, uint32_t* bins, uint32_t max_key) {
    const uint32_t lane = static_cast<uint32_t>(rtk_lane());
    const uint64_t lt = (1ull << lane) - 1ull;
    int passes = 0; { uint32_t m = max_key; while (m) { ++passes; m >>= 8; } if (passes == 0) passes = 1; }
    if (passes & 1) ++passes; // an even number of passes: the result ends in `a`
    uint32_t* src = a; uint32_t* dst = b;
    for (int ps = 0; ps < passes; ++ps) {
        const int sh = 8 * ps;
        for (uint32_t i = lane; i < 256u; i += RTK_WAVE) bins[i] = 0u;
        RTK_WG_SYNC();
        // histogram of the digit
        for (uint32_t c0 = 0; c0 < n; c0 += RTK_WAVE) {
            const uint32_t i = c0 + lane; const bool ok = i < n;
            const uint32_t d = ok ? ((src[i] >> sh) & 0xFFu) : 0x100u;
            uint64_t eq = rtk_ballot(ok);
            for (int bt = 0; bt < 8; ++bt) { const uint64_t bb = rtk_ballot((d >> bt) & 1u); eq &= ((d >> bt) & 1u) ? bb : ~bb; }
            if (ok && (eq & lt) == 0ull) atomicAdd(&bins[d], static_cast<uint32_t>(rtk_popc(eq))); // the first lane of every group of equal digits
        }
        RTK_WG_SYNC();
        { // exclusive prefix over the 256 bins: four bins per lane
            uint32_t v[4]; uint32_t sum = 0;
            for (int x = 0; x < 4; ++x) { v[x] = bins[4u * lane + static_cast<uint32_t>(x)]; sum += v[x]; }
            int tot; uint32_t base = static_cast<uint32_t>(rtk_wave_excl_scan(static_cast<int>(sum), &tot));
            RTK_WG_SYNC();
            for (int x = 0; x < 4; ++x) { bins[4u * lane + static_cast<uint32_t>(x)] = base; base += v[x]; }
        }
        RTK_WG_SYNC();
        // stable scatter, chunk by chunk
        for (uint32_t c0 = 0; c0 < n; c0 += RTK_WAVE) {
            const uint32_t i = c0 + lane; const bool ok = i < n;
            const uint32_t key = ok ? src[i] : 0u;
            const uint32_t d = ok ? ((key >> sh) & 0xFFu) : 0x100u;
            uint64_t eq = rtk_ballot(ok);
            for (int bt = 0; bt < 8; ++bt) { const uint64_t bb = rtk_ballot((d >> bt) & 1u); eq &= ((d >> bt) & 1u) ? bb : ~bb; }
            const uint32_t before = static_cast<uint32_t>(rtk_popc(eq & lt));
            uint32_t base = 0;
            if (ok) base = bins[d];
            RTK_WG_SYNC();
            if (ok && before == 0u) bins[d] = base + static_cast<uint32_t>(rtk_popc(eq));
            if (ok) dst[base + before] = key;
            RTK_WG_SYNC();
        }
        uint32_t* t_ = src; src = dst; dst = t_;
    }
}
#endif

#ifdef RTK_SIM
struct RtkBM { uint64_t w[64]; };
inline RtkBM rtk_bm_zero() { RtkBM r; for (int i = 0; i < 64; ++i) r.w[i] = 0; return r; }
inline RtkBM operator|(const RtkBM& a, const RtkBM& b) { RtkBM r; for (int i = 0; i < 64; ++i) r.w[i] = a.w[i] | b.w[i]; return r; }
inline RtkBM operator&(const RtkBM& a, const RtkBM& b) { RtkBM r; for (int i = 0; i < 64; ++i) r.w[i] = a.w[i] & b.w[i]; return r; }
inline RtkBM rtk_bm_andn(const RtkBM& a, const RtkBM& b) { RtkBM r; for (int i = 0; i < 64; ++i) r.w[i] = a.w[i] & ~b.w[i]; return r; }
inline uint32_t rtk_bm_count(const RtkBM& a) { uint32_t c = 0; for (int i = 0; i < 64; ++i) c += static_cast<uint32_t>(__builtin_popcountll(a.w[i])); return c; }
inline RtkBM rtk_bm_lowest(const RtkBM& a, uint32_t q) { RtkBM r = rtk_bm_zero(); for (int i = 0; i < 64 && q; ++i) { uint64_t x = a.w[i]; while (x && q) { const uint64_t b = x & (~x + 1ull); r.w[i] |= b; x ^= b; --q; } } return r; }
inline RtkBM rtk_bm_load(const uint64_t* p) { RtkBM r; for (int i = 0; i < 64; ++i) r.w[i] = p[i]; return r; }
inline void rtk_bm_store(uint64_t* p, const RtkBM& a) { for (int i = 0; i < 64; ++i) p[i] = a.w[i]; }
#else
typedef uint64_t RtkBM; // word `lane` of a 4096-bit vector
RTK_DEV RtkBM rtk_bm_zero() { return 0ull; }
RTK_DEV RtkBM rtk_bm_andn(RtkBM a, RtkBM b) { return a & ~b; }
RTK_DEV uint32_t rtk_bm_count(RtkBM a) { return static_cast<uint32_t>(rtk_u(rtk_wave_sum(rtk_popc(a)))); }
RTK_DEV RtkBM rtk_bm_lowest(RtkBM a, uint32_t q) { // the q lowest set bits of the 4096-bit vector
    int total; const int before = rtk_wave_excl_scan(rtk_popc(a), &total);
    int keep = static_cast<int>(q) - before; const int mine = rtk_popc(a);
    keep = keep < 0 ? 0 : (keep > mine ? mine : keep);
    uint64_t rest = a; for (int i = 0; i < keep; ++i) rest &= rest - 1ull; // a without its `keep` lowest bits
    return a & ~rest;
}
RTK_DEV RtkBM rtk_bm_load(const uint64_t* p) { return p[rtk_lane()]; }
RTK_DEV void rtk_bm_store(uint64_t* p, RtkBM a) { p[rtk_lane()] = a; }
#endif

// bit vector of the ids of a sorted set inside the universe uni[0..U)
RTK_DEV RtkBM rtk_bm_from_ids(const uint32_t* uni, uint32_t U, uint64_t* scatter, const uint32_t* ids, uint32_t n) {
#ifdef RTK_SIM
    (void)scatter;
    RtkBM r = rtk_bm_zero();
    for (uint32_t i = 0; i < n; ++i) { const uint32_t x = rtk_lower_bound(uni, U, ids[i]); r.w[x >> 6] |= 1ull << (x & 63u); }
    return r;
#else
    scatter[rtk_lane()] = 0ull;
    RTK_WG_SYNC();
    for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < n; i += RTK_WAVE) {
        const uint32_t id = ids[i];
        uint32_t lo = 0, hi = U; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (uni[mid] < id) lo = mid + 1; else hi = mid; }
        atomicOr(reinterpret_cast<unsigned long long*>(scatter) + (lo >> 6), 1ull << (lo & 63u));
    }
    RTK_WG_SYNC();
    return scatter[rtk_lane()];
#endif
}


#ifndef RTK_SIM
// The same selection for the common small case -- at most 512 ids (with repeats) on at most 24 side unitigs -- with every chain of
// dependent memory round trips taken out: slot s lives in lane s (unitig, offsets and sizes of its colour lists, cardinality, flags:
// three round trips for all slots together instead of five per slot and pass), the candidate anchors are ranked in registers, the
// ids go from the colour pool straight into LDS (one flat pass over all lists), and the per-slot bit vectors (8 words at this size)
// stay in LDS. Returns RTK_NONE32 when the case is not small (caller goes on to rtk_choose_colors_bits).
#ifndef RTK_CS_MAX_IDS
#define RTK_CS_MAX_IDS 512u // (developer builds with a smaller LDS buffer lower it: profiles/scripts/build_wpe_variant.sh)
#endif
static_assert(2u * RTK_CS_MAX_IDS + 768u <= RTK_LDS_SET_CAP && RTK_CS_MAX_IDS <= 512u, "small path of the colour selection: universe + unsorted ids + 24 x 2 x 8 vector words in the LDS buffer");
// the 512-bit vectors of the small case live in lanes 0..7 (the other lanes hold zero): sums and prefix sums over eight lanes by DPP
// moves inside one row (quad permutes, half-row mirror, row shifts) instead of six cross-lane permutes through LDS
RTK_DEV int rtk_sum8(int v) { // every lane of 0..7 gets the sum over lanes 0..7 (callers read lane 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); // row_half_mirror: lane i <-> 7 - i
    return v;
}
RTK_DEV uint32_t rtk_bm8_count(RtkBM a) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(rtk_sum8(rtk_popc(a)))); }
RTK_DEV RtkBM rtk_bm8_lowest(RtkBM a, uint32_t q) { // the q lowest set bits
    const int mine = rtk_popc(a);
    int inc = mine; // inclusive prefix sum over the row (zeros shifted in)
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true); // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true); // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true); // row_shr:4
    int keep = static_cast<int>(q) - (inc - mine);
    keep = keep < 0 ? 0 : (keep > mine ? mine : keep);
    uint64_t rest = a; for (int i = 0; i < keep; ++i) rest &= rest - 1ull;
    return a & ~rest;
}
RTK_FN uint32_t rtk_choose_colors_small(const RCtx& c_, const SideList& side_s_, const SideList& side_e_, const SideList& side_w_) {
    const RCtx& c = *rtk_u(&c_); const SideList& side_s = *rtk_u(&side_s_); const SideList& side_e = *rtk_u(&side_e_); const SideList& side_w = *rtk_u(&side_w_); RTK_ASSUME_LDS(&side_s); RTK_ASSUME_LDS(&side_e); RTK_ASSUME_LDS(&side_w);
    RegionScratch& s = rtk_hdr(c);
    const GraphView& g = c.g;
    const uint32_t nw = rtk_u(side_w.n), ne = rtk_u(side_e.n), ns = rtk_u(side_s.n), n_slots = nw + ne + ns; // slot order: middle, right, left
    if (n_slots == 0 || n_slots > RTK_CB_MAX_SLOTS) return RTK_NONE32;
    const uint32_t lane = static_cast<uint32_t>(rtk_lane());
    const uint32_t* const pw = rtk_u(side_w.u); const uint32_t* const pe = rtk_u(side_e.u); const uint32_t* const ps = rtk_u(side_s.u);
    const uint8_t* const qw = rtk_u(side_w.nb); const uint8_t* const qe = rtk_u(side_e.nb); const uint8_t* const qs = rtk_u(side_s.nb);
    const uint32_t* const col = g.col; const uint64_t* const loff = g.loff; const uint64_t* const goff = g.goff; const int32_t* const gid = g.gid; const uint32_t* const cardp = g.card;
    unsigned long long tl_ = rtk_clock();
#define RTK_CS_LAP(i) { (void)tl_; }
    // ---- A. one lane per slot ----
    uint32_t m_u = 0, m_nl = 0, m_ng = 0, m_card = 0, m_nb = 0; uint64_t m_lo = 0, m_go = 0; int32_t m_gi = -1;
    if (lane < n_slots) {
        const uint32_t* pu; const uint8_t* pn; uint32_t i = lane;
        if (i < nw) { pu = pw; pn = qw; } else if (i - nw < ne) { i -= nw; pu = pe; pn = qe; } else { i -= nw + ne; pu = ps; pn = qs; }
        m_u = pu[i]; m_nb = pn[i];
        m_gi = gid[m_u]; m_lo = loff[m_u]; m_nl = static_cast<uint32_t>(loff[m_u + 1] - m_lo); m_card = cardp[m_u];
        if (m_gi >= 0) { m_go = goff[m_gi]; m_ng = static_cast<uint32_t>(goff[m_gi + 1] - m_go); }
    }
    int total = 0; const uint32_t st = static_cast<uint32_t>(rtk_wave_excl_scan(static_cast<int>(m_nl + m_ng), &total)); // first id of the slot in the flat order
    const uint32_t T = static_cast<uint32_t>(rtk_u(total));
    // two sizes: <= 512 ids -> 8-word bit vectors, the unsorted ids and the vectors in LDS; <= 1664 ids -> 64-word vectors in scratch memory
    const bool big = T > RTK_CS_MAX_IDS;
    if (T > RTK_CB_MAX_IDS || (big && (s.set_cap < RTK_CB_MAX_IDS || s.set_cap < 2u * 2u * 64u * RTK_CB_MAX_SLOTS))) return RTK_NONE32;
    RTK_CS_LAP(1)
    // ---- B. candidate anchors: cardinality >= min_cov_vertices, first occurrence of their unitig, ordered by (cardinality, unitig) [D1] ----
    const uint32_t min_cov_v = c.o.min_cov_vertices;
    bool dup = false;
    for (uint32_t j = 0; j + 1 < n_slots; ++j) { const uint32_t uj = rtk_shfl(m_u, static_cast<int>(j)); dup = dup || ((j < lane) && (uj == m_u)); }
    const bool cand = lane < n_slots && m_card >= min_cov_v && !dup;
    const uint64_t cb = rtk_ballot(cand); const uint32_t nsp = static_cast<uint32_t>(rtk_popc(cb));
    const uint64_t key = rtk_d1_key(m_card, m_u, c.o.d1_desc); // distinct among the candidates
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n_slots; ++j) { const uint64_t kj = rtk_shfl(key, static_cast<int>(j)); rank += (((cb >> j) & 1ull) && kj < key) ? 1u : 0u; }
    uint32_t src = 0;
    for (uint32_t j = 0; j < n_slots; ++j) { const uint32_t rj = rtk_shfl(rank, static_cast<int>(j)); if (((cb >> j) & 1ull) && rj == lane) src = j; }
    // lane j < nsp holds the j-th candidate: its slot, cardinality and remaining quota (p_spid.second)
    const uint32_t cov = 30;
    const uint32_t k_slot = src; const uint32_t k_card = rtk_shfl(m_card, static_cast<int>(src));
    uint32_t k_quota = k_card < cov ? k_card : cov;
    RTK_CS_LAP(2)
    // ---- C. universe: every id of every side unitig, straight into LDS, sorted, duplicates dropped ----
    uint32_t* const L = rtk_lds_set_buf();
    uint32_t* const uni = L; uint32_t* const raw = L + RTK_CS_MAX_IDS;
    // small: [0, 512) universe, [512, 1024) unsorted ids, [1024, 1792) 24 x 2 x 8 vector words (before that: second sort buffer + counters)
    // big:   [0, 1664) universe, [1664, 1920) sort counters; second sort buffer = set[1], 64-word vectors = set[2] (scratch memory)
    uint64_t* const cbm = big ? reinterpret_cast<uint64_t*>(s.set[2].get()) : reinterpret_cast<uint64_t*>(L + 2u * RTK_CS_MAX_IDS);
    const uint32_t VW = big ? 64u : 8u; // words per bit vector
    uint32_t P = 64; while (P < T) P <<= 1;
    for (uint32_t t0 = 0; t0 < P; t0 += RTK_WAVE) {
        const uint32_t t = t0 + lane;
        uint32_t i = 0;
        for (uint32_t j = 1; j < n_slots; ++j) { const uint32_t sj = rtk_shfl(st, static_cast<int>(j)); if (sj <= t) i = j; } // the last slot that starts at or before t (empty slots share their start with the next one)
        const uint32_t s_i = rtk_shfl(st, static_cast<int>(i)), nl_i = rtk_shfl(m_nl, static_cast<int>(i));
        const uint64_t lo_i = rtk_shfl(m_lo, static_cast<int>(i)), go_i = rtk_shfl(m_go, static_cast<int>(i));
        uint32_t x = 0xFFFFFFFFu;
        if (t < T) { const uint32_t off = t - s_i; x = col[off < nl_i ? lo_i + off : go_i + (off - nl_i)]; if (!big) raw[t] = x; }
        if (t < T || !big) uni[t] = x;
    }
    RTK_WG_SYNC();
    s.cnt[1] += T;
    RTK_CS_LAP(3)
    { // sorted by id (radix, 8 bits per pass; the second buffer and the counters sit in the part of the LDS buffer the slot bit vectors take later)
        uint32_t mx = 0; for (uint32_t i2 = lane; i2 < T; i2 += RTK_WAVE) mx = uni[i2] > mx ? uni[i2] : mx;
        for (int o = 32; o > 0; o >>= 1) { const uint32_t v2 = static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx), o, 64)); mx = v2 > mx ? v2 : mx; }
        if (big) rtk_radix_sort_u32(uni, s.set[1], T, L + RTK_CB_MAX_IDS, rtk_u(mx));
        else rtk_radix_sort_u32(uni, L + 2u * RTK_CS_MAX_IDS, T, L + 2u * RTK_CS_MAX_IDS + RTK_CS_MAX_IDS, rtk_u(mx));
    }
    uint32_t U = 0;
    for (uint32_t i0 = 0; i0 < T; i0 += RTK_WAVE) {
        const uint32_t i = i0 + lane;
        uint32_t x = 0; bool keep = false;
        if (i < T) { x = uni[i]; keep = (i == 0) || (uni[i - 1] != x); }
        RTK_WG_SYNC();
        const uint64_t bal = rtk_ballot(keep);
        if (keep) uni[U + static_cast<uint32_t>(rtk_popc(bal & ((1ull << lane) - 1ull)))] = x;
        U += static_cast<uint32_t>(rtk_popc(bal));
        RTK_WG_SYNC();
    }
    RTK_CS_LAP(4)
    // ---- D. bit vectors of every slot (local part, global part), 8 words each, in LDS: one flat pass over the gathered ids, every id
    // ranked in the universe by a binary search in LDS and its bit set in the vector of the list it came from ----
    for (uint32_t i = lane; i < n_slots * 2u * VW; i += RTK_WAVE) cbm[i] = 0ull;
    RTK_WG_SYNC();
    for (uint32_t t0 = 0; t0 < T; t0 += RTK_WAVE) {
        const uint32_t t = t0 + lane;
        uint32_t i = 0;
        for (uint32_t j = 1; j < n_slots; ++j) { const uint32_t sj = rtk_shfl(st, static_cast<int>(j)); if (sj <= t) i = j; }
        const uint32_t s_i = rtk_shfl(st, static_cast<int>(i)), nl_i = rtk_shfl(m_nl, static_cast<int>(i));
        const uint64_t lo_i2 = rtk_shfl(m_lo, static_cast<int>(i)), go_i2 = rtk_shfl(m_go, static_cast<int>(i));
        if (t < T) {
            const uint32_t off2 = t - s_i;
            const uint32_t id = big ? col[off2 < nl_i ? lo_i2 + off2 : go_i2 + (off2 - nl_i)] : raw[t];
            uint32_t lo = 0, hi = U; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (uni[mid] < id) lo = mid + 1; else hi = mid; }
            const uint32_t seg = 2u * i + ((t - s_i) >= nl_i ? 1u : 0u);
            atomicOr(reinterpret_cast<unsigned long long*>(cbm) + seg * VW + (lo >> 6), 1ull << (lo & 63u));
        }
    }
    RTK_WG_SYNC();
    RTK_CS_LAP(5)
    // (big: the vectors were written by L2 atomics, so they are read past the L1, whose copy of these lines may be older)
    auto ld = [&](uint32_t idx) -> RtkBM { if (lane >= VW) return 0ull; return big ? static_cast<RtkBM>(__hip_atomic_load(reinterpret_cast<const unsigned long long*>(cbm) + idx * VW + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : cbm[idx * VW + lane]; };
    auto cnt_ = [&](RtkBM a) -> uint32_t { return big ? rtk_bm_count(a) : rtk_bm8_count(a); };
    auto low_ = [&](RtkBM a, uint32_t q) -> RtkBM { return big ? rtk_bm_lowest(a, q) : rtk_bm8_lowest(a, q); };
    // ---- E. the six anchor classes: side (middle, right, left) x branching / non-branching; G2: the global set alone when there is one ----
    RtkBM a[6];
    for (int sh = 0; sh < 6; ++sh) {
        RtkBM acc = 0ull;
        const uint32_t first = (sh % 3 == 0) ? 0u : (sh % 3 == 1 ? nw : nw + ne), cnt = (sh % 3 == 0) ? nw : (sh % 3 == 1 ? ne : ns);
        const uint32_t want_nb = sh >= 3 ? 1u : 0u;
        for (uint32_t slot = first; slot < first + cnt; ++slot) {
            if (rtk_u(rtk_shfl(m_nb, static_cast<int>(slot))) != want_nb) continue;
            const bool has_global = rtk_u(rtk_shfl(m_gi, static_cast<int>(slot))) >= 0;
            acc |= ld(2u * slot + (has_global ? 1u : 0u));
        }
        a[sh] = acc;
    }
    const RtkBM pos0 = a[0] | a[3], pos1 = a[1] | a[4], pos2 = a[2] | a[5];
    const RtkBM a01 = pos0 & pos1, a12 = pos1 & pos2, a02 = pos0 & pos2;
    const RtkBM nobranch_all = a[3] | a[4] | a[5];
    const RtkBM i3 = a01 & a12, i2 = a01 | a12 | a02;
    RtkBM nobranch = nobranch_all, branching = 0ull, prev2 = 0ull, all = 0ull;
    uint32_t nb_unselected = nsp;
    // ---- F. class loop (:331-429) ----
    for (int i = 5; i >= 0; --i) {
        if (nb_unselected == 0) break;
        RtkBM a2;
        if (i == 5) a2 = nobranch & i3;
        else if (i == 4) { nobranch = rtk_bm_andn(nobranch, prev2); a2 = nobranch & i2; }
        else if (i == 3) { nobranch = rtk_bm_andn(nobranch, prev2); a2 = nobranch; }
        else if (i == 2) { branching = rtk_bm_andn(a[0] | a[1] | a[2], nobranch_all); a2 = branching & i3; }
        else if (i == 1) { branching = rtk_bm_andn(branching, prev2); a2 = branching & i2; }
        else { branching = rtk_bm_andn(branching, prev2); a2 = branching; }
        prev2 = a2;
        if (cnt_(a2) == 0) continue;
        nb_unselected = 0;
        RtkBM curr = a2;
        for (uint32_t j = 0; j < nsp; ++j) {
            int quota = static_cast<int>(rtk_u(rtk_shfl(k_quota, static_cast<int>(j))));
            if (quota > 0) {
                const uint32_t slot = rtk_u(rtk_shfl(k_slot, static_cast<int>(j)));
                const RtkBM cu = ld(2u * slot) | ld(2u * slot + 1u); // all colours of the anchor
                if (i == 0 || cnt_(cu & curr) >= 1) {
                    const uint32_t cd = rtk_u(rtk_shfl(k_card, static_cast<int>(j))); const uint32_t min_cov = cd < cov ? cd : cov;
                    const uint32_t sh = cnt_(cu & all);
                    quota = static_cast<int>(min_cov - (sh < min_cov ? sh : min_cov));
                    if (quota > 0) {
                        const uint32_t all_card = cnt_(all);
                        const RtkBM pid = low_(cu & curr, static_cast<uint32_t>(quota));
                        all = all | pid; curr = rtk_bm_andn(curr, pid);
                        const int gained = static_cast<int>(cnt_(all) - all_card);
                        quota -= gained < quota ? gained : quota;
                    }
                }
                if (lane == j) k_quota = static_cast<uint32_t>(quota);
            }
            nb_unselected += quota > 0 ? 1u : 0u;
        }
    }
    RTK_CS_LAP(6)
    // ---- all_pids back to a sorted id list in set[0] ----
    const uint32_t n_all = rtk_bm_count(all);
    if (n_all > s.set_cap) { rtk_fail_ovf(s, 9); return 0; }
    uint32_t* out = s.set[0];
    { int tot2; uint32_t at = static_cast<uint32_t>(rtk_wave_excl_scan(rtk_popc(all), &tot2)); uint64_t x = all;
      while (x) { const int b = __builtin_ctzll(x); out[at++] = uni[64u * lane + static_cast<uint32_t>(b)]; x &= x - 1ull; } }
    rtk_sync();
    return n_all;
}
#endif

// Returns |all_pids| (ids in s.set[0]), or RTK_NONE32 when the anchors' sets do not fit the small universe (caller falls back).
RTK_FN uint32_t rtk_choose_colors_bits(const RCtx& c_, const SideList& side_s_, const SideList& side_e_, const SideList& side_w_) {
    const RCtx& c = *rtk_u(&c_); const SideList& side_s = *rtk_u(&side_s_); const SideList& side_e = *rtk_u(&side_e_); const SideList& side_w = *rtk_u(&side_w_); RTK_ASSUME_LDS(&side_s); RTK_ASSUME_LDS(&side_e); RTK_ASSUME_LDS(&side_w);
    RegionScratch& s = rtk_hdr(c);
    const GraphView& g = c.g;
    const SideList* sides[3] = {&side_w, &side_e, &side_s};
    const uint32_t n_slots = side_w.n + side_e.n + side_s.n;
    if (n_slots == 0 || n_slots > RTK_CB_MAX_SLOTS || s.set_cap < 2 * RTK_CB_MAX_IDS + 64u * 4u * RTK_CB_MAX_SLOTS || s.list_cap < 2 * RTK_CB_MAX_SLOTS) return RTK_NONE32;
    // how many ids in all (global + local of every side unitig)
    uint32_t T = 0;
    for (int sd = 0; sd < 3; ++sd) for (uint32_t i = 0; i < sides[sd]->n; ++i) {
        const uint32_t u = sides[sd]->u[i]; const int32_t gi = g.gid[u];
        T += static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]) + (gi >= 0 ? static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi]) : 0u);
        if (T > RTK_CB_MAX_IDS) return RTK_NONE32;
    }
    // candidate anchors: cardinality >= min_cov_vertices, ordered by (cardinality, unitig id) [D1]; the value carried through the sort is
    // the anchor's slot (its position in the concatenated side lists: middle, right, left)
    uint64_t* keys = s.list[4]; uint64_t* vals = s.list[3]; uint64_t* slot_of = s.list[5];
    uint32_t nsp = 0;
    { uint32_t slot = 0;
      for (int sd = 0; sd < 3; ++sd) for (uint32_t i = 0; i < sides[sd]->n; ++i, ++slot) {
        const uint32_t u = sides[sd]->u[i];
        if (g.card[u] < c.o.min_cov_vertices) continue;
        bool dup = false; for (uint32_t j0 = 0; j0 < nsp && !dup; j0 += RTK_WAVE) { const uint32_t j = j0 + static_cast<uint32_t>(rtk_lane()); dup = rtk_ballot(j < nsp && rtk_d1_unitig(keys[j], c.o.d1_desc) == u) != 0ull; }
        if (dup) continue;
        keys[nsp] = rtk_d1_key(g.card[u], u, c.o.d1_desc); vals[nsp] = slot; ++nsp; rtk_sync();
      } }
    rtk_sort_pairs(keys, vals, nsp); // (uses the LDS buffer: before the universe moves in)
    const uint32_t cov = 30;
    for (uint32_t j = static_cast<uint32_t>(rtk_lane()); j < nsp; j += RTK_WAVE) { slot_of[j] = vals[j]; const uint32_t cd = static_cast<uint32_t>(keys[j] >> 32); vals[j] = cd < cov ? cd : cov; } // remaining quota (p_spid.second)
    rtk_sync();
    // ---- universe: every id of every side unitig, sorted, duplicates dropped ----
    uint32_t* gathered = s.set[1];
    { uint32_t at = 0;
      for (int sd = 0; sd < 3; ++sd) for (uint32_t i = 0; i < sides[sd]->n; ++i) {
        const uint32_t u = sides[sd]->u[i]; const int32_t gi = g.gid[u];
        const uint32_t nl = static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]); const uint32_t* pl = g.col + g.loff[u];
        for (uint32_t x = static_cast<uint32_t>(rtk_lane()); x < nl; x += RTK_WAVE) gathered[at + x] = pl[x];
        at += nl;
        if (gi >= 0) { const uint32_t ng = static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi]); const uint32_t* pg = g.col + g.goff[gi];
                       for (uint32_t x = static_cast<uint32_t>(rtk_lane()); x < ng; x += RTK_WAVE) gathered[at + x] = pg[x]; at += ng; }
      } }
    rtk_sync();
    s.cnt[1] += T;
    uint32_t U = 0;
#ifdef RTK_SIM
    uint32_t* const uni = s.set[1] + RTK_CB_MAX_IDS; uint64_t* const scatter = nullptr;
    { for (uint32_t i = 0; i < T; ++i) uni[i] = gathered[i];
      std::sort(uni, uni + T);
      for (uint32_t i = 0; i < T; ++i) if (i == 0 || uni[i] != uni[i - 1]) uni[U++] = uni[i]; }
#else
    uint32_t* const uni = rtk_lds_set_buf(); uint64_t* const scatter = reinterpret_cast<uint64_t*>(uni + RTK_CB_MAX_IDS + 256u); // [0, 1664) ids, [1664, 1920) radix counters, [1920, 2048) scatter words
    { // radix sort of the ids in LDS (second buffer: the gathered copy in scratch memory)
      uint32_t mx = 0;
      for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < T; i += RTK_WAVE) { const uint32_t x = gathered[i]; uni[i] = x; mx = x > mx ? x : mx; }
      for (int o = 32; o > 0; o >>= 1) { const uint32_t v2 = static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx), o, 64)); mx = v2 > mx ? v2 : mx; }
      RTK_WG_SYNC();
      rtk_radix_sort_u32(uni, gathered, T, uni + RTK_CB_MAX_IDS, rtk_u(mx));
      for (uint32_t i0 = 0; i0 < T; i0 += RTK_WAVE) { // forward compaction of the first elements of the runs
          const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
          uint32_t x = 0; bool keep = false;
          if (i < T) { x = uni[i]; keep = (i == 0) || (uni[i - 1] != x); }
          RTK_WG_SYNC();
          const uint64_t bal = rtk_ballot(keep);
          if (keep) uni[U + static_cast<uint32_t>(rtk_popc(bal & ((1ull << rtk_lane()) - 1ull)))] = x;
          U += static_cast<uint32_t>(rtk_popc(bal));
          RTK_WG_SYNC();
      } }
#endif
    // ---- bit vectors of every side unitig: global part, local part (kept in scratch: 2 x 512 B per slot) ----
    uint64_t* const store = reinterpret_cast<uint64_t*>(s.set[2].get());
    { uint32_t slot = 0;
      for (int sd = 0; sd < 3; ++sd) for (uint32_t i = 0; i < sides[sd]->n; ++i, ++slot) {
        const uint32_t u = sides[sd]->u[i]; const int32_t gi = g.gid[u];
        const RtkBM bl = rtk_bm_from_ids(uni, U, scatter, g.col + g.loff[u], static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]));
        rtk_bm_store(store + (2ull * slot) * 64ull, bl);
        const RtkBM bg = gi >= 0 ? rtk_bm_from_ids(uni, U, scatter, g.col + g.goff[gi], static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi])) : rtk_bm_zero();
        rtk_bm_store(store + (2ull * slot + 1ull) * 64ull, bg);
      } }
    rtk_sync();
    // ---- the six anchor classes: side (middle, right, left) x branching / non-branching; G2: the global set alone when there is one ----
    RtkBM a[6];
    for (int sh = 0; sh < 6; ++sh) {
        RtkBM acc = rtk_bm_zero();
        uint32_t slot = (sh % 3 == 0) ? 0u : (sh % 3 == 1 ? side_w.n : side_w.n + side_e.n);
        const SideList& sl = *sides[sh % 3]; const uint8_t want_nb = sh >= 3 ? 1 : 0;
        for (uint32_t i = 0; i < sl.n; ++i, ++slot) {
            if (sl.nb[i] != want_nb) continue;
            const bool has_global = g.gid[sl.u[i]] >= 0;
            acc = acc | rtk_bm_load(store + (2ull * slot + (has_global ? 1ull : 0ull)) * 64ull);
        }
        a[sh] = acc;
    }
    const RtkBM pos0 = a[0] | a[3], pos1 = a[1] | a[4], pos2 = a[2] | a[5];
    const RtkBM a01 = pos0 & pos1, a12 = pos1 & pos2, a02 = pos0 & pos2;
    const RtkBM nobranch_all = a[3] | a[4] | a[5];
    const RtkBM i3 = a01 & a12, i2 = a01 | a12 | a02;
    RtkBM nobranch = nobranch_all, branching = rtk_bm_zero(), prev2 = rtk_bm_zero(), all = rtk_bm_zero();
    uint32_t nb_unselected = nsp;
    for (int i = 5; i >= 0; --i) {
        if (nb_unselected == 0) break;
        RtkBM a2;
        if (i == 5) a2 = nobranch & i3;
        else if (i == 4) { nobranch = rtk_bm_andn(nobranch, prev2); a2 = nobranch & i2; }
        else if (i == 3) { nobranch = rtk_bm_andn(nobranch, prev2); a2 = nobranch; }
        else if (i == 2) { branching = rtk_bm_andn(a[0] | a[1] | a[2], nobranch_all); a2 = branching & i3; }
        else if (i == 1) { branching = rtk_bm_andn(branching, prev2); a2 = branching & i2; }
        else { branching = rtk_bm_andn(branching, prev2); a2 = branching; }
        prev2 = a2;
        if (rtk_bm_count(a2) == 0) continue;
        nb_unselected = 0;
        RtkBM curr = a2;
        for (uint32_t j = 0; j < nsp; ++j) {
            const uint32_t u = rtk_d1_unitig(rtk_ld(keys + j), c.o.d1_desc);
            int quota = static_cast<int>(rtk_ld(vals + j));
            if (quota > 0) {
                const uint64_t slot = rtk_ld(slot_of + j);
                const RtkBM cu = rtk_bm_load(store + (2ull * slot) * 64ull) | rtk_bm_load(store + (2ull * slot + 1ull) * 64ull); // all colours of u
                if (i == 0 || rtk_bm_count(cu & curr) >= 1) {
                    const uint32_t cd = rtk_ld(g.card.get() + u); const uint32_t min_cov = cd < cov ? cd : cov;
                    const uint32_t sh = rtk_bm_count(cu & all);
                    quota = static_cast<int>(min_cov - (sh < min_cov ? sh : min_cov));
                    if (quota > 0) {
                        const uint32_t all_card = rtk_bm_count(all);
                        const RtkBM pid = rtk_bm_lowest(cu & curr, static_cast<uint32_t>(quota));
                        all = all | pid; curr = rtk_bm_andn(curr, pid);
                        const int gained = static_cast<int>(rtk_bm_count(all) - all_card);
                        quota -= gained < quota ? gained : quota;
                    }
                }
            }
            vals[j] = static_cast<uint64_t>(quota);
            nb_unselected += quota > 0 ? 1u : 0u;
        }
        rtk_sync();
    }
    // ---- all_pids back to a sorted id list in set[0] ----
    const uint32_t n_all = rtk_bm_count(all);
    if (n_all > s.set_cap) { rtk_fail_ovf(s, 9); return 0; }
    uint32_t* out = s.set[0];
#ifdef RTK_SIM
    { uint32_t at = 0; for (uint32_t w = 0; w < 64; ++w) { uint64_t x = all.w[w]; while (x) { const int b = __builtin_ctzll(x); out[at++] = uni[64u * w + static_cast<uint32_t>(b)]; x &= x - 1ull; } } }
#else
    { int total; uint32_t at = static_cast<uint32_t>(rtk_wave_excl_scan(rtk_popc(all), &total)); uint64_t x = all;
      while (x) { const int b = __builtin_ctzll(x); out[at++] = uni[64u * static_cast<uint32_t>(rtk_lane()) + static_cast<uint32_t>(b)]; x &= x - 1ull; } }
#endif
    rtk_sync();
    return n_all;
}

#endif
