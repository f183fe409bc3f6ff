// Hirschberg alignment level by level with the banded half passes of a level packed side by side into the lanes of a wave ("gang").
// (reference: src/edlib.cpp:1234-1399 obtainAlignmentHirschberg, :1164-1216 the traceback / Hirschberg switch; included by rtk_myers.h)
//
// Why: the band of a sub-problem is k + 1 diagonals wide (rtk_myers.h), i.e. ~13 query words for a 10 kb read at its top level and half of
// that at every level below, while a level has twice as many passes as the one above: one pass per sweep leaves 80-95 % of the lanes idle.
// A gang sweep gives every pass of a level its own run of L lanes, L = ceil((band + 99) / 65), in which the words of the pass rotate as in
// the ring sweep (word w on lane w mod L; it enters the band at the bottom, leaves it 64 + band columns later, the lane moves on to word
// w + L); the horizontal delta travels from lane to lane inside a run with one ds_bpermute per step, every lane reads the target character of
// its own column (targets are staged once per sweep, the right halves reversed, so that all lanes walk forwards; characters are fetched four
// steps ahead of their use). The sweep lasts as long as its longest pass: a level of a 10 kb read is n / 2^(l+1) + W steps for ALL its passes.
// Passes whose band does not fit 64 lanes (band > ~4000 diagonals: the top levels of reads above ~40 kb) keep the banded row blocks.
//
// The driver builds the split tree level by level like the multi-wave driver of round 2 did (lists of sub-problems in read order; a round
// of at most RTK_LVL_MAXN sub-problems at a time, their records in the scratch's small stack area), on one wave or -- in the multi-wave
// kernels -- with the gangs and row blocks of a round as work items of the workgroup's waves; the leaves are traced back by all waves.
#ifndef RTK_MYERS_LVL_H
#define RTK_MYERS_LVL_H
#ifndef RTK_SIM

#define RTK_LVL_NODE 12   // ints per sub-problem record of a round
#define RTK_LVL_MAXN 26   // sub-problems per round (26 * 12 ints fit the 5 * 64 ints of MyersScratch::hstack)
#define RTK_GANG_PERIOD 32 // steps between two looks at "is a lane done with its word" (the lazy word switch)
// record: 0 q0, 1 qm, 2 t0, 3 tn, 4 bs (-1: unknown), 5 slot in the next list, 6 offset of its four delta vectors (64-bit words), 7 k of its band,
//         8 / 9 first lane | gang << 8 of the left / right half pass, 10 lanes per pass (0: row blocks), 11 first row-block job (multi-wave)

// lanes of one pass of a gang: a lane must be done with word w (band + 64 columns after it started, looked at every RTK_GANG_PERIOD steps,
// 4 steps of character prefetch) before word w + L starts, 65 L steps after word w did
__device__ __forceinline__ int rtk_gang_lanes(int band_width, int wa) { const int l = (band_width + 63 + RTK_GANG_PERIOD + 4 + 64) / 65; return l < wa ? l : wa; }
__device__ __forceinline__ bool rtk_hb_is_leaf(int qm, int tn) { return qm == 0 || tn == 0 || (2LL * 8 + 4) * ((qm + 63) >> 6) * tn + 8LL * tn < 1024 * 1024; } // edlib.cpp:1191-1193

// Advance_Block (rtk_myers_step) with the outgoing delta read from one half of the delta words (the bit's half and position are per-word constants)
__device__ __forceinline__ int rtk_gang_step64(uint64_t& Pv, uint64_t& Mv, uint64_t Eq, int hin, bool bit_hi, int bit_pos) {
    const uint64_t pv = Pv, mv = Mv;
    const uint64_t Xv = Eq | mv;
    Eq |= static_cast<uint64_t>(static_cast<uint32_t>(hin) >> 31); // hin < 0
    const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
    uint64_t Ph = mv | ~(Xh | pv);
    uint64_t Mh = pv & Xh;
    const uint32_t ph = bit_hi ? static_cast<uint32_t>(Ph >> 32) : static_cast<uint32_t>(Ph), mh = bit_hi ? static_cast<uint32_t>(Mh >> 32) : static_cast<uint32_t>(Mh);
    const int hout = static_cast<int>((ph >> bit_pos) & 1u) - static_cast<int>((mh >> bit_pos) & 1u);
    Ph = (Ph << 1) | (hin > 0 ? 1ull : 0ull); Mh = (Mh << 1) | static_cast<uint64_t>(static_cast<uint32_t>(hin) >> 31);
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}

#ifdef RTK_GANG_OLD_STEP
#define RTK_GANG_STEP_CALL uint64_t Ph_, Mh_; const int hout = rtk_myers_step(nPv, nMv, Eq, hin, bit, Ph_, Mh_);
#else
#define RTK_GANG_STEP_CALL const int hout = rtk_gang_step64(nPv, nMv, Eq, hin, bit_hi, bit_pos);
#endif
template <int PLAIN>
__device__ __forceinline__ void rtk_gang_sweep(const char* __restrict__ q, const char* __restrict__ stage, const uint64_t* __restrict__ peq, uint64_t* __restrict__ fin,
                                               const int32_t* __restrict__ nodes, bool iupac, int my_x, int side, int steps_max) {
    const int lane = rtk_lane();
    const bool has = my_x >= 0;
    const int32_t* nd = nodes + RTK_LVL_NODE * (has ? my_x : 0);
    const int q0 = nd[0], qm = nd[1], t0 = nd[2], tn = nd[3], foff = nd[6], kk = nd[7], L = nd[10], lane0 = nd[8 + side] & 0xFF;
    const int lh = tn / 2, rh = tn - lh, np = has ? (side ? rh : lh) : 0;
    const int W = (qm + 63) >> 6, last_bit = (qm - 1) & 63;
    const RtkBand band = rtk_band_nw(qm, tn, kk);
    const int dlo = band.dlo, dhi = band.dhi;
    const int Wa = has ? rtk_band_words(qm, np, dlo) : 0;
    const int li = lane - lane0;
    const unsigned char* __restrict__ const tp = reinterpret_cast<const unsigned char*>(stage) + (has ? (t0 + side * lh) + 32 * my_x + 16 * side + 8 : 0); // column 0 of this lane's pass
    const uint64_t* __restrict__ const pq = peq + (2ll * foff + static_cast<long long>(side) * 4 * W);
    uint64_t* __restrict__ const fpv = fin + (static_cast<long long>(foff) + static_cast<long long>(side) * 2 * W); uint64_t* __restrict__ const fmv = fpv + W;
    const int src = (has ? (li == 0 ? lane + L - 1 : lane - 1) : lane) << 2; // the lane above in the run (the run closes on itself)
    int w = li; bool live = has && w < Wa;
    uint64_t eqA = 0, eqC = 0, eqG = 0, eqT = 0, Pv = ~0ull, Mv = 0ull;
    int clo_w = 0x7fffffff, chi_w = -1, head_from = 0x7fffffff, bit = 63, bit_pos = 31; bool bit_hi = true;
#define RTK_GANG_LOAD_WORD()                                                                                              \
    {                                                                                                                    \
        eqA = pq[4ll * w]; eqC = pq[4ll * w + 1]; eqG = pq[4ll * w + 2]; eqT = pq[4ll * w + 3];                          \
        Pv = ~0ull; Mv = 0ull;                                                                                           \
        clo_w = rtk_band_clo(w, dlo, 1); chi_w = rtk_band_chi(w, W, np, dhi, 1);                                          \
        head_from = (w == 0) ? -0x7fffffff : 64 * w + dhi;                                                               \
        bit = (w == W - 1) ? last_bit : 63; bit_hi = bit >= 32; bit_pos = bit & 31;                                      \
    }
    if (live) RTK_GANG_LOAD_WORD()
    int hout_prev = 0;
#define RTK_GANG_LOAD4(c_, a0, a1, a2, a3)                                                                                \
    {                                                                                                                    \
        int ca = (c_); ca = ca < -4 ? -4 : ca; ca = ca > np ? np : ca; /* the 8 bytes on either side of a staged target are padding */ \
        const unsigned char* p4 = tp + ca;                                                                               \
        a0 = p4[0]; a1 = p4[1]; a2 = p4[2]; a3 = p4[3];                                                                  \
    }
#define RTK_GANG_STEP(tc_, cc_)                                                                                           \
    {                                                                                                                    \
        const int hraw = __builtin_amdgcn_ds_bpermute(src, hout_prev);                                                    \
        const int cc = (cc_);                                                                                            \
        const int hin = (cc >= head_from) ? 1 : hraw;                                                                    \
        const bool active = cc >= clo_w && cc <= chi_w;                                                                  \
        const unsigned tc = (tc_);                                                                                       \
        if (PLAIN) {                                                                                                     \
            /* 'A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54: bit 1 picks C/G over A/T, bit 2 picks T/G over A/C; bitwise selects, no branches */ \
            const uint64_t m1 = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(__builtin_amdgcn_sbfe(static_cast<int>(tc), 1, 1)))); /* (the builtin returns unsigned) */                 \
            const uint64_t m2 = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(__builtin_amdgcn_sbfe(static_cast<int>(tc), 2, 1))));                 \
            const uint64_t lo_ = (eqC & m1) | (eqA & ~m1), hi_ = (eqG & m1) | (eqT & ~m1);                               \
            const uint64_t Eq = (hi_ & m2) | (lo_ & ~m2);                                                                \
            uint64_t nPv = Pv, nMv = Mv;                                                                                 \
            RTK_GANG_STEP_CALL                                                                                            \
            Pv = active ? nPv : Pv; Mv = active ? nMv : Mv; hout_prev = active ? hout : hout_prev;                       \
        } else if (active) {                                                                                             \
            uint64_t Eq;                                                                                                 \
            if (tc == 'A') Eq = eqA; else if (tc == 'C') Eq = eqC; else if (tc == 'G') Eq = eqG; else if (tc == 'T') Eq = eqT; \
            else { /* rare: IUPAC code, N or foreign byte in the target */                                              \
                Eq = 0; const int lim2 = (qm - 64 * w) < 64 ? (qm - 64 * w) : 64;                                        \
                for (int i = 0; i < lim2; ++i) Eq |= static_cast<uint64_t>(rtk_chars_equal(rtk_job_qc(q + q0, qm, side, 64 * w + i), static_cast<unsigned char>(tc), iupac)) << i; \
            }                                                                                                            \
            uint64_t Ph, Mh;                                                                                             \
            hout_prev = rtk_myers_step(Pv, Mv, Eq, hin, bit, Ph, Mh);                                                    \
        }                                                                                                                \
    }
    for (int s0 = 0; s0 < steps_max; s0 += RTK_GANG_PERIOD) {
        { // words that are past their last column: park the deltas, take the next word of this lane's run
            const bool done = live && (s0 - w > chi_w);
            if (rtk_ballot(done) != 0ull) {
                if (done) {
                    fpv[w] = Pv; fmv[w] = Mv;
                    w += L; live = w < Wa;
                    if (live) RTK_GANG_LOAD_WORD() else { clo_w = 0x7fffffff; chi_w = -1; }
                }
            }
        }
        int c = s0 - w; // this lane's column at step s0
        unsigned a0, a1, a2, a3;
        RTK_GANG_LOAD4(c, a0, a1, a2, a3)
#pragma unroll 1
        for (int g = 0; g < RTK_GANG_PERIOD / 4; ++g) {
            unsigned b0, b1, b2, b3;
            RTK_GANG_LOAD4(c + 4, b0, b1, b2, b3) // the characters of the next four steps, in flight while these four are computed
            RTK_GANG_STEP(a0, c) RTK_GANG_STEP(a1, c + 1) RTK_GANG_STEP(a2, c + 2) RTK_GANG_STEP(a3, c + 3)
            c += 4; a0 = b0; a1 = b1; a2 = b2; a3 = b3;
        }
    }
#undef RTK_GANG_STEP
#undef RTK_GANG_LOAD4
#undef RTK_GANG_LOAD_WORD
    if (live) { fpv[w] = Pv; fmv[w] = Mv; }
}

// One gang sweep: the half passes of the round's sub-problems that carry gang id `gang`.
__device__ __noinline__ void rtk_myers_gang(const RtkGangCtx& C_, int gang_) {
    const char* const q = rtk_u(C_.q); const char* const t = rtk_u(C_.t); char* const stage = rtk_u(C_.stage);
    uint64_t* const peq = rtk_u(C_.peq); uint64_t* const fin = rtk_u(C_.fin); int32_t* const nodes = rtk_u(C_.nodes);
    const int n_nodes = rtk_u(C_.n_nodes), gang = rtk_u(gang_); const bool iupac = rtk_u(C_.iupac) != 0;
    const int lane = rtk_lane();
    int my_x = -1, my_side = 0, steps_max = 0; bool plain = true;
    for (int x = 0; x < n_nodes; ++x) {
        const int32_t* nd = nodes + RTK_LVL_NODE * x;
        const int L = rtk_ld(nd + 10);
        if (L == 0) continue;
        const int q0 = rtk_ld(nd), qm = rtk_ld(nd + 1), t0 = rtk_ld(nd + 2), tn = rtk_ld(nd + 3), foff = rtk_ld(nd + 6), kk = rtk_ld(nd + 7);
        const int lh = tn / 2, rh = tn - lh, W = (qm + 63) >> 6;
        const RtkBand band = rtk_band_nw(qm, tn, kk);
        for (int side = 0; side < 2; ++side) {
            const int lg = rtk_ld(nd + 8 + side);
            if ((lg >> 8) != gang) continue;
            const int lane0 = lg & 0xFF, np = side ? rh : lh;
            if (lane >= lane0 && lane < lane0 + L) { my_x = x; my_side = side; }
            // the target of the pass, the right half reversed, with 8 bytes of padding on both sides
            char* const dst = stage + (t0 + side * lh) + 32 * x + 16 * side + 8;
            const char* const src = t + t0 + side * lh;
            bool ok = true;
            for (int i = lane; i < np; i += RTK_WAVE) { const char ch = side ? src[np - 1 - i] : src[i]; dst[i] = ch; ok = ok && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
            if (lane < 8) { dst[lane - 8] = 'A'; dst[np + lane] = 'A'; }
            plain = plain && (rtk_ballot(!ok) == 0ull);
            // the profile of the query words that get a column (the right half pass reads the query backwards)
            const int Wa = rtk_band_words(qm, np, band.dlo);
            uint64_t* const pq = peq + (2ll * foff + static_cast<long long>(side) * 4 * W);
            for (int w = lane; w < Wa; w += RTK_WAVE) { uint64_t a, c, g, tt; rtk_myers_eq4(q + q0, qm, side, w, iupac, a, c, g, tt); pq[4ll * w] = a; pq[4ll * w + 1] = c; pq[4ll * w + 2] = g; pq[4ll * w + 3] = tt; }
            const int steps = np + Wa - 1;
            steps_max = steps > steps_max ? steps : steps_max;
        }
    }
    rtk_sync(); // staged characters and profile words are read by other lanes than wrote them
    if (plain) rtk_gang_sweep<1>(q, stage, peq, fin, nodes, iupac, my_x, my_side, steps_max);
    else rtk_gang_sweep<0>(q, stage, peq, fin, nodes, iupac, my_x, my_side, steps_max);
    rtk_sync();
}

#ifdef RTK_MULTIWAVE
__device__ __noinline__ void rtk_myers_leaf_item(RtkCoop* st, int wave, int x);
#endif

// The alignment of q against t (NW, distance `best` or unknown when < 0), level by level. Returns false when the problem does not need a split
// or the lists do not fit (nothing emitted: the caller walks depth first).
__device__ __noinline__ bool rtk_myers_alignment_lvl(const MyersScratch& sc, const char* q, int m, const char* t, int n, int best, bool iupac, uint32_t* n_moves, int* best_out) {
    if (rtk_hb_is_leaf(m, n)) return false;
#ifdef RTK_MULTIWAVE
    RtkCoop* st = rtk_coop();
    const int nwv = rtk_coop_ld(&st->n_waves);
#endif
    const int lane = rtk_lane();
    const long long W0 = (m + 63) >> 6;
    uint64_t* const tb = rtk_ld(&sc.tb);
    const uint64_t tbw = rtk_ld(&sc.tb_cap_words), fin_words = 4ull * (static_cast<uint64_t>(W0) + RTK_LVL_MAXN);
    if (tbw < fin_words + 64) return false;
    uint64_t cap = (tbw - fin_words) / 6; // two lists of `cap` sub-problems, 6 ints each, in the words behind the delta vectors
    if (cap > rtk_ld(&sc.r_cap) / 6u) cap = rtk_ld(&sc.r_cap) / 6u;
    if (cap < 8) return false;
    if (4ull * rtk_ld(&sc.t_cap) < static_cast<uint64_t>(n) + 32ull * RTK_LVL_MAXN + 64ull) return false;                 // staged targets (in the column-score array)
    if (15ull * rtk_ld(&sc.w_cap) < 8ull * (static_cast<uint64_t>(W0) + RTK_LVL_MAXN)) return false;                       // profile words of a round
    int32_t* cur = reinterpret_cast<int32_t*>(tb + fin_words); int32_t* nxt = cur + 6 * cap;
    int8_t* const carry = rtk_ld(&sc.carry); int32_t* const colscore = rtk_ld(&sc.colscore); int32_t* const rowL = rtk_ld(&sc.rowL); int32_t* const rowR = rtk_ld(&sc.rowR);
    int32_t* const nodes = rtk_ld(&sc.hstack);
    uint64_t* const peq = rtk_ld(&sc.peq);
    const uint64_t* const need = static_cast<uint64_t*>(rtk_ld(&sc.need_bm)); // (MyersScratch::need_bm: sub-problems nobody looks at are not solved)
    MyersScratch& prof = const_cast<MyersScratch&>(sc); const unsigned long long t_all0 = rtk_clock();
    RtkGangCtx gc; gc.q = q; gc.t = t; gc.stage = reinterpret_cast<char*>(colscore) + 8; gc.peq = peq; gc.fin = tb; gc.nodes = nodes; gc.n_nodes = 0; gc.iupac = iupac ? 1 : 0;
    if (lane == 0) { cur[0] = 0; cur[1] = m; cur[2] = 0; cur[3] = n; cur[4] = best; cur[5] = 0; }
    rtk_sync();
    int n_cur = 1;
    int k_top = rtk_band_guess(m, n); // band of the one problem whose distance is not known (the whole one, when best < 0): see rtk_myers_alignment
    for (;;) {
        bool redo = false;
        // ---- slots of the next list: a leaf keeps one, a sub-problem that is split gets two (its halves, in order) ----
        int n_next = 0, n_split = 0;
        for (int c0 = 0; c0 < n_cur; c0 += RTK_WAVE) {
            const int i = c0 + lane; const bool valid = i < n_cur;
            int e[5] = {0, 0, 0, 0, 0};
            if (valid) for (int k = 0; k < 5; ++k) e[k] = cur[6 * i + k];
            const bool leaf = valid && (rtk_hb_is_leaf(e[1], e[3]) || (need && e[4] >= 0 && rtk_need_none(need, e[2], e[2] + e[3])));
            int total; const int excl = rtk_wave_excl_scan(valid ? (leaf ? 1 : 2) : 0, &total);
            const int pos = n_next + excl;
            if (valid && static_cast<uint64_t>(pos) + 2 <= cap) { if (leaf) { for (int k = 0; k < 5; ++k) nxt[6 * pos + k] = e[k]; nxt[6 * pos + 5] = 0; } else cur[6 * i + 5] = pos; }
            n_next += rtk_u(total); n_split += rtk_popc(rtk_ballot(valid && !leaf));
        }
        if (static_cast<uint64_t>(n_next) > cap) return false; // (nothing emitted yet)
        rtk_sync();
        if (n_split == 0) break;
        // ---- the half passes of the sub-problems that are split, a round at a time ----
        int rn = 0, gang = 0, lanes_used = 0, n_bjobs = 0, b_items = 0; uint64_t fin_off = 0; bool bad = false;
        auto run_round = [&]() {
            const int n_gangs = lanes_used > 0 ? gang + 1 : gang;
            rtk_sync(); // the records of the round (written by lane 0)
            const unsigned long long t_p0 = rtk_clock();
            RtkGangCtx gcr = gc; gcr.n_nodes = rn;
            bool shared = false;
#ifdef RTK_MULTIWAVE
            if (nwv > 1 && n_gangs + b_items > 1) {
                if (lane == 0) { st->gctx = gcr; st->n_gangs = n_gangs; st->n_jobs = n_bjobs; st->n_items = n_gangs + b_items; st->first[n_bjobs] = b_items; }
                rtk_myers_round_coop(st, b_items);
                if (lane == 0) st->n_gangs = 0;
                shared = true;
            }
#endif
            if (!shared) {
                for (int g = 0; g < n_gangs; ++g) rtk_myers_gang(gcr, g);
                for (int x = 0; x < rn; ++x) { // sub-problems whose band is wider than a wave: banded row blocks, one pass after the other
                    if (rtk_ld(nodes + RTK_LVL_NODE * x + 10) != 0) continue;
                    const int q0 = rtk_ld(nodes + RTK_LVL_NODE * x), qm = rtk_ld(nodes + RTK_LVL_NODE * x + 1), t0 = rtk_ld(nodes + RTK_LVL_NODE * x + 2), tn = rtk_ld(nodes + RTK_LVL_NODE * x + 3);
                    const int foff = rtk_ld(nodes + RTK_LVL_NODE * x + 6), kk = rtk_ld(nodes + RTK_LVL_NODE * x + 7);
                    const int Wn = (qm + 63) >> 6, lh = tn / 2, rh = tn - lh;
                    const RtkBand band = rtk_band_nw(qm, tn, kk);
                    uint64_t* const fin = tb + foff;
                    RtkCoopJob j = rtk_make_job(sc, rtk_seq(q + q0, qm), rtk_seq(t + t0, lh), 1, iupac, band, fin, fin + Wn); j.ring = 0;
                    { const int nb = (rtk_band_words(qm, lh, band.dlo) + 63) >> 6; for (int b = 0; b < nb; ++b) rtk_myers_block(j, b, nullptr); }
                    j = rtk_make_job(sc, rtk_seq(q + q0, qm, 1), rtk_seq(t + t0 + lh, rh, 1), 1, iupac, band, fin + 2 * Wn, fin + 3 * Wn); j.ring = 0;
                    { const int nb = (rtk_band_words(qm, rh, band.dlo) + 63) >> 6; for (int b = 0; b < nb; ++b) rtk_myers_block(j, b, nullptr); }
                }
                rtk_sync();
            }
            const unsigned long long t_p1 = rtk_clock(); prof.hb_pass += t_p1 - t_p0;
            for (int x = 0; x < rn && !bad; ++x) {
                const int32_t* nd = nodes + RTK_LVL_NODE * x;
                const int q0 = rtk_ld(nd), qm = rtk_ld(nd + 1), t0 = rtk_ld(nd + 2), tn = rtk_ld(nd + 3), bs_in = rtk_ld(nd + 4), slot = rtk_ld(nd + 5), foff = rtk_ld(nd + 6), kk = rtk_ld(nd + 7), L = rtk_ld(nd + 10);
                const int Wn = (qm + 63) >> 6, lh = tn / 2, rh = tn - lh;
                const RtkBand band = rtk_band_nw(qm, tn, kk); const int gran = L != 0 ? 1 : 64;
                uint64_t* const fin = tb + foff;
                int32_t* const rl = rowL + q0; int32_t* const rr = rowR + q0;
                rtk_myers_column(fin, fin + Wn, qm, lh, rl, band.dlo, band.dhi, gran);
                rtk_myers_column(fin + 2 * Wn, fin + 3 * Wn, qm, rh, rr, band.dlo, band.dhi, gran);
                rtk_sync();
                int bs = bs_in;
                if (bs < 0) { // only the whole problem: the optimum = the smallest left + right sum over every split point
                    int mn = 0x7fffffff;
                    for (int b0 = 0; b0 + 1 < qm; b0 += RTK_WAVE) { const int qi = b0 + lane; if (qi + 1 < qm) { const int v = rl[qi] + rr[qm - 2 - qi]; mn = v < mn ? v : mn; } }
                    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(mn, o, 64); mn = v < mn ? v : mn; }
                    mn = rtk_u(mn);
                    const int e0 = lh + rtk_ld(rr + (qm - 1)), e1 = rtk_ld(rl + (qm - 1)) + rh;
                    mn = e0 < mn ? e0 : mn; mn = e1 < mn ? e1 : mn;
                    if (mn > kk) { k_top = mn; redo = true; continue; } // a real score above the guess: the level again, with the band of that score
                    bs = mn;
                    if (best_out) *best_out = mn;
                }
                int split = -2;
                for (int b0 = 0; b0 + 1 < qm && split == -2; b0 += RTK_WAVE) {
                    const int qi = b0 + lane;
                    const bool ok = (qi + 1 < qm) && (rl[qi] + rr[qm - 2 - qi] == bs);
                    const uint64_t bal = rtk_ballot(ok);
                    if (bal) split = b0 + rtk_ffs(bal) - 1;
                }
                int ls, rs;
                if (split >= 0) { ls = rtk_ld(rl + split); rs = rtk_ld(rr + (qm - 2 - split)); }
                else if (lh + rtk_ld(rr + (qm - 1)) == bs) { split = -1; ls = lh; rs = rtk_ld(rr + (qm - 1)); }
                else if (rtk_ld(rl + (qm - 1)) + rh == bs) { split = qm - 1; ls = rtk_ld(rl + (qm - 1)); rs = rh; }
                else { *sc.overflow = 2; bad = true; break; } // inconsistent optimum: cannot happen for a correct distance
                const int ul = split + 1;
                if (lane == 0) {
                    int32_t* a = nxt + 6 * slot;
                    a[0] = q0; a[1] = ul; a[2] = t0; a[3] = lh; a[4] = ls; a[5] = 0;
                    a[6] = q0 + ul; a[7] = qm - ul; a[8] = t0 + lh; a[9] = rh; a[10] = rs; a[11] = 0;
                }
            }
            prof.hb_split += rtk_clock() - t_p1;
            rtk_sync();
            rn = 0; gang = 0; lanes_used = 0; n_bjobs = 0; b_items = 0; fin_off = 0;
        };
        for (int c0 = 0; c0 < n_cur && !bad; c0 += RTK_WAVE) {
            const int i = c0 + lane; const bool valid = i < n_cur;
            int e[6] = {0, 0, 0, 0, 0, 0};
            if (valid) for (int k = 0; k < 6; ++k) e[k] = cur[6 * i + k];
            uint64_t todo = rtk_ballot(valid && !(rtk_hb_is_leaf(e[1], e[3]) || (need && e[4] >= 0 && rtk_need_none(need, e[2], e[2] + e[3]))));
            while (todo && !bad) {
                const int l = rtk_ffs(todo) - 1; todo &= todo - 1ull;
                const int q0 = rtk_shfl(e[0], l), qm = rtk_shfl(e[1], l), t0 = rtk_shfl(e[2], l), tn = rtk_shfl(e[3], l), bs = rtk_shfl(e[4], l), slot = rtk_shfl(e[5], l);
                const int Wn = (qm + 63) >> 6, lh = tn / 2, rh = tn - lh;
                if (lh == 0) { *sc.overflow = 1; bad = true; break; }
                const int kk = bs >= 0 ? bs : k_top;
                const RtkBand band = rtk_band_nw(qm, tn, kk);
                const int wal = rtk_band_words(qm, lh, band.dlo), war = rtk_band_words(qm, rh, band.dlo);
                long long bw = static_cast<long long>(band.dhi) - band.dlo; if (bw > (1 << 28)) bw = 1 << 28;
                int L = rtk_gang_lanes(static_cast<int>(bw), wal > war ? wal : war);
                if (L < 1) L = 1;
                int itl = 0, itr = 0;
                if (L > 64) { L = 0; itl = (wal + 63) >> 6; itr = (war + 63) >> 6; } // row blocks
#ifdef RTK_MULTIWAVE
                if (itl + itr > RTK_COOP_MAXB) { *sc.overflow = 1; bad = true; break; }
                if (rn == RTK_LVL_MAXN || (L == 0 && (n_bjobs + 2 > RTK_COOP_MAXJ || b_items + itl + itr > RTK_COOP_MAXB))) run_round();
#else
                if (rn == RTK_LVL_MAXN) run_round();
#endif
                if (bad) break;
                int laneL = 0, gangL = 0, laneR = 0, gangR = 0;
                if (L != 0) {
                    if (lanes_used + L > 64) { ++gang; lanes_used = 0; }
                    laneL = lanes_used; gangL = gang; lanes_used += L;
                    if (lanes_used + L > 64) { ++gang; lanes_used = 0; }
                    laneR = lanes_used; gangR = gang; lanes_used += L;
                }
                if (lane == 0) {
                    int32_t* nd = nodes + RTK_LVL_NODE * rn;
                    nd[0] = q0; nd[1] = qm; nd[2] = t0; nd[3] = tn; nd[4] = bs; nd[5] = slot; nd[6] = static_cast<int>(fin_off); nd[7] = kk;
                    nd[8] = laneL | (gangL << 8); nd[9] = laneR | (gangR << 8); nd[10] = L; nd[11] = n_bjobs;
#ifdef RTK_MULTIWAVE
                    if (L == 0) { // the two passes as row-block jobs of the workgroup
                        uint64_t* const fin = tb + fin_off;
                        RtkCoopJob j; j.qp = q + q0; j.tp = t + t0; j.m = qm; j.n = lh; j.qrev = 0; j.trev = 0; j.top_h = 1; j.iupac = iupac ? 1 : 0;
                        j.fin_pv = fin; j.fin_mv = fin + Wn; j.carry = carry + t0; j.colscore = colscore + t0;
                        j.dlo = band.dlo; j.dhi = band.dhi; j.ring = 0; j.peq4 = peq;
                        st->job[n_bjobs] = j;
                        j.tp = t + t0 + lh; j.n = rh; j.qrev = 1; j.trev = 1; j.fin_pv = fin + 2 * Wn; j.fin_mv = fin + 3 * Wn; j.carry = carry + t0 + lh; j.colscore = colscore + t0 + lh;
                        st->job[n_bjobs + 1] = j;
                        st->first[n_bjobs] = b_items; st->first[n_bjobs + 1] = b_items + itl;
                    }
#endif
                }
                if (L == 0) { n_bjobs += 2; b_items += itl + itr; }
                fin_off += 4ull * static_cast<uint64_t>(Wn); ++rn;
            }
        }
        if (rn && !bad) run_round();
        if (bad) { prof.hb_total += rtk_clock() - t_all0; return true; } // the overflow flag is set: the caller's caller retries or gives up, as with the depth-first driver
        if (redo) continue; // (only ever the first level: the one problem without a known distance)
        { int32_t* x = cur; cur = nxt; nxt = x; } n_cur = n_next;
    }
    // ---- leaf problems, in read order: the moves of leaf x go to moves + (sum of the query and target lengths of the leaves before it) and are
    //      moved together afterwards; in the multi-wave kernels every wave of the workgroup takes leaves (own traceback table each) ----
    for (int i = lane; i < 6 * n_cur; i += RTK_WAVE) rowL[i] = cur[i];
    rtk_sync();
    { int run = 0;
      for (int c0 = 0; c0 < n_cur; c0 += RTK_WAVE) {
          const int x = c0 + lane; const int len = x < n_cur ? rowL[6 * x + 1] + rowL[6 * x + 3] : 0;
          int total; const int excl = rtk_wave_excl_scan(len, &total);
          if (x < n_cur) { rowL[6 * x + 5] = run + excl; rowL[6 * x + 4] = -1; }
          run += rtk_u(total);
      } }
    rtk_sync();
    const unsigned long long t_l0 = rtk_clock();
    uint8_t* const mvs = rtk_ld(&sc.moves);
#ifdef RTK_MULTIWAVE
    MyersScratch* const lsc = reinterpret_cast<MyersScratch*>(rtk_u(reinterpret_cast<unsigned long long>(st->lsc))); // work areas of the waves for leaf tracebacks (nullptr: none)
    if (lsc && nwv > 1) {
        if (lane == 0) { st->lq = q; st->lt = t; st->lneed = need; st->liupac = iupac ? 1 : 0; st->llist = rowL; st->lmoves = mvs; st->n_items = n_cur; st->leaf_mode = 1; }
        rtk_myers_round_coop(st, 0);
        if (lane == 0) st->leaf_mode = 0;
        rtk_sync();
    }
#endif
    uint32_t total = 0;
    for (int x = 0; x < n_cur; ++x) {
        const int q0 = rtk_ld(rowL + 6 * x), qm = rtk_ld(rowL + 6 * x + 1), t0 = rtk_ld(rowL + 6 * x + 2), tn = rtk_ld(rowL + 6 * x + 3), off = rtk_ld(rowL + 6 * x + 5);
        const int len = rtk_ld(rowL + 6 * x + 4);
        if (len < 0) { // not done by a round (single wave, or too big for a helper's work area): here, straight to its final place (total <= off: the leaves behind stay intact)
            uint32_t nm = total;
            if (qm == 0 || tn == 0) { rtk_wfill(mvs + nm, qm == 0 ? 2 : 1, static_cast<uint64_t>(qm + tn)); nm += static_cast<uint32_t>(qm + tn); } // edlib.cpp:1171-1178
            else if (need && rtk_need_none(need, t0, t0 + tn)) { rtk_wfill(mvs + nm, 2, static_cast<uint64_t>(tn)); rtk_wfill(mvs + nm + tn, 1, static_cast<uint64_t>(qm)); nm += static_cast<uint32_t>(qm + tn); } // (MyersScratch::need_bm)
            else {
                const long long W = (qm + 63) >> 6;
                if (static_cast<uint64_t>(4 * W * tn) > tbw) { *sc.overflow = 1; break; }
                rtk_myers_traceback(sc, rtk_seq(q + q0, qm), rtk_seq(t + t0, tn), iupac, &nm);
            }
            if (rtk_ld(rtk_ld(&sc.overflow)) != 0) break;
            total = nm;
            continue;
        }
        if (static_cast<uint32_t>(off) != total && len > 0) { rtk_copy_lanes(mvs + total, mvs + off, static_cast<uint64_t>(len)); rtk_sync(); }
        total += static_cast<uint32_t>(len);
    }
    *n_moves += total;
    prof.hb_leaf += rtk_clock() - t_l0;
    prof.hb_total += rtk_clock() - t_all0;
    return true;
}

#endif
#endif
