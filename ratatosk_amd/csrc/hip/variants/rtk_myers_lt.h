// Path alignments of small problems with the traceback table in LDS ("LT": LDS table). Included by rtk_myers.h.
// (reference: src/edlib.cpp:141-296 edlibAlign with EDLIB_TASK_PATH; :945-1144 the traceback; what is computed is what rtk_myers_path computed
// before with its table in device memory: same sweep arithmetic, same walk priorities, same moves.)
//
// Why: the stored sweeps of the region stage wrote 16 bytes per 32-bit word and column into a table in device memory -- 8.5 GB per 64 Mb
// launch, written once and read along one path -- and the region kernel is bound half by its latency per wave and half by the bytes it
// moves through the L2 (k_regions without the table stores: 31.0 -> 27.5 ms; round 4). The walk only ever looks at one stretch of columns
// of one word at a time, and it moves monotonically towards the origin: so the table does not have to exist all at once.
//
// How: the sweep is the anti-diagonal pipeline of rtk_myers_fast32 (lane = 32-bit query word, step s: lane l works on column s - l).
// The steps are cut into chunks of CHS = 2^lg steps, CHS * W <= 512 entries = the 8 KB LDS buffer every wave owns (rtk_lds_set_buf):
// entry (lane, step) lives in slot step mod CHS of the lane's row, so the buffer always holds the last CHS steps (a ring). At every chunk
// boundary the complete pipeline state -- Pv, Mv, the horizontal delta and the character-select masks in flight in every lane, 16 bytes
// per lane -- is parked in device memory (one coalesced store per CHS steps). The walk starts in the chunk the sweep ended in; when it
// leaves a chunk through its first step, the chunk before is recomputed from its checkpoint into the same LDS rows (CHS steps, no table
// traffic). A walk crosses every chunk once, so a path alignment costs at most two sweeps of ALU work and no table bytes in HBM;
// problems of one chunk (the candidates a DFS call scores: ~100 x 150) cost one.
//
// MEASURED (round 4, configs[1], 64 Mb steps, both builds with global-address-space descriptor pointers): bit-identical results (golden vectors,
// correction parity), HBM write traffic of the table gone -- and k_regions 30.55 -> 31.45 ms: a problem of several chunks pays a second sweep of
// ALU work per wave (SQ_INSTS_VALU 11.1 -> 14.1 G per launch) and that costs more latency per wave than the table traffic did. The kernel is not
// bound by the bytes it moves. NOT the default: build with -DRTK_LT to get this route (profiles/scripts/build_variant.sh lt "-DRTK_LT").
//
// Not used under RTK_MULTIWAVE (the waves of those workgroups share the LDS buffer) and not in the simulator.
#ifndef RTK_MYERS_LT_H
#define RTK_MYERS_LT_H
#if !defined(RTK_SIM) && !defined(RTK_MULTIWAVE) && defined(RTK_LT)
#define RTK_HAVE_LT 1

#define RTK_LT_MAX_W 16 // 32-bit query words (m <= 512): chunks of at least 32 steps

__device__ __forceinline__ uint32_t* rtk_lds_set_buf(); // rtk_sets.h (ONE 8 KB buffer per wave for every user; nothing is kept in it across an alignment)
struct RtkLtCk { uint32_t pv, mv, x, pad; }; // pipeline state of one lane at a chunk boundary; x = (hout_prev + 1) | (m1 & 1) << 2 | (m2 & 1) << 3
__device__ __forceinline__ int rtk_lt_lg(int W) { return W <= 1 ? 9 : (W <= 2 ? 8 : (W <= 4 ? 7 : (W <= 8 ? 6 : 5))); }

// One alignment: sweep (+ last-row statistics when TRACK) and, when `walk_mode` >= 0, the traceback from cell (m, tn) with
// tn = n (walk_mode RTK_MODE_NW) or first SHW end + 1 (RTK_MODE_SHW), moves appended to sc.moves like rtk_myers_walk does.
// Returns false (nothing done that matters) when the target holds a character other than A/C/G/T: the caller takes the general route.
// `res`: the edlibAlign result of the mode asked for (distance, first / last end, their number).
template <int TRACK>
__device__ __forceinline__ bool rtk_lt_align(const MyersScratch& sc, const char* __restrict__ qp, int m, const char* __restrict__ tp, int n, bool iupac, int walk_mode,
                                             SweepStat* st_out, uint32_t* n_moves) {
    const int lane = rtk_lane();
    const int W = (m + 31) >> 5, last_bit = (m - 1) & 31;
    const int lg = rtk_lt_lg(W), CHS = 1 << lg, chm = CHS - 1, BLK = CHS < 64 ? CHS : 64;
    const int w = lane;
    const bool has_word = lane < W;
    const int top_h = 1;
    // ---- profile words of this lane (as rtk_myers_fast32)
    uint32_t eqA = 0, eqC = 0, eqG = 0, eqT = 0;
    if (has_word) {
        const int lim = (m - 32 * w) < 32 ? (m - 32 * w) : 32;
        uint64_t qw[4];
        for (int j = 0; j < 4; ++j) { uint64_t x = 0; if (8 * j < lim) __builtin_memcpy(&x, qp + 32 * w + 8 * j, 8); qw[j] = x; } // may read up to 7 bytes past the query inside its padded buffer
        uint32_t p1 = 0, p2 = 0, done = 0;
        for (int j = 0; j < 8; ++j) {
            if (4 * j >= lim) break;
            const uint32_t x = static_cast<uint32_t>(qw[j >> 1] >> (32 * (j & 1)));
            const uint32_t b1 = (x >> 1) & 0x01010101u, b2 = (x >> 2) & 0x01010101u;
            const uint32_t b12 = b1 & b2, b2n = b2 & ~b1;
            const uint32_t recon = 0x41414141u + (b1 << 1) + (b12 << 2) + (b2n << 4) + (b2n << 1) + b2n;
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
    RtkTbHalf* const lrow = reinterpret_cast<RtkTbHalf*>(rtk_lds_set_buf()) + (has_word ? (w << lg) : 0); // this lane's row of the ring
    RtkLtCk* const ck = reinterpret_cast<RtkLtCk*>(rtk_ld(&sc.tb));                                        // [chunk][lane]
    // one step of the pipeline; MASKED = 1 where some word has no column (fill, drain, and every recomputed step)
#define RTK_LT_STEP(MASKED, TRK)                                                                                                            \
    {                                                                                                                                        \
        const int s = c0 + j;                                                                                                                \
        const int in_t = __builtin_amdgcn_readlane(my_t, j);                                                                                 \
        const int hin = __builtin_amdgcn_update_dpp(top_h, hout_prev, 0x138, 0xF, 0xF, false);                                              \
        const uint32_t m1 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 1, 1), static_cast<int>(m1_prev), 0x138, 0xF, 0xF, false)); \
        const uint32_t m2 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(__builtin_amdgcn_sbfe(in_t, 2, 1), static_cast<int>(m2_prev), 0x138, 0xF, 0xF, false)); \
        const uint32_t lo_ = (eqC & m1) | (eqA & ~m1), hi_ = (eqG & m1) | (eqT & ~m1);                                                       \
        const uint32_t Eq = (hi_ & m2) | (lo_ & ~m2);                                                                                        \
        uint32_t nPv = Pv, nMv = Mv, Ph, Mh;                                                                                                 \
        const int hout = rtk_myers_step32(nPv, nMv, Eq, hin, bit, Ph, Mh);                                                                   \
        RtkTbHalf hv; hv.pv = nPv; hv.mv = nMv; hv.ph = Ph; hv.mh = Mh;                                                                      \
        if (MASKED) {                                                                                                                        \
            const int col = s - lane;                                                                                                        \
            const bool active = col >= 0 && col < n;                                                                                         \
            if (active) lrow[s & chm] = hv;                                                                                                  \
            Pv = active ? nPv : Pv; Mv = active ? nMv : Mv;                                                                                  \
            hout_prev = active ? hout : hout_prev;                                                                                           \
            score += active ? hout : 0;                                                                                                      \
        } else {                                                                                                                             \
            lrow[s & chm] = hv;                                                                                                              \
            Pv = nPv; Mv = nMv; hout_prev = hout; score += hout;                                                                             \
        }                                                                                                                                    \
        m1_prev = m1; m2_prev = m2;                                                                                                          \
        if (TRK) {                                                                                                                           \
            const int tcol = s - (W - 1);                                                                                                    \
            if (!(MASKED) || tcol >= 0) {                                                                                                    \
                const bool lt_ = score < vbest, eq_ = score == vbest;                                                                        \
                vbest = lt_ ? score : vbest; vfirst = lt_ ? tcol : vfirst; vlast = (lt_ || eq_) ? tcol : vlast; vcnt = lt_ ? 1 : (vcnt + (eq_ ? 1 : 0)); \
            }                                                                                                                                \
        }                                                                                                                                    \
    }
    // ---- forward sweep: ring in LDS, checkpoint at every chunk boundary
    bool plain = true;
    int nxt_t = 'A';
    if (lane < n && lane < BLK) nxt_t = static_cast<int>(static_cast<unsigned char>(tp[lane]));
    for (int c0 = 0; c0 < steps; c0 += BLK) {
        int my_t = nxt_t;
        { const int cn = c0 + BLK + lane; nxt_t = 'A'; if (cn < n && lane < BLK) nxt_t = static_cast<int>(static_cast<unsigned char>(tp[cn])); }
        if (rtk_ballot(!(my_t == 'A' || my_t == 'C' || my_t == 'G' || my_t == 'T')) != 0ull) { plain = false; break; }
        asm volatile("" : "+v"(my_t));
        if ((c0 & chm) == 0 && has_word) { RtkLtCk k_; k_.pv = Pv; k_.mv = Mv; k_.x = static_cast<uint32_t>(hout_prev + 1) | ((m1_prev & 1u) << 2) | ((m2_prev & 1u) << 3); k_.pad = 0; ck[static_cast<uint32_t>(c0 >> lg) * 64u + static_cast<uint32_t>(lane)] = k_; }
        const int lim = (steps - c0) < BLK ? (steps - c0) : BLK;
        int j_fill = (W - 1) - c0; j_fill = j_fill < 0 ? 0 : (j_fill > lim ? lim : j_fill);
        int j_full = n - c0; j_full = j_full < j_fill ? j_fill : (j_full > lim ? lim : j_full);
        if (has_word) { // (the lanes without a word sit the steps out: no activity test per step for them)
            int j = 0;
            for (; j < j_fill; ++j) RTK_LT_STEP(1, TRACK)
            for (; j < j_full; ++j) RTK_LT_STEP(0, TRACK)
            for (; j < lim; ++j) RTK_LT_STEP(1, TRACK)
        }
    }
    SweepStat st; st.plain = plain; st.final_score = m; st.best = 0x7fffffff; st.first = -1; st.last = -1; st.cnt = 0;
    if (!plain) { *st_out = st; return false; }
    st.final_score = __builtin_amdgcn_readlane(score, W - 1);
    if (TRACK) { st.best = __builtin_amdgcn_readlane(vbest, W - 1); st.first = __builtin_amdgcn_readlane(vfirst, W - 1); st.last = __builtin_amdgcn_readlane(vlast, W - 1); st.cnt = __builtin_amdgcn_readlane(vcnt, W - 1); }
    *st_out = st;
    if (walk_mode < 0) return true;
    // ---- where the walk starts (same bookkeeping as rtk_myers_path)
    int cur, tn;
    if (walk_mode == RTK_MODE_NW) { cur = st.final_score; tn = n; }
    else {
        int best = st.best; const bool pseudo = (m & 63) != 0;
        if (pseudo && m < best) best = m;
        const int first = (pseudo && m == best) ? -1 : st.first;
        cur = best; tn = first + 1;
    }
    if (tn <= 0) return true; // (the caller appends the m inserts of an empty target prefix itself, like rtk_myers_alignment does)
    // ---- walk: cell (i, j), 32-bit word w = (i - 1) >> 5, step of that cell = (j - 1) + w; priorities up (insert) > left (delete) > diagonal
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    int cur_q = (steps - 1) >> lg; // the chunk the ring holds
    const RtkTbHalf* const lbase = reinterpret_cast<const RtkTbHalf*>(rtk_lds_set_buf());
    uint8_t* const tmp = rtk_ld(&sc.moves_tmp); const uint32_t cap = rtk_ld(&sc.mv_cap);
    uint32_t nt = 0;
    int i = m, jj = tn;
    while (i > 0 && jj > 0) {
        const int r = i - 1, c = jj - 1, ww = r >> 5, b = r & 31, s = c + ww, q = s >> lg;
        if (q != cur_q) { // recompute chunk q from its checkpoint: steps [q << lg, (q + 1) << lg), all of them before the sweep's last step
            const RtkLtCk k_ = has_word ? ck[static_cast<uint32_t>(q) * 64u + static_cast<uint32_t>(lane)] : RtkLtCk{~0u, 0u, 1u, 0u};
            Pv = k_.pv; Mv = k_.mv; hout_prev = static_cast<int>(k_.x & 3u) - 1; m1_prev = 0u - ((k_.x >> 2) & 1u); m2_prev = 0u - ((k_.x >> 3) & 1u);
            for (int c0 = q << lg; c0 < ((q + 1) << lg); c0 += BLK) {
                const int cj = c0 + lane;
                int my_t = 'A'; if (cj < n && lane < BLK) my_t = static_cast<int>(static_cast<unsigned char>(tp[cj]));
                asm volatile("" : "+v"(my_t));
                if (has_word) { // (a block of steps in which every word has a column needs no activity tests)
                    if (c0 >= W - 1 && c0 + BLK <= n) { for (int j = 0; j < BLK; ++j) RTK_LT_STEP(0, 0) }
                    else { for (int j = 0; j < BLK; ++j) RTK_LT_STEP(1, 0) }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
            cur_q = q;
        }
        // window: lane l <-> column c - l of word ww, i.e. step s - l; valid while the column exists and the step is in the chunk
        const int nv_ = (c + 1) < (s - (q << lg) + 1) ? (c + 1) : (s - (q << lg) + 1);
        const int nv = nv_ < 64 ? nv_ : 64;
        RtkTbHalf e; e.pv = 0; e.mv = 0; e.ph = 0; e.mh = 0;
        if (lane < nv) e = lbase[(ww << lg) + ((s - lane) & chm)];
        { // run of inserts (moves up column c): consecutive rows from r downwards whose vertical delta is +1 = ones of Pv & ~Mv below bit b
            const uint32_t up = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(e.pv & ~e.mv)));
            if ((up >> b) & 1u) {
                const uint32_t nup = ~(up << (31 - b));
                const int run = nup ? __builtin_clz(nup) : 32; // >= 1, stays inside this 32-row word
                if (lane < run) tmp[cap - (nt + 1u + static_cast<uint32_t>(lane))] = 1;
                i -= run; cur -= run; nt += static_cast<uint32_t>(run);
                continue;
            }
        }
        { // run of deletes (moves left along row r): leading columns whose cell has no +1 vertical delta but a +1 horizontal one
            const int vd_ = static_cast<int>((e.pv >> b) & 1u) - static_cast<int>((e.mv >> b) & 1u);
            const int hd_ = static_cast<int>((e.ph >> b) & 1u) - static_cast<int>((e.mh >> b) & 1u);
            const bool isleft = lane < nv && vd_ != 1 && hd_ == 1;
            const uint64_t nleft = ~rtk_ballot(isleft);
            const int run = nleft ? __builtin_ctzll(nleft) : 64;
            if (run > 0) {
                if (lane < run) tmp[cap - (nt + 1u + static_cast<uint32_t>(lane))] = 2;
                jj -= run; cur -= run; nt += static_cast<uint32_t>(run);
                continue;
            }
        }
        { // run of diagonal moves: the lane holding column c - l looks at cell (r - l, c - l); it needs the vertical delta of (r - l, c - l - 1) from the
          // lane next to it, so the last valid lane and column 0 never take part (the single step below handles them)
            const int l = lane;
            const int rl = r - l;
            const bool valid = l + 1 < nv && rl >= 32 * ww && (c - l) >= 1;
            const int bb = rl & 31, bn = (bb + 1) & 31;
            const int vd_ = static_cast<int>((e.pv >> bb) & 1u) - static_cast<int>((e.mv >> bb) & 1u);
            const int hd_ = static_cast<int>((e.ph >> bb) & 1u) - static_cast<int>((e.mh >> bb) & 1u);
            const int vdn = static_cast<int>((e.pv >> bn) & 1u) - static_cast<int>((e.mv >> bn) & 1u); // my column, one row further down: what lane - 1 needs
            const int vdl = __shfl_down(vdn, 1, 64);
            const bool isdiag = valid && vd_ != 1 && hd_ != 1;
            const uint64_t ndm = ~rtk_ballot(isdiag);
            const int run = ndm ? __builtin_ctzll(ndm) : 64;
            if (run > 0) {
                const bool mine = l < run;
                const bool mism = (hd_ + vdl) != 0;
                const uint64_t mm = rtk_ballot(mine && mism);
                if (mine) tmp[cap - (nt + 1u + static_cast<uint32_t>(l))] = mism ? 3 : 0;
                cur -= rtk_popc(mm); i -= run; jj -= run; nt += static_cast<uint32_t>(run);
                continue;
            }
        }
        // single step at (r, c): lane 0 holds the cell, lane 1 (when valid) the column to its left
        const uint32_t a0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e.pv), 0)), a1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e.mv), 0));
        const uint32_t a2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e.ph), 0)), a3 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e.mh), 0));
        const int vd = static_cast<int>((a0 >> b) & 1u) - static_cast<int>((a1 >> b) & 1u);
        const int hd = static_cast<int>((a2 >> b) & 1u) - static_cast<int>((a3 >> b) & 1u);
        uint8_t mv;
        if (vd == 1) { mv = 1; --i; cur -= 1; }
        else if (hd == 1) { mv = 2; --jj; cur -= 1; }
        else {
            const int left = cur - hd;
            int diag;
            if (c == 0) diag = i - 1;
            else if (nv >= 2) {
                const uint32_t l0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e.pv), 1)), l1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e.mv), 1));
                diag = left - (static_cast<int>((l0 >> b) & 1u) - static_cast<int>((l1 >> b) & 1u));
            } else {
                // the column to the left belongs to the chunk before. The move out of this cell is diagonal for certain (neither +1 delta), so the cell's
                // value is the diagonal neighbour's plus the cost of the character pair: read the pair's equality from the profile word of this row's lane
                const int tc = static_cast<int>(static_cast<unsigned char>(tp[c]));
                const uint32_t sel = (static_cast<uint32_t>(tc) >> 1) & 3u; // 'A' 0, 'C' 1, 'T' 2, 'G' 3
                const uint32_t eqw = (sel & 2u) ? ((sel & 1u) ? eqG : eqT) : ((sel & 1u) ? eqC : eqA);
                const uint32_t eqr = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(eqw), ww));
                diag = cur - (((eqr >> b) & 1u) ? 0 : 1);
            }
            mv = (diag == cur) ? 0 : 3;
            --i; --jj; cur = diag;
        }
        ++nt; tmp[cap - nt] = mv;
    }
#undef RTK_LT_STEP
    if (i > 0) { rtk_wfill(tmp + (cap - nt - static_cast<uint32_t>(i)), 1, static_cast<uint64_t>(i)); nt += static_cast<uint32_t>(i); i = 0; }
    if (jj > 0) { rtk_wfill(tmp + (cap - nt - static_cast<uint32_t>(jj)), 2, static_cast<uint64_t>(jj)); nt += static_cast<uint32_t>(jj); jj = 0; }
    rtk_sync();
    rtk_wcopy(rtk_ld(&sc.moves) + *n_moves, tmp + (cap - nt), nt);
    *n_moves += nt;
    { MyersScratch& msc = const_cast<MyersScratch&>(sc); msc.walk_moves += nt; msc.walk_calls += 1; }
    return true;
}

// can this problem take the LDS-table route? (the sizes rtk_myers_path checks for its in-memory branch, plus the word limit)
__device__ __forceinline__ bool rtk_lt_fits(const MyersScratch& sc, int m, int n) {
    const long long W64 = (m + 63) >> 6;
    return m > 0 && n > 0 && ((m + 31) >> 5) <= RTK_LT_MAX_W && static_cast<uint32_t>(m + n) <= sc.mv_cap && static_cast<uint32_t>(n) <= sc.t_cap && static_cast<uint32_t>(m) <= sc.r_cap &&
           static_cast<uint64_t>(2u * 64u * ((static_cast<uint32_t>(n) + 80u) / 32u + 2u)) <= sc.tb_cap_words && (2LL * 8 + 4) * W64 * n + 8LL * n < 1024 * 1024;
}

#endif
#endif
