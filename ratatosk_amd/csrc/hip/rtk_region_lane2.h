// Second part of the lane-per-region program (included by rtk_region_lane.h): candidate selection, scoring, the bounded DFS, the path search,
// colour selection, SNP annotations, the `correct` lambda, the consensus of the two strands and the driver of one gap.
#ifndef RTK_REGION_LANE2_H
#define RTK_REGION_LANE2_H

// ------------------------------------------------------------------------------------------------ candidate selection (src/Alignment.cpp:3-147, 967-1015)
// handles at word offset list_w; strings are materialised into RL_SB_PATH
RTK_FN void rl_select_best(RlCtx& c, uint32_t list_w, uint32_t n, RlSrc ref, uint32_t ref_len, int mode, double cut, int* best_id, int* best_end) {
    double best = 0.0; int bid = -1, bend = -1;
    RL_CTX(c, RL_P_SELECT);
    for (uint32_t i = 0; i < n && !c.fail(); ++i) {
        const uint32_t sl = rl_rec_to_string(c, rl_ld(c, list_w + i), RL_SB_PATH);
        if (sl == 0xFFFFFFFFu) break;
        const uint32_t norm = (mode == RTK_MODE_NW) ? (sl > ref_len ? sl : ref_len) : sl;
        if (i == 0) {
            const RlAln a = rl_myers(c, rl_src_l(rl_sb(RL_SB_PATH)), static_cast<int>(sl), ref, static_cast<int>(ref_len), -1, mode, true, false, nullptr);
            best = static_cast<double>(a.dist) / static_cast<double>(norm); bend = a.first; bid = 0;
        } else {
            const int kk = static_cast<int>(best * static_cast<double>(norm) + 1.0); // G5: double -> int as edlibNewAlignConfig receives it
            const RlAln a = rl_myers(c, rl_src_l(rl_sb(RL_SB_PATH)), static_cast<int>(sl), ref, static_cast<int>(ref_len), kk, mode, true, false, nullptr);
            if (a.dist >= 0 && (static_cast<double>(a.dist) / static_cast<double>(norm)) < best) { best = static_cast<double>(a.dist) / static_cast<double>(norm); bend = a.first; bid = static_cast<int>(i); }
        }
    }
    if (mode != RTK_MODE_NW && cut > 0.0 && best > cut) { bid = -1; bend = -1; }
    *best_id = bid; *best_end = bend;
    RL_CTX_END(c);
}

// ------------------------------------------------------------------------------------------------ scoring (src/GraphTraversal.cpp:867-909, 722-772)
// the path string is in RL_SB_CAND (length sl)
RTK_DEV double rl_score_path(RlCtx& c, uint32_t sl, RlSrc ref, uint32_t ref_len, bool terminal) {
    double score = 0.0;
    if (sl != 0) {
        const RlSrc str1 = rl_src_l(rl_sb(RL_SB_CAND));
        if (terminal) { const RlAln a = rl_myers(c, str1, static_cast<int>(sl), ref, static_cast<int>(ref_len), -1, RTK_MODE_NW, true, false, nullptr); score = 1.0 - (static_cast<double>(a.dist) / static_cast<double>(sl)); }
        else if (sl >= ref_len) { const RlAln a = rl_myers(c, ref, static_cast<int>(ref_len), str1, static_cast<int>(sl), -1, RTK_MODE_HW, true, false, nullptr); score = 1.0 - (static_cast<double>(a.dist) / static_cast<double>(ref_len)); }
        else {
            const uint64_t cap = static_cast<uint64_t>(static_cast<double>(sl) * (1.0 + static_cast<double>(c.o()->weak_region_len_factor)));
            const uint32_t l_ref_len = ref_len < cap ? ref_len : static_cast<uint32_t>(cap);
            const RlAln a = rl_myers(c, str1, static_cast<int>(sl), ref, static_cast<int>(l_ref_len), -1, RTK_MODE_HW, true, false, nullptr);
            score = 1.0 - (static_cast<double>(a.dist) / static_cast<double>(sl));
        }
        score = score > 0.0 ? score : 0.0; score = score < 1.0 ? score : 1.0;
    }
    return score;
}

// quality string of a path (SHW path alignment against ref) into RL_SB_QUAL[0 .. sl); the path string is in RL_SB_CAND.
// use_saved: the stored sweep of this very path against this very window is still in the table (the DFS's first terminal candidate)
RTK_FN void rl_score_path_qual(RlCtx& c, uint32_t sl, RlSrc ref, uint32_t ref_len, double score_best, double score_second, bool use_saved) {
    const double score_comp = score_best * ((score_best == 0.0) ? 0.0 : (1.0 - (score_second / score_best)));
    uint32_t nm = 0, off = RL_MV_BYTES;
    const uint32_t mvb = rl_mvb(0);
    RL_CTX(c, RL_P_QUAL);
    if (use_saved && c.sv_valid && c.sv_m == sl && c.sv_n == ref_len && sl > 0 && ref_len > 0) {
        nm = rl_myers_walk(c, static_cast<int>(sl), c.sv_first + 1, mvb, &off);
        c.c_align += 1;
    } else {
        rl_align_path(c, rl_src_l(rl_sb(RL_SB_CAND)), static_cast<int>(sl), ref, static_cast<int>(ref_len), RTK_MODE_SHW, mvb, &off, &nm);
        c.sv_valid = 0; // the table now holds this sweep
    }
    if (c.fail()) { RL_CTX_END(c); return; }
    const char c_best = rtk_get_qual(score_best, 0, static_cast<uint64_t>(static_cast<int32_t>(c.o()->max_qual)));
    const char c_comp = rtk_get_qual(score_comp, static_cast<uint64_t>(static_cast<int32_t>(c.o()->out_qual)), static_cast<uint64_t>(static_cast<int32_t>(c.o()->max_qual)));
    if (sl > c.lim_str()) { rl_fail(c, RL_F_STR); RL_CTX_END(c); return; }
    // a base gets the best-score quality when it sits on an identical reference base in an M run; everything else the comparison quality.
    // The moves are walked in order, the quality of every query character written once.
    RlW w = rl_w_open(c, rl_sb(RL_SB_QUAL), 0);
    uint32_t qp = 0, rp = 0;
    for (uint32_t i = 0; i < nm; ++i) {
        const unsigned char mv = rl_ldb(c, mvb + off + i);
        if (mv == 0 || mv == 3) { const bool same = rl_ldb(c, rl_sb(RL_SB_CAND) + qp) == rl_get(c, ref, rp); rl_w_put(c, w, static_cast<unsigned char>(same ? c_best : c_comp)); ++qp; ++rp; }
        else if (mv == 1) { rl_w_put(c, w, static_cast<unsigned char>(c_comp)); ++qp; }
        else ++rp;
    }
    for (; qp < sl; ++qp) rl_w_put(c, w, static_cast<unsigned char>(c_comp)); // (an alignment without moves: zero-length target)
    rl_w_close(c, w);
    RL_CTX_END(c);
}

// ------------------------------------------------------------------------------------------------ colour memo (src/GraphTraversal.cpp:485-487)
// |colours(u) & all_pids| >= min_cov_vertices, all_pids = the sorted ids at RL_OFF_ALL; the count stops at the threshold
RTK_DEV uint32_t rl_shared_with_all(const RlCtx& c, const uint32_t* a, uint32_t na, uint32_t cap, uint32_t* lo_io) {
    uint32_t cnt = 0, lo = *lo_io; const uint32_t nb = c.n_all();
    for (uint32_t i = 0; i < na && cnt < cap && lo < nb; ++i) {
        const uint32_t x = a[i];
        uint32_t l = lo, h = nb;
        while (l < h) { const uint32_t md = (l + h) >> 1; if (rl_ld(c, RL_OFF_ALL + md) < x) l = md + 1; else h = md; }
        lo = l;
        if (lo < nb && rl_ld(c, RL_OFF_ALL + lo) == x) { ++cnt; ++lo; }
    }
    return cnt;
}
RTK_FN bool rl_colour_ok(RlCtx& c, uint32_t u) {
    RL_ENTER(c);
    for (uint32_t i = 0; i < c.memo_n(); ++i) { const uint32_t e = rl_ld(c, RL_OFF_MEMO + i); if ((e >> 1) == u) { RL_LEAVE(c, RL_P_COLOUR_OK); return (e & 1u) != 0; } }
    const uint32_t mcv = c.o()->min_cov_vertices;
    bool ok = c.n_all() == 0;
    const GraphView& g = *c.g();
    if (!ok) {
        uint32_t shared = 0;
        const int32_t gi = g.gid.get()[u];
        if (gi >= 0) { const uint64_t* go = g.goff.get() + gi; uint32_t lo = 0; shared = rl_shared_with_all(c, g.col.get() + go[0], static_cast<uint32_t>(go[1] - go[0]), mcv, &lo); }
        if (shared < mcv) { const uint64_t* lo_ = g.loff.get() + u; uint32_t lo = 0; shared += rl_shared_with_all(c, g.col.get() + lo_[0], static_cast<uint32_t>(lo_[1] - lo_[0]), mcv - shared, &lo); }
        ok = shared >= mcv;
    }
    c.c_colour += g.card.get()[u] + c.n_all();
    if (c.memo_n() < RL_MEMO_CAP) { rl_st(c, RL_OFF_MEMO + c.memo_n(), (u << 1) | (ok ? 1u : 0u)); ++c.memo_n(); }
    RL_LEAVE(c, RL_P_COLOUR_OK);
    return ok;
}

RTK_DEV int rl_nb_successors(const RlCtx& c, const UMap& um) {
    const uint32_t* a = c.g()->adj.get() + 8ull * um.unitig + (um.strand ? 0 : 4);
    int n = 0; for (int b = 0; b < 4; ++b) n += (a[b] != RTK_NONE32) ? 1 : 0; return n;
}

// ------------------------------------------------------------------------------------------------ DFS (src/GraphTraversal.cpp:456-587)
// The program of rtk_explore_subgraph (rtk_region.h), lane by lane: terminal candidates in the list at RL_OFF_T, non-terminal ones at RL_OFF_NT
// (records of the DFS-level arena). Lazy non-terminal paths and the pruned first walk as there (same results as the reference's walk: the
// comment in rtk_region.h gives the argument, the parity tests hold it).
struct RlDfsOut { uint32_t n_t, n_nt; double t1, nt1, nt2; uint32_t nt_score_deferred, nt_qual_deferred; };
struct RlNtPending { uint32_t e; double nt1, nt2; uint32_t score_deferred, qual_deferred; };

RTK_FN RlDfsOut rl_explore_subgraph(RlCtx& c, RlSrc ref, uint32_t ref_len, uint32_t max_len_path, const UMap& um, const UMap& um_e, uint32_t level) {
    RlDfsOut out; out.n_t = 0; out.n_nt = 0; out.t1 = 0.0; out.nt1 = 0.0; out.nt2 = 0.0; out.nt_score_deferred = 0; out.nt_qual_deferred = 0;
    double score_t1 = 0.0, score_nt1 = 0.0, score_t2 = 0.0, score_nt2 = 0.0;
    uint32_t n_t = 0, n_nt = 0;
    RL_CTX(c, RL_P_DFS);
    c.top(2) = 0;
    uint32_t sp = 0;
    rl_st(c, RL_OFF_STK, RL_NOH); rl_st(c, RL_OFF_STK + 1, level); sp = 1;
    const uint32_t W2 = 2; // working path of the DFS
    const bool has_end = !rtk_um_is_empty(um_e);
    const bool lazy_nt = has_end && !(static_cast<double>(c.o()->min_score) > 0.0);
    const uint32_t* const g_adj = c.g()->adj.get();
    uint32_t n_nt_live = 0, n_t_scored = 0, n_pruned = 0, n_tc = 0;
    c.sv_valid = 0;
    const bool rev_a3 = static_cast<uint32_t>(c.o()->a3_strand_order) != 0;
    for (int walk = 0; walk < 2 && !c.fail(); ++walk) {
        const bool prune = lazy_nt && walk == 0, do_terminal = walk == 0;
        if (walk == 1) { if (!(lazy_nt && n_nt_live > 0 && n_pruned > 0)) break; n_nt = 0; n_nt_live = 0; rl_st(c, RL_OFF_STK, RL_NOH); rl_st(c, RL_OFF_STK + 1, level); sp = 1; }
        while (sp > 0 && !c.fail()) {
            --sp;
            const uint32_t hp = rl_ld(c, RL_OFF_STK + 2u * sp), lvl = rl_ld(c, RL_OFF_STK + 2u * sp + 1u);
            const UMap um_start = (hp == RL_NOH) ? um : rl_rec_back(c, hp);
            const uint32_t* adj = g_adj + 8ull * um_start.unitig + (um_start.strand ? 0 : 4);
            ++c.c_expand;
            const uint32_t a4[4] = { adj[0], adj[1], adj[2], adj[3] };
            const uint32_t eb = (rl_flags(c, um_start.unitig) >> (um_start.strand ? 4 : 0)) & 0xFu; // UnitigData::getSharedPids (UnitigData.hpp:275-284)
            const bool rev_order = rev_a3 && !um_start.strand; // [A3] switch
            for (int bi = 0; bi < 4 && !c.fail(); ++bi) {
                const int b = rev_order ? 3 - bi : bi;
                const uint32_t ab = a4[b];
                if (ab == RTK_NONE32) continue;
                UMap sc; sc.unitig = ab >> 1; sc.strand = ab & 1u; sc.dist = 0; sc.len = rl_nkm(c, sc.unitig);
                const bool col_ok = rl_colour_ok(c, sc.unitig);
                if (!(((eb >> b) & 1u) && col_ok)) continue;
                if (do_terminal && has_end && sc.unitig == um_e.unitig && um_e.strand == sc.strand) { // terminal: kept as a candidate, scored after the walk
                    if (hp == RL_NOH) rl_wp_clear(c, W2); else rl_wp_load(c, W2, hp);
                    UMap pref = sc;
                    if (pref.strand) { pref.dist = 0; pref.len = um_e.dist + 1; } else { pref.dist = um_e.dist; pref.len = sc.len - um_e.dist; }
                    rl_wp_extend(c, W2, pref);
                    if (!c.fail() && rl_p_l(c, rl_wp(W2)) <= max_len_path) {
                        if (n_tc >= c.lim_list()) { rl_fail(c, RL_F_LIST); break; }
                        rl_st(c, RL_OFF_TC + n_tc, rl_wp_commit(c, W2, 2)); ++n_tc;
                    }
                }
                if (c.fail()) break;
                { // non-terminal
                    if (prune) { // length of the extension (Path::extend, Path.hpp:319-330) before building it
                        const uint32_t l_new = (hp == RL_NOH) ? (sc.len + c.k() - 1u) : (rl_p_l(c, rl_h_w(hp)) + sc.len);
                        if (l_new > max_len_path) { ++n_pruned; continue; }
                    }
                    if (hp == RL_NOH) rl_wp_clear(c, W2); else rl_wp_load(c, W2, hp);
                    rl_wp_extend(c, W2, sc);
                    if (c.fail()) break;
                    const bool deeper = lvl != 0; // exploreSubGraph descends `level` unitigs (:531-535); (pass 2 is not a lane program)
                    if (deeper) {
                        if (sp + 1 > RL_STK_CAP || 2u * (sp + 1u) > 2u * c.lim_list()) { rl_fail(c, RL_F_LIST); break; }
                        rl_st(c, RL_OFF_STK + 2u * sp, rl_wp_commit(c, W2, 2)); rl_st(c, RL_OFF_STK + 2u * sp + 1u, lvl - 1u); ++sp;
                    } else if (rl_nb_successors(c, sc) > 0) {
                        if (lazy_nt) { // candidate kept in discovery order, scored after the walk (or never)
                            if (n_nt >= c.lim_list()) { rl_fail(c, RL_F_LIST); break; }
                            rl_st(c, RL_OFF_NT + n_nt, rl_wp_commit(c, W2, 2)); ++n_nt;
                            if (rl_p_l(c, rl_wp(W2)) + um.len < max_len_path) ++n_nt_live;
                        } else {
                            const uint32_t sl = rl_wp_to_string(c, W2, RL_SB_CAND);
                            if (sl == 0xFFFFFFFFu) break;
                            const double sco = rl_score_path(c, sl, ref, ref_len, false);
                            if (c.fail()) break;
                            if (sco >= score_nt1) {
                                if (sco > score_nt1) n_nt = 0;
                                if (n_nt >= c.lim_list()) { rl_fail(c, RL_F_LIST); break; }
                                rl_st(c, RL_OFF_NT + n_nt, rl_wp_commit(c, W2, 2)); ++n_nt;
                                score_nt2 = score_nt1; score_nt1 = sco;
                            } else if (sco > score_nt2) score_nt2 = sco;
                        }
                    }
                }
            }
        }
    }
    // The terminal candidates, in the order the walk met them (src/GraphTraversal.cpp:505-523: the reference scores each where it meets it; no score feeds
    // back into the walk, so scoring them here, all lanes of the wave together, gives the same survivors). The first candidate -- usually the only one -- is
    // scored by a stored sweep that its quality string is read from afterwards.
    for (uint32_t i = 0; i < n_tc && !c.fail(); ++i) {
        const uint32_t hc = rl_ld(c, RL_OFF_TC + i);
        const uint32_t sl = rl_rec_to_string(c, hc, RL_SB_CAND);
        if (sl == 0xFFFFFFFFu) break;
        double sco;
        ++n_t_scored;
        if (n_t_scored == 1 && sl != 0 && ref_len != 0 && static_cast<uint32_t>((sl + 63u) >> 6) * ref_len <= c.lim_tb() && sl <= 64u * RL_MAXW) {
            int32_t nw = -1;
            const RlAln a = rl_myers(c, rl_src_l(rl_sb(RL_SB_CAND)), static_cast<int>(sl), ref, static_cast<int>(ref_len), -1, RTK_MODE_SHW, true, true, &nw);
            if (c.fail()) break;
            c.sv_valid = 1; c.sv_m = sl; c.sv_n = ref_len; c.sv_nw = nw; c.sv_best = a.dist; c.sv_first = a.first;
            sco = 1.0 - (static_cast<double>(nw) / static_cast<double>(sl));
            sco = sco > 0.0 ? sco : 0.0; sco = sco < 1.0 ? sco : 1.0;
        } else sco = rl_score_path(c, sl, ref, ref_len, true);
        if (c.fail()) break;
        if (sco >= score_t1) {
            if (sco > score_t1) n_t = 0;
            rl_st(c, RL_OFF_T + n_t, hc); ++n_t; // n_t <= i + 1 <= lim_list
            score_t2 = score_t1; score_t1 = sco;
        } else if (sco > score_t2) score_t2 = sco;
    }
    bool nt_score_deferred = false;
    if (lazy_nt && !c.fail()) {
        if (n_nt_live == 0) n_nt = 0;
        if (n_nt == 1) nt_score_deferred = true; // nobody to compare it with: scored by the caller if the path is ever extended
        else if (n_nt > 1) { // the reference's bookkeeping (:540-549) over the candidates in discovery order
            const uint32_t n_cand = n_nt; n_nt = 0;
            for (uint32_t i = 0; i < n_cand && !c.fail(); ++i) {
                const uint32_t hc = rl_ld(c, RL_OFF_NT + i);
                const uint32_t sl = rl_rec_to_string(c, hc, RL_SB_CAND);
                if (sl == 0xFFFFFFFFu) break;
                const double sco = rl_score_path(c, sl, ref, ref_len, false);
                if (sco >= score_nt1) {
                    if (sco > score_nt1) n_nt = 0;
                    rl_st(c, RL_OFF_NT + n_nt, hc); ++n_nt; // n_nt <= i: survivors move towards the front
                    score_nt2 = score_nt1; score_nt1 = sco;
                } else if (sco > score_nt2) score_nt2 = sco;
            }
        }
    }
    // qualities (:556-584): every surviving path is committed again with its quality string (non-terminal ones: left to the caller when lazy)
    for (int which = 0; which < (lazy_nt ? 1 : 2) && !c.fail(); ++which) {
        const uint32_t lw = which ? RL_OFF_NT : RL_OFF_T; const uint32_t nL = which ? n_nt : n_t;
        for (uint32_t i = 0; i < nL && !c.fail(); ++i) {
            rl_wp_load(c, W2, rl_ld(c, lw + i));
            const uint32_t sl = rl_wp_to_string(c, W2, RL_SB_CAND);
            if (sl == 0xFFFFFFFFu) break;
            rl_score_path_qual(c, sl, ref, ref_len, which ? score_nt1 : score_t1, which ? score_nt2 : score_t2, which == 0 && n_t_scored == 1);
            if (c.fail()) break;
            if (sl == rl_p_l(c, rl_wp(W2))) { rl_copy_words(c, rl_wp(W2) + 4u + 3u * RL_UM_CAP, RL_OFF_STR + RL_SB_QUAL * RL_STR_W, (sl + 3u) >> 2); rl_st(c, rl_wp(W2) + 2, sl); } // Path::setQuality only accepts q.length() == l
            rl_st(c, lw + i, rl_wp_commit(c, W2, 2));
        }
    }
    out.n_t = n_t; out.n_nt = n_nt; out.t1 = score_t1; out.nt1 = score_nt1; out.nt2 = score_nt2;
    out.nt_score_deferred = nt_score_deferred ? 1u : 0u; out.nt_qual_deferred = (lazy_nt && n_nt != 0) ? 1u : 0u;
    RL_CTX_END(c);
    return out;
}

// explore() (src/GraphTraversal.cpp:41-93, 251-304). hp = record of the BFS level. Results stay in the lists at RL_OFF_T / RL_OFF_NT.
RTK_DEV void rl_explore(RlCtx& c, RlSrc ref, uint32_t ref_len, const UMap& um_e, uint32_t hp, uint32_t max_len_path, uint32_t* n_t, uint32_t* n_nt, RlNtPending* pend) {
    *n_t = 0; *n_nt = 0; pend->e = 0; pend->nt1 = 0.0; pend->nt2 = 0.0; pend->score_deferred = 0; pend->qual_deferred = 0;
    const UMap um = rl_rec_back(c, hp);
    const uint32_t path_len = rl_p_l(c, rl_h_w(hp));
    const uint32_t k = c.k();
    const bool non_empty_path = (path_len > (um.len + k - 1u)) && !rtk_um_is_empty(um);
    const uint32_t path_len_prefix = non_empty_path ? (path_len - um.len - k + 1u) : 0u;
    uint32_t end_pos_ref = 0;
    if (non_empty_path) {
        const uint32_t sl = rl_rec_to_string(c, hp, RL_SB_PATH);
        if (sl == 0xFFFFFFFFu) return;
        const RlAln a = rl_myers(c, rl_src_l(rl_sb(RL_SB_PATH)), static_cast<int>(path_len_prefix), ref, static_cast<int>(ref_len), -1, RTK_MODE_SHW, true, false, nullptr);
        if (c.fail()) return;
        end_pos_ref = static_cast<uint32_t>(a.first + 1);
    }
    if ((ref_len - end_pos_ref) != 0 && path_len < max_len_path) {
        RlDfsOut o = rl_explore_subgraph(c, rl_src_add(ref, end_pos_ref), ref_len - end_pos_ref, max_len_path - path_len_prefix, um, um_e, 3);
        if (c.fail()) return;
        const double min_score = c.o()->min_score;
        if (o.n_t && o.t1 < min_score) o.n_t = 0;
        if (o.n_nt && !o.nt_score_deferred && o.nt1 < min_score) o.n_nt = 0;
        pend->e = end_pos_ref; pend->nt1 = o.nt1; pend->nt2 = o.nt2; pend->score_deferred = o.nt_score_deferred; pend->qual_deferred = o.nt_qual_deferred;
        if (o.n_nt > 1) {
            int bid, bend;
            rl_select_best(c, RL_OFF_NT, o.n_nt, rl_src_add(ref, end_pos_ref), ref_len - end_pos_ref, RTK_MODE_HW, -1.0, &bid, &bend);
            if (c.fail()) return;
            rl_st(c, RL_OFF_NT, rl_ld(c, RL_OFF_NT + static_cast<uint32_t>(bid))); o.n_nt = 1;
        }
        *n_t = o.n_t; *n_nt = o.n_nt;
    }
}

// P (+) Q: working path wi extended by every mapping of the record hsub with its quality slice (src/GraphTraversal.cpp:379-390)
RTK_DEV void rl_extend_by(RlCtx& c, uint32_t wi, uint32_t hsub) {
    const uint32_t sw = rl_h_w(hsub); const uint32_t n = rl_p_n(c, sw), ql = rl_p_qlen(c, sw); const uint32_t qb = rl_rec_qb(c, hsub);
    uint32_t j = 0;
    for (uint32_t i = 0; i < n && !c.fail(); ++i) {
        const UMap um = rl_um_ld(c, sw + 4u + 3u * i);
        const uint32_t want = um.len + c.k() - 1u;
        uint32_t qn = 0;
        if (j <= ql) qn = (ql - j) < want ? (ql - j) : want; // std::string::substr clamps
        rl_wp_extend_q(c, wi, um, qb + j, qn);
        j += um.len;
    }
}

RTK_DEV UMap rl_start_suffix(const RlCtx& c, const UMap& um_s) { // src/GraphTraversal.cpp:113-125, 325-338
    UMap t = um_s;
    if (t.strand) { t.dist += t.len - 1u; t.len = rl_nkm(c, um_s.unitig) - t.dist; } else { t.len = um_s.dist + 1u; t.dist = 0; }
    return t;
}

RTK_DEV bool rl_path_has_short_cycle(const RlCtx& c, uint32_t h) {
    const uint32_t pw = rl_h_w(h); const uint32_t n = rl_p_n(c, pw);
    for (uint32_t i = 0; i < n; ++i) if (rl_flags(c, rl_ld(c, pw + 4u + 3u * i) >> 1) & RTK_F_SHORT_CYCLE) return true;
    return false;
}

// explorePathsBFS2 (src/GraphTraversal.cpp:212-454) between two anchors. The queue never holds more than one path (a pop pushes at most one).
// Returns the record (BFS level) of the single resulting path or RL_NOH.
RTK_FN uint32_t rl_explore_paths(RlCtx& c, RlSrc ref, uint32_t ref_len, const UMap& um_s, const UMap& um_e) {
    const uint32_t k = c.k();
    RL_CTX(c, RL_P_SEARCH);
    c.top(1) = 0; c.memo_n() = 0;
    uint32_t nv = 0, nvt = 0;
    const char q_max = rtk_get_qual(1.0, 0, static_cast<uint64_t>(static_cast<int32_t>(c.o()->max_qual)));
    const bool ok_start = !rtk_um_is_empty(um_s) && ((rl_flags(c, um_s.unitig) & RTK_F_EDGE_MASK) != 0);
    const bool ok_end = !rtk_um_is_empty(um_e) && ((rl_flags(c, um_e.unitig) & RTK_F_EDGE_MASK) != 0);
    if (ok_start && ok_end) {
        const uint32_t level = 4;
        uint64_t mn, mx; rtk_min_max_len(ref_len - k, c.o()->weak_region_len_factor, &mn, &mx);
        const uint32_t min_len_path = static_cast<uint32_t>(mn) + k;
        const uint32_t max_len_path = static_cast<uint32_t>(mx > 10 ? mx : 10) + k;
        const uint32_t max_paths = 1024;
        const uint32_t W1 = 1;
        const UMap ust = rl_start_suffix(c, um_s);
        if (um_s.unitig == um_e.unitig && um_s.strand == um_e.strand && ust.dist <= um_e.dist) { // :340-358
            const uint32_t len = (ust.len + k - 1u) - (um_e.strand ? (rl_ulen(c, um_e.unitig) - um_e.dist - k) : um_e.dist);
            if (len >= min_len_path && len <= max_len_path) {
                UMap bt = ust;
                if (bt.strand) bt.len = um_e.dist - bt.dist + 1u; else { bt.dist = um_e.dist; bt.len -= um_e.dist; }
                rl_wp_start(c, W1, bt, q_max);
                if (!c.fail()) { rl_st(c, RL_OFF_V + nv, rl_wp_commit(c, W1, 1)); ++nv; }
            }
        }
        rl_wp_start(c, W1, ust, q_max);
        uint32_t qh = c.fail() ? RL_NOH : rl_wp_commit(c, W1, 1); bool q_has = !c.fail();
        bool q_pending = false; uint32_t pend_hp = 0, pend_hq = 0; RlNtPending pend; pend.e = 0; pend.nt1 = 0.0; pend.nt2 = 0.0; pend.score_deferred = 0; pend.qual_deferred = 0;
        while (q_has && !c.fail()) {
            if (q_pending) { // the pop of src/GraphTraversal.cpp:364-366: only a path shorter than max_len_path is ever looked at again
                q_pending = false;
                const uint32_t qw = rl_h_w(pend_hq); const uint32_t qn = rl_p_n(c, qw);
                uint32_t l_ext = rl_p_l(c, rl_h_w(pend_hp));
                for (uint32_t i = 0; i < qn; ++i) l_ext += rl_ld(c, qw + 4u + 3u * i + 2u); // Path::extend adds um.len per unitig
                if (!(l_ext < max_len_path)) break;
                const uint32_t W2 = 2;
                rl_wp_load(c, W2, pend_hq);
                const uint32_t sl = rl_wp_to_string(c, W2, RL_SB_CAND);
                if (sl == 0xFFFFFFFFu) break;
                double nt1 = pend.nt1; const double nt2 = pend.nt2;
                if (pend.score_deferred) nt1 = rl_score_path(c, sl, rl_src_add(ref, pend.e), ref_len - pend.e, false);
                if (c.fail()) break;
                rl_score_path_qual(c, sl, rl_src_add(ref, pend.e), ref_len - pend.e, nt1, nt2, false);
                if (c.fail()) break;
                if (sl == rl_p_l(c, rl_wp(W2))) { rl_copy_words(c, rl_wp(W2) + 4u + 3u * RL_UM_CAP, RL_OFF_STR + RL_SB_QUAL * RL_STR_W, (sl + 3u) >> 2); rl_st(c, rl_wp(W2) + 2, sl); }
                const uint32_t hq = rl_wp_commit(c, W2, 1);
                if (c.fail()) break;
                rl_wp_load(c, W1, pend_hp); rl_extend_by(c, W1, hq);
                if (c.fail()) break;
                qh = rl_wp_commit(c, W1, 1);
                if (c.fail()) break;
            }
            const uint32_t hp = qh; q_has = false;
            if (rl_p_l(c, rl_h_w(hp)) < max_len_path) {
                uint32_t n_t, n_nt;
                rl_explore(c, ref, ref_len, um_e, hp, max_len_path, &n_t, &n_nt, &pend);
                if (c.fail()) break;
                for (uint32_t i = 0; i < n_t && !c.fail(); ++i) {
                    rl_wp_load(c, W1, hp); rl_extend_by(c, W1, rl_ld(c, RL_OFF_T + i));
                    if (c.fail()) break;
                    if (nvt >= c.lim_list()) { rl_fail(c, RL_F_LIST); break; }
                    rl_st(c, RL_OFF_VT + nvt, rl_wp_commit(c, W1, 1)); ++nvt;
                }
                for (uint32_t i = 0; i < n_nt && !c.fail(); ++i) {
                    const uint32_t hs = rl_ld(c, RL_OFF_NT + i);
                    if (rl_p_n(c, rl_h_w(hs)) == level) { // :395
                        if (pend.qual_deferred) { // keep what is needed to finish Q when (if) the entry is popped: its unitigs move to the BFS-level arena
                            rl_wp_load(c, 2, hs);
                            pend_hq = rl_wp_commit(c, 2, 1); pend_hp = hp; q_pending = true; q_has = true;
                        } else {
                            rl_wp_load(c, W1, hp); rl_extend_by(c, W1, hs);
                            if (c.fail()) break;
                            qh = rl_wp_commit(c, W1, 1); q_has = true;
                        }
                    }
                }
                if (nvt >= max_paths) { rl_fail(c, RL_F_LIST); break; } // (never: the list capacity is far below)
            }
        }
        if (!c.fail()) { // final flush
            for (uint32_t i = 0; i < nvt && !c.fail(); ++i) {
                const uint32_t h = rl_ld(c, RL_OFF_VT + i); const uint32_t l = rl_p_l(c, rl_h_w(h));
                if (l >= min_len_path && l <= max_len_path) { if (nv >= c.lim_list()) { rl_fail(c, RL_F_LIST); break; } rl_st(c, RL_OFF_V + nv, h); ++nv; }
            }
        }
    }
    if (c.fail() || nv == 0) { RL_CTX_END(c); return RL_NOH; }
    if (nv > 1) { int bid, bend; rl_select_best(c, RL_OFF_V, nv, ref, ref_len, RTK_MODE_NW, -1.0, &bid, &bend); if (c.fail()) { RL_CTX_END(c); return RL_NOH; } rl_st(c, RL_OFF_V, rl_ld(c, RL_OFF_V + static_cast<uint32_t>(bid))); }
    const uint32_t h0 = rl_ld(c, RL_OFF_V);
    RL_CTX_END(c);
    if (rl_path_has_short_cycle(c, h0)) { rl_fail(c, RL_F_REPEAT); return RL_NOH; } // fixRepeats (src/GraphTraversal.cpp:1149-1334) has work: wave kernel
    return h0;
}

// ------------------------------------------------------------------------------------------------ anchors of a read in one orientation (src/Correction.cpp:196-213)
struct RlAnch { const uint32_t* pos; const uint64_t* hit; const uint64_t* hits_by_pos; uint32_t n, L, rev, k; };
RTK_DEV uint32_t rl_an_pos(const RlAnch& a, uint32_t i) { return a.rev ? (a.L - a.pos[a.n - 1u - i] - a.k) : a.pos[i]; }
RTK_DEV UMap rl_an_um(const RlAnch& a, uint32_t i) {
    const uint32_t j = a.rev ? (a.n - 1u - i) : i;
    UMap u = rtk_unpack_hit(a.hit ? a.hit[j] : a.hits_by_pos[a.pos[j]]);
    if (a.rev) u.strand ^= 1u;
    return u;
}
// first x in [lo, hi) with pos(x) >= key (strict: > key), else hi
RTK_DEV uint32_t rl_an_search(const RlAnch& a, uint32_t lo, uint32_t hi, uint64_t key, bool strict) {
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; const uint64_t p = rl_an_pos(a, mid); if (strict ? (p <= key) : (p < key)) lo = mid + 1; else hi = mid; }
    return lo;
}
RTK_DEV uint32_t rl_an_first_ge(const RlAnch& a, uint32_t lo, uint32_t hi, uint64_t key) { return rl_an_search(a, lo, hi, key, false); }
RTK_DEV uint32_t rl_an_first_gt(const RlAnch& a, uint32_t lo, uint32_t hi, uint64_t key) { return rl_an_search(a, lo, hi, key, true); }

// ------------------------------------------------------------------------------------------------ extractSemiWeakPaths (src/Correction.cpp:3-157)
// BFS results never hold more than one path, so `paths1` is a single running path (working path 0 / a record of the region level). A dead end
// becomes THE partial path (*partial). Returns the complete path's record or RL_NOH.
RTK_FN uint32_t rl_extract_semi_weak(RlCtx& c, const char* s_read, uint32_t s_len, uint32_t start_pos, const UMap& start_um, uint32_t end_pos_in, const UMap& end_um,
                                     const RlAnch& lvw, uint32_t lvw_lo, uint32_t lvw_hi, uint32_t i_weak, uint32_t* partial) {
    const uint32_t k = c.k();
    if (rtk_um_is_empty(end_um)) { rl_fail(c, RL_F_NOEND); return RL_NOH; }
    const uint32_t pos2 = end_pos_in;
    const uint32_t max_len_weak_region = c.o()->max_len_weak_region1; // :23 (pass 1)
    uint32_t next_weak_pos = 0;
    bool begin = true, end = false;
    rl_wp_start(c, 0, start_um, rtk_get_qual(1.0, 0, static_cast<uint64_t>(static_cast<int32_t>(c.o()->max_qual))));
    if (c.fail()) return RL_NOH;
    uint32_t cur = rl_wp_commit(c, 0, 0); uint32_t cur_pos = start_pos; bool have = !c.fail();
    const uint32_t nw = lvw_hi - lvw_lo;
    if (i_weak < nw) i_weak = rl_an_first_ge(lvw, lvw_lo + i_weak, lvw_lo + nw, start_pos) - lvw_lo;
    if (i_weak < nw) { const uint32_t wp = rl_an_pos(lvw, lvw_lo + i_weak); next_weak_pos = wp > start_pos + k ? wp : start_pos + k; }
    (void)s_len;
    while (have && !end && !c.fail()) {
        if (i_weak < nw) { const uint64_t lim_a = static_cast<uint64_t>(pos2 - k), lim_b = next_weak_pos; i_weak = rl_an_first_ge(lvw, lvw_lo + i_weak, lvw_lo + nw, lim_a < lim_b ? lim_a : lim_b) - lvw_lo; }
        else i_weak = nw;
        end = (i_weak == nw) || (static_cast<uint64_t>(rl_an_pos(lvw, lvw_lo + i_weak)) >= static_cast<uint64_t>(pos2 - k));
        const uint32_t target_pos = end ? pos2 : rl_an_pos(lvw, lvw_lo + i_weak);
        const uint32_t l_len = (target_pos - cur_pos) + k;
        const UMap um_start = begin ? start_um : rl_rec_back(c, cur);
        uint32_t res = RL_NOH; bool called = false;
        {
            UMap um_to = end_um;
            if (end) called = l_len <= max_len_weak_region;
            else if (l_len <= max_len_weak_region) { called = true; um_to = rl_an_um(lvw, lvw_lo + i_weak); }
            if (called) res = rl_explore_paths(c, rl_src_g(s_read + cur_pos), l_len, um_start, um_to);
        }
        if (c.fail()) break;
        if (called && res != RL_NOH) {
            rl_wp_load(c, 0, cur); rl_wp_merge(c, 0, res);
            if (c.fail()) break;
            cur = rl_wp_commit(c, 0, 0); cur_pos = target_pos;
        } else { *partial = cur; have = false; }
        if (!end) next_weak_pos = rl_an_pos(lvw, lvw_lo + i_weak) + k;
        begin = false;
    }
    return (have && !c.fail()) ? cur : RL_NOH;
}

// ------------------------------------------------------------------------------------------------ chooseColors (src/Correction.cpp:215-429)
// side lists: entries unitig << 3 | non-branching << 2 | side (0 middle, 1 right, 2 left), in insertion order
RTK_DEV bool rl_side_insert(RlCtx& c, uint32_t* n_side, uint32_t side, uint32_t u, bool nonbranching) { // true when unseen on this side
    const uint32_t n = *n_side;
    for (uint32_t i = 0; i < n; ++i) { const uint32_t e = rl_ld(c, RL_OFF_SIDE + i); if ((e >> 3) == u && (e & 3u) == side) return false; }
    if (n >= RL_SIDE_CAP) { rl_fail(c, RL_F_SIDE); return true; }
    rl_st(c, RL_OFF_SIDE + n, (u << 3) | (nonbranching ? 4u : 0u) | side); *n_side = n + 1u;
    return true;
}

// bit vectors of the colour universe (RL_CS_VW words of 32 bits) in the work area
RTK_DEV uint32_t rl_vec(uint32_t i) { return RL_OFF_CS_VEC + i * RL_CS_VW; }
RTK_DEV uint32_t rl_row(uint32_t slot, uint32_t global) { return RL_OFF_CS_ROWS + (2u * slot + global) * RL_CS_VW; }
RTK_DEV uint32_t rl_vcount(const RlCtx& c, uint32_t a, uint32_t vw) { uint32_t n = 0; for (uint32_t w = 0; w < vw; ++w) n += static_cast<uint32_t>(__builtin_popcount(rl_ld(c, a + w))); return n; }

// merges the sorted ids list[0 .. nl) (graph memory) into the sorted universe at word offset src (nu ids), result at dst; returns the new size
RTK_DEV uint32_t rl_merge_ids(RlCtx& c, uint32_t src, uint32_t nu, uint32_t dst, const uint32_t* list, uint32_t nl) {
    uint32_t i = 0, j = 0, o = 0;
    while (i < nu || j < nl) {
        uint32_t x;
        if (j >= nl) x = rl_ld(c, src + i++);
        else if (i >= nu) x = list[j++];
        else { const uint32_t a = rl_ld(c, src + i), b = list[j]; if (a < b) { x = a; ++i; } else if (b < a) { x = b; ++j; } else { x = a; ++i; ++j; } }
        if (o >= RL_ALL_CAP) { rl_fail(c, RL_F_IDS); return o; }
        rl_st(c, dst + o++, x);
    }
    return o;
}
// bit row of a sorted id list inside the sorted universe (every id of the list is in the universe)
RTK_DEV void rl_row_from_ids(const RlCtx& c, uint32_t row, uint32_t uni, uint32_t nu, uint32_t vw, const uint32_t* list, uint32_t nl) {
    // both are sorted: one walk over the universe, a bit wherever the next id of the list is met; a row word is stored once
    uint32_t j = 0, x = nl ? list[0] : 0u;
    for (uint32_t w = 0; w < vw; ++w) {
        uint32_t acc = 0;
        const uint32_t lim = (nu - 32u * w) < 32u ? (nu - 32u * w) : 32u;
        for (uint32_t b = 0; b < lim && j < nl; ++b) if (rl_ld(c, uni + 32u * w + b) == x) { acc |= 1u << b; ++j; if (j < nl) x = list[j]; }
        rl_st(c, row + w, acc);
    }
}

// all_pids into RL_OFF_ALL (sorted ids); returns their number. n_side entries at RL_OFF_SIDE.
RTK_FN uint32_t rl_choose_colors(RlCtx& c, uint32_t n_side) {
    const GraphView& g = *c.g();
    if (n_side == 0) return 0;
    RL_CTX(c, RL_P_COLOURS);
    // slots in the order middle, right, left (the order the reference walks its three maps); within a side: insertion order
    uint32_t slot_e[RL_SIDE_CAP]; uint32_t n_slots = 0;
    for (uint32_t sd = 0; sd < 3; ++sd) for (uint32_t i = 0; i < n_side; ++i) { const uint32_t e = rl_ld(c, RL_OFF_SIDE + i); if ((e & 3u) == sd) slot_e[n_slots++] = e; }
    const uint32_t* const col = g.col.get(); const uint64_t* const loff = g.loff.get(); const uint64_t* const goff = g.goff.get(); const int32_t* const gid = g.gid.get(); const uint32_t* const cardp = g.card.get();
    // ---- universe: every id of every side unitig, sorted, duplicates dropped (lists of a unitig / a global set seen before are skipped) ----
    uint32_t ua = RL_OFF_CS_UA, ub = RL_OFF_CS_UB, nu = 0; unsigned long long T = 0;
    for (uint32_t s = 0; s < n_slots && !c.fail(); ++s) {
        const uint32_t u = slot_e[s] >> 3; const int32_t gi = gid[u];
        const uint32_t nl = static_cast<uint32_t>(loff[u + 1] - loff[u]);
        const uint32_t ng = gi >= 0 ? static_cast<uint32_t>(goff[gi + 1] - goff[gi]) : 0u;
        T += nl + ng;
        bool seen_u = false, seen_g = false;
        for (uint32_t s2 = 0; s2 < s; ++s2) { const uint32_t u2 = slot_e[s2] >> 3; if (u2 == u) seen_u = true; if (gi >= 0 && gid[u2] == gi) seen_g = true; }
        if (!seen_u && nl) { nu = rl_merge_ids(c, ua, nu, ub, col + loff[u], nl); const uint32_t t_ = ua; ua = ub; ub = t_; }
        if (!seen_u && !seen_g && ng && !c.fail()) { nu = rl_merge_ids(c, ua, nu, ub, col + goff[gi], ng); const uint32_t t_ = ua; ua = ub; ub = t_; }
    }
    if (c.fail()) { RL_CTX_END(c); return 0; }
    c.c_colour += static_cast<uint32_t>(T);
    const uint32_t U = nu, vw = (U + 31u) >> 5;
    RL_LAP(c, 21);
#if defined(RTK_SIM) && defined(RTK_LANE_PROF)
    if (getenv("RTK_LANE_COST")) fprintf(stderr, "COLOURS slots %u T %llu U %u\n", n_slots, T, U);
#endif
    // ---- bit rows of every slot: local part, global part ----
    for (uint32_t s = 0; s < n_slots; ++s) {
        const uint32_t u = slot_e[s] >> 3; const int32_t gi = gid[u];
        rl_row_from_ids(c, rl_row(s, 0), ua, U, vw, col + loff[u], static_cast<uint32_t>(loff[u + 1] - loff[u]));
        if (gi >= 0) rl_row_from_ids(c, rl_row(s, 1), ua, U, vw, col + goff[gi], static_cast<uint32_t>(goff[gi + 1] - goff[gi]));
        else for (uint32_t w = 0; w < vw; ++w) rl_st(c, rl_row(s, 1) + w, 0);
    }
    RL_LAP(c, 22);
    // ---- candidate anchors: cardinality >= min_cov_vertices, first occurrence of their unitig, ordered by (cardinality, unitig) [D1] ----
    const uint32_t min_cov_v = c.o()->min_cov_vertices; const uint32_t d1 = c.o()->d1_desc;
    uint64_t key[RL_SIDE_CAP]; uint32_t kslot[RL_SIDE_CAP]; int quota[RL_SIDE_CAP]; uint32_t nsp = 0;
    for (uint32_t s = 0; s < n_slots; ++s) {
        const uint32_t u = slot_e[s] >> 3;
        if (cardp[u] < min_cov_v) continue;
        bool dup = false; for (uint32_t j = 0; j < nsp && !dup; ++j) dup = rtk_d1_unitig(key[j], d1) == u;
        if (dup) continue;
        const uint64_t kk = rtk_d1_key(cardp[u], u, d1);
        uint32_t p = nsp; while (p > 0 && key[p - 1] > kk) { key[p] = key[p - 1]; kslot[p] = kslot[p - 1]; --p; } // insertion sort (keys are distinct)
        key[p] = kk; kslot[p] = s; ++nsp;
    }
    const uint32_t cov = 30;
    for (uint32_t j = 0; j < nsp; ++j) { const uint32_t cd = static_cast<uint32_t>(key[j] >> 32); quota[j] = static_cast<int>(cd < cov ? cd : cov); }
    // ---- the six anchor classes: side (middle, right, left) x branching / non-branching; G2: the global set alone when there is one ----
    // vectors: 0..5 a[], 6 pos0, 7 pos1, 8 pos2, 9 a01, 10 a12, 11 a02, 12 nobranch_all, 13 i3, 14 i2, 15 nobranch, 16 branching, 17 prev2, 18 all, 19 curr, 20 a2
    for (uint32_t sh = 0; sh < 6; ++sh) {
        for (uint32_t w = 0; w < vw; ++w) rl_st(c, rl_vec(sh) + w, 0);
        const uint32_t side = sh % 3u, want_nb = sh >= 3 ? 4u : 0u;
        for (uint32_t s = 0; s < n_slots; ++s) {
            if ((slot_e[s] & 3u) != side || (slot_e[s] & 4u) != want_nb) continue;
            const uint32_t row = rl_row(s, gid[slot_e[s] >> 3] >= 0 ? 1u : 0u);
            for (uint32_t w = 0; w < vw; ++w) rl_st(c, rl_vec(sh) + w, rl_ld(c, rl_vec(sh) + w) | rl_ld(c, row + w));
        }
    }
    for (uint32_t w = 0; w < vw; ++w) {
        const uint32_t a0 = rl_ld(c, rl_vec(0) + w), a1 = rl_ld(c, rl_vec(1) + w), a2 = rl_ld(c, rl_vec(2) + w), a3 = rl_ld(c, rl_vec(3) + w), a4 = rl_ld(c, rl_vec(4) + w), a5 = rl_ld(c, rl_vec(5) + w);
        const uint32_t p0 = a0 | a3, p1 = a1 | a4, p2 = a2 | a5;
        const uint32_t a01 = p0 & p1, a12 = p1 & p2, a02 = p0 & p2;
        rl_st(c, rl_vec(12) + w, a3 | a4 | a5); rl_st(c, rl_vec(13) + w, a01 & a12); rl_st(c, rl_vec(14) + w, a01 | a12 | a02);
        rl_st(c, rl_vec(15) + w, a3 | a4 | a5); rl_st(c, rl_vec(16) + w, 0); rl_st(c, rl_vec(17) + w, 0); rl_st(c, rl_vec(18) + w, 0);
    }
    uint32_t nb_unselected = nsp, n_all_bits = 0;
    // ---- class loop (:331-429) ----
    for (int i = 5; i >= 0; --i) {
        if (nb_unselected == 0) break;
        uint32_t n2 = 0;
        for (uint32_t w = 0; w < vw; ++w) {
            const uint32_t prev2 = rl_ld(c, rl_vec(17) + w), i3 = rl_ld(c, rl_vec(13) + w), i2 = rl_ld(c, rl_vec(14) + w);
            uint32_t nob = rl_ld(c, rl_vec(15) + w), br = rl_ld(c, rl_vec(16) + w), a2;
            if (i == 5) a2 = nob & i3;
            else if (i == 4) { nob &= ~prev2; a2 = nob & i2; }
            else if (i == 3) { nob &= ~prev2; a2 = nob; }
            else if (i == 2) { br = (rl_ld(c, rl_vec(0) + w) | rl_ld(c, rl_vec(1) + w) | rl_ld(c, rl_vec(2) + w)) & ~rl_ld(c, rl_vec(12) + w); a2 = br & i3; }
            else if (i == 1) { br &= ~prev2; a2 = br & i2; }
            else { br &= ~prev2; a2 = br; }
            rl_st(c, rl_vec(15) + w, nob); rl_st(c, rl_vec(16) + w, br); rl_st(c, rl_vec(17) + w, a2); rl_st(c, rl_vec(19) + w, a2);
            n2 += static_cast<uint32_t>(__builtin_popcount(a2));
        }
        if (n2 == 0) continue;
        nb_unselected = 0;
        for (uint32_t j = 0; j < nsp; ++j) {
            int q = quota[j];
            if (q > 0) {
                const uint32_t rl_ = rl_row(kslot[j], 0), rg_ = rl_row(kslot[j], 1);
                bool touch = i == 0;
                if (!touch) for (uint32_t w = 0; w < vw && !touch; ++w) touch = ((rl_ld(c, rl_ + w) | rl_ld(c, rg_ + w)) & rl_ld(c, rl_vec(19) + w)) != 0;
                if (touch) {
                    const uint32_t cd = static_cast<uint32_t>(key[j] >> 32); const uint32_t min_cov = cd < cov ? cd : cov;
                    uint32_t sh = 0; for (uint32_t w = 0; w < vw; ++w) sh += static_cast<uint32_t>(__builtin_popcount((rl_ld(c, rl_ + w) | rl_ld(c, rg_ + w)) & rl_ld(c, rl_vec(18) + w)));
                    q = static_cast<int>(min_cov - (sh < min_cov ? sh : min_cov));
                    if (q > 0) { // pid = the q lowest ids of (colours of the anchor & curr); all |= pid; curr -= pid
                        uint32_t left = static_cast<uint32_t>(q), gained = 0;
                        for (uint32_t w = 0; w < vw && left; ++w) {
                            uint32_t x = (rl_ld(c, rl_ + w) | rl_ld(c, rg_ + w)) & rl_ld(c, rl_vec(19) + w), pid = 0;
                            while (x && left) { const uint32_t b = x & (~x + 1u); pid |= b; x ^= b; --left; }
                            if (pid) {
                                const uint32_t al = rl_ld(c, rl_vec(18) + w);
                                gained += static_cast<uint32_t>(__builtin_popcount(pid & ~al));
                                rl_st(c, rl_vec(18) + w, al | pid); rl_st(c, rl_vec(19) + w, rl_ld(c, rl_vec(19) + w) & ~pid);
                            }
                        }
                        n_all_bits += gained;
                        q -= static_cast<int>(gained) < q ? static_cast<int>(gained) : q;
                    }
                }
                quota[j] = q;
            }
            nb_unselected += q > 0 ? 1u : 0u;
        }
    }
    // ---- all_pids back to a sorted id list ----
    if (n_all_bits > RL_ALL_CAP) { rl_fail(c, RL_F_IDS); RL_CTX_END(c); return 0; }
    uint32_t at = 0;
    for (uint32_t w = 0; w < vw; ++w) { uint32_t x = rl_ld(c, rl_vec(18) + w); while (x) { const uint32_t b = static_cast<uint32_t>(__builtin_ctz(x)); rl_st(c, RL_OFF_ALL + at++, rl_ld(c, ua + 32u * w + b)); x &= x - 1u; } }
    RL_CTX_END(c);
    return at;
}

// ------------------------------------------------------------------------------------------------ SNP annotations (rtk_ambiguity.h, lane by lane)
RTK_DEV uint32_t rl_amb_list(uint32_t i) { return RL_OFF_AMB + i * RL_AMB_CAP; } // 0 v_ambiguity, 1 safe / running vector, 2 all / merge buffer, 3 one mapping, 4 linked alleles
RTK_DEV uint32_t rl_amb_mk(uint32_t pos, char ch) { return (pos << 8) | static_cast<uint32_t>(static_cast<unsigned char>(ch)); }
RTK_DEV int rl_amb_find(const RlCtx& c, uint32_t list, uint32_t n, uint32_t pos) { for (uint32_t i = 0; i < n; ++i) if ((rl_ld(c, list + i) >> 8) == pos) return static_cast<int>(i); return -1; }
RTK_DEV char rl_unitig_char(const RlCtx& c, uint32_t u, uint32_t i) { return static_cast<char>((0x54474341u >> (8u * rtk_base(*c.g(), c.g()->uoff.get()[u] + i))) & 0xFFu); }

// UnitigData::get_ambiguity_char(um) (UnitigData.hpp:458-481) into list 3; returns the number of entries
RTK_DEV uint32_t rl_amb_of_um(RlCtx& c, const UMap& um) {
    const GraphView& g = *c.g();
    const uint64_t* const amb = g.amb.get();
    const uint64_t* ent = amb + (static_cast<uint64_t>(static_cast<uint32_t>(g.n_unitigs)) + 1ull);
    const uint64_t a0 = amb[um.unitig], a1 = amb[um.unitig + 1];
    const uint32_t sz = um.len + c.k() - 1u, end = um.dist + sz;
    uint32_t n = 0;
    for (uint64_t j = 0; j < a1 - a0; ++j) {
        const uint64_t e = ent[um.strand ? (a0 + j) : (a1 - 1ull - j)];
        const uint32_t pos = static_cast<uint32_t>(e >> 4); const char ch = rtk_iupac_chr(static_cast<uint32_t>(e & 15ull));
        if (pos < um.dist || pos >= end) continue;
        if (n >= RL_AMB_CAP) { rl_fail(c, RL_F_AMB); return 0; }
        rl_st(c, rl_amb_list(3) + n, um.strand ? rl_amb_mk(pos - um.dist, ch) : rl_amb_mk(sz - (pos - um.dist) - 1u, rtk_iupac_comp(ch))); ++n;
    }
    return n;
}

// getAmbiguityVector(path) (src/GraphTraversal.cpp:966-1036) + the push_back of its callers; returns the new size of v_ambiguity (list 0)
RTK_FN uint32_t rl_amb_collect(RlCtx& c, uint32_t h, uint32_t offset, uint32_t n_amb) {
    if (static_cast<uint64_t>(c.g()->n_amb) == 0) return n_amb;
    RL_ENTER(c);
    const uint32_t pw = rl_h_w(h); const uint32_t n = rl_p_n(c, pw), k1 = c.k() - 1u;
    { bool any = false; const uint64_t* const amb = c.g()->amb.get(); for (uint32_t x = 0; x < n && !any; ++x) { const uint32_t u = rl_ld(c, pw + 4u + 3u * x) >> 1; any = amb[u + 1] != amb[u]; } if (!any) { RL_LEAVE(c, RL_P_AMB); return n_amb; } }
    const uint32_t va = rl_amb_list(1), vt = rl_amb_list(2), vu = rl_amb_list(3), cap = RL_AMB_CAP;
    uint32_t nva = 0, prev_l = 0, pos_prev_l = 0;
    for (uint32_t x = 0; x < n; ++x) {
        const UMap um = rl_um_ld(c, pw + 4u + 3u * x);
        const uint32_t nvu = rl_amb_of_um(c, um);
        if (c.fail()) return n_amb;
        uint32_t nvt = 0, ip = pos_prev_l, ic = 0;
        while (ip != nva && ic != nvu && (rl_ld(c, vu + ic) >> 8) < k1 && nvt < cap) {
            const uint32_t eu = rl_ld(c, vu + ic), ea = rl_ld(c, va + ip);
            const uint32_t cur_pos = (eu >> 8) + prev_l, pp = ea >> 8;
            if (pp < cur_pos) { rl_st(c, vt + nvt++, ea); ++ip; }
            else if (pp > cur_pos) { rl_st(c, vt + nvt++, rl_amb_mk(cur_pos, static_cast<char>(eu & 0xFFu))); ++ic; }
            else { rl_st(c, vt + nvt++, rl_amb_mk(pp, rtk_iupac_chr(rtk_iupac_idx(static_cast<char>(ea & 0xFFu)) | rtk_iupac_idx(static_cast<char>(eu & 0xFFu))))); ++ip; ++ic; }
        }
        if (nvt + (nva - ip) + (nvu - ic) > cap || pos_prev_l + nvt + (nva - ip) + (nvu - ic) > cap) { rl_fail(c, RL_F_AMB); return n_amb; }
        for (; ip != nva; ++ip) rl_st(c, vt + nvt++, rl_ld(c, va + ip));
        for (; ic != nvu; ++ic) { const uint32_t eu = rl_ld(c, vu + ic); rl_st(c, vt + nvt++, rl_amb_mk((eu >> 8) + prev_l, static_cast<char>(eu & 0xFFu))); }
        prev_l += um.len;
        nva = pos_prev_l;
        for (uint32_t i = 0; i < nvt; ++i) { const uint32_t e = rl_ld(c, vt + i); rl_st(c, va + nva++, e); pos_prev_l += ((e >> 8) < prev_l) ? 1u : 0u; }
    }
    uint32_t na = n_amb;
    if (na + nva > cap) { rl_fail(c, RL_F_AMB); return n_amb; }
    for (uint32_t i = 0; i < nva; ++i) { const uint32_t e = rl_ld(c, va + i); rl_st(c, rl_amb_list(0) + na++, rl_amb_mk(offset + (e >> 8), static_cast<char>(e & 0xFFu))); }
    RL_LEAVE(c, RL_P_AMB);
    return na;
}

// Bifrost findUnitig from a k-mer hit [A7] (rtk_extend_hit), on work-area bytes
RTK_DEV UMap rl_extend_hit(const RlCtx& c, uint64_t hit, uint32_t str_b, uint32_t pos, uint32_t len) {
    const uint32_t k = c.k();
    UMap um = rtk_unpack_hit(hit);
    const uint32_t ul = rl_ulen(c, um.unitig);
    const uint32_t j0 = pos + k;
    const uint32_t room_s = len > j0 ? len - j0 : 0u;
    const uint32_t room_u = um.strand ? (ul > um.dist + k ? ul - (um.dist + k) : 0u) : um.dist;
    const uint32_t room = room_s < room_u ? room_s : room_u;
    uint32_t n = 0;
    for (; n < room; ++n) {
        const char uc = um.strand ? rl_unitig_char(c, um.unitig, um.dist + k + n) : rtk_iupac_comp(rl_unitig_char(c, um.unitig, um.dist - 1u - n));
        if (static_cast<char>(rl_ldb(c, str_b + j0 + n)) != uc) break;
    }
    if (!um.strand) um.dist -= n;
    um.len = n + 1u;
    return um;
}

// fixAmbiguity (src/Alignment.cpp:527-844): query / quality = string buffers sq / qq (lengths query_len / quality_len), ref = the raw region
RTK_FN void rl_fix_ambiguity(RlCtx& c, uint32_t sq, uint32_t query_len, uint32_t qq, uint32_t quality_len, RlSrc ref, uint32_t ref_len, uint32_t n_amb) {
    if (n_amb == 0) return;
    const uint32_t k = c.k(), cap = RL_AMB_CAP;
    if (quality_len < query_len) { rl_fail(c, RL_F_OTHER); return; }
    RL_CTX(c, RL_P_FIXAMB);
    const uint64_t oq = static_cast<uint64_t>(static_cast<int32_t>(c.o()->out_qual)), mq = static_cast<uint64_t>(static_cast<int32_t>(c.o()->max_qual));
    const char q_max_corr = rtk_get_qual(1.0, oq, mq), q_min_corr = rtk_get_qual(0.0, oq, mq), q_min_conf_corr = rtk_get_qual(c.o()->min_confidence_snp_corr, 0, mq);
    const char c_no = 'X';
    const uint32_t v = rl_amb_list(0), ms = rl_amb_list(1), ma = rl_amb_list(2), vu = rl_amb_list(3), sa = rl_amb_list(4);
    const uint32_t qb = rl_sb(sq), qqb = rl_sb(qq), qtb = rl_sb(RL_SB_TMP);
    uint32_t nms = 0, nma = 0, nsa = 0;
    for (uint32_t i = 0; i < n_amb; ++i) {
        const uint32_t e = rl_ld(c, v + i); const uint32_t p = e >> 8;
        if (static_cast<char>(rl_ldb(c, qqb + p)) < q_min_conf_corr && rl_amb_find(c, ms, nms, p) < 0) rl_st(c, ms + nms++, e);
    }
    if (nms == 0) { // every annotated base is confident: nothing enters the sets unless the alignment meets a non-ACGT character (:630-678)
        bool odd = false;
        for (uint32_t i = 0; i < query_len && !odd; ++i) odd = !rtk_is_dna(static_cast<char>(rl_ldb(c, qb + i)));
        for (uint32_t i = 0; i < ref_len && !odd; ++i) odd = !rtk_is_dna(static_cast<char>(rl_get(c, ref, i)));
        if (!odd) { RL_CTX_END(c); return; }
    }
    rl_copy_words(c, qtb >> 2, qb >> 2, (query_len + 3u) >> 2); // query_tmp
    for (uint32_t i = 0; i < n_amb; ++i) { const uint32_t e = rl_ld(c, v + i); const uint32_t p = e >> 8; if (static_cast<char>(rl_ldb(c, qqb + p)) < q_min_conf_corr) rl_stb(c, qtb + p, static_cast<unsigned char>(e & 0xFFu)); }
    for (uint32_t i = 0; i < nms; ++i) rl_st(c, ma + i, rl_ld(c, ms + i));
    nma = nms;
    uint32_t nm = 0, off = RL_MV_BYTES; const uint32_t mvb = rl_mvb(0);
    rl_align_path(c, rl_src_l(qtb), static_cast<int>(query_len), ref, static_cast<int>(ref_len), RTK_MODE_SHW, mvb, &off, &nm);
    c.sv_valid = 0;
    if (c.fail()) return;
    { // walk of the alignment (:612-706)
        uint32_t q_pos = 0, t_pos = 0;
        for (uint32_t a = 0; a < nm; ++a) {
            const unsigned char m = rl_ldb(c, mvb + off + a);
            if (m == 0 || m == 3) {
                const char qc = static_cast<char>(rl_ldb(c, qtb + q_pos)), tc = static_cast<char>(rl_get(c, ref, t_pos));
                if (!rtk_is_dna(qc)) {
                    if (!rtk_is_dna(tc)) { const int x = rl_amb_find(c, ms, nms, q_pos); if (x >= 0) rl_st(c, ms + x, rl_amb_mk(q_pos, c_no)); }
                    else if (static_cast<char>(rl_ldb(c, qqb + q_pos)) >= q_min_corr) { if (rtk_iupac_overlap(qc, tc)) { const int x = rl_amb_find(c, ms, nms, q_pos); if (x >= 0) rl_st(c, ms + x, rl_amb_mk(q_pos, tc)); } }
                    const int y = rl_amb_find(c, ma, nma, q_pos); if (y >= 0) rl_st(c, ma + y, rl_amb_mk(q_pos, tc));
                } else if (!rtk_is_dna(tc)) {
                    if (static_cast<char>(rl_ldb(c, qqb + q_pos)) < q_min_conf_corr || !rtk_iupac_overlap(qc, tc)) {
                        if (nms >= cap || nma >= cap) { rl_fail(c, RL_F_AMB); return; }
                        if (rl_amb_find(c, ms, nms, q_pos) < 0) rl_st(c, ms + nms++, rl_amb_mk(q_pos, c_no));
                        if (rl_amb_find(c, ma, nma, q_pos) < 0) rl_st(c, ma + nma++, rl_amb_mk(q_pos, tc));
                    }
                }
                ++q_pos; ++t_pos;
            } else if (m == 1) {
                if (!rtk_is_dna(static_cast<char>(rl_ldb(c, qtb + q_pos)))) {
                    const int x = rl_amb_find(c, ms, nms, q_pos), y = rl_amb_find(c, ma, nma, q_pos);
                    if (x >= 0 && y >= 0) { rl_st(c, ma + y, rl_amb_mk(q_pos, static_cast<char>(rl_ld(c, ms + x) & 0xFFu))); rl_st(c, ms + x, rl_amb_mk(q_pos, c_no)); }
                }
                ++q_pos;
            } else ++t_pos;
        }
    }
    // alleles of the other annotated positions of the unitig a decided SNP lies on (:713-768); q_sub lives in RL_SB_CAND
    for (uint32_t e = 0; e < nms; ++e) {
        const uint32_t me = rl_ld(c, ms + e); const char pc = static_cast<char>(me & 0xFFu);
        if (!rtk_is_dna(pc)) continue;
        const uint32_t p = me >> 8;
        const uint32_t pos_buff = (p < k - 1u) ? 0u : (p - k + 1u);
        const uint32_t len_buff = ((p + k < query_len) ? (p + k) : query_len) - pos_buff;
        const uint32_t pos_snp_buff = p - pos_buff;
        const uint32_t qsb = rl_sb(RL_SB_CAND);
        rl_app(c, qsb, 0, rl_src_l(qb + pos_buff), len_buff);
        if (c.fail()) return;
        rl_stb(c, qsb + pos_snp_buff, static_cast<unsigned char>(pc));
        const uint32_t nwin = len_buff >= k ? len_buff - k + 1u : 0u;
        uint32_t skip_until = 0; bool skip_one = false;
        for (uint32_t w = 0; w < nwin; ++w) { // [A6] KmerIterator: the all-ACGT windows, in order
            bool ok = true; for (uint32_t x = 0; x < k && ok; ++x) ok = rtk_is_dna(static_cast<char>(rl_ldb(c, qsb + w + x)));
            if (!ok) continue;
            if (w < skip_until) continue;
            if (skip_one) { skip_one = false; continue; }
            RtkKm km = rtk_km_zero(); for (uint32_t x = 0; x < k; ++x) km = rtk_km_push(km, static_cast<uint64_t>(rtk_cls(static_cast<unsigned char>(rl_ldb(c, qsb + w + x) & 0xDF))), static_cast<int>(k));
            const uint64_t hit = rtk_find_km(*c.g(), km, nullptr);
            if (hit == RTK_NO_HIT) continue;
            const UMap um = rl_extend_hit(c, hit, qsb, w, len_buff);
            const uint32_t usz = rl_ulen(c, um.unitig);
            UMap full = um; full.dist = 0; full.len = usz - k + 1u;
            const uint32_t nvu = rl_amb_of_um(c, full);
            if (c.fail()) return;
            uint32_t pos_snp_unitig = (pos_snp_buff - w) + um.dist;
            if (!um.strand) pos_snp_unitig = usz - pos_snp_unitig - 1u;
            for (uint32_t a = 0; a < nvu; ++a) {
                const uint32_t ap = rl_ld(c, vu + a) >> 8;
                const int64_t pos = (ap <= pos_snp_unitig) ? (static_cast<int64_t>(p) - static_cast<int64_t>(pos_snp_unitig - ap)) : (static_cast<int64_t>(p) + static_cast<int64_t>(ap - pos_snp_unitig));
                if (pos < 0 || pos >= static_cast<int64_t>(query_len) || pos == static_cast<int64_t>(p)) continue;
                const int x = rl_amb_find(c, ms, nms, static_cast<uint32_t>(pos));
                if (x < 0 || rtk_is_dna(static_cast<char>(rl_ld(c, ms + x) & 0xFFu))) continue;
                const char uc = um.strand ? rl_unitig_char(c, um.unitig, ap) : rtk_iupac_comp(rl_unitig_char(c, um.unitig, usz - 1u - ap));
                const uint32_t ent = rl_amb_mk(static_cast<uint32_t>(pos), uc);
                bool dup = false;
                for (uint32_t z = 0; z < nsa && !dup; ++z) dup = rl_ld(c, sa + z) == ent;
                if (!dup) { if (nsa >= cap) { rl_fail(c, RL_F_AMB); return; } rl_st(c, sa + nsa++, ent); }
            }
            skip_until = w + (um.len - 1u); skip_one = um.len >= 2; // it_km += um.len - 1, then ++it_km
        }
    }
    for (uint32_t i = 0; i < nsa; ++i) { // a linked position with exactly one candidate allele takes it, when compatible (:771-790)
        const uint32_t ei = rl_ld(c, sa + i); const uint32_t pos = ei >> 8;
        uint32_t same = 0;
        for (uint32_t j = 0; j < nsa; ++j) same += ((rl_ld(c, sa + j) >> 8) == pos) ? 1u : 0u;
        if (same != 1) continue;
        const int x = rl_amb_find(c, ms, nms, pos);
        if (x >= 0 && rtk_iupac_overlap(static_cast<char>(ei & 0xFFu), static_cast<char>(rl_ld(c, ms + x) & 0xFFu))) rl_st(c, ms + x, rl_amb_mk(pos, static_cast<char>(ei & 0xFFu)));
    }
    for (uint32_t e = 0; e < nms; ++e) { // :792-838
        const uint32_t me = rl_ld(c, ms + e); const uint32_t p = me >> 8; const char pc = static_cast<char>(me & 0xFFu);
        if (pc == c_no || static_cast<char>(rl_ldb(c, qqb + p)) < q_min_corr) {
            const int y = rl_amb_find(c, ma, nma, p);
            if (y >= 0) { rl_stb(c, qtb + p, static_cast<unsigned char>(rl_ld(c, ma + y) & 0xFFu)); rl_stb(c, qqb + p, static_cast<unsigned char>(q_max_corr)); }
        }
        else if (!rtk_is_dna(pc)) rl_stb(c, qtb + p, rl_ldb(c, qb + p));
        else rl_stb(c, qtb + p, static_cast<unsigned char>(pc));
    }
    rl_copy_words(c, qb >> 2, qtb >> 2, (query_len + 3u) >> 2);
    RL_CTX_END(c);
}

// ------------------------------------------------------------------------------------------------ ResultCorrection (src/ResultCorrection.hpp)
struct RlRes { uint32_t sb_seq, sb_qual, seq_len, qual_len, bm, old_len, is_corrected; }; // bm: word offset of the position bitmap (32-bit words)
RTK_DEV void rl_bm_add_range(const RlCtx& c, uint32_t bm, uint32_t a, uint32_t b) { // [a, b)
    for (uint32_t i = a; i < b;) { const uint32_t w = i >> 5, lo = i & 31u; const uint32_t hi = (b - (i - lo)) < 32u ? (b - (i - lo)) : 32u; // bits [lo, hi) of word w
        const uint32_t mask = ((hi == 32u) ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        rl_st(c, bm + w, rl_ld(c, bm + w) | mask); i += hi - lo; }
}
RTK_DEV uint32_t rl_bm_card(const RlCtx& c, uint32_t bm, uint32_t n) { uint32_t x = 0; for (uint32_t w = 0; w < (n + 31u) / 32u; ++w) x += static_cast<uint32_t>(__builtin_popcount(rl_ld(c, bm + w))); return x; }
RTK_DEV bool rl_bm_get(const RlCtx& c, uint32_t bm, uint32_t i) { return (rl_ld(c, bm + (i >> 5)) >> (i & 31u)) & 1u; }
// first position >= p (capped at n) whose bit equals `want`
RTK_DEV uint32_t rl_bm_next(const RlCtx& c, uint32_t bm, uint32_t n, uint32_t p, bool want) {
    if (p >= n) return n;
    const uint32_t words = (n + 31u) / 32u;
    for (uint32_t w = p >> 5; w < words; ++w) {
        uint32_t x = want ? rl_ld(c, bm + w) : ~rl_ld(c, bm + w);
        if (w == (p >> 5)) x &= ~((1u << (p & 31u)) - 1u);
        if (x) { const uint32_t pos = 32u * w + static_cast<uint32_t>(__builtin_ctz(x)); return pos < n ? pos : n; }
    }
    return n;
}
RTK_DEV uint32_t rl_len_corrected(const RlCtx& c, const RlRes& r, uint32_t p) { return rl_bm_next(c, r.bm, r.old_len, p, false) - (p < r.old_len ? p : r.old_len); }   // :117-128
RTK_DEV uint32_t rl_len_uncorrected(const RlCtx& c, const RlRes& r, uint32_t p) { return rl_bm_next(c, r.bm, r.old_len, p, true) - (p < r.old_len ? p : r.old_len); } // :130-142

// 32 bits of a position bitmap from bit `lo` on (may be negative; bits outside the `words` words read as 0)
RTK_DEV uint32_t rl_bm_window(const RlCtx& c, uint32_t bm, uint32_t words, int32_t lo) {
    if (lo <= -32) return 0u;
    if (lo < 0) return words ? (rl_ld(c, bm) << static_cast<uint32_t>(-lo)) : 0u;
    const uint32_t w = static_cast<uint32_t>(lo) >> 5, sh = static_cast<uint32_t>(lo) & 31u;
    uint32_t x = 0;
    if (w < words) x = rl_ld(c, bm + w) >> sh;
    if (sh && w + 1u < words) x |= rl_ld(c, bm + w + 1u) << (32u - sh);
    return x;
}
RTK_DEV uint32_t rl_brev32(uint32_t x) { x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2); x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); return __builtin_bswap32(x); }
RTK_FN void rl_res_reverse_complement(RlCtx& c, RlRes& r) { // :72-88
    if (r.seq_len == 0) return;
    RL_ENTER(c);
    const uint32_t words = (r.old_len + 31u) / 32u, tmp = RL_OFF_BM + 2u * RL_BM_W;
    // new bit j = old bit old_len - 1 - j: output word ow is the bit reversal of the 32 old bits that end at old_len - 1 - 32 ow
    for (uint32_t ow = 0; ow < words; ++ow) rl_st(c, tmp + ow, rl_brev32(rl_bm_window(c, r.bm, words, static_cast<int32_t>(r.old_len) - 32 - 32 * static_cast<int32_t>(ow))));
    { const uint32_t tail = r.old_len & 31u; if (tail) rl_st(c, tmp + words - 1u, rl_ld(c, tmp + words - 1u) & ((1u << tail) - 1u)); }
    for (uint32_t w = 0; w < words; ++w) rl_st(c, r.bm + w, rl_ld(c, tmp + w));
    const uint32_t tb_ = rl_sb(RL_SB_TMP);
    { // the sequence: four characters per step from the end, complemented (A <-> T: ^ 0x15, C <-> G: ^ 0x04; anything else one by one)
        const RlSrc src = rl_src_l(rl_sb(r.sb_seq)); RlW w = rl_w_open(c, tb_, 0); uint32_t i = 0;
        for (; i + 4u <= r.seq_len; i += 4u) {
            const uint32_t x = rl_get4(c, src, r.seq_len - 4u - i);
            uint32_t y;
            if ((rl_eq4(x, 'A') | rl_eq4(x, 'C') | rl_eq4(x, 'G') | rl_eq4(x, 'T')) == 0xFu) y = x ^ (0x15151515u ^ (((x >> 1) & 0x01010101u) * 0x11u));
            else { y = 0; for (int b = 0; b < 4; ++b) y |= static_cast<uint32_t>(static_cast<unsigned char>(rtk_comp(static_cast<char>((x >> (8 * b)) & 0xFFu)))) << (8 * b); }
            rl_st(c, (tb_ + i) >> 2, __builtin_bswap32(y)); w.len += 4u;
        }
        for (; i < r.seq_len; ++i) rl_w_put(c, w, static_cast<unsigned char>(rtk_comp(static_cast<char>(rl_ldb(c, rl_sb(r.sb_seq) + r.seq_len - 1u - i)))));
        rl_w_close(c, w);
    }
    rl_copy_words(c, rl_sb(r.sb_seq) >> 2, tb_ >> 2, (r.seq_len + 3u) >> 2);
    { const RlSrc src = rl_src_l(rl_sb(r.sb_qual)); RlW w = rl_w_open(c, tb_, 0); uint32_t i = 0;
      for (; i + 4u <= r.qual_len; i += 4u) { rl_st(c, (tb_ + i) >> 2, __builtin_bswap32(rl_get4(c, src, r.qual_len - 4u - i))); w.len += 4u; }
      for (; i < r.qual_len; ++i) rl_w_put(c, w, rl_ldb(c, rl_sb(r.sb_qual) + r.qual_len - 1u - i));
      rl_w_close(c, w); }
    rl_copy_words(c, rl_sb(r.sb_qual) >> 2, tb_ >> 2, (r.qual_len + 3u) >> 2);
    RL_LEAVE(c, RL_P_REVCOMP);
}

// ------------------------------------------------------------------------------------------------ the `correct` lambda (src/Correction.cpp:431-753), pass 1, with end anchor
// visits the anchors x = start, start + step, ... while in_range(pos) holds, fn(um) once per RUN of consecutive anchors on the same unitig
template <class Cond, class Fn>
RTK_DEV void rl_scan_anchor_runs(const RlAnch& a, int64_t start, int step, Cond in_range, Fn fn) {
    uint32_t prev_unitig = RTK_NONE32; bool first = true;
    for (int64_t x = start; x >= 0 && x < static_cast<int64_t>(a.n); x += step) {
        if (!in_range(rl_an_pos(a, static_cast<uint32_t>(x)))) break;
        const UMap um = rl_an_um(a, static_cast<uint32_t>(x));
        if (first || um.unitig != prev_unitig) fn(um);
        prev_unitig = um.unitig; first = false;
    }
}

RTK_FN void rl_correct_region(RlCtx& c, const char* s_read, uint32_t s_len, const RlAnch& v_s, const RlAnch& v_w, uint32_t i_s, uint32_t i_w, bool have_colours, RlRes& res) {
    const uint32_t k = c.k();
    const GraphView& g = *c.g();
    if (!((i_s + 1u) < v_s.n)) { rl_fail(c, RL_F_NOEND); return; }
    uint32_t p1 = rl_an_pos(v_s, i_s); UMap um1 = rl_an_um(v_s, i_s);
    const uint32_t p2 = rl_an_pos(v_s, i_s + 1u);
    const UMap um2 = rl_an_um(v_s, i_s + 1u);
    const uint32_t first_pos = p1;
    uint32_t len_weak_region = p2 - p1 + k;
    const uint64_t u_min_start = static_cast<uint64_t>(p1) - static_cast<uint64_t>(static_cast<uint32_t>(c.o()->insert_sz)); // wraps below insert_sz (G1)
    const uint64_t u_min_end = static_cast<uint64_t>(p2) + static_cast<uint64_t>(static_cast<uint32_t>(c.o()->insert_sz));
    res.old_len = len_weak_region; res.is_corrected = 0; res.seq_len = 0; res.qual_len = 0;
    if ((len_weak_region + 31u) / 32u + 1u > RL_BM_W) { rl_fail(c, RL_F_BM); return; }
    for (uint32_t w = 0; w < (len_weak_region + 31u) / 32u + 1u; ++w) rl_st(c, res.bm + w, 0);
    const char q_min = rtk_get_qual(0.0, 0, static_cast<uint64_t>(static_cast<int32_t>(c.o()->max_qual)));
    const uint32_t max_len_weak_anchors = c.o()->max_len_weak_region1; // :177
    const uint32_t max_km_cov = c.o()->max_km_cov, min_cov_v = c.o()->min_cov_vertices;
    // weak anchors inside the region: l_v_w = v_w[lw_lo .. lw_hi)
    uint32_t lw_lo = 0, lw_hi = 0;
    { const uint32_t v_w_sz = v_w.n;
      if (v_w_sz) { const uint32_t x = i_w - (((i_w != 0) && (i_w >= v_w_sz)) ? 1u : 0u); lw_lo = rl_an_first_ge(v_w, x, v_w_sz, first_pos); lw_hi = rl_an_first_ge(v_w, lw_lo, v_w_sz, p2); } }
    RL_CTX(c, RL_P_ASSEMBLE);
    if (!have_colours) {
        uint32_t n_side = 0;
        const uint32_t* const kcov = g.kcov.get();
        auto consider = [&](uint32_t side, const UMap& um, uint32_t& nb_branching) {
            const uint32_t u = um.unitig; const bool br = (rl_flags(c, u) & RTK_F_BRANCHING) != 0;
            if (kcov[u] < max_km_cov && (!br || nb_branching < 5)) { const bool unseen = rl_side_insert(c, &n_side, side, u, !br); nb_branching += (unseen && br) ? 1u : 0u; }
        };
        { // left (:476-516)
            uint32_t nbb = 0;
            rl_scan_anchor_runs(v_s, static_cast<int64_t>(i_s), -1, [&](uint32_t p) { return static_cast<uint64_t>(p) > u_min_start; }, [&](const UMap& um) { consider(2, um, nbb); });
            const uint32_t v_w_sz = v_w.n;
            if (v_w_sz) {
                const uint32_t x0 = i_w - (((i_w != 0) && (i_w >= v_w_sz)) ? 1u : 0u);
                const uint32_t f = rl_an_first_gt(v_w, 0, x0 + 1u, u_min_start); // the reference walks back while pos > u_min_start and index > 0
                const uint32_t x = f > 0 ? f - 1u : 0u;
                rl_scan_anchor_runs(v_w, static_cast<int64_t>(x), +1, [&](uint32_t p) { return p < first_pos; }, [&](const UMap& um) { consider(2, um, nbb); });
            }
        }
        { // right (:518-561)
            uint32_t nbb = 0;
            rl_scan_anchor_runs(v_s, static_cast<int64_t>(i_s) + 1, +1, [&](uint32_t p) { return static_cast<uint64_t>(p) < u_min_end; }, [&](const UMap& um) { consider(1, um, nbb); });
            const uint32_t v_w_sz = v_w.n;
            if (v_w_sz) {
                const uint32_t x0 = i_w - (((i_w != 0) && (i_w >= v_w_sz)) ? 1u : 0u);
                const uint32_t x = rl_an_first_ge(v_w, x0, v_w_sz, p2);
                rl_scan_anchor_runs(v_w, static_cast<int64_t>(x), +1, [&](uint32_t p) { return static_cast<uint64_t>(p) < u_min_end; }, [&](const UMap& um) { consider(1, um, nbb); });
            }
        }
        if (lw_hi > lw_lo) // middle (:563-585)
            rl_scan_anchor_runs(v_w, static_cast<int64_t>(lw_lo), +1, [&](uint32_t p) { return p < p2; }, [&](const UMap& um) { const uint32_t u = um.unitig; if (kcov[u] < max_km_cov) rl_side_insert(c, &n_side, 0, u, !(rl_flags(c, u) & RTK_F_BRANCHING)); });
        RL_LAP(c, RL_P_SIDE);
        if (c.fail()) return;
        c.n_all() = rl_choose_colors(c, n_side);
        if (c.fail()) return;
    }
    const uint32_t n_all = c.n_all();
    // ---- paths ----
    c.top(0) = 0;
    uint32_t n_amb = 0;
    uint32_t complete = RL_NOH, partial = RL_NOH;
    const uint32_t s_corr = rl_sb(res.sb_seq), q_corr = rl_sb(res.sb_qual); uint32_t sl_ = 0, ql_ = 0;
    const uint32_t nlw = lw_hi - lw_lo;
    auto clamp_len = [&](uint32_t pos, uint32_t len) -> uint32_t { return (pos + len <= s_len) ? len : (pos < s_len ? s_len - pos : 0u); }; // std::string::substr
    auto add_uncorrected = [&](uint32_t pos, uint32_t len, char q) { sl_ = rl_app(c, s_corr, sl_, rl_src_g(s_read + pos), clamp_len(pos, len)); ql_ = rl_app_fill(c, q_corr, ql_, q, len_weak_region); }; // :459-469
    bool first_call = true, found_first = false, do_call = n_all >= min_cov_v;
    uint32_t i_w_s = 0;
    for (;;) {
        if (do_call) { partial = RL_NOH; complete = rl_extract_semi_weak(c, s_read, s_len, p1, um1, p2, um2, v_w, lw_lo, lw_hi, first_call ? 0u : i_w_s, &partial); }
        if (c.fail()) return;
        if (first_call && complete != RL_NOH) found_first = true;
        first_call = false;
        if (!(complete == RL_NOH && partial != RL_NOH && nlw != 0 && n_all >= min_cov_v)) break;
        { // :619-651
            int aid, aend;
            rl_st(c, RL_OFF_V, partial);
            rl_select_best(c, RL_OFF_V, 1, rl_src_g(s_read + p1), len_weak_region, RTK_MODE_SHW, c.o()->weak_region_len_factor, &aid, &aend);
            if (c.fail() || aid == -1) break;
            {
                const uint32_t next_pos = p1 + static_cast<uint32_t>(aend) + k;
                while (i_w_s < nlw && rl_an_pos(v_w, lw_lo + i_w_s) < next_pos) ++i_w_s;
                if (i_w_s >= nlw || static_cast<uint64_t>(rl_an_pos(v_w, lw_lo + i_w_s)) >= static_cast<uint64_t>(p2) - k || (rl_an_pos(v_w, lw_lo + i_w_s) - p1) >= max_len_weak_anchors) break;
            }
            const uint32_t hb = partial;
            const uint32_t wpos = rl_an_pos(v_w, lw_lo + i_w_s);
            const uint32_t pl = rl_rec_to_string(c, hb, RL_SB_PATH); if (pl == 0xFFFFFFFFu) break;
            n_amb = rl_amb_collect(c, hb, sl_, n_amb);
            sl_ = rl_app(c, s_corr, sl_, rl_src_l(rl_sb(RL_SB_PATH)), pl);
            sl_ = rl_app(c, s_corr, sl_, rl_src_g(s_read + p1 + aend + 1), wpos - p1 - static_cast<uint32_t>(aend) - 1u);
            ql_ = rl_app(c, q_corr, ql_, rl_src_l(rl_rec_qb(c, hb)), rl_p_qlen(c, rl_h_w(hb)));
            ql_ = rl_app_fill(c, q_corr, ql_, q_min, wpos - p1 - static_cast<uint32_t>(aend) - 1u);
            if (c.fail()) return;
            rl_bm_add_range(c, res.bm, p1 - first_pos, p1 + static_cast<uint32_t>(aend) + 1u - first_pos);
            p1 = wpos; um1 = rl_an_um(v_w, lw_lo + i_w_s);
            len_weak_region = p2 - p1 + k;
            c.top(0) = 0; partial = RL_NOH; // paths of the previous attempt are dead
            do_call = true;
        }
    }
    if (c.fail()) return;
    if (!found_first) {
        if (complete != RL_NOH) {
            const uint32_t pl = rl_rec_to_string(c, complete, RL_SB_PATH); if (pl == 0xFFFFFFFFu) return;
            n_amb = rl_amb_collect(c, complete, sl_, n_amb);
            sl_ = rl_app(c, s_corr, sl_, rl_src_l(rl_sb(RL_SB_PATH)), pl);
            ql_ = rl_app(c, q_corr, ql_, rl_src_l(rl_rec_qb(c, complete)), rl_p_qlen(c, rl_h_w(complete)));
            rl_bm_add_range(c, res.bm, p1 - first_pos, p2 - first_pos + k);
        } else if (partial != RL_NOH) {
            int aid, aend;
            rl_st(c, RL_OFF_V, partial);
            rl_select_best(c, RL_OFF_V, 1, rl_src_g(s_read + p1), len_weak_region, RTK_MODE_SHW, c.o()->weak_region_len_factor, &aid, &aend);
            if (c.fail()) return;
            if (aid == -1) add_uncorrected(p1, len_weak_region, q_min);
            else {
                const uint32_t hb = partial;
                const uint32_t pl = rl_rec_to_string(c, hb, RL_SB_PATH); if (pl == 0xFFFFFFFFu) return;
                n_amb = rl_amb_collect(c, hb, sl_, n_amb);
                sl_ = rl_app(c, s_corr, sl_, rl_src_l(rl_sb(RL_SB_PATH)), pl);
                const uint32_t rest = len_weak_region - static_cast<uint32_t>(aend) - 1u;
                sl_ = rl_app(c, s_corr, sl_, rl_src_g(s_read + p1 + aend + 1), rest);
                ql_ = rl_app(c, q_corr, ql_, rl_src_l(rl_rec_qb(c, hb)), rl_p_qlen(c, rl_h_w(hb)));
                ql_ = rl_app_fill(c, q_corr, ql_, q_min, rest);
                rl_bm_add_range(c, res.bm, p1 - first_pos, p1 + static_cast<uint32_t>(aend) + 1u - first_pos);
            }
        } else if (sl_ != 0) add_uncorrected(p1, len_weak_region, q_min);
        else { sl_ = 0; ql_ = 0; add_uncorrected(first_pos, len_weak_region, q_min); } // setUncorrected
    } else {
        const uint32_t pl = rl_rec_to_string(c, complete, RL_SB_PATH); if (pl == 0xFFFFFFFFu) return;
        sl_ = 0; ql_ = 0;
        n_amb = rl_amb_collect(c, complete, 0, n_amb);
        sl_ = rl_app(c, s_corr, sl_, rl_src_l(rl_sb(RL_SB_PATH)), pl);
        ql_ = rl_app(c, q_corr, ql_, rl_src_l(rl_rec_qb(c, complete)), rl_p_qlen(c, rl_h_w(complete)));
        rl_bm_add_range(c, res.bm, 0, len_weak_region);
    }
    if (c.fail()) return;
    if (n_amb != 0) { rl_fix_ambiguity(c, res.sb_seq, sl_, res.sb_qual, ql_, rl_src_g(s_read + first_pos), res.old_len, n_amb); if (c.fail()) return; } // :716
    if (rl_bm_card(c, res.bm, res.old_len) == res.old_len) { // :718-725 (G20): last k-mer of the WHOLE read vs last k-mer of the corrected region
        bool same = sl_ >= k && s_len >= k;
        for (uint32_t i = 0; same && i < k; ++i) same = rtk_bifrost_code(s_read[s_len - k + i]) == rtk_bifrost_code(static_cast<char>(rl_ldb(c, s_corr + sl_ - k + i)));
        if (same) res.is_corrected = 1;
    }
    if (!res.is_corrected) { // :727-747 trim the corrected string to the largest SHW end location of the raw region
        RL_LAP(c, RL_P_ASSEMBLE); RL_SETCUR(c, RL_P_TRIM);
        const RlAln a = rl_myers(c, rl_src_g(s_read + first_pos), static_cast<int>(p2 - first_pos + k), rl_src_l(s_corr), static_cast<int>(sl_), -1, RTK_MODE_SHW, true, false, nullptr);
        if (c.fail()) return;
        if (a.dist >= 0) {
            const uint32_t keep = (a.first == -1) ? 0u : static_cast<uint32_t>(a.last + 1); // endLocations[0] == -1 wraps to SIZE_MAX in the reference
            if (keep < sl_) sl_ = keep;
            if (keep < ql_) ql_ = keep;
        }
    }
    res.seq_len = sl_; res.qual_len = ql_;
    RL_CTX_END(c);
}

// ------------------------------------------------------------------------------------------------ generateConsensus (src/Alignment.cpp:309-470)
struct RlCig { uint32_t mvb, off, n, idx, qpos, rpos; }; // op-granular cursor over an alignment (moves 0 / 3 = M, 1 = I, 2 = D)
RTK_DEV char rl_mv_op(unsigned char m) { return (m == 1) ? 'I' : (m == 2 ? 'D' : 'M'); }
RTK_DEV uint32_t rl_op_len(const RlCtx& c, const RlCig& cc) { const char op = rl_mv_op(rl_ldb(c, cc.mvb + cc.off + cc.idx)); uint32_t j = cc.idx + 1u; while (j < cc.n && rl_mv_op(rl_ldb(c, cc.mvb + cc.off + j)) == op) ++j; return j - cc.idx; }
RTK_DEV void rl_move_into_cigar(const RlCtx& c, uint32_t start, uint32_t end, RlCig& cc, uint32_t* rs, uint32_t* re, uint32_t* ref_out) { // moveIntoCIGAR (:354-411)
    uint32_t read_pos_start = cc.qpos, read_pos_end;
    while (cc.idx != cc.n && cc.rpos < start) {
        const uint32_t l = rl_op_len(c, cc); const char op = rl_mv_op(rl_ldb(c, cc.mvb + cc.off + cc.idx));
        if (op == 'M') { if (cc.rpos + l > start) { read_pos_start = cc.qpos + (start - cc.rpos); break; } cc.qpos += l; cc.rpos += l; }
        else if (op == 'I') cc.qpos += l; else cc.rpos += l;
        cc.idx += l; read_pos_start = cc.qpos;
    }
    read_pos_end = read_pos_start;
    while (cc.idx != cc.n && cc.rpos < end) {
        const uint32_t l = rl_op_len(c, cc); const char op = rl_mv_op(rl_ldb(c, cc.mvb + cc.off + cc.idx));
        if (op == 'M') { if (cc.rpos + l > end) { *rs = read_pos_start; *re = cc.qpos + (end - cc.rpos); *ref_out = end; return; } cc.qpos += l; cc.rpos += l; }
        else if (op == 'I') cc.qpos += l; else cc.rpos += l;
        cc.idx += l; read_pos_end = cc.qpos;
    }
    *rs = read_pos_start; *re = read_pos_end; *ref_out = cc.rpos;
}
RTK_DEV bool rl_str_equal(const RlCtx& c, uint32_t a_b, uint32_t b_b, uint32_t n) { for (uint32_t i = 0; i < n; ++i) if (rl_ldb(c, a_b + i) != rl_ldb(c, b_b + i)) return false; return true; }

// the consensus into string buffers RL_SB_CS / RL_SB_CQ; false: "empty" (the caller falls back to the raw region)
RTK_FN bool rl_generate_consensus(RlCtx& c, const RlRes* fw, const RlRes* bw, const char* ref, uint32_t ref_len, double max_norm, uint32_t* out_sl, uint32_t* out_ql) {
    *out_sl = 0; *out_ql = 0;
    RL_LAP(c, RL_P_DRIVER); RL_SETCUR(c, RL_P_CONSENSUS);
    const uint32_t out_s = rl_sb(RL_SB_CS), out_q = rl_sb(RL_SB_CQ);
    const uint32_t nfw = rl_bm_card(c, fw->bm, fw->old_len), nbw = rl_bm_card(c, bw->bm, bw->old_len);
    auto take = [&](const RlRes* r) { *out_sl = rl_app(c, out_s, *out_sl, rl_src_l(rl_sb(r->sb_seq)), r->seq_len); *out_ql = rl_app(c, out_q, *out_ql, rl_src_l(rl_sb(r->sb_qual)), r->qual_len); return true; };
    if (nbw == 0 && nfw != 0) return take(fw);
    else if (nfw == 0 && nbw != 0) return take(bw);
    else if (nfw + nbw == 0) return false;
    if (nbw > nfw) { const RlRes* t = fw; fw = bw; bw = t; }
    const RlSrc rsrc = rl_src_g(ref);
    uint32_t nm_fw = 0, nm_bw = 0, off_fw = RL_MV_BYTES, off_bw = RL_MV_BYTES;
    const RlAln afw = rl_align_path(c, rl_src_l(rl_sb(fw->sb_seq)), static_cast<int>(fw->seq_len), rsrc, static_cast<int>(ref_len), RTK_MODE_NW, rl_mvb(1), &off_fw, &nm_fw);
    if (c.fail()) return false;
    // both directions usually arrive at the same corrected string: its alignment against the raw region is then the one just computed
    const bool same_strings = bw->seq_len == fw->seq_len && rl_str_equal(c, rl_sb(bw->sb_seq), rl_sb(fw->sb_seq), fw->seq_len);
    RlAln abw = afw; uint32_t mvb_bw = rl_mvb(2);
    if (same_strings) { nm_bw = nm_fw; off_bw = off_fw; mvb_bw = rl_mvb(1); }
    else { abw = rl_align_path(c, rl_src_l(rl_sb(bw->sb_seq)), static_cast<int>(bw->seq_len), rsrc, static_cast<int>(ref_len), RTK_MODE_NW, rl_mvb(2), &off_bw, &nm_bw); if (c.fail()) return false; }
    const double n_fw = static_cast<double>(afw.dist) / static_cast<double>(fw->seq_len > ref_len ? fw->seq_len : ref_len);
    const double n_bw = static_cast<double>(abw.dist) / static_cast<double>(bw->seq_len > ref_len ? bw->seq_len : ref_len);
    if (max_norm > 0.0 && (n_fw > max_norm || n_bw > max_norm)) {
        if (n_fw > max_norm && n_bw > max_norm) return false;
        if (n_fw > max_norm) return take(bw);
        return take(fw);
    }
    RlCig cf, cb;
    cf.mvb = rl_mvb(1); cf.off = off_fw; cf.n = nm_fw; cf.idx = 0; cf.qpos = 0; cf.rpos = 0;
    cb.mvb = mvb_bw; cb.off = off_bw; cb.n = nm_bw; cb.idx = 0; cb.qpos = 0; cb.rpos = 0;
    uint32_t i = 0;
    while (i < ref_len && !c.fail()) {
        int64_t len_fw = rl_len_corrected(c, *fw, i), len_bw = rl_len_corrected(c, *bw, i);
        if ((len_fw + len_bw) <= 0) {
            len_fw = rl_len_uncorrected(c, *fw, i); len_bw = rl_len_uncorrected(c, *bw, i);
            if (len_fw > len_bw || len_fw <= 0) len_fw = -1; else len_bw = -1;
        }
        uint32_t rs, re, rout;
        const RlRes* src;
        if (len_fw >= len_bw) { rl_move_into_cigar(c, i, static_cast<uint32_t>(static_cast<int64_t>(i) + len_fw), cf, &rs, &re, &rout); src = fw; }
        else { rl_move_into_cigar(c, i, static_cast<uint32_t>(static_cast<int64_t>(i) + len_bw), cb, &rs, &re, &rout); src = bw; }
        if (re > rs) {
            *out_sl = rl_app(c, out_s, *out_sl, rl_src_l(rl_sb(src->sb_seq) + rs), (rs < src->seq_len) ? ((re - rs) < (src->seq_len - rs) ? (re - rs) : (src->seq_len - rs)) : 0u);
            *out_ql = rl_app(c, out_q, *out_ql, rl_src_l(rl_sb(src->sb_qual) + rs), (rs < src->qual_len) ? ((re - rs) < (src->qual_len - rs) ? (re - rs) : (src->qual_len - rs)) : 0u);
        }
        if (rout == i) { rl_fail(c, RL_F_OTHER); return false; } // no progress: would loop forever in the reference as well
        i = rout;
    }
    if (max_norm > 0.0 && !c.fail()) {
        // the merged string is very often one of the two inputs again: its distance (plain equalities, :460) is then the one computed above when
        // both strings hold A / C / G / T only, and that distance already passed the max_norm test
        const bool is_fw = *out_sl == fw->seq_len && rl_str_equal(c, out_s, rl_sb(fw->sb_seq), fw->seq_len);
        const bool is_bw = !is_fw && *out_sl == bw->seq_len && rl_str_equal(c, out_s, rl_sb(bw->sb_seq), bw->seq_len);
        if (is_fw || is_bw) {
            bool clean = true;
            for (uint32_t x = 0; x < *out_sl && clean; ++x) { const unsigned char ch = rl_ldb(c, out_s + x); clean = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }
            for (uint32_t x = 0; x < ref_len && clean; ++x) { const char ch = ref[x]; clean = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }
            if (clean) return true;
        }
        const RlAln a = rl_myers(c, rl_src_l(out_s), static_cast<int>(*out_sl), rsrc, static_cast<int>(ref_len), -1, RTK_MODE_NW, /*iupac=*/false, false, nullptr); // edlibDefaultAlignConfig (:460)
        if (c.fail()) return false;
        const double nn = static_cast<double>(a.dist) / static_cast<double>(*out_sl > ref_len ? *out_sl : ref_len);
        if (nn > max_norm) { *out_sl = 0; *out_ql = 0; return take(fw); }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ one gap (src/Correction.cpp:803-935), pass 1
// The region's segment lands in string buffers RL_SB_OUTS / RL_SB_OUTQ (*osl / *oql characters); c.fail() != 0: handed on, nothing to emit.
RTK_FN void rl_region_program(RlCtx& c, const RegionDesc* rd, uint32_t* osl_, uint32_t* oql_) {
    const BatchView& bv = *c.bv(); const RegionBatch& rb = *c.rb();
    *osl_ = 0; *oql_ = 0;
    const uint32_t r = rd->read, k = c.k();
    const uint64_t base = bv.roff.get()[r];
    const uint32_t L = static_cast<uint32_t>(bv.roff.get()[r + 1] - base);
    const char* s_fw = bv.seq.get() + base; const char* s_bw = rb.seq_rc.get() + base;
    const uint64_t mq = static_cast<uint64_t>(static_cast<int32_t>(c.o()->max_qual));
    const char q_min = rtk_get_qual(0.0, 0, mq), q_max = rtk_get_qual(1.0, 0, mq);
    RlAnch so, we, so_r, we_r;
    so.pos = bv.s_pos.get() + base; so.hit = nullptr; so.hits_by_pos = bv.hits.get() + base; so.n = bv.n_solid.get()[r]; so.L = L; so.rev = 0; so.k = k;
    { const uint64_t wo = bv.w_off.get()[r]; we.pos = bv.wk_pos.get() + wo; we.hit = bv.wk_hit.get() + wo; } we.hits_by_pos = nullptr; we.n = bv.w_cnt.get()[r]; we.L = L; we.rev = 0; we.k = k;
    so_r = so; so_r.rev = 1; we_r = we; we_r.rev = 1;
    if (rd->kind != RTK_RG_GAP) { rl_fail(c, RL_F_OTHER); return; }
    const uint32_t i = rd->i_solid, prev_pos = rd->prev_pos;
    const uint32_t pa = so.pos[i], pb = so.pos[i + 1];
    if (pb < pa + k) { rl_fail(c, RL_F_OTHER); return; } // (such gaps, and the same-unitig shortcut, are written by k_regions_easy)
    const uint32_t i_weak = rl_an_first_ge(we, 0, we.n, pa); // first weak anchor at or after the left solid anchor (:801)
    RlRes fw, bw;
    fw.sb_seq = RL_SB_FWS; fw.sb_qual = RL_SB_FWQ; fw.bm = RL_OFF_BM; bw.sb_seq = RL_SB_BWS; bw.sb_qual = RL_SB_BWQ; bw.bm = RL_OFF_BM + RL_BM_W;
    uint32_t osl = 0, oql = 0;
    const uint32_t out_s = rl_sb(RL_SB_OUTS), out_q = rl_sb(RL_SB_OUTQ);
    rl_correct_region(c, s_fw, L, so, we, i, i_weak, false, fw);
    if (c.fail()) return;
    const uint32_t l_solid = pa - prev_pos;
    auto emit_minus_k = [&](uint32_t sb_s, uint32_t sl, uint32_t sb_q, uint32_t ql) { // (prefix + x).substr(0, len - k)
        const uint32_t ts = l_solid + sl, tq = l_solid + ql;
        const uint32_t ks = ts >= k ? ts - k : ts, kq = tq >= k ? tq - k : tq;
        osl = rl_app(c, out_s, osl, rl_src_g(s_fw + prev_pos), ks < l_solid ? ks : l_solid); if (ks > l_solid) osl = rl_app(c, out_s, osl, rl_src_l(rl_sb(sb_s)), ks - l_solid);
        oql = rl_app_fill(c, out_q, oql, q_max, kq < l_solid ? kq : l_solid); if (kq > l_solid) oql = rl_app(c, out_q, oql, rl_src_l(rl_sb(sb_q)), kq - l_solid);
    };
    if (fw.is_corrected) emit_minus_k(fw.sb_seq, fw.seq_len, fw.sb_qual, fw.qual_len);
    else {
        const uint32_t i_solid_bw = so.n - i - 2u;
        uint32_t i_weak_bw = we.n - i_weak;
        i_weak_bw = rl_an_first_gt(we_r, 0, i_weak_bw, rl_an_pos(so_r, i_solid_bw));
        rl_correct_region(c, s_bw, L, so_r, we_r, i_solid_bw, i_weak_bw, true, bw);
        if (c.fail()) return;
        rl_res_reverse_complement(c, bw);
        if (bw.is_corrected) emit_minus_k(bw.sb_seq, bw.seq_len, bw.sb_qual, bw.qual_len);
        else {
            const uint32_t ref_len = pb - pa + k;
            uint32_t csl = 0, cql = 0;
            const bool ok = rl_generate_consensus(c, &fw, &bw, s_fw + pa, ref_len, c.o()->weak_region_len_factor, &csl, &cql);
            RL_LAP(c, RL_P_CONSENSUS); RL_SETCUR(c, RL_P_DRIVER);
            if (c.fail()) return;
            if (!ok || csl == 0) { // raw region, k solid qualities then minimum quality (:898-904)
                csl = rl_app(c, rl_sb(RL_SB_CS), 0, rl_src_g(s_fw + pa), ref_len);
                cql = rl_app_fill(c, rl_sb(RL_SB_CQ), 0, q_max, k); cql = rl_app_fill(c, rl_sb(RL_SB_CQ), cql, q_min, pb - pa);
            }
            emit_minus_k(RL_SB_CS, csl, RL_SB_CQ, cql);
        }
    }
    if (c.fail()) return;
    *osl_ = osl; *oql_ = oql;
}

#endif
