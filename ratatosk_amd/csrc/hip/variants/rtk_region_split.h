// Measured-and-rejected variant (-DRTK_REGION_SPLIT, end of round 3; DESIGN_HISTORY.md 9.1): rtk_correct_region as three non-inlined programs that share nothing on the wave's
// stack (side lists + colours | path search | assembly + trim), their common state in the RegionCall record of the LDS header. Same results on the simulator and GPU tiers;
// k_regions 31.0 -> 33.1 ms on configs[1] (the calls cost more callee-saved register rows than the smaller frames return). Included by rtk_region.h inside its own
// #ifdef: not part of the default build.
// The call in three programs that share nothing on the wave's stack (their common state is the RegionCall record in the LDS header): each gets
// its own frame and register allocation, and the frames do not nest.
RTK_FN uint32_t rtk_region_colours(const RCtx& c_, const Anchors& v_s_, const Anchors& v_w_, uint32_t i_s_, uint32_t i_w_) { // side lists + chooseColors (:473-587)
    const RCtx& c = *rtk_u(&c_); const Anchors& v_s = *rtk_u(&v_s_); const Anchors& v_w = *rtk_u(&v_w_); RTK_ASSUME_LDS(&v_s); RTK_ASSUME_LDS(&v_w); const uint32_t i_s = rtk_u(i_s_), i_w = rtk_u(i_w_);
    RegionScratch& s = rtk_hdr(c); const GraphView& g = c.g; RegionCall& st = s.loc.call;
    const bool has_end_pt = st.has_end_pt != 0; const uint32_t p2 = st.p2, first_pos = st.first_pos, s_len = st.s_len, lw_lo = st.lw_lo, lw_hi = st.lw_hi;
    const uint64_t u_min_start = static_cast<uint64_t>(st.p1) - static_cast<uint64_t>(c.o.insert_sz), u_min_end = static_cast<uint64_t>(p2) + static_cast<uint64_t>(c.o.insert_sz);
    uint32_t n_all = 0;
    {
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
        if (sl.n >= cap / 2 || sr.n >= cap / 2 || sm.n >= cap / 2) { rtk_fail_ovf(s, 8); return 0; }
        s.fine[7] += rtk_clock() - t_side0;
        { const unsigned long long t0 = rtk_clock(); n_all = rtk_u(rtk_choose_colors(c, sl, sr, sm)); s.cnt[5] += rtk_clock() - t0; }
        if (rtk_failed(s)) return 0;
        // keep all_pids for the reverse-complement call (rc = &fw): set[0] is preserved by everything below

    }
    return n_all;
}
RTK_FN void rtk_region_search(const RCtx& c_, const Anchors& v_w_, ResCorr& res_) { // extractSemiWeakPaths and its restarts (:609-711)
    const RCtx& c = *rtk_u(&c_); const Anchors& v_w = *rtk_u(&v_w_); RTK_ASSUME_LDS(&v_w); ResCorr& res = *rtk_u(&res_); RTK_ASSUME_LDS(&res);
    RegionScratch& s = rtk_hdr(c); RegionCall& st = s.loc.call; const uint32_t k = static_cast<uint32_t>(c.k);
    const char* s_read = st.s_read; const char* q_read = st.q_read; const bool lrc = st.lrc != 0; const uint32_t s_len = st.s_len;
    uint32_t p1 = st.p1; UMap um1 = st.um1; const uint32_t p2 = st.p2; const UMap um2 = st.um2; const uint32_t first_pos = st.first_pos; uint32_t len_weak_region = st.len_weak_region;
    const uint32_t lw_lo = st.lw_lo, lw_hi = st.lw_hi, n_all = st.n_all;
    const char q_min = rtk_get_qual(0.0, 0, static_cast<uint64_t>(c.o.max_qual));
    const uint32_t max_len_weak_anchors = c.o.long_read_correct ? c.o.max_len_weak_region2 : c.o.max_len_weak_region1;
    const uint32_t* all_pids = s.set[0];
    // ---- paths ----
    s.top[0] = 0;
    uint32_t n_partial = 0, n_amb = 0; // n_amb: size of v_ambiguity (list[RTK_L_AMB])
    uint64_t complete = ~0ull;
    char* s_corr = res.seq; char* q_corr = res.qual; uint32_t& sl_ = s.loc.len[4]; uint32_t& ql_ = s.loc.len[5]; sl_ = 0; ql_ = 0; // (lengths that rtk_app updates through a pointer: LDS words)
    const Anchors& lvw = v_w;
    const uint32_t nlw = lw_hi - lw_lo;
    auto clamp_len = [&](uint32_t pos, uint32_t len) -> uint32_t { return (pos + len <= s_len) ? len : (pos < s_len ? s_len - pos : 0); }; // std::string::substr
    // extractSemiWeakPaths from the left solid anchor (:613), then again from a weak anchor behind the best partial path as long as
    // there is one (:619-651): ONE call site, so that the whole search can be compiled into this function
    bool first_call = true, found_first = false, do_call = n_all >= c.o.min_cov_vertices;
    uint32_t i_w_s = 0;
    for (;;) {
        if (do_call) { const unsigned long long t0 = rtk_clock(); complete = rtk_u(rtk_extract_semi_weak(c, s_read, s_len, all_pids, n_all, p1, um1, p2, um2, lvw, lw_lo, lw_hi, first_call ? 0u : i_w_s, &n_partial)); n_partial = rtk_u(n_partial); s.cnt[6] += rtk_clock() - t0; }
        if (rtk_failed(s)) return;
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

    st.p1 = p1; st.um1 = um1; st.len_weak_region = len_weak_region; st.n_partial = n_partial; st.n_amb = n_amb; st.complete = complete; st.found_first = found_first ? 1u : 0u;
}
RTK_FN void rtk_region_assemble(const RCtx& c_, ResCorr& res_) { // result strings, fixAmbiguity, trim (:655-753)
    const RCtx& c = *rtk_u(&c_); ResCorr& res = *rtk_u(&res_); RTK_ASSUME_LDS(&res);
    RegionScratch& s = rtk_hdr(c); RegionCall& st = s.loc.call; const uint32_t k = static_cast<uint32_t>(c.k);
    const char* s_read = st.s_read; const char* q_read = st.q_read; const bool lrc = st.lrc != 0; const uint32_t s_len = st.s_len;
    const uint32_t p1 = st.p1, p2 = st.p2, first_pos = st.first_pos, len_weak_region = st.len_weak_region, n_partial = st.n_partial; uint32_t n_amb = st.n_amb;
    const uint64_t complete = st.complete; const bool found_first = st.found_first != 0;
    const char q_min = rtk_get_qual(0.0, 0, static_cast<uint64_t>(c.o.max_qual));
    char* s_corr = res.seq; char* q_corr = res.qual; uint32_t& sl_ = s.loc.len[4]; uint32_t& ql_ = s.loc.len[5];
    auto clamp_len = [&](uint32_t pos, uint32_t len) -> uint32_t { return (pos + len <= s_len) ? len : (pos < s_len ? s_len - pos : 0); }; // std::string::substr
    auto add_uncorrected = [&](uint32_t pos, uint32_t len, char q) { rtk_app(s, s_corr, &sl_, s_read + pos, clamp_len(pos, len));
        if (lrc) rtk_app(s, q_corr, &ql_, q_read + pos, clamp_len(pos, len)); else rtk_app_fill(s, q_corr, &ql_, q, len_weak_region); }; // :459-469
    if (rtk_failed(s)) return;
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
    if (n_amb != 0) { const unsigned long long ta0 = rtk_clock(); rtk_fix_ambiguity(c, s_corr, sl_, q_corr, ql_, s_read + first_pos, res.old_len, n_amb); s.fine[9] += rtk_clock() - ta0; if (rtk_failed(s)) return; } // :716
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
}


