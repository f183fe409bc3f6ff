// SNP annotations on the chosen paths of a region: getAmbiguityVector (src/GraphTraversal.cpp:966-1036) and fixAmbiguity
// (src/Alignment.cpp:527-844), for an undetermined haplotype (no phasing input: the isValidHap tests of :741,:801 are
// short-circuited). Included by rtk_region.h; everything here is wave-uniform bookkeeping over a handful of annotated positions,
// except the walk over the alignment moves, which is done 64 moves at a time.
//
// Small sets are kept as arrays of `position << 8 | character` in the region scratch lists:
//   list[RTK_L_AMB]      v_ambiguity of the region (path-string coordinates of s_corrected)
//   list[RTK_L_AMB + 1]  m_ambiguity_safe   (while collecting: the running vector of one path)
//   list[RTK_L_AMB + 2]  m_ambiguity_all    (while collecting: merge buffer)
//   list[RTK_L_AMB + 3]  annotations of one unitig mapping
//   list[RTK_L_AMB + 4]  s_ambiguity (alleles of linked SNPs)
#ifndef RTK_AMBIGUITY_H
#define RTK_AMBIGUITY_H

#define RTK_L_AMB 6

RTK_DEV uint64_t rtk_amb_mk(uint64_t pos, char ch) { return (pos << 8) | static_cast<uint64_t>(static_cast<unsigned char>(ch)); }
RTK_DEV uint32_t rtk_amb_pos(uint64_t e) { return static_cast<uint32_t>(e >> 8); }
RTK_DEV char rtk_amb_chr(uint64_t e) { return static_cast<char>(e & 0xFFull); }

// IUPAC code <-> base set (bit0 A, bit1 C, bit2 G, bit3 T), src/Common.hpp:260,351-400
RTK_DEV uint32_t rtk_iupac_idx(char c) {
    const int cl = rtk_cls(static_cast<unsigned char>(c & 0xDF));
    if (cl < 4) return 1u << cl;
    if (cl >= 15) return 0u;
    // base sets of M R S V W Y H K D B N = 3 5 6 7 9 A B C D E F, one nibble each, M lowest
    return static_cast<uint32_t>((0xFEDCBA97653ull >> (4 * (cl - 4))) & 0xFull);
}
RTK_DEV char rtk_iupac_chr(uint32_t i) { // ".ACMGRSVTWYHKDBN"[i]
    const uint64_t w = (i & 8u) ? 0x4E42444B48595754ull /* T W Y H K D B N */ : 0x565352474D43412Eull /* . A C M G R S V */;
    return static_cast<char>((w >> (8 * (i & 7u))) & 0xFFull);
}
RTK_DEV char rtk_iupac_comp(char c) { // Bifrost reverse_complement(char): swap A<->T and C<->G in the base set; foreign bytes unchanged
    const uint32_t i = rtk_iupac_idx(c);
    if (i == 0) return c;
    return rtk_iupac_chr(((i & 1u) << 3) | ((i & 8u) >> 3) | ((i & 2u) << 1) | ((i & 4u) >> 1));
}
RTK_DEV bool rtk_is_dna(char c) { const char u = static_cast<char>(c & 0xDF); return u == 'A' || u == 'C' || u == 'G' || u == 'T'; }
RTK_DEV bool rtk_iupac_overlap(char a, char b) { return (rtk_iupac_idx(a) & rtk_iupac_idx(b)) != 0u; }

RTK_DEV int rtk_amb_find(const uint64_t* a, uint32_t n, uint32_t pos) { for (uint32_t i = 0; i < n; ++i) if (rtk_amb_pos(a[i]) == pos) return static_cast<int>(i); return -1; }
RTK_DEV char rtk_unitig_char(const GraphView& g, uint32_t u, uint32_t i) { return static_cast<char>((0x54474341u >> (8 * rtk_base(g, g.uoff[u] + i))) & 0xFFu); }

// UnitigData::get_ambiguity_char(um) (UnitigData.hpp:458-481): annotations inside the mapping, in mapping coordinates and orientation
RTK_DEV uint32_t rtk_amb_of_um(const RCtx& c, const UMap& um, uint64_t* out, uint32_t cap) {
    const GraphView& g = c.g;
    const uint64_t* ent = g.amb.get() + (static_cast<uint64_t>(g.n_unitigs) + 1);
    const uint64_t a0 = g.amb[um.unitig], a1 = g.amb[um.unitig + 1];
    const uint32_t sz = um.len + static_cast<uint32_t>(c.k) - 1, end = um.dist + sz;
    uint32_t n = 0;
    for (uint64_t j = 0; j < a1 - a0; ++j) {
        const uint64_t e = ent[um.strand ? (a0 + j) : (a1 - 1 - j)];
        const uint32_t pos = static_cast<uint32_t>(e >> 4); const char ch = rtk_iupac_chr(static_cast<uint32_t>(e & 15ull));
        if (pos < um.dist || pos >= end) continue;
        if (n >= cap) { rtk_fail_ovf(*c.sc, 11); return 0; }
        out[n++] = um.strand ? rtk_amb_mk(pos - um.dist, ch) : rtk_amb_mk(sz - (pos - um.dist) - 1, rtk_iupac_comp(ch));
    }
    return n;
}

// getAmbiguityVector(path) + the `v_ambiguity.push_back({offset + pos, c})` of its callers (src/Correction.cpp:635-637 etc.)
RTK_FN uint32_t rtk_amb_collect(const RCtx& c_, uint64_t h_, uint32_t offset_, uint32_t n_amb_) { // returns the new size of v_ambiguity
    const RCtx& c = *rtk_u(&c_); const uint64_t h = rtk_u(h_); const uint32_t offset = rtk_u(offset_), n_amb = rtk_u(n_amb_);
    if (c.g.n_amb == 0) return n_amb;
    RegionScratch& s = rtk_hdr(c);
    const UMap* ums = rtk_path_ums(s, rtk_h_lvl(h), rtk_h_off(h));
    const uint32_t n = rtk_rec_n(s, h), k1 = static_cast<uint32_t>(c.k) - 1, cap = s.list_cap;
    { // nothing to do unless some unitig of the path is annotated (one lane per unitig)
        bool any = false;
        for (uint32_t x0 = 0; x0 < n && !any; x0 += RTK_WAVE) {
            const uint32_t x = x0 + static_cast<uint32_t>(rtk_lane());
            bool mine = false;
            if (x < n) { const uint32_t u = ums[x].unitig; mine = c.g.amb[u + 1] != c.g.amb[u]; }
            any = rtk_ballot(mine) != 0;
        }
        if (!any) return n_amb;
    }
    uint64_t* va = s.list[RTK_L_AMB + 1]; uint64_t* vt = s.list[RTK_L_AMB + 2]; uint64_t* vu = s.list[RTK_L_AMB + 3];
    uint32_t nva = 0, prev_l = 0, pos_prev_l = 0;
    for (uint32_t x = 0; x < n; ++x) {
        const UMap um = rtk_u(ums[x]);
        const uint32_t nvu = rtk_amb_of_um(c, um, vu, cap);
        if (rtk_failed(s)) return n_amb;
        uint32_t nvt = 0, ip = pos_prev_l, ic = 0;
        // annotations inside the k-1 characters shared with the previous unitig are merged (same position: union of the alleles)
        while (ip != nva && ic != nvu && rtk_amb_pos(vu[ic]) < k1 && nvt < cap) {
            const uint32_t cur_pos = rtk_amb_pos(vu[ic]) + prev_l, pp = rtk_amb_pos(va[ip]);
            if (pp < cur_pos) vt[nvt++] = va[ip++];
            else if (pp > cur_pos) vt[nvt++] = rtk_amb_mk(cur_pos, rtk_amb_chr(vu[ic++]));
            else { vt[nvt++] = rtk_amb_mk(pp, rtk_iupac_chr(rtk_iupac_idx(rtk_amb_chr(va[ip])) | rtk_iupac_idx(rtk_amb_chr(vu[ic])))); ++ip; ++ic; }
        }
        if (nvt + (nva - ip) + (nvu - ic) > cap || pos_prev_l + nvt + (nva - ip) + (nvu - ic) > cap) { rtk_fail_ovf(s, 11); return n_amb; }
        for (; ip != nva; ++ip) vt[nvt++] = va[ip];
        for (; ic != nvu; ++ic) vt[nvt++] = rtk_amb_mk(rtk_amb_pos(vu[ic]) + prev_l, rtk_amb_chr(vu[ic]));
        prev_l += um.len;
        nva = pos_prev_l;
        for (uint32_t i = 0; i < nvt; ++i) { va[nva++] = vt[i]; pos_prev_l += (rtk_amb_pos(vt[i]) < prev_l) ? 1u : 0u; }
    }
    uint64_t* v = s.list[RTK_L_AMB];
    uint32_t na = n_amb;
    if (na + nva > cap) { rtk_fail_ovf(s, 11); return n_amb; }
    for (uint32_t i = 0; i < nva; ++i) v[na++] = rtk_amb_mk(static_cast<uint64_t>(offset) + rtk_amb_pos(va[i]), rtk_amb_chr(va[i]));
    return na;
}

// Bifrost findUnitig(s, pos, len) [A7]: the k-mer at s+pos extended along its unitig while s keeps agreeing; on the reverse strand
// the match runs towards the unitig head and the mapping starts at its lowest forward offset
RTK_DEV UMap rtk_find_unitig(const RCtx& c, const char* str, uint32_t pos, uint32_t len, const RtkKm& fw) {
    const GraphView& g = c.g; const uint32_t k = static_cast<uint32_t>(c.k);
    const uint64_t hit = rtk_find_km(g, fw, nullptr);
    if (hit == RTK_NO_HIT) return rtk_um_empty();
    UMap um = rtk_unpack_hit(hit);
    const uint32_t ul = rtk_ulen(g, um.unitig);
    uint32_t j = pos + k, n = 1;
    if (um.strand) { uint32_t up = um.dist + k; while (j < len && up < ul && str[j] == rtk_unitig_char(g, um.unitig, up)) { ++j; ++up; ++n; } }
    else { int64_t up = static_cast<int64_t>(um.dist) - 1; while (j < len && up >= 0 && str[j] == rtk_iupac_comp(rtk_unitig_char(g, um.unitig, static_cast<uint32_t>(up)))) { ++j; --up; ++n; } um.dist -= (n - 1); }
    um.len = n;
    return um;
}

// The same from a k-mer hit that is already known (the windows of a string are looked up in one lane-parallel batch by the caller);
// the characters behind the k-mer are compared 64 at a time instead of one dependent load per character.
RTK_DEV UMap rtk_extend_hit(const RCtx& c, uint64_t hit, const char* str, uint32_t pos, uint32_t len) {
    const GraphView& g = c.g; const uint32_t k = static_cast<uint32_t>(c.k);
    UMap um = rtk_unpack_hit(hit);
    const uint32_t ul = rtk_ulen(g, um.unitig);
    const uint32_t j0 = pos + k;
    // characters available on both sides
    const uint32_t room_s = len > j0 ? len - j0 : 0u;
    const uint32_t room_u = um.strand ? (ul > um.dist + k ? ul - (um.dist + k) : 0u) : um.dist;
    const uint32_t room = room_s < room_u ? room_s : room_u;
    uint32_t n = 0; bool stop = false;
    for (uint32_t b0 = 0; b0 < room && !stop; b0 += RTK_WAVE) {
        const uint32_t i = b0 + static_cast<uint32_t>(rtk_lane());
        bool bad = false;
        if (i < room) {
            const char uc = um.strand ? rtk_unitig_char(g, um.unitig, um.dist + k + i) : rtk_iupac_comp(rtk_unitig_char(g, um.unitig, um.dist - 1 - i));
            bad = str[j0 + i] != uc;
        }
        const uint64_t bb = rtk_ballot(bad);
        if (bb) { n += static_cast<uint32_t>(rtk_ffs(bb) - 1); stop = true; } else n += (room - b0) < RTK_WAVE ? (room - b0) : RTK_WAVE;
    }
    if (!um.strand) um.dist -= n;
    um.len = n + 1;
    return um;
}

// fixAmbiguity (src/Alignment.cpp:527-844). query/quality = s_corrected/q_corrected of the region (same length), ref = the raw region.
RTK_FN void rtk_fix_ambiguity(const RCtx& c_, char* query_, uint32_t query_len_, char* quality_, uint32_t quality_len_, const char* ref_, uint32_t ref_len_, uint32_t n_amb_) {
    const RCtx& c = *rtk_u(&c_); char* query = rtk_u(query_); char* quality = rtk_u(quality_); const char* ref = rtk_u(ref_);
    const uint32_t query_len = rtk_u(query_len_), quality_len = rtk_u(quality_len_), ref_len = rtk_u(ref_len_), n_amb = rtk_u(n_amb_);
    if (n_amb == 0) return;
    RegionScratch& s = rtk_hdr(c);
    const GraphView& g = c.g;
    const uint32_t k = static_cast<uint32_t>(c.k), cap = s.list_cap;
    if (quality_len < query_len || query_len > s.str_cap) { rtk_fail_ovf(s, 11); return; }
    const char q_max_corr = rtk_get_qual(1.0, static_cast<uint64_t>(c.o.out_qual), static_cast<uint64_t>(c.o.max_qual));
    const char q_min_corr = rtk_get_qual(0.0, static_cast<uint64_t>(c.o.out_qual), static_cast<uint64_t>(c.o.max_qual));
    const char q_min_conf_corr = rtk_get_qual(c.o.min_confidence_snp_corr, 0, static_cast<uint64_t>(c.o.max_qual));
    const char c_no = 'X';
    const uint64_t* v = s.list[RTK_L_AMB];
    uint64_t* ms = s.list[RTK_L_AMB + 1]; uint64_t* ma = s.list[RTK_L_AMB + 2]; uint64_t* vu = s.list[RTK_L_AMB + 3]; uint64_t* sa = s.list[RTK_L_AMB + 4];
    uint32_t nms = 0, nma = 0, nsa = 0;
    for (uint32_t i = 0; i < n_amb; ++i) {
        const uint32_t p = rtk_amb_pos(v[i]);
        if (quality[p] < q_min_conf_corr && rtk_amb_find(ms, nms, p) < 0) ms[nms++] = v[i]; // n_amb <= cap
    }
    s.fine[10] += 1;
    if (nms == 0) {
        // every annotated base is confident: nothing enters the sets unless the alignment meets a non-ACGT character of the
        // corrected or the raw region (:630-678), and with both clean the whole call leaves query and quality as they are
        bool odd = false;
        for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < query_len; i += RTK_WAVE) odd |= !rtk_is_dna(query[i]);
        for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < ref_len; i += RTK_WAVE) odd |= !rtk_is_dna(ref[i]);
        if (rtk_ballot(odd) == 0) { s.fine[11] += 1; return; }
    }
    char* qt = s.str[0]; // query_tmp
    rtk_wcopy(qt, query, query_len);
    rtk_sync();
    for (uint32_t i = 0; i < n_amb; ++i) {
        const uint32_t p = rtk_amb_pos(v[i]);
        if (quality[p] < q_min_conf_corr) qt[p] = rtk_amb_chr(v[i]);
    }
    for (uint32_t i = 0; i < nms; ++i) ma[i] = ms[i];
    nma = nms;
    rtk_sync();
    uint32_t nm = 0;
    unsigned long long tfa_ = rtk_clock();
#define RTK_FA_LAP(i) { const unsigned long long tn_ = rtk_clock(); s.fine[i] += tn_ - tfa_; tfa_ = tn_; }
    RTK_SITE(17); rtk_align_path(c, qt, query_len, ref, ref_len, RTK_MODE_SHW, &nm); nm = rtk_u(nm);
    if (rtk_failed(s)) return;
    RTK_FA_LAP(1)
    { // walk of the alignment (:612-706); only moves touching a non-ACGT character on either side do anything
        const uint8_t* mv = rtk_ld(&s.my.moves);
        uint32_t qp = 0, rp = 0;
        for (uint32_t i0 = 0; i0 < nm; i0 += RTK_WAVE) {
            const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
            const uint8_t m = i < nm ? mv[i] : 255;
            const bool is_m = (m == 0 || m == 3), isq = is_m || m == 1, isr = is_m || m == 2;
            const uint64_t bq = rtk_ballot(isq), br = rtk_ballot(isr);
            const uint64_t lt = (1ull << rtk_lane()) - 1ull;
            const uint32_t myq = qp + static_cast<uint32_t>(rtk_popc(bq & lt)), myr = rp + static_cast<uint32_t>(rtk_popc(br & lt));
            const bool hot = (is_m && (!rtk_is_dna(qt[myq]) || !rtk_is_dna(ref[myr]))) || (m == 1 && !rtk_is_dna(qt[myq]));
            uint64_t todo = rtk_ballot(hot);
            while (todo) {
                const int l = rtk_ffs(todo) - 1; todo &= todo - 1;
                const uint64_t below = (1ull << l) - 1ull;
                const uint32_t q_pos = qp + static_cast<uint32_t>(rtk_popc(bq & below)), t_pos = rp + static_cast<uint32_t>(rtk_popc(br & below));
                const char qc = qt[q_pos];
                if ((bq >> l) & (br >> l) & 1ull) { // 'M'
                    const char tc = ref[t_pos];
                    if (!rtk_is_dna(qc)) {
                        if (!rtk_is_dna(tc)) { const int x = rtk_amb_find(ms, nms, q_pos); if (x >= 0) ms[x] = rtk_amb_mk(q_pos, c_no); }
                        else if (quality[q_pos] >= q_min_corr) { if (rtk_iupac_overlap(qc, tc)) { const int x = rtk_amb_find(ms, nms, q_pos); if (x >= 0) ms[x] = rtk_amb_mk(q_pos, tc); } }
                        const int y = rtk_amb_find(ma, nma, q_pos); if (y >= 0) ma[y] = rtk_amb_mk(q_pos, tc);
                    } else if (quality[q_pos] < q_min_conf_corr || !rtk_iupac_overlap(qc, tc)) { // the read carries a code here
                        if (nms >= cap || nma >= cap) { rtk_fail_ovf(s, 11); return; }
                        if (rtk_amb_find(ms, nms, q_pos) < 0) ms[nms++] = rtk_amb_mk(q_pos, c_no);
                        if (rtk_amb_find(ma, nma, q_pos) < 0) ma[nma++] = rtk_amb_mk(q_pos, tc);
                    }
                } else { // 'I'
                    const int x = rtk_amb_find(ms, nms, q_pos), y = rtk_amb_find(ma, nma, q_pos);
                    if (x >= 0 && y >= 0) { ma[y] = rtk_amb_mk(q_pos, rtk_amb_chr(ms[x])); ms[x] = rtk_amb_mk(q_pos, c_no); }
                }
            }
            qp += static_cast<uint32_t>(rtk_popc(bq)); rp += static_cast<uint32_t>(rtk_popc(br));
        }
    }
    RTK_FA_LAP(2)
    // alleles of the other annotated positions of the unitig a decided SNP lies on (:713-768)
    for (uint32_t e = 0; e < nms; ++e) {
        const char pc = rtk_amb_chr(ms[e]);
        if (!rtk_is_dna(pc)) continue;
        const uint32_t p = rtk_amb_pos(ms[e]);
        const uint32_t pos_buff = (p < k - 1) ? 0 : (p - k + 1);
        const uint32_t len_buff = ((p + k < query_len) ? (p + k) : query_len) - pos_buff;
        const uint32_t pos_snp_buff = p - pos_buff;
        char* q_sub = s.str[1];
        rtk_wcopy(q_sub, query + pos_buff, len_buff);
        rtk_sync();
        q_sub[pos_snp_buff] = pc;
        rtk_sync();
        // every window of the 2k - 1 characters looked up at once, one lane per window (the walk below only follows a few of them, but
        // a lookup is a chain of dependent memory round trips and the chains of one batch overlap)
        const uint32_t nwin = len_buff >= k ? len_buff - k + 1 : 0u; // <= k <= 63
        auto window = [&](uint32_t w, bool* valid) -> uint64_t { // is window w all A/C/G/T, and where is its k-mer in the graph
            const char* wp = q_sub + w;
            bool ok = true; for (uint32_t x = 0; x < k; ++x) ok = ok && rtk_is_dna(wp[x]);
            *valid = ok;
            if (!ok) return RTK_NO_HIT;
            RtkKm km = rtk_km_zero(); for (uint32_t x = 0; x < k; ++x) km = rtk_km_push(km, static_cast<uint64_t>(rtk_cls(static_cast<unsigned char>(wp[x] & 0xDF))), static_cast<int>(k));
            return rtk_find_km(g, km, nullptr);
        };
#ifndef RTK_SIM
        uint64_t my_hit = RTK_NO_HIT; bool my_valid = false;
        if (static_cast<uint32_t>(rtk_lane()) < nwin) my_hit = window(static_cast<uint32_t>(rtk_lane()), &my_valid);
        const uint64_t vmask = rtk_ballot(my_valid);
#endif
        uint32_t skip_until = 0; bool skip_one = false;
        for (uint32_t w = 0; w < nwin; ++w) { // [A6] KmerIterator: the all-ACGT windows, in order
#ifdef RTK_SIM
            bool valid_w = false; const uint64_t hit_w = window(w, &valid_w); // the 1-lane simulator looks the windows up as it meets them
            if (!valid_w) continue;
#else
            if (!((vmask >> w) & 1ull)) continue;
#endif
            if (w < skip_until) continue;
            if (skip_one) { skip_one = false; continue; }
#ifdef RTK_SIM
            const uint64_t hit = hit_w;
#else
            const uint64_t hit = rtk_u(rtk_shfl(my_hit, static_cast<int>(w)));
#endif
            if (hit == RTK_NO_HIT) continue;
            const UMap um = rtk_extend_hit(c, hit, q_sub, w, len_buff);
            const uint32_t usz = rtk_ulen(g, um.unitig);
            UMap full = um; full.dist = 0; full.len = usz - k + 1;
            const uint32_t nvu = rtk_amb_of_um(c, full, vu, cap);
            if (rtk_failed(s)) return;
            uint32_t pos_snp_unitig = (pos_snp_buff - w) + um.dist;
            if (!um.strand) pos_snp_unitig = usz - pos_snp_unitig - 1;
            for (uint32_t a = 0; a < nvu; ++a) {
                const uint32_t ap = rtk_amb_pos(vu[a]);
                const int64_t pos = (ap <= pos_snp_unitig) ? (static_cast<int64_t>(p) - static_cast<int64_t>(pos_snp_unitig - ap)) : (static_cast<int64_t>(p) + static_cast<int64_t>(ap - pos_snp_unitig));
                if (pos < 0 || pos >= static_cast<int64_t>(query_len) || pos == static_cast<int64_t>(p)) continue;
                const int x = rtk_amb_find(ms, nms, static_cast<uint32_t>(pos));
                if (x < 0 || rtk_is_dna(rtk_amb_chr(ms[x]))) continue;
                const char uc = um.strand ? rtk_unitig_char(g, um.unitig, ap) : rtk_iupac_comp(rtk_unitig_char(g, um.unitig, usz - 1 - ap)); // unitig_seq[p_amb.first]
                const uint64_t ent = rtk_amb_mk(static_cast<uint64_t>(pos), uc);
                bool dup = false;
                for (uint32_t z = 0; z < nsa && !dup; ++z) dup = sa[z] == ent;
                if (!dup) { if (nsa >= cap) { rtk_fail_ovf(s, 11); return; } sa[nsa++] = ent; }
            }
            skip_until = w + (um.len - 1); skip_one = um.len >= 2; // it_km += um.len - 1, then ++it_km
        }
        rtk_sync();
    }
    RTK_FA_LAP(3)
    for (uint32_t i = 0; i < nsa; ++i) { // a linked position with exactly one candidate allele takes it, when compatible (:771-790)
        const uint32_t pos = rtk_amb_pos(sa[i]);
        uint32_t same = 0;
        for (uint32_t j = 0; j < nsa; ++j) same += (rtk_amb_pos(sa[j]) == pos) ? 1u : 0u;
        if (same != 1) continue;
        const int x = rtk_amb_find(ms, nms, pos);
        if (x >= 0 && rtk_iupac_overlap(rtk_amb_chr(sa[i]), rtk_amb_chr(ms[x]))) ms[x] = rtk_amb_mk(pos, rtk_amb_chr(sa[i]));
    }
    for (uint32_t e = 0; e < nms; ++e) { // :792-838
        const uint32_t p = rtk_amb_pos(ms[e]); const char pc = rtk_amb_chr(ms[e]);
        if (pc == c_no || quality[p] < q_min_corr) {
            const int y = rtk_amb_find(ma, nma, p);
            if (y >= 0) { qt[p] = rtk_amb_chr(ma[y]); quality[p] = q_max_corr; }
        }
        else if (!rtk_is_dna(pc)) qt[p] = query[p];
        else qt[p] = pc;
    }
    rtk_sync();
    rtk_wcopy(query, qt, query_len);
    rtk_sync();
    RTK_FA_LAP(4)
}

#endif
