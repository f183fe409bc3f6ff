// ORACLE (test infrastructure). Restatement of getSeeds (reference: src/Graph.cpp:3-482) and
// keep_non_overlap (reference: src/Alignment.cpp:1017-1199) for pass 1. See oracle_correct.hpp.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

#include "oracle_correct.hpp"

namespace orc {

namespace {

inline bool isACGT(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// Bifrost cstrMatch: length of the common prefix of two NUL-terminated strings
inline size_t cstrMatch(const char* a, const char* b) { const char* a0 = a; while (*a && *b && *a == *b) { ++a; ++b; } return static_cast<size_t>(a - a0); }

struct CompAnchor { // comp_pair, src/Graph.cpp:9-14
    const Graph& g;
    explicit CompAnchor(const Graph& g_) : g(g_) {}
    bool operator()(const Anchor& a, const Anchor& b) const {
        if (a.first == b.first) return g.mapped(a.second) < g.mapped(b.second);
        return a.first < b.first;
    }
};

void removeEmpty(std::vector<Anchor>& v) { // removeEmptyUnitigMap, src/Graph.cpp:16-26
    std::vector<Anchor> o;
    for (size_t i = 0; i < v.size(); ++i) if (!v[i].second.isEmpty()) o.push_back(v[i]);
    v.swap(o);
}

} // namespace

std::vector<Anchor> searchExact(const Graph& g, const std::string& s, Counters* cnt) { // src/Graph.cpp:97 [A1]
    std::vector<Anchor> v;
    const size_t k = static_cast<size_t>(g.k);
    if (s.size() < k) return v;
    size_t bad = 0; // number of non-ACGT among the current k chars
    for (size_t i = 0; i < k - 1; ++i) bad += !isACGT(s[i]);
    for (size_t p = 0; p + k <= s.size(); ++p) {
        bad += !isACGT(s[p + k - 1]);
        if (bad == 0) {
            if (cnt) ++cnt->n_probe;
            const UM um = g.findKmer(s.c_str() + p);
            if (!um.isEmpty()) { v.push_back(Anchor(p, um)); if (cnt) ++cnt->n_verify; }
        }
        bad -= !isACGT(s[p]);
    }
    return v;
}

std::string maskForInexact(const Graph& g, const Opt& opt, const std::string& s, const std::vector<Anchor>& v_um) { // src/Graph.cpp:102-191
    const size_t k = static_cast<size_t>(g.k);
    std::string l_s(s.length(), 'N');
    for (size_t i = 1; i < v_um.size(); ++i) {
        if (v_um[i].first != v_um[i - 1].first + 1) {
            const size_t diff = v_um[i].first - v_um[i - 1].first;
            if (diff >= opt.insert_sz) l_s.replace(v_um[i - 1].first + k, diff - k, s, v_um[i - 1].first + k, diff - k);
            else if (diff >= opt.insert_sz / 2) {
                const size_t search_space_len = opt.insert_sz - diff;
                const size_t min_pos_left = (v_um[i - 1].first < search_space_len) ? 0 : (v_um[i - 1].first - search_space_len);
                const size_t max_pos_right = v_um[i].first + search_space_len;
                const Anchor* prev_left = nullptr; const Anchor* prev_right = nullptr;
                IdSet pid_left, pid_right;
                size_t i_l = i - 1, i_r = i;
                while (i_l > 0 && v_um[i_l].first > min_pos_left) { // G13: index 0 is never visited
                    const UM& um_left = v_um[i_l].second;
                    if (prev_left == nullptr || !g.sameUnitig(um_left, prev_left->second)) {
                        if (!g.isBranching(um_left.unitig)) pid_left = set_union(pid_left, g.allIds(um_left.unitig));
                        prev_left = &v_um[i_l];
                    }
                    --i_l;
                }
                while (i_r < v_um.size() && v_um[i_r].first < max_pos_right) {
                    const UM& um_right = v_um[i_r].second;
                    if (prev_right == nullptr || !g.sameUnitig(um_right, prev_right->second)) {
                        if (!g.isBranching(um_right.unitig)) pid_right = set_union(pid_right, g.allIds(um_right.unitig));
                        prev_right = &v_um[i_r];
                    }
                    ++i_r;
                }
                if (set_inter_card(pid_left, pid_right) < opt.min_cov_vertices) l_s.replace(v_um[i - 1].first + k, diff - k, s, v_um[i - 1].first + k, diff - k);
            }
        }
    }
    if (!v_um.empty()) {
        if (v_um.front().first >= opt.insert_sz / 2) l_s.replace(0, v_um.front().first + k - 1, s, 0, v_um.front().first + k - 1);
        if (s.length() - v_um.back().first >= opt.insert_sz / 2) l_s.replace(v_um.back().first + 1, s.length() - v_um.back().first - 1, s, v_um.back().first + 1, s.length() - v_um.back().first - 1);
    }
    return l_s;
}

std::vector<Anchor> searchInexact(const Graph& g, const std::string& m, Counters* cnt) { // src/Graph.cpp:193 [A2]
    std::vector<Anchor> v;
    const size_t k = static_cast<size_t>(g.k), n = m.size();
    if (n < k) return v;
    // run[i] = number of consecutive ACGT characters starting at i
    std::vector<uint32_t> run(n + 1, 0);
    for (size_t i = n; i-- > 0;) run[i] = isACGT(m[i]) ? run[i + 1] + 1 : 0;
    std::string km(k, 'A');
    std::set<std::pair<size_t, uint64_t> > seen; // (pos, unitig<<33|dist<<1|strand): one report per (position, mapped k-mer)
    auto probe = [&](size_t p) {
        if (cnt) ++cnt->n_probe;
        const UM um = g.findKmer(km.c_str());
        if (um.isEmpty()) return;
        if (cnt) ++cnt->n_verify;
        const uint64_t key = (static_cast<uint64_t>(um.unitig) << 33) | (static_cast<uint64_t>(um.dist) << 1) | (um.strand ? 1ULL : 0ULL);
        if (seen.insert(std::make_pair(p, key)).second) v.push_back(Anchor(p, um));
    };
    // [A2] as a switch (the readings of Bifrost's or_exclusive_match are kept under test, oracle_graph.hpp). RTK_A2_XOR=union: the hits of all three
    // kinds of edit; exclusive (default): a window is not searched with the next kind once one kind has matched it, kinds in the order substitution ->
    // insertion -> deletion; exclusive-ids: the same with the kinds in the order insertion -> deletion -> substitution
    const char* a2 = getenv("RTK_A2_XOR");
    const int mode = (a2 && !strcmp(a2, "union")) ? 0 : ((a2 && !strcmp(a2, "exclusive-ids")) ? 2 : 1);
    for (size_t p = 0; p + k <= n; ++p) {
        if (run[p] + 1 < k) continue; // fewer than k-1 usable characters: no variant exists
        const size_t n_before = v.size();
        auto subs = [&]() {
            if (run[p] >= k) for (size_t o = 0; o < k; ++o) {
                km.assign(m, p, k);
                for (int b = 0; b < 4; ++b) { if ("ACGT"[b] == m[p + o]) continue; km[o] = "ACGT"[b]; probe(p); }
            }
        };
        auto inss = [&]() { // graph k-mer has one extra base w.r.t. the read ("insertion" in Bifrost's terms)
            for (size_t o = 0; o < k; ++o) for (int b = 0; b < 4; ++b) { km.assign(m, p, o); km.push_back("ACGT"[b]); km.append(m, p + o, k - 1 - o); probe(p); }
        };
        auto dels = [&]() { // graph k-mer lacks one interior read base ("deletion")
            if (run[p] >= k + 1) for (size_t o = 1; o + 1 <= k - 1; ++o) { km.assign(m, p, o); km.append(m, p + o + 1, k - o); probe(p); }
        };
        const bool excl = mode != 0;
        if (mode == 2) { inss(); if (!(excl && v.size() != n_before)) dels(); if (!(excl && v.size() != n_before)) subs(); }
        else { subs(); if (!(excl && v.size() != n_before)) inss(); if (!(excl && v.size() != n_before)) dels(); }
    }
    return v;
}

std::vector<Anchor> keepNonOverlap(const Graph& g, const char* ref, size_t ref_len, const std::vector<Anchor>& v) { // src/Alignment.cpp:1017-1199
    const size_t k = static_cast<size_t>(g.k);
    struct var_info_t { size_t pos_s, pos_e; bool keep; std::vector<uint32_t> pos_v; std::set<std::pair<int32_t, bool> > s_km; var_info_t() : pos_s(0), pos_e(0), keep(true) {} };
    std::map<size_t, var_info_t> m_var;
    for (size_t i = 0; i < v.size(); ++i) {
        const Anchor& p_um = v[i];
        const std::string km_ref(ref + p_um.first, std::min(k, ref_len - p_um.first));
        const std::string km_query = g.mapped(p_um.second);
        const size_t l = cstrMatch(km_ref.c_str(), km_query.c_str());
        int type_var = 0; int mis_ins = 0; // getAmbiguityIndex('.') == 0
        auto ambIdx = [](char c) -> int { switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 0; } };
        // guards on l keep the pointer arithmetic inside the strings exactly as c_str() semantics would (reads stop at NUL)
        const char* r0 = km_ref.c_str(); const char* q0 = km_query.c_str();
        if (l < km_ref.size() && l + cstrMatch(r0 + l + 1, q0 + l + 1) == k - 1) { type_var = 1; mis_ins = ambIdx(km_query[l]); }
        else if (l < km_query.size() && l + cstrMatch(r0 + l, q0 + l + 1) == k - 1) { type_var = 2; mis_ins = ambIdx(km_query[l]); }
        else if (l < km_ref.size() && l + cstrMatch(r0 + l + 1, q0 + l) == k - 1) type_var = 3;
        if (type_var != 0 && l != 0 && l != k - 1) {
            const size_t pos = p_um.first + l;
            const size_t key = (pos << 16) | ((static_cast<size_t>(mis_ins) & 0xffULL) << 8) | (static_cast<size_t>(type_var) & 0xffULL);
            std::pair<std::map<size_t, var_info_t>::iterator, bool> p_it = m_var.insert(std::make_pair(key, var_info_t()));
            var_info_t& vi = p_it.first->second;
            if (p_it.second) { vi.pos_s = p_um.first; vi.pos_e = p_um.first + k; }
            else { vi.pos_s = std::min(vi.pos_s, p_um.first); vi.pos_e = std::max(vi.pos_e, p_um.first + k); }
            vi.s_km.insert(std::make_pair(p_um.second.unitig, p_um.second.strand)); // stands for the strand-normalised unitig head k-mer (:1096)
            vi.pos_v.push_back(static_cast<uint32_t>(i));
        }
    }
    std::set<uint32_t> pos_out;
    for (std::map<size_t, var_info_t>::iterator it1 = m_var.begin(); it1 != m_var.end(); ++it1) {
        if (!it1->second.keep) continue;
        const size_t it1_pos = it1->first >> 16;
        const size_t lower = (it1_pos < k - 1) ? 0 : (it1_pos - k + 1);
        const size_t upper = ((it1_pos + k) >= ref_len) ? ref_len : (it1_pos + k);
        const size_t upper_key = (upper << 16) + 0xffffULL;
        std::map<size_t, var_info_t>::iterator it2 = m_var.lower_bound(lower << 16);
        while (it1->second.keep && it2 != m_var.end() && it2->first <= upper_key) {
            const size_t it2_pos = it2->first >> 16;
            const bool overlap1 = (it1_pos >= it2->second.pos_s) && (it1_pos < it2->second.pos_e);
            const bool overlap2 = (it2_pos >= it1->second.pos_s) && (it2_pos < it1->second.pos_e);
            if (it1->first != it2->first && (overlap1 || overlap2)) {
                bool sameUnitig = false;
                for (std::set<std::pair<int32_t, bool> >::const_iterator a = it1->second.s_km.begin(); a != it1->second.s_km.end() && !sameUnitig; ++a) sameUnitig = it2->second.s_km.count(*a) != 0;
                if (!sameUnitig) { it1->second.keep = false; it2->second.keep = false; }
            }
            ++it2;
        }
        if (it1->second.keep) pos_out.insert(it1->second.pos_v.begin(), it1->second.pos_v.end());
    }
    std::vector<Anchor> out;
    for (std::set<uint32_t>::const_iterator it = pos_out.begin(); it != pos_out.end(); ++it) out.push_back(v[*it]);
    return out;
}

std::pair<std::vector<Anchor>, std::vector<Anchor> > getSeeds(const Graph& g, const Opt& opt, const std::string& s, Counters* cnt) {
    const size_t k = static_cast<size_t>(g.k);
    std::vector<Anchor> v_um, solid, weak;
    if (s.length() <= k) return std::make_pair(solid, weak); // src/Graph.cpp:49
    v_um = searchExact(g, s, cnt);
    std::sort(v_um.begin(), v_um.end(), CompAnchor(g));                          // :104
    if (!opt.long_read_correct) {                                                // :100 pass 2 keeps the exact hits only
        const std::string l_s = maskForInexact(g, opt, s, v_um);                 // :102-191
        const std::vector<Anchor> inexact = searchInexact(g, l_s, cnt);          // :193
        v_um.insert(v_um.end(), inexact.begin(), inexact.end());
    }
    std::sort(v_um.begin(), v_um.end(), CompAnchor(g));                          // :201
    for (size_t i = 0; i < v_um.size(); ++i) {                                   // :209-216
        if (i == 0 || v_um[i] != v_um[i - 1]) {
            if (cstrMatch(g.mapped(v_um[i].second).c_str(), s.c_str() + v_um[i].first) == k) solid.push_back(v_um[i]);
            else weak.push_back(v_um[i]);
        }
    }
    if (solid.size() >= 2) {                                                     // :221-239
        for (size_t i = 1; i < solid.size(); ++i) {
            if (solid[i].first != solid[i - 1].first + 1 && solid[i].first < solid[i - 1].first + k) {
                solid[i - 1].second.unitig = -1;
                int64_t j = static_cast<int64_t>(i) - 2;
                while (j >= 0 && solid[j].first == solid[j + 1].first - 1 && solid[i].first < solid[j].first + k) solid[j--].second.unitig = -1;
            }
        }
        removeEmpty(solid);
    }
    if (!weak.empty()) weak = keepNonOverlap(g, s.c_str(), s.size(), weak);      // :241
    // NB: emptied anchors must keep their unitig identity for isSameReferenceUnitig below, so emptiness is tracked separately
    std::vector<char> empty(solid.size(), 0);
    for (size_t i = 1; i < solid.size(); ++i) {                                  // :329-372
        if (solid[i].first - solid[i - 1].first == 1) {
            const UM& um_left = solid[i - 1].second; const UM& um_right = solid[i].second;
            if (!empty[i - 1] && !empty[i] && !g.sameUnitig(um_left, um_right)) {
                const std::string& sl = g.seq[um_left.unitig]; const std::string& sr = g.seq[um_right.unitig];
                const std::string s_tail_left = um_left.strand ? sl.substr(sl.size() - k) : revcomp(sl.substr(0, k));
                const std::string s_head_right = um_right.strand ? sr.substr(0, k) : revcomp(sr.substr(sr.size() - k));
                bool invalid = (s_tail_left.substr(1, k - 1) != s_head_right.substr(0, k - 1));
                if (!invalid) invalid = (g.sharedCount(um_left.unitig, um_right.unitig) < opt.min_cov_vertices);
                if (invalid) {
                    size_t i_l = i - 1, i_r = i + 1;
                    i_l -= (i_l != 0);
                    while (i_l > 0 && solid[i_l].first == solid[i_l + 1].first - 1 && g.sameUnitig(solid[i_l].second, um_left)) { empty[i_l] = 1; --i_l; }
                    while (i_r < solid.size() && solid[i_r].first == solid[i_r - 1].first + 1 && g.sameUnitig(solid[i_r].second, um_right)) { empty[i_r] = 1; ++i_r; }
                    empty[i - 1] = 1; empty[i] = 1;
                }
            }
        }
    }
    {
        std::vector<Anchor> o;
        for (size_t i = 0; i < solid.size(); ++i) if (!empty[i]) o.push_back(solid[i]);
        solid.swap(o);
    }
    return std::make_pair(solid, weak);
}

} // namespace orc
