// ORACLE (test infrastructure). Restatement of the pass-1 correction path of the reference.
// See oracle_correct.hpp for scope, canonical rules [D1]/[D2] and what is deliberately not restated.
#include "oracle_correct.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>

#include "oracle_myers.hpp"

namespace orc {

ref_edlib_moves_fn g_ref_edlib_moves = nullptr; // set by orc_use_reference_edlib (oracle_capi.cpp)

namespace {

// ---------------------------------------------------------------- helpers (src/Common.hpp:410-438)
inline char getQual(const double score, const size_t qv_min, const size_t qv_max) {
    const char phred_base_std = static_cast<char>(33);
    const char phred_scale_std = static_cast<char>(qv_max);
    const double qv_score = std::min(score, 1.0) * static_cast<double>(phred_scale_std - qv_min);
    return static_cast<char>(qv_score + phred_base_std + qv_min);
}

inline std::pair<size_t, size_t> getMinMaxLength(const size_t l, const double len_factor) {
    return std::make_pair(static_cast<size_t>(std::max(l - (l * len_factor), 1.0)), static_cast<size_t>(std::max(l + (l * len_factor), 1.0)));
}

struct Ctx {
    const Graph& g; const Opt& opt; Counters* cnt; size_t k;
    Ctx(const Graph& g_, const Opt& o_, Counters* c_) : g(g_), opt(o_), cnt(c_), k(static_cast<size_t>(g_.k)) {}
    AlignResult align(const char* q, size_t ql, const char* t, size_t tl, int kk, AlignMode mode, bool path, bool iupac = true) const {
        if (cnt) { ++cnt->n_align; cnt->n_align_cells += static_cast<uint64_t>((ql + 63) / 64) * tl; }
        if (g_ref_edlib_moves) { // the REFERENCE's own edlib (oracle/_ref) instead of the restatement: same results (pinned), banded
            AlignResult r; int nloc = 0, nmv = 0;
            std::vector<int> locs(tl + 2); std::vector<unsigned char> mv(path ? ql + tl + 2 : 1);
            r.editDistance = g_ref_edlib_moves(q, static_cast<int>(ql), t, static_cast<int>(tl), kk, static_cast<int>(mode), path ? 1 : 0, iupac ? 1 : 0, &nloc, locs.data(), static_cast<int>(locs.size()), mv.data(), static_cast<int>(mv.size()), &nmv);
            if (r.editDistance >= 0) { r.endLocations.assign(locs.begin(), locs.begin() + nloc); if (path) r.alignment.assign(mv.begin(), mv.begin() + nmv); }
            return r;
        }
        return myers_align(q, static_cast<int>(ql), t, static_cast<int>(tl), kk, mode, path, iupac);
    }
};

// ---------------------------------------------------------------- Path (src/Path.hpp)
// The reference keeps (start, end, one successor base per intermediate unitig) and re-derives the
// intermediate unitigs as WHOLE unitigs whenever the path is walked (Path.hpp:463-475). Here the unitigs are
// kept explicitly; an element that stops being the last one is normalised to the whole unitig, which is
// exactly what the successor-base encoding does to it.
struct Path {
    std::vector<UM> ums;
    std::string qual;
    size_t l;
    Path() : l(0) {}
    bool isEmpty() const { return ums.empty(); }
    size_t size() const { return ums.size(); }     // Path.hpp:416-419
    size_t length() const { return l; }
    const UM& back() const { return ums.back(); }  // Path.hpp:269-272
    const UM& front() const { return ums.front(); }
};

inline void normaliseBack(const Ctx& c, Path& p) { // what "succ.append(1, ...)" does to the former end (Path.hpp:319-323)
    if (p.ums.size() >= 2) { UM& e = p.ums.back(); e.dist = 0; e.len = c.g.nkm(e.unitig); }
}

bool pathExtend(const Ctx& c, Path& p, const UM& um) { // Path.hpp:308-330
    if (um.isEmpty()) return false;
    if (p.ums.empty()) { p.ums.push_back(um); p.l = um.len + c.k - 1; }
    else { normaliseBack(c, p); p.ums.push_back(um); p.l += um.len; }
    return true;
}

bool pathExtend(const Ctx& c, Path& p, const UM& um, const std::string& qual_s) { // Path.hpp:332-363
    if (um.isEmpty()) return false;
    const size_t um_substr_len = um.len + c.k - 1;
    if (p.ums.empty()) {
        p.ums.push_back(um); p.l = um_substr_len;
        if (qual_s.length() == um_substr_len) p.qual = qual_s; else return false;
    } else {
        normaliseBack(c, p); p.ums.push_back(um); p.l += um.len;
        if (qual_s.length() == um_substr_len) p.qual += qual_s.substr(c.k - 1); else return false;
    }
    return true;
}

bool pathMerge(const Ctx& c, Path& p, const Path& o) { // Path.hpp:366-414
    if (o.length() == 0) return true;
    if (p.length() == 0) { p = o; return true; }
    if (p.qual.empty() != o.qual.empty()) return false;
    const UM& last = p.ums.back();
    if (last.unitig != o.ums.front().unitig || last.strand != o.ums.front().strand) return false;
    if (p.ums.size() == 1) {
        UM& st = p.ums[0];
        if (!st.strand) st.dist = o.ums[0].dist;
        st.len += o.ums[0].len - 1;
        for (size_t i = 1; i < o.ums.size(); ++i) p.ums.push_back(o.ums[i]);
    } else {
        UM& en = p.ums.back();
        if (!en.strand) en.dist = o.ums[0].dist;
        en.len += o.ums[0].len - 1;
        if (o.ums.size() >= 2) { normaliseBack(c, p); for (size_t i = 1; i < o.ums.size(); ++i) p.ums.push_back(o.ums[i]); }
    }
    p.l += o.l - c.k;
    if (o.qual.length() != 0) p.qual.append(o.qual.substr(c.k));
    return true;
}

void pathPrunePrefix(const Ctx& c, Path& p, const size_t len) { // Path.hpp:487-571
    if (p.ums.empty() || p.l == 0 || len >= p.l) return;
    const size_t k = c.k;
    UM& st = p.ums[0];
    if (p.ums.size() == 1) {
        if (!st.strand) st.dist += static_cast<uint32_t>(p.l - len);
        st.len -= static_cast<uint32_t>(p.l - len);
    } else if (st.len + k - 1 >= len) {
        p.l = st.len + k - 1;
        p.ums.resize(1);
        UM& s2 = p.ums[0];
        if (!s2.strand) s2.dist += static_cast<uint32_t>(p.l - len);
        s2.len -= static_cast<uint32_t>(p.l - len);
    } else if (p.ums.size() == 2 || len > (p.l - p.ums.back().len)) {
        UM& en = p.ums.back();
        if (!en.strand) en.dist += static_cast<uint32_t>(p.l - len);
        en.len -= static_cast<uint32_t>(p.l - len);
    } else {
        size_t acc = st.len + k - 1;
        std::vector<UM> kept(1, st);
        bool cut = false;
        for (size_t i = 1; i + 1 < p.ums.size(); ++i) { // intermediate (whole) unitigs
            UM cur = p.ums[i]; cur.dist = 0; cur.len = c.g.nkm(cur.unitig);
            acc += cur.len;
            if (acc < len) kept.push_back(cur);
            else {
                if (!cur.strand) cur.dist += static_cast<uint32_t>(acc - len);
                cur.len -= static_cast<uint32_t>(acc - len);
                kept.push_back(cur);
                cut = true;
                break;
            }
        }
        if (!cut) kept.push_back(p.ums.back()); // the reference leaves `end` untouched when no intermediate reaches len
        p.ums.swap(kept);
    }
    p.l = len;
    if (p.qual.length() != 0) p.qual = p.qual.substr(0, p.l);
}

std::string pathToString(const Ctx& c, const Path& p) { // Path.hpp:449-485
    std::string s;
    if (p.ums.empty()) return s;
    s = c.g.mapped(p.ums[0]);
    for (size_t i = 1; i < p.ums.size(); ++i) s.append(c.g.mapped(p.ums[i]).substr(c.k - 1));
    if (c.cnt) c.cnt->n_path_base += s.size();
    return s;
}

typedef std::vector<std::pair<size_t, char> > AmbVec;

inline char iupacUnion(char a, char b) { return iupacChar(static_cast<uint8_t>(iupacIndex(a) | iupacIndex(b))); } // getAmbiguityRev + getAmbiguity (src/Common.hpp:351-400)
inline bool iupacOverlap(char a, char b) { return (iupacIndex(a) & iupacIndex(b)) != 0; }

// src/GraphTraversal.cpp:966-1036: SNP annotations of the path's unitigs in path-string coordinates; annotations in the k-1
// characters two consecutive unitigs share are merged (union of the alleles)
AmbVec getAmbiguityVector(const Ctx& c, const Path& p) {
    AmbVec v_amb;
    const size_t k = c.k;
    size_t prev_l = 0, pos_prev_l = 0;
    for (size_t x = 0; x < p.ums.size(); ++x) {
        const UM& um = p.ums[x];
        const AmbVec v_amb_um = c.g.ambiguityChars(um);
        AmbVec v_amb_tmp;
        size_t it_prev = pos_prev_l, it_curr = 0;
        while (it_prev != v_amb.size() && it_curr != v_amb_um.size() && v_amb_um[it_curr].first < k - 1) {
            const size_t it_curr_pos = v_amb_um[it_curr].first + prev_l;
            if (v_amb[it_prev].first < it_curr_pos) { v_amb_tmp.push_back(v_amb[it_prev]); ++it_prev; }
            else if (v_amb[it_prev].first > it_curr_pos) { v_amb_tmp.push_back(std::make_pair(it_curr_pos, v_amb_um[it_curr].second)); ++it_curr; }
            else { v_amb_tmp.push_back(std::make_pair(v_amb[it_prev].first, iupacUnion(v_amb[it_prev].second, v_amb_um[it_curr].second))); ++it_prev; ++it_curr; }
        }
        for (; it_prev != v_amb.size(); ++it_prev) v_amb_tmp.push_back(v_amb[it_prev]);
        for (; it_curr != v_amb_um.size(); ++it_curr) v_amb_tmp.push_back(std::make_pair(v_amb_um[it_curr].first + prev_l, v_amb_um[it_curr].second));
        prev_l += um.len;
        v_amb.erase(v_amb.begin() + static_cast<long>(pos_prev_l), v_amb.end());
        for (size_t i = 0; i < v_amb_tmp.size(); ++i) { v_amb.push_back(v_amb_tmp[i]); pos_prev_l += static_cast<size_t>(v_amb.back().first < prev_l); }
    }
    return v_amb;
}

// [A6] positions of the all-ACGT windows of s
std::vector<size_t> kmerWindows(const std::string& s, size_t k) {
    std::vector<size_t> v; size_t run = 0;
    for (size_t i = 0; i < s.length(); ++i) { run = isDNA(s[i]) ? run + 1 : 0; if (run >= k) v.push_back(i + 1 - k); }
    return v;
}

// src/Alignment.cpp:527-844 with hap_id undetermined (no phasing input: every isValidHap test is short-circuited, :741,:801)
void fixAmbiguity(const Ctx& c, std::string& query, std::string& quality, const char* ref_seq, const size_t ref_len, const AmbVec& v_ambiguity) {
    if (v_ambiguity.empty()) return;
    const size_t query_len = query.length();
    const size_t k = c.k;
    if (quality.length() < query_len) { fprintf(stderr, "oracle: fixAmbiguity with a quality string shorter than the sequence\n"); abort(); }
    const char q_max_corr = getQual(1.0, c.opt.out_qual, c.opt.max_qual);
    const char q_min_corr = getQual(0.0, c.opt.out_qual, c.opt.max_qual);
    const char q_min_conf_corr = getQual(c.opt.min_confidence_snp_corr, 0, c.opt.max_qual);
    const char c_noCorrect = 'X';
    std::string query_tmp = query;
    std::map<size_t, char> m_safe, m_all; // the reference's hash maps are only ever used key by key
    for (size_t i = 0; i < v_ambiguity.size(); ++i) {
        const std::pair<size_t, char>& p = v_ambiguity[i];
        if (quality[p.first] < q_min_conf_corr) { m_safe.insert(p); query_tmp[p.first] = p.second; }
    }
    m_all = m_safe;
    const AlignResult align = c.align(query_tmp.c_str(), query_len, ref_seq, ref_len, -1, MODE_SHW, true);
    { // walk of the CIGAR (:612-706), op by op
        size_t q_pos = 0, t_pos = 0; // SHW: startLocations[0] == 0
        for (size_t a = 0; a < align.alignment.size(); ++a) {
            const unsigned char op = align.alignment[a];
            if (op == 0 || op == 3) { // 'M'
                if (!isDNA(query_tmp[q_pos])) {
                    if (!isDNA(ref_seq[t_pos])) { std::map<size_t, char>::iterator it = m_safe.find(q_pos); if (it != m_safe.end()) it->second = c_noCorrect; }
                    else if (quality[q_pos] >= q_min_corr) {
                        if (iupacOverlap(query_tmp[q_pos], ref_seq[t_pos])) { std::map<size_t, char>::iterator it = m_safe.find(q_pos); if (it != m_safe.end()) it->second = ref_seq[t_pos]; }
                    }
                    std::map<size_t, char>::iterator it = m_all.find(q_pos);
                    if (it != m_all.end()) it->second = ref_seq[t_pos];
                } else if (!isDNA(ref_seq[t_pos])) {
                    if (quality[q_pos] < q_min_conf_corr || !iupacOverlap(query_tmp[q_pos], ref_seq[t_pos])) {
                        m_safe.insert(std::make_pair(q_pos, c_noCorrect));
                        m_all.insert(std::make_pair(q_pos, ref_seq[t_pos]));
                    }
                }
                ++q_pos; ++t_pos;
            } else if (op == 1) { // 'I'
                if (!isDNA(query_tmp[q_pos])) {
                    std::map<size_t, char>::iterator it_s = m_safe.find(q_pos), it_a = m_all.find(q_pos);
                    if (it_s != m_safe.end() && it_a != m_all.end()) { it_a->second = it_s->second; it_s->second = c_noCorrect; }
                }
                ++q_pos;
            } else ++t_pos; // 'D'
        }
    }
    std::set<std::pair<size_t, char> > s_ambiguity;
    for (std::map<size_t, char>::const_iterator p = m_safe.begin(); p != m_safe.end(); ++p) { // :713-768 linked SNPs of the same unitig
        if (!isDNA(p->second)) continue;
        const size_t pos_buff = (p->first < (k - 1)) ? 0 : (p->first - k + 1);
        const size_t len_buff = std::min(p->first + k, query_len) - pos_buff;
        const size_t pos_snp_buff = p->first - pos_buff;
        std::string q_sub = query.substr(pos_buff, len_buff);
        q_sub[pos_snp_buff] = p->second;
        const std::vector<size_t> w = kmerWindows(q_sub, k);
        for (size_t wi = 0; wi < w.size(); ++wi) {
            const UM um = c.g.findUnitig(q_sub.c_str(), w[wi], q_sub.length());
            if (um.isEmpty()) continue;
            const UM um_tmp(um.unitig, 0, c.g.nkm(um.unitig), um.strand);
            const std::string unitig_seq = c.g.mapped(um_tmp);
            const AmbVec v_amb = c.g.ambiguityChars(um_tmp);
            size_t pos_snp_unitig = (pos_snp_buff - w[wi]) + um.dist;
            if (!um.strand) pos_snp_unitig = c.g.usize(um.unitig) - pos_snp_unitig - 1;
            for (size_t a = 0; a < v_amb.size(); ++a) {
                int64_t pos = static_cast<int64_t>(v_amb[a].first);
                if (static_cast<size_t>(pos) <= pos_snp_unitig) pos = static_cast<int64_t>(p->first) - static_cast<int64_t>(pos_snp_unitig - static_cast<size_t>(pos));
                else pos = static_cast<int64_t>(p->first) + static_cast<int64_t>(static_cast<size_t>(pos) - pos_snp_unitig);
                if (pos >= 0 && static_cast<size_t>(pos) < query_len && static_cast<size_t>(pos) != p->first) {
                    const std::map<size_t, char>::const_iterator it = m_safe.find(static_cast<size_t>(pos));
                    if (it != m_safe.end() && !isDNA(it->second)) s_ambiguity.insert(std::make_pair(static_cast<size_t>(pos), unitig_seq[v_amb[a].first]));
                }
            }
            const size_t next_pos = w[wi] + (um.len - 1); // it_km += um.len - 1 [A6]
            while (wi < w.size() && w[wi] < next_pos) ++wi;
            if (wi >= w.size()) break;
        }
    }
    {
        const AmbVec v(s_ambiguity.begin(), s_ambiguity.end());
        for (size_t i = 0; i < v.size(); ++i) {
            if ((i == 0 || v[i].first != v[i - 1].first) && (i + 1 == v.size() || v[i].first != v[i + 1].first)) {
                std::map<size_t, char>::iterator it_s = m_safe.find(v[i].first);
                if (it_s != m_safe.end() && iupacOverlap(v[i].second, it_s->second)) it_s->second = v[i].second;
            }
        }
    }
    for (std::map<size_t, char>::const_iterator p = m_safe.begin(); p != m_safe.end(); ++p) { // :792-838
        if (p->second == c_noCorrect || quality[p->first] < q_min_corr) {
            const std::map<size_t, char>::const_iterator it_a = m_all.find(p->first);
            if (it_a != m_all.end()) { query_tmp[p->first] = it_a->second; quality[p->first] = q_max_corr; } // validHap is always true with an undetermined haplotype
        }
        else if (!isDNA(p->second)) query_tmp[p->first] = query[p->first];
        else query_tmp[p->first] = p->second;
    }
    query = query_tmp;
}

// ---------------------------------------------------------------- fixRepeats (src/GraphTraversal.cpp:1149-1334)
// Path(um_start, ext, um_end) (Path.hpp:109-152): um_start, then one whole unitig per character of ext (the successor reached by
// that base), then um_end. Empty when a base has no successor.
Path pathFromCompact(const Ctx& c, const UM& um_start, const std::string& ext, const UM& um_end) {
    Path p;
    UM curr = um_start;
    std::vector<UM> mid;
    for (size_t i = 0; i < ext.size(); ++i) {
        UM out[4]; char base[4]; int n = 0; c.g.successors(curr, out, base, n);
        int f = -1; for (int j = 0; j < n; ++j) if (base[j] == ext[i]) f = j;
        if (f < 0) return Path();
        curr = out[f]; mid.push_back(curr);
    }
    p.ums.push_back(um_start); p.l = um_start.len + c.k - 1;
    for (size_t i = 0; i < mid.size(); ++i) { p.ums.push_back(mid[i]); p.l += mid[i].len; }
    p.ums.push_back(um_end); p.l += um_end.len;
    return p;
}

Path pathRevComp(const Path& p) { // Path.hpp:208-262
    Path o; o.l = p.l; o.qual = std::string(p.qual.rbegin(), p.qual.rend());
    for (size_t i = p.ums.size(); i-- > 0;) { UM u = p.ums[i]; u.strand = !u.strand; o.ums.push_back(u); }
    return o;
}

Path fixRepeats(const Ctx& c, const Path& path_in, const char* ref, const size_t ref_len) {
    Path path = path_in;
    std::vector<UM> v = path.ums;
    std::string s_qual = path.qual;
    const char q_max = getQual(1.0, 0, c.opt.max_qual);
    int64_t editDistance;
    { const std::string s = pathToString(c, path); editDistance = c.align(s.c_str(), s.length(), ref, ref_len, -1, MODE_NW, false).editDistance; }
    for (size_t i = 0; i < v.size(); ++i) {
        const UM um_path = v[i];
        if (!c.g.isShortCycle(um_path.unitig)) continue;
        Path best;
        // the unitig from the mapped start to its end, and from its beginning to the mapped end, both forward (:1213-1224)
        UM um_start = um_path, um_end = um_path;
        um_start.len = c.g.nkm(um_path.unitig) - um_path.dist; um_start.strand = true;
        um_end.dist = 0; um_end.len = um_path.dist + um_path.len; um_end.strand = true;
        const std::vector<std::string>& cyc = c.g.info[um_path.unitig].cycles;
        for (size_t ci = 0; ci < cyc.size(); ++ci) {
            Path rep = pathFromCompact(c, um_start, cyc[ci], um_end);
            if (!um_path.strand) rep = pathRevComp(rep);
            // evaluatePath (:1167-1201): prefix + repeat + suffix, qualities of the replaced unitig set to the maximum
            Path ext; size_t len_prefix = 0;
            for (size_t j = 0; j < i; ++j) { pathExtend(c, ext, v[j]); len_prefix += v[j].len; }
            for (size_t j = 0; j < rep.ums.size(); ++j) pathExtend(c, ext, rep.ums[j]);
            for (size_t j = i + 1; j < v.size(); ++j) pathExtend(c, ext, v[j]);
            std::string lq = s_qual;
            if (len_prefix > lq.size()) { fprintf(stderr, "oracle: fixRepeats on a path without qualities (std::string::replace would throw in the reference)\n"); abort(); }
            lq.replace(len_prefix, um_path.len + c.k - 1, std::string(rep.l, q_max));
            if (lq.length() == ext.l) ext.qual = lq; // Path::setQuality
            const std::string s = pathToString(c, ext);
            const int64_t d = c.align(s.c_str(), s.length(), ref, ref_len, static_cast<int>(editDistance), MODE_NW, false).editDistance;
            if (d >= 0 && d < editDistance) { editDistance = d; best = ext; }
        }
        if (getenv("ORC_TRACE_REPEATS")) fprintf(stderr, "[orc] fixRepeats unitig %d cycles %zu improved %d\n", um_path.unitig, cyc.size(), best.l != 0 ? 1 : 0);
        if (best.l != 0) { // a better aligning path: continue behind the inserted unitigs (:1283-1292)
            const size_t diff = best.size() - path.size();
            path = best; v = path.ums; s_qual = path.qual;
            i += diff - 1;
        } else {
            for (size_t j = i + 1; j < v.size(); ++j) { if (v[j].unitig == um_path.unitig) ++i; else break; }
        }
    }
    return path;
}

// ---------------------------------------------------------------- candidate selection (src/Alignment.cpp)
std::pair<int, int> selectBest(const Ctx& c, const std::vector<const Path*>& cand, const char* ref, const size_t ref_len, const AlignMode mode, const double cut) {
    // NW: selectBestAlignment (:3-48, norm = max(|cand|, |ref|));  SHW: selectBestPrefixAlignment (:50-147);  HW: selectBestSubstringAlignment (:967-1015)
    double best = 0.0; int best_id = -1, best_end = -1;
    for (size_t i = 0; i < cand.size(); ++i) {
        const std::string s = pathToString(c, *cand[i]);
        const size_t norm = (mode == MODE_NW) ? std::max(s.length(), ref_len) : s.length();
        if (i == 0) {
            const AlignResult a = c.align(s.c_str(), s.length(), ref, ref_len, -1, mode, false);
            best = static_cast<double>(a.editDistance) / norm; best_end = a.endLocations[0]; best_id = 0;
        } else {
            const int kk = static_cast<int>(best * norm + 1); // double -> int as edlibNewAlignConfig(int k, ...) receives it (G5)
            const AlignResult a = c.align(s.c_str(), s.length(), ref, ref_len, kk, mode, false);
            if (a.editDistance >= 0 && (static_cast<double>(a.editDistance) / norm) < best) { best = static_cast<double>(a.editDistance) / norm; best_end = a.endLocations[0]; best_id = static_cast<int>(i); }
        }
    }
    if (mode != MODE_NW && cut > 0.0 && best > cut) return std::make_pair(-1, -1);
    return std::make_pair(best_id, best_end);
}

std::pair<int, int> selectBest(const Ctx& c, const std::vector<Path>& cand, const char* ref, const size_t ref_len, const AlignMode mode, const double cut = -1.0) {
    std::vector<const Path*> v; for (size_t i = 0; i < cand.size(); ++i) v.push_back(&cand[i]);
    return selectBest(c, v, ref, ref_len, mode, cut);
}

// ---------------------------------------------------------------- scoring (src/GraphTraversal.cpp:867-909, :722-772)
double getScorePath(const Ctx& c, const Path& path, const char* ref, const size_t ref_len, const bool terminal) {
    double score = 0.0;
    if (path.length() != 0) {
        const std::string s = pathToString(c, path);
        if (terminal) {
            const AlignResult a = c.align(s.c_str(), s.length(), ref, ref_len, -1, MODE_NW, false);
            score = 1.0 - (static_cast<double>(a.editDistance) / s.length());
        } else if (s.length() >= ref_len) {
            const AlignResult a = c.align(ref, ref_len, s.c_str(), s.length(), -1, MODE_HW, false);
            score = 1.0 - (static_cast<double>(a.editDistance) / ref_len);
        } else {
            const size_t l_ref_len = std::min(ref_len, static_cast<size_t>(s.length() * (1.0 + c.opt.weak_region_len_factor)));
            const AlignResult a = c.align(s.c_str(), s.length(), ref, l_ref_len, -1, MODE_HW, false);
            score = 1.0 - (static_cast<double>(a.editDistance) / s.length());
        }
        score = std::min(std::max(score, 0.0), 1.0);
    }
    return score;
}

std::string getScorePathQual(const Ctx& c, const Path& path, const char* ref, const size_t ref_len, const double score_best, const double score_second_best) {
    const double score_comp = score_best * ((score_best == 0.0) ? 0.0 : (1.0 - (score_second_best / score_best)));
    const std::string s = pathToString(c, path);
    const AlignResult a = c.align(s.c_str(), s.length(), ref, ref_len, -1, MODE_SHW, true);
    const char c_best = getQual(score_best, 0, c.opt.max_qual);
    std::string qual_out(s.length(), getQual(score_comp, c.opt.out_qual, c.opt.max_qual));
    size_t qp = 0, rp = 0;
    for (size_t i = 0; i < a.alignment.size(); ++i) { // walking the CIGAR op by op == walking the alignment move by move
        const unsigned char mv = a.alignment[i];
        if (mv == 0 || mv == 3) { if (s[qp] == ref[rp]) qual_out[qp] = c_best; ++qp; ++rp; }
        else if (mv == 1) ++qp; else ++rp;
    }
    return qual_out;
}

// ---------------------------------------------------------------- DFS (src/GraphTraversal.cpp:456-587)
struct SubGraphOut { std::vector<Path> terminal, non_terminal; double t1, nt1; };

void exploreSubGraph(const Ctx& c, const IdSet& all_pids, const char* ref, const size_t ref_len, const size_t max_len_path,
                     const UM& um, const UM& um_e, const size_t level, std::map<int32_t, bool>& memo, SubGraphOut& out) {
    double score_t1 = 0.0, score_nt1 = 0.0, score_t2 = 0.0, score_nt2 = 0.0;
    std::vector<std::pair<Path, size_t> > stck; // LIFO (G6)
    stck.push_back(std::make_pair(Path(), level));
    while (!stck.empty()) {
        const std::pair<Path, size_t> it = stck.back(); stck.pop_back();
        const UM um_start = it.first.isEmpty() ? um : it.first.back();
        UM succ[4]; char base[4]; int ns;
        c.g.successors(um_start, succ, base, ns);
        if (c.cnt) ++c.cnt->n_expand;
        for (int si = 0; si < ns; ++si) {
            const UM& sc = succ[si];
            std::map<int32_t, bool>::iterator mit = memo.find(sc.unitig);
            if (mit == memo.end()) {
                const bool ok = all_pids.empty() || (c.g.sharedCount(sc.unitig, all_pids) >= c.opt.min_cov_vertices);
                if (c.cnt) c.cnt->n_colour_elem += c.g.cardinality(sc.unitig) + all_pids.size();
                mit = memo.insert(std::make_pair(sc.unitig, ok)).first;
            }
            if (!(c.g.getSharedPids(um_start.unitig, um_start.strand, base[si]) && mit->second)) continue;
            if (!um_e.isEmpty() && sc.unitig == um_e.unitig && um_e.strand == sc.strand) { // terminal
                Path path(it.first);
                UM pref(sc);
                if (pref.strand) { pref.dist = 0; pref.len = um_e.dist + 1; }
                else { pref.dist = um_e.dist; pref.len = c.g.nkm(sc.unitig) - um_e.dist; }
                pathExtend(c, path, pref);
                if (path.length() <= max_len_path) {
                    const double sco = getScorePath(c, path, ref, ref_len, true);
                    if (sco >= score_t1) { if (sco > score_t1) out.terminal.clear(); out.terminal.push_back(path); score_t2 = score_t1; score_t1 = sco; }
                    else if (sco > score_t2) score_t2 = sco;
                }
            }
            { // non-terminal
                Path path(it.first);
                pathExtend(c, path, sc);
                // exploreSubGraph descends `level` unitigs (:531-535); exploreSubGraphLong until the sub-path spans k * large_k_factor (:594,:669-671)
                const bool deeper = c.opt.long_read_correct ? (path.length() < static_cast<size_t>(static_cast<double>(c.k) * c.opt.large_k_factor)) : (it.second != 0);
                if (deeper) stck.push_back(std::make_pair(path, it.second ? it.second - 1 : 0));
                else if (c.g.nbSuccessors(sc) > 0) {
                    const double sco = getScorePath(c, path, ref, ref_len, false);
                    if (sco >= score_nt1) { if (sco > score_nt1) out.non_terminal.clear(); out.non_terminal.push_back(path); score_nt2 = score_nt1; score_nt1 = sco; }
                    else if (sco > score_nt2) score_nt2 = sco;
                }
            }
        }
    }
    for (size_t i = 0; i < out.terminal.size(); ++i) { Path& p = out.terminal[i]; const std::string q = getScorePathQual(c, p, ref, ref_len, score_t1, score_t2); if (q.length() == p.l) p.qual = q; }
    for (size_t i = 0; i < out.non_terminal.size(); ++i) { Path& p = out.non_terminal[i]; const std::string q = getScorePathQual(c, p, ref, ref_len, score_nt1, score_nt2); if (q.length() == p.l) p.qual = q; }
    out.t1 = score_t1; out.nt1 = score_nt1;
}

// explore() lambdas (src/GraphTraversal.cpp:41-93 and :251-304)
void explore(const Ctx& c, const IdSet& all_pids, const char* ref, const size_t ref_len, const UM& um_e, const Path& path, const size_t max_len_path,
             std::map<int32_t, bool>& memo, std::vector<Path>& terminal, std::vector<Path>& non_terminal) {
    const UM& um = path.back();
    const size_t path_len = path.length();
    const bool non_empty_path = (path_len > (um.len + c.k - 1)) && !um.isEmpty();
    const size_t path_len_prefix = non_empty_path ? (path_len - um.len - c.k + 1) : 0;
    size_t end_pos_ref = 0;
    if (non_empty_path) {
        const std::string s = pathToString(c, path);
        const AlignResult a = c.align(s.c_str(), path_len_prefix, ref, ref_len, -1, MODE_SHW, false);
        end_pos_ref = static_cast<size_t>(a.endLocations[0] + 1);
    }
    if ((ref_len - end_pos_ref) != 0 && path_len < max_len_path) {
        SubGraphOut o;
        exploreSubGraph(c, all_pids, ref + end_pos_ref, ref_len - end_pos_ref, max_len_path - path_len_prefix, um, um_e, 3, memo, o);
        if (!o.terminal.empty() && o.t1 < c.opt.min_score) o.terminal.clear();
        if (!o.non_terminal.empty() && o.nt1 < c.opt.min_score) o.non_terminal.clear();
        if (o.non_terminal.size() > 1) {
            const int best = selectBest(c, o.non_terminal, ref + end_pos_ref, ref_len - end_pos_ref, MODE_HW).first;
            std::vector<Path> one(1, o.non_terminal[best]);
            o.non_terminal.swap(one);
        }
        terminal.swap(o.terminal); non_terminal.swap(o.non_terminal);
    }
}

void extendBy(const Ctx& c, Path& p_ext, const Path& sub) { // the "P (+) Q" loops (src/GraphTraversal.cpp:379-390, 397-406)
    size_t j = 0;
    for (size_t i = 0; i < sub.ums.size(); ++i) {
        const UM& um = sub.ums[i];
        pathExtend(c, p_ext, um, j <= sub.qual.size() ? sub.qual.substr(j, um.len + c.k - 1) : std::string());
        j += um.len;
    }
}

void resizeToBest(const Ctx& c, std::vector<Path>& v, const char* ref, const size_t ref_len) { // resizeVector (:11-22, :221-232)
    if (v.size() <= 1) return;
    const int best = selectBest(c, v, ref, ref_len, MODE_SHW).first;
    std::vector<Path> one(1, v[best]);
    v.swap(one);
}

Path startSuffix(const Ctx& c, const UM& um_s, UM& um_start_tmp) { // src/GraphTraversal.cpp:113-125, 325-338
    um_start_tmp = um_s;
    if (um_start_tmp.strand) { um_start_tmp.dist += um_start_tmp.len - 1; um_start_tmp.len = c.g.nkm(um_s.unitig) - um_start_tmp.dist; }
    else { um_start_tmp.len = um_s.dist + 1; um_start_tmp.dist = 0; }
    Path p;
    pathExtend(c, p, um_start_tmp, std::string(um_start_tmp.len + c.k - 1, getQual(1.0, 0, c.opt.max_qual)));
    return p;
}

// src/GraphTraversal.cpp:212-454
std::vector<Path> explorePathsBFS2(const Ctx& c, const IdSet& all_pids, const char* ref, const size_t ref_len, const UM& um_s, const UM& um_e) {
    std::vector<Path> v, v_tmp;
    if (!um_s.isEmpty() && !um_e.isEmpty() && c.g.hasSharedPids(um_s.unitig) && c.g.hasSharedPids(um_e.unitig)) {
        const size_t level = 4;
        const size_t min_len_path = getMinMaxLength(ref_len - c.k, c.opt.weak_region_len_factor).first + c.k;
        const size_t max_len_path = std::max(getMinMaxLength(ref_len - c.k, c.opt.weak_region_len_factor).second, static_cast<size_t>(10)) + c.k;
        const size_t max_paths = 1024, max_sz_stck = 512;
        std::map<int32_t, bool> memo;
        std::deque<Path> q;
        {
            UM um_start_tmp;
            Path p_q = startSuffix(c, um_s, um_start_tmp);
            if (c.g.sameUnitig(um_s, um_e) && um_s.strand == um_e.strand && um_start_tmp.dist <= um_e.dist) {
                const size_t len = (um_start_tmp.len + c.k - 1) - (um_e.strand ? (c.g.usize(um_e.unitig) - um_e.dist - c.k) : um_e.dist);
                if (len >= min_len_path && len <= max_len_path) {
                    UM back_tmp(um_start_tmp);
                    if (back_tmp.strand) back_tmp.len = um_e.dist - back_tmp.dist + 1;
                    else { back_tmp.dist = um_e.dist; back_tmp.len -= um_e.dist; }
                    Path p;
                    pathExtend(c, p, back_tmp, std::string(back_tmp.len + c.k - 1, getQual(1.0, 0, c.opt.max_qual)));
                    v.push_back(p);
                }
            }
            q.push_back(p_q);
        }
        auto flush = [&]() {
            for (size_t i = 0; i < v_tmp.size(); ++i) {
                if (v_tmp[i].length() >= min_len_path && v_tmp[i].length() <= max_len_path) {
                    if (v.size() + 1 >= max_paths) resizeToBest(c, v, ref, ref_len);
                    v.push_back(v_tmp[i]);
                }
            }
            v_tmp.clear();
        };
        while (!q.empty()) {
            const Path p = q.front(); q.pop_front();
            if (p.length() < max_len_path) {
                std::vector<Path> term, nterm;
                explore(c, all_pids, ref, ref_len, um_e, p, max_len_path, memo, term, nterm);
                for (size_t i = 0; i < term.size(); ++i) { Path p_ext(p); extendBy(c, p_ext, term[i]); v_tmp.push_back(p_ext); }
                for (size_t i = 0; i < nterm.size(); ++i) {
                    if (c.opt.long_read_correct ? (nterm[i].length() >= static_cast<size_t>(static_cast<double>(c.k) * c.opt.large_k_factor)) : (nterm[i].size() == level)) { // :395
                        Path p_ext(p); extendBy(c, p_ext, nterm[i]);
                        q.push_back(p_ext);
                        if (q.size() >= max_sz_stck) { std::vector<Path> tmp(q.begin(), q.end()); q.clear(); resizeToBest(c, tmp, ref, ref_len); for (size_t j = 0; j < tmp.size(); ++j) q.push_back(tmp[j]); }
                    }
                }
                if (v_tmp.size() >= max_paths) flush();
            }
        }
        flush();
    }
    if (!v.empty()) {
        if (v.size() > 1) { const int b = selectBest(c, v, ref, ref_len, MODE_NW).first; std::vector<Path> one(1, v[b]); v.swap(one); }
        v[0] = fixRepeats(c, v[0], ref, ref_len);
    }
    return v;
}

// src/GraphTraversal.cpp:3-210
std::vector<Path> explorePathsBFS(const Ctx& c, const IdSet& all_pids, const char* ref, const size_t ref_len, const UM& um_s) {
    std::vector<Path> v, v_tmp;
    if (!um_s.isEmpty() && c.g.hasSharedPids(um_s.unitig)) {
        const size_t level = 4;
        const size_t min_len_path = getMinMaxLength(ref_len - c.k, c.opt.weak_region_len_factor).first + c.k;
        const size_t max_len_path = std::max(getMinMaxLength(ref_len - c.k, c.opt.weak_region_len_factor).second, static_cast<size_t>(10)) + c.k;
        const size_t max_paths = 1024, max_sz_stck = 512;
        std::map<int32_t, bool> memo;
        std::deque<Path> q;
        {
            UM um_start_tmp;
            Path p_q = startSuffix(c, um_s, um_start_tmp);
            if ((um_start_tmp.len + c.k - 1) >= min_len_path) {
                UM back(um_start_tmp);
                if ((back.len + c.k - 1) > max_len_path) {
                    if (!back.strand) back.dist = static_cast<uint32_t>(back.len - (max_len_path - c.k + 1));
                    back.len = static_cast<uint32_t>(max_len_path - c.k + 1);
                }
                Path p;
                pathExtend(c, p, back, std::string(back.len + c.k - 1, getQual(1.0, 0, c.opt.max_qual)));
                v.push_back(p);
            }
            q.push_back(p_q);
        }
        auto flush = [&]() { for (size_t i = 0; i < v_tmp.size(); ++i) { pathPrunePrefix(c, v_tmp[i], max_len_path); v.push_back(v_tmp[i]); } v_tmp.clear(); };
        const UM no_end;
        while (!q.empty()) {
            const Path p = q.front(); q.pop_front();
            if (p.length() < max_len_path) {
                std::vector<Path> term, nterm;
                explore(c, all_pids, ref, ref_len, no_end, p, max_len_path, memo, term, nterm);
                for (size_t i = 0; i < nterm.size(); ++i) {
                    const Path& path = nterm[i];
                    Path p_ext(p);
                    size_t j = 0;
                    for (size_t u = 0; u < path.ums.size(); ++u) {
                        pathExtend(c, p_ext, path.ums[u], j <= path.qual.size() ? path.qual.substr(j, path.ums[u].len + c.k - 1) : std::string());
                        if (p_ext.length() >= min_len_path && p_ext.length() <= max_len_path) v_tmp.push_back(p_ext);
                        j += path.ums[u].len;
                    }
                    if (c.opt.long_read_correct ? (path.length() >= static_cast<size_t>(static_cast<double>(c.k) * c.opt.large_k_factor)) : (path.size() == level)) { // :174
                        q.push_back(p_ext);
                        if (q.size() >= max_sz_stck) { std::vector<Path> tmp(q.begin(), q.end()); q.clear(); resizeToBest(c, tmp, ref, ref_len); for (size_t t = 0; t < tmp.size(); ++t) q.push_back(tmp[t]); }
                    }
                }
                if (v_tmp.size() >= max_paths) flush();
            }
        }
        flush();
    }
    if (!v.empty()) {
        if (v.size() > 1) { const int b = selectBest(c, v, ref, ref_len, MODE_NW).first; std::vector<Path> one(1, v[b]); v.swap(one); }
        v[0] = fixRepeats(c, v[0], ref, ref_len);
    }
    return v;
}

// ---------------------------------------------------------------- extractSemiWeakPaths (src/Correction.cpp:3-157)
struct SemiWeak { std::vector<Path> complete, partial; };

SemiWeak extractSemiWeakPaths(const Ctx& c, const std::string& s, const IdSet& all_pids, const Anchor& um_solid_start, const Anchor& um_solid_end,
                              const std::vector<Anchor>& v_um_weak, size_t i_weak) {
    SemiWeak paths;
    std::vector<std::pair<Path, size_t> > paths1, paths2;
    const bool no_end = um_solid_end.second.isEmpty();
    const size_t k = c.k;
    const size_t pos_um_solid2 = no_end ? s.length() - k : um_solid_end.first;
    const size_t len_weak_region = (pos_um_solid2 - um_solid_start.first) + k;
    const size_t max_len_weak_region = c.opt.long_read_correct ? c.opt.max_len_weak_region2 : c.opt.max_len_weak_region1; // :23
    const size_t max_paths = 512;
    size_t next_weak_pos = 0;
    bool begin = true, end = false;
    {
        Path tmp;
        pathExtend(c, tmp, um_solid_start.second, std::string(um_solid_start.second.len + k - 1, getQual(1.0, 0, c.opt.max_qual)));
        paths1.push_back(std::make_pair(tmp, um_solid_start.first));
    }
    while (i_weak < v_um_weak.size() && v_um_weak[i_weak].first < um_solid_start.first) ++i_weak;
    if (i_weak < v_um_weak.size()) next_weak_pos = std::max(v_um_weak[i_weak].first, um_solid_start.first + k);
    while (!paths1.empty() && !end) {
        std::vector<Path> g_prev; bool g_valid = false;
        if (i_weak < v_um_weak.size()) { while (i_weak < v_um_weak.size() && v_um_weak[i_weak].first < (pos_um_solid2 - k) && v_um_weak[i_weak].first < next_weak_pos) ++i_weak; }
        else i_weak = v_um_weak.size();
        // [D2] stable sort by the mapped string of the last unitig
        std::stable_sort(paths1.begin(), paths1.end(), [&](const std::pair<Path, size_t>& a, const std::pair<Path, size_t>& b) { return c.g.mapped(a.first.back()) < c.g.mapped(b.first.back()); });
        end = (i_weak == v_um_weak.size()) || (v_um_weak[i_weak].first >= (pos_um_solid2 - k));
        const size_t target_pos = end ? pos_um_solid2 : v_um_weak[i_weak].first;
        for (size_t i = 0; i < paths1.size(); ++i) {
            const std::pair<Path, size_t>& p = paths1[i];
            const size_t l_len = (target_pos - p.second) + k;
            if (i == 0 || paths1[i].first.back() != paths1[i - 1].first.back()) {
                g_prev.clear(); g_valid = false;
                const UM um_start = begin ? um_solid_start.second : p.first.back();
                if (end) {
                    if (no_end) { if (l_len <= (max_len_weak_region / 2)) { g_prev = explorePathsBFS(c, all_pids, s.c_str() + p.second, l_len, um_start); g_valid = true; } }
                    else if (l_len <= max_len_weak_region) { g_prev = explorePathsBFS2(c, all_pids, s.c_str() + p.second, l_len, um_start, um_solid_end.second); g_valid = true; }
                } else if (l_len <= max_len_weak_region) { g_prev = explorePathsBFS2(c, all_pids, s.c_str() + p.second, l_len, um_start, v_um_weak[i_weak].second); g_valid = true; }
            }
            if (g_valid && !g_prev.empty()) {
                for (size_t j = 0; j < g_prev.size(); ++j) { Path tmp = p.first; pathMerge(c, tmp, g_prev[j]); paths2.push_back(std::make_pair(tmp, target_pos)); }
            } else paths.partial.push_back(p.first);
        }
        if (!end) next_weak_pos = v_um_weak[i_weak].first + k;
        begin = false;
        paths1.swap(paths2); paths2.clear();
        if (!end && paths1.size() > max_paths) {
            std::vector<const Path*> v_ptr; for (size_t i = 0; i < paths1.size(); ++i) v_ptr.push_back(&paths1[i].first);
            const int best_id = selectBest(c, v_ptr, s.c_str() + um_solid_start.first, len_weak_region, MODE_SHW, -1.0).first;
            paths2.push_back(paths1[best_id]);
            paths1.swap(paths2); paths2.clear();
        }
    }
    for (size_t i = 0; i < paths1.size(); ++i) paths.complete.push_back(paths1[i].first);
    return paths;
}

// ---------------------------------------------------------------- ResultCorrection (src/ResultCorrection.hpp)
struct ResultCorrection {
    std::set<uint32_t> pos; std::string seq, qual; IdSet all_pids; size_t old_seq_len; bool is_corrected;
    explicit ResultCorrection(size_t l) : old_seq_len(l), is_corrected(false) {}
    void addRange(uint64_t a, uint64_t b) { for (uint64_t i = a; i < b; ++i) pos.insert(static_cast<uint32_t>(i)); }
    void reverseComplement() { // :72-88
        if (seq.length() != 0) {
            std::set<uint32_t> t; for (std::set<uint32_t>::const_iterator it = pos.begin(); it != pos.end(); ++it) t.insert(static_cast<uint32_t>(old_seq_len - *it - 1));
            pos.swap(t); seq = revcomp(seq); std::reverse(qual.begin(), qual.end());
        }
    }
    size_t lenCorrected(size_t p) const { // :117-128
        size_t next = p; std::set<uint32_t>::const_iterator it = pos.lower_bound(static_cast<uint32_t>(p));
        for (; it != pos.end() && *it < old_seq_len && *it == next; ++it) ++next;
        return next - p;
    }
    size_t lenUncorrected(size_t p) const { // :130-142
        if (p >= old_seq_len) return 0;
        std::set<uint32_t>::const_iterator it = pos.lower_bound(static_cast<uint32_t>(p));
        if (it == pos.end()) return old_seq_len - p;
        return std::min(static_cast<size_t>(*it), old_seq_len) - p;
    }
};

// ---------------------------------------------------------------- generateConsensus (src/Alignment.cpp:309-470)
struct CigarCursor { std::vector<std::pair<size_t, char> > ops; size_t idx, qpos, rpos; CigarCursor() : idx(0), qpos(0), rpos(0) {} };

void cigarOps(const std::vector<unsigned char>& aln, CigarCursor& cc) {
    static const char code[4] = {'M', 'I', 'D', 'M'};
    for (size_t i = 0; i < aln.size();) { size_t j = i; while (j < aln.size() && code[aln[j]] == code[aln[i]]) ++j; cc.ops.push_back(std::make_pair(j - i, code[aln[i]])); i = j; }
}

// moveIntoCIGAR (:354-411) at op granularity: an op is consumed only once the cursor has fully passed it
std::pair<std::pair<size_t, size_t>, size_t> moveIntoCigar(const size_t start, const size_t end, CigarCursor& cc) {
    size_t read_pos_start = cc.qpos, read_pos_end;
    while (cc.idx != cc.ops.size() && cc.rpos < start) {
        const size_t l = cc.ops[cc.idx].first; const char op = cc.ops[cc.idx].second;
        if (op == 'M') {
            if (cc.rpos + l > start) { read_pos_start = cc.qpos + (start - cc.rpos); break; }
            cc.qpos += l; cc.rpos += l;
        } else if (op == 'I') cc.qpos += l; else cc.rpos += l;
        ++cc.idx; read_pos_start = cc.qpos;
    }
    read_pos_end = read_pos_start;
    while (cc.idx != cc.ops.size() && cc.rpos < end) {
        const size_t l = cc.ops[cc.idx].first; const char op = cc.ops[cc.idx].second;
        if (op == 'M') {
            if (cc.rpos + l > end) return std::make_pair(std::make_pair(read_pos_start, cc.qpos + (end - cc.rpos)), end);
            cc.qpos += l; cc.rpos += l;
        } else if (op == 'I') cc.qpos += l; else cc.rpos += l;
        ++cc.idx; read_pos_end = cc.qpos;
    }
    return std::make_pair(std::make_pair(read_pos_start, read_pos_end), cc.rpos);
}

std::pair<std::string, std::string> generateConsensus(const Ctx& c, const ResultCorrection* fw_s, const ResultCorrection* bw_s, const std::string& ref_seq, const double max_norm) {
    if (bw_s->pos.size() == 0 && fw_s->pos.size() != 0) return std::make_pair(fw_s->seq, fw_s->qual);
    else if (fw_s->pos.size() == 0 && bw_s->pos.size() != 0) return std::make_pair(bw_s->seq, bw_s->qual);
    else if (fw_s->pos.size() + bw_s->pos.size() == 0) return std::make_pair(std::string(), std::string());
    if (bw_s->pos.size() > fw_s->pos.size()) std::swap(fw_s, bw_s);
    const AlignResult a_fw = c.align(fw_s->seq.c_str(), fw_s->seq.length(), ref_seq.c_str(), ref_seq.length(), -1, MODE_NW, true);
    const AlignResult a_bw = c.align(bw_s->seq.c_str(), bw_s->seq.length(), ref_seq.c_str(), ref_seq.length(), -1, MODE_NW, true);
    const double n_fw = static_cast<double>(a_fw.editDistance) / std::max(fw_s->seq.length(), ref_seq.length());
    const double n_bw = static_cast<double>(a_bw.editDistance) / std::max(bw_s->seq.length(), ref_seq.length());
    if (max_norm > 0.0 && (n_fw > max_norm || n_bw > max_norm)) {
        if (n_fw > max_norm && n_bw > max_norm) return std::make_pair(std::string(), std::string());
        if (n_fw > max_norm) return std::make_pair(bw_s->seq, bw_s->qual);
        return std::make_pair(fw_s->seq, fw_s->qual);
    }
    CigarCursor cf, cb; cigarOps(a_fw.alignment, cf); cigarOps(a_bw.alignment, cb);
    std::string ss, sq;
    size_t i = 0;
    while (i < ref_seq.length()) {
        int64_t len_fw = static_cast<int64_t>(fw_s->lenCorrected(i)), len_bw = static_cast<int64_t>(bw_s->lenCorrected(i));
        std::pair<std::pair<size_t, size_t>, size_t> pr;
        if ((len_fw + len_bw) <= 0) {
            len_fw = static_cast<int64_t>(fw_s->lenUncorrected(i)); len_bw = static_cast<int64_t>(bw_s->lenUncorrected(i));
            if (len_fw > len_bw || len_fw <= 0) len_fw = -1; else len_bw = -1;
        }
        if (len_fw >= len_bw) {
            pr = moveIntoCigar(i, i + len_fw, cf);
            if (pr.first.second > pr.first.first) { ss += fw_s->seq.substr(pr.first.first, pr.first.second - pr.first.first); sq += fw_s->qual.substr(pr.first.first, pr.first.second - pr.first.first); }
        } else {
            pr = moveIntoCigar(i, i + len_bw, cb);
            if (pr.first.second > pr.first.first) { ss += bw_s->seq.substr(pr.first.first, pr.first.second - pr.first.first); sq += bw_s->qual.substr(pr.first.first, pr.first.second - pr.first.first); }
        }
        i = pr.second;
    }
    if (max_norm > 0.0) {
        const AlignResult a = c.align(ss.c_str(), ss.length(), ref_seq.c_str(), ref_seq.length(), -1, MODE_NW, false, /*iupac=*/false); // edlibDefaultAlignConfig() (:460)
        const double n = static_cast<double>(a.editDistance) / std::max(ss.length(), ref_seq.length());
        if (n > max_norm) return std::make_pair(fw_s->seq, fw_s->qual);
    }
    return std::make_pair(ss, sq);
}

// ---------------------------------------------------------------- chooseColors (src/Correction.cpp:215-429); only all_pids is read downstream
typedef std::map<int32_t, bool> AnchorSets; // unitig -> "is non-branching" (stands for unordered_map<const SharedPairID*, pair<const PairID*, bool>>)

IdSet chooseColors(const Ctx& c, const AnchorSets& s_pid_s, const AnchorSets& s_pid_e, const AnchorSets& s_pid_w) {
    const Graph& g = c.g;
    const AnchorSets* v_s_pid[3] = {&s_pid_w, &s_pid_e, &s_pid_s};
    IdSet a_pid[6];
    std::set<int32_t> s_spid;
    for (size_t i = 0; i < 3; ++i) for (AnchorSets::const_iterator it = v_s_pid[i]->begin(); it != v_s_pid[i]->end(); ++it) {
        const int shift = static_cast<int>(i) + (it->second ? 3 : 0);
        const IdSet* gp = g.globalSet(it->first);
        a_pid[shift] = set_union(a_pid[shift], gp ? *gp : g.info[it->first].local); // G2: only the global set when there is one
        if (c.cnt) c.cnt->n_colour_elem += gp ? gp->size() : g.info[it->first].local.size();
        if (g.cardinality(it->first) >= c.opt.min_cov_vertices) s_spid.insert(it->first);
    }
    IdSet all_pids;
    const IdSet a_pid_pos[3] = {set_union(a_pid[0], a_pid[3]), set_union(a_pid[1], a_pid[4]), set_union(a_pid[2], a_pid[5])};
    const IdSet a01 = set_inter(a_pid_pos[0], a_pid_pos[1]), a12 = set_inter(a_pid_pos[1], a_pid_pos[2]), a02 = set_inter(a_pid_pos[0], a_pid_pos[2]);
    IdSet a_nobranch = set_union(set_union(a_pid[3], a_pid[4]), a_pid[5]);
    const IdSet a_nobranch_cpy = a_nobranch;
    IdSet inter2, inter3, a_branch, a_pid2[6];
    const size_t cov = 30;
    size_t nb_unselected = s_spid.size();
    std::vector<std::pair<int32_t, int> > v_spids;
    for (std::set<int32_t>::const_iterator it = s_spid.begin(); it != s_spid.end(); ++it) v_spids.push_back(std::make_pair(*it, static_cast<int>(std::min(cov, g.cardinality(*it)))));
    const char* d1 = getenv("RTK_D1_ORDER"); const bool d1_desc = d1 && !strcmp(d1, "desc"); // [D1] as a switch (like rtk_opts::d1_desc on the device side): ties by unitig id, ascending (default) or descending
    std::sort(v_spids.begin(), v_spids.end(), [&](const std::pair<int32_t, int>& a, const std::pair<int32_t, int>& b) { // [D1]
        const size_t ca = g.cardinality(a.first), cb = g.cardinality(b.first);
        return ca != cb ? ca < cb : (d1_desc ? a.first > b.first : a.first < b.first);
    });
    for (int i = 5; i >= 0; --i) {
        if (nb_unselected == 0) break;
        if (i == 5) { inter3 = set_inter(a01, a12); a_pid2[5] = set_inter(a_nobranch, inter3); }
        else if (i == 4) { inter2 = set_union(set_union(a01, a12), a02); a_nobranch = set_diff(a_nobranch, a_pid2[5]); a_pid2[4] = set_inter(a_nobranch, inter2); }
        else if (i == 3) { a_nobranch = set_diff(a_nobranch, a_pid2[4]); a_pid2[3] = a_nobranch; }
        else if (i == 2) { a_branch = set_union(set_union(a_pid[0], a_pid[1]), a_pid[2]); a_branch = set_diff(a_branch, a_nobranch_cpy); a_pid2[2] = set_inter(a_branch, inter3); }
        else if (i == 1) { a_branch = set_diff(a_branch, a_pid2[2]); a_pid2[1] = set_inter(a_branch, inter2); }
        else { a_branch = set_diff(a_branch, a_pid2[1]); a_pid2[0] = a_branch; }
        if (!a_pid2[i].empty()) {
            nb_unselected = 0;
            IdSet curr = a_pid2[i];
            for (size_t j = 0; j < v_spids.size(); ++j) {
                std::pair<int32_t, int>& p = v_spids[j];
                if (p.second > 0 && (i == 0 || g.sharedCount(p.first, curr) >= 1)) {
                    const size_t min_cov = std::min(cov, g.cardinality(p.first));
                    p.second = static_cast<int>(min_cov - std::min(g.sharedCount(p.first, all_pids), min_cov));
                    if (p.second > 0) {
                        const size_t all_card = all_pids.size();
                        IdSet pid;
                        const IdSet* gp = g.globalSet(p.first);
                        if (gp) pid = set_inter(*gp, curr);
                        pid = set_union(pid, set_inter(g.info[p.first].local, curr));
                        if (pid.size() > static_cast<size_t>(p.second)) pid.resize(static_cast<size_t>(p.second)); // lowest ids first
                        all_pids = set_union(all_pids, pid);
                        curr = set_diff(curr, pid);
                        p.second -= std::min(static_cast<int>(all_pids.size() - all_card), p.second);
                    }
                }
                nb_unselected += static_cast<size_t>(p.second > 0);
            }
        }
    }
    return all_pids;
}

// Bifrost Kmer(const char*) 2-bit code of any character (used by the end-k-mer test, src/Correction.cpp:720-724)
inline int bifrostCode(char ch) { const int x = (ch & 4) >> 1; return x + ((x ^ (ch & 2)) >> 1); }

} // namespace

// ---------------------------------------------------------------- correctSequence (src/Correction.cpp:159-958)
std::pair<std::string, std::string> correctSequence(const Graph& g, const Opt& opt, const std::string& s_fw, const std::string& q_fw,
                                                    const std::vector<Anchor>& v_um_solid, const std::vector<Anchor>& v_um_weak, Counters* cnt) {
    const Ctx c(g, opt, cnt);
    const size_t k = c.k;
    if (s_fw.length() <= k || v_um_solid.empty() || v_um_solid.size() == s_fw.length() - k + 1) {
        if (opt.long_read_correct) return std::make_pair(s_fw, q_fw); // :167
        if (v_um_solid.size() == s_fw.length() - k + 1) return std::make_pair(s_fw, std::string(s_fw.length(), getQual(1.0, 0, opt.max_qual)));
        return std::make_pair(s_fw, std::string(s_fw.length(), getQual(0.0, 0, opt.max_qual)));
    }
    const size_t seq_len = s_fw.length();
    const std::string s_bw(revcomp(s_fw));
    const bool lrc = opt.long_read_correct;
    const size_t max_len_weak_anchors = lrc ? opt.max_len_weak_region2 : opt.max_len_weak_region1; // :177
    const char q_min = getQual(0.0, 0, opt.max_qual), q_max = getQual(1.0, 0, opt.max_qual);
    std::string q_bw = q_fw;
    size_t prev_pos = v_um_solid[0].first, i_solid = 0, i_weak = 0;
    std::string corrected_s, corrected_q;
    std::vector<Anchor> v_um_solid_rev(v_um_solid), v_um_weak_rev(v_um_weak);
    std::reverse(v_um_solid_rev.begin(), v_um_solid_rev.end());
    std::reverse(v_um_weak_rev.begin(), v_um_weak_rev.end());
    std::reverse(q_bw.begin(), q_bw.end());
    for (size_t i = 0; i < v_um_solid_rev.size(); ++i) { v_um_solid_rev[i].first = seq_len - v_um_solid_rev[i].first - k; v_um_solid_rev[i].second.strand = !v_um_solid_rev[i].second.strand; }
    for (size_t i = 0; i < v_um_weak_rev.size(); ++i) { v_um_weak_rev[i].first = seq_len - v_um_weak_rev[i].first - k; v_um_weak_rev[i].second.strand = !v_um_weak_rev[i].second.strand; }

    // the `correct` lambda (src/Correction.cpp:431-753), pass 1
    auto hasMinQual = [](const std::string& s, const std::string& q, const size_t start, const size_t end, const char min_q) { // src/Correction.hpp:45-52
        bool has = true;
        for (size_t i = start; i < end && has; ++i) has = (q[i] >= min_q) || !isDNA(s[i]);
        return has;
    };
    auto correct = [&](const std::string& s, const std::string& q, const std::vector<Anchor>& v_s, const std::vector<Anchor>& v_w, const size_t i_s, const size_t i_w, const ResultCorrection* rc) -> ResultCorrection {
        if (cnt) ++cnt->n_regions;
        const bool has_end_pt = (i_s + 1) < v_s.size();
        Anchor um_solid1 = v_s[i_s];
        Anchor um_solid2 = has_end_pt ? v_s[i_s + 1] : Anchor(s.length() - k, UM());
        size_t len_weak_region = um_solid2.first - um_solid1.first + k;
        const int64_t min_start = static_cast<int64_t>(um_solid1.first - opt.insert_sz); // wraps below insert_sz (G1)
        const int64_t min_end = static_cast<int64_t>(um_solid2.first + opt.insert_sz);
        const uint64_t u_min_start = static_cast<uint64_t>(min_start), u_min_end = static_cast<uint64_t>(min_end);
        const char* s_start = s.c_str() + um_solid1.first;
        ResultCorrection res(len_weak_region);
        std::string s_corrected, q_corrected;
        std::vector<Anchor> l_v_w;
        IdSet all_pids;
        const double max_cov_d = static_cast<double>(opt.max_km_cov);
        auto setUncorrected = [&](const size_t pos, const size_t len, const char qual) { s_corrected = s.substr(pos, len); q_corrected = lrc ? q.substr(pos, len) : std::string(len_weak_region, qual); }; // :459-463
        auto addUncorrected = [&](const size_t pos, const size_t len, const char qual) { s_corrected += s.substr(pos, len); q_corrected += lrc ? q.substr(pos, len) : std::string(len_weak_region, qual); }; // :465-469
        auto middle = [&](AnchorSets* s_spid_m) { // :563-585 and :593-604
            if (!v_w.empty()) {
                const size_t pos_end = has_end_pt ? v_s[i_s + 1].first : s.length();
                const size_t v_w_sz = v_w.size();
                size_t i_w_s = i_w - static_cast<size_t>((i_w != 0) && (i_w >= v_w_sz));
                while (i_w_s < v_w_sz && v_w[i_w_s].first < v_s[i_s].first) ++i_w_s;
                for (; i_w_s < v_w_sz && v_w[i_w_s].first < pos_end; ++i_w_s) {
                    l_v_w.push_back(v_w[i_w_s]);
                    const int32_t u = v_w[i_w_s].second.unitig;
                    if (s_spid_m && g.kmerCoverage(u) < max_cov_d) s_spid_m->insert(std::make_pair(u, !g.isBranching(u)));
                }
            }
        };
        if (rc == nullptr) {
            AnchorSets s_spid_l, s_spid_m, s_spid_r;
            auto consider = [&](AnchorSets& m, const UM& um, size_t& nb_branching) {
                const int32_t u = um.unitig;
                if (g.kmerCoverage(u) < max_cov_d && (!g.isBranching(u) || nb_branching < 5)) {
                    const bool unseen = m.insert(std::make_pair(u, !g.isBranching(u))).second;
                    nb_branching += static_cast<size_t>(unseen && g.isBranching(u));
                }
            };
            { // left side (:476-516)
                size_t nb_branching = 0;
                for (int64_t i_s_s = static_cast<int64_t>(i_s); i_s_s >= 0 && static_cast<uint64_t>(v_s[i_s_s].first) > u_min_start; --i_s_s) consider(s_spid_l, v_s[i_s_s].second, nb_branching);
                const size_t v_w_sz = v_w.size();
                size_t i_w_s = i_w - static_cast<size_t>((i_w != 0) && (i_w >= v_w_sz));
                while (i_w_s > 0 && static_cast<uint64_t>(v_w[i_w_s].first) > u_min_start) --i_w_s;
                for (; i_w_s < v_w_sz && v_w[i_w_s].first < v_s[i_s].first; ++i_w_s) consider(s_spid_l, v_w[i_w_s].second, nb_branching);
            }
            if (has_end_pt) { // right side (:518-561)
                size_t nb_branching = 0;
                for (size_t i_s_e = i_s + 1; i_s_e < v_s.size() && static_cast<uint64_t>(v_s[i_s_e].first) < u_min_end; ++i_s_e) consider(s_spid_r, v_s[i_s_e].second, nb_branching);
                const size_t v_w_sz = v_w.size();
                size_t i_w_s = i_w - static_cast<size_t>((i_w != 0) && (i_w >= v_w_sz));
                while (i_w_s < v_w_sz && v_w[i_w_s].first < v_s[i_s + 1].first) ++i_w_s;
                for (; i_w_s < v_w_sz && static_cast<uint64_t>(v_w[i_w_s].first) < u_min_end; ++i_w_s) consider(s_spid_r, v_w[i_w_s].second, nb_branching);
            }
            middle(&s_spid_m);
            all_pids = chooseColors(c, s_spid_l, s_spid_r, s_spid_m);
            res.all_pids = all_pids;
        } else { middle(nullptr); all_pids = rc->all_pids; res.all_pids = all_pids; }

        const size_t card_pids = all_pids.size();
        AmbVec v_ambiguity; // :635-637,:657-659,:677-679,:703-705
        auto addAmbiguity = [&](const Path& path, size_t offset) { const AmbVec v = getAmbiguityVector(c, path); for (size_t i = 0; i < v.size(); ++i) v_ambiguity.push_back(std::make_pair(offset + v[i].first, v[i].second)); };
        SemiWeak paths1;
        if (card_pids >= opt.min_cov_vertices) paths1 = extractSemiWeakPaths(c, s, all_pids, um_solid1, um_solid2, l_v_w, 0);
        if (paths1.complete.empty()) {
            size_t i_w_s = 0;
            while (paths1.complete.empty() && !paths1.partial.empty() && !l_v_w.empty() && card_pids >= opt.min_cov_vertices) { // :619-651
                const std::pair<int, int> align = selectBest(c, paths1.partial, s.c_str() + um_solid1.first, len_weak_region, MODE_SHW, opt.weak_region_len_factor);
                if (align.first == -1) break;
                {
                    const size_t next_pos = um_solid1.first + align.second + k;
                    while (i_w_s < l_v_w.size() && l_v_w[i_w_s].first < next_pos) ++i_w_s;
                    if (i_w_s >= l_v_w.size() || l_v_w[i_w_s].first >= um_solid2.first - k || (l_v_w[i_w_s].first - um_solid1.first) >= max_len_weak_anchors) break;
                }
                const Path& best = paths1.partial[align.first];
                addAmbiguity(best, s_corrected.length());
                s_corrected += pathToString(c, best) + s.substr(um_solid1.first + align.second + 1, l_v_w[i_w_s].first - um_solid1.first - align.second - 1);
                q_corrected += best.qual;
                q_corrected += lrc ? q.substr(um_solid1.first + align.second + 1, l_v_w[i_w_s].first - um_solid1.first - align.second - 1) : std::string(l_v_w[i_w_s].first - um_solid1.first - align.second - 1, q_min); // :642-643
                res.addRange(um_solid1.first - v_s[i_s].first, um_solid1.first + align.second + 1 - v_s[i_s].first);
                um_solid1 = l_v_w[i_w_s];
                len_weak_region = um_solid2.first - um_solid1.first + k;
                paths1 = extractSemiWeakPaths(c, s, all_pids, um_solid1, um_solid2, l_v_w, i_w_s);
            }
            if (!paths1.complete.empty()) {
                const std::pair<int, int> p_align = selectBest(c, paths1.complete, s.c_str() + um_solid1.first, len_weak_region, MODE_NW);
                const Path& best = paths1.complete[p_align.first];
                addAmbiguity(best, s_corrected.length());
                s_corrected += pathToString(c, best); q_corrected += best.qual;
                res.addRange(um_solid1.first - v_s[i_s].first, um_solid2.first - v_s[i_s].first + k);
            } else if (!paths1.partial.empty()) {
                const std::pair<int, int> align = selectBest(c, paths1.partial, s.c_str() + um_solid1.first, len_weak_region, MODE_SHW, opt.weak_region_len_factor);
                if (align.first == -1) addUncorrected(um_solid1.first, len_weak_region, q_min);
                else {
                    const Path& best = paths1.partial[align.first];
                    addAmbiguity(best, s_corrected.length());
                    s_corrected += pathToString(c, best) + s.substr(um_solid1.first + align.second + 1, len_weak_region - align.second - 1);
                    q_corrected += best.qual;
                    q_corrected += lrc ? q.substr(um_solid1.first + align.second + 1, len_weak_region - align.second - 1) : std::string(len_weak_region - align.second - 1, q_min); // :684-685
                    res.addRange(um_solid1.first - v_s[i_s].first, um_solid1.first + align.second + 1 - v_s[i_s].first);
                }
            } else if (!s_corrected.empty()) addUncorrected(um_solid1.first, len_weak_region, q_min);
            else setUncorrected(v_s[i_s].first, len_weak_region, q_min);
        } else {
            const std::pair<int, int> p_align = selectBest(c, paths1.complete, s.c_str() + um_solid1.first, len_weak_region, MODE_NW);
            const Path& best = paths1.complete[p_align.first];
            addAmbiguity(best, 0);
            s_corrected = pathToString(c, best); q_corrected = best.qual;
            res.addRange(0, len_weak_region);
        }
        fixAmbiguity(c, s_corrected, q_corrected, s_start, res.old_seq_len, v_ambiguity); // :716
        if (res.pos.size() == res.old_seq_len) { // :718-725 (G20)
            bool same = s_corrected.length() >= k && s.length() >= k;
            for (size_t i = 0; same && i < k; ++i) same = bifrostCode(s[s.length() - k + i]) == bifrostCode(s_corrected[s_corrected.length() - k + i]);
            if (same) res.is_corrected = true;
        }
        if (!res.is_corrected) { // :727-747
            const AlignResult a = c.align(s_start, um_solid2.first - v_s[i_s].first + k, s_corrected.c_str(), s_corrected.length(), -1, MODE_SHW, false);
            if (a.editDistance >= 0) {
                size_t end_location = static_cast<size_t>(static_cast<int64_t>(a.endLocations[0])); // -1 wraps, as in the reference
                for (size_t j = 1; j < a.endLocations.size(); ++j) if (static_cast<size_t>(static_cast<int64_t>(a.endLocations[j])) > end_location) end_location = static_cast<size_t>(static_cast<int64_t>(a.endLocations[j]));
                s_corrected = s_corrected.substr(0, end_location + 1);
                q_corrected = q_corrected.substr(0, end_location + 1);
            }
        }
        res.seq = s_corrected; res.qual = q_corrected;
        return res;
    };

    if (v_um_solid[0].first != 0) { // head region (:776-797)
        // pass 2 leaves a stretch alone when pass 1 already gave every base of it the maximum quality (:779)
        if (!lrc || q_fw.empty() || !hasMinQual(s_fw, q_fw, 0, v_um_solid[0].first + k, q_max)) {
            const size_t i_solid_rev = v_um_solid_rev.size() - 1;
            size_t i_weak_rev = v_um_weak_rev.size();
            while (i_weak_rev > 0 && v_um_weak_rev[i_weak_rev - 1].first > v_um_solid_rev[i_solid_rev].first) --i_weak_rev;
            ResultCorrection bw = correct(s_bw, q_fw, v_um_solid_rev, v_um_weak_rev, i_solid_rev, i_weak_rev, nullptr); // q_fw, not q_bw: as written (:787, G17)
            bw.reverseComplement();
            corrected_s += bw.seq.substr(0, bw.seq.length() - k);
            corrected_q += bw.qual.substr(0, bw.qual.length() - k);
        } else {
            corrected_s += s_fw.substr(0, v_um_solid[0].first);
            corrected_q += lrc ? q_fw.substr(0, v_um_solid[0].first) : std::string(v_um_solid[0].first, q_min);
        }
    }
    while (i_solid < v_um_solid.size() - 1) { // :799-938
        while (i_weak < v_um_weak.size() && v_um_weak[i_weak].first < v_um_solid[i_solid].first) ++i_weak;
        if (v_um_solid[i_solid].first != (v_um_solid[i_solid + 1].first - 1)) {
            bool isUncorrected = false;
            const UM& start_um = v_um_solid[i_solid].second; const UM& end_um = v_um_solid[i_solid + 1].second;
            bool sameUnitig = (start_um.unitig == end_um.unitig) && (start_um.strand == end_um.strand);
            if (lrc && !q_fw.empty() && hasMinQual(s_fw, q_fw, v_um_solid[i_solid].first, v_um_solid[i_solid + 1].first + k, q_max)) isUncorrected = true; // :808
            else if (sameUnitig && !g.isShortCycle(start_um.unitig)) {
                const size_t min_pos = std::min(start_um.dist, end_um.dist), max_pos = std::max(start_um.dist, end_um.dist);
                const size_t len_query_km = v_um_solid[i_solid + 1].first - v_um_solid[i_solid].first;
                const size_t len_unitig_km = max_pos - min_pos;
                const size_t min_len_unitig_km = getMinMaxLength(len_unitig_km, opt.weak_region_len_factor).first;
                const size_t max_len_unitig_km = getMinMaxLength(len_unitig_km, opt.weak_region_len_factor).second;
                sameUnitig = sameUnitig && ((start_um.strand && (start_um.dist < end_um.dist)) || (!start_um.strand && (start_um.dist > end_um.dist)));
                sameUnitig = sameUnitig && (len_query_km >= min_len_unitig_km) && (len_query_km <= max_len_unitig_km);
                if (sameUnitig) {
                    UM um_sub = start_um; um_sub.dist = static_cast<uint32_t>(min_pos); um_sub.len = static_cast<uint32_t>(len_unitig_km + 1);
                    const std::string s_um_sub = g.mapped(um_sub);
                    if (cnt) cnt->n_path_base += s_um_sub.size();
                    corrected_s += s_fw.substr(prev_pos, v_um_solid[i_solid].first - prev_pos) + s_um_sub.substr(0, s_um_sub.length() - k);
                    if (lrc) { // :847-853
                        const size_t buff = (s_um_sub.length() >= 2 * k) ? k : (s_um_sub.length() - k);
                        corrected_q += q_fw.substr(prev_pos, v_um_solid[i_solid].first - prev_pos + buff);
                        if ((s_um_sub.length() - buff - k) > 0) corrected_q += std::string(s_um_sub.length() - buff - k, q_max);
                    } else corrected_q += std::string((v_um_solid[i_solid].first - prev_pos) + (s_um_sub.length() - k), q_max);
                } else isUncorrected = true;
            } else if (v_um_solid[i_solid + 1].first >= (v_um_solid[i_solid].first + k)) {
                const ResultCorrection fw = correct(s_fw, q_fw, v_um_solid, v_um_weak, i_solid, i_weak, nullptr);
                if (fw.is_corrected) {
                    const size_t l_solid = v_um_solid[i_solid].first - prev_pos;
                    const std::string sub_s = s_fw.substr(prev_pos, l_solid) + fw.seq, sub_q = (lrc ? q_fw.substr(prev_pos, l_solid) : std::string(l_solid, q_max)) + fw.qual;
                    corrected_s += sub_s.substr(0, sub_s.length() - k); corrected_q += sub_q.substr(0, sub_q.length() - k);
                } else {
                    const size_t i_solid_bw = v_um_solid_rev.size() - i_solid - 2;
                    size_t i_weak_bw = v_um_weak_rev.size() - i_weak;
                    while (i_weak_bw > 0 && v_um_weak_rev[i_weak_bw - 1].first > v_um_solid_rev[i_solid_bw].first) --i_weak_bw;
                    ResultCorrection bw = correct(s_bw, q_bw, v_um_solid_rev, v_um_weak_rev, i_solid_bw, i_weak_bw, &fw);
                    bw.reverseComplement();
                    if (bw.is_corrected) {
                        const size_t l_solid = (s_bw.length() - v_um_solid_rev[i_solid_bw + 1].first - k) - prev_pos;
                        const std::string sub_s = s_fw.substr(prev_pos, l_solid) + bw.seq, sub_q = (lrc ? q_fw.substr(prev_pos, l_solid) : std::string(l_solid, q_max)) + bw.qual;
                        corrected_s += sub_s.substr(0, sub_s.length() - k); corrected_q += sub_q.substr(0, sub_q.length() - k);
                    } else {
                        std::string l_ref = s_fw.substr(v_um_solid[i_solid].first, v_um_solid[i_solid + 1].first - v_um_solid[i_solid].first + k);
                        std::pair<std::string, std::string> cons = generateConsensus(c, &fw, &bw, l_ref, opt.weak_region_len_factor);
                        if (cons.first.length() == 0) { cons.first = l_ref; cons.second = lrc ? q_fw.substr(v_um_solid[i_solid].first, v_um_solid[i_solid + 1].first - v_um_solid[i_solid].first + k) : (std::string(k, q_max) + std::string(v_um_solid[i_solid + 1].first - v_um_solid[i_solid].first, q_min)); }
                        const size_t l_solid = v_um_solid[i_solid].first - prev_pos;
                        const std::string sub_s = s_fw.substr(prev_pos, l_solid) + cons.first, sub_q = (lrc ? q_fw.substr(prev_pos, l_solid) : std::string(l_solid, q_max)) + cons.second;
                        corrected_s += sub_s.substr(0, sub_s.length() - k); corrected_q += sub_q.substr(0, sub_q.length() - k);
                    }
                }
            } else isUncorrected = true;
            if (isUncorrected) {
                corrected_s += s_fw.substr(prev_pos, v_um_solid[i_solid + 1].first - prev_pos);
                if (lrc) corrected_q += q_fw.substr(prev_pos, v_um_solid[i_solid + 1].first - prev_pos); // :924
                else {
                    corrected_q += std::string(v_um_solid[i_solid].first - prev_pos, q_max);
                    if (v_um_solid[i_solid + 1].first < (v_um_solid[i_solid].first + k)) corrected_q += std::string(v_um_solid[i_solid + 1].first - v_um_solid[i_solid].first, q_max);
                    else corrected_q += std::string(k, q_max) + std::string(v_um_solid[i_solid + 1].first - v_um_solid[i_solid].first - k, q_min);
                }
            }
            prev_pos = v_um_solid[i_solid + 1].first;
        }
        ++i_solid;
    }
    if (v_um_solid[v_um_solid.size() - 1].first < s_fw.length() - k && // tail region (:940-950)
        (!lrc || q_fw.empty() || !hasMinQual(s_fw, q_fw, v_um_solid[v_um_solid.size() - 1].first, s_fw.length(), q_max))) {
        while (i_weak < v_um_weak.size() && v_um_weak[i_weak].first < v_um_solid[i_solid].first) ++i_weak;
        const ResultCorrection fw = correct(s_fw, q_fw, v_um_solid, v_um_weak, i_solid, i_weak, nullptr);
        const size_t l_solid = v_um_solid[i_solid].first - prev_pos;
        corrected_s += s_fw.substr(prev_pos, l_solid) + fw.seq;
        corrected_q += (lrc ? q_fw.substr(prev_pos, l_solid) : std::string(l_solid, q_max)) + fw.qual;
    } else {
        corrected_s += s_fw.substr(prev_pos);
        corrected_q += lrc ? q_fw.substr(prev_pos) : (std::string(v_um_solid[i_solid].first - prev_pos + k, q_max) + std::string(s_fw.length() - v_um_solid[i_solid].first - k, q_min));
    }
    return std::make_pair(corrected_s, corrected_q);
}

std::pair<std::string, std::string> correctRead(const Graph& g, const Opt& opt, std::string seq, std::string qual, Counters* cnt) { // src/Ratatosk.cpp:808-864
    for (size_t i = 0; i < seq.size(); ++i) seq[i] = static_cast<char>(std::toupper(static_cast<unsigned char>(seq[i])));
    for (size_t i = 0; i < qual.size(); ++i) { if (qual[i] < static_cast<char>(33)) qual[i] = static_cast<char>(33); if (qual[i] > static_cast<char>(33 + opt.max_qual)) qual[i] = static_cast<char>(33 + opt.max_qual); }
    // nb_correction_rounds == 1: min_score = 0, factors unchanged (src/Ratatosk.cpp:843-856, G8)
    const std::pair<std::vector<Anchor>, std::vector<Anchor> > seeds = getSeeds(g, opt, seq, cnt);
    return correctSequence(g, opt, seq, qual, seeds.first, seeds.second, cnt);
}

} // namespace orc
