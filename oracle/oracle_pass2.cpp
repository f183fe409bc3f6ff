// ORACLE (test infrastructure, never shipped, never on the product path): the parts of the reference's second correction pass
// (`Ratatosk correct -2`, long_read_correct == true) that pass 1 does not have.
//   phasing()                 src/Graph.cpp:869-1097   per-read pre-filter: pass-1 corrections on unitigs whose colours (= ids of
//                             pass-1 corrected reads) no other, distant unitig of the read shares are reverted to the raw read
//   TinyBloomFilter<size_t>   src/TinyBloomFilter.hpp:8-294 (constructor :13-44, insert :119-137, cardinalities :160-214)
//   per-read body             src/Ratatosk.cpp:808-838; writer with trimming src/Ratatosk.cpp:510-563
// The switches inside getSeeds / extractSemiWeakPaths / exploreSubGraphLong / correctSequence live in oracle_seeds.cpp and
// oracle_correct.cpp (Opt::long_read_correct).
//
// [A9] wyhash. TinyBloomFilter hashes with wyhash(&elem, 8, seed, _wyp) from the header Bifrost bundles (absent here, version
// unknown). Restated: wyhash "final version 3" for an 8-byte key -- a = r4(p) << 32 | r4(p + 4), b = r4(p + 4) << 32 | r4(p),
// result = wymix(secret[1] ^ 8, wymix(a ^ secret[1], b ^ (seed ^ secret[0]))), default secret. PARITY UNPINNED: another wyhash
// revision changes which bits a colour sets, hence (rarely) which unitigs pass the 0.85 similarity test of phasing().
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>

#include "oracle_correct.hpp"
#include "oracle_myers.hpp"

namespace orc {

namespace {

inline uint64_t wymix(uint64_t a, uint64_t b) { const unsigned __int128 r = static_cast<unsigned __int128>(a) * b; return static_cast<uint64_t>(r) ^ static_cast<uint64_t>(r >> 64); }
const uint64_t kWyp[4] = {0xa0761d6478bd642full, 0xe7037ed1a0b428dbull, 0x8ebc6af09c88c6e3ull, 0x589965cc75374cc3ull};

inline size_t rndup(size_t v) { // Bifrost rndup: next power of two (v itself when it is one)
    --v; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32; return v + 1;
}

struct TinyBloom { // src/TinyBloomFilter.hpp
    std::vector<uint64_t> table; uint64_t bits, nb_h;
    static double fpp(size_t bits_per_elem, size_t h) { const double h_d = static_cast<double>(h), b_d = static_cast<double>(bits_per_elem); return std::pow(1.0 - std::exp(-(h_d / b_d)), h_d); } // :283-288
    TinyBloom(size_t nb_elem, size_t bits_per_elem) : bits(0), nb_h(0) { // :13-44
        if (nb_elem != 0 && bits_per_elem != 0) {
            nb_h = static_cast<uint64_t>(static_cast<double>(bits_per_elem) * std::log(2.0));
            nb_h += static_cast<uint64_t>(fpp(bits_per_elem, nb_h) >= fpp(bits_per_elem, nb_h + 1));
            nb_h &= 0xffULL;
            bits = static_cast<uint64_t>(std::max(rndup(bits_per_elem * nb_elem), static_cast<size_t>(64)));
            table.assign(bits / 64, 0);
        }
    }
    void insert(uint64_t elem) { // :119-137: every probed bit ends up set
        if (table.empty()) return;
        const uint64_t mask = bits - 1, hv_2 = wyhash8(elem, 1610612741ull);
        uint64_t hv_1 = wyhash8(elem, 49157ull);
        for (uint64_t i = 0; i != nb_h; ++i) { table[(hv_1 & mask) >> 6] |= 1ULL << (hv_1 & 0x3FULL); hv_1 += hv_2; }
    }
    size_t cardinality_bits() const { size_t c = 0; for (size_t i = 0; i < table.size(); ++i) c += static_cast<size_t>(__builtin_popcountll(table[i])); return c; }
    size_t and_cardinality_bits(const TinyBloom& o) const { size_t c = 0; const size_t n = std::min(table.size(), o.table.size()); for (size_t i = 0; i < n; ++i) c += static_cast<size_t>(__builtin_popcountll(table[i] & o.table[i])); return c; }
};

inline char getQual(const double score, const size_t qv_min, const size_t qv_max) {
    const char phred_base_std = static_cast<char>(33), phred_scale_std = static_cast<char>(qv_max);
    const double qv_score = std::min(score, 1.0) * static_cast<double>(phred_scale_std - qv_min);
    return static_cast<char>(qv_score + phred_base_std + qv_min);
}

} // namespace

uint64_t wyhash8(uint64_t key, uint64_t seed) {
    seed ^= kWyp[0];
    const uint64_t lo = key & 0xFFFFFFFFull, hi = key >> 32; // little-endian: r4(p) = low half, r4(p + 4) = high half
    const uint64_t a = (lo << 32) | hi, b = (hi << 32) | lo;
    return wymix(kWyp[1] ^ 8ull, wymix(a ^ kWyp[1], b ^ seed));
}

std::pair<std::string, std::string> phasing(const Graph& g, const Opt& opt, const std::string& s_raw, const std::string& s_corr, const std::string& q_corr) {
    const size_t k = static_cast<size_t>(g.k);
    const char q_min = getQual(0.0, 0, opt.max_qual), q_max = getQual(1.0, 0, opt.max_qual);
    const double t_bits_sim = 0.85;
    const size_t max_limit_nb_pids = 1000, nb_bits_elem_tbf = 14;
    size_t max_nb_pids = 0;
    std::string s_out, q_out;
    std::vector<char> pos2rm(s_corr.length() + k + 2, 0);
    std::vector<Anchor> v_um;
    { // the read mapped unitig by unitig (:889-917): KmerHashIterator only stops on A/C/G/T windows; a mapped stretch is stepped over
        const size_t len = s_corr.length();
        size_t bad = 0; // non-ACGT characters inside the current window
        for (size_t i = 0; i < len && i + 1 < k; ++i) bad += isDNA(s_corr[i]) ? 0 : 1;
        for (size_t p = 0; p + k <= len;) {
            // window validity by direct scan (reads hold few non-ACGT characters)
            bool ok = true; for (size_t i = 0; i < k && ok; ++i) ok = isDNA(s_corr[p + i]);
            if (!ok) { ++p; continue; }
            const UM um = g.findUnitig(s_corr.c_str(), p, len);
            if (!um.isEmpty()) {
                const size_t card = g.cardinality(um.unitig);
                if (!g.isBranching(um.unitig) && card <= max_limit_nb_pids) { v_um.push_back(Anchor(p, um)); max_nb_pids = std::max(max_nb_pids, card); }
                p += um.len; // ++it after it += um.len - 1
            } else ++p;
        }
        (void)bad;
    }
    {
        const size_t n = v_um.size();
        std::vector<TinyBloom> v_tbf; v_tbf.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            TinyBloom tbf(max_nb_pids, nb_bits_elem_tbf);
            const IdSet ids = g.allIds(v_um[i].second.unitig);
            for (size_t j = 0; j < ids.size(); ++j) tbf.insert(static_cast<uint64_t>(ids[j]));
            v_tbf.push_back(tbf);
        }
        std::vector<char> valid(n, 0), invalid(n, 0);
        for (size_t i = 0; i < n; ++i) {
            if (valid[i]) continue;
            bool found = false, compatible = false;
            const size_t pos_i = v_um[i].first, nb_bits_i = v_tbf[i].cardinality_bits();
            for (size_t j = 0; j < n; ++j) {
                if (invalid[j]) continue;
                const size_t pos_j = v_um[j].first, min_pos_j = (pos_j < opt.insert_sz) ? 0 : (pos_j - opt.insert_sz);
                if (pos_i < min_pos_j || pos_i > pos_j + opt.insert_sz) {
                    const size_t nb_bits_j = v_tbf[j].cardinality_bits(), nb_shared = v_tbf[i].and_cardinality_bits(v_tbf[j]);
                    compatible = true;
                    if (static_cast<double>(nb_shared) >= t_bits_sim * static_cast<double>(nb_bits_i) && static_cast<double>(nb_shared) >= t_bits_sim * static_cast<double>(nb_bits_j)) { found = true; valid[i] = 1; valid[j] = 1; break; }
                }
            }
            if (!found && compatible) {
                const size_t len_i = v_um[i].second.len;
                for (size_t j = pos_i; j < pos_i + len_i + k; ++j) { if (j >= pos2rm.size()) pos2rm.resize(j + 1, 0); pos2rm[j] = 1; }
                invalid[i] = 1;
            }
        }
    }
    auto rm = [&](size_t i) { return i < pos2rm.size() && pos2rm[i]; };
    { // corrected read against the raw one (:975-1092): query = raw, target = corrected
        const AlignResult a = myers_align(s_raw.c_str(), static_cast<int>(s_raw.length()), s_corr.c_str(), static_cast<int>(s_corr.length()), -1, MODE_NW, true, true);
        std::vector<std::pair<size_t, char> > ops; // CIGAR: runs of M (0/3), I (1, query only), D (2, target only)
        for (size_t i = 0; i < a.alignment.size(); ++i) { const char op = a.alignment[i] == 1 ? 'I' : (a.alignment[i] == 2 ? 'D' : 'M'); if (ops.empty() || ops.back().second != op) ops.push_back(std::make_pair(static_cast<size_t>(1), op)); else ++ops.back().first; }
        size_t target_pos = 0, query_pos = 0;
        std::vector<size_t> new_base_pos;
        for (size_t o = 0; o < ops.size(); ++o) {
            const size_t l = ops[o].first;
            if (ops[o].second == 'M') {
                for (size_t i = target_pos; i < target_pos + l; ++i) {
                    if (rm(i)) {
                        if (s_corr[i] == s_raw[query_pos + i - target_pos]) q_out += q_corr[i];
                        else { q_out += q_min; new_base_pos.push_back(s_out.length()); }
                        s_out += s_raw[query_pos + i - target_pos];
                    } else { s_out += s_corr[i]; q_out += q_corr[i]; }
                }
                query_pos += l; target_pos += l;
            } else if (ops[o].second == 'I') {
                if (rm(target_pos)) {
                    for (size_t i = 0, n0 = s_out.length(); i < l; ++i) new_base_pos.push_back(n0 + i);
                    s_out += s_raw.substr(query_pos, l); q_out += std::string(l, q_min);
                }
                query_pos += l;
            } else {
                for (size_t i = target_pos; i < target_pos + l; ++i) if (!rm(i)) { s_out += s_corr[i]; q_out += q_corr[i]; }
                target_pos += l;
            }
        }
        { // bases that came back from the raw read and sit on graph k-mers get the maximum quality again (:1071-1089)
            std::string s_new(s_out.length(), 'N');
            for (size_t x = 0; x < new_base_pos.size(); ++x) {
                const size_t pos = new_base_pos[x];
                const size_t pos_min = (pos < (k - 1)) ? 0 : (pos - k + 1), pos_max = ((pos + k) > s_out.length()) ? s_out.length() : (pos + k);
                s_new.replace(pos_min, pos_max - pos_min, s_out, pos_min, pos_max - pos_min);
            }
            const std::vector<Anchor> hits = searchExact(g, s_new, nullptr);
            for (size_t h = 0; h < hits.size(); ++h) for (size_t j = hits[h].first; j < hits[h].first + k; ++j) if (q_out[j] == q_min) q_out[j] = q_max;
        }
    }
    return std::make_pair(s_out, q_out);
}

// fixSNPs (src/Alignment.cpp:846-965; `-f`, applied to the corrected read before phasing(), src/Ratatosk.cpp:672,828): every character
// of the read that is not A/C/G/T is replaced by a base when exactly ONE of its bases gives the 2k-1 window around it a k-mer of the
// graph. Restated as written, including
//   * the candidate enumeration j < 4 * |v_amb| with digit p of j = (j >> 2p) & 3 (so only the first ambiguity of a window sees all
//     four bases; the p-th one sees the values of j >> 2p below 4 * |v_amb| >> 2p), stopped once two bases are candidates (:922);
//   * `upper_bound(min_pos_amb + len_buff_amb)` (:893): an ambiguous character at the position right BEHIND the window counts in the
//     number of k-mers to query and takes a digit; its substitution writes one past the window copy. [A10] that write does not reach
//     the k-mers of the window (KmerHashIterator is given len_buff_amb and stays inside it);
//   * resolved characters leave m_amb, later windows see the resolved read (out_s), unresolved ones stay ambiguous.
// KmerHashIterator visits the windows of k A/C/G/T characters [A5]; after a valid substitution every character of the window is one.
std::string fixSNPs(const Graph& g, const std::string& s) {
    const size_t k = static_cast<size_t>(g.k), limit_nb_km_cand = 64;
    std::string out_s = s;
    if (s.length() < k) return out_s;
    std::map<size_t, uint8_t> m_amb; // position -> set of bases (bit 0 A, 1 C, 2 G, 3 T; getAmbiguityRev, src/Common.hpp:389-399)
    for (size_t i = 0; i < s.length(); ++i) if (!isDNA(s[i])) m_amb.insert(std::make_pair(i, iupacIndex(s[i])));
    static const char alpha[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 0; i < s.length(); ++i) {
        if (isDNA(out_s[i])) continue;
        const size_t min_pos_amb = (i < (k - 1)) ? 0 : (i - k + 1);
        const size_t len_buff_amb = std::min(i + k, s.length()) - min_pos_amb;
        const size_t pos_amb_buff = i - min_pos_amb;
        const std::string s_sub = out_s.substr(min_pos_amb, len_buff_amb);
        std::vector<std::pair<size_t, uint8_t> > v_amb(m_amb.lower_bound(min_pos_amb), m_amb.upper_bound(min_pos_amb + len_buff_amb));
        if (v_amb.empty()) continue;
        size_t nb_km_cand = 1;
        for (size_t a = 0; a < v_amb.size(); ++a) {
            const uint8_t m = v_amb[a].second;
            nb_km_cand *= static_cast<size_t>(m & 1) + static_cast<size_t>((m >> 1) & 1) + static_cast<size_t>((m >> 2) & 1) + static_cast<size_t>((m >> 3) & 1);
            if (nb_km_cand >= limit_nb_km_cand) break;
        }
        if (nb_km_cand >= limit_nb_km_cand) continue;
        std::set<char> s_amb_cand;
        for (size_t a = 0; a < v_amb.size(); ++a) v_amb[a].first -= min_pos_amb;
        for (size_t j = 0; (j < v_amb.size() * 4) && (s_amb_cand.size() <= 1); ++j) {
            std::string l_s_sub = s_sub + '\0'; // one spare character: the write of an ambiguity right behind the window lands here [A10]
            bool valid = true;
            for (size_t pos_v_amb = 0; (pos_v_amb < v_amb.size()) && valid; ++pos_v_amb) {
                const size_t subpos_v_amb = (j >> ((pos_v_amb << 1) & 63)) & 0x3ULL; // x86-64 shift semantics; shifts >= 64 are not reached (fewer than 6 ambiguities with bases precede an invalid one)
                if ((v_amb[pos_v_amb].second >> subpos_v_amb) & 1) l_s_sub[v_amb[pos_v_amb].first] = alpha[subpos_v_amb];
                else valid = false;
            }
            if (valid && (s_amb_cand.find(l_s_sub[pos_amb_buff]) == s_amb_cand.end())) {
                for (size_t p = 0; p + k <= len_buff_amb; ++p) {
                    bool ok = true; for (size_t x = 0; x < k && ok; ++x) ok = isDNA(l_s_sub[p + x]);
                    if (!ok) continue;
                    if (!g.findUnitig(l_s_sub.c_str(), p, len_buff_amb).isEmpty()) { s_amb_cand.insert(l_s_sub[pos_amb_buff]); break; }
                }
            }
        }
        if (s_amb_cand.size() == 1) { out_s[i] = *s_amb_cand.begin(); m_amb.erase(i); }
    }
    return out_s;
}

std::pair<std::string, std::string> correctRead2(const Graph& g, const Opt& opt_in, std::string seq, std::string qual, const std::string& seq_raw_in, Counters* cnt) {
    Opt opt = opt_in; opt.long_read_correct = true;
    for (size_t i = 0; i < seq.size(); ++i) seq[i] = static_cast<char>(std::toupper(static_cast<unsigned char>(seq[i]))); // :814 (the raw read is used as read: :774-802)
    if (opt.force_unres_snp_corr) seq = fixSNPs(g, seq); // :828
    const std::pair<std::string, std::string> ph = opt.skip_phasing ? std::make_pair(seq, qual) : phasing(g, opt, seq_raw_in, seq, qual); // :832
    const std::pair<std::vector<Anchor>, std::vector<Anchor> > seeds = getSeeds(g, opt, ph.first, cnt);
    return correctSequence(g, opt, ph.first, ph.second, seeds.first, seeds.second, cnt);
}

std::vector<std::pair<std::string, std::pair<std::string, std::string> > > trimRecords(const std::string& name, const std::string& seq, const std::string& qual, size_t k, int trim) {
    std::vector<std::pair<std::string, std::pair<std::string, std::string> > > out;
    if (trim == 0) { out.push_back(std::make_pair(name, std::make_pair(seq, qual))); return out; }
    const char c_min = static_cast<char>(trim + 33);
    const int64_t l_qual = static_cast<int64_t>(qual.length()), kk = static_cast<int64_t>(k);
    int64_t start_pos = -1, len = -1, id_subread = 1;
    auto emit = [&]() { out.push_back(std::make_pair(name + "/" + std::to_string(id_subread++), std::make_pair(seq.substr(static_cast<size_t>(start_pos), static_cast<size_t>(len)), qual.substr(static_cast<size_t>(start_pos), static_cast<size_t>(len))))); };
    for (int64_t pos = 0; pos < l_qual; ++pos) {
        if (qual[static_cast<size_t>(pos)] >= c_min) { if (start_pos == -1) { start_pos = pos; len = 0; } ++len; }
        else { if (len >= kk) emit(); start_pos = -1; len = -1; }
    }
    if (len >= kk) emit();
    return out;
}

} // namespace orc
