// ORACLE (test infrastructure). See oracle_graph.hpp.
#include "oracle_graph.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>

namespace orc {

namespace {

inline int code(char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; } }

inline uint64_t rc_code(uint64_t x, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; ++i) { r = (r << 2) | (3 - (x & 3)); x >>= 2; }
    return r;
}

struct Rd { // little-endian cursor over the .rtsk bytes
    const std::vector<unsigned char>& b; size_t p;
    explicit Rd(const std::vector<unsigned char>& b_) : b(b_), p(0) {}
    bool eof() const { return p >= b.size(); }
    uint64_t u64() { if (p + 8 > b.size()) throw std::runtime_error("oracle: truncated .rtsk"); uint64_t v; memcpy(&v, &b[p], 8); p += 8; return v; }
    uint32_t u32at(size_t q) const { if (q + 4 > b.size()) throw std::runtime_error("oracle: truncated roaring"); uint32_t v; memcpy(&v, &b[q], 4); return v; }
    uint32_t u16at(size_t q) const { if (q + 2 > b.size()) throw std::runtime_error("oracle: truncated roaring"); return b[q] | (b[q + 1] << 8); }
};

// PairID stream (reference: src/PairID.cpp:1137-1215). Roaring portable layout per the public RoaringFormatSpec.
void read_pairid(Rd& r, IdSet& out) {
    out.clear();
    const uint64_t w = r.u64();
    switch (w & 7ULL) {
        case 1: { uint64_t bits = w >> 3; for (int i = 0; i < 61; ++i) if ((bits >> i) & 1ULL) out.push_back(static_cast<uint32_t>(i)); break; }
        case 2: out.push_back(static_cast<uint32_t>(w >> 3)); break;
        case 3: {
            const size_t n = static_cast<uint32_t>(w >> 3), base = r.p;
            if (base + n > r.b.size()) throw std::runtime_error("oracle: truncated roaring payload");
            size_t q = base;
            const uint32_t cookie = r.u32at(q); q += 4;
            uint32_t nc; bool runs = false; size_t run_bm = 0;
            if ((cookie & 0xFFFF) == 12347) { runs = true; nc = (cookie >> 16) + 1; run_bm = q; q += (nc + 7) / 8; }
            else if (cookie == 12346) { nc = r.u32at(q); q += 4; }
            else throw std::runtime_error("oracle: bad roaring cookie");
            std::vector<std::pair<uint32_t, uint32_t> > kc(nc);
            for (uint32_t i = 0; i < nc; ++i) { kc[i].first = r.u16at(q); kc[i].second = r.u16at(q + 2) + 1; q += 4; }
            if (!runs || nc >= 4) q += 4ULL * nc;
            for (uint32_t i = 0; i < nc; ++i) {
                const uint32_t hi = kc[i].first << 16;
                if (runs && ((r.b[run_bm + i / 8] >> (i % 8)) & 1)) {
                    const uint32_t nr = r.u16at(q); q += 2;
                    for (uint32_t j = 0; j < nr; ++j) { const uint32_t s = r.u16at(q), l = r.u16at(q + 2); q += 4; for (uint32_t v = s; v <= s + l; ++v) out.push_back(hi | v); }
                } else if (kc[i].second <= 4096) {
                    for (uint32_t j = 0; j < kc[i].second; ++j) { out.push_back(hi | r.u16at(q)); q += 2; }
                } else {
                    for (uint32_t v = 0; v < 65536; ++v) if ((r.b.at(q + v / 8) >> (v % 8)) & 1) out.push_back(hi | v);
                    q += 8192;
                }
            }
            r.p = base + n;
            break;
        }
        case 0: { // Bifrost TinyBitmap::write payload, assumed layout [A8] (see oracle_graph.hpp): header, cardinality, offset, data
            if (w != 0) throw std::runtime_error("oracle: PairID flag 0 with payload bits in the flag word");
            { const char* allow = getenv("RTK_ALLOW_TINYBITMAP"); if (!(allow && allow[0] == '1')) throw std::runtime_error("oracle: the index holds a Bifrost TinyBitmap colour set (assumption [A8], unverified): set RTK_ALLOW_TINYBITMAP=1 to decode it under that assumption"); }
            const uint32_t header = r.u16at(r.p); const uint32_t sz = header >> 3, mode = header & 6u;
            if (sz == 0) { r.p += 2; break; }
            if (sz < 3 || sz > 4096 || r.p + 2ull * sz > r.b.size()) throw std::runtime_error("oracle: TinyBitmap block malformed [A8]");
            const uint32_t card = r.u16at(r.p + 2), hi = r.u16at(r.p + 4) << 16; const size_t d = r.p + 6;
            if (mode == 0) { for (uint32_t v = 0; v < 16u * (sz - 3); ++v) if ((r.u16at(d + 2 * (v / 16)) >> (v % 16)) & 1u) out.push_back(hi | v); if (out.size() != card) throw std::runtime_error("oracle: TinyBitmap cardinality [A8]"); }
            else if (mode == 2) { if (3 + card > sz) throw std::runtime_error("oracle: TinyBitmap list [A8]"); for (uint32_t i = 0; i < card; ++i) out.push_back(hi | r.u16at(d + 2 * i)); }
            else if (mode == 4) { if ((card & 1u) || 3 + card > sz) throw std::runtime_error("oracle: TinyBitmap runs [A8]"); for (uint32_t i = 0; i < card; i += 2) for (uint32_t v = r.u16at(d + 2 * i); v <= r.u16at(d + 2 * i + 2); ++v) out.push_back(hi | v); }
            else throw std::runtime_error("oracle: TinyBitmap mode [A8]");
            r.p += 2ull * sz;
            break;
        }
        default: throw std::runtime_error("oracle: unknown PairID flag");
    }
}

} // namespace

std::string revcomp(const std::string& s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[s.size() - 1 - i], o;
        switch (c) {
            case 'A': o = 'T'; break; case 'C': o = 'G'; break; case 'G': o = 'C'; break; case 'T': o = 'A'; break;
            case 'M': o = 'K'; break; case 'K': o = 'M'; break; case 'R': o = 'Y'; break; case 'Y': o = 'R'; break;
            case 'V': o = 'B'; break; case 'B': o = 'V'; break; case 'H': o = 'D'; break; case 'D': o = 'H'; break;
            default: o = c; // W, S, N and anything else map to themselves
        }
        r[i] = o;
    }
    return r;
}

IdSet set_union(const IdSet& a, const IdSet& b) { IdSet o; std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(o)); return o; }
IdSet set_inter(const IdSet& a, const IdSet& b) { IdSet o; std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(o)); return o; }
IdSet set_diff(const IdSet& a, const IdSet& b) { IdSet o; std::set_difference(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(o)); return o; }
size_t set_inter_card(const IdSet& a, const IdSet& b) {
    size_t i = 0, j = 0, c = 0;
    while (i < a.size() && j < b.size()) { if (a[i] < b[j]) ++i; else if (b[j] < a[i]) ++j; else { ++c; ++i; ++j; } }
    return c;
}

void Graph::load(const std::string& fasta_gz, const std::string& rtsk, int k_) {
    k = k_;
    if (k < 3 || k > 63) throw std::runtime_error("oracle: k must be in [3,63]");
    seq.clear(); info.clear(); globals.clear(); kmap.clear(); kmap_w.clear();
    gzFile f = gzopen(fasta_gz.c_str(), "rb");
    if (!f) throw std::runtime_error("oracle: cannot open " + fasta_gz);
    {
        std::vector<char> buf(1 << 22);
        std::string cur; bool in_rec = false;
        while (gzgets(f, buf.data(), static_cast<int>(buf.size()))) {
            size_t n = strlen(buf.data());
            const bool full_line = n && buf[n - 1] == '\n';
            while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
            if (n && buf[0] == '>' ) { if (in_rec) seq.push_back(cur); cur.clear(); in_rec = true; if (!full_line) { while (gzgets(f, buf.data(), static_cast<int>(buf.size())) && !strchr(buf.data(), '\n')) {} } continue; }
            cur.append(buf.data(), n);
        }
        if (in_rec) seq.push_back(cur);
    }
    gzclose(f);
    if (k <= 31) { size_t nk = 0; for (size_t u = 0; u < seq.size(); ++u) if (seq[u].size() >= static_cast<size_t>(k)) nk += seq[u].size() - k + 1; kmap.reserve(nk); }
    for (size_t u = 0; u < seq.size(); ++u) {
        std::string& s = seq[u];
        for (size_t i = 0; i < s.size(); ++i) s[i] = static_cast<char>(s[i] & 0xDF);
        if (s.size() < static_cast<size_t>(k)) throw std::runtime_error("oracle: unitig shorter than k");
        if (k > 31) { // canonical = the smaller of the k-mer and its reverse complement as strings (A < C < G < T, the order of the 2-bit codes)
            for (size_t i = 0; i < s.size(); ++i) if (code(s[i]) < 0) throw std::runtime_error("oracle: non-ACGT in unitig");
            for (size_t i = 0; i + static_cast<size_t>(k) <= s.size(); ++i) {
                const std::string fw = s.substr(i, static_cast<size_t>(k)), rc = revcomp(fw);
                const uint64_t val = (static_cast<uint64_t>(u) << 32) | (static_cast<uint64_t>(i) << 1) | (fw < rc ? 1ULL : 0ULL);
                if (!kmap_w.insert(std::make_pair(fw < rc ? fw : rc, val)).second) throw std::runtime_error("oracle: duplicate k-mer across unitigs (input is not a compacted dBG)");
            }
            continue;
        }
        uint64_t fw = 0; const uint64_t mask = (1ULL << (2 * k)) - 1;
        for (size_t i = 0; i < s.size(); ++i) {
            const int c = code(s[i]);
            if (c < 0) throw std::runtime_error("oracle: non-ACGT in unitig");
            fw = ((fw << 2) | static_cast<uint64_t>(c)) & mask;
            if (i + 1 >= static_cast<size_t>(k)) {
                const uint64_t rc = rc_code(fw, k);
                const uint64_t can = fw < rc ? fw : rc;
                const uint64_t val = (static_cast<uint64_t>(u) << 32) | (static_cast<uint64_t>(i + 1 - k) << 1) | (fw < rc ? 1ULL : 0ULL);
                if (!kmap.insert(can, val)) throw std::runtime_error("oracle: duplicate k-mer across unitigs (input is not a compacted dBG)");
            }
        }
    }
    info.assign(seq.size(), UnitigInfo());
    // .rtsk
    std::vector<unsigned char> bytes;
    {
        FILE* fp = fopen(rtsk.c_str(), "rb");
        if (!fp) throw std::runtime_error("oracle: cannot open " + rtsk);
        fseek(fp, 0, SEEK_END); const long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
        bytes.resize(static_cast<size_t>(sz));
        if (sz && fread(bytes.data(), 1, bytes.size(), fp) != bytes.size()) { fclose(fp); throw std::runtime_error("oracle: short read on " + rtsk); }
        fclose(fp);
    }
    Rd r(bytes);
    std::map<IdSet, int32_t> gdedup; // load-time dedup of identical global sets (reference: src/Graph.cpp:748-771)
    std::vector<char> seen(seq.size(), 0);
    while (!r.eof()) {
        uint64_t w[2]; w[0] = r.u64(); w[1] = r.u64();
        std::string head(k, 'A');
        for (int i = 0; i < k; ++i) head[i] = "ACGT"[(w[i / 32] >> (2 * (31 - (i % 32)))) & 3]; // [A4]
        const UM um = findKmer(head.c_str());
        // reference locates the record with find(head, extremities_only=true) and aborts if absent (src/Graph.cpp:742-780)
        if (um.isEmpty() || !(um.dist == 0 || um.dist == nkm(um.unitig) - 1)) throw std::runtime_error("oracle: .rtsk head k-mer is not a unitig extremity of the graph");
        UnitigInfo& ui = info[um.unitig];
        seen[um.unitig] = 1;
        ui.kmcov = r.u64(); ui.shared = r.u64();
        IdSet g, amb, hap;
        read_pairid(r, g); read_pairid(r, ui.local); read_pairid(r, amb); read_pairid(r, hap);
        if (!g.empty()) {
            std::map<IdSet, int32_t>::iterator it = gdedup.find(g);
            if (it == gdedup.end()) { it = gdedup.insert(std::make_pair(g, static_cast<int32_t>(globals.size()))).first; globals.push_back(g); }
            ui.global_id = it->second;
        }
        ui.has_ambiguity = !amb.empty();
        for (size_t a = 0; a < amb.size(); ++a) ui.amb.push_back(std::make_pair(amb[a] >> 4, iupacChar(static_cast<uint8_t>(amb[a] & 0xF)))); // UnitigData.hpp:448-451,565-574
        std::sort(ui.amb.begin(), ui.amb.end());
        const uint64_t ncyc = r.u64();
        if (r.p + ncyc > bytes.size()) throw std::runtime_error("oracle: truncated cycles");
        for (uint64_t a = 0; a < ncyc;) { const size_t l = strnlen(reinterpret_cast<const char*>(&bytes[r.p + a]), ncyc - a); ui.cycles.push_back(std::string(reinterpret_cast<const char*>(&bytes[r.p + a]), l)); a += l + 1; }
        r.p += ncyc;
    }
    for (size_t u = 0; u < seen.size(); ++u) if (!seen[u]) throw std::runtime_error("oracle: unitig without .rtsk record");
}

UM Graph::findKmerCode(uint64_t fw) const {
    const uint64_t rc = rc_code(fw, k);
    const uint64_t can = fw < rc ? fw : rc;
    const uint64_t* it = kmap.find(can);
    if (!it) return UM();
    const bool stored_is_can = *it & 1ULL;
    const bool query_is_can = (fw <= rc);
    return UM(static_cast<int32_t>(*it >> 32), static_cast<uint32_t>((*it & 0xFFFFFFFFULL) >> 1), 1, stored_is_can == query_is_can);
}

static const char kIupac[17] = ".ACMGRSVTWYHKDBN";
char iupacChar(uint8_t idx) { return kIupac[idx & 15]; }
uint8_t iupacIndex(char c) { const char u = static_cast<char>(c & 0xDF); for (uint8_t i = 1; i < 16; ++i) if (u == kIupac[i]) return i; return 0; }
char iupacComplement(char c) {
    const uint8_t i = iupacIndex(c);
    if (i == 0) return c;
    return kIupac[((i & 1) << 3) | ((i & 8) >> 3) | ((i & 2) << 1) | ((i & 4) >> 1)];
}

UM Graph::findUnitig(const char* s, size_t pos, size_t len) const { // [A7]
    if (pos + static_cast<size_t>(k) > len) return UM();
    UM um = findKmer(s + pos);
    if (um.isEmpty()) return um;
    const std::string& u = seq[um.unitig];
    size_t j = pos + k; uint32_t n = 1;
    if (um.strand) { size_t up = um.dist + k; while (j < len && up < u.size() && s[j] == u[up]) { ++j; ++up; ++n; } }
    else { int64_t up = static_cast<int64_t>(um.dist) - 1; while (j < len && up >= 0 && s[j] == iupacComplement(u[static_cast<size_t>(up)])) { ++j; --up; ++n; } um.dist -= (n - 1); }
    um.len = n;
    return um;
}

std::vector<std::pair<size_t, char> > Graph::ambiguityChars(const UM& um) const {
    std::vector<std::pair<size_t, char> > v;
    if (um.isEmpty()) return v;
    const std::vector<std::pair<uint32_t, char> >& a = info[um.unitig].amb;
    const size_t sz = um.len + k - 1, end = um.dist + sz;
    if (um.strand) { for (size_t i = 0; i < a.size(); ++i) if (a[i].first >= um.dist && a[i].first < end) v.push_back(std::make_pair(a[i].first - um.dist, a[i].second)); }
    else for (size_t i = a.size(); i-- > 0;) if (a[i].first >= um.dist && a[i].first < end) v.push_back(std::make_pair(sz - (a[i].first - um.dist) - 1, iupacComplement(a[i].second)));
    return v;
}

UM Graph::findKmer(const char* s) const {
    if (k > 31) {
        for (int i = 0; i < k; ++i) if (code(s[i]) < 0) return UM();
        const std::string fws(s, static_cast<size_t>(k)), rcs = revcomp(fws);
        std::unordered_map<std::string, uint64_t>::const_iterator it = kmap_w.find(fws < rcs ? fws : rcs);
        if (it == kmap_w.end()) return UM();
        const bool stored_is_can = it->second & 1ULL, query_is_can = fws < rcs;
        return UM(static_cast<int32_t>(it->second >> 32), static_cast<uint32_t>((it->second & 0xFFFFFFFFULL) >> 1), 1, stored_is_can == query_is_can);
    }
    uint64_t fw = 0;
    for (int i = 0; i < k; ++i) { const int c = code(s[i]); if (c < 0) return UM(); fw = (fw << 2) | static_cast<uint64_t>(c); }
    return findKmerCode(fw);
}

std::string Graph::mapped(const UM& um) const {
    if (um.isEmpty()) return std::string();
    const std::string sub = seq[um.unitig].substr(um.dist, um.len + k - 1);
    return um.strand ? sub : revcomp(sub);
}

void Graph::successors(const UM& um, UM out[4], char base[4], int& n) const {
    n = 0;
    if (um.isEmpty()) return;
    const std::string& s = seq[um.unitig];
    // last k-mer of the unitig in walk direction (Bifrost neighborIterator: km_tail = strand ? tail : head.twin())
    std::string tail = um.strand ? s.substr(s.size() - k) : revcomp(s.substr(0, k));
    // [A3] switch (RTK_A3_ORDER=walk|strand, like rtk_opts::a3_strand_order on the device side): on the reverse strand the neighbours come in the
    // order of the base appended in walk direction (default) or in the order of the unitig's own strand (T,G,C,A in walk direction)
    const char* const e3 = getenv("RTK_A3_ORDER"); const bool a3_strand = e3 && !strcmp(e3, "strand"); // (read per call: the tests switch between the readings inside one process)
    for (int bi = 0; bi < 4; ++bi) {
        const int b = (a3_strand && !um.strand) ? 3 - bi : bi;
        const std::string next = tail.substr(1) + "ACGT"[b];
        UM f = findKmer(next.c_str());
        if (f.isEmpty()) continue;
        // find(km, extremities_only=true): the k-mer must be the first k-mer of its unitig in walk direction
        if (!((f.strand && f.dist == 0) || (!f.strand && f.dist == nkm(f.unitig) - 1))) continue;
        out[n] = UM(f.unitig, 0, nkm(f.unitig), f.strand);
        base[n] = "ACGT"[b];
        ++n;
    }
}

int Graph::nbSuccessors(const UM& um) const { UM o[4]; char b[4]; int n; successors(um, o, b, n); return n; }

bool Graph::getSharedPids(int32_t u, bool strand, char c) const {
    int idx;
    switch (c) { case 'A': idx = 1; break; case 'C': idx = 2; break; case 'G': idx = 4; break; case 'T': idx = 8; break; default: return false; }
    return strand ? ((info[u].shared & (static_cast<uint64_t>(idx) << 4)) != 0) : ((info[u].shared & static_cast<uint64_t>(idx)) != 0);
}

double Graph::kmerCoverage(int32_t u) const {
    const uint64_t cov = (info[u].kmcov & 0x7fffffffULL) + ((info[u].kmcov >> 31) & 0x7fffffffULL);
    return std::round(static_cast<double>(cov) / static_cast<double>(nkm(u)));
}

size_t Graph::cardinality(int32_t u) const { return info[u].local.size() + (info[u].global_id >= 0 ? globals[info[u].global_id].size() : 0); }

IdSet Graph::allIds(int32_t u) const { return info[u].global_id >= 0 ? set_union(globals[info[u].global_id], info[u].local) : info[u].local; }

size_t Graph::sharedCount(int32_t u, const IdSet& b) const {
    size_t c = set_inter_card(info[u].local, b);
    if (info[u].global_id >= 0) c += set_inter_card(globals[info[u].global_id], b);
    return c;
}

size_t Graph::sharedCount(int32_t a, int32_t b) const {
    // global and local sets of one unitig are disjoint (src/SharedPairID.cpp:274-279), so summing the parts is exact
    size_t c = sharedCount(a, info[b].local);
    if (info[b].global_id >= 0) c += sharedCount(a, globals[info[b].global_id]);
    return c;
}

size_t Graph::maxKmerCoverage(double top_ratio) const {
    std::vector<double> v; v.reserve(seq.size());
    for (size_t u = 0; u < seq.size(); ++u) v.push_back(kmerCoverage(static_cast<int32_t>(u)));
    std::sort(v.begin(), v.end(), [](double a, double b) { return a > b; });
    if (v.empty()) return 0;
    return static_cast<size_t>(v[static_cast<size_t>(static_cast<double>(v.size()) * top_ratio)]);
}

} // namespace orc
