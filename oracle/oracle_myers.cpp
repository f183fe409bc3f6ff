// ORACLE (test infrastructure). See oracle_myers.hpp for what is restated and from where.
#include "oracle_myers.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace orc {

namespace {

struct EqTable {
    bool eq[256][256];
    EqTable() {
        memset(eq, 0, sizeof(eq));
        for (int i = 0; i < 256; ++i) eq[i][i] = true;
        // the 28 (IUPAC code, base) pairs of reference src/Common.hpp:262-274
        static const char* pairs[] = {"MA","MC","RA","RG","SC","SG","VA","VC","VG","WA","WT","YC","YT","HA","HC","HT",
                                      "KG","KT","DA","DG","DT","BC","BG","BT","NA","NC","NG","NT"};
        for (size_t i = 0; i < sizeof(pairs) / sizeof(pairs[0]); ++i) {
            const unsigned char a = static_cast<unsigned char>(pairs[i][0]), b = static_cast<unsigned char>(pairs[i][1]);
            eq[a][b] = eq[b][a] = true;
        }
    }
};

const EqTable& eqt() { static const EqTable t; return t; }

} // namespace

bool iupac_equal(unsigned char a, unsigned char b) { return eqt().eq[a][b]; }

namespace {

inline bool eqc(bool iu, unsigned char a, unsigned char b) { return iu ? iupac_equal(a, b) : (a == b); }

// One unbanded Myers pass of query (m > 0) over target (n > 0) with NW boundaries.
// col_last[i] (optional) = D[i+1][n] for every query row i; stores (optional) keep the delta vectors of every column.
struct Cols { std::vector<uint64_t> Pv, Mv, Ph, Mh; };
int nw_pass(const char* q, int m, const char* t, int n, bool iu, std::vector<int>* col_last, Cols* store) {
    const int W = (m + 63) / 64, last_bit = (m - 1) & 63;
    std::vector<uint64_t> peq(static_cast<size_t>(256) * W, 0);
    bool have[256]; memset(have, 0, sizeof(have));
    for (int j = 0; j < n; ++j) {
        const unsigned char c = static_cast<unsigned char>(t[j]);
        if (have[c]) continue;
        have[c] = true;
        uint64_t* p = &peq[static_cast<size_t>(c) * W];
        for (int i = 0; i < m; ++i) if (eqc(iu, static_cast<unsigned char>(q[i]), c)) p[i >> 6] |= 1ULL << (i & 63);
    }
    std::vector<uint64_t> Pv(W, ~0ULL), Mv(W, 0);
    if (store) { store->Pv.resize(static_cast<size_t>(W) * n); store->Mv.resize(store->Pv.size()); store->Ph.resize(store->Pv.size()); store->Mh.resize(store->Pv.size()); }
    int score = m;
    for (int j = 0; j < n; ++j) {
        const uint64_t* e = &peq[static_cast<size_t>(static_cast<unsigned char>(t[j])) * W];
        int hin = 1;
        for (int w = 0; w < W; ++w) {
            uint64_t Eq = e[w];
            const uint64_t pv = Pv[w], mv = Mv[w];
            const uint64_t Xv = Eq | mv;
            if (hin < 0) Eq |= 1ULL;
            const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv), Mh = pv & Xh;
            if (store) { store->Ph[static_cast<size_t>(j) * W + w] = Ph; store->Mh[static_cast<size_t>(j) * W + w] = Mh; }
            const int bit = (w == W - 1) ? last_bit : 63;
            const int hout = static_cast<int>((Ph >> bit) & 1ULL) - static_cast<int>((Mh >> bit) & 1ULL);
            Ph <<= 1; Mh <<= 1;
            if (hin > 0) Ph |= 1ULL; else if (hin < 0) Mh |= 1ULL;
            Pv[w] = Mh | ~(Xv | Ph); Mv[w] = Ph & Xv;
            if (store) { store->Pv[static_cast<size_t>(j) * W + w] = Pv[w]; store->Mv[static_cast<size_t>(j) * W + w] = Mv[w]; }
            hin = hout;
        }
        score += hin;
    }
    if (col_last) {
        col_last->resize(m);
        int v = n; // D[0][n]
        for (int i = 0; i < m; ++i) { v += static_cast<int>((Pv[i >> 6] >> (i & 63)) & 1ULL) - static_cast<int>((Mv[i >> 6] >> (i & 63)) & 1ULL); (*col_last)[i] = v; }
    }
    return score;
}

// Canonical NW traceback preferring up (insert) > left (delete) > diagonal (reference: src/edlib.cpp:1021-1137).
void traceback(const char* q, int m, const char* t, int n, bool iu, std::vector<unsigned char>& out) {
    Cols c;
    int cur = nw_pass(q, m, t, n, iu, nullptr, &c);
    const int W = (m + 63) / 64;
    std::vector<unsigned char> aln;
    int i = m, j = n;
    while (i > 0 && j > 0) {
        const int r = i - 1, cc = j - 1, w = r >> 6, b = r & 63;
        const size_t idx = static_cast<size_t>(cc) * W + w;
        const int vd = static_cast<int>((c.Pv[idx] >> b) & 1ULL) - static_cast<int>((c.Mv[idx] >> b) & 1ULL);
        const int hd = static_cast<int>((c.Ph[idx] >> b) & 1ULL) - static_cast<int>((c.Mh[idx] >> b) & 1ULL);
        if (vd == 1) { aln.push_back(1); --i; cur -= 1; }
        else if (hd == 1) { aln.push_back(2); --j; cur -= 1; }
        else {
            const int left = cur - hd;
            int diag;
            if (cc == 0) diag = i - 1;
            else { const size_t idl = static_cast<size_t>(cc - 1) * W + w; diag = left - (static_cast<int>((c.Pv[idl] >> b) & 1ULL) - static_cast<int>((c.Mv[idl] >> b) & 1ULL)); }
            aln.push_back(diag == cur ? 0 : 3);
            --i; --j; cur = diag;
        }
    }
    while (i > 0) { aln.push_back(1); --i; }
    while (j > 0) { aln.push_back(2); --j; }
    out.insert(out.end(), aln.rbegin(), aln.rend());
}

// obtainAlignment (reference: src/edlib.cpp:1164-1216): traceback below 1 MB of table, else the Hirschberg split of
// src/edlib.cpp:1234-1399: target halved at n/2, the first query row (ascending) whose left+right scores add up to the
// optimum, then the row -1 boundary, then the last row.
void obtain_alignment(const char* q, int m, const char* t, int n, int best, bool iu, std::vector<unsigned char>& out) {
    if (m == 0 || n == 0) { out.insert(out.end(), static_cast<size_t>(m + n), static_cast<unsigned char>(m == 0 ? 2 : 1)); return; }
    const long long W = (m + 63) / 64;
    if ((2LL * 8 + 4) * W * n + 8LL * n < 1024 * 1024) { traceback(q, m, t, n, iu, out); return; }
    const int lh = n / 2, rh = n - lh;
    if (lh == 0) { fprintf(stderr, "oracle_myers: Hirschberg with a 1-column target is undefined in the reference\n"); abort(); }
    std::vector<int> L, Rr;
    nw_pass(q, m, t, lh, iu, &L, nullptr);
    std::string rq(q, m), rt(t + lh, rh);
    std::reverse(rq.begin(), rq.end()); std::reverse(rt.begin(), rt.end());
    nw_pass(rq.c_str(), m, rt.c_str(), rh, iu, &Rr, nullptr);
    // R(i) = cost of aligning q[i..m) with the right half = Rr[m-1-i]
    int split = -2, left_score = 0, right_score = 0;
    for (int qi = 0; qi + 1 < m; ++qi) if (L[qi] + Rr[m - 2 - qi] == best) { split = qi; left_score = L[qi]; right_score = Rr[m - 2 - qi]; break; }
    if (split == -2 && lh + Rr[m - 1] == best) { split = -1; left_score = lh; right_score = Rr[m - 1]; }
    if (split == -2 && L[m - 1] + rh == best) { split = m - 1; left_score = L[m - 1]; right_score = rh; }
    if (split == -2) { fprintf(stderr, "oracle_myers: no Hirschberg split found (inconsistent best score)\n"); abort(); }
    const int ul = split + 1;
    obtain_alignment(q, ul, t, lh, left_score, iu, out);
    obtain_alignment(q + ul, m - ul, t + lh, rh, right_score, iu, out);
}

} // namespace

AlignResult myers_align(const char* query, int m, const char* target, int n, int k, AlignMode mode, bool want_path, bool use_iupac) {
    AlignResult res;
    // zero-length special cases (reference: src/edlib.cpp:161-179) -- returned before any path is built
    if (m == 0 || n == 0) {
        if (mode == MODE_NW) { res.editDistance = std::max(m, n); res.endLocations.push_back(n - 1); }
        else { res.editDistance = m; res.endLocations.push_back(-1); }
        return res;
    }
    if (mode == MODE_NW && k >= 0 && k < std::abs(n - m)) return res; // src/edlib.cpp:744-747

    const int W = (m + 63) / 64;
    const int last_bit = (m - 1) & 63;

    // query profile for every character occurring in the target
    std::vector<uint64_t> peq(static_cast<size_t>(256) * W, 0);
    bool have[256]; memset(have, 0, sizeof(have));
    for (int j = 0; j < n; ++j) {
        const unsigned char c = static_cast<unsigned char>(target[j]);
        if (have[c]) continue;
        have[c] = true;
        uint64_t* p = &peq[static_cast<size_t>(c) * W];
        for (int i = 0; i < m; ++i) if (use_iupac ? iupac_equal(static_cast<unsigned char>(query[i]), c) : (static_cast<unsigned char>(query[i]) == c)) p[i >> 6] |= 1ULL << (i & 63);
    }

    std::vector<uint64_t> Pv(W, ~0ULL), Mv(W, 0);
    std::vector<int> col_score(n);
    int score = m;
    const int top_h = (mode == MODE_HW) ? 0 : 1;
    for (int j = 0; j < n; ++j) {
        const uint64_t* eqc = &peq[static_cast<size_t>(static_cast<unsigned char>(target[j])) * W];
        int hin = top_h;
        for (int w = 0; w < W; ++w) {
            uint64_t Eq = eqc[w];
            const uint64_t pv = Pv[w], mv = Mv[w];
            const uint64_t Xv = Eq | mv;
            if (hin < 0) Eq |= 1ULL;
            const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv);
            uint64_t Mh = pv & Xh;
            const int bit = (w == W - 1) ? last_bit : 63;
            const int hout = static_cast<int>((Ph >> bit) & 1ULL) - static_cast<int>((Mh >> bit) & 1ULL);
            Ph <<= 1; Mh <<= 1;
            if (hin > 0) Ph |= 1ULL; else if (hin < 0) Mh |= 1ULL;
            Pv[w] = Mh | ~(Xv | Ph);
            Mv[w] = Ph & Xv;
            hin = hout;
        }
        score += hin;
        col_score[j] = score;
    }

    int end0;
    if (mode == MODE_NW) {
        const int d = col_score[n - 1];
        if (k >= 0 && d > k) return res;
        res.editDistance = d;
        res.endLocations.push_back(n - 1);
        end0 = n - 1;
    } else {
        int best = col_score[0];
        for (int j = 1; j < n; ++j) best = std::min(best, col_score[j]);
        const bool pseudo = (m & 63) != 0; // edlib's padded last block exposes target position -1 with score m
        if (pseudo && m < best) best = m;
        if (k >= 0 && best > k) return res;
        res.editDistance = best;
        if (pseudo && m == best) res.endLocations.push_back(-1);
        for (int j = 0; j < n; ++j) if (col_score[j] == best) res.endLocations.push_back(j);
        end0 = res.endLocations[0];
    }

    if (want_path) {
        if (mode == MODE_HW) { fprintf(stderr, "oracle_myers: HW path alignment is never requested by the reference hot path\n"); abort(); }
        // alignment of the whole query against target[0..end0] (reference: src/edlib.cpp:271-284)
        obtain_alignment(query, m, target, end0 + 1, res.editDistance, use_iupac, res.alignment);
    }
    return res;
}

std::string alignment_to_cigar(const std::vector<unsigned char>& aln) {
    static const char code[4] = {'M', 'I', 'D', 'M'};
    std::string out;
    size_t i = 0;
    while (i < aln.size()) {
        size_t j = i;
        while (j < aln.size() && code[aln[j]] == code[aln[i]]) ++j;
        out += std::to_string(j - i);
        out.push_back(code[aln[i]]);
        i = j;
    }
    return out;
}

} // namespace orc
