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

    if (want_path) {
        const long long sz = (2LL * 8 + 4) * W * n + 8LL * n; // the reference's traceback/Hirschberg switch (src/edlib.cpp:1191-1193)
        if (sz >= 1024 * 1024 && mode == MODE_NW) {
            fprintf(stderr, "oracle_myers: problem %dx%d would take edlib's Hirschberg path, which this oracle does not restate\n", m, n);
            abort();
        }
    }

    std::vector<uint64_t> Pv(W, ~0ULL), Mv(W, 0);
    std::vector<uint64_t> sPv, sMv, sPh, sMh; // per column copies for the traceback
    if (want_path) { sPv.resize(static_cast<size_t>(W) * n); sMv.resize(sPv.size()); sPh.resize(sPv.size()); sMh.resize(sPv.size()); }
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
            if (want_path) { sPh[static_cast<size_t>(j) * W + w] = Ph; sMh[static_cast<size_t>(j) * W + w] = Mh; }
            const int bit = (w == W - 1) ? last_bit : 63;
            const int hout = static_cast<int>((Ph >> bit) & 1ULL) - static_cast<int>((Mh >> bit) & 1ULL);
            Ph <<= 1; Mh <<= 1;
            if (hin > 0) Ph |= 1ULL; else if (hin < 0) Mh |= 1ULL;
            Pv[w] = Mh | ~(Xv | Ph);
            Mv[w] = Ph & Xv;
            if (want_path) { sPv[static_cast<size_t>(j) * W + w] = Pv[w]; sMv[static_cast<size_t>(j) * W + w] = Mv[w]; }
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
        std::vector<unsigned char>& aln = res.alignment;
        int i = m, j = end0 + 1; // rows / columns still to consume
        if (j > 0) {
            const long long sz = (2LL * 8 + 4) * W * j + 8LL * j;
            if (sz >= 1024 * 1024) { fprintf(stderr, "oracle_myers: problem %dx%d would take edlib's Hirschberg path\n", m, j); abort(); }
        }
        int cur = (j > 0) ? col_score[j - 1] : m;
        while (i > 0 && j > 0) {
            const int r = i - 1, c = j - 1, w = r >> 6, b = r & 63;
            const size_t idx = static_cast<size_t>(c) * W + w;
            const int vd = static_cast<int>((sPv[idx] >> b) & 1ULL) - static_cast<int>((sMv[idx] >> b) & 1ULL);
            const int hd = static_cast<int>((sPh[idx] >> b) & 1ULL) - static_cast<int>((sMh[idx] >> b) & 1ULL);
            if (vd == 1) { aln.push_back(1); --i; cur -= 1; }
            else if (hd == 1) { aln.push_back(2); --j; cur -= 1; }
            else {
                const int left = cur - hd;
                int diag;
                if (c == 0) diag = i - 1;
                else {
                    const size_t idl = static_cast<size_t>(c - 1) * W + w;
                    diag = left - (static_cast<int>((sPv[idl] >> b) & 1ULL) - static_cast<int>((sMv[idl] >> b) & 1ULL));
                }
                aln.push_back(diag == cur ? 0 : 3);
                --i; --j; cur = diag;
            }
        }
        while (i > 0) { aln.push_back(1); --i; }
        while (j > 0) { aln.push_back(2); --j; }
        std::reverse(aln.begin(), aln.end());
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
