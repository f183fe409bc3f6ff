// fixSNPs (src/Alignment.cpp:846-965; `-f`, run on the corrected read before phasing(), src/Ratatosk.cpp:672,828): a character of the
// read that is not A/C/G/T becomes a base when exactly one of its bases gives the 2k-1 window around it a k-mer of the graph.
// One wave per read. The ambiguous characters of a read are taken in read order (a resolved one changes the windows of the next ones,
// :959-961); for one of them the candidate spellings of its window are tried in the reference's order (:922-952) and the at most k
// k-mers of a spelling are looked up one per lane.
#ifndef RTK_FIXSNPS_H
#define RTK_FIXSNPS_H

#include "rtk_ambiguity.h"

#define RTK_FIXSNPS_WIN 256 // bytes of work area per wave: the window copy (<= 2k-1 = 125 characters) + the 32 bytes rtk_km_from_text may read past it

// win: RTK_FIXSNPS_WIN bytes private to this wave
RTK_FN void rtk_fix_snps_read(const GraphView& g_, char* s_, uint32_t len_, unsigned char* win_) {
    const GraphView& g = *rtk_u(&g_); char* s = rtk_u(s_); const uint32_t len = rtk_u(len_); unsigned char* win = rtk_u(win_);
    const uint32_t k = static_cast<uint32_t>(g.k);
    if (len < k) return; // :880
    for (uint32_t i0 = 0; i0 < len; i0 += RTK_WAVE) {
        const uint32_t ii = i0 + static_cast<uint32_t>(rtk_lane());
        uint64_t todo = rtk_ballot(ii < len && !rtk_is_dna(s[ii])); // only s[i] itself changes while i is handled: the later bits stay valid
        while (todo) {
            const uint32_t i = i0 + static_cast<uint32_t>(rtk_ffs(todo) - 1); todo &= todo - 1ull;
            const uint32_t min_pos = (i < k - 1) ? 0u : (i - k + 1);
            const uint32_t wlen = ((i + k < len) ? (i + k) : len) - min_pos;
            const uint32_t pab = i - min_pos;
            // v_amb: the ambiguous characters at window offsets 0..wlen -- the one right behind the window included (:893 upper_bound) --
            // with their base sets; the number of spellings (:901-906) stops the scan at 64
            uint32_t n_amb = 0, prod = 1; uint32_t a_pos[5] = {0, 0, 0, 0, 0}, a_set[5] = {0, 0, 0, 0, 0};
            for (uint32_t c0 = 0; c0 <= wlen && prod < 64u; c0 += RTK_WAVE) {
                const uint32_t o = c0 + static_cast<uint32_t>(rtk_lane());
                const bool in = o <= wlen && (min_pos + o) < len;
                const char ch = in ? s[min_pos + o] : 'A';
                if (o < wlen) win[o] = static_cast<unsigned char>(ch);
                uint64_t am = rtk_ballot(in && !rtk_is_dna(ch));
                while (am && prod < 64u) {
                    const int l = rtk_ffs(am) - 1; am &= am - 1ull;
                    const uint32_t set = rtk_iupac_idx(static_cast<char>(rtk_shfl(static_cast<uint32_t>(static_cast<unsigned char>(ch)), l)));
                    if (n_amb < 5) { a_pos[n_amb] = c0 + static_cast<uint32_t>(l); a_set[n_amb] = set; }
                    ++n_amb; prod *= static_cast<uint32_t>(rtk_popc(static_cast<uint64_t>(set)));
                }
            }
            // a character outside the IUPAC table has no base: no spelling is valid (prod == 0). Every other ambiguity has two bases or
            // more, so fewer than 64 spellings means at most 5 of them.
            if (prod >= 64u || prod == 0u) continue;
            rtk_sync();
            uint32_t cand = 0; // s_amb_cand as a set of bases
            for (uint32_t j = 0; j < 4u * n_amb && rtk_popc(static_cast<uint64_t>(cand)) <= 1; ++j) {
                bool valid = true; uint32_t digit_i = 0;
                for (uint32_t p = 0; p < n_amb && valid; ++p) {
                    const uint32_t d = (j >> (2u * p)) & 3u;
                    if ((a_set[p] >> d) & 1u) {
                        if (a_pos[p] < wlen && rtk_lane() == 0) win[a_pos[p]] = static_cast<unsigned char>("ACGT"[d]); // (an ambiguity right behind the window takes a digit and writes past the copy)
                        if (a_pos[p] == pab) digit_i = d;
                    } else valid = false;
                }
                if (!valid || ((cand >> digit_i) & 1u)) continue; // a valid spelling rewrites every ambiguity of the window: no stale base survives an invalid one
                rtk_sync();
                bool any = false;
                for (uint32_t t0 = 0; t0 + k <= wlen && !any; t0 += RTK_WAVE) {
                    const uint32_t t = t0 + static_cast<uint32_t>(rtk_lane());
                    bool hit = false;
                    if (t + k <= wlen) { RtkKm code; if (rtk_km_from_text(win + t, static_cast<int>(k), &code)) hit = rtk_find_km(g, code, nullptr) != RTK_NO_HIT; }
                    any = rtk_ballot(hit) != 0ull;
                }
                if (any) cand |= 1u << digit_i;
                rtk_sync();
            }
            if (rtk_popc(static_cast<uint64_t>(cand)) == 1) { if (rtk_lane() == 0) s[i] = "ACGT"[rtk_ffs(static_cast<uint64_t>(cand)) - 1]; rtk_sync(); }
        }
    }
}

#endif
