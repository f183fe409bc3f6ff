// Wave-cooperative algebra on sorted, duplicate-free u32 arrays in global memory: the colour-set ("PairID")
// operations of the hot path (reference: src/PairID.cpp:388-548 and_cardinality, operators |, &, -;
// src/Common.cpp:51-112 getNumberSharedPairID). Every lane takes one element, locates it in the other set by
// binary search, and survivors are compacted with __ballot + popcount prefix. Outputs must not alias inputs.
#ifndef RTK_SETS_H
#define RTK_SETS_H

#include "rtk_wave.h"

RTK_DEV uint32_t rtk_lower_bound(const uint32_t* a, uint32_t n, uint32_t x) { // first index with a[i] >= x
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

RTK_DEV bool rtk_set_contains(const uint32_t* a, uint32_t n, uint32_t x) { const uint32_t i = rtk_lower_bound(a, n, x); return i < n && a[i] == x; }

#ifndef RTK_LDS_SET_CAP
#define RTK_LDS_SET_CAP 2048u // words of the per-wave LDS buffer; the colour selection needs >= 1792 (rtk_colours.h)
#endif
#ifndef RTK_SIM
// The set that is searched is staged in LDS when it fits: a binary search is a chain of dependent reads, and an LDS read returns
// several times sooner than one from L2. One wave per workgroup, so the buffer is private to the wave.
__device__ __forceinline__ uint32_t* rtk_lds_set_buf() { __shared__ uint32_t buf[RTK_LDS_SET_CAP]; return buf; } // ONE 8 KB buffer per wave for every user
RTK_DEV bool rtk_lds_contains(const uint32_t* lds, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lds[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && lds[lo] == x;
}
#endif

// out = { x in a : (x in b) == want_in_b }
RTK_FN uint32_t rtk_set_filter(const uint32_t* a_, uint32_t na_, const uint32_t* b_, uint32_t nb_, bool want_in_b_, uint32_t* out_) {
    const uint32_t* a = rtk_u(a_); uint32_t na = rtk_u(na_); const uint32_t* b = rtk_u(b_); uint32_t nb = rtk_u(nb_); bool want_in_b = rtk_u(want_in_b_); uint32_t* out = rtk_u(out_);
    if (na == 0) return 0;
    if (nb == 0) { // nothing to search in: all of a, or none of it
        if (want_in_b) return 0;
        for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < na; i += RTK_WAVE) out[i] = a[i];
        rtk_sync();
        return na;
    }
    uint32_t base = 0;
#ifndef RTK_SIM
    uint32_t* const lds_b = rtk_lds_set_buf();
    const bool staged = nb <= RTK_LDS_SET_CAP && na >= 16;
    if (staged) { for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < nb; i += RTK_WAVE) lds_b[i] = b[i]; RTK_WG_SYNC(); }
#endif
    for (uint32_t i0 = 0; i0 < na; i0 += RTK_WAVE) {
        const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
        uint32_t x = 0; bool keep = false;
#ifndef RTK_SIM
        if (staged) { if (i < na) { x = a[i]; keep = (rtk_lds_contains(lds_b, nb, x) == want_in_b); } } else
#endif
        if (i < na) { x = a[i]; keep = (rtk_set_contains(b, nb, x) == want_in_b); }
        const uint64_t bal = rtk_ballot(keep);
        if (keep) out[base + static_cast<uint32_t>(rtk_popc(bal & ((1ull << rtk_lane()) - 1ull)))] = x;
        base += static_cast<uint32_t>(rtk_popc(bal));
    }
    rtk_sync();
    return base;
}

RTK_DEV uint32_t rtk_set_inter(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb, uint32_t* out) { return rtk_set_filter(a, na, b, nb, true, out); }
RTK_DEV uint32_t rtk_set_diff(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb, uint32_t* out) { return rtk_set_filter(a, na, b, nb, false, out); }

// |a & b|, stops counting once `cap` is reached (callers only compare against the cap)
RTK_FN uint32_t rtk_set_inter_count(const uint32_t* a_, uint32_t na_, const uint32_t* b_, uint32_t nb_, uint32_t cap_) {
    const uint32_t* a = rtk_u(a_); uint32_t na = rtk_u(na_); const uint32_t* b = rtk_u(b_); uint32_t nb = rtk_u(nb_); uint32_t cap = rtk_u(cap_);
    if (na > nb) { const uint32_t* t = a; a = b; b = t; const uint32_t tn = na; na = nb; nb = tn; }
    if (na == 0 || cap == 0) return 0;
    uint32_t cnt = 0;
#ifndef RTK_SIM
    uint32_t* const lds_b = rtk_lds_set_buf();
    const bool staged = nb <= RTK_LDS_SET_CAP && na >= 16;
    if (staged) { for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < nb; i += RTK_WAVE) lds_b[i] = b[i]; RTK_WG_SYNC(); }
#endif
    for (uint32_t i0 = 0; i0 < na && cnt < cap; i0 += RTK_WAVE) {
        const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
#ifndef RTK_SIM
        const bool in = (i < na) && (staged ? rtk_lds_contains(lds_b, nb, a[i]) : rtk_set_contains(b, nb, a[i]));
#else
        const bool in = (i < na) && rtk_set_contains(b, nb, a[i]);
#endif
        cnt += static_cast<uint32_t>(rtk_popc(rtk_ballot(in)));
    }
    return cnt;
}

// out = a | b ; tmp holds b \ a (capacity >= nb)
RTK_FN uint32_t rtk_set_union(const uint32_t* a_, uint32_t na_, const uint32_t* b_, uint32_t nb_, uint32_t* out_, uint32_t* tmp_) {
    const uint32_t* a = rtk_u(a_); uint32_t na = rtk_u(na_); const uint32_t* b = rtk_u(b_); uint32_t nb = rtk_u(nb_); uint32_t* out = rtk_u(out_); uint32_t* tmp = rtk_u(tmp_);
    if (nb == 0 || na == 0) { // union with the empty set: a copy
        const uint32_t* src = nb == 0 ? a : b; const uint32_t n = nb == 0 ? na : nb;
        for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < n; i += RTK_WAVE) out[i] = src[i];
        rtk_sync();
        return n;
    }
    const uint32_t nd = rtk_set_diff(b, nb, a, na, tmp);
    for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < na; i += RTK_WAVE) out[i + rtk_lower_bound(tmp, nd, a[i])] = a[i];
    for (uint32_t j = static_cast<uint32_t>(rtk_lane()); j < nd; j += RTK_WAVE) out[j + rtk_lower_bound(a, na, tmp[j])] = tmp[j];
    rtk_sync();
    return na + nd;
}

// In-place bitonic sort of n (key, value) pairs by (key, value) ascending. Arrays must have room for the next power of two
// of n (padded with all-ones keys).
#ifndef RTK_SIM
// the same network run in LDS for up to RTK_LDS_SORT_CAP pairs, in the 8 KB buffer of the set searches (one LDS allocation per wave: the
// region kernel keeps 16 waves per CU)
#define RTK_LDS_SORT_CAP (RTK_LDS_SET_CAP / 4)
RTK_DEV uint64_t* rtk_lds_sort_buf() { return reinterpret_cast<uint64_t*>(rtk_lds_set_buf()); }
#endif

RTK_FN void rtk_sort_pairs(uint64_t* key_, uint64_t* val_, uint32_t n_) {
    uint64_t* key = rtk_u(key_); uint64_t* val = rtk_u(val_); uint32_t n = rtk_u(n_);
    if (n < 2) return;
    uint32_t p = 1; while (p < n) p <<= 1;
#ifndef RTK_SIM
    if (p <= RTK_LDS_SORT_CAP) {
        uint64_t* const lk = rtk_lds_sort_buf(); uint64_t* const lv = lk + RTK_LDS_SORT_CAP;
        for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < p; i += RTK_WAVE) { lk[i] = i < n ? key[i] : ~0ull; lv[i] = i < n ? val[i] : ~0ull; }
        RTK_WG_SYNC();
        for (uint32_t kk = 2; kk <= p; kk <<= 1) {
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < p; i += RTK_WAVE) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const uint64_t ki = lk[i], kl = lk[l], vi = lv[i], vl = lv[l];
                        const bool gt = (ki > kl) || (ki == kl && vi > vl);
                        const bool up = ((i & kk) == 0);
                        if (gt == up) { lk[i] = kl; lk[l] = ki; lv[i] = vl; lv[l] = vi; }
                    }
                }
                RTK_WG_SYNC();
            }
        }
        for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < n; i += RTK_WAVE) { key[i] = lk[i]; val[i] = lv[i]; }
        rtk_sync();
        return;
    }
    { // More pairs than the LDS buffer holds (the weak hits of a long read: thousands). The same network, with every run of sub-stages whose
      // partners lie inside one block of RTK_LDS_SORT_CAP pairs done in LDS: a sort of 4096 pairs makes 10 passes over memory instead of 78.
        const uint32_t B = RTK_LDS_SORT_CAP;
        uint64_t* const lk = rtk_lds_sort_buf(); uint64_t* const lv = lk + B;
        for (uint32_t i = n + static_cast<uint32_t>(rtk_lane()); i < p; i += RTK_WAVE) { key[i] = ~0ull; val[i] = ~0ull; }
        rtk_sync();
        auto in_lds = [&](uint32_t kk_lo, uint32_t kk_hi, uint32_t j_hi) { // for every block: stages kk = kk_lo .. kk_hi, sub-stages j = min(kk / 2, j_hi) .. 1
            for (uint32_t b0 = 0; b0 < p; b0 += B) {
                for (uint32_t t = static_cast<uint32_t>(rtk_lane()); t < B; t += RTK_WAVE) { lk[t] = key[b0 + t]; lv[t] = val[b0 + t]; }
                RTK_WG_SYNC();
                for (uint32_t kk = kk_lo; kk <= kk_hi; kk <<= 1) {
                    for (uint32_t j = (kk >> 1) < j_hi ? (kk >> 1) : j_hi; j > 0; j >>= 1) {
                        for (uint32_t t = static_cast<uint32_t>(rtk_lane()); t < B / 2; t += RTK_WAVE) {
                            const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;
                            const uint64_t ki = lk[i], kl = lk[l], vi = lv[i], vl = lv[l];
                            const bool gt = (ki > kl) || (ki == kl && vi > vl);
                            const bool up = (((b0 + i) & kk) == 0);
                            if (gt == up) { lk[i] = kl; lk[l] = ki; lv[i] = vl; lv[l] = vi; }
                        }
                        RTK_WG_SYNC();
                    }
                }
                for (uint32_t t = static_cast<uint32_t>(rtk_lane()); t < B; t += RTK_WAVE) { key[b0 + t] = lk[t]; val[b0 + t] = lv[t]; }
                RTK_WG_SYNC();
            }
        };
        in_lds(2, B, B / 2); // blocks sorted, alternately up and down
        for (uint32_t kk = 2 * B; kk <= p; kk <<= 1) {
            rtk_sync();
            for (uint32_t j = kk >> 1; j >= B; j >>= 1) { // partners in different blocks: through memory
                for (uint32_t t = static_cast<uint32_t>(rtk_lane()); t < p / 2; t += RTK_WAVE) {
                    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;
                    const uint64_t ki = key[i], kl = key[l], vi = val[i], vl = val[l];
                    const bool gt = (ki > kl) || (ki == kl && vi > vl);
                    const bool up = ((i & kk) == 0);
                    if (gt == up) { key[i] = kl; key[l] = ki; val[i] = vl; val[l] = vi; }
                }
                rtk_sync();
            }
            in_lds(kk, kk, B / 2);
        }
        rtk_sync();
        return;
    }
#endif
    for (uint32_t i = n + static_cast<uint32_t>(rtk_lane()); i < p; i += RTK_WAVE) { key[i] = ~0ull; val[i] = ~0ull; }
    rtk_sync();
    for (uint32_t kk = 2; kk <= p; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = static_cast<uint32_t>(rtk_lane()); i < p; i += RTK_WAVE) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t ki = key[i], kl = key[l], vi = val[i], vl = val[l];
                    const bool gt = (ki > kl) || (ki == kl && vi > vl);
                    const bool up = ((i & kk) == 0);
                    if (gt == up) { key[i] = kl; key[l] = ki; val[i] = vl; val[l] = vi; }
                }
            }
            rtk_sync();
        }
    }
}

#endif
