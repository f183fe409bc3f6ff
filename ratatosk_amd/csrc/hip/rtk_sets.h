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
#define RTK_LDS_SORT_CAP (RTK_LDS_SET_CAP >= 2048u ? 512u : 256u) // a power of two (the block size of the in-LDS stages of rtk_sort_pairs): key + payload words of that many pairs fit the buffer
static_assert(4u * RTK_LDS_SORT_CAP <= RTK_LDS_SET_CAP, "LDS sort block");
RTK_DEV uint64_t* rtk_lds_sort_buf() { return reinterpret_cast<uint64_t*>(rtk_lds_set_buf()); }
#endif

#ifndef RTK_SIM
// Stable least-significant-digit radix sort of n (key, payload) pairs of 32-bit words by key, one wave, 8 bits a pass (as rtk_radix_sort_u32 of rtk_colours.h,
// with a payload and the pairs in device memory): ka / pa hold the pairs and the result, kb / pb are the other buffers, the 256 counters live in the wave's
// LDS buffer. A pass is stable -- the 64 keys of a chunk find their equals by eight ballots and rank themselves among them, chunks are taken in order -- so
// pairs that come in payload order leave in (key, payload) order. For the weak hits of a long read (thousands of pairs): three passes over the pairs
// instead of the 78-91 stages of the bitonic network.
RTK_FN void rtk_radix_sort_pairs_u32(uint32_t* ka_, uint32_t* pa_, uint32_t* kb_, uint32_t* pb_, uint32_t n_, uint32_t max_key_) {
    uint32_t* const ka = rtk_gp(rtk_u(ka_)); uint32_t* const pa = rtk_gp(rtk_u(pa_)); uint32_t* const kb = rtk_gp(rtk_u(kb_)); uint32_t* const pb = rtk_gp(rtk_u(pb_)); const uint32_t n = rtk_u(n_);
    uint32_t* const bins = rtk_lds_set_buf();
    const uint32_t lane = static_cast<uint32_t>(rtk_lane());
    const uint64_t lt = (1ull << lane) - 1ull;
    int passes = 0; { uint32_t m = rtk_u(max_key_); while (m) { ++passes; m >>= 8; } if (passes == 0) passes = 1; }
    uint32_t* sk = ka; uint32_t* sp = pa; uint32_t* dk = kb; uint32_t* dp = pb;
    for (int ps = 0; ps < passes; ++ps) {
        const int sh = 8 * ps;
        for (uint32_t i = lane; i < 256u; i += RTK_WAVE) bins[i] = 0u;
        RTK_WG_SYNC();
        for (uint32_t c0 = 0; c0 < n; c0 += RTK_WAVE) { // histogram of the digit
            const uint32_t i = c0 + lane; const bool ok = i < n;
            const uint32_t d = ok ? ((sk[i] >> sh) & 0xFFu) : 0x100u;
            uint64_t eq = rtk_ballot(ok);
            for (int bt = 0; bt < 8; ++bt) { const uint64_t bb = rtk_ballot((d >> bt) & 1u); eq &= ((d >> bt) & 1u) ? bb : ~bb; }
            if (ok && (eq & lt) == 0ull) atomicAdd(&bins[d], static_cast<uint32_t>(rtk_popc(eq))); // the first lane of every group of equal digits
        }
        RTK_WG_SYNC();
        { // exclusive prefix over the 256 bins: four bins per lane
            uint32_t v[4]; uint32_t sum = 0;
            for (int x = 0; x < 4; ++x) { v[x] = bins[4u * lane + static_cast<uint32_t>(x)]; sum += v[x]; }
            int tot; uint32_t base = static_cast<uint32_t>(rtk_wave_excl_scan(static_cast<int>(sum), &tot));
            RTK_WG_SYNC();
            for (int x = 0; x < 4; ++x) { bins[4u * lane + static_cast<uint32_t>(x)] = base; base += v[x]; }
        }
        RTK_WG_SYNC();
        for (uint32_t c0 = 0; c0 < n; c0 += RTK_WAVE) { // stable scatter, chunk by chunk
            const uint32_t i = c0 + lane; const bool ok = i < n;
            const uint32_t key = ok ? sk[i] : 0u, pay = ok ? sp[i] : 0u;
            const uint32_t d = ok ? ((key >> sh) & 0xFFu) : 0x100u;
            uint64_t eq = rtk_ballot(ok);
            for (int bt = 0; bt < 8; ++bt) { const uint64_t bb = rtk_ballot((d >> bt) & 1u); eq &= ((d >> bt) & 1u) ? bb : ~bb; }
            const uint32_t before = static_cast<uint32_t>(rtk_popc(eq & lt));
            uint32_t base = 0;
            if (ok) base = bins[d];
            RTK_WG_SYNC();
            if (ok && before == 0u) bins[d] = base + static_cast<uint32_t>(rtk_popc(eq));
            if (ok) { dk[base + before] = key; dp[base + before] = pay; }
            RTK_WG_SYNC();
        }
        rtk_sync();
        { uint32_t* t_ = sk; sk = dk; dk = t_; t_ = sp; sp = dp; dp = t_; }
    }
    if (sk != ka) { for (uint32_t i = lane; i < n; i += RTK_WAVE) { ka[i] = sk[i]; pa[i] = sp[i]; } rtk_sync(); }
}
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
