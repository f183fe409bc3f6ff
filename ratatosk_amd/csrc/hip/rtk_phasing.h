// phasing() of the second correction pass on the device: one wavefront per read (reference: src/Graph.cpp:869-1097;
// TinyBloomFilter src/TinyBloomFilter.hpp:13-44,119-137; wyhash [A9], see oracle/oracle_pass2.cpp for the assumption).
// The pass-1 corrected read is mapped on the second-pass graph (whose colours are the ids of the pass-1 reads); every mapped,
// non-branching stretch gets a small Bloom filter of its colours; a stretch whose filter resembles (>= 85 % of the set bits, both
// ways) no other stretch further than insert_sz away is not trusted: its positions are taken from the RAW read again, by walking
// the NW alignment of the raw read against the corrected one. Bases that came back and lie on graph k-mers get q_max.
// Output per read: new sequence + quality, appended to a pool (the host re-packs them into a batch for the seed / region stages).
#ifndef RTK_PHASING_H
#define RTK_PHASING_H

#include "rtk_region.h"

struct PhaseView {
    U<const char*> raw; U<const uint64_t*> raw_off;            // uncorrected reads (same order), concatenated; [n_reads + 1] offsets
    U<char*> out_pool; U<uint64_t> out_cap; U<unsigned long long*> out_top;
    U<uint64_t*> out_off; U<uint32_t*> out_len;                // per read: sequence at out_off, quality at out_off + out_len
    U<uint32_t> tbf_nb_h;                                      // hash functions of a TinyBloomFilter with 14 bits per element (computed in double on the host)
    U<uint32_t> align_all;                                     // developer (RTK_PHASE_ALIGN_ALL=1): align every read, also those the alignment cannot change
};

RTK_DEV uint64_t rtk_wymix(uint64_t a, uint64_t b) {
#ifdef RTK_SIM
    const unsigned __int128 r = static_cast<unsigned __int128>(a) * b; return static_cast<uint64_t>(r) ^ static_cast<uint64_t>(r >> 64);
#else
    return (a * b) ^ __umul64hi(a, b);
#endif
}
RTK_DEV uint64_t rtk_wyhash8(uint64_t key, uint64_t seed) { // wyhash(&key, 8, seed, _wyp), final version 3 [A9]
    seed ^= 0xa0761d6478bd642full;
    const uint64_t lo = key & 0xFFFFFFFFull, hi = key >> 32;
    const uint64_t a = (lo << 32) | hi, b = (hi << 32) | lo;
    return rtk_wymix(0xe7037ed1a0b428dbull ^ 8ull, rtk_wymix(a ^ 0xe7037ed1a0b428dbull, b ^ seed));
}
RTK_DEV void rtk_or64(uint64_t* p, uint64_t v) {
#ifdef RTK_SIM
    *p |= v;
#else
    atomicOr(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
#endif
}

RTK_FN void rtk_phase_read(const RCtx& c_, const PhaseView& pv_, uint32_t r_) {
    const RCtx& c = *rtk_u(&c_); const PhaseView& pv = *rtk_u(&pv_); const uint32_t r = rtk_u(r_);
    RegionScratch& s = rtk_hdr(c);
    const GraphView& g = c.g;
    const uint32_t k = static_cast<uint32_t>(c.k);
    const uint64_t base = c.bv.roff[r];
    const uint32_t L = static_cast<uint32_t>(c.bv.roff[r + 1] - base);
    const char* sc_ = c.bv.seq + base; const char* qc = c.bv.qual + base; const uint64_t* hits = c.bv.hits + base;
    const uint64_t rbase = pv.raw_off[r];
    const uint32_t M = static_cast<uint32_t>(pv.raw_off[r + 1] - rbase);
    const char* raw = pv.raw + rbase;
    const char q_min = rtk_get_qual(0.0, 0, static_cast<uint64_t>(c.o.max_qual)), q_max = rtk_get_qual(1.0, 0, static_cast<uint64_t>(c.o.max_qual));
    const uint32_t nwin = L >= k ? L - k + 1 : 0;
    if (L + M + 64 > s.str_cap || (L + k + 63) / 64 + 2 > s.bm_words || (L + M + 63) / 64 + 2 > s.bm_words) { rtk_fail_ovf(s, 7); return; }
    // ---- 1. the read mapped stretch by stretch (:889-917): findUnitig = an exact hit extended while the next windows continue on the unitig [A7]
    uint64_t* run_pos = s.list[0]; uint64_t* run_ul = s.list[1]; // position; unitig << 32 | length in k-mers
    uint32_t n_runs = 0, max_nb_pids = 0;
    auto continues = [&](uint32_t p) -> bool { // window p continues the stretch of window p - 1
        if (p == 0 || p >= nwin) return false;
        const uint64_t h0 = hits[p - 1], h1 = hits[p];
        if (h0 == RTK_NO_HIT || h1 == RTK_NO_HIT) return false;
        const UMap a = rtk_unpack_hit(h0), b = rtk_unpack_hit(h1);
        return a.unitig == b.unitig && a.strand == b.strand && (a.strand ? (b.dist == a.dist + 1) : (b.dist + 1 == a.dist));
    };
    for (uint32_t p0 = 0; p0 < nwin && !rtk_failed(s); p0 += RTK_WAVE) {
        const uint32_t p = p0 + static_cast<uint32_t>(rtk_lane());
        const bool start = p < nwin && hits[p] != RTK_NO_HIT && !continues(p);
        uint64_t bal = rtk_ballot(start);
#ifdef RTK_SIM
        if (start) bal = 1ull;
#endif
        while (bal && !rtk_failed(s)) {
            const uint32_t ps = p0 + static_cast<uint32_t>(rtk_ffs(bal)) - 1u; bal &= bal - 1ull;
            uint32_t len = 1; // length of the stretch: up to the first window that does not continue it
            for (uint32_t x0 = ps + 1;; x0 += RTK_WAVE) {
                const uint32_t x = x0 + static_cast<uint32_t>(rtk_lane());
                const uint64_t brk = rtk_ballot(!continues(x));
                if (brk) { len = (x0 - ps) + static_cast<uint32_t>(rtk_ffs(brk)) - 1u; break; }
            }
            const uint32_t u = rtk_unpack_hit(rtk_ld(hits + ps)).unitig;
            const uint32_t card = rtk_ld(rtk_u(g.card) + u);
            if (!rtk_is_branching(g, u) && card <= 1000u) {
                if (n_runs >= s.list_cap) { rtk_fail_ovf(s, 8); break; }
                run_pos[n_runs] = ps; run_ul[n_runs] = (static_cast<uint64_t>(u) << 32) | len; ++n_runs;
                max_nb_pids = card > max_nb_pids ? card : max_nb_pids;
            }
        }
    }
    rtk_sync();
    if (rtk_failed(s)) return;
    // ---- 2. one TinyBloomFilter per stretch (:921-936) ----
    uint64_t* rm = s.bm[0]; // pos2rm, one bit per position of the corrected read
    bool any_rm = false;    // (wave-uniform) some stretch was not supported: only then does the walk of the alignment change anything
    const uint32_t rm_words = (L + k + 63) / 64 + 1;
    for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < rm_words; w += RTK_WAVE) rm[w] = 0;
    rtk_sync();
    if (n_runs != 0 && max_nb_pids != 0) {
        uint64_t bits = 64; while (bits < 14ull * max_nb_pids) bits <<= 1; // max(rndup(bits_per_elem * nb_elem), 64)
        const uint32_t words = static_cast<uint32_t>(bits / 64);
        const uint64_t mask = bits - 1, nb_h = pv.tbf_nb_h;
        if (static_cast<uint64_t>(n_runs) * words * 8ull > s.arena_cap) { rtk_fail_ovf(s, 3); return; }
        uint64_t* tbf = reinterpret_cast<uint64_t*>(s.arena[0].get());
        uint64_t* nbits = s.list[2]; uint64_t* state = s.list[3]; // set bits of every filter; bit 0 valid, bit 1 invalid
        for (uint64_t x = static_cast<uint64_t>(rtk_lane()); x < static_cast<uint64_t>(n_runs) * words; x += RTK_WAVE) tbf[x] = 0;
        rtk_sync();
        for (uint32_t i = 0; i < n_runs; ++i) {
            const uint32_t u = static_cast<uint32_t>(rtk_ld(run_ul + i) >> 32);
            uint64_t* t = tbf + static_cast<uint64_t>(i) * words;
            const int32_t gi = g.gid[u];
            for (int part = 0; part < 2; ++part) {
                const uint32_t* ids = part ? (gi >= 0 ? g.col + g.goff[gi] : nullptr) : g.col + g.loff[u];
                const uint32_t n = part ? (gi >= 0 ? static_cast<uint32_t>(g.goff[gi + 1] - g.goff[gi]) : 0u) : static_cast<uint32_t>(g.loff[u + 1] - g.loff[u]);
                for (uint32_t x = static_cast<uint32_t>(rtk_lane()); x < n; x += RTK_WAVE) {
                    const uint64_t id = ids[x];
                    const uint64_t hv_2 = rtk_wyhash8(id, 1610612741ull); uint64_t hv_1 = rtk_wyhash8(id, 49157ull);
                    for (uint64_t h = 0; h != nb_h; ++h) { rtk_or64(t + ((hv_1 & mask) >> 6), 1ull << (hv_1 & 0x3Full)); hv_1 += hv_2; }
                }
            }
        }
        rtk_sync();
        for (uint32_t i = 0; i < n_runs; ++i) {
            const uint64_t* t = tbf + static_cast<uint64_t>(i) * words;
            int cb = 0; for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < words; w += RTK_WAVE) cb += rtk_popc(t[w]);
            const uint64_t tot = static_cast<uint64_t>(rtk_u(rtk_wave_sum(cb)));
            nbits[i] = tot; state[i] = 0;
        }
        rtk_sync();
        // ---- 3. which stretches are supported by a distant one (:938-973) ----
        const double t_bits_sim = 0.85;
        const uint64_t insert_sz = c.o.insert_sz;
        for (uint32_t i = 0; i < n_runs; ++i) {
            if (rtk_ld(state + i) & 1ull) continue;
            bool found = false, compatible = false;
            const uint64_t pos_i = rtk_ld(run_pos + i), nb_i = rtk_ld(nbits + i);
            const uint64_t* ti = tbf + static_cast<uint64_t>(i) * words;
            for (uint32_t j = 0; j < n_runs; ++j) {
                if (rtk_ld(state + j) & 2ull) continue;
                const uint64_t pos_j = rtk_ld(run_pos + j), min_pos_j = pos_j < insert_sz ? 0 : pos_j - insert_sz;
                if (pos_i < min_pos_j || pos_i > pos_j + insert_sz) {
                    const uint64_t* tj = tbf + static_cast<uint64_t>(j) * words;
                    int sh = 0; for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < words; w += RTK_WAVE) sh += rtk_popc(ti[w] & tj[w]);
                    const uint64_t shared = static_cast<uint64_t>(rtk_u(rtk_wave_sum(sh))), nb_j = rtk_ld(nbits + j);
                    compatible = true;
                    if (static_cast<double>(shared) >= t_bits_sim * static_cast<double>(nb_i) && static_cast<double>(shared) >= t_bits_sim * static_cast<double>(nb_j)) {
                        found = true; state[i] = rtk_ld(state + i) | 1ull; rtk_sync(); state[j] = rtk_ld(state + j) | 1ull; rtk_sync();
                        break;
                    }
                }
            }
            if (!found && compatible) {
                const uint32_t len_i = static_cast<uint32_t>(rtk_ld(run_ul + i) & 0xFFFFFFFFull);
                rtk_bm_add_range(rm, static_cast<uint32_t>(pos_i), static_cast<uint32_t>(pos_i) + len_i + k); any_rm = true;
                state[i] = rtk_ld(state + i) | 2ull; rtk_sync();
            }
        }
    }
    // ---- 4. corrected read against the raw one (:975-1069): query = raw, target = corrected ----
    // With pos2rm empty every branch of the walk over the CIGAR copies the corrected read and its qualities (M :1000-1020 and D :1036-1048 append s_corr[i] / q_corr[i]
    // for every i not in pos2rm, I :1022-1034 appends nothing) and new_base_pos stays empty (s_new is all N, :1071-1089 finds no k-mer): whatever the alignment
    // is -- NW consumes the whole target -- the result is (s_corr, q_corr). The alignment is the cost of the second pass; reads without an unsupported stretch skip it.
    if (!rtk_u(any_rm) && !pv.align_all) {
        if (rtk_lane() == 0) rtk_atomic_add(c.bv.counters + RTK_CNT_PHASE_SKIPPED, 1ull);
        unsigned long long off0 = 0;
        if (rtk_lane() == 0) off0 = rtk_atomic_add(pv.out_top, 2ull * L);
        off0 = rtk_shfl(off0, 0);
        pv.out_off[r] = off0; pv.out_len[r] = L;
        if (off0 + 2ull * L > pv.out_cap) return; // the host notices and retries with a bigger pool
        rtk_wcopy(pv.out_pool + off0, sc_, L);
        rtk_wcopy(pv.out_pool + off0 + L, qc, L);
        return;
    }
    uint32_t nm = 0;
    // The walk below looks at the moves only where pos2rm has a bit (target position of the move); everywhere else it copies the corrected read. The alignment is told so:
    // the stretches of the Hirschberg recursion without such a position are not solved (MyersScratch::need_bm, rtk_myers.h)
    if (!pv.align_all) { s.my.need_bm = rm; rtk_sync(); }
    { const unsigned long long t0 = rtk_clock(); rtk_myers_path(s.my, raw, static_cast<int>(M), sc_, static_cast<int>(L), RTK_MODE_NW, true, &nm); s.cnt[9] += rtk_clock() - t0; s.cnt[3] += 1; }
    s.my.need_bm = nullptr; rtk_sync();
    nm = rtk_u(nm);
    if (rtk_failed(s)) return;
    char* out_s = s.rbuf[0]; char* out_q = s.rbuf[1];
    uint64_t* newb = s.bm[1]; // one bit per OUTPUT position: the base came from the raw read and differs from the corrected one
    const uint32_t out_words_cap = (L + M + 63) / 64 + 1;
    for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < out_words_cap; w += RTK_WAVE) newb[w] = 0;
    rtk_sync();
    uint32_t tpos0 = 0, qpos0 = 0, olen = 0;
    const uint8_t* mv = rtk_ld(&s.my.moves);
    auto rm_at = [&](uint32_t i) -> bool { return i < 64u * rm_words && ((rm[i >> 6] >> (i & 63u)) & 1ull); };
    for (uint32_t i0 = 0; i0 < nm; i0 += RTK_WAVE) {
        const uint32_t i = i0 + static_cast<uint32_t>(rtk_lane());
        const uint8_t m = i < nm ? mv[i] : 255;
        const bool isq = (m == 0 || m == 3 || m == 1), ist = (m == 0 || m == 3 || m == 2);
        const uint64_t bq = rtk_ballot(isq), bt = rtk_ballot(ist);
        const uint64_t lt = (1ull << rtk_lane()) - 1ull;
        const uint32_t qp = qpos0 + static_cast<uint32_t>(rtk_popc(bq & lt)), tp = tpos0 + static_cast<uint32_t>(rtk_popc(bt & lt));
        bool emit = false, isnew = false; char ch = 0, qu = 0;
        if (m == 0 || m == 3) { // M run: position by position (:1000-1020)
            emit = true;
            if (rm_at(tp)) { ch = raw[qp]; if (sc_[tp] == raw[qp]) qu = qc[tp]; else { qu = q_min; isnew = true; } }
            else { ch = sc_[tp]; qu = qc[tp]; }
        } else if (m == 1) { // raw read has extra characters (:1022-1034): kept only where the corrected stretch is not trusted
            if (rm_at(tp)) { emit = true; ch = raw[qp]; qu = q_min; isnew = true; }
        } else if (m == 2) { // corrected read has extra characters (:1036-1048): dropped where it is not trusted
            if (!rm_at(tp)) { emit = true; ch = sc_[tp]; qu = qc[tp]; }
        }
        const uint64_t be = rtk_ballot(emit);
        const uint32_t op = olen + static_cast<uint32_t>(rtk_popc(be & lt));
        if (emit) { out_s[op] = ch; out_q[op] = qu; if (isnew) rtk_or64(newb + (op >> 6), 1ull << (op & 63u)); }
        olen += static_cast<uint32_t>(rtk_popc(be)); qpos0 += static_cast<uint32_t>(rtk_popc(bq)); tpos0 += static_cast<uint32_t>(rtk_popc(bt));
    }
    rtk_sync();
    // ---- 5. bases that came back and sit on graph k-mers get the maximum quality again (:1071-1089) ----
    if (olen >= k) {
        const uint32_t ow = (olen + 63) / 64;
        uint64_t* cov = s.bm[2]; // position is within k - 1 of a new base: the only characters of s_new that are not 'N'
        for (uint32_t w = static_cast<uint32_t>(rtk_lane()); w < ow + 1; w += RTK_WAVE) cov[w] = 0;
        rtk_sync();
        for (uint32_t p0 = 0; p0 < olen; p0 += RTK_WAVE) {
            const uint32_t p = p0 + static_cast<uint32_t>(rtk_lane());
            bool cv = false;
            if (p < olen) { // a new base within k - 1 positions on either side: two k-bit windows (2k - 1 bits do not fit one word at k = 63)
                const uint64_t kb = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
                cv = ((rtk_bm_window(newb, ow, static_cast<int64_t>(p) - static_cast<int64_t>(k) + 1) | rtk_bm_window(newb, ow, static_cast<int64_t>(p))) & kb) != 0ull;
            }
            const uint64_t b = rtk_ballot(cv);
#ifdef RTK_SIM
            if (cv) cov[p >> 6] |= 1ull << (p & 63u);
#else
            if (rtk_lane() == 0) cov[p0 >> 6] = b;
#endif
            (void)b;
        }
        rtk_sync();
        const uint64_t kmask_bits = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
        for (uint32_t j0 = 0; j0 + k <= olen; j0 += RTK_WAVE) {
            const uint32_t j = j0 + static_cast<uint32_t>(rtk_lane());
            bool hit = false;
            if (j + k <= olen && (rtk_bm_window(cov, ow, static_cast<int64_t>(j)) & kmask_bits) == kmask_bits) {
                RtkKm code;
                if (rtk_km_from_text(reinterpret_cast<const unsigned char*>(out_s) + j, static_cast<int>(k), &code)) hit = rtk_find_km(g, code, nullptr) != RTK_NO_HIT;
            }
            if (hit) for (uint32_t x = j; x < j + k; ++x) if (out_q[x] == q_min) out_q[x] = q_max;
        }
        rtk_sync();
    }
    // ---- 6. hand the new read over ----
    unsigned long long off = 0;
    if (rtk_lane() == 0) off = rtk_atomic_add(pv.out_top, 2ull * olen);
    off = rtk_shfl(off, 0);
    pv.out_off[r] = off; pv.out_len[r] = olen;
    if (off + 2ull * olen > pv.out_cap) return; // the host notices and retries with a bigger pool
    rtk_wcopy(pv.out_pool + off, out_s, olen);
    rtk_wcopy(pv.out_pool + off + olen, out_q, olen);
}

#endif
