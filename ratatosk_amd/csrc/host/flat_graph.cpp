// Flat graph construction on the host (product code): index files -> SoA/CSR arrays that are copied to HBM as-is.
// Reference counterparts: dbg.read() (Bifrost, absent) + readGraphData (src/Graph.cpp:722-784) + the derived
// quantities the hot path asks of UnitigData (src/UnitigData.hpp:275-410) and getMaxKmerCoverage (src/Graph.cpp:825-841).
#include "flat_graph.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <cstdlib>
#include <functional>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../common/fastx.hpp"
#include "../common/kmer.hpp"
#include "../common/rtsk_io.hpp"

namespace rtk {

// n_threads of rtk_graph_load: independent slices [lo, hi) of a range on std::threads (unitigs are independent for the 2-bit packing,
// the half-k-mer pairs, the adjacency lookups; the k-mer table is filled with compare-and-swap on the key word of a slot)
static void parallel_slices(size_t n, int n_threads, const std::function<void(size_t, size_t, int)>& fn) {
    int nt = n_threads < 1 ? 1 : n_threads; if (static_cast<size_t>(nt) > n) nt = n ? static_cast<int>(n) : 1;
    if (nt == 1) { fn(0, n, 0); return; }
    std::vector<std::thread> th; std::vector<std::string> err(nt);
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { try { fn(n * t / nt, n * (t + 1) / nt, t); } catch (const std::exception& e) { err[t] = e.what(); } });
    for (size_t t = 0; t < th.size(); ++t) th[t].join();
    for (int t = 0; t < nt; ++t) if (!err[t].empty()) throw std::runtime_error(err[t]);
}

GraphView FlatGraph::view() const {
    GraphView v;
    v.k = k; v.n_unitigs = n_unitigs(); v.n_kmers = n_kmers; v.ht_slots = ht.size() / 2;
    v.useq = useq.data(); v.uoff = uoff.data(); v.adj = adj.data(); v.flags = flags.data(); v.kcov = kcov.data(); v.card = card.data();
    v.loff = loff.data(); v.gid = gid.data(); v.goff = goff.data(); v.col = col.data(); v.ht = ht.data(); v.bf = bf.data(); v.bf_mask = bf.size() - 1;
    v.bf1 = bf1.data(); v.bf1_mask = bf1.size() * 64 - 1;
    v.cycoff = cycoff.data(); v.cyc = reinterpret_cast<const char*>(cyc.data());
    v.hap = hap.data();
    v.amb = amb.data(); v.n_amb = amb.size() - (static_cast<uint64_t>(n_unitigs()) + 1);
    v.hx = hx.data(); v.hx_mask = hx.size() - 1; v.hxl = hxl.data();
    return v;
}

uint64_t FlatGraph::bytes() const {
    return 8 * (useq.size() + uoff.size() + loff.size() + goff.size() + ht.size() + bf.size() + bf1.size() + cycoff.size() + cyc.size() + amb.size() + hx.size() + hxl.size() + hap.size()) + 4 * (adj.size() + flags.size() + kcov.size() + card.size() + col.size() + gid.size());
}

TableSizes table_sizes(int k, uint64_t n_kmers, uint64_t n_bases) {
    TableSizes t;
    const bool wide = k > 31;
    t.h = (k - 1) / 2;
    { // half-k-mer index: not built with RTK_INEXACT_ENUM=1, above RTK_HX_MAX_GB (default 96), for two-word k-mers (second pass only, which has no 1-edit search: src/Graph.cpp:100)
        const char* e_enum = getenv("RTK_INEXACT_ENUM"); const char* e_gb = getenv("RTK_HX_MAX_GB");
        const double max_gb = e_gb ? atof(e_gb) : 96.0;
        // size before building: the distinct h-mers are at most 4^h (1.07 G for k = 31: a 3 Gb graph saturates them), the slot table a power of two
        // >= twice that, the lists two words per h-mer start + one per distinct h-mer
        const double bases = static_cast<double>(n_bases); const double all_h = std::pow(4.0, t.h); const double uniq = bases < all_h ? bases : all_h;
        double hs = 16; while (hs < 2 * uniq) hs *= 2;
        const double est_gb = (8.0 * hs + 8.0 * (2.0 * bases + uniq)) / 1e9; // (two words per h-mer start)
        t.hx = !((e_enum && e_enum[0] == '1') || est_gb > max_gb || wide);
    }
    // k-mer table, load factor <= 0.5: small graphs keep the round-2 sizes -- a power of two at load 0.25 .. 0.5; above RTK_HT_DENSE_KMERS k-mers (default 2^28: a 4 GB
    // table) the table is sized for load 0.7: the slot of a hash is floor(hash * slots / 2^64), any number of slots will do
    uint64_t slots = 16;
    while (slots < 2 * n_kmers) slots <<= 1;
    { const char* e = getenv("RTK_HT_DENSE_KMERS"); const uint64_t dense_from = e ? strtoull(e, nullptr, 10) : (1ull << 28);
      if (n_kmers >= dense_from) slots = n_kmers + n_kmers * 3 / 7 + 16; }
    t.ht_slots = slots;
    // presence pre-filter in front of the table: blocked Bloom filter, one 64-bit word per query, 2 bits per k-mer.
    // A miss (the common case for 1-edit variants) costs one 8-byte read of a structure 16x smaller than the table.
    // 4 k-mers per word: >= 16 bits per key, 1.3 % false positives.
    uint64_t bf_words = 16; { const char* e = getenv("RTK_BF_KEYS_PER_WORD"); const uint64_t kpw = e ? strtoull(e, nullptr, 10) : 4; while (bf_words * kpw < n_kmers) bf_words <<= 1; }
    t.bf_words = bf_words;
    // First level in front of it: ONE bit per k-mer in 2 MB (>= 3 bits per key, ~27 % false positives; 4 MB for graphs of 5 to 33 M k-mers), meant to stay resident in
    // the 4 MB of L2 next to each XCD so that most absent k-mers never cross the fabric (k_inexact 18.3 -> 12.8 ms per 32 Mb on the 5 Mb
    // configuration; 1 MB / 4 MB arrays measured 14.8 / 13.6 ms). Graphs too large for that get a single all-ones word (every query passes).
    { uint64_t bits = 64; const char* e0 = getenv("RTK_BF1_LOG2BITS"); const uint64_t cap_bits = 1ull << (e0 ? atoi(e0) : (3 * n_kmers > (1ull << 24) ? 25 : 24)); while (bits < 3 * n_kmers && bits < cap_bits) bits <<= 1;
      if (e0) { bits = cap_bits; }
      const char* e1 = getenv("RTK_BF1_OFF");
      t.bf1_words = (n_kmers > cap_bits || (e1 && e1[0] == '1')) ? 1 : bits / 64; } // below one bit per key the array stops paying for itself
    return t;
}

static bool km_from_string(const char* s, int k, RtkKm& out) {
    out = rtk_km_zero();
    for (int i = 0; i < k; ++i) { const int b = base2bits(s[i]); if (b < 0) return false; out = rtk_km_push(out, static_cast<uint64_t>(b), k); }
    return true;
}

void FlatGraph::load(const std::string& fasta_gz, const std::string& rtsk, int k_, int n_threads, bool defer_tables) {
    k = k_; tables_deferred = defer_tables;
    const bool load_trace = getenv("RTK_LOAD_TRACE") != nullptr; // developer: seconds per section of the load
    const auto lt0 = std::chrono::steady_clock::now(); auto lt_last = lt0;
    auto lap = [&](const char* what) { if (!load_trace) return; const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rtk load] %-34s %7.2f s (at %7.2f s)\n", what, std::chrono::duration<double>(now - lt_last).count(), std::chrono::duration<double>(now - lt0).count()); lt_last = now; };
    if (k < 3 || k > RTK_MAX_K || !(k & 1)) throw std::runtime_error("k must be odd and <= 63 (MAX_KMER_SIZE = 64 build of the reference, CMakeLists.txt:6)");
    const bool wide = k > 31; // two-word k-mers: fingerprint keys confirmed against the unitig sequence (rtk_find_kmer_wide)
    // ---- unitigs ----
    std::vector<std::string> seqs;
    {
        FastxReader fr;
        if (!fr.open(fasta_gz, n_threads > 1 ? (n_threads < 8 ? n_threads : 8) : 0)) throw std::runtime_error("cannot open graph file " + fasta_gz); // (threads: the reader's own inflate beside this thread, members side by side)
        std::string name, seq, qual;
        while (fr.next(name, seq, qual)) {
            for (size_t i = 0; i < seq.size(); ++i) seq[i] = static_cast<char>(seq[i] & 0xDF);
            if (seq.size() < static_cast<size_t>(k)) throw std::runtime_error("unitig shorter than k in " + fasta_gz);
            seqs.push_back(seq);
        }
        if (fr.failed()) throw std::runtime_error("graph file " + fasta_gz + " ends in a damaged or cut-short gzip stream");
    }
    lap("unitig FASTA read");
    const size_t n = seqs.size();
    if (n == 0) throw std::runtime_error("empty graph file " + fasta_gz);
    if (n >= 0x7FFFFFFFull) throw std::runtime_error("too many unitigs for 31-bit ids");
    uoff.assign(n + 1, 0);
    for (size_t u = 0; u < n; ++u) uoff[u + 1] = uoff[u] + seqs[u].size();
    useq.assign((uoff[n] + 31) / 32 + 2, 0);
    n_kmers = 0;
    for (size_t u = 0; u < n; ++u) n_kmers += seqs[u].size() - k + 1;
    parallel_slices(n, n_threads, [&](size_t lo, size_t hi, int) {
        for (size_t u = lo; u < hi; ++u) {
            const std::string& s = seqs[u];
            for (size_t i = 0; i < s.size(); ++i) {
                const int b = base2bits(s[i]);
                if (b < 0) throw std::runtime_error("non-ACGT character in unitig");
                const uint64_t p = uoff[u] + i;
                __atomic_fetch_or(&useq[p >> 5], static_cast<uint64_t>(b) << (2 * (p & 31)), __ATOMIC_RELAXED); // a word can straddle two unitigs
            }
        }
    });
    lap("unitigs packed to 2 bits");
    // ---- half-k-mer index: every h-mer (h = (k-1)/2) of the forward unitig sequences -> the places it starts. A graph k-mer one edit
    // away from a read window shares its first or its last h characters with the read (the edit cannot be in both), so the 1-edit search
    // looks up read h-mers here and verifies the few k-mers they belong to instead of spelling every variant of the window.
    // hx: one word per slot, CANONICAL h-mer << 34 | first (the smaller of the h-mer and its reverse complement: one look-up serves both strands; round 5);
    // hxl[first] = number of places, then TWO words per place: the h + 1 bases behind the h-mer (high half) and the
    // h + 1 bases in front of it (low half), first base in the high bits, zeros where the unitig ends; then (the unitig spells the reverse complement of the key) << 63 | unitig << 32 | (h + 1 bases in front exist) << 31 | offset.
    // A candidate k-mer is the h-mer with one of its flanks: the search verifies it from the entry alone, without reading the unitig's bounds or sequence.
    // Not built (a single empty slot; the search then spells the variants) with RTK_INEXACT_ENUM=1 or above RTK_HX_MAX_GB (default 96).
    const TableSizes tsz = table_sizes(k, n_kmers, uoff[n]);
    {
        const int h = tsz.h;
        if (!tsz.hx || defer_tables) { hx.assign(1, RTK_EMPTY_KEY); hxl.assign(1, 0); }
        else {
            // Every h-mer start of every unitig as a pair (h-mer, place), sorted -- by buckets of the leading h-mer bits, so that every step runs on
            // all threads: count per (thread, bucket), scatter to the bucket's range, sort the buckets, lay the lists out bucket by bucket (their
            // offsets from a prefix sum), claim the table slots with compare-and-swap. The lists (hxl) come out as one serial pass would write them;
            // the slot an h-mer lands on depends on who claims first, which no lookup can tell.
            if (n >= (1ull << 31)) throw std::runtime_error("half-k-mer index: more than 2^31 unitigs");
            typedef std::pair<uint64_t, uint64_t> HP;
            const uint64_t n_pairs = uoff[n] - static_cast<uint64_t>(n) * static_cast<uint64_t>(h - 1);
            const uint64_t hm = (1ull << (2 * h)) - 1ull;
            auto flank_words = [&](uint64_t place, uint64_t* fl, uint64_t* pl) { // the two words of a place (see above)
                const uint32_t u = static_cast<uint32_t>(place >> 32); const uint64_t pos = place & 0xFFFFFFFFull; const std::string& sq = seqs[u];
                const uint64_t nbf = static_cast<uint64_t>(h) + 1;
                if (pos >> 31) throw std::runtime_error("half-k-mer index: a unitig of more than 2^31 bases");
                uint64_t after = 0, before = 0; const bool a_ok = pos + h + nbf <= sq.size(), b_ok = pos >= nbf;
                if (a_ok) for (uint64_t x = 0; x < nbf; ++x) after = (after << 2) | static_cast<uint64_t>(base2bits(sq[pos + h + x]));
                if (b_ok) for (uint64_t x = 0; x < nbf; ++x) before = (before << 2) | static_cast<uint64_t>(base2bits(sq[pos - nbf + x]));
                uint64_t fwd = 0; for (int x = 0; x < h; ++x) fwd = (fwd << 2) | static_cast<uint64_t>(base2bits(sq[pos + static_cast<uint64_t>(x)]));
                const bool reversed = rtk_revcomp(fwd, h) < fwd; // the list this place is on is keyed by the reverse complement of what the unitig spells here
                *fl = (after << 32) | before; *pl = (reversed ? (1ull << 63) : 0ull) | (static_cast<uint64_t>(u) << 32) | (a_ok ? (1ull << 31) : 0ull) | pos;
            };
            int nt = n_threads < 1 ? 1 : n_threads; if (static_cast<size_t>(nt) > n) nt = static_cast<int>(n);
            int bbits = 2 * h < 12 ? 2 * h : 12; while (bbits > 0 && (n_pairs >> bbits) < 4096) --bbits; // ~4096 buckets for big graphs, fewer for small ones
            const size_t nb = static_cast<size_t>(1) << bbits; const int bshift = 2 * h - bbits;
            std::vector<std::vector<uint64_t> > cnt(static_cast<size_t>(nt), std::vector<uint64_t>(nb, 0));
            auto each_hmer = [&](size_t lo, size_t hi, const std::function<void(uint64_t, uint64_t)>& fn) {
                for (size_t u = lo; u < hi; ++u) {
                    const std::string& s = seqs[u];
                    uint64_t fw = 0;
                    for (size_t i = 0; i < s.size(); ++i) {
                        fw = ((fw << 2) | static_cast<uint64_t>(base2bits(s[i]))) & hm;
                        if (i + 1 >= static_cast<size_t>(h)) { const uint64_t rc = rtk_revcomp(fw, h); fn(fw < rc ? fw : rc, (static_cast<uint64_t>(u) << 32) | static_cast<uint64_t>(i + 1 - h)); } // keyed by the canonical h-mer
                    }
                }
            };
            parallel_slices(n, nt, [&](size_t lo, size_t hi, int t) { std::vector<uint64_t>& c = cnt[static_cast<size_t>(t)]; each_hmer(lo, hi, [&](uint64_t fw, uint64_t) { ++c[fw >> bshift]; }); });
            std::vector<uint64_t> bstart(nb + 1, 0); // bucket b = [bstart[b], bstart[b + 1]); inside it the threads' shares in thread order
            { uint64_t at = 0; for (size_t bkt = 0; bkt < nb; ++bkt) { bstart[bkt] = at; for (int t = 0; t < nt; ++t) { const uint64_t c = cnt[static_cast<size_t>(t)][bkt]; cnt[static_cast<size_t>(t)][bkt] = at; at += c; } } bstart[nb] = at; }
            std::vector<HP> pairs(n_pairs);
            parallel_slices(n, nt, [&](size_t lo, size_t hi, int t) { std::vector<uint64_t>& c = cnt[static_cast<size_t>(t)]; each_hmer(lo, hi, [&](uint64_t fw, uint64_t place) { pairs[c[fw >> bshift]++] = HP(fw, place); }); });
            std::atomic<size_t> next_b(0);
            std::vector<uint64_t> buniq(nb, 0);
            parallel_slices(static_cast<size_t>(nt), nt, [&](size_t, size_t, int) { // buckets handed out one at a time: sorted, distinct h-mers counted
                for (;;) { const size_t bkt = next_b.fetch_add(1); if (bkt >= nb) break;
                    std::sort(pairs.begin() + static_cast<std::ptrdiff_t>(bstart[bkt]), pairs.begin() + static_cast<std::ptrdiff_t>(bstart[bkt + 1]));
                    uint64_t u = 0; for (uint64_t i = bstart[bkt]; i < bstart[bkt + 1]; ++i) if (i == bstart[bkt] || pairs[i].first != pairs[i - 1].first) ++u;
                    buniq[bkt] = u; } });
            uint64_t uniq = 0; std::vector<uint64_t> lstart(nb + 1, 0); // list words of bucket b start at lstart[b]: one count word per distinct h-mer + its places
            for (size_t bkt = 0; bkt < nb; ++bkt) { lstart[bkt] = 2 * bstart[bkt] + uniq; uniq += buniq[bkt]; }
            lstart[nb] = 2 * n_pairs + uniq;
            if (2 * n_pairs + uniq >= (1ull << 34)) throw std::runtime_error("half-k-mer index: more than 2^34 list words (set RTK_INEXACT_ENUM=1)");
            uint64_t hslots = 16;
            while (hslots < 2 * uniq) hslots <<= 1;
            hx.alloc_uninitialised(hslots); // (every word of the two arrays is written below, by all threads)
            parallel_slices(static_cast<size_t>(hslots), nt, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) hx[i] = RTK_EMPTY_KEY; });
            hxl.alloc_uninitialised(2 * n_pairs + uniq + 1);
            next_b = 0;
            parallel_slices(static_cast<size_t>(nt), nt, [&](size_t, size_t, int) {
                const size_t RING = 16; uint64_t ring[16]; size_t n_pend = 0; // table words waiting for their (prefetched) slot
                auto claim = [&](uint64_t word) { uint64_t q = rtk_hash64(word >> 34) & (hslots - 1);
                    for (;;) { uint64_t expect = RTK_EMPTY_KEY; if (__atomic_load_n(&hx[q], __ATOMIC_RELAXED) == RTK_EMPTY_KEY && __atomic_compare_exchange_n(&hx[q], &expect, word, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break; q = (q + 1) & (hslots - 1); } };
                for (;;) { const size_t bkt = next_b.fetch_add(1); if (bkt >= nb) break;
                    uint64_t w = lstart[bkt];
                    for (uint64_t i = bstart[bkt]; i < bstart[bkt + 1];) {
                        uint64_t j = i; while (j < bstart[bkt + 1] && pairs[j].first == pairs[i].first) ++j;
                        const uint64_t word = (pairs[i].first << 34) | w;
                        if (n_pend >= RING) claim(ring[n_pend & (RING - 1)]);
                        ring[n_pend & (RING - 1)] = word; ++n_pend;
                        __builtin_prefetch(&hx[rtk_hash64(pairs[i].first) & (hslots - 1)], 1);
                        hxl[w++] = j - i;
                        for (uint64_t t = i; t < j; ++t) { uint64_t fl, pl; flank_words(pairs[t].second, &fl, &pl); hxl[w++] = fl; hxl[w++] = pl; }
                        i = j;
                    } }
                for (size_t j = n_pend > RING ? n_pend - RING : 0; j < n_pend; ++j) claim(ring[j & (RING - 1)]);
            });
            hxl[2 * n_pairs + uniq] = 0;
        }
    }
    lap("half-k-mer index");
    // ---- k-mer -> (unitig, offset, orientation) table + its two presence filters (sizes: table_sizes) ----
    const uint64_t slots = defer_tables ? 1 : tsz.ht_slots;
    ht.alloc_uninitialised(2 * slots);
    parallel_slices(static_cast<size_t>(slots), n_threads, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) { ht[2 * i] = RTK_EMPTY_KEY; ht[2 * i + 1] = 0; } });
    const uint64_t bf_words = defer_tables ? 1 : tsz.bf_words;
    bf.assign(bf_words, 0);
    if (tsz.bf1_words == 1 || defer_tables) bf1.assign(1, ~0ull); else bf1.assign(tsz.bf1_words, 0);
    const bool bf1_off = bf1.size() == 1;
    if (!defer_tables) parallel_slices(n, n_threads, [&](size_t lo, size_t hi, int) {
        struct Pend { uint64_t can, hh, val; };
        const size_t RING = 16; Pend ring[16]; size_t n_pend = 0;
        auto insert = [&](const Pend& pe) {
            uint64_t h = rtk_ht_slot(pe.hh, slots);
            while (true) { // claim an empty slot with compare-and-swap on its key word (another thread may be filling the table too)
                uint64_t seen_key = __atomic_load_n(&ht[2 * h], __ATOMIC_RELAXED);
                if (seen_key == RTK_EMPTY_KEY) { uint64_t expect = RTK_EMPTY_KEY; if (__atomic_compare_exchange_n(&ht[2 * h], &expect, pe.can, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break; seen_key = expect; }
                if (!wide && seen_key == pe.can) throw std::runtime_error("k-mer occurs twice in the unitig file: not a compacted de Bruijn graph for this k");
                h = rtk_ht_next(h, slots);
            }
            __atomic_fetch_or(&bf[(pe.hh >> 32) & (bf_words - 1)], (1ull << (pe.hh & 63)) | (1ull << ((pe.hh >> 6) & 63)), __ATOMIC_RELAXED);
            if (!bf1_off) { const uint64_t b1 = (pe.hh >> 12) & (bf1.size() * 64 - 1); __atomic_fetch_or(&bf1[b1 >> 6], 1ull << (b1 & 63ull), __ATOMIC_RELAXED); }
            ht[2 * h + 1] = pe.val;
        };
        for (size_t u = lo; u < hi; ++u) {
            const std::string& s = seqs[u];
            RtkKm fwk = rtk_km_zero();
            for (size_t i = 0; i < s.size(); ++i) {
                fwk = rtk_km_push(fwk, static_cast<uint64_t>(base2bits(s[i])), k);
                if (i + 1 < static_cast<size_t>(k)) continue;
                bool is_fw; uint64_t can, hh; // can: the key word of the slot
                if (!wide) { can = kmer_canonical(fwk.lo, k, &is_fw); hh = rtk_hash64(can); }
                else { const RtkKm rc = rtk_km_revcomp(fwk, k); is_fw = !rtk_km_less(rc, fwk); const RtkKm c2 = is_fw ? fwk : rc; hh = rtk_km_hash(c2); can = rtk_km_fingerprint(c2); }
                // the slot and the filter words are asked for now and written a few k-mers later: a table of GBs is a cache miss per access
                Pend& pe = ring[n_pend & (RING - 1)];
                if (n_pend >= RING) insert(pe);
                pe.can = can; pe.hh = hh; pe.val = (static_cast<uint64_t>(u) << 32) | (static_cast<uint64_t>(i + 1 - k) << 1) | (is_fw ? 1ull : 0ull);
                __builtin_prefetch(&ht[2 * rtk_ht_slot(hh, slots)], 1); __builtin_prefetch(&bf[(hh >> 32) & (bf_words - 1)], 1);
                if (!bf1_off) __builtin_prefetch(&bf1[((hh >> 12) & (bf1.size() * 64 - 1)) >> 6], 1);
                ++n_pend;
            }
        }
        for (size_t j = n_pend > RING ? n_pend - RING : 0; j < n_pend; ++j) insert(ring[j & (RING - 1)]);
    });
    const GraphView gv0 = [&]() { GraphView v; v.k = k; v.ht = ht.data(); v.ht_slots = slots; v.bf = bf.data(); v.bf_mask = bf_words - 1; v.bf1 = bf1.data(); v.bf1_mask = bf1.size() * 64 - 1; v.useq = useq.data(); v.uoff = uoff.data(); return v; }();
    if (wide && !defer_tables) { // fingerprints cannot tell a repeated k-mer while the table is filled: every k-mer has to find ITSELF afterwards
        parallel_slices(n, n_threads, [&](size_t lo, size_t hi, int) {
            for (size_t u = lo; u < hi; ++u) {
                const std::string& s = seqs[u];
                RtkKm fwk = rtk_km_zero();
                for (size_t i = 0; i < s.size(); ++i) {
                    fwk = rtk_km_push(fwk, static_cast<uint64_t>(base2bits(s[i])), k);
                    if (i + 1 < static_cast<size_t>(k)) continue;
                    const uint64_t hit = rtk_find_kmer_wide(gv0, fwk, nullptr);
                    if (hit != rtk_pack_hit(static_cast<uint32_t>(u), static_cast<uint32_t>(i + 1 - k), 1u)) throw std::runtime_error("k-mer occurs twice in the unitig file: not a compacted de Bruijn graph for this k");
                }
            }
        });
    }
    lap("k-mer table + filters");
    // Deferred tables: the .rtsk records name their unitig by its head k-mer, which is a unitig extremity (anything else is an error below): a table of
    // the 2n extremity k-mers (hash of the canonical k-mer -> unitig, confirmed against the sequence) answers that without the table of all k-mers.
    std::vector<uint32_t> ext; uint64_t ext_mask = 0;
    auto ext_kmer = [&](uint32_t u, int end) { RtkKm x; const std::string& s = seqs[u]; km_from_string(end ? s.c_str() + s.size() - k : s.c_str(), k, x); const RtkKm rc = rtk_km_revcomp(x, k); return rtk_km_less(rc, x) ? rc : x; };
    if (defer_tables) {
        uint64_t es = 16; while (es < 4 * n) es <<= 1;
        ext.assign(es, RTK_NONE32); ext_mask = es - 1;
        parallel_slices(n, n_threads, [&](size_t lo, size_t hi, int) {
            for (size_t u = lo; u < hi; ++u) for (int end = 0; end < 2; ++end) {
                if (end == 1 && seqs[u].size() == static_cast<size_t>(k)) break; // one k-mer: both ends are it
                uint64_t q = rtk_km_hash(ext_kmer(static_cast<uint32_t>(u), end)) & ext_mask;
                for (;;) { uint32_t expect = RTK_NONE32; if (__atomic_load_n(&ext[q], __ATOMIC_RELAXED) == RTK_NONE32 && __atomic_compare_exchange_n(&ext[q], &expect, static_cast<uint32_t>(u) << 1 | static_cast<uint32_t>(end), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break; q = (q + 1) & ext_mask; }
            }
        });
        lap("extremity k-mers (deferred tables)");
    }
    auto find_head = [&](const RtkKm& code) -> uint64_t { // packed hit of a head k-mer, RTK_NO_HIT if it is no unitig extremity
        const RtkKm rc = rtk_km_revcomp(code, k); const RtkKm can = rtk_km_less(rc, code) ? rc : code;
        for (uint64_t q = rtk_km_hash(can) & ext_mask;; q = (q + 1) & ext_mask) {
            const uint32_t e = ext[q]; if (e == RTK_NONE32) return RTK_NO_HIT;
            if (rtk_km_eq(ext_kmer(e >> 1, static_cast<int>(e & 1u)), can)) { const uint32_t u = e >> 1; return rtk_pack_hit(u, (e & 1u) ? static_cast<uint32_t>(seqs[u].size()) - k : 0u, 1u); }
        }
    };
    // ---- unitig data (.rtsk) ----
    flags.assign(n, 0); kcov.assign(n, 0); card.assign(n, 0); gid.assign(n, -1); loff.assign(n + 1, 0);
    std::vector<std::vector<uint32_t> > locals(n);
    std::vector<std::vector<uint32_t> > globals;
    std::map<std::vector<uint32_t>, int32_t> gdedup; // identical global sets share one id (reference: src/Graph.cpp:748-771)
    std::vector<char> seen(n, 0);
    std::vector<std::vector<uint32_t> > ambs(n); // SNP annotation ids of each unitig (pos<<4 | IUPAC index)
    std::vector<std::vector<uint32_t> > haps(n); // haplotype ids
    std::vector<std::string> cycles(n); // compact cycles of short-cycle unitigs (UnitigData.hpp:307-327): NUL-terminated strings, concatenated
    {
        // the file in one piece, cut into records by their lengths (no id stream is decoded for that), the records decoded on all threads; what depends on
        // the ORDER of the records -- the numbering of the distinct global sets -- is done afterwards, in file order
        struct Mapped { const char* p = nullptr; size_t n = 0; int fd = -1; ~Mapped() { if (p && n) munmap(const_cast<char*>(p), n); if (fd >= 0) ::close(fd); } } mf; // (mapped, not read: no second copy of a 10 GB file)
        {
            mf.fd = ::open(rtsk.c_str(), O_RDONLY);
            if (mf.fd < 0) throw std::runtime_error("cannot open unitig data file " + rtsk);
            struct stat st; if (fstat(mf.fd, &st) != 0) throw std::runtime_error("cannot stat unitig data file " + rtsk);
            mf.n = static_cast<size_t>(st.st_size);
            if (mf.n) { void* q = mmap(nullptr, mf.n, PROT_READ, MAP_PRIVATE, mf.fd, 0); if (q == MAP_FAILED) { mf.n = 0; throw std::runtime_error("cannot map unitig data file " + rtsk); } mf.p = static_cast<const char*>(q); madvise(q, mf.n, MADV_WILLNEED); }
        }
        const char* file = mf.p; const size_t file_size = mf.n;
        std::vector<size_t> rec_at;
        for (size_t at = 0; at < file_size;) { const size_t len = rtsk_record_bytes(reinterpret_cast<const unsigned char*>(file) + at, file_size - at); rec_at.push_back(at); at += len; }
        rec_at.push_back(file_size);
        const size_t n_rec = rec_at.size() - 1;
        lap("unitig data (.rtsk) in memory");
        std::vector<uint32_t> rec_u(n_rec, RTK_NONE32); std::vector<std::vector<uint32_t> > rec_global(n_rec);
        parallel_slices(n_rec, n_threads, [&](size_t lo_r, size_t hi_r, int) {
            RtskRecord r;
            MemStreamBuf mb; std::istream in(&mb); // (one stream per thread: constructing a std::istream takes a reference on the global locale, a contended atomic with 128 threads doing it per record)
            for (size_t i = lo_r; i < hi_r; ++i) {
                mb.reset(file + rec_at[i], rec_at[i + 1] - rec_at[i]); in.clear();
                if (!rtsk_read_record(in, r)) throw std::runtime_error("rtsk: empty record");
                const std::string head = disk_kmer_to_string(r.head, k);
                RtkKm code;
                if (!km_from_string(head.c_str(), k, code)) throw std::runtime_error(".rtsk: bad head k-mer");
                const uint64_t hit = defer_tables ? find_head(code) : rtk_find_km(gv0, code, nullptr);
                if (hit == RTK_NO_HIT) throw std::runtime_error(defer_tables ? ".rtsk: head k-mer is not a unitig extremity of the graph (reference aborts too, src/Graph.cpp:773-780)" : ".rtsk: head k-mer not found in the graph (reference aborts too, src/Graph.cpp:773-780)");
                const UMap um = rtk_unpack_hit(hit);
                const uint32_t nk = static_cast<uint32_t>(seqs[um.unitig].size()) - k + 1;
                if (!(um.dist == 0 || um.dist == nk - 1)) throw std::runtime_error(".rtsk: head k-mer is not a unitig extremity");
                const uint32_t u = um.unitig;
                if (__atomic_exchange_n(&seen[u], static_cast<char>(1), __ATOMIC_RELAXED)) throw std::runtime_error(".rtsk: two records for one unitig");
                rec_u[i] = u;
                uint32_t f = static_cast<uint32_t>(r.shared & 0xFFull);
                if (r.shared & 0x100ull) f |= RTK_F_SHORT_CYCLE;
                cycles[u] = r.cycles;
                haps[u].swap(r.hap_ids);
                if (r.kmcov >> 63) f |= RTK_F_BRANCHING;
                if (!r.ambiguity_ids.empty()) { f |= RTK_F_AMBIGUITY; ambs[u].swap(r.ambiguity_ids); }
                flags[u] = f;
                const uint64_t cov = (r.kmcov & 0x7fffffffull) + ((r.kmcov >> 31) & 0x7fffffffull); // phased + unphased (UnitigData.hpp:371-384)
                kcov[u] = static_cast<uint32_t>(std::round(static_cast<double>(cov) / static_cast<double>(nk)));
                locals[u].swap(r.local_ids);
                rec_global[i].swap(r.global_ids);
            }
        });
        for (size_t i = 0; i < n_rec; ++i) { // identical global sets share one id, numbered in the order of their first record
            const uint32_t u = rec_u[i];
            if (!rec_global[i].empty()) {
                std::map<std::vector<uint32_t>, int32_t>::iterator it = gdedup.find(rec_global[i]);
                if (it == gdedup.end()) { it = gdedup.insert(std::make_pair(rec_global[i], static_cast<int32_t>(globals.size()))).first; globals.push_back(rec_global[i]); }
                gid[u] = it->second;
            }
            card[u] = static_cast<uint32_t>(locals[u].size() + (gid[u] >= 0 ? globals[gid[u]].size() : 0));
        }
    }
    for (size_t u = 0; u < n; ++u) if (!seen[u]) throw std::runtime_error(".rtsk: unitig without a data record");
    n_global = globals.size();
    for (size_t u = 0; u < n; ++u) loff[u + 1] = loff[u] + locals[u].size();
    goff.assign(globals.size() + 1, 0);
    goff[0] = loff[n];
    for (size_t g = 0; g < globals.size(); ++g) goff[g + 1] = goff[g] + globals[g].size();
    col.assign(goff[globals.size()] + 1, 0);
    parallel_slices(n, n_threads, [&](size_t lo_u, size_t hi_u, int) { for (size_t u = lo_u; u < hi_u; ++u) std::copy(locals[u].begin(), locals[u].end(), col.begin() + loff[u]); });
    for (size_t g = 0; g < globals.size(); ++g) std::copy(globals[g].begin(), globals[g].end(), col.begin() + goff[g]);
    lap("unitig data (.rtsk) read");
    // ---- SNP annotations: get_ambiguity_char() order = (position, IUPAC character) (UnitigData.hpp:557-574) ----
    {
        amb.assign(n + 1, 0);
        static const char codes[17] = ".ACMGRSVTWYHKDBN"; // src/Common.hpp:260
        for (size_t u = 0; u < n; ++u) {
            std::vector<uint32_t>& a = ambs[u];
            std::sort(a.begin(), a.end(), [](uint32_t x, uint32_t y) { return (x >> 4) != (y >> 4) ? (x >> 4) < (y >> 4) : codes[x & 15] < codes[y & 15]; });
            amb[u + 1] = amb[u] + a.size();
        }
        for (size_t u = 0; u < n; ++u) for (size_t i = 0; i < ambs[u].size(); ++i) amb.push_back(ambs[u][i]);
    }
    // ---- haplotype ids ----
    hap.assign(n + 1, 0);
    for (size_t u = 0; u < n; ++u) hap[u + 1] = hap[u] + haps[u].size();
    for (size_t u = 0; u < n; ++u) for (size_t i = 0; i < haps[u].size(); ++i) hap.push_back(haps[u][i]);
    // ---- compact cycles ----
    cycoff.assign(n + 1, 0);
    for (size_t u = 0; u < n; ++u) cycoff[u + 1] = cycoff[u] + cycles[u].size();
    cyc.assign(cycoff[n] / 8 + 2, 0);
    for (size_t u = 0; u < n; ++u) if (!cycles[u].empty()) memcpy(reinterpret_cast<char*>(cyc.data()) + cycoff[u], cycles[u].data(), cycles[u].size());
    lap("annotations, haplotypes, cycles");
    // ---- adjacency ([A3]: neighbours of the unitig end in walk direction, A,C,G,T) ----
    adj.assign(n * 8, RTK_NONE32);
    if (!defer_tables) parallel_slices(n, n_threads, [&](size_t lo_u, size_t hi_u, int) {
    for (size_t u = lo_u; u < hi_u; ++u) {
        const std::string& s = seqs[u];
        RtkKm tail, head;
        km_from_string(s.c_str() + s.size() - k, k, tail);
        km_from_string(s.c_str(), k, head);
        const RtkKm ends[2] = { tail, rtk_km_revcomp(head, k) };
        for (int d = 0; d < 2; ++d) for (uint64_t b = 0; b < 4; ++b) {
            const RtkKm y = rtk_km_push(ends[d], b, k);
            const uint64_t hit = rtk_find_km(gv0, y, nullptr);
            if (hit == RTK_NO_HIT) continue;
            const UMap f = rtk_unpack_hit(hit);
            const uint32_t nk = static_cast<uint32_t>(seqs[f.unitig].size()) - k + 1;
            // find(km, extremities_only=true): the k-mer has to open its unitig in walk direction
            if (!((f.strand && f.dist == 0) || (!f.strand && f.dist == nk - 1))) continue;
            adj[u * 8 + d * 4 + b] = (f.unitig << 1) | f.strand;
        }
    }
    });
    lap("adjacency");
    // ---- getMaxKmerCoverage(dbg, 0.001) ----
    {
        std::vector<uint32_t> v(kcov);
        std::sort(v.begin(), v.end(), [](uint32_t a, uint32_t b) { return a > b; });
        max_km_cov_top = v[static_cast<size_t>(static_cast<double>(v.size()) * 0.001)];
    }
}

} // namespace rtk
