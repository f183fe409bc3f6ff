// rtk_build_index: minimal, deterministic producer of the two index files `Ratatosk correct -1` consumes:
//   PREFIX.index.k31.fasta.gz   unitigs of the compacted de Bruijn graph (one FASTA record per unitig)
//   PREFIX.index.k31.rtsk       per-unitig data records (format: ../common/rtsk_io.hpp)
//
// This is NOT the reference's `index` step (src/Graph.cpp:1561-3366 `addCoverage`, Bifrost `build`):
// that step is out of the hot-path scope and cannot be built here (Bifrost is
// absent). It only has to emit the reference's *file formats* with semantically equivalent content so
// the correction path has inputs:
//   * unitigs        = maximal non-branching paths over canonical k-mers seen >= --min-count times
//   * colours        = ids of the short-read pairs having >= 1 k-mer on the unitig (sorted u32)
//   * kmCov          = number of read k-mers mapped on the unitig (unphased coverage, bits 31..61)
//   * branching bit  = >1 predecessors or >1 successors            (reference: src/Graph.cpp:1997)
//   * edge bits      = neighbour shares >= min_cov_vertices colours (reference: src/Graph.cpp:1999-2017)
//   * global/local   = simplified form of the colour compaction of src/Graph.cpp:2874-2985
//   * short cycles   = restatement of detectShortCycles (src/Graph.cpp:4660-4735), so that fixRepeats has inputs
//   * SNP annotations (--snps only) = restatement of detectSNPs (src/Graph.cpp:484-720) with the breadth-first bubble walk of
//     isValidSNPcandidate (src/GraphTraversal.cpp:1057-1147); haplotype ids stay empty (no phasing input).
#include <dlfcn.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../common/fastx.hpp"
#include "../common/kmer.hpp"
#include "../common/rtsk_io.hpp"

using namespace rtk;

template <class KM> struct KTable { // open addressing: canonical k-mer -> 64-bit value
    const KM EMPTY = ~static_cast<KM>(0);
    std::vector<KM> keys; std::vector<uint64_t> vals;
    size_t n = 0, mask = 0;
    explicit KTable(size_t cap_pow2 = 1 << 20) { keys.assign(cap_pow2, EMPTY); vals.assign(cap_pow2, 0); mask = cap_pow2 - 1; }
    void grow() {
        std::vector<KM> ok; std::vector<uint64_t> ov; ok.swap(keys); ov.swap(vals);
        keys.assign(ok.size() * 2, EMPTY); vals.assign(ok.size() * 2, 0); mask = keys.size() - 1; n = 0;
        for (size_t i = 0; i < ok.size(); ++i) if (ok[i] != EMPTY) *slot(ok[i], true) = ov[i];
    }
    uint64_t* slot(KM key, bool insert) {
        if (insert && (n + 1) * 10 > keys.size() * 6) grow();
        size_t i = hash_km(key) & mask;
        while (true) {
            if (keys[i] == key) return &vals[i];
            if (keys[i] == EMPTY) { if (!insert) return nullptr; keys[i] = key; ++n; return &vals[i]; }
            i = (i + 1) & mask;
        }
    }
};

struct Unitig { std::string seq; std::vector<uint32_t> colours; uint64_t cov = 0; };

// ---------------------------------------------------------------------------------------------- --fast / --gpu: thread-parallel steps (one-word k-mers)
// They must produce what the plain path produces, byte for byte: the plain path stays the definition (and the fallback).
template <class F> static void parallel_for(size_t n, unsigned n_thr, F f) { // f(begin, end, thread)
    if (n_thr < 1) n_thr = 1;
    std::vector<std::thread> th; const size_t per = (n + n_thr - 1) / n_thr;
    for (unsigned t = 0; t < n_thr; ++t) { const size_t b = std::min(n, per * t), e = std::min(n, per * (t + 1)); if (b < e) th.emplace_back([=]() { f(b, e, t); }); }
    for (size_t t = 0; t < th.size(); ++t) th[t].join();
}

// The graph k-mers ONE SUBSTITUTION away from a k-mer, without spelling the 3k variants: such a neighbour shares the first k/2 bases or the
// last k - k/2 bases with it. The sorted CANONICAL solid k-mers are view one as they stand (the first half leads; not copied); view two holds the same
// k-mers rotated so that the last half leads, sorted; a table of the first 24 key bits sits in front of each. An oriented k-mer y is in the graph when
// its canonical form is, so the neighbours of x are the entries one substitution away from x plus the reverse complements of the entries one substitution
// away from rc(x) (offset k-1-j, complemented base): a query scans the few entries that share a half with x, then with rc(x). Round 5: 8 bytes per
// solid k-mer beside the k-mer set (26 GB at 3 Gb; the first version kept both orientations in both views, 103 GB + a sorting copy: `--snps` did not fit
// a 3 Gb run). Used by the SNP search of --fast / --gpu (the plain path probes every variant in the k-mer table).
struct NeighbourIndex {
    int k = 0, hi_n = 0, lo_n = 0; uint64_t lomask = 0;
    const uint64_t* a = nullptr; size_t n = 0; std::vector<uint64_t> b; std::vector<uint64_t> ia, ib; int shift = 0; // ia / ib: first entry of every value of the top 24 bits of the 2k-bit key
    uint64_t rot(uint64_t x) const { return ((x & lomask) << (2 * hi_n)) | (x >> (2 * lo_n)); }
    uint64_t unrot(uint64_t r) const { return ((r & ((1ULL << (2 * hi_n)) - 1ULL)) << (2 * lo_n)) | (r >> (2 * hi_n)); }
    void build(const std::vector<uint64_t>& solid, int k_, unsigned n_thr) {
        k = k_; hi_n = k / 2; lo_n = k - hi_n; lomask = (1ULL << (2 * lo_n)) - 1ULL;
        a = solid.data(); n = solid.size();
        // view two without a second copy: the rotated keys are counted by their top 12 bits per thread slice, scattered to their bucket's place, every bucket sorted by a thread
        const int bsh = 2 * k > 12 ? 2 * k - 12 : 0; const size_t nbk = static_cast<size_t>(1) << (2 * k - bsh);
        if (n_thr == 0) n_thr = 1;
        std::vector<std::vector<size_t> > cnt(n_thr, std::vector<size_t>(nbk, 0));
        parallel_for(n, n_thr, [&](size_t bb, size_t ee, unsigned t) { for (size_t i = bb; i < ee; ++i) ++cnt[t][rot(a[i]) >> bsh]; });
        std::vector<size_t> start(nbk + 1, 0);
        { size_t at = 0; for (size_t q = 0; q < nbk; ++q) { start[q] = at; for (unsigned t = 0; t < n_thr; ++t) { const size_t c = cnt[t][q]; cnt[t][q] = at; at += c; } } start[nbk] = at; }
        b.resize(n);
        parallel_for(n, n_thr, [&](size_t bb, size_t ee, unsigned t) { for (size_t i = bb; i < ee; ++i) { const uint64_t r = rot(a[i]); b[cnt[t][r >> bsh]++] = r; } });
        { std::atomic<size_t> nx(0); std::vector<std::thread> th;
          for (unsigned t = 0; t < n_thr; ++t) th.emplace_back([&]() { for (;;) { const size_t q = nx.fetch_add(1); if (q >= nbk) break; std::sort(b.begin() + start[q], b.begin() + start[q + 1]); } });
          for (size_t t = 0; t < th.size(); ++t) th[t].join(); }
        shift = 2 * k > 24 ? 2 * k - 24 : 0;
        const size_t nb = (static_cast<size_t>(1) << (2 * k - shift)) + 1;
        auto index = [&](const uint64_t* v, std::vector<uint64_t>& ix) { ix.assign(nb, 0); for (size_t i = 0; i < n; ++i) ++ix[(v[i] >> shift) + 1]; for (size_t i = 0; i + 1 < nb; ++i) ix[i + 1] += ix[i]; };
        std::thread t2([&]() { index(b.data(), ib); }); index(a, ia); t2.join();
    }
    // the canonical k-mers one substitution away from x, as (offset << 2 | base) of the ORIENTED neighbour of the caller's k-mer (flipped: x is its reverse complement)
    void scan(uint64_t x, bool flipped, uint32_t* found, int& nf) const {
        auto put = [&](int bit, uint64_t y) { int j = k - 1 - bit / 2; uint32_t base = static_cast<uint32_t>((y >> bit) & 3ULL); if (flipped) { j = k - 1 - j; base = 3u - base; } if (nf < 192) found[nf++] = (static_cast<uint32_t>(j) << 2) | base; };
        { // same first half: the differing base lies in the last lo_n bases
            const uint64_t lo_key = x & ~lomask, hi_key = x | lomask;
            size_t i = ia[lo_key >> shift]; const size_t e = ia[(hi_key >> shift) + 1];
            i = static_cast<size_t>(std::lower_bound(a + i, a + e, lo_key) - a);
            for (; i < e && a[i] <= hi_key; ++i) { const uint64_t d = a[i] ^ x; if (d == 0) continue; const uint64_t m = (d | (d >> 1)) & 0x5555555555555555ULL; if (m & (m - 1)) continue; put(__builtin_ctzll(m), a[i]); }
        }
        { // same last half: the differing base lies in the first hi_n bases
            const uint64_t r = rot(x), himask = (1ULL << (2 * hi_n)) - 1ULL; const uint64_t lo_key = r & ~himask, hi_key = r | himask;
            size_t i = ib[lo_key >> shift]; const size_t e = ib[(hi_key >> shift) + 1];
            i = static_cast<size_t>(std::lower_bound(b.begin() + i, b.begin() + e, lo_key) - b.begin());
            for (; i < e && b[i] <= hi_key; ++i) { const uint64_t y = unrot(b[i]); const uint64_t d = y ^ x; if (d == 0) continue; const uint64_t m = (d | (d >> 1)) & 0x5555555555555555ULL; if (m & (m - 1)) continue; put(__builtin_ctzll(m), y); }
        }
    }
    // calls f(offset j, substituted base) for every graph k-mer one substitution away from x, by (j, base) ascending
    template <class F> void neighbours(uint64_t x, F f) const {
        uint32_t found[192]; int nf = 0; // (j << 2 | base): at most 3 per offset, 3k <= 93 in all (a k-mer of a tandem repeat at small k has dozens: 16 slots lost some, found by tests/test_annotators.py)
        scan(x, false, found, nf); scan(kmer_revcomp(x, k), true, found, nf);
        std::sort(found, found + nf);
        for (int i = 0; i < nf; ++i) f(static_cast<int>(found[i] >> 2), static_cast<uint64_t>(found[i] & 3u));
    }
};

static const std::vector<uint64_t>& solid64(const std::vector<uint64_t>& v) { return v; }
static const std::vector<uint64_t>& solid64(const std::vector<u128>&) { static const std::vector<uint64_t> none; return none; }
// (two-word k-mers take the plain path: these overloads are never reached)
static void fast_table_fill(KTable<u128>&, const std::vector<u128>&, unsigned) {}
struct DeviceUnitigs { const char* pool = nullptr; const uint64_t* off = nullptr; const uint64_t* seeds = nullptr; uint64_t n = 0; const uint64_t* left = nullptr; uint64_t n_left = 0; }; // what rtk_index_unitigs returns (--gpu)
static bool fast_unitigs(KTable<u128>&, const std::vector<u128>&, int, unsigned, std::vector<Unitig>&, const DeviceUnitigs*) { return false; }

// every solid k-mer into the table with value 0 (slots claimed with a compare-and-swap on the key word; the table does not grow here)
static void fast_table_fill(KTable<uint64_t>& km, const std::vector<uint64_t>& solid, unsigned n_thr) {
    const uint64_t EMPTY = ~0ULL;
    uint64_t* keys = km.keys.data(); const size_t mask = km.mask;
    parallel_for(solid.size(), n_thr, [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; ++i) {
            const uint64_t key = solid[i]; size_t s = hash_km(key) & mask;
            while (true) {
                uint64_t cur = __atomic_load_n(&keys[s], __ATOMIC_RELAXED);
                if (cur == key) break;
                if (cur == EMPTY) { uint64_t exp = EMPTY; if (__atomic_compare_exchange_n(&keys[s], &exp, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED) || exp == key) break; continue; }
                s = (s + 1) & mask;
            }
        }
    });
    km.n = solid.size();
}

// Unitigs by walking every maximal chain of mutually unique links from its ends, on all threads. The plain construction starts a unitig at the
// first unvisited k-mer in sorted order, in its canonical orientation, and follows the links both ways: for a chain that never meets one of its
// own k-mers again that is the chain oriented so that its smallest canonical k-mer reads forwards, and the unitigs are numbered by those
// smallest k-mers. Chains that do meet themselves (closed loops, hairpins through a reverse complement) are left to the plain code, which
// then only sees their k-mers; all unitigs are put in the order of their first k-mers at the end. Returns false (nothing kept) if a k-mer
// ended up on two unitigs -- the caller then runs the plain construction.
static bool fast_unitigs(KTable<uint64_t>& km, const std::vector<uint64_t>& solid, int k, unsigned n_thr, std::vector<Unitig>& U, const DeviceUnitigs* dev) {
    const uint64_t mask = kmer_mask(k);
    auto in_graph = [&](uint64_t oriented) -> bool { return km.slot(kmer_canonical(oriented, k), false) != nullptr; };
    auto succs = [&](uint64_t x, uint64_t out[4]) -> int { int n = 0; for (uint64_t b = 0; b < 4; ++b) { const uint64_t y = ((x << 2) | b) & mask; if (in_graph(y)) out[n++] = y; } return n; };
    auto preds = [&](uint64_t x, uint64_t out[4]) -> int { int n = 0; for (uint64_t b = 0; b < 4; ++b) { const uint64_t y = (x >> 2) | (b << (2 * (k - 1))); if (in_graph(y)) out[n++] = y; } return n; };
    auto next = [&](uint64_t x, uint64_t* y) -> bool { uint64_t nb[4], nb2[4]; if (succs(x, nb) != 1) return false; if (preds(nb[0], nb2) != 1) return false; *y = nb[0]; return true; }; // the link the plain code follows forwards
    auto prev = [&](uint64_t x, uint64_t* y) -> bool { uint64_t nb[4], nb2[4]; if (preds(x, nb) != 1) return false; if (succs(nb[0], nb2) != 1) return false; *y = nb[0]; return true; };
    struct Rec { uint64_t seed; std::string seq; };
    std::vector<std::vector<Rec> > out(n_thr);
    std::atomic<bool> clash(false);
    auto claim = [&](uint64_t canonical) { uint64_t* v = km.slot(canonical, false); if (__atomic_exchange_n(v, 1ULL, __ATOMIC_RELAXED) != 0) clash = true; };
    if (!dev) parallel_for(solid.size(), n_thr, [&](size_t b, size_t e, unsigned t) {
        std::vector<uint64_t> path;
        for (size_t i = b; i < e && !clash; ++i) {
            const uint64_t s = solid[i]; uint64_t y;
            const bool has_fw = next(s, &y), has_bw = prev(s, &y);
            if (has_fw && has_bw) continue; // inside a chain (or on a closed loop)
            // walk inwards from this end: forwards from s if nothing links into it from behind, else forwards from its reverse complement
            uint64_t x = has_bw ? kmer_revcomp(s, k) : s;
            path.clear(); path.push_back(x);
            while (next(x, &y)) { path.push_back(y); x = y; if (path.size() > solid.size()) break; }
            const uint64_t end_c = kmer_canonical(path.back(), k);
            if (path.size() > 1 && end_c == s) continue;         // the chain comes back to its own first k-mer (hairpin): plain code
            if (end_c < s) continue;                              // the other end owns the chain
            if (path.size() > solid.size()) continue;
            // orient: the smallest canonical k-mer of the chain reads forwards
            size_t m = 0; uint64_t mc = kmer_canonical(path[0], k);
            for (size_t j = 1; j < path.size(); ++j) { const uint64_t c = kmer_canonical(path[j], k); if (c < mc) { mc = c; m = j; } }
            bool dup = false; // a chain that holds a k-mer and its reverse complement without coming back to its first k-mer cannot exist (the links are symmetric); checked by the claims below
            if (path[m] != mc) { std::reverse(path.begin(), path.end()); for (size_t j = 0; j < path.size(); ++j) path[j] = kmer_revcomp(path[j], k); }
            Rec r; r.seed = mc; r.seq = kmer_decode(path[0], k);
            for (size_t j = 1; j < path.size(); ++j) r.seq.push_back(bits2base(static_cast<int>(path[j] & 3)));
            for (size_t j = 0; j < path.size(); ++j) claim(kmer_canonical(path[j], k));
            (void)dup;
            out[t].push_back(r);
        }
    });
    if (clash) return false;
    // what is left belongs to chains that meet themselves: the plain construction, which finds every other k-mer taken
    std::vector<Rec> rest;
    {
        // (the k-mers no chain has claimed are looked for on all threads -- one table probe per solid k-mer, a cache miss each -- and come out in
        // sorted order, thread after thread; the plain construction then only visits those)
        std::vector<uint64_t> left;
        if (dev) left.assign(dev->left, dev->left + dev->n_left); // (--gpu: the chains were walked, written and claimed on the device)
        else {
            std::vector<std::vector<uint64_t> > left_t(n_thr);
            parallel_for(solid.size(), n_thr, [&](size_t b, size_t e, unsigned t) { for (size_t i = b; i < e; ++i) if (*km.slot(solid[i], false) == 0) left_t[t].push_back(solid[i]); });
            for (unsigned t = 0; t < n_thr; ++t) left.insert(left.end(), left_t[t].begin(), left_t[t].end());
        }
        size_t lcap = 16; while (lcap * 6 < left.size() * 10 + 16) lcap <<= 1; lcap <<= 1;
        KTable<uint64_t> lt(lcap); // the left-over k-mers: 0 = free, 1 = on a unitig built below; a k-mer that is not in it lies on a chain built above
        for (size_t li = 0; li < left.size(); ++li) *lt.slot(left[li], true) = 0;
        auto taken = [&](uint64_t c) -> bool { const uint64_t* v = lt.slot(c, false); return !v || *v != 0; };
        std::set<uint64_t> in_this;
        for (size_t li = 0; li < left.size(); ++li) {
            const uint64_t seed_km = left[li];
            if (taken(seed_km)) continue;
            in_this.clear(); in_this.insert(seed_km);
            std::vector<uint64_t> fwd(1, seed_km), bwd; uint64_t nb[4], nb2[4];
            for (uint64_t x = seed_km;;) { if (succs(x, nb) != 1) break; const uint64_t y = nb[0]; if (preds(y, nb2) != 1) break; const uint64_t cy = kmer_canonical(y, k); if (in_this.count(cy) || taken(cy)) break; in_this.insert(cy); fwd.push_back(y); x = y; }
            for (uint64_t x = seed_km;;) { if (preds(x, nb) != 1) break; const uint64_t y = nb[0]; if (succs(y, nb2) != 1) break; const uint64_t cy = kmer_canonical(y, k); if (in_this.count(cy) || taken(cy)) break; in_this.insert(cy); bwd.push_back(y); x = y; }
            std::vector<uint64_t> path(bwd.rbegin(), bwd.rend()); path.insert(path.end(), fwd.begin(), fwd.end());
            Rec r; r.seed = seed_km; r.seq = kmer_decode(path[0], k);
            for (size_t j = 1; j < path.size(); ++j) r.seq.push_back(bits2base(static_cast<int>(path[j] & 3)));
            for (size_t j = 0; j < path.size(); ++j) *lt.slot(kmer_canonical(path[j], k), false) = 1;
            rest.push_back(r);
        }
    }
    if (getenv("RTK_INDEX_TRACE")) fprintf(stderr, "rtk_build_index: %zu unitigs of chains that meet themselves built by the plain code\n", rest.size());
    // all unitigs in the order of their first k-mers; the table values from the final numbers
    std::vector<Rec*> all;
    std::vector<Rec> dev_recs;
    if (dev) { // (already in the order of their seeds; the sequences are cut out of the pool where the table values are set, below)
        dev_recs.resize(dev->n);
        parallel_for(dev_recs.size(), n_thr, [&](size_t b, size_t e, unsigned) { for (size_t i = b; i < e; ++i) { dev_recs[i].seed = dev->seeds[i]; dev_recs[i].seq.assign(dev->pool + dev->off[i], dev->pool + dev->off[i + 1]); } });
        for (size_t i = 0; i < dev_recs.size(); ++i) all.push_back(&dev_recs[i]);
    }
    for (unsigned t = 0; t < n_thr; ++t) for (size_t i = 0; i < out[t].size(); ++i) all.push_back(&out[t][i]);
    for (size_t i = 0; i < rest.size(); ++i) all.push_back(&rest[i]);
    std::sort(all.begin(), all.end(), [](const Rec* a, const Rec* b) { return a->seed < b->seed; });
    U.resize(all.size());
    parallel_for(all.size(), n_thr, [&](size_t b, size_t e, unsigned) {
        for (size_t uid = b; uid < e; ++uid) {
            U[uid].seq.swap(all[uid]->seq);
            const std::string& q = U[uid].seq; uint64_t fw = 0;
            for (size_t i = 0; i < q.size(); ++i) {
                fw = ((fw << 2) | static_cast<uint64_t>(base2bits(q[i]))) & mask;
                if (i + 1 < static_cast<size_t>(k)) continue;
                bool is_fw; const uint64_t c = kmer_canonical(fw, k, &is_fw);
                *km.slot(c, false) = ((static_cast<uint64_t>(uid) + 1) << 32) | (static_cast<uint64_t>(i + 1 - k) << 1) | (is_fw ? 1ULL : 0ULL);
            }
        }
    });
    return true;
}

template <class KM> static int run(int argc, char** argv) { // KM: uint64_t for k <= 31, u128 for k in 33..63
    const KM EMPTY = ~static_cast<KM>(0);
    std::vector<std::string> in_files;
    std::string prefix = "out";
    int k = 31;
    unsigned min_count = 2;
    size_t min_cov_vertices = 2;
    double global_cov_factor = 3.0, min_color_sharing = 0.5;
    bool detect_cycles = true, detect_snps = false;
    bool fast = false, gpu = false; // --fast: the same files from thread-parallel counting-table build / compaction / adjacency / cycle search; --gpu: --fast with the k-mers counted on the device
    std::string dump_input; // --dump-input FILE: the inputs (sample: sources included) written out as one FASTQ file, nothing else done
    std::vector<std::string> colour_files; // pass-2 index (`Ratatosk index -2`): colours = ids of these (pass-1 corrected long) reads, one id per read
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* n) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "rtk_build_index: missing value for %s\n", n); exit(2); } return argv[++i]; };
        if (a == "-s") in_files.push_back(need("-s"));
        else if (a == "-o") prefix = need("-o");
        else if (a == "-k") k = atoi(need("-k"));
        else if (a == "--min-count") min_count = static_cast<unsigned>(atoi(need("--min-count")));
        else if (a == "--global-cov-factor") global_cov_factor = atof(need("--global-cov-factor"));
        else if (a == "--no-short-cycles") detect_cycles = false;
        else if (a == "--snps") detect_snps = true;
        else if (a == "--fast") fast = true;
        else if (a == "--gpu") { fast = true; gpu = true; }
        else if (a == "--colour-reads") colour_files.push_back(need("--colour-reads"));
        else if (a == "--dump-input") dump_input = need("--dump-input");
        else { fprintf(stderr, "rtk_build_index: unknown option %s\n", a.c_str()); return 2; }
    }
    if (in_files.empty() || k < 3 || k > RTK_MAX_K || !(k & 1)) { fprintf(stderr, "usage: rtk_build_index -s reads.fq [-s ...] -o PREFIX [-k 31 (odd, <=63)] [--min-count 2] [--global-cov-factor 3.0] [--no-short-cycles] [--snps] [--fast | --gpu (k <= 31: same files, threads / the device for the heavy steps)] [--dump-input FILE] [--colour-reads corrected_long_reads.fq: second-pass index, the graph comes from -s, colours and coverage from these reads]\n"); return 2; }
    if (!dump_input.empty()) { // what a `sample:` source stands for, as a file (tests compare the index built from either)
        FILE* fo = fopen(dump_input.c_str(), "wb"); if (!fo) { fprintf(stderr, "rtk_build_index: cannot write %s\n", dump_input.c_str()); return 1; }
        std::string name, seq, qual;
        for (size_t f = 0; f < in_files.size(); ++f) { FastxReader fr; if (!fr.open(in_files[f])) { fprintf(stderr, "rtk_build_index: cannot open %s\n", in_files[f].c_str()); return 1; }
            while (fr.next(name, seq, qual)) fprintf(fo, "@%s\n%s\n+\n%s\n", name.c_str(), seq.c_str(), qual.empty() ? std::string(seq.size(), 'I').c_str() : qual.c_str()); }
        fclose(fo); return 0;
    }
    const KM mask = km_mask<KM>(k);
    if (fast && sizeof(KM) != 8) { fprintf(stderr, "rtk_build_index: --fast / --gpu serve one-word k-mers (k <= 31); k = %d takes the plain path\n", k); fast = false; gpu = false; }
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (getenv("RTK_INDEX_TRACE")) fprintf(stderr, "rtk_build_index: [%8.2f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), what); };

    // ---- pass 1: count canonical k-mers. The k-mer space is cut into one shard per thread by a hash; every thread reads the input
    // itself (parsing is cheap next to a table insert) and counts the k-mers of its shard in a table of its own ----
    unsigned n_thr = std::thread::hardware_concurrency(); if (n_thr == 0) n_thr = 1; if (n_thr > (fast ? 128u : 32u)) n_thr = fast ? 128u : 32u; // (--fast: the steps are random accesses into GB-sized tables: latency-bound, SMT threads help)
    { const char* e = getenv("RTK_INDEX_THREADS"); if (e && atoi(e) > 0) n_thr = static_cast<unsigned>(atoi(e)); }
    std::vector<KM> solid;
    typedef int (*unitigs_fn)(int, int, const uint64_t*, uint64_t, char**, uint64_t**, uint64_t**, uint64_t*, uint64_t**, uint64_t*);
    typedef const char* (*gerr_fn)(void); typedef void (*gfree_fn)(void*);
    unitigs_fn gpu_unitigs_fn = nullptr; gerr_fn gpu_err_fn = nullptr; gfree_fn gpu_free_fn = nullptr;
    typedef int (*col_begin_fn)(int, int, const char*, const uint64_t*, uint64_t, void**); typedef int (*col_chunk_fn)(void*, const char*, uint64_t, const uint64_t*, const uint32_t*, uint32_t); typedef int (*col_end_fn)(void*, uint64_t**, uint64_t*, uint64_t**);
    col_begin_fn gpu_col_begin = nullptr; col_chunk_fn gpu_col_chunk = nullptr; col_end_fn gpu_col_end = nullptr;
    if (gpu) { // the k-mers counted on the device (csrc/hip/rtk_index.hip, through the C ABI of libratatosk_hip.so next to this executable)
        typedef int (*count_fn)(int, int, const char* const*, int, uint32_t, int, uint64_t**, uint64_t*);
        typedef const char* (*err_fn)(void); typedef void (*free_fn)(void*);
        std::string lib = "libratatosk_hip.so";
        { char exe[4096]; const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1); if (n > 0) { exe[n] = 0; std::string d(exe); d = d.substr(0, d.rfind('/')); lib = d + "/../libratatosk_hip.so"; } }
        void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "rtk_build_index: --gpu: cannot load %s (%s)\n", lib.c_str(), dlerror()); return 1; }
        count_fn cf = reinterpret_cast<count_fn>(dlsym(h, "rtk_index_count_kmers")); err_fn ef = reinterpret_cast<err_fn>(dlsym(h, "rtk_last_error")); free_fn ff = reinterpret_cast<free_fn>(dlsym(h, "rtk_free"));
        gpu_unitigs_fn = reinterpret_cast<unitigs_fn>(dlsym(h, "rtk_index_unitigs")); gpu_err_fn = ef; gpu_free_fn = ff;
        gpu_col_begin = reinterpret_cast<col_begin_fn>(dlsym(h, "rtk_index_colour_begin")); gpu_col_chunk = reinterpret_cast<col_chunk_fn>(dlsym(h, "rtk_index_colour_chunk")); gpu_col_end = reinterpret_cast<col_end_fn>(dlsym(h, "rtk_index_colour_end"));
        if (!cf || !ef || !ff) { fprintf(stderr, "rtk_build_index: --gpu: %s lacks the index entry points\n", lib.c_str()); return 1; }
        std::vector<const char*> fp; for (size_t f = 0; f < in_files.size(); ++f) fp.push_back(in_files[f].c_str());
        uint64_t* sk = nullptr; uint64_t ns = 0;
        if (cf(0, k, fp.data(), static_cast<int>(fp.size()), min_count, static_cast<int>(n_thr), &sk, &ns) != 0) { fprintf(stderr, "rtk_build_index: --gpu: %s\n", ef()); return 1; }
        solid.resize(ns);
        for (uint64_t i = 0; i < ns; ++i) solid[i] = static_cast<KM>(sk[i]);
        ff(sk);
    } else
    {
        std::vector<std::vector<KM> > part(n_thr);
        std::vector<int> bad(n_thr, 0);
        auto count_shard = [&](unsigned t) {
            KTable<KM> cnt(1 << 20);
            std::string name, seq, qual;
            for (size_t f = 0; f < in_files.size(); ++f) {
                FastxReader fr;
                if (!fr.open(in_files[f])) { bad[t] = 1; return; }
                while (fr.next(name, seq, qual)) {
                    KM fw = 0; int valid = 0;
                    for (size_t i = 0; i < seq.size(); ++i) {
                        const int b = base2bits(seq[i]);
                        if (b < 0) { valid = 0; fw = 0; continue; }
                        fw = ((fw << 2) | static_cast<KM>(b)) & mask;
                        if (++valid >= k) { const KM c = kmer_canonical(fw, k); if ((hash_km(c) >> 40) % n_thr == t) ++*cnt.slot(c, true); }
                    }
                }
                if (fr.failed()) { bad[t] = 2; return; } // a damaged or cut-short gzip stream is not the end of the reads
            }
            for (size_t i = 0; i < cnt.keys.size(); ++i) if (cnt.keys[i] != EMPTY && cnt.vals[i] >= min_count) part[t].push_back(cnt.keys[i]);
        };
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(count_shard, t);
        for (size_t t = 0; t < th.size(); ++t) th[t].join();
        for (unsigned t = 0; t < n_thr; ++t) if (bad[t]) { fprintf(stderr, bad[t] == 2 ? "rtk_build_index: an input file ends in a damaged or cut-short gzip stream\n" : "rtk_build_index: cannot open an input file\n"); return 1; }
        // ---- solid k-mers, sorted: unitig construction is independent of table layout ----
        for (unsigned t = 0; t < n_thr; ++t) { solid.insert(solid.end(), part[t].begin(), part[t].end()); std::vector<KM>().swap(part[t]); }
        std::sort(solid.begin(), solid.end());
    }
    lap("k-mers counted");
    size_t cap = 16; while (cap * 6 < solid.size() * 10 + 16) cap <<= 1; if (solid.size() < (1ull << 30)) cap <<= 1; // (load <= 0.6; below 2^30 k-mers half of that: a 3 Gb genome's table is 137 GB instead of 275)
    KTable<KM> km(cap); // canonical solid k-mer -> 0 (unvisited) or (unitig+1)<<32 | offset<<1 | fw_flag
    if (fast) fast_table_fill(km, solid, n_thr);
    else for (size_t i = 0; i < solid.size(); ++i) *km.slot(solid[i], true) = 0;
    fprintf(stderr, "rtk_build_index: %zu solid %d-mers\n", solid.size(), k);
    lap("k-mer table filled");

    auto in_graph = [&](KM oriented) -> bool { return km.slot(kmer_canonical(oriented, k), false) != nullptr; };
    auto succs = [&](KM x, KM out[4]) -> int { int n = 0; for (uint64_t b = 0; b < 4; ++b) { const KM y = ((x << 2) | static_cast<KM>(b)) & mask; if (in_graph(y)) out[n++] = y; } return n; };
    auto preds = [&](KM x, KM out[4]) -> int { int n = 0; for (uint64_t b = 0; b < 4; ++b) { const KM y = (x >> 2) | (static_cast<KM>(b) << (2 * (k - 1))); if (in_graph(y)) out[n++] = y; } return n; };

    // ---- unitigs: maximal non-branching paths ----
    std::vector<Unitig> U;
    bool fast_done = false;
    DeviceUnitigs dev_u; bool have_dev_u = false;
    char* du_pool = nullptr; uint64_t* du_off = nullptr; uint64_t* du_seeds = nullptr; uint64_t* du_left = nullptr;
    if (gpu && gpu_unitigs_fn && sizeof(KM) == 8 && !getenv("RTK_INDEX_HOST_UNITIGS")) { // the chains walked and written on the device (csrc/hip/rtk_index.hip rtk_index_unitigs)
        uint64_t nu = 0, nl = 0;
        const int rc = gpu_unitigs_fn(0, k, reinterpret_cast<const uint64_t*>(solid.data()), solid.size(), &du_pool, &du_off, &du_seeds, &nu, &du_left, &nl);
        if (rc == 0) { dev_u.pool = du_pool; dev_u.off = du_off; dev_u.seeds = du_seeds; dev_u.n = nu; dev_u.left = du_left; dev_u.n_left = nl; have_dev_u = true; }
        else fprintf(stderr, "rtk_build_index: --gpu: unitigs on the host threads (%s)\n", gpu_err_fn ? gpu_err_fn() : "?");
    }
    if (fast) fast_done = fast_unitigs(km, solid, k, n_thr, U, have_dev_u ? &dev_u : nullptr);
    if (gpu_free_fn) { gpu_free_fn(du_pool); gpu_free_fn(du_off); gpu_free_fn(du_seeds); gpu_free_fn(du_left); }
    if (!fast_done)
    {
        if (fast) for (size_t i = 0; i < km.vals.size(); ++i) km.vals[i] = 0; // (the thread-parallel construction backed out: every k-mer unvisited again)
        U.clear();
        std::set<KM> in_this; // canonical k-mers of the unitig being built (cycle / hairpin guard)
        for (size_t si = 0; si < solid.size(); ++si) {
            uint64_t* v0 = km.slot(solid[si], false);
            if (*v0 != 0) continue;
            in_this.clear(); in_this.insert(solid[si]);
            std::vector<KM> fwd(1, solid[si]), bwd; // oriented k-mers
            KM nb[4], nb2[4];
            for (KM x = solid[si];;) { // extend forward
                if (succs(x, nb) != 1) break;
                const KM y = nb[0];
                if (preds(y, nb2) != 1) break;
                const KM cy = kmer_canonical(y, k);
                if (in_this.count(cy) || *km.slot(cy, false) != 0) break;
                in_this.insert(cy); fwd.push_back(y); x = y;
            }
            for (KM x = solid[si];;) { // extend backward
                if (preds(x, nb) != 1) break;
                const KM y = nb[0];
                if (succs(y, nb2) != 1) break;
                const KM cy = kmer_canonical(y, k);
                if (in_this.count(cy) || *km.slot(cy, false) != 0) break;
                in_this.insert(cy); bwd.push_back(y); x = y;
            }
            std::vector<KM> path(bwd.rbegin(), bwd.rend());
            path.insert(path.end(), fwd.begin(), fwd.end());
            Unitig u;
            u.seq = km_decode<KM>(path[0], k);
            for (size_t i = 1; i < path.size(); ++i) u.seq.push_back(bits2base(static_cast<int>(static_cast<uint64_t>(path[i]) & 3)));
            const uint64_t uid = U.size();
            for (size_t i = 0; i < path.size(); ++i) {
                bool is_fw; const KM c = kmer_canonical(path[i], k, &is_fw);
                *km.slot(c, false) = ((uid + 1) << 32) | (static_cast<uint64_t>(i) << 1) | (is_fw ? 1ULL : 0ULL);
            }
            U.push_back(u);
        }
    }
    fprintf(stderr, "rtk_build_index: %zu unitigs\n", U.size());
    lap("unitigs built");
    // The unitig FASTA only needs the sequences: with --fast it is compressed on threads of its own while the colours, annotations and records are worked out.
    // Groups of unitigs of >= 32 MB of text are gzip MEMBERS of their own (level 6; a concatenation of members is an ordinary gzip file: zlib's gzread, Bifrost's
    // reader, reads through them): compressed side by side here and inflated side by side by the loader (common/mgzip.hpp). The same bytes with any number of threads.
    const std::string fn_fasta = prefix + ".index.k" + std::to_string(k) + ".fasta.gz";
    std::atomic<int> fasta_rc(0);
    auto write_fasta = [&]() {
        std::vector<size_t> cut(1, 0);
        const size_t member_text = getenv("RTK_FASTA_MEMBER_BYTES") ? static_cast<size_t>(strtoull(getenv("RTK_FASTA_MEMBER_BYTES"), nullptr, 10)) : (32u << 20); // (tests: many small members)
        { size_t bytes = 0; for (size_t u = 0; u < U.size(); ++u) { bytes += U[u].seq.size() + 12; if (bytes >= member_text) { cut.push_back(u + 1); bytes = 0; } } if (cut.back() != U.size() || cut.size() == 1) cut.push_back(U.size()); }
        FILE* fp = fopen(fn_fasta.c_str(), "wb");
        if (!fp) { fasta_rc = 1; return; }
        const size_t n_groups = cut.size() - 1, nt = fast ? std::min<size_t>(std::max<size_t>(1, n_thr / 4), 16) : 1;
        auto member = [&](size_t g, std::string& out) -> bool {
            std::string text; char name[32];
            for (size_t u = cut[g]; u < cut[g + 1]; ++u) { const int nn = snprintf(name, sizeof(name), ">%zu\n", u); text.append(name, static_cast<size_t>(nn)); text += U[u].seq; text.push_back('\n'); }
            z_stream zs; memset(&zs, 0, sizeof(zs));
            if (deflateInit2(&zs, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
            out.resize(deflateBound(&zs, static_cast<uLong>(text.size())) + 64);
            zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(text.data())); zs.avail_in = static_cast<uInt>(text.size());
            zs.next_out = reinterpret_cast<Bytef*>(&out[0]); zs.avail_out = static_cast<uInt>(out.size());
            const int rc = deflate(&zs, Z_FINISH); out.resize(out.size() - zs.avail_out); deflateEnd(&zs);
            return rc == Z_STREAM_END;
        };
        for (size_t g0 = 0; g0 < n_groups && !fasta_rc; g0 += nt) {
            const size_t g1 = std::min(n_groups, g0 + nt);
            std::vector<std::string> out(g1 - g0); std::vector<int> ok(g1 - g0, 0);
            std::vector<std::thread> th;
            for (size_t g = g0 + 1; g < g1; ++g) th.emplace_back([&, g]() { ok[g - g0] = member(g, out[g - g0]) ? 1 : 0; });
            ok[0] = member(g0, out[0]) ? 1 : 0;
            for (size_t t = 0; t < th.size(); ++t) th[t].join();
            for (size_t g = g0; g < g1; ++g) if (!ok[g - g0] || fwrite(out[g - g0].data(), 1, out[g - g0].size(), fp) != out[g - g0].size()) fasta_rc = 1;
        }
        if (fclose(fp) != 0) fasta_rc = 1;
    };
    std::thread fasta_thread;
    if (fast) fasta_thread = std::thread(write_fasta);

    // ---- pass 2: colours (pair ids) and coverage. One reader parses the records and numbers them (a pair keeps one id), worker threads
    // look their k-mers up (the table is only read) and collect (unitig, id) events and per-unitig counts of their own ----
    {
        // second-pass index: the reads that colour the graph are the (pass-1 corrected) long reads, every read its own id
        // (addCoverage(dbg, opt_pass2, ..., long_read_correct = true), src/Ratatosk.cpp:1218)
        const bool by_read = !colour_files.empty();
        const std::vector<std::string>& col_in = by_read ? colour_files : in_files;
        struct Chunk { std::vector<std::string> seq; std::vector<uint32_t> id; size_t bytes = 0; };
        std::mutex mq; std::condition_variable cv_put, cv_get; std::deque<Chunk*> q; bool done = false; int open_failed = 0;
        const size_t n_u = U.size();
        std::vector<std::vector<uint64_t> > t_cov(n_thr); std::vector<std::vector<std::pair<uint32_t, uint32_t> > > t_ev(n_thr);
        // --gpu: the reads are handed to the device chunk by chunk (csrc/hip/rtk_index.hip rtk_index_colour_*: the k-mer table of the unitigs in HBM, one lane per
        // read position); this tool keeps what is its own -- reading, and the numbering of the reads. Every thread fills a chunk of its own.
        void* col_job = nullptr; std::atomic<int> col_failed(0);
        if (gpu && gpu_col_begin && gpu_col_chunk && gpu_col_end && sizeof(KM) == 8 && !getenv("RTK_INDEX_HOST_COLOURS") && n_u > 0) {
            std::vector<uint64_t> off(n_u + 1, 0); for (size_t u = 0; u < n_u; ++u) off[u + 1] = off[u] + U[u].seq.size();
            std::string pool(off[n_u], 'A');
            parallel_for(n_u, n_thr, [&](size_t b, size_t e, unsigned) { for (size_t u = b; u < e; ++u) memcpy(&pool[off[u]], U[u].seq.data(), U[u].seq.size()); });
            if (gpu_col_begin(0, k, pool.data(), off.data(), n_u, &col_job) != 0) { fprintf(stderr, "rtk_build_index: --gpu: colours on the host threads (%s)\n", gpu_err_fn()); col_job = nullptr; }
        }
        struct Feed {
            std::string chars; std::vector<uint64_t> starts; std::vector<uint32_t> ids; void* job; col_chunk_fn fn; std::atomic<int>* failed;
            void flush() { if (ids.empty()) return; if (fn(job, chars.data(), chars.size(), starts.data(), ids.data(), static_cast<uint32_t>(ids.size())) != 0) *failed = 1; chars.clear(); starts.clear(); ids.clear(); }
            void add(const char* seq, size_t len, uint32_t id) {
                if (len + 1 > (60u << 20)) { *failed = 1; return; } // (a read longer than a chunk)
                if (chars.size() + len + 1 > (60u << 20) || ids.size() >= (2u << 20) || (chars.size() >= (24u << 20))) flush();
                starts.push_back(chars.size()); ids.push_back(id); chars.append(seq, len); chars.push_back('\n');
            }
        };
        std::vector<Feed> feeds(n_thr);
        for (unsigned t = 0; t < n_thr; ++t) { feeds[t].job = col_job; feeds[t].fn = gpu_col_chunk; feeds[t].failed = &col_failed; }
        auto work = [&](unsigned t) {
            std::vector<uint64_t>& cov = t_cov[t]; if (!col_job) cov.assign(n_u, 0);
            std::vector<std::pair<uint32_t, uint32_t> >& ev = t_ev[t];
            while (true) {
                Chunk* c = nullptr;
                { std::unique_lock<std::mutex> lk(mq); cv_get.wait(lk, [&]() { return !q.empty() || done; }); if (q.empty()) return; c = q.front(); q.pop_front(); }
                cv_put.notify_one();
                for (size_t r = 0; r < c->seq.size(); ++r) {
                    const std::string& seq = c->seq[r]; const uint32_t pair_id = c->id[r];
                    if (col_job) { feeds[t].add(seq.data(), seq.size(), pair_id); continue; }
                    KM fw = 0; int valid = 0;
                    for (size_t x = 0; x < seq.size(); ++x) {
                        const int b = base2bits(seq[x]);
                        if (b < 0) { valid = 0; fw = 0; continue; }
                        fw = ((fw << 2) | static_cast<KM>(b)) & mask;
                        if (++valid >= k) {
                            const uint64_t* v = km.slot(kmer_canonical(fw, k), false);
                            if (v) { const uint32_t u = static_cast<uint32_t>((*v >> 32) - 1); ++cov[u]; if (ev.empty() || ev.back().first != u || ev.back().second != pair_id) ev.push_back(std::make_pair(u, pair_id)); }
                        }
                    }
                }
                delete c;
            }
        };
        bool par_colour = fast;
        bool all_sampled = fast && !by_read && !col_in.empty();
        for (size_t f = 0; all_sampled && f < col_in.size(); ++f) all_sampled = SampleSource::is_spec(col_in[f]);
        if (all_sampled) {
            // reads sampled from a reference on the fly (common/sample_source.hpp): pair p of a source has the id (pairs of the sources before it) + p
            // (the number of name changes before it: mates share the name "s<p>"); ranges of pairs generated and looked up by all threads
            par_colour = false;
            if (!col_job) for (unsigned t = 0; t < n_thr; ++t) t_cov[t].assign(n_u, 0);
            uint64_t id_base = 0;
            for (size_t f = 0; f < col_in.size() && !open_failed; ++f) {
                std::string err; std::shared_ptr<SampleSource> ss = SampleSource::get(col_in[f], &err);
                if (!ss) { fprintf(stderr, "rtk_build_index: %s\n", err.c_str()); open_failed = 1; break; }
                if (id_base + ss->n_pairs() > 0xFFFFFFFFull) { fprintf(stderr, "rtk_build_index: more than 2^32 read pairs\n"); open_failed = 1; break; }
                const uint64_t per = 1 << 14, n_ch = (ss->n_pairs() + per - 1) / per; const uint32_t L = ss->read_len();
                std::atomic<uint64_t> nx(0);
                std::vector<std::thread> th;
                for (unsigned t = 0; t < n_thr; ++t) th.emplace_back([&, t]() {
                    std::vector<uint64_t>& cov = t_cov[t]; std::vector<std::pair<uint32_t, uint32_t> >& ev = t_ev[t];
                    std::string m(2 * static_cast<size_t>(L), 'A');
                    for (;;) { const uint64_t c = nx.fetch_add(1); if (c >= n_ch) break;
                        const uint64_t p0 = c * per, p1 = std::min<uint64_t>(ss->n_pairs(), p0 + per);
                        for (uint64_t p = p0; p < p1; ++p) {
                            ss->pair(p, &m[0], &m[L]);
                            const uint32_t id = static_cast<uint32_t>(id_base + p);
                            for (int mate = 0; mate < 2; ++mate) {
                                if (col_job) { feeds[t].add(m.data() + mate * L, L, id); continue; }
                                const char* seq = m.data() + mate * L; KM fw = 0; int valid = 0;
                                for (uint32_t y = 0; y < L; ++y) {
                                    const int b = base2bits(seq[y]);
                                    if (b < 0) { valid = 0; fw = 0; continue; }
                                    fw = ((fw << 2) | static_cast<KM>(b)) & mask;
                                    if (++valid >= k) {
                                        const uint64_t* v = km.slot(kmer_canonical(fw, k), false);
                                        if (v) { const uint32_t u = static_cast<uint32_t>((*v >> 32) - 1); ++cov[u]; if (ev.empty() || ev.back().first != u || ev.back().second != id) ev.push_back(std::make_pair(u, id)); }
                                    }
                                }
                            }
                        }
                    } });
                for (size_t t = 0; t < th.size(); ++t) th[t].join();
                id_base += ss->n_pairs();
            }
        }
        for (size_t f = 0; par_colour && f < col_in.size(); ++f) par_colour = PlainChunks::is_plain(col_in[f]);
        if (par_colour) {
            // --fast on plain files: byte ranges of the files parsed and looked up by all threads. The id of a read is the number of name changes
            // before it (every read with --colour-reads), so a first sweep over the ranges counts the changes inside each and notes its first and last
            // name; the running sums give every range the id of its first read; the second sweep maps the reads.
            struct RangeInfo { uint32_t changes = 0; uint64_t n_reads = 0; std::string first, last; };
            auto base_name = [](const PackedReads& r, size_t i, const char** p, size_t* n) { *p = r.name(i); *n = r.name_len(i); if (*n > 2 && (*p)[*n - 2] == '/' && ((*p)[*n - 1] == '1' || (*p)[*n - 1] == '2')) *n -= 2; };
            if (!col_job) for (unsigned t = 0; t < n_thr; ++t) t_cov[t].assign(n_u, 0);
            uint32_t next_id = 0; bool have_prev = false; std::string prev_last;
            for (size_t f = 0; f < col_in.size() && !open_failed; ++f) {
                PlainChunks pc; if (!pc.open(col_in[f], 32u << 20)) { fprintf(stderr, "rtk_build_index: cannot open %s\n", col_in[f].c_str()); open_failed = 1; break; }
                const size_t nc = pc.n_chunks();
                std::vector<RangeInfo> info(nc);
                std::atomic<size_t> nx(0); std::atomic<int> bad(0);
                { std::vector<std::thread> th;
                  for (unsigned t = 0; t < n_thr; ++t) th.emplace_back([&]() {
                      for (;;) { const size_t i = nx.fetch_add(1); if (i >= nc) break;
                          PackedReads r(false); if (!pc.parse_chunk(i, r)) { bad = 1; break; }
                          RangeInfo& ri = info[i]; ri.n_reads = r.size();
                          const char* pp = nullptr; size_t pn = 0;
                          for (size_t x = 0; x < r.size(); ++x) { const char* p; size_t n; base_name(r, x, &p, &n); if (x == 0) ri.first.assign(p, n); else if (by_read || n != pn || memcmp(p, pp, n) != 0) ++ri.changes; pp = p; pn = n; }
                          if (r.size()) ri.last.assign(pp, pn);
                      } });
                  for (size_t t = 0; t < th.size(); ++t) th[t].join(); }
                if (bad) { open_failed = 1; break; }
                std::vector<uint32_t> id0(nc, 0); // id of the first read of every range
                for (size_t i = 0; i < nc; ++i) {
                    if (info[i].n_reads == 0) { id0[i] = next_id; continue; }
                    if (have_prev && (by_read || info[i].first != prev_last)) ++next_id;
                    id0[i] = next_id; next_id += info[i].changes; have_prev = true; prev_last = info[i].last;
                }
                nx = 0;
                { std::vector<std::thread> th;
                  for (unsigned t = 0; t < n_thr; ++t) th.emplace_back([&, t]() {
                      std::vector<uint64_t>& cov = t_cov[t]; std::vector<std::pair<uint32_t, uint32_t> >& ev = t_ev[t];
                      for (;;) { const size_t i = nx.fetch_add(1); if (i >= nc) break;
                          PackedReads r(false); if (!pc.parse_chunk(i, r)) { bad = 1; break; }
                          uint32_t id = id0[i]; const char* pp = nullptr; size_t pn = 0;
                          for (size_t x = 0; x < r.size(); ++x) {
                              const char* p; size_t n; base_name(r, x, &p, &n);
                              if (x != 0 && (by_read || n != pn || memcmp(p, pp, n) != 0)) ++id;
                              pp = p; pn = n;
                              const char* seq = r.seq(x); const size_t sl = r.seq_len(x);
                              if (col_job) { feeds[t].add(seq, sl, id); continue; }
                              KM fw = 0; int valid = 0;
                              for (size_t y = 0; y < sl; ++y) {
                                  const int b = base2bits(seq[y]);
                                  if (b < 0) { valid = 0; fw = 0; continue; }
                                  fw = ((fw << 2) | static_cast<KM>(b)) & mask;
                                  if (++valid >= k) {
                                      const uint64_t* v = km.slot(kmer_canonical(fw, k), false);
                                      if (v) { const uint32_t u = static_cast<uint32_t>((*v >> 32) - 1); ++cov[u]; if (ev.empty() || ev.back().first != u || ev.back().second != id) ev.push_back(std::make_pair(u, id)); }
                                  }
                              }
                          }
                      } });
                  for (size_t t = 0; t < th.size(); ++t) th[t].join(); }
                if (bad) { open_failed = 1; break; }
            }
        }
        std::vector<std::thread> th;
        if (!par_colour && !all_sampled) for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(work, t);
        if (!par_colour && !all_sampled) {
            std::string name, seq, qual, prev_name;
            uint32_t pair_id = 0; bool first = true;
            Chunk* cur = new Chunk();
            auto flush = [&]() { if (cur->seq.empty()) return; { std::unique_lock<std::mutex> lk(mq); cv_put.wait(lk, [&]() { return q.size() < 4u * n_thr; }); q.push_back(cur); } cv_get.notify_one(); cur = new Chunk(); };
            for (size_t f = 0; f < col_in.size() && !open_failed; ++f) {
                FastxReader fr; if (!fr.open(col_in[f], fast ? static_cast<int>(n_thr < 8 ? n_thr : 8) : 0)) { fprintf(stderr, "rtk_build_index: cannot open %s\n", col_in[f].c_str()); open_failed = 1; break; }
                while (fr.next(name, seq, qual)) {
                    for (size_t x = 0; x < seq.size(); ++x) seq[x] = static_cast<char>(seq[x] & 0xDF);
                    if (name.size() > 2 && name[name.size() - 2] == '/' && (name[name.size() - 1] == '1' || name[name.size() - 1] == '2')) name.erase(name.size() - 2);
                    if (first) { first = false; prev_name = name; }
                    else if (by_read || name != prev_name) { ++pair_id; prev_name = name; }
                    cur->bytes += seq.size(); cur->seq.push_back(std::string()); cur->seq.back().swap(seq); cur->id.push_back(pair_id);
                    if (cur->bytes >= (1u << 20)) flush();
                }
                if (fr.failed()) { fprintf(stderr, "rtk_build_index: %s ends in a damaged or cut-short gzip stream\n", col_in[f].c_str()); open_failed = 1; }
            }
            flush(); delete cur;
            { std::lock_guard<std::mutex> lk(mq); done = true; }
            cv_get.notify_all();
        }
        for (size_t t = 0; t < th.size(); ++t) th[t].join();
        if (col_job) { // the distinct (unitig, id) events in sorted order and the coverages, back from the device
            for (unsigned t = 0; t < n_thr; ++t) feeds[t].flush();
            uint64_t* ev = nullptr; uint64_t* cv = nullptr; uint64_t n_ev = 0;
            if (gpu_col_end(col_job, &ev, &n_ev, &cv) != 0 || col_failed) { fprintf(stderr, "rtk_build_index: --gpu: colouring on the device failed (%s)\n", gpu_err_fn()); if (fasta_thread.joinable()) fasta_thread.join(); return 1; }
            parallel_for(n_u, n_thr, [&](size_t b, size_t e, unsigned) {
                if (b >= e) return;
                const uint64_t* p = std::lower_bound(ev, ev + n_ev, static_cast<uint64_t>(b) << 32);
                for (size_t u = b; u < e; ++u) { U[u].cov = cv[u]; const uint64_t* q = p; while (q < ev + n_ev && (*q >> 32) == u) ++q; U[u].colours.resize(static_cast<size_t>(q - p)); for (size_t i = 0; p + i < q; ++i) U[u].colours[i] = static_cast<uint32_t>(p[i] & 0xFFFFFFFFull); p = q; }
            });
            gpu_free_fn(ev); gpu_free_fn(cv);
        }
        if (open_failed) { if (fasta_thread.joinable()) fasta_thread.join(); return 1; }
        for (unsigned t = 0; t < n_thr; ++t) {
            if (t_cov[t].size() != n_u) continue; // (--gpu: nothing was counted here)
            for (size_t u = 0; u < n_u; ++u) U[u].cov += t_cov[t][u];
            for (size_t e = 0; e < t_ev[t].size(); ++e) U[t_ev[t][e].first].colours.push_back(t_ev[t][e].second);
            std::vector<uint64_t>().swap(t_cov[t]); std::vector<std::pair<uint32_t, uint32_t> >().swap(t_ev[t]);
        }
        if (!col_job) parallel_for(U.size(), fast ? n_thr : 1u, [&](size_t b, size_t e, unsigned) { for (size_t i = b; i < e; ++i) { std::sort(U[i].colours.begin(), U[i].colours.end()); U[i].colours.erase(std::unique(U[i].colours.begin(), U[i].colours.end()), U[i].colours.end()); } });
    }

    // ---- adjacency, branching, edge bits ----
    const size_t n = U.size();
    struct Nb { int64_t u[2][4]; }; // [dir 0 = fw successors, 1 = successors of the reverse strand][base] -> unitig id or -1
    std::vector<Nb> adj(n);
    std::vector<uint64_t> kmcov(n, 0), shared(n, 0);
    auto shared_count = [&](const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) -> size_t {
        size_t i = 0, j = 0, c = 0;
        while (i < a.size() && j < b.size()) { if (a[i] < b[j]) ++i; else if (b[j] < a[i]) ++j; else { ++c; ++i; ++j; } }
        return c;
    };
    lap("colours and coverage done");
    auto adjacency_of = [&](size_t u) {
        const std::string& s = U[u].seq;
        KM tail = 0, head = 0;
        km_encode<KM>(s.c_str() + s.size() - k, k, tail);
        km_encode<KM>(s.c_str(), k, head);
        const KM ends[2] = { tail, kmer_revcomp(head, k) }; // last k-mer in walk direction fw / rev
        int deg[2] = {0, 0};
        for (int d = 0; d < 2; ++d) for (uint64_t b = 0; b < 4; ++b) {
            adj[u].u[d][b] = -1;
            const KM y = ((ends[d] << 2) | static_cast<KM>(b)) & mask;
            const uint64_t* v = km.slot(kmer_canonical(y, k), false);
            if (!v) continue;
            const size_t w = (*v >> 32) - 1;
            adj[u].u[d][b] = static_cast<int64_t>(w);
            ++deg[d];
            if (shared_count(U[u].colours, U[w].colours) >= min_cov_vertices) shared[u] |= (d == 0) ? ((1ULL << b) << 4) : (1ULL << b); // idx(A,C,G,T)=1,2,4,8 (src/Common.hpp:260,358)
        }
        const uint64_t cov = std::min<uint64_t>(U[u].cov, 0x7fffffffULL);
        kmcov[u] = (cov << 31) | ((deg[0] > 1 || deg[1] > 1) ? (1ULL << 63) : 0ULL);
    };
    if (fast) parallel_for(n, n_thr, [&](size_t b, size_t e, unsigned) { for (size_t u = b; u < e; ++u) adjacency_of(u); }); // (unitigs are independent: the table is only read)
    else for (size_t u = 0; u < n; ++u) adjacency_of(u);
    lap("adjacency and edge bits done");

    // ---- short cycles (restatement of detectShortCycles, src/Graph.cpp:4660-4735): for every unitig U in forward direction, breadth
    // first over paths U -> X1 .. Xm -> U whose interior spans fewer than k + 1 k-mers, following only edges carrying an edge bit and
    // unitigs sharing >= min_cov colours with U; a cycle counts when its interior unitigs are distinct and U's colours intersected
    // with theirs keep >= min_cov ids. Stored per unitig as the entering bases of X1..Xm (Path::getMiddleCompactedPath), NUL-terminated.
    std::vector<std::string> cycles(n);
    if (detect_cycles) {
        std::vector<KM> headk(n);
        for (size_t u = 0; u < n; ++u) km_encode<KM>(U[u].seq.c_str(), k, headk[u]);
        auto n_km = [&](size_t u) { return U[u].seq.size() - static_cast<size_t>(k) + 1; };
        struct Step { size_t u; bool fw; char base; };
        size_t n_cyc_unitigs = 0;
        auto cycles_of = [&](size_t u0) {
            std::queue<std::vector<Step> > q;
            { std::vector<Step> p0; Step s0; s0.u = u0; s0.fw = true; s0.base = 0; p0.push_back(s0); q.push(p0); }
            while (!q.empty()) {
                const std::vector<Step> path = q.front(); q.pop();
                const Step cur = path.back();
                const std::string& cs = U[cur.u].seq;
                KM tail = 0, head = 0;
                km_encode<KM>(cs.c_str() + cs.size() - k, k, tail); km_encode<KM>(cs.c_str(), k, head);
                const KM endk = cur.fw ? tail : kmer_revcomp(head, k);
                for (uint64_t b = 0; b < 4; ++b) {
                    const int64_t w = adj[cur.u].u[cur.fw ? 0 : 1][b];
                    if (w < 0) continue;
                    const uint64_t bit = cur.fw ? ((1ULL << b) << 4) : (1ULL << b);
                    if (!(shared[cur.u] & bit)) continue;                                                       // edge seen in enough reads
                    if (shared_count(U[cur.u].colours, U[u0].colours) < min_cov_vertices) continue;            // still read-compatible with the start
                    const KM y = ((endk << 2) | static_cast<KM>(b)) & mask;
                    const bool w_fw = (y == headk[static_cast<size_t>(w)]);
                    if (static_cast<size_t>(w) == u0 && w_fw) { // came back to the start unitig, same strand
                        bool distinct = true;
                        for (size_t i = 1; i < path.size() && distinct; ++i) for (size_t j = i + 1; j < path.size() && distinct; ++j) if (path[i].u == path[j].u && path[i].fw == path[j].fw) distinct = false;
                        if (!distinct) continue;
                        std::vector<uint32_t> pid = U[u0].colours;
                        for (size_t i = 1; i < path.size() && pid.size() >= min_cov_vertices; ++i) { std::vector<uint32_t> t; std::set_intersection(pid.begin(), pid.end(), U[path[i].u].colours.begin(), U[path[i].u].colours.end(), std::back_inserter(t)); pid.swap(t); }
                        if (pid.size() >= min_cov_vertices) { std::string c; for (size_t i = 1; i < path.size(); ++i) c.push_back(path[i].base); cycles[u0] += c; cycles[u0].push_back('\0'); }
                    } else {
                        size_t interior = 0; for (size_t i = 1; i < path.size(); ++i) interior += n_km(path[i].u);
                        if (interior + static_cast<size_t>(k) - 1 < 2 * static_cast<size_t>(k)) { // path.length() - um_start.len < 2k
                            std::vector<Step> nx = path; Step st; st.u = static_cast<size_t>(w); st.fw = w_fw; st.base = "ACGT"[b]; nx.push_back(st); q.push(nx);
                        }
                    }
                }
            }
        };
        // (the search of one unitig reads the edge bits of others: the short-cycle flags are set afterwards, not during the searches)
        if (fast) parallel_for(n, n_thr, [&](size_t b, size_t e, unsigned) { for (size_t u = b; u < e; ++u) cycles_of(u); });
        else for (size_t u0 = 0; u0 < n; ++u0) cycles_of(u0);
        for (size_t u0 = 0; u0 < n; ++u0) if (!cycles[u0].empty()) { shared[u0] |= 0x100ULL; ++n_cyc_unitigs; }
        fprintf(stderr, "rtk_build_index: %zu unitigs in short cycles\n", n_cyc_unitigs);
        lap("short cycles done");
    }

    // ---- SNP annotations: restatement of detectSNPs (src/Graph.cpp:484-720) with isValidSNPcandidate (src/GraphTraversal.cpp:1057-1147).
    // For every unitig with an edge bit: every graph k-mer ONE SUBSTITUTION away from one of its windows (searchSequence(seq, false,
    // false, false, true, false), [A2]) that lies on another unitig is a SNP candidate; the position gets the IUPAC union of its base
    // and the candidate's base when the other unitig passes isValidSNPcandidate: a breadth-first walk from this unitig, forwards and
    // backwards, over edges carrying an edge bit and unitigs sharing >= min_cov colours with this one, until a unitig shares >= min_cov
    // colours with the candidate (or 65536 unitigs were seen). The two walks keep their state from candidate to candidate, and a unitig
    // that answered one candidate is not expanded further -- reproduced as written. Candidates are visited by (window, substituted
    // offset, substituted base): Bifrost's own order inside one window is not known ([D3], canonical rule).
    std::vector<std::vector<uint32_t> > ambiguity(n);
    if (detect_snps) {
        std::vector<KM> headk(n), tailk(n);
        for (size_t u = 0; u < n; ++u) { km_encode<KM>(U[u].seq.c_str(), k, headk[u]); km_encode<KM>(U[u].seq.c_str() + U[u].seq.size() - k, k, tailk[u]); }
        struct Node { size_t u; bool fw; };
        // successors of (u, strand) in A,C,G,T order with the base that is appended
        auto successors = [&](const Node& x, Node out[4], int base[4]) -> int {
            int m = 0;
            const KM endk = x.fw ? tailk[x.u] : kmer_revcomp(headk[x.u], k);
            for (uint64_t b = 0; b < 4; ++b) {
                const int64_t w = adj[x.u].u[x.fw ? 0 : 1][b];
                if (w < 0) continue;
                const KM y = ((endk << 2) | static_cast<KM>(b)) & mask;
                out[m].u = static_cast<size_t>(w); out[m].fw = (y == headk[static_cast<size_t>(w)]); base[m] = static_cast<int>(b); ++m;
            }
            return m;
        };
        auto edge_bit = [&](const Node& x, int b) -> bool { return (shared[x.u] & (x.fw ? ((1ULL << b) << 4) : (1ULL << b))) != 0; };
        struct Walk { std::set<std::pair<size_t, bool> > seen; std::vector<size_t> seen_units; std::queue<Node> q; };
        const size_t limit_sz_stack = 65536;
        auto explore = [&](Walk& lgt, const Node& a, size_t ub) -> bool {
            if (U[a.u].colours.size() < min_cov_vertices || U[ub].colours.size() < min_cov_vertices) return false;
            if (lgt.seen.empty()) { lgt.q.push(a); lgt.seen.insert(std::make_pair(a.u, a.fw)); lgt.seen_units.push_back(a.u); }
            else if (lgt.seen.size() >= limit_sz_stack) return true;
            while (!lgt.q.empty()) {
                const Node x = lgt.q.front(); lgt.q.pop();
                Node nb[4]; int bs[4];
                const int m = successors(x, nb, bs);
                for (int i = 0; i < m; ++i) {
                    if (!edge_bit(x, bs[i])) continue;
                    if (!lgt.seen.insert(std::make_pair(nb[i].u, nb[i].fw)).second) continue; // visited (keyed by the mapped head k-mer: unitig + strand)
                    lgt.seen_units.push_back(nb[i].u);
                    if (shared_count(U[nb[i].u].colours, U[a.u].colours) >= min_cov_vertices) {
                        if (shared_count(U[nb[i].u].colours, U[ub].colours) >= min_cov_vertices) return true;
                        lgt.q.push(nb[i]);
                    }
                }
                if (lgt.seen.size() >= limit_sz_stack) return true;
            }
            return false;
        };
        auto is_valid = [&](Walk& fw, Walk& bw, size_t ua, size_t ub) -> bool {
            bool ok_fw = false, ok_bw = false;
            for (size_t i = 0; i < fw.seen_units.size() && !ok_fw; ++i) ok_fw = shared_count(U[fw.seen_units[i]].colours, U[ub].colours) >= min_cov_vertices;
            if (!ok_fw) { Node a; a.u = ua; a.fw = true; ok_fw = explore(fw, a, ub); }
            if (ok_fw) {
                for (size_t i = 0; i < bw.seen_units.size() && !ok_bw; ++i) ok_bw = shared_count(U[bw.seen_units[i]].colours, U[ub].colours) >= min_cov_vertices;
                if (!ok_bw) { Node a; a.u = ua; a.fw = false; ok_bw = explore(bw, a, ub); }
            }
            return ok_fw && ok_bw;
        };
        auto amb_bits = [](char c) -> unsigned { // getAmbiguityRev (src/Common.hpp:351-399): bit0 A, bit1 C, bit2 G, bit3 T
            switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; case 'M': return 3; case 'R': return 5; case 'S': return 6; case 'V': return 7;
                         case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12; case 'D': return 13; case 'B': return 14; case 'N': return 15; default: return 0; } };
        static const char amb_char[16] = {'.', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'}; // getAmbiguity
        std::unique_ptr<NeighbourIndex> nbx_own;
        if (fast) { nbx_own.reset(new NeighbourIndex()); nbx_own->build(solid64(solid), k, n_thr); lap("1-substitution neighbour index built"); }
        const NeighbourIndex* const nbx = nbx_own.get();
        auto annotate = [&](size_t u) {
            if (!(shared[u] & 0xffULL)) return; // hasSharedPids (src/Graph.cpp:500)
            const std::string& s = U[u].seq;
            std::string seq_final = s, seq_tried = s;
            std::set<size_t> ok, bad;
            Walk lgt_fw, lgt_bw;
            KM fw = 0;
            for (size_t i = 0; i < s.size(); ++i) {
                fw = ((fw << 2) | static_cast<KM>(base2bits(s[i]))) & mask;
                if (i + 1 < static_cast<size_t>(k)) continue;
                const size_t p = i + 1 - static_cast<size_t>(k);
                auto candidate = [&](int j, uint64_t alt) { // the graph holds the window with base `alt` at offset j
                    const int sh = 2 * (k - 1 - j);
                    const KM y = (fw & ~(static_cast<KM>(3) << sh)) | (static_cast<KM>(alt) << sh);
                    const uint64_t* v = km.slot(kmer_canonical(y, k), false);
                    if (!v) return;
                    const size_t w = (*v >> 32) - 1;
                    if (w == u) return; // a SNP candidate cannot be on the same unitig (src/Graph.cpp:523)
                    const size_t at = p + static_cast<size_t>(j); // pos_snp_km = first mismatch = the substituted offset
                    const unsigned f = amb_bits(seq_final[at]), t = amb_bits(seq_tried[at]), kk = 1u << alt;
                    const char cf = amb_char[f | kk], ct = amb_char[t | kk];
                    if (seq_tried[at] == ct) return; // that base was tried at this position before
                    seq_tried[at] = ct;
                    if (ok.count(w)) seq_final[at] = cf;
                    else if (!bad.count(w)) {
                        if (is_valid(lgt_fw, lgt_bw, u, w)) { seq_final[at] = cf; ok.insert(w); } else bad.insert(w);
                    }
                };
                if (nbx) nbx->neighbours(static_cast<uint64_t>(fw), candidate); // --fast: the neighbours from the two sorted views of the k-mer set, same order
                else for (int j = 0; j < k; ++j) {
                    const uint64_t cur = static_cast<uint64_t>(fw >> (2 * (k - 1 - j))) & 3ULL;
                    for (uint64_t alt = 0; alt < 4; ++alt) if (alt != cur) candidate(j, alt);
                }
            }
            for (size_t i = 0; i < seq_final.size(); ++i) if (seq_final[i] != 'A' && seq_final[i] != 'C' && seq_final[i] != 'G' && seq_final[i] != 'T') ambiguity[u].push_back(static_cast<uint32_t>((i << 4) + amb_bits(seq_final[i]))); // UnitigData.hpp:448-451
        };
        { // unitigs are independent and the k-mer table is only read: one strided slice per thread
            unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 1; if (nt > std::max(64u, n_thr)) nt = std::max(64u, n_thr);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t]() { for (size_t u = t; u < n; u += nt) annotate(u); });
            for (size_t t = 0; t < th.size(); ++t) th[t].join();
        }
        size_t n_amb = 0, n_amb_unitigs = 0;
        for (size_t u = 0; u < n; ++u) { n_amb += ambiguity[u].size(); n_amb_unitigs += ambiguity[u].empty() ? 0 : 1; }
        fprintf(stderr, "rtk_build_index: %zu SNP annotations on %zu unitigs\n", n_amb, n_amb_unitigs);
        lap("SNP annotations done");
    }

    // ---- global / local colour split (simplified restatement of src/Graph.cpp:2874-2985) ----
    std::vector<std::vector<uint32_t> > global_ids(n), local_ids(n);
    {
        double tot_cov = 0, tot_km = 0;
        for (size_t u = 0; u < n; ++u) { tot_cov += static_cast<double>(U[u].cov); tot_km += static_cast<double>(U[u].seq.size() - k + 1); }
        const double est_cov = tot_km > 0 ? tot_cov / tot_km : 0.0;
        auto kcov = [&](size_t u) { return static_cast<double>(static_cast<long long>(static_cast<double>(U[u].cov) / static_cast<double>(U[u].seq.size() - k + 1) + 0.5)); };
        std::vector<std::pair<double, size_t> > seeds;
        for (size_t u = 0; u < n; ++u) if ((kmcov[u] >> 63) && kcov(u) >= global_cov_factor * est_cov) seeds.push_back(std::make_pair(-kcov(u), u));
        std::sort(seeds.begin(), seeds.end());
        std::vector<char> visited(n, 0);
        for (size_t si = 0; si < seeds.size(); ++si) {
            const size_t u0 = seeds[si].second;
            if (visited[u0]) continue;
            std::vector<uint32_t> inter = U[u0].colours;
            size_t max_card_inter = static_cast<size_t>(static_cast<double>(inter.size()) * min_color_sharing);
            std::set<size_t> seen, valid; seen.insert(u0);
            std::queue<size_t> q; q.push(u0);
            while (!q.empty()) {
                const size_t x = q.front(); q.pop();
                std::vector<std::pair<double, size_t> > nbs;
                for (int d = 0; d < 2; ++d) for (int b = 0; b < 4; ++b) { const int64_t w = adj[x].u[d][b]; if (w >= 0 && seen.insert(static_cast<size_t>(w)).second && !visited[w]) nbs.push_back(std::make_pair(-kcov(static_cast<size_t>(w)), static_cast<size_t>(w))); }
                std::sort(nbs.begin(), nbs.end());
                for (size_t j = 0; j < nbs.size(); ++j) {
                    const size_t w = nbs[j].second;
                    std::vector<uint32_t> li;
                    std::set_intersection(inter.begin(), inter.end(), U[w].colours.begin(), U[w].colours.end(), std::back_inserter(li));
                    if (static_cast<double>(li.size()) >= static_cast<double>(U[w].colours.size()) * min_color_sharing && li.size() >= max_card_inter && !li.empty()) {
                        inter.swap(li);
                        max_card_inter = std::max(max_card_inter, static_cast<size_t>(static_cast<double>(U[w].colours.size()) * min_color_sharing));
                        valid.insert(w); q.push(w);
                    }
                }
            }
            if (!valid.empty()) {
                valid.insert(u0);
                for (std::set<size_t>::const_iterator it = valid.begin(); it != valid.end(); ++it) {
                    global_ids[*it] = inter; visited[*it] = 1;
                    std::set_difference(U[*it].colours.begin(), U[*it].colours.end(), inter.begin(), inter.end(), std::back_inserter(local_ids[*it]));
                }
            }
        }
        size_t ng = 0;
        for (size_t u = 0; u < n; ++u) { if (global_ids[u].empty()) local_ids[u] = U[u].colours; else ++ng; }
        fprintf(stderr, "rtk_build_index: est. k-mer coverage %.2f, %zu unitigs carry a global colour set\n", est_cov, ng);
        lap("global / local colour sets done");
    }

    // ---- write ----
    {
        if (fast) fasta_thread.join(); else write_fasta();
        if (fasta_rc) { fprintf(stderr, "rtk_build_index: cannot write fasta.gz\n"); return 1; }
        std::ofstream out((prefix + ".index.k" + std::to_string(k) + ".rtsk").c_str(), std::ios::binary);
        // records are encoded (Roaring containers of the colour sets) by ranges of unitigs on all threads and written in order: the same bytes
        const unsigned n_w = fast ? n_thr : 1u; const size_t per = (n + n_w - 1) / n_w;
        std::vector<std::string> part(n_w);
        parallel_for(n, n_w, [&](size_t b, size_t e, unsigned) {
            std::ostringstream os(std::ios::binary);
            for (size_t u = b; u < e; ++u) {
                RtskRecord r;
                disk_kmer_from_string(U[u].seq.c_str(), k, r.head);
                r.kmcov = kmcov[u]; r.shared = shared[u];
                r.global_ids = global_ids[u]; r.local_ids = local_ids[u]; r.ambiguity_ids = ambiguity[u]; r.cycles = cycles[u];
                rtsk_write_record(os, r);
            }
            part[per ? b / per : 0] = os.str();
        });
        for (unsigned t = 0; t < n_w; ++t) out.write(part[t].data(), static_cast<std::streamsize>(part[t].size()));
        if (!out.good()) { fprintf(stderr, "rtk_build_index: cannot write the .rtsk file\n"); return 1; }
    }
    lap("files written");
    return 0;
}

int main(int argc, char** argv) {
    int k = 31;
    for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], "-k")) k = atoi(argv[i + 1]);
    return k <= 31 ? run<uint64_t>(argc, argv) : run<u128>(argc, argv);
}
