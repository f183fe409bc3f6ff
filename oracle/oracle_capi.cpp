// ORACLE (test infrastructure): flat C entry points so tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive the CPU restatement through ctypes. Nothing in the product links this.
#include <dlfcn.h>
#include <malloc.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "oracle_correct.hpp"
#include "oracle_myers.hpp"

using namespace orc;

extern "C" {

struct orc_opts { // mirrors the pass-1 fields of Correct_Opt (reference: src/Common.hpp:101-156)
    uint64_t insert_sz, min_cov_vertices, max_len_weak_region1, max_km_cov;
    double weak_region_len_factor, large_k_factor, min_score;
    int32_t max_qual, out_qual;
    double min_confidence_snp_corr;
    uint64_t max_len_weak_region2; // pass 2 (-W, 5000)
    int32_t skip_phasing, force_unres_snp_corr; // test switch, see Opt::skip_phasing; -f
};

static Opt toOpt(const orc_opts* o) {
    Opt r;
    if (o) { r.insert_sz = o->insert_sz; r.min_cov_vertices = o->min_cov_vertices; r.max_len_weak_region1 = o->max_len_weak_region1; r.max_km_cov = o->max_km_cov;
             r.weak_region_len_factor = o->weak_region_len_factor; r.large_k_factor = o->large_k_factor; r.min_score = o->min_score; r.max_qual = o->max_qual; r.out_qual = o->out_qual; r.min_confidence_snp_corr = o->min_confidence_snp_corr; if (o->max_len_weak_region2) r.max_len_weak_region2 = o->max_len_weak_region2; r.skip_phasing = o->skip_phasing != 0; r.force_unres_snp_corr = o->force_unres_snp_corr != 0; }
    return r;
}

void* orc_graph_load(const char* fasta_gz, const char* rtsk, int k, char* err, int errcap) {
    Graph* g = new Graph();
    try { g->load(fasta_gz, rtsk, k); }
    catch (const std::exception& e) { if (err && errcap > 0) { strncpy(err, e.what(), errcap - 1); err[errcap - 1] = 0; } delete g; return nullptr; }
    return g;
}

void orc_graph_free(void* g) { delete static_cast<Graph*>(g); }

void orc_graph_stats(void* gp, uint64_t* n_unitigs, uint64_t* n_kmers, uint64_t* max_km_cov_001) {
    const Graph* g = static_cast<const Graph*>(gp);
    *n_unitigs = g->seq.size(); *n_kmers = g->kmap.size() + g->kmap_w.size(); *max_km_cov_001 = g->maxKmerCoverage(0.001);
}

// unitig record: sequence + the two data words + colour sets, for cross-checking the product's flat graph
int orc_unitig(void* gp, uint64_t u, char* seq, uint64_t seq_cap, uint64_t* seq_len, uint64_t* kmcov, uint64_t* shared, int64_t* global_id,
               uint32_t* local, uint64_t local_cap, uint64_t* n_local) {
    const Graph* g = static_cast<const Graph*>(gp);
    if (u >= g->seq.size()) return -1;
    *seq_len = g->seq[u].size();
    if (seq && seq_cap >= g->seq[u].size()) memcpy(seq, g->seq[u].data(), g->seq[u].size());
    *kmcov = g->info[u].kmcov; *shared = g->info[u].shared; *global_id = g->info[u].global_id;
    *n_local = g->info[u].local.size();
    for (size_t i = 0; i < g->info[u].local.size() && i < local_cap; ++i) local[i] = g->info[u].local[i];
    return 0;
}

int orc_global_set(void* gp, int64_t id, uint32_t* out, uint64_t cap, uint64_t* n) {
    const Graph* g = static_cast<const Graph*>(gp);
    if (id < 0 || static_cast<size_t>(id) >= g->globals.size()) return -1;
    *n = g->globals[id].size();
    for (size_t i = 0; i < g->globals[id].size() && i < cap; ++i) out[i] = g->globals[id][i];
    return 0;
}

// annotations of unitig u as the index holds them: SNP annotations (position, IUPAC code; UnitigData.hpp:557-574) and the compact
// cycles (UnitigData.hpp:312-327) as one byte string of NUL-terminated successor-base strings. For tests/test_annotators.py, which
// recomputes both from the graph (oracle/oracle_annot.py) and compares.
int orc_unitig_annotations(void* gp, uint64_t u, uint32_t* amb_pos, char* amb_code, uint64_t amb_cap, uint64_t* n_amb,
                           char* cyc, uint64_t cyc_cap, uint64_t* n_cyc_bytes) {
    const Graph* g = static_cast<const Graph*>(gp);
    if (u >= g->seq.size()) return -1;
    const UnitigInfo& d = g->info[u];
    *n_amb = d.amb.size();
    for (size_t i = 0; i < d.amb.size() && i < amb_cap; ++i) { amb_pos[i] = d.amb[i].first; amb_code[i] = d.amb[i].second; }
    std::string all;
    for (size_t i = 0; i < d.cycles.size(); ++i) { all += d.cycles[i]; all.push_back('\0'); }
    *n_cyc_bytes = all.size();
    if (cyc && cyc_cap >= all.size()) memcpy(cyc, all.data(), all.size());
    return 0;
}

// neighbours of unitig u walking fw (dir 0) / on the reverse strand (dir 1): out[b] = unitig<<1|strand or -1, b in A,C,G,T
void orc_neighbours(void* gp, uint64_t u, int dir, int64_t out[4]) {
    const Graph* g = static_cast<const Graph*>(gp);
    UM um(static_cast<int32_t>(u), 0, g->nkm(static_cast<int32_t>(u)), dir == 0);
    UM s[4]; char b[4]; int n;
    g->successors(um, s, b, n);
    for (int i = 0; i < 4; ++i) out[i] = -1;
    for (int i = 0; i < n; ++i) { const int bi = b[i] == 'A' ? 0 : (b[i] == 'C' ? 1 : (b[i] == 'G' ? 2 : 3)); out[bi] = (static_cast<int64_t>(s[i].unitig) << 1) | (s[i].strand ? 1 : 0); }
}

int orc_myers(const char* q, int qlen, const char* t, int tlen, int k, int mode, int want_path, int use_iupac,
              int* n_loc, int* end_locs, int cap_locs, char* cigar, int cap_cigar) {
    const AlignResult a = myers_align(q, qlen, t, tlen, k, static_cast<AlignMode>(mode), want_path != 0, use_iupac != 0);
    *n_loc = static_cast<int>(a.endLocations.size());
    for (size_t i = 0; i < a.endLocations.size() && static_cast<int>(i) < cap_locs; ++i) end_locs[i] = a.endLocations[i];
    if (cigar && cap_cigar > 0) { const std::string c = want_path ? alignment_to_cigar(a.alignment) : std::string(); strncpy(cigar, c.c_str(), cap_cigar - 1); cigar[cap_cigar - 1] = 0; }
    return a.editDistance;
}

// per-window exact hits [A1]: out[p] = unitig<<33 | dist<<1 | strand, or -1
void orc_exact(void* gp, const char* seq, uint64_t len, int64_t* out) {
    const Graph* g = static_cast<const Graph*>(gp);
    const std::string s(seq, len);
    const size_t nw = len >= static_cast<size_t>(g->k) ? len - g->k + 1 : 0;
    for (size_t i = 0; i < nw; ++i) out[i] = -1;
    const std::vector<Anchor> v = searchExact(*g, s, nullptr);
    for (size_t i = 0; i < v.size(); ++i) out[v[i].first] = (static_cast<int64_t>(v[i].second.unitig) << 33) | (static_cast<int64_t>(v[i].second.dist) << 1) | (v[i].second.strand ? 1 : 0);
}

static void packAnchors(const std::vector<Anchor>& v, int64_t* out, uint64_t cap) {
    for (size_t i = 0; i < v.size() && i < cap; ++i) { out[4 * i] = static_cast<int64_t>(v[i].first); out[4 * i + 1] = v[i].second.unitig; out[4 * i + 2] = v[i].second.dist; out[4 * i + 3] = v[i].second.strand ? 1 : 0; }
}

// full getSeeds: anchors as (pos, unitig, dist, strand) quadruples
int orc_seeds(void* gp, const orc_opts* o, const char* seq, uint64_t len, uint64_t* n_solid, int64_t* solid, uint64_t* n_weak, int64_t* weak, uint64_t cap) {
    const Graph* g = static_cast<const Graph*>(gp);
    const Opt opt = toOpt(o);
    const std::pair<std::vector<Anchor>, std::vector<Anchor> > p = getSeeds(*g, opt, std::string(seq, len), nullptr);
    *n_solid = p.first.size(); *n_weak = p.second.size();
    packAnchors(p.first, solid, cap); packAnchors(p.second, weak, cap);
    return (p.first.size() <= cap && p.second.size() <= cap) ? 0 : 1;
}

// mask + raw inexact hits (before sort/dedup/filters), for checking the device inexact-lookup stage on its own
int orc_inexact(void* gp, const orc_opts* o, const char* seq, uint64_t len, char* masked_out, uint64_t* n_hits, int64_t* hits, uint64_t cap) {
    const Graph* g = static_cast<const Graph*>(gp);
    const Opt opt = toOpt(o);
    const std::string s(seq, len);
    std::vector<Anchor> ex = searchExact(*g, s, nullptr); // already position-sorted, one hit per position
    const std::string m = maskForInexact(*g, opt, s, ex);
    if (masked_out) memcpy(masked_out, m.data(), m.size());
    const std::vector<Anchor> v = searchInexact(*g, m, nullptr);
    *n_hits = v.size();
    packAnchors(v, hits, cap);
    return v.size() <= cap ? 0 : 1;
}

// The per-read body of the reference's worker loop over a batch, N threads pulling read tickets
// (reference: src/Ratatosk.cpp:727-904). Outputs are malloc'd; free with orc_free.
int orc_correct_batch(void* gp, const orc_opts* o, uint64_t n, const char* const* seq, const char* const* qual, const uint32_t* len,
                      char** out_seq, char** out_qual, uint32_t* out_len, int n_threads, uint64_t* counters /*8*/) {
    const Graph* g = static_cast<const Graph*>(gp);
    const Opt opt = toOpt(o);
    // (host-side scaling of the CPU leg of bench.py, round 5: the per-thread counters live on the threads' own stacks -- as neighbours in one vector they shared
    // cache lines, every count a line ping-pong between cores; reads are handed out longest first so that no thread starts a 60 kb read when the others are done;
    // big temporaries stay in the heap instead of one mmap + munmap each, which serialises the threads of a process in the kernel)
    static const bool malloc_tuned = [] { mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); return true; }(); (void)malloc_tuned;
    std::atomic<uint64_t> ticket(0);
    std::vector<Counters> cs(n_threads > 0 ? n_threads : 1);
    std::vector<uint64_t> order(n); for (uint64_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return len[x] > len[y]; });
    auto work = [&](int t) {
        Counters mine;
        while (true) {
            const uint64_t at = ticket.fetch_add(1);
            if (at >= n) break;
            const uint64_t i = order[at];
            const std::pair<std::string, std::string> r = correctRead(*g, opt, std::string(seq[i], len[i]), qual && qual[i] ? std::string(qual[i], len[i]) : std::string(), &mine);
            out_len[i] = static_cast<uint32_t>(r.first.size());
            out_seq[i] = static_cast<char*>(malloc(r.first.size() + 1)); memcpy(out_seq[i], r.first.c_str(), r.first.size() + 1);
            out_qual[i] = static_cast<char*>(malloc(r.second.size() + 1)); memcpy(out_qual[i], r.second.c_str(), r.second.size() + 1);
        }
        cs[t] = mine;
    };
    if (n_threads <= 1) work(0);
    else { std::vector<std::thread> th; for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t); for (size_t t = 0; t < th.size(); ++t) th[t].join(); }
    if (counters) {
        Counters tot; for (size_t t = 0; t < cs.size(); ++t) tot.add(cs[t]);
        counters[0] = tot.n_probe; counters[1] = tot.n_verify; counters[2] = tot.n_expand; counters[3] = tot.n_colour_elem;
        counters[4] = tot.n_path_base; counters[5] = tot.n_align; counters[6] = tot.n_align_cells; counters[7] = tot.n_regions;
    }
    return 0;
}

// Pass 2 (`correct -2`): seq / qual = the pass-1 corrected reads, raw = the uncorrected reads in the same order (src/Ratatosk.cpp:774-838).
int orc_correct_batch2(void* gp, const orc_opts* o, uint64_t n, const char* const* seq, const char* const* qual, const uint32_t* len,
                       const char* const* raw, const uint32_t* raw_len, char** out_seq, char** out_qual, uint32_t* out_len, int n_threads) {
    const Graph* g = static_cast<const Graph*>(gp);
    const Opt opt = toOpt(o);
    std::atomic<uint64_t> ticket(0);
    auto work = [&]() {
        while (true) {
            const uint64_t i = ticket.fetch_add(1);
            if (i >= n) break;
            const std::pair<std::string, std::string> r = correctRead2(*g, opt, std::string(seq[i], len[i]), qual && qual[i] ? std::string(qual[i], len[i]) : std::string(), std::string(raw[i], raw_len[i]), nullptr);
            out_len[i] = static_cast<uint32_t>(r.first.size());
            out_seq[i] = static_cast<char*>(malloc(r.first.size() + 1)); memcpy(out_seq[i], r.first.c_str(), r.first.size() + 1);
            out_qual[i] = static_cast<char*>(malloc(r.second.size() + 1)); memcpy(out_qual[i], r.second.c_str(), r.second.size() + 1);
        }
    };
    if (n_threads <= 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < n_threads; ++t) th.emplace_back(work); for (size_t t = 0; t < th.size(); ++t) th[t].join(); }
    return 0;
}

// fixSNPs() of one read; out has room for len characters (the length does not change)
void orc_fix_snps(void* gp, const char* seq, uint64_t len, char* out) { const std::string r = fixSNPs(*static_cast<Graph*>(gp), std::string(seq, len)); memcpy(out, r.data(), r.size()); }

uint64_t orc_wyhash8(uint64_t key, uint64_t seed) { return wyhash8(key, seed); }

void orc_free(void* p) { free(p); }

// path == NULL: back to the restatement. Returns 0 on success, -1 when the library or the symbol is missing.
int orc_use_reference_edlib(const char* so_path) {
    if (!so_path) { g_ref_edlib_moves = nullptr; return 0; }
    void* h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    void* f = dlsym(h, "ref_edlib_moves");
    if (!f) return -1;
    g_ref_edlib_moves = reinterpret_cast<ref_edlib_moves_fn>(f);
    return 0;
}

} // extern "C"
