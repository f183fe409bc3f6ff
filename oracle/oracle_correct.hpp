// ORACLE (test infrastructure, never shipped, never on the product path).
//
// CPU restatement of the reference's per-long-read correction for pass 1 (`Ratatosk correct -1`,
// long_read_correct=false, hap_id undetermined): getSeeds (src/Graph.cpp:3-482), keep_non_overlap
// (src/Alignment.cpp:1017-1199), correctSequence (src/Correction.cpp:159-958), extractSemiWeakPaths
// (src/Correction.cpp:3-157), explorePathsBFS/BFS2/exploreSubGraph/getScorePath (src/GraphTraversal.cpp),
// selectBest*Alignment / generateConsensus (src/Alignment.cpp), Path (src/Path.hpp), ResultCorrection.
//
// PARITY UNPINNED for everything except the alignment results (oracle_myers.*): the reference ships no
// tests and cannot be built here (Bifrost absent). Canonical determinism rules replacing the reference's
// address-dependent orders:
//   [D1] chooseColors anchor order = (colour-set cardinality asc, unitig id asc)   (src/Correction.cpp:286-293)
//   [D2] extractSemiWeakPaths path grouping = STABLE sort by mapped string of the last unitig (src/Correction.cpp:52)
// fixRepeats on short-cycle unitigs (src/GraphTraversal.cpp:1149-1334) is restated; its inputs come from the restatement of
// detectShortCycles in ratatosk_amd/csrc/tools/build_index.cpp.
// fixAmbiguity/getAmbiguityVector on SNP-annotated unitigs (src/Alignment.cpp:527-844, src/GraphTraversal.cpp:966-1036) are
// restated for an undetermined haplotype (no phasing input); they lean on Bifrost's findUnitig and KmerIterator, assumptions
// [A6]/[A7] of oracle_graph.hpp. Test inputs come from `build_index --snps` (a simplified stand-in for detectSNPs).
// Pass 2 (long_read_correct): restated incl. phasing() and exploreSubGraphLong; leans on one more assumption, [A9] = the wyhash
// version behind TinyBloomFilter (oracle_pass2.cpp). Not restated: haplotype input (-p/-P, hap ids).
#ifndef RTK_ORACLE_CORRECT_HPP
#define RTK_ORACLE_CORRECT_HPP

#include <string>
#include <utility>
#include <vector>

#include "oracle_graph.hpp"

namespace orc {

struct Opt { // the Correct_Opt fields the pass-1 hot path reads (src/Common.hpp:101-156)
    size_t insert_sz;
    size_t min_cov_vertices;
    size_t max_len_weak_region1;
    double weak_region_len_factor;
    double large_k_factor;
    double min_score;
    int max_qual;
    int out_qual;
    double min_confidence_snp_corr; // -m (src/Common.hpp:147)
    size_t max_km_cov; // = max(getMaxKmerCoverage(dbg, 0.001), 128) (src/Ratatosk.cpp:625)
    // pass 2 (`correct -2`, long_read_correct == true in the reference): graph coloured by the pass-1 reads, qualities carried over
    bool long_read_correct;
    size_t max_len_weak_region2; // -W, 5000 (src/Common.hpp:110)
    bool skip_phasing;           // test switch (not in the reference): pass 2 without the phasing() pre-filter, to check the rest on its own
    bool force_unres_snp_corr;   // -f (src/Ratatosk.cpp:279): fixSNPs() on the corrected read before phasing()
    Opt() : insert_sz(500), min_cov_vertices(2), max_len_weak_region1(1000), weak_region_len_factor(0.25), large_k_factor(1.5),
            min_score(0.0), max_qual(40), out_qual(1), min_confidence_snp_corr(0.9), max_km_cov(128), long_read_correct(false), max_len_weak_region2(5000), skip_phasing(false), force_unres_snp_corr(false) {}
};

struct Counters { // event counts feeding the algorithmic-bytes model of SURVEY.md §8(d)
    uint64_t n_probe, n_verify, n_expand, n_colour_elem, n_path_base, n_align, n_align_cells, n_regions;
    Counters() : n_probe(0), n_verify(0), n_expand(0), n_colour_elem(0), n_path_base(0), n_align(0), n_align_cells(0), n_regions(0) {}
    void add(const Counters& o) { n_probe += o.n_probe; n_verify += o.n_verify; n_expand += o.n_expand; n_colour_elem += o.n_colour_elem; n_path_base += o.n_path_base; n_align += o.n_align; n_align_cells += o.n_align_cells; n_regions += o.n_regions; }
};

typedef std::pair<size_t, UM> Anchor;

// Optional: alignments through the REFERENCE's own edlib (oracle/_ref/libedlib_ref.so, ref_edlib_moves of ref_edlib_shim.cpp) instead
// of oracle_myers.cpp. Same results (oracle_myers.cpp is pinned against it); used by bench.py's cpu_baseline so that the CPU leg
// carries edlib's banded Myers, and as a cross-check in tests.
typedef int (*ref_edlib_moves_fn)(const char*, int, const char*, int, int, int, int, int, int*, int*, int, unsigned char*, int, int*);
extern ref_edlib_moves_fn g_ref_edlib_moves;

// src/Graph.cpp:3-482. Returns (solid, weak); opt.long_read_correct: exact hits only (:100).
std::pair<std::vector<Anchor>, std::vector<Anchor> > getSeeds(const Graph& g, const Opt& opt, const std::string& s, Counters* cnt = nullptr);

// individual stages of getSeeds, exposed so the device stages can be checked one by one
std::vector<Anchor> searchExact(const Graph& g, const std::string& s, Counters* cnt = nullptr);           // [A1]
std::string maskForInexact(const Graph& g, const Opt& opt, const std::string& s, const std::vector<Anchor>& exact_sorted); // src/Graph.cpp:102-191
std::vector<Anchor> searchInexact(const Graph& g, const std::string& masked, Counters* cnt = nullptr);   // [A2]
std::vector<Anchor> keepNonOverlap(const Graph& g, const char* ref, size_t ref_len, const std::vector<Anchor>& v); // src/Alignment.cpp:1017-1199

// src/Correction.cpp:159-958 (pass 1). Returns (sequence, quality).
std::pair<std::string, std::string> correctSequence(const Graph& g, const Opt& opt, const std::string& s_fw, const std::string& q_fw,
                                                    const std::vector<Anchor>& solid, const std::vector<Anchor>& weak, Counters* cnt = nullptr);

// per-read body of the worker loop (src/Ratatosk.cpp:808-864): upper-case, clamp qualities, seeds, correct.
std::pair<std::string, std::string> correctRead(const Graph& g, const Opt& opt, std::string seq, std::string qual, Counters* cnt = nullptr);

// ---- pass 2 (oracle_pass2.cpp) ----
// phasing() (src/Graph.cpp:869-1097): reverts the pass-1 corrections that sit on unitigs whose colours (pass-1 read ids) are not
// shared with any unitig further than insert_sz away on the same read; TinyBloomFilter (src/TinyBloomFilter.hpp) + wyhash [A9].
std::pair<std::string, std::string> phasing(const Graph& g, const Opt& opt, const std::string& s_raw, const std::string& s_corr, const std::string& q_corr);
// fixSNPs() (src/Alignment.cpp:846-965), `-f`: ambiguous characters of a read with exactly one base that has graph support
std::string fixSNPs(const Graph& g, const std::string& s);
// per-read body for long_read_correct == true (src/Ratatosk.cpp:808-838): upper-case, phasing, getSeeds (exact hits only), correctSequence
std::pair<std::string, std::string> correctRead2(const Graph& g, const Opt& opt, std::string seq_corr, std::string qual_corr, const std::string& seq_raw, Counters* cnt = nullptr);
// writeCorrectedOutput with trimming (src/Ratatosk.cpp:510-563): the records one corrected read becomes (name suffixes as the reference writes them)
std::vector<std::pair<std::string, std::pair<std::string, std::string> > > trimRecords(const std::string& name, const std::string& seq, const std::string& qual, size_t k, int trim_qual);
uint64_t wyhash8(uint64_t key, uint64_t seed); // wyhash(&key, 8, seed, _wyp) [A9]

} // namespace orc

#endif
