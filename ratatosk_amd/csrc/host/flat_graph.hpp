// Host image of the flat graph (product code). Built from the reference's on-disk index
// (reference: src/Ratatosk.cpp:1087-1089 dbg.read + readGraphData; formats in SURVEY.md Appendix B).
#ifndef RTK_FLAT_GRAPH_HPP
#define RTK_FLAT_GRAPH_HPP

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../hip/rtk_types.h"

namespace rtk {

// The three biggest buffers (k-mer table, half-k-mer index: tens of GB each for a whole-genome graph) are filled by all loader threads:
// a std::vector would write every word once more on one thread (value initialisation) before the first useful store.
struct BigWords {
    BigWords() : p_(nullptr), n_(0) {}
    ~BigWords() { delete[] p_; }
    BigWords(const BigWords&) = delete; BigWords& operator=(const BigWords&) = delete;
    size_t size() const { return n_; }
    uint64_t* data() { return p_; } const uint64_t* data() const { return p_; }
    uint64_t& operator[](size_t i) { return p_[i]; } const uint64_t& operator[](size_t i) const { return p_[i]; }
    void alloc_uninitialised(size_t n) { delete[] p_; p_ = nullptr; n_ = 0; p_ = new uint64_t[n ? n : 1]; n_ = n; } // (default-initialised: the pages are not touched here)
    void assign(size_t n, uint64_t v) { alloc_uninitialised(n); for (size_t i = 0; i < n; ++i) p_[i] = v; }    // small cases; the loader fills big ones itself
private:
    uint64_t* p_; size_t n_;
};

struct FlatGraph {
    int k = 31;
    uint64_t n_kmers = 0, n_global = 0;
    BigWords ht;                                                  // k-mer table (GraphView::ht)
    std::vector<uint64_t> useq, uoff, loff, goff, bf, cycoff; // cycoff[u]..cycoff[u+1]: bytes of unitig u's compact cycles in `cyc`
    std::vector<uint64_t> bf1;                                    // first-level presence bits
    BigWords hx, hxl;                                             // half-k-mer index (GraphView::hx / hxl)
    std::vector<uint64_t> amb;                                    // SNP annotations (GraphView::amb): n+1 offsets, then the entries
    std::vector<uint64_t> hap;                                    // haplotype ids of each unitig (UnitigData::hap_ids, src/UnitigData.hpp:493-517): n+1 offsets, then the ids. Only the
                                                                  // phased-input options (-p/-P, out of scope) read them; kept so that a reference-written index loads without loss
    std::vector<uint64_t> cyc;                                    // NUL-terminated successor-base strings, packed 8 characters per word
    std::vector<uint32_t> adj, flags, kcov, card, col;
    std::vector<int32_t> gid;
    uint64_t max_km_cov_top = 0; // getMaxKmerCoverage(dbg, 0.001) (reference: src/Graph.cpp:825-841)

    uint32_t n_unitigs() const { return static_cast<uint32_t>(flags.size()); }
    GraphView view() const; // pointers into the host vectors
    uint64_t bytes() const;
    // throws std::runtime_error. defer_tables: the lookup structures (k-mer table and its two filters, half-k-mer index, adjacency) are left
    // empty -- rtk_graph_upload builds them on the device from the packed unitigs (hip/rtk_graph_tables.hip); tables_deferred says so.
    void load(const std::string& fasta_gz, const std::string& rtsk, int k_, int n_threads, bool defer_tables = false);
    bool tables_deferred = false;
};

// sizes of the lookup structures of a graph (one policy for the host and the device builders; the environment knobs are read here)
struct TableSizes {
    uint64_t ht_slots;   // 16-byte slots of the k-mer table
    uint64_t bf_words;   // words of the blocked Bloom filter (a power of two)
    uint64_t bf1_words;  // words of the first-level bit array (a power of two), 1 = disabled (a single all-ones word)
    bool hx;             // half-k-mer index built (else one empty slot: the 1-edit search spells the variants)
    int h;               // its h = (k - 1) / 2
};
TableSizes table_sizes(int k, uint64_t n_kmers, uint64_t n_bases);

// the flat buffers in a fixed order (upload / RCCL broadcast order)
enum { RTK_BUF_USEQ = 0, RTK_BUF_UOFF, RTK_BUF_ADJ, RTK_BUF_FLAGS, RTK_BUF_KCOV, RTK_BUF_CARD, RTK_BUF_LOFF, RTK_BUF_GID, RTK_BUF_GOFF, RTK_BUF_COL, RTK_BUF_HT, RTK_BUF_BF, RTK_BUF_CYCOFF, RTK_BUF_CYC, RTK_BUF_BF1, RTK_BUF_AMB, RTK_BUF_HX, RTK_BUF_HXL, RTK_BUF_HAP, RTK_N_BUFS };

} // namespace rtk

#endif
