// ORACLE (test infrastructure, never shipped, never on the product path).
//
// Minimal CPU model of the graph queries the reference's hot path makes into Bifrost
// (absent from /root/reference: un-vendored submodule `pmelsted/bifrost`, pinned commit unknown,
// API level >= v1.2 -- SURVEY.md §0.3, §8c). Every function cites the reference call site whose
// behaviour it stands for. Assumptions about Bifrost semantics are tagged [A1]..[A5]:
//   [A1] searchSequence exact: one (pos, UM{len=1}) per all-ACGT window whose canonical k-mer is in the graph.
//   [A2] searchSequence inexact: graph k-mers reachable from the read by ONE edit, reported at the read
//        position of the first read base they use:
//          substitution  read[p..p+k) with one base replaced;
//          "insertion"   k-1 read bases read[p..p+k-1) plus one inserted base (read lacks a base);
//          "deletion"    k+1 read bases read[p..p+k+1) minus one interior base (read has an extra base);
//        a window touching a non-ACGT character never matches.
//        The call passes or_exclusive_match = true (src/Graph.cpp:193: searchSequence(l_s, exact = false, insertion, deletion, substitution, or_exclusive_match)).
//        Three readings, switchable at run time (RTK_A2_XOR, read by oracle_seeds.cpp and by rtk_opts_default on the device side):
//          exclusive (default since round 4: what the flag's name says) = per window the kinds of edit are searched one after the other and a window one
//            kind has matched is not searched with the next: substitution, then insertion, then deletion (the order of the blocks of Bifrost's
//            published searchSequence as we remember it: a Roaring set of matched positions grows block by block);
//          exclusive-ids = the same with the kinds in the order of the function's parameters (insertion, deletion, substitution);
//          union = the hits of all three kinds (the default of rounds 1-3).
//        tests/test_a2_switch.py holds oracle and device to each other under every reading; profiles/r04_a2_count.json counts what they decide.
//   [A3] getSuccessors(): existing neighbours of the unitig end in walk direction, base order A,C,G,T,
//        as whole-unitig mappings (dist=0, len=size-k+1).
//        Switchable at run time (RTK_A3_ORDER=walk|strand; rtk_opts::a3_strand_order on the device side): walk (default) = by the base appended in
//        walk direction on both strands; strand = on the reverse strand by the base as the unitig's own strand spells it (T,G,C,A in walk
//        direction). The order only breaks ties between candidates of equal score (tests/test_a3_switch.py; counts in DESIGN_HISTORY.md section 5).
//   [A4] on-disk Kmer = 2 x u64, 2 bits/base (A0 C1 G2 T3), first base in the MSBs of word 0.
//   [A5] FASTA/FASTQ records: name = header up to the first whitespace.
//   [A6] KmerIterator visits the all-ACGT windows of a string in order; `it += n` moves to the first such window at or after
//        position(it) + n (src/Alignment.cpp:565,741,813 `it_km += um.len - 1`).
//   [A7] findUnitig(s, pos, len) = find(k-mer at s+pos) extended along the unitig while s keeps agreeing: forward strand
//        {dist = d, len = 1 + agreeing characters}; reverse strand the match runs towards the unitig head and
//        {dist = d - (len - 1)} (the mapping always starts at its lowest forward offset).
//   [A8] TinyBitmap::write payload behind a PairID flag-0 word (src/PairID.cpp:1158-1167): uint16 words, word 0 = size << 3 | mode,
//        word 1 = cardinality (words in use for the list modes), word 2 = the high 16 bits shared by all values, data from word 3;
//        mode 0 bitmap, 2 ascending list, 4 ascending (first, last) runs; empty = the single word 0. From Bifrost's published
//        TinyBitmap source; not verifiable here (no Bifrost, no reference-written .rtsk).
// "Parity unpinned" for everything that depends on [A1]-[A3]: the reference has no tests and its binary
// cannot be built here.
#ifndef RTK_ORACLE_GRAPH_HPP
#define RTK_ORACLE_GRAPH_HPP

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace orc {

typedef std::vector<uint32_t> IdSet; // sorted, unique pair ids ("colours")

struct UM { // restatement of Bifrost const_UnitigMap as used by the reference
    int32_t unitig;  // -1 == isEmpty
    uint32_t dist;   // 0-based k-mer offset on the forward unitig
    uint32_t len;    // number of k-mers mapped
    bool strand;
    UM() : unitig(-1), dist(0), len(0), strand(true) {}
    UM(int32_t u, uint32_t d, uint32_t l, bool s) : unitig(u), dist(d), len(l), strand(s) {}
    bool isEmpty() const { return unitig < 0; }
    bool operator==(const UM& o) const { return unitig == o.unitig && dist == o.dist && len == o.len && strand == o.strand; }
    bool operator!=(const UM& o) const { return !(*this == o); }
};

struct UnitigInfo { // restatement of the read side of src/UnitigData.hpp:258-491
    uint64_t kmcov;   // bit63 branching, bits31..61 unphased cov, bits0..30 phased cov (UnitigData.hpp:576)
    uint64_t shared;  // bit8 short cycle, bits4..7 fw edge mask, bits0..3 bw edge mask (UnitigData.hpp:577)
    int32_t global_id; // index into Graph::globals or -1 (SharedPairID global pointer)
    IdSet local;
    bool has_ambiguity;
    std::vector<std::pair<uint32_t, char> > amb; // get_ambiguity_char() (UnitigData.hpp:565-574): (position on the forward unitig, IUPAC code), sorted
    std::vector<std::string> cycles; // getCompactCycles() (UnitigData.hpp:312-327): successor-base strings of the short cycles through the unitig
    UnitigInfo() : kmcov(0), shared(0), global_id(-1), has_ambiguity(false) {}
};

// canonical k-mer (k <= 31: never all ones) -> value; open addressing, sized once from the number of k-mers of the unitigs. (A node-based
// std::unordered_map took minutes and several GB on the 60 Mb graphs of the bench / configs[2] tests; same contents, same answers.)
struct KmerMap64 {
    std::vector<uint64_t> keys, vals; size_t n, mask;
    KmerMap64() : n(0), mask(0) {}
    void clear() { keys.clear(); vals.clear(); n = 0; mask = 0; }
    void reserve(size_t want) { size_t cap = 16; while (cap < 2 * want + 16) cap <<= 1; keys.assign(cap, ~0ULL); vals.assign(cap, 0); n = 0; mask = cap - 1; }
    static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
    bool insert(uint64_t key, uint64_t val) { // false if the key is there already
        if (keys.empty()) reserve(1024);
        if (2 * (n + 1) > keys.size()) { KmerMap64 b; b.reserve(2 * n + 16); for (size_t i = 0; i < keys.size(); ++i) if (keys[i] != ~0ULL) b.insert(keys[i], vals[i]); keys.swap(b.keys); vals.swap(b.vals); mask = b.mask; }
        size_t i = mix(key) & mask;
        while (keys[i] != ~0ULL) { if (keys[i] == key) return false; i = (i + 1) & mask; }
        keys[i] = key; vals[i] = val; ++n; return true;
    }
    const uint64_t* find(uint64_t key) const { if (keys.empty()) return nullptr; size_t i = mix(key) & mask; while (keys[i] != ~0ULL) { if (keys[i] == key) return &vals[i]; i = (i + 1) & mask; } return nullptr; }
    size_t size() const { return n; }
};

struct Graph {
    int k;
    std::vector<std::string> seq;
    std::vector<UnitigInfo> info;
    std::vector<IdSet> globals;
    KmerMap64 kmap; // canonical k-mer -> unitig<<32 | offset<<1 | (stored orientation == canonical)
    std::unordered_map<std::string, uint64_t> kmap_w; // the same for k in 33..63 (second pass, k2 = 63), keyed by the canonical k-mer as TEXT

    // loads PREFIX unitig FASTA(.gz) + .rtsk (formats: SURVEY.md Appendix B). Throws std::runtime_error.
    void load(const std::string& fasta_gz, const std::string& rtsk, int k_);

    size_t usize(int32_t u) const { return seq[u].size(); }
    uint32_t nkm(int32_t u) const { return static_cast<uint32_t>(seq[u].size()) - k + 1; }

    // Bifrost find(km, extremities_only=false) on a k-mer given as text; empty UM if absent / non-ACGT. [A1]
    UM findKmer(const char* s) const;
    UM findKmerCode(uint64_t fw_code) const;

    // Bifrost findUnitig(s, pos, len): the k-mer at s+pos, extended along its unitig while the following characters of s agree. [A7]
    UM findUnitig(const char* s, size_t pos, size_t len) const;
    // UnitigData::get_ambiguity_char(um) (UnitigData.hpp:458-481): annotations inside the mapping, in mapping coordinates and orientation
    std::vector<std::pair<size_t, char> > ambiguityChars(const UM& um) const;

    std::string mapped(const UM& um) const;                 // const_UnitigMap::mappedSequenceToString
    bool sameUnitig(const UM& a, const UM& b) const { return a.unitig == b.unitig; } // isSameReferenceUnitig
    void successors(const UM& um, UM out[4], char base[4], int& n) const; // getSuccessors() [A3]
    int nbSuccessors(const UM& um) const;

    // UnitigData accessors
    bool isBranching(int32_t u) const { return (info[u].kmcov >> 63) & 1ULL; }
    bool isShortCycle(int32_t u) const { return (info[u].shared >> 8) & 1ULL; }
    bool hasSharedPids(int32_t u) const { return (info[u].shared & 0xffULL) != 0; }
    bool getSharedPids(int32_t u, bool strand, char c) const; // UnitigData.hpp:275-284
    double kmerCoverage(int32_t u) const;                     // UnitigData.hpp:396-399
    size_t cardinality(int32_t u) const;                      // SharedPairID::cardinality
    IdSet allIds(int32_t u) const;                            // SharedPairID::toPairID
    const IdSet* globalSet(int32_t u) const { return info[u].global_id >= 0 ? &globals[info[u].global_id] : nullptr; }
    size_t sharedCount(int32_t u, const IdSet& b) const;      // getNumberSharedPairID(SharedPairID, PairID) exact
    size_t sharedCount(int32_t a, int32_t b) const;           // getNumberSharedPairID(SharedPairID, SharedPairID) exact

    size_t maxKmerCoverage(double top_ratio) const;           // src/Graph.cpp:825-841
};

// sorted-set algebra standing in for PairID operators (src/PairID.cpp)
IdSet set_union(const IdSet& a, const IdSet& b);
IdSet set_inter(const IdSet& a, const IdSet& b);
IdSet set_diff(const IdSet& a, const IdSet& b);
size_t set_inter_card(const IdSet& a, const IdSet& b);

std::string revcomp(const std::string& s);

// IUPAC helpers (src/Common.hpp:260,351-400; Bifrost isDNA / reverse_complement(char))
uint8_t iupacIndex(char c);   // bit0 A, bit1 C, bit2 G, bit3 T; 0 for anything outside the table
char iupacChar(uint8_t idx);  // ".ACMGRSVTWYHKDBN"[idx]
char iupacComplement(char c);
inline bool isDNA(char c) { const char u = static_cast<char>(c & 0xDF); return u == 'A' || u == 'C' || u == 'G' || u == 'T'; }

} // namespace orc

#endif
