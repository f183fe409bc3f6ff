// ORACLE (test infrastructure, never shipped, never on the product path).
//
// CPU restatement of the alignment results the reference obtains from its vendored edlib
// (reference: src/edlib.cpp:141-296 `edlibAlign`, :298-347 `edlibAlignmentToCigar`,
// :547-704 semi-global, :730-931 NW, :945-1144 traceback; src/Common.hpp:262-276 IUPAC equalities).
//
// It is NOT a copy of edlib: it computes the *unbanded* Myers/Hyyro bit-vector recurrence over whole
// columns and derives the same observable results:
//   editDistance   exact distance, or -1 when it exceeds a non-negative k (edlib.cpp:194-212,744-747)
//   endLocations   every target position whose last-row score equals the best score, ascending;
//                  SHW/HW additionally report position -1 (score = |query|) when |query| % 64 != 0,
//                  which is what edlib's padded last block yields (edlib.cpp:658-692, note at :232-244)
//   alignment      NW traceback between query and target[0..endLocations[0]] preferring
//                  up (EDLIB_EDOP_INSERT) > left (EDLIB_EDOP_DELETE) > diagonal (edlib.cpp:1021-1137)
// The Ukkonen band of edlib only prunes cells that cannot lie on an optimal path, so these results are
// band independent; tests/test_oracle_myers.py pins this file against the reference's own edlib.cpp
// compiled into oracle/_ref (parity pinned for this component).
// When 20*ceil(q/64)*t+8*t >= 2^20 edlib switches from traceback to its Hirschberg split (edlib.cpp:1191-1214,
// :1234-1399); pass 1 reaches it whenever a long non-terminal sub-path is given qualities
// (src/GraphTraversal.cpp:573 -> :722-730). The split rule is restated canonically (see obtain_alignment).
#ifndef RTK_ORACLE_MYERS_HPP
#define RTK_ORACLE_MYERS_HPP

#include <cstdint>
#include <string>
#include <vector>

namespace orc {

enum AlignMode { MODE_NW = 0, MODE_SHW = 1, MODE_HW = 2 };

struct AlignResult {
    int editDistance;               // -1 if > k
    std::vector<int> endLocations;  // empty if editDistance == -1
    std::vector<unsigned char> alignment; // 0 match, 1 insert (query only), 2 delete (target only), 3 mismatch
    AlignResult() : editDistance(-1) {}
};

bool iupac_equal(unsigned char a, unsigned char b);

// k < 0: unbounded.
// use_iupac=false restates edlibDefaultAlignConfig() (no additional equalities; src/Alignment.cpp:460).
AlignResult myers_align(const char* query, int qlen, const char* target, int tlen, int k, AlignMode mode, bool want_path, bool use_iupac = true);

std::string alignment_to_cigar(const std::vector<unsigned char>& aln); // standard CIGAR (M/I/D)

} // namespace orc

#endif
