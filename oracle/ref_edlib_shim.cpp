// ORACLE support (test infrastructure): flat C wrapper around the REFERENCE's own edlib so python/ctypes
// can call it without marshalling structs by value. Compiled together with
// /root/reference/src/edlib.cpp (taken where it lies; never copied) into oracle/_ref/libedlib_ref.so by
// oracle/Makefile. Only used to pin oracle_myers.cpp and to generate tests/golden/edlib_golden.tsv.
#include <cstdlib>
#include <cstring>

#include "edlib.h" // from -I/root/reference/src

static const EdlibEqualityPair kIupac[28] = { // the equality table the reference passes at every call site (src/Common.hpp:262-274)
    {'M','A'},{'M','C'},{'R','A'},{'R','G'},{'S','C'},{'S','G'},{'V','A'},{'V','C'},{'V','G'},{'W','A'},{'W','T'},
    {'Y','C'},{'Y','T'},{'H','A'},{'H','C'},{'H','T'},{'K','G'},{'K','T'},{'D','A'},{'D','G'},{'D','T'},
    {'B','C'},{'B','G'},{'B','T'},{'N','A'},{'N','C'},{'N','G'},{'N','T'}};

extern "C" int ref_edlib(const char* q, int qlen, const char* t, int tlen, int k, int mode /*0 NW,1 SHW,2 HW*/, int want_path, int use_iupac,
                         int* n_loc, int* end_locs, int cap_locs, char* cigar, int cap_cigar) {
    const EdlibAlignMode m = mode == 0 ? EDLIB_MODE_NW : (mode == 1 ? EDLIB_MODE_SHW : EDLIB_MODE_HW);
    EdlibAlignConfig cfg = edlibNewAlignConfig(k, m, want_path ? EDLIB_TASK_PATH : EDLIB_TASK_DISTANCE, use_iupac ? kIupac : NULL, use_iupac ? 28 : 0);
    EdlibAlignResult r = edlibAlign(q, qlen, t, tlen, cfg);
    const int d = r.editDistance;
    *n_loc = r.numLocations;
    for (int i = 0; i < r.numLocations && i < cap_locs; ++i) end_locs[i] = r.endLocations[i];
    if (cigar && cap_cigar > 0) {
        cigar[0] = 0;
        if (want_path && d >= 0) {
            char* c = edlibAlignmentToCigar(r.alignment, r.alignmentLength, EDLIB_CIGAR_STANDARD);
            if (c) { strncpy(cigar, c, cap_cigar - 1); cigar[cap_cigar - 1] = 0; free(c); }
        }
    }
    edlibFreeAlignResult(r);
    return d;
}

// Same call, results as edlib reports them: every end location and the raw alignment operations (EDLIB_EDOP_MATCH 0, INSERT 1,
// DELETE 2, MISMATCH 3). Lets the oracle's correction run on the REFERENCE's alignment layer (bench.py cpu_baseline, cross-checks).
extern "C" int ref_edlib_moves(const char* q, int qlen, const char* t, int tlen, int k, int mode, int want_path, int use_iupac,
                               int* n_loc, int* end_locs, int cap_locs, unsigned char* moves, int cap_moves, int* n_moves) {
    const EdlibAlignMode m = mode == 0 ? EDLIB_MODE_NW : (mode == 1 ? EDLIB_MODE_SHW : EDLIB_MODE_HW);
    EdlibAlignConfig cfg = edlibNewAlignConfig(k, m, want_path ? EDLIB_TASK_PATH : EDLIB_TASK_DISTANCE, use_iupac ? kIupac : NULL, use_iupac ? 28 : 0);
    EdlibAlignResult r = edlibAlign(q, qlen, t, tlen, cfg);
    const int d = r.editDistance;
    *n_loc = r.numLocations; *n_moves = 0;
    for (int i = 0; i < r.numLocations && i < cap_locs; ++i) end_locs[i] = r.endLocations[i];
    if (want_path && d >= 0 && r.alignment) { *n_moves = r.alignmentLength; for (int i = 0; i < r.alignmentLength && i < cap_moves; ++i) moves[i] = r.alignment[i]; }
    edlibFreeAlignResult(r);
    return d;
}
