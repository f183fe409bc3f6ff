// Region stage, one weak region per LANE (64 regions per wavefront): the program of rtk_region.h -- the `correct` lambda of
// src/Correction.cpp:431-753 with chooseColors :215-429, extractSemiWeakPaths :3-157, explorePathsBFS2 src/GraphTraversal.cpp:212-454,
// exploreSubGraph :456-587, getScorePath :722-772 / :867-909, selectBest*Alignment src/Alignment.cpp:3-147 / :967-1015, fixAmbiguity
// :527-844, generateConsensus :309-470, edlibAlign src/edlib.cpp:586-677 / :945-1144 -- restated for ONE lane that owns ONE gap between two solid
// anchors. The wave-per-region kernel keeps 3-10 % of its lanes busy in the small alignments of such a region and pays ~700 dependent memory
// round trips per region with one region in flight per wave (DESIGN_HISTORY.md section 3.5); here a wave has 64 regions in flight, every lane runs the
// whole program on its own compact records, and the lanes meet in the same loops (the Myers column sweep, the 2-bit decode, the set walks).
//
// Rules of this file:
//  * no cross-lane operation anywhere: a lane's code is a sequential program, so the 1-lane host simulator (tests/hostsim) runs exactly
//    what a lane runs on the device; nothing here goes through rtk_u / rtk_ld / U<> descriptors that are not launch-uniform;
//  * per-lane state lives in a work area interleaved by lane (word i of lane l at base[i * 64 + l]: lanes that touch the same logical
//    word -- they mostly do, they run the same loops -- fetch one contiguous 256-byte row), sized for the light class of regions
//    (gaps under RTK_LANE_MAX_GAP bases); the match vectors of the alignment in LDS; the traceback table (Pv / Mv of every column, 16 bytes
//    per word and column) in a second interleaved area;
//  * anything this program does not hold -- a capacity that runs out, a short-cycle unitig (fixRepeats), a region without end anchor, a target
//    character outside A C G T N -- sets a status code and the region is handed to the wave kernel (k_regions, rd->status != 0): same results
//    either way, the share of handed-on regions is counted and reported.
#ifndef RTK_REGION_LANE_H
#define RTK_REGION_LANE_H

#include "rtk_region.h"

#define RL_STRIDE RTK_WAVE
// layout of a wave's work area and traceback table in device memory: RL_MS = 1: one contiguous slice per lane (a lane's walk over its own words stays in the
// cache lines it has fetched: the lane programs are bound by the latency of their own dependent accesses, not by bandwidth); RL_MS = RL_STRIDE: words interleaved
// by lane (the lanes' accesses to the same logical word coalesce; -DRL_INTERLEAVED, measured slower)
#ifdef RL_INTERLEAVED
#define RL_MS RL_STRIDE
#define RL_LANE_WORDS(words) 1ull
#else
#define RL_MS 1
#define RL_LANE_WORDS(words) static_cast<uint64_t>(words)
#endif

// ---- capacities (compile-time layout; RlCtx::lim_* are the run-time limits the checks use: a test hook lowers them) ----
#ifndef RL_STR_BYTES
#define RL_STR_BYTES 1024u           // every string buffer
#endif
#define RL_STR_W (RL_STR_BYTES / 4u)
#define RL_NSTR 12u
#define RL_MV_BYTES (2u * RL_STR_BYTES) // a move list
#define RL_MV_W (RL_MV_BYTES / 4u)
#define RL_NMV 3u
#define RL_UM_CAP 48u                // unitigs of a path
#define RL_WP_W (4u + 3u * RL_UM_CAP + RL_STR_W)
#define RL_NWP 4u
#define RL_A0_W 1024u                // arena of the region level (words)
#define RL_A1_W 3072u                // BFS level
#define RL_A2_W 3072u                // DFS level
#define RL_LIST_CAP 32u
#define RL_STK_CAP 64u
#define RL_MEMO_CAP 96u
#define RL_SIDE_CAP 24u              // side-list slots, the three sides together
#define RL_AMB_CAP 48u
#define RL_BM_W 32u                  // position bitmaps: 1024 bits
#define RL_ALL_CAP 1024u             // ids of all_pids / of the colour universe
#define RL_MAXW 8                    // 64-bit words of an alignment's query
#define RL_NSYM 5                    // target characters A C T G N
#ifndef RL_TB_WORDCOLS
#define RL_TB_WORDCOLS 3072u         // word-columns of a lane's traceback table (16 bytes each)
#endif

// string buffers
#define RL_SB_PATH 0u   // path string (to_string)
#define RL_SB_CAND 1u   // candidate string of the DFS
#define RL_SB_QUAL 2u   // quality scratch
#define RL_SB_TMP 3u    // temporary (reverse complement, query_tmp, q_sub)
#define RL_SB_FWS 4u
#define RL_SB_FWQ 5u
#define RL_SB_BWS 6u
#define RL_SB_BWQ 7u
#define RL_SB_OUTS 8u
#define RL_SB_OUTQ 9u
#define RL_SB_CS 10u
#define RL_SB_CQ 11u

// word offsets of the work area
#define RL_OFF_STR 0u
#define RL_OFF_MV (RL_OFF_STR + RL_NSTR * RL_STR_W)
#define RL_OFF_WP (RL_OFF_MV + RL_NMV * RL_MV_W)
#define RL_OFF_A0 (RL_OFF_WP + RL_NWP * RL_WP_W)
#define RL_OFF_A1 (RL_OFF_A0 + RL_A0_W)
#define RL_OFF_A2 (RL_OFF_A1 + RL_A1_W)
#define RL_OFF_T (RL_OFF_A2 + RL_A2_W)           // handle lists
#define RL_OFF_NT (RL_OFF_T + RL_LIST_CAP)
#define RL_OFF_V (RL_OFF_NT + RL_LIST_CAP)
#define RL_OFF_VT (RL_OFF_V + RL_LIST_CAP)
#define RL_OFF_TC (RL_OFF_VT + RL_LIST_CAP)      // terminal candidates of a DFS call
#define RL_OFF_STK (RL_OFF_TC + RL_LIST_CAP)     // (handle, level) pairs
#define RL_OFF_MEMO (RL_OFF_STK + 2u * RL_STK_CAP)
#define RL_OFF_SIDE (RL_OFF_MEMO + RL_MEMO_CAP)  // unitig << 2 | side of entry ... see rl_side_*
#define RL_OFF_AMB (RL_OFF_SIDE + RL_SIDE_CAP + 8u) // five lists of RL_AMB_CAP entries (position << 8 | character)
#define RL_OFF_BM (RL_OFF_AMB + 5u * RL_AMB_CAP) // fw, bw, tmp
#define RL_OFF_ALL (RL_OFF_BM + 3u * RL_BM_W)
#define RL_WORDS (RL_OFF_ALL + RL_ALL_CAP)
// the colour selection runs before the path search: its universe, bit rows and vectors overlay the BFS / DFS arenas
#define RL_CS_VW 32u                              // 32-bit words of a bit vector (RL_ALL_CAP bits)
#define RL_OFF_CS_UA RL_OFF_A1
#define RL_OFF_CS_UB (RL_OFF_CS_UA + RL_ALL_CAP)
#define RL_OFF_CS_ROWS (RL_OFF_CS_UB + RL_ALL_CAP)                 // slot x {local, global} x RL_CS_VW
#define RL_OFF_CS_VEC (RL_OFF_CS_ROWS + RL_SIDE_CAP * 2u * RL_CS_VW) // vectors
#define RL_CS_NVEC 24u
#if (RL_OFF_CS_VEC + RL_CS_NVEC * RL_CS_VW) > (RL_OFF_A2 + RL_A2_W)
#error "colour selection does not fit the arenas it overlays"
#endif

RTK_HD uint64_t rl_area_bytes() { return static_cast<uint64_t>(RL_WORDS) * 4ull * RL_STRIDE; }             // per wave
RTK_HD uint64_t rl_table_bytes() { return static_cast<uint64_t>(RL_TB_WORDCOLS) * 16ull * RL_STRIDE; }      // per wave

// status codes of a region the lane program hands on (rd->status; the wave kernel redoes it)
#define RL_F_STR 1u      // a string buffer
#define RL_F_UM 2u       // unitigs of a path
#define RL_F_ARENA 3u    // a path arena
#define RL_F_LIST 4u     // a handle list / the DFS stack
#define RL_F_SIDE 5u     // side lists / slots of the colour selection
#define RL_F_IDS 6u      // ids of the colour universe / all_pids
#define RL_F_ALIGN 7u    // query longer than RL_MAXW words, target character outside A C G T N, traceback table
#define RL_F_AMB 8u      // SNP-annotation lists
#define RL_F_REPEAT 9u   // a path through a short-cycle unitig (fixRepeats: wave kernel)
#define RL_F_NOEND 10u   // a search without end anchor (explorePathsBFS: wave kernel)
#define RL_F_BM 11u      // region longer than the position bitmaps
#define RL_F_OTHER 12u

// lap profile of a lane (developer build -DRTK_LANE_PROF): slots
#define RL_P_DRIVER 0
#define RL_P_SIDE 1
#define RL_P_COLOURS 2
#define RL_P_SEARCH 3     // extractSemiWeakPaths / explorePathsBFS2 / explore glue
#define RL_P_DFS 4        // the walk outside the calls below
#define RL_P_COLOUR_OK 5
#define RL_P_TOSTRING 6
#define RL_P_PEQ 7
#define RL_P_SWEEP 8
#define RL_P_WALK 9
#define RL_P_RECORDS 10   // commit / load / extend / merge of path records
#define RL_P_QUAL 11
#define RL_P_SELECT 12
#define RL_P_AMB 13
#define RL_P_FIXAMB 14
#define RL_P_ASSEMBLE 15
#define RL_P_TRIM 16
#define RL_P_REVCOMP 17
#define RL_P_CONSENSUS 18
#define RL_P_EMIT 19
#define RL_P_IDLE 20      // waiting at the end of a round for the slowest lane
#define RL_NPROF 24
#ifdef RTK_LANE_PROF
// (prof2: the same time divided by the number of lanes that arrive at the lap together, x 64: summed over the lanes it estimates the WAVE's time in the slot)
#define RL_LAP(c, slot) do { const unsigned long long t_ = rtk_clock(); const unsigned long long d_ = t_ - (c).prof_t; (c).prof[slot] += d_; (c).prof2[slot] += d_ * 64ull / static_cast<unsigned long long>(rtk_popc(rtk_ballot(true))); (c).prof_t = t_; } while (0)
// a leaf: what ran since the last lap belongs to the context that called; what runs until RL_LEAVE to the leaf
#define RL_ENTER(c) RL_LAP(c, (c).prof_cur)
#define RL_LEAVE(c, slot) RL_LAP(c, slot)
// a context: the same, and the calls inside it (other than leaves) are attributed to `slot` until RL_CTX_END restores the caller's
#define RL_CTX(c, slot) const uint32_t prof_prev_ = (c).prof_cur; RL_LAP(c, prof_prev_); (c).prof_cur = (slot)
#define RL_CTX_END(c) do { RL_LAP(c, (c).prof_cur); (c).prof_cur = prof_prev_; } while (0)
#define RL_SETCUR(c, slot) ((c).prof_cur = (slot))
#else
#define RL_LAP(c, slot) ((void)0)
#define RL_ENTER(c) ((void)0)
#define RL_LEAVE(c, slot) ((void)0)
#define RL_CTX(c, slot) ((void)0)
#define RL_CTX_END(c) ((void)0)
#define RL_SETCUR(c, slot) ((void)0)
#endif

// The context of a lane. What every step of the program reads -- the views of the launch, the bases of the wave's work areas, the capacities -- is wave-uniform
// and lives in LDS (rl_env), the few mutable control words of a lane (status, arena tops, sizes) in per-lane LDS slots (rl_lst): neither can alias the
// work area in device memory, so the compiler keeps them in registers across its stores (a context struct handed by reference to the non-inlined
// programs lived on the lane's stack, and every work-area store forced every field to be read again: a private-memory round trip per access).
struct RlEnv {
    const GraphView* g; const OptsView* o; const BatchView* bv; const RegionBatch* rb;
    uint32_t* m;      // word 0 of the wave's work area (lane l: + l)
    uint64_t* tb;     // slot 0 of the wave's traceback table
    uint32_t k, lim_str, lim_um, lim_list, lim_tb, lim_arena[3];
};
#define RL_S_FAIL 0
#define RL_S_TOP0 1
#define RL_S_MEMO 4
#define RL_S_NALL 5
#define RL_S_N 6
#ifdef RTK_SIM
extern thread_local RlEnv rl_env;
extern thread_local uint32_t rl_lst[RL_S_N * RL_STRIDE];
extern thread_local uint64_t rl_peq_all[RL_NSYM * RL_MAXW * RL_STRIDE];
#else
__shared__ RlEnv rl_env;
__shared__ uint32_t rl_lst[RL_S_N * RL_STRIDE];
__shared__ uint64_t rl_peq_all[RL_NSYM * RL_MAXW * RL_STRIDE]; // match vectors: word (s * RL_MAXW + w) of lane l at [(s * RL_MAXW + w) * RL_STRIDE + l]
#endif
struct RlCtx {
    RTK_DEV const GraphView* g() const { return rl_env.g; }
    RTK_DEV const OptsView* o() const { return rl_env.o; }
    RTK_DEV const BatchView* bv() const { return rl_env.bv; }
    RTK_DEV const RegionBatch* rb() const { return rl_env.rb; }
    RTK_DEV uint32_t* m() const { return rtk_gp(rl_env.m) + static_cast<uint64_t>(rtk_lane()) * RL_LANE_WORDS(RL_WORDS); }
    RTK_DEV uint64_t* tb() const { return rtk_gp(rl_env.tb) + static_cast<uint64_t>(rtk_lane()) * RL_LANE_WORDS(RL_TB_WORDCOLS * 2u); }
    RTK_DEV uint64_t* peq() const { return rl_peq_all + rtk_lane(); }
    RTK_DEV uint32_t k() const { return rl_env.k; }
    RTK_DEV uint32_t lim_str() const { return rl_env.lim_str; }
    RTK_DEV uint32_t lim_um() const { return rl_env.lim_um; }
    RTK_DEV uint32_t lim_list() const { return rl_env.lim_list; }
    RTK_DEV uint32_t lim_tb() const { return rl_env.lim_tb; }
    RTK_DEV uint32_t lim_arena(int i) const { return rl_env.lim_arena[i]; }
    RTK_DEV uint32_t& fail() const { return rl_lst[RL_S_FAIL * RL_STRIDE + rtk_lane()]; }
    RTK_DEV uint32_t& top(int i) const { return rl_lst[(RL_S_TOP0 + i) * RL_STRIDE + rtk_lane()]; }
    RTK_DEV uint32_t& memo_n() const { return rl_lst[RL_S_MEMO * RL_STRIDE + rtk_lane()]; }
    RTK_DEV uint32_t& n_all() const { return rl_lst[RL_S_NALL * RL_STRIDE + rtk_lane()]; }
    uint32_t sv_valid, sv_m, sv_n, sv_W; int32_t sv_nw, sv_best, sv_first; // the stored sweep of the DFS's first terminal candidate
    uint32_t c_expand, c_colour, c_pathbase, c_align; unsigned long long c_cells;
#ifdef RTK_LANE_PROF
    unsigned long long prof[RL_NPROF], prof2[RL_NPROF], prof_t; uint32_t prof_cur; // developer build: every cycle of a lane attributed to one slot (RL_P_*)
#endif
};

RTK_DEV void rl_fail(RlCtx& c, uint32_t code) { uint32_t& f = c.fail(); if (!f) f = code; }

// ------------------------------------------------------------------------------------------------ work-area access
#ifdef RTK_SIM
extern std::atomic<unsigned long long> rl_sim_acc[4]; // developer statistics (simulator): word loads, word stores, byte loads, byte stores of the lanes' work areas
#define RL_ACC(i) (rl_sim_acc[i] += 1)
#else
#define RL_ACC(i) ((void)0)
#endif
RTK_DEV uint32_t rl_ld(const RlCtx& c, uint32_t w) { RL_ACC(0); return c.m()[static_cast<uint64_t>(w) * RL_MS]; }
RTK_DEV void rl_st(const RlCtx& c, uint32_t w, uint32_t v) { RL_ACC(1); c.m()[static_cast<uint64_t>(w) * RL_MS] = v; }
// byte b of the work area (b = 4 * word + byte)
RTK_DEV unsigned char rl_ldb(const RlCtx& c, uint32_t b) { RL_ACC(2); return reinterpret_cast<const unsigned char*>(c.m() + static_cast<uint64_t>(b >> 2) * RL_MS)[b & 3u]; }
RTK_DEV void rl_stb(const RlCtx& c, uint32_t b, unsigned char v) { RL_ACC(3); reinterpret_cast<unsigned char*>(c.m() + static_cast<uint64_t>(b >> 2) * RL_MS)[b & 3u] = v; }
RTK_DEV uint32_t rl_sb(uint32_t i) { return 4u * (RL_OFF_STR + i * RL_STR_W); }   // byte offset of string buffer i
RTK_DEV uint32_t rl_mvb(uint32_t i) { return 4u * (RL_OFF_MV + i * RL_MV_W); }    // byte offset of move buffer i

// a character sequence: contiguous device memory (the reads) or bytes of the lane's work area
struct RlSrc { const char* g; uint32_t b; };
RTK_DEV RlSrc rl_src_g(const char* p) { RlSrc s; s.g = p; s.b = 0; return s; }
RTK_DEV RlSrc rl_src_l(uint32_t byte_off) { RlSrc s; s.g = nullptr; s.b = byte_off; return s; }
RTK_DEV RlSrc rl_src_add(const RlSrc& s, uint32_t d) { RlSrc r; r.g = s.g ? s.g + d : nullptr; r.b = s.b + d; return r; }
RTK_DEV unsigned char rl_get(const RlCtx& c, const RlSrc& s, uint32_t i) { return s.g ? static_cast<unsigned char>(s.g[i]) : rl_ldb(c, s.b + i); }

// appending writer on work-area bytes: one word store per four characters
struct RlW { uint32_t b0, len, acc; }; // byte offset of the buffer, characters so far, the open word
RTK_DEV RlW rl_w_open(const RlCtx& c, uint32_t b0, uint32_t len) { RlW w; w.b0 = b0; w.len = len; w.acc = (len & 3u) ? (rl_ld(c, (b0 + len) >> 2) & ((1u << (8u * (len & 3u))) - 1u)) : 0u; return w; }
RTK_DEV void rl_w_put(const RlCtx& c, RlW& w, unsigned char ch) {
    w.acc |= static_cast<uint32_t>(ch) << (8u * (w.len & 3u)); ++w.len;
    if ((w.len & 3u) == 0u) { rl_st(c, (w.b0 + w.len - 4u) >> 2, w.acc); w.acc = 0u; }
}
RTK_DEV void rl_w_close(const RlCtx& c, RlW& w) { if (w.len & 3u) rl_st(c, (w.b0 + w.len) >> 2, w.acc); }
// (buffers start on word boundaries: b0 is a multiple of 4)

// four characters of a sequence from character i on, first character in the low byte (what lies behind the end of a sequence is read, never used: the
// read buffers are padded by 64 bytes, a string buffer is followed by the next one)
RTK_DEV uint32_t rl_get4(const RlCtx& c, const RlSrc& s, uint32_t i) {
    if (s.g) { uint32_t v; __builtin_memcpy(&v, s.g + i, 4); return v; }
    const uint32_t b = s.b + i, sh = 8u * (b & 3u);
    const uint32_t w0 = rl_ld(c, b >> 2);
    if (sh == 0u) return w0;
    return (w0 >> sh) | (rl_ld(c, (b >> 2) + 1u) << (32u - sh));
}
// bit j = byte j of x equals ch (j < 4)
RTK_DEV uint32_t rl_eq4(uint32_t x, uint32_t ch) {
    const uint32_t y = x ^ (ch * 0x01010101u);
    const uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
    return (((z >> 7) * 0x00204081u) >> 21) & 0xFu;
}

// dst[dl ..) += src[0 .. n); returns the new length (or fails). Bytes up to the next word of dst, then whole words, then the rest.
RTK_DEV uint32_t rl_app(RlCtx& c, uint32_t dst_b, uint32_t dl, const RlSrc& src, uint32_t n) {
    if (dl + n > c.lim_str()) { rl_fail(c, RL_F_STR); return dl; }
    RlW w = rl_w_open(c, dst_b, dl);
    uint32_t i = 0;
    for (; i < n && (w.len & 3u); ++i) rl_w_put(c, w, rl_get(c, src, i));
    for (; i + 4u <= n; i += 4u) { rl_st(c, (dst_b + w.len) >> 2, rl_get4(c, src, i)); w.len += 4u; }
    for (; i < n; ++i) rl_w_put(c, w, rl_get(c, src, i));
    rl_w_close(c, w);
    return dl + n;
}
RTK_DEV uint32_t rl_app_fill(RlCtx& c, uint32_t dst_b, uint32_t dl, char ch, uint32_t n) {
    if (dl + n > c.lim_str()) { rl_fail(c, RL_F_STR); return dl; }
    RlW w = rl_w_open(c, dst_b, dl);
    uint32_t i = 0;
    const uint32_t ch4 = static_cast<uint32_t>(static_cast<unsigned char>(ch)) * 0x01010101u;
    for (; i < n && (w.len & 3u); ++i) rl_w_put(c, w, static_cast<unsigned char>(ch));
    for (; i + 4u <= n; i += 4u) { rl_st(c, (dst_b + w.len) >> 2, ch4); w.len += 4u; }
    for (; i < n; ++i) rl_w_put(c, w, static_cast<unsigned char>(ch));
    rl_w_close(c, w);
    return dl + n;
}
RTK_DEV void rl_copy_words(const RlCtx& c, uint32_t dst_w, uint32_t src_w, uint32_t n) { // (four loads in flight, then their stores)
    uint32_t i = 0;
    for (; i + 4u <= n; i += 4u) { const uint32_t a = rl_ld(c, src_w + i), b = rl_ld(c, src_w + i + 1u), d = rl_ld(c, src_w + i + 2u), e = rl_ld(c, src_w + i + 3u); rl_st(c, dst_w + i, a); rl_st(c, dst_w + i + 1u, b); rl_st(c, dst_w + i + 2u, d); rl_st(c, dst_w + i + 3u, e); }
    for (; i < n; ++i) rl_st(c, dst_w + i, rl_ld(c, src_w + i));
}

// ------------------------------------------------------------------------------------------------ graph
RTK_DEV uint32_t rl_ulen(const RlCtx& c, uint32_t u) { const uint64_t* uo = c.g()->uoff.get() + u; return static_cast<uint32_t>(uo[1] - uo[0]); }
RTK_DEV uint32_t rl_nkm(const RlCtx& c, uint32_t u) { return rl_ulen(c, u) - c.k() + 1u; }
RTK_DEV uint32_t rl_flags(const RlCtx& c, uint32_t u) { return c.g()->flags.get()[u]; }

// ------------------------------------------------------------------------------------------------ alignment (src/edlib.cpp:586-677 block recurrence, :161-179, :744-747, :945-1144 walk)
// symbols of the target: 0 A, 1 C, 2 T, 3 G ((ch >> 1) & 3), 4 N; 5 = not a character of this program
RTK_DEV int rl_sym(uint32_t ch) { const uint32_t s = (ch >> 1) & 3u; return (ch == ((0x47544341u >> (8u * s)) & 0xFFu)) ? static_cast<int>(s) : (ch == 'N' ? 4 : 5); }

struct RlAln { int32_t dist, first, last; };

// Distance and end locations of edlibAlign(q, t, k, mode) with the IUPAC equalities (iupac) or plain equality; store: keep Pv / Mv of every
// word and column in the lane's table (for rl_myers_walk). The recurrence of NW and SHW is the same (the top row counts a gap), so a stored
// SHW sweep also yields the NW distance of the whole target (*nw_all).
RTK_FN RlAln rl_myers(RlCtx& c, RlSrc q, int m, RlSrc t, int n, int k, int mode, bool iupac, bool store, int32_t* nw_all) {
    RlAln r; r.dist = -1; r.first = -1; r.last = -1;
    if (nw_all) *nw_all = -1;
    RL_ENTER(c);
    c.c_align += 1; c.c_cells += static_cast<unsigned long long>((m + 63) / 64) * static_cast<unsigned long long>(n);
    if (m == 0 || n == 0) { // edlib.cpp:161-179
        if (mode == RTK_MODE_NW) { r.dist = m > n ? m : n; r.first = r.last = n - 1; } else { r.dist = m; r.first = r.last = -1; }
        if (nw_all) *nw_all = m > n ? m : n;
        return r;
    }
    if (m > 64 * RL_MAXW) { rl_fail(c, RL_F_ALIGN); return r; }
    if (mode == RTK_MODE_NW && k >= 0 && k < (n > m ? n - m : m - n)) return r; // edlib.cpp:744-747
    const int W = (m + 63) >> 6, last_bit = (m - 1) & 63;
    if (store && static_cast<uint32_t>(W) * static_cast<uint32_t>(n) > c.lim_tb()) { rl_fail(c, RL_F_ALIGN); return r; }
    uint64_t* const peq = c.peq();
    { // match vectors of the five target characters, four query characters per step
        uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
        for (int i = 0; i < m; i += 4) {
            const uint32_t x = rl_get4(c, q, static_cast<uint32_t>(i));
            const int sh = i & 63; // (four characters never straddle a word of the query: i is a multiple of 4)
            const uint32_t live = (m - i) >= 4 ? 0xFu : ((1u << (m - i)) - 1u);
            const uint32_t a4 = rl_eq4(x, 'A') & live, c4 = rl_eq4(x, 'C') & live, t4 = rl_eq4(x, 'T') & live, g4 = rl_eq4(x, 'G') & live;
            const uint32_t base4 = a4 | c4 | t4 | g4;
            e0 |= static_cast<uint64_t>(a4) << sh; e1 |= static_cast<uint64_t>(c4) << sh; e2 |= static_cast<uint64_t>(t4) << sh; e3 |= static_cast<uint64_t>(g4) << sh;
            if (iupac) e4 |= static_cast<uint64_t>(base4) << sh; // a base equals a target N under the IUPAC equalities
            for (uint32_t other = live & ~base4; other; other &= other - 1u) { // a code (a merged SNP, an N of the read): character by character
                const int b = __builtin_ctz(other); const unsigned char qc = static_cast<unsigned char>(x >> (8 * b)); const uint64_t bit = 1ull << (sh + b);
                if (rtk_chars_equal(qc, 'A', iupac)) e0 |= bit; if (rtk_chars_equal(qc, 'C', iupac)) e1 |= bit; if (rtk_chars_equal(qc, 'T', iupac)) e2 |= bit;
                if (rtk_chars_equal(qc, 'G', iupac)) e3 |= bit; if (rtk_chars_equal(qc, 'N', iupac)) e4 |= bit;
            }
            if (sh == 60 || i + 4 >= m) {
                const uint64_t w = static_cast<uint64_t>(i >> 6);
                peq[(0 * RL_MAXW + w) * RL_STRIDE] = e0; peq[(1 * RL_MAXW + w) * RL_STRIDE] = e1; peq[(2 * RL_MAXW + w) * RL_STRIDE] = e2;
                peq[(3 * RL_MAXW + w) * RL_STRIDE] = e3; peq[(4 * RL_MAXW + w) * RL_STRIDE] = e4;
                e0 = e1 = e2 = e3 = e4 = 0;
            }
        }
    }
    RL_LEAVE(c, RL_P_PEQ);
    uint64_t Pv[RL_MAXW], Mv[RL_MAXW];
#pragma unroll
    for (int w = 0; w < RL_MAXW; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
    int score = m, best = 0x7fffffff, first = -2, last = -2;
    const int top_h = (mode == RTK_MODE_HW) ? 0 : 1;
    const bool every_column = mode != RTK_MODE_NW;
    uint64_t* const tb = c.tb();
    uint32_t tw = 0;
    for (int j = 0; j < n; ++j) {
        if ((j & 3) == 0) tw = rl_get4(c, t, static_cast<uint32_t>(j));
        const int s = rl_sym(tw & 0xFFu); tw >>= 8;
        if (s > 4) { rl_fail(c, RL_F_ALIGN); return r; }
        const uint64_t* const e = peq + static_cast<uint64_t>(s * RL_MAXW) * RL_STRIDE;
        int hin = top_h;
#pragma unroll
        for (int w = 0; w < RL_MAXW; ++w) if (w < W) { // one word of the block recurrence
            uint64_t Eq = e[static_cast<uint64_t>(w) * RL_STRIDE];
            const uint64_t pv = Pv[w], mv = Mv[w];
            const uint64_t Xv = Eq | mv;
            if (hin < 0) Eq |= 1ull;
            const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv), Mh = pv & Xh;
            const int bit = (w == W - 1) ? last_bit : 63;
            const int hout = static_cast<int>((Ph >> bit) & 1ull) - static_cast<int>((Mh >> bit) & 1ull);
            Ph <<= 1; Mh <<= 1;
            if (hin > 0) Ph |= 1ull; else if (hin < 0) Mh |= 1ull;
            Pv[w] = Mh | ~(Xv | Ph); Mv[w] = Ph & Xv;
            if (store) { uint64_t* const ent = tb + static_cast<uint64_t>(j * W + w) * 2ull * RL_MS; ent[0] = Pv[w]; ent[RL_MS] = Mv[w]; }
            hin = hout;
        }
        score += hin;
        if (every_column) { if (score < best) { best = score; first = j; last = j; } else if (score == best) last = j; }
    }
    RL_LEAVE(c, RL_P_SWEEP);
    if (nw_all) *nw_all = score;
    if (mode == RTK_MODE_NW) {
        if (k >= 0 && score > k) return r;
        r.dist = score; r.first = r.last = n - 1;
        return r;
    }
    if ((m & 63) != 0) { // edlib's padded last block exposes target position -1 with score m
        if (m < best) { best = m; first = -1; last = -1; }
        else if (m == best) first = -1;
    }
    if (k >= 0 && best > k) return r;
    r.dist = best; r.first = first; r.last = last;
    return r;
}

// The path of the stored sweep: from cell (m, n_cols) back to the origin, up before left before diagonal (edlib.cpp:1021-1137). Cell values come
// from the table alone: D(i, j) = j + (set bits of Pv_j below row i) - (set bits of Mv_j below row i). Moves (0 match, 1 insertion = query character
// alone, 2 deletion, 3 mismatch) are written last first at the END of move buffer `mv_b` (byte offset); returns their number, *off = where they start.
RTK_FN uint32_t rl_myers_walk(RlCtx& c, int m, int n_cols, uint32_t mv_b, uint32_t* off) {
    const int W = (m + 63) >> 6;
    uint32_t o = RL_MV_BYTES;
    *off = o;
    RL_ENTER(c);
    if (static_cast<uint32_t>(m + n_cols) > RL_MV_BYTES || static_cast<uint32_t>(m + n_cols) > 2u * c.lim_str()) { rl_fail(c, RL_F_STR); return 0; }
    const uint64_t* const tb = c.tb();
    auto cell = [&](int i, int j) -> int { // D(i, j), j >= 1: column j - 1 of the table
        const uint64_t* const col = tb + static_cast<uint64_t>((j - 1) * W) * 2ull * RL_MS;
        int v = j;
        const int fw = i >> 6, rb = i & 63;
        for (int w = 0; w < fw; ++w) v += rtk_popc(col[static_cast<uint64_t>(w) * 2ull * RL_MS]) - rtk_popc(col[(static_cast<uint64_t>(w) * 2ull + 1ull) * RL_MS]);
        if (rb) { const uint64_t mk = (1ull << rb) - 1ull; v += rtk_popc(col[static_cast<uint64_t>(fw) * 2ull * RL_MS] & mk) - rtk_popc(col[(static_cast<uint64_t>(fw) * 2ull + 1ull) * RL_MS] & mk); }
        return v;
    };
    auto vdelta = [&](int i, int j) -> int { // D(i, j) - D(i - 1, j), i >= 1, j >= 1
        const int r = i - 1; const uint64_t* const ent = tb + static_cast<uint64_t>((j - 1) * W + (r >> 6)) * 2ull * RL_MS;
        return static_cast<int>((ent[0] >> (r & 63)) & 1ull) - static_cast<int>((ent[RL_MS] >> (r & 63)) & 1ull);
    };
    int i = m, j = n_cols;
    int cur = j > 0 ? cell(i, j) : i;
    int left = j > 1 ? cell(i, j - 1) : i; // D(i, j - 1); column 0 holds D(i, 0) = i
    while (i > 0 && j > 0) {
        const int vd = vdelta(i, j);
        const int vl = j > 1 ? vdelta(i, j - 1) : 1; // vertical delta of the column to the left (column 0: D(i, 0) - D(i - 1, 0) = 1)
        if (vd == 1) { rl_stb(c, mv_b + (--o), 1); --i; cur -= 1; left -= vl; }
        else if (cur - left == 1) { rl_stb(c, mv_b + (--o), 2); --j; cur = left; left = j > 1 ? cell(i, j - 1) : i; }
        else {
            const int diag = left - vl;
            rl_stb(c, mv_b + (--o), static_cast<unsigned char>(diag == cur ? 0 : 3));
            --i; --j; cur = diag; left = j > 1 ? cell(i, j - 1) : i;
        }
    }
    while (i > 0) { rl_stb(c, mv_b + (--o), 1); --i; }
    while (j > 0) { rl_stb(c, mv_b + (--o), 2); --j; }
    *off = o;
    RL_LEAVE(c, RL_P_WALK);
    return RL_MV_BYTES - o;
}

// edlibAlign with its path (task = path): SHW or NW of q against t; the moves land in move buffer mv_b from *off on
RTK_DEV RlAln rl_align_path(RlCtx& c, RlSrc q, int m, RlSrc t, int n, int mode, uint32_t mv_b, uint32_t* off, uint32_t* n_moves) {
    *n_moves = 0; *off = RL_MV_BYTES;
    const RlAln a = rl_myers(c, q, m, t, n, -1, mode, true, true, nullptr);
    if (c.fail() || a.dist < 0) return a;
    if (m > 0 && n > 0) {
        const int cols = (mode == RTK_MODE_NW) ? n : (a.first + 1);
        *n_moves = rl_myers_walk(c, m, cols, mv_b, off);
    } else if (m > 0) { // an empty target: the query's characters alone (edlib.cpp:161-179 returns no alignment; callers walk zero moves)
        *n_moves = 0;
    }
    return a;
}

// ------------------------------------------------------------------------------------------------ paths (src/Path.hpp)
// record in an arena / working path: [n, l, qlen, -] then n x (unitig << 1 | strand, dist, len), then the quality bytes
RTK_DEV uint32_t rl_wp(uint32_t i) { return RL_OFF_WP + i * RL_WP_W; }
RTK_DEV uint32_t rl_wp_qb(uint32_t i) { return 4u * (rl_wp(i) + 4u + 3u * RL_UM_CAP); } // byte offset of a working path's qualities
RTK_DEV uint32_t rl_arena(int lvl) { return lvl == 0 ? RL_OFF_A0 : (lvl == 1 ? RL_OFF_A1 : RL_OFF_A2); }
RTK_DEV uint32_t rl_h_mk(int lvl, uint32_t off) { return (static_cast<uint32_t>(lvl) << 30) | off; }
RTK_DEV uint32_t rl_h_w(uint32_t h) { return rl_arena(static_cast<int>(h >> 30)) + (h & 0x3FFFFFFFu); } // word offset of the record
#define RL_NOH 0xFFFFFFFFu

RTK_DEV UMap rl_um_ld(const RlCtx& c, uint32_t w) { UMap u; const uint32_t a = rl_ld(c, w); u.unitig = a >> 1; u.strand = a & 1u; u.dist = rl_ld(c, w + 1); u.len = rl_ld(c, w + 2); return u; }
RTK_DEV void rl_um_st(const RlCtx& c, uint32_t w, const UMap& u) { rl_st(c, w, (u.unitig << 1) | (u.strand & 1u)); rl_st(c, w + 1, u.dist); rl_st(c, w + 2, u.len); }

RTK_DEV uint32_t rl_p_n(const RlCtx& c, uint32_t pw) { return rl_ld(c, pw); }
RTK_DEV uint32_t rl_p_l(const RlCtx& c, uint32_t pw) { return rl_ld(c, pw + 1); }
RTK_DEV uint32_t rl_p_qlen(const RlCtx& c, uint32_t pw) { return rl_ld(c, pw + 2); }
RTK_DEV UMap rl_rec_back(const RlCtx& c, uint32_t h) { const uint32_t pw = rl_h_w(h); return rl_um_ld(c, pw + 4u + 3u * (rl_p_n(c, pw) - 1u)); }
RTK_DEV uint32_t rl_rec_qb(const RlCtx& c, uint32_t h) { const uint32_t pw = rl_h_w(h); return 4u * (pw + 4u + 3u * rl_p_n(c, pw)); } // byte offset of a record's qualities

RTK_DEV void rl_wp_clear(const RlCtx& c, uint32_t wi) { const uint32_t pw = rl_wp(wi); rl_st(c, pw, 0); rl_st(c, pw + 1, 0); rl_st(c, pw + 2, 0); }

RTK_FN uint32_t rl_wp_commit(RlCtx& c, uint32_t wi, int lvl) { // working path -> record
    RL_ENTER(c);
    const uint32_t pw = rl_wp(wi);
    const uint32_t n = rl_p_n(c, pw), l = rl_p_l(c, pw), ql = rl_p_qlen(c, pw);
    const uint32_t words = 4u + 3u * n + ((ql + 3u) >> 2);
    const uint32_t off = c.top(lvl);
    if (off + words > c.lim_arena(lvl)) { rl_fail(c, RL_F_ARENA); return RL_NOH; }
    c.top(lvl) = off + words;
    const uint32_t rw = rl_arena(lvl) + off;
    rl_st(c, rw, n); rl_st(c, rw + 1, l); rl_st(c, rw + 2, ql); rl_st(c, rw + 3, 0);
    rl_copy_words(c, rw + 4u, pw + 4u, 3u * n);
    rl_copy_words(c, rw + 4u + 3u * n, pw + 4u + 3u * RL_UM_CAP, (ql + 3u) >> 2);
    RL_LEAVE(c, RL_P_RECORDS);
    return rl_h_mk(lvl, off);
}
RTK_FN void rl_wp_load(RlCtx& c, uint32_t wi, uint32_t h) {
    RL_ENTER(c);
    const uint32_t pw = rl_wp(wi), rw = rl_h_w(h);
    const uint32_t n = rl_p_n(c, rw), l = rl_p_l(c, rw), ql = rl_p_qlen(c, rw);
    if (n > c.lim_um()) { rl_fail(c, RL_F_UM); rl_wp_clear(c, wi); return; }
    if (ql > c.lim_str()) { rl_fail(c, RL_F_STR); rl_wp_clear(c, wi); return; }
    rl_st(c, pw, n); rl_st(c, pw + 1, l); rl_st(c, pw + 2, ql);
    rl_copy_words(c, pw + 4u, rw + 4u, 3u * n);
    rl_copy_words(c, pw + 4u + 3u * RL_UM_CAP, rw + 4u + 3u * n, (ql + 3u) >> 2);
    RL_LEAVE(c, RL_P_RECORDS);
}

RTK_DEV void rl_wp_norm_back(const RlCtx& c, uint32_t pw, uint32_t n) { // the former end becomes a whole unitig (Path.hpp:319-323)
    if (n >= 2) { const uint32_t ew = pw + 4u + 3u * (n - 1u); const uint32_t u = rl_ld(c, ew) >> 1; rl_st(c, ew + 1, 0); rl_st(c, ew + 2, rl_nkm(c, u)); }
}
RTK_DEV void rl_wp_extend(RlCtx& c, uint32_t wi, const UMap& um) { // Path.hpp:308-330
    if (rtk_um_is_empty(um)) return;
    const uint32_t pw = rl_wp(wi); const uint32_t n = rl_p_n(c, pw);
    if (n >= c.lim_um()) { rl_fail(c, RL_F_UM); return; }
    if (n == 0) { rl_um_st(c, pw + 4u, um); rl_st(c, pw, 1); rl_st(c, pw + 1, um.len + c.k() - 1u); }
    else { rl_wp_norm_back(c, pw, n); rl_um_st(c, pw + 4u + 3u * n, um); rl_st(c, pw, n + 1u); rl_st(c, pw + 1, rl_p_l(c, pw) + um.len); }
}
// extend with the quality slice q[0 .. qn) (Path.hpp:332-363): appended only when its length equals um.len + k - 1
RTK_DEV void rl_wp_extend_q(RlCtx& c, uint32_t wi, const UMap& um, uint32_t q_b, uint32_t qn) {
    if (rtk_um_is_empty(um)) return;
    const uint32_t pw = rl_wp(wi); const uint32_t n = rl_p_n(c, pw);
    if (n >= c.lim_um()) { rl_fail(c, RL_F_UM); return; }
    const uint32_t want = um.len + c.k() - 1u;
    if (n == 0) {
        rl_um_st(c, pw + 4u, um); rl_st(c, pw, 1); rl_st(c, pw + 1, want);
        if (qn == want) { const uint32_t nl = rl_app(c, rl_wp_qb(wi), 0, rl_src_l(q_b), qn); if (!c.fail()) rl_st(c, pw + 2, nl); }
    } else {
        rl_wp_norm_back(c, pw, n); rl_um_st(c, pw + 4u + 3u * n, um); rl_st(c, pw, n + 1u); rl_st(c, pw + 1, rl_p_l(c, pw) + um.len);
        if (qn == want) { const uint32_t nl = rl_app(c, rl_wp_qb(wi), rl_p_qlen(c, pw), rl_src_l(q_b + (c.k() - 1u)), qn - (c.k() - 1u)); if (!c.fail()) rl_st(c, pw + 2, nl); }
    }
}
// a fresh single-unitig path whose qualities are all `ch` (string(len + k - 1, getQual(1.0)))
RTK_DEV void rl_wp_start(RlCtx& c, uint32_t wi, const UMap& um, char ch) {
    rl_wp_clear(c, wi);
    const uint32_t pw = rl_wp(wi); const uint32_t want = um.len + c.k() - 1u;
    if (want > c.lim_str()) { rl_fail(c, RL_F_STR); return; }
    rl_um_st(c, pw + 4u, um); rl_st(c, pw, 1); rl_st(c, pw + 1, want);
    rl_app_fill(c, rl_wp_qb(wi), 0, ch, want); rl_st(c, pw + 2, want);
}
// p.merge(o), o a record (Path.hpp:366-414)
RTK_FN void rl_wp_merge(RlCtx& c, uint32_t wi, uint32_t ho) {
    const uint32_t pw = rl_wp(wi), ow = rl_h_w(ho);
    const uint32_t on = rl_p_n(c, ow), ol = rl_p_l(c, ow), oq = rl_p_qlen(c, ow);
    if (ol == 0) return;
    if (rl_p_l(c, pw) == 0) { rl_wp_load(c, wi, ho); return; }
    uint32_t pn = rl_p_n(c, pw);
    if ((rl_p_qlen(c, pw) == 0) != (oq == 0)) return;
    const UMap last = rl_um_ld(c, pw + 4u + 3u * (pn - 1u)); const UMap o0 = rl_um_ld(c, ow + 4u);
    if (last.unitig != o0.unitig || last.strand != o0.strand) return;
    if (pn + on > c.lim_um()) { rl_fail(c, RL_F_UM); return; }
    UMap en = last;
    if (!en.strand) en.dist = o0.dist;
    en.len += o0.len - 1u;
    rl_um_st(c, pw + 4u + 3u * (pn - 1u), en);
    if (pn == 1) { for (uint32_t i = 1; i < on; ++i) { rl_um_st(c, pw + 4u + 3u * pn, rl_um_ld(c, ow + 4u + 3u * i)); ++pn; } }
    else if (on >= 2) { rl_wp_norm_back(c, pw, pn); for (uint32_t i = 1; i < on; ++i) { rl_um_st(c, pw + 4u + 3u * pn, rl_um_ld(c, ow + 4u + 3u * i)); ++pn; } }
    rl_st(c, pw, pn);
    rl_st(c, pw + 1, rl_p_l(c, pw) + ol - c.k());
    if (oq != 0) {
        const uint32_t add = oq > c.k() ? oq - c.k() : 0u; // o.qual.substr(k)
        const uint32_t nl = rl_app(c, rl_wp_qb(wi), rl_p_qlen(c, pw), rl_src_l(rl_rec_qb(c, ho) + c.k()), add);
        if (!c.fail()) rl_st(c, pw + 2, nl);
    }
}

// mappedSequenceToString of one mapping appended to a writer (2-bit decode, reverse complement on the fly), without its first `skip` characters
RTK_DEV void rl_um_decode(const RlCtx& c, RlW& w, const UMap& um, uint32_t skip) {
    const uint32_t n = um.len + c.k() - 1u;
    const uint64_t b0 = c.g()->uoff.get()[um.unitig] + um.dist;
    const uint64_t* const useq = c.g()->useq.get();
    if (um.strand) {
        uint64_t pos = b0 + skip; uint64_t word = useq[pos >> 5];
        for (uint32_t i = skip; i < n; ++i, ++pos) {
            if ((pos & 31ull) == 0ull) word = useq[pos >> 5];
            const uint32_t b = static_cast<uint32_t>((word >> (2u * (pos & 31ull))) & 3ull);
            rl_w_put(c, w, static_cast<unsigned char>((0x54474341u >> (8u * b)) & 0xFFu)); // "ACGT"
        }
    } else {
        uint64_t pos = b0 + (n - 1u - skip); uint64_t word = useq[pos >> 5];
        for (uint32_t i = skip; i < n; ++i, --pos) {
            if ((pos & 31ull) == 31ull) word = useq[pos >> 5];
            const uint32_t b = 3u - static_cast<uint32_t>((word >> (2u * (pos & 31ull))) & 3ull);
            rl_w_put(c, w, static_cast<unsigned char>((0x54474341u >> (8u * b)) & 0xFFu));
        }
    }
}
// Path::toString (Path.hpp:449-485) of the n mappings at word offset ums_w into string buffer `sb`; returns the length (0xFFFFFFFF: failed)
RTK_FN uint32_t rl_ums_to_string(RlCtx& c, uint32_t ums_w, uint32_t n, uint32_t sb) {
    uint32_t len = 0;
    RL_ENTER(c);
    for (uint32_t i = 0; i < n; ++i) { const uint32_t l = rl_ld(c, ums_w + 3u * i + 2u); len += l + (i ? 0u : c.k() - 1u); }
    if (len > c.lim_str()) { rl_fail(c, RL_F_STR); return 0xFFFFFFFFu; }
    RlW w = rl_w_open(c, rl_sb(sb), 0);
    for (uint32_t i = 0; i < n; ++i) rl_um_decode(c, w, rl_um_ld(c, ums_w + 3u * i), i ? c.k() - 1u : 0u);
    rl_w_close(c, w);
    c.c_pathbase += len;
    RL_LEAVE(c, RL_P_TOSTRING);
    return len;
}
RTK_DEV uint32_t rl_rec_to_string(RlCtx& c, uint32_t h, uint32_t sb) { const uint32_t pw = rl_h_w(h); return rl_ums_to_string(c, pw + 4u, rl_p_n(c, pw), sb); }
RTK_DEV uint32_t rl_wp_to_string(RlCtx& c, uint32_t wi, uint32_t sb) { const uint32_t pw = rl_wp(wi); return rl_ums_to_string(c, pw + 4u, rl_p_n(c, pw), sb); }

#include "rtk_region_lane2.h"

#endif
