/*
 * ratatosk_hip.h -- C ABI of the MI355X-native Ratatosk per-long-read correction path.
 *
 * The reference (DecodeGenetics/Ratatosk v0.9.0) has no FFI; its drop-in seam is the body of the
 * per-read worker loop of `Ratatosk correct -1` (reference: src/Ratatosk.cpp:808-864):
 *     getSeeds(opt, dbg, read, qual, false, ~0, m_km_um, false)            src/Graph.hpp:14-18
 *     correctSequence(dbg, opt, read, qual, solid, weak, false, nullptr, ~0, hap_reads, max_km_cov)
 *                                                                           src/Correction.hpp:32-35
 * called on a shared read-only graph loaded by dbg.read() + readGraphData() (src/Ratatosk.cpp:1087-1089,
 * src/Graph.cpp:722). Each entry point below names the reference interface it replaces. Plain pointers
 * and sizes only; every function returns 0 on success and a negative code on failure, with a message
 * available from rtk_last_error() (the reference prints to cerr and exit(1)s; a library must not).
 * All compute runs in HIP kernels on gfx950; there is no CPU fallback: a call fails with
 * RTK_ERR_NO_DEVICE when no GPU/extension is usable.
 */
#ifndef RATATOSK_HIP_H
#define RATATOSK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTK_OK 0
#define RTK_ERR_IO (-1)
#define RTK_ERR_FORMAT (-2)
#define RTK_ERR_ARG (-3)
#define RTK_ERR_NO_DEVICE (-4)
#define RTK_ERR_DEVICE (-5)
#define RTK_ERR_UNSUPPORTED (-6)

typedef struct rtk_graph rtk_graph; /* compacted coloured de Bruijn graph: host image + HBM image */
typedef struct rtk_batch rtk_batch; /* one batch of long reads resident in HBM (reference: the >=1 MiB ticket unit, src/Common.hpp:138) */

/* POD mirror of the Correct_Opt fields read by the pass-1 hot path (reference: src/Common.hpp:101-156). */
typedef struct rtk_opts {
    uint64_t insert_sz;             /* -i, 500  */
    uint64_t min_cov_vertices;      /* 2        */
    uint64_t max_len_weak_region1;  /* -w, 1000 */
    uint64_t max_km_cov;            /* max(getMaxKmerCoverage(dbg,0.001),128), src/Ratatosk.cpp:625 */
    double weak_region_len_factor;  /* 0.25     */
    double large_k_factor;          /* 1.5      */
    double min_score;               /* 0.0 with one correction round, src/Ratatosk.cpp:847 */
    int32_t max_qual;               /* -Q, 40   */
    int32_t out_qual;               /* 1        */
    double min_confidence_snp_corr; /* -m, 0.9 (src/Common.hpp:147): below this confidence a SNP-annotated base is re-decided against the read */
    /* second correction pass (`correct -2`; `long_read_correct` of search(), src/Ratatosk.cpp:618): the graph is coloured by the pass-1
     * reads, the reads' qualities are carried over, regions pass 1 already gave the maximum quality are left alone, no 1-edit search.
     * Needs batches created WITH quality strings. */
    int32_t long_read_correct;      /* 0 = pass 1 */
    int32_t force_unres_snp_corr;   /* -f (src/Ratatosk.cpp:279): second pass only, fixSNPs() (src/Alignment.cpp:846-965) on every read before phasing() (src/Ratatosk.cpp:828) */
    uint64_t max_len_weak_region2;  /* -W, 5000 (src/Common.hpp:110) */
    /* Bifrost assumption [A2] as a switch (the reference calls searchSequence(l_s, false, true, true, true, or_exclusive_match = true),
     * src/Graph.cpp:193, and Bifrost is not in the tree): 0 = every graph k-mer one substitution / insertion / deletion away from a window is
     * reported (union of the three searches); 1 = the searches run substitution -> insertion -> deletion and a window that one of them
     * matched is not searched by the next; 2 = the same in the order insertion -> deletion -> substitution. rtk_opts_default takes it from the
     * environment: RTK_A2_XOR=exclusive (default since round 4: the reading the flag's name asks for) | exclusive-ids | union. */
    int32_t a2_exclusive;
    /* Bifrost assumption [A3] as a switch: the order in which getSuccessors() hands out the (up to four) neighbours of a unitig end, which is the
     * order exploreSubGraph pushes them (src/GraphTraversal.cpp:456-587) and so decides between candidates of equal score. 0 = by the base
     * appended in walk direction, A,C,G,T, on both strands (default); 1 = on the reverse strand by the base as the unitig's own strand spells it
     * (A,C,G,T there = T,G,C,A in walk direction). rtk_opts_default takes it from the environment: RTK_A3_ORDER=walk (default) | strand. */
    int32_t a3_strand_order;
    /* Canonical rule [D1] as a switch. chooseColors visits the anchor colour sets "sorted by cardinality" after iterating an unordered_map keyed by
     * pointers (src/Correction.cpp:286-293: heap addresses + an unstable sort decide the order of sets of EQUAL cardinality); this build orders
     * such ties by unitig id: 0 = ascending (default), 1 = descending. rtk_opts_default takes it from the environment: RTK_D1_ORDER=asc | desc.
     * profiles/r04_d1_count.json counts the reads the choice decides. */
    int32_t d1_desc;
    /* sizeof(rtk_opts) of the library that filled the struct (rtk_opts_default writes it; round 5). Every entry that takes an rtk_opts refuses one whose
     * struct_size is not its own sizeof(rtk_opts) with RTK_ERR_ARG: a caller compiled against an older header (the struct has grown in rounds 2, 3, 4 and 5), or one
     * that zero-fills the struct instead of calling rtk_opts_default, is told so instead of running with fields it never set. New fields go BEFORE this one. */
    uint32_t struct_size;
} rtk_opts;

typedef struct rtk_graph_info {
    uint64_t n_unitigs, n_kmers, n_bases, n_colour_ids, n_global_sets;
    uint64_t table_slots, hbm_bytes;
    uint64_t max_km_cov_top; /* getMaxKmerCoverage(dbg, 0.001), src/Graph.cpp:825-841 */
    int32_t k, device;
} rtk_graph_info;

/* Kernel timing and event counters of the last rtk_batch_run (HIP events on the launch stream). */
typedef struct rtk_stats {
    double ms_total, ms_lookup_exact, ms_mask, ms_lookup_inexact, ms_seeds, ms_regions, ms_correct, ms_stitch;
    uint64_t n_windows, n_probes_exact, n_probes_inexact, n_hits_inexact, n_regions, n_region_items, n_arena_overflow;
    uint64_t n_expand, n_colour_elem, n_path_base, n_align, n_align_cells;
    uint64_t in_bases, out_bases;
    /* wave-cycles (s_memtime ticks summed over all waves) spent by k_regions per phase: colour selection, graph traversal incl.
     * its alignments, consensus + trimming alignments, everything else; and inside those, Myers passes and set algebra */
    uint64_t cyc_colour, cyc_paths, cyc_consensus, cyc_total, cyc_myers, cyc_sets /* path record commit+load */, cyc_tostring, cyc_pathqual;
    /* n_probes_* = k-mer queries (one 8-byte pre-filter word each); n_slots_* = 16-byte table slots visited behind the filter */
    uint64_t n_slots_exact, n_slots_inexact;
    /* traceback walks inside cyc_myers: wave-cycles and alignment moves produced */
    uint64_t cyc_walk, n_moves;
    /* lane-per-region kernel (k_regions_lanes, hip/rtk_region_lane.h; round 5): its time (also part of ms_correct), the regions of its class, and how many of
     * them it handed on to the wave kernel (a capacity, a short-cycle unitig, a character outside A C G T N ...: same results, counted) */
    double ms_lanes;
    uint64_t n_lane_regions, n_lane_handed;
    /* second pass: time of the phasing() step of the ticket, and the reads whose whole-read alignment was skipped because no stretch of theirs was marked
     * (the walk over the CIGAR then returns the corrected read as it is, src/Graph.cpp:975-1069) */
    double ms_phase;
    uint64_t n_phase_skipped;
} rtk_stats;

/* dbg.read(G.fasta.gz) + readGraphData(G.rtsk) (reference: src/Ratatosk.cpp:1087-1089; src/Graph.cpp:722-784).
 * Parses the unitig FASTA(.gz) and the .rtsk records, rebuilds the k-mer index (the .bfi file is ignored),
 * flattens everything to SoA/CSR arrays. k must be odd and <= 63: one-word k-mers (k <= 31) serve both passes, two-word k-mers
 * (33 .. 63) the second pass only (long_read_correct = 1; the 1-edit anchor search of the first pass is built on one-word k-mers). */
int rtk_graph_load(const char* unitig_fasta_gz, const char* rtsk, int k, int n_threads, rtk_graph** out);
/* The same with flags. RTK_LOAD_DEVICE_TABLES: the host only parses the two files and packs the unitigs; the lookup structures Bifrost keeps behind
 * find() / getSuccessors() -- the k-mer table with its presence filters, the half-k-mer index of the 1-edit search, the adjacency -- are built in HBM by
 * rtk_graph_upload (hip/rtk_graph_tables.hip; a k-mer that occurs twice in the unitig file is then reported by rtk_graph_upload, RTK_ERR_FORMAT).
 * What `Ratatosk correct` and the Python front end use; rtk_graph_load (flags = 0: everything on the host threads) stays the definition. */
#define RTK_LOAD_DEVICE_TABLES 1u
int rtk_graph_load2(const char* unitig_fasta_gz, const char* rtsk, int k, int n_threads, uint32_t flags, rtk_graph** out);

/* Copies the flat graph into HBM of `device` (HIP device ordinal). Must precede any compute call. */
int rtk_graph_upload(rtk_graph* g, int device);

/* Multi-GPU: rank 0 has loaded the graph; the others create an empty shell with rtk_graph_shell(), every rank
 * exposes its flat buffers so the caller can broadcast them (RCCL over xGMI via torch.distributed), then calls
 * rtk_graph_adopt_device() to mark the HBM image valid. Buffer order is fixed; sizes come from rank 0's info. */
int rtk_graph_shell(int k, rtk_graph** out);
int rtk_graph_n_buffers(const rtk_graph* g);
int rtk_graph_buffer(rtk_graph* g, int idx, void** dev_ptr, uint64_t* bytes);
int rtk_graph_alloc_buffers(rtk_graph* g, int device, const uint64_t* bytes, int n, const rtk_graph_info* info);
int rtk_graph_adopt_device(rtk_graph* g);
/* Same with caller-owned HBM buffers (e.g. torch tensors that torch.distributed broadcasts into); never freed by the library. */
int rtk_graph_attach_buffers(rtk_graph* g, int device, void* const* dev_ptrs, const uint64_t* bytes, int n, const rtk_graph_info* info);
int rtk_graph_buffer_bytes(const rtk_graph* g, uint64_t* bytes, int n);
/* A flat buffer of a resident graph handed over to caller-owned HBM (of at least its rtk_graph_buffer_bytes): copied device to device, the library's own
 * copy freed. For rank 0 of a multi-GPU job whose tables were built by rtk_graph_upload, before it broadcasts them. */
int rtk_graph_move_buffer(rtk_graph* g, int idx, void* dev_ptr, uint64_t bytes);
/* (tests, tools) flat buffer idx of the host image: a pointer into the graph's own memory, valid until rtk_graph_free */
int rtk_graph_host_buffer(const rtk_graph* g, int idx, const void** p, uint64_t* bytes);
/* (tests, tools) `bytes` bytes of flat buffer idx of a resident graph copied to host memory */
int rtk_graph_download_buffer(const rtk_graph* g, int idx, void* host_dst, uint64_t bytes);

/* Single-process multi-GPU (the C++ `Ratatosk correct` driver): the index is parsed and flattened ONCE, uploaded to one GPU and
 * replicated to the others device-to-device (xGMI peer copies, one per flat buffer) -- the reference's worker threads likewise
 * share one graph (src/Ratatosk.cpp:618,727). The clone has no host image; release it with rtk_graph_free. */
int rtk_n_devices(void); /* HIP devices visible to the process (0: every compute call fails with RTK_ERR_NO_DEVICE) */
/* free and total bytes of a device's memory (hipMemGetInfo): a host driver sizes its number of tickets in flight with it */
int rtk_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
int rtk_graph_clone_to_device(const rtk_graph* src, int device, rtk_graph** out);

int rtk_graph_get_info(const rtk_graph* g, rtk_graph_info* info);
void rtk_graph_free(rtk_graph* g);
/* Indexes written by the reference carry short-cycle annotations (detectShortCycles, src/Graph.cpp:4660, always) and SNP-ambiguity
 * annotations (detectSNPs, src/Graph.cpp:484, unless -F); both are consumed (fixRepeats, fixAmbiguity). This call drops BOTH kinds
 * from a loaded graph BEFORE rtk_graph_upload, which makes the graph identical to one whose index was written without them.
 * Returns the number of unitigs that carried an annotation, < 0 on error. */
long long rtk_graph_strip_annotations(rtk_graph* g);

/* Optional: allocate the per-wave scratch slabs of the seed and region stages (tens of GB, seconds of hipMalloc) for reads up to
 * max_read_len on `device` ahead of time, e.g. from a helper thread while rtk_graph_load parses the index. The first graph uploaded
 * to that device adopts them. Not part of the reference's seam; there is nothing to release. */
int rtk_reserve_scratch(int device, uint32_t max_read_len);
/* Second pass: n_tickets x (work area of the phasing step, work area of the region stage) reserved ahead; every ticket in flight owns a pair.
 * The first call for a device also reserves the work areas of the per-read seed kernels (one slab per graph). */
int rtk_reserve_second_pass(int device, uint32_t n_tickets, uint64_t phase_bytes, uint64_t region_bytes);
/* Round 6: the device buffers of n_batches tickets of about bases_per_batch bases / reads_per_batch reads (with_qualities: FASTQ input kept, second pass; raw_bases_per_batch != 0: the
 * second pass, which also holds the uncorrected reads) taken from the device and parked in the graph's pool, so that the first tickets of a run find them instead of calling hipMalloc inside the
 * correction phase (a hipMalloc stalls every stream of the device). A batch carves all its arrays out of ONE such buffer. RTK_ERR_DEVICE when an eighth of the device memory would not stay free
 * (what could be reserved stays reserved). */
int rtk_graph_reserve_batches(rtk_graph* g, uint32_t n_batches, uint64_t bases_per_batch, uint32_t reads_per_batch, int with_qualities, uint64_t raw_bases_per_batch);

/* Correct_Opt defaults + max_km_cov derived from the graph (reference: src/Common.hpp:101-156, src/Ratatosk.cpp:625). */
int rtk_opts_default(const rtk_graph* g, rtk_opts* opts);

/* The per-read loop body over one batch (reference: src/Ratatosk.cpp:808-864): upper-casing, quality clamp,
 * getSeeds, correctSequence. Inputs are borrowed; out_seq[i]/out_qual[i] are malloc'd NUL-terminated strings
 * owned by the caller (release with rtk_free). qual may be NULL (FASTA input). */
int rtk_correct_batch(rtk_graph* g, const rtk_opts* opts, uint32_t n, const char* const* seq, const char* const* qual,
                      const uint32_t* len, char** out_seq, char** out_qual, uint32_t* out_len);
/* Revision 6: the tickets of CONCURRENT rtk_correct_batch callers are merged into one launch. The reference's workers each take a ticket of
 * >= 1 MiB of bases (src/Common.hpp:138, src/Ratatosk.cpp:757-772) and call the loop body on it from `-c` threads at once; a launch that small
 * lasts as long as its longest read and leaves the device idle, so a call that arrives while another caller gathers a group joins it (first-pass
 * tickets below RTK_COALESCE_BASES = 16 Mi bases, same rtk_opts bytes, same kind of input; the gatherer waits <= RTK_COALESCE_WAIT_US = 1 500 us
 * when the device idles and another caller has been seen, not at all when it is alone, and for as long as a group in flight has not reached its region stage), the group runs as one batch and every caller copies out its own reads.
 * Results are those of the calls made one by one (reads are independent). RTK_COALESCE_BASES=0 switches it off.
 * rtk_coalesce_stats: groups launched and tickets they held since the graph was loaded (a group of one is a call that ran on its own). */
int rtk_coalesce_stats(rtk_graph* g, uint64_t* n_groups, uint64_t* n_tickets);

/* Same work split so that a caller can keep the batch resident in HBM across the timed region:
 * create = pack + H2D copy, run = kernels only (synchronous), fetch = D2H + unpack. */
int rtk_batch_create(rtk_graph* g, uint32_t n, const char* const* seq, const char* const* qual, const uint32_t* len, rtk_batch** out);
/* Second pass (`correct -2`): the pass-1 corrected reads WITH their qualities plus the uncorrected reads in the same order (the
 * reference reads the two files in lock-step and aborts when the names disagree, src/Ratatosk.cpp:774-802). Run with
 * opts->long_read_correct = 1: phasing() (src/Graph.cpp:869-1097) runs on the device before the seed stage. */
int rtk_batch_create2(rtk_graph* g, uint32_t n, const char* const* seq, const char* const* qual, const uint32_t* len,
                      const char* const* raw, const uint32_t* raw_len, rtk_batch** out);
int rtk_batch_run(rtk_batch* b, const rtk_opts* opts);
/* rtk_batch_run = rtk_batch_run_seeds (getSeeds of every read: src/Graph.cpp:3-482) followed by rtk_batch_run_regions (correctSequence
 * of every read: src/Correction.cpp:159-958). Every batch owns a HIP stream, and each call only waits for that stream: calling the
 * two stages of DIFFERENT batches from two host threads overlaps them on the device (the seed stage is latency-bound and leaves the
 * vector ALUs to the region stage of the previous batch), the way the reference overlaps its worker threads (src/Ratatosk.cpp:727). */
int rtk_batch_run_seeds(rtk_batch* b, const rtk_opts* opts);
int rtk_batch_run_regions(rtk_batch* b, const rtk_opts* opts);
int rtk_batch_fetch(rtk_batch* b, char** out_seq, char** out_qual, uint32_t* out_len);
/* rtk_batch_fetch without the per-read allocations: the corrected records stay packed in (pinned) host memory owned by the batch --
 * read i is pool[off[i] .. +len[i]) followed by its quality string pool[off[i]+len[i] .. +len[i]) -- valid until the batch is freed
 * or run again. What a writer that formats FASTQ blocks needs (reference: writeCorrectedOutput, src/Ratatosk.cpp:510-520). */
int rtk_batch_fetch_view(rtk_batch* b, const char** pool, const uint64_t** off, const uint32_t** len);
int rtk_batch_get_stats(const rtk_batch* b, rtk_stats* stats);
void rtk_batch_free(rtk_batch* b);

/* ---- stage-level entry points (used by the parity tests; same kernels as rtk_batch_run) ---- */

/* dbg.searchSequence(s, true, false, false, false, false) (reference: src/Graph.cpp:97): for every k-mer window p of
 * seq, hits[p] = unitig<<33 | dist<<1 | strand, or -1 when the window is not in the graph / holds a non-ACGT. */
int rtk_lookup_exact(rtk_graph* g, const char* seq, uint32_t len, int64_t* hits);

/* fixSNPs (reference: src/Alignment.cpp:846-965; `-f`, src/Ratatosk.cpp:828) of one read, upper-cased first (src/Ratatosk.cpp:814):
 * out[0..len) = the read with every ambiguous character that has exactly one graph-supported base replaced by it. */
int rtk_fix_snps(rtk_graph* g, const char* seq, uint32_t len, char* out);

/* getSeeds (reference: src/Graph.cpp:3-482): anchors as (pos, unitig, dist, strand) quadruples. */
int rtk_seeds(rtk_graph* g, const rtk_opts* opts, const char* seq, uint32_t len,
              uint64_t* n_solid, int64_t* solid, uint64_t* n_weak, int64_t* weak, uint64_t cap);

/* edlibAlign + edlibAlignmentToCigar (reference: src/edlib.cpp:141,298) with the IUPAC equality table of
 * src/Common.hpp:262-276 (use_iupac=0: edlibDefaultAlignConfig). mode 0 NW, 1 SHW, 2 HW; k<0 unbounded.
 * dist[i] = editDistance or -1; end_locs holds up to cap_locs locations per problem; cigars (standard format)
 * are written at cigar + i*cap_cigar when want_path. */
int rtk_myers_batch(uint32_t n, const char* const* query, const uint32_t* qlen, const char* const* target, const uint32_t* tlen,
                    const int32_t* k, const int32_t* mode, int want_path, int use_iupac,
                    int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, char* cigar, uint32_t cap_cigar);

/* The same problems on workgroups of `waves` wavefronts (2..16): the schedule the second pass uses for the whole-read alignment of its long
 * reads (reference: phasing(), src/Graph.cpp:975) -- wave 0 runs the alignment program, the others take the row blocks / banded passes it
 * publishes. waves <= 1 is rtk_myers_batch. Stage entry for parity tests and timing of that schedule; same results. */
int rtk_myers_batch_waves(uint32_t n, const char* const* query, const uint32_t* qlen, const char* const* target, const uint32_t* tlen,
                          const int32_t* k, const int32_t* mode, int want_path, int use_iupac,
                          int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, char* cigar, uint32_t cap_cigar, int waves);

/* rtk_myers_batch with ONE PROBLEM PER LANE: queries of up to 512 characters against targets of up to 2048 over A, C, G, T, N are computed column by column in a
 * lane's registers, 64 problems per wavefront (csrc/hip/rtk_myers_lane.h); with want_path the lane keeps the delta vectors of its columns (up to 4096
 * word-columns) and walks them back. What does not fit that takes the wave route of rtk_myers_batch. Same results
 * (reference: edlibAlign, src/edlib.cpp:131-296). Stage entry: one alignment per lane as the lane-per-region kernel does them (DESIGN_HISTORY.md section 3.8), for parity tests and
 * timing; the correction path does not call it. */
int rtk_myers_batch_lanes(uint32_t n, const char* const* query, const uint32_t* qlen, const char* const* target, const uint32_t* tlen,
                          const int32_t* k, const int32_t* mode, int want_path, int use_iupac,
                          int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, char* cigar, uint32_t cap_cigar);
/* How the calling thread's last rtk_myers_batch_lanes call split its problems: *lane_route = computed one per lane, *wave_route = handed on to rtk_myers_batch
 * (tests hold the lane route to every problem that fits it: a lane kernel that took nothing would otherwise pass on the wave route's results). */
void rtk_myers_lanes_last_routes(uint64_t* lane_route, uint64_t* wave_route);

/* Index build, device side (the reference builds its index on the CPU: `Ratatosk index`, src/Ratatosk.cpp:1066-1067, Bifrost
 * build + addCoverage src/Graph.cpp:1561). rtk_index_count_kmers: the canonical k-mers (odd k <= 31, A=0 C=1 G=2 T=3, first base in the
 * high bits) that the reads of `files` (plain or gzipped FASTA/FASTQ) hold at least min_count times, sorted ascending; *solid is freed
 * with rtk_free. The tool csrc/tools/build_index.cpp (--gpu) builds the same files with it as its CPU path does. */
int rtk_index_count_kmers(int device, int k, const char* const* files, int n_files, uint32_t min_count, int n_threads, uint64_t** solid, uint64_t* n_solid);
/* rtk_index_unitigs: the unitigs of a sorted set of canonical solid k-mers (Bifrost's compaction behind CompactedDBG::build, src/Ratatosk.cpp:1100-1118):
 * the k-mers in a table in HBM with their eight edge bits, every maximal chain of mutually unique links walked from its ends by one thread each, written
 * with its smallest canonical k-mer reading forwards, the unitigs in the order of those k-mers: unitig j = seq_pool[seq_off[j] .. seq_off[j + 1]), seeds[j]
 * its smallest k-mer. Chains that meet themselves (closed loops, hairpins) are not built: their k-mers come back in `left` (sorted) for the caller.
 * Outputs are freed with rtk_free. RTK_ERR_FORMAT if a k-mer ended up on two unitigs (the tool then takes its plain construction). */
int rtk_index_unitigs(int device, int k, const uint64_t* solid, uint64_t n_solid, char** seq_pool, uint64_t** seq_off, uint64_t** seeds, uint64_t* n_unitigs,
                      uint64_t** left, uint64_t* n_left);

/* rtk_index_colour_*: colours and coverage of an index build (addCoverage, src/Graph.cpp:1561-1985: every k-mer of every read mapped onto its unitig).
 * begin: unitig u = seq_pool[seq_off[u] .. seq_off[u + 1]); packs them and builds their k-mer table in HBM. chunk (any thread; calls are serialised inside):
 * sequences separated by '\n', read r starting at starts[r] with the id ids[r] (the caller numbers its reads: a pair keeps one id), at most 64 MB and
 * 4 M reads per call. end: the distinct events unitig << 32 | id in ascending order and the k-mer coverage of every unitig (rtk_free); releases the job
 * (with null outputs: only that). */
int rtk_index_colour_begin(int device, int k, const char* seq_pool, const uint64_t* seq_off, uint64_t n_unitigs, void** job);
int rtk_index_colour_chunk(void* job, const char* chars, uint64_t n_chars, const uint64_t* starts, const uint32_t* ids, uint32_t n_reads);
int rtk_index_colour_end(void* job, uint64_t** events, uint64_t* n_events, uint64_t** cov);

void rtk_free(void* p);
/* rtk_free of p[0 .. n) (the out_seq / out_qual arrays of rtk_correct_batch in one call; entries are set to NULL). */
void rtk_free_many(void** p, uint32_t n);
const char* rtk_last_error(void);
const char* rtk_version(void);
/* Interface revision, raised whenever a struct of this header grows or a default changes (5: rtk_opts.struct_size, rtk_stats lane fields, a2_exclusive default 1). */
#define RTK_API_REVISION 6
int rtk_api_revision(void);

#ifdef __cplusplus
}
#endif

#endif /* RATATOSK_HIP_H */
