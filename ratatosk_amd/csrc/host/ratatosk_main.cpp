// `Ratatosk` host driver for the in-scope branch of the reference CLI: `Ratatosk correct -1 -g G -d D -l reads -o OUT`
// (reference: src/Ratatosk.cpp:145-301 option table, :303-508 validation, :618-1000 search(), :1029-1037 file names,
// :1083-1095/:1145-1149 pass-1 branch). C++11 host orchestration over the C ABI of libratatosk_hip.so.
//
// Shape of the pipeline (the reference: N threads, each under a spin-lock reads a >= 1 MiB batch + takes a ticket, corrects it,
// under a second lock appends its block; blocks are re-ordered by ticket at the end, src/Ratatosk.cpp:727-999):
//   reader thread   parses FASTA/FASTQ(.gz) into packed ticket batches (one buffer per batch, no per-record strings), bounded queue;
//   GPU workers     `--workers-per-gpu` (3) threads per GPU: pack + H2D, kernels, D2H of one ticket each through the C ABI; the
//                   library overlaps the stages of different tickets on the device, the host work of one hides behind the kernels of
//                   the others; formatting of the FASTQ block happens here too, straight from the pinned fetch view;
//   ordered writer  whoever completes the next ticket in line writes it and every consecutive finished one (input order is
//                   guaranteed for pass 1: src/Ratatosk.cpp:919); workers that run too far ahead of the writer wait.
// The index is parsed and flattened ONCE (-c threads), uploaded to the first GPU and replicated device-to-device to the others.
// `-c` keeps the reference's meaning (host threads, <= hardware concurrency: src/Ratatosk.cpp:318); GPUs are chosen with --gpus.
// `correct -2 -g G2 -d D2 -l OUT.2.fastq -L raw_reads -o OUT` is the second pass (src/Ratatosk.cpp:1163-1262 with a pre-built second
// index): the uncorrected reads are read in lock-step with the pass-1 reads (:774-802), qualities are kept, output goes to OUT.fastq
// (:622), optionally gzipped (-G; one gzip member per ticket block, compressed by the workers) and trimmed / split at low-quality
// bases (-t, :508-563). `-f` = fixSNPs() on the reads of the second pass (:828). Everything else (`index`, `-u`, `-p/-P`, `-a`) is out of scope.
#include <getopt.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "../common/fastx.hpp"
#include "ratatosk_hip.h"

struct Opt {
    std::vector<std::string> in_long, in_long_raw;
    std::string out, graph, udata;
    bool workers_given = false;
    int cores = 1, gpus = 0, workers_per_gpu = 0, k1 = 31, k2 = 63, max_qual = 40, trim = 0, rounds = 1;
    bool force_snp = false;
    double min_conf_snp = 0.9;
    size_t insert_sz = 500, w1 = 1000, w2 = 5000, batch_bases = 0; // 0: not given (64 Mi for the first pass, 32 Mi for the second)
    bool pass1 = false, pass2 = false, verbose = false, correct = false, strip = false, gzip = false, parse_only = false;
};

static void usage() {
    fprintf(stderr, "Ratatosk (MI355X hot-path build)\n\nUsage: Ratatosk correct -1 -g <graph.fasta.gz> -d <unitig_data.rtsk> -l <long_reads> -o <out_prefix> [options]\n"
                    "  -c, --cores           number of host threads (default 1): index parsing, FASTQ formatting\n"
                    "      --gpus            number of GPUs to use (default: all visible)\n"
                    "      --workers-per-gpu tickets in flight per GPU (default 3; with -2 up to 8, as many as the device memory holds)\n"
                    "  -B, --batch-bases     long-read bases per ticket (default 64 Mi with -1, 32 Mi with -2)\n"
                    "  -i, --insert-sz       insert size of the short reads (default 500)\n"
                    "  -k, --k1              k-mer length of the 1st pass graph (default 31, <= 31)\n  -w, --max-len-weak1   maximum weak region length, 1st pass (default 1000)\n"
                    "  -r, --correction-rounds  correction rounds of the 1st pass (default 1)\n  -Q, --max-base-qual   maximum base quality (default 40)\n  -m, --min-conf-snp-corr  minimum confidence threshold to correct a SNP (default 0.9)\n  -v, --verbose\n"
                    "      --parse-only      developer: read and parse the long reads with -c threads, report the rate, correct nothing\n"
                    "      --allow-tinybitmap  decode colour sets stored as Bifrost TinyBitmap streams (most sets of an index written by the reference) with the layout\n"
                    "                        this build ASSUMES for them (unverified here: Bifrost is not part of the reference checkout); refused without it\n"
                    "      --strip-annotations  drop the short-cycle / SNP annotations of the index before correcting (fixRepeats / fixAmbiguity then have nothing to do)\n"
                    "Writes <out_prefix>.2.fastq (plain FASTQ, input order).\n\n"
                    "       Ratatosk correct -2 -g <graph2.fasta.gz> -d <unitig_data2.rtsk> -l <out_prefix>.2.fastq -L <long_reads> -o <out_prefix> [options]\n"
                    "  -L, --in-long-raw     the uncorrected long reads, same order as -l\n  -K, --k2              k-mer length of the 2nd pass graph (default 63)\n"
                    "  -W, --max-len-weak2   maximum weak region length, 2nd pass (default 5000)\n  -t, --trim-split      trim and split reads at bases with quality below this (default 0: off)\n"
                    "  -G, --gzip-out        write <out_prefix>.fastq.gz\n  -f, --force-correct-snp  resolve ambiguous characters of the reads that have one graph-supported base before correcting (default off)\nWrites <out_prefix>.fastq. Only `correct` with a pre-built index is in scope.\n");
}

struct Ticket { // one batch of reads, packed: the reference's >= buffer_sz unit of work (src/Common.hpp:138, src/Ratatosk.cpp:757)
    size_t id = 0;
    rtk::PackedReads reads, raw; // raw: second pass only
    Ticket(bool keep_qual) : reads(keep_qual), raw(false) {}
};

// writeCorrectedOutput with trim != 0 (src/Ratatosk.cpp:522-562): maximal runs of bases with quality >= trim, at least k long, as NAME/1, NAME/2, ...
static void append_trimmed(std::string& out, const char* name, size_t name_len, const char* seq, const char* qual, size_t len, int k, int trim) {
    const char c_min = static_cast<char>(trim + 33);
    long long start = -1, run = -1, id = 1;
    auto emit = [&]() {
        out += '@'; out.append(name, name_len); out += '/'; out += std::to_string(id++); out += '\n';
        out.append(seq + start, static_cast<size_t>(run)); out += "\n+\n"; out.append(qual + start, static_cast<size_t>(run)); out += '\n';
    };
    for (size_t pos = 0; pos < len; ++pos) {
        if (qual[pos] >= c_min) { if (start == -1) { start = static_cast<long long>(pos); run = 0; } ++run; }
        else { if (run >= k) emit(); start = -1; run = -1; }
    }
    if (run >= k) emit();
}

// -G: a ticket's FASTQ block as blocked gzip (BGZF, common/bgzf.hpp): gzip members of <= 64 KiB that concatenate into a valid .gz -- what
// `gunzip` / the reference's gzFile reader see is the same byte stream as from one big member, and a block-aware reader (this tool's own
// first-pass reader, bgzip, htslib) can inflate it on many threads. (Blocks are small, so zlib's 32-bit counters are never near their limit.)
static bool gzip_member(const std::string& in, std::string& out) { out.clear(); return rtk::bgzf_compress(in.data(), in.size(), out); }

int main(int argc, char** argv) {
    // Every ticket in flight owns a HIP stream (the phasing step of the second pass three), and their kernels are mostly narrow (one long read,
    // one big region): they only overlap if the runtime gives the streams hardware queues of their own. ROCm's default is 4 per process.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    Opt opt;
    if (argc <= 1 || !strcmp(argv[1], "--help")) { usage(); return 0; }
    if (!strcmp(argv[1], "--version")) { printf("%s\n", rtk_version()); return 0; }
    if (!strcmp(argv[1], "correct")) opt.correct = true;
    else if (!strcmp(argv[1], "index")) { fprintf(stderr, "Ratatosk::index: not in scope of this build (use the reference `index` step; its files are read as-is)\n"); return 1; }
    else { usage(); return 0; }
    static struct option lo[] = {{"in-long", required_argument, 0, 'l'}, {"out-long", required_argument, 0, 'o'}, {"cores", required_argument, 0, 'c'},
        {"in-graph", required_argument, 0, 'g'}, {"in-unitig-data", required_argument, 0, 'd'}, {"insert-sz", required_argument, 0, 'i'}, {"k1", required_argument, 0, 'k'},
        {"max-len-weak1", required_argument, 0, 'w'}, {"max-base-qual", required_argument, 0, 'Q'}, {"min-conf-snp-corr", required_argument, 0, 'm'}, {"1st-pass-only", no_argument, 0, '1'}, {"2nd-pass-only", no_argument, 0, '2'},
        {"in-long-raw", required_argument, 0, 'L'}, {"k2", required_argument, 0, 'K'}, {"max-len-weak2", required_argument, 0, 'W'}, {"trim-split", required_argument, 0, 't'}, {"gzip-out", no_argument, 0, 'G'},
        {"correction-rounds", required_argument, 0, 'r'}, {"no-snp-correction", no_argument, 0, 'F'}, {"force-io-order", no_argument, 0, 'O'}, {"no-graph-index", no_argument, 0, 'I'},
        {"in-unmapped-short", required_argument, 0, 'u'}, {"in-accurate-long", required_argument, 0, 'a'}, {"in-short-phase", required_argument, 0, 'p'}, {"in-long-phase", required_argument, 0, 'P'}, {"force-correct-snp", no_argument, 0, 'f'},
        {"sampling", required_argument, 0, 'S'}, {"min-conf-color2", required_argument, 0, 'M'}, {"min-len-color2", required_argument, 0, 'C'},
        {"batch-bases", required_argument, 0, 'B'}, {"strip-annotations", no_argument, 0, 1001}, {"gpus", required_argument, 0, 1002}, {"workers-per-gpu", required_argument, 0, 1003}, {"parse-only", no_argument, 0, 1004}, {"allow-tinybitmap", no_argument, 0, 1005}, {"verbose", no_argument, 0, 'v'}, {0, 0, 0, 0}};
    int c, idx = 0;
    while ((c = getopt_long(argc - 1, argv + 1, "s:l:o:c:g:d:i:k:w:Q:m:B:L:K:W:t:r:u:a:p:P:S:M:C:GFOIf12v", lo, &idx)) != -1) {
        switch (c) {
            case 'l': opt.in_long.push_back(optarg); break;
            case 'o': opt.out = optarg; break;
            case 'c': opt.cores = atoi(optarg); break;
            case 'g': opt.graph = optarg; break;
            case 'd': opt.udata = optarg; break;
            case 'i': opt.insert_sz = strtoull(optarg, nullptr, 10); break;
            case 'k': opt.k1 = atoi(optarg); break;
            case 'w': opt.w1 = strtoull(optarg, nullptr, 10); break;
            case 'Q': opt.max_qual = atoi(optarg); break;
            case 'm': opt.min_conf_snp = atof(optarg); break;
            case 'B': opt.batch_bases = strtoull(optarg, nullptr, 10); break;
            case 'r': opt.rounds = atoi(optarg); break;
            case 'F': case 'I': case 'S': case 'M': case 'C': break; // only read by `index` (detectSNPs, .bfi, addCoverage: src/Ratatosk.cpp:1067,1124; src/Graph.cpp:1573,1796,2117)
            case 'O': break; // output is in input order in both passes here (src/Ratatosk.cpp:919 re-orders only when asked in pass 2)
            case 'f': opt.force_snp = true; break; // fixSNPs() before phasing() in the second pass (src/Ratatosk.cpp:279,828); the first pass does not look at it
            case 'u': case 'a': case 'p': case 'P': fprintf(stderr, "Ratatosk::correct: -%c (unmapped-read rescue / helper long reads / phased input) is not in scope of this build\n", c); return 1;
            case 'L': opt.in_long_raw.push_back(optarg); break;
            case 'K': opt.k2 = atoi(optarg); break;
            case 'W': opt.w2 = strtoull(optarg, nullptr, 10); break;
            case 't': opt.trim = atoi(optarg); break;
            case 'G': opt.gzip = true; break;
            case '1': opt.pass1 = true; break;
            case '2': opt.pass2 = true; break;
            case 'v': opt.verbose = true; break;
            case 1001: opt.strip = true; break;
            case 1002: opt.gpus = atoi(optarg); break;
            case 1003: opt.workers_per_gpu = atoi(optarg); opt.workers_given = true; break;
            case 1004: opt.parse_only = true; break;
            case 1005: setenv("RTK_ALLOW_TINYBITMAP", "1", 1); break; // the loader reads it (common/rtsk_io.hpp): colour sets written as Bifrost TinyBitmap streams are decoded under assumption [A8]
            case 's': fprintf(stderr, "Ratatosk::correct: short reads are only needed by `index` (not in scope); ignored\n"); break;
            default: usage(); return 0; // the reference returns 0 on option errors too (src/Ratatosk.cpp:1018)
        }
    }
    // the reference drops arguments that belong to no option without a word (getopt_long permutes them to the end, src/Ratatosk.cpp:186-300):
    // `-l a.fq b.fq` corrects a.fq only. Same here, but said aloud.
    for (int i = optind; i < argc - 1; ++i) fprintf(stderr, "Ratatosk::correct: argument '%s' belongs to no option and is ignored (several input files: one -l each, or a text file of paths)\n", argv[1 + i]);
    if (opt.pass1 == opt.pass2) { fprintf(stderr, "Ratatosk::correct: one pass per run with a pre-built index (-g, -d): give -1 or -2\n"); return 1; }
    const bool lrc = opt.pass2;
    // ticket size when -B is not given. First pass: 64 Mi (1.22 x 10^9 bases/s file to file against 1.19 at 32 Mi and 1.05-1.20 at 96 Mi, profiles/r06_split_and_cli_B.txt: a ticket's launches carry
    // ~10 ms of latency whatever it holds). Second pass: 32 Mi (eight tickets in flight with 24 GB of work areas each)
    if (opt.batch_bases == 0) opt.batch_bases = lrc ? (32u << 20) : (64u << 20);
    if (lrc && opt.in_long_raw.empty()) { fprintf(stderr, "Ratatosk::correct: -2 needs the uncorrected long reads (-L) next to the pass-1 reads (-l)\n"); return 0; }
    if (opt.rounds < 1) { fprintf(stderr, "Ratatosk::Ratatosk(): Number of correction rounds cannot be less than 1.\n"); return 0; } // src/Ratatosk.cpp:348-352
    if (opt.trim < 0 || opt.trim > opt.max_qual) { fprintf(stderr, "Ratatosk::Ratatosk(): Quality score trimming threshold cannot be less than 0 or more than %d (%d given).\n", opt.max_qual, opt.trim); /* src/Ratatosk.cpp:324-326 */ return 0; }
    if (!opt.parse_only && (opt.graph.empty() || opt.udata.empty() || opt.in_long.empty() || opt.out.empty())) { fprintf(stderr, "Ratatosk::correct: -g, -d, -l and -o are required\n"); return 0; }
    { // src/Ratatosk.cpp:312-322
        const unsigned hc = std::thread::hardware_concurrency();
        if (opt.cores <= 0) { fprintf(stderr, "Ratatosk::Ratatosk(): Number of threads cannot be less than or equal to 0.\n"); return 0; }
        if (hc && static_cast<unsigned>(opt.cores) > hc) { fprintf(stderr, "Ratatosk::Ratatosk(): Number of threads cannot be greater than or equal to %u.\n", hc); return 0; }
    }
    if (opt.min_conf_snp < 0.0 || opt.min_conf_snp > 1.0) { fprintf(stderr, "Ratatosk::Ratatosk(): Minimum confidence threshold to correct a SNP must be in [0.0, 1.0].\n"); return 0; } // src/Ratatosk.cpp:366-376
    if (opt.workers_per_gpu < 1) opt.workers_per_gpu = opt.pass2 ? 8 : 3; // second pass: a ticket's launches are long and mostly narrow (its longest read, its biggest region): more of them in flight
    if (opt.batch_bases < 1) opt.batch_bases = 1;

    if (opt.parse_only) { // reader alone: how fast do -c threads turn the input files into tickets?
        std::vector<std::string> fl; for (size_t i = 0; i < opt.in_long.size(); ++i) { const std::vector<std::string> v = rtk::expand_input_list(opt.in_long[i]); fl.insert(fl.end(), v.begin(), v.end()); }
        const auto t0 = std::chrono::steady_clock::now();
        std::atomic<unsigned long long> bases(0), bytes(0), reads(0); int n_plain = 0;
        for (size_t f = 0; f < fl.size(); ++f) {
            if (rtk::PlainChunks::is_plain(fl[f])) {
                ++n_plain;
                rtk::PlainChunks pc; if (!pc.open(fl[f], 2 * opt.batch_bases + (opt.batch_bases >> 4))) { fprintf(stderr, "cannot open %s\n", fl[f].c_str()); return 1; }
                std::atomic<size_t> next(0); std::atomic<bool> bad(false), malformed(false); std::vector<std::thread> th;
                for (int t = 0; t < opt.cores; ++t) th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= pc.n_chunks() || bad.load()) break; rtk::PackedReads r(false); if (!pc.parse_chunk(i, r)) { if (r.malformed()) malformed = true; bad = true; break; } bases += r.n_bases(); reads += r.size(); } });
                for (size_t t = 0; t < th.size(); ++t) th[t].join();
                if (bad.load()) { fprintf(stderr, "Ratatosk::search(): %s %s\n", malformed.load() ? "a record that is not laid out as 4-line FASTQ in" : "read error on", fl[f].c_str()); return 1; }
                bytes += pc.file_bytes();
            } else { rtk::FastxReader rd; if (!rd.open(fl[f], std::max(1, std::min(opt.cores, 16)))) { fprintf(stderr, "cannot open %s\n", fl[f].c_str()); return 1; } rtk::PackedReads r(false); while (rd.next_packed(r)) { if (r.n_bases() > opt.batch_bases) { bases += r.n_bases(); reads += r.size(); rtk::PackedReads fresh(false); r = std::move(fresh); } } bases += r.n_bases(); reads += r.size(); }
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("Ratatosk::parse-only: %zu file(s) (%d plain or blocked gzip, read as byte ranges by %d threads), %llu reads, %llu bases, %.3f s: %.3g bases/s, %.2f GB/s of FASTA/FASTQ text\n", fl.size(), n_plain, opt.cores,
               reads.load(), bases.load(), dt, dt > 0 ? bases.load() / dt : 0.0, dt > 0 ? bytes.load() / dt / 1e9 : 0.0);
        return 0;
    }
    const int n_dev = rtk_n_devices();
    if (n_dev <= 0) { fprintf(stderr, "Ratatosk::Ratatosk(): no HIP device visible: the correction path has no CPU fallback\n"); return 1; }
    if (opt.gpus < 0 || opt.gpus > n_dev) { fprintf(stderr, "Ratatosk::Ratatosk(): --gpus %d but %d HIP device(s) are visible\n", opt.gpus, n_dev); return 1; }
    const int n_gpus = opt.gpus ? opt.gpus : n_dev;

    // input files (a text file lists one path per line: src/Common.cpp:428-446)
    std::vector<std::string> files, files_raw;
    for (size_t i = 0; i < opt.in_long.size(); ++i) { const std::vector<std::string> v = rtk::expand_input_list(opt.in_long[i]); files.insert(files.end(), v.begin(), v.end()); }
    for (size_t i = 0; i < opt.in_long_raw.size(); ++i) { const std::vector<std::string> v = rtk::expand_input_list(opt.in_long_raw[i]); files_raw.insert(files_raw.end(), v.begin(), v.end()); }
    const int k_graph = lrc ? opt.k2 : opt.k1;

    auto now_us = []() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    if (opt.verbose) printf("Ratatosk::Ratatosk(): Reading graph.\n");
    const long long t_load0 = now_us();
    std::vector<rtk_graph*> graphs(n_gpus, nullptr);
    // the per-wave scratch slabs (tens of GB per GPU, seconds of hipMalloc) are reserved while the index files are parsed
    std::vector<std::thread> reservers;
    for (int w = 0; w < n_gpus; ++w) {
        if (!lrc) reservers.emplace_back([w]() { rtk_reserve_scratch(w, 131072u); });
        else for (int t = 0; t < std::min(opt.workers_per_gpu, 4); ++t) reservers.emplace_back([w, &opt]() { // a 32 Mi ticket of long reads: ~12 GB for the phasing step, ~9 GB for its regions
            if (rtk_reserve_second_pass(w, 1u, 14ull << 30, 10ull << 30) != RTK_OK && opt.verbose) fprintf(stderr, "Ratatosk::Ratatosk(): %s\n", rtk_last_error()); });
    }
    { // ONE parse + flatten, ONE host image; the other GPUs get device-to-device copies of the flat buffers
        bool ok = rtk_graph_load2(opt.graph.c_str(), opt.udata.c_str(), k_graph, opt.cores, RTK_LOAD_DEVICE_TABLES, &graphs[0]) == RTK_OK; // (k-mer table, half-k-mer index, adjacency: built in HBM by the upload)
        if (ok && opt.strip) { const long long ns = rtk_graph_strip_annotations(graphs[0]); if (ns > 0) fprintf(stderr, "Ratatosk::Ratatosk(): dropped the short-cycle / SNP annotations of %lld unitigs\n", ns); }
        ok = ok && rtk_graph_upload(graphs[0], 0) == RTK_OK;
        for (int w = 1; ok && w < n_gpus; ++w) ok = rtk_graph_clone_to_device(graphs[0], w, &graphs[w]) == RTK_OK;
        if (!ok) { fprintf(stderr, "Ratatosk::Ratatosk(): %s\n", rtk_last_error()); for (size_t i = 0; i < reservers.size(); ++i) reservers[i].join(); exit(1); }
    }
    for (size_t i = 0; i < reservers.size(); ++i) reservers[i].join();
    const int first_reserved = std::min(opt.workers_per_gpu, 4);
    if (lrc && !opt.workers_given) { // second pass, tickets in flight not given: as many as the memory next to the graph image holds (~24 GB of work areas + ~4 GB of buffers each)
        uint64_t fr = 0, tot = 0; int fit = opt.workers_per_gpu;
        for (int w = 0; w < n_gpus; ++w) if (rtk_device_memory(w, &fr, &tot) == RTK_OK) {
            const uint64_t reserved = static_cast<uint64_t>(std::min(opt.workers_per_gpu, 4)) * (24ull << 30);
            const uint64_t avail = fr + reserved > tot / 8 ? fr + reserved - tot / 8 : 0;
            fit = std::min<int>(fit, static_cast<int>(avail / (28ull << 30)));
        }
        opt.workers_per_gpu = std::max(2, fit);
    }
    if (lrc) {
        // the work areas of the other tickets in flight, now that their number is known: a hipMalloc of 10+ GB in the middle of the correction phase takes up to a second
        // and every kernel of the device waits for it (the tickets running beside it took 1.5 s instead of 0.25 in the trace that found this)
        std::vector<std::thread> more;
        for (int w = 0; w < n_gpus; ++w) for (int t = first_reserved; t < opt.workers_per_gpu; ++t) more.emplace_back([w]() { rtk_reserve_second_pass(w, 1u, 14ull << 30, 10ull << 30); });
        for (size_t i = 0; i < more.size(); ++i) more[i].join();
    }
    { // the device buffers of the tickets that will be in flight (one buffer per ticket: the library carves a ticket's arrays out of it), before the correction phase starts
        const uint64_t bb = opt.batch_bases + (opt.batch_bases >> 3); const uint32_t rr = static_cast<uint32_t>(std::min<uint64_t>(bb / 1000 + 64, 1u << 30));
        std::vector<std::thread> rs;
        for (int w = 0; w < n_gpus; ++w) rs.emplace_back([&, w]() { if (rtk_graph_reserve_batches(graphs[w], static_cast<uint32_t>(opt.workers_per_gpu) + 3u /* tickets whose records the formatter threads still read are alive too */, bb, rr, lrc ? 1 : 0, lrc ? bb : 0) != RTK_OK && opt.verbose) fprintf(stderr, "Ratatosk::Ratatosk(): %s\n", rtk_last_error()); });
        for (size_t i = 0; i < rs.size(); ++i) rs[i].join();
    }
    const long long t_load1 = now_us();
    rtk_opts ro; rtk_opts_default(graphs[0], &ro);
    ro.insert_sz = opt.insert_sz; ro.max_len_weak_region1 = opt.w1; ro.max_len_weak_region2 = opt.w2; ro.max_qual = opt.max_qual; ro.min_confidence_snp_corr = opt.min_conf_snp;
    ro.long_read_correct = lrc ? 1 : 0;
    ro.force_unres_snp_corr = (lrc && opt.force_snp) ? 1 : 0;
    const bool gz_out = opt.gzip && lrc; // compress_out && long_read_correct (src/Ratatosk.cpp:620); output order is kept either way here
    const int trim = lrc ? opt.trim : 0; // pass 1 qualities are placeholders: the reference trims the final output only (src/Ratatosk.cpp:965-975)

    // pass 1: opt_pass1.filename_long_out += ".2" (src/Ratatosk.cpp:1079) + ".fastq" (:622); pass 2: OUT.fastq[.gz]
    const std::string fn_out = opt.out + (lrc ? ".fastq" : ".2.fastq") + (gz_out ? ".gz" : "");
    // Blocks of formatted records are written with pwrite at offsets handed out in ticket order (a block's offset is the sum of the sizes of the
    // blocks before it, known as soon as those are formatted): the writes themselves run side by side on the threads that formatted them.
    const int fd_out = ::open(fn_out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd_out < 0) { fprintf(stderr, "Ratatosk::search(): cannot open %s for writing\n", fn_out.c_str()); exit(1); }

    if (opt.verbose) printf("Ratatosk::Ratatosk(): Correcting long reads (%d/2).\n", lrc ? 2 : 1);
    const int n_workers = opt.workers_per_gpu * n_gpus;
    // formatters: FASTQ blocks from the fetched views, gzip (-G), pwrite. Their own threads, so that a GPU worker goes on to its next ticket as soon as
    // the corrected records of the last one are in host memory.
    const int n_fmt = gz_out ? std::max(2, std::min(opt.cores, 32)) : std::max(2, std::min(opt.cores, 2 * n_workers));
    const size_t q_cap = static_cast<size_t>(n_workers) + 2, ahead_cap = 2 * static_cast<size_t>(n_workers) + 2 + static_cast<size_t>(n_fmt);
    struct Fetched { std::unique_ptr<Ticket> t; rtk_batch* b = nullptr; const char* pool = nullptr; const uint64_t* off = nullptr; const uint32_t* olen = nullptr; };
    std::mutex m_f; std::condition_variable cv_f_empty, cv_f_full; std::deque<Fetched> fq;
    int gpu_workers_left = n_workers; size_t n_fetched_alive = 0; // fetched batches not yet formatted: each still holds its device buffers and its pinned view
    const size_t fetched_cap = static_cast<size_t>(n_workers) + 2;
    std::mutex m_in, m_out; std::condition_variable cv_in_full, cv_in_empty, cv_out;
    std::deque<std::unique_ptr<Ticket> > queue; bool reader_done = false;
    size_t next_to_write = 0; // first ticket whose block has no offset yet (every ticket before it is formatted)
    unsigned long long out_off = 0; // offset of that block in the output file
    std::map<size_t, std::string> done; // ticket -> formatted FASTQ block waiting for the blocks before it, at most ahead_cap entries
    std::atomic<bool> failed(false);
    std::string fail_msg; std::mutex m_fail;
    std::atomic<long long> us_parse(0), us_correct(0), us_format(0), us_write(0), n_reads(0), n_bases(0);
    const long long t_begin = now_us();
    const bool cli_trace = getenv("RTK_CLI_TRACE") != nullptr; // developer: per-ticket times of the workers
    std::function<void()> on_fail_extra; // (wakes the threads of the two-file reader, declared further down)
    auto fail = [&](const std::string& msg) {
        { std::lock_guard<std::mutex> lk(m_fail); if (fail_msg.empty()) fail_msg = msg; }
        failed = true;
        { std::lock_guard<std::mutex> lk(m_in); } cv_in_full.notify_all(); cv_in_empty.notify_all();
        { std::lock_guard<std::mutex> lk(m_out); } cv_out.notify_all();
        { std::lock_guard<std::mutex> lk(m_f); } cv_f_empty.notify_all(); cv_f_full.notify_all();
        if (on_fail_extra) on_fail_extra();
    };

    // Reader. First pass on plain (uncompressed) or blocked-gzip (BGZF: bgzip output, this tool's own -G output) files: the files are cut into
    // byte ranges of their text of about one ticket each and -c threads inflate + parse them independently (rtk::PlainChunks; ticket id =
    // range number, so the output keeps the input order). Everything else -- ordinary gzip input (one deflate stream can only be inflated
    // from its start), the lock-step pair of files of the second pass -- goes through the one reader thread below.
    bool par_read = !lrc && !getenv("RTK_SERIAL_READER"); // (the environment switch: A/B against the one-thread reader)
    for (size_t i = 0; par_read && i < files.size(); ++i) par_read = rtk::PlainChunks::is_plain(files[i]);
    std::vector<std::unique_ptr<rtk::PlainChunks> > pcs; std::vector<size_t> pc_first; size_t n_chunks_all = 0;
    if (par_read) {
        for (size_t i = 0; i < files.size(); ++i) {
            pcs.emplace_back(new rtk::PlainChunks());
            if (!pcs.back()->open(files[i], 2 * opt.batch_bases + (opt.batch_bases >> 4))) { fprintf(stderr, "Ratatosk::search(): cannot open input file %s\n", files[i].c_str()); exit(1); }
            pc_first.push_back(n_chunks_all); n_chunks_all += pcs.back()->n_chunks();
        }
    }
    std::atomic<size_t> next_chunk(0); std::atomic<int> parsers_left(0);
    auto parser = [&]() {
        for (;;) {
            const size_t id = next_chunk.fetch_add(1);
            if (id >= n_chunks_all || failed) break;
            { std::unique_lock<std::mutex> lk(m_out); cv_out.wait(lk, [&]() { return id < next_to_write + ahead_cap || failed; }); if (failed) break; } // bounded run-ahead of the writer
            const long long tp0 = now_us();
            size_t f = pcs.size() - 1; while (pc_first[f] > id) --f;
            std::unique_ptr<Ticket> t(new Ticket(lrc)); t->id = id;
            if (!pcs[f]->parse_chunk(id - pc_first[f], t->reads)) { fail(t->reads.malformed() ? "Ratatosk::search(): " + files[f] + " starts as 4-line FASTQ but holds a record laid out differently (multi-line?): the byte-range reader refuses it; RTK_SERIAL_READER=1 reads such a file on one thread" : "Ratatosk::search(): read error on " + files[f]); break; }
            us_parse += now_us() - tp0;
            n_bases += static_cast<long long>(t->reads.n_bases());
            if (opt.verbose) { const long long a = n_reads.fetch_add(static_cast<long long>(t->reads.size())), b2 = a + static_cast<long long>(t->reads.size()); if (a / 1000 != b2 / 1000) printf("Ratatosk::correct(): Processed %lld reads \n", b2 / 1000 * 1000); }
            std::unique_lock<std::mutex> lk(m_in);
            cv_in_full.wait(lk, [&]() { return queue.size() < q_cap + static_cast<size_t>(opt.cores) || failed; });
            if (failed) break;
            queue.push_back(std::move(t));
            cv_in_empty.notify_one();
        }
        if (parsers_left.fetch_sub(1) == 1) { { std::lock_guard<std::mutex> lk(m_in); reader_done = true; } cv_in_empty.notify_all(); }
    };
    std::vector<std::thread> parser_threads;
    if (par_read) { const int np = std::max(1, std::min(opt.cores, (!pcs.empty() && pcs[0]->is_bgzf()) ? 32 : 16)); /* inflating: ~0.4 GB/s of text per thread */ parsers_left = np; for (int i = 0; i < np; ++i) parser_threads.emplace_back(parser); }

    // One-thread reader (ordinary gzip input; the second pass). The second pass reads two files in lock-step (src/Ratatosk.cpp:774-802): the
    // corrected reads fill a ticket on this thread, the uncorrected reads of the same ticket are parsed and checked on a second one
    // (raw_thread) while this one is already on the next ticket.
    std::mutex m_half; std::condition_variable cv_half_full, cv_half_empty; std::deque<std::unique_ptr<Ticket> > half; bool half_done = false;
    const char* out_of_step = "Ratatosk::correct(): Corrected read file is not in the same order as input long read file. Abort."; // src/Ratatosk.cpp:787,796
    on_fail_extra = [&]() { { std::lock_guard<std::mutex> lk(m_half); } cv_half_full.notify_all(); cv_half_empty.notify_all(); };
    std::thread reader_thread([&]() {
        if (par_read) return;
        rtk::FastxReader reader; size_t file_i = 0; bool file_open = false, eof_all = false; size_t ticket = 0;
        const int inflate_threads = std::max(1, std::min(lrc ? opt.cores / 2 : opt.cores, 16)); // gzip input of several members is inflated on these (common/mgzip.hpp)
        while (!eof_all && !failed) {
            const long long tp0 = now_us();
            std::unique_ptr<Ticket> t(new Ticket(lrc)); t->id = ticket;
            t->reads.reserve(opt.batch_bases + (opt.batch_bases >> 3));
            while (t->reads.n_bases() < opt.batch_bases) {
                if (!file_open) { if (file_i >= files.size()) { eof_all = true; break; } if (!reader.open(files[file_i++], inflate_threads)) { fail("Ratatosk::search(): cannot open input file " + files[file_i - 1]); return; } file_open = true; }
                if (!reader.next_packed(t->reads)) { if (reader.failed()) { fail("Ratatosk::search(): " + files[file_i - 1] + " ends in a damaged or cut-short gzip stream"); return; } file_open = false; continue; }
                if (opt.verbose && ((n_reads.fetch_add(1) + 1) % 1000 == 0)) printf("Ratatosk::correct(): Processed %lld reads \n", n_reads.load());
            }
            us_parse += now_us() - tp0;
            if (t->reads.size() == 0) break;
            n_bases += static_cast<long long>(t->reads.n_bases());
            ++ticket;
            if (lrc) {
                std::unique_lock<std::mutex> lk(m_half);
                cv_half_full.wait(lk, [&]() { return half.size() < 3 || failed; });
                if (failed) break;
                half.push_back(std::move(t));
                cv_half_empty.notify_one();
            } else {
                std::unique_lock<std::mutex> lk(m_in);
                cv_in_full.wait(lk, [&]() { return queue.size() < q_cap || failed; });
                if (failed) break;
                queue.push_back(std::move(t));
                cv_in_empty.notify_one();
            }
        }
        if (lrc) { { std::lock_guard<std::mutex> lk(m_half); half_done = true; } cv_half_empty.notify_all(); return; }
        { std::lock_guard<std::mutex> lk(m_in); reader_done = true; }
        cv_in_empty.notify_all();
    });
    std::thread raw_thread([&]() {
        if (par_read || !lrc) return;
        rtk::FastxReader reader_raw; size_t file_raw_i = 0; bool file_raw_open = false;
        while (!failed) {
            std::unique_ptr<Ticket> t;
            {
                std::unique_lock<std::mutex> lk(m_half);
                cv_half_empty.wait(lk, [&]() { return !half.empty() || half_done || failed; });
                if (failed || half.empty()) break;
                t = std::move(half.front()); half.pop_front();
                cv_half_full.notify_one();
            }
            const long long tp0 = now_us();
            t->raw.reserve(t->reads.n_bases() + (t->reads.n_bases() >> 3));
            while (t->raw.size() < t->reads.size()) { // the uncorrected reads in lock-step (src/Ratatosk.cpp:774-802)
                if (!file_raw_open) { if (file_raw_i >= files_raw.size()) break; if (!reader_raw.open(files_raw[file_raw_i++], std::max(1, std::min(opt.cores / 2, 16)))) { fail("Ratatosk::search(): cannot open input file " + files_raw[file_raw_i - 1]); return; } file_raw_open = true; }
                if (!reader_raw.next_packed(t->raw)) { if (reader_raw.failed()) { fail("Ratatosk::search(): " + files_raw[file_raw_i - 1] + " ends in a damaged or cut-short gzip stream"); return; } file_raw_open = false; continue; }
            }
            if (t->raw.size() != t->reads.size()) { fail(out_of_step); return; }
            for (size_t i = 0; i < t->reads.size(); ++i) { // names are compared from their second character on, like the reference (:794)
                const size_t la = t->reads.name_len(i), lb = t->raw.name_len(i);
                if (la == 0 || lb == 0 || la != lb || memcmp(t->reads.name(i) + 1, t->raw.name(i) + 1, la - 1) != 0) { fail(out_of_step); return; }
                if (t->reads.qual(i) == nullptr && t->reads.seq_len(i)) { fail("Ratatosk::correct(): the second pass needs the base qualities of the first (FASTQ input for -l)"); return; }
            }
            us_parse += now_us() - tp0;
            std::unique_lock<std::mutex> lk(m_in);
            cv_in_full.wait(lk, [&]() { return queue.size() < q_cap || failed; });
            if (failed) break;
            queue.push_back(std::move(t));
            cv_in_empty.notify_one();
        }
        { std::lock_guard<std::mutex> lk(m_in); reader_done = true; }
        cv_in_empty.notify_all();
    });

    auto worker = [&](int w) {
        rtk_graph* g = graphs[w % n_gpus];
        while (!failed) {
            std::unique_ptr<Ticket> t;
            {
                std::unique_lock<std::mutex> lk(m_in);
                cv_in_empty.wait(lk, [&]() { return !queue.empty() || reader_done || failed; });
                if (failed || queue.empty()) return;
                size_t pick = 0; for (size_t i = 1; i < queue.size(); ++i) if (queue[i]->id < queue[pick]->id) pick = i; // (the parser threads deliver out of order: the ticket the writer waits for goes first)
                t = std::move(queue[pick]); queue.erase(queue.begin() + static_cast<std::ptrdiff_t>(pick));
                cv_in_full.notify_one();
            }
            { // do not run further ahead of the writer than the block map may grow
                std::unique_lock<std::mutex> lk(m_out);
                cv_out.wait(lk, [&]() { return t->id < next_to_write + ahead_cap || failed; });
                if (failed) return;
            }
            const rtk::PackedReads& R = t->reads;
            const uint32_t n = static_cast<uint32_t>(R.size());
            Fetched f;
            if (n != 0) { // (a byte range of the parallel reader may hold no record start at all)
            std::vector<const char*> ps(n); std::vector<uint32_t> len(n);
            for (uint32_t i = 0; i < n; ++i) { ps[i] = R.seq(i); len[i] = R.seq_len(i); }
            const long long tc0 = now_us();
            rtk_batch* b = nullptr;
            rtk_batch* b_prev = nullptr; // -r N: the previous round's batch (its fetch view is the next round's input)
            int rc;
            if (lrc) {
                static const char none[1] = {0};
                std::vector<const char*> pq(n), pr(n); std::vector<uint32_t> rlen(n);
                for (uint32_t i = 0; i < n; ++i) { pq[i] = R.qual(i) ? R.qual(i) : none; pr[i] = t->raw.seq(i); rlen[i] = t->raw.seq_len(i); }
                rc = rtk_batch_create2(g, n, ps.data(), pq.data(), len.data(), pr.data(), rlen.data(), &b);
            } else rc = rtk_batch_create(g, n, ps.data(), nullptr, len.data(), &b); // pass 1 replaces every quality (src/Correction.cpp:184-185)
            const long long tc1 = now_us();
            const char* pool = nullptr; const uint64_t* off = nullptr; const uint32_t* olen = nullptr;
            const int n_rounds = lrc ? 1 : opt.rounds;
            for (int j = 0; j < n_rounds && rc == RTK_OK; ++j) { // src/Ratatosk.cpp:847-866: every round corrects the output of the one before with its own thresholds
                rtk_opts rj = ro;
                if (n_rounds > 1) {
                    const double step_min_score = 1.0 / static_cast<double>(n_rounds);
                    const double step_f = (ro.weak_region_len_factor - 0.10) / static_cast<double>(n_rounds - 1);
                    const uint64_t step_w = ro.max_len_weak_region1 / static_cast<uint64_t>(n_rounds);
                    rj.min_score = 1.0 - static_cast<double>(j + 1) * step_min_score;
                    rj.weak_region_len_factor = ro.weak_region_len_factor - static_cast<double>(n_rounds - j - 1) * step_f;
                    rj.max_len_weak_region1 = static_cast<uint64_t>(j + 1) * step_w;
                }
                if (j > 0) { // the reads of this round = the records of the last one
                    for (uint32_t i = 0; i < n; ++i) { ps[i] = pool + off[i]; len[i] = olen[i]; }
                    b_prev = b; b = nullptr;
                    rc = rtk_batch_create(g, n, ps.data(), nullptr, len.data(), &b);
                    rtk_batch_free(b_prev); b_prev = nullptr;
                    if (rc != RTK_OK) break;
                }
                rc = rtk_batch_run(b, &rj);
                const long long tc2 = now_us();
                if (rc == RTK_OK) rc = rtk_batch_fetch_view(b, &pool, &off, &olen);
                if (cli_trace) fprintf(stderr, "[cli trace] ticket %zu worker %d: start %+.1f ms, create %.1f, run %.1f, fetch %.1f ms\n", t->id, w, 1e-3 * (tc0 - t_begin), 1e-3 * (tc1 - tc0), 1e-3 * (tc2 - tc1), 1e-3 * (now_us() - tc2));
            }
            us_correct += now_us() - tc0;
            if (rc != RTK_OK) { fail(std::string("Ratatosk::correct(): ") + rtk_last_error()); if (b) rtk_batch_free(b); return; }
            f.b = b; f.pool = pool; f.off = off; f.olen = olen;
            }
            f.t = std::move(t);
            { // hand the fetched ticket to the formatters and go on with the next one: the GPU side never waits for memcpy, gzip or the file system
                std::unique_lock<std::mutex> lk(m_f);
                cv_f_full.wait(lk, [&]() { return n_fetched_alive < fetched_cap || failed; });
                if (failed) { if (f.b) rtk_batch_free(f.b); return; }
                ++n_fetched_alive; fq.push_back(std::move(f));
            }
            cv_f_empty.notify_one();
        }
    };
    auto formatter = [&]() {
        for (;;) {
            Fetched f;
            {
                std::unique_lock<std::mutex> lk(m_f);
                cv_f_empty.wait(lk, [&]() { return !fq.empty() || gpu_workers_left == 0 || failed; });
                if (fq.empty()) return; // (after a failure: the blocks left in the queue are dropped with the run)
                f = std::move(fq.front()); fq.pop_front();
            }
            const rtk::PackedReads& R = f.t->reads;
            const uint32_t n = static_cast<uint32_t>(R.size());
            const char* pool = f.pool; const uint64_t* off = f.off; const uint32_t* olen = f.olen;
            std::string block;
            const long long tf0 = now_us();
            if (n != 0 && !failed) {
            if (trim) {
                for (uint32_t i = 0; i < n; ++i) append_trimmed(block, R.name(i), R.name_len(i), pool + off[i], pool + off[i] + olen[i], olen[i], k_graph, trim);
            } else {
              { size_t tot = 0; for (uint32_t i = 0; i < n; ++i) tot += R.name_len(i) + 2ull * olen[i] + 6; block.resize(tot); }
              // writeCorrectedOutput, trim == 0 (src/Ratatosk.cpp:516-520): "@name\nseq\n+\nqual\n"
                char* p = &block[0];
                for (uint32_t i = 0; i < n; ++i) {
                    *p++ = '@'; memcpy(p, R.name(i), R.name_len(i)); p += R.name_len(i); *p++ = '\n';
                    memcpy(p, pool + off[i], olen[i]); p += olen[i]; *p++ = '\n'; *p++ = '+'; *p++ = '\n';
                    memcpy(p, pool + off[i] + olen[i], olen[i]); p += olen[i]; *p++ = '\n';
                }
            }
            }
            if (f.b) rtk_batch_free(f.b);
            { std::lock_guard<std::mutex> lk(m_f); --n_fetched_alive; } cv_f_full.notify_one(); // the batch (device buffers, pinned view) is free again: gzip and the write do not hold it
            if (failed) continue;
            if (gz_out && n != 0) { std::string z; if (!gzip_member(block, z)) { fail("Ratatosk::search(): gzip compression failed"); return; } block.swap(z); }
            us_format += now_us() - tf0;
            std::vector<std::pair<unsigned long long, std::string> > to_write; // blocks that got their offset by this ticket's arrival (its own, and later ones that were waiting for it)
            {
                std::unique_lock<std::mutex> lk(m_out);
                done[f.t->id].swap(block);
                while (!done.empty() && done.begin()->first == next_to_write) {
                    to_write.emplace_back(out_off, std::string()); to_write.back().second.swap(done.begin()->second);
                    out_off += to_write.back().second.size();
                    done.erase(done.begin()); ++next_to_write;
                }
            }
            cv_out.notify_all();
            const long long tw0 = now_us();
            for (size_t i = 0; i < to_write.size(); ++i) {
                const std::string& blk = to_write[i].second; size_t w_done = 0;
                while (w_done < blk.size()) {
                    const ssize_t nw = pwrite(fd_out, blk.data() + w_done, blk.size() - w_done, static_cast<off_t>(to_write[i].first + w_done));
                    if (nw <= 0) { fail("Ratatosk::search(): write error on " + fn_out); return; }
                    w_done += static_cast<size_t>(nw);
                }
            }
            us_write += now_us() - tw0;
        }
    };
    std::vector<std::thread> fmt_threads;
    for (int i = 0; i < n_fmt; ++i) fmt_threads.emplace_back(formatter);
    std::vector<std::thread> th;
    for (int w = 0; w < n_workers; ++w) th.emplace_back(worker, w);
    for (size_t i = 0; i < th.size(); ++i) th[i].join();
    { std::lock_guard<std::mutex> lk(m_f); gpu_workers_left = 0; } cv_f_empty.notify_all();
    for (size_t i = 0; i < fmt_threads.size(); ++i) fmt_threads[i].join();
    { std::lock_guard<std::mutex> lk(m_in); } cv_in_full.notify_all();
    { std::lock_guard<std::mutex> lk(m_half); } cv_half_full.notify_all(); cv_half_empty.notify_all();
    reader_thread.join(); raw_thread.join();
    for (size_t i = 0; i < parser_threads.size(); ++i) parser_threads[i].join();
    if (gz_out && !failed) { std::string e; rtk::bgzf_append_eof(e); if (pwrite(fd_out, e.data(), e.size(), static_cast<off_t>(out_off)) != static_cast<ssize_t>(e.size())) fail("Ratatosk::search(): write error on " + fn_out); }
    const bool write_ok = ::close(fd_out) == 0;
    const double wall = 1e-6 * (now_us() - t_begin); // the correction phase ends with the output file closed (the device images and pinned pools are torn down after it)
    for (int w = 0; w < n_gpus; ++w) rtk_graph_free(graphs[w]);
    if (failed || !write_ok || !done.empty()) { // a partial OUT.2.fastq must not look like a result
        remove(fn_out.c_str());
        fprintf(stderr, "%s\n", fail_msg.empty() ? "Ratatosk::search(): output incomplete" : fail_msg.c_str());
        exit(1);
    }
    if (opt.verbose || getenv("RTK_CLI_STATS"))
        fprintf(opt.verbose ? stdout : stderr, "Ratatosk::correct(): graph load + upload %.2f s; correction phase %.2f s wall, %lld bases, %.3g bases/s on %d GPU(s) x %d workers; "
                "thread-seconds: parse %.2f, correct (pack + GPU + fetch) %.2f, format %.2f, write %.2f\n", 1e-6 * (t_load1 - t_load0), wall, n_bases.load(),
                wall > 0 ? static_cast<double>(n_bases.load()) / wall : 0.0, n_gpus, opt.workers_per_gpu, 1e-6 * us_parse.load(), 1e-6 * us_correct.load(), 1e-6 * us_format.load(), 1e-6 * us_write.load());
    return 0;
}
