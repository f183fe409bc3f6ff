// `Ratatosk` host driver for the in-scope branch of the reference CLI: `Ratatosk correct -1 -g G -d D -l reads -o OUT`
// (reference: src/Ratatosk.cpp:145-301 option table, :303-508 validation, :618-1000 search(), :1029-1037 file names,
// :1083-1095/:1145-1149 pass-1 branch). C++11 host orchestration over the C ABI of libratatosk_hip.so: one worker thread
// per GPU (two, so that consecutive batches overlap) pulls read batches by ticket, corrects them on its device, and the writer emits blocks in ticket order (pass-1
// output is always in input order: src/Ratatosk.cpp:919). Everything else (`index`, `-2`, `-u`, `-p/-P`) is out of scope.
#include <getopt.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../common/fastx.hpp"
#include "ratatosk_hip.h"

struct Opt {
    std::vector<std::string> in_long;
    std::string out, graph, udata;
    int cores = 1, k1 = 31, max_qual = 40;
    double min_conf_snp = 0.9;
    size_t insert_sz = 500, w1 = 1000, batch_bases = 64u << 20;
    bool pass1 = false, pass2 = false, verbose = false, correct = false, strip = false;
};

static void usage() {
    fprintf(stderr, "Ratatosk (MI355X hot-path build)\n\nUsage: Ratatosk correct -1 -g <graph.fasta.gz> -d <unitig_data.rtsk> -l <long_reads> -o <out_prefix> [options]\n"
                    "  -c, --cores           number of GPUs/worker threads to use (default 1)\n  -i, --insert-sz       insert size of the short reads (default 500)\n"
                    "  -k, --k1              k-mer length of the 1st pass graph (default 31, <= 31)\n  -w, --max-len-weak1   maximum weak region length, 1st pass (default 1000)\n"
                    "  -Q, --max-base-qual   maximum base quality (default 40)\n  -m, --min-conf-snp-corr  minimum confidence threshold to correct a SNP (default 0.9)\n  -v, --verbose\n"
                    "      --strip-annotations  drop the short-cycle / SNP annotations of the index before correcting (fixRepeats / fixAmbiguity then have nothing to do)\n"
                    "Writes <out_prefix>.2.fastq (plain FASTQ, input order). Only the `correct -1` step with a pre-built index is in scope.\n");
}

int main(int argc, char** argv) {
    Opt opt;
    if (argc <= 1 || !strcmp(argv[1], "--help")) { usage(); return 0; }
    if (!strcmp(argv[1], "--version")) { printf("%s\n", rtk_version()); return 0; }
    if (!strcmp(argv[1], "correct")) opt.correct = true;
    else if (!strcmp(argv[1], "index")) { fprintf(stderr, "Ratatosk::index: not in scope of this build (use the reference `index` step; its files are read as-is)\n"); return 1; }
    else { usage(); return 0; }
    static struct option lo[] = {{"in-long", required_argument, 0, 'l'}, {"out-long", required_argument, 0, 'o'}, {"cores", required_argument, 0, 'c'},
        {"in-graph", required_argument, 0, 'g'}, {"in-unitig-data", required_argument, 0, 'd'}, {"insert-sz", required_argument, 0, 'i'}, {"k1", required_argument, 0, 'k'},
        {"max-len-weak1", required_argument, 0, 'w'}, {"max-base-qual", required_argument, 0, 'Q'}, {"min-conf-snp-corr", required_argument, 0, 'm'}, {"1st-pass-only", no_argument, 0, '1'}, {"2nd-pass-only", no_argument, 0, '2'},
        {"batch-bases", required_argument, 0, 'B'}, {"strip-annotations", no_argument, 0, 1001}, {"verbose", no_argument, 0, 'v'}, {0, 0, 0, 0}};
    int c, idx = 0;
    while ((c = getopt_long(argc - 1, argv + 1, "s:l:o:c:g:d:i:k:w:Q:m:B:12v", lo, &idx)) != -1) {
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
            case '1': opt.pass1 = true; break;
            case '2': opt.pass2 = true; break;
            case 'v': opt.verbose = true; break;
            case 1001: opt.strip = true; break;
            case 's': fprintf(stderr, "Ratatosk::correct: short reads are only needed by `index` (not in scope); ignored\n"); break;
            default: usage(); return 0; // the reference returns 0 on option errors too (src/Ratatosk.cpp:1018)
        }
    }
    if (opt.pass2 || !opt.pass1) { fprintf(stderr, "Ratatosk::correct: only the first pass (-1) with a pre-built index (-g, -d) is in scope of this build\n"); return 1; }
    if (opt.graph.empty() || opt.udata.empty() || opt.in_long.empty() || opt.out.empty()) { fprintf(stderr, "Ratatosk::correct: -g, -d, -l and -o are required\n"); return 0; }
    if (opt.cores < 1) opt.cores = 1;
    if (opt.min_conf_snp < 0.0 || opt.min_conf_snp > 1.0) { fprintf(stderr, "Ratatosk::Ratatosk(): Minimum confidence threshold to correct a SNP must be in [0.0, 1.0].\n"); return 0; } // src/Ratatosk.cpp:366-376

    // input files (a text file lists one path per line: src/Common.cpp:428-446)
    std::vector<std::string> files;
    for (size_t i = 0; i < opt.in_long.size(); ++i) { const std::vector<std::string> v = rtk::expand_input_list(opt.in_long[i]); files.insert(files.end(), v.begin(), v.end()); }

    if (opt.verbose) printf("Ratatosk::Ratatosk(): Reading graph.\n");
    // -c N = N GPUs. Two host workers per GPU share its graph: while one batch is in its region stage the next one runs its seed
    // stage on its own stream (rtk_correct_batch = create + seeds + regions + fetch; the stages of different batches overlap).
    const int n_gpus = opt.cores, n_workers = 2 * opt.cores;
    std::vector<rtk_graph*> graphs(n_gpus, nullptr);
    // the per-wave scratch slabs (tens of GB per GPU, seconds of hipMalloc) are reserved while the index files are parsed
    std::vector<std::thread> reservers;
    for (int w = 0; w < n_gpus; ++w) reservers.emplace_back([w]() { rtk_reserve_scratch(w, 131072u); });
    for (int w = 0; w < n_gpus; ++w) {
        bool ok = rtk_graph_load(opt.graph.c_str(), opt.udata.c_str(), opt.k1, 1, &graphs[w]) == RTK_OK;
        if (ok && opt.strip) { const long long ns = rtk_graph_strip_annotations(graphs[w]); if (w == 0 && ns > 0) fprintf(stderr, "Ratatosk::Ratatosk(): dropped the short-cycle / SNP annotations of %lld unitigs\n", ns); }
        if (!ok || rtk_graph_upload(graphs[w], w) != RTK_OK) {
            fprintf(stderr, "Ratatosk::Ratatosk(): %s\n", rtk_last_error());
            exit(1);
        }
    }
    for (size_t i = 0; i < reservers.size(); ++i) reservers[i].join();
    rtk_opts ro; rtk_opts_default(graphs[0], &ro);
    ro.insert_sz = opt.insert_sz; ro.max_len_weak_region1 = opt.w1; ro.max_qual = opt.max_qual; ro.min_confidence_snp_corr = opt.min_conf_snp;

    const std::string fn_out = opt.out + ".2.fastq"; // opt_pass1.filename_long_out += ".2" (src/Ratatosk.cpp:1079) + ".fastq" (:622)
    FILE* fout = fopen(fn_out.c_str(), "w");
    if (!fout) { fprintf(stderr, "Ratatosk::search(): cannot open %s for writing\n", fn_out.c_str()); exit(1); }

    if (opt.verbose) printf("Ratatosk::Ratatosk(): Correcting long reads (1/2).\n");
    std::mutex m_in, m_out; std::condition_variable cv_out;
    rtk::FastxReader reader; size_t file_i = 0; bool file_open = false, stop = false;
    size_t ticket_dispenser = 0, next_to_write = 0, n_reads = 0;
    std::map<size_t, std::string> done; // ticket -> formatted FASTQ block
    bool failed = false;
    std::atomic<long long> us_parse(0), us_correct(0), us_format(0);
    auto now_us = []() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const long long t_begin = now_us();

    auto worker = [&](int w) {
        while (true) {
            std::vector<std::string> names, seqs, quals; size_t ticket, bases = 0;
            {
                std::lock_guard<std::mutex> lk(m_in);
                if (stop) return;
                const long long tp0 = now_us();
                ticket = ticket_dispenser++;
                std::string n, s, q;
                while (bases < opt.batch_bases) {
                    if (!file_open) { if (file_i >= files.size()) { stop = true; break; } if (!reader.open(files[file_i++])) { fprintf(stderr, "Ratatosk::search(): cannot open input file\n"); exit(1); } file_open = true; }
                    if (!reader.next(n, s, q)) { file_open = false; continue; }
                    bases += s.size(); names.push_back(n); seqs.push_back(s); quals.push_back(q);
                    if (opt.verbose && (++n_reads % 1000 == 0)) printf("Ratatosk::correct(): Processed %zu reads \n", n_reads);
                }
                us_parse += now_us() - tp0;
            }
            std::string block;
            if (!seqs.empty()) {
                const uint32_t n = static_cast<uint32_t>(seqs.size());
                std::vector<const char*> ps(n), pq(n); std::vector<uint32_t> len(n), olen(n); std::vector<char*> os(n, nullptr), oq(n, nullptr);
                for (uint32_t i = 0; i < n; ++i) { ps[i] = seqs[i].c_str(); pq[i] = quals[i].empty() ? nullptr : quals[i].c_str(); len[i] = static_cast<uint32_t>(seqs[i].size()); }
                const long long tc0 = now_us();
                const int rc_ = rtk_correct_batch(graphs[w % n_gpus], &ro, n, ps.data(), pq.data(), len.data(), os.data(), oq.data(), olen.data());
                us_correct += now_us() - tc0;
                const long long tf0 = now_us();
                if (rc_ != RTK_OK) {
                    fprintf(stderr, "Ratatosk::correct(): %s\n", rtk_last_error()); failed = true;
                } else {
                    for (uint32_t i = 0; i < n; ++i) { // writeCorrectedOutput, trim == 0 (src/Ratatosk.cpp:516-520)
                        block += "@"; block += names[i]; block += "\n"; block.append(os[i], olen[i]); block += "\n+\n"; block += oq[i]; block += "\n";
                        rtk_free(os[i]); rtk_free(oq[i]);
                    }
                }
                us_format += now_us() - tf0;
            }
            {
                std::unique_lock<std::mutex> lk(m_out);
                done[ticket] = block;
                while (!done.empty() && done.begin()->first == next_to_write) { fwrite(done.begin()->second.data(), 1, done.begin()->second.size(), fout); done.erase(done.begin()); ++next_to_write; }
            }
            if (failed) return;
        }
    };
    std::vector<std::thread> th;
    for (int w = 0; w < n_workers; ++w) th.emplace_back(worker, w);
    for (size_t i = 0; i < th.size(); ++i) th[i].join();
    for (std::map<size_t, std::string>::iterator it = done.begin(); it != done.end(); ++it) fwrite(it->second.data(), 1, it->second.size(), fout);
    fclose(fout);
    if (opt.verbose) printf("Ratatosk::correct(): correction phase %.2f s wall; summed over workers: parse %.2f s, correct (pack + GPU + unpack) %.2f s, format %.2f s\n",
                            1e-6 * (now_us() - t_begin), 1e-6 * us_parse.load(), 1e-6 * us_correct.load(), 1e-6 * us_format.load());
    for (int w = 0; w < n_gpus; ++w) rtk_graph_free(graphs[w]);
    if (failed) exit(1);
    return 0;
}
