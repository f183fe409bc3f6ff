// rtk_bgzip IN OUT [-@ THREADS] [-l LEVEL]: IN (plain file) as blocked gzip (BGZF, common/bgzf.hpp). A stand-in for htslib's bgzip where that is not
// installed: test inputs and I/O measurements of the first-pass reader. Pieces of 16 MiB are compressed side by side and written in order.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../common/bgzf.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: rtk_bgzip IN OUT [-@ threads] [-l level]\n"); return 1; }
    int threads = 8, level = 1;
    for (int i = 3; i + 1 < argc; i += 2) { if (!strcmp(argv[i], "-@")) threads = atoi(argv[i + 1]); else if (!strcmp(argv[i], "-l")) level = atoi(argv[i + 1]); }
    const int fd = open(argv[1], O_RDONLY); if (fd < 0) { fprintf(stderr, "rtk_bgzip: cannot open %s\n", argv[1]); return 1; }
    struct stat st; if (fstat(fd, &st) != 0) return 1;
    const size_t n = static_cast<size_t>(st.st_size);
    const char* in = n ? static_cast<const char*>(mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0)) : "";
    if (n && in == MAP_FAILED) { fprintf(stderr, "rtk_bgzip: cannot map %s\n", argv[1]); return 1; }
    const size_t piece = rtk::BGZF_BLOCK_BYTES * 256; // a whole number of blocks, so the result does not depend on the number of threads
    const size_t n_pieces = (n + piece - 1) / piece;
    std::vector<std::string> out(n_pieces); std::atomic<size_t> next(0); std::atomic<bool> bad(false);
    std::vector<std::thread> th;
    for (int t = 0; t < std::max(1, threads); ++t) th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n_pieces) break; if (!rtk::bgzf_compress(in + i * piece, std::min(piece, n - i * piece), out[i], level)) bad = true; } });
    for (size_t t = 0; t < th.size(); ++t) th[t].join();
    if (bad) { fprintf(stderr, "rtk_bgzip: compression failed\n"); return 1; }
    FILE* fo = fopen(argv[2], "wb"); if (!fo) { fprintf(stderr, "rtk_bgzip: cannot write %s\n", argv[2]); return 1; }
    for (size_t i = 0; i < n_pieces; ++i) if (fwrite(out[i].data(), 1, out[i].size(), fo) != out[i].size()) { fprintf(stderr, "rtk_bgzip: write error\n"); return 1; }
    std::string e; rtk::bgzf_append_eof(e); fwrite(e.data(), 1, e.size(), fo);
    return fclose(fo) == 0 ? 0 : 1;
}
