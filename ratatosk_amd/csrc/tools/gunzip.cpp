// rtk_gunzip IN [-@ threads] [-o OUT] [--check]: the text of a gzip file through the reader the host driver uses (common/mgzip.hpp: members
// inflated on several threads by common/finflate.hpp, chain of members, CRC-32 of every member checked). Writes the text to OUT (default:
// standard output), or with --check only its length and CRC-32 ("bytes crc members"). Exit code 1 on damaged input. For the test tier
// (tests/test_inflate.py) and as the counterpart of rtk_bgzip.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../common/mgzip.hpp"

int main(int argc, char** argv) {
    std::string in, out; int threads = 4; bool check = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-@" && i + 1 < argc) threads = atoi(argv[++i]);
        else if (a == "-o" && i + 1 < argc) out = argv[++i];
        else if (a == "--check") check = true;
        else in = a;
    }
    if (in.empty()) { fprintf(stderr, "usage: rtk_gunzip IN [-@ threads] [-o OUT] [--check]\n"); return 2; }
    rtk::MemberGzipReader r;
    if (!r.open(in, threads)) { fprintf(stderr, "rtk_gunzip: %s is not a gzip file\n", in.c_str()); return 1; }
    FILE* f = check ? nullptr : (out.empty() ? stdout : fopen(out.c_str(), "wb"));
    if (!check && !f) { fprintf(stderr, "rtk_gunzip: cannot write %s\n", out.c_str()); return 1; }
    std::vector<char> buf(1 << 20);
    unsigned long long total = 0; uint32_t crc = 0; long n;
    while ((n = r.read(buf.data(), buf.size())) > 0) {
        total += static_cast<unsigned long long>(n);
        if (check) crc = rtk::fast_crc32(crc, reinterpret_cast<const unsigned char*>(buf.data()), static_cast<size_t>(n));
        else if (fwrite(buf.data(), 1, static_cast<size_t>(n), f) != static_cast<size_t>(n)) { fprintf(stderr, "rtk_gunzip: write error\n"); return 1; }
    }
    if (f && f != stdout) fclose(f);
    if (n < 0 || r.failed()) { fprintf(stderr, "rtk_gunzip: %s ends in a damaged or cut-short gzip stream (%llu bytes were good)\n", in.c_str(), total); return 1; }
    if (check) printf("%llu %08x %zu\n", total, crc, r.members());
    return 0;
}
