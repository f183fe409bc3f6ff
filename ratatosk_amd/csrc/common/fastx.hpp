// FASTA/FASTQ (optionally gzip) record reader and writer for the product host code.
//
// Record conventions follow what the reference expects from Bifrost's FileParser
// (reference: src/Ratatosk.cpp:658,767 — name = header up to the first whitespace, quality only for
// FASTQ) and what it writes (src/Ratatosk.cpp:510-520 — "@name\nseq\n+\nqual\n").
// Multi-line FASTA is supported; FASTQ is the 4-line form.
#ifndef RTK_COMMON_FASTX_HPP
#define RTK_COMMON_FASTX_HPP

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rtk {

class FastxReader {
public:
    FastxReader() : fp_(nullptr), pos_(0), end_(0), eof_(false), has_peek_(false) {}
    ~FastxReader() { close(); }

    bool open(const std::string& fn) {
        close();
        fp_ = gzopen(fn.c_str(), "rb"); // zlib reads plain files transparently
        if (!fp_) return false;
        gzbuffer(fp_, 1 << 20);
        pos_ = end_ = 0; eof_ = false; has_peek_ = false;
        return true;
    }

    void close() { if (fp_) { gzclose(fp_); fp_ = nullptr; } }

    // Reads next record. qual is cleared for FASTA records.
    bool next(std::string& name, std::string& seq, std::string& qual) {
        std::string line;
        name.clear(); seq.clear(); qual.clear();
        // find header
        while (true) {
            if (!getline(line)) return false;
            if (!line.empty() && (line[0] == '>' || line[0] == '@')) break;
        }
        const bool fastq = (line[0] == '@');
        size_t e = 1;
        while (e < line.size() && !isspace(static_cast<unsigned char>(line[e]))) ++e;
        name.assign(line, 1, e - 1);
        if (fastq) {
            if (!getline(seq)) return false;
            if (!getline(line)) return false; // '+'
            if (!getline(qual)) return false;
            return true;
        }
        // FASTA: concatenate lines until next header
        while (true) {
            if (!getline(line)) break;
            if (!line.empty() && (line[0] == '>' || line[0] == '@')) { peek_ = line; has_peek_ = true; break; }
            seq += line;
        }
        return true;
    }

private:
    bool getline(std::string& out) {
        if (has_peek_) { out.swap(peek_); has_peek_ = false; return true; }
        out.clear();
        bool got = false;
        while (true) {
            if (pos_ == end_) {
                if (eof_) break;
                const int n = gzread(fp_, buf_, sizeof(buf_));
                if (n <= 0) { eof_ = true; break; }
                pos_ = 0; end_ = static_cast<size_t>(n);
            }
            const char* p = static_cast<const char*>(memchr(buf_ + pos_, '\n', end_ - pos_));
            got = true;
            if (p) {
                out.append(buf_ + pos_, p - (buf_ + pos_));
                pos_ = static_cast<size_t>(p - buf_) + 1;
                break;
            }
            out.append(buf_ + pos_, end_ - pos_);
            pos_ = end_;
        }
        if (!out.empty() && out[out.size() - 1] == '\r') out.erase(out.size() - 1);
        return got;
    }

    gzFile fp_;
    char buf_[1 << 16];
    size_t pos_, end_;
    bool eof_;
    std::string peek_;
    bool has_peek_;
};

// Reads a text file listing one path per line if `fn` is not itself FASTA/FASTQ
// (reference: src/Common.cpp:412-446 treats non-FASTX inputs as lists of files).
inline std::vector<std::string> expand_input_list(const std::string& fn) {
    std::vector<std::string> out;
    gzFile f = gzopen(fn.c_str(), "rb");
    if (!f) { out.push_back(fn); return out; }
    char c = 0;
    const int n = gzread(f, &c, 1);
    gzclose(f);
    if (n == 1 && (c == '>' || c == '@')) { out.push_back(fn); return out; }
    FILE* t = fopen(fn.c_str(), "r");
    if (!t) { out.push_back(fn); return out; }
    char line[4096];
    while (fgets(line, sizeof(line), t)) {
        std::string s(line);
        while (!s.empty() && (s[s.size() - 1] == '\n' || s[s.size() - 1] == '\r' || s[s.size() - 1] == ' ')) s.erase(s.size() - 1);
        if (!s.empty()) out.push_back(s);
    }
    fclose(t);
    return out;
}

} // namespace rtk

#endif
