// FASTA/FASTQ (optionally gzip) record reader and writer for the product host code.
//
// Record conventions follow what the reference expects from Bifrost's FileParser
// (reference: src/Ratatosk.cpp:658,767 — name = header up to the first whitespace, quality only for
// FASTQ) and what it writes (src/Ratatosk.cpp:510-520 — "@name\nseq\n+\nqual\n").
// Multi-line FASTA is supported; FASTQ is the 4-line form.
#ifndef RTK_COMMON_FASTX_HPP
#define RTK_COMMON_FASTX_HPP

#include <zlib.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rtk {

// A batch of reads in ONE buffer (names, sequences and, on request, qualities back to back) + offsets: what a ticket of the host
// driver carries from the reader thread to a GPU worker without per-record strings.
class PackedReads {
public:
    explicit PackedReads(bool keep_qual = false) : keep_qual_(keep_qual), n_bases_(0) {}
    void reserve(size_t bytes) { buf_.reserve(bytes); }
    size_t size() const { return rec_.size(); }
    size_t n_bases() const { return n_bases_; }
    bool keeps_qual() const { return keep_qual_; }
    const char* name(size_t i) const { return buf_.data() + rec_[i].name_off; }
    uint32_t name_len(size_t i) const { return rec_[i].name_len; }
    const char* seq(size_t i) const { return buf_.data() + rec_[i].seq_off; }
    uint32_t seq_len(size_t i) const { return rec_[i].seq_len; }
    const char* qual(size_t i) const { return rec_[i].has_qual ? buf_.data() + rec_[i].seq_off + rec_[i].seq_len : nullptr; } // seq_len characters
private:
    friend class FastxReader;
    struct Rec { size_t name_off, seq_off; uint32_t name_len, seq_len; bool has_qual; };
    std::vector<char> buf_;
    std::vector<Rec> rec_;
    bool keep_qual_;
    size_t n_bases_;
};

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

    // Same record conventions as next(), appended to `out` (sequence lines of a multi-line FASTA record are concatenated in place).
    bool next_packed(PackedReads& out) {
        std::vector<char>& b = out.buf_;
        size_t h0;
        while (true) { // header
            h0 = b.size();
            if (!getline_append(b)) return false;
            if (b.size() > h0 && (b[h0] == '>' || b[h0] == '@')) break;
            b.resize(h0);
        }
        const bool fastq = (b[h0] == '@');
        size_t e = h0 + 1;
        while (e < b.size() && !isspace(static_cast<unsigned char>(b[e]))) ++e;
        memmove(&b[h0], &b[h0 + 1], e - (h0 + 1)); // drop the marker, keep the name
        PackedReads::Rec r; r.name_off = h0; r.name_len = static_cast<uint32_t>(e - (h0 + 1)); r.has_qual = false;
        b.resize(h0 + r.name_len);
        r.seq_off = b.size();
        if (fastq) {
            if (!getline_append(b)) { b.resize(h0); return false; }
            r.seq_len = static_cast<uint32_t>(b.size() - r.seq_off);
            const size_t p0 = b.size();
            if (!getline_append(b)) { b.resize(h0); return false; } // '+'
            b.resize(p0);
            if (!getline_append(b)) { b.resize(h0); return false; }
            if (out.keep_qual_ && b.size() - p0 == r.seq_len) r.has_qual = true; else b.resize(p0);
        } else {
            while (true) {
                const size_t p0 = b.size();
                if (!getline_append(b)) break;
                if (b.size() > p0 && (b[p0] == '>' || b[p0] == '@')) { peek_.assign(&b[p0], b.size() - p0); has_peek_ = true; b.resize(p0); break; }
            }
            r.seq_len = static_cast<uint32_t>(b.size() - r.seq_off);
        }
        out.rec_.push_back(r); out.n_bases_ += r.seq_len;
        return true;
    }

private:
    bool getline_append(std::vector<char>& out) {
        if (has_peek_) { out.insert(out.end(), peek_.begin(), peek_.end()); has_peek_ = false; return true; }
        const size_t start = out.size();
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
            if (p) { out.insert(out.end(), static_cast<const char*>(buf_ + pos_), p); pos_ = static_cast<size_t>(p - buf_) + 1; break; }
            out.insert(out.end(), buf_ + pos_, buf_ + end_);
            pos_ = end_;
        }
        if (out.size() > start && out.back() == '\r') out.pop_back();
        return got;
    }

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
    char buf_[1 << 18];
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
