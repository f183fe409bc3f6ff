// FASTA/FASTQ (optionally gzip) record reader and writer for the product host code.
//
// Record conventions follow what the reference expects from Bifrost's FileParser
// (reference: src/Ratatosk.cpp:658,767 — name = header up to the first whitespace, quality only for
// FASTQ) and what it writes (src/Ratatosk.cpp:510-520 — "@name\nseq\n+\nqual\n").
// Multi-line FASTA is supported. FASTQ follows kseq (what Bifrost's FileParser reads with): sequence lines up to the line that
// starts with '+', then quality lines until the quality is as long as the sequence. The byte-range reader (PlainChunks) only takes the
// 4-line form: a file whose first record is laid out differently goes through the one-thread reader (is_plain() says no), and a
// record that is not 4-line deeper inside a file makes parse_chunk() fail instead of being misread.
#ifndef RTK_COMMON_FASTX_HPP
#define RTK_COMMON_FASTX_HPP

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "bgzf.hpp"
#include "mgzip.hpp"
#include "sample_source.hpp"

namespace rtk {

// A batch of reads in ONE buffer (names, sequences and, on request, qualities back to back) + offsets: what a ticket of the host
// driver carries from the reader thread to a GPU worker without per-record strings.
// big character buffers travel from the reader to the workers and back: a fresh 100 MB vector costs its page faults every time
inline std::vector<std::vector<char> >& packed_pool_() { static std::vector<std::vector<char> > p; return p; }
inline std::mutex& packed_pool_lock_() { static std::mutex m; return m; }
inline std::vector<char> packed_buffer_take() {
    std::lock_guard<std::mutex> lk(packed_pool_lock_());
    std::vector<std::vector<char> >& p = packed_pool_();
    if (p.empty()) return std::vector<char>();
    std::vector<char> v; v.swap(p.back()); p.pop_back(); v.clear(); return v;
}
inline void packed_buffer_give(std::vector<char>& v) {
    if (v.capacity() < (1u << 20)) return;
    std::lock_guard<std::mutex> lk(packed_pool_lock_());
    if (packed_pool_().size() < 64) { packed_pool_().emplace_back(); packed_pool_().back().swap(v); }
}

class PackedReads {
public:
    explicit PackedReads(bool keep_qual = false) : keep_qual_(keep_qual), n_bases_(0), malformed_(false) {}
    ~PackedReads() { packed_buffer_give(buf_); }
    PackedReads(const PackedReads&) = delete; PackedReads& operator=(const PackedReads&) = delete;
    PackedReads(PackedReads&& o) : buf_(std::move(o.buf_)), rec_(std::move(o.rec_)), keep_qual_(o.keep_qual_), n_bases_(o.n_bases_), malformed_(o.malformed_) {}
    PackedReads& operator=(PackedReads&& o) { packed_buffer_give(buf_); buf_ = std::move(o.buf_); rec_ = std::move(o.rec_); keep_qual_ = o.keep_qual_; n_bases_ = o.n_bases_; malformed_ = o.malformed_; return *this; }
    void reserve(size_t bytes) { buf_.reserve(bytes); }
    size_t size() const { return rec_.size(); }
    size_t n_bases() const { return n_bases_; }
    bool keeps_qual() const { return keep_qual_; }
    bool malformed() const { return malformed_; } // set by PlainChunks::parse_chunk: a FASTQ record that is not on four lines
    const char* name(size_t i) const { return buf_.data() + rec_[i].name_off; }
    uint32_t name_len(size_t i) const { return rec_[i].name_len; }
    const char* seq(size_t i) const { return buf_.data() + rec_[i].seq_off; }
    uint32_t seq_len(size_t i) const { return rec_[i].seq_len; }
    const char* qual(size_t i) const { return rec_[i].has_qual ? buf_.data() + rec_[i].qual_off : nullptr; } // seq_len characters
private:
    friend class FastxReader;
    friend class PlainChunks;
    struct Rec { size_t name_off, seq_off, qual_off; uint32_t name_len, seq_len; bool has_qual; };
    std::vector<char> buf_;
    std::vector<Rec> rec_;
    bool keep_qual_;
    size_t n_bases_;
    bool malformed_; // the byte-range parser met a FASTQ record that is not on four lines
};

class FastxReader {
public:
    FastxReader() : fp_(nullptr), failed_(false), pos_(0), end_(0), eof_(false), has_peek_(false) {}
    ~FastxReader() { close(); }

    // inflate_threads >= 1: a gzip file is inflated by the reader's own decoder on that many threads BESIDE the calling one (mgzip.hpp: a file of
    // several gzip members, the usual `cat *.fastq.gz`, scales with them; a single member streams, inflate and parsing side by side).
    // 0: zlib's gzread on the calling thread.
    bool open(const std::string& fn, int inflate_threads = 0) {
        close();
        pos_ = end_ = 0; eof_ = false; has_peek_ = false; failed_ = false;
        if (SampleSource::is_spec(fn)) { // reads sampled from a reference on the fly (sample_source.hpp): delivered as the FASTQ text they stand for
            std::string err; ss_ = SampleSource::get(fn, &err); ss_next_ = 0; ss_pend_.clear(); ss_pend_off_ = 0;
            if (!ss_) fprintf(stderr, "%s\n", err.c_str());
            return ss_ != nullptr;
        }
        if (inflate_threads >= 1 && MemberGzipReader::looks_like_gzip(fn)) { mg_.reset(new MemberGzipReader()); if (mg_->open(fn, inflate_threads)) return true; mg_.reset(); }
        fp_ = gzopen(fn.c_str(), "rb"); // zlib reads plain files transparently
        if (!fp_) return false;
        gzbuffer(fp_, 1 << 20);
        return true;
    }

    void close() { if (fp_) { gzclose(fp_); fp_ = nullptr; } mg_.reset(); ss_.reset(); }
    // the input ended on a damaged or cut-short gzip stream (what was read before it has been delivered): callers must not take that for the end of the file
    bool failed() const { return failed_; }

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
        if (fastq) { // kseq: sequence lines up to a line that starts with '+' (or with the next header: a record without quality)
            while (true) {
                if (!getline(line)) return true;
                if (!line.empty() && line[0] == '+') break;
                if (!line.empty() && (line[0] == '>' || line[0] == '@')) { peek_ = line; has_peek_ = true; return true; }
                seq += line;
            }
            do { if (!getline(line)) break; qual += line; } while (qual.size() < seq.size()); // at least one quality line, more while it is short
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
        PackedReads::Rec r; r.name_off = h0; r.name_len = static_cast<uint32_t>(e - (h0 + 1)); r.has_qual = false; r.qual_off = 0;
        b.resize(h0 + r.name_len);
        r.seq_off = b.size();
        if (fastq) { // kseq: sequence lines up to the '+' line, then quality lines until the quality is as long as the sequence
            bool plus = false;
            while (true) {
                const size_t p0 = b.size();
                if (!getline_append(b)) break;
                if (b.size() > p0 && b[p0] == '+') { b.resize(p0); plus = true; break; }
                if (b.size() > p0 && (b[p0] == '>' || b[p0] == '@')) { peek_.assign(&b[p0], b.size() - p0); has_peek_ = true; b.resize(p0); break; }
            }
            r.seq_len = static_cast<uint32_t>(b.size() - r.seq_off);
            const size_t p0 = b.size();
            if (plus) { do { if (!getline_append(b)) break; } while (b.size() - p0 < r.seq_len); }
            r.qual_off = p0;
            if (plus && out.keep_qual_ && b.size() - p0 == r.seq_len) r.has_qual = true; else b.resize(p0);
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
                const long n = fill_();
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
                const long n = fill_();
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

    long fill_() { // next bytes of the text into buf_
        if (ss_) {
            size_t n = 0;
            while (n < sizeof(buf_)) {
                if (ss_pend_off_ == ss_pend_.size()) { // the next pair as two FASTQ records
                    if (ss_next_ >= ss_->n_pairs()) break;
                    const uint32_t L = ss_->read_len();
                    std::string m1(L, 'A'), m2(L, 'A'); ss_->pair(ss_next_, &m1[0], &m2[0]);
                    const std::string nm = "s" + std::to_string(ss_next_), q(L, 'I');
                    ss_pend_ = "@" + nm + "\n" + m1 + "\n+\n" + q + "\n@" + nm + "\n" + m2 + "\n+\n" + q + "\n"; ss_pend_off_ = 0; ++ss_next_;
                }
                const size_t c = std::min(sizeof(buf_) - n, ss_pend_.size() - ss_pend_off_);
                memcpy(buf_ + n, ss_pend_.data() + ss_pend_off_, c); n += c; ss_pend_off_ += c;
            }
            return static_cast<long>(n);
        }
        if (mg_) { const long n = mg_->read(buf_, sizeof(buf_)); if (n < 0 || mg_->failed()) failed_ = true; return n; }
        const int n = gzread(fp_, buf_, sizeof(buf_));
        if (n < static_cast<int>(sizeof(buf_))) { int e = Z_OK; gzerror(fp_, &e); if (n < 0 || (e != Z_OK && e != Z_STREAM_END)) failed_ = true; } // (unexpected end of file: Z_BUF_ERROR)
        return n;
    }
    gzFile fp_;
    std::unique_ptr<MemberGzipReader> mg_;
    std::shared_ptr<SampleSource> ss_; uint64_t ss_next_ = 0; std::string ss_pend_; size_t ss_pend_off_ = 0;
    bool failed_;
    char buf_[1 << 18];
    size_t pos_, end_;
    bool eof_;
    std::string peek_;
    bool has_peek_;
};

// Plain (uncompressed) FASTA / FASTQ file read as independent byte ranges, so that any number of threads can parse one file: range i is
// [i * chunk_bytes, (i + 1) * chunk_bytes) and owns the records that START inside it. A record start is the start of a line that begins
// with '>' (FASTA) or with '@' and is followed two lines further down by a line that begins with '+' (4-line FASTQ: a quality line may begin
// with '@' too, but then the line two below it is a sequence line). Same record conventions as FastxReader::next_packed.
class PlainChunks {
public:
    PlainChunks() : fd_(-1), size_(0), chunk_(0), fastq_(false) {}
    ~PlainChunks() { close(); }
    // a file whose bytes can be reached at any offset: plain (not gzip, starts like a FASTA / FASTQ file) or blocked gzip (BGZF, bgzf.hpp);
    // an ordinary gzip stream can only be inflated from its start and goes through the one-thread reader
    static bool is_plain(const std::string& fn) {
        FILE* f = fopen(fn.c_str(), "rb"); if (!f) return false;
        unsigned char m[18]; memset(m, 0, sizeof(m)); const size_t n = fread(m, 1, sizeof(m), f); fclose(f);
        if (n >= 1 && m[0] == '>') return true;
        if (n >= 1 && m[0] == '@') { std::vector<char> h(4u << 20); FILE* g = fopen(fn.c_str(), "rb"); if (!g) return false; h.resize(fread(h.data(), 1, h.size(), g)); fclose(g); return four_line_fastq(h.data(), h.size()); }
        if (!BgzfFile::looks_like(m, n)) return false;
        BgzfFile z; if (!z.open(fn)) return false; // every block checked (a file that merely starts with a BGZF block is streamed)
        std::vector<char> h(static_cast<size_t>(z.size() < (4u << 20) ? z.size() : (4u << 20)));
        if (h.empty() || !z.read(0, h.size(), h.data())) return false;
        return h[0] == '>' || (h[0] == '@' && four_line_fastq(h.data(), h.size()));
    }
    // is the first record of a FASTQ file laid out on four lines (header, sequence, '+', quality)? Judged from the head of the file; a first
    // record longer than that head is taken to be (one line of megabases is the 4-line form)
    static bool four_line_fastq(const char* b, size_t n) {
        const char* l1 = static_cast<const char*>(memchr(b, '\n', n)); if (!l1) return true;
        const char* l2 = static_cast<const char*>(memchr(l1 + 1, '\n', n - static_cast<size_t>(l1 + 1 - b))); if (!l2 || static_cast<size_t>(l2 + 1 - b) >= n) return true;
        return l2[1] == '+';
    }
    bool open(const std::string& fn, size_t chunk_bytes) {
        close();
        fd_ = ::open(fn.c_str(), O_RDONLY); if (fd_ < 0) return false;
        struct stat st; if (fstat(fd_, &st) != 0) { close(); return false; }
        size_ = static_cast<size_t>(st.st_size); chunk_ = chunk_bytes < 256 ? 256 : chunk_bytes;
        { unsigned char m[18]; memset(m, 0, sizeof(m)); const ssize_t n = pread(fd_, m, sizeof(m), 0);
          if (n > 0 && BgzfFile::looks_like(m, static_cast<size_t>(n))) { bgzf_.reset(new BgzfFile()); if (!bgzf_->open(fn)) { close(); return false; } size_ = static_cast<size_t>(bgzf_->size()); } }
        if (size_ > chunk_) { const size_t nc = (size_ + chunk_ - 1) / chunk_; chunk_ = (size_ + nc - 1) / nc; } // ranges of equal size (a short last range would be a short ticket: the kernels of a ticket have tails that do not shrink with it)
        char c = 0; fastq_ = size_ > 0 && (bgzf_ ? bgzf_->read(0, 1, &c) : pread(fd_, &c, 1, 0) == 1) && c == '@';
        return true;
    }
    void close() { if (fd_ >= 0) { ::close(fd_); fd_ = -1; } bgzf_.reset(); }
    bool is_bgzf() const { return bgzf_ != nullptr; }
    size_t n_chunks() const { return size_ == 0 ? 0 : (size_ + chunk_ - 1) / chunk_; }
    size_t file_bytes() const { return size_; }
    // the records of range i appended to `out`; false on a read error. Thread-safe (pread).
    bool parse_chunk(size_t i, PackedReads& out) const {
        const size_t lo = i * chunk_, hi = (i + 1) * chunk_ < size_ ? (i + 1) * chunk_ : size_;
        std::vector<char> buf = packed_buffer_take(); // (recycled: becomes the ticket's character buffer, the records point into it)
        size_t base = lo ? lo - 1 : 0; // one byte back: is `lo` the start of a line?
        if (!fill(buf, base, hi + (1u << 16))) return false;
        auto locate = [&](size_t off, size_t* res) -> bool { // record_start with the buffer grown until the answer is known (a long record may reach far beyond `hi`)
            for (;;) {
                const size_t e = record_start(buf, base, off);
                if (e != static_cast<size_t>(-1)) { *res = e; return true; }
                if (base + buf.size() >= size_) { *res = size_; return true; }
                if (!fill(buf, base, base + buf.size() + (4u << 20))) return false;
            }
        };
        size_t start = 0, end = hi;
        if (!locate(lo, &start)) return false;
        if (start >= hi) { packed_buffer_give(buf); return true; } // no record starts in this range
        if (hi < size_ && !locate(hi, &end)) return false;
        if (out.rec_.empty() && out.buf_.empty()) { out.buf_.swap(buf); parse_in_place(out, start - base, end - base); }
        else { parse_in_place_from(buf, start - base, end - base, out); packed_buffer_give(buf); }
        return !out.malformed_;
    }
private:
    bool fill(std::vector<char>& buf, size_t base, size_t upto) const { // buf = file[base, min(upto, size))
        if (upto > size_) upto = size_;
        size_t have = buf.size();
        if (base + have >= upto) return true;
        buf.resize(upto - base);
        if (bgzf_) return bgzf_->read(base + have, upto - base - have, buf.data() + have);
        while (base + have < upto) { const ssize_t n = pread(fd_, buf.data() + have, upto - base - have, static_cast<off_t>(base + have)); if (n <= 0) return false; have += static_cast<size_t>(n); }
        return true;
    }
    // first record start at or after file offset `off`, judged from what `buf` holds; size_ when the file ends first; -1 when `buf` ends first
    size_t record_start(const std::vector<char>& buf, size_t base, size_t off) const {
        const char* b = buf.data(); const size_t n = buf.size();
        size_t p = off - base;
        if (off != 0 && p >= 1) { // move to the start of a line
            if (b[p - 1] != '\n') { const char* q = static_cast<const char*>(memchr(b + p, '\n', n - p)); if (!q) return base + n >= size_ ? size_ : static_cast<size_t>(-1); p = static_cast<size_t>(q - b) + 1; }
        }
        while (true) {
            if (p >= n) return base + n >= size_ ? size_ : static_cast<size_t>(-1);
            if (!fastq_) { if (b[p] == '>') return base + p; }
            else if (b[p] == '@') { // a header when the line two below starts with '+'; a '@' line with fewer than two lines below it before the
                // end of the FILE is the quality line of the last record (taking it for a header cut that record's quality off and lost the read)
                const char* l1 = static_cast<const char*>(memchr(b + p, '\n', n - p));
                const char* l2 = l1 ? static_cast<const char*>(memchr(l1 + 1, '\n', n - static_cast<size_t>(l1 + 1 - b))) : nullptr;
                if (l2 && static_cast<size_t>(l2 + 1 - b) < n) { if (l2[1] == '+') return base + p; }
                else if (base + n < size_) return static_cast<size_t>(-1); // the buffer ends before the answer is known
            }
            const char* q = static_cast<const char*>(memchr(b + p, '\n', n - p));
            if (!q) return base + n >= size_ ? size_ : static_cast<size_t>(-1);
            p = static_cast<size_t>(q - b) + 1;
        }
    }
    // the records of buf[from, to): the ticket keeps `buf` itself and its records point into it (nothing is copied; the lines of a multi-line
    // FASTA record are moved together in place)
    void parse_in_place(PackedReads& out, size_t from, size_t to) const {
        char* s = out.buf_.data(); const size_t n = to;
        size_t p = from;
        auto line = [&](size_t& ls, size_t& le) -> bool { // next line [ls, le) without its end-of-line characters
            if (p >= n) return false;
            ls = p; const char* q = static_cast<const char*>(memchr(s + p, '\n', n - p));
            le = q ? static_cast<size_t>(q - s) : n; p = q ? le + 1 : n;
            if (le > ls && s[le - 1] == '\r') --le;
            return true;
        };
        size_t ls, le;
        bool have = line(ls, le);
        while (have) {
            if (le == ls || (s[ls] != '>' && s[ls] != '@')) { have = line(ls, le); continue; } // not a header: skipped like FastxReader does
            const bool fq = s[ls] == '@';
            size_t e = ls + 1; while (e < le && !isspace(static_cast<unsigned char>(s[e]))) ++e;
            PackedReads::Rec r; r.name_off = ls + 1; r.name_len = static_cast<uint32_t>(e - (ls + 1)); r.has_qual = false; r.qual_off = 0;
            if (fq) {
                size_t s0, s1, q0, q1;
                if (!line(s0, s1)) return;
                r.seq_off = s0; r.seq_len = static_cast<uint32_t>(s1 - s0);
                if (!line(q0, q1)) return; // '+' line
                if (q1 == q0 || s[q0] != '+') { out.malformed_ = true; return; } // not the 4-line form: refused, not misread
                if (!line(q0, q1)) return; // quality line
                if (out.keep_qual_ && q1 - q0 == r.seq_len) { r.qual_off = q0; r.has_qual = true; }
                have = line(ls, le);
            } else {
                r.seq_off = p; size_t w = p; // sequence lines are moved together, starting where the first one starts
                while ((have = line(ls, le)) && !(le > ls && (s[ls] == '>' || s[ls] == '@'))) { if (w != ls) memmove(s + w, s + ls, le - ls); w += le - ls; }
                r.seq_len = static_cast<uint32_t>(w - r.seq_off);
            }
            out.rec_.push_back(r); out.n_bases_ += r.seq_len;
        }
    }
    // (a ticket that already holds records: the new ones are appended to its buffer)
    void parse_in_place_from(std::vector<char>& buf, size_t from, size_t to, PackedReads& out) const {
        PackedReads tmp(out.keep_qual_); tmp.buf_.swap(buf); parse_in_place(tmp, from, to); if (tmp.malformed_) out.malformed_ = true;
        const size_t shift = out.buf_.size();
        out.buf_.insert(out.buf_.end(), tmp.buf_.begin(), tmp.buf_.begin() + to);
        for (size_t i = 0; i < tmp.rec_.size(); ++i) { PackedReads::Rec r = tmp.rec_[i]; r.name_off += shift; r.seq_off += shift; r.qual_off += shift; out.rec_.push_back(r); out.n_bases_ += r.seq_len; }
        buf.swap(tmp.buf_);
    }
    int fd_; size_t size_, chunk_; bool fastq_;
    std::unique_ptr<BgzfFile> bgzf_; // blocked gzip input: offsets are those of the uncompressed stream
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
