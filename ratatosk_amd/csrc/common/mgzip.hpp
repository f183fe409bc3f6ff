// Ordinary gzip input inflated on several threads when the file is a CONCATENATION of gzip members.
//
// A deflate stream can only be inflated from its start, so one `gzip big.fastq` is sequential work (zlib: ~0.35 GB/s of text, an order of
// magnitude below what one GPU corrects). But long-read sets are usually many files written by the sequencer software and joined with
// `cat *.fastq.gz > all.fastq.gz`: one gzip MEMBER per original file, each decodable on its own once its start is known. The reader below
// finds member starts without an index:
//   * candidates = every offset that holds the gzip magic 1f 8b 08 with the three reserved flag bits clear (one false hit per ~130 MB of
//     compressed data; a false hit fails at its header or first block and costs nothing worth counting);
//   * worker threads inflate the candidates in offset order, a bounded number of them ahead of the consumer, each into its own chunk list;
//     zlib checks the CRC-32 and the length at the end of every member;
//   * the consumer walks the CHAIN: the member at offset 0, then the one that starts exactly where it ended, and so on; candidates the
//     chain steps over were false and are dropped. What the chain delivers is byte for byte what gzread() delivers (trailing bytes that
//     are not a member are ignored like zlib ignores them; a damaged member is an error here and there).
//   * the member at the head of the chain is consumed WHILE it is inflated, so a file that is one big member streams like before (on a
//     thread of its own) instead of being buffered; members ahead of the head stop at 128 MB of buffered text until they become the head.
// Blocked gzip (BGZF) is a multi-member file too and goes through here whenever the byte-range reader does not take it.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "finflate.hpp"

namespace rtk {

class MemberGzipReader {
public:
    MemberGzipReader() : fd_(-1), data_(nullptr), size_(0), scan_(0), pos_(0), max_ahead_(2), scan_done_(false), stop_(false), error_(false), started_(false), n_members_(0) {}
    ~MemberGzipReader() { close(); }
    MemberGzipReader(const MemberGzipReader&) = delete; MemberGzipReader& operator=(const MemberGzipReader&) = delete;

    static bool looks_like_gzip(const std::string& fn) {
        FILE* f = fopen(fn.c_str(), "rb"); if (!f) return false;
        unsigned char m[4] = {0, 0, 0, 0}; const size_t n = fread(m, 1, 4, f); fclose(f);
        return n == 4 && m[0] == 0x1f && m[1] == 0x8b && m[2] == 8 && (m[3] & 0xE0) == 0;
    }

    bool open(const std::string& fn, int n_threads) {
        close();
        fd_ = ::open(fn.c_str(), O_RDONLY); if (fd_ < 0) return false;
        struct stat st; if (fstat(fd_, &st) != 0 || st.st_size < 18) { ::close(fd_); fd_ = -1; return false; }
        size_ = static_cast<size_t>(st.st_size);
        void* p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (p == MAP_FAILED) { ::close(fd_); fd_ = -1; return false; }
        data_ = static_cast<const unsigned char*>(p);
        madvise(p, size_, MADV_SEQUENTIAL);
        if (!is_magic(0)) { close(); return false; }
        scan_ = 0; pos_ = 0; scan_done_ = false; stop_ = false; error_ = false; started_ = true; n_members_ = 0; cur_.reset(); cur_chunk_off_ = 0;
        if (n_threads < 1) n_threads = 1;
        max_ahead_ = static_cast<size_t>(n_threads) + 1;
        for (int t = 0; t < n_threads; ++t) workers_.emplace_back([this]() { work(); });
        return true;
    }

    void close() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_work_.notify_all(); cv_data_.notify_all();
        for (size_t i = 0; i < workers_.size(); ++i) workers_[i].join();
        workers_.clear(); tasks_.clear(); cur_.reset();
        if (data_) { munmap(const_cast<unsigned char*>(data_), size_); data_ = nullptr; }
        if (fd_ >= 0) { ::close(fd_); fd_ = -1; }
        started_ = false;
    }

    bool failed() const { return error_; }
    size_t members() const { return n_members_; }

    // like gzread(): bytes delivered, 0 at the end, -1 on a damaged member
    long read(char* dst, size_t want) {
        if (!started_) return -1;
        size_t got = 0;
        while (got < want) {
            if (!cur_) { // next member of the chain
                std::unique_lock<std::mutex> lk(m_);
                if (pos_.load() >= size_) break; // end of the file
                if (!is_magic(pos_.load())) {
                    // a member header of which fewer than 18 bytes are left is a cut-short member (gzread: unexpected end of file), not trailing junk
                    const size_t o = pos_.load();
                    if (size_ - o < 18 && size_ - o >= 2 && data_[o] == 0x1f && data_[o + 1] == 0x8b) { error_ = true; return got ? static_cast<long>(got) : -1; }
                    break; // bytes that are no member: ignored like zlib does
                }
                cv_data_.wait(lk, [&]() { return tasks_.count(pos_.load()) != 0 || scan_done_ || stop_; });
                std::map<size_t, std::shared_ptr<Task> >::iterator it = tasks_.find(pos_.load());
                if (it == tasks_.end()) { error_ = true; return -1; } // (cannot happen: every magic offset becomes a task)
                cur_ = it->second; cur_->head = true; cur_chunk_off_ = 0;
                cv_work_.notify_all(); // a paused member that became the head goes on
            }
            Chunk spent; bool ended = false, bad = false;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_data_.wait(lk, [&]() { return !cur_->chunks.empty() || cur_->done || stop_; });
                if (!cur_->chunks.empty()) {
                    Chunk& c = cur_->chunks.front();
                    const size_t n = std::min(want - got, c.n - cur_chunk_off_);
                    memcpy(dst + got, c.p.get() + c.off + cur_chunk_off_, n); got += n; cur_chunk_off_ += n;
                    if (cur_chunk_off_ == c.n) { cur_->buffered -= c.n; spent.p.swap(c.p); cur_->chunks.pop_front(); cur_chunk_off_ = 0; cv_work_.notify_all(); } // (freed outside the lock)
                } else if (cur_->done) { ended = true; bad = !cur_->ok; }
                else return -1; // stopped
            }
            if (ended) {
                if (bad) { error_ = true; return got ? static_cast<long>(got) : -1; } // a damaged member (the text before it was delivered, like gzread does)
                std::lock_guard<std::mutex> lk(m_);
                ++n_members_;
                pos_ = cur_->end;
                cur_.reset();
                while (!tasks_.empty() && tasks_.begin()->first < pos_.load()) { tasks_.begin()->second->cancel = true; tasks_.erase(tasks_.begin()); } // consumed, or candidates the chain stepped over
                cv_work_.notify_all();
            }
        }
        return static_cast<long>(got);
    }

private:
    struct Chunk { std::unique_ptr<char[]> p; size_t n, off; Chunk() : n(0), off(0) {} }; // text = p[off, off + n) (no value initialisation of megabytes per member; the fast decoder keeps the 32 KB before the text in front of it)
    struct Task {
        size_t off, end, buffered, produced; bool done, ok, head, cancel;
        std::deque<Chunk> chunks;
        Task() : off(0), end(0), buffered(0), produced(0), done(false), ok(false), head(false), cancel(false) {}
    };
    bool is_magic(size_t o) const { return o + 18 <= size_ && data_[o] == 0x1f && data_[o + 1] == 0x8b && data_[o + 2] == 8 && (data_[o + 3] & 0xE0) == 0; }

    void work() {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&]() { return stop_ || (!scan_done_ && ahead() < max_ahead_); });
                if (stop_) return;
            }
            // next candidate at or after scan_ (never before the chain position: what lies behind it is consumed); the scan touches file pages
            // that may have to come from the disk, so it has a lock of its own and the consumer is not held up by it
            std::shared_ptr<Task> t;
            {
                std::lock_guard<std::mutex> sl(scan_m_);
                const size_t p = pos_.load();
                size_t o = scan_ < p ? p : scan_;
                while (o < size_) {
                    const void* q = memchr(data_ + o, 0x1f, size_ - o);
                    if (!q) { o = size_; break; }
                    o = static_cast<size_t>(static_cast<const unsigned char*>(q) - data_);
                    if (is_magic(o)) break;
                    ++o;
                }
                scan_ = o < size_ ? o + 1 : size_;
                // the candidate enters the table before the scan lock is released: "the scan is over" must never become visible before a
                // candidate that an earlier scan step found (the consumer would take the chain for broken)
                std::lock_guard<std::mutex> lk(m_);
                if (o >= size_) { scan_done_ = true; cv_data_.notify_all(); cv_work_.notify_all(); continue; }
                if (o < pos_.load()) continue; // the chain moved past it while it was being found: a false start inside a consumed member
                t.reset(new Task()); t->off = o; tasks_[o] = t;
                cv_data_.notify_all();
            }
            inflate_member(*t);
        }
    }
    size_t ahead() const { size_t n = 0; for (std::map<size_t, std::shared_ptr<Task> >::const_iterator it = tasks_.begin(); it != tasks_.end(); ++it) n += it->second->done && !it->second->ok ? 0 : 1; return n; } // (failed candidates hold nothing)

    // hands a finished chunk to the consumer; false when the task was cancelled or the reader is closing
    bool deliver_(Task& t, Chunk& c) {
        std::unique_lock<std::mutex> lk(m_);
        t.chunks.emplace_back(); t.chunks.back().p.swap(c.p); t.chunks.back().n = c.n; t.chunks.back().off = c.off; t.buffered += c.n; t.produced += c.n;
        cv_data_.notify_all();
        cv_work_.wait(lk, [&]() { return stop_ || t.cancel || t.head || t.off < pos_.load() || t.buffered < (128u << 20); }); // ahead of the head: bounded
        if (stop_ || t.cancel || (!t.head && t.off < pos_.load())) return false; // (behind the chain position without being its head: a false start)
        if (t.head) cv_work_.wait(lk, [&]() { return stop_ || t.buffered < (64u << 20); }); // the head: a few chunks in front of the consumer
        return !stop_;
    }
    void finish_(Task& t, bool ok, size_t end) {
        std::lock_guard<std::mutex> lk(m_);
        t.ok = ok; t.end = end; t.done = true;
        if (!ok) { t.chunks.clear(); t.buffered = 0; }
        cv_data_.notify_all(); cv_work_.notify_all();
    }
    static bool use_zlib_() { static const bool z = []() { const char* e = getenv("RTK_ZLIB_INFLATE"); return e && e[0] == '1'; }(); return z; }

    void inflate_member(Task& t) {
        if (use_zlib_()) { inflate_member_zlib(t); return; }
        FastInflate fi; // (finflate.hpp: ~2x zlib on FASTQ text; CRC-32 and length of the member checked at its end)
        bool ok = false; size_t end = t.off;
        if (fi.begin(data_ + t.off, data_ + size_)) {
            const size_t H = FastInflate::HIST;
            std::unique_ptr<unsigned char[]> hist(new unsigned char[H]); size_t hn = 0; // the last bytes of the text so far, at the END of hist
            size_t n_chunks = 0;
            for (;;) {
                const size_t CH = n_chunks == 0 ? (128u << 10) : (n_chunks == 1 ? (1u << 20) : (4u << 20)); ++n_chunks;
                Chunk c; c.p.reset(new char[H + CH + FastInflate::SLACK]); c.off = H;
                unsigned char* base = reinterpret_cast<unsigned char*>(c.p.get());
                memcpy(base + H - hn, hist.get() + H - hn, hn);
                const size_t n = fi.decode(base + H, CH, hn);
                if (fi.failed()) break;
                const size_t nn = hn + n < H ? hn + n : H;
                memcpy(hist.get() + H - nn, base + H + n - nn, nn); hn = nn;
                const bool last = fi.done();
                if (n) { c.n = n; if (!deliver_(t, c)) break; }
                if (last) { ok = true; end = static_cast<size_t>(fi.member_end() - data_); break; }
                if (n == 0) break; // (cannot happen: no progress without an end or an error)
            }
        }
        finish_(t, ok, end);
    }

    void inflate_member_zlib(Task& t) {
        z_stream z; memset(&z, 0, sizeof(z));
        bool ok = false; size_t end = t.off;
        if (inflateInit2(&z, 15 + 16) == Z_OK) {
            size_t in_pos = t.off;
            size_t n_chunks = 0;
            for (;;) {
                if (z.avail_in == 0) { const size_t n = std::min<size_t>(size_ - in_pos, 1u << 30); z.next_in = const_cast<unsigned char*>(data_ + in_pos); z.avail_in = static_cast<uInt>(n); in_pos += n; }
                const size_t CH = n_chunks == 0 ? (128u << 10) : (n_chunks == 1 ? (1u << 20) : (4u << 20)); ++n_chunks; // (a BGZF member is 64 KB of text, a sequencer file tens of MB)
                Chunk c; c.p.reset(new char[CH]);
                z.next_out = reinterpret_cast<unsigned char*>(c.p.get()); z.avail_out = static_cast<uInt>(CH);
                const int r = inflate(&z, Z_NO_FLUSH);
                const size_t n = CH - z.avail_out;
                if (r != Z_OK && r != Z_STREAM_END) break; // damaged, cut short, or not a member at all
                if (n) {
                    c.n = n;
                    std::unique_lock<std::mutex> lk(m_);
                    t.chunks.emplace_back(); t.chunks.back().p.swap(c.p); t.chunks.back().n = n; t.buffered += n; t.produced += n;
                    cv_data_.notify_all();
                    cv_work_.wait(lk, [&]() { return stop_ || t.cancel || t.head || t.buffered < (128u << 20); }); // ahead of the head: bounded
                    if (stop_ || t.cancel) break;
                    if (t.head) cv_work_.wait(lk, [&]() { return stop_ || t.buffered < (64u << 20); }); // the head: a few chunks in front of the consumer
                    if (stop_) break;
                }
                if (r == Z_STREAM_END) { ok = true; end = in_pos - z.avail_in; break; }
            }
            inflateEnd(&z);
        }
        std::lock_guard<std::mutex> lk(m_);
        t.ok = ok; t.end = end; t.done = true;
        if (!ok) { t.chunks.clear(); t.buffered = 0; }
        cv_data_.notify_all(); cv_work_.notify_all();
    }

    int fd_; const unsigned char* data_; size_t size_;
    std::mutex m_; std::condition_variable cv_work_, cv_data_;
    std::map<size_t, std::shared_ptr<Task> > tasks_;
    std::vector<std::thread> workers_;
    std::mutex scan_m_; size_t scan_; std::atomic<size_t> pos_; size_t max_ahead_; bool scan_done_;
    bool stop_, error_, started_;
    size_t n_members_;
    std::shared_ptr<Task> cur_; size_t cur_chunk_off_;
};

} // namespace rtk
