// Blocked gzip (BGZF, the container of bgzip / htslib: SAM specification section 4.1). A BGZF file is a series of gzip members of at most
// 64 KiB, each with an extra field "BC" that holds the member's compressed size: any gzip reader inflates it as one stream, and a reader that
// knows the field can hop from member to member without inflating and inflate them side by side.
//   * writer: the CLI's -G output (one or more BGZF blocks per ticket block; the reference writes one plain gzip stream, src/Ratatosk.cpp:510:
//     the decompressed bytes are the same);
//   * reader: first-pass input as byte ranges of the UNCOMPRESSED stream, inflated and parsed by the -c threads (rtk::PlainChunks).
#ifndef RTK_BGZF_HPP
#define RTK_BGZF_HPP

#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "finflate.hpp"

namespace rtk {

static const size_t BGZF_BLOCK_BYTES = 0xff00; // uncompressed bytes per block (the size htslib uses: a stored block still fits 64 KiB)

// `in` as BGZF blocks appended to `out`
inline bool bgzf_compress(const char* in, size_t n, std::string& out, int level = Z_DEFAULT_COMPRESSION) {
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    z_stream z0; memset(&z0, 0, sizeof(z0)); bool have_z0 = false; // (stored blocks for data that does not compress)
    const size_t bound = deflateBound(&zs, BGZF_BLOCK_BYTES) + 64;
    bool ok = true;
    for (size_t ip = 0; ok && ip < n; ip += BGZF_BLOCK_BYTES) {
        const size_t len = std::min(BGZF_BLOCK_BYTES, n - ip);
        const size_t at = out.size();
        out.resize(at + 18 + bound + 8);
        unsigned char* o = reinterpret_cast<unsigned char*>(&out[at]);
        size_t clen = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            z_stream* z = &zs;
            if (attempt == 1) { if (!have_z0) { if (deflateInit2(&z0, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok = false; break; } have_z0 = true; } z = &z0; }
            deflateReset(z);
            z->next_in = reinterpret_cast<Bytef*>(const_cast<char*>(in + ip)); z->avail_in = static_cast<uInt>(len);
            z->next_out = o + 18; z->avail_out = static_cast<uInt>(bound);
            if (deflate(z, Z_FINISH) != Z_STREAM_END) { ok = false; break; }
            clen = bound - z->avail_out;
            if (18 + clen + 8 <= 65536) break;
            if (attempt == 1) ok = false;
        }
        if (!ok) break;
        const uint32_t bsize = static_cast<uint32_t>(18 + clen + 8 - 1);
        static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(o, hdr, 16); o[16] = static_cast<unsigned char>(bsize & 0xff); o[17] = static_cast<unsigned char>(bsize >> 8);
        const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(in + ip), static_cast<uInt>(len)));
        unsigned char* t = o + 18 + clen;
        for (int i = 0; i < 4; ++i) { t[i] = static_cast<unsigned char>(crc >> (8 * i)); t[4 + i] = static_cast<unsigned char>(static_cast<uint32_t>(len) >> (8 * i)); }
        out.resize(at + 18 + clen + 8);
    }
    deflateEnd(&zs); if (have_z0) deflateEnd(&z0);
    return ok;
}
inline void bgzf_append_eof(std::string& out) { // the empty block that ends a BGZF file
    static const unsigned char eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    out.append(reinterpret_cast<const char*>(eof), 28);
}

// Block index of a BGZF file + random access into its uncompressed stream. Thread-safe after open().
class BgzfFile {
public:
    BgzfFile() : map_(nullptr), map_bytes_(0), usize_(0) {}
    ~BgzfFile() { close(); }
    static bool looks_like(const unsigned char* h, size_t n) { return n >= 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C'; }
    // false: not a BGZF file from the first to the last byte (the caller falls back to the streaming reader)
    bool open(const std::string& fn) {
        close();
        const int fd = ::open(fn.c_str(), O_RDONLY); if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0 || st.st_size < 28) { ::close(fd); return false; }
        map_bytes_ = static_cast<size_t>(st.st_size);
        void* m = mmap(nullptr, map_bytes_, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) { map_ = nullptr; return false; }
        map_ = static_cast<const unsigned char*>(m);
        size_t off = 0; uint64_t u = 0;
        while (off < map_bytes_) {
            if (off + 18 > map_bytes_) { close(); return false; }
            const unsigned char* h = map_ + off;
            if (!(h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4)) || (h[3] & ~4)) { close(); return false; } // (other header fields would move the data)
            const size_t xlen = h[10] | (static_cast<size_t>(h[11]) << 8);
            if (off + 12 + xlen > map_bytes_) { close(); return false; }
            size_t bsize = 0;
            for (size_t x = 0; x + 4 <= xlen;) { // subfields: SI1 SI2 LEN(2) data
                const unsigned char* f = h + 12 + x; const size_t fl = f[2] | (static_cast<size_t>(f[3]) << 8);
                if (f[0] == 'B' && f[1] == 'C' && fl == 2 && x + 6 <= xlen) bsize = (f[4] | (static_cast<size_t>(f[5]) << 8)) + 1;
                x += 4 + fl;
            }
            if (bsize < 12 + xlen + 8 + 2 || off + bsize > map_bytes_) { close(); return false; }
            const unsigned char* t = h + bsize - 4;
            const uint32_t isize = t[0] | (static_cast<uint32_t>(t[1]) << 8) | (static_cast<uint32_t>(t[2]) << 16) | (static_cast<uint32_t>(t[3]) << 24);
            if (isize > 65536) { close(); return false; }
            if (isize) { coff_.push_back(off); hdr_.push_back(static_cast<uint32_t>(12 + xlen)); blen_.push_back(static_cast<uint32_t>(bsize)); uoff_.push_back(u); u += isize; }
            off += bsize;
        }
        uoff_.push_back(u); usize_ = u;
        return true;
    }
    void close() { if (map_) munmap(const_cast<unsigned char*>(map_), map_bytes_); map_ = nullptr; coff_.clear(); hdr_.clear(); blen_.clear(); uoff_.clear(); usize_ = 0; }
    uint64_t size() const { return usize_; }          // uncompressed bytes
    size_t compressed_bytes() const { return map_bytes_; }
    // dst = uncompressed[off, off + len); false on corrupt data
    bool read(uint64_t off, size_t len, char* dst) const {
        if (len == 0) return true;
        if (off + len > usize_) return false;
        size_t b = static_cast<size_t>(std::upper_bound(uoff_.begin(), uoff_.end(), off) - uoff_.begin()) - 1;
        static const bool use_zlib = []() { const char* e = getenv("RTK_ZLIB_INFLATE"); return e && e[0] == '1'; }();
        if (!use_zlib) { // the reader's own inflate (finflate.hpp): the whole member, header and trailer included, so its CRC-32 and length are checked too
            FastBlock& fb = fast_block();
            const size_t cap = sizeof(fb.buf) - FastInflate::SLACK;
            while (len) {
                const uint64_t u0 = uoff_[b]; const size_t ulen = static_cast<size_t>(uoff_[b + 1] - u0);
                const size_t skip = static_cast<size_t>(off - u0), take = std::min(len, ulen - skip);
                if (!fb.fi.begin(map_ + coff_[b], map_ + coff_[b] + blen_[b])) return false;
                size_t got = 0;
                while (!fb.fi.done()) { const size_t n = fb.fi.decode(fb.buf + got, cap - got, got); if (fb.fi.failed() || (n == 0 && !fb.fi.done())) return false; got += n; if (got > 65536) return false; }
                if (got != ulen) return false;
                memcpy(dst, fb.buf + skip, take);
                dst += take; off += take; len -= take; ++b;
            }
            return true;
        }
        Inflater& z = inflater();
        if (!z.ok) return false;
        unsigned char tmp[65536];
        while (len) {
            const uint64_t u0 = uoff_[b]; const size_t ulen = static_cast<size_t>(uoff_[b + 1] - u0);
            const size_t hdr = hdr_[b], bsize = blen_[b];
            const size_t skip = static_cast<size_t>(off - u0), take = std::min(len, ulen - skip);
            const bool direct = skip == 0 && take == ulen; // the whole block lands in dst
            inflateReset(&z.zs);
            z.zs.next_in = const_cast<Bytef*>(map_ + coff_[b] + hdr); z.zs.avail_in = static_cast<uInt>(bsize - hdr - 8);
            z.zs.next_out = direct ? reinterpret_cast<Bytef*>(dst) : tmp; z.zs.avail_out = static_cast<uInt>(direct ? ulen : sizeof(tmp));
            if (inflate(&z.zs, Z_FINISH) != Z_STREAM_END || z.zs.total_out != ulen) return false;
            { const unsigned char* tr = map_ + coff_[b] + bsize - 8; // the member's CRC-32 (raw inflate does not look at it)
              const uint32_t want = tr[0] | (tr[1] << 8) | (tr[2] << 16) | (static_cast<uint32_t>(tr[3]) << 24);
              if (fast_crc32(0, direct ? reinterpret_cast<const unsigned char*>(dst) : tmp, ulen) != want) return false; }
            if (!direct) memcpy(dst, tmp + skip, take);
            dst += take; off += take; len -= take; ++b;
        }
        return true;
    }
private:
    struct Inflater { z_stream zs; bool ok; Inflater() { memset(&zs, 0, sizeof(zs)); ok = inflateInit2(&zs, -15) == Z_OK; } ~Inflater() { if (ok) inflateEnd(&zs); } };
    static Inflater& inflater() { static thread_local Inflater z; return z; }
    struct FastBlock { FastInflate fi; unsigned char buf[65536 + 192 + FastInflate::SLACK]; };
    static FastBlock& fast_block() { static thread_local std::unique_ptr<FastBlock> p(new FastBlock()); return *p; }
    const unsigned char* map_; size_t map_bytes_;
    std::vector<uint64_t> coff_, uoff_; std::vector<uint32_t> hdr_, blen_; // per data block: file offset, header bytes, block bytes; uncompressed offset (+ one past the end)
    uint64_t usize_;
};

} // namespace rtk

#endif
