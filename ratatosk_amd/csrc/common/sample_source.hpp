// Synthetic short reads sampled on the fly from a reference FASTA -- an input "file" of the index tool that never exists on disk.
//
//   sample:REF.fa?cov=30&len=150&insert=400&err=0.005&seed=7
//
// stands for an interleaved paired-end FASTQ of cov x (length of one haplotype) bases: pair p is taken from haplotype p mod H (H = records of
// REF.fa: the two haplotypes of a diploid set alternate pair by pair), from a fragment of `insert` bases at a position, on a strand and with
// substitution errors that are all functions of (seed, p) alone. Mates carry the same name, "s<p>". Any read can therefore be produced by any
// thread at any time, and a pass over the input is a loop over pair numbers: the whole-genome-scale set of BASELINE.json configs[4] (3 Gb
// reference, 30x short reads = 180 GB of FASTQ) is counted and coloured without the file (SURVEY.md 8(d) row "config 5"; the box's disk holds
// 79 GB). There is no counterpart in the reference: it is the data generator's job done inside the consumer, like rtk_simulate outside it.
//
// The plain code path of the tool reads such a source through FastxReader like any file (one thread, record after record: the definition);
// the thread-parallel paths (`--fast` colouring, `--gpu` counting) take pair ranges. `rtk_build_index --dump-input` writes a source out as the FASTQ
// file it stands for (tests: the index built from the source and from that file are the same bytes).
#ifndef RTK_SAMPLE_SOURCE_HPP
#define RTK_SAMPLE_SOURCE_HPP

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <zlib.h>

namespace rtk {

class SampleSource {
public:
    static bool is_spec(const std::string& fn) { return fn.compare(0, 7, "sample:") == 0; }

    // the source a specification stands for (loaded once per process: several passes and threads share the reference). nullptr + *err on failure.
    static std::shared_ptr<SampleSource> get(const std::string& spec, std::string* err) {
        static std::mutex m; static std::map<std::string, std::shared_ptr<SampleSource> > cache;
        std::lock_guard<std::mutex> lk(m);
        std::map<std::string, std::shared_ptr<SampleSource> >::iterator it = cache.find(spec);
        if (it != cache.end()) return it->second;
        std::shared_ptr<SampleSource> s(new SampleSource());
        if (!s->init(spec, err)) return std::shared_ptr<SampleSource>();
        cache[spec] = s;
        return s;
    }

    uint64_t n_pairs() const { return n_pairs_; }
    uint64_t n_reads() const { return 2 * n_pairs_; }
    uint32_t read_len() const { return len_; }
    uint64_t n_bases() const { return n_reads() * len_; }

    // the two mates of pair p (upper-case A/C/G/T, `len` characters each) into m1 / m2
    void pair(uint64_t p, char* m1, char* m2) const {
        uint64_t z = seed_ * 0x9E3779B97F4A7C15ULL + p * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL;
        const uint64_t r0 = mix(z), r1 = mix(z + 1);
        const std::string& hap = haps_[p % haps_.size()];
        const uint64_t span = hap.size() - insert_ + 1;
        const uint64_t start = static_cast<uint64_t>((static_cast<unsigned __int128>(r0) * span) >> 64);
        const bool rev = (r1 & 1ULL) != 0;
        const char* f = hap.data() + start;
        // mate 1 = the first len bases of the fragment as sequenced, mate 2 = the first len bases of its reverse complement
        if (!rev) { memcpy(m1, f, len_); for (uint32_t i = 0; i < len_; ++i) m2[i] = comp(f[insert_ - 1 - i]); }
        else { for (uint32_t i = 0; i < len_; ++i) m1[i] = comp(f[insert_ - 1 - i]); memcpy(m2, f, len_); }
        if (err_ > 0.0) { // substitutions at geometric distances over the 2 x len bases of the pair
            uint64_t s = r1 >> 1 | 1ULL;
            for (uint64_t at = gap(s);; at += 1 + gap(s)) {
                if (at >= 2ull * len_) break;
                char* c = at < len_ ? m1 + at : m2 + (at - len_);
                s = step(s);
                const int shift = 1 + static_cast<int>((s >> 33) % 3ULL);
                *c = "ACGT"[(code(*c) + shift) & 3];
            }
        }
    }

private:
    SampleSource() : n_pairs_(0), len_(150), insert_(400), err_(0.0), seed_(1), inv_log_(0.0) {}
    static uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; return x ^ (x >> 31); }
    static uint64_t step(uint64_t s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    static char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }
    static int code(char c) { return c == 'A' ? 0 : (c == 'C' ? 1 : (c == 'G' ? 2 : 3)); }
    uint64_t gap(uint64_t& s) const { // bases until the next substitution: floor(log(u) / log(1 - err))
        s = step(s);
        const double u = (static_cast<double>(s >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        const double g = __builtin_log(u) * inv_log_;
        return g > 1e15 ? static_cast<uint64_t>(1e15) : static_cast<uint64_t>(g);
    }

    bool init(const std::string& spec, std::string* err) {
        std::string rest = spec.substr(7), path = rest; double cov = 30.0;
        const size_t q = rest.find('?');
        if (q != std::string::npos) {
            path = rest.substr(0, q);
            std::string opts = rest.substr(q + 1);
            for (size_t a = 0; a < opts.size();) {
                size_t e = opts.find('&', a); if (e == std::string::npos) e = opts.size();
                const std::string kv = opts.substr(a, e - a); const size_t eq = kv.find('=');
                if (eq == std::string::npos) { *err = "sample source: option without a value: " + kv; return false; }
                const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
                if (k == "cov") cov = atof(v.c_str()); else if (k == "len") len_ = static_cast<uint32_t>(atoi(v.c_str())); else if (k == "insert") insert_ = static_cast<uint32_t>(atoi(v.c_str()));
                else if (k == "err") err_ = atof(v.c_str()); else if (k == "seed") seed_ = strtoull(v.c_str(), nullptr, 10);
                else { *err = "sample source: unknown option " + k; return false; }
                a = e + 1;
            }
        }
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) { *err = "sample source: cannot open " + path; return false; }
        gzbuffer(f, 1 << 22);
        std::vector<char> buf(1 << 22);
        while (gzgets(f, buf.data(), static_cast<int>(buf.size()))) {
            size_t n = strlen(buf.data());
            while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
            if (n && buf[0] == '>') { haps_.push_back(std::string()); continue; }
            if (haps_.empty()) haps_.push_back(std::string());
            std::string& h = haps_.back(); const size_t at = h.size(); h.resize(at + n);
            for (size_t i = 0; i < n; ++i) { const char c = static_cast<char>(buf[i] & 0xDF); h[at + i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'A'; }
        }
        gzclose(f);
        if (haps_.empty() || len_ < 1 || insert_ < len_) { *err = "sample source: empty reference or insert < len"; return false; }
        for (size_t i = 0; i < haps_.size(); ++i) if (haps_[i].size() < insert_) { *err = "sample source: a reference record is shorter than the insert"; return false; }
        if (err_ < 0.0 || err_ >= 0.5) { *err = "sample source: err out of range"; return false; }
        if (err_ > 0.0) inv_log_ = 1.0 / __builtin_log(1.0 - err_);
        n_pairs_ = static_cast<uint64_t>(cov * static_cast<double>(haps_[0].size()) / (2.0 * len_));
        if (n_pairs_ < 1) n_pairs_ = 1;
        return true;
    }

    std::vector<std::string> haps_;
    uint64_t n_pairs_; uint32_t len_, insert_; double err_; uint64_t seed_; double inv_log_;
};

} // namespace rtk

#endif
