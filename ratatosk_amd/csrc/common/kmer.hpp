// 2-bit k-mer utilities shared by the product host code (index reader, flat-graph builder, tools).
//
// Encoding follows the on-disk `Kmer` convention the reference inherits from Bifrost
// (reference: src/Graph.cpp:786-801 writes `um.getUnitigHead().write(out)`):
// A=0, C=1, G=2, T=3, first base in the most significant bits. For k <= 31 a k-mer is one u64
// whose low 2k bits hold the bases (base i at bits [2(k-1-i), 2(k-1-i)+1]).
//
// NOT part of oracle/: the oracle has its own, independent restatement.
#ifndef RTK_COMMON_KMER_HPP
#define RTK_COMMON_KMER_HPP

#include <cstdint>
#include <string>

namespace rtk {

static const int RTK_MAX_K = 63; // MAX_KMER_SIZE = 64 build of the reference (CMakeLists.txt:6): k1 = 31 one-word codes, k2 = 63 two-word codes

inline int base2bits(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

inline char bits2base(int b) { return "ACGT"[b & 3]; }

inline char complement_base(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        // IUPAC codes (reference: Bifrost reverse_complement handles them; used by ResultCorrection.hpp:82)
        case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
        case 'W': return 'W'; case 'S': return 'S'; case 'V': return 'B'; case 'B': return 'V';
        case 'H': return 'D'; case 'D': return 'H'; case 'N': return 'N';
        default: return c;
    }
}

inline std::string reverse_complement(const std::string& s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) r[s.size() - 1 - i] = complement_base(s[i]);
    return r;
}

inline uint64_t kmer_mask(int k) { return (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1ULL); }

// Reverse complement of a k-mer held in the low 2k bits.
inline uint64_t kmer_revcomp(uint64_t x, int k) {
    x = ~x; // complement every 2-bit base (A<->T, C<->G)
    // reverse the order of the 32 2-bit groups
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

inline uint64_t kmer_canonical(uint64_t fw, int k, bool* is_fw = nullptr) {
    const uint64_t rc = kmer_revcomp(fw, k);
    if (is_fw) *is_fw = (fw <= rc);
    return fw <= rc ? fw : rc;
}

// ---- k in 33..63: the same operations on 128-bit codes (host tools only; the device has its own two-word form, hip/rtk_types.h) ----
typedef unsigned __int128 u128;
template <class KM> inline KM km_mask(int k) { return (2 * k >= static_cast<int>(8 * sizeof(KM))) ? ~static_cast<KM>(0) : ((static_cast<KM>(1) << (2 * k)) - static_cast<KM>(1)); }
inline u128 kmer_revcomp(u128 x, int k) {
    const u128 full = (static_cast<u128>(kmer_revcomp(static_cast<uint64_t>(x), 32)) << 64) | static_cast<u128>(kmer_revcomp(static_cast<uint64_t>(x >> 64), 32));
    return full >> (128 - 2 * k);
}
inline u128 kmer_canonical(u128 fw, int k, bool* is_fw = nullptr) {
    const u128 rc = kmer_revcomp(fw, k);
    if (is_fw) *is_fw = (fw <= rc);
    return fw <= rc ? fw : rc;
}
inline uint64_t hash64(uint64_t x);
inline uint64_t hash_km(uint64_t x) { return hash64(x); }
inline uint64_t hash_km(u128 x) { return hash64(static_cast<uint64_t>(x) ^ hash64(static_cast<uint64_t>(x >> 64) ^ 0x9e3779b97f4a7c15ULL)); }
template <class KM> inline bool km_encode(const char* s, int k, KM& out) {
    KM x = 0;
    for (int i = 0; i < k; ++i) { const int b = base2bits(s[i]); if (b < 0) return false; x = (x << 2) | static_cast<KM>(b); }
    out = x;
    return true;
}
template <class KM> inline std::string km_decode(KM x, int k) {
    std::string s(k, 'A');
    for (int i = 0; i < k; ++i) s[i] = bits2base(static_cast<int>(static_cast<uint64_t>(x >> (2 * (k - 1 - i))) & 3));
    return s;
}

// Encode s[0..k) ; returns false if a non-ACGT character is met.
inline bool kmer_encode(const char* s, int k, uint64_t& out) {
    uint64_t x = 0;
    for (int i = 0; i < k; ++i) {
        const int b = base2bits(s[i]);
        if (b < 0) return false;
        x = (x << 2) | static_cast<uint64_t>(b);
    }
    out = x;
    return true;
}

inline std::string kmer_decode(uint64_t x, int k) {
    std::string s(k, 'A');
    for (int i = 0; i < k; ++i) s[i] = bits2base(static_cast<int>((x >> (2 * (k - 1 - i))) & 3));
    return s;
}

// 64-bit finaliser (splitmix64 / murmur3 style); the k-mer table hash of the flat graph.
inline uint64_t hash64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// Deterministic PRNG for the synthetic generators (splitmix64 seeding xoshiro256**).
struct Rng {
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        uint64_t z = seed;
        for (int i = 0; i < 4; ++i) {
            z += 0x9e3779b97f4a7c15ULL;
            uint64_t t = z;
            t = (t ^ (t >> 30)) * 0xbf58476d1ce4e5b9ULL;
            t = (t ^ (t >> 27)) * 0x94d049bb133111ebULL;
            s[i] = t ^ (t >> 31);
        }
    }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    inline uint64_t next() {
        const uint64_t result = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return result;
    }
    inline double uniform() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
    inline uint64_t below(uint64_t n) { return n ? next() % n : 0; }
};

} // namespace rtk

#endif
