// POD views shared by the host loader, the HIP kernels and the C-ABI layer.
//
// Flat, read-only image of the compacted coloured de Bruijn graph as it lives in HBM (SoA / CSR):
// everything the reference reaches through Bifrost's CompactedDBG<UnitigData> on the hot path
// (reference: src/UnitigData.hpp:258-491, src/SharedPairID.cpp, Bifrost find()/getSuccessors()).
#ifndef RTK_TYPES_H
#define RTK_TYPES_H

#include <stdint.h>

#if defined(__HIPCC__) && !defined(RTK_SIM)
#include <hip/hip_runtime.h>
#define RTK_HD __host__ __device__ __forceinline__
#else
#define RTK_HD inline
#endif

// ---- wave-uniform values -------------------------------------------------------------------------------------------------
// rtk_u(v): "v is the same in every lane". Identity on the value; on the device it moves the value to scalar registers, so that
// the wave-level programs keep their control state, pointers and loop counters in SGPRs (scalar ALU and branches, spills into
// VGPR lanes instead of 64-wide stores to the stack). Only ever applied to values that are wave-uniform by construction.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTK_SIM)
template <class T> __device__ __forceinline__ T rtk_u(T v) {
    static_assert(sizeof(T) <= 8, "rtk_u: scalar types only");
    if (sizeof(T) <= 4) {
        uint32_t x = 0; __builtin_memcpy(&x, &v, sizeof(T));
        x = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(x)));
        T r; __builtin_memcpy(&r, &x, sizeof(T)); return r;
    } else {
        uint64_t x = 0; __builtin_memcpy(&x, &v, sizeof(T));
        const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(x & 0xFFFFFFFFull)));
        const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(x >> 32)));
        x = (static_cast<uint64_t>(hi) << 32) | lo;
        T r; __builtin_memcpy(&r, &x, sizeof(T)); return r;
    }
}
#else
template <class T> RTK_HD T rtk_u(T v) { return v; }
#endif

// rtk_gp(p): "p points into device (global) memory" -- never into LDS or the wave's private stack. Pointers that the wave programs read from
// their descriptors are generic to the compiler (they come out of LDS or out of structs), and a generic access is a FLAT instruction: it takes the
// LDS address path as well as the memory path, counts in both wait counters (so that a wait for an LDS read also waits for it) and computes its
// 64-bit address per lane. With the address space known the same access is a GLOBAL instruction (scalar base + lane offset, memory counter only).
// Applied to every pointer FIELD of the views and work-area descriptors (U<T*>); the few fields that may point into LDS are UL<T*> below.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTK_SIM) && !defined(RTK_NO_GLOBAL_PTRS)
template <class T> __device__ __forceinline__ T rtk_gp(T v) { return v; }
// (through an integer: a generic -> global -> generic pointer cast is folded away, and an assumption about the address space is not picked up)
template <class T> __device__ __forceinline__ T* rtk_gp(T* p) { return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)(p); }
// A pointer-to-const FIELD of a view (the graph's arrays, the reads, the anchors of the seed stage: everything a kernel is handed as input) points to
// memory that no running kernel writes: the CONSTANT address space. An access through it with a wave-uniform address -- and most of the wave programs'
// reads of the graph are that: neighbour slots, flag words, offsets of ONE unitig -- becomes a SCALAR load (s_load: scalar cache, no vector-memory
// instruction, the value lands in SGPRs), with a per-lane address it stays a global load.
// RULE: a pointer-to-const field must never point to memory the same launch writes (the scalar cache is not coherent with the vector stores of the wave: a read may return
// what was there before). A bitmap, list or string a wave program builds and then reads is held in a pointer to NON-const (MyersScratch::need_bm is the case that went wrong).
#ifndef RTK_NO_CONST_PTRS
template <class T> __device__ __forceinline__ const T* rtk_gp(const T* p) { return (const T*)(__attribute__((address_space(4))) const T*)(unsigned long long)(p); }
#endif
#else
template <class T> RTK_HD T rtk_gp(T v) { return v; }
#endif

// U<T>: a struct field that holds a wave-uniform value (all the view / scratch descriptors below are per wave or per launch).
// Reads go through rtk_u, so every use site gets the scalar form without being written differently. Same layout as T.
template <class T> struct U {
    T v;
    RTK_HD operator T() const { return rtk_gp(rtk_u(v)); }
    RTK_HD T get() const { return rtk_gp(rtk_u(v)); }
    RTK_HD U& operator=(T x) { v = x; return *this; }
    template <class X> RTK_HD U& operator+=(X x) { v = static_cast<T>(rtk_u(v) + x); return *this; }
    template <class X> RTK_HD U& operator-=(X x) { v = static_cast<T>(rtk_u(v) - x); return *this; }
    template <class X> RTK_HD U& operator*=(X x) { v = static_cast<T>(rtk_u(v) * x); return *this; }
    template <class X> RTK_HD U& operator|=(X x) { v = static_cast<T>(rtk_u(v) | x); return *this; }
    RTK_HD U& operator++() { v = rtk_u(v) + 1; return *this; }
    RTK_HD U& operator--() { v = rtk_u(v) - 1; return *this; }
    RTK_HD T operator++(int) { const T o = rtk_u(v); v = o + 1; return o; }
    RTK_HD T operator--(int) { const T o = rtk_u(v); v = o - 1; return o; }
    RTK_HD T operator->() const { return rtk_gp(rtk_u(v)); }
    RTK_HD decltype(auto) operator*() const { return *rtk_gp(rtk_u(v)); }
    template <class I> RTK_HD decltype(auto) operator[](I i) const { return rtk_gp(rtk_u(v))[i]; }
};
// UL<T*>: a wave-uniform pointer field that may point into LDS (the work-area header of the region kernels and the overflow word inside it)
template <class T> struct UL {
    T v;
    RTK_HD operator T() const { return rtk_u(v); }
    RTK_HD T get() const { return rtk_u(v); }
    RTK_HD UL& operator=(T x) { v = x; return *this; }
    RTK_HD T operator->() const { return rtk_u(v); }
    RTK_HD decltype(auto) operator*() const { return *rtk_u(v); }
};

#define RTK_NONE32 0xFFFFFFFFu
#define RTK_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

// per-unitig flag word
#define RTK_F_EDGE_MASK 0xFFu      // bits 4..7 fw successor-base mask, bits 0..3 bw mask (UnitigData.hpp:275-289)
#define RTK_F_SHORT_CYCLE (1u << 8) // UnitigData.hpp:302-305
#define RTK_F_BRANCHING (1u << 9)   // UnitigData.hpp:407-410
#define RTK_F_AMBIGUITY (1u << 10)  // UnitigData.hpp:483-486

struct GraphView {
    U<int32_t> k;
    U<uint32_t> n_unitigs;
    U<uint64_t> n_kmers;
    U<uint64_t> ht_slots;         // slots of the k-mer table (any number: a hash is mapped to its slot by multiply-high, rtk_ht_slot)
    U<const uint64_t*> useq;      // unitig bases, 2 bits each, base i of the pool at bits [2*(i&31), +1] of word i>>5
    U<const uint64_t*> uoff;      // [n+1] first base of unitig u in the pool
    U<const uint32_t*> adj;       // [n*8] fw A,C,G,T then reverse-strand A,C,G,T: neighbour unitig<<1|strand or RTK_NONE32
    U<const uint32_t*> flags;     // [n]
    U<const uint32_t*> kcov;      // [n] round(cov/(size-k+1)) (UnitigData.hpp:396-399), precomputed in double on the host
    U<const uint32_t*> card;      // [n] |global| + |local|
    U<const uint64_t*> loff;      // [n+1] local colour set of u = col[loff[u] .. loff[u+1])
    U<const int32_t*> gid;        // [n] global colour set id or -1
    U<const uint64_t*> goff;      // [n_global+1] global set g = col[goff[g] .. goff[g+1])
    U<const uint32_t*> col;       // sorted u32 pair ids
    U<const uint64_t*> ht;        // [2*slots] {canonical k-mer, unitig<<32 | dist<<1 | stored_is_canonical}, empty key = RTK_EMPTY_KEY
    U<const uint64_t*> bf;        // [bf_mask+1] blocked Bloom filter over the canonical k-mers (2 bits of one 64-bit word per k-mer)
    U<uint64_t> bf_mask;
    U<const uint64_t*> bf1;       // cache-sized first-level filter in front of `bf`: one bit per k-mer in a bit array of bf1_mask + 1 bits (a single all-ones word when disabled)
    U<uint64_t> bf1_mask;
    U<const uint64_t*> cycoff;    // [n+1] compact cycles of unitig u = cyc[cycoff[u] .. cycoff[u+1]) (NUL-terminated strings of successor bases)
    U<const char*> cyc;
    U<const uint64_t*> amb;       // SNP annotations: amb[u]..amb[u+1] index the entries of unitig u, entry j = amb[n_unitigs + 1 + j] = position<<4 | IUPAC index, by (position, code)
    U<const uint64_t*> hap;       // haplotype ids: hap[u]..hap[u+1] index the ids of unitig u at hap[n_unitigs + 1 + j] (not read by `correct` without -p/-P)
    U<uint64_t> n_amb;            // number of annotation entries (0: getAmbiguityVector / fixAmbiguity are identities)
    U<const uint64_t*> hx;        // [hx_mask+1] half-k-mer index for the 1-edit search: slot = CANONICAL h-mer << 34 | first (h = (k-1)/2; the smaller of an h-mer of the forward unitig
    U<uint64_t> hx_mask;          //   sequences and its reverse complement), empty = RTK_EMPTY_KEY; hxl[first] = number of places, then two words per place: flanks (h + 1 bases in front << 32 | h + 1 bases behind, as the unitig spells them), reversed<<63 | unitig<<32 | front-exists<<31 | offset.
    U<const uint64_t*> hxl;       //   hx_mask == 0: no index (the search spells the variants instead)
};

RTK_HD uint32_t rtk_ulen(const GraphView& g, uint32_t u) { return static_cast<uint32_t>(g.uoff[u + 1] - g.uoff[u]); }
RTK_HD uint32_t rtk_nkm(const GraphView& g, uint32_t u) { return rtk_ulen(g, u) - static_cast<uint32_t>(g.k) + 1u; }
RTK_HD uint32_t rtk_base(const GraphView& g, uint64_t pos) { return static_cast<uint32_t>((g.useq[pos >> 5] >> (2 * (pos & 31))) & 3ull); }

RTK_HD uint64_t rtk_hash64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

RTK_HD uint64_t rtk_revcomp(uint64_t x, int k) {
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

// Unitig mapping (restates the fields of Bifrost's const_UnitigMap the hot path reads).
struct UMap {
    uint32_t unitig; // RTK_NONE32 == isEmpty
    uint32_t dist;
    uint32_t len;
    uint32_t strand; // 1 = forward
};

RTK_HD UMap rtk_um_empty() { UMap u; u.unitig = RTK_NONE32; u.dist = 0; u.len = 0; u.strand = 1; return u; }
RTK_HD bool rtk_um_is_empty(const UMap& u) { return u.unitig == RTK_NONE32; }
RTK_HD bool rtk_um_eq(const UMap& a, const UMap& b) { return a.unitig == b.unitig && a.dist == b.dist && a.len == b.len && a.strand == b.strand; }

// packed anchor hit: unitig<<33 | dist<<1 | strand ; all ones = no hit
#define RTK_NO_HIT 0xFFFFFFFFFFFFFFFFull
RTK_HD uint64_t rtk_pack_hit(uint32_t unitig, uint32_t dist, uint32_t strand) { return (static_cast<uint64_t>(unitig) << 33) | (static_cast<uint64_t>(dist) << 1) | (strand & 1u); }
RTK_HD UMap rtk_unpack_hit(uint64_t h) { UMap u; u.unitig = static_cast<uint32_t>(h >> 33); u.dist = static_cast<uint32_t>((h >> 1) & 0xFFFFFFFFull); u.len = 1; u.strand = static_cast<uint32_t>(h & 1ull); return u; }

// Exact k-mer lookup (Bifrost find(km,false) [A1]): fw = k-mer code in read orientation. *n_probes = 16-byte table slots visited
// (0 when the pre-filter already answered).
// slot of a hash in a table of `slots` entries: floor(hash * slots / 2^64) -- no power-of-two table sizes needed (a 3 Gb graph: 69 GB at load 0.7
// instead of 137 GB), the high hash bits decide; next slot of the linear probe
RTK_HD uint64_t rtk_ht_slot(uint64_t hh, uint64_t slots) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTK_SIM)
    return __umul64hi(hh, slots);
#else
    return static_cast<uint64_t>((static_cast<unsigned __int128>(hh) * static_cast<unsigned __int128>(slots)) >> 64);
#endif
}
RTK_HD uint64_t rtk_ht_next(uint64_t i, uint64_t slots) { return i + 1 == slots ? 0 : i + 1; }

RTK_HD uint64_t rtk_find_kmer(const GraphView& g, uint64_t fw, uint32_t* n_probes) {
    const uint64_t rc = rtk_revcomp(fw, g.k);
    const uint64_t can = fw < rc ? fw : rc;
    const uint64_t hh = rtk_hash64(can);
    { // first level: one bit in an array small enough to stay in L2
        const uint64_t b1 = (hh >> 12) & g.bf1_mask;
        if (!((g.bf1[b1 >> 6] >> (b1 & 63ull)) & 1ull)) { if (n_probes) *n_probes = 0; return RTK_NO_HIT; }
    }
    { // pre-filter: absent k-mers (the bulk of the 1-edit variants) stop here after one 8-byte read
        const uint64_t bits = (1ull << (hh & 63)) | (1ull << ((hh >> 6) & 63));
        if ((g.bf[(hh >> 32) & g.bf_mask] & bits) != bits) { if (n_probes) *n_probes = 0; return RTK_NO_HIT; }
    }
    uint64_t i = rtk_ht_slot(hh, g.ht_slots);
    uint32_t np = 0;
    while (true) {
        const uint64_t key = g.ht[2 * i];
        ++np;
        if (key == can) {
            const uint64_t v = g.ht[2 * i + 1];
            const uint32_t stored_is_can = static_cast<uint32_t>(v & 1ull), query_is_can = (fw <= rc) ? 1u : 0u;
            if (n_probes) *n_probes = np;
            return rtk_pack_hit(static_cast<uint32_t>(v >> 32), static_cast<uint32_t>((v & 0xFFFFFFFFull) >> 1), stored_is_can == query_is_can ? 1u : 0u);
        }
        if (key == RTK_EMPTY_KEY) { if (n_probes) *n_probes = np; return RTK_NO_HIT; }
        i = rtk_ht_next(i, g.ht_slots);
    }
}


// ---- k in 33..63 (second pass, k2 = 63): two-word k-mers ------------------------------------------------------------------------
// Code = 2k bits right-aligned in {hi, lo}, first base in the most significant bits (k <= 32: hi = 0 and lo is the one-word code).
// The table keeps its 16-byte slots: the key word is a 64-bit fingerprint of the canonical k-mer and a fingerprint match is
// confirmed against the 2-bit unitig sequence the value word points at, so lookups stay exact.
struct RtkKm { uint64_t hi, lo; };
RTK_HD RtkKm rtk_km_zero() { RtkKm x; x.hi = 0; x.lo = 0; return x; }
RTK_HD bool rtk_km_eq(const RtkKm& a, const RtkKm& b) { return a.hi == b.hi && a.lo == b.lo; }
RTK_HD bool rtk_km_less(const RtkKm& a, const RtkKm& b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }
RTK_HD RtkKm rtk_km_mask(int k) { RtkKm m; m.lo = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull); m.hi = (k <= 32) ? 0ull : ((1ull << (2 * k - 64)) - 1ull); return m; }
RTK_HD RtkKm rtk_km_push(const RtkKm& x, uint64_t b, int k) { // append base b (0..3) at the end, drop the first
    const RtkKm m = rtk_km_mask(k); RtkKm r; r.hi = ((x.hi << 2) | (x.lo >> 62)) & m.hi; r.lo = ((x.lo << 2) | b) & m.lo; return r;
}
RTK_HD RtkKm rtk_km_revcomp(const RtkKm& x, int k) {
    if (k <= 32) { RtkKm r; r.hi = 0; r.lo = rtk_revcomp(x.lo, k); return r; }
    const uint64_t a = rtk_revcomp(x.lo, 32), b = rtk_revcomp(x.hi, 32); // reverse complement of all 128 bits = {a, b}; the k-mer sits in its top 2k bits
    const int s = 128 - 2 * k; // 2..62
    RtkKm r; r.lo = (b >> s) | (a << (64 - s)); r.hi = a >> s; return r;
}
RTK_HD uint64_t rtk_km_hash(const RtkKm& can) { return rtk_hash64(can.lo ^ rtk_hash64(can.hi ^ 0x9e3779b97f4a7c15ull)); }
RTK_HD uint64_t rtk_km_fingerprint(const RtkKm& can) { const uint64_t f = rtk_hash64(can.lo + 0x9e3779b97f4a7c15ull) ^ rtk_hash64(can.hi ^ 0xd6e8feb86659fd93ull); return f == RTK_EMPTY_KEY ? 0ull : f; }
// reverse complement code of the k-mer starting at base `pos` of the 2-bit unitig pool (base p at bits 2(p & 31) of word p >> 5, i.e.
// last base in the most significant bits: complementing it IS the reverse-complement code). k in 33..63.
RTK_HD RtkKm rtk_km_rc_of_unitig(const GraphView& g, uint64_t pos, int k) {
    const uint64_t w = pos >> 5; const int sh = static_cast<int>(2 * (pos & 31));
    const uint64_t x0 = g.useq[w], x1 = g.useq[w + 1];
    RtkKm l;
    if (sh == 0) { l.lo = x0; l.hi = x1; }
    else { l.lo = (x0 >> sh) | (x1 << (64 - sh)); l.hi = x1 >> sh; if (sh + 2 * k > 128) l.hi |= g.useq[w + 2] << (64 - sh); }
    const RtkKm m = rtk_km_mask(k);
    l.lo = ~l.lo & m.lo; l.hi = ~l.hi & m.hi;
    return l;
}
RTK_HD uint64_t rtk_find_kmer_wide(const GraphView& g, const RtkKm& fw, uint32_t* n_probes) {
    const RtkKm rc = rtk_km_revcomp(fw, g.k);
    const RtkKm can = rtk_km_less(fw, rc) ? fw : rc;
    const uint64_t hh = rtk_km_hash(can);
    { const uint64_t b1 = (hh >> 12) & g.bf1_mask; if (!((g.bf1[b1 >> 6] >> (b1 & 63ull)) & 1ull)) { if (n_probes) *n_probes = 0; return RTK_NO_HIT; } }
    { const uint64_t bits = (1ull << (hh & 63)) | (1ull << ((hh >> 6) & 63)); if ((g.bf[(hh >> 32) & g.bf_mask] & bits) != bits) { if (n_probes) *n_probes = 0; return RTK_NO_HIT; } }
    const uint64_t fp = rtk_km_fingerprint(can);
    uint64_t i = rtk_ht_slot(hh, g.ht_slots);
    uint32_t np = 0;
    while (true) {
        const uint64_t key = g.ht[2 * i];
        ++np;
        if (key == fp) {
            const uint64_t v = g.ht[2 * i + 1];
            const uint32_t u = static_cast<uint32_t>(v >> 32), off = static_cast<uint32_t>((v & 0xFFFFFFFFull) >> 1);
            const RtkKm urc = rtk_km_rc_of_unitig(g, g.uoff[u] + off, g.k);
            if (rtk_km_eq(urc, rc)) { if (n_probes) *n_probes = np; return rtk_pack_hit(u, off, 1u); } // the query reads like the unitig
            if (rtk_km_eq(urc, fw)) { if (n_probes) *n_probes = np; return rtk_pack_hit(u, off, 0u); } // the query is its reverse complement
        }
        if (key == RTK_EMPTY_KEY) { if (n_probes) *n_probes = np; return RTK_NO_HIT; }
        i = rtk_ht_next(i, g.ht_slots);
    }
}
RTK_HD uint64_t rtk_find_km(const GraphView& g, const RtkKm& fw, uint32_t* n_probes) { return g.k <= 31 ? rtk_find_kmer(g, fw.lo, n_probes) : rtk_find_kmer_wide(g, fw, n_probes); }

// Split form of rtk_find_kmer for callers that want several filter reads in flight before any of them is consumed.
RTK_HD void rtk_kmer_prepare(uint64_t fw, int k, uint64_t* can, uint64_t* hh, uint32_t* query_is_can) {
    const uint64_t rc = rtk_revcomp(fw, k);
    *can = fw < rc ? fw : rc; *query_is_can = (fw <= rc) ? 1u : 0u; *hh = rtk_hash64(*can);
}
RTK_HD bool rtk_filter1_pass(uint64_t word1, uint64_t hh, uint64_t bf1_mask) { const uint64_t b1 = (hh >> 12) & bf1_mask; return (word1 >> (b1 & 63ull)) & 1ull; }
RTK_HD bool rtk_filter_pass(uint64_t word, uint64_t hh) { const uint64_t bits = (1ull << (hh & 63)) | (1ull << ((hh >> 6) & 63)); return (word & bits) == bits; }
RTK_HD uint64_t rtk_table_probe_from(const GraphView& g, uint64_t can, uint64_t i, uint32_t query_is_can, uint32_t* n_slots);
RTK_HD uint64_t rtk_table_lookup(const GraphView& g, uint64_t can, uint64_t hh, uint32_t query_is_can, uint32_t* n_slots) { return rtk_table_probe_from(g, can, rtk_ht_slot(hh, g.ht_slots), query_is_can, n_slots); }
// the linear probe from slot i on (a caller that has read the home slot itself continues at rtk_ht_next of it)
RTK_HD uint64_t rtk_table_probe_from(const GraphView& g, uint64_t can, uint64_t i, uint32_t query_is_can, uint32_t* n_slots) {
    uint32_t np = 0;
    while (true) {
        const uint64_t key = g.ht[2 * i]; ++np;
        if (key == can) { const uint64_t v = g.ht[2 * i + 1]; *n_slots = np; return rtk_pack_hit(static_cast<uint32_t>(v >> 32), static_cast<uint32_t>((v & 0xFFFFFFFFull) >> 1), (static_cast<uint32_t>(v & 1ull) == query_is_can) ? 1u : 0u); }
        if (key == RTK_EMPTY_KEY) { *n_slots = np; return RTK_NO_HIT; }
        i = rtk_ht_next(i, g.ht_slots);
    }
}

#endif
