// Lookup structures of the flat graph built ON THE DEVICE from the packed unitigs (rtk_graph_tables.hip): the k-mer table with its two presence
// filters, the half-k-mer index and the adjacency. Called by rtk_graph_upload for a graph loaded with RTK_LOAD_DEVICE_TABLES.
#ifndef RTK_GRAPH_TABLES_H
#define RTK_GRAPH_TABLES_H

#include <stdint.h>

namespace rtk {

struct DeviceTables {                     // device pointers (hipMalloc'd here, owned by the caller afterwards) and their sizes in bytes
    void* ht; uint64_t ht_bytes, ht_slots;
    void* bf; uint64_t bf_bytes;
    void* bf1; uint64_t bf1_bytes;
    void* hx; uint64_t hx_bytes;
    void* hxl; uint64_t hxl_bytes;
    void* adj; uint64_t adj_bytes;
    double seconds[4];                    // k-mer table + filters, half-k-mer index, adjacency, total
};

// d_useq / d_uoff: the packed unitigs on the current device (GraphView::useq / uoff); throws std::runtime_error (a k-mer that occurs twice: the
// file is no compacted de Bruijn graph for this k; out of memory; more than 2^34 list words)
void device_tables_build(const uint64_t* d_useq, const uint64_t* d_uoff, uint32_t n_unitigs, uint64_t n_bases, uint64_t n_kmers, int k, DeviceTables* out);

// the k-mer table alone, without filters, at load 0.7 (16-byte slots {canonical k-mer, unitig << 32 | offset << 1 | stored_is_canonical}, slot of a hash by
// multiply-high): what the index build's colouring pass looks the read k-mers up in. One-word k-mers.
void device_kmer_table(const uint64_t* d_useq, const uint64_t* d_uoff, uint32_t n_unitigs, uint64_t n_bases, uint64_t n_kmers, int k, void** ht_out, uint64_t* slots_out);

} // namespace rtk

#endif
