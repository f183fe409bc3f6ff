// libratatosk_hip.so: HIP kernels for gfx950 + the C ABI of include/ratatosk_hip.h.
// (Compiled a second time with -DRTK_SIM by tests/hostsim into a developer simulator; see rtk_wave.h.)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/ratatosk_hip.h"
#include "../host/flat_graph.hpp"
#include "rtk_graph_tables.h"
#include "rtk_mem.h"
#include "rtk_myers.h"
#include "rtk_myers_lane.h"
#include "rtk_types.h"
#include "rtk_wave.h"

#ifdef RTK_SIM
thread_local int rtk_sim_block_id = 0;
thread_local int rtk_sim_site = 0;
std::atomic<unsigned long long> rtk_sim_site_stat[32][8];
std::atomic<unsigned long long> rl_sim_acc[4];
extern "C" void rtk_sim_lane_accesses(unsigned long long* out, int reset) { for (int i = 0; i < 4; ++i) { out[i] = rl_sim_acc[i].load(); if (reset) rl_sim_acc[i] = 0; } }
extern "C" void rtk_sim_site_stats(unsigned long long* out, int reset) { for (int i = 0; i < 32; ++i) for (int j = 0; j < 8; ++j) { out[8 * i + j] = rtk_sim_site_stat[i][j].load(); if (reset) rtk_sim_site_stat[i][j] = 0; } }
#endif

// ------------------------------------------------------------------------------------------------ error handling
static thread_local std::string g_last_error;
int rtk_fail(int code, const std::string& msg) { g_last_error = msg; return code; } // (also used by the other translation units of the library)

extern "C" const char* rtk_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* rtk_version(void) {
#ifdef RTK_SIM
    return "ratatosk-mi355x 0.6 (host simulator)";
#else
    return "ratatosk-mi355x 0.6 (gfx950)";
#endif
}
extern "C" int rtk_api_revision(void) { return RTK_API_REVISION; }
extern "C" void rtk_free(void* p) { free(p); }
extern "C" void rtk_free_many(void** p, uint32_t n) { if (p) for (uint32_t i = 0; i < n; ++i) { free(p[i]); p[i] = nullptr; } }

// ------------------------------------------------------------------------------------------------ graph object
struct rtk_graph {
    rtk::FlatGraph host;
    bool has_host = false, on_device = false, owns_buffers = true;
    uint32_t moved = 0;         // bit i: flat buffer i lives in caller-owned memory (rtk_graph_move_buffer)
    double table_seconds = 0.0; // device build of the lookup structures (RTK_LOAD_DEVICE_TABLES)
    int device = -1;
    void* dbuf[rtk::RTK_N_BUFS];
    uint64_t dbytes[rtk::RTK_N_BUFS];
    GraphView dview;
    rtk_graph_info info;
    // per-wave work areas, kept across batches: slot 0 for the seed stage, slot 1 for the region stage. A stage holds its slot's lock
    // while it runs, so the seed stage of one batch and the region stage of another overlap, two stages of one kind queue up.
    void* scratch[3] = {nullptr, nullptr, nullptr}; uint64_t scratch_bytes_[3] = {0, 0, 0}; std::mutex scratch_lock[3]; // 0 seed stage, 1 region stage (wave kernel), 2 region stage (lane kernel; under lock 1)
    // device buffers of finished batches, by size: a ticket's ~25 buffers are taken from here instead of hipMalloc / hipFree, which
    // cost milliseconds each and (hipFree) wait for the whole device, i.e. for the other batch's kernels
    std::mutex pool_lock; std::multimap<uint64_t, void*> pool; uint64_t pool_bytes = 0;
    std::atomic<int> refs{1}; // the caller's handle + one per live batch
    void* pool_take(uint64_t bytes, uint64_t* got) {
        bytes = (bytes + 4095) / 4096 * 4096;
        // coarse size classes above 1 MiB (eight per octave, <= 12.5 % over): the buffers of consecutive tickets differ by a fraction of a per cent, and a parked buffer only serves a
        // request it is at least as big as -- with exact sizes every second ticket of a steady run missed the pool and called hipMalloc, which stalls every stream of the device (round 6:
        // tickets of distinct reads on the configs[4] graph, 64 instead of 51 ms per ticket)
        if (bytes >= (1ull << 20)) { uint64_t step = 1ull << 17; while ((step << 4) <= bytes) step <<= 1; bytes = (bytes + step - 1) / step * step; }
        { std::lock_guard<std::mutex> h(pool_lock);
          std::multimap<uint64_t, void*>::iterator it = pool.lower_bound(bytes);
          if (it != pool.end() && it->first <= bytes + bytes / 4 + (1u << 20)) { void* p = it->second; *got = it->first; pool_bytes -= it->first; pool.erase(it); return p; } }
        if (bytes >= (64u << 20) && getenv("RTK_TRACE")) { const auto t0 = std::chrono::steady_clock::now(); void* p = rtk_dmalloc(bytes); fprintf(stderr, "[rtk trace] pool_take: %.2f GB of new device memory in %.1f ms\n", bytes / 1073741824.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); *got = bytes; return p; }
        *got = bytes; return rtk_dmalloc(bytes);
    }
    void pool_give(void* p, uint64_t bytes) {
        std::lock_guard<std::mutex> h(pool_lock);
        if (pool_bytes + bytes > (64ull << 30)) { if (bytes >= (64u << 20) && getenv("RTK_TRACE")) fprintf(stderr, "[rtk trace] pool_give: %.2f GB freed (64 GB parked already)\n", bytes / 1073741824.0); rtk_dfree(p); return; } // keep at most 64 GB parked (eleven second-pass tickets of ~5 GB: eight in flight, three being formatted)
        pool.insert(std::make_pair(bytes, p)); pool_bytes += bytes;
    }
    // work areas of the phasing step (second pass): one per ticket in flight, so that the hour-glass launches of several tickets (each as
    // long as its longest read) overlap instead of queueing behind one lock; kept until the graph goes (tens of GB each: never hipFree'd mid-run)
    std::vector<std::pair<void*, uint64_t> > phase_free;
    void* phase_take(uint64_t bytes, uint64_t* got); // (defined below the reserved-slab list)
    void phase_give(void* p, uint64_t bytes) { std::lock_guard<std::mutex> h(pool_lock); phase_free.push_back(std::make_pair(p, bytes)); }
    void pool_clear() { std::lock_guard<std::mutex> h(pool_lock); for (std::multimap<uint64_t, void*>::iterator it = pool.begin(); it != pool.end(); ++it) rtk_dfree(it->second); pool.clear(); pool_bytes = 0;
                        for (size_t i = 0; i < phase_free.size(); ++i) rtk_dfree(phase_free[i].first); phase_free.clear();
                        for (std::multimap<uint64_t, void*>::iterator it = hpool.begin(); it != hpool.end(); ++it) rtk_hfree_pinned(it->second); hpool.clear(); hpool_bytes = 0; }
    // pinned host staging buffers of finished batches (packed reads in, packed records out): hipHostMalloc costs milliseconds per call
    std::multimap<uint64_t, void*> hpool; uint64_t hpool_bytes = 0;
    void* stage_take(uint64_t bytes, uint64_t* got) {
        bytes = (bytes + (1u << 20) - 1) >> 20 << 20;
        // coarse size classes: the staging buffers of consecutive tickets differ by a few per cent, and a parked buffer only serves a request it is at least as big as -- with exact
        // sizes two of three requests of a steady run missed the pool and pinned new memory (hipHostMalloc of 67 MB: 22 ms of the worker, round 5 trace)
        if (bytes >= (32ull << 20)) bytes = (bytes + (16ull << 20) - 1) / (16ull << 20) * (16ull << 20); else if (bytes >= (4ull << 20)) bytes = (bytes + (4ull << 20) - 1) / (4ull << 20) * (4ull << 20);
        { std::lock_guard<std::mutex> h(pool_lock);
          std::multimap<uint64_t, void*>::iterator it = hpool.lower_bound(bytes);
          if (it != hpool.end() && it->first <= 2 * bytes + (4u << 20)) { void* p = it->second; *got = it->first; hpool_bytes -= it->first; hpool.erase(it); return p; } }
        if (getenv("RTK_TRACE")) { const auto t0 = std::chrono::steady_clock::now(); void* p = rtk_hmalloc_pinned(bytes); fprintf(stderr, "[rtk trace] stage_take: %.1f MB of new pinned host memory in %.1f ms\n", bytes / 1048576.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); *got = bytes; return p; }
        *got = bytes; return rtk_hmalloc_pinned(bytes);
    }
    void stage_give(void* p, uint64_t bytes) {
        std::lock_guard<std::mutex> h(pool_lock);
        if (hpool_bytes + bytes > (8ull << 30)) { rtk_hfree_pinned(p); return; }
        hpool.insert(std::make_pair(bytes, p)); hpool_bytes += bytes;
    }
    // The region stage's graph-wide work areas (slot 1) as two halves: a small first-pass ticket that meets another ticket at the region stage runs its persistent
    // kernel on 2 048 waves in one half while the other half serves the next ticket -- a launch of a few Mb lasts as long as its heaviest region with most wave
    // slots idle (round 6, tickets of the reference's size). A big ticket (or a lone one, or one whose half would be too small) takes both.
    std::mutex rs_m; std::condition_variable rs_cv; bool rs_busy[2] = {false, false}; int rs_whole_waiting = 0;
    int region_slab_take(bool half_if_contended) { // 0 / 1: that half; 2: both
        std::unique_lock<std::mutex> lk(rs_m);
        if (half_if_contended && (rs_busy[0] || rs_busy[1] || rs_whole_waiting)) {
            rs_cv.wait(lk, [&] { return rs_whole_waiting == 0 && (!rs_busy[0] || !rs_busy[1]); });
            const int h = rs_busy[0] ? 1 : 0; rs_busy[h] = true; return h;
        }
        ++rs_whole_waiting; rs_cv.wait(lk, [&] { return !rs_busy[0] && !rs_busy[1]; }); --rs_whole_waiting;
        rs_busy[0] = rs_busy[1] = true; return 2;
    }
    void region_slab_give(int what) { { std::lock_guard<std::mutex> lk(rs_m); if (what == 2) rs_busy[0] = rs_busy[1] = false; else rs_busy[what] = false; } rs_cv.notify_all(); }
    // Tickets of concurrent rtk_correct_batch callers that are merged into one launch (rtk_pipeline_run.inc, "ticket coalescing"): the queue of waiting
    // tickets, whether a caller is gathering a group right now, the groups whose batch is being created / run / fetched, how many tickets the last group held
    std::mutex co_m; std::condition_variable co_cv; std::vector<struct CoTicket*> co_q; bool co_gathering = false; int co_running = 0, co_pre_region = 0; uint32_t co_last_group = 0, co_inflight_tickets = 0;
    unsigned long long co_groups = 0, co_tickets = 0; // (statistics: rtk_coalesce_stats)
    rtk_graph() { for (int i = 0; i < rtk::RTK_N_BUFS; ++i) { dbuf[i] = nullptr; dbytes[i] = 0; } memset(&dview, 0, sizeof(dview)); memset(&info, 0, sizeof(info)); }
};

static void graph_set_view(rtk_graph* g) {
    GraphView& v = g->dview;
    v.k = g->info.k; v.n_unitigs = static_cast<uint32_t>(g->info.n_unitigs); v.n_kmers = g->info.n_kmers; v.ht_slots = g->info.table_slots;
    v.useq = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_USEQ]); v.uoff = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_UOFF]);
    v.adj = static_cast<const uint32_t*>(g->dbuf[rtk::RTK_BUF_ADJ]); v.flags = static_cast<const uint32_t*>(g->dbuf[rtk::RTK_BUF_FLAGS]);
    v.kcov = static_cast<const uint32_t*>(g->dbuf[rtk::RTK_BUF_KCOV]); v.card = static_cast<const uint32_t*>(g->dbuf[rtk::RTK_BUF_CARD]);
    v.loff = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_LOFF]); v.gid = static_cast<const int32_t*>(g->dbuf[rtk::RTK_BUF_GID]);
    v.goff = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_GOFF]); v.col = static_cast<const uint32_t*>(g->dbuf[rtk::RTK_BUF_COL]);
    v.ht = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_HT]);
    v.bf = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_BF]); v.bf_mask = g->dbytes[rtk::RTK_BUF_BF] / 8 - 1;
    v.bf1 = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_BF1]); v.bf1_mask = g->dbytes[rtk::RTK_BUF_BF1] * 8 - 1;
    v.cycoff = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_CYCOFF]); v.cyc = static_cast<const char*>(g->dbuf[rtk::RTK_BUF_CYC]);
    v.hap = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_HAP]);
    v.amb = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_AMB]); v.n_amb = g->dbytes[rtk::RTK_BUF_AMB] / 8 - (static_cast<uint64_t>(v.n_unitigs) + 1);
    v.hx = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_HX]); v.hx_mask = g->dbytes[rtk::RTK_BUF_HX] / 8 - 1; v.hxl = static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_HXL]);
}

extern "C" int rtk_graph_load2(const char* unitig_fasta_gz, const char* rtsk, int k, int n_threads, uint32_t flags, rtk_graph** out) {
    if (!unitig_fasta_gz || !rtsk || !out) return rtk_fail(RTK_ERR_ARG, "rtk_graph_load: null argument");
    std::unique_ptr<rtk_graph> g(new rtk_graph());
#ifdef RTK_SIM
    flags &= ~static_cast<uint32_t>(RTK_LOAD_DEVICE_TABLES); // (the simulator has no device: its tables are the host's)
#endif
    try { g->host.load(unitig_fasta_gz, rtsk, k, n_threads, (flags & RTK_LOAD_DEVICE_TABLES) != 0); }
    catch (const std::exception& e) { return rtk_fail(RTK_ERR_FORMAT, std::string("rtk_graph_load: ") + e.what()); }
    g->has_host = true;
    rtk_graph_info& i = g->info;
    i.k = k; i.device = -1; i.n_unitigs = g->host.n_unitigs(); i.n_kmers = g->host.n_kmers; i.n_bases = g->host.uoff.back();
    i.n_colour_ids = g->host.col.size() - 1; i.n_global_sets = g->host.n_global; i.table_slots = g->host.ht.size() / 2; i.hbm_bytes = g->host.bytes();
    if (g->host.tables_deferred) i.table_slots = rtk::table_sizes(k, g->host.n_kmers, g->host.uoff.back()).ht_slots; // (hbm_bytes: set by rtk_graph_upload, which builds the tables)
    i.max_km_cov_top = g->host.max_km_cov_top;
    *out = g.release();
    return RTK_OK;
}
extern "C" int rtk_graph_load(const char* unitig_fasta_gz, const char* rtsk, int k, int n_threads, rtk_graph** out) { return rtk_graph_load2(unitig_fasta_gz, rtsk, k, n_threads, 0u, out); }

extern "C" int rtk_graph_shell(int k, rtk_graph** out) {
    if (!out) return rtk_fail(RTK_ERR_ARG, "rtk_graph_shell: null argument");
    rtk_graph* g = new rtk_graph(); g->info.k = k; *out = g; return RTK_OK;
}

extern "C" int rtk_graph_n_buffers(const rtk_graph*) { return rtk::RTK_N_BUFS; }

static int require_device(int device) {
    const int n = rtk_device_count();
    if (n <= 0) return rtk_fail(RTK_ERR_NO_DEVICE, "no HIP device visible: the correction path has no CPU fallback");
    if (device < 0 || device >= n) return rtk_fail(RTK_ERR_ARG, "bad device ordinal");
    return RTK_OK;
}

extern "C" int rtk_graph_alloc_buffers(rtk_graph* g, int device, const uint64_t* bytes, int n, const rtk_graph_info* info) {
    if (!g || !bytes || !info || n != rtk::RTK_N_BUFS) return rtk_fail(RTK_ERR_ARG, "rtk_graph_alloc_buffers: bad argument");
    int rc = require_device(device); if (rc) return rc;
    try {
        rtk_set_device(device);
        for (int i = 0; i < n; ++i) { g->dbuf[i] = rtk_dmalloc(bytes[i]); g->dbytes[i] = bytes[i]; }
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, e.what()); }
    g->info = *info; g->info.device = device; g->device = device;
    return RTK_OK;
}

extern "C" int rtk_graph_attach_buffers(rtk_graph* g, int device, void* const* dev_ptrs, const uint64_t* bytes, int n, const rtk_graph_info* info) {
    if (!g || !dev_ptrs || !bytes || !info || n != rtk::RTK_N_BUFS) return rtk_fail(RTK_ERR_ARG, "rtk_graph_attach_buffers: bad argument");
    int rc = require_device(device); if (rc) return rc;
    for (int i = 0; i < n; ++i) { if (g->dbuf[i] && g->owns_buffers) rtk_dfree(g->dbuf[i]); g->dbuf[i] = dev_ptrs[i]; g->dbytes[i] = bytes[i]; }
    g->owns_buffers = false;
    const bool had_host = g->has_host;
    if (!had_host) g->info = *info;
    g->info.device = device; g->device = device;
    return RTK_OK;
}

extern "C" int rtk_graph_buffer_bytes(const rtk_graph* g, uint64_t* bytes, int n) {
    if (!g || !bytes || n != rtk::RTK_N_BUFS || !(g->has_host || g->on_device)) return rtk_fail(RTK_ERR_ARG, "rtk_graph_buffer_bytes: needs a loaded graph");
    if (g->on_device) { for (int i = 0; i < n; ++i) bytes[i] = g->dbytes[i]; return RTK_OK; }
    if (g->host.tables_deferred) return rtk_fail(RTK_ERR_ARG, "rtk_graph_buffer_bytes: the tables of this graph are built by rtk_graph_upload (RTK_LOAD_DEVICE_TABLES): call it first");
    const rtk::FlatGraph& h = g->host;
    const uint64_t b[rtk::RTK_N_BUFS] = { 8 * h.useq.size(), 8 * h.uoff.size(), 4 * h.adj.size(), 4 * h.flags.size(), 4 * h.kcov.size(), 4 * h.card.size(), 8 * h.loff.size(), 4 * h.gid.size(), 8 * h.goff.size(), 4 * h.col.size(), 8 * h.ht.size(), 8 * h.bf.size(), 8 * h.cycoff.size(), 8 * h.cyc.size(), 8 * h.bf1.size(), 8 * h.amb.size(), 8 * h.hx.size(), 8 * h.hxl.size(), 8 * h.hap.size() };
    for (int i = 0; i < n; ++i) bytes[i] = b[i];
    return RTK_OK;
}

extern "C" int rtk_graph_upload(rtk_graph* g, int device) {
    if (!g || !g->has_host) return rtk_fail(RTK_ERR_ARG, "rtk_graph_upload: graph has no host image");
    int rc = require_device(device); if (rc) return rc;
    const rtk::FlatGraph& h = g->host;
    const void* src[rtk::RTK_N_BUFS] = { h.useq.data(), h.uoff.data(), h.adj.data(), h.flags.data(), h.kcov.data(), h.card.data(), h.loff.data(), h.gid.data(), h.goff.data(), h.col.data(), h.ht.data(), h.bf.data(), h.cycoff.data(), h.cyc.data(), h.bf1.data(), h.amb.data(), h.hx.data(), h.hxl.data(), h.hap.data() };
    const uint64_t bytes[rtk::RTK_N_BUFS] = { 8 * h.useq.size(), 8 * h.uoff.size(), 4 * h.adj.size(), 4 * h.flags.size(), 4 * h.kcov.size(), 4 * h.card.size(), 8 * h.loff.size(), 4 * h.gid.size(), 8 * h.goff.size(), 4 * h.col.size(), 8 * h.ht.size(), 8 * h.bf.size(), 8 * h.cycoff.size(), 8 * h.cyc.size(), 8 * h.bf1.size(), 8 * h.amb.size(), 8 * h.hx.size(), 8 * h.hxl.size(), 8 * h.hap.size() };
    static const int built_here[6] = { rtk::RTK_BUF_HT, rtk::RTK_BUF_BF, rtk::RTK_BUF_BF1, rtk::RTK_BUF_HX, rtk::RTK_BUF_HXL, rtk::RTK_BUF_ADJ };
    const bool deferred = h.tables_deferred;
    try {
        rtk_set_device(device);
        for (int i = 0; i < rtk::RTK_N_BUFS; ++i) {
            bool skip = false; for (int j = 0; j < 6 && deferred; ++j) skip = skip || built_here[j] == i;
            if (skip) { // built below, in HBM; a second upload (a retry after a device error, another device) frees the tables of the first and builds them again
                if (g->dbuf[i]) { if (!g->owns_buffers) return rtk_fail(RTK_ERR_ARG, "rtk_graph_upload: the table buffers of this graph were attached by the caller (rtk_graph_attach_buffers): it cannot be uploaded again with RTK_LOAD_DEVICE_TABLES"); rtk_dfree(g->dbuf[i]); g->dbuf[i] = nullptr; g->dbytes[i] = 0; }
                continue; }
            if (!g->dbuf[i]) { g->dbuf[i] = rtk_dmalloc(bytes[i]); g->dbytes[i] = bytes[i]; } rtk_h2d(g->dbuf[i], src[i], bytes[i]);
        }
#ifndef RTK_SIM
        if (deferred) { // the lookup structures from the packed unitigs, in HBM (hip/rtk_graph_tables.hip)
            rtk::DeviceTables t; memset(&t, 0, sizeof(t));
            try { rtk::device_tables_build(static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_USEQ]), static_cast<const uint64_t*>(g->dbuf[rtk::RTK_BUF_UOFF]), h.n_unitigs(), h.uoff.back(), h.n_kmers, h.k, &t); }
            catch (const std::exception& e) { return rtk_fail(strstr(e.what(), "occurs twice") ? RTK_ERR_FORMAT : RTK_ERR_DEVICE, std::string("rtk_graph_upload: ") + e.what()); }
            void* const p[6] = { t.ht, t.bf, t.bf1, t.hx, t.hxl, t.adj }; const uint64_t b[6] = { t.ht_bytes, t.bf_bytes, t.bf1_bytes, t.hx_bytes, t.hxl_bytes, t.adj_bytes };
            for (int j = 0; j < 6; ++j) { g->dbuf[built_here[j]] = p[j]; g->dbytes[built_here[j]] = b[j]; }
            g->info.table_slots = t.ht_slots; g->info.hbm_bytes = 0; for (int i = 0; i < rtk::RTK_N_BUFS; ++i) g->info.hbm_bytes += g->dbytes[i];
            g->table_seconds = t.seconds[3];
        }
#endif
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, e.what()); }
    g->device = device; g->info.device = device; g->on_device = true;
    graph_set_view(g);
    return RTK_OK;
}

extern "C" int rtk_n_devices(void) { return rtk_device_count(); }
#ifdef RTK_SIM
// (the simulator build has the entry point for ABI completeness only: k-mer counting on the device is csrc/hip/rtk_index.hip, its CPU form the index tool's plain path)
extern "C" int rtk_index_count_kmers(int, int, const char* const*, int, uint32_t, int, uint64_t**, uint64_t*) { return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_count_kmers: not part of the simulator build"); }
extern "C" int rtk_index_colour_begin(int, int, const char*, const uint64_t*, uint64_t, void**) { return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_colour_begin: not part of the simulator build"); }
extern "C" int rtk_index_colour_chunk(void*, const char*, uint64_t, const uint64_t*, const uint32_t*, uint32_t) { return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_colour_chunk: not part of the simulator build"); }
extern "C" int rtk_index_colour_end(void*, uint64_t**, uint64_t*, uint64_t**) { return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_colour_end: not part of the simulator build"); }
extern "C" int rtk_index_unitigs(int, int, const uint64_t*, uint64_t, char**, uint64_t**, uint64_t**, uint64_t*, uint64_t**, uint64_t*) { return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_index_unitigs: not part of the simulator build"); }
#endif
extern "C" int rtk_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (!free_bytes || !total_bytes) return rtk_fail(RTK_ERR_ARG, "rtk_device_memory: null argument");
    if (rtk_device_count() <= device || device < 0) return rtk_fail(RTK_ERR_NO_DEVICE, "rtk_device_memory: no such HIP device");
#ifdef RTK_SIM
    *free_bytes = 64ull << 30; *total_bytes = 64ull << 30;
#else
    try { rtk_set_device(device); size_t fr = 0, tot = 0; rtk_check(hipMemGetInfo(&fr, &tot), "hipMemGetInfo"); *free_bytes = fr; *total_bytes = tot; }
    catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, e.what()); }
#endif
    return RTK_OK;
}

// One more replica of a resident graph on another GPU of the same process: the flat buffers go device to device (xGMI peers), the host
// image is neither parsed again nor copied (reference counterpart: ONE graph shared by all worker threads, src/Ratatosk.cpp:618,727).
extern "C" int rtk_graph_clone_to_device(const rtk_graph* src, int device, rtk_graph** out) {
    if (!src || !out) return rtk_fail(RTK_ERR_ARG, "rtk_graph_clone_to_device: null argument");
    if (!src->on_device) return rtk_fail(RTK_ERR_ARG, "rtk_graph_clone_to_device: the source graph is not resident on a device");
    int rc = require_device(device); if (rc) return rc;
    std::unique_ptr<rtk_graph> g(new rtk_graph());
    try {
        rtk_set_device(device);
        for (int i = 0; i < rtk::RTK_N_BUFS; ++i) { g->dbuf[i] = rtk_dmalloc(src->dbytes[i]); g->dbytes[i] = src->dbytes[i]; rtk_d2d_peer(g->dbuf[i], device, src->dbuf[i], src->device, src->dbytes[i]); }
        rtk_dsync();
    } catch (const std::exception& e) { for (int i = 0; i < rtk::RTK_N_BUFS; ++i) rtk_dfree(g->dbuf[i]); return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_graph_clone_to_device: ") + e.what()); }
    g->info = src->info; g->info.device = device; g->device = device; g->on_device = true;
    graph_set_view(g.get());
    *out = g.release();
    return RTK_OK;
}

extern "C" int rtk_graph_buffer(rtk_graph* g, int idx, void** dev_ptr, uint64_t* bytes) {
    if (!g || idx < 0 || idx >= rtk::RTK_N_BUFS || !dev_ptr || !bytes) return rtk_fail(RTK_ERR_ARG, "rtk_graph_buffer: bad argument");
    *dev_ptr = g->dbuf[idx]; *bytes = g->dbytes[idx];
    return RTK_OK;
}

// A resident flat buffer of a graph moved into caller-owned HBM (a torch tensor that torch.distributed then broadcasts from): device to device, the
// library's own copy freed. For graphs whose tables were built by rtk_graph_upload -- their sizes are not known before. One buffer per call, so that a
// caller can allocate the destinations one at a time (a whole-genome graph does not fit twice).
extern "C" int rtk_graph_move_buffer(rtk_graph* g, int idx, void* dev_ptr, uint64_t bytes) {
    if (!g || idx < 0 || idx >= rtk::RTK_N_BUFS || !dev_ptr || !g->on_device) return rtk_fail(RTK_ERR_ARG, "rtk_graph_move_buffer: needs a resident graph, a buffer index and a destination");
    if (bytes < g->dbytes[idx]) return rtk_fail(RTK_ERR_ARG, "rtk_graph_move_buffer: the destination is smaller than the buffer");
    try {
        rtk_set_device(g->device);
        rtk_d2d_peer(dev_ptr, g->device, g->dbuf[idx], g->device, g->dbytes[idx]); rtk_dsync();
        if (g->owns_buffers && !(g->moved >> idx & 1u)) rtk_dfree(g->dbuf[idx]);
        g->dbuf[idx] = dev_ptr; g->moved |= 1u << idx;
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, std::string("rtk_graph_move_buffer: ") + e.what()); }
    graph_set_view(g);
    return RTK_OK;
}

// (tests, tools) a flat buffer of the HOST image: pointer into the graph's own memory, valid until rtk_graph_free
extern "C" int rtk_graph_host_buffer(const rtk_graph* g, int idx, const void** p, uint64_t* bytes) {
    if (!g || idx < 0 || idx >= rtk::RTK_N_BUFS || !p || !bytes || !g->has_host) return rtk_fail(RTK_ERR_ARG, "rtk_graph_host_buffer: needs a loaded graph");
    const rtk::FlatGraph& h = g->host;
    const void* src[rtk::RTK_N_BUFS] = { h.useq.data(), h.uoff.data(), h.adj.data(), h.flags.data(), h.kcov.data(), h.card.data(), h.loff.data(), h.gid.data(), h.goff.data(), h.col.data(), h.ht.data(), h.bf.data(), h.cycoff.data(), h.cyc.data(), h.bf1.data(), h.amb.data(), h.hx.data(), h.hxl.data(), h.hap.data() };
    const uint64_t b[rtk::RTK_N_BUFS] = { 8 * h.useq.size(), 8 * h.uoff.size(), 4 * h.adj.size(), 4 * h.flags.size(), 4 * h.kcov.size(), 4 * h.card.size(), 8 * h.loff.size(), 4 * h.gid.size(), 8 * h.goff.size(), 4 * h.col.size(), 8 * h.ht.size(), 8 * h.bf.size(), 8 * h.cycoff.size(), 8 * h.cyc.size(), 8 * h.bf1.size(), 8 * h.amb.size(), 8 * h.hx.size(), 8 * h.hxl.size(), 8 * h.hap.size() };
    *p = src[idx]; *bytes = b[idx];
    return RTK_OK;
}

// (tests, tools) a flat buffer of the resident graph copied to host memory
extern "C" int rtk_graph_download_buffer(const rtk_graph* g, int idx, void* host_dst, uint64_t bytes) {
    if (!g || idx < 0 || idx >= rtk::RTK_N_BUFS || !host_dst || !g->on_device || bytes > g->dbytes[idx]) return rtk_fail(RTK_ERR_ARG, "rtk_graph_download_buffer: bad argument");
    try { rtk_set_device(g->device); rtk_d2h(host_dst, g->dbuf[idx], bytes); } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, e.what()); }
    return RTK_OK;
}

extern "C" int rtk_graph_adopt_device(rtk_graph* g) {
    if (!g) return rtk_fail(RTK_ERR_ARG, "rtk_graph_adopt_device: null");
    for (int i = 0; i < rtk::RTK_N_BUFS; ++i) if (!g->dbuf[i]) return rtk_fail(RTK_ERR_ARG, "rtk_graph_adopt_device: buffers not allocated");
    g->on_device = true; graph_set_view(g);
    return RTK_OK;
}

extern "C" int rtk_graph_get_info(const rtk_graph* g, rtk_graph_info* info) { if (!g || !info) return rtk_fail(RTK_ERR_ARG, "rtk_graph_get_info: null"); *info = g->info; return RTK_OK; }

extern "C" long long rtk_graph_strip_annotations(rtk_graph* g) {
    if (!g) return rtk_fail(RTK_ERR_ARG, "rtk_graph_strip_annotations: null");
    if (!g->has_host || g->on_device) return rtk_fail(RTK_ERR_ARG, "rtk_graph_strip_annotations: call it on a loaded graph before rtk_graph_upload");
    long long n = 0;
    for (size_t u = 0; u < g->host.flags.size(); ++u) if (g->host.flags[u] & (RTK_F_SHORT_CYCLE | RTK_F_AMBIGUITY)) { g->host.flags[u] &= ~static_cast<uint32_t>(RTK_F_SHORT_CYCLE | RTK_F_AMBIGUITY); ++n; }
    g->host.amb.assign(g->host.flags.size() + 1, 0);
    return n;
}

// A graph is shared by its batches (buffer pool, scratch slots): it is destroyed when the caller has released it AND its last batch
// is gone, whichever comes last (callers with garbage collectors free the two in any order).
static void graph_release(rtk_graph* g) {
    if (g->refs.fetch_sub(1) != 1) return;
    if (g->owns_buffers) for (int i = 0; i < rtk::RTK_N_BUFS; ++i) if (!(g->moved >> i & 1u)) rtk_dfree(g->dbuf[i]);
    rtk_dfree(g->scratch[0]); rtk_dfree(g->scratch[1]); rtk_dfree(g->scratch[2]); g->pool_clear();
    delete g;
}
extern "C" void rtk_graph_free(rtk_graph* g) { if (g) graph_release(g); }

extern "C" int rtk_opts_default(const rtk_graph* g, rtk_opts* o) {
    if (!o) return rtk_fail(RTK_ERR_ARG, "rtk_opts_default: null");
    o->insert_sz = 500; o->min_cov_vertices = 2; o->max_len_weak_region1 = 1000;
    o->max_km_cov = 128; if (g && g->info.max_km_cov_top > 128) o->max_km_cov = g->info.max_km_cov_top; // src/Ratatosk.cpp:625
    o->weak_region_len_factor = 0.25; o->large_k_factor = 1.5; o->min_score = 0.0; o->max_qual = 40; o->out_qual = 1; o->min_confidence_snp_corr = 0.9;
    o->long_read_correct = 0; o->force_unres_snp_corr = 0; o->max_len_weak_region2 = 5000; o->struct_size = static_cast<uint32_t>(sizeof(rtk_opts));
    { const char* e = getenv("RTK_A2_XOR"); o->a2_exclusive = (e && !strcmp(e, "union")) ? 0 : ((e && !strcmp(e, "exclusive-ids")) ? 2 : 1); const char* e3 = getenv("RTK_A3_ORDER"); o->a3_strand_order = (e3 && !strcmp(e3, "strand")) ? 1 : 0;  const char* e4 = getenv("RTK_D1_ORDER"); o->d1_desc = (e4 && !strcmp(e4, "desc")) ? 1 : 0; } // [A2] switch, see rtk_opts
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------ per-wave scratch
// Slabs reserved ahead of any graph (rtk_reserve_scratch): the first hipMalloc of a tens-of-GB slab takes seconds, a caller can have
// it done while the graph files are still being parsed. A graph that needs a slab takes the smallest reserved one that is large enough.
struct ReservedSlab { int device; void* p; uint64_t bytes; };
static std::mutex g_reserved_lock;
static std::vector<ReservedSlab> g_reserved;

void* rtk_graph::phase_take(uint64_t bytes, uint64_t* got) {
    { std::lock_guard<std::mutex> h(pool_lock);
      int best = -1; // the smallest one that fits (the slabs of the phasing step and of the region stage of a second-pass ticket circulate in the same list)
      for (size_t i = 0; i < phase_free.size(); ++i) if (phase_free[i].second >= bytes && (best < 0 || phase_free[i].second < phase_free[static_cast<size_t>(best)].second)) best = static_cast<int>(i);
      if (best >= 0) { void* p = phase_free[static_cast<size_t>(best)].first; *got = phase_free[static_cast<size_t>(best)].second; phase_free.erase(phase_free.begin() + best); return p; } }
    { std::lock_guard<std::mutex> lk(g_reserved_lock); // reserved ahead (rtk_reserve_second_pass): the smallest one that fits, but not one several times too big (those are the first pass's)
      int best = -1;
      for (size_t i = 0; i < g_reserved.size(); ++i) if (g_reserved[i].device == device && g_reserved[i].bytes >= bytes && g_reserved[i].bytes <= 4 * bytes + (1ull << 30) && (best < 0 || g_reserved[i].bytes < g_reserved[static_cast<size_t>(best)].bytes)) best = static_cast<int>(i);
      if (best < 0) // nothing of a fitting size: any reserved slab that is large enough, before more memory is asked for next to the reservation
          for (size_t i = 0; i < g_reserved.size(); ++i) if (g_reserved[i].device == device && g_reserved[i].bytes >= bytes && (best < 0 || g_reserved[i].bytes < g_reserved[static_cast<size_t>(best)].bytes)) best = static_cast<int>(i);
      if (best >= 0) { void* p = g_reserved[static_cast<size_t>(best)].p; *got = g_reserved[static_cast<size_t>(best)].bytes; g_reserved.erase(g_reserved.begin() + best); return p; } }
    // nothing held back fits: new memory, in steps of 2 GB so that the slabs of tickets of slightly different sizes serve one another (the work area of a ticket follows
    // its longest reads; slabs that just miss the next ticket's size would pile up in the free list)
    bytes = (bytes + (2ull << 30) - 1) / (2ull << 30) * (2ull << 30);
    const auto t0 = std::chrono::steady_clock::now();
    void* p = rtk_dmalloc(bytes);
    if (getenv("RTK_TRACE")) fprintf(stderr, "[rtk trace] phase_take: %.1f GB of new device memory in %.1f ms\n", bytes / 1073741824.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    *got = bytes; return p;
}

static char* graph_scratch(rtk_graph* g, int slot, uint64_t bytes) { // grows monotonically; hipMalloc/hipFree of tens of GB per batch would dominate a step
    if (bytes > g->scratch_bytes_[slot]) {
        rtk_dfree(g->scratch[slot]); g->scratch[slot] = nullptr; g->scratch_bytes_[slot] = 0;
        {
            std::lock_guard<std::mutex> lk(g_reserved_lock);
            int best = -1;
            for (size_t i = 0; i < g_reserved.size(); ++i) if (g_reserved[i].device == g->device && g_reserved[i].bytes >= bytes && (best < 0 || g_reserved[i].bytes < g_reserved[static_cast<size_t>(best)].bytes)) best = static_cast<int>(i);
            if (best >= 0) { g->scratch[slot] = g_reserved[static_cast<size_t>(best)].p; g->scratch_bytes_[slot] = g_reserved[static_cast<size_t>(best)].bytes; g_reserved.erase(g_reserved.begin() + best); }
        }
        if (!g->scratch[slot]) { g->scratch[slot] = rtk_dmalloc(bytes); g->scratch_bytes_[slot] = bytes; }
    }
    return static_cast<char*>(g->scratch[slot]);
}

// ------------------------------------------------------------------------------------------------ K1: exact k-mer lookup
// dbg.searchSequence(s, exact) (reference: src/Graph.cpp:97 [A1]). One lane per k-mer window; windows are addressed
// by their base position in the concatenated read buffer. hits[b] = packed (unitig, dist, strand) or RTK_NO_HIT.
RTK_GLOBAL void k_lookup_exact(GraphView g, const char* seq, const uint64_t* roff, uint32_t n_reads, uint64_t n_bases, int grid, uint64_t* hits, uint64_t* hitmap, uint64_t* n_probes_out) {
    const uint64_t n_tiles = (n_bases + RTK_WAVE - 1) / RTK_WAVE;
    uint32_t probes = 0, slots = 0;
    for (uint64_t tile = static_cast<uint64_t>(RTK_BLOCK_ID); tile < n_tiles; tile += static_cast<uint64_t>(grid)) {
        const uint64_t b = tile * RTK_WAVE + static_cast<uint64_t>(rtk_lane());
        uint64_t h = RTK_NO_HIT;
        // owning read: largest r with roff[r] <= b. One scalar search for the tile's first base; the other lanes step forward from it
        // (a tile rarely spans more than one read boundary).
        const uint32_t lo0 = rtk_owner_read(roff, n_reads, tile * RTK_WAVE);
#ifndef RTK_SIM
        // k <= 32: the 2-bit codes of the tile's 64 + 64 characters as bit planes (one character of each half per lane, three ballots per half: low bit, high
        // bit, is-a-base); a lane's k-mer is k bits of each plane from its own position on, reversed (the first character sits in the high bits of a code)
        // and interleaved. Packing the k characters from text in every lane (rtk_km_from_text: four unaligned words, two 64-bit multiplies each) was ~45 % of
        // this kernel's issue slots.
        RtkKm fw_t; fw_t.hi = 0; fw_t.lo = 0; bool ok_t = false; const bool planes = g.k <= 32;
        if (planes) {
            const uint64_t a0 = tile * RTK_WAVE + static_cast<uint64_t>(rtk_lane()), a1 = a0 + RTK_WAVE;
            const unsigned char ca = a0 < n_bases + 64 ? static_cast<unsigned char>(seq[a0]) : 'N', cb = a1 < n_bases + 64 ? static_cast<unsigned char>(seq[a1]) : 'N'; // (the read buffer is padded by 64 bytes)
            const bool va = ca == 'A' || ca == 'C' || ca == 'G' || ca == 'T', vb = cb == 'A' || cb == 'C' || cb == 'G' || cb == 'T';
            const uint32_t b1a = (ca >> 1) & 1u, b2a = (ca >> 2) & 1u, b1b = (cb >> 1) & 1u, b2b = (cb >> 2) & 1u; // bits 1-2 of 'A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54: code = b2 : b1 ^ b2
            const uint64_t L0a = rtk_ballot((b1a ^ b2a) != 0), L1a = rtk_ballot(b2a != 0), Va = rtk_ballot(va), L0b = rtk_ballot((b1b ^ b2b) != 0), L1b = rtk_ballot(b2b != 0), Vb = rtk_ballot(vb);
            const int sh = rtk_lane(), k_ = g.k;
            const uint64_t km = k_ < 64 ? ((1ull << k_) - 1ull) : ~0ull;
            const uint64_t p0 = (sh ? ((L0a >> sh) | (L0b << (64 - sh))) : L0a) & km, p1 = (sh ? ((L1a >> sh) | (L1b << (64 - sh))) : L1a) & km, pv = (sh ? ((Va >> sh) | (Vb << (64 - sh))) : Va) & km;
            ok_t = pv == km;
            auto spread = [](uint64_t x) { x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x << 2)) & 0x3333333333333333ull; return (x | (x << 1)) & 0x5555555555555555ull; };
            fw_t.lo = (spread(rtk_brev64(p1) >> (64 - k_)) << 1) | spread(rtk_brev64(p0) >> (64 - k_));
        }
#endif
        if (b < n_bases) {
            uint32_t lo = lo0;
            while (lo + 1 < n_reads && roff[lo + 1] <= b) ++lo;
            if (b + static_cast<uint64_t>(g.k) <= roff[lo + 1]) {
                RtkKm fw; bool ok; // the read buffer is padded by 64 bytes
#ifndef RTK_SIM
                if (planes) { fw = fw_t; ok = ok_t; } else
#endif
                ok = rtk_km_from_text(reinterpret_cast<const unsigned char*>(seq) + b, g.k, &fw);
                if (ok) { uint32_t np; h = rtk_find_km(g, fw, &np); probes += 1; slots += np; }
            }
#if defined(RTK_SIM) || defined(RTK_NO_NT_STORES)
            hits[b] = h;
#else
            __builtin_nontemporal_store(h, hits + b); // 8 B per window streamed out once: kept from evicting the first-level filter (the whole L2 of an XCD for graphs above 5 M k-mers)
#endif
        }
        // presence bit of every window (bit b & 63 of word b >> 6): the per-read programs scan 64 windows per word instead of 64 hits
        if (hitmap) {
#ifdef RTK_SIM
            if (h != RTK_NO_HIT) __atomic_fetch_or(hitmap + (b >> 6), 1ull << (b & 63), __ATOMIC_RELAXED);
#else
            const uint64_t bal = rtk_ballot(h != RTK_NO_HIT);
            if (rtk_lane() == 0) hitmap[tile] = bal;
#endif
        }
    }
    if (n_probes_out) { // n_probes_out[0] += k-mer queries, n_probes_out[11] += 16-byte table slots visited (RTK_CNT_PROBES_EXACT -> RTK_CNT_SLOTS_EXACT)
        const int tot = rtk_wave_sum(static_cast<int>(probes)), tots = rtk_wave_sum(static_cast<int>(slots));
        if (rtk_lane() == 0) { if (tot) rtk_atomic_add(reinterpret_cast<unsigned long long*>(n_probes_out), static_cast<unsigned long long>(tot)); if (tots) rtk_atomic_add(reinterpret_cast<unsigned long long*>(n_probes_out) + 11, static_cast<unsigned long long>(tots)); }
    }
}

static int default_grid() {
#ifdef RTK_SIM
    return 64;
#else
    return 256 * 16; // 256 CUs x 16 single-wave workgroups
#endif
}

extern "C" int rtk_lookup_exact(rtk_graph* g, const char* seq, uint32_t len, int64_t* hits) {
    if (!g || !seq || !hits) return rtk_fail(RTK_ERR_ARG, "rtk_lookup_exact: null argument");
    if (!g->on_device) return rtk_fail(RTK_ERR_NO_DEVICE, "rtk_lookup_exact: graph is not resident on a device (call rtk_graph_upload)");
    try {
        rtk_set_device(g->device);
        std::string up(seq, len);
        for (size_t i = 0; i < up.size(); ++i) up[i] = static_cast<char>(toupper(static_cast<unsigned char>(up[i])));
        char* dseq = static_cast<char*>(rtk_dmalloc(len + 64));
        uint64_t* droff = static_cast<uint64_t*>(rtk_dmalloc(16));
        uint64_t* dhits = static_cast<uint64_t*>(rtk_dmalloc(8ull * (len + 1)));
        const uint64_t roff[2] = {0, len};
        rtk_h2d(dseq, up.data(), len); rtk_h2d(droff, roff, 16);
        rtk_launch(k_lookup_exact, default_grid(), 0, g->dview, static_cast<const char*>(dseq), static_cast<const uint64_t*>(droff), 1u, static_cast<uint64_t>(len), default_grid(), dhits, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr));
        rtk_dsync();
        std::vector<uint64_t> h(len + 1);
        rtk_d2h(h.data(), dhits, 8ull * len);
        const uint32_t nw = len >= static_cast<uint32_t>(g->info.k) ? len - g->info.k + 1 : 0;
        for (uint32_t i = 0; i < nw; ++i) hits[i] = (h[i] == RTK_NO_HIT) ? -1 : static_cast<int64_t>(h[i]);
        rtk_dfree(dseq); rtk_dfree(droff); rtk_dfree(dhits);
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, e.what()); }
    return RTK_OK;
}

// ------------------------------------------------------------------------------------------------ K6/K7: Myers batch
// one problem of the batch on this wave (shared with the multi-wave variant of the kernel, rtk_phase_long.hip)
RTK_FN void rtk_myers_batch_item(const MyersScratch& sc, const MyersProb& p, uint32_t i, const char* pool, int want_path, int use_iupac, int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs,
                                 uint8_t* moves_out, uint32_t* n_moves_out, uint32_t cap_moves, uint32_t* status) {
    *sc.overflow = 0;
    const char* q = pool + p.q_off; const char* t = pool + p.t_off;
    const MyersResult r = rtk_myers_distance(sc, q, static_cast<int>(p.qlen), t, static_cast<int>(p.tlen), p.k, p.mode, use_iupac != 0, cap_locs ? end_locs + static_cast<uint64_t>(i) * cap_locs : nullptr, static_cast<int>(cap_locs)); // cap_locs == 0: no list of end locations (the route the region program takes)
    dist[i] = r.dist; n_loc[i] = r.nloc;
    uint32_t nm = 0;
    if (want_path && r.dist >= 0 && p.qlen > 0 && p.tlen > 0) { // edlib.cpp:271-284 (zero-length inputs return before any path is built)
        if (p.k < 0 && p.mode != RTK_MODE_HW) { // the single-sweep route the region program takes for NW / SHW paths
            const MyersResult r2 = rtk_myers_path(sc, q, static_cast<int>(p.qlen), t, static_cast<int>(p.tlen), p.mode, use_iupac != 0, &nm);
            if (r2.dist != r.dist || r2.first != r.first) *sc.overflow = 3; // must agree with the distance pass
        } else
        rtk_myers_alignment(sc, q, static_cast<int>(p.qlen), t, r.first + 1, r.dist, use_iupac != 0, &nm);
        if (nm <= cap_moves) rtk_wcopy(moves_out + static_cast<uint64_t>(i) * cap_moves, sc.moves, nm);
    }
    n_moves_out[i] = nm;
    status[i] = *sc.overflow;
}

RTK_GLOBAL void k_myers_batch(const MyersProb* probs, uint32_t n, const char* pool, int want_path, int use_iupac, char* scratch, uint64_t scratch_stride, ScratchCfg cfg,
                              int grid, int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, uint8_t* moves_out, uint32_t* n_moves_out, uint32_t cap_moves, uint32_t* status, unsigned long long* prof) {
#ifdef RTK_SIM
    const MyersScratch sc = scratch_carve(scratch + static_cast<uint64_t>(RTK_BLOCK_ID) * scratch_stride, cfg);
#else
    __shared__ MyersScratch sc; // in LDS like the header it is part of in the region kernels (the alignment code assumes so: RTK_ASSUME_LDS)
    sc = scratch_carve(scratch + static_cast<uint64_t>(RTK_BLOCK_ID) * scratch_stride, cfg); // every lane stores the same words
    __syncthreads();
#endif
    for (uint32_t i = static_cast<uint32_t>(RTK_BLOCK_ID); i < n; i += static_cast<uint32_t>(grid))
        rtk_myers_batch_item(sc, probs[i], i, pool, want_path, use_iupac, dist, n_loc, end_locs, cap_locs, moves_out, n_moves_out, cap_moves, status);
    if (prof && rtk_lane() == 0) { rtk_atomic_add(prof + 0, sc.hb_total.get()); rtk_atomic_add(prof + 1, sc.hb_pass.get()); rtk_atomic_add(prof + 2, sc.hb_split.get()); rtk_atomic_add(prof + 3, sc.hb_leaf.get()); rtk_atomic_add(prof + 4, sc.walk_cycles.get()); }
}

#ifndef RTK_SIM
// multi-wave variant of k_myers_batch (rtk_phase_long.hip)
void rtk_launch_myers_batch_waves(int grid, int waves, const MyersProb* probs, uint32_t n, const char* pool, int want_path, int use_iupac, char* scratch, uint64_t scratch_stride, const ScratchCfg& cfg,
                                  int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, uint8_t* moves_out, uint32_t* n_moves_out, uint32_t cap_moves, uint32_t* status, unsigned long long* prof);
#endif
extern "C" int rtk_myers_batch(uint32_t n, const char* const* query, const uint32_t* qlen, const char* const* target, const uint32_t* tlen,
                               const int32_t* k, const int32_t* mode, int want_path, int use_iupac,
                               int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, char* cigar, uint32_t cap_cigar) {
    return rtk_myers_batch_waves(n, query, qlen, target, tlen, k, mode, want_path, use_iupac, dist, n_loc, end_locs, cap_locs, cigar, cap_cigar, 1);
}
extern "C" int rtk_myers_batch_waves(uint32_t n, const char* const* query, const uint32_t* qlen, const char* const* target, const uint32_t* tlen,
                                     const int32_t* k, const int32_t* mode, int want_path, int use_iupac,
                                     int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, char* cigar, uint32_t cap_cigar, int waves) {
    if (waves > 16) return rtk_fail(RTK_ERR_ARG, "rtk_myers_batch_waves: at most 16 waves per workgroup");
    if (!query || !qlen || !target || !tlen || !k || !mode || !dist || !n_loc || (!end_locs && cap_locs)) return rtk_fail(RTK_ERR_ARG, "rtk_myers_batch: null argument");
    if (rtk_device_count() <= 0) return rtk_fail(RTK_ERR_NO_DEVICE, "rtk_myers_batch: no HIP device visible (no CPU fallback)");
    if (n == 0) return RTK_OK;
    try {
        std::vector<MyersProb> probs(n);
        std::string pool;
        uint32_t max_q = 1, max_t = 1;
        for (uint32_t i = 0; i < n; ++i) {
            probs[i].q_off = pool.size(); pool.append(query[i], qlen[i]);
            probs[i].t_off = pool.size(); pool.append(target[i], tlen[i]);
            probs[i].qlen = qlen[i]; probs[i].tlen = tlen[i]; probs[i].k = k[i]; probs[i].mode = mode[i];
            if (mode[i] == RTK_MODE_HW && want_path) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_myers_batch: HW path alignment is not on the hot path");
            max_q = std::max(max_q, qlen[i]); max_t = std::max(max_t, tlen[i]);
        }
        ScratchCfg cfg;
        cfg.w_cap = (max_q + 63) / 64 + 1; cfg.t_cap = max_t + 64; cfg.r_cap = max_q + 64; cfg.mv_cap = max_q + max_t + 64;
        cfg.tb_cap_words = std::max<uint64_t>(4ull * 52429 + 64, 4ull * cfg.w_cap + 64);
        const int grid = static_cast<int>(std::min<uint32_t>(n, static_cast<uint32_t>(default_grid() / 4 > 0 ? default_grid() / 4 : 1)));
        const uint64_t stride = scratch_bytes(cfg) + (waves > 1 ? static_cast<uint64_t>((waves < RTK_LEAF_WAVES ? waves : RTK_LEAF_WAVES) - 1) * scratch_bytes(rtk_leaf_cfg()) : 0ull); // + the leaf-traceback areas of the helper waves
        const uint32_t cap_moves = want_path ? (max_q + max_t + 8) : 1;
        struct Held { std::vector<void*> v; void* get(uint64_t bytes) { v.push_back(nullptr); v.back() = rtk_dmalloc(bytes); return v.back(); } void release() { for (void* p : v) if (p) rtk_dfree(p); v.clear(); } ~Held() { release(); } } held; // (freed on every way out)
        char* dpool = static_cast<char*>(held.get(pool.size() + 64));
        MyersProb* dprobs = static_cast<MyersProb*>(held.get(sizeof(MyersProb) * n));
        char* dscr = static_cast<char*>(held.get(stride * grid));
        int32_t* ddist = static_cast<int32_t*>(held.get(4ull * n)); int32_t* dnloc = static_cast<int32_t*>(held.get(4ull * n));
        int32_t* dlocs = static_cast<int32_t*>(held.get(4ull * n * cap_locs + 8)); uint32_t* dst = static_cast<uint32_t*>(held.get(4ull * n));
        uint8_t* dmoves = static_cast<uint8_t*>(held.get(static_cast<uint64_t>(n) * cap_moves + 8)); uint32_t* dnm = static_cast<uint32_t*>(held.get(4ull * n));
        rtk_h2d(dpool, pool.data(), pool.size()); rtk_h2d(dprobs, probs.data(), sizeof(MyersProb) * n);
        unsigned long long* dprof = nullptr; // RTK_MYERS_PROF=1: cycle counters of the Hirschberg drivers (developer)
        if (getenv("RTK_MYERS_PROF")) { dprof = static_cast<unsigned long long*>(rtk_dmalloc(64)); const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; rtk_h2d(dprof, z, 64); }
#ifndef RTK_SIM
        if (waves > 1) rtk_launch_myers_batch_waves(grid, waves, static_cast<const MyersProb*>(dprobs), n, static_cast<const char*>(dpool), want_path, use_iupac, dscr, stride, cfg, ddist, dnloc, dlocs, cap_locs, dmoves, dnm, cap_moves, dst, dprof);
        else
#endif
        { RtkTimer tk; const bool timed = getenv("RTK_MYERS_TIME") != nullptr && waves <= 1; if (timed) tk.start(0);
          rtk_launch(k_myers_batch, grid, 0, static_cast<const MyersProb*>(dprobs), n, static_cast<const char*>(dpool), want_path, use_iupac, dscr, stride, cfg, grid, ddist, dnloc, dlocs, cap_locs, dmoves, dnm, cap_moves, dst, dprof);
          if (timed) { tk.stop(0); rtk_dsync(); fprintf(stderr, "[rtk myers time] one wave per problem: %u problems, %d waves, kernel %.3f ms\n", n, grid, tk.elapsed()); } }
        if (dprof) { rtk_dsync(); unsigned long long pc[8]; rtk_d2h(pc, dprof, 64); rtk_dfree(dprof);
            fprintf(stderr, "[rtk myers prof] waves %d: Hirschberg driver %.3g cycles (half passes %.3g, columns + split %.3g, leaf tracebacks %.3g of which walks %.3g)\n", waves, double(pc[0]), double(pc[1]), double(pc[2]), double(pc[3]), double(pc[4])); }
        rtk_dsync();
        std::vector<uint32_t> st(n), nm(n);
        rtk_d2h(dist, ddist, 4ull * n); rtk_d2h(n_loc, dnloc, 4ull * n); if (cap_locs) rtk_d2h(end_locs, dlocs, 4ull * n * cap_locs);
        rtk_d2h(st.data(), dst, 4ull * n); rtk_d2h(nm.data(), dnm, 4ull * n);
        int rc = RTK_OK;
        for (uint32_t i = 0; i < n; ++i) if (st[i]) rc = rtk_fail(RTK_ERR_DEVICE, "rtk_myers_batch: scratch capacity exceeded on device");
        if (want_path && cigar && rc == RTK_OK) {
            std::vector<uint8_t> mv(static_cast<size_t>(n) * cap_moves);
            rtk_d2h(mv.data(), dmoves, mv.size());
            static const char code[4] = {'M', 'I', 'D', 'M'};
            for (uint32_t i = 0; i < n; ++i) { // edlibAlignmentToCigar, EDLIB_CIGAR_STANDARD (edlib.cpp:298-347)
                std::string c;
                const uint8_t* a = &mv[static_cast<size_t>(i) * cap_moves];
                for (uint32_t x = 0; x < nm[i];) { uint32_t y = x; while (y < nm[i] && code[a[y]] == code[a[x]]) ++y; c += std::to_string(y - x); c.push_back(code[a[x]]); x = y; }
                if (c.size() + 1 > cap_cigar) { rc = rtk_fail(RTK_ERR_ARG, "rtk_myers_batch: cigar buffer too small"); break; }
                memcpy(cigar + static_cast<size_t>(i) * cap_cigar, c.c_str(), c.size() + 1);
            }
        }
        held.release();
        return rc;
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, e.what()); }
}

// ---- stage entry rtk_myers_batch_lanes: one problem per LANE (csrc/hip/rtk_myers_lane.h) ----
RTK_GLOBAL void k_myers_batch_lanes(const MyersProb* probs, uint32_t n, const char* pool, int want_path, int use_iupac, char* scratch, uint64_t stride, int grid, int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs,
                                    uint8_t* moves_out, uint32_t* n_moves_out, uint32_t cap_moves, uint32_t* status) {
#ifdef RTK_SIM
    static thread_local uint64_t lpeq[RTK_ML_PEQ_WORDS];
#else
    __shared__ uint64_t lpeq[RTK_ML_PEQ_WORDS]; // 20 KB per wave: the match vectors of its 64 problems
#endif
    uint64_t* const peq = lpeq + rtk_lane();
    char* const area = scratch + static_cast<uint64_t>(RTK_BLOCK_ID) * stride;
    int32_t* const cs = reinterpret_cast<int32_t*>(area) + rtk_lane();
    uint64_t* const tb = reinterpret_cast<uint64_t*>(area + rtk_ml_scratch_bytes()) + rtk_lane(); // (only there when want_path)
    for (uint64_t i0 = static_cast<uint64_t>(RTK_BLOCK_ID) * RTK_WAVE; i0 < n; i0 += static_cast<uint64_t>(grid) * RTK_WAVE) {
        const uint64_t i = i0 + static_cast<uint64_t>(rtk_lane());
        if (i >= n) continue;
        const MyersProb p = probs[i];
        const int m = static_cast<int>(p.qlen), tn = static_cast<int>(p.tlen);
        int32_t d = -1, nl = 0, first = -1; uint32_t nm = 0;
        uint32_t st = rtk_myers_lane(pool + p.q_off, m, pool + p.t_off, tn, p.k, p.mode, use_iupac != 0, peq, cs, &d, &nl, cap_locs ? end_locs + i * cap_locs : nullptr, static_cast<int>(cap_locs), &first);
        if (st == 0u && want_path && d >= 0 && m > 0 && tn > 0) { // edlib.cpp:271-284: the whole query against target[0 .. first end location]
            uint8_t* const mv = moves_out + i * cap_moves;
            if (first + 1 == 0) { if (static_cast<uint32_t>(m) > cap_moves) st = 1u; else { for (int x = 0; x < m; ++x) mv[x] = 1; nm = static_cast<uint32_t>(m); } } // (the padded block's position -1: nothing of the target)
            else st = rtk_myers_lane_path(m, pool + p.t_off, first + 1, peq, tb, mv, cap_moves, &nm);
        }
        status[i] = st; dist[i] = d; n_loc[i] = nl; n_moves_out[i] = nm;
    }
}

static thread_local uint64_t g_ml_routes[2] = {0, 0}; // problems of the calling thread's last rtk_myers_batch_lanes call: taken by the lane route, handed on to the wave route
extern "C" void rtk_myers_lanes_last_routes(uint64_t* lane_route, uint64_t* wave_route) { if (lane_route) *lane_route = g_ml_routes[0]; if (wave_route) *wave_route = g_ml_routes[1]; }
extern "C" int rtk_myers_batch_lanes(uint32_t n, const char* const* query, const uint32_t* qlen, const char* const* target, const uint32_t* tlen,
                                     const int32_t* k, const int32_t* mode, int want_path, int use_iupac,
                                     int32_t* dist, int32_t* n_loc, int32_t* end_locs, uint32_t cap_locs, char* cigar, uint32_t cap_cigar) {
    if (!query || !qlen || !target || !tlen || !k || !mode || !dist || !n_loc || (!end_locs && cap_locs)) return rtk_fail(RTK_ERR_ARG, "rtk_myers_batch_lanes: null argument");
    if (rtk_device_count() <= 0) return rtk_fail(RTK_ERR_NO_DEVICE, "rtk_myers_batch_lanes: no HIP device visible (no CPU fallback)");
    g_ml_routes[0] = 0; g_ml_routes[1] = 0;
    if (n == 0) return RTK_OK;
    try {
        std::vector<MyersProb> probs(n);
        std::string pool;
        uint32_t max_q = 1, max_t = 1;
        for (uint32_t i = 0; i < n; ++i) {
            probs[i].q_off = pool.size(); pool.append(query[i], qlen[i]);
            probs[i].t_off = pool.size(); pool.append(target[i], tlen[i]);
            probs[i].qlen = qlen[i]; probs[i].tlen = tlen[i]; probs[i].k = k[i]; probs[i].mode = mode[i];
            if (mode[i] == RTK_MODE_HW && want_path) return rtk_fail(RTK_ERR_UNSUPPORTED, "rtk_myers_batch_lanes: HW path alignment is not on the hot path");
            max_q = std::max(max_q, qlen[i]); max_t = std::max(max_t, tlen[i]);
        }
        // the path table holds the word-columns the biggest problem of the call can ask for (words x (first end location + 1 <= tlen), never above the route's limit)
        uint64_t tb_cols = 1;
        if (want_path) for (uint32_t i = 0; i < n; ++i) tb_cols = std::max(tb_cols, std::min<uint64_t>(static_cast<uint64_t>((qlen[i] + 63u) >> 6) * tlen[i], RTK_ML_TB_WORDCOLS));
        const uint64_t stride = rtk_ml_scratch_bytes() + (want_path ? static_cast<uint64_t>(RTK_WAVE) * 32ull * tb_cols : 0ull);
        const int grid = static_cast<int>(std::min<uint64_t>((static_cast<uint64_t>(n) + RTK_WAVE - 1) / RTK_WAVE, static_cast<uint64_t>(want_path ? std::min(default_grid(), 512) : default_grid())));
        const uint32_t cap_moves = want_path ? (std::min<uint32_t>(max_q, 64u * RTK_ML_MAXW) + std::min<uint32_t>(max_t, RTK_ML_MAXN) + 8) : 1;
        struct Held { std::vector<void*> v; void* get(uint64_t bytes) { v.push_back(nullptr); v.back() = rtk_dmalloc(bytes); return v.back(); } void release() { for (void* p : v) if (p) rtk_dfree(p); v.clear(); } ~Held() { release(); } } held; // (freed on every way out)
        char* dpool = static_cast<char*>(held.get(pool.size() + 64));
        MyersProb* dprobs = static_cast<MyersProb*>(held.get(sizeof(MyersProb) * n));
        char* dscr = static_cast<char*>(held.get(stride * grid));
        int32_t* ddist = static_cast<int32_t*>(held.get(4ull * n)); int32_t* dnloc = static_cast<int32_t*>(held.get(4ull * n));
        int32_t* dlocs = static_cast<int32_t*>(held.get(4ull * n * cap_locs + 8)); uint32_t* dst = static_cast<uint32_t*>(held.get(4ull * n));
        uint8_t* dmoves = static_cast<uint8_t*>(held.get(static_cast<uint64_t>(n) * cap_moves + 8)); uint32_t* dnm = static_cast<uint32_t*>(held.get(4ull * n));
        rtk_h2d(dpool, pool.data(), pool.size()); rtk_h2d(dprobs, probs.data(), sizeof(MyersProb) * n);
        { RtkTimer tk; const bool timed = getenv("RTK_MYERS_TIME") != nullptr; if (timed) tk.start(0);
          rtk_launch(k_myers_batch_lanes, grid, 0, static_cast<const MyersProb*>(dprobs), n, static_cast<const char*>(dpool), want_path, use_iupac, dscr, stride, grid, ddist, dnloc, dlocs, cap_locs, dmoves, dnm, cap_moves, dst);
          if (timed) { tk.stop(0); rtk_dsync(); fprintf(stderr, "[rtk myers time] one lane per problem%s: %u problems, %d waves, kernel %.3f ms\n", want_path ? " (with paths)" : "", n, grid, tk.elapsed()); } }
        rtk_dsync();
        std::vector<uint32_t> st(n), nm(n);
        rtk_d2h(dist, ddist, 4ull * n); rtk_d2h(n_loc, dnloc, 4ull * n); if (cap_locs) rtk_d2h(end_locs, dlocs, 4ull * n * cap_locs); rtk_d2h(st.data(), dst, 4ull * n); rtk_d2h(nm.data(), dnm, 4ull * n);
        int rc = RTK_OK;
        if (want_path && cigar) {
            std::vector<uint8_t> mv(static_cast<size_t>(n) * cap_moves);
            rtk_d2h(mv.data(), dmoves, mv.size());
            static const char code[4] = {'M', 'I', 'D', 'M'};
            for (uint32_t i = 0; i < n && rc == RTK_OK; ++i) { // edlibAlignmentToCigar, EDLIB_CIGAR_STANDARD (edlib.cpp:298-347)
                if (st[i]) continue;
                std::string c;
                const uint8_t* a = &mv[static_cast<size_t>(i) * cap_moves];
                for (uint32_t x = 0; x < nm[i];) { uint32_t y = x; while (y < nm[i] && code[a[y]] == code[a[x]]) ++y; c += std::to_string(y - x); c.push_back(code[a[x]]); x = y; }
                if (c.size() + 1 > cap_cigar) { rc = rtk_fail(RTK_ERR_ARG, "rtk_myers_batch_lanes: cigar buffer too small"); break; }
                memcpy(cigar + static_cast<size_t>(i) * cap_cigar, c.c_str(), c.size() + 1);
            }
        }
        held.release();
        if (rc != RTK_OK) return rc;
        // the problems that are not for this route (query above 512 characters, target above 2048 or with a character outside ACGTN, a path table above 4096 word-columns): one wave each
        std::vector<uint32_t> rest; for (uint32_t i = 0; i < n; ++i) if (st[i]) rest.push_back(i);
        g_ml_routes[0] = n - rest.size(); g_ml_routes[1] = rest.size();
        if (getenv("RTK_MYERS_TIME")) fprintf(stderr, "[rtk myers time] %zu of %u problems handed on to the wave route\n", rest.size(), n);
        if (!rest.empty()) {
            const uint32_t nr = static_cast<uint32_t>(rest.size());
            std::vector<const char*> q2(nr), t2(nr); std::vector<uint32_t> ql2(nr), tl2(nr); std::vector<int32_t> k2(nr), m2(nr), d2(nr), nl2(nr), loc2(static_cast<size_t>(nr) * cap_locs + 1);
            std::vector<char> cg2(want_path && cigar ? static_cast<size_t>(nr) * cap_cigar : 1);
            for (uint32_t x = 0; x < nr; ++x) { const uint32_t i = rest[x]; q2[x] = query[i]; t2[x] = target[i]; ql2[x] = qlen[i]; tl2[x] = tlen[i]; k2[x] = k[i]; m2[x] = mode[i]; }
            rc = rtk_myers_batch(nr, q2.data(), ql2.data(), t2.data(), tl2.data(), k2.data(), m2.data(), want_path, use_iupac, d2.data(), nl2.data(), loc2.data(), cap_locs, want_path && cigar ? cg2.data() : nullptr, cap_cigar);
            if (rc != RTK_OK) return rc;
            for (uint32_t x = 0; x < nr; ++x) { const uint32_t i = rest[x]; dist[i] = d2[x]; n_loc[i] = nl2[x]; for (uint32_t y = 0; y < cap_locs; ++y) end_locs[static_cast<size_t>(i) * cap_locs + y] = loc2[static_cast<size_t>(x) * cap_locs + y];
                if (want_path && cigar) memcpy(cigar + static_cast<size_t>(i) * cap_cigar, &cg2[static_cast<size_t>(x) * cap_cigar], cap_cigar); }
        }
        return RTK_OK;
    } catch (const std::exception& e) { return rtk_fail(RTK_ERR_DEVICE, e.what()); }
}

#include "rtk_pipeline.inc"
