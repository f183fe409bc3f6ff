// Device memory / launch shim: HIP on gfx950; plain host memory and a thread pool for the RTK_SIM developer simulator.
#ifndef RTK_MEM_H
#define RTK_MEM_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <stdexcept>
#include <string>

#ifdef RTK_SIM

#include <atomic>
#include <thread>
#include <vector>

#define RTK_GLOBAL inline
extern thread_local int rtk_sim_block_id;
#define RTK_BLOCK_ID rtk_sim_block_id
typedef int rtk_stream_t;

inline void* rtk_dmalloc(uint64_t bytes) { void* p = calloc(bytes ? bytes : 1, 1); if (!p) throw std::runtime_error("sim: out of memory"); return p; }
inline void rtk_dfree(void* p) { free(p); }
inline void rtk_h2d(void* d, const void* h, uint64_t n) { if (n) memcpy(d, h, n); }
inline void rtk_d2h(void* h, const void* d, uint64_t n) { if (n) memcpy(h, d, n); }
inline void rtk_dzero(void* d, uint64_t n) { if (n) memset(d, 0, n); }
inline void rtk_dsync() {}
inline rtk_stream_t rtk_stream_create() { return 0; }
inline void rtk_stream_destroy(rtk_stream_t) {}
inline rtk_stream_t rtk_stream_create_high() { return 0; }
inline int rtk_cu_split(bool*) { return 0; }
inline rtk_stream_t rtk_stream_create_masked(bool) { return 0; }
inline void rtk_ssync(rtk_stream_t) {}
typedef int rtk_event_t;
inline rtk_event_t rtk_event_create() { return 0; }
inline void rtk_event_destroy(rtk_event_t) {}
inline void rtk_event_record(rtk_event_t, rtk_stream_t) {}
inline void rtk_stream_wait(rtk_stream_t, rtk_event_t) {}
inline void rtk_d2h_s(void* h, const void* d, uint64_t n, rtk_stream_t) { if (n) memcpy(h, d, n); }
inline void rtk_dzero_s(void* d, uint64_t n, rtk_stream_t) { if (n) memset(d, 0, n); }
inline void rtk_dfill_s(void* d, int c, uint64_t n, rtk_stream_t) { if (n) memset(d, c, n); }
inline int rtk_device_count() { const char* e = getenv("RTK_SIM_DEVICES"); const int n = e ? atoi(e) : 1; return n > 0 ? n : 1; } // pretend GPUs: the multi-GPU host plumbing runs on CPU
inline void rtk_set_device(int) {}
inline void rtk_d2d_peer(void* d, int, const void* s, int, uint64_t n) { if (n) memcpy(d, s, n); }
inline void* rtk_hmalloc_pinned(uint64_t bytes) { void* p = malloc(bytes ? bytes : 1); if (!p) throw std::runtime_error("sim: out of memory"); return p; }
inline void rtk_hfree_pinned(void* p) { free(p); }
inline void rtk_h2d_s(void* d, const void* h, uint64_t n, rtk_stream_t) { if (n) memcpy(d, h, n); }
inline void rtk_check_async_d2h(void* h, const void* d, uint64_t n, rtk_stream_t) { if (n) memcpy(h, d, n); }

template <class F, class... A>
inline void rtk_launch(F f, int grid, rtk_stream_t, A... a) {
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 1; if (nt > 16) nt = 16;
    if (static_cast<int>(nt) > grid) nt = static_cast<unsigned>(grid > 0 ? grid : 1);
    std::atomic<int> next(0);
    auto work = [&]() { while (true) { const int b = next.fetch_add(1); if (b >= grid) break; rtk_sim_block_id = b; f(a...); } };
    if (nt <= 1) { work(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
    for (size_t t = 0; t < th.size(); ++t) th[t].join();
}

struct RtkTimer { double ms; void start(rtk_stream_t) { ms = 0; } void stop(rtk_stream_t) {} double elapsed() { return 0.0; } };

#else

#include <hip/hip_runtime.h>

#define RTK_GLOBAL __global__
#define RTK_BLOCK_ID (static_cast<int>(blockIdx.x))
typedef hipStream_t rtk_stream_t;

inline void rtk_check(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
inline void* rtk_dmalloc(uint64_t bytes) { void* p = nullptr; rtk_check(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc"); return p; }
inline void rtk_dfree(void* p) { if (p) (void)hipFree(p); }
inline void rtk_h2d(void* d, const void* h, uint64_t n) { if (n) rtk_check(hipMemcpy(d, h, n, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
inline void rtk_d2h(void* h, const void* d, uint64_t n) { if (n) rtk_check(hipMemcpy(h, d, n, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
inline void rtk_dzero(void* d, uint64_t n) { if (n) rtk_check(hipMemset(d, 0, n), "hipMemset"); }
inline void rtk_dsync() { rtk_check(hipDeviceSynchronize(), "hipDeviceSynchronize"); }
// every batch owns a non-blocking stream: its kernels, clears, read-backs and timers are ordered on it and only it is waited for,
// so the stages of different batches overlap on the device
inline rtk_stream_t rtk_stream_create() { hipStream_t s = nullptr; rtk_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"); return s; }
inline void rtk_stream_destroy(rtk_stream_t s) { if (s) (void)hipStreamDestroy(s); }
// CU partition (RTK_CU_SPLIT=<n>[,i]: developer knob of round 6, profiles/r06_cu_mask_probe.txt): the seed stage of a batch on a stream whose kernels only get n of the
// device's compute units, its region stage on a stream that gets the others -- so that the seed kernels of step s + 1 find wave slots while the persistent region
// kernel of step s holds every slot of its own CUs. `,i`: the n CUs are every (total / n)-th bit of the mask (spread over the XCDs) instead of the lowest n bits.
inline int rtk_cu_split(bool* interleaved) { static const int v = [] { const char* e = getenv("RTK_CU_SPLIT"); return e ? atoi(e) : 0; }(); static const bool il = [] { const char* e = getenv("RTK_CU_SPLIT"); return e && strstr(e, ",i"); }(); if (interleaved) *interleaved = il; return v; }
inline rtk_stream_t rtk_stream_create_masked(bool seed_side) {
    bool il = false; const int n = rtk_cu_split(&il);
    int dev = 0, cus = 0; if (n <= 0 || hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= n || cus > 1024) return rtk_stream_create();
    uint32_t mask[32]; for (int i = 0; i < 32; ++i) mask[i] = 0;
    for (int c = 0; c < cus; ++c) { const bool seed_cu = il ? ((c % (cus / n)) == 0 && (c / (cus / n)) < n) : (c < n); if (seed_cu == seed_side) mask[c >> 5] |= 1u << (c & 31); }
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, static_cast<uint32_t>((cus + 31) / 32), mask) != hipSuccess) { (void)hipGetLastError(); return rtk_stream_create(); }
    return s;
}
// a stream whose kernels are dispatched ahead of those of the default-priority streams (the lane kernel of the region stage: few waves, each one long dependent chain)
inline rtk_stream_t rtk_stream_create_high() { int lo = 0, hi = 0; hipStream_t s = nullptr; if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return rtk_stream_create(); rtk_check(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi), "hipStreamCreateWithPriority"); return s; }
inline void rtk_ssync(rtk_stream_t s) { rtk_check(hipStreamSynchronize(s), "hipStreamSynchronize"); }
// ordering between two streams of a batch (the lane kernel of the region stage runs beside the wave kernel)
typedef hipEvent_t rtk_event_t;
inline rtk_event_t rtk_event_create() { hipEvent_t e = nullptr; rtk_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); return e; }
inline void rtk_event_destroy(rtk_event_t e) { if (e) (void)hipEventDestroy(e); }
inline void rtk_event_record(rtk_event_t e, rtk_stream_t s) { rtk_check(hipEventRecord(e, s), "hipEventRecord"); }
inline void rtk_stream_wait(rtk_stream_t s, rtk_event_t e) { rtk_check(hipStreamWaitEvent(s, e, 0), "hipStreamWaitEvent"); }
inline void rtk_d2h_s(void* h, const void* d, uint64_t n, rtk_stream_t s) { if (n) { rtk_check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "hipMemcpyAsync D2H"); rtk_ssync(s); } }
inline void rtk_dzero_s(void* d, uint64_t n, rtk_stream_t s) { if (n) rtk_check(hipMemsetAsync(d, 0, n, s), "hipMemsetAsync"); }
inline void rtk_dfill_s(void* d, int c, uint64_t n, rtk_stream_t s) { if (n) rtk_check(hipMemsetAsync(d, c, n, s), "hipMemsetAsync"); }
inline int rtk_device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
inline void rtk_set_device(int d) { rtk_check(hipSetDevice(d), "hipSetDevice"); }
// GPU -> GPU copy of a flat graph buffer (xGMI when the two devices are peers; the runtime stages through the host otherwise)
inline void rtk_d2d_peer(void* d, int ddev, const void* s, int sdev, uint64_t n) { if (n) rtk_check(hipMemcpyPeer(d, ddev, s, sdev, n), "hipMemcpyPeer"); }
// pinned host staging memory: H2D / D2H copies from it run at PCIe speed and asynchronously on the batch's stream
inline void* rtk_hmalloc_pinned(uint64_t bytes) { void* p = nullptr; rtk_check(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault), "hipHostMalloc"); return p; }
inline void rtk_hfree_pinned(void* p) { if (p) (void)hipHostFree(p); }
inline void rtk_h2d_s(void* d, const void* h, uint64_t n, rtk_stream_t s) { if (n) rtk_check(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s), "hipMemcpyAsync H2D"); }
inline void rtk_check_async_d2h(void* h, const void* d, uint64_t n, rtk_stream_t s) { if (n) rtk_check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "hipMemcpyAsync D2H"); } // completed by the next rtk_d2h_s / rtk_ssync

template <class F, class... A>
inline void rtk_launch(F f, int grid, rtk_stream_t s, A... a) {
    hipLaunchKernelGGL(f, dim3(static_cast<unsigned>(grid)), dim3(64), 0, s, a...);
    rtk_check(hipGetLastError(), "kernel launch");
}

struct RtkTimer { // HIP events on the stream the kernels are launched on
    hipEvent_t a, b; bool ok;
    RtkTimer() : ok(false) { if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) ok = true; }
    ~RtkTimer() { if (ok) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
    void start(rtk_stream_t s) { if (ok) (void)hipEventRecord(a, s); }
    void stop(rtk_stream_t s) { if (ok) (void)hipEventRecord(b, s); }
    double elapsed() { float ms = 0; if (ok) { (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); } return ms; }
};

#endif

#endif
