// VALU issue rate per wave as a function of the waves resident on one SIMD (one workgroup of 64 * w threads: w = 4 -> one wave per SIMD,
// 8 -> two, 16 -> four): dependent chains, and a Myers-like step loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }
__global__ void k_chain(unsigned* out, unsigned long long* t, unsigned a) {
    unsigned x = threadIdx.x + a;
    __syncthreads();
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a)); }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x; if ((threadIdx.x & 63) == 0) t[threadIdx.x >> 6] = t1 - t0;
}
__global__ void k_chain_salu(unsigned* out, unsigned long long* t, unsigned a) {
    unsigned x = __builtin_amdgcn_readfirstlane(a);
    __syncthreads();
    unsigned long long t0 = clk();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(a)); }
    }
    unsigned long long t1 = clk();
    out[threadIdx.x] = x; if ((threadIdx.x & 63) == 0) t[threadIdx.x >> 6] = t1 - t0;
}
int main() {
    unsigned* out; unsigned long long* t; hipMalloc(&out, 4096 * 4); hipMalloc(&t, 64 * 8);
    unsigned long long ht[16];
    for (int w : {1, 4, 8, 12, 16}) {
        hipLaunchKernelGGL(k_chain, 1, 64 * w, 0, 0, out, t, 3u); hipDeviceSynchronize(); hipMemcpy(ht, t, 8 * w, hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < w; ++i) mx = ht[i] > mx ? ht[i] : mx;
        printf("VALU chain, %2d waves in the workgroup (%d per SIMD): %.2f cycles per instruction per wave\n", w, (w + 3) / 4, mx / (N_IT * 16.0));
        hipLaunchKernelGGL(k_chain_salu, 1, 64 * w, 0, 0, out, t, 3u); hipDeviceSynchronize(); hipMemcpy(ht, t, 8 * w, hipMemcpyDeviceToHost);
        mx = 0; for (int i = 0; i < w; ++i) mx = ht[i] > mx ? ht[i] : mx;
        printf("SALU chain, %2d waves in the workgroup (%d per SIMD): %.2f cycles per instruction per wave\n", w, (w + 3) / 4, mx / (N_IT * 16.0));
    }
    return 0;
}
