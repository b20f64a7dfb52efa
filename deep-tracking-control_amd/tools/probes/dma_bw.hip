// Probe (round 4): how many bytes per clock does ONE CU get into LDS through LDS-DMA (buffer_load_dwordx4 ... lds) in the access
// pattern of the split GEMM kernels -- 3 workgroups of 4 waves per CU, every wave 6 pieces of 1 KiB per stage, one barrier per stage --
// from an L2-resident image (the weight side: 1.5 MB re-read by every workgroup) and from a 37 MB image streamed once per XCD (the X
// side)?  DEPTH = stages in flight (1: wait for everything at each barrier, as the kernels do; 2: one stage of slack).
//   hipcc --offload-arch=gfx950 -O3 dma_bw.hip -o dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int u32;
template <int DEPTH, int PIECES>
__global__ __launch_bounds__(256, 3) void k(const char* w, long long wbytes, const char* x, long long xbytes, int stages, int col_tiles, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char buf0[12288];
    __shared__ __attribute__((aligned(16))) char buf1[12288];
    __shared__ __attribute__((aligned(16))) char buf2[12288];
    __shared__ __attribute__((aligned(16))) char buf3[12288];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, local = j / col_tiles, tc = j - local * col_tiles, tr = xcd + 8 * local;
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, (int)wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x), 0, (int)xbytes, 0x00020000);
    u32 xo = (u32)(tr * stages) * 12288u, wo = (u32)(tc * stages) * 12288u;
    for (int s = 0; s < stages; ++s) {
        char* xb = (s & 1) ? buf1 : buf0;
        char* wb = (s & 1) ? buf3 : buf2;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (PIECES > p) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(xb + p * 4096 + wave * 1024), 16, tid * 16, xo + p * 4096, 0, 0);
            if (PIECES > 3 + p) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(wb + p * 4096 + wave * 1024), 16, tid * 16, wo + p * 4096, 0, 0);
        }
        xo += 12288u;
        wo += 12288u;
        if (DEPTH == 1) __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) (lgkmcnt / expcnt untouched)
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES));
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = *(volatile unsigned*)buf0 + *(volatile unsigned*)buf2;
}
template <int DEPTH, int PIECES>
float run(const char* w, long long wb, const char* x, long long xb, int stages, unsigned* sink, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<DEPTH, PIECES><<<768, 256>>>(w, wb, x, xb, stages, 4, sink);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k<DEPTH, PIECES><<<768, 256>>>(w, wb, x, xb, stages, 4, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    const int stages = 32;
    const long long wb = 4ll * stages * 12288, xb = 192ll * stages * 12288;
    char *w, *x;
    unsigned* sink;
    hipMalloc(&w, wb);
    hipMalloc(&x, xb);
    hipMalloc(&sink, 4096);
    hipMemset(w, 1, wb);
    hipMemset(x, 1, xb);
    const double per_wg = (double)stages * 12288;
    struct { const char* name; float us; double bytes; } r[] = {
        {"X + W (6 pieces / wave / stage), wait all", run<1, 6>(w, wb, x, xb, stages, sink, 20), 2 * per_wg * 768},
        {"X + W, one stage of slack", run<2, 6>(w, wb, x, xb, stages, sink, 20), 2 * per_wg * 768},
        {"X only (3 pieces), wait all", run<1, 3>(w, wb, x, xb, stages, sink, 20), per_wg * 768},
        {"X only, one stage of slack", run<2, 3>(w, wb, x, xb, stages, sink, 20), per_wg * 768},
    };
    for (auto& e : r)
        printf("%-46s %7.1f us   %6.2f TB/s into LDS   %5.1f B/clk/CU at 2.4 GHz\n", e.name, e.us, e.bytes / e.us / 1e6, e.bytes / (e.us * 1e-6) / 256 / 2.4e9);
    return 0;
}
