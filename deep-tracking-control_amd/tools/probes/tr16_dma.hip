// Probe (round 4): (1) which element does lane l, slot j of ds_read_b64_tr_b16 return when lane l supplies the address of the
// l-th 8-byte chunk of a region (= what the weight-gradient kernel's fragment reads rely on); (2) what does an LDS-DMA
// (buffer_load_dwordx4 ... lds) write for a lane whose offset is out of the descriptor's range (zeros, or nothing)?
//   hipcc --offload-arch=gfx950 tr16_dma.hip -o tr16_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned short u16;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const u16* src, int src_bytes, u16* out_tr, unsigned* out_dma) {
    __shared__ __attribute__((aligned(16))) u16 buf[1024];
    __shared__ __attribute__((aligned(16))) unsigned dma[256];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) buf[i] = (u16)i;          // element i holds its own index
    for (int i = l; i < 256; i += 64) dma[i] = 0xdeadbeefu;
    __syncthreads();
    // (1) lane l points at chunk l (4 consecutive u16 at element 4 l)
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(buf + 4 * l));
    for (int j = 0; j < 4; ++j) out_tr[l * 4 + j] = (u16)v[j];
    // (2) lanes 0..31 in range, lanes 32..63 far out of range
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(src), 0, src_bytes, 0x00020000);
    const unsigned off = l < 32 ? l * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dma, 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = l; i < 256; i += 64) out_dma[i] = dma[i];
}
int main() {
    u16 h[1024], *d, *o, ho[256];
    unsigned *od, hd[256];
    for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
    hipMalloc(&d, 2048); hipMalloc(&o, 512); hipMalloc(&od, 1024);
    hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, 512, o, od);                  // descriptor covers 512 bytes = lanes 0..31 x 16 B
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    hipMemcpy(hd, od, 1024, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16: lane -> the 4 element indices it received (lane l supplied chunk l = elements 4l..4l+3)\n");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3], (l & 3) == 3 ? "\n" : "  |");
    printf("LDS-DMA: dwords written per lane (in-range lanes 0..31, out-of-range lanes 32..63; 0xdeadbeef = untouched)\n");
    for (int l = 0; l < 64; l += 8) printf("  lane %2d: %08x %08x %08x %08x\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
    return 0;
}
