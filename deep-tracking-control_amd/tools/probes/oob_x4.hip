// Probe: how does gfx950 range-check a raw buffer_load_dwordx4 that straddles num_records -- per dword or all-or-nothing?
// (and is a 4-byte aligned, not 16-byte aligned, dwordx4 load legal?)   hipcc --offload-arch=gfx950 oob_x4.hip -o oob_x4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int num_bytes) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, num_bytes, 0x00020000);
    const int off = threadIdx.x * 4;             // lane l loads floats [l, l+4)
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
    // same with the offset split into voffset + soffset (soffset = 8 bytes)
    f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 8, 0));
    for (int e = 0; e < 4; ++e) out[256 + threadIdx.x * 4 + e] = w[e];
}
int main() {
    float h[64], *d, *o, ho[512];
    for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    hipMalloc(&d, 256); hipMalloc(&o, 2048);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 16>>>(d, o, 10 * 4);                  // num_records = 10 floats
    hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 12; ++l) printf("lane %2d (floats %2d..%2d, records 10): %6.0f %6.0f %6.0f %6.0f   | +soffset 8B: %6.0f %6.0f %6.0f %6.0f\n", l, l, l + 3,
                                        ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3], ho[256 + l * 4], ho[256 + l * 4 + 1], ho[256 + l * 4 + 2], ho[256 + l * 4 + 3]);
    return 0;
}
