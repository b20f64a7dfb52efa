// Shared device pieces of the split-precision (bf16 x 3) kernels: csrc/gemm_s3.hip (forward / data gradient) and
// csrc/wgrad_s3.hip (weight gradients).  See the header comment of gemm_s3.hip.
#pragma once
#include "gemm_core.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (bf16(lo), bf16(hi)), round to nearest even: one v_cvt_pk_bf16_f32 (a builtin conversion, not inline asm: the instruction
// scheduler can see and place it)
__device__ __forceinline__ u32 cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(u32, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf_lo(u32 p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(u32 p) { return __uint_as_float(p & 0xffff0000u); }

struct Split3 {
    u32x2 p[3];
};
// four consecutive k values -> their three bf16 planes (8 bytes each)
__device__ __forceinline__ Split3 split3(f32x4 v) {
    Split3 s;
    s.p[0] = u32x2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
    float r0 = v[0] - bf_lo(s.p[0].x), r1 = v[1] - bf_hi(s.p[0].x), r2 = v[2] - bf_lo(s.p[0].y), r3 = v[3] - bf_hi(s.p[0].y);
    s.p[1] = u32x2{cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3)};
    r0 -= bf_lo(s.p[1].x);
    r1 -= bf_hi(s.p[1].x);
    r2 -= bf_lo(s.p[1].y);
    r3 -= bf_hi(s.p[1].y);
    s.p[2] = u32x2{cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3)};
    return s;
}

// LDS plane: 128 (or BN) rows x 16 k bf16 = 32 bytes per row = four 8-byte slots.  Row r keeps its k half hh (8 bf16 = 16
// bytes) at half position hh ^ ((r >> 3) & 1): the 16 lanes of a ds_read_b128 group (rows i, half h) then touch 16
// different 16-byte slots of the 256-byte bank row, and the 16 lanes of a ds_write_b64 group write 4 whole rows.
__device__ __forceinline__ int wslot(int r, int lch) { return r * 4 + 2 * ((lch >> 1) ^ ((r >> 3) & 1)) + (lch & 1); }   // u32x2 index
__device__ __forceinline__ int rslot(int r, int h) { return r * 2 + (h ^ ((r >> 3) & 1)); }                                 // u32x4 index


// the six leading cross products of one 32 x 32 x 16 block, smallest terms first: (a3 b1, a2 b2, a1 b3), (a2 b1, a1 b2), a1 b1
__device__ __forceinline__ void mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

}  // namespace
