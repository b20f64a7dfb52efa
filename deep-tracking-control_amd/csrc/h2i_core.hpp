// Block-scaled two-term fp16 operand images ("h2i", round 5): the ONE operand representation of the wide GEMM family
// (csrc/gemm_h2i.hip: forward / data gradient / fused MSE layer; csrc/wgrad_h2i.hip: weight gradients) for the nn.Linear stacks of
// rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349 under ppo.py:197-218, 252, 265, 289, 333.
//
// Why.  Round 4's two-term fp16 kernels scaled every operand by ONE power of two per tensor (amax records, a Python-side registry) and
// converted fp32 -> fp16 x 2 inside every K loop (6.5 / 8.8 VALU per MFMA, MFMA pipe 0.27-0.35 busy).  Here an operand lives in HBM as
// the planes the K loops read by LDS-DMA -- 4 bytes per element, the size of the fp32 tensor it replaces -- written ONCE by the
// epilogue that produces it, with exponents chosen LOCALLY:
//   image(M, K):  chunk (row tile r, stage s) = [plane 2][slot 256][16 bytes] (8 KiB) at ((r * stages + s) * 8 KiB);
//                 plane 0 = hi, plane 1 = lo; slot = rslot(row in tile, k half) holds 8 consecutive k of one row (s3_core.hpp);
//                 behind the last chunk: int32 exps[row tile][k block][128] -- one exponent per ROW and block of 128 columns
//                 (8 stages): the stored pair is (hi, lo) = split(x * 2^e), e = 14 - floor(log2 max|finite x| of the row's block).
//   A row whose block is all zero (or holds no finite value) carries HI_EZERO: consumers skip it when they derive scales.
// hi carries 11 significant bits, lo the next 11; a product is lo hi' + hi lo' + hi hi' (three v_mfma_f32_32x32x16_f16, exact in the
// fp32 accumulator; dropped: lo lo' = 2^-22 of the product).  Every element is exact to 2^-22 of ITS ROW BLOCK's largest element --
// the error model of an fp32 dot product per row, whatever the other rows of the tensor hold -- and an inf / NaN element stays in its
// row (the exponent comes from the finite elements; inf / NaN convert to fp16 inf / NaN and propagate through the MFMA to the
// outputs that depend on them, exactly as in the fp32 reference).
// Consumers:
//   * forward / data gradient (row operand = image, weight = image with one exponent per 128 x 128 block): the accumulators of a row
//     are rescaled by 2^(e_new - e_old) (v_ldexp_f32, exact) at the borders of the 128-column blocks: 64 VALU per 96 MFMAs;
//   * weight gradient (both operands images, the reduction index is the batch ROW): fragments of one operand are multiplied by
//     2^(T - eZ[m] - eX[m]) <= 1 per batch row (4 v_pk_mul_f16 per fragment; T = the block's smallest exponent sum), accumulators
//     rescaled by one scalar at the borders of the 128-row blocks.
#pragma once
#include "s3_core.hpp"

namespace {

constexpr int HI_PLANE = 4096, HI_CHUNK = 2 * HI_PLANE;   // bytes
constexpr int HI_KB = 8;                                  // stages (16 columns each) per exponent block
constexpr int HI_EZERO = 0x7fff;                          // "nothing here": an all-zero / all-non-finite row block
constexpr int HI_EMAX = 100, HI_EMIN = -113;              // clamp: |x| < 2^-86 keeps fewer bits (absurdly small data), 2^127 still fits
typedef __attribute__((address_space(3))) void lds_void;

__host__ __device__ inline long long hi_stages(long long K) { return (K + 15) / 16; }
__host__ __device__ inline long long hi_kblocks(long long K) { return (hi_stages(K) + HI_KB - 1) / HI_KB; }
__host__ __device__ inline long long hi_rtiles(long long M) { return (M + 127) / 128; }
__host__ __device__ inline long long hi_data_bytes(long long M, long long K) { return hi_rtiles(M) * hi_stages(K) * HI_CHUNK; }
__host__ __device__ inline long long hi_bytes(long long M, long long K) { return hi_data_bytes(M, K) + hi_rtiles(M) * hi_kblocks(K) * 512; }

// bit pattern of |v| if v is finite, else 0 (non-finite elements do not take part in the choice of an exponent)
__device__ __forceinline__ u32 finite_bits(float v) {
    const u32 b = abs_bits(v);
    return b < 0x7f800000u ? b : 0u;
}
// exponent of a block whose largest finite |x| has bit pattern `bits`
__device__ __forceinline__ int hi_exp(u32 bits) {
    const int e = 141 - (int)(bits >> 23);
    return bits == 0u ? HI_EZERO : (e > HI_EMAX ? HI_EMAX : e);
}
// 8 consecutive k of one row -> the 16-byte pieces of the two planes
struct HiPiece {
    u32x4 p[2];
};
__device__ __forceinline__ HiPiece hi_split8(const f32x4 (&v)[2], int e) {
    const Split2 a = split2(v[0], e), b = split2(v[1], e);
    HiPiece r;
    r.p[0] = u32x4{a.p[0].x, a.p[0].y, b.p[0].x, b.p[0].y};
    r.p[1] = u32x4{a.p[1].x, a.p[1].y, b.p[1].x, b.p[1].y};
    return r;
}

// ---- narrow rows written straight by the kernel that produces them (latent / loss kernels; K <= 128: one exponent per row): element
// (row, col) of image(M, K) as two 2-byte stores, and the row's exponent.  Same values as h2i_pack_kernel writes for the same row
// (x 2^e rounded to nearest-even fp16, remainder exact in fp32 then rounded); the buffer's padding columns stay as allocated (zero).
__device__ __forceinline__ void hi_store_elem(void* img, int K, int row, int col, float x, int e) {
    const float xs = __builtin_ldexpf(x, e == HI_EZERO ? 0 : e);
    const _Float16 h = (_Float16)xs;
    const _Float16 l = (_Float16)(xs - (float)h);
    const int r = row & 127;
    const long long off = ((long long)(row >> 7) * hi_stages(K) + (col >> 4)) * HI_CHUNK +
                          (2 * r + (((col >> 3) & 1) ^ ((r >> 3) & 1))) * 16 + (col & 7) * 2;
    *reinterpret_cast<_Float16*>(static_cast<char*>(img) + off) = h;
    *reinterpret_cast<_Float16*>(static_cast<char*>(img) + off + HI_PLANE) = l;
}
// four consecutive columns (col0 % 4 == 0) of one row: 8-byte stores into both planes
__device__ __forceinline__ void hi_store4(void* img, int K, int row, int col0, f32x4 v, int e) {
    const Split2 s = split2(v, e == HI_EZERO ? 0 : e);
    const int r = row & 127;
    const long long off = ((long long)(row >> 7) * hi_stages(K) + (col0 >> 4)) * HI_CHUNK +
                          (2 * r + (((col0 >> 3) & 1) ^ ((r >> 3) & 1))) * 16 + (col0 & 7) * 2;
    *reinterpret_cast<u32x2*>(static_cast<char*>(img) + off) = s.p[0];
    *reinterpret_cast<u32x2*>(static_cast<char*>(img) + off + HI_PLANE) = s.p[1];
}
__device__ __forceinline__ void hi_store_row_exp(void* img, long long M, int K, int row, int e) {
    int* exps = reinterpret_cast<int*>(static_cast<char*>(img) + hi_data_bytes(M, K));
    exps[(long long)(row >> 7) * hi_kblocks(K) * 128 + (row & 127)] = e;
}

}  // namespace
