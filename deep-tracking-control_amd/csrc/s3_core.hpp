// Shared device pieces of the split-precision (bf16 x 3) kernels: csrc/gemm_s3.hip (forward / data gradient) and
// csrc/wgrad_s3.hip (weight gradients).  See the header comment of gemm_s3.hip.
#pragma once
#include "amax.hpp"
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

// ---- two-term fp16 split (round 4, "h2") --------------------------------------------------------------------------------------
// An fp32 operand x of a tensor whose largest magnitude is amax is scaled by a power of two (exact) so that amax lands in
// [2^14, 2^15), then written as hi + lo with hi = fp16(x 2^e), lo = fp16(x 2^e - hi) (the remainder is an exact fp32 subtraction; both
// conversions round to nearest even).  hi carries 11 significant bits, lo the next 11: |x 2^e - (hi + lo)| <= max(2^-22 |x 2^e|, 2^-25)
// -- the second bound is fp16's subnormal spacing, reached by elements more than 2^18 below amax (their error is 2^-40 amax).
// A product needs THREE fp16 MFMAs (lo hi', hi lo', hi hi': 11-bit x 11-bit products are exact in the fp32 accumulator; the dropped
// lo lo' is 2^-22 of the product) instead of six bf16 ones, and two planes instead of three in LDS; the accumulator is scaled back by
// 2^-(e + e') (v_ldexp_f32, exact) before the epilogue.  amax travels as the BIT PATTERN of |x| in a u32 slot (unsigned order = float
// order for non-negative floats; a NaN pattern is larger than every number, so a poisoned operand poisons the scale and the result).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 cvt_pk_f16(float lo, float hi) {         // one v_cvt_pk_f16_f32
    return __builtin_bit_cast(u32, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
__device__ __forceinline__ float f16_lo(u32 p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float f16_hi(u32 p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }
// scale exponent of an operand whose amax slot holds `bits`: 14 - floor(log2 amax) for a normal amax; zero / subnormal amax: 141 (the
// scaled values stay below 2^15); inf / NaN: -114 (the values are inf / NaN whatever the scale)
__device__ __forceinline__ int h2_exp(u32 bits) { return 141 - (int)((bits >> 23) & 0xffu); }
struct Split2 {
    u32x2 p[2];
};
__device__ __forceinline__ Split2 split2(f32x4 v, int e) {
    const float x0 = __builtin_ldexpf(v[0], e), x1 = __builtin_ldexpf(v[1], e), x2 = __builtin_ldexpf(v[2], e), x3 = __builtin_ldexpf(v[3], e);
    // (scalar halves on purpose: with the pair taken back out of the u32x2 through a vector bit_cast, hipcc 7.2 subtracted the FIRST
    // pair's values from the second pair as well)
    const u32 h0 = cvt_pk_f16(x0, x1), h1 = cvt_pk_f16(x2, x3);
    Split2 s;
    s.p[0] = u32x2{h0, h1};
    s.p[1] = u32x2{cvt_pk_f16(x0 - f16_lo(h0), x1 - f16_hi(h0)), cvt_pk_f16(x2 - f16_lo(h1), x3 - f16_hi(h1))};
    return s;
}
// the two representations behind one interface: planes per operand, MFMA passes per 16-k block (smallest terms first), the split
template <bool H2>
struct Prec {
    static constexpr int NP = H2 ? 2 : 3, NT = H2 ? 3 : 6;
    u32x2 p[NP];
    static __device__ __forceinline__ int pa(int t) { return H2 ? (t == 0 ? 1 : 0) : (t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0); }
    static __device__ __forceinline__ int pb(int t) { return H2 ? (t == 1 ? 1 : 0) : (t == 2 ? 2 : (t == 1 || t == 4) ? 1 : 0); }
    static __device__ __forceinline__ Prec split(f32x4 v, int e) {
        Prec r;
        if constexpr (H2) {
            const Split2 s = split2(v, e);
            r.p[0] = s.p[0];
            r.p[1] = s.p[1];
        } else {
            const Split3 s = split3(v);
            r.p[0] = s.p[0];
            r.p[1] = s.p[1];
            r.p[2] = s.p[2];
        }
        return r;
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        if constexpr (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
// amax of operands that come without a slot (include/dtc_hip.h: a NULL DtcSeg.amax / dz_amax): ONE memset + ONE launch for all such
// operands of a call, into slots of the call's own scratch.  Item = a column block of a matrix, rows < M (through idx where gathered);
// block = 8192 elements of one item.
struct AmaxItem {
    const float* ptr;
    const long long* idx;       // row map (NULL: plain rows)
    long long ld;
    int col0, width, rows;
    int tw_shift;               // threads across a row = 1 << tw_shift (the power of two >= width, at most 256)
    int rows_per_block, block_end;      // block_end: running sum of blocks over the items
    u32* slot;
};
constexpr int AMAX_MAX_ITEMS = 60, AMAX_BLOCK_ELEMS = 8192;
struct AmaxGroup {
    int count;
    AmaxItem it[AMAX_MAX_ITEMS];
};
// block = rows_per_block whole rows of one item (~8192 elements); thread = (row of the pass, column): consecutive lanes read consecutive
// floats of a row, no division anywhere
__global__ __launch_bounds__(256) void amax_group_kernel(const AmaxGroup G) {
    int i = 0, b = blockIdx.x;
    while (i < G.count - 1 && b >= G.it[i].block_end) ++i;
    if (i > 0) b -= G.it[i - 1].block_end;
    const AmaxItem& I = G.it[i];
    const int tw = 1 << I.tw_shift, c0 = threadIdx.x & (tw - 1), rstep = 256 >> I.tw_shift;
    const int r_end = min(I.rows, (b + 1) * I.rows_per_block);
    u32 m = 0u;
    for (int row = b * I.rows_per_block + (threadIdx.x >> I.tw_shift); row < r_end; row += rstep) {
        const float* src = I.ptr + (I.idx ? I.idx[row] : (long long)row) * I.ld + I.col0;
        for (int c = c0; c < I.width; c += tw) {
            const u32 v = abs_bits(src[c]);
            m = v > m ? v : m;
        }
    }
    amax_publish(I.slot, m);
}
// host side: add an item, run the group (slots: the items' records, `bytes` bytes of device scratch, zeroed here)
inline void amax_item(AmaxGroup& G, const float* ptr, const long long* idx, long long ld, int col0, int width, int rows, u32* slot) {
    AmaxItem& I = G.it[G.count];
    I = AmaxItem{ptr, idx, ld, col0, width, rows, 0, 0, 0, slot};
    while ((1 << I.tw_shift) < width && I.tw_shift < 8) ++I.tw_shift;
    I.rows_per_block = AMAX_BLOCK_ELEMS / width > 0 ? AMAX_BLOCK_ELEMS / width : 1;
    const int step = 256 >> I.tw_shift;
    I.rows_per_block = (I.rows_per_block + step - 1) / step * step;
    I.block_end = (G.count > 0 ? G.it[G.count - 1].block_end : 0) + (rows + I.rows_per_block - 1) / I.rows_per_block;
    ++G.count;
}
__global__ __launch_bounds__(256) void amax_zero_kernel(u32* p, int words) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < words) p[i] = 0u;
}
inline bool amax_group_run(const AmaxGroup& G, void* slots, size_t bytes, hipStream_t s) {
    if (G.count == 0) return true;
    const int words = (int)(bytes / 4);
    hipLaunchKernelGGL(amax_zero_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, (u32*)slots, words);
    hipLaunchKernelGGL(amax_group_kernel, dim3((unsigned)G.it[G.count - 1].block_end), dim3(256), 0, s, G);
    return true;
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
