// Shared device / host pieces of the fp32 MFMA dense-layer kernels (csrc/gemm.hip: forward, data gradient, fused GRU
// step; csrc/wgrad.hip: weight gradients).  See the header comment of gemm.hip for the design.
#pragma once
#include <stdlib.h>

#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

constexpr int BM = 128;
#ifndef DTC_BK
#define DTC_BK 16
#endif
constexpr int BK = DTC_BK;             // K step; loaders derive their geometry from it
constexpr int RP = 256 / BK;           // tile rows covered per loader pass of the k-contiguous operands
constexpr int PAD = 4;

// split partials: row n of a split holds dW[n, 0:K] and the bias-gradient partial at column K; rows are padded to a
// multiple of 4 floats so that the reduction streams them with 16-byte loads
__host__ __device__ __forceinline__ int part_ld(int K) { return (K + 1 + 3) & ~3; }

struct SegDev {
    float* ptr;
    long long ld;
    int col0, start, width, gather, accumulate;
    int rows;           // inputs: rows of the source matrix (bounds the buffer descriptor: reads past it return 0)
};
struct SegMatDev {
    int nseg, gathers;      // gathers: number of row-gathered segments (host-computed: no kernarg walk in the kernels)
    const long long* idx;
    SegDev s[4];
};

__device__ __forceinline__ int find_seg(const SegMatDev& X, int k) {
    int s = 0;
    if (X.nseg > 1 && k >= X.s[1].start) s = 1;
    if (X.nseg > 2 && k >= X.s[2].start) s = 2;
    if (X.nseg > 3 && k >= X.s[3].start) s = 3;
    return s;
}

constexpr float SELU_L = 1.0507009873554804934193349852946f, SELU_LA = 1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f;
// The model's own activations (none / ReLU / ELU) are inlined into the unrolled epilogues; the rest of the reference's
// table goes through out-of-line functions so that the hot kernels keep their register budget.
__device__ __attribute__((noinline)) float act_fwd_rare(float v, int act) {
    switch (act) {
        case DTC_ACT_SELU: return v > 0.f ? SELU_L * v : SELU_LA * expm1f(v);
        case DTC_ACT_LRELU: return v > 0.f ? v : 0.01f * v;
        case DTC_ACT_TANH: return tanhf(v);
        case DTC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}
__device__ __attribute__((noinline)) float act_bwd_rare(float g, float y, int act) {
    switch (act) {
        case DTC_ACT_SELU: return y > 0.f ? g * SELU_L : g * (y + SELU_LA);
        case DTC_ACT_LRELU: return y > 0.f ? g : 0.01f * g;
        case DTC_ACT_TANH: return g * (1.0f - y * y);
        case DTC_ACT_SIGMOID: return g * (y * (1.0f - y));
        default: return g;
    }
}
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == DTC_ACT_RELU) return v <= 0.f ? 0.f : v;          // (NaN passes through, as torch.relu)
    if (act == DTC_ACT_ELU) return v > 0.f ? v : expm1f(v);
    if (act > DTC_ACT_ELU) return act_fwd_rare(v, act);
    return v;
}
// derivative expressed through the saved post-activation output y
__device__ __forceinline__ float act_bwd(float g, float y, int act) {
    if (act == DTC_ACT_RELU) return y > 0.f ? g : 0.f;
    if (act == DTC_ACT_ELU) return y > 0.f ? g : g * (y + 1.0f);
    if (act > DTC_ACT_ELU) return act_bwd_rare(g, y, act);
    return g;
}

// Buffer loads: `buffer_load_dword v, voff, s[rsrc], soff offen` -- 128-bit descriptor + uniform byte offset in
// SGPRs, 32-bit lane offset in a VGPR: zero address arithmetic per load inside the K loop.  A lane offset of
// INVALID (>= num_records) makes the hardware return 0 without touching memory: that is how row / k tails
// are zero-filled (no clamps, no masks, no over-reads).  All valid offsets must stay below 2 GiB.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr u32 INVALID = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)INVALID, 0x00020000);
}
__device__ __forceinline__ rsrc_t make_rsrc_bytes(const void* p, long long bytes) {   // loads past `bytes` return 0
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)(u32)bytes, 0x00020000);
}
// sign-bit mask: INVALID when x > limit (both < 2^31), else 0 -- pure arithmetic, because hipcc turns a
// `cond ? INVALID : off` select feeding a load into two predicated loads behind exec-mask branches
__device__ __forceinline__ u32 oob_mask(int x, int limit) { return (u32)(limit - x) & INVALID; }
__device__ __forceinline__ float bload(rsrc_t r, u32 voff, u32 soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

template <int BN>
struct Cfg {
    static constexpr int WM = (BN == 128) ? 2 : 4;      // waves along the row dimension
    static constexpr int WN = 4 / WM;                   // waves along the column dimension
    static constexpr int TM = BM / (32 * WM);           // 32x32 MFMA tiles per wave (rows)
    static constexpr int TN = BN / (32 * WN);           // 32x32 MFMA tiles per wave (cols)
    static constexpr int LDA = BM + PAD;
    static constexpr int LDB = BN + PAD;
};

// XCD-aware tile mapping: returns false for padding blocks.
__device__ __forceinline__ bool map_tile(int b, int row_tiles, int col_tiles, int& tr, int& tc) {
    const int xcd = b & 7;
    const int j = b >> 3;
    const int local = j / col_tiles;
    tc = j - local * col_tiles;
    tr = xcd + 8 * local;
    return tr < row_tiles;
}
inline int grid_for(int row_tiles, int col_tiles) { return 8 * (int)dtc::ceil_div(row_tiles, 8) * col_tiles; }
// extra dynamic LDS per workgroup so that only `occ` workgroups of a kernel fit on a CU (DTC_GEMM_OCC_<which>=occ overrides
// the default; 0 = whatever registers and the static LDS allow).  Default: while the caller runs weight gradients on a
// second stream (dtc_set_concurrency_hint(1)) the data-gradient kernel is capped at 5 (its 80 registers would allow 6 and fill
// the register file): the sixth slot is worth more to the concurrent weight-gradient workgroups (measured: 95.1 -> 94.1 ms
// per step; alone on the device the cap costs the kernel ~3 %; every other cap tried on the forward / weight-gradient
// kernels was neutral or slower)
inline unsigned occ_pad(const char* which, unsigned static_lds, int default_occ = 0) {
    char name[64];
    snprintf(name, sizeof name, "DTC_GEMM_OCC_%s", which);
    const char* e = getenv(name);
    const int occ = e ? atoi(e) : default_occ;
    if (occ <= 0) return 0;
    const unsigned per = (160u * 1024u / (unsigned)occ) & ~255u;
    return per > static_lds + 512u ? per - static_lds - 256u : 0u;
}

template <int BN>
__device__ __forceinline__ void mfma_step(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[Cfg<BN>::TM][Cfg<BN>::TN], int lane, int wm_off, int wn_off) {
    using C = Cfg<BN>;
    const int half = lane >> 5, l31 = lane & 31;
    const float* ap = As + half * C::LDA + wm_off + l31;
    const float* bp = Bs + half * C::LDB + wn_off + l31;
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
        float a[C::TM], b[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i) a[i] = ap[2 * kp * C::LDA + 32 * i];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) b[j] = bp[2 * kp * C::LDB + 32 * j];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

struct Masked { static constexpr bool value = true; };
struct Full { static constexpr bool value = false; };
struct S0 { static constexpr int value = 0; };      // register-set selectors of the two-step-ahead loaders
struct S1 { static constexpr int value = 1; };

template <int BN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[Cfg<BN>::TM][Cfg<BN>::TN]) {
#pragma unroll
    for (int i = 0; i < Cfg<BN>::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg<BN>::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload4(rsrc_t r, u32 voff, u32 soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ int kslot(int r, int chunk) { return r * 4 + (chunk ^ ((r >> 2) & 3)); }


// wide-store epilogue helper: a wave moves one 32x32 accumulator tile (lane = column, 16 rows per lane) through a
// private LDS patch so that every lane ends up with 4 consecutive columns of one row (4 rows per lane): 4 dwordx4
// accesses instead of 16 dword accesses per tile.  One wave's LDS operations execute in order: no barrier.
constexpr int LDW = 32;      // unpadded 128-byte rows: the b32 writes (32 consecutive floats per half-wave) and the b128 reads
                             // (lane groups {0-3, 12-15, 20-27} -> rows 0..3 at chunks 0-3 / 4-7 / 4-7 / 0-3) are conflict free;
                             // 4 waves x 32 x 32 floats = 16 KiB = the A stage buffers the patches live in
__device__ __forceinline__ void patch_put(float* patch, const f32x16& v, int half, int l31) {
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * LDW + l31] = v[r];
}
__device__ __forceinline__ f32x4 patch_get(const float* patch, int row, int c4) {
    return *reinterpret_cast<const f32x4*>(&patch[row * LDW + 4 * c4]);
}


constexpr long long MAX_ELEMS = (1ll << 29) - 1;     // lane byte offsets must stay below 2 GiB (INVALID = 2^31)

int to_dev(const DtcSegMat* h, SegMatDev& d, int expect_cols, bool is_output, long long rows_bound) {
    DTC_REQUIRE(h != nullptr, "segmented matrix is null");
    DTC_REQUIRE(h->nseg >= 1 && h->nseg <= 4, "nseg=%d out of range", h->nseg);
    d.nseg = h->nseg;
    d.idx = (const long long*)h->idx;
    int start = 0;
    for (int i = 0; i < 4; ++i) {
        SegDev& s = d.s[i];
        if (i < h->nseg) {
            const DtcSeg& hs = h->seg[i];
            DTC_REQUIRE(hs.width > 0 && hs.col0 >= 0, "segment %d: bad width/col0", i);
            DTC_REQUIRE(is_output || hs.ptr != nullptr, "segment %d: null source", i);
            DTC_REQUIRE(!hs.gather || h->idx != nullptr, "segment %d: gather without idx", i);
            DTC_REQUIRE(!(is_output && hs.gather), "segment %d: gathered destination unsupported", i);
            DTC_REQUIRE(hs.gather || hs.ld * rows_bound <= MAX_ELEMS, "segment %d: matrix exceeds 2^29 elements (2 GiB)", i);
            DTC_REQUIRE(is_output || (hs.rows > 0 && hs.rows * hs.ld <= MAX_ELEMS && (hs.gather || hs.rows >= rows_bound)),
                        "segment %d: rows=%lld (source matrix rows) missing, too small or beyond 2^29 elements", i, (long long)hs.rows);
            DTC_REQUIRE(is_output || hs.col0 + hs.width <= hs.ld, "segment %d: columns [%d, %d) outside the %lld-wide source", i,
                        hs.col0, hs.col0 + hs.width, (long long)hs.ld);
            s.ptr = hs.ptr;
            s.ld = hs.ld;
            s.col0 = hs.col0;
            s.start = start;
            s.width = hs.width;
            s.gather = hs.gather;
            s.accumulate = hs.accumulate;
            s.rows = (int)hs.rows;
            start += hs.width;
        } else {
            s = SegDev{nullptr, 0, 0, 0x7fffffff, 0, 0, 0, 0};
        }
    }
    d.gathers = 0;
    for (int i = 0; i < h->nseg; ++i) d.gathers += h->seg[i].gather != 0;
    DTC_REQUIRE(start == expect_cols, "segments cover %d columns, expected %d", start, expect_cols);
    return DTC_OK;
}

}  // namespace
