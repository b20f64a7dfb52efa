// Split-precision weight gradients for gfx950: dW[N,K] = dZ[M,N]^T X[M,K], db[N] = column sums of dZ (the
// `loss.backward()` share of nn.Linear in rsl_rl/rsl_rl/algorithms/ppo.py:252, 333) on the bf16 matrix pipe, every fp32
// operand as three bf16 terms and six MFMA passes (csrc/gemm_s3.hip explains the arithmetic).
//
// Both operands have the REDUCTION index (the batch row m) as their row, while a lane's MFMA fragment needs 8 consecutive
// m of ONE output row / column.  The transposition happens in registers, for free, on the way into LDS:
//   * a loader thread owns a 4 (batch rows) x 4 (consecutive n resp. c) block: four dwordx4 loads along n / c (coalesced:
//     32 threads cover 128 columns of a batch row), then for each of its four columns the four batch-row values are one
//     k chunk of that column's plane row: split3 -> three ds_write_b64.  Threads 0..127 stage dZ, threads 128..255 stage X;
//   * the plane rows are stored in PHYSICAL order p(n) = (n & 3) * 32 + (n >> 2): the 16 lanes of a ds_write_b64 group then
//     write four consecutive physical rows (conflict free) although their logical columns are 4 apart.  MFMA tile t of the
//     128-row plane therefore holds the logical rows n = 4 i + t -- which output element an accumulator register belongs
//     to is only a matter of addressing: the partial slabs are written in physical order (coalesced float4 rows) and the
//     reduce kernel maps them back while it sums the batch slices in a fixed order (deterministic, no atomics).
// Tiles: 128 (n) x 128 (c, inside ONE segment of X), 2 x 2 waves of 64 x 64, stage = 16 batch rows, double-buffered LDS
// (48 KiB); grouped launch over all layers of a gradient bucket and 8 k batch slices (slice s runs on XCD s % 8), as
// csrc/wgrad.hip does for the single-pass kernels.
#include <type_traits>

#include "s3_core.hpp"

namespace {

constexpr int MAX_JOBS_S3 = 12;
constexpr int TILE = 128;

struct S3Job {
    const float* dZ;
    long long lddz;
    long long dz_rows;      // > 0: dZ's rows are gathered through X.idx as well (see DtcWgradJob)
    SegMatDev X;
    float* part;            // [splits][tiles][128][128] physical-order partial products
    float* bpart;           // [splits][row_tiles][128] bias-gradient partials (logical order)
    float* dW;
    float* db;
    int N, K, col_tiles, row_tiles;
    int tile_end;           // running sum of tiles over the jobs
    const u32* za;          // two-term fp16 launches: amax slots of dZ and of X's segments (s3_core.hpp)
    const u32* xa[4];
};
struct S3Group {
    int count, M, rows_per_split, splits, tiles_total;
    int per_xcd;            // > 0: balanced map -- the (slice, tile) items in slice-major order, cut into 8 equal runs, one per XCD
    S3Job job[MAX_JOBS_S3];
};

__device__ __forceinline__ int phys(int x) { return (x & 3) * 32 + (x >> 2); }          // logical -> physical (0..127)
__device__ __forceinline__ int logical(int p) { return 4 * (p & 31) + (p >> 5); }       // physical -> logical

// GATHER_A: some job of the group maps dZ's rows through X.idx (DtcWgradJob.dz_rows); the plain kernel keeps the row arithmetic of
// the dZ loader out of its K loop (with it in, for jobs that do not need it, the loop issued 2.8 instead of 0.55 scalar and 5.9
// instead of 4.8 vector instructions per MFMA, 180 instead of 168 registers -- two instead of three workgroups per CU -- and the
// bench step's grouped launches ran 318 instead of 270 us).  The row-map variant keeps 180 registers (budget 3 would spill one)
// H2 (round 4): operands as two fp16 terms, three MFMA passes per 16 batch rows (s3_core.hpp) instead of three bf16 terms and six
template <bool GATHER_A, bool H2 = false>
#ifndef DTC_WGRAD_H2_WG
#define DTC_WGRAD_H2_WG 2
#endif
__global__ __launch_bounds__(256, H2 ? DTC_WGRAD_H2_WG : 2) void wgrad_s3_group_kernel(const S3Group G) {
    using P = Prec<H2>;
    constexpr int NP = P::NP, NT = P::NT;
    __shared__ __attribute__((aligned(16))) u32x2 As[2][3][TILE * 4];          // (three planes either way: the epilogue's patches need the 48 KiB)
    __shared__ __attribute__((aligned(16))) u32x2 Bs[2][3][TILE * 4];
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    int split, t;
    if (G.per_xcd > 0) {                               // any slice count: every XCD gets the same number of workgroups
        const int q = xcd * G.per_xcd + jb;
        if (q >= G.splits * G.tiles_total) return;
        split = q / G.tiles_total;
        t = q - split * G.tiles_total;
    } else {
        split = xcd + 8 * (jb / G.tiles_total);
        if (split >= G.splits) return;
        t = jb % G.tiles_total;
    }
    int j = 0;
    while (j < G.count - 1 && t >= G.job[j].tile_end) ++j;
    if (j > 0) t -= G.job[j - 1].tile_end;
    const S3Job& J = G.job[j];
    const int tr = t / J.col_tiles;
    int tc = t - tr * J.col_tiles;
    int seg = 0;
    for (; seg < J.X.nseg - 1; ++seg) {
        const int nt = (J.X.s[seg].width + TILE - 1) / TILE;
        if (tc < nt) break;
        tc -= nt;
    }
    const SegDev sd = J.X.s[seg];
    const int N = J.N, M = G.M;
    const int n0 = tr * TILE, lc0 = tc * TILE;
    const int m_begin = split * G.rows_per_split;
    const int m_end = min(M, m_begin + G.rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // ---- loaders: threads 0..127 stage dZ (A), 128..255 stage X (B); thread (g, lch): columns 4 g .. 4 g + 3, batch rows
    // 4 lch .. 4 lch + 3 of the stage
    const bool is_a = tid < 128;
    // g is the fast index: a load instruction covers two whole 512-byte runs of two batch rows.  (With lch fast -- 4 rows x 256
    // bytes per instruction -- the ds_write_b64 of a 16-lane group cover four whole plane rows and SQ_LDS_BANK_CONFLICT drops from
    // 95 % of the LDS cycles to 0, but the kernel runs SLOWER, 325 vs 266 us on the 61-tile bucket: the LDS is not what bounds it.
    // Second attempt with the global pattern UNCHANGED -- lane = 2 g + (lch & 1), planes as [k half][row][16 bytes]: the 16 lanes of
    // a write group then cover 128 contiguous bytes, conflict free, fragment reads stay one ds_read_b128 -- 64.96 vs 64.24 ms per
    // step (3 interleaved runs each): no gain either.)
    const int lt = tid & 127, g = lt & 31, lch = lt >> 5;
    const long long lddz = J.lddz;
    const bool gather_a = GATHER_A && J.dz_rows > 0;
    const rsrc_t ares = make_rsrc_bytes(J.dZ, (gather_a ? J.dz_rows : (long long)M) * lddz * 4);
    const rsrc_t bres = make_rsrc_bytes(sd.ptr, (long long)sd.rows * sd.ld * 4);
    // column offset of this thread's float4 (INVALID behind the matrix / the segment: those lanes stage zeros)
    const u32 acol = (n0 + 4 * g < N) ? (u32)(n0 + 4 * g) * 4u : INVALID;
    const u32 bcol = (lc0 + 4 * g < sd.width) ? (u32)(sd.col0 + lc0 + 4 * g) * 4u : INVALID;
    const u32 ldb = (u32)sd.ld * 4u, lda = (u32)lddz * 4u;
    // physical row of column 4 g + e = 32 e + g: rows 32 apart share the half-swap parity, so the four slots are slot0 + 128 e
    // (immediate offsets of one address register)
    const int slot0 = wslot(g, lch);
    int ez = 0, exx = 0;                               // fp16 path: scale exponents of dZ and of this tile's segment of X
    bool poison = false;                               // an inf / NaN amax: the tile's result turns NaN
    if constexpr (H2) {
        const u32 mz = amax_read(J.za), mx = amax_read(J.xa[seg]);
        ez = __builtin_amdgcn_readfirstlane(h2_exp(mz));
        exx = __builtin_amdgcn_readfirstlane(h2_exp(mx));
        poison = __builtin_amdgcn_readfirstlane((mz >= 0x7f800000u || mx >= 0x7f800000u) ? 1 : 0) != 0;
    }
    const int e_mine = is_a ? ez : exx;

    // AHW (fp16 kernels, round 4; -DDTC_H2_WGRAD_AHEAD2=1, OFF): the operand loads TWO stages ahead of the MFMAs (two register sets, set =
    // parity of the stage), as linear_s3_kernel<.., H2> has them.  Here both operands travel through registers: 180 instead of 152
    // registers = two instead of three workgroups per CU, 59.4 vs 54.5 ms per step; squeezed into three (-DDTC_WGRAD_H2_WG=3): 58.8 vs 57.3.
#ifndef DTC_H2_WGRAD_AHEAD2
#define DTC_H2_WGRAD_AHEAD2 0
#endif
    constexpr bool AHW = H2 && !GATHER_A && (DTC_H2_WGRAD_AHEAD2 != 0);
    f32x4 vs[AHW ? 2 : 1][4];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    u32 rnext[4];
    auto rows_of = [&](int mb) {                       // source rows of X for the stage starting at batch row mb
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mb + 4 * lch + q;
            const int mc = m < m_end ? m : m_end - 1;
            rnext[q] = (is_a ? gather_a : sd.gather != 0) ? (u32)J.X.idx[mc] : (u32)mc;
        }
    };
    if (GATHER_A || !is_a) rows_of(m_begin);
    auto load_stage = [&](int mb, auto setc) {
        f32x4(&v)[4] = vs[AHW ? decltype(setc)::value : 0];
        if (is_a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = mb + 4 * lch + q;
                v[q] = bload4(ares, (acol + (GATHER_A ? rnext[q] : (u32)m) * lda) | oob_mask(m, m_end - 1), 0u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = mb + 4 * lch + q;
                v[q] = bload4(bres, (bcol + rnext[q] * ldb) | oob_mask(m, m_end - 1), 0u);
            }
        }
        if (GATHER_A || !is_a) rows_of(mb + BK);
    };
    // BIAS (workgroup-uniform): the first column tile of a layer also sums dZ's columns (12 of the ~100 VALU of a stage that the
    // other tiles, and the X side everywhere, do not need to issue)
    auto store_stage = [&](int buf, auto bias, auto setc) {
        f32x4(&v)[4] = vs[AHW ? decltype(setc)::value : 0];
        u32x2(*dst)[TILE * 4] = is_a ? As[buf] : Bs[buf];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 col = {v[0][e], v[1][e], v[2][e], v[3][e]};          // four batch rows of column 4 g + e
            const P s = P::split(col, e_mine);
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[p][slot0 + 128 * e] = s.p[p];
            if constexpr (decltype(bias)::value) bsum[e] += (col[0] + col[1]) + (col[2] + col[3]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    auto mfma_stage = [&](int buf) {
        u32x4 a[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i][p] = reinterpret_cast<const u32x4*>(&As[buf][p][0])[rslot((2 * wr + i) * 32 + l31, half)];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            u32x4 b[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                b[p] = reinterpret_cast<const u32x4*>(&Bs[buf][p][0])[rslot((2 * wc + jj) * 32 + l31, half)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[i][jj] = P::mfma(a[i][P::pa(t)], b[P::pb(t)], acc[i][jj]);
        }
    };

    // fused stage: the MFMAs of LDS[buf] with the conversion + LDS store of the loaded block (-> LDS[buf ^ 1]) placed between the
    // MFMAs of the second column tile, one column of the block per three MFMAs (see linear_s3_kernel)
    auto stage_ilv = [&](int buf, auto bias, auto setc) {        // setc: the register set that is converted (-> LDS[buf ^ 1])
        f32x4(&v)[4] = vs[AHW ? decltype(setc)::value : 0];
        u32x4 a[2][NP], b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i][p] = reinterpret_cast<const u32x4*>(&As[buf][p][0])[rslot((2 * wr + i) * 32 + l31, half)];
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = reinterpret_cast<const u32x4*>(&Bs[buf][p][0])[rslot((2 * wc) * 32 + l31, half)];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][0] = P::mfma(a[i][P::pa(t)], b[P::pb(t)], acc[i][0]);
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = reinterpret_cast<const u32x4*>(&Bs[buf][p][0])[rslot((2 * wc + 1) * 32 + l31, half)];
        u32x2(*dst)[TILE * 4] = is_a ? As[buf ^ 1] : Bs[buf ^ 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 col = {v[0][e], v[1][e], v[2][e], v[3][e]};
            const P sp = P::split(col, e_mine);
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[p][slot0 + 128 * e] = sp.p[p];
            if constexpr (decltype(bias)::value) bsum[e] += (col[0] + col[1]) + (col[2] + col[3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = (2 * NT * e) / 4; m < (2 * NT * (e + 1)) / 4; ++m) {      // 3 (fp16 terms: 2, 1, 2, 1) MFMAs behind every column
                const int t = m >> 1, i = m & 1;
                acc[i][1] = P::mfma(a[i][P::pa(t)], b[P::pb(t)], acc[i][1]);
            }
        }
    };

    const int KT = (m_end - m_begin + BK - 1) / BK;
    auto k_loop = [&](auto bias) {
        if constexpr (AHW) {
            // stage kt lives in register set kt & 1 and goes to LDS[kt & 1]; half step HS<P>(kt), P = kt & 1: the loads of stage kt + 1
            // are issued (into the set stage kt - 1 was converted from), LDS[P ^ 1] (stage kt - 1) feeds the MFMAs, set P (stage kt,
            // loaded one half step earlier) is converted into LDS[P].  Loads past the slice read masked rows: zeros, never stored.
            load_stage(m_begin, S0{});
            store_stage(0, bias, S0{});
            load_stage(m_begin + BK, S1{});
            __syncthreads();
            int kt = 1;
            for (; kt + 1 < KT; kt += 2) {                   // kt is odd here
                load_stage(m_begin + (kt + 1) * BK, S0{});
                stage_ilv(0, bias, S1{});
                __syncthreads();
                load_stage(m_begin + (kt + 2) * BK, S1{});
                stage_ilv(1, bias, S0{});
                __syncthreads();
            }
            if (kt < KT) {
                load_stage(m_begin + (kt + 1) * BK, S0{});
                stage_ilv(0, bias, S1{});
                __syncthreads();
            }
            mfma_stage((KT - 1) & 1);
            return;
        }
        int buf = 0;
        load_stage(m_begin, S0{});
        store_stage(0, bias, S0{});
        __syncthreads();
        for (int kt = 1; kt < KT; ++kt) {
            load_stage(m_begin + kt * BK, S0{});
#ifndef DTC_S3_NO_ILV
            stage_ilv(buf, bias, S0{});
#else
            mfma_stage(buf);
            store_stage(buf ^ 1, bias, S0{});
#endif
            __syncthreads();
            buf ^= 1;
        }
        mfma_stage(buf);
    };
    const bool need_bias = seg == 0 && tc == 0;
    if (KT > 0) {
        if (need_bias) k_loop(std::true_type{});
        else k_loop(std::false_type{});
    }

    // ---- epilogue: the accumulators in physical order -> slab tile [128][128] (float4 rows through the wave's LDS patch)
    if constexpr (H2) {                                // the sums carry 2^(ez + exx): scaled back exactly
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = poison ? __builtin_nanf("") : __builtin_ldexpf(acc[i][jj][r], -(ez + exx));
    }
    __syncthreads();
    float* slab = J.part + ((long long)split * (J.tile_end - (j > 0 ? G.job[j - 1].tile_end : 0)) + t) * (TILE * TILE);
    float* patch = reinterpret_cast<float*>(&As[0][0][0]) + wave * (32 * LDW);
    const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            patch_put(patch, acc[i][jj], half, l31);
            float* q = slab + (long long)((2 * wr + i) * 32 + prow) * TILE + (2 * wc + jj) * 32 + 4 * pc4;
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(q + (long long)(8 * p) * TILE) = patch_get(patch, prow + 8 * p, pc4);
        }
    // bias-gradient partial (first column tile of the layer only): thread (g, lch) summed columns 4 g + e over its batch rows
    if (need_bias) {
        float* red = reinterpret_cast<float*>(&Bs[0][0][0]);          // [4][128], disjoint from the patches in As
        if (is_a) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[lch * TILE + 4 * g + e] = bsum[e];
        }
        __syncthreads();
        if (tid < TILE) J.bpart[((long long)split * J.row_tiles + tr) * TILE + tid] = ((red[tid] + red[TILE + tid]) + red[2 * TILE + tid]) + red[3 * TILE + tid];
    }
}

// Sum of the batch slices in a fixed order, physical -> logical mapping, dW / db written once.
// block = (tile, 8 physical rows); thread = (physical row, float4 of physical columns)
__global__ __launch_bounds__(256) void wgrad_s3_reduce_kernel(const S3Group G) {
    int b = blockIdx.x;
    const int blocks_tiles = G.tiles_total * 16;
    if (b >= blocks_tiles) {                           // bias blocks: one per (job, row tile)
        b -= blocks_tiles;
        int j = 0, rt = b;
        while (j < G.count - 1 && rt >= G.job[j].row_tiles) { rt -= G.job[j].row_tiles; ++j; }
        const S3Job& J = G.job[j];
        if (rt >= J.row_tiles || J.db == nullptr || threadIdx.x >= TILE) return;
        const int n = rt * TILE + threadIdx.x;
        if (n >= J.N) return;
        float s = 0.f;
        for (int sp = 0; sp < G.splits; ++sp) s += J.bpart[((long long)sp * J.row_tiles + rt) * TILE + threadIdx.x];
        J.db[n] = s;
        return;
    }
    int t = b >> 4;
    const int rg = b & 15;
    int j = 0;
    while (j < G.count - 1 && t >= G.job[j].tile_end) ++j;
    const int tiles_j = G.job[j].tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
    if (j > 0) t -= G.job[j - 1].tile_end;
    const S3Job& J = G.job[j];
    const int tr = t / J.col_tiles;
    int tc = t - tr * J.col_tiles;
    int seg = 0;
    for (; seg < J.X.nseg - 1; ++seg) {
        const int nt = (J.X.s[seg].width + TILE - 1) / TILE;
        if (tc < nt) break;
        tc -= nt;
    }
    const SegDev& sd = J.X.s[seg];
    const int pn = rg * 8 + (threadIdx.x >> 5), pc4 = threadIdx.x & 31;
    const int n = tr * TILE + logical(pn);
    if (n >= J.N) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(J.part + (long long)t * (TILE * TILE) + (long long)pn * TILE) + pc4;
    const long long step = (long long)tiles_j * (TILE * TILE / 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 3 < G.splits; sp += 4) {
        const f32x4 v0 = p[(long long)sp * step], v1 = p[(long long)(sp + 1) * step], v2 = p[(long long)(sp + 2) * step], v3 = p[(long long)(sp + 3) * step];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (((acc[e] + v0[e]) + v1[e]) + v2[e]) + v3[e];
    }
    for (; sp < G.splits; ++sp) {
        const f32x4 v0 = p[(long long)sp * step];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v0[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int lc = tc * TILE + logical(4 * pc4 + e);
        if (lc < sd.width) J.dW[(long long)n * J.K + sd.start + lc] = acc[e];
    }
}

// DTC_WGRAD_S3_FILL=1: as many batch slices as fill the chip's 768 workgroup slots ONCE (61 tiles: 12 slices = 732 workgroups
// instead of 8 = 488 on 768 slots), with the balanced (slice, tile) -> XCD map
bool fill_slots() {
    static const bool on = [] {
        return false;                                // (measured no faster; kept as code for the balanced map below)
    }();
    return on;
}

int split_cap() {
    static const int cap = [] {
        return 24;
    }();
    return cap;
}

int group_splits_s3(int M, int tiles_total) {
    // 768: eight batch slices for the 61..70-tile groups of the bench step.  Measured (tools/jobs/r3_sweep2.sh): 512..1536 within
    // 0.5 % of each other in step time, 2048 (32 slices) no faster; the partial slabs are HBM traffic written and read once
    // per slice, so the smallest count that still fills the chip wins (family traffic 1.30 -> 1.15 x the algorithmic bytes)
    // (round 4, two-term fp16 kernels -- half the matrix work per workgroup: 1536 = 24 slices for the 61..64-tile groups, 16 for the
    // 70-tile one: 57.4 vs 60.2 ms per step; 2048 / 3072 with a higher cap: 59.7 / 61.0 vs 59.1 on another box)
    constexpr int target = 1536;
    int s = target / (tiles_total > 0 ? tiles_total : 1);
    if (!fill_slots()) s = s / 8 * 8;                 // the slice -> XCD map needs whole groups of eight
    if (s < 8) s = 8;
    if (s > split_cap()) s = split_cap();               // small groups: bounded slab traffic (64 KiB per tile and slice, written and re-read)
    const int max_s = (int)dtc::ceil_div(dtc::ceil_div(M, BK * 8), 8) * 8;
    if (s > max_s) s = max_s;
    return s;
}

constexpr int SLOT_BYTES = MAX_JOBS_S3 * 5 * AMAX_RECORD_BYTES;      // 12 jobs x (dZ + 4 segments of X) records
struct S3Plan {
    S3Group dev;
    u32* slots;
    long long bytes;
    int red_blocks;
    double flop, algo_bytes;
};

int plan_s3(const DtcWgradJob* jobs, int count, int M, void* workspace, S3Plan& P) {
    DTC_REQUIRE(jobs != nullptr && count >= 1 && count <= MAX_JOBS_S3, "job count %d outside 1..%d", count, MAX_JOBS_S3);
    DTC_REQUIRE(M > 0, "bad M=%d", M);
    S3Group& G = P.dev;
    G.count = count;
    G.M = M;
    int tiles = 0, row_tiles = 0;
    for (int j = 0; j < count; ++j) {
        const DtcWgradJob& h = jobs[j];
        DTC_REQUIRE(h.N > 0 && h.K > 0 && h.lddz >= h.N, "job %d: bad shape N=%d K=%d lddz=%lld", j, h.N, h.K, (long long)h.lddz);
        DTC_REQUIRE(h.dZ && h.dW, "job %d: null pointer", j);
        DTC_REQUIRE((h.dz_rows > 0 ? h.dz_rows : (long long)M) * h.lddz <= MAX_ELEMS, "job %d: matrix too large", j);
        DTC_REQUIRE(h.dz_rows >= 0 && (h.dz_rows == 0 || h.X.idx != nullptr), "job %d: dz_rows needs the row map X.idx", j);
        S3Job& d = G.job[j];
        int rc = to_dev(&h.X, d.X, h.K, false, M);
        if (rc != DTC_OK) return rc;
        d.dZ = h.dZ;
        d.lddz = h.lddz;
        d.dz_rows = h.dz_rows;
        d.dW = h.dW;
        d.db = h.db;
        d.za = h.dz_amax;
        for (int i = 0; i < 4; ++i) d.xa[i] = i < h.X.nseg ? h.X.seg[i].amax : nullptr;
        d.N = h.N;
        d.K = h.K;
        d.col_tiles = 0;
        for (int i = 0; i < d.X.nseg; ++i) d.col_tiles += (int)dtc::ceil_div(d.X.s[i].width, TILE);
        d.row_tiles = (int)dtc::ceil_div(h.N, TILE);
        tiles += d.row_tiles * d.col_tiles;
        row_tiles += d.row_tiles;
        d.tile_end = tiles;
    }
    G.tiles_total = tiles;
    G.splits = group_splits_s3(M, tiles);
    G.rows_per_split = (int)dtc::ceil_div(dtc::ceil_div(M, G.splits), BK) * BK;
    G.per_xcd = (G.splits % 8) ? (int)dtc::ceil_div((long long)G.splits * tiles, 8) : 0;
    long long off = 0;
    P.flop = P.algo_bytes = 0.0;
    for (int j = 0; j < count; ++j) {
        S3Job& d = G.job[j];
        const long long tiles_j = d.tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
        d.part = workspace ? (float*)((char*)workspace + off) : nullptr;
        off += (long long)G.splits * tiles_j * TILE * TILE * (long long)sizeof(float);
        d.bpart = workspace ? (float*)((char*)workspace + off) : nullptr;
        off += (long long)G.splits * d.row_tiles * TILE * (long long)sizeof(float);
        P.flop += 2.0 * M * (double)d.N * d.K;
        P.algo_bytes += 4.0 * ((double)M * d.N + (double)M * d.K + (double)d.N * (d.K + 1));
    }
    P.slots = workspace ? (u32*)((char*)workspace + off) : nullptr;      // fp16 launches: amax slots of operands that came without one
    off += SLOT_BYTES;
    P.bytes = off;
    P.red_blocks = tiles * 16 + row_tiles;
    return DTC_OK;
}

}  // namespace

extern "C" int64_t dtc_wgrad_group_s3_workspace(const DtcWgradJob* jobs, int count, int M) {
    S3Plan P;
    if (plan_s3(jobs, count, M, nullptr, P) != DTC_OK) return -1;
    return P.bytes;
}

namespace {
int wgrad_group_s3(const DtcWgradJob* jobs, int count, int M, void* workspace, void* stream, bool h2) {
    DTC_REQUIRE(workspace != nullptr && dtc::aligned16(workspace), "wgrad group workspace must be a 16-byte aligned device buffer");
    S3Plan P;
    int rc = plan_s3(jobs, count, M, workspace, P);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (h2) {
        // operands that came without an amax slot: one memset + one launch for all of them, slots in the workspace's tail
        AmaxGroup A;
        A.count = 0;
        S3Group& Gm = P.dev;
        for (int j = 0; j < Gm.count; ++j) {
            S3Job& d = Gm.job[j];
            if (d.za == nullptr) {
                u32* slot = P.slots + (5 * j) * (AMAX_RECORD_BYTES / 4);
                amax_item(A, d.dZ, d.dz_rows > 0 ? d.X.idx : nullptr, d.lddz, 0, d.N, M, slot);
                d.za = slot;
            }
            for (int i = 0; i < d.X.nseg; ++i)
                if (d.xa[i] == nullptr) {
                    u32* slot = P.slots + (5 * j + 1 + i) * (AMAX_RECORD_BYTES / 4);
                    const SegDev& sd = d.X.s[i];
                    amax_item(A, sd.ptr, sd.gather ? d.X.idx : nullptr, sd.ld, sd.col0, sd.width, M, slot);
                    d.xa[i] = slot;
                }
        }
        DTC_REQUIRE(amax_group_run(A, P.slots, SLOT_BYTES, s), "hipMemsetAsync failed");
    }
    const S3Group& G = P.dev;
    {
        dtc::ProfScope prof(dtc::prof_shape_name("linear_wgrad", M, G.tiles_total, count), P.flop, s, P.algo_bytes);
        const int grid = G.per_xcd > 0 ? 8 * G.per_xcd : G.tiles_total * 8 * (int)dtc::ceil_div(G.splits, 8);
        bool any_rows = false;
        for (int j = 0; j < G.count; ++j) any_rows = any_rows || G.job[j].dz_rows > 0;
        if (h2) {
            if (any_rows) hipLaunchKernelGGL((wgrad_s3_group_kernel<true, true>), dim3(grid), dim3(256), 0, s, G);
            else hipLaunchKernelGGL((wgrad_s3_group_kernel<false, true>), dim3(grid), dim3(256), 0, s, G);
        } else if (any_rows) hipLaunchKernelGGL((wgrad_s3_group_kernel<true>), dim3(grid), dim3(256), 0, s, G);
        else hipLaunchKernelGGL((wgrad_s3_group_kernel<false>), dim3(grid), dim3(256), 0, s, G);
    }
    {
        dtc::ProfScope prof(dtc::prof_shape_name("wgrad_reduce", G.splits, G.tiles_total, count), (double)P.bytes + P.bytes / (double)G.splits, s);
        hipLaunchKernelGGL(wgrad_s3_reduce_kernel, dim3(P.red_blocks), dim3(256), 0, s, G);
    }
    return dtc::check_launch("wgrad_group_s3");
}
}  // namespace

extern "C" int dtc_wgrad_group_s3(const DtcWgradJob* jobs, int count, int M, void* workspace, void* stream) {
    return wgrad_group_s3(jobs, count, M, workspace, stream, false);
}
// the same launches on the two-term fp16 path (include/dtc_hip.h: every job brings dz_amax and X.seg[i].amax)
extern "C" int dtc_wgrad_group_h2(const DtcWgradJob* jobs, int count, int M, void* workspace, void* stream) {
    return wgrad_group_s3(jobs, count, M, workspace, stream, true);
}
