// Weight gradients from block-scaled two-term fp16 operand images (round 5; format: csrc/h2i_core.hpp): dW [N, K] = dZ^T X and
// db = colsum(dZ) for the nn.Linear stacks of rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349 under loss.backward()
// (ppo.py:252, 333), with BOTH operands given as the images their producers wrote.
//
// The reduction index is the batch row = the ROW index of both images, so a stage (16 batch rows x 128 columns of each operand) arrives
// by LDS-DMA as 16-byte pieces (8 columns of one batch row) in the sub-block layout
//   stage image [R = row / 4][Q = column / 32][a = row % 4][b = (column / 8) % 4][16 bytes]            (4 KiB per plane and operand)
// and the MFMA fragments (8 consecutive batch rows of ONE column per lane) come out of it by the hardware transpose read
// ds_read_b64_tr_b16 (tools/probes/tr16_dma.hip; conflict free).  The images carry one exponent per batch row and 128-column block:
// row m of dZ is stored times 2^eZ[m], row m of X times 2^eX[m].  Per 128-row block the workgroup forms T = min_m (eZ[m] + eX[m]) and the
// fp16 factors f[m] = 2^(T - eZ[m] - eX[m]) <= 1; the X fragments are multiplied by f (4 v_pk_mul_f16 per fragment: exact for the rows
// that carry the block's weight, rows far below it lose low-order bits of products that are far below the sum), the accumulators hold
// 2^T x the true sums and are rescaled by ONE scalar at the borders of the 128-row blocks.  The bias gradient is dZ^T x fb on the matrix
// pipe, fb[m] = 2^(Tz - eZ[m]).  Per wave and stage: 4 LDS-DMA pieces, 16 transpose reads, 2 table reads, 16 v_pk_mul_f16, 12 MFMAs.
// The stage loop is unrolled over three 128-row blocks (24 stages: the period of (position in the block, stage buffer)).
// Partial slabs [batch slice][tile][128][128]; wgrad_h2i_reduce_kernel sums the slices in a fixed order.
#include "h2i_core.hpp"

namespace {

constexpr int TILE = 128;
constexpr int MAX_JOBS_H = 12;
typedef __fp16 h16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h16x8 __attribute__((ext_vector_type(8)));

struct HJob {
    const u32x4* dZ;        // image(M, N)
    const u32x4* X;         // image(M, K)
    const int* ez;          // exps of dZ: [row tile][kbs_n][128]
    const int* ex;
    u32 dz_bytes, x_bytes;
    float* dW;
    float* db;
    float* part;            // [splits][tiles][128][128]
    float* bpart;           // [splits][col_tiles][row_tiles][128]
    long long ldw;
    int N, K, col_tiles, row_tiles, st_n, st_k, kbs_n, kbs_k;
    int tile_end;           // running sum of tiles over the jobs
};
struct HGroup {
    int count, M, mtiles, rows_per_split, splits, tiles_total;
    HJob job[MAX_JOBS_H];
};

__global__ __launch_bounds__(256, 3) void wgrad_h2i_group_kernel(const HGroup G) {
    // separate objects per stage buffer: an LDS-DMA into one cannot alias the fragment reads of the other
    // (three stage buffers: the transfers run two stages ahead of the MFMAs, see linear_h2i_kernel)
    __shared__ __attribute__((aligned(16))) unsigned char A0[2][4096];
    __shared__ __attribute__((aligned(16))) unsigned char A1[2][4096];
    __shared__ __attribute__((aligned(16))) unsigned char A2[2][4096];
    __shared__ __attribute__((aligned(16))) unsigned char B0[2][4096];
    __shared__ __attribute__((aligned(16))) unsigned char B1[2][4096];
    __shared__ __attribute__((aligned(16))) unsigned char B2[2][4096];
    __shared__ __attribute__((aligned(16))) _Float16 Ft[2][128];      // f[m] of the block in flight / the next one
    __shared__ __attribute__((aligned(16))) _Float16 Fb[2][128];      // fb[m]
    __shared__ int Tt[2][2];                                          // (T, Tz) per table
    __shared__ int Tred[4][2];
#define AS(b) ((b) == 0 ? A0 : (b) == 1 ? A1 : A2)
#define BS(b) ((b) == 0 ? B0 : (b) == 1 ? B1 : B2)
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int split = xcd + 8 * (jb / G.tiles_total);
    if (split >= G.splits) return;
    int t = jb % G.tiles_total;
    int j = 0;
    while (j < G.count - 1 && t >= G.job[j].tile_end) ++j;
    const int tiles_j = G.job[j].tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
    if (j > 0) t -= G.job[j - 1].tile_end;
    const HJob& J = G.job[j];
    const int tr = t / J.col_tiles, tc = t - tr * J.col_tiles;
    const int m_begin = split * G.rows_per_split;                    // multiples of 128: a slice is a whole number of exponent blocks
    const int m_end = min(G.mtiles * 128, m_begin + G.rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- LDS-DMA geometry: lane t of the workgroup = piece (R = wave, Q, a, b) of the stage image, for A (dZ) and B (X) alike
    const int Q = (tid >> 4) & 3, a4 = (tid >> 2) & 3, b4 = tid & 3;
    const int mrow = 4 * wave + a4;                                          // batch row inside the stage (0..15)
    const u32 slot = (u32)(mrow * 2 + ((b4 & 1) ^ ((mrow >> 3) & 1))) * 16u;   // rslot(row, half) of the source chunk (stage rows are 16-aligned)
    const int sa = tr * 8 + 2 * Q + (b4 >> 1), sb = tc * 8 + 2 * Q + (b4 >> 1);   // source stage (16 columns) of this lane's piece
    const u32 aoff = sa < J.st_n ? (u32)sa * (u32)HI_CHUNK + slot : INVALID;  // columns behind the matrix: zeros land
    const u32 boff = sb < J.st_k ? (u32)sb * (u32)HI_CHUNK + slot : INVALID;
    const rsrc_t ares = make_rsrc_bytes(J.dZ, J.dz_bytes), bres = make_rsrc_bytes(J.X, J.x_bytes);
    auto load_stage = [&](auto nbc, int mb) {
        constexpr int nbuf = decltype(nbc)::value;
        // batch rows mb .. mb + 15 sit in row tile mb >> 7 at local rows (mb & 127) ..; rows >= m_end: nothing valid -> zeros
        const u32 ua = (u32)((mb >> 7) * J.st_n) * (u32)HI_CHUNK + (u32)(mb & 127) * 32u;
        const u32 ub = (u32)((mb >> 7) * J.st_k) * (u32)HI_CHUNK + (u32)(mb & 127) * 32u;
        const u32 dead = oob_mask(mb + mrow, m_end - 1);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_void*)&AS(nbuf)[p][wave * 1024], 16, aoff | dead, ua + p * HI_PLANE, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bres, (lds_void*)&BS(nbuf)[p][wave * 1024], 16, boff | dead, ub + p * HI_PLANE, 0, 0);
        }
    };

    // ---- the scale tables of a 128-row block, built by threads < 128 (one batch row each) in three steps that ride on the stages of the
    // block before it: (1) request the two exponents, (2) minima over the block -> Tred (per wave), (3) after a barrier: T, Tz and the
    // factors -> Ft / Fb / Tt of the table the next block reads
    int ez_n = HI_EZERO, ex_n = HI_EZERO;
    auto table_request = [&](int mb) {
        if (tid < 128) {
            const int mt = mb >> 7;
            const bool ok = mt < G.mtiles;
            ez_n = ok ? J.ez[((long long)mt * J.kbs_n + tr) * 128 + tid] : HI_EZERO;
            ex_n = ok ? J.ex[((long long)mt * J.kbs_k + tc) * 128 + tid] : HI_EZERO;
        }
    };
    auto table_minima = [&]() {
        if (tid < 128) {
            const bool live = ez_n != HI_EZERO && ex_n != HI_EZERO;
            int tm = live ? ez_n + ex_n : 0x7fffffff, tz = ez_n != HI_EZERO ? ez_n : 0x7fffffff;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                tm = min(tm, __shfl_xor(tm, off, 64));
                tz = min(tz, __shfl_xor(tz, off, 64));
            }
            if (lane == 0) {
                Tred[wave][0] = tm;
                Tred[wave][1] = tz;
            }
        }
    };
    auto table_finish = [&](int which, int t_prev, int tz_prev) {
        if (tid < 128) {
            int T = min(Tred[0][0], Tred[1][0]), Tz = min(Tred[0][1], Tred[1][1]);
            T = T == 0x7fffffff ? t_prev : T;           // a block without content keeps the scale (nothing to add, nothing to rescale)
            Tz = Tz == 0x7fffffff ? tz_prev : Tz;
            const bool live = ez_n != HI_EZERO && ex_n != HI_EZERO;
            const int d = live ? T - (ez_n + ex_n) : 0, dz = ez_n != HI_EZERO ? Tz - ez_n : 0;
            Ft[which][tid] = (_Float16)__builtin_ldexpf(1.0f, d < -30 ? -30 : d);
            Fb[which][tid] = (_Float16)__builtin_ldexpf(1.0f, dz < -30 ? -30 : dz);
            if (tid == 0) {
                Tt[which][0] = T;
                Tt[which][1] = Tz;
            }
        }
    };

    // ---- fragment geometry (ds_read_b64_tr_b16): 16-lane group g: k half g >> 1, 16-column sub-block g & 1 of the 32-column tile;
    // lane i of the group supplies chunk i (row i >> 2, columns 4 (i & 3) .. + 3) and receives column i, rows 0..3
    const int g = lane >> 4, i16 = lane & 15;
    const int khalf = g >> 1, nsub = g & 1;
    const int frag = ((i16 >> 2) * 4 + 2 * nsub + ((i16 & 3) >> 1)) * 16 + 8 * (i16 & 1);          // (a, b, 8-byte half) inside a sub-block
    // sub-block (R, Q) at ((R * 4 + Q) * 256) bytes; R = 2 khalf + r (r = 0, 1: rows 0..3 / 4..7 of the half)
    const int a_base = (2 * khalf * 4 + 2 * wr) * 256 + frag, b_base = (2 * khalf * 4 + 2 * wc) * 256 + frag;
    auto rd = [&](const unsigned char* plane, int off) {
        const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)(plane + off));
        const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)(plane + off + 4 * 256));
        return __builtin_bit_cast(f16x8, h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    };

    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    }

    const int KT = (m_end - m_begin + BK - 1) / BK;               // stages of this slice (a multiple of 8)
    const bool want_bias = J.db != nullptr;
    int t_cur = 0, tz_cur = 0;                                     // scale of the accumulators (T, Tz of the block in flight)
    if (KT > 0) {
        table_request(m_begin);
        load_stage(S0{}, m_begin);
        load_stage(S1{}, m_begin + BK);
        table_minima();
        __syncthreads();
        table_finish(0, 0, 0);
        __syncthreads();
        t_cur = Tt[0][0];
        tz_cur = Tt[0][1];
    }
    int bias_phase = 0;
    const int col_tiles_u = __builtin_amdgcn_readfirstlane(J.col_tiles);
    // ---- a stage, with its position inside the 128-row block (ph) and its buffer as compile-time constants: no branch on the phase, the table
    // reads and the transfers' row offsets are immediates, a block's validity is one scalar -- ~80 instructions per stage, 12-16 of them MFMAs
    // (round 6; with ph = kt % 8 and the buffer as run-time / per-trip values the stage had ~130 and five taken branches: -6..9 % for the launch
    // alone, -1.15 ms per bench step, same bits; profiles/r06_wgrad_unroll_*.txt).  Three blocks (24 stages) per trip: stage kt sits in buffer kt % 3.
    auto load_at = [&](auto nbc, u32 ua, u32 ub, u32 dead) {
        constexpr int nbuf = decltype(nbc)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_void*)&AS(nbuf)[p][wave * 1024], 16, aoff | dead, ua + p * HI_PLANE, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bres, (lds_void*)&BS(nbuf)[p][wave * 1024], 16, boff | dead, ub + p * HI_PLANE, 0, 0);
        }
    };
    struct Blk {
        int kt0, which;
        bool more;
        u32 ua, ub, dead, ua_n, ub_n, dead_n;       // byte offsets of this block's / the next block's row tile in the two images; INVALID behind the slice
    };
    auto stage = [&](auto bc, auto phc, const Blk& B) {
        constexpr int buf = decltype(bc)::value, ph = decltype(phc)::value;
        const int which = B.which;
        const bool bias = want_bias && bias_phase == tc;
        bias_phase = bias_phase + 1 == col_tiles_u ? 0 : bias_phase + 1;
        if constexpr (ph == 0) {
            if (B.kt0 > 0) {                                  // (uniform) first stage of a block: the accumulators change scale
                const int tn = Tt[which][0], tzn = Tt[which][1];
                const int d = tn - t_cur, dz = tzn - tz_cur;
                if (d != 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][jj][r] = __builtin_ldexpf(acc[i][jj][r], d);
                }
                if (dz != 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accb[i][r] = __builtin_ldexpf(accb[i][r], dz);
                }
                t_cur = tn;
                tz_cur = tzn;
            }
        }
        if constexpr (ph == 7) {
            if (B.more) table_finish(which ^ 1, t_cur, tz_cur);
        }
        if constexpr (ph == 6) {
            if (B.more) table_request(m_begin + (B.kt0 + 8) * BK);
        }
        const f16x8 fv = *reinterpret_cast<const f16x8*>(&Ft[which][ph * 16 + 8 * khalf]);
        const f16x8 fbv = *reinterpret_cast<const f16x8*>(&Fb[which][ph * 16 + 8 * khalf]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph < 6) load_at(std::integral_constant<int, (buf + 2) % 3>{}, B.ua + (ph + 2) * 512, B.ub + (ph + 2) * 512, B.dead);
        else load_at(std::integral_constant<int, (buf + 2) % 3>{}, B.ua_n + (ph - 6) * 512, B.ub_n + (ph - 6) * 512, B.dead_n);
        __builtin_amdgcn_sched_barrier(0);
        f16x8 a[2][2], b[2][2];
#pragma unroll
        for (int p = 1; p >= 0; --p) {
            a[0][p] = rd(AS(buf)[p], a_base);
            b[0][p] = rd(BS(buf)[p], b_base);
            a[1][p] = rd(AS(buf)[p], a_base + 256);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) b[1][p] = rd(BS(buf)[p], b_base + 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2; ++p) b[0][p] = b[0][p] * fv;
        // smallest terms first: lo hi', hi lo', hi hi'
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[0][0], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 2; ++p) b[1][p] = b[1][p] * fv;
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[0][1], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[0][0], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[1][0], acc[i][1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[1][1], acc[i][1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[1][0], acc[i][1], 0, 0, 0);
        if (bias && wc == 0) {                               // wave-uniform: this tile's share of the column sums of dZ (the wc = 1 waves hold the same rows)
#pragma unroll
            for (int p = 1; p >= 0; --p)
#pragma unroll
                for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][p], fbv, accb[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph == 6) {
            if (B.more) table_minima();                       // the exponents requested at the stage's head have long arrived
        }
        __builtin_amdgcn_s_waitcnt(0x0070 | 4);               // vmcnt(4), lgkmcnt(0), expcnt untouched
        __builtin_amdgcn_s_barrier();
    };
    auto block = [&](auto b0c, int kt0) {
        constexpr int b0 = decltype(b0c)::value;              // buffer of the block's first stage
        Blk B;
        B.kt0 = kt0;
        B.which = (kt0 >> 3) & 1;
        B.more = kt0 + 8 < KT;
        const int mb0 = m_begin + kt0 * BK;
        B.ua = (u32)((mb0 >> 7) * J.st_n) * (u32)HI_CHUNK;
        B.ub = (u32)((mb0 >> 7) * J.st_k) * (u32)HI_CHUNK;
        B.ua_n = B.ua + (u32)J.st_n * (u32)HI_CHUNK;
        B.ub_n = B.ub + (u32)J.st_k * (u32)HI_CHUNK;
        B.dead = mb0 >= m_end ? INVALID : 0u;                 // (slices are whole 128-row blocks)
        B.dead_n = mb0 + 128 >= m_end ? INVALID : 0u;
        stage(std::integral_constant<int, (b0 + 0) % 3>{}, std::integral_constant<int, 0>{}, B);
        stage(std::integral_constant<int, (b0 + 1) % 3>{}, std::integral_constant<int, 1>{}, B);
        stage(std::integral_constant<int, (b0 + 2) % 3>{}, std::integral_constant<int, 2>{}, B);
        stage(std::integral_constant<int, (b0 + 3) % 3>{}, std::integral_constant<int, 3>{}, B);
        stage(std::integral_constant<int, (b0 + 4) % 3>{}, std::integral_constant<int, 4>{}, B);
        stage(std::integral_constant<int, (b0 + 5) % 3>{}, std::integral_constant<int, 5>{}, B);
        stage(std::integral_constant<int, (b0 + 6) % 3>{}, std::integral_constant<int, 6>{}, B);
        stage(std::integral_constant<int, (b0 + 7) % 3>{}, std::integral_constant<int, 7>{}, B);
    };
    for (int kt0 = 0; kt0 < KT; kt0 += 24) {                  // KT is a multiple of 8; stage kt sits in buffer kt % 3
        block(std::integral_constant<int, 0>{}, kt0);
        if (kt0 + 8 < KT) block(std::integral_constant<int, 2>{}, kt0 + 8);
        if (kt0 + 16 < KT) block(std::integral_constant<int, 1>{}, kt0 + 16);
    }

    // ---- epilogue: accumulators (2^T x the sums) -> slab tile [128][128] in logical order (float4 rows through the wave's LDS patch)
    __syncthreads();
    const int half = lane >> 5, l31 = lane & 31;
    float* P = J.part + ((long long)split * tiles_j + t) * (TILE * TILE);
    float* patch = reinterpret_cast<float*>(wave < 2 ? &A0[wave][0] : &A1[wave - 2][0]);
    const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = __builtin_ldexpf(acc[i][jj][r], -t_cur);
            patch_put(patch, acc[i][jj], half, l31);
            float* q = P + (long long)((2 * wr + i) * 32 + prow) * TILE + (2 * wc + jj) * 32 + 4 * pc4;
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(q + (long long)(8 * p) * TILE) = patch_get(patch, prow + 8 * p, pc4);
        }
    if (want_bias && wc == 0 && l31 == 0) {                 // column 0 of accb holds the sums (every column is the same); rows = dZ's columns
        float* bp = J.bpart + (((long long)split * J.col_tiles + tc) * J.row_tiles + tr) * TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bp[(2 * wr + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = __builtin_ldexpf(accb[i][r], -tz_cur);
    }
#undef AS
#undef BS
}

// Sum of the batch slices in a fixed order; dW / db written once.  block = (tile, 8 rows); thread = (row, float4 of columns)
__global__ __launch_bounds__(256) void wgrad_h2i_reduce_kernel(const HGroup G) {
    int b = blockIdx.x;
    const int blocks_tiles = G.tiles_total * 16;
    if (b >= blocks_tiles) {                           // bias blocks: four per (job, row tile), 32 features each
        b -= blocks_tiles;
        const int quarter = b & 3;
        int j = 0, rt = b >> 2;
        while (j < G.count - 1 && rt >= G.job[j].row_tiles) { rt -= G.job[j].row_tiles; ++j; }
        const HJob& J = G.job[j];
        if (rt >= J.row_tiles || J.db == nullptr) return;
        // thread = (group g of 8, feature): group g sums the partials q = g, g + 8, ... of the (slice, column tile) pairs -- all its loads
        // in flight at once, a fixed order of additions --, then the eight group sums are added in order
        __shared__ float gsum[8][32];
        const int nl = quarter * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
        const int pairs = G.splits * J.col_tiles;
        const long long stride = (long long)J.row_tiles * TILE;               // between consecutive (slice, column tile) pairs
        const float* src = J.bpart + (long long)rt * TILE + nl;
        float acc = 0.f;
#pragma unroll 16
        for (int q = grp; q < pairs; q += 8) acc += src[(long long)q * stride];
        gsum[grp][threadIdx.x & 31] = acc;
        __syncthreads();
        const int n = rt * TILE + nl;
        if (grp == 0 && n < J.N) {
            float t = gsum[0][threadIdx.x];
#pragma unroll
            for (int g2 = 1; g2 < 8; ++g2) t += gsum[g2][threadIdx.x];
            J.db[n] = t;
        }
        return;
    }
    int t = b >> 4;
    const int rg = b & 15;
    int j = 0;
    while (j < G.count - 1 && t >= G.job[j].tile_end) ++j;
    const int tiles_j = G.job[j].tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
    if (j > 0) t -= G.job[j - 1].tile_end;
    const HJob& J = G.job[j];
    const int tr = t / J.col_tiles, tc = t - tr * J.col_tiles;
    const int rl = rg * 8 + (threadIdx.x >> 5), c4 = threadIdx.x & 31;
    const int n = tr * TILE + rl;
    if (n >= J.N) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(J.part + (long long)t * (TILE * TILE) + (long long)rl * TILE) + c4;
    const long long step = (long long)tiles_j * (TILE * TILE / 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 7 < G.splits; sp += 8) {               // eight slabs in flight per thread; the order of the additions is fixed
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(sp + u) * step];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[u][e];
    }
    for (; sp < G.splits; ++sp) {
        const f32x4 v0 = p[(long long)sp * step];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v0[e];
    }
    const int k = tc * TILE + 4 * c4;
    float* dst = J.dW + (long long)n * J.ldw + k;
    if (k + 4 <= J.K && (reinterpret_cast<unsigned long long>(dst) & 15ull) == 0) {
        *reinterpret_cast<f32x4*>(dst) = acc;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < J.K) dst[e] = acc[e];
    }
}

struct HPlan {
    HGroup dev;
    long long bytes;
    int red_blocks;
    double flop, algo_bytes;
};

int h2i_splits(int M, int tiles_total) {
    // 768 (round 5; 8 batch slices for the 61..70-tile buckets of the bench step): with both operands arriving by LDS-DMA the slices need
    // not be short to hide their loads -- 1024 / 1536 / 2048 workgroups: 50.3 / 50.1 / 50.2 ms per step on one box, 512 / 768 / 1024: 46.67 /
    // 46.70 / 46.94 on another (interleaved pairs; round 4's converting kernel wanted 1536) -- and a third of the partial slabs is a third
    // of the reduce traffic (again on round 6's kernel: 8 / 16 / 24 slices 46.3 / 46.8 / 47.2 ms per step, three interleaved rounds)
    constexpr int target = 768, cap = 24;
    int s = target / (tiles_total > 0 ? tiles_total : 1) / 8 * 8;
    if (s < 8) s = 8;
    if (s > cap) s = cap;
    const int max_s = (int)dtc::ceil_div(dtc::ceil_div(M, 128), 8) * 8;     // a slice holds at least one 128-row block
    if (s > max_s) s = max_s;
    return s;
}

int plan_h2i(const DtcWgradH2iJob* jobs, int count, int M, void* workspace, HPlan& P) {
    DTC_REQUIRE(jobs != nullptr && count >= 1 && count <= MAX_JOBS_H, "job count %d outside 1..%d", count, MAX_JOBS_H);
    DTC_REQUIRE(M > 0, "bad M=%d", M);
    HGroup& G = P.dev;
    G.count = count;
    G.M = M;
    G.mtiles = (int)hi_rtiles(M);
    int tiles = 0, row_tiles = 0;
    for (int j = 0; j < count; ++j) {
        const DtcWgradH2iJob& h = jobs[j];
        DTC_REQUIRE(h.N > 0 && h.K > 0 && h.dZimg && h.Ximg && h.dW && dtc::aligned16(h.dZimg) && dtc::aligned16(h.Ximg), "job %d: bad shape / null or unaligned pointer", j);
        DTC_REQUIRE(h.ldw >= h.K + h.wcol0 && h.wcol0 >= 0, "job %d: columns [%d, %d) outside the %lld-wide gradient", j, h.wcol0, h.wcol0 + h.K, (long long)h.ldw);
        HJob& d = G.job[j];
        d.dZ = (const u32x4*)h.dZimg;
        d.X = (const u32x4*)h.Ximg;
        d.st_n = (int)hi_stages(h.N);
        d.st_k = (int)hi_stages(h.K);
        d.kbs_n = (int)hi_kblocks(h.N);
        d.kbs_k = (int)hi_kblocks(h.K);
        DTC_REQUIRE(hi_bytes(M, h.N) < (1ll << 31) && hi_bytes(M, h.K) < (1ll << 31), "job %d: image beyond 2 GiB", j);
        d.dz_bytes = (u32)hi_data_bytes(M, h.N);
        d.x_bytes = (u32)hi_data_bytes(M, h.K);
        d.ez = reinterpret_cast<const int*>(reinterpret_cast<const char*>(h.dZimg) + d.dz_bytes);
        d.ex = reinterpret_cast<const int*>(reinterpret_cast<const char*>(h.Ximg) + d.x_bytes);
        d.dW = h.dW + h.wcol0;
        d.ldw = h.ldw;
        d.db = h.db;
        d.N = h.N;
        d.K = h.K;
        d.col_tiles = (int)dtc::ceil_div(h.K, TILE);
        d.row_tiles = (int)dtc::ceil_div(h.N, TILE);
        tiles += d.row_tiles * d.col_tiles;
        row_tiles += d.row_tiles;
        d.tile_end = tiles;
    }
    G.tiles_total = tiles;
    G.splits = h2i_splits(M, tiles);
    G.rows_per_split = (int)dtc::ceil_div(dtc::ceil_div(M, G.splits), 128) * 128;
    long long off = 0;
    P.flop = P.algo_bytes = 0.0;
    for (int j = 0; j < count; ++j) {
        HJob& d = G.job[j];
        const long long tiles_j = d.tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
        d.part = workspace ? (float*)((char*)workspace + off) : nullptr;
        off += (long long)G.splits * tiles_j * TILE * TILE * (long long)sizeof(float);
        d.bpart = workspace ? (float*)((char*)workspace + off) : nullptr;
        off += (long long)G.splits * d.col_tiles * d.row_tiles * TILE * (long long)sizeof(float);
        P.flop += 2.0 * M * (double)d.N * d.K;
        P.algo_bytes += 4.0 * ((double)M * d.N + (double)M * d.K) + 4.0 * (double)d.N * (d.K + 1);
    }
    P.bytes = off;
    P.red_blocks = tiles * 16 + 4 * row_tiles;
    return DTC_OK;
}

}  // namespace

extern "C" int64_t dtc_wgrad_group_h2i_workspace(const DtcWgradH2iJob* jobs, int count, int M) {
    HPlan P;
    if (plan_h2i(jobs, count, M, nullptr, P) != DTC_OK) return -1;
    return P.bytes;
}

extern "C" int dtc_wgrad_group_h2i(const DtcWgradH2iJob* jobs, int count, int M, void* workspace, void* stream) {
    DTC_REQUIRE(workspace != nullptr && dtc::aligned16(workspace), "wgrad group workspace must be a 16-byte aligned device buffer");
    HPlan P;
    int rc = plan_h2i(jobs, count, M, workspace, P);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const HGroup& G = P.dev;
    {
        dtc::ProfScope prof(dtc::prof_shape_name("linear_wgrad", M, G.tiles_total, count), P.flop, s, P.algo_bytes);
        hipLaunchKernelGGL(wgrad_h2i_group_kernel, dim3(G.tiles_total * 8 * (int)dtc::ceil_div(G.splits, 8)), dim3(256), 0, s, G);
    }
    {
        dtc::ProfScope prof(dtc::prof_shape_name("wgrad_reduce", G.splits, G.tiles_total, count), (double)P.bytes + P.bytes / (double)G.splits, s);
        hipLaunchKernelGGL(wgrad_h2i_reduce_kernel, dim3(P.red_blocks), dim3(256), 0, s, G);
    }
    return dtc::check_launch("wgrad_group_h2i");
}
