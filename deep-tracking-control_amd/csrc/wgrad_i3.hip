// Weight gradients from ACTIVATION IMAGES (round 4): dW [N, K] = dZ^T X and db = colsum(dZ) for the nn.Linear stacks of
// rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349 under loss.backward() (ppo.py:252, 333), with BOTH operands given as the
// images their producers wrote (include/dtc_hip.h, csrc/gemm_s3.hip: [128-row tile][16-column stage][plane 3][slot 256][16 bytes]).
//
// wgrad_s3_group_kernel converts both operands fp32 -> bf16 x 3 inside its K loop -- every dZ tile once per K / 128 column tiles,
// every X tile once per N / 128 row tiles (5.2 VALU per MFMA, plane stores with 4-way LDS conflicts: profiles/r03_gemm_pmc.md).
// Here the reduction index is the batch row, which is the ROW index of both images, so a stage (16 batch rows x 128 columns of each
// operand) is 8 x 512 contiguous bytes per plane in HBM.  LDS-DMA copies it as 16-byte pieces (8 columns of one batch row) into
//   stage image [R = row / 4][Q = column / 32][a = row % 4][b = (column / 8) % 4][16 bytes]            (4 KiB per plane and operand)
// -- the 1 KiB a wave's piece covers is four batch rows x 256 bytes of the source -- and the MFMA fragments (8 consecutive batch
// rows of ONE column per lane) come out of it by the hardware transpose read ds_read_b64_tr_b16: a 16-lane group reads one
// [4 rows][16 columns] block, lane i supplying the address of its i-th 8-byte chunk and receiving column i (tools/probes/tr16_dma.hip);
// the 32 lanes of a read pass cover one contiguous 256-byte sub-block (R, Q): conflict free.  Per wave and stage: 6 LDS-DMA pieces,
// 24 transpose reads, 24 MFMAs (+ the bias MFMAs below), no VALU work, no operand registers, no ds_write.
// Bias gradient: column sums of dZ as dZ^T x ones on the matrix pipe -- the workgroup of column tile tc does it in the stages
// kt % col_tiles == tc (6 extra MFMAs in 1 / col_tiles of its stages), so the work is spread evenly over the tiles of a row.
// Partial slabs in LOGICAL order, [batch slice][tile][128][128]; wgrad_i3_reduce_kernel sums the slices in a fixed order.
#include "s3_core.hpp"

namespace {

constexpr int TILE = 128;
constexpr int CHUNK = 3 * 128 * 32;         // bytes of one image chunk (row tile, stage)
constexpr int PLANE = 128 * 32;
constexpr int MAX_JOBS_I3 = 12;
typedef __attribute__((address_space(3))) void lds_void;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct I3Job {
    const u32x4* dZ;        // image(M, N)
    const u32x4* X;         // image(M, K)
    long long dz_bytes, x_bytes;
    float* dW;
    float* db;
    float* part;            // [splits][tiles][128][128]
    float* bpart;           // [splits][col_tiles][row_tiles][128]
    int N, K, col_tiles, row_tiles, st_n, st_k;
    int tile_end;           // running sum of tiles over the jobs
};
struct I3Group {
    int count, M, rows_per_split, splits, tiles_total;
    I3Job job[MAX_JOBS_I3];
};

__global__ __launch_bounds__(256, 3) void wgrad_i3_group_kernel(const I3Group G) {
    // separate objects per stage buffer: an LDS-DMA into one cannot alias the fragment reads of the other
    __shared__ __attribute__((aligned(16))) unsigned char A0[3][4096];
    __shared__ __attribute__((aligned(16))) unsigned char A1[3][4096];
    __shared__ __attribute__((aligned(16))) unsigned char B0[3][4096];
    __shared__ __attribute__((aligned(16))) unsigned char B1[3][4096];
#define AS(b) ((b) ? A1 : A0)
#define BS(b) ((b) ? B1 : B0)
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int split = xcd + 8 * (jb / G.tiles_total);
    if (split >= G.splits) return;
    int t = jb % G.tiles_total;
    int j = 0;
    while (j < G.count - 1 && t >= G.job[j].tile_end) ++j;
    const int tiles_j = G.job[j].tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
    if (j > 0) t -= G.job[j - 1].tile_end;
    const I3Job& J = G.job[j];
    const int tr = t / J.col_tiles, tc = t - tr * J.col_tiles;
    const int M = G.M;
    const int m_begin = split * G.rows_per_split;
    const int m_end = min(M, m_begin + G.rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- LDS-DMA geometry: lane t of the workgroup = piece (R = wave, Q, a, b) of the stage image, for A (dZ) and B (X) alike
    const int Q = (tid >> 4) & 3, a4 = (tid >> 2) & 3, b4 = tid & 3;
    const int mrow = 4 * wave + a4;                                          // batch row inside the stage (0..15)
    const u32 slot = (u32)(mrow * 2 + ((b4 & 1) ^ ((mrow >> 3) & 1))) * 16u;   // rslot(row, half) of the source chunk (stage rows are 16-aligned)
    const int sa = tr * 8 + 2 * Q + (b4 >> 1), sb = tc * 8 + 2 * Q + (b4 >> 1);   // source stage (16 columns) of this lane's piece
    const u32 aoff = sa < J.st_n ? (u32)sa * (u32)CHUNK + slot : INVALID;     // columns behind the matrix: zeros land
    const u32 boff = sb < J.st_k ? (u32)sb * (u32)CHUNK + slot : INVALID;
    const rsrc_t ares = make_rsrc_bytes(J.dZ, J.dz_bytes), bres = make_rsrc_bytes(J.X, J.x_bytes);
    auto load_stage = [&](auto nbc, int mb) {
        constexpr int nbuf = decltype(nbc)::value;
        // batch rows mb .. mb + 15 sit in row tile mb >> 7 at local rows (mb & 127) ..; rows >= m_end: nothing valid -> zeros
        const u32 ua = (u32)((mb >> 7) * J.st_n) * (u32)CHUNK + (u32)(mb & 127) * 32u;
        const u32 ub = (u32)((mb >> 7) * J.st_k) * (u32)CHUNK + (u32)(mb & 127) * 32u;
        const u32 dead = oob_mask(mb + mrow, m_end - 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_void*)&AS(nbuf)[p][wave * 1024], 16, aoff | dead, ua + p * PLANE, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bres, (lds_void*)&BS(nbuf)[p][wave * 1024], 16, boff | dead, ub + p * PLANE, 0, 0);
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
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(plane + off));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(plane + off + 4 * 256));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
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
    const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};

    auto stage = [&](auto bc, int mb_next, bool bias) {
        constexpr int buf = decltype(bc)::value;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};      // smallest terms first
        __builtin_amdgcn_sched_barrier(0);
        load_stage(std::integral_constant<int, buf ^ 1>{}, mb_next);              // the next stage's pieces first (see linear_i3_kernel)
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int p = 2; p >= 0; --p) {
            a[0][p] = rd(AS(buf)[p], a_base);
            b[0][p] = rd(BS(buf)[p], b_base);
            a[1][p] = rd(AS(buf)[p], a_base + 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[tt]], b[0][PB[tt]], acc[i][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) b[1][p] = rd(BS(buf)[p], b_base + 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 3; tt < 6; ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[tt]], b[0][PB[tt]], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[tt]], b[1][PB[tt]], acc[i][1], 0, 0, 0);
        if (bias && wc == 0) {                               // wave-uniform: this tile's share of the column sums of dZ (the wc = 1 waves hold the same rows)
#pragma unroll
            for (int p = 2; p >= 0; --p)
#pragma unroll
                for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][p], ones, accb[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };

    const int KT = (m_end - m_begin + BK - 1) / BK;
    const bool want_bias = J.db != nullptr;
    if (KT > 0) {
        load_stage(S0{}, m_begin);
        __syncthreads();
        int kt = 0;
        // two stages per trip (constant buffer indices); a request past the slice's last stage lands zeros (rows >= m_end are dead)
        for (; kt < KT; kt += 2) {
            stage(S0{}, m_begin + (kt + 1) * BK, want_bias && (kt % J.col_tiles) == tc);
            stage(S1{}, m_begin + (kt + 2) * BK, want_bias && ((kt + 1) % J.col_tiles) == tc);
        }
    }

    // ---- epilogue: accumulators -> slab tile [128][128] in logical order (float4 rows through the wave's LDS patch)
    __syncthreads();
    const int half = lane >> 5, l31 = lane & 31;
    float* P = J.part + ((long long)split * tiles_j + t) * (TILE * TILE);
    float* patch = reinterpret_cast<float*>(wave < 3 ? &A0[wave][0] : &A1[0][0]);
    const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
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
            for (int r = 0; r < 16; ++r) bp[(2 * wr + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = accb[i][r];
    }
#undef AS
#undef BS
}

// Sum of the batch slices in a fixed order; dW / db written once.  block = (tile, 8 rows); thread = (row, float4 of columns)
__global__ __launch_bounds__(256) void wgrad_i3_reduce_kernel(const I3Group G) {
    int b = blockIdx.x;
    const int blocks_tiles = G.tiles_total * 16;
    if (b >= blocks_tiles) {                           // bias blocks: one per (job, row tile)
        b -= blocks_tiles;
        int j = 0, rt = b;
        while (j < G.count - 1 && rt >= G.job[j].row_tiles) { rt -= G.job[j].row_tiles; ++j; }
        const I3Job& J = G.job[j];
        if (rt >= J.row_tiles || J.db == nullptr || threadIdx.x >= TILE) return;
        const int n = rt * TILE + threadIdx.x;
        if (n >= J.N) return;
        float s = 0.f;
        for (int sp = 0; sp < G.splits; ++sp)
            for (int c = 0; c < J.col_tiles; ++c) s += J.bpart[(((long long)sp * J.col_tiles + c) * J.row_tiles + rt) * TILE + threadIdx.x];
        J.db[n] = s;
        return;
    }
    int t = b >> 4;
    const int rg = b & 15;
    int j = 0;
    while (j < G.count - 1 && t >= G.job[j].tile_end) ++j;
    const int tiles_j = G.job[j].tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
    if (j > 0) t -= G.job[j - 1].tile_end;
    const I3Job& J = G.job[j];
    const int tr = t / J.col_tiles, tc = t - tr * J.col_tiles;
    const int rl = rg * 8 + (threadIdx.x >> 5), c4 = threadIdx.x & 31;
    const int n = tr * TILE + rl;
    if (n >= J.N) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(J.part + (long long)t * (TILE * TILE) + (long long)rl * TILE) + c4;
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
    const int k = tc * TILE + 4 * c4;
    float* dst = J.dW + (long long)n * J.K + k;
    if (k + 4 <= J.K && (J.K & 3) == 0 && (reinterpret_cast<unsigned long long>(J.dW) & 15ull) == 0) {
        *reinterpret_cast<f32x4*>(dst) = acc;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < J.K) dst[e] = acc[e];
    }
}

struct I3Plan {
    I3Group dev;
    long long bytes;
    int red_blocks;
    double flop, algo_bytes;
};

int split_cap() {
    static const int cap = [] {
        const char* e = getenv("DTC_WGRAD_SPLIT_CAP");
        return e ? atoi(e) : 24;
    }();
    return cap;
}

int i3_splits(int M, int tiles_total) {
    static const char* target_env = getenv("DTC_WGRAD_I3_BLOCKS");
    const int target = target_env ? atoi(target_env) : 768;     // one residency round of 3 workgroups per CU (as wgrad_s3_group_kernel)
    int s = target / (tiles_total > 0 ? tiles_total : 1) / 8 * 8;
    if (s < 8) s = 8;
    if (s > split_cap()) s = split_cap();               // small groups: bounded slab traffic (64 KiB per tile and slice, written and re-read)
    const int max_s = (int)dtc::ceil_div(dtc::ceil_div(M, BK * 8), 8) * 8;
    if (s > max_s) s = max_s;
    return s;
}

int plan_i3(const DtcWgradImgJob* jobs, int count, int M, void* workspace, I3Plan& P) {
    DTC_REQUIRE(jobs != nullptr && count >= 1 && count <= MAX_JOBS_I3, "job count %d outside 1..%d", count, MAX_JOBS_I3);
    DTC_REQUIRE(M > 0, "bad M=%d", M);
    I3Group& G = P.dev;
    G.count = count;
    G.M = M;
    int tiles = 0, row_tiles = 0;
    for (int j = 0; j < count; ++j) {
        const DtcWgradImgJob& h = jobs[j];
        DTC_REQUIRE(h.N > 0 && h.K > 0 && h.dZimg && h.Ximg && h.dW && dtc::aligned16(h.dZimg) && dtc::aligned16(h.Ximg), "job %d: bad shape / null or unaligned pointer", j);
        I3Job& d = G.job[j];
        d.dZ = (const u32x4*)h.dZimg;
        d.X = (const u32x4*)h.Ximg;
        d.st_n = (int)dtc::ceil_div(h.N, BK);
        d.st_k = (int)dtc::ceil_div(h.K, BK);
        d.dz_bytes = (long long)CHUNK * dtc::ceil_div(M, 128) * d.st_n;
        d.x_bytes = (long long)CHUNK * dtc::ceil_div(M, 128) * d.st_k;
        DTC_REQUIRE(d.dz_bytes < (1ll << 31) && d.x_bytes < (1ll << 31), "job %d: image beyond 2 GiB", j);
        d.dW = h.dW;
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
    G.splits = i3_splits(M, tiles);
    G.rows_per_split = (int)dtc::ceil_div(dtc::ceil_div(M, G.splits), BK) * BK;
    long long off = 0;
    P.flop = P.algo_bytes = 0.0;
    for (int j = 0; j < count; ++j) {
        I3Job& d = G.job[j];
        const long long tiles_j = d.tile_end - (j > 0 ? G.job[j - 1].tile_end : 0);
        d.part = workspace ? (float*)((char*)workspace + off) : nullptr;
        off += (long long)G.splits * tiles_j * TILE * TILE * (long long)sizeof(float);
        d.bpart = workspace ? (float*)((char*)workspace + off) : nullptr;
        off += (long long)G.splits * d.col_tiles * d.row_tiles * TILE * (long long)sizeof(float);
        P.flop += 2.0 * M * (double)d.N * d.K;
        P.algo_bytes += 6.0 * ((double)M * d.N + (double)M * d.K) + 4.0 * (double)d.N * (d.K + 1);
    }
    P.bytes = off;
    P.red_blocks = tiles * 16 + row_tiles;
    return DTC_OK;
}

}  // namespace

extern "C" int64_t dtc_wgrad_group_i3_workspace(const DtcWgradImgJob* jobs, int count, int M) {
    I3Plan P;
    if (plan_i3(jobs, count, M, nullptr, P) != DTC_OK) return -1;
    return P.bytes;
}

// dW_j = dZ_j^T X_j, db_j = colsum(dZ_j) for `count` layers in ONE launch pair, both operands of every layer given as activation
// images over the same M batch rows (include/dtc_hip.h)
extern "C" int dtc_wgrad_group_i3(const DtcWgradImgJob* jobs, int count, int M, void* workspace, void* stream) {
    DTC_REQUIRE(workspace != nullptr && dtc::aligned16(workspace), "wgrad group workspace must be a 16-byte aligned device buffer");
    I3Plan P;
    int rc = plan_i3(jobs, count, M, workspace, P);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const I3Group& G = P.dev;
    {
        dtc::ProfScope prof(dtc::prof_shape_name("linear_wgrad", M, G.tiles_total, count), P.flop, s, P.algo_bytes);
        hipLaunchKernelGGL(wgrad_i3_group_kernel, dim3(G.tiles_total * 8 * (int)dtc::ceil_div(G.splits, 8)), dim3(256), 0, s, G);
    }
    {
        dtc::ProfScope prof(dtc::prof_shape_name("wgrad_reduce", G.splits, G.tiles_total, count), (double)P.bytes + P.bytes / (double)G.splits, s);
        hipLaunchKernelGGL(wgrad_i3_reduce_kernel, dim3(P.red_blocks), dim3(256), 0, s, G);
    }
    return dtc::check_launch("wgrad_group_i3");
}
