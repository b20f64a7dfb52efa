// Weight gradients of the fp32 MFMA dense layers (gfx950): dW[N,K] = dZ[M,N]^T X[M,K], db[N] = column sums of dZ --
// the `loss.backward()` share of nn.Linear in rsl_rl/rsl_rl/algorithms/ppo.py:252, 333 (all layers of
// rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349).  Same block loop as csrc/gemm.hip (see its header).
//
// The reduction runs over the mini-batch (M = 24576 rows) while the output is small (<= 693 x 752), so the batch is
// split: block (tile, split) writes the partial product of its batch slice into a slab, and a reduce kernel adds the
// slabs in a fixed order (deterministic, no atomics).  Two entry points:
//   dtc_linear_wgrad : one layer = one partial launch + one reduce launch (used by the recurrent trainers and tests);
//   dtc_wgrad_group  : ALL layers of a gradient bucket in ONE partial launch + ONE reduce launch.  Weight gradients are
//       only needed by the optimiser (and the data-parallel exchange) at the end of the backward pass, so PPO.update
//       queues them and flushes a bucket when its last data gradient has been issued: 26 + 26 launches per mini-batch
//       become 4 + 4, every block of the launch has the same length (uniform batch slices over all layers), and the
//       many-short-blocks launches of the narrow layers (128x265 ... 1x128: 15-40 us each for < 2 GFLOP) disappear.
#include "gemm_core.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// Weight gradient (split over the batch): part[s][n][c] = sum_{m in split s} dZ[m,n] X[m,c],
// c == K holds the bias-gradient partial.  A second kernel reduces the splits.
// ------------------------------------------------------------------------------------------
// Column tiles are aligned to the segments of X (tile = (segment, tile inside the segment)), so every block reads ONE
// source matrix.  Both operands have the reduction index (the batch row m) as their ROW and are contiguous along the
// output dimensions: float4 loads along n (dZ) and c (X), LDS images [m][n] / [m][c], ds_read_b32 fragments.  Elements a
// float4 drags in from beyond N / beyond the segment only feed output rows / columns that are never stored (behind the
// last row of a source the descriptor returns 0), so the loads carry row masks only.  The gathered row index of X is a
// per-lane 8-byte load issued one stage ahead of the tile it addresses.
// MFMA phase with INTERLEAVED column tiles (BN = 64: the wave's two 32 x 32 tiles own the even and the odd columns of its
// 64-column strip).  The B operand of lane (column l, k half h) for k-step kp is then Bs[2 kp + h][2 l] and [2 l + 1]: ONE
// ds_read_b64 feeds both tiles (32 lanes x 8 bytes = 256 contiguous bytes: conflict free) instead of two ds_read_b32 -- 16
// instead of 24 LDS reads per 16 MFMAs, no transposition anywhere; which output column an accumulator register belongs to
// is a matter of the epilogue's addressing only, every output element still sums its products in the same order.
__device__ __forceinline__ void mfma_step_ilv(const float* __restrict__ As, const float* __restrict__ Bs,
                                              f32x16 (&acc)[1][2], int lane, int wm_off) {
    using C = Cfg<64>;
    const int half = lane >> 5, l31 = lane & 31;
    const float* ap = As + half * C::LDA + wm_off + l31;
    const float* bp = Bs + half * C::LDB + 2 * l31;
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
        const float a = ap[2 * kp * C::LDA];
        const float2 b = *reinterpret_cast<const float2*>(&bp[2 * kp * C::LDB]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc[0][1], 0, 0, 0);
    }
}

template <int BN, bool ILV = false>
__device__ __forceinline__ void wgrad_block(const float* __restrict__ dZ, long long lddz, const SegMatDev& X,
                                            float* __restrict__ part, int M, int N, int K, int rows_per_split,
                                            int col_tiles, int split, int t, float (*As)[BK][Cfg<BN>::LDA],
                                            float (*Bs)[BK][Cfg<BN>::LDB]) {
    using C = Cfg<BN>;
    const int tr = t / col_tiles;
    int tc = t - tr * col_tiles;
    // (segment, local tile) of this column tile
    int seg = 0;
    for (; seg < X.nseg - 1; ++seg) {
        const int nt = (X.s[seg].width + BN - 1) / BN;
        if (tc < nt) break;
        tc -= nt;
    }
    const SegDev sd = X.s[seg];
    const int n0 = tr * BM, lc0 = tc * BN;           // lc0: first column inside the segment
    const int m_begin = split * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = (wave / C::WN) * (32 * C::TM), wn_off = (wave % C::WN) * (32 * C::TN);

    // A loader: As[m][n0 + 4*an4 ..] = dZ[m][...], batch rows am and am + 8 of the stage
    const int am = tid >> 5, an4 = 4 * (tid & 31);
    constexpr int NA = BK / 8;
    u32 aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) aoff[i] = n0 + an4 < N ? (u32)((long long)(am + 8 * i) * lddz + n0 + an4) * 4u : INVALID;
    // B loader: Bs[m][4*bc4 ..] = X[row(m)][col0 + lc0 + 4*bc4 ..], one batch row per thread and stage
    constexpr int C4 = BN / 4;
    const bool bthread = tid < BK * C4;
    const int bm = tid / C4, bc = 4 * (tid % C4);
    const u32 ldb = (u32)sd.ld * 4u;
    const u32 bcolb = (bthread && lc0 + bc < sd.width) ? (u32)(sd.col0 + lc0 + bc) * 4u : INVALID;
    const rsrc_t ares = make_rsrc_bytes(dZ, (long long)M * lddz * 4), bres = make_rsrc_bytes(sd.ptr, (long long)sd.rows * sd.ld * 4);

    f32x4 ra[NA], rb;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    auto row_of = [&](int m) -> u32 {                // source row of batch row m (clamped into the split: masked lanes)
        const int mc = m < m_end ? m : m_end - 1;
        return sd.gather ? (u32)X.idx[mc] : (u32)mc;
    };
    u32 rnext = row_of(m_begin + bm);                // row index of the first stage
    auto load_tile = [&](auto masked, int mb) {
        constexpr bool MK = decltype(masked)::value;
        const u32 sa = (u32)mb * (u32)lddz * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload4(ares, aoff[i] | (MK ? oob_mask(mb + am + 8 * i, m_end - 1) : 0u), sa);
        rb = bload4(bres, (bcolb + rnext * ldb) | (MK ? oob_mask(mb + bm, m_end - 1) : 0u), 0u);
        rnext = row_of(mb + BK + bm);                // for the next stage (a clamped re-read past the split's end)
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            *reinterpret_cast<f32x4*>(&As[buf][am + 8 * i][an4]) = ra[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[e] += ra[i][e];
        }
        if (BN >= 64 || bthread) *reinterpret_cast<f32x4*>(&Bs[buf][bm][bc]) = rb;
    };

    f32x16 acc[C::TM][C::TN];
    zero_acc<BN>(acc);

    const int KT = (m_end - m_begin + BK - 1) / BK;
    if (KT > 0) {                                   // uniform per block (an empty trailing split writes zeros)
        int buf = 0;
        auto mfma = [&](int b) {
            if constexpr (ILV) mfma_step_ilv(&As[b][0][0], &Bs[b][0][0], acc, lane, wm_off);
            else mfma_step<BN>(&As[b][0][0], &Bs[b][0][0], acc, lane, wm_off, wn_off);
        };
        auto step = [&](auto masked, int kt_next) {
            load_tile(masked, m_begin + kt_next * BK);
            mfma(buf);
            store_tile(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        };
        load_tile(Masked{}, m_begin);
        store_tile(0);
        __syncthreads();
        for (int kt = 1; kt + 1 < KT; ++kt) step(Full{}, kt);  // branch-free steady state (full tiles)
        if (KT > 1) step(Masked{}, KT - 1);                    // batch tail of the split
        mfma(buf);
    }

    const long long ldp = part_ld(K);
    float* P = part + (long long)split * N * ldp;
    const int half = lane >> 5, l31 = lane & 31;
    const bool bias_block = seg == 0 && tc == 0;
    __syncthreads();                                // every wave is past its last operand read (LDS is reused below)
    if constexpr (ILV) {
        // interleaved tiles: lane (l, half) holds columns 2 l and 2 l + 1 of its 64-column strip for 16 rows
        const int cl = lc0 + 2 * l31;
        if (n0 + BM <= N && lc0 + BN <= sd.width && ((sd.start + lc0) & 1) == 0) {
            float* q = P + (long long)(n0 + wm_off + 4 * half) * ldp + sd.start + cl;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<float2*>(q + (long long)((r & 3) + 8 * (r >> 2)) * ldp) = make_float2(acc[0][0][r], acc[0][1][r]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (cl + j >= sd.width) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = n0 + wm_off + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < N) P[(long long)row * ldp + sd.start + cl + j] = acc[0][j][r];
                }
            }
        }
    } else
    if (n0 + BM <= N && lc0 + BN <= sd.width && ((sd.start + lc0) & 3) == 0) {
        // wide stores: each wave transposes its 32x32 tiles through a private LDS patch -> dwordx4 rows of the slab
        float* patch = &As[0][0][0] + wave * (32 * LDW);     // 4 x 4 KiB inside the A stage buffers
        const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
                patch_put(patch, acc[i][j], half, l31);
                float* q = P + (long long)(n0 + wm_off + 32 * i + prow) * ldp + sd.start + lc0 + wn_off + 32 * j + 4 * pc4;
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(q + (long long)(8 * p) * ldp) = patch_get(patch, prow + 8 * p, pc4);
            }
    } else {
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
            const int lcol = lc0 + wn_off + 32 * j + l31;
            if (lcol >= sd.width) continue;
            const int col = sd.start + lcol;
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = n0 + wm_off + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < N) P[(long long)row * ldp + col] = acc[i][j][r];
                }
            }
        }
    }
    if (bias_block) {   // bias-gradient partial: 8 threads (batch-row groups) staged each group of 4 dZ columns
        float* red = &Bs[0][0][0];                   // [8][128] floats, disjoint from the patches in As
        *reinterpret_cast<f32x4*>(&red[am * BM + an4]) = bias4;
        __syncthreads();
        if (tid < BM && n0 + tid < N) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += red[g * BM + tid];
            P[(long long)(n0 + tid) * ldp + K] = s;
        }
    }
}

template <int BN>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dZ, long long lddz,
                                                           const SegMatDev X, float* __restrict__ part, int M, int N,
                                                           int K, int rows_per_split, int col_tiles, int splits) {
    using C = Cfg<BN>;
    __shared__ float As[2][BK][C::LDA];
    __shared__ float Bs[2][BK][C::LDB];
    const int tiles = ((N + BM - 1) / BM) * col_tiles;
    // block b runs on XCD b%8: every XCD owns whole batch slices (splits), so each slice of dZ / X is
    // pulled from HBM into ONE L2 and shared there by all output tiles
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int split = xcd + 8 * (jb / tiles);
    if (split >= splits) return;                    // padding block of the last (partial) group of 8 splits
    wgrad_block<BN>(dZ, lddz, X, part, M, N, K, rows_per_split, col_tiles, split, jb % tiles, As, Bs);
}

// ---- grouped launch: the layers of one gradient bucket ------------------------------------------------------------
constexpr int MAX_JOBS = 12;
struct WJobDev {
    const float* dZ;
    long long lddz;
    SegMatDev X;
    float* part;            // [splits][N][part_ld(K)]
    float* dW;
    float* db;
    int N, K, col_tiles;
    int tile_end;           // running sum of tiles over the jobs (job j owns tiles [tile_end[j-1], tile_end[j]))
    int red_end;            // running sum of reduce blocks
};
struct WGroupDev {
    int count, M, rows_per_split, splits, tiles_total;
    WJobDev job[MAX_JOBS];
};

// block b -> XCD b%8 -> batch slice (split) xcd + 8*(j / tiles_total), tile j % tiles_total of the job list: all tiles of
// all layers that read the same batch slice run on one XCD back to back (dZ of layer l is X-adjacent data of layer l-1 ...)
template <bool ILV>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const WGroupDev G) {
    using C = Cfg<64>;
    __shared__ float As[2][BK][C::LDA];
    __shared__ float Bs[2][BK][C::LDB];
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int split = xcd + 8 * (jb / G.tiles_total);
    if (split >= G.splits) return;
    int t = jb % G.tiles_total;
    int j = 0;
    while (j < G.count - 1 && t >= G.job[j].tile_end) ++j;
    if (j > 0) t -= G.job[j - 1].tile_end;
    const WJobDev& J = G.job[j];
    wgrad_block<64, ILV>(J.dZ, J.lddz, J.X, J.part, G.M, J.N, J.K, G.rows_per_split, J.col_tiles, split, t, As, Bs);
}

// Sum of the split partials in a FIXED order (deterministic): block = 64 float4 columns x G split groups; group g
// adds splits g, g+G, g+2G, ... (4 loads in flight), the groups are then added in order 0..G-1 through LDS.
template <int G>
__global__ __launch_bounds__(64 * G) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW,
                                                              float* __restrict__ db, int N, int K, int splits) {
    __shared__ float4 red[G][64];
    const int ldp = part_ld(K);
    const long long total = (long long)N * ldp;
    const int n = blockIdx.y;
    const int c4 = blockIdx.x * 64 + threadIdx.x;          // float4 column
    const int g = threadIdx.y;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 * 4 < ldp) {
        const float4* p = reinterpret_cast<const float4*>(part + (long long)n * ldp) + c4;
        const long long step = total / 4;
        int s = g;
        for (; s + 3 * G < splits; s += 4 * G) {
            const float4 v0 = p[(long long)s * step], v1 = p[(long long)(s + G) * step];
            const float4 v2 = p[(long long)(s + 2 * G) * step], v3 = p[(long long)(s + 3 * G) * step];
            acc.x = (((acc.x + v0.x) + v1.x) + v2.x) + v3.x;
            acc.y = (((acc.y + v0.y) + v1.y) + v2.y) + v3.y;
            acc.z = (((acc.z + v0.z) + v1.z) + v2.z) + v3.z;
            acc.w = (((acc.w + v0.w) + v1.w) + v2.w) + v3.w;
        }
        for (; s < splits; s += G) {
            const float4 v = p[(long long)s * step];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[g][threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && c4 * 4 < ldp) {
        for (int j = 1; j < G; ++j) {
            const float4 v = red[j][threadIdx.x];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const float out[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c4 * 4 + i;
            if (c < K) dW[(long long)n * K + c] = out[i];
            else if (c == K && db) db[n] = out[i];
        }
    }
}
// The same sum for every layer of a grouped launch (one launch per bucket): block -> (job, output row n, 64 float4 columns)
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(const WGroupDev G) {
    constexpr int GR = 4;
    __shared__ float4 red[GR][64];
    int b = blockIdx.x, j = 0;
    while (j < G.count - 1 && b >= G.job[j].red_end) ++j;
    if (j > 0) b -= G.job[j - 1].red_end;
    const WJobDev& J = G.job[j];
    const int N = J.N, K = J.K, splits = G.splits;
    const int ldp = part_ld(K);
    const int chunks = (ldp / 4 + 63) / 64;
    const int n = b / chunks;
    const int c4 = (b - n * chunks) * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const long long total = (long long)N * ldp;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 * 4 < ldp) {
        const float4* p = reinterpret_cast<const float4*>(J.part + (long long)n * ldp) + c4;
        const long long step = total / 4;
        int s = g;
        for (; s + 3 * GR < splits; s += 4 * GR) {
            const float4 v0 = p[(long long)s * step], v1 = p[(long long)(s + GR) * step];
            const float4 v2 = p[(long long)(s + 2 * GR) * step], v3 = p[(long long)(s + 3 * GR) * step];
            acc.x = (((acc.x + v0.x) + v1.x) + v2.x) + v3.x;
            acc.y = (((acc.y + v0.y) + v1.y) + v2.y) + v3.y;
            acc.z = (((acc.z + v0.z) + v1.z) + v2.z) + v3.z;
            acc.w = (((acc.w + v0.w) + v1.w) + v2.w) + v3.w;
        }
        for (; s < splits; s += GR) {
            const float4 v = p[(long long)s * step];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c4 * 4 < ldp) {
        for (int q = 1; q < GR; ++q) {
            const float4 v = red[q][threadIdx.x];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const float out[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c4 * 4 + i;
            if (c < K) J.dW[(long long)n * K + c] = out[i];
            else if (c == K && J.db) J.db[n] = out[i];
        }
    }
}

// wgrad column-tile width: 64 measured at least as fast as 128 on every layer of this model (sweep in
// tools/microbench.py wgrad)
int pick_bn(int cols) {
    return cols <= 32 ? 32 : 64;
}

// wgrad split heuristic -------------------------------------------------------------------------------------
int wgrad_splits(int M, int tiles) {
    constexpr int target = 1024;
    // whole splits per XCD (multiple of 8) measured 10-20 % faster than filling the wave with an arbitrary count
    // (44 tiles x 23 splits = 1012 blocks ran slower than 44 x 16 = 704)
    int s = target / tiles / 8 * 8;
    if (s < 8) s = 8;
    const int max_s = (int)dtc::ceil_div(dtc::ceil_div(M, BK * 8), 8) * 8;
    if (s > max_s) s = max_s;
    return s;
}
// upper bound over every segmentation of X (segment-aligned column tiles only add tiles -> fewer splits)
int wgrad_splits_bound(int M, int N, int K) {
    return wgrad_splits(M, (int)(dtc::ceil_div(N, BM) * dtc::ceil_div(K, pick_bn(K))));
}

// grouped launch: one split count for all jobs (every block then reduces the same number of batch rows)
int group_splits(int M, int tiles_total) {
    constexpr int target = 3072;      // sweep 1024 ... 4096 with bench.py: 3072 best (24 batch slices per layer)
    int s = target / (tiles_total > 0 ? tiles_total : 1) / 8 * 8;
    if (s < 8) s = 8;
    const int max_s = (int)dtc::ceil_div(dtc::ceil_div(M, BK * 8), 8) * 8;
    if (s > max_s) s = max_s;
    return s;
}

struct GroupPlan {
    WGroupDev dev;
    long long bytes;
    int red_blocks;
    double flop, algo_bytes;   // algo_bytes: dZ, X read once, dW + db written once (the partial slabs are overhead)
};

int plan_group(const DtcWgradJob* jobs, int count, int M, void* workspace, GroupPlan& P) {
    DTC_REQUIRE(jobs != nullptr && count >= 1 && count <= MAX_JOBS, "job count %d outside 1..%d", count, MAX_JOBS);
    DTC_REQUIRE(M > 0, "bad M=%d", M);
    WGroupDev& G = P.dev;
    G.count = count;
    G.M = M;
    int tiles = 0;
    for (int j = 0; j < count; ++j) {
        const DtcWgradJob& h = jobs[j];
        DTC_REQUIRE(h.N > 0 && h.K > 0 && h.lddz >= h.N, "job %d: bad shape N=%d K=%d lddz=%lld", j, h.N, h.K, (long long)h.lddz);
        DTC_REQUIRE(h.dZ && h.dW, "job %d: null pointer", j);
        DTC_REQUIRE(h.dz_rows == 0, "job %d: a row map for dZ (dz_rows) is a feature of the split-precision path (dtc_wgrad_group_s3)", j);
        DTC_REQUIRE((long long)M * h.lddz <= MAX_ELEMS, "job %d: matrix too large", j);
        WJobDev& d = G.job[j];
        int rc = to_dev(&h.X, d.X, h.K, false, M);
        if (rc != DTC_OK) return rc;
        d.dZ = h.dZ;
        d.lddz = h.lddz;
        d.dW = h.dW;
        d.db = h.db;
        d.N = h.N;
        d.K = h.K;
        d.col_tiles = 0;
        for (int i = 0; i < d.X.nseg; ++i) d.col_tiles += (int)dtc::ceil_div(d.X.s[i].width, 64);
        tiles += (int)dtc::ceil_div(h.N, BM) * d.col_tiles;
        d.tile_end = tiles;
    }
    G.tiles_total = tiles;
    G.splits = group_splits(M, tiles);
    G.rows_per_split = (int)dtc::ceil_div(dtc::ceil_div(M, G.splits), BK) * BK;
    long long off = 0;
    int red = 0;
    P.flop = P.algo_bytes = 0.0;
    for (int j = 0; j < count; ++j) {
        WJobDev& d = G.job[j];
        d.part = workspace ? (float*)((char*)workspace + off) : nullptr;
        const long long ldp = part_ld(d.K);
        off += (((long long)G.splits * d.N * ldp * (long long)sizeof(float)) + 15) & ~15ll;
        red += d.N * (int)dtc::ceil_div(ldp / 4, 64);
        d.red_end = red;
        P.flop += 2.0 * M * (double)d.N * d.K;
        P.algo_bytes += 4.0 * ((double)M * d.N + (double)M * d.K + (double)d.N * (d.K + 1));
    }
    P.bytes = off;
    P.red_blocks = red;
    return DTC_OK;
}

}  // namespace

namespace {
// one layer as a job of the split-precision grouped kernel (a single plain segment; callers with segmented X pass their own)
DtcWgradJob one_job(const float* dZ, int64_t lddz, const DtcSegMat* X, float* dW, float* db, int N, int K) {
    DtcWgradJob j;
    j.dZ = dZ;
    j.lddz = lddz;
    j.X = *X;
    j.dW = dW;
    j.db = db;
    j.N = N;
    j.K = K;
    j.dz_rows = 0;
    return j;
}
// worst case of a one-layer split launch: every segment of X (<= 4) may add a partial 128-column tile
int64_t s3_one_layer_bound(int M, int N, int K) {
    const int64_t tiles = dtc::ceil_div(N, 128) * (dtc::ceil_div(K, 128) + 3);
    int64_t splits = 2048 / (tiles > 0 ? tiles : 1) / 8 * 8;
    if (splits < 8) splits = 8;
    const int64_t max_s = dtc::ceil_div(dtc::ceil_div(M, 16 * 8), 8) * 8;
    if (splits > max_s) splits = max_s;
    return splits * (tiles * 128 * 128 + dtc::ceil_div(N, 128) * 128) * (int64_t)sizeof(float) + 64;
}
}  // namespace

extern "C" int64_t dtc_linear_wgrad_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int64_t a = (int64_t)wgrad_splits_bound(M, N, K) * N * part_ld(K) * (int64_t)sizeof(float);
    const int64_t b = dtc_get_gemm_split() ? s3_one_layer_bound(M, N, K) : 0;
    return a > b ? a : b;
}

extern "C" int dtc_linear_wgrad(const float* dZ, int64_t lddz, const DtcSegMat* X, float* dW, float* db, void* workspace,
                                int M, int N, int K, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddz >= N, "bad shape");
    DTC_REQUIRE(dZ && dW && workspace, "null pointer");
    DTC_REQUIRE(dtc::aligned16(workspace), "wgrad workspace must be 16-byte aligned");
    DTC_REQUIRE((long long)M * lddz <= MAX_ELEMS, "matrix too large");
    if (dtc_get_gemm_split() && (long long)N * K >= 128 * 128 && M >= 1024) {
        // split-precision path (dtc_set_gemm_split): this layer as a one-job grouped launch; the caller's workspace was sized by
        // dtc_linear_wgrad_workspace for either path
        const DtcWgradJob job = one_job(dZ, lddz, X, dW, db, N, K);
        if (dtc_wgrad_group_s3_workspace(&job, 1, M) <= dtc_linear_wgrad_workspace(M, N, K)) return dtc_wgrad_group_s3(&job, 1, M, workspace, stream);
    }
    SegMatDev xd;
    int rc = to_dev(X, xd, K, false, M);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int bn = pick_bn(K);
    int col_tiles = 0;
    for (int i = 0; i < xd.nseg; ++i) col_tiles += (int)dtc::ceil_div(xd.s[i].width, bn);
    const int tiles = (int)dtc::ceil_div(N, BM) * col_tiles;
    const int splits = wgrad_splits(M, tiles);
    int rows_per_split = (int)dtc::ceil_div(M, splits);
    rows_per_split = (int)dtc::ceil_div(rows_per_split, BK) * BK;
    float* part = (float*)workspace;
    {
        dtc::ProfScope prof(dtc::prof_shape_name("linear_wgrad", M, N, K), 2.0 * M * (double)N * K, s,
                            4.0 * ((double)M * N + (double)M * K + (double)N * (K + 1)));      // dZ, X, dW + db (partials are overhead)
        const int grid = tiles * 8 * (int)dtc::ceil_div(splits, 8);
        if (bn == 64) hipLaunchKernelGGL(linear_wgrad_kernel<64>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, xd, part, M, N, K, rows_per_split, col_tiles, splits);
        else hipLaunchKernelGGL(linear_wgrad_kernel<32>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, xd, part, M, N, K, rows_per_split, col_tiles, splits);
    }
    {
        const long long total = (long long)N * part_ld(K);
        dtc::ProfScope prof(dtc::prof_shape_name("wgrad_reduce", splits, N, K), (double)total * 4.0 * (splits + 1), s);
        const dim3 grid((unsigned)dtc::ceil_div(part_ld(K) / 4, 64), (unsigned)N);
        // more split groups when the output is small (few blocks): the sum over splits is then the latency chain
        if (total >= (1 << 17) || splits <= 16)
            hipLaunchKernelGGL(wgrad_reduce_kernel<4>, grid, dim3(64, 4), 0, s, part, dW, db, N, K, splits);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel<16>, grid, dim3(64, 16), 0, s, part, dW, db, N, K, splits);
    }
    return dtc::check_launch("linear_wgrad");
}

// dW = dZ[idx]^T X[idx], db = column sums of dZ[idx]: both operands through ONE row map (see DtcWgradJob.dz_rows)
extern "C" int dtc_linear_wgrad_rows(const float* dZ, int64_t lddz, int64_t dz_rows, const float* X, int64_t ldx, int64_t x_rows,
                                     const int64_t* idx, float* dW, float* db, void* workspace, int M, int N, int K, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddz >= N && ldx >= K && dz_rows > 0 && x_rows > 0, "bad shape");
    DTC_REQUIRE(dZ && X && idx && dW && workspace && dtc::aligned16(workspace), "null pointer / unaligned workspace");
    DTC_REQUIRE(dtc_get_gemm_split(), "row-mapped weight gradients need the split-precision path (dtc_set_gemm_split)");
    DtcSegMat xs;
    xs.nseg = 1;
    xs.cols = K;
    xs.idx = const_cast<int64_t*>(idx);
    xs.seg[0] = DtcSeg{const_cast<float*>(X), ldx, 0, K, 1, 0, x_rows};
    DtcWgradJob job = one_job(dZ, lddz, &xs, dW, db, N, K);
    job.dz_rows = dz_rows;
    DTC_REQUIRE(dtc_wgrad_group_s3_workspace(&job, 1, M) <= dtc_linear_wgrad_workspace(M, N, K), "workspace too small");
    return dtc_wgrad_group_s3(&job, 1, M, workspace, stream);
}

extern "C" int64_t dtc_wgrad_group_workspace(const DtcWgradJob* jobs, int count, int M) {
    GroupPlan P;
    if (plan_group(jobs, count, M, nullptr, P) != DTC_OK) return -1;
    return P.bytes;
}

extern "C" int dtc_wgrad_group(const DtcWgradJob* jobs, int count, int M, void* workspace, void* stream) {
    DTC_REQUIRE(workspace != nullptr && dtc::aligned16(workspace), "wgrad group workspace must be a 16-byte aligned device buffer");
    GroupPlan P;
    int rc = plan_group(jobs, count, M, workspace, P);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const WGroupDev& G = P.dev;
    {
        dtc::ProfScope prof(dtc::prof_shape_name("linear_wgrad", M, G.tiles_total, count), P.flop, s, P.algo_bytes);
        const int grid = G.tiles_total * 8 * (int)dtc::ceil_div(G.splits, 8);
        // interleaved column tiles (one ds_read_b64 per operand pair): measured 31.95 vs 31.8 ms per step for the weight
        // gradients (round 3, three interleaved runs) -- the LDS read count is not what bounds this kernel; off by default
        hipLaunchKernelGGL(wgrad_group_kernel<false>, dim3(grid), dim3(256), occ_pad("WGRAD", 25600), s, G);
    }
    {
        dtc::ProfScope prof(dtc::prof_shape_name("wgrad_reduce", G.splits, G.tiles_total, count), (double)P.bytes + P.bytes / (double)G.splits, s);
        hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(P.red_blocks), dim3(256), 0, s, G);
    }
    return dtc::check_launch("wgrad_group");
}
