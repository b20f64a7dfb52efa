// fp32 MFMA dense layers for gfx950: forward, data-gradient and weight-gradient GEMMs of the
// nn.Linear stacks of rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349 as they
// are used by PPO.update (rsl_rl/rsl_rl/algorithms/ppo.py:197-218, 265, 289, 252, 333).
//
// All three products run on v_mfma_f32_32x32x2_f32 (exact fp32: bitwise a k-ordered fmaf
// chain, 157.3 TFLOP/s peak, there is no xf32/TF32 on gfx950).  The instruction takes 64
// cycles per issue, so one 128x64x16 block step is 1024 MFMA cycles per SIMD against 12 dword
// loads + 12 LDS writes + 24 LDS reads per lane: the kernels are MFMA-bound *if* the non-matrix
// instructions stay out of the way.  The design therefore
//   (1) removes HBM passes instead of polishing them:
//       - torch.cat([...], dim=1) operands (ppo.py:201, actor_critic_decoder.py:431,550) and the
//         mini-batch gather `tensor[batch_idx]` (rollout_storage.py:195-209) are folded into the
//         operand loader through a "segmented matrix" descriptor (DtcSegMat): up to 4 column
//         blocks, each optionally row-gathered.  Neither the cat nor the gathered batch exists in HBM;
//       - bias + ReLU/ELU are fused into the forward epilogue; the activation derivative is fused
//         into the data-gradient epilogue (needs only the saved post-activation output); the bias
//         gradient (column sums of dZ) is accumulated by the weight-gradient kernel while it stages
//         dZ, so no extra pass over dZ exists;
//       - K and N tails (53, 265, 531, 584, 693, 752, 12, 1 ...) are zero-padded in LDS, never in HBM;
//   (2) keeps address arithmetic out of the K loop: every operand load is a raw buffer load
//       `descriptor (SGPRs) + uniform step offset (SGPR) + loop-invariant 32-bit lane offset (VGPR)`; row and
//       k tails use an out-of-range lane offset, for which the hardware returns 0 without touching memory
//       (no clamps, no masks, no over-reads);
//   (3) keeps control flow out of the K loop: the steady state (full tiles of one segment) is ONE basic block --
//       loads of tile k+1, MFMAs of tile k, LDS stores, barrier; the masked last tile of a segment and the hop
//       into the next segment are peeled copies of that block.  Loads are unconditional (a predicated load
//       makes hipcc wrap each load in its own exec branch with a wait in between).
// Tiling: 256 threads = 4 waves; block tile 128 x {64,32}; each wave owns 32x32 MFMA sub-tiles; K step 16; LDS
// tiles are stored [k][row] (+4 pad) so every MFMA operand read is a conflict-free ds_read of 32 consecutive
// floats per half-wave; double-buffered LDS, one barrier per step, 4 workgroups per CU (latency is hidden by
// the other three -- tools/mfma_ceiling.hip measures what each ingredient of this loop costs).
// Workgroup -> tile mapping is XCD-aware (block b runs on XCD b%8): all column tiles of one row panel
// (fwd/dgrad) resp. all output tiles of one batch slice (wgrad) run on the same XCD, so the panel is
// fetched from HBM into ONE L2 and shared there.
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
};
struct SegMatDev {
    int nseg, cols;
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

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == DTC_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DTC_ACT_ELU) return v > 0.f ? v : expm1f(v);
    return v;
}
// derivative expressed through the saved post-activation output y
__device__ __forceinline__ float act_bwd(float g, float y, int act) {
    if (act == DTC_ACT_RELU) return y > 0.f ? g : 0.f;
    if (act == DTC_ACT_ELU) return y > 0.f ? g : g * (y + 1.0f);
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

// ------------------------------------------------------------------------------------------
// Forward: Y[M,N] = act(X[M,K] W[N,K]^T + b)
// ------------------------------------------------------------------------------------------
// optional loss epilogue of the forward kernel (dtc_linear_fwd_mse): the layer output feeds an MSE against a
// row-gathered target; the epilogue writes dL/dY instead of Y and one double partial of sum(e^2) per workgroup
struct MseEpi {
    const float* target;
    const long long* tidx;
    long long ldt;
    long long target_bytes;
    int tcol0;
    float scale;
    double* part;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int BN, bool MSE = false>
__global__ __launch_bounds__(256, 3) void linear_fwd_kernel(const SegMatDev X, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Y,
                                                         long long ldy, int M, int N, int K, int act, const MseEpi mse) {
    using C = Cfg<BN>;
    __shared__ float As[2][BK][C::LDA];
    __shared__ float Bs[2][BK][C::LDB];
    int tr, tc;
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, tr, tc)) {
        if (MSE && threadIdx.x == 0) mse.part[blockIdx.x] = 0.0;      // padding block: its partial slot still gets summed
        return;
    }
    const int m0 = tr * BM, n0 = tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = (wave / C::WN) * (32 * C::TM), wn_off = (wave % C::WN) * (32 * C::TN);

    // loader geometry: thread owns k = kk and rows rbase + 16*i of both operand tiles
    const int kk = tid & (BK - 1), rbase = tid / BK;
    constexpr int NA = BM / RP, NB = BN / RP;
    int arow[NA], grow[NA];
    u32 woff[NB];                                   // lane byte offset into W (loop invariant)
    bool any_gather = false;
    for (int s = 0; s < X.nseg; ++s) any_gather |= X.s[s].gather != 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + rbase + RP * i;
        arow[i] = m < M ? m : -1;
        grow[i] = (any_gather && m < M) ? (int)X.idx[m] : arow[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + rbase + RP * i;
        woff[i] = n < N ? (u32)(n * K + kk) * 4u : INVALID;
    }
    const rsrc_t wres = make_rsrc(W);

    // K loop: segment by segment.  The steady-state loop over the full tiles of a segment is ONE basic
    // block (loads of tile kt+1, MFMAs of tile kt, LDS stores, barrier) with no masks and no branches; the
    // last tile of a segment (k tail -> masked) and the hop into the next segment are peeled copies.
    SegDev sd = X.s[0];
    rsrc_t ares;
    u32 aoff[NA];                                   // lane byte offset of the A element inside the segment
    auto enter_segment = [&]() {
        ares = make_rsrc(sd.ptr);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = sd.gather ? grow[i] : arow[i];
            aoff[i] = r >= 0 ? ((u32)r * (u32)sd.ld + (u32)(sd.col0 + kk)) * 4u : INVALID;
        }
    };
    float ra[NA], rb[NB];
    auto load_tile = [&](auto masked, int kt) {     // tile kt of the current segment -> registers
        const u32 kmask = decltype(masked)::value ? oob_mask(kt * BK + kk, sd.width - 1) : 0u;
        const u32 ka = (u32)(kt * BK) * 4u, kw = (u32)(sd.start + kt * BK) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload(ares, aoff[i] | kmask, ka);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload(wres, woff[i] | kmask, kw);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) As[buf][kk][rbase + RP * i] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][kk][rbase + RP * i] = rb[i];
    };

    f32x16 acc[C::TM][C::TN];
    zero_acc<BN>(acc);

    int buf = 0;
    auto step = [&](auto masked, int kt_next) {     // stage tile kt_next while the MFMAs consume LDS[buf]
        load_tile(masked, kt_next);
        mfma_step<BN>(&As[buf][0][0], &Bs[buf][0][0], acc, lane, wm_off, wn_off);
        store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    };
    enter_segment();
    load_tile(Masked{}, 0);
    store_tile(0);
    __syncthreads();
    for (int seg = 0;;) {
        const int n = (sd.width + BK - 1) / BK;
        for (int kt = 1; kt + 1 < n; ++kt) step(Full{}, kt);
        if (n > 1) step(Masked{}, n - 1);
        if (++seg == X.nseg) break;
        sd = X.s[seg];
        enter_segment();
        step(Masked{}, 0);
    }
    mfma_step<BN>(&As[buf][0][0], &Bs[buf][0][0], acc, lane, wm_off, wn_off);

    const int half = lane >> 5, l31 = lane & 31;
    const bool full = (m0 + BM <= M) && (n0 + BN <= N);
    if (MSE) {
        // e = (acc + bias) - target[tidx[row], tcol0 + col];  dY = e * scale;  partial = sum e^2 (double)
        const rsrc_t tres = make_rsrc_bytes(mse.target, mse.target_bytes);
        double sq = 0.0;
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
            const int col = n0 + wn_off + 32 * j + l31;
            const bool cok = col < N;
            const float bv = (bias && cok) ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
                const int row0 = m0 + wm_off + 32 * i + 4 * half;
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {          // 16 gathered target loads in flight (rows past M read row 0, masked below)
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    const long long src = mse.tidx[row < M ? row : 0];
                    t[r] = bload(tres, (u32)((src * mse.ldt + mse.tcol0 + col) * 4) | (cok ? 0u : INVALID), 0u);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    if (cok && row < M) {
                        const float e = (acc[i][j][r] + bv) - t[r];
                        Y[(long long)row * ldy + col] = e * mse.scale;
                        sq += (double)e * (double)e;
                    }
                }
            }
        }
        sq = wave_sum_f64(sq);
        double* red = reinterpret_cast<double*>(&As[0][0][0]);
        __syncthreads();                                // all waves are past their last LDS read
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) mse.part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
        return;
    }
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
        const int col = n0 + wn_off + 32 * j + l31;
        const bool cok = col < N;
        const float bv = (bias && cok) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
            float* yp = Y + (long long)(m0 + wm_off + 32 * i + 4 * half) * ldy + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float v = act_fwd(acc[i][j][r] + bv, act);
                if (full) yp[(long long)ro * ldy] = v;
                else if (cok && m0 + wm_off + 32 * i + 4 * half + ro < M) yp[(long long)ro * ldy] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// One fused GRU time step (forward): gh = h_{t-1} W_hh^T + b_hh for the three gates of 32 hidden units,
// then the gate math of torch.nn.GRU in the epilogue -- the [R,3H] recurrent pre-activations never touch HBM:
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h_t = (1 - z) * n + z * h_{t-1}
// Block tile: 128 rows x (32 units x 3 gates); wave w owns rows 32w..32w+31 and three 32x32 MFMA tiles (r, z, n of
// the same units), so every lane holds the three pre-activations of its (row, unit) pairs.  Same K loop as
// linear_fwd_kernel (plain single-segment operands).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256, 3) void gru_step_fwd_kernel(const float* __restrict__ hprev, const float* __restrict__ Whh,
                                                           const float* __restrict__ bhh, const float* __restrict__ gi,
                                                           float* __restrict__ hout, float* __restrict__ gates,
                                                           float* __restrict__ hn, int R, int H) {
    constexpr int GB = 96, LDAg = BM + PAD, LDBg = GB + PAD;
    __shared__ float As[2][BK][LDAg];
    __shared__ float Bs[2][BK][LDBg];
    int tr, tc;
    if (!map_tile(blockIdx.x, (R + BM - 1) / BM, H / 32, tr, tc)) return;
    const int m0 = tr * BM, j0 = tc * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = wave * 32;
    const int kk = tid & (BK - 1), rbase = tid / BK;
    constexpr int NA = BM / RP, NB = GB / RP;
    u32 aoff[NA], woff[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + rbase + RP * i;
        aoff[i] = m < R ? (u32)(m * H + kk) * 4u : INVALID;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = rbase + RP * i;                       // 0..95: gate n / 32, unit j0 + n % 32
        woff[i] = (u32)(((n >> 5) * H + j0 + (n & 31)) * H + kk) * 4u;
    }
    const rsrc_t ares = make_rsrc(hprev), wres = make_rsrc(Whh);
    // One GRU step has ~12 row tiles x 16 unit tiles = 192 workgroups for 256 CUs: one wave per SIMD, nothing else to
    // hide the load latency behind.  Loads therefore run TWO K steps ahead of the MFMAs (two register sets, loop
    // unrolled by two): a tile has two MFMA phases (~1.5 us) to arrive instead of one.
    float ra[2][NA], rb[2][NB];
    auto load_tile = [&](auto set, int kt) {
        constexpr int S = decltype(set)::value;
        const u32 ko = (u32)(kt * BK) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[S][i] = bload(ares, aoff[i], ko);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[S][i] = bload(wres, woff[i], ko);
    };
    auto store_tile = [&](auto set, int buf) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) As[buf][kk][rbase + RP * i] = ra[S][i];
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][kk][rbase + RP * i] = rb[S][i];
    };
    f32x16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    const int half = lane >> 5, l31 = lane & 31;
    auto mfma = [&](int buf) {
        const float* ap = &As[buf][0][0] + half * LDAg + wm_off + l31;
        const float* bp = &Bs[buf][0][0] + half * LDBg + l31;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const float a = ap[2 * kp * LDAg];
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[2 * kp * LDBg + 32 * g], acc[g], 0, 0, 0);
        }
    };
    const int KT = H / BK;                                  // even: H is a multiple of 32
    int buf = 0;
    load_tile(S0{}, 0);
    store_tile(S0{}, 0);
    __syncthreads();
    load_tile(S1{}, 1);                                     // invariant: LDS[buf] = tile kt, set 1 = tile kt+1 in flight
    for (int kt = 0; kt + 2 < KT; kt += 2) {
        load_tile(S0{}, kt + 2);
        mfma(buf);
        store_tile(S1{}, buf ^ 1);
        __syncthreads();
        buf ^= 1;
        load_tile(S1{}, kt + 3);
        mfma(buf);
        store_tile(S0{}, buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    mfma(buf);
    store_tile(S1{}, buf ^ 1);
    __syncthreads();
    mfma(buf ^ 1);

    // epilogue: gate math; gi / h_{t-1} come in through unconditional buffer loads (rows >= R read 0)
    const int j = j0 + l31;
    const float br = bhh[j], bz = bhh[H + j], bn = bhh[2 * H + j];
    const rsrc_t gres = make_rsrc_bytes(gi, (long long)R * 3 * H * 4), hres = make_rsrc_bytes(hprev, (long long)R * H * 4);
    const int row0 = m0 + wm_off + 4 * half;
    float gr[16], gz[16], gn[16], hp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
        const u32 go = (u32)((row0 + ro) * 3 * H + j) * 4u;
        gr[r] = bload(gres, go, 0u);
        gz[r] = bload(gres, go, (u32)H * 4u);
        gn[r] = bload(gres, go, (u32)H * 8u);
        hp[r] = bload(hres, (u32)((row0 + ro) * H + j) * 4u, 0u);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        const float rg = sigmoid_f(gr[r] + (acc[0][r] + br));
        const float zg = sigmoid_f(gz[r] + (acc[1][r] + bz));
        const float ghn = acc[2][r] + bn;
        const float ng = tanhf(gn[r] + rg * ghn);
        if (row < R) {
            const long long e = (long long)row * H + j;
            float* gp = gates + (long long)row * 3 * H + j;
            hout[e] = (1.0f - zg) * ng + zg * hp[r];
            gp[0] = rg;
            gp[H] = zg;
            gp[2 * H] = ng;
            hn[e] = ghn;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Data gradient: dX[M,K] = (dZ[M,N] W[N,K]) * act'(Xsaved)   (reduction over N)
// ------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256, 3) void linear_dgrad_kernel(const float* __restrict__ dZ, long long lddz,
                                                           const float* __restrict__ W, const SegMatDev dX,
                                                           const float* __restrict__ Xs, long long ldxs, int M, int N,
                                                           int K, int act, int split_n, long long split_dst, int col_skip) {
    using C = Cfg<BN>;
    __shared__ float As[2][BK][C::LDA];
    __shared__ float Bs[2][BK][C::LDB];
    int tr, tc;
    // col_skip: leading columns whose destination is NULL (inputs that need no gradient, e.g. the raw observations in
    // front of the actor's features) -- the column tiles start behind them, nothing is computed for them
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (K - col_skip + BN - 1) / BN, tr, tc)) return;
    // split reduction (dtc_linear_dgrad_split): grid.y = chunk g of the reduction index; chunk g multiplies columns
    // [g*split_n, (g+1)*split_n) of dZ with the matching rows of W and writes its own destination matrix
    long long dst_off = 0;
    if (split_n > 0) {
        dZ += (long long)blockIdx.y * split_n;
        W += (long long)blockIdx.y * split_n * K;
        N = split_n;
        dst_off = (long long)blockIdx.y * split_dst;
    }
    const int m0 = tr * BM, c0 = col_skip + tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = (wave / C::WN) * (32 * C::TM), wn_off = (wave % C::WN) * (32 * C::TN);

    const int kk = tid & (BK - 1), rbase = tid / BK;      // A loader (dZ rows, reduction index n contiguous)
    constexpr int NA = BM / RP;
    constexpr int RPP = 256 / BN;                   // B loader: reduction rows per pass
    constexpr int NB = BK / RPP;
    const int bj = tid % BN, bk0 = tid / BN;
    u32 aoff[NA], boff[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + rbase + RP * i;
        aoff[i] = m < M ? (u32)((long long)m * lddz + kk) * 4u : INVALID;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
        boff[i] = c0 + bj < K ? (u32)((bk0 + RPP * i) * K + c0 + bj) * 4u : INVALID;
    const rsrc_t ares = make_rsrc(dZ), bres = make_rsrc(W);

    float ra[NA], rb[NB];
    auto load_tile = [&](auto masked, int n_0) {
        constexpr bool MK = decltype(masked)::value;
        const u32 nmask = MK ? oob_mask(n_0 + kk, N - 1) : 0u;
        const u32 sa = (u32)n_0 * 4u, sb = (u32)n_0 * (u32)K * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload(ares, aoff[i] | nmask, sa);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload(bres, boff[i] | (MK ? oob_mask(n_0 + bk0 + RPP * i, N - 1) : 0u), sb);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) As[buf][kk][rbase + RP * i] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][bk0 + RPP * i][bj] = rb[i];
    };

    f32x16 acc[C::TM][C::TN];
    zero_acc<BN>(acc);

    int buf = 0;
    auto step = [&](auto masked, int kt_next) {
        load_tile(masked, kt_next * BK);
        mfma_step<BN>(&As[buf][0][0], &Bs[buf][0][0], acc, lane, wm_off, wn_off);
        store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    };
    const int KT = (N + BK - 1) / BK;
    load_tile(Masked{}, 0);
    store_tile(0);
    __syncthreads();
    for (int kt = 1; kt + 1 < KT; ++kt) step(Full{}, kt);      // branch-free steady state (full tiles)
    if (KT > 1) step(Masked{}, KT - 1);                        // N tail
    mfma_step<BN>(&As[buf][0][0], &Bs[buf][0][0], acc, lane, wm_off, wn_off);

    // epilogue: the saved activations come in through unconditional buffer loads (rows >= M fall past the
    // descriptor's size, columns >= K get the INVALID offset -> 0), so all 16 loads of a 32x32 tile are in
    // flight together instead of one exec-guarded load -> select -> store chain per element
    const int half = lane >> 5, l31 = lane & 31;
    const rsrc_t xres = make_rsrc_bytes(Xs, (long long)M * ldxs * 4);
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
        const int col = c0 + wn_off + 32 * j + l31;
        const bool cok = col < K;
        const SegDev sd = dX.s[find_seg(dX, cok ? col : 0)];
        const bool live = cok && sd.ptr != nullptr;
        float* dst = sd.ptr + dst_off + sd.col0 + (col - sd.start);
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
            const int row0 = m0 + wm_off + 32 * i + 4 * half;
            float y[16];
            if (act != DTC_ACT_NONE) {
                const u32 xoff = ((u32)row0 * (u32)ldxs + (u32)col) * 4u | (cok ? 0u : INVALID);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    y[r] = bload(xres, xoff, (u32)(((r & 3) + 8 * (r >> 2)) * (int)ldxs) * 4u);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r];
                if (act != DTC_ACT_NONE) v = act_bwd(v, y[r], act);
                if (live && row < M) {
                    float* q = dst + (long long)row * sd.ld;
                    *q = sd.accumulate ? (*q + v) : v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient (split over the batch): part[s][n][c] = sum_{m in split s} dZ[m,n] X[m,c],
// c == K holds the bias-gradient partial.  A second kernel reduces the splits.
// ------------------------------------------------------------------------------------------
// Column tiles are aligned to the segments of X (tile = (segment, tile inside the segment)), so every block
// reads ONE source matrix: uniform descriptor, and -- because all lanes of a wave stage the same batch row --
// the (optionally gathered) row offset is a SCALAR: idx[m] comes in through s_load, row*ld goes into the
// SGPR offset of the buffer load, the lane offset is just the column.  Zero VALU per load.
template <int BN>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dZ, long long lddz,
                                                           const SegMatDev X, float* __restrict__ part, int M, int N,
                                                           int K, int rows_per_split, int col_tiles, int splits) {
    using C = Cfg<BN>;
    __shared__ float As[2][BK][C::LDA];
    __shared__ float Bs[2][BK][C::LDB];
    const int row_tiles = (N + BM - 1) / BM;
    const int tiles = row_tiles * col_tiles;
    // block b runs on XCD b%8: every XCD owns whole batch slices (splits), so each slice of dZ / X is
    // pulled from HBM into ONE L2 and shared there by all output tiles
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int split = xcd + 8 * (jb / tiles);
    if (split >= splits) return;                    // padding block of the last (partial) group of 8 splits
    const int t = jb % tiles;
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

    // A loader: As[kk][i] = dZ[m][n0+i], i = tid % 128, two reduction rows per pass
    const int ai = tid & 127, ak0 = tid >> 7;
    constexpr int NA = BK / 2;
    const bool arow_ok = n0 + ai < N;
    // B loader: Bs[kk][j] = X[m][col]; RPP batch rows per pass
    constexpr int RPP = 256 / BN;
    constexpr int NB = BK / RPP;
    constexpr bool ROW_UNIFORM = BN >= 64;          // all lanes of a wave stage the same batch row
    const int bj = tid % BN, bk0 = tid / BN;
    const int bk0u = ROW_UNIFORM ? __builtin_amdgcn_readfirstlane(bk0) : bk0;
    const bool bcol_ok = lc0 + bj < sd.width;
    const u32 ldb = (u32)sd.ld * 4u;
    const u32 bcolb = bcol_ok ? (u32)(sd.col0 + lc0 + bj) * 4u : INVALID;
    u32 aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) aoff[i] = arow_ok ? (u32)((long long)(ak0 + 2 * i) * lddz + n0 + ai) * 4u : INVALID;
    const rsrc_t ares = make_rsrc(dZ), bres = make_rsrc(sd.ptr);

    float ra[NA], rb[NB];
    float bias_acc = 0.f;
    auto load_tile = [&](auto masked, int mb) {
        constexpr bool MK = decltype(masked)::value;
        const u32 sa = (u32)mb * (u32)lddz * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload(ares, aoff[i] | (MK ? oob_mask(mb + ak0 + 2 * i, m_end - 1) : 0u), sa);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = mb + bk0u + RPP * i;
            const int mc = (!MK || m < m_end) ? m : m_end - 1;
            const u32 r = sd.gather ? (u32)X.idx[mc] : (u32)mc;          // scalar when ROW_UNIFORM
            rb[i] = bload(bres, bcolb | (MK ? oob_mask(m, m_end - 1) : 0u), r * ldb);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            As[buf][ak0 + 2 * i][ai] = ra[i];
            bias_acc += ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][bk0 + RPP * i][bj] = rb[i];
    };

    f32x16 acc[C::TM][C::TN];
    zero_acc<BN>(acc);

    const int KT = (m_end - m_begin + BK - 1) / BK;
    if (KT > 0) {                                   // uniform per block (an empty trailing split writes zeros)
        int buf = 0;
        auto step = [&](auto masked, int kt_next) {
            load_tile(masked, m_begin + kt_next * BK);
            mfma_step<BN>(&As[buf][0][0], &Bs[buf][0][0], acc, lane, wm_off, wn_off);
            store_tile(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        };
        load_tile(Masked{}, m_begin);
        store_tile(0);
        __syncthreads();
        for (int kt = 1; kt + 1 < KT; ++kt) step(Full{}, kt);  // branch-free steady state (full tiles)
        if (KT > 1) step(Masked{}, KT - 1);                    // batch tail of the split
        mfma_step<BN>(&As[buf][0][0], &Bs[buf][0][0], acc, lane, wm_off, wn_off);
    }

    const long long ldp = part_ld(K);
    float* P = part + (long long)split * N * ldp;
    const int half = lane >> 5, l31 = lane & 31;
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
    if (seg == 0 && tc == 0) {   // bias-gradient partial: two threads staged each dZ column
        float* red = &As[0][0][0];
        __syncthreads();
        if (ak0 == 1) red[ai] = bias_acc;
        __syncthreads();
        if (ak0 == 0 && arow_ok) P[(long long)(n0 + ai) * ldp + K] = bias_acc + red[ai];
    }
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
            s.ptr = hs.ptr;
            s.ld = hs.ld;
            s.col0 = hs.col0;
            s.start = start;
            s.width = hs.width;
            s.gather = hs.gather;
            s.accumulate = hs.accumulate;
            start += hs.width;
        } else {
            s = SegDev{nullptr, 0, 0, 0x7fffffff, 0, 0, 0};
        }
    }
    d.cols = start;
    DTC_REQUIRE(start == expect_cols, "segments cover %d columns, expected %d", start, expect_cols);
    return DTC_OK;
}

// wgrad column-tile width: 64 measured at least as fast as 128 on every layer of this model (sweep in
// tools/microbench.py wgrad); DTC_WGRAD_BN overrides for experiments
int pick_bn(int cols) {
    static const char* force = getenv("DTC_WGRAD_BN");
    if (force && cols > 64) return atoi(force);
    return cols <= 32 ? 32 : 64;
}

// fwd / dgrad: 128x64 tiles measured faster than 128x128 at every layer width of this model (twice the
// workgroups -> prologue / epilogue of one block overlap the MFMA phase of its neighbours, 4 waves/SIMD);
// when even those leave the chip short of workgroups (narrow layers, the ~1500-row GEMMs of one GRU time
// step) 128x32 tiles double the count again
int pick_bn_rows(int rows, int cols) {
    static const char* force = getenv("DTC_GEMM_BN");
    if (force && cols > 64) return atoi(force);
    if (cols <= 32) return 32;
    static const char* thr_env = getenv("DTC_GEMM_MIN_BLOCKS");
    const long long min_blocks = thr_env ? atoi(thr_env) : 320;     // measured with DTC_GEMM_MIN_BLOCKS sweeps of bench.py
    return dtc::ceil_div(rows, BM) * dtc::ceil_div(cols, 64) >= min_blocks ? 64 : 32;
}

// wgrad split heuristic -------------------------------------------------------------------------------------
int wgrad_splits(int M, int tiles) {
    static const char* target_env = getenv("DTC_WGRAD_BLOCKS");
    const int target = target_env ? atoi(target_env) : 1024;
    // whole splits per XCD (multiple of 8) measured 10-20 % faster than filling the wave with an arbitrary count
    // (DTC_WGRAD_ANYSPLIT=1: 44 tiles x 23 splits = 1012 blocks ran slower than 44 x 16 = 704)
    static const bool mult8 = getenv("DTC_WGRAD_ANYSPLIT") == nullptr;
    int s = target / tiles;
    if (mult8) s = s / 8 * 8;
    if (s < 8) s = 8;
    const int max_s = (int)dtc::ceil_div(dtc::ceil_div(M, BK * 8), 8) * 8;
    if (s > max_s) s = max_s;
    return s;
}
// upper bound over every segmentation of X (segment-aligned column tiles only add tiles -> fewer splits)
int wgrad_splits_bound(int M, int N, int K) {
    return wgrad_splits(M, (int)(dtc::ceil_div(N, BM) * dtc::ceil_div(K, pick_bn(K))));
}

}  // namespace

// NOTE on sizes: lane offsets are 32-bit byte offsets below 2 GiB, so every operand matrix must stay below
// 2^29 elements; gathered sources are bounded by the caller (row index * ld < 2^29), which holds for the
// rollouts of up to ~15000 envs x 24 steps per GPU this path is designed for (4096 x 24 x 1389 = 1.4e8).

extern "C" int dtc_linear_fwd(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, int M, int N,
                              int K, int act, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && ldy >= N, "bad shape M=%d N=%d K=%d ldy=%lld", M, N, K, (long long)ldy);
    DTC_REQUIRE(W && Y, "null pointer");
    DTC_REQUIRE(act >= 0 && act <= 2, "bad activation %d", act);
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * ldy <= MAX_ELEMS * 4, "matrix too large");
    SegMatDev xd;
    int rc = to_dev(X, xd, K, false, M);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int bn = pick_bn_rows(M, N);
    const int grid = grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, bn));
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, K), 2.0 * M * (double)N * K, s);
    if (bn == 128) hipLaunchKernelGGL((linear_fwd_kernel<128, false>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy, M, N, K, act, MseEpi{});
    else if (bn == 64) hipLaunchKernelGGL((linear_fwd_kernel<64, false>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy, M, N, K, act, MseEpi{});
    else hipLaunchKernelGGL((linear_fwd_kernel<32, false>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy, M, N, K, act, MseEpi{});
    return dtc::check_launch("linear_fwd");
}

extern "C" int64_t dtc_linear_fwd_mse_parts(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 64));
}

extern "C" int dtc_linear_fwd_mse(const DtcSegMat* X, const float* W, const float* b, const float* target, int64_t ldt,
                                  int64_t target_rows, int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy,
                                  double* sq_part, int M, int N, int K, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N, "bad shape M=%d N=%d K=%d", M, N, K);
    DTC_REQUIRE(W && target && tidx && dY && sq_part, "null pointer");
    DTC_REQUIRE(tcol0 >= 0 && tcol0 + N <= ldt && target_rows > 0, "target columns [%d, %d) outside its %lld-wide rows", tcol0,
                tcol0 + N, (long long)ldt);
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * lddy <= MAX_ELEMS && target_rows * ldt <= MAX_ELEMS, "matrix too large");
    SegMatDev xd;
    int rc = to_dev(X, xd, K, false, M);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 64));
    const MseEpi mse{target, (const long long*)tidx, (long long)ldt, target_rows * ldt * 4, tcol0, scale, sq_part};
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, K), 2.0 * M * (double)N * K, s);
    hipLaunchKernelGGL((linear_fwd_kernel<64, true>), dim3(grid), dim3(256), 0, s, xd, W, b, dY, (long long)lddy, M, N, K,
                       (int)DTC_ACT_NONE, mse);
    return dtc::check_launch("linear_fwd_mse");
}

extern "C" int dtc_gru_step_fwd(const float* hprev, const float* W_hh, const float* b_hh, const float* gi_t, float* hout,
                                float* gates_t, float* hn_t, int R, int H, void* stream) {
    DTC_REQUIRE(R > 0 && H > 0 && H % 32 == 0, "bad shape R=%d H=%d (H must be a multiple of 32)", R, H);
    DTC_REQUIRE(hprev && W_hh && b_hh && gi_t && hout && gates_t && hn_t, "null pointer");
    DTC_REQUIRE((long long)R * 3 * H <= MAX_ELEMS && 3ll * H * H <= MAX_ELEMS, "matrix too large");
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((int)dtc::ceil_div(R, BM), H / 32);
    dtc::ProfScope prof(dtc::prof_shape_name("gru_step_fwd", R, 3 * H, H), 2.0 * R * 3.0 * H * H, s);
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(grid), dim3(256), 0, s, hprev, W_hh, b_hh, gi_t, hout, gates_t, hn_t, R, H);
    return dtc::check_launch("gru_step_fwd");
}

extern "C" int dtc_linear_dgrad(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const float* Xsaved,
                                int64_t ldxs, int M, int N, int K, int act, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddz >= N, "bad shape");
    DTC_REQUIRE(dZ && W, "null pointer");
    DTC_REQUIRE(act >= 0 && act <= 2, "bad activation %d", act);
    DTC_REQUIRE(act == DTC_ACT_NONE || (Xsaved != nullptr && ldxs >= K), "activation derivative needs Xsaved");
    DTC_REQUIRE(act == DTC_ACT_NONE || (dX && dX->nseg == 1), "activation derivative needs a single-segment destination");
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * lddz <= MAX_ELEMS, "matrix too large");
    DTC_REQUIRE(act == DTC_ACT_NONE || (long long)M * ldxs <= MAX_ELEMS, "saved activation matrix too large");
    SegMatDev xd;
    int rc = to_dev(dX, xd, K, true, 0);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    int col_skip = 0;
    for (int i = 0; i < xd.nseg && xd.s[i].ptr == nullptr; ++i) col_skip += xd.s[i].width;
    DTC_REQUIRE(col_skip < K, "every destination segment is NULL");
    const int bn = pick_bn_rows(M, K - col_skip);
    const int grid = grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(K - col_skip, bn));
    dtc::ProfScope prof(dtc::prof_shape_name("linear_dgrad", M, N, K), 2.0 * M * (double)N * (K - col_skip), s);
    if (bn == 128) hipLaunchKernelGGL(linear_dgrad_kernel<128>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, W, xd, Xsaved, (long long)ldxs, M, N, K, act, 0, 0ll, col_skip);
    else if (bn == 64) hipLaunchKernelGGL(linear_dgrad_kernel<64>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, W, xd, Xsaved, (long long)ldxs, M, N, K, act, 0, 0ll, col_skip);
    else hipLaunchKernelGGL(linear_dgrad_kernel<32>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, W, xd, Xsaved, (long long)ldxs, M, N, K, act, 0, 0ll, col_skip);
    return dtc::check_launch("linear_dgrad");
}

extern "C" int dtc_linear_dgrad_split(const float* dZ, int64_t lddz, const float* W, float* dX, int64_t lddx,
                                      int64_t split_stride, int M, int N, int K, int nsplit, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && nsplit >= 1 && N % nsplit == 0 && lddz >= N && lddx >= K, "bad shape");
    DTC_REQUIRE(dZ && W && dX, "null pointer");
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * lddz <= MAX_ELEMS, "matrix too large");
    SegMatDev xd;
    xd.nseg = 1;
    xd.cols = K;
    xd.idx = nullptr;
    for (int i = 0; i < 4; ++i) xd.s[i] = SegDev{nullptr, 0, 0, 0x7fffffff, 0, 0, 0};
    xd.s[0] = SegDev{dX, (long long)lddx, 0, 0, K, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    const int chunk = N / nsplit;
    // tile width from the work of ALL chunks (they run side by side in one launch)
    const int row_tiles = (int)dtc::ceil_div(M, BM);
    const int bn = (K <= 32 || (long long)row_tiles * dtc::ceil_div(K, 64) * nsplit < 320) ? 32 : 64;
    const dim3 grid((unsigned)grid_for(row_tiles, (int)dtc::ceil_div(K, bn)), (unsigned)nsplit);
    dtc::ProfScope prof(dtc::prof_shape_name("linear_dgrad", M, N, K), 2.0 * M * (double)N * K, s);
    if (bn == 64) hipLaunchKernelGGL(linear_dgrad_kernel<64>, grid, dim3(256), 0, s, dZ, (long long)lddz, W, xd, nullptr, 0ll, M, N, K, (int)DTC_ACT_NONE, chunk, (long long)split_stride, 0);
    else hipLaunchKernelGGL(linear_dgrad_kernel<32>, grid, dim3(256), 0, s, dZ, (long long)lddz, W, xd, nullptr, 0ll, M, N, K, (int)DTC_ACT_NONE, chunk, (long long)split_stride, 0);
    return dtc::check_launch("linear_dgrad_split");
}

extern "C" int64_t dtc_linear_wgrad_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (int64_t)wgrad_splits_bound(M, N, K) * N * part_ld(K) * (int64_t)sizeof(float);
}

extern "C" int dtc_linear_wgrad(const float* dZ, int64_t lddz, const DtcSegMat* X, float* dW, float* db, void* workspace,
                                int M, int N, int K, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddz >= N, "bad shape");
    DTC_REQUIRE(dZ && dW && workspace, "null pointer");
    DTC_REQUIRE(dtc::aligned16(workspace), "wgrad workspace must be 16-byte aligned");
    DTC_REQUIRE((long long)M * lddz <= MAX_ELEMS, "matrix too large");
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
        dtc::ProfScope prof(dtc::prof_shape_name("linear_wgrad", M, N, K), 2.0 * M * (double)N * K, s);
        const int grid = tiles * 8 * (int)dtc::ceil_div(splits, 8);
        if (bn == 128) hipLaunchKernelGGL(linear_wgrad_kernel<128>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, xd, part, M, N, K, rows_per_split, col_tiles, splits);
        else if (bn == 64) hipLaunchKernelGGL(linear_wgrad_kernel<64>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, xd, part, M, N, K, rows_per_split, col_tiles, splits);
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
