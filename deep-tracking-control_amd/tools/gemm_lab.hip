// GEMM lab (gfx950, stand-alone, no torch): A/B bench + fp64 check of candidate block loops for the fp32 MFMA dense
// layers before they move into csrc/gemm.hip.
//   hipcc -O3 --offload-arch=gfx950 gemm_lab.hip -o _bin/gemm_lab && _bin/gemm_lab
// Y[M,N] = relu(X[M,K] W[N,K]^T + b)   (both operands reduction-contiguous: forward layers; data gradients against a
// pre-transposed weight have the same form)
//   v1: the round-1 loop -- dword buffer loads, LDS tiles [k][row]+4, ds_read_b32 fragments
//   v2: dwordx4 buffer loads along k, LDS tiles [row][16 k] with a 16-byte-chunk XOR swizzle (conflict-free
//       ds_write_b128 and ds_read_b128), one b128 fragment read feeds 4 MFMA k-steps (k order inside a 16-k stage is
//       permuted identically for both operands)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <vector>

#include "../../include/dtc_hip.h"
typedef int (*fwd_fn)(const DtcSegMat*, const float*, const float*, float*, int64_t, int, int, int, int, void*);
static fwd_fn g_prod_fwd = nullptr;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr u32 INVALID = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)INVALID, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, u32 voff, u32 soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 bload4(rsrc_t r, u32 voff, u32 soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ u32 oob_mask(int x, int limit) { return (u32)(limit - x) & INVALID; }

__device__ __forceinline__ bool map_tile(int b, int row_tiles, int col_tiles, int& tr, int& tc) {
    const int xcd = b & 7, j = b >> 3, local = j / col_tiles;
    tc = j - local * col_tiles;
    tr = xcd + 8 * local;
    return tr < row_tiles;
}
static int grid_for(int row_tiles, int col_tiles) { return 8 * ((row_tiles + 7) / 8) * col_tiles; }

// ------------------------------------------------------------------------------------------------ v1 (round-1 loop)
template <int BN>
__global__ __launch_bounds__(256, 3) void fwd_v1(const float* __restrict__ X, const float* __restrict__ W,
                                                 const float* __restrict__ bias, float* __restrict__ Y, int M, int N, int K) {
    constexpr int BM = 128, BK = 16, PAD = 4, RP = 16;
    constexpr int WM = 4, TN = BN / 32, LDA = BM + PAD, LDB = BN + PAD;
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];
    int tr, tc;
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, tr, tc)) return;
    const int m0 = tr * BM, n0 = tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = wave * 32;
    const int kk = tid & 15, rbase = tid >> 4;
    constexpr int NA = BM / RP, NB = BN / RP;
    u32 aoff[NA], woff[NB];
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + rbase + RP * i;
        aoff[i] = m < M ? (u32)(m * K + kk) * 4u : INVALID;
    }
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + rbase + RP * i;
        woff[i] = n < N ? (u32)(n * K + kk) * 4u : INVALID;
    }
    const rsrc_t ares = make_rsrc(X), wres = make_rsrc(W);
    float ra[NA], rb[NB];
    f32x16 acc[TN];
    for (int j = 0; j < TN; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int half = lane >> 5, l31 = lane & 31;
    auto load = [&](bool masked, int kt) {
        const u32 km = masked ? oob_mask(kt * BK + kk, K - 1) : 0u;
        const u32 ko = (u32)(kt * BK) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload(ares, aoff[i] | km, ko);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload(wres, woff[i] | km, ko);
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) As[buf][kk][rbase + RP * i] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][kk][rbase + RP * i] = rb[i];
    };
    auto mfma = [&](int buf) {
        const float* ap = &As[buf][0][0] + half * LDA + wm_off + l31;
        const float* bp = &Bs[buf][0][0] + half * LDB + l31;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const float a = ap[2 * kp * LDA];
            float b[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bp[2 * kp * LDB + 32 * j];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
        }
    };
    const int KT = (K + BK - 1) / BK;
    int buf = 0;
    load(true, 0);
    store(0);
    __syncthreads();
    for (int kt = 1; kt + 1 < KT; ++kt) {
        load(false, kt);
        mfma(buf);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (KT > 1) {
        load(true, KT - 1);
        mfma(buf);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    mfma(buf);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + 32 * j + l31;
        const bool cok = col < N;
        const float bv = cok ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm_off + 4 * half + (r & 3) + 8 * (r >> 2);
            const float v = acc[j][r] + bv;
            if (cok && row < M) Y[(long long)row * N + col] = v > 0.f ? v : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ v2
// Stage = 16 k.  LDS image of an operand tile: row r holds its 16 k-values as four 16-byte chunks; chunk c is
// stored at position c ^ ((r >> 2) & 3).  Loader thread t: row t/4 (+64 per pass), chunk t%4 -> one dwordx4 buffer
// load + one ds_write_b128 (the 8 lanes of a write group cover two rows x 4 chunks = banks 0-15 / 16-31: no conflict).
// MFMA lane (row i = lane & 31, half h = lane >> 5) reads chunk 2q + h of its row (q = 0, 1) with ONE ds_read_b128
// and feeds the four values to four consecutive MFMA k-steps; half 0 / half 1 of k-step (q, t) therefore multiply
// k = 8q + t and k = 8q + 4 + t -- both operands use the same assignment, every k of the stage is used exactly once.
__device__ unsigned long long* g_dbg = nullptr;
template <int BN, int OCC, int VAR>
__global__ __launch_bounds__(256, OCC) void fwd_v2(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int M, int N, int K) {
    constexpr int BM = 128, BK = 16;
    unsigned long long t0 = 0, r0 = 0;
    if (VAR == 1) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int TN = BN / 32;
    constexpr int NA = BM / 64, NB = (BN + 63) / 64;
    __shared__ f32x4 As[2][BM * 4];
    __shared__ f32x4 Bs[2][BN * 4];
    extern __shared__ float dyn_pad[];          // residency cap for experiments (dynamic LDS bytes chosen by the host)
    if (M < 0) dyn_pad[threadIdx.x] = 0.f;
    int tr, tc;
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, tr, tc)) return;
    const int m0 = tr * BM, n0 = tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = wave * 32;
    const int lrow = tid >> 2, lch = tid & 3;
    const bool bthread = BN >= 64 || tid < BN * 4;
    u32 aoff[NA], woff[NB];
    int aslot[NA], bslot[NB];
    for (int i = 0; i < NA; ++i) {
        const int r = lrow + 64 * i, m = m0 + r;
        aoff[i] = m < M ? (u32)(m * K + 4 * lch) * 4u : INVALID;
        aslot[i] = r * 4 + (lch ^ ((r >> 2) & 3));
    }
    for (int i = 0; i < NB; ++i) {
        const int r = lrow + 64 * i, n = n0 + r;
        woff[i] = (n < N && bthread) ? (u32)(n * K + 4 * lch) * 4u : INVALID;
        bslot[i] = r * 4 + (lch ^ ((r >> 2) & 3));
    }
    const rsrc_t ares = make_rsrc(X), wres = make_rsrc(W);
    f32x4 ra[NA], rb[NB];
    f32x16 acc[TN];
    for (int j = 0; j < TN; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int half = lane >> 5, l31 = lane & 31;
    auto load = [&](int kt) {
        const u32 ko = (u32)(kt * BK) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload4(ares, aoff[i], ko);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload4(wres, woff[i], ko);
    };
    auto load_masked = [&](int kt) {           // k tail: per-dword loads, elements past K read 0
        const u32 ko = (u32)(kt * BK) * 4u;
        const int kb = kt * BK + 4 * lch;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[i][e] = bload(ares, (aoff[i] + 4u * e) | oob_mask(kb + e, K - 1), ko);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[i][e] = bload(wres, (woff[i] + 4u * e) | oob_mask(kb + e, K - 1), ko);
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) As[buf][aslot[i]] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (BN >= 64 || bthread) Bs[buf][bslot[i]] = rb[i];
    };
    const int arow = wm_off + l31;
    const int asw = (arow >> 2) & 3;
    auto mfma = [&](int buf) {
        f32x4 a[2], b[TN][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a[q] = As[buf][arow * 4 + ((2 * q + half) ^ asw)];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int br = 32 * j + l31;
                b[j][q] = Bs[buf][br * 4 + ((2 * q + half) ^ ((br >> 2) & 3))];
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][t], b[j][q][t], acc[j], 0, 0, 0);
    };
    const int KT = (K + BK - 1) / BK;
    const bool tail = (K % BK) != 0;
    int buf = 0;
    if (KT == 1 && tail) load_masked(0); else load(0);
    store(0);
    __syncthreads();
    unsigned long long t_pro = 0, t_loop = 0;
    if (VAR == 1) t_pro = __builtin_amdgcn_s_memtime();
    const int full_end = tail ? KT - 1 : KT;
    if (VAR == 2) {
        // progress-based priority: a wave in an earlier quarter of its K loop outranks waves that are further along, so
        // the blocks of a CU finish together instead of oldest-first (no long tail with 1-2 waves per SIMD)
        const int q1 = KT / 4, q2 = KT / 2, q3 = (3 * KT) / 4;
        __builtin_amdgcn_s_setprio(3);
        for (int kt = 1; kt < full_end; ++kt) {
            if (kt == q1) __builtin_amdgcn_s_setprio(2);
            if (kt == q2) __builtin_amdgcn_s_setprio(1);
            if (kt == q3) __builtin_amdgcn_s_setprio(0);
            load(kt);
            mfma(buf);
            store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    } else
    for (int kt = 1; kt < full_end; ++kt) {
        load(kt);
        if (VAR == 4) __builtin_amdgcn_sched_barrier(0);
        mfma(buf);
        if (VAR == 4) __builtin_amdgcn_sched_barrier(0);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (tail && KT > 1) {
        load_masked(KT - 1);
        mfma(buf);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    mfma(buf);
    if (VAR == 1) t_loop = __builtin_amdgcn_s_memtime();
    if ((VAR == 3 || VAR == 4) && (N & 3) == 0 && m0 + BM <= M && n0 + BN <= N) {
        // wide-store epilogue: each wave transposes its 32x32 sub-tiles through a private LDS patch (lane = column ->
        // lane = 4 consecutive columns of one row) and writes dwordx4: 8 store instructions per wave instead of 32
        constexpr int LDW = 36;
        __syncthreads();                                  // every wave is past its last operand read
        float* patch = reinterpret_cast<float*>(&As[0][0]) + wave * (32 * LDW);
        const int rrow = lane >> 3, rc4 = lane & 7;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bv = bias[n0 + 32 * j + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[j][r] + bv;
                patch[((r & 3) + 8 * (r >> 2) + 4 * half) * LDW + l31] = v > 0.f ? v : 0.f;
            }
            // same wave wrote and reads: LDS ops of one wave execute in order, no barrier needed
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = rrow + 8 * p;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&patch[row * LDW + 4 * rc4]);
                *reinterpret_cast<f32x4*>(&Y[(long long)(m0 + wm_off + row) * N + n0 + 32 * j + 4 * rc4]) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + 32 * j + l31;
        const bool cok = col < N;
        const float bv = cok ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm_off + 4 * half + (r & 3) + 8 * (r >> 2);
            const float v = acc[j][r] + bv;
            if (cok && row < M) Y[(long long)row * N + col] = v > 0.f ? v : 0.f;
        }
    }
    if (VAR == 1 && threadIdx.x == 0) {
        g_dbg[(size_t)gridDim.x * 6 + (size_t)blockIdx.x * 2] = t_pro;
        g_dbg[(size_t)gridDim.x * 6 + (size_t)blockIdx.x * 2 + 1] = t_loop;
        unsigned long long* d = g_dbg + (size_t)blockIdx.x * 6;
        u32 hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        u32 xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[0] = t0; d[1] = __builtin_amdgcn_s_memtime(); d[2] = r0; d[3] = __builtin_amdgcn_s_memrealtime(); d[4] = hwid; d[5] = xcc;
    }
}


// ------------------------------------------------------------------------------------------------ v4
// 128 x 128 workgroup tile, four waves as 2 x 2, each wave 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers): 8
// ds_read_b128 feed 32 MFMAs per stage (v2: 6 for 16) and one barrier covers twice the MFMA work; 3 workgroups per CU.
// K % 16 == 0, M % 128 == 0, N % 128 == 0 only (lab).
template <int OCC, bool PIN = false>
__global__ __launch_bounds__(256, OCC) void fwd_v4(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int M, int N, int K) {
    constexpr int BM = 128, BN = 128, BK = 16;
    __shared__ f32x4 As[2][BM * 4];
    __shared__ f32x4 Bs[2][BN * 4];
    int tr, tc;
    if (!map_tile(blockIdx.x, M / BM, N / BN, tr, tc)) return;
    const int m0 = tr * BM, n0 = tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = (wave >> 1) * 64, wn_off = (wave & 1) * 64;
    const int lrow = tid >> 2, lch = tid & 3;
    u32 aoff[2], woff[2];
    int slot[2];
    for (int i = 0; i < 2; ++i) {
        const int r = lrow + 64 * i;
        aoff[i] = (u32)((m0 + r) * K + 4 * lch) * 4u;
        woff[i] = (u32)((n0 + r) * K + 4 * lch) * 4u;
        slot[i] = r * 4 + (lch ^ ((r >> 2) & 3));
    }
    const rsrc_t ares = make_rsrc(X), wres = make_rsrc(W);
    f32x4 ra[2], rb[2];
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int half = lane >> 5, l31 = lane & 31;
    auto load = [&](int kt) {
        const u32 ko = (u32)(kt * BK) * 4u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = bload4(ares, aoff[i], ko);
            rb[i] = bload4(wres, woff[i], ko);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            As[buf][slot[i]] = ra[i];
            Bs[buf][slot[i]] = rb[i];
        }
    };
    auto mfma = [&](int buf) {
        f32x4 a[2][2], b[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ar = wm_off + 32 * i + l31, br = wn_off + 32 * i + l31;
                a[i][q] = As[buf][ar * 4 + ((2 * q + half) ^ ((ar >> 2) & 3))];
                b[i][q] = Bs[buf][br * 4 + ((2 * q + half) ^ ((br >> 2) & 3))];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q][t], b[j][q][t], acc[i][j], 0, 0, 0);
    };
    const int KT = K / BK;
    int buf = 0;
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 1; kt < KT; ++kt) {
        load(kt);
        if (PIN) __builtin_amdgcn_sched_barrier(0);      // keep the loads of the next stage in front of this stage's MFMAs
        mfma(buf);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    mfma(buf);
    // wide-store epilogue through a per-wave LDS patch (as v2 VAR 3)
    constexpr int LDW = 36;
    __syncthreads();
    float* patch = reinterpret_cast<float*>(&As[0][0]) + wave * (32 * LDW);
    const int rrow = lane >> 3, rc4 = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = bias[n0 + wn_off + 32 * j + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][j][r] + bv;
                patch[((r & 3) + 8 * (r >> 2) + 4 * half) * LDW + l31] = v > 0.f ? v : 0.f;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = rrow + 8 * p;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&patch[row * LDW + 4 * rc4]);
                *reinterpret_cast<f32x4*>(&Y[(long long)(m0 + wm_off + 32 * i + row) * N + n0 + wn_off + 32 * j + 4 * rc4]) = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------ v3
// 512-thread workgroups: waves w and w+4 share the 32 x 64 output sub-tile of rows 32w and split the stage's k range
// (wave group 0 multiplies k-steps q = 0, group 1 q = 1), each into its own accumulators; after the K loop the two
// groups swap halves through LDS, add them, and each of the 8 waves finishes ONE 32 x 32 tile.  A workgroup then has two
// waves per SIMD: a workgroup that is alone on its CU (the tail of a launch) still overlaps its own loads / LDS
// traffic with MFMAs of its other wave.
template <int OCC>
__global__ __launch_bounds__(512, OCC) void fwd_v3(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int M, int N, int K) {
    constexpr int BM = 128, BN = 64, BK = 16, TN = 2;
    __shared__ f32x4 As[2][BM * 4];
    __shared__ f32x4 Bs[2][BN * 4];
    int tr, tc;
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, tr, tc)) return;
    const int m0 = tr * BM, n0 = tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wm_off = (wave & 3) * 32;
    // loaders: 512 A chunks (one per thread), 256 B chunks (threads 0..255)
    const int lrow = tid >> 2, lch = tid & 3;
    const int am = m0 + lrow;
    const u32 aoff = am < M ? (u32)(am * K + 4 * lch) * 4u : INVALID;
    const int aslot = lrow * 4 + (lch ^ ((lrow >> 2) & 3));
    const bool bthread = tid < 256;
    const int bn_ = n0 + (lrow & 63);
    const u32 woff = (bthread && bn_ < N) ? (u32)(bn_ * K + 4 * lch) * 4u : INVALID;
    const int bslot = (lrow & 63) * 4 + (lch ^ (((lrow & 63) >> 2) & 3));
    const rsrc_t ares = make_rsrc(X), wres = make_rsrc(W);
    f32x4 ra, rb;
    f32x16 acc[TN];
    for (int j = 0; j < TN; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int half = lane >> 5, l31 = lane & 31;
    auto load = [&](int kt) {
        const u32 ko = (u32)(kt * BK) * 4u;
        ra = bload4(ares, aoff, ko);
        rb = bload4(wres, woff, ko);
    };
    auto store = [&](int buf) {
        As[buf][aslot] = ra;
        if (bthread) Bs[buf][bslot] = rb;
    };
    const int arow = wm_off + l31;
    const int asw = (arow >> 2) & 3;
    auto mfma = [&](int buf) {               // this wave group's half of the stage: chunk 2*grp + half
        const f32x4 a = As[buf][arow * 4 + ((2 * grp + half) ^ asw)];
        f32x4 b[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int br = 32 * j + l31;
            b[j] = Bs[buf][br * 4 + ((2 * grp + half) ^ ((br >> 2) & 3))];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j][t], acc[j], 0, 0, 0);
    };
    const int KT = K / BK;                   // lab: K % 16 == 0 only
    int buf = 0;
    load(0);
    store(0);
    __syncthreads();
    for (int kt = 1; kt < KT; ++kt) {
        load(kt);
        mfma(buf);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    mfma(buf);
    __syncthreads();
    // swap halves: group g keeps sub-tile j = g and hands sub-tile 1-g to its partner wave (4 KiB per wave)
    float* xch = reinterpret_cast<float*>(&As[0][0]);           // 8 waves x 1024 floats = 32 KiB: As (16) + Bs (8) is too small -> two rounds
    f32x16 mine = grp == 0 ? acc[0] : acc[1];
    const f32x16 give = grp == 0 ? acc[1] : acc[0];
    // round 1: group 1 -> group 0 ; round 2: group 0 -> group 1  (4 waves x 4 KiB = 16 KiB per round)
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int giver = 1 - round;
        if (grp == giver) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(wave & 3) * 1024 + r * 64 + lane] = give[r];
        }
        __syncthreads();
        if (grp != giver) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] += xch[(wave & 3) * 1024 + r * 64 + lane];
        }
        __syncthreads();
    }
    const int j = grp;
    const int col = n0 + 32 * j + l31;
    const bool cok = col < N;
    const float bv = cok ? bias[col] : 0.f;
    if ((N & 3) == 0 && m0 + BM <= M && n0 + BN <= N) {
        float* patch = xch + wave * (32 * 32);                  // 8 x 4 KiB = 32 KiB > As+Bs: use 2 rounds of 4 waves
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (grp == round) {
                float* pp = xch + (wave & 3) * (32 * 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = mine[r] + bv;
                    pp[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = v > 0.f ? v : 0.f;
                }
                const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = prow + 8 * p;
                    *reinterpret_cast<f32x4*>(&Y[(long long)(m0 + wm_off + row) * N + n0 + 32 * j + 4 * pc4]) =
                        *reinterpret_cast<const f32x4*>(&pp[row * 32 + 4 * pc4]);
                }
            }
            __syncthreads();
        }
        (void)patch;
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm_off + 4 * half + (r & 3) + 8 * (r >> 2);
        const float v = mine[r] + bv;
        if (cok && row < M) Y[(long long)row * N + col] = v > 0.f ? v : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ host
static void fill(std::vector<float>& h, unsigned long long seed, float scale) {
    unsigned long long st = seed;
    for (auto& x : h) {
        float v = -2.0f;
        for (int q = 0; q < 4; ++q) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            v += (float)(st >> 40) * (1.0f / 16777216.0f);
        }
        x = v * 1.7f * scale;
    }
}

struct Prob {
    int M, N, K;
    float *X, *W, *b, *Y;
    std::vector<float> hX, hW, hb;
};

static double check(Prob& p) {             // sampled fp64 reference
    std::vector<float> hY((size_t)p.M * p.N);
    hipMemcpy(hY.data(), p.Y, hY.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    unsigned long long st = 12345;
    for (int s = 0; s < 4000; ++s) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        int m = (int)((st >> 20) % p.M), n = (int)((st >> 44) % p.N);
        if (s < 64) m = p.M - 1 - s % 3, n = p.N - 1 - (s >> 2) % 16 % p.N;
        double acc = p.hb[n];
        for (int k = 0; k < p.K; ++k) acc += (double)p.hX[(size_t)m * p.K + k] * (double)p.hW[(size_t)n * p.K + k];
        if (acc < 0) acc = 0;
        const double e = fabs(acc - (double)hY[(size_t)m * p.N + n]);
        if (e > worst) worst = e;
    }
    return worst;
}

template <typename F>
static float timeit(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // the chip drops its clock within milliseconds of idling (the fp64 check between two measurements is enough) and
    // takes tens of ms of load to come back: 150 ms of the same kernel before the timed launches
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float one = 0;
    hipEventElapsedTime(&one, e0, e1);
    const int warm = (int)(150.0f / (one > 1e-3f ? one : 1e-3f)) + 1;
    for (int i = 0; i < warm; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

// calibration: N dependent MFMAs on ONE accumulator (64 cycles each, issue = latency) -> shader cycles by construction
__global__ __launch_bounds__(256) void mfma_chain(float* out, unsigned long long* t, int n) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.f) out[0] = s;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = t1 - t0; t[blockIdx.x * 2 + 1] = r1 - r0; }
}

static void calibrate(int blocks) {
    unsigned long long* d;
    float* o;
    hipMalloc(&d, blocks * 16);
    hipMalloc(&o, 64);
    const int n = 1 << 16;
    for (int rep = 0; rep < 3; ++rep) mfma_chain<<<blocks, 256>>>(o, d, n);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
    double mt = 0, rt = 0;
    for (int b = 0; b < blocks; ++b) { mt += h[2 * b]; rt += h[2 * b + 1]; }
    printf("calibration %4d blocks x 4 waves: %d dependent MFMAs = %d MFMA cycles; s_memtime %.0f ticks (%.3f per MFMA cycle), "
           "s_memrealtime %.0f ticks -> MFMA clock %.3f GHz if realtime = 100 MHz\n", blocks, n, n * 64, mt / blocks,
           mt / blocks / (n * 64.0), rt / blocks, n * 64.0 / (rt / blocks) * 0.1);
    hipFree(d);
    hipFree(o);
}

static void product_trace(Prob& p) {
    void* h = dlopen("deep-tracking-control_amd/tools/_bin/libdtc_hip_trace.so", RTLD_NOW);
    if (!h) { printf("trace library missing\n"); return; }
    fwd_fn fwd = (fwd_fn)dlsym(h, "dtc_linear_fwd");
    typedef int (*set_fn)(unsigned long long*);
    set_fn set = (set_fn)dlsym(h, "dtc_debug_set_trace");
    const int rt = (p.M + 127) / 128, grid = grid_for(rt, (p.N + 63) / 64);
    unsigned long long* d;
    hipMalloc(&d, (size_t)grid * 32);
    hipMemset(d, 0, (size_t)grid * 32);
    set(d);
    DtcSegMat xs;
    memset(&xs, 0, sizeof(xs));
    xs.nseg = 1; xs.cols = p.K;
    xs.seg[0].ptr = p.X; xs.seg[0].ld = p.K; xs.seg[0].col0 = 0; xs.seg[0].width = p.K; xs.seg[0].rows = p.M;
    for (int i = 0; i < 1500; ++i) fwd(&xs, p.W, p.b, p.Y, p.N, p.M, p.N, p.K, DTC_ACT_RELU, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t((size_t)grid * 4);
    hipMemcpy(t.data(), d, t.size() * 8, hipMemcpyDeviceToHost);
    double pro = 0, loop = 0, epi = 0;
    int n = 0;
    for (int b = 0; b < grid; ++b) {
        const unsigned long long* e = &t[(size_t)b * 4];
        if (!e[3]) continue;
        pro += (double)(e[1] - e[0]); loop += (double)(e[2] - e[1]); epi += (double)(e[3] - e[2]); ++n;
    }
    printf("[product] trace M=%d N=%d K=%d: %d blocks; phases per block (cycles): prologue %.0f  K loop %.0f  epilogue %.0f  total %.0f\n",
           p.M, p.N, p.K, n, pro / n, loop / n, epi / n, (pro + loop + epi) / n);
    set(nullptr);
    hipFree(d);
}

template <int OCC>
static void trace_run(Prob& p) {
    const int rt = (p.M + 127) / 128, grid = grid_for(rt, (p.N + 63) / 64);
    printf("[occ %d] ", OCC);
    unsigned long long* d;
    hipMalloc(&d, (size_t)grid * 8 * 8);
    hipMemset(d, 0, (size_t)grid * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &d, sizeof(d));
    for (int i = 0; i < 1500; ++i) fwd_v2<64, OCC, 1><<<grid, 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)grid * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    {   // phases per block and per-XCD spans, in shader cycles (XCD counters are not synchronised with each other)
        double pro = 0, loop = 0, epi = 0;
        unsigned long long xmin[8], xmax[8];
        for (int x = 0; x < 8; ++x) { xmin[x] = ~0ull; xmax[x] = 0; }
        int n = 0;
        for (int b = 0; b < grid; ++b) {
            const unsigned long long* e = &h[(size_t)b * 6];
            if (e[1] == 0) continue;
            const unsigned long long tp = h[(size_t)grid * 6 + (size_t)b * 2], tl = h[(size_t)grid * 6 + (size_t)b * 2 + 1];
            pro += (double)(tp - e[0]); loop += (double)(tl - tp); epi += (double)(e[1] - tl);
            const int x = (int)e[5] & 7;
            if (e[0] < xmin[x]) xmin[x] = e[0];
            if (e[1] > xmax[x]) xmax[x] = e[1];
            ++n;
        }
        printf("  phases per block (cycles): prologue %.0f  K loop %.0f  epilogue %.0f;  per-XCD span:", pro / n, loop / n, epi / n);
        for (int x = 0; x < 8; ++x) printf(" %llu", xmax[x] - xmin[x]);
        printf("\n");
    }
    unsigned long long tmin = ~0ull, tmax = 0, rmin = ~0ull, rmax = 0;
    double dur = 0, rdur = 0;
    int nb = 0;
    std::vector<int> per_cu(8 * 256, 0);
    for (int b = 0; b < grid; ++b) {
        const unsigned long long* e = &h[(size_t)b * 6];
        if (e[1] == 0) continue;
        ++nb;
        if (e[0] < tmin) tmin = e[0];
        if (e[1] > tmax) tmax = e[1];
        if (e[2] < rmin) rmin = e[2];
        if (e[3] > rmax) rmax = e[3];
        dur += (double)(e[1] - e[0]);
        rdur += (double)(e[3] - e[2]);
        const unsigned hw = (unsigned)e[4];
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;       // HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
        per_cu[((int)e[5] & 7) * 256 + ((se * 2 + sh) * 16 + cu) % 256] += 1;
    }
    const double span_us = (double)(rmax - rmin) / 100.0, ghz = dur / rdur * 0.1;
    printf("trace M=%d N=%d K=%d: %d blocks, span %.1f us, shader clock %.3f GHz (s_memtime / s_memrealtime over all blocks), mean block %.0f cycles (%.1f us)\n", p.M, p.N,
           p.K, nb, span_us, ghz, dur / nb, dur / nb / ghz * 1e-3);
    int hist[16] = {0}, used = 0;
    for (int c : per_cu) if (c) { ++used; hist[c < 15 ? c : 15]++; }
    printf("  CUs used %d; blocks-per-CU histogram:", used);
    for (int i = 1; i < 16; ++i) if (hist[i]) printf(" %dx%d", i, hist[i]);
    printf("\n");
    // start-time profile: how many blocks start in each 5 us bucket
    printf("  block starts per 10us bucket:");
    std::vector<int> st(64, 0), en(64, 0);
    for (int b = 0; b < grid; ++b) {
        const unsigned long long* e = &h[(size_t)b * 6];
        if (e[1] == 0) continue;
        int bs = (int)((double)(e[2] - rmin) / 1000.0), be = (int)((double)(e[3] - rmin) / 1000.0);
        if (bs < 64) st[bs]++;
        if (be < 64) en[be]++;
    }
    for (int i = 0; i < 16; ++i) printf(" %d/%d", st[i], en[i]);
    printf("\n");
    hipFree(d);
}

int main(int argc, char** argv) {
    {   // the product library (same directory layout as the repo): dtc_linear_fwd under the lab's clock
        void* h = dlopen("deep-tracking-control_amd/dtc_amd/lib/libdtc_hip.so", RTLD_NOW);
        if (h) g_prod_fwd = (fwd_fn)dlsym(h, "dtc_linear_fwd");
        printf("product library: %s\n", g_prod_fwd ? "loaded" : "NOT loaded");
    }
    const int shapes[][3] = {{24576, 512, 512}, {24576, 512, 693}, {24576, 693, 512}, {24576, 512, 752}, {24576, 512, 584},
                             {24576, 256, 512}, {24576, 128, 256}, {24576, 128, 265}, {24576, 64, 531}, {24576, 64, 128},
                             {24576, 35, 64}, {24576, 12, 128}, {1470, 512, 512}};
    for (auto& s : shapes) {
        Prob p;
        p.M = s[0]; p.N = s[1]; p.K = s[2];
        p.hX.resize((size_t)p.M * p.K); p.hW.resize((size_t)p.N * p.K); p.hb.resize(p.N);
        fill(p.hX, 88172645463325252ull, 1.0f);
        fill(p.hW, 1234567891234567ull, 1.0f / sqrtf((float)p.K));
        fill(p.hb, 99887766554433ull, 0.3f);
        hipMalloc(&p.X, p.hX.size() * 4); hipMalloc(&p.W, p.hW.size() * 4); hipMalloc(&p.b, p.hb.size() * 4);
        hipMalloc(&p.Y, (size_t)p.M * p.N * 4);
        hipMemcpy(p.X, p.hX.data(), p.hX.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(p.W, p.hW.data(), p.hW.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(p.b, p.hb.data(), p.hb.size() * 4, hipMemcpyHostToDevice);
        const double fl = 2.0 * p.M * p.N * (double)p.K;
        const int rt = (p.M + 127) / 128;
        auto report = [&](const char* tag, float ms) {
            printf("M=%5d N=%3d K=%3d %-18s %8.1f us %7.1f TF  maxerr %.2e\n", p.M, p.N, p.K, tag, ms * 1e3, fl / ms * 1e-9, check(p));
            fflush(stdout);
        };
        const int reps = 100;
        if (argc > 1 && !strcmp(argv[1], "trace")) {
            if (&s == &shapes[0]) { calibrate(1); calibrate(256); calibrate(1024); }
            if (p.N >= 256 && p.M > 2000) { trace_run<4>(p); trace_run<6>(p); product_trace(p); }
            hipFree(p.X); hipFree(p.W); hipFree(p.b); hipFree(p.Y);
            continue;
        }
        for (int round = 0; round < 2; ++round) {
            if (g_prod_fwd) {
                DtcSegMat xs;
                memset(&xs, 0, sizeof(xs));
                xs.nseg = 1; xs.cols = p.K;
                xs.seg[0].ptr = p.X; xs.seg[0].ld = p.K; xs.seg[0].col0 = 0; xs.seg[0].width = p.K; xs.seg[0].rows = p.M;
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("product fwd", timeit([&] { g_prod_fwd(&xs, p.W, p.b, p.Y, p.N, p.M, p.N, p.K, DTC_ACT_RELU, nullptr); }, reps));
            }
            if (p.N > 32) {
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("v1 BN64", timeit([&] { fwd_v1<64><<<grid_for(rt, (p.N + 63) / 64), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("v2 BN64 occ3", timeit([&] { fwd_v2<64, 3, 0><<<grid_for(rt, (p.N + 63) / 64), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("v2 BN64 occ4", timeit([&] { fwd_v2<64, 4, 0><<<grid_for(rt, (p.N + 63) / 64), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("v2 occ4 wide-st", timeit([&] { fwd_v2<64, 4, 3><<<grid_for(rt, (p.N + 63) / 64), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("v2 occ6 wide-st", timeit([&] { fwd_v2<64, 6, 3><<<grid_for(rt, (p.N + 63) / 64), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("v2 occ4 pin wide", timeit([&] { fwd_v2<64, 4, 4><<<grid_for(rt, (p.N + 63) / 64), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                report("v2 occ5 pin wide", timeit([&] { fwd_v2<64, 5, 4><<<grid_for(rt, (p.N + 63) / 64), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                if (p.K % 16 == 0 && p.N % 128 == 0 && p.M % 128 == 0) {
                    hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                    report("v4 128x128 occ3", timeit([&] { fwd_v4<3><<<grid_for(p.M / 128, p.N / 128), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                    hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                    report("v4 128x128 pin3", timeit([&] { fwd_v4<3, true><<<grid_for(p.M / 128, p.N / 128), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                    hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                    report("v4 128x128 pin2", timeit([&] { fwd_v4<2, true><<<grid_for(p.M / 128, p.N / 128), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                }
                if (p.K % 16 == 0) {
                    hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                    report("v3 512thr occ2", timeit([&] { fwd_v3<2><<<grid_for(rt, (p.N + 63) / 64), 512>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                    hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
                    report("v3 512thr occ3", timeit([&] { fwd_v3<3><<<grid_for(rt, (p.N + 63) / 64), 512>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                }
                for (int cap = 6; cap <= 6; ++cap) {         // blocks per CU capped through dynamic LDS: 160 KiB / cap - static 24 KiB
                    const int dyn = cap == 6 ? 0 : (160 * 1024 / cap - 24 * 1024 - 512) & ~255;
                    char tag[32];
                    snprintf(tag, sizeof(tag), "v2 occ6 cap%d", cap);
                    hipFuncSetAttribute((const void*)fwd_v2<64, 6, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
                    report(tag, timeit([&] { fwd_v2<64, 6, 0><<<grid_for(rt, (p.N + 63) / 64), 256, dyn>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
                }
            }
            hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
            report("v1 BN32", timeit([&] { fwd_v1<32><<<grid_for(rt, (p.N + 31) / 32), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
            hipMemset(p.Y, 0, (size_t)p.M * p.N * 4);
            report("v2 BN32", timeit([&] { fwd_v2<32, 4, 0><<<grid_for(rt, (p.N + 31) / 32), 256>>>(p.X, p.W, p.b, p.Y, p.M, p.N, p.K); }, reps));
        }
        hipFree(p.X); hipFree(p.W); hipFree(p.b); hipFree(p.Y);
    }
    return 0;
}
