// Dense layers on block-scaled two-term fp16 operand images (round 5; format and rationale: csrc/h2i_core.hpp):
// the nn.Linear products of rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349 under ppo.py:197-218, 252, 265, 289, 333.
//
//   h2i_pack_kernel     fp32 (segmented / row-gathered: the rollout storage, narrow hand-over tensors) -> image, exponent per row block
//   h2i_wpack_kernel    weights (W or W^T, any column window / segment walk) -> image, exponent per 128 x 128 block; one launch per phase
//   linear_h2i_kernel   Y = act(X W^T + b) | dX = (dZ W) act' | the terrain decoder's output layer fused with its MSE:
//                       BOTH operands by LDS-DMA (no conversion, no operand registers, no ds_write in the K loop: 4 LDS-DMA pieces +
//                       8 ds_read_b128 + 12 MFMAs per wave and 16-k stage), accumulator rows rescaled at the 128-column block borders,
//                       results as fp32 and / or as the image the next consumer reads (per-row exponents chosen in the epilogue)
//   h2i_unpack_kernel   image -> fp32 (tests, debugging)
// Block tile 128 x 128 x 16, 2 x 2 waves of 64 x 64, double-buffered LDS (48 KiB + 4.25 KiB of exponent deltas), 3 workgroups per CU.
#include <type_traits>

#include "h2i_core.hpp"

namespace {

constexpr int EPI_FWD = 0, EPI_DGRAD = 1, EPI_MSE = 2;
#ifndef DTC_H2I_ABL
#define DTC_H2I_ABL 0         // timing ablations of the K loop (tools/jobs/r6_h2i_ablate.sh; WRONG results): 2 no MFMA, 4 no fragment reads, 8 no transfers
#endif
#ifndef DTC_H2I_SELF
#define DTC_H2I_SELF 1        // 128-row tiles: the fragments of the tile a wave fetched itself are read ABOVE the stage barrier (h2i_tile)
#endif
#ifndef DTC_H2I_NT_STORES
#define DTC_H2I_NT_STORES 0   // image stores of the GEMM epilogue with the non-temporal hint (measured: DESIGN.md 4.5)
#endif
#ifndef DTC_H2I_PROBE
#define DTC_H2I_PROBE 0       // timing ladder of the epilogue (tools/jobs/r5_epi_ladder.sh): 1..5 drop its parts from the end (WRONG results)
#endif
constexpr int MAX_TB = 16;           // exponent blocks along the reduction (sum over the row operand's segments): 2048 columns

// ---- fp32 -> image ---------------------------------------------------------------------------------------------------------------
// block = (row tile, k block); thread = (row r = tid & 127, k half h = tid >> 7): 8 stages x 8 consecutive columns in registers,
// row maximum (with the other half's thread through LDS), exponent, split, 16-byte pieces.  A wave holds 64 consecutive rows and ONE
// column group, so the segment a column group lies in is wave-uniform.
struct PackArgs {
    SegMatDev X;
    int M, K, stages, kbs;
    u32x4* img;
    int* exps;
};
__global__ __launch_bounds__(256) void h2i_pack_kernel(const PackArgs P) {
    const int tr = blockIdx.x / P.kbs, kb = blockIdx.x - tr * P.kbs;
    const int tid = threadIdx.x, r = tid & 127, h = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int m = tr * 128 + r;
    const SegMatDev& X = P.X;
    float v[HI_KB][8];
    u32 mx = 0u;
    const long long src_row = (m < P.M && X.gathers > 0) ? X.idx[m] : (long long)m;
#pragma unroll
    for (int s = 0; s < HI_KB; ++s) {
        const int c = (kb * HI_KB + s) * 16 + 8 * h;          // first of this thread's 8 columns (wave-uniform)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[s][e] = 0.f;
        if (c >= P.K) continue;
        const int sg = find_seg(X, c);
        const SegDev sd = X.s[sg];
        const long long row = sd.gather ? src_row : (long long)m;
        if (c + 8 <= sd.start + sd.width) {                    // the 8 columns lie in one block: two 16-byte loads (dword-aligned)
            const rsrc_t res = make_rsrc_bytes(sd.ptr, (long long)sd.rows * sd.ld * 4);
            const u32 off = m < P.M ? (u32)((row * sd.ld + sd.col0 + (c - sd.start)) * 4) : INVALID;
            const f32x4 a = bload4(res, off, 0u), b = bload4(res, off, 16u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[s][e] = a[e];
                v[s][4 + e] = b[e];
            }
        } else {                                               // a block border (or the matrix's right edge) inside the group
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ce = c + e;
                if (ce < P.K && m < P.M) {
                    const SegDev se = X.s[find_seg(X, ce)];
                    v[s][e] = se.ptr[(se.gather ? src_row : (long long)m) * se.ld + se.col0 + (ce - se.start)];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const u32 b = finite_bits(v[s][e]);
            mx = b > mx ? b : mx;
        }
    }
    __shared__ u32 rm[2][128];
    rm[h][r] = mx;
    __syncthreads();
    mx = rm[0][r] > rm[1][r] ? rm[0][r] : rm[1][r];
    const int e = hi_exp(mx);
    if (h == 0) P.exps[(tr * P.kbs + kb) * 128 + r] = e;
    const int ee = e == HI_EZERO ? 0 : e;
#pragma unroll
    for (int s = 0; s < HI_KB; ++s) {
        const int st = kb * HI_KB + s;
        if (st >= P.stages) break;
        const f32x4 q[2] = {f32x4{v[s][0], v[s][1], v[s][2], v[s][3]}, f32x4{v[s][4], v[s][5], v[s][6], v[s][7]}};
        const HiPiece pc = hi_split8(q, ee);
        u32x4* chunk = P.img + ((long long)tr * P.stages + st) * (HI_CHUNK / 16);
        chunk[rslot(r, h)] = pc.p[0];
        chunk[256 + rslot(r, h)] = pc.p[1];
    }
}

// ---- weights -> image ------------------------------------------------------------------------------------------------------------
// Image rows = `nr` rows of the operand starting at r0; reduction = the concatenation of up to 4 column ranges, each padded to whole
// stages and carrying its own exponent blocks (the row operand's segments are separate images with their own blocks).  trans: the
// operand is W^T (element (row, c) = W[c * ld + r0 + row]) -- the data gradient's weight image, rows = a window of W's columns.
// block = (job, row tile, exponent block); ONE exponent per block (the weight gradient never reads these images, and a 128 x 128 block
// of a trained layer spans a few octaves).
constexpr int WP_MAX_JOBS = 40;
struct WpackJob {
    const float* W;
    long long ld;
    u32x4* img;
    int* exps;               // [row tiles][tblocks]
    int trans, nrows, r0[2], nr[2], nseg, c0[4], cw[4];
    int tstages, tblocks, block_end;      // block_end: running sum of (row tiles x tblocks) over the jobs
};
struct WpackGroup {
    int count;
    WpackJob job[WP_MAX_JOBS];
};
// WP_SQ stage groups per block (threads = 256 x WP_SQ): thread (row r, k half h, stage group sq) holds HI_KB / WP_SQ stages.  The launch
// sits alone on the chip between the optimiser step and the next forward (one block per CU), so what counts is a thread's chain:
// loads -> block maximum -> split -> stores.  Measured per launch / per bench step (tools/jobs/r5_wp.sh, three interleaved rounds):
// 1 group 28.5 us / 49.5 ms, 2 groups 22.0 us / 49.1 ms, 4 groups 19.9 us / 49.25 ms: 2 is the default (-DDTC_WPACK_SQ=1 | 4).
#ifndef DTC_WPACK_SQ
#define DTC_WPACK_SQ 2
#endif
constexpr int WP_SQ = DTC_WPACK_SQ, WP_ST = HI_KB / WP_SQ;
__global__ __launch_bounds__(256 * WP_SQ) void h2i_wpack_kernel(const WpackGroup G) {
    int j = 0, b = blockIdx.x;
    while (j < G.count - 1 && b >= G.job[j].block_end) ++j;
    if (j > 0) b -= G.job[j - 1].block_end;
    const WpackJob& J = G.job[j];
    const int ct = b / J.tblocks;
    int gb = b - ct * J.tblocks, sg = 0, gs0 = 0;              // gb -> (segment, block inside it); gs0: first global stage of the segment
    while (sg + 1 < J.nseg && gb >= (int)hi_kblocks(J.cw[sg])) {
        gb -= (int)hi_kblocks(J.cw[sg]);
        gs0 += (int)hi_stages(J.cw[sg]);
        ++sg;
    }
    const int tid = threadIdx.x, r = tid & 127, h = (tid >> 7) & 1, sq = tid >> 8;
    const int t0 = (J.nr[0] + 127) >> 7, rg = (J.nrows > 1 && ct >= t0) ? 1 : 0;        // the row range this tile lies in
    const int row = (ct - (rg ? t0 : 0)) * 128 + r, nrg = J.nr[rg], src0 = J.r0[rg];
    const int nst = (int)hi_stages(J.cw[sg]);
    float v[WP_ST][8];
    u32 mx = 0u;
    // W as stored (not transposed): a thread's 8 columns of a stage are consecutive in memory -- two 16-byte loads where the row allows
    // it (uniform: leading dimension and first column multiples of 4, 16-byte aligned base)
    const bool vec = !J.trans && (J.ld & 3) == 0 && (J.c0[sg] & 3) == 0 && (reinterpret_cast<unsigned long long>(J.W) & 15ull) == 0;
#pragma unroll
    for (int s = 0; s < WP_ST; ++s) {
        const int cb = (gb * HI_KB + sq * WP_ST + s) * 16 + 8 * h;
        if (vec && row < nrg && cb + 8 <= J.cw[sg]) {
            const f32x4* src = reinterpret_cast<const f32x4*>(J.W + (long long)(src0 + row) * J.ld + J.c0[sg] + cb);
            const f32x4 a = src[0], b2 = src[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[s][e] = a[e];
                v[s][4 + e] = b2[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = cb + e;
                const bool ok = row < nrg && c < J.cw[sg];
                const long long cc = J.c0[sg] + c;
                v[s][e] = ok ? (J.trans ? J.W[cc * J.ld + src0 + row] : J.W[(long long)(src0 + row) * J.ld + cc]) : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const u32 bb = finite_bits(v[s][e]);
            mx = bb > mx ? bb : mx;
        }
    }
    mx = wave_max_u32(mx);
    __shared__ u32 red[4 * WP_SQ];
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 4 * WP_SQ; ++w) mx = red[w] > mx ? red[w] : mx;
    const int e = hi_exp(mx);
    int gblock = gb;                                             // index of this block in the image's walk
    for (int i = 0; i < sg; ++i) gblock += (int)hi_kblocks(J.cw[i]);
    if (tid == 0) J.exps[ct * J.tblocks + gblock] = e;
    const int ee = e == HI_EZERO ? 0 : e;
#pragma unroll
    for (int s = 0; s < WP_ST; ++s) {
        const int st = gb * HI_KB + sq * WP_ST + s;
        if (st >= nst) break;
        const f32x4 q[2] = {f32x4{v[s][0], v[s][1], v[s][2], v[s][3]}, f32x4{v[s][4], v[s][5], v[s][6], v[s][7]}};
        const HiPiece pc = hi_split8(q, ee);
        u32x4* chunk = J.img + ((long long)ct * J.tstages + gs0 + st) * (HI_CHUNK / 16);
        chunk[rslot(r, h)] = pc.p[0];
        chunk[256 + rslot(r, h)] = pc.p[1];
    }
}

// ---- image -> fp32 (tests) -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void h2i_unpack_kernel(const u32x4* __restrict__ img, const int* __restrict__ exps, int M, int K, int stages,
                                                         int kbs, float* __restrict__ out, long long ld) {
    const int tr = blockIdx.x / stages, st = blockIdx.x - tr * stages;
    const int r = threadIdx.x & 127, h = threadIdx.x >> 7;
    const int m = tr * 128 + r;
    if (m >= M) return;
    const u32x4* chunk = img + (long long)blockIdx.x * (HI_CHUNK / 16);
    const u32x4 hi = chunk[rslot(r, h)], lo = chunk[256 + rslot(r, h)];
    int e = exps[(tr * kbs + st / HI_KB) * 128 + r];
    e = e == HI_EZERO ? 0 : e;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = st * 16 + 8 * h + 2 * q;
        const float a = __builtin_ldexpf(f16_lo(hi[q]) + f16_lo(lo[q]), -e), b = __builtin_ldexpf(f16_hi(hi[q]) + f16_hi(lo[q]), -e);
        if (c < K) out[(long long)m * ld + c] = a;
        if (c + 1 < K) out[(long long)m * ld + c + 1] = b;
    }
}

// ---- the GEMM ----------------------------------------------------------------------------------------------------------------------
struct HSeg {
    const u32x4* img;
    const int* exps;
    u32 bytes;               // of the chunks (LDS-DMA descriptor bound)
    int stages, kbs;
};
struct HOperand {            // row operand: up to 4 images over the same M rows, side by side along the reduction
    int nseg, total, tblocks;
    HSeg s[4];
};
struct HOut {
    u32x4* img;              // NULL: no image of the result
    int* exps;
    int stages, kbs;
};
struct MseEpiH {
    const float* target;
    const long long* tidx;
    long long ldt, target_bytes;
    int tcol0;
    float scale;
    double* part;
};
struct DgradEpiH {
    SegMatDev dX;            // fp32 destination over the computed column window (has_dx == 0: none)
    int has_dx, wide_segs;
    const float* add;        // fp32 [M, ld_add] added to the product before anything is stored (may be NULL)
    long long ld_add;
    const float* Xs;         // saved post-activation output (act != none without a sign record)
    long long ldxs;
    const unsigned short* rmask;
    int ldm;
};

__device__ __forceinline__ void store8(float* base, long long ld, int row, int col, int M, int N, bool wide, const f32x4 (&v)[2]) {
    if (row >= M) return;
    float* p = base + (long long)row * ld + col;
    if (wide && col + 8 <= N) {
        *reinterpret_cast<f32x4*>(p) = v[0];
        *reinterpret_cast<f32x4*>(p + 4) = v[1];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (col + e < N) p[e] = v[e >> 2][e & 3];
    }
}
__device__ __forceinline__ void load8(const float* base, long long ld, int row, int col, int M, int N, bool wide, f32x4 (&v)[2]) {
    v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row >= M) return;
    const float* p = base + (long long)row * ld + col;
    if (wide && col + 8 <= N) {
        v[0] = *reinterpret_cast<const f32x4*>(p);
        v[1] = *reinterpret_cast<const f32x4*>(p + 4);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (col + e < N) v[e >> 2][e & 3] = p[e];
    }
}


// LDS of the GEMM kernels, at namespace scope: every instance of the tile code shares ONE set.
// Separate objects per stage buffer: an LDS-DMA into one cannot alias the fragment reads of the other (see linear_s3_kernel).
// THREE stage buffers: the LDS-DMA of stage s + 2 is issued while stage s computes.  (One stage ahead -- the first version of this
// kernel -- left the launches with few workgroups per CU waiting at their barriers: 24576 x 256 x 512 48.8 -> 40.7 us, the weight
// gradients' twin 162 -> 150 us per three-layer bucket; the 768-workgroup launches are bound by LDS traffic + matrix pipe either way.)
__shared__ __attribute__((aligned(16))) u32x2 Xs0[2][BM * 4];
__shared__ __attribute__((aligned(16))) u32x2 Xs1[2][BM * 4];
__shared__ __attribute__((aligned(16))) u32x2 Xs2[2][BM * 4];
__shared__ __attribute__((aligned(16))) u32x2 Ws0[2][128 * 4];
__shared__ __attribute__((aligned(16))) u32x2 Ws1[2][128 * 4];
__shared__ __attribute__((aligned(16))) u32x2 Ws2[2][128 * 4];
__shared__ __attribute__((aligned(16))) short Dt[MAX_TB + 1][128];    // exponent deltas per block border and row (|e| <= 2 x 113: int16); [tblocks]: the final scale

// ---- device code of the GEMM kernels.  The descriptors are read through the CONSTANT address space, in place from the kernel-argument
// segment: invariant scalar loads, and no private copy of the by-value argument (binding a generic reference to it copies the struct to
// scratch as soon as one of its arrays is indexed at run time).  The host pass of hipcc does not convert between address spaces
// implicitly, so these bodies exist for the device pass only.
#define AS4 __attribute__((address_space(4)))
#ifdef __HIP_DEVICE_COMPILE__
template <class OP>
__device__ __forceinline__ HSeg hseg_at(const OP& A, int i) {
    HSeg s = A.s[0];
    if (i == 1) s = A.s[1];
    if (i == 2) s = A.s[2];
    if (i == 3) s = A.s[3];
    return s;
}
// segment i of a descriptor by static selects (a dynamic index into a by-value kernel argument would pin the whole struct to scratch)
template <class SM>
__device__ __forceinline__ int find_seg_t(const SM& X, int k) {
    int s = 0;
    if (X.nseg > 1 && k >= X.s[1].start) s = 1;
    if (X.nseg > 2 && k >= X.s[2].start) s = 2;
    if (X.nseg > 3 && k >= X.s[3].start) s = 3;
    return s;
}
template <class SM>
__device__ __forceinline__ SegDev seg_at(const SM& X, int i) {
    SegDev s = X.s[0];
    if (i == 1) s = X.s[1];
    if (i == 2) s = X.s[2];
    if (i == 3) s = X.s[3];
    return s;
}

#endif  // __HIP_DEVICE_COMPILE__

// Everything one 128 x 128 result tile needs (N: result columns -- EPI_DGRAD: the computed window of the layer's input columns; wide bit 0:
// Y takes 16-byte accesses, bit 1: Xs does, bit 2: add does)
struct TileArgs {
    HOperand A;
    const u32x4* wimg;
    const int* wexps;
    u32 wimg_bytes;
    int M, N, act, wide, ldwm;
    const float* bias;
    float* Y;
    long long ldy;
    HOut yo;
    unsigned short* wmask;
    DgradEpiH dg;
};

// One result tile (tr, tc): K loop + epilogue.  `slot`: index of the tile's loss partial (EPI_MSE) / trace record.  Every thread of the
// workgroup calls it; on return the tile's global stores are ISSUED (not necessarily complete).
#ifdef __HIP_DEVICE_COMPILE__
typedef const AS4 TileArgs CTileArgs;
// TM = 2: 128-row tiles (2 x 2 waves of 64 x 64); TM = 1: 64-row tiles (2 x 2 waves of 32 x 64: half a chunk of the row operand per
// stage) for the launches whose 128-row tiles would leave CUs without a workgroup (the 128-column layers: 192 tiles on 256 CUs).
template <int EPI, int TM>
__device__ __forceinline__ void h2i_tile(CTileArgs& L, const MseEpiH& mse, const int tr, const int tc, const int slot,
                                         unsigned long long* __restrict__ trace) {
    const auto& A = L.A;
    const u32x4* __restrict__ wimg = L.wimg;
    const int* __restrict__ wexps = L.wexps;
    const u32 wimg_bytes = L.wimg_bytes;
    const float* __restrict__ bias = L.bias;
    float* __restrict__ Y = L.Y;
    const long long ldy = L.ldy;
    const auto& yo = L.yo;
    const int M = L.M, N = L.N, act = L.act, wide = L.wide, ldwm = L.ldwm;
    unsigned short* __restrict__ wmask = L.wmask;
    const auto& dg = L.dg;
    constexpr int BN = 128, WN = 2, TN = 2, BMT = 64 * TM;
#define XS(b) ((b) == 0 ? Xs0 : (b) == 1 ? Xs1 : Xs2)
#define WS(b) ((b) == 0 ? Ws0 : (b) == 1 ? Ws1 : Ws2)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (trace && tid == 0) {                              // debug (dtc_h2i_trace): per-workgroup time stamps (100 MHz) and placement
        trace[4 * slot] = __builtin_amdgcn_s_memrealtime();
        // HW_REG_HW_ID (id 4: cu_id [11:8], sh_id [12], se_id [15:13]) | HW_REG_XCC_ID (id 20) << 32
        trace[4 * slot + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                              ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
    const int m0 = tr * BMT, n0 = tc * BN;
    const int ctile = m0 >> 7, crow0 = m0 & 127;                      // the 128-row chunk tile this tile lies in, its first row there
    const u32 xsub = (u32)crow0 * 32u;                                // ... = byte offset inside a plane (slot = 2 x row)
    const u32 xlane = (TM == 2 || tid < 128) ? 0u : INVALID;          // 64-row tiles: half a plane per stage (the upper lanes fetch nothing)
    const int wm_off = (wave / WN) * (32 * TM), wn_off = (wave % WN) * (32 * TN);
    // Which 32-row / 32-column block the wave's tile index i / j stands for.  128-row tiles (SELF): tile (0, 0) is the one whose
    // operand pieces THIS wave fetches (wave (wm, wn) copies rows 64 wm + 32 wn of the X stage and rows 64 wn + 32 wm of the W stage),
    // so its fragments can be read as soon as the wave's own LDS-DMA has landed -- above the stage barrier, whose wait then hides
    // their LDS latency (DTC_H2I_SELF=0 at build time: the plain order, tile (i, j) = block (i, j), all reads behind the barrier).
    constexpr bool SELF = TM == 2 && DTC_H2I_SELF != 0;
    const int swn = SELF ? (wave % WN) : 0, swm = SELF ? (wave / WN) : 0;
    const int ro[2] = {wm_off + 32 * swn, wm_off + 32 * (1 - swn)};      // (TM == 1 uses ro[0] only: swn = 0)
    const int co[2] = {wn_off + 32 * swm, wn_off + 32 * (1 - swm)};
    const int wpiece = SELF ? 2 * (wave % WN) + (wave / WN) : wave;       // the 32-row piece of the W stage this wave copies
    const int half = lane >> 5, l31 = lane & 31;
    const rsrc_t wres = make_rsrc_bytes(wimg, wimg_bytes);
    const u32 lane_off = (u32)(tid * 16);
    const u32 wlane_off = (u32)((wpiece * 64 + lane) * 16);

    // The K loop exists twice: the row operand as ONE image (every layer but the actor's first) keeps no descriptor state in the loop
    const int a_total = A.total, a_nseg = A.nseg, a_tblocks = A.tblocks;
    f32x16 acc[TM][TN];
    auto k_loop = [&](auto multi_c) {
    constexpr bool MULTI = decltype(multi_c)::value;
    // ---- loader cursor (one stage ahead of the MFMAs)
    int lseg = 0, lleft = A.s[0].stages, left = A.total;
    rsrc_t xres = make_rsrc_bytes(A.s[0].img, A.s[0].bytes);
    u32 xchunk = (u32)(ctile * A.s[0].stages) * (u32)HI_CHUNK + xsub, wchunk = (u32)(tc * A.total) * (u32)HI_CHUNK;
    auto load_stage = [&](auto nbc) {                   // next stage -> LDS[nbuf]; past the last stage: out-of-range lanes, zeros land
        constexpr int nbuf = decltype(nbc)::value;
        const u32 voff = lane_off | (left > 0 ? 0u : INVALID);
#if !(DTC_H2I_ABL & 8)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_void*)&XS(nbuf)[p][wave * 128], 16, voff | xlane, xchunk + p * HI_PLANE, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)&WS(nbuf)[p][wpiece * 128], 16, wlane_off | (left > 0 ? 0u : INVALID),
                                                     wchunk + p * HI_PLANE, 0, 0);
        }
#else
        (void)voff;
#endif
        xchunk += HI_CHUNK;
        wchunk += HI_CHUNK;
        --left;
        if (MULTI && --lleft == 0 && lseg + 1 < a_nseg) {        // (uniform) into the next image of the row operand
            ++lseg;
            const HSeg sg = hseg_at(A, lseg);
            lleft = sg.stages;
            xres = make_rsrc_bytes(sg.img, sg.bytes);
            xchunk = (u32)(ctile * sg.stages) * (u32)HI_CHUNK + xsub;
        }
    };
    load_stage(S0{});
    load_stage(S1{});

    // ---- exponent deltas: thread r < 128 walks the blocks of row r.  e(b) = row exponent + weight-block exponent; a block without
    // content (HI_EZERO) inherits its predecessor's (its products are zero whatever the scale)
    if (tid < BMT) {
        int prev_a = 0, prev_w = 0, prev = 0, b = 0;
        for (int i = 0; i < A.nseg; ++i) {
            const HSeg sg = hseg_at(A, i);
            const int* ex = sg.exps + (long long)ctile * sg.kbs * 128 + crow0 + tid;
            for (int k = 0; k < sg.kbs; ++k, ++b) {
                const int ea = ex[k * 128], ew = wexps[tc * A.tblocks + b];
                prev_a = ea == HI_EZERO ? prev_a : ea;
                prev_w = ew == HI_EZERO ? prev_w : ew;
                const int e = prev_a + prev_w;
                Dt[b][tid] = (short)(e - prev);
                prev = e;
            }
        }
        Dt[A.tblocks][tid] = (short)(-prev);
        // (the first block's "delta" Dt[0] = e(0) is never applied: the accumulators start at zero)
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // rows of this lane's accumulator registers: ro[i] + 4 half + 8 g + e, register 4 g + e
    // (The compiler cannot tell Dt from the stage buffers the LDS-DMA in flight writes and puts an s_waitcnt vmcnt(0) in front of one of the
    // merged table reads of a border: measured harmless -- the same reads by inline assembly, without that wait: 50.02 vs 50.05 ms per step.)
    auto rescale = [&](int b) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int2 d = *reinterpret_cast<const int2*>(&Dt[b][ro[i] + 4 * half + 8 * g]);      // four int16
                const int dv[4] = {(d.x << 16) >> 16, d.x >> 16, (d.y << 16) >> 16, d.y >> 16};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][4 * g + e] = __builtin_ldexpf(acc[i][j][4 * g + e], dv[e]);
            }
    };

    // ---- compute cursor
    int cseg = 0, cst = 0, blk = 0, done = 0;
    u32x4 sa[2], sb[2];                                   // SELF: the fragments of tile (0, 0) of the stage that runs next
    auto read_self = [&](auto bc) {
        constexpr int buf = decltype(bc)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#if DTC_H2I_ABL & 4
            sa[p] = u32x4{(u32)lane, 0x3c003c00u, (u32)p, 0u};
            sb[p] = u32x4{0x3c003c00u, (u32)lane, 0u, (u32)p};
#else
            sa[p] = reinterpret_cast<const u32x4*>(&XS(buf)[p][0])[rslot(ro[0] + l31, half)];
            sb[p] = reinterpret_cast<const u32x4*>(&WS(buf)[p][0])[rslot(co[0] + l31, half)];
#endif
        }
    };
    auto stage = [&](auto bc) {
        constexpr int buf = decltype(bc)::value;
        // (uniform) a block border: the rows change scale.  In front of the LDS-DMA: behind it the compiler would wait for the transfer
        // before the table reads (it cannot tell the two LDS objects apart)
        if (done > 0 && done < a_total && (cst & (HI_KB - 1)) == 0) {
            ++blk;
            rescale(blk);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_stage(std::integral_constant<int, (buf + 2) % 3>{});   // the pieces of stage s + 2 first (hipcc otherwise sinks them behind the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        u32x4 a[TM][2], b[TN][2];
#if DTC_H2I_ABL & 4
        auto rda = [&](int i, int p) { a[i][p] = u32x4{(u32)lane, 0x3c003c00u, (u32)p, (u32)i}; };
        auto rdb = [&](int j, int p) { b[j][p] = u32x4{0x3c003c00u, (u32)lane, (u32)j, (u32)p}; };
#else
        auto rda = [&](int i, int p) { a[i][p] = reinterpret_cast<const u32x4*>(&XS(buf)[p][0])[rslot(ro[i] + l31, half)]; };
        auto rdb = [&](int j, int p) { b[j][p] = reinterpret_cast<const u32x4*>(&WS(buf)[p][0])[rslot(co[j] + l31, half)]; };
#endif
#if DTC_H2I_ABL & 2
        struct P {          // no MFMA: the fragments still have to arrive
            static __device__ __forceinline__ f32x16 mfma(u32x4 x, u32x4 y, f32x16 c) {
                c[0] += __uint_as_float((x[0] ^ y[0]) & 0x007fffffu);
                return c;
            }
        };
#else
        using P = Prec<true>;
#endif
        if constexpr (SELF) {
            // tile (0, 0) from the fragments read above the barrier; the other three tiles' fragments are requested now and arrive
            // under its MFMAs.  Per tile the order of the terms is the plain path's (smallest first: lo hi', hi lo', hi hi'), so the
            // results are bit-identical
            a[0][0] = sa[0]; a[0][1] = sa[1]; b[0][0] = sb[0]; b[0][1] = sb[1];
            rda(1, 1); rda(1, 0); rdb(1, 0); rdb(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = P::mfma(a[0][1], b[0][0], acc[0][0]);
            acc[0][0] = P::mfma(a[0][0], b[0][1], acc[0][0]);
            acc[0][0] = P::mfma(a[0][0], b[0][0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = P::mfma(a[1][1], b[0][0], acc[1][0]);
            acc[0][1] = P::mfma(a[0][1], b[1][0], acc[0][1]);
            acc[1][1] = P::mfma(a[1][1], b[1][0], acc[1][1]);
            acc[1][0] = P::mfma(a[1][0], b[0][1], acc[1][0]);
            acc[0][1] = P::mfma(a[0][0], b[1][1], acc[0][1]);
            acc[1][1] = P::mfma(a[1][0], b[1][1], acc[1][1]);
            acc[1][0] = P::mfma(a[1][0], b[0][0], acc[1][0]);
            acc[0][1] = P::mfma(a[0][0], b[1][0], acc[0][1]);
            acc[1][1] = P::mfma(a[1][0], b[1][0], acc[1][1]);
        } else {
        rda(0, 1); rdb(0, 0);
        if constexpr (TM == 2) rda(1, 1);
        rda(0, 0); rdb(0, 1);
        if constexpr (TM == 2) rda(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        // smallest terms first: lo hi', hi lo', hi hi'
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][0] = P::mfma(a[i][1], b[0][0], acc[i][0]);
        __builtin_amdgcn_sched_barrier(0);
        rdb(1, 0); rdb(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][0] = P::mfma(a[i][0], b[0][1], acc[i][0]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][0] = P::mfma(a[i][0], b[0][0], acc[i][0]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][1] = P::mfma(a[i][1], b[1][0], acc[i][1]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][1] = P::mfma(a[i][0], b[1][1], acc[i][1]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][1] = P::mfma(a[i][0], b[1][0], acc[i][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        ++done;
        ++cst;
        if (MULTI && cst == hseg_at(A, cseg).stages && cseg + 1 < a_nseg) {
            ++cseg;
            cst = 0;
        }
        // stage s + 1 has landed (this wave's four newest transfers -- stage s + 2 -- may still be in flight: vmcnt(4), the other counters
        // untouched); then every wave's share has.  A raw barrier: __syncthreads() would wait for ALL transfers
        __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
        if constexpr (SELF) {                             // this wave's own pieces of stage s + 1 are in LDS: its tile (0, 0) fragments
            read_self(std::integral_constant<int, (buf + 1) % 3>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_barrier();
    };
    __syncthreads();
    if constexpr (SELF) read_self(S0{});
    for (int trip = (a_total + 2) / 3; trip > 0; --trip) {
        stage(S0{});
        stage(S1{});
        stage(std::integral_constant<int, 2>{});
    }
    __syncthreads();                                     // (the transfers past the last stage -- zeros -- have landed too)
    rescale(a_tblocks);                                  // back to the values themselves (2^-e of the last block, exact)
    };
    if (a_nseg > 1) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    if (trace && tid == 0) trace[4 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    // every wave is past its last fragment read and every LDS-DMA has landed (the barrier's wait): LDS becomes the patches
    float* patch = reinterpret_cast<float*>(wave < 2 ? &Xs0[0][0] : &Xs1[0][0]) + (wave & 1) * (32 * LDW);
    const bool full = (m0 + BMT <= M) && (n0 + BN <= N);

    double sq = 0.0;
    if constexpr (EPI == EPI_MSE) {
        // (the loss is formed behind the transposition below, where a lane holds 8 consecutive columns of a row)
    } else if constexpr (EPI == EPI_FWD && DTC_H2I_PROBE < 5) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + co[j] + l31;
            const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
            if (wmask && col < N) {                         // sign record of a ReLU layer (see linear_fwd_kernel); M % 128 == 0 (host)
                unsigned bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) bits |= acc[i][j][r] > 0.f ? (1u << r) : 0u;
                wmask[((long long)((m0 + ro[i]) >> 5) * 2 + half) * ldwm + col] = (unsigned short)bits;
            }
            if (act == DTC_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] <= 0.f ? 0.f : acc[i][j][r];      // (NaN passes through, as torch.relu)
            } else if (act == DTC_ACT_ELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] > 0.f ? acc[i][j][r] : expm1f(acc[i][j][r]);
            } else if (act != DTC_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = act_fwd(acc[i][j][r], act);
            }
        }
    } else {
        if (dg.rmask) {                                     // M % 128 == 0 (host); columns past the window read no record (results unused)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int mcol = n0 + co[j] + l31;
                const unsigned bits = mcol < N ? dg.rmask[((long long)((m0 + ro[i]) >> 5) * 2 + half) * dg.ldm + mcol] : 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = (bits >> r) & 1u ? acc[i][j][r] : 0.f;
            }
        }
    }

    // ---- through the patch: lane -> rows r16 + 16 q of the 32 x 32 tile, the 8 columns 8 c8 .. + 7; final values, fp32 stores, row maxima
    const int r16 = lane >> 2, c8 = lane & 3;
    f32x4 T[TM][TN][2][2];
    u32 mrow[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) mrow[i][0] = mrow[i][1] = 0u;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (DTC_H2I_PROBE < 4) patch_put(patch, acc[i][j], half, l31);
        const int col = n0 + co[j] + 8 * c8;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rl = r16 + 16 * q, row = m0 + ro[i] + rl;
            f32x4(&v)[2] = T[i][j][q];
            if (DTC_H2I_PROBE < 4) {
                v[0] = patch_get(patch, rl, 2 * c8);
                v[1] = patch_get(patch, rl, 2 * c8 + 1);
            } else {
                v[0] = f32x4{acc[i][j][8 * q], acc[i][j][8 * q + 1], acc[i][j][8 * q + 2], acc[i][j][8 * q + 3]};
                v[1] = f32x4{acc[i][j][8 * q + 4], acc[i][j][8 * q + 5], acc[i][j][8 * q + 6], acc[i][j][8 * q + 7]};
            }
            if constexpr (EPI == EPI_MSE) {
                // e = (acc + bias) - target[tidx[row], tcol0 + col];  dY = e * scale;  partial = sum e^2 (double)
                const rsrc_t tres = make_rsrc_bytes(mse.target, mse.target_bytes), bres = make_rsrc_bytes(bias, bias ? (long long)N * 4 : 0);
                const long long src = mse.tidx[row < M ? row : 0];
                const u32 toff = (u32)((src * mse.ldt + mse.tcol0 + col) * 4) | ((row < M && col < N) ? 0u : INVALID);
                const f32x4 tg[2] = {bload4(tres, toff, 0u), bload4(tres, toff, 16u)};
                const f32x4 bv[2] = {bload4(bres, (u32)col * 4u, 0u), bload4(bres, (u32)col * 4u, 16u)};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (row < M && col + e < N) ? (v[e >> 2][e & 3] + bv[e >> 2][e & 3]) - tg[e >> 2][e & 3] : 0.f;
                    v[e >> 2][e & 3] = d * mse.scale;
                    sq += (double)d * (double)d;
                }
                if (Y) store8(Y, ldy, row, col, M, N, (wide & 1) != 0, v);
            } else if constexpr (EPI == EPI_DGRAD) {
                if (!dg.rmask && act != DTC_ACT_NONE) {
                    f32x4 y[2];
                    load8(dg.Xs, dg.ldxs, row, col, M, N, (wide & 2) != 0, y);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = act_bwd(v[e >> 2][e & 3], y[e >> 2][e & 3], act);
                }
                if (dg.add) {
                    f32x4 o[2];
                    load8(dg.add, dg.ld_add, row, col, M, N, (wide & 4) != 0, o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = o[e >> 2][e & 3] + v[e >> 2][e & 3];
                }
                if (dg.has_dx && col < N) {
                    const auto& dX = dg.dX;
                    const int sj = find_seg_t(dX, col);
                    const SegDev sdj = seg_at(dX, sj);
                    if (col + 8 <= sdj.start + sdj.width || (sj == dX.nseg - 1)) {      // inside one destination block
                        if (sdj.ptr) {
                            const int lc = col - sdj.start;
                            const bool w16 = ((dg.wide_segs >> sj) & 1) && ((lc + sdj.col0) & 3) == 0;
                            float* base = sdj.ptr + sdj.col0;
                            if (sdj.accumulate) {
                                f32x4 o[2];
                                load8(base, sdj.ld, row, lc, M, sdj.width, w16, o);
                                f32x4 s2[2];
#pragma unroll
                                for (int e = 0; e < 8; ++e) s2[e >> 2][e & 3] = o[e >> 2][e & 3] + v[e >> 2][e & 3];
                                store8(base, sdj.ld, row, lc, M, sdj.width, w16, s2);
                            } else {
                                store8(base, sdj.ld, row, lc, M, sdj.width, w16, v);
                            }
                        }
                    } else if (row < M) {                   // the group straddles a block border: element by element
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int c = col + e;
                            if (c >= N) break;
                            const SegDev sc = seg_at(dX, find_seg_t(dX, c));
                            if (sc.ptr == nullptr) continue;
                            float* qd = sc.ptr + sc.col0 + (c - sc.start) + (long long)row * sc.ld;
                            *qd = sc.accumulate ? (*qd + v[e >> 2][e & 3]) : v[e >> 2][e & 3];
                        }
                    }
                }
            } else {
                if (Y) store8(Y, ldy, row, col, M, N, (wide & 1) != 0, v);
            }
            if (!full) {                                    // (uniform) images hold zeros behind the matrix
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = (row < M && col + e < N) ? v[e >> 2][e & 3] : 0.f;
            }
            // largest |value| of the row so far as a bit pattern (2 VALU per element); a non-finite element makes the pattern >= inf's and
            // sends the wave through the filtered pass below
#pragma unroll
            for (int e = 0; e < (DTC_H2I_PROBE < 3 ? 8 : 0); ++e) {
                const u32 bb = EPI == EPI_MSE ? finite_bits(v[e >> 2][e & 3]) : abs_bits(v[e >> 2][e & 3]);      // (the loss epilogue has no registers to spare for the second pass)
                mrow[i][q] = bb > mrow[i][q] ? bb : mrow[i][q];
            }
        }
    }
    if constexpr (EPI != EPI_MSE) {
        u32 any = 0u;
#pragma unroll
        for (int i = 0; i < TM; ++i) any |= mrow[i][0] | mrow[i][1];
        if (__builtin_amdgcn_ballot_w64(any >= 0x7f800000u) != 0ull) {      // (rare) some row of this wave holds inf / NaN: finite elements only
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32 m = 0u;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const u32 bb = finite_bits(T[i][j][q][e >> 2][e & 3]);
                            m = bb > m ? bb : m;
                        }
                    mrow[i][q] = m;
                }
        }
    }

    if constexpr (EPI == EPI_MSE) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    }
    if constexpr (DTC_H2I_PROBE >= 2) {                 // timing probe: everything above stays live, nothing below runs
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) sink += T[i][j][q][e >> 2][e & 3];
        u32 ms = 0u;
#pragma unroll
        for (int i = 0; i < TM; ++i) ms |= mrow[i][0] | mrow[i][1];
        if (sink == 1.2345e-30f || ms == 0x12345u) yo.exps[0] = 1;
        return;
    }
    // ---- the image of the result: row maxima over the tile's 128 columns (4 lanes, then the neighbouring wave through LDS), exponents,
    // split, 16-byte pieces.  (The stage buffers of W are free: every wave passed the loop's last barrier.)
    u32* rm = reinterpret_cast<u32*>(&Ws0[0][0]);           // [2][128]
    double* red = reinterpret_cast<double*>(&Ws1[0][0]);
    if (yo.img) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32 m = mrow[i][q];
                const u32 o1 = (u32)__shfl_xor((int)m, 1, 64);
                m = o1 > m ? o1 : m;
                const u32 o2 = (u32)__shfl_xor((int)m, 2, 64);
                m = o2 > m ? o2 : m;
                mrow[i][q] = m;
                if (c8 == 0) rm[(wave % WN) * 128 + ro[i] + r16 + 16 * q] = m;
            }
    }
    if (EPI == EPI_MSE && lane == 0) red[wave] = sq;
    if (yo.img || EPI == EPI_MSE) __syncthreads();
    if (EPI == EPI_MSE && tid == 0) mse.part[slot] = ((red[0] + red[1]) + red[2]) + red[3];
    if (yo.img == nullptr) {
        if (trace && tid == 0) trace[4 * slot + 2] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    if constexpr (DTC_H2I_PROBE == 1) {
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) sink += T[i][j][q][e >> 2][e & 3];
        u32 ms = 0u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) ms |= rm[ro[i] + r16 + 16 * q] | rm[128 + ro[i] + r16 + 16 * q];
        if (sink == 1.2345e-30f || ms == 0x12345u) yo.exps[0] = 1;
        return;
    }
    u32x4* tile_chunks = yo.img + (long long)ctile * yo.stages * (HI_CHUNK / 16);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rloc = ro[i] + r16 + 16 * q;
            const u32 m = rm[rloc] > rm[128 + rloc] ? rm[rloc] : rm[128 + rloc];
            const int e = hi_exp(m);
            const int rch = crow0 + rloc;                      // the row inside its 128-row chunk
            if (c8 == 0 && (wave % WN) == 0 && tc < yo.kbs) yo.exps[((long long)ctile * yo.kbs + tc) * 128 + rch] = e;
            const int ee = e == HI_EZERO ? 0 : e;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int lc = n0 + co[j] + 8 * c8;
                if ((lc >> 4) >= yo.stages) continue;
                const HiPiece pc = hi_split8(T[i][j][q], ee);
                u32x4* chunk = tile_chunks + (long long)(lc >> 4) * (HI_CHUNK / 16);
#if DTC_H2I_NT_STORES
                __builtin_nontemporal_store(pc.p[0], &chunk[rslot(rch, (lc >> 3) & 1)]);
                __builtin_nontemporal_store(pc.p[1], &chunk[256 + rslot(rch, (lc >> 3) & 1)]);
#else
                chunk[rslot(rch, (lc >> 3) & 1)] = pc.p[0];
                chunk[256 + rslot(rch, (lc >> 3) & 1)] = pc.p[1];
#endif
            }
        }
    if (trace && tid == 0) trace[4 * slot + 2] = __builtin_amdgcn_s_memrealtime();
#undef XS
#undef WS
}

#endif  // __HIP_DEVICE_COMPILE__

// one launch = one layer: XCD-aware block -> tile map (all column tiles of a row tile share an L2)
template <int EPI, int TM>
__global__ __launch_bounds__(256, 3) void linear_h2i_kernel(const TileArgs L_, const MseEpiH mse, unsigned long long* __restrict__ trace) {
    // the descriptors are read IN PLACE from the kernel-argument segment (first argument = offset 0): binding a reference to the by-value
    // argument would copy it to scratch as soon as one of its arrays is indexed at run time
#ifdef __HIP_DEVICE_COMPILE__
    CTileArgs& L = *(CTileArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)L_;
    int tr, tc;
    if (!map_tile(blockIdx.x, (L.M + 64 * TM - 1) / (64 * TM), (L.N + 127) / 128, tr, tc)) {
        if (EPI == EPI_MSE && threadIdx.x == 0) mse.part[blockIdx.x] = 0.0;
        return;
    }
    h2i_tile<EPI, TM>(L, mse, tr, tc, (int)blockIdx.x, trace);
#endif
}

// A chain of up to three layers whose results are at most 128 columns wide (ONE column tile): the workgroup that owns a row tile runs
// layer after layer on it -- layer l + 1 reads, as its row operand, the image rows this workgroup wrote as layer l's result (global
// stores ordered by the workgroup barrier between the layers; the rows of other tiles are never touched).  The narrow stacks of the
// CE-net (encoder 265 -> 128 -> 64 -> 35, decoder 531 -> 64 -> 128 -> 53: actor_critic_decoder.py:98-142) as one launch per direction
// instead of three (two) latency-bound ones; every intermediate still reaches HBM as an image (the weight gradients read it).
constexpr int CHAIN_MAX = 3, CHAIN_MAX_COLS = 512;
struct ChainArgs {
    int count;
    TileArgs layer[CHAIN_MAX];
};
// (128-row tiles -- M > 32768 rows only -- at two workgroups per CU: the column-tile loops around three inlined tile bodies do not fit 168
// registers; the 64-row form every size of this project takes keeps three)
template <int EPI, int TM>
__global__ __launch_bounds__(256, TM == 2 ? 2 : 3) void chain_h2i_kernel(const ChainArgs C_, unsigned long long* __restrict__ trace) {
#ifdef __HIP_DEVICE_COMPILE__
    typedef const AS4 ChainArgs CChainArgs;
    CChainArgs& C = *(CChainArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)C_;
    (void)trace;
    int tr, tc;
    if (!map_tile(blockIdx.x, (C.layer[0].M + 64 * TM - 1) / (64 * TM), 1, tr, tc)) return;
    const MseEpiH none{};
    // (round 6) a layer may be up to CHAIN_MAX_COLS columns wide: the workgroup runs its column tiles one after the other -- the actor's and
    // the critic's tails 512 -> 256 -> 128 (actor_critic_decoder.py:323-349) forward, 128 -> 256 -> 512 backward
    for (int tc0 = 0; tc0 < (C.layer[0].N + 127) / 128; ++tc0) {
            if (tc0 > 0) __syncthreads();        // (the previous tile's epilogue still uses the stage buffers as its patches)
            h2i_tile<EPI, TM>(C.layer[0], none, tr, tc0, (int)blockIdx.x, nullptr);
        }
    if (C.count > 1) {
        __syncthreads();                    // (workgroup-scope release / acquire: this tile's image rows and exponents are visible to all its waves)
        for (int tc1 = 0; tc1 < (C.layer[1].N + 127) / 128; ++tc1) {
            if (tc1 > 0) __syncthreads();        // (the previous tile's epilogue still uses the stage buffers as its patches)
            h2i_tile<EPI, TM>(C.layer[1], none, tr, tc1, (int)blockIdx.x, nullptr);
        }
    }
    if (C.count > 2) {
        __syncthreads();
        for (int tc2 = 0; tc2 < (C.layer[2].N + 127) / 128; ++tc2) {
            if (tc2 > 0) __syncthreads();        // (the previous tile's epilogue still uses the stage buffers as its patches)
            h2i_tile<EPI, TM>(C.layer[2], none, tr, tc2, (int)blockIdx.x, nullptr);
        }
    }
#endif
}

unsigned long long* g_trace = nullptr;      // debug: dtc_h2i_trace

// ---- host side ---------------------------------------------------------------------------------------------------------------------
int check_img(const void* img, const char* what) {
    DTC_REQUIRE(img != nullptr && dtc::aligned16(img), "%s: null / unaligned image", what);
    return DTC_OK;
}

int to_operand(const DtcH2iOperand* X, int M, HOperand& A, int& K) {
    DTC_REQUIRE(X != nullptr && X->nseg >= 1 && X->nseg <= 4, "row operand: 1..4 images");
    A = HOperand{};
    A.nseg = X->nseg;
    K = 0;
    for (int i = 0; i < X->nseg; ++i) {
        DTC_REQUIRE(X->img[i] != nullptr && dtc::aligned16(X->img[i]) && X->width[i] > 0, "row operand: image %d null / unaligned / empty", i);
        DTC_REQUIRE(hi_bytes(M, X->width[i]) < (1ll << 31), "row operand: image %d beyond 2 GiB", i);
        HSeg& s = A.s[i];
        s.img = (const u32x4*)X->img[i];
        s.bytes = (u32)hi_data_bytes(M, X->width[i]);
        s.exps = reinterpret_cast<const int*>(reinterpret_cast<const char*>(X->img[i]) + s.bytes);
        s.stages = (int)hi_stages(X->width[i]);
        s.kbs = (int)hi_kblocks(X->width[i]);
        A.total += s.stages;
        A.tblocks += s.kbs;
        K += X->width[i];
    }
    DTC_REQUIRE(A.tblocks <= MAX_TB, "row operand: %d exponent blocks along the reduction, at most %d (2048 columns)", A.tblocks, MAX_TB);
    return DTC_OK;
}

HOut to_out(void* img, int M, int N) {
    HOut o{};
    if (img) {
        o.img = (u32x4*)img;
        o.exps = reinterpret_cast<int*>(reinterpret_cast<char*>(img) + hi_data_bytes(M, N));
        o.stages = (int)hi_stages(N);
        o.kbs = (int)hi_kblocks(N);
    }
    return o;
}

// the weight image the caller built with dtc_h2i_wimage_group for exactly this product: rows = nr, reduction walk = the operand's images
struct WimgView {
    const u32x4* img;
    const int* exps;
    u32 bytes;
};
WimgView wimg_view(const void* wimg, int nr, const HOperand& A) {
    WimgView v;
    v.img = (const u32x4*)wimg;
    v.bytes = (u32)(dtc::ceil_div(nr, 128) * A.total * HI_CHUNK);
    v.exps = reinterpret_cast<const int*>(reinterpret_cast<const char*>(wimg) + v.bytes);
    return v;
}

}  // namespace

// debug: the image-operand GEMM launches that follow write per-workgroup records {start, K loop done, end (100 MHz ticks), HW_ID} to
// `buf` (4 x grid uint64, device; NULL: off)
extern "C" void dtc_h2i_trace(void* buf) { g_trace = (unsigned long long*)buf; }

extern "C" int64_t dtc_h2i_bytes(int M, int K) {
    if (M <= 0 || K <= 0) return 0;
    return hi_bytes(M, K);
}

// image(M, X->cols) of the fp32 operand X (segments side by side, row-gathered where a segment asks for it)
extern "C" int dtc_h2i_pack(const DtcSegMat* X, int M, void* img, void* stream) {
    DTC_REQUIRE(X != nullptr && M > 0, "null operand / bad M");
    int rc = check_img(img, "dtc_h2i_pack");
    if (rc != DTC_OK) return rc;
    PackArgs P;
    rc = to_dev(X, P.X, X->cols, false, M);
    if (rc != DTC_OK) return rc;
    P.M = M;
    P.K = X->cols;
    DTC_REQUIRE(hi_bytes(M, P.K) < (1ll << 31), "image beyond 2 GiB");
    P.stages = (int)hi_stages(P.K);
    P.kbs = (int)hi_kblocks(P.K);
    P.img = (u32x4*)img;
    P.exps = reinterpret_cast<int*>(reinterpret_cast<char*>(img) + hi_data_bytes(M, P.K));
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof("h2i_pack", 0.0, s, 8.0 * M * (double)P.K);
    hipLaunchKernelGGL(h2i_pack_kernel, dim3((unsigned)(hi_rtiles(M) * P.kbs)), dim3(256), 0, s, P);
    return dtc::check_launch("h2i_pack");
}

extern "C" int dtc_h2i_unpack(const void* img, int M, int K, float* out, int64_t ld, void* stream) {
    DTC_REQUIRE(out != nullptr && M > 0 && K > 0 && ld >= K, "null pointer / bad shape");
    int rc = check_img(img, "dtc_h2i_unpack");
    if (rc != DTC_OK) return rc;
    const int stages = (int)hi_stages(K);
    hipLaunchKernelGGL(h2i_unpack_kernel, dim3((unsigned)(hi_rtiles(M) * stages)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)img,
                       reinterpret_cast<const int*>(reinterpret_cast<const char*>(img) + hi_data_bytes(M, K)), M, K, stages, (int)hi_kblocks(K),
                       out, (long long)ld);
    return dtc::check_launch("h2i_unpack");
}

namespace {
long long wjob_tiles(const DtcH2iWJob& h) {
    if (h.nrows < 1 || h.nrows > 2) return -1;
    long long ct = 0;
    for (int i = 0; i < h.nrows; ++i) {
        if (h.nr[i] <= 0 || h.r0[i] < 0 || (i + 1 < h.nrows && h.nr[i] % 128 != 0)) return -1;
        ct += dtc::ceil_div(h.nr[i], 128);
    }
    return ct;
}
}  // namespace

extern "C" int64_t dtc_h2i_wimage_bytes(const DtcH2iWJob* job) {
    if (job == nullptr || job->nseg < 1 || job->nseg > 4) return -1;
    const long long ct = wjob_tiles(*job);
    if (ct < 0) return -1;
    long long st = 0, tb = 0;
    for (int i = 0; i < job->nseg; ++i) {
        if (job->cw[i] <= 0) return -1;
        st += hi_stages(job->cw[i]);
        tb += hi_kblocks(job->cw[i]);
    }
    return ct * st * HI_CHUNK + ((ct * tb * 4 + 15) & ~15ll);
}

// the weight images of `count` products in one launch per WP_MAX_JOBS jobs (a trainer builds the images of all layers of an optimisation
// step at its start, after the optimiser wrote the weights)
extern "C" int dtc_h2i_wimage_group(const DtcH2iWJob* jobs, int count, void* stream) {
    DTC_REQUIRE(jobs != nullptr && count > 0, "no jobs");
    hipStream_t s = (hipStream_t)stream;
    double elems = 0.0;
    for (int i = 0; i < count; ++i) {
        int red = 0;
        for (int k = 0; k < jobs[i].nseg && k < 4; ++k) red += jobs[i].cw[k];
        elems += (double)(jobs[i].nr[0] + (jobs[i].nrows > 1 ? jobs[i].nr[1] : 0)) * red;
    }
    dtc::ProfScope prof("wimage", 0.0, s, 8.0 * elems);
    WpackGroup G;
    G.count = 0;
    auto flush = [&]() {
        if (G.count == 0) return;
        hipLaunchKernelGGL(h2i_wpack_kernel, dim3((unsigned)G.job[G.count - 1].block_end), dim3(256 * WP_SQ), 0, s, G);
        G.count = 0;
    };
    for (int i = 0; i < count; ++i) {
        const DtcH2iWJob& h = jobs[i];
        DTC_REQUIRE(h.W && h.img && dtc::aligned16(h.img) && h.ld > 0 && h.nseg >= 1 && h.nseg <= 4, "job %d: null pointer / bad shape", i);
        const long long ct = wjob_tiles(h);
        DTC_REQUIRE(ct > 0, "job %d: bad row ranges (1..2, all but the last a multiple of 128 long)", i);
        WpackJob& J = G.job[G.count];
        J.W = h.W;
        J.ld = h.ld;
        J.img = (u32x4*)h.img;
        J.trans = h.trans;
        J.nrows = h.nrows;
        for (int k = 0; k < 2; ++k) {
            J.r0[k] = k < h.nrows ? h.r0[k] : 0;
            J.nr[k] = k < h.nrows ? h.nr[k] : 0;
        }
        J.nseg = h.nseg;
        J.tstages = J.tblocks = 0;
        for (int k = 0; k < 4; ++k) {
            J.c0[k] = k < h.nseg ? h.c0[k] : 0;
            J.cw[k] = k < h.nseg ? h.cw[k] : 0;
            if (k < h.nseg) {
                DTC_REQUIRE(h.cw[k] > 0 && h.c0[k] >= 0, "job %d: bad reduction range %d", i, k);
                J.tstages += (int)hi_stages(h.cw[k]);
                J.tblocks += (int)hi_kblocks(h.cw[k]);
            }
        }
        DTC_REQUIRE(J.tblocks <= MAX_TB, "job %d: %d exponent blocks along the reduction, at most %d", i, J.tblocks, MAX_TB);
        DTC_REQUIRE(ct * J.tstages * HI_CHUNK < (1ll << 31), "job %d: image beyond 2 GiB", i);
        J.exps = reinterpret_cast<int*>(reinterpret_cast<char*>(h.img) + ct * J.tstages * HI_CHUNK);
        J.block_end = (int)(ct * J.tblocks) + (G.count > 0 ? G.job[G.count - 1].block_end : 0);
        if (++G.count == WP_MAX_JOBS) flush();
    }
    flush();
    return dtc::check_launch("h2i_wimage_group");
}

namespace {
// 64-row tiles when 128-row tiles would leave CUs without a workgroup (DTC_H2I_ROWS64_MAX: largest 128-row tile count that still takes
// them, default 256 = one per CU; 0: never)
int g_rows64_max = -1;       // dtc_h2i_rows64_max; -1: the environment's value / the default
bool rows64(int M, int N) {
    constexpr int env_tiles = 256;
    const int max_tiles = g_rows64_max >= 0 ? g_rows64_max : env_tiles;
    return dtc::ceil_div(M, BM) * dtc::ceil_div(N, 128) <= max_tiles && M > 64;
}
// arguments of one forward layer / one data-gradient layer: validation + descriptors, one place for both
struct LayerInfo {
    TileArgs t;
    int K;              // reduction length (profile names)
    double flop, bytes;
};
int fwd_args(const DtcH2iOperand* X, const void* wimg, const float* b, float* Y, int64_t ldy, void* Yimg, uint16_t* relu_mask, int M, int N,
             int act, LayerInfo& out) {
    DTC_REQUIRE(M > 0 && N > 0 && (Y == nullptr || ldy >= N), "bad shape M=%d N=%d ldy=%lld", M, N, (long long)ldy);
    DTC_REQUIRE((Y || Yimg) && dtc::aligned16(Yimg), "no result / unaligned image");
    DTC_REQUIRE(act >= 0 && act <= DTC_ACT_SIGMOID, "bad activation %d", act);
    int rc = check_img(wimg, "forward layer: weight image");
    if (rc != DTC_OK) return rc;
    TileArgs& t = out.t;
    t = TileArgs{};
    rc = to_operand(X, M, t.A, out.K);
    if (rc != DTC_OK) return rc;
    DTC_REQUIRE((long long)M * (ldy > N ? ldy : N) <= MAX_ELEMS * 4 && hi_bytes(M, N) < (1ll << 31), "matrix too large");
    if (relu_mask) DTC_REQUIRE(act == DTC_ACT_RELU && M % BM == 0 && (N % 128 == 0 || N == 64), "sign record: M=%d must be a multiple of 128, N=%d a multiple of 128 or 64", M, N);
    const WimgView wv = wimg_view(wimg, N, t.A);
    t.wimg = wv.img;
    t.wexps = wv.exps;
    t.wimg_bytes = wv.bytes;
    t.M = M;
    t.N = N;
    t.act = act;
    t.wide = (Y && ldy % 4 == 0 && dtc::aligned16(Y)) ? 1 : 0;
    t.ldwm = N;
    t.bias = b;
    t.Y = Y;
    t.ldy = ldy;
    t.yo = to_out(Yimg, M, N);
    t.wmask = (unsigned short*)relu_mask;
    out.flop = 2.0 * M * (double)N * out.K;
    out.bytes = 4.0 * M * (double)out.K + 4.0 * N * (double)out.K + (Y ? 4.0 : 0.0) * M * N + (Yimg ? 4.0 : 0.0) * M * N;
    return DTC_OK;
}
int dgrad_args(const void* dZimg, int N, const void* wimgT, int Kwin, const DtcSegMat* dX, void* dXimg, int img_cols, const float* add,
               int64_t ld_add, const float* Xsaved, int64_t ldxs, const uint16_t* relu_mask, int M, int act, LayerInfo& out) {
    DTC_REQUIRE(M > 0 && N > 0 && Kwin > 0 && img_cols >= 0 && img_cols <= Kwin, "bad shape");
    if (img_cols == 0) img_cols = Kwin;
    DTC_REQUIRE(img_cols == Kwin || img_cols % 128 == 0, "image of the window's first %d columns: must be whole 128-column blocks", img_cols);
    DTC_REQUIRE((dX || dXimg) && dtc::aligned16(dXimg), "no result / unaligned image");
    DTC_REQUIRE(act >= 0 && act <= DTC_ACT_SIGMOID, "bad activation %d", act);
    DTC_REQUIRE(relu_mask || act == DTC_ACT_NONE || (Xsaved != nullptr && ldxs >= Kwin), "activation derivative needs Xsaved");
    DTC_REQUIRE(add == nullptr || ld_add >= Kwin, "bad ld_add");
    int rc = check_img(dZimg, "data gradient: dZ");
    if (rc != DTC_OK) return rc;
    rc = check_img(wimgT, "data gradient: weight image");
    if (rc != DTC_OK) return rc;
    if (relu_mask) DTC_REQUIRE(M % BM == 0 && (Kwin % 128 == 0 || Kwin == 64), "sign record: M=%d must be a multiple of 128, the window %d a multiple of 128 or 64", M, Kwin);
    DTC_REQUIRE(hi_bytes(M, Kwin) < (1ll << 31), "matrix too large");
    DtcH2iOperand zo{};
    zo.nseg = 1;
    zo.img[0] = dZimg;
    zo.width[0] = N;
    TileArgs& t = out.t;
    t = TileArgs{};
    int nn;
    rc = to_operand(&zo, M, t.A, nn);
    if (rc != DTC_OK) return rc;
    DgradEpiH& dg = t.dg;
    if (dX) {
        rc = to_dev(dX, dg.dX, Kwin, true, 0);
        if (rc != DTC_OK) return rc;
        dg.has_dx = 1;
        for (int i = 0; i < dg.dX.nseg; ++i) {
            const SegDev& sd = dg.dX.s[i];
            if (sd.ptr == nullptr || (dtc::aligned16(sd.ptr) && (sd.ld & 3) == 0)) dg.wide_segs |= 1 << i;
        }
    }
    dg.add = add;
    dg.ld_add = ld_add;
    dg.Xs = relu_mask ? nullptr : Xsaved;
    dg.ldxs = ldxs;
    dg.rmask = (const unsigned short*)relu_mask;
    dg.ldm = Kwin;
    const WimgView wv = wimg_view(wimgT, Kwin, t.A);
    t.wimg = wv.img;
    t.wexps = wv.exps;
    t.wimg_bytes = wv.bytes;
    t.M = M;
    t.N = Kwin;
    t.act = relu_mask ? (int)DTC_ACT_RELU : act;
    t.wide = ((dg.Xs && ldxs % 4 == 0 && dtc::aligned16(dg.Xs)) ? 2 : 0) | ((add && ld_add % 4 == 0 && dtc::aligned16(add)) ? 4 : 0);
    t.yo = to_out(dXimg, M, img_cols);
    out.K = N;
    out.flop = 2.0 * M * (double)N * Kwin;
    out.bytes = 4.0 * M * (double)N + 4.0 * N * (double)Kwin + (dXimg ? 4.0 : 0.0) * M * img_cols + (add ? 4.0 : 0.0) * M * Kwin;
    if (dX)
        for (int i = 0; i < dg.dX.nseg; ++i)
            if (dg.dX.s[i].ptr) out.bytes += 4.0 * M * dg.dX.s[i].width * (dg.dX.s[i].accumulate ? 2.0 : 1.0);
    if (relu_mask) out.bytes += 0.125 * M * (double)Kwin;
    else if (act != DTC_ACT_NONE) out.bytes += 4.0 * M * (double)Kwin;
    return DTC_OK;
}
}  // namespace

// forward / data-gradient launches with at most `max_tiles` 128 x 128 tiles run on 64-row tiles (default 256 = one tile per CU; 0: never;
// -1: back to DTC_H2I_ROWS64_MAX / the default).  Results do not depend on it bit for bit (same per-row arithmetic, same exponents).
extern "C" void dtc_h2i_rows64_max(int max_tiles) { g_rows64_max = max_tiles; }

// Y = act(X W^T + b): X = images side by side, wimg = the image of W built for that walk (trans = 0, rows (0, N), the operand's widths
// as reduction ranges).  Results: fp32 Y [M, >= N] (may be NULL) and / or the image of Y (Yimg = image(M, N), may be NULL) -- at least one.
extern "C" int dtc_linear_fwd_h2i(const DtcH2iOperand* X, const void* wimg, const float* b, float* Y, int64_t ldy, void* Yimg,
                                  uint16_t* relu_mask, int M, int N, int act, void* stream) {
    LayerInfo L;
    int rc = fwd_args(X, wimg, b, Y, ldy, Yimg, relu_mask, M, N, act, L);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, L.K), L.flop, s, L.bytes);
    if (rows64(M, N))
        hipLaunchKernelGGL((linear_h2i_kernel<EPI_FWD, 1>), dim3(grid_for((int)dtc::ceil_div(M, 64), (int)dtc::ceil_div(N, 128))), dim3(256), 0, s, L.t,
                           MseEpiH{}, g_trace);
    else
        hipLaunchKernelGGL((linear_h2i_kernel<EPI_FWD, 2>), dim3(grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 128))), dim3(256), 0, s, L.t,
                           MseEpiH{}, g_trace);
    return dtc::check_launch("linear_fwd_h2i");
}

extern "C" int64_t dtc_linear_fwd_mse_h2i_parts(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 128));
}

// the layer fused with its MSE against target[tidx[row], tcol0 + col] (dtc_linear_fwd_mse): dL/dY = (Y - target) * scale as fp32 (dY, may
// be NULL) and / or as an image (dYimg); sq_part: dtc_linear_fwd_mse_h2i_parts(M, N) doubles (partials of sum (Y - target)^2)
extern "C" int dtc_linear_fwd_mse_h2i(const DtcH2iOperand* X, const void* wimg, const float* b, const float* target, int64_t ldt,
                                      int64_t target_rows, int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy, void* dYimg,
                                      double* sq_part, int M, int N, void* stream) {
    DTC_REQUIRE(target && tidx && sq_part, "null pointer");
    DTC_REQUIRE(tcol0 >= 0 && tcol0 + N <= ldt && target_rows > 0, "target columns [%d, %d) outside its %lld-wide rows", tcol0, tcol0 + N, (long long)ldt);
    DTC_REQUIRE(target_rows * ldt <= MAX_ELEMS, "matrix too large");
    LayerInfo L;
    int rc = fwd_args(X, wimg, b, dY, lddy, dYimg, nullptr, M, N, (int)DTC_ACT_NONE, L);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const MseEpiH mse{target, (const long long*)tidx, (long long)ldt, target_rows * ldt * 4, tcol0, scale, sq_part};
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, L.K), L.flop, s, L.bytes + 4.0 * M * N);
    hipLaunchKernelGGL((linear_h2i_kernel<EPI_MSE, 2>), dim3(grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 128))), dim3(256), 0, s, L.t,
                       mse, g_trace);
    return dtc::check_launch("linear_fwd_mse_h2i");
}

// dX[:, window] = (dZ W[:, window]) * act'(.): dZ = image(M, N); wimgT = the image of W^T built for the window (trans = 1, rows = the
// window's column ranges of W, one reduction range (0, N)).  Results over the window: fp32 destination blocks dX (may be NULL;
// accumulate flags honoured) and / or the image dXimg of the window's first img_cols columns.  add (may be NULL): fp32 [M, ld_add] added
// to the product first.  Activation derivative: relu_mask (sign record) or Xsaved with act; both need the window to start at the saved
// tensor's column 0.
extern "C" int dtc_linear_dgrad_h2i(const void* dZimg, int N, const void* wimgT, int Kwin, const DtcSegMat* dX, void* dXimg, int img_cols,
                                    const float* add, int64_t ld_add, const float* Xsaved, int64_t ldxs, const uint16_t* relu_mask, int M, int act,
                                    void* stream) {
    LayerInfo L;
    int rc = dgrad_args(dZimg, N, wimgT, Kwin, dX, dXimg, img_cols, add, ld_add, Xsaved, ldxs, relu_mask, M, act, L);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof(dtc::prof_shape_name("linear_dgrad", M, N, Kwin), L.flop, s, L.bytes);
    if (rows64(M, Kwin))
        hipLaunchKernelGGL((linear_h2i_kernel<EPI_DGRAD, 1>), dim3(grid_for((int)dtc::ceil_div(M, 64), (int)dtc::ceil_div(Kwin, 128))), dim3(256), 0, s,
                           L.t, MseEpiH{}, g_trace);
    else
        hipLaunchKernelGGL((linear_h2i_kernel<EPI_DGRAD, 2>), dim3(grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(Kwin, 128))), dim3(256), 0, s,
                           L.t, MseEpiH{}, g_trace);
    return dtc::check_launch("linear_dgrad_h2i");
}

// ---- chains of narrow layers (one launch; see chain_h2i_kernel) ---------------------------------------------------------------------
namespace {
int chain_launch(bool dgrad, const LayerInfo* L, int count, int M, hipStream_t s) {
    ChainArgs C;
    C.count = count;
    double flop = 0.0, bytes = 0.0;
    for (int i = 0; i < count; ++i) {
        C.layer[i] = L[i].t;
        flop += L[i].flop;
        bytes += L[i].bytes;
    }
    const bool r64 = rows64(M, 128);
    const int row_tiles = (int)dtc::ceil_div(M, r64 ? 64 : BM);
    dtc::ProfScope prof(dtc::prof_shape_name(dgrad ? "linear_dgrad_chain" : "linear_fwd_chain", M, count, L[0].K), flop, s, bytes);
    if (dgrad) {
        if (r64) hipLaunchKernelGGL((chain_h2i_kernel<EPI_DGRAD, 1>), dim3(grid_for(row_tiles, 1)), dim3(256), 0, s, C, g_trace);
        else hipLaunchKernelGGL((chain_h2i_kernel<EPI_DGRAD, 2>), dim3(grid_for(row_tiles, 1)), dim3(256), 0, s, C, g_trace);
    } else {
        if (r64) hipLaunchKernelGGL((chain_h2i_kernel<EPI_FWD, 1>), dim3(grid_for(row_tiles, 1)), dim3(256), 0, s, C, g_trace);
        else hipLaunchKernelGGL((chain_h2i_kernel<EPI_FWD, 2>), dim3(grid_for(row_tiles, 1)), dim3(256), 0, s, C, g_trace);
    }
    return dtc::check_launch(dgrad ? "linear_dgrad_chain_h2i" : "linear_fwd_chain_h2i");
}
}  // namespace

// `count` (1..3) forward layers of at most 128 output columns each, layer i + 1 reading layer i's image result (layers[i + 1].X must be
// exactly one image: layers[i].Yimg): the same results, bit for bit, as `count` dtc_linear_fwd_h2i calls issued in order.
extern "C" int dtc_linear_fwd_chain_h2i(const DtcH2iFwdLayer* layers, int count, int M, void* stream) {
    DTC_REQUIRE(layers != nullptr && count >= 1 && count <= CHAIN_MAX, "chain of %d layers (1..%d)", count, CHAIN_MAX);
    LayerInfo L[CHAIN_MAX];
    for (int i = 0; i < count; ++i) {
        const DtcH2iFwdLayer& h = layers[i];
        DTC_REQUIRE(h.N >= 1 && h.N <= CHAIN_MAX_COLS, "chain layer %d: %d output columns (1..%d)", i, h.N, CHAIN_MAX_COLS);
        if (i > 0) DTC_REQUIRE(h.X.nseg == 1 && h.X.img[0] == layers[i - 1].Yimg && layers[i - 1].Yimg != nullptr && h.X.width[0] == layers[i - 1].N,
                               "chain layer %d must read the image result of layer %d", i, i - 1);
        int rc = fwd_args(&h.X, h.wimg, h.b, h.Y, h.ldy, h.Yimg, h.relu_mask, M, h.N, h.act, L[i]);
        if (rc != DTC_OK) return rc;
    }
    return chain_launch(false, L, count, M, (hipStream_t)stream);
}

// `count` (1..3) data-gradient layers whose computed windows are at most 128 columns wide, layer i + 1 reading layer i's image result
// (layers[i + 1].dZimg == layers[i].dXimg, layers[i + 1].N == layers[i].img_cols or Kwin): bit for bit `count` dtc_linear_dgrad_h2i calls.
extern "C" int dtc_linear_dgrad_chain_h2i(const DtcH2iDgradLayer* layers, int count, int M, void* stream) {
    DTC_REQUIRE(layers != nullptr && count >= 1 && count <= CHAIN_MAX, "chain of %d layers (1..%d)", count, CHAIN_MAX);
    LayerInfo L[CHAIN_MAX];
    for (int i = 0; i < count; ++i) {
        const DtcH2iDgradLayer& h = layers[i];
        DTC_REQUIRE(h.Kwin >= 1 && h.Kwin <= CHAIN_MAX_COLS, "chain layer %d: window of %d columns (1..%d)", i, h.Kwin, CHAIN_MAX_COLS);
        if (i > 0) {
            const DtcH2iDgradLayer& p = layers[i - 1];
            DTC_REQUIRE(p.dXimg != nullptr && h.dZimg == p.dXimg && h.N == (p.img_cols ? p.img_cols : p.Kwin),
                        "chain layer %d must read the image result of layer %d", i, i - 1);
        }
        int rc = dgrad_args(h.dZimg, h.N, h.wimgT, h.Kwin, h.dX, h.dXimg, h.img_cols, h.add, h.ld_add, h.Xsaved, h.ldxs, h.relu_mask, M, h.act, L[i]);
        if (rc != DTC_OK) return rc;
    }
    return chain_launch(true, L, count, M, (hipStream_t)stream);
}
