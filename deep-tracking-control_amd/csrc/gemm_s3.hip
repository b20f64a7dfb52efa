// Split-precision dense layers for gfx950: the fp32 products of the nn.Linear stacks (rsl_rl/rsl_rl/modules/
// actor_critic_decoder.py:98-188, 323-349 under ppo.py:197-218, 265, 289, 252, 333) computed on the bf16 matrix pipe.
//
// Why: gfx950 has no TF32-like mode; v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s), 1/16 of
// v_mfma_f32_32x32x16_bf16.  An fp32 number is the exact sum of three bf16 numbers (8 significant bits each, round to
// nearest even on the successive remainders: a = a1 + a2 + a3 with |a - (a1 + a2 + a3)| <= 2^-24 |a|, the remainders
// a - a1 and (a - a1) - a2 are exact fp32 subtractions).  A product a b is then
//     a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)        + O(2^-24 |a b|) dropped (a2 b3, a3 b2, a3 b3)
// -- SIX bf16 MFMAs whose 16-bit x 16-bit products are exact in the fp32 accumulator; what is dropped is of the order of
// the rounding error of ONE fp32 multiplication.  Six passes at 16x the rate = 2.67x the fp32 MFMA peak (419 TFLOP/s of
// fp32-equivalent work).  Accuracy against fp64 is measured next to the single-pass fp32 kernels in
// tests/test_hip_split.py (same error level; the sum is not bit-identical to the fmaf chain, nor is the reference's
// own CPU GEMM between two thread counts).
//
// Data path (forward product Y = X W^T):
//   X: global (fp32, dwordx4 per 4 k, the segmented / gathered operand descriptors of gemm.hip) -> registers -> split into three
//   packed-bf16 planes (v_cvt_pk_bf16_f32: 5.5 VALU per element, placed between the MFMAs of the stage in flight) -> LDS planes
//   [row][16 k] (32-byte rows, the two 16-byte halves of a row swapped on odd 8-row groups: conflict-free ds_write_b64 and
//   ds_read_b128) -> one ds_read_b128 per plane and 32 x 32 tile = the 8 bf16 of a lane's MFMA fragment;
//   W: a WEIGHT IMAGE -- the same planes, built once per call (or once per optimisation step for all layers:
//   dtc_s3_wimage_group) by wimage_kernel, one 12 KiB chunk per 128-column tile and 16-k stage -- copied into LDS by LDS-DMA.
//   Block tile 128 x 128 x 16, 2 x 2 waves of 64 x 64, 24 MFMAs per wave and stage, double-buffered LDS (48 KiB), 3
//   workgroups per CU.
// The data gradient uses the same kernel with the image of W^T (the builder reads W transposed).
//
// Round 4: the kernels are templated on the REPRESENTATION (s3_core.hpp: Prec<H2>).  H2 = two fp16 terms of the operand scaled by a power
// of two from its tensor's amax, three v_mfma_f32_32x32x16_f16 passes, 8 KiB image chunks, 32 KiB of LDS, the sums scaled back with
// v_ldexp_f32 before the (unchanged) epilogue, which also publishes the result's amax for the next consumer (amax.hpp).  It is the
// default (DTC_GEMM_SPLIT=2); the text above describes DTC_GEMM_SPLIT=1.
#include <type_traits>

#include "s3_core.hpp"

namespace {

// epilogue modes
constexpr int EPI_FWD = 0, EPI_DGRAD = 1, EPI_MSE = 2;

// loss epilogue (dtc_linear_fwd_mse_s3): the layer output feeds an MSE against a row-gathered target; the epilogue writes
// dL/dY instead of Y and one double partial of sum(e^2) per workgroup
struct MseEpiS3 {
    const float* target;
    const long long* tidx;
    long long ldt, target_bytes;
    int tcol0;
    float scale;
    double* part;
};

struct DgradEpi {
    SegMatDev dX;                 // segmented destination (accumulate flags, NULL segments)
    const float* Xs;              // saved post-activation output of the previous layer (act != none), or NULL
    long long ldxs;
    const unsigned short* rmask;  // sign record of a ReLU layer (replaces Xs), or NULL
    int ldm, col_skip, wide_segs;
};

// Weight image (round 3): the W operand of a launch as the kernel's LDS planes, built ONCE per call by wimage_kernel instead of
// once per row tile inside the K loop (192 times for M = 24576): for every 128-column tile and every 16-k stage of the segment
// walk one 12 KiB chunk [plane 3][row 128][16 k bf16, halves swapped as rslot() says], k tails and rows past N zero-filled.
// The K loop copies a chunk into LDS with three LDS-DMA instructions per wave (buffer_load_dwordx4 ... lds: no registers, no
// conversion, no ds_write for the W side).
constexpr int WIMG_PLANE = 128 * 32;            // bytes of one plane of one stage
constexpr int WIMG_CHUNK = 3 * WIMG_PLANE;      // one stage of one 128-column tile
struct WimgSegs {
    int nseg, start[4], width[4];
};
typedef __attribute__((address_space(3))) void lds_void;

// One image = one job; a launch builds up to WIMG_MAX_JOBS of them (all layers of a trainer phase: dtc_s3_wimage_group).
// block = (job, column tile, stage), thread = (row of the tile, k half); trans: the operand is W^T (element (row, k) = W[k * ld + row])
constexpr int WIMG_MAX_JOBS = 24;
struct WimgJobDev {
    const float* W;
    u32x4* img;
    long long ld;
    int rows, row0, trans, total, block_end;      // block_end: running sum of (column tiles x stages) over the jobs
    WimgSegs sg;
    long long welems;                             // fp16 images: elements of the whole weight matrix and the WPART partial maxima of
    u32* wa;                                      // their |w| (right behind the image's last chunk)
};
struct WimgGroup {
    int count;
    WimgJobDev job[WIMG_MAX_JOBS];
};
// fp16 images, first step: the largest |w| of every job's matrix as WPART partial maxima (block = (job, part); no atomics, nothing to
// zero); the image builder and the GEMM kernels take the maximum of them
constexpr int WPART = 32;
// behind an fp16 image's last chunk: the WPART partial maxima, then up to four amax records of operands the call computes itself
// (inside the slack dtc_s3_planes_bytes leaves: its chunks are sized for three planes and four spare stages; checked per call)
constexpr long long H2_TAIL_BYTES = 4 * WPART + 4 * AMAX_RECORD_BYTES;
__device__ __forceinline__ u32 wpart_max(const u32* wa) {          // lane i < WPART loads partial i (every lane of the wave must call it)
    const int lane = threadIdx.x & 63;
    return wave_max_u32(lane < WPART ? __hip_atomic_load(wa + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
}
__global__ __launch_bounds__(256) void wamax_kernel(const WimgGroup G) {
    const WimgJobDev& J = G.job[blockIdx.x / WPART];
    const int e = blockIdx.x % WPART;
    u32 m = 0u;
    if ((reinterpret_cast<unsigned long long>(J.W) & 15ull) == 0 && (J.welems & 3) == 0) {       // (uniform) 16-byte loads
        const f32x4* W4 = reinterpret_cast<const f32x4*>(J.W);
        for (long long i = (long long)e * 256 + threadIdx.x; i < (J.welems >> 2); i += WPART * 256) {
            const f32x4 v = W4[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) m = abs_bits(v[q]) > m ? abs_bits(v[q]) : m;
        }
    } else {
        for (long long i = (long long)e * 256 + threadIdx.x; i < J.welems; i += WPART * 256) {
            const u32 b = abs_bits(J.W[i]);
            m = b > m ? b : m;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u32 o = (u32)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    __shared__ u32 red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
        J.wa[e] = m;
    }
}

template <bool H2>
__global__ __launch_bounds__(256) void wimage_kernel(const WimgGroup G) {
    int j = 0, b = blockIdx.x;
    while (j < G.count - 1 && b >= G.job[j].block_end) ++j;
    if (j > 0) b -= G.job[j - 1].block_end;
    const WimgJobDev& J = G.job[j];
    const float* __restrict__ W = J.W;
    const WimgSegs& sg = J.sg;
    const int tc = b / J.total;
    int kt = b - tc * J.total, seg = 0;
    while (seg + 1 < sg.nseg && kt >= (sg.width[seg] + BK - 1) / BK) {
        kt -= (sg.width[seg] + BK - 1) / BK;
        ++seg;
    }
    const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
    const int row = J.row0 + tc * 128 + r, k0 = kt * BK + 8 * h;
    f32x4 v[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int jj = k0 + e;
        const long long col = sg.start[seg] + jj;
        const bool ok = row < J.rows && jj < sg.width[seg];
        v[e >> 2][e & 3] = ok ? (J.trans ? W[col * J.ld + row] : W[(long long)row * J.ld + col]) : 0.f;
    }
    int ew = 0;
    if constexpr (H2) {
        ew = h2_exp(wpart_max(J.wa));
    }
    using P = Prec<H2>;
    const P s0 = P::split(v[0], ew), s1 = P::split(v[1], ew);
    u32x4* dst = J.img + (long long)b * (P::NP * WIMG_PLANE / 16);
#pragma unroll
    for (int p = 0; p < P::NP; ++p) dst[p * 256 + rslot(r, h)] = u32x4{s0.p[p].x, s0.p[p].y, s1.p[p].x, s1.p[p].y};
}

// amax slots of a two-term fp16 launch (s3_core.hpp): operand segments in, weight image partials in, results out
struct H2Arg {
    const u32* xa[4];             // one slot per segment of the row operand (X resp. dZ)
    const u32* wa;                // the weight image's WPART partial maxima (behind its last chunk)
    u32* ya[4];                   // EPI_FWD / EPI_MSE: [0] = slot of the result; EPI_DGRAD: one per destination block (NULL: not published)
};

// H2 (round 4): operands as TWO fp16 terms and three MFMA passes (s3_core.hpp) instead of three bf16 terms and six passes
template <int EPI, bool WIMG, bool H2 = false>
__global__ __launch_bounds__(256, 3) void linear_s3_kernel(const SegMatDev X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ Y,
                                                           long long ldy, int M, int N, int K, int act, int wide,
                                                           unsigned short* __restrict__ wmask, int ldwm, const DgradEpi dg,
                                                           const MseEpiS3 mse, const u32x4* __restrict__ wimg, long long wimg_bytes,
                                                           const H2Arg h2) {
    static_assert(!H2 || WIMG, "the fp16 path reads its weights as an image");
    using P = Prec<H2>;
    constexpr int NP = P::NP, NT = P::NT, WCH = NP * WIMG_PLANE;
    constexpr int BN = 128, WN = 2, TM = 2, TN = 2, NA = 2, NB = WIMG ? 0 : 2;
    __shared__ __attribute__((aligned(16))) u32x2 As[2][NP][BM * 4];
    // two separate objects: the compiler then knows that an LDS-DMA into one stage buffer cannot alias the fragment reads of the
    // other (with one array it waits for the DMA before the first ds_read of every stage); the K loop is unrolled by two so that
    // the buffer index is a compile-time constant
    __shared__ __attribute__((aligned(16))) u32x2 Bs0[NP][BN * 4];
    __shared__ __attribute__((aligned(16))) u32x2 Bs1[NP][BN * 4];
#define BS(b) ((b) ? Bs1 : Bs0)
    int tr, tc;
    const int ncols = EPI == EPI_DGRAD ? N - dg.col_skip : N;          // output columns that are computed
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (ncols + BN - 1) / BN, tr, tc)) {
        if (EPI == EPI_MSE && threadIdx.x == 0) mse.part[blockIdx.x] = 0.0;      // padding block: its partial slot still gets summed
        return;
    }
    const int m0 = tr * BM, n0 = (EPI == EPI_DGRAD ? dg.col_skip : 0) + tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = (wave / WN) * (32 * TM), wn_off = (wave % WN) * (32 * TN);
    const int half = lane >> 5, l31 = lane & 31;

    // loader geometry (as linear_fwd_kernel): thread owns k chunk lch (4 k) of rows lrow and lrow + 64 of X and of W.
    // (Measured alternatives, round 3: the weight pre-split into bf16 planes by a small kernel + 8 k per thread with
    // ds_write_b128 -- fewer conversion instructions, 3.1 instead of 5.8 VALU per MFMA -- ran SLOWER, 84 vs 76 us on
    // 24576 x 512 x 512: the kernel is bound by the barrier-coupled wait / issue chain of a stage, not by its VALU count.)
    const int lrow = tid >> 2, lch = tid & 3;
    const int aslot0 = wslot(lrow, lch);                       // rows lrow and lrow + 64 share the half-swap parity: + 64 * 4 slots
    u32 woff[NB ? NB : 1];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int wn = n0 + lrow + 64 * i;
        woff[i] = wn < N ? (u32)(wn * K + 4 * lch) * 4u : INVALID;
    }
    const rsrc_t wres = WIMG ? make_rsrc_bytes(wimg, wimg_bytes) : make_rsrc_bytes(W, (long long)N * K * 4);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int total = 0;
    for (int i = 0; i < X.nseg; ++i) total += (X.s[i].width + BK - 1) / BK;
    u32 wchunk = (u32)(tc * total) * (u32)WCH;              // byte offset of the next stage's chunk of this column tile
    // fp16 path: the scale exponents of the two operands (uniform: scalar loads)
    int ex = 0, ew = 0;
    bool poison = false;                                    // an inf / NaN amax: the whole result turns NaN (never a silently wrong scale)
    if constexpr (H2) {
        u32 mx = 0u;
        for (int i = 0; i < X.nseg; ++i) {
            const u32 v = amax_read(h2.xa[i]);
            mx = v > mx ? v : mx;
        }
        const u32 mw = wpart_max(h2.wa);
        ex = __builtin_amdgcn_readfirstlane(h2_exp(mx));
        ew = __builtin_amdgcn_readfirstlane(h2_exp(mw));
        poison = __builtin_amdgcn_readfirstlane((mx >= 0x7f800000u || mw >= 0x7f800000u) ? 1 : 0) != 0;
    }

    // ONE loop over the stages of all segments; the (rare) hop into the next segment re-derives the row offsets (the gathered
    // row index is re-read from idx: two loads per thread and segment instead of registers held across the whole K loop)
    SegDev sd = X.s[0];
    rsrc_t ares;
    u32 aoff[NA];
    auto enter_segment = [&]() {
        ares = make_rsrc_bytes(sd.ptr, (long long)sd.rows * sd.ld * 4);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + lrow + 64 * i;
            const bool ok = m < M;
            const u32 r = ok ? (sd.gather ? (u32)X.idx[m] : (u32)m) : 0u;
            aoff[i] = ok ? (r * (u32)sd.ld + (u32)(sd.col0 + 4 * lch)) * 4u : INVALID;
        }
    };
    int seg = 0, kt = 0, nst = (sd.width + BK - 1) / BK;
    // operand loads run ONE stage ahead of the MFMAs.  (Two stages ahead -- two register sets -- measured slower: 91 vs 73 us
    // on 24576 x 512 x 512; the third workgroup per CU that the registers of the second set cost is worth more.)
    // AH2 (fp16 path, round 4): the X loads run TWO stages ahead (two register sets, set = parity of the stage; the weight image's LDS-DMA
    // stays one stage ahead, an explicit s_waitcnt before the barrier completes it while the X loads of the stage after next stay in
    // flight).  With three MFMA passes a stage's matrix phase is ~160 ns per wave, short against a global load issued one stage earlier.
    // Measured at the end of the round (the first A/B, taken while every wave still paid an atomic in its epilogue, had shown nothing):
    // the launches that leave a CU one workgroup or less gain most -- 24576 x 128 x 256: 30.0 -> 27.5 us, 256 x 512: 44.4 -> 39.6, the
    // 693-wide MSE layer 107 -> 98; 512 x 512 and the data gradients unchanged -- and the step 55.4 -> 54.4 ms (three interleaved pairs;
    // forward kernels only, -DDTC_H2_AHEAD2=2: 55.0).  8 more registers (132-141: three workgroups per CU).  -DDTC_H2_AHEAD2=0: one stage.
#ifndef DTC_H2_AHEAD2
#define DTC_H2_AHEAD2 1
#endif
    constexpr bool AH2 = H2 && WIMG && (DTC_H2_AHEAD2 == 1 || (DTC_H2_AHEAD2 == 2 && EPI != EPI_DGRAD));
    f32x4 ras[AH2 ? 2 : 1][NA], rb[NB ? NB : 1];
    int kls[AH2 ? 2 : 1];                           // last valid element (0..3, < 0: none) of the loaded k chunks; >= 3: no tail
    auto load_w = [&](auto nbc) {                   // (W image) next stage's chunk -> LDS[nbuf] (no wait)
        constexpr int nbuf = decltype(nbc)::value;
        if constexpr (WIMG) {                       // issued first: the compiler waits for ALL loads once an LDS-DMA is in flight
#pragma unroll
            for (int p = 0; p < NP; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_void*)&BS(nbuf)[p][wave_u * 128], 16, tid * 16, wchunk + p * WIMG_PLANE, 0, 0);
            wchunk += WCH;
            if constexpr (AH2) __builtin_amdgcn_sched_barrier(0);     // the X loads below must stay BEHIND the LDS-DMA (vmcnt arithmetic of step())
        }
    };
    auto load_x = [&](auto setc) {                  // next stage of the X cursor -> register set (no wait)
        constexpr int set = decltype(setc)::value;
        const u32 ka = (u32)(kt * BK) * 4u, kw = (u32)(sd.start + kt * BK) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ras[set][i] = bload4(ares, aoff[i], ka);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload4(wres, woff[i], kw);
        kls[set] = sd.width - 1 - (kt * BK + 4 * lch);
        if (++kt == nst && seg + 1 < X.nseg) {
            ++seg;
            sd = X.s[seg];
            enter_segment();
            kt = 0;
            nst = (sd.width + BK - 1) / BK;
        }
    };

    auto store_stage = [&](auto bc) {               // registers -> (k-tail mask) -> three bf16 planes -> LDS
        constexpr int buf = decltype(bc)::value;
        f32x4(&ra)[NA] = ras[AH2 ? buf : 0];
        const int klast = kls[AH2 ? buf : 0];
        if (__builtin_amdgcn_readfirstlane(klast + 4 * lch) < BK - 1) {      // (the same value in every lane) last stage of a ragged segment
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = e <= klast ? ra[i][e] : 0.f;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[i][e] = e <= klast ? rb[i][e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const P s = P::split(ra[i], ex);
#pragma unroll
            for (int p = 0; p < NP; ++p) As[buf][p][aslot0 + 256 * i] = s.p[p];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const P s = P::split(rb[i], 0);
#pragma unroll
            for (int p = 0; p < NP; ++p) BS(buf)[p][aslot0 + 256 * i] = s.p[p];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef DTC_S3_NO_ILV
    auto mfma_stage = [&](auto bc) {
        constexpr int buf = decltype(bc)::value;
        // fragments: A planes of both row tiles stay live (24 registers), B planes are read per column tile (12 registers)
        // smallest terms first: (a3 b1, a2 b2, a1 b3), (a2 b1, a1 b2), a1 b1 (Prec::pa / pb)
        u32x4 a[TM][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i][p] = reinterpret_cast<const u32x4*>(&As[buf][p][0])[rslot(wm_off + 32 * i + l31, half)];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            u32x4 b[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                b[p] = reinterpret_cast<const u32x4*>(&BS(buf)[p][0])[rslot(wn_off + 32 * j + l31, half)];
            // the two row tiles alternate: consecutive MFMAs never wait for each other's accumulator
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][j] = P::mfma(a[i][P::pa(t)], b[P::pb(t)], acc[i][j]);
        }
    };
#endif

#ifndef DTC_S3_NO_ILV
    // Fused stage (round 3: 77.8 -> 74.9 us on 24576 x 512 x 512, 71.3 -> 69.7 ms per step; -DDTC_S3_NO_ILV restores the two phases): the MFMAs of LDS[buf] with the conversion + LDS store of the loaded registers (-> LDS[buf ^ 1])
    // placed INTO the second half of the MFMA sequence (sched_barrier fences between the pieces; sched_group_barrier pipelines were
    // ignored by this compiler): the conversion of a stage is ~110 VALU + 12 LDS stores, 24 MFMAs leave 24 x 28 idle issue cycles.
    auto stage_ilv = [&](auto bc) {
        constexpr int buf = decltype(bc)::value;
        f32x4(&ra)[NA] = ras[AH2 ? (buf ^ 1) : 0];
        const int klast = kls[AH2 ? (buf ^ 1) : 0];
        u32x4 a[TM][NP], b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i][p] = reinterpret_cast<const u32x4*>(&As[buf][p][0])[rslot(wm_off + 32 * i + l31, half)];
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = reinterpret_cast<const u32x4*>(&BS(buf)[p][0])[rslot(wn_off + l31, half)];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0] = P::mfma(a[i][P::pa(t)], b[P::pb(t)], acc[i][0]);
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = reinterpret_cast<const u32x4*>(&BS(buf)[p][0])[rslot(wn_off + 32 + l31, half)];
        // branch-free k-tail masks (the block must stay one basic block for the scheduler)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[i][e] = e <= klast ? ra[i][e] : 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[i][e] = e <= klast ? rb[i][e] : 0.f;
#ifdef DTC_S3_FINE
        if constexpr (WIMG && !H2) {
            // (opt-in, measured round 3: 76.0 -> 74.0 us on 24576 x 512 x 512 alone, but 65.4 vs 64.9 ms per step in the overlapped
            // schedule, so off.)  Only the X side is converted: its seven dependency levels (both float4 of the thread side by
            // side, 4 to 8 VALU each) go one per MFMA gap -- an MFMA occupies the pipe for 32 cycles = 8 issue slots, so a level
            // issues under the MFMA in front of it instead of 28 VALU in a row pausing this wave's MFMA stream
            float r[8];
            u32 pk[3][4], u[8];
            auto mm = [&](int m) {
                const int t = m >> 1, i = m & 1;
                acc[i][1] = P::mfma(a[i][P::pa(t)], b[P::pb(t)], acc[i][1]);
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            mm(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = ra[e >> 2][e & 3];
#pragma unroll
            for (int q = 0; q < 4; ++q) pk[0][q] = cvt_pk_bf16(r[2 * q], r[2 * q + 1]);
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
#pragma unroll
            for (int lv = 0; lv < 2; ++lv) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u[2 * q] = pk[lv][q] << 16;
                    u[2 * q + 1] = pk[lv][q] & 0xffff0000u;
                }
                __builtin_amdgcn_sched_barrier(0);
                mm(2 + 3 * lv);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] -= __uint_as_float(u[e]);
                __builtin_amdgcn_sched_barrier(0);
                mm(3 + 3 * lv);
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[lv + 1][q] = cvt_pk_bf16(r[2 * q], r[2 * q + 1]);
                __builtin_amdgcn_sched_barrier(0);
                mm(4 + 3 * lv);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) As[buf ^ 1][p][aslot0 + 256 * i] = u32x2{pk[p][2 * i], pk[p][2 * i + 1]};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 8; m < 12; ++m) mm(m);
            return;
        }
#endif
        constexpr int NQ = NA + NB, MQ = 2 * NT / NQ;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // hard ordering between the pieces: 3 (W image: 6; fp16 terms: 3) MFMAs are issued, then the conversion of one float4 per
            // thread (~28 VALU + 3 LDS stores; fp16 terms: ~16 + 2) issues while they execute
            __builtin_amdgcn_sched_barrier(0);
            const P sp = P::split(q < NA ? ra[q] : rb[NB ? q - NA : 0], q < NA ? ex : 0);
#pragma unroll
            for (int p = 0; p < NP; ++p) (q < NA ? As[buf ^ 1] : BS(buf ^ 1))[p][aslot0 + 256 * (q & 1)] = sp.p[p];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t3 = 0; t3 < MQ; ++t3) {
                const int m = MQ * q + t3, t = m >> 1, i = m & 1;
                acc[i][1] = P::mfma(a[i][P::pa(t)], b[P::pb(t)], acc[i][1]);
            }
        }
    };
#endif

    enter_segment();
    load_w(S0{});
    load_x(S0{});
    store_stage(S0{});
    if constexpr (AH2) load_x(S1{});                // stage 1 on its way while stage 0's planes settle
    __syncthreads();
    // stage st in flight while the MFMAs consume the other buffer; two stages per trip (constant buffer indices)
    auto step = [&](auto bc) {
        constexpr int buf = decltype(bc)::value;
        load_w(std::integral_constant<int, buf ^ 1>{});
        load_x(std::integral_constant<int, AH2 ? buf : 0>{});        // AH2: stage + 2 into the set stage `buf` was converted from
#ifndef DTC_S3_NO_ILV
        stage_ilv(bc);
#else
        mfma_stage(bc);
        store_stage(std::integral_constant<int, buf ^ 1>{});
#endif
        // AH2: nothing in the stage waited for the weight chunk's LDS-DMA (the X loads the conversion waited for were issued BEFORE it):
        // vmcnt(NA) = only the NA X loads issued after it may still be in flight (expcnt / lgkmcnt untouched: 7 / 15)
        if constexpr (AH2) __builtin_amdgcn_s_waitcnt(0x0F70 | NA);
        __syncthreads();
    };
    // stages 0 .. total-1 are consumed two per trip; past the last stage the cursor loads nothing valid (klast < 0: the k-tail masks
    // zero the X side, so the W side of a pad stage -- whatever finite planes the buffer holds -- contributes nothing): an even
    // total ends with one unused load, an odd total with one all-zero stage
    for (int trip = (total + 1) >> 1; trip > 0; --trip) {
        step(S0{});
        step(S1{});
    }

    const bool full = (m0 + BM <= M) && (n0 + BN <= N);
    if constexpr (H2) {                                 // the sums carry 2^(ex + ew): scaled back exactly
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = poison ? __builtin_nanf("") : __builtin_ldexpf(acc[i][j][r], -(ex + ew));
    }
    u32 am = 0u;                                        // fp16 path: this lane's largest |stored value| (bit pattern)
    u32* am_lds = reinterpret_cast<u32*>(&Bs1[0][0]);   // four words for the workgroup-wide maximum (no epilogue touches this stage buffer)
    auto seen = [&](float v) {
        if constexpr (H2) {
            const u32 b = abs_bits(v);
            am = b > am ? b : am;
        }
    };
    __syncthreads();                                    // every wave is past its last operand read: LDS becomes the patches
    float* patch = reinterpret_cast<float*>(&As[0][0][0]) + wave * (32 * LDW);
    const int prow = lane >> 3, pc4 = lane & 7;

    if constexpr (EPI == EPI_MSE) {
        // e = (acc + bias) - target[tidx[row], tcol0 + col];  dY = e * scale;  partial = sum e^2 (double)
        const rsrc_t tres = make_rsrc_bytes(mse.target, mse.target_bytes);
        double sq = 0.0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn_off + 32 * j + l31;
            const bool cok = col < N;
            const float bv = (bias && cok) ? bias[col] : 0.f;
            const int row0 = m0 + wm_off + 32 * i + 4 * half;
            float tg[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {          // 16 gathered target loads in flight (rows past M read row 0, masked below)
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                const long long src = mse.tidx[row < M ? row : 0];
                tg[r] = bload(tres, (u32)((src * mse.ldt + mse.tcol0 + col) * 4) | (cok ? 0u : INVALID), 0u);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                if (cok && row < M) {
                    const float e = (acc[i][j][r] + bv) - tg[r];
                    Y[(long long)row * ldy + col] = e * mse.scale;
                    seen(e * mse.scale);
                    sq += (double)e * (double)e;
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
        double* red = reinterpret_cast<double*>(&Bs0[0][0]);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) mse.part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
        if constexpr (H2) amax_publish_block(h2.ya[0], am, am_lds);
        return;
    }
    if constexpr (EPI == EPI_FWD) {
        if (wide && full) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float bv = bias ? bias[n0 + wn_off + 32 * j + l31] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
                if (wmask) {                            // sign record of a ReLU layer (see linear_fwd_kernel)
                    unsigned bits = 0u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) bits |= acc[i][j][r] > 0.f ? (1u << r) : 0u;
                    wmask[((long long)((m0 + wm_off + 32 * i) >> 5) * 2 + half) * ldwm + n0 + wn_off + 32 * j + l31] = (unsigned short)bits;
                }
                patch_put(patch, acc[i][j], half, l31);
                float* yp = &Y[(long long)(m0 + wm_off + 32 * i + prow) * ldy + n0 + wn_off + 32 * j + 4 * pc4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = patch_get(patch, prow + 8 * p, pc4);
                    if (act == DTC_ACT_ELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);
                    } else if (act == DTC_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] <= 0.f ? 0.f : v[e];      // (NaN passes through, as torch.relu)
                    } else if (act != DTC_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], act);
                    }
                    *reinterpret_cast<f32x4*>(yp + (long long)(8 * p) * ldy) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) seen(v[e]);
                }
            }
            if constexpr (H2) amax_publish_block(h2.ya[0], am, am_lds);
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn_off + 32 * j + l31;
            const bool cok = col < N;
            const float bv = (bias && cok) ? bias[col] : 0.f;
            float* yp = Y + (long long)(m0 + wm_off + 32 * i + 4 * half) * ldy + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float v = act_fwd(acc[i][j][r] + bv, act);
                if (cok && m0 + wm_off + 32 * i + 4 * half + ro < M) {
                    yp[(long long)ro * ldy] = v;
                    seen(v);
                }
            }
        }
        if constexpr (H2) amax_publish_block(h2.ya[0], am, am_lds);
    } else {
        // ---- data-gradient epilogue: activation derivative, segmented destination (csrc/gemm.hip: linear_dgrad_kernel)
        const SegMatDev& dX = dg.dX;
        if (dg.rmask && full) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned bits = dg.rmask[((long long)((m0 + wm_off + 32 * i) >> 5) * 2 + half) * dg.ldm + n0 + wn_off + 32 * j + l31];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = (bits >> r) & 1u ? acc[i][j][r] : 0.f;
            }
            act = DTC_ACT_NONE;
        }
        const rsrc_t xres = make_rsrc_bytes(dg.Xs, (long long)M * dg.ldxs * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cj = n0 + wn_off + 32 * j;
            const int row_t = m0 + wm_off + 32 * i;
            const int sj = find_seg(dX, cj < N ? cj : 0);
            const SegDev sdj = dX.s[sj];
            const bool tile_wide = full && ((dg.wide_segs >> sj) & 1) && cj + 32 <= sdj.start + sdj.width && ((cj - sdj.start) & 3) == 0 &&
                                   (act == DTC_ACT_NONE || ((dg.wide_segs >> 4) & 1));
            if (dX.nseg > 1) am = 0u;                   // (fp16 path) several destination blocks: published per tile, each block has its own slot
            if (tile_wide) {                            // wave-uniform
                patch_put(patch, acc[i][j], half, l31);
                if (sdj.ptr == nullptr) continue;
                const int c4 = cj + 4 * pc4;
                float* dst = sdj.ptr + sdj.col0 + (c4 - sdj.start) + (long long)(row_t + prow) * sdj.ld;
                const float* ys = dg.Xs + (long long)(row_t + prow) * dg.ldxs + c4;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = patch_get(patch, prow + 8 * p, pc4);
                    if (act != DTC_ACT_NONE) {
                        const f32x4 y = *reinterpret_cast<const f32x4*>(ys + (long long)(8 * p) * dg.ldxs);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_bwd(v[e], y[e], act);
                    }
                    f32x4* q = reinterpret_cast<f32x4*>(dst + (long long)(8 * p) * sdj.ld);
                    if (sdj.accumulate) {
                        const f32x4 o = *q;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = o[e] + v[e];
                    }
                    *q = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) seen(v[e]);
                }
                if constexpr (H2) {
                    if (dX.nseg > 1) amax_publish(h2.ya[sj], am);
                }
                continue;
            }
            const int col = cj + l31;
            const bool cok = col < N;
            const SegDev sc = dX.s[find_seg(dX, cok ? col : 0)];
            const bool live = cok && sc.ptr != nullptr;
            float* dst = sc.ptr + sc.col0 + (col - sc.start);
            const int row0 = row_t + 4 * half;
            float y[16];
            if (act != DTC_ACT_NONE) {
                const u32 xoff = ((u32)row0 * (u32)dg.ldxs + (u32)col) * 4u | (cok ? 0u : INVALID);
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = bload(xres, xoff, (u32)(((r & 3) + 8 * (r >> 2)) * (int)dg.ldxs) * 4u);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r];
                if (act != DTC_ACT_NONE) v = act_bwd(v, y[r], act);
                if (live && row < M) {
                    float* q = dst + (long long)row * sc.ld;
                    v = sc.accumulate ? (*q + v) : v;
                    *q = v;
                    seen(v);
                }
            }
            if constexpr (H2) {                         // lanes of this tile may write different destination blocks: one update per lane
                if (dX.nseg > 1) {
                    u32* slot = live ? h2.ya[find_seg(dX, col)] : nullptr;
                    if (slot) slot += (lane & (AMAX_SUB - 1)) * AMAX_STRIDE;
                    if (slot && am > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, am);
                }
            }
        }
        if constexpr (H2) {                             // one destination block: the workgroup's maximum, one atomic
            if (dX.nseg == 1) amax_publish_block(h2.ya[0], am, am_lds);
        }
    }
}

// W [N, K] -> W^T [K, N] (the data gradient's reduction-contiguous operand; once per layer and optimiser step)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ W, float* __restrict__ WT, int N, int K) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int n = by + r, k = bx + tx;
        t[r][tx] = (n < N && k < K) ? W[(long long)n * K + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = bx + r, n = by + tx;
        if (k < K && n < N) WT[(long long)k * N + n] = t[tx][r];
    }
}

}  // namespace

namespace {

int wide_mask_s3(const SegMatDev& xd, const float* Xsaved, long long ldxs, int col_skip) {
    if (col_skip & 3) return 0;
    int m = 0;
    for (int i = 0; i < xd.nseg; ++i) {
        const SegDev& sd = xd.s[i];
        if (sd.ptr == nullptr || (dtc::aligned16(sd.ptr) && (sd.ld & 3) == 0 && ((sd.col0 - sd.start) & 3) == 0)) m |= 1 << i;
    }
    if (Xsaved == nullptr || (dtc::aligned16(Xsaved) && (ldxs & 3) == 0)) m |= 1 << 4;
    return m;
}

// W operand of one launch as the kernel's LDS planes (see wimage_kernel); DTC_S3_WIMG=0: the kernels split W inside their K loop
// as they do X (round-3 A/B switch; the data gradient then reads a transposed fp32 copy)
bool wimage_on() {
    static const bool on = [] {
        const char* e = getenv("DTC_S3_WIMG");
        return e ? atoi(e) != 0 : true;
    }();
    return on;
}
// job of one image: the column tiles [0, col_tiles) of the operand whose first row is row0, the stages of xd's segment walk
long long wimage_job(WimgJobDev& J, const float* W, void* img, int rows, int row0, long long ld, int trans, const SegMatDev& xd, int col_tiles,
                     bool h2 = false, long long welems = 0) {
    J.W = W;
    J.img = (u32x4*)img;
    J.ld = ld;
    J.rows = rows;
    J.row0 = row0;
    J.trans = trans;
    J.sg = WimgSegs{};
    J.sg.nseg = xd.nseg;
    J.total = 0;
    for (int i = 0; i < xd.nseg; ++i) {
        J.sg.start[i] = xd.s[i].start;
        J.sg.width[i] = xd.s[i].width;
        J.total += (xd.s[i].width + BK - 1) / BK;
    }
    J.block_end = col_tiles * J.total;
    const long long bytes = (long long)col_tiles * J.total * (h2 ? 2 * WIMG_PLANE : WIMG_CHUNK);
    J.welems = welems;
    J.wa = h2 ? reinterpret_cast<u32*>(reinterpret_cast<char*>(img) + bytes) : nullptr;      // inside dtc_s3_planes_bytes' slack
    return bytes;
}
// the image of ONE call, built on the call's stream right in front of the GEMM; returns the image bytes
long long build_wimage(const float* W, void* img, int rows, int row0, long long ld, int trans, const SegMatDev& xd, int col_tiles, hipStream_t s,
                       bool ready, bool h2 = false, long long welems = 0) {
    WimgGroup G;
    G.count = 1;
    const long long bytes = wimage_job(G.job[0], W, img, rows, row0, ld, trans, xd, col_tiles, h2, welems);
    if (!ready) {
        if (h2) {
            hipLaunchKernelGGL(wamax_kernel, dim3(WPART), dim3(256), 0, s, G);
            hipLaunchKernelGGL(wimage_kernel<true>, dim3((unsigned)G.job[0].block_end), dim3(256), 0, s, G);
        } else {
            hipLaunchKernelGGL(wimage_kernel<false>, dim3((unsigned)G.job[0].block_end), dim3(256), 0, s, G);
        }
    }
    return bytes;
}
// the data gradient's row operand (dZ [M, N], one plain segment) and the leading destination columns nothing is stored for
void dgrad_operands(SegMatDev& zin, const SegMatDev& dX, const float* dZ, long long lddz, int M, int N, int K, int& col_skip) {
    zin.nseg = 1;
    zin.gathers = 0;
    zin.idx = nullptr;
    for (int i = 0; i < 4; ++i) zin.s[i] = SegDev{nullptr, 0, 0, 0x7fffffff, 0, 0, 0, 0};
    zin.s[0] = SegDev{const_cast<float*>(dZ), lddz, 0, 0, N, 0, 0, M};
    col_skip = 0;
    for (int i = 0; i < dX.nseg && dX.s[i].ptr == nullptr; ++i) col_skip += dX.s[i].width;
    if (col_skip < K) col_skip &= ~31;              // whole 32-column sub-tiles only (the rest of a NULL segment is computed, not stored)
}

}  // namespace

// scratch of one split-path call: the weight image (at least the transposed fp32 weight of the DTC_S3_WIMG=0 data gradient)
extern "C" int64_t dtc_s3_planes_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    const long long img = (long long)WIMG_CHUNK * dtc::ceil_div(N, 128) * (dtc::ceil_div(K, BK) + 4);
    const long long wt = 4ll * N * K;
    return (img > wt ? img : wt) + 64;
}

namespace {
int fwd_s3(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask, void* wplanes, int wimage_ready,
           int M, int N, int K, int act, void* stream, bool h2, uint32_t* y_amax);
int dgrad_s3(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const float* Xsaved, int64_t ldxs, const uint16_t* relu_mask,
             void* wplanes, int wimage_ready, int M, int N, int K, int act, void* stream, bool h2, const uint32_t* dz_amax);
}  // namespace

// Y = act(X W^T + b) [+ the ReLU sign record when relu_mask != NULL] on the split-precision path
extern "C" int dtc_linear_fwd_s3(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                                 void* wplanes, int wimage_ready, int M, int N, int K, int act, void* stream) {
    DTC_REQUIRE(Y, "null pointer");
    return fwd_s3(X, W, b, Y, ldy, relu_mask, wplanes, wimage_ready, M, N, K, act, stream, false, nullptr);
}

namespace {
// The row operand's amax slots of an fp16 launch.  A segment that brings none gets the amax of exactly its block (rows < M, its
// columns) computed here, into the call's own scratch: `scratch` = four records behind the weight image's partial maxima
int h2_operand(const DtcSegMat* X, const SegMatDev& xd, int M, H2Arg& a, void* scratch, hipStream_t s) {
    a = H2Arg{};
    AmaxGroup G;
    G.count = 0;
    for (int i = 0; i < X->nseg; ++i) {
        if (X->seg[i].amax != nullptr) {
            a.xa[i] = X->seg[i].amax;
        } else {
            u32* slot = reinterpret_cast<u32*>(scratch) + i * (AMAX_RECORD_BYTES / 4);
            a.xa[i] = slot;
            amax_item(G, xd.s[i].ptr, xd.s[i].gather ? xd.idx : nullptr, xd.s[i].ld, xd.s[i].col0, xd.s[i].width, M, slot);
        }
    }
    DTC_REQUIRE(amax_group_run(G, scratch, 4 * AMAX_RECORD_BYTES, s), "hipMemsetAsync failed");
    return DTC_OK;
}
int fwd_s3(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
           void* wplanes, int wimage_ready, int M, int N, int K, int act, void* stream, bool h2, uint32_t* y_amax) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && Y != nullptr && ldy >= N, "bad shape M=%d N=%d K=%d ldy=%lld", M, N, K, (long long)ldy);
    DTC_REQUIRE(W, "null pointer");
    DTC_REQUIRE(act >= 0 && act <= DTC_ACT_SIGMOID, "bad activation %d", act);
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * ldy <= MAX_ELEMS * 4, "matrix too large");
    SegMatDev xd;
    int rc = to_dev(X, xd, K, false, M);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 128));
    const int wide = (Y && ldy % 4 == 0 && dtc::aligned16(Y)) ? 1 : 0;
    if (relu_mask)
        DTC_REQUIRE(act == DTC_ACT_RELU && wide && M % BM == 0 && N % 128 == 0, "sign record (split path): M=%d and N=%d must be multiples of 128", M, N);
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, K), 2.0 * M * (double)N * K, s, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    if (h2) {
        DTC_REQUIRE(wimage_on(), "fp16 path: needs weight images (DTC_S3_WIMG)");
        DTC_REQUIRE(wplanes && dtc::aligned16(wplanes), "null / unaligned weight-image scratch");
        H2Arg a;
        const long long ib = build_wimage(W, wplanes, N, 0, K, 0, xd, (int)dtc::ceil_div(N, 128), s, wimage_ready != 0, true, (long long)N * K);
        DTC_REQUIRE(ib + H2_TAIL_BYTES <= dtc_s3_planes_bytes(N, K), "weight-image buffer too small for the fp16 path's tail");
        rc = h2_operand(X, xd, M, a, reinterpret_cast<char*>(wplanes) + ib + 4 * WPART, s);
        if (rc != DTC_OK) return rc;
        a.ya[0] = y_amax;
        a.wa = reinterpret_cast<const u32*>(reinterpret_cast<const char*>(wplanes) + ib);
        hipLaunchKernelGGL((linear_s3_kernel<EPI_FWD, true, true>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy,
                           M, N, K, act, wide, (unsigned short*)relu_mask, N, DgradEpi{}, MseEpiS3{}, (const u32x4*)wplanes, ib, a);
    } else if (wimage_on()) {
        DTC_REQUIRE(wplanes && dtc::aligned16(wplanes), "null / unaligned weight-image scratch");
        const long long ib = build_wimage(W, wplanes, N, 0, K, 0, xd, (int)dtc::ceil_div(N, 128), s, wimage_ready != 0);
        hipLaunchKernelGGL((linear_s3_kernel<EPI_FWD, true>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy,
                           M, N, K, act, wide, (unsigned short*)relu_mask, N, DgradEpi{}, MseEpiS3{}, (const u32x4*)wplanes, ib, H2Arg{});
    } else {
        hipLaunchKernelGGL((linear_s3_kernel<EPI_FWD, false>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy,
                           M, N, K, act, wide, (unsigned short*)relu_mask, N, DgradEpi{}, MseEpiS3{}, (const u32x4*)nullptr, 0ll, H2Arg{});
    }
    return dtc::check_launch("linear_fwd_s3");
}
}  // namespace

// The same layer on the two-term fp16 path (s3_core.hpp): a segment of X brings the amax slot of its source tensor (DtcSeg.amax) or
// none (NULL: computed here, one memset + one small launch in front of the GEMM), y_amax (may be NULL) receives max(*y_amax, largest |Y| written) -- zero it before the first kernel that writes Y
extern "C" int dtc_linear_fwd_h2(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                                 void* wplanes, int wimage_ready, uint32_t* y_amax, int M, int N, int K, int act, void* stream) {
    DTC_REQUIRE(Y, "null pointer");
    return fwd_s3(X, W, b, Y, ldy, relu_mask, wplanes, wimage_ready, M, N, K, act, stream, true, y_amax);
}

// dX = (dZ W) * act'(.), W [N, K] as stored (the kernel's reduction-contiguous operand, the planes of W^T, is prepared here);
// same destination contract as dtc_linear_dgrad / dtc_linear_dgrad_mask (relu_mask != NULL: the ReLU derivative from the sign
// record, Xsaved unused)
extern "C" int dtc_linear_dgrad_s3(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const float* Xsaved,
                                   int64_t ldxs, const uint16_t* relu_mask, void* wplanes, int wimage_ready, int M, int N, int K, int act,
                                   void* stream) {
    return dgrad_s3(dZ, lddz, W, dX, Xsaved, ldxs, relu_mask, wplanes, wimage_ready, M, N, K, act, stream, false, nullptr);
}

namespace {
int dgrad_s3(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX,
             const float* Xsaved, int64_t ldxs, const uint16_t* relu_mask, void* wplanes, int wimage_ready, int M, int N,
             int K, int act, void* stream, bool h2, const uint32_t* dz_amax) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddz >= N, "bad shape");
    DTC_REQUIRE(dZ && W && wplanes && dtc::aligned16(wplanes), "null pointer / unaligned scratch");
    DTC_REQUIRE(act >= 0 && act <= DTC_ACT_SIGMOID, "bad activation %d", act);
    DTC_REQUIRE(relu_mask || act == DTC_ACT_NONE || (Xsaved != nullptr && ldxs >= K), "activation derivative needs Xsaved");
    DTC_REQUIRE((act == DTC_ACT_NONE && !relu_mask) || (dX && dX->nseg == 1), "activation derivative needs a single-segment destination");
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * lddz <= MAX_ELEMS, "matrix too large");
    DgradEpi dg{};
    int rc = to_dev(dX, dg.dX, K, true, 0);
    if (rc != DTC_OK) return rc;
    int col_skip;
    SegMatDev zin;                                    // the row operand of the product: dZ [M, N], one plain segment
    dgrad_operands(zin, dg.dX, dZ, (long long)lddz, M, N, K, col_skip);
    DTC_REQUIRE(col_skip < K, "every destination segment is NULL");
    dg.Xs = relu_mask ? nullptr : Xsaved;
    dg.ldxs = ldxs;
    dg.rmask = (const unsigned short*)relu_mask;
    dg.ldm = K;
    dg.col_skip = col_skip;
    dg.wide_segs = wide_mask_s3(dg.dX, dg.Xs, ldxs, col_skip);
    if (relu_mask) DTC_REQUIRE(M % BM == 0 && K % 128 == 0 && col_skip == 0, "sign record (split path): M=%d and K=%d must be multiples of 128", M, K);
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(K - col_skip, 128));
    double bytes = 4.0 * ((double)M * N + (double)N * K);
    for (int i = 0; i < dg.dX.nseg; ++i)
        if (dg.dX.s[i].ptr) bytes += 4.0 * M * dg.dX.s[i].width * (dg.dX.s[i].accumulate ? 2.0 : 1.0);
    if (relu_mask) bytes += 0.125 * M * (double)K;
    else if (act != DTC_ACT_NONE) bytes += 4.0 * M * (double)K;
    dtc::ProfScope prof(dtc::prof_shape_name("linear_dgrad", M, N, K), 2.0 * M * (double)N * (K - col_skip), s, bytes);
    // roles inside the kernel: output columns = K of the layer, reduction = N of the layer
    if (h2) {
        DTC_REQUIRE(wimage_on(), "fp16 path: needs weight images");
        H2Arg a{};
        for (int i = 0; i < dX->nseg; ++i) a.ya[i] = dX->seg[i].amax;
        const long long ib = build_wimage(W, wplanes, K, col_skip, K, 1, zin, (int)dtc::ceil_div(K - col_skip, 128), s, wimage_ready != 0, true, (long long)N * K);
        DTC_REQUIRE(ib + H2_TAIL_BYTES <= dtc_s3_planes_bytes(K, N), "weight-image buffer too small for the fp16 path's tail");
        a.xa[0] = dz_amax;
        if (dz_amax == nullptr) {                     // dZ came without a slot: its amax into the call's scratch behind the image
            u32* slot = reinterpret_cast<u32*>(reinterpret_cast<char*>(wplanes) + ib + 4 * WPART);
            AmaxGroup G;
            G.count = 0;
            amax_item(G, dZ, nullptr, (long long)lddz, 0, N, M, slot);
            DTC_REQUIRE(amax_group_run(G, slot, AMAX_RECORD_BYTES, s), "hipMemsetAsync failed");
            a.xa[0] = slot;
        }
        a.wa = reinterpret_cast<const u32*>(reinterpret_cast<const char*>(wplanes) + ib);
        hipLaunchKernelGGL((linear_s3_kernel<EPI_DGRAD, true, true>), dim3(grid), dim3(256), 0, s, zin, (const float*)nullptr, (const float*)nullptr,
                           (float*)nullptr, 0ll, M, K, N, relu_mask ? (int)DTC_ACT_RELU : act, 0, (unsigned short*)nullptr, 0, dg, MseEpiS3{},
                           (const u32x4*)wplanes, ib, a);
    } else if (wimage_on()) {
        const long long ib = build_wimage(W, wplanes, K, col_skip, K, 1, zin, (int)dtc::ceil_div(K - col_skip, 128), s, wimage_ready != 0);
        hipLaunchKernelGGL((linear_s3_kernel<EPI_DGRAD, true>), dim3(grid), dim3(256), 0, s, zin, (const float*)nullptr, (const float*)nullptr,
                           (float*)nullptr, 0ll, M, K, N, relu_mask ? (int)DTC_ACT_RELU : act, 0, (unsigned short*)nullptr, 0, dg, MseEpiS3{},
                           (const u32x4*)wplanes, ib, H2Arg{});
    } else {
        float* WT = (float*)wplanes;
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)dtc::ceil_div(K, 32), (unsigned)dtc::ceil_div(N, 32)), dim3(256), 0, s, W, WT, N, K);
        hipLaunchKernelGGL((linear_s3_kernel<EPI_DGRAD, false>), dim3(grid), dim3(256), 0, s, zin, (const float*)WT, (const float*)nullptr,
                           (float*)nullptr, 0ll, M, K, N, relu_mask ? (int)DTC_ACT_RELU : act, 0, (unsigned short*)nullptr, 0, dg, MseEpiS3{},
                           (const u32x4*)nullptr, 0ll, H2Arg{});
    }
    return dtc::check_launch("linear_dgrad_s3");
}
}  // namespace

// The data gradient on the two-term fp16 path: dz_amax = amax slot of dZ; every destination block with a slot (dX->seg[i].amax, may be
// NULL) receives max(slot, largest |value| written to it) -- for an accumulating block the value written is the sum
extern "C" int dtc_linear_dgrad_h2(const float* dZ, int64_t lddz, const uint32_t* dz_amax, const float* W, const DtcSegMat* dX, const float* Xsaved,
                                   int64_t ldxs, const uint16_t* relu_mask, void* wplanes, int wimage_ready, int M, int N, int K, int act,
                                   void* stream) {
    return dgrad_s3(dZ, lddz, W, dX, Xsaved, ldxs, relu_mask, wplanes, wimage_ready, M, N, K, act, stream, true, dz_amax);
}

extern "C" int64_t dtc_linear_fwd_mse_s3_parts(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 128));
}

// dtc_linear_fwd_mse on the split-precision path (sq_part: dtc_linear_fwd_mse_s3_parts(M, N) doubles)
namespace {
int fwd_mse_s3(const DtcSegMat* X, const float* W, const float* b, const float* target, int64_t ldt,
               int64_t target_rows, int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy,
               double* sq_part, void* wplanes, int wimage_ready, int M, int N, int K, void* stream, bool h2, uint32_t* dy_amax) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N, "bad shape M=%d N=%d K=%d", M, N, K);
    DTC_REQUIRE(W && target && tidx && dY && sq_part, "null pointer");
    DTC_REQUIRE(tcol0 >= 0 && tcol0 + N <= ldt && target_rows > 0, "target columns [%d, %d) outside its %lld-wide rows", tcol0,
                tcol0 + N, (long long)ldt);
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * lddy <= MAX_ELEMS && target_rows * ldt <= MAX_ELEMS, "matrix too large");
    SegMatDev xd;
    int rc = to_dev(X, xd, K, false, M);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, 128));
    const MseEpiS3 mse{target, (const long long*)tidx, (long long)ldt, target_rows * ldt * 4, tcol0, scale, sq_part};
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, K), 2.0 * M * (double)N * K, s,
                        4.0 * ((double)M * K + (double)N * K + 2.0 * M * N));
    if (h2) {
        DTC_REQUIRE(wimage_on() && wplanes && dtc::aligned16(wplanes), "fp16 path: needs weight images; null / unaligned weight-image scratch");
        H2Arg a;
        const long long ib = build_wimage(W, wplanes, N, 0, K, 0, xd, (int)dtc::ceil_div(N, 128), s, wimage_ready != 0, true, (long long)N * K);
        DTC_REQUIRE(ib + H2_TAIL_BYTES <= dtc_s3_planes_bytes(N, K), "weight-image buffer too small for the fp16 path's tail");
        rc = h2_operand(X, xd, M, a, reinterpret_cast<char*>(wplanes) + ib + 4 * WPART, s);
        if (rc != DTC_OK) return rc;
        a.ya[0] = dy_amax;
        a.wa = reinterpret_cast<const u32*>(reinterpret_cast<const char*>(wplanes) + ib);
        hipLaunchKernelGGL((linear_s3_kernel<EPI_MSE, true, true>), dim3(grid), dim3(256), 0, s, xd, W, b, dY, (long long)lddy,
                           M, N, K, (int)DTC_ACT_NONE, 0, (unsigned short*)nullptr, 0, DgradEpi{}, mse, (const u32x4*)wplanes, ib, a);
    } else if (wimage_on()) {
        DTC_REQUIRE(wplanes && dtc::aligned16(wplanes), "null / unaligned weight-image scratch");
        const long long ib = build_wimage(W, wplanes, N, 0, K, 0, xd, (int)dtc::ceil_div(N, 128), s, wimage_ready != 0);
        hipLaunchKernelGGL((linear_s3_kernel<EPI_MSE, true>), dim3(grid), dim3(256), 0, s, xd, W, b, dY, (long long)lddy,
                           M, N, K, (int)DTC_ACT_NONE, 0, (unsigned short*)nullptr, 0, DgradEpi{}, mse, (const u32x4*)wplanes, ib, H2Arg{});
    } else {
        hipLaunchKernelGGL((linear_s3_kernel<EPI_MSE, false>), dim3(grid), dim3(256), 0, s, xd, W, b, dY, (long long)lddy,
                           M, N, K, (int)DTC_ACT_NONE, 0, (unsigned short*)nullptr, 0, DgradEpi{}, mse, (const u32x4*)nullptr, 0ll, H2Arg{});
    }
    return dtc::check_launch("linear_fwd_mse_s3");
}
}  // namespace

extern "C" int dtc_linear_fwd_mse_s3(const DtcSegMat* X, const float* W, const float* b, const float* target, int64_t ldt,
                                     int64_t target_rows, int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy,
                                     double* sq_part, void* wplanes, int wimage_ready, int M, int N, int K, void* stream) {
    return fwd_mse_s3(X, W, b, target, ldt, target_rows, tcol0, tidx, scale, dY, lddy, sq_part, wplanes, wimage_ready, M, N, K, stream, false, nullptr);
}

// dtc_linear_fwd_mse_s3 on the two-term fp16 path; dy_amax (may be NULL): amax slot of the gradient dY it writes
extern "C" int dtc_linear_fwd_mse_h2(const DtcSegMat* X, const float* W, const float* b, const float* target, int64_t ldt,
                                     int64_t target_rows, int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy,
                                     double* sq_part, void* wplanes, int wimage_ready, uint32_t* dy_amax, int M, int N, int K, void* stream) {
    return fwd_mse_s3(X, W, b, target, ldt, target_rows, tcol0, tidx, scale, dY, lddy, sq_part, wplanes, wimage_ready, M, N, K, stream, true, dy_amax);
}

// The weight images of `count` later calls in one launch per WIMG_MAX_JOBS jobs (a trainer builds the images of all layers of an
// optimisation step at its start and passes wimage_ready = 1 to the calls): job i describes the call exactly as the call will --
// trans == 0: dtc_linear_fwd_s3 / dtc_linear_fwd_mse_s3 with operand `seg` = X; trans == 1: dtc_linear_dgrad_s3 with `seg` = dX.
namespace {
int wimage_group(const DtcWimgJob* jobs, int count, void* stream, bool h2) {
    DTC_REQUIRE(jobs && count > 0, "no jobs");
    DTC_REQUIRE(wimage_on(), "weight images are switched off (DTC_S3_WIMG=0)");
    hipStream_t s = (hipStream_t)stream;
    double elems = 0.0;
    for (int i = 0; i < count; ++i) elems += (double)jobs[i].N * jobs[i].K;
    dtc::ProfScope prof("wimage", 0.0, s, (h2 ? 12.0 : 10.0) * elems);   // 4 bytes read, 6 written per weight (fp16 terms: read twice, 4 written)
    WimgGroup G;
    G.count = 0;
    auto flush = [&]() {
        if (G.count == 0) return;
        if (h2) {
            hipLaunchKernelGGL(wamax_kernel, dim3((unsigned)(WPART * G.count)), dim3(256), 0, s, G);
            hipLaunchKernelGGL(wimage_kernel<true>, dim3((unsigned)G.job[G.count - 1].block_end), dim3(256), 0, s, G);
        } else {
            hipLaunchKernelGGL(wimage_kernel<false>, dim3((unsigned)G.job[G.count - 1].block_end), dim3(256), 0, s, G);
        }
        G.count = 0;
    };
    for (int i = 0; i < count; ++i) {
        const DtcWimgJob& h = jobs[i];
        DTC_REQUIRE(h.W && h.img && dtc::aligned16(h.img) && h.seg && h.N > 0 && h.K > 0, "job %d: null pointer / bad shape", i);
        DTC_REQUIRE((long long)h.N * h.K <= MAX_ELEMS, "job %d: matrix too large", i);
        SegMatDev xd;
        WimgJobDev& J = G.job[G.count];
        if (h.trans) {
            int rc = to_dev(h.seg, xd, h.K, true, 0);
            if (rc != DTC_OK) return rc;
            SegMatDev zin;
            int col_skip;
            dgrad_operands(zin, xd, nullptr, 0, 0, h.N, h.K, col_skip);
            DTC_REQUIRE(col_skip < h.K, "job %d: every destination segment is NULL", i);
            wimage_job(J, h.W, h.img, h.K, col_skip, h.K, 1, zin, (int)dtc::ceil_div(h.K - col_skip, 128), h2, (long long)h.N * h.K);
        } else {
            int rc = to_dev(h.seg, xd, h.K, false, 0);
            if (rc != DTC_OK) return rc;
            wimage_job(J, h.W, h.img, h.N, 0, h.K, 0, xd, (int)dtc::ceil_div(h.N, 128), h2, (long long)h.N * h.K);
        }
        if (G.count > 0) J.block_end += G.job[G.count - 1].block_end;
        if (++G.count == WIMG_MAX_JOBS) flush();
    }
    flush();
    return dtc::check_launch("s3_wimage_group");
}
}  // namespace

extern "C" int dtc_s3_wimage_group(const DtcWimgJob* jobs, int count, void* stream) { return wimage_group(jobs, count, stream, false); }
// the images of the two-term fp16 calls (dtc_linear_fwd_h2 / _mse_h2 / dtc_linear_dgrad_h2): same jobs, same buffers (dtc_s3_planes_bytes)
extern "C" int dtc_h2_wimage_group(const DtcWimgJob* jobs, int count, void* stream) { return wimage_group(jobs, count, stream, true); }

// ---- amax of an operand that no kernel published one for (rollout storage, a tensor written by a torch op): slot = bit pattern of the
// largest |x| over the rows < M (gathered through X.idx where a segment asks for it) and the columns of every segment
namespace {
__global__ __launch_bounds__(256) void amax_kernel(const SegMatDev X, int M, u32* __restrict__ slot) {
    u32 m = 0u;
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
        for (int i = 0; i < X.nseg; ++i) {
            const SegDev& sd = X.s[i];
            const float* src = sd.ptr + (sd.gather ? X.idx[row] : (long long)row) * sd.ld + sd.col0;
            for (int c = threadIdx.x; c < sd.width; c += 256) {
                const u32 b = abs_bits(src[c]);
                m = b > m ? b : m;
            }
        }
    }
    amax_publish(slot, m);
}
}  // namespace

extern "C" int64_t dtc_amax_record_bytes(void) { return AMAX_RECORD_BYTES; }

extern "C" int dtc_amax(const DtcSegMat* X, int M, uint32_t* slot, void* stream) {
    DTC_REQUIRE(X && slot && M > 0, "null pointer / bad shape");
    SegMatDev xd;
    int rc = to_dev(X, xd, X->cols, false, M);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    DTC_REQUIRE(hipMemsetAsync(slot, 0, AMAX_RECORD_BYTES, s) == hipSuccess, "hipMemsetAsync failed");
    const int grid = M < 2048 ? M : 2048;
    hipLaunchKernelGGL(amax_kernel, dim3(grid), dim3(256), 0, s, xd, M, (u32*)slot);
    return dtc::check_launch("amax");
}
