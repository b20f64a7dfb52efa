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
#include "amax.hpp"
#include "gemm_core.hpp"

namespace {

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

// ---- stage image of a reduction-contiguous operand (X rows, W rows of the forward product; dZ rows of the data
// gradient).  Stage = 16 k.  Row r of the tile holds its 16 k-values as four 16-byte chunks, chunk c stored at position
// c ^ ((r >> 2) & 3).  Loader thread t: row t / 4 (+ 64 per pass), chunk t % 4 -> ONE dwordx4 buffer load + ONE
// ds_write_b128 per 4 k-values (the 8 lanes of a write group cover two rows x 4 chunks = banks 0-15 / 16-31: conflict
// free).  MFMA lane (row i = lane & 31, half h = lane >> 5) reads chunk 2q + h of its row (q = 0, 1) with ONE
// ds_read_b128 (the XOR spreads the 16 lanes of a read group over all 64 banks) and feeds the four values to four
// consecutive MFMA k-steps; k-step (q, t) therefore multiplies k = 8q + t in half 0 and k = 8q + 4 + t in half 1.  Both
// operands use the same assignment, every k of the stage is used exactly once: the sum is a permuted-order fp32 fma
// chain (deterministic).  Against the [k][row] image + dword loads of round 1: 3 instead of 12 VMEM and LDS-write
// instructions and 6 instead of 24 LDS reads per 16 MFMAs.
// k tail of a stage: the chunk is loaded whole and the elements past the segment's last column are zeroed in the DATA
// (v_cndmask), not by per-dword addresses -- one address register per chunk, same register footprint as a full stage.
// The over-read (at most 12 bytes past the segment's columns) stays inside the source matrix or, behind its last row,
// behind the descriptor's num_records, where the hardware returns 0 (gfx950 range-checks a dwordx4 buffer load per dword:
// tools/probes/oob_x4.hip).
__device__ __forceinline__ f32x4 ktail(f32x4 v, int k0, int klast) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k0 + e <= klast) ? v[e] : 0.f;
    return v;
}

#ifdef DTC_TRACE
// tools/gemm_lab: per-block shader-clock stamps (start, after the first stage, after the K loop, end) of the forward kernel
__device__ unsigned long long* g_trace = nullptr;
#define DTC_STAMP(i) if (threadIdx.x == 0 && g_trace) g_trace[(size_t)blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define DTC_STAMP(i)
#endif

#ifndef DTC_FWD_WAVES
#define DTC_FWD_WAVES 4      // workgroups per CU the register allocator aims at for the forward kernel (tuning aid)
#endif
// DEEP = true: the variant for launches that leave a CU with about one workgroup (rollout-sized batches): operand loads run TWO stages ahead of the MFMAs (two register sets, k-tail masks applied when a set is stored
// to LDS, every stage takes the masked form) -- the single-stage pipeline relies on the other workgroups of the CU to
// cover the load latency, and there are none.
template <int BN, bool MSE = false, bool DEEP = false>
__global__ __launch_bounds__(256, DEEP ? 3 : (BN == 128 ? 2 : DTC_FWD_WAVES)) void linear_fwd_kernel(const SegMatDev X, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Y,
                                                         long long ldy, int M, int N, int K, int act, int wide,
                                                         const MseEpi mse, unsigned short* __restrict__ rmask = nullptr,
                                                         int ldm = 0, amax_u32* __restrict__ yamax = nullptr) {
    // yamax (round 4): amax record of Y for the two-term fp16 GEMM path that may consume it (amax.hpp); NULL: nothing published
    amax_u32 am = 0u;
    auto seen = [&](float v) { am = abs_bits(v) > am ? abs_bits(v) : am; };
    // wave layout: BN <= 64: four waves stacked along the rows, each 32 x BN; BN = 128: 2 x 2 waves, each 64 x 64 (2 x 2 MFMA
    // tiles: 8 ds_read_b128 feed 32 MFMAs per stage instead of 6 for 16, one barrier covers twice the MFMA work)
    constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int NA = BM / 64;                      // loader passes over the 128 X rows
    constexpr int NB = (BN + 63) / 64;               // loader passes over the W rows of the tile
    DTC_STAMP(0);
    __shared__ f32x4 As[2][BM * 4];
    __shared__ f32x4 Bs[2][BN * 4];
    int tr, tc;
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, tr, tc)) {
        if (MSE && threadIdx.x == 0) mse.part[blockIdx.x] = 0.0;      // padding block: its partial slot still gets summed
        return;
    }
    const int m0 = tr * BM, n0 = tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = (wave / WN) * (32 * TM), wn_off = (wave % WN) * (32 * TN);
    const int half = lane >> 5, l31 = lane & 31;

    // loader geometry: thread owns chunk lch of rows lrow (+64) of X and of row lrow of W
    const int lrow = tid >> 2, lch = tid & 3;
    const bool bthread = BN >= 64 || tid < BN * 4;
    int arow[NA], grow[NA], aslot[NA];
    const bool any_gather = X.gathers != 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = lrow + 64 * i, m = m0 + r;
        arow[i] = m < M ? m : -1;
        grow[i] = (any_gather && m < M) ? (int)X.idx[m] : arow[i];
        aslot[i] = kslot(r, lch);
    }
    u32 woff[NB];
    int bslot[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int wn = n0 + lrow + 64 * i;
        woff[i] = (bthread && wn < N) ? (u32)(wn * K + 4 * lch) * 4u : INVALID;
        bslot[i] = kslot(lrow + 64 * i, lch);
    }
    const rsrc_t wres = make_rsrc_bytes(W, (long long)N * K * 4);

    // K loop: segment by segment.  The steady-state loop over the full stages of a segment is ONE basic block (loads
    // of stage kt+1, MFMAs of stage kt, LDS stores, barrier) with no masks and no branches; the last stage of a
    // segment (k tail -> per-dword masks) and the hop into the next segment are peeled copies.
    SegDev sd = X.s[0];
    rsrc_t ares;
    u32 aoff[NA];
    auto enter_segment = [&]() {
        ares = make_rsrc_bytes(sd.ptr, (long long)sd.rows * sd.ld * 4);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = sd.gather ? grow[i] : arow[i];
            aoff[i] = r >= 0 ? ((u32)r * (u32)sd.ld + (u32)(sd.col0 + 4 * lch)) * 4u : INVALID;
        }
    };
    f32x4 ra[NA], rb[NB];
    auto load_stage = [&](auto masked, int kt) {     // stage kt of the current segment -> registers
        const u32 ka = (u32)(kt * BK) * 4u, kw = (u32)(sd.start + kt * BK) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload4(ares, aoff[i], ka);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload4(wres, woff[i], kw);
        if (decltype(masked)::value) {
            const int k0 = kt * BK + 4 * lch;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = ktail(ra[i], k0, sd.width - 1);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = ktail(rb[i], k0, sd.width - 1);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) As[buf][aslot[i]] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (BN >= 64 || bthread) Bs[buf][bslot[i]] = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mfma_stage = [&](int buf) {
        f32x4 a[TM][2], b[TN][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i][q] = As[buf][kslot(wm_off + 32 * i + l31, 2 * q + half)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j][q] = Bs[buf][kslot(wn_off + 32 * j + l31, 2 * q + half)];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q][t], b[j][q][t], acc[i][j], 0, 0, 0);
    };

    int buf = 0;
    auto step = [&](auto masked, int kt_next) {     // stage kt_next in flight while the MFMAs consume LDS[buf]
        load_stage(masked, kt_next);
        mfma_stage(buf);
        store_stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    };
    enter_segment();
    if constexpr (DEEP) {
        int total = 0;
        for (int i = 0; i < X.nseg; ++i) total += (X.s[i].width + BK - 1) / BK;
        int lseg = 0, lkt = 0, ln = (sd.width + BK - 1) / BK;          // load cursor: next stage to request
        f32x4 qa[2][NA], qb[2][NB];
        int qtail[2];                                                    // last valid element (0..3) of a set's chunks
        auto request = [&](f32x4 (&a)[NA], f32x4 (&b)[NB], int& tail) {
            const u32 ka = (u32)(lkt * BK) * 4u, kw = (u32)(sd.start + lkt * BK) * 4u;
#pragma unroll
            for (int i = 0; i < NA; ++i) a[i] = bload4(ares, aoff[i], ka);
#pragma unroll
            for (int i = 0; i < NB; ++i) b[i] = bload4(wres, woff[i], kw);
            tail = (sd.width - 1) - (lkt * BK + 4 * lch);
            if (++lkt == ln && ++lseg < X.nseg) {
                sd = X.s[lseg];
                enter_segment();
                lkt = 0;
                ln = (sd.width + BK - 1) / BK;
            }
        };
        auto commit = [&](int b, f32x4 (&a)[NA], f32x4 (&w)[NB], int tail) {   // registers -> LDS (waits for the loads here)
#pragma unroll
            for (int i = 0; i < NA; ++i) As[b][aslot[i]] = ktail(a[i], 0, tail);
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (BN >= 64 || bthread) Bs[b][bslot[i]] = ktail(w[i], 0, tail);
        };
        request(qa[0], qb[0], qtail[0]);
        if (total > 1) request(qa[1], qb[1], qtail[1]);
        commit(0, qa[0], qb[0], qtail[0]);
        __syncthreads();
        for (int st = 0; st < total; st += 2) {
            if (st + 2 < total) request(qa[0], qb[0], qtail[0]);
            mfma_stage(0);
            if (st + 1 < total) commit(1, qa[1], qb[1], qtail[1]);
            __syncthreads();
            if (st + 1 >= total) break;
            if (st + 3 < total) request(qa[1], qb[1], qtail[1]);
            mfma_stage(1);
            if (st + 2 < total) commit(0, qa[0], qb[0], qtail[0]);
            __syncthreads();
        }
    } else {
    load_stage(Masked{}, 0);
    store_stage(0);
    __syncthreads();
    DTC_STAMP(1);
    for (int seg = 0;;) {
        const int n = (sd.width + BK - 1) / BK;
        const int nfull = sd.width / BK;                         // stages without a k tail
        for (int kt = 1; kt < nfull; ++kt) step(Full{}, kt);
        if (n > nfull && n > 1) step(Masked{}, n - 1);
        if (++seg == X.nseg) break;
        sd = X.s[seg];
        enter_segment();
        step(Masked{}, 0);
    }
    mfma_stage(buf);
    }
    DTC_STAMP(2);

    const bool full = (m0 + BM <= M) && (n0 + BN <= N);
    if (MSE) {
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
                    seen(e * mse.scale);
                    sq += (double)e * (double)e;
                }
            }
        }
        sq = wave_sum_f64(sq);
        double* red = reinterpret_cast<double*>(&As[0][0]);
        __syncthreads();                                // all waves are past their last LDS read
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) mse.part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
        amax_publish(yamax, am);
        return;
    }
    if (wide && full) {
        // bias + activation in the accumulator layout, transpose through the wave's LDS patch, dwordx4 stores
        __syncthreads();                                // every wave is past its last operand read
        float* patch = reinterpret_cast<float*>(&As[0][0]) + wave * (32 * LDW);
        const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bv = bias ? bias[n0 + wn_off + 32 * j + l31] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
            if (rmask) {
                // sign record of the ReLU (dtc_linear_fwd_mask): one bit per output, packed in the accumulator layout --
                // lane (column, half) holds 16 rows of its column -> one 16-bit word per (32-row block, half, column).  The
                // data gradient reads 1 bit instead of the 4-byte activation (y > 0 <=> pre-activation > 0)
                unsigned bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) bits |= acc[i][j][r] > 0.f ? (1u << r) : 0u;
                rmask[((long long)((m0 + wm_off + 32 * i) >> 5) * 2 + half) * ldm + n0 + wn_off + 32 * j + l31] = (unsigned short)bits;
            }
            patch_put(patch, acc[i][j], half, l31);
            float* yp = &Y[(long long)(m0 + wm_off + 32 * i + prow) * ldy + n0 + wn_off + 32 * j + 4 * pc4];
            // the activation is wave-uniform: one branch per tile, not a select around expm1f per element
            if (act == DTC_ACT_ELU) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = patch_get(patch, prow + 8 * p, pc4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);
                    *reinterpret_cast<f32x4*>(yp + (long long)(8 * p) * ldy) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) seen(v[e]);
                }
            } else if (act == DTC_ACT_RELU) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = patch_get(patch, prow + 8 * p, pc4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] <= 0.f ? 0.f : v[e];      // (NaN passes through, as torch.relu)
                    *reinterpret_cast<f32x4*>(yp + (long long)(8 * p) * ldy) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) seen(v[e]);
                }
            } else if (act == DTC_ACT_NONE) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 v = patch_get(patch, prow + 8 * p, pc4);
                    *reinterpret_cast<f32x4*>(yp + (long long)(8 * p) * ldy) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) seen(v[e]);
                }
            } else {                                // selu / lrelu / tanh / sigmoid: not on this model's path, generic form
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = patch_get(patch, prow + 8 * p, pc4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], act);
                    *reinterpret_cast<f32x4*>(yp + (long long)(8 * p) * ldy) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) seen(v[e]);
                }
            }
        }
        DTC_STAMP(3);
        amax_publish(yamax, am);
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
            if (full || (cok && m0 + wm_off + 32 * i + 4 * half + ro < M)) {
                yp[(long long)ro * ldy] = v;
                seen(v);
            }
        }
    }
    amax_publish(yamax, am);
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
// A = dZ rows (reduction index n contiguous): the stage image of the forward kernel (dwordx4 loads, ds_read_b128
// fragments, k-step (q, t) = n 8q + t / 8q + 4 + t in the two lane halves).  B = W[n][c] has the reduction index as its
// ROW: float4 loads along c, LDS image [n][col] (+4 pad), one ds_read_b32 per MFMA step from row 8q + 4h + t.
// Columns of B past K read the next row of W (or 0 behind its last row): they only feed output columns that are never
// stored, so the B loads carry no column masks at all.
template <int BN, bool DEEP = false>
__global__ __launch_bounds__(256, DEEP ? 3 : 6) void linear_dgrad_kernel(const float* __restrict__ dZ, long long lddz,
                                                           const float* __restrict__ W, const SegMatDev dX,
                                                           const float* __restrict__ Xs, long long ldxs, int M, int N,
                                                           int K, int act, int split_n, long long split_dst, int col_skip,
                                                           int wide_segs, const unsigned short* __restrict__ rmask = nullptr,
                                                           int ldm = 0, amax_u32* __restrict__ xamax = nullptr) {
    // xamax (round 4): amax record of a SINGLE whole-tensor destination (two-term fp16 GEMM path, amax.hpp); NULL: nothing published
    amax_u32 am = 0u;
    constexpr int TN = BN / 32;
    constexpr int NA = BM / 64;
    constexpr int LDB = BN + 4;
    __shared__ f32x4 As[2][BM * 4];
    __shared__ float Bs[2][BK][LDB];
    int tr, tc;
    // col_skip: leading columns whose destination is NULL (inputs that need no gradient, e.g. the raw observations in
    // front of the actor's features) -- the column tiles start behind them, nothing is computed for them
    if (!map_tile(blockIdx.x, (M + BM - 1) / BM, (K - col_skip + BN - 1) / BN, tr, tc)) return;
    // split reduction (dtc_linear_dgrad_split): grid.y = chunk g of the reduction index; chunk g multiplies columns
    // [g*split_n, (g+1)*split_n) of dZ with the matching rows of W and writes its own destination matrix
    long long dst_off = 0;
    int z0 = 0;                                      // first dZ column / W row of this chunk
    const int n_total = N;
    if (split_n > 0) {
        z0 = blockIdx.y * split_n;
        N = split_n;
        dst_off = (long long)blockIdx.y * split_dst;
    }
    const int m0 = tr * BM, c0 = col_skip + tc * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = wave * 32;
    const int half = lane >> 5, l31 = lane & 31;

    const int lrow = tid >> 2, lch = tid & 3;       // A loader: chunk lch of rows lrow (+64)
    u32 aoff[NA];
    int aslot[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = lrow + 64 * i, m = m0 + r;
        aoff[i] = m < M ? (u32)((long long)m * lddz + z0 + 4 * lch) * 4u : INVALID;
        aslot[i] = kslot(r, lch);
    }
    constexpr int C4 = BN / 4;                      // float4 columns per B row
    const bool bthread = tid < BK * C4;
    const int bk = tid / C4, bc = 4 * (tid % C4);   // B loader: W row n_0 + bk, columns c0 + bc .. +3
    const u32 boff = bthread ? (u32)((z0 + bk) * K + c0 + bc) * 4u : INVALID;
    const rsrc_t ares = make_rsrc_bytes(dZ, (long long)M * lddz * 4), bres = make_rsrc_bytes(W, (long long)n_total * K * 4);

    f32x4 ra[NA], rb;
    auto load_stage = [&](auto masked, int n_0) {
        const u32 sa = (u32)n_0 * 4u, sb = (u32)n_0 * (u32)K * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload4(ares, aoff[i], sa);
        if (decltype(masked)::value) {              // N tail: dZ columns past N are zeroed, W rows past N are not read
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = ktail(ra[i], n_0 + 4 * lch, N - 1);
            rb = bload4(bres, boff | oob_mask(n_0 + bk, N - 1), sb);
        } else {
            rb = bload4(bres, boff, sb);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) As[buf][aslot[i]] = ra[i];
        if (BN >= 64 || bthread) *reinterpret_cast<f32x4*>(&Bs[buf][bk][bc]) = rb;
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int frow = wm_off + l31, fsw = (frow >> 2) & 3;
    auto mfma_stage = [&](int buf) {
        f32x4 a[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) a[q] = As[buf][frow * 4 + ((2 * q + half) ^ fsw)];
        const float* bp = &Bs[buf][4 * half][l31];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float b[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = bp[(8 * q + t) * LDB + 32 * j];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][t], b[j], acc[j], 0, 0, 0);
            }
    };

    int buf = 0;
    auto step = [&](auto masked, int kt_next) {
        load_stage(masked, kt_next * BK);
        mfma_stage(buf);
        store_stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    };
    const int KT = (N + BK - 1) / BK, KF = N / BK;
    if constexpr (DEEP) {                                      // loads two stages ahead (see linear_fwd_kernel)
        f32x4 qa[2][NA], qb[2];
        int qtail[2];
        int lkt = 0;
        auto request = [&](f32x4 (&a)[NA], f32x4& b, int& tail) {
            const int n_0 = lkt * BK;
            const u32 sa = (u32)n_0 * 4u, sb = (u32)n_0 * (u32)K * 4u;
#pragma unroll
            for (int i = 0; i < NA; ++i) a[i] = bload4(ares, aoff[i], sa);
            b = bload4(bres, boff | oob_mask(n_0 + bk, N - 1), sb);
            tail = (N - 1) - (n_0 + 4 * lch);
            ++lkt;
        };
        auto commit = [&](int b, f32x4 (&a)[NA], f32x4& w, int tail) {
#pragma unroll
            for (int i = 0; i < NA; ++i) As[b][aslot[i]] = ktail(a[i], 0, tail);
            if (BN >= 64 || bthread) *reinterpret_cast<f32x4*>(&Bs[b][bk][bc]) = w;
        };
        request(qa[0], qb[0], qtail[0]);
        if (KT > 1) request(qa[1], qb[1], qtail[1]);
        commit(0, qa[0], qb[0], qtail[0]);
        __syncthreads();
        for (int st = 0; st < KT; st += 2) {
            if (st + 2 < KT) request(qa[0], qb[0], qtail[0]);
            mfma_stage(0);
            if (st + 1 < KT) commit(1, qa[1], qb[1], qtail[1]);
            __syncthreads();
            if (st + 1 >= KT) break;
            if (st + 3 < KT) request(qa[1], qb[1], qtail[1]);
            mfma_stage(1);
            if (st + 2 < KT) commit(0, qa[0], qb[0], qtail[0]);
            __syncthreads();
        }
    } else {
    if (KF >= 1) load_stage(Full{}, 0); else load_stage(Masked{}, 0);
    store_stage(0);
    __syncthreads();
    for (int kt = 1; kt < KF; ++kt) step(Full{}, kt);          // branch-free steady state (full stages)
    if (KT > KF && KT > 1) step(Masked{}, KT - 1);             // N tail
    mfma_stage(buf);
    }

    // ---- epilogue: activation derivative (through the saved post-activation output), segmented destination
    if (rmask) {
        // ReLU derivative from the forward kernel's sign record (dtc_linear_dgrad_mask): 2 bytes per lane and 32 x 32 tile
        // instead of the 64 bytes of saved activations; full tiles only (checked by the host).  The result has the same bits
        // as `y > 0 ? g : 0` on the saved output.
        unsigned bits[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bits[j] = rmask[((long long)((m0 + wm_off) >> 5) * 2 + half) * ldm + c0 + 32 * j + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = (bits[j] >> r) & 1u ? acc[j][r] : 0.f;
        act = DTC_ACT_NONE;
    }
    const rsrc_t xres = make_rsrc_bytes(Xs, (long long)M * ldxs * 4);
    if (m0 + BM <= M && c0 + BN <= K) {
        // wide path per 32-column sub-tile that lies inside ONE destination segment with 16-byte aligned rows: transpose
        // through the wave's LDS patch, then float4 loads of the saved activation / the accumulated destination and
        // float4 stores (4 instead of 16 memory instructions per operand and sub-tile)
        bool all_wide = true;
        int segj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cj = c0 + 32 * j;
            segj[j] = find_seg(dX, cj);
            const SegDev& sd = dX.s[segj[j]];
            all_wide = all_wide && ((wide_segs >> segj[j]) & 1) && cj + 32 <= sd.start + sd.width && ((cj - sd.start) & 3) == 0 &&
                       (act == DTC_ACT_NONE || ((wide_segs >> 4) & 1));
        }
        if (all_wide) {
            __syncthreads();                            // every wave is past its last operand read
            float* patch = reinterpret_cast<float*>(&As[0][0]) + wave * (32 * LDW);
            const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const SegDev sd = dX.s[segj[j]];
                const int cj = c0 + 32 * j + 4 * pc4;
                patch_put(patch, acc[j], half, l31);
                if (sd.ptr == nullptr) continue;
                float* dst = sd.ptr + dst_off + sd.col0 + (cj - sd.start) + (long long)(m0 + wm_off + prow) * sd.ld;
                const float* ys = Xs + (long long)(m0 + wm_off + prow) * ldxs + cj;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x4 v = patch_get(patch, prow + 8 * p, pc4);
                    if (act != DTC_ACT_NONE) {
                        const f32x4 y = *reinterpret_cast<const f32x4*>(ys + (long long)(8 * p) * ldxs);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_bwd(v[e], y[e], act);
                    }
                    f32x4* q = reinterpret_cast<f32x4*>(dst + (long long)(8 * p) * sd.ld);
                    if (sd.accumulate) {
                        const f32x4 o = *q;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = o[e] + v[e];
                    }
                    *q = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) am = abs_bits(v[e]) > am ? abs_bits(v[e]) : am;
                }
            }
            amax_publish(xamax, am);
            return;
        }
    }
    // scalar path: the saved activations come in through unconditional buffer loads (rows >= M fall past the
    // descriptor's size, columns >= K get the INVALID offset -> 0), so all 16 loads of a 32x32 tile are in
    // flight together instead of one exec-guarded load -> select -> store chain per element
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = c0 + 32 * j + l31;
        const bool cok = col < K;
        const SegDev sd = dX.s[find_seg(dX, cok ? col : 0)];
        const bool live = cok && sd.ptr != nullptr;
        float* dst = sd.ptr + dst_off + sd.col0 + (col - sd.start);
        const int row0 = m0 + wm_off + 4 * half;
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
            float v = acc[j][r];
            if (act != DTC_ACT_NONE) v = act_bwd(v, y[r], act);
            if (live && row < M) {
                float* q = dst + (long long)row * sd.ld;
                v = sd.accumulate ? (*q + v) : v;
                *q = v;
                am = abs_bits(v) > am ? abs_bits(v) : am;
            }
        }
    }
    amax_publish(xamax, am);
}

// fwd / dgrad: 128x64 tiles measured faster than 128x128 at every layer width of this model (twice the
// workgroups -> prologue / epilogue of one block overlap the MFMA phase of its neighbours, 4 waves/SIMD);
// when even those leave the chip short of workgroups (narrow layers, the ~1500-row GEMMs of one GRU time
// step) 128x32 tiles double the count again
// 128 x 128 tiles (2 x 2 waves of 64 x 64) for launches that still fill the chip with them: >= 700 tiles, i.e. three per CU
// (DTC_GEMM_128_BLOCKS; lab: 133.6 vs 129-130 TFLOP/s on 24576x512x512, 99 vs 121 on 24576x256x512 where only 384 tiles remain)
bool wide_tiles(int rows, int cols) {
    constexpr long long thr = 700;
    return cols >= 128 && dtc::ceil_div(rows, BM) * dtc::ceil_div(cols, 128) >= thr;
}

int pick_bn_rows(int rows, int cols) {
    if (cols <= 32) return 32;
    constexpr long long min_blocks = 320;                            // measured with threshold sweeps of bench.py (round 2)
    return dtc::ceil_div(rows, BM) * dtc::ceil_div(cols, 64) >= min_blocks ? 64 : 32;
}

int g_concurrency_hint = 0;          // dtc_set_concurrency_hint: another stream runs weight gradients next to this one

// launches of at most this many workgroups (about one per CU: rollout-sized batches, M = 4096) take the two-stages-ahead
// variants (DTC_GEMM_DEEP_BLOCKS; 0 = never).  Measured: 4096x512x752 61 -> 51 us, 4096x256x512 28 -> 24 us; the narrow layers
// of the update (24576 rows, 384-768 workgroups) do not profit (9.06 vs 9.17 ms per step with the threshold at 704): their
// time is the serial MFMA + epilogue chain of the one or two workgroups a CU gets, not exposed load latency.
bool deep_variant(int grid) {
    const char* e = getenv("DTC_GEMM_DEEP_BLOCKS");          // read per call: tests flip it between two launches
    return grid <= (e ? atoi(e) : 300);
}

// dgrad wide-store eligibility: bit s = destination segment s can be read/written with float4 accesses (16-byte aligned
// base + column origin, row stride a multiple of 4 floats, tile origin on a 4-column boundary); bit 4 = the same for
// the saved-activation matrix
int wide_mask(const SegMatDev& xd, const float* Xsaved, long long ldxs, int col_skip, long long split_stride) {
    if ((col_skip & 3) || (split_stride & 3)) return 0;
    int m = 0;
    for (int i = 0; i < xd.nseg; ++i) {
        const SegDev& sd = xd.s[i];
        if (sd.ptr == nullptr || (dtc::aligned16(sd.ptr) && (sd.ld & 3) == 0 && ((sd.col0 - sd.start) & 3) == 0)) m |= 1 << i;
    }
    if (Xsaved == nullptr || (dtc::aligned16(Xsaved) && (ldxs & 3) == 0)) m |= 1 << 4;
    return m;
}

}  // namespace

// NOTE on sizes: lane offsets are 32-bit byte offsets below 2 GiB, so every operand matrix must stay below
// 2^29 elements; gathered sources are bounded by the caller (row index * ld < 2^29), which holds for the
// rollouts of up to ~15000 envs x 24 steps per GPU this path is designed for (4096 x 24 x 1389 = 1.4e8).

#ifdef DTC_TRACE
extern "C" int dtc_debug_set_trace(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

// rows of 16-bit words per matrix of M rows: 2 per 32-row block, row tiles of 128 rows are always written whole
static inline long long mask_rows(int M) { return dtc::ceil_div(M, BM) * (BM / 32) * 2; }

extern "C" int64_t dtc_relu_mask_elems(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return mask_rows(M) * (int64_t)N;
}

static int linear_fwd_impl(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, int M, int N,
                           int K, int act, uint16_t* rmask, void* stream, uint32_t* y_amax = nullptr) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && ldy >= N, "bad shape M=%d N=%d K=%d ldy=%lld", M, N, K, (long long)ldy);
    DTC_REQUIRE(W && Y, "null pointer");
    DTC_REQUIRE(act >= 0 && act <= DTC_ACT_SIGMOID, "bad activation %d", act);
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * ldy <= MAX_ELEMS * 4, "matrix too large");
    SegMatDev xd;
    int rc = to_dev(X, xd, K, false, M);
    if (rc != DTC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int bn = wide_tiles(M, N) ? 128 : pick_bn_rows(M, N);
    const int grid = grid_for((int)dtc::ceil_div(M, BM), (int)dtc::ceil_div(N, bn));
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, K), 2.0 * M * (double)N * K, s, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    // dwordx4 stores of the output need 16-byte aligned rows (full tiles only; checked per block)
    const int wide = (ldy % 4 == 0 && dtc::aligned16(Y)) ? 1 : 0;
    if (rmask) {
        // the sign record is written by the wide epilogue of FULL tiles only: every tile of the launch must be one
        DTC_REQUIRE(act == DTC_ACT_RELU, "a sign record needs the ReLU activation (got %d)", act);
        DTC_REQUIRE(wide && M % BM == 0 && N % bn == 0, "sign record: M=%d must be a multiple of %d, N=%d of the %d-wide tile, Y 16-byte aligned rows", M, BM, N, bn);
    }
    unsigned short* rm = rmask;
    const int ldm = N;
    if (bn == 128) {
        hipLaunchKernelGGL((linear_fwd_kernel<128, false>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy, M, N, K, act, wide, MseEpi{}, rm, ldm, (amax_u32*)y_amax);
    } else if (deep_variant(grid)) {
        if (bn == 64) hipLaunchKernelGGL((linear_fwd_kernel<64, false, true>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy, M, N, K, act, wide, MseEpi{}, rm, ldm, (amax_u32*)y_amax);
        else hipLaunchKernelGGL((linear_fwd_kernel<32, false, true>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy, M, N, K, act, wide, MseEpi{}, rm, ldm, (amax_u32*)y_amax);
    } else if (bn == 64) hipLaunchKernelGGL((linear_fwd_kernel<64, false>), dim3(grid), dim3(256), occ_pad("FWD", 24576), s, xd, W, b, Y, (long long)ldy, M, N, K, act, wide, MseEpi{}, rm, ldm, (amax_u32*)y_amax);
    else hipLaunchKernelGGL((linear_fwd_kernel<32, false>), dim3(grid), dim3(256), 0, s, xd, W, b, Y, (long long)ldy, M, N, K, act, wide, MseEpi{}, rm, ldm, (amax_u32*)y_amax);
    return dtc::check_launch("linear_fwd");
}

extern "C" int dtc_linear_fwd(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, int M, int N,
                              int K, int act, void* stream) {
    return linear_fwd_impl(X, W, b, Y, ldy, M, N, K, act, nullptr, stream);
}

extern "C" int dtc_linear_fwd_mask(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                                   int M, int N, int K, void* stream) {
    DTC_REQUIRE(relu_mask != nullptr, "null sign record");
    return linear_fwd_impl(X, W, b, Y, ldy, M, N, K, (int)DTC_ACT_RELU, relu_mask, stream);
}

// dtc_linear_fwd / dtc_linear_fwd_mask (relu_mask may be NULL) that also adds the largest |Y| it writes to the amax record y_amax
// (two-term fp16 GEMM path, include/dtc_hip.h: DtcSeg.amax) -- for narrow layers whose result a split-path kernel consumes
extern "C" int dtc_linear_fwd_amax(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                                   uint32_t* y_amax, int M, int N, int K, int act, void* stream) {
    return linear_fwd_impl(X, W, b, Y, ldy, M, N, K, relu_mask ? (int)DTC_ACT_RELU : act, relu_mask, stream, y_amax);
}

extern "C" void dtc_set_concurrency_hint(int side_stream_active) { g_concurrency_hint = side_stream_active ? 1 : 0; }

extern "C" int dtc_linear_fwd_list(const DtcFwdLayer* layers, int count, int M, void* stream) {
    DTC_REQUIRE(layers != nullptr && count >= 1 && count <= 64, "layer list: count %d outside 1..64", count);
    for (int i = 0; i < count; ++i) {
        const DtcFwdLayer& L = layers[i];
        const int rc = dtc_linear_fwd(&L.X, L.W, L.b, L.Y, L.ldy, M, L.N, L.K, L.act, stream);
        if (rc != DTC_OK) return rc;
    }
    return DTC_OK;
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
    dtc::ProfScope prof(dtc::prof_shape_name("linear_fwd", M, N, K), 2.0 * M * (double)N * K, s,
                        4.0 * ((double)M * K + (double)N * K + 2.0 * M * N));      // X, W, the gathered target, dL/dY
    hipLaunchKernelGGL((linear_fwd_kernel<64, true>), dim3(grid), dim3(256), 0, s, xd, W, b, dY, (long long)lddy, M, N, K,
                       (int)DTC_ACT_NONE, 0, mse);
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

static int linear_dgrad_impl(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const float* Xsaved,
                             int64_t ldxs, int M, int N, int K, int act, const uint16_t* rmask, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && lddz >= N, "bad shape");
    DTC_REQUIRE(dZ && W, "null pointer");
    DTC_REQUIRE(act >= 0 && act <= DTC_ACT_SIGMOID, "bad activation %d", act);
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
    const int wide = wide_mask(xd, Xsaved, ldxs, col_skip, 0);
    double bytes = 4.0 * ((double)M * N + (double)N * K);                           // dZ, W
    for (int i = 0; i < xd.nseg; ++i)
        if (xd.s[i].ptr) bytes += 4.0 * M * xd.s[i].width * (xd.s[i].accumulate ? 2.0 : 1.0);   // dX written (+ read when accumulated)
    if (rmask) bytes += 0.125 * M * (double)K;                                       // one bit per output: the sign record
    else if (act != DTC_ACT_NONE) bytes += 4.0 * M * (double)K;                     // saved activations
    if (rmask) DTC_REQUIRE(M % BM == 0 && K % bn == 0 && col_skip == 0, "sign record: M=%d must be a multiple of %d and K=%d of the %d-wide tile", M, BM, K, bn);
    const unsigned short* rm = rmask;
    const int ldm = K;
    amax_u32* xam = (dX->nseg == 1) ? (amax_u32*)dX->seg[0].amax : nullptr;       // (DtcSeg.amax of a single destination: published)
    dtc::ProfScope prof(dtc::prof_shape_name("linear_dgrad", M, N, K), 2.0 * M * (double)N * (K - col_skip), s, bytes);
    if (deep_variant(grid)) {
        if (bn == 64) hipLaunchKernelGGL((linear_dgrad_kernel<64, true>), dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, W, xd, Xsaved, (long long)ldxs, M, N, K, act, 0, 0ll, col_skip, wide, rm, ldm, xam);
        else hipLaunchKernelGGL((linear_dgrad_kernel<32, true>), dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, W, xd, Xsaved, (long long)ldxs, M, N, K, act, 0, 0ll, col_skip, wide, rm, ldm, xam);
    } else if (bn == 64) hipLaunchKernelGGL(linear_dgrad_kernel<64>, dim3(grid), dim3(256), occ_pad("DGRAD", 25088, g_concurrency_hint ? 5 : 0), s, dZ, (long long)lddz, W, xd, Xsaved, (long long)ldxs, M, N, K, act, 0, 0ll, col_skip, wide, rm, ldm, xam);
    else hipLaunchKernelGGL(linear_dgrad_kernel<32>, dim3(grid), dim3(256), 0, s, dZ, (long long)lddz, W, xd, Xsaved, (long long)ldxs, M, N, K, act, 0, 0ll, col_skip, wide, rm, ldm, xam);
    return dtc::check_launch("linear_dgrad");
}

extern "C" int dtc_linear_dgrad(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const float* Xsaved,
                                int64_t ldxs, int M, int N, int K, int act, void* stream) {
    return linear_dgrad_impl(dZ, lddz, W, dX, Xsaved, ldxs, M, N, K, act, nullptr, stream);
}

extern "C" int dtc_linear_dgrad_mask(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const uint16_t* relu_mask,
                                     int M, int N, int K, void* stream) {
    DTC_REQUIRE(relu_mask != nullptr, "null sign record");
    DTC_REQUIRE(dX && dX->nseg == 1, "the ReLU derivative needs a single-segment destination");
    return linear_dgrad_impl(dZ, lddz, W, dX, nullptr, 0, M, N, K, (int)DTC_ACT_NONE, relu_mask, stream);
}

extern "C" int dtc_linear_dgrad_split(const float* dZ, int64_t lddz, const float* W, float* dX, int64_t lddx,
                                      int64_t split_stride, int M, int N, int K, int nsplit, void* stream) {
    DTC_REQUIRE(M > 0 && N > 0 && K > 0 && nsplit >= 1 && N % nsplit == 0 && lddz >= N && lddx >= K, "bad shape");
    DTC_REQUIRE(dZ && W && dX, "null pointer");
    DTC_REQUIRE((long long)N * K <= MAX_ELEMS && (long long)M * lddz <= MAX_ELEMS, "matrix too large");
    SegMatDev xd;
    xd.nseg = 1;
    xd.gathers = 0;
    xd.idx = nullptr;
    for (int i = 0; i < 4; ++i) xd.s[i] = SegDev{nullptr, 0, 0, 0x7fffffff, 0, 0, 0, 0};
    xd.s[0] = SegDev{dX, (long long)lddx, 0, 0, K, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    const int chunk = N / nsplit;
    // tile width from the work of ALL chunks (they run side by side in one launch)
    const int row_tiles = (int)dtc::ceil_div(M, BM);
    const int bn = (K <= 32 || (long long)row_tiles * dtc::ceil_div(K, 64) * nsplit < 320) ? 32 : 64;
    const dim3 grid((unsigned)grid_for(row_tiles, (int)dtc::ceil_div(K, bn)), (unsigned)nsplit);
    dtc::ProfScope prof(dtc::prof_shape_name("linear_dgrad", M, N, K), 2.0 * M * (double)N * K, s,
                        4.0 * ((double)M * N + (double)N * K + (double)nsplit * M * K));
    const int wide = wide_mask(xd, nullptr, 0, 0, split_stride);
    if (bn == 64) hipLaunchKernelGGL(linear_dgrad_kernel<64>, grid, dim3(256), 0, s, dZ, (long long)lddz, W, xd, nullptr, 0ll, M, N, K, (int)DTC_ACT_NONE, chunk, (long long)split_stride, 0, wide);
    else hipLaunchKernelGGL(linear_dgrad_kernel<32>, grid, dim3(256), 0, s, dZ, (long long)lddz, W, xd, nullptr, 0ll, M, N, K, (int)DTC_ACT_NONE, chunk, (long long)split_stride, 0, wide);
    return dtc::check_launch("linear_dgrad_split");
}
