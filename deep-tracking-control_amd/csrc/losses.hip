// Fused loss forward + gradient kernels for gfx950.
//
//  * dtc_vae_loss      rsl_rl/rsl_rl/algorithms/ppo.py:205-247  (recons / height / vel / KLD losses of
//                      the CE-net VAE step; torch runs ~25 elementwise + reduction ops and autograd
//                      re-reads every operand; here: one pass producing the 4 scalars and the three
//                      gradient tensors, with the gathered targets read through the mini-batch index)
//  * dtc_ppo_loss      ppo.py:288-327  (Normal log-prob, entropy, KL-adaptive learning rate, clipped
//                      surrogate + clipped value loss) -> 4 scalars, dL/dmean, dL/dvalue, dL/dstd, and
//                      the learning-rate decision taken ON DEVICE (the reference syncs the host here)
//  * dtc_gaussian_act  ppo.py:141-148 (rollout side: sample, log-prob, mu, sigma)
//
// All reductions: per-block partials in fp64 + a single-block finalize (deterministic order).
#include "amax.hpp"
#include "gemm_core.hpp"
#include "h2i_core.hpp"

namespace {

constexpr int OBS = 53, HGT = 693, PRIV = 1389, LD = 35, LAT = 16;
constexpr int MAX_BLK = 4096;
constexpr int MAX_ACT = 32;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;
}

// ---------------------------------------------------------------------------------- VAE losses
// block ranges: [0, nb_h) height, [nb_h, nb_h+nb_r) recons, rest latent rows
__global__ __launch_bounds__(256) void vae_loss_kernel(const float* __restrict__ recons, const float* __restrict__ hrecon,
                                                       const float* __restrict__ mulv, const float* __restrict__ next_obs,
                                                       const float* __restrict__ priv, const float* __restrict__ base_vel,
                                                       const long long* __restrict__ idx, float* __restrict__ d_recons,
                                                       float* __restrict__ d_hrecon, float* __restrict__ dmulv,
                                                       double* __restrict__ part, int B, int nb_h, int nb_r,
                                                       amax_u32* __restrict__ drec_amax, void* __restrict__ drec_img) {
    __shared__ double sh[4];
    __shared__ amax_u32 red_m[4];
    amax_u32 mrec = 0u;                                // largest |d_recons| this thread writes (amax record, two-term fp16 GEMM path)
    const int blk = blockIdx.x;
    double acc0 = 0.0, acc1 = 0.0;
    int slot0 = 3, slot1 = -1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blk < nb_h) {
        // one wavefront per batch row: the gather index is a scalar, no per-element division, 64 consecutive floats
        // per load instruction
        const float scale = 2.0f / ((float)HGT * (float)B);
        for (int b = blk * 4 + wv; b < B; b += nb_h * 4) {
            const float* src = priv + idx[b] * PRIV + (HGT + 3);
            const float* hr = hrecon + (long long)b * HGT;
            float* dh = d_hrecon + (long long)b * HGT;
            for (int c = lane; c < HGT; c += 64) {
                const float diff = hr[c] - src[c];
                acc0 += (double)diff * (double)diff;
                dh[c] = diff * scale;
            }
        }
        slot0 = 3;
    } else if (blk < nb_h + nb_r) {
        const float scale = 2.0f / ((float)OBS * (float)B);
        for (int b = (blk - nb_h) * 4 + wv; b < B; b += nb_r * 4) {
            float gv = 0.0f;
            if (lane < OBS) {
                const float diff = recons[(long long)b * OBS + lane] - next_obs[idx[b] * OBS + lane];
                acc0 += (double)diff * (double)diff;
                gv = diff * scale;
                d_recons[(long long)b * OBS + lane] = gv;
                mrec = abs_bits(gv) > mrec ? abs_bits(gv) : mrec;
            }
            if (drec_img) {
                // operand image of dL/d recons (53 columns) written here instead of by a pack launch: one row per wavefront
                u32 mb = finite_bits(gv);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const u32 o = (u32)__shfl_xor((int)mb, off, 64);
                    mb = o > mb ? o : mb;
                }
                const int ex = hi_exp(mb);
                hi_store_elem(drec_img, OBS, b, lane, gv, ex);            // lanes 53..63: the padding columns (zero)
                if (lane == 0) hi_store_row_exp(drec_img, B, OBS, b, ex);
            }
        }
        slot0 = 0;
    } else {
        const int b = (blk - nb_h - nb_r) * 256 + threadIdx.x;
        slot0 = 1;
        slot1 = 2;
        if (b < B) {
            const float* m = mulv + (long long)b * LD;
            float* g = dmulv + (long long)b * LD;
            const float invB = 1.0f / (float)B;
            const float vscale = 2.0f / (3.0f * (float)B);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float diff = m[j] - base_vel[idx[b] * 3 + j];
                acc0 += (double)diff * (double)diff;
                g[j] = diff * vscale;
            }
            float row = 0.f;
#pragma unroll
            for (int j = 0; j < LAT; ++j) {
                const float mu = m[3 + j], lv = m[19 + j];
                const float ex = expf(lv);
                row += ((1.0f + lv) - mu * mu) - ex;
                g[3 + j] = (4.0f * mu) * invB;                 // d(4*kld)/dmu
                g[19 + j] = (-2.0f * (1.0f - ex)) * invB;      // d(4*kld)/dlv = 4 * -0.5 * (1 - e^lv) / B
            }
            acc1 = (double)(-0.5f * row);
        }
    }
    amax_publish_block(drec_amax, mrec, red_m);
    acc0 = block_sum_d(acc0, sh);
    if (slot1 >= 0) acc1 = block_sum_d(acc1, sh);
    if (threadIdx.x == 0) {
        double* p = part + (long long)blk * 4;
        p[0] = p[1] = p[2] = p[3] = 0.0;
        p[slot0] = acc0;
        if (slot1 >= 0) p[slot1] = acc1;
    }
}

// hpart / nh: optional extra partials of the height term (one double per workgroup of the fused terrain-decoder layer)
__global__ __launch_bounds__(256) void vae_loss_finalize_kernel(const double* __restrict__ part, int nblk, int B,
                                                                float* __restrict__ losses,
                                                                const double* __restrict__ hpart, int nh) {
    __shared__ double sh[4];
    double a[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += blockDim.x)
        for (int k = 0; k < 4; ++k) a[k] += part[(long long)i * 4 + k];
    for (int i = threadIdx.x; i < nh; i += blockDim.x) a[3] += hpart[i];
    for (int k = 0; k < 4; ++k) a[k] = block_sum_d(a[k], sh);
    if (threadIdx.x == 0) {
        losses[0] = (float)(a[0] / ((double)OBS * B));   // recons
        losses[1] = (float)(a[1] / (3.0 * B));           // vel
        losses[2] = (float)(a[2] / (double)B);           // kld
        losses[3] = (float)(a[3] / ((double)HGT * B));   // height
    }
}

// ---------------------------------------------------------------------------------- PPO losses
// One row of ppo.py:288-327: Normal log-prob, KL to the rollout policy, clipped surrogate, (clipped) value loss.
// In: mean[A] / value of this row, r = row of the rollout tensors.  Out: dmean[A], dvalue, dstd terms (ds[A]), sums.
struct RowLoss {
    double s_sur, s_val, s_kl;
    float dvalue;
};
// AC > 0: the action count as a compile-time constant (== A): both loops unroll, so the row's 3 A gathered loads are all in flight at
// once instead of one loop iteration at a time, and mean_row / dmean_row / ds_row stay in registers.  Same operations in the same
// order; NOT bit-identical to the runtime-A form (AC = 0) all the same -- the compiler contracts other multiply-add pairs into FMAs
// in the unrolled code (last-bit differences, both forms inside the parity tests' tolerances): fused heads + finalize 46.0 -> 35.7 us
// per call at B = 24576 (tools/heads_hash.py), the bench step -0.25 ms (tools/jobs/r5_heads_unroll.sh).
template <int AC = 0>
__device__ __forceinline__ RowLoss ppo_row_loss(const float* mean_row, float v, const float* sstd, const float* __restrict__ actions,
                                                const float* __restrict__ old_logp, const float* __restrict__ old_mu,
                                                const float* __restrict__ old_sigma, const float* __restrict__ adv,
                                                const float* __restrict__ returns, const float* __restrict__ old_values,
                                                long long r, const DtcPpoCfg& cfg, float invB, int A, float* dmean_row, float* ds_row) {
    RowLoss o;
    float logp = 0.f, kl = 0.f;
    const int An = AC > 0 ? AC : A;
#pragma unroll AC > 0 ? AC : 1
    for (int j = 0; j < An; ++j) {
        const float sg = sstd[j], mu = mean_row[j], a = actions[r * A + j];
        const float d = a - mu;
        logp += (-(d * d) / (2.0f * (sg * sg)) - logf(sg)) - 0.918938533204672742f;
        const float so = old_sigma[r * A + j], mo = old_mu[r * A + j];
        const float dm = mo - mu;
        kl += (logf(sg / so + 1.e-5f) + (so * so + dm * dm) / (2.0f * (sg * sg))) - 0.5f;
    }
    o.s_kl = (double)kl;
    const float ratio = expf(logp - old_logp[r]);
    const float ad = adv[r];
    const float lo = 1.0f - cfg.clip_param, hi = 1.0f + cfg.clip_param;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float s1 = -ad * ratio, s2 = -ad * rc;
    o.s_sur = (double)fmaxf(s1, s2);
    const bool inrange = ratio >= lo && ratio <= hi;
    float w = 0.f;                                            // weight of the d(-A*ratio) path
    if (s1 > s2) w = 1.0f;
    else if (s1 == s2) w = inrange ? 1.0f : 0.5f;            // tie: half to each branch of max()
    const float dlogp_scale = w * (-ad) * ratio * invB;       // dL/dlogp of this row
    const float ret = returns[r];
    float dv;
    if (cfg.use_clipped_value_loss) {
        const float tv = old_values[r];
        const float dlt = v - tv;
        const float vc = tv + fminf(fmaxf(dlt, -cfg.clip_param), cfg.clip_param);
        const float e1 = v - ret, e2 = vc - ret;
        const float l1 = e1 * e1, l2 = e2 * e2;
        o.s_val = (double)fmaxf(l1, l2);
        const bool clip_pass = dlt >= -cfg.clip_param && dlt <= cfg.clip_param;
        const float g1 = 2.0f * e1, g2 = clip_pass ? 2.0f * e2 : 0.f;
        dv = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * (g1 + g2));
    } else {
        const float e1 = ret - v;
        o.s_val = (double)(e1 * e1);
        dv = -2.0f * e1;
    }
    o.dvalue = cfg.value_loss_coef * dv * invB;
#pragma unroll AC > 0 ? AC : 1
    for (int j = 0; j < An; ++j) {
        const float sg = sstd[j], mu = mean_row[j], a = actions[r * A + j];
        const float d = a - mu;
        dmean_row[j] = dlogp_scale * (d / (sg * sg));
        ds_row[j] = dlogp_scale * ((d * d) / (sg * sg * sg) - 1.0f / sg);
    }
    return o;
}

// all 3 + A block sums with ONE barrier: every wave reduces its values with shuffles, lane 0 parks them in LDS, then
// thread k adds the (at most 4) wave results of value k in wave order.  per-block partial layout: [0] surrogate sum,
// [1] value-loss sum, [2] kl sum, [3 .. 3+A) dstd sums
template <int AC = 0>
__device__ __forceinline__ void ppo_block_partials(double s_sur, double s_val, double s_kl, const float* ds_row, bool ok, int A,
                                                   double* __restrict__ part) {
    __shared__ double red[4][3 + MAX_ACT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double v0 = wave_sum_d(s_sur), v1 = wave_sum_d(s_val), v2 = wave_sum_d(s_kl);
    if (lane == 0) {
        red[wv][0] = v0;
        red[wv][1] = v1;
        red[wv][2] = v2;
    }
    const int An = AC > 0 ? AC : A;
#pragma unroll AC > 0 ? AC : 1
    for (int j = 0; j < An; ++j) {
        const double ds = wave_sum_d(ok ? (double)ds_row[j] : 0.0);
        if (lane == 0) red[wv][3 + j] = ds;
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 + A) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += red[w][threadIdx.x];
        part[(long long)blockIdx.x * (3 + MAX_ACT) + threadIdx.x] = t;
    }
}

// (AC: the action count as a template constant, see ppo_row_loss; 0 = run-time A)
template <int AC = 0>
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* __restrict__ mean, const float* __restrict__ stdp,
                                                       const float* __restrict__ value, const float* __restrict__ actions,
                                                       const float* __restrict__ old_logp, const float* __restrict__ old_mu,
                                                       const float* __restrict__ old_sigma, const float* __restrict__ adv,
                                                       const float* __restrict__ returns, const float* __restrict__ old_values,
                                                       const long long* __restrict__ idx, DtcPpoCfg cfg,
                                                       float* __restrict__ dmean, float* __restrict__ dvalue,
                                                       double* __restrict__ part, int B, int A) {
    __shared__ float sstd[MAX_ACT];
    if (threadIdx.x < A) sstd[threadIdx.x] = stdp[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.x * 256 + threadIdx.x;
    const bool ok = b < B;
    RowLoss o{0.0, 0.0, 0.0, 0.f};
    constexpr int NA = AC > 0 ? AC : MAX_ACT;
    const int An = AC > 0 ? AC : A;
    float mrow[NA], dm[NA], ds[NA];
    if (ok) {
        const long long r = idx ? idx[b] : (long long)b;
#pragma unroll AC > 0 ? AC : 1
        for (int j = 0; j < An; ++j) mrow[j] = mean[(long long)b * A + j];
        o = ppo_row_loss<AC>(mrow, value[b], sstd, actions, old_logp, old_mu, old_sigma, adv, returns, old_values, r, cfg,
                             1.0f / (float)B, A, dm, ds);
        dvalue[b] = o.dvalue;
#pragma unroll AC > 0 ? AC : 1
        for (int j = 0; j < An; ++j) dmean[(long long)b * A + j] = dm[j];
    } else if (AC > 0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) ds[j] = 0.f;
    }
    ppo_block_partials<AC>(o.s_sur, o.s_val, o.s_kl, ds, ok, A, part);
}

// ---------------------------------------------------------------------------------- heads + PPO losses, fused
// The last layers of the actor (H -> A) and the critic (H -> 1), the PPO losses, and the data gradients of those two
// layers in ONE kernel (round 2): five latency-bound launches on the critical path of every policy step (two 12 us head
// GEMMs, the loss, two head data gradients) become one.  256 threads = 64 rows x 4 threads; thread (row, p) owns the
// 16-byte chunks c = p + 4 i of its row of both hidden activations (kept in registers from the forward dot products to
// the backward products), the head weights sit in LDS (broadcast reads), the four partial dot products of a row are
// combined by two quad butterflies, every thread of the row then holds mean / value and evaluates the row's loss.
// dH = (d_out W) * act'(H) with the derivative through the saved post-activation value, as linear_dgrad_kernel does.
// operand images (csrc/h2i_core.hpp) of the four gradients the kernel writes, for the image-operand consumers of the trainer: the two
// [B, H] data gradients (H <= 128: one exponent per row) and dmean [B, A] / dvalue [B, 1] (the heads' own weight gradients); each may be NULL
struct HeadImgs {
    void *dHa, *dHc, *dmean, *dval;
};

template <int H, int TPR = 4, int AC = 0>
__global__ __launch_bounds__(256) void ppo_heads_loss_kernel(
    const float* __restrict__ Ha, long long ldha, const float* __restrict__ Hc, long long ldhc, const float* __restrict__ Wa,
    const float* __restrict__ ba, const float* __restrict__ Wc, const float* __restrict__ bc, int act_prev,
    const float* __restrict__ stdp, const float* __restrict__ actions, const float* __restrict__ old_logp,
    const float* __restrict__ old_mu, const float* __restrict__ old_sigma, const float* __restrict__ adv,
    const float* __restrict__ returns, const float* __restrict__ old_values, const long long* __restrict__ idx, DtcPpoCfg cfg,
    float* __restrict__ mean, float* __restrict__ value, float* __restrict__ dmean, float* __restrict__ dvalue,
    float* __restrict__ dHa, long long lddha, float* __restrict__ dHc, long long lddhc, double* __restrict__ part, int B, int A,
    amax_u32* __restrict__ dha_amax, amax_u32* __restrict__ dhc_amax, amax_u32* __restrict__ dmean_amax, amax_u32* __restrict__ dval_amax,
    const HeadImgs im) {
    // TPR threads per row (4: 64 rows per workgroup; 8: 32 rows per workgroup = twice the workgroups, half the serial work
    // per thread -- the kernel is a latency chain per row, not a bandwidth problem)
    constexpr int NC = H / (4 * TPR);                // chunks per thread
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float W[(MAX_ACT + 1) * H];      // rows 0..A-1: actor head, row A: critic head
    __shared__ float sstd[MAX_ACT], sb[MAX_ACT + 1];
    for (int e = threadIdx.x; e < (A + 1) * H; e += 256) W[e] = e < A * H ? Wa[e] : Wc[e - A * H];
    if ((int)threadIdx.x < A) {
        sstd[threadIdx.x] = stdp[threadIdx.x];
        sb[threadIdx.x] = ba ? ba[threadIdx.x] : 0.f;
    }
    if (threadIdx.x == 0) sb[A] = bc ? bc[0] : 0.f;
    __syncthreads();
    const int row = blockIdx.x * (256 / TPR) + (int)(threadIdx.x / TPR), p = threadIdx.x % TPR;
    const bool rok = row < B;
    const long long rr = rok ? row : 0;
    f4 ha[NC], hc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        ha[i] = *reinterpret_cast<const f4*>(Ha + rr * ldha + 4 * (p + TPR * i));
        hc[i] = *reinterpret_cast<const f4*>(Hc + rr * ldhc + 4 * (p + TPR * i));
    }
    // ---- forward: mean = Ha Wa^T + ba, value = Hc Wc^T + bc
    constexpr int NA = AC > 0 ? AC : MAX_ACT;        // (AC > 0: A == AC, checked by the host; every loop over the actions unrolls)
    const int An = AC > 0 ? AC : A;
    float mrow[NA], v = 0.f;
#pragma unroll AC > 0 ? AC : 1
    for (int j = 0; j < An; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const f4 w = *reinterpret_cast<const f4*>(&W[j * H + 4 * (p + TPR * i)]);
            acc += ha[i].x * w.x + ha[i].y * w.y + ha[i].z * w.z + ha[i].w * w.w;
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (TPR == 8) acc += __shfl_xor(acc, 4, 64);
        mrow[j] = acc + sb[j];
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const f4 w = *reinterpret_cast<const f4*>(&W[A * H + 4 * (p + TPR * i)]);
        v += hc[i].x * w.x + hc[i].y * w.y + hc[i].z * w.z + hc[i].w * w.w;
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if (TPR == 8) v += __shfl_xor(v, 4, 64);
    v += sb[A];
    // ---- the row's loss (every thread of the row evaluates it; thread p = 0 stores and contributes the sums)
    RowLoss o{0.0, 0.0, 0.0, 0.f};
    float dm[NA], ds[NA];
    const bool ok = rok && p == 0;
    if (rok) {
        const long long r = idx ? idx[row] : (long long)row;
        o = ppo_row_loss<AC>(mrow, v, sstd, actions, old_logp, old_mu, old_sigma, adv, returns, old_values, r, cfg, 1.0f / (float)B, A,
                             dm, ds);
    } else if (AC > 0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) dm[j] = ds[j] = 0.f;
    }
    amax_u32 mdm = 0u, mdv = 0u;           // largest |dmean| / |dvalue| this thread writes
    if (ok) {
        value[row] = v;
        dvalue[row] = o.dvalue;
        mdv = abs_bits(o.dvalue);
#pragma unroll AC > 0 ? AC : 1
        for (int j = 0; j < An; ++j) {
            mean[(long long)row * A + j] = mrow[j];
            dmean[(long long)row * A + j] = dm[j];
            mdm = abs_bits(dm[j]) > mdm ? abs_bits(dm[j]) : mdm;
        }
        if (im.dmean) {
            u32 mb = 0u;
#pragma unroll AC > 0 ? AC : 1
            for (int j = 0; j < An; ++j) mb = finite_bits(dm[j]) > mb ? finite_bits(dm[j]) : mb;
            const int ex = hi_exp(mb);
#pragma unroll AC > 0 ? AC : 1
            for (int j = 0; j < An; ++j) hi_store_elem(im.dmean, A, row, j, dm[j], ex);
            hi_store_row_exp(im.dmean, B, A, row, ex);
        }
        if (im.dval) {
            const int ex = hi_exp(finite_bits(o.dvalue));
            hi_store_elem(im.dval, 1, row, 0, o.dvalue, ex);
            hi_store_row_exp(im.dval, B, 1, row, ex);
        }
    }
    // ---- backward through the two heads: dHa = (dmean Wa) * act'(Ha), dHc = (dvalue Wc) * act'(Hc)
    amax_u32 ma = 0u, mc = 0u;             // largest |dHa| / |dHc| this thread writes (amax records of the two-term fp16 GEMM path)
    f4 GA[NC], GC[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) GA[i] = GC[i] = f4{0.f, 0.f, 0.f, 0.f};
    if (rok) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            f4 ga = {0.f, 0.f, 0.f, 0.f};
#pragma unroll AC > 0 ? AC : 1
            for (int j = 0; j < An; ++j) {
                const f4 w = *reinterpret_cast<const f4*>(&W[j * H + 4 * (p + TPR * i)]);
                ga += dm[j] * w;
            }
            const f4 wc = *reinterpret_cast<const f4*>(&W[A * H + 4 * (p + TPR * i)]);
            f4 gc = o.dvalue * wc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ga[e] = act_bwd(ga[e], ha[i][e], act_prev);
                gc[e] = act_bwd(gc[e], hc[i][e], act_prev);
                ma = abs_bits(ga[e]) > ma ? abs_bits(ga[e]) : ma;
                mc = abs_bits(gc[e]) > mc ? abs_bits(gc[e]) : mc;
            }
            if (dHa) *reinterpret_cast<f4*>(dHa + rr * lddha + 4 * (p + TPR * i)) = ga;      // (NULL: the image is the only copy)
            if (dHc) *reinterpret_cast<f4*>(dHc + rr * lddhc + 4 * (p + TPR * i)) = gc;
            GA[i] = ga;
            GC[i] = gc;
        }
    }
    if (im.dHa || im.dHc) {                    // (uniform) the row's threads agree on the row's exponents, then store their own chunks
        u32 xa = 0u, xc = 0u;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xa = finite_bits(GA[i][e]) > xa ? finite_bits(GA[i][e]) : xa;
                xc = finite_bits(GC[i][e]) > xc ? finite_bits(GC[i][e]) : xc;
            }
#pragma unroll
        for (int off = 1; off < TPR; off <<= 1) {
            const u32 oa = (u32)__shfl_xor((int)xa, off, 64), oc = (u32)__shfl_xor((int)xc, off, 64);
            xa = oa > xa ? oa : xa;
            xc = oc > xc ? oc : xc;
        }
        const int ea = hi_exp(xa), ec = hi_exp(xc);
        if (rok) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (im.dHa) hi_store4(im.dHa, H, row, 4 * (p + TPR * i), GA[i], ea);
                if (im.dHc) hi_store4(im.dHc, H, row, 4 * (p + TPR * i), GC[i], ec);
            }
            if (p == 0) {
                if (im.dHa) hi_store_row_exp(im.dHa, B, H, row, ea);
                if (im.dHc) hi_store_row_exp(im.dHc, B, H, row, ec);
            }
        }
    }
    __shared__ amax_u32 red_a[4], red_c[4], red_m[4], red_v[4];
    amax_publish_block(dha_amax, ma, red_a);
    amax_publish_block(dhc_amax, mc, red_c);
    amax_publish_block(dmean_amax, mdm, red_m);
    amax_publish_block(dval_amax, mdv, red_v);
    ppo_block_partials<AC>(ok ? o.s_sur : 0.0, ok ? o.s_val : 0.0, ok ? o.s_kl : 0.0, ds, ok, A, part);
}

// wave w owns the values k = w, w + 4, ...: lanes add the per-block partials in a fixed stride order, one shuffle
// reduction per value, no block barrier
// One block of 1024 threads: wave w owns value w (3 loss sums + A std-gradient sums <= 16 waves), so all values are reduced side by side
// -- the launch sits on the critical path between the loss and the backward pass, and with four waves taking four values each in turn,
// then one thread evaluating A logarithms one after the other, it took 16.6 us (rocprofv3) for ~100 KB of partials.  Per value the
// summation order is unchanged (lanes add the per-block partials in a fixed stride order, one shuffle reduction), and the entropy is
// still added up by one thread in index order: results are bit-identical to the four-wave form.
__global__ __launch_bounds__(1024) void ppo_loss_finalize_kernel(const double* __restrict__ part, int nblk, int B, int A,
                                                                 const float* __restrict__ stdp, DtcPpoCfg cfg,
                                                                 float* __restrict__ dstd, float* __restrict__ losses,
                                                                 double* __restrict__ lr) {
    const int stride = 3 + MAX_ACT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ float lg[MAX_ACT];
    if ((int)threadIdx.x < A) lg[threadIdx.x] = logf(stdp[threadIdx.x]);
    for (int k = wv; k < 3 + A; k += 16) {
        double a = 0.0;
        for (int i = lane; i < nblk; i += 64) a += part[(long long)i * stride + k];
        a = wave_sum_d(a);
        if (lane == 0) {
            if (k == 0) losses[0] = (float)(a / B);
            else if (k == 1) losses[1] = (float)(a / B);
            else if (k == 2) {
                const float klm = (float)(a / B);
                losses[3] = klm;
                if (cfg.kl_mirror) *cfg.kl_mirror = klm;       // data parallel: the gradient header's KL slot, written here (no copy launch)
                if (cfg.adaptive_schedule && lr) {
                    double cur = *lr;
                    if (klm > cfg.desired_kl * 2.0f) cur = fmax(1e-5, cur / 1.5);
                    else if (klm < cfg.desired_kl / 2.0f && klm > 0.0f) cur = fmin(1e-2, cur * 1.5);
                    *lr = cur;
                }
            } else {
                const int j = k - 3;
                dstd[j] = (float)a - cfg.entropy_coef / stdp[j];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float h = 0.f;
        for (int j = 0; j < A; ++j) h += 1.418938533204672742f + lg[j];   // 0.5 + 0.5*ln(2*pi) + ln(sigma)
        losses[2] = h;
    }
}

__global__ void lr_adapt_kernel(float* __restrict__ kl_mean, double* __restrict__ lr, float desired_kl, float* __restrict__ kl_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // The slot is CONSUMED: it is left holding a NaN with a payload of its own (0x7fc0dead), so a step that forgets to deposit its KL
        // fails loudly (lr = NaN) instead of re-using the previous value.  A KL that is itself NaN (any other payload: a diverged
        // policy) leaves the learning rate unchanged -- both comparisons of ppo.py:301-307 are false for NaN.  (Data parallel: the slot
        // rides in the header of a gradient bucket; _kl_to_header re-deposits it before the one exchange whose result is read.)
        constexpr unsigned SENTINEL = 0x7fc0deadu;
        const float klm = *kl_mean;
        if (kl_out) *kl_out = klm;
        const bool missing = __float_as_uint(klm) == SENTINEL;
        *kl_mean = __uint_as_float(SENTINEL);
        double cur = *lr;
        if (missing) cur = (double)__builtin_nanf("");  // nothing was deposited: fail loudly
        if (klm > desired_kl * 2.0f) cur = fmax(1e-5, cur / 1.5);
        else if (klm < desired_kl / 2.0f && klm > 0.0f) cur = fmin(1e-2, cur * 1.5);
        *lr = cur;
    }
}

__global__ __launch_bounds__(256) void gaussian_act_kernel(const float* __restrict__ mean, const float* __restrict__ stdp,
                                                           const float* __restrict__ noise, float* __restrict__ actions,
                                                           float* __restrict__ logp, float* __restrict__ mu_out,
                                                           float* __restrict__ sigma_out, int B, int A) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float lp = 0.f;
    for (int j = 0; j < A; ++j) {
        const long long o = (long long)b * A + j;
        const float sg = stdp[j], mu = mean[o];
        const float a = noise[o] * sg + mu;
        const float d = a - mu;
        lp += (-(d * d) / (2.0f * (sg * sg)) - logf(sg)) - 0.918938533204672742f;
        actions[o] = a;
        if (mu_out) mu_out[o] = mu;
        if (sigma_out) sigma_out[o] = sg;
    }
    logp[b] = lp;
}

}  // namespace

// The loss kernels with the action count as a template constant for the reference's shapes (A = 12): fused heads + finalize 46.0 -> 35.7 us per
// call, -0.25 ms per bench step.  DTC_HEADS_UNROLL=0: the run-time-A kernels.  (Round 5 shipped them off: the 2-rank data-parallel test differed
// from run to run with them.  Round 6: the kernel is bit-reproducible on fixed inputs -- 3e5 launches beside other processes' work, poisoned LDS /
// VGPRs / torch.empty buffers, 50 one-rank updates; the differences need two lock-stepped processes on ONE device with high-priority queues, which
// the trainers no longer create in that configuration: DESIGN.md §5.)
static bool heads_unrolled() {
    static const bool on = !(getenv("DTC_HEADS_UNROLL") && atoi(getenv("DTC_HEADS_UNROLL")) == 0);
    return on;
}

extern "C" int64_t dtc_loss_workspace(int B) {
    (void)B;
    return (int64_t)sizeof(double) * MAX_BLK * (3 + MAX_ACT);
}

extern "C" int dtc_vae_loss(const float* recons, const float* hrecon, const float* mulv, const float* next_obs,
                            const float* priv, const float* base_vel, const int64_t* idx, float* d_recons,
                            float* d_hrecon, float* dmulv, float* losses, void* workspace, int B, uint32_t* drec_amax, void* stream) {
    DTC_REQUIRE(B > 0, "bad batch %d", B);
    DTC_REQUIRE(recons && hrecon && mulv && next_obs && priv && base_vel && idx, "null input");
    DTC_REQUIRE(d_recons && d_hrecon && dmulv && losses && workspace, "null output");
    hipStream_t s = (hipStream_t)stream;
    int nb_h = (int)dtc::ceil_div(B, 4 * 3);          // 4 rows per block pass, ~3 passes
    int nb_r = (int)dtc::ceil_div(B, 4 * 16);
    const int nb_l = (int)dtc::ceil_div(B, 256);
    if (nb_h > 2048) nb_h = 2048;
    if (nb_r > 512) nb_r = 512;
    const int nblk = nb_h + nb_r + nb_l;
    DTC_REQUIRE(nblk <= MAX_BLK, "batch too large for the loss workspace");
    double* part = (double*)workspace;
    dtc::ProfScope prof("vae_loss", (double)B * (HGT * 12.0 + OBS * 12.0 + LD * 8.0), s);
    hipLaunchKernelGGL(vae_loss_kernel, dim3(nblk), dim3(256), 0, s, recons, hrecon, mulv, next_obs, priv, base_vel,
                       (const long long*)idx, d_recons, d_hrecon, dmulv, part, B, nb_h, nb_r, (amax_u32*)drec_amax, (void*)nullptr);
    hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(256), 0, s, part, nblk, B, losses, (const double*)nullptr, 0);
    return dtc::check_launch("vae_loss");
}

extern "C" int dtc_vae_loss_fused(const float* recons, const float* mulv, const float* next_obs, const float* base_vel,
                                  const int64_t* idx, float* d_recons, float* dmulv, const double* height_sq_part,
                                  int n_height_part, float* losses, void* workspace, int B, uint32_t* drec_amax, void* stream) {
    return dtc_vae_loss_fused_img(recons, mulv, next_obs, base_vel, idx, d_recons, dmulv, height_sq_part, n_height_part, losses, workspace, B,
                                  drec_amax, nullptr, stream);
}

extern "C" int dtc_vae_loss_fused_img(const float* recons, const float* mulv, const float* next_obs, const float* base_vel,
                                      const int64_t* idx, float* d_recons, float* dmulv, const double* height_sq_part,
                                      int n_height_part, float* losses, void* workspace, int B, uint32_t* drec_amax, void* drec_img,
                                      void* stream) {
    DTC_REQUIRE(dtc::aligned16(drec_img), "unaligned image");
    DTC_REQUIRE(B > 0 && n_height_part >= 0, "bad batch %d", B);
    DTC_REQUIRE(recons && mulv && next_obs && base_vel && idx, "null input");
    DTC_REQUIRE(d_recons && dmulv && losses && workspace && (height_sq_part || n_height_part == 0), "null output");
    hipStream_t s = (hipStream_t)stream;
    int nb_r = (int)dtc::ceil_div(B, 4 * 16);
    const int nb_l = (int)dtc::ceil_div(B, 256);
    if (nb_r > 512) nb_r = 512;
    const int nblk = nb_r + nb_l;
    DTC_REQUIRE(nblk <= MAX_BLK, "batch too large for the loss workspace");
    double* part = (double*)workspace;
    dtc::ProfScope prof("vae_loss", (double)B * (OBS * 12.0 + LD * 8.0), s);
    hipLaunchKernelGGL(vae_loss_kernel, dim3(nblk), dim3(256), 0, s, recons, (const float*)nullptr, mulv, next_obs,
                       (const float*)nullptr, base_vel, (const long long*)idx, d_recons, (float*)nullptr, dmulv, part, B, 0, nb_r,
                       (amax_u32*)drec_amax, drec_img);
    hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(256), 0, s, part, nblk, B, losses, height_sq_part, n_height_part);
    return dtc::check_launch("vae_loss_fused");
}

extern "C" int dtc_ppo_loss(const float* mean, const float* std, const float* value, const float* actions,
                            const float* old_logp, const float* old_mu, const float* old_sigma, const float* advantages,
                            const float* returns, const float* old_values, const int64_t* idx, const DtcPpoCfg* cfg,
                            float* dmean, float* dvalue, float* dstd, float* losses, double* lr, void* workspace, int B,
                            int num_actions, void* stream) {
    DTC_REQUIRE(B > 0 && num_actions > 0 && num_actions <= MAX_ACT, "bad shape B=%d A=%d", B, num_actions);
    DTC_REQUIRE(mean && std && value && actions && old_logp && old_mu && old_sigma && advantages && returns && old_values,
                "null input");
    DTC_REQUIRE(cfg && dmean && dvalue && dstd && losses && workspace, "null output");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (int)dtc::ceil_div(B, 256);
    DTC_REQUIRE(nblk <= MAX_BLK, "batch too large for the loss workspace");
    double* part = (double*)workspace;
    dtc::ProfScope prof("ppo_loss", (double)B * num_actions * 24.0, s);
    if (num_actions == 12 && heads_unrolled())
        hipLaunchKernelGGL(ppo_loss_kernel<12>, dim3(nblk), dim3(256), 0, s, mean, std, value, actions, old_logp, old_mu, old_sigma,
                           advantages, returns, old_values, (const long long*)idx, *cfg, dmean, dvalue, part, B, num_actions);
    else
        hipLaunchKernelGGL(ppo_loss_kernel<0>, dim3(nblk), dim3(256), 0, s, mean, std, value, actions, old_logp, old_mu,
                           old_sigma, advantages, returns, old_values, (const long long*)idx, *cfg, dmean, dvalue, part, B,
                           num_actions);
    hipLaunchKernelGGL(ppo_loss_finalize_kernel, dim3(1), dim3(1024), 0, s, part, nblk, B, num_actions, std, *cfg, dstd,
                       losses, lr);
    return dtc::check_launch("ppo_loss");
}

extern "C" int dtc_ppo_heads_loss(const float* Ha, int64_t ldha, const float* Hc, int64_t ldhc, int H, const float* Wa,
                                  const float* ba, const float* Wc, const float* bc, int act_prev, const float* std,
                                  const float* actions, const float* old_logp, const float* old_mu, const float* old_sigma,
                                  const float* advantages, const float* returns, const float* old_values, const int64_t* idx,
                                  const DtcPpoCfg* cfg, float* mean, float* value, float* dmean, float* dvalue, float* dHa,
                                  int64_t lddha, float* dHc, int64_t lddhc, float* dstd, float* losses, double* lr,
                                  void* workspace, int B, int num_actions, uint32_t* dha_amax, uint32_t* dhc_amax, uint32_t* dmean_amax,
                                  uint32_t* dval_amax, void* stream) {
    return dtc_ppo_heads_loss_img(Ha, ldha, Hc, ldhc, H, Wa, ba, Wc, bc, act_prev, std, actions, old_logp, old_mu, old_sigma, advantages, returns,
                                  old_values, idx, cfg, mean, value, dmean, dvalue, dHa, lddha, dHc, lddhc, dstd, losses, lr, workspace, B,
                                  num_actions, dha_amax, dhc_amax, dmean_amax, dval_amax, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int dtc_ppo_heads_loss_img(const float* Ha, int64_t ldha, const float* Hc, int64_t ldhc, int H, const float* Wa,
                                      const float* ba, const float* Wc, const float* bc, int act_prev, const float* std,
                                      const float* actions, const float* old_logp, const float* old_mu, const float* old_sigma,
                                      const float* advantages, const float* returns, const float* old_values, const int64_t* idx,
                                      const DtcPpoCfg* cfg, float* mean, float* value, float* dmean, float* dvalue, float* dHa,
                                      int64_t lddha, float* dHc, int64_t lddhc, float* dstd, float* losses, double* lr,
                                      void* workspace, int B, int num_actions, uint32_t* dha_amax, uint32_t* dhc_amax,
                                      uint32_t* dmean_amax, uint32_t* dval_amax, void* dHa_img, void* dHc_img, void* dmean_img,
                                      void* dval_img, void* stream) {
    DTC_REQUIRE(B > 0 && num_actions > 0 && num_actions <= MAX_ACT, "bad shape B=%d A=%d", B, num_actions);
    DTC_REQUIRE((!dHa_img && !dHc_img) || H <= 128, "operand images of the hidden gradients need H <= 128 (one exponent per row), H = %d", H);
    DTC_REQUIRE(dtc::aligned16(dHa_img) && dtc::aligned16(dHc_img) && dtc::aligned16(dmean_img) && dtc::aligned16(dval_img), "unaligned image");
    const HeadImgs him{dHa_img, dHc_img, dmean_img, dval_img};
    DTC_REQUIRE(H == 64 || H == 128 || H == 256, "hidden width %d unsupported by the fused heads (64, 128, 256)", H);
    DTC_REQUIRE(Ha && Hc && Wa && Wc && std && actions && old_logp && old_mu && old_sigma && advantages && returns && old_values,
                "null input");
    DTC_REQUIRE(cfg && mean && value && dmean && dvalue && dstd && losses && workspace, "null output");
    // dHa / dHc: fp32 [B, H], or NULL where the gradient is wanted as its operand image only (no fp32 copy is written)
    DTC_REQUIRE((dHa || dHa_img) && (dHc || dHc_img), "dHa / dHc: an fp32 destination or an image is required");
    if (!dHa) lddha = H;
    if (!dHc) lddhc = H;
    DTC_REQUIRE(ldha >= H && ldhc >= H && lddha >= H && lddhc >= H && ldha % 4 == 0 && ldhc % 4 == 0 && lddha % 4 == 0 &&
                    lddhc % 4 == 0 && dtc::aligned16(Ha) && dtc::aligned16(Hc) && dtc::aligned16(dHa) && dtc::aligned16(dHc),
                "hidden activations must be 16-byte aligned with row strides that are multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    // threads per row: 4 (64 rows per workgroup) measured faster than 8 (32 rows: every thread of a row evaluates the row's
    // loss, so twice the threads per row doubles the transcendental work): 1.08 vs 1.51 ms per step (round 3); DTC_HEADS_TPR=8
    const int tpr = 4;
    const int nblk = (int)dtc::ceil_div(B, 256 / tpr);
    DTC_REQUIRE(nblk <= MAX_BLK, "batch too large for the loss workspace");
    double* part = (double*)workspace;
    dtc::ProfScope prof("ppo_heads_loss", (double)B * (4.0 * H * 4 + num_actions * 32.0), s);
#define DTC_HL_ARGS Ha, (long long)ldha, Hc, (long long)ldhc, Wa, ba, Wc, bc, act_prev, std, actions, old_logp, old_mu, old_sigma, \
                    advantages, returns, old_values, (const long long*)idx, *cfg, mean, value, dmean, dvalue, dHa, (long long)lddha, \
                    dHc, (long long)lddhc, part, B, num_actions, (amax_u32*)dha_amax, (amax_u32*)dhc_amax, (amax_u32*)dmean_amax, \
                    (amax_u32*)dval_amax, him
    if (tpr == 8) {
        if (H == 64) hipLaunchKernelGGL((ppo_heads_loss_kernel<64, 8>), dim3(nblk), dim3(256), 0, s, DTC_HL_ARGS);
        else if (H == 128) hipLaunchKernelGGL((ppo_heads_loss_kernel<128, 8>), dim3(nblk), dim3(256), 0, s, DTC_HL_ARGS);
        else hipLaunchKernelGGL((ppo_heads_loss_kernel<256, 8>), dim3(nblk), dim3(256), 0, s, DTC_HL_ARGS);
    } else if (H == 64) hipLaunchKernelGGL(ppo_heads_loss_kernel<64>, dim3(nblk), dim3(256), 0, s, DTC_HL_ARGS);
    else if (H == 128 && num_actions == 12 && heads_unrolled())      // the reference's shapes (128 -> 12 / 1): action loops unrolled
        hipLaunchKernelGGL((ppo_heads_loss_kernel<128, 4, 12>), dim3(nblk), dim3(256), 0, s, DTC_HL_ARGS);
    else if (H == 128) hipLaunchKernelGGL(ppo_heads_loss_kernel<128>, dim3(nblk), dim3(256), 0, s, DTC_HL_ARGS);
    else hipLaunchKernelGGL(ppo_heads_loss_kernel<256>, dim3(nblk), dim3(256), 0, s, DTC_HL_ARGS);
#undef DTC_HL_ARGS
    hipLaunchKernelGGL(ppo_loss_finalize_kernel, dim3(1), dim3(1024), 0, s, part, nblk, B, num_actions, std, *cfg, dstd, losses, lr);
    return dtc::check_launch("ppo_heads_loss");
}

// actor_critic_decoder.py:404-407: 1 - tanh(std(r) / mean(r)), unbiased std (torch.std).  One block: the buffer is one reward
// per env (4096-32768 values); two passes in double so that a buffer of near-equal rewards keeps its small variance.
__global__ __launch_bounds__(1024) void bootstrap_prob_kernel(const float* __restrict__ r, int64_t n, float* __restrict__ out) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    auto block_sum = [&](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        double t = 0;
        for (int i = 0; i < 16; ++i) t += red[i];
        return t;
    };
    double a = 0;
    for (int64_t i = tid; i < n; i += 1024) a += (double)r[i];
    const double mean = block_sum(a) / (double)n;
    double q = 0;
    for (int64_t i = tid; i < n; i += 1024) { const double d = (double)r[i] - mean; q += d * d; }
    const double var = block_sum(q) / (double)(n - 1);        // n == 1: 0 / 0 = NaN, as torch.std
    if (tid == 0) {
        // the reference's arithmetic from here on is fp32 tensors: std and mean rounded to float before the division
        const float cv = (float)sqrt(var) / (float)mean;
        out[0] = 1.0f - tanhf(cv);
    }
}

extern "C" int dtc_bootstrap_probability(const float* rewards, int64_t n, float* out, void* stream) {
    DTC_REQUIRE(n > 0, "empty reward buffer");
    DTC_REQUIRE(rewards && out, "null pointer");
    hipLaunchKernelGGL(bootstrap_prob_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rewards, n, out);
    return dtc::check_launch("bootstrap_probability");
}

extern "C" int dtc_gaussian_act(const float* mean, const float* std, const float* noise, float* actions, float* logp,
                                float* mu_out, float* sigma_out, int B, int num_actions, void* stream) {
    DTC_REQUIRE(B > 0 && num_actions > 0, "bad shape");
    DTC_REQUIRE(mean && std && noise && actions && logp, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gaussian_act_kernel, dim3((unsigned)dtc::ceil_div(B, 256)), dim3(256), 0, s, mean, std, noise,
                       actions, logp, mu_out, sigma_out, B, num_actions);
    return dtc::check_launch("gaussian_act");
}

extern "C" int dtc_lr_adapt(float* kl_mean, double* lr, float desired_kl, float* kl_out, void* stream) {
    DTC_REQUIRE(kl_mean && lr, "null pointer");
    hipLaunchKernelGGL(lr_adapt_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kl_mean, lr, desired_kl, kl_out);
    return dtc::check_launch("lr_adapt");
}
