// CE-net latent block for gfx950: outlier -> median replacement + reparameterisation, fwd and bwd.
//
// Reference: rsl_rl/rsl_rl/modules/actor_critic_decoder.py:274-302
//     mean = lv.mean(); std = lv.std(); out = (lv < mean-2std) | (lv > mean+2std)
//     lv[out] = lv[~out].median()            (lower median, batch-global)
//     z = eps * exp(0.5*lv) + mu[:, 3:]
// The torch version is ~10 passes over lv plus a sort-based median on ~4e5 values.  Here:
//   stats (sum, sum^2 in fp64)  ->  3-level MSB radix select (11+11+10 bits) over the
//   non-outliers with LDS histograms  ->  one apply pass that also writes z and the mask.
// Every pass reads the 1.5 MB lv column block (L2 resident): 5 short launches (the first one also zeroes the scratch header).  (A single persistent launch
// with device-scope counter barriers between the phases was built and measured in round 2: 56-92 us against 51 us --
// with a few dozen workgroups the LDS histogram atomics on two or three hot bins serialise, with 256 workgroups the
// barriers cost as much as the launch boundaries they replace; see DESIGN.md "measured and rejected".)
// Backward = ONE kernel: the last workgroup to finish (arrival ticket) adds the replaced entries' gradient to the median
// element; what it reads from other workgroups travels write-through (agent-scope relaxed stores / loads), no
// cache-maintenance fence (an agent-scope release would write back the whole XCD L2).
// mulv is the [B,35] output of the fused (latent_mu | latent_var) head: cols 0..18 mu, 19..34 lv.
#include "amax.hpp"
#include <stddef.h>
#include <stdlib.h>

#include "common.hpp"
#include "h2i_core.hpp"

// The backward kernel's cross-workgroup hand-off (partial sums, the median element, the arrival ticket) uses RELAXED
// agent-scope atomics + `s_waitcnt vmcnt(0)` instead of a release / acquire pair.  That is an architecture contract, not
// the HIP memory model: on gfx9 (gfx90a / gfx942 / gfx950) an agent-scope atomic store or RMW is performed AT the L2 (sc1
// write-through, never held in the per-CU vector cache), agent-scope atomic loads bypass the vector cache, and vmcnt counts
// stores, so `store; s_waitcnt vmcnt(0); ticket RMW` orders the store before the ticket at the L2 every workgroup of one
// XCD-spanning launch reads through.  A release fence would be correct everywhere but writes back the whole XCD L2
// (dirty activations of unrelated kernels: measured 22 -> 14 us without it).  Any other target must use the fences:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "csrc/latent.hip: the relaxed write-through hand-off of lat_bwd_kernel is only valid on gfx9 (see the comment above)"
#endif
// Workspace contract (dtc_cenet_latent_fwd / _bwd): a backward call consumes what the forward call on the SAME workspace
// left behind (mask, info, the zeroed ticket), so the two are ordered by data dependence; a forward call for the next
// step must not be issued concurrently with a backward call still using the workspace (it would overwrite mask / info
// as well as re-zero the ticket).

namespace {

constexpr int LAT = 16, MU = 19, LD = 35;
constexpr int NB1 = 2048, NB2 = 2048, NB3 = 1024;
constexpr int MAX_BLK = 256;

struct Ws {
    double part[MAX_BLK * 2];
    float gpart[MAX_BLK];
    unsigned hist1[NB1];        // hist1 .. ticket: contiguous scratch header, zeroed by the stats kernel
    unsigned hist2[NB2];
    unsigned hist3[NB3];
    unsigned spare;
    unsigned ticket;            // arrival ticket of the backward kernel (the last workgroup resets it)
};

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void lat_stats_kernel(const float* __restrict__ mulv, long long n, Ws* ws,
                                                        int* __restrict__ info) {
    __shared__ double sh[4];
    if (blockIdx.x == 0 && threadIdx.x < 4) info[threadIdx.x] = threadIdx.x == 1 ? 0x7f7f7f7f : 0;     // consumed 4 launches later
    // scratch header (the three histograms of the later launches + the backward kernel's ticket): zeroed here instead of
    // by a memset node in front of the chain
    static_assert(offsetof(Ws, ticket) == offsetof(Ws, hist1) + sizeof(unsigned) * (NB1 + NB2 + NB3 + 1), "scratch header must be contiguous");
    unsigned* head = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + offsetof(Ws, hist1));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NB1 + NB2 + NB3 + 2; i += gridDim.x * blockDim.x) head[i] = 0u;
    double s = 0.0, q = 0.0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const double x = (double)mulv[(e >> 4) * LD + MU + (e & 15)];
        s += x;
        q += x * x;
    }
    s = block_sum_d(s, sh);
    q = block_sum_d(q, sh);
    if (threadIdx.x == 0) {
        ws->part[blockIdx.x * 2] = s;
        ws->part[blockIdx.x * 2 + 1] = q;
    }
}

// thresholds from the fp64 moments (every block recomputes them: nblk <= 256 partials)
__device__ __forceinline__ void thresholds(const Ws* ws, int nblk, long long n, float& lo, float& hi, double* sh) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        s += ws->part[i * 2];
        q += ws->part[i * 2 + 1];
    }
    s = block_sum_d(s, sh);
    q = block_sum_d(q, sh);
    const double mean = s / (double)n;
    double var = (q - s * s / (double)n) / (double)(n - 1);
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, thr = 2.0f * (float)sqrt(var);
    lo = meanf - thr;
    hi = meanf + thr;
}

// Block-wide: find the bin holding 0-based rank k in hist[nbins] (nbins = 256*per).  Returns the
// bin and the rank inside it; also the histogram total.  All threads get the result.
__device__ void find_bin(const unsigned* __restrict__ hist, int nbins, unsigned long long k, bool k_is_median,
                         unsigned& bin, unsigned long long& krem, unsigned long long& total, unsigned long long* sh) {
    const int per = nbins / 256;
    unsigned long long mine = 0;
    unsigned hv[8];                                                // per <= 8
    for (int i = 0; i < per; ++i) {
        hv[i] = hist[threadIdx.x * per + i];
        mine += hv[i];
    }
    sh[threadIdx.x] = mine;
    __syncthreads();
    // inclusive scan (Hillis-Steele) over 256 entries
    for (int off = 1; off < 256; off <<= 1) {
        unsigned long long v = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    total = sh[255];
    if (k_is_median) k = total > 0 ? (total - 1) / 2 : 0;
    const unsigned long long incl = sh[threadIdx.x], excl = incl - mine;
    __syncthreads();
    if (k >= excl && k < incl) {
        unsigned long long c = excl;
        for (int i = 0; i < per; ++i) {
            const unsigned h = hv[i];
            if (k < c + h) {
                sh[256] = threadIdx.x * per + i;
                sh[257] = k - c;
                break;
            }
            c += h;
        }
    }
    __syncthreads();
    bin = (unsigned)sh[256];
    krem = sh[257];
    __syncthreads();
}

template <int LEVEL>
__global__ __launch_bounds__(256) void lat_hist_kernel(const float* __restrict__ mulv, long long n, int nblk_stats,
                                                       Ws* ws) {
    __shared__ double shd[4];
    __shared__ unsigned long long shs[258];
    __shared__ unsigned lh[2048];
    float lo, hi;
    thresholds(ws, nblk_stats, n, lo, hi, shd);
    unsigned b1 = 0, b2 = 0;
    unsigned long long krem = 0, total = 0;
    if (LEVEL >= 2) find_bin(ws->hist1, NB1, 0, true, b1, krem, total, shs);
    if (LEVEL >= 3) find_bin(ws->hist2, NB2, krem, false, b2, krem, total, shs);
    constexpr int NB = LEVEL == 3 ? NB3 : 2048;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float x = mulv[(e >> 4) * LD + MU + (e & 15)];
        if (x < lo || x > hi) continue;
        const unsigned key = f2key(x);
        if (LEVEL == 1) atomicAdd(&lh[key >> 21], 1u);
        if (LEVEL == 2 && (key >> 21) == b1) atomicAdd(&lh[(key >> 10) & 2047u], 1u);
        if (LEVEL == 3 && (key >> 21) == b1 && ((key >> 10) & 2047u) == b2) atomicAdd(&lh[key & 1023u], 1u);
    }
    __syncthreads();
    unsigned* gh = LEVEL == 1 ? ws->hist1 : (LEVEL == 2 ? ws->hist2 : ws->hist3);
    for (int i = threadIdx.x; i < NB; i += blockDim.x)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

__global__ __launch_bounds__(256) void lat_apply_kernel(float* __restrict__ mulv, const float* __restrict__ eps,
                                                        float* __restrict__ z, uint8_t* __restrict__ mask,
                                                        int* __restrict__ info, long long n, int nblk_stats, Ws* ws,
                                                        amax_u32* __restrict__ z_amax, void* __restrict__ zmu_img) {
    __shared__ amax_u32 red_z[4];
    amax_u32 mz = 0u;                                  // largest |z| this thread writes (amax record, two-term fp16 GEMM path)
    __shared__ double shd[4];
    __shared__ unsigned long long shs[258];
    __shared__ int cnt[4];
    float lo, hi;
    thresholds(ws, nblk_stats, n, lo, hi, shd);
    unsigned b1, b2, b3;
    unsigned long long krem, total;
    find_bin(ws->hist1, NB1, 0, true, b1, krem, total, shs);
    find_bin(ws->hist2, NB2, krem, false, b2, krem, total, shs);
    find_bin(ws->hist3, NB3, krem, false, b3, krem, total, shs);
    const unsigned mkey = (b1 << 21) | (b2 << 10) | b3;
    const float median = key2f(mkey);
    int nout = 0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const long long b = e >> 4;
        const int j = (int)(e & 15);
        float x = mulv[b * LD + MU + j];
        const bool out = (x < lo) || (x > hi);
        if (out) {
            x = median;
            mulv[b * LD + MU + j] = x;
            ++nout;
        } else if (f2key(x) == mkey) {
            atomicMin(&info[1], (int)e);
        }
        mask[e] = out ? 1 : 0;
        const float zv = eps[e] * expf(0.5f * x) + mulv[b * LD + 3 + j];
        z[e] = zv;
        mz = abs_bits(zv) > mz ? abs_bits(zv) : mz;
        if (zmu_img) {
            // operand image of [z | mu[:, :3]] (19 columns: the decoder's / the actor's narrow input block), written here instead of by
            // a pack launch: the 16 lanes of a row (n and the stride are multiples of 16) agree on the row's exponent
            const float mu3 = j < 3 ? mulv[b * LD + j] : 0.0f;
            static_assert(LAT == 16 && 256 % LAT == 0, "the row's 16 lanes enter the loop together: n and the grid stride are multiples of 16");
            u32 mb = finite_bits(zv);
            mb = finite_bits(mu3) > mb ? finite_bits(mu3) : mb;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const u32 o = (u32)__shfl_xor((int)mb, off, 16);
                mb = o > mb ? o : mb;
            }
            const int ex = hi_exp(mb);
            const long long B = n >> 4;
            hi_store_elem(zmu_img, MU, (int)b, j, zv, ex);
            hi_store_elem(zmu_img, MU, (int)b, 16 + j, mu3, ex);          // columns 16..18 = mu[:, :3], 19..31 = padding (zero)
            if (j == 0) hi_store_row_exp(zmu_img, B, MU, (int)b, ex);
        }
    }
    amax_publish_block(z_amax, mz, red_z);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nout += __shfl_xor(nout, off, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = nout;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&info[0], cnt[0] + cnt[1] + cnt[2] + cnt[3]);
        if (blockIdx.x == 0) info[2] = (int)__float_as_uint(median);
    }
}


__global__ __launch_bounds__(256) void lat_bwd_kernel(float* __restrict__ dmulv, const float* __restrict__ dz,
                                                      const float* __restrict__ eps, const float* __restrict__ mulv,
                                                      const uint8_t* __restrict__ mask, const int* __restrict__ info,
                                                      long long n, Ws* ws, amax_u32* __restrict__ dmulv_amax) {
    // dmulv_amax: amax record of the WHOLE [B, 35] gradient after this call (the columns this kernel rewrites, the three leading ones it
    // leaves alone -- read for the record only --, and the median element's late update)
    __shared__ float shf[4];
    __shared__ amax_u32 red_m[4];
    amax_u32 md = 0u;
    float acc = 0.f;
    const int em = info[1];                           // flat index of the median element (written by the forward call)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const long long b = e >> 4;
        const int j = (int)(e & 15);
        const float g = dz[e];
        const float gmu = dmulv[b * LD + 3 + j] + g;                  // z = ... + mu[:, 3:]
        dmulv[b * LD + 3 + j] = gmu;
        md = abs_bits(gmu) > md ? abs_bits(gmu) : md;
        if (dmulv_amax != nullptr && j < 3) md = abs_bits(dmulv[b * LD + j]) > md ? abs_bits(dmulv[b * LD + j]) : md;
        const float lv = mulv[b * LD + MU + j];
        float glv = dmulv[b * LD + MU + j] + g * eps[e] * (0.5f * expf(0.5f * lv));
        if (mask[e]) {
            acc += glv;
            glv = 0.f;
        }
        // the median element is read again by the LAST workgroup of this launch: write it through (sc1), all others plain
        if (e == (long long)em) __hip_atomic_store(&dmulv[b * LD + MU + j], glv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dmulv[b * LD + MU + j] = glv;
        md = abs_bits(glv) > md ? abs_bits(glv) : md;
    }
    amax_publish_block(dmulv_amax, md, red_m);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) shf[threadIdx.x >> 6] = acc;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's stores (incl. the write-through one) have landed
    __syncthreads();
    // gradient of the median: the summed gradient of all replaced entries flows to the median element.  The last
    // workgroup to arrive (ticket) adds the per-workgroup sums in index order (deterministic).  What it reads from other
    // workgroups -- their partial sums and the median element's own gradient -- was written through (agent-scope relaxed
    // stores) and is read with agent-scope loads: no cache-maintenance fence (see the header comment).
    __shared__ unsigned last;
    __shared__ float red[MAX_BLK];
    if (threadIdx.x == 0) {
        __hip_atomic_store(&ws->gpart[blockIdx.x], (shf[0] + shf[1]) + (shf[2] + shf[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(&ws->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (last) {                                       // whole workgroup: one load per thread, then the sum in index order
        red[threadIdx.x] = threadIdx.x < gridDim.x ? __hip_atomic_load(&ws->gpart[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (unsigned i = 0; i < gridDim.x; ++i) tot += red[i];
            if (em != 0x7f7f7f7f && em >= 0) {
                float* p = &dmulv[(long long)(em >> 4) * LD + MU + (em & 15)];
                const float cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *p = cur + tot;
                if (dmulv_amax != nullptr) atomicMax(dmulv_amax, abs_bits(cur + tot));
            }
            __hip_atomic_store(&ws->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
        }
    }
}

int grid_for(long long n) {
    long long g = dtc::ceil_div(n, 1024);
    return (int)(g < 1 ? 1 : (g > MAX_BLK ? MAX_BLK : g));
}

}  // namespace

extern "C" int64_t dtc_cenet_workspace(int B) {
    (void)B;
    return (int64_t)sizeof(Ws);
}

extern "C" int dtc_cenet_latent_fwd(float* mulv, const float* eps, float* z, uint8_t* mask, int32_t* info,
                                    void* workspace, int B, uint32_t* z_amax, void* stream) {
    return dtc_cenet_latent_fwd_img(mulv, eps, z, mask, info, workspace, B, z_amax, nullptr, stream);
}

extern "C" int dtc_cenet_latent_fwd_img(float* mulv, const float* eps, float* z, uint8_t* mask, int32_t* info,
                                        void* workspace, int B, uint32_t* z_amax, void* zmu_img, void* stream) {
    DTC_REQUIRE(B > 0 && (long long)B * LAT >= 2, "bad batch %d", B);
    DTC_REQUIRE(dtc::aligned16(zmu_img), "unaligned image");
    DTC_REQUIRE(mulv && eps && z && mask && info && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    Ws* ws = (Ws*)workspace;
    const long long n = (long long)B * LAT;
    const int g = grid_for(n);
    dtc::ProfScope prof("cenet_latent_fwd", (double)n * 4.0 * 6, s);
    hipLaunchKernelGGL(lat_stats_kernel, dim3(g), dim3(256), 0, s, mulv, n, ws, info);
    hipLaunchKernelGGL(lat_hist_kernel<1>, dim3(g), dim3(256), 0, s, mulv, n, g, ws);
    hipLaunchKernelGGL(lat_hist_kernel<2>, dim3(g), dim3(256), 0, s, mulv, n, g, ws);
    hipLaunchKernelGGL(lat_hist_kernel<3>, dim3(g), dim3(256), 0, s, mulv, n, g, ws);
    hipLaunchKernelGGL(lat_apply_kernel, dim3(g), dim3(256), 0, s, mulv, eps, z, mask, info, n, g, ws, (amax_u32*)z_amax, zmu_img);
    return dtc::check_launch("cenet_latent_fwd");
}

extern "C" int dtc_cenet_latent_bwd(float* dmulv, const float* dz, const float* eps, const float* mulv,
                                    const uint8_t* mask, const int32_t* info, void* workspace, int B, uint32_t* dmulv_amax,
                                    void* stream) {
    DTC_REQUIRE(B > 0, "bad batch %d", B);
    DTC_REQUIRE(dmulv && dz && eps && mulv && mask && info && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    Ws* ws = (Ws*)workspace;
    const long long n = (long long)B * LAT;
    const int g = grid_for(n);
    dtc::ProfScope prof("cenet_latent_bwd", (double)n * 4.0 * 6, s);
    hipLaunchKernelGGL(lat_bwd_kernel, dim3(g), dim3(256), 0, s, dmulv, dz, eps, mulv, mask, info, n, ws, (amax_u32*)dmulv_amax);
    return dtc::check_launch("cenet_latent_bwd");
}
