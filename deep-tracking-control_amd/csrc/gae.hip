// GAE(lambda) scan + advantage normalisation + mini-batch row gather for gfx950.
//
//  * dtc_gae            rsl_rl/rsl_rl/storage/rollout_storage.py:138-150  (24-iteration Python loop
//                       of [N,1] torch ops -> one kernel, lane per env, time loop in registers;
//                       every load/store is coalesced over envs: 17 B/env-step)
//  * dtc_adv_sqdev /    rollout_storage.py:151-152  (mean, unbiased std over all T*N samples; the
//    dtc_adv_normalize  two reductions are exposed separately so data-parallel ranks can all-reduce
//                       the two scalars in between -- SURVEY.md 8e item 2)
//  * dtc_gather_rows    rollout_storage.py:195-209  (`tensor.flatten(0,1)[batch_idx]`)
//
// Compiled with -ffp-contract=off: the scan reproduces oracle/gae.py bit for bit.
#include "amax.hpp"
#include "common.hpp"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) t += sh[w];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                  const uint8_t* __restrict__ dones,
                                                  const float* __restrict__ last_values, float gamma, float lam,
                                                  float* __restrict__ returns, float* __restrict__ adv_out,
                                                  double* __restrict__ partials, int T, int N) {
    __shared__ double sh[4];
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    double local = 0.0;
    if (n < N) {
        float adv = 0.0f;
        float next_v = last_values[n];
        for (int t = T - 1; t >= 0; --t) {
            const int64_t o = (int64_t)t * N + n;
            const float v = values[o];
            const float nnt = 1.0f - (float)dones[o];
            const float ng = nnt * gamma;
            const float delta = (rewards[o] + ng * next_v) - v;
            adv = delta + (ng * lam) * adv;
            const float ret = adv + v;
            returns[o] = ret;
            const float a = ret - v;
            adv_out[o] = a;
            local += (double)a;
            next_v = v;
        }
    }
    const double tot = block_sum(local, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// stats[slot] = sum(partials[0..np))   (single block; deterministic order)
__global__ __launch_bounds__(256) void finalize_sum_kernel(const double* __restrict__ partials, int np,
                                                           double* __restrict__ stats, int slot) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) v += partials[i];
    const double tot = block_sum(v, sh);
    if (threadIdx.x == 0) stats[slot] = tot;
}

__global__ __launch_bounds__(256) void sqdev_kernel(const float* __restrict__ adv, const double* __restrict__ stats,
                                                    double count, double* __restrict__ partials, int64_t n) {
    __shared__ double sh[4];
    const double mean = stats[0] / count;
    double v = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = (double)adv[i] - mean;
        v += d * d;
    }
    const double tot = block_sum(v, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void normalize_kernel(float* __restrict__ adv, const double* __restrict__ stats,
                                                        double count, int64_t n) {
    const float mean = (float)(stats[0] / count);
    const float stdv = (float)sqrt(stats[1] / (count - 1.0));
    const float den = stdv + 1e-8f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        adv[i] = (adv[i] - mean) / den;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx,
                                                          T* __restrict__ dst, int64_t rows, int64_t row_elems) {
    const int64_t total = rows * row_elems;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / row_elems;
        const int64_t c = e - r * row_elems;
        dst[e] = src[idx[r] * row_elems + c];
    }
}

// 16 bytes per thread (rows of a multiple of 4 floats, 16-byte aligned matrices): one row per 32-bit division instead of one
// 64-bit division per element -- the recurrent trainers scatter [24576, 512] gradients into the padded trajectories 40 times per
// update (101 -> ~35 us each)
__global__ __launch_bounds__(256) void scatter_rows4_kernel(const float4* __restrict__ src, const int64_t* __restrict__ idx,
                                                            float4* __restrict__ dst, int rows, int row_vec) {
    const long long total = (long long)rows * row_vec;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / row_vec), c = (int)(e - (long long)r * row_vec);
        dst[idx[r] * row_vec + c] = src[e];
    }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                           float* __restrict__ dst, int64_t rows, int64_t row_elems) {
    const int64_t total = rows * row_elems;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / row_elems;
        dst[idx[r] * row_elems + (e - r * row_elems)] = src[e];
    }
}

constexpr int MAX_PARTIALS = 4096;
// Reduction partials: one lazily allocated scratch PER DEVICE (keyed by the device that is current at the call, which
// is the device of the caller's tensors).  Calls that share a device must be ordered on one stream -- the storage
// issues dtc_gae / dtc_adv_sqdev back to back on torch's current stream; they are not meant to run concurrently.
double* partial_buffer() {
    constexpr int MAX_DEV = 64;
    static double* buf[MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    if (!buf[dev]) {
        if (hipMalloc(&buf[dev], sizeof(double) * MAX_PARTIALS) != hipSuccess) buf[dev] = nullptr;
    }
    return buf[dev];
}

}  // namespace

extern "C" int dtc_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values,
                       float gamma, float lam, float* returns, float* advantages, double* stats, int T, int N,
                       void* stream) {
    DTC_REQUIRE(T > 0 && N > 0, "bad shape T=%d N=%d", T, N);
    DTC_REQUIRE(rewards && values && dones && last_values && returns && advantages && stats, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    double* partials = partial_buffer();
    DTC_REQUIRE(partials != nullptr, "partial buffer allocation failed");
    const int grid = (int)dtc::ceil_div(N, 256);
    DTC_REQUIRE(grid <= MAX_PARTIALS, "N too large for one call (max %d envs)", MAX_PARTIALS * 256);
    {
        dtc::ProfScope prof("gae_scan", (double)T * N * 17.0, s);
        hipLaunchKernelGGL(gae_kernel, dim3(grid), dim3(256), 0, s, rewards, values, dones, last_values, gamma, lam,
                           returns, advantages, partials, T, N);
    }
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, s, partials, grid, stats, 0);
    return dtc::check_launch("gae");
}

extern "C" int dtc_adv_sqdev(const float* advantages, double* stats, int64_t n_local, double count, void* stream) {
    DTC_REQUIRE(advantages && stats && n_local > 0 && count > 1.0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    double* partials = partial_buffer();
    DTC_REQUIRE(partials != nullptr, "partial buffer allocation failed");
    const int grid = (int)(dtc::ceil_div(n_local, 256) < 1024 ? dtc::ceil_div(n_local, 256) : 1024);
    hipLaunchKernelGGL(sqdev_kernel, dim3(grid), dim3(256), 0, s, advantages, stats, count, partials, n_local);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, s, partials, grid, stats, 1);
    return dtc::check_launch("adv_sqdev");
}

extern "C" int dtc_adv_normalize(float* advantages, const double* stats, int64_t n_local, double count, void* stream) {
    DTC_REQUIRE(advantages && stats && n_local > 0 && count > 1.0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)(dtc::ceil_div(n_local, 256) < 2048 ? dtc::ceil_div(n_local, 256) : 2048);
    hipLaunchKernelGGL(normalize_kernel, dim3(grid), dim3(256), 0, s, advantages, stats, count, n_local);
    return dtc::check_launch("adv_normalize");
}

extern "C" int dtc_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t row_bytes,
                               void* stream) {
    DTC_REQUIRE(rows >= 0 && row_bytes > 0, "bad shape");
    if (rows == 0) return DTC_OK;
    DTC_REQUIRE(src && idx && dst, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof("gather_rows", 2.0 * (double)rows * (double)row_bytes, s);
    auto grid_for = [](int64_t total) { return (unsigned)(dtc::ceil_div(total, 256) < 8192 ? dtc::ceil_div(total, 256) : 8192); };
    if (row_bytes % 16 == 0 && dtc::aligned16(src) && dtc::aligned16(dst)) {
        const int64_t re = row_bytes / 16;
        hipLaunchKernelGGL(gather_rows_kernel<uint4>, dim3(grid_for(rows * re)), dim3(256), 0, s, (const uint4*)src, idx,
                           (uint4*)dst, rows, re);
    } else if (row_bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 &&
               (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
        const int64_t re = row_bytes / 4;
        hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(grid_for(rows * re)), dim3(256), 0, s,
                           (const uint32_t*)src, idx, (uint32_t*)dst, rows, re);
    } else {
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, dim3(grid_for(rows * row_bytes)), dim3(256), 0, s,
                           (const uint8_t*)src, idx, (uint8_t*)dst, rows, row_bytes);
    }
    return dtc::check_launch("gather_rows");
}

namespace {
// dst[row, 0:cols] = the segments of a DtcSegMat side by side (row-gathered where a segment asks for it): one thread per element
__global__ __launch_bounds__(256) void pack_cols_kernel(const DtcSegMat X, float* __restrict__ dst, long long ld_dst, long long rows,
                                                        amax_u32* __restrict__ dst_amax) {
    const long long total = rows * X.cols;
    amax_u32 m = 0u;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / X.cols;
        int c = (int)(e - r * X.cols), s = 0;
        while (s < X.nseg - 1 && c >= X.seg[s].width) c -= X.seg[s++].width;
        const DtcSeg& g = X.seg[s];
        const long long src_row = g.gather ? X.idx[r] : r;
        const float v = g.ptr[src_row * g.ld + g.col0 + c];
        dst[r * ld_dst + (e - r * X.cols)] = v;
        m = abs_bits(v) > m ? abs_bits(v) : m;
    }
    __shared__ amax_u32 red[4];
    amax_publish_block(dst_amax, m, red);
}
}  // namespace

// The narrow leading blocks of a layer input (e.g. the actor's [obs 53 | z 16 | mu 3], the critic's [obs 53 | base_vel 3]) packed
// into ONE dense [rows, cols] matrix: the GEMM kernels pad every segment of an operand to whole 16-k stages / 128-column tiles,
// so three narrow segments cost three stages / three tiles where the packed block costs one (actor_critic_decoder.py:431, 550
// `torch.cat` of the same pieces -- here only of the narrow ones; wide blocks stay in place as segments).
extern "C" int dtc_pack_cols(const DtcSegMat* X, float* dst, int64_t ld_dst, int64_t rows, uint32_t* dst_amax, void* stream) {
    DTC_REQUIRE(X && dst && rows >= 0 && X->nseg >= 1 && X->nseg <= 4 && ld_dst >= X->cols, "bad arguments");
    if (rows == 0) return DTC_OK;
    int cols = 0;
    for (int i = 0; i < X->nseg; ++i) {
        DTC_REQUIRE(X->seg[i].ptr && X->seg[i].width > 0 && (!X->seg[i].gather || X->idx), "segment %d: null source / gather without idx", i);
        cols += X->seg[i].width;
    }
    DTC_REQUIRE(cols == X->cols, "segments cover %d columns, descriptor says %d", cols, X->cols);
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof("pack_cols", 8.0 * (double)rows * cols, s);
    const long long total = rows * (long long)cols;
    const unsigned grid = (unsigned)(dtc::ceil_div(total, 256) < 16384 ? dtc::ceil_div(total, 256) : 16384);
    hipLaunchKernelGGL(pack_cols_kernel, dim3(grid), dim3(256), 0, s, *X, dst, (long long)ld_dst, (long long)rows, (amax_u32*)dst_amax);
    return dtc::check_launch("pack_cols");
}

extern "C" int dtc_scatter_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int64_t row_floats,
                                void* stream) {
    DTC_REQUIRE(rows >= 0 && row_floats > 0, "bad shape");
    if (rows == 0) return DTC_OK;
    DTC_REQUIRE(src && idx && dst, "null pointer");
    const int64_t total = rows * row_floats;
    if (row_floats % 4 == 0 && dtc::aligned16(src) && dtc::aligned16(dst) && rows < (1ll << 31)) {
        const unsigned grid4 = (unsigned)(dtc::ceil_div(total / 4, 256) < 16384 ? dtc::ceil_div(total / 4, 256) : 16384);
        hipLaunchKernelGGL(scatter_rows4_kernel, dim3(grid4), dim3(256), 0, (hipStream_t)stream, (const float4*)src, idx, (float4*)dst,
                           (int)rows, (int)(row_floats / 4));
        return dtc::check_launch("scatter_rows");
    }
    const unsigned grid = (unsigned)(dtc::ceil_div(total, 256) < 8192 ? dtc::ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, idx, dst, rows, row_floats);
    return dtc::check_launch("scatter_rows");
}
