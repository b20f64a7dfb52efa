// GRU recurrence (forward + BPTT) for gfx950, built on the fp32 MFMA GEMMs of gemm.hip.
//
// Reference: torch.nn.GRU(input, hidden=512, 1 layer) inside `Memory`
// (rsl_rl/rsl_rl/modules/actor_critic_recurrent.py:92-116, twin at actor_critic_decoder.py:584-614), run over
// padded trajectories [T, n_traj, .] with saved initial hidden states during the policy update (BPTT) and
// over [1, N, .] during the rollout.
//
// Structure (all launches are issued from this C++ loop -- no Python between time steps):
//   forward  t = 0..T-1 : gh = h_{t-1} W_hh^T + b_hh and the gate math in its epilogue: ONE kernel per step
//                         (dtc_gru_step_fwd in gemm.hip; saves r,z,n and gh_n; DTC_GRU_UNFUSED=1 selects the older
//                         dtc_linear_fwd + gru_gate_fwd_kernel pair)
//   backward t = T-1..0 : gate derivatives                  (gru_gate_bwd_kernel: dgi_t, dgh_t, dh*z)
//                         dh_{t-1} += dgh_t W_hh            (dtc_linear_dgrad_split: the 3H-long reduction runs as three
//                                                            H-long chunks side by side -- one step has only ~12 row
//                                                            tiles -- and the next gate kernel adds the three partial
//                                                            products in a fixed order)
//            after loop : dW_hh, db_hh = [dgh_0..dgh_{T-1}]^T [h_{-1}..h_{T-2}]   (ONE dtc_linear_wgrad over T*R rows)
// The input projection gi = x W_ih^T + b_ih (all T*R rows at once) and its weight gradient are plain
// dtc_linear_fwd / dtc_linear_wgrad calls made by the caller.  Padded steps need no masks: their output
// gradients are zero, so every quantity flowing backwards through them is zero as well.
#include <stdlib.h>

#include "common.hpp"

namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// one thread per (row, hidden unit)
__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ hprev, float* __restrict__ hout,
                                                           float* __restrict__ gates, float* __restrict__ hn, int R,
                                                           int H) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)R * H) return;
    const long long row = e / H;
    const int j = (int)(e - row * H);
    const float* gir = gi + row * 3 * H;
    const float* ghr = gh + row * 3 * H;
    const float r = sigmoidf(gir[j] + ghr[j]);
    const float z = sigmoidf(gir[H + j] + ghr[H + j]);
    const float ghn = ghr[2 * H + j];
    const float n = tanhf(gir[2 * H + j] + r * ghn);
    const float hp = hprev[e];
    hout[e] = (1.0f - z) * n + z * hp;
    float* g = gates + row * 3 * H;
    g[j] = r;
    g[H + j] = z;
    g[2 * H + j] = n;
    hn[e] = ghn;
}

// dh (in/out): on entry the direct part (dh_{t+1} * z_{t+1}) of the gradient flowing into h_t from step t+1 (zero at
// t = T-1); `part` holds the three chunks of its W_hh part (dgh_{t+1} W_hh, NULL at t = T-1); dhs_t is added here.
// On exit dh holds dh_t * z (the direct path to h_{t-1}); the W_hh path is produced by the following split dgrad.
__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(const float* __restrict__ dhs_t, float* __restrict__ dh,
                                                           const float* __restrict__ part, const float* __restrict__ gates,
                                                           const float* __restrict__ hn, const float* __restrict__ hprev,
                                                           float* __restrict__ dgi, float* __restrict__ dgh, int R, int H) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)R * H) return;
    const long long row = e / H;
    const int j = (int)(e - row * H);
    const float* g = gates + row * 3 * H;
    const float r = g[j], z = g[H + j], n = g[2 * H + j];
    float d = dhs_t[e] + dh[e];
    if (part) {
        const long long rh = (long long)R * H;
        d = ((d + part[e]) + part[rh + e]) + part[2 * rh + e];
    }
    const float ghn = hn[e];
    const float dn = d * (1.0f - z);
    const float dz = d * (hprev[e] - n);
    const float da_n = dn * (1.0f - n * n);
    const float da_z = dz * (z * (1.0f - z));
    const float da_r = (da_n * ghn) * (r * (1.0f - r));
    float* gi_o = dgi + row * 3 * H;
    float* gh_o = dgh + row * 3 * H;
    gi_o[j] = da_r;
    gi_o[H + j] = da_z;
    gi_o[2 * H + j] = da_n;
    gh_o[j] = da_r;
    gh_o[H + j] = da_z;
    gh_o[2 * H + j] = da_n * r;
    dh[e] = d * z;
}

// dh0 <- dh0 + the three chunks of the last W_hh product
__global__ __launch_bounds__(256) void gru_add_parts_kernel(float* __restrict__ dh, const float* __restrict__ part, long long rh) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < rh) dh[e] = ((dh[e] + part[e]) + part[rh + e]) + part[2 * rh + e];
}

DtcSegMat plain(const float* p, int64_t ld, int cols, int64_t rows) {
    DtcSegMat m;
    m.nseg = 1;
    m.cols = cols;
    m.idx = nullptr;
    m.seg[0] = DtcSeg{const_cast<float*>(p), ld, 0, cols, 0, 0, rows};
    return m;
}

}  // namespace

// workspace layout: [ gh: R*3H floats | dgh_all: T*R*3H floats | wgrad partials ]
extern "C" int64_t dtc_gru_workspace(int T, int R, int H) {
    if (T <= 0 || R <= 0 || H <= 0) return 0;
    const int64_t a = (int64_t)R * 3 * H * sizeof(float);
    const int64_t b = (int64_t)T * R * 3 * H * sizeof(float);
    return a + b + 16 + dtc_linear_wgrad_workspace(T * R, 3 * H, H);     // +16: the partials start 16-byte aligned
}

extern "C" int dtc_gru_fwd(const float* gi, const float* h0, const float* W_hh, const float* b_hh, float* hs_all,
                           float* gates, float* hn, void* workspace, int T, int R, int H, void* stream) {
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    DTC_REQUIRE(gi && h0 && W_hh && b_hh && hs_all && gates && hn && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* gh = (float*)workspace;
    const size_t RH = (size_t)R * H;
    if (hipMemcpyAsync(hs_all, h0, RH * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        dtc::set_error("gru_fwd: h0 copy failed");
        return DTC_ERR_LAUNCH;
    }
    static const bool unfused = getenv("DTC_GRU_UNFUSED") != nullptr;      // two-kernel step (GEMM + gate kernel)
    const unsigned grid = (unsigned)dtc::ceil_div((int64_t)RH, 256);
    for (int t = 0; t < T; ++t) {
        const float* hprev = hs_all + (size_t)t * RH;
        if (!unfused && H % 32 == 0) {
            int rc = dtc_gru_step_fwd(hprev, W_hh, b_hh, gi + (size_t)t * R * 3 * H, hs_all + (size_t)(t + 1) * RH,
                                      gates + (size_t)t * R * 3 * H, hn + (size_t)t * RH, R, H, stream);
            if (rc != DTC_OK) return rc;
            continue;
        }
        const DtcSegMat X = plain(hprev, H, H, R);
        int rc = dtc_linear_fwd(&X, W_hh, b_hh, gh, 3 * H, R, 3 * H, H, DTC_ACT_NONE, stream);
        if (rc != DTC_OK) return rc;
        dtc::ProfScope prof("gru_gate_fwd", (double)RH * 4.0 * 12, s);
        hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(grid), dim3(256), 0, s, gi + (size_t)t * R * 3 * H, gh, hprev,
                           hs_all + (size_t)(t + 1) * RH, gates + (size_t)t * R * 3 * H, hn + (size_t)t * RH, R, H);
    }
    return dtc::check_launch("gru_fwd");
}

extern "C" int dtc_gru_bwd(const float* dhs, const float* hs_all, const float* gates, const float* hn, const float* W_hh,
                           float* dgi, float* dW_hh, float* db_hh, float* dh0, void* workspace, int T, int R, int H,
                           void* stream) {
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    DTC_REQUIRE(dhs && hs_all && gates && hn && W_hh && dgi && dW_hh && db_hh && dh0 && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t RH = (size_t)R * H, R3H = (size_t)R * 3 * H;
    float* dgh_all = (float*)workspace + R3H;
    void* wg_ws = (void*)(((uintptr_t)(dgh_all + (size_t)T * R3H) + 15) & ~(uintptr_t)15);
    if (hipMemsetAsync(dh0, 0, RH * sizeof(float), s) != hipSuccess) {
        dtc::set_error("gru_bwd: memset failed");
        return DTC_ERR_LAUNCH;
    }
    const unsigned grid = (unsigned)dtc::ceil_div((int64_t)RH, 256);
    float* part = (float*)workspace;              // [3][R][H]: the region dtc_gru_fwd uses for gh
    for (int t = T - 1; t >= 0; --t) {
        float* dgh_t = dgh_all + (size_t)t * R3H;
        {
            dtc::ProfScope prof("gru_gate_bwd", (double)RH * 4.0 * 17, s);
            hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(grid), dim3(256), 0, s, dhs + (size_t)t * RH, dh0,
                               t == T - 1 ? (const float*)nullptr : (const float*)part, gates + (size_t)t * R3H,
                               hn + (size_t)t * RH, hs_all + (size_t)t * RH, dgi + (size_t)t * R3H, dgh_t, R, H);
        }
        int rc = dtc_linear_dgrad_split(dgh_t, 3 * H, W_hh, part, H, (int64_t)RH, R, 3 * H, H, 3, stream);
        if (rc != DTC_OK) return rc;
    }
    hipLaunchKernelGGL(gru_add_parts_kernel, dim3(grid), dim3(256), 0, s, dh0, part, (long long)RH);
    const DtcSegMat Hprev = plain(hs_all, H, H, (int64_t)T * R);
    int rc = dtc_linear_wgrad(dgh_all, 3 * H, &Hprev, dW_hh, db_hh, wg_ws, T * R, 3 * H, H, stream);
    if (rc != DTC_OK) return rc;
    return dtc::check_launch("gru_bwd");
}
