// LSTM recurrence (forward + BPTT) for gfx950, built on the fp32 MFMA GEMMs of gemm.hip.
//
// Reference: torch.nn.LSTM inside `Memory` (rsl_rl/rsl_rl/modules/actor_critic_recurrent.py:92-116; `rnn_type='lstm'` is the
// class default, :93-97), gate order (i, f, g, o):
//     a = gi_t + h_{t-1} W_hh^T + b_hh;  i, f, o = sigmoid(a_i, a_f, a_o), g = tanh(a_g)
//     c_t = f * c_{t-1} + i * g;  h_t = o * tanh(c_t)
// Same structure as gru.hip (all launches issued from this C++ loop, no Python between time steps):
//   forward  t = 0..T-1 : gh = h_{t-1} W_hh^T + b_hh (dtc_linear_fwd, R rows) + lstm_gate_fwd_kernel (saves i, f, g, o)
//   backward t = T-1..0 : lstm_gate_bwd_kernel (da_t -> dgi_t, dc_{t-1});  dh_{t-1} = da_t W_hh as FOUR H-long chunks side by
//                         side (dtc_linear_dgrad_split), added in a fixed order by the next gate kernel
//            after loop : dW_hh, db_hh = [da_0..da_{T-1}]^T [h_{-1}..h_{T-2}]   (ONE dtc_linear_wgrad over T*R rows; for an LSTM
//                         the gradient w.r.t. gi IS the gradient w.r.t. gh, so dgi doubles as the operand)
// The input projection gi = x W_ih^T + b_ih and its weight gradient are plain dtc_linear_fwd / dtc_linear_wgrad calls made
// by the caller.  Padded steps need no masks: their output gradients are zero.
#include <stdlib.h>

#include "common.hpp"

namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// one thread per (row, hidden unit)
__global__ __launch_bounds__(256) void lstm_gate_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                            const float* __restrict__ cprev, float* __restrict__ hout,
                                                            float* __restrict__ cout, float* __restrict__ gates, int R, int H) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)R * H) return;
    const long long row = e / H;
    const int j = (int)(e - row * H);
    const float* a = gi + row * 4 * H;
    const float* b = gh + row * 4 * H;
    const float i = sigmoidf(a[j] + b[j]);
    const float f = sigmoidf(a[H + j] + b[H + j]);
    const float g = tanhf(a[2 * H + j] + b[2 * H + j]);
    const float o = sigmoidf(a[3 * H + j] + b[3 * H + j]);
    const float c = f * cprev[e] + i * g;
    cout[e] = c;
    hout[e] = o * tanhf(c);
    float* s = gates + row * 4 * H;
    s[j] = i;
    s[H + j] = f;
    s[2 * H + j] = g;
    s[3 * H + j] = o;
}

// dc (in/out): gradient flowing into c_t from step t+1 on entry (zero at t = T-1), into c_{t-1} on exit.  `part` holds
// the four chunks of dh_t's recurrent part (da_{t+1} W_hh; NULL at t = T-1); dhs_t is the output gradient of step t.
__global__ __launch_bounds__(256) void lstm_gate_bwd_kernel(const float* __restrict__ dhs_t, const float* __restrict__ part,
                                                            float* __restrict__ dc, const float* __restrict__ gates,
                                                            const float* __restrict__ cprev, const float* __restrict__ cnow,
                                                            float* __restrict__ dgi, int R, int H) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)R * H) return;
    const long long row = e / H;
    const int j = (int)(e - row * H);
    const float* s = gates + row * 4 * H;
    const float i = s[j], f = s[H + j], g = s[2 * H + j], o = s[3 * H + j];
    float dh = dhs_t[e];
    if (part) {
        const long long rh = (long long)R * H;
        dh = (((dh + part[e]) + part[rh + e]) + part[2 * rh + e]) + part[3 * rh + e];
    }
    const float tc = tanhf(cnow[e]);
    const float dct = dc[e] + dh * o * (1.0f - tc * tc);
    float* d = dgi + row * 4 * H;
    d[j] = (dct * g) * (i * (1.0f - i));
    d[H + j] = (dct * cprev[e]) * (f * (1.0f - f));
    d[2 * H + j] = (dct * i) * (1.0f - g * g);
    d[3 * H + j] = (dh * tc) * (o * (1.0f - o));
    dc[e] = dct * f;
}

// dh0 <- the four chunks of the last W_hh product
__global__ __launch_bounds__(256) void lstm_sum_parts_kernel(float* __restrict__ dh, const float* __restrict__ part, long long rh) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < rh) dh[e] = ((part[e] + part[rh + e]) + part[2 * rh + e]) + part[3 * rh + e];
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

// workspace layout: [ gh (forward) / the four dh chunks (backward): R*4H floats | wgrad partials ]
extern "C" int64_t dtc_lstm_workspace(int T, int R, int H) {
    if (T <= 0 || R <= 0 || H <= 0) return 0;
    return (int64_t)R * 4 * H * sizeof(float) + 16 + dtc_linear_wgrad_workspace(T * R, 4 * H, H);
}

extern "C" int dtc_lstm_fwd(const float* gi, const float* h0, const float* c0, const float* W_hh, const float* b_hh,
                            float* hs_all, float* cs_all, float* gates, void* workspace, int T, int R, int H, void* stream) {
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    DTC_REQUIRE(gi && h0 && c0 && W_hh && b_hh && hs_all && cs_all && gates && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* gh = (float*)workspace;
    const size_t RH = (size_t)R * H, R4H = 4 * RH;
    if (hipMemcpyAsync(hs_all, h0, RH * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(cs_all, c0, RH * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        dtc::set_error("lstm_fwd: initial state copy failed");
        return DTC_ERR_LAUNCH;
    }
    const unsigned grid = (unsigned)dtc::ceil_div((int64_t)RH, 256);
    for (int t = 0; t < T; ++t) {
        const float* hprev = hs_all + (size_t)t * RH;
        const DtcSegMat X = plain(hprev, H, H, R);
        int rc = dtc_linear_fwd(&X, W_hh, b_hh, gh, 4 * H, R, 4 * H, H, DTC_ACT_NONE, stream);
        if (rc != DTC_OK) return rc;
        dtc::ProfScope prof("lstm_gate_fwd", (double)RH * 4.0 * 15, s);
        hipLaunchKernelGGL(lstm_gate_fwd_kernel, dim3(grid), dim3(256), 0, s, gi + (size_t)t * R4H, gh, cs_all + (size_t)t * RH,
                           hs_all + (size_t)(t + 1) * RH, cs_all + (size_t)(t + 1) * RH, gates + (size_t)t * R4H, R, H);
    }
    return dtc::check_launch("lstm_fwd");
}

extern "C" int dtc_lstm_bwd(const float* dhs, const float* hs_all, const float* cs_all, const float* gates, const float* W_hh,
                            float* dgi, float* dW_hh, float* db_hh, float* dh0, float* dc0, void* workspace, int T, int R,
                            int H, void* stream) {
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    DTC_REQUIRE(dhs && hs_all && cs_all && gates && W_hh && dgi && dW_hh && db_hh && dh0 && dc0 && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t RH = (size_t)R * H, R4H = 4 * RH;
    float* part = (float*)workspace;              // [4][R][H]
    void* wg_ws = (void*)(((uintptr_t)(part + R4H) + 15) & ~(uintptr_t)15);
    if (hipMemsetAsync(dc0, 0, RH * sizeof(float), s) != hipSuccess) {
        dtc::set_error("lstm_bwd: memset failed");
        return DTC_ERR_LAUNCH;
    }
    const unsigned grid = (unsigned)dtc::ceil_div((int64_t)RH, 256);
    for (int t = T - 1; t >= 0; --t) {
        float* da_t = dgi + (size_t)t * R4H;
        {
            dtc::ProfScope prof("lstm_gate_bwd", (double)RH * 4.0 * 16, s);
            hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3(grid), dim3(256), 0, s, dhs + (size_t)t * RH,
                               t == T - 1 ? (const float*)nullptr : (const float*)part, dc0, gates + (size_t)t * R4H,
                               cs_all + (size_t)t * RH, cs_all + (size_t)(t + 1) * RH, da_t, R, H);
        }
        int rc = dtc_linear_dgrad_split(da_t, 4 * H, W_hh, part, H, (int64_t)RH, R, 4 * H, H, 4, stream);
        if (rc != DTC_OK) return rc;
    }
    hipLaunchKernelGGL(lstm_sum_parts_kernel, dim3(grid), dim3(256), 0, s, dh0, part, (long long)RH);
    const DtcSegMat Hprev = plain(hs_all, H, H, (int64_t)T * R);
    int rc = dtc_linear_wgrad(dgi, 4 * H, &Hprev, dW_hh, db_hh, wg_ws, T * R, 4 * H, H, stream);
    if (rc != DTC_OK) return rc;
    return dtc::check_launch("lstm_bwd");
}
