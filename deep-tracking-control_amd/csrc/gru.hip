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
                                                           float* __restrict__ dgi, float* __restrict__ dgh, int R, int H, int nparts) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)R * H) return;
    const long long row = e / H;
    const int j = (int)(e - row * H);
    const float* g = gates + row * 3 * H;
    const float r = g[j], z = g[H + j], n = g[2 * H + j];
    float d = dhs_t[e] + dh[e];
    if (part) {
        const long long rh = (long long)R * H;
        for (int c = 0; c < nparts; ++c) d += part[c * rh + e];              // fixed order
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

// the same with four consecutive hidden units per thread (H % 4 == 0, 16-byte aligned rows): 16-byte loads / stores -- the kernel moves
// 20 floats per (row, unit) and is bound by that traffic (66 MB per launch at R ~ 1470, H = 512)
// (blockIdx.y: which recurrence -- dtc_gru_bwd_multi runs the time step of two recurrences of one shape as one launch)
struct GateBwdPtrs {
    const float* dhs_t;
    float* dh;
    const float* part;
    const float* gates;
    const float* hn;
    const float* hprev;
    float* dgi;
    float* dgh;
};
__global__ __launch_bounds__(256) void gru_gate_bwd4_kernel(const GateBwdPtrs p0, const GateBwdPtrs p1, int R, int H, int nparts) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const GateBwdPtrs P = blockIdx.y == 0 ? p0 : p1;
    const float* __restrict__ dhs_t = P.dhs_t;
    float* __restrict__ dh = P.dh;
    const float* __restrict__ part = P.part;
    const float* __restrict__ gates = P.gates;
    const float* __restrict__ hn = P.hn;
    const float* __restrict__ hprev = P.hprev;
    float* __restrict__ dgi = P.dgi;
    float* __restrict__ dgh = P.dgh;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // group of four units
    const int hq = H >> 2;
    if (q >= (long long)R * hq) return;
    const long long row = q / hq;
    const int j = (int)(q - row * hq) * 4;
    const long long e = row * H + j;
    const float* g = gates + row * 3 * H;
    const f4 r = *reinterpret_cast<const f4*>(g + j), z = *reinterpret_cast<const f4*>(g + H + j), n = *reinterpret_cast<const f4*>(g + 2 * H + j);
    f4 d = *reinterpret_cast<const f4*>(dhs_t + e) + *reinterpret_cast<const f4*>(dh + e);
    if (part) {
        const long long rh = (long long)R * H;
        for (int c = 0; c < nparts; ++c) d += *reinterpret_cast<const f4*>(part + c * rh + e);      // fixed order
    }
    const f4 ghn = *reinterpret_cast<const f4*>(hn + e), hp = *reinterpret_cast<const f4*>(hprev + e);
    f4 da_r, da_z, da_n, da_nr, dz4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                       // the scalar kernel's arithmetic, element by element (same rounding)
        const float dn = d[k] * (1.0f - z[k]);
        const float dz = d[k] * (hp[k] - n[k]);
        da_n[k] = dn * (1.0f - n[k] * n[k]);
        da_z[k] = dz * (z[k] * (1.0f - z[k]));
        da_r[k] = (da_n[k] * ghn[k]) * (r[k] * (1.0f - r[k]));
        da_nr[k] = da_n[k] * r[k];
        dz4[k] = d[k] * z[k];
    }
    float* gi_o = dgi + row * 3 * H;
    float* gh_o = dgh + row * 3 * H;
    *reinterpret_cast<f4*>(gi_o + j) = da_r;
    *reinterpret_cast<f4*>(gi_o + H + j) = da_z;
    *reinterpret_cast<f4*>(gi_o + 2 * H + j) = da_n;
    *reinterpret_cast<f4*>(gh_o + j) = da_r;
    *reinterpret_cast<f4*>(gh_o + H + j) = da_z;
    *reinterpret_cast<f4*>(gh_o + 2 * H + j) = da_nr;
    *reinterpret_cast<f4*>(dh + e) = dz4;
}

// dh0 <- dh0 + the three chunks of the last W_hh product
__global__ __launch_bounds__(256) void gru_add_parts_kernel(float* __restrict__ dh, const float* __restrict__ part, long long rh, int nparts) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rh) return;
    float d = dh[e];
    for (int c = 0; c < nparts; ++c) d += part[c * rh + e];
    dh[e] = d;
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

// workspace layout: [ gh / the chunks of the W_hh data gradient: MAX_PARTS * R * H floats | dgh_all: T*R*3H floats | wgrad partials |
// the image of W_hh^T (split-precision path) ]
constexpr int MAX_PARTS = 6;
namespace {
// chunks of the 3H-long reduction of dh += dgh W_hh that run side by side: 3 on the single-pass path; the split path runs 128 x 128
// tiles (4 column tiles for H = 512), so it takes 6 to put ~290 workgroups on the chip (DTC_GRU_S3_PARTS = 1, 2, 3 or 6)
int gru_parts(int H, bool s3) {
    if (!s3) return 3;
    constexpr int env = 6;
    const int p = (env == 1 || env == 2 || env == 3 || env == 6) ? env : 6;
    return (3 * H / p) % 16 == 0 ? p : 3;
}
bool gru_s3(int H) {
    static const bool on = !(getenv("DTC_S3_WIMG") && atoi(getenv("DTC_S3_WIMG")) == 0) && !(getenv("DTC_GRU_S3") && atoi(getenv("DTC_GRU_S3")) == 0);
    return dtc_get_gemm_split() && on && H % 128 == 0;
}
void* gru_image_slot(void* workspace, int T, int R, int H) {
    float* dgh_all = (float*)workspace + (size_t)MAX_PARTS * R * H;
    void* wg_ws = (void*)(((uintptr_t)(dgh_all + (size_t)T * R * 3 * H) + 15) & ~(uintptr_t)15);
    return (char*)wg_ws + ((dtc_linear_wgrad_workspace(T * R, 3 * H, H) + 15) & ~(int64_t)15);
}
}  // namespace
extern "C" int64_t dtc_gru_workspace(int T, int R, int H) {
    if (T <= 0 || R <= 0 || H <= 0) return 0;
    const int64_t a = (int64_t)R * MAX_PARTS * H * sizeof(float);
    const int64_t b = (int64_t)T * R * 3 * H * sizeof(float);
    const int64_t img = dtc_s3_planes_bytes(H, 3 * H) > dtc_gru_s3_image_bytes(H) ? dtc_s3_planes_bytes(H, 3 * H) : dtc_gru_s3_image_bytes(H);
    const int64_t per_step = a + b + 16 + ((dtc_linear_wgrad_workspace(T * R, 3 * H, H) + 15) & ~(int64_t)15) + img;     // +16: 16-byte alignment
    return ((per_step + 255) & ~(int64_t)255) + dtc_gru_seq_workspace(R, H);            // + the persistent kernels' exchange buffers (csrc/gru_seq.hip)
}
namespace {
// the persistent kernels' region of a dtc_gru_workspace() buffer (its tail), 256-byte aligned inside the buffer's own alignment
void* gru_seq_slot(void* workspace, int T, int R, int H) {
    const int64_t total = dtc_gru_workspace(T, R, H), seq = dtc_gru_seq_workspace(R, H);
    return (char*)workspace + ((total - seq) & ~(int64_t)255);
}
}  // namespace

// byte offset of dgh_all [T, R, 3H] (the gradient w.r.t. the recurrent pre-activations, written by dtc_gru_bwd) inside the workspace
extern "C" int64_t dtc_gru_dgh_offset(int T, int R, int H) {
    (void)T;
    if (R <= 0 || H <= 0) return -1;
    return (int64_t)MAX_PARTS * R * H * (int64_t)sizeof(float);
}

extern "C" int dtc_gru_fwd(const float* gi, const float* h0, const float* W_hh, const float* b_hh, float* hs_all,
                           float* gates, float* hn, void* workspace, int T, int R, int H, void* stream) {
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    DTC_REQUIRE(gi && h0 && W_hh && b_hh && hs_all && gates && hn && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* gh = (float*)workspace;
    const size_t RH = (size_t)R * H;
    static const bool unfused = getenv("DTC_GRU_UNFUSED") != nullptr;      // two-kernel step (GEMM + gate kernel)
    // the whole recurrence as ONE persistent launch (csrc/gru_seq.hip) where the shape is served and the buffers allow 16-byte accesses
    if (!unfused && dtc_get_gemm_split() && dtc_gru_seq_supported(T, R, H, 0) && dtc::aligned16(gi) && dtc::aligned16(h0) && dtc::aligned16(hs_all) &&
        dtc::aligned16(gates) && dtc::aligned16(hn) && dtc::aligned16(workspace))
        return dtc_gru_seq_fwd(gi, h0, W_hh, b_hh, hs_all, gates, hn, gru_seq_slot(workspace, T, R, H), T, R, H, stream);
    if (hipMemcpyAsync(hs_all, h0, RH * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        dtc::set_error("gru_fwd: h0 copy failed");
        return DTC_ERR_LAUNCH;
    }
    const unsigned grid = (unsigned)dtc::ceil_div((int64_t)RH, 256);
    // split-precision steps (csrc/gru_s3.hip) for the passes of the update (T time steps share ONE image of W_hh); the one-step
    // calls of the rollout keep the single-pass kernel
    const bool s3 = !unfused && T >= 4 && gru_s3(H);
    void* img = s3 ? gru_image_slot(workspace, T, R, H) : nullptr;
    if (s3) {
        int rc = dtc_gru_s3_image(W_hh, img, H, 0, stream);
        if (rc != DTC_OK) return rc;
    }
    for (int t = 0; t < T; ++t) {
        const float* hprev = hs_all + (size_t)t * RH;
        if (s3) {
            int rc = dtc_gru_step_fwd_s3(hprev, img, b_hh, gi + (size_t)t * R * 3 * H, hs_all + (size_t)(t + 1) * RH,
                                         gates + (size_t)t * R * 3 * H, hn + (size_t)t * RH, R, H, stream);
            if (rc != DTC_OK) return rc;
            continue;
        }
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
                           float* dgi, float* dW_hh, float* db_hh, float* dh0, void* workspace, const int64_t* valid_rows, int n_valid,
                           int T, int R, int H, void* stream) {
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    DTC_REQUIRE(dhs && hs_all && gates && hn && W_hh && dgi && dh0 && workspace, "null pointer");
    DTC_REQUIRE((dW_hh == nullptr) == (db_hh == nullptr), "dW_hh and db_hh: both or neither");
    hipStream_t s = (hipStream_t)stream;
    const size_t RH = (size_t)R * H, R3H = (size_t)R * 3 * H;
    float* dgh_all = (float*)workspace + (size_t)MAX_PARTS * RH;
    void* wg_ws = (void*)(((uintptr_t)(dgh_all + (size_t)T * R3H) + 15) & ~(uintptr_t)15);
    void* wimage = gru_image_slot(workspace, T, R, H);
    const bool s3 = gru_s3(H);
    const int nparts = gru_parts(H, s3);
    if (s3) {
        int rc = dtc_gru_s3_image(W_hh, wimage, H, 1, stream);
        if (rc != DTC_OK) return rc;
    }
    if (hipMemsetAsync(dh0, 0, RH * sizeof(float), s) != hipSuccess) {
        dtc::set_error("gru_bwd: memset failed");
        return DTC_ERR_LAUNCH;
    }
    const unsigned grid = (unsigned)dtc::ceil_div((int64_t)RH, 256);
    float* part = (float*)workspace;              // [nparts][R][H]: the region dtc_gru_fwd uses for gh
    // four units per thread when every row of every operand starts on a 16-byte boundary (DTC_GRU_GATE_VEC=0: one unit per thread)
    static const bool vec_on = !(getenv("DTC_GRU_GATE_VEC") && atoi(getenv("DTC_GRU_GATE_VEC")) == 0);
    auto a16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec4 = vec_on && H % 4 == 0 && a16(dhs) && a16(dh0) && a16(part) && a16(gates) && a16(hn) && a16(hs_all) && a16(dgi) && a16(dgh_all);
    const unsigned grid4 = (unsigned)dtc::ceil_div((int64_t)RH / 4, 256);
    for (int t = T - 1; t >= 0; --t) {
        float* dgh_t = dgh_all + (size_t)t * R3H;
        {
            dtc::ProfScope prof("gru_gate_bwd", (double)RH * 4.0 * 17, s);
            if (vec4) {
                const GateBwdPtrs gp{dhs + (size_t)t * RH, dh0, t == T - 1 ? (const float*)nullptr : (const float*)part, gates + (size_t)t * R3H,
                                     hn + (size_t)t * RH, hs_all + (size_t)t * RH, dgi + (size_t)t * R3H, dgh_t};
                hipLaunchKernelGGL(gru_gate_bwd4_kernel, dim3(grid4), dim3(256), 0, s, gp, gp, R, H, nparts);
            } else
                hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(grid), dim3(256), 0, s, dhs + (size_t)t * RH, dh0,
                                   t == T - 1 ? (const float*)nullptr : (const float*)part, gates + (size_t)t * R3H,
                                   hn + (size_t)t * RH, hs_all + (size_t)t * RH, dgi + (size_t)t * R3H, dgh_t, R, H, nparts);
        }
        // split path: ONE image of W_hh^T serves all T steps
        int rc = s3 ? dtc_gru_dgrad_parts_s3(dgh_t, wimage, part, (int64_t)RH, R, H, nparts, stream)
                    : dtc_linear_dgrad_split(dgh_t, 3 * H, W_hh, part, H, (int64_t)RH, R, 3 * H, H, nparts, stream);
        if (rc != DTC_OK) return rc;
    }
    hipLaunchKernelGGL(gru_add_parts_kernel, dim3(grid), dim3(256), 0, s, dh0, part, (long long)RH, nparts);
    // dW_hh == NULL: the caller forms the W_hh weight gradient itself from dgh_all (workspace + dtc_gru_dgh_offset: the operand-image
    // trainers pack it with the other operands of their grouped weight-gradient launch)
    if (dW_hh == nullptr) return dtc::check_launch("gru_bwd");
    // the padding slots of the padded trajectory layout have dgh = 0: with the caller's list of valid slots the product skips them
    if (valid_rows && n_valid >= 1024 && n_valid < T * R && dtc_get_gemm_split() && 3ll * H * H >= 128 * 128)
        return dtc_linear_wgrad_rows(dgh_all, 3 * H, (int64_t)T * R, hs_all, H, (int64_t)T * R, valid_rows, dW_hh, db_hh, wg_ws, n_valid, 3 * H, H,
                                     stream);
    const DtcSegMat Hprev = plain(hs_all, H, H, (int64_t)T * R);
    int rc = dtc_linear_wgrad(dgh_all, 3 * H, &Hprev, dW_hh, db_hh, wg_ws, T * R, 3 * H, H, stream);
    if (rc != DTC_OK) return rc;
    return dtc::check_launch("gru_bwd");
}

// ---- several recurrences of ONE shape, one launch per time step (the actor's and the critic's GRU of ActorCriticRecurrent /
// ActorCriticDecoderRecurrent: rsl_rl/rsl_rl/modules/actor_critic_recurrent.py:45-46, 92-116).  A time step of one recurrence is a
// latency-bound launch of ~190-290 workgroups; two of them on two streams overlap by ~20 % (tools/gru_pair_probe.py: 1.12 ms for two
// forward passes against 0.72 for one).  The same two as ONE launch: 1.09 ms -- a launch with twice the workgroups takes 1.5 x as long,
// and in the trainers the merged chain is slower than two chains on two lanes: an option, not the default.  Results
// are bit-identical to the single calls (same kernels, same tiles).  A shape / setting without the split-path step kernels, and
// count == 1: the single calls, one after the other.
int dtc_gru_step_fwd_s3_pair(const float* const* hprev, const void* const* img, const float* const* b_hh, const float* const* gi_t,
                             float* const* hout, float* const* gates_t, float* const* hn_t, int R, int H, void* stream);
int dtc_gru_dgrad_parts_s3_pair(const float* const* dgh_t, const void* const* img, float* const* part, int64_t part_stride, int R, int H,
                                int nparts, void* stream);

extern "C" int dtc_gru_fwd_multi(const DtcGruFwdItem* items, int count, int T, int R, int H, void* stream) {
    DTC_REQUIRE(items != nullptr && count >= 1 && count <= DTC_GRU_MULTI_MAX, "count = %d out of range (1..%d)", count, DTC_GRU_MULTI_MAX);
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    static const bool off = getenv("DTC_GRU_MULTI") && atoi(getenv("DTC_GRU_MULTI")) == 0;
    static const bool unfused = getenv("DTC_GRU_UNFUSED") != nullptr;
    // both recurrences as ONE persistent launch (csrc/gru_seq.hip), one after the other inside it -- it may take every CU
    if (count == 2 && !unfused && dtc_get_gemm_split() && dtc_gru_seq_supported(T, R, H, 1)) {
        const float *gi[2], *h0[2], *W[2], *b[2];
        float *hs[2], *gt[2], *hn[2];
        void* sw[2];
        bool ok = true;
        for (int i = 0; i < 2; ++i) {
            const DtcGruFwdItem& it = items[i];
            DTC_REQUIRE(it.gi && it.h0 && it.W_hh && it.b_hh && it.hs_all && it.gates && it.hn && it.workspace, "item %d: null pointer", i);
            gi[i] = it.gi; h0[i] = it.h0; W[i] = it.W_hh; b[i] = it.b_hh; hs[i] = it.hs_all; gt[i] = it.gates; hn[i] = it.hn;
            sw[i] = gru_seq_slot(it.workspace, T, R, H);
            ok = ok && dtc::aligned16(it.gi) && dtc::aligned16(it.h0) && dtc::aligned16(it.hs_all) && dtc::aligned16(it.gates) &&
                 dtc::aligned16(it.hn) && dtc::aligned16(it.workspace);
        }
        if (ok) return dtc_gru_seq_fwd_pair(gi, h0, W, b, hs, gt, hn, sw, T, R, H, stream);
    }
    const bool pair = count == 2 && !off && !unfused && T >= 4 && gru_s3(H);
    if (!pair) {
        for (int i = 0; i < count; ++i) {
            const DtcGruFwdItem& it = items[i];
            int rc = dtc_gru_fwd(it.gi, it.h0, it.W_hh, it.b_hh, it.hs_all, it.gates, it.hn, it.workspace, T, R, H, stream);
            if (rc != DTC_OK) return rc;
        }
        return DTC_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t RH = (size_t)R * H, R3H = (size_t)R * 3 * H;
    const void* img[2];
    for (int i = 0; i < 2; ++i) {
        const DtcGruFwdItem& it = items[i];
        DTC_REQUIRE(it.gi && it.h0 && it.W_hh && it.b_hh && it.hs_all && it.gates && it.hn && it.workspace, "item %d: null pointer", i);
        if (hipMemcpyAsync(it.hs_all, it.h0, RH * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
            dtc::set_error("gru_fwd_multi: h0 copy failed");
            return DTC_ERR_LAUNCH;
        }
        void* im = gru_image_slot(it.workspace, T, R, H);
        int rc = dtc_gru_s3_image(it.W_hh, im, H, 0, stream);
        if (rc != DTC_OK) return rc;
        img[i] = im;
    }
    for (int t = 0; t < T; ++t) {
        const float *hprev[2], *bhh[2], *gi_t[2];
        float *hout[2], *gates_t[2], *hn_t[2];
        for (int i = 0; i < 2; ++i) {
            const DtcGruFwdItem& it = items[i];
            hprev[i] = it.hs_all + (size_t)t * RH;
            bhh[i] = it.b_hh;
            gi_t[i] = it.gi + (size_t)t * R3H;
            hout[i] = it.hs_all + (size_t)(t + 1) * RH;
            gates_t[i] = it.gates + (size_t)t * R3H;
            hn_t[i] = it.hn + (size_t)t * RH;
        }
        int rc = dtc_gru_step_fwd_s3_pair(hprev, img, bhh, gi_t, hout, gates_t, hn_t, R, H, stream);
        if (rc != DTC_OK) return rc;
    }
    return dtc::check_launch("gru_fwd_multi");
}

// BPTT of `count` recurrences without their W_hh weight gradients (dgh_all of item i at its workspace + dtc_gru_dgh_offset, as
// dtc_gru_bwd with dW_hh = NULL leaves it)
extern "C" int dtc_gru_bwd_multi(const DtcGruBwdItem* items, int count, int T, int R, int H, void* stream) {
    DTC_REQUIRE(items != nullptr && count >= 1 && count <= DTC_GRU_MULTI_MAX, "count = %d out of range (1..%d)", count, DTC_GRU_MULTI_MAX);
    DTC_REQUIRE(T > 0 && R > 0 && H > 0, "bad shape T=%d R=%d H=%d", T, R, H);
    static const bool off = getenv("DTC_GRU_MULTI") && atoi(getenv("DTC_GRU_MULTI")) == 0;
    static const bool vec_on = !(getenv("DTC_GRU_GATE_VEC") && atoi(getenv("DTC_GRU_GATE_VEC")) == 0);
    auto a16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const size_t RH = (size_t)R * H, R3H = (size_t)R * 3 * H;
    bool pair = count == 2 && !off && vec_on && gru_s3(H) && H % 4 == 0;
    for (int i = 0; pair && i < count; ++i) {
        const DtcGruBwdItem& it = items[i];
        DTC_REQUIRE(it.dhs && it.hs_all && it.gates && it.hn && it.W_hh && it.dgi && it.dh0 && it.workspace, "item %d: null pointer", i);
        pair = a16(it.dhs) && a16(it.dh0) && a16(it.workspace) && a16(it.gates) && a16(it.hn) && a16(it.hs_all) && a16(it.dgi) &&
               a16((float*)it.workspace + (size_t)MAX_PARTS * RH);
    }
    if (!pair) {
        for (int i = 0; i < count; ++i) {
            const DtcGruBwdItem& it = items[i];
            int rc = dtc_gru_bwd(it.dhs, it.hs_all, it.gates, it.hn, it.W_hh, it.dgi, nullptr, nullptr, it.dh0, it.workspace, nullptr, 0, T, R, H, stream);
            if (rc != DTC_OK) return rc;
        }
        return DTC_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    const int nparts = gru_parts(H, true);
    const void* img[2];
    float *part[2], *dgh_all[2];
    for (int i = 0; i < 2; ++i) {
        const DtcGruBwdItem& it = items[i];
        part[i] = (float*)it.workspace;
        dgh_all[i] = (float*)it.workspace + (size_t)MAX_PARTS * RH;
        void* im = gru_image_slot(it.workspace, T, R, H);
        int rc = dtc_gru_s3_image(it.W_hh, im, H, 1, stream);
        if (rc != DTC_OK) return rc;
        img[i] = im;
        if (hipMemsetAsync(it.dh0, 0, RH * sizeof(float), s) != hipSuccess) {
            dtc::set_error("gru_bwd_multi: memset failed");
            return DTC_ERR_LAUNCH;
        }
    }
    const unsigned grid = (unsigned)dtc::ceil_div((int64_t)RH, 256), grid4 = (unsigned)dtc::ceil_div((int64_t)RH / 4, 256);
    for (int t = T - 1; t >= 0; --t) {
        GateBwdPtrs gp[2];
        const float* dgh_t[2];
        for (int i = 0; i < 2; ++i) {
            const DtcGruBwdItem& it = items[i];
            dgh_t[i] = dgh_all[i] + (size_t)t * R3H;
            gp[i] = GateBwdPtrs{it.dhs + (size_t)t * RH, it.dh0, t == T - 1 ? (const float*)nullptr : (const float*)part[i], it.gates + (size_t)t * R3H,
                                it.hn + (size_t)t * RH, it.hs_all + (size_t)t * RH, it.dgi + (size_t)t * R3H, dgh_all[i] + (size_t)t * R3H};
        }
        {
            dtc::ProfScope prof("gru_gate_bwd", 2.0 * (double)RH * 4.0 * 17, s);
            hipLaunchKernelGGL(gru_gate_bwd4_kernel, dim3(grid4, 2), dim3(256), 0, s, gp[0], gp[1], R, H, nparts);
        }
        int rc = dtc_gru_dgrad_parts_s3_pair(dgh_t, img, part, (int64_t)RH, R, H, nparts, stream);
        if (rc != DTC_OK) return rc;
    }
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL(gru_add_parts_kernel, dim3(grid), dim3(256), 0, s, items[i].dh0, part[i], (long long)RH, nparts);
    return dtc::check_launch("gru_bwd_multi");
}
