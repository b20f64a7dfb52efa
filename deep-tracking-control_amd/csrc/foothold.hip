// Fused Raibert-heuristic + terrain-score + per-leg argmin foothold planner for gfx950.
//
// Replaces legged_gym/envs/base/legged_robot_dtc.py:98-201 (about 45 torch temporaries incl.
// several [N,693,4,3] repeats) by ONE kernel.  HBM-bound: 2772 B heights row + 116 B of state
// in, 208 B out per env (3096 B/env).
//
// Mapping: one 64-lane wavefront per env, 4 envs per 256-thread workgroup.  The 4 height rows
// of a workgroup are one contiguous, 16-byte aligned chunk of 4*P floats (P = nx*ny = 693),
// so the workgroup streams it with float4 loads (16 B/lane, fully coalesced) into LDS; each
// wave then works out of LDS: clamp -> mean/var by lane-strided partial sums + xor-butterfly,
// central-difference slope from LDS neighbours, distance to the 4 nominal footholds, running
// (min,index) per leg in registers, and a 64-lane (value,index) butterfly for the argmin with
// ties resolved to the lowest index (torch.topk(k=1, largest=False) on CPU).
//
// Numerics: compiled with -ffp-contract=off; every operation is a single IEEE float32 op in
// the order of oracle/foothold.py + oracle/quat.py, so all outputs match the oracle bit for bit.
#include "common.hpp"

namespace {

constexpr int ENVS_PER_BLOCK = 4;

struct GridParams {
    int nx, ny, P;
    float t_half, k_fb;
    float x[64];
    float y[32];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// Cody-Waite reduction + minimax polynomials; constants/order == oracle/quat.py:sincos
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs) {
    const float k = rintf(x * 0.636619772367581343f);
    float r = ((x - k * 1.5703125f) - k * 4.837512969970703125e-4f) - k * 7.54978995489188216e-8f;
    const float r2 = r * r;
    const float s = r + (r * r2) * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
    const float c = (1.0f - 0.5f * r2) +
                    (r2 * r2) * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    const int q = ((int)k) & 3;
    sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 quat_rotate_inverse(float qx, float qy, float qz, float qw, V3 v) {
    const float s = 2.0f * (qw * qw) - 1.0f;
    const float ax = v.x * s, ay = v.y * s, az = v.z * s;
    const float cx = qy * v.z - qz * v.y;
    const float cy = qz * v.x - qx * v.z;
    const float cz = qx * v.y - qy * v.x;
    const float bx = (cx * qw) * 2.0f, by = (cy * qw) * 2.0f, bz = (cz * qw) * 2.0f;
    const float dot = (qx * v.x + qy * v.y) + qz * v.z;
    const float ex = (qx * dot) * 2.0f, ey = (qy * dot) * 2.0f, ez = (qz * dot) * 2.0f;
    return V3{(ax - bx) + ex, (ay - by) + ey, (az - bz) + ez};
}

template <bool VEC>
__global__ __launch_bounds__(256) void foothold_plan_kernel(
    const float* __restrict__ mh, const float* __restrict__ root, const float* __restrict__ thigh,
    const float* __restrict__ cmd, const GridParams gp, int64_t* __restrict__ idx_out,
    float* __restrict__ obs_out, float* __restrict__ world_out, float* __restrict__ pred_out,
    float* __restrict__ p2r_out, float* __restrict__ score_out, int64_t* __restrict__ nom_out,
    float* __restrict__ slope_out, float* __restrict__ hw_out, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = gp.P, nx = gp.nx, ny = gp.ny;
    const int P4 = (ENVS_PER_BLOCK * P + 3) & ~3;
    float* raw = smem;             // [4][P] measured heights of the block's envs
    float* gbuf = smem + P4;       // [4][P] clamped, base-relative heights
    float* xs = gbuf + P4;         // [64]
    float* ys = xs + 64;           // [32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t env0 = (int64_t)blockIdx.x * ENVS_PER_BLOCK;
    const int64_t base_f = env0 * P;
    const int64_t remain = (int64_t)N * P - base_f;
    const int nfl = (int)(remain < (int64_t)ENVS_PER_BLOCK * P ? remain : (int64_t)ENVS_PER_BLOCK * P);

    if (tid < 64) xs[tid] = gp.x[tid];
    else if (tid < 96) ys[tid - 64] = gp.y[tid - 64];
    if (VEC) {
        const float4* src = reinterpret_cast<const float4*>(mh + base_f);
        for (int f = tid; f * 4 < nfl; f += 256) {
            if (f * 4 + 3 < nfl) {
                reinterpret_cast<float4*>(raw)[f] = src[f];
            } else {
                for (int e = f * 4; e < nfl; ++e) raw[e] = mh[base_f + e];
            }
        }
    } else {
        for (int e = tid; e < nfl; e += 256) raw[e] = mh[base_f + e];
    }
    __syncthreads();

    const int64_t n = env0 + wave;
    const bool active = n < N;                      // wave-uniform
    const int64_t nn = active ? n : 0;
    const float* rawE = raw + wave * P;
    float* gE = gbuf + wave * P;

    // ---- per-env state (same address in every lane -> broadcast loads)
    const float* rs = root + nn * 13;
    const float bx = rs[0], by = rs[1], bz = rs[2];
    const float qx = rs[3], qy = rs[4], qz = rs[5], qw = rs[6];
    const V3 vw{rs[7], rs[8], rs[9]};
    const float c0 = cmd[nn * 4 + 0], c1 = cmd[nn * 4 + 1], c2 = cmd[nn * 4 + 2];

    // ---- Raibert nominal footholds (legged_robot_dtc.py:100-120)
    const V3 vb = quat_rotate_inverse(qx, qy, qz, qw, vw);
    float sn, cs;
    sincos_cw(c2, sn, cs);
    const float symx = gp.t_half * vb.x + gp.k_fb * (vb.x - c0);
    const float symy = gp.t_half * vb.y + gp.k_fb * (vb.y - c1);
    const float symz = gp.t_half * vb.z + gp.k_fb * (vb.z - 0.0f);
    float predx[4], predy[4], predz[4];
    V3 p2r[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float hx = thigh[nn * 12 + l * 3 + 0] - bx;
        const float hy = thigh[nn * 12 + l * 3 + 1] - by;
        const float hz = thigh[nn * 12 + l * 3 + 2] - bz;
        const float rx = cs * hx + (-sn) * hy;
        const float ry = sn * hx + cs * hy;
        predx[l] = (bx + rx) + symx;
        predy[l] = (by + ry) + symy;
        predz[l] = (bz + hz) + symz;
        p2r[l] = quat_rotate_inverse(qx, qy, qz, qw, V3{predx[l] - bx, predy[l] - by, predz[l] - bz});
    }

    // ---- clamp + mean / unbiased variance (legged_robot_dtc.py:127-141)
    float psum = 0.0f;
    for (int i = lane; i < P; i += 64) {
        const float v = rawE[i] - bz;
        const float gc = fminf(fmaxf(v, -0.5f), 0.5f);
        gE[i] = gc;
        psum = psum + gc;
    }
    const float mean = wave_sum(psum) / (float)P;
    float qsum = 0.0f;
    for (int i = lane; i < P; i += 64) {
        const float d = gE[i] - mean;
        qsum = qsum + d * d;
    }
    const float var = wave_sum(qsum) / (float)(P - 1);
    const float edge = fminf(fmaxf(sqrtf(var), 0.0f), 0.3f);
    __syncthreads();   // gE written by other lanes is read below

    // ---- yaw-only attitude (math.py:8-12)
    float nq = sqrtf(qz * qz + qw * qw);
    nq = fmaxf(nq, 1e-9f);
    const float zq = qz / nq, wq = qw / nq;

    float best[4], nbest[4];
    int bidx[4], nidx[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        best[l] = __builtin_inff();
        nbest[l] = __builtin_inff();
        bidx[l] = 0x7fffffff;
        nidx[l] = 0x7fffffff;
    }
    const bool want_nom = nom_out != nullptr;

    for (int i = lane; i < P; i += 64) {
        const int ix = i / ny, iy = i - ix * ny;
        const float gc = gE[i];
        float dx, dy;
        if (ix == 0) dx = (gE[i + ny] - gc) / 0.05f;
        else if (ix == nx - 1) dx = (gc - gE[i - ny]) / 0.05f;
        else dx = (gE[i + ny] - gE[i - ny]) / 0.1f;
        if (iy == 0) dy = (gE[i + 1] - gc) / 0.05f;
        else if (iy == ny - 1) dy = (gc - gE[i - 1]) / 0.05f;
        else dy = (gE[i + 1] - gE[i - 1]) / 0.1f;
        const float slope = sqrtf(dx * dx + dy * dy);
        const float rough = fabsf(gc - mean);
        const float s_raw = (0.2f * edge + slope) + 0.3f * rough;
        const float s = s_raw < 0.1f ? s_raw : 10.0f;
        const float rawv = rawE[i];
        const float rel = rawv - bz;
        const bool exc = (rel > 1.0f) | (rel < -1.0f);
        const float px = xs[ix], py = ys[iy];
        const float t0 = -(zq * py) * 2.0f;
        const float t1 = (zq * px) * 2.0f;
        const float hx = ((px + wq * t0) + (-(zq * t1))) + bx;
        const float hy = ((py + wq * t1) + (zq * t0)) + by;
        if (active && slope_out) slope_out[nn * P + i] = slope;
        if (active && hw_out) {
            float* o = hw_out + (nn * P + i) * 3;
            o[0] = hx;
            o[1] = hy;
            o[2] = rawv;
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float ddx = predx[l] - hx, ddy = predy[l] - hy;
            float d = sqrtf(ddx * ddx + ddy * ddy);
            d = d < 0.16f ? d : 10.0f;
            float tot = s * 0.2f + d * 0.8f;
            tot = exc ? 10.0f : tot;
            if (tot < best[l]) {
                best[l] = tot;
                bidx[l] = i;
            }
            if (want_nom && d < nbest[l]) {
                nbest[l] = d;
                nidx[l] = i;
            }
            if (active && score_out) score_out[(nn * P + i) * 4 + l] = tot;
        }
    }

    // ---- 64-lane (value, index) butterfly: min value, ties -> lowest index
#pragma unroll
    for (int l = 0; l < 4; ++l) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best[l], off, 64);
            const int oi = __shfl_xor(bidx[l], off, 64);
            if (ov < best[l] || (ov == best[l] && oi < bidx[l])) {
                best[l] = ov;
                bidx[l] = oi;
            }
        }
        if (bidx[l] == 0x7fffffff) bidx[l] = 0;
        if (want_nom) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(nbest[l], off, 64);
                const int oi = __shfl_xor(nidx[l], off, 64);
                if (ov < nbest[l] || (ov == nbest[l] && oi < nidx[l])) {
                    nbest[l] = ov;
                    nidx[l] = oi;
                }
            }
            if (nidx[l] == 0x7fffffff) nidx[l] = 0;
        }
    }

    // ---- decode (legged_robot_dtc.py:184-201); lane l < 4 writes leg l
    if (active && lane < 4) {
        const int l = lane;
        const int bi = l == 0 ? bidx[0] : (l == 1 ? bidx[1] : (l == 2 ? bidx[2] : bidx[3]));
        const float pxl = l == 0 ? predx[0] : (l == 1 ? predx[1] : (l == 2 ? predx[2] : predx[3]));
        const float pyl = l == 0 ? predy[0] : (l == 1 ? predy[1] : (l == 2 ? predy[2] : predy[3]));
        const float pzl = l == 0 ? predz[0] : (l == 1 ? predz[1] : (l == 2 ? predz[2] : predz[3]));
        const V3 pr = l == 0 ? p2r[0] : (l == 1 ? p2r[1] : (l == 2 ? p2r[2] : p2r[3]));
        idx_out[nn * 4 + l] = (int64_t)bi;
        if (want_nom) {
            const int ni = l == 0 ? nidx[0] : (l == 1 ? nidx[1] : (l == 2 ? nidx[2] : nidx[3]));
            nom_out[nn * 4 + l] = (int64_t)ni;
        }
        const int xi = bi % ny, yi = bi / ny;
        // the reference gathers the x table with the y-index and vice versa (sic)
        obs_out[nn * 8 + l] = xs[xi % nx];
        obs_out[nn * 8 + 4 + l] = ys[yi % ny];
        const float px = xs[yi], py = ys[xi];
        const float t0 = -(zq * py) * 2.0f;
        const float t1 = (zq * px) * 2.0f;
        world_out[nn * 12 + l * 3 + 0] = ((px + wq * t0) + (-(zq * t1))) + bx;
        world_out[nn * 12 + l * 3 + 1] = ((py + wq * t1) + (zq * t0)) + by;
        world_out[nn * 12 + l * 3 + 2] = rawE[bi];
        pred_out[nn * 12 + l * 3 + 0] = pxl;
        pred_out[nn * 12 + l * 3 + 1] = pyl;
        pred_out[nn * 12 + l * 3 + 2] = pzl;
        p2r_out[nn * 12 + l * 3 + 0] = pr.x;
        p2r_out[nn * 12 + l * 3 + 1] = pr.y;
        p2r_out[nn * 12 + l * 3 + 2] = pr.z;
    }
}

// LeggedRobot._get_heights (legged_robot.py:1279-1317): one thread per (env, grid point).
__global__ __launch_bounds__(256) void get_heights_kernel(const int16_t* __restrict__ hs, int rows, int cols,
                                                          const float* __restrict__ root, const GridParams gp,
                                                          float border, float hscale, float vscale,
                                                          float* __restrict__ out, int N) {
    __shared__ float xs[64], ys[32];
    if (threadIdx.x < 64) xs[threadIdx.x] = gp.x[threadIdx.x];
    else if (threadIdx.x < 96) ys[threadIdx.x - 64] = gp.y[threadIdx.x - 64];
    __syncthreads();
    const int P = gp.P;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)N * P) return;
    const int64_t n = e / P;
    const int i = (int)(e - n * P);
    const int ix = i / gp.ny, iy = i - ix * gp.ny;
    const float* rs = root + n * 13;
    const float qz = rs[5], qw = rs[6];
    float nq = sqrtf(qz * qz + qw * qw);
    nq = fmaxf(nq, 1e-9f);
    const float zq = qz / nq, wq = qw / nq;
    const float px = xs[ix], py = ys[iy];
    const float t0 = -(zq * py) * 2.0f;
    const float t1 = (zq * px) * 2.0f;
    const float wx = ((((px + wq * t0) + (-(zq * t1))) + rs[0]) + border) / hscale;
    const float wy = ((((py + wq * t1) + (zq * t0)) + rs[1]) + border) / hscale;
    long long cx = (long long)wx, cy = (long long)wy;   // .long(): truncation toward zero
    cx = cx < 0 ? 0 : (cx > rows - 2 ? rows - 2 : cx);
    cy = cy < 0 ? 0 : (cy > cols - 2 ? cols - 2 : cy);
    const int16_t h1 = hs[cx * cols + cy], h2 = hs[(cx + 1) * cols + cy], h3 = hs[cx * cols + cy + 1];
    int16_t h = h1 < h2 ? h1 : h2;
    h = h < h3 ? h : h3;
    out[e] = (float)h * vscale;
}

int make_params(const DtcGridCfg* cfg, GridParams& gp) {
    DTC_REQUIRE(cfg != nullptr, "grid cfg is null");
    DTC_REQUIRE(cfg->nx >= 2 && cfg->nx <= 64 && cfg->ny >= 2 && cfg->ny <= 32, "grid %dx%d unsupported (nx<=64, ny<=32)",
                cfg->nx, cfg->ny);
    gp.nx = cfg->nx;
    gp.ny = cfg->ny;
    gp.P = cfg->nx * cfg->ny;
    gp.t_half = (float)((double)cfg->t_stance * 0.5);
    gp.k_fb = cfg->fdbk_gain;
    for (int i = 0; i < 64; ++i) gp.x[i] = i < cfg->nx ? cfg->x[i] : 0.f;
    for (int i = 0; i < 32; ++i) gp.y[i] = i < cfg->ny ? cfg->y[i] : 0.f;
    return DTC_OK;
}

}  // namespace

extern "C" int dtc_foothold_plan(const float* measured_heights, const float* root_states, const float* thigh_pos,
                                 const float* commands, const DtcGridCfg* cfg, int64_t* idx, float* foothold_obs,
                                 float* opt_world, float* pred, float* pred_to_robot, float* score_or_null,
                                 int64_t* nominal_idx_or_null, float* slope_or_null, float* heights_world_or_null,
                                 int N, void* stream) {
    GridParams gp;
    int rc = make_params(cfg, gp);
    if (rc != DTC_OK) return rc;
    DTC_REQUIRE(N >= 0, "N < 0");
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(measured_heights && root_states && thigh_pos && commands, "null input");
    DTC_REQUIRE(idx && foothold_obs && opt_world && pred && pred_to_robot, "null output");
    hipStream_t s = (hipStream_t)stream;
    const int P4 = (ENVS_PER_BLOCK * gp.P + 3) & ~3;
    const size_t lds = (size_t)(2 * P4 + 96) * sizeof(float);
    const int grid = (int)dtc::ceil_div(N, ENVS_PER_BLOCK);
    const double bytes = (double)N * (gp.P * 4.0 + 13 * 4 + 4 * 4 + 12 * 4 + 32 + 32 + 48 + 96);
    dtc::ProfScope prof("foothold_plan", bytes, s);
    if (dtc::aligned16(measured_heights)) {
        hipLaunchKernelGGL(foothold_plan_kernel<true>, dim3(grid), dim3(256), lds, s, measured_heights, root_states,
                           thigh_pos, commands, gp, idx, foothold_obs, opt_world, pred, pred_to_robot, score_or_null,
                           nominal_idx_or_null, slope_or_null, heights_world_or_null, N);
    } else {
        hipLaunchKernelGGL(foothold_plan_kernel<false>, dim3(grid), dim3(256), lds, s, measured_heights, root_states,
                           thigh_pos, commands, gp, idx, foothold_obs, opt_world, pred, pred_to_robot, score_or_null,
                           nominal_idx_or_null, slope_or_null, heights_world_or_null, N);
    }
    return dtc::check_launch("foothold_plan");
}

extern "C" int dtc_get_heights(const int16_t* height_samples, int rows, int cols, const float* root_states,
                               const DtcGridCfg* cfg, float border_size, float horizontal_scale, float vertical_scale,
                               float* measured_heights, int N, void* stream) {
    GridParams gp;
    int rc = make_params(cfg, gp);
    if (rc != DTC_OK) return rc;
    DTC_REQUIRE(N >= 0 && rows >= 2 && cols >= 2, "bad shape");
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(height_samples && root_states && measured_heights, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)N * gp.P;
    dtc::ProfScope prof("get_heights", (double)total * 10.0, s);
    hipLaunchKernelGGL(get_heights_kernel, dim3((unsigned)dtc::ceil_div(total, 256)), dim3(256), 0, s, height_samples,
                       rows, cols, root_states, gp, border_size, horizontal_scale, vertical_scale, measured_heights, N);
    return dtc::check_launch("get_heights");
}
