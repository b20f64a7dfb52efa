// Fused Raibert-heuristic + terrain-score + per-leg argmin foothold planner for gfx950.
//
// Replaces legged_gym/envs/base/legged_robot_dtc.py:98-201 (about 45 torch temporaries incl.
// several [N,693,4,3] repeats) by ONE kernel.  HBM-bound: 2772 B heights row + 116 B of state
// in, 208 B out per env (3096 B/env).
//
// Mapping: one 64-lane wavefront per env, 4 envs per 256-thread workgroup.  The 4 height rows
// of a workgroup are one contiguous, 16-byte aligned chunk of 4*P floats (P = nx*ny = 693),
// so the workgroup streams it with float4 loads (16 B/lane, fully coalesced) into LDS; each
// wave then works out of LDS:
//   1. clamp -> mean / unbiased variance by lane-strided partial sums + xor-butterfly;
//   2. CANDIDATE WINDOW (fast path): a grid point can only win the argmin with a "valid" score
//      (< 0.148) if it lies within 0.16 m of the leg's nominal foothold, i.e. inside a <= 7x7 cell
//      patch around it.  Each leg therefore evaluates ONE 8x8 patch (64 lanes = 64 candidates:
//      central-difference slope from LDS neighbours, score, exception mask, distance, total) and
//      reduces it with a 64-lane (value,index) butterfly -- ties resolve to the lowest flat index
//      exactly like torch.topk(k=1, largest=False) on CPU;
//   3. FULL SCAN (exact fallback + debug outputs): if a leg has no valid in-radius candidate
//      (all 693 totals then belong to the sentinel classes 2.0-2.13 / 8.0 / 10 and the argmin is
//      decided among them), or if the caller asks for the [N,693,4] score table, the wave scans
//      all 693 points -- this is the reference's algorithm verbatim.
// Both paths evaluate a point with the same device function, so the window result equals the
// full-scan result whenever the window result is accepted (proof sketch in DESIGN.md 4.1).
//
// Numerics: compiled with -ffp-contract=off; every operation is a single IEEE float32 op in
// the order of oracle/foothold.py + oracle/quat.py, so all outputs match the oracle bit for bit.
// Division by the two grid-spacing constants uses q = fma(fma(-x*y, c, x), y, x*y) with y = 1/c,
// which is bit-identical to IEEE x/c for every |x| in (1e-30, 1e30) (exhaustively verified for
// c = 0.05f and 0.1f) and falls back to the IEEE division below that.
#include "common.hpp"
#include "wave.hpp"

namespace {

constexpr int ENVS_PER_BLOCK = 4;

struct GridParams {
    int nx, ny, P;
    float t_half, k_fb;
    float x[64];
    float y[32];
};

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

// x / c, bit-identical to IEEE division (see header comment); inv = RN(1/c) (20.0f / 10.0f, exact)
__device__ __forceinline__ float div_const(float x, float c, float inv) {
    const float q0 = x * inv;
    const float r = __builtin_fmaf(-q0, c, x);
    float q = __builtin_fmaf(r, inv, q0);
    // tiny NON-ZERO numerators (never produced by heights quantised to 5 mm) take the IEEE division; the test is a
    // wave-level vote so that the ~13-instruction division is branched over, not if-converted into every call.
    // x = +-0 stays on the fast path: it yields +-0 / +0, which every consumer squares.
    const bool tiny = fabsf(x) < 1e-30f && x != 0.0f;
    if (__builtin_amdgcn_ballot_w64(tiny) != 0ull) q = tiny ? x / c : q;
    return q;
}

struct EnvCtx {            // wave-uniform per-env quantities
    float bx, by, bz, zq, wq, mean, edge;
    int nx, ny;
};

struct PointEval {
    float s, hx, hy, slope, rawv;
    bool exc;
};

// terrain score + world xy of grid point i = ix*ny + iy  (legged_robot_dtc.py:127-160)
__device__ __forceinline__ PointEval eval_point(const EnvCtx& c, const float* __restrict__ rawE,
                                                const float* __restrict__ gE, const float* __restrict__ xs,
                                                const float* __restrict__ ys, int i, int ix, int iy) {
    PointEval o;
    const int ny = c.ny;
    const float gc = gE[i];
    // torch.gradient: central difference (g[i+1] - g[i-1]) / 0.1 inside, one-sided (g[1] - g[0]) / 0.05 at the borders
    // -- the same expression with the neighbour indices clamped to the grid, so ONE division per axis (the three-way
    // branch made every lane of a border-touching patch evaluate all three forms)
    const int xp = ix < c.nx - 1 ? ny : 0, xm = ix > 0 ? ny : 0;
    const int yp = iy < ny - 1 ? 1 : 0, ym = iy > 0 ? 1 : 0;
    const bool x_in = (xp != 0) & (xm != 0), y_in = (yp != 0) & (ym != 0);
    const float dx = div_const(gE[i + xp] - gE[i - xm], x_in ? 0.1f : 0.05f, x_in ? 10.0f : 20.0f);
    const float dy = div_const(gE[i + yp] - gE[i - ym], y_in ? 0.1f : 0.05f, y_in ? 10.0f : 20.0f);
    o.slope = sqrtf(dx * dx + dy * dy);
    const float rough = fabsf(gc - c.mean);
    const float s_raw = (0.2f * c.edge + o.slope) + 0.3f * rough;
    o.s = s_raw < 0.1f ? s_raw : 10.0f;
    o.rawv = rawE[i];
    const float rel = o.rawv - c.bz;
    o.exc = (rel > 1.0f) | (rel < -1.0f);
    const float px = xs[ix], py = ys[iy];
    const float t0 = -(c.zq * py) * 2.0f;
    const float t1 = (c.zq * px) * 2.0f;
    o.hx = ((px + c.wq * t0) + (-(c.zq * t1))) + c.bx;
    o.hy = ((py + c.wq * t1) + (c.zq * t0)) + c.by;
    return o;
}

__device__ __forceinline__ float total_score(const PointEval& p, float predx, float predy, float& d_out) {
    const float ddx = predx - p.hx, ddy = predy - p.hy;
    float d = sqrtf(ddx * ddx + ddy * ddy);
    d = d < 0.16f ? d : 10.0f;
    d_out = d;
    float tot = p.s * 0.2f + d * 0.8f;
    return p.exc ? 10.0f : tot;
}

__device__ __forceinline__ void wave_argmin(float& best, int& bidx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (ov < best || (ov == best && oi < bidx)) {
            best = ov;
            bidx = oi;
        }
    }
}

template <bool VEC, bool DEBUG>
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
    EnvCtx c;
    c.nx = nx;
    c.ny = ny;
    c.bx = rs[0];
    c.by = rs[1];
    c.bz = rs[2];
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
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float hx = thigh[nn * 12 + l * 3 + 0] - c.bx;
        const float hy = thigh[nn * 12 + l * 3 + 1] - c.by;
        const float hz = thigh[nn * 12 + l * 3 + 2] - c.bz;
        const float rx = cs * hx + (-sn) * hy;
        const float ry = sn * hx + cs * hy;
        predx[l] = (c.bx + rx) + symx;
        predy[l] = (c.by + ry) + symy;
        predz[l] = (c.bz + hz) + symz;
    }

    // ---- clamp + mean / unbiased variance (legged_robot_dtc.py:127-141)
    float psum = 0.0f;
    for (int i = lane; i < P; i += 64) {
        const float v = rawE[i] - c.bz;
        const float gc = fminf(fmaxf(v, -0.5f), 0.5f);
        gE[i] = gc;
        psum = psum + gc;
    }
    c.mean = wave_sum(psum) / (float)P;
    float qsum = 0.0f;
    for (int i = lane; i < P; i += 64) {
        const float d = gE[i] - c.mean;
        qsum = qsum + d * d;
    }
    const float var = wave_sum(qsum) / (float)(P - 1);
    c.edge = fminf(fmaxf(sqrtf(var), 0.0f), 0.3f);
    __syncthreads();   // gE written by other lanes is read below

    // ---- yaw-only attitude (math.py:8-12)
    float nq = sqrtf(qz * qz + qw * qw);
    nq = fmaxf(nq, 1e-9f);
    c.zq = qz / nq;
    c.wq = qw / nq;

    float best[4];
    int bidx[4], bix[4], biy[4];          // winner per leg: flat index and its (ix, iy)
    bool need_full = DEBUG;
    if (!DEBUG) {
        // ---- fast path: one 8x8 candidate patch per leg
        const int wx = lane >> 3, wy = lane & 7;
        // grid spacing from the coordinate tables (uniform grids; used only to PLACE the patch)
        const float x0 = xs[0], y0 = ys[0];
        const float inv_dx = (float)(nx - 1) / (xs[nx - 1] - x0), inv_dy = (float)(ny - 1) / (ys[ny - 1] - y0);
        const float rad_x = 0.16f * inv_dx + 0.05f, rad_y = 0.16f * inv_dy + 0.05f;   // radius in cells + slack
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            // nominal foothold in the yaw frame of the grid: R(-yaw) * (pred - base)
            const float rx = predx[l] - c.bx, ry = predy[l] - c.by;
            const float cyaw = c.wq * c.wq - c.zq * c.zq, syaw = 2.0f * c.wq * c.zq;
            const float lx = cyaw * rx + syaw * ry, ly = -syaw * rx + cyaw * ry;
            const float u = (lx - x0) * inv_dx, v = (ly - y0) * inv_dy;
            // all in-radius cells have |iu - u| < rad: they fit in 8 consecutive cells iff 2*rad < 7
            const bool fits = (2.0f * rad_x < 7.0f) && (2.0f * rad_y < 7.0f) && (fabsf(u) < 1e6f) && (fabsf(v) < 1e6f) &&
                              ny >= 8;   // flat index must grow with the lane id inside the patch (see below)
            const int sx = (int)floorf(u - rad_x) + 1, sy = (int)floorf(v - rad_y) + 1;
            const int ix = sx + wx, iy = sy + wy;
            float tot = __builtin_inff();
            int ii = 0x7fffffff;
            if (fits && ix >= 0 && ix < nx && iy >= 0 && iy < ny) {
                ii = ix * ny + iy;
                const PointEval p = eval_point(c, rawE, gE, xs, ys, ii, ix, iy);
                float d;
                tot = total_score(p, predx[l], predy[l], d);
            }
            // argmin = DPP min of the totals, then the LOWEST lane holding it (lane = wx*8 + wy and ny >= 8,
            // so the flat index ix*ny + iy grows with the lane id: lowest lane == lowest index == torch's tie rule)
            const float mn = wave_min(tot);
            const unsigned long long hit = __ballot(tot == mn);
            const int src = hit ? __ffsll((long long)hit) - 1 : 0;
            best[l] = mn;
            const int srcu = __builtin_amdgcn_readfirstlane(src);
            bidx[l] = __builtin_amdgcn_readlane(ii, srcu);
            bix[l] = __builtin_amdgcn_readlane(ix, srcu);
            biy[l] = __builtin_amdgcn_readlane(iy, srcu);
            // accept only a "valid" winner: every total < 1 is an in-radius, non-sentinel point, and all
            // such points lie inside the patch, so the patch minimum is then the global minimum
            need_full |= !(mn < 1.0f);
        }
    }

    if (need_full) {      // wave-uniform
        float nbest[4];
        int nidx[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            best[l] = __builtin_inff();
            nbest[l] = __builtin_inff();
            bidx[l] = 0x7fffffff;
            nidx[l] = 0x7fffffff;
        }
        for (int i = lane; i < P; i += 64) {
            const int ix = i / ny, iy = i - ix * ny;
            const PointEval p = eval_point(c, rawE, gE, xs, ys, i, ix, iy);
            if (DEBUG) {
                if (active && slope_out) slope_out[nn * P + i] = p.slope;
                if (active && hw_out) {
                    float* o = hw_out + (nn * P + i) * 3;
                    o[0] = p.hx;
                    o[1] = p.hy;
                    o[2] = p.rawv;
                }
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                float d;
                const float tot = total_score(p, predx[l], predy[l], d);
                if (tot < best[l]) {
                    best[l] = tot;
                    bidx[l] = i;
                }
                if (DEBUG) {
                    if (d < nbest[l]) {
                        nbest[l] = d;
                        nidx[l] = i;
                    }
                    if (active && score_out) score_out[(nn * P + i) * 4 + l] = tot;
                }
            }
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            wave_argmin(best[l], bidx[l]);
            if (bidx[l] == 0x7fffffff) bidx[l] = 0;
            bix[l] = bidx[l] / ny;
            biy[l] = bidx[l] - bix[l] * ny;
            if (DEBUG) {
                wave_argmin(nbest[l], nidx[l]);
                if (nidx[l] == 0x7fffffff) nidx[l] = 0;
                if (active && nom_out && lane == 0) nom_out[nn * 4 + l] = (int64_t)nidx[l];
            }
        }
    }

    // ---- decode (legged_robot_dtc.py:184-201); lane l < 4 writes leg l
    if (active && lane < 4) {
        const int l = lane;
        auto pick_i = [&](const int (&a)[4]) { return l == 0 ? a[0] : (l == 1 ? a[1] : (l == 2 ? a[2] : a[3])); };
        auto pick_f = [&](const float (&a)[4]) { return l == 0 ? a[0] : (l == 1 ? a[1] : (l == 2 ? a[2] : a[3])); };
        const int bi = pick_i(bidx), xi = pick_i(biy), yi = pick_i(bix);      // xi = idx % ny, yi = idx / ny
        const float pxl = pick_f(predx), pyl = pick_f(predy), pzl = pick_f(predz);
        idx_out[nn * 4 + l] = (int64_t)bi;
        // the reference gathers the x table with the y-index and vice versa (sic); xi < ny, yi < nx
        obs_out[nn * 8 + l] = xs[xi < nx ? xi : xi % nx];
        obs_out[nn * 8 + 4 + l] = ys[yi < ny ? yi : yi % ny];
        const float px = xs[yi], py = ys[xi];
        const float t0 = -(c.zq * py) * 2.0f;
        const float t1 = (c.zq * px) * 2.0f;
        world_out[nn * 12 + l * 3 + 0] = ((px + c.wq * t0) + (-(c.zq * t1))) + c.bx;
        world_out[nn * 12 + l * 3 + 1] = ((py + c.wq * t1) + (c.zq * t0)) + c.by;
        world_out[nn * 12 + l * 3 + 2] = rawE[bi];
        pred_out[nn * 12 + l * 3 + 0] = pxl;
        pred_out[nn * 12 + l * 3 + 1] = pyl;
        pred_out[nn * 12 + l * 3 + 2] = pzl;
        const V3 pr = quat_rotate_inverse(qx, qy, qz, qw, V3{pxl - c.bx, pyl - c.by, pzl - c.bz});
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

// legged_robot_dtc.py:577-586 / :536-539 -- one thread per env
__global__ __launch_bounds__(256) void foothold_rewards_kernel(const float* __restrict__ foot, const float* __restrict__ opt,
                                                               const uint8_t* __restrict__ contact, float* __restrict__ tracking,
                                                               float* __restrict__ miss, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float sum = 0.0f, minz = __builtin_inff();
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float* f = foot + (long long)n * 12 + l * 3;
        const float* o = opt + (long long)n * 12 + l * 3;
        const float dx = f[0] - o[0], dy = f[1] - o[1];
        const float dis = sqrtf(dx * dx + dy * dy);
        const float r = -logf(0.8f + dis);
        sum = sum + (contact[n * 4 + l] ? r : 0.0f);
        minz = fminf(minz, f[2]);
    }
    if (tracking) tracking[n] = sum;
    if (miss) miss[n] = minz < 0.0f ? 1.0f : 0.0f;
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

template <bool VEC, bool DEBUG>
void launch(int grid, size_t lds, hipStream_t s, const float* mh, const float* rs, const float* th, const float* cmd,
            const GridParams& gp, int64_t* idx, float* obs, float* world, float* pred, float* p2r, float* score,
            int64_t* nom, float* slope, float* hw, int N) {
    hipLaunchKernelGGL((foothold_plan_kernel<VEC, DEBUG>), dim3(grid), dim3(256), lds, s, mh, rs, th, cmd, gp, idx, obs,
                       world, pred, p2r, score, nom, slope, hw, N);
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
    const bool debug = score_or_null || nominal_idx_or_null || slope_or_null || heights_world_or_null;
    const bool vec = dtc::aligned16(measured_heights);
    dtc::ProfScope prof(debug ? "foothold_plan_debug" : "foothold_plan", bytes, s);
#define DTC_FH_ARGS grid, lds, s, measured_heights, root_states, thigh_pos, commands, gp, idx, foothold_obs, opt_world, \
                    pred, pred_to_robot, score_or_null, nominal_idx_or_null, slope_or_null, heights_world_or_null, N
    if (vec && debug) launch<true, true>(DTC_FH_ARGS);
    else if (vec) launch<true, false>(DTC_FH_ARGS);
    else if (debug) launch<false, true>(DTC_FH_ARGS);
    else launch<false, false>(DTC_FH_ARGS);
#undef DTC_FH_ARGS
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

extern "C" int dtc_foothold_rewards(const float* foot_positions, const float* opt_world, const uint8_t* contact,
                                    float* tracking, float* miss, int N, void* stream) {
    DTC_REQUIRE(N >= 0, "N < 0");
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(foot_positions && opt_world && contact && (tracking || miss), "null pointer");
    hipLaunchKernelGGL(foothold_rewards_kernel, dim3((unsigned)dtc::ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       foot_positions, opt_world, contact, tracking, miss, N);
    return dtc::check_launch("foothold_rewards");
}
