// Fused Raibert-heuristic + terrain-score + per-leg argmin foothold planner for gfx950.
//
// Replaces legged_gym/envs/base/legged_robot_dtc.py:98-201 (about 45 torch temporaries incl.
// several [N,693,4,3] repeats) by ONE kernel.  HBM-bound: 2772 B heights row + 116 B of state
// in, 208 B out per env (3096 B/env).
//
// Two kernels share the per-point arithmetic:
//   * foothold_plan_fast_kernel  -- the reference's 33 x 21 grid without debug outputs (the hot path; see its header);
//   * foothold_plan_kernel       -- any grid up to 64 x 32 and the debug outputs (score table, nominal index, slope).
//
// Generic kernel: one 64-lane wavefront per env, 4 envs per 256-thread workgroup.  The 4 height rows
// of a workgroup are one contiguous, 16-byte aligned chunk of 4*P floats (P = nx*ny = 693),
// so the workgroup streams it with float4 loads (16 B/lane, fully coalesced) into LDS; each
// wave then works out of LDS:
//   1. clamp -> mean / unbiased variance: per-lane partial sums in the canonical order (below) + xor-butterfly;
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
// c = 0.05f and 0.1f).  Below 1e-30 the quotient may differ in its low bits, but the ONLY consumer of the
// two quotients is dx*dx + dy*dy, and any |d| < 2^-75 (2.6e-23) squares to +0 in float32 either way -- so the
// slope, and everything after it, is still bit-identical (round 1 branched to an IEEE division there).
//
// Canonical reduction order (mean / variance of the P clamped heights; oracle/foothold.py:wave_sum states the same):
// lane k owns the float4 chunks q = k + 64 j (elements 4q .. 4q+3), keeps TWO running sums -- elements 0, 2 of every
// chunk in one, elements 1, 3 in the other, chunks in ascending j (that is what a packed v_pk_add_f32 does) -- adds
// the two, and the 64 lane sums are folded by the xor-butterfly of wave_sum().
#include "common.hpp"
#include "wave.hpp"

namespace {

constexpr int ENVS_PER_BLOCK = 4;

struct GridParams {
    int nx, ny, P;
    float t_half, k_fb;
    // placement of the 8 x 8 candidate patch (uniform grids; used only to PLACE the patch): origin, cells per metre,
    // search radius in cells + slack, and whether every in-radius cell fits into 8 consecutive cells
    float x0, y0, inv_dx, inv_dy, rad_x, rad_y;
    int patch_ok;
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

// x / c, bit-identical to IEEE division wherever it matters (see header comment); inv = RN(1/c) (20.0f / 10.0f, exact)
__device__ __forceinline__ float div_const(float x, float c, float inv) {
    const float q0 = x * inv;
    const float r = __builtin_fmaf(-q0, c, x);
    return __builtin_fmaf(r, inv, q0);
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

    // ---- clamp + mean / unbiased variance (legged_robot_dtc.py:127-141), canonical order (header)
    float se = 0.0f, so = 0.0f;
    for (int q = lane; 4 * q < P; q += 64) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int i = 4 * q + cc;
            float gc = 0.0f;
            if (i < P) {
                gc = fminf(fmaxf(rawE[i] - c.bz, -0.5f), 0.5f);
                gE[i] = gc;
            }
            if (cc & 1) so = so + gc;
            else se = se + gc;
        }
    }
    c.mean = wave_sum(se + so) / (float)P;
    se = 0.0f;
    so = 0.0f;
    for (int q = lane; 4 * q < P; q += 64) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int i = 4 * q + cc;
            float dd = 0.0f;
            if (i < P) {
                const float d = gE[i] - c.mean;
                dd = d * d;
            }
            if (cc & 1) so = so + dd;
            else se = se + dd;
        }
    }
    const float var = wave_sum(se + so) / (float)(P - 1);
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
        const float x0 = gp.x0, y0 = gp.y0, inv_dx = gp.inv_dx, inv_dy = gp.inv_dy, rad_x = gp.rad_x, rad_y = gp.rad_y;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            // nominal foothold in the yaw frame of the grid: R(-yaw) * (pred - base)
            const float rx = predx[l] - c.bx, ry = predy[l] - c.by;
            const float cyaw = c.wq * c.wq - c.zq * c.zq, syaw = 2.0f * c.wq * c.zq;
            const float lx = cyaw * rx + syaw * ry, ly = -syaw * rx + cyaw * ry;
            const float u = (lx - x0) * inv_dx, v = (ly - y0) * inv_dy;
            // all in-radius cells have |iu - u| < rad: they fit in 8 consecutive cells iff 2*rad < 7
            const bool fits = gp.patch_ok && (fabsf(u) < 1e6f) && (fabsf(v) < 1e6f) &&
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

// ---------------------------------------------------------------------------------------------------------------------
// Fast planner for the reference's grid (NX x NY = 33 x 21, no debug outputs).  Same results as the generic kernel, bit
// for bit; built around the instruction count, because the planner is VALU-bound (round 1: ~1500 vector instructions per
// env, 182 us for 98 304 envs against a 40 us HBM floor):
//   * persistent launch: the grid is what fits on the chip at once; every wave owns a contiguous range of envs and
//     walks it in groups of four;
//   * per-env scalar algebra (Raibert heuristic, quaternion rotations, sin/cos, yaw quaternion, patch placement, the
//     pred / pred_to_robot outputs) runs ONCE per group with lane = (env of the group, leg) instead of once per env with all
//     64 lanes doing the same thing; the per-env values travel to the scoring loop through v_readlane;
//   * the height row comes in with three unaligned 16-byte buffer loads per lane (row-bounded descriptor: the tail reads
//     0), is clamped and reduced from registers with packed fp32 adds / multiplies, and is written to LDS once; the next
//     env's row is requested before the current one is scored;
//   * a leg's 8 x 8 candidate patch that lies strictly inside the grid (the usual case) takes a branch-free path: fixed
//     LDS offsets for the four neighbours, the constant 0.1 divisor, both axes in packed arithmetic;
//   * the decode (index, observation, world position) runs once per group on 16 lanes.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// min over the wave of NON-NEGATIVE floats (or +inf / NaN): their bit patterns order like unsigned integers, and
// v_min_u32 takes the DPP operand directly (fminf costs a canonicalising v_max per step)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float wave_min_nonneg(float f) {
    unsigned v = (unsigned)__float_as_int(f);
    v = min(v, dpp_u<0xB1, 0xF>(v));
    v = min(v, dpp_u<0x4E, 0xF>(v));
    v = min(v, dpp_u<0x141, 0xF>(v));
    v = min(v, dpp_u<0x140, 0xF>(v));
    v = min(v, dpp_u<0x142, 0xA>(v));
    v = min(v, dpp_u<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ float rlf(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// x / C for the two point-count constants: the same two-fma quotient as div_const (exact for 1e-30 <= |x| <= 1e30, all
// four constants checked over all 2^32 inputs by tools/probes/div_const_exact.hip); x is wave-uniform here, so the range
// test is a scalar branch to the IEEE division
template <int C>
__device__ __forceinline__ float div_count(float x) {
    constexpr float c = (float)C, inv = 1.0f / (float)C;
    const float ax = fabsf(x);
    if (__builtin_amdgcn_readfirstlane(ax >= 1e-30f && ax <= 1e30f)) return div_const(x, c, inv);
    return x / c;
}

// xor-butterfly sum with the partner order of wave_sum() (1, 2, 4, 8 by DPP) and the two cross-row steps by gfx950's
// v_permlane16_swap / v_permlane32_swap (each lane adds the same two partial sums wave_sum() adds: bit-identical);
// every lane returns the total
__device__ __forceinline__ float wave_sum_all(float v) {
    v = v + dpp_f<0xB1, 0xF>(0.f, v);
    v = v + dpp_f<0x4E, 0xF>(0.f, v);
    v = v + dpp_f<0x141, 0xF>(0.f, v);
    v = v + dpp_f<0x140, 0xF>(0.f, v);
    u2 r = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(v), (unsigned)__float_as_int(v), false, false);
    v = __int_as_float((int)r.x) + __int_as_float((int)r.y);
    r = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(v), (unsigned)__float_as_int(v), false, false);
    return __int_as_float((int)r.x) + __int_as_float((int)r.y);
}

#ifndef DTC_FH_ABL
#define DTC_FH_ABL 0        // timing ablation of the scoring passes (tools/jobs/r6_planner_bound.sh); 0 = the product
#endif
#ifndef DTC_FH_WAVES
#define DTC_FH_WAVES 5      // waves per SIMD the register allocator aims at (tuning aid)
#endif
constexpr int FAST_ENVS_PER_WAVE = 4, FAST_ENVS_PER_BLOCK = 16;

// terrain table of LeggedRobot._get_heights (legged_robot.py:1279-1317) for the fused variant (TABLE = true): the kernel
// samples the int16 height field itself, writes the measured heights out (the observations need them) and plans from
// the registers it just filled -- one launch and no read-back of the [N, P] matrix
struct TableParams {
    const int16_t* hs;
    int rows, cols;
    float border, hscale, vscale;
};

// the rest of one env step (STEP = true): what LeggedRobotDTC.post_physics_step does with the SAME per-env data right after the
// foothold block -- check_termination (legged_robot_dtc.py:229-248), the two foothold rewards (:577-586, :536-539) and
// compute_observations (:255-288) -- appended to the planner's per-env loop while the env's height row is still in LDS.  Every
// value is produced by the operations, lane assignment and reduction order of the stand-alone kernels (check_termination_kernel,
// foothold_rewards_kernel, env_observations_kernel), so the outputs are the same bits (tests/test_hip_envstep.py).
struct StepArgs {
    // check_termination
    const float* contact_forces;
    const int* term_idx;
    const long long* episode_length;
    const float* gravity;
    unsigned char *reset_buf, *time_out_buf;
    float* height_mean;
    long long max_episode_length;
    int num_bodies, n_term;
    // rewards
    const float* foot_positions;
    const unsigned char* contact_filt;
    float *rew_tracking, *rew_miss;
    // compute_observations
    const float *ang_vel, *dof_pos, *default_dof_pos, *dof_vel, *actions, *forces, *noise_offset, *u_obs, *noise_scale, *u_heights;
    long long ld_forces;
    float *obs, *priv, *heights;
    DtcObsCfg cfg;
};

template <int NX, int NY, bool TABLE, bool STEP = false>
__global__ __launch_bounds__(256, (TABLE || STEP) ? 3 : DTC_FH_WAVES) void foothold_plan_fast_kernel(
    float* __restrict__ mh, const float* __restrict__ root, const float* __restrict__ thigh,
    const float* __restrict__ cmd, const GridParams gp, const TableParams tp, int64_t* __restrict__ idx_out,
    float* __restrict__ obs_out, float* __restrict__ world_out, float* __restrict__ pred_out, float* __restrict__ p2r_out,
    int N, const StepArgs sa) {
    constexpr int P = NX * NY, NCH = (P + 255) / 256, RS = (P + 3) & ~3;
    constexpr int PX = (NX + 7) / 8, PY = (NY + 7) / 8;                // 8 x 8 patches that tile the grid (fallback)
    static_assert(NY >= 10 && NX >= 10, "the flat index must grow with the lane id inside a patch");
    __shared__ __attribute__((aligned(16))) float rows[4][2][RS];      // per wave: clamped heights | measured heights
    __shared__ float xs[64], ys[32];
    typedef __attribute__((address_space(3))) float lds_f;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 64) xs[tid] = gp.x[tid];
    else if (tid < 96) ys[tid - 64] = gp.y[tid - 64];
    __syncthreads();
    float* gE = rows[wave][0];
    float* rawE = rows[wave][1];

    // the four envs of this wave
    const int g0 = (blockIdx.x * 4 + wave) * FAST_ENVS_PER_WAVE;
    const int cnt = N - g0 < FAST_ENVS_PER_WAVE ? N - g0 : FAST_ENVS_PER_WAVE;
    if (cnt <= 0) return;

    const int wx = lane >> 3, wy = lane & 7, L0 = wx * NY + wy;
    const unsigned g_lds = (unsigned)(unsigned long long)(lds_f*)gE, L0x4 = 4u * (unsigned)L0;
    const int e_l = (lane & 15) >> 2, l_l = lane & 3;
    // tail masks of the last chunk (elements >= P read 0 from the bounded descriptor but must not enter the sums)
    bool tail_ok[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) tail_ok[cc] = 4 * (lane + 64 * (NCH - 1)) + cc < P;

    auto load_row = [&](int n, f4 (&v)[NCH]) {
        const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(mh + (long long)n * P, 0, P * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            v[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (lane + 64 * j) * 16, 0, 0));
    };
    // TABLE: lane k samples the grid points of its chunks (the same float32 operations, in the same order, as
    // get_heights_kernel), keeps them as the row and writes them to the measured-heights matrix
    auto sample_row = [&](int n, float bx, float by, float zq, float wq, f4 (&v)[NCH]) {
        const rsrc_t r_hs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(tp.hs), 0, tp.rows * tp.cols * 2, 0x00020000);
        const rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(mh + (long long)n * P, 0, P * 4, 0x00020000);
        const bool fast_div = tp.hscale == 0.05f || tp.hscale == 0.1f;      // constants covered by div_const_exact.hip
        const float inv = 1.0f / tp.hscale;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int i = 4 * (lane + 64 * j) + cc;
                const int ix = i / NY, iy = i - ix * NY;
                const float px = xs[ix & 63], py = ys[iy];
                const float t0 = -(zq * py) * 2.0f;
                const float t1 = (zq * px) * 2.0f;
                const float ax = (((px + wq * t0) + (-(zq * t1))) + bx) + tp.border;
                const float ay = (((py + wq * t1) + (zq * t0)) + by) + tp.border;
                float qx = div_const(ax, tp.hscale, inv), qy = div_const(ay, tp.hscale, inv);
                // outside the verified range of div_const (and for any other scale) take the IEEE quotient; below 1e-30 both
                // quotients truncate to 0
                if (__builtin_amdgcn_ballot_w64(!fast_div || fabsf(ax) > 1e30f || fabsf(ay) > 1e30f) != 0ull) {
                    qx = ax / tp.hscale;
                    qy = ay / tp.hscale;
                }
                long long cx = (long long)qx, cy = (long long)qy;        // .long(): truncation toward zero
                cx = cx < 0 ? 0 : (cx > tp.rows - 2 ? tp.rows - 2 : cx);
                cy = cy < 0 ? 0 : (cy > tp.cols - 2 ? tp.cols - 2 : cy);
                const int o = ((int)cx * tp.cols + (int)cy) * 2;
                const int16_t h1 = (int16_t)__builtin_amdgcn_raw_buffer_load_b16(r_hs, o, 0, 0);
                const int16_t h2 = (int16_t)__builtin_amdgcn_raw_buffer_load_b16(r_hs, o + tp.cols * 2, 0, 0);
                const int16_t h3 = (int16_t)__builtin_amdgcn_raw_buffer_load_b16(r_hs, o + 2, 0, 0);
                int16_t h = h1 < h2 ? h1 : h2;
                h = h < h3 ? h : h3;
                v[j][cc] = i < P ? (float)h * tp.vscale : 0.0f;          // past the row: what the bounded load returns
            }
            if (4 * (lane + 64 * j) + 3 < P) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v[j]), r_out, (lane + 64 * j) * 16, 0, 0);
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    if (4 * (lane + 64 * j) + cc < P)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v[j][cc]), r_out, (lane + 64 * j) * 16 + 4 * cc, 0, 0);
            }
        }
    };
    f4 cur[NCH];
    if (!TABLE) load_row(g0, cur);

    // per-(env, leg) inputs and outputs go through bounded buffer descriptors: lane offset (constant) + wave offset
    // (SGPR) -- no 64-bit per-lane addresses; lanes past N read 0 and never store
    auto whole = [&](const void* ptr, int bytes_per_env) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)((unsigned)N * (unsigned)bytes_per_env), 0x00020000);
    };
    const rsrc_t r_root = whole(root, 52), r_cmd = whole(cmd, 16), r_thigh = whole(thigh, 48), r_idx = whole(idx_out, 32),
                 r_obs = whole(obs_out, 32), r_world = whole(world_out, 48), r_pred = whole(pred_out, 48),
                 r_p2r = whole(p2r_out, 48);
    const int vo_root = e_l * 52, vo_cmd = e_l * 16, vo_leg = e_l * 48 + l_l * 12, vo_idx = e_l * 32 + l_l * 8,
              vo_obs = e_l * 32 + l_l * 4;
    auto ld1 = [](rsrc_t r, int vo, int so) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); };
    auto ld3 = [](rsrc_t r, int vo, int so) { return __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(r, vo, so, 0)); };
    auto ld4 = [](rsrc_t r, int vo, int so) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0)); };
    auto st3 = [](rsrc_t r, int vo, int so, float x, float y, float z) {
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u3, f3{x, y, z}), r, vo, so, 0);
    };

    // ================= per-(env, leg) algebra on lanes 0..15 (the other lanes mirror them)
    const bool lane_valid = lane < 16 && e_l < cnt;
    const f4 ra = ld4(r_root, vo_root, g0 * 52), rb = ld4(r_root, vo_root + 16, g0 * 52);
    const f2 rc = {ld1(r_root, vo_root + 32, g0 * 52), ld1(r_root, vo_root + 36, g0 * 52)};
    const float bx_l = ra.x, by_l = ra.y, bz_l = ra.z;
    const float qx = ra.w, qy = rb.x, qz = rb.y, qw = rb.z;
    const V3 vw{rb.w, rc.x, rc.y};
    const f3 cm = ld3(r_cmd, vo_cmd, g0 * 16);
    const float c0 = cm.x, c1 = cm.y, c2 = cm.z;
    const f3 thv = ld3(r_thigh, vo_leg, g0 * 48);
    const float th0 = thv.x, th1 = thv.y, th2 = thv.z;
    // Raibert nominal foothold of this leg (legged_robot_dtc.py:100-120)
    const V3 vb = quat_rotate_inverse(qx, qy, qz, qw, vw);
    float sn, cs;
    sincos_cw(c2, sn, cs);
    const float symx = gp.t_half * vb.x + gp.k_fb * (vb.x - c0);
    const float symy = gp.t_half * vb.y + gp.k_fb * (vb.y - c1);
    const float symz = gp.t_half * vb.z + gp.k_fb * (vb.z - 0.0f);
    const float hx0 = th0 - bx_l, hy0 = th1 - by_l, hz0 = th2 - bz_l;
    const float rx = cs * hx0 + (-sn) * hy0;
    const float ry = sn * hx0 + cs * hy0;
    const float predx_l = (bx_l + rx) + symx;
    const float predy_l = (by_l + ry) + symy;
    const float predz_l = (bz_l + hz0) + symz;
    // yaw-only attitude (math.py:8-12)
    float nq = sqrtf(qz * qz + qw * qw);
    nq = fmaxf(nq, 1e-9f);
    const float zq_l = qz / nq, wq_l = qw / nq;
    // 8 x 8 candidate patch: nominal foothold in the yaw frame of the grid, R(-yaw) * (pred - base)
    int sx_l, sy_l;
    {
        const float prx = predx_l - bx_l, pry = predy_l - by_l;
        const float cyaw = wq_l * wq_l - zq_l * zq_l, syaw = 2.0f * wq_l * zq_l;
        const float lx = cyaw * prx + syaw * pry, ly = -syaw * prx + cyaw * pry;
        const float u = (lx - gp.x0) * gp.inv_dx, v = (ly - gp.y0) * gp.inv_dy;
        // all in-radius cells have |iu - u| < rad: they fit in 8 consecutive cells iff 2*rad < 7 (patch_ok, host)
        const bool fits = gp.patch_ok && (fabsf(u) < 1e6f) && (fabsf(v) < 1e6f);
        sx_l = fits ? (int)floorf(u - gp.rad_x) + 1 : -(1 << 20);
        sy_l = fits ? (int)floorf(v - gp.rad_y) + 1 : -(1 << 20);
    }
    if (lane_valid) {
        st3(r_pred, vo_leg, g0 * 48, predx_l, predy_l, predz_l);
        const V3 pr = quat_rotate_inverse(qx, qy, qz, qw, V3{predx_l - bx_l, predy_l - by_l, predz_l - bz_l});
        st3(r_p2r, vo_leg, g0 * 48, pr.x, pr.y, pr.z);
    }

    int my_bi = 0;
    float my_z = 0.0f;
    for (int e = 0; e < cnt; ++e) {
        const int src = e * 4;
        const float bx = rlf(bx_l, src), by = rlf(by_l, src), bz = rlf(bz_l, src);
        const float zq = rlf(zq_l, src), wq = rlf(wq_l, src);
        if (TABLE) sample_row(g0 + e, bx, by, zq, wq, cur);

        // ================= clamp + mean / unbiased variance from registers (legged_robot_dtc.py:127-141)
        f4 g[NCH];
        f2 acc = {0.0f, 0.0f};
        const f2 bz2 = {bz, bz};
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const f2 lo = cur[j].xy - bz2, hi = cur[j].zw - bz2;
            f4 v = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                v[cc] = __builtin_amdgcn_fmed3f(v[cc], -0.5f, 0.5f);
                if (j == NCH - 1) v[cc] = tail_ok[cc] ? v[cc] : 0.0f;
            }
            g[j] = v;
            acc = acc + v.xy;
            acc = acc + v.zw;
        }
        const float mean = div_count<P>(wave_sum_all(acc.x + acc.y));
        acc = f2{0.0f, 0.0f};
        const f2 mean2 = {mean, mean};
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f2 lo = g[j].xy - mean2, hi = g[j].zw - mean2;
            if (j == NCH - 1) {
                lo.x = tail_ok[0] ? lo.x : 0.0f;
                lo.y = tail_ok[1] ? lo.y : 0.0f;
                hi.x = tail_ok[2] ? hi.x : 0.0f;
                hi.y = tail_ok[3] ? hi.y : 0.0f;
            }
            acc = acc + lo * lo;
            acc = acc + hi * hi;
        }
        const float var = div_count<P - 1>(wave_sum_all(acc.x + acc.y));
        const float edge = fminf(fmaxf(sqrt_rn(var), 0.0f), 0.3f);
        const float edge02 = 0.2f * edge;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int q4 = 4 * (lane + 64 * j);
            if (q4 < RS) {
                *reinterpret_cast<f4*>(gE + q4) = g[j];
                *reinterpret_cast<f4*>(rawE + q4) = cur[j];
            }
        }
        if (!TABLE && e + 1 < cnt) load_row(g0 + e + 1, cur);          // next env's row travels while this one is scored

        if constexpr (STEP) {
            const int n = g0 + e;
            const DtcObsCfg& c = sa.cfg;
            // ---- check_termination: lanes stride over the termination bodies / the height slice exactly as check_termination_kernel
            bool hit = false;
            for (int j = lane; j < sa.n_term; j += 64) {
                const float* f = sa.contact_forces + ((long long)n * sa.num_bodies + sa.term_idx[j]) * 3;
                hit |= sqrtf((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]) > 100.0f;
            }
            const bool contact = __ballot(hit) != 0ull;
            float part = 0.f;
            bool first = true;
            for (int p = c.term_row0 + lane; p < c.term_row1; p += 64) {
                const float d = bz - fmaxf(rawE[p], -0.0f);
                part = first ? d : part + d;
                first = false;
            }
            const float tmean = wave_sum(part) / (float)(c.term_row1 - c.term_row0);
            if (lane == 0) {
                const bool to = sa.episode_length[n] > sa.max_episode_length;
                const bool r = contact || to || sa.gravity[n * 3 + 2] > 0.2f || tmean < c.term_height;
                sa.reset_buf[n] = r ? 1 : 0;
                if (sa.time_out_buf) sa.time_out_buf[n] = to ? 1 : 0;
                if (sa.height_mean) sa.height_mean[n] = tmean;
            }
            // ---- height part of compute_observations: privileged = [noisy heights | force | clean heights]
            const float zt = bz - c.base_height_target;
            float* pv = sa.priv + (long long)n * (2 * P + 3);
            for (int p = lane; p < P; p += 64) {
                const float h = fminf(fmaxf(zt - rawE[p], -1.0f), 1.0f) * c.height_measurements;
                float nz = h;
                if (sa.u_heights) nz = nz + (2.0f * sa.u_heights[(long long)n * P + p] - 1.0f) * c.height_noise;
                if (sa.noise_offset) nz = nz + sa.noise_offset[(long long)n * P + p];
                pv[p] = nz;
                pv[P + 3 + p] = h;
                if (sa.heights) sa.heights[(long long)n * P + p] = h;
            }
            if (lane < 3) pv[P + lane] = sa.forces[(long long)n * sa.ld_forces + lane] * c.force;
        }

        // ================= scoring of one 8 x 8 patch whose candidates and their four neighbours are all grid points
        // (1 <= sx, sx + 8 <= NX - 1, same in y): fixed LDS offsets, the constant 0.1 divisor, both axes packed
        auto score_interior = [&](int sx, int sy, float pxl, float pyl, float& tot, int& ii) {
            const int sbase = __builtin_amdgcn_readfirstlane(sx * NY + sy);
            ii = sbase + L0;
            // LDS byte address of g[ii - NY]: uniform part + lane part, so that the neighbours, the centre and the
            // measured height are immediate offsets of ONE address register
            const unsigned pa = (unsigned)__builtin_amdgcn_readfirstlane((int)(g_lds + 4u * (unsigned)(sbase - NY))) + L0x4;
            const lds_f* p = (const lds_f*)pa;
            const f2 num = f2{p[2 * NY], p[NY + 1]} - f2{p[0], p[NY - 1]};
            const f2 q0 = num * 10.0f;
            const f2 r = fma2(-q0, f2{0.1f, 0.1f}, num);
            const f2 dv = fma2(r, f2{10.0f, 10.0f}, q0);          // == num / 0.1f (div_const)
            const f2 sq = dv * dv;
            const float slope = sqrt_rn(sq.x + sq.y);
            const float rough = fabsf(p[NY] - mean);
            const float s_raw = (edge02 + slope) + 0.3f * rough;
            const float s = s_raw < 0.1f ? s_raw : 10.0f;
            const float rel = p[RS + NY] - bz;
            const bool exc = (rel > 1.0f) | (rel < -1.0f);
            const f2 pp = {xs[sx + wx], ys[sy + wy]};
            const f2 t = (zq * pp.yx) * f2{-2.0f, 2.0f};          // t0 = -(zq*py)*2, t1 = (zq*px)*2
            const f2 u = pp + wq * t;
            const f2 z2 = zq * t.yx;                              // (zq*t1, zq*t0)
            const f2 h = (u + f2{-z2.x, z2.y}) + f2{bx, by};
            const f2 dd = f2{pxl, pyl} - h;
            const f2 d2 = dd * dd;
            float d = sqrt_rn(d2.x + d2.y);
            d = d < 0.16f ? d : 10.0f;
            tot = s * 0.2f + d * 0.8f;
            tot = exc ? 10.0f : tot;
        };
        auto interior = [&](int sx, int sy) { return sx >= 1 && sx + 8 <= NX - 1 && sy >= 1 && sy + 8 <= NY - 1; };

        // ================= usual case: the four candidate patches are interior.  Score them, then ONE butterfly for the four
        // minima: totals are sums of non-negative terms, so their bit patterns order like unsigned integers; after the
        // xor-1 step the legs are interleaved by lane parity, after the xor-2 step by lane & 3, and the remaining steps
        // (rotations by 4 and 8 inside a row, row / half swaps) keep lane & 3: lane k < 4 ends with the minimum of leg k
        bool fast4 = true;
#pragma unroll
        for (int l = 0; l < 4; ++l)
            fast4 &= interior(__builtin_amdgcn_readlane(sx_l, src + l), __builtin_amdgcn_readlane(sy_l, src + l));
        unsigned need = 0xFu;                  // legs that still have to take the general path below
        if (fast4) {
            float tot[4];
            int ii[4];
            unsigned key[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int sl = src + l;
#if DTC_FH_ABL
                // timing ablation (WRONG results): legs >= DTC_FH_ABL are scored by the distance term alone -- the most ANY compaction of the in-radius
                // candidates could save if it packed four legs' slope / roughness evaluations into (DTC_FH_ABL) passes at no cost
                if (l >= DTC_FH_ABL) {
                    const int sx = __builtin_amdgcn_readlane(sx_l, sl), sy = __builtin_amdgcn_readlane(sy_l, sl);
                    ii[l] = __builtin_amdgcn_readfirstlane(sx * NY + sy) + L0;
                    const f2 pp = {xs[sx + wx], ys[sy + wy]};
                    const f2 t = (zq * pp.yx) * f2{-2.0f, 2.0f};
                    const f2 u = pp + wq * t;
                    const f2 z2 = zq * t.yx;
                    const f2 h = (u + f2{-z2.x, z2.y}) + f2{bx, by};
                    const f2 dd = f2{rlf(predx_l, sl), rlf(predy_l, sl)} - h;
                    const f2 d2 = dd * dd;
                    float d = sqrt_rn(d2.x + d2.y);
                    d = d < 0.16f ? d : 10.0f;
                    tot[l] = d * 0.8f;
                } else
#endif
                score_interior(__builtin_amdgcn_readlane(sx_l, sl), __builtin_amdgcn_readlane(sy_l, sl), rlf(predx_l, sl),
                               rlf(predy_l, sl), tot[l], ii[l]);
                key[l] = (unsigned)__float_as_int(tot[l]);
                key[l] = min(key[l], dpp_u<0xB1, 0xF>(key[l]));
            }
            unsigned kab = (lane & 1) ? key[1] : key[0], kcd = (lane & 1) ? key[3] : key[2];
            kab = min(kab, dpp_u<0x4E, 0xF>(kab));
            kcd = min(kcd, dpp_u<0x4E, 0xF>(kcd));
            unsigned kk = (lane & 2) ? kcd : kab;
            kk = min(kk, dpp_u<0x124, 0xF>(kk));                 // row_ror:4
            kk = min(kk, dpp_u<0x128, 0xF>(kk));                 // row_ror:8
            u2 sw = __builtin_amdgcn_permlane16_swap(kk, kk, false, false);
            kk = min(sw.x, sw.y);
            sw = __builtin_amdgcn_permlane32_swap(kk, kk, false, false);
            kk = min(sw.x, sw.y);
            need = 0u;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const float best = __int_as_float(__builtin_amdgcn_readlane((int)kk, l));
                // the LOWEST lane holding the minimum (the flat index grows with the lane id inside a patch: lowest lane
                // == lowest index == torch's tie rule)
                const unsigned long long hit = __ballot(tot[l] == best);
                const int srcl = hit ? __ffsll((long long)hit) - 1 : 0;
                const int bwin = __builtin_amdgcn_readlane(ii[l], __builtin_amdgcn_readfirstlane(srcl));
                // a "valid" winner of the candidate patch is the global argmin (see the generic kernel)
                if (best < 1.0f) {
                    if (lane == src + l) {
                        my_bi = bwin;
                        my_z = rawE[bwin];
                    }
                } else {
                    need |= 1u << l;
                }
            }
        }
        // ================= general path (rare), one leg at a time: the candidate patch when it touches the border or lies
        // outside the grid, and -- if it holds no valid candidate -- the PX x PY patches that tile the whole grid, i.e. the
        // reference's full argmin (ties: lowest flat index)
        for (int l = 0; need != 0u && l < 4; ++l) {
            if (!((need >> l) & 1u)) continue;
            const int sl = src + l;
            const float pxl = rlf(predx_l, sl), pyl = rlf(predy_l, sl);
            int sx = __builtin_amdgcn_readlane(sx_l, sl), sy = __builtin_amdgcn_readlane(sy_l, sl);
            float best = __builtin_inff();
            int bwin = 0x7fffffff;
            for (int k = fast4 ? 0 : -1; k < PX * PY; ++k) {
                if (k >= 0) {
                    sx = (k / PY) * 8;
                    sy = (k % PY) * 8;
                }
                float tot = __builtin_inff();
                int ii = 0x7fffffff;
                if (interior(sx, sy)) {
                    score_interior(sx, sy, pxl, pyl, tot, ii);
                } else {
                    const int ix = sx + wx, iy = sy + wy;
                    if (ix >= 0 && ix < NX && iy >= 0 && iy < NY) {
                        ii = ix * NY + iy;
                        const EnvCtx c{bx, by, bz, zq, wq, mean, edge, NX, NY};
                        const PointEval pe = eval_point(c, rawE, gE, xs, ys, ii, ix, iy);
                        float d;
                        tot = total_score(pe, pxl, pyl, d);
                    }
                }
                const float mn = wave_min_nonneg(tot);
                const unsigned long long hit = __ballot(tot == mn);
                const int srcl = hit ? __ffsll((long long)hit) - 1 : 0;
                const int win = __builtin_amdgcn_readlane(ii, __builtin_amdgcn_readfirstlane(srcl));
                if (mn < best || (mn == best && win < bwin)) {
                    best = mn;
                    bwin = win;
                }
                if (k < 0 && mn < 1.0f) break;          // valid winner of the candidate patch
            }
            if (bwin == 0x7fffffff) bwin = 0;
            if (lane == sl) {
                my_bi = bwin;
                my_z = rawE[bwin];
            }
        }
    }

    // ================= decode (legged_robot_dtc.py:184-201) on lanes (env of the wave, leg)
    float fo_x = 0.0f, fo_y = 0.0f, ow_x = 0.0f, ow_y = 0.0f;
    if (lane_valid) {
        const int bi = my_bi, yi = bi / NY, xi = bi - yi * NY;       // xi = idx % ny, yi = idx / ny
        __builtin_amdgcn_raw_buffer_store_b64(u2{(unsigned)bi, 0u}, r_idx, vo_idx, g0 * 32, 0);
        // the reference gathers the x table with the y-index and vice versa (sic); xi < ny, yi < nx
        fo_x = xs[xi < NX ? xi : xi % NX];
        fo_y = ys[yi < NY ? yi : yi % NY];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(fo_x), r_obs, vo_obs, g0 * 32, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(fo_y), r_obs, vo_obs + 16, g0 * 32, 0);
        const float px = xs[yi], py = ys[xi];
        const float t0 = -(zq_l * py) * 2.0f;
        const float t1 = (zq_l * px) * 2.0f;
        ow_x = ((px + wq_l * t0) + (-(zq_l * t1))) + bx_l;
        ow_y = ((py + wq_l * t1) + (zq_l * t0)) + by_l;
        st3(r_world, vo_leg, g0 * 48, ow_x, ow_y, my_z);
    }

    if constexpr (STEP) {
        const DtcObsCfg& c = sa.cfg;
        // ---- foothold rewards: lane (env, leg) evaluates its leg, the env's first lane adds the four terms in leg order
        // (foothold_rewards_kernel: sum = ((0 + t0) + t1) + t2) + t3)
        float term = 0.0f, fz = __builtin_inff();
        if (lane_valid) {
            const float* f = sa.foot_positions + (long long)(g0 + e_l) * 12 + l_l * 3;
            const float dx = f[0] - ow_x, dy = f[1] - ow_y;
            const float dis = sqrtf(dx * dx + dy * dy);
            const float r = -logf(0.8f + dis);
            term = sa.contact_filt[(g0 + e_l) * 4 + l_l] ? r : 0.0f;
            fz = f[2];
        }
        float sum = 0.0f, minz = __builtin_inff();
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            sum = sum + __shfl(term, (lane & ~3) + l, 64);
            minz = fminf(minz, __shfl(fz, (lane & ~3) + l, 64));
        }
        if (lane_valid && l_l == 0) {
            if (sa.rew_tracking) sa.rew_tracking[g0 + e_l] = sum;
            if (sa.rew_miss) sa.rew_miss[g0 + e_l] = minz < 0.0f ? 1.0f : 0.0f;
        }
        // ---- proprioceptive observation rows (one lane per element, as env_observations_kernel); the eight foothold
        // observations come from the lanes that decoded them
        const int D = c.num_dof, F = c.num_foothold_obs, n_obs = 9 + 3 * D + F;
        for (int e = 0; e < cnt; ++e) {
            const int n = g0 + e;
            for (int base = 0; base < n_obs; base += 64) {
                const int el = base + lane;
                const int k = el - (9 + 3 * D);
                const int srcl = e * 4 + (k >= 0 && k < 8 ? (k & 3) : 0);
                const float fx = __shfl(fo_x, srcl, 64), fy = __shfl(fo_y, srcl, 64);
                if (el < n_obs) {
                    float v;
                    if (el < 3) v = sa.ang_vel[n * 3 + el] * c.ang_vel;
                    else if (el < 6) v = sa.gravity[n * 3 + (el - 3)];
                    else if (el < 9) v = cmd[n * 4 + (el - 6)] * c.commands_scale[el - 6];
                    else if (el < 9 + D) v = (sa.dof_pos[n * D + (el - 9)] - sa.default_dof_pos[el - 9]) * c.dof_pos;
                    else if (el < 9 + 2 * D) v = sa.dof_vel[n * D + (el - 9 - D)] * c.dof_vel;
                    else if (el < 9 + 3 * D) v = sa.actions[n * D + (el - 9 - 2 * D)];
                    else v = k < 4 ? fx : fy;
                    if (sa.u_obs) v = v + (2.0f * sa.u_obs[(long long)n * n_obs + el] - 1.0f) * sa.noise_scale[el];
                    sa.obs[(long long)n * n_obs + el] = v;
                }
            }
        }
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
    gp.x0 = gp.x[0];
    gp.y0 = gp.y[0];
    gp.inv_dx = (float)(gp.nx - 1) / (gp.x[gp.nx - 1] - gp.x0);
    gp.inv_dy = (float)(gp.ny - 1) / (gp.y[gp.ny - 1] - gp.y0);
    gp.rad_x = 0.16f * gp.inv_dx + 0.05f;
    gp.rad_y = 0.16f * gp.inv_dy + 0.05f;
    gp.patch_ok = (2.0f * gp.rad_x < 7.0f) && (2.0f * gp.rad_y < 7.0f) ? 1 : 0;
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
    if (!debug && gp.nx == 33 && gp.ny == 21 && !getenv("DTC_PLANNER_GENERIC")) {
        // four envs per wave, 16 per workgroup (a persistent split measured slower: without new workgroups to backfill, the
        // chip drains unevenly); the descriptors of the per-env arrays need N * 52 bytes < 4 GiB
        DTC_REQUIRE(N <= 40000000, "N too large for the 32-bit buffer offsets of the planner");
        const int fgrid = (int)dtc::ceil_div((int64_t)N, FAST_ENVS_PER_BLOCK);
        dtc::ProfScope prof("foothold_plan", bytes, s);
        hipLaunchKernelGGL((foothold_plan_fast_kernel<33, 21, false>), dim3(fgrid), dim3(256), 0, s,
                           const_cast<float*>(measured_heights), root_states, thigh_pos, commands, gp, TableParams{}, idx,
                           foothold_obs, opt_world, pred, pred_to_robot, N, StepArgs{});
        return dtc::check_launch("foothold_plan");
    }
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

extern "C" int dtc_foothold_plan_from_table(const int16_t* height_samples, int rows, int cols, float border_size,
                                            float horizontal_scale, float vertical_scale, const float* root_states,
                                            const float* thigh_pos, const float* commands, const DtcGridCfg* cfg,
                                            float* measured_heights, int64_t* idx, float* foothold_obs, float* opt_world,
                                            float* pred, float* pred_to_robot, int N, void* stream) {
    GridParams gp;
    int rc = make_params(cfg, gp);
    if (rc != DTC_OK) return rc;
    DTC_REQUIRE(N >= 0 && rows >= 2 && cols >= 2, "bad shape");
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(height_samples && root_states && thigh_pos && commands && measured_heights, "null pointer");
    DTC_REQUIRE(idx && foothold_obs && opt_world && pred && pred_to_robot, "null output");
    if (gp.nx != 33 || gp.ny != 21 || (int64_t)rows * cols >= (1ll << 30) || getenv("DTC_PLANNER_GENERIC")) {
        // any other grid: the two launches the fused kernel replaces
        rc = dtc_get_heights(height_samples, rows, cols, root_states, cfg, border_size, horizontal_scale, vertical_scale,
                             measured_heights, N, stream);
        if (rc != DTC_OK) return rc;
        return dtc_foothold_plan(measured_heights, root_states, thigh_pos, commands, cfg, idx, foothold_obs, opt_world, pred,
                                 pred_to_robot, nullptr, nullptr, nullptr, nullptr, N, stream);
    }
    DTC_REQUIRE(N <= 40000000, "N too large for the 32-bit buffer offsets of the planner");
    hipStream_t s = (hipStream_t)stream;
    const double bytes = (double)N * (gp.P * 4.0 + 13 * 4 + 4 * 4 + 12 * 4 + 32 + 32 + 48 + 96);
    dtc::ProfScope prof("foothold_plan_from_table", bytes, s);
    const TableParams tp{height_samples, rows, cols, border_size, horizontal_scale, vertical_scale};
    hipLaunchKernelGGL((foothold_plan_fast_kernel<33, 21, true>), dim3((unsigned)dtc::ceil_div((int64_t)N, FAST_ENVS_PER_BLOCK)),
                       dim3(256), 0, s, measured_heights, root_states, thigh_pos, commands, gp, tp, idx, foothold_obs,
                       opt_world, pred, pred_to_robot, N, StepArgs{});
    return dtc::check_launch("foothold_plan_from_table");
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

// One env step's post-physics block as ONE launch (33 x 21 grid): [heights from the terrain table +] foothold plan + check_termination
// + foothold rewards + compute_observations.  Any other grid: the separate launches, in the reference's order.
extern "C" int dtc_env_post_physics(const DtcEnvStep* st, const DtcGridCfg* grid, const DtcObsCfg* oc, int N, void* stream) {
    DTC_REQUIRE(st && grid && oc, "null descriptor");
    GridParams gp;
    int rc = make_params(grid, gp);
    if (rc != DTC_OK) return rc;
    DTC_REQUIRE(N >= 0, "N < 0");
    if (N == 0) return DTC_OK;
    const bool table = st->height_samples != nullptr;
    DTC_REQUIRE(!table || (st->rows >= 2 && st->cols >= 2), "bad terrain table shape");
    DTC_REQUIRE(st->root_states && st->thigh_pos && st->commands && st->measured_heights, "null planner input");
    DTC_REQUIRE(st->idx && st->foothold_obs && st->opt_world && st->pred && st->pred_to_robot, "null planner output");
    DTC_REQUIRE(st->contact_forces && st->episode_length_buf && st->projected_gravity && st->reset_buf, "null termination argument");
    DTC_REQUIRE(st->n_term >= 0 && (st->n_term == 0 || st->termination_contact_indices), "bad termination index list");
    DTC_REQUIRE(st->foot_positions && st->contact_filt && (st->rew_tracking || st->rew_miss), "null reward argument");
    DTC_REQUIRE(st->base_ang_vel && st->dof_pos && st->default_dof_pos && st->dof_vel && st->actions && st->forces && st->obs_buf &&
                st->privileged_obs_buf, "null observation argument");
    DTC_REQUIRE(!st->u_obs || st->noise_scale_vec, "observation noise needs noise_scale_vec");
    DTC_REQUIRE(oc->num_dof > 0 && oc->num_points == gp.P && oc->num_foothold_obs == 8 && st->ld_forces >= 3,
                "observation config does not match the grid (num_points %d vs %d, 8 foothold observations)", oc->num_points, gp.P);
    DTC_REQUIRE(oc->term_row0 >= 0 && oc->term_row1 > oc->term_row0 && oc->term_row1 <= oc->num_points, "bad height slice");
    if (gp.nx != 33 || gp.ny != 21 || (table && (int64_t)st->rows * st->cols >= (1ll << 30)) || getenv("DTC_PLANNER_GENERIC")) {
        rc = table ? dtc_foothold_plan_from_table(st->height_samples, st->rows, st->cols, st->border_size, st->horizontal_scale,
                                                  st->vertical_scale, st->root_states, st->thigh_pos, st->commands, grid,
                                                  st->measured_heights, st->idx, st->foothold_obs, st->opt_world, st->pred,
                                                  st->pred_to_robot, N, stream)
                   : dtc_foothold_plan(st->measured_heights, st->root_states, st->thigh_pos, st->commands, grid, st->idx, st->foothold_obs,
                                       st->opt_world, st->pred, st->pred_to_robot, nullptr, nullptr, nullptr, nullptr, N, stream);
        if (rc != DTC_OK) return rc;
        rc = dtc_check_termination(st->contact_forces, st->num_bodies, st->termination_contact_indices, st->n_term, st->episode_length_buf,
                                   st->max_episode_length, st->projected_gravity, st->root_states, st->measured_heights, oc, st->reset_buf,
                                   st->time_out_buf, st->height_mean, N, stream);
        if (rc != DTC_OK) return rc;
        rc = dtc_foothold_rewards(st->foot_positions, st->opt_world, st->contact_filt, st->rew_tracking, st->rew_miss, N, stream);
        if (rc != DTC_OK) return rc;
        return dtc_compute_observations(st->base_ang_vel, st->projected_gravity, st->commands, st->dof_pos, st->default_dof_pos, st->dof_vel,
                                        st->actions, st->foothold_obs, st->root_states, st->measured_heights, st->forces, st->ld_forces,
                                        st->height_noise_offset, st->u_obs, st->noise_scale_vec, st->u_heights, oc, st->obs_buf,
                                        st->privileged_obs_buf, st->heights, N, stream);
    }
    DTC_REQUIRE(N <= 40000000, "N too large for the 32-bit buffer offsets of the planner");
    hipStream_t s = (hipStream_t)stream;
    StepArgs sa{};
    sa.contact_forces = st->contact_forces;
    sa.term_idx = st->termination_contact_indices;
    sa.episode_length = (const long long*)st->episode_length_buf;
    sa.gravity = st->projected_gravity;
    sa.reset_buf = st->reset_buf;
    sa.time_out_buf = st->time_out_buf;
    sa.height_mean = st->height_mean;
    sa.max_episode_length = (long long)st->max_episode_length;
    sa.num_bodies = st->num_bodies;
    sa.n_term = st->n_term;
    sa.foot_positions = st->foot_positions;
    sa.contact_filt = st->contact_filt;
    sa.rew_tracking = st->rew_tracking;
    sa.rew_miss = st->rew_miss;
    sa.ang_vel = st->base_ang_vel;
    sa.dof_pos = st->dof_pos;
    sa.default_dof_pos = st->default_dof_pos;
    sa.dof_vel = st->dof_vel;
    sa.actions = st->actions;
    sa.forces = st->forces;
    sa.noise_offset = st->height_noise_offset;
    sa.u_obs = st->u_obs;
    sa.noise_scale = st->noise_scale_vec;
    sa.u_heights = st->u_heights;
    sa.ld_forces = (long long)st->ld_forces;
    sa.obs = st->obs_buf;
    sa.priv = st->privileged_obs_buf;
    sa.heights = st->heights;
    sa.cfg = *oc;
    // bytes: the planner's 3096 B/env + what the three other kernels move that the row in LDS does not already cover
    const double bytes = (double)N * (gp.P * 4.0 * (1 + 3 + (st->u_heights ? 1 : 0) + (st->height_noise_offset ? 1 : 0)) + 13 * 4 + 16 + 48 +
                                      208 + 53 * 4.0 * 3 + st->n_term * 12.0 + 64);
    dtc::ProfScope prof("env_post_physics", bytes, s);
    const TableParams tp{st->height_samples, st->rows, st->cols, st->border_size, st->horizontal_scale, st->vertical_scale};
    const dim3 g((unsigned)dtc::ceil_div((int64_t)N, FAST_ENVS_PER_BLOCK));
    if (table)
        hipLaunchKernelGGL((foothold_plan_fast_kernel<33, 21, true, true>), g, dim3(256), 0, s, st->measured_heights, st->root_states,
                           st->thigh_pos, st->commands, gp, tp, st->idx, st->foothold_obs, st->opt_world, st->pred, st->pred_to_robot, N, sa);
    else
        hipLaunchKernelGGL((foothold_plan_fast_kernel<33, 21, false, true>), g, dim3(256), 0, s, st->measured_heights, st->root_states,
                           st->thigh_pos, st->commands, gp, TableParams{}, st->idx, st->foothold_obs, st->opt_world, st->pred,
                           st->pred_to_robot, N, sa);
    return dtc::check_launch("env_post_physics");
}
