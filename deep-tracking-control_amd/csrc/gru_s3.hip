// GRU recurrence on the split-precision path (gfx950): the two per-time-step products of torch.nn.GRU's BPTT
// (rsl_rl/rsl_rl/modules/actor_critic_recurrent.py:92-116 under ppo.py:265-335) with every fp32 operand as three bf16 terms and six
// v_mfma_f32_32x32x16_bf16 passes per product (csrc/gemm_s3.hip explains the arithmetic and its accuracy):
//   forward  : gh = h_{t-1} W_hh^T for the three gates of 32 hidden units + the gate math in the epilogue (the split twin of
//              gru_step_fwd_kernel in gemm.hip) -- block tile 128 rows x (3 gates x 32 units), wave = 32 rows x 96 columns;
//   backward : the chunks of dh_{t-1} += dgh_t W_hh (the split twin of dtc_linear_dgrad_split) -- block tile 128 x 128, 2 x 2 waves.
// One time step has R ~ 1500 rows: 12 row tiles, 150-300 workgroups, ONE workgroup per CU and one wave per SIMD -- nothing hides a
// load behind another wave, and the general kernels of gemm_s3.hip (loads one stage ahead, built for three workgroups per CU) spend
// most of such a launch waiting (29 us for the backward chunks).  These kernels are built for that regime instead:
//   * W_hh comes as an LDS image (gemm_s3.hip: wimage) built ONCE per forward / backward pass and copied by LDS-DMA one stage ahead:
//     it serves all T time steps, and it stays in L2;
//   * the row operand (h_{t-1} resp. dgh_t, L2-resident: the previous kernel wrote it) is loaded TWO stages ahead into two register
//     sets, converted one stage ahead (between the MFMAs of the stage in flight) -- registers are free at one wave per SIMD;
//   * the K loop is unrolled by two, so stage buffers and register sets have constant indices.
#include <type_traits>

#include "s3_core.hpp"

namespace {

constexpr int MODE_FWD = 0, MODE_BWD = 1;
template <int MODE>
struct Geo {
    static constexpr int WN = MODE == MODE_FWD ? 1 : 2;          // waves along the columns
    static constexpr int WM = 4 / WN;
    static constexpr int TM = BM / (32 * WM);                    // 32 x 32 tiles per wave
    static constexpr int TN = MODE == MODE_FWD ? 3 : 2;
    static constexpr int BN = 32 * TN * WN;                      // 96 / 128 columns per tile
    static constexpr int PLANE = 128 * 32;                       // bytes of one plane of one stage: 128 rows in both images (the forward
                                                                 // tile's last 32 are zero padding: every wave then issues three LDS-DMA
                                                                 // pieces per stage, no per-wave branches in the K loop)
    static constexpr int CHUNK = 3 * PLANE;                      // 12 KiB
};

struct GruS3Args {
    const float* A;             // [R, lda]: h_{t-1} (forward) / dgh_t (backward)
    long long lda;
    const u32x4* img;           // image of W_hh (forward: gate-interleaved unit tiles) / W_hh^T (backward)
    long long img_bytes;
    int R, H;
    int stages;                 // stages of one block's reduction (H / 16 forward; chunk / 16 backward)
    int stages_tile;            // stages of a whole column tile in the image (backward: 3 H / 16)
    // forward epilogue
    const float* bhh;
    const float* gi;
    float* hout;
    float* gates;
    float* hn;
    // backward epilogue: chunk c -> part + c * part_stride, [R, H]
    float* part;
    long long part_stride;
    int colmap;                 // workgroup -> tile map by COLUMN tile (and chunk) per XCD: see gru_map
    int nparts;
};

// forward image: tile tc = units [32 tc, 32 tc + 32), row n of the tile = gate n / 32 of unit 32 tc + n % 32, reduction = hidden
// index; backward image: tile tc = output columns [128 tc, +128) of dh, row = column, reduction = the 3H gate index (W_hh^T)
template <int MODE>
__global__ __launch_bounds__(256) void gru_wimage_kernel(const float* __restrict__ Whh, u32x4* __restrict__ img, int H, int stages_tile) {
    using G = Geo<MODE>;
    const int tc = blockIdx.x / stages_tile, st = blockIdx.x - tc * stages_tile;
    const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
    f32x4 v[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = st * BK + 8 * h + e;
        float x;
        if (MODE == MODE_FWD) x = r < G::BN ? Whh[((long long)(r >> 5) * H + tc * 32 + (r & 31)) * H + k] : 0.f;
        else x = Whh[(long long)k * H + tc * 128 + r];
        v[e >> 2][e & 3] = x;
    }
    const Split3 s0 = split3(v[0]), s1 = split3(v[1]);
    u32x4* dst = img + (long long)blockIdx.x * (G::CHUNK / 16);
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p * (G::PLANE / 16) + rslot(r, h)] = u32x4{s0.p[p].x, s0.p[p].y, s1.p[p].x, s1.p[p].y};
}

__device__ __forceinline__ float sigmoid_s3(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef __attribute__((address_space(3))) void lds_void_t;

// blockIdx.z: which recurrence (dtc_gru_fwd_multi / dtc_gru_bwd_multi run the time step of up to two recurrences of one shape -- the
// actor's and the critic's -- as ONE launch: two such launches on two streams overlap by only ~20 %, tools/gru_pair_probe.py)
template <int MODE>
__global__ __launch_bounds__(256, 2) void gru_s3_kernel(const GruS3Args a0, const GruS3Args a1) {
    using G = Geo<MODE>;
    const GruS3Args a = blockIdx.z == 0 ? a0 : a1;
    constexpr int TM = G::TM, TN = G::TN, NA = 2, NM = 6 * TM * TN;
    constexpr int GAP0 = 4;                                   // the conversion levels follow MFMAs GAP0 .. GAP0 + 7 of a stage
    constexpr int PLANE = G::PLANE, CHUNK = G::CHUNK;        // local copies: the generic lambdas below must not odr-use the members
    __shared__ __attribute__((aligned(16))) u32x2 As[2][3][BM * 4];
    // image ring: four stage buffers as SEPARATE objects (the compiler then knows that the LDS-DMA into one cannot alias the fragment
    // reads of another, see linear_s3_kernel)
    __shared__ __attribute__((aligned(16))) u32x2 Bs0[3][128 * 4];
    __shared__ __attribute__((aligned(16))) u32x2 Bs1[3][128 * 4];
    __shared__ __attribute__((aligned(16))) u32x2 Bs2[3][128 * 4];
    __shared__ __attribute__((aligned(16))) u32x2 Bs3[3][128 * 4];
#define GBS(b) ((b) == 0 ? Bs0 : (b) == 1 ? Bs1 : (b) == 2 ? Bs2 : Bs3)
    int tr, tc, chunk = MODE == MODE_BWD ? blockIdx.y : 0;
    const int col_tiles = MODE == MODE_FWD ? a.H / 32 : a.H / 128;
    if (a.colmap) {
        // XCD x (= blockIdx.x & 7: workgroups go round-robin over the XCDs) owns a fixed set of (column tile, chunk) pairs and runs them
        // for ALL row tiles: its slice of the W_hh image (1/8 of 4.7 MB) stays in its 4 MiB L2 over the 24 time steps, and what it
        // fetches from the Infinity Cache per step is the row operand.  With the row-tile map every XCD walks the WHOLE image once per
        // step -- more than its L2 holds -- and a time step's time follows the bytes that miss: two recurrences in one launch took 1.6 x
        // the time of one (tools/gru_pair_probe.py).
        const int row_tiles = (a.R + BM - 1) / BM, combos = col_tiles * a.nparts, per_xcd = (combos + 7) >> 3;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, cl = j / row_tiles;
        tr = j - cl * row_tiles;
        const int combo = xcd * per_xcd + cl;
        if (combo >= combos) return;
        chunk = combo / col_tiles;
        tc = combo - chunk * col_tiles;
    } else if (!map_tile(blockIdx.x, (a.R + BM - 1) / BM, col_tiles, tr, tc)) return;
    const int m0 = tr * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm_off = (wave / G::WN) * (32 * TM), wn_off = (wave % G::WN) * (32 * TN);
    const int half = lane >> 5, l31 = lane & 31;
    const int lrow = tid >> 2, lch = tid & 3;
    const int aslot0 = wslot(lrow, lch);

    const rsrc_t ares = make_rsrc_bytes(a.A, (long long)a.R * a.lda * 4);
    const rsrc_t ires = make_rsrc_bytes(a.img, a.img_bytes);
    u32 aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + lrow + 64 * i;
        aoff[i] = m < a.R ? (u32)((long long)m * a.lda + (long long)chunk * a.stages * BK + 4 * lch) * 4u : INVALID;
    }
    u32 ichunk = (u32)(tc * a.stages_tile + chunk * a.stages) * (u32)CHUNK;      // next stage's chunk of the image
    u32 ka = 0;                                                                      // next stage's byte offset along the row operand

    // Both operands run D - 1 = 3 stages ahead of the MFMAs: vmcnt counts in order, so a stage-ahead DMA that is waited for at every
    // barrier would drag every older row load with it -- the image chunks and the row loads have to be equally deep.  The rows of
    // a time step were written by the previous kernel (another XCD's L2 or HBM: ~1-2 us away), and with two stages in flight a
    // 16-stage launch spent 1.2 us per stage waiting for them.
    constexpr int D = 4;                            // register sets of the row operand = stage buffers of the image
    f32x4 ra[D][NA];
    auto dma = [&](auto nbc) {                      // the image chunk of the next stage -> LDS
        constexpr int nb = decltype(nbc)::value;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ires, (lds_void_t*)&GBS(nb)[p][wave_u * 128], 16, tid * 16, (int)(ichunk + p * PLANE), 0, 0);
        ichunk += CHUNK;
    };
    auto load_a = [&](auto setc) {                  // rows of the stage after next -> register set (no wait)
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[S][i] = bload4(ares, aoff[i], ka);
        ka += BK * 4;
    };
    auto conv = [&](auto setc, auto nbc, int i) {   // one float4 of register set S -> three planes -> LDS[nb]
        constexpr int S = decltype(setc)::value, nb = decltype(nbc)::value;
        const Split3 sp = split3(ra[S][i]);
#pragma unroll
        for (int p = 0; p < 3; ++p) As[nb][p][aslot0 + 256 * i] = sp.p[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one stage: MFMAs of LDS[b]; the conversion of register set b ^ 1 (the NEXT stage, loaded one step earlier) -> LDS[b ^ 1] runs
    // one dependency level per MFMA gap (both float4 of the thread side by side, 4 to 8 VALU per level: an MFMA holds the pipe for 32
    // cycles = 8 issue slots, and with ONE wave per SIMD nothing else would fill them; a 28-instruction block between two MFMAs
    // idles the pipe instead); all fragments are requested at the top of the stage (one exposed LDS latency, not one per column tile)
    auto step = [&](auto kc) {
        constexpr int k = decltype(kc)::value;       // stage index modulo D: LDS buffer k & 1; register set k is free (its stage went
                                                     // into LDS one step ago), set (k + 1) % D holds the NEXT stage
        constexpr int b = k & 1, nb = b ^ 1, cs = (k + 1) % D;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
        dma(std::integral_constant<int, (k + D - 1) % D>{});       // stage s + D - 1 -> the ring slot stage s - 1 was read from
        __builtin_amdgcn_sched_barrier(0);                         // (issue order is part of the vmcnt arithmetic below)
        load_a(kc);                                                // stage s + D -> set k
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i][p] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(&As[b][p][0])[rslot(wm_off + 32 * i + l31, half)]);
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[jj][p] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(&GBS(k)[p][0])[rslot(wn_off + 32 * jj + l31, half)]);
        float r[8];
        u32 pk[3][4], u[8];
        auto level = [&](int l) {                    // l is a constant after unrolling
            if (l == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = ra[cs][e >> 2][e & 3];
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[0][q] = cvt_pk_bf16(r[2 * q], r[2 * q + 1]);
            } else if (l == 1 || l == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u[2 * q] = pk[l / 3][q] << 16;
                    u[2 * q + 1] = pk[l / 3][q] & 0xffff0000u;
                }
            } else if (l == 2 || l == 5) {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] -= __uint_as_float(u[e]);
            } else if (l == 3 || l == 6) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[l / 3][q] = cvt_pk_bf16(r[2 * q], r[2 * q + 1]);
            } else if (l == 7) {
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) As[nb][p][aslot0 + 256 * i] = u32x2{pk[p][2 * i], pk[p][2 * i + 1]};
            }
        };
        // term t of all the wave's tiles in turn: consecutive MFMAs never share an accumulator (a filler between two MFMAs on the
        // SAME accumulator costs ~40 cycles, MI355X_MICROARCH.md)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = (t * TN + jj) * TM + i;
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[jj][PB[t]], acc[i][jj], 0, 0, 0);
                    if (m >= GAP0 && m < GAP0 + 8) {
                        __builtin_amdgcn_sched_barrier(0);
                        level(m - GAP0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        // End of the stage: this wave's plane stores and ITS pieces of the NEXT stage's image chunk must have landed before anybody
        // reads them.  __syncthreads() would wait for every DMA in flight (its release fence covers all LDS writes: vmcnt(2)), i.e.
        // for the chunk issued a moment ago as well; the chunk of stage s + 1 was issued two steps back, and exactly 12 vector-memory
        // operations of this wave follow it (2 + 3 + 2 + 3 + 2, the order pinned above), so vmcnt(12) is the wait that is needed
        // (the first steps after the prologue have more behind it: the same immediate waits longer there, never shorter).
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x007C);          // vmcnt(12) expcnt(7) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // prologue: stage 0 -> LDS (image by DMA, rows through set 0); image chunks of stages 1, 2 and the rows of stages 1 .. 3 in flight
    dma(S0{});
    load_a(S0{});
    dma(S1{});
    dma(std::integral_constant<int, 2>{});
    load_a(S1{});
    load_a(std::integral_constant<int, 2>{});
    load_a(std::integral_constant<int, 3>{});
    conv(S0{}, S0{}, 0);
    conv(S0{}, S0{}, 1);
    __syncthreads();
    // past the last stage the loads fetch rows / chunks nobody consumes (the buffer descriptors bound them)
    for (int s = 0; s < a.stages; s += D) {
        step(S0{});
        step(S1{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
    }
    __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0): the run-ahead loads / image chunks land before this workgroup's LDS
    asm volatile("" ::: "memory");                   // and registers are given back

    if constexpr (MODE == MODE_FWD) {
        // gate math of torch.nn.GRU (gru_step_fwd_kernel): r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z),
        // n = tanh(gi_n + r * gh_n), h_t = (1 - z) * n + z * h_{t-1}; every lane holds the three pre-activations of its (row, unit) pairs
        const int H = a.H, R = a.R;
        const int j = tc * 32 + l31;
        const float br = a.bhh[j], bz = a.bhh[H + j], bn = a.bhh[2 * H + j];
        const rsrc_t gres = make_rsrc_bytes(a.gi, (long long)R * 3 * H * 4), hres = make_rsrc_bytes(a.A, (long long)R * a.lda * 4);
        const int row0 = m0 + wm_off + 4 * half;
        float gr[16], gz[16], gn[16], hp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            const u32 go = (u32)((row0 + ro) * 3 * H + j) * 4u;
            gr[r] = bload(gres, go, 0u);
            gz[r] = bload(gres, go, (u32)H * 4u);
            gn[r] = bload(gres, go, (u32)H * 8u);
            hp[r] = bload(hres, (u32)((long long)(row0 + ro) * a.lda + j) * 4u, 0u);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            const float rg = sigmoid_s3(gr[r] + (acc[0][0][r] + br));
            const float zg = sigmoid_s3(gz[r] + (acc[0][1][r] + bz));
            const float ghn = acc[0][2][r] + bn;
            const float ng = tanhf(gn[r] + rg * ghn);
            if (row < R) {
                const long long e = (long long)row * H + j;
                float* gp = a.gates + (long long)row * 3 * H + j;
                a.hout[e] = (1.0f - zg) * ng + zg * hp[r];
                gp[0] = rg;
                gp[H] = zg;
                gp[2 * H] = ng;
                a.hn[e] = ghn;
            }
        }
    } else {
        float* P = a.part + (long long)chunk * a.part_stride;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = tc * 128 + wn_off + 32 * j + l31;
                const int row0 = m0 + wm_off + 32 * i + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    if (row < a.R) P[(long long)row * a.H + col] = acc[i][j][r];
                }
            }
    }
#undef GBS
}

bool shapes_ok(int R, int H) { return R > 0 && H >= 128 && H % 128 == 0 && (long long)R * 3 * H <= MAX_ELEMS && 3ll * H * H <= MAX_ELEMS; }

}  // namespace

// bytes of either image of W_hh [3H, H]
extern "C" int64_t dtc_gru_s3_image_bytes(int H) {
    if (H <= 0) return 0;
    const int64_t fwd = (int64_t)Geo<MODE_FWD>::CHUNK * (H / 32) * (H / BK), bwd = (int64_t)Geo<MODE_BWD>::CHUNK * dtc::ceil_div(H, 128) * (3 * H / BK);
    return (fwd > bwd ? fwd : bwd) + 64;
}

// image of W_hh for dtc_gru_step_fwd_s3 (backward = 0) / dtc_gru_dgrad_parts_s3 (backward = 1): once per pass over the T time steps
extern "C" int dtc_gru_s3_image(const float* W_hh, void* img, int H, int backward, void* stream) {
    DTC_REQUIRE(W_hh && img && dtc::aligned16(img) && shapes_ok(1, H), "bad arguments (H = %d must be a multiple of 128)", H);
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof("wimage", 0.0, s, 30.0 * H * (double)H);
    if (backward) {
        const int st = 3 * H / BK;
        hipLaunchKernelGGL(gru_wimage_kernel<MODE_BWD>, dim3((unsigned)((H / 128) * st)), dim3(256), 0, s, W_hh, (u32x4*)img, H, st);
    } else {
        const int st = H / BK;
        hipLaunchKernelGGL(gru_wimage_kernel<MODE_FWD>, dim3((unsigned)((H / 32) * st)), dim3(256), 0, s, W_hh, (u32x4*)img, H, st);
    }
    return dtc::check_launch("gru_s3_image");
}

// dtc_gru_step_fwd on the split-precision path; `img` = dtc_gru_s3_image(W_hh, backward = 0)
namespace {
// DTC_GRU_XCD_COLS=0: the row-tile map of the general GEMM kernels (map_tile)
int gru_colmap() {
    constexpr int on = 1;
    return on;
}
int fwd_args(GruS3Args& a, const float* hprev, const void* img, const float* b_hh, const float* gi_t, float* hout, float* gates_t,
             float* hn_t, int R, int H) {
    DTC_REQUIRE(shapes_ok(R, H), "bad shape R=%d H=%d (H must be a multiple of 128)", R, H);
    DTC_REQUIRE(hprev && img && b_hh && gi_t && hout && gates_t && hn_t, "null pointer");
    a = GruS3Args{};
    a.A = hprev;
    a.lda = H;
    a.img = (const u32x4*)img;
    a.img_bytes = (long long)Geo<MODE_FWD>::CHUNK * (H / 32) * (H / BK);
    a.R = R;
    a.H = H;
    a.stages = a.stages_tile = H / BK;
    a.bhh = b_hh;
    a.gi = gi_t;
    a.hout = hout;
    a.gates = gates_t;
    a.hn = hn_t;
    a.colmap = gru_colmap();
    a.nparts = 1;
    return DTC_OK;
}
int launch_fwd(const GruS3Args& a0, const GruS3Args& a1, int count, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int R = a0.R, H = a0.H;
    dtc::ProfScope prof(dtc::prof_shape_name("gru_step_fwd", R, 3 * H, H), count * 2.0 * R * 3.0 * H * H, s);
    const unsigned gx = a0.colmap ? (unsigned)(8 * dtc::ceil_div(H / 32, 8) * dtc::ceil_div(R, BM)) : (unsigned)grid_for((int)dtc::ceil_div(R, BM), H / 32);
    hipLaunchKernelGGL(gru_s3_kernel<MODE_FWD>, dim3(gx, 1, (unsigned)count), dim3(256), 0, s, a0, a1);
    return dtc::check_launch("gru_step_fwd_s3");
}
}  // namespace
extern "C" int dtc_gru_step_fwd_s3(const float* hprev, const void* img, const float* b_hh, const float* gi_t, float* hout, float* gates_t,
                                   float* hn_t, int R, int H, void* stream) {
    GruS3Args a;
    int rc = fwd_args(a, hprev, img, b_hh, gi_t, hout, gates_t, hn_t, R, H);
    if (rc != DTC_OK) return rc;
    return launch_fwd(a, a, 1, stream);
}
// the same time step of TWO recurrences of one shape in one launch (dtc_gru_fwd_multi)
int dtc_gru_step_fwd_s3_pair(const float* const* hprev, const void* const* img, const float* const* b_hh, const float* const* gi_t,
                             float* const* hout, float* const* gates_t, float* const* hn_t, int R, int H, void* stream) {
    GruS3Args a[2];
    for (int i = 0; i < 2; ++i) {
        int rc = fwd_args(a[i], hprev[i], img[i], b_hh[i], gi_t[i], hout[i], gates_t[i], hn_t[i], R, H);
        if (rc != DTC_OK) return rc;
    }
    return launch_fwd(a[0], a[1], 2, stream);
}

// the `nparts` chunks of dgh_t [R, 3H] W_hh [3H, H] side by side: chunk c -> part + c * part_stride ([R, H]); the caller adds
// them in a fixed order; `img` = dtc_gru_s3_image(W_hh, backward = 1); (3H / nparts) must be a multiple of 64
namespace {
int bwd_args(GruS3Args& a, const float* dgh_t, const void* img, float* part, int64_t part_stride, int R, int H, int nparts) {
    DTC_REQUIRE(shapes_ok(R, H) && nparts >= 1 && (3 * H) % nparts == 0 && (3 * H / nparts) % (4 * BK) == 0, "bad shape R=%d H=%d nparts=%d", R, H, nparts);
    DTC_REQUIRE(dgh_t && img && part && part_stride >= (int64_t)R * H, "null pointer / overlapping chunks");
    a = GruS3Args{};
    a.A = dgh_t;
    a.lda = 3 * H;
    a.img = (const u32x4*)img;
    a.img_bytes = (long long)Geo<MODE_BWD>::CHUNK * (H / 128) * (3 * H / BK);
    a.R = R;
    a.H = H;
    a.stages = 3 * H / nparts / BK;
    a.stages_tile = 3 * H / BK;
    a.part = part;
    a.part_stride = part_stride;
    a.colmap = gru_colmap();
    a.nparts = nparts;
    return DTC_OK;
}
int launch_bwd(const GruS3Args& a0, const GruS3Args& a1, int count, int nparts, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int R = a0.R, H = a0.H;
    dtc::ProfScope prof(dtc::prof_shape_name("linear_dgrad", R, 3 * H, H), count * 2.0 * R * 3.0 * H * H, s,
                        count * 4.0 * ((double)R * 3 * H + 3.0 * H * H + (double)nparts * R * H));
    const dim3 grid = a0.colmap ? dim3((unsigned)(8 * dtc::ceil_div((H / 128) * nparts, 8) * dtc::ceil_div(R, BM)), 1, (unsigned)count)
                                : dim3((unsigned)grid_for((int)dtc::ceil_div(R, BM), H / 128), (unsigned)nparts, (unsigned)count);
    hipLaunchKernelGGL(gru_s3_kernel<MODE_BWD>, grid, dim3(256), 0, s, a0, a1);
    return dtc::check_launch("gru_dgrad_parts_s3");
}
}  // namespace
extern "C" int dtc_gru_dgrad_parts_s3(const float* dgh_t, const void* img, float* part, int64_t part_stride, int R, int H, int nparts,
                                      void* stream) {
    GruS3Args a;
    int rc = bwd_args(a, dgh_t, img, part, part_stride, R, H, nparts);
    if (rc != DTC_OK) return rc;
    return launch_bwd(a, a, 1, nparts, stream);
}
int dtc_gru_dgrad_parts_s3_pair(const float* const* dgh_t, const void* const* img, float* const* part, int64_t part_stride, int R, int H,
                                int nparts, void* stream) {
    GruS3Args a[2];
    for (int i = 0; i < 2; ++i) {
        int rc = bwd_args(a[i], dgh_t[i], img[i], part[i], part_stride, R, H, nparts);
        if (rc != DTC_OK) return rc;
    }
    return launch_bwd(a[0], a[1], 2, nparts, stream);
}
