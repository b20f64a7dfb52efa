// Ceiling probe for the fp32 MFMA dense-layer kernels (gfx950): how much of the v_mfma_f32_32x32x2_f32
// peak survives each ingredient of the block loop of csrc/gemm.hip.  Stand-alone (no torch):
//   hipcc -O3 --offload-arch=gfx950 mfma_ceiling.hip -o mfma_ceiling && ./mfma_ceiling
// Variants (cumulative):  0 MFMA on register operands | 1 + LDS fragment reads | 2 + one barrier per K step
//                         3 + LDS tile writes         | 4 + buffer loads of the next tile (L2 resident)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int BM = 128, BK = 16;

template <int V, int BN, int PAD, int OCC, bool EPI>
__global__ __launch_bounds__(256, OCC) void probe(const float* __restrict__ src, float* __restrict__ out, int steps) {
    constexpr int WM = (BN == 128) ? 2 : 4, WN = 4 / WM, TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int LDA = BM + PAD, LDB = BN + PAD;
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm_off = (wave / WN) * (32 * TM), wn_off = (wave % WN) * (32 * TN);
    const int half = lane >> 5, l31 = lane & 31;
    const int kk = tid & 15, rbase = tid >> 4;
    constexpr int NA = BM / 16, NB = BN / 16;
    for (int i = tid; i < 2 * BK * LDA; i += 256) (&As[0][0][0])[i] = 1.0f + i * 1e-6f;
    for (int i = tid; i < 2 * BK * LDB; i += 256) (&Bs[0][0][0])[i] = 1.0f - i * 1e-6f;
    __syncthreads();
    rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 30, 0x00020000);
    u32 aoff[NA], boff[NB];
    for (int i = 0; i < NA; ++i) aoff[i] = (u32)(((blockIdx.x & 63) * 128 + rbase + 16 * i) * 512 + kk) * 4u;
    for (int i = 0; i < NB; ++i) boff[i] = (u32)((rbase + 16 * i) * 512 + kk) * 4u;
    int tile_r = blockIdx.x >> 3, tile_c = blockIdx.x & 7;
    if (V >= 5) {   // real operands: X[24576][512] at src, W[512][512] behind it, XCD-aware tile map of gemm.hip
        const int col_tiles = 512 / BN, xcd = blockIdx.x & 7, j = blockIdx.x >> 3, local = j / col_tiles;
        tile_c = j - local * col_tiles;
        tile_r = xcd + 8 * local;
        for (int i = 0; i < NA; ++i) aoff[i] = (u32)((tile_r * 128 + rbase + 16 * i) * 512 + kk) * 4u;
        for (int i = 0; i < NB; ++i) boff[i] = (u32)(24576 * 512 + (tile_c * BN + rbase + 16 * i) * 512 + kk) * 4u;
    }

    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[NA], rb[NB];
    for (int i = 0; i < NA; ++i) ra[i] = 0.f;
    for (int i = 0; i < NB; ++i) rb[i] = 0.f;
    float a0 = 1.0f + lane, b0 = 2.0f - lane;

    for (int t = 0; t < steps; ++t) {
        const int buf = t & 1;
        if (V >= 4) {
            const u32 soff = (u32)(((V >= 5 ? t + 1 : t) & 31) * 16) * 4u;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                ra[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res, (int)aoff[i], (int)soff, 0));
#pragma unroll
            for (int i = 0; i < NB; ++i)
                rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res, (int)boff[i], (int)soff, 0));
        }
#ifdef PROBE_FENCE
        __builtin_amdgcn_sched_barrier(0);      // keep the global loads ahead of the MFMA phase
#endif
        if (V == 0) {
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i][j], 0, 0, 0);
        } else {
#ifdef PROBE_SETPRIO
            __builtin_amdgcn_s_setprio(PROBE_SETPRIO);
#endif
#ifdef PROBE_IGLP
            __builtin_amdgcn_iglp_opt(PROBE_IGLP);
#endif
            const float* ap = &As[buf][0][0] + half * LDA + wm_off + l31;
            const float* bp = &Bs[buf][0][0] + half * LDB + wn_off + l31;
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = ap[2 * kp * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = bp[2 * kp * LDB + 32 * j];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
#ifdef PROBE_SETPRIO
        if (V >= 1) __builtin_amdgcn_s_setprio(0);
#endif
        if (V >= 3) {
#pragma unroll
            for (int i = 0; i < NA; ++i) As[buf ^ 1][kk][rbase + 16 * i] = ra[i] + 1.0f;
#pragma unroll
            for (int i = 0; i < NB; ++i) Bs[buf ^ 1][kk][rbase + 16 * i] = rb[i] + 1.0f;
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (EPI) {   // the forward kernel's epilogue: bias + ReLU + 32 row-strided stores per 32x32 tile
        float* yp = out + (size_t)tile_r * 128 * 512 + tile_c * BN;
        for (int j = 0; j < TN; ++j)
            for (int i = 0; i < TM; ++i)
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2) + 4 * half + wm_off + 32 * i;
                    const float v = acc[i][j][r] + a0;
                    yp[(size_t)ro * 512 + wn_off + 32 * j + l31] = v > 0.f ? v : 0.f;
                }
    } else if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int V, int BN, int PAD, int OCC, bool EPI = false>
void run(const float* src, float* out, const char* tag, int steps = 2048, int blocks = 256 * OCC * 4) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<V, BN, PAD, OCC, EPI><<<blocks, 256>>>(src, out, 64);
    hipDeviceSynchronize();
    const int reps = steps >= 1024 ? 1 : 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) probe<V, BN, PAD, OCC, EPI><<<blocks, 256>>>(src, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * BM * BN * BK * (double)steps * blocks * reps;
    printf("%-28s V=%d BN=%3d PAD=%d OCC=%d steps=%4d blocks=%5d epi=%d : %8.3f ms  %7.1f TFLOP/s\n", tag, V, BN, PAD, OCC, steps, blocks, (int)EPI, ms / reps, flop / ms * 1e-9);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 << 20);
    hipMemset(src, 0, 64 << 20);
    {
        float* h = (float*)malloc(52 << 20);
        unsigned long long st = 88172645463325252ull;     // xorshift -> sum of 4 uniforms (bell-shaped, full mantissas)
        for (size_t i = 0; i < (52u << 20) / 4; ++i) {
            float v = -2.0f;
            for (int q = 0; q < 4; ++q) {
                st ^= st << 13; st ^= st >> 7; st ^= st << 17;
                v += (float)(st >> 40) * (1.0f / 16777216.0f);
            }
            h[i] = v * 1.7f;
        }
        hipMemcpy(src, h, 52 << 20, hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&out, 64 << 20);
    run<0, 64, 4, 4>(src, out, "mfma only");
    run<0, 128, 4, 2>(src, out, "mfma only");
    run<1, 64, 4, 4>(src, out, "+lds reads");
    run<2, 64, 4, 4>(src, out, "+barrier");
    run<3, 64, 4, 4>(src, out, "+lds writes (2-way)");
    run<3, 64, 2, 4>(src, out, "+lds writes (pad 2)");
    run<4, 64, 4, 4>(src, out, "+buffer loads");
    run<4, 64, 2, 4>(src, out, "+buffer loads (pad 2)");
    run<4, 64, 2, 3>(src, out, "+buffer loads (pad 2)");
    run<1, 128, 4, 2>(src, out, "+lds reads");
    run<2, 128, 4, 2>(src, out, "+barrier");
    run<3, 128, 4, 2>(src, out, "+lds writes");
    run<4, 128, 4, 2>(src, out, "+buffer loads");
    run<4, 128, 2, 2>(src, out, "+buffer loads (pad 2)");
    run<4, 128, 4, 3>(src, out, "+buffer loads");
    // the real launch geometry of a 24576 x 512 x 512 layer: 32 K steps per block, 1536 (BN=64) / 768 (BN=128) blocks
    run<4, 64, 4, 4>(src, out, "short blocks", 32, 1536);
    run<4, 64, 4, 4, true>(src, out, "short blocks + epilogue", 32, 1536);
    run<4, 64, 4, 4, true>(src, out, "short blocks + epilogue", 32, 2048);
    run<4, 64, 4, 4, true>(src, out, "short blocks + epilogue", 44, 1536);
    run<4, 128, 4, 2>(src, out, "short blocks", 32, 768);
    run<4, 128, 4, 2, true>(src, out, "short blocks + epilogue", 32, 768);
    run<4, 128, 4, 3, true>(src, out, "short blocks + epilogue", 32, 768);
    run<4, 64, 4, 4, true>(src, out, "short blocks + epilogue", 128, 1536);
    run<4, 64, 4, 4, true>(src, out, "short blocks + epilogue", 512, 1536);
    run<5, 64, 2, 4, true>(src, out, "real operands (pad 2)", 32, 1536);
    run<5, 64, 2, 4, true>(src, out, "real operands (pad 2)", 44, 1536);
    run<5, 64, 4, 4, true>(src, out, "real operands", 32, 1536);
    run<5, 64, 4, 3, true>(src, out, "real operands", 32, 1536);
    run<5, 128, 4, 2, true>(src, out, "real operands", 32, 768);
    run<5, 128, 4, 3, true>(src, out, "real operands", 32, 768);
    // tail / balance: one exact wave of blocks, and 2 tiles' worth of K per block at 3 blocks per CU
    run<5, 64, 4, 4, true>(src, out, "one wave", 32, 1024);
    run<5, 64, 4, 4, true>(src, out, "one wave, 1.5x K", 48, 1024);
    run<5, 64, 4, 3, true>(src, out, "3/CU, 2x K", 64, 768);
    run<5, 64, 4, 2, true>(src, out, "2/CU, 3x K", 96, 512);
    run<5, 64, 4, 4, true>(src, out, "1.25 waves", 32, 1280);
    run<5, 64, 4, 6, true>(src, out, "occ 6", 32, 1536);
    return 0;
}
