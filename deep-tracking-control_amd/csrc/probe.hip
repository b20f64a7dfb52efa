// Measurement aid (round 4): the rate the matrix pipe SUSTAINS on this chip for the split-precision kernels' MFMA stream -- 24
// v_mfma_f32_32x32x16_bf16 per wave and stage on four accumulators, three waves per SIMD, operands in registers, nothing else in the
// loop -- as a function of the operand bits.  MI355X clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"): the same
// instruction stream runs ~20 % faster on all-zero operands than on the dense bit patterns the low-order planes of a bf16 x 3 split are
// made of, so the dense-bf16 peak of the data sheet (2.5 PFLOP/s = 419 TFLOP/s of fp32-equivalent work at six passes) is not what a
// kernel can reach on real data.  bench.py times this next to the GEMM family and reports both roofs.
#include "s3_core.hpp"

namespace {

// ORDER: the sequence of the six plane pairs of a product.  0 = the kernels' order (smallest terms first); 1 = B-stationary (b1 x3, b2 x2,
// b3); 2 = A-stationary -- does keeping one operand's bits on the pipe's inputs for consecutive instructions change what it sustains?
template <int ORDER>
__global__ __launch_bounds__(256, 3) void mfma_stream_kernel(const u32x4* __restrict__ operands, int iters, float* __restrict__ sink) {
    const int tid = threadIdx.x;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[i][p] = __builtin_bit_cast(bf16x8, operands[((i * 3 + p) * 256 + tid) & 4095]);
            b[i][p] = __builtin_bit_cast(bf16x8, operands[((6 + i * 3 + p) * 256 + tid) & 4095]);
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int PA[6] = {2, ORDER == 0 ? 1 : ORDER == 1 ? 1 : 1, ORDER == 0 ? 0 : ORDER == 1 ? 0 : 1, ORDER == 0 ? 1 : ORDER == 1 ? 1 : 0, 0, 0};
    constexpr int PB[6] = {0, ORDER == 0 ? 1 : ORDER == 1 ? 0 : 0, ORDER == 0 ? 2 : ORDER == 1 ? 0 : 1, ORDER == 0 ? 0 : ORDER == 1 ? 1 : 0, 1, ORDER == 2 ? 2 : ORDER == 1 ? 2 : 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[0] = s;                       // keeps the loop alive, never true for the operands used
}

// the two-term fp16 kernels' stream: 12 v_mfma_f32_32x32x16_f16 per wave and stage (3 passes x 2 x 2 tiles) on four accumulators
__global__ __launch_bounds__(256, 3) void mfma_stream_h2_kernel(const u32x4* __restrict__ operands, int iters, float* __restrict__ sink) {
    const int tid = threadIdx.x;
    u32x4 a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            a[i][p] = operands[((i * 2 + p) * 256 + tid) & 4095];
            b[i][p] = operands[((4 + i * 2 + p) * 256 + tid) & 4095];
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = Prec<true>::mfma(a[i][Prec<true>::pa(t)], b[j][Prec<true>::pb(t)], acc[i][j]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[0] = s;
}

}  // namespace

// the fp16 stream: blocks * 4 * iters * 12 * 32768 fp16 FLOP = that / 3 of fp32-equivalent work
extern "C" int dtc_probe_mfma_stream_h2(const void* operands, int blocks, int iters, float* sink, void* stream) {
    DTC_REQUIRE(operands && sink && blocks > 0 && iters > 0 && dtc::aligned16(operands), "null / unaligned pointer or bad size");
    hipLaunchKernelGGL(mfma_stream_h2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, iters, sink);
    return dtc::check_launch("probe_mfma_stream_h2");
}

// One launch of the bare MFMA stream: `blocks` workgroups of 4 waves, each `iters` stages of 24 MFMAs (6 passes x 2 x 2 tiles of
// 32 x 32 x 16): blocks * 4 * iters * 24 * 32768 bf16 FLOP = that / 6 of fp32-equivalent work.  operands: 64 KiB of device memory
// whose bits are the MFMA inputs (zeros, or random bf16 patterns).
extern "C" int dtc_probe_mfma_stream(const void* operands, int blocks, int iters, float* sink, void* stream) {
    DTC_REQUIRE(operands && sink && blocks > 0 && iters > 0 && dtc::aligned16(operands), "null / unaligned pointer or bad size");
    static const int order = getenv("DTC_PROBE_ORDER") ? atoi(getenv("DTC_PROBE_ORDER")) : 0;
    if (order == 1) hipLaunchKernelGGL(mfma_stream_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, iters, sink);
    else if (order == 2) hipLaunchKernelGGL(mfma_stream_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, iters, sink);
    else hipLaunchKernelGGL(mfma_stream_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, iters, sink);
    return dtc::check_launch("probe_mfma_stream");
}
