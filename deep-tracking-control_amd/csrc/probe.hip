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

// Debugging aid: leaves `pattern` in all 64 KiB of LDS a workgroup can own and in ~240 VGPRs of every lane -- whatever the NEXT kernel on
// this CU reads without having written it (LDS words, registers) then depends on `pattern` instead of on the previous tenant.
__global__ __launch_bounds__(256) void poison_kernel(unsigned pattern, float* __restrict__ sink) {
    __shared__ unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = pattern;
    const float f = __uint_as_float(pattern);
    __syncthreads();
    // v8 .. v247 by name: the register allocator would keep an array of 240 values in far fewer registers
    asm volatile("v_mov_b32 v8, %0\n" "v_mov_b32 v9, %0\n" "v_mov_b32 v10, %0\n" "v_mov_b32 v11, %0\n" "v_mov_b32 v12, %0\n" "v_mov_b32 v13, %0\n" "v_mov_b32 v14, %0\n" "v_mov_b32 v15, %0\n" "v_mov_b32 v16, %0\n" "v_mov_b32 v17, %0\n" "v_mov_b32 v18, %0\n" "v_mov_b32 v19, %0\n" "v_mov_b32 v20, %0\n" "v_mov_b32 v21, %0\n" "v_mov_b32 v22, %0\n" "v_mov_b32 v23, %0\n" "v_mov_b32 v24, %0\n" "v_mov_b32 v25, %0\n" "v_mov_b32 v26, %0\n" "v_mov_b32 v27, %0\n" "v_mov_b32 v28, %0\n" "v_mov_b32 v29, %0\n" "v_mov_b32 v30, %0\n" "v_mov_b32 v31, %0\n" "v_mov_b32 v32, %0\n" "v_mov_b32 v33, %0\n" "v_mov_b32 v34, %0\n" "v_mov_b32 v35, %0\n" "v_mov_b32 v36, %0\n" "v_mov_b32 v37, %0\n" "v_mov_b32 v38, %0\n" "v_mov_b32 v39, %0\n" "v_mov_b32 v40, %0\n" "v_mov_b32 v41, %0\n" "v_mov_b32 v42, %0\n" "v_mov_b32 v43, %0\n" "v_mov_b32 v44, %0\n" "v_mov_b32 v45, %0\n" "v_mov_b32 v46, %0\n" "v_mov_b32 v47, %0\n" "v_mov_b32 v48, %0\n" "v_mov_b32 v49, %0\n" "v_mov_b32 v50, %0\n" "v_mov_b32 v51, %0\n" "v_mov_b32 v52, %0\n" "v_mov_b32 v53, %0\n" "v_mov_b32 v54, %0\n" "v_mov_b32 v55, %0\n" "v_mov_b32 v56, %0\n" "v_mov_b32 v57, %0\n" "v_mov_b32 v58, %0\n" "v_mov_b32 v59, %0\n" "v_mov_b32 v60, %0\n" "v_mov_b32 v61, %0\n" "v_mov_b32 v62, %0\n" "v_mov_b32 v63, %0\n" "v_mov_b32 v64, %0\n" "v_mov_b32 v65, %0\n" "v_mov_b32 v66, %0\n" "v_mov_b32 v67, %0\n" "v_mov_b32 v68, %0\n" "v_mov_b32 v69, %0\n" "v_mov_b32 v70, %0\n" "v_mov_b32 v71, %0\n" "v_mov_b32 v72, %0\n" "v_mov_b32 v73, %0\n" "v_mov_b32 v74, %0\n" "v_mov_b32 v75, %0\n" "v_mov_b32 v76, %0\n" "v_mov_b32 v77, %0\n" "v_mov_b32 v78, %0\n" "v_mov_b32 v79, %0\n" "v_mov_b32 v80, %0\n" "v_mov_b32 v81, %0\n" "v_mov_b32 v82, %0\n" "v_mov_b32 v83, %0\n" "v_mov_b32 v84, %0\n" "v_mov_b32 v85, %0\n" "v_mov_b32 v86, %0\n" "v_mov_b32 v87, %0\n" "v_mov_b32 v88, %0\n" "v_mov_b32 v89, %0\n" "v_mov_b32 v90, %0\n" "v_mov_b32 v91, %0\n" "v_mov_b32 v92, %0\n" "v_mov_b32 v93, %0\n" "v_mov_b32 v94, %0\n" "v_mov_b32 v95, %0\n" "v_mov_b32 v96, %0\n" "v_mov_b32 v97, %0\n" "v_mov_b32 v98, %0\n" "v_mov_b32 v99, %0\n" "v_mov_b32 v100, %0\n" "v_mov_b32 v101, %0\n" "v_mov_b32 v102, %0\n" "v_mov_b32 v103, %0\n" "v_mov_b32 v104, %0\n" "v_mov_b32 v105, %0\n" "v_mov_b32 v106, %0\n" "v_mov_b32 v107, %0\n" "v_mov_b32 v108, %0\n" "v_mov_b32 v109, %0\n" "v_mov_b32 v110, %0\n" "v_mov_b32 v111, %0\n" "v_mov_b32 v112, %0\n" "v_mov_b32 v113, %0\n" "v_mov_b32 v114, %0\n" "v_mov_b32 v115, %0\n" "v_mov_b32 v116, %0\n" "v_mov_b32 v117, %0\n" "v_mov_b32 v118, %0\n" "v_mov_b32 v119, %0\n" "v_mov_b32 v120, %0\n" "v_mov_b32 v121, %0\n" "v_mov_b32 v122, %0\n" "v_mov_b32 v123, %0\n" "v_mov_b32 v124, %0\n" "v_mov_b32 v125, %0\n" "v_mov_b32 v126, %0\n" "v_mov_b32 v127, %0\n" "v_mov_b32 v128, %0\n" "v_mov_b32 v129, %0\n" "v_mov_b32 v130, %0\n" "v_mov_b32 v131, %0\n" "v_mov_b32 v132, %0\n" "v_mov_b32 v133, %0\n" "v_mov_b32 v134, %0\n" "v_mov_b32 v135, %0\n" "v_mov_b32 v136, %0\n" "v_mov_b32 v137, %0\n" "v_mov_b32 v138, %0\n" "v_mov_b32 v139, %0\n" "v_mov_b32 v140, %0\n" "v_mov_b32 v141, %0\n" "v_mov_b32 v142, %0\n" "v_mov_b32 v143, %0\n" "v_mov_b32 v144, %0\n" "v_mov_b32 v145, %0\n" "v_mov_b32 v146, %0\n" "v_mov_b32 v147, %0\n" "v_mov_b32 v148, %0\n" "v_mov_b32 v149, %0\n" "v_mov_b32 v150, %0\n" "v_mov_b32 v151, %0\n" "v_mov_b32 v152, %0\n" "v_mov_b32 v153, %0\n" "v_mov_b32 v154, %0\n" "v_mov_b32 v155, %0\n" "v_mov_b32 v156, %0\n" "v_mov_b32 v157, %0\n" "v_mov_b32 v158, %0\n" "v_mov_b32 v159, %0\n" "v_mov_b32 v160, %0\n" "v_mov_b32 v161, %0\n" "v_mov_b32 v162, %0\n" "v_mov_b32 v163, %0\n" "v_mov_b32 v164, %0\n" "v_mov_b32 v165, %0\n" "v_mov_b32 v166, %0\n" "v_mov_b32 v167, %0\n" "v_mov_b32 v168, %0\n" "v_mov_b32 v169, %0\n" "v_mov_b32 v170, %0\n" "v_mov_b32 v171, %0\n" "v_mov_b32 v172, %0\n" "v_mov_b32 v173, %0\n" "v_mov_b32 v174, %0\n" "v_mov_b32 v175, %0\n" "v_mov_b32 v176, %0\n" "v_mov_b32 v177, %0\n" "v_mov_b32 v178, %0\n" "v_mov_b32 v179, %0\n" "v_mov_b32 v180, %0\n" "v_mov_b32 v181, %0\n" "v_mov_b32 v182, %0\n" "v_mov_b32 v183, %0\n" "v_mov_b32 v184, %0\n" "v_mov_b32 v185, %0\n" "v_mov_b32 v186, %0\n" "v_mov_b32 v187, %0\n" "v_mov_b32 v188, %0\n" "v_mov_b32 v189, %0\n" "v_mov_b32 v190, %0\n" "v_mov_b32 v191, %0\n" "v_mov_b32 v192, %0\n" "v_mov_b32 v193, %0\n" "v_mov_b32 v194, %0\n" "v_mov_b32 v195, %0\n" "v_mov_b32 v196, %0\n" "v_mov_b32 v197, %0\n" "v_mov_b32 v198, %0\n" "v_mov_b32 v199, %0\n" "v_mov_b32 v200, %0\n" "v_mov_b32 v201, %0\n" "v_mov_b32 v202, %0\n" "v_mov_b32 v203, %0\n" "v_mov_b32 v204, %0\n" "v_mov_b32 v205, %0\n" "v_mov_b32 v206, %0\n" "v_mov_b32 v207, %0\n" "v_mov_b32 v208, %0\n" "v_mov_b32 v209, %0\n" "v_mov_b32 v210, %0\n" "v_mov_b32 v211, %0\n" "v_mov_b32 v212, %0\n" "v_mov_b32 v213, %0\n" "v_mov_b32 v214, %0\n" "v_mov_b32 v215, %0\n" "v_mov_b32 v216, %0\n" "v_mov_b32 v217, %0\n" "v_mov_b32 v218, %0\n" "v_mov_b32 v219, %0\n" "v_mov_b32 v220, %0\n" "v_mov_b32 v221, %0\n" "v_mov_b32 v222, %0\n" "v_mov_b32 v223, %0\n" "v_mov_b32 v224, %0\n" "v_mov_b32 v225, %0\n" "v_mov_b32 v226, %0\n" "v_mov_b32 v227, %0\n" "v_mov_b32 v228, %0\n" "v_mov_b32 v229, %0\n" "v_mov_b32 v230, %0\n" "v_mov_b32 v231, %0\n" "v_mov_b32 v232, %0\n" "v_mov_b32 v233, %0\n" "v_mov_b32 v234, %0\n" "v_mov_b32 v235, %0\n" "v_mov_b32 v236, %0\n" "v_mov_b32 v237, %0\n" "v_mov_b32 v238, %0\n" "v_mov_b32 v239, %0\n" "v_mov_b32 v240, %0\n" "v_mov_b32 v241, %0\n" "v_mov_b32 v242, %0\n" "v_mov_b32 v243, %0\n" "v_mov_b32 v244, %0\n" "v_mov_b32 v245, %0\n" "v_mov_b32 v246, %0\n" "v_mov_b32 v247, %0\n" : : "v"(f) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
    const float s = f;
    if (s == 123.456f && lds[threadIdx.x] == 77u) sink[0] = s;
}

}  // namespace

extern "C" int dtc_probe_poison(uint32_t pattern, int blocks, float* sink, void* stream) {
    DTC_REQUIRE(sink && blocks > 0, "null pointer or bad size");
    hipLaunchKernelGGL(poison_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pattern, sink);
    return dtc::check_launch("probe_poison");
}

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
    constexpr int order = 0;
    if (order == 1) hipLaunchKernelGGL(mfma_stream_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, iters, sink);
    else if (order == 2) hipLaunchKernelGGL(mfma_stream_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, iters, sink);
    else hipLaunchKernelGGL(mfma_stream_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)operands, iters, sink);
    return dtc::check_launch("probe_mfma_stream");
}
