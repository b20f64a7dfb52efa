// amax records of the two-term fp16 GEMM path (csrc/s3_core.hpp, include/dtc_hip.h): shared by every kernel that publishes the largest
// |value| it writes (the GEMM epilogues, dtc_pack_cols, the fused PPO heads) or reads one.
#pragma once
#include <hip/hip_runtime.h>

namespace {

typedef unsigned int amax_u32;
__device__ __forceinline__ amax_u32 abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// An amax slot is a RECORD of AMAX_SUB words, one per 128-byte line: a kernel's ~3000 waves reach their epilogue together, and that
// many atomic maxima on ONE address cost ~20 us (measured: 65 instead of 46 us for the 24576 x 512 x 512 forward layer); spread over
// 16 lines (wave -> line by workgroup and wave index) they cost nothing measurable.  Readers take the maximum of the 16 words.
constexpr int AMAX_SUB = 16, AMAX_STRIDE = 32, AMAX_RECORD_BYTES = AMAX_SUB * AMAX_STRIDE * 4;
// wave-wide maximum of one value per lane (every lane of the wave must take part)
__device__ __forceinline__ amax_u32 wave_max_u32(amax_u32 m) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const amax_u32 o = (amax_u32)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    return m;
}
// the record's value: lane i < 16 loads word i, the wave reduces (one load instruction per wave instead of 16; every lane of the wave
// must call it).  Device-scope loads: the words were written by atomics of other kernels
__device__ __forceinline__ amax_u32 amax_read(const amax_u32* rec) {
    const int lane = threadIdx.x & 63;
    const amax_u32 v = lane < AMAX_SUB ? __hip_atomic_load(rec + lane * AMAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    return wave_max_u32(v);
}
// wave-wide maximum of the lanes' |value| bit patterns -> the tensor's amax record (skipped when the line already holds as much)
__device__ __forceinline__ void amax_publish(amax_u32* rec, amax_u32 m) {
    if (rec == nullptr) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const amax_u32 o = (amax_u32)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    amax_u32* slot = rec + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (AMAX_SUB - 1)) * AMAX_STRIDE;
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
}

// the same for a whole 256-thread workgroup: ONE atomic per workgroup (every thread of the workgroup must call it; `lds4`: four words of
// LDS nobody else uses across the call).  A GEMM launch publishes ~800 values instead of ~3000 -- the waves of a launch reach their
// epilogues together and nearly all of them see a slot that is still zero, so the filter above does not thin them out.
__device__ __forceinline__ void amax_publish_block(amax_u32* rec, amax_u32 m, amax_u32* lds4) {
    if (rec == nullptr) return;                           // (uniform)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const amax_u32 o = (amax_u32)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) m = lds4[w] > m ? lds4[w] : m;
        amax_u32* slot = rec + (blockIdx.x & (AMAX_SUB - 1)) * AMAX_STRIDE;
        if (m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
    }
}

}  // namespace
