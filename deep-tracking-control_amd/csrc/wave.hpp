// 64-lane wavefront reductions shared by the env-step kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// 64-lane reductions on DPP (no LDS round trips): quad_perm xor1, xor2, row_half_mirror, row_mirror leave the
// 16-lane row total in every lane of the row; row_bcast15 / row_bcast31 fold the four rows into lane 63.
// For a sum this is exactly the xor-butterfly with offsets 1,2,4,8,16,32 (the order oracle/foothold.py uses):
// every step adds two values that are uniform over the sub-group they came from, and a+b == b+a bitwise.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = v + dpp_f<0xB1, 0xF>(0.f, v);
    v = v + dpp_f<0x4E, 0xF>(0.f, v);
    v = v + dpp_f<0x141, 0xF>(0.f, v);
    v = v + dpp_f<0x140, 0xF>(0.f, v);
    v = v + dpp_f<0x142, 0xA>(0.f, v);
    v = v + dpp_f<0x143, 0xC>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min(float v) {
    const float inf = __builtin_inff();
    v = fminf(v, dpp_f<0xB1, 0xF>(inf, v));
    v = fminf(v, dpp_f<0x4E, 0xF>(inf, v));
    v = fminf(v, dpp_f<0x141, 0xF>(inf, v));
    v = fminf(v, dpp_f<0x140, 0xF>(inf, v));
    v = fminf(v, dpp_f<0x142, 0xA>(inf, v));
    v = fminf(v, dpp_f<0x143, 0xC>(inf, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Correctly rounded sqrtf for the hot path.  hipcc's expansion of sqrtf (-fhip-fp32-correctly-rounded-divide-sqrt) is
// v_sqrt_f32 + "try the two neighbours, keep the one whose residual changes sign" wrapped in a 2^32 pre-scale for
// x < 2^-96 and a zero / inf fix-up (16 instructions).  This is the same core without the wrapper (10): the residual
// test is exact for every x >= 2^-96 and leaves +-0, +inf, NaN and negative inputs as v_sqrt_f32 returns them (all
// comparisons against the NaN residuals are false); a wave in which ANY lane holds a non-zero |x| < 2^-96 takes the
// compiler's sqrtf instead.  tools/probes/sqrt_exact.hip checks all 2^32 inputs against sqrtf on the device.
__device__ __forceinline__ float sqrt_rn(float x) {
    const unsigned xb = (unsigned)__float_as_int(x);
    if (__builtin_amdgcn_ballot_w64((xb << 1) - 2u < (0x0F800000u << 1) - 2u) != 0ull) return sqrtf(x);   // 0 < |x| < 2^-96
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
    float r = em <= 0.0f ? sm : s;
    r = ep > 0.0f ? sp : r;
    return r;
}

}  // namespace
