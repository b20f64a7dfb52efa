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

}  // namespace
