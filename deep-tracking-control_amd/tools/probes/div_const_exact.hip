// Probe: q = fma(fma(-x*y, c, x), y, x*y), y = RN(1/c), against the IEEE quotient x / c for ALL 2^32 float32 x and the four
// constants the planner divides by (grid spacings 0.05 / 0.1, point counts 693 / 692).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/probes/div_const_exact.hip -o /tmp/div_const_exact && /tmp/div_const_exact
// Prints, per constant, the number of mismatching x inside and outside 1e-30 <= |x| <= 1e30 (NaN == NaN).
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float div_const(float x, float c, float inv) {
    const float q0 = x * inv;
    const float r = __builtin_fmaf(-q0, c, x);
    return __builtin_fmaf(r, inv, q0);
}

__global__ void probe(float c, float inv, unsigned long long* bad_in, unsigned long long* bad_out) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bi = 0, bo = 0;
    for (unsigned long long i = t; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __int_as_float((int)(unsigned)i);
        const float a = div_const(x, c, inv), r = x / c;
        const bool same = (__float_as_int(a) == __float_as_int(r)) || (a != a && r != r);
        if (!same) {
            const float ax = fabsf(x);
            if (ax >= 1e-30f && ax <= 1e30f) ++bi;
            else ++bo;
        }
    }
    if (bi) atomicAdd(bad_in, bi);
    if (bo) atomicAdd(bad_out, bo);
}

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 16);
    const float cs[4] = {0.05f, 0.1f, 693.0f, 692.0f};
    for (float c : cs) {
        const float inv = 1.0f / c;
        (void)hipMemset(d, 0, 16);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, c, inv, d, d + 1);
        unsigned long long h[2] = {0, 0};
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("c = %-6g y = RN(1/c) = %.9g : %llu mismatches with 1e-30 <= |x| <= 1e30, %llu outside\n", c, inv, h[0], h[1]);
    }
    return 0;
}
