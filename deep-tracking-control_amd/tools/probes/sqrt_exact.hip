// Probe: sqrt_rn() of csrc/wave.hpp against the compiler's correctly rounded sqrtf for ALL 2^32 float32 bit patterns.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I csrc tools/probes/sqrt_exact.hip -o /tmp/sqrt_exact && /tmp/sqrt_exact
// Inputs are visited twice: in natural order (a wave holds 64 neighbouring bit patterns: every wave without a tiny input
// runs the short sequence) and bit-reversed (every wave mixes all magnitudes and therefore takes the sqrtf branch: checks
// that branch and the vote).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "wave.hpp"

__global__ void probe(unsigned long long* bad, unsigned* first_bad, int reversed) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mism = 0;
    for (unsigned long long i = t; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned b = reversed ? __brev((unsigned)i) : (unsigned)i;
        const float x = __int_as_float((int)b);
        const float a = sqrt_rn(x), r = sqrtf(x);
        const bool same = (__float_as_int(a) == __float_as_int(r)) || (a != a && r != r);
        if (!same) {
            ++mism;
            atomicMin(first_bad, b);
        }
    }
    if (mism) atomicAdd(bad, mism);
}

int main() {
    unsigned long long* bad;
    unsigned* first;
    (void)hipMalloc(&bad, 8);
    (void)hipMalloc(&first, 4);
    for (int rev = 0; rev < 2; ++rev) {
        (void)hipMemset(bad, 0, 8);
        (void)hipMemset(first, 0xff, 4);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, bad, first, rev);
        unsigned long long h = 0;
        unsigned f = 0;
        (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
        printf("%s order: %llu mismatches of 4294967296 (lowest mismatching bit pattern 0x%08x)\n", rev ? "bit-reversed" : "natural", h, f);
    }
    return 0;
}
