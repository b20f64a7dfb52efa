// Device random draws of PPO.update for gfx950: the mini-batch permutation (rollout_storage.py:165 `torch.randperm`) and
// the reparameterisation noise of the CE-net (actor_critic_decoder.py:283 `torch.randn_like`).  The reference takes both
// from torch's generator; parity tests inject the reference's draws (PPO.update(perm=, eps1=, eps2=)), production draws
// them here: counter-based, one launch each, no sort, no host synchronisation.
//
//   dtc_randn    : Philox4x32-10 (Salmon et al. 2011; the generator family torch's device RNG uses), counter = element
//                  group, key = seed; four 32-bit words -> two Box-Muller pairs -> four N(0,1) floats per thread.
//   dtc_randperm : a keyed bijection of [0, 2^k), k = ceil(log2 n), applied to i and re-applied while the image is >= n
//                  (cycle walking: the restriction of a bijection to [0, n) closed under re-application is a permutation
//                  of [0, n); < 2 applications on average).  The bijection is an alternating unbalanced Feistel network
//                  (8 rounds; round function = a 32-bit finaliser hash of the other half and the Philox-expanded round
//                  key): every element's position is O(1) work -- no sort, no atomics, deterministic in (seed, n).
#include <math.h>

#include "common.hpp"

namespace {

struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
        c = U4{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& n0, float& n1) {
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-08f;       // (0, 1]: 24 uniform bits
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-08f;              // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    n0 = r * c;
    n1 = r * s;
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long long n, unsigned long long seed,
                                                    unsigned long long offset) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;         // group of four outputs
    if (4 * g >= n) return;
    const unsigned long long ctr = (unsigned long long)g + offset;
    const U4 x = philox4x32_10(U4{(unsigned)ctr, (unsigned)(ctr >> 32), 0x6474635fu, 0x726e646eu}, (unsigned)seed, (unsigned)(seed >> 32));
    float v[4];
    box_muller(x.x, x.y, v[0], v[1]);
    box_muller(x.z, x.w, v[2], v[3]);
    if (4 * g + 4 <= n && ((reinterpret_cast<unsigned long long>(out) & 15ull) == 0)) {
        *reinterpret_cast<float4*>(out + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int e = 0; e < 4 && 4 * g + e < n; ++e) out[4 * g + e] = v[e];
    }
}

__device__ __forceinline__ unsigned mix32(unsigned h) {          // murmur3 finaliser: a bijection of 32-bit words
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

struct PermKeys { unsigned k[8]; };

__global__ __launch_bounds__(256) void randperm_kernel(long long* __restrict__ out, long long n, int lbits, int rbits, PermKeys K) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned lmask = lbits ? ((1u << lbits) - 1u) : 0u, rmask = (1u << rbits) - 1u;
    unsigned long long x = (unsigned long long)i;
    do {
        unsigned L = (unsigned)(x >> rbits) & lmask, R = (unsigned)x & rmask;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            L ^= mix32(R ^ K.k[r]) & lmask;
            R ^= mix32(L ^ K.k[r + 1]) & rmask;
        }
        x = ((unsigned long long)L << rbits) | R;
    } while ((long long)x >= n);
    out[i] = (long long)x;
}

// host twin of philox4x32_10 (round keys of the permutation)
void philox_host(unsigned c[4], unsigned k0, unsigned k1) {
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

}  // namespace

extern "C" int dtc_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    DTC_REQUIRE(n >= 0 && (out || n == 0), "bad arguments");
    if (n == 0) return DTC_OK;
    hipStream_t s = (hipStream_t)stream;
    const long long groups = dtc::ceil_div(n, 4);
    dtc::ProfScope prof("randn", (double)n * 4.0, s);
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)dtc::ceil_div(groups, 256)), dim3(256), 0, s, out, (long long)n,
                       (unsigned long long)seed, (unsigned long long)offset);
    return dtc::check_launch("randn");
}

extern "C" int dtc_randperm(int64_t* out, int64_t n, uint64_t seed, void* stream) {
    DTC_REQUIRE(n >= 0 && n <= (1ll << 40) && (out || n == 0), "bad arguments");
    if (n == 0) return DTC_OK;
    hipStream_t s = (hipStream_t)stream;
    int k = 1;
    while ((1ll << k) < n) ++k;                      // domain 2^k >= n, k >= 1
    const int rbits = (k + 1) / 2, lbits = k - rbits;
    DTC_REQUIRE(rbits <= 31, "permutation too long");
    PermKeys K;
    unsigned c[4] = {0x6474635fu, 0x7065726du, (unsigned)n, (unsigned)(n >> 32)};
    philox_host(c, (unsigned)seed, (unsigned)(seed >> 32));
    for (int i = 0; i < 4; ++i) K.k[i] = c[i];
    c[0] ^= 0x9E3779B9u;
    philox_host(c, (unsigned)(seed >> 32), (unsigned)seed);
    for (int i = 0; i < 4; ++i) K.k[4 + i] = c[i];
    dtc::ProfScope prof("randperm", (double)n * 8.0, s);
    hipLaunchKernelGGL(randperm_kernel, dim3((unsigned)dtc::ceil_div(n, 256)), dim3(256), 0, s, (long long*)out, (long long)n,
                       lbits, rbits, K);
    return dtc::check_launch("randperm");
}
