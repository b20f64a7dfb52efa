// clip_grad_norm_ + Adam over one flat fp32 parameter range, for gfx950.
//
// Reference: rsl_rl/rsl_rl/algorithms/ppo.py:253-254 and :334-335
//     nn.utils.clip_grad_norm_(params, max_grad_norm); optimizer.step()      (torch.optim.Adam)
// torch launches ~6 kernels per parameter tensor (43 / 26 tensors per optimiser); the parameters
// of each optimiser live in ONE contiguous range of a flat buffer here, so a step is two launches:
//   1. sum of squares of the gradient range (fp64 partials);
//   2. every block re-reduces the partials -> global norm -> clip coefficient, then applies
//      Adam (torch's single-tensor formulas, same operation order) to its slice.
// The learning rate is read from device memory (it is adapted on device by dtc_ppo_loss), so the
// whole mini-batch step needs no host synchronisation.  HBM-bound: 28 B per parameter.
#include <math.h>

#include "common.hpp"

namespace {

constexpr int MAX_PART = 1024;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ part) {
    __shared__ double sh[4];
    double a = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double x = (double)g[i];
        a += x * x;
    }
    a = block_sum_d(a, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, const double* __restrict__ part,
                                                        int npart, float max_norm, const double* __restrict__ lr,
                                                        double inv_bc1, float bc2_sqrt, float w1, float beta2, float w2,
                                                        float eps, float* __restrict__ gnorm_out) {
    __shared__ double sh[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < npart; i += blockDim.x) a += part[i];
    a = block_sum_d(a, sh);
    const float total_norm = (float)sqrt(a);
    float coef = max_norm / (total_norm + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
    if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm_out) *gnorm_out = total_norm;
    const float neg_step = -(float)((*lr) * inv_bc1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        g[i] = gi;                                               // clip_grad_norm_ scales .grad in place
        float mi = m[i], vi = v[i];
        mi = mi + w1 * (gi - mi);                                // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * beta2;                                         // exp_avg_sq.mul_(beta2)
        vi = vi + (w2 * gi) * gi;                                //   .addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;          // (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps)
        p[i] = p[i] + (neg_step * mi) / denom;                   // param.addcdiv_(exp_avg, denom, value=-step_size)
        m[i] = mi;
        v[i] = vi;
    }
}

}  // namespace

extern "C" int64_t dtc_adam_workspace(int64_t n) {
    (void)n;
    return (int64_t)sizeof(double) * MAX_PART;
}

extern "C" int dtc_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float max_grad_norm, const double* lr, double beta1, double beta2, double eps, int64_t step,
                             float* gnorm_out, void* workspace, void* stream) {
    DTC_REQUIRE(n > 0 && step >= 1, "bad n/step");
    DTC_REQUIRE(params && grads && exp_avg && exp_avg_sq && lr && workspace, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    int g1 = (int)dtc::ceil_div(n, 256 * 8);
    if (g1 > MAX_PART) g1 = MAX_PART;
    if (g1 < 1) g1 = 1;
    // bias corrections in double on the host, as torch does (python floats)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
    dtc::ProfScope prof("clip_adam", (double)n * 32.0, s);
    hipLaunchKernelGGL(sumsq_kernel, dim3(g1), dim3(256), 0, s, grads, (long long)n, part);
    int g2 = (int)dtc::ceil_div(n, 256 * 4);
    if (g2 > 2048) g2 = 2048;
    hipLaunchKernelGGL(clip_adam_kernel, dim3(g2), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (long long)n, part,
                       g1, max_grad_norm, lr, 1.0 / bc1, (float)sqrt(bc2), w1, (float)beta2, w2, (float)eps, gnorm_out);
    return dtc::check_launch("clip_adam");
}
