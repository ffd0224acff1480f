// a15: AdamW step (train.py:41 `torch.optim.AdamW`), one fused elementwise kernel per parameter tensor.
#include "common.h"

namespace {
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);                 // decoupled weight decay
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}
}  // namespace

extern "C" int ddsp_adamw_step(ddsp_ctx* ctx, void* stream, float* param, const float* grad, float* exp_avg,
                               float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step) {
    DDSP_REQUIRE(ctx, ctx && param && grad && exp_avg && exp_avg_sq, "ddsp_adamw_step: null argument");
    DDSP_REQUIRE(ctx, n >= 0 && step >= 1, "ddsp_adamw_step: bad size or step");
    if (n == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_HIP(ctx, hipSetDevice(ctx->device));
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 2048) blocks = 2048;
    ddsp_prof_begin(ctx, st, PF_OPTIM);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
    ddsp_prof_end(ctx, st, 12.0 * n, 28.0 * n);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
