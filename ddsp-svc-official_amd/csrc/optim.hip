// a15: AdamW step (train.py:41 `torch.optim.AdamW`): one fused elementwise kernel per parameter tensor
// (ddsp_adamw_step) or per batch of up to 24 tensors (ddsp_adamw_step_multi, what the optimizer class uses).
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

// Up to ADAMW_NB parameter tensors per launch: the pointer table travels in the kernel arguments (no device-side table
// to keep in sync with torch's gradient allocations), a block finds its tensor by scanning <= 24 scalar offsets.
// A training step has ~100 parameter tensors, most of them a few hundred floats: one launch each made the optimizer
// host-bound (100 launches, ~1.5 ms of Python + launch time for 0.46 ms of kernels).
constexpr int ADAMW_NB = 24;
constexpr int ADAMW_CHUNK = 256 * 8;    // elements per block
struct AdamwBatch {
    float* p[ADAMW_NB];
    const float* g[ADAMW_NB];
    float* m[ADAMW_NB];
    float* v[ADAMW_NB];
    int64_t n[ADAMW_NB];
    int blk0[ADAMW_NB + 1];             // first block of tensor t; blk0[count] = grid size
    int count;
};

__global__ void __launch_bounds__(256) adamw_multi_kernel(AdamwBatch a, float lr, float beta1, float beta2, float eps,
                                                          float wd, float bc1, float bc2_sqrt) {
    const int b = blockIdx.x;
    int t = 0;
    while (t + 1 < a.count && b >= a.blk0[t + 1]) ++t;      // wave-uniform
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const int64_t n = a.n[t];
    const int64_t lo = (int64_t)(b - a.blk0[t]) * ADAMW_CHUNK;
    const int64_t hi = lo + ADAMW_CHUNK < n ? lo + ADAMW_CHUNK : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);                 // same arithmetic, in the same order, as adamw_kernel
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

extern "C" int ddsp_adamw_step_multi(ddsp_ctx* ctx, void* stream, int n_tensors, float* const* params,
                                     const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                     const int64_t* numel, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, int64_t step) {
    DDSP_REQUIRE(ctx, ctx && n_tensors >= 0 && step >= 1, "ddsp_adamw_step_multi: bad count or step");
    if (n_tensors == 0) return DDSP_OK;
    DDSP_REQUIRE(ctx, params && grads && exp_avg && exp_avg_sq && numel, "ddsp_adamw_step_multi: null table");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    double total = 0;
    for (int i = 0; i < n_tensors; ++i) {
        DDSP_REQUIRE(ctx, numel[i] >= 0, "ddsp_adamw_step_multi: negative size");
        DDSP_REQUIRE(ctx, numel[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i]),
                     "ddsp_adamw_step_multi: null tensor");
        total += (double)numel[i];
    }
    ddsp_prof_begin(ctx, st, PF_OPTIM);
    int i = 0;
    while (i < n_tensors) {
        AdamwBatch a;
        a.count = 0;
        int blocks = 0;
        // fill one launch: up to ADAMW_NB non-empty tensors and a grid that stays far below 2^31 blocks
        while (i < n_tensors && a.count < ADAMW_NB) {
            const int64_t n = numel[i];
            if (n > 0) {
                const int64_t nb = ceil_div64(n, ADAMW_CHUNK);
                if (a.count > 0 && blocks + nb > (1 << 24)) break;
                DDSP_REQUIRE(ctx, nb <= (1 << 30), "ddsp_adamw_step_multi: tensor too large for one launch");
                a.p[a.count] = params[i];
                a.g[a.count] = grads[i];
                a.m[a.count] = exp_avg[i];
                a.v[a.count] = exp_avg_sq[i];
                a.n[a.count] = n;
                a.blk0[a.count] = blocks;
                blocks += (int)nb;
                ++a.count;
            }
            ++i;
        }
        if (a.count == 0) break;
        a.blk0[a.count] = blocks;
        hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, lr, beta1, beta2, eps,
                           weight_decay, (float)bc1, (float)sqrt(bc2));
    }
    ddsp_prof_end(ctx, st, 12.0 * total, 28.0 * total);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_adamw_step(ddsp_ctx* ctx, void* stream, float* param, const float* grad, float* exp_avg,
                               float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step) {
    DDSP_REQUIRE(ctx, ctx && param && grad && exp_avg && exp_avg_sq, "ddsp_adamw_step: null argument");
    DDSP_REQUIRE(ctx, n >= 0 && step >= 1, "ddsp_adamw_step: bad size or step");
    if (n == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
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
