// a13: random-scale spectral loss, forward value and gradient w.r.t. the predicted signal.
//
// Replaces ddsp/loss.py:7-43 (SSSLoss / RSSLoss; torchaudio Spectrogram(n_fft=N, hop=N, power=1, normalized=True,
// center=False) restated as a Hann(periodic)-windowed non-overlapping one-sided DFT magnitude / sqrt(sum w^2)).
// N is a random integer in [256, 2048) (primes included), so the transform is a dense contraction on the fp32
// matrix pipe against a per-call table T[i][2f | 2f+1] = w[i]*(cos, -sin)(2 pi f i / N) / sqrt(sum w^2):
//   X = frames(x) x T               (rows = B*floor(T/N) frames, 2*(N/2+1) interleaved re/im columns)
//   S = |X| + eps                   (GEMM epilogue; adjacent lanes hold re and im of one bin)
//   per utterance: d2 = sum (S_t-S_p)^2, s2 = sum (S_t+S_p)^2, l1 = sum |ln S_t - ln S_p|
//   L_N = mean_b sqrt(d2_b/s2_b) + alpha * sum_b l1_b / (B*F*Mb);   loss = mean_N L_N
// Gradient: dL/dS_p -> dL/dX (re, im) in one elementwise pass, then dx = dX x T^T with the same table.
// Bound: fp32 matrix pipe (4*N FLOP per sample and scale as a DFT-GEMM), HBM 32 B/sample forward.
#include "gemm_f32.h"
#include "tables.h"

namespace {

constexpr double kTwoPi = 6.283185307179586476925286766559;

__global__ void stft_table_kernel(float* __restrict__ tab, int N, int ld) {
    // norm = sqrt(sum_i hann_p(N)[i]^2) = sqrt(3N/8) (exact for N >= 3)
    const double inv = 1.0 / sqrt(0.375 * (double)N);
    const int Mb = N / 2 + 1;
    const int64_t total = (int64_t)N * ld;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / ld), col = (int)(idx % ld);
        float v = 0.f;
        if (col < 2 * Mb) {
            const int f = col >> 1;
            const int64_t r = ((int64_t)f * i) % N;
            const double ang = kTwoPi * (double)r / (double)N;
            const double w = (0.5 - 0.5 * cos(kTwoPi * (double)i / (double)N)) * inv;
            v = (float)((col & 1) ? -w * sin(ang) : w * cos(ang));
        }
        tab[idx] = v;
    }
}

struct EpiMag {  // S[z*F + m][f] = |X| + eps ; optionally keep X (interleaved re/im) for the backward pass
    static constexpr bool kPair = true;
    float* S;
    float* X;  // may be null
    int F, Mb, ldx;
    float eps;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int z, int m, int n, float v, float other) const {
        const int64_t row = (int64_t)z * F + m;
        if (X) X[row * ldx + n] = v;
        if ((n & 1) == 0) S[row * Mb + (n >> 1)] = sqrtf(fmaf(v, v, other * other)) + eps;
    }
};

// LS_CHUNKS workgroups per utterance: partial d2, s2, l1 in fp64 (combined by loss_combine_kernel)
constexpr int LS_CHUNKS = 16;
__global__ void __launch_bounds__(256) loss_stats_kernel(const float* __restrict__ St, const float* __restrict__ Sp,
                                                         int64_t per_b, double* __restrict__ stats) {
    const int b = blockIdx.x, ch = blockIdx.y;
    const float* t = St + (int64_t)b * per_b;
    const float* p = Sp + (int64_t)b * per_b;
    const int64_t per = (per_b + LS_CHUNKS - 1) / LS_CHUNKS;
    const int64_t i0 = ch * per, i1 = (i0 + per < per_b) ? i0 + per : per_b;
    double d2 = 0, s2 = 0, l1 = 0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float a = t[i], c = p[i];
        const float d = a - c, s = a + c;
        d2 += (double)d * d;
        s2 += (double)s * s;
        l1 += (double)fabsf(logf(a) - logf(c));
    }
    d2 = wave_sum_d(d2);
    s2 = wave_sum_d(s2);
    l1 = wave_sum_d(l1);
    __shared__ double red[12];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[w] = d2;
        red[4 + w] = s2;
        red[8 + w] = l1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = stats + ((int64_t)b * LS_CHUNKS + ch) * 3;
        o[0] = red[0] + red[1] + red[2] + red[3];
        o[1] = red[4] + red[5] + red[6] + red[7];
        o[2] = red[8] + red[9] + red[10] + red[11];
    }
}

// folds the chunk partials of every utterance into stats[b][0..2] (first chunk slot) in a fixed order
__global__ void loss_fold_kernel(double* __restrict__ stats, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double d2 = 0, s2 = 0, l1 = 0;
    for (int c = 0; c < LS_CHUNKS; ++c) {
        const double* o = stats + ((int64_t)b * LS_CHUNKS + c) * 3;
        d2 += o[0];
        s2 += o[1];
        l1 += o[2];
    }
    double* o = stats + (int64_t)b * LS_CHUNKS * 3;
    o[0] = d2;
    o[1] = s2;
    o[2] = l1;
}

// loss += weight * (mean_b sqrt(d2/s2) + alpha * sum l1 / count)
__global__ void loss_combine_kernel(const double* __restrict__ stats, int B, double count, double alpha, double weight,
                                    float* __restrict__ loss, int first) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double conv = 0.0, l1 = 0.0;
    for (int b = 0; b < B; ++b) {
        conv += sqrt(stats[b * LS_CHUNKS * 3 + 0]) / sqrt(stats[b * LS_CHUNKS * 3 + 1]);
        l1 += stats[b * LS_CHUNKS * 3 + 2];
    }
    const double v = weight * (conv / B + alpha * l1 / count);
    loss[0] = (first ? 0.f : loss[0]) + (float)v;
}

// dX[row][2f | 2f+1] = dL/dS_p * (re, im) / |X|
__global__ void __launch_bounds__(256) loss_grad_kernel(const float* __restrict__ St, const float* __restrict__ Sp,
                                                        float* __restrict__ X, const double* __restrict__ stats, int B,
                                                        int F, int Mb, int ldx, double alpha, double weight, float eps) {
    const int64_t per_b = (int64_t)F * Mb;
    const int64_t total = (int64_t)B * per_b;
    const double count = (double)total;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_b);
        const int64_t row = i / Mb;
        const int f = (int)(i % Mb);
        const double rd = sqrt(stats[b * LS_CHUNKS * 3 + 0]), rs = sqrt(stats[b * LS_CHUNKS * 3 + 1]);
        const float a = St[i], c = Sp[i];
        double g = 0.0;
        if (rd > 0.0) g += (-(double)(a - c) / (rd * rs)) / B;
        g += (-rd * (double)(a + c) / (rs * rs * rs)) / B;
        const float dl = logf(c) - logf(a);
        g += alpha * ((dl > 0.f) - (dl < 0.f)) / ((double)c * count);
        g *= weight;
        float* xr = X + row * ldx + 2 * f;
        const float re = xr[0], im = xr[1];
        const float mag = c - eps;  // |X|
        const float k = mag > 0.f ? (float)g / mag : 0.f;
        xr[0] = k * re;
        xr[1] = k * im;
    }
}

struct EpiGradFrames {  // grad[(m / F)*T + (m % F)*N + n] (+)= acc   (m runs over all B*F frames)
    float* grad;
    int64_t T;
    int N, F;
    int accumulate;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int, int m, int n, float v, float) const {
        float* p = grad + (int64_t)(m / F) * T + (int64_t)(m % F) * N + n;
        *p = accumulate ? *p + v : v;
    }
};

__global__ void zero_tail_kernel(float* __restrict__ grad, int64_t T, int64_t used, int B) {
    const int64_t tail = T - used;
    const int64_t total = (int64_t)B * tail;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        grad[(i / tail) * T + used + (i % tail)] = 0.f;
}

}  // namespace

extern "C" int ddsp_rss_loss(ddsp_ctx* ctx, void* stream, const float* x_pred, const float* x_true, int64_t B,
                             int64_t T, const int* n_ffts_host, int n_scale, float alpha, float eps, float* loss,
                             float* grad_pred) {
    DDSP_REQUIRE(ctx, ctx && x_pred && x_true && n_ffts_host && loss, "ddsp_rss_loss: null argument");
    DDSP_REQUIRE(ctx, B >= 1 && B <= 32768 && T >= 4 && n_scale >= 1 && n_scale <= 64, "ddsp_rss_loss: bad shape");
    DDSP_REQUIRE(ctx, T % 4 == 0 && ((uintptr_t)x_pred % 16) == 0 && ((uintptr_t)x_true % 16) == 0,
                 "ddsp_rss_loss: signals must be 16-byte aligned with T % 4 == 0");
    int maxN = 0, minN = 1 << 30;
    for (int s = 0; s < n_scale; ++s) {
        DDSP_REQUIRE(ctx, n_ffts_host[s] >= 8 && n_ffts_host[s] <= T && n_ffts_host[s] <= 8192, "ddsp_rss_loss: n_fft out of range");
        maxN = n_ffts_host[s] > maxN ? n_ffts_host[s] : maxN;
        minN = n_ffts_host[s] < minN ? n_ffts_host[s] : minN;
    }
    hipStream_t st = (hipStream_t)stream;
    DDSP_HIP(ctx, hipSetDevice(ctx->device));
    // scratch sized for the worst scale: table N x ld, S_t, S_p (rows x Mb), X_p (rows x ld) when a gradient is wanted
    const size_t tab_f = (size_t)maxN * ddsp_pad4(2 * (maxN / 2 + 1));
    const size_t rows_max = (size_t)B * (size_t)(T / minN);
    const size_t s_f = (size_t)B * (size_t)(T / 2 + T / minN + 8);  // rows*Mb <= B*(T/2 + F)
    const size_t x_f = grad_pred ? (size_t)B * (size_t)(T + 4 * (T / minN) + 16) : 0;
    (void)rows_max;
    int rc = ddsp_scratch_reserve_bytes(ctx, (tab_f + 2 * s_f + x_f) * sizeof(float) + (size_t)B * 16 * 3 * sizeof(double) + 8192);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float *tab, *St, *Sp, *Xp = nullptr;
    double* stats;
    if ((rc = ddsp_scratch_get(ctx, tab_f * sizeof(float), (void**)&tab))) return rc;
    if ((rc = ddsp_scratch_get(ctx, s_f * sizeof(float), (void**)&St))) return rc;
    if ((rc = ddsp_scratch_get(ctx, s_f * sizeof(float), (void**)&Sp))) return rc;
    if (grad_pred && (rc = ddsp_scratch_get(ctx, x_f * sizeof(float), (void**)&Xp))) return rc;
    if ((rc = ddsp_scratch_get(ctx, (size_t)B * LS_CHUNKS * 3 * sizeof(double), (void**)&stats))) return rc;

    ddsp_prof_begin(ctx, st, PF_RSS_LOSS);
    double flops = 0.0;
    // samples no scale covers keep a zero gradient; frames of later scales accumulate into earlier ones
    int64_t covered = 0;
    for (int s = 0; s < n_scale; ++s) {
        const int N = n_ffts_host[s], Mb = N / 2 + 1, ld = ddsp_pad4(2 * Mb);
        const int F = (int)(T / N);
        const double weight = 1.0 / n_scale;
        hipLaunchKernelGGL(stft_table_kernel, dim3(1024), dim3(256), 0, st, tab, N, ld);
        // all B*F frames form one M dimension (frame-remapped loader): full tiles even when F is 43
        gemm::Args g = gemm::make(x_true, N, tab, ld, (int)(B * F), 2 * Mb, N);
        g.Fr = F;
        g.sA_hi = T;
        EpiMag et{St, nullptr, F, Mb, ld, eps};
        gemm::launch<true, false, gemm::A_FRAMES>(st, g, 1, et);
        g.A = x_pred;
        EpiMag ep{Sp, Xp, F, Mb, ld, eps};
        gemm::launch<true, false, gemm::A_FRAMES>(st, g, 1, ep);
        hipLaunchKernelGGL(loss_stats_kernel, dim3((unsigned)B, LS_CHUNKS), dim3(256), 0, st, St, Sp, (int64_t)F * Mb, stats);
        hipLaunchKernelGGL(loss_fold_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, stats, (int)B);
        hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, st, stats, (int)B, (double)B * F * Mb,
                           (double)alpha, weight, loss, s == 0 ? 1 : 0);
        flops += 2.0 * 2.0 * B * F * (double)N * 2 * Mb;
        if (grad_pred) {
            const int64_t total = (int64_t)B * F * Mb;
            int64_t blocks = ceil_div64(total, 256);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, St, Sp, Xp, stats, (int)B, F,
                               Mb, ld, (double)alpha, weight, eps);
            const int64_t used = (int64_t)F * N;
            // dx = dX x T^T : A = dX (rows x 2Mb), B(k, n) = T[n][k]
            gemm::Args gb = gemm::make(Xp, ld, tab, ld, (int)(B * F), N, 2 * Mb);
            if (s == 0) {
                EpiGradFrames eg{grad_pred, T, N, F, 0};
                gemm::launch<true, true, gemm::A_PLAIN>(st, gb, 1, eg);
                if (used < T) hipLaunchKernelGGL(zero_tail_kernel, dim3(64), dim3(256), 0, st, grad_pred, T, used, (int)B);
                covered = used;
            } else {
                // first fill any not-yet-covered stretch this scale reaches with zeros, then accumulate
                EpiGradFrames eg{grad_pred, T, N, F, 1};
                gemm::launch<true, true, gemm::A_PLAIN>(st, gb, 1, eg);
                if (used > covered) covered = used;
            }
            flops += 2.0 * B * F * (double)N * 2 * Mb;
        }
    }
    (void)covered;
    ddsp_prof_end(ctx, st, flops, 4.0 * B * T * (2.0 * n_scale + (grad_pred ? 1.0 : 0.0)));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
