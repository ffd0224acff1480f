// a13: random-scale spectral loss, forward value and gradient w.r.t. the predicted signal.
//
// Replaces ddsp/loss.py:7-43 (SSSLoss / RSSLoss; torchaudio Spectrogram(n_fft=N, hop=int(N*(1-overlap)), power=1,
// normalized=True, center=False) restated as a Hann(periodic)-windowed one-sided DFT magnitude / sqrt(sum w^2)).
// N is a random integer in [256, 2048) (primes included), so the transform is a dense contraction on the fp32
// matrix pipe against a per-call table T[i][2f | 2f+1] = w[i]*(cos, -sin)(2 pi f i / N) / sqrt(sum w^2):
//   X = frames(x) x T               (rows = B*F frames, F = 1 + (T-N)/hop, 2*(N/2+1) interleaved re/im columns)
// Data path (round 2): the frames of both signals are copied into ONE zero-padded matrix (row pitch Kp = N rounded up to 32
// floats: rows 16-byte aligned, k-steps whole) and the table is generated transposed with the same pitch, so that the
// contraction runs on the persistent LDS-DMA GEMM (gemm_f32.h kernel_dma: both operands streamed k-contiguous into the
// LDS) instead of the register-staged kernel - one launch for both signals.  The copy costs 2 x 4 B per sample and scale.
//   S = |X| + eps                   (GEMM epilogue; adjacent lanes hold re and im of one bin)
//   per utterance: d2 = sum (S_t-S_p)^2, s2 = sum (S_t+S_p)^2, l1 = sum |ln S_t - ln S_p|
//   L_N = mean_b sqrt(d2_b/s2_b) + alpha * sum_b l1_b / (B*F*Mb);   loss = mean_N L_N
// Gradient: dL/dS_p -> dL/dX (re, im) in one elementwise pass, then dx = dX x T^T with the same table.
// Bound: fp32 matrix pipe (4*N FLOP per sample and scale as a DFT-GEMM), HBM 32 B/sample forward.
#include "gemm_f32.h"

#include <algorithm>
#include "tables.h"

namespace {

constexpr double kTwoPi = 6.283185307179586476925286766559;

// tabT[col][i] (pitch Kp, zero for i >= N): the forward operand, k = i contiguous; tab[i][col] (pitch Kb, zero for
// col >= 2 Mb; only when the gradient is wanted): the backward operand, k = col contiguous
// The same tables from per-scale 1-D factors (round 3): t1 = [cos(2 pi j / N) | sin(2 pi j / N) | window_j / norm] in fp64, the very
// expressions of stft_table_kernel, so the entries are bit-identical; an entry then costs an integer remainder and two loads instead
// of three fp64 transcendentals (17 -> ~6 us per scale).
constexpr int T1_SCALES = 8;
struct Table1dArgs {
    int n;
    int N[T1_SCALES];
    double* t1[T1_SCALES];   // 3 N doubles each
};
__global__ void __launch_bounds__(256) stft_table1d_kernel(Table1dArgs a) {
    const int s = blockIdx.y;
    if (s >= a.n) return;
    const int N = a.N[s];
    const double inv = 1.0 / sqrt(0.375 * (double)N);
    double* __restrict__ t = a.t1[s];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < N; j += gridDim.x * 256) {
        const double ang = kTwoPi * (double)j / (double)N;
        t[j] = cos(ang);
        t[N + j] = sin(ang);
        t[2 * N + j] = (0.5 - 0.5 * cos(kTwoPi * (double)j / (double)N)) * inv;
    }
}
__global__ void stft_table_from1d_kernel(const double* __restrict__ t1, float* __restrict__ tabT, float* __restrict__ tab, int N,
                                         int Kp, int Kb) {
    const int Mb = N / 2 + 1;
    const int64_t total = (int64_t)2 * Mb * Kp;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx / Kp), i = (int)(idx % Kp);
        float v = 0.f;
        if (i < N) {
            const int f = col >> 1;
            const int r = (int)(((unsigned)f * (unsigned)i) % (unsigned)N);   // f <= 4096, i < 8192
            const double w = t1[2 * N + i];
            v = (float)((col & 1) ? -w * t1[N + r] : w * t1[r]);
            if (tab) tab[(int64_t)i * Kb + col] = v;
        }
        tabT[idx] = v;
    }
    if (tab) {
        const int pad = Kb - 2 * Mb;
        const int64_t tz = (int64_t)N * pad;
        for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tz; idx += (int64_t)gridDim.x * blockDim.x)
            tab[(idx / pad) * Kb + 2 * Mb + (idx % pad)] = 0.f;
    }
}
__global__ void stft_table_kernel(float* __restrict__ tabT, float* __restrict__ tab, int N, int Kp, int Kb) {
    // norm = sqrt(sum_i hann_p(N)[i]^2) = sqrt(3N/8) (exact for N >= 3)
    const double inv = 1.0 / sqrt(0.375 * (double)N);
    const int Mb = N / 2 + 1;
    const int64_t total = (int64_t)2 * Mb * Kp;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx / Kp), i = (int)(idx % Kp);
        float v = 0.f;
        if (i < N) {
            const int f = col >> 1;
            const int64_t r = ((int64_t)f * i) % N;
            const double ang = kTwoPi * (double)r / (double)N;
            const double w = (0.5 - 0.5 * cos(kTwoPi * (double)i / (double)N)) * inv;
            v = (float)((col & 1) ? -w * sin(ang) : w * cos(ang));
            if (tab) tab[(int64_t)i * Kb + col] = v;
        }
        tabT[idx] = v;
    }
    if (tab) {
        const int pad = Kb - 2 * Mb;
        const int64_t tz = (int64_t)N * pad;
        for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tz; idx += (int64_t)gridDim.x * blockDim.x)
            tab[(idx / pad) * Kb + 2 * Mb + (idx % pad)] = 0.f;
    }
}

// Xf[sig][b*F + f][k] = k < N ? x_sig[b][f*hop + k] : 0   (sig 0 = target, 1 = prediction; row pitch Kp)
__global__ void __launch_bounds__(256) frame_pad_kernel(const float* __restrict__ x_true, const float* __restrict__ x_pred,
                                                        int64_t T, int N, int hop, int F, int64_t M, int Kp,
                                                        float* __restrict__ Xf) {
    const int q4 = Kp / 4;
    const int64_t total = 2 * M * q4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / q4;
        const int k = (int)(idx % q4) * 4;
        const int64_t m = r >= M ? r - M : r;
        const float* src = (r >= M ? x_pred : x_true) + (m / F) * T + (m % F) * (int64_t)hop + k;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k + 3 < N) {
            v = *(const gemm::f32x4_u*)src;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < N) v[j] = src[j];
        }
        *(f32x4*)(Xf + r * Kp + k) = v;
    }
}

struct EpiMag {  // z = 0: target, z = 1: prediction.  S_z[m][f] = |X| + eps ; X (interleaved re/im) of the prediction kept for the backward pass
    static constexpr bool kPair = true;
    float* St;
    float* Sp;
    float* X;  // may be null
    int Mb, ldx;
    float eps;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int z, int m, int n, float v, float other) const {
        if (z && X) X[(int64_t)m * ldx + n] = v;
        if ((n & 1) == 0) (z ? Sp : St)[(int64_t)m * Mb + (n >> 1)] = sqrtf(fmaf(v, v, other * other)) + eps;
    }
    // whole column tiles: a lane receives four consecutive columns of a row = (re, im) of two bins - one 16-byte store
    // of X, the two magnitudes formed in the lane (the same expressions as above)
    static constexpr bool kStore4 = true;
    __device__ __forceinline__ bool vec_ok() const { return X == nullptr || (((uintptr_t)X % 16) == 0 && ldx % 4 == 0); }
    __device__ __forceinline__ void store4(int z, int m, int n, f32x4 v) const {
        if (z && X) *(f32x4*)(X + (int64_t)m * ldx + n) = v;
        float* s = (z ? Sp : St) + (int64_t)m * Mb + (n >> 1);
        s[0] = sqrtf(fmaf(v[0], v[0], v[1] * v[1])) + eps;
        s[1] = sqrtf(fmaf(v[2], v[2], v[3] * v[3])) + eps;
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
        if (f == Mb - 1)   // the row's pad columns (they meet zero rows of the table in the adjoint product; < 32 of them)
            for (int p = 2 * Mb; p < ldx; ++p) X[row * ldx + p] = 0.f;
    }
}

// grad[b][t] (+)= sum over the frames that cover t of dXf[b*F + f][t - f*hop]   (gather form of the overlap-add: one
// term with hop == N; samples no frame covers get 0 on the first scale and stay as they are afterwards)
__global__ void __launch_bounds__(256) frames_ola_kernel(const float* __restrict__ dXf, int Kp, int64_t T, int N, int hop,
                                                         int F, int64_t B, int accumulate, float* __restrict__ grad) {
    const int64_t total = B * T;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = idx / T, t = idx % T;
        int64_t f1 = t / hop;
        if (f1 > F - 1) f1 = F - 1;
        int64_t f0 = t - N + 1 > 0 ? (t - N + hop) / hop : 0;   // ceil((t - N + 1) / hop)
        float s = 0.f;
        for (int64_t f = f0; f <= f1; ++f) s += dXf[(b * F + f) * Kp + (t - f * hop)];
        grad[idx] = accumulate ? grad[idx] + s : s;
    }
}

}  // namespace

extern "C" int ddsp_rss_loss(ddsp_ctx* ctx, void* stream, const float* x_pred, const float* x_true, int64_t B,
                             int64_t T, const int* n_ffts_host, const int* hops_host, int n_scale, float alpha, float eps,
                             float* loss, float* grad_pred) {
    DDSP_REQUIRE(ctx, ctx && x_pred && x_true && n_ffts_host && loss, "ddsp_rss_loss: null argument");
    DDSP_REQUIRE(ctx, B >= 1 && B <= 32768 && T >= 4 && n_scale >= 1 && n_scale <= 64, "ddsp_rss_loss: bad shape");
    DDSP_REQUIRE(ctx, T % 4 == 0 && ((uintptr_t)x_pred % 16) == 0 && ((uintptr_t)x_true % 16) == 0,
                 "ddsp_rss_loss: signals must be 16-byte aligned with T % 4 == 0");
    auto hop_of = [&](int s) { return hops_host ? hops_host[s] : n_ffts_host[s]; };
    auto round32 = [](int v) { return (v + 31) & ~31; };
    size_t tabT_f = 0, tab_f = 0, xf_f = 0, s_f = 0, xp_f = 0;
    for (int s = 0; s < n_scale; ++s) {
        const int N = n_ffts_host[s], hop = hop_of(s);
        DDSP_REQUIRE(ctx, N >= 8 && N <= T && N <= 8192, "ddsp_rss_loss: n_fft out of range");
        DDSP_REQUIRE(ctx, hop >= 1 && hop <= N, "ddsp_rss_loss: hop must lie in [1, n_fft]");
        const size_t Mb = N / 2 + 1, Kp = round32(N), Kb = round32(2 * (int)Mb), M = (size_t)B * (size_t)((T - N) / hop + 1);
        DDSP_REQUIRE(ctx, M * Kp < ((size_t)1 << 31) && 2 * M < ((size_t)1 << 31), "ddsp_rss_loss: too many frames for one call");
        tabT_f = std::max(tabT_f, 2 * Mb * Kp);
        tab_f = std::max(tab_f, (size_t)N * Kb);
        xf_f = std::max(xf_f, 2 * M * Kp);
        s_f = std::max(s_f, M * Mb);
        xp_f = std::max(xp_f, M * Kb);
    }
    if (!grad_pred) tab_f = xp_f = 0;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    // scratch sized for the worst scale: the two tables, the framed copy of both signals, S_t, S_p, X_p
    static const bool table_1d = [] { const char* e = getenv("DDSP_LOSS_TABLE_1D"); return !(e && e[0] == '0'); }();
    const bool use_1d = table_1d && n_scale <= T1_SCALES;
    size_t t1_doubles = 0;
    if (use_1d)
        for (int s = 0; s < n_scale; ++s) t1_doubles += 3 * (size_t)n_ffts_host[s] + 2;
    int rc = ddsp_scratch_reserve_bytes(ctx, (tabT_f + tab_f + xf_f + 2 * s_f + xp_f) * sizeof(float) +
                                                 (size_t)B * LS_CHUNKS * 3 * sizeof(double) + t1_doubles * sizeof(double) + 16384);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float *tabT, *tab = nullptr, *Xf, *St, *Sp, *Xp = nullptr;
    double* stats;
    if ((rc = ddsp_scratch_get(ctx, tabT_f * sizeof(float), (void**)&tabT))) return rc;
    if (grad_pred && (rc = ddsp_scratch_get(ctx, tab_f * sizeof(float), (void**)&tab))) return rc;
    if ((rc = ddsp_scratch_get(ctx, xf_f * sizeof(float), (void**)&Xf))) return rc;
    if ((rc = ddsp_scratch_get(ctx, s_f * sizeof(float), (void**)&St))) return rc;
    if ((rc = ddsp_scratch_get(ctx, s_f * sizeof(float), (void**)&Sp))) return rc;
    if (grad_pred && (rc = ddsp_scratch_get(ctx, xp_f * sizeof(float), (void**)&Xp))) return rc;
    if ((rc = ddsp_scratch_get(ctx, (size_t)B * LS_CHUNKS * 3 * sizeof(double), (void**)&stats))) return rc;
    Table1dArgs t1a{};
    if (use_1d) {
        double* t1 = nullptr;
        if ((rc = ddsp_scratch_get(ctx, t1_doubles * sizeof(double), (void**)&t1))) return rc;
        t1a.n = n_scale;
        for (int s = 0; s < n_scale; ++s) {
            t1a.N[s] = n_ffts_host[s];
            t1a.t1[s] = t1;
            t1 += 3 * (size_t)n_ffts_host[s] + 2;
        }
    }
    // product arithmetic of the two DFT contractions: fp32 matrix products.  (The error of the spectra enters the gradient
    // through 1/S in near-empty bins, so the split-bf16 x3 class is not an option here; DDSP_LOSS_MATH=6 selects the
    // six-product form for measurements.)
    static int loss_math = -1;
    if (loss_math < 0) {
        const char* e = getenv("DDSP_LOSS_MATH");
        loss_math = e ? atoi(e) : 0;
    }

    // The adjoint contraction dXf = dX T^T is a LINEAR map of the spectral gradient: a 4e-6 product error there is a 4e-6 relative
    // error of the gradient, nothing is amplified (unlike the forward spectra, whose error meets 1 / S) - it runs in the context's
    // arithmetic (split-bf16 by default; DDSP_LOSS_BWD_MATH=0 forces fp32 products, ddsp_ctx_set_math(FP32) does too).
    static int loss_bwd_env = -2;
    if (loss_bwd_env == -2) {
        const char* e = getenv("DDSP_LOSS_BWD_MATH");
        loss_bwd_env = e ? atoi(e) : -1;
    }
    const int loss_bwd_math = loss_bwd_env >= 0 ? loss_bwd_env : (ctx->math == DDSP_MATH_FP32 ? 0 : DDSP_MATH_SPLIT_BF16);
    ddsp_prof_begin(ctx, st, PF_RSS_LOSS);
    double flops = 0.0;
    for (int s = 0; s < n_scale; ++s) {
        const int N = n_ffts_host[s], hop = hop_of(s), Mb = N / 2 + 1, Kp = round32(N), Kb = round32(2 * Mb);
        const int F = (int)((T - N) / hop + 1);
        const int64_t M = B * F;
        const double weight = 1.0 / n_scale;
        if (use_1d) {
            if (s == 0) hipLaunchKernelGGL(stft_table1d_kernel, dim3(8, (unsigned)n_scale), dim3(256), 0, st, t1a);
            hipLaunchKernelGGL(stft_table_from1d_kernel, dim3(1024), dim3(256), 0, st, t1a.t1[s], tabT, tab, N, Kp, Kb);
        } else
            hipLaunchKernelGGL(stft_table_kernel, dim3(1024), dim3(256), 0, st, tabT, tab, N, Kp, Kb);
        hipLaunchKernelGGL(frame_pad_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(2 * M * (Kp / 4), 256), 16384)), dim3(256),
                           0, st, x_true, x_pred, T, N, hop, F, M, Kp, Xf);
        // (the pad columns of Xp multiply zeros of the table: loss_grad_kernel clears them, the spectra kernel writes the rest)
        {
            gemm::Args g = gemm::make(Xf, Kp, tabT, Kp, (int)M, 2 * Mb, Kp);
            g.sA_hi = M * Kp;   // z = 0 target, 1 prediction
            g.math = loss_math;
            EpiMag e{St, Sp, Xp, Mb, Kb, eps};
            gemm::launch<true, true, gemm::A_PLAIN>(st, g, 2, e);
        }
        hipLaunchKernelGGL(loss_stats_kernel, dim3((unsigned)B, LS_CHUNKS), dim3(256), 0, st, St, Sp, (int64_t)F * Mb, stats);
        hipLaunchKernelGGL(loss_fold_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, stats, (int)B);
        hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, st, stats, (int)B, (double)B * F * Mb,
                           (double)alpha, weight, loss, s == 0 ? 1 : 0);
        flops += 2.0 * 2.0 * M * (double)N * 2 * Mb;
        if (grad_pred) {
            const int64_t total = M * Mb;
            int64_t blocks = ceil_div64(total, 256);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, St, Sp, Xp, stats, (int)B, F,
                               Mb, Kb, (double)alpha, weight, eps);
            // per-frame gradient dXf = dX x T^T (A = dX (M x Kb), B(k, n) = tab[n][k]) into the framed buffer, then the
            // overlap-add back onto the signal axis
            gemm::Args gb = gemm::make(Xp, Kb, tab, Kb, (int)M, N, Kb);
            gb.math = loss_bwd_math;
            gemm::EpiStore eg{Xf, Kp, nullptr, 1, 0, 0};
            gemm::launch<true, true, gemm::A_PLAIN>(st, gb, 1, eg);
            hipLaunchKernelGGL(frames_ola_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(B * T, 256), 16384)), dim3(256), 0,
                               st, Xf, Kp, T, N, hop, F, B, s == 0 ? 0 : 1, grad_pred);
            flops += 2.0 * M * (double)N * 2 * Mb;
        }
    }
    ddsp_prof_end(ctx, st, flops, 4.0 * B * T * (2.0 * n_scale + (grad_pred ? 1.0 : 0.0)));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
