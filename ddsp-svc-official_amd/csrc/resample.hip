// SURVEY 8(f) rank 3 - device-side sample-rate conversion: the windowed-sinc polyphase resampler the reference takes
// from torchaudio (`Resample(orig, new, lowpass_filter_width=128)`: gui.py:399-404 between the model's and the audio
// device's rate, enhancer.py:50-53,69-73 around the adaptive-key trick).  torchaudio is a third-party package that is
// not in the image: the algorithm below is its published one (torchaudio.functional.resample, `sinc_interp_hann`,
// rolloff 0.99), PARITY UNPINNED at that boundary; tests hold the kernels to an fp64 evaluation of the same formulas.
//   orig, new reduced by their gcd; base = min(orig, new) * rolloff; width = ceil(lowpass_width * orig / base);
//   tap[p][k] = sinc(pi t) * cos^2(pi t / (2 lowpass_width)) * base / orig,
//       t = clamp((-p / new + (k - width) / orig) * base, -lowpass_width, lowpass_width),   p < new, k < 2 width + orig;
//   out[l * new + p] = sum_k x[l * orig + k - width] * tap[p][k]   (x zero outside [0, T)),  T_out = ceil(new * T / orig).
// The taps are evaluated in fp64 and rounded to fp32 once (torchaudio computes them in fp64 too when no dtype is given);
// the table is cached in the context (one per rate pair), stored [k][phase].  Convolution: a thread owns one phase for 8
// consecutive input frames (8 outputs `new` samples apart): a tap is loaded once - the 256 phases of a workgroup read 1 KB
// rows of the table - and meets the workgroup's input window (7 orig + K samples in the LDS, broadcast reads) eight times.
// (One thread per output with the table stored [phase][k] moved 64 lanes x K floats of table per wave and output: 0.55 ms
// for 10 s of audio, bound by the L2.)  Rate pairs whose window does not fit the LDS take the one-output-per-thread kernel.
#include "common.h"

#include <math.h>

namespace {

__global__ void __launch_bounds__(256) resample_taps_kernel(float* __restrict__ taps, int orig, int nw, int width, int K,
                                                            double base, double lpw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)nw * K) return;
    const int k = (int)(i / nw), p = (int)(i - (int64_t)k * nw);   // stored [k][p]: the phases of a tap are contiguous
    double t = (-(double)p / (double)nw + (double)(k - width) / (double)orig) * base;
    t = t < -lpw ? -lpw : (t > lpw ? lpw : t);
    const double c = cos(t * M_PI / lpw / 2.0);
    const double tp = t * M_PI;
    const double s = tp == 0.0 ? 1.0 : sin(tp) / tp;
    taps[i] = (float)(s * c * c * (base / (double)orig));
}

__global__ void __launch_bounds__(256) resample_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                       int64_t T, int64_t T_out, int orig, int nw, int width, int K,
                                                       int64_t total, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / T_out, o = i - b * T_out;
        const int64_t l = o / nw;
        const int p = (int)(o - l * nw);
        const float* xr = x + b * T;
        const float* tp = taps + p;   // [k][phase]
        const int64_t first = l * orig - width;
        int k0 = first < 0 ? (int)(-first) : 0;
        int k1 = first + K > T ? (int)(T - first) : K;
        // four partial sums (the order of a long fp32 sum matters at the 1e-7 level; a blocked sum is closer to the exact
        // value than a running one)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = k0;
        for (; k + 3 < k1; k += 4) {
            a0 = fmaf(xr[first + k], tp[(int64_t)k * nw], a0);
            a1 = fmaf(xr[first + k + 1], tp[(int64_t)(k + 1) * nw], a1);
            a2 = fmaf(xr[first + k + 2], tp[(int64_t)(k + 2) * nw], a2);
            a3 = fmaf(xr[first + k + 3], tp[(int64_t)(k + 3) * nw], a3);
        }
        for (; k < k1; ++k) a0 = fmaf(xr[first + k], tp[(int64_t)k * nw], a0);
        out[i] = (a0 + a1) + (a2 + a3);
    }
}

constexpr int RS_RB = 8;   // input frames per thread
__global__ void __launch_bounds__(256) resample_blocked_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                               int64_t T, int64_t T_out, int orig, int nw, int width, int K,
                                                               float* __restrict__ out) {
    extern __shared__ float win[];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.y * RS_RB, b = blockIdx.z;
    const int wlen = (RS_RB - 1) * orig + K;
    const int64_t first = l0 * orig - width;
    const float* xr = x + b * T;
    for (int i = threadIdx.x; i < wlen; i += 256) {
        const int64_t pos = first + i;
        win[i] = (pos >= 0 && pos < T) ? xr[pos] : 0.f;
    }
    __syncthreads();
    if (p >= nw) return;
    float acc[RS_RB][2];
#pragma unroll
    for (int r = 0; r < RS_RB; ++r) acc[r][0] = acc[r][1] = 0.f;
    const float* tp = taps + p;
    int k = 0;
    for (; k + 1 < K; k += 2) {
        const float t0 = tp[(int64_t)k * nw], t1 = tp[(int64_t)(k + 1) * nw];
#pragma unroll
        for (int r = 0; r < RS_RB; ++r) {
            acc[r][0] = fmaf(win[r * orig + k], t0, acc[r][0]);
            acc[r][1] = fmaf(win[r * orig + k + 1], t1, acc[r][1]);
        }
    }
    if (k < K) {
        const float t0 = tp[(int64_t)k * nw];
#pragma unroll
        for (int r = 0; r < RS_RB; ++r) acc[r][0] = fmaf(win[r * orig + k], t0, acc[r][0]);
    }
#pragma unroll
    for (int r = 0; r < RS_RB; ++r) {
        const int64_t o = (l0 + r) * nw + p;
        if (o < T_out) out[b * T_out + o] = acc[r][0] + acc[r][1];
    }
}

int gcd_int(int a, int b) {
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    return a;
}

}  // namespace

// table kinds 100.. (tables.hip keeps 0..): key n0 = orig / g, n1 = (new / g) * 8192 + lowpass width (both fit an int: the
// reduced rates are < 65536, the width <= 4096).  At most RS_TABLES_MAX tap tables live in a context: `Enhancer.enhance` with
// adaptive_key='auto' asks for two new rate pairs per distinct key, so a long-running server evicts the least recently used
// one (a hipFree, i.e. a device synchronisation, like the first use of any table) instead of filling the cache.
constexpr int TAB_RESAMPLE = 100;
constexpr int RS_TABLES_MAX = 8;

extern "C" int64_t ddsp_resample_length(int64_t T, int orig_freq, int new_freq) {
    if (T < 0 || orig_freq < 1 || new_freq < 1) return -1;
    const int g = gcd_int(orig_freq, new_freq);
    const int64_t o = orig_freq / g, n = new_freq / g;
    return (n * T + o - 1) / o;
}

extern "C" int ddsp_resample(ddsp_ctx* ctx, void* stream, const float* x, int64_t B, int64_t T, int orig_freq, int new_freq,
                             int lowpass_filter_width, float* out) {
    DDSP_REQUIRE(ctx, ctx && x && out, "ddsp_resample: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && T >= 1 && orig_freq >= 1 && new_freq >= 1 && lowpass_filter_width >= 1 &&
                          lowpass_filter_width <= 4096,
                 "ddsp_resample: bad argument");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int g = gcd_int(orig_freq, new_freq);
    const int orig = orig_freq / g, nw = new_freq / g;
    DDSP_REQUIRE(ctx, orig < 65536 && nw < 65536, "ddsp_resample: rate ratio too fine (reduced rates must be < 65536)");
    const double rolloff = 0.99;
    const double base = (double)(orig < nw ? orig : nw) * rolloff;
    const int width = (int)ceil((double)lowpass_filter_width * (double)orig / base);
    const int K = 2 * width + orig;
    DDSP_REQUIRE(ctx, (int64_t)nw * K < (1 << 28), "ddsp_resample: tap table too large");
    float* taps = nullptr;
    const int key0 = orig, key1 = nw * 8192 + lowpass_filter_width;
    const uint64_t now = ++ctx->table_clock;
    int n_rs = 0, lru = -1;
    for (int i = 0; i < ctx->n_tables; ++i) {
        ddsp_table& t = ctx->tables[i];
        if (t.kind != TAB_RESAMPLE) continue;
        ++n_rs;
        if (t.n0 == key0 && t.n1 == key1) {
            taps = t.dev;
            t.last_use = now;
        } else if (lru < 0 || t.last_use < ctx->tables[lru].last_use) {
            lru = i;
        }
    }
    if (!taps) {
        if ((n_rs >= RS_TABLES_MAX || ctx->n_tables >= 64) && lru >= 0) {
            // kernels of earlier calls may still read the evicted table: hipFree waits for the device
            DDSP_HIP(ctx, hipFree(ctx->tables[lru].dev));
            ctx->tables[lru] = ctx->tables[--ctx->n_tables];
        }
        if (ctx->n_tables >= 64) return ddsp_fail(ctx, DDSP_ERR_OOM, "table cache full", "");
        const size_t bytes = (size_t)nw * K * sizeof(float);
        hipError_t e = hipMalloc((void**)&taps, bytes);
        if (e != hipSuccess) return ddsp_fail(ctx, DDSP_ERR_OOM, "resample taps hipMalloc", hipGetErrorString(e));
        hipLaunchKernelGGL(resample_taps_kernel, dim3((unsigned)(((int64_t)nw * K + 255) / 256)), dim3(256), 0, st, taps, orig,
                           nw, width, K, base, (double)lowpass_filter_width);
        DDSP_LAUNCH_CHECK(ctx);
        ddsp_table& t = ctx->tables[ctx->n_tables++];
        t.kind = TAB_RESAMPLE;
        t.n0 = key0;
        t.n1 = key1;
        t.dev = taps;
        t.bytes = bytes;
        t.last_use = now;
    }
    const int64_t T_out = ((int64_t)nw * T + orig - 1) / orig;
    const int64_t total = B * T_out;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    ddsp_prof_begin(ctx, st, PF_OTHER);
    const size_t lds = ((size_t)(RS_RB - 1) * orig + K) * sizeof(float);
    const int64_t n_frames = (T_out + nw - 1) / nw;
    if (lds <= 64 * 1024 && B < 65536 && (n_frames + RS_RB - 1) / RS_RB < 65536)
        hipLaunchKernelGGL(resample_blocked_kernel, dim3((unsigned)((nw + 255) / 256), (unsigned)((n_frames + RS_RB - 1) / RS_RB), (unsigned)B),
                           dim3(256), lds, st, x, taps, T, T_out, orig, nw, width, K, out);
    else
        hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, taps, T, T_out, orig, nw, width, K, total,
                           out);
    ddsp_prof_end(ctx, st, 2.0 * total * K, 4.0 * (B * T + total));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
