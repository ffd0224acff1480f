// SURVEY 8(f) rank 1 - the NSF-HiFiGAN post-net of the reference's `Enhancer` (enhancer.py:24-101, nsf_hifigan/models.py:
// 106-276): harmonic source (SineGen + SourceModuleHnNSF), the generator's convolution stack, and the log-mel front end
// (nsf_hifigan/nvSTFT.py:65-119).  Batch 1 (the reference calls it with one utterance), activations frame-major (T, C).
//   * every Conv1d / ConvTranspose1d with more than one input channel is a GEMM on the fp32 matrix pipe with an implicit
//     im2col loader (gemm_f32.h, A_CONVK: k taps `dil` frames apart, "same" padding, the preceding leaky-ReLU applied while
//     the operand is loaded; bias and the residual / source addition in the epilogue).  A transposed convolution of stride
//     u and kernel 2u is the 3-tap case: output frame t*u + r takes inputs t-1, t (r < u/2) or t, t+1 (r >= u/2), so with the
//     weights re-packed to (u*Cout, 3*Cin) - zeros where a tap does not reach - the GEMM's (T, u*Cout) result IS the
//     (T*u, Cout) output;
//   * the source: per harmonic the phase is the running sum of f0*h/sr over samples (fp64; the reference's cumsum with its
//     wrap-around bookkeeping differs from any other integer-wrapped sum by whole cycles only), frame prefix by a scan over
//     frames, samples in closed form inside a frame; tanh(Linear(9 -> 1)) fused;
//   * the 1-channel strided "noise" convolutions, the 1-output-channel post convolution (+ tanh) and the mean over the
//     residual blocks are small streaming kernels.
// These kernels are correct-first: fp32 MFMA on the register-staged GEMM kernel, no tuning (the enhancer is the step after
// the hot path, not part of the benchmarked one).
#include "gemm_f32.h"
#include "gemm_ws.h"

#include <stdlib.h>

#include <math.h>

namespace {

constexpr int NH = 9;   // harmonics of the source module (harmonic_num = 8, nsf_hifigan/models.py:227-230)

// ---- source -------------------------------------------------------------------------------------------------------------
// prefix[l][h] = sum over frames l' < l of upp * rad[l'][h] (mod 1), rad = frac(f0 * (h + 1) / sr) (+ rand_ini on frame 0), fp64
// One wavefront per harmonic; lane l owns the frames [l*per, (l+1)*per): it sums their increments (mod 1), the lanes exchange an
// exclusive scan of those sums, and each lane then writes the prefixes of its own frames.  (One lane per harmonic walking all
// frames took 130 us at 860 frames - a serial chain of fmod, divide and fp64 add per frame.)  The running value is reduced
// mod 1 after every addition as before, so every partial result stays below 2 and the fp64 rounding stays at the 1e-16 level.
__global__ void __launch_bounds__(64) nsf_frame_prefix_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                              int L, int upp, float sr, double* __restrict__ prefix) {
    const int h = blockIdx.x, lane = threadIdx.x;
    const int per = (L + 63) / 64;
    const int l0 = lane * per, l1 = l0 + per < L ? l0 + per : L;
    auto inc = [&](int l) -> double {
        float rad = fmodf(__fdiv_rn(f0[l] * (float)(h + 1), sr), 1.0f);      // (fn / sampling_rate) % 1 in fp32
        if (l == 0) rad += rand_ini[h];
        return (double)upp * (double)rad;
    };
    double part = 0.0;
    for (int l = l0; l < l1; ++l) {
        part += inc(l);
        part -= floor(part);
    }
    // exclusive scan (mod 1) of the lane sums
    double incl = part;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double n = __shfl_up(incl, o, 64);
        if (lane >= o) {
            incl += n;
            incl -= floor(incl);
        }
    }
    double acc = incl - part;
    acc -= floor(acc);
    for (int l = l0; l < l1; ++l) {
        prefix[(int64_t)l * NH + h] = acc;
        acc += inc(l);
        acc -= floor(acc);
    }
}

__global__ void __launch_bounds__(256) nsf_source_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                         const double* __restrict__ prefix, const float* __restrict__ w,
                                                         const float* __restrict__ b, int L, int upp, float sr, float amp,
                                                         float* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= (int64_t)L * upp) return;
    const int l = (int)(n / upp), j = (int)(n - (int64_t)l * upp);
    float s = b[0];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float rad = fmodf(__fdiv_rn(f0[l] * (float)(h + 1), sr), 1.0f);
        if (l == 0) rad += rand_ini[h];
        double ph = prefix[(int64_t)l * NH + h] + (double)(j + 1) * (double)rad;
        ph -= floor(ph);
        const float sine = (float)sin(ph * 6.283185307179586) * amp;
        s = fmaf(w[h], sine, s);
    }
    out[n] = tanhf(s);
}

// ---- 1-channel strided convolution: out[t][c] = b[c] + sum_j w[c][j] * src[t*s - pad + j] ----------------------------------------
// A thread owns channel c for NC_TT consecutive output frames: each tap weight is loaded once and used NC_TT times against
// the block's source window in the LDS (broadcast reads); 256 / C frame groups share a block when C < 256.  (One thread per
// output re-read its K weights from rows K floats apart: 1.1 ms of the generator's 8.3 at 860 frames.)
constexpr int NC_TT = 8;
__global__ void __launch_bounds__(256) nsf_noise_conv_kernel(const float* __restrict__ src, int64_t T_src,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             int64_t T_out, int C, int K, int stride, int pad,
                                                             float* __restrict__ out) {
    extern __shared__ float win[];
    const int Cb = C < 256 ? C : 256;            // channels per block (blockIdx.y walks wider layers)
    const int G = 256 / Cb;                      // frame groups per block
    const int c = blockIdx.y * Cb + threadIdx.x % Cb, gq = threadIdx.x / Cb;
    const int64_t t_block = (int64_t)blockIdx.x * G * NC_TT;
    const int wlen = (G * NC_TT - 1) * stride + K;
    const int64_t first = t_block * stride - pad;
    for (int i = threadIdx.x; i < wlen; i += 256) {
        const int64_t p = first + i;
        win[i] = (p >= 0 && p < T_src) ? src[p] : 0.f;
    }
    __syncthreads();
    if (gq >= G || c >= C) return;
    const int64_t t0 = t_block + (int64_t)gq * NC_TT;
    float acc[NC_TT];
    const float bc = b[c];
#pragma unroll
    for (int i = 0; i < NC_TT; ++i) acc[i] = bc;
    const float* wr = w + (int64_t)c * K;
    const float* s0 = win + gq * NC_TT * stride;
    for (int j = 0; j < K; ++j) {
        const float wj = wr[j];
#pragma unroll
        for (int i = 0; i < NC_TT; ++i) acc[i] = fmaf(wj, s0[i * stride + j], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < NC_TT; ++i)
        if (t0 + i < T_out) out[(t0 + i) * C + c] = acc[i];
}

// ---- post: tanh(conv7(leaky_relu(x, 0.01))) with one output channel ------------------------------------------------------------
// A workgroup's 256 outputs need rows [t0 - (K-1)/2, t0 + 256 + (K-1)/2) of x: they are loaded once, coalesced and activated,
// into the LDS (row pitch C + 1), and every thread reads its K x C window from there.  (Reading it from global memory, thread
// t walked K rows of 64 bytes 64 bytes apart from its neighbours': 110 us for 440 k samples.)  Layers wider than the LDS
// budget take the direct path.
__global__ void __launch_bounds__(256) nsf_post_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, int64_t T, int C, int K, float slope,
                                                       float* __restrict__ out, int use_lds) {
    extern __shared__ float win[];
    const int64_t t0 = (int64_t)blockIdx.x * 256, t = t0 + threadIdx.x;
    const int hk = (K - 1) / 2;
    if (use_lds) {
        const int rows = 256 + K - 1, P = C + 1;
        for (int i = threadIdx.x; i < rows * C; i += 256) {
            const int r = i / C, c = i - r * C;
            const int64_t p = t0 - hk + r;
            float v = 0.f;
            if (p >= 0 && p < T) {
                v = x[p * C + c];
                v = v > 0.f ? v : v * slope;
            }
            win[r * P + c] = v;
        }
        __syncthreads();
        if (t >= T) return;
        float acc = b[0];
        for (int j = 0; j < K; ++j) {
            const float* xr = win + (threadIdx.x + j) * P;
            for (int c = 0; c < C; ++c) acc = fmaf(w[j * C + c], xr[c], acc);
        }
        out[t] = tanhf(acc);
        return;
    }
    if (t >= T) return;
    float acc = b[0];
    for (int j = 0; j < K; ++j) {
        const int64_t p = t + j - hk;
        if (p < 0 || p >= T) continue;
        const float* xr = x + p * C;
        for (int c = 0; c < C; ++c) {
            const float v = xr[c];
            acc = fmaf(w[j * C + c], v > 0.f ? v : v * slope, acc);
        }
    }
    out[t] = tanhf(acc);
}

__global__ void __launch_bounds__(256) nsf_mean_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ c, int n_terms, int64_t n,
                                                       float* __restrict__ out, float* __restrict__ out_act, float slope) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = a[i];
        if (n_terms > 1) s += b[i];
        if (n_terms > 2) s += c[i];
        s = s / (float)n_terms;
        if (out) out[i] = s;
        if (out_act) out_act[i] = s > 0.f ? s : s * slope;
    }
}
// the same for a split activated output: a thread owns one group of 8 consecutive elements
__global__ void __launch_bounds__(256) nsf_mean_split_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ c, int n_terms, int64_t groups,
                                                             float* __restrict__ out, float* __restrict__ out_act, float slope) {
    for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * 256) {
        float x[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 s = *(const f32x4*)(a + 8 * gi + 4 * h);
            if (n_terms > 1) s += *(const f32x4*)(b + 8 * gi + 4 * h);
            if (n_terms > 2) s += *(const f32x4*)(c + 8 * gi + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] = s[j] / (float)n_terms;
                x[4 * h + j] = s[j] > 0.f ? s[j] : s[j] * slope;
            }
            if (out) *(f32x4*)(out + 8 * gi + 4 * h) = s;
        }
        ddsp_u32x4 hi, lo;
        ddsp_split8(x, hi, lo);
        *(ddsp_u32x4*)(out_act + 8 * gi) = hi;
        *(ddsp_u32x4*)(out_act + 8 * gi + 4) = lo;
    }
}

// ---- log-mel: out[t][m] = log(max(sum_f mel[m][f] * sqrt(re^2 + im^2 + 1e-9), clip)) ------------------------------------------
__global__ void __launch_bounds__(256) nsf_magnitude_kernel(const float* __restrict__ spec, int64_t rows, int bins, int ld,
                                                            float* __restrict__ mag, int ldm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ldm) return;
    const int64_t r = i / ldm;
    const int f = (int)(i - r * ldm);
    float v = 0.f;
    if (f < bins) {
        const float re = spec[r * ld + 2 * f], im = spec[r * ld + 2 * f + 1];
        v = sqrtf(re * re + im * im + 1e-9f);
    }
    mag[i] = v;
}
__global__ void __launch_bounds__(256) nsf_logclamp_kernel(float* __restrict__ x, int64_t n, float clip) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = logf(fmaxf(x[i], clip));
}

// ---- convolutions of the narrow stages (16 and 32 channels, Cin == Cout) -------------------------------------------------------
// The implicit-im2col GEMM wastes 50-75 % of a 64-wide tile on them and re-reads the input once per tap from the L2.  Here a
// wavefront owns 64 consecutive frames: the window it needs (64 + 2 halo rows, activated while it is staged) and the whole
// weight (k x C x C, transposed to [tap][ci][co]) sit in the LDS, the products run on the fp32 matrix pipe with tiles that
// fit the channel count exactly - v_mfma_f32_16x16x4_f32 for C = 16 (4 frame tiles per window), v_mfma_f32_32x32x2_f32 for
// C = 32 (2 frame tiles) -, a weight fragment being fetched once per window and used for every tile.  Window rows are
// C + 1 floats apart (the 16 / 32 rows a fragment reads then fall on different banks).
struct ConvSmallArgs {
    const float* x;
    const float* w;      // packed (C, k*C), column tap*C + ci
    const float* bias;
    const float* res;
    float* out;
    float* out_act;
    int64_t T;
    int ktaps, dil;
    float in_slope, act_slope;
};
constexpr int CS_TW = 64;   // frames per window
constexpr int CS_MAX_HALO = 25;   // (k - 1) / 2 * dilation the register prefetch is sized for (11 taps, dilation 5)

template <int C>
__global__ void __launch_bounds__(256) conv_small_kernel(ConvSmallArgs g) {
    extern __shared__ float lds[];
    constexpr int P = C + 1;
    const int halo = (g.ktaps - 1) / 2 * g.dil, rows = CS_TW + 2 * halo;
    float* const wl = lds;                                             // [tap][ci][co]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* const win = lds + g.ktaps * C * C + wave * rows * P;        // this wave's window
    for (int i = threadIdx.x; i < g.ktaps * C * C; i += 256) {         // read in storage order (coalesced), scatter into the LDS
        const int co = i / (g.ktaps * C), r = i % (g.ktaps * C);       // r = tap*C + ci
        wl[r * C + co] = g.w[i];
    }
    __syncthreads();
    const int64_t nchunks = (g.T + CS_TW - 1) / CS_TW;
    // the next window's rows travel global -> registers while the current one feeds the matrix pipe, and are written to the
    // LDS (activated) after its last read: one wave per SIMD cannot hide a global-load latency per row otherwise
    constexpr int NV = ((CS_TW + 2 * CS_MAX_HALO) * C / 4 + 63) / 64;
    const int nvec = rows * C / 4;
    f32x4 pre[NV];
    auto load_window = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            const int64_t t = t0 - halo + e / C;
            pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lane + 64 * i < nvec && t >= 0 && t < g.T) pre[i] = *(const f32x4*)(g.x + t * C + e % C);
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            if (lane + 64 * i < nvec) {
                float* p = win + (e / C) * P + e % C;
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = pre[i][j] > 0.f ? pre[i][j] : pre[i][j] * g.in_slope;
            }
        }
    };
    const int64_t first = (int64_t)blockIdx.x * 4 + wave, step = (int64_t)gridDim.x * 4;
    if (first < nchunks) {
        load_window(first * CS_TW);
        store_window();
    }
    for (int64_t chunk = first; chunk < nchunks; chunk += step) {
        const int64_t t0 = chunk * CS_TW;
        const bool more = chunk + step < nchunks;
        if (more) load_window((chunk + step) * CS_TW);
        __builtin_amdgcn_wave_barrier();
        if constexpr (C == 16) {
            f32x4 acc[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int li = lane & 15, lk = lane >> 4;
            for (int tap = 0; tap < g.ktaps; ++tap) {
                const float* wr = wl + (tap * C + lk) * C + li;
                const float* xr = win + (halo + (tap - (g.ktaps - 1) / 2) * g.dil + li) * P + lk;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float b = wr[4 * qq * C];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[tt * 16 * P + 4 * qq], b, acc[tt], 0, 0, 0);
                }
            }
            // D[4*(l>>4) + r][l & 15]
            const float bc = g.bias ? g.bias[li] : 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t t = t0 + tt * 16 + 4 * lk + r;
                    if (t < g.T) {
                        const int64_t o = t * C + li;
                        float y = acc[tt][r] + bc;
                        if (g.res) y += g.res[o];
                        if (g.out) g.out[o] = y;
                        if (g.out_act) g.out_act[o] = y > 0.f ? y : y * g.act_slope;
                    }
                }
        } else {
            f32x16 acc[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
            const int li = lane & 31, lk = lane >> 5;
            for (int tap = 0; tap < g.ktaps; ++tap) {
                const float* wr = wl + (tap * C + lk) * C + li;
                const float* xr = win + (halo + (tap - (g.ktaps - 1) / 2) * g.dil + li) * P + lk;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const float b = wr[2 * kk * C];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[tt * 32 * P + 2 * kk], b, acc[tt], 0, 0, 0);
                }
            }
            // D[(r & 3) + 8*(r >> 2) + 4*(l >> 5)][l & 31]
            const float bc = g.bias ? g.bias[li] : 0.f;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t t = t0 + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (t < g.T) {
                        const int64_t o = t * C + li;
                        float y = acc[tt][r] + bc;
                        if (g.res) y += g.res[o];
                        if (g.out) g.out[o] = y;
                        if (g.out_act) g.out_act[o] = y > 0.f ? y : y * g.act_slope;
                    }
                }
        }
        __builtin_amdgcn_wave_barrier();   // the window is re-staged next
        if (more) store_window();
    }
}

// ---- one residual pair of the 16-channel stage in ONE kernel (round 3) ------------------------------------------------------
// nsf_hifigan/models.py ResBlock1: xt = c1_dilated(leaky_relu(x)); xt = c2(leaky_relu(xt)); x = xt + x.  As two conv_small launches
// the stage moves four 28 MB tensors per convolution at 860 frames (activated input and residual in, raw and activated
// output out).  Here only x is read and the result written, and c1's output never leaves the CU: a wavefront owns 64 output frames,
// computes c1 for those frames plus c2's halo (5 tiles of 16 frames), writes it activated - and zeroed outside the signal,
// c2 pads ITS input with zeros - into a second LDS window, and runs c2 from there.  Both weights sit in the LDS.
struct ConvPairRawArgs {
    const float* x;       // (T, C) raw: the window (activated while it is staged) AND the residual
    const float *w1, *b1, *w2, *b2;
    float* out;           // may be null
    float* out_act;       // may be null
    int64_t T;
    int ktaps, dil;
    float slope;          // of the activations in front of c1 and c2, and of out_act
};
typedef uint32_t ddsp_u32x2 __attribute__((ext_vector_type(2)));
constexpr int CP_MID = 80;          // c1 output rows per window: 64 + 2 * 5 halo rows of c2, whole 16-row tiles
constexpr int CP_MAX_HALO = 30;     // (k - 1) / 2 * (dil + 1)

__global__ void __launch_bounds__(256) conv_pair16_kernel(ConvPairRawArgs g) {
    constexpr int C = 16, P = C + 1;
    extern __shared__ float lds[];
    const int h2 = (g.ktaps - 1) / 2, h1 = h2 * g.dil, halo = h1 + h2, rows = CS_TW + 2 * halo;
    float* const wl1 = lds;                                            // [tap][ci][co]
    float* const wl2 = lds + g.ktaps * C * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* const win = lds + 2 * g.ktaps * C * C + wave * (rows + CP_MID) * P;   // x_act window, then the c1 output window
    float* const mid = win + rows * P;
    for (int i = threadIdx.x; i < g.ktaps * C * C; i += 256) {
        const int co = i / (g.ktaps * C), r = i % (g.ktaps * C);
        wl1[r * C + co] = g.w1[i];
        wl2[r * C + co] = g.w2[i];
    }
    __syncthreads();
    const int64_t nchunks = (g.T + CS_TW - 1) / CS_TW;
    constexpr int NV = ((CS_TW + 2 * CP_MAX_HALO) * C / 4 + 63) / 64;
    const int nvec = rows * C / 4;
    f32x4 pre[NV];
    auto load_window = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            const int64_t t = t0 - halo + e / C;
            pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lane + 64 * i < nvec && t >= 0 && t < g.T) pre[i] = *(const f32x4*)(g.x + t * C + e % C);
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            if (lane + 64 * i < nvec) {
                float* p = win + (e / C) * P + e % C;
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = pre[i][j] > 0.f ? pre[i][j] : pre[i][j] * g.slope;
            }
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    const float bc1 = g.b1 ? g.b1[li] : 0.f, bc2 = g.b2 ? g.b2[li] : 0.f;
    const int64_t first = (int64_t)blockIdx.x * 4 + wave, step = (int64_t)gridDim.x * 4;
    if (first < nchunks) {
        load_window(first * CS_TW);
        store_window();
    }
    for (int64_t chunk = first; chunk < nchunks; chunk += step) {
        const int64_t t0 = chunk * CS_TW;
        const bool more = chunk + step < nchunks;
        if (more) load_window((chunk + step) * CS_TW);
        __builtin_amdgcn_wave_barrier();
        // ---- c1 on frames t0 - h2 .. t0 - h2 + 79 (window row of output u at tap: u + h1 + (tap - h2) * dil) ----
        {
            f32x4 acc[5];
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int tap = 0; tap < g.ktaps; ++tap) {
                const float* wr = wl1 + (tap * C + lk) * C + li;
                const float* xr = win + (h1 + (tap - h2) * g.dil + li) * P + lk;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float b = wr[4 * qq * C];
#pragma unroll
                    for (int tt = 0; tt < 5; ++tt) {
                        // (the fifth tile reads up to 80 - (64 + 2 h2) rows past the window: inside this wave's LDS region,
                        // the rows of `mid`; its surplus outputs are never read by c2)
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[tt * 16 * P + 4 * qq], b, acc[tt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();      // every read of the x window is done before `mid` (behind it) is written
#pragma unroll
            for (int tt = 0; tt < 5; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = tt * 16 + 4 * lk + r;
                    const int64_t t = t0 - h2 + u;
                    float y = acc[tt][r] + bc1;
                    y = y > 0.f ? y : y * g.slope;
                    mid[u * P + li] = (t >= 0 && t < g.T) ? y : 0.f;       // c2 zero-pads its input
                }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- c2 (dilation 1) on frames t0 .. t0 + 63: mid row of output o at tap = o + tap ----
        {
            f32x4 acc[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int tap = 0; tap < g.ktaps; ++tap) {
                const float* wr = wl2 + (tap * C + lk) * C + li;
                const float* xr = mid + (tap + li) * P + lk;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float b = wr[4 * qq * C];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[tt * 16 * P + 4 * qq], b, acc[tt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t t = t0 + tt * 16 + 4 * lk + r;
                    if (t < g.T) {
                        const int64_t o = t * C + li;
                        const float y = acc[tt][r] + bc2 + g.x[o];
                        if (g.out) g.out[o] = y;
                        if (g.out_act) g.out_act[o] = y > 0.f ? y : y * g.slope;
                    }
                }
        }
        __builtin_amdgcn_wave_barrier();   // both windows are re-staged next
        if (more) store_window();
    }
}

// ---- the 32-channel stage with split-bf16 products -----------------------------------------------------------------------------
// Same ownership as conv_small_kernel (a wavefront owns 64 frames, window + whole weight in the LDS), but window and weight
// are converted to bf16 hi / lo halves ONCE while they are staged - row image [32 hi | 32 lo], 144 bytes apart so that the
// 16-byte fragment reads of 16 consecutive rows fall on different banks - and every fp32 product is three
// v_mfma_f32_32x32x16_bf16 (lo*hi + hi*lo + hi*hi): 12 matrix instructions of 32 cycles per tap and window instead of 32 of
// 64 cycles on the fp32 pipe, which leaves the kernel to the memory system (four 28 MB tensors per convolution).
constexpr int CB_PITCH = 144 / 4;   // dwords per row image
typedef __bf16 nsf_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){a, b}, bf16x2v));
    const f32x2v rem = (f32x2v){a, b} - (f32x2v){__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, bf16x2v));
}
// ---- one residual pair of the 16-channel stage with split-bf16 products (round 3) ---------------------------------------------
// x -> x + c2(leaky_relu(c1_dilated(leaky_relu(x)))) reading only x and writing only the result: 56 MB per pair at 860 frames
// where the two-launch form moves 224 MB (activated copies written by the producer for the consumer, the residual read apart).
// A wavefront owns RT - 2 h2 output frames (RT = 16 NT rows; h2 = c2's halo): it stages the window of x it needs - activated and
// converted to bf16 hi / lo halves once, row image [16 hi | 16 lo] 80 bytes apart - computes c1 on RT rows as
// D[co][frame] = W[co][k] X[k][frame] with v_mfma_f32_16x16x32_bf16 (K = two taps x 16 input channels, three instructions per
// fp32 product), writes the activated result (zero outside the signal, as c2 pads it) over the START of the same window - c1
// is done with it - in the same row image, and runs c2 from there.  The accumulator layout (four consecutive output channels
// of one frame per lane) makes both the LDS write and the global store whole 16-byte pieces.
constexpr int P16 = 20;    // dwords per row image (80 bytes: sixteen consecutive rows start in sixteen different bank quads)
constexpr int PW16 = 36;   // dwords per weight row image: one (tap pair, co) = [2 taps x 16 ci hi | the same lo], 144 bytes apart
template <int NT>
__global__ void __launch_bounds__(512) conv_pair16_bf16_kernel(ConvPairRawArgs g) {
    constexpr int C = 16, RT = 16 * NT;
    extern __shared__ uint32_t ldsu[];
    const int h2 = (g.ktaps - 1) / 2, h1 = h2 * g.dil, npair = (g.ktaps + 1) / 2;
    const int tw = RT - 2 * h2, rows = RT + 2 * h1;       // window row w <-> frame t0 - h2 - h1 + w
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t* const wl1 = ldsu;
    uint32_t* const wl2 = ldsu + npair * C * PW16;
    uint32_t* const win = ldsu + 2 * npair * C * PW16 + wave * rows * P16;
    for (int i = threadIdx.x; i < 2 * npair * C * PW16; i += blockDim.x) ldsu[i] = 0;     // (the tap after the last is a zero weight)
    __syncthreads();
    for (int i4 = threadIdx.x; i4 < g.ktaps * C * C / 4; i4 += blockDim.x) {             // four consecutive ci of one (co, tap)
        const int e = i4 * 4, co = e / (g.ktaps * C), r = e % (g.ktaps * C), tap = r / C, ci = r % C;
        const int at = ((tap >> 1) * C + co) * PW16 + ((tap & 1) * C + ci) / 2;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const f32x4 v = *(const f32x4*)((which ? g.w2 : g.w1) + e);
            uint32_t ha, la, hb, lb;
            split2(v[0], v[1], ha, la);
            split2(v[2], v[3], hb, lb);
            uint32_t* p = (which ? wl2 : wl1) + at;
            p[0] = ha;
            p[1] = hb;
            p[16] = la;
            p[17] = lb;
        }
    }
    __syncthreads();
    const int64_t nchunks = (g.T + tw - 1) / tw;
    constexpr int NV = ((RT + 2 * 25) * C / 4 + 63) / 64;
    const int nvec = rows * C / 4;
    f32x4 pre[NV];
    auto load_window = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            const int64_t t = t0 - h2 - h1 + e / C;
            pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lane + 64 * i < nvec && t >= 0 && t < g.T) pre[i] = *(const f32x4*)(g.x + t * C + e % C);
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v4 = lane + 64 * i;
            if (v4 < nvec) {
                f32x4 v = pre[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * g.slope;
                uint32_t ha, la, hb, lb;
                split2(v[0], v[1], ha, la);
                split2(v[2], v[3], hb, lb);
                uint32_t* p = win + (v4 >> 2) * P16 + 2 * (v4 & 3);
                *(ddsp_u32x2*)p = ddsp_u32x2{ha, hb};
                *(ddsp_u32x2*)(p + 8) = ddsp_u32x2{la, lb};
            }
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    const f32x4 bc1 = g.b1 ? *(const f32x4*)(g.b1 + 4 * lk) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 bc2 = g.b2 ? *(const f32x4*)(g.b2 + 4 * lk) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t first = (int64_t)blockIdx.x * nwaves + wave, step = (int64_t)gridDim.x * nwaves;
    if (first < nchunks) {
        load_window(first * tw);
        store_window();
    }
    // one convolution over NT tiles of 16 rows: `src` row of (output row u, tap) = u + tap * dd
    auto conv = [&](const uint32_t* wl, const uint32_t* src, int dd, f32x4 (&acc)[NT]) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < npair; ++j) {
            const uint32_t* wr = wl + (j * C + li) * PW16 + 4 * lk;
            const nsf_bf16x8 wh = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)wr);
            const nsf_bf16x8 wlo = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(wr + 16));
            int tap = 2 * j + (lk >> 1);
            tap = tap < g.ktaps ? tap : g.ktaps - 1;          // the zero weight's data: any finite rows
            const uint32_t* xr = src + (li + tap * dd) * P16 + 4 * (lk & 1);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const nsf_bf16x8 xh = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(xr + tt * 16 * P16));
                const nsf_bf16x8 xl = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(xr + tt * 16 * P16 + 8));
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh, acc[tt], 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[tt], 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[tt], 0, 0, 0);
            }
        }
    };
    for (int64_t chunk = first; chunk < nchunks; chunk += step) {
        const int64_t t0 = chunk * tw;
        const bool more = chunk + step < nchunks;
        f32x4 res[NT];                                         // the residual rows in the accumulator layout (cache hits: the
#pragma unroll                                                 // window load fetched the same lines), asked for before the products
        for (int tt = 0; tt < NT; ++tt) {
            const int o = tt * 16 + li;
            res[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (o < tw && t0 + o < g.T) res[tt] = *(const f32x4*)(g.x + (t0 + o) * C + 4 * lk);
        }
        if (more) load_window((chunk + step) * tw);
        __builtin_amdgcn_wave_barrier();
        f32x4 acc[NT];
        conv(wl1, win, g.dil, acc);                            // c1 on frames t0 - h2 .. t0 - h2 + RT - 1
        __builtin_amdgcn_wave_barrier();                       // every read of the window is done: its head becomes c2's input
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int u = tt * 16 + li;
            const int64_t t = t0 - h2 + u;
            f32x4 y = acc[tt] + bc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (t >= 0 && t < g.T) ? (y[r] > 0.f ? y[r] : y[r] * g.slope) : 0.f;
            uint32_t ha, la, hb, lb;
            split2(y[0], y[1], ha, la);
            split2(y[2], y[3], hb, lb);
            uint32_t* p = win + u * P16 + 2 * lk;
            *(ddsp_u32x2*)p = ddsp_u32x2{ha, hb};
            *(ddsp_u32x2*)(p + 8) = ddsp_u32x2{la, lb};
        }
        __builtin_amdgcn_wave_barrier();
        conv(wl2, win, 1, acc);                                // c2 on frames t0 .. t0 + RT - 1, of which tw are this window's
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int o = tt * 16 + li;
            if (o < tw && t0 + o < g.T) {
                const int64_t at = (t0 + o) * C + 4 * lk;
                f32x4 y = acc[tt] + bc2 + res[tt];
                if (g.out) *(f32x4*)(g.out + at) = y;
                if (g.out_act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = y[r] > 0.f ? y[r] : y[r] * g.slope;
                    *(f32x4*)(g.out_act + at) = y;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the window is re-staged next
        if (more) store_window();
    }
}

template <int NT>   // frame tiles of 32 per window: 2, or 1 when the halo would otherwise leave fewer than 8 windows in the LDS
__global__ void __launch_bounds__(512) conv_small32_bf16_kernel(ConvSmallArgs g) {   // 4 .. 8 waves: as many windows as fit the LDS beside the weight
    constexpr int C = 32, TWK = 32 * NT;
    extern __shared__ uint32_t ldsu[];
    const int halo = (g.ktaps - 1) / 2 * g.dil, rows = TWK + 2 * halo;
    uint32_t* const wl = ldsu;                                          // [tap][co] row images of the 32 ci
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* const win = ldsu + g.ktaps * C * CB_PITCH + wave * rows * CB_PITCH;
    const int nwaves = blockDim.x >> 6;
    for (int i4 = threadIdx.x; i4 < g.ktaps * C * C / 4; i4 += blockDim.x) {   // four consecutive ci of one (co, tap)
        const int e = i4 * 4, co = e / (g.ktaps * C), r = e % (g.ktaps * C), tap = r / C, ci = r % C;
        const f32x4 v = *(const f32x4*)(g.w + e);
        uint32_t h0, l0, h1, l1;
        split2(v[0], v[1], h0, l0);
        split2(v[2], v[3], h1, l1);
        uint32_t* p = wl + (tap * C + co) * CB_PITCH + ci / 2;
        p[0] = h0;
        p[1] = h1;
        p[16] = l0;
        p[17] = l1;
    }
    __syncthreads();
    const int64_t nchunks = (g.T + TWK - 1) / TWK;
    constexpr int NV = ((TWK + 2 * CS_MAX_HALO) * C / 4 + 63) / 64;
    const int nvec = rows * C / 4;
    f32x4 pre[NV];
    auto load_window = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            const int64_t t = t0 - halo + e / C;
            pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lane + 64 * i < nvec && t >= 0 && t < g.T) pre[i] = *(const f32x4*)(g.x + t * C + e % C);
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            if (lane + 64 * i < nvec) {
                f32x4 v = pre[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * g.in_slope;
                uint32_t h0, l0, h1, l1;
                split2(v[0], v[1], h0, l0);
                split2(v[2], v[3], h1, l1);
                uint32_t* p = win + (e / C) * CB_PITCH + (e % C) / 2;
                p[0] = h0;
                p[1] = h1;
                p[16] = l0;
                p[17] = l1;
            }
        }
    };
    const int64_t first = (int64_t)blockIdx.x * nwaves + wave, step = (int64_t)gridDim.x * nwaves;
    if (first < nchunks) {
        load_window(first * TWK);
        store_window();
    }
    const int li = lane & 31, lh = lane >> 5;
    for (int64_t chunk = first; chunk < nchunks; chunk += step) {
        const int64_t t0 = chunk * TWK;
        const bool more = chunk + step < nchunks;
        if (more) load_window((chunk + step) * TWK);
        __builtin_amdgcn_wave_barrier();
        f32x16 acc[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
        for (int tap = 0; tap < g.ktaps; ++tap) {
            // fragment of lane (i, h) in k-step s: ci 16 s + 8 h .. + 7 = dwords 8 s + 4 h .. + 3 of the hi half, + 16 for lo
            const uint32_t* wr = wl + (tap * C + li) * CB_PITCH + 4 * lh;
            const uint32_t* xr = win + (halo + (tap - (g.ktaps - 1) / 2) * g.dil + li) * CB_PITCH + 4 * lh;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const nsf_bf16x8 bh = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(wr + 8 * s));
                const nsf_bf16x8 bl = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(wr + 8 * s + 16));
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const nsf_bf16x8 ah = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(xr + tt * 32 * CB_PITCH + 8 * s));
                    const nsf_bf16x8 al = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(xr + tt * 32 * CB_PITCH + 8 * s + 16));
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[tt], 0, 0, 0);
                }
            }
        }
        // D[(r & 3) + 8*(r >> 2) + 4*(l >> 5)][l & 31]: row = frame (the A operand's row), column = co
        const float bc = g.bias ? g.bias[li] : 0.f;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t t = t0 + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (t < g.T) {
                    const int64_t o = t * C + li;
                    float y = acc[tt][r] + bc;
                    if (g.res) y += g.res[o];
                    if (g.out) g.out[o] = y;
                    if (g.out_act) g.out_act[o] = y > 0.f ? y : y * g.act_slope;
                }
            }
        __builtin_amdgcn_wave_barrier();
        if (more) store_window();
    }
}

static int launch_conv_small32_bf16(ddsp_ctx* ctx, hipStream_t st, const ConvSmallArgs& g) {
    const int halo = (g.ktaps - 1) / 2 * g.dil;
    const size_t w_bytes = (size_t)g.ktaps * 32 * CB_PITCH * sizeof(uint32_t);
    // two frame tiles per window unless that leaves fewer than 8 windows: then one (11 taps at dilation 5: 6 x 114 rows -> 8 x 82)
    int nt = 2;
    if ((160 * 1024 - w_bytes) / ((size_t)(64 + 2 * halo) * CB_PITCH * sizeof(uint32_t)) < 8) nt = 1;
    const int twk = 32 * nt, rows = twk + 2 * halo;
    const size_t win_bytes = (size_t)rows * CB_PITCH * sizeof(uint32_t);
    // one wave per SIMD waits out every LDS fragment read (the same convolution: 54 us with two workgroups per CU, 78 us with
    // one): as many wavefronts (windows) per workgroup as fit beside the weight image, 8 at most
    int nw = (int)((160 * 1024 - w_bytes) / win_bytes);
    nw = nw > 8 ? 8 : nw;
    if (nw < 1) return ddsp_fail(ctx, DDSP_ERR_ARG, "ddsp_conv1d", "window too long for the LDS");
    const size_t lds = w_bytes + nw * win_bytes;
    DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)conv_small32_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                         DDSP_HIP(ctx, hipFuncSetAttribute((const void*)conv_small32_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    const int64_t nchunks = (g.T + twk - 1) / twk;
    int64_t blocks = (nchunks + nw - 1) / nw;
    if (blocks > 256) blocks = 256;
    if (nt == 2)
        hipLaunchKernelGGL(conv_small32_bf16_kernel<2>, dim3((unsigned)blocks), dim3(64 * nw), lds, st, g);
    else
        hipLaunchKernelGGL(conv_small32_bf16_kernel<1>, dim3((unsigned)blocks), dim3(64 * nw), lds, st, g);
    return DDSP_OK;
}

static bool conv_small32_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_CONV_SMALL32");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

template <int C>
static int launch_conv_small(ddsp_ctx* ctx, hipStream_t st, const ConvSmallArgs& g) {
    const int halo = (g.ktaps - 1) / 2 * g.dil, rows = CS_TW + 2 * halo;
    const size_t lds = ((size_t)g.ktaps * C * C + 4 * (size_t)rows * (C + 1)) * sizeof(float);
    if (lds > 160 * 1024) return ddsp_fail(ctx, DDSP_ERR_ARG, "ddsp_conv1d", "window too long for the LDS");
    DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)conv_small_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    const int64_t nchunks = (g.T + CS_TW - 1) / CS_TW;
    int64_t blocks = (nchunks + 3) / 4;
    const int64_t cap = 256 * (lds > 80 * 1024 ? 1 : lds > 53 * 1024 ? 2 : 3);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((conv_small_kernel<C>), dim3((unsigned)blocks), dim3(256), lds, st, g);
    return DDSP_OK;
}

struct EpiAddBias {   // y = acc + bias[n] (+ res): C = y and / or Cact = leaky_relu(y, slope) - the consumer convolutions of the
                      // generator all read an activated input, the residual paths the raw one
    float* C;
    float* Cact;
    const float* res;
    int64_t ldc;
    const float* bias;
    float slope;
    int act_split;   // Cact is written as bf16 hi/lo groups (the A operand layout of gemm::Args::A_split); vector path only
    __device__ __forceinline__ float col(int n) const { return bias ? bias[n] : 0.f; }
    __device__ __forceinline__ void operator()(int, int m, int n, float v, float cb) const {
        const int64_t o = (int64_t)m * ldc + n;
        float y = v + cb;
        if (res) y += res[o];
        if (C) C[o] = y;
        if (Cact) Cact[o] = y > 0.f ? y : y * slope;
    }
    static constexpr bool kStore4 = true;
    __device__ __forceinline__ bool vec_ok() const {
        return (((uintptr_t)C | (uintptr_t)Cact | (uintptr_t)res) % 16) == 0 && ldc % 4 == 0;
    }
    __device__ __forceinline__ void store4(int, int m, int n, f32x4 v) const {
        const int64_t o = (int64_t)m * ldc + n;
        if (bias) v += *(const gemm::f32x4_u*)(bias + n);
        if (res) v += *(const f32x4*)(res + o);
        if (C) *(f32x4*)(C + o) = v;
        if (Cact) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
            if (act_split)   // the lane 4 further on holds the other half of this group of 8 columns (same row)
                *(ddsp_u32x4*)(Cact + o) = ddsp_split4_pair(v, (n & 4) != 0, 4);
            else
                *(f32x4*)(Cact + o) = v;
        }
    }
};

// EpiAddBias for the wave-specialised kernel (gemm_ws.h): bias through the loaders' LDS copy, the residual tile (RES) too
template <bool RES>
struct WsConv {
    float* C;
    float* Cact;
    const float* res;
    int64_t ldc;
    const float* bias;
    float slope;
    int act_split;
    static constexpr bool kExtra = RES, kGated = false;
    __device__ __forceinline__ f32x4 bias4(int n) const { return bias ? *(const gemm::f32x4_u*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ float bias1(int n) const { return bias ? bias[n] : 0.f; }
    __device__ __forceinline__ const float* extra_ptr() const { return res; }
    __device__ __forceinline__ void emit4(int, int m, int n, f32x4 v, f32x4 e) const {
        const int64_t o = (int64_t)m * ldc + n;
        if constexpr (RES) v = v + e;
        if (C) *(f32x4*)(C + o) = v;
        if (Cact) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
            if (act_split)   // the lane 4 further on holds the other half of this group of 8 columns (same row)
                *(ddsp_u32x4*)(Cact + o) = ddsp_split4_pair(v, (n & 4) != 0, 4);
            else
                *(f32x4*)(Cact + o) = v;
        }
    }
    __device__ __forceinline__ void emit1(int, int m, int n, float v) const {   // (edge path: never taken, the launcher asks for whole tiles)
        const int64_t o = (int64_t)m * ldc + n;
        if (RES) v += res[o];
        if (C) C[o] = v;
        if (Cact && !act_split) Cact[o] = v > 0.f ? v : v * slope;
    }
    __host__ __device__ const float* bias_ptr() const { return bias; }
    __host__ bool vec_ok() const {
        return (((uintptr_t)C | (uintptr_t)Cact | (uintptr_t)res | (uintptr_t)bias) % 16) == 0 && ldc % 4 == 0;
    }
};

}  // namespace

// ---- the same for the 32-channel stage: v_mfma_f32_32x32x16_bf16, two k-steps per tap, row images as in conv_small32_bf16_kernel ----
// Both weights (hi / lo: 4 bytes a weight, 2 x 50 KB at 11 taps) stay in the LDS, so the number of windows (wavefronts) of a
// workgroup follows from what is left: 8 at 3 taps, 7 at 7, 3 - 5 at 11 (NT = 1: 32-row windows there when that wins).
template <int NT>
__global__ void __launch_bounds__(512) conv_pair32_bf16_kernel(ConvPairRawArgs g) {
    constexpr int C = 32, RT = 32 * NT;
    extern __shared__ uint32_t ldsu[];
    const int h2 = (g.ktaps - 1) / 2, h1 = h2 * g.dil;
    const int tw = RT - 2 * h2, rows = RT + 2 * h1;       // window row w <-> frame t0 - h2 - h1 + w
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t* const wl1 = ldsu;
    uint32_t* const wl2 = ldsu + g.ktaps * C * CB_PITCH;
    float* const bias = (float*)(ldsu + 2 * g.ktaps * C * CB_PITCH);                      // b1 | b2
    uint32_t* const win = ldsu + 2 * g.ktaps * C * CB_PITCH + 2 * C + wave * rows * CB_PITCH;
    for (int i4 = threadIdx.x; i4 < g.ktaps * C * C / 4; i4 += blockDim.x) {             // four consecutive ci of one (co, tap)
        const int e = i4 * 4, co = e / (g.ktaps * C), r = e % (g.ktaps * C), tap = r / C, ci = r % C;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const f32x4 v = *(const f32x4*)((which ? g.w2 : g.w1) + e);
            uint32_t ha, la, hb, lb;
            split2(v[0], v[1], ha, la);
            split2(v[2], v[3], hb, lb);
            uint32_t* p = (which ? wl2 : wl1) + (tap * C + co) * CB_PITCH + ci / 2;
            p[0] = ha;
            p[1] = hb;
            p[16] = la;
            p[17] = lb;
        }
    }
    if (threadIdx.x < 2 * C) {
        const float* b = threadIdx.x < C ? g.b1 : g.b2;
        bias[threadIdx.x] = b ? b[threadIdx.x % C] : 0.f;
    }
    __syncthreads();
    const int64_t nchunks = (g.T + tw - 1) / tw;
    constexpr int NV = ((RT + 2 * 25) * C / 4 + 63) / 64;
    const int nvec = rows * C / 4;
    f32x4 pre[NV];
    auto load_window = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (lane + 64 * i) * 4;
            const int64_t t = t0 - h2 - h1 + e / C;
            pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lane + 64 * i < nvec && t >= 0 && t < g.T) pre[i] = *(const f32x4*)(g.x + t * C + e % C);
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v4 = lane + 64 * i;
            if (v4 < nvec) {
                f32x4 v = pre[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * g.slope;
                uint32_t ha, la, hb, lb;
                split2(v[0], v[1], ha, la);
                split2(v[2], v[3], hb, lb);
                uint32_t* p = win + (v4 >> 3) * CB_PITCH + 2 * (v4 & 7);
                *(ddsp_u32x2*)p = ddsp_u32x2{ha, hb};
                *(ddsp_u32x2*)(p + 16) = ddsp_u32x2{la, lb};
            }
        }
    };
    const int li = lane & 31, lh = lane >> 5;
    const int64_t first = (int64_t)blockIdx.x * nwaves + wave, step = (int64_t)gridDim.x * nwaves;
    if (first < nchunks) {
        load_window(first * tw);
        store_window();
    }
    // D[co = (r & 3) + 8 (r >> 2) + 4 lh][frame = li] = bias[co] + sum over (tap, ci) of W[co][tap, ci] * src[frame + tap * dd][ci]
    auto conv = [&](const uint32_t* wl, const float* bz, const uint32_t* src, int dd, f32x16 (&acc)[NT]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *(const f32x4*)(bz + 8 * q + 4 * lh);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tt][4 * q + r] = b[r];
        }
        for (int tap = 0; tap < g.ktaps; ++tap) {
            const uint32_t* wr = wl + (tap * C + li) * CB_PITCH + 4 * lh;
            const uint32_t* xr = src + (li + tap * dd) * CB_PITCH + 4 * lh;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const nsf_bf16x8 wh = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(wr + 8 * s));
                const nsf_bf16x8 wlo = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(wr + 8 * s + 16));
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const nsf_bf16x8 xh = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(xr + tt * 32 * CB_PITCH + 8 * s));
                    const nsf_bf16x8 xl = __builtin_bit_cast(nsf_bf16x8, *(const ddsp_u32x4*)(xr + tt * 32 * CB_PITCH + 8 * s + 16));
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[tt], 0, 0, 0);
                }
            }
        }
    };
    for (int64_t chunk = first; chunk < nchunks; chunk += step) {
        const int64_t t0 = chunk * tw;
        const bool more = chunk + step < nchunks;
        f32x4 res[NT][4];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int o = tt * 32 + li;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                res[tt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (o < tw && t0 + o < g.T) res[tt][q] = *(const f32x4*)(g.x + (t0 + o) * C + 8 * q + 4 * lh);
            }
        }
        if (more) load_window((chunk + step) * tw);
        __builtin_amdgcn_wave_barrier();
        f32x16 acc[NT];
        conv(wl1, bias, win, g.dil, acc);
        __builtin_amdgcn_wave_barrier();                       // every read of the window is done: its head becomes c2's input
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int u = tt * 32 + li;
            const int64_t t = t0 - h2 + u;
            const bool in = t >= 0 && t < g.T;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[tt][4 * q + r];
                    y[r] = in ? (v > 0.f ? v : v * g.slope) : 0.f;
                }
                uint32_t ha, la, hb, lb;
                split2(y[0], y[1], ha, la);
                split2(y[2], y[3], hb, lb);
                uint32_t* p = win + u * CB_PITCH + 4 * q + 2 * lh;
                *(ddsp_u32x2*)p = ddsp_u32x2{ha, hb};
                *(ddsp_u32x2*)(p + 16) = ddsp_u32x2{la, lb};
            }
        }
        __builtin_amdgcn_wave_barrier();
        conv(wl2, bias + C, win, 1, acc);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int o = tt * 32 + li;
            if (o < tw && t0 + o < g.T) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t at = (t0 + o) * C + 8 * q + 4 * lh;
                    f32x4 y = res[tt][q];
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] += acc[tt][4 * q + r];
                    if (g.out) *(f32x4*)(g.out + at) = y;
                    if (g.out_act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = y[r] > 0.f ? y[r] : y[r] * g.slope;
                        *(f32x4*)(g.out_act + at) = y;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the window is re-staged next
        if (more) store_window();
    }
}

// ---- one ResBlock1 pair x -> x + c2(leaky_relu(c1_dilated(leaky_relu(x)))) in one launch ---------------------------------------
static size_t pair16_bf16_lds(int ktaps, int dil, int nt, int nw) {
    const int npair = (ktaps + 1) / 2, rows = 16 * nt + (ktaps - 1) * dil;
    return ((size_t)2 * npair * 16 * PW16 + (size_t)nw * rows * P16) * sizeof(uint32_t);
}
// 1 when ddsp_conv1d_pair takes this geometry in the context's arithmetic
// the 32-channel kernel's window height (tiles of 32 rows) and wavefront count for a geometry: both weights stay in the LDS
static void pair32_shape(int ktaps, int dil, int& nt, int& nw, size_t& lds) {
    const size_t fixed = ((size_t)2 * ktaps * 32 * CB_PITCH + 64) * sizeof(uint32_t);
    double best = 1e30;
    nt = 0;
    nw = 0;
    lds = 0;
    for (int cand = 2; cand >= 1; --cand) {
        const int rt = 32 * cand, tw = rt - (ktaps - 1), rows = rt + (ktaps - 1) * dil;
        if (tw < 8) continue;
        const size_t per = (size_t)rows * CB_PITCH * sizeof(uint32_t);
        if (fixed + per > 160 * 1024) continue;
        int n = (int)((160 * 1024 - fixed) / per);
        n = n > 8 ? 8 : n;
        // matrix time per output frame with n wavefronts on 4 SIMDs, computed rows over delivered frames
        const double cost = ((double)rt / tw) / (n < 4 ? n : 4);
        if (cost < best * 0.97) {
            best = cost;
            nt = cand;
            nw = n;
            lds = fixed + n * per;
        }
    }
}
// 1 when ddsp_conv1d_pair takes this geometry in the context's arithmetic
extern "C" int ddsp_conv1d_pair_supported(ddsp_ctx* ctx, int C, int ktaps, int dil) {
    if (!ctx || ktaps < 1 || ktaps % 2 == 0 || ktaps > 11 || dil < 1 || dil > 5) return 0;
    if (C == 16) return 1;
    if (C == 32 && ctx->math != DDSP_MATH_FP32) {
        int nt, nw;
        size_t lds;
        pair32_shape(ktaps, dil, nt, nw, lds);
        return nw >= 2 ? 1 : 0;
    }
    return 0;
}
extern "C" int ddsp_conv1d_pair(ddsp_ctx* ctx, void* stream, const float* x, const float* w1, const float* b1, const float* w2,
                                const float* b2, int64_t T, int C, int ktaps, int dil, float slope, float* out, float* out_act) {
    DDSP_REQUIRE(ctx, ctx && x && w1 && w2 && (out || out_act), "ddsp_conv1d_pair: null argument");
    DDSP_REQUIRE(ctx, T >= 1 && T < (1 << 30), "ddsp_conv1d_pair: bad length");
    DDSP_REQUIRE(ctx, ddsp_conv1d_pair_supported(ctx, C, ktaps, dil) == 1,
                 "ddsp_conv1d_pair: 16 channels (32 with split-bf16 products), odd tap counts up to 11, dilation up to 5 (ask ddsp_conv1d_pair_supported)");
    DDSP_REQUIRE(ctx, (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)out | (uintptr_t)out_act) % 16) == 0 &&
                          x != out && x != out_act,
                 "ddsp_conv1d_pair: 16-byte aligned tensors, not in place");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    ConvPairRawArgs a{x, w1, b1, w2, b2, out, out_act, T, ktaps, dil, slope};
    ddsp_prof_begin(ctx, st, PF_OTHER);
    if (C == 32) {
        int nt, nw;
        size_t lds;
        pair32_shape(ktaps, dil, nt, nw, lds);
        DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)conv_pair32_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                             DDSP_HIP(ctx, hipFuncSetAttribute((const void*)conv_pair32_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        const int tw = 32 * nt - (ktaps - 1);
        const int64_t nchunks = (T + tw - 1) / tw;
        int64_t blocks = (nchunks + nw - 1) / nw;
        if (blocks > 256) blocks = 256;
        if (nt == 2)
            hipLaunchKernelGGL(conv_pair32_bf16_kernel<2>, dim3((unsigned)blocks), dim3(64 * nw), lds, st, a);
        else
            hipLaunchKernelGGL(conv_pair32_bf16_kernel<1>, dim3((unsigned)blocks), dim3(64 * nw), lds, st, a);
    } else if (ctx->math == DDSP_MATH_FP32) {
        const int halo = (ktaps - 1) / 2 * (dil + 1), rows = CS_TW + 2 * halo;
        const size_t lds = ((size_t)2 * ktaps * C * C + 4 * (size_t)(rows + CP_MID) * (C + 1)) * sizeof(float);
        DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)conv_pair16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        const int64_t nchunks = (T + CS_TW - 1) / CS_TW;
        int64_t blocks = (nchunks + 3) / 4;
        const int64_t cap = 256 * (lds > 80 * 1024 ? 1 : lds > 53 * 1024 ? 2 : 3);
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(conv_pair16_kernel, dim3((unsigned)blocks), dim3(256), lds, st, a);
    } else {
        constexpr int NT = 6;
        int nw = 8;
        while (nw > 1 && pair16_bf16_lds(ktaps, dil, NT, nw) > 160 * 1024) --nw;
        const size_t lds = pair16_bf16_lds(ktaps, dil, NT, nw);
        DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)conv_pair16_bf16_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        const int tw = 16 * NT - (ktaps - 1);
        const int64_t nchunks = (T + tw - 1) / tw;
        int64_t blocks = (nchunks + nw - 1) / nw;
        if (blocks > 256) blocks = 256;       // (210 VGPRs: the 8 wavefronts of one workgroup are a CU's share)
        hipLaunchKernelGGL(conv_pair16_bf16_kernel<NT>, dim3((unsigned)blocks), dim3(64 * nw), lds, st, a);
    }
    ddsp_prof_end(ctx, st, 4.0 * T * C * C * ktaps, 8.0 * T * C);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_conv1d(ddsp_ctx* ctx, void* stream, const float* x, const float* w_packed, const float* bias, int64_t T,
                           int Cin, int Cout, int ktaps, int dil, float in_slope, const float* residual, float* out,
                           float* out_act, float act_slope, const float* w_split, int flags) {
    DDSP_REQUIRE(ctx, ctx && x && w_packed && (out || out_act), "ddsp_conv1d: null argument");
    DDSP_REQUIRE(ctx, T >= 1 && T < (1 << 30) && Cin >= 4 && Cin % 4 == 0 && Cout >= 1 && ktaps >= 1 && ktaps % 2 == 1 &&
                          ktaps <= 63 && dil >= 1 && dil <= 64,
                 "ddsp_conv1d: bad shape (Cin % 4 == 0, odd tap count)");
    DDSP_REQUIRE(ctx, x != out && x != out_act, "ddsp_conv1d: in-place convolution is not possible");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    gemm::Args g = gemm::make(x, Cin, w_packed, (int64_t)ktaps * Cin, (int)T, Cout, ktaps * Cin);
    g.Fr = (int)T;
    g.Cin = Cin;
    g.ktaps = ktaps;
    g.dil = dil;
    g.in_slope = in_slope;
    const bool x_split = (flags & DDSP_CONV_X_SPLIT) != 0, act_split = (flags & DDSP_CONV_ACT_SPLIT) != 0;
    EpiAddBias e{out, out_act, residual, Cout, bias, act_slope, act_split ? 1 : 0};
    const bool dma = in_slope == 1.0f && Cin % 32 == 0 && gemm::dma_ok(g);
    DDSP_REQUIRE(ctx, !(x_split || act_split) || (dma && ctx->math != DDSP_MATH_FP32 && w_split && gemm::dma_ok(g)),
                 "ddsp_conv1d: split operands need the LDS-DMA path (in_slope = 1, Cin % 32 == 0), split-bf16 arithmetic and w_split");
    DDSP_REQUIRE(ctx, !act_split || (out_act && Cout % 64 == 0 && (Cout <= 256 || Cout % 128 == 0) &&
                                     (((uintptr_t)out | (uintptr_t)out_act | (uintptr_t)residual) % 16) == 0),
                 "ddsp_conv1d: a split activated output needs Cout % 64 == 0 and 16-byte aligned tensors");
    auto blocks = [&](int bm, int bn) { return (int64_t)((T + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
    ddsp_prof_begin(ctx, st, PF_OTHER);
    if (Cin == Cout && Cin == 16 && !x_split && !act_split && ((uintptr_t)x % 16) == 0 && (ktaps - 1) / 2 * dil <= CS_MAX_HALO) {
        // the narrow last stage: fp32 matrix products on tiles of its own width (conv_small_kernel; 1.87 -> 0.74 ms for the 18
        // convolutions of the 16-channel stage at 860 frames).  The 32-channel instantiation is kept for measurements
        // (DDSP_CONV_SMALL32=1): at 1.48 ms per stage it does not beat the LDS-DMA GEMM's 1.28 - one wave per SIMD (105 KB of
        // LDS per workgroup) on the fp32 matrix pipe reaches a third of its rate.
        ConvSmallArgs a{x, w_packed, bias, residual, out, out_act, T, ktaps, dil, in_slope, act_slope};
        if (int rc = launch_conv_small<16>(ctx, st, a)) return rc;
    } else if (Cin == Cout && Cin == 32 && ctx->math != DDSP_MATH_FP32 && !conv_small32_enabled() && !x_split && !act_split &&
               (((uintptr_t)x | (uintptr_t)w_packed) % 16) == 0 && (ktaps - 1) / 2 * dil <= CS_MAX_HALO) {
        // the 32-channel stage in split-bf16 arithmetic: the narrow kernel with bf16 fragments
        ConvSmallArgs a{x, w_packed, bias, residual, out, out_act, T, ktaps, dil, in_slope, act_slope};
        if (int rc = launch_conv_small32_bf16(ctx, st, a)) return rc;
    } else if (Cin == Cout && Cin == 32 && conv_small32_enabled() && !x_split && !act_split && ((uintptr_t)x % 16) == 0 &&
               (ktaps - 1) / 2 * dil <= CS_MAX_HALO) {
        ConvSmallArgs a{x, w_packed, bias, residual, out, out_act, T, ktaps, dil, in_slope, act_slope};
        if (int rc = launch_conv_small<32>(ctx, st, a)) return rc;
    } else if (dma) {
        // an input that needs no activation on load: the LDS-DMA kernel with per-tap row pointers, products in the context's
        // arithmetic (split-bf16 by default, ddsp_ctx_set_math(FP32) for fp32 products)
        if (int rc = ddsp_zero_page(ctx, &g.zeros)) return rc;
        g.math = ctx->math == DDSP_MATH_FP32 ? 0 : DDSP_MATH_SPLIT_BF16;
        if (g.math == DDSP_MATH_SPLIT_BF16 && w_split) {
            g.B_split = w_split;
            g.A_split = x_split ? 1 : 0;
            if (x_split) g.B = w_split;   // (mode 8 reads only the split copies)
        }
        // Long products with pre-split weights at sizes that fill the chip CAN run on the wave-specialised kernel (gemm_ws.h) with
        // per-tap row pointers - 128x128 tiles without a residual, 128x64 with one (its tile travels through the LDS too).
        // Measured at 860 frames (rocprofv3, per launch): 128x128 without residual 53.4 -> 48.7 us; 128x64 without residual 45.2 ->
        // 44.9; with the residual tile (128x64 also for the 128-channel stage, whose rows are then staged twice) 63 us against
        // 45 - 53 on kernel_dma; the generator as a whole 3.66 ms with either choice (3.69 with the 128x128 form, 3.64 with all).
        // OFF by default: DDSP_CONV_WS bit 1 = the 128x128 form, bit 2 = the 128x64 forms as well (measurement aid).
        static int conv_ws = -1;
        if (conv_ws < 0) {
            const char* ev = getenv("DDSP_CONV_WS");
            conv_ws = ev ? atoi(ev) : 0;
        }
        if (conv_ws && g.math == DDSP_MATH_SPLIT_BF16 && g.B_split && Cout % 64 == 0 && gemm::ws_ok(g, 5, residual != nullptr)) {
            hipError_t he = hipErrorInvalidValue;
            bool ran = false;
            if (residual) {
                WsConv<true> ew{out, out_act, residual, Cout, bias, act_slope, act_split ? 1 : 0};
                if ((conv_ws & 2) && ew.vec_ok() && blocks(128, 64) >= 256) {
                    he = gemm::ws_conv_go<128, 64, WsConv<true>, 5>(st, g, ew);
                    ran = true;
                }
            } else {
                WsConv<false> ew{out, out_act, nullptr, Cout, bias, act_slope, act_split ? 1 : 0};
                if (ew.vec_ok() && Cout % 128 == 0 && blocks(128, 128) >= 256) {
                    he = gemm::ws_conv_go<128, 128, WsConv<false>, 4>(st, g, ew);
                    ran = true;
                } else if ((conv_ws & 2) && ew.vec_ok() && blocks(128, 64) >= 256) {
                    he = gemm::ws_conv_go<128, 64, WsConv<false>, 5>(st, g, ew);
                    ran = true;
                }
            }
            if (ran) {
                DDSP_HIP(ctx, he);
                ddsp_prof_end(ctx, st, 2.0 * T * (double)Cout * ktaps * Cin, 4.0 * T * ((double)Cin + Cout));
                DDSP_LAUNCH_CHECK(ctx);
                return DDSP_OK;
            }
        }
        static int conv_tile = -1;
        if (conv_tile < 0) {
            // tile choice (860 frames, generator 5.86 / 5.53 / 5.19 ms with 0 / 1 / 2): 0 = 64x64 on 4 waves everywhere,
            // 1 = 128x64 on 8 waves where that still gives 512 workgroups, 2 = also 128x128 for 128-channel multiples from 256
            // workgroups (the 128-channel stage: 1.45 -> 0.99 ms, one column tile instead of two re-reading the input)
            // 3 = also 64x128 on 4 waves for the 128-channel multiples that fill neither (the 256-channel stage at 6880 rows and
            // the transposed convolutions: 5.17 -> 4.98 ms; 4 = 128x128 there instead: 5.18)
            const char* ev = getenv("DDSP_CONV_TILE");
            conv_tile = ev ? atoi(ev) : 3;
        }
        if (Cout > 256 && blocks(128, 128) >= 512)
            gemm::dma_go<128, 128, EpiAddBias, 2, 8, gemm::A_CONVK>(st, g, 1, e);
        else if (conv_tile >= 2 && Cout % 128 == 0 && blocks(128, 128) >= 256)
            gemm::dma_go<128, 128, EpiAddBias, 2, 8, gemm::A_CONVK>(st, g, 1, e);
        else if (conv_tile >= 1 && blocks(128, 64) >= 512)
            gemm::dma_go<128, 64, EpiAddBias, 3, 8, gemm::A_CONVK>(st, g, 1, e);
        else if (conv_tile == 3 && Cout % 128 == 0 && blocks(64, 128) >= 128)
            gemm::dma_go<64, 128, EpiAddBias, 3, 4, gemm::A_CONVK>(st, g, 1, e);
        else if (conv_tile == 4 && Cout % 128 == 0 && blocks(128, 128) >= 64)
            gemm::dma_go<128, 128, EpiAddBias, 2, 8, gemm::A_CONVK>(st, g, 1, e);
        else
            gemm::dma_go<64, 64, EpiAddBias, 3, 4, gemm::A_CONVK>(st, g, 1, e);
    } else if (blocks(128, 64) >= 512)
        gemm::launch_tile<128, 64, true, true, gemm::A_CONVK, EpiAddBias, 8>(st, g, 1, e);
    else
        gemm::launch_tile<64, 64, true, true, gemm::A_CONVK, EpiAddBias, 4>(st, g, 1, e);
    ddsp_prof_end(ctx, st, 2.0 * T * (double)Cout * ktaps * Cin, 4.0 * T * ((double)Cin + Cout));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_nsf_source(ddsp_ctx* ctx, void* stream, const float* f0, const float* rand_ini, const float* lin_w,
                               const float* lin_b, int64_t L, int upp, int sr, float sine_amp, float* out) {
    DDSP_REQUIRE(ctx, ctx && f0 && rand_ini && lin_w && lin_b && out, "ddsp_nsf_source: null argument");
    DDSP_REQUIRE(ctx, L >= 1 && L < (1 << 24) && upp >= 1 && upp <= 65536 && sr >= 1, "ddsp_nsf_source: bad shape");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    int rc = ddsp_scratch_reserve_bytes(ctx, (size_t)L * NH * sizeof(double) + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    double* prefix = nullptr;
    if ((rc = ddsp_scratch_get(ctx, (size_t)L * NH * sizeof(double), (void**)&prefix))) return rc;
    hipLaunchKernelGGL(nsf_frame_prefix_kernel, dim3(NH), dim3(64), 0, st, f0, rand_ini, (int)L, upp, (float)sr, prefix);
    const int64_t n = L * upp;
    hipLaunchKernelGGL(nsf_source_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f0, rand_ini, prefix, lin_w, lin_b,
                       (int)L, upp, (float)sr, sine_amp, out);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_nsf_noise_conv(ddsp_ctx* ctx, void* stream, const float* src, int64_t T_src, const float* w,
                                   const float* b, int C, int K, int stride, int pad, int64_t T_out, float* out) {
    DDSP_REQUIRE(ctx, ctx && src && w && b && out, "ddsp_nsf_noise_conv: null argument");
    DDSP_REQUIRE(ctx, T_src >= 1 && T_out >= 1 && C >= 1 && K >= 1 && stride >= 1 && pad >= 0 && T_out * C < ((int64_t)1 << 40),
                 "ddsp_nsf_noise_conv: bad shape");
    DDSP_REQUIRE(ctx, (T_out - 1) * stride - pad < T_src, "ddsp_nsf_noise_conv: output longer than the source allows");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int Cb = C < 256 ? C : 256, G = 256 / Cb;
    const size_t lds = ((size_t)(G * NC_TT - 1) * stride + K) * sizeof(float);
    DDSP_REQUIRE(ctx, lds <= 64 * 1024, "ddsp_nsf_noise_conv: source window too long for the LDS");
    hipLaunchKernelGGL(nsf_noise_conv_kernel, dim3((unsigned)((T_out + (int64_t)G * NC_TT - 1) / ((int64_t)G * NC_TT)), (unsigned)((C + Cb - 1) / Cb)), dim3(256), lds, st, src,
                       T_src, w, b, T_out, C, K, stride, pad, out);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_nsf_post(ddsp_ctx* ctx, void* stream, const float* x, const float* w, const float* b, int64_t T, int C, int K,
                             float slope, float* out) {
    DDSP_REQUIRE(ctx, ctx && x && w && b && out, "ddsp_nsf_post: null argument");
    DDSP_REQUIRE(ctx, T >= 1 && C >= 1 && K >= 1 && K % 2 == 1, "ddsp_nsf_post: bad shape");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const size_t lds = (size_t)(256 + K - 1) * (C + 1) * sizeof(float);
    const int use_lds = lds <= 64 * 1024 ? 1 : 0;
    hipLaunchKernelGGL(nsf_post_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), use_lds ? lds : 0, st, x, w, b, T, C, K, slope, out,
                       use_lds);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_nsf_mean(ddsp_ctx* ctx, void* stream, const float* a, const float* b, const float* c, int n_terms, int64_t n,
                             float* out, float* out_act, float act_slope, int flags) {
    DDSP_REQUIRE(ctx, ctx && a && (out || out_act) && n_terms >= 1 && n_terms <= 3 && (n_terms < 2 || b) && (n_terms < 3 || c) && n >= 0,
                 "ddsp_nsf_mean: bad argument");
    if (n == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (flags & DDSP_CONV_ACT_SPLIT) {
        DDSP_REQUIRE(ctx, out_act && n % 8 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out | (uintptr_t)out_act) % 16) == 0,
                     "ddsp_nsf_mean: a split activated output needs n % 8 == 0 and 16-byte aligned tensors");
        blocks = (n / 8 + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(nsf_mean_split_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, c, n_terms, n / 8, out, out_act,
                           act_slope);
    } else
        hipLaunchKernelGGL(nsf_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, c, n_terms, n, out, out_act, act_slope);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_log_mel(ddsp_ctx* ctx, void* stream, const float* frames, const float* dft_table, const float* mel_basis,
                            int64_t n_frames, int n_fft, int n_mels, float clip, float* out) {
    // frames (n_frames, n_fft) already windowed-by-table: dft_table (2*bins_padded, n_fft) rows 2f = w*cos, 2f+1 = -w*sin;
    // mel_basis (n_mels, bins_padded) zero-padded columns; out (n_frames, n_mels)
    DDSP_REQUIRE(ctx, ctx && frames && dft_table && mel_basis && out, "ddsp_log_mel: null argument");
    DDSP_REQUIRE(ctx, n_frames >= 1 && n_frames < (1 << 24) && n_fft >= 4 && n_fft % 4 == 0 && n_mels >= 1,
                 "ddsp_log_mel: bad shape");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int bins = n_fft / 2 + 1, ldm = (bins + 3) & ~3, lds = 2 * ldm;
    int rc = ddsp_scratch_reserve_bytes(ctx, (size_t)n_frames * (lds + ldm) * sizeof(float) + 8192);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float *spec = nullptr, *mag = nullptr;
    if ((rc = ddsp_scratch_get(ctx, (size_t)n_frames * lds * sizeof(float), (void**)&spec))) return rc;
    if ((rc = ddsp_scratch_get(ctx, (size_t)n_frames * ldm * sizeof(float), (void**)&mag))) return rc;
    {
        gemm::Args g = gemm::make(frames, n_fft, dft_table, n_fft, (int)n_frames, lds, n_fft);
        gemm::EpiStore e{spec, lds, nullptr, 1, 0, 0};
        gemm::launch_tile<64, 64, true, true, gemm::A_PLAIN, gemm::EpiStore, 4>(st, g, 1, e);
    }
    hipLaunchKernelGGL(nsf_magnitude_kernel, dim3((unsigned)((n_frames * ldm + 255) / 256)), dim3(256), 0, st, spec, n_frames, bins,
                       lds, mag, ldm);
    {
        gemm::Args g = gemm::make(mag, ldm, mel_basis, ldm, (int)n_frames, n_mels, ldm);
        gemm::EpiStore e{out, n_mels, nullptr, 1, 0, 0};
        gemm::launch_tile<64, 64, true, true, gemm::A_PLAIN, gemm::EpiStore, 4>(st, g, 1, e);
    }
    hipLaunchKernelGGL(nsf_logclamp_kernel, dim3((unsigned)((n_frames * n_mels + 255) / 256)), dim3(256), 0, st, out,
                       n_frames * n_mels, clip);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
