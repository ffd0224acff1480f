// a14-a15: caller-side glue of the real-time and offline paths, kept on the device so a stream never leaves HBM:
// SOLA splice (gui.py:405-430, windows gui.py:349-351) and the volume gate (main.py:111-116,159 / gui.py:108-112,127).
#include "common.h"

namespace {

// score[l] = sum_i x[l+i]*buf[i] / sqrt(sum_i x[l+i]^2 + 1e-8),  l = 0..search   (one workgroup per lag)
__global__ void __launch_bounds__(256) sola_score_kernel(const float* __restrict__ x, const float* __restrict__ buf,
                                                         int xfade, float* __restrict__ score) {
    const int l = blockIdx.x;
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < xfade; i += 256) {
        const float v = x[l + i];
        num = fmaf(v, buf[i], num);
        den = fmaf(v, v, den);
    }
    num = wave_sum(num);
    den = wave_sum(den);
    __shared__ float red[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[w] = num;
        red[4 + w] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = (red[0] + red[1]) + (red[2] + red[3]);
        const float d = (red[4] + red[5]) + (red[6] + red[7]);
        score[l] = n / sqrtf(d + 1e-8f);
    }
}

// argmax (first maximum, like torch.argmax), then cross-fade the head with the kept tail and emit the block
__global__ void __launch_bounds__(256) sola_splice_kernel(const float* __restrict__ x, const float* __restrict__ score,
                                                          int search, int block, int xfade,
                                                          float* __restrict__ sola_buf, float* __restrict__ emitted,
                                                          int* __restrict__ shift_out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    float best = -3.0e38f;
    int idx = 0x7fffffff;
    for (int l = threadIdx.x; l <= search; l += 256) {
        const float s = score[l];
        if (s > best) {
            best = s;
            idx = l;
        }
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float ov = bv[threadIdx.x + o];
            const int oi = bi[threadIdx.x + o];
            if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) {
                bv[threadIdx.x] = ov;
                bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int shift = bi[0];
    if (threadIdx.x == 0) *shift_out = shift;
    const float* src = x + shift;
    // head: tmp[i] = tmp[i]*fade_in[i] + buf[i]*(1 - fade_in[i]),  fade_in = sin^2(pi/2 * i/xfade)
    const float step = 1.0f / (float)xfade;
    for (int i = threadIdx.x; i < block; i += 256) {
        float v = src[i];
        if (i < xfade) {
            const float sn = sinf(__fdiv_rn(__fmul_rn(3.14159274101257324f, __fmul_rn((float)i, step)), 2.0f));
            const float fin = sn * sn;
            v = fmaf(sola_buf[i], 1.0f - fin, v * fin);
        }
        emitted[i] = v;
    }
    __syncthreads();  // every read of the old buffer is done before it is overwritten
    for (int i = threadIdx.x; i < xfade; i += 256) sola_buf[i] = src[block + i];
}

// signal[b][t] *= upsample(dilate9(volume > thr))[t]
__global__ void __launch_bounds__(256) volume_gate_kernel(float* __restrict__ signal, const float* __restrict__ volume,
                                                          float thr, int64_t B, int Fr, int hop) {
    const int64_t total = B * Fr * (int64_t)hop;
    const float inv_hop = 1.0f / (float)hop;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / ((int64_t)Fr * hop);
        const int64_t t = i - b * (int64_t)Fr * hop;
        const int m = (int)(t / hop), j = (int)(t % hop);
        const float* v = volume + b * Fr;
        float g[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int mm = m + q;
            if (mm > Fr - 1) mm = Fr - 1;
            float mx = 0.f;
            for (int d = -4; d <= 4; ++d) {  // edge-replicated 9-frame max
                int f = mm + d;
                f = f < 0 ? 0 : (f > Fr - 1 ? Fr - 1 : f);
                mx = fmaxf(mx, v[f] > thr ? 1.0f : 0.0f);
            }
            g[q] = mx;
        }
        const float w1 = (float)j * inv_hop;
        signal[i] *= fmaf(1.0f - w1, g[0], __fmul_rn(w1, g[1]));
    }
}

// ---- phase-vocoder cross-fade (gui.py:14-31, optional `use_phase_vocoder` splice of gui.py:417-423) ----------------
// result[t] = a[t]*fo[t]^2 + b[t]*fi[t]^2 + (sum_f A_f cos(w_f * t/n + phi_f)) * fo[t] * fi[t] / n  with, per rfft bin f,
//   A_f = (|Fa_f| + |Fb_f|) * (2 inside the spectrum, 1 at DC and at the Nyquist bin of an even n),
//   phi_f = angle(Fa_f),  w_f = 2 pi f + wrap(angle(Fb_f) - angle(Fa_f)),  wrap(x) = x - 2 pi floor(x/2/pi + 0.5).
// The per-bin frequencies are not harmonic, so the synthesis is a dense n x (n/2+1) cosine sum, not an inverse FFT.
// All fp32 steps after the two DFTs follow the reference's operation order (the argument w*t + phi reaches 5.5e3 rad at
// n = 1764: its fp32 rounding, 2.4e-4 rad, is part of the reference's result).
// (1) one wavefront per bin: direct DFT of a and b with exact integer phase reduction
__global__ void __launch_bounds__(256) pv_spectrum_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                          float* __restrict__ amp, float* __restrict__ phia,
                                                          float* __restrict__ wf) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int F = n / 2 + 1;
    if (f >= F) return;
    double ra = 0, ia = 0, rb = 0, ib = 0;
    for (int t = lane; t < n; t += 64) {
        const int r = (int)(((int64_t)f * t) % n);
        double sn, cs;
        sincos(-6.283185307179586476925286766559 * (double)r / (double)n, &sn, &cs);
        const double xa = a[t], xb = b[t];
        ra += xa * cs;
        ia += xa * sn;
        rb += xb * cs;
        ib += xb * sn;
    }
    ra = wave_sum_d(ra);
    ia = wave_sum_d(ia);
    rb = wave_sum_d(rb);
    ib = wave_sum_d(ib);
    if (lane == 0) {
        const float fra = (float)ra, fia = (float)ia, frb = (float)rb, fib = (float)ib;   // complex64 spectra
        float ab = hypotf(fra, fia) + hypotf(frb, fib);
        const bool inner = (n % 2 == 0) ? (f >= 1 && f <= F - 2) : (f >= 1);
        if (inner) ab = __fmul_rn(ab, 2.0f);
        const float pa = atan2f(fia, fra), pb = atan2f(fib, frb);
        float d = __fsub_rn(pb, pa);
        // deltaphase - 2*pi*floor(deltaphase / 2 / pi + 0.5), every step rounded to fp32 like the tensor expression
        const float k = floorf(__fadd_rn(__fdiv_rn(__fdiv_rn(d, 2.0f), 3.14159274101257324f), 0.5f));
        d = __fsub_rn(d, __fmul_rn(6.28318548202514648f, k));
        amp[f] = ab;
        phia[f] = pa;
        wf[f] = __fadd_rn(__fmul_rn(6.28318548202514648f, (float)f), d);
    }
}

// (2) one wavefront per output sample: the cosine sum over the bins, then the three-term mix
__global__ void __launch_bounds__(256) pv_synth_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ fo, const float* __restrict__ fi, int n,
                                                       const float* __restrict__ amp, const float* __restrict__ phia,
                                                       const float* __restrict__ wf, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n) return;
    const int F = n / 2 + 1;
    const float tt = __fdiv_rn((float)t, (float)n);
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s += __fmul_rn(amp[f], cosf(__fadd_rn(__fmul_rn(wf[f], tt), phia[f])));
    s = wave_sum(s);
    if (lane == 0) {
        const float o = fo[t], i = fi[t];
        const float osc = __fdiv_rn(__fmul_rn(__fmul_rn(s, o), i), (float)n);
        out[t] = __fadd_rn(__fadd_rn(__fmul_rn(a[t], __fmul_rn(o, o)), __fmul_rn(b[t], __fmul_rn(i, i))), osc);
    }
}

}  // namespace

extern "C" int ddsp_phase_vocoder(ddsp_ctx* ctx, void* stream, const float* a, const float* b, const float* fade_out,
                                  const float* fade_in, int n, float* out) {
    DDSP_REQUIRE(ctx, ctx && a && b && fade_out && fade_in && out, "ddsp_phase_vocoder: null argument");
    DDSP_REQUIRE(ctx, n >= 2 && n <= (1 << 16), "ddsp_phase_vocoder: bad length");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int F = n / 2 + 1;
    int rc = ddsp_scratch_reserve_bytes(ctx, (size_t)3 * F * sizeof(float) + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float* spec = nullptr;
    if ((rc = ddsp_scratch_get(ctx, (size_t)3 * F * sizeof(float), (void**)&spec))) return rc;
    ddsp_prof_begin(ctx, st, PF_SOLA);
    hipLaunchKernelGGL(pv_spectrum_kernel, dim3((F + 3) / 4), dim3(256), 0, st, a, b, n, spec, spec + F, spec + 2 * F);
    hipLaunchKernelGGL(pv_synth_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a, b, fade_out, fade_in, n, spec, spec + F,
                       spec + 2 * F, out);
    ddsp_prof_end(ctx, st, 10.0 * n * (double)F, 4.0 * 5 * n);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_sola(ddsp_ctx* ctx, void* stream, const float* audio, int64_t n_audio, int block, int xfade,
                         int search, int delay, float* sola_buffer, float* emitted, int* shift) {
    DDSP_REQUIRE(ctx, ctx && audio && sola_buffer && emitted && shift, "ddsp_sola: null argument");
    DDSP_REQUIRE(ctx, block >= xfade && xfade >= 1 && search >= 0 && delay >= 1, "ddsp_sola: bad sizes");
    DDSP_REQUIRE(ctx, n_audio >= (int64_t)block + xfade + search + delay, "ddsp_sola: window shorter than block+xfade+search+delay");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    int rc = ddsp_scratch_reserve_bytes(ctx, (size_t)(search + 1) * sizeof(float) + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float* score = nullptr;
    if ((rc = ddsp_scratch_get(ctx, (size_t)(search + 1) * sizeof(float), (void**)&score))) return rc;
    // temp_wav = audio[-block-xfade-search-delay : -delay]
    const float* x = audio + (n_audio - block - xfade - search - delay);
    ddsp_prof_begin(ctx, st, PF_SOLA);
    hipLaunchKernelGGL(sola_score_kernel, dim3(search + 1), dim3(256), 0, st, x, sola_buffer, xfade, score);
    hipLaunchKernelGGL(sola_splice_kernel, dim3(1), dim3(256), 0, st, x, score, search, block, xfade, sola_buffer,
                       emitted, shift);
    ddsp_prof_end(ctx, st, 4.0 * (search + 1) * xfade, 4.0 * ((double)block * 2 + xfade * 3 + search));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_volume_gate(ddsp_ctx* ctx, void* stream, float* signal, const float* volume, float threshold,
                                int64_t B, int64_t Fr, int hop) {
    DDSP_REQUIRE(ctx, ctx && signal && volume, "ddsp_volume_gate: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && Fr >= 1 && hop >= 1 && (hop & (hop - 1)) == 0, "ddsp_volume_gate: bad shape");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int64_t total = B * Fr * hop;
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 4096) blocks = 4096;
    ddsp_prof_begin(ctx, st, PF_OTHER);
    hipLaunchKernelGGL(volume_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, st, signal, volume, threshold, B,
                       (int)Fr, hop);
    ddsp_prof_end(ctx, st, 0.0, 8.0 * total);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
