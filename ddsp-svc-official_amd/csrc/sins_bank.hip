// a10: additive sinusoid bank  s[t] = sum_k lerp_t(A_k) * sin(k * phase[t]),  k = 1..H.
//
// Replaces ddsp/vocoder.py:397,402-412 (exp/128, remove_above_fmax, chunked upsample*sin sum) and
// ddsp/core.py:24-28.  The reference materialises (B,T,32) amplitude and phase tensors per chunk (721 MB
// each at B=64); here one workgroup renders one frame (hop samples): the two bracketing amplitude frames
// are activated once into LDS (2*H floats), every lane owns hop/256 samples, and the harmonics are walked
// with a complex rotation recurrence z_k = z_{k-1} * z_1 that is re-seeded from an accurate sincosf every
// 32 harmonics, so the cost per (sample, harmonic) is 7 VALU ops instead of a full sinf.
// Bound: fp32 vector ALU (H*7 FLOP per sample vs 4*H/hop + 8 B per sample of HBM traffic).
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int RESEED = 32;

__global__ void __launch_bounds__(256) sins_bank_kernel(const float* __restrict__ ctrl, int64_t ld, int H,
                                                        const float* __restrict__ f0_frames,
                                                        const float* __restrict__ phase, int Fr, int hop, float fmax,
                                                        float* __restrict__ out) {
    extern __shared__ float amp[];  // [2][H]: activated amplitudes of frame m and m+1 (clamped)
    const int m = blockIdx.x, b = blockIdx.y;
    const int64_t row = (int64_t)b * Fr + m;
    const int m1 = (m + 1 < Fr) ? 1 : 0;
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) {
        const int which = i / H, k = i % H;
        const int64_t r = row + (which ? m1 : 0);
        const float a = __fdiv_rn(expf(ctrl[r * ld + k]), 128.0f);
        // remove_above_fmax: (k*f0 < fmax) + 1e-7, all in fp32
        const float pk = __fmul_rn(f0_frames[r], (float)(k + 1));
        const float aa = __fadd_rn(pk < fmax ? 1.0f : 0.0f, 1e-7f);
        amp[i] = __fmul_rn(a, aa);
    }
    __syncthreads();
    const float* a0 = amp;
    const float* a1 = amp + H;
    const float inv_hop = 1.0f / (float)hop;  // hop is a power of two: exact
    if (hop == 2 * (int)blockDim.x && (H & 3) == 0) {
        // hop = 512: a thread owns samples j and j + 256 and walks the harmonics for both at once in packed fp32
        // (v_pk_fma_f32 / v_pk_mul_f32: two samples per instruction), with the window taken out of the harmonic sum:
        //   sum_k (w0 A0_k + w1 A1_k) sin_k = w0 sum_k A0_k sin_k + w1 sum_k A1_k sin_k,
        // so a harmonic costs 6 packed instructions for two samples (3 per sample instead of 7) and its two amplitudes are
        // read from the LDS once for both samples, four harmonics per 16-byte read.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const int j = threadIdx.x;
        const int64_t t = row * hop + j;
        const f32x2 ph = {phase[t], phase[t + blockDim.x]};
        float sa, ca, sb, cb;
        sincosf(ph[0], &sa, &ca);
        sincosf(ph[1], &sb, &cb);
        const f32x2 s1 = {sa, sb}, c1 = {ca, cb};
        f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
        for (int k0 = 0; k0 < H; k0 += RESEED) {
            // seed at harmonic k0+1 with the reference's own argument rounding: sin(fp32((k0+1) * ph))
            sincosf(__fmul_rn(ph[0], (float)(k0 + 1)), &sa, &ca);
            sincosf(__fmul_rn(ph[1], (float)(k0 + 1)), &sb, &cb);
            f32x2 zs = {sa, sb}, zc = {ca, cb};
            const int kend = (k0 + RESEED < H) ? k0 + RESEED : H;
            for (int k = k0; k < kend; k += 4) {
                const f32x4 A0 = *(const f32x4*)(a0 + k), A1 = *(const f32x4*)(a1 + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_elementwise_fma((f32x2){A0[e], A0[e]}, zs, acc0);
                    acc1 = __builtin_elementwise_fma((f32x2){A1[e], A1[e]}, zs, acc1);
                    const f32x2 nc = __builtin_elementwise_fma(zc, c1, -(zs * s1));
                    const f32x2 ns = __builtin_elementwise_fma(zs, c1, zc * s1);
                    zc = nc;
                    zs = ns;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float w1 = (float)(j + i * (int)blockDim.x) * inv_hop, w0 = 1.0f - w1;
            out[t + i * blockDim.x] = fmaf(w0, acc0[i], w1 * acc1[i]);
        }
        return;
    }
    for (int j = threadIdx.x; j < hop; j += blockDim.x) {
        const int64_t t = row * hop + j;
        const float ph = phase[t];
        const float w1 = (float)j * inv_hop, w0 = 1.0f - w1;
        float s1, c1;
        sincosf(ph, &s1, &c1);
        float acc = 0.f;
        for (int k0 = 0; k0 < H; k0 += RESEED) {
            // seed at harmonic k0+1 with the reference's own argument rounding: sin(fp32((k0+1) * ph))
            float zs, zc;
            sincosf(__fmul_rn(ph, (float)(k0 + 1)), &zs, &zc);
            const int kend = (k0 + RESEED < H) ? k0 + RESEED : H;
            for (int k = k0; k < kend; ++k) {
                const float a = fmaf(w0, a0[k], __fmul_rn(w1, a1[k]));  // same rounding as the upsampler
                acc = fmaf(a, zs, acc);
                const float nc = fmaf(zc, c1, -zs * s1);
                const float ns = fmaf(zs, c1, zc * s1);
                zc = nc;
                zs = ns;
            }
        }
        out[t] = acc;
    }
}


// Backward w.r.t. the amplitude controls.  out[t] = sum_k lerp_t(A_k) sin(k phase_t) and A = exp(ctrl)/128 * aa, so
//   d ctrl[m][k] = A[m][k] * ( sum_{t in seg m} (1 - j/hop) sin(k phase_t) d_t  +  sum_{t in seg m-1} (j/hop) sin(k phase_t) d_t )
// (frame Fr-1 also receives the j/hop share of its own segment: the upsampler repeats the last frame).
// One workgroup per (utterance, segment): every lane walks the harmonics of its samples with the same rotation
// recurrence as the forward kernel; per harmonic the two window-weighted sums are reduced across the workgroup
// (DPP wave reduction + LDS), and written as partials [seg][2][H] that a second pass combines.
__global__ void __launch_bounds__(256) sins_bank_bwd_partial_kernel(const float* __restrict__ phase,
                                                                    const float* __restrict__ dout, int H, int Fr, int hop,
                                                                    float* __restrict__ partial) {
    extern __shared__ float red[];  // [4 waves][2][H]
    const int m = blockIdx.x, b = blockIdx.y;
    const int64_t row = (int64_t)b * Fr + m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_hop = 1.0f / (float)hop;
    // this lane's samples (hop / 256 of them, at most 4)
    float ph[4], g0[4], g1[4], s1[4], c1[4];
    const int per = (hop + 255) / 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = threadIdx.x + 256 * i;
        ph[i] = 0.f;
        g0[i] = 0.f;
        g1[i] = 0.f;
        if (i < per && j < hop) {
            const int64_t t = row * hop + j;
            ph[i] = phase[t];
            const float d = dout[t], w1 = (float)j * inv_hop;
            g0[i] = (1.0f - w1) * d;
            g1[i] = w1 * d;
        }
        sincosf(ph[i], &s1[i], &c1[i]);
    }
    for (int k0 = 0; k0 < H; k0 += RESEED) {
        float zs[4], zc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sincosf(__fmul_rn(ph[i], (float)(k0 + 1)), &zs[i], &zc[i]);
        const int kend = (k0 + RESEED < H) ? k0 + RESEED : H;
        for (int k = k0; k < kend; ++k) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0 = fmaf(g0[i], zs[i], a0);
                a1 = fmaf(g1[i], zs[i], a1);
                const float nc = fmaf(zc[i], c1[i], -zs[i] * s1[i]);
                const float ns = fmaf(zs[i], c1[i], zc[i] * s1[i]);
                zc[i] = nc;
                zs[i] = ns;
            }
            a0 = wave_sum(a0);
            a1 = wave_sum(a1);
            if (lane == 0) {
                red[(wave * 2 + 0) * H + k] = a0;
                red[(wave * 2 + 1) * H + k] = a1;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * H; i += 256) {
        const int which = i / H, k = i % H;
        partial[(row * 2 + which) * H + k] = (red[(0 * 2 + which) * H + k] + red[(1 * 2 + which) * H + k]) +
                                             (red[(2 * 2 + which) * H + k] + red[(3 * 2 + which) * H + k]);
    }
}

// The same partial sums for hop <= 512 and H % 32 == 0 (round 3): the kernel above reduces every harmonic's two sums across the
// wave on its own (2 x 6 shuffle steps per harmonic: more instructions than the 12 of the products and the rotation).  Here a
// lane keeps the 2 x 32 partial sums of a re-seed block in registers and the wave reduces them TOGETHER by a transposing
// butterfly - at step m (32, 16, ..., 1) a lane sends the half of its values its partner keeps and adds the half it keeps, so 64
// values cost 63 exchanges instead of 64 x 6, and lane l ends with the wave's total of value l (l >> 5: window share, l & 31:
// harmonic of the block).  Same sums, different order of the additions.
__global__ void __launch_bounds__(256) sins_bank_bwd_partial32_kernel(const float* __restrict__ phase, const float* __restrict__ dout,
                                                                      int H, int Fr, int hop, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int m = blockIdx.x, b = blockIdx.y;
    const int64_t row = (int64_t)b * Fr + m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_hop = 1.0f / (float)hop;
    float ph[2], g0[2], g1[2], s1[2], c1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = threadIdx.x + 256 * i;
        ph[i] = 0.f;
        g0[i] = 0.f;
        g1[i] = 0.f;
        if (j < hop) {
            const int64_t t = row * hop + j;
            ph[i] = phase[t];
            const float d = dout[t], w1 = (float)j * inv_hop;
            g0[i] = (1.0f - w1) * d;
            g1[i] = w1 * d;
        }
        sincosf(ph[i], &s1[i], &c1[i]);
    }
    for (int k0 = 0; k0 < H; k0 += 32) {
        float zs[2], zc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) sincosf(__fmul_rn(ph[i], (float)(k0 + 1)), &zs[i], &zc[i]);
        float a[64];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a0 = fmaf(g0[i], zs[i], a0);
                a1 = fmaf(g1[i], zs[i], a1);
                const float nc = fmaf(zc[i], c1[i], -zs[i] * s1[i]);
                const float ns = fmaf(zs[i], c1[i], zc[i] * s1[i]);
                zc[i] = nc;
                zs[i] = ns;
            }
            a[kk] = a0;
            a[32 + kk] = a1;
        }
        // transposing butterfly over the 64 lanes
#pragma unroll
        for (int mbit = 32, c = 32; mbit >= 1; mbit >>= 1, c >>= 1) {
            const bool upper = (lane & mbit) != 0;
#pragma unroll
            for (int i = 0; i < c; ++i) {
                const float send = upper ? a[i] : a[i + c];
                const float keep = upper ? a[i + c] : a[i];
                a[i] = keep + __shfl_xor(send, mbit, 64);
            }
        }
        red[wave][lane] = a[0];
        __syncthreads();
        if (threadIdx.x < 64) {
            const int which = threadIdx.x >> 5, kk = threadIdx.x & 31;
            partial[(row * 2 + which) * H + k0 + kk] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) sins_bank_bwd_combine_kernel(const float* __restrict__ ctrl, int64_t ld, int H,
                                                                    const float* __restrict__ f0_frames,
                                                                    const float* __restrict__ partial, int64_t rows,
                                                                    int Fr, float fmax, float* __restrict__ d_ctrl,
                                                                    int64_t ldo) {
    const int64_t total = rows * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / H;
        const int k = (int)(i % H);
        const int m = (int)(r % Fr);
        float g = partial[(r * 2 + 0) * H + k];                     // (1 - j/hop) share of its own segment
        if (m > 0) g += partial[((r - 1) * 2 + 1) * H + k];         // j/hop share of the previous segment
        if (m == Fr - 1) g += partial[(r * 2 + 1) * H + k];         // last frame is its own successor
        const float a = __fdiv_rn(expf(ctrl[r * ld + k]), 128.0f);
        const float pk = __fmul_rn(f0_frames[r], (float)(k + 1));
        const float aa = __fadd_rn(pk < fmax ? 1.0f : 0.0f, 1e-7f);
        d_ctrl[r * ldo + k] = g * __fmul_rn(a, aa);
    }
}

}  // namespace

extern "C" int ddsp_sins_bank(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, int n_harmonics,
                              const float* f0_frames, const float* phase, int64_t B, int64_t Fr, int hop, int sr,
                              float* out) {
    DDSP_REQUIRE(ctx, ctx && ctrl && f0_frames && phase && out, "ddsp_sins_bank: null argument");
    DDSP_REQUIRE(ctx, n_harmonics >= 1 && n_harmonics <= 4096 && ctrl_ld >= n_harmonics, "ddsp_sins_bank: bad harmonics");
    DDSP_REQUIRE(ctx, B >= 0 && B <= 65535 && Fr >= 1 && hop >= 1 && (hop & (hop - 1)) == 0, "ddsp_sins_bank: bad shape (hop must be a power of two)");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    ddsp_prof_begin(ctx, st, PF_SINS_BANK);
    hipLaunchKernelGGL(sins_bank_kernel, dim3((unsigned)Fr, (unsigned)B), dim3(256), 2 * n_harmonics * sizeof(float), st,
                       ctrl, ctrl_ld, n_harmonics, f0_frames, phase, (int)Fr, hop, (float)sr / 2.0f, out);
    ddsp_prof_end(ctx, st, 7.0 * B * Fr * hop * (double)n_harmonics,
                  4.0 * B * Fr * ((double)n_harmonics + 1 + 2.0 * hop));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_sins_bank_bwd(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, int n_harmonics,
                                  const float* f0_frames, const float* phase, const float* d_out, int64_t B, int64_t Fr,
                                  int hop, int sr, float* d_ctrl, int64_t d_ctrl_ld) {
    DDSP_REQUIRE(ctx, ctx && ctrl && f0_frames && phase && d_out && d_ctrl, "ddsp_sins_bank_bwd: null argument");
    DDSP_REQUIRE(ctx, n_harmonics >= 1 && n_harmonics <= 4096 && ctrl_ld >= n_harmonics && d_ctrl_ld >= n_harmonics,
                 "ddsp_sins_bank_bwd: bad harmonics");
    DDSP_REQUIRE(ctx, B >= 0 && B <= 65535 && Fr >= 1 && hop >= 1 && hop <= 1024 && (hop & (hop - 1)) == 0,
                 "ddsp_sins_bank_bwd: bad shape (hop must be a power of two <= 1024)");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int64_t rows = B * Fr;
    const size_t pf = (size_t)rows * 2 * n_harmonics;
    int rc = ddsp_scratch_reserve_bytes(ctx, pf * sizeof(float) + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float* partial = nullptr;
    if ((rc = ddsp_scratch_get(ctx, pf * sizeof(float), (void**)&partial))) return rc;
    ddsp_prof_begin(ctx, st, PF_SINS_BANK);
    static int bwd32 = -1;   // DDSP_SINS_BWD32=0: the per-harmonic reduction at every shape (measurement aid)
    if (bwd32 < 0) {
        const char* e = getenv("DDSP_SINS_BWD32");
        bwd32 = (e && e[0] == '0') ? 0 : 1;
    }
    if (bwd32 && hop <= 512 && n_harmonics % 32 == 0)
        hipLaunchKernelGGL(sins_bank_bwd_partial32_kernel, dim3((unsigned)Fr, (unsigned)B), dim3(256), 0, st, phase, d_out,
                           n_harmonics, (int)Fr, hop, partial);
    else
        hipLaunchKernelGGL(sins_bank_bwd_partial_kernel, dim3((unsigned)Fr, (unsigned)B), dim3(256),
                           8 * n_harmonics * sizeof(float), st, phase, d_out, n_harmonics, (int)Fr, hop, partial);
    int64_t blocks = ceil_div64(rows * n_harmonics, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sins_bank_bwd_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ctrl, ctrl_ld, n_harmonics,
                       f0_frames, partial, rows, (int)Fr, (float)sr / 2.0f, d_ctrl, d_ctrl_ld);
    ddsp_prof_end(ctx, st, 10.0 * B * Fr * hop * (double)n_harmonics, 4.0 * B * Fr * (3.0 * n_harmonics + 2.0 * hop));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
