// a11: CombSubFast DSP stage - sqrt-Hann windowed frames, one 1024-point spectrum product per frame, OLA.
//
// Replaces ddsp/vocoder.py:462-490.  One workgroup (256 threads) owns one frame m of one utterance
// (m = 0..Fr, frame Fr reuses filter Fr-1):
//   z = win*(comb + j*noise)                      two real signals packed into one complex sequence
//   Z = FFT_1024(z)                                radix-4 Stockham, 5 passes through LDS, fp32
//   C_f = (Z_f + conj Z_{N-f})/2, W_f = (Z_f - conj Z_{N-f})/(2j)        un-pack the two real spectra
//   Y_f = C_f * exp(hm_f + j*pi*hp_f) + W_f * exp(nm_f)/128,  f = 0..512  (imag of Y_0, Y_512 ignored: C2R)
//   y = IFFT_1024(Hermitian extension of Y) * win  -> frame scratch (B, Fr+1, 1024)
// and a second pass adds the two half-overlapping frames per output sample.  The circular (not zero-padded)
// convolution is what the reference computes.  Bound: HBM for the control frames (3*513*4 B per 512 output
// samples = 12 B/sample) plus 8 B/sample of frame scratch; the FFT work is ~0.2 kFLOP/sample.
#include "common.h"

namespace {

constexpr int N = 1024, HOP = 512, NB = 513;
constexpr double kTwoPi = 6.283185307179586476925286766559;

struct c32 {
    float x, y;
};
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return {fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)}; }
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }

// tw[k] = exp(-2*pi*i*k/N), k < N (fp64-generated table).  DIR = -1 forward, +1 inverse (conjugate twiddles).
template <int DIR>
__device__ __forceinline__ void fft1024(c32* __restrict__ a, c32* __restrict__ bbuf, const c32* __restrict__ tw,
                                        int j) {
    c32* src = a;
    c32* dst = bbuf;
#pragma unroll
    for (int Ns = 1; Ns < N; Ns *= 4) {
        const int k = j & (Ns - 1);
        const int tstep = N / (4 * Ns);  // twiddle index scale: angle = -2*pi*k*r/(4*Ns)
        c32 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = src[j + r * (N / 4)];
            if (r > 0 && Ns > 1) {
                c32 w = tw[(k * r * tstep) & (N - 1)];
                if (DIR > 0) w.y = -w.y;
                v[r] = cmul(v[r], w);
            }
        }
        // radix-4 butterfly (forward: -j rotation; inverse: +j)
        const c32 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]);
        const c32 s1 = cadd(v[1], v[3]), d1 = csub(v[1], v[3]);
        const c32 jd1 = (DIR < 0) ? c32{d1.y, -d1.x} : c32{-d1.y, d1.x};
        const int j0 = ((j - k) << 2) + k;
        dst[j0] = cadd(s0, s1);
        dst[j0 + Ns] = cadd(d0, jd1);
        dst[j0 + 2 * Ns] = csub(s0, s1);
        dst[j0 + 3 * Ns] = csub(d0, jd1);
        __syncthreads();
        c32* t = src;
        src = dst;
        dst = t;
    }
    // 5 passes: result is in `bbuf` (odd number of swaps)
}

__global__ void twiddle_kernel(float* __restrict__ tab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const double ang = kTwoPi * (double)k / (double)N;
    tab[2 * k] = (float)cos(ang);
    tab[2 * k + 1] = (float)(-sin(ang));
    // sqrt(hann_periodic(N)) behind the twiddles
    tab[2 * N + k] = (float)sqrt(0.5 - 0.5 * cos(ang));
}

__device__ __forceinline__ float noise_u(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
    return u * 2.0f - 1.0f;
}

__global__ void __launch_bounds__(256) spectral_frame_kernel(const float* __restrict__ ctrl, int64_t ld,
                                                             const float* __restrict__ comb,
                                                             const float* __restrict__ noise, int excitation,
                                                             uint64_t seed, const float* __restrict__ tab, int Fr,
                                                             float* __restrict__ frames) {
    __shared__ c32 A[N];
    __shared__ c32 Bf[N];
    __shared__ c32 tw[N];
    __shared__ float win[N];
    const int m = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t T = (int64_t)Fr * HOP;
    for (int i = tid; i < N; i += 256) {
        tw[i] = {tab[2 * i], tab[2 * i + 1]};
        win[i] = tab[2 * N + i];
    }
    __syncthreads();
    // frame m covers original samples [HOP*(m-1), HOP*(m+1)), zero outside [0, T)
    for (int i = tid; i < N; i += 256) {
        const int64_t t = (int64_t)HOP * (m - 1) + i;
        float c = 0.f, z = 0.f;
        if (t >= 0 && t < T) {
            c = comb[(int64_t)b * T + t];
            if (excitation == DDSP_EXC_GENERATE)
                z = noise_u(seed, (uint64_t)b * T + t);
            else
                z = __fadd_rn(__fmul_rn(noise[(int64_t)b * T + t], 2.0f), -1.0f);
        }
        A[i] = {c * win[i], z * win[i]};
    }
    __syncthreads();
    fft1024<-1>(A, Bf, tw, tid);  // spectrum in Bf
    // un-pack, apply the frame's filters, write the Hermitian spectrum into A
    const int mi = m < Fr ? m : Fr - 1;
    const float* cr = ctrl + ((int64_t)b * Fr + mi) * ld;
    const float pi_f = 3.14159274101257324f;
    for (int f = tid; f <= N / 2; f += 256) {
        const c32 zf = Bf[f], zn = Bf[(N - f) & (N - 1)];
        const c32 C = {0.5f * (zf.x + zn.x), 0.5f * (zf.y - zn.y)};   // spectrum of the comb frame
        const c32 W = {0.5f * (zf.y + zn.y), -0.5f * (zf.x - zn.x)};  // spectrum of the noise frame
        const float mag = expf(cr[f]);
        float sn, cs;
        sincosf(__fmul_rn(pi_f, cr[NB + f]), &sn, &cs);
        const c32 Hh = {mag * cs, mag * sn};
        const float g = __fdiv_rn(expf(cr[2 * NB + f]), 128.0f);
        c32 Y = cmul(C, Hh);
        Y.x = fmaf(W.x, g, Y.x);
        Y.y = fmaf(W.y, g, Y.y);
        if (f == 0 || f == N / 2) Y.y = 0.f;  // C2R ignores these imaginary parts
        A[f] = Y;
        if (f > 0 && f < N / 2) A[N - f] = {Y.x, -Y.y};
    }
    __syncthreads();
    fft1024<1>(A, Bf, tw, tid);
    float* dst = frames + ((int64_t)b * (Fr + 1) + m) * N;
    for (int i = tid; i < N; i += 256) dst[i] = Bf[i].x * (1.0f / N) * win[i];
}

__global__ void __launch_bounds__(256) overlap_add_kernel(const float* __restrict__ frames, int Fr, int64_t total,
                                                          float* __restrict__ out) {
    // out[b][HOP*s + j] = frames[b][s][HOP + j] + frames[b][s+1][j]   (drops HOP samples at each end)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total / 4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t4 = i * 4;
        const int64_t b = t4 / ((int64_t)Fr * HOP);
        const int64_t t = t4 - b * (int64_t)Fr * HOP;
        const int s = (int)(t / HOP), j = (int)(t % HOP);
        const float* f0 = frames + ((int64_t)b * (Fr + 1) + s) * N;
        const f32x4 u = *(const f32x4*)(f0 + HOP + j);
        const f32x4 v = *(const f32x4*)(f0 + N + j);
        *(f32x4*)(out + t4) = u + v;
    }
}


// Backward w.r.t. the three control blocks.  With g = win * d_out (the frame's slice of the upstream gradient) and
// its forward DFT G, the gradient on the one-sided spectrum is dY_f = c_f/N * G_f (c = 1 at DC/Nyquist, else 2;
// imaginary parts of those two bins carry none), and with Y = C*H + W*g_n, H = exp(hm + j*pi*hp), g_n = exp(nm)/128:
//   dH = conj(C) dY,  d_hm = Re(conj(H) dH),  d_hp = pi * Im(conj(H) dH),  d_nm = g_n * Re(conj(W) dY).
// One workgroup per filter frame m < Fr; frame Fr shares filter Fr-1 and is folded into that workgroup.
__global__ void __launch_bounds__(256) spectral_frame_bwd_kernel(const float* __restrict__ ctrl, int64_t ld,
                                                                 const float* __restrict__ comb,
                                                                 const float* __restrict__ noise, int excitation,
                                                                 uint64_t seed, const float* __restrict__ dout,
                                                                 const float* __restrict__ tab, int Fr,
                                                                 float* __restrict__ d_ctrl, int64_t ldo) {
    __shared__ c32 A[N];
    __shared__ c32 Bf[N];
    __shared__ c32 Gs[N];
    __shared__ c32 tw[N];
    __shared__ float win[N];
    const int m0 = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t T = (int64_t)Fr * HOP;
    for (int i = tid; i < N; i += 256) {
        tw[i] = {tab[2 * i], tab[2 * i + 1]};
        win[i] = tab[2 * N + i];
    }
    const float* cr = ctrl + ((int64_t)b * Fr + m0) * ld;
    float* dr = d_ctrl + ((int64_t)b * Fr + m0) * ldo;
    const float pi_f = 3.14159274101257324f;
    float acc[3][3];  // [bin slot of this thread][hm, hp, nm]; bins f = tid, tid+256, 512 (tid == 0)
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 0.f;
    const int n_pass = (m0 == Fr - 1) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int m = m0 + pass;
        __syncthreads();
        for (int i = tid; i < N; i += 256) {
            const int64_t t = (int64_t)HOP * (m - 1) + i;
            float c = 0.f, z = 0.f, g = 0.f;
            if (t >= 0 && t < T) {
                c = comb[(int64_t)b * T + t];
                if (excitation == DDSP_EXC_GENERATE)
                    z = noise_u(seed, (uint64_t)b * T + t);
                else
                    z = __fadd_rn(__fmul_rn(noise[(int64_t)b * T + t], 2.0f), -1.0f);
                g = dout[(int64_t)b * T + t];
            }
            A[i] = {c * win[i], z * win[i]};
            Gs[i] = {g * win[i], 0.f};
        }
        __syncthreads();
        fft1024<-1>(A, Bf, tw, tid);      // packed spectra of comb / noise frames in Bf
        // the second transform needs two scratch buffers: A is free again, result lands in A (odd number of passes
        // starting from Gs)
        fft1024<-1>(Gs, A, tw, tid);      // G in A
#pragma unroll
        for (int slot = 0; slot < 3; ++slot) {
            const int f = slot < 2 ? tid + 256 * slot : N / 2;
            if (slot == 2 && tid != 0) continue;
            const c32 zf = Bf[f], zn = Bf[(N - f) & (N - 1)];
            const c32 C = {0.5f * (zf.x + zn.x), 0.5f * (zf.y - zn.y)};
            const c32 W = {0.5f * (zf.y + zn.y), -0.5f * (zf.x - zn.x)};
            const float cf = (f == 0 || f == N / 2) ? 1.0f : 2.0f;
            c32 dY = {cf * (1.0f / N) * A[f].x, cf * (1.0f / N) * A[f].y};
            if (f == 0 || f == N / 2) dY.y = 0.f;
            const float mag = expf(cr[f]);
            float sn, cs;
            sincosf(__fmul_rn(pi_f, cr[NB + f]), &sn, &cs);
            const c32 Hh = {mag * cs, mag * sn};
            const c32 dH = {C.x * dY.x + C.y * dY.y, C.x * dY.y - C.y * dY.x};          // conj(C) * dY
            const c32 hd = {Hh.x * dH.x + Hh.y * dH.y, Hh.x * dH.y - Hh.y * dH.x};      // conj(H) * dH
            const float gn = __fdiv_rn(expf(cr[2 * NB + f]), 128.0f);
            acc[slot][0] += hd.x;
            acc[slot][1] += pi_f * hd.y;
            acc[slot][2] += gn * (W.x * dY.x + W.y * dY.y);                              // Re(conj(W) dY)
        }
    }
#pragma unroll
    for (int slot = 0; slot < 3; ++slot) {
        const int f = slot < 2 ? tid + 256 * slot : N / 2;
        if (slot == 2 && tid != 0) continue;
        dr[f] = acc[slot][0];
        dr[NB + f] = acc[slot][1];
        dr[2 * NB + f] = acc[slot][2];
    }
}

}  // namespace

extern "C" int ddsp_spectral_ola(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, const float* comb,
                                 const float* noise, int excitation, uint64_t noise_seed, int64_t B, int64_t Fr,
                                 int hop, float* out) {
    DDSP_REQUIRE(ctx, ctx && ctrl && comb && out, "ddsp_spectral_ola: null argument");
    DDSP_REQUIRE(ctx, hop == HOP, "ddsp_spectral_ola: only hop == 512 is built");
    DDSP_REQUIRE(ctx, ctrl_ld >= 3 * NB && B >= 0 && B <= 65535 && Fr >= 1, "ddsp_spectral_ola: bad shape");
    DDSP_REQUIRE(ctx, excitation == DDSP_EXC_GENERATE || (excitation == DDSP_EXC_UNIT_NOISE && noise),
                 "ddsp_spectral_ola: excitation must be UNIT_NOISE (with a noise buffer) or GENERATE");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const size_t fbytes = (size_t)B * (Fr + 1) * N * sizeof(float);
    int rc = ddsp_scratch_reserve_bytes(ctx, fbytes + 3 * N * sizeof(float) + 8192);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float *frames = nullptr, *tab = nullptr;
    if ((rc = ddsp_scratch_get(ctx, fbytes, (void**)&frames))) return rc;
    if ((rc = ddsp_scratch_get(ctx, 3 * N * sizeof(float), (void**)&tab))) return rc;
    ddsp_prof_begin(ctx, st, PF_SPECTRAL_OLA);
    hipLaunchKernelGGL(twiddle_kernel, dim3(N / 256), dim3(256), 0, st, tab);
    hipLaunchKernelGGL(spectral_frame_kernel, dim3((unsigned)(Fr + 1), (unsigned)B), dim3(256), 0, st, ctrl, ctrl_ld,
                       comb, noise, excitation, noise_seed, tab, (int)Fr, frames);
    const int64_t total = B * Fr * HOP;
    int64_t blocks = ceil_div64(total / 4, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, frames, (int)Fr, total, out);
    ddsp_prof_end(ctx, st, 2.0 * B * (Fr + 1) * 5.0 * N * 10.0 * 2.0,
                  4.0 * B * Fr * (3.0 * NB + 2.0 * HOP + 1.0 * HOP) + 2.0 * fbytes);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_spectral_ola_bwd(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, const float* comb,
                                     const float* noise, int excitation, uint64_t noise_seed, const float* d_out,
                                     int64_t B, int64_t Fr, int hop, float* d_ctrl, int64_t d_ctrl_ld) {
    DDSP_REQUIRE(ctx, ctx && ctrl && comb && d_out && d_ctrl, "ddsp_spectral_ola_bwd: null argument");
    DDSP_REQUIRE(ctx, hop == HOP, "ddsp_spectral_ola_bwd: only hop == 512 is built");
    DDSP_REQUIRE(ctx, ctrl_ld >= 3 * NB && d_ctrl_ld >= 3 * NB && B >= 0 && B <= 65535 && Fr >= 1, "ddsp_spectral_ola_bwd: bad shape");
    DDSP_REQUIRE(ctx, excitation == DDSP_EXC_GENERATE || (excitation == DDSP_EXC_UNIT_NOISE && noise),
                 "ddsp_spectral_ola_bwd: excitation must be UNIT_NOISE (with a noise buffer) or GENERATE");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    int rc = ddsp_scratch_reserve_bytes(ctx, 3 * N * sizeof(float) + 8192);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float* tab = nullptr;
    if ((rc = ddsp_scratch_get(ctx, 3 * N * sizeof(float), (void**)&tab))) return rc;
    ddsp_prof_begin(ctx, st, PF_SPECTRAL_OLA);
    hipLaunchKernelGGL(twiddle_kernel, dim3(N / 256), dim3(256), 0, st, tab);
    hipLaunchKernelGGL(spectral_frame_bwd_kernel, dim3((unsigned)Fr, (unsigned)B), dim3(256), 0, st, ctrl, ctrl_ld, comb,
                       noise, excitation, noise_seed, d_out, tab, (int)Fr, d_ctrl, d_ctrl_ld);
    ddsp_prof_end(ctx, st, 2.0 * B * (Fr + 1) * 5.0 * N * 10.0 * 2.0, 4.0 * B * Fr * (6.0 * NB + 3.0 * HOP));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
