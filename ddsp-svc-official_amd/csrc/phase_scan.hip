// a1-a3: frame->sample upsampler, fp64 wavefront-prefix-summed phase integrator, combtooth.
//
// Replaces (reference paths): ddsp/core.py:7-21 upsample, ddsp/core.py:31-51 fo_to_rot,
// ddsp/vocoder.py:517 phase_frames, ddsp/vocoder.py:539 / :459-460 combtooth.
//
// Layout: f0 frames (B,Fr) are the only HBM input (4 B per 512 output samples); every per-sample
// quantity is recomputed from the two bracketing frame values in registers.  One wavefront owns one
// frame (hop samples, 8 consecutive per lane for hop=512): pass 1 reduces the frame's fp64 increment
// sum, pass 2 adds the exclusive prefix of the preceding frame sums of the same utterance and does
// the in-frame scan with a 64-lane fp64 shuffle scan.  Outputs are written 16 B per lane.
#include "common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

// ATen's align_corners linear kernel: src = scale*dst, w1 = src - floor(src), w0 = 1 - w1,
// out = fma(w0, x0, w1*x1) (bit-exact with the CPU path on this image, tests/test_oracle_dsp.py).
__device__ __forceinline__ float lerp_frame(float x0, float x1, float w1) {
    return __fmaf_rn(1.0f - w1, x0, __fmul_rn(w1, x1));
}

__device__ __forceinline__ double wave_excl_scan_d(double v, int lane) {
    double incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    return incl - v;
}

template <bool PRECISE>
__device__ __forceinline__ double increment(float f, float srf, double srd) {
    // precise mode: the reference divides in fp64 (f0.double() / sr); here the fp64 product with 1/sr, which differs from
    // the quotient by at most one fp64 ulp per sample (~1e-11 cycles over a 2 s clip against the 1e-6 the result is
    // compared at after the wrap) and costs a multiply instead of a division - the division was this kernel's time
    // (VALU 1.15 in the round-1 counters).  `srd` carries 1/sr in this mode.
    if (PRECISE) return (double)f * srd;
    return (double)__fdiv_rn(f, srf);
}

// pass 1: frame_sum[b][m] = sum_j inc(b, m*hop + j)
template <bool PRECISE>
__global__ void __launch_bounds__(256) frame_sum_kernel(const float* __restrict__ f0_frames, int64_t n_frames_total,
                                                        int Fr, int hop, float scale, int sr,
                                                        double* __restrict__ frame_sum) {
    const int lane = threadIdx.x & 63;
    const int64_t fidx = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (fidx >= n_frames_total) return;
    const int m = (int)(fidx % Fr);
    const float x0 = f0_frames[fidx];
    const float x1 = (m + 1 < Fr) ? f0_frames[fidx + 1] : x0;
    const float srf = (float)sr;
    const double srd = PRECISE ? 1.0 / (double)sr : (double)sr;
    double acc = 0.0;
    const int per_lane = (hop + 63) / 64;
    for (int i = 0; i < per_lane; ++i) {
        int j = lane * per_lane + i;
        if (j < hop) {
            float src = __fmul_rn(scale, (float)(m * hop + j));
            float w1 = src - (float)m;
            acc += increment<PRECISE>(lerp_frame(x0, x1, w1), srf, srd);
        }
    }
    acc = wave_sum_d(acc);
    if (lane == 0) frame_sum[fidx] = acc;
}

struct ScanOut {
    float* rot;
    float* phase;
    float* comb;
    float* f0_up;
    float* phase_frames;
};

// Any hop up to 1024 (a lane owns ceil(hop / 64) consecutive samples, dword stores).  hop == 512 with 16-byte aligned outputs
// - every shipped configuration - runs frame_scan512_kernel below: the same arithmetic per sample, 32 contiguous bytes per lane
// and output array leaving as two 16-byte stores.
template <bool PRECISE>
__global__ void __launch_bounds__(256) frame_scan_kernel(const float* __restrict__ f0_frames,
                                                         const double* __restrict__ frame_sum,
                                                         const float* __restrict__ initial_phase,
                                                         int64_t n_frames_total, int Fr, int hop, float scale, int sr,
                                                         int comb_mode, ScanOut out) {
    const int lane = threadIdx.x & 63;
    const int64_t fidx = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (fidx >= n_frames_total) return;
    const int64_t b = fidx / Fr;
    const int m = (int)(fidx % Fr);
    const float x0 = f0_frames[fidx];
    const float x1 = (m + 1 < Fr) ? f0_frames[fidx + 1] : x0;
    const float srf = (float)sr;
    const double srd = PRECISE ? 1.0 / (double)sr : (double)sr;

    // exclusive prefix over the preceding frames of this utterance
    double before = 0.0;
    for (int i = lane; i < m; i += 64) before += frame_sum[b * Fr + i];
    before = wave_sum_d(before);

    constexpr int NL = 16;
    const int per_lane = (hop + 63) / 64;  // host guarantees per_lane <= 16
    float fval[NL];
    double local[NL];
    double run = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if (i < per_lane) {
            int j = lane * per_lane + i;
            float f = 0.f;
            double inc = 0.0;
            if (j < hop) {
                float src = __fmul_rn(scale, (float)(m * hop + j));
                float w1 = src - (float)m;
                f = lerp_frame(x0, x1, w1);
                inc = increment<PRECISE>(f, srf, srd);
            }
            run += inc;
            fval[i] = f;
            local[i] = run;
        }
    }
    const double base = before + wave_excl_scan_d(run, lane);

    double init_d = 0.0;
    float init_f = 0.f;
    const bool has_init = initial_phase != nullptr;
    if (has_init) {
        float ip = initial_phase[b];
        init_d = (double)ip / 2.0 / 3.141592653589793;              // .to(fp64)/2/np.pi
        init_f = __fdiv_rn(__fdiv_rn(ip, 2.0f), 3.14159274101257324f);  // fp32 /2 /np.pi
    }
    const float two_pi_f = 6.28318548202514648f;  // fp32(2*np.pi)
    const float pi_f = 3.14159274101257324f;
    const int64_t t0 = fidx * (int64_t)hop;

#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if (i < per_lane) {
            int j = lane * per_lane + i;
            if (j < hop) {
                double S = base + local[i];
                float r;
                if (PRECISE) {
                    if (has_init) S += init_d;
                    r = (float)(S - rint(S));
                } else {
                    float Sf = (float)S;  // ATen CPU cumsum: fp64 accumulator, fp32 output per element
                    if (has_init) Sf = __fadd_rn(Sf, init_f);
                    r = Sf - rintf(Sf);
                }
                const int64_t t = t0 + j;
                const float ph = __fmul_rn(two_pi_f, r);
                if (j == 0) out.phase_frames[fidx] = ph;
                float c = 0.f;
                if (out.comb) {
                    // x = sr * rot / (f0 + 1e-3), comb = sin(pi x) / (pi x) (vocoder.py:539).  Round 3: the two IEEE
                    // divisions and the full-range sinf were this kernel's time (VALU 1.17 in r02_e_pmc_sq_synth.txt:
                    // ~45 of ~75 instructions per sample).  Now: quotient by v_rcp_f32 + one Newton step (<= 1 ulp);
                    // x reduced EXACTLY to [-1, 1] (x - 2 rint(x / 2)) and folded to t in [-0.5, 0.5] with
                    // sin(pi x) = sin(pi t); sin(pi t) by its odd polynomial to t^13 (|error| < 1e-8).  The reference takes
                    // the sine of the ROUNDED product fl32(pi * x) - |x| reaches 340 - so the two agree to ~1e-7 in comb
                    // (both are ~1 ulp evaluations of a value of magnitude <= 1 / (pi x)); tests hold 2e-5.
                    const float den = __fadd_rn(fval[i], 1e-3f);
                    const float num = __fmul_rn(srf, r);
                    const float rc = __builtin_amdgcn_rcpf(den);
                    float x = num * rc;
                    x = fmaf(fmaf(-den, x, num), rc, x);
                    const float xr = fmaf(-2.0f, rintf(0.5f * x), x);
                    const float t = fabsf(xr) > 0.5f ? copysignf(1.0f, xr) - xr : xr;
                    const float t2 = t * t;
                    float pl = 0.00046630281f;                     // pi^13 / 13!, -pi^11 / 11!, ..., pi (Taylor; next term 2e-9)
                    pl = fmaf(pl, t2, -0.0073704309f);
                    pl = fmaf(pl, t2, 0.082145887f);
                    pl = fmaf(pl, t2, -0.59926453f);
                    pl = fmaf(pl, t2, 2.5501640f);
                    pl = fmaf(pl, t2, -5.1677128f);
                    pl = fmaf(pl, t2, 3.1415927f);
                    const float sn = t * pl;                        // sin(pi x)
                    const float px = pi_f * x;
                    c = (x == 0.0f) ? 1.0f : sn * __builtin_amdgcn_rcpf(px);
                    if (comb_mode == DDSP_COMB_SINC_GATED && fval[i] <= 0.0f) c = 0.0f;
                }
                if (out.rot) out.rot[t] = r;
                if (out.phase) out.phase[t] = ph;
                if (out.f0_up) out.f0_up[t] = fval[i];
                if (out.comb) out.comb[t] = c;
            }
        }
    }
}

// ---- hop == 512 (every shipped configuration): the two passes without the generic path's per-sample bounds tests --------------
// scale = Fr / (Fr * 512) is exactly 2^-9 in fp32, so src = scale * t is exact and w1 = (t mod 512) / 512 = lane / 64 + i / 512
// needs no conversion per sample.  The in-frame prefix is a DPP scan (row shifts and row broadcasts of the two halves of the
// fp64 value) instead of six LDS-crossbar shuffles; the preceding frames' sums are reduced the same way.  The sums themselves
// are unchanged: the same fp32 frame values, the same fp64 increments, added in fp64.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch_d(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ double wave_incl_scan_dpp(double v) {
    v += dpp_fetch_d<0x111, 0xf>(v);   // row_shr:1
    v += dpp_fetch_d<0x112, 0xf>(v);   // row_shr:2
    v += dpp_fetch_d<0x114, 0xf>(v);   // row_shr:4
    v += dpp_fetch_d<0x118, 0xf>(v);   // row_shr:8
    v += dpp_fetch_d<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_fetch_d<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double wave_total_of_scan(double incl) {
    const uint64_t u = __builtin_bit_cast(uint64_t, incl);
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)u, 63), hi = __builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <bool PRECISE>
__global__ void __launch_bounds__(256) frame_sum512_kernel(const float* __restrict__ f0_frames, int64_t n_frames_total, int Fr,
                                                           int sr, double* __restrict__ frame_sum) {
    const int lane = threadIdx.x & 63;
    const uint32_t fidx = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);   // (the host holds B * Fr below 2^31)
    if (fidx >= n_frames_total) return;
    const int m = (int)(fidx % (uint32_t)Fr);
    const float x0 = f0_frames[fidx];
    const float x1 = (m + 1 < Fr) ? f0_frames[fidx + 1] : x0;
    const float srf = (float)sr;
    const double srd = PRECISE ? 1.0 / (double)sr : (double)sr;
    const float wb = (float)lane * 0.015625f;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += increment<PRECISE>(lerp_frame(x0, x1, wb + (float)i * 0.001953125f), srf, srd);
    acc = wave_total_of_scan(wave_incl_scan_dpp(acc));
    if (lane == 0) frame_sum[fidx] = acc;
}

template <bool PRECISE>
__global__ void __launch_bounds__(256) frame_scan512_kernel(const float* __restrict__ f0_frames, const double* __restrict__ frame_sum,
                                                            const float* __restrict__ initial_phase, int64_t n_frames_total,
                                                            int Fr, int sr, int comb_mode, ScanOut out) {
    const int lane = threadIdx.x & 63;
    const uint32_t fidx = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (fidx >= n_frames_total) return;
    const uint32_t b = fidx / (uint32_t)Fr;
    const int m = (int)(fidx - b * (uint32_t)Fr);
    const float x0 = f0_frames[fidx];
    const float x1 = (m + 1 < Fr) ? f0_frames[fidx + 1] : x0;
    const float srf = (float)sr;
    const double srd = PRECISE ? 1.0 / (double)sr : (double)sr;

    double before = 0.0;                                  // the preceding frames of this utterance
    for (int i = lane; i < m; i += 64) before += frame_sum[(int64_t)b * Fr + i];
    before = wave_total_of_scan(wave_incl_scan_dpp(before));

    const float wb = (float)lane * 0.015625f;
    float fval[8];
    double local[8];
    double run = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fval[i] = lerp_frame(x0, x1, wb + (float)i * 0.001953125f);
        run += increment<PRECISE>(fval[i], srf, srd);
        local[i] = run;
    }
    double base = before + (wave_incl_scan_dpp(run) - run);

    float init_f = 0.f;
    const bool has_init = initial_phase != nullptr;
    if (has_init) {
        const float ip = initial_phase[b];
        if (PRECISE) base += (double)ip / 2.0 / 3.141592653589793;             // .to(fp64)/2/np.pi
        init_f = __fdiv_rn(__fdiv_rn(ip, 2.0f), 3.14159274101257324f);        // fp32 /2 /np.pi
    }
    const float two_pi_f = 6.28318548202514648f, pi_f = 3.14159274101257324f;
    const bool gated = comb_mode == DDSP_COMB_SINC_GATED;
    float r_v[8], c_v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double S = base + local[i];
        float r;
        if (PRECISE) {
            r = (float)(S - rint(S));
        } else {
            float Sf = (float)S;                           // ATen CPU cumsum: fp64 accumulator, fp32 output per element
            if (has_init) Sf = __fadd_rn(Sf, init_f);
            r = Sf - rintf(Sf);
        }
        r_v[i] = r;
        // comb = sin(pi x) / (pi x), x = sr * rot / (f0 + 1e-3) (vocoder.py:539): see frame_scan_kernel for the evaluation
        const float den = __fadd_rn(fval[i], 1e-3f);
        const float num = __fmul_rn(srf, r);
        const float rc = __builtin_amdgcn_rcpf(den);
        float x = num * rc;
        x = fmaf(fmaf(-den, x, num), rc, x);
        const float xr = fmaf(-2.0f, rintf(0.5f * x), x);
        const float t = fabsf(xr) > 0.5f ? copysignf(1.0f, xr) - xr : xr;
        const float t2 = t * t;
        float pl = 0.00046630281f;
        pl = fmaf(pl, t2, -0.0073704309f);
        pl = fmaf(pl, t2, 0.082145887f);
        pl = fmaf(pl, t2, -0.59926453f);
        pl = fmaf(pl, t2, 2.5501640f);
        pl = fmaf(pl, t2, -5.1677128f);
        pl = fmaf(pl, t2, 3.1415927f);
        float c = (x == 0.0f) ? 1.0f : (t * pl) * __builtin_amdgcn_rcpf(pi_f * x);
        if (gated && fval[i] <= 0.0f) c = 0.0f;
        c_v[i] = c;
    }
    if (lane == 0) out.phase_frames[fidx] = __fmul_rn(two_pi_f, r_v[0]);
    const int64_t t0 = (int64_t)fidx * 512 + lane * 8;
    auto put = [&](float* dst, const float (&v)[8]) {
        *(f32x4*)(dst + t0) = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(dst + t0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
    };
    if (out.comb) put(out.comb, c_v);
    if (out.rot) put(out.rot, r_v);
    if (out.f0_up) put(out.f0_up, fval);
    if (out.phase) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r_v[i] = __fmul_rn(two_pi_f, r_v[i]);
        put(out.phase, r_v);
    }
}

__global__ void __launch_bounds__(256) upsample_kernel(const float* __restrict__ x, int64_t B, int64_t Fr, int64_t C,
                                                       int hop, float scale, float* __restrict__ out) {
    // one thread per output element, channels fastest (coalesced on C, or on t when C == 1)
    const int64_t total = B * Fr * hop * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx % C;
        const int64_t t = (idx / C) % (Fr * hop);
        const int64_t b = idx / (C * Fr * hop);
        const float src = __fmul_rn(scale, (float)t);
        const int64_t i0 = (int64_t)src;
        const int64_t i1 = (i0 + 1 < Fr) ? i0 + 1 : Fr - 1;
        const float w1 = src - (float)i0;
        const float* row = x + b * Fr * C;
        out[idx] = lerp_frame(row[i0 * C + c], row[i1 * C + c], w1);
    }
}

}  // namespace

extern "C" int ddsp_upsample(ddsp_ctx* ctx, void* stream, const float* x, int64_t B, int64_t Fr, int64_t C, int hop,
                             float* out) {
    DDSP_REQUIRE(ctx, ctx && x && out, "ddsp_upsample: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && Fr >= 1 && C >= 1 && hop >= 1, "ddsp_upsample: bad shape");
    DDSP_REQUIRE(ctx, Fr * (int64_t)hop < (1 << 24), "ddsp_upsample: Fr*hop must stay below 2^24 (fp32 index grid)");
    if (B == 0) return DDSP_OK;
    DDSP_ENTER_DEVICE(ctx);
    const float scale = (float)Fr / (float)(Fr * hop);
    const int64_t total = B * Fr * hop * C;
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, B, Fr, C, hop,
                       scale, out);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_phase_scan(ddsp_ctx* ctx, void* stream, const float* f0_frames, const float* initial_phase,
                               int64_t B, int64_t Fr, int hop, int sr, int precise, int comb_mode, float* rot,
                               float* phase, float* comb, float* f0_up, float* phase_frames) {
    DDSP_REQUIRE(ctx, ctx && f0_frames && phase_frames, "ddsp_phase_scan: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && Fr >= 1 && hop >= 1 && hop <= 1024 && sr > 0, "ddsp_phase_scan: bad shape");
    DDSP_REQUIRE(ctx, Fr * (int64_t)hop < (1 << 24) && B * Fr < (int64_t)1 << 31, "ddsp_phase_scan: Fr*hop must stay below 2^24, B*Fr below 2^31");
    DDSP_REQUIRE(ctx, comb_mode >= 0 && comb_mode <= 2, "ddsp_phase_scan: unknown comb_mode");
    DDSP_REQUIRE(ctx, (comb_mode == DDSP_COMB_NONE) == (comb == nullptr), "ddsp_phase_scan: comb buffer vs comb_mode");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nf = B * Fr;
    DDSP_ENTER_DEVICE(ctx);
    int rc = ddsp_scratch_reserve_bytes(ctx, (size_t)nf * sizeof(double) + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    double* fsum = nullptr;
    rc = ddsp_scratch_get(ctx, (size_t)nf * sizeof(double), (void**)&fsum);
    if (rc) return rc;
    const float scale = (float)Fr / (float)(Fr * hop);
    const unsigned blocks = (unsigned)ceil_div64(nf, WAVES_PER_BLOCK);
    ScanOut o{rot, phase, comb, f0_up, phase_frames};
    ddsp_prof_begin(ctx, st, PF_PHASE_SCAN);
    const bool vec8 = hop == 512 && (((uintptr_t)rot | (uintptr_t)phase | (uintptr_t)comb | (uintptr_t)f0_up) % 16) == 0;
    if (vec8 && precise) {
        hipLaunchKernelGGL(frame_sum512_kernel<true>, dim3(blocks), dim3(256), 0, st, f0_frames, nf, (int)Fr, sr, fsum);
        hipLaunchKernelGGL(frame_scan512_kernel<true>, dim3(blocks), dim3(256), 0, st, f0_frames, fsum, initial_phase, nf, (int)Fr, sr,
                           comb_mode, o);
    } else if (vec8) {
        hipLaunchKernelGGL(frame_sum512_kernel<false>, dim3(blocks), dim3(256), 0, st, f0_frames, nf, (int)Fr, sr, fsum);
        hipLaunchKernelGGL(frame_scan512_kernel<false>, dim3(blocks), dim3(256), 0, st, f0_frames, fsum, initial_phase, nf, (int)Fr, sr,
                           comb_mode, o);
    } else if (precise) {
        hipLaunchKernelGGL(frame_sum_kernel<true>, dim3(blocks), dim3(256), 0, st, f0_frames, nf, (int)Fr, hop, scale, sr, fsum);
        hipLaunchKernelGGL(frame_scan_kernel<true>, dim3(blocks), dim3(256), 0, st, f0_frames, fsum, initial_phase, nf, (int)Fr, hop,
                           scale, sr, comb_mode, o);
    } else {
        hipLaunchKernelGGL(frame_sum_kernel<false>, dim3(blocks), dim3(256), 0, st, f0_frames, nf, (int)Fr, hop, scale, sr, fsum);
        hipLaunchKernelGGL(frame_scan_kernel<false>, dim3(blocks), dim3(256), 0, st, f0_frames, fsum, initial_phase, nf, (int)Fr, hop,
                           scale, sr, comb_mode, o);
    }
    {
        const double outs = (rot ? 1 : 0) + (phase ? 1 : 0) + (comb ? 1 : 0) + (f0_up ? 1 : 0);
        ddsp_prof_end(ctx, st, 0.0, 4.0 * nf * (2.0 + outs * hop));
    }
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
