// a5-a6: control frames -> linear-phase FIR frames (ctrl activation + inverse real DFT + window).
//
// Replaces ddsp/vocoder.py:521-523 (pi*tanh / exp / exp/128), `exp(1j*cumsum(group_delay))` (:540) and
// ddsp/core.py:306-328 `_frequency_impulse_response` with its three window branches (:242-289, :292-303).
// n = 2*(n_mag-1) is 510 or 1022 (not FFT-friendly: 2*3*5*17, 2*7*73), and the transform is applied to
// 11k frames at once, so it is a dense contraction against a constant matrix: IR = act(ctrl) x T on the
// fp32 matrix pipe (gemm_f32.h), T from tables.hip with the n/2 rotation and the static Hann folded in.
// The dynamic raised-cosine window depends on the frame's f0 and is applied in the GEMM epilogue.
#include "gemm_f32.h"
#include "tables.h"

#include <stdlib.h>

namespace {

// One wavefront per frame row.  REAL modes: out[f] = exp(ctrl[f]) * scale.
// ALLPASS: gd = pi*tanh(ctrl); phi = cumsum_f(gd) (fp64 running sum rounded to fp32 per bin, as ATen's
// CPU cumsum does); out = [cos(phi) | sin(phi)].
// split != 0: the row is written as bf16 hi/lo groups of 8 (A operand of the split-bf16 inverse-DFT GEMM, gemm A_split);
// M % 8 == 0 then, and every lane of a group of 8 takes part in the exchange (values past M are zeros).
__global__ void __launch_bounds__(256) fir_act_kernel(int mode, const float* __restrict__ ctrl, int64_t ld, int M,
                                                      int ldo, int64_t rows, float* __restrict__ out, int split) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = ctrl + row * ld;
    if (mode != DDSP_FIR_ALLPASS) {
        float* dst = out + row * ldo;
        const float scale = (mode == DDSP_FIR_STATIC) ? (1.0f / 128.0f) : 1.0f;
        if (split) {
            for (int f0 = 0; f0 < M; f0 += 64) {
                const int f = f0 + lane;
                const float v = f < M ? expf(src[f]) * scale : 0.f;
                const uint32_t w = ddsp_split1_group8(v, lane);
                if (f < M) ((uint32_t*)dst)[f] = w;
            }
            return;
        }
        for (int f = lane; f < M; f += 64) dst[f] = expf(src[f]) * scale;  // /128 is exact scaling
        return;
    }
    float* dst = out + row * ldo;
    const float pi_f = 3.14159274101257324f;
    double carry = 0.0;
    for (int f0 = 0; f0 < M; f0 += 64) {
        const int f = f0 + lane;
        const float gd = (f < M) ? __fmul_rn(pi_f, tanhf(src[f])) : 0.f;
        double incl = (double)gd;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            double nb = __shfl_up(incl, o, 64);
            if (lane >= o) incl += nb;
        }
        const float phi = (float)(carry + incl);
        carry += __shfl(incl, 63, 64);
        // cos / sin of the fp32 phase in fp32, like the reference's complex64 `exp(1j * cumsum)` (the fp64 evaluation used
        // until round 2 was this kernel's time)
        float s = 0.f, c = 0.f;
        if (f < M) sincosf(phi, &s, &c);
        if (split) {
            const uint32_t wc = ddsp_split1_group8(c, lane), ws = ddsp_split1_group8(s, lane);
            if (f < M) {
                ((uint32_t*)dst)[f] = wc;
                ((uint32_t*)dst)[M + f] = ws;
            }
        } else if (f < M) {
            dst[f] = c;
            dst[M + f] = s;
        }
    }
}

// Real responses give filters that are even about tap n/2 (after the n/2 rotation), and the folded static Hann is even
// too, so the GEMM only produces taps 0..n/2 and the epilogue mirrors them: tap k and tap n-k share the DFT value.
// The dynamic raised-cosine is NOT even (ddsp/core.py:292-303: w>1 is zeroed BEFORE the cosine, w<-1 is not clamped),
// so it is evaluated separately for both taps.
struct EpiDynWindow {
    float* ir;
    int n;
    const float* f0;  // per row
    float sr15;       // 1.5 * sr as fp32
    int fast;         // split-bf16 products (taps carry ~4e-6 anyway): the window by one reciprocal per row and v_cos_f32 (~1e-6 of
                      // the weight) instead of two IEEE divisions and cosf per tap - the epilogue was 0.53 of the launch's vector issue
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ float window_fast(float inv_hw, int k) const {
        float w = (float)(k - n / 2) * inv_hw;
        if (w > 1.0f) w = 0.0f;
        return fmaf(0.5f, __builtin_amdgcn_cosf(0.5f * w), 0.5f);      // cos(pi w): v_cos_f32 takes revolutions
    }
    __device__ __forceinline__ float window(float hw, int k) const {
        float w = __fdiv_rn((float)(k - n / 2), hw);
        if (w > 1.0f) w = 0.0f;
        return __fdiv_rn(__fadd_rn(1.0f, cosf(__fmul_rn(3.14159274101257324f, w))), 2.0f);
    }
    __device__ __forceinline__ void operator()(int, int m, int k, float v, float) const {
        float* row = ir + (int64_t)m * n;
        if (fast) {
            const float inv_hw = __fdiv_rn(__fadd_rn(f0[m], 1e-3f), sr15);
            row[k] = v * window_fast(inv_hw, k);
            if (k > 0 && k < n / 2) row[n - k] = v * window_fast(inv_hw, n - k);
            return;
        }
        const float hw = __fdiv_rn(sr15, __fadd_rn(f0[m], 1e-3f));
        row[k] = __fmul_rn(v, window(hw, k));
        if (k > 0 && k < n / 2) row[n - k] = __fmul_rn(v, window(hw, n - k));
    }
};

struct EpiMirrorStore {  // row[k] = row[n-k] = acc
    float* ir;
    int n;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int, int m, int k, float v, float) const {
        float* row = ir + (int64_t)m * n;
        row[k] = v;
        if (k > 0 && k < n / 2) row[n - k] = v;
    }
};


// The same activations with several consecutive bins per lane (round 3): the kernel above gives a lane ONE bin per pass, so every
// value of a split row costs an 8-lane exchange and every store is 4 bytes - 41 us per step for its three launches, at 0.53 of
// the vector-issue rate.  REAL modes: 8 bins per lane (two 16-byte loads, the split group formed in the lane, two 16-byte stores).
__global__ void __launch_bounds__(256) fir_act_exp8_kernel(float scale, const float* __restrict__ ctrl, int64_t ld, int M, int ldo,
                                                           int64_t rows, float* __restrict__ out, int split) {
    const int per_row = M / 8;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    const int64_t row = i / per_row;
    const int gq = (int)(i % per_row);
    const float* src = ctrl + row * ld + 8 * gq;
    const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = expf(a[e]) * scale;      // (/128 is an exact scaling)
        v[4 + e] = expf(b[e]) * scale;
    }
    float* dst = out + row * ldo + 8 * gq;
    if (split) {
        ddsp_u32x4 hi, lo;
        ddsp_split8(v, hi, lo);
        *(ddsp_u32x4*)dst = hi;
        *(ddsp_u32x4*)(dst + 4) = lo;
    } else {
        *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// ALLPASS with 4 bins per lane: gd = pi tanh(ctrl), phi = the fp64 running sum rounded to fp32 per bin (ATen's CPU cumsum), out =
// [cos phi | sin phi].  A lane sums its four increments in fp64, the wave scans the lane totals once (row shifts and broadcasts of
// the two halves of the fp64 value), chunks of 256 bins carry on.  One wavefront per row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double act_dpp_d(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__global__ void __launch_bounds__(256) fir_act_allpass4_kernel(const float* __restrict__ ctrl, int64_t ld, int M, int ldo, int64_t rows,
                                                               float* __restrict__ out, int split) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = ctrl + row * ld;
    float* dst = out + row * ldo;
    const float pi_f = 3.14159274101257324f;
    double carry = 0.0;
    for (int f0 = 0; f0 < M; f0 += 256) {
        const int f = f0 + 4 * lane;                    // M % 8 == 0: a lane's four bins are all inside the row or all outside
        const bool in = f < M;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (in) x = *(const f32x4*)(src + f);
        double run = 0.0, loc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gd = in ? __fmul_rn(pi_f, tanhf(x[e])) : 0.f;
            run += (double)gd;
            loc[e] = run;
        }
        double incl = run;
        incl += act_dpp_d<0x111, 0xf>(incl);
        incl += act_dpp_d<0x112, 0xf>(incl);
        incl += act_dpp_d<0x114, 0xf>(incl);
        incl += act_dpp_d<0x118, 0xf>(incl);
        incl += act_dpp_d<0x142, 0xa>(incl);
        incl += act_dpp_d<0x143, 0xc>(incl);
        const double base = carry + (incl - run);
        f32x4 c, sn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float phi = (float)(base + loc[e]);
            float s_ = 0.f, c_ = 0.f;
            if (in) sincosf(phi, &s_, &c_);
            c[e] = c_;
            sn[e] = s_;
        }
        {
            const uint64_t u = __builtin_bit_cast(uint64_t, incl);
            const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)u, 63), hi = __builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
            carry += __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
        }
        if (split) {
            // lanes 2j, 2j + 1 own one group of 8 bins (every lane takes part in the exchange)
            const ddsp_u32x4 pc = ddsp_split4_pair(c, (lane & 1) != 0, 1), ps = ddsp_split4_pair(sn, (lane & 1) != 0, 1);
            if (in) {
                *(ddsp_u32x4*)(dst + f) = pc;
                *(ddsp_u32x4*)(dst + M + f) = ps;
            }
        } else if (in) {
            *(f32x4*)(dst + f) = c;
            *(f32x4*)(dst + M + f) = sn;
        }
    }
}

// ---- backward -----------------------------------------------------------------------------------------------
// rows of n floats (n even: 8-byte aligned) -> rows of ld >= n floats, zero beyond n
__global__ void __launch_bounds__(256) repack_rows_kernel(const float* __restrict__ src, int n, int64_t rows, int ld,
                                                          float* __restrict__ dst) {
    typedef float f32x2r __attribute__((ext_vector_type(2)));
    const int half = ld / 2;
    const int64_t total = rows * half;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / half;
        const int k = 2 * (int)(i % half);
        f32x2r v = {0.f, 0.f};
        if (k < n) v = *(const f32x2r*)(src + m * n + k);
        *(f32x2r*)(dst + m * ld + k) = v;
    }
}

__global__ void __launch_bounds__(256) dyn_window_scale_kernel(float* __restrict__ d_ir, const float* __restrict__ f0,
                                                               int n, int64_t rows, float sr15) {
    const int64_t total = rows * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / n;
        const int k = (int)(i % n);
        const float hw = __fdiv_rn(sr15, __fadd_rn(f0[m], 1e-3f));
        float w = __fdiv_rn((float)(k - n / 2), hw);
        if (w > 1.0f) w = 0.0f;
        d_ir[i] *= __fdiv_rn(__fadd_rn(1.0f, cosf(__fmul_rn(3.14159274101257324f, w))), 2.0f);
    }
}

// d_ctrl from d_act (gradient w.r.t. the activated responses), one wavefront per frame row
__global__ void __launch_bounds__(256) fir_act_bwd_kernel(int mode, const float* __restrict__ ctrl, int64_t ld, int M,
                                                          const float* __restrict__ d_act, int lda, int64_t rows,
                                                          float* __restrict__ d_ctrl, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = ctrl + row * ld;
    const float* g = d_act + row * lda;
    float* dst = d_ctrl + row * ldo;
    if (mode != DDSP_FIR_ALLPASS) {
        const float scale = (mode == DDSP_FIR_STATIC) ? (1.0f / 128.0f) : 1.0f;
        for (int f = lane; f < M; f += 64) dst[f] = g[f] * (expf(src[f]) * scale);
        return;
    }
    const float pi_f = 3.14159274101257324f;
    // forward phases again (same arithmetic as fir_act_kernel), kept in registers: up to 16 chunks of 64 bins
    float th[16], dphi[16];
    double carry = 0.0;
    const int nchunk = (M + 63) / 64;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        th[c] = 0.f;
        dphi[c] = 0.f;
        if (c < nchunk) {
            const int f = c * 64 + lane;
            const float t = (f < M) ? tanhf(src[f]) : 0.f;
            th[c] = t;
            double incl = (double)((f < M) ? __fmul_rn(pi_f, t) : 0.f);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                double nb = __shfl_up(incl, o, 64);
                if (lane >= o) incl += nb;
            }
            const float phi = (float)(carry + incl);
            carry += __shfl(incl, 63, 64);
            if (f < M) {
                double sn, cs;
                sincos((double)phi, &sn, &cs);
                dphi[c] = (float)(-sn * (double)g[f] + cs * (double)g[M + f]);
            }
        }
    }
    // reverse inclusive cumulative sum of dphi over bins, then through pi*tanh
    float tail = 0.f;
#pragma unroll
    for (int c = 15; c >= 0; --c) {
        if (c < nchunk) {
            float incl = dphi[c];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                float nb = __shfl_down(incl, o, 64);
                if (lane + o < 64) incl += nb;
            }
            const float G = incl + tail;
            tail += __shfl(incl, 0, 64);
            const int f = c * 64 + lane;
            if (f < M) dst[f] = G * pi_f * (1.0f - th[c] * th[c]);
        }
    }
}

}  // namespace


static bool dyn_fast() {   // DDSP_FIR_DYN_FAST=0: the exact window arithmetic in every mode (measurement aid)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_FIR_DYN_FAST");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

extern "C" int ddsp_fir_from_ctrl(ddsp_ctx* ctx, void* stream, int mode, const float* ctrl, int64_t ctrl_ld,
                                  int n_mag, const float* f0_frames, int64_t rows, int sr, float* ir) {
    DDSP_REQUIRE(ctx, ctx && ctrl && ir, "ddsp_fir_from_ctrl: null argument");
    DDSP_REQUIRE(ctx, mode >= 0 && mode <= 2, "ddsp_fir_from_ctrl: unknown mode");
    DDSP_REQUIRE(ctx, n_mag >= 3 && n_mag <= 1024 && ctrl_ld >= n_mag && rows >= 0 && rows < (1 << 30),
                 "ddsp_fir_from_ctrl: bad shape");
    DDSP_REQUIRE(ctx, mode != DDSP_FIR_DYNAMIC || f0_frames, "ddsp_fir_from_ctrl: DYNAMIC needs f0_frames");
    if (rows == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int M = n_mag, n = 2 * (n_mag - 1);
    const int K = (mode == DDSP_FIR_ALLPASS) ? 2 * M : M;
    const int lda = ddsp_pad4(K);
    int rc = ddsp_scratch_reserve_bytes(ctx, (size_t)rows * lda * sizeof(float) + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float* act = nullptr;
    rc = ddsp_scratch_get(ctx, (size_t)rows * lda * sizeof(float), (void**)&act);
    if (rc) return rc;
    float* tab = nullptr;
    const int kind = mode == DDSP_FIR_ALLPASS ? TAB_IRDFT_CPLX : (mode == DDSP_FIR_STATIC ? TAB_IRDFT_RE_HANN : TAB_IRDFT_RE);
    // K a multiple of 32 (every shipped n_mag): the table is read tap-major, [n][K], so that both GEMM operands are
    // k-contiguous and the product runs on the LDS-DMA kernel; otherwise frequency-major on the register-staged one
    const bool tap_major = K % 32 == 0;
    rc = ddsp_get_table(ctx, st, kind, M, tap_major ? 1 : 0, &tab);
    if (rc) return rc;
    float* tab_split = nullptr;   // the same table already split into bf16 hi/lo (B operand of the split-bf16 GEMM)
    if (tap_major && ctx->math != DDSP_MATH_FP32 && (rc = ddsp_get_table(ctx, st, kind, M, 2, &tab_split))) return rc;

    // large row counts with split-bf16 products: the activations are written as bf16 hi/lo groups and the GEMM reads both
    // operands already split (the DMA kernel is certain at these sizes; B = B_split makes a fallback fail loudly)
    const bool presplit = tap_major && tab_split && rows >= 8192 && M % 8 == 0 && ctx->math != 4;
    ddsp_prof_begin(ctx, st, PF_FIR_ACT);
    static int act_wide = -1;   // DDSP_FIR_ACT_WIDE=0: one bin per lane and pass at every shape (measurement aid)
    if (act_wide < 0) {
        const char* e = getenv("DDSP_FIR_ACT_WIDE");
        act_wide = (e && e[0] == '0') ? 0 : 1;
    }
    const bool wide = act_wide && M % 8 == 0 && ctrl_ld % 4 == 0 && lda % 8 == 0 && ((uintptr_t)ctrl % 16) == 0;
    if (wide && mode != DDSP_FIR_ALLPASS)
        hipLaunchKernelGGL(fir_act_exp8_kernel, dim3((unsigned)ceil_div64(rows * (M / 8), 256)), dim3(256), 0, st,
                           mode == DDSP_FIR_STATIC ? 1.0f / 128.0f : 1.0f, ctrl, ctrl_ld, M, lda, rows, act, presplit ? 1 : 0);
    else if (wide)
        hipLaunchKernelGGL(fir_act_allpass4_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, ctrl, ctrl_ld, M, lda, rows,
                           act, presplit ? 1 : 0);
    else
        hipLaunchKernelGGL(fir_act_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, mode, ctrl, ctrl_ld, M,
                           lda, rows, act, presplit ? 1 : 0);
    ddsp_prof_end(ctx, st, 0.0, 4.0 * rows * (M + K));
    DDSP_LAUNCH_CHECK(ctx);
    ddsp_prof_begin(ctx, st, PF_FIR_DFT_GEMM);

    auto run = [&](gemm::Args g, const auto& epi) {
        // (split-bf16 products like the control network's inference GEMMs: the taps then carry ~4e-6 relative error)
        g.math = tap_major ? (ctx->math == 4 ? DDSP_MATH_SPLIT_BF16 : ctx->math) : 0;   // ddsp_ctx_set_math
        g.B_split = tab_split;
        if (presplit) {
            g.A_split = 1;
            g.B = tab_split;
        }
        if (tap_major)
            gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, epi);
        else
            gemm::launch<true, false, gemm::A_PLAIN>(st, g, 1, epi);
    };
    const int64_t ldb = tap_major ? K : ddsp_pad4(n);
    if (mode == DDSP_FIR_ALLPASS) {
        gemm::Args g = gemm::make(act, lda, tab, ldb, (int)rows, n, K);
        run(g, gemm::EpiStore{ir, n, nullptr, 1, 0, 0});
    } else {
        // even filter: taps 0..n/2 from the GEMM, the rest mirrored by the epilogue
        gemm::Args g = gemm::make(act, lda, tab, ldb, (int)rows, n / 2 + 1, K);
        if (mode == DDSP_FIR_DYNAMIC)
            run(g, EpiDynWindow{ir, n, f0_frames, 1.5f * (float)sr, (tap_major && ctx->math != DDSP_MATH_FP32 && dyn_fast()) ? 1 : 0});
        else
            run(g, EpiMirrorStore{ir, n});
    }
    ddsp_prof_end(ctx, st, 2.0 * rows * (double)n * K, 4.0 * rows * (K + n));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_fir_from_ctrl_bwd(ddsp_ctx* ctx, void* stream, int mode, const float* ctrl, int64_t ctrl_ld,
                                      int n_mag, const float* f0_frames, int64_t rows, int sr, float* d_ir,
                                      float* d_ctrl, int64_t d_ctrl_ld) {
    DDSP_REQUIRE(ctx, ctx && ctrl && d_ir && d_ctrl, "ddsp_fir_from_ctrl_bwd: null argument");
    DDSP_REQUIRE(ctx, mode >= 0 && mode <= 2, "ddsp_fir_from_ctrl_bwd: unknown mode");
    DDSP_REQUIRE(ctx, n_mag >= 3 && n_mag <= 1024 && ctrl_ld >= n_mag && d_ctrl_ld >= n_mag && rows >= 0 && rows < (1 << 30),
                 "ddsp_fir_from_ctrl_bwd: bad shape");
    DDSP_REQUIRE(ctx, mode != DDSP_FIR_DYNAMIC || f0_frames, "ddsp_fir_from_ctrl_bwd: DYNAMIC needs f0_frames");
    if (rows == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int M = n_mag, n = 2 * (n_mag - 1);
    const int K = (mode == DDSP_FIR_ALLPASS) ? 2 * M : M;
    const int lda = ddsp_pad4(K);
    // The contraction over the n taps (510 / 1022: rows of d_ir are not 16-byte aligned) is a linear map of the gradient: with
    // split-bf16 products allowed it runs on the LDS-DMA kernel from a copy of d_ir with the table's row pitch (512 / 1024, zero
    // pad columns) instead of the register-staged fp32 kernel.  DDSP_FIR_BWD_DMA=0: the fp32 kernel (measurement aid)
    static int bwd_dma = -1;
    if (bwd_dma < 0) {
        const char* e = getenv("DDSP_FIR_BWD_DMA");
        bwd_dma = (e && e[0] == '0') ? 0 : 1;
    }
    const int ldp = ddsp_pad4(n);
    const bool repack = bwd_dma && ctx->math != DDSP_MATH_FP32 && ldp % 32 == 0 && rows >= 1024;
    int rc = ddsp_scratch_reserve_bytes(ctx, (size_t)rows * (lda + (repack ? ldp : 0)) * sizeof(float) + 8192);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float* d_act = nullptr;
    if ((rc = ddsp_scratch_get(ctx, (size_t)rows * lda * sizeof(float), (void**)&d_act))) return rc;
    float* d_pad = nullptr;
    if (repack && (rc = ddsp_scratch_get(ctx, (size_t)rows * ldp * sizeof(float), (void**)&d_pad))) return rc;
    float* tab = nullptr;
    const int kind = mode == DDSP_FIR_ALLPASS ? TAB_IRDFT_CPLX : (mode == DDSP_FIR_STATIC ? TAB_IRDFT_RE_HANN : TAB_IRDFT_RE);
    if ((rc = ddsp_get_table(ctx, st, kind, M, 0, &tab))) return rc;
    ddsp_prof_begin(ctx, st, PF_FIR_SYNTH_BWD);
    if (mode == DDSP_FIR_DYNAMIC) {
        int64_t blocks = ceil_div64(rows * n, 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(dyn_window_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_ir, f0_frames, n, rows,
                           1.5f * (float)sr);
    }
    // d_act[m][f] = sum_k d_ir[m][k] * T[f][k]
    gemm::EpiStore e{d_act, lda, nullptr, 1, 0, 0};
    if (repack) {
        int64_t blocks = ceil_div64(rows * (ldp / 2), 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_ir, n, rows, ldp, d_pad);
        gemm::Args g = gemm::make(d_pad, ldp, tab, ldp, (int)rows, K, ldp);
        g.math = DDSP_MATH_SPLIT_BF16;
        gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e);
    } else {
        gemm::Args g = gemm::make(d_ir, n, tab, ddsp_pad4(n), (int)rows, K, n);
        gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e);
    }
    hipLaunchKernelGGL(fir_act_bwd_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, mode, ctrl, ctrl_ld, M,
                       d_act, lda, rows, d_ctrl, d_ctrl_ld);
    ddsp_prof_end(ctx, st, 2.0 * rows * (double)n * K, 4.0 * rows * (2.0 * K + n));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
