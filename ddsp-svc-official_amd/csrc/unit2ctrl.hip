// a4: the unit -> control network (conv prenet + side embeddings + 3 x [Performer attention, Conformer conv
// module] + LayerNorm + weight-normed head), non-causal.
//
// Replaces ddsp/unit2control.py:23-101 and ddsp/pcmer.py:11-63,69-77,123-159,191-251.
// Activations live frame-major (rows = B*Fr frames, 256/512/1024 channels contiguous), so every Linear /
// 1x1 conv is a plain row-major GEMM and the k=3 convolutions are GEMMs with an implicit im2col loader
// (gemm_f32.h, A_CONV3).  All contractions run on the fp32 matrix pipe (exact fp32: exp() of the control
// values amplifies GEMM error, so bf16/fp16 operands would break the 1e-4 waveform gate).
// Linear attention is evaluated per (utterance, head) with batched GEMMs: k'^T v (266x64 context) and
// q' ctx; the softmax-kernel feature maps are a GEMM against the fixed 266x64 projection followed by a
// row-wise exp pass.  Everything else (GroupNorm, LayerNorm, GLU, depthwise k=31 conv + SiLU, embeddings,
// weight-norm) is small fused elementwise / row-reduction kernels.
#include "gemm_f32.h"

namespace {

constexpr int D = 256;        // model width
constexpr int H = 8;          // heads
constexpr int DH = 64;        // head dim
constexpr int INNER = 512;    // H*DH, also the conv-module inner width
constexpr int NF = 266;       // random features, int(64*ln 64)
constexpr int LDF = 268;      // padded leading dim of feature rows (16-byte aligned rows)
constexpr int DWK = 31;       // depthwise kernel

// ---- weight preparation --------------------------------------------------------------------------
// conv weight (Cout, Cin, 3) -> (Cout, 3*Cin) with k = tap*Cin + c  (matches gemm A_CONV3)
__global__ void pack_conv3_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ out) {
    const int64_t total = (int64_t)Cout * Cin * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 3);
        const int c = (int)((i / 3) % Cin);
        const int o = (int)(i / (3 * Cin));
        out[(int64_t)o * 3 * Cin + (int64_t)tap * Cin + c] = w[i];
    }
}

// W[o][:] = g[o] * v[o][:] / ||v[o]||_2   (old-style weight_norm, ddsp/unit2control.py:61); one wave per row
__global__ void __launch_bounds__(256) weight_norm_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                          int n_out, int n_in, float* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n_out) return;
    const float* row = v + (int64_t)o * n_in;
    float ss = 0.f;
    for (int i = lane; i < n_in; i += 64) ss = fmaf(row[i], row[i], ss);
    ss = wave_sum(ss);
    const float scale = g[o] / sqrtf(ss);
    for (int i = lane; i < n_in; i += 64) w[(int64_t)o * n_in + i] = row[i] * scale;
}

// ---- GroupNorm(4, 256) over (64 channels x all frames) per utterance + LeakyReLU -------------------
__global__ void __launch_bounds__(256) groupnorm_stats_kernel(const float* __restrict__ x, int Fr,
                                                              float* __restrict__ stats) {
    // block = (group g, utterance b); 256 threads = 64 channels x 4 frame lanes
    const int g = blockIdx.x, b = blockIdx.y;
    const int c = threadIdx.x & 63, fl = threadIdx.x >> 6;
    const float* base = x + ((int64_t)b * Fr) * D + g * 64 + c;
    double s = 0.0, ss = 0.0;
    for (int f = fl; f < Fr; f += 4) {
        const double v = (double)base[(int64_t)f * D];
        s += v;
        ss += v * v;
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) {
        red[fl] = s;
        red[4 + fl] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = 64.0 * Fr;
        const double S = red[0] + red[1] + red[2] + red[3];
        const double SS = red[4] + red[5] + red[6] + red[7];
        const double mean = S / n;
        double var = SS / n - mean * mean;
        if (var < 0) var = 0;
        stats[(b * 4 + g) * 2 + 0] = (float)mean;
        stats[(b * 4 + g) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

__global__ void __launch_bounds__(256) groupnorm_lrelu_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int64_t rows, int Fr,
                                                              float* __restrict__ out) {
    const int64_t total = rows * (D / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / (D / 4);
        const int c4 = (int)(i % (D / 4)) * 4;
        const int b = (int)(m / Fr), g = c4 >> 6;
        const float mean = stats[(b * 4 + g) * 2], rstd = stats[(b * 4 + g) * 2 + 1];
        const f32x4 v = *(const f32x4*)(x + m * D + c4);
        const f32x4 ga = *(const f32x4*)(gamma + c4), be = *(const f32x4*)(beta + c4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y = fmaf((v[j] - mean) * rstd, ga[j], be[j]);
            o[j] = y > 0.f ? y : 0.01f * y;
        }
        *(f32x4*)(out + m * D + c4) = o;
    }
}

// ---- side embeddings: x += Lin(ln(1+f0/700)) + Lin(phase/pi) + Lin(volume) + speaker ---------------
struct MixArgs {
    int n;
    long long ids[16];
    float w[16];
};

__global__ void __launch_bounds__(256) embed_add_kernel(float* __restrict__ x, const float* __restrict__ f0,
                                                        const float* __restrict__ phase, const float* __restrict__ vol,
                                                        const ddsp_u2c_weights w, const int64_t* __restrict__ spk_id,
                                                        int64_t n_spk_id, MixArgs mix, int64_t rows, int Fr) {
    const int64_t total = rows * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / D;
        const int c = (int)(i % D);
        const float lf0 = logf(1.0f + __fdiv_rn(f0[m], 700.0f));
        const float ph = __fdiv_rn(phase[m], 3.14159274101257324f);
        float v = x[i];
        v += fmaf(lf0, w.f0_w[c], w.f0_b[c]);
        v += fmaf(ph, w.phase_w[c], w.phase_b[c]);
        v += fmaf(vol[m], w.volume_w[c], w.volume_b[c]);
        if (mix.n > 0) {
            for (int k = 0; k < mix.n; ++k) v += mix.w[k] * w.spk_table[(mix.ids[k] - 1) * D + c];
        } else {
            const int64_t b = m / Fr;
            const int64_t id = spk_id[n_spk_id == 1 ? 0 : b];
            v += w.spk_table[(id - 1) * D + c];
        }
        x[i] = v;
    }
}

// ---- LayerNorm over 256 channels, one wave per row (4 channels per lane) ----------------------------
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t rows,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const f32x4 v = *(const f32x4*)(x + m * D + lane * 4);
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / D);
    f32x4 d;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        d[j] = v[j] - mean;
        ss = fmaf(d[j], d[j], ss);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / D) + 1e-5f);
    const f32x4 ga = *(const f32x4*)(gamma + lane * 4), be = *(const f32x4*)(beta + lane * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaf(d[j] * rstd, ga[j], be[j]);
    *(f32x4*)(out + m * D + lane * 4) = o;
}

// ---- softmax-kernel feature map (ddsp/pcmer.py:123-159), in place on the projected rows ---------------
// feat rows (rows8 = B*Fr*8, LDF): raw = data . P^T ; data rows (rows8, 64).
// query: r*(exp(dn*raw - diag - max_j(dn*raw)) + 1e-4) ; key: r*exp(dn*raw - diag + 1e-4)
template <bool QUERY>
__global__ void __launch_bounds__(256) feature_map_kernel(float* __restrict__ feat, const float* __restrict__ data,
                                                          int64_t rows8) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows8) return;
    const float dn = 0.35355339059327373f;   // 64^-0.25
    const float ratio = 0.06131393394849658f;  // 266^-0.5
    const float x = data[r * DH + lane];
    const float diag = wave_sum(x * x) * 0.5f * (dn * dn);
    float* row = feat + r * LDF;
    float dd[5];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int j = lane + 64 * i;
        dd[i] = (j < NF) ? dn * row[j] : -3.0e38f;
        mx = fmaxf(mx, dd[i]);
    }
    if (QUERY) mx = wave_max(mx);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int j = lane + 64 * i;
        if (j < NF) {
            float o;
            if (QUERY)
                o = ratio * (expf((dd[i] - diag) - mx) + 1e-4f);
            else
                o = ratio * expf((dd[i] - diag) + 1e-4f);
            row[j] = o;
        } else if (j < LDF) {
            row[j] = 0.f;
        }
    }
}

// ks[b,h,j] = sum_n k'[b,n,h,j]   block = (b*8+h); thread j strides over features, loops frames
__global__ void __launch_bounds__(320) key_sum_kernel(const float* __restrict__ kf, int Fr, float* __restrict__ ks) {
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int j = threadIdx.x;
    if (j >= LDF) return;
    const float* base = kf + (((int64_t)b * Fr) * H + h) * LDF + j;
    float s = 0.f;
    for (int n = 0; n < Fr; ++n) s += base[(int64_t)n * H * LDF];
    ks[(int64_t)bh * LDF + j] = s;
}

// dinv[r] = 1 / (q'[r,:] . ks[b,h,:] + 1e-8)   one wave per (frame, head) row
__global__ void __launch_bounds__(256) attn_denominator_kernel(const float* __restrict__ qf, const float* __restrict__ ks,
                                                               int Fr, int64_t rows8, float* __restrict__ dinv) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows8) return;
    const int h = (int)(r % H);
    const int64_t b = (r / H) / Fr;
    const float* q = qf + r * LDF;
    const float* k = ks + (b * H + h) * LDF;
    float s = 0.f;
    for (int j = lane; j < NF; j += 64) s = fmaf(q[j], k[j], s);
    s = wave_sum(s);
    if (lane == 0) dinv[r] = 1.0f / (s + 1e-8f);
}

struct EpiAttnOut {  // out[(b*Fr+n)*512 + h*64 + e] = dinv[(b*Fr+n)*8+h] * acc   (z = b*8+h, m = n, col = e)
    float* out;
    const float* dinv;
    int Fr;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int z, int m, int e, float v, float) const {
        const int b = z / H, h = z % H;
        const int64_t row = (int64_t)b * Fr + m;
        out[row * INNER + h * DH + e] = dinv[row * H + h] * v;
    }
};

// ---- conformer conv module pieces --------------------------------------------------------------------
__global__ void __launch_bounds__(256) glu_kernel(const float* __restrict__ g1, int64_t rows, float* __restrict__ out) {
    const int64_t total = rows * (INNER / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / (INNER / 4);
        const int c4 = (int)(i % (INNER / 4)) * 4;
        const f32x4 a = *(const f32x4*)(g1 + m * 2 * INNER + c4);
        const f32x4 g = *(const f32x4*)(g1 + m * 2 * INNER + INNER + c4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] * (1.0f / (1.0f + expf(-g[j])));
        *(f32x4*)(out + m * INNER + c4) = o;
    }
}

// depthwise Conv1d(k=31, pad 15, groups=512) over frames + SiLU; weight (512,1,31).
// One thread owns one channel for a run of DW_RUN consecutive frames of one utterance: its 31 taps and a
// sliding window of DW_RUN+30 inputs stay in registers (2.9 loads per output instead of 31); lanes walk
// channels, so every load/store of a wavefront is one contiguous 256-B row segment.
constexpr int DW_RUN = 16;
__global__ void __launch_bounds__(256) dwconv_silu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int B, int Fr,
                                                          float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;       // channel (INNER = 512 -> 2 blocks in x)
    const int runs = (Fr + DW_RUN - 1) / DW_RUN;
    const int b = blockIdx.y / runs, f0 = (blockIdx.y % runs) * DW_RUN;
    float wt[DWK];
#pragma unroll
    for (int t = 0; t < DWK; ++t) wt[t] = w[c * DWK + t];
    const float* xb = x + ((int64_t)b * Fr) * INNER + c;
    float win[DW_RUN + DWK - 1];
#pragma unroll
    for (int i = 0; i < DW_RUN + DWK - 1; ++i) {
        const int f = f0 + i - DWK / 2;
        win[i] = (f >= 0 && f < Fr) ? xb[(int64_t)f * INNER] : 0.f;
    }
    const float bi = bias[c];
#pragma unroll
    for (int o = 0; o < DW_RUN; ++o) {
        float acc = bi;
#pragma unroll
        for (int t = 0; t < DWK; ++t) acc = fmaf(wt[t], win[o + t], acc);
        if (f0 + o < Fr) out[((int64_t)b * Fr + f0 + o) * INNER + c] = acc * (1.0f / (1.0f + expf(-acc)));
    }
}

#define PROF(id, flops, bytes, ...)              \
    do {                                          \
        ddsp_prof_begin(ctx, st, id);             \
        __VA_ARGS__;                              \
        ddsp_prof_end(ctx, st, (double)(flops), (double)(bytes)); \
    } while (0)

inline unsigned grid_for(int64_t total, int per_block = 256, int cap = 8192) {
    int64_t g = (total + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int ddsp_unit2ctrl_fwd(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* wp, const float* units,
                                  const float* f0_frames, const float* phase_frames, const float* volume,
                                  const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                                  const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, float* ctrl) {
    DDSP_REQUIRE(ctx, ctx && wp && units && f0_frames && phase_frames && volume && ctrl, "ddsp_unit2ctrl_fwd: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && Fr >= 1 && B * Fr < (1 << 26), "ddsp_unit2ctrl_fwd: bad shape");
    DDSP_REQUIRE(ctx, n_mix >= 0 && n_mix <= 16, "ddsp_unit2ctrl_fwd: at most 16 mixed speakers");
    DDSP_REQUIRE(ctx, n_mix > 0 || (spk_id && (n_spk_id == 1 || n_spk_id == B)), "ddsp_unit2ctrl_fwd: spk_id must hold 1 or B ids");
    DDSP_REQUIRE(ctx, n_mix == 0 || (mix_ids_host && mix_w_host), "ddsp_unit2ctrl_fwd: mix arrays missing");
    const ddsp_u2c_weights w = *wp;
    DDSP_REQUIRE(ctx, w.n_unit >= 4 && w.n_unit % 4 == 0 && w.n_out >= 1 && w.n_spk >= 1, "ddsp_unit2ctrl_fwd: bad widths");
    for (int k = 0; k < n_mix; ++k)
        DDSP_REQUIRE(ctx, mix_ids_host[k] >= 1 && mix_ids_host[k] <= w.n_spk, "ddsp_unit2ctrl_fwd: mixed speaker id out of range");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t M = B * Fr, M8 = M * H;
    const int iM = (int)M;

    // ---- scratch plan (floats) ----
    const size_t n_w1 = (size_t)D * 3 * w.n_unit, n_w2 = (size_t)D * 3 * D, n_wh = (size_t)w.n_out * D;
    const size_t plan[] = {n_w1, n_w2, n_wh,
                           (size_t)M * D,       // t1
                           (size_t)M * D,       // t2
                           (size_t)M * D,       // x
                           (size_t)M * D,       // y
                           (size_t)M * INNER,   // q
                           (size_t)M * INNER,   // k
                           (size_t)M * INNER,   // v
                           (size_t)M8 * LDF,    // qf
                           (size_t)M8 * LDF,    // kf
                           (size_t)B * H * LDF, // ks
                           (size_t)B * H * NF * DH,  // ctx
                           (size_t)M8,          // dinv
                           (size_t)M * INNER,   // attn
                           (size_t)M * 2 * INNER,  // g1
                           (size_t)B * 4 * 2};  // gn stats
    size_t total = 0;
    for (size_t s : plan) total += ((s * sizeof(float) + 255) & ~(size_t)255) + 256;
    int rc = ddsp_scratch_reserve_bytes(ctx, total);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float* buf[sizeof(plan) / sizeof(plan[0])];
    for (size_t i = 0; i < sizeof(plan) / sizeof(plan[0]); ++i) {
        rc = ddsp_scratch_get(ctx, plan[i] * sizeof(float), (void**)&buf[i]);
        if (rc) return rc;
    }
    float *w1 = buf[0], *w2 = buf[1], *wh = buf[2], *t1 = buf[3], *t2 = buf[4], *x = buf[5], *y = buf[6], *q = buf[7],
          *k = buf[8], *v = buf[9], *qf = buf[10], *kf = buf[11], *ks = buf[12], *cx = buf[13], *dinv = buf[14],
          *attn = buf[15], *g1 = buf[16], *gst = buf[17];
    float* glu = q;  // q/k are dead once the attention output exists
    float* dw = k;

    // ---- weight preparation ----
    PROF(PF_U2C_PREP, 0, 8.0 * (n_w1 + n_w2 + n_wh),
         hipLaunchKernelGGL(pack_conv3_kernel, dim3(grid_for(n_w1)), dim3(256), 0, st, w.prenet_conv1_w, D, w.n_unit, w1);
         hipLaunchKernelGGL(pack_conv3_kernel, dim3(grid_for(n_w2)), dim3(256), 0, st, w.prenet_conv2_w, D, D, w2);
         hipLaunchKernelGGL(weight_norm_kernel, dim3((w.n_out + 3) / 4), dim3(256), 0, st, w.head_g, w.head_v, w.n_out, D, wh));

    // ---- prenet: conv k3 -> GroupNorm(4) -> LeakyReLU -> conv k3 ----
    {
        gemm::Args g = gemm::make(units, w.n_unit, w1, 3 * w.n_unit, iM, D, 3 * w.n_unit);
        g.Fr = (int)Fr;
        g.Cin = w.n_unit;
        gemm::EpiStore e{t1, D, w.prenet_conv1_b, 1, 0, 0};
        PROF(PF_U2C_GEMM_CONV3, 2.0 * M * D * 3 * w.n_unit, 4.0 * M * (w.n_unit + D),
             (gemm::launch<true, true, gemm::A_CONV3>(st, g, 1, e)));
    }
    PROF(PF_U2C_ROWWISE, 0, 4.0 * M * D,
         hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(4, (unsigned)B), dim3(256), 0, st, t1, (int)Fr, gst));
    PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
         hipLaunchKernelGGL(groupnorm_lrelu_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, st, t1, gst,
                            w.prenet_gn_w, w.prenet_gn_b, M, (int)Fr, t2));
    {
        gemm::Args g = gemm::make(t2, D, w2, 3 * D, iM, D, 3 * D);
        g.Fr = (int)Fr;
        g.Cin = D;
        gemm::EpiStore e{x, D, w.prenet_conv2_b, 1, 0, 0};
        PROF(PF_U2C_GEMM_CONV3, 2.0 * M * D * 3 * D, 8.0 * M * D, (gemm::launch<true, true, gemm::A_CONV3>(st, g, 1, e)));
    }
    MixArgs mix;
    mix.n = n_mix;
    for (int i = 0; i < n_mix; ++i) {
        mix.ids[i] = mix_ids_host[i];
        mix.w[i] = mix_w_host[i];
    }
    PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
         hipLaunchKernelGGL(embed_add_kernel, dim3(grid_for(M * D)), dim3(256), 0, st, x, f0_frames, phase_frames, volume,
                            w, spk_id, n_spk_id, mix, M, (int)Fr));
    DDSP_LAUNCH_CHECK(ctx);

    const unsigned rows_g = (unsigned)ceil_div64(M, 4), rows8_g = (unsigned)ceil_div64(M8, 4);
    for (int l = 0; l < 3; ++l) {
        const ddsp_u2c_layer& L = w.layer[l];
        // -- x += to_out(linear_attention(LN(x)))
        PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
             hipLaunchKernelGGL(layernorm_kernel, dim3(rows_g), dim3(256), 0, st, x, L.norm_w, L.norm_b, M, y));
        const float* pw[3] = {L.q_w, L.k_w, L.v_w};
        const float* pb[3] = {L.q_b, L.k_b, L.v_b};
        float* po[3] = {q, k, v};
        for (int i = 0; i < 3; ++i) {
            gemm::Args g = gemm::make(y, D, pw[i], D, iM, INNER, D);
            gemm::EpiStore e{po[i], INNER, pb[i], 1, 0, 0};
            PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * INNER * D, 4.0 * M * (D + INNER),
                 (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
        }
        {   // random-feature projections: (M*8, 64) x (266, 64)^T
            gemm::Args g = gemm::make(q, DH, L.proj, DH, (int)M8, NF, DH);
            gemm::EpiStore e{qf, LDF, nullptr, 1, 0, 0};
            PROF(PF_U2C_GEMM_FEAT, 2.0 * M8 * NF * DH, 4.0 * M8 * (DH + NF),
                 (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
            g.A = k;
            e.C = kf;
            PROF(PF_U2C_GEMM_FEAT, 2.0 * M8 * NF * DH, 4.0 * M8 * (DH + NF),
                 (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
        }
        PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * (2 * NF + DH),
             hipLaunchKernelGGL(feature_map_kernel<true>, dim3(rows8_g), dim3(256), 0, st, qf, q, M8));
        PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * (2 * NF + DH),
             hipLaunchKernelGGL(feature_map_kernel<false>, dim3(rows8_g), dim3(256), 0, st, kf, k, M8));
        PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * NF,
             hipLaunchKernelGGL(key_sum_kernel, dim3((unsigned)(B * H)), dim3(320), 0, st, kf, (int)Fr, ks));
        {   // ctx[b,h] (266 x 64) = k'^T v : A stored [n][j] (K x M), B stored [n][e] (K x N)
            gemm::Args g = gemm::make(kf, (int64_t)H * LDF, v, INNER, NF, DH, (int)Fr);
            g.zdiv = H;
            g.sA_hi = (int64_t)Fr * H * LDF;
            g.sA_lo = LDF;
            g.sB_hi = (int64_t)Fr * INNER;
            g.sB_lo = DH;
            gemm::EpiStore e{cx, DH, nullptr, 1, (int64_t)NF * DH, 0};
            PROF(PF_U2C_GEMM_CTX, 2.0 * M8 * NF * DH, 4.0 * M8 * (NF + DH),
                 (gemm::launch_tile<64, 64, false, false, gemm::A_PLAIN>(st, g, (int)(B * H), e)));
        }
        PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * NF,
             hipLaunchKernelGGL(attn_denominator_kernel, dim3(rows8_g), dim3(256), 0, st, qf, ks, (int)Fr, M8, dinv));
        {   // out[b,n,h,:] = dinv * (q'[b,n,h,:] ctx[b,h])
            gemm::Args g = gemm::make(qf, (int64_t)H * LDF, cx, DH, (int)Fr, DH, NF);
            g.zdiv = H;
            g.sA_hi = (int64_t)Fr * H * LDF;
            g.sA_lo = LDF;
            g.sB_hi = (int64_t)H * NF * DH;
            g.sB_lo = (int64_t)NF * DH;
            EpiAttnOut e{attn, dinv, (int)Fr};
            PROF(PF_U2C_GEMM_ATTNOUT, 2.0 * M8 * NF * DH, 4.0 * M8 * (NF + DH),
                 (gemm::launch_tile<64, 64, true, false, gemm::A_PLAIN>(st, g, (int)(B * H), e)));
        }
        {
            gemm::Args g = gemm::make(attn, INNER, L.out_w, INNER, iM, D, INNER);
            gemm::EpiResidual e{x, x, D, L.out_b};
            PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * D * INNER, 4.0 * M * (INNER + 2 * D),
                 (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
        }
        // -- x += conv_module(x)
        PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
             hipLaunchKernelGGL(layernorm_kernel, dim3(rows_g), dim3(256), 0, st, x, L.cm_ln_w, L.cm_ln_b, M, y));
        {
            gemm::Args g = gemm::make(y, D, L.cm_pw1_w, D, iM, 2 * INNER, D);
            gemm::EpiStore e{g1, 2 * INNER, L.cm_pw1_b, 1, 0, 0};
            PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * 2 * INNER * D, 4.0 * M * (D + 2 * INNER),
                 (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
        }
        PROF(PF_U2C_ROWWISE, 0, 12.0 * M * INNER,
             hipLaunchKernelGGL(glu_kernel, dim3(grid_for(M * (INNER / 4))), dim3(256), 0, st, g1, M, glu));
        PROF(PF_U2C_ROWWISE, 2.0 * M * INNER * DWK, 8.0 * M * INNER,
             hipLaunchKernelGGL(dwconv_silu_kernel, dim3(INNER / 256, (unsigned)(B * ((Fr + DW_RUN - 1) / DW_RUN))),
                                dim3(256), 0, st, glu, L.cm_dw_w, L.cm_dw_b, (int)B, (int)Fr, dw));
        {
            gemm::Args g = gemm::make(dw, INNER, L.cm_pw2_w, INNER, iM, D, INNER);
            gemm::EpiResidual e{x, x, D, L.cm_pw2_b};
            PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * D * INNER, 4.0 * M * (INNER + 2 * D),
                 (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
        }
        DDSP_LAUNCH_CHECK(ctx);
    }
    // ---- LayerNorm -> weight-normed head ----
    PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
         hipLaunchKernelGGL(layernorm_kernel, dim3(rows_g), dim3(256), 0, st, x, w.final_ln_w, w.final_ln_b, M, y));
    {
        gemm::Args g = gemm::make(y, D, wh, D, iM, w.n_out, D);
        gemm::EpiStore e{ctrl, w.n_out, w.head_b, 1, 0, 0};
        PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * w.n_out * D, 4.0 * M * (D + w.n_out),
             (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
    }
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
