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
#include "gemm_ws.h"
#include "gemm_ln.h"
#include "wgrad_bf16.h"
#include "performer_attn.h"

namespace {

constexpr int D = 256;        // model width
constexpr int H = 8;          // heads
constexpr int DH = 64;        // head dim
constexpr int INNER = 512;    // H*DH, also the conv-module inner width
constexpr int NF = 266;       // random features, int(64*ln 64)
constexpr int LDF = 268;      // padded leading dim of feature rows (16-byte aligned rows)
constexpr int DWK = 31;       // depthwise kernel

// ---- weight preparation --------------------------------------------------------------------------
// Every GEMM weight of a forward is re-packed by ONE launch (u2c_prepare_kernel).  Each job produces groups of 8
// consecutive k-values of one packed row and stores them either as 8 floats or, for the split-bf16 GEMMs that read B
// already split (gemm::Args::B_split), as 8 bf16 hi parts | 8 bf16 lo parts - the same 32 bytes.
__device__ __forceinline__ void store_group8(float* __restrict__ dst, const float (&x)[8], bool split) {
    if (split) {
        ddsp_u32x4 hi, lo;
        ddsp_split8(x, hi, lo);
        *(ddsp_u32x4*)dst = hi;
        *(ddsp_u32x4*)(dst + 4) = lo;
    } else {
        *(f32x4*)dst = f32x4{x[0], x[1], x[2], x[3]};
        *(f32x4*)(dst + 4) = f32x4{x[4], x[5], x[6], x[7]};
    }
}

// conv weight (Cout, Cin, 3) -> (Cout, 3*Cin) with k = tap*Cin + c  (matches gemm A_CONV3); Cin % 8 == 0
__device__ __forceinline__ void pack_conv3_body(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ out,
                                                int vb, int vgrid, bool split) {   // vb / vgrid: block id / grid size of this job
    const int64_t groups = (int64_t)Cout * 3 * (Cin / 8);
    for (int64_t gi = (int64_t)vb * 256 + threadIdx.x; gi < groups; gi += (int64_t)vgrid * 256) {
        const int c8 = (int)(gi % (Cin / 8));
        const int tap = (int)((gi / (Cin / 8)) % 3);
        const int o = (int)(gi / (3 * (Cin / 8)));
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w[((int64_t)o * Cin + 8 * c8 + e) * 3 + tap];
        store_group8(out + (int64_t)o * 3 * Cin + (int64_t)tap * Cin + 8 * c8, x, split);
    }
}
// fp32 variant for a Cin that is only a multiple of 4 (small problems that run the register-staged kernel)
__device__ __forceinline__ void pack_conv3_scalar_body(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ out,
                                                       int vb, int vgrid) {
    const int64_t total = (int64_t)Cout * Cin * 3;
    for (int64_t i = (int64_t)vb * 256 + threadIdx.x; i < total; i += (int64_t)vgrid * 256) {
        const int tap = (int)(i % 3);
        const int c = (int)((i / 3) % Cin);
        const int o = (int)(i / (3 * Cin));
        out[(int64_t)o * 3 * Cin + (int64_t)tap * Cin + c] = w[i];
    }
}
__global__ void __launch_bounds__(256) pack_conv3_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ out) {
    pack_conv3_scalar_body(w, Cout, Cin, out, blockIdx.x, gridDim.x);
}

// a plain (rows, K) matrix copied row by row (K % 8 == 0): the out-projection and pw2 weights for the split GEMMs
__device__ __forceinline__ void pack_copy_body(const float* __restrict__ w, int64_t n, float* __restrict__ out, int vb,
                                               int vgrid, bool split) {
    typedef gemm::f32x4_u v4;
    for (int64_t gi = (int64_t)vb * 256 + threadIdx.x; gi < n / 8; gi += (int64_t)vgrid * 256) {
        const v4 a = *(const v4*)(w + 8 * gi), c = *(const v4*)(w + 8 * gi + 4);
        const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        store_group8(out + 8 * gi, x, split);
    }
}

// q/k/v projections as ONE GEMM: rows [q_w; k_w; v_w] (3*512 x 256) and the three biases behind each other
__device__ __forceinline__ void pack_qkv_body(const float* __restrict__ qw, const float* __restrict__ kw,
                                              const float* __restrict__ vw, const float* __restrict__ qb,
                                              const float* __restrict__ kb, const float* __restrict__ vb_,
                                              float* __restrict__ w, float* __restrict__ bias, int vb, int vgrid, bool split) {
    typedef gemm::f32x4_u v4;
    const int n = INNER * D;
    for (int gi = vb * 256 + threadIdx.x; gi < 3 * n / 8; gi += vgrid * 256) {
        const int i = 8 * gi, which = i / n, j = i - which * n;
        const float* src = (which == 0 ? qw : (which == 1 ? kw : vw)) + j;
        const v4 a = *(const v4*)src, c = *(const v4*)(src + 4);
        const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        store_group8(w + i, x, split);
        if (j < INNER) {
            const float* bs = which == 0 ? qb : (which == 1 ? kb : vb_);
#pragma unroll
            for (int e = 0; e < 8; ++e) bias[which * INNER + j + e] = bs[j + e];
        }
    }
}

// pw1 (1024 x 256, rows 0..511 values, 512..1023 gates of the GLU, ddsp/pcmer.py conformer conv module) re-ordered
// for the gated-pair GEMM epilogue: packed rows 64t..64t+31 = values, 64t+32..64t+63 = gates of channels 32t..32t+31.
__device__ __forceinline__ void pack_glu_body(const float* __restrict__ w, const float* __restrict__ bias,
                                              float* __restrict__ wo, float* __restrict__ bo, int vb, int vgrid, bool split) {
    for (int i = vb * 256 + threadIdx.x; i < 2 * INNER * (D / 8); i += vgrid * 256) {
        const int p = i / (D / 8), k8 = (i % (D / 8)) * 8;
        const int t = p >> 6, within = p & 63;
        const int src = within < 32 ? 32 * t + within : INNER + 32 * t + (within - 32);
        const f32x4 a = *(const f32x4*)(w + (size_t)src * D + k8), c = *(const f32x4*)(w + (size_t)src * D + k8 + 4);
        const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        store_group8(wo + (size_t)p * D + k8, x, split);
        if (k8 == 0) bo[p] = bias[src];
    }
}

struct EpiGlu {  // out[m][c] = (a + bias_a) * sigmoid(g + bias_g), formed inside the GEMM (see gemm_f32.h kGatedPair)
    static constexpr bool kGatedPair = true;
    float* out;          // (rows, 512)
    const float* bias;   // packed like the weight rows
    __device__ __forceinline__ float col(int n) const { return bias[n]; }
    __device__ __forceinline__ void store4(int, int m, int n, f32x4 v) const { *(f32x4*)(out + (int64_t)m * INNER + n) = v; }
};

struct EpiSplit3 {  // column block n / 512 selects the destination matrix (q, k or v), each (rows, 512)
    float* out[3];
    const float* bias;
    __device__ __forceinline__ float col(int n) const { return bias[n]; }
    __device__ __forceinline__ void operator()(int, int m, int n, float v, float cb) const {
        out[n >> 9][(int64_t)m * INNER + (n & (INNER - 1))] = v + cb;
    }
    static constexpr bool kStore4 = true;
    __device__ __forceinline__ bool vec_ok() const {
        return (((uintptr_t)out[0] | (uintptr_t)out[1] | (uintptr_t)out[2]) % 16) == 0;
    }
    __device__ __forceinline__ void store4(int, int m, int n, f32x4 v) const {
        *(f32x4*)(out[n >> 9] + (int64_t)m * INNER + (n & (INNER - 1))) = v + *(const gemm::f32x4_u*)(bias + n);
    }
};

// W[o][:] = g[o] * v[o][:] / ||v[o]||_2   (old-style weight_norm, ddsp/unit2control.py:61); one wave per row
__device__ __forceinline__ void weight_norm_body(const float* __restrict__ g, const float* __restrict__ v, int n_out,
                                                 int n_in, float* __restrict__ w, int vb, bool split) {
    const int lane = threadIdx.x & 63;
    const int o = vb * 4 + (threadIdx.x >> 6);
    if (o >= n_out) return;
    const float* row = v + (int64_t)o * n_in;
    float ss = 0.f;
    for (int i = lane; i < n_in; i += 64) ss = fmaf(row[i], row[i], ss);
    ss = wave_sum(ss);
    const float scale = g[o] / sqrtf(ss);
    if (n_in % 8 == 0) {
        for (int i = 8 * lane; i < n_in; i += 512) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = row[i + e] * scale;
            store_group8(w + (int64_t)o * n_in + i, x, split);
        }
    } else {
        for (int i = lane; i < n_in; i += 64) w[(int64_t)o * n_in + i] = row[i] * scale;
    }
}
__global__ void __launch_bounds__(256) weight_norm_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                          int n_out, int n_in, float* __restrict__ w) {
    weight_norm_body(g, v, n_out, n_in, w, blockIdx.x, false);
}

// All weight preparation of one forward in ONE launch (it used to be 6-7: two conv packs, the head's weight norm,
// three QKV packs, the GLU re-ordering): the jobs are independent, each gets a range of blocks.  The weights arrive
// as raw pointers on every call, so they are re-prepared every call; what can be saved is the launches.
struct PrepArgs {
    ddsp_u2c_weights w;
    float *w1, *w2, *wh, *wqkv, *bqkv, *wglu, *bglu, *wdw, *wout, *wpw2;
    int end[7];          // one past the last block of: conv1 | conv2 | head | qkv (3 layers) | glu (3 layers, may be empty) |
                         // out-projection + pw2 copies (3 layers each, split mode only) | dw taps
    int qkv_blocks, glu_blocks, copy_blocks;   // blocks per layer
    int split;           // the packed matrices are written as bf16 hi/lo groups (gemm::Args::B_split)
};
__global__ void __launch_bounds__(256) u2c_prepare_kernel(PrepArgs a) {
    const int b = blockIdx.x;
    const bool split = a.split != 0;
    if (b < a.end[0]) {
        if (a.w.n_unit % 8 == 0)
            pack_conv3_body(a.w.prenet_conv1_w, D, a.w.n_unit, a.w1, b, a.end[0], split);
        else
            pack_conv3_scalar_body(a.w.prenet_conv1_w, D, a.w.n_unit, a.w1, b, a.end[0]);
    } else if (b < a.end[1]) {
        pack_conv3_body(a.w.prenet_conv2_w, D, D, a.w2, b - a.end[0], a.end[1] - a.end[0], split);
    } else if (b < a.end[2]) {
        weight_norm_body(a.w.head_g, a.w.head_v, a.w.n_out, D, a.wh, b - a.end[1], split);
    } else if (b < a.end[3]) {
        const int r = b - a.end[2], l = r / a.qkv_blocks;
        const ddsp_u2c_layer& L = a.w.layer[l];
        pack_qkv_body(L.q_w, L.k_w, L.v_w, L.q_b, L.k_b, L.v_b, a.wqkv + (size_t)l * 3 * INNER * D,
                      a.bqkv + (size_t)l * 3 * INNER, r - l * a.qkv_blocks, a.qkv_blocks, split);
    } else if (b < a.end[4]) {
        const int r = b - a.end[3], l = r / a.glu_blocks;
        const ddsp_u2c_layer& L = a.w.layer[l];
        pack_glu_body(L.cm_pw1_w, L.cm_pw1_b, a.wglu + (size_t)l * 2 * INNER * D, a.bglu + (size_t)l * 2 * INNER,
                      r - l * a.glu_blocks, a.glu_blocks, split);
    } else if (b < a.end[5]) {
        const int r = b - a.end[4], j = r / a.copy_blocks, l = j >> 1;     // job j: layer j/2, out-projection | pw2
        const ddsp_u2c_layer& L = a.w.layer[l];
        pack_copy_body((j & 1) ? L.cm_pw2_w : L.out_w, (int64_t)D * INNER, ((j & 1) ? a.wpw2 : a.wout) + (size_t)l * D * INNER,
                       r - j * a.copy_blocks, a.copy_blocks, true);
    } else {
        // depthwise taps (512, 1, 31) -> [tap][channel], all three layers: 3 * 31 * 512 elements
        for (int i = (b - a.end[5]) * 256 + threadIdx.x; i < 3 * DWK * INNER; i += (a.end[6] - a.end[5]) * 256) {
            const int l = i / (DWK * INNER), r = i - l * DWK * INNER;
            const int t = r / INNER, c = r - t * INNER;
            a.wdw[i] = a.w.layer[l].cm_dw_w[c * DWK + t];
        }
    }
}

// ---- GroupNorm(4, 256) over (64 channels x all frames) per utterance + LeakyReLU -------------------
constexpr int GN_LANES = 16;   // frame lanes (waves) per block of the statistics kernel
// part != null (small batches, conv1 run as a K-split): x[row][ch] is first completed as bias[ch] + sum_s part[s][row][ch]
// (rows = all frames of the call) and written back for the normalisation pass.
__global__ void __launch_bounds__(64 * GN_LANES) groupnorm_stats_kernel(float* __restrict__ x, int Fr, float* __restrict__ stats,
                                                                       const float* __restrict__ part = nullptr,
                                                                       const float* __restrict__ bias = nullptr, int64_t rows = 0) {
    // block = (group g, utterance b); 1024 threads = 64 channels x 16 frame lanes.  (With 4 frame lanes every thread
    // walked 43 dependent loads, one in flight at a time: 14 us for 11 MB.  The sums are combined in a fixed order.)
    const int g = blockIdx.x, b = blockIdx.y;
    const int c = threadIdx.x & 63, fl = threadIdx.x >> 6;
    float* base = x + ((int64_t)b * Fr) * D + g * 64 + c;
    double s = 0.0, ss = 0.0;
    int f = fl;
    if (part) {
        const float* pb = part + ((int64_t)b * Fr) * D + g * 64 + c;
        const float bc = bias[g * 64 + c];
        for (int ff = fl; ff < Fr; ff += GN_LANES) {
            const int64_t o = (int64_t)ff * D;
            const float v = ((pb[o] + pb[rows * D + o]) + (pb[2 * rows * D + o] + pb[3 * rows * D + o])) + bc;
            base[o] = v;
            s += (double)v;
            ss += (double)v * (double)v;
        }
        f = Fr;
    }
    for (; f + GN_LANES < Fr; f += 2 * GN_LANES) {          // two independent loads per trip
        const double v0 = (double)base[(int64_t)f * D], v1 = (double)base[(int64_t)(f + GN_LANES) * D];
        s += v0 + v1;
        ss += v0 * v0 + v1 * v1;
    }
    if (f < Fr) {
        const double v = (double)base[(int64_t)f * D];
        s += v;
        ss += v * v;
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    __shared__ double red[2 * GN_LANES];
    if ((threadIdx.x & 63) == 0) {
        red[fl] = s;
        red[GN_LANES + fl] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = 64.0 * Fr;
        double S = 0.0, SS = 0.0;
        for (int i = 0; i < GN_LANES; ++i) {
            S += red[i];
            SS += red[GN_LANES + i];
        }
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
                                                              float* __restrict__ out, int split) {
    // split != 0: the output is the A operand of a split-bf16 GEMM and is written as bf16 hi/lo groups (8 channels = two
    // neighbouring threads; rows * 64 threads, so a pair never straddles a wave)
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
        if (split)
            *(ddsp_u32x4*)(out + m * D + c4) = ddsp_split4_pair(o, (i & 1) != 0, 1);
        else
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
                                                        int64_t n_spk_id, MixArgs mix, int64_t rows, int Fr,
                                                        int* __restrict__ err) {
    // one wave per row, 4 channels per lane (the per-row log / divisions used to be redone for every element)
    const int lane = threadIdx.x & 63, c = lane * 4;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const float lf0 = logf(1.0f + __fdiv_rn(f0[m], 700.0f));
    const float ph = __fdiv_rn(phase[m], 3.14159274101257324f);
    const float vl = vol[m];
    typedef gemm::f32x4_u v4;                       // parameter tensors are only guaranteed dword-aligned
    const v4 fw = *(const v4*)(w.f0_w + c), fb = *(const v4*)(w.f0_b + c);
    const v4 pw = *(const v4*)(w.phase_w + c), pb = *(const v4*)(w.phase_b + c);
    const v4 vw = *(const v4*)(w.volume_w + c), vb = *(const v4*)(w.volume_b + c);
    f32x4 v = *(const f32x4*)(x + m * D + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] += fmaf(lf0, fw[j], fb[j]);
        v[j] += fmaf(ph, pw[j], pb[j]);
        v[j] += fmaf(vl, vw[j], vb[j]);
    }
    if (mix.n > 0) {
        for (int k = 0; k < mix.n; ++k) {
            const v4 e = *(const v4*)(w.spk_table + (mix.ids[k] - 1) * D + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += mix.w[k] * e[j];
        }
    } else {
        const int64_t id = spk_id[n_spk_id == 1 ? 0 : m / Fr];
        if (id >= 1 && id <= w.n_spk) {
            const v4 e = *(const v4*)(w.spk_table + (id - 1) * D + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += e[j];
        } else if (lane == 0) {
            // outside the table (the reference's nn.Embedding raises): nothing is read, the host is told
            __hip_atomic_store(err, DDSP_DEV_ERR_SPK_ID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    *(f32x4*)(x + m * D + c) = v;
}

// The same additions as the epilogue of the second prenet convolution (one launch and one pass over x less): same operation
// order as embed_add_kernel, so the result is bit-identical.
struct EpiEmbed {
    float* out;                 // (rows, 256)
    const float* bias;          // conv2 bias
    const float *f0, *phase, *vol;
    ddsp_u2c_weights w;
    const int64_t* spk_id;
    int64_t n_spk_id;
    MixArgs mix;
    int Fr;
    int* err;
    int64_t zstride;            // K-split launches: batch z stores its partial product at out + z * zstride, z = 0 adds bias + embeddings
    __device__ __forceinline__ float col(int n) const { return bias[n]; }
    __device__ __forceinline__ float embed(int m, int n, float v) const {
        const float lf0 = logf(1.0f + __fdiv_rn(f0[m], 700.0f));
        const float ph = __fdiv_rn(phase[m], 3.14159274101257324f);
        v += fmaf(lf0, w.f0_w[n], w.f0_b[n]);
        v += fmaf(ph, w.phase_w[n], w.phase_b[n]);
        v += fmaf(vol[m], w.volume_w[n], w.volume_b[n]);
        if (mix.n > 0) {
            for (int k = 0; k < mix.n; ++k) v += mix.w[k] * w.spk_table[(mix.ids[k] - 1) * D + n];
        } else {
            const int64_t id = spk_id[n_spk_id == 1 ? 0 : m / Fr];
            if (id >= 1 && id <= w.n_spk)
                v += w.spk_table[(id - 1) * D + n];
            else
                __hip_atomic_store(err, DDSP_DEV_ERR_SPK_ID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return v;
    }
    __device__ __forceinline__ void operator()(int z, int m, int n, float v, float cb) const {
        out[z * zstride + (int64_t)m * D + n] = z == 0 ? embed(m, n, v + cb) : v;
    }
    static constexpr bool kStore4 = true;
    __device__ __forceinline__ bool vec_ok() const { return ((uintptr_t)out % 16) == 0 && zstride % 4 == 0; }
    __device__ __forceinline__ void store4(int z, int m, int n, f32x4 v) const {
        typedef gemm::f32x4_u v4;
        if (z != 0) {
            *(f32x4*)(out + z * zstride + (int64_t)m * D + n) = v;
            return;
        }
        *(f32x4*)(out + (int64_t)m * D + n) = embed4(m, n, v);
    }
    // bias + the side embeddings of row m for columns n .. n + 3 (the operation order of store4 above, which calls it)
    __device__ __forceinline__ f32x4 embed4(int m, int n, f32x4 v) const {
        typedef gemm::f32x4_u v4;
        v += *(const v4*)(bias + n);
        const float lf0 = logf(1.0f + __fdiv_rn(f0[m], 700.0f));
        const float ph = __fdiv_rn(phase[m], 3.14159274101257324f);
        const float vl = vol[m];
        const v4 fw = *(const v4*)(w.f0_w + n), fb = *(const v4*)(w.f0_b + n);
        const v4 pw = *(const v4*)(w.phase_w + n), pb = *(const v4*)(w.phase_b + n);
        const v4 vw = *(const v4*)(w.volume_w + n), vb = *(const v4*)(w.volume_b + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += fmaf(lf0, fw[j], fb[j]);
            v[j] += fmaf(ph, pw[j], pb[j]);
            v[j] += fmaf(vl, vw[j], vb[j]);
        }
        if (mix.n > 0) {
            for (int k = 0; k < mix.n; ++k) {
                const v4 e = *(const v4*)(w.spk_table + (mix.ids[k] - 1) * D + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += mix.w[k] * e[j];
            }
        } else {
            const int64_t id = spk_id[n_spk_id == 1 ? 0 : m / Fr];
            if (id >= 1 && id <= w.n_spk) {
                const v4 e = *(const v4*)(w.spk_table + (id - 1) * D + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += e[j];
            } else {
                __hip_atomic_store(err, DDSP_DEV_ERR_SPK_ID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return v;
    }
};

// the same additions as the `Pre` of gemm::kernel_res_ln: second prenet convolution + embeddings + the first LayerNorm in one kernel
struct PreEmbed {
    EpiEmbed e;
    struct State {};
    template <int RB>
    __device__ __forceinline__ void load(State&, const gemm::LnArgs&, int, int, int, int, int) const {}
    __device__ __forceinline__ f32x4 apply(const State&, const gemm::LnArgs&, int, int, int m, int c0, f32x4 acc) const {
        return e.embed4(m, c0, acc);
    }
};

// ---- LayerNorm over 256 channels, one wave per row (4 channels per lane) ----------------------------
// Small batches (K-split GEMMs, see u2c_forward): the row to normalise is first completed here,
//   x_out[m] = x[m] + bias + sum_s part[s][m]   (the residual GEMM's epilogue, deferred: its K range was cut into KS_SPLITS
// workgroups per tile that each stored a partial product), written back for the next residual and then normalised.
struct LnPending {
    const float* part;   // [KS_SPLITS][rows][D] or null
    const float* bias;   // or null
    float* x_out;
    int has_res;         // the kernel's x argument is the residual to add (else the row is the partial sum alone)
};
constexpr int KS_SPLITS = 4;
constexpr int KS_MAX_ROWS = 256;   // up to 4 row tiles x 4 column tiles x 4 splits = 64 workgroups (344 rows measured slower than whole-K launches)
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t rows,
                                                        float* __restrict__ out, int split, LnPending pend = LnPending{nullptr, nullptr, nullptr, 0}) {
    // split != 0: the row is written as bf16 hi/lo groups (A operand of a split-bf16 GEMM): lanes 2j, 2j+1 own one group
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    f32x4 v;
    if (pend.part) {
        const f32x4 p0 = *(const f32x4*)(pend.part + m * D + lane * 4), p1 = *(const f32x4*)(pend.part + (rows + m) * D + lane * 4);
        const f32x4 p2 = *(const f32x4*)(pend.part + (2 * rows + m) * D + lane * 4), p3 = *(const f32x4*)(pend.part + (3 * rows + m) * D + lane * 4);
        v = (p0 + p1) + (p2 + p3);
        if (pend.bias) v = v + *(const f32x4*)(pend.bias + lane * 4);
        if (pend.has_res) v = *(const f32x4*)(x + m * D + lane * 4) + v;
        *(f32x4*)(pend.x_out + m * D + lane * 4) = v;
    } else {
        v = *(const f32x4*)(x + m * D + lane * 4);
    }
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
    if (split)
        *(ddsp_u32x4*)(out + m * D + lane * 4) = ddsp_split4_pair(o, (lane & 1) != 0, 1);
    else
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

// ks[b,h,j] = sum_n k'[b,n,h,j]   block = (b*8+h): 72 threads x 16 bytes cover the 288 padded features of a row, 8 frame
// lanes walk the frames with two rows in flight each and meet in the LDS (one thread per feature walking all frames alone
// read at 1.2 TB/s: 41 us per call at the training shape)
constexpr int KS_T = LDF / 4;   // threads per row
static_assert(LDF % 4 == 0 && KS_T * 8 <= 1024, "key_sum block");
__global__ void __launch_bounds__(KS_T * 8) key_sum_kernel(const float* __restrict__ kf, int Fr, float* __restrict__ ks) {
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int j4 = threadIdx.x % KS_T, fl = threadIdx.x / KS_T;
    const float* base = kf + (((int64_t)b * Fr) * H + h) * LDF + 4 * j4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    int n = fl;
    for (; n + 8 < Fr; n += 16) {
        s0 += *(const f32x4*)(base + (int64_t)n * H * LDF);
        s1 += *(const f32x4*)(base + (int64_t)(n + 8) * H * LDF);
    }
    if (n < Fr) s0 += *(const f32x4*)(base + (int64_t)n * H * LDF);
    __shared__ f32x4 red[KS_T * 8];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (fl == 0) {
        f32x4 t = red[j4];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += red[i * KS_T + j4];
        *(f32x4*)(ks + (int64_t)bh * LDF + 4 * j4) = t;
    }
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

// DDSP_CONV_LN=0: the second prenet convolution and the first LayerNorm as two launches at every size (measurement aid)
static bool conv_ln_on() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_CONV_LN");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// DDSP_CAUSAL_CHUNKED=0: the causal network's inference attention on the sequential kernel below (measurement aid)
static bool causal_chunked() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_CAUSAL_CHUNKED");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// ---- causal linear attention (ddsp/pcmer.py:170-188, `c: true`), inference ---------------------------------------------------
// out[n] = (q'_n . sum_{m<=n} k'_m (x) v_m) / (q'_n . (sum_{m<=n} k'_m + 1e-6)).  One workgroup per (utterance, head) walks
// the frames in order; thread t owns channel e = t & 63 of rows j = (t >> 6) + 4 i of the running 266 x 64 state (67 registers).
// `fast_transformers.CausalDotProduct` (the numerator) is a third-party CUDA extension that is not in the image: it is
// restated from its definition; the normaliser is the reference's own code.  Correct-first: one frame per step, two barriers.
__global__ void __launch_bounds__(256) causal_attention_kernel(const float* __restrict__ qf, const float* __restrict__ kf,
                                                               const float* __restrict__ v, int Fr, float* __restrict__ out) {
    constexpr int ROWS = (NF + 3) / 4;       // 67
    __shared__ float sq[LDF], sk[LDF], sv[DH], part[4 * DH], dpart[4];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int t = threadIdx.x, e = t & 63, r0 = t >> 6;
    float S[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) S[i] = 0.f;
    float ksum[2] = {0.f, 0.f};              // running key sums of features t and t + 256
    for (int n = 0; n < Fr; ++n) {
        const int64_t row8 = ((int64_t)b * Fr + n) * H + h;
        for (int j = t; j < LDF; j += 256) {
            sq[j] = qf[row8 * LDF + j];
            sk[j] = kf[row8 * LDF + j];
        }
        if (t < DH) sv[t] = v[((int64_t)b * Fr + n) * INNER + h * DH + t];
        __syncthreads();
        const float ve = sv[e];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int j = r0 + 4 * i;
            if (j < NF) {
                S[i] = fmaf(sk[j], ve, S[i]);
                acc = fmaf(sq[j], S[i], acc);
            }
        }
        part[r0 * DH + e] = acc;
        // denominator: q' . (cumulative k' + eps), features t and t + 256
        float d = 0.f;
        if (t < NF) {
            ksum[0] += sk[t];
            d = sq[t] * (ksum[0] + 1e-6f);
        }
        if (t + 256 < NF) {
            ksum[1] += sk[t + 256];
            d = fmaf(sq[t + 256], ksum[1] + 1e-6f, d);
        }
        d = wave_sum(d);
        if ((t & 63) == 0) dpart[t >> 6] = d;
        __syncthreads();
        if (t < DH) {
            const float num = (part[t] + part[DH + t]) + (part[2 * DH + t] + part[3 * DH + t]);
            const float den = (dpart[0] + dpart[1]) + (dpart[2] + dpart[3]);
            out[((int64_t)b * Fr + n) * INNER + h * DH + t] = num * (1.0f / den);
        }
    }
}

// ---- backward of the causal attention (training of `c: true` networks; correct first, one frame per step) ------------------
// With S_n = sum_{m<=n} k'_m (x) v_m, z_n = sum_{m<=n} k'_m, den_n = q'_n.(z_n + 1e-6), out_n = q'_n S_n / den_n and the upstream
// d_n = dL/d out_n:   dnum_n = d_n / den_n,   dden_n = -(d_n . out_n) / den_n,
//   dq'_n = S_n dnum_n + dden_n (z_n + 1e-6)                              (forward scan, pass Q)
//   dk'_n = G_n v_n + g_n,  dv_n = G_n^T k'_n   with G_n = sum_{m>=n} q'_m (x) dnum_m,  g_n = sum_{m>=n} dden_m q'_m   (reverse scans, passes K and V)
// Passes Q and K: 320 threads, thread j owns row j of the 266 x 64 state (64 registers), so the products with a 64-vector
// are in-thread; pass V uses the forward kernel's layout (thread (e, r0) owns column e of rows r0 + 4 i), in which the
// product with a 266-vector is in-thread.  Pass Q also leaves 1/den_n and dden_n for the other two.
__global__ void __launch_bounds__(320) causal_attn_bwd_q_kernel(const float* __restrict__ qf, const float* __restrict__ kf,
                                                                const float* __restrict__ v, const float* __restrict__ dout,
                                                                const float* __restrict__ out, int Fr, float* __restrict__ dqf,
                                                                float* __restrict__ dinv_out, float* __restrict__ dden_out) {
    __shared__ float sv[2][DH], sd[2][DH], so[2][DH], red[2][8];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int t = threadIdx.x;
    const bool active = t < NF;
    float S[DH];
#pragma unroll
    for (int e = 0; e < DH; ++e) S[e] = 0.f;
    float z = 0.f;
    for (int n = 0; n < Fr; ++n) {
        const int buf = n & 1;
        const int64_t row = (int64_t)b * Fr + n, row8 = row * H + h;
        if (t < DH) {
            sv[buf][t] = v[row * INNER + h * DH + t];
            sd[buf][t] = dout[row * INNER + h * DH + t];
            so[buf][t] = out[row * INNER + h * DH + t];
        }
        const float qj = active ? qf[row8 * LDF + t] : 0.f, kj = active ? kf[row8 * LDF + t] : 0.f;
        z += kj;
        const float dp = wave_sum(qj * (z + 1e-6f));
        if ((t & 63) == 0) red[buf][t >> 6] = dp;
        __syncthreads();
        const float den = ((red[buf][0] + red[buf][1]) + (red[buf][2] + red[buf][3])) + red[buf][4];
        const float dinv = 1.0f / den;
        float c = 0.f, acc = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) {
            c = fmaf(sd[buf][e], so[buf][e], c);
            S[e] = fmaf(kj, sv[buf][e], S[e]);
            acc = fmaf(S[e], sd[buf][e], acc);
        }
        const float dden = -c * dinv;
        if (active) dqf[row8 * LDF + t] = fmaf(dden, z + 1e-6f, acc * dinv);
        if (t == 0) {
            dinv_out[row8] = dinv;
            dden_out[row8] = dden;
        }
    }
}

__global__ void __launch_bounds__(320) causal_attn_bwd_k_kernel(const float* __restrict__ qf, const float* __restrict__ v,
                                                                const float* __restrict__ dout, const float* __restrict__ dinv_in,
                                                                const float* __restrict__ dden_in, int Fr,
                                                                float* __restrict__ dkf) {
    __shared__ float sv[2][DH], sd[2][DH];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int t = threadIdx.x;
    const bool active = t < NF;
    float G[DH];
#pragma unroll
    for (int e = 0; e < DH; ++e) G[e] = 0.f;
    float g = 0.f;
    for (int n = Fr - 1; n >= 0; --n) {
        const int buf = n & 1;
        const int64_t row = (int64_t)b * Fr + n, row8 = row * H + h;
        const float di = dinv_in[row8], dd = dden_in[row8];
        if (t < DH) {
            sv[buf][t] = v[row * INNER + h * DH + t];
            sd[buf][t] = dout[row * INNER + h * DH + t] * di;   // dnum
        }
        const float qj = active ? qf[row8 * LDF + t] : 0.f;
        g = fmaf(dd, qj, g);
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) {
            G[e] = fmaf(qj, sd[buf][e], G[e]);
            acc = fmaf(G[e], sv[buf][e], acc);
        }
        if (active) dkf[row8 * LDF + t] = acc + g;
    }
}

__global__ void __launch_bounds__(256) causal_attn_bwd_v_kernel(const float* __restrict__ qf, const float* __restrict__ kf,
                                                                const float* __restrict__ dout, const float* __restrict__ dinv_in,
                                                                int Fr, float* __restrict__ dv) {
    constexpr int ROWS = (NF + 3) / 4;       // 67
    __shared__ float sq[LDF], sk[LDF], sdn[DH], part[4 * DH];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int t = threadIdx.x, e = t & 63, r0 = t >> 6;
    float G[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) G[i] = 0.f;
    for (int n = Fr - 1; n >= 0; --n) {
        const int64_t row = (int64_t)b * Fr + n, row8 = row * H + h;
        for (int j = t; j < LDF; j += 256) {
            sq[j] = qf[row8 * LDF + j];
            sk[j] = kf[row8 * LDF + j];
        }
        if (t < DH) sdn[t] = dout[row * INNER + h * DH + t] * dinv_in[row8];
        __syncthreads();
        const float dne = sdn[e];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int j = r0 + 4 * i;
            if (j < NF) {
                G[i] = fmaf(sq[j], dne, G[i]);
                acc = fmaf(sk[j], G[i], acc);
            }
        }
        part[r0 * DH + e] = acc;
        __syncthreads();
        if (t < DH) dv[row * INNER + h * DH + t] = (part[t] + part[DH + t]) + (part[2 * DH + t] + part[3 * DH + t]);
    }
}

struct EpiAttnOut {  // out[(b*Fr+n)*512 + h*64 + e] = dinv[(b*Fr+n)*8+h] * acc   (z = b*8+h, m = n, col = e); dinv null -> 1
    float* out;
    const float* dinv;
    int Fr;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int z, int m, int e, float v, float) const {
        const int b = z / H, h = z % H;
        const int64_t row = (int64_t)b * Fr + m;
        out[row * INNER + h * DH + e] = dinv ? dinv[row * H + h] * v : v;
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
// sliding window of DW_RUN+30 inputs stay in registers (1.9 loads per output instead of 31); lanes walk
// channels, so every load/store of a wavefront is one contiguous 256-B row segment.  The taps are addressed as
// w[c*wsc + t*wst]: the forward pass reads a copy laid out [tap][channel] (wsc = 1, wst = 512; made by the weight
// preparation launch) so that the 31 tap loads are contiguous rows too - in the (512, 31) parameter layout every
// tap load of a wavefront touches 64 cache lines, which cost more than the convolution itself.
constexpr int DW_RUN = 32;
// SILU: apply SiLU (forward) and optionally keep the pre-activation; FLIP: correlate with reversed taps and no
// bias (the input-gradient of the same convolution).
template <bool SILU, bool FLIP, int RUN = DW_RUN>
__global__ void __launch_bounds__(256) dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int B, int Fr,
                                                     float* __restrict__ out, float* __restrict__ pre, int wsc, int wst,
                                                     int left, int split) {   // left = DWK / 2: centred taps; DWK - 1: causal taps (frames t-30 .. t)
    // split != 0 (forward, inference): `out` is written as bf16 hi/lo groups of 8 channels (A operand of the pw2 GEMM)
    const int c = blockIdx.x * 256 + threadIdx.x;       // channel (INNER = 512 -> 2 blocks in x)
    const int runs = (Fr + RUN - 1) / RUN;
    const int b = blockIdx.y / runs, f0 = (blockIdx.y % runs) * RUN;
    float wt[DWK];
#pragma unroll
    for (int t = 0; t < DWK; ++t) wt[t] = w[c * wsc + (FLIP ? DWK - 1 - t : t) * wst];
    const float* xb = x + ((int64_t)b * Fr) * INNER + c;
    float win[RUN + DWK - 1];
#pragma unroll
    for (int i = 0; i < RUN + DWK - 1; ++i) {
        const int f = f0 + i - left;
        win[i] = (f >= 0 && f < Fr) ? xb[(int64_t)f * INNER] : 0.f;
    }
    const float bi = FLIP ? 0.f : bias[c];
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
        float acc = bi;
#pragma unroll
        for (int t = 0; t < DWK; ++t) acc = fmaf(wt[t], win[o + t], acc);
        if (f0 + o < Fr) {
            const int64_t idx = ((int64_t)b * Fr + f0 + o) * INNER + c;
            if (SILU) {
                if (pre) pre[idx] = acc;
                const float y = acc * (1.0f / (1.0f + expf(-acc)));
                if (split)
                    ((uint32_t*)out)[idx] = ddsp_split1_group8(y, threadIdx.x & 63);
                else
                    out[idx] = y;
            } else {
                out[idx] = acc;
            }
        }
    }
}

// The same convolution + SiLU for the inference forward at large batches, two ADJACENT channels per thread (round 3).
// The one-channel kernel above is bound by vector-instruction issue, not by memory (r02_e_pmc_sq_synth.txt: 2300 instructions
// per wave for 32 outputs of 64 channels, VALU 0.62 at ~4 cycles per instruction, 24.5 us for 45 MB): 31 multiply-adds per
// output are its floor.  With a channel PAIR per thread the pair is the packed operand - taps, window and accumulator of the two
// channels sit in adjacent registers as they come from memory (8-byte loads), so every `v_pk_fma_f32` does two outputs and no
// register re-alignment is needed (the packed form ACROSS frames of round 2 needed moves for every other window position).
// Sigmoid by v_exp_f32 / v_rcp_f32 (~2 ulp).  Split output: the four lanes of a channel octet exchange their bf16 pairs so
// that each stores 8 contiguous bytes of the (8 hi | 8 lo) group.
typedef float f32x2_dw __attribute__((ext_vector_type(2)));
template <int RUN>
__global__ void __launch_bounds__(256, 3) dwconv_pair_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int B, int Fr,
                                                          float* __restrict__ out, int left, int split) {
    const int cp = blockIdx.x * 256 + threadIdx.x;      // channel pair (INNER / 2 = 256 pairs: one block in x)
    const int c = 2 * cp;
    const int runs = (Fr + RUN - 1) / RUN;
    const int b = blockIdx.y / runs, f0 = (blockIdx.y % runs) * RUN;
    f32x2_dw wt[DWK];
#pragma unroll
    for (int t = 0; t < DWK; ++t) wt[t] = *(const f32x2_dw*)(w + t * INNER + c);      // [tap][channel] copy
    const float* xb = x + ((int64_t)b * Fr) * INNER + c;
    f32x2_dw win[RUN + DWK - 1];
#pragma unroll
    for (int i = 0; i < RUN + DWK - 1; ++i) {
        const int f = f0 + i - left;
        win[i] = (f >= 0 && f < Fr) ? *(const f32x2_dw*)(xb + (int64_t)f * INNER) : f32x2_dw{0.f, 0.f};
    }
    const f32x2_dw bi = *(const f32x2_dw*)(bias + c);
    const int q = threadIdx.x & 3;                       // position in the channel octet
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
        f32x2_dw acc = bi;
#pragma unroll
        for (int t = 0; t < DWK; ++t) acc = __builtin_elementwise_fma(wt[t], win[o + t], acc);
        f32x2_dw y;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            y[e] = acc[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * acc[e]));
        const bool ok = f0 + o < Fr;
        float* dst = out + ((int64_t)b * Fr + f0 + o) * INNER + c;
        if (split) {
            // octet = lanes 4g..4g+3 (channels 8g..8g+7): lane q holds bf16 pairs h_q, l_q; the group is [h0 h1 h2 h3 | l0 l1 l2 l3]
            typedef __bf16 bf16x2_dw __attribute__((ext_vector_type(2)));
            const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(y, bf16x2_dw));
            const f32x2_dw rem = y - f32x2_dw{__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
            const uint32_t l = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, bf16x2_dw));
            // lane 0 stores (h0, h1), lane 1 (h2, h3), lane 2 (l0, l1), lane 3 (l2, l3)
            const int lane = threadIdx.x & 63, base = lane & ~3;
            const int s0 = base + 2 * (q & 1), s1 = s0 + 1;
            const uint32_t h0 = (uint32_t)__shfl((int)h, s0, 64), h1 = (uint32_t)__shfl((int)h, s1, 64);
            const uint32_t l0 = (uint32_t)__shfl((int)l, s0, 64), l1 = (uint32_t)__shfl((int)l, s1, 64);
            if (ok) {
                uint32_t* g = (uint32_t*)(out + ((int64_t)b * Fr + f0 + o) * INNER + (c & ~7)) + 2 * q;
                g[0] = q < 2 ? h0 : l0;
                g[1] = q < 2 ? h1 : l1;
            }
        } else if (ok) {
            *(f32x2_dw*)dst = y;
        }
    }
}

// The pair kernel above holds 46 frames of window and 31 taps per thread (168 registers: three waves per SIMD) and every wave runs
// load -> products -> store in lockstep with its round: 0.45 of its cycles wait.  LDS-tiled form (round 3): a workgroup stages
// 64 + 30 frames x 64 channels ONCE with 16-byte loads (1.47x read amplification instead of 2.9x, six loads per thread instead
// of 46) beside the 31 x 64 tap table, a thread makes 8 frames x 2 channels from a register window filled by 8-byte LDS reads;
// 158 registers (three waves per SIMD, as before - at four the compiler spills 40), 32 KB of LDS; the workgroups of a CU are in
// different phases.  Row-kernel family 0.081 -> 0.070 ms per step (21.5 -> 17.8 us per launch).
constexpr int DWT_F = 64, DWT_C = 64, DWT_RUN = 8;
__global__ void __launch_bounds__(256, 3) dwconv_tile_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B, int Fr,
                                                             float* __restrict__ out, int left, int split) {
    constexpr int ROWS = DWT_F + DWK - 1;
    __shared__ float tile[ROWS * DWT_C];
    __shared__ float taps[DWK * DWT_C];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * DWT_C;
    const int ftiles = (Fr + DWT_F - 1) / DWT_F;
    const int b = blockIdx.y / ftiles, f0 = (blockIdx.y % ftiles) * DWT_F;
    const float* xb = x + ((int64_t)b * Fr) * INNER + c0;
#pragma unroll
    for (int k = 0; k < (ROWS * (DWT_C / 4) + 255) / 256; ++k) {
        const int i = tid + 256 * k;
        if (i < ROWS * (DWT_C / 4)) {
            const int row = i / (DWT_C / 4), c4 = i % (DWT_C / 4);
            const int f = f0 + row - left;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f >= 0 && f < Fr) v = *(const f32x4*)(xb + (int64_t)f * INNER + 4 * c4);
            *(f32x4*)(tile + row * DWT_C + 4 * c4) = v;
        }
    }
    for (int i = tid; i < DWK * (DWT_C / 4); i += 256) {
        const int t = i / (DWT_C / 4), c4 = i % (DWT_C / 4);
        *(f32x4*)(taps + t * DWT_C + 4 * c4) = *(const f32x4*)(w + t * INNER + c0 + 4 * c4);      // [tap][channel] copy
    }
    __syncthreads();
    const int cp = tid & 31, fg = tid >> 5;               // channel pair of the tile, group of 8 frames
    const int c = c0 + 2 * cp, fo = DWT_RUN * fg;
    f32x2_dw win[DWT_RUN + DWK - 1];
#pragma unroll
    for (int i = 0; i < DWT_RUN + DWK - 1; ++i) win[i] = *(const f32x2_dw*)(tile + (fo + i) * DWT_C + 2 * cp);
    const f32x2_dw bi = *(const f32x2_dw*)(bias + c);
    f32x2_dw acc[DWT_RUN];
#pragma unroll
    for (int o = 0; o < DWT_RUN; ++o) acc[o] = bi;
#pragma unroll
    for (int t = 0; t < DWK; ++t) {
        const f32x2_dw wt = *(const f32x2_dw*)(taps + t * DWT_C + 2 * cp);
#pragma unroll
        for (int o = 0; o < DWT_RUN; ++o) acc[o] = __builtin_elementwise_fma(wt, win[o + t], acc[o]);
    }
    const int q = tid & 3;                                // position in the channel octet
#pragma unroll
    for (int o = 0; o < DWT_RUN; ++o) {
        f32x2_dw y;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            y[e] = acc[o][e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * acc[o][e]));
        const int f = f0 + fo + o;
        const bool ok = f < Fr;
        if (split) {
            typedef __bf16 bf16x2_dw __attribute__((ext_vector_type(2)));
            const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(y, bf16x2_dw));
            const f32x2_dw rem = y - f32x2_dw{__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
            const uint32_t l = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, bf16x2_dw));
            const int lane = tid & 63, base = lane & ~3;
            const int s0 = base + 2 * (q & 1), s1 = s0 + 1;
            const uint32_t h0 = (uint32_t)__shfl((int)h, s0, 64), h1 = (uint32_t)__shfl((int)h, s1, 64);
            const uint32_t l0 = (uint32_t)__shfl((int)l, s0, 64), l1 = (uint32_t)__shfl((int)l, s1, 64);
            if (ok) {
                uint32_t* g = (uint32_t*)(out + ((int64_t)b * Fr + f) * INNER + (c & ~7)) + 2 * q;
                g[0] = q < 2 ? h0 : l0;
                g[1] = q < 2 ? h1 : l1;
            }
        } else if (ok) {
            *(f32x2_dw*)(out + ((int64_t)b * Fr + f) * INNER + c) = y;
        }
    }
}

// =====================================================================================================
// Backward kernels (training).  Every contraction is again a GEMM on the fp32 matrix pipe (dX = dY W,
// dW = dY^T X as split-K batches + a reduction); the kernels below are the row-wise adjoints around them.
// =====================================================================================================

// LayerNorm backward, one wave per row: dx = rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)) (+ res);
// gx = dy*xhat is written for the column reduction that yields d gamma.
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, const float* __restrict__ res,
                                                            int64_t rows, float* __restrict__ dx,
                                                            float* __restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const f32x4 v = *(const f32x4*)(x + m * D + lane * 4);
    const f32x4 g = *(const f32x4*)(dy + m * D + lane * 4);
    const f32x4 ga = *(const f32x4*)(gamma + lane * 4);
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / D);
    f32x4 xh, dg;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xh[j] = v[j] - mean;
        ss = fmaf(xh[j], xh[j], ss);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / D) + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xh[j] *= rstd;
        dg[j] = g[j] * ga[j];
        s1 += dg[j];
        s2 = fmaf(dg[j], xh[j], s2);
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
    f32x4 o, gxo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = rstd * (dg[j] - s1 - xh[j] * s2);
        gxo[j] = g[j] * xh[j];
    }
    if (res) o += *(const f32x4*)(res + m * D + lane * 4);
    *(f32x4*)(dx + m * D + lane * 4) = o;
    *(f32x4*)(gx + m * D + lane * 4) = gxo;
}

// Column sums over rows, optionally weighted per row: out_partial[chunk][c] = sum_{r in chunk} X[r][c] * w(r).
// wmode: 0 none, 1 w = wsrc[r], 2 w = ln(1 + wsrc[r]/700), 3 w = wsrc[r]/pi     (the three side embeddings)
constexpr int CS_CHUNKS = 128;
// 16 bytes per lane (4 columns), four row lanes per block, two independent accumulators per lane: the variant for
// 16-byte-aligned matrices whose width is a multiple of 4 (every caller but the odd-width heads).  Round 2: the scalar
// kernel below ran at 0.9 TB/s (13 us for an 11 MB matrix, 54 launches per training step).
__global__ void __launch_bounds__(256) colsum_partial_v4_kernel(const float* __restrict__ X, int64_t ld, int64_t rows,
                                                                int cols, const float* __restrict__ wsrc, int wmode,
                                                                float* __restrict__ partial) {
    const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int rl = threadIdx.x >> 6;
    const int64_t per = (rows + CS_CHUNKS - 1) / CS_CHUNKS;
    const int64_t r0 = (int64_t)blockIdx.y * per;
    int64_t r1 = r0 + per;
    if (r1 > rows) r1 = rows;
    auto weight = [&](int64_t r) -> float {
        if (wmode == 1) return wsrc[r];
        if (wmode == 2) return logf(1.0f + __fdiv_rn(wsrc[r], 700.0f));
        if (wmode == 3) return __fdiv_rn(wsrc[r], 3.14159274101257324f);
        return 1.0f;
    };
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        int64_t r = r0 + rl;
        for (; r + 4 < r1; r += 8) {
            const f32x4 a = *(const f32x4*)(X + r * ld + c), b = *(const f32x4*)(X + (r + 4) * ld + c);
            const float wa = weight(r), wb = weight(r + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = fmaf(a[j], wa, s0[j]);
                s1[j] = fmaf(b[j], wb, s1[j]);
            }
        }
        if (r < r1) {
            const f32x4 a = *(const f32x4*)(X + r * ld + c);
            const float wa = weight(r);
#pragma unroll
            for (int j = 0; j < 4; ++j) s0[j] = fmaf(a[j], wa, s0[j]);
        }
    }
    __shared__ f32x4 red[256];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < cols)
        *(f32x4*)(partial + (int64_t)blockIdx.y * cols + c) =
            (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ X, int64_t ld, int64_t rows,
                                                             int cols, const float* __restrict__ wsrc, int wmode,
                                                             float* __restrict__ partial) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int64_t per = (rows + CS_CHUNKS - 1) / CS_CHUNKS;
    const int64_t r0 = (int64_t)blockIdx.y * per;
    int64_t r1 = r0 + per;
    if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (c < cols) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            float w = 1.0f;
            if (wmode == 1) w = wsrc[r];
            else if (wmode == 2) w = logf(1.0f + __fdiv_rn(wsrc[r], 700.0f));
            else if (wmode == 3) w = __fdiv_rn(wsrc[r], 3.14159274101257324f);
            s = fmaf(X[r * ld + c], w, s);
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < cols)
        partial[(int64_t)blockIdx.y * cols + c] = (red[threadIdx.x] + red[threadIdx.x + 64]) +
                                                  (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// Up to four column sums that share their shape in ONE pass (blockIdx.z = job): the two LayerNorm / GroupNorm parameter
// gradients (sums of gx and of dA), the three side-embedding weights and their bias (the same dX under four row weights).
struct ColsumJobs {
    const float* X[4];
    const float* wsrc[4];
    int wmode[4];
    float* out[4];
    int n;
};
__global__ void __launch_bounds__(256) colsum_multi_kernel(ColsumJobs jb, int64_t ld, int64_t rows, int cols,
                                                           float* __restrict__ partial) {
    const int job = blockIdx.z;
    const float* __restrict__ X = jb.X[job];
    const float* __restrict__ wsrc = jb.wsrc[job];
    const int wmode = jb.wmode[job];
    const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int rl = threadIdx.x >> 6;
    const int64_t per = (rows + CS_CHUNKS - 1) / CS_CHUNKS;
    const int64_t r0 = (int64_t)blockIdx.y * per;
    int64_t r1 = r0 + per;
    if (r1 > rows) r1 = rows;
    auto weight = [&](int64_t r) -> float {
        if (wmode == 1) return wsrc[r];
        if (wmode == 2) return logf(1.0f + __fdiv_rn(wsrc[r], 700.0f));
        if (wmode == 3) return __fdiv_rn(wsrc[r], 3.14159274101257324f);
        return 1.0f;
    };
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        int64_t r = r0 + rl;
        for (; r + 4 < r1; r += 8) {
            const f32x4 a = *(const f32x4*)(X + r * ld + c), b = *(const f32x4*)(X + (r + 4) * ld + c);
            const float wa = weight(r), wb = weight(r + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = fmaf(a[j], wa, s0[j]);
                s1[j] = fmaf(b[j], wb, s1[j]);
            }
        }
        if (r < r1) {
            const f32x4 a = *(const f32x4*)(X + r * ld + c);
            const float wa = weight(r);
#pragma unroll
            for (int j = 0; j < 4; ++j) s0[j] = fmaf(a[j], wa, s0[j]);
        }
    }
    __shared__ f32x4 red[256];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < cols)
        *(f32x4*)(partial + ((int64_t)job * CS_CHUNKS + blockIdx.y) * cols + c) =
            (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}
// the matching reduction: blockIdx.y = job, partial[job][chunk][cols] -> jb.out[job][cols]
__global__ void __launch_bounds__(256) reduce_multi_kernel(const float* __restrict__ partial, ColsumJobs jb, int cols) {
    const int job = blockIdx.y;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), zl = threadIdx.x >> 6;
    const float* p = partial + (int64_t)job * CS_CHUNKS * cols;
    float s0 = 0.f, s1 = 0.f;
    if (i < cols) {
#pragma unroll 4
        for (int z = zl; z < CS_CHUNKS; z += 8) {
            s0 += p[(int64_t)z * cols + i];
            s1 += p[(int64_t)(z + 4) * cols + i];
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (zl == 0 && i < cols)
        jb.out[job][i] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// out[i] = sum_{z < nz} partial[z][i]  (also the split-K reduction of the weight-gradient GEMMs)
// block = 64 elements x 4 z-lanes (launch with 256 threads, ceil(n / 64) blocks): a thread adds every fourth partial in
// two chains, the lanes meet in the LDS.  (One thread per element walked 128 partials alone: 10.5 us per call, 47 calls
// per training step.)
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partial, int nz, int64_t n,
                                                              float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int zl = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int z = zl;
        for (; z + 4 < nz; z += 8) {
            s0 += partial[(int64_t)z * n + i];
            s1 += partial[(int64_t)(z + 4) * n + i];
        }
        if (z < nz) s0 += partial[(int64_t)z * n + i];
    }
    __shared__ float red[256];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (zl == 0 && i < n)
        out[i] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

__global__ void __launch_bounds__(256) silu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dout,
                                                       int64_t n, float* __restrict__ dpre) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = pre[i], sg = 1.0f / (1.0f + expf(-p));
        dpre[i] = dout[i] * (sg * (1.0f + p * (1.0f - sg)));
    }
}

// GLU backward: g1 = [a | g] (rows x 1024), out = a*sigmoid(g): d_a = d*s, d_g = d*a*s*(1-s)
__global__ void __launch_bounds__(256) glu_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ dglu,
                                                      int64_t rows, float* __restrict__ dg1) {
    const int64_t total = rows * INNER;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / INNER;
        const int c = (int)(i % INNER);
        const float a = g1[m * 2 * INNER + c], g = g1[m * 2 * INNER + INNER + c], d = dglu[i];
        const float sg = 1.0f / (1.0f + expf(-g));
        dg1[m * 2 * INNER + c] = d * sg;
        dg1[m * 2 * INNER + INNER + c] = d * a * sg * (1.0f - sg);
    }
}

// depthwise-conv weight gradient dW[c][t] = sum_{b, f} dpre[b,f,c] * x[b, f+t-left, c] with the forward kernel's register window:
// one thread owns one channel for a run of DW_RUN frames, its DW_RUN upstream gradients and DW_RUN + 30 inputs stay in
// registers and every tap is a 32-long dot product of them (round 3; the round-2 kernel re-read 8 inputs per multiply-add
// batch through the L1 and took 79 us per layer at B = 32, memory-instruction bound).  partial[(b, run)][tap][channel]: coalesced 256-byte rows per tap; dw_wgrad_reduce_kernel sums the runs and
// transposes to the parameter's (512, 1, 31) layout.
__global__ void __launch_bounds__(256) dwconv_wgrad_run_kernel(const float* __restrict__ dpre, const float* __restrict__ x,
                                                               int B, int Fr, float* __restrict__ partial, int left) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int runs = (Fr + DW_RUN - 1) / DW_RUN;
    const int b = blockIdx.y / runs, f0 = (blockIdx.y % runs) * DW_RUN;
    const float* dp = dpre + ((int64_t)b * Fr) * INNER + c;
    const float* xp = x + ((int64_t)b * Fr) * INNER + c;
    float d[DW_RUN], win[DW_RUN + DWK - 1];
#pragma unroll
    for (int o = 0; o < DW_RUN; ++o) d[o] = f0 + o < Fr ? dp[(int64_t)(f0 + o) * INNER] : 0.f;
#pragma unroll
    for (int i = 0; i < DW_RUN + DWK - 1; ++i) {
        const int f = f0 + i - left;
        win[i] = (f >= 0 && f < Fr) ? xp[(int64_t)f * INNER] : 0.f;
    }
    float* out = partial + ((int64_t)blockIdx.y * DWK) * INNER + c;
#pragma unroll
    for (int t = 0; t < DWK; ++t) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int o = 0; o < DW_RUN; o += 2) {
            a0 = fmaf(d[o], win[o + t], a0);
            a1 = fmaf(d[o + 1], win[o + 1 + t], a1);
        }
        out[(int64_t)t * INNER] = a0 + a1;
    }
}
// dW[c][t] = sum_p partial[p][t][c]; block = 64 channels x 4 partial lanes
__global__ void __launch_bounds__(256) dw_wgrad_reduce_kernel(const float* __restrict__ partial, int np, float* __restrict__ dW) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), t = blockIdx.y, zl = threadIdx.x >> 6;
    float s0 = 0.f;
    for (int z = zl; z < np; z += 4) s0 += partial[((int64_t)z * DWK + t) * INNER + c];
    __shared__ float red[256];
    red[threadIdx.x] = s0;
    __syncthreads();
    if (zl == 0) dW[c * DWK + t] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// attention output adjoint, one wave per (frame, head) row: out = num * dinv  ->  d_num = d_out*dinv (in place),
// d_D = -(d_out . out) * dinv
__global__ void __launch_bounds__(256) attn_out_bwd_kernel(float* __restrict__ d_attn, const float* __restrict__ attn,
                                                           const float* __restrict__ dinv, int64_t rows8,
                                                           float* __restrict__ dD) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows8) return;
    const int64_t off = (r / H) * INNER + (r % H) * DH + lane;
    const float d = d_attn[off], o = attn[off], di = dinv[r];
    const float dot = wave_sum(d * o);
    d_attn[off] = d * di;
    if (lane == 0) dD[r] = -dot * di;
}

// d_ks[b,h,j] = sum_n q'[b,n,h,j] * d_D[b,n,h]   (block layout of key_sum_kernel)
__global__ void __launch_bounds__(KS_T * 8) weighted_key_sum_kernel(const float* __restrict__ qf, const float* __restrict__ dD,
                                                                    int Fr, float* __restrict__ dks) {
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int j4 = threadIdx.x % KS_T, fl = threadIdx.x / KS_T;
    const int64_t r0 = ((int64_t)b * Fr) * H + h;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    int n = fl;
    for (; n + 8 < Fr; n += 16) {
        const f32x4 a = *(const f32x4*)(qf + (r0 + (int64_t)n * H) * LDF + 4 * j4);
        const f32x4 c = *(const f32x4*)(qf + (r0 + (int64_t)(n + 8) * H) * LDF + 4 * j4);
        const float wa = dD[r0 + (int64_t)n * H], wc = dD[r0 + (int64_t)(n + 8) * H];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s0[e] = fmaf(a[e], wa, s0[e]);
            s1[e] = fmaf(c[e], wc, s1[e]);
        }
    }
    if (n < Fr) {
        const f32x4 a = *(const f32x4*)(qf + (r0 + (int64_t)n * H) * LDF + 4 * j4);
        const float wa = dD[r0 + (int64_t)n * H];
#pragma unroll
        for (int e = 0; e < 4; ++e) s0[e] = fmaf(a[e], wa, s0[e]);
    }
    __shared__ f32x4 red[KS_T * 8];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (fl == 0) {
        f32x4 t = red[j4];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += red[i * KS_T + j4];
        *(f32x4*)(dks + (int64_t)bh * LDF + 4 * j4) = t;
    }
}

// feature-map adjoint, in place on d_feat: d_feat <- dn * d(dd);  coef[r] = -dn^2 * sum_j d_feat_j * E_j
// key:   feat_j = r*exp(dd_j - diag + eps)            E_j = feat_j
// query: feat_j = r*(exp(dd_j - diag - max) + eps)    E_j = feat_j - r*eps, and the max-subtraction routes -sum to argmax
template <bool QUERY>
__global__ void __launch_bounds__(256) feature_map_bwd_kernel(const float* __restrict__ feat, float* __restrict__ dfeat,
                                                              int64_t rows8, float* __restrict__ coef) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows8) return;
    const float dn = 0.35355339059327373f, ratio = 0.06131393394849658f;
    const float* f = feat + r * LDF;
    float* d = dfeat + r * LDF;
    float dd[5], fv[5];
    float t = 0.f, best = -3.0e38f;
    int arg = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int j = lane + 64 * i;
        dd[i] = 0.f;
        fv[i] = 0.f;
        if (j < NF) {
            fv[i] = f[j];
            const float E = QUERY ? fv[i] - ratio * 1e-4f : fv[i];
            dd[i] = d[j] * E;
            t += dd[i];
            if (QUERY && (fv[i] > best)) {
                best = fv[i];
                arg = j;
            }
        }
    }
    t = wave_sum(t);
    if (QUERY) {
        // first index of the row maximum (torch.max returns one arg max; ties are measure-zero for real data)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) {
                best = ob;
                arg = oa;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int j = lane + 64 * i;
        if (j < NF) {
            float v = dd[i];
            if (QUERY && j == arg) v -= t;
            d[j] = dn * v;
        } else if (j < LDF) {
            d[j] = 0.f;
        }
    }
    if (lane == 0) coef[r] = -(dn * dn) * t;
}

// ---- fused adjoint of one feature-map side of the attention (round 3, split-bf16 arithmetic) ----
// What five launches did through HBM (EpiRowOuter product -> feature_map_bwd_kernel -> product with the projection), one wavefront
// does in registers for 16 frames of one (utterance, head):
//   S[j][n]  = sum_k mat[j][k] rows[n][k] + rowscale[n] colvec[j]        (j: 266 features, k: 64; 17 x 2 x 3 MFMA 16x16x32)
//   dd       = S * E(feat);  t[n] = sum_j dd;  d_feat = dn (dd - [j == argmax_j feat] t)   (query side; key side: no arg-max term)
//   out[n][d] = sum_j d_feat[j][n] P[j][d] - dn^2 t[n] src[n][d]         (9 x 4 x 3 MFMA)
// The first product is taken transposed (features on the M axis), so a lane owns one frame and 4 consecutive features per
// 16-feature block: the sums over features are in-lane plus two cross-lane steps, and those registers ARE the K operand of the
// second product when the projection is stored in the matching slot order (feat_proj_prep_kernel: slot (g, s) of k-step ks
// holds feature 32 ks + 4 g + s for s < 4 and 32 ks + 16 + 4 g + s - 4 above).  A linear map of a gradient: 3 split products.
constexpr int FB_KS = 9;                              // k-steps of the second product (266 features padded to 288)
constexpr int FB_PT_VEC = FB_KS * 4 * 64 * 2;         // 16-byte vectors of one prepared projection
__global__ void __launch_bounds__(256) feat_proj_prep_kernel(const float* __restrict__ P, ddsp_u32x4* __restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // (ks, blk, lane)
    if (idx >= FB_KS * 4 * 64) return;
    P += (int64_t)blockIdx.y * NF * DH;               // (blockIdx.y > 0: one (266, 64) matrix per (utterance, head) - d_ctx for the d_v product)
    dst += (int64_t)blockIdx.y * FB_PT_VEC;
    const int lane = idx & 63, blk = (idx >> 6) & 3, ks = idx >> 8;
    const int d = 16 * blk + (lane & 15), g = lane >> 4;
    float x[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int j = 32 * ks + (s < 4 ? 4 * g + s : 16 + 4 * g + s - 4);
        x[s] = j < NF ? P[(int64_t)j * DH + d] : 0.f;
    }
    ddsp_u32x4 hi, lo;
    ddsp_split8(x, hi, lo);
    dst[(idx >> 6) * 128 + lane] = hi;   // planar: 64 lanes of hi, then 64 lanes of lo (conflict-free 16-byte LDS reads)
    dst[(idx >> 6) * 128 + 64 + lane] = lo;
}
struct FeatBwdArgs {
    const float* rows;       // query side: d_num (M, 512);            key side: v (M, 512)
    const float* mat;        // query side: ctx (B*H, NF, DH);          key side: d_ctx
    const float* rowscale;   // query side: d_D (M8);                   key side: null (1)
    const float* colvec;     // query side: ks (B*H, LDF);              key side: d_ks
    const float* feat;       // q' / k' (M8, LDF)
    const ddsp_u32x4* pt;    // prepared projection
    const float* proj;       // query side: the projection as stored (NF, DH) - the arg-max term of the feature-map adjoint is the
                             // rank-1 correction -dn t P[arg] of the epilogue (exact fp32) instead of 68 compares per lane
    const float* src;        // q / k (M, 512)
    float* out;              // d_q / d_k (M, 512)
    int Fr;
    const ddsp_u32x4* mat_t; // key side: d_ctx of every (utterance, head) in the projection's layout, for d_v = k' d_ctx; or null
    float* out_v;            // d_v (M, 512)
    int ablate;              // timing experiments only (DDSP_FEAT_ABLATE): 1 no feature-row loads, 2 no first product, 4 no second product, 8 no staging
};
typedef __bf16 fb_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int FB_WAVES = 4;                                 // 16-frame tiles per workgroup
constexpr int FB_MAT_VEC = 17 * 2 * 64 * 2;                 // 16-byte vectors of the staged, split `mat` operand
// LDS: one region holds the split matrix during the first product and the prepared projection during the second (74 816 bytes:
// two workgroups per CU - at ~250 VGPRs a SIMD holds two wavefronts, one of each), then the column vector
constexpr int FB_REGION_VEC = FB_PT_VEC > FB_MAT_VEC ? FB_PT_VEC : FB_MAT_VEC;
constexpr int FB_LDS_BYTES = (FB_REGION_VEC + 68) * 16;
template <bool QUERY>
__global__ void __launch_bounds__(64 * FB_WAVES, 2) attn_feat_bwd_kernel(FeatBwdArgs a) {
    extern __shared__ ddsp_u32x4 fb_lds[];
    ddsp_u32x4* const mats = fb_lds;                  // [blk 17][k-half 2][hi | lo][lane 64]: the A operand of the first product
    ddsp_u32x4* const pts = fb_lds;                   // later: the prepared projection, as stored
    f32x4* const cvs = reinterpret_cast<f32x4*>(fb_lds + FB_REGION_VEC);   // colvec (LDF = 268 floats = 67 vectors)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    // blockIdx.x = (utterance, head), blockIdx.y = 64-frame block: workgroups are dealt to the XCDs round-robin in x-major
    // order, so the full blocks come first and the short tail blocks (172 frames: 4 + 4 + 3 tiles) last
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int f0 = (blockIdx.y * FB_WAVES + wave) * 16;
    const bool active = f0 < a.Fr;                    // (idle wavefronts of a tail block still stage and meet the barriers)
    const bool live = f0 + n < a.Fr;
    const int f = live ? f0 + n : a.Fr - 1;
    const int64_t row = ((int64_t)b * a.Fr + f) * H + h;
    const float dn = 0.35355339059327373f, ratio = 0.06131393394849658f;

    {   // stage: every thread splits its share of the (266, 64) matrix once for the workgroup
        const float* mat = a.mat + (int64_t)bh * NF * DH;
        constexpr int NT = 64 * FB_WAVES, MAT_IT = (17 * 2 * 64 + NT - 1) / NT;
        f32x4 mu[MAT_IT], mv[MAT_IT];
#pragma unroll
        for (int i = 0; i < MAT_IT; ++i) {   // all loads of the stage are issued before the first use
            const int e = threadIdx.x + i * NT;
            const int el = e & 63, kh = (e >> 6) & 1, blk = e >> 7;
            const int j = 16 * blk + (el & 15);
            mu[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            mv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < 17 * 2 * 64 && j < NF && !(a.ablate & 8)) {
                const float* p = mat + (int64_t)j * DH + 32 * kh + 8 * (el >> 4);
                mu[i] = *(const f32x4*)p;
                mv[i] = *(const f32x4*)(p + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < MAT_IT; ++i) {
            const int e = threadIdx.x + i * NT;
            if (e < 17 * 2 * 64) {
                const float x[8] = {mu[i][0], mu[i][1], mu[i][2], mu[i][3], mv[i][0], mv[i][1], mv[i][2], mv[i][3]};
                ddsp_u32x4 hi, lo;
                ddsp_split8(x, hi, lo);
                mats[(e >> 6) * 128 + (e & 63)] = hi;
                mats[(e >> 6) * 128 + 64 + (e & 63)] = lo;
            }
        }
        if (threadIdx.x < LDF / 4) cvs[threadIdx.x] = *(const f32x4*)(a.colvec + (int64_t)bh * LDF + 4 * threadIdx.x);
    }
    fb_bf16x8 xh[2], xl[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const float* p = a.rows + row * DH + 32 * kh + 8 * g;
        const f32x4 u = *(const f32x4*)p, v = *(const f32x4*)(p + 4);
        const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        ddsp_u32x4 hi, lo;
        ddsp_split8(x, hi, lo);
        xh[kh] = __builtin_bit_cast(fb_bf16x8, hi);
        xl[kh] = __builtin_bit_cast(fb_bf16x8, lo);
    }
    // this lane's slice of the feature row (17 x 16 bytes, the only large HBM stream of the kernel), the row scale and the
    // epilogue's source row: in flight while the first product runs
    const float* fr = a.feat + row * LDF;
    f32x4 fvv[17];
#pragma unroll
    for (int blk = 0; blk < 17; ++blk) {
        fvv[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (16 * blk + 4 * g < NF && !(a.ablate & 1)) fvv[blk] = *(const f32x4*)(fr + 16 * blk + 4 * g);   // (rows are LDF = 268 floats: the last group that holds a feature is 264..267)
    }
    const float rs = QUERY ? a.rowscale[row] : 1.0f;
    f32x4 s4[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) s4[blk] = *(const f32x4*)(a.src + row * DH + 16 * blk + 4 * g);
    __syncthreads();
    f32x4 S[17];
#pragma unroll
    for (int blk = 0; blk < 17; ++blk) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (active && !(a.ablate & 2)) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const ddsp_u32x4* p = mats + (blk * 2 + kh) * 128 + lane;
                const fb_bf16x8 mh = __builtin_bit_cast(fb_bf16x8, p[0]), ml = __builtin_bit_cast(fb_bf16x8, p[64]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ml, xh[kh], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mh, xl[kh], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mh, xh[kh], acc, 0, 0, 0);
            }
        }
        S[blk] = acc;
    }
    __syncthreads();   // every wavefront is done with the matrix
    if (!QUERY && a.mat_t) {
        // key side, d_v[n][d] = sum_j k'[n][j] d_ctx[j][d]: the lane's slice of k' is already the K operand (same slot order as
        // d_feat below), d_ctx^T in the projection's layout (one launch of feat_proj_prep_kernel per layer) takes the region first
        const ddsp_u32x4* src_t = a.mat_t + (int64_t)bh * FB_PT_VEC;
#pragma unroll
        for (int i = 0; i < FB_PT_VEC / 64 / FB_WAVES; ++i) {
            const int piece = wave + FB_WAVES * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_t + 64 * piece + lane),
                                             (__attribute__((address_space(3))) void*)(pts + 64 * piece), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 o3[4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) o3[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active) {
#pragma unroll
            for (int ks = 0; ks < FB_KS; ++ks) {
                float y[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[r] = 32 * ks + 4 * g + r < NF ? fvv[2 * ks][r] : 0.f;
                    y[4 + r] = (2 * ks + 1 < 17 && 32 * ks + 16 + 4 * g + r < NF) ? fvv[2 * ks + 1 < 17 ? 2 * ks + 1 : 16][r] : 0.f;
                }
                ddsp_u32x4 hi, lo;
                ddsp_split8(y, hi, lo);
                const fb_bf16x8 yh = __builtin_bit_cast(fb_bf16x8, hi), yl = __builtin_bit_cast(fb_bf16x8, lo);
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const ddsp_u32x4* p = pts + (ks * 4 + blk) * 128 + lane;
                    const fb_bf16x8 ph = __builtin_bit_cast(fb_bf16x8, p[0]), pl = __builtin_bit_cast(fb_bf16x8, p[64]);
                    o3[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, yh, o3[blk], 0, 0, 0);
                    o3[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, yl, o3[blk], 0, 0, 0);
                    o3[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, yh, o3[blk], 0, 0, 0);
                }
            }
            if (live) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) *(f32x4*)(a.out_v + row * DH + 16 * blk + 4 * g) = o3[blk];
            }
        }
        __syncthreads();   // done with d_ctx^T
    }
    // the projection streams into the same region (no registers)
#pragma unroll
    for (int i = 0; i < FB_PT_VEC / 64 / FB_WAVES; ++i) {
        const int piece = wave + FB_WAVES * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.pt + 64 * piece + lane),
                                         (__attribute__((address_space(3))) void*)(pts + 64 * piece), 16, 0, 0);
    }
    static_assert(FB_PT_VEC % (64 * FB_WAVES) == 0, "whole 1 KiB pieces per wavefront");
    // feature-map adjoint on the registers (feature of S[blk][r]: 16 blk + 4 g + r) while the projection arrives
    float t = 0.f, best = -3.0e38f;
    int arg = 0x7fffffff;
#pragma unroll
    for (int blk = 0; blk < 17; ++blk) {
        const int j0 = 16 * blk + 4 * g;
        const f32x4 fv = fvv[blk];
        f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
        if (j0 < NF) c4 = cvs[4 * blk + g];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = blk < 16 || j0 + r < NF;   // (only the last block holds pad features)
            const float E = QUERY ? fv[r] - ratio * 1e-4f : fv[r];
            const float dd = valid ? fmaf(rs, c4[r], S[blk][r]) * E : 0.f;
            S[blk][r] = dd;
            t += dd;
            if (QUERY && valid && fv[r] > best) {
                best = fv[r];
                arg = j0 + r;
            }
        }
    }
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (QUERY) {   // first index of the row maximum, like feature_map_bwd_kernel
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) {
                best = ob;
                arg = oa;
            }
        }
    }
    f32x4 pa[4];
    if (QUERY) {   // row `arg` of the projection (arg is the same in the four lanes of a frame), in flight under the second product
        const int ja = arg < NF ? arg : 0;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) pa[blk] = *(const f32x4*)(a.proj + (int64_t)ja * DH + 16 * blk + 4 * g);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) return;
    f32x4 o4[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) o4[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < FB_KS; ++ks) {
        float y[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y[r] = dn * S[2 * ks][r];
            y[4 + r] = 2 * ks + 1 < 17 ? dn * S[2 * ks + 1 < 17 ? 2 * ks + 1 : 16][r] : 0.f;
        }
        ddsp_u32x4 hi, lo;
        ddsp_split8(y, hi, lo);
        const fb_bf16x8 yh = __builtin_bit_cast(fb_bf16x8, hi), yl = __builtin_bit_cast(fb_bf16x8, lo);
        if (a.ablate & 4) continue;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const ddsp_u32x4* p = pts + (ks * 4 + blk) * 128 + lane;
            const fb_bf16x8 ph = __builtin_bit_cast(fb_bf16x8, p[0]), pl = __builtin_bit_cast(fb_bf16x8, p[64]);
            o4[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, yh, o4[blk], 0, 0, 0);
            o4[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, yl, o4[blk], 0, 0, 0);
            o4[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, yh, o4[blk], 0, 0, 0);
        }
    }
    if (!live) return;
    const float coef = -(dn * dn) * t;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int64_t off = row * DH + 16 * blk + 4 * g;
        f32x4 r4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            r4[r] = fmaf(coef, s4[blk][r], o4[blk][r]);
            if (QUERY) r4[r] = fmaf(arg < NF ? -dn * t : 0.f, pa[blk][r], r4[r]);
        }
        *(f32x4*)(a.out + off) = r4;
    }
}

struct EpiRowOuter {  // out[(b*Fr+m)*8+h][j] = acc + rowscale[row] * colvec[(b*8+h)][j]   (z = b*8+h); rowscale null -> 1
    float* out;
    const float* rowscale;
    const float* colvec;
    int Fr;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int z, int m, int j, float v, float) const {
        const int b = z / H, h = z % H;
        const int64_t row = ((int64_t)b * Fr + m) * H + h;
        const float rs = rowscale ? rowscale[row] : 1.0f;
        out[row * LDF + j] = fmaf(rs, colvec[(int64_t)z * LDF + j], v);
    }
};

struct EpiAxpyRow {  // C[r][d] = acc + coef[r] * src[r][d]
    float* C;
    const float* coef;
    const float* src;
    int ld;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int, int m, int n, float v, float) const {
        const int64_t o = (int64_t)m * ld + n;
        C[o] = fmaf(coef[m], src[o], v);
    }
};

struct EpiAccumulate {  // C += acc
    float* C;
    int64_t ldc;
    __device__ __forceinline__ float col(int) const { return 0.f; }
    __device__ __forceinline__ void operator()(int, int m, int n, float v, float) const { C[(int64_t)m * ldc + n] += v; }
    static constexpr bool kStore4 = true;
    __device__ __forceinline__ bool vec_ok() const { return ((uintptr_t)C % 16) == 0 && ldc % 4 == 0; }
    __device__ __forceinline__ void store4(int, int m, int n, f32x4 v) const {
        f32x4* p = (f32x4*)(C + (int64_t)m * ldc + n);
        *p = *p + v;
    }
};

// GroupNorm(4) + LeakyReLU backward.  Pass 1 (per utterance, group): s1 = sum dy*g, s2 = sum dy*g*xhat over 64 ch x Fr.
__global__ void __launch_bounds__(256) groupnorm_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const float* __restrict__ dy, const float* __restrict__ stats,
                                                                  const float* __restrict__ gamma, int Fr,
                                                                  float* __restrict__ bstats) {
    const int g = blockIdx.x, b = blockIdx.y;
    const int c = threadIdx.x & 63, fl = threadIdx.x >> 6;
    const int ch = g * 64 + c;
    const float mean = stats[(b * 4 + g) * 2], rstd = stats[(b * 4 + g) * 2 + 1];
    const float ga = gamma[ch];
    double s1 = 0.0, s2 = 0.0;
    for (int f = fl; f < Fr; f += 4) {
        const int64_t i = ((int64_t)b * Fr + f) * D + ch;
        const float slope = y[i] > 0.f ? 1.0f : 0.01f;   // y = lrelu(gn(x)); sign(y) = sign(gn(x))
        const float dgn = dy[i] * slope;
        const float xh = (x[i] - mean) * rstd;
        s1 += (double)(dgn * ga);
        s2 += (double)(dgn * ga * xh);
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) {
        red[fl] = s1;
        red[4 + fl] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = 64.0 * Fr;
        bstats[(b * 4 + g) * 2 + 0] = (float)((red[0] + red[1] + red[2] + red[3]) / n);
        bstats[(b * 4 + g) * 2 + 1] = (float)((red[4] + red[5] + red[6] + red[7]) / n);
    }
}

// Pass 2: dx = rstd*(dgn*gamma - m1 - xhat*m2);  also gxh = dgn*xhat and dgn itself (for d gamma / d beta column sums)
__global__ void __launch_bounds__(256) groupnorm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const float* __restrict__ dy, const float* __restrict__ stats,
                                                                  const float* __restrict__ bstats,
                                                                  const float* __restrict__ gamma, int64_t rows, int Fr,
                                                                  float* __restrict__ dx, float* __restrict__ gxh,
                                                                  float* __restrict__ dgn_out) {
    const int64_t total = rows * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / D;
        const int ch = (int)(i % D);
        const int b = (int)(m / Fr), g = ch >> 6;
        const float mean = stats[(b * 4 + g) * 2], rstd = stats[(b * 4 + g) * 2 + 1];
        const float m1 = bstats[(b * 4 + g) * 2], m2 = bstats[(b * 4 + g) * 2 + 1];
        const float slope = y[i] > 0.f ? 1.0f : 0.01f;
        const float dgn = dy[i] * slope;
        const float xh = (x[i] - mean) * rstd;
        dx[i] = rstd * (dgn * gamma[ch] - m1 - xh * m2);
        gxh[i] = dgn * xh;
        dgn_out[i] = dgn;
    }
}

// shifted copy over the frame axis with zero fill at utterance edges: out[b,f,:] = x[b,f+shift,:]
__global__ void __launch_bounds__(256) shift_rows_kernel(const float* __restrict__ x, int64_t rows, int Fr, int C, int shift,
                                                         float* __restrict__ out) {
    const int64_t total = rows * (C / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / (C / 4);
        const int c4 = (int)(i % (C / 4)) * 4;
        const int f = (int)(m % Fr) + shift;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (f >= 0 && f < Fr) v = *(const f32x4*)(x + (m + shift) * C + c4);
        *(f32x4*)(out + m * C + c4) = v;
    }
}

// packed (Cout, 3, Cin) gradient -> torch layout (Cout, Cin, 3)
__global__ void unpack_conv3_kernel(const float* __restrict__ packed, int Cout, int Cin, float* __restrict__ out) {
    const int64_t total = (int64_t)Cout * Cin * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 3);
        const int c = (int)((i / 3) % Cin);
        const int o = (int)(i / (3 * Cin));
        out[i] = packed[(int64_t)o * 3 * Cin + (int64_t)tap * Cin + c];
    }
}

// conv weight (Cout, Cin, 3) -> transposed-conv operand (Cin, 3*Cout): out[c][tap*Cout + o] = w[o][c][2 - tap]
__global__ void pack_conv3_transposed_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ out) {
    const int64_t total = (int64_t)Cout * Cin * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 3);
        const int c = (int)((i / 3) % Cin);
        const int o = (int)(i / (3 * Cin));
        out[(int64_t)c * 3 * Cout + (int64_t)(2 - tap) * Cout + o] = w[i];
    }
}

// weight-norm backward, one wave per output row: W = g*v/|v|  ->  d_g = (dW . v)/|v|,  d_v = g/|v| * (dW - (dW.vhat) vhat)
__global__ void __launch_bounds__(256) weight_norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                              const float* __restrict__ dW, int n_out, int n_in,
                                                              float* __restrict__ dg, float* __restrict__ dv) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n_out) return;
    const float* vr = v + (int64_t)o * n_in;
    const float* dr = dW + (int64_t)o * n_in;
    float ss = 0.f, dot = 0.f;
    for (int i = lane; i < n_in; i += 64) {
        ss = fmaf(vr[i], vr[i], ss);
        dot = fmaf(dr[i], vr[i], dot);
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float nrm = sqrtf(ss);
    if (lane == 0) dg[o] = dot / nrm;
    const float s = g[o] / nrm, proj = dot / ss;
    for (int i = lane; i < n_in; i += 64) dv[(int64_t)o * n_in + i] = s * (dr[i] - proj * vr[i]);
}

// Gradient of the speaker table, deterministic (no float atomics: the training trajectory is chaotic enough without a
// run-to-run difference in the last bit).  Pass 1: usum[b][c] = sum_f dx[b,f,c], one block per utterance, 64 lanes x 16
// bytes across the channels, 16 frame lanes; it also reports an id outside [1, n_spk] (the row is then skipped, like the
// forward's table read).  Pass 2: one block per table row adds the utterances that use it in ascending order.
__global__ void __launch_bounds__(1024) utterance_sum_kernel(const float* __restrict__ dx, int Fr,
                                                             const int64_t* __restrict__ spk_id, int64_t n_spk_id, int n_spk,
                                                             float* __restrict__ usum, int* __restrict__ err) {
    const int64_t b = blockIdx.x;
    const int c4 = threadIdx.x & 63, fl = threadIdx.x >> 6;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int f = fl; f < Fr; f += 16) s += *(const f32x4*)(dx + ((int64_t)b * Fr + f) * D + 4 * c4);
    __shared__ f32x4 red[1024];
    red[threadIdx.x] = s;
    __syncthreads();
    if (fl == 0) {
        f32x4 t = red[c4];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += red[64 * i + c4];
        *(f32x4*)(usum + b * D + 4 * c4) = t;
    }
    if (threadIdx.x == 0 && spk_id) {
        const int64_t id = spk_id[n_spk_id == 1 ? 0 : b];
        if (id < 1 || id > n_spk) __hip_atomic_store(err, DDSP_DEV_ERR_SPK_ID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void __launch_bounds__(256) spk_table_grad_kernel(const float* __restrict__ usum, int64_t B,
                                                             const int64_t* __restrict__ spk_id, int64_t n_spk_id,
                                                             MixArgs mix, float* __restrict__ dtable) {
    const int r = blockIdx.x, c = threadIdx.x;  // D == 256 channels
    float s = 0.f;
    if (mix.n > 0) {
        float cr = 0.f;
        for (int k = 0; k < mix.n; ++k)
            if (mix.ids[k] - 1 == r) cr += mix.w[k];
        if (cr != 0.f) {
            for (int64_t b = 0; b < B; ++b) s += usum[b * D + c];
            s *= cr;
        }
    } else if (n_spk_id == 1) {
        if (spk_id[0] - 1 == r)
            for (int64_t b = 0; b < B; ++b) s += usum[b * D + c];
    } else {
        for (int64_t b = 0; b < B; ++b)
            if (spk_id[b] - 1 == r) s += usum[b * D + c];
    }
    dtable[(int64_t)r * D + c] = s;
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

// ---- buffers of one forward pass --------------------------------------------------------------------------------
// Inference aliases the three layers onto one set and updates the residual stream in place; training keeps every
// activation the adjoints need (about 38 KB per frame and layer).
struct LayerBufs {
    float *x_in, *y, *q, *k, *v, *qf, *kf, *ks, *cx, *dinv, *attn, *x_mid, *y2, *g1, *glu, *pre, *dwo, *x_out;
};
struct U2CBufs {
    float *w1, *w2, *wh, *wqkv, *bqkv, *wglu, *bglu, *wdw, *wout, *wpw2, *p3, *t1, *t2, *gst, *y_final, *kpart;
    LayerBufs l[3];
    // the weight buffers live in the context's prepared-weight slot: what it holds (bit 0 prepared, bit 1 with split copies, bit 2
    // in fused-GLU order, bit 3 attention pieces), updated by the forward that prepares; null: prepare on every call
    int* wstate = nullptr;
};

struct Arena {  // sizes first (dry run), then pointers
    ddsp_ctx* ctx;
    bool dry;
    size_t total;
    int rc;
    char* ext = nullptr;   // a caller-owned region instead of the context's scratch (kept activations of a training step)
    size_t ext_cap = 0;
    float* get(size_t n_floats) {
        const size_t bytes = ((n_floats * sizeof(float) + 255) & ~(size_t)255) + 256;
        if (dry) {
            total += bytes;
            return nullptr;
        }
        if (ext) {
            if (total + bytes > ext_cap) {
                rc = DDSP_ERR_ARG;
                return nullptr;
            }
            float* p = reinterpret_cast<float*>(ext + total);
            total += bytes;
            return p;
        }
        void* p = nullptr;
        const int r = ddsp_scratch_get(ctx, n_floats * sizeof(float), &p);
        if (r) rc = r;
        return (float*)p;
    }
};

// the prepared weights (u2c_prepare_kernel, performer_p3): from the arena, or (cached != null) from the context's slot
static void plan_weights(Arena& a, U2CBufs& bf, const ddsp_u2c_weights& w) {
    bf.w1 = a.get((size_t)D * 3 * w.n_unit);
    bf.w2 = a.get((size_t)D * 3 * D);
    bf.wh = a.get((size_t)w.n_out * D);
    bf.wqkv = a.get((size_t)3 * 3 * INNER * D);
    bf.bqkv = a.get((size_t)3 * 3 * INNER);
    bf.wglu = a.get((size_t)3 * 2 * INNER * D);    // pw1 re-ordered for the fused GLU epilogue (inference)
    bf.bglu = a.get((size_t)3 * 2 * INNER);
    bf.wdw = a.get((size_t)3 * DWK * INNER);         // depthwise taps as [tap][channel]
    bf.wout = a.get((size_t)3 * D * INNER);          // out-projection / pw2 weights as bf16 hi/lo groups (split GEMMs)
    bf.wpw2 = a.get((size_t)3 * D * INNER);
    bf.p3 = a.get((size_t)3 * PERFORMER_P3_BYTES / 4);  // projection matrices as bf16 pieces (split-bf16 attention)
}

static void plan_forward(Arena& a, U2CBufs& bf, const ddsp_u2c_weights& w, int64_t B, int64_t Fr, bool keep, bool weights_cached = false) {
    const size_t M = (size_t)(B * Fr), M8 = M * H;
    if (!weights_cached) plan_weights(a, bf, w);
    bf.t1 = a.get(M * D);
    bf.t2 = a.get(M * D);
    bf.gst = a.get((size_t)B * 4 * 2);
    bf.y_final = a.get(M * D);
    bf.kpart = a.get((size_t)KS_SPLITS * (M <= KS_MAX_ROWS ? M : 1) * D);   // K-split partial products (small batches only)
    auto one = [&](LayerBufs& L, float* x_in) {
        L.x_in = x_in;
        L.y = a.get(M * D);
        L.q = a.get(M * INNER);
        L.k = a.get(M * INNER);
        L.v = a.get(M * INNER);
        L.qf = a.get(M8 * LDF);
        L.kf = a.get(M8 * LDF);
        L.ks = a.get((size_t)B * H * PERFORMER_KS_STRIDE);           // (the fused inference kernels pad features to 272)
        L.cx = a.get((size_t)B * H * (PERFORMER_CTXS_FLOATS > PERFORMER_LDJ * DH ? PERFORMER_CTXS_FLOATS : PERFORMER_LDJ * DH));
        L.dinv = a.get(M8);
        L.attn = a.get(M * INNER);
        L.g1 = a.get(M * 2 * INNER);
        if (keep) {
            L.x_mid = a.get(M * D);
            L.y2 = a.get(M * D);
            L.glu = a.get(M * INNER);
            L.pre = a.get(M * INNER);
            L.dwo = a.get(M * INNER);
            L.x_out = a.get(M * D);
        } else {
            L.x_mid = L.x_in;  // residuals in place
            L.x_out = L.x_in;
            L.y2 = L.y;
            L.glu = L.q;       // q / k are dead once the attention output exists
            L.dwo = L.k;
            L.pre = nullptr;
        }
    };
    float* x0 = a.get(M * D);
    one(bf.l[0], x0);
    if (keep) {
        one(bf.l[1], bf.l[0].x_out);
        one(bf.l[2], bf.l[1].x_out);
    } else {
        bf.l[1] = bf.l[0];
        bf.l[2] = bf.l[0];
    }
}

struct U2CInputs {
    const float *units, *f0, *phase, *volume;
    const int64_t* spk_id;
    int64_t n_spk_id;
    MixArgs mix;
    int64_t B, Fr;
};

static int u2c_forward(ddsp_ctx* ctx, hipStream_t st, const ddsp_u2c_weights& w, const U2CInputs& in, U2CBufs& bf,
                       float* ctrl) {
    const int64_t B = in.B, Fr = in.Fr, M = B * Fr, M8 = M * H;
    const int iM = (int)M;
    const size_t n_w1 = (size_t)D * 3 * w.n_unit, n_w2 = (size_t)D * 3 * D, n_wh = (size_t)w.n_out * D;
    // product arithmetic of the Linear / conv GEMMs (gemm::Args::math): inference uses the split-bf16 mode, the
    // training forward (activations kept for the backward pass) stays on fp32 MFMA like the backward GEMMs
    // (ctx->math 4 = split-bf16 products with every operand split inside the GEMM loops: the pre-round-2 path, kept as a
    // measurement / bit-identity aid)
    const int lin_math = bf.l[0].pre ? 0 : (ctx->math == 4 ? DDSP_MATH_SPLIT_BF16 : ctx->math);
    const float* zero_page = nullptr;   // source of the conv taps that fall off an utterance (LDS-DMA conv GEMM)
    if (int rc = ddsp_zero_page(ctx, &zero_page)) return rc;
    int* dev_err = nullptr;
    if (int rc = ddsp_dev_error_ptr(ctx, &dev_err)) return rc;
    // Inference, and enough rows that the Linear layers run the 128x128 DMA tile anyway: GLU is formed inside the pw1
    // GEMM (half the store, no glu kernel).  Training keeps pw1's raw output for the backward pass; small batches keep
    // the tile shapes that suit them and the separate glu kernel.
    // (round 3: at smaller batches too, on 64x128 tiles of 4 waves - one launch and one round trip of the 2 x 512-wide pw1
    // output fewer per layer; DDSP_GLU_SMALL=0 restores the separate glu kernel below 8065 rows, measurement aid)
    static int glu_small = -1;
    if (glu_small < 0) {
        const char* e = getenv("DDSP_GLU_SMALL");
        glu_small = (e && e[0] == '0') ? 0 : 1;
    }
    const bool glu_large = !bf.l[0].pre && (int64_t)((M + 127) / 128) * (2 * INNER / 128) >= 512;
    bool fuse_glu = !bf.l[0].pre && (glu_large || glu_small);
    for (int l = 0; l < 3; ++l)
        fuse_glu = fuse_glu && ((uintptr_t)w.layer[l].cm_pw1_w % 16) == 0;
    // Split-bf16 products at a size where every GEMM of the network runs the LDS-DMA kernel: nothing is split inside the
    // GEMM loops.  The weights are packed as bf16 hi/lo groups by the preparation launch (B_split) and the producers of
    // the A operands (GroupNorm+LeakyReLU, the LayerNorms, the attention kernel) write them in that layout (A_split);
    // conv1 reads the caller's fp32 units: it alone still splits A in the kernel.
    // (presplit_w: the weights alone, at any batch size - every GEMM of the inference forward runs the DMA kernel now, which
    // then splits only its A operand in the loop (mode 7); presplit: the activations too, from the size at which the fused-GLU
    // tiling is used)
    const bool presplit_w = lin_math == DDSP_MATH_SPLIT_BF16 && ctx->math != 4 && w.n_unit % 32 == 0 &&
                            w.n_unit + 32 <= DDSP_ZERO_FLOATS && w.n_out >= 256 && ((uintptr_t)in.units % 16) == 0;
    static int64_t presplit_min_rows = -1;   // DDSP_U2C_PRESPLIT_MIN_ROWS: measurement aid
    if (presplit_min_rows < 0) {
        const char* e = getenv("DDSP_U2C_PRESPLIT_MIN_ROWS");
        presplit_min_rows = e ? atoll(e) : 8192;
    }
    const bool presplit = presplit_w && fuse_glu && glu_large && M >= presplit_min_rows;
    static int64_t attn_bf16_min = -1;   // DDSP_ATTN_BF16_MIN: (utterance, head) pairs from which the split-bf16 attention runs
    if (attn_bf16_min < 0) {
        const char* e = getenv("DDSP_ATTN_BF16_MIN");
        attn_bf16_min = e ? atoll(e) : 256;
    }
    const bool attn_bf16 = !bf.l[0].pre && lin_math == DDSP_MATH_SPLIT_BF16 && B * H >= attn_bf16_min && !w.causal;
    const int asplit = presplit ? 1 : 0;
    // A handful of rows (the real-time block: 87): the N = 256, K = 512 residual GEMMs (out-projection, pw2) would run on 8
    // workgroups walking 16 k-steps each.  Their K range is cut over KS_SPLITS workgroups per tile instead (32-96 workgroups,
    // 4 k-steps each, partial products stored) and the sum + bias + residual is formed by the LayerNorm that follows.
    static int ksplit_on = -1;
    if (ksplit_on < 0) {
        const char* e = getenv("DDSP_U2C_KSPLIT");   // measurement aid: 0 restores whole-K launches
        ksplit_on = (e && e[0] == '0') ? 0 : 1;
    }
    const bool ksplit = ksplit_on && !bf.l[0].pre && M <= KS_MAX_ROWS;
    LnPending pending{nullptr, nullptr, nullptr, 0};
    auto residual_gemm = [&](gemm::Args g, const float* x_res, float* x_dst, const float* bias) {
        // x_dst = x_res + A B^T + bias, now or (ksplit) when the next LayerNorm reads it
        if (ksplit && gemm::dma_ok(g) && g.K % (32 * KS_SPLITS) == 0) {
            const int kc = g.K / KS_SPLITS;
            g.K = kc;
            g.sA_hi = kc;
            g.sB_hi = kc;
            gemm::EpiStore e{bf.kpart, D, nullptr, 1, M * D, 0};
            gemm::dma_go<64, 64, gemm::EpiStore, 4, 4>(st, g, KS_SPLITS, e);
            pending = LnPending{bf.kpart, bias, x_dst, 1};
            return x_res;   // the LayerNorm reads the residual from here
        }
        gemm::EpiResidual e{x_dst, x_res, D, bias};
        gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e);
        return (const float*)x_dst;
    };
    // Large batches with pre-split operands: the residual layer AND the LayerNorm that follows it in one kernel (gemm_ln.h: a
    // workgroup owns 64 rows x all 256 columns; same bits as the two launches).  DDSP_GEMM_LN=0 restores the pair.
    static int gemm_ln_on = -1;
    if (gemm_ln_on < 0) {
        const char* e = getenv("DDSP_GEMM_LN");
        gemm_ln_on = (e && e[0] == '0') ? 0 : 1;
    }
    static int64_t gemm_ln_min = -1;   // DDSP_GEMM_LN_MIN: rows from which the fused kernel runs (measurement aid)
    if (gemm_ln_min < 0) {
        const char* e = getenv("DDSP_GEMM_LN_MIN");
        // (measured: B = 8 / 1376 rows 0.476 -> 0.519 ms, B = 24 / 4128 rows 0.826 -> 0.839 ms per forward WITH the fused kernel -
        // a few workgroups each pull 640 KB of operands through one CU's load path; B = 64 / 11008 rows 1.138 -> 1.126)
        gemm_ln_min = e ? atoll(e) : 8192;
    }
    bool ln_done = false;   // the next LayerNorm's output has been written by the producer of its input
    auto ln_args = [&](const gemm::Args& g, const float* x_res, float* x_dst, const float* bias, const float* gamma,
                       const float* beta, float* y_out) {
        return gemm::LnArgs{g.A, g.B_split, g.lda, g.ldb, g.M, g.K, bias, x_res, x_dst, gamma, beta, y_out, asplit};
    };
    // (the A operand pre-split at large batches, fp32 below; small batches keep the separate LayerNorm, which sums their K-split partials)
    auto ln_fusable = [&](const gemm::Args& g, const gemm::LnArgs& a) {
        return gemm_ln_on && !ksplit && !bf.l[0].pre && g.math == DDSP_MATH_SPLIT_BF16 && g.B_split && (g.A_split || !asplit) &&
               g.N == D && M >= gemm_ln_min && gemm::res_ln_ok(a);
    };
    const int want_state = 1 | (presplit_w ? 2 : 0) | (fuse_glu ? 4 : 0);
    const bool prepared = bf.wstate && (*bf.wstate & 7) == want_state;
    const bool p3_ready = prepared && (*bf.wstate & 8);
    if (!prepared) {   // weight preparation, one launch (u2c_prepare_kernel)
        PrepArgs pa;
        pa.w = w;
        pa.w1 = bf.w1;
        pa.w2 = bf.w2;
        pa.wh = bf.wh;
        pa.wqkv = bf.wqkv;
        pa.bqkv = bf.bqkv;
        pa.wglu = bf.wglu;
        pa.bglu = bf.bglu;
        pa.wout = bf.wout;
        pa.wpw2 = bf.wpw2;
        pa.split = presplit_w ? 1 : 0;
        pa.qkv_blocks = 192;
        pa.glu_blocks = 128;
        pa.copy_blocks = 32;
        pa.end[0] = (int)grid_for((int64_t)n_w1 / 4, 256, 256);
        pa.end[1] = pa.end[0] + (int)grid_for((int64_t)n_w2 / 8, 256, 256);
        pa.end[2] = pa.end[1] + (w.n_out + 3) / 4;
        pa.end[3] = pa.end[2] + 3 * pa.qkv_blocks;
        pa.end[4] = pa.end[3] + (fuse_glu ? 3 * pa.glu_blocks : 0);
        pa.end[5] = pa.end[4] + (presplit_w ? 6 * pa.copy_blocks : 0);
        pa.end[6] = pa.end[5] + 48;
        pa.wdw = bf.wdw;
        PROF(PF_U2C_PREP, 0, 8.0 * (n_w1 + n_w2 + n_wh + 3.0 * 3 * INNER * D + (fuse_glu ? 3.0 * 2 * INNER * D : 0.0) +
                                    (presplit_w ? 6.0 * D * INNER : 0.0)),
             hipLaunchKernelGGL(u2c_prepare_kernel, dim3((unsigned)pa.end[6]), dim3(256), 0, st, pa));
        if (bf.wstate) *bf.wstate = want_state;
    }
    if (attn_bf16 && !p3_ready) {   // the three projection matrices as bf16 pieces for the split attention kernels, one launch
        PROF(PF_U2C_PREP, 0, 3.0 * (4.0 * NF * DH + PERFORMER_P3_BYTES),
             performer_p3(st, w.layer[0].proj, w.layer[1].proj, w.layer[2].proj, bf.p3));
        if (bf.wstate) *bf.wstate |= 8;
    }
    // with B_split the GEMM reads ONLY the split copy: both pointers name the same packed matrix
    auto set_b = [&](gemm::Args& g, const float* packed, int a_is_split) {
        g.math = lin_math;
        if (presplit_w) {
            g.B = packed;
            g.B_split = packed;
            g.A_split = a_is_split;
        }
    };
    // Large batches: the wave-specialised kernel (gemm_ws.h: loader / product waves, epilogue pieces riding in the next tile)
    // for the K = 256 Linear layers with wide outputs, once its 128x128 tiles make at least two rounds over the 256 CUs.  Same
    // bits as kernel_dma.  Which layers: measured INSIDE the forward (rocprofv3, B = 64: DESIGN section 9) - the head gains
    // (31.7 -> 28.4 us), QKV and pw1 + GLU do not (42.5 -> 43.9, 25.0 -> 27.9 us, although alone, with operands resident in the
    // L2, they run 44.5 -> 38.8 and 30.0 -> 25.3), so only the head uses it.  DDSP_GEMM_WS = bit mask 1 QKV | 2 GLU | 4 head.
    static int ws_mask = -1;
    if (ws_mask < 0) {
        const char* e = getenv("DDSP_GEMM_WS");
        ws_mask = e ? atoi(e) : 4;
    }
    auto use_ws = [&](const gemm::Args& g, int layer_bit) {
        return (ws_mask & layer_bit) && presplit_w && g.math == DDSP_MATH_SPLIT_BF16 && gemm::ws_ok(g) && g.N % 128 == 0 &&
               (int64_t)((g.M + 127) / 128) * (g.N / 128) >= 512;
    };
    static int dw_pair = -1;   // DDSP_DW_PAIR (measurement aid): 0 the one-channel depthwise kernel at every size, 1 / 2 the register pair kernel (runs of 16 / 14 frames), 3 (default) the LDS-tiled kernel
    if (dw_pair < 0) {
        const char* e = getenv("DDSP_DW_PAIR");
        dw_pair = e ? atoi(e) : 3;
    }
    bool conv_split = false;
    // ---- prenet: conv k3 -> GroupNorm(4) -> LeakyReLU -> conv k3 ----
    {
        gemm::Args g = gemm::make(in.units, w.n_unit, bf.w1, 3 * w.n_unit, iM, D, 3 * w.n_unit);
        set_b(g, bf.w1, 0);
        g.tap_shift = w.causal ? -1 : 0;
        g.Fr = (int)Fr;
        g.Cin = w.n_unit;
        g.zeros = zero_page;
        // (small batches: the k = 3 Cin range of the two prenet convolutions is cut over KS_SPLITS workgroups per tile as
        // well; conv1's partial products are summed by the GroupNorm statistics pass, conv2's by the first LayerNorm)
        conv_split = ksplit && gemm::dma_ok(g) && g.K % (32 * KS_SPLITS) == 0 && w.n_unit % 32 == 0 && (3 * D) % (32 * KS_SPLITS) == 0 &&
                     w.n_unit + 32 <= DDSP_ZERO_FLOATS;
        if (conv_split) {
            g.K /= KS_SPLITS;
            g.kz = g.K;
            gemm::EpiStore e{bf.kpart, D, nullptr, 1, M * D, 0};
            PROF(PF_U2C_GEMM_CONV3, 2.0 * M * D * 3 * w.n_unit, 4.0 * M * (w.n_unit + D),
                 (gemm::dma_go<64, 64, gemm::EpiStore, 4, 4, gemm::A_CONV3>(st, g, KS_SPLITS, e)));
        } else {
            gemm::EpiStore e{bf.t1, D, w.prenet_conv1_b, 1, 0, 0};
            PROF(PF_U2C_GEMM_CONV3, 2.0 * M * D * 3 * w.n_unit, 4.0 * M * (w.n_unit + D),
                 (gemm::launch<true, true, gemm::A_CONV3>(st, g, 1, e)));
        }
    }
    PROF(PF_U2C_ROWWISE, 0, 4.0 * M * D,
         hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(4, (unsigned)B), dim3(64 * GN_LANES), 0, st, bf.t1, (int)Fr, bf.gst,
                            conv_split ? bf.kpart : nullptr, w.prenet_conv1_b, M));
    PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
         hipLaunchKernelGGL(groupnorm_lrelu_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, st, bf.t1, bf.gst,
                            w.prenet_gn_w, w.prenet_gn_b, M, (int)Fr, bf.t2, asplit));
    float* x = bf.l[0].x_in;
    {
        gemm::Args g = gemm::make(bf.t2, D, bf.w2, 3 * D, iM, D, 3 * D);
        set_b(g, bf.w2, asplit);
        g.tap_shift = w.causal ? -1 : 0;
        g.Fr = (int)Fr;
        g.Cin = D;
        g.zeros = zero_page;
        // the side embeddings (f0, phase, volume, speaker) are added in this GEMM's epilogue
        if (conv_split) {
            g.K /= KS_SPLITS;
            g.kz = g.K;
            EpiEmbed e{bf.kpart, w.prenet_conv2_b, in.f0, in.phase, in.volume, w, in.spk_id, in.n_spk_id, in.mix, (int)Fr, dev_err, M * D};
            PROF(PF_U2C_GEMM_CONV3, 2.0 * M * D * 3 * D, 8.0 * M * D,
                 (gemm::dma_go<64, 64, EpiEmbed, 4, 4, gemm::A_CONV3>(st, g, KS_SPLITS, e)));
            pending = LnPending{bf.kpart, nullptr, x, 0};   // layer 0's LayerNorm sums the four partial products into x
        } else {
            EpiEmbed e{x, w.prenet_conv2_b, in.f0, in.phase, in.volume, w, in.spk_id, in.n_spk_id, in.mix, (int)Fr, dev_err, 0};
            // large batches: the convolution, the embeddings AND the first block's LayerNorm in one kernel (gemm_ln.h with the
            // 3-tap loader; same bits as the pair of launches)
            gemm::LnArgs la = ln_args(g, x, x, w.prenet_conv2_b, w.layer[0].norm_w, w.layer[0].norm_b, bf.l[0].y);
            la.Fr = (int)Fr;
            la.Cin = D;
            la.tap_shift = g.tap_shift;
            la.zeros = zero_page;
            if (conv_ln_on() && ln_fusable(g, la) && g.A_split && D + 32 <= DDSP_ZERO_FLOATS) {
                PROF(PF_U2C_GEMM_CONV3, 2.0 * M * D * 3 * D, 12.0 * M * D,
                     DDSP_HIP(ctx, (gemm::launch_res_ln_rb<2, true, gemm::A_CONV3, PreEmbed>(st, la, PreEmbed{e}))));
                ln_done = true;
            } else
                PROF(PF_U2C_GEMM_CONV3, 2.0 * M * D * 3 * D, 8.0 * M * D, (gemm::launch<true, true, gemm::A_CONV3>(st, g, 1, e)));
        }
    }
    DDSP_LAUNCH_CHECK(ctx);

    const unsigned rows_g = (unsigned)ceil_div64(M, 4), rows8_g = (unsigned)ceil_div64(M8, 4);
    const float* ln_src = nullptr;   // where the next LayerNorm finds the residual stream (null: the layer's own x_in)
    for (int l = 0; l < 3; ++l) {
        const ddsp_u2c_layer& L = w.layer[l];
        LayerBufs& b = bf.l[l];
        // -- x_mid = x_in + to_out(linear_attention(LN(x_in)))
        if (!ln_done)
            PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
                 hipLaunchKernelGGL(layernorm_kernel, dim3(rows_g), dim3(256), 0, st, ln_src ? ln_src : b.x_in, L.norm_w, L.norm_b, M, b.y, asplit, pending));
        ln_done = false;
        pending = LnPending{nullptr, nullptr, nullptr, 0};
        {
            gemm::Args g = gemm::make(b.y, D, bf.wqkv + (size_t)l * 3 * INNER * D, D, iM, 3 * INNER, D);
            set_b(g, bf.wqkv + (size_t)l * 3 * INNER * D, asplit);
            EpiSplit3 e{{b.q, b.k, b.v}, bf.bqkv + (size_t)l * 3 * INNER};
            gemm::WsSplit3 ew{{b.q, b.k, b.v}, bf.bqkv + (size_t)l * 3 * INNER};
            if (use_ws(g, 1) && ew.vec_ok()) {
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * 3 * INNER * D, 4.0 * M * (D + 3 * INNER),
                     DDSP_HIP(ctx, (gemm::ws_go<128, 128, gemm::WsSplit3, 4>(st, g, ew))));
            } else {
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * 3 * INNER * D, 4.0 * M * (D + 3 * INNER),
                     (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
            }
        }
        if (attn_bf16) {
            // inference, split-bf16 products, enough (utterance, head) pairs to fill the chip with one workgroup each:
            // the LDS-staged bf16 kernels (performer_attn_bf16.hip); the output is written as the out-projection's A operand
            void* p3 = (char*)bf.p3 + (size_t)l * PERFORMER_P3_BYTES;
            if (performer_fused_enabled()) {
                // both sides in one kernel per (utterance, head): ctx and ks stay in its LDS (round 3)
                PROF(PF_U2C_GEMM_CTX, 8.0 * M8 * NF * DH, 4.0 * M * 4 * INNER,
                     DDSP_HIP(ctx, performer_fused_bf16(st, b.q, b.k, b.v, p3, (int)B, (int)Fr, b.attn, asplit)));
            } else {
                PROF(PF_U2C_GEMM_CTX, 4.0 * M8 * NF * DH, 4.0 * M * 2 * INNER,
                     performer_kv_bf16(st, b.k, b.v, p3, (int)B, (int)Fr, b.cx, b.ks));
                PROF(PF_U2C_GEMM_ATTNOUT, 4.0 * M8 * NF * DH, 4.0 * M * 2 * INNER,
                     performer_q_bf16(st, b.q, p3, b.cx, b.ks, (int)B, (int)Fr, b.attn, 0, asplit));
            }
        } else if (w.causal && !b.pre && causal_chunked()) {
            // causal mode, inference: chunked linear attention in one kernel (performer_attn.hip); q' / k' never reach HBM
            PROF(PF_U2C_GEMM_ATTNOUT, 2.0 * M8 * (3.0 * NF * DH + 16.0 * (NF + DH)) + 4.0 * M8 * NF * DH, 4.0 * M * 4 * INNER,
                 performer_causal(st, b.q, b.k, b.v, L.proj, (int)B, (int)Fr, b.attn));
        } else if (w.causal) {
            // causal mode, training forward: feature maps through the GEMM + row kernels (q', k' stay in the arena for the
            // backward pass), then the sequential causal attention kernel
            gemm::Args g = gemm::make(b.q, DH, L.proj, DH, (int)M8, NF, DH);
            gemm::EpiStore e{b.qf, LDF, nullptr, 1, 0, 0};
            PROF(PF_U2C_GEMM_FEAT, 2.0 * M8 * NF * DH, 4.0 * M8 * (DH + NF), (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
            g.A = b.k;
            e.C = b.kf;
            PROF(PF_U2C_GEMM_FEAT, 2.0 * M8 * NF * DH, 4.0 * M8 * (DH + NF), (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
            PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * (2 * NF + DH),
                 hipLaunchKernelGGL(feature_map_kernel<true>, dim3(rows8_g), dim3(256), 0, st, b.qf, b.q, M8));
            PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * (2 * NF + DH),
                 hipLaunchKernelGGL(feature_map_kernel<false>, dim3(rows8_g), dim3(256), 0, st, b.kf, b.k, M8));
            PROF(PF_U2C_GEMM_ATTNOUT, 4.0 * M8 * NF * DH, 4.0 * M8 * (2 * NF + 2 * DH),
                 hipLaunchKernelGGL(causal_attention_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, b.qf, b.kf, b.v, (int)Fr, b.attn));
        } else if (!b.pre) {
            // inference: fused feature maps + linear attention (performer_attn.hip); q'/k' never reach HBM
            PROF(PF_U2C_GEMM_CTX, 4.0 * M8 * NF * DH, 4.0 * M * 2 * INNER,
                 performer_kv(st, b.k, b.v, L.proj, (int)B, (int)Fr, b.cx, b.ks));
            PROF(PF_U2C_GEMM_ATTNOUT, 4.0 * M8 * NF * DH, 4.0 * M * 2 * INNER,
                 performer_q(st, b.q, L.proj, b.cx, b.ks, (int)B, (int)Fr, b.attn));
        } else {
            {   // random-feature projections: (M*8, 64) x (266, 64)^T
                gemm::Args g = gemm::make(b.q, DH, L.proj, DH, (int)M8, NF, DH);
                gemm::EpiStore e{b.qf, LDF, nullptr, 1, 0, 0};
                PROF(PF_U2C_GEMM_FEAT, 2.0 * M8 * NF * DH, 4.0 * M8 * (DH + NF),
                     (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
                g.A = b.k;
                e.C = b.kf;
                PROF(PF_U2C_GEMM_FEAT, 2.0 * M8 * NF * DH, 4.0 * M8 * (DH + NF),
                     (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
            }
            PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * (2 * NF + DH),
                 hipLaunchKernelGGL(feature_map_kernel<true>, dim3(rows8_g), dim3(256), 0, st, b.qf, b.q, M8));
            PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * (2 * NF + DH),
                 hipLaunchKernelGGL(feature_map_kernel<false>, dim3(rows8_g), dim3(256), 0, st, b.kf, b.k, M8));
            PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * NF,
                 hipLaunchKernelGGL(key_sum_kernel, dim3((unsigned)(B * H)), dim3(KS_T * 8), 0, st, b.kf, (int)Fr, b.ks));
            {   // ctx[b,h] (266 x 64) = k'^T v : A stored [n][j] (K x M), B stored [n][e] (K x N)
                gemm::Args g = gemm::make(b.kf, (int64_t)H * LDF, b.v, INNER, NF, DH, (int)Fr);
                g.zdiv = H;
                g.sA_hi = (int64_t)Fr * H * LDF;
                g.sA_lo = LDF;
                g.sB_hi = (int64_t)Fr * INNER;
                g.sB_lo = DH;
                gemm::EpiStore e{b.cx, DH, nullptr, 1, (int64_t)NF * DH, 0};
                PROF(PF_U2C_GEMM_CTX, 2.0 * M8 * NF * DH, 4.0 * M8 * (NF + DH),
                     (gemm::launch_tile<64, 64, false, false, gemm::A_PLAIN>(st, g, (int)(B * H), e)));
            }
            PROF(PF_U2C_ROWWISE, 0, 4.0 * M8 * NF,
                 hipLaunchKernelGGL(attn_denominator_kernel, dim3(rows8_g), dim3(256), 0, st, b.qf, b.ks, (int)Fr, M8, b.dinv));
            {   // out[b,n,h,:] = dinv * (q'[b,n,h,:] ctx[b,h])
                gemm::Args g = gemm::make(b.qf, (int64_t)H * LDF, b.cx, DH, (int)Fr, DH, NF);
                g.zdiv = H;
                g.sA_hi = (int64_t)Fr * H * LDF;
                g.sA_lo = LDF;
                g.sB_hi = (int64_t)H * NF * DH;
                g.sB_lo = (int64_t)NF * DH;
                EpiAttnOut e{b.attn, b.dinv, (int)Fr};
                PROF(PF_U2C_GEMM_ATTNOUT, 2.0 * M8 * NF * DH, 4.0 * M8 * (NF + DH),
                     (gemm::launch_tile<64, 64, true, false, gemm::A_PLAIN>(st, g, (int)(B * H), e)));
            }
        }
        {
            gemm::Args g = gemm::make(b.attn, INNER, L.out_w, INNER, iM, D, INNER);
            set_b(g, bf.wout + (size_t)l * D * INNER, attn_bf16 ? asplit : 0);
            const gemm::LnArgs la = ln_args(g, b.x_in, b.x_mid, L.out_b, L.cm_ln_w, L.cm_ln_b, b.y2);
            if (ln_fusable(g, la)) {
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * D * INNER, 4.0 * M * (INNER + 4 * D), DDSP_HIP(ctx, gemm::launch_res_ln(st, la, g.A_split != 0)));
                ln_done = true;
                ln_src = b.x_mid;
            } else
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * D * INNER, 4.0 * M * (INNER + 2 * D), ln_src = residual_gemm(g, b.x_in, b.x_mid, L.out_b));
        }
        // -- x_out = x_mid + conv_module(x_mid)
        if (!ln_done)
            PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
                 hipLaunchKernelGGL(layernorm_kernel, dim3(rows_g), dim3(256), 0, st, ln_src, L.cm_ln_w, L.cm_ln_b, M, b.y2, asplit, pending));
        ln_done = false;
        pending = LnPending{nullptr, nullptr, nullptr, 0};
        if (fuse_glu) {
            gemm::Args g = gemm::make(b.y2, D, bf.wglu + (size_t)l * 2 * INNER * D, D, iM, 2 * INNER, D);
            set_b(g, bf.wglu + (size_t)l * 2 * INNER * D, asplit);
            EpiGlu e{b.glu, bf.bglu + (size_t)l * 2 * INNER};
            DDSP_REQUIRE(ctx, gemm::dma_ok(g) && ((uintptr_t)b.glu % 16) == 0, "unit2ctrl: fused GLU needs aligned activations");
            gemm::WsGlu ew{b.glu, INNER, bf.bglu + (size_t)l * 2 * INNER};
            if (use_ws(g, 2) && ew.vec_ok()) {
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * 2 * INNER * D, 4.0 * M * (D + INNER),
                     DDSP_HIP(ctx, (gemm::ws_go<128, 128, gemm::WsGlu, 4>(st, g, ew))));
            } else if (glu_large) {
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * 2 * INNER * D, 4.0 * M * (D + INNER),
                     (gemm::dma_go<128, 128, EpiGlu, 2>(st, g, 1, e)));
            } else {
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * 2 * INNER * D, 4.0 * M * (D + INNER),
                     (gemm::dma_go<64, 128, EpiGlu, 3, 4>(st, g, 1, e)));
            }
        } else {
            {
                gemm::Args g = gemm::make(b.y2, D, L.cm_pw1_w, D, iM, 2 * INNER, D);
                g.math = lin_math;
                gemm::EpiStore e{b.g1, 2 * INNER, L.cm_pw1_b, 1, 0, 0};
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * 2 * INNER * D, 4.0 * M * (D + 2 * INNER),
                     (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
            }
            PROF(PF_U2C_ROWWISE, 0, 12.0 * M * INNER,
                 hipLaunchKernelGGL(glu_kernel, dim3(grid_for(M * (INNER / 4))), dim3(256), 0, st, b.g1, M, b.glu));
        }
        PROF(PF_U2C_ROWWISE, 2.0 * M * INNER * DWK, 8.0 * M * INNER,
             if (!b.pre && dw_pair == 3 && B * ((Fr + 15) / 16) >= 512)
                 // inference, large batches: LDS-staged tiles of 64 frames x 64 channels
                 hipLaunchKernelGGL(dwconv_tile_kernel, dim3(INNER / DWT_C, (unsigned)(B * ((Fr + DWT_F - 1) / DWT_F))), dim3(256), 0, st,
                                    b.glu, bf.wdw + (size_t)l * DWK * INNER, L.cm_dw_b, (int)B, (int)Fr, b.dwo, w.causal ? DWK - 1 : DWK / 2, asplit);
             else if (!b.pre && dw_pair == 1 && B * ((Fr + 15) / 16) >= 512)
                 // inference, large batches: two channels per thread on packed multiply-adds, runs of 16 frames
                 hipLaunchKernelGGL((dwconv_pair_kernel<16>), dim3(1, (unsigned)(B * ((Fr + 15) / 16))), dim3(256), 0, st,
                                    b.glu, bf.wdw + (size_t)l * DWK * INNER, L.cm_dw_b, (int)B, (int)Fr, b.dwo, w.causal ? DWK - 1 : DWK / 2, asplit);
             else if (!b.pre && dw_pair == 2 && B * ((Fr + 13) / 14) >= 512)
                 hipLaunchKernelGGL((dwconv_pair_kernel<14>), dim3(1, (unsigned)(B * ((Fr + 13) / 14))), dim3(256), 0, st,
                                    b.glu, bf.wdw + (size_t)l * DWK * INNER, L.cm_dw_b, (int)B, (int)Fr, b.dwo, w.causal ? DWK - 1 : DWK / 2, asplit);
             else if (B * ((Fr + DW_RUN - 1) / DW_RUN) >= 64)
                 hipLaunchKernelGGL((dwconv_kernel<true, false>), dim3(INNER / 256, (unsigned)(B * ((Fr + DW_RUN - 1) / DW_RUN))),
                                    dim3(256), 0, st, b.glu, bf.wdw + (size_t)l * DWK * INNER, L.cm_dw_b, (int)B, (int)Fr, b.dwo, b.pre, 1, INNER, w.causal ? DWK - 1 : DWK / 2, asplit);
             else   // a few utterances (the real-time block): runs of 8 frames, four times as many workgroups
                 hipLaunchKernelGGL((dwconv_kernel<true, false, 8>), dim3(INNER / 256, (unsigned)(B * ((Fr + 7) / 8))),
                                    dim3(256), 0, st, b.glu, bf.wdw + (size_t)l * DWK * INNER, L.cm_dw_b, (int)B, (int)Fr, b.dwo, b.pre, 1, INNER, w.causal ? DWK - 1 : DWK / 2, asplit));
        {
            gemm::Args g = gemm::make(b.dwo, INNER, L.cm_pw2_w, INNER, iM, D, INNER);
            set_b(g, bf.wpw2 + (size_t)l * D * INNER, asplit);
            // (the LayerNorm behind pw2 is the next layer's, or the final one)
            const float* n_g = l + 1 < 3 ? w.layer[l + 1].norm_w : w.final_ln_w;
            const float* n_b = l + 1 < 3 ? w.layer[l + 1].norm_b : w.final_ln_b;
            float* n_y = l + 1 < 3 ? bf.l[l + 1].y : bf.y_final;
            const gemm::LnArgs la = ln_args(g, b.x_mid, b.x_out, L.cm_pw2_b, n_g, n_b, n_y);
            if (ln_fusable(g, la)) {
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * D * INNER, 4.0 * M * (INNER + 4 * D), DDSP_HIP(ctx, gemm::launch_res_ln(st, la, g.A_split != 0)));
                ln_done = true;
                ln_src = b.x_out;
            } else
                PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * D * INNER, 4.0 * M * (INNER + 2 * D), ln_src = residual_gemm(g, b.x_mid, b.x_out, L.cm_pw2_b));
        }
        DDSP_LAUNCH_CHECK(ctx);
    }
    // ---- LayerNorm -> weight-normed head ----
    if (!ln_done)
        PROF(PF_U2C_ROWWISE, 0, 8.0 * M * D,
             hipLaunchKernelGGL(layernorm_kernel, dim3(rows_g), dim3(256), 0, st, ln_src, w.final_ln_w, w.final_ln_b,
                                M, bf.y_final, asplit, pending));
    {
        gemm::Args g = gemm::make(bf.y_final, D, bf.wh, D, iM, w.n_out, D);
        set_b(g, bf.wh, asplit);
        gemm::EpiStore e{ctrl, w.n_out, w.head_b, 1, 0, 0};
        gemm::WsStore ew{ctrl, w.n_out, w.head_b};
        if (use_ws(g, 4) && ew.vec_ok()) {
            PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * w.n_out * D, 4.0 * M * (D + w.n_out),
                 DDSP_HIP(ctx, (gemm::ws_go<128, 128, gemm::WsStore, 4>(st, g, ew))));
        } else {
            PROF(PF_U2C_GEMM_LINEAR, 2.0 * M * w.n_out * D, 4.0 * M * (D + w.n_out),
                 (gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e)));
        }
    }
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

static int check_inputs(ddsp_ctx* ctx, const ddsp_u2c_weights* wp, const float* units, const float* f0_frames,
                        const float* phase_frames, const float* volume, const int64_t* spk_id, int64_t n_spk_id,
                        const int64_t* mix_ids_host, const float* mix_w_host, int n_mix, int64_t B, int64_t Fr,
                        U2CInputs& in) {
    DDSP_REQUIRE(ctx, ctx && wp && units && f0_frames && phase_frames && volume, "ddsp_unit2ctrl: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && B <= 4096 && Fr >= 1 && B * Fr < (1 << 26), "ddsp_unit2ctrl: bad shape (B <= 4096 per call)");
    DDSP_REQUIRE(ctx, n_mix >= 0 && n_mix <= 16, "ddsp_unit2ctrl: at most 16 mixed speakers");
    DDSP_REQUIRE(ctx, n_mix > 0 || (spk_id && (n_spk_id == 1 || n_spk_id == B)), "ddsp_unit2ctrl: spk_id must hold 1 or B ids");
    DDSP_REQUIRE(ctx, n_mix == 0 || (mix_ids_host && mix_w_host), "ddsp_unit2ctrl: mix arrays missing");
    DDSP_REQUIRE(ctx, wp->n_unit >= 4 && wp->n_unit % 4 == 0 && wp->n_out >= 1 && wp->n_spk >= 1, "ddsp_unit2ctrl: bad widths");
    DDSP_REQUIRE(ctx, wp->causal == 0 || wp->causal == 1, "ddsp_unit2ctrl: causal must be 0 or 1");
    for (int k = 0; k < n_mix; ++k)
        DDSP_REQUIRE(ctx, mix_ids_host[k] >= 1 && mix_ids_host[k] <= wp->n_spk, "ddsp_unit2ctrl: mixed speaker id out of range");
    in.units = units;
    in.f0 = f0_frames;
    in.phase = phase_frames;
    in.volume = volume;
    in.spk_id = spk_id;
    in.n_spk_id = n_spk_id;
    in.mix.n = n_mix;
    for (int i = 0; i < n_mix; ++i) {
        in.mix.ids[i] = mix_ids_host[i];
        in.mix.w[i] = mix_w_host[i];
    }
    in.B = B;
    in.Fr = Fr;
    return DDSP_OK;
}

// ---- helpers of the backward pass ------------------------------------------------------------------------------
// out[o*ldo + coff + c] = sum_m dY[m][o] * X[m][c]   (split-K batches on the matrix pipe + one reduction)
static int wgrad(ddsp_ctx* ctx, hipStream_t st, const float* dY, int64_t ldy, int O, const float* X, int64_t ldx, int C,
                 int64_t M, float* partial, float* out, int64_t ldo, int coff);
__global__ void __launch_bounds__(256) reduce_partials_2d_kernel(const float* __restrict__ partial, int nz, int O, int C,
                                                                 float* __restrict__ out, int64_t ldo, int coff) {
    const int64_t n = (int64_t)O * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += partial[(int64_t)z * n + i];
        out[(i / C) * ldo + coff + (i % C)] = s;
    }
}
// Deferred reductions (round 3): the 18 Linear layers of the three blocks write their split partial sums (and those of their bias
// column sums) into regions of their own and ONE launch at the end of the backward pass adds them all up - 36 launches of 5-6 us
// fewer per training step.  Same order of the additions (z ascending) as reduce_partials_2d_kernel.
constexpr int WG_SPLITS = 16;   // row splits of a weight-gradient product (buffers and the reduction's register array are sized for it)
constexpr int RED_JOBS_MAX = 48;
struct RedJobs {
    int n;
    const float* src[RED_JOBS_MAX];
    float* dst[RED_JOBS_MAX];
    int nz[RED_JOBS_MAX], O[RED_JOBS_MAX], C[RED_JOBS_MAX];
    int ldo[RED_JOBS_MAX];
    int bend[RED_JOBS_MAX];   // exclusive end of the job's range of workgroups
};
__global__ void __launch_bounds__(256) reduce_jobs_kernel(RedJobs J) {
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.bend[j]) ++j;
    const int b0 = j ? J.bend[j - 1] : 0, nb = J.bend[j] - b0;
    const float* __restrict__ partial = J.src[j];
    float* __restrict__ out = J.dst[j];
    const int nz = J.nz[j], C = J.C[j];
    const int64_t n = (int64_t)J.O[j] * C, ldo = J.ldo[j];
    if (nz <= WG_SPLITS + 1 && ((n | C | ldo) & 3) == 0 && (((uintptr_t)partial | (uintptr_t)out) & 15) == 0) {
        // four outputs per thread and every split's load in flight before the first addition (same z-ascending order of the sums)
        for (int64_t i = ((int64_t)((int)blockIdx.x - b0) * 256 + threadIdx.x) * 4; i < n; i += (int64_t)nb * 1024) {
            f32x4 v[WG_SPLITS + 1];
#pragma unroll
            for (int z = 0; z < WG_SPLITS + 1; ++z)
                if (z < nz) v[z] = *(const f32x4*)(partial + (int64_t)z * n + i);
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int z = 0; z < WG_SPLITS + 1; ++z)
                if (z < nz) s4 += v[z];
            *(f32x4*)(out + (i / C) * ldo + (i % C)) = s4;
        }
        return;
    }
    for (int64_t i = (int64_t)((int)blockIdx.x - b0) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += partial[(int64_t)z * n + i];
        out[(i / C) * ldo + (i % C)] = s;
    }
}
struct WgDefer {
    RedJobs jobs;
    float* pool;
    size_t used, cap;   // floats
    float* take(size_t n) {
        n = (n + 63) & ~(size_t)63;
        if (used + n > cap) return nullptr;
        float* p = pool + used;
        used += n;
        return p;
    }
    bool push(const float* src, float* dst, int nz, int O, int C, int64_t ldo) {
        if (jobs.n >= RED_JOBS_MAX) return false;
        const int j = jobs.n++;
        jobs.src[j] = src;
        jobs.dst[j] = dst;
        jobs.nz[j] = nz;
        jobs.O[j] = O;
        jobs.C[j] = C;
        jobs.ldo[j] = (int)ldo;
        int64_t blocks = ((int64_t)O * C + 1023) / 1024;       // four outputs per thread
        if (blocks > 256) blocks = 256;
        if (blocks < 1) blocks = 1;
        jobs.bend[j] = (j ? jobs.bend[j - 1] : 0) + (int)blocks;
        return true;
    }
};
static void flush_deferred(hipStream_t st, WgDefer& df) {
    if (df.jobs.n == 0) return;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)df.jobs.bend[df.jobs.n - 1]), dim3(256), 0, st, df.jobs);
    df.jobs.n = 0;
    df.used = 0;
}

static int wgrad(ddsp_ctx* ctx, hipStream_t st, const float* dY, int64_t ldy, int O, const float* X, int64_t ldx, int C,
                 int64_t M, float* partial, float* out, int64_t ldo, int coff) {
    int64_t chunk = (M + WG_SPLITS - 1) / WG_SPLITS;
    chunk = (chunk + 31) & ~(int64_t)31;
    const int nfull = (int)(M / chunk);
    const int64_t tail = M - (int64_t)nfull * chunk;
    ddsp_prof_begin(ctx, st, PF_U2C_BWD);
    if (nfull > 0) {
        gemm::Args g = gemm::make(dY, ldy, X, ldx, O, C, (int)chunk);
        g.sA_hi = chunk * ldy;
        g.sB_hi = chunk * ldx;
        gemm::EpiStore e{partial, C, nullptr, 1, (int64_t)O * C, 0};
        gemm::launch_tile<64, 64, false, false, gemm::A_PLAIN>(st, g, nfull, e);
    }
    if (tail > 0) {
        gemm::Args g = gemm::make(dY + (int64_t)nfull * chunk * ldy, ldy, X + (int64_t)nfull * chunk * ldx, ldx, O, C, (int)tail);
        gemm::EpiStore e{partial + (int64_t)nfull * O * C, C, nullptr, 1, 0, 0};
        gemm::launch_tile<64, 64, false, false, gemm::A_PLAIN>(st, g, 1, e);
    }
    hipLaunchKernelGGL(reduce_partials_2d_kernel, dim3(grid_for((int64_t)O * C)), dim3(256), 0, st, partial,
                       nfull + (tail > 0 ? 1 : 0), O, C, out, ldo, coff);
    ddsp_prof_end(ctx, st, 2.0 * M * O * (double)C, 4.0 * M * (O + C));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

// out[c] = sum_r X[r][c] * w(r)
static int colsum(ddsp_ctx* ctx, hipStream_t st, const float* X, int64_t ld, int64_t rows, int cols, const float* wsrc,
                  int wmode, float* partial, float* out) {
    if (cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)partial % 16) == 0)
        hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3((cols + 255) / 256, CS_CHUNKS), dim3(256), 0, st, X, ld, rows, cols,
                           wsrc, wmode, partial);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 63) / 64, CS_CHUNKS), dim3(256), 0, st, X, ld, rows, cols, wsrc,
                           wmode, partial);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, partial, CS_CHUNKS, (int64_t)cols, out);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

// several column sums of one shape in one pass (cols % 4 == 0, 16-byte aligned rows; cpart holds 4 x CS_CHUNKS x cols)
static int colsum_multi(ddsp_ctx* ctx, hipStream_t st, const ColsumJobs& jb, int64_t ld, int64_t rows, int cols, float* partial) {
    bool vec = cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)partial % 16) == 0 && (size_t)jb.n * cols <= 2048;
    for (int i = 0; i < jb.n; ++i) vec = vec && ((uintptr_t)jb.X[i] % 16) == 0;
    if (!vec) {
        for (int i = 0; i < jb.n; ++i)
            if (int rc = colsum(ctx, st, jb.X[i], ld, rows, cols, jb.wsrc[i], jb.wmode[i], partial, jb.out[i])) return rc;
        return DDSP_OK;
    }
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((cols + 255) / 256, CS_CHUNKS, jb.n), dim3(256), 0, st, jb, ld, rows, cols, partial);
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((cols + 63) / 64, jb.n), dim3(256), 0, st, partial, jb, cols);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
static int colsum_pair(ddsp_ctx* ctx, hipStream_t st, const float* X0, const float* X1, int64_t ld, int64_t rows, int cols,
                       float* partial, float* out0, float* out1) {
    ColsumJobs jb{};
    jb.n = 2;
    jb.X[0] = X0;
    jb.X[1] = X1;
    jb.out[0] = out0;
    jb.out[1] = out1;
    return colsum_multi(ctx, st, jb, ld, rows, cols, partial);
}

// Weight and bias gradient of one Linear (taps = 1) or Conv1d k=3 (taps = 3, X = the layer's input, out = the packed
// (O, 3*C) matrix) from the same dY.  With split-bf16 products (the context's default arithmetic) one launch of the
// transposing bf16 kernel (wgrad_bf16.h) produces the split partials of both and two small kernels add them; with
// ddsp_ctx_set_math(FP32) the round-1 path runs: fp32-MFMA split-K batches, shifted copies of X per tap, a separate column sum.
static int wgrad_tile_choice() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_WGRAD_TILE");   // measurement aid: 11, 21, 12, 22 = (TM, TN)
        v = e ? atoi(e) : 0;
    }
    return v;
}
static bool attn_feat_fused_on() {   // DDSP_ATTN_FEAT_FUSED=0: the feature-map adjoints back on the five-launch chain through HBM
    static const bool v = [] {
        const char* e = getenv("DDSP_ATTN_FEAT_FUSED");
        return !(e && e[0] == '0');
    }();
    return v;
}
static bool attn_wgrad_on() {   // DDSP_ATTN_WGRAD=0: the K = frames product of the attention adjoint back on the register-staged fp32 kernel
    static const bool v = [] {
        const char* e = getenv("DDSP_ATTN_WGRAD");
        return !(e && e[0] == '0');
    }();
    return v;
}
static int layer_grads(ddsp_ctx* ctx, hipStream_t st, const float* dY, int64_t ldy, int O, const float* X, int64_t ldx, int C,
                       int taps, int Fr, int64_t M, float* wpart, float* cpart, float* xs, float* w_out, int64_t ldo,
                       float* b_out, int tap_shift = 0, WgDefer* defer = nullptr) {   // tap_shift: 0 centred taps, -1 causal taps (taps == 3 only)
    int rc;
    const bool split = ctx->math == DDSP_MATH_SPLIT_BF16 && (taps == 1 || C % 128 == 0);
    if (!split) {
        if (taps == 1) {
            if ((rc = wgrad(ctx, st, dY, ldy, O, X, ldx, C, M, wpart, w_out, ldo, 0))) return rc;
        } else {
            for (int tap = 0; tap < taps; ++tap) {
                hipLaunchKernelGGL(shift_rows_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, st, X, M, Fr, C, tap - 1 + tap_shift, xs);
                if ((rc = wgrad(ctx, st, dY, ldy, O, xs, C, C, M, wpart, w_out, ldo, tap * C))) return rc;
            }
        }
        if (b_out) return colsum(ctx, st, dY, ldy, M, O, nullptr, 0, cpart, b_out);
        return DDSP_OK;
    }
    wgrad::Args g;
    g.dY = dY;
    g.ldy = ldy;
    g.O = O;
    g.X = X;
    g.ldx = ldx;
    g.C = C;
    g.taps = taps;
    g.tap_shift = taps == 3 ? tap_shift : 0;
    g.Fr = Fr;
    g.M = M;
    static const int want_splits = [] {   // DDSP_WGRAD_SPLITS (1..16): row splits of a weight-gradient product (buffers are sized for 16)
        const char* e = getenv("DDSP_WGRAD_SPLITS");
        const int v = e ? atoi(e) : WG_SPLITS;
        return v < 1 ? 1 : (v > WG_SPLITS ? WG_SPLITS : v);
    }();
    g.chunk = wgrad::chunk_for(M, want_splits);
    const int nz = wgrad::splits_for(M, g.chunk), N = taps * C;
    // deferred: partial sums into regions of the caller's pool, added up by ONE launch at the end of the backward pass
    float* dpart = nullptr;
    float* dbias = nullptr;
    if (defer && defer->jobs.n + 2 <= RED_JOBS_MAX) {
        const size_t u0 = defer->used;
        dpart = defer->take((size_t)nz * O * N);
        dbias = b_out ? defer->take((size_t)nz * O) : nullptr;
        if (!dpart || (b_out && !dbias)) {
            defer->used = u0;
            dpart = dbias = nullptr;
        }
    }
    g.partial = dpart ? dpart : wpart;
    g.bias_partial = b_out ? (dpart ? dbias : cpart) : nullptr;
    ddsp_prof_begin(ctx, st, PF_U2C_BWD);
    int tile = wgrad_tile_choice();
    // 64x64 tiles: with 16 splits every layer of the network gives 512-1280 workgroups; the larger tiles stage less per
    // product but leave CUs idle at these sizes (r02, training step B=32: 8.30 ms against 8.45 / 8.45 / 8.59 with 128x64 /
    // 64x128 / 128x128)
    if (tile == 0) tile = 11;
    if (tile == 22) wgrad::launch<2, 2>(st, g);
    else if (tile == 21) wgrad::launch<2, 1>(st, g);
    else if (tile == 12) wgrad::launch<1, 2>(st, g);
    else wgrad::launch<1, 1>(st, g);
    if (dpart) {
        defer->push(dpart, w_out, nz, O, N, ldo);
        if (b_out) defer->push(dbias, b_out, nz, 1, O, O);
    } else {
        hipLaunchKernelGGL(reduce_partials_2d_kernel, dim3(grid_for((int64_t)O * N)), dim3(256), 0, st, wpart, nz, O, N, w_out, ldo, 0);
        if (b_out) hipLaunchKernelGGL(reduce_partials_kernel, dim3((O + 63) / 64), dim3(256), 0, st, cpart, nz, (int64_t)O, b_out);
    }
    ddsp_prof_end(ctx, st, 2.0 * M * O * (double)N, 4.0 * M * (O + C));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

// dX[m][c] = sum_o dY[m][o] * W[o][c]   (W stored (O, C) like nn.Linear.weight), optionally accumulated into dX
// WT: the weight transposed to (C, O) in the pre-split operand layout (transpose_split_kernel), or null.  With it the
// product runs on the LDS-DMA kernel in split-bf16 arithmetic (dY split in the loop, W read split); without it on the
// register-staged fp32 kernel, which can read W as stored.
static void dgrad(hipStream_t st, const float* dY, int64_t ldy, const float* W, int O, int C, int64_t M, float* dX,
                  bool accumulate, const float* WT = nullptr) {
    if (WT) {
        gemm::Args g = gemm::make(dY, ldy, WT, O, (int)M, C, O);
        g.math = DDSP_MATH_SPLIT_BF16;
        g.B_split = WT;
        if (accumulate) {
            EpiAccumulate e{dX, C};
            gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e);
        } else {
            gemm::EpiStore e{dX, C, nullptr, 1, 0, 0};
            gemm::launch<true, true, gemm::A_PLAIN>(st, g, 1, e);
        }
        return;
    }
    gemm::Args g = gemm::make(dY, ldy, W, C, (int)M, C, O);
    if (accumulate) {
        EpiAccumulate e{dX, C};
        gemm::launch<true, false, gemm::A_PLAIN>(st, g, 1, e);
    } else {
        gemm::EpiStore e{dX, C, nullptr, 1, 0, 0};
        gemm::launch<true, false, gemm::A_PLAIN>(st, g, 1, e);
    }
}

// Transposed, pre-split copies of the weights the input gradients multiply by: dst[c][o] (pitch O) in the (8 hi | 8 lo)
// group layout of gemm::Args::B_split.  One launch for all matrices of the network (blockIdx.y = matrix); a thread owns
// one (column c, group of 8 rows o) record, like the staging of wgrad_bf16.h.
constexpr int TS_MAX = 20;
struct TsArgs {
    const float* src[TS_MAX];
    float* dst[TS_MAX];
    int O[TS_MAX], C[TS_MAX];
};
__global__ void __launch_bounds__(256) transpose_split_kernel(TsArgs t) {
    const int mi = blockIdx.y;
    const float* __restrict__ src = t.src[mi];
    float* __restrict__ dst = t.dst[mi];
    const int O = t.O[mi], C = t.C[mi];
    const int tiles_c = C / 64, tiles = tiles_c * (O / 32);
    const int c_in = threadIdx.x & 63, og = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int c = (tile % tiles_c) * 64 + c_in, o = (tile / tiles_c) * 32 + og * 8;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = src[(int64_t)(o + j) * C + c];
        ddsp_u32x4 hi, lo;
        ddsp_split8(x, hi, lo);
        ddsp_u32x4* p = reinterpret_cast<ddsp_u32x4*>(dst + (int64_t)c * O + o);
        p[0] = hi;
        p[1] = lo;
    }
}

}  // namespace

extern "C" int ddsp_unit2ctrl_fwd(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* wp, const float* units,
                                  const float* f0_frames, const float* phase_frames, const float* volume,
                                  const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                                  const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, float* ctrl) {
    U2CInputs in;
    int rc = check_inputs(ctx, wp, units, f0_frames, phase_frames, volume, spk_id, n_spk_id, mix_ids_host, mix_w_host,
                          n_mix, B, Fr, in);
    if (rc) return rc;
    DDSP_REQUIRE(ctx, ctrl, "ddsp_unit2ctrl_fwd: null ctrl");
    if ((rc = ddsp_take_dev_error(ctx))) return rc;
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const ddsp_u2c_weights w = *wp;
    U2CBufs bf;
    // prepared-weight slot of the context (ddsp_u2c_weights::version != 0; never while the stream is being captured: a graph
    // replay must prepare the weights of ITS time)
    bool cached = false;
    if (w.version != 0) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) cs = hipStreamCaptureStatusActive;
        if (cs == hipStreamCaptureStatusNone) {
            uint64_t key = 1469598103934665603ull;   // FNV-1a over the struct without its counter: pointers and sizes
            const unsigned char* bytes = (const unsigned char*)&w;
            for (size_t i = 0; i < offsetof(ddsp_u2c_weights, version); ++i) key = (key ^ bytes[i]) * 1099511628211ull;
            if (key == 0) key = 1;
            Arena wdry{ctx, true, 0, 0};
            plan_weights(wdry, bf, w);
            if (ctx->wcache_bytes < wdry.total) {
                // (the stream may still read the old buffer: wait for it before it goes)
                DDSP_HIP(ctx, hipStreamSynchronize(st));
                if (ctx->wcache) (void)hipFree(ctx->wcache);
                ctx->wcache = nullptr;
                ctx->wcache_bytes = 0;
                ctx->wcache_key = 0;
                DDSP_HIP(ctx, hipMalloc((void**)&ctx->wcache, wdry.total));
                ctx->wcache_bytes = wdry.total;
            }
            if (ctx->wcache_key != key || ctx->wcache_version != w.version) {
                ctx->wcache_key = key;
                ctx->wcache_version = w.version;
                ctx->wcache_flags = 0;
            }
            Arena wa{ctx, false, 0, 0};
            wa.ext = ctx->wcache;
            wa.ext_cap = ctx->wcache_bytes;
            plan_weights(wa, bf, w);
            if (wa.rc) return wa.rc;
            bf.wstate = &ctx->wcache_flags;
            cached = true;
        }
    }
    // (the arena is sized for a call WITHOUT the slot too: a capture that follows warm-up calls must not have to grow it)
    Arena dry{ctx, true, 0, 0};
    {
        U2CBufs scratch_plan;
        plan_forward(dry, scratch_plan, w, B, Fr, false, false);
    }
    rc = ddsp_scratch_reserve_bytes(ctx, dry.total + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    Arena a{ctx, false, 0, 0};
    plan_forward(a, bf, w, B, Fr, false, cached);
    if (a.rc) return a.rc;
    return u2c_forward(ctx, st, w, in, bf, ctrl);
}

// the backward pass; `keep` = the activation region a ddsp_unit2ctrl_fwd_keep call filled (then nothing is recomputed), or
// null: the forward is re-run here with its activations in the scratch arena
static int u2c_backward(ddsp_ctx* ctx, hipStream_t st, const ddsp_u2c_weights& w, const ddsp_u2c_weights& gr, const U2CInputs& in,
                        int64_t B, int64_t Fr, const float* d_ctrl, float* ctrl_out, void* keep, size_t keep_bytes) {
    int rc;
#define G(p) const_cast<float*>(gr.p)
    const int64_t M = B * Fr, M8 = M * H;
    const int NO = w.n_out;

    // ---- arena: kept forward activations + backward temporaries ----
    U2CBufs bf;
    float *ctrl = nullptr, *dX, *dA, *dB512, *dC512, *dV512, *dG1, *dQF, *dKF, *dcx, *dks, *dD, *coefq, *coefk, *gx, *wpart, *cpart,
        *dWh, *pk, *xs, *dwpart, *gbst, *w2t, *wts, *wpool, *ptp, *dcxt;
    // partial sums of the 18 Linear layers of the blocks, reduced by one launch at the end (WgDefer): 17 splits at most
    const size_t wpool_floats = (size_t)(WG_SPLITS + 1) * 3 * ((size_t)2 * D * INNER + (size_t)2 * INNER * D + (size_t)3 * INNER * D + 2 * D + 2 * INNER + 3 * INNER + 1024);
    auto plan_bwd = [&](Arena& a) {
        if (!keep) {
            plan_forward(a, bf, w, B, Fr, true);
            ctrl = a.get((size_t)M * NO);
        }
        dX = a.get((size_t)M * D);          // gradient on the residual stream
        dA = a.get((size_t)M * D);          // second stream-sized temporary
        dB512 = a.get((size_t)M * INNER);   // d_dwo / d_pre / d_attn->d_num / d_q
        dC512 = a.get((size_t)M * INNER);   // d_glu / d_k
        dV512 = a.get((size_t)M * INNER);   // d_v
        dG1 = a.get((size_t)M * 2 * INNER);
        dQF = a.get((size_t)M8 * LDF);
        dKF = a.get((size_t)M8 * LDF);
        dcx = a.get((size_t)B * H * NF * DH);
        dks = a.get((size_t)B * H * LDF);
        dD = a.get((size_t)M8);
        coefq = a.get((size_t)M8);
        coefk = a.get((size_t)M8);
        gx = a.get((size_t)M * D);
        const size_t omax = (size_t)(NO > 2 * INNER ? NO : 2 * INNER);
        wpart = a.get((size_t)(WG_SPLITS + 1) * omax * (size_t)(w.n_unit > INNER ? w.n_unit : INNER));
        cpart = a.get((size_t)CS_CHUNKS * (omax > 2048 ? omax : 2048));
        dWh = a.get((size_t)NO * D);
        pk = a.get((size_t)D * 3 * (w.n_unit > D ? w.n_unit : D));
        xs = a.get((size_t)M * (w.n_unit > D ? w.n_unit : D));
        dwpart = a.get((size_t)B * ((Fr + DW_RUN - 1) / DW_RUN) * INNER * DWK);
        gbst = a.get((size_t)B * 4 * 2);
        w2t = a.get((size_t)D * 3 * D);
        wts = a.get((size_t)NO * D + 3 * ((size_t)2 * D * INNER + (size_t)2 * INNER * D + (size_t)3 * INNER * D));
        wpool = a.get(wpool_floats);
        ptp = a.get((size_t)3 * FB_PT_VEC * 4);   // prepared projections of attn_feat_bwd_kernel
        dcxt = a.get((size_t)B * H * FB_PT_VEC * 4);   // d_ctx^T of every (utterance, head) in the same layout
    };
    if (keep) {
        Arena k{ctx, false, 0, 0, (char*)keep, keep_bytes};
        plan_forward(k, bf, w, B, Fr, true);
        DDSP_REQUIRE(ctx, !k.rc, "ddsp_unit2ctrl_bwd_kept: the activation region is smaller than ddsp_unit2ctrl_keep_bytes says");
    }
    Arena dry{ctx, true, 0, 0};
    plan_bwd(dry);
    rc = ddsp_scratch_reserve_bytes(ctx, dry.total + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    Arena a{ctx, false, 0, 0};
    plan_bwd(a);
    if (a.rc) return a.rc;

    if (!keep) {
        rc = u2c_forward(ctx, st, w, in, bf, ctrl);
        if (rc) return rc;
        if (ctrl_out) DDSP_HIP(ctx, hipMemcpyAsync(ctrl_out, ctrl, (size_t)M * NO * sizeof(float), hipMemcpyDeviceToDevice, st));
    }

    const unsigned rows_g = (unsigned)ceil_div64(M, 4), rows8_g = (unsigned)ceil_div64(M8, 4);
    // ---- transposed split copies of the weights for the input-gradient products (split-bf16 arithmetic only) ----
    const float *wt_head = nullptr, *wt_pw2[3] = {}, *wt_pw1[3] = {}, *wt_out[3] = {}, *wt_qkv[3][3] = {};
    if (ctx->math == DDSP_MATH_SPLIT_BF16) {
        TsArgs ts;
        int n = 0;
        float* next = wts;
        auto add = [&](const float* src, int O, int C) -> const float* {
            if (O % 32 != 0 || C % 64 != 0 || O < 256) return nullptr;   // (gemm::launch sends K >= 256 to the DMA kernel at every M)
            ts.src[n] = src;
            ts.dst[n] = next;
            ts.O[n] = O;
            ts.C[n] = C;
            ++n;
            next += (size_t)O * C;
            return next - (size_t)O * C;
        };
        wt_head = add(bf.wh, NO, D);
        for (int l = 0; l < 3; ++l) {
            wt_pw2[l] = add(w.layer[l].cm_pw2_w, D, INNER);
            wt_pw1[l] = add(w.layer[l].cm_pw1_w, 2 * INNER, D);
            wt_out[l] = add(w.layer[l].out_w, D, INNER);
            wt_qkv[l][0] = add(w.layer[l].q_w, INNER, D);
            wt_qkv[l][1] = add(w.layer[l].k_w, INNER, D);
            wt_qkv[l][2] = add(w.layer[l].v_w, INNER, D);
        }
        static_assert(TS_MAX >= 19, "table too small");
        hipLaunchKernelGGL(transpose_split_kernel, dim3(64, n), dim3(256), 0, st, ts);
    }
    const bool feat_fused = ctx->math != DDSP_MATH_FP32 && !w.causal && attn_feat_fused_on();
    if (feat_fused)
        for (int l = 0; l < 3; ++l)
            hipLaunchKernelGGL(feat_proj_prep_kernel, dim3((FB_KS * 4 * 64 + 255) / 256), dim3(256), 0, st, w.layer[l].proj,
                               reinterpret_cast<ddsp_u32x4*>(ptp) + (size_t)l * FB_PT_VEC);
    WgDefer df{};
    df.pool = wpool;
    df.cap = wpool_floats;
    static int defer_on = -1;   // DDSP_WGRAD_DEFER=0: every layer's partial sums reduced right behind its product (measurement aid)
    if (defer_on < 0) {
        const char* e = getenv("DDSP_WGRAD_DEFER");
        defer_on = (e && e[0] == '0') ? 0 : 1;
    }
    WgDefer* const dfp = defer_on ? &df : nullptr;
    // ---- head: ctrl = LN(x) W^T + b, W = g v/|v| ----
    if ((rc = layer_grads(ctx, st, d_ctrl, NO, NO, bf.y_final, D, D, 1, (int)Fr, M, wpart, cpart, xs, dWh, D, G(head_b)))) return rc;
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((NO + 3) / 4), dim3(256), 0, st, w.head_g, w.head_v, dWh, NO, D,
                       G(head_g), G(head_v));
    dgrad(st, d_ctrl, NO, bf.wh, NO, D, M, dA, false, wt_head);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(rows_g), dim3(256), 0, st, bf.l[2].x_out, w.final_ln_w, dA, nullptr, M, dX, gx);
    if ((rc = colsum_pair(ctx, st, gx, dA, D, M, D, cpart, G(final_ln_w), G(final_ln_b)))) return rc;

    for (int l = 2; l >= 0; --l) {
        const ddsp_u2c_layer& L = w.layer[l];
        const ddsp_u2c_layer& GL = gr.layer[l];
        LayerBufs& b = bf.l[l];
#define GLP(p) const_cast<float*>(GL.p)
        // ===== conv module: x_out = x_mid + pw2(silu(dw(glu(pw1(LN(x_mid)))))) =====
        if ((rc = layer_grads(ctx, st, dX, D, D, b.dwo, INNER, INNER, 1, (int)Fr, M, wpart, cpart, xs, GLP(cm_pw2_w), INNER,
                              GLP(cm_pw2_b), 0, dfp))) return rc;
        dgrad(st, dX, D, L.cm_pw2_w, D, INNER, M, dB512, false, wt_pw2[l]);                                  // d_dwo
        hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(M * INNER)), dim3(256), 0, st, b.pre, dB512, M * INNER, dB512);  // d_pre
        {
            const int runs = (int)((Fr + DW_RUN - 1) / DW_RUN);
            hipLaunchKernelGGL(dwconv_wgrad_run_kernel, dim3(INNER / 256, (unsigned)(B * runs)), dim3(256), 0, st, dB512, b.glu, (int)B,
                               (int)Fr, dwpart, w.causal ? DWK - 1 : DWK / 2);
            hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(INNER / 64, DWK), dim3(256), 0, st, dwpart, (int)B * runs, GLP(cm_dw_w));
        }
        if ((rc = colsum(ctx, st, dB512, INNER, M, INNER, nullptr, 0, cpart, GLP(cm_dw_b)))) return rc;
        hipLaunchKernelGGL((dwconv_kernel<false, true>), dim3(INNER / 256, (unsigned)(B * ((Fr + DW_RUN - 1) / DW_RUN))),
                           dim3(256), 0, st, dB512, L.cm_dw_w, nullptr, (int)B, (int)Fr, dC512, nullptr, DWK, 1, w.causal ? 0 : DWK / 2, 0);   // d_glu (adjoint taps: left' = DWK - 1 - left)
        hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(M * INNER)), dim3(256), 0, st, b.g1, dC512, M, dG1);
        if ((rc = layer_grads(ctx, st, dG1, 2 * INNER, 2 * INNER, b.y2, D, D, 1, (int)Fr, M, wpart, cpart, xs, GLP(cm_pw1_w), D,
                              GLP(cm_pw1_b), 0, dfp))) return rc;
        dgrad(st, dG1, 2 * INNER, L.cm_pw1_w, 2 * INNER, D, M, dA, false, wt_pw1[l]);                         // d_y2
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(rows_g), dim3(256), 0, st, b.x_mid, L.cm_ln_w, dA, dX, M, dX, gx);
        if ((rc = colsum_pair(ctx, st, gx, dA, D, M, D, cpart, GLP(cm_ln_w), GLP(cm_ln_b)))) return rc;
        // dX now holds d x_mid

        // ===== attention: x_mid = x_in + to_out(attn) =====
        if ((rc = layer_grads(ctx, st, dX, D, D, b.attn, INNER, INNER, 1, (int)Fr, M, wpart, cpart, xs, GLP(out_w), INNER,
                              GLP(out_b), 0, dfp))) return rc;
        dgrad(st, dX, D, L.out_w, D, INNER, M, dB512, false, wt_out[l]);                                       // d_attn
        // the two K = 64 adjoint products of the attention (both operands K-contiguous, one problem per (utterance, head)): on the
        // LDS-DMA kernel in the context's arithmetic (round 3; the register-staged fp32 kernel took 51 us each at B = 32)
        auto attn_k64 = [&](hipStream_t s_, gemm::Args g, int batch, const EpiRowOuter& e) {
            if (ctx->math != DDSP_MATH_FP32 && gemm::dma_ok(g)) {
                g.math = DDSP_MATH_SPLIT_BF16;
                gemm::dma_go<64, 64, EpiRowOuter, 3, 4>(s_, g, batch, e);
            } else
                gemm::launch_tile<64, 64, true, true, gemm::A_PLAIN>(s_, g, batch, e);
        };
        if (w.causal) {
            // causal attention: three sequential scans per (utterance, head) (dD / coefq hold 1/den and d den between them)
            hipLaunchKernelGGL(causal_attn_bwd_q_kernel, dim3((unsigned)(B * H)), dim3(320), 0, st, b.qf, b.kf, b.v, dB512, b.attn,
                               (int)Fr, dQF, dD, coefq);
            hipLaunchKernelGGL(causal_attn_bwd_k_kernel, dim3((unsigned)(B * H)), dim3(320), 0, st, b.qf, b.v, dB512, dD, coefq,
                               (int)Fr, dKF);
            hipLaunchKernelGGL(causal_attn_bwd_v_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, b.qf, b.kf, dB512, dD, (int)Fr,
                               dV512);
        } else {
            bool dv_done = false;   // d_v folded into the key side of attn_feat_bwd_kernel
            hipLaunchKernelGGL(attn_out_bwd_kernel, dim3(rows8_g), dim3(256), 0, st, dB512, b.attn, b.dinv, M8, dD);  // d_num, d_D
            if (!feat_fused) {   // d_q' = d_num ctx^T + d_D ks^T
                gemm::Args g = gemm::make(dB512, INNER, b.cx, DH, (int)Fr, NF, DH);
                g.zdiv = H;
                g.sA_hi = (int64_t)Fr * INNER;
                g.sA_lo = DH;
                g.sB_hi = (int64_t)H * NF * DH;
                g.sB_lo = (int64_t)NF * DH;
                EpiRowOuter e{dQF, dD, b.ks, (int)Fr};
                attn_k64(st, g, (int)(B * H), e);
            }
            if (ctx->math != DDSP_MATH_FP32 && attn_wgrad_on()) {   // d_ctx = q'^T d_num: the frames are the slow axis of both operands,
                wgrad::Args g;                                        // the weight-gradient kernel with one problem per (utterance, head)
                g.dY = b.qf;
                g.ldy = (int64_t)H * LDF;
                g.O = NF;
                g.X = dB512;
                g.ldx = INNER;
                g.C = DH;
                g.taps = 1;
                g.tap_shift = 0;
                g.Fr = (int)Fr;
                g.M = Fr;
                g.chunk = 32;
                g.partial = dcx;
                g.bias_partial = dks;          // d_ks = q'^T d_D rides in the staging threads (was weighted_key_sum_kernel)
                g.ldb = LDF;
                g.bias_w = dD;
                g.ldw = H;
                g.sW_hi = (int64_t)Fr * H;
                g.sW_lo = 1;
                g.zdiv = H;
                g.sY_hi = (int64_t)Fr * H * LDF;
                g.sY_lo = LDF;
                g.sX_hi = (int64_t)Fr * INNER;
                g.sX_lo = DH;
                wgrad::launch<1, 1>(st, g, (int)(B * H));   // (128-row tiles: no difference, 5.72 / 5.75 ms over two runs)
            } else {   // d_ctx = q'^T d_num
                gemm::Args g = gemm::make(b.qf, (int64_t)H * LDF, dB512, INNER, NF, DH, (int)Fr);
                g.zdiv = H;
                g.sA_hi = (int64_t)Fr * H * LDF;
                g.sA_lo = LDF;
                g.sB_hi = (int64_t)Fr * INNER;
                g.sB_lo = DH;
                gemm::EpiStore e{dcx, DH, nullptr, 1, (int64_t)NF * DH, 0};
                gemm::launch_tile<64, 64, false, false, gemm::A_PLAIN>(st, g, (int)(B * H), e);
            }
            if (!(ctx->math != DDSP_MATH_FP32 && attn_wgrad_on()))
                hipLaunchKernelGGL(weighted_key_sum_kernel, dim3((unsigned)(B * H)), dim3(KS_T * 8), 0, st, b.qf, dD, (int)Fr, dks);
            if (feat_fused) {   // d_q (in place of d_num, which the d_ctx launch above has consumed) and d_k: attn_feat_bwd_kernel
                const dim3 fgrid((unsigned)(B * H), (unsigned)((Fr + 16 * FB_WAVES - 1) / (16 * FB_WAVES)));
                DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)attn_feat_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS_BYTES));
                                     DDSP_HIP(ctx, hipFuncSetAttribute((const void*)attn_feat_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS_BYTES)));
                const ddsp_u32x4* pt = reinterpret_cast<const ddsp_u32x4*>(ptp) + (size_t)l * FB_PT_VEC;
                static const int ablate = [] { const char* e = getenv("DDSP_FEAT_ABLATE"); return e ? atoi(e) : 0; }();
                static const bool fold_v = [] { const char* e = getenv("DDSP_ATTN_DV_FOLD"); return !(e && e[0] == '0'); }();
                if (fold_v)
                    hipLaunchKernelGGL(feat_proj_prep_kernel, dim3((FB_KS * 4 * 64 + 255) / 256, (unsigned)(B * H)), dim3(256), 0, st, dcx,
                                       reinterpret_cast<ddsp_u32x4*>(dcxt));
                dv_done = fold_v;
                FeatBwdArgs fq{dB512, b.cx, dD, b.ks, b.qf, pt, L.proj, b.q, dB512, (int)Fr, nullptr, nullptr, ablate};
                hipLaunchKernelGGL(attn_feat_bwd_kernel<true>, fgrid, dim3(64 * FB_WAVES), FB_LDS_BYTES, st, fq);
                FeatBwdArgs fk{b.v, dcx, nullptr, dks, b.kf, pt, nullptr, b.k, dC512, (int)Fr,
                               fold_v ? reinterpret_cast<const ddsp_u32x4*>(dcxt) : nullptr, dV512, ablate};
                hipLaunchKernelGGL(attn_feat_bwd_kernel<false>, fgrid, dim3(64 * FB_WAVES), FB_LDS_BYTES, st, fk);
            } else {   // d_k' = v d_ctx^T + d_ks^T
                gemm::Args g = gemm::make(b.v, INNER, dcx, DH, (int)Fr, NF, DH);
                g.zdiv = H;
                g.sA_hi = (int64_t)Fr * INNER;
                g.sA_lo = DH;
                g.sB_hi = (int64_t)H * NF * DH;
                g.sB_lo = (int64_t)NF * DH;
                EpiRowOuter e{dKF, nullptr, dks, (int)Fr};
                attn_k64(st, g, (int)(B * H), e);
            }
            if (!dv_done) {   // d_v = k' d_ctx
                gemm::Args g = gemm::make(b.kf, (int64_t)H * LDF, dcx, DH, (int)Fr, DH, NF);
                g.zdiv = H;
                g.sA_hi = (int64_t)Fr * H * LDF;
                g.sA_lo = LDF;
                g.sB_hi = (int64_t)H * NF * DH;
                g.sB_lo = (int64_t)NF * DH;
                EpiAttnOut e{dV512, nullptr, (int)Fr};
                gemm::launch_tile<64, 64, true, false, gemm::A_PLAIN>(st, g, (int)(B * H), e);
            }
        }
        if (!(feat_fused && !w.causal)) {
        hipLaunchKernelGGL(feature_map_bwd_kernel<true>, dim3(rows8_g), dim3(256), 0, st, b.qf, dQF, M8, coefq);
        hipLaunchKernelGGL(feature_map_bwd_kernel<false>, dim3(rows8_g), dim3(256), 0, st, b.kf, dKF, M8, coefk);
        }
        if (!(feat_fused && !w.causal)) {   // d_q = d_raw_q P + coef_q q   (rows = (frame, head), 64 columns == the (M, 512) layout of q)
            gemm::Args g = gemm::make(dQF, LDF, L.proj, DH, (int)M8, DH, NF);
            EpiAxpyRow e{dB512, coefq, b.q, DH};
            gemm::launch<true, false, gemm::A_PLAIN>(st, g, 1, e);
            g.A = dKF;
            EpiAxpyRow e2{dC512, coefk, b.k, DH};
            gemm::launch<true, false, gemm::A_PLAIN>(st, g, 1, e2);
        }
        DDSP_LAUNCH_CHECK(ctx);
        const float* dqkv[3] = {dB512, dC512, dV512};
        const float* pw[3] = {L.q_w, L.k_w, L.v_w};
        float* gw[3] = {GLP(q_w), GLP(k_w), GLP(v_w)};
        float* gb[3] = {GLP(q_b), GLP(k_b), GLP(v_b)};
        for (int i = 0; i < 3; ++i) {
            if ((rc = layer_grads(ctx, st, dqkv[i], INNER, INNER, b.y, D, D, 1, (int)Fr, M, wpart, cpart, xs, gw[i], D, gb[i], 0, dfp))) return rc;
            dgrad(st, dqkv[i], INNER, pw[i], INNER, D, M, dA, i > 0, wt_qkv[l][i]);                               // d_y (summed)
        }
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(rows_g), dim3(256), 0, st, b.x_in, L.norm_w, dA, dX, M, dX, gx);
        if ((rc = colsum_pair(ctx, st, gx, dA, D, M, D, cpart, GLP(norm_w), GLP(norm_b)))) return rc;
        // dX now holds d x_in of this layer
#undef GLP
    }

    // ---- side embeddings (x0 = conv2 + Lin(lf0) + Lin(phase/pi) + Lin(vol) + spk) ----
    {
        ColsumJobs jb{};
        jb.n = 4;
        const float* ws[4] = {in.f0, in.phase, in.volume, nullptr};
        const int wm[4] = {2, 3, 1, 0};
        float* outs[4] = {G(f0_w), G(phase_w), G(volume_w), G(f0_b)};
        for (int i = 0; i < 4; ++i) {
            jb.X[i] = dX;
            jb.wsrc[i] = ws[i];
            jb.wmode[i] = wm[i];
            jb.out[i] = outs[i];
        }
        if ((rc = colsum_multi(ctx, st, jb, D, M, D, cpart))) return rc;
    }
    DDSP_HIP(ctx, hipMemcpyAsync(G(phase_b), G(f0_b), D * sizeof(float), hipMemcpyDeviceToDevice, st));
    DDSP_HIP(ctx, hipMemcpyAsync(G(volume_b), G(f0_b), D * sizeof(float), hipMemcpyDeviceToDevice, st));
    int* dev_err = nullptr;
    if ((rc = ddsp_dev_error_ptr(ctx, &dev_err))) return rc;
    // (dA is free between the last LayerNorm adjoint and the conv2 input gradient: it holds the per-utterance sums)
    hipLaunchKernelGGL(utterance_sum_kernel, dim3((unsigned)B), dim3(1024), 0, st, dX, (int)Fr, in.mix.n > 0 ? nullptr : in.spk_id,
                       in.n_spk_id, w.n_spk, dA, dev_err);
    hipLaunchKernelGGL(spk_table_grad_kernel, dim3((unsigned)w.n_spk), dim3(D), 0, st, dA, B, in.spk_id, in.n_spk_id, in.mix,
                       G(spk_table));
    // ---- prenet conv2: weight gradient over the three taps, input gradient as the flipped conv ----
    if ((rc = layer_grads(ctx, st, dX, D, D, bf.t2, D, D, 3, (int)Fr, M, wpart, cpart, xs, pk, 3 * D, G(prenet_conv2_b),
                          w.causal ? -1 : 0))) return rc;
    hipLaunchKernelGGL(unpack_conv3_kernel, dim3(grid_for((int64_t)D * D * 3)), dim3(256), 0, st, pk, D, D, G(prenet_conv2_w));
    hipLaunchKernelGGL(pack_conv3_transposed_kernel, dim3(grid_for((int64_t)D * D * 3)), dim3(256), 0, st, w.prenet_conv2_w, D, D, w2t);
    {
        gemm::Args g = gemm::make(dX, D, w2t, 3 * D, (int)M, D, 3 * D);
        g.Fr = (int)Fr;
        g.Cin = D;
        g.tap_shift = w.causal ? 1 : 0;   // the adjoint of taps (-2, -1, 0) reads frames (0, +1, +2) of dX
        if (int rc = ddsp_zero_page(ctx, &g.zeros)) return rc;
        gemm::EpiStore e{dA, D, nullptr, 1, 0, 0};
        gemm::launch<true, true, gemm::A_CONV3>(st, g, 1, e);                                      // d_t2
    }
    // ---- GroupNorm + LeakyReLU ----
    hipLaunchKernelGGL(groupnorm_bwd_stats_kernel, dim3(4, (unsigned)B), dim3(256), 0, st, bf.t1, bf.t2, dA, bf.gst,
                       w.prenet_gn_w, (int)Fr, gbst);
    hipLaunchKernelGGL(groupnorm_bwd_apply_kernel, dim3(grid_for(M * D)), dim3(256), 0, st, bf.t1, bf.t2, dA, bf.gst, gbst,
                       w.prenet_gn_w, M, (int)Fr, dX, gx, dA);                                     // dX = d_t1, dA = d_gn
    if ((rc = colsum_pair(ctx, st, gx, dA, D, M, D, cpart, G(prenet_gn_w), G(prenet_gn_b)))) return rc;
    // ---- prenet conv1 (the units carry no gradient) ----
    if ((rc = layer_grads(ctx, st, dX, D, D, in.units, w.n_unit, w.n_unit, 3, (int)Fr, M, wpart, cpart, xs, pk, 3 * w.n_unit,
                          G(prenet_conv1_b), w.causal ? -1 : 0))) return rc;
    hipLaunchKernelGGL(unpack_conv3_kernel, dim3(grid_for((int64_t)D * w.n_unit * 3)), dim3(256), 0, st, pk, D, w.n_unit,
                       G(prenet_conv1_w));
    // the deferred partial sums of the blocks' Linear layers, one launch
    PROF(PF_U2C_BWD, 0, 4.0 * df.used, flush_deferred(st, df));
    DDSP_LAUNCH_CHECK(ctx);
#undef G
    return DDSP_OK;
}

extern "C" int ddsp_unit2ctrl_bwd(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* wp, const float* units,
                                  const float* f0_frames, const float* phase_frames, const float* volume,
                                  const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                                  const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, const float* d_ctrl,
                                  const ddsp_u2c_weights* grads_host, float* ctrl_out) {
    U2CInputs in;
    int rc = check_inputs(ctx, wp, units, f0_frames, phase_frames, volume, spk_id, n_spk_id, mix_ids_host, mix_w_host,
                          n_mix, B, Fr, in);
    if (rc) return rc;
    DDSP_REQUIRE(ctx, d_ctrl && grads_host, "ddsp_unit2ctrl_bwd: null argument");
    if ((rc = ddsp_take_dev_error(ctx))) return rc;
    if (B == 0) return DDSP_OK;
    DDSP_ENTER_DEVICE(ctx);
    return u2c_backward(ctx, (hipStream_t)stream, *wp, *grads_host, in, B, Fr, d_ctrl, ctrl_out, nullptr, 0);
}

// ---- a training step's pair: a forward that leaves its activations in a caller-owned region, a backward that starts from them ----
extern "C" int64_t ddsp_unit2ctrl_keep_bytes(const ddsp_u2c_weights* wp, int64_t B, int64_t Fr) {
    if (!wp || B < 0 || Fr < 1) return -1;
    U2CBufs bf;
    Arena dry{nullptr, true, 0, 0};
    plan_forward(dry, bf, *wp, B, Fr, true);
    return (int64_t)dry.total + 256;
}

extern "C" int ddsp_unit2ctrl_fwd_keep(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* wp, const float* units,
                                       const float* f0_frames, const float* phase_frames, const float* volume,
                                       const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                                       const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, void* keep,
                                       int64_t keep_bytes, float* ctrl) {
    U2CInputs in;
    int rc = check_inputs(ctx, wp, units, f0_frames, phase_frames, volume, spk_id, n_spk_id, mix_ids_host, mix_w_host,
                          n_mix, B, Fr, in);
    if (rc) return rc;
    DDSP_REQUIRE(ctx, ctrl && keep && ((uintptr_t)keep % 256) == 0, "ddsp_unit2ctrl_fwd_keep: null ctrl / keep, or keep not 256-byte aligned");
    if ((rc = ddsp_take_dev_error(ctx))) return rc;
    if (B == 0) return DDSP_OK;
    DDSP_ENTER_DEVICE(ctx);
    U2CBufs bf;
    Arena k{ctx, false, 0, 0, (char*)keep, (size_t)keep_bytes};
    plan_forward(k, bf, *wp, B, Fr, true);
    DDSP_REQUIRE(ctx, !k.rc, "ddsp_unit2ctrl_fwd_keep: keep_bytes is smaller than ddsp_unit2ctrl_keep_bytes says");
    return u2c_forward(ctx, (hipStream_t)stream, *wp, in, bf, ctrl);
}

extern "C" int ddsp_unit2ctrl_bwd_kept(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* wp, const float* units,
                                       const float* f0_frames, const float* phase_frames, const float* volume,
                                       const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                                       const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, void* keep,
                                       int64_t keep_bytes, const float* d_ctrl, const ddsp_u2c_weights* grads_host) {
    U2CInputs in;
    int rc = check_inputs(ctx, wp, units, f0_frames, phase_frames, volume, spk_id, n_spk_id, mix_ids_host, mix_w_host,
                          n_mix, B, Fr, in);
    if (rc) return rc;
    DDSP_REQUIRE(ctx, d_ctrl && grads_host && keep && ((uintptr_t)keep % 256) == 0, "ddsp_unit2ctrl_bwd_kept: null argument or keep not 256-byte aligned");
    if ((rc = ddsp_take_dev_error(ctx))) return rc;
    if (B == 0) return DDSP_OK;
    DDSP_ENTER_DEVICE(ctx);
    return u2c_backward(ctx, (hipStream_t)stream, *wp, *grads_host, in, B, Fr, d_ctrl, nullptr, keep, (size_t)keep_bytes);
}
