// a4 (attention part), large batches with DDSP_MATH_SPLIT_BF16: the fused Performer attention of performer_attn.hip
// (ddsp/pcmer.py:69-77,123-159) re-tiled for the bf16 matrix pipe.
//
// Every fp32 product is formed from bf16 pieces with fp32 accumulation (v_mfma_f32_32x32x16_bf16):
//   * the random-feature projections S = x P^T (x = k or q) enter an exponential, so BOTH operands are cut into THREE
//     bf16 pieces (24 mantissa bits) and six piece products are kept (a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0; the dropped
//     ones are ~2^-24 of the product): fp32-class accuracy at 6/16 of the fp32-MFMA matrix time;
//   * the second products (k'^T v and ctx^T q') use two pieces and three products like the Linear layers (~4e-6).
// One workgroup owns one (utterance, head) - a feature group of it on the key side, a frame group on the query side - so
// every shared operand is converted ONCE per workgroup into an LDS image laid out in MFMA operand order (a wave reads a
// fragment as one conflict-free ds_read_b128 per lane), instead of once per wave as in the fp32 kernels, whose waves read
// their operands straight from L1/L2.  A 32x32 tile of projected values lives in the accumulator (16 registers), is
// exponentiated in place and its registers 8s..8s+7 ARE the 8 k-slots of step s of the next product (element j of lane
// half h = row 16s + 8(j>>2) + 4h + (j&3)); the other operand of that product is staged in the same slot order.
//   K kernel, workgroup (b, h) = 9 product waves (one per feature tile of 32) + 3 staging waves; per 32-frame tile the
//     staging waves convert k (3 pieces), -0.5 dn^2 |k|^2 (one extra k-step against constant ones) and v (2 pieces, gathered
//     in slot order) into the other half of the double-buffered LDS stage, their loads one tile ahead, while the product
//     waves do  S = k P_jt^T, k' = exp2(S), ks += column sums, ctx_jt += k'^T v  on the current half.
//     ctx leaves as bf16 hi/lo pieces in the operand order the Q kernel stages (so its staging is a plain DMA).
//   Q kernel, workgroup (b, h, group of up to 6 frame tiles of 32), wave = one frame tile with its q pieces in registers;
//     per feature tile: P_jt pieces (prepared once per forward) and ctx_jt pieces stream global -> LDS with
//     global_load_lds (no registers, no conversion); wave: S^T = P_jt q^T, running row maximum, q' in place, D += q'.ks,
//     out^T += ctx_jt^T q'.
// 32-wide tiles pad 266 features to 288 and 172 frames to 192 (21 % more MACs than 16-wide tiles would; they halve the LDS
// bytes per MAC and the instruction count, and the matrix time is no longer what bounds these kernels).
#include "performer_attn.h"

#include <atomic>
#include <stdlib.h>

namespace {

constexpr int H = 8, DH = 64, INNER = 512, NF = 266;
constexpr int NJT = PERFORMER_NJT32;           // 9 feature tiles of 32
constexpr int KG = 9, KS_WAVES = 3;           // K workgroup: one product wave per feature tile + three staging waves
constexpr int QW = 6;                          // frame tiles (= waves) per Q workgroup
constexpr float DN = 0.35355339059327373f;     // 64^-0.25
constexpr float RATIO = 0.06131393394849658f;  // 266^-0.5
constexpr float EPS = 1e-4f;
constexpr float LOG2E = 1.44269504088896341f;
constexpr float PSCALE = DN * LOG2E;           // exponents are carried in the base-2 domain (v_exp_f32 is exp2)
constexpr float NEG_HALF = -0.5f * DN * DN * LOG2E;
constexpr float KOFF = EPS * LOG2E - 4.0276409f;   // eps (inside the key exponential) and log2(266^-0.5)
constexpr float MASKED = -1000.f;

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// x[0..7] -> NP bf16 pieces (round to nearest even each time, remainder carried on): p[0] + p[1] (+ p[2]) ~ x
template <int NP>
__device__ __forceinline__ void split(const float (&x)[8], u32x4 (&p)[NP]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        f32x2_t r = {x[2 * d], x[2 * d + 1]};
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const uint32_t hw = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
            p[q][d] = hw;
            if (q + 1 < NP)
                r = r - (f32x2_t){__builtin_bit_cast(float, hw << 16), __builtin_bit_cast(float, hw & 0xffff0000u)};
        }
    }
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// ---- P3: the projection matrix, scaled by dn*log2(e), as three bf16 pieces in operand order ---------------------------------
// chunk (jt, s, piece, h, i) = 8 bf16 = pieces of PSCALE * P[32 jt + i][16 s + 8 h + 0..7] (zero rows for features >= 266);
// flat index (((jt*4 + s)*3 + piece)*2 + h)*32 + i, 16 bytes each: PERFORMER_P3_BYTES per layer.
__global__ void __launch_bounds__(256) performer_p3_kernel(const float* __restrict__ P0, const float* __restrict__ P1,
                                                           const float* __restrict__ P2, uint4* __restrict__ p3) {
    const int t = blockIdx.x * 256 + threadIdx.x;            // (jt, s, h, i); blockIdx.y = layer
    if (t >= NJT * 4 * 2 * 32) return;
    const float* P = blockIdx.y == 0 ? P0 : (blockIdx.y == 1 ? P1 : P2);
    p3 += (size_t)blockIdx.y * (PERFORMER_P3_BYTES / 16);
    const int i = t & 31, h = (t >> 5) & 1, s = (t >> 6) & 3, jt = t >> 8;
    const int j = 32 * jt + i;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = j < NF ? PSCALE * P[(int64_t)j * DH + 16 * s + 8 * h + e] : 0.f;
    u32x4 pc[3];
    split<3>(x, pc);
#pragma unroll
    for (int q = 0; q < 3; ++q) p3[(((jt * 4 + s) * 3 + q) * 2 + h) * 32 + i] = __builtin_bit_cast(uint4, pc[q]);
}

// ---- K side ---------------------------------------------------------------------------------------------------------------
// LDS stage (16-byte units): Kp[(s*3 + piece)*2 + h][i] (768) | Kd[h][i] (64) | Vp[((step*2 + piece)*2 + ctile)*2 + h][n] (512)
constexpr int K_KP = 0, K_KD = 768, K_VP = 832, K_STAGE = 1344;      // 21504 bytes per stage

// ABL (measurement only, tools/attn_ablate.py): 1 = the staging wave does no work, 2 = no first product, 4 = no second product
template <int ABL>
__global__ void __launch_bounds__(64 * (KG + KS_WAVES), 3) performer_kv_bf16_kernel(const float* __restrict__ k,
                                                                             const float* __restrict__ v,
                                                                             const uint4* __restrict__ p3, int Fr,
                                                                             uint4* __restrict__ ctxS, float* __restrict__ ks) {
    __shared__ uint4 lds[2 * K_STAGE];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int n_ft = (Fr + 31) / 32;

    if (wave >= KG) {
        // ---- the staging waves: convert frame tile ft + 1 into the other LDS stage while waves 0..8 multiply tile ft ----
        // 512 tasks per tile in 8 rounds of 64 (one per lane); staging wave ws takes rounds ws, ws + 3, ws + 6:
        //   rounds 0..3  k chunk: frame fi = 8 r + (lane & 7), channels 8 c8 .. 8 c8 + 7, c8 = lane >> 3 (two 16-byte loads)
        //                -> three pieces, one 16-byte LDS write each; |k|^2 of a frame = the 8 lanes 8 apart.  (The 8 lanes of
        //                a write group hold 8 consecutive frames = 8 consecutive 16-byte slots; with c8 = lane & 7 they held
        //                the 8 chunks of ONE frame, whose slots lie 512 bytes apart - the same banks: 39 % of this kernel's
        //                LDS cycles were conflicts, r02_e_pmc_sq_synth.txt);
        //   rounds 4..7  v chunk: channel ch = lane, frame octet o = r - 4 = (step, half): the 8 frames of its slots (eight
        //                4-byte loads, each one 256-byte row segment across the wave) -> two pieces, already in slot order.
        const int ws = wave - KG;
        const float* kb = k + ((int64_t)b * Fr) * INNER + h * DH;
        const float* vb = v + ((int64_t)b * Fr) * INNER + h * DH;
        if (ws == 0) lds[(lane >> 5) * K_STAGE + K_KD + 32 + l31] = uint4{0u, 0u, 0u, 0u};   // half 1 of the extra A operand
        float sink = 0.f;
        auto load_round = [&](int r, int ft, float (&x)[8]) __attribute__((always_inline)) {
            if (ABL & 16) {              // no loads
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = 0.25f * (float)(lane + e + ft);
                return;
            }
            if (r < 4) {
                const int fi = 8 * r + (lane & 7), f = 32 * ft + fi;
                const float* src = kb + (int64_t)(f < Fr ? f : Fr - 1) * INNER + 8 * (lane >> 3);
                const f32x4_t a0 = *(const f32x4_t*)src, a1 = *(const f32x4_t*)(src + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = f < Fr ? (e < 4 ? a0[e & 3] : a1[e & 3]) : 0.f;
            } else if (r < 8) {
                const int o = r - 4;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * ft + 16 * (o >> 1) + 4 * (o & 1) + (e & 3) + 8 * (e >> 2);
                    x[e] = f < Fr ? vb[(int64_t)f * INNER + lane] : 0.f;
                }
            }
        };
        auto store_round = [&](int r, int ft, const float (&x)[8]) __attribute__((always_inline)) {
            if (ABL & 8) {               // loads only: consume the values
#pragma unroll
                for (int e = 0; e < 8; ++e) sink += x[e];
                return;
            }
            uint4* st = lds + (ft & 1) * K_STAGE;
            if (r < 4) {
                const int fi = 8 * r + (lane & 7), c8 = lane >> 3;
                u32x4 pc[3];
                split<3>(x, pc);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    st[K_KP + (((c8 >> 1) * 3 + q) * 2 + (c8 & 1)) * 32 + fi] = __builtin_bit_cast(uint4, pc[q]);
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
                ss += __shfl_xor(ss, 8, 64);
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                if (c8 == 0) {
                    const float d[8] = {NEG_HALF * ss, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    u32x4 dp[3];
                    split<3>(d, dp);       // element 0 of the three pieces -> slots 0, 1, 2 of half 0
                    const u32x4 o_ = {(dp[0][0] & 0xffffu) | (dp[1][0] << 16), dp[2][0] & 0xffffu, 0u, 0u};
                    st[K_KD + fi] = __builtin_bit_cast(uint4, o_);
                }
            } else if (r < 8) {
                const int o = r - 4;
                u32x4 pc[2];
                split<2>(x, pc);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    st[K_VP + ((((o >> 1) * 2 + q) * 2 + lh) * 2 + (o & 1)) * 32 + l31] = __builtin_bit_cast(uint4, pc[q]);
            }
        };
        // tile ft is converted and stored, then the loads of tile ft + 1 are issued, then the barrier: the loads are in
        // flight while this wave waits for the product waves (one register set; the wait before the conversion is a plain
        // vmcnt(0) - with two sets the compiler cannot count the loads of the two tiles apart and waits for both)
        float x[3][8];
        if (!(ABL & 1)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) load_round(ws + 3 * i, 0, x[i]);
        }
#pragma unroll 1
        for (int ft = 0; ft < n_ft; ++ft) {
            if (!(ABL & 1)) {
#pragma unroll
                for (int i = 0; i < 3; ++i) store_round(ws + 3 * i, ft, x[i]);
                if (ft + 1 < n_ft) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) load_round(ws + 3 * i, ft + 1, x[i]);
                }
            }
            __syncthreads();           // tile ft is staged (and the product waves have finished tile ft - 1)
        }
        __syncthreads();
        if ((ABL & 8) && sink == 12345.678f) ks[0] = sink;
        return;
    }

    const int jt = wave;
    // this wave's P pieces: B operand (feature l31, half lh) of step s
    bf16x8 pb[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            pb[s][q] = __builtin_bit_cast(bf16x8, p3[(((jt * 4 + s) * 3 + q) * 2 + lh) * 32 + l31]);
    // B operand of the extra step: ones against the three pieces of -0.5 dn^2 |k|^2 (slots 0..2 of half 0)
    u32x4 ones = {0u, 0u, 0u, 0u};
    if (lh == 0) {
        ones[0] = 0x3f803f80u;      // bf16 1.0 | 1.0
        ones[1] = 0x00003f80u;      // 1.0 | 0
    }
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float ksum = 0.f;
    const float c_init = 32 * jt + l31 < NF ? KOFF : MASKED;      // pad features are switched off in the C init

    __syncthreads();                   // tile 0 is staged
#pragma unroll 1
    for (int ft = 0; ft < n_ft; ++ft) {
        const uint4* st = lds + (ft & 1) * K_STAGE;
        // ---- S = k P^T (+ diag, + log2 of the constant factors): six piece products per k-step ----
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = c_init;
        if (ft == n_ft - 1) {                   // ragged last tile: frames past the end are switched off
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * ft + (r & 3) + 8 * (r >> 2) + 4 * lh >= Fr) S[r] = MASKED;
        }
        // (the LDS reads of step s + 1 are issued before the products of step s: left to itself the compiler reads each
        // fragment right before its first use and waits for it)
        bf16x8 a[2][3];
        const uint4* kp = st + K_KP + lh * 32 + l31;
#pragma unroll
        for (int q = 0; q < 3; ++q) a[0][q] = __builtin_bit_cast(bf16x8, kp[q * 64]);
        const bf16x8 ad = __builtin_bit_cast(bf16x8, st[K_KD + lh * 32 + l31]);
#pragma unroll
        for (int s = 0; s < ((ABL & 2) ? 0 : 4); ++s) {
            if (s + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 3; ++q) a[(s + 1) & 1][q] = __builtin_bit_cast(bf16x8, kp[((s + 1) * 3 + q) * 64]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8(&c)[3] = a[s & 1];
            S = MFMA_BF16(c[2], pb[s][0], S);      // smallest terms first
            S = MFMA_BF16(c[0], pb[s][2], S);
            S = MFMA_BF16(c[1], pb[s][1], S);
            S = MFMA_BF16(c[1], pb[s][0], S);
            S = MFMA_BF16(c[0], pb[s][1], S);
            S = MFMA_BF16(c[0], pb[s][0], S);
            __builtin_amdgcn_sched_barrier(0);
        }
        S = MFMA_BF16(ad, as_bf(ones), S);
        // fragments of the second product's first (step, channel tile) pair: in flight under the exponentials
        const uint4* vp = st + K_VP + lh * 32 + l31;
        bf16x8 vf[2][2];
        vf[0][0] = __builtin_bit_cast(bf16x8, vp[0]);
        vf[0][1] = __builtin_bit_cast(bf16x8, vp[2 * 64]);
        __builtin_amdgcn_sched_barrier(0);
        float kf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            kf[r] = __builtin_amdgcn_exp2f(S[r]);
            ksum += kf[r];
        }
        // ---- ctx_jt += k'^T v: the accumulator registers are the k-slots ----
#pragma unroll
        for (int step = 0; step < ((ABL & 4) ? 0 : 2); ++step) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = kf[8 * step + e];
            u32x4 ap[2];
            split<2>(x, ap);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int pr = step * 2 + ct;              // pair index; its fragments were read one pair ago
                if (pr + 1 < 4) {
                    const int ns = (pr + 1) >> 1, nc = (pr + 1) & 1;
                    vf[(pr + 1) & 1][0] = __builtin_bit_cast(bf16x8, vp[(((ns * 2 + 0) * 2 + nc) * 2) * 32]);
                    vf[(pr + 1) & 1][1] = __builtin_bit_cast(bf16x8, vp[(((ns * 2 + 1) * 2 + nc) * 2) * 32]);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[ct] = MFMA_BF16(as_bf(ap[1]), vf[pr & 1][0], acc[ct]);
                acc[ct] = MFMA_BF16(as_bf(ap[0]), vf[pr & 1][1], acc[ct]);
                acc[ct] = MFMA_BF16(as_bf(ap[0]), vf[pr & 1][0], acc[ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();               // this stage may be overwritten; the next one is complete
    }
    // ---- results: ks (288 | column-sum parts 9 x 64 | ks-sum parts 9), ctx pieces in the Q kernel's operand order ----
    float* kr = ks + (int64_t)bh * PERFORMER_KS_STRIDE;
    ksum += __shfl_xor(ksum, 32, 64);
    if (lh == 0) kr[32 * jt + l31] = ksum;                 // pad features are exact zeros
    float ktot = ksum;
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) ktot += __shfl_xor(ktot, m, 64);
    if (lane == 0) kr[PERFORMER_OFF_KPART32 + jt] = ktot;
    uint4* cd = ctxS + ((int64_t)bh * NJT + jt) * 512;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cs += acc[ct][r];
        cs += __shfl_xor(cs, 32, 64);
        if (lh == 0) kr[PERFORMER_OFF_CPART32 + jt * DH + 32 * ct + l31] = cs;
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = acc[ct][8 * step + e];
            u32x4 pc[2];
            split<2>(x, pc);
#pragma unroll
            for (int q = 0; q < 2; ++q) cd[(((step * 2 + q) * 2 + ct) * 2 + lh) * 32 + l31] = __builtin_bit_cast(uint4, pc[q]);
        }
    }
}

// ---- Q side ---------------------------------------------------------------------------------------------------------------
// LDS stage (16-byte units): Pp[(s*3 + piece)*2 + h][i] (768) | Cp[((step*2 + piece)*2 + ctile)*2 + h][i] (512) | ks (8)
constexpr int Q_PP = 0, Q_CP = 768, Q_KS = 1280, Q_STAGE = 1344;     // 21504 bytes per stage (1 KiB aligned pieces)

// ABL (measurement only): 1 = no DMA staging, 2 = no first product, 4 = no second product
template <int ABL>
__global__ void __launch_bounds__(64 * QW, 3) performer_q_bf16_kernel(const float* __restrict__ q, const uint4* __restrict__ p3,
                                                                   const uint4* __restrict__ ctxS, const float* __restrict__ ks,
                                                                   int Fr, int n_fg, float* __restrict__ attn, int out_split) {
    __shared__ __attribute__((aligned(1024))) uint4 lds[2 * Q_STAGE];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int fg = slot % n_fg, bh = (slot / n_fg) * 8 + xcd, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int ft = fg * QW + wave;
    const int frame = 32 * ft + l31;                    // this lane's frame (column of S^T and of out^T)
    const bool frame_ok = frame < Fr;

    // stage jt: 20 one-KiB pieces (12 of P3, 8 of ctx) by global_load_lds, piece = wave + QW*i; ks by wave 0
    const float* kr = ks + (int64_t)bh * PERFORMER_KS_STRIDE;
    const uint4* cd = ctxS + (int64_t)bh * NJT * 512;
    auto issue = [&](int jt, int buf) {
#pragma unroll
        for (int i = 0; i < (20 + QW - 1) / QW; ++i) {
            const int piece = wave + QW * i;
            if (piece < 20 && !(ABL & 1)) {
                const uint4* src = piece < 12 ? p3 + (jt * 12 + piece) * 64 + lane : cd + (jt * 8 + piece - 12) * 64 + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + buf * Q_STAGE + piece * 64),
                                                 16, 0, 0);
            }
        }
    };
    issue(0, 0);


    // q pieces of this wave's 32 frames: B operand (frame l31, half lh); |q|^2 per frame
    bf16x8 qb[4][3];
    float ss = 0.f;
    {
        const float* qr = q + ((int64_t)b * Fr + (frame_ok ? frame : Fr - 1)) * INNER + h * DH + 8 * lh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4_t a0 = *(const f32x4_t*)(qr + 16 * s), a1 = *(const f32x4_t*)(qr + 16 * s + 4);
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] = frame_ok ? (e < 4 ? a0[e] : a1[e - 4]) : 0.f;
                ss = fmaf(x[e], x[e], ss);
            }
            u32x4 pc[3];
            split<3>(x, pc);
#pragma unroll
            for (int p = 0; p < 3; ++p) qb[s][p] = as_bf(pc[p]);
        }
    }
    ss += __shfl_xor(ss, 32, 64);
    // (the row maximum runs over the projections WITHOUT the diagonal term, ddsp/pcmer.py:73)
    const float diag2 = -NEG_HALF * ss;
    // eps terms: column sums of ctx (lane = channel) and the sum of ks, from the parts the K kernel left per feature tile
    float cs_lane = 0.f, ks_tot = lane < NJT ? kr[PERFORMER_OFF_KPART32 + lane] : 0.f;
#pragma unroll
    for (int t = 0; t < NJT; ++t) cs_lane += kr[PERFORMER_OFF_CPART32 + t * DH + lane];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) ks_tot += __shfl_xor(ks_tot, m, 64);

    float m_run = -3.0e38f, Dacc = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;

#pragma unroll 1
    for (int jt = 0; jt < NJT; ++jt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                  // stage jt has landed for every wave; stage jt-1 is free
        if (jt + 1 < NJT) issue(jt + 1, (jt + 1) & 1);
        // ks of the tile's features (row 4 lh + 8 g4 + e of the tile) straight from L2, in flight under the first product
        f32x4_t kv[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) kv[g4] = *(const f32x4_t*)(kr + 32 * jt + 8 * g4 + 4 * lh);
        const uint4* st = lds + (jt & 1) * Q_STAGE;
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        if (jt == NJT - 1) {                              // features 266..287 are switched off in the C init
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * jt + (r & 3) + 8 * (r >> 2) + 4 * lh >= NF) S[r] = MASKED;
        }
        bf16x8 a[2][3];                                   // (reads of step s + 1 ahead of the products of step s)
        const uint4* pp = st + Q_PP + lh * 32 + l31;
#pragma unroll
        for (int p = 0; p < 3; ++p) a[0][p] = __builtin_bit_cast(bf16x8, pp[p * 64]);
#pragma unroll
        for (int s = 0; s < ((ABL & 2) ? 0 : 4); ++s) {
            if (s + 1 < 4) {
#pragma unroll
                for (int p = 0; p < 3; ++p) a[(s + 1) & 1][p] = __builtin_bit_cast(bf16x8, pp[((s + 1) * 3 + p) * 64]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8(&c)[3] = a[s & 1];
            S = MFMA_BF16(c[2], qb[s][0], S);
            S = MFMA_BF16(c[0], qb[s][2], S);
            S = MFMA_BF16(c[1], qb[s][1], S);
            S = MFMA_BF16(c[1], qb[s][0], S);
            S = MFMA_BF16(c[0], qb[s][1], S);
            S = MFMA_BF16(c[0], qb[s][0], S);
            __builtin_amdgcn_sched_barrier(0);
        }
        // fragments of the second product's first (step, channel tile) pair: in flight under the exponentials
        const uint4* cp = st + Q_CP + lh * 32 + l31;
        bf16x8 cf[2][2];
        cf[0][0] = __builtin_bit_cast(bf16x8, cp[0]);
        cf[0][1] = __builtin_bit_cast(bf16x8, cp[2 * 64]);
        __builtin_amdgcn_sched_barrier(0);
        // running row maximum over the features seen so far: what the frame has accumulated is rescaled when it rises
        float tmax = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, S[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float sc = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        const float dm = m_run + diag2;
        if (__any(sc != 1.0f)) {                          // (after the first tiles the maximum rarely rises)
            Dacc *= sc;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] *= sc;
        }
        float u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = __builtin_amdgcn_exp2f(S[r] - dm);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int e = 0; e < 4; ++e) Dacc = fmaf(u[4 * g4 + e], kv[g4][e], Dacc);
#pragma unroll
        for (int step = 0; step < ((ABL & 4) ? 0 : 2); ++step) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = u[8 * step + e];
            u32x4 bp[2];
            split<2>(x, bp);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int pr = step * 2 + ct;
                if (pr + 1 < 4) {
                    const int ns = (pr + 1) >> 1, nc = (pr + 1) & 1;
                    cf[(pr + 1) & 1][0] = __builtin_bit_cast(bf16x8, cp[(((ns * 2 + 0) * 2 + nc) * 2) * 32]);
                    cf[(pr + 1) & 1][1] = __builtin_bit_cast(bf16x8, cp[(((ns * 2 + 1) * 2 + nc) * 2) * 32]);
                }
                __builtin_amdgcn_sched_barrier(0);
                o[ct] = MFMA_BF16(cf[pr & 1][1], as_bf(bp[0]), o[ct]);
                o[ct] = MFMA_BF16(cf[pr & 1][0], as_bf(bp[1]), o[ct]);
                o[ct] = MFMA_BF16(cf[pr & 1][0], as_bf(bp[0]), o[ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    Dacc += __shfl_xor(Dacc, 32, 64);
    const float D = RATIO * fmaf(EPS, ks_tot, Dacc);
    const float dinv = RATIO / (D + 1e-8f);
    // o[ct][r] = out^T[channel 32 ct + (r&3) + 8 (r>>2) + 4 lh][frame l31]: four consecutive channels per register group
    float* orow = attn + ((int64_t)b * Fr + frame) * INNER + h * DH;
    const float cs_eps = EPS * cs_lane;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int ch = 32 * ct + 8 * g4 + 4 * lh;
            f32x4_t res;
#pragma unroll
            for (int e = 0; e < 4; ++e) res[e] = (o[ct][4 * g4 + e] + __shfl(cs_eps, ch + e, 64)) * dinv;
            if (out_split) {
                // A operand of the out-projection's split-bf16 GEMM: the lane pair (lh = 0, 1) owns one group of 8 channels
                const ddsp_u32x4 sp = ddsp_split4_pair(res, lh != 0, 32);
                if (frame_ok) *(ddsp_u32x4*)(orow + ch) = sp;
            } else if (frame_ok) {
                *(f32x4_t*)(orow + ch) = res;
            }
        }
}


// ---- both sides in ONE kernel (round 3) -------------------------------------------------------------------------------------
// One workgroup of 12 wavefronts per (utterance, head) runs the key side and then the query side; the context matrix (288 x 64
// as bf16 hi / lo pieces, 72 KiB) and the key sums never leave the LDS - the round-2 pair wrote 38 MB of ctx pieces per layer to
// HBM and streamed them back (86.5 MB per launch each, 0.05 of the bf16 peak), and paid a kernel boundary in between.
//   key phase    = performer_kv_bf16_kernel: 9 product waves (one per 32-feature tile) + 3 staging waves per 32-frame tile;
//                  its results go to LDS regions C (ctx pieces, in the query side's operand order) and S (ks, eps parts).
//   query phase  : passes of up to 6 frame tiles; TWO waves per frame tile, one taking the even, one the odd feature tiles
//                  (5 rounds; each round stages the projection pieces of its two feature tiles by global_load_lds), each with
//                  its own running row maximum; the odd wave hands (m, D, out^T) to the even one through the LDS, which
//                  rescales both to the common maximum - the maximum is the exact row maximum over all 266 features, as the
//                  reference's `softmax_kernel` needs it for the weight of its eps term (pcmer.py:140-141).
// LDS (16-byte units): A [0, 3072) key-phase stages / query-phase projection stages (2 x 24 pieces) and the hand-over;
// C [3072, 7680) ctx pieces; S [7680, 7900): ks[288] | cpart[9][64] | kpart[9..] floats.
constexpr int F_A = 0, F_C = 3072, F_S = 7680, F_UNITS = 7900;
constexpr int F_QBUF = 1536;                    // one query-phase stage: 24 one-KiB pieces
constexpr int FQ_TILES = 6;                     // frame tiles per query pass (two waves each)
static_assert(2 * K_STAGE <= F_C && 2 * F_QBUF <= F_C, "region A holds either phase's stages");

__global__ void __launch_bounds__(64 * (KG + KS_WAVES), 3) performer_fused_bf16_kernel(const float* __restrict__ q,
                                                                                 const float* __restrict__ k,
                                                                                 const float* __restrict__ v,
                                                                                 const uint4* __restrict__ p3, int Fr,
                                                                                 float* __restrict__ attn, int out_split) {
    extern __shared__ __attribute__((aligned(1024))) uint4 lds[];
    float* const sks = (float*)(lds + F_S);                   // ks[288] | cpart[9][64] | kpart[9]
    constexpr int S_CPART = 288, S_KPART = 288 + NJT * DH;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int n_ft = (Fr + 31) / 32;

    // ============================== key phase ==============================
    if (wave >= KG) {
        // staging waves: see performer_kv_bf16_kernel
        const int ws = wave - KG;
        const float* kb = k + ((int64_t)b * Fr) * INNER + h * DH;
        const float* vb = v + ((int64_t)b * Fr) * INNER + h * DH;
        if (ws == 0) lds[F_A + (lane >> 5) * K_STAGE + K_KD + 32 + l31] = uint4{0u, 0u, 0u, 0u};
        auto load_round = [&](int r, int ft, float (&x)[8]) __attribute__((always_inline)) {
            if (r < 4) {
                const int fi = 8 * r + (lane & 7), f = 32 * ft + fi;
                const float* src = kb + (int64_t)(f < Fr ? f : Fr - 1) * INNER + 8 * (lane >> 3);
                const f32x4_t a0 = *(const f32x4_t*)src, a1 = *(const f32x4_t*)(src + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = f < Fr ? (e < 4 ? a0[e & 3] : a1[e & 3]) : 0.f;
            } else if (r < 8) {
                const int o = r - 4;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * ft + 16 * (o >> 1) + 4 * (o & 1) + (e & 3) + 8 * (e >> 2);
                    x[e] = f < Fr ? vb[(int64_t)f * INNER + lane] : 0.f;
                }
            }
        };
        auto store_round = [&](int r, int ft, const float (&x)[8]) __attribute__((always_inline)) {
            uint4* st = lds + F_A + (ft & 1) * K_STAGE;
            if (r < 4) {
                const int fi = 8 * r + (lane & 7), c8 = lane >> 3;
                u32x4 pc[3];
                split<3>(x, pc);
#pragma unroll
                for (int qq = 0; qq < 3; ++qq)
                    st[K_KP + (((c8 >> 1) * 3 + qq) * 2 + (c8 & 1)) * 32 + fi] = __builtin_bit_cast(uint4, pc[qq]);
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
                ss += __shfl_xor(ss, 8, 64);
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                if (c8 == 0) {
                    const float d[8] = {NEG_HALF * ss, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    u32x4 dp[3];
                    split<3>(d, dp);
                    const u32x4 o_ = {(dp[0][0] & 0xffffu) | (dp[1][0] << 16), dp[2][0] & 0xffffu, 0u, 0u};
                    st[K_KD + fi] = __builtin_bit_cast(uint4, o_);
                }
            } else if (r < 8) {
                const int o = r - 4;
                u32x4 pc[2];
                split<2>(x, pc);
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                    st[K_VP + ((((o >> 1) * 2 + qq) * 2 + lh) * 2 + (o & 1)) * 32 + l31] = __builtin_bit_cast(uint4, pc[qq]);
            }
        };
        float x[3][8];
#pragma unroll
        for (int i = 0; i < 3; ++i) load_round(ws + 3 * i, 0, x[i]);
#pragma unroll 1
        for (int ft = 0; ft < n_ft; ++ft) {
#pragma unroll
            for (int i = 0; i < 3; ++i) store_round(ws + 3 * i, ft, x[i]);
            if (ft + 1 < n_ft) {
#pragma unroll
                for (int i = 0; i < 3; ++i) load_round(ws + 3 * i, ft + 1, x[i]);
            }
            __syncthreads();
        }
        __syncthreads();
    } else {
        const int jt = wave;
        bf16x8 pb[4][3];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int qq = 0; qq < 3; ++qq)
                pb[s2][qq] = __builtin_bit_cast(bf16x8, p3[(((jt * 4 + s2) * 3 + qq) * 2 + lh) * 32 + l31]);
        u32x4 ones = {0u, 0u, 0u, 0u};
        if (lh == 0) {
            ones[0] = 0x3f803f80u;
            ones[1] = 0x00003f80u;
        }
        f32x16 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        float ksum = 0.f;
        const float c_init = 32 * jt + l31 < NF ? KOFF : MASKED;
        __syncthreads();                   // tile 0 is staged
#pragma unroll 1
        for (int ft = 0; ft < n_ft; ++ft) {
            const uint4* st = lds + F_A + (ft & 1) * K_STAGE;
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = c_init;
            if (ft == n_ft - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * ft + (r & 3) + 8 * (r >> 2) + 4 * lh >= Fr) S[r] = MASKED;
            }
            bf16x8 a[2][3];
            const uint4* kp = st + K_KP + lh * 32 + l31;
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) a[0][qq] = __builtin_bit_cast(bf16x8, kp[qq * 64]);
            const bf16x8 ad = __builtin_bit_cast(bf16x8, st[K_KD + lh * 32 + l31]);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                if (s2 + 1 < 4) {
#pragma unroll
                    for (int qq = 0; qq < 3; ++qq) a[(s2 + 1) & 1][qq] = __builtin_bit_cast(bf16x8, kp[((s2 + 1) * 3 + qq) * 64]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8(&c)[3] = a[s2 & 1];
                S = MFMA_BF16(c[2], pb[s2][0], S);
                S = MFMA_BF16(c[0], pb[s2][2], S);
                S = MFMA_BF16(c[1], pb[s2][1], S);
                S = MFMA_BF16(c[1], pb[s2][0], S);
                S = MFMA_BF16(c[0], pb[s2][1], S);
                S = MFMA_BF16(c[0], pb[s2][0], S);
                __builtin_amdgcn_sched_barrier(0);
            }
            S = MFMA_BF16(ad, as_bf(ones), S);
            const uint4* vp = st + K_VP + lh * 32 + l31;
            bf16x8 vf[2][2];
            vf[0][0] = __builtin_bit_cast(bf16x8, vp[0]);
            vf[0][1] = __builtin_bit_cast(bf16x8, vp[2 * 64]);
            __builtin_amdgcn_sched_barrier(0);
            float kf[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                kf[r] = __builtin_amdgcn_exp2f(S[r]);
                ksum += kf[r];
            }
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = kf[8 * step + e];
                u32x4 ap[2];
                split<2>(x, ap);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int pr = step * 2 + ct;
                    if (pr + 1 < 4) {
                        const int ns = (pr + 1) >> 1, nc = (pr + 1) & 1;
                        vf[(pr + 1) & 1][0] = __builtin_bit_cast(bf16x8, vp[(((ns * 2 + 0) * 2 + nc) * 2) * 32]);
                        vf[(pr + 1) & 1][1] = __builtin_bit_cast(bf16x8, vp[(((ns * 2 + 1) * 2 + nc) * 2) * 32]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[ct] = MFMA_BF16(as_bf(ap[1]), vf[pr & 1][0], acc[ct]);
                    acc[ct] = MFMA_BF16(as_bf(ap[0]), vf[pr & 1][1], acc[ct]);
                    acc[ct] = MFMA_BF16(as_bf(ap[0]), vf[pr & 1][0], acc[ct]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
        // results stay in the LDS: ks | eps parts (region S), ctx pieces in the query side's operand order (region C)
        ksum += __shfl_xor(ksum, 32, 64);
        if (lh == 0) sks[32 * jt + l31] = ksum;
        float ktot = ksum;
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) ktot += __shfl_xor(ktot, m, 64);
        if (lane == 0) sks[S_KPART + jt] = ktot;
        uint4* cd = lds + F_C + jt * 512;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) cs += acc[ct][r];
            cs += __shfl_xor(cs, 32, 64);
            if (lh == 0) sks[S_CPART + jt * DH + 32 * ct + l31] = cs;
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = acc[ct][8 * step + e];
                u32x4 pc[2];
                split<2>(x, pc);
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) cd[(((step * 2 + qq) * 2 + ct) * 2 + lh) * 32 + l31] = __builtin_bit_cast(uint4, pc[qq]);
            }
        }
    }
    __syncthreads();      // ctx, ks and the eps parts are complete; region A is free

    // ============================== query phase ==============================
    const int ftw = wave % FQ_TILES, par = wave / FQ_TILES;           // frame tile of the pass, feature-tile parity
    constexpr int ROUNDS = (NJT + 1) / 2;
    // stage round r: projection pieces of feature tiles 2r (pieces 0..11) and 2r + 1 (12..23), two pieces per wave
    auto issue = [&](int r, int buf) {
        // (r made opaque: the compiler otherwise computes the ten 64-bit source addresses of all rounds once per kernel, keeps
        // them across the pass loop and spills them - 41 registers of scratch in the first build)
        asm volatile("" : "+s"(r));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 12 * i;
            const int jt = 2 * r + piece / 12;
            if (jt < NJT)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p3 + (jt * 12 + piece % 12) * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds + F_A + buf * F_QBUF + piece * 64),
                                                 16, 0, 0);
        }
    };
#pragma unroll 1
    for (int pass = 0; pass * FQ_TILES < n_ft; ++pass) {
        const int ft = pass * FQ_TILES + ftw;
        const int frame = 32 * ft + l31;
        const bool frame_ok = frame < Fr;
        issue(0, 0);
        bf16x8 qb[4][3];
        float ss = 0.f;
        {
            const float* qr = q + ((int64_t)b * Fr + (frame_ok ? frame : Fr - 1)) * INNER + h * DH + 8 * lh;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const f32x4_t a0 = *(const f32x4_t*)(qr + 16 * s2), a1 = *(const f32x4_t*)(qr + 16 * s2 + 4);
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    x[e] = frame_ok ? (e < 4 ? a0[e] : a1[e - 4]) : 0.f;
                    ss = fmaf(x[e], x[e], ss);
                }
                u32x4 pc[3];
                split<3>(x, pc);
#pragma unroll
                for (int pp2 = 0; pp2 < 3; ++pp2) qb[s2][pp2] = as_bf(pc[pp2]);
            }
        }
        ss += __shfl_xor(ss, 32, 64);
        const float diag2 = -NEG_HALF * ss;
        float m_run = -3.0e38f, Dacc = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
#pragma unroll 1
        for (int r = 0; r < ROUNDS; ++r) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // round r has landed for every wave; the other buffer is free
            if (r + 1 < ROUNDS) issue(r + 1, (r + 1) & 1);
            const int jt = 2 * r + par;
            if (jt < NJT) {
                const uint4* st = lds + F_A + (r & 1) * F_QBUF + par * 768;
                f32x16 S;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) S[rr] = 0.f;
                if (jt == NJT - 1) {
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr)
                        if (32 * jt + (rr & 3) + 8 * (rr >> 2) + 4 * lh >= NF) S[rr] = MASKED;
                }
                const uint4* pp = st + lh * 32 + l31;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    // (one fragment set: the two other waves of the SIMD cover the LDS round trip; a second set, as in the
                    // stand-alone query kernel, spills 10 registers here)
                    bf16x8 c[3];
#pragma unroll
                    for (int pq = 0; pq < 3; ++pq) c[pq] = __builtin_bit_cast(bf16x8, pp[(s2 * 3 + pq) * 64]);
                    __builtin_amdgcn_sched_barrier(0);
                    S = MFMA_BF16(c[2], qb[s2][0], S);
                    S = MFMA_BF16(c[0], qb[s2][2], S);
                    S = MFMA_BF16(c[1], qb[s2][1], S);
                    S = MFMA_BF16(c[1], qb[s2][0], S);
                    S = MFMA_BF16(c[0], qb[s2][1], S);
                    S = MFMA_BF16(c[0], qb[s2][0], S);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const uint4* cp = lds + F_C + jt * 512 + lh * 32 + l31;
                bf16x8 cf[2][2];
                cf[0][0] = __builtin_bit_cast(bf16x8, cp[0]);
                cf[0][1] = __builtin_bit_cast(bf16x8, cp[2 * 64]);
                __builtin_amdgcn_sched_barrier(0);
                float tmax = S[0];
#pragma unroll
                for (int rr = 1; rr < 16; ++rr) tmax = fmaxf(tmax, S[rr]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run, tmax);
                const float sc = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                const float dm = m_run + diag2;
                if (__any(sc != 1.0f)) {
                    Dacc *= sc;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) o[c][rr] *= sc;
                }
                float u[16];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) u[rr] = __builtin_amdgcn_exp2f(S[rr] - dm);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4_t kv = *(const f32x4_t*)(sks + 32 * jt + 8 * g4 + 4 * lh);    // ks of the tile's features
#pragma unroll
                    for (int e = 0; e < 4; ++e) Dacc = fmaf(u[4 * g4 + e], kv[e], Dacc);
                }
#pragma unroll
                for (int step = 0; step < 2; ++step) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = u[8 * step + e];
                    u32x4 bp[2];
                    split<2>(x, bp);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int pr = step * 2 + ct;
                        if (pr + 1 < 4) {
                            const int ns = (pr + 1) >> 1, nc = (pr + 1) & 1;
                            cf[(pr + 1) & 1][0] = __builtin_bit_cast(bf16x8, cp[(((ns * 2 + 0) * 2 + nc) * 2) * 32]);
                            cf[(pr + 1) & 1][1] = __builtin_bit_cast(bf16x8, cp[(((ns * 2 + 1) * 2 + nc) * 2) * 32]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        o[ct] = MFMA_BF16(cf[pr & 1][1], as_bf(bp[0]), o[ct]);
                        o[ct] = MFMA_BF16(cf[pr & 1][0], as_bf(bp[1]), o[ct]);
                        o[ct] = MFMA_BF16(cf[pr & 1][0], as_bf(bp[0]), o[ct]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        // hand-over odd -> even wave of a frame tile through region A (free after the last round's barrier ... of the NEXT
        // barrier: every wave has finished reading its last stage when it passes the first barrier below)
        Dacc += __shfl_xor(Dacc, 32, 64);
        float* xch = (float*)(lds + F_A);                     // [ftw][18][64] floats per half
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if (par == 1) {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) xch[(ftw * 18 + rr) * 64 + lane] = o[half][rr];
                if (half == 0) {
                    xch[(ftw * 18 + 16) * 64 + lane] = m_run;
                    xch[(ftw * 18 + 17) * 64 + lane] = Dacc;
                }
            }
            __syncthreads();
            if (par == 0) {
                if (half == 0) {
                    const float m1 = xch[(ftw * 18 + 16) * 64 + lane], D1 = xch[(ftw * 18 + 17) * 64 + lane];
                    const float m = fmaxf(m_run, m1);
                    const float s0 = __builtin_amdgcn_exp2f(m_run - m), s1 = __builtin_amdgcn_exp2f(m1 - m);
                    Dacc = Dacc * s0 + D1 * s1;
                    // keep the two scale factors for both halves in m_run (s0) and ss (s1)
                    m_run = s0;
                    ss = s1;
                }
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) o[half][rr] = o[half][rr] * m_run + xch[(ftw * 18 + rr) * 64 + lane] * ss;
            }
        }
        if (par == 0) {
            // eps terms: column sums of ctx (lane = channel) and the sum of ks, from the parts of the key phase
            float cs_lane = 0.f, ks_tot = lane < NJT ? sks[S_KPART + lane] : 0.f;
#pragma unroll
            for (int t = 0; t < NJT; ++t) cs_lane += sks[S_CPART + t * DH + lane];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) ks_tot += __shfl_xor(ks_tot, m, 64);
            const float cs_eps = EPS * cs_lane;
            const float D = RATIO * fmaf(EPS, ks_tot, Dacc);
            const float dinv = RATIO / (D + 1e-8f);
            float* orow = attn + ((int64_t)b * Fr + frame) * INNER + h * DH;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int ch = 32 * ct + 8 * g4 + 4 * lh;
                    f32x4_t res;
#pragma unroll
                    for (int e = 0; e < 4; ++e) res[e] = (o[ct][4 * g4 + e] + __shfl(cs_eps, ch + e, 64)) * dinv;
                    if (out_split) {
                        const ddsp_u32x4 sp = ddsp_split4_pair(res, lh != 0, 32);
                        if (frame_ok) *(ddsp_u32x4*)(orow + ch) = sp;
                    } else if (frame_ok) {
                        *(f32x4_t*)(orow + ch) = res;
                    }
                }
        }
        __syncthreads();      // region A is free for the next pass's stages
    }
}

}  // namespace

void performer_p3(hipStream_t st, const float* P0, const float* P1, const float* P2, void* p3) {
    const unsigned layers = P2 ? 3u : (P1 ? 2u : 1u);
    hipLaunchKernelGGL(performer_p3_kernel, dim3((NJT * 4 * 2 * 32 + 255) / 256, layers), dim3(256), 0, st, P0, P1, P2,
                       (uint4*)p3);
}

void performer_kv_bf16(hipStream_t st, const float* k, const float* v, const void* p3, int B, int Fr, float* ctxS, float* ks,
                       int ablate) {
#define KV_ABL(A)                                                                                                      \
    if (ablate == A) {                                                                                                 \
        hipLaunchKernelGGL(performer_kv_bf16_kernel<A>, dim3((unsigned)(B * H)), dim3(64 * (KG + KS_WAVES)), 0, st, k, \
                           v, (const uint4*)p3, Fr, (uint4*)ctxS, ks);                                                 \
        return;                                                                                                        \
    }
    KV_ABL(1) KV_ABL(2) KV_ABL(4) KV_ABL(6) KV_ABL(7) KV_ABL(14) KV_ABL(22)
#undef KV_ABL
    hipLaunchKernelGGL(performer_kv_bf16_kernel<0>, dim3((unsigned)(B * H)), dim3(64 * (KG + KS_WAVES)), 0, st, k, v,
                       (const uint4*)p3, Fr, (uint4*)ctxS, ks);
}

void performer_q_bf16(hipStream_t st, const float* q, const void* p3, const float* ctxS, const float* ks, int B, int Fr,
                      float* attn, int ablate, int out_split) {
    const int n_fg = ((Fr + 31) / 32 + QW - 1) / QW;
#define Q_ABL(A)                                                                                                       \
    if (ablate == A) {                                                                                                 \
        hipLaunchKernelGGL(performer_q_bf16_kernel<A>, dim3((unsigned)(n_fg * B * H)), dim3(64 * QW), 0, st, q,        \
                           (const uint4*)p3, (const uint4*)ctxS, ks, Fr, n_fg, attn, out_split);                       \
        return;                                                                                                        \
    }
    Q_ABL(1) Q_ABL(2) Q_ABL(4) Q_ABL(6) Q_ABL(7)
#undef Q_ABL
    hipLaunchKernelGGL(performer_q_bf16_kernel<0>, dim3((unsigned)(n_fg * B * H)), dim3(64 * QW), 0, st, q, (const uint4*)p3,
                       (const uint4*)ctxS, ks, Fr, n_fg, attn, out_split);
}

// DDSP_ATTN_FUSED=0: the round-2 kernel pair (measurement aid)
bool performer_fused_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("DDSP_ATTN_FUSED");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

hipError_t performer_fused_bf16(hipStream_t st, const float* q, const float* k, const float* v, const void* p3, int B, int Fr,
                                float* attn, int out_split) {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void*)performer_fused_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_UNITS * 16);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(performer_fused_bf16_kernel, dim3((unsigned)(B * H)), dim3(64 * (KG + KS_WAVES)), F_UNITS * 16, st, q, k, v,
                       (const uint4*)p3, Fr, attn, out_split);
    return hipSuccess;
}

// ---- building block exposed for unit tests and measurements: one attention of pcmer.py:221-251 without its Linear layers ----
extern "C" int ddsp_performer_attention(ddsp_ctx* ctx, void* stream, const float* q, const float* k, const float* v,
                                        const float* proj, int64_t B, int64_t Fr, float* out, int math) {
    DDSP_REQUIRE(ctx, ctx && q && k && v && proj && out, "ddsp_performer_attention: null argument");
    DDSP_REQUIRE(ctx, B >= 1 && B <= 4096 && Fr >= 1 && B * Fr < (1 << 26), "ddsp_performer_attention: bad shape");
    if (math == DDSP_ATTENTION_CAUSAL) {     // the causal network's attention (`c: true`): chunked kernel, fp32 products
        hipStream_t stc = (hipStream_t)stream;
        DDSP_ENTER_DEVICE(ctx);
        ddsp_prof_begin(ctx, stc, PF_U2C_GEMM_ATTNOUT);
        performer_causal(stc, q, k, v, proj, (int)B, (int)Fr, out);
        ddsp_prof_end(ctx, stc, 10.0 * B * Fr * H * NF * DH, 4.0 * B * Fr * 4 * INNER);
        DDSP_LAUNCH_CHECK(ctx);
        return DDSP_OK;
    }
    // (math = 100 + ablation mask: measurement aid of tools/attn_ablate.py, results are meaningless)
    const int ablate = math >= 100 ? math - 100 : 0;
    const bool want_pair = math >= 100;      // 100 + mask: the round-2 kernel pair (mask 0: nothing switched off)
    if (math >= 100) math = DDSP_MATH_SPLIT_BF16;
    DDSP_REQUIRE(ctx, math == DDSP_MATH_FP32 || math == DDSP_MATH_SPLIT_BF16, "ddsp_performer_attention: unknown math");
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const size_t n_cx = (size_t)B * H * (PERFORMER_CTXS_FLOATS > PERFORMER_LDJ * DH ? PERFORMER_CTXS_FLOATS : PERFORMER_LDJ * DH);
    const size_t n_ks = (size_t)B * H * PERFORMER_KS_STRIDE;
    int rc = ddsp_scratch_reserve_bytes(ctx, (n_cx + n_ks) * sizeof(float) + PERFORMER_P3_BYTES + 4096);
    if (rc) return rc;
    ddsp_scratch_reset(ctx);
    float *cx = nullptr, *ksb = nullptr;
    void* p3 = nullptr;
    if ((rc = ddsp_scratch_get(ctx, n_cx * sizeof(float), (void**)&cx))) return rc;
    if ((rc = ddsp_scratch_get(ctx, n_ks * sizeof(float), (void**)&ksb))) return rc;
    if ((rc = ddsp_scratch_get(ctx, PERFORMER_P3_BYTES, &p3))) return rc;
    if (math == DDSP_MATH_SPLIT_BF16 && performer_fused_enabled() && !want_pair) {
        performer_p3(st, proj, nullptr, nullptr, p3);
        ddsp_prof_begin(ctx, st, PF_U2C_GEMM_CTX);
        DDSP_HIP(ctx, performer_fused_bf16(st, q, k, v, p3, (int)B, (int)Fr, out, 0));
        ddsp_prof_end(ctx, st, 8.0 * B * Fr * H * NF * DH, 4.0 * B * Fr * 4 * INNER);
    } else if (math == DDSP_MATH_SPLIT_BF16) {
        performer_p3(st, proj, nullptr, nullptr, p3);
        ddsp_prof_begin(ctx, st, PF_U2C_GEMM_CTX);
        performer_kv_bf16(st, k, v, p3, (int)B, (int)Fr, cx, ksb, ablate);
        ddsp_prof_end(ctx, st, 4.0 * B * Fr * H * NF * DH, 4.0 * B * Fr * 2 * INNER);
        ddsp_prof_begin(ctx, st, PF_U2C_GEMM_ATTNOUT);
        performer_q_bf16(st, q, p3, cx, ksb, (int)B, (int)Fr, out, ablate);
        ddsp_prof_end(ctx, st, 4.0 * B * Fr * H * NF * DH, 4.0 * B * Fr * 2 * INNER);
    } else {
        ddsp_prof_begin(ctx, st, PF_U2C_GEMM_CTX);
        performer_kv(st, k, v, proj, (int)B, (int)Fr, cx, ksb);
        ddsp_prof_end(ctx, st, 4.0 * B * Fr * H * NF * DH, 4.0 * B * Fr * 2 * INNER);
        ddsp_prof_begin(ctx, st, PF_U2C_GEMM_ATTNOUT);
        performer_q(st, q, proj, cx, ksb, (int)B, (int)Fr, out);
        ddsp_prof_end(ctx, st, 4.0 * B * Fr * H * NF * DH, 4.0 * B * Fr * 2 * INNER);
    }
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
