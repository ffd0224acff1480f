// a7: frame-varying FIR on the fp32 matrix pipe (Toeplitz-block products, v_mfma_f32_16x16x4_f32).
//
// Replaces ddsp/core.py:185-239 `_fft_convolve` + :147-182 `_crop_and_compensate_delay`.
// The reference frames the input at 50% overlap with a triangular window, multiplies spectra
// (FFT length 2*hop+n-1) and overlap-adds; any zero-padded length gives the same linear convolution,
// so this kernel evaluates it directly, output-stationary:
//
//   out[u] = sum_m sum_k  ir[min(m,Fr-1)][k] * xw_m[u + n/2 - k],   xw_m[t] = x[t] * tri(t - hop*(m-1))
//
// with tri the periodic Bartlett window of length 2*hop and m = 0..Fr (frame Fr reuses filter Fr-1).
//
// MFMA mapping.  Time is cut into columns of 16 samples.  For an output tile O[i][j] = out[16*(J0+j)+i]
// (16x16 = 256 samples, one wavefront) and a column shift q, the contribution of frame m is
//   O += A_q * B_q,  A_q[i][r] = ir_m[16q + n/2 + i - r] (a Toeplitz block),  B_q[r][j] = xw_m[16*(J0+j-q) + r],
// i.e. one K=16 product = four 16x16x4 MFMAs.  Only (m, q) pairs whose 16 input columns touch frame m's
// 64-column support are issued (about 80% of the issued MACs are useful; the rest multiply zero pad).
// LDS holds, per frame of the block: the filter row with 32 zeros either side, and the pre-windowed
// input transposed to [r][col] with row stride 112 (= 16 mod 32) plus a 2*(r>>1) column skew, which
// makes both the fill (lanes along t) and the B-operand fetch (lanes along col, rows r and r+1)
// conflict-free.  Outputs leave as one float4 per lane, 1 KiB contiguous per wavefront.
//
// Bound: fp32 matrix pipe (64 FLOP/clk/SIMD); algorithmic HBM bytes are 4 B in + 4 B out per sample plus
// 4*n/hop B of filter per sample.
#include "common.h"

namespace {

constexpr int HOP = 512;
constexpr int HOPC = HOP / 16;  // 32 columns per hop
constexpr int PADC = 16;
constexpr int RS = 112;         // row stride of the transposed input image (floats)
constexpr int XS = 16 * RS;     // floats per frame image
constexpr int IRPAD = 32;
constexpr int SEG = 4;          // segments (hops) per block: 8 wavefronts, one 256-sample tile each; 2 blocks per CU by LDS

__device__ __forceinline__ float noise_u(uint64_t seed, uint64_t idx) {
    // A stateless hash of the sample counter (every block that needs sample idx regenerates it).  Round 3: two rounds of a 32-bit
    // integer finaliser (multiply / xor-shift, constants of the "lowbias32" family) instead of the splitmix64 finaliser - its three
    // 64 x 64-bit multiplies were ~45 vector instructions per sample in the FIR kernel's staging waves, which regenerate every
    // sample twice (50 % overlapped frames); this is ~16.  The high words of counter and seed enter between the rounds.
    uint32_t x = (uint32_t)idx ^ (uint32_t)seed;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    x += (uint32_t)(idx >> 32) * 0x9E3779B9u + (uint32_t)(seed >> 32);
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);   // 24-bit U[0,1) like torch.rand fp32
    return u * 2.0f - 1.0f;                                   // exact in fp32
}

// acc[s] += sum_{q=qa..qb} A_q * B_q for the four k-steps s of a 16x16x16 Toeplitz-block product:
//   a = ap[16q + SA*4s], b = bp[s*bstep + SB*q]   (lane-constant parts already folded into ap / bp).
// Software-pipelined with two register sets: operands of shift q+1 are fetched from LDS before the MFMAs of
// shift q issue (sched_group_barrier pins [8 LDS reads][4 MFMAs]; hipcc otherwise sinks the reads).
// qa, qb MUST be wave-uniform (SGPR) values - see the readfirstlane note in the kernels.
template <int SA, int SB>
__device__ __forceinline__ void toeplitz_accumulate(const float* __restrict__ ap, const float* __restrict__ bp,
                                                    int bstep, int qa, int qb, f32x4 (&acc)[4]) {
    if (qa > qb) return;
    float a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a0[s] = ap[16 * qa + SA * 4 * s];
        b0[s] = bp[s * bstep + SB * qa];
    }
    int q = qa;
    for (; q + 1 <= qb; q += 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a1[s] = ap[16 * (q + 1) + SA * 4 * s];
            b1[s] = bp[s * bstep + SB * (q + 1)];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0[s], acc[s], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        const int q2 = (q + 2 <= qb) ? q + 2 : qb;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a0[s] = ap[16 * q2 + SA * 4 * s];
            b0[s] = bp[s * bstep + SB * q2];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b1[s], acc[s], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    if (q == qb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0[s], acc[s], 0, 0, 0);
    }
}

struct FirArgs {
    const float* audio;  // (B,T) or null
    int excitation;      // DDSP_EXC_*
    uint64_t seed;
    const float* ir;     // (B,Fr,n)
    const float* add_in; // (B,T) or null
    float* out;          // (B,T) or null: filtered signal
    float* out_sum;      // (B,T) or null: filtered signal + add_in
    int Fr, n;
    int q_lo, q_hi;      // column-shift range of the filter
    int nfr;             // frames staged per block
    int m_off;           // first staged frame relative to the block's first segment
    int irs;             // floats per staged filter row (n + 2*IRPAD, rounded)
};

__global__ void __launch_bounds__(64 * 2 * SEG) ltv_fir_kernel(FirArgs g) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave-uniform on purpose: loop bounds derived from it must live in SGPRs, otherwise hipcc treats the
    // MFMA loops as divergent and round-trips the accumulators through VGPRs (s_nop + v_accvgpr_read) per iteration
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int s0 = blockIdx.x * SEG;  // first segment of the block
    const int64_t T = (int64_t)g.Fr * HOP;
    const int c = g.n / 2;

    // frames whose 64-column support can reach this block's outputs: m in [m_lo, m_lo + nfr)
    // (input columns touched: [32*s0 - q_hi, 32*(s0+SEG) - 1 - q_lo]; m_off = -ceil(q_hi/32))
    const int m_lo = s0 + g.m_off;
    float* irs = lds;                          // [nfr][g.irs]
    float* xs = lds + (size_t)g.nfr * g.irs;   // [nfr][XS]

    // ---- stage filters and windowed input ----------------------------------------------------
    // Wavefront f stages frame m_lo + f on its own: zero image, filter row, windowed input, with every global load
    // of a step in flight before the first one is consumed (a block-wide loop over the frames paid one load
    // latency per frame, twice: ~40 us of every launch were staging).  LDS accesses of one wave are ordered, so the
    // only barrier is the one before the products.
    for (int f = wave; f < g.nfr; f += 2 * SEG) {
        const int m = m_lo + f;
        if (m < 0 || m > g.Fr) continue;               // the product loop skips these frames too
        float* xd = xs + (size_t)f * XS;
        // (1) zero the input image (pads must read as zero), 16 B per lane
        for (int i = lane; i < XS / 4; i += 64) ((f32x4*)xd)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (2) filter row: 32 zeros | n taps | zeros; rows of n floats are 8-byte aligned (n is even)
        {
            const int mi = m < g.Fr ? m : g.Fr - 1;
            const float2* src = (const float2*)(g.ir + ((int64_t)b * g.Fr + mi) * g.n);
            float2* dst = (float2*)(irs + (size_t)f * g.irs);
            const int per_row = g.irs / 2, half_n = g.n / 2;
            for (int base = 0; base < per_row; base += 64 * 8) {
                float2 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int k2 = base + 64 * k + lane - IRPAD / 2;
                    v[k] = float2{0.f, 0.f};
                    if (k2 >= 0 && k2 < half_n) v[k] = src[k2];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = base + 64 * k + lane;
                    if (i < per_row) dst[i] = v[k];
                }
            }
        }
        // (3) windowed input: aligned float4s of x, four per lane, scattered to image rows r..r+3 of one column
        {
            const int64_t tb = (int64_t)HOP * (m - 1);
            f32x4 x[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int z = 4 * (lane + 64 * k);
                const int64_t t = tb + z;
                ok[k] = t >= 0 && t < T;
                x[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok[k]) {
                    if (g.excitation == DDSP_EXC_GENERATE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[k][e] = noise_u(g.seed, (uint64_t)b * T + t + e);
                    } else {
                        x[k] = *(const f32x4*)(g.audio + (int64_t)b * T + t);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!ok[k]) continue;
                const int z = 4 * (lane + 64 * k);
                if (g.excitation == DDSP_EXC_UNIT_NOISE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[k][e] = __fadd_rn(__fmul_rn(x[k][e], 2.0f), -1.0f);
                }
                const int r0 = z & 15, col = z >> 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int zz = z + e, r = r0 + e;
                    const float w = (zz < HOP) ? (float)zz * (1.0f / HOP) : (float)(2 * HOP - zz) * (1.0f / HOP);
                    xd[r * RS + PADC + col + 2 * (r >> 1)] = x[k][e] * w;
                }
            }
        }
    }
    __syncthreads();

    const int seg = s0 + (wave >> 1);
    if (seg >= g.Fr) return;
    const int J0 = HOPC * seg + 16 * (wave & 1);  // first output column of this wavefront's tile
    const int li = lane & 15, lk = lane >> 4;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane-constant parts of the operand addresses: k-step s uses row r = 4s + lk, so
    //   A index = a_base + 16q - 4s,   B index = b_base + s*(4*RS + 4) - q     (immediate offsets in s)
    const int a_base = IRPAD + c + li - lk;
    const int b_base = lk * RS + PADC + 2 * (lk >> 1) + li;
    constexpr int BSTEP = 4 * RS + 4;

    for (int f = 0; f < g.nfr; ++f) {
        const int m = m_lo + f;
        if (m < 0 || m > g.Fr) continue;
        const int cbase = HOPC * (m - 1);  // absolute column of the frame image's column 0
        int qa = J0 - (cbase + 2 * HOPC) + 1;
        int qb = J0 + 15 - cbase;
        if (qa < g.q_lo) qa = g.q_lo;
        if (qb > g.q_hi) qb = g.q_hi;
        const float* ap = irs + (size_t)f * g.irs + a_base;
        const float* bp = xs + (size_t)f * XS + (J0 - cbase) + b_base;
        toeplitz_accumulate<-1, -1>(ap, bp, BSTEP, qa, qb, acc);
    }
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4*(lane >> 4) + reg
    const int64_t u = (int64_t)b * T + 16 * (int64_t)(J0 + li) + 4 * lk;
    if (g.out) *(f32x4*)(g.out + u) = o;
    if (g.out_sum) {
        const f32x4 ad = *(const f32x4*)(g.add_in + u);
        *(f32x4*)(g.out_sum + u) = ad + o;
    }
}


// ---- forward, split-bf16 products ---------------------------------------------------------------------------
// Same sum as ltv_fir_kernel, with each fp32 product formed from three bf16 MFMAs (hi*hi + lo*hi + hi*lo, fp32
// accumulate; ~4e-6 relative error) on v_mfma_f32_16x16x32_bf16: one instruction covers a column
// shift of 32 samples, A_Q[i][r] = ir[32Q + c + i - r], B_Q[r][j] = xw_m[16*(J0+j) - 32Q + r], r = 0..31.
// The operands are split once, while staging:
//   * the windowed input of a frame in natural order as two bf16 planes (hi, lo) with 272 zeros either side, so
//     a lane's 8 consecutive r values are one aligned 16-byte read, the 64 lanes of a wave reading 1 KiB contiguously;
//   * the filter row reversed, S[y] = ir[n-1-(y-FPL)], as dwords of two neighbours in two copies one tap apart
//     (C0[d] = S[2d],S[2d+1]; C1[d] = S[2d+1],S[2d+2]) for hi and lo: a lane's 8 taps start at an offset whose parity is
//     fixed by its row i, so it reads 4 dwords from "its" copy.  The copies sit 16 banks apart (fd = 16 mod 32).
// The matrix pipe needs 48 cycles per (tile, Q); the LDS reads (8 ds_read_b32 + 2 ds_read_b128 per tile and Q = 24 LDS
// cycles, shared by the four SIMDs) bound the product loop.
// Two kernels share the staging and product code:
//   * ltv_fir_bf16_kernel<SEGB>: one block = SEGB segments, stage everything, barrier, multiply (any shape);
//   * ltv_fir_bf16_march_kernel<SEGB, SW>: one block walks a long run of segments of one batch row, SEGB at a time, with
//     the frames in an LDS ring: 2*SEGB wavefronts multiply while SEGB*SW others stage the SEGB frames the next step
//     adds, so no frame is staged twice and staging hides behind the products (large batches).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int FPL = 48, FPR = 80;      // zero taps either side of the reversed filter row
constexpr int XPAD = 272;              // zero samples either side of a frame image
constexpr int XPLANE = (2 * HOP + 2 * XPAD) / 2;  // dwords per image plane (784 = 16 mod 32)
constexpr int FKMAX = 6;               // filter dwords per lane and staging pass
constexpr int STAGE_FILTER = 1, STAGE_IMAGE = 2;

struct FirBfArgs {
    const float* audio;
    int excitation;
    uint64_t seed;
    const float* ir;
    const float* add_in;
    float* out;
    float* out_sum;
    int Fr, n;
    int Q_lo, Q_hi;   // shift range of the filter (units of 32 samples)
    int nfr, m_off;   // frames a step needs (SEGB + extra), first of them relative to the step's first segment
    int fd;           // dwords per filter copy
    int chunk;        // march kernel: segments per block (multiple of SEGB)
    int ring;         // march kernel: frames in the LDS ring (nfr + SEGB)
};

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// (a, b) -> dwords of two bf16: hi = (bf16(a), bf16(b)), lo = (bf16(a - hi_a), bf16(b - hi_b)), a in the low half.
// Two values per v_cvt_pk_bf16_f32 (round to nearest even), the hi parts widened back by a shift / a mask.
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){a, b}, bf16x2v));
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){a - ha, b - hb}, bf16x2v));
}

// One wavefront stages frame m of batch row b into fr = [C0h | C1h | C0l | C1l | Xh | Xl].  Every global load of a part
// is issued before its first use; LDS accesses of one wave are ordered, so the caller's barrier is the only one.
__device__ __forceinline__ void fir_bf16_stage(const FirBfArgs& g, int b, int m, uint32_t* fr, int lane, int parts) {
    const int fd = g.fd;
    const int64_t T = (int64_t)g.Fr * HOP;
    f32x4 x[4];
    if (parts & STAGE_IMAGE) {
        const int64_t tb = (int64_t)HOP * (m - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t t = tb + 4 * (lane + 64 * k);
            x[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t >= 0 && t < T) {
                if (g.excitation == DDSP_EXC_GENERATE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[k][e] = noise_u(g.seed, (uint64_t)b * T + t + e);
                } else {
                    x[k] = *(const f32x4*)(g.audio + (int64_t)b * T + t);
                    if (g.excitation == DDSP_EXC_UNIT_NOISE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[k][e] = __fadd_rn(__fmul_rn(x[k][e], 2.0f), -1.0f);
                    }
                }
            }
        }
    }
    if (parts & STAGE_FILTER) {
        // the row reversed, as (S[2d+1], S[2d]) pairs plus S[2d+2]
        const int mi = m < g.Fr ? m : g.Fr - 1;
        const float* src = g.ir + ((int64_t)b * g.Fr + mi) * g.n;
        for (int base = 0; base < fd; base += 64 * FKMAX) {
            float2 pr[FKMAX];
            float nx[FKMAX];
#pragma unroll
            for (int k = 0; k < FKMAX; ++k) {
                const int t1 = g.n - 2 - (2 * (base + 64 * k + lane) - FPL);   // tap held by S[2d+1]; S[2d] holds t1 + 1 (t1 is even)
                pr[k] = float2{0.f, 0.f};
                nx[k] = 0.f;
                if (t1 >= 0 && t1 < g.n) pr[k] = *(const float2*)(src + t1);
                if (t1 >= 1 && t1 <= g.n) nx[k] = src[t1 - 1];
            }
#pragma unroll
            for (int k = 0; k < FKMAX; ++k) {
                const int d = base + 64 * k + lane;
                if (d < fd) {
                    uint32_t h0, l0, h1, l1;
                    split_bf16_pair(pr[k].y, pr[k].x, h0, l0);   // S[2d], S[2d+1]
                    split_bf16_pair(pr[k].x, nx[k], h1, l1);     // S[2d+1], S[2d+2]
                    fr[d] = h0;
                    fr[fd + d] = h1;
                    fr[2 * fd + d] = l0;
                    fr[3 * fd + d] = l1;
                }
            }
        }
    }
    if (parts & STAGE_IMAGE) {
        uint32_t* xh = fr + 4 * fd;
        uint32_t* xl = xh + XPLANE;
        for (int i = lane; i < XPAD / 2; i += 64) {
            xh[i] = 0u;
            xl[i] = 0u;
            xh[XPAD / 2 + HOP + i] = 0u;
            xl[XPAD / 2 + HOP + i] = 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = 4 * (lane + 64 * k);
            float xw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // triangular window: passes k = 0, 1 are the rising half (z < hop), k = 2, 3 the falling one
                const int zz = z + e;
                const float w = (k < 2) ? (float)zz * (1.0f / HOP) : (float)(2 * HOP - zz) * (1.0f / HOP);
                xw[e] = x[k][e] * w;
            }
            uint32_t h0, l0, h1, l1;
            split_bf16_pair(xw[0], xw[1], h0, l0);
            split_bf16_pair(xw[2], xw[3], h1, l1);
            *(u32x2*)(xh + (XPAD + z) / 2) = (u32x2){h0, h1};
            *(u32x2*)(xl + (XPAD + z) / 2) = (u32x2){l0, l1};
        }
    }
}

// Product loops.  ah / al: the lane's filter pointers for Q = 0 (hi, lo), bh / bl: its image pointers for Q = 0.
// Three operand sets in rotation, reads of shift Q+2 issued before the MFMAs of shift Q (prefetch indices clamp to Qb:
// a repeated read, never a stray one).  Qa, Qb wave-uniform.
// One tile:
__device__ __forceinline__ void fir_bf16_run1(const uint32_t* ah, const uint32_t* al, const uint32_t* bh, const uint32_t* bl,
                                              int Qa, int Qb, f32x4 (&acc)[3]) {
    if (Qa > Qb) return;
    struct Ops { u32x4 ah, al, bh, bl; };
    auto fetch = [&](int Q, Ops& o) {
        const uint32_t* pa = ah - 16 * Q;
        const uint32_t* pl = al - 16 * Q;
        o.ah = (u32x4){pa[0], pa[1], pa[2], pa[3]};
        o.al = (u32x4){pl[0], pl[1], pl[2], pl[3]};
        o.bh = *(const u32x4*)(bh - 16 * Q);
        o.bl = *(const u32x4*)(bl - 16 * Q);
    };
    auto mm = [&](const Ops& o) {
        const bf16x8 a_h = __builtin_bit_cast(bf16x8, o.ah), a_l = __builtin_bit_cast(bf16x8, o.al);
        const bf16x8 b_h = __builtin_bit_cast(bf16x8, o.bh), b_l = __builtin_bit_cast(bf16x8, o.bl);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l, acc[2], 0, 0, 0);
    };
    auto upto = [&](int Q) { return Q < Qb ? Q : Qb; };
    Ops o0, o1, o2;
    fetch(Qa, o0);
    fetch(upto(Qa + 1), o1);
    int Q = Qa;
    for (; Q + 2 <= Qb; Q += 3) {
        fetch(Q + 2, o2);
        mm(o0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        fetch(upto(Q + 3), o0);
        mm(o1);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        fetch(upto(Q + 4), o1);
        mm(o2);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    }
    if (Q <= Qb) mm(o0);
    if (Q + 1 <= Qb) mm(o1);
}
// Two neighbouring tiles (the second 256 samples = 128 dwords further on) sharing the filter operand: 4 + 4 LDS read
// instructions (32 LDS cycles) for 6 MFMAs instead of 2 x (4 + 2) (48 cycles).
__device__ __forceinline__ void fir_bf16_run2(const uint32_t* ah, const uint32_t* al, const uint32_t* bh, const uint32_t* bl,
                                              int Qa, int Qb, f32x4 (&acc0)[3], f32x4 (&acc1)[3]) {
    if (Qa > Qb) return;
    struct Ops { u32x4 ah, al, bh0, bl0, bh1, bl1; };
    auto fetch = [&](int Q, Ops& o) {
        const uint32_t* pa = ah - 16 * Q;
        const uint32_t* pl = al - 16 * Q;
        o.ah = (u32x4){pa[0], pa[1], pa[2], pa[3]};
        o.al = (u32x4){pl[0], pl[1], pl[2], pl[3]};
        o.bh0 = *(const u32x4*)(bh - 16 * Q);
        o.bl0 = *(const u32x4*)(bl - 16 * Q);
        o.bh1 = *(const u32x4*)(bh + 128 - 16 * Q);
        o.bl1 = *(const u32x4*)(bl + 128 - 16 * Q);
    };
    auto mm = [&](const Ops& o) {
        const bf16x8 a_h = __builtin_bit_cast(bf16x8, o.ah), a_l = __builtin_bit_cast(bf16x8, o.al);
        const bf16x8 b_h0 = __builtin_bit_cast(bf16x8, o.bh0), b_l0 = __builtin_bit_cast(bf16x8, o.bl0);
        const bf16x8 b_h1 = __builtin_bit_cast(bf16x8, o.bh1), b_l1 = __builtin_bit_cast(bf16x8, o.bl1);
        acc0[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h0, acc0[0], 0, 0, 0);
        acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h1, acc1[0], 0, 0, 0);
        acc0[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h0, acc0[1], 0, 0, 0);
        acc1[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h1, acc1[1], 0, 0, 0);
        acc0[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l0, acc0[2], 0, 0, 0);
        acc1[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l1, acc1[2], 0, 0, 0);
    };
    auto upto = [&](int Q) { return Q < Qb ? Q : Qb; };
    Ops o0, o1, o2;
    fetch(Qa, o0);
    fetch(upto(Qa + 1), o1);
    int Q = Qa;
    for (; Q + 2 <= Qb; Q += 3) {
        fetch(Q + 2, o2);
        mm(o0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        fetch(upto(Q + 3), o0);
        mm(o1);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        fetch(upto(Q + 4), o1);
        mm(o2);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
    if (Q <= Qb) mm(o0);
    if (Q + 1 <= Qb) mm(o1);
}

// Shift range of frame m for the 256-sample tile whose first output sits at frame-relative position zt
__device__ __forceinline__ void fir_bf16_range(const FirBfArgs& g, int zt, int& Qa, int& Qb) {
    Qa = -floor_div(1023 - zt, 32);       // ceil((zt - 1023) / 32): shifts that reach the frame's 1024 samples
    Qb = floor_div(zt + 271, 32);
    if (Qa < g.Q_lo) Qa = g.Q_lo;
    if (Qb > g.Q_hi) Qb = g.Q_hi;
}

// acc += contribution of staged frame m (at fr) to the 256-sample tile whose first column is J0.
// a_off: the lane's filter offset (copy and dword for Q = 0), lb = 8*li + 4*lk its image offset.  J0, m wave-uniform.
__device__ __forceinline__ void fir_bf16_products(const FirBfArgs& g, const uint32_t* fr, int m, int J0, int a_off, int lb,
                                                  f32x4 (&acc)[3]) {
    const int fd = g.fd;
    const int zt = 16 * J0 - HOP * (m - 1);   // frame-relative position of the tile's first output
    int Qa, Qb;
    fir_bf16_range(g, zt, Qa, Qb);
    const uint32_t* ah = fr + a_off;
    const uint32_t* bh = fr + 4 * fd + (XPAD + zt) / 2 + lb;
    fir_bf16_run1(ah, ah + 2 * fd, bh, bh + XPLANE, Qa, Qb, acc);
}

// The same for the two tiles J0 and J0 + 16 (one segment): shifts that only one of them needs run as single-tile loops
// either side of the shared range (the second tile's range is the first one's moved up by 8 shifts, then clipped).
__device__ __forceinline__ void fir_bf16_products2(const FirBfArgs& g, const uint32_t* fr, int m, int J0, int a_off, int lb,
                                                   f32x4 (&acc0)[3], f32x4 (&acc1)[3]) {
    const int fd = g.fd;
    const int zt = 16 * J0 - HOP * (m - 1);
    int a0, b0, a1, b1;
    fir_bf16_range(g, zt, a0, b0);
    fir_bf16_range(g, zt + 256, a1, b1);
    const uint32_t* ah = fr + a_off;
    const uint32_t* al = ah + 2 * fd;
    const uint32_t* bh = fr + 4 * fd + (XPAD + zt) / 2 + lb;
    const uint32_t* bl = bh + XPLANE;
    fir_bf16_run1(ah, al, bh, bl, a0, b0 < a1 - 1 ? b0 : a1 - 1, acc0);
    fir_bf16_run2(ah, al, bh, bl, a1, b0, acc0, acc1);
    fir_bf16_run1(ah, al, bh + 128, bl + 128, b0 + 1 > a1 ? b0 + 1 : a1, b1, acc1);
}

// C/D map of the 16x16 MFMA: col = lane & 15, row = 4*(lane >> 4) + reg  ->  one float4 per lane
__device__ __forceinline__ void fir_bf16_store(const FirBfArgs& g, int b, int J0, int li, int lk, const f32x4 (&acc)[3]) {
    const int64_t T = (int64_t)g.Fr * HOP;
    const f32x4 o = acc[0] + (acc[1] + acc[2]);
    const int64_t u = (int64_t)b * T + 16 * (int64_t)(J0 + li) + 4 * lk;
    if (g.out) *(f32x4*)(g.out + u) = o;
    if (g.out_sum) {
        const f32x4 ad = *(const f32x4*)(g.add_in + u);
        *(f32x4*)(g.out_sum + u) = ad + o;
    }
}

template <int SEGB>
__global__ void __launch_bounds__(64 * 2 * SEGB, (SEGB <= 4 ? 4 : 3)) ltv_fir_bf16_kernel(FirBfArgs g) {
    extern __shared__ __align__(16) uint32_t ldsw[];
    constexpr int NWAVE = 2 * SEGB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, s0 = blockIdx.x * SEGB;
    const int fstride = 4 * g.fd + 2 * XPLANE;   // dwords per staged frame
    const int m_lo = s0 + g.m_off;

    for (int f = wave; f < g.nfr; f += NWAVE) {
        const int m = m_lo + f;
        if (m < 0 || m > g.Fr) continue;         // the product loop skips these frames too
        fir_bf16_stage(g, b, m, ldsw + (size_t)f * fstride, lane, STAGE_FILTER | STAGE_IMAGE);
    }
    __syncthreads();

    const int J0 = HOPC * s0 + 16 * wave;        // wave w owns the 256-sample tile starting at column J0
    if (J0 >= HOPC * g.Fr) return;
    const int li = lane & 15, lk = lane >> 4;
    // A: taps S[y0 + e], y0 = FPL + n-1 - c - li + 8*lk - 32Q  ->  copy (y0 & 1), dword (y0 >> 1)
    const int y00 = FPL + g.n - 1 - g.n / 2 - li + 8 * lk;
    const int a_off = (y00 & 1) * g.fd + (y00 >> 1);
    f32x4 acc[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < g.nfr; ++f) {
        const int m = m_lo + f;
        if (m < 0 || m > g.Fr) continue;
        fir_bf16_products(g, ldsw + (size_t)f * fstride, m, J0, a_off, 8 * li + 4 * lk, acc);
    }
    fir_bf16_store(g, b, J0, li, lk, acc);
}

// PAIR = false: a product wavefront owns one 256-sample tile.  PAIR = true: two product wavefronts own one segment (two
// neighbouring tiles sharing the filter operand), one taking the even, the other the odd frames of the step; the odd one
// hands its partial sums over through the LDS.  Either way 2*SEGB product wavefronts.
template <int SEGB, int SW, bool PAIR>
__global__ void __launch_bounds__(64 * (2 + SW) * SEGB) ltv_fir_bf16_march_kernel(FirBfArgs g) {
    extern __shared__ __align__(16) uint32_t ldsw[];
    constexpr int NCOMP = 2 * SEGB, NWAVE = (2 + SW) * SEGB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int hop0 = blockIdx.x * g.chunk;
    const int hop_end = hop0 + g.chunk < g.Fr ? hop0 + g.chunk : g.Fr;
    const int fstride = 4 * g.fd + 2 * XPLANE;
    const int m_base = hop0 + g.m_off;           // ring slot of frame m: (m - m_base) mod ring
    auto slot = [&](int m) { return ldsw + (size_t)((m - m_base) % g.ring) * fstride; };
    f32x4* handover = (f32x4*)(ldsw + (size_t)g.ring * fstride);   // PAIR: [SEGB][2 tiles][64 lanes] partial sums

    // prologue: the frames of the first step, spread over all wavefronts
    for (int f = wave; f < g.nfr; f += NWAVE) {
        const int m = m_base + f;
        if (m < 0 || m > g.Fr) continue;
        fir_bf16_stage(g, b, m, slot(m), lane, STAGE_FILTER | STAGE_IMAGE);
    }
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    const int y00 = FPL + g.n - 1 - g.n / 2 - li + 8 * lk;
    const int a_off = (y00 & 1) * g.fd + (y00 >> 1);
    for (int s0 = hop0; s0 < hop_end; s0 += SEGB) {
        f32x4 acc[PAIR ? 2 : 1][3];
        const int seg = wave >> 1, odd = wave & 1;          // PAIR: the wave's segment of the step and its share of the frames
        const int J0 = PAIR ? HOPC * (s0 + seg) : HOPC * s0 + 16 * wave;
        const bool mine = wave < NCOMP && J0 < HOPC * hop_end;
        if (wave >= NCOMP) {
            // staging wavefronts: frame j of the SEGB frames the next step adds (part = filter / image when SW == 2)
            const int sw = wave - NCOMP;
            const int m = s0 + g.m_off + g.nfr + sw / SW;
            if (s0 + SEGB < hop_end && m >= 0 && m <= g.Fr)
                fir_bf16_stage(g, b, m, slot(m), lane, SW == 1 ? (STAGE_FILTER | STAGE_IMAGE) : (sw % SW ? STAGE_IMAGE : STAGE_FILTER));
        } else if (mine) {
#pragma unroll
            for (int t = 0; t < (PAIR ? 2 : 1); ++t)
#pragma unroll
                for (int x = 0; x < 3; ++x) acc[t][x] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (PAIR) {
                for (int f = odd; f < g.nfr; f += 2) {
                    const int m = s0 + g.m_off + f;
                    if (m < 0 || m > g.Fr) continue;
                    fir_bf16_products2(g, slot(m), m, J0, a_off, 8 * li + 4 * lk, acc[0], acc[1]);
                }
                if (odd) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) handover[(seg * 2 + t) * 64 + lane] = acc[t][0] + (acc[t][1] + acc[t][2]);
                }
            } else {
                for (int f = 0; f < g.nfr; ++f) {
                    const int m = s0 + g.m_off + f;
                    if (m < 0 || m > g.Fr) continue;
                    fir_bf16_products(g, slot(m), m, J0, a_off, 8 * li + 4 * lk, acc[0]);
                }
                fir_bf16_store(g, b, J0, li, lk, acc[0]);
            }
        }
        if constexpr (PAIR) {
            __syncthreads();
            if (mine && !odd) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t][0] += handover[(seg * 2 + t) * 64 + lane];
                    fir_bf16_store(g, b, J0 + 16 * t, li, lk, acc[t]);
                }
            }
        }
        __syncthreads();
    }
}

template <int SEGB>
int launch_fir_bf16(ddsp_ctx* ctx, hipStream_t st, FirBfArgs g, int64_t B, int64_t Fr, size_t lds_bytes) {
    DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)ltv_fir_bf16_kernel<SEGB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    dim3 grid((unsigned)((Fr + SEGB - 1) / SEGB), (unsigned)B);
    hipLaunchKernelGGL((ltv_fir_bf16_kernel<SEGB>), grid, dim3(64 * 2 * SEGB), lds_bytes, st, g);
    return DDSP_OK;
}

template <int SEGB, int SW, bool PAIR>
int launch_fir_bf16_march(ddsp_ctx* ctx, hipStream_t st, FirBfArgs g, int64_t B, int nchunks, size_t lds_bytes) {
    DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)ltv_fir_bf16_march_kernel<SEGB, SW, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    hipLaunchKernelGGL((ltv_fir_bf16_march_kernel<SEGB, SW, PAIR>), dim3((unsigned)nchunks, (unsigned)B),
                       dim3(64 * (2 + SW) * SEGB), lds_bytes, st, g);
    return DDSP_OK;
}

// ---- backward ---------------------------------------------------------------------------------------------
// y[u] = sum_m sum_k ir_m[k] * (x*W_m)[u + c - k]  (c = n/2).  Two adjoints are needed for training:
//   (1) d_x[t]     = sum_{m in {s, s+1}} W_m[t] * sum_k ir_m[k] * d_y[t - c + k]        (s = segment of t)
//   (2) d_ir[m][k] = sum_z (x*W_m)[tb + z] * d_y[tb + z - c + k],  tb = hop*(m-1), z in [0, 2*hop)
// Both are the same Toeplitz-block products as the forward kernel with other operands.

// (1) input gradient: filters reversed (k' = n-1-k, delay c' = n/2 - 1), d_y staged once UNwindowed, the frame
// window applied to each frame's accumulator on the way out.  No zero-padding waste: 2 frames x full q range.
struct FirBwdInArgs {
    const float* dy;  // (B,T)
    const float* ir;  // (B,Fr,n)
    float* dx;        // (B,T)
    int Fr, n, q_lo, q_hi, irs, rsb, col_off;  // col_off = q_hi (image column 0 = absolute column 32*s0 - q_hi)
};

__global__ void __launch_bounds__(64 * 2 * SEG) ltv_fir_bwd_input_kernel(FirBwdInArgs g) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, s0 = blockIdx.x * SEG;
    const int64_t T = (int64_t)g.Fr * HOP;
    const int cp = g.n / 2 - 1;
    float* irs = lds;                               // [SEG+1][g.irs]  reversed filters of frames s0 .. s0+SEG
    float* img = lds + (size_t)(SEG + 1) * g.irs;   // [16][g.rsb]     raw d_y, transposed + skewed like the forward image
    for (int f = 0; f <= SEG; ++f) {
        const int m = s0 + f;
        const bool live = m <= g.Fr;
        const int mi = m < g.Fr ? m : g.Fr - 1;
        const float* src = g.ir + ((int64_t)b * g.Fr + (live ? mi : 0)) * g.n;
        float* dst = irs + (size_t)f * g.irs;
        for (int i = tid; i < g.irs; i += blockDim.x) {
            const int k = i - IRPAD;
            dst[i] = (live && k >= 0 && k < g.n) ? src[g.n - 1 - k] : 0.f;
        }
    }
    const int ncol = g.rsb - 16;                    // columns the image must cover (multiple of 16 by construction)
    const int64_t t_base = 16 * ((int64_t)HOPC * s0 - g.col_off);
    for (int z = 4 * tid; z < 16 * ncol; z += 4 * blockDim.x) {
        const int64_t t = t_base + z;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) v = *(const f32x4*)(g.dy + (int64_t)b * T + t);
        const int r0 = z & 15, col = z >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) img[(r0 + e) * g.rsb + col + 2 * ((r0 + e) >> 1)] = v[e];
    }
    __syncthreads();
    const int seg = s0 + (wave >> 1);
    if (seg >= g.Fr) return;
    const int J0 = HOPC * seg + 16 * (wave & 1);
    const int li = lane & 15, lk = lane >> 4;
    const int a_base = IRPAD + cp + li - lk;
    const int b_base = lk * g.rsb + 2 * (lk >> 1) + li + (J0 - (HOPC * s0 - g.col_off));
    const int bstep = 4 * g.rsb + 4;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    const int j0 = (16 * (J0 - HOPC * seg) + 16 * li + 4 * lk);  // position of this lane's first sample in its segment
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int f = (seg - s0) + which;
        toeplitz_accumulate<-1, -1>(irs + (size_t)f * g.irs + a_base, img + b_base, bstep, g.q_lo, g.q_hi, acc);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float j = (float)(j0 + e) * (1.0f / HOP);
            const float w = which ? j : 1.0f - j;
            o[e] = fmaf(w, (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]), o[e]);
        }
    }
    *(f32x4*)(g.dx + (int64_t)b * T + 16 * (int64_t)(J0 + li) + 4 * lk) = o;
}

// (2) filter gradient: one workgroup per (utterance, frame); "taps" = the windowed input frame (2*hop values),
// "signal" = d_y around the frame, outputs = the n filter taps.  O[i][j] = d_ir[16(J+j)+i]:
//   A_q[i][r] = xw[16q + r - i],  B_q[r][j] = d_y[tb - c + 16(J+j+q) + r],  q = 0..2*hop/16.
struct FirBwdIrArgs {
    const float* audio;  // forward input (B,T) or null
    int excitation;
    uint64_t seed;
    const float* dy;     // (B,T)
    float* dir;          // (B,Fr,n)
    int Fr, n, rsf, tiles;
};

__global__ void __launch_bounds__(256) ltv_fir_bwd_filter_kernel(FirBwdIrArgs g) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, m0 = blockIdx.x;       // frame m0 < Fr; frame Fr (same filter) is folded into Fr-1
    const int64_t T = (int64_t)g.Fr * HOP;
    const int c = g.n / 2;
    constexpr int HROW = 2 * HOP + 2 * IRPAD;        // windowed frame with 32 zeros either side
    float* hrow = lds;
    float* img = lds + HROW;                         // [16][g.rsf]
    const int li = lane & 15, lk = lane >> 4;
    const int bstep = 4 * g.rsf + 4;
    const int n_pass = (m0 == g.Fr - 1) ? 2 : 1;
    f32x4 out[4];                                    // this wave's tiles: wave, wave+4, ... (up to 4 for n <= 4096)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < n_pass; ++pass) {
        const int m = m0 + pass;
        const int64_t tb = (int64_t)HOP * (m - 1);
        __syncthreads();
        for (int i = tid; i < HROW; i += 256) {
            const int z = i - IRPAD;
            float v = 0.f;
            const int64_t t = tb + z;
            if (z >= 0 && z < 2 * HOP && t >= 0 && t < T) {
                float x;
                if (g.excitation == DDSP_EXC_GENERATE) {
                    x = noise_u(g.seed, (uint64_t)b * T + t);
                } else {
                    x = g.audio[(int64_t)b * T + t];
                    if (g.excitation == DDSP_EXC_UNIT_NOISE) x = __fadd_rn(__fmul_rn(x, 2.0f), -1.0f);
                }
                const float w = (z < HOP) ? (float)z * (1.0f / HOP) : (float)(2 * HOP - z) * (1.0f / HOP);
                v = x * w;
            }
            hrow[i] = v;
        }
        const int ncol = g.rsf - 16;
        for (int z = 4 * tid; z < 16 * ncol; z += 4 * 256) {
            const int64_t t = tb - c + z;                 // tb and c: t is a multiple of 4 only when c is;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};               // c = n/2 may be odd -> element-wise bounds and loads
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t + e >= 0 && t + e < T) v[e] = g.dy[(int64_t)b * T + t + e];
            const int r0 = z & 15, col = z >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) img[(r0 + e) * g.rsf + col + 2 * ((r0 + e) >> 1)] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int tile = wave + 4 * it;
            if (tile < g.tiles) {
                const int J = 16 * tile;
                f32x4 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* ap = hrow + IRPAD + lk - li;
                const float* bp = img + lk * g.rsf + 2 * (lk >> 1) + li + J;
                toeplitz_accumulate<1, 1>(ap, bp, bstep, 0, 2 * HOP / 16, acc);
#pragma unroll
                for (int e = 0; e < 4; ++e) out[it][e] += (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int tile = wave + 4 * it;
        if (tile < g.tiles) {
            const int k0 = 16 * (16 * tile + li) + 4 * lk;
            float* dst = g.dir + ((int64_t)b * g.Fr + m0) * g.n;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < g.n) dst[k0 + e] = out[it][e];
        }
    }
}

}  // namespace

extern "C" int ddsp_ltv_fir(ddsp_ctx* ctx, void* stream, const float* audio, int excitation, uint64_t noise_seed,
                            const float* ir, int64_t B, int64_t Fr, int hop, int n, const float* add_in, float* out,
                            float* out_sum, int math) {
    DDSP_REQUIRE(ctx, ctx && ir && (out || out_sum), "ddsp_ltv_fir: null argument");
    DDSP_REQUIRE(ctx, math == 0 || math == 3 || (math >= 31 && math <= 36) || (math >= 41 && math <= 48) || (math >= 51 && math <= 58), "ddsp_ltv_fir: math must be 0 (fp32) or 3 (split-bf16)");
    DDSP_REQUIRE(ctx, excitation >= 0 && excitation <= 2, "ddsp_ltv_fir: unknown excitation");
    DDSP_REQUIRE(ctx, (excitation == DDSP_EXC_GENERATE) || audio, "ddsp_ltv_fir: audio is null");
    DDSP_REQUIRE(ctx, (out_sum == nullptr) == (add_in == nullptr), "ddsp_ltv_fir: out_sum and add_in go together");
    DDSP_REQUIRE(ctx, hop == HOP, "ddsp_ltv_fir: only hop == 512 is built");
    DDSP_REQUIRE(ctx, n >= 32 && n <= 2046 && (n % 2) == 0, "ddsp_ltv_fir: n must be even, 32..2046");
    DDSP_REQUIRE(ctx, B >= 0 && Fr >= 1 && B <= 65535, "ddsp_ltv_fir: bad shape");
    DDSP_REQUIRE(ctx, ((uintptr_t)out % 16) == 0 && ((uintptr_t)add_in % 16) == 0 && ((uintptr_t)out_sum % 16) == 0 &&
                          ((uintptr_t)audio % 16) == 0 && ((uintptr_t)ir % 8) == 0,
                 "ddsp_ltv_fir: audio/out/add_in/out_sum must be 16-byte aligned (ir 8-byte)");
    if (B == 0) return DDSP_OK;
    const double Tsig = (double)Fr * HOP, nbat = (double)B;
    // algorithmic work: every input sample meets two frame filters of n taps (2 FLOP per tap);
    // algorithmic bytes: x in (0 when generated), filters in, y out (+ add_in, + second output)
    const double alg_flops = nbat * Tsig * n * 2.0 * 2.0;
    const double alg_bytes = 4.0 * nbat * (Tsig * ((audio ? 1 : 0) + (out ? 1 : 0) + (out_sum ? 2 : 0)) + (double)Fr * n);
    if (math != 0) {
        // split-bf16 kernel: largest block that fits the LDS; filters too long for it take the fp32 kernel below
        const int c = n / 2;
        FirBfArgs g;
        g.audio = audio;
        g.excitation = excitation;
        g.seed = noise_seed;
        g.ir = ir;
        g.add_in = add_in;
        g.out = out;
        g.out_sum = out_sum;
        g.Fr = (int)Fr;
        g.n = n;
        g.Q_lo = -((c + 15) / 32);
        g.Q_hi = (n + 30 - c) / 32;
        g.m_off = -((c - 1 + HOP - 1) / HOP);             // floor((1 - c) / hop)
        const int extra = (c - 1) / HOP + 1 - g.m_off + 1;  // frames staged beyond the block's own segments
        g.fd = ((n + FPL + FPR) / 2 + 4 + 15) / 32 * 32 + 16;
        const size_t per_frame = (size_t)(4 * g.fd + 2 * XPLANE) * sizeof(uint32_t);
        g.chunk = 0;
        g.ring = 0;
        DDSP_ENTER_DEVICE(ctx);
        // (a) marching blocks when every CU can own a long run of segments: block shape by LDS, <segments per step,
        //     staging waves per frame>.  math 41..44 force a shape (tools/fir_bf16_check.py).
        {
            static const int mseg[9] = {0, 3, 3, 4, 4, 3, 3, 4, 4}, msw[9] = {0, 1, 2, 1, 2, 1, 2, 1, 2};
            int mc = (math >= 41 && math <= 48) ? math - 40 : 0;   // 41..44: a tile per product wave, 45..48: a segment per pair
            if (math >= 51 && math <= 58) {   // the same shapes, staging only (no products; output zero): a measurement aid
                mc = math - 50;
                g.Q_lo = 1;
                g.Q_hi = 0;
            }
            if (math == 3) {
                // measured order (tools/fir_bf16_check.py): paired product waves, 4 segments per step if the ring fits
                static const int pref[4] = {7, 6, 4, 2};
                for (int i = 0; i < 4 && mc == 0; ++i) {
                    const int cand = pref[i];
                    if ((size_t)(2 * mseg[cand] + extra) * per_frame + (cand > 4 ? 2048 * (size_t)mseg[cand] : 0) <= 160 * 1024) mc = cand;
                }
            }
            if (mc != 0) {
                const int segb = mseg[mc];
                g.nfr = segb + extra;
                g.ring = g.nfr + segb;
                const size_t lds_bytes = (size_t)g.ring * per_frame + (mc > 4 ? 2048 * (size_t)segb : 0);   // + hand-over area
                // blocks: about one per CU, each at least four steps long
                int nchunks = B >= 256 ? 1 : (int)(256 / B);
                const int max_chunks = (int)(Fr / (4 * segb));
                if (nchunks > max_chunks) nchunks = max_chunks;
                const bool worth = nchunks >= 1 && B * nchunks >= 128;
                if (lds_bytes <= 160 * 1024 && (worth || math != 3) && nchunks >= 1) {
                    g.chunk = (int)((Fr + nchunks - 1) / nchunks);
                    g.chunk = (g.chunk + segb - 1) / segb * segb;
                    nchunks = (int)((Fr + g.chunk - 1) / g.chunk);
                    ddsp_prof_begin(ctx, (hipStream_t)stream, PF_LTV_FIR);
                    int rc;
                    hipStream_t st = (hipStream_t)stream;
                    switch (mc) {
                        case 1: rc = launch_fir_bf16_march<3, 1, false>(ctx, st, g, B, nchunks, lds_bytes); break;
                        case 2: rc = launch_fir_bf16_march<3, 2, false>(ctx, st, g, B, nchunks, lds_bytes); break;
                        case 3: rc = launch_fir_bf16_march<4, 1, false>(ctx, st, g, B, nchunks, lds_bytes); break;
                        case 4: rc = launch_fir_bf16_march<4, 2, false>(ctx, st, g, B, nchunks, lds_bytes); break;
                        case 5: rc = launch_fir_bf16_march<3, 1, true>(ctx, st, g, B, nchunks, lds_bytes); break;
                        case 6: rc = launch_fir_bf16_march<3, 2, true>(ctx, st, g, B, nchunks, lds_bytes); break;
                        case 7: rc = launch_fir_bf16_march<4, 1, true>(ctx, st, g, B, nchunks, lds_bytes); break;
                        default: rc = launch_fir_bf16_march<4, 2, true>(ctx, st, g, B, nchunks, lds_bytes); break;
                    }
                    if (rc != DDSP_OK) return rc;
                    ddsp_prof_end(ctx, (hipStream_t)stream, alg_flops, alg_bytes);
                    DDSP_LAUNCH_CHECK(ctx);
                    return DDSP_OK;
                }
            }
        }
        // (b) one block per group of segments.  math 31..34 force a shape, 35 / 36 run staging only (measurement aid).
        int cfg = (math >= 31 && math <= 36) ? math - 30 : 0;
        static const int seg_of[5] = {0, 4, 6, 7, 5};
        if (cfg >= 5) {
            cfg -= 4;
            g.Q_lo = 1;
            g.Q_hi = 0;
        }
        // two resident 4-segment blocks per CU (one stages while the other multiplies) beat one 6-segment block
        if (cfg == 0)
            cfg = ((size_t)(4 + extra) * per_frame <= 80 * 1024)    ? 1
                  : ((size_t)(6 + extra) * per_frame <= 160 * 1024) ? 2
                                                                    : 1;
        const int segb = seg_of[cfg];
        g.nfr = segb + extra;
        const size_t lds_bytes = (size_t)g.nfr * per_frame;
        if (lds_bytes <= 160 * 1024) {
            ddsp_prof_begin(ctx, (hipStream_t)stream, PF_LTV_FIR);
            int rc;
            switch (cfg) {
                case 1: rc = launch_fir_bf16<4>(ctx, (hipStream_t)stream, g, B, Fr, lds_bytes); break;
                case 2: rc = launch_fir_bf16<6>(ctx, (hipStream_t)stream, g, B, Fr, lds_bytes); break;
                case 3: rc = launch_fir_bf16<7>(ctx, (hipStream_t)stream, g, B, Fr, lds_bytes); break;
                default: rc = launch_fir_bf16<5>(ctx, (hipStream_t)stream, g, B, Fr, lds_bytes); break;
            }
            if (rc != DDSP_OK) return rc;
            ddsp_prof_end(ctx, (hipStream_t)stream, alg_flops, alg_bytes);
            DDSP_LAUNCH_CHECK(ctx);
            return DDSP_OK;
        }
    }
    FirArgs g;
    g.audio = audio;
    g.excitation = excitation;
    g.seed = noise_seed;
    g.ir = ir;
    g.add_in = add_in;
    g.out = out;
    g.out_sum = out_sum;
    g.Fr = (int)Fr;
    g.n = n;
    const int c = n / 2;
    g.q_lo = -((c + 15) / 16);
    g.q_hi = (c + 14) / 16;
    // input columns touched by one block: [32*s0 - q_hi, 32*(s0+SEG) - 1 - q_lo]
    g.m_off = -((g.q_hi + HOPC - 1) / HOPC);
    g.nfr = SEG + (-1 - g.q_lo) / HOPC + 1 - g.m_off + 1;
    g.irs = (n + 2 * IRPAD + 3) & ~3;
    const size_t lds_bytes = (size_t)g.nfr * (g.irs + XS) * sizeof(float);
    DDSP_REQUIRE(ctx, lds_bytes <= 160 * 1024, "ddsp_ltv_fir: filter too long for the LDS staging");
    DDSP_ENTER_DEVICE(ctx);
    DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)ltv_fir_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    dim3 grid((unsigned)((Fr + SEG - 1) / SEG), (unsigned)B);
    ddsp_prof_begin(ctx, (hipStream_t)stream, PF_LTV_FIR);
    hipLaunchKernelGGL(ltv_fir_kernel, grid, dim3(64 * 2 * SEG), lds_bytes, (hipStream_t)stream, g);
    ddsp_prof_end(ctx, (hipStream_t)stream, alg_flops, alg_bytes);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_ltv_fir_bwd(ddsp_ctx* ctx, void* stream, const float* audio, int excitation, uint64_t noise_seed,
                                const float* ir, const float* d_out, int64_t B, int64_t Fr, int hop, int n,
                                float* d_audio, float* d_ir) {
    DDSP_REQUIRE(ctx, ctx && ir && d_out && (d_audio || d_ir), "ddsp_ltv_fir_bwd: null argument");
    DDSP_REQUIRE(ctx, excitation >= 0 && excitation <= 2, "ddsp_ltv_fir_bwd: unknown excitation");
    DDSP_REQUIRE(ctx, !d_ir || excitation == DDSP_EXC_GENERATE || audio, "ddsp_ltv_fir_bwd: d_ir needs the forward input");
    DDSP_REQUIRE(ctx, hop == HOP, "ddsp_ltv_fir_bwd: only hop == 512 is built");
    DDSP_REQUIRE(ctx, n >= 32 && n <= 2046 && (n % 2) == 0, "ddsp_ltv_fir_bwd: n must be even, 32..2046");
    DDSP_REQUIRE(ctx, B >= 0 && Fr >= 1 && B <= 65535, "ddsp_ltv_fir_bwd: bad shape");
    DDSP_REQUIRE(ctx, ((uintptr_t)d_out % 16) == 0 && ((uintptr_t)d_audio % 16) == 0, "ddsp_ltv_fir_bwd: d_out/d_audio must be 16-byte aligned");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    DDSP_ONCE_PER_DEVICE(ctx, DDSP_HIP(ctx, hipFuncSetAttribute((const void*)ltv_fir_bwd_input_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); DDSP_HIP(ctx, hipFuncSetAttribute((const void*)ltv_fir_bwd_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    const double Ts = (double)Fr * HOP;
    if (d_audio) {
        FirBwdInArgs g;
        g.dy = d_out;
        g.ir = ir;
        g.dx = d_audio;
        g.Fr = (int)Fr;
        g.n = n;
        const int cp = n / 2 - 1;
        g.q_lo = -((cp + 15) / 16);
        g.q_hi = (cp + 14) / 16;
        g.irs = (n + 2 * IRPAD + 3) & ~3;
        g.col_off = g.q_hi;
        // image columns: [32*s0 - q_hi, 32*(s0+SEG) - 1 - q_lo] (+ up to 14 of skew), row stride = 16 (mod 32)
        const int width = HOPC * SEG + g.q_hi - g.q_lo;
        const int ncol = (width + 31) & ~31;   // multiple of 32 -> row stride = 16 (mod 32): conflict-free operand fetch
        g.rsb = ncol + 16;
        const size_t lds_bytes = ((size_t)(SEG + 1) * g.irs + 16 * (size_t)g.rsb) * sizeof(float);
        DDSP_REQUIRE(ctx, lds_bytes <= 160 * 1024, "ddsp_ltv_fir_bwd: filter too long for the LDS staging");
        ddsp_prof_begin(ctx, st, PF_LTV_FIR_BWD);
        hipLaunchKernelGGL(ltv_fir_bwd_input_kernel, dim3((unsigned)((Fr + SEG - 1) / SEG), (unsigned)B),
                           dim3(64 * 2 * SEG), lds_bytes, st, g);
        ddsp_prof_end(ctx, st, (double)B * Ts * n * 4.0, 4.0 * B * (2.0 * Ts + (double)Fr * n));
    }
    if (d_ir) {
        FirBwdIrArgs g;
        g.audio = audio;
        g.excitation = excitation;
        g.seed = noise_seed;
        g.dy = d_out;
        g.dir = d_ir;
        g.Fr = (int)Fr;
        g.n = n;
        g.tiles = (n + 255) / 256;
        // signal positions used: [0, 2*hop - 1 + n) -> columns, + skew, row stride = 16 (mod 32)
        const int ncol = ((2 * HOP + n + 15) / 16 + 16 + 31) & ~31;
        g.rsf = ncol + 16;
        const size_t lds_bytes = ((size_t)(2 * HOP + 2 * IRPAD) + 16 * (size_t)g.rsf) * sizeof(float);
        ddsp_prof_begin(ctx, st, PF_LTV_FIR_BWD);
        hipLaunchKernelGGL(ltv_fir_bwd_filter_kernel, dim3((unsigned)Fr, (unsigned)B), dim3(256), lds_bytes, st, g);
        ddsp_prof_end(ctx, st, (double)B * Ts * n * 4.0, 4.0 * B * (2.0 * Ts + (double)Fr * n));
    }
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
