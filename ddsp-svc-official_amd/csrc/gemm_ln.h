// Residual Linear layer + the LayerNorm that follows it, in one kernel (round 3).
//
//   x_new = res + (A W^T + bias)            (M x 256; A: M x K, W: 256 x K, both in the pre-split bf16 hi/lo operand layout)
//   y     = LayerNorm(x_new) * gamma + beta  (written as bf16 hi/lo groups: the A operand of the next split-bf16 GEMM)
//
// Replaces, for the out-projection and pw2 layers of the control network (ddsp/pcmer.py:221-251, 42-63) at large batches, the
// pair [kernel_dma<64,64,EpiResidual> -> layernorm_kernel]: the 64x64 tiling streamed the 512 KB weight image once per 64x64
// tile (176 MB L2 -> LDS per call at 11008 rows, which is what bounded it: ~31 B/clk per CU), and the LayerNorm read the
// result again.  Here a workgroup of 16 waves owns 64 rows x ALL 256 columns: the weights stream once per 64 rows (110 MB per
// call), the row statistics are available in the workgroup, and the LayerNorm costs two barriers.
//
// Same staging protocol as gemm::kernel_dma (4-stage LDS ring of 32-k row images filled by global_load_lds, XOR slot swizzle,
// one barrier per k-step); the product is formed TRANSPOSED - D[column][row] = W X^T - so that a lane ends up with 4 x 4
// consecutive columns of ONE row: the row sums stay in the lane, the residual is read and both results are written as 16-byte
// pieces, and the lane pair (l, l ^ 32) owns one 8-column split group.  The three bf16 products per fp32 product are the same
// and in the same order as in kernel_dma (act_lo w_hi, act_hi w_lo, act_hi w_hi over the same k sequence), and the LayerNorm
// statistics are summed in the order of layernorm_kernel's 64-lane butterfly (4-column partials, pairs 32 / 16 / ... / 1
// apart): the results are bit-identical to the two-kernel path (tests/test_gpu_gemm.py::test_residual_layernorm_kernel).
#pragma once
#include "gemm_f32.h"

#include <atomic>
#include <stdlib.h>

namespace gemm {

struct LnArgs {
    const float* A;       // (M, K) rows: pre-split, or fp32 (launch_res_ln(..., a_split = false))
    const float* W;       // (256, K) pre-split rows
    int64_t lda, ldw;     // row pitches in floats
    int M, K;             // K % 32 == 0
    const float* bias;    // [256]
    const float* res;     // (M, 256)
    float* X;             // (M, 256): res + A W^T + bias
    const float* gamma;   // [256]
    const float* beta;    // [256]
    float* Y;             // (M, 256): LayerNorm(X) as split groups (y_split) or fp32
    int y_split;
    // A_CONV3 (implicit im2col of a 3-tap "same" convolution, k = tap * Cin + c, Cin % 32 == 0): frames per utterance, input
    // channels, tap shift (0 centred, -1 causal) and a page of >= Cin + 32 zero floats for the taps that fall off an utterance
    int Fr = 1, Cin = 32, tap_shift = 0;
    const float* zeros = nullptr;
};

// What is added to the product before the LayerNorm: the default - bias and the residual rows, fetched before the k loop
struct PreResidual {
    struct State {
        f32x4 res[2][4];
        f32x4 bias4[4];
    };
    template <int RB>
    __device__ __forceinline__ void load(State& s, const LnArgs& g, int m0, int rg0, int lr, int cg, int lh) const {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row_m = m0 + 32 * (rg0 + b) + lr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s.res[b][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row_m < g.M) s.res[b][q] = *(const f32x4*)(g.res + (int64_t)row_m * 256 + 32 * cg + 8 * q + 4 * lh);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) s.bias4[q] = *(const f32x4*)(g.bias + 32 * cg + 8 * q + 4 * lh);
    }
    // x piece of row m (clamped into range by the caller), columns c0 .. c0 + 3 = group q of the lane; acc = the product
    __device__ __forceinline__ f32x4 apply(const State& s, const LnArgs&, int b, int q, int, int, f32x4 acc) const {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s.res[b][q][e] + (acc[e] + s.bias4[q][e]);
        return v;
    }
};

constexpr int LN_BM = 64, LN_N = 256, LN_NS = 4;   // 4 x 40 KB = the whole LDS: three k-steps in flight
constexpr int LN_STAGE = (LN_BM + LN_N) * 32;                 // floats per stage: 40 KB
constexpr size_t LN_LDS_BYTES = (size_t)LN_NS * LN_STAGE * sizeof(float);

// NS ring stages; RB row blocks of 32 per wave: 1 = 16 waves (2 x 8 grid), 2 = 8 waves (each both row blocks); ASPLIT: A arrives in
// the pre-split layout (else fp32 rows, split into bf16 hi / lo per fragment in the loop like kernel_dma's mode 7: the same bits)
// A_MODE: A_PLAIN or A_CONV3 (the loader moves an A row's source by the tap's frame offset, or to the zero page); Pre: what joins the
// product before the LayerNorm (PreResidual; unit2ctrl.hip's PreEmbed for the second prenet convolution)
template <int NS, int RB, bool ASPLIT, int A_MODE, class Pre>
__global__ void __launch_bounds__(1024 / RB) kernel_res_ln(LnArgs g, Pre pre) {
    extern __shared__ __attribute__((aligned(1024))) float ln_lds[];
    float* const lds = ln_lds;
    constexpr int NWAVE = 16 / RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg0 = RB == 1 ? wave >> 3 : 0, cg = wave & 7, lr = lane & 31, lh = lane >> 5;
    const int nk = g.K / 32;
    const int m0 = blockIdx.x * LN_BM;

    // the epilogue's row operands first: they are the oldest vector-memory operations of the wave, so the ring's vmcnt waits cover them
    typename Pre::State pst;
    pre.template load<RB>(pst, g, m0, rg0, lr, cg, lh);
    f32x4 gam4[4], bet4[4];   // (loaded now: one workgroup per CU, nothing would hide their latency in the epilogue)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * cg + 8 * q + 4 * lh;
        gam4[q] = *(const f32x4*)(g.gamma + c0);
        bet4[q] = *(const f32x4*)(g.beta + c0);
    }

    // 40 one-KiB pieces per stage (8 row images each): 16 waves: wave w fetches pieces w, w + 16 and, the first eight, w + 32;
    // 8 waves: w, w + 8, ..., w + 32
    constexpr int MAXP = RB == 1 ? 3 : 5;
    const float* src[MAXP];
    const float *src_m1 = nullptr, *src_p1 = nullptr;   // A_CONV3: taps 0 and 2 of this lane's A row (its piece is i = 0: wave < 8)
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int piece = wave + NWAVE * i;
        const int row = piece * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        if (row < LN_BM) {
            int m = m0 + row;
            m = m < g.M ? m : g.M - 1;
            src[i] = g.A + (int64_t)m * g.lda + slot * 4;
            if constexpr (A_MODE == A_CONV3) {
                const int f = m % g.Fr, sh = g.tap_shift;
                const float* base = src[i];
                const float* zp = g.zeros + slot * 4;
                src_m1 = (f - 1 + sh >= 0 && f - 1 + sh < g.Fr) ? base + (int64_t)(sh - 1) * g.lda : zp;
                src_p1 = (f + 1 + sh >= 0 && f + 1 + sh < g.Fr) ? base + (int64_t)(sh + 1) * g.lda : zp;
                src[i] = (f + sh >= 0 && f + sh < g.Fr) ? base + (int64_t)sh * g.lda : zp;
            }
        } else {
            const int n = row - LN_BM < LN_N ? row - LN_BM : LN_N - 1;
            src[i] = g.W + (int64_t)n * g.ldw + slot * 4;
        }
    }
    static_assert(A_MODE == A_PLAIN || (A_MODE == A_CONV3 && RB == 2), "conv mode: eight waves, one A piece each");
    auto issue = [&](int kt) {
        float* const st = lds + (kt % NS) * LN_STAGE;
        int tap = 1, koff = kt * 32;
        if constexpr (A_MODE == A_CONV3) {
            tap = koff / g.Cin;
            koff -= tap * g.Cin;
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i)
            if (RB == 2 || i < 2 || wave < 8) {
                const float* p = src[i] + kt * 32;
                if constexpr (A_MODE == A_CONV3) {
                    if (i == 0) p = (tap == 0 ? src_m1 : tap == 1 ? src[0] : src_p1) + koff;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(st + (wave + NWAVE * i) * 256), 16, 0, 0);
            }
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < nk) issue(i);

    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x16 acc[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
        // step kt has landed once at most the pieces of the NS - 2 following steps are still in flight
        if (kt + NS - 2 < nk) {
            if (RB == 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (NS - 2)) : "memory");
            else if (wave < 8)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NS - 2)) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 2)) : "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
        const float* st = lds + (kt % NS) * LN_STAGE;
        f32x4 xv[RB][4], wv[4];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row = 32 * (rg0 + b) + lr;
            const float* rp = st + row * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) xv[b][c] = *(const f32x4*)(rp + 4 * ((4 * lh + c) ^ ((row >> 1) & 7)));
        }
        {
            const int row = LN_BM + 32 * cg + lr;
            const float* rp = st + row * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) wv[c] = *(const f32x4*)(rp + 4 * ((4 * lh + c) ^ ((row >> 1) & 7)));
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wv[2 * half]), wl = __builtin_bit_cast(bf16x8, wv[2 * half + 1]);
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                bf16x8 xh, xl;
                if constexpr (ASPLIT) {
                    xh = __builtin_bit_cast(bf16x8, xv[b][2 * half]);
                    xl = __builtin_bit_cast(bf16x8, xv[b][2 * half + 1]);
                } else {
                    // kernel_dma's split of a lane's 8 k-values: round to nearest even twice, two values per conversion
                    typedef float f32x2p __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
                    ddsp_u32x4 wh_, wl_;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const f32x4& src = d < 2 ? xv[b][2 * half] : xv[b][2 * half + 1];
                        f32x2p r = {src[(2 * d) & 3], src[(2 * d + 1) & 3]};
                        const uint32_t hh = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2p));
                        wh_[d] = hh;
                        r = r - (f32x2p){__builtin_bit_cast(float, hh << 16), __builtin_bit_cast(float, hh & 0xffff0000u)};
                        wl_[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2p));
                    }
                    xh = __builtin_bit_cast(bf16x8, wh_);
                    xl = __builtin_bit_cast(bf16x8, wl_);
                }
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[b], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: acc[b][4 q + e] = column 32 cg + 8 q + 4 lh + e of row 32 (rg0 + b) + lr ----
    constexpr int PP = 65;                                      // pitch of a row of 64 four-column partials
    float* const P1 = lds;
    float* const P2 = lds + LN_BM * PP;
    float* const TOT = lds + 2 * LN_BM * PP;                    // [64] row totals
    __syncthreads();                                            // the ring is dead
    f32x4 v[RB][4];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int row = 32 * (rg0 + b) + lr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mr = m0 + row < g.M ? m0 + row : g.M - 1;
            v[b][q] = pre.apply(pst, g, b, q, mr, 32 * cg + 8 * q + 4 * lh,
                                f32x4{acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]});
            P1[row * PP + 8 * cg + 2 * q + lh] = (v[b][q][0] + v[b][q][1]) + (v[b][q][2] + v[b][q][3]);
            if (m0 + row < g.M) *(f32x4*)(g.X + (int64_t)(m0 + row) * LN_N + 32 * cg + 8 * q + 4 * lh) = v[b][q];
        }
    }
    __syncthreads();
    // the 64-lane butterfly sum of layernorm_kernel (partners 32, 16, 8, 4, 2, 1 apart): each wave sums four rows per pass, 16 lanes
    // a row (lane j: partials j, j + 16, j + 32, j + 48, then the 8 / 4 / 2 / 1 exchanges), and leaves the totals in the LDS
    auto tree = [&](const float* Pm, float (&out)[RB]) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int trow = 32 * (rg0 + b) + 4 * cg + (lane >> 4), j = lane & 15;
            const float* p = Pm + trow * PP;
            float t = (p[j] + p[j + 32]) + (p[j + 16] + p[j + 48]);
            t += __shfl_xor(t, 8, 64);
            t += __shfl_xor(t, 4, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 1, 64);
            if (j == 0) TOT[trow] = t;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < RB; ++b) out[b] = TOT[32 * (rg0 + b) + lr];
        __syncthreads();                                      // (TOT is written again by the second sum)
    };
    float mean[RB], rstd[RB];
    tree(P1, mean);
    f32x4 d[RB][4];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        mean[b] *= (1.0f / LN_N);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[b][q][e] = v[b][q][e] - mean[b];
                ss = fmaf(d[b][q][e], d[b][q][e], ss);
            }
            P2[(32 * (rg0 + b) + lr) * PP + 8 * cg + 2 * q + lh] = ss;
        }
    }
    __syncthreads();
    tree(P2, rstd);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int row_m = m0 + 32 * (rg0 + b) + lr;
        const bool row_ok = row_m < g.M;
        const float rs = 1.0f / sqrtf(rstd[b] * (1.0f / LN_N) + 1e-5f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 32 * cg + 8 * q + 4 * lh;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(d[b][q][e] * rs, gam4[q][e], bet4[q][e]);
            if (g.y_split) {
                const ddsp_u32x4 pk = ddsp_split4_pair(o, lh != 0, 32);      // (every lane takes part in the exchange)
                if (row_ok) *(ddsp_u32x4*)(g.Y + (int64_t)row_m * LN_N + c0) = pk;
            } else if (row_ok)
                *(f32x4*)(g.Y + (int64_t)row_m * LN_N + c0) = o;
        }
    }
}

inline bool res_ln_ok(const LnArgs& g) {
    return g.K % 32 == 0 && g.K >= 64 && g.lda % 4 == 0 && g.ldw % 4 == 0 && g.M >= 1 &&
           (((uintptr_t)g.A | (uintptr_t)g.W | (uintptr_t)g.bias | (uintptr_t)g.res | (uintptr_t)g.X | (uintptr_t)g.gamma |
             (uintptr_t)g.beta | (uintptr_t)g.Y) % 16) == 0;
}

template <int RB, bool ASPLIT, int A_MODE = A_PLAIN, class Pre = PreResidual>
inline hipError_t launch_res_ln_rb(hipStream_t st, const LnArgs& g, const Pre& pre = Pre{}) {
    static std::atomic<uint64_t> done{0};   // devices on which the dynamic-LDS attribute has been set
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    auto kfn = kernel_res_ln<LN_NS, RB, ASPLIT, A_MODE, Pre>;
    if (!(done.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LN_LDS_BYTES);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)((g.M + LN_BM - 1) / LN_BM)), dim3(1024 / RB), LN_LDS_BYTES, st, g, pre);
    return hipSuccess;
}
// a_split: A is in the pre-split layout (else fp32 rows)
inline hipError_t launch_res_ln(hipStream_t st, const LnArgs& g, bool a_split = true) {
    static int rb = -1;     // DDSP_GEMM_LN_RB=1: sixteen waves of one 32 x 32 block each (measurement aid)
    if (rb < 0) {
        const char* e = getenv("DDSP_GEMM_LN_RB");
        rb = (e && e[0] == '1') ? 1 : 2;
    }
    if (!a_split) return launch_res_ln_rb<2, false>(st, g);
    return rb == 1 ? launch_res_ln_rb<1, true>(st, g) : launch_res_ln_rb<2, true>(st, g);
}

}  // namespace gemm
