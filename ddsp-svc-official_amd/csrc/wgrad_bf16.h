// Weight gradients of the Linear / Conv1d(k=3) layers on the bf16 matrix pipe (split-bf16, 3 products, fp32 accumulation).
//
//   out[o][t*C + c] = sum_m dY[m][o] * X[m + t - (taps-1)/2 + tap_shift][c]   (rows outside the utterance of m contribute 0)
//   bias[o]         = sum_m dY[m][o]                                 (optional)
//
// Both operands are stored with the CONTRACTION index m as the slow one (frames x channels), the opposite of what the
// LDS-DMA GEMM (gemm_f32.h) streams.  The transposition happens while the operands are staged: a thread owns one
// (channel, group of 8 frames) record, fetches its 8 values with 8 dword loads - lanes are consecutive channels, so every
// load instruction reads whole 256-byte row pieces -, splits them into 8 bf16 hi and 8 bf16 lo parts and writes the two
// 16-byte halves into the LDS image [k-group][hi|lo][channel]: the MFMA operand of lane (i = l & 31, h = l >> 5) in
// k-step s is then ONE ds_read_b128 each for hi and lo at [2s + h][.][i], conflict-free (32 consecutive 16-byte slots).
// The split costs ~2 vector instructions per value and is paid once per staged value, i.e. 1/32 .. 1/64 of a value's uses.
//
// The frame axis is cut into `splits` chunks (blockIdx.z) that write fp32 partial sums; a second kernel adds them in a
// fixed order (deterministic).  The bias column sums ride along in the staging threads of the first column tile (the
// values are in registers anyway) and replace a separate pass over dY.
//
// Block = 256 threads = 2 x 2 wavefronts, block tile 64 TM x 64 TN, wave tile 32 TM x 32 TN, BK = 32 frames.
#pragma once
#include "common.h"

namespace wgrad {

struct Args {
    const float* dY;
    int64_t ldy;
    int O;
    const float* X;
    int64_t ldx;
    int C;       // channels per tap
    int taps;    // 1 (Linear) or 3 (Conv1d k=3): column block t reads X shifted by t - 1 + tap_shift frames
    int tap_shift;   // 0: centred taps, -1: causal taps (frames -2, -1, 0)
    int Fr;      // frames per utterance (the shift does not cross utterances)
    int64_t M;   // rows = utterances x frames
    int chunk;   // rows per split, a multiple of 32
    float* partial;        // [splits][O][taps * C]
    float* bias_partial;   // [splits][O] or null
    // batched mode (zdiv > 0): blockIdx.z is an independent problem (hi, lo) = (z / zdiv, z % zdiv) over all M rows instead of a
    // split of the rows; operand bases move by the strides below and `partial` holds one [O][C] product per problem
    int zdiv = 0;
    int64_t sY_hi = 0, sY_lo = 0, sX_hi = 0, sX_lo = 0;
    // weighted column sums (batched mode): bias_partial[z * ldb + o] = sum_m dY[m][o] * bias_w[m * ldw] instead of the plain sums
    const float* bias_w = nullptr;
    int64_t ldw = 0, sW_hi = 0, sW_lo = 0;
    int64_t ldb = 0;   // row stride of bias_partial (0: O)
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TM, int TN>
__global__ void __launch_bounds__(256) kernel(Args g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int STAGE = 8 * (BM + BN);               // 16-byte slots per stage
    __shared__ ddsp_u32x4 lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * 32 * TM, wn = (wave & 1) * 32 * TN;
    const int o0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int N = g.taps * g.C;
    const int tap = n0 / g.C, c0 = n0 - tap * g.C;
    const int shift = tap - (g.taps - 1) / 2 + g.tap_shift;
    const int z = blockIdx.z;
    if (g.zdiv > 0) {
        const int zh = z / g.zdiv, zl = z - zh * g.zdiv;
        g.dY += zh * g.sY_hi + zl * g.sY_lo;
        g.X += zh * g.sX_hi + zl * g.sX_lo;
        if (g.bias_w) g.bias_w += zh * g.sW_hi + zl * g.sW_lo;
    }
    const int64_t kbeg = g.zdiv > 0 ? 0 : (int64_t)z * g.chunk;
    const int64_t kend = g.zdiv > 0 ? g.M : (kbeg + g.chunk < g.M ? kbeg + g.chunk : g.M);
    const int nk = (int)((kend - kbeg + 31) / 32);
    const bool want_bias = g.bias_partial != nullptr && blockIdx.x == 0;
    const bool a_cols_full = o0 + BM <= g.O, b_cols_full = c0 + BN <= g.C;

    float ra[TM][8], rb[TN][8], bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] = 0.f;

    auto load_tiles = [&](int kt) {
        const int64_t r0 = kbeg + (int64_t)kt * 32;
        const bool rows_full = r0 + 32 <= kend;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int q = tid + 256 * i, o = q % BM, kg = q / BM;
            const float* p = g.dY + (r0 + kg * 8) * g.ldy + o0 + o;
            if (rows_full && a_cols_full) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ra[i][j] = p[j * g.ldy];
            } else {
                const bool c_ok = o0 + o < g.O;
#pragma unroll
                for (int j = 0; j < 8; ++j) ra[i][j] = (c_ok && r0 + kg * 8 + j < kend) ? p[j * g.ldy] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int q = tid + 256 * i, c = q % BN, kg = q / BN;
            const float* p = g.X + (r0 + kg * 8 + shift) * g.ldx + c0 + c;
            if (rows_full && b_cols_full && shift == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) rb[i][j] = p[j * g.ldx];
            } else {
                const bool c_ok = c0 + c < g.C;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t m = r0 + kg * 8 + j;
                    bool ok = c_ok && m < kend;
                    if (shift != 0) {
                        const int f = (int)(m % g.Fr) + shift;
                        ok = ok && f >= 0 && f < g.Fr;
                    }
                    rb[i][j] = ok ? p[j * g.ldx] : 0.f;
                }
            }
        }
    };
    auto store_tiles = [&](int buf, int kt) {
        ddsp_u32x4* as = lds + buf * STAGE;
        ddsp_u32x4* bs = as + 8 * BM;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int q = tid + 256 * i, o = q % BM, kg = q / BM;
            ddsp_u32x4 hi, lo;
            ddsp_split8(ra[i], hi, lo);
            as[(2 * kg) * BM + o] = hi;
            as[(2 * kg + 1) * BM + o] = lo;
            if (want_bias) {
                if (g.bias_w) {
                    const int64_t m0 = kbeg + (int64_t)kt * 32 + kg * 8;
                    float wt[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) wt[j] = m0 + j < kend ? g.bias_w[(m0 + j) * g.ldw] : 0.f;
                    bsum[i] += ((ra[i][0] * wt[0] + ra[i][1] * wt[1]) + (ra[i][2] * wt[2] + ra[i][3] * wt[3])) +
                               ((ra[i][4] * wt[4] + ra[i][5] * wt[5]) + (ra[i][6] * wt[6] + ra[i][7] * wt[7]));
                } else
                    bsum[i] += ((ra[i][0] + ra[i][1]) + (ra[i][2] + ra[i][3])) + ((ra[i][4] + ra[i][5]) + (ra[i][6] + ra[i][7]));
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int q = tid + 256 * i, c = q % BN, kg = q / BN;
            ddsp_u32x4 hi, lo;
            ddsp_split8(rb[i], hi, lo);
            bs[(2 * kg) * BN + c] = hi;
            bs[(2 * kg + 1) * BN + c] = lo;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_tiles(0);
        store_tiles(0, 0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const ddsp_u32x4* as = lds + buf * STAGE + wm + lr;
        const ddsp_u32x4* bs = lds + buf * STAGE + 8 * BM + wn + lr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kg = 2 * s + lh;
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, as[(2 * kg) * BM + 32 * i]);
                al[i] = __builtin_bit_cast(bf16x8, as[(2 * kg + 1) * BM + 32 * i]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = __builtin_bit_cast(bf16x8, bs[(2 * kg) * BN + 32 * j]);
                bl[j] = __builtin_bit_cast(bf16x8, bs[(2 * kg + 1) * BN + 32 * j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1, kt + 1);
        __syncthreads();
    }

    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
    float* const part = g.partial + (int64_t)z * g.O * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = c0 + wn + 32 * j + lr;
            if (c >= g.C) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < g.O) part[(int64_t)o * N + tap * g.C + c] = acc[i][j][r];
            }
        }
    if (want_bias) {   // the four k-group owners of a channel meet in the LDS (the loop's last barrier has passed)
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int i = 0; i < TM; ++i) red[tid + 256 * i] = bsum[i];   // index = kg * BM + o
        __syncthreads();
        if (tid < BM && o0 + tid < g.O)
            g.bias_partial[(int64_t)z * (g.ldb ? g.ldb : g.O) + o0 + tid] = (red[tid] + red[BM + tid]) + (red[2 * BM + tid] + red[3 * BM + tid]);
    }
}

// rows per split: ~`want` splits, chunks a multiple of 32 rows
inline int chunk_for(int64_t M, int want) {
    int64_t chunk = (M + want - 1) / want;
    chunk = (chunk + 31) & ~(int64_t)31;
    return (int)chunk;
}
inline int splits_for(int64_t M, int chunk) { return (int)((M + chunk - 1) / chunk); }

template <int TM, int TN>
inline void launch(hipStream_t st, const Args& g, int batch = 0) {
    const int N = g.taps * g.C;
    dim3 grid((N + 64 * TN - 1) / (64 * TN), (g.O + 64 * TM - 1) / (64 * TM), batch > 0 ? batch : splits_for(g.M, g.chunk));
    hipLaunchKernelGGL((kernel<TM, TN>), grid, dim3(256), 0, st, g);
}

}  // namespace wgrad
