// fp32-in / fp32-accumulate MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: bit-exact fmaf chains).
//
// C[m][n] = epilogue( sum_k A(m,k) * B(k,n) ), optionally batched over blockIdx.z.
//
// Operand addressing (template flags say which index is contiguous in memory):
//   A_KC = true : A(m,k) = A[m*lda + k]   (row-major M x K: activations, frames x channels)
//   A_KC = false: A(m,k) = A[k*lda + m]   (stored K x M: transposed use, e.g. k'^T v)
//   A_MODE = A_CONV3: A is X (rows x Cin); k = tap*Cin + c and A(m,k) = X[m + tap - 1][c] when frame
//                 (m % Fr) + tap - 1 stays inside [0, Fr), else 0: a zero-padded k=3 Conv1d over the
//                 frame axis without materialising im2col.
//   A_MODE = A_CONVK: the same with `ktaps` taps `dil` frames apart, centred ("same" padding: tap t reads frame
//                 m + (t - (ktaps-1)/2)*dil): the dilated convolutions of the NSF-HiFiGAN generator.  The register-staged
//                 kernel can apply a leaky-ReLU (slope `in_slope`) to A as it is loaded; the LDS-DMA kernel (Cin % 32 == 0,
//                 in_slope == 1, `zeros` set) cannot - its callers feed it activated inputs.
//   A_MODE = A_FRAMES: rows are non-overlapping length-lda frames of B signals: row m lives at
//                 A[(m / Fr)*sA_hi + (m % Fr)*lda] (Fr frames per signal, signals sA_hi apart) - the STFT framing of
//                 the spectral loss without a copy.
//   B_KC = true : B(k,n) = B[n*ldb + k]   (weights stored [N][K] like nn.Linear)
//   B_KC = false: B(k,n) = B[k*ldb + n]   (stored K x N)
//
// Tiling: 256 threads = 4 wavefronts in a 2x2 grid; block tile BM x BN (64 or 128 each), wave tile
// (BM/2) x (BN/2) as TM x TN MFMA tiles of 32x32; BK = 32.  Both operands are staged through LDS in
// [k][m] / [k][n] order (leading dim BM+1 / BN+1) so the MFMA operand fetch (lane l: A[i=l&31][k=l>>5])
// is a conflict-free ds_read_b32 of 32 consecutive floats per k.  Global->register prefetch of tile
// t+1 is issued before the MFMAs of tile t and written to the other LDS buffer after them: one
// barrier per k-tile.  The fp32 MFMA issues once per 64 cycles per SIMD, so with 4-16 MFMAs per
// k-pair the staging traffic is far below the LDS and L2 rates; the kernel is bound by the fp32
// matrix pipe (157.3 TFLOP/s dense peak, MI355X_MICROARCH "Matrix cores").
#pragma once
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace gemm {

enum { A_PLAIN = 0, A_CONV3 = 1, A_FRAMES = 2, A_CONVK = 3 };

struct Args {
    const float* A;
    const float* B;
    int M, N, K;
    int64_t lda, ldb;
    // batching over blockIdx.z: z -> (z / zdiv, z % zdiv)
    int zdiv;
    int64_t sA_hi, sA_lo, sB_hi, sB_lo;
    // conv3 / convK mode
    int Fr, Cin;
    int tap_shift;        // A_CONV3: 0 = centred taps (frames -1, 0, +1), -1 = causal taps (frames -2, -1, 0)
    int ktaps, dil;       // A_CONVK
    float in_slope;       // A_CONVK: leaky-ReLU slope applied to A on load (1 = none)
    const float* zeros;   // >= Cin + 32 zero floats (DMA kernel, conv3 mode: source of the taps that fall off an utterance)
    // remainder mode of the DMA kernel (sub_from > 0): tile t of this launch is quadrant t & 3 of the PARENT tiling's tile
    // sub_from + t / 4, the parent tiles being twice as large in both directions and parent_tn of them per row
    int sub_from, parent_tn;
    // product arithmetic of the DMA kernel: 0 = fp32 MFMA, 3 = split-bf16 (3 bf16 MFMAs per fp32 product, ~4e-6 relative
    // error per GEMM instead of 3e-7, 1.5-1.8x faster at the control network's shapes); set per call by the caller
    int math;
    // optional copy of B for math == 3 on the DMA kernel, same shape and pitch, already split: every group of 8 consecutive
    // k-values (32 bytes) holds 8 bf16 hi parts then 8 bf16 lo parts (ddsp_presplit_b).  A lane's B fragment is four
    // 16-byte slots of one row, so slots 2h / 2h+1 ARE the hi / lo operand of half h and the kernel does not split B.
    // Paths that do not use it (fp32 products, the register-staged kernel) read B.
    const float* B_split;
    // A is ALREADY in that (8 hi | 8 lo) layout (written so by its producer: LayerNorm, GroupNorm+LeakyReLU, the attention
    // kernel): with math == 3 and B_split the kernel then splits nothing (MATH = 8: bit casts and MFMAs only)
    int A_split;
    // DMA kernel, one batch: XCD-aware tile order.  Workgroup b runs on XCD b & 7; with xcd != 0 XCD x owns the row blocks
    // x, x + 8, ... with ALL their column tiles, so the A rows of a row block are fetched into one L2 instead of eight
    // (the leftover tiles_m % 8 row blocks are dealt round-robin as before).
    int xcd;
    // ... and the launch covers all tiles but the last xcd_cut of the linear order (a multiple of 8, at most the leftover
    // region: every XCD's queue is one tile shorter per 8), which a second launch runs as quadrant tiles (sub_from)
    int xcd_cut;
    // DMA kernel: batch index z also starts its k range kz floats into A's and B's rows (the taps of an implicit-im2col
    // convolution follow from the shifted k): a K-split whose partial products the caller's epilogue stores per z
    int kz;
};

constexpr int BK = 32;
// global 16-byte loads at dword alignment (rows of a signal framed at an arbitrary length N start anywhere)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

// An epilogue that declares `static constexpr bool kPair = true` is called as epi(z, m, n, v, v_of_column_n^1).
template <class E, class = void>
struct epi_wants_pair : std::false_type {};
template <class E>
struct epi_wants_pair<E, std::void_t<decltype(E::kPair)>> : std::true_type {};
// An epilogue that offers `vec_ok()` / `store4(z, m, n, f32x4)` gets whole 16-byte row pieces: the accumulator tile
// is transposed 4x4 across lane quads first, so that one store instruction writes eight 128-byte row segments
// instead of two (dword stores are issue-bound: tools/gemm_ab.py, 20 us of 95 at N=1536, K=256).
template <class E, class = void>
struct epi_has_store4 : std::false_type {};
template <class E>
struct epi_has_store4<E, std::void_t<decltype(E::kStore4)>> : std::true_type {};

// An epilogue that declares `kGatedPair` (DMA kernel, 128-wide tiles only: each wave owns two 32-column blocks) gets
// the two blocks of a wave as (value, gate): B rows are packed so that packed columns 64t..64t+31 are the values and
// 64t+32..64t+63 the gates of output channels 32t..32t+31.  Both then sit in the same lane and register, the gated
// product is formed in place and only HALF the columns are stored: out[m][32t + c] = (a + bias_a) * sigmoid(g + bias_g).
template <class E, class = void>
struct epi_is_gated : std::false_type {};
template <class E>
struct epi_is_gated<E, std::void_t<decltype(E::kGatedPair)>> : std::true_type {};

// 4x4 transpose across the four lanes of a quad: in x[i] = (row i, column q) for lane q; out x[k] = (row q, column k)
__device__ __forceinline__ void quad_transpose(float (&x)[4], int q) {
    const bool odd = q & 1, hi = q & 2;
    auto swap1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); };
    auto swap2 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); };
    // round 1 (lane ^ 1): even lanes end with rows 0 and 2, odd lanes with rows 1 and 3, each for columns (q&~1, q|1)
    const float r0 = swap1(odd ? x[0] : x[1]);
    const float r1 = swap1(odd ? x[2] : x[3]);
    const float a_lo = odd ? r0 : x[0], a_hi = odd ? x[1] : r0;   // row (q&1)
    const float b_lo = odd ? r1 : x[2], b_hi = odd ? x[3] : r1;   // row 2 + (q&1)
    // round 2 (lane ^ 2): lanes 0,1 collect columns 2,3 of their low row, lanes 2,3 columns 0,1 of their high row
    const float s_lo = swap2(hi ? a_lo : b_lo);
    const float s_hi = swap2(hi ? a_hi : b_hi);
    x[0] = hi ? s_lo : a_lo;
    x[1] = hi ? s_hi : a_hi;
    x[2] = hi ? b_lo : s_lo;
    x[3] = hi ? b_hi : s_hi;
}

template <int BM, int BN, bool A_KC, bool B_KC, int A_MODE, class Epi, int NW = 4>
__global__ void __launch_bounds__(64 * NW) kernel(Args g, Epi epi) {
    // wave grid: 2x2 (NW = 4) or 4(M) x 2(N) (NW = 8); wave tile = TM x TN MFMA tiles of 32x32
    constexpr int WGM = NW / 2, WGN = 2, NT = 64 * NW;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    static_assert(TM >= 1 && TN >= 1, "tile too small for the wave grid");
    constexpr int LDM = BM + 1, LDN = BN + 1;
    __shared__ float lds[2 * BK * (LDM + LDN)];
    float* const As0 = lds;                 // two [BK][LDM] buffers
    float* const Bs0 = lds + 2 * BK * LDM;  // two [BK][LDN] buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / WGM);
    const int wn = (wave & 1) * (BN / WGN);
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const float* __restrict__ A = g.A + (A_MODE == A_FRAMES ? 0 : (int64_t)(z / g.zdiv) * g.sA_hi + (int64_t)(z % g.zdiv) * g.sA_lo);
    const float* __restrict__ B = g.B + (int64_t)(z / g.zdiv) * g.sB_hi + (int64_t)(z % g.zdiv) * g.sB_lo;

    // staging registers: BM*BK/256/4 float4 for A, BN*BK/256/4 for B
    constexpr int NA = BM * BK / (4 * NT), NB = BN * BK / (4 * NT);
    static_assert(NA >= 1 && NB >= 1, "tile too small for the thread count");
    f32x4 ra[NA], rb[NB];

    auto arow = [&](int m) -> const float* {
        if (A_MODE == A_FRAMES) return A + (int64_t)(m / g.Fr) * g.sA_hi + (int64_t)(m % g.Fr) * g.lda;
        return A + (int64_t)m * g.lda;
    };
    // Interior tiles (the common case) take a branch-free path; edge tiles bounds-check every element.
    auto load_tiles = [&](int k0) {
        const bool k_full = (k0 + BK <= g.K);
        const bool a_full = k_full && (m0 + BM <= g.M);
        const bool b_full = k_full && (n0 + BN <= g.N);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT;  // float4 index in the tile
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (A_KC) {
                const int r = q / (BK / 4), kq = q % (BK / 4);
                const int m = m0 + r, k = k0 + kq * 4;
                if (A_MODE == A_CONV3) {
                    // k = tap*Cin + c ; Cin % 4 == 0 so a float4 never straddles taps
                    if (m < g.M && k < g.K) {
                        const int tap = k / g.Cin, c = k - tap * g.Cin;
                        const int off = tap - 1 + g.tap_shift;
                        const int f = m % g.Fr + off;
                        if (f >= 0 && f < g.Fr) v = *(const f32x4_u*)(A + (int64_t)(m + off) * g.lda + c);
                    }
                } else if (A_MODE == A_CONVK) {
                    if (m < g.M && k < g.K) {
                        const int tap = k / g.Cin, c = k - tap * g.Cin;
                        const int off = (tap - (g.ktaps - 1) / 2) * g.dil;
                        const int f = m % g.Fr + off;
                        if (f >= 0 && f < g.Fr) {
                            v = *(const f32x4_u*)(A + (int64_t)(m + off) * g.lda + c);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * g.in_slope;
                        }
                    }
                } else if (a_full) {
                    v = *(const f32x4_u*)(arow(m) + k);
                } else if (m < g.M) {
                    const float* p = arow(m) + k;
                    if (k + 3 < g.K) {
                        v = *(const f32x4_u*)p;
                    } else {
                        if (k < g.K) v[0] = p[0];
                        if (k + 1 < g.K) v[1] = p[1];
                        if (k + 2 < g.K) v[2] = p[2];
                    }
                }
            } else {
                const int kr = q / (BM / 4), mq = q % (BM / 4);
                const int k = k0 + kr, m = m0 + mq * 4;
                if (a_full) {
                    v = *(const f32x4_u*)(A + (int64_t)k * g.lda + m);
                } else if (k < g.K) {
                    const float* p = A + (int64_t)k * g.lda + m;
                    if (m + 3 < g.M) {
                        v = *(const f32x4_u*)p;
                    } else {
                        if (m < g.M) v[0] = p[0];
                        if (m + 1 < g.M) v[1] = p[1];
                        if (m + 2 < g.M) v[2] = p[2];
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (B_KC) {
                const int r = q / (BK / 4), kq = q % (BK / 4);
                const int n = n0 + r, k = k0 + kq * 4;
                if (b_full) {
                    v = *(const f32x4_u*)(B + (int64_t)n * g.ldb + k);
                } else if (n < g.N) {
                    const float* p = B + (int64_t)n * g.ldb + k;
                    if (k + 3 < g.K) {
                        v = *(const f32x4_u*)p;
                    } else {
                        if (k < g.K) v[0] = p[0];
                        if (k + 1 < g.K) v[1] = p[1];
                        if (k + 2 < g.K) v[2] = p[2];
                    }
                }
            } else {
                const int kr = q / (BN / 4), nq = q % (BN / 4);
                const int k = k0 + kr, n = n0 + nq * 4;
                if (b_full) {
                    v = *(const f32x4_u*)(B + (int64_t)k * g.ldb + n);
                } else if (k < g.K) {
                    const float* p = B + (int64_t)k * g.ldb + n;
                    if (n + 3 < g.N) {
                        v = *(const f32x4_u*)p;
                    } else {
                        if (n < g.N) v[0] = p[0];
                        if (n + 1 < g.N) v[1] = p[1];
                        if (n + 2 < g.N) v[2] = p[2];
                    }
                }
            }
            rb[i] = v;
        }
    };

    auto store_tiles = [&](int buf) {
        float* as = As0 + buf * (BK * LDM);
        float* bs = Bs0 + buf * (BK * LDN);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT;
            if (A_KC) {
                const int r = q / (BK / 4), kq = q % (BK / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) as[(kq * 4 + j) * LDM + r] = ra[i][j];
            } else {
                const int kr = q / (BM / 4), mq = q % (BM / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) as[kr * LDM + mq * 4 + j] = ra[i][j];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT;
            if (B_KC) {
                const int r = q / (BK / 4), kq = q % (BK / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) bs[(kq * 4 + j) * LDN + r] = rb[i][j];
            } else {
                const int kr = q / (BN / 4), nq = q % (BN / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) bs[kr * LDN + nq * 4 + j] = rb[i][j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (g.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const float* as = As0 + buf * (BK * LDM) + wm + lr + lh * LDM;
        const float* bs = Bs0 + buf * (BK * LDN) + wn + lr + lh * LDN;
        // operand fetch is software-pipelined one k-pair ahead of the MFMAs that consume it
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = as[32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = bs[32 * j];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float an[TM], bn[TN];
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) an[i] = as[(kk + 2) * LDM + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) bn[j] = bs[(kk + 2) * LDN + 32 * j];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            // pin the order [LDS reads of k-pair kk+2][MFMAs of k-pair kk] (hipcc otherwise sinks the reads
            // back to just before their use, re-exposing the LDS latency every 4 MFMAs)
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = an[i];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = bn[j];
            }
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + 32 * j + lr;
            const bool n_ok = n < g.N;
            // per-column operand of the epilogue (bias), fetched once instead of once per accumulator register
            const float cb = n_ok ? epi.col(n) : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if constexpr (epi_wants_pair<Epi>::value) {
                    // columns (2f, 2f+1) sit on adjacent lanes: hand each lane its neighbour's value too
                    const float other = __shfl_xor(acc[i][j][r], 1, 64);
                    if (n_ok && m < g.M) epi(z, m, n, acc[i][j][r], other);
                } else {
                    if (n_ok && m < g.M) epi(z, m, n, acc[i][j][r], cb);
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------------------
// LDS-DMA pipelined variant for the common case (row-major A, nn.Linear-layout B, K % 32 == 0, 16-byte aligned
// rows): the activations/weights of k-tile t+2 stream HBM/L2 -> LDS with `global_load_lds_dwordx4` (no VGPR
// staging, no ds_write pass) into a 3-stage ring while tile t feeds the MFMAs; a counted `s_waitcnt vmcnt(N)`
// keeps one tile in flight across the raw `s_barrier`.  With K = 256..1024 and 8..32 k-tiles per block the
// register-staged kernel above exposes one global-load latency per k-tile; here it is hidden behind two tiles.
//
// LDS image per stage: (BM + BN) rows of 32 floats (128 B), unpadded (one DMA wave-instruction writes 1 KiB =
// 8 whole rows).  The 16-byte slot p of row r holds global slot p ^ ((r >> 1) & 7) (swizzle applied on the SOURCE
// address), so that the MFMA operand fetch - lane (i = l&31, h = l>>5) reads its 16 k-values 16h..16h+15 of row i
// as four ds_read_b128 - is bank-conflict-free.  The MFMA k-order is permuted (step s of lane half h is
// k = 16h + s) identically for A and B, which leaves the sum unchanged.
// NW = 8 waves as a 4(M) x 2(N) grid, or NW = 4 as 2 x 2 (64x64 tiles: four times as many workgroups for the skinny
// N = 256 layers, whose 128-row tilings leave a third of the CUs without work).
// MATH: 0 = fp32 MFMA; 3 = split-bf16 (Args::math == 3: the inference GEMMs): each fp32 operand is split into bf16 hi and
// lo pieces in registers and the product formed from 3 bf16 MFMAs (32x32x16: hi*hi + hi*lo + lo*hi) with fp32 accumulation;
// 7 = the same with B read already split (Args::B_split); 8 = A and B both already split (Args::A_split); 6 = three pieces,
// 6 MFMAs (ddsp_gemm_f32 tile 31 only).
// Speed and error: DESIGN.md section 9.
template <int BM, int BN, class Epi, int NS = 3, int ABLATE = 0, int NW = 8, int A_MODE = A_PLAIN, int MATH = 0>  // ABLATE bit mask (timing experiments only): 1 no MFMA, 2 no DMA, 4 no epilogue stores, 8 no barrier
__global__ void __launch_bounds__(64 * NW, (NW == 4 ? 3 : BM * BN <= 128 * 128 ? 4 : 2)) kernel_dma(Args g, Epi epi, int tiles_m, int tiles_n, int total_tiles) {
    constexpr int WGM = NW / 2, WGN = 2;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int ROWS = BM + BN;            // rows per stage
    constexpr int STAGE = ROWS * 32;         // floats per stage
    constexpr int PIECES = ROWS / 8;         // 1-KiB DMA pieces per stage
    constexpr int PPW = PIECES / NW;         // pieces per wave
    static_assert(PIECES % NW == 0, "stage must split evenly over the waves");
    __shared__ __attribute__((aligned(1024))) float lds[NS * STAGE];   // NS-stage ring (NS-1 tiles in flight)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int wm = (wave >> 1) * (BM / WGM), wn = (wave & 1) * (BN / WGN);
    const int lr = lane & 31, lh = lane >> 5;
    const int nk = g.K / 32;

    // Persistent workgroup: walks output tiles blockIdx.x, +gridDim.x, ... and treats their k-tiles as ONE stream
    // (step = tile_i*nk + kt), so the DMA of the next tile's first k-tiles is already in flight while the current
    // tile's last MFMAs and its epilogue stores run: no per-tile prologue latency.
    const bool xcd_order = g.xcd && (gridDim.x & 7) == 0 && g.sub_from == 0 && total_tiles == tiles_m * tiles_n - g.xcd_cut;
    const int xq0 = (int)blockIdx.x >> 3, xgs = (int)gridDim.x >> 3, xx = (int)blockIdx.x & 7;
    const int xown = (tiles_m / 8) * tiles_n, xleft = (tiles_m % 8) * tiles_n - g.xcd_cut;
    int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (xcd_order) {
        const int qx = xown + (xleft > xx ? (xleft - xx + 7) / 8 : 0);
        my_tiles = xq0 < qx ? (qx - xq0 + xgs - 1) / xgs : 0;
    }
    // the i-th tile of this workgroup as a linear id mt * tiles_n + nt
    auto tile_of = [&](int i) -> int {
        if (!xcd_order) return (int)blockIdx.x + i * (int)gridDim.x;
        const int q = xq0 + i * xgs;
        if (q < xown) return (xx + 8 * (q / tiles_n)) * tiles_n + q % tiles_n;
        const int u = (q - xown) * 8 + xx;
        return ((tiles_m / 8) * 8 + u / tiles_n) * tiles_n + u % tiles_n;
    };
    const int steps = my_tiles * nk;

    // per-lane source pointers (k = 0) of this wave's DMA pieces for a given tile; rows past the edge re-read
    // the last row (their products are never stored)
    // conv3 mode (implicit im2col, k = tap*Cin + c, A[m][k] = X[m + tap - 1][c] inside the utterance, else 0): the A
    // pieces keep one source pointer per tap; a tap that falls off the utterance points at the zero page.  Cin % 32 == 0,
    // so a k-step never straddles taps and the tap is wave-uniform.
    constexpr int PA = (BM / 8) / NW;        // this wave's pieces i < PA are A rows, the others B rows
    static_assert((BM / 8) % NW == 0, "A rows must split evenly over the waves");
    const float* src_m1[PA];                 // taps 0 and 2 (tap 1 lives in src[])
    const float* src_p1[PA];
    int conv_f[PA];                          // A_CONVK: frame (inside its utterance) of each A row this wave fetches
    int kbase_cur = 0;                       // Args::kz: k offset of the tile whose pieces are being issued
    auto tile_src = [&](int tile, const float* (&src)[PPW]) {
        const int per_z = tiles_m * tiles_n;
        int z = tile / per_z, rem = tile - z * per_z;
        int m0 = (rem / tiles_n) * BM, n0 = (rem % tiles_n) * BN;
        if (g.sub_from > 0) {                     // remainder mode: quadrants of the parent tiling's last tiles
            const int big = g.sub_from + (tile >> 2), q = tile & 3;
            z = 0;
            m0 = (2 * (big / g.parent_tn) + (q >> 1)) * BM;
            n0 = (2 * (big % g.parent_tn) + (q & 1)) * BN;
        }
        const float* A = g.A + (int64_t)(z / g.zdiv) * g.sA_hi + (int64_t)(z % g.zdiv) * g.sA_lo;
        const float* B = g.B + (int64_t)(z / g.zdiv) * g.sB_hi + (int64_t)(z % g.zdiv) * g.sB_lo;
        kbase_cur = z * g.kz;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + NW * i;
            const int row = piece * 8 + (lane >> 3);         // row of the stage image
            const int slot = (lane & 7) ^ ((row >> 1) & 7);  // logical 16-byte slot this lane fetches
            if (row < BM) {
                int m = m0 + row;
                m = m < g.M ? m : g.M - 1;
                src[i] = A + (int64_t)m * g.lda + slot * 4;
                if constexpr (A_MODE == A_CONVK) {
                    // (k = tap*Cin + c; the tap of a k-step is wave-uniform because Cin % 32 == 0: issue() moves the row
                    // pointer by the tap's frame offset, or to the zero page when that frame is outside the utterance)
                    if (i < PA) conv_f[i < PA ? i : 0] = m % g.Fr;
                }
                if constexpr (A_MODE == A_CONV3) {
                    if (i < PA) {
                        const int f = m % g.Fr;
                        // taps at frames f - 1 + s, f + s, f + 1 + s (s = tap_shift); off the utterance -> the zero page
                        const int s = g.tap_shift;
                        const float* base = src[i];
                        const float* zp = g.zeros + slot * 4;
                        src_m1[i < PA ? i : 0] = (f - 1 + s >= 0 && f - 1 + s < g.Fr) ? base + (int64_t)(s - 1) * g.lda : zp;
                        src_p1[i < PA ? i : 0] = (f + 1 + s >= 0 && f + 1 + s < g.Fr) ? base + (int64_t)(s + 1) * g.lda : zp;
                        src[i] = (f + s >= 0 && f + s < g.Fr) ? base + (int64_t)s * g.lda : zp;
                    }
                }
            } else {
                int n = n0 + row - BM;
                n = n < g.N ? n : g.N - 1;
                src[i] = B + (int64_t)n * g.ldb + slot * 4;
            }
        }
    };
    auto issue = [&](const float* const (&src)[PPW], int kt, int stage) {
        int tap = 1, koff = kt * 32 + kbase_cur;
        if constexpr (A_MODE == A_CONV3 || A_MODE == A_CONVK) {
            tap = koff / g.Cin;
            koff -= tap * g.Cin;
        }
        const int conv_off = A_MODE == A_CONVK ? (tap - (g.ktaps - 1) / 2) * g.dil : 0;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + NW * i;
            const float* p = src[i] + kt * 32 + kbase_cur;
            if constexpr (A_MODE == A_CONV3) {
                if (i < PA) p = (tap == 0 ? src_m1[i < PA ? i : 0] : tap == 1 ? src[i] : src_p1[i < PA ? i : 0]) + koff;
            }
            if constexpr (A_MODE == A_CONVK) {
                if (i < PA) {
                    const int f = conv_f[i < PA ? i : 0] + conv_off;
                    p = (f >= 0 && f < g.Fr) ? src[i] + (int64_t)conv_off * g.lda + koff : g.zeros + (lane & 7) * 4;
                }
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(lds + stage * STAGE + piece * 256),
                                             16, 0, 0);
        }
    };

    const float* src_cur[PPW];   // tile that step `issued` belongs to
    int issued = 0;              // next step whose DMA has not been issued yet
    int issue_tile = 0, issue_kt = 0;
    if (steps > 0) tile_src(tile_of(0), src_cur);
    auto issue_next = [&]() {
        if (issued >= steps) return;
        if (!(ABLATE & 2)) issue(src_cur, issue_kt, issued % NS);
        ++issued;
        if (++issue_kt == nk) {
            issue_kt = 0;
            ++issue_tile;
            if (issue_tile < my_tiles) tile_src(tile_of(issue_tile), src_cur);
        }
    };
    // NS - 1 steps in flight
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) issue_next();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int kt = 0, tile_i = 0;
    for (int step = 0; step < steps; ++step) {
        // step has landed once at most the next step's pieces of THIS wave are still in flight ...
        // (deeper rings: the last NS - 3 steps of the stream wait for everything, their successors being fewer than NS - 2)
        static_assert((NS - 2) * PPW <= 63, "vmcnt is a 6-bit counter");
        if (NS > 2 && step + NS - 2 < steps)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and every other wave says the same; the barrier also fences the reads of stage (step+2)%3 (step-1)
        if (!(ABLATE & 8)) __builtin_amdgcn_s_barrier();
        issue_next();  // step + 2
        const float* st = lds + (step % NS) * STAGE;
        f32x4 av[TM][4], bv[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm + 32 * i + lr;
            const float* rp = st + row * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) av[i][c] = *(const f32x4*)(rp + 4 * ((4 * lh + c) ^ ((row >> 1) & 7)));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = BM + wn + 32 * j + lr;
            const float* rp = st + row * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[j][c] = *(const f32x4*)(rp + 4 * ((4 * lh + c) ^ ((row >> 1) & 7)));
        }
        if (ABLATE & 1) {
            // keep the operand reads alive without the matrix pipe
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][0] += av[i][0][0] + bv[j][3][3];
        } else if constexpr (MATH != 0) {
            // split-bf16 experiment: the lane's 16 k-values form two K=16 steps of 8 values per lane half; A and B use the
            // same (half, element) slots, so whatever k order the instruction assigns to them the pairs match
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            // two values at a time: one v_cvt_pk_bf16_f32 (round to nearest even) per piece and pair, the piece widened back
            // by a shift / a mask, the remainders by one packed subtract - 5 vector instructions per pair and piece
            typedef float f32x2p __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
            typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
            auto split = [](const f32x4& x0, const f32x4& x1, bf16x8 (&p)[3]) {
                constexpr int NQ = MATH == 6 ? 3 : 2;
                u32x4p w[3];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    f32x2p r = d < 2 ? (f32x2p){x0[2 * d], x0[2 * d + 1]} : (f32x2p){x1[2 * d - 4], x1[2 * d - 3]};
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2p));
                        w[q][d] = h;
                        if (q + 1 < NQ)
                            r = r - (f32x2p){__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
                    }
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) p[q] = __builtin_bit_cast(bf16x8, w[q]);
            };
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf16x8 ap[TM][3], bp[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (MATH == 8) {   // A arrives split (Args::A_split)
                        ap[i][0] = __builtin_bit_cast(bf16x8, av[i][2 * half]);
                        ap[i][1] = __builtin_bit_cast(bf16x8, av[i][2 * half + 1]);
                    } else {
                        split(av[i][2 * half], av[i][2 * half + 1], ap[i]);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (MATH == 7 || MATH == 8) {   // B arrives split (Args::B_split)
                        bp[j][0] = __builtin_bit_cast(bf16x8, bv[j][2 * half]);
                        bp[j][1] = __builtin_bit_cast(bf16x8, bv[j][2 * half + 1]);
                    } else {
                        split(bv[j][2 * half], bv[j][2 * half + 1], bp[j]);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        auto mm = [&](int x, int y) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][x], bp[j][y], acc[i][j], 0, 0, 0);
                        };
                        if constexpr (MATH == 6) {      // smallest terms first
                            mm(2, 0);
                            mm(0, 2);
                            mm(1, 1);
                        }
                        mm(1, 0);
                        mm(0, 1);
                        mm(0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s >> 2][s & 3], bv[j][s >> 2][s & 3],
                                                                         acc[i][j], 0, 0, 0);
        }
        if (++kt == nk) {
            // ---- epilogue of this tile (stores drain while the next tile's MFMAs run) ----
            const int tile = tile_of(tile_i);
            const int per_z = tiles_m * tiles_n;
            int z = tile / per_z, rem = tile - z * per_z;
            int m0 = (rem / tiles_n) * BM, n0 = (rem % tiles_n) * BN;
            if (g.sub_from > 0) {
                const int big = g.sub_from + (tile >> 2), q = tile & 3;
                z = 0;
                m0 = (2 * (big / g.parent_tn) + (q >> 1)) * BM;
                n0 = (2 * (big % g.parent_tn) + (q & 1)) * BN;
            }
            bool vec = false;
            if constexpr (epi_has_store4<Epi>::value) vec = epi.vec_ok() && n0 + BN <= g.N;
            if constexpr (epi_is_gated<Epi>::value) {
                static_assert(TN == 2, "gated-pair epilogue needs the wave's two column blocks");
                // (the launcher has checked N % BN == 0 and the output alignment)
                const int q = lane & 3;
                const int na = n0 + wn + lr;                       // packed column of this lane's value; gate at +32
                const float ba = epi.col(na), bg = epi.col(na + 32);
                const int oc = ((n0 + wn) >> 1) + (lr & ~3);       // first of the lane's four output channels
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        float x[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = acc[i][0][4 * rg + r] + ba, gt = acc[i][1][4 * rg + r] + bg;
                            // split-bf16 products (values carry ~4e-6): sigmoid by v_exp_f32 / v_rcp_f32 (~2 ulp) instead of expf and an IEEE
                            // division - 0.44 of this launch's vector issue was the epilogue; fp32 products keep the exact form
                            if constexpr (MATH != 0)
                                x[r] = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * gt));
                            else
                                x[r] = a * (1.0f / (1.0f + expf(-gt)));
                            acc[i][0][4 * rg + r] = 0.f;
                            acc[i][1][4 * rg + r] = 0.f;
                        }
                        quad_transpose(x, q);
                        const int m = m0 + wm + 32 * i + q + 8 * rg + 4 * lh;
                        if (m < g.M) epi.store4(z, m, oc, f32x4{x[0], x[1], x[2], x[3]});
                    }
            } else if (vec) {
                if constexpr (epi_has_store4<Epi>::value) {
                    const int q = lane & 3;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = n0 + wn + 32 * j + (lr & ~3);   // first of this lane's four columns
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                float x[4] = {acc[i][j][4 * rg], acc[i][j][4 * rg + 1], acc[i][j][4 * rg + 2], acc[i][j][4 * rg + 3]};
                                quad_transpose(x, q);
                                const int m = m0 + wm + 32 * i + q + 8 * rg + 4 * lh;
                                if constexpr ((ABLATE & 4) != 0) {
                                    if (x[0] == 12345.678f) epi.store4(z, m, n, f32x4{x[0], x[1], x[2], x[3]});
                                } else {
                                    if (m < g.M) epi.store4(z, m, n, f32x4{x[0], x[1], x[2], x[3]});
                                }
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[i][j][4 * rg + r] = 0.f;
                            }
                        }
                }
            } else
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn + 32 * j + lr;
                    const bool n_ok = n < g.N;
                    const float cb = n_ok ? epi.col(n) : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if constexpr (epi_wants_pair<Epi>::value) {
                            const float other = __shfl_xor(acc[i][j][r], 1, 64);
                            if (n_ok && m < g.M) epi(z, m, n, acc[i][j][r], other);
                        } else if constexpr ((ABLATE & 4) != 0) {
                            if (acc[i][j][r] == 12345.678f) epi(z, m, n, acc[i][j][r], cb);
                        } else {
                            if (n_ok && m < g.M) epi(z, m, n, acc[i][j][r], cb);
                        }
                        acc[i][j][r] = 0.f;
                    }
                }
            kt = 0;
            ++tile_i;
        }
    }
}

inline bool dma_ok(const Args& g) {
    return g.K % 32 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 &&
           g.sA_hi % 4 == 0 && g.sA_lo % 4 == 0 && g.sB_hi % 4 == 0 && g.sB_lo % 4 == 0;
}

template <int BM, int BN, class Epi, int NS = 3, int ABLATE = 0, int NW = 8, int A_MODE = A_PLAIN, int MATH = 0>
inline void launch_dma(hipStream_t st, const Args& g, int batch, const Epi& epi, int total_override = -1) {
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int total = total_override >= 0 ? total_override : tiles_m * tiles_n * batch;
    if (total == 0) return;
    // resident workgroups per CU by LDS (NS stages of (BM+BN) x 128 B), 8 waves each
    constexpr int lds_bytes = NS * (BM + BN) * 128;
    constexpr int by_waves = 16 / NW;   // 4 waves per SIMD
    constexpr int by_lds = 160 * 1024 / lds_bytes;
    constexpr int per_cu = by_lds < by_waves ? by_lds : by_waves;
    int grid = 256 * per_cu;
    if (grid > total) grid = total;
    hipLaunchKernelGGL((kernel_dma<BM, BN, Epi, NS, ABLATE, NW, A_MODE, MATH>), dim3(grid), dim3(64 * NW), 0, st, g, epi, tiles_m, tiles_n, total);
}

// launch_dma with the product arithmetic chosen at run time (Args::math)
// XCD-aware tile order: on for the split-bf16 GEMMs, whose loops no longer hide the eightfold A re-reads of the
// round-robin order (r02: Linear family 0.395 -> 0.373 ms, HBM/fabric traffic of a pw1 call 114 -> ~35 MB); off for fp32
// products, where round 1 measured it slower (0.76 -> 0.84 ms).  DDSP_GEMM_XCD = 0 / 1 forces it (measurement aid).
inline int xcd_default(int math) {
    static int env = -2;
    if (env == -2) {
        const char* e = getenv("DDSP_GEMM_XCD");
        env = e ? (e[0] == '1' ? 1 : 0) : -1;
    }
    return env >= 0 ? env : (math == 3 ? 1 : 0);
}

template <int BM, int BN, class Epi, int NS = 3, int NW = 8, int A_MODE = A_PLAIN>
inline void dma_go(hipStream_t st, const Args& g0, int batch, const Epi& epi, int total_override = -1) {
    Args g = g0;
    if (g.xcd < 0) g.xcd = (batch == 1 && (total_override < 0 || g.xcd_cut > 0)) ? xcd_default(g.math) : 0;
    if (g.math == 3 && g.B_split && g.A_split) {
        Args h = g;
        h.B = g.B_split;
        launch_dma<BM, BN, Epi, NS, 0, NW, A_MODE, 8>(st, h, batch, epi, total_override);
    } else if (g.math == 3 && g.B_split) {
        Args h = g;
        h.B = g.B_split;
        launch_dma<BM, BN, Epi, NS, 0, NW, A_MODE, 7>(st, h, batch, epi, total_override);
    } else if (g.math == 3)
        launch_dma<BM, BN, Epi, NS, 0, NW, A_MODE, 3>(st, g, batch, epi, total_override);
    else if (g.math == 6)   // three bf16 pieces per operand, six products: fp32-class error (2.4e-7), callers that ask for it
        launch_dma<BM, BN, Epi, NS, 0, NW, A_MODE, 6>(st, g, batch, epi, total_override);
    else
        launch_dma<BM, BN, Epi, NS, 0, NW, A_MODE, 0>(st, g, batch, epi, total_override);
}

template <int BM, int BN, bool A_KC, bool B_KC, int A_MODE, class Epi, int NW = 4>
inline void launch_tile(hipStream_t st, const Args& g, int batch, const Epi& epi) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch);
    hipLaunchKernelGGL((kernel<BM, BN, A_KC, B_KC, A_MODE, Epi, NW>), grid, dim3(64 * NW), 0, st, g, epi);
}

// workgroups from which the 128x128 tile is preferred to 128x64 (DDSP_GEMM_T128_MIN: measurement aid; 256 instead of 512:
// training step 6.40 -> 6.51 ms, the B = 47 forward 1.05 -> 1.14 ms - unlike the enhancer's dilated convolutions, which gain
// from 128x128 tiles from 256 workgroups on, nsf.hip)
inline int64_t t128_min() {
    static int64_t v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_GEMM_T128_MIN");
        v = e ? atoll(e) : 512;
    }
    return v;
}

// Tile choice (measured on MI355X, tools/gemm_ab.py, fp32 TFLOP/s at M = 11008):
//   K%32==0, row-major A, [N][K] B  -> persistent LDS-DMA kernel, 128x64 tiles, 8 waves  (69-83 at K=256, N=512..1536)
//   otherwise                        -> register-staged kernel: 128x64 / 8 waves when the grid fills the chip,
//                                       64x64 / 4 waves for small or skinny problems.
// Larger tiles raise FLOP per staged byte (the L2->LDS operand stream, ~4-6 TB/s chip-wide, is what caps these
// K=256..768 shapes near 85 TFLOP/s) but leave CUs idle at 11008 = 2*43*128 rows; see DESIGN.md section 9.
inline bool half_tiles() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_GEMM_HALF_TILES");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// DDSP_GEMM_TAIL_CUT=1 (measurement aid, off): with the XCD-aware order too, the few tiles of a last, nearly empty round are cut
// out of the launch and run as quadrant tiles in a second one.  Measured at the bench shape (QKV: 1032 = 2 x 512 + 8 tiles):
// Linear family 0.361 -> 0.372 ms - the eight tail tiles run alone on their CUs and take about what the second launch costs.
inline bool tail_cut() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DDSP_GEMM_TAIL_CUT");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

template <bool A_KC, bool B_KC, int A_MODE, class Epi>
inline void launch(hipStream_t st, const Args& g, int batch, const Epi& epi) {
    auto blocks = [&](int bm, int bn) {
        return (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * batch;
    };
    if constexpr (A_KC && B_KC && A_MODE == A_CONV3) {
        // implicit-im2col convs (N = 256): the 4-wave 64x64 DMA tile with one source pointer per tap
        if (dma_ok(g) && g.zeros && g.Cin % 32 == 0 && g.Cin + 32 <= DDSP_ZERO_FLOATS && g.N <= 256) {
            // (also for a few tiles: see the small-problem note below)
            if (half_tiles() && blocks(64, 64) < 128 && blocks(64, 64) > 16)
                dma_go<32, 64, Epi, 4, 2, A_CONV3>(st, g, batch, epi);
            else
                dma_go<64, 64, Epi, 3, 4, A_CONV3>(st, g, batch, epi);
            return;
        }
    }
    if constexpr (A_KC && B_KC && A_MODE == A_PLAIN) {
        // a few tiles and a long K (the 87-frame real-time block: 8 workgroups walking K = 512): the 3-stage DMA ring keeps
        // two k-tiles in flight, the register-staged kernel's one-tile prefetch leaves a global-load latency per k-tile
        // exposed (real-time block 0.67 -> 0.50 ms).  DDSP_GEMM_SMALL_DMA=0 restores the old choice (measurement aid).
        static int small_dma = -1;
        if (small_dma < 0) {
            const char* e = getenv("DDSP_GEMM_SMALL_DMA");
            small_dma = (e && e[0] == '0') ? 0 : 1;
        }
        // Fewer than half the CUs busy with 64x64 tiles (B = 8: 1376 rows x 256 columns = 88 tiles): what bounds a workgroup
        // there is its CU's operand load path, so 32x64 tiles on two waves - twice the workgroups, 3/4 of the bytes each - finish
        // sooner although they stage more in total.  DDSP_GEMM_HALF_TILES=0 restores 64x64 (measurement aid).
        if (small_dma && half_tiles() && dma_ok(g) && blocks(64, 64) < 128 && blocks(64, 64) > 16 && g.K >= 256) {
            dma_go<32, 64, Epi, 4, 2>(st, g, batch, epi);
            return;
        }
        if (small_dma && dma_ok(g) && blocks(64, 64) < 256 && g.K >= 256) {
            static int small_ns = -1;
            if (small_ns < 0) {
                const char* e = getenv("DDSP_GEMM_SMALL_NS");
                small_ns = e ? atoi(e) : 4;   // r02: real-time block replay 0.449 / 0.422 / 0.427 ms with 3 / 4 / 6 stages
            }
            if (small_ns == 6) dma_go<64, 64, Epi, 6, 4>(st, g, batch, epi);
            else if (small_ns == 4) dma_go<64, 64, Epi, 4, 4>(st, g, batch, epi);
            else dma_go<64, 64, Epi, 3, 4>(st, g, batch, epi);
            return;
        }
        if (dma_ok(g) && g.N <= 256 && blocks(64, 64) >= 256) {
            // skinny layers also at half the bench batch (training, B = 32: 5504 rows -> 344 tiles of 64x64)
            dma_go<64, 64, Epi, 3, 4>(st, g, batch, epi);
            return;
        }
        if (dma_ok(g) && blocks(128, 64) >= 256) {
            // 128x128 tiles (32 FLOP per staged byte, 2-stage ring, 2 workgroups per CU) when they still fill the
            // 512 resident slots; else 128x64
            // tiles with the 3-stage ring (tools/gemm_ab.py: N=1024 86 vs 78, N=256 67-72 vs 65-68, N=512 59 vs 70)
            // ... and 64x64 tiles on 4 waves for the skinny N <= 256 layers: 128-row tilings give them at most 344
            // workgroups (a third of the CUs idle or doubly loaded), 64x64 gives 688 on 768 slots
            // (tools/gemm_ab.py, M=11008: N=256 K=512 35.3 vs 38.8 us, K=768 49.2 vs 54.8 us)
            if (g.N <= 256)
                dma_go<64, 64, Epi, 3, 4>(st, g, batch, epi);
            else if (blocks(128, 128) >= t128_min()) {
                // 512 resident workgroups walk the tiles in rounds.  When the last round holds only a few tiles (QKV at
                // the bench shape: 1032 = 2 * 512 + 8) those run alone on their CUs for a whole tile time; they are
                // cut out of this launch and run as 64x64 tiles on 4 waves instead (4 small workgroups per tile).
                const int total = (int)blocks(128, 128);
                const int rem = total % 512;
                bool split = false;
                if constexpr (!epi_is_gated<Epi>::value) {
                    const bool xcd_on = g.xcd > 0 || (g.xcd < 0 && xcd_default(g.math));
                    // (XCD order: the cut tiles must come off every XCD's queue evenly and out of the leftover row blocks)
                    const int tm = (g.M + 127) / 128, tn = (g.N + 127) / 128;
                    const bool cut_ok = !xcd_on || (tail_cut() && rem % 8 == 0 && rem <= (tm % 8) * tn);
                    if (batch == 1 && rem > 0 && rem <= 64 && cut_ok) {
                        split = true;
                        Args m = g;
                        if (xcd_on) m.xcd_cut = rem;
                        dma_go<128, 128, Epi, 2>(st, m, batch, epi, total - rem);
                        Args r = g;
                        r.xcd = 0;
                        r.sub_from = total - rem;
                        r.parent_tn = (g.N + 127) / 128;
                        dma_go<64, 64, Epi, 3, 4>(st, r, 1, epi, 4 * rem);
                    }
                }
                if (!split) dma_go<128, 128, Epi, 2>(st, g, batch, epi);
            }
            else
                dma_go<128, 64, Epi, 3>(st, g, batch, epi);
            return;
        }
        if (dma_ok(g) && blocks(64, 64) >= 256) {
            // the sizes between the small-problem rule and the 128-row tilings (N > 256 at 2-4 k rows)
            dma_go<64, 64, Epi, 3, 4>(st, g, batch, epi);
            return;
        }
    }
    if ((g.B_split && g.B_split == g.B) || g.A_split) {
        // operands that exist ONLY pre-split can be read by the DMA kernel alone (a caller that also has the fp32 matrix passes
        // it as B and lands here legitimately): the caller's size predicate and this function disagree
        // nothing is launched; the caller's DDSP_LAUNCH_CHECK reports it (the outputs of this call are then undefined)
        ddsp_launch_refusal() = "pre-split GEMM operands reached the register-staged path (size predicates disagree)";
        return;
    }
    if (blocks(128, 64) >= 512)
        launch_tile<128, 64, A_KC, B_KC, A_MODE, Epi, 8>(st, g, batch, epi);
    else
        launch_tile<64, 64, A_KC, B_KC, A_MODE, Epi, 4>(st, g, batch, epi);
}

inline Args make(const float* A, int64_t lda, const float* B, int64_t ldb, int M, int N, int K) {
    Args g;
    g.A = A;
    g.B = B;
    g.M = M;
    g.N = N;
    g.K = K;
    g.lda = lda;
    g.ldb = ldb;
    g.zdiv = 1;
    g.sA_hi = g.sA_lo = g.sB_hi = g.sB_lo = 0;
    g.Fr = 1;
    g.Cin = 4;
    g.tap_shift = 0;
    g.ktaps = 3;
    g.dil = 1;
    g.in_slope = 1.f;
    g.zeros = nullptr;
    g.sub_from = 0;
    g.parent_tn = 0;
    g.math = 0;
    g.B_split = nullptr;
    g.A_split = 0;
    g.xcd_cut = 0;
    g.xcd = -1;   // decided by dma_go from the product arithmetic (see there)
    g.kz = 0;
    return g;
}

// ---- common epilogues -----------------------------------------------------------------------
struct EpiStore {  // C = acc (+ bias[n])
    float* C;
    int64_t ldc;
    const float* bias;
    int zdiv;
    int64_t sC_hi, sC_lo;
    __device__ __forceinline__ float col(int n) const { return bias ? bias[n] : 0.f; }
    __device__ __forceinline__ void operator()(int z, int m, int n, float v, float cb) const {
        float* c = C + (int64_t)(z / zdiv) * sC_hi + (int64_t)(z % zdiv) * sC_lo;
        c[(int64_t)m * ldc + n] = v + cb;
    }
    static constexpr bool kStore4 = true;
    __device__ __forceinline__ bool vec_ok() const {
        return ((uintptr_t)C % 16) == 0 && ldc % 4 == 0 && sC_hi % 4 == 0 && sC_lo % 4 == 0;
    }
    __device__ __forceinline__ void store4(int z, int m, int n, f32x4 v) const {
        float* c = C + (int64_t)(z / zdiv) * sC_hi + (int64_t)(z % zdiv) * sC_lo;
        if (bias) v += *(const f32x4_u*)(bias + n);
        *(f32x4*)(c + (int64_t)m * ldc + n) = v;
    }
};

struct EpiResidual {  // C = res + acc + bias[n]   (res may alias C)
    float* C;
    const float* res;
    int64_t ldc;
    const float* bias;
    __device__ __forceinline__ float col(int n) const { return bias[n]; }
    __device__ __forceinline__ void operator()(int, int m, int n, float v, float cb) const {
        const int64_t o = (int64_t)m * ldc + n;
        C[o] = res[o] + (v + cb);
    }
    static constexpr bool kStore4 = true;
    __device__ __forceinline__ bool vec_ok() const { return (((uintptr_t)C | (uintptr_t)res) % 16) == 0 && ldc % 4 == 0; }
    __device__ __forceinline__ void store4(int, int m, int n, f32x4 v) const {
        const int64_t o = (int64_t)m * ldc + n;
        *(f32x4*)(C + o) = *(const f32x4*)(res + o) + (v + *(const f32x4_u*)(bias + n));
    }
};

}  // namespace gemm
