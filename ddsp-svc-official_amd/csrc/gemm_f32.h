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

#include <type_traits>

namespace gemm {

enum { A_PLAIN = 0, A_CONV3 = 1 };

struct Args {
    const float* A;
    const float* B;
    int M, N, K;
    int64_t lda, ldb;
    // batching over blockIdx.z: z -> (z / zdiv, z % zdiv)
    int zdiv;
    int64_t sA_hi, sA_lo, sB_hi, sB_lo;
    // conv3 mode
    int Fr, Cin;
};

constexpr int BK = 32;
// global 16-byte loads at dword alignment (rows of a signal framed at an arbitrary length N start anywhere)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

// An epilogue that declares `static constexpr bool kPair = true` is called as epi(z, m, n, v, v_of_column_n^1).
template <class E, class = void>
struct epi_wants_pair : std::false_type {};
template <class E>
struct epi_wants_pair<E, std::void_t<decltype(E::kPair)>> : std::true_type {};

template <int BM, int BN, bool A_KC, bool B_KC, int A_MODE, class Epi>
__global__ void __launch_bounds__(256) kernel(Args g, Epi epi) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LDM = BM + 1, LDN = BN + 1;
    __shared__ float lds[2 * BK * (LDM + LDN)];
    float* const As0 = lds;                 // two [BK][LDM] buffers
    float* const Bs0 = lds + 2 * BK * LDM;  // two [BK][LDN] buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / 2);
    const int wn = (wave & 1) * (BN / 2);
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const float* __restrict__ A = g.A + (int64_t)(z / g.zdiv) * g.sA_hi + (int64_t)(z % g.zdiv) * g.sA_lo;
    const float* __restrict__ B = g.B + (int64_t)(z / g.zdiv) * g.sB_hi + (int64_t)(z % g.zdiv) * g.sB_lo;

    // staging registers: BM*BK/256/4 float4 for A, BN*BK/256/4 for B
    constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;
    f32x4 ra[NA], rb[NB];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * 256;  // float4 index in the tile
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (A_KC) {
                const int r = q / (BK / 4), kq = q % (BK / 4);
                const int m = m0 + r, k = k0 + kq * 4;
                if (m < g.M) {
                    if (A_MODE == A_CONV3) {
                        // k = tap*Cin + c ; Cin % 4 == 0 so a float4 never straddles taps
                        if (k < g.K) {
                            const int tap = k / g.Cin, c = k - tap * g.Cin;
                            const int f = m % g.Fr + tap - 1;
                            if (f >= 0 && f < g.Fr) v = *(const f32x4_u*)(A + (int64_t)(m + tap - 1) * g.lda + c);
                        }
                    } else {
                        const float* p = A + (int64_t)m * g.lda + k;
                        if (k + 3 < g.K) {
                            v = *(const f32x4_u*)p;
                        } else {
                            if (k < g.K) v[0] = p[0];
                            if (k + 1 < g.K) v[1] = p[1];
                            if (k + 2 < g.K) v[2] = p[2];
                        }
                    }
                }
            } else {
                const int kr = q / (BM / 4), mq = q % (BM / 4);
                const int k = k0 + kr, m = m0 + mq * 4;
                if (k < g.K) {
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
            const int q = tid + i * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (B_KC) {
                const int r = q / (BK / 4), kq = q % (BK / 4);
                const int n = n0 + r, k = k0 + kq * 4;
                if (n < g.N) {
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
                if (k < g.K) {
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
            const int q = tid + i * 256;
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
            const int q = tid + i * 256;
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
        const float* as = As0 + buf * (BK * LDM) + wm + lr;
        const float* bs = Bs0 + buf * (BK * LDN) + wn + lr;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(kk + lh) * LDM + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(kk + lh) * LDN + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if constexpr (epi_wants_pair<Epi>::value) {
                    // columns (2f, 2f+1) sit on adjacent lanes: hand each lane its neighbour's value too
                    const float other = __shfl_xor(acc[i][j][r], 1, 64);
                    if (m < g.M && n < g.N) epi(z, m, n, acc[i][j][r], other);
                } else {
                    if (m < g.M && n < g.N) epi(z, m, n, acc[i][j][r]);
                }
            }
        }
}

template <int BM, int BN, bool A_KC, bool B_KC, int A_MODE, class Epi>
inline void launch_tile(hipStream_t st, const Args& g, int batch, const Epi& epi) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch);
    hipLaunchKernelGGL((kernel<BM, BN, A_KC, B_KC, A_MODE, Epi>), grid, dim3(256), 0, st, g, epi);
}

// Picks the block tile so that the grid has at least ~2 blocks per CU when the problem allows it.
template <bool A_KC, bool B_KC, int A_MODE, class Epi>
inline void launch(hipStream_t st, const Args& g, int batch, const Epi& epi) {
    auto blocks = [&](int bm, int bn) {
        return (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * batch;
    };
    if (blocks(128, 128) >= 512)
        launch_tile<128, 128, A_KC, B_KC, A_MODE, Epi>(st, g, batch, epi);
    else if (blocks(64, 128) >= 384 || g.M <= 64)
        launch_tile<64, 128, A_KC, B_KC, A_MODE, Epi>(st, g, batch, epi);
    else
        launch_tile<64, 64, A_KC, B_KC, A_MODE, Epi>(st, g, batch, epi);
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
    return g;
}

// ---- common epilogues -----------------------------------------------------------------------
struct EpiStore {  // C = acc (+ bias[n])
    float* C;
    int64_t ldc;
    const float* bias;
    int zdiv;
    int64_t sC_hi, sC_lo;
    __device__ __forceinline__ void operator()(int z, int m, int n, float v) const {
        float* c = C + (int64_t)(z / zdiv) * sC_hi + (int64_t)(z % zdiv) * sC_lo;
        if (bias) v += bias[n];
        c[(int64_t)m * ldc + n] = v;
    }
};

struct EpiResidual {  // C = res + acc + bias[n]   (res may alias C)
    float* C;
    const float* res;
    int64_t ldc;
    const float* bias;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const {
        const int64_t o = (int64_t)m * ldc + n;
        C[o] = res[o] + (v + bias[n]);
    }
};

}  // namespace gemm
