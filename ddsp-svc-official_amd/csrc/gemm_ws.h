// Wave-specialised LDS-DMA GEMM for gfx950 (round 3): C = epilogue(A B^T), row-major A (M x K), nn.Linear-layout B (N x K),
// K % 32 == 0, K >= 256, 16-byte aligned rows - the Linear / 1x1-conv layers of the control network at large batches.
//
// Why a second kernel beside gemm::kernel_dma (gemm_f32.h): the round-2 ablation of that kernel at the QKV shape
// (M = 11008, N = 1536, K = 256, split-bf16 products) gave 40 us whole and ~30 us with the operand DMA, the matrix
// instructions or the epilogue stores removed - three ~10 us phases that every wave runs one after the other.  Two
// structural reasons, both removed here:
//   * `s_waitcnt vmcnt` counts a wave's loads, LDS-DMA pieces AND stores in issue order, so a wave that stores its tile
//     and then waits for its next operand pieces also waits for the stores to reach L2;
//   * every wave reads its fragments right after the barrier that publishes them, so the matrix pipe idles for an LDS
//     round trip per k-step, and all waves reach the epilogue of a tile together.
// Structure: one workgroup of 12 wavefronts per CU (persistent, walks tiles in the XCD-aware order of kernel_dma).
//   * waves 8..11 are LOADERS: they alone issue `global_load_lds_dwordx4` into an NS-stage ring (one stage = BM + BN rows
//     of 32 floats, same XOR-swizzled image as kernel_dma) and alone wait on `vmcnt`; NS - 1 stages are in flight.
//   * waves 0..7 are PRODUCT waves (4 x 2 grid, 32 x BN/2 accumulator each).  They never wait on vmcnt: their stores
//     drain in the background.  Fragments of k-step s + 1 are read from the LDS into the registers that step s has just
//     consumed, half a step at a time, BEHIND the matrix instructions of step s (the loaders publish a stage one step
//     before it is multiplied), so no product waits for the LDS.
//   * the epilogue of tile t is cut into 16-byte row pieces (one 4 x 4 lane-quad transpose each) that ride in the first
//     k-steps of tile t + 1: two accumulator sets alternate, bias vectors are fetched a tile ahead and residual operands a
//     piece ahead, so that no wait of a product wave covers a store.
// One `s_barrier` per k-step joins all 12 waves: it publishes step s + 1 (loaders -> product waves) and returns stage
// s % NS to the loaders (every product wave has its fragments of step s in registers: `lgkmcnt(0)` precedes the barrier).
// The order of the partial sums of an output element (k ascending; per 8 k-values lo*hi, hi*lo, hi*hi on
// v_mfma_f32_32x32x16_bf16) is that of kernel_dma, so the two kernels give the same bits.
#pragma once
#include "gemm_f32.h"

#include <type_traits>

namespace gemm {

// ---- epilogues of kernel_ws ------------------------------------------------------------------------------------------
// bias_ptr(): per-column constants (copied into the LDS by the loader waves, a tile at a time); kExtra: the store needs a
// (rows, ldc) operand from memory, extra_ptr() (the residual: its tile is copied into the LDS by the loader waves as well);
// emit4(z, m, n, acc + bias, extra) stores columns n..n+3 of row m.  kGated: value / gate column blocks of a
// wave (EpiGlu's packing): bias1(n) per packed column, emit4(z, m, oc, gated) stores output channels oc..oc+3.
struct WsStore {   // C = acc + bias
    float* C;
    int64_t ldc;
    const float* bias;   // may be null
    static constexpr bool kExtra = false, kGated = false;
    __device__ __forceinline__ f32x4 bias4(int n) const { return bias ? *(const f32x4_u*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ float bias1(int n) const { return bias ? bias[n] : 0.f; }
    __device__ __forceinline__ const float* extra_ptr() const { return nullptr; }
    __device__ __forceinline__ void emit4(int, int m, int n, f32x4 v, f32x4) const { *(f32x4*)(C + (int64_t)m * ldc + n) = v; }
    __device__ __forceinline__ void emit1(int, int m, int n, float v) const { C[(int64_t)m * ldc + n] = v; }
    __host__ __device__ const float* bias_ptr() const { return bias; }
    __host__ bool vec_ok() const { return ((uintptr_t)C % 16) == 0 && ldc % 4 == 0 && ((uintptr_t)bias % 16) == 0; }
};
struct WsResidual {   // C = res + (acc + bias)   (res may alias C)
    float* C;
    const float* res;
    int64_t ldc;
    const float* bias;
    static constexpr bool kExtra = true, kGated = false;
    __device__ __forceinline__ f32x4 bias4(int n) const { return *(const f32x4_u*)(bias + n); }
    __device__ __forceinline__ float bias1(int n) const { return bias[n]; }
    __device__ __forceinline__ const float* extra_ptr() const { return res; }   // (rows, ldc): fetched by the loader waves
    __device__ __forceinline__ void emit4(int, int m, int n, f32x4 v, f32x4 e) const { *(f32x4*)(C + (int64_t)m * ldc + n) = e + v; }
    __device__ __forceinline__ void emit1(int, int m, int n, float v) const {
        const int64_t o = (int64_t)m * ldc + n;
        C[o] = res[o] + v;
    }
    __host__ __device__ const float* bias_ptr() const { return bias; }
    __host__ bool vec_ok() const { return (((uintptr_t)C | (uintptr_t)res | (uintptr_t)bias) % 16) == 0 && ldc % 4 == 0; }
};
struct WsSplit3 {   // column block n / 512 selects the destination matrix (q, k or v), each (rows, 512)
    float* out[3];
    const float* bias;
    static constexpr bool kExtra = false, kGated = false;
    __device__ __forceinline__ f32x4 bias4(int n) const { return *(const f32x4_u*)(bias + n); }
    __device__ __forceinline__ float bias1(int n) const { return bias[n]; }
    __device__ __forceinline__ const float* extra_ptr() const { return nullptr; }
    __device__ __forceinline__ void emit4(int, int m, int n, f32x4 v, f32x4) const {
        *(f32x4*)(out[n >> 9] + (int64_t)m * 512 + (n & 511)) = v;
    }
    __device__ __forceinline__ void emit1(int, int m, int n, float v) const { out[n >> 9][(int64_t)m * 512 + (n & 511)] = v; }
    __host__ __device__ const float* bias_ptr() const { return bias; }
    __host__ bool vec_ok() const { return (((uintptr_t)out[0] | (uintptr_t)out[1] | (uintptr_t)out[2] | (uintptr_t)bias) % 16) == 0; }
};
struct WsGlu {   // out[m][c] = (a + bias_a) * sigmoid(g + bias_g): packed columns 64t..64t+31 values, 64t+32..64t+63 gates
    float* out;          // (rows, ldo)
    int64_t ldo;
    const float* bias;   // packed like the weight rows
    static constexpr bool kExtra = false, kGated = true;
    __device__ __forceinline__ f32x4 bias4(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ float bias1(int n) const { return bias[n]; }
    __device__ __forceinline__ const float* extra_ptr() const { return nullptr; }
    __device__ __forceinline__ void emit4(int, int m, int n, f32x4 v, f32x4) const { *(f32x4*)(out + (int64_t)m * ldo + n) = v; }
    __device__ __forceinline__ void emit1(int, int, int, float) const {}
    __host__ __device__ const float* bias_ptr() const { return bias; }
    __host__ bool vec_ok() const { return (((uintptr_t)out | (uintptr_t)bias) % 16) == 0 && ldo % 4 == 0; }
};

template <int V>
using ic = std::integral_constant<int, V>;
// f(ic<I>{}) for I = I0 .. N-1 with the index a compile-time constant (accumulator registers must be indexed statically)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(ic<I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int WS_PRODUCT_WAVES = 8, WS_LOADER_WAVES = 4, WS_THREADS = 64 * (WS_PRODUCT_WAVES + WS_LOADER_WAVES);

// MATH: 0 fp32 MFMA (32x32x2), 3 split-bf16 with both operands split in the loop, 7 B pre-split, 8 both pre-split.
// VEC: the epilogue's rows are 16-byte aligned and N % BN == 0 (checked by the launcher); else dword stores per element.
// ABLATE (timing experiments only): 1 no matrix instructions, 2 no DMA, 4 no epilogue stores, 8 product waves only join the
// barriers, 16 no lgkmcnt(0) before the barrier (WRONG results), 32 no epilogue pieces at all, 64 loaders issue every other piece.
// A_MODE: A_PLAIN, or A_CONVK (implicit im2col of a "same" convolution with Args::ktaps taps and dilation Args::dil, k = tap *
// Cin + c, Cin % 32 == 0: the loader waves move an A row's source by the tap's frame offset, or to the zero page when that frame
// lies outside the utterance - as kernel_dma does).
template <int BM, int BN, class Epi, int NS, int MATH, bool VEC, int ABLATE = 0, int A_MODE = A_PLAIN>
__global__ void __launch_bounds__(WS_THREADS) kernel_ws(Args g, Epi epi, int tiles_m, int tiles_n, int total_tiles) {
    constexpr int TM = BM / 128, TN = BN / 64;            // 32x32 accumulator tiles per product wave (4 x 2 wave grid)
    static_assert(BM % 128 == 0 && BN % 64 == 0 && TM >= 1 && TN >= 1, "tile must split over the 4 x 2 product waves");
    constexpr int ROWS = BM + BN, STAGE = ROWS * 32;      // floats per stage
    constexpr int PIECES = ROWS / 8;                       // 1-KiB DMA pieces per stage
    static_assert(PIECES % WS_LOADER_WAVES == 0, "stage must split evenly over the loader waves");
    constexpr int PPL = PIECES / WS_LOADER_WAVES;          // pieces per loader wave and stage
    static_assert((NS - 1) * PPL <= 63, "vmcnt is a 6-bit counter");
    static_assert(NS >= 3, "ring: one stage being read, at least two in flight");
    constexpr bool GATED = Epi::kGated;
    static_assert(!GATED || TN == 2, "gated-pair epilogue needs the wave's two column blocks");
    // epilogue pieces of a wave's accumulator (4 registers = 4 rows x 1 column per lane -> one 16-byte row piece per lane)
    constexpr int NG = GATED ? TM * 4 : TM * TN * 4;
    constexpr int NPS = 8;                                  // k-steps of a tile that carry pieces (K >= 256)
    constexpr int PPS = (NG + NPS - 1) / NPS;              // pieces per such step
    // LDS: the ring | two tiles' bias vectors (2 x 256 floats) | kExtra: one tile of the residual operand, 16-byte chunk c of
    // row r at chunk (c + 4 (r & 3)) mod (BN / 4) of its row (the epilogue's reads are then conflict-free)
    constexpr int BIAS_AT = NS * STAGE, RES_AT = BIAS_AT + 512;
    constexpr int RES_PIECES = Epi::kExtra ? BM * BN / 256 : 0, RPL = RES_PIECES / WS_LOADER_WAVES;   // 1-KiB pieces of that tile
    constexpr int RES_RPP = 256 / BN;                       // rows per piece
    static_assert(!Epi::kExtra || (BN == 64 && RES_PIECES % WS_LOADER_WAVES == 0), "residual tile: 64 columns");
    extern __shared__ __attribute__((aligned(1024))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.K / 32;

    // persistent workgroup: tile enumeration of kernel_dma (XCD x owns row blocks x, x + 8, ... with all their column tiles)
    const bool xcd_order = g.xcd && (gridDim.x & 7) == 0 && total_tiles == tiles_m * tiles_n;
    const int xq0 = (int)blockIdx.x >> 3, xgs = (int)gridDim.x >> 3, xx = (int)blockIdx.x & 7;
    const int xown = (tiles_m / 8) * tiles_n, xleft = (tiles_m % 8) * tiles_n;
    int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (xcd_order) {
        const int qx = xown + (xleft > xx ? (xleft - xx + 7) / 8 : 0);
        my_tiles = xq0 < qx ? (qx - xq0 + xgs - 1) / xgs : 0;
    }
    auto tile_of = [&](int i) -> int {
        if (!xcd_order) return (int)blockIdx.x + i * (int)gridDim.x;
        const int q = xq0 + i * xgs;
        if (q < xown) return (xx + 8 * (q / tiles_n)) * tiles_n + q % tiles_n;
        const int u = (q - xown) * 8 + xx;
        return ((tiles_m / 8) * 8 + u / tiles_n) * tiles_n + u % tiles_n;
    };
    const int steps = my_tiles * nk;
    const int per_z = tiles_m * tiles_n;

    if (wave >= WS_PRODUCT_WAVES) {
        // ================================ loader waves ================================
        const int lw = wave - WS_PRODUCT_WAVES;
        const float* src[PPL];
        constexpr int PA = BM / 8 / WS_LOADER_WAVES;   // this wave's pieces i < PA are A rows
        int conv_f[PA];                                // A_CONVK: frame (inside its utterance) of each A row this lane fetches
        auto tile_src = [&](int tile) {
            const int z = tile / per_z, rem = tile - z * per_z;
            const int m0 = (rem / tiles_n) * BM, n0 = (rem % tiles_n) * BN;
            const float* A = g.A + (int64_t)(z / g.zdiv) * g.sA_hi + (int64_t)(z % g.zdiv) * g.sA_lo;
            const float* B = g.B + (int64_t)(z / g.zdiv) * g.sB_hi + (int64_t)(z % g.zdiv) * g.sB_lo;
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                const int piece = lw + WS_LOADER_WAVES * i;
                const int row = piece * 8 + (lane >> 3);
                const int slot = (lane & 7) ^ ((row >> 1) & 7);
                if (row < BM) {
                    int m = m0 + row;
                    m = m < g.M ? m : g.M - 1;     // rows past the edge re-read the last row (never stored)
                    src[i] = A + (int64_t)m * g.lda + slot * 4;
                    if constexpr (A_MODE == A_CONVK) conv_f[i < PA ? i : 0] = m % g.Fr;
                } else {
                    int n = n0 + row - BM;
                    n = n < g.N ? n : g.N - 1;
                    src[i] = B + (int64_t)n * g.ldb + slot * 4;
                }
            }
        };
        int issued = 0, ikt = 0, itile = 0, istage = 0;
        if (steps > 0) tile_src(tile_of(0));
        // The bias of a tile's columns travels with the tile's first k-step (loader wave 0, one extra piece into a region
        // behind the ring, two tiles deep): the product waves then read it from the LDS and issue no vector-memory load of
        // their own, so no `vmcnt` wait of theirs can cover a store.  (The extra piece makes the counted waits below
        // conservative - they wait for at most one piece more than they must.)
        const bool bias_dma = VEC && lw == 0 && epi.bias_ptr() != nullptr;
        auto issue_next = [&]() {
            if (issued >= steps) return;
            if (bias_dma && ikt == 0) {
                const int tile = tile_of(itile);
                const int n0 = ((tile % per_z) % tiles_n) * BN;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(epi.bias_ptr() + n0 + 4 * (lane % (BN / 4))),
                                                 (__attribute__((address_space(3))) void*)(lds + BIAS_AT + (itile & 1) * 256),
                                                 16, 0, 0);
            }
            if (!(ABLATE & 2)) {
                int koff = ikt * 32, conv_off = 0;
                if constexpr (A_MODE == A_CONVK) {
                    const int tap = koff / g.Cin;
                    koff -= tap * g.Cin;
                    conv_off = (tap - (g.ktaps - 1) / 2) * g.dil;
                }
#pragma unroll
                for (int i = 0; i < PPL; i += ((ABLATE & 64) ? 2 : 1)) {
                    const int piece = lw + WS_LOADER_WAVES * i;
                    const float* p = src[i] + ikt * 32;
                    if constexpr (A_MODE == A_CONVK) {
                        if (i < PA) {
                            const int f = conv_f[i < PA ? i : 0] + conv_off;
                            p = (f >= 0 && f < g.Fr) ? src[i] + (int64_t)conv_off * g.lda + koff : g.zeros + (lane & 7) * 4;
                        }
                    }
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                     (__attribute__((address_space(3))) void*)(lds + istage * STAGE + piece * 256),
                                                     16, 0, 0);
                }
            }
            if constexpr (Epi::kExtra && VEC) {
                // The residual tile of tile `itile` travels behind the pieces of its LAST k-step: it is complete at the
                // barrier that publishes the step after next (or at the last barrier), before the first epilogue piece of
                // that tile is formed; the region is free again because K >= 32 (NS + 8) (checked by the launcher).
                if (ikt == nk - 1) {
                    const int tile = tile_of(itile);
                    const int rem = tile % per_z;
                    const int m0 = (rem / tiles_n) * BM, n0 = (rem % tiles_n) * BN;
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        const int piece = lw + WS_LOADER_WAVES * i;
                        const int r = piece * RES_RPP + lane / (BN / 4), cp = lane % (BN / 4);
                        const int c = (cp - 4 * (r & 3)) & (BN / 4 - 1);
                        int m = m0 + r;
                        m = m < g.M ? m : g.M - 1;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(epi.extra_ptr() + (int64_t)m * epi.ldc + n0 + 4 * c),
                                                         (__attribute__((address_space(3))) void*)(lds + RES_AT + piece * 256),
                                                         16, 0, 0);
                    }
                }
            }
            ++issued;
            if (++istage == NS) istage = 0;
            if (++ikt == nk) {
                ikt = 0;
                ++itile;
                if (itile < my_tiles) tile_src(tile_of(itile));
            }
        };
#pragma unroll
        for (int i = 0; i < NS; ++i) issue_next();
        // B_0: step 0 has landed
        if (steps >= NS)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PPL) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < steps; ++s) {
            // B_{s+1}: step s + 1 has landed (outstanding: steps s + 2 .. s + NS - 1) and stage s % NS is free again
            if (s + NS - 1 < steps)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPL) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (!(ABLATE & 256)) __builtin_amdgcn_s_barrier();
            issue_next();   // step s + NS into stage s % NS
        }
        return;
    }

    // ================================ product waves ================================
    const int wm = (wave >> 1) * (BM / 4), wn = (wave & 1) * (BN / 2);
    const int lr = lane & 31, lh = lane >> 5, q = lane & 3;
    // fragment addresses: lane (i = lr, h = lh) reads its 16 k-values 16h .. 16h+15 of row i as four 16-byte slots
    // (4h + c) ^ swizzle(row); rows of one wave differ by multiples of 32, which leave the swizzle alone
    int offc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) offc[c] = lr * 32 + 4 * ((4 * lh + c) ^ ((lr >> 1) & 7));
    const int offA = wm * 32, offB = (BM + wn) * 32;

    f32x16 acc[2][TM][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;
    f32x4 fa[2][TM][2], fb[2][TN][2];   // [half][tile][hi | lo slot pair]
    f32x4 bvec[2][TN];                  // bias of the lane's 4 columns, per accumulator set
    float bgate[2][2];                  // gated epilogue: bias of the lane's value / gate column
    f32x4 ex[2];                        // residual operand of the current / next piece
    int pm0 = 0, pn0 = 0, pz = 0;       // tile whose accumulator set awaits its epilogue
    bool pend = false;

    auto read_half = [&](const float* st, auto H) {
        constexpr int h = decltype(H)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            fa[h][i][0] = *(const f32x4*)(st + offA + i * 1024 + offc[2 * h]);
            fa[h][i][1] = *(const f32x4*)(st + offA + i * 1024 + offc[2 * h + 1]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            fb[h][j][0] = *(const f32x4*)(st + offB + j * 1024 + offc[2 * h]);
            fb[h][j][1] = *(const f32x4*)(st + offB + j * 1024 + offc[2 * h + 1]);
        }
    };
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x2p __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
    auto split = [](const f32x4& x0, const f32x4& x1, bf16x8 (&p)[2]) {   // kernel_dma's split: same bits
        u32x4p w[2];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x2p r = d < 2 ? (f32x2p){x0[2 * d], x0[2 * d + 1]} : (f32x2p){x1[2 * d - 4], x1[2 * d - 3]};
            const uint32_t hh = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2p));
            w[0][d] = hh;
            r = r - (f32x2p){__builtin_bit_cast(float, hh << 16), __builtin_bit_cast(float, hh & 0xffff0000u)};
            w[1][d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2p));
        }
        p[0] = __builtin_bit_cast(bf16x8, w[0]);
        p[1] = __builtin_bit_cast(bf16x8, w[1]);
    };
    auto mfma_half = [&](auto CUR, auto H) {
        constexpr int cur = decltype(CUR)::value, h = decltype(H)::value;
        if constexpr ((ABLATE & 1) != 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[cur][i][j][0] += fa[h][i][0][0] + fb[h][j][1][3];
        } else if constexpr (MATH == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[cur][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i][s >> 2][s & 3], fb[h][j][s >> 2][s & 3],
                                                                              acc[cur][i][j], 0, 0, 0);
        } else {
            bf16x8 ap[TM][2], bp[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (MATH == 8) {
                    ap[i][0] = __builtin_bit_cast(bf16x8, fa[h][i][0]);
                    ap[i][1] = __builtin_bit_cast(bf16x8, fa[h][i][1]);
                } else {
                    split(fa[h][i][0], fa[h][i][1], ap[i]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (MATH == 7 || MATH == 8) {
                    bp[j][0] = __builtin_bit_cast(bf16x8, fb[h][j][0]);
                    bp[j][1] = __builtin_bit_cast(bf16x8, fb[h][j][1]);
                } else {
                    split(fb[h][j][0], fb[h][j][1], bp[j]);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[cur][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[j][0], acc[cur][i][j], 0, 0, 0);
                    acc[cur][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[j][1], acc[cur][i][j], 0, 0, 0);
                    acc[cur][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[j][0], acc[cur][i][j], 0, 0, 0);
                }
        }
    };
    // rows / columns of piece G of the pending tile for this lane (after the quad transpose lane q of a quad owns row q)
    auto piece_m = [&](int G) { return pm0 + wm + 32 * (GATED ? G / 4 : G / (TN * 4)) + q + 8 * (G & 3) + 4 * lh; };
    auto piece_n = [&](int G) { return pn0 + wn + 32 * (GATED ? 0 : (G / 4) % TN) + (lr & ~3); };
    auto res_read = [&](int G) -> f32x4 {   // the lane's 16 bytes of piece G in the LDS residual tile
        const int r = wm + 32 * (G / (TN * 4)) + q + 8 * (G & 3) + 4 * lh;
        const int c = (wn + 32 * ((G / 4) % TN) + (lr & ~3)) >> 2;
        return *(const f32x4*)(lds + RES_AT + r * BN + 4 * ((c + 4 * (r & 3)) & (BN / 4 - 1)));
    };
    auto emit_piece = [&](auto PREV, auto GI, f32x4 extra) {
        constexpr int prev = decltype(PREV)::value, G = decltype(GI)::value;
        constexpr int rg = G & 3;
        const int m = piece_m(G);
        if constexpr (GATED) {
            constexpr int i = G / 4;
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = acc[prev][i][0][4 * rg + r] + bgate[prev][0], gt = acc[prev][i][1][4 * rg + r] + bgate[prev][1];
                // split-bf16 products (values carry ~4e-6): sigmoid by v_exp_f32 / v_rcp_f32 (~2 ulp) instead of expf and an IEEE
                // division - 0.44 of this launch's vector issue was the epilogue; fp32 products keep the exact form
                if constexpr (MATH != 0)
                    x[r] = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * gt));
                else
                    x[r] = a * (1.0f / (1.0f + expf(-gt)));
                acc[prev][i][0][4 * rg + r] = 0.f;
                acc[prev][i][1][4 * rg + r] = 0.f;
            }
            quad_transpose(x, q);
            const int oc = ((pn0 + wn) >> 1) + (lr & ~3);
            if (!(ABLATE & 4) || x[0] == 12345.678f)
                if (m < g.M) epi.emit4(pz, m, oc, f32x4{x[0], x[1], x[2], x[3]}, extra);
        } else {
            constexpr int i = G / (TN * 4), j = (G / 4) % TN;
            if constexpr (VEC) {
                float x[4] = {acc[prev][i][j][4 * rg], acc[prev][i][j][4 * rg + 1], acc[prev][i][j][4 * rg + 2],
                              acc[prev][i][j][4 * rg + 3]};
                quad_transpose(x, q);
                const f32x4 v = f32x4{x[0], x[1], x[2], x[3]} + bvec[prev][j];
                if (!(ABLATE & 4) || x[0] == 12345.678f)
                    if (m < g.M) epi.emit4(pz, m, piece_n(G), v, extra);
            } else {
                // edge tiles / unaligned outputs: one dword per accumulator register (column = lane, rows in the registers)
                const int n = pn0 + wn + 32 * j + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mr = pm0 + wm + 32 * i + r + 8 * rg + 4 * lh;
                    if (n < g.N && mr < g.M) epi.emit1(pz, mr, n, acc[prev][i][j][4 * rg + r] + bvec[prev][j][0]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[prev][i][j][4 * rg + r] = 0.f;
        }
    };
    const bool has_bias = epi.bias_ptr() != nullptr;
    auto load_bias = [&](auto CUR, int n0, int ti) {   // from the LDS copy the loaders made (its tile has landed)
        constexpr int cur = decltype(CUR)::value;
        const float* bl = lds + BIAS_AT + (ti & 1) * 256;
        if constexpr (GATED) {
            bgate[cur][0] = bl[wn + lr];
            bgate[cur][1] = bl[wn + lr + 32];
        } else if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bvec[cur][j] = has_bias ? *(const f32x4*)(bl + wn + 32 * j + (lr & ~3)) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn + 32 * j + lr;
                bvec[cur][j][0] = n < g.N ? epi.bias1(n) : 0.f;
            }
        }
    };

    int rd_stage = 0;   // stage of the step whose fragments are read next
    auto next_stage = [&]() -> const float* {
        const float* st = lds + rd_stage * STAGE;
        if (++rd_stage == NS) rd_stage = 0;
        return st;
    };
    // one k-step of tile set CUR; PS >= 0: it also carries pieces PS*PPS .. of the pending set
    auto step = [&](auto CUR, auto PSI) {
        constexpr int cur = decltype(CUR)::value, PS = decltype(PSI)::value;
        if constexpr (!(ABLATE & 16)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments of this step are in registers ...
        if constexpr (!(ABLATE & 256)) __builtin_amdgcn_s_barrier();   // ... the next step has landed, this step's stage is free
        if constexpr ((ABLATE & 8) != 0) return;
        const float* st = next_stage();
        if constexpr (PS >= 0 && Epi::kExtra && VEC) {
            // residual operands of this step's pieces, from the LDS copy of the pending tile
            static_for<0, PPS>([&](auto P) {
                constexpr int G = PS * PPS + decltype(P)::value;
                if constexpr (G < NG) ex[G & 1] = res_read(G);
            });
        }
        mfma_half(CUR, ic<0>{});
        if constexpr (!(ABLATE & 128)) read_half(st, ic<0>{});
        mfma_half(CUR, ic<1>{});
        if constexpr (!(ABLATE & 128)) read_half(st, ic<1>{});
        if constexpr (PS >= 0 && !(ABLATE & 32)) {
            if (pend) {
                static_for<0, PPS>([&](auto P) {
                    constexpr int G = PS * PPS + decltype(P)::value;
                    if constexpr (G < NG) emit_piece(ic<cur ^ 1>{}, ic<G>{}, ex[G & 1]);
                });
            }
        }
    };
    auto run_tile = [&](auto CUR, int ti) {
        const int tile = tile_of(ti);
        const int z = tile / per_z, rem = tile - z * per_z;
        const int m0 = (rem / tiles_n) * BM, n0 = (rem % tiles_n) * BN;
        load_bias(CUR, n0, ti);
        step(CUR, ic<0>{});
        step(CUR, ic<1>{});
        step(CUR, ic<2>{});
        step(CUR, ic<3>{});
        step(CUR, ic<4>{});
        step(CUR, ic<5>{});
        step(CUR, ic<6>{});
        step(CUR, ic<7>{});
        for (int kt = NPS; kt < nk; ++kt) step(CUR, ic<-1>{});
        pend = true;
        pm0 = m0;
        pn0 = n0;
        pz = z;
    };

    __builtin_amdgcn_s_barrier();   // B_0
    {
        const float* st = next_stage();
        read_half(st, ic<0>{});
        read_half(st, ic<1>{});
    }
    for (int ti = 0; ti < my_tiles; ti += 2) {
        run_tile(ic<0>{}, ti);
        if (ti + 1 < my_tiles) run_tile(ic<1>{}, ti + 1);
    }
    // the last tile's epilogue (set 0 when the tile count is odd)
    if (pend) {
        auto flush = [&](auto PREV) {
            static_for<0, NG>([&](auto GI) {
                constexpr int G = decltype(GI)::value;
                f32x4 e = {0.f, 0.f, 0.f, 0.f};
                if constexpr (Epi::kExtra && VEC) e = res_read(G);
                emit_piece(PREV, GI, e);
            });
        };
        if (my_tiles & 1)
            flush(ic<0>{});
        else
            flush(ic<1>{});
    }
}

template <int BM, int BN, class Epi, int NS, int MATH, bool VEC, int ABLATE = 0, int A_MODE = A_PLAIN>
inline hipError_t launch_ws_one(hipStream_t st, const Args& g, const Epi& epi) {
    constexpr size_t lds_bytes = (size_t)NS * (BM + BN) * 128 + 2048 + (Epi::kExtra ? (size_t)BM * BN * 4 : 0);   // ring + bias + residual tile
    static_assert(lds_bytes <= 160 * 1024, "ring does not fit the LDS");
    static_assert(lds_bytes > 80 * 1024, "one workgroup per CU is assumed");
    auto kfn = kernel_ws<BM, BN, Epi, NS, MATH, VEC, ABLATE, A_MODE>;
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int total = tiles_m * tiles_n;
    if (total == 0) return hipSuccess;
    const int grid = total < 256 ? total : 256;   // one persistent workgroup per CU
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WS_THREADS), lds_bytes, st, g, epi, tiles_m, tiles_n, total);
    return hipSuccess;
}

// what kernel_ws handles (its callers fall back to gemm::launch otherwise)
inline bool ws_ok(const Args& g, int ns = 4, bool extra = false) {
    return dma_ok(g) && g.K >= 256 && (!extra || g.K >= 32 * (ns + 8)) && g.zdiv == 1 && g.sub_from == 0 && g.kz == 0;
}

// The callers' entry: split-bf16 products with pre-split weights (Args::B_split; Args::A_split picks mode 8 or 7), whole
// tiles in N and a 16-byte aligned epilogue - what the control network's large-batch Linear layers have (the callers check
// ws_ok(), vec_ok() and N % BN == 0 and otherwise stay on gemm::launch).
template <int BM, int BN, class Epi, int NS>
inline hipError_t ws_go(hipStream_t st, const Args& g0, const Epi& epi) {
    Args g = g0;
    if (g.xcd < 0) g.xcd = 1;
    if (!(g.math == 3 && g.B_split && epi.vec_ok() && g.N % BN == 0)) return hipErrorInvalidValue;
    g.B = g.B_split;
    return g.A_split ? launch_ws_one<BM, BN, Epi, NS, 8, true>(st, g, epi) : launch_ws_one<BM, BN, Epi, NS, 7, true>(st, g, epi);
}

// the same for a "same" convolution as an implicit-im2col product (A_CONVK: Args::Fr / Cin / ktaps / dil / zeros set by the caller,
// Cin % 32 == 0); pre-split weights, activations pre-split (mode 8) or fp32 (mode 7)
template <int BM, int BN, class Epi, int NS>
inline hipError_t ws_conv_go(hipStream_t st, const Args& g0, const Epi& epi) {
    Args g = g0;
    if (g.xcd < 0) g.xcd = 1;
    if (!(g.math == 3 && g.B_split && g.zeros && g.Cin % 32 == 0 && epi.vec_ok() && g.N % BN == 0)) return hipErrorInvalidValue;
    g.B = g.B_split;
    return g.A_split ? launch_ws_one<BM, BN, Epi, NS, 8, true, 0, A_CONVK>(st, g, epi)
                     : launch_ws_one<BM, BN, Epi, NS, 7, true, 0, A_CONVK>(st, g, epi);
}

}  // namespace gemm
