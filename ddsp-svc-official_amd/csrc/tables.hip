// Constant DFT tables, generated on the device in fp64 (integer argument reduction) and cached in the ctx.
#include "tables.h"

namespace {

constexpr double kTwoPi = 6.283185307179586476925286766559;

// Inverse real DFT, n = 2*(M-1) taps, rotated by n/2 (the reference's `roll(ir, n//2)`, ddsp/core.py:326,300,287):
//   ir[k] = sum_f Re X_f * T[f][k] + Im X_f * T[M+f][k]
//   T[f][k]   =  c_f/n * cos(2*pi*f*(k-n/2)/n),  c_0 = c_{M-1} = 1, else 2
//   T[M+f][k] = -c_f/n * sin(2*pi*f*(k-n/2)/n),  zero for f in {0, M-1} (C2R ignores those imaginary parts)
// kind TAB_IRDFT_RE      : rows [0, M)            (real responses, no window)
// kind TAB_IRDFT_RE_HANN : rows [0, M) times hann_periodic(n)[k]   (ddsp/core.py:262,272,276,287 folded)
// kind TAB_IRDFT_CPLX    : rows [0, 2M)
// n1 = 1 (key of the cache): the same table TAP-MAJOR, [n][rows], for the LDS-DMA GEMM of the forward filter synthesis
// (both operands k-contiguous); the frequency-major layout stays for the backward pass.
__global__ void irdft_table_kernel(float* __restrict__ tab, int M, int n, int ld, int rows, int hann, int tap_major) {
    const int64_t total = tap_major ? (int64_t)n * rows : (int64_t)rows * ld;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = tap_major ? (int)(idx % rows) : (int)(idx / ld);
        const int k = tap_major ? (int)(idx / rows) : (int)(idx % ld);
        if (k >= n) {
            tab[idx] = 0.f;
            continue;
        }
        const int f = row < M ? row : row - M;
        const double cf = (f == 0 || f == M - 1) ? 1.0 : 2.0;
        int64_t r = ((int64_t)f * (k - n / 2)) % n;
        if (r < 0) r += n;
        const double ang = kTwoPi * (double)r / (double)n;
        double v;
        if (row < M)
            v = cf / n * cos(ang);
        else
            v = (f == 0 || f == M - 1) ? 0.0 : -cf / n * sin(ang);
        if (hann) v *= 0.5 - 0.5 * cos(kTwoPi * (double)k / (double)n);
        tab[idx] = (float)v;
    }
}

// Forward / inverse real DFT of length N = n0 for the circular windowed OLA of CombSubFast
// (ddsp/vocoder.py:434,463-486), window sqrt(hann_periodic(N)) folded in on both sides:
// kind TAB_RDFT_FWD_W : [N][ld >= 2*(N/2+1)]  x[i] -> (Re X_f | Im X_f):  w[i]*cos(2 pi f i/N) | -w[i]*sin(...)
// kind TAB_RDFT_INV_W : [2*(N/2+1)][N]  (Re | Im) -> w[k] * irfft
__global__ void rdft_w_table_kernel(float* __restrict__ tab, int N, int ld, int inverse) {
    const int Mb = N / 2 + 1;
    const int64_t total = inverse ? (int64_t)N * 2 * Mb : (int64_t)N * ld;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int i, col;
        if (!inverse) {
            i = (int)(idx / ld);
            col = (int)(idx % ld);
            if (col >= 2 * Mb) {
                tab[idx] = 0.f;
                continue;
            }
        } else {
            col = (int)(idx / N);
            i = (int)(idx % N);
        }
        const int f = col < Mb ? col : col - Mb;
        const bool im = col >= Mb;
        const int64_t r = ((int64_t)f * i) % N;
        const double ang = kTwoPi * (double)r / (double)N;
        const double w = sqrt(0.5 - 0.5 * cos(kTwoPi * (double)i / (double)N));
        double v;
        if (!inverse) {
            v = im ? -sin(ang) : cos(ang);
        } else {
            const double cf = (f == 0 || f == N / 2) ? 1.0 : 2.0;
            v = im ? ((f == 0 || f == N / 2) ? 0.0 : -cf / N * sin(ang)) : cf / N * cos(ang);
        }
        tab[idx] = (float)(v * w);
    }
}

// dst = src with every group of 8 consecutive floats replaced by 8 bf16 hi parts (16 bytes) and 8 bf16 lo parts (16 bytes):
// the B operand layout of gemm::Args::B_split.  Same rounding as the in-kernel split (round to nearest even, twice).
__global__ void presplit_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst, int64_t groups) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * blockDim.x) {
        const float* s = src + 8 * gidx;
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float a = s[2 * d], b = s[2 * d + 1];
            hi[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){a, b}, bf16x2v));
            const float ha = __builtin_bit_cast(float, hi[d] << 16), hb = __builtin_bit_cast(float, hi[d] & 0xffff0000u);
            lo[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){a - ha, b - hb}, bf16x2v));
        }
        uint32_t* o = dst + 8 * gidx;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            o[d] = hi[d];
            o[4 + d] = lo[d];
        }
    }
}

}  // namespace

int ddsp_get_table(ddsp_ctx* ctx, hipStream_t st, int kind, int n0, int n1, float** out) {
    for (int i = 0; i < ctx->n_tables; ++i) {
        ddsp_table& t = ctx->tables[i];
        if (t.kind == kind && t.n0 == n0 && t.n1 == n1) {
            *out = t.dev;
            return DDSP_OK;
        }
    }
    if (ctx->n_tables >= 64) return ddsp_fail(ctx, DDSP_ERR_OOM, "table cache full", "");
    if (n1 == 2) {
        // the tap-major table (n1 == 1) split into bf16 hi/lo groups: B operand of the split-bf16 GEMM (gemm::Args::B_split)
        float* plain = nullptr;
        int rc = ddsp_get_table(ctx, st, kind, n0, 1, &plain);
        if (rc) return rc;
        if (ctx->n_tables >= 64) return ddsp_fail(ctx, DDSP_ERR_OOM, "table cache full", "");
        size_t bytes = 0;
        for (int i = 0; i < ctx->n_tables; ++i)
            if (ctx->tables[i].dev == plain) bytes = ctx->tables[i].bytes;
        if (bytes == 0 || bytes % 32 != 0) return ddsp_fail(ctx, DDSP_ERR_ARG, "table cannot be pre-split", "");
        float* dev = nullptr;
        hipError_t e = hipMalloc((void**)&dev, bytes);
        if (e != hipSuccess) return ddsp_fail(ctx, DDSP_ERR_OOM, "table hipMalloc", hipGetErrorString(e));
        const int64_t groups = (int64_t)(bytes / 32);
        hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((groups + 255) / 256 > 2048 ? 2048 : (groups + 255) / 256)), dim3(256), 0,
                           st, plain, (uint32_t*)dev, groups);
        DDSP_LAUNCH_CHECK(ctx);
        ddsp_table& t = ctx->tables[ctx->n_tables++];
        t.kind = kind;
        t.n0 = n0;
        t.n1 = n1;
        t.dev = dev;
        t.bytes = bytes;
        *out = dev;
        return DDSP_OK;
    }
    size_t elems = 0;
    switch (kind) {
        case TAB_IRDFT_RE:
        case TAB_IRDFT_RE_HANN:
            elems = n1 ? (size_t)n0 * 2 * (n0 - 1) : (size_t)n0 * ddsp_pad4(2 * (n0 - 1));
            break;
        case TAB_IRDFT_CPLX:
            elems = n1 ? (size_t)2 * n0 * 2 * (n0 - 1) : (size_t)2 * n0 * ddsp_pad4(2 * (n0 - 1));
            break;
        case TAB_RDFT_FWD_W:
            elems = (size_t)n0 * ddsp_pad4(2 * (n0 / 2 + 1));
            break;
        case TAB_RDFT_INV_W:
            elems = (size_t)n0 * 2 * (n0 / 2 + 1);
            break;
        default:
            return ddsp_fail(ctx, DDSP_ERR_ARG, "unknown table kind", "");
    }
    float* dev = nullptr;
    hipError_t e = hipMalloc((void**)&dev, elems * sizeof(float));
    if (e != hipSuccess) return ddsp_fail(ctx, DDSP_ERR_OOM, "table hipMalloc", hipGetErrorString(e));
    const unsigned blocks = (unsigned)((elems + 255) / 256 > 2048 ? 2048 : (elems + 255) / 256);
    if (kind == TAB_IRDFT_RE || kind == TAB_IRDFT_RE_HANN || kind == TAB_IRDFT_CPLX) {
        const int M = n0, n = 2 * (n0 - 1);
        const int rows = kind == TAB_IRDFT_CPLX ? 2 * M : M;
        hipLaunchKernelGGL(irdft_table_kernel, dim3(blocks), dim3(256), 0, st, dev, M, n, ddsp_pad4(n), rows,
                           kind == TAB_IRDFT_RE_HANN ? 1 : 0, n1 ? 1 : 0);
    } else {
        hipLaunchKernelGGL(rdft_w_table_kernel, dim3(blocks), dim3(256), 0, st, dev, n0,
                           ddsp_pad4(2 * (n0 / 2 + 1)), kind == TAB_RDFT_INV_W ? 1 : 0);
    }
    DDSP_LAUNCH_CHECK(ctx);
    ddsp_table& t = ctx->tables[ctx->n_tables++];
    t.kind = kind;
    t.n0 = n0;
    t.n1 = n1;
    t.dev = dev;
    t.bytes = elems * sizeof(float);
    *out = dev;
    return DDSP_OK;
}
