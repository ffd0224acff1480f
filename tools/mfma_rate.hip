// Sustained fp32 MFMA rate per instruction shape (one launch each, all CUs, 4 waves per SIMD):
//   32x32x2 (4096 FLOP, 64 cycles nominal) and 16x16x4 (2048 FLOP, 32 cycles nominal), with 1, 2, 4 or 8 independent
//   accumulator chains per wave.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <int CH>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a, float b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
    if (s == 12345.f) out[threadIdx.x] = s;
}
// Do fp32 MFMA and vector-ALU instructions of DIFFERENT waves on one SIMD overlap?  Blocks 0..511 run an MFMA loop,
// blocks 512..1023 a dependent v_fma loop (4 chains per lane); with 4 blocks per CU every SIMD hosts waves of both
// kinds.  If the two pipes overlap, the mixed launch takes max(T_mfma, T_valu); if they share the issue slot /
// datapath, it takes about the sum.
__global__ void __launch_bounds__(256) kmix(float* out, int it_mfma, int it_valu, int mode, float a, float b) {
    const bool mfma_role = blockIdx.x < gridDim.x / 2;
    float s = 0.f;
    if (mfma_role) {
        if (mode & 1) {
            f32x4 acc[4];
            for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < it_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            }
            for (int c = 0; c < 4; ++c) s += acc[c][0];
        }
    } else if (mode & 2) {
        float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
        for (int i = 0; i < it_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                v0 = fmaf(v0, a, b);
                v1 = fmaf(v1, a, b);
                v2 = fmaf(v2, a, b);
                v3 = fmaf(v3, a, b);
            }
        }
        s = (v0 + v1) + (v2 + v3);
    }
    if (s == 12345.f) out[threadIdx.x] = s;
}
// the same question for the bf16 matrix instruction (32x32x16, 32 KFLOP): role A = bf16 MFMA loop, role B = v_fma loop
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) kmix_bf16(float* out, int it_mfma, int it_valu, int mode, float a, float b) {
    const bool mfma_role = blockIdx.x < gridDim.x / 2;
    float s = 0.f;
    if (mfma_role) {
        if ((mode & 1) && (mode & 16)) {           // fp32 16x16x4 MFMA role (4 chains)
            f32x4 acc4[4];
            for (int c = 0; c < 4; ++c) acc4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < it_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[c], 0, 0, 0);
            }
            for (int c = 0; c < 4; ++c) s += acc4[c][0];
        } else if (mode & 1) {
            bf16x8 va, vb;
            for (int i = 0; i < 8; ++i) {
                va[i] = (__bf16)a;
                vb[i] = (__bf16)b;
            }
            f32x16 acc[2];
            for (int c = 0; c < 2; ++c)
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
            for (int i = 0; i < it_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[c], 0, 0, 0);
            }
            s = acc[0][0] + acc[1][5];
        }
    } else if (mode & 2) {
        float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
        for (int i = 0; i < it_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                v0 = fmaf(v0, a, b);
                v1 = fmaf(v1, a, b);
                v2 = fmaf(v2, a, b);
                v3 = fmaf(v3, a, b);
            }
        }
        s = (v0 + v1) + (v2 + v3);
    } else if (mode & 4) {                      // plain (unpacked) v_fma_f32, forced through inline asm
        float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
        for (int i = 0; i < it_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
            }
        }
        s = (v0 + v1) + (v2 + v3);
    } else if (mode & 8) {                      // integer vector ops (address-arithmetic-like)
        unsigned v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 + 7, v3 = v0 ^ 5;
        const unsigned k = (unsigned)it_valu | 1;
        for (int i = 0; i < it_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(k));
            }
        }
        s = (float)((v0 + v1) ^ (v2 + v3));
    }
    if (s == 12345.f) out[threadIdx.x] = s;
}
static float time_mix_bf16(float* out, int mode, int it_mfma, int it_valu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kmix_bf16, dim3(1024), dim3(256), 0, 0, out, 10, 10, mode, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kmix_bf16, dim3(1024), dim3(256), 0, 0, out, it_mfma, it_valu, mode, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

static float time_mix(float* out, int mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kmix, dim3(1024), dim3(256), 0, 0, out, 10, 10, mode, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kmix, dim3(1024), dim3(256), 0, 0, out, 4000, 13000, mode, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <class F>
static void run(const char* name, F launch, double flop_per_mfma, int ch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 4;      // 4 blocks of 4 waves per CU = 4 waves per SIMD
    launch(blocks, 10);                             // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 8 * ch;
    printf("%-10s chains=%d  %.3f ms  %.1f TFLOP/s\n", name, ch, ms, mfmas * flop_per_mfma / (ms * 1e-3) / 1e12);
}
int main() {
    float* out;
    hipMalloc(&out, 4096);
#define R16(C) run("16x16x4", [&](int b, int it) { hipLaunchKernelGGL(k16<C>, dim3(b), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 2048.0, C)
#define R32(C) run("32x32x2", [&](int b, int it) { hipLaunchKernelGGL(k32<C>, dim3(b), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 4096.0, C)
    R32(1); R32(2); R32(4);
    R16(1); R16(2); R16(4); R16(8);
    const float tm = time_mix(out, 1), tv = time_mix(out, 2), tb = time_mix(out, 3);
    printf("mixed waves per SIMD: MFMA-only %.3f ms, VALU-only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", tm, tv, tb,
           tm + tv, tm > tv ? tm : tv);
    const int im = 16000, iv = 13000;
    const float bm = time_mix_bf16(out, 1, im, iv), bv = time_mix_bf16(out, 2, im, iv), bb = time_mix_bf16(out, 3, im, iv);
    for (int vm = 4; vm <= 8; vm *= 2) {
        const float xv = time_mix_bf16(out, vm, im, iv * 2), xb = time_mix_bf16(out, 1 | vm, im, iv * 2);
        printf("bf16 MFMA + %s waves per SIMD: MFMA-only %.3f ms, VALU-only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n",
               vm == 4 ? "asm v_fma_f32" : "integer v_add/v_xor", bm, xv, xb, bm + xv, bm > xv ? bm : xv);
    }
    {
        const int fm = 4000;                        // fp32 16x16x4 MFMA role: 4000 x 8 x 4 MFMAs per wave
        const float fmo = time_mix_bf16(out, 1 | 16, fm, 0);
        for (int vm = 4; vm <= 8; vm *= 2) {
            const float xv = time_mix_bf16(out, vm | 16, fm, iv * 2), xb = time_mix_bf16(out, 1 | 16 | vm, fm, iv * 2);
            printf("fp32 MFMA + %s waves per SIMD: MFMA-only %.3f ms, VALU-only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n",
                   vm == 4 ? "asm v_fma_f32" : "integer v_add/v_xor", fmo, xv, xb, fmo + xv, fmo > xv ? fmo : xv);
        }
    }
    printf("bf16 32x32x16 MFMA waves + v_fma waves per SIMD: MFMA-only %.3f ms (%.0f TFLOP/s), VALU-only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n",
           bm, 512.0 * 4 * im * 16 * 32768.0 / (bm * 1e-3) / 1e12, bv, bb, bm + bv, bm > bv ? bm : bv);
    return 0;
}
