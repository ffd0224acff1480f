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
    return 0;
}
