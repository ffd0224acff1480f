// SURVEY 8(f) rank 2 - the two device-side steps immediately BEFORE the synthesis path that need no pretrained model:
//   * frame-wise RMS volume (`Volume_Extractor.extract`, ddsp/vocoder.py:116-137): reflect-pad by (hop/2, (hop+1)/2),
//     mean of squares over non-overlapping hop-sized blocks, square root; T/hop + 1 frames;
//   * nearest-frame alignment of encoder units to the synthesiser's frame rate (`Units_Encoder.encode`,
//     ddsp/vocoder.py:201-211): frame i takes unit row min(rint(fp32(ratio) * i), Lu - 1), rint = half to even.
// Both are HBM-bound streaming kernels: 4 B read per sample / 8 B per copied feature.
#include "common.h"

namespace {

// one wavefront per (utterance, frame)
__global__ void __launch_bounds__(256) volume_kernel(const float* __restrict__ audio, int64_t T, int hop, int64_t n_frames,
                                                     int64_t total, float* __restrict__ vol) {
    const int lane = threadIdx.x & 63;
    const int64_t fidx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fidx >= total) return;
    const int64_t b = fidx / n_frames, n = fidx - b * n_frames;
    const float* x = audio + b * T;
    const int64_t first = n * hop - hop / 2;          // index of the block's first sample in the UNpadded signal
    double s = 0.0;                                   // (numpy sums the fp32 squares pairwise; fp64 is at least as close)
    for (int j = lane; j < hop; j += 64) {
        int64_t i = first + j;
        if (i < 0) i = -i;                            // numpy 'reflect': edge sample not repeated
        if (i >= T) i = 2 * (T - 1) - i;
        const float v = x[i];
        s += (double)(v * v);                         // the square is rounded to fp32 first, like `audio ** 2`
    }
    s = wave_sum_d(s);
    if (lane == 0) vol[fidx] = sqrtf((float)(s / (double)hop));
}

// one wavefront per output row (utterance, frame)
__global__ void __launch_bounds__(256) align_units_kernel(const float* __restrict__ units, int64_t Lu, int64_t C,
                                                          int64_t n_frames, float ratio, int64_t total,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total) return;
    const int64_t b = row / n_frames, i = row - b * n_frames;
    int64_t src = (int64_t)rintf(__fmul_rn(ratio, (float)i));     // torch.round of an fp32 product: half to even
    src = src < Lu - 1 ? src : Lu - 1;
    src = src < 0 ? 0 : src;
    const float* s = units + (b * Lu + src) * C;
    float* d = out + row * C;
    if ((C & 3) == 0 && (((uintptr_t)units | (uintptr_t)out) & 15) == 0) {
        for (int64_t c = 4 * lane; c < C; c += 256) *(f32x4*)(d + c) = *(const f32x4*)(s + c);
    } else {
        for (int64_t c = lane; c < C; c += 64) d[c] = s[c];
    }
}

// f0 track re-timed for the enhancer (enhancer.py:56-62): out[i] = numpy.interp(i * step_dst, knots j * step_num / div,
// fl32(f0[j] * scale)) with the end values held outside the knots; one thread per output frame, fp64 like numpy.
__global__ void __launch_bounds__(256) retime_f0_kernel(const float* __restrict__ f0, int64_t n_src, double step_num, double div,
                                                        float scale, double step_dst, int64_t n_dst, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_dst) return;
    auto knot = [&](int64_t j) { return (step_num * (double)j) / div; };              // (hop / sr) * arange(n) / real_factor
    auto val = [&](int64_t j) { return (double)__fmul_rn(f0[j], scale); };            // the fp32 in-place `f0_np *= real_factor`
    const double x = step_dst * (double)i;
    double y;
    if (n_src == 1 || x <= knot(0)) {
        y = val(0);
    } else if (x >= knot(n_src - 1)) {
        y = val(n_src - 1);
    } else {
        int64_t j = (int64_t)(x * div / step_num);
        j = j < 0 ? 0 : (j > n_src - 2 ? n_src - 2 : j);
        while (j > 0 && knot(j) > x) --j;
        while (j < n_src - 2 && knot(j + 1) <= x) ++j;
        const double slope = (val(j + 1) - val(j)) / (knot(j + 1) - knot(j));
        y = slope * (x - knot(j)) + val(j);
    }
    out[i] = (float)y;
}

}  // namespace

extern "C" int ddsp_retime_f0(ddsp_ctx* ctx, void* stream, const float* f0, int64_t n_src, double step_num, double div,
                              float scale, double step_dst, int64_t n_dst, float* out) {
    DDSP_REQUIRE(ctx, ctx && f0 && out, "ddsp_retime_f0: null argument");
    DDSP_REQUIRE(ctx, n_src >= 1 && n_dst >= 0 && step_num > 0 && div > 0 && step_dst > 0, "ddsp_retime_f0: bad shape or step");
    if (n_dst == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    hipLaunchKernelGGL(retime_f0_kernel, dim3((unsigned)ceil_div64(n_dst, 256)), dim3(256), 0, st, f0, n_src, step_num, div, scale,
                       step_dst, n_dst, out);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_volume_extract(ddsp_ctx* ctx, void* stream, const float* audio, int64_t B, int64_t T, int hop,
                                   float* volume) {
    DDSP_REQUIRE(ctx, ctx && audio && volume, "ddsp_volume_extract: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && hop >= 1 && hop <= (1 << 20), "ddsp_volume_extract: bad shape");
    // numpy's reflect padding needs the pad (at most (hop+1)/2) to be smaller than the signal
    DDSP_REQUIRE(ctx, T > (hop + 1) / 2, "ddsp_volume_extract: signal shorter than the reflect padding");
    if (B == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int64_t n_frames = T / hop + 1, total = B * n_frames;
    ddsp_prof_begin(ctx, st, PF_OTHER);
    hipLaunchKernelGGL(volume_kernel, dim3((unsigned)ceil_div64(total, 4)), dim3(256), 0, st, audio, T, hop, n_frames, total,
                       volume);
    ddsp_prof_end(ctx, st, 2.0 * B * T, 4.0 * (B * (double)T + total));
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}

extern "C" int ddsp_align_units(ddsp_ctx* ctx, void* stream, const float* units, int64_t B, int64_t Lu, int64_t C,
                                int64_t n_frames, float ratio, float* out) {
    DDSP_REQUIRE(ctx, ctx && units && out, "ddsp_align_units: null argument");
    DDSP_REQUIRE(ctx, B >= 0 && Lu >= 1 && C >= 1 && n_frames >= 0 && ratio >= 0.f && ratio == ratio,
                 "ddsp_align_units: bad shape or ratio");
    if (B == 0 || n_frames == 0) return DDSP_OK;
    hipStream_t st = (hipStream_t)stream;
    DDSP_ENTER_DEVICE(ctx);
    const int64_t total = B * n_frames;
    ddsp_prof_begin(ctx, st, PF_OTHER);
    hipLaunchKernelGGL(align_units_kernel, dim3((unsigned)ceil_div64(total, 4)), dim3(256), 0, st, units, Lu, C, n_frames,
                       ratio, total, out);
    ddsp_prof_end(ctx, st, 0.0, 8.0 * total * (double)C);
    DDSP_LAUNCH_CHECK(ctx);
    return DDSP_OK;
}
