// Handle, scratch arena and error text for libddsp_amd.
#include "common.h"

#include <stdlib.h>

#define DDSP_ABI_VERSION 1

extern "C" int ddsp_abi_version(void) { return DDSP_ABI_VERSION; }

extern "C" int ddsp_ctx_create(ddsp_ctx** out, int device) {
    if (!out) return DDSP_ERR_ARG;
    ddsp_ctx* c = (ddsp_ctx*)calloc(1, sizeof(ddsp_ctx));
    if (!c) return DDSP_ERR_OOM;
    c->device = device;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        free(c);
        return DDSP_ERR_HIP;  // no silent CPU fallback: the library is unusable without a GPU
    }
    if (device < 0 || device >= ndev) {
        free(c);
        return DDSP_ERR_ARG;
    }
    *out = c;
    return DDSP_OK;
}

extern "C" int ddsp_ctx_destroy(ddsp_ctx* ctx) {
    if (!ctx) return DDSP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->packed) (void)hipFree(ctx->packed);
    for (int i = 0; i < ctx->n_tables; ++i)
        if (ctx->tables[i].dev) (void)hipFree(ctx->tables[i].dev);
    free(ctx);
    return DDSP_OK;
}

extern "C" const char* ddsp_last_error(const ddsp_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

int ddsp_scratch_reserve_bytes(ddsp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return DDSP_OK;
    // Growing: earlier kernels may still read the old arena, so drain the device first.
    DDSP_HIP(ctx, hipSetDevice(ctx->device));
    DDSP_HIP(ctx, hipDeviceSynchronize());
    if (ctx->scratch) DDSP_HIP(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = bytes + (bytes >> 3) + (1u << 20);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return ddsp_fail(ctx, DDSP_ERR_OOM, "scratch hipMalloc", hipGetErrorString(e));
    ctx->scratch = (char*)p;
    ctx->scratch_bytes = want;
    return DDSP_OK;
}

extern "C" int ddsp_ctx_reserve(ddsp_ctx* ctx, uint64_t bytes) {
    if (!ctx) return DDSP_ERR_ARG;
    return ddsp_scratch_reserve_bytes(ctx, (size_t)bytes);
}

int ddsp_scratch_reset(ddsp_ctx* ctx) {
    ctx->scratch_used = 0;
    return DDSP_OK;
}

int ddsp_scratch_get(ddsp_ctx* ctx, size_t bytes, void** out) {
    size_t off = (ctx->scratch_used + 255) & ~(size_t)255;
    if (off + bytes > ctx->scratch_bytes)
        return ddsp_fail(ctx, DDSP_ERR_OOM, "scratch arena too small", "internal sizing error");
    *out = ctx->scratch + off;
    ctx->scratch_used = off + bytes;
    return DDSP_OK;
}
