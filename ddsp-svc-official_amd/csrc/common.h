// Shared host/device helpers for libddsp_amd (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ddsp_amd.h"

#define DDSP_WAVE 64

// Opaque handle behind ddsp_ctx: device scratch that outlives a call, DFT tables keyed by
// filter length, and the last error text.  One handle per stream/thread (SURVEY 8b "Threading").
struct ddsp_table {
    int kind;          // table family (see tables.hip)
    int n0, n1;        // key
    float* dev;        // device pointer
    size_t bytes;
    uint64_t last_use; // resample tap tables only: value of ddsp_ctx::table_clock at the last call that used it (LRU eviction)
};

// kernel families for the built-in HIP-event profiler (ddsp_profile_begin / _end)
enum {
    PF_PHASE_SCAN = 0, PF_FIR_ACT, PF_FIR_DFT_GEMM, PF_LTV_FIR, PF_U2C_PREP, PF_U2C_GEMM_CONV3, PF_U2C_GEMM_LINEAR,
    PF_U2C_GEMM_FEAT, PF_U2C_GEMM_CTX, PF_U2C_GEMM_ATTNOUT, PF_U2C_ROWWISE, PF_SINS_BANK, PF_SPECTRAL_OLA,
    PF_RSS_LOSS, PF_SOLA, PF_UPSAMPLE, PF_OTHER, PF_LTV_FIR_BWD, PF_FIR_SYNTH_BWD, PF_U2C_BWD, PF_OPTIM, PF_COUNT
};

struct ddsp_prof_rec {
    hipEvent_t start, stop;
    int id;
    double flops, bytes;
};
#define DDSP_PROF_CAP 16384

struct ddsp_ctx {
    int device;
    int math;   // product arithmetic of the inference GEMMs (ddsp_ctx_set_math): DDSP_MATH_SPLIT_BF16 (default) or DDSP_MATH_FP32
    // profiler
    uint64_t prof_mask;
    ddsp_prof_rec* prof;
    int prof_n, prof_events_made;
    int prof_open;
    char err[512];
    // bump-allocated scratch, grown on demand between calls (never inside a captured region)
    char* scratch;
    size_t scratch_bytes;
    size_t scratch_used;
    // prepared weights of the control network (ddsp_u2c_weights::version != 0): one slot
    char* wcache;
    size_t wcache_bytes;
    uint64_t wcache_key, wcache_version;   // hash of the caller's struct (pointers, sizes) / its change counter; key 0 = empty
    int wcache_flags;                      // what the slot holds: bit 0 split copies, bit 1 fused-GLU order, bit 2 attention pieces
    ddsp_table tables[64];
    int n_tables;
    uint64_t table_clock;
    // packed control-net weights (prepared by ddsp_u2c_prepare)
    float* packed;
    size_t packed_bytes;
    // 8 KiB of zeros: the source of out-of-range conv taps in the LDS-DMA GEMM (a DMA cannot be predicated to zero)
    float* zero_page;
    // device-side contract violations (a speaker id outside the table): one int in host-mapped memory that kernels set
    // with a system-scope store; the next entry point that takes ids, or ddsp_ctx_poll_error, reports and clears it
    int* dev_error_host;
    int* dev_error_dev;
};

static inline int ddsp_fail(ddsp_ctx* ctx, int code, const char* what, const char* detail) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, detail ? detail : "");
    return code;
}

#define DDSP_HIP(ctx, call)                                                     \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) return ddsp_fail(ctx, DDSP_ERR_HIP, #call, hipGetErrorString(e_)); \
    } while (0)

// The calling thread's current device is changed only for the duration of a library call: DDSP_ENTER_DEVICE
// selects the context's device and the guard's destructor restores the caller's when the function returns.
struct ddsp_device_guard {
    int prev = -1;
    bool changed = false;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev == dev) return hipSuccess;
        e = hipSetDevice(dev);
        changed = (e == hipSuccess);
        return e;
    }
    ~ddsp_device_guard() {
        if (changed) (void)hipSetDevice(prev);
    }
};
#define DDSP_ENTER_DEVICE(ctx)   \
    ddsp_device_guard dev_guard_; \
    DDSP_HIP(ctx, dev_guard_.enter((ctx)->device))
// hipFuncSetAttribute and its like are per device: run `...` once for each device a context of this process lives on
#define DDSP_ONCE_PER_DEVICE(ctx, ...)                                                   \
    do {                                                                                 \
        static std::atomic<uint64_t> done_{0};                                           \
        const uint64_t bit_ = 1ull << ((ctx)->device & 63);                              \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {                           \
            __VA_ARGS__;                                                                 \
            done_.fetch_or(bit_, std::memory_order_release);                             \
        }                                                                                \
    } while (0)

#define DDSP_REQUIRE(ctx, cond, msg)                                            \
    do {                                                                        \
        if (!(cond)) return ddsp_fail(ctx, DDSP_ERR_ARG, msg, #cond);           \
    } while (0)

// A launch helper that finds its arguments inconsistent (gemm::launch: operands that exist only pre-split on a path that
// cannot read them) records it here instead of launching; the entry point's next DDSP_LAUNCH_CHECK turns it into an error
// return (it used to abort() the host process).  Per thread, like the calls.
inline const char*& ddsp_launch_refusal() {
    static thread_local const char* what = nullptr;
    return what;
}
#define DDSP_LAUNCH_CHECK(ctx)                                                  \
    do {                                                                        \
        if (const char* r_ = ddsp_launch_refusal()) {                           \
            ddsp_launch_refusal() = nullptr;                                    \
            return ddsp_fail(ctx, DDSP_ERR_ARG, "kernel launch refused", r_);   \
        }                                                                       \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) return ddsp_fail(ctx, DDSP_ERR_HIP, "kernel launch", hipGetErrorString(e_)); \
    } while (0)

// scratch (ctx.hip)
int ddsp_scratch_reset(ddsp_ctx* ctx);
int ddsp_scratch_get(ddsp_ctx* ctx, size_t bytes, void** out);
int ddsp_scratch_reserve_bytes(ddsp_ctx* ctx, size_t bytes);
// tables (tables.hip)
int ddsp_get_table(ddsp_ctx* ctx, hipStream_t st, int kind, int n0, int n1, float** out);
// device pointer to DDSP_ZERO_FLOATS zeros (allocated on first use; a first-use synchronisation like the tables)
constexpr int DDSP_ZERO_FLOATS = 2048;
int ddsp_zero_page(ddsp_ctx* ctx, const float** out);
// device pointer to the context's error flag (allocated on first use); ddsp_take_dev_error returns DDSP_ERR_ARG and
// clears the flag when a kernel of an earlier, completed call raised it
int ddsp_dev_error_ptr(ddsp_ctx* ctx, int** out);
int ddsp_take_dev_error(ddsp_ctx* ctx);
#define DDSP_DEV_ERR_SPK_ID 1

// profiler hooks (ctx.hip): bracket ONE kernel launch (or a tight group) on `st` when the family is enabled
void ddsp_prof_begin(ddsp_ctx* ctx, hipStream_t st, int id);
void ddsp_prof_end(ddsp_ctx* ctx, hipStream_t st, double flops, double bytes);

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// x[0..7] -> 8 bf16 hi parts (16 bytes) and 8 bf16 lo parts (16 bytes): the pre-split operand layout of the split-bf16 GEMM
// (gemm::Args::A_split / B_split), same rounding as its in-kernel split (round to nearest even, twice)
typedef uint32_t ddsp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ddsp_split8(const float (&x)[8], ddsp_u32x4& hi, ddsp_u32x4& lo) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const f32x2v r = {x[2 * d], x[2 * d + 1]};
        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2v));
        hi[d] = h;
        const f32x2v rem = r - (f32x2v){__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
        lo[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, bf16x2v));
    }
}
// The same for a lane that owns 4 of the 8 values, its lane-pair partner (lane ^ xor_mask) the other 4: the lane with the
// first four values ends up with the group's 8 hi parts, the other with the 8 lo parts - each lane still stores 16 bytes
// at its own slot, only the contents differ from an fp32 store.
__device__ __forceinline__ ddsp_u32x4 ddsp_split4_pair(f32x4 v, bool second, int xor_mask) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    uint32_t hi[2], lo[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const f32x2v r = {v[2 * d], v[2 * d + 1]};
        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2v));
        hi[d] = h;
        const f32x2v rem = r - (f32x2v){__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
        lo[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rem, bf16x2v));
    }
    // first lane keeps hi and needs the partner's hi; second lane keeps lo and needs the partner's lo
    const uint32_t s0 = second ? hi[0] : lo[0], s1 = second ? hi[1] : lo[1];
    const uint32_t r0 = (uint32_t)__shfl_xor((int)s0, xor_mask, 64), r1 = (uint32_t)__shfl_xor((int)s1, xor_mask, 64);
    return second ? ddsp_u32x4{r0, r1, lo[0], lo[1]} : ddsp_u32x4{hi[0], hi[1], r0, r1};
}

// The same for EIGHT neighbouring lanes that own one value each (lane j = lane & 7 of the group owns element j): returns
// the dword this lane must store at ITS OWN element's position - dwords 0..3 of the group are the hi pairs, 4..7 the lo pairs.
__device__ __forceinline__ uint32_t ddsp_split1_group8(float x, int lane) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){x, 0.f}, bf16x2v)) & 0xffffu;
    const float rem = x - __builtin_bit_cast(float, h << 16);
    const uint32_t l = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){rem, 0.f}, bf16x2v)) & 0xffffu;
    const uint32_t pk = h | (l << 16);
    const int j = lane & 7, src = (lane & ~7) + 2 * (j & 3);
    const uint32_t a = (uint32_t)__shfl((int)pk, src, 64), b = (uint32_t)__shfl((int)pk, src + 1, 64);
    return j < 4 ? (a & 0xffffu) | (b << 16) : (a >> 16) | (b & 0xffff0000u);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif
