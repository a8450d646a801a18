// Internal helpers shared by the HIP translation units of libcognn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include "cognn_spec.h"

#include <atomic>
#include <vector>
struct cognn_timer_pair { hipEvent_t b, e; };
struct cognn_ctx;
// Element-wise launches of one kind queued between cognn_batch_begin / cognn_batch_end (kernels_elementwise.hip): the
// descriptors of up to kBatchMax calls are passed by value to ONE launch.
struct cognn_pending_batch {
    int (*flush)(cognn_ctx*) = nullptr;        // launcher of the queued functor type (nullptr: nothing queued)
    alignas(16) unsigned char storage[8192];
};
struct cognn_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    std::vector<cognn_timer_pair> timers[10];  // per kind
    std::vector<hipEvent_t> open_begin[10];
    std::vector<hipEvent_t> event_pool;        // timing events handed back by cognn_timer_reset: creating one costs more than a small kernel
    int batch_depth = 0;
    cognn_pending_batch pending;
    // launch lanes (cognn_lane_begin): auxiliary streams, created on first use
    hipStream_t main_stream = nullptr;
    std::vector<hipStream_t> lanes;
    std::vector<hipEvent_t> lane_done;
    std::vector<char> lane_used;
    hipEvent_t lane_fork = nullptr;
    int lanes_active = 0;
    unsigned long long* salt_sym[3] = {nullptr, nullptr, nullptr};   // the kernel translation units' copies of the epoch salt (cognn_spec.h)
    bool capturing = false;                                          // between cognn_graph_capture_begin / _end
    int chunk_c = 0, chunk_C = 1;                                    // chunk window of the element-wise entry points (cognn_ctx_set_chunk)
};
int cognn_set_error(const char* fmt, ...);
extern std::atomic<cognn_ctx*> cg_salt_owner;      // the context that holds a non-zero device-side epoch salt, if any (capi.hip)
static inline int cg_salt_guard(cognn_ctx* ctx) {
    cognn_ctx* o = cg_salt_owner.load(std::memory_order_relaxed);
    return (o && o != ctx) ? cognn_set_error("another context's epoch salt is set on the device (a recorded epoch is in flight): refused") : 0;
}
// stream-ordered zeroing by a kernel of this library (capi.hip).  Not hipMemsetAsync: a memset NODE of a recorded launch sequence
// replays wrongly on ROCm 7.2 once the process has used the legacy default stream in between - the zeroing is no longer ordered
// before the node that follows it (tools/repro_graph_memset_node.py; memcpy and kernel nodes are not affected).
int cg_zero(cognn_ctx* ctx, void* p, size_t bytes);
// launches whatever is queued
static inline int cg_flush_pending(cognn_ctx* ctx) {
    if (int rc = cg_salt_guard(ctx)) return rc;
    return (ctx && ctx->pending.flush) ? ctx->pending.flush(ctx) : 0;
}
// ... as every entry point other than the element-wise ones does first, so that stream order is preserved; those take no chunk window
static inline int cg_flush(cognn_ctx* ctx) {
    if (ctx && ctx->chunk_C > 1) return cognn_set_error("a chunk window is set (cognn_ctx_set_chunk): only the element-wise entry points may be called");
    return cg_flush_pending(ctx);
}

#define CG_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return cognn_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

#define CG_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return cognn_set_error(__VA_ARGS__);      \
    } while (0)

#define CG_LAUNCH_CHECK() CG_HIP(hipGetLastError())

// Dynamic LDS beyond the default limit needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  The attribute is process-wide
// state of the runtime: it is raised once per (device, kernel) to the largest size asked for so far, under a lock - not re-set before
// every launch (one runtime call less per launch, and several host threads launching the same kernel from their own contexts
// never lower it under each other's launches).
int cg_ensure_dynamic_lds(const void* kernel, int bytes);          // capi.hip

typedef unsigned long long u64;
typedef ulonglong2 u64x2;

static inline int cg_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool cg_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// floor(num / den) for 0 <= num < 2^52, 0 < den < 2^52 through one double-precision division and an exact integer correction - the same
// quotient as the 64-bit integer division (which the compiler expands to a long software sequence) at a fraction of its cost
__device__ __forceinline__ u64 div_floor_small(u64 num, u64 den) {
    u64 q = (u64)((double)num / (double)den);
    const long long rem = (long long)num - (long long)(q * den);
    if (rem < 0) --q;
    else if ((u64)rem >= den) ++q;
    return q;
}
