// Context, memory and error plumbing of the C ABI (include/cognn_hip.h).
#include "common.h"
#include "../../include/cognn_hip.h"
#include <string.h>
#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[1024] = "";

int cognn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// The device-side epoch salt is ONE set of symbols per process (cognn_spec.h), so at most one context may hold it non-zero at a
// time, and while it does every entry point of every OTHER context is refused (cg_flush / launch_ew check cg_salt_owner): a
// second context never silently evaluates dealer streams under a foreign salt.  The engine only uses it for recorded epochs
// (COGNN_OPT_GRAPH_EPOCHS); eager launches carry the salt in their keys.
std::atomic<cognn_ctx*> cg_salt_owner{nullptr};
int cg_ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> raised;
    int dev = 0;
    CG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    int& cur = raised[{dev, kernel}];
    if (bytes <= cur) return 0;
    CG_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    cur = bytes;
    return 0;
}
void* cg_salt_symbol_kernels_elementwise();
void* cg_salt_symbol_kernels_gather();
void* cg_salt_symbol_kernels_gemm();
namespace {
__global__ void salt_set_kernel(unsigned long long* a, unsigned long long* b, unsigned long long* c, unsigned long long v) {
    if (threadIdx.x == 0) { *a = v; *b = v; *c = v; }
}
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned long long* p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0ull;
}
__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* p, size_t bytes) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < bytes; i += (size_t)gridDim.x * 256) p[i] = 0;
}
}  // namespace
int cg_zero(cognn_ctx* ctx, void* p, size_t bytes) {
    if (!bytes) return 0;
    static const bool with_memset = getenv("COGNN_ZERO_WITH_MEMSET") != nullptr;   // (the runtime's memset: only for tools/repro_graph_memset_node.py)
    if (with_memset) { CG_HIP(hipMemsetAsync(p, 0, bytes, ctx->stream)); return 0; }
    const bool words = (((uintptr_t)p | bytes) & 7u) == 0;
    const size_t n = words ? bytes / 8 : bytes;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
    if (words) hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(256), 0, ctx->stream, (unsigned long long*)p, n);
    else hipLaunchKernelGGL(zero_bytes_kernel, dim3(grid), dim3(256), 0, ctx->stream, (unsigned char*)p, n);
    CG_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int cognn_abi_version(void) { return COGNN_ABI_VERSION; }
const char* cognn_last_error(void) { return g_err; }

int cognn_ctx_create(int device, void* stream, cognn_ctx** out) {
    CG_REQUIRE(out, "cognn_ctx_create: null out");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return cognn_set_error("cognn_ctx_create: no HIP device available (%s); the engine has no CPU fallback",
                               e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    CG_REQUIRE(device >= 0 && device < count, "cognn_ctx_create: device %d out of range (count %d)", device, count);
    CG_HIP(hipSetDevice(device));
    cognn_ctx* c = new cognn_ctx();
    c->device = device;
    c->own_stream = false;
    c->stream = (hipStream_t)stream;          /* NULL = the device's default (null) stream */
    *out = c;
    return 0;
}
int cognn_ctx_create_private(int device, cognn_ctx** out) {
    int rc = cognn_ctx_create(device, nullptr, out);
    if (rc) return rc;
    CG_HIP(hipStreamCreateWithFlags(&(*out)->stream, hipStreamNonBlocking));
    (*out)->own_stream = true;
    return 0;
}
int cognn_ctx_destroy(cognn_ctx* ctx) {
    if (!ctx) return 0;
    if (cg_salt_owner.load() == ctx) (void)cognn_set_epoch_salt(ctx, 0);
    (void)cognn_timer_reset(ctx);
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    ctx->event_pool.clear();
    if (ctx->lanes_active) ctx->stream = ctx->main_stream;
    for (auto st : ctx->lanes) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto ev : ctx->lane_done) (void)hipEventDestroy(ev);
    if (ctx->lane_fork) (void)hipEventDestroy(ctx->lane_fork);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}
int cognn_ctx_sync(cognn_ctx* ctx) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx, "cognn_ctx_sync: null ctx");
    CG_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
int cognn_lane_begin(cognn_ctx* ctx, int32_t lanes) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && lanes >= 1 && lanes <= 4 && !ctx->lanes_active, "cognn_lane_begin: bad arguments or lanes already open");
    while ((int)ctx->lanes.size() < lanes) {
        hipStream_t st; hipEvent_t ev;
        CG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ctx->lanes.push_back(st);
        CG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->lane_done.push_back(ev);
    }
    if (!ctx->lane_fork) CG_HIP(hipEventCreateWithFlags(&ctx->lane_fork, hipEventDisableTiming));
    CG_HIP(hipEventRecord(ctx->lane_fork, ctx->stream));
    for (int l = 0; l < lanes; ++l) CG_HIP(hipStreamWaitEvent(ctx->lanes[l], ctx->lane_fork, 0));
    ctx->main_stream = ctx->stream;
    ctx->lanes_active = lanes;
    ctx->lane_used.assign((size_t)lanes, 0);
    return 0;
}
int cognn_lane_select(cognn_ctx* ctx, int32_t lane) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && ctx->lanes_active && lane >= 0 && lane < ctx->lanes_active, "cognn_lane_select: no such lane");
    ctx->stream = ctx->lanes[lane];
    ctx->lane_used[lane] = 1;
    return 0;
}
int cognn_lane_end(cognn_ctx* ctx) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && ctx->lanes_active, "cognn_lane_end: no lanes open");
    ctx->stream = ctx->main_stream;
    const int n = ctx->lanes_active;
    ctx->lanes_active = 0;
    for (int l = 0; l < n; ++l) {
        if (!ctx->lane_used[l]) continue;
        CG_HIP(hipEventRecord(ctx->lane_done[l], ctx->lanes[l]));
        CG_HIP(hipStreamWaitEvent(ctx->stream, ctx->lane_done[l], 0));
    }
    return 0;
}
int cognn_batch_begin(cognn_ctx* ctx) {
    CG_REQUIRE(ctx, "cognn_batch_begin: null context");
    ++ctx->batch_depth;
    return 0;
}
int cognn_batch_end(cognn_ctx* ctx) {
    CG_REQUIRE(ctx && ctx->batch_depth > 0, "cognn_batch_end: no open batch");
    return --ctx->batch_depth == 0 ? cg_flush_pending(ctx) : 0;
}
void cognn_chunk_range(int64_t n, int32_t c, int32_t C, int64_t* lo, int64_t* hi) {
    if (C <= 1) { *lo = 0; *hi = n; return; }
    *lo = (n * c / C) & ~(int64_t)1;
    *hi = (c + 1 == C) ? n : ((n * (c + 1) / C) & ~(int64_t)1);
}
int cognn_ctx_set_chunk(cognn_ctx* ctx, int32_t c, int32_t C) {
    CG_REQUIRE(ctx && ((C <= 1 && c == 0) || (C >= 2 && C <= 64 && c >= 0 && c < C)), "cognn_ctx_set_chunk: chunk %d of %d", (int)c, (int)C);
    ctx->chunk_c = C <= 1 ? 0 : c; ctx->chunk_C = C <= 1 ? 1 : C;     // (queued calls keep the window they were issued under)
    return 0;
}
int cognn_malloc(cognn_ctx* ctx, void** ptr, size_t bytes) {
    CG_REQUIRE(ctx && ptr, "cognn_malloc: null argument");
    CG_HIP(hipSetDevice(ctx->device));
    CG_HIP(hipMalloc(ptr, bytes ? bytes : 16));
    return 0;
}
int cognn_free(cognn_ctx* ctx, void* ptr) {
    CG_REQUIRE(ctx, "cognn_free: null ctx");
    if (ptr) CG_HIP(hipFree(ptr));
    return 0;
}
int cognn_memcpy_h2d(cognn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx, "cognn_memcpy_h2d: null ctx");
    if (bytes) { CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream)); CG_HIP(hipStreamSynchronize(ctx->stream)); }
    return 0;
}
int cognn_memcpy_d2h(cognn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx, "cognn_memcpy_d2h: null ctx");
    if (bytes) { CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream)); CG_HIP(hipStreamSynchronize(ctx->stream)); }
    return 0;
}
int cognn_memcpy_d2d(cognn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx, "cognn_memcpy_d2d: null ctx");
    if (bytes) CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
int cognn_memset0(cognn_ctx* ctx, void* dst, size_t bytes) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx, "cognn_memset0: null ctx");
    if (int rc = cg_zero(ctx, dst, bytes)) return rc;
    return 0;
}
static int timer_event(cognn_ctx* ctx, hipEvent_t* ev) {
    if (!ctx->event_pool.empty()) { *ev = ctx->event_pool.back(); ctx->event_pool.pop_back(); return 0; }
    CG_HIP(hipEventCreate(ev));
    return 0;
}
int cognn_timer_begin(cognn_ctx* ctx, int kind) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && kind >= 0 && kind < 10, "cognn_timer_begin: bad arguments");
    hipEvent_t ev;
    if (int rc = timer_event(ctx, &ev)) return rc;
    CG_HIP(hipEventRecord(ev, ctx->stream));
    ctx->open_begin[kind].push_back(ev);
    return 0;
}
int cognn_timer_end(cognn_ctx* ctx, int kind) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && kind >= 0 && kind < 10 && !ctx->open_begin[kind].empty(), "cognn_timer_end: no open timer");
    hipEvent_t ev;
    if (int rc = timer_event(ctx, &ev)) return rc;
    CG_HIP(hipEventRecord(ev, ctx->stream));
    cognn_timer_pair pr{ctx->open_begin[kind].back(), ev};
    ctx->open_begin[kind].pop_back();
    ctx->timers[kind].push_back(pr);
    return 0;
}
int cognn_timer_read(cognn_ctx* ctx, int kind, int64_t* launches, double* total_ms) {
    CG_REQUIRE(ctx && kind >= 0 && kind < 10 && launches && total_ms, "cognn_timer_read: bad arguments");
    CG_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0;
    for (auto& pr : ctx->timers[kind]) {
        float ms = 0;
        CG_HIP(hipEventElapsedTime(&ms, pr.b, pr.e));
        tot += ms;
    }
    *launches = (int64_t)ctx->timers[kind].size();
    *total_ms = tot;
    return 0;
}
int cognn_timer_reset(cognn_ctx* ctx) {
    CG_REQUIRE(ctx, "cognn_timer_reset: null ctx");
    for (size_t k = 0; k < sizeof(ctx->timers) / sizeof(ctx->timers[0]); ++k) {
        for (auto& pr : ctx->timers[k]) { ctx->event_pool.push_back(pr.b); ctx->event_pool.push_back(pr.e); }
        ctx->timers[k].clear();
        for (auto& ev : ctx->open_begin[k]) ctx->event_pool.push_back(ev);
        ctx->open_begin[k].clear();
    }
    return 0;
}
int cognn_set_epoch_salt(cognn_ctx* ctx, uint64_t salt) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx, "cognn_set_epoch_salt: null ctx");
    if (!ctx->salt_sym[0]) {
        ctx->salt_sym[0] = (unsigned long long*)cg_salt_symbol_kernels_elementwise();
        ctx->salt_sym[1] = (unsigned long long*)cg_salt_symbol_kernels_gather();
        ctx->salt_sym[2] = (unsigned long long*)cg_salt_symbol_kernels_gemm();
        CG_REQUIRE(ctx->salt_sym[0] && ctx->salt_sym[1] && ctx->salt_sym[2], "cognn_set_epoch_salt: cannot resolve the salt symbols");
    }
    if (salt != 0) {
        cognn_ctx* none = nullptr;
        if (!cg_salt_owner.compare_exchange_strong(none, ctx) && none != ctx)
            return cognn_set_error("cognn_set_epoch_salt: another context holds a non-zero epoch salt (one recorded-epoch engine per process at a time)");
        if (none == nullptr) CG_HIP(hipDeviceSynchronize());   // just acquired: what other contexts have in flight evaluates its streams under salt 0
    }
    hipLaunchKernelGGL(salt_set_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->salt_sym[0], ctx->salt_sym[1], ctx->salt_sym[2], (unsigned long long)salt);
    CG_LAUNCH_CHECK();
    if (salt == 0 && cg_salt_owner.load() == ctx) {
        CG_HIP(hipStreamSynchronize(ctx->stream));          // the symbols read 0 again before any other context may launch
        cg_salt_owner.store(nullptr);
    }
    return 0;
}
// ---- a recorded sequence of launches (hipGraph): the engine records one epoch and replays it (COGNN_OPT_GRAPH_EPOCHS) ----------
int cognn_ctx_use_private_stream(cognn_ctx* ctx) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && !ctx->lanes_active && !ctx->capturing, "cognn_ctx_use_private_stream: bad state");
    if (ctx->own_stream) return 0;
    CG_HIP(hipStreamSynchronize(ctx->stream));                 // everything issued so far on the caller's stream is done
    hipStream_t st;
    CG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));   // (the legacy default stream cannot be captured)
    ctx->stream = st;
    ctx->own_stream = true;
    return 0;
}
int cognn_graph_capture_begin(cognn_ctx* ctx) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && ctx->own_stream && !ctx->capturing && !ctx->lanes_active, "cognn_graph_capture_begin: needs a private stream (cognn_ctx_use_private_stream) and no open lanes");
    CG_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return 0;
}
int cognn_graph_capture_end(cognn_ctx* ctx, void** exec_out) {
    CG_REQUIRE(ctx && ctx->capturing && exec_out, "cognn_graph_capture_end: not capturing");
    const int rc_flush = cg_flush(ctx);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    ctx->capturing = false;
    if (rc_flush) { if (graph) (void)hipGraphDestroy(graph); return rc_flush; }
    if (e != hipSuccess || !graph) return cognn_set_error("cognn_graph_capture_end: hipStreamEndCapture failed: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) return cognn_set_error("cognn_graph_capture_end: hipGraphInstantiate failed: %s", hipGetErrorString(ie));
    *exec_out = (void*)exec;
    return 0;
}
int cognn_graph_launch(cognn_ctx* ctx, void* exec) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && exec && !ctx->capturing, "cognn_graph_launch: bad arguments");
    CG_HIP(hipGraphLaunch((hipGraphExec_t)exec, ctx->stream));
    return 0;
}
int cognn_graph_destroy(cognn_ctx* ctx, void* exec) {
    CG_REQUIRE(ctx, "cognn_graph_destroy: null ctx");
    if (exec) { (void)hipStreamSynchronize(ctx->stream); CG_HIP(hipGraphExecDestroy((hipGraphExec_t)exec)); }
    return 0;
}
void cognn_make_keys(uint64_t seed, uint64_t owner, uint64_t iter, uint64_t op, cognn_keys* out) {
    cognn_opkeys k = cognn_make_opkeys(seed, owner, iter, op);
    for (int i = 0; i < COGNN_SL_COUNT; ++i) out->k[i] = k.k[i];
}

}  // extern "C"
