// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// Plain C++ (simple loops, host memory) implementation of every entry point of include/cognn_hip.h, bound
// as the arithmetic backend of cognn_amd/host/engine.cpp in oracle/libcognn_engine_cpu.so.  It exists so that
//   (1) the engine's host logic (layout, CSR construction, schedule, multi-rank exchange lists) can be
//       checked on CPU against oracle/cognn_oracle.py, including world_size-2 runs over gloo, and
//   (2) bench.py's cpu_baseline leg can time a compiled CPU port next to the GPU number.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (cognn_amd/libcognn_hip.so) neither links nor falls back to it and fails loudly without a GPU.
// Definitions follow DESIGN.md §3 and cognn_amd/csrc/cognn_spec.h (the numpy oracle restates them independently).
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define COGNN_HOST_SALT                                     /* cognn_prng adds cognn_host_salt (the epoch salt of cognn_spec.h) */
#include "../cognn_amd/csrc/cognn_spec.h"
#include "../cognn_amd/host/backend.h"

uint64_t cognn_host_salt = 0;

// OpenMP over loops whose iterations are independent (bench.py's cpu_baseline uses all host cores; tests run with 1 thread)
#define CG_PAR _Pragma("omp parallel for schedule(static)")

typedef uint64_t u64;

struct cognn_ctx {
    std::vector<std::chrono::high_resolution_clock::time_point> open[10];
    double total_ms[10] = {0};
    int64_t count[10] = {0};
};

static thread_local char g_err[512] = "";
// chunk window of the element-wise entry points (cognn_ctx_set_chunk); one engine per thread in the tests
static thread_local int g_chunk_c = 0, g_chunk_C = 1;
#define CHUNK(n) int64_t lo_, hi_; cognn_chunk_range((n), g_chunk_c, g_chunk_C, &lo_, &hi_);
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define REQ(c, msg) do { if (!(c)) return fail("%s", msg); } while (0)
// the alignment contract of the HIP entry points (kernels_elementwise.hip: al()), so that the host logic meets it on this backend too
#define AL(p) ((((uintptr_t)(const void*)(p)) & 15u) == 0)

static inline cognn_opkeys K(const cognn_keys* k) {
    cognn_opkeys r;
    for (int i = 0; i < COGNN_SL_COUNT; ++i) r.k[i] = k->k[i];
    return r;
}
static inline u64 trunc_r(const cognn_opkeys& k, int p, u64 i) {
    u64 r0 = cognn_prng(k.k[COGNN_SL_R0], i);
    return p == 0 ? r0 : (cognn_prng(k.k[COGNN_SL_R], i) & COGNN_TRUNC_MASK) - r0;
}
static inline u64 trunc_rp(const cognn_opkeys& k, int p, u64 i) {
    u64 rp0 = cognn_prng(k.k[COGNN_SL_RP0], i);
    return p == 0 ? rp0 : ((cognn_prng(k.k[COGNN_SL_R], i) & COGNN_TRUNC_MASK) >> COGNN_FX_BITS) - rp0;
}
static inline u64 beaver_mul(const cognn_opkeys& k, int p, u64 e, u64 g, u64 i, u64 bi) {
    u64 a0 = cognn_prng(k.k[COGNN_SL_A0], i), b0 = cognn_prng(k.k[COGNN_SL_B0], bi), c0 = cognn_prng(k.k[COGNN_SL_C0], i);
    if (p == 0) return e * b0 + a0 * g + c0;
    u64 a1 = cognn_prng(k.k[COGNN_SL_A1], i), b1 = cognn_prng(k.k[COGNN_SL_B1], bi);
    return e * g + e * b1 + a1 * g + ((a0 + a1) * (b0 + b1) - c0);
}
static inline u64 lidx(int64_t i, int64_t rows, int64_t cols, int transposed) {
    if (transposed != 1) return (u64)i;                    // 0: row-major; 2: transposed storage, mask indexed in storage order
    const u64 k = (u64)i / (u64)rows, m = (u64)i % (u64)rows;     // storage [cols x rows]
    return m * (u64)cols + k;
}

extern "C" {

int cognn_abi_version(void) { return COGNN_ABI_VERSION; }
const char* cognn_last_error(void) { return g_err; }
int cognn_ctx_create(int, void*, cognn_ctx** out) { REQ(out, "ctx_create: null"); *out = new cognn_ctx(); return 0; }
int cognn_ctx_create_private(int d, cognn_ctx** out) { return cognn_ctx_create(d, nullptr, out); }
int cognn_ctx_destroy(cognn_ctx* c) { delete c; return 0; }
int cognn_ctx_sync(cognn_ctx*) { return 0; }
int cognn_malloc(cognn_ctx*, void** p, size_t bytes) {
    REQ(p, "malloc: null");
    if (posix_memalign(p, 64, bytes ? bytes : 64) != 0) return fail("out of host memory");
    return 0;
}
int cognn_free(cognn_ctx*, void* p) { free(p); return 0; }
int cognn_memcpy_h2d(cognn_ctx*, void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int cognn_memcpy_d2h(cognn_ctx*, void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int cognn_memcpy_d2d(cognn_ctx*, void* d, const void* s, size_t n) { if (n) memmove(d, s, n); return 0; }
int cognn_memset0(cognn_ctx*, void* d, size_t n) { if (n) memset(d, 0, n); return 0; }
void cognn_make_keys(uint64_t seed, uint64_t owner, uint64_t iter, uint64_t op, cognn_keys* out) {
    cognn_opkeys k = cognn_make_opkeys(seed, owner, iter, op);
    for (int i = 0; i < COGNN_SL_COUNT; ++i) out->k[i] = k.k[i];
}

int cognn_fx_encode_f64(cognn_ctx*, const double* in, const double* rs, uint64_t* fx, int64_t rows, int64_t cols) {
    CG_PAR
    for (int64_t i = 0; i < rows * cols; ++i) {
        double v = in[i];
        if (rs) v *= rs[i / cols];
        fx[i] = (u64)(long long)llround(v * (double)COGNN_FX_ONE);
    }
    return 0;
}
int cognn_share_split_u64(cognn_ctx*, const uint64_t* fx, uint64_t key, uint64_t* s0, uint64_t* s1, int64_t n) {
    CG_PAR
    for (int64_t i = 0; i < n; ++i) {
        const u64 b = cognn_prng(key, (u64)i);
        if (s0) s0[i] = fx[i] - b;
        if (s1) s1[i] = b;
    }
    return 0;
}
void cognn_chunk_range(int64_t n, int32_t c, int32_t C, int64_t* lo, int64_t* hi) {
    if (C <= 1) { *lo = 0; *hi = n; return; }
    *lo = (n * c / C) & ~(int64_t)1;
    *hi = (c + 1 == C) ? n : ((n * (c + 1) / C) & ~(int64_t)1);
}
int cognn_ctx_set_chunk(cognn_ctx*, int32_t c, int32_t C) {
    REQ((C <= 1 && c == 0) || (C >= 2 && C <= 64 && c >= 0 && c < C), "ctx_set_chunk: bad window");
    g_chunk_c = C <= 1 ? 0 : c; g_chunk_C = C <= 1 ? 1 : C;
    return 0;
}
int cognn_batch_begin(cognn_ctx*) { return 0; }              // the reference backend runs every call immediately
int cognn_batch_end(cognn_ctx*) { return 0; }
int cognn_lane_begin(cognn_ctx*, int32_t) { return 0; }      // ... and in program order
int cognn_lane_select(cognn_ctx*, int32_t) { return 0; }
int cognn_lane_end(cognn_ctx*) { return 0; }
int cognn_prng_fill_u64(cognn_ctx*, uint64_t* out, uint64_t key, int64_t n) {
    REQ(AL(out), "cognn_prng_fill_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CG_PAR
    for (int64_t i = 0; i < n; ++i) out[i] = cognn_prng(key, (u64)i);
    return 0;
}
int cognn_pack48_u64(cognn_ctx*, void* packed, const uint64_t* src, int64_t n) {     // the 6-byte wire form of an opened truncation share
    uint32_t* mid = (uint32_t*)packed; uint16_t* top = (uint16_t*)((unsigned char*)packed + 4 * n);
    for (int64_t i = 0; i < n; ++i) { mid[i] = (uint32_t)(src[i] >> 16); top[i] = (uint16_t)(src[i] >> 48); }
    return 0;
}
int cognn_unpack48_u64(cognn_ctx*, uint64_t* dst, const void* packed, int64_t n) {
    const uint32_t* mid = (const uint32_t*)packed; const uint16_t* top = (const uint16_t*)((const unsigned char*)packed + 4 * n);
    for (int64_t i = 0; i < n; ++i) dst[i] = ((u64)mid[i] << 16) | ((u64)top[i] << 48);
    return 0;
}
int cognn_gemm_mask_fill_u64(cognn_ctx*, uint64_t* out, uint64_t key, int64_t n) {
    REQ(AL(out), "cognn_gemm_mask_fill_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");   // a product's A mask: limb-form values (cognn_spec.h)
    CG_PAR
    for (int64_t i = 0; i < n; ++i) out[i] = cognn_gemm_mask(key, (u64)i);
    return 0;
}
int cognn_gather_csr_u64(cognn_ctx*, uint64_t* out, const uint64_t* base, const uint64_t* table, const uint32_t* rowptr,
                         const uint32_t* col, int64_t n_rows, int64_t F) {
    CG_PAR
    for (int64_t r = 0; r < n_rows; ++r) {
        std::vector<u64> acc((size_t)F);                    // out may alias base/table rows of other segments, never row r's sources
        for (int64_t j = 0; j < F; ++j) acc[j] = base ? base[r * F + j] : 0;
        for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const u64* src = table + (size_t)col[e] * F;
            for (int64_t j = 0; j < F; ++j) acc[j] += src[j];
        }
        for (int64_t j = 0; j < F; ++j) out[r * F + j] = acc[j];
    }
    return 0;
}
int cognn_gather_csr_open_u64(cognn_ctx* c, uint64_t* out, const uint64_t* base, const uint64_t* table, const uint32_t* rowptr,
                              const uint32_t* col, int64_t n_rows, int64_t F, int32_t nseg, const int64_t* sb, const int64_t* se,
                              const uint64_t* sk) {
    cognn_gather_csr_u64(c, out, base, table, rowptr, col, n_rows, F);
    for (int s = 0; s < nseg; ++s)
        for (int64_t r = sb[s]; r < se[s]; ++r)
            for (int64_t j = 0; j < F; ++j) out[r * F + j] -= cognn_prng(sk[s], (u64)((r - sb[s]) * F + j));
    return 0;
}
int cognn_scatter_add_rows_u64(cognn_ctx*, uint64_t* v, const uint64_t* part, const uint32_t* idx, int64_t n, int64_t F) {
    for (int64_t q = 0; q < n; ++q)
        for (int64_t j = 0; j < F; ++j) v[(size_t)idx[q] * F + j] += part[q * F + j];
    return 0;
}
int cognn_ring_gemm_u64(cognn_ctx*, uint64_t* C, const uint64_t* A, const uint64_t* B, int64_t M, int64_t N, int64_t K,
                        int transA, int accumulate) {
    if (!accumulate) memset(C, 0, (size_t)M * N * 8);
    CG_PAR
    for (int64_t m = 0; m < M; ++m)
        for (int64_t k = 0; k < K; ++k) {
            const u64 a = transA ? A[k * M + m] : A[m * K + k];
            const u64* b = B + k * N;
            u64* c = C + m * N;
            for (int64_t n = 0; n < N; ++n) c[n] += a * b[n];
        }
    return 0;
}
int cognn_mask_open_u64(cognn_ctx*, uint64_t* E, const uint64_t* X, uint64_t key, int64_t rows, int64_t cols, int tr) {
    REQ(AL(E) && AL(X), "cognn_mask_open_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    const bool limb = (tr & COGNN_MASK_OPEN_LIMB) != 0;     // the A mask of a product: limb form
    tr &= ~COGNN_MASK_OPEN_LIMB;
    auto mask = [&](u64 idx) { return limb ? cognn_gemm_mask(key, idx) : cognn_prng(key, idx); };
    if (tr == 3) {                                         // X stored [cols x rows], E in logical order
        for (int64_t i = 0; i < rows * cols; ++i) E[i] = X[(i % cols) * rows + i / cols] - mask((u64)i);
        return 0;
    }
    CG_PAR
    for (int64_t i = 0; i < rows * cols; ++i) E[i] = X[i] - mask(lidx(i, rows, cols, tr));
    return 0;
}
int cognn_add_u64(cognn_ctx*, uint64_t* o, const uint64_t* a, const uint64_t* b, int64_t n) {
    REQ(AL(o) && AL(a) && AL(b), "cognn_add_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) o[i] = a[i] + b[i];
    return 0;
}
int cognn_sum_u64(cognn_ctx*, uint64_t* out, const uint64_t* const* in, int32_t count, int64_t n) {
    REQ(out && in && count >= 1 && count <= 16, "sum: bad arguments");
    for (int64_t i = 0; i < n; ++i) { u64 r = 0; for (int c = 0; c < count; ++c) r += in[c][i]; out[i] = r; }
    return 0;
}
int cognn_fanout_u64(cognn_ctx*, uint64_t* const* out, int32_t count, const uint64_t* in, int64_t n) {
    REQ(out && in && count >= 1 && count <= 16, "fanout: bad arguments");
    for (int c = 0; c < count; ++c) if (out[c] != in) memcpy(out[c], in, (size_t)n * 8);
    return 0;
}
int cognn_sub_u64(cognn_ctx*, uint64_t* o, const uint64_t* a, const uint64_t* b, int64_t n) {
    REQ(AL(o) && AL(a) && AL(b), "cognn_sub_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) o[i] = a[i] - b[i];
    return 0;
}
int cognn_dealer_gemm_c1_u64(cognn_ctx* c, uint64_t* C1, const cognn_keys* keys, int64_t M, int64_t N, int64_t K, int transA,
                             uint64_t* sa, uint64_t* sb) {
    CG_PAR
    for (int64_t i = 0; i < M * K; ++i) {
        const u64 li = lidx(i, M, K, transA);
        sa[i] = cognn_gemm_mask(keys->k[COGNN_SL_A0], li) + cognn_gemm_mask(keys->k[COGNN_SL_A1], li);   // A masks: limb form
    }
    CG_PAR
    for (int64_t i = 0; i < K * N; ++i) sb[i] = cognn_prng(keys->k[COGNN_SL_B0], (u64)i) + cognn_prng(keys->k[COGNN_SL_B1], (u64)i);
    cognn_ring_gemm_u64(c, C1, sa, sb, M, N, K, transA, 0);
    CG_PAR
    for (int64_t i = 0; i < M * N; ++i) C1[i] -= cognn_prng(keys->k[COGNN_SL_C0], (u64)i);
    return 0;
}
int cognn_dealer_gemm_c1_tn_group_u64(cognn_ctx* c, const cognn_dealer_tn_job* jobs, int32_t count) {
    for (int32_t j = 0; j < count; ++j)
        if (int rc = cognn_dealer_gemm_c1_u64(c, jobs[j].C1, &jobs[j].keys, jobs[j].M, jobs[j].N, jobs[j].K, jobs[j].transA, jobs[j].scratchA, jobs[j].scratchB)) return rc;
    return 0;
}
int cognn_dealer_gemm_c1_groupable(int64_t, int64_t) { return 0; }       // the reference backend deals job by job
int cognn_dealer_gemm_c1_group_u64(cognn_ctx* c, const cognn_dealer_job* jobs, int32_t count, int64_t N, int64_t K) {
    for (int32_t j = 0; j < count; ++j) {
        std::vector<u64> scratch((size_t)(jobs[j].M * K + K * N));
        if (int rc = cognn_dealer_gemm_c1_u64(c, jobs[j].C1, &jobs[j].keys, jobs[j].M, N, K, 0, scratch.data(), scratch.data() + jobs[j].M * K)) return rc;
    }
    return 0;
}
int cognn_ring_gemm2_u64(cognn_ctx* c, uint64_t* C, const uint64_t* A1, const uint64_t* A2, const uint64_t* B, int64_t M, int64_t N,
                         int64_t K, int transA, int accumulate) {
    if (!A2) return cognn_ring_gemm_u64(c, C, A1, B, M, N, K, transA, accumulate);
    std::vector<u64> a((size_t)M * K);
    CG_PAR
    for (int64_t i = 0; i < M * K; ++i) a[i] = A1[i] + A2[i];
    return cognn_ring_gemm_u64(c, C, a.data(), B, M, N, K, transA, accumulate);
}
int cognn_beaver_gemm_close_u64(cognn_ctx* c, uint64_t* Z, const uint64_t* E, const uint64_t* E1, const uint64_t* F, const uint64_t* c1,
                                const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA, uint64_t* scratch) {
    REQ(p == 0 || c1, "beaver_gemm_close: p=1 needs c1");
    u64* Ap = scratch; u64* Bp = scratch + M * K;
    CG_PAR
    for (int64_t i = 0; i < M * K; ++i) Ap[i] = cognn_gemm_mask(keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], lidx(i, M, K, transA));
    CG_PAR
    for (int64_t i = 0; i < K * N; ++i) Bp[i] = cognn_prng(keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], (u64)i) + (p == 1 ? F[i] : 0);
    CG_PAR
    for (int64_t i = 0; i < M * N; ++i) Z[i] = p == 0 ? cognn_prng(keys->k[COGNN_SL_C0], (u64)i) : c1[i];
    cognn_ring_gemm2_u64(c, Z, E, E1, Bp, M, N, K, transA, 1);
    return cognn_ring_gemm_u64(c, Z, Ap, F, M, N, K, transA, 1);
}
int cognn_beaver_gemm_fusable(int64_t, int64_t, int64_t, int) { return 0; }   // the reference backend has one path only
// product without the dealer share C_p (the engine never asks this backend for it - cognn_beaver_gemm_fusable is 0 - but the
// differential tests compare the HIP fast path with it)
int cognn_beaver_gemm_close_raw_u64(cognn_ctx* c, uint64_t* Z, const uint64_t* E, const uint64_t* E1, const uint64_t* F, const cognn_keys* keys,
                                    int p, int64_t M, int64_t N, int64_t K, uint64_t* scratch) {
    u64* Ap = scratch; u64* Bp = scratch + M * K;
    CG_PAR
    for (int64_t i = 0; i < M * K; ++i) Ap[i] = cognn_gemm_mask(keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], (u64)i);
    CG_PAR
    for (int64_t i = 0; i < K * N; ++i) Bp[i] = cognn_prng(keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], (u64)i) + (p == 1 ? F[i] : 0);
    cognn_ring_gemm2_u64(c, Z, E, E1, Bp, M, N, K, 0, 0);
    return cognn_ring_gemm_u64(c, Z, Ap, F, M, N, K, 0, 1);
}
int cognn_beaver_gemm_close2_u64(cognn_ctx* c, uint64_t* Z, const uint64_t* E0, const uint64_t* E1, const uint64_t* F0, const uint64_t* F1,
                                 const uint64_t* c1, const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA,
                                 uint64_t* scratch, int raw) {
    std::vector<u64> f((size_t)(K * N));
    for (int64_t i = 0; i < K * N; ++i) f[i] = F0[i] + (F1 ? F1[i] : 0);
    return raw ? cognn_beaver_gemm_close_raw_u64(c, Z, E0, E1, f.data(), keys, p, M, N, K, scratch)
               : cognn_beaver_gemm_close_u64(c, Z, E0, E1, f.data(), c1, keys, p, M, N, K, transA, scratch);
}
int cognn_set_epoch_salt(cognn_ctx*, uint64_t salt) { cognn_host_salt = salt; return 0; }
int cognn_ctx_use_private_stream(cognn_ctx*) { return 0; }
int cognn_graph_capture_begin(cognn_ctx*) { return fail("the CPU stand-in records no launch graphs"); }
int cognn_graph_capture_end(cognn_ctx*, void**) { return fail("the CPU stand-in records no launch graphs"); }
int cognn_graph_launch(cognn_ctx*, void*) { return fail("the CPU stand-in records no launch graphs"); }
int cognn_graph_destroy(cognn_ctx*, void*) { return 0; }
int64_t cognn_pair_chain_dealt_slots(int32_t, int32_t) { return 0; }     // the CPU stand-in regenerates every dealer value
int cognn_pair_chain_deal_u64(cognn_ctx*, const cognn_pair_chain*, uint64_t*) { return 0; }
int cognn_beaver_gemm_group_takes_epilogue(int64_t, int64_t, int64_t) { return 0; }
int cognn_beaver_gemm_group_is_whole_k(int64_t, int64_t, int64_t) { return 0; }   // the CPU stand-in runs product and chain separately
int64_t cognn_gemm_presplit_bytes(int64_t, int64_t) { return 0; }      // the CPU stand-in has no fragment-ordered form
int cognn_gemm_presplit_u64(cognn_ctx*, void*, const uint64_t*, const uint64_t*, int64_t, int64_t) { return 0; }
int64_t cognn_gemm_presplit_tn_bytes(int64_t, int64_t) { return 0; }
int cognn_gemm_presplit_tn_u64(cognn_ctx*, void*, const uint64_t*, const uint64_t*, uint64_t, int, int64_t, int64_t) { return 0; }
int cognn_beaver_gemm_tn_groupable(int64_t, int64_t, int64_t, int) { return 0; }    // the CPU stand-in runs the weight gradients job by job
int cognn_beaver_gemm_close_group_tn_u64(cognn_ctx*, const cognn_gemm_job*, int32_t, int64_t, int64_t, int) { return fail("not groupable on the CPU stand-in"); }
int cognn_beaver_gemm_close_group_u64(cognn_ctx* c, const cognn_gemm_job* jobs, int32_t count, int64_t N, int64_t K, int raw) {
    for (int32_t j = 0; j < count; ++j) {
        const cognn_gemm_job& J = jobs[j];
        if (cognn_beaver_gemm_close2_u64(c, J.Z, J.E0, J.E1, J.F0, J.F1, J.c1, &J.keys, J.p, J.M, N, K, 0, J.scratch, raw)) return 1;
    }
    return 0;
}
int cognn_trunc_open_add_u64(cognn_ctx*, uint64_t* c, const uint64_t* x, const uint64_t* c1, const cognn_keys* gkeys,
                             const cognn_keys* tkeys, int p, int64_t n) {
    REQ(AL(c) && AL(x) && AL(c1), "cognn_trunc_open_add_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    const cognn_opkeys tk = K(tkeys);
    for (int64_t i = lo_; i < hi_; ++i)
        c[i] = x[i] + (p == 0 ? cognn_prng(gkeys->k[COGNN_SL_C0], (u64)i) : c1[i]) + trunc_r(tk, p, (u64)i) + (p == 0 ? COGNN_TRUNC_OFFSET : 0);
    return 0;
}
int cognn_trunc_open_u64(cognn_ctx*, uint64_t* c, const uint64_t* x, uint64_t mul, const cognn_keys* keys, int p, int64_t n) {
    REQ(AL(c) && AL(x), "cognn_trunc_open_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    const cognn_opkeys k = K(keys);
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) c[i] = x[i] * mul + trunc_r(k, p, (u64)i) + (p == 0 ? COGNN_TRUNC_OFFSET : 0);
    return 0;
}
int cognn_trunc_close_u64(cognn_ctx*, uint64_t* out, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys, int p,
                          int mode, int64_t n) {
    REQ(AL(out) && AL(c0) && AL(c1), "cognn_trunc_close_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    REQ(p == 1 || (c0 && c1), "trunc_close: p=0 needs both opened values");
    const cognn_opkeys k = K(keys);
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) {
        u64 y = p == 0 ? cognn_open_hi48(c0[i], c1[i]) - (COGNN_TRUNC_OFFSET >> COGNN_FX_BITS) - trunc_rp(k, 0, (u64)i)
                       : 0ull - trunc_rp(k, 1, (u64)i);
        out[i] = mode == 1 ? out[i] - y : y;
    }
    return 0;
}
int cognn_trunc_close_open_u64(cognn_ctx* c, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                               int p, uint64_t key_open, int64_t n) {
    REQ(AL(out) && AL(E) && AL(c0) && AL(c1), "cognn_trunc_close_open_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    const int rc = cognn_trunc_close_u64(c, out, c0, c1, keys, p, 0, n);
    if (rc) return rc;
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) E[i] = out[i] - cognn_prng(key_open, (u64)i);
    return 0;
}
// both parties' closes side by side, then the opening from their sum - what the two-round form would have exchanged
int cognn_trunc_close_pub_u64(cognn_ctx* c, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                              int p, uint64_t key_open0, uint64_t key_open1, int reveal, int64_t n) {
    REQ(AL(out) && AL(E) && AL(c0) && AL(c1), "cognn_trunc_close_pub_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    REQ(E && c0 && c1, "trunc_close_pub: both opened values are needed");
    std::vector<u64> y0((size_t)n), y1((size_t)n);
    if (int rc = cognn_trunc_close_u64(c, y0.data(), c0, c1, keys, 0, 0, n)) return rc;
    if (int rc = cognn_trunc_close_u64(c, y1.data(), nullptr, nullptr, keys, 1, 0, n)) return rc;
    for (int64_t i = lo_; i < hi_; ++i) {
        const u64 e0 = reveal ? y0[(size_t)i] : y0[(size_t)i] - cognn_prng(key_open0, (u64)i);
        const u64 e1 = reveal ? y1[(size_t)i] : y1[(size_t)i] - cognn_prng(key_open1, (u64)i);
        E[i] = e0 + e1;
        if (out) out[i] = p == 0 ? y0[(size_t)i] : y1[(size_t)i];
    }
    return 0;
}
int cognn_trunc_close_pub_dealt_u64(cognn_ctx*, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const uint64_t* t,
                                    const uint64_t* rp_own, int p, int64_t n) {
    CHUNK(n)
    REQ(E && c0 && c1 && t && (!out || rp_own), "trunc_close_pub_dealt: missing tensor");
    for (int64_t i = lo_; i < hi_; ++i) {
        const u64 hi = cognn_open_hi48(c0[i], c1[i]) - (COGNN_TRUNC_OFFSET >> COGNN_FX_BITS);
        E[i] = hi - t[i];
        if (out) out[i] = (p == 0 ? hi : 0ull) - rp_own[i];
    }
    return 0;
}
int cognn_dealer_trunc_pub_u64(cognn_ctx*, uint64_t* t, uint64_t* rp0, uint64_t* rp1, const cognn_keys* keys, uint64_t key_open0,
                               uint64_t key_open1, int reveal, int64_t n) {
    const cognn_opkeys k = K(keys);
    for (int64_t i = 0; i < n; ++i) {
        const u64 rp = (cognn_prng(k.k[COGNN_SL_R], (u64)i) & COGNN_TRUNC_MASK) >> COGNN_FX_BITS;
        t[i] = reveal ? rp : rp + cognn_prng(key_open0, (u64)i) + cognn_prng(key_open1, (u64)i);
        if (rp0) rp0[i] = trunc_rp(k, 0, (u64)i);
        if (rp1) rp1[i] = rp - trunc_rp(k, 0, (u64)i);
    }
    return 0;
}
int cognn_rowscale_open_u64(cognn_ctx*, uint64_t* E, uint64_t* G, const uint64_t* V, const uint64_t* s, const cognn_keys* keys,
                            int p, int64_t rows, int64_t F) {
    REQ(AL(E) && AL(G) && AL(V) && AL(s), "cognn_rowscale_open_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    if (E) for (int64_t i = 0; i < rows * F; ++i) E[i] = V[i] - cognn_prng(keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], (u64)i);
    CG_PAR
    for (int64_t r = 0; r < rows; ++r) G[r] = s[r] - cognn_prng(keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], (u64)r);
    return 0;
}
int cognn_rowscale_close_u64(cognn_ctx*, uint64_t* c, const uint64_t* E, const uint64_t* E1, const uint64_t* G, const uint64_t* G1,
                             const cognn_keys* keys, const cognn_keys* tkeys, int p, int64_t rows, int64_t F) {
    REQ(AL(c) && AL(E) && AL(E1), "cognn_rowscale_close_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(rows * F)
    const cognn_opkeys k = K(keys), tk = K(tkeys);
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) {
        const u64 row = (u64)(i / F);
        const u64 e = E[i] + (E1 ? E1[i] : 0), g = G[row] + (G1 ? G1[row] : 0);
        c[i] = beaver_mul(k, p, e, g, (u64)i, row) + trunc_r(tk, p, (u64)i) + (p == 0 ? COGNN_TRUNC_OFFSET : 0);
    }
    return 0;
}
int cognn_relu_open_u64(cognn_ctx*, uint64_t* E, uint64_t* G, const uint64_t* z, const cognn_keys* keys, int p, int64_t n) {
    REQ(AL(E) && AL(G) && AL(z), "cognn_relu_open_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) {
        const u64 t0 = cognn_prng(keys->k[COGNN_SL_T0], (u64)i);
        const u64 tp = p == 0 ? t0 : ((cognn_prng(keys->k[COGNN_SL_T], (u64)i) & 0xFFFFFull) | COGNN_RELU_T_MIN) - t0;
        E[i] = z[i] - cognn_prng(keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], (u64)i);
        if (G) G[i] = tp - cognn_prng(keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], (u64)i);
    }
    return 0;
}
int cognn_relu_mul_u64(cognn_ctx*, uint64_t* w, const uint64_t* E, const uint64_t* E1, const uint64_t* G, const uint64_t* G1,
                       const cognn_keys* keys, int p, int64_t n) {
    REQ(AL(w) && AL(E) && AL(G) && AL(E1) && AL(G1), "cognn_relu_mul_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    const cognn_opkeys k = K(keys);
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) {
        // G == NULL: dealer-published g = t - (b0 + b1)
        const u64 g = G ? G[i] + (G1 ? G1[i] : 0)
                        : ((cognn_prng(keys->k[COGNN_SL_T], (u64)i) & 0xFFFFFull) | COGNN_RELU_T_MIN) - cognn_prng(keys->k[COGNN_SL_B0], (u64)i) -
                              cognn_prng(keys->k[COGNN_SL_B1], (u64)i);
        w[i] = beaver_mul(k, p, E[i] + (E1 ? E1[i] : 0), g, (u64)i, (u64)i);
    }
    return 0;
}
int cognn_relu_close_u64(cognn_ctx*, uint64_t* h, uint8_t* mask, const uint64_t* z, const uint64_t* w0, const uint64_t* w1, int64_t n) {
    REQ(AL(h) && AL(z) && AL(w0) && AL(w1), "cognn_relu_close_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) {
        const bool pos = cognn_relu_positive(w0[i], w1[i]);
        h[i] = pos ? z[i] : 0;
        if (mask) mask[i] = pos;
    }
    return 0;
}
int cognn_relu_close_open_u64(cognn_ctx* c, uint64_t* h, uint64_t* E, uint8_t* mask, const uint64_t* z, const uint64_t* w0, const uint64_t* w1,
                              uint64_t key_open, int64_t n) {
    REQ(AL(h) && AL(E) && AL(z) && AL(w0) && AL(w1), "cognn_relu_close_open_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    cognn_relu_close_u64(c, h, mask, z, w0, w1, n);
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) E[i] = h[i] - cognn_gemm_mask(key_open, (u64)i);   // the next product's A mask
    return 0;
}
int cognn_mask_select_u64(cognn_ctx*, uint64_t* out, const uint64_t* in, const uint8_t* mask, int64_t n) {
    REQ(AL(out) && AL(in), "cognn_mask_select_u64: misaligned tensor (the HIP kernels move 16-byte element pairs)");
    CHUNK(n)
    CG_PAR
    for (int64_t i = lo_; i < hi_; ++i) out[i] = mask[i] ? in[i] : 0;
    return 0;
}
int cognn_softmax_u64(cognn_ctx*, uint64_t* p_out, uint64_t* d_out, uint64_t* pfx_out, const uint64_t* z0, const uint64_t* z1,
                      const int32_t* labels, const cognn_keys* keys, int p, int64_t rows, int64_t L, int64_t train_rows) {
    REQ(p == 1 || (z0 && z1 && labels), "softmax: owner side needs z0, z1, labels");
    CG_PAR
    for (int64_t r = 0; r < rows; ++r) {
        const bool keep = r < train_rows;
        long long m = 0, S = 0;
        if (p == 0) {
            m = (long long)(z0[r * L] + z1[r * L]);
            for (int64_t j = 1; j < L; ++j) { long long v = (long long)(z0[r * L + j] + z1[r * L + j]); if (v > m) m = v; }
            for (int64_t j = 0; j < L; ++j) S += cognn_exp_neg_q30(m - (long long)(z0[r * L + j] + z1[r * L + j]));
        }
        for (int64_t j = 0; j < L; ++j) {
            const u64 rho = cognn_prng(keys->k[COGNN_SL_RHO], (u64)(r * L + j));
            if (p == 1) {
                if (p_out) p_out[r * L + j] = rho;
                d_out[r * L + j] = keep ? rho : 0;
                continue;
            }
            const long long e = cognn_exp_neg_q30(m - (long long)(z0[r * L + j] + z1[r * L + j]));
            const u64 pf = (u64)(((e << 16) + (S >> 1)) / S);
            const u64 p0 = pf - rho;
            if (pfx_out) pfx_out[r * L + j] = pf;
            if (p_out) p_out[r * L + j] = p0;
            d_out[r * L + j] = keep ? p0 - (j == labels[r] ? COGNN_FX_ONE : 0) : 0;
        }
    }
    return 0;
}
int cognn_metrics_q16(cognn_ctx*, const uint64_t* pfx, const int32_t* labels, const uint8_t* border, int64_t rows, int64_t L,
                      int64_t train_rows, int64_t val_rows, int64_t* c, double* loss) {
    for (int i = 0; i < 6; ++i) c[i] = 0;
    double ls = 0;
    for (int64_t r = 0; r < rows; ++r) {                    // serial: counters and an order-dependent float sum
        int best = 0;
        for (int64_t j = 1; j < L; ++j) if (pfx[r * L + j] > pfx[r * L + best]) best = (int)j;
        const bool ok = best == labels[r], b = border && border[r], tr = r < train_rows, te = r >= train_rows + val_rows;
        if (ok) { c[0]++; if (tr) c[1]++; if (tr && b) c[2]++; if (te) c[3]++; if (te && b) c[4]++; }
        double pl = (double)pfx[r * L + labels[r]] / (double)COGNN_FX_ONE;
        if (pl == 0.0) pl = 0.001;
        ls += -log(pl);
    }
    *loss = ls;
    return 0;
}
// the batched prediction layer = the per-side softmax followed by the metrics on its revealed probabilities
int cognn_softmax_jobs_u64(cognn_ctx* ctx, const cognn_softmax_job* jobs, int32_t count, int64_t L) {
    for (int32_t c = 0; c < count; ++c) {
        const cognn_softmax_job& s = jobs[c];
        if (s.p == 1) {
            if (int rc = cognn_softmax_u64(ctx, nullptr, s.d_out, nullptr, nullptr, nullptr, nullptr, &s.keys, 1, s.rows, L, s.train_rows)) return rc;
            continue;
        }
        std::vector<u64> pfx((size_t)std::max<int64_t>(s.rows * L, 1)), zero;
        if (!s.z1) zero.assign((size_t)std::max<int64_t>(s.rows * L, 1), 0);      // z0 is the revealed z itself
        if (int rc = cognn_softmax_u64(ctx, nullptr, s.d_out, pfx.data(), s.z0, s.z1 ? s.z1 : zero.data(), s.labels, &s.keys, 0, s.rows, L, s.train_rows)) return rc;
        if (int rc = cognn_metrics_q16(ctx, pfx.data(), s.labels, s.border, s.rows, L, s.train_rows, s.val_rows, s.counts6, s.loss)) return rc;
    }
    return 0;
}
// The pair chain of the HIP library fuses both sides' steps into one kernel; this reference runs the SAME steps through the
// per-side entry points above, one after the other, with the exchanged values in ordinary buffers - an independent statement
// of what the fused kernel has to produce.
int cognn_pair_chain_u64(cognn_ctx* ctx, const cognn_pair_chain* chains, int32_t count) {
    for (int32_t ci = 0; ci < count; ++ci) {
        const cognn_pair_chain& s = chains[ci];
        const int64_t n = s.rows * s.F;
        if (n <= 0) continue;
        REQ(s.x[0] && s.x[1], "pair_chain: null input");
        std::vector<u64> cur[2], o0[2], o1[2];
        for (int p = 0; p < 2; ++p) { cur[p].assign(s.x[p], s.x[p] + n); o0[p].resize((size_t)n); o1[p].resize((size_t)n); }
        if (s.flags & COGNN_PC_CLEAR_INPUT)                 // the product buffers go back clean
            for (int p = 0; p < 2; ++p) memset(const_cast<uint64_t*>(s.x[p]), 0, (size_t)n * 8);
        auto select = [&]() -> int {                        // cognn_mask_select_u64 on both sides' current values
            for (int p = 0; p < 2; ++p) {
                std::vector<u64> sel((size_t)n);
                if (int rc = cognn_mask_select_u64(ctx, sel.data(), cur[p].data(), s.mask_in, n)) return rc;
                cur[p].swap(sel);
            }
            return 0;
        };
        const bool mask_mid = (s.flags & COGNN_PC_MASK_AFTER_TRUNC) != 0;
        REQ(!mask_mid || ((s.flags & COGNN_PC_TRUNC_IN) && s.mask_in), "pair_chain: COGNN_PC_MASK_AFTER_TRUNC wants COGNN_PC_TRUNC_IN and mask_in");
        if (s.mask_in && !mask_mid) if (int rc = select()) return rc;
        bool opened = (s.flags & COGNN_PC_INPUT_OPENED) != 0;
        if (s.flags & COGNN_PC_TRUNC_IN) {
            for (int p = 0; p < 2; ++p) {
                int rc = (s.flags & COGNN_PC_NO_C) ? cognn_trunc_open_u64(ctx, o0[p].data(), cur[p].data(), 1, &s.trunc_in_keys, p, n)
                                                 : cognn_trunc_open_add_u64(ctx, o0[p].data(), cur[p].data(), s.c1, &s.gemm_keys, &s.trunc_in_keys, p, n);
                if (rc) return rc;
            }
            for (int p = 0; p < 2; ++p)
                if (int rc = cognn_trunc_close_u64(ctx, cur[p].data(), p == 0 ? o0[0].data() : nullptr, p == 0 ? o0[1].data() : nullptr,
                                                   &s.trunc_in_keys, p, 0, n)) return rc;
            if (mask_mid) if (int rc = select()) return rc;
        }
        if (s.flags & COGNN_PC_SCALE) {
            REQ(s.scale[0] && s.scale[1], "pair_chain: null scale");
            std::vector<u64> g[2];
            for (int p = 0; p < 2; ++p) {
                g[p].resize((size_t)s.rows);
                if (int rc = cognn_rowscale_open_u64(ctx, opened ? nullptr : o0[p].data(), g[p].data(), cur[p].data(), s.scale[p], &s.scale_keys, p,
                                                     s.rows, s.F)) return rc;
                if (opened) o0[p] = cur[p];
            }
            for (int p = 0; p < 2; ++p)
                if (int rc = cognn_rowscale_close_u64(ctx, o1[p].data(), o0[p].data(), o0[1 - p].data(), g[p].data(), g[1 - p].data(), &s.scale_keys,
                                                      &s.scale_trunc_keys, p, s.rows, s.F)) return rc;
            for (int p = 0; p < 2; ++p)
                if (int rc = cognn_trunc_close_u64(ctx, cur[p].data(), p == 0 ? o1[0].data() : nullptr, p == 0 ? o1[1].data() : nullptr,
                                                   &s.scale_trunc_keys, p, 0, n)) return rc;
        }
        if (s.flags & COGNN_PC_RELU) {
            std::vector<u64> h[2];
            for (int p = 0; p < 2; ++p)
                if (int rc = cognn_relu_open_u64(ctx, o0[p].data(), nullptr, cur[p].data(), &s.relu_keys, p, n)) return rc;
            for (int p = 0; p < 2; ++p)
                if (int rc = cognn_relu_mul_u64(ctx, o1[p].data(), o0[p].data(), o0[1 - p].data(), nullptr, nullptr, &s.relu_keys, p, n)) return rc;
            for (int p = 0; p < 2; ++p) {
                h[p].resize((size_t)n);
                if (int rc = cognn_relu_close_u64(ctx, h[p].data(), p == 0 ? s.mask : nullptr, cur[p].data(), o1[p].data(), o1[1 - p].data(), n)) return rc;
            }
            for (int p = 0; p < 2; ++p) cur[p].swap(h[p]);
        }
        const bool limb = (s.flags & COGNN_PC_OPEN_LIMB) != 0;    // the opening of a product's left operand: limb-form masks
        auto omask = [&](int p, int64_t i) { return limb ? cognn_gemm_mask(s.open_key[p], (u64)i) : cognn_prng(s.open_key[p], (u64)i); };
        if (s.flags & COGNN_PC_OPEN_SUM) {                  // what each party holds after exchanging the two openings
            REQ(!s.open[1], "pair_chain: COGNN_PC_OPEN_SUM writes open[0] only");
            if (s.open[0])
                for (int64_t i = 0; i < n; ++i)
                    s.open[0][i] = (cur[0][(size_t)i] - omask(0, i)) + (cur[1][(size_t)i] - omask(1, i));
        }
        for (int p = 0; p < 2; ++p) {
            if (s.open[p] && !(s.flags & COGNN_PC_OPEN_SUM))
                for (int64_t i = 0; i < n; ++i) s.open[p][i] = cur[p][(size_t)i] - omask(p, i);
            if (s.out[p]) memcpy(s.out[p], cur[p].data(), (size_t)n * 8);
        }
    }
    return 0;
}
// the fused gather = the plain gather of both sides' row segments into temporaries, then the pair chain on them
// the per-side sequence the fused kernel replaces, call by call
int cognn_pair_weight_update_u64(cognn_ctx* ctx, const cognn_pair_wupdate* jobs, int32_t count, const cognn_keys* avg_keys, uint64_t avg_mul,
                                 int32_t average) {
    REQ(!average || (count >= 1 && count <= 16), "pair_weight_update: the averaging form takes 1..16 pairs");
    for (int32_t j = 0; j < count; ++j) {
        const cognn_pair_wupdate& s = jobs[j];
        const int64_t n = s.n;
        if (n == 0) continue;
        REQ(s.z[0] && s.z[1] && s.W[0] && s.W[1], "pair_weight_update: null tensor");
        std::vector<u64> c0(n), c1(n), d0(n), d1(n);
        if (s.flags & COGNN_PC_NO_C) {
            cognn_trunc_open_u64(ctx, c0.data(), s.z[0], 1, &s.trunc_keys[0], 0, n);
            cognn_trunc_open_u64(ctx, c1.data(), s.z[1], 1, &s.trunc_keys[0], 1, n);
        } else {
            REQ(s.c1, "pair_weight_update: side 1's product share is missing");
            cognn_trunc_open_add_u64(ctx, c0.data(), s.z[0], nullptr, &s.gemm_keys, &s.trunc_keys[0], 0, n);
            cognn_trunc_open_add_u64(ctx, c1.data(), s.z[1], s.c1, &s.gemm_keys, &s.trunc_keys[0], 1, n);
        }
        cognn_trunc_close_u64(ctx, d0.data(), c0.data(), c1.data(), &s.trunc_keys[0], 0, 0, n);
        cognn_trunc_close_u64(ctx, d1.data(), nullptr, nullptr, &s.trunc_keys[0], 1, 0, n);
        for (int t = 1; t <= 2; ++t) {
            cognn_trunc_open_u64(ctx, c0.data(), d0.data(), s.mul[t - 1], &s.trunc_keys[t], 0, n);
            cognn_trunc_open_u64(ctx, c1.data(), d1.data(), s.mul[t - 1], &s.trunc_keys[t], 1, n);
            const int mode = t == 2 ? 1 : 0;
            cognn_trunc_close_u64(ctx, mode ? s.W[0] : d0.data(), c0.data(), c1.data(), &s.trunc_keys[t], 0, mode, n);
            cognn_trunc_close_u64(ctx, mode ? s.W[1] : d1.data(), nullptr, nullptr, &s.trunc_keys[t], 1, mode, n);
        }
        if (s.mul[2]) {
            cognn_trunc_open_u64(ctx, c0.data(), s.W[0], s.mul[2], &s.trunc_keys[3], 0, n);
            cognn_trunc_open_u64(ctx, c1.data(), s.W[1], s.mul[2], &s.trunc_keys[3], 1, n);
            cognn_trunc_close_u64(ctx, s.W[0], c0.data(), c1.data(), &s.trunc_keys[3], 0, 0, n);
            cognn_trunc_close_u64(ctx, s.W[1], nullptr, nullptr, &s.trunc_keys[3], 1, 0, n);
        }
        if (s.flags & COGNN_WU_CLEAR_Z)
            for (int p = 0; p < 2; ++p) memset(const_cast<uint64_t*>(s.z[p]), 0, (size_t)n * 8);
    }
    if (average) {                                         // cognn_sum_u64 x 2, the 1/k truncation, cognn_fanout_u64 x 2
        const int64_t n = jobs[0].n;
        std::vector<u64> s0(n, 0), s1(n, 0), c0(n), c1(n);
        for (int32_t j = 0; j < count; ++j) {
            REQ(jobs[j].n == n, "pair_weight_update: the averaged matrices must have one size");
            const int sw = (jobs[j].flags & COGNN_WU_SWAP) ? 1 : 0;
            for (int64_t i = 0; i < n; ++i) { s0[i] += jobs[j].W[sw][i]; s1[i] += jobs[j].W[1 - sw][i]; }
        }
        if (avg_mul) {
            REQ(avg_keys, "pair_weight_update: the scale's keys are missing");
            cognn_trunc_open_u64(ctx, c0.data(), s0.data(), avg_mul, avg_keys, 0, n);
            cognn_trunc_open_u64(ctx, c1.data(), s1.data(), avg_mul, avg_keys, 1, n);
            cognn_trunc_close_u64(ctx, s0.data(), c0.data(), c1.data(), avg_keys, 0, 0, n);
            cognn_trunc_close_u64(ctx, s1.data(), nullptr, nullptr, avg_keys, 1, 0, n);
        }
        for (int32_t j = 0; j < count; ++j) {
            const int sw = (jobs[j].flags & COGNN_WU_SWAP) ? 1 : 0;
            memcpy(jobs[j].W[sw], s0.data(), (size_t)n * 8); memcpy(jobs[j].W[1 - sw], s1.data(), (size_t)n * 8);
        }
    }
    return 0;
}
// original-gcn message passing for one destination party: the per-edge sequence (two row scales with truncation per edge, then
// the sums), both share-holders side by side
static void cpu_scale_trunc_pair(const cognn_keys& sk, const cognn_keys& tk, u64 s0, u64 s1, u64 row, u64 idx, u64& v0, u64& v1) {
    const cognn_opkeys k = K(&sk), t = K(&tk);
    const u64 b0 = cognn_prng(k.k[COGNN_SL_B0], row), b1 = cognn_prng(k.k[COGNN_SL_B1], row);
    const u64 g = (s0 - b0) + (s1 - b1);
    const u64 a0 = cognn_prng(k.k[COGNN_SL_A0], idx), a1 = cognn_prng(k.k[COGNN_SL_A1], idx), c0m = cognn_prng(k.k[COGNN_SL_C0], idx);
    const u64 c1m = (a0 + a1) * (b0 + b1) - c0m;
    const u64 e = (v0 - a0) + (v1 - a1);
    const u64 z0 = e * b0 + a0 * g + c0m, z1 = e * g + e * b1 + a1 * g + c1m;
    const u64 c0 = z0 + trunc_r(t, 0, idx) + COGNN_TRUNC_OFFSET, c1 = z1 + trunc_r(t, 1, idx);
    v0 = cognn_open_hi48(c0, c1) - (COGNN_TRUNC_OFFSET >> COGNN_FX_BITS) - trunc_rp(t, 0, idx);
    v1 = 0ull - trunc_rp(t, 1, idx);
}
int cognn_scatter_gather_original_u64(cognn_ctx*, uint64_t* outA, uint64_t* outB, const uint64_t* selfA, const uint64_t* selfB,
                                      const uint64_t* self_scale0, const uint64_t* self_scale1, const cognn_keys* self_scale_keys,
                                      const cognn_keys* self_trunc_keys, int64_t rows, int64_t F, const uint32_t* rowptr,
                                      const uint32_t* ent_src, const uint32_t* ent_pair, const uint32_t* ent_q,
                                      const cognn_scatter_pair* pairs, int32_t npairs) {
    REQ(outA && outB && selfA && selfB && rowptr && npairs >= 0 && npairs <= 16, "scatter_gather_original: bad arguments");
    CG_PAR
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t j = 0; j < F; ++j) {
            const u64 own = (u64)(r * F + j);
            u64 a = selfA[own], c = selfB[own];
            if (self_scale0) cpu_scale_trunc_pair(*self_scale_keys, *self_trunc_keys, self_scale0[r], self_scale1[r], (u64)r, own, a, c);
            for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                const cognn_scatter_pair& P = pairs[ent_pair[e]];
                const u64 q = ent_q[e], src = (u64)ent_src[e] * (u64)F + (u64)j, idx = q * (u64)F + (u64)j;
                u64 u0 = P.srcA[src], u1 = P.srcB[src];
                cpu_scale_trunc_pair(P.scale0, P.trunc0, P.n0[q], 0, q, idx, u0, u1);
                cpu_scale_trunc_pair(P.scale1, P.trunc1, P.n1_from_server ? 0 : P.n1[q], P.n1_from_server ? P.n1[q] : 0, q, idx, u0, u1);
                if (P.crossed) { a += u1; c += u0; } else { a += u0; c += u1; }
            }
            outA[own] = a; outB[own] = c;
        }
    return 0;
}
int cognn_gather_pair_chain_u64(cognn_ctx* ctx, const uint64_t* table, const uint32_t* rowptr, const uint32_t* col, int64_t F,
                                const cognn_gather_pair* pairs, int32_t count) {
    return cognn_gather_pair_chain_base_u64(ctx, table, nullptr, rowptr, col, F, pairs, count);
}
int cognn_gather_pair_chain_base_u64(cognn_ctx* ctx, const uint64_t* table, const uint64_t* base, const uint32_t* rowptr, const uint32_t* col, int64_t F,
                                     const cognn_gather_pair* pairs, int32_t count) {
    for (int32_t c = 0; c < count; ++c) {
        const cognn_gather_pair& p = pairs[c];
        const int64_t rows = p.chain.rows;
        if (rows <= 0) continue;
        std::vector<u64> v[2];
        const int64_t r0[2] = {p.a_row0, p.b_row0};
        for (int sd = 0; sd < 2; ++sd) {
            v[sd].resize((size_t)(rows * F));
            // rowptr is indexed by table row: shift it so that row 0 of the temporary is the segment's first row
            if (int rc = cognn_gather_csr_u64(ctx, v[sd].data(), (base ? base : table) + r0[sd] * F, table, rowptr + r0[sd], col, rows, F)) return rc;
        }
        cognn_pair_chain ch = p.chain;
        ch.x[0] = v[0].data(); ch.x[1] = v[1].data(); ch.F = F;
        std::vector<u64> z[2];
        if (p.softmax[0]) {                                  // the prediction layer as the plain sequence: the chain's outputs, then the two jobs
            REQ(p.softmax[1], "gather_pair_chain: softmax wants both jobs");
            for (int sd = 0; sd < 2; ++sd) if (!ch.out[sd]) { z[sd].resize((size_t)(rows * F)); ch.out[sd] = z[sd].data(); }
        }
        if (int rc = cognn_pair_chain_u64(ctx, &ch, 1)) return rc;
        if (p.softmax[0]) {
            cognn_softmax_job jobs[2] = {*p.softmax[0], *p.softmax[1]};
            jobs[0].z0 = ch.out[0]; jobs[0].z1 = ch.out[1];
            if (int rc = cognn_softmax_jobs_u64(ctx, jobs, 2, F)) return rc;
        }
    }
    return 0;
}
int cognn_gather_pair_chain_takes_softmax(int64_t F) { return (F >= 1 && F <= 64) ? 1 : 0; }
// the device index construction, as plain loops (count, scan, fill in edge order)
int cognn_graph_build_colocated(cognn_ctx*, int64_t V, int64_t E, int32_t undirected, const int64_t* src, const int64_t* dst, const int32_t* tid,
                                const uint32_t* row_of_vid, const int64_t* a_off, const int64_t* b_off, int64_t table_rows, uint32_t* rowptr,
                                uint32_t* col, uint32_t* true_in_deg, uint32_t* in_deg, uint32_t* out_deg, uint8_t* is_border, uint8_t* self_dummy,
                                uint32_t* scratch) {
    uint32_t* local_in = scratch;
    std::vector<uint32_t> cursor((size_t)table_rows, 0);
    memset(scratch, 0, (size_t)V * 4);
    memset(true_in_deg, 0, (size_t)V * 4); memset(out_deg, 0, (size_t)V * 4); memset(is_border, 0, (size_t)V);
    memset(rowptr, 0, (size_t)(table_rows + 1) * 4);
    const int64_t total = undirected ? 2 * E : E;
    auto edge = [&](int64_t i, int64_t& s, int64_t& d) {
        const int64_t e = undirected ? i >> 1 : i;
        s = src[e]; d = dst[e];
        if (undirected && (i & 1)) std::swap(s, d);
    };
    auto A = [&](int64_t v) { return (uint32_t)(a_off[tid[v]] + row_of_vid[v]); };
    auto B = [&](int64_t v) { return (uint32_t)(b_off[tid[v]] + row_of_vid[v]); };
    for (int64_t i = 0; i < total; ++i) {
        int64_t s, d;
        edge(i, s, d);
        REQ(s >= 0 && s < V && d >= 0 && d < V, "edge list: vertex id out of range");
        out_deg[s]++; true_in_deg[d]++;
        if (tid[s] == tid[d]) local_in[d]++; else is_border[s] = 1;
        rowptr[A(d) + 1]++; rowptr[B(d) + 1]++;
    }
    for (int64_t r = 0; r < table_rows; ++r) rowptr[r + 1] += rowptr[r];
    for (int64_t i = 0; i < total; ++i) {
        int64_t s, d;
        edge(i, s, d);
        const bool same = tid[s] == tid[d];
        col[rowptr[A(d)] + cursor[A(d)]++] = same ? A(s) : B(s);
        col[rowptr[B(d)] + cursor[B(d)]++] = same ? B(s) : A(s);
    }
    for (int64_t v = 0; v < V; ++v) {
        const bool dummy = local_in[v] == 0;
        self_dummy[v] = dummy; in_deg[v] = true_in_deg[v] + dummy; out_deg[v] += dummy;
    }
    return 0;
}
int cognn_transpose_u64(cognn_ctx*, uint64_t* out, const uint64_t* in, int64_t rows, int64_t cols) {
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) out[c * rows + r] = in[r * cols + c];
    return 0;
}
int cognn_timer_begin(cognn_ctx* c, int kind) { c->open[kind].push_back(std::chrono::high_resolution_clock::now()); return 0; }
int cognn_timer_end(cognn_ctx* c, int kind) {
    REQ(!c->open[kind].empty(), "timer_end: no open timer");
    c->total_ms[kind] += std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - c->open[kind].back()).count();
    c->open[kind].pop_back();
    c->count[kind]++;
    return 0;
}
int cognn_timer_read(cognn_ctx* c, int kind, int64_t* n, double* ms) { *n = c->count[kind]; *ms = c->total_ms[kind]; return 0; }
int cognn_timer_reset(cognn_ctx* c) {
    for (int k = 0; k < 8; ++k) { c->open[k].clear(); c->total_ms[k] = 0; c->count[k] = 0; }
    return 0;
}

const cognn_backend* cognn_default_backend(void) {
    static const cognn_backend be = {
#define X(name) &name,
        COGNN_BACKEND_FUNCS(X)
#undef X
    };
    return &be;
}

}  // extern "C"
