// Element-wise share arithmetic on flat uint64 tensors (HBM-bound; dealer randomness is
// evaluated in registers from counter-PRNG streams, so it costs no HBM traffic).
// Stand-ins for the local halves of sci::twoPartyGCN{VectorScale,Relu,MatrixScale,ApplyGradient,
// CondVectorAddition,ForwardNNPredictionWithoutWeight} (gcn.h:247,476,549,578,676,678) and
// CryptoUtil::intoShares (gcn.h:70); definitions in DESIGN.md §3.
#include "common.h"
#include <algorithm>
#include <cstring>
#include "../../include/cognn_hip.h"
#include "pair_chain.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void ld2(const u64* p, int64_t i, int w, u64 v[2]) {
    if (w == 2) { u64x2 t = *reinterpret_cast<const u64x2*>(p + i); v[0] = t.x; v[1] = t.y; }
    else { v[0] = p[i]; v[1] = 0; }
}
__device__ __forceinline__ void st2(u64* p, int64_t i, int w, const u64 v[2]) {
    if (w == 2) { u64x2 t; t.x = v[0]; t.y = v[1]; *reinterpret_cast<u64x2*>(p + i) = t; }
    else p[i] = v[0];
}

// dealer streams ------------------------------------------------------------------------------
__device__ __forceinline__ u64 trunc_r(const cognn_opkeys& k, int p, u64 idx) {
    u64 r0 = cognn_prng(k.k[COGNN_SL_R0], idx);
    if (p == 0) return r0;
    return (cognn_prng(k.k[COGNN_SL_R], idx) & COGNN_TRUNC_MASK) - r0;
}
__device__ __forceinline__ u64 trunc_rp(const cognn_opkeys& k, int p, u64 idx) {
    u64 rp0 = cognn_prng(k.k[COGNN_SL_RP0], idx);
    if (p == 0) return rp0;
    return ((cognn_prng(k.k[COGNN_SL_R], idx) & COGNN_TRUNC_MASK) >> COGNN_FX_BITS) - rp0;
}
// z_p = p*e*g + e*b_p + a_p*g + c_p for an element-wise Beaver product with masks a (idx) and b (bidx)
// (b0, b1 = the b masks of both parties, evaluated by the caller: one pair serves a whole row of a row scale; b1 unused for p == 0)
__device__ __forceinline__ u64 beaver_mul_b(const cognn_opkeys& k, int p, u64 e, u64 g, u64 idx, u64 b0, u64 b1) {
    u64 a0 = cognn_prng(k.k[COGNN_SL_A0], idx);
    u64 c0 = cognn_prng(k.k[COGNN_SL_C0], idx);
    if (p == 0) return e * b0 + a0 * g + c0;
    u64 a1 = cognn_prng(k.k[COGNN_SL_A1], idx);
    u64 c1 = (a0 + a1) * (b0 + b1) - c0;
    return e * g + e * b1 + a1 * g + c1;
}
__device__ __forceinline__ u64 beaver_mul(const cognn_opkeys& k, int p, u64 e, u64 g, u64 idx, u64 bidx) {
    return beaver_mul_b(k, p, e, g, idx, cognn_prng(k.k[COGNN_SL_B0], bidx), p == 1 ? cognn_prng(k.k[COGNN_SL_B1], bidx) : 0ull);
}

// generic pair launcher: thread t handles flat elements 2t, 2t+1
// (i0: first element of the chunk window, even; n: one past its last)
template <class F>
__global__ __launch_bounds__(kThreads) void ew_kernel(int64_t i0, int64_t n, F f) {
    int64_t i = i0 + 2 * ((int64_t)blockIdx.x * kThreads + threadIdx.x);
    if (i + 1 < n) f(i, 2);
    else if (i < n) f(i, 1);
}
// Batched form: the same functor type for several independent tensors (the sides of one protocol phase) in ONE launch.
// The descriptors travel by value in the kernel arguments; a workgroup finds its segment by a short uniform scan.
constexpr int kBatchMax = 16;
template <class F>
struct EwBatch {
    F f[kBatchMax];
    int64_t i0[kBatchMax], n[kBatchMax];       // element range of each tensor (the whole tensor unless a chunk window is set)
    unsigned blk_end[kBatchMax];               // exclusive prefix of workgroup counts
    int count;
};
template <class F>
__global__ __launch_bounds__(kThreads) void ew_batch_kernel(EwBatch<F> b) {
    const unsigned blk = blockIdx.x;
    int seg = 0;
    while (seg < b.count - 1 && blk >= b.blk_end[seg]) ++seg;
    const unsigned blk0 = seg ? b.blk_end[seg - 1] : 0u;
    const int64_t n = b.n[seg];
    const int64_t i = b.i0[seg] + 2 * ((int64_t)(blk - blk0) * kThreads + threadIdx.x);
    if (i + 1 < n) b.f[seg](i, 2);
    else if (i < n) b.f[seg](i, 1);
}
template <class F>
int flush_ew(cognn_ctx* ctx) {
    static_assert(sizeof(EwBatch<F>) <= sizeof(ctx->pending.storage), "batch descriptor does not fit the pending buffer");
    EwBatch<F>* b = reinterpret_cast<EwBatch<F>*>(ctx->pending.storage);
    ctx->pending.flush = nullptr;
    if (b->count <= 0) return 0;
    const unsigned blocks = b->blk_end[b->count - 1];
    if (b->count == 1) hipLaunchKernelGGL(ew_kernel<F>, dim3(blocks), dim3(kThreads), 0, ctx->stream, b->i0[0], b->n[0], b->f[0]);
    else hipLaunchKernelGGL(ew_batch_kernel<F>, dim3(blocks), dim3(kThreads), 0, ctx->stream, *b);
    b->count = 0;
    CG_LAUNCH_CHECK();
    return 0;
}
template <class F>
int launch_ew(cognn_ctx* ctx, int64_t n_all, F f) {
    int64_t i0, n;
    cognn_chunk_range(n_all, ctx->chunk_c, ctx->chunk_C, &i0, &n);
    if (n <= i0) return 0;
    const int64_t pairs = (n - i0 + 1) / 2;
    const unsigned blocks = (unsigned)((pairs + kThreads - 1) / kThreads);
    if (ctx->batch_depth > 0) {
        int rc;
        if (ctx->pending.flush && ctx->pending.flush != &flush_ew<F>) { if ((rc = ctx->pending.flush(ctx))) return rc; }
        EwBatch<F>* b = reinterpret_cast<EwBatch<F>*>(ctx->pending.storage);
        if (!ctx->pending.flush) { b->count = 0; ctx->pending.flush = &flush_ew<F>; }
        const unsigned base = b->count ? b->blk_end[b->count - 1] : 0u;
        if ((uint64_t)base + blocks > 0x7fffffffull) { if ((rc = flush_ew<F>(ctx))) return rc; return launch_ew(ctx, n_all, f); }
        b->f[b->count] = f; b->i0[b->count] = i0; b->n[b->count] = n; b->blk_end[b->count] = base + blocks;
        if (++b->count == kBatchMax) return flush_ew<F>(ctx);
        return 0;
    }
    int rc;
    if ((rc = cg_flush_pending(ctx))) return rc;
    hipLaunchKernelGGL(ew_kernel<F>, dim3(blocks), dim3(kThreads), 0, ctx->stream, i0, n, f);
    CG_LAUNCH_CHECK();
    return 0;
}

struct PrngFill {
    u64* out; u64 key; int limb;
    __device__ void operator()(int64_t i, int w) const {
        u64 v[2] = {cognn_prng(key, (u64)i), cognn_prng(key, (u64)i + 1)};
        if (limb) { v[0] = cognn_limb_value(v[0]); v[1] = cognn_limb_value(v[1]); }
        st2(out, i, w, v);
    }
};
struct ShareSplit {
    const u64* fx; u64 key; u64* s0; u64* s1;
    __device__ void operator()(int64_t i, int w) const {
        u64 x[2], a[2], b[2];
        ld2(fx, i, w, x);
        for (int j = 0; j < 2; ++j) { b[j] = cognn_prng(key, (u64)(i + j)); a[j] = x[j] - b[j]; }
        if (s0) st2(s0, i, w, a);
        if (s1) st2(s1, i, w, b);
    }
};
struct MaskOpen {       // E = X - prng(key, logical idx); limb: the stream is a product's A mask (limb form, cognn_gemm_mask)
    u64* E; const u64* X; u64 key; int64_t rows, cols; int transposed; int limb;
    __device__ u64 mask(u64 idx) const { const u64 w = cognn_prng(key, idx); return limb ? cognn_limb_value(w) : w; }
    __device__ void operator()(int64_t i, int w) const {
        u64 x[2], e[2];
        if (transposed == 3) {             // X stored [cols x rows], E written in logical [rows x cols] order (a small weight matrix)
            for (int j = 0; j < w; ++j) {
                const u64 idx = (u64)(i + j), m = idx / (u64)cols, k = idx % (u64)cols;
                e[j] = X[k * (u64)rows + m] - mask(idx);
            }
            st2(E, i, w, e);
            return;
        }
        ld2(X, i, w, x);
        for (int j = 0; j < 2; ++j) {
            u64 idx = (u64)(i + j);
            if (transposed == 1) {         // X stored [cols x rows]; logical element (m,k) = X[k][m]  (2: mask in storage order)
                u64 k = idx / (u64)rows, m = idx % (u64)rows;
                idx = m * (u64)cols + k;
            }
            e[j] = x[j] - mask(idx);
        }
        st2(E, i, w, e);
    }
};
// the public-opening close fed by a dealer's tensors instead of stream keys: t = r' + a_0 + a_1 (or r' for a reveal) and the
// party's own r'_p arrive from memory (cognn_trunc_close_pub_dealt_u64); DealPub is what that dealer computes
struct TruncClosePubDealt {
    u64* out; u64* E; const u64* c0; const u64* c1; const u64* t; const u64* rp; int p;
    __device__ void operator()(int64_t i, int w) const {
        u64 a[2], b[2], tt[2], r[2] = {0, 0}, y[2], e[2];
        ld2(c0, i, w, a); ld2(c1, i, w, b); ld2(t, i, w, tt);
        if (out) ld2(rp, i, w, r);
        for (int j = 0; j < 2; ++j) {
            const u64 hi = cognn_open_hi48(a[j], b[j]) - (COGNN_TRUNC_OFFSET >> COGNN_FX_BITS);
            e[j] = hi - tt[j];
            y[j] = (p == 0 ? hi : 0ull) - r[j];
        }
        if (out) st2(out, i, w, y);
        st2(E, i, w, e);
    }
};
struct DealPub {
    u64* t; u64* rp0; u64* rp1; cognn_opkeys k; u64 key_open0, key_open1; int reveal;
    __device__ void operator()(int64_t i, int w) const {
        u64 tt[2], r0[2], r1[2];
        for (int j = 0; j < 2; ++j) {
            const u64 idx = (u64)(i + j);
            const u64 rp = (cognn_prng(k.k[COGNN_SL_R], idx) & COGNN_TRUNC_MASK) >> COGNN_FX_BITS;
            r0[j] = trunc_rp(k, 0, idx); r1[j] = rp - r0[j];
            tt[j] = reveal ? rp : rp + cognn_prng(key_open0, idx) + cognn_prng(key_open1, idx);
        }
        st2(t, i, w, tt);
        if (rp0) st2(rp0, i, w, r0);
        if (rp1) st2(rp1, i, w, r1);
    }
};
struct AddSub {
    u64* out; const u64* a; const u64* b; int sub;
    __device__ void operator()(int64_t i, int w) const {
        u64 x[2], y[2], r[2];
        ld2(a, i, w, x); ld2(b, i, w, y);
        r[0] = sub ? x[0] - y[0] : x[0] + y[0];
        r[1] = sub ? x[1] - y[1] : x[1] + y[1];
        st2(out, i, w, r);
    }
};
constexpr int kFanMax = 16;
struct SumN {                                  // out = in[0] + ... + in[count-1]
    u64* out; const u64* in[kFanMax]; int count;
    __device__ void operator()(int64_t i, int w) const {
        u64 r[2] = {0, 0}, x[2];
        for (int c = 0; c < count; ++c) { ld2(in[c], i, w, x); r[0] += x[0]; r[1] += x[1]; }
        st2(out, i, w, r);
    }
};
struct FanOut {                                // out[0..count) = in
    u64* out[kFanMax]; const u64* in; int count;
    __device__ void operator()(int64_t i, int w) const {
        u64 x[2];
        ld2(in, i, w, x);
        for (int c = 0; c < count; ++c) st2(out[c], i, w, x);
    }
};
struct TruncOpen {
    u64* c; const u64* x; u64 mul; cognn_opkeys k; int p;
    __device__ void operator()(int64_t i, int w) const {
        u64 v[2], r[2];
        ld2(x, i, w, v);
        for (int j = 0; j < 2; ++j)
            r[j] = v[j] * mul + trunc_r(k, p, (u64)(i + j)) + (p == 0 ? COGNN_TRUNC_OFFSET : 0ull);
        st2(c, i, w, r);
    }
};
struct TruncOpenAdd {    // c = x + C_p + r_p (+offset): opening of a raw Beaver product
    u64* c; const u64* x; const u64* c1; u64 keyC0; cognn_opkeys k; int p;
    __device__ void operator()(int64_t i, int w) const {
        u64 v[2], a[2] = {0, 0}, r[2];
        ld2(x, i, w, v);
        if (p == 1) ld2(c1, i, w, a);
        for (int j = 0; j < 2; ++j) {
            const u64 cp = (p == 0) ? cognn_prng(keyC0, (u64)(i + j)) : a[j];
            r[j] = v[j] + cp + trunc_r(k, p, (u64)(i + j)) + (p == 0 ? COGNN_TRUNC_OFFSET : 0ull);
        }
        st2(c, i, w, r);
    }
};
struct TruncClose {
    // pub = 0: E (optional) = y_p - prng(key_open): this party's share of the next opening
    // pub = 1: both opened values are present on BOTH parties: E = y_0 + y_1 - a_0 - a_1, the next opening itself (public)
    // pub = 2: E = y_0 + y_1 (the result revealed to the caller)
    u64* out; const u64* c0; const u64* c1; cognn_opkeys k; int p; int mode; u64* E; u64 key_open; u64 key_open1; int pub;
    __device__ void operator()(int64_t i, int w) const {
        u64 y[2], yt[2] = {0, 0};
        if (p == 0 || pub) {
            u64 a[2], b[2];
            ld2(c0, i, w, a); ld2(c1, i, w, b);
            for (int j = 0; j < 2; ++j) {
                const u64 hi = cognn_open_hi48(a[j], b[j]) - (COGNN_TRUNC_OFFSET >> COGNN_FX_BITS);
                if (pub) yt[j] = hi - ((cognn_prng(k.k[COGNN_SL_R], (u64)(i + j)) & COGNN_TRUNC_MASK) >> COGNN_FX_BITS);
                y[j] = p == 0 ? hi - trunc_rp(k, 0, (u64)(i + j)) : 0ull - trunc_rp(k, 1, (u64)(i + j));
            }
        } else {
            for (int j = 0; j < 2; ++j) y[j] = 0ull - trunc_rp(k, 1, (u64)(i + j));
        }
        if (mode == 1) {
            u64 o[2];
            ld2(out, i, w, o);
            y[0] = o[0] - y[0]; y[1] = o[1] - y[1];
        }
        if (out) st2(out, i, w, y);
        if (E) {
            u64 e[2];
            for (int j = 0; j < 2; ++j)
                e[j] = pub == 0 ? y[j] - cognn_prng(key_open, (u64)(i + j))
                     : pub == 1 ? yt[j] - cognn_prng(key_open, (u64)(i + j)) - cognn_prng(key_open1, (u64)(i + j)) : yt[j];
            st2(E, i, w, e);
        }
    }
};
struct RowscaleOpenE {   // E = V - a_p
    u64* E; const u64* V; cognn_opkeys k; int p;
    __device__ void operator()(int64_t i, int w) const {
        u64 v[2], e[2];
        ld2(V, i, w, v);
        const u64 ka = k.k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1];
        e[0] = v[0] - cognn_prng(ka, (u64)i); e[1] = v[1] - cognn_prng(ka, (u64)i + 1);
        st2(E, i, w, e);
    }
};
struct RowscaleOpenG {   // G[r] = s_p[r] - b_p[r]
    u64* G; const u64* s; cognn_opkeys k; int p;
    __device__ void operator()(int64_t i, int w) const {
        u64 v[2], g[2];
        ld2(s, i, w, v);
        const u64 kb = k.k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1];
        g[0] = v[0] - cognn_prng(kb, (u64)i); g[1] = v[1] - cognn_prng(kb, (u64)i + 1);
        st2(G, i, w, g);
    }
};
struct RowscaleClose {   // c_out = beaver(E0+E1, (G0+G1)[row]) + trunc mask
    u64* c; const u64* E; const u64* E1; const u64* G; const u64* G1; cognn_opkeys k; cognn_opkeys tk; int p; uint32_t F;
    __device__ void operator()(int64_t i, int w) const {
        u64 e[2], r[2];
        ld2(E, i, w, e);
        if (E1) { u64 e1[2]; ld2(E1, i, w, e1); e[0] += e1[0]; e[1] += e1[1]; }
        if (w == 2 && (F & 1u) == 0) {                      // even width: both elements of the pair sit in one row
            const u64 row = (u64)((uint32_t)i / F);
            u64 g = G[row];
            if (G1) g += G1[row];
            const u64 b0 = cognn_prng(k.k[COGNN_SL_B0], row), b1 = p == 1 ? cognn_prng(k.k[COGNN_SL_B1], row) : 0ull;
            for (int j = 0; j < 2; ++j) {
                const u64 idx = (u64)(i + j);
                r[j] = beaver_mul_b(k, p, e[j], g, idx, b0, b1) + trunc_r(tk, p, idx) + (p == 0 ? COGNN_TRUNC_OFFSET : 0ull);
            }
        } else {
            for (int j = 0; j < w; ++j) {
                u64 idx = (u64)(i + j);
                u64 row = (u64)((uint32_t)idx / F);
                u64 g = G[row];
                if (G1) g += G1[row];
                u64 z = beaver_mul(k, p, e[j], g, idx, row);
                r[j] = z + trunc_r(tk, p, idx) + (p == 0 ? COGNN_TRUNC_OFFSET : 0ull);
            }
        }
        st2(c, i, w, r);
    }
};
struct ReluOpen {        // E = z - a_p ; G = t_p - b_p (G optional: the dealer can publish g = t - b offline, see ReluMul)
    u64* E; u64* G; const u64* z; cognn_opkeys k; int p;
    __device__ void operator()(int64_t i, int w) const {
        u64 v[2], e[2], g[2];
        ld2(z, i, w, v);
        for (int j = 0; j < 2; ++j) {
            u64 idx = (u64)(i + j);
            e[j] = v[j] - cognn_prng(k.k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], idx);
            if (G) {
                u64 t0 = cognn_prng(k.k[COGNN_SL_T0], idx);
                u64 tp = t0;
                if (p == 1) tp = ((cognn_prng(k.k[COGNN_SL_T], idx) & 0xFFFFFull) | COGNN_RELU_T_MIN) - t0;
                g[j] = tp - cognn_prng(k.k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], idx);
            }
        }
        st2(E, i, w, e);
        if (G) st2(G, i, w, g);
    }
};
struct ReluMul {
    // G == nullptr: g = t - (b0 + b1) is independent of the inputs (b masks t perfectly), so the dealer publishes it in the
    // offline phase and no G opening is exchanged online; here it is regenerated from the dealer streams like every mask
    u64* wout; const u64* E; const u64* E1; const u64* G; const u64* G1; cognn_opkeys k; int p;
    __device__ void operator()(int64_t i, int w) const {
        u64 e[2], g[2], r[2];
        ld2(E, i, w, e);
        if (E1) { u64 t[2]; ld2(E1, i, w, t); e[0] += t[0]; e[1] += t[1]; }
        if (G) {
            ld2(G, i, w, g);
            if (G1) { u64 t[2]; ld2(G1, i, w, t); g[0] += t[0]; g[1] += t[1]; }
        } else {
            for (int j = 0; j < 2; ++j) {
                u64 idx = (u64)(i + j);
                g[j] = ((cognn_prng(k.k[COGNN_SL_T], idx) & 0xFFFFFull) | COGNN_RELU_T_MIN) - cognn_prng(k.k[COGNN_SL_B0], idx) -
                       cognn_prng(k.k[COGNN_SL_B1], idx);
            }
        }
        for (int j = 0; j < 2; ++j) r[j] = beaver_mul(k, p, e[j], g[j], (u64)(i + j), (u64)(i + j));
        st2(wout, i, w, r);
    }
};
struct ReluClose {
    u64* h; uint8_t* mask; const u64* z; const u64* w0; const u64* w1; u64* E; u64 key_open;
    __device__ void operator()(int64_t i, int w) const {
        u64 a[2], b[2], v[2], r[2] = {0, 0};
        ld2(w0, i, w, a); ld2(w1, i, w, b); ld2(z, i, w, v);
        for (int j = 0; j < w; ++j) {
            bool pos = cognn_relu_positive(a[j], b[j]);
            r[j] = pos ? v[j] : 0ull;
            if (mask) mask[i + j] = pos ? 1 : 0;
        }
        st2(h, i, w, r);
        if (E) {
            u64 e[2] = {r[0] - cognn_gemm_mask(key_open, (u64)i), r[1] - cognn_gemm_mask(key_open, (u64)i + 1)};   // the next product's A mask
            st2(E, i, w, e);
        }
    }
};
struct MaskSelect {
    u64* out; const u64* in; const uint8_t* mask;
    __device__ void operator()(int64_t i, int w) const {
        u64 v[2], r[2];
        ld2(in, i, w, v);
        r[0] = mask[i] ? v[0] : 0ull;
        r[1] = (w == 2 && mask[i + 1]) ? v[1] : 0ull;
        st2(out, i, w, r);
    }
};
struct FxEncode {
    const double* in; const double* rowscale; u64* fx; uint32_t cols;
    __device__ void operator()(int64_t i, int w) const {
        u64 r[2] = {0, 0};
        for (int j = 0; j < w; ++j) {
            double v = in[i + j];
            if (rowscale) v *= rowscale[(uint32_t)(i + j) / cols];
            r[j] = (u64)(long long)llround(v * (double)COGNN_FX_ONE);
        }
        st2(fx, i, w, r);
    }
};

// G lanes per row (G = power of two >= L, <= 64): lane j owns column j, row reductions by shuffles, so
// every global access is coalesced.
template <int G>
__global__ __launch_bounds__(kThreads) void softmax_kernel(u64* p_out, u64* d_out, u64* pfx_out, const u64* z0, const u64* z1,
                                                            const int32_t* labels, cognn_opkeys k, int p, int64_t rows, int L,
                                                            int64_t train_rows) {
    const int64_t r = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G;
    const int j = threadIdx.x % G;
    const bool valid = r < rows && j < L;
    const bool keep = r < train_rows;
    const int64_t idx = r * L + j;
    const u64 rho = valid ? cognn_prng(k.k[COGNN_SL_RHO], (u64)idx) : 0ull;
    if (p == 1) {
        if (valid) {
            if (p_out) p_out[idx] = rho;
            d_out[idx] = keep ? rho : 0ull;
        }
        return;
    }
    const long long NEG = -(1ll << 62);
    long long z = valid ? (long long)(z0[idx] + z1[idx]) : NEG;
    long long m = z;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) { long long t = __shfl_xor(m, o, G); m = t > m ? t : m; }
    long long e = valid ? cognn_exp_neg_q30(m - z) : 0;
    long long S = e;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) S += __shfl_xor(S, o, G);
    if (!valid) return;
    const u64 pf = (u64)(((e << 16) + (S >> 1)) / S);
    const u64 p0 = pf - rho;
    if (pfx_out) pfx_out[idx] = pf;
    if (p_out) p_out[idx] = p0;
    d_out[idx] = keep ? (p0 - (j == labels[r] ? COGNN_FX_ONE : 0ull)) : 0ull;
}

template <int G>
__global__ __launch_bounds__(kThreads) void metrics_kernel(const u64* pfx, const int32_t* labels, const uint8_t* border,
                                                            int64_t rows, int L, int64_t train_rows, int64_t val_rows,
                                                            unsigned long long* counts, double* loss) {
    // grid-stride over row groups; per-thread partial counts -> LDS -> one set of global atomics per workgroup
    __shared__ unsigned long long s_cnt[5];
    __shared__ double s_loss;
    if (threadIdx.x < 5) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_loss = 0.0;
    __syncthreads();
    const int j = threadIdx.x % G;
    unsigned int c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    double ls = 0.0;
    // A group of G lanes owns G consecutive rows per pass: the argmax of each row is a G-lane shuffle reduction (lane 0 ends up
    // with the row's result), then lane i takes over row i's log() so that the double-precision log runs on all lanes
    const int64_t groups = (int64_t)gridDim.x * (kThreads / G);
    const int64_t gid = (int64_t)blockIdx.x * (kThreads / G) + threadIdx.x / G;
    for (int64_t rb = gid * G; rb < rows; rb += groups * G) {          // uniform inside every shuffle group
        double my_pl = 1.0;                                            // log(1) = 0 for rows beyond the end
        for (int i = 0; i < G; ++i) {
            const int64_t r = rb + i;
            const bool valid = r < rows && j < L;
            u64 bv = valid ? pfx[r * L + j] : 0ull;
            int bj = valid ? j : (1 << 20);
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {          // argmax, first maximum wins ties
                u64 ov = __shfl_xor(bv, o, G);
                int oj = __shfl_xor(bj, o, G);
                if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
            }
            // after the butterfly every lane holds the row's argmax: lane i keeps the bookkeeping of row i
            if (j == i && r < rows) {
                const int lab = labels[r];
                const bool ok = bj == lab;
                const bool b = border ? border[r] != 0 : false;
                const bool tr = r < train_rows, te = r >= train_rows + val_rows;
                c0 += ok; c1 += ok && tr; c2 += ok && tr && b; c3 += ok && te; c4 += ok && te && b;
                my_pl = (double)pfx[r * L + lab] / (double)COGNN_FX_ONE;
                if (my_pl == 0.0) my_pl = 0.001;           /* gcn.h:613-615 */
            }
        }
        ls += -log(my_pl);
    }
    // per-thread partials -> LDS
    if (c0) atomicAdd(&s_cnt[0], (unsigned long long)c0);
    if (c1) atomicAdd(&s_cnt[1], (unsigned long long)c1);
    if (c2) atomicAdd(&s_cnt[2], (unsigned long long)c2);
    if (c3) atomicAdd(&s_cnt[3], (unsigned long long)c3);
    if (c4) atomicAdd(&s_cnt[4], (unsigned long long)c4);
    if (ls != 0.0) atomicAdd(&s_loss, ls);
    __syncthreads();
    if (threadIdx.x < 5 && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    if (threadIdx.x == 0) atomicAdd(loss, s_loss);
}

// ---- prediction layer of all hosted sides in one launch; owner jobs fuse the metrics (gcn.h:578-632) --------------------------
struct SoftmaxJobDev {
    u64* d_out; const u64* z0; const u64* z1; const int32_t* labels; const uint8_t* border;
    u64 keyRho; int p; int64_t rows, train_rows, val_rows;
    unsigned long long* counts; double* loss;
};
constexpr int kSoftmaxJobsMax = 16;
struct SoftmaxBatch {
    SoftmaxJobDev j[kSoftmaxJobsMax];
    unsigned blk_end[kSoftmaxJobsMax];
    int count;
};
__global__ void softmax_jobs_zero_kernel(SoftmaxBatch b) {
    const int t = threadIdx.x;
    if (t < b.count * 8 && b.j[t >> 3].p == 0) {
        if ((t & 7) < 6) b.j[t >> 3].counts[t & 7] = 0ull;
        else if ((t & 7) == 6) *b.j[t >> 3].loss = 0.0;
    }
}
// A group of G lanes (G = power of two >= L) owns G consecutive rows: row i of the group is evaluated across the G lanes
// (lane j = column j, reductions by shuffles, coalesced 8*L-byte row accesses), and lane i keeps row i's bookkeeping, so that
// the double-precision log of the loss runs once per lane instead of once per row on 1 lane in G.
// rpg = rows per lane group: G for large inputs (every lane ends up with one row's bookkeeping), 1 for small ones (a group walks one row
// instead of G in sequence: G times the parallelism when the whole input is a few workgroups)
template <int G>
__global__ __launch_bounds__(kThreads) void softmax_jobs_kernel(SoftmaxBatch b, int L, int rpg) {
    __shared__ unsigned long long s_cnt[5];
    __shared__ double s_loss;
    const unsigned blk = blockIdx.x;
    int seg = 0;
    while (seg < b.count - 1 && blk >= b.blk_end[seg]) ++seg;
    const SoftmaxJobDev& d = b.j[seg];
    const unsigned blk0 = seg ? b.blk_end[seg - 1] : 0u;
    const int j = threadIdx.x % G;
    const int64_t rb = (((int64_t)(blk - blk0) * kThreads + threadIdx.x) / G) * rpg;   // first row of this lane group
    if (d.p == 1) {                                          // (uniform per workgroup) the co-party's share is its mask
        for (int i = 0; i < rpg; ++i) {
            const int64_t r = rb + i, idx = r * L + j;
            if (r < d.rows && j < L) d.d_out[idx] = r < d.train_rows ? cognn_prng(d.keyRho, (u64)idx) : 0ull;
        }
        return;
    }
    if (threadIdx.x < 5) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_loss = 0.0;
    __syncthreads();
    const long long NEG = -(1ll << 62);
    double my_pl = 1.0;                                      // log(1) = 0 for lanes without a row
    bool ok = false, tr = false, te = false, bd = false;
    for (int i = 0; i < rpg; ++i) {                          // (uniform inside every shuffle group)
        const int64_t r = rb + i;
        const bool valid = r < d.rows && j < L;
        const int64_t idx = r * L + j;
        long long z = valid ? (long long)(d.z0[idx] + (d.z1 ? d.z1[idx] : 0ull)) : NEG;
        long long m = z;
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) { long long t = __shfl_xor(m, o, G); m = t > m ? t : m; }
        long long e = valid ? cognn_exp_neg_q30(m - z) : 0;
        long long S = e;
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) S += __shfl_xor(S, o, G);
        // the revealed Q16 probability (softmax_kernel); e <= 2^30, S <= 64 * 2^30: the dividend stays below 2^47
        const u64 pf = valid ? div_floor_small((u64)((e << 16) + (S >> 1)), (u64)S) : 0ull;
        const int lab = r < d.rows ? d.labels[r] : 0;
        if (valid) {
            const u64 rho = cognn_prng(d.keyRho, (u64)idx);
            d.d_out[idx] = r < d.train_rows ? ((pf - rho) - (j == lab ? COGNN_FX_ONE : 0ull)) : 0ull;
        }
        // metrics on the revealed probabilities (metrics_kernel): argmax with the first maximum winning ties, p[label]
        u64 bv = valid ? pf : 0ull;
        int bj = valid ? j : (1 << 20);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            u64 ov = __shfl_xor(bv, o, G);
            int oj = __shfl_xor(bj, o, G);
            if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
        }
        const u64 p_lab = __shfl(pf, lab, G);
        if (j == i && r < d.rows) {                          // lane i takes over row i
            ok = bj == lab;
            bd = d.border ? d.border[r] != 0 : false;
            tr = r < d.train_rows; te = r >= d.train_rows + d.val_rows;
            my_pl = (double)p_lab / (double)COGNN_FX_ONE;
            if (my_pl == 0.0) my_pl = 0.001;                 /* gcn.h:613-615 */
        }
    }
    const double ls = -log(my_pl);
    if (ok) {
        atomicAdd(&s_cnt[0], 1ull);
        if (tr) atomicAdd(&s_cnt[1], 1ull);
        if (tr && bd) atomicAdd(&s_cnt[2], 1ull);
        if (te) atomicAdd(&s_cnt[3], 1ull);
        if (te && bd) atomicAdd(&s_cnt[4], 1ull);
    }
    if (ls != 0.0) atomicAdd(&s_loss, ls);
    __syncthreads();
    if (threadIdx.x < 5 && s_cnt[threadIdx.x]) atomicAdd(&d.counts[threadIdx.x], s_cnt[threadIdx.x]);
    if (threadIdx.x == 0 && s_loss != 0.0) atomicAdd(d.loss, s_loss);
}

__global__ __launch_bounds__(kThreads) void transpose_kernel(u64* out, const u64* in, int64_t rows, int64_t cols) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i % cols;
    out[c * rows + r] = in[i];
}

// ---- pair chain launcher (device code: pair_chain.h) ---------------------------------------------------------------------
constexpr int kPairBatchMax = 8;
struct PairBatch {
    PairChainDev d[kPairBatchMax];
    unsigned blk_end[kPairBatchMax];
    int count;
};
template <int STREAM>   // STREAM 1 / 2: every chain of the batch brings its dealt slab and reads all of it / the dealer's corrections only (pair_chain.h)
__global__ __launch_bounds__(kThreads) void pair_chain_kernel(PairBatch b) {
    const unsigned blk = blockIdx.x;
    int seg = 0;
    while (seg < b.count - 1 && blk >= b.blk_end[seg]) ++seg;
    const PairChainDev& d = b.d[seg];
    const unsigned blk0 = seg ? b.blk_end[seg - 1] : 0u;
    const int64_t i = 2 * ((int64_t)(blk - blk0) * kThreads + threadIdx.x);
    if (i >= d.n) return;
    const int w = (i + 1 < d.n) ? 2 : 1;
    const uint32_t flags = d.flags;
    const PcSlotBase SB = pc_slot_base(flags, d.open0 != nullptr || d.open1 != nullptr);
    u64 v0[2], v1[2];
    ld2(d.x0, i, w, v0); ld2(d.x1, i, w, v1);
    if (flags & COGNN_PC_CLEAR_INPUT) {                      // the product buffers go back clean (cognn_gemm_job::Z_zeroed)
        const u64 zero[2] = {0, 0};
        st2(const_cast<u64*>(d.x0), i, w, zero); st2(const_cast<u64*>(d.x1), i, w, zero);
    }
    const bool mask_mid = (flags & COGNN_PC_MASK_AFTER_TRUNC) != 0;
    if (d.mask_in && !mask_mid) {                            // MaskSelect on both sides' inputs
        if (!d.mask_in[i]) { v0[0] = 0; v1[0] = 0; }
        if (w == 2 && !d.mask_in[i + 1]) { v0[1] = 0; v1[1] = 0; }
    }
    if (flags & COGNN_PC_TRUNC_IN) {
        u64 cc[2] = {0, 0};
        const bool addc = !(flags & COGNN_PC_NO_C);
        if (addc) ld2(d.c1, i, w, cc);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u64 idx = (u64)(i + j);
            if (addc) { v0[j] += STREAM == 1 ? d.slab[(u64)(SB.ti + PCS_TI_C0) * (u64)d.n + idx] : cognn_prng(d.keyC0, idx); v1[j] += cc[j]; }     // TruncOpenAdd: x + C_p
            if (j < w) pair_trunc<STREAM>(d, SB.ti + PCS_TI_R0, d.tiR, d.tiR0, d.tiRP0, idx, v0[j], v1[j]);
        }
    }
    if (d.mask_in && mask_mid) {                             // ... or on the truncated product (COGNN_PC_MASK_AFTER_TRUNC)
        if (!d.mask_in[i]) { v0[0] = 0; v1[0] = 0; }
        if (w == 2 && !d.mask_in[i + 1]) { v0[1] = 0; v1[1] = 0; }
    }
    if (flags & COGNN_PC_SCALE) {
        // i is even: with an even row width both elements lie in one row, whose values are formed once
        const uint32_t row0 = (uint32_t)i / d.F, row1 = (d.F & 1u) ? (uint32_t)(i + 1) / d.F : row0;
        const PairRow rw0 = pair_row(d, (u64)row0);
        PairRow rw1 = rw0;
        if (row1 != row0 && w == 2) rw1 = pair_row(d, (u64)row1);
        pair_scale<STREAM>(d, SB.sc, (u64)i, rw0, (flags & COGNN_PC_INPUT_OPENED) != 0, v0[0], v1[0]);
        if (w == 2) pair_scale<STREAM>(d, SB.sc, (u64)i + 1, rw1, (flags & COGNN_PC_INPUT_OPENED) != 0, v0[1], v1[1]);
    }
    if (flags & COGNN_PC_RELU) {
        bool pos[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) pos[j] = (j < w) ? pair_relu<STREAM>(d, SB.re, (u64)(i + j), v0[j], v1[j]) : false;
        if (d.mask) { d.mask[i] = pos[0] ? 1 : 0; if (w == 2) d.mask[i + 1] = pos[1] ? 1 : 0; }
    }
    if (d.out0) st2(d.out0, i, w, v0);
    if (d.out1) st2(d.out1, i, w, v1);
    if (d.open0 || d.open1) {
        u64 a0[2] = {0, 0}, a1[2] = {0, 0};
        for (int j = 0; j < w; ++j) pair_open_masks<STREAM>(d, SB.op, (u64)(i + j), a0[j], a1[j]);
        if (d.flags & COGNN_PC_OPEN_SUM) {
            if (d.open0) { u64 e[2] = {(v0[0] - a0[0]) + (v1[0] - a1[0]), (v0[1] - a0[1]) + (v1[1] - a1[1])}; st2(d.open0, i, w, e); }
            return;
        }
        if (d.open0) { u64 e[2] = {v0[0] - a0[0], v0[1] - a0[1]}; st2(d.open0, i, w, e); }
        if (d.open1) { u64 e[2] = {v1[0] - a1[0], v1[1] - a1[1]}; st2(d.open1, i, w, e); }
    }
}
// the offline phase of one chain: its dealt slab (cognn_pair_chain_deal_u64)
__global__ __launch_bounds__(kThreads) void pair_chain_deal_kernel(PairChainDev d, u64* slab, int has_open) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= d.n) return;
    pair_deal_element(d, slab, (u64)i, (u64)((uint32_t)i / d.F), has_open != 0);
}

// ---- weight update of co-located pairs (cognn_pair_weight_update_u64) -----------------------------------------------------
struct PairWUpdateDev {
    const u64* z0; const u64* z1; const u64* c1; u64* W0; u64* W1;
    u64 keyC0;
    u64 kR[4], kR0[4], kRP0[4];      // the three truncation streams of: product, gradient scale, learning rate, post scale
    u64 mul[3];
    int64_t n; int addc; int swap; int clear;
};
constexpr int kWUpdateMax = 16;
struct PairWUpdateBatch {
    PairWUpdateDev d[kWUpdateMax];
    unsigned blk_end[kWUpdateMax];
    int count;
    u64 aR, aR0, aRP0, amul;         // averaging form: the truncation streams and multiplier of the 1/k scale (amul = 0: none)
};
__device__ __forceinline__ void trunc2(u64 kR, u64 kR0, u64 kRP0, u64 idx, u64& v0, u64& v1) {
    PairChainDev none;                                       // (the PRNG form of pair_trunc reads nothing from the chain)
    none.slab = nullptr; none.n = 0;
    pair_trunc<false>(none, 0, kR, kR0, kRP0, idx, v0, v1);
}
// new weight shares of one pair for elements i, i + 1
__device__ __forceinline__ void wupdate_pair(const PairWUpdateDev& d, int64_t i, int w, u64* w0, u64* w1) {
    u64 v0[2], v1[2], cc[2] = {0, 0};
    ld2(d.z0, i, w, v0); ld2(d.z1, i, w, v1);
    if (d.clear) {
        const u64 zero[2] = {0, 0};
        st2(const_cast<u64*>(d.z0), i, w, zero); st2(const_cast<u64*>(d.z1), i, w, zero);
    }
    ld2(d.W0, i, w, w0); ld2(d.W1, i, w, w1);
    if (d.addc) ld2(d.c1, i, w, cc);
    for (int j = 0; j < w; ++j) {
        const u64 idx = (u64)(i + j);
        if (d.addc) { v0[j] += cognn_prng(d.keyC0, idx); v1[j] += cc[j]; }                      // TruncOpenAdd: x + C_p
        trunc2(d.kR[0], d.kR0[0], d.kRP0[0], idx, v0[j], v1[j]);                                // d = h_t^T . in
        v0[j] *= d.mul[0]; v1[j] *= d.mul[0];
        trunc2(d.kR[1], d.kR0[1], d.kRP0[1], idx, v0[j], v1[j]);                                // d / |train|
        v0[j] *= d.mul[1]; v1[j] *= d.mul[1];
        trunc2(d.kR[2], d.kR0[2], d.kRP0[2], idx, v0[j], v1[j]);                                // lr . d
        w0[j] -= v0[j]; w1[j] -= v1[j];                                                         // TruncClose, mode 1
        if (d.mul[2]) {
            w0[j] *= d.mul[2]; w1[j] *= d.mul[2];
            trunc2(d.kR[3], d.kR0[3], d.kRP0[3], idx, w0[j], w1[j]);                            // W / k (inference variant)
        }
    }
}
// independent pairs: one segment of the grid each
__global__ __launch_bounds__(kThreads) void pair_wupdate_kernel(PairWUpdateBatch b) {
    const unsigned blk = blockIdx.x;
    int seg = 0;
    while (seg < b.count - 1 && blk >= b.blk_end[seg]) ++seg;
    const PairWUpdateDev& d = b.d[seg];
    const unsigned blk0 = seg ? b.blk_end[seg - 1] : 0u;
    const int64_t i = 2 * ((int64_t)(blk - blk0) * kThreads + threadIdx.x);
    if (i >= d.n) return;
    const int w = (i + 1 < d.n) ? 2 : 1;
    u64 w0[2], w1[2];
    wupdate_pair(d, i, w, w0, w1);
    st2(d.W0, i, w, w0); st2(d.W1, i, w, w1);
}
// every party's pair is here: update + the weight average of gcn.h:753-778 (sum of one share of every party's weights on party 0,
// of the other on party 1, the 1/k scale between those two, redistribution) element by element
__global__ __launch_bounds__(kThreads) void pair_wupdate_avg_kernel(PairWUpdateBatch b) {
    const int64_t n = b.d[0].n;
    const int64_t i = 2 * ((int64_t)blockIdx.x * kThreads + threadIdx.x);
    if (i >= n) return;
    const int w = (i + 1 < n) ? 2 : 1;
    u64 s0[2] = {0, 0}, s1[2] = {0, 0};
    for (int c = 0; c < b.count; ++c) {
        u64 w0[2] = {0, 0}, w1[2] = {0, 0};
        wupdate_pair(b.d[c], i, w, w0, w1);
        const bool sw = b.d[c].swap != 0;
        for (int j = 0; j < 2; ++j) { s0[j] += sw ? w1[j] : w0[j]; s1[j] += sw ? w0[j] : w1[j]; }
    }
    if (b.amul)
        for (int j = 0; j < w; ++j) {
            s0[j] *= b.amul; s1[j] *= b.amul;
            trunc2(b.aR, b.aR0, b.aRP0, (u64)(i + j), s0[j], s1[j]);
        }
    for (int c = 0; c < b.count; ++c) {
        const bool sw = b.d[c].swap != 0;
        st2(sw ? b.d[c].W1 : b.d[c].W0, i, w, s0);
        st2(sw ? b.d[c].W0 : b.d[c].W1, i, w, s1);
    }
}

inline cognn_opkeys K(const cognn_keys* k) {
    cognn_opkeys r;
    for (int i = 0; i < COGNN_SL_COUNT; ++i) r.k[i] = k->k[i];
    return r;
}
inline bool al(const void* p) { return p == nullptr || cg_aligned16(p); }

}  // namespace

static_assert(COGNN_NUM_SLOTS == COGNN_SL_COUNT, "slot count mismatch between ABI header and spec");

extern "C" {

int cognn_prng_fill_u64(cognn_ctx* ctx, uint64_t* out, uint64_t key, int64_t n) {
    CG_REQUIRE(ctx && out && al(out), "cognn_prng_fill_u64: bad arguments");
    return launch_ew(ctx, n, PrngFill{(u64*)out, key, 0});
}
// Packed openings (COGNN_OPT_PACKED_OPENINGS): of an opened truncation share only the top 48 bits matter (cognn_open_hi48), so between
// ranks 6 bytes per element travel: a plane of n 32-bit words (bits 16..47) followed by a plane of n 16-bit words (bits 48..63).
namespace {
__global__ __launch_bounds__(256) void pack48_kernel(uint32_t* mid, uint16_t* top, const u64* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const u64 v = src[i];
        mid[i] = (uint32_t)(v >> 16); top[i] = (uint16_t)(v >> 48);
    }
}
__global__ __launch_bounds__(256) void unpack48_kernel(u64* dst, const uint32_t* __restrict__ mid, const uint16_t* __restrict__ top, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = ((u64)mid[i] << 16) | ((u64)top[i] << 48);
}
}  // namespace
int cognn_pack48_u64(cognn_ctx* ctx, void* packed, const uint64_t* src, int64_t n) {
    { const int rc_flush_ = cg_flush_pending(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && n >= 0 && (n == 0 || (packed && src && (((uintptr_t)packed) & 3u) == 0)), "cognn_pack48_u64: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack48_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 16384)), dim3(256), 0, ctx->stream, (uint32_t*)packed,
                       (uint16_t*)((unsigned char*)packed + 4 * n), (const u64*)src, n);
    CG_LAUNCH_CHECK();
    return 0;
}
int cognn_unpack48_u64(cognn_ctx* ctx, uint64_t* dst, const void* packed, int64_t n) {
    { const int rc_flush_ = cg_flush_pending(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && n >= 0 && (n == 0 || (packed && dst && (((uintptr_t)packed) & 3u) == 0)), "cognn_unpack48_u64: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(unpack48_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 16384)), dim3(256), 0, ctx->stream, (u64*)dst, (const uint32_t*)packed,
                       (const uint16_t*)((const unsigned char*)packed + 4 * n), n);
    CG_LAUNCH_CHECK();
    return 0;
}
int cognn_gemm_mask_fill_u64(cognn_ctx* ctx, uint64_t* out, uint64_t key, int64_t n) {
    CG_REQUIRE(ctx && out && al(out), "cognn_gemm_mask_fill_u64: bad arguments");
    return launch_ew(ctx, n, PrngFill{(u64*)out, key, 1});
}
int cognn_share_split_u64(cognn_ctx* ctx, const uint64_t* fx, uint64_t key, uint64_t* s0, uint64_t* s1, int64_t n) {
    CG_REQUIRE(ctx && fx && al(fx) && al(s0) && al(s1), "cognn_share_split_u64: bad arguments");
    return launch_ew(ctx, n, ShareSplit{(const u64*)fx, key, (u64*)s0, (u64*)s1});
}
int cognn_fx_encode_f64(cognn_ctx* ctx, const double* in, const double* rowscale, uint64_t* fx, int64_t rows, int64_t cols) {
    CG_REQUIRE(ctx && in && fx && al(fx), "cognn_fx_encode_f64: bad arguments");
    CG_REQUIRE(rows * cols < (1ll << 32), "cognn_fx_encode_f64: tensor too large");
    return launch_ew(ctx, rows * cols, FxEncode{in, rowscale, (u64*)fx, (uint32_t)cols});
}
int cognn_mask_open_u64(cognn_ctx* ctx, uint64_t* E, const uint64_t* X, uint64_t key, int64_t rows, int64_t cols, int transposed) {
    CG_REQUIRE(ctx && E && X && al(E) && al(X), "cognn_mask_open_u64: bad arguments");
    const int limb = (transposed & COGNN_MASK_OPEN_LIMB) ? 1 : 0;
    transposed &= ~COGNN_MASK_OPEN_LIMB;
    CG_REQUIRE(transposed >= 0 && transposed <= 3, "cognn_mask_open_u64: bad `transposed`");
    return launch_ew(ctx, rows * cols, MaskOpen{(u64*)E, (const u64*)X, key, rows, cols, transposed, limb});
}
int cognn_add_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* b, int64_t n) {
    CG_REQUIRE(ctx && out && a && b && al(out) && al(a) && al(b), "cognn_add_u64: bad arguments");
    return launch_ew(ctx, n, AddSub{(u64*)out, (const u64*)a, (const u64*)b, 0});
}
int cognn_sum_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* const* in, int32_t count, int64_t n) {
    CG_REQUIRE(ctx && out && in && count >= 1 && count <= kFanMax && al(out), "cognn_sum_u64: bad arguments (1..%d inputs)", kFanMax);
    SumN f;
    f.out = (u64*)out; f.count = count;
    for (int c = 0; c < count; ++c) { CG_REQUIRE(in[c] && al(in[c]), "cognn_sum_u64: input %d is null or misaligned", c); f.in[c] = (const u64*)in[c]; }
    return launch_ew(ctx, n, f);
}
int cognn_fanout_u64(cognn_ctx* ctx, uint64_t* const* out, int32_t count, const uint64_t* in, int64_t n) {
    CG_REQUIRE(ctx && out && in && count >= 1 && count <= kFanMax && al(in), "cognn_fanout_u64: bad arguments (1..%d outputs)", kFanMax);
    FanOut f;
    f.in = (const u64*)in; f.count = count;
    for (int c = 0; c < count; ++c) { CG_REQUIRE(out[c] && al(out[c]), "cognn_fanout_u64: output %d is null or misaligned", c); f.out[c] = (u64*)out[c]; }
    return launch_ew(ctx, n, f);
}
int cognn_sub_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* a, const uint64_t* b, int64_t n) {
    CG_REQUIRE(ctx && out && a && b && al(out) && al(a) && al(b), "cognn_sub_u64: bad arguments");
    return launch_ew(ctx, n, AddSub{(u64*)out, (const u64*)a, (const u64*)b, 1});
}
int cognn_trunc_open_u64(cognn_ctx* ctx, uint64_t* c, const uint64_t* x, uint64_t mul, const cognn_keys* keys, int p, int64_t n) {
    CG_REQUIRE(ctx && c && x && keys && (p == 0 || p == 1) && al(c) && al(x), "cognn_trunc_open_u64: bad arguments");
    return launch_ew(ctx, n, TruncOpen{(u64*)c, (const u64*)x, mul, K(keys), p});
}
int cognn_trunc_open_add_u64(cognn_ctx* ctx, uint64_t* c, const uint64_t* x, const uint64_t* c1, const cognn_keys* gkeys,
                             const cognn_keys* tkeys, int p, int64_t n) {
    CG_REQUIRE(ctx && c && x && gkeys && tkeys && (p == 0 || (p == 1 && c1)) && al(c) && al(x) && al(c1), "cognn_trunc_open_add_u64: bad arguments");
    return launch_ew(ctx, n, TruncOpenAdd{(u64*)c, (const u64*)x, (const u64*)c1, gkeys->k[COGNN_SL_C0], K(tkeys), p});
}
int cognn_trunc_close_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                          int p, int mode, int64_t n) {
    CG_REQUIRE(ctx && out && keys && (p == 0 || p == 1) && al(out) && al(c0) && al(c1), "cognn_trunc_close_u64: bad arguments");
    CG_REQUIRE(p == 1 || (c0 && c1), "cognn_trunc_close_u64: p=0 needs both opened values");
    return launch_ew(ctx, n, TruncClose{(u64*)out, (const u64*)c0, (const u64*)c1, K(keys), p, mode, nullptr, 0, 0, 0});
}
int cognn_trunc_close_open_u64(cognn_ctx* ctx, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                               int p, uint64_t key_open, int64_t n) {
    CG_REQUIRE(ctx && out && E && keys && (p == 0 || p == 1) && al(out) && al(E) && al(c0) && al(c1), "cognn_trunc_close_open_u64: bad arguments");
    CG_REQUIRE(p == 1 || (c0 && c1), "cognn_trunc_close_open_u64: p=0 needs both opened values");
    return launch_ew(ctx, n, TruncClose{(u64*)out, (const u64*)c0, (const u64*)c1, K(keys), p, 0, (u64*)E, key_open, 0, 0});
}
int cognn_trunc_close_pub_u64(cognn_ctx* ctx, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                              int p, uint64_t key_open0, uint64_t key_open1, int reveal, int64_t n) {
    CG_REQUIRE(ctx && E && c0 && c1 && keys && (p == 0 || p == 1) && al(out) && al(E) && al(c0) && al(c1), "cognn_trunc_close_pub_u64: bad arguments (both opened values are needed)");
    return launch_ew(ctx, n, TruncClose{(u64*)out, (const u64*)c0, (const u64*)c1, K(keys), p, 0, (u64*)E, key_open0, key_open1, reveal ? 2 : 1});
}
int cognn_trunc_close_pub_dealt_u64(cognn_ctx* ctx, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const uint64_t* t,
                                    const uint64_t* rp_own, int p, int64_t n) {
    CG_REQUIRE(ctx && E && c0 && c1 && t && (p == 0 || p == 1) && (!out || rp_own) && al(out) && al(E) && al(c0) && al(c1) && al(t) && al(rp_own),
               "cognn_trunc_close_pub_dealt_u64: bad arguments (both opened values, the published t and - with out - the party's r' share are needed)");
    return launch_ew(ctx, n, TruncClosePubDealt{(u64*)out, (u64*)E, (const u64*)c0, (const u64*)c1, (const u64*)t, (const u64*)rp_own, p});
}
int cognn_dealer_trunc_pub_u64(cognn_ctx* ctx, uint64_t* t, uint64_t* rp0, uint64_t* rp1, const cognn_keys* keys, uint64_t key_open0,
                               uint64_t key_open1, int reveal, int64_t n) {
    CG_REQUIRE(ctx && t && keys && al(t) && al(rp0) && al(rp1), "cognn_dealer_trunc_pub_u64: bad arguments");
    return launch_ew(ctx, n, DealPub{(u64*)t, (u64*)rp0, (u64*)rp1, K(keys), key_open0, key_open1, reveal});
}
int cognn_rowscale_open_u64(cognn_ctx* ctx, uint64_t* E, uint64_t* G, const uint64_t* V, const uint64_t* s,
                            const cognn_keys* keys, int p, int64_t rows, int64_t F) {
    CG_REQUIRE(ctx && G && s && keys && (E == nullptr || V) && al(E) && al(G) && al(V) && al(s), "cognn_rowscale_open_u64: bad arguments");
    int rc = E ? launch_ew(ctx, rows * F, RowscaleOpenE{(u64*)E, (const u64*)V, K(keys), p}) : 0;   // E == NULL: opened elsewhere
    if (rc) return rc;
    return launch_ew(ctx, rows, RowscaleOpenG{(u64*)G, (const u64*)s, K(keys), p});
}
int cognn_rowscale_close_u64(cognn_ctx* ctx, uint64_t* c_out, const uint64_t* E, const uint64_t* E1, const uint64_t* G, const uint64_t* G1,
                             const cognn_keys* keys, const cognn_keys* tkeys, int p, int64_t rows, int64_t F) {
    CG_REQUIRE(ctx && c_out && E && G && keys && tkeys && al(c_out) && al(E) && al(E1), "cognn_rowscale_close_u64: bad arguments");
    CG_REQUIRE(rows * F < (1ll << 32), "cognn_rowscale_close_u64: tensor too large");
    return launch_ew(ctx, rows * F, RowscaleClose{(u64*)c_out, (const u64*)E, (const u64*)E1, (const u64*)G, (const u64*)G1, K(keys), K(tkeys), p, (uint32_t)F});
}
int cognn_relu_open_u64(cognn_ctx* ctx, uint64_t* E, uint64_t* G, const uint64_t* z, const cognn_keys* keys, int p, int64_t n) {
    CG_REQUIRE(ctx && E && z && keys && al(E) && al(G) && al(z), "cognn_relu_open_u64: bad arguments");
    return launch_ew(ctx, n, ReluOpen{(u64*)E, (u64*)G, (const u64*)z, K(keys), p});
}
int cognn_relu_mul_u64(cognn_ctx* ctx, uint64_t* w, const uint64_t* E, const uint64_t* E1, const uint64_t* G, const uint64_t* G1,
                       const cognn_keys* keys, int p, int64_t n) {
    CG_REQUIRE(ctx && w && E && keys && (G || !G1) && al(w) && al(E) && al(G) && al(E1) && al(G1), "cognn_relu_mul_u64: bad arguments");
    return launch_ew(ctx, n, ReluMul{(u64*)w, (const u64*)E, (const u64*)E1, (const u64*)G, (const u64*)G1, K(keys), p});
}
int cognn_relu_close_u64(cognn_ctx* ctx, uint64_t* h, uint8_t* mask, const uint64_t* z, const uint64_t* w0, const uint64_t* w1, int64_t n) {
    CG_REQUIRE(ctx && h && z && w0 && w1 && al(h) && al(z) && al(w0) && al(w1), "cognn_relu_close_u64: bad arguments");
    return launch_ew(ctx, n, ReluClose{(u64*)h, mask, (const u64*)z, (const u64*)w0, (const u64*)w1, nullptr, 0});
}
int cognn_relu_close_open_u64(cognn_ctx* ctx, uint64_t* h, uint64_t* E, uint8_t* mask, const uint64_t* z, const uint64_t* w0, const uint64_t* w1,
                              uint64_t key_open, int64_t n) {
    CG_REQUIRE(ctx && h && E && z && w0 && w1 && al(h) && al(E) && al(z) && al(w0) && al(w1), "cognn_relu_close_open_u64: bad arguments");
    return launch_ew(ctx, n, ReluClose{(u64*)h, mask, (const u64*)z, (const u64*)w0, (const u64*)w1, (u64*)E, key_open});
}
int cognn_mask_select_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* in, const uint8_t* mask, int64_t n) {
    CG_REQUIRE(ctx && out && in && mask && al(out) && al(in), "cognn_mask_select_u64: bad arguments");
    return launch_ew(ctx, n, MaskSelect{(u64*)out, (const u64*)in, mask});
}
int cognn_softmax_u64(cognn_ctx* ctx, uint64_t* p_out, uint64_t* d_out, uint64_t* pfx_out, const uint64_t* z0, const uint64_t* z1,
                      const int32_t* labels, const cognn_keys* keys, int p, int64_t rows, int64_t L, int64_t train_rows) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && d_out && keys && (p == 0 || p == 1), "cognn_softmax_u64: bad arguments");
    CG_REQUIRE(p == 1 || (z0 && z1 && labels), "cognn_softmax_u64: owner side needs z0, z1, labels");
    CG_REQUIRE(L > 0 && L <= 64, "cognn_softmax_u64: unsupported label count %lld (max 64)", (long long)L);
    if (rows <= 0) return 0;
    int G = 1;
    while (G < L) G <<= 1;
#define CG_SM_CASE(g)                                                                                                       \
    case g:                                                                                                                  \
        hipLaunchKernelGGL(softmax_kernel<g>, dim3(cg_div_up(rows * g, kThreads)), dim3(kThreads), 0, ctx->stream, (u64*)p_out, \
                           (u64*)d_out, (u64*)pfx_out, (const u64*)z0, (const u64*)z1, labels, K(keys), p, rows, (int)L, train_rows); \
        break;
    switch (G) { CG_SM_CASE(1) CG_SM_CASE(2) CG_SM_CASE(4) CG_SM_CASE(8) CG_SM_CASE(16) CG_SM_CASE(32) CG_SM_CASE(64) }
#undef CG_SM_CASE
    CG_LAUNCH_CHECK();
    return 0;
}
int cognn_metrics_q16(cognn_ctx* ctx, const uint64_t* pfx, const int32_t* labels, const uint8_t* border,
                      int64_t rows, int64_t L, int64_t train_rows, int64_t val_rows, int64_t* counts6, double* loss) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && pfx && labels && counts6 && loss, "cognn_metrics_q16: bad arguments");
    if (int rc = cg_zero(ctx, counts6, 6 * sizeof(int64_t))) return rc;
    if (int rc = cg_zero(ctx, loss, sizeof(double))) return rc;
    if (rows <= 0) return 0;
    CG_REQUIRE(L > 0 && L <= 64, "cognn_metrics_q16: unsupported label count %lld (max 64)", (long long)L);
    int G = 1;
    while (G < L) G <<= 1;
#define CG_MT_CASE(g)                                                                                                     \
    case g:                                                                                                                \
        hipLaunchKernelGGL(metrics_kernel<g>, dim3(std::min(cg_div_up(rows, kThreads), 512)), dim3(kThreads), 0, ctx->stream, (const u64*)pfx, \
                           labels, border, rows, (int)L, train_rows, val_rows, (unsigned long long*)counts6, loss);       \
        break;
    switch (G) { CG_MT_CASE(1) CG_MT_CASE(2) CG_MT_CASE(4) CG_MT_CASE(8) CG_MT_CASE(16) CG_MT_CASE(32) CG_MT_CASE(64) }
#undef CG_MT_CASE
    CG_LAUNCH_CHECK();
    return 0;
}

int cognn_softmax_jobs_u64(cognn_ctx* ctx, const cognn_softmax_job* jobs, int32_t count, int64_t L) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && (count == 0 || jobs) && count >= 0, "cognn_softmax_jobs_u64: bad arguments");
    CG_REQUIRE(L > 0 && L <= 64, "cognn_softmax_jobs_u64: unsupported label count %lld (max 64)", (long long)L);
    int G = 1;
    while (G < L) G <<= 1;
    int64_t rows_all = 0;
    for (int32_t c = 0; c < count; ++c) rows_all += std::max<int64_t>(jobs[c].rows, 0);
    const int rpg = (rows_all * G <= 2048ll * kThreads) ? 1 : G;     // small inputs: one row per lane group
    for (int32_t c0 = 0; c0 < count; c0 += kSoftmaxJobsMax) {
        SoftmaxBatch b;
        b.count = 0;
        for (int32_t c = c0; c < count && c < c0 + kSoftmaxJobsMax; ++c) {
            const cognn_softmax_job& s = jobs[c];
            CG_REQUIRE(s.d_out && (s.p == 0 || s.p == 1) && s.rows >= 0, "cognn_softmax_jobs_u64: job %d is malformed", c);
            CG_REQUIRE(s.p == 1 || (s.z0 && s.labels && s.counts6 && s.loss), "cognn_softmax_jobs_u64: owner job %d needs z0, labels, counts6, loss", c);
            SoftmaxJobDev& d = b.j[b.count];
            d.d_out = (u64*)s.d_out; d.z0 = (const u64*)s.z0; d.z1 = (const u64*)s.z1; d.labels = s.labels; d.border = s.border;
            d.keyRho = s.keys.k[COGNN_SL_RHO]; d.p = s.p; d.rows = s.rows; d.train_rows = s.train_rows; d.val_rows = s.val_rows;
            d.counts = (unsigned long long*)s.counts6; d.loss = s.loss;
            const unsigned blocks = (unsigned)cg_div_up(cg_div_up(s.rows, rpg) * (int64_t)G, kThreads);   // a lane group of G lanes owns rpg rows
            b.blk_end[b.count] = (b.count ? b.blk_end[b.count - 1] : 0u) + blocks;
            ++b.count;
        }
        hipLaunchKernelGGL(softmax_jobs_zero_kernel, dim3(1), dim3(128), 0, ctx->stream, b);
        CG_LAUNCH_CHECK();
        const unsigned blocks = b.blk_end[b.count - 1];
        if (blocks == 0) continue;
#define CG_SJ_CASE(g) case g: hipLaunchKernelGGL(softmax_jobs_kernel<g>, dim3(blocks), dim3(kThreads), 0, ctx->stream, b, (int)L, rpg); break;
        switch (G) { CG_SJ_CASE(1) CG_SJ_CASE(2) CG_SJ_CASE(4) CG_SJ_CASE(8) CG_SJ_CASE(16) CG_SJ_CASE(32) CG_SJ_CASE(64) }
#undef CG_SJ_CASE
        CG_LAUNCH_CHECK();
    }
    return 0;
}

int cognn_pair_chain_u64(cognn_ctx* ctx, const cognn_pair_chain* chains, int32_t count) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && (count == 0 || chains) && count >= 0, "cognn_pair_chain_u64: bad arguments");
    PairBatch b;
    b.count = 0;
    int nstream = 0;
    int nminimal = 0;
    for (int32_t c = 0; c < count; ++c) { if (chains[c].dealt) ++nstream; if (chains[c].dealt && (chains[c].flags & COGNN_PC_DEALT_MINIMAL)) ++nminimal; }
    CG_REQUIRE(nstream == 0 || nstream == count, "cognn_pair_chain_u64: either every chain of a call brings its dealt values or none does");
    CG_REQUIRE(nminimal == 0 || nminimal == count, "cognn_pair_chain_u64: COGNN_PC_DEALT_MINIMAL on every chain of a call or on none");
    auto launch = [&]() -> int {
        if (b.count == 0) return 0;
        if (nstream && nminimal) hipLaunchKernelGGL(pair_chain_kernel<2>, dim3(b.blk_end[b.count - 1]), dim3(kThreads), 0, ctx->stream, b);
        else if (nstream) hipLaunchKernelGGL(pair_chain_kernel<1>, dim3(b.blk_end[b.count - 1]), dim3(kThreads), 0, ctx->stream, b);
        else hipLaunchKernelGGL(pair_chain_kernel<0>, dim3(b.blk_end[b.count - 1]), dim3(kThreads), 0, ctx->stream, b);
        b.count = 0;
        CG_LAUNCH_CHECK();
        return 0;
    };
    for (int32_t c = 0; c < count; ++c) {
        const cognn_pair_chain& s = chains[c];
        const int64_t n = s.rows * s.F;
        const int fl = s.flags;
        CG_REQUIRE(s.rows >= 0 && s.F >= 0 && n < (1ll << 32), "cognn_pair_chain_u64: chain %d: tensor too large", c);
        CG_REQUIRE((fl & (COGNN_PC_TRUNC_IN | COGNN_PC_SCALE | COGNN_PC_RELU)) != 0, "cognn_pair_chain_u64: chain %d has no step", c);
        CG_REQUIRE(!((fl & COGNN_PC_INPUT_OPENED) && (!(fl & COGNN_PC_SCALE) || (fl & COGNN_PC_TRUNC_IN))),
                   "cognn_pair_chain_u64: chain %d: an opened input must feed the row scale directly", c);
        if (n == 0) continue;
        CG_REQUIRE(s.x[0] && s.x[1] && al(s.x[0]) && al(s.x[1]) && al(s.out[0]) && al(s.out[1]) && al(s.open[0]) && al(s.open[1]) && al(s.c1),
                   "cognn_pair_chain_u64: chain %d: null or misaligned tensor", c);
        CG_REQUIRE(!(fl & COGNN_PC_TRUNC_IN) || (fl & COGNN_PC_NO_C) || s.c1, "cognn_pair_chain_u64: chain %d needs side 1's product share", c);
        CG_REQUIRE(!(fl & COGNN_PC_SCALE) || (s.scale[0] && s.scale[1]), "cognn_pair_chain_u64: chain %d needs both scale shares", c);
        CG_REQUIRE(!(fl & COGNN_PC_OPEN_SUM) || !s.open[1], "cognn_pair_chain_u64: chain %d: COGNN_PC_OPEN_SUM writes open[0] only", c);
        PairChainDev d;
        d.x0 = (const u64*)s.x[0]; d.x1 = (const u64*)s.x[1]; d.c1 = (const u64*)s.c1; d.sc0 = (const u64*)s.scale[0]; d.sc1 = (const u64*)s.scale[1];
        d.out0 = (u64*)s.out[0]; d.out1 = (u64*)s.out[1]; d.open0 = (u64*)s.open[0]; d.open1 = (u64*)s.open[1]; d.mask = s.mask;
        d.mask_in = s.mask_in;
        CG_REQUIRE(!s.mask_in || !(fl & COGNN_PC_INPUT_OPENED), "cognn_pair_chain_u64: chain %d: mask_in needs plain input shares", c);
        CG_REQUIRE(!(fl & COGNN_PC_MASK_AFTER_TRUNC) || ((fl & COGNN_PC_TRUNC_IN) && s.mask_in), "cognn_pair_chain_u64: chain %d: COGNN_PC_MASK_AFTER_TRUNC wants COGNN_PC_TRUNC_IN and mask_in", c);
        CG_REQUIRE(!(fl & COGNN_PC_CLEAR_INPUT) || ((fl & COGNN_PC_TRUNC_IN) && s.x[0] != s.out[0] && s.x[1] != s.out[1] && s.x[0] != s.out[1] && s.x[1] != s.out[0]),
                   "cognn_pair_chain_u64: chain %d: COGNN_PC_CLEAR_INPUT is for product buffers that are not also an output", c);
        pair_chain_fill_keys(d, s);
        d.n = n; d.F = (uint32_t)std::max<int64_t>(s.F, 1); d.flags = (uint32_t)fl;
        d.slab = (const u64*)s.dealt;
        const int64_t pairs = (n + 1) / 2;
        const unsigned blocks = (unsigned)((pairs + kThreads - 1) / kThreads);
        const unsigned base = b.count ? b.blk_end[b.count - 1] : 0u;
        if ((uint64_t)base + blocks > 0x7fffffffull) { const int rc = launch(); if (rc) return rc; }
        b.d[b.count] = d;
        b.blk_end[b.count] = (b.count ? b.blk_end[b.count - 1] : 0u) + blocks;
        if (++b.count == kPairBatchMax) { const int rc = launch(); if (rc) return rc; }
    }
    return launch();
}

int cognn_pair_weight_update_u64(cognn_ctx* ctx, const cognn_pair_wupdate* jobs, int32_t count, const cognn_keys* avg_keys, uint64_t avg_mul,
                                 int32_t average) {
    CG_REQUIRE(ctx && (count == 0 || jobs) && count >= 0, "cognn_pair_weight_update_u64: bad arguments");
    CG_REQUIRE(!average || (count >= 1 && count <= kWUpdateMax && (avg_mul == 0 || avg_keys)),
               "cognn_pair_weight_update_u64: the averaging form takes 1..%d pairs (and the keys of its scale)", kWUpdateMax);
    int rc;
    if ((rc = cg_flush(ctx))) return rc;
    PairWUpdateBatch b; b.count = 0; b.amul = 0; b.aR = b.aR0 = b.aRP0 = 0;
    auto flush = [&]() {
        if (!b.count) return;
        hipLaunchKernelGGL(pair_wupdate_kernel, dim3(b.blk_end[b.count - 1]), dim3(kThreads), 0, ctx->stream, b);
        b.count = 0;
    };
    for (int c = 0; c < count; ++c) {
        const cognn_pair_wupdate& s = jobs[c];
        CG_REQUIRE(s.n >= 0 && s.n < (1ll << 32), "cognn_pair_weight_update_u64: job %d: matrix too large", c);
        CG_REQUIRE(!average || (s.n == jobs[0].n && s.n > 0), "cognn_pair_weight_update_u64: job %d: the averaged matrices must have one (non-zero) size", c);
        if (s.n == 0) continue;
        const bool addc = !(s.flags & COGNN_PC_NO_C);
        CG_REQUIRE(s.z[0] && s.z[1] && s.W[0] && s.W[1] && al(s.z[0]) && al(s.z[1]) && al(s.W[0]) && al(s.W[1]) && al(s.c1),
                   "cognn_pair_weight_update_u64: job %d: null or misaligned tensor", c);
        CG_REQUIRE(!addc || s.c1, "cognn_pair_weight_update_u64: job %d needs side 1's product share", c);
        CG_REQUIRE(s.W[0] != s.W[1], "cognn_pair_weight_update_u64: job %d: the two sides share a weight buffer", c);
        PairWUpdateDev& d = b.d[b.count];
        d.z0 = (const u64*)s.z[0]; d.z1 = (const u64*)s.z[1]; d.c1 = (const u64*)s.c1; d.W0 = (u64*)s.W[0]; d.W1 = (u64*)s.W[1];
        d.keyC0 = s.gemm_keys.k[COGNN_SL_C0];
        for (int t = 0; t < 4; ++t) {
            d.kR[t] = s.trunc_keys[t].k[COGNN_SL_R]; d.kR0[t] = s.trunc_keys[t].k[COGNN_SL_R0]; d.kRP0[t] = s.trunc_keys[t].k[COGNN_SL_RP0];
        }
        for (int t = 0; t < 3; ++t) d.mul[t] = s.mul[t];
        d.n = s.n; d.addc = addc ? 1 : 0; d.swap = (s.flags & COGNN_WU_SWAP) ? 1 : 0; d.clear = (s.flags & COGNN_WU_CLEAR_Z) ? 1 : 0;
        const unsigned blocks = (unsigned)(((s.n + 1) / 2 + kThreads - 1) / kThreads);
        b.blk_end[b.count] = (b.count ? b.blk_end[b.count - 1] : 0u) + blocks;
        ++b.count;
        if (!average && b.count == kWUpdateMax) flush();
    }
    if (average) {
        if (avg_mul) { b.amul = avg_mul; b.aR = avg_keys->k[COGNN_SL_R]; b.aR0 = avg_keys->k[COGNN_SL_R0]; b.aRP0 = avg_keys->k[COGNN_SL_RP0]; }
        hipLaunchKernelGGL(pair_wupdate_avg_kernel, dim3(b.blk_end[0]), dim3(kThreads), 0, ctx->stream, b);
    } else flush();
    CG_LAUNCH_CHECK();
    return 0;
}
int64_t cognn_pair_chain_dealt_slots(int32_t flags, int32_t has_open) {
    return (int64_t)pc_slot_base((uint32_t)flags, has_open != 0).total;
}
int cognn_pair_chain_deal_u64(cognn_ctx* ctx, const cognn_pair_chain* chain, uint64_t* dealt) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && chain && dealt && al(dealt), "cognn_pair_chain_deal_u64: bad arguments");
    const int64_t n = chain->rows * chain->F;
    CG_REQUIRE(chain->rows >= 0 && chain->F >= 0 && n < (1ll << 32), "cognn_pair_chain_deal_u64: tensor too large");
    if (n == 0) return 0;
    PairChainDev d;
    memset(&d, 0, sizeof(d));
    pair_chain_fill_keys(d, *chain);
    d.n = n; d.F = (uint32_t)std::max<int64_t>(chain->F, 1); d.flags = (uint32_t)chain->flags;
    hipLaunchKernelGGL(pair_chain_deal_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, d, (u64*)dealt,
                       (chain->open[0] || chain->open[1]) ? 1 : 0);
    CG_LAUNCH_CHECK();
    return 0;
}

int cognn_transpose_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* in, int64_t rows, int64_t cols) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && out && in && out != in, "cognn_transpose_u64: bad arguments");
    if (rows * cols <= 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3(cg_div_up(rows * cols, kThreads)), dim3(kThreads), 0, ctx->stream, (u64*)out,
                       (const u64*)in, rows, cols);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// device address of this translation unit's copy of the epoch salt (cognn_spec.h), for cognn_set_epoch_salt
void* cg_salt_symbol_kernels_elementwise() {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(cognn_epoch_salt_dev)) == hipSuccess ? p : nullptr;
}
