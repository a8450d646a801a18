// Gather: additive-secret-shared neighbour aggregation as a CSR SpMM with implicit unit values.
//
//   out[r,:] = base[r,:] + sum_{e in row r} table[col[e],:]          (uint64, mod 2^64)
//
// Replaces, fused into one pass over one share, the reference's per-iteration chain
//   OEP localVertexPos -> updateSrcVertexPos   (ss_vertex_centric_algo_kernel.h:752-763)
//   ScatterComp (=copy)                        (optimize-gcn/gcn.h:257-307)
//   prefix_network_aggregate (OGA)             (optimize-gcn/gcn.h:309-342)
//   OEP updateDstVertexPos -> localVertexPos   (ss_...h:818-821, 847-854)
//   twoPartyGCNCondVectorAddition              (optimize-gcn/gcn.h:454-463)
// Rows that the reference masks as dummy simply have no CSR entries.
//
// Kernel shape (HBM-bound): a 256-thread workgroup owns a tile of consecutive
// output rows.  rowptr and the tile's contiguous col slice are staged coalesced into LDS once;
// a group of LPR lanes then owns one output row at a time, each lane holding one 16-byte chunk
// of the row (8-byte when F is odd) and streaming the source rows with independent 16-byte
// loads (unrolled so several row fetches are in flight per lane).
#include "common.h"
#include <algorithm>
#include "../../include/cognn_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileRows = 64;      // output rows per workgroup tile
constexpr int kColCap = 3072;      // staged col entries per tile (12 KiB of LDS)

template <int W> struct Chunk;
template <> struct Chunk<2> {
    u64x2 v;
    __device__ __forceinline__ void zero() { v.x = 0; v.y = 0; }
    __device__ __forceinline__ void load(const u64* p) { v = *reinterpret_cast<const u64x2*>(p); }
    __device__ __forceinline__ void add(const Chunk& o) { v.x += o.v.x; v.y += o.v.y; }
    __device__ __forceinline__ void store(u64* p) const { *reinterpret_cast<u64x2*>(p) = v; }
    __device__ __forceinline__ void sub_prng(u64 key, u64 idx) { v.x -= cognn_prng(key, idx); v.y -= cognn_prng(key, idx + 1); }
};
template <> struct Chunk<1> {
    u64 v;
    __device__ __forceinline__ void zero() { v = 0; }
    __device__ __forceinline__ void load(const u64* p) { v = *p; }
    __device__ __forceinline__ void add(const Chunk& o) { v += o.v; }
    __device__ __forceinline__ void store(u64* p) const { *p = v; }
    __device__ __forceinline__ void sub_prng(u64 key, u64 idx) { v -= cognn_prng(key, idx); }
};

struct OpenSegs {                 // row segments whose output is written as a Beaver opening (value - dealer mask)
    int n;
    int begin[32], end[32];
    u64 key[32];
};

template <int LPR, int W>
__global__ __launch_bounds__(kThreads) void gather_csr_kernel(u64* out, const u64* base,
                                                               const u64* __restrict__ table,
                                                               const uint32_t* __restrict__ rowptr,
                                                               const uint32_t* __restrict__ col, int n_rows, int F, OpenSegs segs) {
    __shared__ uint32_t s_rp[kTileRows + 1];
    __shared__ uint32_t s_col[kColCap];
    constexpr int kGroups = kThreads / LPR;
    const int tid = threadIdx.x;
    const int grp = tid / LPR, ln = tid % LPR;
    const int nchunk = F / W;                       // chunks per row
    const int ntiles = (n_rows + kTileRows - 1) / kTileRows;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = tile * kTileRows;
        const int nr = min(kTileRows, n_rows - r0);
        __syncthreads();                            // previous tile's LDS reads are done
        for (int i = tid; i <= nr; i += kThreads) s_rp[i] = rowptr[r0 + i];
        __syncthreads();
        const uint32_t e0 = s_rp[0], e1 = s_rp[nr];
        const bool staged = (e1 - e0) <= (uint32_t)kColCap;
        if (staged) {
            for (uint32_t i = tid; i < e1 - e0; i += kThreads) s_col[i] = col[e0 + i];
            __syncthreads();
        }
        for (int lr = grp; lr < nr; lr += kGroups) {
            const uint32_t b = s_rp[lr], e = s_rp[lr + 1];
            const size_t orow = (size_t)(r0 + lr) * (size_t)F;
            u64 okey = 0; int obeg = -1;
            for (int sidx = 0; sidx < segs.n; ++sidx)
                if (r0 + lr >= segs.begin[sidx] && r0 + lr < segs.end[sidx]) { okey = segs.key[sidx]; obeg = segs.begin[sidx]; }
            for (int c = ln; c < nchunk; c += LPR) {
                const int off = c * W;
                Chunk<W> acc;
                if (base) acc.load(base + orow + off); else acc.zero();
                uint32_t q = b;
                if (staged) {
                    for (; q + 4 <= e; q += 4) {
                        Chunk<W> t0, t1, t2, t3;
                        const uint32_t c0 = s_col[q - e0], c1 = s_col[q + 1 - e0], c2 = s_col[q + 2 - e0], c3 = s_col[q + 3 - e0];
                        t0.load(table + (size_t)c0 * F + off);
                        t1.load(table + (size_t)c1 * F + off);
                        t2.load(table + (size_t)c2 * F + off);
                        t3.load(table + (size_t)c3 * F + off);
                        t0.add(t1); t2.add(t3); acc.add(t0); acc.add(t2);
                    }
                    for (; q < e; ++q) {
                        Chunk<W> t;
                        t.load(table + (size_t)s_col[q - e0] * F + off);
                        acc.add(t);
                    }
                } else {
                    for (; q < e; ++q) {
                        Chunk<W> t;
                        t.load(table + (size_t)col[q] * F + off);
                        acc.add(t);
                    }
                }
                if (obeg >= 0) acc.sub_prng(okey, (u64)(r0 + lr - obeg) * (u64)F + (u64)off);
                acc.store(out + orow + off);
            }
        }
    }
}

// v[row_index[q],:] += partial[q,:]
template <int LPR, int W>
__global__ __launch_bounds__(kThreads) void scatter_add_rows_kernel(u64* __restrict__ v, const u64* __restrict__ partial,
                                                                     const uint32_t* __restrict__ row_index, int n_partial, int F) {
    constexpr int kGroups = kThreads / LPR;
    const int grp = threadIdx.x / LPR, ln = threadIdx.x % LPR;
    const int nchunk = F / W;
    for (int q = blockIdx.x * kGroups + grp; q < n_partial; q += gridDim.x * kGroups) {
        const size_t dst = (size_t)row_index[q] * F, src = (size_t)q * F;
        for (int c = ln; c < nchunk; c += LPR) {
            Chunk<W> a, b;
            a.load(v + dst + c * W);
            b.load(partial + src + c * W);
            a.add(b);
            a.store(v + dst + c * W);
        }
    }
}

int pick_lpr(int nchunk) {
    int l = 1;
    while (l < nchunk && l < 64) l <<= 1;
    return l;
}

template <int W>
int launch_gather(cognn_ctx* ctx, u64* out, const u64* base, const u64* table, const uint32_t* rowptr, const uint32_t* col,
                  int n_rows, int F, const OpenSegs& segs) {
    const int nchunk = F / W;
    const int lpr = pick_lpr(nchunk);
    const int ntiles = (n_rows + kTileRows - 1) / kTileRows;
    dim3 grid((unsigned)std::min(ntiles, 256 * 16)), block(kThreads);
#define CG_GATHER_CASE(L)                                                                                              \
    case L:                                                                                                             \
        hipLaunchKernelGGL((gather_csr_kernel<L, W>), grid, block, 0, ctx->stream, out, base, table, rowptr, col, n_rows, F, segs); \
        break;
    switch (lpr) {
        CG_GATHER_CASE(1) CG_GATHER_CASE(2) CG_GATHER_CASE(4) CG_GATHER_CASE(8) CG_GATHER_CASE(16) CG_GATHER_CASE(32)
        CG_GATHER_CASE(64)
        default: return cognn_set_error("gather: bad lanes-per-row %d", lpr);
    }
#undef CG_GATHER_CASE
    CG_LAUNCH_CHECK();
    return 0;
}

template <int W>
int launch_scatter(cognn_ctx* ctx, u64* v, const u64* partial, const uint32_t* row_index, int n_partial, int F) {
    const int nchunk = F / W;
    const int lpr = pick_lpr(nchunk);
    const int groups = kThreads / lpr;
    dim3 grid((unsigned)std::min((n_partial + groups - 1) / groups, 256 * 16)), block(kThreads);
#define CG_SCATTER_CASE(L)                                                                                        \
    case L:                                                                                                        \
        hipLaunchKernelGGL((scatter_add_rows_kernel<L, W>), grid, block, 0, ctx->stream, v, partial, row_index, n_partial, F); \
        break;
    switch (lpr) {
        CG_SCATTER_CASE(1) CG_SCATTER_CASE(2) CG_SCATTER_CASE(4) CG_SCATTER_CASE(8) CG_SCATTER_CASE(16) CG_SCATTER_CASE(32)
        CG_SCATTER_CASE(64)
        default: return cognn_set_error("scatter_add: bad lanes-per-row %d", lpr);
    }
#undef CG_SCATTER_CASE
    CG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

static int gather_impl(cognn_ctx* ctx, uint64_t* out, const uint64_t* base, const uint64_t* table, const uint32_t* rowptr,
                       const uint32_t* col, int64_t n_rows, int64_t F, const OpenSegs& segs);

int cognn_gather_csr_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* base, const uint64_t* table,
                         const uint32_t* rowptr, const uint32_t* col, int64_t n_rows, int64_t F) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    OpenSegs segs;
    segs.n = 0;
    return gather_impl(ctx, out, base, table, rowptr, col, n_rows, F, segs);
}

int cognn_gather_csr_open_u64(cognn_ctx* ctx, uint64_t* out, const uint64_t* base, const uint64_t* table,
                              const uint32_t* rowptr, const uint32_t* col, int64_t n_rows, int64_t F,
                              int32_t nseg, const int64_t* seg_begin, const int64_t* seg_end, const uint64_t* seg_key) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(nseg >= 0 && nseg <= 32 && (nseg == 0 || (seg_begin && seg_end && seg_key)), "cognn_gather_csr_open_u64: bad segment list");
    OpenSegs segs;
    segs.n = nseg;
    for (int i = 0; i < nseg; ++i) { segs.begin[i] = (int)seg_begin[i]; segs.end[i] = (int)seg_end[i]; segs.key[i] = seg_key[i]; }
    return gather_impl(ctx, out, base, table, rowptr, col, n_rows, F, segs);
}

static int gather_impl(cognn_ctx* ctx, uint64_t* out, const uint64_t* base, const uint64_t* table, const uint32_t* rowptr,
                       const uint32_t* col, int64_t n_rows, int64_t F, const OpenSegs& segs) {
    CG_REQUIRE(ctx && out && table && rowptr, "cognn_gather_csr_u64: null argument");
    CG_REQUIRE(n_rows >= 0 && n_rows < (1ll << 31) && F > 0 && F < (1 << 20), "cognn_gather_csr_u64: bad shape %lld x %lld",
               (long long)n_rows, (long long)F);
    if (n_rows == 0) return 0;
    const bool vec = (F % 2 == 0) && cg_aligned16(out) && cg_aligned16(table) && (base == nullptr || cg_aligned16(base));
    if (vec) return launch_gather<2>(ctx, (u64*)out, (const u64*)base, (const u64*)table, rowptr, col, (int)n_rows, (int)F, segs);
    return launch_gather<1>(ctx, (u64*)out, (const u64*)base, (const u64*)table, rowptr, col, (int)n_rows, (int)F, segs);
}

int cognn_scatter_add_rows_u64(cognn_ctx* ctx, uint64_t* v, const uint64_t* partial, const uint32_t* row_index,
                               int64_t n_partial, int64_t F) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && v && (n_partial == 0 || (partial && row_index)), "cognn_scatter_add_rows_u64: null argument");
    CG_REQUIRE(n_partial >= 0 && n_partial < (1ll << 31) && F > 0 && F < (1 << 20), "cognn_scatter_add_rows_u64: bad shape");
    if (n_partial == 0) return 0;
    const bool vec = (F % 2 == 0) && cg_aligned16(v) && cg_aligned16(partial);
    if (vec) return launch_scatter<2>(ctx, (u64*)v, (const u64*)partial, row_index, (int)n_partial, (int)F);
    return launch_scatter<1>(ctx, (u64*)v, (const u64*)partial, row_index, (int)n_partial, (int)F);
}

}  // extern "C"
