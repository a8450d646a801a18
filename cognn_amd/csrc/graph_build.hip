// Index construction on the device for a run whose parties all live in ONE process (include/cognn_hip.h,
// cognn_graph_build_colocated): degree accounting (include/graph.h:607-633, include/graph_io_util.h:167-177), the
// dummy-source rule of onPreprocessClient (ss_...h:411-418) and the aggregate CSR of the fused message passing
// straight from the edge list: count -> scan -> fill, no sort.  The order of the entries inside a CSR row is whatever the
// atomics produce: the gather adds uint64 shares, so any order gives the same bits.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include "../../include/cognn_hip.h"

namespace {

// table row of vertex v in the owner-side (A) / co-party-side (B) segment of its party
struct Layout {
    const int32_t* tid; const uint32_t* row_of_vid; const int64_t* a_off; const int64_t* b_off;
    __device__ uint32_t a(uint32_t v) const { return (uint32_t)(a_off[tid[v]] + row_of_vid[v]); }
    __device__ uint32_t b(uint32_t v) const { return (uint32_t)(b_off[tid[v]] + row_of_vid[v]); }
};

template <bool FILL>
__global__ __launch_bounds__(256) void edge_pass_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int undirected,
                                                         int64_t V, Layout L, uint32_t* in_deg, uint32_t* out_deg, uint8_t* border, uint32_t* local_in,
                                                         uint32_t* count, const uint32_t* __restrict__ rowptr, uint32_t* cursor, uint32_t* col,
                                                         int* bad) {
    const int64_t total = undirected ? 2 * E : E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t e = undirected ? i >> 1 : i;
        int64_t s = src[e], d = dst[e];
        if (undirected && (i & 1)) { const int64_t t = s; s = d; d = t; }       // graph_io_util.h:161-163
        if (s < 0 || s >= V || d < 0 || d >= V) { *bad = 1; continue; }
        const uint32_t u = (uint32_t)s, v = (uint32_t)d;
        const bool same = L.tid[u] == L.tid[v];
        // the owner side of v aggregates its own party's shares for local sources and the CO-share replica of the source's
        // party otherwise; the co-party side the other way round (ss_...h:1063-1067, 1089-1100)
        const uint32_t rowA = L.a(v), rowB = L.b(v);
        const uint32_t colA = same ? L.a(u) : L.b(u);
        const uint32_t colB = same ? L.b(u) : L.a(u);
        if (!FILL) {
            atomicAdd(&out_deg[u], 1u);
            atomicAdd(&in_deg[v], 1u);
            if (same) atomicAdd(&local_in[v], 1u); else border[u] = 1;
            atomicAdd(&count[rowA], 1u);
            atomicAdd(&count[rowB], 1u);
        } else {
            col[rowptr[rowA] + atomicAdd(&cursor[rowA], 1u)] = colA;
            col[rowptr[rowB] + atomicAdd(&cursor[rowB], 1u)] = colB;
        }
    }
}

// The atomics of the fill pass leave the entries of a row in a run-dependent order.  Results do not depend on it, but the
// gather's memory behaviour does (measured: 0.82-0.88 of the HBM peak between runs of the same build), so every row is put
// into ONE fixed order: ascending by a hash of the source row - deterministic, and spread over the table (a source-sorted row
// walks the table's party segments one after the other, a hashed one touches them in mixed order).
__device__ __forceinline__ uint32_t row_order_key(uint32_t c) {
    c ^= c >> 16; c *= 0x7feb352du; c ^= c >> 15; c *= 0x846ca68bu; c ^= c >> 16;
    return c;
}
__device__ __forceinline__ unsigned long long row_order_full_key(uint32_t c) { return ((unsigned long long)row_order_key(c) << 32) | c; }
// Rows of up to kSmallRow entries: one thread each, insertion sort.  Longer rows (hubs of a skewed graph: a single lane would
// run O(d^2) dependent loads) are only listed here and sorted by a whole workgroup each in row_order_big_kernel.
constexpr uint32_t kSmallRow = 64;
__global__ __launch_bounds__(256) void row_order_kernel(const uint32_t* __restrict__ rowptr, uint32_t* col, int64_t rows, uint32_t* big_rows,
                                                         uint32_t* big_count, uint32_t big_cap) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const uint32_t b = rowptr[r], e = rowptr[r + 1];
    if (e - b > kSmallRow) {
        const uint32_t slot = atomicAdd(big_count, 1u);
        if (slot < big_cap) big_rows[slot] = (uint32_t)r;    // (cap = entries / kSmallRow + 1: cannot overflow)
        return;
    }
    for (uint32_t i = b + 1; i < e; ++i) {                   // insertion sort (rows hold about average-degree entries)
        const uint32_t c = col[i], kc = row_order_key(c);
        uint32_t j = i;
        while (j > b) {
            const uint32_t p = col[j - 1], kp = row_order_key(p);
            if (kp < kc || (kp == kc && p <= c)) break;
            col[j] = p; --j;
        }
        col[j] = c;
    }
}
// One workgroup per long row: bitonic network with all comparators ascending (positions beyond the row count as +infinity,
// so comparators that reach them are skipped), in LDS when the row fits, in place in global memory otherwise.
constexpr uint32_t kLdsRow = 8192;
__global__ __launch_bounds__(256) void row_order_big_kernel(const uint32_t* __restrict__ rowptr, uint32_t* col, const uint32_t* __restrict__ big_rows,
                                                             const uint32_t* __restrict__ big_count, uint32_t big_cap) {
    __shared__ uint32_t s_col[kLdsRow];
    const uint32_t nbig = min(*big_count, big_cap);
    for (uint32_t w = blockIdx.x; w < nbig; w += gridDim.x) {
        const uint32_t r = big_rows[w];
        const uint32_t b = rowptr[r], n = rowptr[r + 1] - b;
        uint32_t* a = col + b;
        const bool lds = n <= kLdsRow;
        __syncthreads();
        if (lds) { for (uint32_t i = threadIdx.x; i < n; i += 256) s_col[i] = a[i]; a = s_col; }
        __syncthreads();
        uint32_t np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (uint32_t k = 2; k <= np2; k <<= 1) {
            for (uint32_t j = k >> 1, first = 1; j > 0; j >>= 1, first = 0) {
                for (uint32_t i = threadIdx.x; i < np2; i += 256) {
                    const uint32_t l = first ? (i ^ (k - 1)) : (i ^ j);
                    if (l > i && l < n) {
                        const uint32_t x = a[i], y = a[l];
                        if (row_order_full_key(x) > row_order_full_key(y)) { a[i] = y; a[l] = x; }
                    }
                }
                __syncthreads();
            }
        }
        if (lds) for (uint32_t i = threadIdx.x; i < n; i += 256) col[b + i] = s_col[i];
    }
}

// vertices without a local in-edge get the dummy self source: it contributes nothing but inflates both degrees (ss_...h:411-418)
__global__ __launch_bounds__(256) void dummy_rule_kernel(int64_t V, const uint32_t* __restrict__ local_in, const uint32_t* __restrict__ true_in,
                                                          uint32_t* in_deg, uint32_t* out_deg, uint8_t* self_dummy) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const bool dummy = local_in[v] == 0;
    self_dummy[v] = dummy ? 1 : 0;
    in_deg[v] = true_in[v] + (dummy ? 1u : 0u);
    if (dummy) out_deg[v] += 1u;
}

}  // namespace

extern "C" int cognn_graph_build_colocated(cognn_ctx* ctx, int64_t V, int64_t E, int32_t undirected, const int64_t* src, const int64_t* dst,
                                           const int32_t* tid, const uint32_t* row_of_vid, const int64_t* a_off, const int64_t* b_off,
                                           int64_t table_rows, uint32_t* rowptr, uint32_t* col, uint32_t* true_in_deg, uint32_t* in_deg,
                                           uint32_t* out_deg, uint8_t* is_border, uint8_t* self_dummy, uint32_t* scratch) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && tid && row_of_vid && a_off && b_off && rowptr && true_in_deg && in_deg && out_deg && is_border && self_dummy && scratch,
               "cognn_graph_build_colocated: null argument");
    CG_REQUIRE(V >= 0 && E >= 0 && (E == 0 || (src && dst && col)), "cognn_graph_build_colocated: bad edge list");
    const int64_t total = undirected ? 2 * E : E;
    CG_REQUIRE(2 * total < (1ll << 32) && table_rows < (1ll << 32), "cognn_graph_build_colocated: graph too large for 32-bit CSR indices");
    uint32_t* local_in = scratch;                              // [V]
    uint32_t* count = scratch + V;                             // [table_rows + 1]
    uint32_t* cursor = count + table_rows + 1;                 // [table_rows]
    int* bad = (int*)(cursor + table_rows);                    // [1]
    CG_HIP(hipMemsetAsync(scratch, 0, (size_t)(V + 2 * table_rows + 2) * 4, ctx->stream));
    CG_HIP(hipMemsetAsync(true_in_deg, 0, (size_t)V * 4, ctx->stream));
    CG_HIP(hipMemsetAsync(out_deg, 0, (size_t)V * 4, ctx->stream));
    CG_HIP(hipMemsetAsync(is_border, 0, (size_t)V, ctx->stream));
    Layout L{tid, row_of_vid, a_off, b_off};
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 64);
    if (total > 0) {
        hipLaunchKernelGGL(edge_pass_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, src, dst, E, (int)undirected, V, L, true_in_deg, out_deg,
                           is_border, local_in, count, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, bad);
        CG_LAUNCH_CHECK();
    }
    // row pointers = exclusive scan of the per-row counts (count has table_rows + 1 entries, the last one 0)
    size_t tmp_bytes = 0;
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, count, rowptr, (int)(table_rows + 1), ctx->stream));
    void* tmp = nullptr;
    CG_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    hipError_t se = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, count, rowptr, (int)(table_rows + 1), ctx->stream);
    if (se != hipSuccess) { (void)hipFree(tmp); return cognn_set_error("cognn_graph_build_colocated: scan failed: %s", hipGetErrorString(se)); }
    if (total > 0) {
        hipLaunchKernelGGL(edge_pass_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, src, dst, E, (int)undirected, V, L, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)rowptr, cursor, col, bad);
    }
    uint32_t* big = nullptr;                                  // [1 + cap]: count, then the rows longer than kSmallRow
    if (total > 0 && table_rows > 0) {
        const uint32_t big_cap = (uint32_t)(2 * total / kSmallRow + 1);
        hipError_t be = hipMalloc((void**)&big, ((size_t)big_cap + 1) * 4);
        if (be != hipSuccess) { (void)hipFree(tmp); return cognn_set_error("cognn_graph_build_colocated: %s", hipGetErrorString(be)); }
        (void)hipMemsetAsync(big, 0, 4, ctx->stream);
        hipLaunchKernelGGL(row_order_kernel, dim3((unsigned)((table_rows + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)rowptr, col, table_rows,
                           big + 1, big, big_cap);
        hipLaunchKernelGGL(row_order_big_kernel, dim3((unsigned)std::min<uint32_t>(big_cap, 2048u)), dim3(256), 0, ctx->stream, (const uint32_t*)rowptr, col,
                           (const uint32_t*)(big + 1), (const uint32_t*)big, big_cap);
    }
    if (V > 0)
        hipLaunchKernelGGL(dummy_rule_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, ctx->stream, V, local_in, true_in_deg, in_deg, out_deg,
                           self_dummy);
    int host_bad = 0;
    hipError_t ce = hipMemcpyAsync(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (big) (void)hipFree(big);
    if (ce != hipSuccess) return cognn_set_error("cognn_graph_build_colocated: %s", hipGetErrorString(ce));
    CG_LAUNCH_CHECK();
    CG_REQUIRE(!host_bad, "edge list: vertex id out of range");
    return 0;
}
