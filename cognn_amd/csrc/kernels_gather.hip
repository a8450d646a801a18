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
#include <cstring>
#include "../../include/cognn_hip.h"
#include "pair_chain.h"

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
    __device__ __forceinline__ void get(u64* a) const { a[0] = v.x; a[1] = v.y; }
    static __device__ __forceinline__ void put(u64* p, const u64* a) { u64x2 t; t.x = a[0]; t.y = a[1]; *reinterpret_cast<u64x2*>(p) = t; }
};
template <> struct Chunk<1> {
    u64 v;
    __device__ __forceinline__ void zero() { v = 0; }
    __device__ __forceinline__ void load(const u64* p) { v = *p; }
    __device__ __forceinline__ void add(const Chunk& o) { v += o.v; }
    __device__ __forceinline__ void store(u64* p) const { *p = v; }
    __device__ __forceinline__ void sub_prng(u64 key, u64 idx) { v -= cognn_prng(key, idx); }
    __device__ __forceinline__ void get(u64* a) const { a[0] = v; a[1] = 0; }
    static __device__ __forceinline__ void put(u64* p, const u64* a) { *p = a[0]; }
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

// ---- gather with the pair chain as its epilogue (cognn_gather_pair_chain_u64) ---------------------------------------------------
// A tile is 32 vertices of one owner; the lane group that owns vertex r aggregates the owner-side row and the co-party-side
// row one after the other (same 16-byte chunk of both), then runs the chain on the two sums in registers.
constexpr int kPairTile = 32;
constexpr int kPairColCap = 1536;                            // staged col entries per side and tile
constexpr int kGatherPairsMax = 8;
struct GatherPairSeg {
    PairChainDev d;                                          // x0/x1/c1 unused; n = rows * F
    int a_row0, b_row0, rows, tile_end;                      // tile_end: exclusive prefix of tile counts
};
struct GatherPairBatch {
    GatherPairSeg s[kGatherPairsMax];
    int count;
};
// the prediction layer of a pair as the launch's second epilogue (cognn_gather_pair::softmax): what softmax_jobs_kernel does for an
// owner job and its co-party job, on the chain's results in registers
struct SoftmaxPairDev {
    u64* d_out0; u64* d_out1; const int32_t* labels; const uint8_t* border;
    u64 keyRho; int64_t train_rows, val_rows;
    unsigned long long* counts; double* loss;
};
struct GatherSoftmaxBatch { SoftmaxPairDev s[kGatherPairsMax]; };
__global__ void gather_softmax_zero_kernel(GatherSoftmaxBatch q, int count) {
    const int t = threadIdx.x;
    if (t < count * 8) {
        if ((t & 7) < 6) q.s[t >> 3].counts[t & 7] = 0ull;
        else if ((t & 7) == 6) *q.s[t >> 3].loss = 0.0;
    }
}
// W: u64 per lane and step - 2 (16-byte accesses) for an even width, 1 for an odd one (7 or 3 labels)
// SMX: the prediction layer follows as a second epilogue (LPR lanes hold a row of F <= LPR * W logits: row reductions by shuffles)
template <int LPR, int W, int STREAM, bool SMX = false>
__device__ __forceinline__ void gather_pair_chain_body(const u64* __restrict__ table, const u64* __restrict__ base, const uint32_t* __restrict__ rowptr,
                                                       const uint32_t* __restrict__ col, int F, int ntiles, const GatherPairBatch& b,
                                                       uint32_t (&s_rp)[2][kPairTile + 1], uint32_t (&s_col)[2][kPairColCap],
                                                       const GatherSoftmaxBatch* sm = nullptr, unsigned long long* s_cnt = nullptr,
                                                       double* s_loss = nullptr) {
    constexpr int kGroups = kThreads / LPR;
    const int tid = threadIdx.x;
    const int grp = tid / LPR, ln = tid % LPR;
    const int nchunk = F / W;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int seg = 0;
        while (seg < b.count - 1 && tile >= b.s[seg].tile_end) ++seg;
        const GatherPairSeg& S = b.s[seg];
        const PairChainDev& d = S.d;
        const int t0 = seg ? b.s[seg - 1].tile_end : 0;
        const int v0row = (tile - t0) * kPairTile;           // first vertex (row inside the owner's tensors) of this tile
        const int nr = min(kPairTile, S.rows - v0row);
        const int rbase[2] = {S.a_row0 + v0row, S.b_row0 + v0row};
        __syncthreads();                                     // previous tile's LDS reads are done
        for (int i = tid; i < 2 * (nr + 1); i += kThreads) { const int sd = i / (nr + 1), j = i % (nr + 1); s_rp[sd][j] = rowptr[rbase[sd] + j]; }
        if (SMX) { if (tid < 5) s_cnt[tid] = 0ull; if (tid == 0) *s_loss = 0.0; }
        __syncthreads();
        bool staged[2];
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const uint32_t e0 = s_rp[sd][0], e1 = s_rp[sd][nr];
            staged[sd] = (e1 - e0) <= (uint32_t)kPairColCap;
            if (staged[sd]) for (uint32_t i = tid; i < e1 - e0; i += kThreads) s_col[sd][i] = col[e0 + i];
        }
        __syncthreads();
        for (int lr = grp; lr < nr; lr += kGroups) {
            const int vr = v0row + lr;                       // vertex row
            for (int c = ln; c < (SMX ? LPR : nchunk); c += LPR) {   // (SMX: every lane of the group runs this once - the shuffles below)
                const bool have = !SMX || c < nchunk;
                const int off = c * W;
                const u64 idx = (u64)vr * (u64)F + (u64)off;
                u64 a[2] = {0, 0}, bb[2] = {0, 0};
                if (have) {
                Chunk<W> sum[2];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd) {
                    const uint32_t e0 = s_rp[sd][0], bq = s_rp[sd][lr], eq = s_rp[sd][lr + 1];
                    Chunk<W> acc;
                    acc.load((base ? base : table) + (size_t)(rbase[sd] + lr) * F + off);   // the self row (or the sums a first launch left)
                    uint32_t q = bq;
                    if (staged[sd]) {
                        for (; q + 4 <= eq; q += 4) {
                            Chunk<W> t0c, t1c, t2c, t3c;
                            const uint32_t c0 = s_col[sd][q - e0], c1 = s_col[sd][q + 1 - e0], c2 = s_col[sd][q + 2 - e0], c3 = s_col[sd][q + 3 - e0];
                            t0c.load(table + (size_t)c0 * F + off);
                            t1c.load(table + (size_t)c1 * F + off);
                            t2c.load(table + (size_t)c2 * F + off);
                            t3c.load(table + (size_t)c3 * F + off);
                            t0c.add(t1c); t2c.add(t3c); acc.add(t0c); acc.add(t2c);
                        }
                        for (; q < eq; ++q) { Chunk<W> t; t.load(table + (size_t)s_col[sd][q - e0] * F + off); acc.add(t); }
                    } else {
                        for (; q < eq; ++q) { Chunk<W> t; t.load(table + (size_t)col[q] * F + off); acc.add(t); }
                    }
                    sum[sd] = acc;
                }
                // the chain on the two sides' sums (pair_chain.h): row scale + truncation [+ ReLU], outputs / openings
                sum[0].get(a); sum[1].get(bb);
                bool pos[2] = {true, true};
                PairRow rw = {0, 0, 0};
                const PcSlotBase SB = pc_slot_base(d.flags, d.open0 != nullptr || d.open1 != nullptr);
                if (d.flags & COGNN_PC_SCALE) rw = pair_row(d, (u64)vr);
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    if (d.flags & COGNN_PC_SCALE) pair_scale<STREAM>(d, SB.sc, idx + j, rw, false, a[j], bb[j]);
                    if (d.flags & COGNN_PC_RELU) pos[j] = pair_relu<STREAM>(d, SB.re, idx + j, a[j], bb[j]);
                }
                if ((d.flags & COGNN_PC_RELU) && d.mask) { d.mask[idx] = pos[0] ? 1 : 0; if (W == 2) d.mask[idx + 1] = pos[1] ? 1 : 0; }
                if (d.out0) Chunk<W>::put(d.out0 + idx, a);
                if (d.out1) Chunk<W>::put(d.out1 + idx, bb);
                if (d.open0 || d.open1) {
                    u64 m0[2], m1[2];
                    m0[1] = m1[1] = 0;
#pragma unroll
                    for (int j = 0; j < W; ++j) pair_open_masks<STREAM>(d, SB.op, idx + j, m0[j], m1[j]);
                    if (d.flags & COGNN_PC_OPEN_SUM) {
                        if (d.open0) { const u64 t[2] = {(a[0] - m0[0]) + (bb[0] - m1[0]), (a[1] - m0[1]) + (bb[1] - m1[1])}; Chunk<W>::put(d.open0 + idx, t); }
                    } else {
                        if (d.open0) { const u64 t[2] = {a[0] - m0[0], a[1] - m0[1]}; Chunk<W>::put(d.open0 + idx, t); }
                        if (d.open1) { const u64 t[2] = {bb[0] - m1[0], bb[1] - m1[1]}; Chunk<W>::put(d.open1 + idx, t); }
                    }
                }
                }
                if (SMX) {
                    // the prediction layer on the row (a + bb = z, the logits): softmax_jobs_kernel's arithmetic with W columns per lane
                    const SoftmaxPairDev& q = sm->s[seg];
                    const long long NEG = -(1ll << 62);
                    long long z[W], e[W];
#pragma unroll
                    for (int j = 0; j < W; ++j) z[j] = have ? (long long)(a[j] + bb[j]) : NEG;
                    long long m = z[0];
                    if (W == 2 && z[W - 1] > m) m = z[W - 1];
#pragma unroll
                    for (int o = LPR / 2; o > 0; o >>= 1) { const long long t = __shfl_xor(m, o, LPR); m = t > m ? t : m; }
                    long long S = 0;
#pragma unroll
                    for (int j = 0; j < W; ++j) { e[j] = have ? cognn_exp_neg_q30(m - z[j]) : 0; S += e[j]; }
#pragma unroll
                    for (int o = LPR / 2; o > 0; o >>= 1) S += __shfl_xor(S, o, LPR);
                    const int lab = q.labels[vr];
                    const bool trn = vr < q.train_rows;
                    u64 pf[W], d0[2] = {0, 0}, d1[2] = {0, 0};
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        pf[j] = have ? div_floor_small((u64)((e[j] << 16) + (S >> 1)), (u64)S) : 0ull;
                        const u64 rho = cognn_prng(q.keyRho, idx + j);
                        d0[j] = trn ? ((pf[j] - rho) - (off + j == lab ? COGNN_FX_ONE : 0ull)) : 0ull;
                        d1[j] = trn ? rho : 0ull;
                    }
                    if (have) { Chunk<W>::put(q.d_out0 + idx, d0); Chunk<W>::put(q.d_out1 + idx, d1); }
                    // metrics on the revealed probabilities: argmax (the first maximum wins ties), p[label]
                    u64 bv = have ? pf[0] : 0ull;
                    int bj = have ? off : (1 << 20);
                    if (W == 2 && have && pf[W - 1] > bv) { bv = pf[W - 1]; bj = off + 1; }
#pragma unroll
                    for (int o = LPR / 2; o > 0; o >>= 1) {
                        const u64 ov = __shfl_xor(bv, o, LPR);
                        const int oj = __shfl_xor(bj, o, LPR);
                        if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
                    }
                    const u64 mine = (W == 2 && (lab & 1)) ? pf[W - 1] : pf[0];
                    const u64 p_lab = __shfl(mine, lab / W, LPR);
                    if (ln == 0) {                           // one lane keeps the row's books
                        const bool te = vr >= q.train_rows + q.val_rows, bd = q.border ? q.border[vr] != 0 : false;
                        if (bj == lab) {
                            atomicAdd(&s_cnt[0], 1ull);
                            if (trn) atomicAdd(&s_cnt[1], 1ull);
                            if (trn && bd) atomicAdd(&s_cnt[2], 1ull);
                            if (te) atomicAdd(&s_cnt[3], 1ull);
                            if (te && bd) atomicAdd(&s_cnt[4], 1ull);
                        }
                        double my_pl = (double)p_lab / (double)COGNN_FX_ONE;
                        if (my_pl == 0.0) my_pl = 0.001;     /* gcn.h:613-615 */
                        const double ls = -log(my_pl);
                        if (ls != 0.0) atomicAdd(s_loss, ls);
                    }
                }
            }
        }
        if (SMX) {                                           // this tile's counts and loss (a tile belongs to one pair)
            __syncthreads();
            const SoftmaxPairDev& q = sm->s[seg];
            if (tid < 5 && s_cnt[tid]) atomicAdd(&q.counts[tid], s_cnt[tid]);
            if (tid == 0 && *s_loss != 0.0) atomicAdd(q.loss, *s_loss);
        }
    }
}

template <int LPR, int STREAM, int W = 2>   // STREAM 1 / 2: every pair brings the dealt slab of its chain, read whole / the corrections only (pair_chain.h)
__global__ __launch_bounds__(kThreads) void gather_pair_chain_kernel(const u64* __restrict__ table, const u64* __restrict__ base,
                                                                      const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, int F,
                                                                      int ntiles, GatherPairBatch b) {
    __shared__ uint32_t s_rp[2][kPairTile + 1];
    __shared__ uint32_t s_col[2][kPairColCap];
    gather_pair_chain_body<LPR, W, STREAM>(table, base, rowptr, col, F, ntiles, b, s_rp, s_col);
}
template <int LPR, int W>
__global__ __launch_bounds__(kThreads) void gather_pair_softmax_kernel(const u64* __restrict__ table, const u64* __restrict__ base,
                                                                        const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, int F,
                                                                        int ntiles, GatherPairBatch b, GatherSoftmaxBatch sm) {
    __shared__ uint32_t s_rp[2][kPairTile + 1];
    __shared__ uint32_t s_col[2][kPairColCap];
    __shared__ unsigned long long s_cnt[5];
    __shared__ double s_loss;
    gather_pair_chain_body<LPR, W, 0, true>(table, base, rowptr, col, F, ntiles, b, s_rp, s_col, &sm, s_cnt, &s_loss);
}
// Grid of the gather kernels: one workgroup per tile up to this cap (the kernels keep their grid-stride loop).  Measured on
// MI355X (config5, fused F = 64 launch): persistent grids lose to the hardware dispatcher - 1792 workgroups (one per slot)
// 2.09 ms average, 4096: 1.93, 7168: 1.82, 28672: 1.78 - tiles differ in cost (degree), and a retiring workgroup's slot is
// refilled at once.  (Forcing 8 waves per SIMD on the fused kernel - 64 VGPRs, 3 spilled - is slower: 1.89 against 1.78 ms.)
int gather_grid(int ntiles) {
    static const int cap = getenv("COGNN_GATHER_GRID_CAP") ? std::max(1, atoi(getenv("COGNN_GATHER_GRID_CAP"))) : (1 << 20);
    return std::min(ntiles, cap);
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
    dim3 grid((unsigned)gather_grid(ntiles)), block(kThreads);
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


// ---- original-gcn: per-edge two-normaliser Scatter + segmented sum + masked add (cognn_scatter_gather_original_u64) ----------
struct ScatterPairDev {
    const u64* srcA; const u64* srcB; const u64* n0; const u64* n1;
    u64 k0[8], k1[8];            // per scale: A0, A1, B0, B1, C0 of the Beaver product, R, R0, RP0 of its truncation
    int n1_from_server, crossed;
};
constexpr int kScatterPairsMax = 16;
struct ScatterBatch {
    ScatterPairDev p[kScatterPairsMax];
    u64 ks[8];                   // the self row's scale (forward iterations)
    const u64* self_s0; const u64* self_s1;
};
// one share x shared-row-scale product + truncation for both share-holders: RowscaleOpen / RowscaleClose / TruncClose of the per-side
// kernels with the opened values handed over in registers (the formulas of pair_scale, pair_chain.h)
__device__ __forceinline__ void scale_trunc_pair(const u64* K, u64 s0, u64 s1, u64 row, u64 idx, u64& v0, u64& v1) {
    const u64 b0 = cognn_prng(K[2], row), b1 = cognn_prng(K[3], row);
    const u64 g = (s0 - b0) + (s1 - b1);
    const u64 a0 = cognn_prng(K[0], idx), a1 = cognn_prng(K[1], idx), c0m = cognn_prng(K[4], idx);
    const u64 c1m = (a0 + a1) * (b0 + b1) - c0m;
    const u64 e = (v0 - a0) + (v1 - a1);
    u64 z0 = e * b0 + a0 * g + c0m;
    u64 z1 = e * g + e * b1 + a1 * g + c1m;
    PairChainDev none;
    none.slab = nullptr; none.n = 0;
    pair_trunc<false>(none, 0, K[5], K[6], K[7], idx, z0, z1);
    v0 = z0; v1 = z1;
}
__global__ __launch_bounds__(256) void scatter_gather_original_kernel(u64* outA, u64* outB, const u64* selfA, const u64* selfB, int rows, int F,
                                                                       const uint32_t* rowptr, const uint32_t* ent_src, const uint32_t* ent_pair,
                                                                       const uint32_t* ent_q, ScatterBatch b) {
    const int r = blockIdx.x;
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (r >= rows || j >= F) return;
    const u64 own = (u64)r * (u64)F + (u64)j;
    u64 a = selfA[own], c = selfB[own];
    if (b.self_s0) scale_trunc_pair(b.ks, b.self_s0[r], b.self_s1[r], (u64)r, own, a, c);
    for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
        const ScatterPairDev& P = b.p[ent_pair[e]];
        const u64 q = ent_q[e], src = (u64)ent_src[e] * (u64)F + (u64)j, idx = q * (u64)F + (u64)j;
        u64 u0 = P.srcA[src], u1 = P.srcB[src];
        scale_trunc_pair(P.k0, P.n0[q], 0ull, q, idx, u0, u1);
        const u64 n1 = P.n1[q];
        scale_trunc_pair(P.k1, P.n1_from_server ? 0ull : n1, P.n1_from_server ? n1 : 0ull, q, idx, u0, u1);
        if (P.crossed) { a += u1; c += u0; } else { a += u0; c += u1; }
    }
    outA[own] = a; outB[own] = c;
}
static void fill_scale_keys(u64* K, const cognn_keys& sk, const cognn_keys& tk) {
    K[0] = sk.k[COGNN_SL_A0]; K[1] = sk.k[COGNN_SL_A1]; K[2] = sk.k[COGNN_SL_B0]; K[3] = sk.k[COGNN_SL_B1]; K[4] = sk.k[COGNN_SL_C0];
    K[5] = tk.k[COGNN_SL_R]; K[6] = tk.k[COGNN_SL_R0]; K[7] = tk.k[COGNN_SL_RP0];
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

int cognn_gather_pair_chain_takes_softmax(int64_t F) {      // the widths cognn_gather_pair::softmax is served for
    return (F >= 1 && ((F & 1) ? F <= 31 : F <= 64)) ? 1 : 0;
}
int cognn_gather_pair_chain_u64(cognn_ctx* ctx, const uint64_t* table, const uint32_t* rowptr, const uint32_t* col, int64_t F,
                                const cognn_gather_pair* pairs, int32_t count) {
    return cognn_gather_pair_chain_base_u64(ctx, table, nullptr, rowptr, col, F, pairs, count);
}
int cognn_gather_pair_chain_base_u64(cognn_ctx* ctx, const uint64_t* table, const uint64_t* base, const uint32_t* rowptr, const uint32_t* col, int64_t F,
                                     const cognn_gather_pair* pairs, int32_t count) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(!base || cg_aligned16(base), "cognn_gather_pair_chain_base_u64: misaligned base");
    CG_REQUIRE(ctx && table && rowptr && (count == 0 || pairs) && count >= 0 && count <= kGatherPairsMax, "cognn_gather_pair_chain_u64: bad arguments");
    CG_REQUIRE(F > 0 && F < (1 << 20) && cg_aligned16(table), "cognn_gather_pair_chain_u64: bad width or misaligned table");
    const bool odd = (F & 1) != 0;                           // 8-byte lanes: the row stride is not a multiple of 16 bytes
    GatherPairBatch b;
    GatherSoftmaxBatch sm;
    b.count = 0;
    int ntiles = 0, nstream = 0, nsoftmax = 0, nminimal = 0;
    for (int32_t c = 0; c < count; ++c) {
        const cognn_gather_pair& p = pairs[c];
        const cognn_pair_chain& s = p.chain;
        if (s.rows <= 0) continue;
        const int fl = s.flags;
        if (p.softmax[0] || p.softmax[1]) {                  // the prediction layer rides along: the logits need not be written
            const cognn_softmax_job* o = p.softmax[0];
            const cognn_softmax_job* t = p.softmax[1];
            CG_REQUIRE(o && t && o->p == 0 && t->p == 1 && o->rows == s.rows && t->rows == s.rows && o->train_rows == t->train_rows,
                       "cognn_gather_pair_chain_u64: pair %d: softmax wants the owner's (p = 0) and the co-party's (p = 1) job of the chain's rows", c);
            CG_REQUIRE(o->d_out && t->d_out && o->labels && o->counts6 && o->loss && (odd || (cg_aligned16(o->d_out) && cg_aligned16(t->d_out))),
                       "cognn_gather_pair_chain_u64: pair %d: softmax job is malformed", c);
            CG_REQUIRE(cognn_gather_pair_chain_takes_softmax(F) && !(fl & COGNN_PC_RELU) && !s.open[0] && !s.open[1] && !s.dealt,
                       "cognn_gather_pair_chain_u64: pair %d: softmax follows a plain or scaled aggregate of at most 64 (odd: 31) labels, without opening or dealt values", c);
            SoftmaxPairDev& q = sm.s[b.count];
            q.d_out0 = (u64*)o->d_out; q.d_out1 = (u64*)t->d_out; q.labels = o->labels; q.border = o->border;
            q.keyRho = o->keys.k[COGNN_SL_RHO]; q.train_rows = o->train_rows; q.val_rows = o->val_rows;
            q.counts = (unsigned long long*)o->counts6; q.loss = o->loss;
            ++nsoftmax;
        }
        CG_REQUIRE(!(fl & (COGNN_PC_TRUNC_IN | COGNN_PC_INPUT_OPENED)), "cognn_gather_pair_chain_u64: pair %d: the aggregate is neither a raw product nor an opening", c);
        CG_REQUIRE((fl & COGNN_PC_SCALE) || !(fl & COGNN_PC_RELU), "cognn_gather_pair_chain_u64: pair %d: a ReLU follows the row scale only", c);
        CG_REQUIRE((!(fl & COGNN_PC_SCALE) || (s.scale[0] && s.scale[1])) && s.rows * F < (1ll << 32) && p.a_row0 >= 0 && p.b_row0 >= 0, "cognn_gather_pair_chain_u64: pair %d is malformed", c);
        CG_REQUIRE(s.out[0] || s.out[1] || s.open[0] || s.open[1] || p.softmax[0], "cognn_gather_pair_chain_u64: pair %d writes nothing", c);
        CG_REQUIRE(!(fl & COGNN_PC_OPEN_SUM) || !s.open[1], "cognn_gather_pair_chain_u64: pair %d: COGNN_PC_OPEN_SUM writes open[0] only", c);
        CG_REQUIRE(odd || (cg_aligned16(s.out[0]) && cg_aligned16(s.out[1]) && cg_aligned16(s.open[0]) && cg_aligned16(s.open[1])), "cognn_gather_pair_chain_u64: pair %d: misaligned output", c);
        GatherPairSeg& g = b.s[b.count];
        PairChainDev& d = g.d;
        d.x0 = d.x1 = d.c1 = nullptr; d.mask_in = nullptr;
        d.sc0 = (const u64*)s.scale[0]; d.sc1 = (const u64*)s.scale[1];
        d.out0 = (u64*)s.out[0]; d.out1 = (u64*)s.out[1]; d.open0 = (u64*)s.open[0]; d.open1 = (u64*)s.open[1]; d.mask = s.mask;
        pair_chain_fill_keys(d, s);
        d.n = s.rows * F; d.F = (uint32_t)F; d.flags = (uint32_t)fl;
        d.slab = (const u64*)s.dealt;
        if (d.slab) ++nstream;
        if (d.slab && (fl & COGNN_PC_DEALT_MINIMAL)) ++nminimal;
        g.a_row0 = (int)p.a_row0; g.b_row0 = (int)p.b_row0; g.rows = (int)s.rows;
        ntiles += (int)((s.rows + kPairTile - 1) / kPairTile);
        g.tile_end = ntiles;
        ++b.count;
    }
    if (ntiles == 0) return 0;
    const int lpr = pick_lpr(odd ? (int)F : (int)(F / 2));
    dim3 grid((unsigned)gather_grid(ntiles)), block(kThreads);
    CG_REQUIRE(nstream == 0 || nstream == b.count, "cognn_gather_pair_chain_u64: either every pair brings its dealt values or none does");
    CG_REQUIRE(nminimal == 0 || (nminimal == b.count && !odd), "cognn_gather_pair_chain_u64: COGNN_PC_DEALT_MINIMAL on every pair or on none (even widths)");
    CG_REQUIRE(nsoftmax == 0 || nsoftmax == b.count, "cognn_gather_pair_chain_u64: either every pair brings its softmax jobs or none does");
    if (nsoftmax) {
        hipLaunchKernelGGL(gather_softmax_zero_kernel, dim3(1), dim3(64), 0, ctx->stream, sm, b.count);
        CG_LAUNCH_CHECK();
#define CG_GS_CASE(L)                                                                                                                              \
    case L:                                                                                                                                        \
        if (odd) hipLaunchKernelGGL((gather_pair_softmax_kernel<L, 1>), grid, block, 0, ctx->stream, (const u64*)table, (const u64*)base, rowptr, col, (int)F, ntiles, b, sm); \
        else hipLaunchKernelGGL((gather_pair_softmax_kernel<L, 2>), grid, block, 0, ctx->stream, (const u64*)table, (const u64*)base, rowptr, col, (int)F, ntiles, b, sm);    \
        break;
        switch (lpr) {
            CG_GS_CASE(1) CG_GS_CASE(2) CG_GS_CASE(4) CG_GS_CASE(8) CG_GS_CASE(16) CG_GS_CASE(32)
            default: return cognn_set_error("gather_pair_chain: bad lanes-per-row %d for the softmax epilogue", lpr);
        }
#undef CG_GS_CASE
        CG_LAUNCH_CHECK();
        return 0;
    }
#define CG_GP_CASE(L)                                                                                                                              \
    case L:                                                                                                                                        \
        if (odd && nstream) hipLaunchKernelGGL((gather_pair_chain_kernel<L, 1, 1>), grid, block, 0, ctx->stream, (const u64*)table, (const u64*)base, rowptr, col, (int)F, ntiles, b); \
        else if (odd) hipLaunchKernelGGL((gather_pair_chain_kernel<L, 0, 1>), grid, block, 0, ctx->stream, (const u64*)table, (const u64*)base, rowptr, col, (int)F, ntiles, b);    \
        else if (nstream && nminimal) hipLaunchKernelGGL((gather_pair_chain_kernel<L, 2>), grid, block, 0, ctx->stream, (const u64*)table, (const u64*)base, rowptr, col, (int)F, ntiles, b); \
        else if (nstream) hipLaunchKernelGGL((gather_pair_chain_kernel<L, 1>), grid, block, 0, ctx->stream, (const u64*)table, (const u64*)base, rowptr, col, (int)F, ntiles, b); \
        else hipLaunchKernelGGL((gather_pair_chain_kernel<L, 0>), grid, block, 0, ctx->stream, (const u64*)table, (const u64*)base, rowptr, col, (int)F, ntiles, b);        \
        break;
    switch (lpr) {
        CG_GP_CASE(1) CG_GP_CASE(2) CG_GP_CASE(4) CG_GP_CASE(8) CG_GP_CASE(16) CG_GP_CASE(32) CG_GP_CASE(64)
        default: return cognn_set_error("gather_pair_chain: bad lanes-per-row %d", lpr);
    }
#undef CG_GP_CASE
    CG_LAUNCH_CHECK();
    return 0;
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

int cognn_scatter_gather_original_u64(cognn_ctx* ctx, uint64_t* outA, uint64_t* outB, const uint64_t* selfA, const uint64_t* selfB,
                                      const uint64_t* self_scale0, const uint64_t* self_scale1, const cognn_keys* self_scale_keys,
                                      const cognn_keys* self_trunc_keys, int64_t rows, int64_t F, const uint32_t* rowptr,
                                      const uint32_t* ent_src, const uint32_t* ent_pair, const uint32_t* ent_q,
                                      const cognn_scatter_pair* pairs, int32_t npairs) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && outA && outB && selfA && selfB && rowptr && npairs >= 0 && npairs <= kScatterPairsMax && (npairs == 0 || pairs),
               "cognn_scatter_gather_original_u64: bad arguments (at most %d Scatter instances per destination party)", kScatterPairsMax);
    CG_REQUIRE(rows >= 0 && rows < (1ll << 31) && F > 0 && F < (1 << 24), "cognn_scatter_gather_original_u64: bad shape");
    CG_REQUIRE(!self_scale0 || (self_scale1 && self_scale_keys && self_trunc_keys), "cognn_scatter_gather_original_u64: the self scale needs both shares and its keys");
    CG_REQUIRE(outA != selfA && outB != selfB, "cognn_scatter_gather_original_u64: in-place aggregation is not possible (rows are sources too)");
    if (rows == 0) return 0;
    ScatterBatch b;
    memset(&b, 0, sizeof(b));
    for (int32_t i = 0; i < npairs; ++i) {
        const cognn_scatter_pair& s = pairs[i];
        CG_REQUIRE(s.srcA && s.srcB && s.n0 && s.n1, "cognn_scatter_gather_original_u64: instance %d has a null tensor", i);
        ScatterPairDev& d = b.p[i];
        d.srcA = (const u64*)s.srcA; d.srcB = (const u64*)s.srcB; d.n0 = (const u64*)s.n0; d.n1 = (const u64*)s.n1;
        fill_scale_keys(d.k0, s.scale0, s.trunc0); fill_scale_keys(d.k1, s.scale1, s.trunc1);
        d.n1_from_server = s.n1_from_server; d.crossed = s.crossed;
    }
    if (self_scale0) { fill_scale_keys(b.ks, *self_scale_keys, *self_trunc_keys); b.self_s0 = (const u64*)self_scale0; b.self_s1 = (const u64*)self_scale1; }
    hipLaunchKernelGGL(scatter_gather_original_kernel, dim3((unsigned)rows, (unsigned)((F + 255) / 256)), dim3(256), 0, ctx->stream, (u64*)outA, (u64*)outB,
                       (const u64*)selfA, (const u64*)selfB, (int)rows, (int)F, rowptr, ent_src, ent_pair, ent_q, b);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// device address of this translation unit's copy of the epoch salt (cognn_spec.h), for cognn_set_epoch_salt
void* cg_salt_symbol_kernels_gather() {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(cognn_epoch_salt_dev)) == hipSuccess ? p : nullptr;
}
