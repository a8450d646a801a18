/* cognn_hip.h — C ABI of the MI355X-native secret-shared GCN engine (libcognn_hip.so).
 *
 * This is the drop-in boundary for CoGNN's hot path.  The reference has no
 * FFI: its GAS callbacks (algo_kernels/vertex_centric/optimize-gcn/gcn.h) call free functions
 * of external libraries (sci::*, *_oblivious_mapper_online, prefix_network_aggregate,
 * CryptoUtil::*) on nested std::vector<uint64_t>.  Every entry point below names the reference
 * call site(s) it replaces.  Conventions:
 *   - all tensors are flat row-major uint64 additive shares mod 2^64 in DEVICE memory
 *     (ShareVecVec, include/task/task.h:237-272, flattened), caller-owned;
 *   - every call is asynchronous on the context's HIP stream and returns 0 on success,
 *     non-zero on error (cognn_last_error() has the text); nothing calls exit() — the
 *     reference's printf+exit(-1) sites (ss_...h:386,794,869,1116) become error returns;
 *   - `p` is the share index of the calling side: 0 = owner/client (sci::ALICE),
 *     1 = co-party/server (sci::BOB)  (ss_...h:743,993; gcn.h:532-533);
 *   - dealer randomness is addressed by (seed, owner, iter, op) -> cognn_opkeys, see
 *     cognn_amd/csrc/cognn_spec.h; "open" steps produce the value a party sends to its peer,
 *     "close" steps consume both parties' opened values.
 */
#ifndef COGNN_HIP_H_
#define COGNN_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COGNN_ABI_VERSION 3
#define COGNN_NUM_SLOTS 11

typedef struct cognn_ctx cognn_ctx;
typedef struct { uint64_t k[COGNN_NUM_SLOTS]; } cognn_keys;   /* == cognn_opkeys */

/* ---- context / memory ------------------------------------------------------------------ */
int cognn_abi_version(void);
const char* cognn_last_error(void);
/* stream: a hipStream_t owned by the caller (e.g. torch's current stream); NULL = the default stream */
int cognn_ctx_create(int device, void* stream, cognn_ctx** out);
/* same, on a private non-blocking stream owned by the context */
int cognn_ctx_create_private(int device, cognn_ctx** out);
int cognn_ctx_destroy(cognn_ctx* ctx);
int cognn_ctx_sync(cognn_ctx* ctx);
/* Launch batching: between begin and end, consecutive element-wise calls of the SAME kind (trunc / rowscale / relu /
 * mask / add ...) are queued and issued as one launch of up to 16 tensors.  The caller guarantees that the queued calls are
 * independent of each other (the sides of one protocol phase); any other entry point, a different kind of call, or end()
 * launches what is queued first, so stream order is otherwise preserved.  Nestable. */
int cognn_batch_begin(cognn_ctx*);
int cognn_batch_end(cognn_ctx*);
/* Launch lanes: independent launch sequences (the per-side products of one protocol phase) issued round-robin on auxiliary
 * streams, so that one sequence's start-up (its operand preparation, its first workgroups) runs in the drain of another's.
 * lane_begin forks `lanes` (1..4) auxiliary streams after everything enqueued on the context's stream so far; lane_select
 * directs the following calls to lane l; lane_end makes the context's stream wait for every lane and returns to it.  The
 * caller guarantees that work on different lanes touches disjoint buffers. */
int cognn_lane_begin(cognn_ctx*, int32_t lanes);
int cognn_lane_select(cognn_ctx*, int32_t lane);
int cognn_lane_end(cognn_ctx*);
/* Chunk window.  While one is set (C >= 2), the element-wise entry points - codec, mask_open, add / sub / sum / fanout, trunc_*,
 * rowscale_*, relu_*, mask_select - process only elements [lo, hi) = cognn_chunk_range(n, c, C) of their flat element range n
 * (pointers stay those of the whole tensors, dealer streams are addressed by the absolute element index: the C windows together
 * do exactly what the unwindowed call does) and every other entry point fails.  The engine uses it to run the open -> exchange ->
 * close steps of share-holders on different GPUs in row chunks, chunk c's messages in flight behind chunk c+1's kernels.
 * c = 0, C <= 1 clears the window.  cognn_rowscale_open_u64 applies the window to E (rows x F) and to G (rows) separately. */
/* lo = floor(n c / C) rounded down to even (the kernels move 16-byte element pairs), hi = the next chunk's lo, n for the last */
void cognn_chunk_range(int64_t n, int32_t c, int32_t C, int64_t* lo, int64_t* hi);
int cognn_ctx_set_chunk(cognn_ctx*, int32_t c, int32_t C);
int cognn_malloc(cognn_ctx* ctx, void** ptr, size_t bytes);
int cognn_free(cognn_ctx* ctx, void* ptr);
int cognn_memcpy_h2d(cognn_ctx* ctx, void* dst, const void* src, size_t bytes);
int cognn_memcpy_d2h(cognn_ctx* ctx, void* dst, const void* src, size_t bytes);
int cognn_memcpy_d2d(cognn_ctx* ctx, void* dst, const void* src, size_t bytes);
int cognn_memset0(cognn_ctx* ctx, void* dst, size_t bytes);
/* The epoch salt (cognn_amd/csrc/cognn_spec.h): every stream evaluated on the device is addressed by key + salt.  0 unless set;
 * the engine sets it to epoch * 0x9E3779B97F4A7C15 around the iterations of an epoch (and back to 0), so that the kernel
 * arguments of an epoch do not depend on the epoch number.  Stream-ordered on the context's stream. */
int cognn_set_epoch_salt(cognn_ctx*, uint64_t salt);
/* Recording and replaying a launch sequence (hipGraph).  capture_begin / capture_end bracket calls of this ABI on the context's
 * stream - which must be a private one: cognn_ctx_create_private, or cognn_ctx_use_private_stream on a context created on the
 * caller's stream - that are recorded instead of executed (no allocation, no synchronising call in between); the handle is
 * replayed with cognn_graph_launch any number of times. */
int cognn_ctx_use_private_stream(cognn_ctx*);
int cognn_graph_capture_begin(cognn_ctx*);
int cognn_graph_capture_end(cognn_ctx*, void** exec);
int cognn_graph_launch(cognn_ctx*, void* exec);
int cognn_graph_destroy(cognn_ctx*, void* exec);
/* derive the slot keys of one dealer op instance (host-side helper) */
void cognn_make_keys(uint64_t seed, uint64_t owner, uint64_t iter, uint64_t op, cognn_keys* out);

/* ---- codec / sharing: CryptoUtil::intoShares, encodeDoubleAsFixedPoint (gcn.h:64-99,220) -- */
/* fx[i] = llround(in[i] * rowscale[i / cols] * 2^16); rowscale may be NULL */
int cognn_fx_encode_f64(cognn_ctx*, const double* in, const double* rowscale, uint64_t* fx, int64_t rows, int64_t cols);
/* s1[i] = prng(key, i); s0[i] = fx[i] - s1[i]  (either output may be NULL) */
int cognn_share_split_u64(cognn_ctx*, const uint64_t* fx, uint64_t key, uint64_t* s0, uint64_t* s1, int64_t n);
int cognn_prng_fill_u64(cognn_ctx*, uint64_t* out, uint64_t key, int64_t n);

/* ---- Gather: OEP + ScatterComp + prefix_network_aggregate + OEP + CondVectorAddition fused
 *      (ss_...h:752-763,815-827,847-854,866-880; gcn.h:257-342,454-463) -------------------- */
/* out[r,:] = (base ? base[r,:] : 0) + sum_{e in [rowptr[r],rowptr[r+1])} table[col[e],:]
 * CSR values are implicitly 1; rows with no entries are the reference's masked (dummy) rows. */
int cognn_gather_csr_u64(cognn_ctx*, uint64_t* out, const uint64_t* base, const uint64_t* table,
                         const uint32_t* rowptr, const uint32_t* col, int64_t n_rows, int64_t F);
/* Same gather, but rows inside one of the `nseg` row segments are written as the Beaver opening of the row scale that
 * follows (twoPartyGCNVectorScale after GatherComp, gcn.h:476): out[r,j] = V[r,j] - prng(seg_key[s], (r-seg_begin[s])*F + j)
 * where V is the gathered value; rows outside every segment are written as V.  nseg <= 32. */
int cognn_gather_csr_open_u64(cognn_ctx*, uint64_t* out, const uint64_t* base, const uint64_t* table,
                              const uint32_t* rowptr, const uint32_t* col, int64_t n_rows, int64_t F,
                              int32_t nseg, const int64_t* seg_begin, const int64_t* seg_end, const uint64_t* seg_key);
/* v[row_index[q],:] += partial[q,:]  (row_index entries distinct): receive side of the mirror
 * vertex update exchange (ss_...h:847-854 with allow-missing, then gcn.h:456). */
int cognn_scatter_add_rows_u64(cognn_ctx*, uint64_t* v, const uint64_t* partial, const uint32_t* row_index,
                               int64_t n_partial, int64_t F);

/* ---- ring GEMM on shares: the local part of sci::twoPartyGCNMatMul (gcn.h:233,665,671,710) - */
/* C[MxN] = (accumulate ? C : 0) + op(A)[MxK] . B[KxN]  mod 2^64.  transA != 0: A is stored [KxM].
 * In the Beaver entry points transA selects how the A mask streams are indexed: 1 = by the logical element (m, k),
 * 2 = by the storage element (k, m) - the mask was dealt for the operand's untransposed use and its opening is reused
 * (the input features serve the layer-0 forward product X.W and the layer-0 weight gradient X^T.g, gcn.h:233,710). */
int cognn_ring_gemm_u64(cognn_ctx*, uint64_t* C, const uint64_t* A, const uint64_t* B,
                        int64_t M, int64_t N, int64_t K, int transA, int accumulate);
/* same with A = A1 + A2 formed on the fly (A2 may be NULL) */
int cognn_ring_gemm2_u64(cognn_ctx*, uint64_t* C, const uint64_t* A1, const uint64_t* A2, const uint64_t* B,
                         int64_t M, int64_t N, int64_t K, int transA, int accumulate);
/* Beaver reveal share: E_p = X_p - prng(key, logical idx). transposed: X is stored [cols x rows]
 * while the logical (masked) matrix is its transpose [rows x cols]. */
#define COGNN_MASK_OPEN_LIMB 16   /* OR into `transposed`: the stream is the A mask of a Beaver product - limb form: mask value = signed-digit
                                    * reading of the PRNG word, a = w - (((w >> 7) & 0x0101..01) << 8) (cognn_spec.h), what the product
                                    * kernels generate for their A operand; plain prng(key, idx) otherwise (B masks, element-wise masks) */
int cognn_mask_open_u64(cognn_ctx*, uint64_t* E, const uint64_t* X, uint64_t key, int64_t rows, int64_t cols, int transposed);
/* The 6-byte wire form of an opened truncation share (or of the ReLU's opened product): only its top 48 bits enter the close
 * (cognn_spec.h, cognn_open_hi48).  packed = n 32-bit words (bits 16..47) followed by n 16-bit words (bits 48..63); unpack restores
 * the share with its low 16 bits zero, which closes to exactly the same result.  packed must be 4-byte aligned.  Not chunk-windowed:
 * the caller passes the range. */
int cognn_pack48_u64(cognn_ctx*, void* packed, const uint64_t* src, int64_t n);
int cognn_unpack48_u64(cognn_ctx*, uint64_t* dst, const void* packed, int64_t n);
/* out[i] = the A mask VALUE of a product (limb form) for element i: what a dealt A mask / a mask image holds */
int cognn_gemm_mask_fill_u64(cognn_ctx*, uint64_t* out, uint64_t key, int64_t n);
int cognn_add_u64(cognn_ctx*, uint64_t* out, const uint64_t* a, const uint64_t* b, int64_t n);
int cognn_sub_u64(cognn_ctx*, uint64_t* out, const uint64_t* a, const uint64_t* b, int64_t n);
/* out = in[0] + ... + in[count-1]  /  out[0..count) = in; count <= 16 (weight averaging, gcn.h:753-778: the weight shares of all
 * hosted sides are summed per holder and the average is handed back to every side: one launch each instead of one per side) */
int cognn_sum_u64(cognn_ctx*, uint64_t* out, const uint64_t* const* in, int32_t count, int64_t n);
int cognn_fanout_u64(cognn_ctx*, uint64_t* const* out, int32_t count, const uint64_t* in, int64_t n);
/* dealer (offline): C1 = (A0+A1).(B0+B1) - C0 with all five streams evaluated from keys */
int cognn_dealer_gemm_c1_u64(cognn_ctx*, uint64_t* C1, const cognn_keys* keys, int64_t M, int64_t N, int64_t K, int transA,
                             uint64_t* scratchA /*MxK*/, uint64_t* scratchB /*KxN*/);
/* ... of several triples that share (N, K) in ONE launch of the grouped MFMA kernel (K ranges split over workgroups when there are few
 * row tiles or K is long: the dataset-shaped layer-0 products): every operand is generated in registers - the limb bytes of the
 * two parties' A masks are the two halves of the A fragment, B_0 + B_1 fills both segments of the B fragments, the epilogue subtracts
 * the C_0 stream - so nothing is read or materialised.  keys: the triple's (A0, A1, B0, B1, C0).  transA = 0 only. */
typedef struct {
    uint64_t* C1;        /* [M x N] out */
    cognn_keys keys;
    int64_t M;
} cognn_dealer_job;
int cognn_dealer_gemm_c1_groupable(int64_t N, int64_t K);
/* ... and of triples whose left operand is used transposed (transA = 1 or 2 as in cognn_dealer_gemm_c1_u64: the weight-gradient
 * products, K = #vertices): the operand fills of all jobs are one launch, each product then accumulates onto its C_1 = -C_0.
 * scratchA: M x K, scratchB: K x N u64 per job (distinct per job). */
typedef struct {
    uint64_t* C1;
    cognn_keys keys;
    int64_t M, N, K;
    int32_t transA;
    uint64_t* scratchA;
    uint64_t* scratchB;
} cognn_dealer_tn_job;
int cognn_dealer_gemm_c1_tn_group_u64(cognn_ctx*, const cognn_dealer_tn_job* jobs, int32_t count);
int cognn_dealer_gemm_c1_group_u64(cognn_ctx*, const cognn_dealer_job* jobs, int32_t count, int64_t N, int64_t K);
/* Z_p = p*E.F + E.B_p + A_p.F + C_p with E = E0 + E1 (the two parties' opened shares; E1 may be NULL) [MxK]
 * and F [KxN] the opened sum; A_p/B_p/C_0 come from keys, C_1 from `c1` (p==1).  transA: E is stored [KxM]
 * (as produced by cognn_mask_open_u64 with transposed=1).  scratch: MxK + KxN u64. */
int cognn_beaver_gemm_close_u64(cognn_ctx*, uint64_t* Z, const uint64_t* E0, const uint64_t* E1, const uint64_t* F, const uint64_t* c1,
                                const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA, uint64_t* scratch);

/* Fast path of the same product for tall-skinny NN shapes: 1 when (M,N,K) is served by the single-launch fused
 * kernel (E streamed once, A_p generated in registers, B limb planes pre-split). */
int cognn_beaver_gemm_fusable(int64_t M, int64_t N, int64_t K, int transA);
/* Raw product share WITHOUT the dealer's C_p: Z = p*E.F + E.B_p + A_p.F (fusable shapes only).  The consumer adds C_p
 * while opening the truncation (cognn_trunc_open_add_u64), so the GEMM kernel has no loads besides its operands. */
int cognn_beaver_gemm_close_raw_u64(cognn_ctx*, uint64_t* Z, const uint64_t* E0, const uint64_t* E1, const uint64_t* F,
                                    const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, uint64_t* scratch);

/* Both of the above with the opened F given as its two shares (F = F0 + F1, summed on the fly like E0 + E1; F1 may be
 * NULL): saves the pass that would form the sum.  raw != 0: without C_p (c1 unused; fusable shapes only). */
int cognn_beaver_gemm_close2_u64(cognn_ctx*, uint64_t* Z, const uint64_t* E0, const uint64_t* E1, const uint64_t* F0, const uint64_t* F1,
                                 const uint64_t* c1, const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA,
                                 uint64_t* scratch, int raw);

/* The products of ONE protocol phase - every hosted side's PreScatter / Apply product (gcn.h:233,665) - as one grouped launch:
 * each job is one cognn_beaver_gemm_close2_u64 call (NN; raw != 0: without C_p).  When all jobs share (N, K), N <= 64 and
 * there are enough row tiles to fill the chip, a single persistent kernel runs them all: every workgroup serves one job,
 * builds that job's weight-operand limb planes ([B_p + p F | F] in MFMA fragment order) in LDS in its prologue - no
 * separate preparation launch, no plane scratch in HBM - and walks the job's row tiles.  Other shapes run job by job through
 * cognn_beaver_gemm_close2_u64 (same results either way).  count <= 16. */
typedef struct {
    uint64_t* Z;
    const uint64_t* E0; const uint64_t* E1;      /* opened left operand [M x K]: one pre-summed tensor or two shares (E1 may be NULL) */
    const uint64_t* F0; const uint64_t* F1;      /* opened right operand [K x N], likewise */
    const uint64_t* c1;                          /* dealt product share (p == 1, raw == 0) */
    cognn_keys keys;
    int32_t p;
    int64_t M;
    uint64_t* scratch;                           /* M x K + K x N u64: used by the per-job fallback only */
    const void* E_presplit;                      /* optional: E0 + E1 already limb-split in MFMA fragment order (cognn_gemm_presplit_u64);
                                                  * used by the grouped kernel when EVERY job of the call brings one */
    const void* A_presplit;                      /* optional, with E_presplit (every job of the call or none; whole-K form): this party's mask A_p of the
                                                  * operand in the same fragment order (cognn_gemm_presplit_u64 of cognn_prng_fill_u64 with its A key) -
                                                  * for an operand whose mask is dealt ONCE (the constant feature tensor): the K loop then has no
                                                  * producer arithmetic, at the price of 8 more bytes read per operand element */
    const uint64_t* A_dealt;                     /* optional (COGNN_OPT_DEALER_STREAMS): this party's mask A_p [M x K] as dealt (cognn_prng_fill_u64
                                                  * with its A key): read instead of regenerated (grouped kernel only) */
    int64_t K;                                   /* cognn_beaver_gemm_close_group_tn_u64 only: the job's inner dimension (rows of its party) */
    const struct cognn_pair_chain_s* epilogue;   /* optional (co-located pairs; every job of the call or none; p == 1; raw; count <= 8; shapes for which
                                                  * cognn_beaver_gemm_group_takes_epilogue says yes): the truncation (+ row scale) chain of the pair
                                                  * runs on this product's tiles while they are still in registers - epilogue->x[0] = the raw product
                                                  * of the pair's p = 0 side (written by an earlier call), x[1] unused, flags within COGNN_PC_TRUNC_IN |
                                                  * COGNN_PC_SCALE | COGNN_PC_NO_C, out[0], out[1] set, no opening / mask / dealt values - and its
                                                  * outputs are stored INSTEAD of Z: bit-identical to the product followed by cognn_pair_chain_u64,
                                                  * without writing and re-reading this side's product */
    int32_t Z_zeroed;                            /* the caller guarantees that Z[0 .. M x N) is zero on entry (its last reader cleared it:
                                                  * COGNN_PC_CLEAR_INPUT / COGNN_WU_CLEAR_Z): the split-K forms, which add partial tiles
                                                  * into Z, then skip their own zeroing launch; ignored by the whole-K form */
} cognn_gemm_job;
int cognn_beaver_gemm_close_group_u64(cognn_ctx*, const cognn_gemm_job* jobs, int32_t count, int64_t N, int64_t K, int raw);
/* The weight-gradient products of one phase, d = h_t^T . in of every hosted side (gcn.h:671,710), as one launch: logical A [M x K]
 * stored [K x M] (E0 / E1), F0 / F1 [K x N]; M and N are common, K is per job (job.K: the rows of the job's party; job.M is
 * unused).  RAW products (without C_p: the consumer's truncation opening adds it) into Z, which the call zeroes first.
 * storage_order_mask != 0: the A mask streams are indexed by the storage element (k, m) (transA = 2 of the per-job calls).
 * Only shapes with cognn_beaver_gemm_tn_groupable(M, N, K, two_share_operands) != 0 (two_share_operands: any job passes E1 or F1). */
int cognn_beaver_gemm_tn_groupable(int64_t M, int64_t N, int64_t K, int two_share_operands);
int cognn_beaver_gemm_close_group_tn_u64(cognn_ctx*, const cognn_gemm_job* jobs, int32_t count, int64_t M, int64_t N, int storage_order_mask);
/* The same idea as cognn_gemm_presplit_u64 for the left operand of the weight-gradient products when it is constant and its mask is
 * dealt once (the feature tensor of the layer-0 gradient, gcn.h:710): the opening (E0 (+ E1), stored [K x M]) or - E0 == NULL - the
 * mask stream `key` addressed as that call addresses it (storage_order_mask), in the order of the TN kernel's A fragments.  A call
 * of cognn_beaver_gemm_close_group_tn_u64 whose every job brings both images (job.E_presplit from the opening, job.A_presplit from
 * this party's A key) with N > 16 and single-tensor operands reads them instead of the operand and the mask stream; other calls
 * ignore them.  image: _bytes(M, K) bytes, 16-byte aligned. */
int64_t cognn_gemm_presplit_tn_bytes(int64_t M, int64_t K);
int cognn_gemm_presplit_tn_u64(cognn_ctx*, void* image, const uint64_t* E0, const uint64_t* E1, uint64_t key, int storage_order_mask, int64_t M, int64_t K);
/* An opened left operand that many products reuse - the constant input-feature opening of the layer-0 product (gcn.h:233 in
 * every epoch) - limb-split and byte-transposed ONCE into the order the grouped kernel's A fragments have: the same 8 bytes per
 * element (rows padded to 16, K to 32), so a pass reads as many bytes as before and skips the split.  image: _bytes(M, K) bytes. */
/* whether the grouped launch of products [. x K] . [K x N] with `row_tiles` 16-row tiles in all takes cognn_gemm_job::epilogue (the
 * whole-K form: the weight planes of every K step fit the LDS image and there are row tiles enough to fill the chip) */
int cognn_beaver_gemm_group_takes_epilogue(int64_t N, int64_t K, int64_t row_tiles);
/* whether that grouped launch runs in the whole-K form at all (the form that reads cognn_gemm_job::A_presplit) */
int cognn_beaver_gemm_group_is_whole_k(int64_t N, int64_t K, int64_t row_tiles);
int64_t cognn_gemm_presplit_bytes(int64_t M, int64_t K);
int cognn_gemm_presplit_u64(cognn_ctx*, void* image, const uint64_t* E0, const uint64_t* E1, int64_t M, int64_t K);

/* ---- truncation by 2^16 (implicit in every sci:: fixed-point op) ------------------------- */
/* c_p = mul * x_p + r_p (+2^61 if p==0) */
int cognn_trunc_open_u64(cognn_ctx*, uint64_t* c, const uint64_t* x, uint64_t mul, const cognn_keys* keys, int p, int64_t n);
/* c_p = x_p + C_p + r_p (+2^61 if p==0) with C_p = prng(gkeys C0 stream) for p==0 and c1[i] for p==1: truncation
 * opening of a raw Beaver product (cognn_beaver_gemm_close_raw_u64) */
int cognn_trunc_open_add_u64(cognn_ctx*, uint64_t* c, const uint64_t* x, const uint64_t* c1, const cognn_keys* gkeys,
                             const cognn_keys* tkeys, int p, int64_t n);
/* y_p = p==0 ? ((c0+c1)>>16) - 2^45 - rp0 : -rp1.  mode 0: out = y; mode 1: out = out - y
 * (twoPartyGCNApplyGradient, gcn.h:678,730). */
int cognn_trunc_close_u64(cognn_ctx*, uint64_t* out, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                          int p, int mode, int64_t n);
/* mode-0 close that also emits the mask opening of the op that consumes the result: out = y, E = y - prng(key_open, i)
 * (E_p = X_p - A_p of the next Beaver product / row scale / ReLU, gcn.h:233,247,549) - one pass instead of two. */
int cognn_trunc_close_open_u64(cognn_ctx*, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                               int p, uint64_t key_open, int64_t n);
/* The same close when BOTH parties hold both opened values c0, c1: besides its share out = y_p (may be NULL) each party derives
 * the next op's opening ITSELF, E = y_0 + y_1 - a_0 - a_1 = ((c0+c1)>>16) - 2^45 - (r>>16) - a_0 - a_1, with the dealer's published
 * t = (r>>16) + a_0 + a_1 (a_p = prng(key_open_p, i)): identical on both parties and identical to the sum of the two E_p of
 * cognn_trunc_close_open_u64, but no second exchange round carries it.  reveal != 0: E = y_0 + y_1, the
 * result itself (the co-party's reveal of z to the owner before the softmax, gcn.h:603-604). */
int cognn_trunc_close_pub_u64(cognn_ctx*, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const cognn_keys* keys,
                              int p, uint64_t key_open0, uint64_t key_open1, int reveal, int64_t n);
/* The same close as a party fed by a real dealer would run it - no stream key on the party's side: t[i] = (r>>16) + a_0 + a_1
 * (reveal: (r>>16)) is the value the dealer PUBLISHES to both parties and rp_own the party's share r'_p, both as device tensors
 * (+8 B per element each): E = ((c0+c1)>>16) - 2^45 - t, out (may be NULL) = (p == 0 ? ((c0+c1)>>16) - 2^45 : 0) - rp_own.
 * cognn_dealer_trunc_pub_u64 is that dealer: it fills t (and, when given, rp0 / rp1) from the streams the key form derives in
 * registers, so the two forms are bit-identical (tests/test_ops_gpu.py).  The engine uses the key form (no HBM bytes). */
int cognn_trunc_close_pub_dealt_u64(cognn_ctx*, uint64_t* out, uint64_t* E, const uint64_t* c0, const uint64_t* c1, const uint64_t* t,
                                    const uint64_t* rp_own, int p, int64_t n);
int cognn_dealer_trunc_pub_u64(cognn_ctx*, uint64_t* t, uint64_t* rp0, uint64_t* rp1, const cognn_keys* keys, uint64_t key_open0,
                               uint64_t key_open1, int reveal, int64_t n);

/* ---- sci::twoPartyGCNVectorScale (gcn.h:247,476): row scale by an additively shared vector - */
/* E_p = V_p - a_p [rows x F] (skipped when E == NULL: already opened by cognn_gather_csr_open_u64), G_p = s_p - b_p [rows] */
int cognn_rowscale_open_u64(cognn_ctx*, uint64_t* E, uint64_t* G, const uint64_t* V, const uint64_t* s,
                            const cognn_keys* keys, int p, int64_t rows, int64_t F);
/* z_p = p*E*G + E*b_p + a_p*G + c_p, immediately followed by trunc_open with tkeys:
 * c_out = z_p + r_p (+2^61).  E = E0+E1 [rows x F], G = G0+G1 [rows]: the two parties' opened shares, summed
 * on the fly (E1/G1 may be NULL when the sum was formed elsewhere). */
int cognn_rowscale_close_u64(cognn_ctx*, uint64_t* c_out, const uint64_t* E0, const uint64_t* E1, const uint64_t* G0, const uint64_t* G1,
                             const cognn_keys* keys, const cognn_keys* tkeys, int p, int64_t rows, int64_t F);

/* ---- sci::twoPartyGCNRelu / twoPartyGCNBackwardNNWithoutAH (gcn.h:549,705) ---------------- */
/* E_p = z_p - a_p, G_p = t_p - b_p.  G may be NULL: g = t - b does not depend on the inputs, so the dealer can publish it
 * in the offline phase and only E (and later w) travel online; cognn_relu_mul_u64 then takes G0 = G1 = NULL. */
int cognn_relu_open_u64(cognn_ctx*, uint64_t* E, uint64_t* G, const uint64_t* z, const cognn_keys* keys, int p, int64_t n);
/* w_p = Beaver product share of z*t from the opened E = E0+E1, G = G0+G1 (E1/G1 may be NULL; G0 NULL: dealer-published g,
 * regenerated from the dealer streams) */
int cognn_relu_mul_u64(cognn_ctx*, uint64_t* w, const uint64_t* E0, const uint64_t* E1, const uint64_t* G0, const uint64_t* G1,
                       const cognn_keys* keys, int p, int64_t n);
/* h_p = (int64)(w0+w1) > 0 ? z_p : 0; mask (1 byte/element, public) may be NULL */
int cognn_relu_close_u64(cognn_ctx*, uint64_t* h, uint8_t* mask, const uint64_t* z, const uint64_t* w0, const uint64_t* w1, int64_t n);
int cognn_mask_select_u64(cognn_ctx*, uint64_t* out, const uint64_t* in, const uint8_t* mask, int64_t n);
/* relu_close fused with the Beaver opening of the next product: also writes E[i] = h[i] - prng(key_open, i) */
int cognn_relu_close_open_u64(cognn_ctx*, uint64_t* h, uint64_t* E, uint8_t* mask, const uint64_t* z, const uint64_t* w0, const uint64_t* w1,
                              uint64_t key_open, int64_t n);

/* ---- twoPartyGCNForwardNNPredictionWithoutWeight + getPlainShareVecVec (gcn.h:578-604) ---- */
/* owner side (p=0): z=z0+z1 -> integer softmax pfx (Q16) ; p0 = pfx - rho ; d0 = p0 - onehot(label),
 * rows >= train_rows zeroed.  co side (p=1): pass z1=NULL,labels=NULL: p = rho, d = rho (zeroed rows). */
int cognn_softmax_u64(cognn_ctx*, uint64_t* p_out, uint64_t* d_out, uint64_t* pfx_out, const uint64_t* z0, const uint64_t* z1,
                      const int32_t* labels, const cognn_keys* keys, int p, int64_t rows, int64_t L, int64_t train_rows);
/* sci::cross_entropy_loss / accuracy (gcn.h:620-632) on the revealed Q16 probabilities.
 * out[0..5] (int64 counts: correct full/train/border-train/test/border-test, n) + loss (double). */
int cognn_metrics_q16(cognn_ctx*, const uint64_t* pfx, const int32_t* labels, const uint8_t* border,
                      int64_t rows, int64_t L, int64_t train_rows, int64_t val_rows, int64_t* counts6, double* loss);

/* The prediction layer of every side hosted by a process in ONE launch: owner jobs (p = 0) run cognn_softmax_u64 and
 * cognn_metrics_q16 fused (the revealed probabilities stay in registers: no pfx tensor is written or re-read), co-party jobs
 * (p = 1) write their mask share.  d_out as in cognn_softmax_u64; counts6 / loss as in cognn_metrics_q16 (owner jobs only).
 * z1 == NULL in an owner job: z0 is the revealed z itself (cognn_trunc_close_pub_u64 with reveal). */
typedef struct {
    uint64_t* d_out;
    const uint64_t* z0; const uint64_t* z1;
    const int32_t* labels; const uint8_t* border;
    cognn_keys keys;
    int32_t p;
    int64_t rows, train_rows, val_rows;
    int64_t* counts6; double* loss;
} cognn_softmax_job;
int cognn_softmax_jobs_u64(cognn_ctx*, const cognn_softmax_job* jobs, int32_t count, int64_t L);

/* ---- both share-holders on one device: a chain of protocol steps with the exchange in registers ---------------------------
 * When the owner AND the co-party of a vertex set are hosted by the same process (BASELINE configs "co-located on one
 * MI355X, in-device share exchange") every opening of a two-party step would be written to HBM by one side only to be read
 * back by the other.  A pair chain runs the steps between two linear ops (GEMM / Gather) for BOTH sides in one kernel: each
 * thread evaluates side 0's and side 1's local arithmetic exactly as the per-side entry points above do (same dealer
 * streams, same formulas) and hands the opened values over in registers.  Results are bit-identical to the per-side
 * sequence; only HBM passes disappear.  Steps, in this order, selected by `flags`:
 *   COGNN_PC_TRUNC_IN   x_p is a raw Beaver product share: truncation of x_p + C_p (C_0 from gemm_keys, C_1 = c1[]; with
 *                       COGNN_PC_NO_C the product already contains C_p)       = trunc_open[_add] + exchange + trunc_close
 *   COGNN_PC_SCALE      row scale by the shared vector (scale[0], scale[1]) then truncation = rowscale_open + exchange +
 *                       rowscale_close + exchange + trunc_close; with COGNN_PC_INPUT_OPENED x_p already is V_p - a_p (written
 *                       by cognn_gather_csr_open_u64)
 *   COGNN_PC_RELU       masked-sign ReLU = relu_open(G = NULL) + exchange + relu_mul + exchange + relu_close; mask optional
 * Outputs: out[p] (may be NULL) and, when open[p] != NULL, the opening of the op that consumes the result:
 * open[p][i] = out_p[i] - prng(open_key[p], i).  rows * F < 2^32. */
enum { COGNN_PC_TRUNC_IN = 1, COGNN_PC_SCALE = 2, COGNN_PC_RELU = 4, COGNN_PC_INPUT_OPENED = 8, COGNN_PC_NO_C = 16,
       /* open[0] receives (out_0 - a_0) + (out_1 - a_1), the opening as both parties hold it after the exchange; open[1] must be
        * NULL: one tensor written instead of two, and the consuming product streams one operand instead of summing two */
       COGNN_PC_OPEN_SUM = 32,
       /* COGNN_PC_TRUNC_IN chains only: x[0], x[1] are zeroed behind the read - the product buffer is handed back clean to the next
        * split-K product (cognn_gemm_job::Z_zeroed), which saves that product's zeroing launch; worth it for small tensors only
        * (+16 B of writes per element pair) */
       COGNN_PC_CLEAR_INPUT = 128,
       /* with COGNN_PC_TRUNC_IN and mask_in: the selection applies to the truncated product (mask_in[i] ? v_p : 0 on both sides after
        * the truncation) instead of the raw input - g = (p - y) . W^T truncated, then the backward ReLU' (gcn.h:702-708), then the
        * steps that follow, in one chain */
       COGNN_PC_MASK_AFTER_TRUNC = 256,
       /* with `dealt`: read only what a PRG-compressed dealer must send - party 1's correction shares (c_1 of the element-wise triples,
        * r_1 and r'_1 of the truncations) and the ReLU's published g - and regenerate what each party derives from its own seed */
       COGNN_PC_DEALT_MINIMAL = 512,
       /* the opening written by the chain (open / open_key) is the left operand of a Beaver PRODUCT: its masks are in limb form
        * (cognn_spec.h, cognn_gemm_mask) like every A mask the product kernels generate */
       COGNN_PC_OPEN_LIMB = 1024 };
typedef struct cognn_pair_chain_s {
    const uint64_t* x[2];        /* the two sides' input shares [rows x F] */
    const uint64_t* c1;          /* side 1's dealt product share (COGNN_PC_TRUNC_IN without COGNN_PC_NO_C) */
    const uint64_t* scale[2];    /* the two sides' shares of the row scale [rows] (COGNN_PC_SCALE) */
    uint64_t* out[2];
    uint64_t* open[2];
    uint8_t* mask;               /* public ReLU sign, 1 byte per element (COGNN_PC_RELU, optional) */
    uint64_t open_key[2];
    cognn_keys gemm_keys, trunc_in_keys, scale_keys, scale_trunc_keys, relu_keys;
    int64_t rows, F;
    int32_t flags;
    const uint8_t* mask_in;      /* optional: x_p is taken as mask_in[i] ? x_p[i] : 0 - the backward ReLU' (cognn_mask_select_u64 with the public
                                  * sign mask, gcn.h:702-708) folded into the chain that consumes its result; not with COGNN_PC_INPUT_OPENED */
    const uint64_t* dealt;       /* optional (COGNN_OPT_DEALER_STREAMS): the chain's per-element dealer values as the offline phase hands them to
                                  * the two parties, [slot][rows x F] (cognn_pair_chain_deal_u64): read instead of regenerated from the counter
                                  * PRNG - identical results, +8 bytes of HBM traffic per slot and element */
} cognn_pair_chain;
/* The dealt form of one chain (README.md:215-216, ss_...h:536-610: the reference's online phase consumes preprocessed correlations
 * from memory): per element, in this order, the slots the chain's steps use -
 *   COGNN_PC_TRUNC_IN: C_0, r_0, r_1, r'_0, r'_1;  COGNN_PC_SCALE: a_0, a_1, c_0, c_1, r_0, r_1, r'_0, r'_1 (b_p: one value per row,
 *   regenerated);  COGNN_PC_RELU: a_0, a_1, b_0, b_1, c_0, c_1, g;  an opening (open[0] or open[1] set): a_0, a_1.
 * _slots: their number; _deal fills `dealt` (slots x rows x F u64) from the chain's keys. */
int64_t cognn_pair_chain_dealt_slots(int32_t flags, int32_t has_open);
int cognn_pair_chain_deal_u64(cognn_ctx*, const cognn_pair_chain* chain, uint64_t* dealt);
/* independent chains (the co-located pairs of one protocol phase) are issued as shared launches */
int cognn_pair_chain_u64(cognn_ctx*, const cognn_pair_chain* chains, int32_t count);

/* The weight update of ApplyComp's second backward iteration (gcn.h:671-684, 710-736; optimize-gcn-inference/gcn.h:680-681,
 * 732-733) for a co-located pair, as one pass over the (small) weight matrix: with d_p = z_p (+ C_p unless COGNN_PC_NO_C) the
 * weight-gradient product shares,
 *   d = trunc(d; trunc_keys[0]);  d = trunc(d * mul[0]; trunc_keys[1]);  u = trunc(d * mul[1]; trunc_keys[2]);  W -= u;
 *   mul[2] != 0:  W = trunc(W * mul[2]; trunc_keys[3])
 * where trunc(v; k) is cognn_trunc_open[_add]_u64 on both sides + exchange + cognn_trunc_close_u64 on both sides, evaluated for
 * both sides by one thread (same dealer streams, same formulas: bit-identical to that sequence).  Jobs of one call (the pairs
 * of a GAS iteration) share a launch.
 * average != 0 (every party's pair is hosted by the caller; count <= 16, one n): the weight average that follows (gcn.h:753-778) in
 * the same pass - sum_0 = sum over jobs of W[swap ? 1 : 0], sum_1 = of the other share (owner 0 keeps (s0, s1), owners >= 1
 * (s1, s0): COGNN_WU_SWAP); avg_mul != 0: (sum_0, sum_1) = trunc(sum * avg_mul; avg_keys) between parties 0 and 1; every job's
 * W[swap ? 1 : 0] = sum_0 and its other share = sum_1.  Bit-identical to cognn_sum_u64 x 2, the truncation and cognn_fanout_u64 x 2. */
enum { COGNN_WU_SWAP = 64, COGNN_WU_CLEAR_Z = 128 /* z[0], z[1] are zeroed behind the read (see COGNN_PC_CLEAR_INPUT) */ };
typedef struct {
    const uint64_t* z[2];        /* the two sides' product shares [n] */
    const uint64_t* c1;          /* side 1's dealt product share (flags without COGNN_PC_NO_C) */
    uint64_t* W[2];              /* the two sides' weight shares [n], updated in place */
    cognn_keys gemm_keys;        /* C_0 stream of the product */
    cognn_keys trunc_keys[4];
    uint64_t mul[3];             /* gradient scale 1/|train|, learning rate, post scale (0: none) - Q16 */
    int64_t n;
    int32_t flags;               /* COGNN_PC_NO_C, COGNN_WU_SWAP, COGNN_WU_CLEAR_Z */
} cognn_pair_wupdate;
int cognn_pair_weight_update_u64(cognn_ctx*, const cognn_pair_wupdate* jobs, int32_t count, const cognn_keys* avg_keys, uint64_t avg_mul,
                                 int32_t average);

/* Gather whose epilogue IS the pair chain (co-located pairs, single process): for every owner, row r of its owner-side segment
 * (first row a_row0) and row r of its co-party-side segment (b_row0) are aggregated by the same lanes -
 *   V_p[r,:] = table[row_p(r),:] + sum_{e in CSR row row_p(r)} table[col[e],:]        (the self row is the base)
 * - and the chain (COGNN_PC_SCALE, optionally | COGNN_PC_RELU; chain.x / rows-of-x are unused, chain.rows = rows of the segment)
 * runs on (V_0, V_1) in registers: the aggregate is never written, only the chain's outputs / openings are.  Bit-identical to
 * cognn_gather_csr_u64 followed by cognn_pair_chain_u64.  flags = 0 (the last backward Gather has no scale, gcn.h:470): the
 * outputs / openings are those of the aggregate itself.  rowptr / col index rows of `table`; count <= 8.  An odd F (7 or 3 labels) runs with 8-byte lanes.
 * softmax[0], softmax[1] (every pair of the call or none; F = the label count with cognn_gather_pair_chain_takes_softmax(F) != 0;
 * chain.flags without COGNN_PC_RELU, no opening, no dealt values): the prediction layer that consumes this Gather (gcn.h:578-632) runs
 * as the launch's second epilogue - the owner's (p = 0) and the co-party's (p = 1) job as cognn_softmax_jobs_u64 takes them (rows =
 * chain.rows; z0 / z1 are ignored: z is the chain's result, still in registers), and chain.out may be NULL: the logits are neither
 * written nor re-read.  d_out and counts6 bit-identical to cognn_softmax_jobs_u64 on the chain's outputs, loss up to the order of its
 * floating-point atomic additions. */
typedef struct {
    int64_t a_row0, b_row0;
    cognn_pair_chain chain;
    const cognn_softmax_job* softmax[2];
} cognn_gather_pair;
int cognn_gather_pair_chain_takes_softmax(int64_t F);
/* The same with the rows' starting values taken from `base` (same row numbering as `table`) instead of the rows of `table`
 * themselves: base = the sums a first launch (cognn_gather_csr_u64 over the entries that read rows held by this rank) left, rowptr /
 * col = the entries that read rows received from other ranks - a multi-rank run aggregates the local part while the messages
 * travel and lets the launch over the received part carry the epilogue.  base == NULL: cognn_gather_pair_chain_u64. */
int cognn_gather_pair_chain_base_u64(cognn_ctx*, const uint64_t* table, const uint64_t* base, const uint32_t* rowptr, const uint32_t* col,
                                     int64_t F, const cognn_gather_pair* pairs, int32_t count);
int cognn_gather_pair_chain_u64(cognn_ctx*, const uint64_t* table, const uint32_t* rowptr, const uint32_t* col, int64_t F,
                                const cognn_gather_pair* pairs, int32_t count);

/* ---- original-gcn message passing (algo_kernels/vertex_centric/original-gcn/gcn.h:211-405) for ONE destination party of a
 *      single-process run: ScatterComp + UpdatePreMergeComp + GatherComp of every in-edge of its vertices in one launch -----
 * The unoptimised kernel scales every message on its edge: for edge q of the Scatter instance (client P = source party, server =
 * the destination party, or P's co-party for P's local edges) the pair computes
 *     m = trunc(trunc(u * n0[q]) * n1[q]),   u = the source row, both shares (srcA: client's, srcB: server's)
 * with two Beaver row scales (a per element, b per edge; streams of `scale0` / `trunc0` / `scale1` / `trunc1`, addressed by the
 * edge's position q in the instance's edge list: element q * F + j) - n0 = (outDeg_src + 1)^-1/2 is the client's value (share
 * (n0, 0)), n1 = (inDeg_dst + 1)^-1/2 the client's for local edges, the server's for the others (share (0, n1);
 * ss_...h:800,1041-1043).  Then, per destination vertex r (both share-holders' rows by the same thread):
 *     out_A[r] = self_A[r] + sum over local edges m_A + sum over the other edges m_B,   out_B[r] likewise with A / B swapped
 * (which share feeds which side is the crossing of ss_...h:1063-1100), where, in a forward iteration (self_scale != NULL), the self
 * row first goes through the same scale + truncation with the vertex's own normaliser (gcn.h:365-381; streams of self_scale_keys /
 * self_trunc_keys at element r * F + j).  Entries of a destination row: ent_src (row inside the source party), ent_pair (index
 * into pairs[]), ent_q.  Bit-identical to the per-edge sequence of the oracle (oracle/original_gcn.py). npairs <= 16. */
typedef struct {
    const uint64_t* srcA;        /* the source party's tensor, owner-side share [n_src x F] */
    const uint64_t* srcB;        /* ... co-party-side share */
    const uint64_t* n0;          /* [edges of the instance] */
    const uint64_t* n1;
    cognn_keys scale0, trunc0, scale1, trunc1;
    int32_t n1_from_server;      /* 0: n1 is the client's share, 1: the server's */
    int32_t crossed;             /* 0: m_A -> out_A (the source party's local edges); 1: m_A -> out_B, m_B -> out_A */
} cognn_scatter_pair;
int cognn_scatter_gather_original_u64(cognn_ctx*, uint64_t* outA, uint64_t* outB, const uint64_t* selfA, const uint64_t* selfB,
                                      const uint64_t* self_scale0, const uint64_t* self_scale1, const cognn_keys* self_scale_keys,
                                      const cognn_keys* self_trunc_keys, int64_t rows, int64_t F, const uint32_t* rowptr,
                                      const uint32_t* ent_src, const uint32_t* ent_pair, const uint32_t* ent_q,
                                      const cognn_scatter_pair* pairs, int32_t npairs);

/* ---- onPreprocessClient index construction (ss_...h:295-534) + degree accounting (graph.h:607-633, graph_io_util.h:167-177)
 *      on the device, for a run whose parties are all hosted by ONE process ------------------------------------------------
 * All pointers are DEVICE pointers.  In: the directed edge list in file order (both directions are generated when
 * `undirected`), vid -> party (`tid`), vid -> row inside its party (`row_of_vid`, ascending vid = localVertexPos order), and the
 * share-table offsets of every party's owner-side (a_off) and co-party-side (b_off) row segment.
 * Out: the aggregate CSR over the table (rowptr[table_rows + 1], col[2 * #directed edges]; entry order inside a row is
 * unspecified - uint64 addition commutes), and per vertex the in-degree as onAlgoKernelStart sees it (true_in_deg), the
 * degrees after the dummy-source rule of ss_...h:411-418 (in_deg, out_deg), isLocalVertexBorder and the dummy flag.
 * scratch: V + 2 * table_rows + 2 uint32.  Synchronises the stream (it reports malformed edges). */
int cognn_graph_build_colocated(cognn_ctx*, int64_t V, int64_t E, int32_t undirected, const int64_t* src, const int64_t* dst,
                                const int32_t* tid, const uint32_t* row_of_vid, const int64_t* a_off, const int64_t* b_off,
                                int64_t table_rows, uint32_t* rowptr, uint32_t* col, uint32_t* true_in_deg, uint32_t* in_deg,
                                uint32_t* out_deg, uint8_t* is_border, uint8_t* self_dummy, uint32_t* scratch);

/* out[c,r] = in[r,c] for a small [rows x cols] matrix (transpose(), include/task/task.h:243; gcn.h:648) */
int cognn_transpose_u64(cognn_ctx*, uint64_t* out, const uint64_t* in, int64_t rows, int64_t cols);

/* ---- HIP-event timers on the context's stream (print_duration stand-in, ss_...h:188-192) ---- */
int cognn_timer_begin(cognn_ctx*, int kind);
int cognn_timer_end(cognn_ctx*, int kind);
/* synchronises the stream; sums all completed begin/end pairs of `kind` since the last reset */
int cognn_timer_read(cognn_ctx*, int kind, int64_t* launches, double* total_ms);
int cognn_timer_reset(cognn_ctx*);

#ifdef __cplusplus
}
#endif
#endif /* COGNN_HIP_H_ */
