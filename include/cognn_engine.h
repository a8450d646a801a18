/* cognn_engine.h — C ABI of the C++ GAS engine inside libcognn_hip.so.
 *
 * The engine is the MI355X-native counterpart of SSEdgeCentricAlgoKernel::operator()
 * (include/ss_vertex_centric_algo_kernel.h:168-277) driving the optimize-gcn callbacks
 * (algo_kernels/vertex_centric/optimize-gcn/gcn.h): preprocessing -> share distribution ->
 * per-iteration PreScatter / Scatter+PreMerge+Gather (fused CSR) / Apply.  One engine instance
 * runs every party hosted on this process' GPU ("rank"); parties are mapped to ranks in
 * contiguous blocks (party P lives on rank P / (k/world)), so with world == 1 all k parties are
 * co-located and every exchange is an in-device hand-off.  With world > 1 the engine calls the
 * host-supplied exchange function for the share-exchange / Beaver-reveal rounds that the reference
 * carries over TCP (comm_sync.h:245-277, TaskComm); the Python host implements it with
 * torch.distributed (RCCL) p2p groups.
 */
#ifndef COGNN_ENGINE_H_
#define COGNN_ENGINE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cognn_engine cognn_engine;

enum { COGNN_VARIANT_OPTIMIZE_GCN = 0, COGNN_VARIANT_OPTIMIZE_GCN_INFERENCE = 1,
       /* algo_kernels/vertex_centric/original-gcn/gcn.h (bin: gcn-original, BASELINE config 1): aggregate-then-transform, 4 GAS
        * iterations per epoch, messages of width {input_dim, hidden_dim, -, hidden_dim}, every message scaled on its edge by the two
        * degree normalisers (ScatterComp :211-251), forward product after the aggregation.  One process (every party's two
        * share-holders hosted together: one fused launch per destination party) or, in the party placement, one rank per block of
        * parties (the roles of every Scatter instance exchange their openings); the offline call is a no-op (product shares are dealt when used). */
       COGNN_VARIANT_ORIGINAL_GCN = 2 };

/* GNNParam (include/task/task.h:78-170) + harness flags (include/harness.h:123) */
typedef struct {
    int32_t num_parties;     /* -t / -g */
    int32_t rank, world;     /* this process and the number of processes (GPUs) */
    int32_t variant;         /* gcn-optimize or gcn-inference-optimize */
    int32_t num_layers, num_labels, input_dim, hidden_dim;
    double learning_rate, train_ratio, val_ratio, test_ratio;
    uint64_t seed;           /* dealer / sharing seed (stands in for the -s setting string) */
    int32_t device;
    void* stream;            /* hipStream_t (NULL = default stream) */
    int32_t undirected;      /* -u */
    int32_t verbose;         /* record per-phase HIP-event timers of every GAS iteration (cognn_engine_get_phase_seconds):
                                the source of the reference's "::<tag> took X seconds" lines */
    int32_t placement;       /* world > 1: which rank holds which share.  COGNN_PLACE_PARTY (0): a rank is a set of parties - it holds
                                their own shares and the co-shares THEY hold of the previous party's vertices, as a party's machine does
                                in the reference (engine.h:157-201); every two-party opening of an owner / co-party pair on different
                                ranks crosses the link.  COGNN_PLACE_VERTEX_SET (1): a rank holds BOTH shares of its parties' vertex
                                sets - the single-GPU co-located mode extended to several GPUs of one trusted node: every two-party
                                step stays in registers (pair chains) and only Gather traffic (replicas of the share-table segments)
                                and the weight average cross the links.  Same results. */
} cognn_engine_config;
enum { COGNN_PLACE_PARTY = 0, COGNN_PLACE_VERTEX_SET = 1 };

/* one logical message of an exchange round; buffers are device pointers on this rank */
typedef struct {
    int32_t peer;            /* rank */
    int32_t is_send;
    void* ptr;
    int64_t bytes;
} cognn_xfer;
/* must perform all transfers of the list as one p2p group, ordered after previously enqueued work of
 * the engine's stream and complete (stream-ordered) before returning control for later kernels */
typedef int (*cognn_exchange_fn)(void* user, const cognn_xfer* xfers, int32_t n);
/* Asynchronous form: with a wait function registered, cognn_exchange_fn only ENQUEUES the round (ordered after the work
 * already enqueued on the engine's stream) and returns; cognn_exchange_wait_fn makes later work of the engine's stream wait
 * for every enqueued round.  The engine then runs the kernels of the sides whose peer is on the same rank between the two
 * calls, so their compute overlaps the messages of the sides whose peer is remote.  Up to two rounds may be enqueued before a
 * wait (message passing: the co-share replication and the partial-sum exchange travel while the local part of the aggregate
 * runs); the wait covers every enqueued round. */
typedef int (*cognn_exchange_wait_fn)(void* user);
/* Optional third callback: makes later work of the engine's stream wait for the rounds enqueued so far UP TO AND INCLUDING
 * round `round` (rounds are numbered from 0 in the order cognn_exchange_fn was called since the callbacks were registered;
 * a transport completes them in that order), while later rounds stay in flight.  The chunked open -> exchange -> close
 * pipeline (COGNN_OPT_EXCHANGE_CHUNKS) uses it to close chunk c while the messages of chunks c+1.. still travel; without it
 * the engine falls back to cognn_exchange_wait_fn, which waits for everything enqueued. */
typedef int (*cognn_exchange_wait_round_fn)(void* user, int64_t round);

const char* cognn_engine_last_error(void);
int cognn_engine_create(const cognn_engine_config* cfg, int64_t num_vertices, int64_t num_edges,
                        const int64_t* src, const int64_t* dst, const int32_t* part, cognn_engine** out);
int cognn_engine_destroy(cognn_engine* e);
int cognn_engine_set_exchange(cognn_engine* e, cognn_exchange_fn fn, void* user);
int cognn_engine_set_exchange_async(cognn_engine* e, cognn_exchange_fn begin_fn, cognn_exchange_wait_fn wait_fn, void* user);
int cognn_engine_set_exchange_async2(cognn_engine* e, cognn_exchange_fn begin_fn, cognn_exchange_wait_fn wait_fn,
                                     cognn_exchange_wait_round_fn wait_round_fn, void* user);
/* rows (local vertices) of a party and their vids in row order (localVertexPos, ss_...h:474) */
int cognn_engine_party_rows(cognn_engine* e, int32_t party, int64_t* rows);
int cognn_engine_party_vids(cognn_engine* e, int32_t party, int64_t* vids);
int cognn_engine_party_degrees(cognn_engine* e, int32_t party, int64_t* true_in_deg, int64_t* inflated_in_deg, uint8_t* is_border);
/* raw features / labels of a hosted party in row order (harness.cpp:21-48 loadVertexData) */
int cognn_engine_set_party_data(cognn_engine* e, int32_t party, const double* features, const int32_t* labels);
/* optional: replace the srand(42) Glorot initialisation (gcn.h:838-852); w0 [in x hid], w1 [hid x lab] */
int cognn_engine_set_weights(cognn_engine* e, const double* w0, const double* w1);
/* onAlgoKernelStart + share distribution (gcn.h:854-887, ss_...h:205-232) */
int cognn_engine_start(cognn_engine* e);
/* dealer ("offline") phase for iterations [begin,end): Beaver-triple product shares of every GEMM */
int cognn_engine_offline(cognn_engine* e, int64_t iter_begin, int64_t iter_end);
/* A dealt product share is released as soon as the product that consumes it has run, so device memory stays bounded however
 * many iterations a run has (the reference default is -m 1000): deal one epoch ahead.  Replaying the same iterations
 * (bench.py's repeated inference pass) keeps them instead: cognn_engine_set_option(e, COGNN_OPT_RETAIN_OFFLINE, 1). */
/* COGNN_OPT_PAIR_FUSION (default 1): when both share-holders of a vertex set are hosted by this process, their two-party
 * steps between two linear ops run as one pair chain (cognn_pair_chain_u64: exchange in registers) instead of per-side
 * open -> HBM -> close passes.  0 forces the per-side kernels everywhere (the ones a one-party-per-GPU run uses); the shares
 * are bit-identical either way.  Change it only between iterations whose first GAS iteration is a multiple of the epoch.
 * COGNN_OPT_FORWARD_ONLY (default 0): a promise that only forward iterations will run (gcn-inference-optimize with -m 2,
 * tools/tmp_run_cluster.py:396-415).  Co-located pairs then skip the stores that only the backward pass reads - the hidden
 * activation h_t and the public ReLU sign mask - so cognn_engine_get_shares after a hidden-layer iteration is unspecified for
 * them; the prediction layer's shares and metrics are unaffected.  A backward iteration is refused while it is set.
 * COGNN_OPT_PUBLIC_OPENINGS (default 1): share-holders that are NOT run as a pair chain (the peer is on another rank, or pair
 * fusion is off) close a truncation from both opened values and derive the opening of the op that consumes the result
 * themselves (cognn_trunc_close_pub_u64), so the exchange round that carried that opening - and the co-party's reveal of z
 * before the softmax - disappears: 3 of the 17 rounds and 15 % of the bytes of an inference pass.  0: every opening is
 * exchanged as two shares.  Shares and metrics are bit-identical either way.
 * COGNN_OPT_DEALER_STREAMS (default 0; needs COGNN_OPT_RETAIN_OFFLINE): the dealt form of the online phase.  By default
 * every dealer value is regenerated in registers from the counter PRNG (0 HBM bytes).  The reference's online phase instead
 * consumes correlations its offline phase wrote to memory (README.md:215-216, ss_...h:536-610).  With this option the masks of
 * the co-located pairs' chains (cognn_pair_chain_deal_u64: what each party receives for the truncations, row scales, ReLUs and
 * openings) and the A masks of the grouped products are materialised in HBM the first time an iteration runs and READ from
 * there afterwards: +8 bytes per dealt value, shares bit-identical.  A measurement mode (bench.py reports it beside the
 * headline): the per-side kernels of multi-rank runs and the weight-sized operands keep regenerating theirs.
 * Value 2 = the corrections-only form: each party regenerates whatever it derives from its own seed (a_p, b_p, c_0, C_0, r_0, r'_0,
 * the masks of the next opening, the products' A masks) and reads only what a PRG-compressed dealer must SEND it - party 1's
 * correction shares c_1 (element-wise triples), r_1 and r'_1 (truncations), and the ReLU's published g: 7 of the 22 slots.
 * COGNN_OPT_GRAPH_EPOCHS (default 0; single process): cognn_engine_run calls that cover whole epochs (first iteration a multiple
 * of the epoch length) run each such epoch as ONE recorded launch sequence (hipGraph): the first epoch eagerly, the second while
 * it is recorded, every later one as a replay under its own epoch salt (cognn_set_epoch_salt) - dataset-sized graphs spend
 * their epoch in launch overhead.  The engine moves to a private stream, deals product shares inside the recording (the
 * offline call becomes a no-op unless COGNN_OPT_RETAIN_OFFLINE keeps one epoch's products for replays of that epoch) and
 * renews the feature operand's Beaver mask and opening every epoch (the eager form deals that mask once).  Shares, weights and
 * metrics are bit-identical to an eager run that does the same (set the option and call one GAS iteration at a time); against the
 * deal-once form they differ by single LSBs - the carry the 48-bit truncation opening drops depends on how a product is split into
 * shares.
 * COGNN_OPT_EXCHANGE_CHUNKS (default 1; 1..8; matters only for share-holders whose peer is on another rank): the element-wise
 * open -> exchange -> close steps of those sides (product truncation, row scale, ReLU) run in that many row chunks
 * (cognn_ctx_set_chunk): chunk c's messages are enqueued as their own round as soon as chunk c is opened and travel while
 * chunk c+1 is opened and chunk c-1 closed, so those kernels hide behind the link instead of waiting for it.  More, smaller
 * rounds (C times as many for the chunked steps); shares bit-identical for every value. 
 * COGNN_OPT_PACKED_OPENINGS (default 0; matters only between ranks): the opened shares of every truncation and of the ReLU's masked
 * product cross the link as 6 bytes per element instead of 8 - both parties form the opened value from the TOP 48 BITS of the two
 * shares (cognn_spec.h, cognn_open_hi48; the ReLU's multiplier is at least 2^17 so that the sign survives), so the low 16 bits never
 * matter: the open kernels' outboxes are packed into wire buffers (cognn_pack48_u64), those travel, and the inboxes are restored
 * after the round's wait.  Shares bit-identical with and without.  240 of the 471 units a crossing pair exchanges per inference pass
 * are such shares: 494 -> 431 MB per pair and direction on config5. */
enum { COGNN_OPT_RETAIN_OFFLINE = 1, COGNN_OPT_PAIR_FUSION = 2, COGNN_OPT_FORWARD_ONLY = 3, COGNN_OPT_PUBLIC_OPENINGS = 4,
       COGNN_OPT_DEALER_STREAMS = 5, COGNN_OPT_GRAPH_EPOCHS = 6, COGNN_OPT_EXCHANGE_CHUNKS = 7, COGNN_OPT_PACKED_OPENINGS = 8 };
int cognn_engine_set_option(cognn_engine* e, int32_t option, int64_t value);
/* Offline-phase cache on disk, the counterpart of the reference's preprocess/<setting>/ directory reused with `-n 1`
 * (include/harness.h:140-146, README.md:215-216): save writes every dealt product share currently held on this rank to
 * <dir>/c1_r<rank>_o<owner>_i<iter>_op<op>.bin.  The header records magic, seed, M, N, K, transA and a fingerprint of
 * the run (parties, ranks, dimensions, graph size, rows of the owner); load reads the files whose header matches the
 * product of that (owner, iteration) exactly and returns their number in *loaded - anything else (stale cache of another
 * dataset / partition / shape) is ignored and dealt on demand. */
/* hands the product shares dealt for iterations [begin,end) that no iteration has consumed back to the engine's buffer pool (a run that
 * was cut short, a measurement of the dealer phase alone); discarded (may be NULL): how many */
int cognn_engine_offline_discard(cognn_engine* e, int64_t iter_begin, int64_t iter_end, int64_t* discarded);
int cognn_engine_offline_save(cognn_engine* e, const char* dir);
int cognn_engine_offline_load(cognn_engine* e, const char* dir, int64_t iter_begin, int64_t iter_end, int64_t* loaded);
/* GAS iterations [begin,end) (ss_...h:239-248); asynchronous on the engine's stream */
int cognn_engine_run(cognn_engine* e, int64_t iter_begin, int64_t iter_end);
/* waits until everything enqueued so far has finished on the device (what the reference's print_duration sites measure) */
int cognn_engine_sync(cognn_engine* e);
/* Per-phase device time of the LAST iteration run with cfg.verbose != 0, in seconds (HIP events on the engine's stream):
 * out[0] PreScatterComp (gcn.h:198-255), out[1] the fused message passing = Scatter_preparation + Scatter_computation +
 * premerging + premerged_extraction + Gather_preparation + the masked additions of GatherComp (ss_...h:748-856),
 * out[2] the post-gather scale of GatherComp (gcn.h:470-483), out[3] ApplyComp (gcn.h:515-811),
 * out[4] weight averaging (gcn.h:747-802), out[5] number of exchange rounds of that iteration. */
int cognn_engine_get_phase_seconds(cognn_engine* e, double* out6);
/* live device allocations of the engine (count, bytes): stays constant over iterations */
int cognn_engine_get_memory(cognn_engine* e, int64_t* allocations, int64_t* bytes);
/* current vertex tensor share of `owner` held by side 0 (owner) / 1 (co-party); host_out may be NULL to query shape */
int cognn_engine_get_shares(cognn_engine* e, int32_t owner, int32_t side, uint64_t* host_out, int64_t* rows, int64_t* cols);
int cognn_engine_get_weight(cognn_engine* e, int32_t owner, int32_t side, int32_t layer, uint64_t* host_out);
/* metrics of the last prediction layer of a hosted party: out[0..4] = accuracies (full, train, border-train,
 * test, border-test), out[5] = cross-entropy loss, out[6] = #vertices, out[7] = #border (gcn.h:620-632) */
int cognn_engine_get_metrics(cognn_engine* e, int32_t party, double* out8);
/* kernel timing (HIP events on the engine's stream). kind: 0 gather-aggregate launches of the hidden-wide message-passing rounds,
 * 4 those of the label-wide rounds (a different kernel when the prediction layer rides in the launch), 1 gather-partials launches, 2 the
 * Beaver product phase of a GAS iteration (all hosted sides' products; they overlap each other on two launch lanes, so the unit
 * timed is the phase, `launches` = number of phases; the launches that are products and nothing else), 3 the product launches that
 * carry the co-located pairs' truncation chain as their epilogue (cognn_gemm_job::epilogue; algo = the product's operations) */
int cognn_engine_enable_timing(cognn_engine* e, int32_t on);
int cognn_engine_get_timing(cognn_engine* e, int32_t kind, int64_t* launches, double* total_ms, double* algo_bytes_or_ops);
/* static workload numbers: per message-passing round at width F=1 (multiply by F):
 * out[0] edge-rows gathered by aggregate launch, out[1] rows written by it, out[2] edge-rows of the partial launch,
 * out[3] rows written by it, out[4] total directed edges, out[5] table rows */
int cognn_engine_get_workload(cognn_engine* e, int64_t* out6);

#ifdef __cplusplus
}
#endif
#endif /* COGNN_ENGINE_H_ */
