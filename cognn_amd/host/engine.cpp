// Host engine: the MI355X-first counterpart of SSEdgeCentricAlgoKernel (include/
// ss_vertex_centric_algo_kernel.h:168-277, 680-910, 912-1189) running the optimize-gcn callbacks
// (algo_kernels/vertex_centric/optimize-gcn/gcn.h:198-811) for every party hosted on this rank.
//
// Differences from the reference's structure (results on shares are those of oracle/cognn_oracle.py):
//  * no thread-per-peer: one HIP stream, kernels batched over all hosted parties;
//  * a "Side" is one share-holder role for one owner's vertices: p=0 the owner (client, sci::ALICE),
//    p=1 the co-party (owner+1)%k (server, sci::BOB).  Every two-party op is open -> exchange -> close;
//    when both sides live on this rank the exchange is a pointer hand-off;
//  * Scatter/PreMerge/Gather are two CSR launches per round over a rank-wide share table
//    (partials for remote destinations, then the aggregate of every hosted row), see DESIGN.md §4.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/cognn_engine.h"
#include "../csrc/cognn_spec.h"
#include "backend.h"
#include "graph.h"

typedef uint64_t u64;

namespace {

thread_local std::string g_engine_error;

struct EngineError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// T_AGG / T_AGG_LAB: the aggregate launches of the message-passing rounds at the first forward round's width (hidden_dim) and at the
// other width (num_labels) - two different kernels in the fused form (pair chain epilogue / prediction-layer epilogue)
enum { T_AGG = 0, T_PART = 1, T_GEMM = 2, T_PH_PRESCATTER = 3, T_PH_MP = 4, T_PH_GATHER = 5, T_PH_APPLY = 6, T_PH_WAVG = 7, T_GEMM_EPI = 8, T_AGG_LAB = 9 };

struct Side {
    int owner = 0, p = 0, n = 0;
    int peer_rank = 0;
    Side* peer = nullptr;          // non-null when the other share-holder is hosted on this rank
    u64* feat = nullptr;           // [n x in] input-feature share (localVertexSvvBackup / remoteVertexSvvsBackup)
    u64* featE = nullptr;          // E_p = feat_p - A_p of the layer-0 product, opened once (fixed-operand mask reuse)
    u64* featE_peer = nullptr;     // the peer's opening (alias when co-located)
    const u64* featSum = nullptr;  // E = E_0 + E_1 of that product, summed once in start() (one copy per co-located pair)
    const void* featPl = nullptr;  // the same opening limb-split in MFMA fragment order (cognn_gemm_presplit_u64), for the grouped forward product
    const void* featMaskPl = nullptr;   // ... and this side's mask A_p of the feature operand in that order (COGNN_GEMM_MASK_IMAGE; dealt once like the opening)
    const void* featTnPl = nullptr, *featMaskTnPl = nullptr;   // both once more in the order of the weight-gradient kernel's A fragments (training only)
    u64* h1E = nullptr;            // E_p = h_p - A_p of the layer-1 forward product (written by the ReLU close); kept for the epoch:
    u64* h1E_peer = nullptr;       // the layer-1 weight gradient h^T.g reuses mask and opening (alias when co-located)
    const uint8_t* cur_mask = nullptr;   // co-located pairs, between the backward ReLU' and the row scale that consumes it: the tensor is
                                         // cur (.) cur_mask, the selection rides in that row scale's pair chain (apply_cur_mask otherwise)
    u64* W[2] = {nullptr, nullptr};
    u64* h1 = nullptr;             // h_t of layer 1 [n x hid]   (vertexInterData["h_t"], gcn.h:230-231)
    u64* g = nullptr;              // vertexInterData["g"] [n x hid]
    u64* ah[2] = {nullptr, nullptr};   // original-gcn: vertexInterData["ah_t"] of both layers, stored untransposed [n x in], [n x hid] (gcn.h:452)
    uint8_t* relu_mask = nullptr;  // public sign of z[0] (revealed by the masked-sign ReLU)
    u64* cur = nullptr;            // current vertex tensor share [n x curF]
    int curF = 0;
    u64* buf[2] = {nullptr, nullptr};
    u64* ob[3] = {nullptr, nullptr, nullptr};      // outboxes: 0/1 = Beaver openings, 2 = truncation / product opening
    u64* ib_store[3] = {nullptr, nullptr, nullptr};
    u64* ib[3] = {nullptr, nullptr, nullptr};      // peer's outboxes (aliases when the peer is on this rank)
    u64* scratch = nullptr;
    u64* zbuf = nullptr;           // untruncated GEMM output
    // zbuf[z_dirty, z_zero) is known to be zero: a product dirties [0, M x N); the pair kernels that consume a SMALL product clear
    // it behind their read (COGNN_PC_CLEAR_INPUT / COGNN_WU_CLEAR_Z), so the next split-K product skips its zeroing launch
    int64_t z_dirty = 0, z_zero = 0;
    u64* small[3] = {nullptr, nullptr, nullptr};   // [in x hid]-sized temporaries for the weight chain
    u64* svec = nullptr;           // normaliser share [n]
    int32_t* labels = nullptr;
    uint8_t* border = nullptr;
    u64* pfx = nullptr;            // revealed Q16 probabilities (owner only)
    int64_t* counts = nullptr;
    double* loss = nullptr;
    bool has_metrics = false;
    struct C1 { u64* ptr; int64_t elems; };
    std::map<std::pair<int64_t, int>, C1> c1;     // dealt product shares not consumed yet, (iter, op) -> [M x N]
};

}  // namespace

struct cognn_engine {
    cognn_engine_config cfg;
    const cognn_backend* be = nullptr;
    cognn_ctx* ctx = nullptr;
    cognn::PartitionedGraph G;
    int k = 0, world = 1, rank = 0, m = 1;
    std::vector<int> hosted, cohosted;
    std::vector<Side> sides;
    std::vector<void*> allocs;
    int64_t alloc_bytes = 0;
    std::map<int64_t, std::vector<u64*>> c1_pool;   // released product-share buffers by element count, reused by later deals
    bool retain_offline = false;                    // COGNN_OPT_RETAIN_OFFLINE
    int gemm_lanes = getenv("COGNN_GEMM_LANES") ? atoi(getenv("COGNN_GEMM_LANES")) : 2;   // launch lanes of the per-side products (A/B switch: 1 = one stream)
    bool gemm_group = !getenv("COGNN_GEMM_PER_SIDE");       // one grouped launch per phase (A/B switch: the per-side launch sequences)
    // co-located pairs: the product's chain as the epilogue of the p = 1 side's launch (cognn_gemm_job::epilogue).  Opt-in: measured
    // 1.6 % faster on config5 before the mask image (5.40 vs 5.47 ms), no difference with it (5.36-5.40 both ways); COGNN_GEMM_EPILOGUE=1
    bool gemm_epilogue = getenv("COGNN_GEMM_EPILOGUE") != nullptr;
    bool wupdate_fusion = !getenv("COGNN_NO_WUPDATE_FUSION"); // co-located pairs: weight update (+ average) as one pass (A/B switch)
    // the feature operand's mask A_p (dealt once, like its opening) kept in fragment order too: the layer-0 product's K loop then has no
    // producer arithmetic (+ 8 B read per operand element; config5 5.66 -> 5.6 ms, product phases 0.43 -> 0.45 of the i8 peak; more than one
    // column tile only: at hidden_dim <= 16 the bytes cost more than the arithmetic).  A/B switch.
    bool gemm_mask_image = !getenv("COGNN_GEMM_NO_MASK_IMAGE");
    bool gemm_presplit = !getenv("COGNN_GEMM_NO_PRESPLIT"); // the constant feature opening kept in MFMA fragment order (A/B switch)
    bool public_openings = true;                    // COGNN_OPT_PUBLIC_OPENINGS (see pub_open)
    bool h1e_pairs_summed = false;                  // the co-located pairs' h1E holds E_0 + E_1 (written by a pair chain), not E_p
    // a training epoch inside ONE cognn_engine_run call (nobody can read the state between its iterations): the chain that truncates
    // g = (p - y) . W1^T already applies the backward ReLU' and the PreScatter scale of three iterations later and writes that
    // iteration's share table (a second table) - g itself and the later scale pass are never written / run (A/B switch)
    bool backward_fusion = !getenv("COGNN_NO_BACKWARD_FUSION");
    int64_t run_end = 0;                            // end of the running cognn_engine_run call (exclusive)
    int64_t prescaled_it = -1;                      // the iteration whose PreScatter result already sits in table2
    u64* table2 = nullptr;
    bool softmax_fusion = !getenv("COGNN_NO_SOFTMAX_FUSION");   // the prediction layer as the second epilogue of the label-wide Gather (A/B switch)
    bool pair_fusion = true;                        // COGNN_OPT_PAIR_FUSION: co-located share-holders run their two-party steps as pair chains
    bool forward_only = false;                      // COGNN_OPT_FORWARD_ONLY: no backward iteration will follow (inference, -m 2)
    bool graph_epochs = false;                      // COGNN_OPT_GRAPH_EPOCHS: whole epochs are recorded once (hipGraph) and replayed
    bool graph_warm = false, graph_unsupported = false;
    void* graph_exec = nullptr;
    int64_t graph_epoch = -1;                       // the epoch the recorded graph was captured in (retained products are tied to it)
    u64 salt_now = 0;                               // the epoch salt of the iteration being issued (added to the keys on the host, or - recorded epochs - on the device)
    u64 salt_on_device = 0;                         // what cognn_set_epoch_salt last set (recorded epochs only)
    bool dealer_group = getenv("COGNN_NO_DEALER_GROUP") == nullptr;   // offline phase: the product shares of one shape in one grouped MFMA launch
    int dealer_streams = 0;                         // COGNN_OPT_DEALER_STREAMS: 1 = dealt values of the pair chains / grouped products read from HBM; 2 = only the dealer's corrections
    std::map<std::tuple<int, int64_t, int>, u64*> dealt;   // (owner, iteration, place) -> slab, filled at first use, kept (retain_offline)
    int64_t dealt_bytes = 0;
    double phase_s[6] = {0, 0, 0, 0, 0, 0};         // cognn_engine_get_phase_seconds
    int64_t rounds = 0;                             // exchange rounds started (all iterations)
    cognn_exchange_fn xfn = nullptr;
    cognn_exchange_wait_fn xwait = nullptr;   // set: xfn only enqueues the round, xwait completes it (asynchronous exchange)
    void* xuser = nullptr;
    bool xpending = false;                     // an enqueued round has not been waited for yet
    cognn_exchange_wait_round_fn xwait_round = nullptr;   // optional: completes the rounds up to a given one (chunked pipelines)
    int64_t xbegun = 0, xdone = 0;             // rounds enqueued on this transport / of them completed (a prefix: transports complete in order)
    int chunks = 1;                            // COGNN_OPT_EXCHANGE_CHUNKS
    // original-gcn: per destination party the in-edge entries of its rows (source row, Scatter instance, position in the instance's
    // edge list) and per Scatter instance (client P, destination g) the two per-edge normalisers (build_original_index)
    struct OrigDst { uint32_t* rowptr = nullptr; uint32_t* src = nullptr; uint32_t* pair = nullptr; uint32_t* q = nullptr; int64_t entries = 0; };
    struct OrigPair { u64* n0 = nullptr; u64* n1 = nullptr; int64_t edges = 0; };
    std::vector<OrigDst> orig_dst;             // [g]
    std::vector<OrigPair> orig_pair;           // [P * k + g]
    bool started = false, timing = false;
    int64_t gemm_x_opened_for = -1;    // iteration whose PreScatter GEMM input was already opened by the previous ReLU close
    // share table of the current message-passing round
    int64_t tableRows = 0, aggRows = 0, inboxRows = 0, inboxLocalOff = 0, partRows = 0;
    int Fmp = 0;
    std::vector<int64_t> A_off, B_off;
    u64* table = nullptr;
    u64* aggOut = nullptr;
    uint32_t *agg_rowptr = nullptr, *agg_col = nullptr, *part_rowptr = nullptr, *part_col = nullptr;
    uint32_t *rem_rowptr = nullptr, *rem_col = nullptr;   // world > 1: the aggregate's entries that read RECEIVED rows (replicas, inbox)
    int64_t aggEdges = 0, partEdges = 0, remEdges = 0;
    // One partial-sum segment per (source rank, destination owner g): row i = sum over ALL parties Q hosted by the source rank
    // of Q's own-share rows over the edges Q -> rows_vid[i] (pre-summed on the sender: one row per destination vertex however
    // many of the sender's parties reach it).
    struct Seg { int dst_owner; int64_t rows, inbox_off, out_off; int src_rank, dst_rank; std::vector<uint64_t> rows_vid; };
    std::vector<Seg> segs;         // partial-sum segments this rank sends or receives
    std::vector<std::vector<double>> hostFeat;
    std::vector<std::vector<int32_t>> hostLabels;
    std::vector<double> w0, w1;
    double algo[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int agg_timer = T_AGG;                          // which of the two aggregate timers the current round's launches count under
    u64* wa[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // weight-averaging temporaries
    u64* wa_recv[2] = {nullptr, nullptr};                                    // [world x wa_stride] each
    size_t wa_stride = 0;                                                     // even element count: every rank's slot is 16-byte aligned

    int in() const { return cfg.input_dim; }
    int hid() const { return cfg.hidden_dim; }
    int lab() const { return cfg.num_labels; }
    int rank_of(int party) const { return party / m; }
    int co(int owner) const { return (owner + 1) % k; }
    // the rank that holds share p of owner o's vertex set (cognn_engine_config::placement)
    int holder(int o, int p) const { return cfg.placement == COGNN_PLACE_VERTEX_SET ? rank_of(o) : rank_of(p == 0 ? o : co(o)); }
    // the owners whose co-share rank r holds, in the order of that rank's table (and of the partial-sum segments sent to it)
    std::vector<int> cohosted_of(int r) const {
        std::vector<int> v;
        for (int p = r * m; p < (r + 1) * m; ++p) v.push_back(cfg.placement == COGNN_PLACE_VERTEX_SET ? p : (p + k - 1) % k);
        return v;
    }
    Side* side(int owner, int p) {
        for (auto& s : sides) if (s.owner == owner && s.p == p) return &s;
        return nullptr;
    }
};

namespace {

#define BE(call)                                                                   \
    do {                                                                           \
        if ((E->be->call) != 0) throw EngineError(std::string(E->be->cognn_last_error())); \
    } while (0)
// Queues the element-wise launches issued while it is alive (same kind, independent tensors: the sides of one phase) into
// shared launches - see cognn_batch_begin in include/cognn_hip.h.  Only around loops whose iterations do not depend on each other.
struct Batch {
    cognn_engine* E;
    explicit Batch(cognn_engine* e);
    ~Batch();
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;
};
Batch::Batch(cognn_engine* e) : E(e) { BE(cognn_batch_begin(E->ctx)); }
Batch::~Batch() { E->be->cognn_batch_end(E->ctx); }         // a failure here resurfaces at the next call (sticky HIP error)

template <class T>
T* dalloc(cognn_engine* E, size_t count) {
    void* p = nullptr;
    BE(cognn_malloc(E->ctx, &p, std::max<size_t>(count, 2) * sizeof(T)));
    E->allocs.push_back(p);
    E->alloc_bytes += (int64_t)(std::max<size_t>(count, 2) * sizeof(T));
    return (T*)p;
}
// product-share buffers are recycled: a consumed share goes back to the pool and serves a later deal of the same size
// (everything runs on one stream, so reuse is ordered after the last reader)
u64* c1_alloc(cognn_engine* E, int64_t elems) {
    auto& v = E->c1_pool[elems];
    if (!v.empty()) { u64* p = v.back(); v.pop_back(); return p; }
    return dalloc<u64>(E, (size_t)elems);
}
void c1_release(cognn_engine* E, Side& s, std::pair<int64_t, int> key) {
    auto f = s.c1.find(key);
    if (f == s.c1.end() || E->retain_offline) return;
    E->c1_pool[f->second.elems].push_back(f->second.ptr);
    s.c1.erase(f);
}
// HIP-event bracket of one phase of an iteration (cfg.verbose)
struct Phase {
    cognn_engine* E; int kind; bool on;
    Phase(cognn_engine* e, int k) : E(e), kind(k), on(e->cfg.verbose != 0) { if (on) BE(cognn_timer_begin(E->ctx, kind)); }
    void end() { if (on) { on = false; BE(cognn_timer_end(E->ctx, kind)); } }
    ~Phase() { if (on) E->be->cognn_timer_end(E->ctx, kind); }
};
template <class T>
T* upload(cognn_engine* E, const std::vector<T>& v) {
    T* d = dalloc<T>(E, v.size());
    if (!v.empty()) BE(cognn_memcpy_h2d(E->ctx, d, v.data(), v.size() * sizeof(T)));
    return d;
}

// A dealer stream of GAS iteration `it` is addressed by (seed, owner, it % epoch, op, slot) through the key derivation and by the
// epoch number through the salt that the device adds to every key (cognn_spec.h): the arguments of an epoch's kernels do not
// depend on the epoch, so a recorded epoch can be replayed.
bool original(const cognn_engine* E) { return E->cfg.variant == COGNN_VARIANT_ORIGINAL_GCN; }
int epoch_len(const cognn_engine* E) { return (original(E) ? 2 : 3) * E->cfg.num_layers; }   // getEpochLayerNum: original-gcn/gcn.h:842-845, optimize-gcn/gcn.h:942-945
cognn_keys keys(cognn_engine* E, u64 owner, int64_t it, int op) {
    cognn_keys k;
    cognn_opkeys o = cognn_make_opkeys(E->cfg.seed, owner, (u64)(it % epoch_len(E)), (u64)op);
    // eager launches: the epoch salt is added to the keys here, on the host, and the device-side salt stays 0 - nothing
    // process-global changes, so any number of engines / contexts may run side by side.  Recorded epochs need epoch-independent
    // kernel arguments: there the device adds it (cognn_set_epoch_salt, exclusive to one context at a time).
    const u64 add = E->graph_epochs ? 0 : E->salt_now;
    for (int i = 0; i < COGNN_SL_COUNT; ++i) k.k[i] = o.k[i] + add;
    return k;
}
void set_salt_value(cognn_engine* E, u64 salt) {
    if (E->salt_now == salt) return;
    if (E->graph_epochs) { BE(cognn_set_epoch_salt(E->ctx, salt)); E->salt_on_device = salt; }
    E->salt_now = salt;
}
void set_salt(cognn_engine* E, int64_t it) { set_salt_value(E, (u64)(it / epoch_len(E)) * COGNN_GAMMA); }
// layer-0 PreScatter product: the feature operand's mask A is dealt once (iteration 0), B and C per iteration
// (op = COGNN_OP_PS_GEMM for the forward product, COGNN_OP_AP_GEMM for the layer-0 weight gradient on the transposed features)
// Default: the mask is the one of iteration 0 in EVERY epoch - the device adds the current epoch's salt to whatever key it is
// given, so the salt is taken off here.  With recorded epochs (COGNN_OPT_GRAPH_EPOCHS) kernel arguments must not depend on the
// epoch: the mask is then dealt per epoch (salted like every other stream) and the opening is renewed at the first iteration of
// each epoch (open_features).  The shares after the product's truncation are the same either way: they depend on the exact
// product and on the truncation's streams, not on how the operands were masked.
cognn_keys feature_gemm_keys(cognn_engine* E, u64 owner, int64_t it, int op = COGNN_OP_PS_GEMM) {
    cognn_keys k = keys(E, owner, it, op), k0 = keys(E, owner, 0, COGNN_OP_PS_GEMM);
    const u64 off = E->graph_epochs ? 0 : E->salt_now;
    k.k[COGNN_SL_A0] = k0.k[COGNN_SL_A0] - off;
    k.k[COGNN_SL_A1] = k0.k[COGNN_SL_A1] - off;
    return k;
}

struct GemmSpec;
cognn_keys gemm_keys(cognn_engine* E, Side& s, int64_t it, const GemmSpec& g);
void open_features(cognn_engine* E);

u64 fx_llround(double x) { return (u64)(long long)llround(x * (double)COGNN_FX_ONE); }
u64 fx_trunc(double x) { return (u64)(x * (double)COGNN_FX_ONE); }   // static_cast as in gcn.h:676,678,764

// ---------------------------------------------------------------------------------------------
// exchange
// ---------------------------------------------------------------------------------------------
struct XList {
    std::vector<cognn_xfer> v;
    void send(int peer, void* p, int64_t bytes) { if (bytes > 0) v.push_back(cognn_xfer{peer, 1, p, bytes}); }
    void recv(int peer, void* p, int64_t bytes) { if (bytes > 0) v.push_back(cognn_xfer{peer, 0, p, bytes}); }
};
// completes the round that is still in flight (asynchronous exchange); a no-op otherwise
void exchange_wait(cognn_engine* E) {
    if (!E->xpending) return;
    E->xpending = false;
    E->xdone = E->xbegun;
    if (E->xwait(E->xuser) != 0) throw EngineError("engine: exchange wait function failed");
}
// completes the rounds up to and including `round` (numbered by xbegun at their start); later rounds stay in flight when the
// transport can tell them apart (cognn_exchange_wait_round_fn) - otherwise everything enqueued is completed
void exchange_wait_round(cognn_engine* E, int64_t round) {
    if (!E->xpending || round < E->xdone) return;
    if (!E->xwait_round || round + 1 >= E->xbegun) { exchange_wait(E); return; }
    E->xdone = round + 1;
    if (E->xwait_round(E->xuser, round) != 0) throw EngineError("engine: exchange wait function failed");
}
// starts a round.  With a wait function registered the call only enqueues the messages; whoever consumes received data - or
// overwrites a buffer that is being sent - calls exchange_wait first (for_sides does, before it touches a side whose peer is remote).
void run_exchange(cognn_engine* E, XList& xl, bool keep_inflight = false) {
    // keep_inflight: the round already in flight keeps going (its buffers are disjoint from this round's and from the kernels
    // launched in between); exchange_wait then completes both
    if (!keep_inflight) exchange_wait(E);
    if (xl.v.empty()) return;
    if (!E->xfn) throw EngineError("engine: world > 1 needs an exchange function (cognn_engine_set_exchange)");
    if (E->xfn(E->xuser, xl.v.data(), (int32_t)xl.v.size()) != 0) throw EngineError("engine: exchange function failed");
    ++E->rounds;
    ++E->xbegun;
    if (E->xwait) E->xpending = true;
    else E->xdone = E->xbegun;
}
void run_exchange_sync(cognn_engine* E, XList& xl) {
    run_exchange(E, xl);
    exchange_wait(E);
}
// pairwise swap of outbox j (elems u64 each) between the two sides of every owner
void exchange_ob(cognn_engine* E, int j, const std::vector<int64_t>& elems) {
    XList xl;
    for (size_t i = 0; i < E->sides.size(); ++i) {
        Side& s = E->sides[i];
        if (s.peer) continue;
        xl.send(s.peer_rank, s.ob[j], elems[i] * 8);
        xl.recv(s.peer_rank, s.ib[j], elems[i] * 8);
    }
    run_exchange(E, xl);
}
// two outboxes in ONE round (fewer, larger p2p groups: every round costs a host round trip through the exchange callback)
void exchange_ob2(cognn_engine* E, int j0, const std::vector<int64_t>& e0, int j1, const std::vector<int64_t>& e1) {
    XList xl;
    for (size_t i = 0; i < E->sides.size(); ++i) {
        Side& s = E->sides[i];
        if (s.peer) continue;
        xl.send(s.peer_rank, s.ob[j0], e0[i] * 8);
        xl.recv(s.peer_rank, s.ib[j0], e0[i] * 8);
        xl.send(s.peer_rank, s.ob[j1], e1[i] * 8);
        xl.recv(s.peer_rank, s.ib[j1], e1[i] * 8);
    }
    run_exchange(E, xl);
}
std::vector<int64_t> per_side(cognn_engine* E, int64_t (*f)(cognn_engine*, Side&)) {
    std::vector<int64_t> r;
    for (auto& s : E->sides) r.push_back(f(E, s));
    return r;
}

// ---------------------------------------------------------------------------------------------
// two-party stages (all hosted sides advance together)
// ---------------------------------------------------------------------------------------------
struct GemmSpec {            // logical Z[MxN] = X[MxK] . Wm[KxN]
    int64_t M, N, K;
    int transA;
    int op, top;             // dealer op ids for the product and its truncation
    int feature = 0;         // constant feature operand, opening cached in Side::featSum: 1 layer-0 forward product X.W0,
                             // 2 layer-0 weight gradient X^T.g (same mask, transposed use: transA = 2)
    int xsrc = 0;            // where the opening of X comes from: X_OPEN_HERE, X_H1E_FRESH (written by the ReLU close of the previous
                             // iteration, still to be exchanged), X_H1E_REUSE (the layer-1 forward opening, exchanged two iterations ago)
    int64_t akey_it = -1;    // iteration whose COGNN_OP_PS_GEMM A streams mask X (X_H1E_REUSE); -1: this product's own streams
    int transB = 0;          // Wm(side) is stored [N x K] (a weight matrix used transposed, gcn.h:648): its opening reads it across
};
enum { X_OPEN_HERE = 0, X_H1E_FRESH = 1, X_H1E_REUSE = 2 };

// Runs fn(side, index) for every hosted side: first the sides whose peer lives on this rank, then - once the exchange round
// that may still be in flight has completed - the sides whose peer is remote.  With an asynchronous exchange the interior
// sides' kernels overlap the boundary sides' messages.  batched: the calls are independent element-wise launches of one
// kind (see Batch).
// Co-located share-holders (both sides of an owner on this rank) run their two-party steps as ONE pair chain per owner
// (cognn_pair_chain_u64: both sides' local arithmetic in one kernel, opened values handed over in registers) instead of
// open -> HBM -> close passes; the per-side stages below then skip those sides.
bool paired(const cognn_engine* E, const Side& s) { return E->pair_fusion && s.peer != nullptr; }
// product-buffer bookkeeping (Side::z_dirty / z_zero): small products are handed back clean by their consumer
const int64_t kClearMaxElems = 1 << 19;                    // 4 MiB per side: above that the extra writes cost more than a zeroing launch
bool z_is_zero(const Side& s, int64_t n) { return s.z_dirty == 0 && s.z_zero >= n; }
void z_written(Side& s, int64_t n) { s.z_dirty = std::max(s.z_dirty, n); }
bool z_clear_wanted(const Side& s, int64_t n) { return n <= kClearMaxElems && n >= s.z_dirty; }   // the clear leaves the whole buffer clean
void z_cleared(Side& s, int64_t n) { if (n >= s.z_dirty) s.z_dirty = 0; }
// A pair chain writes the opening of the step that follows it ONCE, as the sum of both parties' shares of it, into the owner
// side's buffer (COGNN_PC_OPEN_SUM): both sides of the pair read it from there as a pre-summed operand.
template <class Sel>
const u64* pair_opening(Side& s, Sel sel) { return s.p == 0 ? sel(s) : sel(*s.peer); }
// materialises a deferred ReLU' selection (Side::cur_mask) for a reader other than the pair chain it was deferred for
void apply_cur_mask(cognn_engine* E, Side& s) {
    if (!s.cur_mask) return;
    u64* dstb = (s.cur == s.buf[1]) ? s.buf[0] : s.buf[1];
    BE(cognn_mask_select_u64(E->ctx, dstb, s.cur, s.cur_mask, (int64_t)s.n * s.curF));
    s.cur = dstb;
    s.cur_mask = nullptr;
}

// lanes > 1 (independent multi-launch sequences per side, disjoint buffers): the sides of a pass go round-robin to that many
// launch lanes (cognn_lane_begin), joined before the pass ends.
struct Lanes {
    cognn_engine* E;
    int n, next = 0;
    Lanes(cognn_engine* e, int lanes) : E(e), n(lanes) { if (n > 1) BE(cognn_lane_begin(E->ctx, n)); }
    void advance() { if (n > 1) { BE(cognn_lane_select(E->ctx, next)); next = (next + 1) % n; } }
    ~Lanes() { if (n > 1) E->be->cognn_lane_end(E->ctx); }   // a failure here resurfaces at the next call (sticky HIP error)
};
template <class Fn>
void for_sides(cognn_engine* E, bool batched, Fn fn, bool skip_paired = false, int lanes = 0) {
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) exchange_wait(E);
        auto body = [&] {
            Lanes ln(E, lanes);
            for (size_t i = 0; i < E->sides.size(); ++i) {
                Side& s = E->sides[i];
                if ((s.peer != nullptr) != (pass == 0) || (skip_paired && paired(E, s))) continue;
                ln.advance();
                fn(s, i);
            }
        };
        if (batched) { Batch batch(E); body(); }
        else body();
    }
}
// The open -> exchange -> close [-> exchange -> close ...] steps of an element-wise stage.  steps[j].fn(side, i) is the call
// for_sides would issue for step j (it consumes what round j - 1 delivered); steps[j].msg(xl, side, i, c, C), if set, appends the
// messages that step j's output needs exchanged - chunk c of C of side i (C = 1: whole tensors; element ranges by
// cognn_chunk_range, msg_range).  Sides whose peer is hosted here run step j as one launch, beside the remote sides' messages.
// Sides whose peer is remote, with COGNN_OPT_EXCHANGE_CHUNKS = C > 1: step j runs chunk by chunk under the chunk window; chunk
// c's messages are enqueued as their own round right away and travel while chunk c + 1 is computed, and step j + 1 waits, chunk
// by chunk, for that chunk's round only (cognn_exchange_wait_round_fn) while the later ones are still in flight.
// whole: the step's calls are not chunk-safe (they write tensors of several sizes): they run once, unwindowed, before chunk 0's
// messages; the messages still go chunk by chunk.
struct Step {
    std::function<void(Side&, size_t)> fn;
    std::function<void(XList&, Side&, size_t, int, int)> msg;
    bool whole = false;
};
struct ChunkGuard {           // the window never outlives the step that set it (an exception included)
    cognn_engine* E;
    ~ChunkGuard() { E->be->cognn_ctx_set_chunk(E->ctx, 0, 1); }
};
void msg_range(XList& xl, Side& s, u64* out, u64* in, int64_t elems, int c, int C) {
    int64_t lo, hi;
    cognn_chunk_range(elems, c, C, &lo, &hi);
    xl.send(s.peer_rank, out + lo, (hi - lo) * 8);
    xl.recv(s.peer_rank, in + lo, (hi - lo) * 8);
}
void chunked_rounds(cognn_engine* E, const std::vector<Step>& steps, bool skip_paired = false) {
    bool remote = false;
    for (auto& s : E->sides) remote = remote || !s.peer;
    const int C = remote ? E->chunks : 1;
    auto each = [&](bool local, const std::function<void(Side&, size_t)>& fn) {
        if (!fn) return;
        Batch batch(E);
        for (size_t i = 0; i < E->sides.size(); ++i) {
            Side& s = E->sides[i];
            if ((s.peer != nullptr) != local || (skip_paired && paired(E, s))) continue;
            fn(s, i);
        }
    };
    std::vector<int64_t> ticket((size_t)C, -1), next((size_t)C, -1);
    for (size_t j = 0; j < steps.size(); ++j) {
        const Step& st = steps[j];
        each(true, st.fn);
        if (j == 0) exchange_wait(E);                      // (the remote sides' inputs may still be arriving)
        ChunkGuard guard{E};
        for (int c = 0; c < C; ++c) {
            if (ticket[(size_t)c] >= 0) exchange_wait_round(E, ticket[(size_t)c]);
            if (st.whole) { if (c == 0) each(false, st.fn); }
            else {
                if (C > 1) BE(cognn_ctx_set_chunk(E->ctx, c, C));
                each(false, st.fn);
                if (C > 1) BE(cognn_ctx_set_chunk(E->ctx, 0, 1));
            }
            next[(size_t)c] = -1;
            if (!st.msg) continue;
            XList xl;
            for (size_t i = 0; i < E->sides.size(); ++i)
                if (!E->sides[i].peer && !(skip_paired && paired(E, E->sides[i]))) st.msg(xl, E->sides[i], i, c, C);
            if (!xl.v.empty()) { run_exchange(E, xl, true); next[(size_t)c] = E->xbegun - 1; }
        }
        ticket = next;
    }
    exchange_wait(E);                                      // (nothing is left in flight unless the last step had messages)
}
// the pair chains of one phase: filled per owner (from its p = 0 side), launched together
struct PairChains {
    std::vector<cognn_pair_chain> v;
    cognn_pair_chain& add(Side& s0, const u64* x0, const u64* x1, int64_t rows, int64_t F) {
        cognn_pair_chain c;
        memset(&c, 0, sizeof(c));
        c.x[0] = x0; c.x[1] = x1; c.rows = rows; c.F = F;
        v.push_back(c);
        (void)s0;
        return v.back();
    }
    void launch(cognn_engine* E) {
        if (!v.empty()) BE(cognn_pair_chain_u64(E->ctx, v.data(), (int32_t)v.size()));
        v.clear();
    }
};
// COGNN_OPT_DEALER_STREAMS: the dealt slab of a chain / the dealt A mask of a product, materialised at first use
enum { DEAL_GEMM_CHAIN = 0, DEAL_SCALE_CHAIN = 1, DEAL_GATHER_CHAIN = 2, DEAL_RELU_CHAIN = 3, DEAL_GEMM_A0 = 4, DEAL_GEMM_A1 = 5 };
bool streams_on(const cognn_engine* E) { return E->dealer_streams != 0 && E->retain_offline; }
void attach_dealt(cognn_engine* E, cognn_pair_chain& c, int owner, int64_t it, int place) {
    if (!streams_on(E)) return;
    const int64_t slots = E->be->cognn_pair_chain_dealt_slots(c.flags, (c.open[0] || c.open[1]) ? 1 : 0);
    if (slots <= 0 || c.rows * c.F <= 0) return;
    if (E->dealer_streams == 2 && (c.F & 1)) return;       // (the corrections-only form is built for the 16-byte lanes; an odd width regenerates - same values)
    auto key = std::make_tuple(owner, it, place);
    auto f = E->dealt.find(key);
    if (f == E->dealt.end()) {
        u64* slab = dalloc<u64>(E, (size_t)(slots * c.rows * c.F));
        E->dealt_bytes += slots * c.rows * c.F * 8;
        BE(cognn_pair_chain_deal_u64(E->ctx, &c, slab));
        f = E->dealt.emplace(key, slab).first;
    }
    c.dealt = f->second;
    if (E->dealer_streams == 2) c.flags |= COGNN_PC_DEALT_MINIMAL;
}
// slots of a chain's slab that the launch reads per element: all of them, or (corrections-only form) r_1, r'_1 per truncation, c_1 per
// element-wise triple, c_1 and g of the ReLU
double dealt_slots_read(cognn_engine* E, const cognn_pair_chain& c) {
    if (!c.dealt) return 0.0;
    if (!(c.flags & COGNN_PC_DEALT_MINIMAL)) return (double)E->be->cognn_pair_chain_dealt_slots(c.flags, (c.open[0] || c.open[1]) ? 1 : 0);
    return ((c.flags & COGNN_PC_TRUNC_IN) ? 2.0 : 0.0) + ((c.flags & COGNN_PC_SCALE) ? 3.0 : 0.0) + ((c.flags & COGNN_PC_RELU) ? 2.0 : 0.0);
}
const u64* dealt_mask(cognn_engine* E, int owner, int64_t it, int place, u64 key, int64_t elems) {
    if (!streams_on(E) || elems <= 0 || E->dealer_streams == 2) return nullptr;   // (a product's A mask is the party's own PRG stream: the corrections-only form regenerates it)
    auto k = std::make_tuple(owner, it, place);
    auto f = E->dealt.find(k);
    if (f == E->dealt.end()) {
        u64* m = dalloc<u64>(E, (size_t)elems);
        E->dealt_bytes += elems * 8;
        BE(cognn_gemm_mask_fill_u64(E->ctx, m, key, elems));   // (a product's A mask: limb-form values)
        f = E->dealt.emplace(k, m).first;
    }
    return f->second;
}
// what follows a Beaver product on the same tensor (PreScatterComp: product, then row scale, gcn.h:233-254)
struct FollowScale {
    int op = 0, top = 0;
    std::function<u64*(Side&)> dst;
    int64_t it = -1;                                  // the iteration whose scale this is (its dealer streams); -1: the product's own
    std::function<const uint8_t*(Side&)> mask;        // a public selection between the truncation and the scale (COGNN_PC_MASK_AFTER_TRUNC)
    explicit operator bool() const { return (bool)dst; }
};

// truncation of every side's `x` (elems) scaled by `mul`; result written/applied to dst(side)
template <class DstFn>
void trunc_stage(cognn_engine* E, int64_t it, int op, u64 mul, const std::vector<u64*>& x, const std::vector<int64_t>& elems,
                 DstFn dst, int mode, u64 owner_override = ~0ull, bool skip_paired = false) {
    std::vector<Step> steps(2);
    steps[0].fn = [&](Side& s, size_t i) {
        cognn_keys k = keys(E, owner_override == ~0ull ? (u64)s.owner : owner_override, it, op);
        BE(cognn_trunc_open_u64(E->ctx, s.ob[2], x[i], mul, &k, s.p, elems[i]));
    };
    steps[0].msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(xl, s, s.ob[2], s.ib[2], elems[i], c, C); };
    steps[1].fn = [&](Side& s, size_t i) {
        cognn_keys k = keys(E, owner_override == ~0ull ? (u64)s.owner : owner_override, it, op);
        BE(cognn_trunc_close_u64(E->ctx, dst(s), s.p == 0 ? s.ob[2] : nullptr, s.p == 0 ? s.ib[2] : nullptr, &k, s.p, mode, elems[i]));
    };
    chunked_rounds(E, steps, skip_paired);
}

// dealer streams of a Beaver product: its own (owner, iteration, op) streams, except that a reused operand keeps the A streams
// it was first masked with
cognn_keys gemm_keys(cognn_engine* E, Side& s, int64_t it, const GemmSpec& g) {
    if (g.feature) return feature_gemm_keys(E, s.owner, it, g.op);
    cognn_keys k = keys(E, s.owner, it, g.op);
    if (g.akey_it >= 0) {
        cognn_keys ka = keys(E, s.owner, g.akey_it, COGNN_OP_PS_GEMM);
        k.k[COGNN_SL_A0] = ka.k[COGNN_SL_A0];
        k.k[COGNN_SL_A1] = ka.k[COGNN_SL_A1];
    }
    return k;
}

// closing step of a truncation; open_next (optional) returns the mask key of the op that consumes dst(side): the close then
// also writes that op's opening E = dst - mask into ob[0] (one pass less, see cognn_trunc_close_open_u64)
struct OpenNext {
    std::function<u64(Side&, int)> keyp;   // mask key of party p's share of the operand; empty: plain close
    int ob = 0;                            // outbox that receives the opening (0: left / element-wise operand, 1: right GEMM operand)
    OpenNext() {}
    bool reveal = false;                   // no opening follows, but the owner needs the result itself (softmax): with public openings its
                                           // close writes z = y_0 + y_1 into ob[0] and the co-party sends nothing
    OpenNext(std::function<u64(Side&, int)> k, int o = 0) : keyp(std::move(k)), ob(o) {}
    static OpenNext Reveal() { OpenNext r; r.reveal = true; return r; }
    u64 key(Side& s) const { return keyp(s, s.p); }
    explicit operator bool() const { return (bool)keyp; }
};
// Public openings (COGNN_OPT_PUBLIC_OPENINGS): a truncation whose result feeds a Beaver opening is closed by
// BOTH parties from both opened values, and each derives the next opening E itself (cognn_trunc_close_pub_u64): ob[] then
// holds E, not E_p, the consumer takes it as the pre-summed opening and the exchange round that carried E_p disappears.
// Applies to the sides that are not part of a pair chain.
bool pub_open(const cognn_engine* E, const Side& s) { return E->public_openings && !paired(E, s); }
template <class DstFn>
void trunc_close_one(cognn_engine* E, int64_t it, int top, DstFn& dst, const std::vector<int64_t>& elems, const OpenNext& open_next, Side& s, size_t i) {
    cognn_keys tk = keys(E, s.owner, it, top);
    const u64* c0 = s.p == 0 ? s.ob[2] : nullptr;
    const u64* c1 = s.p == 0 ? s.ib[2] : nullptr;
    if (open_next.reveal && pub_open(E, s) && s.p == 0)
        BE(cognn_trunc_close_pub_u64(E->ctx, dst(s), s.ob[0], s.ob[2], s.ib[2], &tk, 0, 0, 0, 1, elems[i]));
    else if (open_next && pub_open(E, s))
        BE(cognn_trunc_close_pub_u64(E->ctx, dst(s), s.ob[open_next.ob], s.p == 0 ? s.ob[2] : s.ib[2], s.p == 0 ? s.ib[2] : s.ob[2], &tk, s.p,
                                     open_next.keyp(s, 0), open_next.keyp(s, 1), 0, elems[i]));
    else if (open_next) BE(cognn_trunc_close_open_u64(E->ctx, dst(s), s.ob[open_next.ob], c0, c1, &tk, s.p, open_next.key(s), elems[i]));
    else BE(cognn_trunc_close_u64(E->ctx, dst(s), c0, c1, &tk, s.p, 0, elems[i]));
}
// the exchange of the truncation openings in ob[2] and the closes that consume them (open: what still has to produce ob[2] for
// the sides whose peer is remote - empty when that already happened)
template <class DstFn>
void trunc_exchange_close(cognn_engine* E, int64_t it, int top, DstFn dst, const std::vector<int64_t>& elems, const OpenNext& open_next,
                          bool skip_paired, std::function<void(Side&, size_t)> open = nullptr) {
    std::vector<Step> steps(2);
    steps[0].fn = open;
    steps[0].msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(xl, s, s.ob[2], s.ib[2], elems[i], c, C); };
    steps[1].fn = [&](Side& s, size_t i) { trunc_close_one(E, it, top, dst, elems, open_next, s, i); };
    chunked_rounds(E, steps, skip_paired);
}

// Beaver GEMM for every side: X(side), Wm(side) -> truncated product written to dst(side)
// pairs_raw: the co-located pairs' product shares stay in zbuf as the product kernels left them - the caller's chain consumes them
// (weight_update_chain); returns whether those are raw products (C_p still to be added)
template <class XFn, class WFn, class SpecFn, class DstFn>
bool gemm_stage(cognn_engine* E, int64_t it, XFn X, WFn Wm, SpecFn spec, DstFn dst, bool x_opened = false,
                const OpenNext& open_next = OpenNext(), bool w_opened = false, const FollowScale& follow = FollowScale(),
                bool pairs_raw = false) {
    // follow (co-located pairs only): the row scale that consumes the product joins the pair chain; the caller's rowscale_stage
    // then handles the other sides
    // w_opened: ob[1] already holds F_p = W_p - B_p (written by the truncation close that produced W)
    const size_t ns = E->sides.size();
    std::vector<int64_t> e0(ns), e1(ns), eo(ns);
    const bool feature = spec(E->sides[0]).feature != 0;
    const int xsrc = x_opened ? X_H1E_FRESH : spec(E->sides[0]).xsrc;
    auto gkeys = [&](Side& s, const GemmSpec& g) { return gemm_keys(E, s, it, g); };
    for_sides(E, true, [&](Side& s, size_t i) {
        GemmSpec g = spec(s);
        cognn_keys k = gkeys(s, g);
        if (xsrc == X_OPEN_HERE && !feature)
            BE(cognn_mask_open_u64(E->ctx, s.ob[0], X(s), k.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], g.M, g.K, g.transA | COGNN_MASK_OPEN_LIMB));
        if (!w_opened) BE(cognn_mask_open_u64(E->ctx, s.ob[1], Wm(s), k.k[s.p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], g.K, g.N, g.transB ? 3 : 0));
        e0[i] = g.M * g.K; e1[i] = g.K * g.N; eo[i] = g.M * g.N;
    });
    const bool w_public = w_opened && E->public_openings;   // ob[1] holds F itself on the sides outside pair chains: nothing to exchange
    if (feature || xsrc == X_H1E_REUSE) {
        if (!w_public) exchange_ob(E, 1, e1);               // the opening of X was exchanged earlier (start() / two iterations ago)
    } else if (xsrc == X_H1E_FRESH) {
        XList xl;                                           // the ReLU close left E_p in h1E: it travels with the W opening
        for (size_t i = 0; i < ns; ++i) {
            Side& s = E->sides[i];
            if (s.peer) continue;
            xl.send(s.peer_rank, s.h1E, e0[i] * 8);
            xl.recv(s.peer_rank, s.h1E_peer, e0[i] * 8);
            xl.send(s.peer_rank, s.ob[1], e1[i] * 8);
            xl.recv(s.peer_rank, s.ib[1], e1[i] * 8);
        }
        run_exchange(E, xl);
    } else if (w_public) {
        exchange_ob(E, 0, e0);
    } else {
        exchange_ob2(E, 0, e0, 1, e1);
    }
    std::vector<u64*> z(ns);                                // (F = F0 + F1 is summed inside the product kernels)
    std::vector<const u64*> c1_of(ns, nullptr);            // side 1's dealt product share (the map entry may be released before its last use)
    const bool chunk_trunc = E->chunks > 1;                 // the truncation opening of a side whose peer is remote runs in row chunks
    GemmSpec g0 = spec(E->sides[0]);
    bool all_raw = true;
    for (auto& s : E->sides) { GemmSpec g = spec(s); all_raw = all_raw && E->be->cognn_beaver_gemm_fusable(g.M, g.N, g.K, g.transA); }
    // weight gradients (A stored transposed, K = rows of the side's party): one grouped launch of raw products when every side's
    // shape is served by the register-direct TN kernel (cognn_beaver_gemm_close_group_tn_u64)
    bool tn_group = E->gemm_group && ns <= 16 && g0.transA != 0;
    if (tn_group) {
        bool two = false;
        for (auto& s : E->sides) {                          // does any side pass an operand as two shares (E1 / F1, see the jobs below)?
            const bool h1e_sum = xsrc != X_OPEN_HERE && paired(E, s) && E->h1e_pairs_summed;
            const bool e_two = !(feature || h1e_sum), f_two = !(w_opened && (paired(E, s) || w_public));
            two = two || e_two || f_two;
        }
        for (auto& s : E->sides) {
            GemmSpec g = spec(s);
            tn_group = tn_group && g.transA == g0.transA && g.M == g0.M && g.N == g0.N &&
                       (g.K == 0 || E->be->cognn_beaver_gemm_tn_groupable(g.M, g.N, g.K, two ? 1 : 0));
        }
        if (tn_group) all_raw = true;                       // C_p joins in the truncation opening / the pair chain, as for the NN products
    }
    // every side's product is its own launch sequence (operand planes, product, truncation opening) on its own buffers: two
    // launch lanes, so that one side's start-up runs in the drain of another's - unless a product share still has to be dealt
    // here (its buffer comes from a pool that the releases below feed)
    // (small products are launch-bound: the fork / join would cost more than the overlap gains)
    bool dealt = true, large = false;
    for (auto& s : E->sides) {
        GemmSpec g = spec(s);
        dealt = dealt && (s.p != 1 || s.c1.count({it, g.op}));
        large = large || g.M * g.K * g.N >= (1ll << 27);
    }
    const int lanes = (dealt && large && E->gemm_lanes > 1 && ns > 1 && !E->graph_epochs) ? E->gemm_lanes : 0;   // (no auxiliary streams inside a recording)
    // timed as one phase on the engine's stream (the lanes overlap each other): all sides' products of this stage, with their
    // operand preparation (and, for sides whose peer is remote, the truncation opening and the wait for the peer's opening)
    bool tg_open = E->timing;
    if (E->timing) BE(cognn_timer_begin(E->ctx, T_GEMM));
    // The sides' products of this phase as ONE grouped launch (cognn_beaver_gemm_close_group_u64: every workgroup builds its job's
    // weight planes in its prologue): possible when all of them are raw fusable products of one (N, K) - the PreScatter products
    // and g = (p - y) . W^T.  First the sides whose peer is hosted here, then - once their openings have arrived - the others.
    bool same_nk = true;
    for (auto& s : E->sides) { GemmSpec g = spec(s); same_nk = same_nk && g.N == g0.N && g.K == g0.K && g.transA == 0; }
    const bool grouped = tn_group || (all_raw && same_nk && ns <= 16 && E->gemm_group);
    // the chain that consumes a co-located pair's product: truncation (+ the row scale that follows, + the next opening)
    auto chain_of = [&](Side& s) {                           // s: the pair's p = 0 side
        Side& t = *s.peer;
        GemmSpec g = spec(s);
        cognn_pair_chain c;
        memset(&c, 0, sizeof(c));
        c.x[0] = s.zbuf; c.x[1] = t.zbuf; c.rows = g.M; c.F = g.N;
        c.flags = COGNN_PC_TRUNC_IN | (all_raw ? 0 : COGNN_PC_NO_C);
        c.gemm_keys = gkeys(s, g);
        c.trunc_in_keys = keys(E, s.owner, it, g.top);
        if (all_raw) c.c1 = t.c1.at({it, g.op}).ptr;
        if (follow) {
            c.flags |= COGNN_PC_SCALE;
            c.scale[0] = s.svec; c.scale[1] = t.svec;
            const int64_t fit = follow.it >= 0 ? follow.it : it;
            c.scale_keys = keys(E, s.owner, fit, follow.op);
            c.scale_trunc_keys = keys(E, s.owner, fit, follow.top);
            c.out[0] = follow.dst(s); c.out[1] = follow.dst(t);
            if (follow.mask) { c.mask_in = follow.mask(s); c.flags |= COGNN_PC_MASK_AFTER_TRUNC; }
        } else {
            c.out[0] = dst(s); c.out[1] = dst(t);
            if (open_next) {
                c.open[0] = s.ob[open_next.ob]; c.flags |= COGNN_PC_OPEN_SUM;
                c.open_key[0] = open_next.key(s); c.open_key[1] = open_next.key(t);
            }
        }
        return c;
    };
    // ... as the EPILOGUE of the p = 1 side's product when the grouped launch takes one (whole-K form, no opening to write): the
    // p = 0 sides' products go first, then one launch computes the p = 1 products and runs the chains on their tiles in registers
    // - the p = 1 product is never written or re-read, and the chain launch disappears
    bool epi = false;
    if (grouped && !tn_group && all_raw && E->gemm_epilogue && !pairs_raw && !streams_on(E) && (!open_next || follow) && !follow.mask) {
        int64_t tilesB = 0; int nB = 0;
        for (auto& s : E->sides) if (paired(E, s) && s.p == 1) { tilesB += (spec(s).M + 15) / 16; ++nB; }
        epi = nB >= 1 && nB <= 8 && E->be->cognn_beaver_gemm_group_takes_epilogue(g0.N, g0.K, tilesB) != 0;
    }
    if (grouped) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) exchange_wait(E);
            std::vector<cognn_gemm_job> jobs, jobs_epi;
            std::vector<cognn_pair_chain> chains;
            chains.reserve(ns);                                // (the jobs point into it)
            std::vector<size_t> idx;
            for (size_t i = 0; i < ns; ++i) {
                Side& s = E->sides[i];
                if ((s.peer != nullptr) != (pass == 0)) continue;
                GemmSpec g = spec(s);
                cognn_keys k = gkeys(s, g);
                const bool h1e_sum = xsrc != X_OPEN_HERE && paired(E, s) && E->h1e_pairs_summed;
                cognn_gemm_job J;
                memset(&J, 0, sizeof(J));
                J.E0 = feature ? s.featSum : h1e_sum ? pair_opening(s, [](Side& x) { return x.h1E; }) : xsrc != X_OPEN_HERE ? s.h1E : s.ob[0];
                J.E1 = (feature || h1e_sum) ? nullptr : xsrc != X_OPEN_HERE ? s.h1E_peer : s.ib[0];
                if (s.p == 1 && !s.c1.count({it, g.op})) {            // dealer product share not precomputed: do it now
                    u64* c = c1_alloc(E, eo[i]);
                    BE(cognn_dealer_gemm_c1_u64(E->ctx, c, &k, g.M, g.N, g.K, g.transA, s.scratch, s.scratch + g.M * g.K));
                    s.c1.emplace(std::make_pair(it, g.op), Side::C1{c, eo[i]});
                }
                const bool f_sum = w_opened && (paired(E, s) || w_public);
                J.F0 = (w_opened && paired(E, s)) ? pair_opening(s, [](Side& x) { return x.ob[1]; }) : s.ob[1];
                J.F1 = f_sum ? nullptr : s.ib[1];
                J.Z = s.zbuf; J.keys = k; J.p = s.p; J.M = g.M; J.K = g.K; J.scratch = s.scratch;
                J.Z_zeroed = z_is_zero(s, eo[i]) ? 1 : 0;
                z_written(s, eo[i]);
                if (g.feature == 1) { J.E_presplit = s.featPl; J.A_presplit = E->graph_epochs ? nullptr : s.featMaskPl; }   // (recorded epochs: a new mask every epoch)
                if (g.feature == 2 && tn_group && !E->graph_epochs) { J.E_presplit = s.featTnPl; J.A_presplit = s.featMaskTnPl; }
                if (!tn_group) J.A_dealt = dealt_mask(E, s.owner, it, s.p == 0 ? DEAL_GEMM_A0 : DEAL_GEMM_A1, k.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], g.M * g.K);
                if (epi && paired(E, s) && s.p == 1) {
                    chains.push_back(chain_of(*s.peer));
                    J.epilogue = &chains.back();
                    jobs_epi.push_back(J);
                } else jobs.push_back(J);
                idx.push_back(i);
                if (E->timing) E->algo[J.epilogue ? T_GEMM_EPI : T_GEMM] += 2.0 * 36 * 2 * (double)g.M * g.K * g.N;
                z[i] = s.zbuf;
            }
            if (jobs.empty() && jobs_epi.empty()) continue;
            if (tn_group) BE(cognn_beaver_gemm_close_group_tn_u64(E->ctx, jobs.data(), (int32_t)jobs.size(), g0.M, g0.N, g0.transA == 2 ? 1 : 0));
            else {
                if (!jobs.empty()) BE(cognn_beaver_gemm_close_group_u64(E->ctx, jobs.data(), (int32_t)jobs.size(), g0.N, g0.K, 1));
                if (!jobs_epi.empty()) {                       // (timed apart from the pure products: this launch also does the chains' work)
                    if (tg_open) { BE(cognn_timer_end(E->ctx, T_GEMM)); tg_open = false; }   // (nothing but this launch follows for these sides)
                    if (E->timing) BE(cognn_timer_begin(E->ctx, T_GEMM_EPI));
                    BE(cognn_beaver_gemm_close_group_u64(E->ctx, jobs_epi.data(), (int32_t)jobs_epi.size(), g0.N, g0.K, 1));
                    if (E->timing) BE(cognn_timer_end(E->ctx, T_GEMM_EPI));
                }
            }
            Batch batch(E);                                    // the truncation openings of the sides outside pair chains: one launch
            for (size_t i : idx) {
                Side& s = E->sides[i];
                if (s.p == 1) c1_of[i] = s.c1.at({it, spec(s).op}).ptr;
                if (paired(E, s) || (chunk_trunc && !s.peer)) continue;   // (chunked: opened chunk by chunk below)
                GemmSpec g = spec(s);
                cognn_keys k = gkeys(s, g), tk = keys(E, s.owner, it, g.top);
                BE(cognn_trunc_open_add_u64(E->ctx, s.ob[2], s.zbuf, c1_of[i], &k, &tk, s.p, eo[i]));
            }
        }
        for (auto& s : E->sides)
            if (s.p == 1 && !paired(E, s)) c1_release(E, s, {it, spec(s).op});   // consumed: the buffer serves a later deal
    } else
    for_sides(E, false, [&](Side& s, size_t i) {
        GemmSpec g = spec(s);
        cognn_keys k = gkeys(s, g);
        const bool h1e_sum = xsrc != X_OPEN_HERE && paired(E, s) && E->h1e_pairs_summed;   // written by a pair chain as E_0 + E_1
        const u64* e_own = feature ? s.featSum : h1e_sum ? pair_opening(s, [](Side& x) { return x.h1E; })   // featSum is already the sum of both shares
                                   : xsrc != X_OPEN_HERE ? s.h1E : s.ob[0];
        const u64* e_peer = (feature || h1e_sum) ? nullptr : xsrc != X_OPEN_HERE ? s.h1E_peer : s.ib[0];
        const u64* c1 = nullptr;
        if (s.p == 1) {
            auto f = s.c1.find({it, g.op});
            if (f == s.c1.end()) {                       // dealer product share not precomputed: do it now
                u64* c = c1_alloc(E, eo[i]);
                BE(cognn_dealer_gemm_c1_u64(E->ctx, c, &k, g.M, g.N, g.K, g.transA, s.scratch, s.scratch + g.M * g.K));
                f = s.c1.emplace(std::make_pair(it, g.op), Side::C1{c, eo[i]}).first;
            }
            c1 = f->second.ptr;
        }
        // all_raw: fused single-launch product without C_p; C_p joins in the truncation opening below
        // the opened right operand: two shares (ob[1], ib[1]) - or already F itself: derived by both parties (public openings) or
        // written once by the pair chain that produced W
        const bool f_sum = w_opened && (paired(E, s) || w_public);
        const u64* f_own = (w_opened && paired(E, s)) ? pair_opening(s, [](Side& x) { return x.ob[1]; }) : s.ob[1];
        BE(cognn_beaver_gemm_close2_u64(E->ctx, s.zbuf, e_own, e_peer, f_own, f_sum ? nullptr : s.ib[1], c1, &k, s.p, g.M, g.N,
                                        g.K, g.transA, s.scratch, all_raw ? 1 : 0));
        if (E->timing) E->algo[T_GEMM] += 2.0 * 36 * 2 * (double)g.M * g.K * g.N;
        c1_of[i] = c1;
        z_written(s, eo[i]);
        if (all_raw && !paired(E, s) && !(chunk_trunc && !s.peer)) {
            cognn_keys tk = keys(E, s.owner, it, g.top);
            BE(cognn_trunc_open_add_u64(E->ctx, s.ob[2], s.zbuf, c1, &k, &tk, s.p, eo[i]));
        }
        if (s.p == 1 && !paired(E, s)) c1_release(E, s, {it, g.op});   // consumed: the buffer serves a later deal
        z[i] = s.zbuf;
    }, false, lanes);
    if (tg_open) BE(cognn_timer_end(E->ctx, T_GEMM));
    // co-located pairs: truncation of the product (+ the row scale that follows) as one chain per owner
    if (!pairs_raw) {
        PairChains pc;
        for (auto& s : E->sides) {
            if (!paired(E, s) || s.p != 0 || epi) continue;  // (epi: the chains ran inside the p = 1 sides' product launch)
            Side& t = *s.peer;
            GemmSpec g = spec(s);
            pc.v.push_back(chain_of(s));
            cognn_pair_chain& c = pc.v.back();
            if (z_clear_wanted(s, g.M * g.N) && z_clear_wanted(t, g.M * g.N)) { c.flags |= COGNN_PC_CLEAR_INPUT; z_cleared(s, g.M * g.N); z_cleared(t, g.M * g.N); }
            attach_dealt(E, c, s.owner, it, DEAL_GEMM_CHAIN);
        }
        if (streams_on(E)) { bool all = true; for (auto& c : pc.v) all = all && c.dealt; if (!all) for (auto& c : pc.v) c.dealt = nullptr; }
        pc.launch(E);
        for (auto& s : E->sides)
            if (paired(E, s) && s.p == 1) c1_release(E, s, {it, spec(s).op});
    }
    // all GEMMs of one stage share the truncation op id
    // (chunked: the sides whose peer is remote open their truncation here, chunk by chunk, each chunk's messages leaving at once)
    trunc_exchange_close(E, it, g0.top, dst, eo, open_next, true, [&](Side& s, size_t i) {
        cognn_keys tk = keys(E, s.owner, it, g0.top);
        if (!all_raw) BE(cognn_trunc_open_u64(E->ctx, s.ob[2], z[i], 1, &tk, s.p, eo[i]));
        else if (chunk_trunc && !s.peer) {
            GemmSpec g = spec(s);
            cognn_keys k = gkeys(s, g);
            BE(cognn_trunc_open_add_u64(E->ctx, s.ob[2], s.zbuf, c1_of[i], &k, &tk, s.p, eo[i]));
        }
    });
    return all_raw;
}

// row scale by the (owner-known) normaliser followed by truncation; x(side) [n x F] -> dst(side)
enum { E_FROM_X = 0, E_IN_X = 1, E_IN_OB0 = 2 };
template <class XFn, class DstFn>
void rowscale_stage(cognn_engine* E, int64_t it, int op, int top, int F, XFn X, DstFn dst, int e_mode = E_FROM_X,
                    const OpenNext& open_next = OpenNext(), bool pairs_done = false, bool relu_follows = false) {
    // e_mode: E_FROM_X  open E_p = X_p - a_p here;
    //         E_IN_X    X(side) already holds E_p (written by the gather epilogue);
    //         E_IN_OB0  ob[0] already holds E_p (written by the truncation close that produced X)
    // (modes of the sides whose peer is remote; a co-located pair always hands its chain the plain X)
    // pairs_done: the co-located pairs ran this scale inside the chain of the product before it (gemm_stage, FollowScale);
    // relu_follows: the ReLU of ApplyComp consumes the result - co-located pairs run it in the same chain (relu_stage skips them)
    const bool e_opened = e_mode == E_IN_X;
    const bool e_public = e_mode == E_IN_OB0 && E->public_openings;   // ob[0] holds E itself (cognn_trunc_close_pub_u64)
    const size_t ns = E->sides.size();
    std::vector<int64_t> eF(ns), e1(ns);
    if (!pairs_done) {
        PairChains pc;
        for (auto& s : E->sides) {
            if (!paired(E, s) || s.p != 0) continue;
            Side& t = *s.peer;
            const uint8_t* mask_in = nullptr;               // the deferred ReLU' selection, if both sides still carry it
            if (s.cur_mask || t.cur_mask) {
                if (s.cur_mask && t.cur_mask && X(s) == s.cur && X(t) == t.cur) { mask_in = s.cur_mask; s.cur_mask = t.cur_mask = nullptr; }
                else { apply_cur_mask(E, s); apply_cur_mask(E, t); }
            }
            cognn_pair_chain& c = pc.add(s, X(s), X(t), s.n, F);
            c.flags = COGNN_PC_SCALE;
            c.mask_in = mask_in;
            c.scale[0] = s.svec; c.scale[1] = t.svec;
            c.scale_keys = keys(E, s.owner, it, op);
            c.scale_trunc_keys = keys(E, s.owner, it, top);
            if (relu_follows) {
                // H is the next iteration's PreScatter GEMM input (layer 1): straight into its h_t slot, with the Beaver opening
                // E_p = H_p - A_p of that product (gcn.h:230-239 of iteration it+1), as relu_stage does for the other sides
                c.flags |= COGNN_PC_RELU;
                c.relu_keys = keys(E, s.owner, it, COGNN_OP_AP_RELU);
                c.mask = s.relu_mask;
                cognn_keys nk = keys(E, s.owner, it + 1, COGNN_OP_PS_GEMM);
                c.out[0] = s.h1; c.out[1] = t.h1;
                c.open[0] = s.h1E; c.flags |= COGNN_PC_OPEN_SUM;      // E_0 + E_1 once, into the owner side's buffer (pair_opening)
                E->h1e_pairs_summed = true;
                c.open_key[0] = nk.k[COGNN_SL_A0]; c.open_key[1] = nk.k[COGNN_SL_A1]; c.flags |= COGNN_PC_OPEN_LIMB;   // (a product's A masks: limb form)
                if (E->forward_only) {                       // inference: the next product only reads the opening; h_t and the sign mask
                    c.out[0] = c.out[1] = nullptr;           // serve the backward pass, which will not run
                    c.mask = nullptr;
                }
            } else {
                c.out[0] = dst(s); c.out[1] = dst(t);
                if (open_next) {
                    c.open[0] = s.ob[open_next.ob]; c.flags |= COGNN_PC_OPEN_SUM;
                    c.open_key[0] = open_next.key(s); c.open_key[1] = open_next.key(t);
                }
            }
        }
        pc.launch(E);
    }
    for (size_t i = 0; i < ns; ++i) { eF[i] = (int64_t)E->sides[i].n * F; e1[i] = E->sides[i].n; }
    std::vector<Step> steps(2);
    // the openings: G_p = s_p - b_p (one value per row) and, unless it exists already, E_p = X_p - a_p.  Two tensors of different
    // sizes: not a chunk-window call; the messages leave in chunks all the same (G whole with chunk 0)
    steps[0].whole = true;
    steps[0].fn = [&](Side& s, size_t) {
        cognn_keys k = keys(E, s.owner, it, op);
        BE(cognn_rowscale_open_u64(E->ctx, e_mode == E_FROM_X ? s.ob[0] : nullptr, s.ob[1], X(s), s.svec, &k, s.p, s.n, F));
    };
    steps[0].msg = [&](XList& xl, Side& s, size_t i, int c, int C) {
        if (!e_public) msg_range(xl, s, e_opened ? X(s) : s.ob[0], s.ib[0], eF[i], c, C);   // (public: ob[0] holds E itself, only the scale openings travel)
        if (c == 0) msg_range(xl, s, s.ob[1], s.ib[1], e1[i], 0, 1);
    };
    steps[1].fn = [&](Side& s, size_t) {                    // the opened sums E0+E1, G0+G1 are formed inside the kernel
        cognn_keys k = keys(E, s.owner, it, op), tk = keys(E, s.owner, it, top);
        const u64* e_own = e_opened ? X(s) : s.ob[0];
        const u64* e_peer = e_public ? nullptr : e_opened ? (s.peer ? X(*s.peer) : s.ib[0]) : s.ib[0];
        BE(cognn_rowscale_close_u64(E->ctx, s.ob[2], e_own, e_peer, s.ob[1], s.ib[1], &k, &tk, s.p, s.n, F));
    };
    steps[1].msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(xl, s, s.ob[2], s.ib[2], eF[i], c, C); };
    steps.emplace_back();
    steps.back().fn = [&](Side& s, size_t i) { trunc_close_one(E, it, top, dst, eF, open_next, s, i); };
    chunked_rounds(E, steps, true);
}

void relu_stage(cognn_engine* E, int64_t it, bool e_opened, bool pairs_done) {
    // e_opened: ob[0] already holds E = z - a (written by the truncation close that produced z)
    // pairs_done: the co-located pairs ran the ReLU inside the chain of the scale before it (rowscale_stage, relu_follows)
    const int F = E->hid();
    std::vector<int64_t> eF(E->sides.size());
    if (!pairs_done) {
        PairChains pc;
        for (auto& s : E->sides) {
            if (!paired(E, s) || s.p != 0) continue;
            Side& t = *s.peer;
            cognn_pair_chain& c = pc.add(s, s.cur, t.cur, s.n, F);
            c.flags = COGNN_PC_RELU;
            c.relu_keys = keys(E, s.owner, it, COGNN_OP_AP_RELU);
            c.mask = s.relu_mask;
            cognn_keys nk = keys(E, s.owner, it + 1, COGNN_OP_PS_GEMM);
            c.out[0] = s.h1; c.out[1] = t.h1;
            c.open[0] = s.h1E; c.flags |= COGNN_PC_OPEN_SUM;      // E_0 + E_1 once, into the owner side's buffer (pair_opening)
                E->h1e_pairs_summed = true;
            c.open_key[0] = nk.k[COGNN_SL_A0]; c.open_key[1] = nk.k[COGNN_SL_A1]; c.flags |= COGNN_PC_OPEN_LIMB;   // (a product's A masks: limb form)
            if (E->forward_only) { c.out[0] = c.out[1] = nullptr; c.mask = nullptr; }
        }
        pc.launch(E);
    }
    for (auto& s : E->sides) if (paired(E, s)) s.cur = s.h1;
    for (size_t i = 0; i < E->sides.size(); ++i) eF[i] = (int64_t)E->sides[i].n * F;
    const bool e_public = e_opened && E->public_openings;   // ob[0] holds E itself: no exchange
    std::vector<Step> steps;
    if (!e_public) {
        steps.emplace_back();
        steps.back().fn = [&](Side& s, size_t i) {
            cognn_keys k = keys(E, s.owner, it, COGNN_OP_AP_RELU);
            // only E = z - a is opened online: g = t - b is input-independent and published by the dealer offline
            if (!e_opened) BE(cognn_relu_open_u64(E->ctx, s.ob[0], nullptr, s.cur, &k, s.p, eF[i]));
        };
        steps.back().msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(xl, s, s.ob[0], s.ib[0], eF[i], c, C); };
    }
    steps.emplace_back();
    steps.back().fn = [&](Side& s, size_t i) {
        cognn_keys k = keys(E, s.owner, it, COGNN_OP_AP_RELU);
        BE(cognn_relu_mul_u64(E->ctx, s.ob[2], s.ob[0], e_public ? nullptr : s.ib[0], nullptr, nullptr, &k, s.p, eF[i]));
    };
    steps.back().msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(xl, s, s.ob[2], s.ib[2], eF[i], c, C); };
    // H is the next iteration's PreScatter GEMM input (layer 1): write it straight into its h_t slot and emit the
    // Beaver opening E_p = H_p - A_p of that product in the same pass (gcn.h:230-239 of iteration it+1)
    steps.emplace_back();
    steps.back().fn = [&](Side& s, size_t i) {
        cognn_keys nk = keys(E, s.owner, it + 1, COGNN_OP_PS_GEMM);
        BE(cognn_relu_close_open_u64(E->ctx, s.h1, s.h1E, s.relu_mask, s.cur, s.ob[2], s.ib[2],
                                     nk.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], eF[i]));
    };
    chunked_rounds(E, steps, true);
    for (auto& s : E->sides) if (!paired(E, s)) s.cur = s.h1;
    E->gemm_x_opened_for = it + 1;
}

void softmax_stage(cognn_engine* E, int64_t it, bool revealed) {
    // revealed: the truncation close before this stage left z itself in the owner's ob[0] (sides outside pair chains)
    const int L = E->lab();
    XList xl;                                            // the co-party reveals its share of z to the owner
    for (auto& s : E->sides) {
        if (s.peer || (revealed && pub_open(E, s))) continue;
        if (s.p == 1) xl.send(s.peer_rank, s.cur, (int64_t)s.n * L * 8);
        else xl.recv(s.peer_rank, s.ib[0], (int64_t)s.n * L * 8);
    }
    run_exchange_sync(E, xl);
    std::vector<cognn_softmax_job> jobs;                   // every hosted side in one launch, the owners' metrics fused in
    for (auto& s : E->sides) {
        cognn_softmax_job j;
        memset(&j, 0, sizeof(j));
        j.keys = keys(E, s.owner, it, COGNN_OP_AP_SOFTMAX);
        j.p = s.p; j.rows = s.n;
        j.train_rows = (int64_t)((double)s.n * E->cfg.train_ratio);      // gcn.h:560
        j.val_rows = (int64_t)((double)s.n * E->cfg.val_ratio);
        j.d_out = (s.cur == s.buf[1]) ? s.buf[0] : s.buf[1];
        if (s.p == 0) {
            if (revealed && pub_open(E, s)) { j.z0 = s.ob[0]; j.z1 = nullptr; }
            else { j.z0 = s.cur; j.z1 = s.peer ? s.peer->cur : s.ib[0]; }
            j.labels = s.labels; j.border = s.border; j.counts6 = s.counts; j.loss = s.loss;
            s.has_metrics = true;
        }
        jobs.push_back(j);
    }
    BE(cognn_softmax_jobs_u64(E->ctx, jobs.data(), (int32_t)jobs.size(), L));
    for (auto& s : E->sides) s.cur = (s.cur == s.buf[1]) ? s.buf[0] : s.buf[1];   // after every owner has read its peer's z
}

// ---------------------------------------------------------------------------------------------
// message passing: Scatter + PreMerge + Gather fused into two CSR launches over the share table
// ---------------------------------------------------------------------------------------------
u64* table_seg(cognn_engine* E, Side& s, int F) {
    const int64_t off = s.p == 0 ? E->A_off[s.owner] : E->B_off[s.owner];
    return E->table + off * F;
}

// the cross-rank part of a message-passing round on the share table T: both rounds are enqueued and left in flight
void mp_exchange(cognn_engine* E, int F, u64* T) {
    // replicate the co-party's fresh share of every owner to the other ranks (ss_...h:997-1002 / :982)
    if (E->world > 1) {
        XList xl;
        for (int o = 0; o < E->k; ++o) {
            const int rc = E->holder(o, 1);
            const int64_t bytes = (int64_t)E->G.party[o].localVertexPos.size() * F * 8;
            u64* seg = T + E->B_off[o] * F;
            if (rc == E->rank) {
                for (int r = 0; r < E->world; ++r) {
                    if (r == E->rank) continue;
                    if (E->m == 1 && r == E->rank_of(o)) continue;   // that rank hosts only the owner itself
                    xl.send(r, seg, bytes);
                }
            } else if (!(E->m == 1 && E->rank == E->rank_of(o))) {
                xl.recv(rc, seg, bytes);
            }
            if (E->cfg.placement == COGNN_PLACE_VERTEX_SET) {   // ... and the own share likewise (see build_layout)
                u64* sega = T + E->A_off[o] * F;
                if (rc == E->rank) { for (int r = 0; r < E->world; ++r) if (r != E->rank) xl.send(r, sega, bytes); }
                else xl.recv(rc, sega, bytes);
            }
        }
        run_exchange(E, xl);                                // in flight during the partial-sum launch below (it reads own-share rows only)
    }
    // partial sums of every hosted party for its remote destinations (ss_...h:827-835, 1063-1067)
    if (E->partRows > 0) {
        if (E->timing) BE(cognn_timer_begin(E->ctx, T_PART));
        BE(cognn_gather_csr_u64(E->ctx, T + E->inboxLocalOff * F, nullptr, T, E->part_rowptr, E->part_col, E->partRows, F));
        if (E->timing) {
            BE(cognn_timer_end(E->ctx, T_PART));
            E->algo[T_PART] += 8.0 * F * ((double)E->partEdges + E->partRows) + 4.0 * E->partEdges + 4.0 * (E->partRows + 1);
        }
    }
    if (E->world > 1) {
        XList xl;
        for (auto& sg : E->segs) {
            if (sg.src_rank == E->rank && sg.dst_rank != E->rank) xl.send(sg.dst_rank, T + sg.out_off * F, sg.rows * F * 8);
            if (sg.dst_rank == E->rank && sg.src_rank != E->rank) xl.recv(sg.src_rank, T + sg.inbox_off * F, sg.rows * F * 8);
        }
        run_exchange(E, xl, true);                          // the replication round may still be in flight: both travel while the
    }                                                       // local part of the aggregate runs
}

void message_passing(cognn_engine* E, int F, int64_t it, bool open_scale) {
    mp_exchange(E, F, E->table);
    // aggregate: out = self + local in-edges + replica in-edges (owner rows) / + received partials (co rows).  With several
    // ranks it is split: the entries that read rows held on this rank run now, beside the two exchange rounds; the entries
    // that read received rows (co-share replicas, partial-sum inbox) are added in place once the messages have arrived.
    std::vector<int64_t> sb, se;
    std::vector<u64> sk;
    if (open_scale) {
        // the row scale that follows needs E_p = V_p - a_p: emit it from the (last) gather epilogue instead of V_p
        for (auto& s : E->sides) {
            if (paired(E, s)) continue;                       // a co-located pair hands the plain result to its chain
            const int64_t off = s.p == 0 ? E->A_off[s.owner] : E->B_off[s.owner];
            cognn_keys k = keys(E, s.owner, it, COGNN_OP_GA_SCALE);
            sb.push_back(off); se.push_back(off + s.n); sk.push_back(k.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1]);
        }
    }
    auto aggregate = [&](const u64* base, const uint32_t* rowptr, const uint32_t* col, int64_t edges, bool last) {
        if (E->timing) BE(cognn_timer_begin(E->ctx, E->agg_timer));
        if (last && !sb.empty())
            BE(cognn_gather_csr_open_u64(E->ctx, E->aggOut, base, E->table, rowptr, col, E->aggRows, F, (int32_t)sb.size(), sb.data(), se.data(), sk.data()));
        else
            BE(cognn_gather_csr_u64(E->ctx, E->aggOut, base, E->table, rowptr, col, E->aggRows, F));
        if (E->timing) {
            BE(cognn_timer_end(E->ctx, E->agg_timer));
            E->algo[E->agg_timer] += 8.0 * F * ((double)edges + 2.0 * E->aggRows) + 4.0 * edges + 4.0 * (E->aggRows + 1);
        }
    };
    const bool split = E->world > 1 && E->remEdges > 0;
    aggregate(E->table, E->agg_rowptr, E->agg_col, E->aggEdges, !split);
    exchange_wait(E);
    if (split) aggregate(E->aggOut, E->rem_rowptr, E->rem_col, E->remEdges, true);
    for (auto& s : E->sides) {
        const int64_t off = s.p == 0 ? E->A_off[s.owner] : E->B_off[s.owner];
        s.cur = E->aggOut + off * F;
        s.curF = F;
    }
}

// Single process, every pair co-located: the aggregate launch carries GatherComp's scale (+ the ReLU of ApplyComp) as its
// epilogue (cognn_gather_pair_chain_u64): the lanes that aggregate vertex r's owner-side row also aggregate its co-party-side
// row and run the pair chain on the two sums in registers, so the aggregate itself is never written or re-read.
bool can_fuse_gather_chain(const cognn_engine* E, int F) {
    // (several ranks: the vertex-set placement - the local part of the aggregate runs while the messages travel, the launch over the
    // received rows carries the epilogue)
    if (!E->pair_fusion || E->hosted.size() > 8) return false;
    for (auto& s : E->sides) if (!s.peer) return false;
    return true;
}
void message_passing_fused(cognn_engine* E, int F, int64_t it, bool scale, bool relu_follows, const OpenNext& open_next, bool out_read,
                           bool softmax_follows = false, const u64* table = nullptr) {
    if (!table) table = E->table;
    // scale: GatherComp's post-gather scale follows (every Gather but the last of an epoch, gcn.h:470); out_read: somebody reads
    // the result itself, not only its opening (the weight-gradient product reads the opening alone); softmax_follows: ApplyComp is
    // the prediction layer (softmax_stage) and runs as this launch's second epilogue - the logits are not written
    std::vector<cognn_gather_pair> gp;
    std::vector<cognn_softmax_job> sj;
    sj.reserve(E->sides.size());                             // (the pairs point into it)
    double out_bytes = 0;
    for (auto& s : E->sides) {
        if (s.p != 0) continue;
        Side& t = *s.peer;
        cognn_gather_pair g;
        memset(&g, 0, sizeof(g));
        g.a_row0 = E->A_off[s.owner]; g.b_row0 = E->B_off[s.owner];
        cognn_pair_chain& c = g.chain;
        c.rows = s.n; c.F = F;
        if (scale) {
            c.flags = COGNN_PC_SCALE;
            c.scale[0] = s.svec; c.scale[1] = t.svec;
            c.scale_keys = keys(E, s.owner, it, COGNN_OP_GA_SCALE);
            c.scale_trunc_keys = keys(E, s.owner, it, COGNN_OP_GA_SCALE_TRUNC);
        }
        if (relu_follows) {
            c.flags |= COGNN_PC_RELU;
            c.relu_keys = keys(E, s.owner, it, COGNN_OP_AP_RELU);
            c.mask = s.relu_mask;
            cognn_keys nk = keys(E, s.owner, it + 1, COGNN_OP_PS_GEMM);
            c.out[0] = s.h1; c.out[1] = t.h1;
            c.open[0] = s.h1E; c.flags |= COGNN_PC_OPEN_SUM;      // E_0 + E_1 once, into the owner side's buffer (pair_opening)
                E->h1e_pairs_summed = true;
            c.open_key[0] = nk.k[COGNN_SL_A0]; c.open_key[1] = nk.k[COGNN_SL_A1]; c.flags |= COGNN_PC_OPEN_LIMB;   // (a product's A masks: limb form)
            if (E->forward_only) { c.out[0] = c.out[1] = nullptr; c.mask = nullptr; }
        } else if (softmax_follows) {
            for (Side* x : {&s, &t}) {
                cognn_softmax_job j;
                memset(&j, 0, sizeof(j));
                j.keys = keys(E, s.owner, it, COGNN_OP_AP_SOFTMAX);
                j.p = x->p; j.rows = x->n;
                j.train_rows = (int64_t)((double)x->n * E->cfg.train_ratio);      // gcn.h:560
                j.val_rows = (int64_t)((double)x->n * E->cfg.val_ratio);
                j.d_out = x->buf[0];
                if (x->p == 0) { j.labels = x->labels; j.border = x->border; j.counts6 = x->counts; j.loss = x->loss; x->has_metrics = true; }
                sj.push_back(j);
                g.softmax[x->p] = &sj.back();
            }
        } else {
            if (out_read || !open_next) { c.out[0] = s.buf[1]; c.out[1] = t.buf[1]; }
            if (open_next) {
                c.open[0] = s.ob[open_next.ob]; c.flags |= COGNN_PC_OPEN_SUM;
                c.open_key[0] = open_next.key(s); c.open_key[1] = open_next.key(t);
            }
        }
        const double elems = (double)s.n * F;
        out_bytes += 8.0 * elems * ((c.out[0] ? 2 : 0) + (c.open[0] ? ((c.flags & COGNN_PC_OPEN_SUM) ? 1 : 2) : 0)) + (c.mask ? elems : 0.0);
        if (softmax_follows) out_bytes += 8.0 * elems * 2 + 4.0 * (double)s.n;      // both sides' d_out, the labels
        if (!softmax_follows) attach_dealt(E, c, s.owner, it, DEAL_GATHER_CHAIN);
        out_bytes += 8.0 * elems * dealt_slots_read(E, c);   // the dealt values it reads
        gp.push_back(g);
    }
    if (streams_on(E)) { bool all = true; for (auto& g : gp) all = all && g.chain.dealt; if (!all) for (auto& g : gp) g.chain.dealt = nullptr; }
    if (E->world > 1) {
        // several ranks: both exchange rounds travel while the entries that read rows held here are aggregated (plain launch into aggOut);
        // the launch over the received rows (co-share replicas, partial-sum inbox) then starts from those sums and carries the epilogue
        mp_exchange(E, F, const_cast<u64*>(table));
        if (E->timing) BE(cognn_timer_begin(E->ctx, E->agg_timer));
        BE(cognn_gather_csr_u64(E->ctx, E->aggOut, table, table, E->agg_rowptr, E->agg_col, E->aggRows, F));
        if (E->timing) {
            BE(cognn_timer_end(E->ctx, E->agg_timer));
            E->algo[E->agg_timer] += 8.0 * F * ((double)E->aggEdges + 2.0 * E->aggRows) + 4.0 * E->aggEdges + 4.0 * (E->aggRows + 1);
        }
        exchange_wait(E);
    }
    const bool split = E->world > 1;
    const double edges = split ? (double)E->remEdges : (double)E->aggEdges;
    if (E->timing) BE(cognn_timer_begin(E->ctx, E->agg_timer));
    BE(cognn_gather_pair_chain_base_u64(E->ctx, table, split ? E->aggOut : nullptr, split ? E->rem_rowptr : E->agg_rowptr, split ? E->rem_col : E->agg_col, F,
                                        gp.data(), (int32_t)gp.size()));
    if (E->timing) {
        BE(cognn_timer_end(E->ctx, E->agg_timer));
        // source row per entry, base row per output row, u32 col / rowptr + what the epilogue writes
        E->algo[E->agg_timer] += 8.0 * F * (edges + (double)E->aggRows) + 4.0 * edges + 4.0 * (E->aggRows + 1) + out_bytes;
    }
    for (auto& s : E->sides) {
        s.cur = relu_follows ? s.h1 : softmax_follows ? s.buf[0] : s.buf[1];
        s.curF = F;
    }
    if (relu_follows) E->gemm_x_opened_for = it + 1;
}

// ---------------------------------------------------------------------------------------------
// weight averaging (gcn.h:747-802)
// ---------------------------------------------------------------------------------------------
void weight_average(cognn_engine* E, int64_t it, int layer) {
    // Party 1 sums the owner shares of parties >= 1 plus its co-share of party 0's weights, party 0 sums its own
    // share plus every other co-share (gcn.h:753-762); both scale by 1/k (training variant only, :763-764) and
    // the results are redistributed (:765-778).  Each rank pre-sums its local contributions, so the exchange is
    // one small message per rank towards each holder and one back.
    const int k = E->k;
    const int64_t elems = layer == 0 ? (int64_t)E->in() * E->hid() : (int64_t)E->hid() * E->lab();
    const size_t bytes = (size_t)elems * 8;
    const int r0 = E->rank_of(0), r1 = E->rank_of(1);
    u64* part[2] = {E->wa[0], E->wa[1]};
    {
        std::vector<const uint64_t*> in[2];
        for (auto& s : E->sides) in[(s.owner == 0) ? s.p : 1 - s.p].push_back(s.W[layer]);   // (0,0)->sum0 (0,1)->sum1 ; (o,1)->sum0 (o,0)->sum1 for o>=1
        const bool one_each = in[0].size() <= 15 && in[1].size() <= 15;   // then the two sums are independent launches: one batch
        std::unique_ptr<Batch> batch(one_each ? new Batch(E) : nullptr);
        for (int h = 0; h < 2; ++h) {
            if (in[h].empty()) { BE(cognn_memset0(E->ctx, part[h], bytes)); continue; }
            for (size_t b = 0; b < in[h].size(); b += 15) {             // 16 inputs per launch; later launches carry the running sum
                std::vector<const uint64_t*> v;
                if (b) v.push_back(part[h]);
                v.insert(v.end(), in[h].begin() + b, in[h].begin() + std::min(in[h].size(), b + 15));
                BE(cognn_sum_u64(E->ctx, part[h], v.data(), (int32_t)v.size(), elems));
            }
        }
    }
    const int holder[2] = {r0, r1};
    {
        XList xl;
        for (int h = 0; h < 2; ++h) {
            if (E->rank != holder[h]) xl.send(holder[h], part[h], (int64_t)bytes);
            else
                for (int r = 0; r < E->world; ++r)
                    if (r != E->rank) xl.recv(r, E->wa_recv[h] + (size_t)r * E->wa_stride, (int64_t)bytes);
        }
        run_exchange_sync(E, xl);
    }
    for (int h = 0; h < 2; ++h)
        if (E->rank == holder[h])
            for (int r = 0; r < E->world; ++r)
                if (r != E->rank) BE(cognn_add_u64(E->ctx, part[h], part[h], E->wa_recv[h] + (size_t)r * E->wa_stride, elems));
    if (E->cfg.variant != COGNN_VARIANT_OPTIMIZE_GCN_INFERENCE) {   // twoPartyGCNMatrixScale between parties 0 and 1
        cognn_keys tk = keys(E, COGNN_OWNER_WAVG, it, COGNN_OP_WAVG_TRUNC);
        const u64 ws = fx_trunc(1.0 / k);
        u64* c0 = E->wa[2];
        u64* c1 = E->wa[3];
        if (E->rank == r0) BE(cognn_trunc_open_u64(E->ctx, c0, part[0], ws, &tk, 0, elems));
        if (E->rank == r1) BE(cognn_trunc_open_u64(E->ctx, c1, part[1], ws, &tk, 1, elems));
        if (r0 != r1) {
            XList xl;
            if (E->rank == r1) xl.send(r0, c1, (int64_t)bytes);
            if (E->rank == r0) xl.recv(r1, c1, (int64_t)bytes);
            run_exchange_sync(E, xl);
        }
        if (E->rank == r0) BE(cognn_trunc_close_u64(E->ctx, part[0], c0, c1, &tk, 0, 0, elems));
        if (E->rank == r1) BE(cognn_trunc_close_u64(E->ctx, part[1], nullptr, nullptr, &tk, 1, 0, elems));
    }
    u64* avg[2] = {part[0], part[1]};                      // share 0 / share 1 of the averaged weights
    {
        XList xl;
        for (int h = 0; h < 2; ++h) {
            if (E->rank == holder[h]) {
                for (int r = 0; r < E->world; ++r) if (r != E->rank) xl.send(r, part[h], (int64_t)bytes);
            } else {
                avg[h] = E->wa[4 + h];
                xl.recv(holder[h], avg[h], (int64_t)bytes);
            }
        }
        run_exchange_sync(E, xl);
    }
    {                                                      // owner 0 keeps (s0, s1); owners >= 1 keep (s1, s0)
        std::vector<uint64_t*> out[2];
        for (auto& s : E->sides) out[(s.owner == 0) ? s.p : 1 - s.p].push_back(s.W[layer]);
        Batch batch(E);
        for (int h = 0; h < 2; ++h)
            for (size_t b = 0; b < out[h].size(); b += 16)
                BE(cognn_fanout_u64(E->ctx, out[h].data() + b, (int32_t)std::min<size_t>(16, out[h].size() - b), avg[h], elems));
    }
}

// ---------------------------------------------------------------------------------------------
// schedule (Appendix A of SURVEY.md; gcn.h:893-948)
// ---------------------------------------------------------------------------------------------
struct IterInfo {
    int e, f, ep, layer;
    bool fwd, apply_only;
};
IterInfo iter_info(cognn_engine* E, int64_t it) {
    IterInfo r;
    r.f = E->cfg.num_layers;
    r.ep = 3 * E->cfg.num_layers;
    r.e = (int)(it % r.ep);
    r.fwd = r.e < r.f;
    r.layer = r.fwd ? r.e : r.f - 1 - ((r.e - r.f) / 2);
    r.apply_only = (r.e != 0 && r.e % r.f == 0);          // ss_...h:709, 941
    return r;
}
int mp_width(cognn_engine* E, int e) {                    // getPlainNumPerOperand(iter), gcn.h:898-927
    switch (e) { case 0: return E->hid(); case 1: case 2: case 3: return E->lab(); default: return E->hid(); }
}

GemmSpec prescatter_spec(cognn_engine* E, Side& s, int layer) {
    GemmSpec g{s.n, layer == 0 ? E->hid() : E->lab(), layer == 0 ? E->in() : E->hid(), 0, COGNN_OP_PS_GEMM, COGNN_OP_PS_GEMM_TRUNC};
    g.feature = (layer == 0) ? 1 : 0;
    return g;
}

// d = h_t^T . in (gcn.h:671,710); for layer 0 h_t is the transposed feature tensor: mask and opening of the forward product
// for layer 1 h_t is the transposed hidden activation whose opening the layer-1 forward product left in h1E two GAS
// iterations earlier (same epoch): mask and opening are reused too
GemmSpec wgrad_spec(cognn_engine* E, Side& s, int layer, int64_t it) {
    GemmSpec g{layer == 0 ? E->in() : E->hid(), layer == 0 ? E->hid() : E->lab(), s.n, 2, COGNN_OP_AP_GEMM, COGNN_OP_AP_GEMM_TRUNC};
    if (layer == 0) g.feature = 2;
    else { g.xsrc = X_H1E_REUSE; g.akey_it = it - (it % (3 * E->cfg.num_layers)) + layer; }   // the forward iteration of that layer
    return g;
}

// pairs_fused: the co-located pairs' products are still in zbuf (gemm_stage, pairs_raw): product truncation, both scales and the
// update run as one pass per pair (cognn_pair_weight_update_u64) - and, when every party's pair is hosted here, the weight
// average too (returns true: weight_average has been done)
bool weight_update_chain(cognn_engine* E, int64_t it, int layer, bool pairs_fused, bool raw,
                         const std::function<GemmSpec(Side&)>& specfn = nullptr) {
    auto wspec = [&](Side& s) { return specfn ? specfn(s) : wgrad_spec(E, s, layer, it); };   // the product whose result is being consumed
    // d (in side.small[0]) -> *1/trainSetSize -> W -= lr*d   (gcn.h:673-678, 720-730)
    const int64_t elems = layer == 0 ? (int64_t)E->in() * E->hid() : (int64_t)E->hid() * E->lab();
    const u64 lr = fx_trunc(E->cfg.learning_rate);
    const bool inference = E->cfg.variant == COGNN_VARIANT_OPTIMIZE_GCN_INFERENCE;
    auto gscale = [&](Side& s) {                           // gradient scale: per-owner constant
        const int64_t train = (int64_t)((double)s.n * E->cfg.train_ratio);
        return train > 0 ? fx_trunc(1.0 / (double)train) : (u64)0;
    };
    bool averaged = false;
    if (pairs_fused) {
        std::vector<cognn_pair_wupdate> jobs;
        bool all = true;
        for (auto& s : E->sides) {
            if (!paired(E, s)) { all = false; continue; }
            if (s.p != 0) continue;
            Side& t = *s.peer;
            cognn_pair_wupdate J;
            memset(&J, 0, sizeof(J));
            GemmSpec g = wspec(s);
            J.z[0] = s.zbuf; J.z[1] = t.zbuf; J.W[0] = s.W[layer]; J.W[1] = t.W[layer];
            if (raw) J.c1 = t.c1.at({it, g.op}).ptr;
            J.gemm_keys = gemm_keys(E, s, it, g);
            J.trunc_keys[0] = keys(E, s.owner, it, g.top);
            J.trunc_keys[1] = keys(E, s.owner, it, COGNN_OP_AP_GSCALE_TRUNC);
            J.trunc_keys[2] = keys(E, s.owner, it, COGNN_OP_AP_LR_TRUNC);
            J.trunc_keys[3] = keys(E, s.owner, it, COGNN_OP_WAVG_TRUNC);
            J.mul[0] = gscale(s); J.mul[1] = lr; J.mul[2] = inference ? fx_trunc(1.0 / E->k) : 0;   // optimize-gcn-inference/gcn.h:680-681,732-733
            J.n = elems;
            J.flags = (raw ? 0 : COGNN_PC_NO_C) | (s.owner == 0 ? 0 : COGNN_WU_SWAP);   // owner 0 keeps (s0, s1); owners >= 1 keep (s1, s0)
            if (z_clear_wanted(s, elems) && z_clear_wanted(t, elems)) { J.flags |= COGNN_WU_CLEAR_Z; z_cleared(s, elems); z_cleared(t, elems); }
            jobs.push_back(J);
        }
        averaged = all && E->world == 1 && jobs.size() <= 16 && elems > 0;
        cognn_keys ak = keys(E, COGNN_OWNER_WAVG, it, COGNN_OP_WAVG_TRUNC);
        const u64 amul = E->cfg.variant != COGNN_VARIANT_OPTIMIZE_GCN_INFERENCE ? fx_trunc(1.0 / E->k) : 0;   // twoPartyGCNMatrixScale between parties 0 and 1 (gcn.h:763-764)
        BE(cognn_pair_weight_update_u64(E->ctx, jobs.data(), (int32_t)jobs.size(), &ak, averaged ? amul : 0, averaged ? 1 : 0));
        for (auto& s : E->sides)
            if (paired(E, s) && s.p == 1) c1_release(E, s, {it, wspec(s).op});
        if (all) return averaged;
    }
    std::vector<u64*> d, d2;
    std::vector<int64_t> el;
    for (auto& s : E->sides) { d.push_back(s.small[0]); d2.push_back(s.small[1]); el.push_back(elems); }
    for_sides(E, true, [&](Side& s, size_t i) {
        cognn_keys k = keys(E, s.owner, it, COGNN_OP_AP_GSCALE_TRUNC);
        BE(cognn_trunc_open_u64(E->ctx, s.ob[2], d[i], gscale(s), &k, s.p, elems));
    }, pairs_fused);
    exchange_ob(E, 2, el);
    for_sides(E, true, [&](Side& s, size_t i) {
        cognn_keys k = keys(E, s.owner, it, COGNN_OP_AP_GSCALE_TRUNC);
        BE(cognn_trunc_close_u64(E->ctx, d2[i], s.p == 0 ? s.ob[2] : nullptr, s.p == 0 ? s.ib[2] : nullptr, &k, s.p, 0, elems));
    }, pairs_fused);
    trunc_stage(E, it, COGNN_OP_AP_LR_TRUNC, lr, d2, el, [&](Side& s) { return s.W[layer]; }, 1, ~0ull, pairs_fused);
    if (inference) {                                       // optimize-gcn-inference/gcn.h:680-681,732-733
        std::vector<u64*> w;
        for (auto& s : E->sides) w.push_back(s.W[layer]);
        trunc_stage(E, it, COGNN_OP_WAVG_TRUNC, fx_trunc(1.0 / E->k), w, el, [&](Side& s) { return s.W[layer]; }, 0, ~0ull, pairs_fused);
    }
    return false;
}

void run_iteration_original(cognn_engine* E, int64_t it);
void run_iteration(cognn_engine* E, int64_t it) {
    if (original(E)) { run_iteration_original(E, it); return; }
    const IterInfo I = iter_info(E, it);
    if (E->forward_only && !I.fwd) throw EngineError("engine: COGNN_OPT_FORWARD_ONLY is set but a backward iteration was requested");
    bool relu_opened = false, wgrad_w_opened = false, relu_pairs_done = false, gather_chain_fused = false, z_revealed = false, softmax_done = false, prescaled = false;
    set_salt(E, it);                                       // (a launch only when the epoch changes: never inside a recorded epoch)
    if (I.e == 0) {                                        // ss_...h:695, 938: back to the input features
        for (auto& s : E->sides) { s.cur = s.feat; s.curF = E->in(); s.cur_mask = nullptr; }
        if (E->graph_epochs) open_features(E);             // this epoch's feature mask (feature_gemm_keys)
    }
    // a deferred ReLU' selection is consumed by the backward PreScatter row scale of the co-located pairs; anybody else gets the
    // selected tensor
    if (!(!I.apply_only && !I.fwd && E->pair_fusion))
        for (auto& s : E->sides) apply_cur_mask(E, s);
    if (!I.apply_only) {
        const int F = mp_width(E, I.e);
        // ---- PreScatterComp (gcn.h:198-255) ----
        Phase ph_ps(E, T_PH_PRESCATTER);
        if (I.fwd) {
            bool x_opened = (I.layer == 1 && E->gemm_x_opened_for == it);   // H already sits in h_t[1], its opening in h1E
            if (I.layer == 1 && !x_opened) {               // (not reached in a normal run: the ReLU close of iteration it-1 does both)
                E->h1e_pairs_summed = false;               // every side writes its own share of the opening here
                for (auto& s : E->sides) {
                    BE(cognn_memcpy_d2d(E->ctx, s.h1, s.cur, (size_t)s.n * E->hid() * 8));   // h_t[1]
                    cognn_keys k = keys(E, s.owner, it, COGNN_OP_PS_GEMM);
                    BE(cognn_mask_open_u64(E->ctx, s.h1E, s.cur, k.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], s.n, E->hid(), COGNN_MASK_OPEN_LIMB));
                }
                x_opened = true;
            }
            const bool scale_follows = I.e != 0;
            // the truncation close of the product also opens the row scale that consumes it
            OpenNext open_scale([&](Side& s, int p) { return keys(E, s.owner, it, COGNN_OP_PS_SCALE).k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1]; });
            FollowScale follow;
            if (scale_follows) { follow.op = COGNN_OP_PS_SCALE; follow.top = COGNN_OP_PS_SCALE_TRUNC; follow.dst = [&](Side& s) { return table_seg(E, s, F); }; }
            gemm_stage(E, it, [&](Side& s) { return s.cur; }, [&](Side& s) { return s.W[I.layer]; },
                       [&](Side& s) { return prescatter_spec(E, s, I.layer); },
                       [&](Side& s) { return scale_follows ? s.buf[1] : table_seg(E, s, F); }, x_opened,
                       scale_follows ? open_scale : OpenNext(), false, follow);
            if (scale_follows)
                rowscale_stage(E, it, COGNN_OP_PS_SCALE, COGNN_OP_PS_SCALE_TRUNC, F, [&](Side& s) { return s.buf[1]; },
                               [&](Side& s) { return table_seg(E, s, F); }, E_IN_OB0, OpenNext(), true);
        } else if (E->prescaled_it == it) {                 // done by the chain of iteration it - 3 (see there): the result sits in table2
            for (auto& s : E->sides) s.cur_mask = nullptr;
            prescaled = true;
        } else {
            rowscale_stage(E, it, COGNN_OP_PS_SCALE, COGNN_OP_PS_SCALE_TRUNC, F, [&](Side& s) { return s.cur; },
                           [&](Side& s) { return table_seg(E, s, F); });
        }
        ph_ps.end();
        // ---- Scatter / PreMerge / Gather ----
        const bool gscale = (it + 1) % I.ep != 0;          // gcn.h:470
        const bool fuse_open = gscale && E->sides.size() <= 32;
        E->agg_timer = (I.e == I.f - 1 || I.e == I.f + 1) ? T_AGG_LAB : T_AGG;   // the label-wide rounds of an epoch (widths {hid, lab, -, lab, -, hid})
        relu_opened = false;
        if (can_fuse_gather_chain(E, F)) {                 // (no scale <=> last iteration of an epoch, a backward one)
            // the scale (and ReLU) of the co-located pairs rides in the aggregate launch's epilogue; in a backward iteration
            // the aggregate's only reader is the weight-gradient product, which takes it as an opening
            relu_opened = gscale && I.fwd && I.e != I.f - 1;
            wgrad_w_opened = !I.fwd;
            OpenNext open_wgrad([&](Side& s, int p) { return gemm_keys(E, s, it, wgrad_spec(E, s, I.layer, it)).k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1]; }, 1);
            // ... and in the last forward iteration ApplyComp's prediction layer rides along as well
            softmax_done = I.fwd && I.e == I.f - 1 && E->softmax_fusion && !streams_on(E) && E->be->cognn_gather_pair_chain_takes_softmax(F) != 0;
            Phase ph_mp(E, T_PH_MP);
            message_passing_fused(E, F, it, gscale, relu_opened, wgrad_w_opened ? open_wgrad : OpenNext(), !wgrad_w_opened, softmax_done,
                                  prescaled ? E->table2 : E->table);
            if (prescaled) E->prescaled_it = -1;
            relu_pairs_done = relu_opened;
            gather_chain_fused = true;
        } else {
            Phase ph_mp(E, T_PH_MP);
            message_passing(E, F, it, fuse_open);
        }
        Phase ph_ga(E, T_PH_GATHER);
        if (gscale && !gather_chain_fused) {
            // a hidden forward layer feeds the ReLU next: the close of this scale already opens it
            relu_opened = I.fwd && I.e != I.f - 1;
            // ... and in a backward iteration the weight-gradient product d = h_t^T . in is next: its right operand is this result
            wgrad_w_opened = !I.fwd;
            OpenNext open_relu([&](Side& s, int p) { return keys(E, s.owner, it, COGNN_OP_AP_RELU).k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1]; });
            OpenNext open_wgrad([&](Side& s, int p) { return gemm_keys(E, s, it, wgrad_spec(E, s, I.layer, it)).k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1]; }, 1);
            rowscale_stage(E, it, COGNN_OP_GA_SCALE, COGNN_OP_GA_SCALE_TRUNC, F, [&](Side& s) { return s.cur; },
                           [&](Side& s) { return s.buf[1]; }, fuse_open ? E_IN_X : E_FROM_X,
                           relu_opened ? open_relu : wgrad_w_opened ? open_wgrad : I.fwd ? OpenNext::Reveal() : OpenNext(), false, relu_opened);
            z_revealed = I.fwd && !relu_opened;
            relu_pairs_done = relu_opened;
            for (auto& s : E->sides) if (!(relu_pairs_done && paired(E, s))) s.cur = s.buf[1];
        }
    }
    // ---- ApplyComp (gcn.h:515-811) ----
    Phase ph_ap(E, T_PH_APPLY);
    if (I.fwd) {
        if (I.e != I.f - 1) relu_stage(E, it, relu_opened, relu_pairs_done);
        else if (!softmax_done) softmax_stage(E, it, z_revealed);
        for (auto& s : E->sides) s.curF = (I.e != I.f - 1) ? E->hid() : E->lab();
        return;
    }
    const bool first_of_two = ((I.e - I.f) % 2 == 0);
    if (first_of_two) {
        if (I.layer == I.f - 1) {                          // g = (p-y) . W1^T, out = in  (gcn.h:664-669)
            // (W1 is read across by the opening of the right operand: no transposed copy)
            // g's only reader is the PreScatter scale of iteration it + 3 (after the ReLU' of it + 2): when that iteration runs inside this
            // very cognn_engine_run call, the chain that truncates the product applies both and writes that iteration's share table
            // (table2: the table itself is used by iteration it + 1) - same dealer streams, same values, g never goes to memory
            FollowScale follow;
            if (E->backward_fusion && I.f == 2 && it + 3 < E->run_end && can_fuse_gather_chain(E, E->hid()) && !streams_on(E) && !E->graph_epochs &&
                !E->cfg.verbose) {
                if (!E->table2) E->table2 = dalloc<u64>(E, (size_t)E->tableRows * E->hid());
                follow.op = COGNN_OP_PS_SCALE; follow.top = COGNN_OP_PS_SCALE_TRUNC; follow.it = it + 3;
                follow.dst = [&](Side& s) { return E->table2 + (s.p == 0 ? E->A_off[s.owner] : E->B_off[s.owner]) * E->hid(); };
                follow.mask = [&](Side& s) { return (const uint8_t*)s.relu_mask; };
                E->prescaled_it = it + 3;
            }
            gemm_stage(E, it, [&](Side& s) { return s.cur; }, [&](Side& s) { return s.W[1]; },
                       [&](Side& s) { GemmSpec g{s.n, E->hid(), E->lab(), 0, COGNN_OP_AP_GEMM, COGNN_OP_AP_GEMM_TRUNC}; g.transB = 1; return g; },
                       [&](Side& s) { return s.g; }, false, OpenNext(), false, follow);
        } else {                                           // out = in * 1[z>0]  (gcn.h:702-708; g' skipped for layer 0)
            Batch batch(E);
            for (auto& s : E->sides) {
                const uint8_t* mask = (paired(E, s) && s.p == 1) ? s.peer->relu_mask : s.relu_mask;   // a pair chain writes one (public) mask
                if (paired(E, s)) { s.cur_mask = mask; continue; }    // deferred: the next iteration's row-scale chain selects while it reads
                u64* dstb = (s.cur == s.buf[1]) ? s.buf[0] : s.buf[1];
                BE(cognn_mask_select_u64(E->ctx, dstb, s.cur, mask, (int64_t)s.n * E->hid()));
                s.cur = dstb;
            }
        }
        return;
    }
    // d = h_t^T . in ; scale ; W -= lr d ; out = g  (gcn.h:671-684, 710-736)
    const bool pairs_fused = E->pair_fusion && E->wupdate_fusion && !streams_on(E);
    const bool raw = gemm_stage(E, it, [&](Side& s) { return I.layer == 0 ? s.feat : s.h1; }, [&](Side& s) { return s.cur; },
                                [&](Side& s) { return wgrad_spec(E, s, I.layer, it); },
                                [&](Side& s) { return s.small[0]; }, false, OpenNext(), wgrad_w_opened, FollowScale(), pairs_fused);
    const bool averaged = weight_update_chain(E, it, I.layer, pairs_fused, raw);
    for (auto& s : E->sides) {
        if (I.layer == I.f - 1) { s.cur = s.g; s.curF = E->hid(); }
        else { s.curF = 0; }                               // vertexInterData["g"] is empty for the first layer
    }
    ph_ap.end();
    Phase ph_wa(E, T_PH_WAVG);
    if (!averaged) weight_average(E, it, I.layer);
    exchange_wait(E);
}

// ---------------------------------------------------------------------------------------------
// original-gcn (algo_kernels/vertex_centric/original-gcn/gcn.h; BASELINE config 1) - single process
// ---------------------------------------------------------------------------------------------
// Index of the per-edge Scatter: the instance (client P, destination party g) lists P's edges into g ordered by destination vid,
// then source vid (updateSrcVertexPos[g] / updateDstVertexPos[g], ss_...h:467-504) - for g == P with one dummy self entry for every
// vertex without a local in-edge (ss_...h:411-418), which occupies a position of the list (its dealer streams are never drawn) and
// contributes nothing (isGatherDstVertexDummy).  Normalisers per edge: n0 = (outDeg_src + 1)^-1/2 from the client; n1 =
// (inDeg_dst + 1)^-1/2 from the client for its local edges, from the server (the destination party) otherwise (gcn.h:228-229,
// ss_...h:800,1041-1043); degrees after the dummy inflation, 0 -> 0 (gcn.h:219-221).
void build_original_index(cognn_engine* E) {
    const int k = E->k;
    auto& G = E->G;
    auto norm = [](uint32_t deg) { return deg == 0 ? (u64)0 : fx_llround(std::pow((double)deg + 1.0, -0.5)); };
    E->orig_dst.assign((size_t)k, cognn_engine::OrigDst());
    E->orig_pair.assign((size_t)k * k, cognn_engine::OrigPair());
    std::vector<std::vector<std::vector<uint32_t>>> rows_src((size_t)k), rows_pair((size_t)k), rows_q((size_t)k);
    for (int g = 0; g < k; ++g) {
        const size_t n = G.party[g].localVertexPos.size();
        rows_src[g].resize(n); rows_pair[g].resize(n); rows_q[g].resize(n);
    }
    for (int P = 0; P < k; ++P)
        for (int g = 0; g < k; ++g) {
            const cognn::EdgeBlock& blk = G.party[P].out[g];
            std::vector<u64> n0, n1;
            for (size_t r = 0; r + 1 < blk.rowptr.size(); ++r) {
                const uint32_t dr = G.row_of_vid[blk.rows_vid[r]];
                if (g == P && blk.rowptr[r + 1] == blk.rowptr[r]) { n0.push_back(0); n1.push_back(0); continue; }   // the dummy self entry's slot
                for (uint32_t e = blk.rowptr[r]; e < blk.rowptr[r + 1]; ++e) {
                    rows_src[g][dr].push_back(blk.col[e]);
                    rows_pair[g][dr].push_back((uint32_t)P);
                    rows_q[g][dr].push_back((uint32_t)n0.size());
                    n0.push_back(norm(G.party[P].outDeg[blk.col[e]]));
                    n1.push_back(norm(G.party[g].inDeg[dr]));
                }
            }
            auto& op = E->orig_pair[(size_t)P * k + g];
            op.edges = (int64_t)n0.size();
            op.n0 = upload(E, n0); op.n1 = upload(E, n1);
        }
    for (int g = 0; g < k; ++g) {
        std::vector<uint32_t> rp{0}, src, pr, q;
        for (size_t r = 0; r < rows_src[g].size(); ++r) {
            src.insert(src.end(), rows_src[g][r].begin(), rows_src[g][r].end());
            pr.insert(pr.end(), rows_pair[g][r].begin(), rows_pair[g][r].end());
            q.insert(q.end(), rows_q[g][r].begin(), rows_q[g][r].end());
            rp.push_back((uint32_t)src.size());
        }
        auto& od = E->orig_dst[g];
        od.entries = (int64_t)src.size();
        od.rowptr = upload(E, rp); od.src = upload(E, src); od.pair = upload(E, pr); od.q = upload(E, q);
    }
}
u64 scatter_tag(int P, int g) { return 0x10000ull + (u64)P * 256 + (u64)g; }   // dealer "owner" of the Scatter instance (several run per owner and iteration)

// ScatterComp + UpdatePreMergeComp + GatherComp of one GAS iteration: every side's tensor src(side) [n x F] -> dst(side)
template <class SrcFn, class DstFn>
void original_message_passing(cognn_engine* E, int64_t it, int F, bool fwd, SrcFn src, DstFn dst) {
    const int k = E->k;
    for (int g = 0; g < k; ++g) {
        Side* a = E->side(g, 0); Side* b = E->side(g, 1);
        std::vector<cognn_scatter_pair> pairs((size_t)k);
        for (int P = 0; P < k; ++P) {
            cognn_scatter_pair& sp = pairs[(size_t)P];
            memset(&sp, 0, sizeof(sp));
            const auto& op = E->orig_pair[(size_t)P * k + g];
            sp.srcA = src(*E->side(P, 0)); sp.srcB = src(*E->side(P, 1));
            sp.n0 = op.n0; sp.n1 = op.n1;
            const u64 tag = scatter_tag(P, g);
            sp.scale0 = keys(E, tag, it, COGNN_OP_SC_SCALE0); sp.trunc0 = keys(E, tag, it, COGNN_OP_SC_SCALE0_TRUNC);
            sp.scale1 = keys(E, tag, it, COGNN_OP_SC_SCALE1); sp.trunc1 = keys(E, tag, it, COGNN_OP_SC_SCALE1_TRUNC);
            sp.n1_from_server = P == g ? 0 : 1;
            sp.crossed = P == g ? 0 : 1;
        }
        cognn_keys sk = keys(E, (u64)g, it, COGNN_OP_GA_SCALE), tk = keys(E, (u64)g, it, COGNN_OP_GA_SCALE_TRUNC);
        const auto& od = E->orig_dst[(size_t)g];
        if (E->timing) BE(cognn_timer_begin(E->ctx, T_AGG));
        BE(cognn_scatter_gather_original_u64(E->ctx, dst(*a), dst(*b), src(*a), src(*b), fwd ? a->svec : nullptr, fwd ? b->svec : nullptr, &sk, &tk,
                                             (int64_t)a->n, F, od.rowptr, od.src, od.pair, od.q, pairs.data(), k));
        if (E->timing) {
            BE(cognn_timer_end(E->ctx, T_AGG));
            E->algo[T_AGG] += 16.0 * F * ((double)od.entries + 2.0 * a->n);       // both shares of every source row read, of every vertex row read and written
        }
    }
}

void run_iteration_original(cognn_engine* E, int64_t it) {
    const int f = E->cfg.num_layers, ep = 2 * f;
    const int e = (int)(it % ep);
    const bool fwd = e < f;
    const int layer = fwd ? e : f - 1 - (e - f);           // :337-340, 431-434
    const bool apply_only = (e != 0 && e % f == 0);        // ss_...h:709, 941
    const int in = E->in(), hid = E->hid(), lab = E->lab();
    set_salt(E, it);
    if (e == 0)                                            // ss_...h:695, 938: back to the input features
        for (auto& s : E->sides) { s.cur = s.feat; s.curF = in; s.cur_mask = nullptr; }
    if (!apply_only) {                                     // PreScatterComp is a copy (:198-209)
        const int F = e == 0 ? in : hid;                   // getPlainNumPerOperand :807-830 ({in, hid, lab, hid}; e = 2 is apply-only)
        Phase ph_mp(E, T_PH_MP);
        // forward: the aggregate IS ah_t of the layer (:452); backward: into the scratch buffer that is not the input
        auto out = [&](Side& s) { return fwd ? s.ah[layer] : (s.cur == s.buf[1] ? s.buf[0] : s.buf[1]); };
        original_message_passing(E, it, F, fwd, [&](Side& s) { return s.cur; }, out);
        for (auto& s : E->sides) { s.cur = out(s); s.curF = F; }
    }
    Phase ph_ap(E, T_PH_APPLY);
    if (fwd) {                                             // twoPartyGCNForwardNN / ForwardNNPrediction (:459, 493): z = in . W, then ReLU / softmax
        const int K = layer == 0 ? in : hid, N = layer == 0 ? hid : lab;
        gemm_stage(E, it, [&](Side& s) { return s.cur; }, [&](Side& s) { return s.W[layer]; },
                   [&](Side& s) { return GemmSpec{s.n, N, K, 0, COGNN_OP_AP_FWD_GEMM, COGNN_OP_AP_FWD_GEMM_TRUNC}; },
                   [&](Side& s) { return s.buf[0]; });
        for (auto& s : E->sides) { s.cur = s.buf[0]; s.curF = N; }
        if (layer != f - 1) relu_stage(E, it, false, false);
        else softmax_stage(E, it, false);
        return;
    }
    // backward: BackwardNNInit (:586, last layer) / BackwardNN (:622): gz = in (.) 1[z > 0] (not for the last layer), d = ah_t^T . gz,
    // g = gz . W^T with the weights before the update (not for the first layer), then the update and the weight average (:659-711)
    if (layer != f - 1) {
        Batch batch(E);
        for (auto& s : E->sides) {
            const uint8_t* mask = (paired(E, s) && s.p == 1) ? s.peer->relu_mask : s.relu_mask;   // a pair chain writes one (public) mask
            u64* dstb = (s.cur == s.buf[1]) ? s.buf[0] : s.buf[1];
            BE(cognn_mask_select_u64(E->ctx, dstb, s.cur, mask, (int64_t)s.n * hid));
            s.cur = dstb;
        }
    }
    const int M = layer == 0 ? in : hid, N = layer == 0 ? hid : lab;
    if (layer != 0)                                        // first: it reads the weights the update below changes
        gemm_stage(E, it, [&](Side& s) { return s.cur; }, [&](Side& s) { return s.W[layer]; },
                   [&](Side& s) { GemmSpec g{s.n, M, N, 0, COGNN_OP_AP_GEMM, COGNN_OP_AP_GEMM_TRUNC}; g.transB = 1; return g; },
                   [&](Side& s) { return s.g; });
    auto dspec = [&](Side& s) { return GemmSpec{M, N, s.n, 1, COGNN_OP_AP_DGEMM, COGNN_OP_AP_DGEMM_TRUNC}; };
    const bool pairs_fused = E->pair_fusion && E->wupdate_fusion && !streams_on(E);
    const bool raw = gemm_stage(E, it, [&](Side& s) { return s.ah[layer]; }, [&](Side& s) { return s.cur; }, dspec,
                                [&](Side& s) { return s.small[0]; }, false, OpenNext(), false, FollowScale(), pairs_fused);
    const bool averaged = weight_update_chain(E, it, layer, pairs_fused, raw, dspec);
    for (auto& s : E->sides) {
        if (layer != 0) { s.cur = s.g; s.curF = hid; }
        else s.curF = 0;                                   // :620-621: no g for the first layer
    }
    ph_ap.end();
    Phase ph_wa(E, T_PH_WAVG);
    if (!averaged) weight_average(E, it, layer);
    exchange_wait(E);
}

// the Beaver product a side runs in GAS iteration `it` (at most one: PreScatter in forward iterations, Apply in backward ones)
bool gemm_of_iteration(cognn_engine* E, Side& s, int64_t it, GemmSpec& g) {
    const IterInfo I = iter_info(E, it);
    if (!I.apply_only && I.fwd) { g = prescatter_spec(E, s, I.layer); return true; }
    if (!I.fwd && (I.e - I.f) % 2 == 0 && I.layer == I.f - 1) {
        g = GemmSpec{s.n, E->hid(), E->lab(), 0, COGNN_OP_AP_GEMM, COGNN_OP_AP_GEMM_TRUNC};
        return true;
    }
    if (!I.fwd && (I.e - I.f) % 2 == 1) { g = wgrad_spec(E, s, I.layer, it); return true; }
    return false;
}

// dealer phase: product shares of every Beaver GEMM in [it0,it1).  The products of one shape (N, K) - all sides' triples of a protocol
// phase, and the same phase of later iterations - go to the grouped MFMA launch (cognn_dealer_gemm_c1_group_u64: operands
// generated in registers, nothing materialised), up to 16 per launch; the weight-gradient triples (transposed left operand,
// K = #vertices) stay on the per-triple path (a K-split MFMA kernel of its own).
void run_offline(cognn_engine* E, int64_t it0, int64_t it1) {
    // recorded epochs deal their product shares inside the recording (on demand), unless the caller replays ONE epoch and keeps them
    if (E->graph_epochs && !E->retain_offline) return;
    if (original(E)) return;                               // original-gcn: every product share is dealt when its product runs
    struct Pending { std::vector<cognn_dealer_job> jobs; };
    std::map<std::pair<int64_t, int64_t>, Pending> groups;  // (N, K) -> jobs waiting for a launch
    auto flush = [&](std::pair<int64_t, int64_t> nk, Pending& p) {
        if (p.jobs.empty()) return;
        BE(cognn_dealer_gemm_c1_group_u64(E->ctx, p.jobs.data(), (int32_t)p.jobs.size(), nk.first, nk.second));
        p.jobs.clear();
    };
    std::vector<cognn_dealer_tn_job> tn_jobs;
    for (int64_t it = it0; it < it1; ++it) {
        const u64 salt_before = E->salt_now;
        set_salt(E, it);
        if (E->graph_epochs && E->salt_now != salt_before)  // (recorded epochs: the salt lives on the device - jobs of two epochs cannot share a launch)
            for (auto& g : groups) flush(g.first, g.second);
        for (auto& s : E->sides) {
            GemmSpec g;
            if (s.p != 1 || !gemm_of_iteration(E, s, it, g) || s.c1.count({it, g.op})) continue;
            cognn_keys k = gemm_keys(E, s, it, g);
            u64* c = c1_alloc(E, g.M * g.N);
            s.c1[{it, g.op}] = Side::C1{c, g.M * g.N};
            if (g.transA == 0 && E->dealer_group && E->be->cognn_dealer_gemm_c1_groupable(g.N, g.K)) {
                Pending& p = groups[{g.N, g.K}];
                cognn_dealer_job j;
                j.C1 = c; j.keys = k; j.M = g.M;
                p.jobs.push_back(j);
                if (p.jobs.size() == 16) flush({g.N, g.K}, p);
            } else if (g.transA != 0 && E->dealer_group) {  // (every side has scratch of its own: the jobs of one iteration share their fills)
                cognn_dealer_tn_job j;
                j.C1 = c; j.keys = k; j.M = g.M; j.N = g.N; j.K = g.K; j.transA = g.transA; j.scratchA = s.scratch; j.scratchB = s.scratch + g.M * g.K;
                tn_jobs.push_back(j);
            } else {
                BE(cognn_dealer_gemm_c1_u64(E->ctx, c, &k, g.M, g.N, g.K, g.transA, s.scratch, s.scratch + g.M * g.K));
            }
        }
        for (size_t b = 0; b < tn_jobs.size(); b += 16)
            BE(cognn_dealer_gemm_c1_tn_group_u64(E->ctx, tn_jobs.data() + b, (int32_t)std::min<size_t>(16, tn_jobs.size() - b)));
        tn_jobs.clear();
    }
    for (auto& g : groups) flush(g.first, g.second);
    set_salt_value(E, 0);
}

// One whole epoch [it, it + epoch) of a single-process run as a recorded launch sequence (COGNN_OPT_GRAPH_EPOCHS): the first
// epoch runs eagerly (allocations, pools and kernel attributes settle), the second is recorded while it is issued, every later
// one is the same recording replayed under its own epoch salt - dataset-sized graphs spend their epoch in launch overhead
// (about 100 launches of a few microseconds of work each).
void run_epoch(cognn_engine* E, int64_t it) {
    const int ep = epoch_len(E);
    auto eager = [&] { for (int j = 0; j < ep; ++j) { run_iteration(E, it + j); exchange_wait(E); } };
    set_salt(E, it);
    if (!E->graph_warm || E->graph_unsupported) { eager(); E->graph_warm = true; return; }
    if (E->graph_exec && E->retain_offline && E->graph_epoch != it / ep) { eager(); return; }   // retained products belong to the epoch they were dealt in
    if (!E->graph_exec) {
        if (E->be->cognn_graph_capture_begin(E->ctx) != 0) { E->graph_unsupported = true; eager(); return; }
        void* exec = nullptr;
        try {
            for (int j = 0; j < ep; ++j) run_iteration(E, it + j);
        } catch (...) {
            E->be->cognn_graph_capture_end(E->ctx, &exec);
            if (exec) E->be->cognn_graph_destroy(E->ctx, exec);
            throw;
        }
        BE(cognn_graph_capture_end(E->ctx, &exec));
        E->graph_exec = exec;
        E->graph_epoch = it / ep;
    }
    BE(cognn_graph_launch(E->ctx, E->graph_exec));
}

// identifies the run a cached product share belongs to: parties, ranks, variant, dimensions, graph size, rows of the owner
u64 run_fingerprint(cognn_engine* E, const Side& s) {
    u64 h = 0xcbf29ce484222325ull;
    auto mix = [&](u64 v) { for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xff; h *= 0x100000001b3ull; } };
    mix((u64)E->k); mix((u64)E->world); mix((u64)E->rank); mix((u64)E->cfg.variant);
    mix((u64)E->in()); mix((u64)E->hid()); mix((u64)E->lab());
    mix((u64)E->G.num_edges); mix((u64)E->G.row_of_vid.size()); mix((u64)s.owner); mix((u64)s.n);
    return h;
}
const u64 kC1Magic = 0x32435F4E4E474F43ull;                   // "COGNN_C2": header = magic, seed, M, N, K, transA, fingerprint

// ---------------------------------------------------------------------------------------------
// setup
// ---------------------------------------------------------------------------------------------
void build_layout(cognn_engine* E) {
    const int k = E->k;
    auto& G = E->G;
    auto nrows = [&](int p) { return (int64_t)G.party[p].localVertexPos.size(); };
    E->hosted.clear(); E->cohosted.clear();
    for (int p = 0; p < k; ++p) if (E->rank_of(p) == E->rank) E->hosted.push_back(p);
    E->cohosted = E->cohosted_of(E->rank);
    // sides in canonical (owner, p) order
    E->sides.clear();
    for (int o = 0; o < k; ++o) {
        for (int p = 0; p < 2; ++p) {
            if (E->holder(o, p) != E->rank) continue;
            Side s;
            s.owner = o; s.p = p; s.n = (int)nrows(o);
            s.peer_rank = E->holder(o, 1 - p);
            E->sides.push_back(s);
        }
    }
    for (auto& s : E->sides) s.peer = (s.peer_rank == E->rank) ? E->side(s.owner, 1 - s.p) : nullptr;
    // table rows: [A of hosted][B of co-hosted owners][B replicas of the others][inbox remote][inbox local][outbox]
    E->A_off.assign(k, -1); E->B_off.assign(k, -1);
    // every segment starts on an even row so that its byte offset is 16-byte aligned for any row width
    int64_t off = 0;
    auto even = [&]() { off = (off + 1) & ~(int64_t)1; };
    for (int p : E->hosted) { even(); E->A_off[p] = off; off += nrows(p); }
    for (int o : E->cohosted) { even(); E->B_off[o] = off; off += nrows(o); }
    even();
    E->aggRows = off;
    for (int o = 0; o < k; ++o) if (E->B_off[o] < 0) { even(); E->B_off[o] = off; off += nrows(o); }
    E->segs.clear();
    if (E->cfg.placement == COGNN_PLACE_VERTEX_SET) {
        // one trusted node: the own shares of the other ranks' vertex sets are replicated like the co-shares, and a co-share row
        // aggregates its remote sources' own-share rows directly - no partial-sum launch, no inbox, one exchange round per Gather
        // (the same bytes on the wire: a partial-sum segment of a dense graph has a row for nearly every vertex)
        for (int o = 0; o < k; ++o) if (E->A_off[o] < 0) { even(); E->A_off[o] = off; off += nrows(o); }
        even();
        E->inboxLocalOff = off; E->inboxRows = 0; E->partRows = 0;
        E->tableRows = off;
        return;
    }
    // partial-sum segments: (source rank -> g) goes to rank(co(g)); receiver-side order: source rank, then g
    const int64_t inbox0 = off;
    auto seg_vids = [&](int sr, int g) {                    // distinct destination vertices of g reached from any party of rank sr
        std::vector<uint64_t> v;
        for (int Q = sr * E->m; Q < (sr + 1) * E->m; ++Q) {
            if (Q == g) continue;
            const auto& rv = G.party[Q].out[g].rows_vid;
            v.insert(v.end(), rv.begin(), rv.end());
        }
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        return v;
    };
    std::vector<int> src_order;
    for (int r = 0; r < E->world; ++r) if (r != E->rank) src_order.push_back(r);
    src_order.push_back(E->rank);                          // own-rank segments last, adjacent to the outbox
    for (int sr : src_order) {
        if (sr == E->rank) { even(); E->inboxLocalOff = off; continue; }   // same-rank producers: read in place, see build_csrs
        for (int g : E->cohosted) {
            cognn_engine::Seg sg{g, 0, off, -1, sr, E->rank, seg_vids(sr, g)};
            sg.rows = (int64_t)sg.rows_vid.size();
            off += sg.rows;
            E->segs.push_back(std::move(sg));
        }
    }
    E->inboxRows = off - inbox0;
    // outbox: one segment per owner co-hosted elsewhere, grouped by destination rank in the receiver's order (its cohosted list)
    for (int dr = 0; dr < E->world; ++dr) {
        if (dr == E->rank) continue;
        for (int g : E->cohosted_of(dr)) {                  // owners whose co-share lives on rank dr
            cognn_engine::Seg sg{g, 0, -1, off, E->rank, dr, seg_vids(E->rank, g)};
            sg.rows = (int64_t)sg.rows_vid.size();
            off += sg.rows;
            E->segs.push_back(std::move(sg));
        }
    }
    E->tableRows = off;
    E->partRows = off - E->inboxLocalOff;
}

void build_csrs(cognn_engine* E) {
    auto& G = E->G;
    const int k = E->k;
    // ---- partial launch: rows = segments produced on this rank, in table order from inboxLocalOff ----
    std::vector<uint32_t> prp{0}, pcol;
    std::vector<const cognn_engine::Seg*> produced;
    for (auto& sg : E->segs) if (sg.src_rank == E->rank) produced.push_back(&sg);
    std::sort(produced.begin(), produced.end(), [](const cognn_engine::Seg* a, const cognn_engine::Seg* b) { return a->out_off < b->out_off; });
    int64_t expect = E->inboxLocalOff;
    for (auto* sg : produced) {
        if (sg->out_off != expect) throw EngineError("engine: partial segment layout is not contiguous");
        // per destination vertex: the source rows of every hosted party's edges into it
        std::vector<std::vector<uint32_t>> rows((size_t)sg->rows);
        for (int Q : E->hosted) {
            if (Q == sg->dst_owner) continue;
            const cognn::EdgeBlock& blk = G.party[Q].out[sg->dst_owner];
            const int64_t abase = E->A_off[Q];
            for (size_t r = 0; r + 1 < blk.rowptr.size(); ++r) {
                const size_t idx = (size_t)(std::lower_bound(sg->rows_vid.begin(), sg->rows_vid.end(), blk.rows_vid[r]) - sg->rows_vid.begin());
                for (uint32_t q = blk.rowptr[r]; q < blk.rowptr[r + 1]; ++q) rows[idx].push_back((uint32_t)(abase + blk.col[q]));
            }
        }
        for (auto& l : rows) { pcol.insert(pcol.end(), l.begin(), l.end()); prp.push_back((uint32_t)pcol.size()); }
        expect += sg->rows;
    }
    E->partEdges = (int64_t)pcol.size();
    // ---- aggregate launch: rows = A rows of hosted parties then B rows of co-hosted owners ----
    std::vector<std::vector<uint32_t>> lists((size_t)E->aggRows);
    for (int P : E->hosted) {
        const int64_t rbase = E->A_off[P];
        const cognn::EdgeBlock& self = G.party[P].out[P];
        for (size_t r = 0; r + 1 < self.rowptr.size(); ++r)
            for (uint32_t q = self.rowptr[r]; q < self.rowptr[r + 1]; ++q) lists[rbase + r].push_back((uint32_t)(rbase + self.col[q]));
        for (int Q = 0; Q < k; ++Q) {                      // in-edges from party Q, evaluated on the replica of Q's co-share
            if (Q == P) continue;
            const cognn::EdgeBlock& blk = G.party[Q].out[P];
            for (size_t r = 0; r + 1 < blk.rowptr.size(); ++r) {
                const uint32_t lr = G.row_of_vid[blk.rows_vid[r]];
                for (uint32_t q = blk.rowptr[r]; q < blk.rowptr[r + 1]; ++q) lists[rbase + lr].push_back((uint32_t)(E->B_off[Q] + blk.col[q]));
            }
        }
    }
    for (int o : E->cohosted) {
        const int64_t rbase = E->B_off[o];
        const cognn::EdgeBlock& self = G.party[o].out[o];
        for (size_t r = 0; r + 1 < self.rowptr.size(); ++r)
            for (uint32_t q = self.rowptr[r]; q < self.rowptr[r + 1]; ++q) lists[rbase + r].push_back((uint32_t)(rbase + self.col[q]));
    }
    // in-device exchange: when the producing party is hosted on this rank too, the co-party rows gather the producer's
    // own-share rows directly (edge by edge) instead of going through materialised partial sums (vertex-set placement: every
    // producer - the remote ones through the replicas of their own shares)
    std::vector<int> producers = E->hosted;
    if (E->cfg.placement == COGNN_PLACE_VERTEX_SET) { producers.clear(); for (int Q = 0; Q < k; ++Q) producers.push_back(Q); }
    for (int g : E->cohosted)
        for (int Q : producers) {
            if (Q == g) continue;
            const cognn::EdgeBlock& blk = G.party[Q].out[g];
            const int64_t rbase = E->B_off[g];
            for (size_t r = 0; r + 1 < blk.rowptr.size(); ++r) {
                const uint32_t lr = G.row_of_vid[blk.rows_vid[r]];
                for (uint32_t q = blk.rowptr[r]; q < blk.rowptr[r + 1]; ++q) lists[rbase + lr].push_back((uint32_t)(E->A_off[Q] + blk.col[q]));
            }
        }
    for (auto& sg : E->segs) {                              // partial rows received from other ranks
        if (sg.dst_rank != E->rank) continue;
        const int64_t rbase = E->B_off[sg.dst_owner];
        for (size_t r = 0; r < sg.rows_vid.size(); ++r) lists[rbase + G.row_of_vid[sg.rows_vid[r]]].push_back((uint32_t)(sg.inbox_off + r));
    }
    // entries that read rows held on this rank (table rows < aggRows) / rows received from other ranks (replicas, inbox)
    std::vector<uint32_t> arp{0}, acol, rrp{0}, rcol;
    for (auto& l : lists) {
        for (uint32_t c : l) ((int64_t)c < E->aggRows ? acol : rcol).push_back(c);
        arp.push_back((uint32_t)acol.size()); rrp.push_back((uint32_t)rcol.size());
    }
    E->aggEdges = (int64_t)acol.size(); E->remEdges = (int64_t)rcol.size();
    E->agg_rowptr = upload(E, arp); E->agg_col = upload(E, acol);
    E->rem_rowptr = upload(E, rrp); E->rem_col = upload(E, rcol);
    E->part_rowptr = upload(E, prp); E->part_col = upload(E, pcol);
}

// Single-process runs (every party hosted here): degrees and the aggregate CSR are built on the device straight from the edge
// list (cognn_graph_build_colocated) instead of bucketing / sorting edges and assembling per-row lists on the host.
void build_graph_on_device(cognn_engine* E, int64_t V, int64_t Ecount, const int64_t* src, const int64_t* dst, bool undirected) {
    auto& G = E->G;
    const int64_t total = undirected ? 2 * Ecount : Ecount;
    G.num_edges = total;
    struct Tmp {                                             // device temporaries of the build, released when it is done
        cognn_engine* E; std::vector<void*> p;
        void* get(size_t bytes) { void* q = nullptr; if (E->be->cognn_malloc(E->ctx, &q, bytes ? bytes : 16) != 0) throw EngineError(std::string(E->be->cognn_last_error())); p.push_back(q); return q; }
        ~Tmp() { for (void* q : p) E->be->cognn_free(E->ctx, q); }
    } tmp{E, {}};
    int64_t* dsrc = (int64_t*)tmp.get((size_t)Ecount * 8);
    int64_t* ddst = (int64_t*)tmp.get((size_t)Ecount * 8);
    int32_t* dtid = (int32_t*)tmp.get((size_t)V * 4);
    uint32_t* drow = (uint32_t*)tmp.get((size_t)V * 4);
    int64_t* daoff = (int64_t*)tmp.get((size_t)E->k * 8);
    int64_t* dboff = (int64_t*)tmp.get((size_t)E->k * 8);
    uint32_t* dtin = (uint32_t*)tmp.get((size_t)V * 4);
    uint32_t* din = (uint32_t*)tmp.get((size_t)V * 4);
    uint32_t* dout = (uint32_t*)tmp.get((size_t)V * 4);
    uint8_t* dborder = (uint8_t*)tmp.get((size_t)V);
    uint8_t* ddummy = (uint8_t*)tmp.get((size_t)V);
    uint32_t* scratch = (uint32_t*)tmp.get((size_t)(V + 2 * E->tableRows + 2) * 4);
    if (Ecount > 0) {
        BE(cognn_memcpy_h2d(E->ctx, dsrc, src, (size_t)Ecount * 8));
        BE(cognn_memcpy_h2d(E->ctx, ddst, dst, (size_t)Ecount * 8));
    }
    if (V > 0) {
        BE(cognn_memcpy_h2d(E->ctx, dtid, G.tid.data(), (size_t)V * 4));
        BE(cognn_memcpy_h2d(E->ctx, drow, G.row_of_vid.data(), (size_t)V * 4));
    }
    BE(cognn_memcpy_h2d(E->ctx, daoff, E->A_off.data(), (size_t)E->k * 8));
    BE(cognn_memcpy_h2d(E->ctx, dboff, E->B_off.data(), (size_t)E->k * 8));
    E->aggEdges = 2 * total; E->remEdges = 0; E->partEdges = 0;
    E->agg_rowptr = dalloc<uint32_t>(E, (size_t)E->tableRows + 1);
    E->agg_col = dalloc<uint32_t>(E, (size_t)E->aggEdges);
    BE(cognn_graph_build_colocated(E->ctx, V, Ecount, undirected ? 1 : 0, dsrc, ddst, dtid, drow, daoff, dboff, E->tableRows, E->agg_rowptr, E->agg_col,
                                   dtin, din, dout, dborder, ddummy, scratch));
    const std::vector<uint32_t> zero{0};
    E->part_rowptr = upload(E, zero); E->part_col = upload(E, std::vector<uint32_t>());
    E->rem_rowptr = upload(E, std::vector<uint32_t>((size_t)E->aggRows + 1, 0)); E->rem_col = upload(E, std::vector<uint32_t>());
    // the per-vertex results the host still needs (feature pre-scale, normalisers, metrics, cognn_engine_party_degrees)
    std::vector<uint32_t> tin((size_t)V), in((size_t)V), out((size_t)V);
    std::vector<uint8_t> border((size_t)V), dummy((size_t)V);
    if (V > 0) {
        BE(cognn_memcpy_d2h(E->ctx, tin.data(), dtin, (size_t)V * 4));
        BE(cognn_memcpy_d2h(E->ctx, in.data(), din, (size_t)V * 4));
        BE(cognn_memcpy_d2h(E->ctx, out.data(), dout, (size_t)V * 4));
        BE(cognn_memcpy_d2h(E->ctx, border.data(), dborder, (size_t)V));
        BE(cognn_memcpy_d2h(E->ctx, dummy.data(), ddummy, (size_t)V));
    }
    for (int64_t v = 0; v < V; ++v) {
        auto& pg = G.party[G.tid[v]];
        const uint32_t r = G.row_of_vid[v];
        pg.trueInDeg[r] = tin[v]; pg.inDeg[r] = in[v]; pg.outDeg[r] = out[v]; pg.isBorder[r] = border[v]; pg.selfDummy[r] = dummy[v];
    }
}

void alloc_sides(cognn_engine* E) {
    const int in = E->in(), hid = E->hid(), lab = E->lab();
    const int fm = std::max(hid, lab);
    E->Fmp = fm;
    E->table = dalloc<u64>(E, (size_t)E->tableRows * fm);
    E->aggOut = dalloc<u64>(E, (size_t)E->aggRows * fm);
    {
        const size_t welems = std::max((size_t)in * hid, (size_t)hid * lab);
        for (int j = 0; j < 6; ++j) E->wa[j] = dalloc<u64>(E, welems);
        E->wa_stride = (welems + 1) & ~(size_t)1;
        for (int j = 0; j < 2; ++j) E->wa_recv[j] = dalloc<u64>(E, E->wa_stride * (size_t)E->world);
    }
    for (auto& s : E->sides) {
        const size_t n = (size_t)s.n;
        const size_t big = std::max<size_t>({n * (size_t)in, (size_t)in * hid, n * (size_t)fm, (size_t)hid * lab});
        s.feat = dalloc<u64>(E, n * in);
        s.featE = dalloc<u64>(E, n * in);
        s.W[0] = dalloc<u64>(E, (size_t)in * hid);
        s.W[1] = dalloc<u64>(E, (size_t)hid * lab);
        s.h1 = dalloc<u64>(E, n * hid);
        s.h1E = dalloc<u64>(E, n * hid);
        s.g = dalloc<u64>(E, n * hid);
        if (original(E)) { s.ah[0] = dalloc<u64>(E, n * in); s.ah[1] = dalloc<u64>(E, n * hid); }
        s.relu_mask = dalloc<uint8_t>(E, n * hid);
        for (int j = 0; j < 2; ++j) s.buf[j] = dalloc<u64>(E, n * fm);
        for (int j = 0; j < 3; ++j) s.ob[j] = dalloc<u64>(E, big);
        for (int j = 0; j < 3; ++j) s.small[j] = dalloc<u64>(E, (size_t)in * hid + (size_t)hid * lab);
        s.scratch = dalloc<u64>(E, big + std::max<size_t>({(size_t)in * hid, n * (size_t)fm, (size_t)hid * lab}));
        {
            const size_t zcap = std::max<size_t>({(size_t)in * hid, n * (size_t)fm, (size_t)hid * lab});
            s.zbuf = dalloc<u64>(E, zcap);
            BE(cognn_memset0(E->ctx, s.zbuf, zcap * 8));
            s.z_dirty = 0; s.z_zero = (int64_t)zcap;
        }
        s.svec = dalloc<u64>(E, n);
        if (s.p == 0) {
            s.labels = dalloc<int32_t>(E, n);
            s.border = dalloc<uint8_t>(E, n);
            s.pfx = dalloc<u64>(E, n * lab);
            s.counts = dalloc<int64_t>(E, 6);
            s.loss = dalloc<double>(E, 1);
        }
    }
    for (auto& s : E->sides) {
        const size_t n = (size_t)s.n;
        const size_t big = std::max<size_t>({n * (size_t)in, (size_t)in * hid, n * (size_t)fm, (size_t)hid * lab});
        s.featE_peer = s.peer ? s.peer->featE : dalloc<u64>(E, n * (size_t)in);
        s.h1E_peer = s.peer ? s.peer->h1E : dalloc<u64>(E, n * (size_t)hid);
        for (int j = 0; j < 3; ++j) {
            if (s.peer) s.ib[j] = s.peer->ob[j];           // in-device exchange: read the peer's outbox directly
            else { s.ib_store[j] = dalloc<u64>(E, big); s.ib[j] = s.ib_store[j]; }
        }
    }
}

std::vector<double> glorot(int d0, int d1) {               // gcn.h:838-852, libc rand() re-seeded per matrix
    // libc's generator is ONE state per process: two engines starting side by side (one per host thread) would interleave their
    // srand / rand sequences and end up with different weights - the whole sequence of a matrix runs under one lock
    static std::mutex libc_rand_mu;
    std::lock_guard<std::mutex> lock(libc_rand_mu);
    std::vector<double> w((size_t)d0 * d1);
    std::srand(42);
    const double limit = std::sqrt(6.0 / (d0 + d1));
    for (int i = 0; i < d0; ++i)
        for (int j = 0; j < d1; ++j) w[(size_t)i * d1 + j] = (double)std::rand() / RAND_MAX * 2 * limit - limit;
    return w;
}

// The layer-0 product's feature opening E_p = X_p - A_p (fixed-operand mask reuse): opened, exchanged and summed
// once in start() - or, with recorded epochs, at the first iteration of every epoch with that epoch's mask.
void open_features(cognn_engine* E) {
    const int in = E->in();
    for (auto& s : E->sides) {
        cognn_keys fk = feature_gemm_keys(E, s.owner, 0);
        BE(cognn_mask_open_u64(E->ctx, s.featE, s.feat, fk.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], (int64_t)s.n, in, COGNN_MASK_OPEN_LIMB));
    }
    {
        XList xl;
        for (auto& s : E->sides) {
            if (s.peer) continue;
            xl.send(s.peer_rank, s.featE, (int64_t)s.n * in * 8);
            xl.recv(s.peer_rank, s.featE_peer, (int64_t)s.n * in * 8);
        }
        run_exchange_sync(E, xl);
    }
    // E = E_0 + E_1: summed once (in place; a co-located pair shares the owner side's copy, so its two GEMMs read the same
    // 8*n*in bytes)
    for (auto& s : E->sides) {
        if (s.peer && s.p == 1) continue;
        BE(cognn_add_u64(E->ctx, s.featE, s.featE, s.featE_peer, (int64_t)s.n * in));
    }
    for (auto& s : E->sides) s.featSum = (s.peer && s.p == 1) ? s.peer->featE : s.featE;
    // ... and, for the grouped layer-0 product, once more in the order its A fragments have (same size, no split in the K loop)
    if (E->gemm_group && E->gemm_presplit) {
        for (auto& s : E->sides) {
            if (s.peer && s.p == 1) continue;
            const int64_t bytes = E->be->cognn_gemm_presplit_bytes((int64_t)s.n, in);
            if (bytes <= 0) continue;
            if (!s.featPl) s.featPl = dalloc<unsigned char>(E, (size_t)bytes);
            BE(cognn_gemm_presplit_u64(E->ctx, (void*)s.featPl, s.featSum, nullptr, (int64_t)s.n, in));
        }
        for (auto& s : E->sides) if (s.peer && s.p == 1) s.featPl = s.peer->featPl;
        int64_t tiles_all = 0;
        for (auto& s : E->sides) tiles_all += (s.n + 15) / 16;
        // (the mask is the same in every epoch only outside recorded epochs; only the whole-K form of the grouped launch reads the image)
        if (E->gemm_mask_image && !E->graph_epochs && E->hid() > 16 && E->be->cognn_beaver_gemm_group_is_whole_k(E->hid(), in, tiles_all))
            for (auto& s : E->sides) {
                const int64_t bytes = E->be->cognn_gemm_presplit_bytes((int64_t)s.n, in);
                if (bytes <= 0 || !s.featPl) continue;
                cognn_keys fk = feature_gemm_keys(E, s.owner, 0);
                if (!s.featMaskPl) s.featMaskPl = dalloc<unsigned char>(E, (size_t)bytes);
                BE(cognn_gemm_mask_fill_u64(E->ctx, s.ob[0], fk.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], (int64_t)s.n * in));   // mask VALUES (limb form)
                BE(cognn_gemm_presplit_u64(E->ctx, (void*)s.featMaskPl, s.ob[0], nullptr, (int64_t)s.n, in));
            }
        // the layer-0 weight gradient X^T . g reads the same opening and the same mask along the other axis: its kernel's A fragments,
        // for N = hidden_dim > 16 (cognn_gemm_presplit_tn_u64)
        if (E->gemm_mask_image && !E->graph_epochs && E->cfg.variant != COGNN_VARIANT_OPTIMIZE_GCN_INFERENCE && E->hid() > 16)
            for (auto& s : E->sides) {
                const int64_t bytes = E->be->cognn_gemm_presplit_tn_bytes(in, (int64_t)s.n);
                if (bytes <= 0) continue;
                cognn_keys fk = feature_gemm_keys(E, s.owner, 0);
                if (!(s.peer && s.p == 1)) {
                    if (!s.featTnPl) s.featTnPl = dalloc<unsigned char>(E, (size_t)bytes);
                    BE(cognn_gemm_presplit_tn_u64(E->ctx, (void*)s.featTnPl, s.featSum, nullptr, 0, 1, in, (int64_t)s.n));
                }
                if (!s.featMaskTnPl) s.featMaskTnPl = dalloc<unsigned char>(E, (size_t)bytes);
                BE(cognn_gemm_presplit_tn_u64(E->ctx, (void*)s.featMaskTnPl, nullptr, nullptr, fk.k[s.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], 1, in, (int64_t)s.n));
            }
        for (auto& s : E->sides) if (s.peer && s.p == 1) s.featTnPl = s.peer->featTnPl;
    }
}

// Ranks that disagree on what they run (placement, variant, dimensions, seed, graph) would issue mismatched send / receive lists
// and hang in the transport: every rank swaps a fingerprint of its configuration with every other rank first and fails fast.
// (The receive slots start out holding this rank's own value, so a transport that moves nothing - tools/rank_compute_probe.py -
// passes.)
void config_handshake(cognn_engine* E) {
    if (E->world == 1) return;
    u64 h = 0xcbf29ce484222325ull;
    auto mix = [&](u64 v) { for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xff; h *= 0x100000001b3ull; } };
    mix((u64)E->k); mix((u64)E->world); mix((u64)E->cfg.variant); mix((u64)E->cfg.placement); mix((u64)E->in()); mix((u64)E->hid());
    mix((u64)E->lab()); mix(E->cfg.seed); mix((u64)E->G.num_edges); mix((u64)E->G.row_of_vid.size()); mix((u64)E->cfg.undirected);
    std::vector<u64> host((size_t)E->world + 1, h);
    u64* d = upload(E, host);
    XList xl;
    for (int r = 0; r < E->world; ++r) {
        if (r == E->rank) continue;
        xl.send(r, d + E->world, 8);
        xl.recv(r, d + r, 8);
    }
    run_exchange_sync(E, xl);
    BE(cognn_ctx_sync(E->ctx));
    BE(cognn_memcpy_d2h(E->ctx, host.data(), d, host.size() * 8));
    for (int r = 0; r < E->world; ++r)
        if (host[(size_t)r] != h)
            throw EngineError("engine: rank " + std::to_string(r) + " runs a different configuration than rank " + std::to_string(E->rank) +
                              " (placement, variant, dimensions, seed or graph differ)");
}

void start(cognn_engine* E) {
    const int in = E->in(), hid = E->hid(), lab = E->lab();
    config_handshake(E);
    if (E->w0.empty()) { E->w0 = glorot(in, hid); E->w1 = glorot(hid, lab); }
    std::vector<u64> wfx[2];
    for (double v : E->w0) wfx[0].push_back(fx_llround(v));
    for (double v : E->w1) wfx[1].push_back(fx_llround(v));
    u64* dW[2] = {upload(E, wfx[0]), upload(E, wfx[1])};
    for (auto& s : E->sides) {
        const cognn::PartyGraph& pg = E->G.party[s.owner];
        const size_t n = (size_t)s.n;
        const u64 fkey = cognn_stream_key(E->cfg.seed, (u64)s.owner, 0, COGNN_OP_SHARE_FEAT, 0);
        if (s.p == 0) {
            const int slot = (int)(std::find(E->hosted.begin(), E->hosted.end(), s.owner) - E->hosted.begin());
            if (E->hostFeat[slot].size() != n * in) throw EngineError("engine: features of party " + std::to_string(s.owner) + " were not set");
            std::vector<double> rs(n);
            for (size_t r = 0; r < n; ++r) rs[r] = std::pow((double)pg.trueInDeg[r] + 1.0, -0.5);   // normalizeFeatureVec, gcn.h:819-835
            double* dF = upload(E, E->hostFeat[slot]);
            double* dR = upload(E, rs);
            BE(cognn_fx_encode_f64(E->ctx, dF, dR, s.ob[0], (int64_t)n, in));
            BE(cognn_share_split_u64(E->ctx, s.ob[0], fkey, s.feat, nullptr, (int64_t)n * in));    // CryptoUtil::intoShares, gcn.h:64-75
            std::vector<u64> sv(n);
            for (size_t r = 0; r < n; ++r) sv[r] = pg.inDeg[r] == 0 ? 0 : fx_llround(std::pow((double)pg.inDeg[r] + 1.0, -0.5));   // gcn.h:219-221
            BE(cognn_memcpy_h2d(E->ctx, s.svec, sv.data(), n * 8));
            BE(cognn_memcpy_h2d(E->ctx, s.labels, E->hostLabels[slot].data(), n * 4));
            BE(cognn_memcpy_h2d(E->ctx, s.border, pg.isBorder.data(), n));
        } else {
            BE(cognn_prng_fill_u64(E->ctx, s.feat, fkey, (int64_t)n * in));   // the co-party's share is the mask itself
            BE(cognn_memset0(E->ctx, s.svec, n * 8));                          // server passes zero degrees (ss_...h:985-989)
        }
        for (int l = 0; l < 2; ++l) {                      // intoShareTensor, gcn.h:85-99,880-882; ring hand-off ss_...h:231-232
            const u64 wkey = cognn_stream_key(E->cfg.seed, (u64)s.owner, 0, COGNN_OP_SHARE_W, (u64)l);
            const int64_t elems = l == 0 ? (int64_t)in * hid : (int64_t)hid * lab;
            BE(cognn_share_split_u64(E->ctx, dW[l], wkey, s.p == 0 ? s.W[l] : nullptr, s.p == 1 ? s.W[l] : nullptr, elems));
        }
        s.cur = s.feat; s.curF = in;
    }
    if (!original(E)) open_features(E);                     // (original-gcn has no product on the constant feature tensor)
    BE(cognn_ctx_sync(E->ctx));
    E->started = true;
}

int guard(const std::function<void()>& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& ex) {
        g_engine_error = ex.what();
        return 1;
    }
}
}  // namespace

extern "C" {

const char* cognn_engine_last_error(void) { return g_engine_error.c_str(); }

int cognn_engine_create(const cognn_engine_config* cfg, int64_t V, int64_t Ecount, const int64_t* src, const int64_t* dst,
                        const int32_t* part, cognn_engine** out) {
    return guard([&] {
        if (!cfg || !out || !part || (Ecount > 0 && (!src || !dst))) throw EngineError("cognn_engine_create: null argument");
        if (cfg->num_parties < 2) throw EngineError("cognn_engine_create: need at least 2 parties");
        if (cfg->world < 1 || cfg->num_parties % cfg->world != 0 || cfg->rank < 0 || cfg->rank >= cfg->world)
            throw EngineError("cognn_engine_create: the number of parties must be a multiple of the number of ranks");
        if (cfg->num_layers != 2) throw EngineError("cognn_engine_create: num_layers must be 2 (as in every reference config)");
        cognn_engine* E = new cognn_engine();
        E->cfg = *cfg;
        if (E->cfg.placement != COGNN_PLACE_PARTY && E->cfg.placement != COGNN_PLACE_VERTEX_SET) { delete E; throw EngineError("cognn_engine_create: unknown placement"); }
        E->be = cognn_default_backend();
        E->k = cfg->num_parties; E->world = cfg->world; E->rank = cfg->rank; E->m = E->k / E->world;
        if (E->be->cognn_ctx_create(cfg->device, cfg->stream, &E->ctx) != 0) {
            std::string msg = E->be->cognn_last_error();
            delete E;
            throw EngineError(msg);
        }
        try {
            // COGNN_HOST_GRAPH_BUILD forces the host builder (tests compare the two)
            if (original(E) && E->world != 1) throw EngineError("cognn_engine_create: original-gcn runs as a single process (world = 1)");
            if (original(E) && E->k > 16) throw EngineError("cognn_engine_create: original-gcn takes at most 16 parties");
            const bool device_build = E->world == 1 && !getenv("COGNN_HOST_GRAPH_BUILD") && !original(E);   // (original-gcn indexes single edges: host builder)
            if (device_build) {
                E->G = cognn::build_vertex_layout(E->k, V, part);
                build_layout(E);
                build_graph_on_device(E, V, Ecount, src, dst, cfg->undirected != 0);
            } else {
                E->G = cognn::build_partitioned_graph(E->k, V, Ecount, src, dst, part, cfg->undirected != 0);
                build_layout(E);
                build_csrs(E);
            }
            if (original(E)) build_original_index(E);
            alloc_sides(E);
            E->hostFeat.resize(E->hosted.size());
            E->hostLabels.resize(E->hosted.size());
        } catch (...) {
            cognn_engine_destroy(E);
            throw;
        }
        *out = E;
    });
}

int cognn_engine_destroy(cognn_engine* E) {
    if (!E) return 0;
    if (E->ctx) {
        if (E->graph_exec) E->be->cognn_graph_destroy(E->ctx, E->graph_exec);
        if (E->salt_on_device) E->be->cognn_set_epoch_salt(E->ctx, 0);
        for (void* p : E->allocs) E->be->cognn_free(E->ctx, p);
        E->be->cognn_ctx_destroy(E->ctx);
    }
    delete E;
    return 0;
}

int cognn_engine_set_exchange(cognn_engine* E, cognn_exchange_fn fn, void* user) {
    return guard([&] {
        if (!E) throw EngineError("null engine");
        E->xfn = fn; E->xwait = nullptr; E->xwait_round = nullptr; E->xuser = user; E->xbegun = E->xdone = 0;
    });
}
int cognn_engine_set_exchange_async2(cognn_engine* E, cognn_exchange_fn begin_fn, cognn_exchange_wait_fn wait_fn,
                                     cognn_exchange_wait_round_fn wait_round_fn, void* user) {
    return guard([&] {
        if (!E) throw EngineError("null engine");
        if (!begin_fn || !wait_fn) throw EngineError("cognn_engine_set_exchange_async2: the begin and wait functions are required");
        E->xfn = begin_fn; E->xwait = wait_fn; E->xwait_round = wait_round_fn; E->xuser = user; E->xbegun = E->xdone = 0;
    });
}
int cognn_engine_set_exchange_async(cognn_engine* E, cognn_exchange_fn begin_fn, cognn_exchange_wait_fn wait_fn, void* user) {
    return guard([&] {
        if (!E) throw EngineError("null engine");
        if (!begin_fn || !wait_fn) throw EngineError("cognn_engine_set_exchange_async: both functions are required");
        E->xfn = begin_fn; E->xwait = wait_fn; E->xwait_round = nullptr; E->xuser = user; E->xbegun = E->xdone = 0;
    });
}

int cognn_engine_party_rows(cognn_engine* E, int32_t party, int64_t* rows) {
    return guard([&] {
        if (!E || party < 0 || party >= E->k || !rows) throw EngineError("cognn_engine_party_rows: bad arguments");
        *rows = (int64_t)E->G.party[party].localVertexPos.size();
    });
}
int cognn_engine_party_vids(cognn_engine* E, int32_t party, int64_t* vids) {
    return guard([&] {
        if (!E || party < 0 || party >= E->k || !vids) throw EngineError("cognn_engine_party_vids: bad arguments");
        const auto& v = E->G.party[party].localVertexPos;
        for (size_t i = 0; i < v.size(); ++i) vids[i] = (int64_t)v[i];
    });
}
int cognn_engine_party_degrees(cognn_engine* E, int32_t party, int64_t* tdeg, int64_t* ideg, uint8_t* border) {
    return guard([&] {
        if (!E || party < 0 || party >= E->k) throw EngineError("cognn_engine_party_degrees: bad arguments");
        const auto& pg = E->G.party[party];
        for (size_t i = 0; i < pg.localVertexPos.size(); ++i) {
            if (tdeg) tdeg[i] = pg.trueInDeg[i];
            if (ideg) ideg[i] = pg.inDeg[i];
            if (border) border[i] = pg.isBorder[i];
        }
    });
}
int cognn_engine_set_party_data(cognn_engine* E, int32_t party, const double* features, const int32_t* labels) {
    return guard([&] {
        if (!E || !features || !labels) throw EngineError("cognn_engine_set_party_data: null argument");
        auto it = std::find(E->hosted.begin(), E->hosted.end(), party);
        if (it == E->hosted.end()) throw EngineError("cognn_engine_set_party_data: party " + std::to_string(party) + " is not hosted on this rank");
        const size_t slot = (size_t)(it - E->hosted.begin());
        const size_t n = E->G.party[party].localVertexPos.size();
        E->hostFeat[slot].assign(features, features + n * (size_t)E->in());
        E->hostLabels[slot].assign(labels, labels + n);
        for (size_t i = 0; i < n; ++i)
            if (labels[i] < 0 || labels[i] >= E->lab()) throw EngineError("cognn_engine_set_party_data: label out of range");
    });
}
int cognn_engine_set_weights(cognn_engine* E, const double* w0, const double* w1) {
    return guard([&] {
        if (!E || !w0 || !w1) throw EngineError("cognn_engine_set_weights: null argument");
        E->w0.assign(w0, w0 + (size_t)E->in() * E->hid());
        E->w1.assign(w1, w1 + (size_t)E->hid() * E->lab());
    });
}
int cognn_engine_start(cognn_engine* E) {
    return guard([&] { if (!E) throw EngineError("null engine"); start(E); });
}
int cognn_engine_offline(cognn_engine* E, int64_t it0, int64_t it1) {
    return guard([&] {
        if (!E || !E->started) throw EngineError("cognn_engine_offline: engine not started");
        run_offline(E, it0, it1);
    });
}
static std::string c1_path(cognn_engine* E, const char* dir, const Side& s, int64_t it, int op) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s/c1_r%d_o%d_i%lld_op%d.bin", dir, E->rank, s.owner, (long long)it, op);
    return buf;
}
int cognn_engine_offline_save(cognn_engine* E, const char* dir) {
    return guard([&] {
        if (!E || !dir) throw EngineError("cognn_engine_offline_save: bad arguments");
        for (auto& s : E->sides)
            for (auto& kv : s.c1) {
                GemmSpec g;
                if (!gemm_of_iteration(E, s, kv.first.first, g) || g.op != kv.first.second || g.M * g.N != kv.second.elems)
                    throw EngineError("cognn_engine_offline_save: inconsistent product share table");
                const int64_t elems = kv.second.elems;
                std::vector<u64> host((size_t)elems);
                BE(cognn_memcpy_d2h(E->ctx, host.data(), kv.second.ptr, (size_t)elems * 8));
                const std::string path = c1_path(E, dir, s, kv.first.first, kv.first.second);
                FILE* f = fopen(path.c_str(), "wb");
                if (!f) throw EngineError("cognn_engine_offline_save: cannot write " + path);
                const u64 hdr[7] = {kC1Magic, E->cfg.seed, (u64)g.M, (u64)g.N, (u64)g.K, (u64)g.transA, run_fingerprint(E, s)};
                const bool ok = fwrite(hdr, 8, 7, f) == 7 && fwrite(host.data(), 8, (size_t)elems, f) == (size_t)elems;
                fclose(f);
                if (!ok) throw EngineError("cognn_engine_offline_save: short write to " + path);
            }
    });
}
int cognn_engine_offline_load(cognn_engine* E, const char* dir, int64_t it0, int64_t it1, int64_t* loaded) {
    return guard([&] {
        if (!E || !dir || !E->started) throw EngineError("cognn_engine_offline_load: bad arguments or engine not started");
        int64_t n = 0;
        for (auto& s : E->sides) {
            if (s.p != 1) continue;
            for (int64_t it = it0; it < it1; ++it) {
                GemmSpec g;
                if (!gemm_of_iteration(E, s, it, g) || s.c1.count({it, g.op})) continue;
                FILE* f = fopen(c1_path(E, dir, s, it, g.op).c_str(), "rb");
                if (!f) continue;
                u64 hdr[7];
                // only a file written for exactly this product of exactly this run is accepted; anything else is dealt on demand
                const bool match = fread(hdr, 8, 7, f) == 7 && hdr[0] == kC1Magic && hdr[1] == E->cfg.seed && hdr[2] == (u64)g.M &&
                                   hdr[3] == (u64)g.N && hdr[4] == (u64)g.K && hdr[5] == (u64)g.transA && hdr[6] == run_fingerprint(E, s);
                if (match) {
                    std::vector<u64> host((size_t)(g.M * g.N));
                    if (fread(host.data(), 8, host.size(), f) == host.size() && fgetc(f) == EOF) {
                        u64* c = c1_alloc(E, g.M * g.N);
                        BE(cognn_memcpy_h2d(E->ctx, c, host.data(), host.size() * 8));
                        s.c1[{it, g.op}] = Side::C1{c, g.M * g.N};
                        ++n;
                    }
                }
                fclose(f);
            }
        }
        if (loaded) *loaded = n;
    });
}
int cognn_engine_run(cognn_engine* E, int64_t it0, int64_t it1) {
    return guard([&] {
        if (!E || !E->started) throw EngineError("cognn_engine_run: engine not started");
        const int ep = epoch_len(E);
        struct SaltReset { cognn_engine* E; ~SaltReset() { if (E->salt_on_device) { E->be->cognn_set_epoch_salt(E->ctx, 0); E->salt_on_device = 0; } E->salt_now = 0; } } salt_reset{E};
        E->run_end = it1;
        E->prescaled_it = -1;                               // (nothing is carried from one call to the next)
        for (int64_t it = it0; it < it1; ++it) {
            if (E->graph_epochs && E->world == 1 && !E->cfg.verbose && !E->timing && it % ep == 0 && it + ep <= it1) {
                run_epoch(E, it);
                it += ep - 1;
                continue;
            }
            const int64_t rounds0 = E->rounds;
            double before[5] = {0, 0, 0, 0, 0};
            int64_t nl;
            if (E->cfg.verbose)
                for (int j = 0; j < 5; ++j) BE(cognn_timer_read(E->ctx, T_PH_PRESCATTER + j, &nl, &before[j]));
            run_iteration(E, it);
            exchange_wait(E);                               // nothing stays in flight across iterations / API calls
            if (E->cfg.verbose) {                           // (reads synchronise the stream)
                for (int j = 0; j < 5; ++j) {
                    double after = 0;
                    BE(cognn_timer_read(E->ctx, T_PH_PRESCATTER + j, &nl, &after));
                    E->phase_s[j] = (after - before[j]) * 1e-3;
                }
                E->phase_s[5] = (double)(E->rounds - rounds0);
                if (!E->timing) BE(cognn_timer_reset(E->ctx));   // keep the event lists short
            }
        }
    });
}
int cognn_engine_sync(cognn_engine* E) {
    return guard([&] { if (!E) throw EngineError("null engine"); exchange_wait(E); BE(cognn_ctx_sync(E->ctx)); });
}
int cognn_engine_get_phase_seconds(cognn_engine* E, double* out6) {
    return guard([&] {
        if (!E || !out6) throw EngineError("cognn_engine_get_phase_seconds: bad arguments");
        for (int j = 0; j < 6; ++j) out6[j] = E->phase_s[j];
    });
}
int cognn_engine_set_option(cognn_engine* E, int32_t option, int64_t value) {
    return guard([&] {
        if (!E) throw EngineError("null engine");
        if (option == COGNN_OPT_RETAIN_OFFLINE) E->retain_offline = value != 0;
        else if (option == COGNN_OPT_PAIR_FUSION) E->pair_fusion = value != 0;
        else if (option == COGNN_OPT_FORWARD_ONLY) E->forward_only = value != 0;
        else if (option == COGNN_OPT_PUBLIC_OPENINGS) E->public_openings = value != 0;
        else if (option == COGNN_OPT_DEALER_STREAMS) {
            if (value < 0 || value > 2) throw EngineError("cognn_engine_set_option: COGNN_OPT_DEALER_STREAMS takes 0, 1 (every dealt value streamed) or 2 (the dealer's corrections only)");
            E->dealer_streams = (int)value;
        }
        else if (option == COGNN_OPT_GRAPH_EPOCHS) {
            if (value != 0 && original(E)) throw EngineError("cognn_engine_set_option: recorded epochs are not supported for original-gcn");
            if (value != 0 && !E->graph_epochs) BE(cognn_ctx_use_private_stream(E->ctx));   // (the caller's stream may be the default stream, which cannot record)
            if (E->graph_exec) { E->be->cognn_graph_destroy(E->ctx, E->graph_exec); E->graph_exec = nullptr; }
            E->graph_epochs = value != 0; E->graph_warm = false;
        }
        else if (option == COGNN_OPT_EXCHANGE_CHUNKS) {
            if (value < 1 || value > 8) throw EngineError("cognn_engine_set_option: COGNN_OPT_EXCHANGE_CHUNKS takes 1..8");
            E->chunks = (int)value;
        }
        else throw EngineError("cognn_engine_set_option: unknown option");
    });
}
int cognn_engine_get_memory(cognn_engine* E, int64_t* allocations, int64_t* bytes) {
    return guard([&] {
        if (!E) throw EngineError("null engine");
        if (allocations) *allocations = (int64_t)E->allocs.size();
        if (bytes) *bytes = E->alloc_bytes;
    });
}
int cognn_engine_get_shares(cognn_engine* E, int32_t owner, int32_t sd, uint64_t* host_out, int64_t* rows, int64_t* cols) {
    return guard([&] {
        Side* s = E ? E->side(owner, sd) : nullptr;
        if (!s) throw EngineError("cognn_engine_get_shares: that share is not held on this rank");
        if (rows) *rows = s->n;
        if (cols) *cols = s->curF;
        if (host_out && s->curF > 0) {
            apply_cur_mask(E, *s);                          // (a deferred ReLU' selection becomes real for this reader)
            BE(cognn_memcpy_d2h(E->ctx, host_out, s->cur, (size_t)s->n * s->curF * 8));
        }
    });
}
int cognn_engine_get_weight(cognn_engine* E, int32_t owner, int32_t sd, int32_t layer, uint64_t* host_out) {
    return guard([&] {
        Side* s = E ? E->side(owner, sd) : nullptr;
        if (!s || layer < 0 || layer > 1 || !host_out) throw EngineError("cognn_engine_get_weight: bad arguments");
        const size_t elems = layer == 0 ? (size_t)E->in() * E->hid() : (size_t)E->hid() * E->lab();
        BE(cognn_memcpy_d2h(E->ctx, host_out, s->W[layer], elems * 8));
    });
}
int cognn_engine_get_metrics(cognn_engine* E, int32_t party, double* out8) {
    return guard([&] {
        Side* s = E ? E->side(party, 0) : nullptr;
        if (!s || !out8 || !s->has_metrics) throw EngineError("cognn_engine_get_metrics: no prediction layer has run for that party here");
        int64_t c[6]; double loss;
        BE(cognn_memcpy_d2h(E->ctx, c, s->counts, sizeof(c)));
        BE(cognn_memcpy_d2h(E->ctx, &loss, s->loss, sizeof(loss)));
        const auto& pg = E->G.party[party];
        const int64_t n = s->n;
        const int64_t train = (int64_t)((double)n * E->cfg.train_ratio), val = (int64_t)((double)n * E->cfg.val_ratio);
        int64_t nb = 0, nbt = 0, nbe = 0;
        for (int64_t r = 0; r < n; ++r) {
            if (!pg.isBorder[r]) continue;
            nb++;
            if (r < train) nbt++;
            if (r >= train + val) nbe++;
        }
        auto ratio = [](int64_t a, int64_t b) { return b > 0 ? (double)a / (double)b : 0.0; };
        out8[0] = ratio(c[0], n); out8[1] = ratio(c[1], train); out8[2] = ratio(c[2], nbt);
        out8[3] = ratio(c[3], n - train - val); out8[4] = ratio(c[4], nbe);
        out8[5] = n > 0 ? loss / (double)n : 0.0; out8[6] = (double)n; out8[7] = (double)nb;
    });
}
int cognn_engine_enable_timing(cognn_engine* E, int32_t on) {
    return guard([&] {
        if (!E) throw EngineError("null engine");
        E->timing = on != 0;
        BE(cognn_timer_reset(E->ctx));
        E->algo[0] = E->algo[1] = E->algo[2] = E->algo[T_GEMM_EPI] = E->algo[T_AGG_LAB] = 0;
    });
}
int cognn_engine_get_timing(cognn_engine* E, int32_t kind, int64_t* launches, double* total_ms, double* algo) {
    return guard([&] {
        if (!E || kind < 0 || kind > 4) throw EngineError("cognn_engine_get_timing: bad arguments");
        const int t = kind == 3 ? T_GEMM_EPI : kind == 4 ? T_AGG_LAB : kind;
        BE(cognn_timer_read(E->ctx, t, launches, total_ms));
        if (algo) *algo = E->algo[t];
    });
}
int cognn_engine_get_workload(cognn_engine* E, int64_t* out6) {
    return guard([&] {
        if (!E || !out6) throw EngineError("cognn_engine_get_workload: bad arguments");
        out6[0] = E->aggEdges + E->remEdges; out6[1] = E->aggRows; out6[2] = E->partEdges; out6[3] = E->partRows;
        out6[4] = E->G.num_edges; out6[5] = E->tableRows;
    });
}

}  // extern "C"
