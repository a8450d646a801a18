// Internal header of the host engine (cognn_amd/host/engine*.cpp): the state of a run (Side, cognn_engine), the small helpers and
// templates every translation unit uses, and the declarations of what the units export to each other.  Nothing here is API.
#pragma once
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


namespace cognn_eng {


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

}  // namespace cognn_eng
using namespace cognn_eng;   // (internal header: the engine state below is the C API's opaque type, at global scope)

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
    bool packed_openings = false;                   // COGNN_OPT_PACKED_OPENINGS: opened truncation / ReLU-product shares cross ranks as 6 bytes
    struct Wire { unsigned char* out; unsigned char* in; int64_t elems; };
    std::map<const u64*, Wire> wire;               // wire buffers of an outbox (keyed by the outbox), allocated at first use
    struct Unpack { int64_t round; u64* dst; const unsigned char* src; int64_t n; };
    std::vector<Unpack> unpacks;                    // received packed messages, restored once their round has completed
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
    // original-gcn across ranks (engine_original.cpp, build_original_ranks): the Scatter instances (client P -> destination party g) in
    // which this rank plays a role - the client (it holds A_P) or the server (B_P: the co-party of P for g == P, party g with its
    // replica otherwise) - as row ranges of ONE per-edge tensor
    struct OrigRole { int P = 0, g = 0, p = 0, peer_rank = 0, peer_role = -1; int64_t edges = 0, row0 = 0; u64* s[2] = {nullptr, nullptr}; };
    std::vector<OrigRole> orig_roles;          // canonical order: (P, g), client before server
    int64_t orig_rows = 0, orig_tab_rows = 0, orig_out_rows = 0;
    uint32_t *orig_id_rowptr = nullptr, *orig_src_col = nullptr;      // edge tensor <- one source row of the share table per edge
    uint32_t *orig_agg_rowptr = nullptr, *orig_agg_col = nullptr;     // hosted destination rows <- scaled edge rows
    uint32_t *orig_out_rowptr = nullptr, *orig_out_col = nullptr;     // client results for destination owners whose co-party is elsewhere
    struct OrigBlock { int g = 0, rank = 0; int64_t rows = 0, off = 0; };
    std::vector<OrigBlock> orig_send, orig_recv;                      // per (destination owner, peer rank), pre-summed over the rank's parties
    u64 *orig_tab = nullptr, *orig_acc = nullptr, *orig_edge = nullptr, *orig_out = nullptr, *orig_inb = nullptr;
    u64* orig_w[4] = {nullptr, nullptr, nullptr, nullptr};           // openings: E out / in, c out / in  [orig_rows x F]
    u64* orig_g[2] = {nullptr, nullptr};                              // scale openings out / in [orig_rows]
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

namespace cognn_eng {

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
// HIP-event bracket of one phase of an iteration (cfg.verbose)
struct Phase {
    cognn_engine* E; int kind; bool on;
    Phase(cognn_engine* e, int k) : E(e), kind(k), on(e->cfg.verbose != 0) { if (on) BE(cognn_timer_begin(E->ctx, kind)); }
    void end() { if (on) { on = false; BE(cognn_timer_end(E->ctx, kind)); } }
    ~Phase() { if (on) E->be->cognn_timer_end(E->ctx, kind); }
};

struct GemmSpec;

// ---------------------------------------------------------------------------------------------
// exchange
// ---------------------------------------------------------------------------------------------
struct XList {
    std::vector<cognn_xfer> v;
    void send(int peer, void* p, int64_t bytes) { if (bytes > 0) v.push_back(cognn_xfer{peer, 1, p, bytes}); }
    void recv(int peer, void* p, int64_t bytes) { if (bytes > 0) v.push_back(cognn_xfer{peer, 0, p, bytes}); }
};

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
// product-buffer bookkeeping (Side::z_dirty / z_zero): small products are handed back clean by their consumer
const int64_t kClearMaxElems = 1 << 19;                    // 4 MiB per side: above that the extra writes cost more than a zeroing launch

// lanes > 1 (independent multi-launch sequences per side, disjoint buffers): the sides of a pass go round-robin to that many
// launch lanes (cognn_lane_begin), joined before the pass ends.
struct Lanes {
    cognn_engine* E;
    int n, next = 0;
    Lanes(cognn_engine* e, int lanes) : E(e), n(lanes) { if (n > 1) BE(cognn_lane_begin(E->ctx, n)); }
    void advance() { if (n > 1) { BE(cognn_lane_select(E->ctx, next)); next = (next + 1) % n; } }
    ~Lanes() { if (n > 1) E->be->cognn_lane_end(E->ctx); }   // a failure here resurfaces at the next call (sticky HIP error)
};
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
// what follows a Beaver product on the same tensor (PreScatterComp: product, then row scale, gcn.h:233-254)
struct FollowScale {
    int op = 0, top = 0;
    std::function<u64*(Side&)> dst;
    int64_t it = -1;                                  // the iteration whose scale this is (its dealer streams); -1: the product's own
    std::function<const uint8_t*(Side&)> mask;        // a public selection between the truncation and the scale (COGNN_PC_MASK_AFTER_TRUNC)
    explicit operator bool() const { return (bool)dst; }
};

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

// row scale by the (owner-known) normaliser followed by truncation; x(side) [n x F] -> dst(side)
enum { E_FROM_X = 0, E_IN_X = 1, E_IN_OB0 = 2 };

// ---------------------------------------------------------------------------------------------
// schedule (Appendix A of SURVEY.md; gcn.h:893-948)
// ---------------------------------------------------------------------------------------------
struct IterInfo {
    int e, f, ep, layer;
    bool fwd, apply_only;
};
const u64 kC1Magic = 0x32435F4E4E474F43ull;                   // "COGNN_C2": header = magic, seed, M, N, K, transA, fingerprint

// ---- defined in the engine*.cpp translation units ------------------------------------------------------------------------
// engine_common.cpp
u64* c1_alloc(cognn_engine* E, int64_t elems);
void c1_release(cognn_engine* E, Side& s, std::pair<int64_t, int> key);
cognn_keys keys(cognn_engine* E, u64 owner, int64_t it, int op);
void set_salt_value(cognn_engine* E, u64 salt);
void set_salt(cognn_engine* E, int64_t it);
cognn_keys feature_gemm_keys(cognn_engine* E, u64 owner, int64_t it, int op = COGNN_OP_PS_GEMM);
void apply_cur_mask(cognn_engine* E, Side& s);
void attach_dealt(cognn_engine* E, cognn_pair_chain& c, int owner, int64_t it, int place);
double dealt_slots_read(cognn_engine* E, const cognn_pair_chain& c);
const u64* dealt_mask(cognn_engine* E, int owner, int64_t it, int place, u64 key, int64_t elems);
cognn_keys gemm_keys(cognn_engine* E, Side& s, int64_t it, const GemmSpec& g);
// engine.cpp
std::vector<double> glorot(int d0, int d1);
void open_features(cognn_engine* E);
void start(cognn_engine* E);
int guard(const std::function<void()>& f);
// engine_exchange.cpp
void exchange_wait(cognn_engine* E);
void exchange_wait_round(cognn_engine* E, int64_t round);
void run_exchange(cognn_engine* E, XList& xl, bool keep_inflight = false);
void run_exchange_sync(cognn_engine* E, XList& xl);
void exchange_ob(cognn_engine* E, int j, const std::vector<int64_t>& elems);
void exchange_ob2(cognn_engine* E, int j0, const std::vector<int64_t>& e0, int j1, const std::vector<int64_t>& e1);
std::vector<int64_t> per_side(cognn_engine* E, int64_t (*f)(cognn_engine*, Side&));
void msg_range(cognn_engine* E, XList& xl, Side& s, u64* out, u64* in, int64_t elems, int c, int C, bool packable = false);
void chunked_rounds(cognn_engine* E, const std::vector<Step>& steps, bool skip_paired = false);
void mp_exchange(cognn_engine* E, int F, u64* T);
void config_handshake(cognn_engine* E);
// engine_stages.cpp
void relu_stage(cognn_engine* E, int64_t it, bool e_opened, bool pairs_done);
void softmax_stage(cognn_engine* E, int64_t it, bool revealed);
void weight_average(cognn_engine* E, int64_t it, int layer);
bool weight_update_chain(cognn_engine* E, int64_t it, int layer, bool pairs_fused, bool raw,
                         const std::function<GemmSpec(Side&)>& specfn = nullptr);
// engine_schedule.cpp
u64* table_seg(cognn_engine* E, Side& s, int F);
void message_passing(cognn_engine* E, int F, int64_t it, bool open_scale);
bool can_fuse_gather_chain(const cognn_engine* E, int F);
void message_passing_fused(cognn_engine* E, int F, int64_t it, bool scale, bool relu_follows, const OpenNext& open_next, bool out_read,
                           bool softmax_follows = false, const u64* table = nullptr);
IterInfo iter_info(cognn_engine* E, int64_t it);
int mp_width(cognn_engine* E, int e);
GemmSpec prescatter_spec(cognn_engine* E, Side& s, int layer);
GemmSpec wgrad_spec(cognn_engine* E, Side& s, int layer, int64_t it);
void run_iteration(cognn_engine* E, int64_t it);
void run_epoch(cognn_engine* E, int64_t it);
// engine_original.cpp
void build_original_index(cognn_engine* E);
void build_original_ranks(cognn_engine* E);
u64 scatter_tag(int P, int g);
void run_iteration_original(cognn_engine* E, int64_t it);
// engine_offline.cpp
bool gemm_of_iteration(cognn_engine* E, Side& s, int64_t it, GemmSpec& g);
void run_offline(cognn_engine* E, int64_t it0, int64_t it1);
u64 run_fingerprint(cognn_engine* E, const Side& s);
// engine_layout.cpp
void build_layout(cognn_engine* E);
void build_csrs(cognn_engine* E);
void build_graph_on_device(cognn_engine* E, int64_t V, int64_t Ecount, const int64_t* src, const int64_t* dst, bool undirected);
void alloc_sides(cognn_engine* E);

inline Batch::Batch(cognn_engine* e) : E(e) { BE(cognn_batch_begin(E->ctx)); }
inline Batch::~Batch() { E->be->cognn_batch_end(E->ctx); }         // a failure here resurfaces at the next call (sticky HIP error)

template <class T>
T* dalloc(cognn_engine* E, size_t count) {
    void* p = nullptr;
    BE(cognn_malloc(E->ctx, &p, std::max<size_t>(count, 2) * sizeof(T)));
    E->allocs.push_back(p);
    E->alloc_bytes += (int64_t)(std::max<size_t>(count, 2) * sizeof(T));
    return (T*)p;
}
template <class T>
T* upload(cognn_engine* E, const std::vector<T>& v) {
    T* d = dalloc<T>(E, v.size());
    if (!v.empty()) BE(cognn_memcpy_h2d(E->ctx, d, v.data(), v.size() * sizeof(T)));
    return d;
}

// A dealer stream of GAS iteration `it` is addressed by (seed, owner, it % epoch, op, slot) through the key derivation and by the
// epoch number through the salt that the device adds to every key (cognn_spec.h): the arguments of an epoch's kernels do not
// depend on the epoch, so a recorded epoch can be replayed.
inline bool original(const cognn_engine* E) { return E->cfg.variant == COGNN_VARIANT_ORIGINAL_GCN; }
inline int epoch_len(const cognn_engine* E) { return (original(E) ? 2 : 3) * E->cfg.num_layers; }   // getEpochLayerNum: original-gcn/gcn.h:842-845, optimize-gcn/gcn.h:942-945

inline u64 fx_llround(double x) { return (u64)(long long)llround(x * (double)COGNN_FX_ONE); }
inline u64 fx_trunc(double x) { return (u64)(x * (double)COGNN_FX_ONE); }   // static_cast as in gcn.h:676,678,764

// Runs fn(side, index) for every hosted side: first the sides whose peer lives on this rank, then - once the exchange round
// that may still be in flight has completed - the sides whose peer is remote.  With an asynchronous exchange the interior
// sides' kernels overlap the boundary sides' messages.  batched: the calls are independent element-wise launches of one
// kind (see Batch).
// Co-located share-holders (both sides of an owner on this rank) run their two-party steps as ONE pair chain per owner
// (cognn_pair_chain_u64: both sides' local arithmetic in one kernel, opened values handed over in registers) instead of
// open -> HBM -> close passes; the per-side stages below then skip those sides.
inline bool paired(const cognn_engine* E, const Side& s) { return E->pair_fusion && s.peer != nullptr; }
inline bool z_is_zero(const Side& s, int64_t n) { return s.z_dirty == 0 && s.z_zero >= n; }
inline void z_written(Side& s, int64_t n) { s.z_dirty = std::max(s.z_dirty, n); }
inline bool z_clear_wanted(const Side& s, int64_t n) { return n <= kClearMaxElems && n >= s.z_dirty; }   // the clear leaves the whole buffer clean
inline void z_cleared(Side& s, int64_t n) { if (n >= s.z_dirty) s.z_dirty = 0; }
// A pair chain writes the opening of the step that follows it ONCE, as the sum of both parties' shares of it, into the owner
// side's buffer (COGNN_PC_OPEN_SUM): both sides of the pair read it from there as a pre-summed operand.
template <class Sel>
const u64* pair_opening(Side& s, Sel sel) { return s.p == 0 ? sel(s) : sel(*s.peer); }
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
inline bool streams_on(const cognn_engine* E) { return E->dealer_streams != 0 && E->retain_offline; }

// truncation of every side's `x` (elems) scaled by `mul`; result written/applied to dst(side)
template <class DstFn>
void trunc_stage(cognn_engine* E, int64_t it, int op, u64 mul, const std::vector<u64*>& x, const std::vector<int64_t>& elems,
                 DstFn dst, int mode, u64 owner_override = ~0ull, bool skip_paired = false) {
    std::vector<Step> steps(2);
    steps[0].fn = [&](Side& s, size_t i) {
        cognn_keys k = keys(E, owner_override == ~0ull ? (u64)s.owner : owner_override, it, op);
        BE(cognn_trunc_open_u64(E->ctx, s.ob[2], x[i], mul, &k, s.p, elems[i]));
    };
    steps[0].msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(E, xl, s, s.ob[2], s.ib[2], elems[i], c, C, true); };   // an opened truncation share: 6 bytes matter
    steps[1].fn = [&](Side& s, size_t i) {
        cognn_keys k = keys(E, owner_override == ~0ull ? (u64)s.owner : owner_override, it, op);
        BE(cognn_trunc_close_u64(E->ctx, dst(s), s.p == 0 ? s.ob[2] : nullptr, s.p == 0 ? s.ib[2] : nullptr, &k, s.p, mode, elems[i]));
    };
    chunked_rounds(E, steps, skip_paired);
}
// Public openings (COGNN_OPT_PUBLIC_OPENINGS): a truncation whose result feeds a Beaver opening is closed by
// BOTH parties from both opened values, and each derives the next opening E itself (cognn_trunc_close_pub_u64): ob[] then
// holds E, not E_p, the consumer takes it as the pre-summed opening and the exchange round that carried E_p disappears.
// Applies to the sides that are not part of a pair chain.
inline bool pub_open(const cognn_engine* E, const Side& s) { return E->public_openings && !paired(E, s); }
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
    steps[0].msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(E, xl, s, s.ob[2], s.ib[2], elems[i], c, C, true); };   // an opened truncation share: 6 bytes matter
    steps[1].fn = [&](Side& s, size_t i) { trunc_close_one(E, it, top, dst, elems, open_next, s, i); };
    chunked_rounds(E, steps, skip_paired);
}

}  // namespace cognn_eng
