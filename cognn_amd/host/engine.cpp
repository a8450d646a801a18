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
#include "engine_internal.h"

namespace cognn_eng {

thread_local std::string g_engine_error;


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

}  // namespace cognn_eng

using namespace cognn_eng;

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
            if (original(E) && E->world == 1) build_original_index(E);
            if (original(E) && E->world > 1) build_original_ranks(E);
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
        else if (option == COGNN_OPT_PACKED_OPENINGS) { exchange_wait(E); E->packed_openings = value != 0; }
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

