// Host engine, exchange rounds: the transport callbacks, outbox swaps, the chunked open -> exchange -> close pipeline, the message-passing exchange, the configuration hand-shake.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"

namespace cognn_eng {

// completes the round that is still in flight (asynchronous exchange); a no-op otherwise
static void unpack_completed(cognn_engine* E);
void exchange_wait(cognn_engine* E) {
    if (!E->xpending) { unpack_completed(E); return; }     // (a blocking transport completes its rounds inside run_exchange)
    E->xpending = false;
    E->xdone = E->xbegun;
    if (E->xwait(E->xuser) != 0) throw EngineError("engine: exchange wait function failed");
    unpack_completed(E);
}
// completes the rounds up to and including `round` (numbered by xbegun at their start); later rounds stay in flight when the
// transport can tell them apart (cognn_exchange_wait_round_fn) - otherwise everything enqueued is completed
void exchange_wait_round(cognn_engine* E, int64_t round) {
    if (!E->xpending || round < E->xdone) { unpack_completed(E); return; }
    if (!E->xwait_round || round + 1 >= E->xbegun) { exchange_wait(E); return; }
    E->xdone = round + 1;
    if (E->xwait_round(E->xuser, round) != 0) throw EngineError("engine: exchange wait function failed");
    unpack_completed(E);
}
// starts a round.  With a wait function registered the call only enqueues the messages; whoever consumes received data - or
// overwrites a buffer that is being sent - calls exchange_wait first (for_sides does, before it touches a side whose peer is remote).
void run_exchange(cognn_engine* E, XList& xl, bool keep_inflight) {
    // keep_inflight: the round already in flight keeps going (its buffers are disjoint from this round's and from the kernels
    // launched in between); exchange_wait then completes both
    if (!keep_inflight) exchange_wait(E);
    if (xl.v.empty()) return;
    if (!E->xfn) throw EngineError("engine: world > 1 needs an exchange function (cognn_engine_set_exchange)");
    if (E->xfn(E->xuser, xl.v.data(), (int32_t)xl.v.size()) != 0) throw EngineError("engine: exchange function failed");
    ++E->rounds;
    ++E->xbegun;
    if (E->xwait) E->xpending = true;
    else { E->xdone = E->xbegun; unpack_completed(E); }
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
// packable (an opened truncation share, or the ReLU's opened product: only the top 48 bits enter the close, cognn_open_hi48) with
// COGNN_OPT_PACKED_OPENINGS: the chunk is packed to 6 bytes per element into the outbox's wire buffer, the wire buffers travel, and the
// received one is restored into the inbox once its round has completed (exchange_wait / exchange_wait_round)
void msg_range(cognn_engine* E, XList& xl, Side& s, u64* out, u64* in, int64_t elems, int c, int C, bool packable) {
    int64_t lo, hi;
    cognn_chunk_range(elems, c, C, &lo, &hi);
    if (hi <= lo) return;
    if (!packable || !E->packed_openings) {
        xl.send(s.peer_rank, out + lo, (hi - lo) * 8);
        xl.recv(s.peer_rank, in + lo, (hi - lo) * 8);
        return;
    }
    auto f = E->wire.find(out);
    if (f == E->wire.end() || f->second.elems < elems) {
        cognn_engine::Wire w;
        w.elems = elems;
        w.out = dalloc<unsigned char>(E, (size_t)elems * 6 + 16);
        w.in = dalloc<unsigned char>(E, (size_t)elems * 6 + 16);
        f = E->wire.insert_or_assign(out, w).first;       // (a grown buffer replaces the old one; the old one is freed with the engine)
    }
    const int64_t n = hi - lo;                             // (chunk bounds are even: lo * 6 is a multiple of 4)
    BE(cognn_pack48_u64(E->ctx, f->second.out + lo * 6, out + lo, n));
    xl.send(s.peer_rank, f->second.out + lo * 6, n * 6);
    xl.recv(s.peer_rank, f->second.in + lo * 6, n * 6);
    E->unpacks.push_back(cognn_engine::Unpack{E->xbegun, in + lo, f->second.in + lo * 6, n});   // the round about to begin
}
// restores the packed messages of the rounds that have completed
static void unpack_completed(cognn_engine* E) {
    if (E->unpacks.empty()) return;
    std::vector<cognn_engine::Unpack> later;
    for (auto& u : E->unpacks) {
        if (u.round < E->xdone) BE(cognn_unpack48_u64(E->ctx, u.dst, u.src, u.n));
        else later.push_back(u);
    }
    E->unpacks.swap(later);
}
void chunked_rounds(cognn_engine* E, const std::vector<Step>& steps, bool skip_paired) {
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

}  // namespace cognn_eng

