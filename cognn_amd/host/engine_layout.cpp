// Host engine, layout of a rank: hosted sides, share table, CSRs (host builder and device builder), buffers.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"

namespace cognn_eng {


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

}  // namespace cognn_eng

