// Host engine, the unoptimised kernel (original-gcn/gcn.h, BASELINE config 1): edge index, fused Scatter + Gather, its iteration.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"
#include "engine_stages.h"

namespace cognn_eng {


// ---------------------------------------------------------------------------------------------
// original-gcn (algo_kernels/vertex_centric/original-gcn/gcn.h; BASELINE config 1) - all parties in one process
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

// ---------------------------------------------------------------------------------------------
// original-gcn across ranks (party placement; the reference runs it as k processes, original-gcn/gcn.h:224-251, ss_...h:748-856 /
// :1005-1080).  The two roles of a Scatter instance (P -> g) - client P with its own share of P's rows, server with the co-share:
// the co-party of P for the local edges, party g with its replica (ss_...h:997-1002) otherwise - may sit on different ranks, so the
// per-edge scales run as per-side launches with their openings exchanged: every role hosted here is a row range of one per-edge
// tensor [edges x F]; one gather fills it from the share table, the two Beaver row scales (+ truncations) run range by range
// (batched launches, one exchange round per opening), one gather adds the scaled rows to the hosted destination rows - the
// client's result share of a remote instance crosses to the co-party side of the destination, the server's to its owner side
// (ss_...h:1063-1100) - and the client results for destination owners whose co-party lives elsewhere travel pre-summed per rank.
// ---------------------------------------------------------------------------------------------
void build_original_ranks(cognn_engine* E) {
    const int k = E->k;
    auto& G = E->G;
    if (E->cfg.placement != COGNN_PLACE_PARTY) throw EngineError("cognn_engine_create: original-gcn across ranks runs in the party placement");
    auto norm = [](uint32_t deg) { return deg == 0 ? (u64)0 : fx_llround(std::pow((double)deg + 1.0, -0.5)); };
    auto nrows = [&](int o) { return (int64_t)G.party[o].localVertexPos.size(); };
    E->orig_tab_rows = 0;
    for (int o = 0; o < k; ++o) {
        if (E->A_off[o] >= 0) E->orig_tab_rows = std::max(E->orig_tab_rows, E->A_off[o] + nrows(o));
        E->orig_tab_rows = std::max(E->orig_tab_rows, E->B_off[o] + nrows(o));
    }
    E->orig_roles.clear();
    std::vector<uint32_t> src_col;
    // destination rows of this rank (the aggregate's layout): entries = scaled edge rows
    std::vector<std::vector<uint32_t>> agg((size_t)E->aggRows);
    std::map<int, std::vector<std::vector<uint32_t>>> out_rows;     // destination owner g (co-party elsewhere) -> per vertex of g: edge rows
    for (int P = 0; P < k; ++P)
        for (int g = 0; g < k; ++g) {
            const int server_party = g == P ? E->co(P) : g;
            const int rc = E->rank_of(P), rs = E->rank_of(server_party);
            for (int p = 0; p < 2; ++p) {
                if ((p == 0 ? rc : rs) != E->rank) continue;
                const cognn::EdgeBlock& blk = G.party[P].out[g];
                cognn_engine::OrigRole R;
                if (src_col.size() & 1) src_col.push_back(0);              // ranges start on even rows: the kernels move 16-byte element pairs
                R.P = P; R.g = g; R.p = p; R.peer_rank = p == 0 ? rs : rc; R.row0 = (int64_t)src_col.size();
                std::vector<u64> s0, s1;
                const int64_t tab0 = p == 0 ? E->A_off[P] : E->B_off[P];   // the rows this role reads: P's own share / the co-share (replica)
                for (size_t r = 0; r + 1 < blk.rowptr.size(); ++r) {
                    const uint32_t dr = G.row_of_vid[blk.rows_vid[r]];
                    if (g == P && blk.rowptr[r + 1] == blk.rowptr[r]) {  // the dummy self entry keeps its position in the list and contributes nothing
                        src_col.push_back((uint32_t)(tab0 + dr)); s0.push_back(0); s1.push_back(0);
                        continue;
                    }
                    for (uint32_t e = blk.rowptr[r]; e < blk.rowptr[r + 1]; ++e) {
                        const uint32_t erow = (uint32_t)src_col.size();
                        src_col.push_back((uint32_t)(tab0 + blk.col[e]));
                        const u64 n0 = norm(G.party[P].outDeg[blk.col[e]]), n1 = norm(G.party[g].inDeg[dr]);
                        // the client supplies the source normaliser and, for its local edges, the destination normaliser; the server the
                        // destination normaliser of remote edges (gcn.h:228-229, ss_...h:800,1041-1043); the other party's share is 0
                        s0.push_back(p == 0 ? n0 : 0);
                        s1.push_back(g == P ? (p == 0 ? n1 : 0) : (p == 0 ? 0 : n1));
                        // where the role's result share of this edge goes: local edges stay on their side, remote ones cross
                        const bool to_owner_side = (g == P) ? (p == 0) : (p == 1);
                        if (to_owner_side) agg[(size_t)(E->A_off[g] + dr)].push_back(erow);            // (the owner side of g is hosted wherever this role runs)
                        else if (E->holder(g, 1) == E->rank) agg[(size_t)(E->B_off[g] + dr)].push_back(erow);
                        else {
                            auto& v = out_rows[g];
                            if (v.empty()) v.resize((size_t)nrows(g));
                            v[dr].push_back(erow);
                        }
                    }
                }
                R.edges = (int64_t)src_col.size() - R.row0;
                R.s[0] = upload(E, s0); R.s[1] = upload(E, s1);
                E->orig_roles.push_back(R);
            }
        }
    for (size_t a = 0; a < E->orig_roles.size(); ++a)           // both roles of an instance on this rank: they read each other's openings in place
        for (size_t b = 0; b < E->orig_roles.size(); ++b)
            if (a != b && E->orig_roles[a].P == E->orig_roles[b].P && E->orig_roles[a].g == E->orig_roles[b].g) E->orig_roles[a].peer_role = (int)b;
    E->orig_rows = (int64_t)src_col.size();
    std::vector<uint32_t> idp((size_t)E->orig_rows + 1);
    for (size_t i = 0; i < idp.size(); ++i) idp[i] = (uint32_t)i;
    E->orig_id_rowptr = upload(E, idp); E->orig_src_col = upload(E, src_col);
    auto flatten = [&](const std::vector<std::vector<uint32_t>>& rows, std::vector<uint32_t>& rp, std::vector<uint32_t>& col) {
        for (auto& r : rows) { col.insert(col.end(), r.begin(), r.end()); rp.push_back((uint32_t)col.size()); }
    };
    {
        std::vector<uint32_t> rp{0}, col;
        flatten(agg, rp, col);
        E->orig_agg_rowptr = upload(E, rp); E->orig_agg_col = upload(E, col);
    }
    // outgoing client results, one block per destination owner (receiver: rank of its co-party), in ascending owner order on both ends
    E->orig_send.clear(); E->orig_recv.clear();
    {
        std::vector<uint32_t> rp{0}, col;
        int64_t off = 0;
        for (auto& kv : out_rows) {
            cognn_engine::OrigBlock b;
            b.g = kv.first; b.rank = E->holder(kv.first, 1); b.rows = nrows(kv.first); b.off = off;
            flatten(kv.second, rp, col);
            off += b.rows;
            E->orig_send.push_back(b);
        }
        E->orig_out_rows = off;
        E->orig_out_rowptr = upload(E, rp); E->orig_out_col = upload(E, col);
    }
    {   // incoming: for every co-hosted owner g, one block from every other rank that hosts a party with edges into g
        int64_t off = 0;
        for (int r = 0; r < E->world; ++r) {
            if (r == E->rank) continue;
            std::vector<int> owners(E->cohosted.begin(), E->cohosted.end());
            std::sort(owners.begin(), owners.end());
            for (int g : owners) {
                bool any = false;
                for (int P = r * E->m; P < (r + 1) * E->m; ++P)
                    if (P != g) for (size_t q = 0; q + 1 < G.party[P].out[g].rowptr.size(); ++q) any = any || G.party[P].out[g].rowptr[q + 1] > G.party[P].out[g].rowptr[q];
                if (!any) continue;
                cognn_engine::OrigBlock b;
                b.g = g; b.rank = r; b.rows = nrows(g); b.off = off;
                off += b.rows + (b.rows & 1);                        // (16-byte aligned blocks for the add)
                E->orig_recv.push_back(b);
            }
        }
        const int fm = std::max(E->in(), std::max(E->hid(), E->lab()));
        E->orig_inb = dalloc<u64>(E, (size_t)std::max<int64_t>(off, 1) * fm);
        E->orig_out = dalloc<u64>(E, (size_t)std::max<int64_t>(E->orig_out_rows, 1) * fm);
        E->orig_tab = dalloc<u64>(E, (size_t)E->orig_tab_rows * fm);
        E->orig_acc = dalloc<u64>(E, (size_t)std::max<int64_t>(E->aggRows, 1) * fm);
        E->orig_edge = dalloc<u64>(E, (size_t)std::max<int64_t>(E->orig_rows, 1) * fm);
        for (int j = 0; j < 4; ++j) E->orig_w[j] = dalloc<u64>(E, (size_t)std::max<int64_t>(E->orig_rows, 1) * fm);
        for (int j = 0; j < 2; ++j) E->orig_g[j] = dalloc<u64>(E, (size_t)std::max<int64_t>(E->orig_rows, 1) + 2);
    }
    // a sender lists a block only when it has edges into g: the receiver's `any` test above is the same predicate
}

// One message-passing round of the unoptimised kernel across ranks: every side's tensor src(side) [n x F] -> dst(side)
template <class SrcFn, class DstFn>
void original_message_passing_ranks(cognn_engine* E, int64_t it, int F, bool fwd, SrcFn src, DstFn dst) {
    auto seg = [&](u64* base, Side& s) { return base + (s.p == 0 ? E->A_off[s.owner] : E->B_off[s.owner]) * (int64_t)F; };
    // the share table of the round: every hosted side's tensor, the co-shares replicated to the ranks whose parties serve remote edges
    for (auto& s : E->sides) BE(cognn_memcpy_d2d(E->ctx, seg(E->orig_tab, s), src(s), (size_t)s.n * F * 8));
    {
        XList xl;
        for (int o = 0; o < E->k; ++o) {
            const int rc = E->holder(o, 1);
            const int64_t bytes = (int64_t)E->G.party[o].localVertexPos.size() * F * 8;
            u64* sg = E->orig_tab + E->B_off[o] * (int64_t)F;
            if (rc == E->rank) {
                for (int r = 0; r < E->world; ++r) {
                    if (r == E->rank || (E->m == 1 && r == E->rank_of(o))) continue;   // (that rank hosts only the owner itself)
                    xl.send(r, sg, bytes);
                }
            } else if (!(E->m == 1 && E->rank == E->rank_of(o))) xl.recv(rc, sg, bytes);
        }
        run_exchange(E, xl);                                // in flight during the self-row scale below
    }
    // the accumulators start from the self rows - scaled by the owner-known normaliser in a forward iteration (:365-381)
    if (fwd)
        rowscale_stage(E, it, COGNN_OP_GA_SCALE, COGNN_OP_GA_SCALE_TRUNC, F, [&](Side& s) { return (u64*)src(s); }, [&](Side& s) { return seg(E->orig_acc, s); });
    else
        for (auto& s : E->sides) BE(cognn_memcpy_d2d(E->ctx, seg(E->orig_acc, s), src(s), (size_t)s.n * F * 8));
    exchange_wait(E);
    if (E->orig_rows > 0) BE(cognn_gather_csr_u64(E->ctx, E->orig_edge, nullptr, E->orig_tab, E->orig_id_rowptr, E->orig_src_col, E->orig_rows, F));
    // the two per-edge scales: share x shared normaliser (Beaver, one triple per element, b per edge) + truncation, range by range
    for (int t = 0; t < 2; ++t) {
        const int op = t == 0 ? COGNN_OP_SC_SCALE0 : COGNN_OP_SC_SCALE1, top = t == 0 ? COGNN_OP_SC_SCALE0_TRUNC : COGNN_OP_SC_SCALE1_TRUNC;
        auto kof = [&](const cognn_engine::OrigRole& R, int o) { return keys(E, scatter_tag(R.P, R.g), it, o); };
        auto peer_ptr = [&](const cognn_engine::OrigRole& R, u64* own_out, u64* own_in, int64_t stride) {
            return R.peer_role >= 0 ? own_out + E->orig_roles[(size_t)R.peer_role].row0 * stride : own_in + R.row0 * stride;
        };
        {   // E_p = V_p - a_p, G_p = s_p - b_p
            Batch batch(E);
            for (auto& R : E->orig_roles) {
                if (!R.edges) continue;
                cognn_keys kk = kof(R, op);
                BE(cognn_rowscale_open_u64(E->ctx, E->orig_w[0] + R.row0 * F, E->orig_g[0] + R.row0, E->orig_edge + R.row0 * F, R.s[t], &kk, R.p, R.edges, F));
            }
        }
        {
            XList xl;
            for (auto& R : E->orig_roles) {
                if (!R.edges || R.peer_role >= 0) continue;
                xl.send(R.peer_rank, E->orig_w[0] + R.row0 * F, R.edges * F * 8); xl.recv(R.peer_rank, E->orig_w[1] + R.row0 * F, R.edges * F * 8);
                xl.send(R.peer_rank, E->orig_g[0] + R.row0, R.edges * 8); xl.recv(R.peer_rank, E->orig_g[1] + R.row0, R.edges * 8);
            }
            run_exchange_sync(E, xl);
        }
        {   // the product share and the opening of its truncation
            Batch batch(E);
            for (auto& R : E->orig_roles) {
                if (!R.edges) continue;
                cognn_keys kk = kof(R, op), tk = kof(R, top);
                BE(cognn_rowscale_close_u64(E->ctx, E->orig_w[2] + R.row0 * F, E->orig_w[0] + R.row0 * F, peer_ptr(R, E->orig_w[0], E->orig_w[1], F),
                                            E->orig_g[0] + R.row0, peer_ptr(R, E->orig_g[0], E->orig_g[1], 1), &kk, &tk, R.p, R.edges, F));
            }
        }
        {
            XList xl;
            for (auto& R : E->orig_roles) {
                if (!R.edges || R.peer_role >= 0) continue;
                if (R.p == 1) xl.send(R.peer_rank, E->orig_w[2] + R.row0 * F, R.edges * F * 8);   // (the client closes with both, the server with its dealt share alone)
                else xl.recv(R.peer_rank, E->orig_w[3] + R.row0 * F, R.edges * F * 8);
            }
            run_exchange_sync(E, xl);
        }
        {
            Batch batch(E);
            for (auto& R : E->orig_roles) {
                if (!R.edges) continue;
                cognn_keys tk = kof(R, top);
                const u64* c0 = R.p == 0 ? E->orig_w[2] + R.row0 * F : nullptr;
                const u64* c1 = R.p == 0 ? peer_ptr(R, E->orig_w[2], E->orig_w[3], F) : nullptr;
                BE(cognn_trunc_close_u64(E->ctx, E->orig_edge + R.row0 * F, c0, c1, &tk, R.p, 0, R.edges * F));
            }
        }
    }
    // UpdatePreMergeComp + GatherComp: the scaled rows of every edge into a hosted destination row, and the client results for
    // destination owners whose co-party lives on another rank - pre-summed over this rank's parties - on their way
    if (E->orig_out_rows > 0) BE(cognn_gather_csr_u64(E->ctx, E->orig_out, nullptr, E->orig_edge, E->orig_out_rowptr, E->orig_out_col, E->orig_out_rows, F));
    {
        XList xl;
        for (auto& b : E->orig_send) xl.send(b.rank, E->orig_out + b.off * F, b.rows * F * 8);
        for (auto& b : E->orig_recv) xl.recv(b.rank, E->orig_inb + b.off * F, b.rows * F * 8);
        run_exchange(E, xl);
    }
    if (E->aggRows > 0) BE(cognn_gather_csr_u64(E->ctx, E->orig_acc, E->orig_acc, E->orig_edge, E->orig_agg_rowptr, E->orig_agg_col, E->aggRows, F));
    exchange_wait(E);
    for (auto& b : E->orig_recv) {
        u64* a = E->orig_acc + E->B_off[b.g] * (int64_t)F;
        BE(cognn_add_u64(E->ctx, a, a, E->orig_inb + b.off * F, b.rows * F));
    }
    for (auto& s : E->sides) BE(cognn_memcpy_d2d(E->ctx, dst(s), seg(E->orig_acc, s), (size_t)s.n * F * 8));
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
        if (E->world > 1) original_message_passing_ranks(E, it, F, fwd, [&](Side& s) { return s.cur; }, out);
        else original_message_passing(E, it, F, fwd, [&](Side& s) { return s.cur; }, out);
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

}  // namespace cognn_eng

