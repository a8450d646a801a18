// Host engine, the online schedule: message passing and one GAS iteration of optimize-gcn (onIteration / runAlgoKernelServer, ss_...h:680-1189), recorded epochs.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"
#include "engine_stages.h"

namespace cognn_eng {


// ---------------------------------------------------------------------------------------------
// message passing: Scatter + PreMerge + Gather fused into two CSR launches over the share table
// ---------------------------------------------------------------------------------------------
u64* table_seg(cognn_engine* E, Side& s, int F) {
    const int64_t off = s.p == 0 ? E->A_off[s.owner] : E->B_off[s.owner];
    return E->table + off * F;
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
                           bool softmax_follows, const u64* table) {
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

}  // namespace cognn_eng

