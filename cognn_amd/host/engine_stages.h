// The two big templated stages of a GAS iteration - the Beaver product stage and the row-scale stage - shared by the online schedule
// (engine_schedule.cpp) and the original-gcn schedule (engine_original.cpp).
#pragma once
#include "engine_internal.h"

namespace cognn_eng {


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
        if (!e_public) msg_range(E, xl, s, e_opened ? X(s) : s.ob[0], s.ib[0], eF[i], c, C);   // (public: ob[0] holds E itself, only the scale openings travel)
        if (c == 0) msg_range(E, xl, s, s.ob[1], s.ib[1], e1[i], 0, 1);
    };
    steps[1].fn = [&](Side& s, size_t) {                    // the opened sums E0+E1, G0+G1 are formed inside the kernel
        cognn_keys k = keys(E, s.owner, it, op), tk = keys(E, s.owner, it, top);
        const u64* e_own = e_opened ? X(s) : s.ob[0];
        const u64* e_peer = e_public ? nullptr : e_opened ? (s.peer ? X(*s.peer) : s.ib[0]) : s.ib[0];
        BE(cognn_rowscale_close_u64(E->ctx, s.ob[2], e_own, e_peer, s.ob[1], s.ib[1], &k, &tk, s.p, s.n, F));
    };
    steps[1].msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(E, xl, s, s.ob[2], s.ib[2], eF[i], c, C, true); };
    steps.emplace_back();
    steps.back().fn = [&](Side& s, size_t i) { trunc_close_one(E, it, top, dst, eF, open_next, s, i); };
    chunked_rounds(E, steps, true);
}

}  // namespace cognn_eng
