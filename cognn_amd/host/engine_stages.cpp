// Host engine, element-wise two-party stages (ReLU, prediction layer), weight update and weight averaging.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"

namespace cognn_eng {


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
        steps.back().msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(E, xl, s, s.ob[0], s.ib[0], eF[i], c, C); };
    }
    steps.emplace_back();
    steps.back().fn = [&](Side& s, size_t i) {
        cognn_keys k = keys(E, s.owner, it, COGNN_OP_AP_RELU);
        BE(cognn_relu_mul_u64(E->ctx, s.ob[2], s.ob[0], e_public ? nullptr : s.ib[0], nullptr, nullptr, &k, s.p, eF[i]));
    };
    steps.back().msg = [&](XList& xl, Side& s, size_t i, int c, int C) { msg_range(E, xl, s, s.ob[2], s.ib[2], eF[i], c, C, true); };   // the opened product w: its top 48 bits carry the sign
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

// pairs_fused: the co-located pairs' products are still in zbuf (gemm_stage, pairs_raw): product truncation, both scales and the
// update run as one pass per pair (cognn_pair_weight_update_u64) - and, when every party's pair is hosted here, the weight
// average too (returns true: weight_average has been done)
bool weight_update_chain(cognn_engine* E, int64_t it, int layer, bool pairs_fused, bool raw,
                         const std::function<GemmSpec(Side&)>& specfn) {
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

}  // namespace cognn_eng

