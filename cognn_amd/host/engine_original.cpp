// Host engine, the unoptimised kernel (original-gcn/gcn.h, BASELINE config 1): edge index, fused Scatter + Gather, its iteration.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"
#include "engine_stages.h"

namespace cognn_eng {


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

}  // namespace cognn_eng

