"""CPU ORACLE, original-gcn variant (test infrastructure, NOT product code).

BASELINE.json configs[0]: "2-party original-gcn on Cora, CPU reference path with loopback comm_sync (smallest test,
no GPU)".  This module restates the unoptimised CoGNN kernel — aggregate-then-transform, 4 GAS iterations per epoch,
message width {input_dim, hidden_dim, num_labels, hidden_dim} — on top of the shared engine restatement in
cognn_oracle.py (preprocessing, OEP/OGA stand-ins, client/server schedule of ss_vertex_centric_algo_kernel.h).
Only tests/ import it.  The engine runs this variant as COGNN_VARIANT_ORIGINAL_GCN (bin/gcn-original, single process) and is
checked against this module after every GAS iteration (it exists in the reference as the CPU baseline of the paper's
"unoptimised" experiments, tools/tmp_run_cluster.py:285-286).

PARITY UNPINNED, as for the rest of the oracle: the fused external ops this variant calls
(sci::twoPartyGCNForwardNN / ForwardNNPrediction / BackwardNNInit / BackwardNN, the two-normaliser
twoPartyGCNVectorScale overload) are absent from the reference tree; their input/output relation is INFERRED from the
call sites cited below and built from the share-arithmetic definitions frozen in DESIGN.md §3.

What is restated line by line (all in algo_kernels/vertex_centric/original-gcn/gcn.h):
  PreScatterComp = copy                                   :198-209
  ScatterComp: per-edge scale by (outDeg_src+1)^-1/2 and (inDeg_dst+1)^-1/2          :211-251
    (who supplies which normaliser: ss_...h:800 client, :1041-1043 server)
  UpdatePreMergeComp = prefix_network_aggregate           :260-290
  GatherComp: forward only, self row *= (inDeg+1)^-1/2 once (updateSrcTid == 0), then masked add   :323-405
  ApplyComp forward NN / prediction / backward-init / backward + weight averaging    :426-713
  schedule constants (epoch = 4, widths, dims)            :802-851
"""
import numpy as np

import cognn_oracle as co
from cognn_oracle import U64

# dealer op ids used only by this variant (cognn_oracle.OP_* occupy 1..22)
(OP_SC_SCALE0, OP_SC_SCALE0_TRUNC, OP_SC_SCALE1, OP_SC_SCALE1_TRUNC,
 OP_AP_DGEMM, OP_AP_DGEMM_TRUNC, OP_AP_FWD_GEMM, OP_AP_FWD_GEMM_TRUNC) = range(30, 38)


def pair_tag(P, i):
    """Dealer 'owner' tag of the (client P, server i) Scatter instance: several run per owner and iteration."""
    return 0x10000 + P * 256 + i


class OriginalOracleEngine(co.OracleEngine):
    """Sequential emulation of k parties running gcn-original."""

    def __init__(self, k, src, dst, part, features, labels, param, seed=0xC06A11, weights=None):
        super().__init__(k, src, dst, part, features, labels, param, seed=seed, variant="original-gcn", weights=weights)

    # -- schedule constants (original-gcn/gcn.h:802-851) -------------------------------------
    def epoch_len(self):
        return 2 * self.param.num_layers                      # getEpochLayerNum :842-845

    def co_forward_layer(self, it):                           # :337-340, :431-434
        e = it % self.epoch_len(); f = self.fwd_layers()
        return e if e < f else f - 1 - (e - f)

    def mp_width(self, it):                                   # getPlainNumPerOperand(layer) :807-830
        g = self.param
        return [g.input_dim, g.hidden_dim, g.num_labels, g.hidden_dim][it % (2 * g.num_layers)]

    # -- PreScatterComp (:198-209): scaledVertexSvv = vertexSvv, then the usual re-replication ----
    def _prescatter_pair(self, P, it):
        c = self.co(P)
        xB = self.states[c].remoteVertexSvvs[P]
        for j in range(self.k):                               # ss_...h:997-1002 / :982
            if j != P and j != c:
                self.states[j].remoteVertexSvvs[P] = xB.copy()

    # -- ScatterComp (:211-251) for the pair (client P holding updA, server S holding updB) on edges P -> i ----
    def _scatter_pair(self, P, i, it, updA, updB):
        gsP = self.states[P]
        n0A = co.normalizer(gsP.updateSrcOutDeg[i])           # :228, degrees passed at ss_...h:800
        n1A = co.normalizer(gsP.updateDstInDeg[i])            # :229 (all zero for i != P, ss_...h:499)
        zero = np.zeros(len(n0A), dtype=U64)
        n0B = zero                                            # the server always passes zeroDeg as source degree (ss_...h:1041,1043)
        n1B = zero if i == P else co.normalizer(self.states[i].remoteUpdateDstInDeg[P])   # ss_...h:1043, :527-531
        tag = pair_tag(P, i)
        # sci::twoPartyGCNVectorScale(in, normalizer0, normalizer1, out) (:243-250), INFERRED: two successive
        # share x shared-row-scale products, each followed by truncation
        zA, zB = co.beaver_rowscale_pair(updA, updB, n0A, n0B, self.key_of(tag, it, OP_SC_SCALE0))
        xA, xB = co.trunc_pair(zA, zB, self.key_of(tag, it, OP_SC_SCALE0_TRUNC))
        zA, zB = co.beaver_rowscale_pair(xA, xB, n1A, n1B, self.key_of(tag, it, OP_SC_SCALE1))
        return co.trunc_pair(zA, zB, self.key_of(tag, it, OP_SC_SCALE1_TRUNC))

    def _message_passing_it(self, P, it):
        """ss_...h:748-856 (client) / :1005-1080 (server) with this variant's ScatterComp."""
        k = self.k; c = self.co(P); gsP = self.states[P]; gsC = self.states[c]
        lpos = gsP.localVertexPos
        # local edges: pair (P client, c server)
        updA = co.oep(lpos, gsP.updateSrcVertexPos[P], gsP.localVertexSvv)            # ss_...h:752
        updB = co.oep(lpos, gsP.updateSrcVertexPos[P], gsC.remoteVertexSvvs[P])       # ss_...h:1011
        dupA, dupB = self._scatter_pair(P, P, it, updA, updB)
        dupA = co.prefix_network_aggregate(gsP.updateDstVertexPos[P], dupA)           # :260-290
        dupB = co.prefix_network_aggregate(gsP.updateDstVertexPos[P], dupB)
        gsP.localUpdateSvvs[P] = co.oep(gsP.updateDstVertexPos[P], lpos, dupA)        # ss_...h:818
        gsC.remoteUpdateSvvs[c] = co.oep(gsP.updateDstVertexPos[P], lpos, dupB)       # ss_...h:1057
        for i in range(k):                                                            # edges P -> i
            if i == P:
                continue
            gsI = self.states[i]
            updA = co.oep(lpos, gsP.updateSrcVertexPos[i], gsP.localVertexSvv)        # ss_...h:760
            updB = co.oep(lpos, gsP.updateSrcVertexPos[i], gsI.remoteVertexSvvs[P])   # ss_...h:1016
            dupA, dupB = self._scatter_pair(P, i, it, updA, updB)
            gsP.remoteUpdateSvvs[i] = co.prefix_network_aggregate(gsP.updateDstVertexPos[i], dupA)   # ss_...h:827,835
            gsI.localUpdateSvvs[P] = co.prefix_network_aggregate(gsP.updateDstVertexPos[i], dupB)    # ss_...h:1063,1067

    # -- GatherComp over all source parties (:323-405; ss_...h:866-880, 1087-1135) ----------------
    def _gather_pair(self, P, it):
        k = self.k; c = self.co(P); gsP = self.states[P]; gsC = self.states[c]
        vA = gsP.localVertexSvv; vB = gsC.remoteVertexSvvs[P]
        remote = [None] * k                                   # ss_...h:1089-1100
        remote[c] = gsC.remoteUpdateSvvs[P] if c != P else None
        remote[P] = gsC.remoteUpdateSvvs[c]
        for j in range(k):
            if j != c and j != P:
                remote[j] = self.states[j].remoteUpdateSvvs[P]
        fwd = (it % self.epoch_len()) < self.fwd_layers()
        with np.errstate(over="ignore"):
            for j in range(k):
                if fwd and j == 0:                            # :365-381: the self row is scaled once, before the first addition
                    sA = co.normalizer(gsP.localVertexInDeg); sB = np.zeros(len(sA), dtype=U64)   # ss_...h:874 vs :1125
                    zA, zB = co.beaver_rowscale_pair(vA, vB, sA, sB, self.key_of(P, it, co.OP_GA_SCALE))
                    vA, vB = co.trunc_pair(zA, zB, self.key_of(P, it, co.OP_GA_SCALE_TRUNC))
                cond = ~np.array(gsP.isGatherDstVertexDummy[j], dtype=bool)   # :386-387 (owner's mask)
                vA = vA + np.where(cond[:, None], gsP.localUpdateSvvs[j], U64(0))
                vB = vB + np.where(cond[:, None], remote[j], U64(0))
        gsP.localVertexSvv = vA; gsC.remoteVertexSvvs[P] = vB

    # -- ApplyComp (:426-713), both roles -----------------------------------------------------
    def _gemm_trunc(self, P, it, xA, xB, wA, wB, op, top):
        zA, zB = co.beaver_gemm_pair(xA, xB, wA, wB, self.key_of(P, it, op))
        return co.trunc_pair(zA, zB, self.key_of(P, it, top))

    def _apply_pair(self, P, it):
        g = self.param; c = self.co(P); gsP = self.states[P]; gsC = self.states[c]
        e = it % self.epoch_len(); f = self.fwd_layers(); fwd = e < f; layer = self.co_forward_layer(it)
        inA = gsP.localVertexSvv; inB = gsC.remoteVertexSvvs[P]
        LI = gsP.localInter[layer]; RI = gsC.remoteInter[layer]
        n = inA.shape[0]
        train = int(n * g.train_ratio); val = int(n * g.val_ratio)
        WA = gsP.localWeight[layer]; WB = gsC.remoteWeight[layer]
        if fwd:
            LI["ah_t"] = inA.copy(); RI["ah_t"] = inB.copy()  # :452 (stored untransposed; used as its transpose below)
            zA, zB = self._gemm_trunc(P, it, inA, inB, WA, WB, OP_AP_FWD_GEMM, OP_AP_FWD_GEMM_TRUNC)
            LI["z"] = zA; RI["z"] = zB
            if e != f - 1:                                    # twoPartyGCNForwardNN :459 — INFERRED z = in.W, new_h = ReLU(z)
                outA, outB, pos = co.relu_pair(zA, zB, self.key_of(P, it, co.OP_AP_RELU))
                LI["relu_mask"] = pos; RI["relu_mask"] = pos
            else:                                             # twoPartyGCNForwardNNPrediction :493,508 + reveal :523
                p0, p1, outA, outB, plainP = co.softmax_pair(zA, zB, gsP.labels, train, self.key_of(P, it, co.OP_AP_SOFTMAX))
                LI["p"] = p0; RI["p"] = p1
                self.metrics.append(self._metrics(P, it, plainP, train, val))
        else:
            WTA = WA.T.copy(); WTB = WB.T.copy()              # :565-568 (taken before the update)
            ahA = LI["ah_t"].T.copy(); ahB = RI["ah_t"].T.copy()
            if layer == f - 1:                                # twoPartyGCNBackwardNNInit :586 — INFERRED d = ah_t.in, g = in.W^T
                gzA, gzB = inA, inB
                first = False
            else:                                             # twoPartyGCNBackwardNN :622 — INFERRED in (.) 1[z>0] first
                pos = LI["relu_mask"]
                gzA = np.where(pos, inA, U64(0)); gzB = np.where(pos, inB, U64(0))
                first = (layer == 0)                          # :620-621
            dA, dB = self._gemm_trunc(P, it, ahA, ahB, gzA, gzB, OP_AP_DGEMM, OP_AP_DGEMM_TRUNC)
            if first:
                outA = np.zeros((n, 0), dtype=U64); outB = np.zeros((n, 0), dtype=U64)   # g skipped for the first layer
            else:
                outA, outB = self._gemm_trunc(P, it, gzA, gzB, WTA, WTB, co.OP_AP_GEMM, co.OP_AP_GEMM_TRUNC)
            gscale = co.fx_encode_trunc(1.0 / train) if train > 0 else U64(0)          # :588-591, :632-635
            dA, dB = co.const_scale_trunc_pair(dA, dB, gscale, self.key_of(P, it, co.OP_AP_GSCALE_TRUNC))
            lr = co.fx_encode_trunc(g.learning_rate)                                   # :593, :642
            uA, uB = co.const_scale_trunc_pair(dA, dB, lr, self.key_of(P, it, co.OP_AP_LR_TRUNC))
            with np.errstate(over="ignore"):
                gsP.localWeight[layer] = WA - uA
                gsC.remoteWeight[layer] = WB - uB
            LI["d"] = dA; RI["d"] = dB
        gsP.localVertexSvv = outA; gsC.remoteVertexSvvs[P] = outB
        for j in range(self.k):                               # ss_...h:967-972, 1153-1158
            if j != P and j != c:
                self.states[j].remoteVertexSvvs[P] = outB.copy()

    def _weight_average(self, it):
        """:659-711 — the same topology as optimize-gcn (parties 0/1 sum, scale by 1/k, redistribute), after EVERY backward Apply."""
        self.variant = "optimize-gcn"                         # select the scaled branch of the shared implementation (:675-676)
        try:
            super()._weight_average(it)
        finally:
            self.variant = "original-gcn"

    # -- one GAS iteration for all parties ------------------------------------------------------
    def iteration(self, it):
        k = self.k; ep = self.epoch_len(); f = self.fwd_layers(); e = it % ep
        if e == 0:                                            # ss_...h:695, 938
            for P in range(k):
                gs = self.states[P]
                gs.localVertexSvv = gs.localVertexSvvBackup.copy()
                gs.remoteVertexSvvs = [None if x is None else x.copy() for x in gs.remoteVertexSvvsBackup]
        if e != 0 and e % f == 0:                             # apply-only, ss_...h:709-732, 941-979
            for P in range(k):
                self._apply_pair(P, it)
            self._weight_average(it)
            return
        for P in range(k):
            self._prescatter_pair(P, it)
        for P in range(k):
            self._message_passing_it(P, it)
        for P in range(k):
            self._extend_updates(P)
        for P in range(k):
            self._gather_pair(P, it)
        for P in range(k):
            self._apply_pair(P, it)
        if e >= f:
            self._weight_average(it)


class OriginalPlainEngine(co.PlainEngine):
    """float64 plaintext of the same schedule (global view), for reconstruction tests."""

    def _norms(self):
        if getattr(self, "_edge_norm", None) is None:
            import scipy.sparse as sp
            o = self.o
            off = np.cumsum([0] + [len(o.states[P].localVertexPos) for P in range(self.k)])
            rows, cols, vals = [], [], []

            def nz(d):
                return 0.0 if d == 0 else (d + 1.0) ** -0.5
            for P in range(self.k):
                gs = o.states[P]
                for i in range(self.k):
                    din = gs.updateDstInDeg[i] if i == P else o.states[i].remoteUpdateDstInDeg[P]
                    for q, (s, d) in enumerate(zip(gs.updateSrcVertexPos[i], gs.updateDstVertexPos[i])):
                        if i == P and s == d and gs.isGatherDstVertexDummy[P][gs.reorderedIndex[d]]:
                            continue
                        (ps, rs), (pd, rd) = self.vid_row[s], self.vid_row[d]
                        rows.append(off[pd] + rd); cols.append(off[ps] + rs)
                        vals.append(nz(gs.updateSrcOutDeg[i][q]) * nz(din[q]))
            n = int(off[-1])
            self._edge_norm = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
            self._off2 = off
        return self._edge_norm, self._off2

    def iteration(self, it):
        o = self.o; g = o.param; ep = o.epoch_len(); f = o.fwd_layers(); e = it % ep
        layer = o.co_forward_layer(it); fwd = e < f
        if e == 0:
            self.X = [x.copy() for x in self.X0]
        apply_only = (e != 0 and e % f == 0)
        if not apply_only:
            A, off = self._norms()
            allx = np.vstack(self.X)
            selfx = allx.copy()
            if fwd:
                selfx = np.vstack([self.X[P] * self._s(P)[:, None] for P in range(self.k)])
            allx = selfx + A @ allx
            self.X = [allx[off[P]:off[P + 1]] for P in range(self.k)]
        for P in range(self.k):
            I = self.inter[P][layer]; x = self.X[P]; n = x.shape[0]
            train = int(n * g.train_ratio)
            if fwd:
                I["ah"] = x.copy()
                z = x @ self.W[P][layer]
                I["z"] = z
                if e != f - 1:
                    self.X[P] = np.maximum(z, 0)
                else:
                    m = z.max(axis=1, keepdims=True)
                    ex = np.exp(z - m); p = ex / ex.sum(axis=1, keepdims=True)
                    y = np.zeros_like(p); y[np.arange(n), o.states[P].labels] = 1.0
                    d = p - y; d[train:] = 0
                    I["p"] = p; self.X[P] = d
                    self.metrics.append(o._metrics(P, it, p, train, int(n * g.val_ratio)))   # what the client prints (gcn.h:511-523)
            else:
                gz = x if layer == f - 1 else x * (I["z"] > 0)
                d = I["ah"].T @ gz
                out = gz @ self.W[P][layer].T if layer != 0 else np.zeros((n, 0))
                d = d * (1.0 / train if train > 0 else 0.0)
                self.W[P][layer] = self.W[P][layer] - g.learning_rate * d
                self.X[P] = out
        if e >= f:
            avg = sum(self.W[P][layer] for P in range(self.k)) / self.k
            for P in range(self.k):
                self.W[P][layer] = avg.copy()
