// Test program of the drop-in shim (include/cognn_sci_shim.hpp): two parties in one process run GAS iterations 0 and 1 of
// gcn-optimize written with the reference's call shapes - the callbacks of algo_kernels/vertex_centric/optimize-gcn/gcn.h
// (PreScatterComp :198-255, ScatterComp :257-307, UpdatePreMergeComp :309-342, GatherComp :375-494, ApplyComp forward
// :515-643) driven by the client / server thread structure of include/ss_vertex_centric_algo_kernel.h:680-910, 912-1189 -
// against the sci:: / oblivious-mapper / prefix_network_aggregate names, which the shim forwards to the HIP library.
// tests/test_shim_gpu.py writes the inputs (index arrays from the oracle's preprocessing, initial shares) and compares the
// outputs with the oracle bit for bit.   usage: shim_iteration <input file> <output file>
#include <cstdio>
#include <functional>
#include <thread>

#include "../include/cognn_sci_shim.hpp"

namespace {

struct Reader {
    FILE* f;
    std::vector<uint64_t> vec() {
        uint64_t n = 0;
        if (fread(&n, 8, 1, f) != 1) throw std::runtime_error("short input");
        std::vector<uint64_t> v(n);
        if (n && fread(v.data(), 8, n, f) != n) throw std::runtime_error("short input");
        return v;
    }
    uint64_t one() { return vec().at(0); }
    ShareVecVec mat() {
        const uint64_t r = one(), c = one();
        std::vector<uint64_t> flat = vec();
        ShareVecVec m(r, ShareVec(c));
        for (uint64_t i = 0; i < r; ++i)
            for (uint64_t j = 0; j < c; ++j) m[i][j] = flat[i * c + j];
        return m;
    }
};
void put(FILE* f, const ShareVecVec& m) {
    const uint64_t r = m.size(), c = r ? m[0].size() : 0;
    fwrite(&r, 8, 1, f); fwrite(&c, 8, 1, f);
    for (auto& row : m) fwrite(row.data(), 8, c, f);
}

struct Semaphore {
    std::mutex m; std::condition_variable cv; int n = 0;
    void release() { std::lock_guard<std::mutex> lk(m); ++n; cv.notify_one(); }
    void acquire() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return n > 0; }); --n; }
};

// the per-party state the reference keeps in GraphSummary (ss_...h:24-58), for two parties
struct GraphSummary {
    uint64_t tileIndex = 0;
    std::vector<uint64_t> localVertexPos, localVertexInDeg, labels;
    std::vector<std::vector<uint64_t>> updateSrcVertexPos, updateDstVertexPos, remoteMirrorVertexPos;
    std::vector<std::vector<bool>> isGatherDstVertexDummy;
    ShareVecVec localVertexSvv;
    std::vector<ShareVecVec> remoteVertexSvvs, localUpdateSvvs, remoteUpdateSvvs;
    std::vector<ShareTensor> localWeight, remoteWeight;
    uint64_t trainSetSize = 0;
    Semaphore localUpdateReady, remoteUpdateReady;
    DoubleTensor plainP;
};
const uint64_t tileNum = 2;
const uint32_t forwardLayerNum = 2, epochLayerNum = 6;

std::vector<uint64_t> normalizerOf(const std::vector<uint64_t>& deg) {           // gcn.h:219-221, 471-474
    std::vector<uint64_t> n(deg.size());
    for (size_t i = 0; i < deg.size(); ++i) n[i] = deg[i] == 0 ? 0 : CryptoUtil::encodeDoubleAsFixedPoint(std::pow((double)deg[i] + 1, -0.5));
    return n;
}

// gcn.h:198-255
void PreScatterComp(GraphSummary& gs, const ShareVecVec& vertexSvv, std::vector<uint64_t>& vertexOutDeg, ShareVecVec& scaledVertexSvv,
                    uint64_t iter, uint64_t coTid, int party) {
    const bool isForward = (iter % epochLayerNum) < forwardLayerNum;
    const uint32_t coForwardLayer = (uint32_t)(iter % epochLayerNum);
    std::vector<uint64_t> normalizer = normalizerOf(vertexOutDeg);
    ShareVecVec cur = vertexSvv;
    if (isForward) {
        const ShareTensor& weight = party == sci::ALICE ? gs.localWeight[coForwardLayer] : gs.remoteWeight[coForwardLayer];
        ShareVecVec product;
        sci::twoPartyGCNMatMul(cur, weight, product, coTid, party);
        cur.swap(product);
    }
    if (iter % epochLayerNum != 0) {
        ShareVecVec scaled;
        sci::twoPartyGCNVectorScale(cur, normalizer, scaled, true, coTid, party);
        cur.swap(scaled);
    }
    scaledVertexSvv.swap(cur);
}
// gcn.h:257-307 (CoGNN-Opt: a copy)
void ScatterComp(ShareVecVec& updateSrcSvv, ShareVecVec& duplicatedUpdateSvv) { duplicatedUpdateSvv = updateSrcSvv; }
// gcn.h:309-342
void UpdatePreMergeComp(ShareVecVec& duplicatedUpdateSvv, std::vector<uint64_t>& updateDstVertexPos, uint64_t coTid, int party) {
    duplicatedUpdateSvv = prefix_network_aggregate(updateDstVertexPos, duplicatedUpdateSvv, AggregationOp::ADD_AGG, coTid, party, true);
}
// gcn.h:375-494
void GatherComp(ShareVecVec& vertexSvv, ShareVecVec& updateSvv, std::vector<bool>& isGatherDstVertexDummy, std::vector<uint64_t>& localVertexInDeg,
                uint64_t iter, uint64_t updateSrcTid, uint64_t coTid, int party) {
    std::vector<bool> cond = isGatherDstVertexDummy;
    for (size_t i = 0; i < cond.size(); ++i) cond[i] = !cond[i];
    sci::twoPartyGCNCondVectorAddition(vertexSvv, updateSvv, cond, vertexSvv, coTid, party);
    if (updateSrcTid == tileNum - 1 && (iter + 1) % epochLayerNum != 0) {
        std::vector<uint64_t> normalizer = normalizerOf(localVertexInDeg);
        sci::twoPartyGCNVectorScale(vertexSvv, normalizer, vertexSvv, true, coTid, party);
    }
}
// gcn.h:515-643, forward branch
void ApplyComp(GraphSummary& gs, uint64_t iter, const ShareVecVec& vertexDataVec, ShareVecVec& dstVec, uint64_t dstTid, bool isClient) {
    const int party = isClient ? sci::ALICE : sci::BOB;
    const size_t vecSize = vertexDataVec.size();
    if (iter % epochLayerNum != forwardLayerNum - 1) {                          // GCN_FORWARD_NN
        ShareTensor h;
        sci::twoPartyGCNRelu(vertexDataVec, h, dstTid, party);
        dstVec.swap(h);
        return;
    }
    const size_t numLabels = vecSize ? vertexDataVec[0].size() : 0;               // GCN_FORWARD_PREDICTION
    ShareVecVec label;
    for (size_t i = 0; i < vecSize; ++i) label.push_back(isClient ? toShareVec((int)gs.labels[i], (int)numLabels) : ShareVec(numLabels, 0));
    ShareTensor p, p_minus_y;
    sci::twoPartyGCNForwardNNPredictionWithoutWeight(vertexDataVec, label, p, p_minus_y, dstTid, party);
    DoubleTensor plainP;
    sci::getPlainShareVecVec(p, plainP, dstTid, party);
    if (isClient) gs.plainP = plainP;
    for (size_t i = gs.trainSetSize; i < vecSize; ++i) p_minus_y[i] = ShareVec(numLabels, 0);   // gcn.h:639-641
    dstVec.swap(p_minus_y);
}

// ss_...h:704-902 for the single peer i of a two-party run (i is also the co-party)
void clientIteration(GraphSummary& gs, uint64_t iter) {
    const uint64_t tileIndex = gs.tileIndex, i = 1 - tileIndex;
    const uint32_t width = gs.localVertexSvv.empty() ? 0 : (uint32_t)gs.localVertexSvv[0].size();
    uint32_t preprocessId = 0;
    PreScatterComp(gs, gs.localVertexSvv, gs.localVertexInDeg, gs.localVertexSvv, iter, i, sci::ALICE);
    std::vector<ShareVecVec> updateSrcs(tileNum);
    client_oblivious_mapper_online(gs.localVertexPos, gs.updateSrcVertexPos[tileIndex], gs.localVertexSvv, updateSrcs[tileIndex], width, iter, preprocessId++, i);
    client_oblivious_mapper_online(gs.localVertexPos, gs.updateSrcVertexPos[i], gs.localVertexSvv, updateSrcs[i], width, iter, preprocessId++, i);
    auto clientComputeUpdate = [&](ShareVecVec& updateSrc, ShareVecVec& duplicatedUpdateSvv, uint64_t dstTid) {
        duplicatedUpdateSvv.clear();
        ScatterComp(updateSrc, duplicatedUpdateSvv);
        UpdatePreMergeComp(duplicatedUpdateSvv, gs.updateDstVertexPos[dstTid], i, sci::ALICE);
    };
    ShareVecVec duplicatedUpdateSvv;
    clientComputeUpdate(updateSrcs[tileIndex], duplicatedUpdateSvv, tileIndex);
    client_oblivious_mapper_online(gs.updateDstVertexPos[tileIndex], gs.localVertexPos, duplicatedUpdateSvv, gs.localUpdateSvvs[tileIndex], width, iter,
                                   preprocessId++, i);
    clientComputeUpdate(updateSrcs[i], duplicatedUpdateSvv, i);
    gs.remoteUpdateSvvs[i].swap(duplicatedUpdateSvv);
    gs.remoteUpdateReady.release();                                              // hand-off with this party's server thread, ss_...h:838-841
    gs.localUpdateReady.acquire();
    ShareVecVec tmpUpdateSvv;
    client_oblivious_mapper_online(gs.remoteMirrorVertexPos[i], gs.localVertexPos, gs.localUpdateSvvs[i], tmpUpdateSvv, width, iter, preprocessId++, i, true);
    gs.localUpdateSvvs[i].swap(tmpUpdateSvv);
    for (uint64_t j = 0; j < tileNum; ++j)
        GatherComp(gs.localVertexSvv, gs.localUpdateSvvs[j], gs.isGatherDstVertexDummy[j], gs.localVertexInDeg, iter, j, i, sci::ALICE);
    ShareVecVec curResult;
    ApplyComp(gs, iter, gs.localVertexSvv, curResult, i, true);
    gs.localVertexSvv.swap(curResult);
}

// ss_...h:936-1184 for the single peer i of a two-party run (this party is i's co-party)
void serverIteration(GraphSummary& gs, uint64_t iter) {
    const uint64_t tileIndex = gs.tileIndex, i = 1 - tileIndex;
    std::vector<uint64_t> zeroDeg(gs.remoteVertexSvvs[i].size(), 0);
    PreScatterComp(gs, gs.remoteVertexSvvs[i], zeroDeg, gs.remoteVertexSvvs[i], iter, i, sci::BOB);
    uint32_t preprocessId = 0;
    ShareVecVec coUpdateSrc, updateSrc;
    server_oblivious_mapper_online(gs.remoteVertexSvvs[i], coUpdateSrc, iter, preprocessId++, i);
    server_oblivious_mapper_online(gs.remoteVertexSvvs[i], updateSrc, iter, preprocessId++, i);
    auto serverComputeUpdate = [&](ShareVecVec& src, ShareVecVec& duplicatedUpdateSvv) {
        duplicatedUpdateSvv.clear();
        ScatterComp(src, duplicatedUpdateSvv);
        std::vector<uint64_t> zeroPosVec(src.size(), 0);
        UpdatePreMergeComp(duplicatedUpdateSvv, zeroPosVec, i, sci::BOB);
    };
    ShareVecVec duplicatedUpdateSvv;
    serverComputeUpdate(coUpdateSrc, duplicatedUpdateSvv);
    server_oblivious_mapper_online(duplicatedUpdateSvv, gs.remoteUpdateSvvs[tileIndex], iter, preprocessId++, i);
    duplicatedUpdateSvv.clear();
    serverComputeUpdate(updateSrc, duplicatedUpdateSvv);
    gs.localUpdateSvvs[i].swap(duplicatedUpdateSvv);
    gs.localUpdateReady.release();                                               // ss_...h:1069-1072
    gs.remoteUpdateReady.acquire();
    ShareVecVec tmpUpdateSvv;
    server_oblivious_mapper_online(gs.remoteUpdateSvvs[i], tmpUpdateSvv, iter, preprocessId++, i);
    gs.remoteUpdateSvvs[i].swap(tmpUpdateSvv);
    std::vector<ShareVecVec> remoteUpdateSvvs(tileNum);                           // ss_...h:1092-1094
    remoteUpdateSvvs[tileIndex].swap(gs.remoteUpdateSvvs[i]);
    remoteUpdateSvvs[i].swap(gs.remoteUpdateSvvs[tileIndex]);
    for (uint64_t j = 0; j < tileNum; ++j) {
        std::vector<bool> zeroIsDummy(gs.remoteVertexSvvs[i].size(), false);
        std::vector<uint64_t> zeros(gs.remoteVertexSvvs[i].size(), 0);
        GatherComp(gs.remoteVertexSvvs[i], remoteUpdateSvvs[j], zeroIsDummy, zeros, iter, j, i, sci::BOB);
    }
    ShareVecVec curResult;
    ApplyComp(gs, iter, gs.remoteVertexSvvs[i], curResult, i, false);
    gs.remoteVertexSvvs[i].swap(curResult);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <input> <output>\n", argv[0]); return 2; }
    try {
        Reader in{fopen(argv[1], "rb")};
        if (!in.f) throw std::runtime_error("cannot open input");
        const uint64_t seed = in.one(), iters = in.one();
        GraphSummary gs[2];
        for (uint64_t t = 0; t < 2; ++t) {
            GraphSummary& g = gs[t];
            g.tileIndex = t;
            g.localVertexPos = in.vec(); g.localVertexInDeg = in.vec(); g.labels = in.vec(); g.trainSetSize = in.one();
            g.updateSrcVertexPos.resize(2); g.updateDstVertexPos.resize(2); g.remoteMirrorVertexPos.resize(2); g.isGatherDstVertexDummy.resize(2);
            for (int j = 0; j < 2; ++j) {
                g.updateSrcVertexPos[j] = in.vec(); g.updateDstVertexPos[j] = in.vec(); g.remoteMirrorVertexPos[j] = in.vec();
                std::vector<uint64_t> d = in.vec();
                g.isGatherDstVertexDummy[j].assign(d.begin(), d.end());
            }
            g.localVertexSvv = in.mat();
            g.remoteVertexSvvs.resize(2); g.localUpdateSvvs.resize(2); g.remoteUpdateSvvs.resize(2);
            g.remoteVertexSvvs[1 - t] = in.mat();
            for (int l = 0; l < 2; ++l) g.localWeight.push_back(in.mat());
            for (int l = 0; l < 2; ++l) g.remoteWeight.push_back(in.mat());
        }
        fclose(in.f);
        // one channel per data owner: the owner's client thread (ALICE) <-> the co-party's server thread (BOB)
        cognn_shim::LocalPipe pipe[2];
        for (uint64_t t = 0; t < 2; ++t) {
            cognn_shim::open_session(t, 1 - t, sci::ALICE, seed, pipe[t].alice());
            cognn_shim::open_session(1 - t, t, sci::BOB, seed, pipe[t].bob());
        }
        FILE* out = fopen(argv[2], "wb");
        if (!out) throw std::runtime_error("cannot open output");
        std::string failure;
        std::mutex fm;
        for (uint64_t iter = 0; iter < iters; ++iter) {
            std::vector<std::thread> threads;
            for (uint64_t t = 0; t < 2; ++t) {
                auto guarded = [&, t](std::function<void()> body) {
                    cognn_shim::self_tid() = t;
                    try { body(); } catch (const std::exception& ex) { std::lock_guard<std::mutex> lk(fm); failure = ex.what(); fprintf(stderr, "%s\n", ex.what()); std::_Exit(1); }
                };
                threads.emplace_back(guarded, [&, t] { clientIteration(gs[t], iter); });
                threads.emplace_back(guarded, [&, t] { serverIteration(gs[t], iter); });
            }
            for (auto& th : threads) th.join();
            for (uint64_t t = 0; t < 2; ++t) { put(out, gs[t].localVertexSvv); put(out, gs[1 - t].remoteVertexSvvs[t]); }
        }
        // the probabilities revealed to each owner at the prediction layer
        for (uint64_t t = 0; t < 2; ++t) {
            ShareVecVec pm;
            for (auto& row : gs[t].plainP) { ShareVec r; for (double v : row) r.push_back((uint64_t)std::llround(v * 65536.0)); pm.push_back(r); }
            put(out, pm);
        }
        fclose(out);
        cognn_shim::close_sessions();
    } catch (const std::exception& ex) {
        fprintf(stderr, "shim_iteration: %s\n", ex.what());
        return 1;
    }
    return 0;
}
