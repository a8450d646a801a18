// cognn_gas_kernel.hpp — the upper seam of the drop-in boundary (seam 1): CoGNN's Scatter-Gather-Apply
// operator API - the pure virtuals of SSEdgeCentricAlgoKernel (include/ss_vertex_centric_algo_kernel.h:78-133) with their
// names, argument order and argument meaning - over tensors that STAY ON THE DEVICE.
//
//   cognn_gas::SSEdgeCentricAlgoKernel<Svv>   declares PreScatterComp / ScatterComp / UpdatePreMergeComp / GatherComp / ApplyComp /
//                                             onAlgoKernelStart and the dimension getters, and drives them: onIteration (the k-1
//                                             client threads of a party, ss_...h:680-910) and runAlgoKernelServer (its k-1
//                                             persistent server threads, ss_...h:912-1189), including the apply-only iterations,
//                                             the k >= 3 re-replication of the co-party's share (:997-1002, :1152-1158) and the
//                                             delegation of the other servers' update shares to the co-party (:1087-1100);
//   cognn_gas::GCNEdgeCentricAlgoKernel<Svv>  the CoGNN-Opt callbacks (algo_kernels/vertex_centric/optimize-gcn/gcn.h:198-811):
//                                             forward, prediction with the revealed-p metrics, backward (g = (p-y).W1^T, ReLU',
//                                             weight gradients, gradient scale / apply) and weight averaging (:747-802).
//
// Svv is the share-tensor container: cognn_shim::DevMat - a [rows x cols] uint64 tensor resident in HBM; every callback then
// runs HIP kernels of libcognn_hip.so on device pointers, the two roles of a protocol call hand their openings over device to
// device, and index structure is uploaded once (include/cognn_sci_shim.hpp) - or ShareVecVec, the reference's nested host
// vectors, through the same code (the reference's own data flow: one upload / download per call).  tests/gas_epochs.cpp runs
// whole training epochs of k = 2 and k = 3 parties through this header in both forms; tests/test_shim_gpu.py compares every
// share after every GAS iteration with the oracle, bit for bit.
//
// What is NOT here: the legacy task-queue virtuals (genScatterTask / genGatherTask / genApplyTaskVec / write*TaskResult,
// ss_...h:83,108-111,122-123): their payload types (include/task/task.h:287-1463) belong to the HE / SGX back ends, the SS path
// only constructs and drops them (ss_...h:1120-1122).  The reference's TaskComm / CommSync singletons are the Transport
// argument below.
#ifndef COGNN_GAS_KERNEL_HPP_
#define COGNN_GAS_KERNEL_HPP_
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <thread>

#include "cognn_sci_shim.hpp"

namespace cognn_gas {

// ---- threads: include/utils/threads.h:55-110 (barrier), the Semaphore of TaskUtil.h ---------------------------------------
class Semaphore {
public:
    void release() { std::lock_guard<std::mutex> lk(m_); ++n_; cv_.notify_one(); }
    void acquire() { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return n_ > 0; }); --n_; }
private:
    std::mutex m_; std::condition_variable cv_; int n_ = 0;
};
class bar_t {
public:
    explicit bar_t(size_t n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        const uint64_t gen = gen_;
        if (++arrived_ == n_) { arrived_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen_ != gen; });
    }
private:
    std::mutex m_; std::condition_variable cv_; size_t n_, arrived_ = 0; uint64_t gen_ = 0;
};

// ---- GNNParam (include/task/task.h:78-170) ----------------------------------------------------------------------------------
struct GNNParam {
    uint32_t num_layers = 2, num_labels = 0, input_dim = 0, hidden_dim = 0;
    double learning_rate = 0.1, train_ratio = 0.2, val_ratio = 0.2, test_ratio = 0.6;
};

// ---- what TaskComm::send/recvShareVecVec (weights, gcn.h:752-778) and CommSync::send/recvShareVecVec (co-share replicas and
//      delegated updates, ss_...h:970,1000,1090-1097,1156) carry between parties ----------------------------------------------
enum { MSG_REPLICA = 1, MSG_UPDATE = 2, MSG_WEIGHT_LOCAL = 3, MSG_WEIGHT_REMOTE = 4, MSG_WEIGHT_BACK_LOCAL = 5, MSG_WEIGHT_BACK_REMOTE = 6 };
template <class Svv>
struct Transport {
    virtual ~Transport() {}
    // `owner`: the party whose vertices (or, for weights, whose pair) the tensor belongs to - distinguishes the messages that
    // different threads of one party exchange with the same peer
    virtual void send(const Svv& v, uint64_t srcTid, uint64_t dstTid, int kind, uint64_t owner) = 0;
    virtual void recv(Svv& v, uint64_t srcTid, uint64_t dstTid, int kind, uint64_t owner) = 0;
};
// all parties in one process: a mailbox (device tensors are handed over as they are: nobody modifies a sent tensor)
template <class Svv>
class LocalMesh : public Transport<Svv> {
public:
    void send(const Svv& v, uint64_t s, uint64_t d, int kind, uint64_t owner) override {
        std::lock_guard<std::mutex> lk(m_);
        box_[std::make_tuple(s, d, kind, owner)].push_back(v);
        cv_.notify_all();
    }
    void recv(Svv& v, uint64_t s, uint64_t d, int kind, uint64_t owner) override {
        std::unique_lock<std::mutex> lk(m_);
        auto key = std::make_tuple(s, d, kind, owner);
        cv_.wait(lk, [&] { return !box_[key].empty(); });
        v = box_[key].front();
        box_[key].pop_front();
    }
private:
    std::mutex m_; std::condition_variable cv_;
    std::map<std::tuple<uint64_t, uint64_t, int, uint64_t>, std::deque<Svv>> box_;
};

// ---- GraphSummary (ss_...h:24-58): the per-party state of a run --------------------------------------------------------------
template <class Svv>
struct GraphSummary {
    typedef std::map<std::string, std::vector<Svv>> TensorVecMap;
    size_t tileNum = 0, tileIndex = 0;
    // onPreprocessClient's arrays (ss_...h:295-534; App. B of SURVEY.md)
    std::vector<uint64_t> localVertexPos, localVertexInDeg;
    std::vector<int> localVertexLabel;                       // localVertexVec[i]->data().label
    std::vector<bool> isLocalVertexBorder;
    std::vector<std::vector<uint64_t>> updateSrcVertexPos, updateDstVertexPos, remoteMirrorVertexPos;
    std::vector<std::vector<uint64_t>> updateSrcOutDeg, updateDstInDeg, remoteUpdateDstInDeg;   // read by original-gcn's ScatterComp only
    std::vector<std::vector<bool>> isGatherDstVertexDummy;
    // shares
    Svv localVertexSvv, localVertexSvvBackup;
    std::vector<Svv> remoteVertexSvvs, remoteVertexSvvsBackup, localUpdateSvvs, remoteUpdateSvvs;
    std::vector<Svv> localWeight, remoteWeight;              // per layer
    std::vector<TensorVecMap> localVertexInterDataTVs, remoteVertexInterDataTVs;   // per layer: "h_t", "z", "p", "g", "d"
    double learningRate = 0.1;
    // the semaphores TaskComm hands out (ss_...h:838-841,1069-1072; gcn.h:747-751,780,791-792)
    std::vector<std::unique_ptr<Semaphore>> localUpdateReadySmp, remoteUpdateReadySmp;
    Semaphore remoteWeightReadySmp, weightAvgFinishedSmp;
    // what the client printed at the last prediction layer (gcn.h:620-632)
    struct Metrics { double loss = 0, full = 0, train = 0, borderTrain = 0, test = 0, borderTest = 0; size_t vertices = 0, border = 0; bool valid = false; } metrics;
    DoubleTensor plainP;

    void init(size_t tiles, size_t index, uint32_t layers) {
        tileNum = tiles; tileIndex = index;
        updateSrcVertexPos.resize(tiles); updateDstVertexPos.resize(tiles); remoteMirrorVertexPos.resize(tiles);
        updateSrcOutDeg.resize(tiles); updateDstInDeg.resize(tiles); remoteUpdateDstInDeg.resize(tiles);
        isGatherDstVertexDummy.resize(tiles);
        remoteVertexSvvs.resize(tiles); remoteVertexSvvsBackup.resize(tiles); localUpdateSvvs.resize(tiles); remoteUpdateSvvs.resize(tiles);
        localWeight.resize(layers); remoteWeight.resize(layers);
        localVertexInterDataTVs.resize(layers); remoteVertexInterDataTVs.resize(layers);
        for (size_t i = 0; i < tiles; ++i) {
            localUpdateReadySmp.emplace_back(new Semaphore());
            remoteUpdateReadySmp.emplace_back(new Semaphore());
        }
    }
};

// ---- the GAS operator API and its driver (ss_...h) ---------------------------------------------------------------------------
template <class Svv>
class SSEdgeCentricAlgoKernel {
public:
    typedef GraphSummary<Svv> GS;
    virtual ~SSEdgeCentricAlgoKernel() {}

    // ss_...h:78-82
    virtual uint32_t getPlainNumPerOperand() const = 0;
    virtual uint32_t getPlainNumPerOperand(uint64_t layer) const = 0;
    virtual uint32_t getForwardLayerNum() const = 0;
    virtual uint32_t getBackwardLayerNum() const = 0;
    virtual std::vector<uint32_t> getDimensionVec() const = 0;
    // ss_...h:84-92
    virtual void PreScatterComp(GS& gs, const Svv& vertexSvv, std::vector<uint64_t>& vertexOutDeg, Svv& scaledVertexSvv, uint64_t iter, uint64_t coTid,
                                int party) const = 0;
    // ss_...h:93-100
    virtual void ScatterComp(Svv& updateSrcSvv, std::vector<uint64_t>& updateSrcOutDeg, std::vector<uint64_t>& updateDstInDeg, Svv& duplicatedUpdateSvv,
                             uint64_t coTid, int party) const = 0;
    // ss_...h:101-106
    virtual void UpdatePreMergeComp(Svv& duplicatedUpdateSvv, std::vector<uint64_t>& updateDstVertexPos, uint64_t coTid, int party) const = 0;
    // ss_...h:110-119
    virtual void GatherComp(Svv& vertexSvv, Svv& updateSvv, std::vector<bool>& isGatherDstVertexDummy, std::vector<uint64_t>& localVertexInDeg, uint64_t iter,
                            uint64_t updateSrcTid, uint64_t coTid, int party) const = 0;
    // ss_...h:122-131
    virtual void ApplyComp(GS& gs, uint64_t iter, const Svv& vertexDataVec, const std::vector<uint64_t>& localVertexInDeg, Svv& dstVec, uint64_t tileIndex,
                           uint64_t dstTid, bool isClient) const = 0;
    // ss_...h:133 (the graph tile's vertex data arrive as the plain feature rows of the local vertices, ascending vid, and their
    // in-degrees at load time; the shares of the co-party go out through `secondShare`)
    virtual void onAlgoKernelStart(GS& gs, const std::vector<std::vector<double>>& features, const std::vector<uint64_t>& inDegAtLoad,
                                   ShareVecVec& secondShare, std::vector<ShareTensor>& secondWeightShare) const = 0;

    Transport<Svv>* comm = nullptr;                          // CommSync / TaskComm stand-in, set before the run

    bool applyOnly(uint64_t iter) const {                    // ss_...h:709, 941
        const uint32_t f = getForwardLayerNum(), e = f + getBackwardLayerNum();
        return iter % e != 0 && (iter % e) % f == 0;
    }

    // One GAS iteration of party gs.tileIndex as data owner: ss_...h:680-910.
    bool onIteration(GS& gs, uint64_t iter) const {
        const size_t tileNum = gs.tileNum, tileIndex = gs.tileIndex;
        const uint32_t plainNumPerOperand = getPlainNumPerOperand(iter);
        const uint32_t epochLayerNum = getForwardLayerNum() + getBackwardLayerNum();
        if (iter % epochLayerNum == 0) gs.localVertexSvv = cognn_shim::svv_clone(gs.localVertexSvvBackup);   // back to the first layer
        std::vector<Svv> updateSrcs(tileNum);
        std::vector<std::thread> threads;
        bar_t barrier(tileNum - 1);
        // The reference's threads for the other peers read gs.localVertexSvv while the co-party's thread may still be inside
        // PreScatterComp (ss_...h:734-763: no ordering between them; in practice the peer's server only joins the mapper after it
        // has received the co-party's replica).  Here the order is explicit: they wait until PreScatterComp has replaced the tensor.
        Semaphore preScatterDone;
        for (size_t i = 0; i < tileNum; ++i) {
            if (i == tileIndex) continue;
            threads.emplace_back([&, i]() {
                cognn_shim::self_tid() = tileIndex;
                try {
                    const size_t co = (tileIndex + 1) % tileNum;
                    uint32_t preprocessId = 0;
                    if (applyOnly(iter)) {                   // first backward step of a layer: Apply only, ss_...h:709-732
                        if (i == co) {
                            Svv curResult;
                            ApplyComp(gs, iter, gs.localVertexSvv, gs.localVertexInDeg, curResult, tileIndex, i, true);
                            gs.localVertexSvv.swap(curResult);
                        }
                        return;
                    }
                    if (i == co) {
                        PreScatterComp(gs, gs.localVertexSvv, gs.localVertexInDeg, gs.localVertexSvv, iter, i, sci::ALICE);
                        for (size_t t = 2; t < tileNum; ++t) preScatterDone.release();
                    } else {
                        preScatterDone.acquire();
                    }
                    // rows of the sources of the local edges / of the edges towards party i (ss_...h:751-763)
                    if (i == co)
                        client_oblivious_mapper_online(gs.localVertexPos, gs.updateSrcVertexPos[tileIndex], gs.localVertexSvv, updateSrcs[tileIndex],
                                                       plainNumPerOperand, iter, preprocessId++, i);
                    client_oblivious_mapper_online(gs.localVertexPos, gs.updateSrcVertexPos[i], gs.localVertexSvv, updateSrcs[i], plainNumPerOperand, iter,
                                                   preprocessId++, i);
                    auto clientComputeUpdate = [&](Svv& updateSrc, Svv& duplicatedUpdateSvv, size_t dstTid) {   // ss_...h:788-811
                        duplicatedUpdateSvv.clear();
                        ScatterComp(updateSrc, gs.updateSrcOutDeg[dstTid], gs.updateDstInDeg[dstTid], duplicatedUpdateSvv, i, sci::ALICE);
                        UpdatePreMergeComp(duplicatedUpdateSvv, gs.updateDstVertexPos[dstTid], i, sci::ALICE);
                    };
                    Svv duplicatedUpdateSvv;
                    if (i == co) {                           // per-vertex sums over the local edges, ss_...h:815-821
                        clientComputeUpdate(updateSrcs[tileIndex], duplicatedUpdateSvv, tileIndex);
                        client_oblivious_mapper_online(gs.updateDstVertexPos[tileIndex], gs.localVertexPos, duplicatedUpdateSvv, gs.localUpdateSvvs[tileIndex],
                                                       plainNumPerOperand, iter, preprocessId++, i);
                    }
                    clientComputeUpdate(updateSrcs[i], duplicatedUpdateSvv, i);
                    gs.remoteUpdateSvvs[i].swap(duplicatedUpdateSvv);
                    gs.remoteUpdateReadySmp[i]->release();   // hand-off with this party's server thread for i, ss_...h:838-841
                    gs.localUpdateReadySmp[i]->acquire();
                    Svv tmpUpdateSvv;                        // per-vertex update from party i's edges, ss_...h:847-854
                    client_oblivious_mapper_online(gs.remoteMirrorVertexPos[i], gs.localVertexPos, gs.localUpdateSvvs[i], tmpUpdateSvv, plainNumPerOperand,
                                                   iter, preprocessId++, i, true);
                    gs.localUpdateSvvs[i].swap(tmpUpdateSvv);
                    barrier.wait();
                    if (i == co) {                           // Gather over every source party, then Apply: ss_...h:866-896
                        for (size_t j = 0; j < tileNum; ++j) {
                            if (gs.localVertexSvv.size() != gs.localUpdateSvvs[j].size()) throw cognn_shim::Error("client: unmatched update num and vertex num");
                            GatherComp(gs.localVertexSvv, gs.localUpdateSvvs[j], gs.isGatherDstVertexDummy[j], gs.localVertexInDeg, iter, j, i, sci::ALICE);
                        }
                        Svv curResult;
                        ApplyComp(gs, iter, gs.localVertexSvv, gs.localVertexInDeg, curResult, tileIndex, i, true);
                        gs.localVertexSvv.swap(curResult);
                    }
                } catch (const std::exception& ex) {
                    fprintf(stderr, "cognn_gas: party %zu client thread %zu: %s\n", tileIndex, i, ex.what());
                    std::_Exit(1);                           // a peer blocked in a rendezvous cannot be unwound (the reference: exit(-1))
                }
            });
        }
        for (auto& t : threads) t.join();
        return false;
    }

    // The k-1 persistent server threads of party gs.tileIndex (one per data owner i): ss_...h:912-1189.
    // onIterationDone(i, iter): called by the thread of owner i after each iteration (test hook; may be empty).
    void runAlgoKernelServer(std::vector<std::thread>& threads, GS& gs, uint64_t maxIters,
                             std::function<void(size_t, uint64_t)> onIterationDone = nullptr) const {
        const size_t tileNum = gs.tileNum, tileIndex = gs.tileIndex;
        const uint32_t epochLayerNum = getForwardLayerNum() + getBackwardLayerNum();
        std::shared_ptr<bar_t> barrier(new bar_t(tileNum - 1));
        for (size_t i = 0; i < tileNum; ++i) {
            if (i == tileIndex) continue;
            threads.emplace_back([this, &gs, i, tileNum, tileIndex, epochLayerNum, maxIters, barrier, onIterationDone]() {
                cognn_shim::self_tid() = tileIndex;
                try {
                    const size_t coOfI = (i + 1) % tileNum;  // the computing server of owner i
                    const bool iAmCo = tileIndex == coOfI;
                    auto replicate = [&]() {                 // the co-party's fresh share goes to every other server, ss_...h:997-1002
                        for (size_t j = 0; j < tileNum; ++j)
                            if (j != tileIndex && j != i) comm->send(gs.remoteVertexSvvs[i], tileIndex, j, MSG_REPLICA, i);
                    };
                    for (uint64_t iter = 0; iter < maxIters; ++iter) {
                        if (iter % epochLayerNum == 0) gs.remoteVertexSvvs[i] = cognn_shim::svv_clone(gs.remoteVertexSvvsBackup[i]);
                        std::vector<uint64_t> zeroDeg(gs.remoteVertexSvvs[i].size(), 0);
                        if (applyOnly(iter)) {               // ss_...h:941-970
                            if (!iAmCo) comm->recv(gs.remoteVertexSvvs[i], coOfI, tileIndex, MSG_REPLICA, i);
                            else {
                                Svv curResult;
                                ApplyComp(gs, iter, gs.remoteVertexSvvs[i], zeroDeg, curResult, tileIndex, i, false);
                                gs.remoteVertexSvvs[i].swap(curResult);
                                replicate();
                            }
                            if (onIterationDone) onIterationDone(i, iter);
                            continue;
                        }
                        if (!iAmCo) comm->recv(gs.remoteVertexSvvs[i], coOfI, tileIndex, MSG_REPLICA, i);
                        else {
                            PreScatterComp(gs, gs.remoteVertexSvvs[i], zeroDeg, gs.remoteVertexSvvs[i], iter, i, sci::BOB);
                            replicate();
                        }
                        uint32_t preprocessId = 0;
                        Svv coUpdateSrc, updateSrc;
                        if (iAmCo) server_oblivious_mapper_online(gs.remoteVertexSvvs[i], coUpdateSrc, iter, preprocessId++, i);
                        server_oblivious_mapper_online(gs.remoteVertexSvvs[i], updateSrc, iter, preprocessId++, i);
                        auto serverComputeUpdate = [&](Svv& src, Svv& duplicatedUpdateSvv, bool dstIsLocal) {     // ss_...h:1020-1050
                            std::vector<uint64_t> zeros(src.size(), 0);
                            duplicatedUpdateSvv.clear();
                            ScatterComp(src, zeros, dstIsLocal ? zeros : gs.remoteUpdateDstInDeg[i], duplicatedUpdateSvv, i, sci::BOB);
                            std::vector<uint64_t> zeroPosVec(src.size(), 0);
                            UpdatePreMergeComp(duplicatedUpdateSvv, zeroPosVec, i, sci::BOB);
                        };
                        Svv duplicatedUpdateSvv;
                        if (iAmCo) {
                            serverComputeUpdate(coUpdateSrc, duplicatedUpdateSvv, true);
                            server_oblivious_mapper_online(duplicatedUpdateSvv, gs.remoteUpdateSvvs[tileIndex], iter, preprocessId++, i);
                            duplicatedUpdateSvv.clear();
                        }
                        serverComputeUpdate(updateSrc, duplicatedUpdateSvv, false);
                        gs.localUpdateSvvs[i].swap(duplicatedUpdateSvv);
                        gs.localUpdateReadySmp[i]->release();   // ss_...h:1069-1072
                        gs.remoteUpdateReadySmp[i]->acquire();
                        Svv tmpUpdateSvv;
                        server_oblivious_mapper_online(gs.remoteUpdateSvvs[i], tmpUpdateSvv, iter, preprocessId++, i);
                        gs.remoteUpdateSvvs[i].swap(tmpUpdateSvv);
                        barrier->wait();
                        std::vector<Svv> remoteUpdateSvvs;
                        if (!iAmCo) comm->send(gs.remoteUpdateSvvs[i], tileIndex, coOfI, MSG_UPDATE, i);   // delegate to the co-party, ss_...h:1089-1090
                        else {
                            remoteUpdateSvvs.resize(tileNum);
                            remoteUpdateSvvs[tileIndex].swap(gs.remoteUpdateSvvs[i]);
                            remoteUpdateSvvs[i].swap(gs.remoteUpdateSvvs[tileIndex]);
                            for (size_t j = 0; j < tileNum; ++j)
                                if (j != tileIndex && j != i) comm->recv(remoteUpdateSvvs[j], j, tileIndex, MSG_UPDATE, i);
                        }
                        barrier->wait();
                        if (!iAmCo) comm->recv(gs.remoteVertexSvvs[i], coOfI, tileIndex, MSG_REPLICA, i);
                        else {
                            for (size_t j = 0; j < tileNum; ++j) {
                                if (remoteUpdateSvvs[j].size() != gs.remoteVertexSvvs[i].size()) throw cognn_shim::Error("server: unmatched update num and vertex num");
                                std::vector<bool> zeroIsDummy(remoteUpdateSvvs[j].size(), false);
                                GatherComp(gs.remoteVertexSvvs[i], remoteUpdateSvvs[j], zeroIsDummy, zeroDeg, iter, j, i, sci::BOB);
                            }
                            Svv curResult;
                            ApplyComp(gs, iter, gs.remoteVertexSvvs[i], zeroDeg, curResult, tileIndex, i, false);
                            gs.remoteVertexSvvs[i].swap(curResult);
                            replicate();                     // ss_...h:1152-1158
                        }
                        barrier->wait();
                        if (onIterationDone) onIterationDone(i, iter);
                    }
                } catch (const std::exception& ex) {
                    fprintf(stderr, "cognn_gas: party %zu server thread %zu: %s\n", tileIndex, i, ex.what());
                    std::_Exit(1);
                }
            });
        }
    }
};

// ---- CoGNN-Opt: algo_kernels/vertex_centric/optimize-gcn/gcn.h ---------------------------------------------------------------
template <class Svv>
class GCNEdgeCentricAlgoKernel : public SSEdgeCentricAlgoKernel<Svv> {
public:
    typedef GraphSummary<Svv> GS;
    typedef typename GS::TensorVecMap TensorVecMap;
    GNNParam gnnParam;
    uint64_t sharingSeed = 0;                                // keys CryptoUtil::intoShares in onAlgoKernelStart
    bool printMetrics = false;                               // the client's printf lines of gcn.h:620-632
    bool inferenceVariant = false;                           // optimize-gcn-inference/gcn.h (bin/gcn-inference-optimize): each party scales its updated
                                                             // weights by 1 / tileNum itself (:680-681, 732-733) and the averaging step does not (:763-764 absent)
    explicit GCNEdgeCentricAlgoKernel(const GNNParam& p) : gnnParam(p) {}

    // gcn.h:893-948
    uint32_t getForwardLayerNum() const override { return gnnParam.num_layers; }
    uint32_t getBackwardLayerNum() const override { return 2 * gnnParam.num_layers; }
    uint32_t getEpochLayerNum() const { return getForwardLayerNum() + getBackwardLayerNum(); }
    std::vector<uint32_t> getDimensionVec() const override {
        return {gnnParam.hidden_dim, gnnParam.num_labels, gnnParam.num_labels, gnnParam.num_labels, gnnParam.hidden_dim, gnnParam.hidden_dim};
    }
    uint32_t getPlainNumPerOperand() const override { return gnnParam.hidden_dim; }
    uint32_t getPlainNumPerOperand(uint64_t iter) const override { return getDimensionVec()[iter % getEpochLayerNum()]; }

    static std::vector<uint64_t> normalizerOf(const std::vector<uint64_t>& deg) {              // gcn.h:219-221, 471-474, 536-539
        std::vector<uint64_t> n(deg.size());
        for (size_t i = 0; i < deg.size(); ++i) n[i] = deg[i] == 0 ? 0 : CryptoUtil::encodeDoubleAsFixedPoint(std::pow((double)deg[i] + 1, -0.5));
        return n;
    }
    uint32_t coForwardLayerOf(uint64_t iter) const {          // gcn.h:211-214
        const uint32_t e = (uint32_t)(iter % getEpochLayerNum()), f = getForwardLayerNum();
        return e < f ? e : f - 1 - ((e - f) / 2);
    }

    // gcn.h:198-255.  vertexSvv and scaledVertexSvv are the same object at both call sites (ss_...h:736-741, 985-993): the product
    // replaces the tensor, the row scale then reads it.
    void PreScatterComp(GS& gs, const Svv& vertexSvv, std::vector<uint64_t>& vertexOutDeg, Svv& scaledVertexSvv, uint64_t iter, uint64_t coTid,
                        int party) const override {
        const uint32_t epochLayerNum = getEpochLayerNum(), coForwardLayer = coForwardLayerOf(iter);
        const bool isForward = (iter % epochLayerNum) < getForwardLayerNum(), isClient = party == sci::ALICE;
        std::vector<uint64_t> normalizer = normalizerOf(vertexOutDeg);
        if (isForward) {
            const Svv& weight = isClient ? gs.localWeight[coForwardLayer] : gs.remoteWeight[coForwardLayer];
            TensorVecMap& vertexInterData = isClient ? gs.localVertexInterDataTVs[coForwardLayer] : gs.remoteVertexInterDataTVs[coForwardLayer];
            if (isClient && coTid == (gs.tileIndex + 1) % gs.tileNum) vertexInterData["h_t"] = {transpose(vertexSvv)};
            if (!isClient && (coTid + 1) % gs.tileNum == gs.tileIndex) vertexInterData["h_t"] = {transpose(vertexSvv)};
            sci::twoPartyGCNMatMul(vertexSvv, weight, scaledVertexSvv, coTid, party);
        }
        if (iter % epochLayerNum == 0) return;                // the input features were scaled in the clear (onAlgoKernelStart)
        sci::twoPartyGCNVectorScale(vertexSvv, normalizer, scaledVertexSvv, true, coTid, party);
    }

    // gcn.h:257-307: CoGNN-Opt scatters the transformed row unchanged
    void ScatterComp(Svv& updateSrcSvv, std::vector<uint64_t>&, std::vector<uint64_t>&, Svv& duplicatedUpdateSvv, uint64_t, int) const override {
        duplicatedUpdateSvv = updateSrcSvv;
    }
    // gcn.h:309-342
    void UpdatePreMergeComp(Svv& duplicatedUpdateSvv, std::vector<uint64_t>& updateDstVertexPos, uint64_t coTid, int party) const override {
        duplicatedUpdateSvv = prefix_network_aggregate(updateDstVertexPos, duplicatedUpdateSvv, AggregationOp::ADD_AGG, coTid, party, true);
    }
    // gcn.h:375-494
    void GatherComp(Svv& vertexSvv, Svv& updateSvv, std::vector<bool>& isGatherDstVertexDummy, std::vector<uint64_t>& localVertexInDeg, uint64_t iter,
                    uint64_t updateSrcTid, uint64_t coTid, int party) const override {
        std::vector<bool> cond(isGatherDstVertexDummy.size());
        for (size_t i = 0; i < cond.size(); ++i) cond[i] = !isGatherDstVertexDummy[i];
        sci::twoPartyGCNCondVectorAddition(vertexSvv, updateSvv, cond, vertexSvv, coTid, party);
        if (updateSrcTid == tileNum - 1 && (iter + 1) % getEpochLayerNum() != 0) {
            std::vector<uint64_t> normalizer = normalizerOf(localVertexInDeg);
            sci::twoPartyGCNVectorScale(vertexSvv, normalizer, vertexSvv, true, coTid, party);
        }
    }

    // gcn.h:515-811
    void ApplyComp(GS& gs, uint64_t iter, const Svv& vertexDataVec, const std::vector<uint64_t>& /*localVertexInDeg*/, Svv& dstVec, uint64_t tileIndex,
                   uint64_t dstTid, bool isClient) const override {
        const uint32_t epochLayerNum = getEpochLayerNum(), forwardLayerNum = getForwardLayerNum(), coForwardLayer = coForwardLayerOf(iter);
        const bool isForward = (iter % epochLayerNum) < forwardLayerNum;
        const size_t vecSize = vertexDataVec.size();
        const int party = isClient ? sci::ALICE : sci::BOB;
        const uint64_t trainSetSize = (uint64_t)(vecSize * gnnParam.train_ratio), valSetSize = (uint64_t)(vecSize * gnnParam.val_ratio);
        TensorVecMap& vertexInterData = isClient ? gs.localVertexInterDataTVs[coForwardLayer] : gs.remoteVertexInterDataTVs[coForwardLayer];
        if (isForward) {
            if (iter % epochLayerNum != forwardLayerNum - 1) {           // GCN_FORWARD_NN, gcn.h:546-558
                vertexInterData["z"] = {vertexDataVec};
                Svv new_h;
                sci::twoPartyGCNRelu(vertexDataVec, new_h, dstTid, party);
                dstVec.swap(new_h);
                return;
            }
            vertexInterData["z"] = {vertexDataVec};                     // GCN_FORWARD_PREDICTION, gcn.h:559-643
            ShareVecVec label;                                           // the client's one-hot labels; the server's are zero rows
            for (size_t i = 0; i < vecSize; ++i)
                label.push_back(isClient ? toShareVec(gs.localVertexLabel[i], (int)gnnParam.num_labels) : ShareVec(gnnParam.num_labels, 0));
            Svv p, p_minus_y;
            sci::twoPartyGCNForwardNNPredictionWithoutWeight(vertexDataVec, label, p, p_minus_y, dstTid, party);
            DoubleTensor plainP;
            sci::getPlainShareVecVec(p, plainP, dstTid, party);
            if (isClient) reportMetrics(gs, plainP, trainSetSize, valSetSize);
            vertexInterData["p"] = {p};
            cognn_shim::svv_zero_rows_from(p_minus_y, trainSetSize);   // gradients of the training set only, gcn.h:639-641
            dstVec.swap(p_minus_y);
            return;
        }
        // BACKWARD, gcn.h:644-745
        Svv& weightRef = isClient ? gs.localWeight[coForwardLayer] : gs.remoteWeight[coForwardLayer];
        Svv& coWeightRef = isClient ? gs.remoteWeight[coForwardLayer] : gs.localWeight[coForwardLayer];
        Svv weightT = transpose(weightRef);
        const bool isFirstOfTwo = ((iter % epochLayerNum) - forwardLayerNum) % 2 == 0;
        if (isFirstOfTwo) {
            if (coForwardLayer == forwardLayerNum - 1) {                // g = (p - y) . W^T, out = in: gcn.h:664-669
                Svv g;
                sci::twoPartyGCNMatMul(vertexDataVec, weightT, g, dstTid, party);
                vertexInterData["g"] = {g};
                dstVec = vertexDataVec;
            } else {                                                    // out = in (.) 1[z > 0]: gcn.h:702-708
                Svv g;
                sci::twoPartyGCNBackwardNNWithoutAH(vertexDataVec, vertexInterData["z"][0], weightT, dstVec, g, coForwardLayer == 0, dstTid, party);
                vertexInterData["g"] = {g};
            }
            return;
        }
        Svv d;                                                          // d = h_t . in ; scale ; W -= lr d ; out = g: gcn.h:671-684, 710-736
        sci::twoPartyGCNMatMul(vertexInterData["h_t"][0], vertexDataVec, d, dstTid, party);
        const double gradientScaler = (double)1 / trainSetSize;
        sci::twoPartyGCNMatrixScale(d, static_cast<uint64_t>(gradientScaler * (1 << SCALER_BIT_LENGTH)), d, dstTid, party);
        sci::twoPartyGCNApplyGradient(weightRef, d, static_cast<uint64_t>(gs.learningRate * (1 << SCALER_BIT_LENGTH)), weightRef, dstTid, party);
        if (inferenceVariant) {
            const double weightScaler = (double)1 / gs.tileNum;
            sci::twoPartyGCNMatrixScale(weightRef, static_cast<uint64_t>(weightScaler * (1 << SCALER_BIT_LENGTH)), weightRef, dstTid, party);
        }
        vertexInterData["d"] = {d};
        dstVec.swap(vertexInterData["g"][0]);
        averageWeights(gs, tileIndex, coForwardLayer, isClient, weightRef, coWeightRef);
    }

    // weight averaging, gcn.h:747-802 (original-gcn/gcn.h:659-711): parties >= 2 ship both their weight shares to parties 1 (local
    // share) and 0 (the share they hold as a co-party); 0 and 1 sum, scale by 1 / tileNum between themselves and hand the average back
    void averageWeights(GS& gs, uint64_t tileIndex, uint32_t coForwardLayer, bool isClient, Svv& weightRef, Svv& coWeightRef) const {
        const size_t tileNum = gs.tileNum;
        Transport<Svv>* comm = this->comm;
        if (isClient) {
            gs.remoteWeightReadySmp.acquire();                          // this party's server thread updated its share of the layer
            if (tileIndex == 0 || tileIndex == 1) {
                for (size_t i = 0; i < tileNum; ++i) {
                    if (i == tileIndex || i == 1 - tileIndex) continue;
                    Svv fromOther;
                    comm->recv(fromOther, i, tileIndex, tileIndex == 0 ? MSG_WEIGHT_REMOTE : MSG_WEIGHT_LOCAL, coForwardLayer);
                    sci::plaintext_add_matrix_in_place(weightRef, fromOther);
                }
                sci::plaintext_add_matrix_in_place(weightRef, coWeightRef);
                if (!inferenceVariant) {
                    const double weightScaler = (double)1 / tileNum;
                    sci::twoPartyGCNMatrixScale(weightRef, static_cast<uint64_t>(weightScaler * (1 << SCALER_BIT_LENGTH)), weightRef, 1 - tileIndex, (int)tileIndex + 1);
                }
                coWeightRef = weightRef;
                for (size_t i = 0; i < tileNum; ++i)
                    if (i != tileIndex && i != 1 - tileIndex)
                        comm->send(weightRef, tileIndex, i, tileIndex == 0 ? MSG_WEIGHT_BACK_REMOTE : MSG_WEIGHT_BACK_LOCAL, coForwardLayer);
            } else {
                comm->send(weightRef, tileIndex, 1, MSG_WEIGHT_LOCAL, coForwardLayer);
                comm->send(coWeightRef, tileIndex, 0, MSG_WEIGHT_REMOTE, coForwardLayer);
                comm->recv(weightRef, 1, tileIndex, MSG_WEIGHT_BACK_LOCAL, coForwardLayer);
                comm->recv(coWeightRef, 0, tileIndex, MSG_WEIGHT_BACK_REMOTE, coForwardLayer);
            }
            gs.weightAvgFinishedSmp.release();
        } else {
            gs.remoteWeightReadySmp.release();
            gs.weightAvgFinishedSmp.acquire();
        }
    }

    // gcn.h:819-887: feature rows * (inDeg + 1)^-1/2 in double, Glorot-uniform weights from srand(42) / rand() (the same on every
    // party), both split into two additive shares (CryptoUtil::intoShares).  The local shares go into gs, the second shares are
    // returned for distribution (ss_...h:205-232: features to every other party, weights to the next party of the ring).
    void onAlgoKernelStart(GS& gs, const std::vector<std::vector<double>>& features, const std::vector<uint64_t>& inDegAtLoad, ShareVecVec& secondShare,
                           std::vector<ShareTensor>& secondWeightShare) const override {
        const size_t n = features.size();
        ShareVecVec first(n);
        secondShare.assign(n, ShareVec());
        CryptoUtil::sharingSeedIs(sharingSeed, gs.tileIndex);
        for (size_t r = 0; r < n; ++r) {
            const double nm = std::pow((double)inDegAtLoad[r] + 1, -0.5);
            first[r].resize(features[r].size()); secondShare[r].resize(features[r].size());
            for (size_t j = 0; j < features[r].size(); ++j) CryptoUtil::intoShares(features[r][j] * nm, first[r][j], secondShare[r][j]);
        }
        cognn_shim::Session& s = cognn_shim::session((gs.tileIndex + 1) % gs.tileNum, sci::ALICE);
        cognn_shim::svv_from_host(s, first, gs.localVertexSvv);
        gs.localVertexSvvBackup = cognn_shim::svv_clone(gs.localVertexSvv);
        const uint32_t dims[3] = {gnnParam.input_dim, gnnParam.hidden_dim, gnnParam.num_labels};
        secondWeightShare.clear();
        for (uint32_t l = 0; l < gnnParam.num_layers && l < 2; ++l) {
            std::srand(42);                                             // re-seeded per matrix, gcn.h:841
            const double limit = std::sqrt(6.0 / (dims[l] + dims[l + 1]));
            ShareTensor w0(dims[l], ShareVec(dims[l + 1])), w1(dims[l], ShareVec(dims[l + 1]));
            for (uint32_t i = 0; i < dims[l]; ++i)
                for (uint32_t j = 0; j < dims[l + 1]; ++j) CryptoUtil::intoShares((double)std::rand() / RAND_MAX * 2 * limit - limit, w0[i][j], w1[i][j]);
            cognn_shim::svv_from_host(s, w0, gs.localWeight[l]);
            secondWeightShare.push_back(w1);
        }
        gs.learningRate = gnnParam.learning_rate;
    }

    // GatherComp has no GraphSummary argument: the number of parties of the run (TaskComm::getTileNum() in the reference)
    size_t tileNum = 0;
    void tileNumIs(size_t n) { tileNum = n; }

protected:
    void reportMetrics(GS& gs, DoubleTensor plainP, uint64_t trainSetSize, uint64_t valSetSize) const {   // gcn.h:606-632
        const size_t vecSize = plainP.size(), L = gnnParam.num_labels;
        DoubleTensor y(vecSize, std::vector<double>(L, 0.0));
        for (size_t i = 0; i < vecSize; ++i) {
            y[i][(size_t)gs.localVertexLabel[i]] = 1.0;
            for (size_t j = 0; j < L; ++j) if (plainP[i][j] == 0) plainP[i][j] = 0.001;
        }
        typename GS::Metrics m;
        m.loss = sci::cross_entropy_loss(y, plainP);
        DoubleTensor trainingY(y.begin(), y.begin() + trainSetSize), testY(y.begin() + trainSetSize + valSetSize, y.end());
        DoubleTensor trainingP(plainP.begin(), plainP.begin() + trainSetSize), testP(plainP.begin() + trainSetSize + valSetSize, plainP.end());
        std::vector<bool> border = gs.isLocalVertexBorder;
        border.resize(vecSize, false);
        std::vector<bool> trainingIsBorder(border.begin(), border.begin() + trainSetSize), testIsBorder(border.begin() + trainSetSize + valSetSize, border.end());
        m.full = sci::accuracy(y, plainP); m.train = sci::accuracy(trainingY, trainingP); m.borderTrain = sci::accuracy(trainingY, trainingP, trainingIsBorder);
        m.test = sci::accuracy(testY, testP); m.borderTest = sci::accuracy(testY, testP, testIsBorder);
        m.vertices = vecSize; m.border = sci::count_true(border); m.valid = true;
        gs.metrics = m;
        gs.plainP = plainP;
        if (printMetrics) {
            printf("cross-entropy-loss = %lf\n", m.loss);
            printf("full set accuracy = %lf\n", m.full);
            printf("training set accuracy = %lf\n", m.train);
            printf("border training set accuracy = %lf\n", m.borderTrain);
            printf("test set accuracy = %lf\n", m.test);
            printf("border test set accuracy = %lf\n", m.borderTest);
            printf("the number of vertices is %lu, the number of border vertices is %lu\n", (unsigned long)m.vertices, (unsigned long)m.border);
        }
    }
};

// ---- CoGNN (unoptimised): algo_kernels/vertex_centric/original-gcn/gcn.h -----------------------------------------------------
// Aggregate-then-transform: PreScatterComp copies, ScatterComp scales every message on its edge by the two degree normalisers,
// GatherComp scales the vertex's own row once in forward iterations, ApplyComp runs the fused ForwardNN / Prediction / BackwardNNInit /
// BackwardNN ops (cognn_sci_shim.hpp) and averages the weights after both backward iterations; 4 GAS iterations per epoch.
template <class Svv>
class GCNOriginalEdgeCentricAlgoKernel : public GCNEdgeCentricAlgoKernel<Svv> {
public:
    typedef GCNEdgeCentricAlgoKernel<Svv> Base;
    typedef GraphSummary<Svv> GS;
    typedef typename GS::TensorVecMap TensorVecMap;
    using Base::gnnParam;
    explicit GCNOriginalEdgeCentricAlgoKernel(const GNNParam& p) : Base(p) {}

    // original-gcn/gcn.h:802-851
    uint32_t getBackwardLayerNum() const override { return gnnParam.num_layers; }
    std::vector<uint32_t> getDimensionVec() const override { return {gnnParam.input_dim, gnnParam.hidden_dim, gnnParam.num_labels, gnnParam.hidden_dim}; }
    uint32_t getPlainNumPerOperand() const override { return gnnParam.input_dim; }
    uint32_t getPlainNumPerOperand(uint64_t iter) const override { return getDimensionVec()[iter % this->getEpochLayerNum()]; }
    uint32_t coForwardLayerOfOriginal(uint64_t iter) const {  // :337-340, 431-434
        const uint32_t e = (uint32_t)(iter % this->getEpochLayerNum()), f = this->getForwardLayerNum();
        return e < f ? e : f - 1 - (e - f);
    }

    // :198-209 - both arguments are the same object at the call sites: nothing to do
    void PreScatterComp(GS&, const Svv& vertexSvv, std::vector<uint64_t>&, Svv& scaledVertexSvv, uint64_t, uint64_t, int) const override {
        if (&vertexSvv != &scaledVertexSvv) scaledVertexSvv = cognn_shim::svv_clone(vertexSvv);
    }
    // :211-251 - the client passes the real degrees of its edges, the server zeros / the destination's in-degrees (ss_...h:800, 1041-1043)
    void ScatterComp(Svv& updateSrcSvv, std::vector<uint64_t>& updateSrcOutDeg, std::vector<uint64_t>& updateDstInDeg, Svv& duplicatedUpdateSvv, uint64_t coTid,
                     int party) const override {
        std::vector<uint64_t> normalizer0 = Base::normalizerOf(updateSrcOutDeg), normalizer1 = Base::normalizerOf(updateDstInDeg);
        sci::twoPartyGCNVectorScale(updateSrcSvv, normalizer0, normalizer1, duplicatedUpdateSvv, coTid, party);
    }
    // :323-405 - forward: the vertex's own row is scaled once, before the first addition
    void GatherComp(Svv& vertexSvv, Svv& updateSvv, std::vector<bool>& isGatherDstVertexDummy, std::vector<uint64_t>& localVertexInDeg, uint64_t iter,
                    uint64_t updateSrcTid, uint64_t coTid, int party) const override {
        const bool isForward = (iter % this->getEpochLayerNum()) < this->getForwardLayerNum();
        if (isForward && updateSrcTid == 0) {
            std::vector<uint64_t> normalizer = Base::normalizerOf(localVertexInDeg);
            sci::twoPartyGCNVectorScale(vertexSvv, normalizer, vertexSvv, true, coTid, party);
        }
        std::vector<bool> cond(isGatherDstVertexDummy.size());
        for (size_t i = 0; i < cond.size(); ++i) cond[i] = !isGatherDstVertexDummy[i];
        sci::twoPartyGCNCondVectorAddition(vertexSvv, updateSvv, cond, vertexSvv, coTid, party);
    }
    // :426-713
    void ApplyComp(GS& gs, uint64_t iter, const Svv& vertexDataVec, const std::vector<uint64_t>& /*localVertexInDeg*/, Svv& dstVec, uint64_t tileIndex,
                   uint64_t dstTid, bool isClient) const override {
        const uint32_t epochLayerNum = this->getEpochLayerNum(), forwardLayerNum = this->getForwardLayerNum(), coForwardLayer = coForwardLayerOfOriginal(iter);
        const bool isForward = (iter % epochLayerNum) < forwardLayerNum;
        const size_t vecSize = vertexDataVec.size();
        const int party = isClient ? sci::ALICE : sci::BOB;
        const uint64_t trainSetSize = (uint64_t)(vecSize * gnnParam.train_ratio), valSetSize = (uint64_t)(vecSize * gnnParam.val_ratio);
        const std::vector<uint64_t> normalizer;               // (declared and left empty at :446)
        TensorVecMap& vertexInterData = isClient ? gs.localVertexInterDataTVs[coForwardLayer] : gs.remoteVertexInterDataTVs[coForwardLayer];
        if (isForward) {
            const Svv& weight = isClient ? gs.localWeight[coForwardLayer] : gs.remoteWeight[coForwardLayer];
            vertexInterData["ah_t"] = {transpose(vertexDataVec)};       // :452
            if (iter % epochLayerNum != forwardLayerNum - 1) {         // GCN_FORWARD_NN :455-485
                Svv z, new_h;
                sci::twoPartyGCNForwardNN(vertexDataVec, weight, normalizer, z, new_h, dstTid, party);
                vertexInterData["z"] = {z};
                dstVec.swap(new_h);
                return;
            }
            ShareVecVec label;                                          // GCN_FORWARD_PREDICTION :486-560
            for (size_t i = 0; i < vecSize; ++i)
                label.push_back(isClient ? toShareVec(gs.localVertexLabel[i], (int)gnnParam.num_labels) : ShareVec(gnnParam.num_labels, 0));
            Svv z, p, p_minus_y;
            sci::twoPartyGCNForwardNNPrediction(vertexDataVec, weight, label, normalizer, z, p, p_minus_y, dstTid, party);
            DoubleTensor plainP;
            sci::getPlainShareVecVec(p, plainP, dstTid, party);
            if (isClient) this->reportMetrics(gs, plainP, trainSetSize, valSetSize);
            vertexInterData["z"] = {z};
            vertexInterData["p"] = {p};
            cognn_shim::svv_zero_rows_from(p_minus_y, trainSetSize);
            dstVec.swap(p_minus_y);
            return;
        }
        Svv& weightRef = isClient ? gs.localWeight[coForwardLayer] : gs.remoteWeight[coForwardLayer];
        Svv& coWeightRef = isClient ? gs.remoteWeight[coForwardLayer] : gs.localWeight[coForwardLayer];
        Svv weightT = transpose(weightRef);                             // :565-568, taken before the update
        Svv d, g;
        if (coForwardLayer == forwardLayerNum - 1)                      // GCN_BACKWARD_NN_INIT :573-612
            sci::twoPartyGCNBackwardNNInit(vertexDataVec, vertexInterData["ah_t"][0], weightT, normalizer, d, g, dstTid, party);
        else                                                            // GCN_BACKWARD_NN :613-655
            sci::twoPartyGCNBackwardNN(vertexDataVec, vertexInterData["ah_t"][0], vertexInterData["z"][0], weightT, normalizer, d, g, coForwardLayer == 0, dstTid, party);
        const double gradientScaler = (double)1 / trainSetSize;
        sci::twoPartyGCNMatrixScale(d, static_cast<uint64_t>(gradientScaler * (1 << SCALER_BIT_LENGTH)), d, dstTid, party);
        sci::twoPartyGCNApplyGradient(weightRef, d, static_cast<uint64_t>(gs.learningRate * (1 << SCALER_BIT_LENGTH)), weightRef, dstTid, party);
        vertexInterData["d"] = {d};
        dstVec.swap(g);
        this->averageWeights(gs, tileIndex, coForwardLayer, isClient, weightRef, coWeightRef);   // :659-711, after BOTH backward iterations
    }
};

}  // namespace cognn_gas
#endif  // COGNN_GAS_KERNEL_HPP_
