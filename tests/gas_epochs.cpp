// Test program of the drop-in boundary: k parties in one process run whole gcn-optimize training epochs through
// include/cognn_gas_kernel.hpp - the GAS operator API of include/ss_vertex_centric_algo_kernel.h:78-133 and the client / server
// thread structure that drives it (:680-910, :912-1189), with the CoGNN-Opt callbacks of
// algo_kernels/vertex_centric/optimize-gcn/gcn.h:198-811 (forward, prediction, backward, weight averaging) - on top of
// include/cognn_sci_shim.hpp -> C ABI -> HIP.   Two container modes:
//   device   cognn_shim::DevMat: the share tensors live in HBM for the whole run (the upper seam);
//   host     ShareVecVec: the reference's nested vectors, one upload / download per sci:: call (the lower seam's wrappers).
// tests/test_shim_gpu.py writes the inputs (the oracle's preprocessing arrays and initial shares) and compares every share
// after every GAS iteration, the weights and the revealed probabilities with the oracle, bit for bit.
//   usage: gas_epochs <device|host> <input file> <output file>
#include <cstdio>

#include "../include/cognn_gas_kernel.hpp"

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
    double real() { const uint64_t u = one(); double d; memcpy(&d, &u, 8); return d; }
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

// what the input file holds for one party (host side)
struct PartyInput {
    std::vector<uint64_t> localVertexPos, localVertexInDeg, labels, border;
    std::vector<std::vector<uint64_t>> updateSrcVertexPos, updateDstVertexPos, remoteMirrorVertexPos, dummy;
    std::vector<std::vector<uint64_t>> updateSrcOutDeg, updateDstInDeg, remoteUpdateDstInDeg;   // original-gcn only (its ScatterComp reads them)
    ShareVecVec localVertexSvv;
    std::vector<ShareVecVec> remoteVertexSvvs;
    std::vector<ShareVecVec> localWeight, remoteWeight;
};

template <class Svv>
int run(uint64_t k, uint64_t seed, uint64_t iters, const cognn_gas::GNNParam& param, const std::vector<PartyInput>& in, const char* out_path, bool original,
        bool inference = false) {
    using namespace cognn_gas;
    // one channel per ordered pair (owner t, server j): t's client thread for j (ALICE) <-> j's server thread for t (BOB)
    std::vector<std::unique_ptr<cognn_shim::LocalPipe>> pipes(k * k);
    for (uint64_t t = 0; t < k; ++t)
        for (uint64_t j = 0; j < k; ++j) {
            if (j == t) continue;
            pipes[t * k + j].reset(new cognn_shim::LocalPipe());
            cognn_shim::open_session(t, j, sci::ALICE, seed, pipes[t * k + j]->alice());
            cognn_shim::open_session(j, t, sci::BOB, seed, pipes[t * k + j]->bob());
        }
    LocalMesh<Svv> mesh;
    // the callbacks of optimize-gcn/gcn.h, or of original-gcn/gcn.h (the unoptimised kernel: 4 GAS iterations per epoch)
    std::unique_ptr<GCNEdgeCentricAlgoKernel<Svv>> kernel_ptr(original ? new GCNOriginalEdgeCentricAlgoKernel<Svv>(param) : new GCNEdgeCentricAlgoKernel<Svv>(param));
    GCNEdgeCentricAlgoKernel<Svv>& kernel = *kernel_ptr;
    kernel.comm = &mesh;
    kernel.inferenceVariant = inference;                      // optimize-gcn-inference/gcn.h
    kernel.tileNumIs(k);
    std::vector<std::unique_ptr<GraphSummary<Svv>>> gs(k);
    for (uint64_t t = 0; t < k; ++t) {
        gs[t].reset(new GraphSummary<Svv>());
        GraphSummary<Svv>& g = *gs[t];
        const PartyInput& pi = in[t];
        g.init(k, t, param.num_layers);
        g.localVertexPos = pi.localVertexPos; g.localVertexInDeg = pi.localVertexInDeg;
        g.localVertexLabel.assign(pi.labels.begin(), pi.labels.end());
        g.isLocalVertexBorder.assign(pi.border.begin(), pi.border.end());
        g.learningRate = param.learning_rate;
        cognn_shim::self_tid() = t;
        cognn_shim::Session& mine = cognn_shim::session((t + 1) % k, sci::ALICE);
        for (uint64_t j = 0; j < k; ++j) {
            g.updateSrcVertexPos[j] = pi.updateSrcVertexPos[j]; g.updateDstVertexPos[j] = pi.updateDstVertexPos[j];
            g.remoteMirrorVertexPos[j] = pi.remoteMirrorVertexPos[j];
            g.isGatherDstVertexDummy[j].assign(pi.dummy[j].begin(), pi.dummy[j].end());
            if (original) { g.updateSrcOutDeg[j] = pi.updateSrcOutDeg[j]; g.updateDstInDeg[j] = pi.updateDstInDeg[j]; g.remoteUpdateDstInDeg[j] = pi.remoteUpdateDstInDeg[j]; }
            if (j != t) {
                cognn_shim::svv_from_host(mine, pi.remoteVertexSvvs[j], g.remoteVertexSvvs[j]);
                g.remoteVertexSvvsBackup[j] = cognn_shim::svv_clone(g.remoteVertexSvvs[j]);     // ss_...h:226-227
            }
        }
        cognn_shim::svv_from_host(mine, pi.localVertexSvv, g.localVertexSvv);
        g.localVertexSvvBackup = cognn_shim::svv_clone(g.localVertexSvv);
        for (uint32_t l = 0; l < param.num_layers; ++l) {
            cognn_shim::svv_from_host(mine, pi.localWeight[l], g.localWeight[l]);
            cognn_shim::svv_from_host(mine, pi.remoteWeight[l], g.remoteWeight[l]);
        }
    }
    // snapshots[iter][owner] = (owner share, co-party share), taken by the threads that own them right after the iteration
    std::vector<std::vector<std::pair<ShareVecVec, ShareVecVec>>> snap(iters, std::vector<std::pair<ShareVecVec, ShareVecVec>>(k));
    std::vector<std::thread> servers, clients;
    for (uint64_t t = 0; t < k; ++t)
        kernel.runAlgoKernelServer(servers, *gs[t], iters, [&, t](size_t i, uint64_t iter) {
            if (t == (i + 1) % k) cognn_shim::svv_to_host(gs[t]->remoteVertexSvvs[i], snap[iter][i].second);
        });
    for (uint64_t t = 0; t < k; ++t)
        clients.emplace_back([&, t]() {
            for (uint64_t iter = 0; iter < iters; ++iter) {  // while (iter < maxIters) onIteration(...), ss_...h:239-248
                kernel.onIteration(*gs[t], iter);
                cognn_shim::svv_to_host(gs[t]->localVertexSvv, snap[iter][t].first);
            }
        });
    for (auto& th : clients) th.join();
    for (auto& th : servers) th.join();
    FILE* out = fopen(out_path, "wb");
    if (!out) throw std::runtime_error("cannot open output");
    for (uint64_t iter = 0; iter < iters; ++iter)
        for (uint64_t t = 0; t < k; ++t) { put(out, snap[iter][t].first); put(out, snap[iter][t].second); }
    for (uint64_t t = 0; t < k; ++t)
        for (uint32_t l = 0; l < param.num_layers; ++l) {
            ShareVecVec w;
            cognn_shim::svv_to_host(gs[t]->localWeight[l], w); put(out, w);
            cognn_shim::svv_to_host(gs[t]->remoteWeight[l], w); put(out, w);
        }
    for (uint64_t t = 0; t < k; ++t) {                       // the probabilities revealed to each owner at the last prediction layer (Q16)
        ShareVecVec pm;
        for (auto& row : gs[t]->plainP) { ShareVec r; for (double v : row) r.push_back((uint64_t)std::llround(v * 65536.0)); pm.push_back(r); }
        put(out, pm);
        ShareVecVec mm(1);                                    // metrics as the client printed them, in 1e-6 units
        const auto& m = gs[t]->metrics;
        for (double v : {m.loss, m.full, m.train, m.borderTrain, m.test, m.borderTest}) mm[0].push_back((uint64_t)std::llround(v * 1e6));
        put(out, mm);
    }
    fclose(out);
    gs.clear();
    cognn_shim::close_sessions();
    return 0;
}

// onAlgoKernelStart alone (gcn.h:819-887): plain feature rows and load-time in-degrees in, the shares it deals out
int run_start(Reader& in, const char* out_path) {
    using namespace cognn_gas;
    const uint64_t k = in.one(), seed = in.one();
    GNNParam param;
    param.input_dim = (uint32_t)in.one(); param.hidden_dim = (uint32_t)in.one(); param.num_labels = (uint32_t)in.one();
    FILE* out = fopen(out_path, "wb");
    if (!out) throw std::runtime_error("cannot open output");
    cognn_shim::LocalPipe pipe;
    GCNEdgeCentricAlgoKernel<cognn_shim::DevMat> kernel(param);
    kernel.sharingSeed = seed;
    for (uint64_t t = 0; t < k; ++t) {
        const uint64_t n = in.one();
        std::vector<std::vector<double>> feat(n, std::vector<double>(param.input_dim));
        for (auto& row : feat) for (auto& v : row) v = in.real();
        std::vector<uint64_t> deg = in.vec();
        cognn_shim::self_tid() = t;
        cognn_shim::open_session(t, (t + 1) % k, sci::ALICE, seed, pipe.alice());
        GraphSummary<cognn_shim::DevMat> gs;
        gs.init(k, t, param.num_layers);
        ShareVecVec second;
        std::vector<ShareTensor> secondW;
        kernel.onAlgoKernelStart(gs, feat, deg, second, secondW);
        ShareVecVec h;
        gs.localVertexSvv.to_host(h); put(out, h);
        put(out, second);
        for (uint32_t l = 0; l < 2; ++l) { gs.localWeight[l].to_host(h); put(out, h); put(out, secondW[l]); }
    }
    fclose(out);
    cognn_shim::close_sessions();
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <device|host|idevice|odevice|ohost|start> <input> <output>\n", argv[0]); return 2; }
    try {
        Reader in{fopen(argv[2], "rb")};
        if (!in.f) throw std::runtime_error("cannot open input");
        if (std::string(argv[1]) == "start") return run_start(in, argv[3]);
        const std::string mode = argv[1];
        const bool original = mode == "odevice" || mode == "ohost";   // the input then carries the per-edge degree vectors too
        const uint64_t k = in.one(), seed = in.one(), iters = in.one();
        cognn_gas::GNNParam param;
        param.input_dim = (uint32_t)in.one(); param.hidden_dim = (uint32_t)in.one(); param.num_labels = (uint32_t)in.one();
        param.learning_rate = in.real(); param.train_ratio = in.real(); param.val_ratio = in.real(); param.test_ratio = in.real();
        std::vector<PartyInput> parties(k);
        for (uint64_t t = 0; t < k; ++t) {
            PartyInput& p = parties[t];
            p.localVertexPos = in.vec(); p.localVertexInDeg = in.vec(); p.labels = in.vec(); p.border = in.vec();
            p.updateSrcVertexPos.resize(k); p.updateDstVertexPos.resize(k); p.remoteMirrorVertexPos.resize(k); p.dummy.resize(k);
            p.updateSrcOutDeg.resize(k); p.updateDstInDeg.resize(k); p.remoteUpdateDstInDeg.resize(k);
            for (uint64_t j = 0; j < k; ++j) {
                p.updateSrcVertexPos[j] = in.vec(); p.updateDstVertexPos[j] = in.vec(); p.remoteMirrorVertexPos[j] = in.vec(); p.dummy[j] = in.vec();
                if (original) { p.updateSrcOutDeg[j] = in.vec(); p.updateDstInDeg[j] = in.vec(); p.remoteUpdateDstInDeg[j] = in.vec(); }
            }
            p.localVertexSvv = in.mat();
            p.remoteVertexSvvs.resize(k);
            for (uint64_t j = 0; j < k; ++j) if (j != t) p.remoteVertexSvvs[j] = in.mat();
            for (int l = 0; l < 2; ++l) p.localWeight.push_back(in.mat());
            for (int l = 0; l < 2; ++l) p.remoteWeight.push_back(in.mat());
        }
        fclose(in.f);
        if (mode == "idevice") return run<cognn_shim::DevMat>(k, seed, iters, param, parties, argv[3], false, true);
        if (mode == "device" || mode == "odevice") return run<cognn_shim::DevMat>(k, seed, iters, param, parties, argv[3], original);
        if (mode == "host" || mode == "ohost") return run<ShareVecVec>(k, seed, iters, param, parties, argv[3], original);
        throw std::runtime_error("mode must be device, host, idevice, odevice or ohost");
    } catch (const std::exception& ex) {
        fprintf(stderr, "gas_epochs: %s\n", ex.what());
        return 1;
    }
}
