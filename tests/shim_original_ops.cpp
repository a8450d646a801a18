// Test program (plain g++): the fused ops of the unoptimised kernel as include/cognn_sci_shim.hpp exposes them -
// sci::twoPartyGCNVectorScale with two normalisers, ForwardNN, ForwardNNPrediction, BackwardNNInit, BackwardNN
// (algo_kernels/vertex_centric/original-gcn/gcn.h:243,459,493,586,622) - run by a client (ALICE) and a server (BOB) thread over
// a LocalPipe, on device-resident tensors or on the reference's nested vectors.  Mode `fused` calls them; mode `prim` calls the
// sequence of single ops each of the four Apply ops is defined as.  Same seed, same inputs: the outputs must be identical (tests/test_shim_gpu.py).
//   usage: shim_original_ops <fused|prim> <device|host> <output file>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cognn_sci_shim.hpp"

namespace {
uint64_t rnd(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
ShareVecVec rand_mat(uint64_t& s, size_t r, size_t c, int bits) {
    ShareVecVec m(r, std::vector<uint64_t>(c));
    for (auto& row : m) for (auto& v : row) v = bits >= 64 ? rnd(s) : (uint64_t)((int64_t)(rnd(s) >> (64 - bits)) - (1ll << (bits - 1)));
    return m;
}
void put(FILE* f, const ShareVecVec& m) {
    const uint64_t r = m.size(), c = r ? m[0].size() : 0;
    fwrite(&r, 8, 1, f); fwrite(&c, 8, 1, f);
    for (auto& row : m) fwrite(row.data(), 8, c, f);
}
ShareVecVec to_host(const ShareVecVec& m) { return m; }
ShareVecVec to_host(const cognn_shim::DevMat& m) { ShareVecVec h; m.to_host(h); return h; }
template <class Mat> Mat from_host(cognn_shim::Session& s, const ShareVecVec& m);
template <> ShareVecVec from_host<ShareVecVec>(cognn_shim::Session&, const ShareVecVec& m) { return m; }
template <> cognn_shim::DevMat from_host<cognn_shim::DevMat>(cognn_shim::Session& s, const ShareVecVec& m) { return cognn_shim::DevMat::from_host(s.ctx, m); }
template <class Mat> Mat transposed(cognn_shim::Session& s, const Mat& m) { return from_host<Mat>(s, transpose(to_host(m))); }

struct Inputs { ShareVecVec x, w0, w1; std::vector<uint64_t> n0, n1; ShareVecVec label; };

template <class Mat>
void role(bool fused, int party, uint64_t tid, uint64_t coTid, const Inputs& in, std::vector<ShareVecVec>& out) {
    cognn_shim::self_tid() = tid;
    cognn_shim::Session& s = cognn_shim::session(coTid, party);
    const Mat x = from_host<Mat>(s, in.x), w0 = from_host<Mat>(s, in.w0), w1 = from_host<Mat>(s, in.w1);
    const std::vector<uint64_t> none;
    Mat sc, z0, h, z1, p, pmy, d1, g1, d0, g0;
    sci::twoPartyGCNVectorScale(x, in.n0, in.n1, sc, coTid, party);   // (one protocol call with its own streams: no single-op form)
    if (fused) sci::twoPartyGCNForwardNN(sc, w0, none, z0, h, coTid, party);
    else { sci::twoPartyGCNMatMul(sc, w0, z0, coTid, party); sci::twoPartyGCNRelu(z0, h, coTid, party); }
    if (fused) sci::twoPartyGCNForwardNNPrediction(h, w1, in.label, none, z1, p, pmy, coTid, party);
    else { sci::twoPartyGCNMatMul(h, w1, z1, coTid, party); sci::twoPartyGCNForwardNNPredictionWithoutWeight(z1, in.label, p, pmy, coTid, party); }
    const Mat ah1 = transposed(s, h), w1t = transposed(s, w1), ah0 = transposed(s, sc), w0t = transposed(s, w0);
    if (fused) sci::twoPartyGCNBackwardNNInit(pmy, ah1, w1t, none, d1, g1, coTid, party);
    else { sci::twoPartyGCNMatMul(pmy, w1t, g1, coTid, party); sci::twoPartyGCNMatMul(ah1, pmy, d1, coTid, party); }
    if (fused) sci::twoPartyGCNBackwardNN(g1, ah0, z0, w0t, none, d0, g0, true, coTid, party);
    else { Mat gz; sci::twoPartyGCNBackwardNNWithoutAH(g1, z0, w0t, gz, g0, true, coTid, party); sci::twoPartyGCNMatMul(ah0, gz, d0, coTid, party); }
    for (const Mat* m : {&sc, &z0, &h, &z1, &p, &pmy, &d1, &g1, &d0}) out.push_back(to_host(*m));
    if (!to_host(g0).empty() && !to_host(g0)[0].empty()) throw std::runtime_error("the first layer's g must be empty");
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <fused|prim> <device|host> <output>\n", argv[0]); return 2; }
    try {
        const bool fused = std::string(argv[1]) == "fused", device = std::string(argv[2]) == "device";
        const size_t n = 40, in_dim = 12, hid = 8, lab = 5;
        uint64_t s = 0x9E3779B97F4A7C15ull;
        Inputs a, b;                                          // ALICE's and BOB's views: shares of x, W0, W1; normalisers and labels are ALICE's
        const ShareVecVec x = rand_mat(s, n, in_dim, 20), w0 = rand_mat(s, in_dim, hid, 15), w1 = rand_mat(s, hid, lab, 15);
        b.x = rand_mat(s, n, in_dim, 64); b.w0 = rand_mat(s, in_dim, hid, 64); b.w1 = rand_mat(s, hid, lab, 64);
        a.x = x; a.w0 = w0; a.w1 = w1;
        for (size_t r = 0; r < n; ++r) for (size_t j = 0; j < in_dim; ++j) a.x[r][j] -= b.x[r][j];
        for (size_t r = 0; r < in_dim; ++r) for (size_t j = 0; j < hid; ++j) a.w0[r][j] -= b.w0[r][j];
        for (size_t r = 0; r < hid; ++r) for (size_t j = 0; j < lab; ++j) a.w1[r][j] -= b.w1[r][j];
        a.n0.resize(n); a.n1.resize(n); b.n0.assign(n, 0); b.n1.assign(n, 0);
        for (size_t r = 0; r < n; ++r) { a.n0[r] = 20000 + rnd(s) % 40000; a.n1[r] = r % 7 == 0 ? 0 : 30000 + rnd(s) % 30000; }
        a.label.assign(n, std::vector<uint64_t>(lab, 0)); b.label = a.label;
        for (size_t r = 0; r < n; ++r) a.label[r][rnd(s) % lab] = 1;
        cognn_shim::LocalPipe pipe;
        cognn_shim::open_session(0, 1, sci::ALICE, 77, pipe.alice());
        cognn_shim::open_session(1, 0, sci::BOB, 77, pipe.bob());
        std::vector<ShareVecVec> oa, ob;
        std::string err;
        auto guarded = [&](auto fn) { try { fn(); } catch (const std::exception& ex) { err = ex.what(); } };
        std::thread ta([&] { guarded([&] { if (device) role<cognn_shim::DevMat>(fused, sci::ALICE, 0, 1, a, oa); else role<ShareVecVec>(fused, sci::ALICE, 0, 1, a, oa); }); });
        std::thread tb([&] { guarded([&] { if (device) role<cognn_shim::DevMat>(fused, sci::BOB, 1, 0, b, ob); else role<ShareVecVec>(fused, sci::BOB, 1, 0, b, ob); }); });
        ta.join(); tb.join();
        if (!err.empty()) throw std::runtime_error(err);
        FILE* f = fopen(argv[3], "wb");
        if (!f) throw std::runtime_error("cannot open output");
        for (auto& m : oa) put(f, m);
        for (auto& m : ob) put(f, m);
        fclose(f);
        cognn_shim::close_sessions();
        return 0;
    } catch (const std::exception& ex) {
        fprintf(stderr, "shim_original_ops: %s\n", ex.what());
        return 1;
    }
}
