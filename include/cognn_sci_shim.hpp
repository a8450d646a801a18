// cognn_sci_shim.hpp — header-only drop-in for the external arithmetic layer of CoGNN's GAS callbacks.
//
// The reference's callbacks (algo_kernels/vertex_centric/optimize-gcn/gcn.h) and its engine
// (include/ss_vertex_centric_algo_kernel.h) call free functions of libraries that are not in its tree (SCIHarness.h,
// ObliviousMapper.h, SecureAggregation.h, TaskUtil.h) on nested std::vector<uint64_t> containers.  This header defines
// those containers as include/task/task.h:237,240,272 does and implements every function the optimize-gcn path calls,
// with the call-site signatures, on top of the C ABI of libcognn_hip.so (include/cognn_hip.h): upload -> HIP kernels ->
// download, the two roles of a call meeting through a Channel.  A maintainer compiles gcn.h / ss_...h against this
// header instead of the absent ones and links -lcognn_hip.
//
//   reference call site                                                         here
//   sci::twoPartyGCNMatMul(A, B, out, coTid, party)            gcn.h:233,665,671,710   Beaver product + truncation
//   sci::twoPartyGCNVectorScale(in, scale, out, signed, ..)    gcn.h:247,476           shared row scale + truncation
//   sci::twoPartyGCNCondVectorAddition(a, b, cond, out, ..)    gcn.h:456               local masked add
//   sci::twoPartyGCNRelu(in, out, tid, party)                  gcn.h:549               masked-sign ReLU
//   sci::twoPartyGCNForwardNNPredictionWithoutWeight(...)      gcn.h:578,591           softmax, p - y
//   sci::getPlainShareVecVec(p, plain, tid, party)             gcn.h:604               reveal
//   sci::twoPartyGCNMatrixScale(m, fx, out, tid, party)        gcn.h:676,723,764       public constant * share, truncation
//   sci::twoPartyGCNApplyGradient(W, d, fxLr, Wout, ..)        gcn.h:678,730           W - trunc(lr * d)
//   sci::twoPartyGCNBackwardNNWithoutAH(in, z, wT, out, g, first, ..)  gcn.h:705       in (.) 1[z > 0]  (+ g = out . wT)
//   sci::plaintext_add_matrix[_in_place](a, b)                 gcn.h:758,762           local add
//   sci::cross_entropy_loss / accuracy / count_true            gcn.h:620-632           host metrics on the revealed p
//   prefix_network_aggregate(pos, svv, ADD_AGG, coTid, party, b)   gcn.h:328-335       segmented inclusive prefix sum
//   client_/server_oblivious_mapper_online(...)                ss_...h:752-854,1011-1076   row gather by position
//   CryptoUtil::{intoShares, mergeShareAsDouble, encodeDoubleAsFixedPoint}   gcn.h:70,80,220
//   transpose / toShareVec(hot, n)                             task.h:243-249
//
// Share-arithmetic definitions: DESIGN.md §3 (ring Z_2^64, f = 16, counter-PRNG dealer, Beaver products, dealer-assisted
// truncation, masked-sign ReLU, integer softmax).  NOT the reference's privacy: the two roles exchange index structure in
// the clear (positions of the oblivious mapper, segment boundaries) and ReLU signs are public.
//
// Dealer addressing: a session numbers its protocol calls 0,1,2,...; call number c of the pair owned by tid `owner` draws
// its masks from (seed, owner, c, op) - both roles issue the same sequence of calls (they do in the reference: client and
// server threads mirror each other), so their counters agree without any message.
#ifndef COGNN_SCI_SHIM_HPP_
#define COGNN_SCI_SHIM_HPP_
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "cognn_hip.h"

// ---- containers (include/task/task.h:237,239,240,272) ---------------------------------------------------------------------
typedef std::vector<uint64_t> ShareVec;
typedef std::vector<std::vector<double>> DoubleTensor;
typedef std::vector<std::vector<uint64_t>> ShareTensor;
typedef std::vector<ShareTensor> ShareTensorVec;
typedef std::vector<ShareVec> ShareVecVec;

#ifndef SCALER_BIT_LENGTH
#define SCALER_BIT_LENGTH 16            /* external in the reference (only "< 31" is pinned, gcn.h:191); DESIGN.md §3.1 */
#endif

namespace cognn_shim {

// dealer op ids / slots of cognn_amd/csrc/cognn_spec.h (kept numerically identical; checked by tests/test_shim_gpu.py)
enum { OP_GEMM = 10, OP_GEMM_TRUNC = 11, OP_SCALE = 12, OP_SCALE_TRUNC = 13, OP_RELU = 16, OP_SOFTMAX = 17, OP_MSCALE_TRUNC = 20,
       OP_LR_TRUNC = 21, OP_SHARE = 1 };
enum { SL_A0 = 0, SL_A1, SL_B0, SL_B1, SL_C0 };

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc, const char* what) {
    if (rc != 0) throw Error(std::string(what) + ": " + cognn_last_error());
}

// One round between the two roles of a pair: each side hands over `send` and receives what the other side handed over.
struct Channel {
    virtual ~Channel() {}
    virtual void exchange(const void* send, size_t send_bytes, void* recv, size_t recv_bytes) = 0;
};

// Both roles in one process (two threads): a rendezvous in host memory.  The remote case is a Channel over whatever
// transport the two processes share (the reference's TaskComm / CommSync sockets, or cognn_exchange.h for device buffers).
class LocalPipe {
public:
    class End : public Channel {
    public:
        End(LocalPipe* p, int me) : pipe_(p), me_(me) {}
        void exchange(const void* send, size_t sb, void* recv, size_t rb) override {
            std::unique_lock<std::mutex> lk(pipe_->m_);
            pipe_->cv_.wait(lk, [&] { return !pipe_->full_[me_]; });
            pipe_->buf_[me_].assign((const uint8_t*)send, (const uint8_t*)send + sb);
            pipe_->full_[me_] = true;
            pipe_->cv_.notify_all();
            pipe_->cv_.wait(lk, [&] { return pipe_->full_[1 - me_]; });
            if (pipe_->buf_[1 - me_].size() != rb) {
                pipe_->full_[1 - me_] = false;
                pipe_->cv_.notify_all();
                throw Error("cognn_shim: the two roles disagree on a message size (protocol calls out of step)");
            }
            if (rb) std::memcpy(recv, pipe_->buf_[1 - me_].data(), rb);
            pipe_->full_[1 - me_] = false;
            pipe_->cv_.notify_all();
        }
    private:
        LocalPipe* pipe_;
        int me_;
    };
    LocalPipe() : a_(this, 0), b_(this, 1) {}
    Channel* alice() { return &a_; }
    Channel* bob() { return &b_; }
private:
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<uint8_t> buf_[2];
    bool full_[2] = {false, false};
    End a_, b_;
};

// device buffer of the C ABI
class Dev {
public:
    Dev(cognn_ctx* c, size_t elems, size_t elem_bytes = 8) : ctx_(c), bytes_(elems * elem_bytes) {
        check(cognn_malloc(ctx_, &p_, bytes_ ? bytes_ : 16), "cognn_malloc");
    }
    ~Dev() { cognn_free(ctx_, p_); }
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    uint64_t* u64() const { return (uint64_t*)p_; }
    void* ptr() const { return p_; }
    void up(const void* host) { if (bytes_) check(cognn_memcpy_h2d(ctx_, p_, host, bytes_), "cognn_memcpy_h2d"); }
    void down(void* host) const { if (bytes_) check(cognn_memcpy_d2h(ctx_, host, p_, bytes_), "cognn_memcpy_d2h"); }
    size_t bytes() const { return bytes_; }
private:
    cognn_ctx* ctx_;
    void* p_ = nullptr;
    size_t bytes_;
};

inline std::vector<uint64_t> flatten(const ShareVecVec& v, size_t* rows, size_t* cols) {
    *rows = v.size();
    *cols = v.empty() ? 0 : v[0].size();
    std::vector<uint64_t> f(*rows * *cols);
    for (size_t r = 0; r < *rows; ++r) {
        if (v[r].size() != *cols) throw Error("cognn_shim: ragged ShareVecVec");
        std::memcpy(f.data() + r * *cols, v[r].data(), *cols * 8);
    }
    return f;
}
inline void unflatten(const std::vector<uint64_t>& f, size_t rows, size_t cols, ShareVecVec& out) {
    ShareVecVec o(rows, ShareVec(cols));
    for (size_t r = 0; r < rows; ++r) std::memcpy(o[r].data(), f.data() + r * cols, cols * 8);
    out.swap(o);
}

// One role (ALICE = owner share p 0, BOB = co-party share p 1) of one pair.
struct Session {
    cognn_ctx* ctx = nullptr;
    int p = 0;
    uint64_t seed = 0, owner = 0, counter = 0;
    Channel* ch = nullptr;

    cognn_keys keys(int op) const {
        cognn_keys k;
        cognn_make_keys(seed, owner, counter, (uint64_t)op, &k);
        return k;
    }
    // d2h mine, swap with the other role, h2d theirs
    void swap_dev(const Dev& mine, Dev& theirs) {
        std::vector<uint8_t> a(mine.bytes()), b(theirs.bytes());
        mine.down(a.data());
        ch->exchange(a.data(), a.size(), b.data(), b.size());
        theirs.up(b.data());
    }
    // truncation by 2^f of `x` times the public constant `mul`; mode 1: out = out - result
    void trunc(Dev& out, const Dev& x, uint64_t mul, int op, int64_t n, int mode = 0) {
        cognn_keys tk = keys(op);
        Dev c(ctx, (size_t)n), cp(ctx, (size_t)n);
        check(cognn_trunc_open_u64(ctx, c.u64(), x.u64(), mul, &tk, p, n), "cognn_trunc_open_u64");
        swap_dev(c, cp);
        check(cognn_trunc_close_u64(ctx, out.u64(), p == 0 ? c.u64() : nullptr, p == 0 ? cp.u64() : nullptr, &tk, p, mode, n),
              "cognn_trunc_close_u64");
    }
};

struct Registry {
    std::mutex m;
    std::map<std::tuple<uint64_t, uint64_t, int>, std::unique_ptr<Session>> sessions;   // (self tid, co tid, party)
    static Registry& get() { static Registry r; return r; }
};
// The reference runs one party per process, so (coTid, party) names a session; tests that host several parties in one
// process set the calling thread's own tid first.
inline uint64_t& self_tid() { static thread_local uint64_t t = 0; return t; }

// party: sci::ALICE (1) or sci::BOB (2).  `stream`: hipStream_t of the calling thread (NULL = default stream).
inline void open_session(uint64_t selfTid, uint64_t coTid, int party, uint64_t seed, Channel* ch, int device = 0, void* stream = nullptr) {
    std::unique_ptr<Session> s(new Session());
    check(cognn_ctx_create(device, stream, &s->ctx), "cognn_ctx_create");
    s->p = party - 1;
    s->seed = seed;
    s->owner = party == 1 ? selfTid : coTid;                 // the data owner is the ALICE side of the pair
    s->ch = ch;
    Registry& r = Registry::get();
    std::lock_guard<std::mutex> lk(r.m);
    r.sessions[std::make_tuple(selfTid, coTid, party)] = std::move(s);
}
inline void close_sessions() {
    Registry& r = Registry::get();
    std::lock_guard<std::mutex> lk(r.m);
    for (auto& kv : r.sessions) cognn_ctx_destroy(kv.second->ctx);
    r.sessions.clear();
}
inline Session& session(uint64_t coTid, int party) {
    Registry& r = Registry::get();
    std::lock_guard<std::mutex> lk(r.m);
    auto f = r.sessions.find(std::make_tuple(self_tid(), coTid, party));
    if (f == r.sessions.end()) throw Error("cognn_shim: no session for this (tid, coTid, party); call cognn_shim::open_session first");
    return *f->second;
}

// row gather out[r] = sum_{q in rowptr[r]..rowptr[r+1]} table[col[q]] (+ base[r]) through cognn_gather_csr_u64
inline void gather_rows(Session& s, const ShareVecVec& table, const ShareVecVec* base, const std::vector<uint32_t>& rowptr,
                        const std::vector<uint32_t>& col, size_t width, ShareVecVec& out) {
    size_t tr, tc;
    std::vector<uint64_t> ft = flatten(table, &tr, &tc);
    if (tr && tc != width) throw Error("cognn_shim: row width mismatch");
    const size_t rows = rowptr.size() - 1;
    Dev dt(s.ctx, tr * width), dout(s.ctx, rows * width), drp(s.ctx, rowptr.size(), 4), dcol(s.ctx, col.size(), 4);
    dt.up(ft.data()); drp.up(rowptr.data()); dcol.up(col.data());
    std::unique_ptr<Dev> db;
    if (base) {
        size_t br, bc;
        std::vector<uint64_t> fb = flatten(*base, &br, &bc);
        if (br != rows || (br && bc != width)) throw Error("cognn_shim: base shape mismatch");
        db.reset(new Dev(s.ctx, br * width));
        db->up(fb.data());
    }
    if (rows && width)
        check(cognn_gather_csr_u64(s.ctx, dout.u64(), db ? db->u64() : nullptr, dt.u64(), (const uint32_t*)drp.ptr(), (const uint32_t*)dcol.ptr(),
                                   (int64_t)rows, (int64_t)width), "cognn_gather_csr_u64");
    std::vector<uint64_t> fo(rows * width);
    dout.down(fo.data());
    unflatten(fo, rows, width, out);
}

// the position arrays travel from the client to the server in the clear (see header comment)
inline void share_positions(Session& s, std::vector<uint64_t>& a, std::vector<uint64_t>& b) {
    if (s.p == 0) {
        uint64_t n[2] = {a.size(), b.size()};
        s.ch->exchange(n, sizeof(n), nullptr, 0);
        std::vector<uint64_t> both(a);
        both.insert(both.end(), b.begin(), b.end());
        s.ch->exchange(both.data(), both.size() * 8, nullptr, 0);
    } else {
        uint64_t n[2];
        s.ch->exchange(nullptr, 0, n, sizeof(n));
        std::vector<uint64_t> both(n[0] + n[1]);
        s.ch->exchange(nullptr, 0, both.data(), both.size() * 8);
        a.assign(both.begin(), both.begin() + n[0]);
        b.assign(both.begin() + n[0], both.end());
    }
}

inline void mapper(Session& s, std::vector<uint64_t> srcPos, std::vector<uint64_t> dstPos, const ShareVecVec& src, ShareVecVec& dst,
                   bool allowMissing) {
    share_positions(s, srcPos, dstPos);
    std::map<uint64_t, uint32_t> last;                       // dst[r] = src[last q with srcPos[q] == dstPos[r]]
    for (size_t q = 0; q < srcPos.size(); ++q) last[srcPos[q]] = (uint32_t)q;
    std::vector<uint32_t> rowptr{0}, col;
    for (uint64_t d : dstPos) {
        auto f = last.find(d);
        if (f != last.end()) col.push_back(f->second);
        else if (!allowMissing) throw Error("cognn_shim: oblivious mapper: destination position missing in the source");
        rowptr.push_back((uint32_t)col.size());
    }
    gather_rows(s, src, nullptr, rowptr, col, src.empty() ? 0 : src[0].size(), dst);
}

}  // namespace cognn_shim

// ---- free functions of include/task/task.h:243-249 (defined externally in the reference) --------------------------------
inline ShareTensor transpose(const ShareTensor& st) {
    const size_t r = st.size(), c = r ? st[0].size() : 0;
    ShareTensor t(c, ShareVec(r));
    for (size_t i = 0; i < r; ++i)
        for (size_t j = 0; j < c; ++j) t[j][i] = st[i][j];
    return t;
}
inline ShareVec toShareVec(int hotIndex, int vecSize) {
    ShareVec v((size_t)vecSize, 0);
    if (hotIndex >= 0 && hotIndex < vecSize) v[(size_t)hotIndex] = 1ull << SCALER_BIT_LENGTH;
    return v;
}

// ---- CryptoUtil (TaskUtil.h in the reference; call sites gcn.h:70,80,220) ------------------------------------------------
class CryptoUtil {
public:
    static uint64_t encodeDoubleAsFixedPoint(double v) { return (uint64_t)(int64_t)std::llround(v * (double)(1ull << SCALER_BIT_LENGTH)); }
    static double mergeShareAsDouble(uint64_t s0, uint64_t s1) { return (double)(int64_t)(s0 + s1) / (double)(1ull << SCALER_BIT_LENGTH); }
    // share 1 is the next value of the sharing stream (seed set with sharingSeedIs), share 0 the difference
    static void intoShares(double v, uint64_t& s0, uint64_t& s1) {
        cognn_keys k;
        cognn_make_keys(state().seed, state().tag, 0, cognn_shim::OP_SHARE, &k);
        // prng(key, idx) of cognn_amd/csrc/cognn_spec.h restated for host use: four multiply-fold rounds on z = idx ^ key
        const uint64_t key = k.k[0], hi = 0xFFFFFFFF00000000ull;
        const uint32_t mul[4] = {0x97D6730Eu, 0xC140D344u, 0xF849CBC2u, 0xEC6F6B54u};
        uint64_t z = state().idx++ ^ key;
        for (int i = 0; i < 4; ++i) {
            z += (uint64_t)(uint32_t)z * mul[i];
            if (i < 3) z = (z & hi) | (uint32_t)(((uint32_t)z ^ (uint32_t)(z >> 32)) + (i == 1 ? (uint32_t)(key >> 32) : 0u));
        }
        s1 = z;
        s0 = encodeDoubleAsFixedPoint(v) - s1;
    }
    static void sharingSeedIs(uint64_t seed, uint64_t tag) { state().seed = seed; state().tag = tag; state().idx = 0; }
private:
    struct State { uint64_t seed = 0, tag = 0, idx = 0; };
    static State& state() { static thread_local State s; return s; }
};

// ---- SecureAggregation.h: prefix_network_aggregate (gcn.h:328-335) --------------------------------------------------------
enum class AggregationOp { ADD_AGG };
inline ShareVecVec prefix_network_aggregate(std::vector<uint64_t> pos, const ShareVecVec& svv, AggregationOp, uint64_t coTid, int party, bool) {
    cognn_shim::Session& s = cognn_shim::session(coTid, party);
    std::vector<uint64_t> none;
    cognn_shim::share_positions(s, pos, none);               // the server passes a zero vector (ss_...h:1047): it learns the runs here
    if (pos.size() != svv.size()) throw cognn_shim::Error("prefix_network_aggregate: one position per row expected");
    // inclusive prefix sum inside each run of equal consecutive positions: row q gathers rows start(q)..q
    std::vector<uint32_t> rowptr{0}, col;
    size_t start = 0;
    for (size_t q = 0; q < pos.size(); ++q) {
        if (q && pos[q] != pos[q - 1]) start = q;
        for (size_t j = start; j <= q; ++j) col.push_back((uint32_t)j);
        rowptr.push_back((uint32_t)col.size());
    }
    ShareVecVec out;
    cognn_shim::gather_rows(s, svv, nullptr, rowptr, col, svv.empty() ? 0 : svv[0].size(), out);
    return out;
}

// ---- ObliviousMapper.h (ss_...h:752-763,818,848 client; :1011-1016,1057,1075 server) ---------------------------------------
inline void client_oblivious_mapper_online(const std::vector<uint64_t>& srcPos, const std::vector<uint64_t>& dstPos, const ShareVecVec& srcSvv,
                                           ShareVecVec& dstSvv, uint32_t /*plainNumPerOperand*/, uint64_t /*iter*/, uint32_t /*preprocessId*/,
                                           uint64_t coTid, bool allowMissing = false) {
    cognn_shim::Session& s = cognn_shim::session(coTid, 1);
    if (allowMissing) { uint64_t f = 1; s.ch->exchange(&f, 8, nullptr, 0); } else { uint64_t f = 0; s.ch->exchange(&f, 8, nullptr, 0); }
    cognn_shim::mapper(s, srcPos, dstPos, srcSvv, dstSvv, allowMissing);
}
inline void server_oblivious_mapper_online(const ShareVecVec& srcSvv, ShareVecVec& dstSvv, uint64_t /*iter*/, uint32_t /*preprocessId*/, uint64_t coTid) {
    cognn_shim::Session& s = cognn_shim::session(coTid, 2);
    uint64_t f = 0;
    s.ch->exchange(nullptr, 0, &f, 8);
    cognn_shim::mapper(s, {}, {}, srcSvv, dstSvv, f != 0);
}

// ---- SCIHarness.h -----------------------------------------------------------------------------------------------------------
namespace sci {
enum { ALICE = 1, BOB = 2 };
using cognn_shim::Dev;
using cognn_shim::Session;
using cognn_shim::check;

inline void twoPartyGCNMatMul(const ShareVecVec& A, const ShareTensor& B, ShareVecVec& out, uint64_t coTid, int party) {
    Session& s = cognn_shim::session(coTid, party);
    size_t M, K, K2, N;
    std::vector<uint64_t> fa = cognn_shim::flatten(A, &M, &K), fb = cognn_shim::flatten(B, &K2, &N);
    if (K != K2) throw cognn_shim::Error("twoPartyGCNMatMul: inner dimensions differ");
    const int p = s.p;
    cognn_keys k = s.keys(cognn_shim::OP_GEMM);
    Dev dA(s.ctx, M * K), dB(s.ctx, K * N), E(s.ctx, M * K), Ep(s.ctx, M * K), F(s.ctx, K * N), Fp(s.ctx, K * N), Z(s.ctx, M * N),
        scratch(s.ctx, M * K + K * N);
    dA.up(fa.data()); dB.up(fb.data());
    check(cognn_mask_open_u64(s.ctx, E.u64(), dA.u64(), k.k[p ? cognn_shim::SL_A1 : cognn_shim::SL_A0], (int64_t)M, (int64_t)K, 0), "cognn_mask_open_u64");
    check(cognn_mask_open_u64(s.ctx, F.u64(), dB.u64(), k.k[p ? cognn_shim::SL_B1 : cognn_shim::SL_B0], (int64_t)K, (int64_t)N, 0), "cognn_mask_open_u64");
    s.swap_dev(E, Ep);                                       // Beaver reveal
    s.swap_dev(F, Fp);
    check(cognn_add_u64(s.ctx, F.u64(), F.u64(), Fp.u64(), (int64_t)(K * N)), "cognn_add_u64");
    std::unique_ptr<Dev> c1;
    if (p == 1) {                                            // the dealer's product share (offline phase; dealt in place here)
        c1.reset(new Dev(s.ctx, M * N));
        check(cognn_dealer_gemm_c1_u64(s.ctx, c1->u64(), &k, (int64_t)M, (int64_t)N, (int64_t)K, 0, scratch.u64(), scratch.u64() + M * K),
              "cognn_dealer_gemm_c1_u64");
    }
    check(cognn_beaver_gemm_close_u64(s.ctx, Z.u64(), E.u64(), Ep.u64(), F.u64(), c1 ? c1->u64() : nullptr, &k, p, (int64_t)M, (int64_t)N,
                                      (int64_t)K, 0, scratch.u64()), "cognn_beaver_gemm_close_u64");
    Dev O(s.ctx, M * N);
    s.trunc(O, Z, 1, cognn_shim::OP_GEMM_TRUNC, (int64_t)(M * N));
    std::vector<uint64_t> fo(M * N);
    O.down(fo.data());
    cognn_shim::unflatten(fo, M, N, out);
    ++s.counter;
}

inline void twoPartyGCNVectorScale(const ShareVecVec& in, const std::vector<uint64_t>& rowScale, ShareVecVec& out, bool /*isSigned*/, uint64_t coTid,
                                   int party) {
    Session& s = cognn_shim::session(coTid, party);
    size_t n, F;
    std::vector<uint64_t> fi = cognn_shim::flatten(in, &n, &F);
    if (rowScale.size() != n) throw cognn_shim::Error("twoPartyGCNVectorScale: one scale per row expected");
    cognn_keys k = s.keys(cognn_shim::OP_SCALE), tk = s.keys(cognn_shim::OP_SCALE_TRUNC);
    Dev V(s.ctx, n * F), S(s.ctx, n), E(s.ctx, n * F), Ep(s.ctx, n * F), G(s.ctx, n), Gp(s.ctx, n), c(s.ctx, n * F), cp(s.ctx, n * F), O(s.ctx, n * F);
    V.up(fi.data()); S.up(rowScale.data());
    check(cognn_rowscale_open_u64(s.ctx, E.u64(), G.u64(), V.u64(), S.u64(), &k, s.p, (int64_t)n, (int64_t)F), "cognn_rowscale_open_u64");
    s.swap_dev(E, Ep);
    s.swap_dev(G, Gp);
    check(cognn_rowscale_close_u64(s.ctx, c.u64(), E.u64(), Ep.u64(), G.u64(), Gp.u64(), &k, &tk, s.p, (int64_t)n, (int64_t)F), "cognn_rowscale_close_u64");
    s.swap_dev(c, cp);
    check(cognn_trunc_close_u64(s.ctx, O.u64(), s.p == 0 ? c.u64() : nullptr, s.p == 0 ? cp.u64() : nullptr, &tk, s.p, 0, (int64_t)(n * F)),
          "cognn_trunc_close_u64");
    std::vector<uint64_t> fo(n * F);
    O.down(fo.data());
    cognn_shim::unflatten(fo, n, F, out);
    ++s.counter;
}

// out = a + (cond ? b : 0) row by row.  The condition is the client's private input - the server passes a placeholder
// (zeroIsDummy, ss_...h:1124-1126) - so ALICE's vector is the one both roles apply; here it reaches BOB in the clear.
inline void twoPartyGCNCondVectorAddition(ShareVecVec& a, ShareVecVec& b, std::vector<bool>& cond, ShareVecVec& out, uint64_t coTid, int party) {
    Session& s = cognn_shim::session(coTid, party);
    if (a.size() != b.size() || cond.size() != a.size()) throw cognn_shim::Error("twoPartyGCNCondVectorAddition: shape mismatch");
    std::vector<uint8_t> c(cond.size());
    if (s.p == 0) {
        for (size_t r = 0; r < cond.size(); ++r) c[r] = cond[r] ? 1 : 0;
        s.ch->exchange(c.data(), c.size(), nullptr, 0);
    } else {
        s.ch->exchange(nullptr, 0, c.data(), c.size());
    }
    std::vector<uint32_t> rowptr{0}, col;
    for (size_t r = 0; r < a.size(); ++r) {
        if (c[r]) col.push_back((uint32_t)r);
        rowptr.push_back((uint32_t)col.size());
    }
    ShareVecVec res;
    cognn_shim::gather_rows(s, b, &a, rowptr, col, a.empty() ? 0 : a[0].size(), res);
    out.swap(res);
}

// masked-sign ReLU; returns the public sign mask through `mask` when asked
inline void relu_core(Session& s, const std::vector<uint64_t>& fz, size_t n, std::vector<uint64_t>* h, std::vector<uint8_t>* mask) {
    cognn_keys k = s.keys(cognn_shim::OP_RELU);
    Dev z(s.ctx, n), E(s.ctx, n), Ep(s.ctx, n), w(s.ctx, n), wp(s.ctx, n), H(s.ctx, n), Mk(s.ctx, n, 1);
    z.up(fz.data());
    check(cognn_relu_open_u64(s.ctx, E.u64(), nullptr, z.u64(), &k, s.p, (int64_t)n), "cognn_relu_open_u64");
    s.swap_dev(E, Ep);
    check(cognn_relu_mul_u64(s.ctx, w.u64(), E.u64(), Ep.u64(), nullptr, nullptr, &k, s.p, (int64_t)n), "cognn_relu_mul_u64");
    s.swap_dev(w, wp);
    check(cognn_relu_close_u64(s.ctx, H.u64(), (uint8_t*)Mk.ptr(), z.u64(), w.u64(), wp.u64(), (int64_t)n), "cognn_relu_close_u64");
    if (h) { h->resize(n); H.down(h->data()); }
    if (mask) { mask->resize(n); Mk.down(mask->data()); }
    ++s.counter;
}

inline void twoPartyGCNRelu(const ShareVecVec& in, ShareTensor& out, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    size_t n, F;
    std::vector<uint64_t> fz = cognn_shim::flatten(in, &n, &F), h;
    relu_core(s, fz, n * F, &h, nullptr);
    cognn_shim::unflatten(h, n, F, out);
}

inline void twoPartyGCNBackwardNNWithoutAH(const ShareVecVec& in, const ShareTensor& z, const ShareTensor& weightT, ShareVecVec& dstVec, ShareTensor& g,
                                           bool isFirstLayer, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    size_t n, F, n2, F2;
    std::vector<uint64_t> fin = cognn_shim::flatten(in, &n, &F), fz = cognn_shim::flatten(z, &n2, &F2);
    if (n != n2 || F != F2) throw cognn_shim::Error("twoPartyGCNBackwardNNWithoutAH: shape mismatch");
    std::vector<uint8_t> mask;
    relu_core(s, fz, n * F, nullptr, &mask);                 // 1[z > 0], public
    Dev I(s.ctx, n * F), O(s.ctx, n * F), Mk(s.ctx, n * F, 1);
    I.up(fin.data()); Mk.up(mask.data());
    check(cognn_mask_select_u64(s.ctx, O.u64(), I.u64(), (const uint8_t*)Mk.ptr(), (int64_t)(n * F)), "cognn_mask_select_u64");
    std::vector<uint64_t> fo(n * F);
    O.down(fo.data());
    ShareVecVec masked;
    cognn_shim::unflatten(fo, n, F, masked);
    if (!isFirstLayer) twoPartyGCNMatMul(masked, weightT, g, tid, party);      // gradient for the layer below; skipped for the first layer
    else g.clear();
    dstVec.swap(masked);
}

inline void twoPartyGCNForwardNNPredictionWithoutWeight(const ShareVecVec& z, const ShareVecVec& label, ShareTensor& p, ShareTensor& p_minus_y,
                                                        uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    size_t n, L;
    std::vector<uint64_t> fz = cognn_shim::flatten(z, &n, &L);
    cognn_keys k = s.keys(cognn_shim::OP_SOFTMAX);
    Dev Z(s.ctx, n * L), Zp(s.ctx, s.p == 0 ? n * L : 0), P(s.ctx, n * L), D(s.ctx, n * L), PF(s.ctx, n * L), Lb(s.ctx, n, 4);
    Z.up(fz.data());
    if (s.p == 0) {                                          // the owner receives the co-party's share of z (it learns p anyway, gcn.h:603-604)
        std::vector<uint64_t> zp(n * L);
        s.ch->exchange(nullptr, 0, zp.data(), zp.size() * 8);
        Zp.up(zp.data());
        std::vector<int32_t> lab(n, 0);
        for (size_t r = 0; r < n; ++r)
            for (size_t j = 0; j < L && r < label.size(); ++j)
                if (label[r][j] != 0) lab[r] = (int32_t)j;
        Lb.up(lab.data());
        check(cognn_softmax_u64(s.ctx, P.u64(), D.u64(), PF.u64(), Z.u64(), Zp.u64(), (const int32_t*)Lb.ptr(), &k, 0, (int64_t)n, (int64_t)L, (int64_t)n),
              "cognn_softmax_u64");
    } else {
        s.ch->exchange(fz.data(), fz.size() * 8, nullptr, 0);
        check(cognn_softmax_u64(s.ctx, P.u64(), D.u64(), nullptr, nullptr, nullptr, nullptr, &k, 1, (int64_t)n, (int64_t)L, (int64_t)n), "cognn_softmax_u64");
    }
    std::vector<uint64_t> fp(n * L), fd(n * L);
    P.down(fp.data()); D.down(fd.data());
    cognn_shim::unflatten(fp, n, L, p);
    cognn_shim::unflatten(fd, n, L, p_minus_y);
    ++s.counter;
}

inline void getPlainShareVecVec(const ShareVecVec& sv, DoubleTensor& plain, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    size_t n, L;
    std::vector<uint64_t> f = cognn_shim::flatten(sv, &n, &L), o(n * L);
    s.ch->exchange(f.data(), f.size() * 8, o.data(), o.size() * 8);
    plain.assign(n, std::vector<double>(L));
    for (size_t r = 0; r < n; ++r)
        for (size_t j = 0; j < L; ++j) plain[r][j] = CryptoUtil::mergeShareAsDouble(f[r * L + j], o[r * L + j]);
}

inline void twoPartyGCNMatrixScale(const ShareVecVec& m, uint64_t fxScale, ShareVecVec& out, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    size_t r, c;
    std::vector<uint64_t> f = cognn_shim::flatten(m, &r, &c);
    Dev X(s.ctx, r * c), O(s.ctx, r * c);
    X.up(f.data());
    s.trunc(O, X, fxScale, cognn_shim::OP_MSCALE_TRUNC, (int64_t)(r * c));
    O.down(f.data());
    cognn_shim::unflatten(f, r, c, out);
    ++s.counter;
}

inline void twoPartyGCNApplyGradient(const ShareVecVec& W, const ShareVecVec& d, uint64_t fxLr, ShareVecVec& Wout, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    size_t r, c, r2, c2;
    std::vector<uint64_t> fw = cognn_shim::flatten(W, &r, &c), fd = cognn_shim::flatten(d, &r2, &c2);
    if (r != r2 || c != c2) throw cognn_shim::Error("twoPartyGCNApplyGradient: shape mismatch");
    Dev Wd(s.ctx, r * c), D(s.ctx, r * c);
    Wd.up(fw.data()); D.up(fd.data());
    s.trunc(Wd, D, fxLr, cognn_shim::OP_LR_TRUNC, (int64_t)(r * c), 1);        // W -= trunc(lr * d)
    Wd.down(fw.data());
    cognn_shim::unflatten(fw, r, c, Wout);
    ++s.counter;
}

inline void plaintext_add_matrix_in_place(ShareVecVec& a, const ShareVecVec& b) {
    if (a.size() != b.size()) throw cognn_shim::Error("plaintext_add_matrix_in_place: shape mismatch");
    for (size_t r = 0; r < a.size(); ++r)
        for (size_t j = 0; j < a[r].size(); ++j) a[r][j] += b[r][j];
}
inline ShareVecVec plaintext_add_matrix(const ShareVecVec& a, const ShareVecVec& b) {
    ShareVecVec o(a);
    plaintext_add_matrix_in_place(o, b);
    return o;
}

// metrics on the revealed probabilities (gcn.h:611-632)
inline double cross_entropy_loss(const DoubleTensor& y, const DoubleTensor& p) {
    double loss = 0;
    for (size_t r = 0; r < y.size(); ++r)
        for (size_t j = 0; j < y[r].size(); ++j)
            if (y[r][j] != 0) loss -= y[r][j] * std::log(p[r][j]);
    return y.empty() ? 0.0 : loss / (double)y.size();
}
inline size_t argmax(const std::vector<double>& v) {
    size_t b = 0;
    for (size_t j = 1; j < v.size(); ++j) if (v[j] > v[b]) b = j;
    return b;
}
// per cent, like the reference's log (README.md:226-236: "full set accuracy = 19.188192" = 104 of 542 vertices)
inline double accuracy(const DoubleTensor& y, const DoubleTensor& p) {
    size_t ok = 0;
    for (size_t r = 0; r < y.size(); ++r) ok += argmax(y[r]) == argmax(p[r]);
    return y.empty() ? 0.0 : 100.0 * (double)ok / (double)y.size();
}
inline double accuracy(const DoubleTensor& y, const DoubleTensor& p, const std::vector<bool>& sel) {
    size_t ok = 0, n = 0;
    for (size_t r = 0; r < y.size(); ++r)
        if (sel[r]) { ++n; ok += argmax(y[r]) == argmax(p[r]); }
    return n ? 100.0 * (double)ok / (double)n : 0.0;
}
inline size_t count_true(const std::vector<bool>& v) {
    size_t n = 0;
    for (bool b : v) n += b;
    return n;
}

}  // namespace sci
#endif  // COGNN_SCI_SHIM_HPP_
