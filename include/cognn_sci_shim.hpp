// cognn_sci_shim.hpp — header-only drop-in for the external arithmetic layer of CoGNN's GAS callbacks.
//
// The reference's callbacks (algo_kernels/vertex_centric/optimize-gcn/gcn.h) and its engine
// (include/ss_vertex_centric_algo_kernel.h) call free functions of libraries that are not in its tree (SCIHarness.h,
// ObliviousMapper.h, SecureAggregation.h, TaskUtil.h) on nested std::vector<uint64_t> containers.  This header defines
// those containers as include/task/task.h:237,240,272 does and implements every function the optimize-gcn path calls,
// with the call-site signatures, on top of the C ABI of libcognn_hip.so (include/cognn_hip.h).  A maintainer compiles
// gcn.h / ss_...h against this header instead of the absent ones and links -lcognn_hip.
//
// Every function exists for TWO container types:
//   * ShareVecVec (nested host vectors, exactly the reference's type): upload -> HIP kernels -> download per call;
//   * cognn_shim::DevMat (a [rows x cols] uint64 tensor resident in device memory): no tensor leaves the device - the two
//     roles of a call hand their openings over device to device (Channel::exchange_device), index structure (oblivious-mapper
//     positions, aggregation runs, masks, normalisers) is turned into device arrays once and cached per session.
//   include/cognn_gas_kernel.hpp (the GAS callbacks of ss_...h:78-133 as a class template over the container type) is built on
//   the second form.
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
//   sci::twoPartyGCNVectorScale(in, n0, n1, out, ..) / ForwardNN / ForwardNNPrediction / BackwardNNInit / BackwardNN
//                                                              original-gcn/gcn.h:243,459,493,586,622   compositions of the above
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
    // The same round on DEVICE buffers.  Default: staged through host memory (any transport that moves host bytes works);
    // a transport that can move device memory itself (the in-process pipe below, cognn_exchange.h between GPUs) overrides it.
    virtual void exchange_device(cognn_ctx* ctx, const void* send_dev, size_t send_bytes, void* recv_dev, size_t recv_bytes) {
        std::vector<uint8_t> a(send_bytes), b(recv_bytes);
        if (send_bytes) check(cognn_memcpy_d2h(ctx, a.data(), send_dev, send_bytes), "cognn_memcpy_d2h");
        exchange(a.data(), send_bytes, b.data(), recv_bytes);
        if (recv_bytes) check(cognn_memcpy_h2d(ctx, recv_dev, b.data(), recv_bytes), "cognn_memcpy_h2d");
    }
};

// Both roles in one process (two threads): a rendezvous.  Host rounds copy through the sender's buffer, device rounds are one
// device-to-device copy per side.  The remote case is a Channel over whatever transport the two processes share (the
// reference's TaskComm / CommSync sockets, or cognn_exchange.h for device buffers).
class LocalPipe {
public:
    class End : public Channel {
    public:
        End(LocalPipe* p, int me) : pipe_(p), me_(me) {}
        void exchange(const void* send, size_t sb, void* recv, size_t rb) override { round(nullptr, send, sb, recv, rb); }
        void exchange_device(cognn_ctx* ctx, const void* send_dev, size_t sb, void* recv_dev, size_t rb) override {
            check(cognn_ctx_sync(ctx), "cognn_ctx_sync");            // what is handed over is complete
            round(ctx, send_dev, sb, recv_dev, rb);
        }
    private:
        // both sides publish their send buffer, meet, copy the other side's buffer, meet again (then a sender may reuse its buffer)
        void round(cognn_ctx* ctx, const void* send, size_t sb, void* recv, size_t rb) {
            LocalPipe& P = *pipe_;
            std::unique_lock<std::mutex> lk(P.m_);
            P.ptr_[me_] = send; P.bytes_[me_] = sb;
            P.meet(lk);
            const void* theirs = P.ptr_[1 - me_];
            const bool ok = P.bytes_[1 - me_] == rb;
            lk.unlock();
            if (ok && rb) {
                if (ctx) { check(cognn_memcpy_d2d(ctx, recv, theirs, rb), "cognn_memcpy_d2d"); check(cognn_ctx_sync(ctx), "cognn_ctx_sync"); }
                else std::memcpy(recv, theirs, rb);
            }
            lk.lock();
            P.meet(lk);
            if (!ok) throw Error("cognn_shim: the two roles disagree on a message size (protocol calls out of step)");
        }
        LocalPipe* pipe_;
        int me_;
    };
    LocalPipe() : a_(this, 0), b_(this, 1) {}
    Channel* alice() { return &a_; }
    Channel* bob() { return &b_; }
private:
    void meet(std::unique_lock<std::mutex>& lk) {                        // reusable two-party barrier
        const uint64_t gen = gen_;
        if (++arrived_ == 2) { arrived_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen_ != gen; });
    }
    std::mutex m_;
    std::condition_variable cv_;
    const void* ptr_[2] = {nullptr, nullptr};
    size_t bytes_[2] = {0, 0};
    int arrived_ = 0;
    uint64_t gen_ = 0;
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
    cognn_ctx* ctx() const { return ctx_; }
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

// ShareVecVec on the device: a flat row-major [rows x cols] uint64 tensor (the a1 row of SURVEY.md §8).  Copies share the
// buffer (a tensor is never modified once another DevMat may see it: functions that "update in place" rebind to a new buffer).
class DevMat {
public:
    DevMat() {}
    DevMat(cognn_ctx* c, size_t r, size_t cl) : buf_(std::make_shared<Dev>(c, r * cl)), rows_(r), cols_(cl) {}
    static DevMat from_host(cognn_ctx* c, const ShareVecVec& v) {
        size_t r, cl;
        std::vector<uint64_t> f = flatten(v, &r, &cl);
        DevMat m(c, r, cl);
        m.buf_->up(f.data());
        return m;
    }
    void to_host(ShareVecVec& out) const {
        std::vector<uint64_t> f(elems());
        if (buf_) buf_->down(f.data());
        unflatten(f, rows_, cols_, out);
    }
    DevMat clone(cognn_ctx* c = nullptr) const {
        if (!buf_) return DevMat();
        DevMat m(c ? c : buf_->ctx(), rows_, cols_);
        check(cognn_memcpy_d2d(m.ctx(), m.u64(), u64(), elems() * 8), "cognn_memcpy_d2d");
        check(cognn_ctx_sync(m.ctx()), "cognn_ctx_sync");
        return m;
    }
    size_t size() const { return rows_; }                    // number of rows, like ShareVecVec::size()
    bool empty() const { return rows_ == 0; }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    size_t elems() const { return rows_ * cols_; }
    uint64_t* u64() const { return buf_ ? buf_->u64() : nullptr; }
    cognn_ctx* ctx() const { return buf_ ? buf_->ctx() : nullptr; }
    const Dev& dev() const { return *buf_; }
    Dev& dev() { return *buf_; }
    bool shared() const { return buf_ && buf_.use_count() > 1; }
    void clear() { buf_.reset(); rows_ = cols_ = 0; }
    void swap(DevMat& o) { buf_.swap(o.buf_); std::swap(rows_, o.rows_); std::swap(cols_, o.cols_); }
private:
    std::shared_ptr<Dev> buf_;
    size_t rows_ = 0, cols_ = 0;
};

// index structure of a row gather (oblivious mapper, aggregation runs, masked add) as device CSR arrays, built once
struct Plan {
    std::unique_ptr<Dev> rowptr, col;
    size_t rows = 0;
};

inline uint64_t fnv(const void* p, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

// One role (ALICE = owner share p 0, BOB = co-party share p 1) of one pair.
struct Session {
    cognn_ctx* ctx = nullptr;
    int p = 0;
    uint64_t seed = 0, owner = 0, counter = 0;
    uint64_t server = 0;                                     // the BOB side of the pair (owner = the ALICE side)
    Channel* ch = nullptr;
    std::map<uint64_t, std::unique_ptr<Plan>> plans;         // by hash of the index structure (both roles hold the same plans)
    std::map<uint64_t, std::unique_ptr<Dev>> vectors;        // host side vectors already on the device (normalisers, labels)

    cognn_keys keys(int op) const {
        cognn_keys k;
        cognn_make_keys(seed, owner, counter, (uint64_t)op, &k);
        return k;
    }
    // the streams of a Scatter instance (client = owner, server): several sessions of one owner run such calls with equal call
    // numbers, so they are addressed by the pair, not by the owner (the tag of oracle/original_gcn.py pair_tag)
    cognn_keys pair_keys(int op) const {
        cognn_keys k;
        cognn_make_keys(seed, 0x10000ull + owner * 256 + server, counter, (uint64_t)op, &k);
        return k;
    }
    // one exchange round on device buffers
    void swap_dev(const Dev& mine, Dev& theirs) { ch->exchange_device(ctx, mine.ptr(), mine.bytes(), theirs.ptr(), theirs.bytes()); }
    void swap_mat(const DevMat& mine, DevMat& theirs) { ch->exchange_device(ctx, mine.u64(), mine.elems() * 8, theirs.u64(), theirs.elems() * 8); }
    // truncation by 2^f of `x` times the public constant `mul`; mode 1: out = out - result
    void trunc(uint64_t* out, const uint64_t* x, uint64_t mul, int op, int64_t n, int mode = 0) {
        cognn_keys tk = keys(op);
        Dev c(ctx, (size_t)n), cp(ctx, (size_t)n);
        check(cognn_trunc_open_u64(ctx, c.u64(), x, mul, &tk, p, n), "cognn_trunc_open_u64");
        swap_dev(c, cp);
        check(cognn_trunc_close_u64(ctx, out, p == 0 ? c.u64() : nullptr, p == 0 ? cp.u64() : nullptr, &tk, p, mode, n),
              "cognn_trunc_close_u64");
        check(cognn_ctx_sync(ctx), "cognn_ctx_sync");        // c / cp are released on return
    }
    // a host vector on the device, uploaded once per distinct content
    const Dev& cached(const void* host, size_t elems, size_t elem_bytes) {
        const uint64_t h = fnv(host, elems * elem_bytes, 0x9E3779B97F4A7C15ull + elems * 31 + elem_bytes);
        auto f = vectors.find(h);
        if (f == vectors.end()) {
            std::unique_ptr<Dev> d(new Dev(ctx, elems, elem_bytes));
            d->up(host);
            f = vectors.emplace(h, std::move(d)).first;
        }
        return *f->second;
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
    s->server = party == 1 ? coTid : selfTid;
    s->ch = ch;
    Registry& r = Registry::get();
    std::lock_guard<std::mutex> lk(r.m);
    r.sessions[std::make_tuple(selfTid, coTid, party)] = std::move(s);
}
inline void close_sessions() {
    Registry& r = Registry::get();
    std::lock_guard<std::mutex> lk(r.m);
    for (auto& kv : r.sessions) {
        kv.second->plans.clear(); kv.second->vectors.clear();
        cognn_ctx_destroy(kv.second->ctx);
    }
    r.sessions.clear();
}
inline Session& session(uint64_t coTid, int party) {
    Registry& r = Registry::get();
    std::lock_guard<std::mutex> lk(r.m);
    auto f = r.sessions.find(std::make_tuple(self_tid(), coTid, party));
    if (f == r.sessions.end()) throw Error("cognn_shim: no session for this (tid, coTid, party); call cognn_shim::open_session first");
    return *f->second;
}

// ---- index plans --------------------------------------------------------------------------------------------------------
// The client (ALICE) owns the index structure; the server passes placeholders (zero position vectors, ss_...h:1047,1124-1126).
// Per call the client sends the 64-bit hash of its structure; on a miss - the first use - the structure itself travels, in the
// clear, and both roles build the same device CSR.  Later calls with the same structure move 16 bytes and no index data.
enum { PLAN_MAPPER = 1, PLAN_AGGREGATE = 2, PLAN_COND = 3 };
inline void send_vec(Session& s, std::vector<uint64_t>& v) {  // client -> server
    uint64_t n = v.size();
    if (s.p == 0) { s.ch->exchange(&n, 8, nullptr, 0); s.ch->exchange(v.data(), n * 8, nullptr, 0); }
    else { s.ch->exchange(nullptr, 0, &n, 8); v.resize(n); s.ch->exchange(nullptr, 0, v.data(), n * 8); }
}
template <class Build>
inline const Plan& plan_for(Session& s, int kind, uint64_t flag, std::vector<uint64_t> a, std::vector<uint64_t> b, Build build) {
    uint64_t msg[2] = {0, flag};
    if (s.p == 0) {
        uint64_t h = fnv(a.data(), a.size() * 8, 0xcbf29ce484222325ull ^ (uint64_t)kind * 0x9E3779B97F4A7C15ull ^ flag);
        h = fnv(b.data(), b.size() * 8, h ^ (a.size() * 0x100000001b3ull));
        msg[0] = h;
        s.ch->exchange(msg, 16, nullptr, 0);
    } else {
        s.ch->exchange(nullptr, 0, msg, 16);
    }
    auto f = s.plans.find(msg[0]);
    if (f != s.plans.end()) return *f->second;
    send_vec(s, a);
    send_vec(s, b);
    std::vector<uint32_t> rowptr{0}, col;
    build(a, b, msg[1], rowptr, col);
    std::unique_ptr<Plan> pl(new Plan());
    pl->rows = rowptr.size() - 1;
    pl->rowptr.reset(new Dev(s.ctx, rowptr.size(), 4));
    pl->col.reset(new Dev(s.ctx, col.size(), 4));
    pl->rowptr->up(rowptr.data());
    pl->col->up(col.data());
    return *s.plans.emplace(msg[0], std::move(pl)).first->second;
}

// out[r] = (base ? base[r] : 0) + sum_{q in rowptr[r]..rowptr[r+1]} table[col[q]]  (cognn_gather_csr_u64)
inline DevMat gather(Session& s, const DevMat& table, const DevMat* base, const Plan& pl, size_t width) {
    if (table.rows() && table.cols() != width) throw Error("cognn_shim: row width mismatch");
    if (base && (base->rows() != pl.rows || (base->rows() && base->cols() != width))) throw Error("cognn_shim: base shape mismatch");
    DevMat out(s.ctx, pl.rows, width);
    if (pl.rows && width) {
        // (an empty table still needs a valid pointer)
        Dev none(s.ctx, 2);
        check(cognn_gather_csr_u64(s.ctx, out.u64(), base ? base->u64() : nullptr, table.u64() ? table.u64() : none.u64(), (const uint32_t*)pl.rowptr->ptr(),
                                   (const uint32_t*)pl.col->ptr(), (int64_t)pl.rows, (int64_t)width), "cognn_gather_csr_u64");
        check(cognn_ctx_sync(s.ctx), "cognn_ctx_sync");
    }
    return out;
}

// ---- the protocol functions on device tensors ------------------------------------------------------------------------------
namespace dev {

inline DevMat mapper(Session& s, const std::vector<uint64_t>& srcPos, const std::vector<uint64_t>& dstPos, const DevMat& src, bool allowMissing) {
    const Plan& pl = plan_for(s, PLAN_MAPPER, allowMissing ? 1 : 0, srcPos, dstPos,
        [](const std::vector<uint64_t>& sp, const std::vector<uint64_t>& dp, uint64_t allow, std::vector<uint32_t>& rowptr, std::vector<uint32_t>& col) {
            std::map<uint64_t, uint32_t> last;               // dst[r] = src[last q with srcPos[q] == dstPos[r]]
            for (size_t q = 0; q < sp.size(); ++q) last[sp[q]] = (uint32_t)q;
            for (uint64_t d : dp) {
                auto f = last.find(d);
                if (f != last.end()) col.push_back(f->second);
                else if (!allow) throw Error("cognn_shim: oblivious mapper: destination position missing in the source");
                rowptr.push_back((uint32_t)col.size());
            }
        });
    return gather(s, src, nullptr, pl, src.cols());
}

// inclusive prefix sum inside each run of equal consecutive positions: row q gathers rows start(q)..q
inline DevMat aggregate(Session& s, const std::vector<uint64_t>& pos, const DevMat& svv) {
    const Plan& pl = plan_for(s, PLAN_AGGREGATE, 0, pos, std::vector<uint64_t>(),
        [](const std::vector<uint64_t>& ps, const std::vector<uint64_t>&, uint64_t, std::vector<uint32_t>& rowptr, std::vector<uint32_t>& col) {
            size_t start = 0;
            for (size_t q = 0; q < ps.size(); ++q) {
                if (q && ps[q] != ps[q - 1]) start = q;
                for (size_t j = start; j <= q; ++j) col.push_back((uint32_t)j);
                rowptr.push_back((uint32_t)col.size());
            }
        });
    if (pl.rows != svv.rows()) throw Error("prefix_network_aggregate: one position per row expected");
    return gather(s, svv, nullptr, pl, svv.cols());
}

// out = a + (cond ? b : 0) row by row; the condition is the client's (ALICE's) vector
inline DevMat cond_add(Session& s, const DevMat& a, const DevMat& b, const std::vector<bool>& cond) {
    if (a.rows() != b.rows() || cond.size() != a.rows()) throw Error("twoPartyGCNCondVectorAddition: shape mismatch");
    std::vector<uint64_t> c(cond.size());
    for (size_t r = 0; r < cond.size(); ++r) c[r] = cond[r] ? 1 : 0;
    const Plan& pl = plan_for(s, PLAN_COND, 0, c, std::vector<uint64_t>(),
        [](const std::vector<uint64_t>& cs, const std::vector<uint64_t>&, uint64_t, std::vector<uint32_t>& rowptr, std::vector<uint32_t>& col) {
            for (size_t r = 0; r < cs.size(); ++r) {
                if (cs[r]) col.push_back((uint32_t)r);
                rowptr.push_back((uint32_t)col.size());
            }
        });
    return gather(s, b, &a, pl, a.cols());
}

inline DevMat matmul(Session& s, const DevMat& A, const DevMat& B) {
    const size_t M = A.rows(), K = A.cols(), N = B.cols();
    if (K != B.rows()) throw Error("twoPartyGCNMatMul: inner dimensions differ");
    const int p = s.p;
    cognn_keys k = s.keys(OP_GEMM);
    DevMat E(s.ctx, M, K), Ep(s.ctx, M, K), F(s.ctx, K, N), Fp(s.ctx, K, N), Z(s.ctx, M, N), O(s.ctx, M, N);
    Dev scratch(s.ctx, M * K + K * N);
    check(cognn_mask_open_u64(s.ctx, E.u64(), A.u64(), k.k[p ? SL_A1 : SL_A0], (int64_t)M, (int64_t)K, COGNN_MASK_OPEN_LIMB), "cognn_mask_open_u64");
    check(cognn_mask_open_u64(s.ctx, F.u64(), B.u64(), k.k[p ? SL_B1 : SL_B0], (int64_t)K, (int64_t)N, 0), "cognn_mask_open_u64");
    s.swap_mat(E, Ep);                                       // Beaver reveal
    s.swap_mat(F, Fp);
    std::unique_ptr<Dev> c1;
    if (p == 1) {                                            // the dealer's product share (offline phase; dealt in place here)
        c1.reset(new Dev(s.ctx, M * N));
        check(cognn_dealer_gemm_c1_u64(s.ctx, c1->u64(), &k, (int64_t)M, (int64_t)N, (int64_t)K, 0, scratch.u64(), scratch.u64() + M * K),
              "cognn_dealer_gemm_c1_u64");
    }
    check(cognn_beaver_gemm_close2_u64(s.ctx, Z.u64(), E.u64(), Ep.u64(), F.u64(), Fp.u64(), c1 ? c1->u64() : nullptr, &k, p, (int64_t)M, (int64_t)N,
                                       (int64_t)K, 0, scratch.u64(), 0), "cognn_beaver_gemm_close2_u64");
    s.trunc(O.u64(), Z.u64(), 1, OP_GEMM_TRUNC, (int64_t)(M * N));
    ++s.counter;
    return O;
}

// (k_over / tk_over: streams other than the session's own for this call - the two scales of a Scatter instance; advance: the call
// number moves on)
inline DevMat vector_scale(Session& s, const DevMat& in, const std::vector<uint64_t>& rowScale, const cognn_keys* k_over = nullptr,
                           const cognn_keys* tk_over = nullptr, bool advance = true) {
    const size_t n = in.rows(), F = in.cols();
    if (rowScale.size() != n) throw Error("twoPartyGCNVectorScale: one scale per row expected");
    cognn_keys k = k_over ? *k_over : s.keys(OP_SCALE), tk = tk_over ? *tk_over : s.keys(OP_SCALE_TRUNC);
    const Dev& S = s.cached(rowScale.data(), n, 8);
    DevMat E(s.ctx, n, F), Ep(s.ctx, n, F), c(s.ctx, n, F), cp(s.ctx, n, F), O(s.ctx, n, F);
    Dev G(s.ctx, n), Gp(s.ctx, n);
    check(cognn_rowscale_open_u64(s.ctx, E.u64(), G.u64(), in.u64(), S.u64(), &k, s.p, (int64_t)n, (int64_t)F), "cognn_rowscale_open_u64");
    s.swap_mat(E, Ep);
    s.swap_dev(G, Gp);
    check(cognn_rowscale_close_u64(s.ctx, c.u64(), E.u64(), Ep.u64(), G.u64(), Gp.u64(), &k, &tk, s.p, (int64_t)n, (int64_t)F), "cognn_rowscale_close_u64");
    s.swap_mat(c, cp);
    check(cognn_trunc_close_u64(s.ctx, O.u64(), s.p == 0 ? c.u64() : nullptr, s.p == 0 ? cp.u64() : nullptr, &tk, s.p, 0, (int64_t)(n * F)),
          "cognn_trunc_close_u64");
    check(cognn_ctx_sync(s.ctx), "cognn_ctx_sync");
    if (advance) ++s.counter;
    return O;
}
// ScatterComp of the unoptimised kernel (original-gcn/gcn.h:243-250): trunc(trunc(in * n0) * n1), one normaliser pair per row
// (= per edge of the Scatter instance); ONE protocol call with four stream families (ops 30..33) addressed by the pair
enum { OP_SC_SCALE0 = 30, OP_SC_SCALE0_TRUNC = 31, OP_SC_SCALE1 = 32, OP_SC_SCALE1_TRUNC = 33 };
inline DevMat vector_scale2(Session& s, const DevMat& in, const std::vector<uint64_t>& n0, const std::vector<uint64_t>& n1) {
    const cognn_keys k0 = s.pair_keys(OP_SC_SCALE0), t0 = s.pair_keys(OP_SC_SCALE0_TRUNC), k1 = s.pair_keys(OP_SC_SCALE1), t1 = s.pair_keys(OP_SC_SCALE1_TRUNC);
    DevMat mid = vector_scale(s, in, n0, &k0, &t0, false);
    DevMat out = vector_scale(s, mid, n1, &k1, &t1, false);
    ++s.counter;
    return out;
}

// masked-sign ReLU: h (optional) and the public sign mask (optional, 1 byte per element)
inline void relu_core(Session& s, const DevMat& z, DevMat* h, std::unique_ptr<Dev>* mask) {
    const size_t n = z.elems();
    cognn_keys k = s.keys(OP_RELU);
    Dev E(s.ctx, n), Ep(s.ctx, n), w(s.ctx, n), wp(s.ctx, n);
    DevMat H(s.ctx, z.rows(), z.cols());
    std::unique_ptr<Dev> Mk(new Dev(s.ctx, n, 1));
    check(cognn_relu_open_u64(s.ctx, E.u64(), nullptr, z.u64(), &k, s.p, (int64_t)n), "cognn_relu_open_u64");
    s.swap_dev(E, Ep);
    check(cognn_relu_mul_u64(s.ctx, w.u64(), E.u64(), Ep.u64(), nullptr, nullptr, &k, s.p, (int64_t)n), "cognn_relu_mul_u64");
    s.swap_dev(w, wp);
    check(cognn_relu_close_u64(s.ctx, H.u64(), (uint8_t*)Mk->ptr(), z.u64(), w.u64(), wp.u64(), (int64_t)n), "cognn_relu_close_u64");
    check(cognn_ctx_sync(s.ctx), "cognn_ctx_sync");
    if (h) *h = H;
    if (mask) *mask = std::move(Mk);
    ++s.counter;
}

inline void backward_nn_without_ah(Session& s, const DevMat& in, const DevMat& z, const DevMat& weightT, DevMat& dstVec, DevMat& g, bool isFirstLayer) {
    if (in.rows() != z.rows() || in.cols() != z.cols()) throw Error("twoPartyGCNBackwardNNWithoutAH: shape mismatch");
    std::unique_ptr<Dev> mask;
    relu_core(s, z, nullptr, &mask);                         // 1[z > 0], public
    DevMat masked(s.ctx, in.rows(), in.cols());
    check(cognn_mask_select_u64(s.ctx, masked.u64(), in.u64(), (const uint8_t*)mask->ptr(), (int64_t)in.elems()), "cognn_mask_select_u64");
    check(cognn_ctx_sync(s.ctx), "cognn_ctx_sync");
    if (!isFirstLayer) g = matmul(s, masked, weightT);       // gradient for the layer below; skipped for the first layer
    else g.clear();
    dstVec = masked;
}

// labels: the owner's class index per row (the one-hot rows of the reference's `label` argument); the co-party passes none
inline void prediction(Session& s, const DevMat& z, const std::vector<int32_t>& labels, DevMat& p, DevMat& p_minus_y) {
    const size_t n = z.rows(), L = z.cols();
    cognn_keys k = s.keys(OP_SOFTMAX);
    DevMat P(s.ctx, n, L), D(s.ctx, n, L);
    if (s.p == 0) {                                          // the owner receives the co-party's share of z (it learns p anyway, gcn.h:603-604)
        DevMat Zp(s.ctx, n, L), PF(s.ctx, n, L);
        s.ch->exchange_device(s.ctx, nullptr, 0, Zp.u64(), n * L * 8);
        if (labels.size() != n) throw Error("twoPartyGCNForwardNNPredictionWithoutWeight: one label per row expected");
        const Dev& Lb = s.cached(labels.data(), n, 4);
        check(cognn_softmax_u64(s.ctx, P.u64(), D.u64(), PF.u64(), z.u64(), Zp.u64(), (const int32_t*)Lb.ptr(), &k, 0, (int64_t)n, (int64_t)L, (int64_t)n),
              "cognn_softmax_u64");
    } else {
        s.ch->exchange_device(s.ctx, z.u64(), n * L * 8, nullptr, 0);
        check(cognn_softmax_u64(s.ctx, P.u64(), D.u64(), nullptr, nullptr, nullptr, nullptr, &k, 1, (int64_t)n, (int64_t)L, (int64_t)n), "cognn_softmax_u64");
    }
    check(cognn_ctx_sync(s.ctx), "cognn_ctx_sync");
    p = P; p_minus_y = D;
    ++s.counter;
}

inline DevMat matrix_scale(Session& s, const DevMat& m, uint64_t fxScale) {
    DevMat O(s.ctx, m.rows(), m.cols());
    s.trunc(O.u64(), m.u64(), fxScale, OP_MSCALE_TRUNC, (int64_t)m.elems());
    ++s.counter;
    return O;
}

inline DevMat apply_gradient(Session& s, const DevMat& W, const DevMat& d, uint64_t fxLr) {
    if (W.rows() != d.rows() || W.cols() != d.cols()) throw Error("twoPartyGCNApplyGradient: shape mismatch");
    DevMat O = W.clone(s.ctx);
    s.trunc(O.u64(), d.u64(), fxLr, OP_LR_TRUNC, (int64_t)W.elems(), 1);       // W -= trunc(lr * d)
    ++s.counter;
    return O;
}

}  // namespace dev
}  // namespace cognn_shim

// ---- free functions of include/task/task.h:243-249 (defined externally in the reference) --------------------------------
inline ShareTensor transpose(const ShareTensor& st) {
    const size_t r = st.size(), c = r ? st[0].size() : 0;
    ShareTensor t(c, ShareVec(r));
    for (size_t i = 0; i < r; ++i)
        for (size_t j = 0; j < c; ++j) t[j][i] = st[i][j];
    return t;
}
inline cognn_shim::DevMat transpose(const cognn_shim::DevMat& st) {
    cognn_shim::DevMat t(st.ctx(), st.cols(), st.rows());
    if (st.elems()) {
        cognn_shim::check(cognn_transpose_u64(st.ctx(), t.u64(), st.u64(), (int64_t)st.rows(), (int64_t)st.cols()), "cognn_transpose_u64");
        cognn_shim::check(cognn_ctx_sync(st.ctx()), "cognn_ctx_sync");
    }
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
// (the server passes a zero position vector, ss_...h:1047: it learns the runs from the client)
enum class AggregationOp { ADD_AGG };
inline cognn_shim::DevMat prefix_network_aggregate(const std::vector<uint64_t>& pos, const cognn_shim::DevMat& svv, AggregationOp, uint64_t coTid, int party, bool) {
    cognn_shim::Session& s = cognn_shim::session(coTid, party);
    return cognn_shim::dev::aggregate(s, s.p == 0 ? pos : std::vector<uint64_t>(), svv);
}
inline ShareVecVec prefix_network_aggregate(const std::vector<uint64_t>& pos, const ShareVecVec& svv, AggregationOp op, uint64_t coTid, int party, bool b) {
    cognn_shim::Session& s = cognn_shim::session(coTid, party);
    ShareVecVec out;
    prefix_network_aggregate(pos, cognn_shim::DevMat::from_host(s.ctx, svv), op, coTid, party, b).to_host(out);
    return out;
}

// ---- ObliviousMapper.h (ss_...h:752-763,818,848 client; :1011-1016,1057,1075 server) ---------------------------------------
inline void client_oblivious_mapper_online(const std::vector<uint64_t>& srcPos, const std::vector<uint64_t>& dstPos, const cognn_shim::DevMat& srcSvv,
                                           cognn_shim::DevMat& dstSvv, uint32_t /*plainNumPerOperand*/, uint64_t /*iter*/, uint32_t /*preprocessId*/,
                                           uint64_t coTid, bool allowMissing = false) {
    dstSvv = cognn_shim::dev::mapper(cognn_shim::session(coTid, 1), srcPos, dstPos, srcSvv, allowMissing);
}
inline void server_oblivious_mapper_online(const cognn_shim::DevMat& srcSvv, cognn_shim::DevMat& dstSvv, uint64_t /*iter*/, uint32_t /*preprocessId*/, uint64_t coTid) {
    dstSvv = cognn_shim::dev::mapper(cognn_shim::session(coTid, 2), {}, {}, srcSvv, false);
}
inline void client_oblivious_mapper_online(const std::vector<uint64_t>& srcPos, const std::vector<uint64_t>& dstPos, const ShareVecVec& srcSvv,
                                           ShareVecVec& dstSvv, uint32_t width, uint64_t iter, uint32_t preprocessId, uint64_t coTid, bool allowMissing = false) {
    cognn_shim::DevMat d;
    client_oblivious_mapper_online(srcPos, dstPos, cognn_shim::DevMat::from_host(cognn_shim::session(coTid, 1).ctx, srcSvv), d, width, iter, preprocessId, coTid, allowMissing);
    d.to_host(dstSvv);
}
inline void server_oblivious_mapper_online(const ShareVecVec& srcSvv, ShareVecVec& dstSvv, uint64_t iter, uint32_t preprocessId, uint64_t coTid) {
    cognn_shim::DevMat d;
    server_oblivious_mapper_online(cognn_shim::DevMat::from_host(cognn_shim::session(coTid, 2).ctx, srcSvv), d, iter, preprocessId, coTid);
    d.to_host(dstSvv);
}

// ---- SCIHarness.h -----------------------------------------------------------------------------------------------------------
namespace sci {
enum { ALICE = 1, BOB = 2 };
using cognn_shim::Dev;
using cognn_shim::DevMat;
using cognn_shim::Session;
using cognn_shim::check;

// -- device tensors ------------------------------------------------------------------------------------------------------
inline void twoPartyGCNMatMul(const DevMat& A, const DevMat& B, DevMat& out, uint64_t coTid, int party) {
    out = cognn_shim::dev::matmul(cognn_shim::session(coTid, party), A, B);
}
inline void twoPartyGCNVectorScale(const DevMat& in, const std::vector<uint64_t>& rowScale, DevMat& out, bool /*isSigned*/, uint64_t coTid, int party) {
    out = cognn_shim::dev::vector_scale(cognn_shim::session(coTid, party), in, rowScale);
}
// The condition is the client's private input - the server passes a placeholder (zeroIsDummy, ss_...h:1124-1126) - so
// ALICE's vector is the one both roles apply; here it reaches BOB in the clear.
inline void twoPartyGCNCondVectorAddition(DevMat& a, DevMat& b, std::vector<bool>& cond, DevMat& out, uint64_t coTid, int party) {
    out = cognn_shim::dev::cond_add(cognn_shim::session(coTid, party), a, b, cond);
}
inline void twoPartyGCNRelu(const DevMat& in, DevMat& out, uint64_t tid, int party) {
    cognn_shim::dev::relu_core(cognn_shim::session(tid, party), in, &out, nullptr);
}
inline void twoPartyGCNBackwardNNWithoutAH(const DevMat& in, const DevMat& z, const DevMat& weightT, DevMat& dstVec, DevMat& g, bool isFirstLayer,
                                           uint64_t tid, int party) {
    cognn_shim::dev::backward_nn_without_ah(cognn_shim::session(tid, party), in, z, weightT, dstVec, g, isFirstLayer);
}
// `label`: the client's one-hot rows (toShareVec), host side - they are its plaintext; the server passes zero rows
inline void twoPartyGCNForwardNNPredictionWithoutWeight(const DevMat& z, const ShareVecVec& label, DevMat& p, DevMat& p_minus_y, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    std::vector<int32_t> lab;
    if (s.p == 0) {
        lab.assign(z.rows(), 0);
        for (size_t r = 0; r < z.rows() && r < label.size(); ++r)
            for (size_t j = 0; j < label[r].size(); ++j)
                if (label[r][j] != 0) lab[r] = (int32_t)j;
    }
    cognn_shim::dev::prediction(s, z, lab, p, p_minus_y);
}
inline void getPlainShareVecVec(const DevMat& sv, DoubleTensor& plain, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    DevMat other(s.ctx, sv.rows(), sv.cols());
    s.swap_mat(sv, other);
    check(cognn_add_u64(s.ctx, other.u64(), other.u64(), sv.u64(), (int64_t)sv.elems()), "cognn_add_u64");
    std::vector<uint64_t> f(sv.elems());
    other.dev().down(f.data());                              // the revealed values are the only tensor that leaves the device
    plain.assign(sv.rows(), std::vector<double>(sv.cols()));
    for (size_t r = 0; r < sv.rows(); ++r)
        for (size_t j = 0; j < sv.cols(); ++j) plain[r][j] = (double)(int64_t)f[r * sv.cols() + j] / (double)(1ull << SCALER_BIT_LENGTH);
}
inline void twoPartyGCNMatrixScale(const DevMat& m, uint64_t fxScale, DevMat& out, uint64_t tid, int party) {
    out = cognn_shim::dev::matrix_scale(cognn_shim::session(tid, party), m, fxScale);
}
inline void twoPartyGCNApplyGradient(const DevMat& W, const DevMat& d, uint64_t fxLr, DevMat& Wout, uint64_t tid, int party) {
    Wout = cognn_shim::dev::apply_gradient(cognn_shim::session(tid, party), W, d, fxLr);
}
inline DevMat plaintext_add_matrix(const DevMat& a, const DevMat& b) {
    if (a.rows() != b.rows() || a.cols() != b.cols()) throw cognn_shim::Error("plaintext_add_matrix: shape mismatch");
    DevMat o(a.ctx(), a.rows(), a.cols());
    if (a.elems()) {
        check(cognn_add_u64(a.ctx(), o.u64(), a.u64(), b.u64(), (int64_t)a.elems()), "cognn_add_u64");
        check(cognn_ctx_sync(a.ctx()), "cognn_ctx_sync");
    }
    return o;
}
inline void plaintext_add_matrix_in_place(DevMat& a, const DevMat& b) { a = plaintext_add_matrix(a, b); }

// -- nested host vectors (the reference's containers): upload, the same device functions, download -------------------------
inline void twoPartyGCNMatMul(const ShareVecVec& A, const ShareTensor& B, ShareVecVec& out, uint64_t coTid, int party) {
    Session& s = cognn_shim::session(coTid, party);
    DevMat o;
    twoPartyGCNMatMul(DevMat::from_host(s.ctx, A), DevMat::from_host(s.ctx, B), o, coTid, party);
    o.to_host(out);
}
inline void twoPartyGCNVectorScale(const ShareVecVec& in, const std::vector<uint64_t>& rowScale, ShareVecVec& out, bool isSigned, uint64_t coTid, int party) {
    Session& s = cognn_shim::session(coTid, party);
    DevMat o;
    twoPartyGCNVectorScale(DevMat::from_host(s.ctx, in), rowScale, o, isSigned, coTid, party);
    o.to_host(out);
}
inline void twoPartyGCNCondVectorAddition(ShareVecVec& a, ShareVecVec& b, std::vector<bool>& cond, ShareVecVec& out, uint64_t coTid, int party) {
    Session& s = cognn_shim::session(coTid, party);
    DevMat da = DevMat::from_host(s.ctx, a), db = DevMat::from_host(s.ctx, b), o;
    twoPartyGCNCondVectorAddition(da, db, cond, o, coTid, party);
    o.to_host(out);
}
inline void twoPartyGCNRelu(const ShareVecVec& in, ShareTensor& out, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    DevMat o;
    twoPartyGCNRelu(DevMat::from_host(s.ctx, in), o, tid, party);
    o.to_host(out);
}
inline void twoPartyGCNBackwardNNWithoutAH(const ShareVecVec& in, const ShareTensor& z, const ShareTensor& weightT, ShareVecVec& dstVec, ShareTensor& g,
                                           bool isFirstLayer, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    DevMat d, gg;
    twoPartyGCNBackwardNNWithoutAH(DevMat::from_host(s.ctx, in), DevMat::from_host(s.ctx, z), DevMat::from_host(s.ctx, weightT), d, gg, isFirstLayer, tid, party);
    d.to_host(dstVec);
    if (isFirstLayer) g.clear(); else gg.to_host(g);
}
inline void twoPartyGCNForwardNNPredictionWithoutWeight(const ShareVecVec& z, const ShareVecVec& label, ShareTensor& p, ShareTensor& p_minus_y,
                                                        uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    DevMat dp, dd;
    twoPartyGCNForwardNNPredictionWithoutWeight(DevMat::from_host(s.ctx, z), label, dp, dd, tid, party);
    dp.to_host(p); dd.to_host(p_minus_y);
}
inline void getPlainShareVecVec(const ShareVecVec& sv, DoubleTensor& plain, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    getPlainShareVecVec(DevMat::from_host(s.ctx, sv), plain, tid, party);
}
inline void twoPartyGCNMatrixScale(const ShareVecVec& m, uint64_t fxScale, ShareVecVec& out, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    DevMat o;
    twoPartyGCNMatrixScale(DevMat::from_host(s.ctx, m), fxScale, o, tid, party);
    o.to_host(out);
}
inline void twoPartyGCNApplyGradient(const ShareVecVec& W, const ShareVecVec& d, uint64_t fxLr, ShareVecVec& Wout, uint64_t tid, int party) {
    Session& s = cognn_shim::session(tid, party);
    DevMat o;
    twoPartyGCNApplyGradient(DevMat::from_host(s.ctx, W), DevMat::from_host(s.ctx, d), fxLr, o, tid, party);
    o.to_host(Wout);
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

// ---- the fused ops of the unoptimised kernel (algo_kernels/vertex_centric/original-gcn/gcn.h), for either container ------------
// Their definitions live in the absent SCI library; the input / output relation is INFERRED from the call sites and stated here
// as compositions of the ops above (oracle/original_gcn.py restates the same compositions for the engine's original-gcn variant).
// The `normalizer` arguments of the Apply ops are accepted and unused: in the reference's call sites the degree normalisation has
// already happened in ScatterComp / GatherComp.
// two-normaliser scale of ScatterComp (:243-250): out = trunc(trunc(in * n0) * n1), one normaliser pair per row (= per edge); its
// dealer streams belong to the Scatter instance (cognn_shim::dev::vector_scale2)
inline void twoPartyGCNVectorScale(const DevMat& in, const std::vector<uint64_t>& normalizer0, const std::vector<uint64_t>& normalizer1, DevMat& out,
                                   uint64_t coTid, int party) {
    out = cognn_shim::dev::vector_scale2(cognn_shim::session(coTid, party), in, normalizer0, normalizer1);
}
inline void twoPartyGCNVectorScale(const ShareVecVec& in, const std::vector<uint64_t>& normalizer0, const std::vector<uint64_t>& normalizer1, ShareVecVec& out,
                                   uint64_t coTid, int party) {
    Session& s = cognn_shim::session(coTid, party);
    DevMat o;
    twoPartyGCNVectorScale(DevMat::from_host(s.ctx, in), normalizer0, normalizer1, o, coTid, party);
    o.to_host(out);
}
// GCN_FORWARD_NN (:459): z = in . W, new_h = ReLU(z)
template <class Mat, class Ten>
inline void twoPartyGCNForwardNN(const Mat& in, const Ten& weight, const std::vector<uint64_t>& /*normalizer*/, Ten& z, Ten& new_h, uint64_t tid, int party) {
    twoPartyGCNMatMul(in, weight, z, tid, party);
    twoPartyGCNRelu(z, new_h, tid, party);
}
// prediction layer (:493,508): z = in . W, then softmax, p - y
template <class Mat, class Ten>
inline void twoPartyGCNForwardNNPrediction(const Mat& in, const Ten& weight, const ShareVecVec& label, const std::vector<uint64_t>& /*normalizer*/, Ten& z, Ten& p,
                                           Ten& p_minus_y, uint64_t tid, int party) {
    twoPartyGCNMatMul(in, weight, z, tid, party);
    twoPartyGCNForwardNNPredictionWithoutWeight(z, label, p, p_minus_y, tid, party);
}
// last layer's backward (:586): d = ah_t . in (ah_t: the transposed aggregate saved by the forward pass), g = in . W^T
template <class Mat, class Ten>
inline void twoPartyGCNBackwardNNInit(const Mat& in, const Ten& ah_t, const Ten& weightT, const std::vector<uint64_t>& /*normalizer*/, Ten& d, Ten& g, uint64_t tid,
                                      int party) {
    twoPartyGCNMatMul(in, weightT, g, tid, party);
    twoPartyGCNMatMul(ah_t, in, d, tid, party);
}
// hidden layers' backward (:622): gz = in (.) 1[z > 0], d = ah_t . gz, g = gz . W^T unless this is the first layer
template <class Mat, class Ten>
inline void twoPartyGCNBackwardNN(const Mat& in, const Ten& ah_t, const Ten& z, const Ten& weightT, const std::vector<uint64_t>& /*normalizer*/, Ten& d, Ten& g,
                                  bool isFirstLayer, uint64_t tid, int party) {
    Mat gz;
    twoPartyGCNBackwardNNWithoutAH(in, z, weightT, gz, g, isFirstLayer, tid, party);
    twoPartyGCNMatMul(ah_t, gz, d, tid, party);
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

// ---- container helpers the callbacks use on either type (include/cognn_gas_kernel.hpp) --------------------------------------
namespace cognn_shim {
inline size_t svv_cols(const ShareVecVec& v) { return v.empty() ? 0 : v[0].size(); }
inline size_t svv_cols(const DevMat& v) { return v.cols(); }
inline ShareVecVec svv_clone(const ShareVecVec& v) { return v; }
inline DevMat svv_clone(const DevMat& v) { return v.clone(); }
// rows [first, rows) <- 0   (gcn.h:639-641)
inline void svv_zero_rows_from(ShareVecVec& v, size_t first) {
    for (size_t r = first; r < v.size(); ++r) std::fill(v[r].begin(), v[r].end(), 0);
}
inline void svv_zero_rows_from(DevMat& v, size_t first) {
    if (first >= v.rows() || !v.cols()) return;
    if (v.shared()) v = v.clone();
    check(cognn_memset0(v.ctx(), v.u64() + first * v.cols(), (v.rows() - first) * v.cols() * 8), "cognn_memset0");
    check(cognn_ctx_sync(v.ctx()), "cognn_ctx_sync");
}
// host <-> container of the calling role's session
inline void svv_from_host(Session&, const ShareVecVec& h, ShareVecVec& out) { out = h; }
inline void svv_from_host(Session& s, const ShareVecVec& h, DevMat& out) { out = DevMat::from_host(s.ctx, h); }
inline void svv_to_host(const ShareVecVec& v, ShareVecVec& out) { out = v; }
inline void svv_to_host(const DevMat& v, ShareVecVec& out) { v.to_host(out); }
}  // namespace cognn_shim
#endif  // COGNN_SCI_SHIM_HPP_
