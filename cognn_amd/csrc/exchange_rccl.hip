// Native share exchange over RCCL (include/cognn_exchange.h): one p2p group per round on a communication stream, ordered with
// the engine's compute stream by HIP events only.  Stands in for the reference's TCP mesh and message helpers
// (include/engine.h:157-201, include/comm_sync.h:245-277).
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <utility>
#include <vector>

#include "common.h"
#include "../../include/cognn_exchange.h"

static thread_local char g_xerr[1024] = "";
static int xerr(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_xerr, sizeof(g_xerr), fmt, ap);
    va_end(ap);
    return 1;
}
#define X_HIP(expr)                                                                                            \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) return xerr("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)
#define X_NCCL(expr)                                                                                           \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) return xerr("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
    } while (0)
#define X_REQUIRE(cond, ...)                     \
    do {                                         \
        if (!(cond)) return xerr(__VA_ARGS__);   \
    } while (0)

struct cognn_rccl_exchange {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t compute = nullptr;      // the engine's stream (not owned)
    hipStream_t comm_stream = nullptr;  // owned
    hipEvent_t ready = nullptr;         // compute -> comm: the round's send buffers are written
    hipEvent_t done = nullptr;          // comm -> compute: the messages of the newest round (and, in stream order, all before) have arrived
    // ... and one event per round, in a ring: a consumer of round r waits for r only while r + 1.. are still in flight
    // (cognn_rccl_exchange_wait_round; a slot that has been reused since belongs to a LATER round, which over-waits, never under-waits)
    static constexpr int kRing = 64;
    hipEvent_t done_ring[kRing] = {};
    int64_t round_base = 0;             // rounds begun before the engine's numbering started (cognn_engine_set_exchange_rccl)
    // every round is bracketed by a pair of timing events on the communication stream: its duration there is the time the
    // p2p group took (cognn_rccl_exchange_time)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing, spare;
    double comm_ms = 0;
    int64_t rounds = 0, sent = 0, received = 0;
    uint64_t* scratch = nullptr;        // barrier payload
};

extern "C" {

const char* cognn_exchange_last_error(void) { return g_xerr; }

int cognn_rccl_unique_id(void* id128) {
    X_REQUIRE(id128, "cognn_rccl_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == COGNN_RCCL_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    X_NCCL(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

static int write_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
        ssize_t w = send(fd, c, n, MSG_NOSIGNAL);
        if (w <= 0) { if (errno == EINTR) continue; return 1; }
        c += w; n -= (size_t)w;
    }
    return 0;
}
static int read_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n) {
        ssize_t r = recv(fd, c, n, 0);
        if (r <= 0) { if (r < 0 && errno == EINTR) continue; return 1; }
        c += r; n -= (size_t)r;
    }
    return 0;
}

int cognn_rccl_rendezvous_tcp(const char* addr, int port, int rank, int world, double timeout_s, void* id128) {
    X_REQUIRE(addr && id128 && world >= 1 && rank >= 0 && rank < world, "cognn_rccl_rendezvous_tcp: bad arguments");
    if (port <= 0) port = 1712;                                     // the reference's base port (engine.h:171)
    sockaddr_in sa;
    memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    X_REQUIRE(inet_pton(AF_INET, addr, &sa.sin_addr) == 1, "cognn_rccl_rendezvous_tcp: '%s' is not an IPv4 address", addr);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
    if (rank == 0) {
        if (cognn_rccl_unique_id(id128)) return 1;
        if (world == 1) return 0;
        int ls = socket(AF_INET, SOCK_STREAM, 0);
        X_REQUIRE(ls >= 0, "cognn_rccl_rendezvous_tcp: socket: %s", strerror(errno));
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        if (bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0 || listen(ls, world) != 0) {
            const int e = errno; close(ls);
            return xerr("cognn_rccl_rendezvous_tcp: cannot listen on %s:%d: %s", addr, port, strerror(e));
        }
        timeval tv; tv.tv_sec = 1; tv.tv_usec = 0;
        setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));    // accept() wakes up once a second to check the deadline
        // A stray or garbled connection (port scanner, a client that dies mid-handshake) is dropped and accepting goes on until
        // the deadline; a rank is counted once however often it connects (a retry gets the id again).
        std::vector<char> seen((size_t)world, 0);
        auto remaining = [&] { return std::chrono::duration<double>(deadline - std::chrono::steady_clock::now()).count(); };
        for (int served = 0; served < world - 1;) {
            if (remaining() <= 0) { close(ls); return xerr("cognn_rccl_rendezvous_tcp: %d of %d ranks connected before the timeout", served, world - 1); }
            int c = accept(ls, nullptr, nullptr);
            if (c < 0) continue;                                    // (woken once a second: the deadline is checked at the top)
            // the accepted socket inherits the listener's 1 s: too short for a slow client; but never longer than what is left
            const double left = std::min(10.0, std::max(0.05, remaining()));
            timeval ctv; ctv.tv_sec = (long)left; ctv.tv_usec = (long)((left - (double)(long)left) * 1e6);
            setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &ctv, sizeof(ctv));
            setsockopt(c, SOL_SOCKET, SO_SNDTIMEO, &ctv, sizeof(ctv));
            int32_t peer = -1;
            const int bad = read_all(c, &peer, sizeof(peer)) || peer <= 0 || peer >= world || write_all(c, id128, COGNN_RCCL_ID_BYTES);
            close(c);
            if (bad) continue;                                      // the client connects again (below) until the deadline
            if (!seen[(size_t)peer]) { seen[(size_t)peer] = 1; ++served; }
        }
        close(ls);
        return 0;
    }
    // a client whose connection is refused (rank 0 not listening yet) or whose hand-shake is dropped half-way connects again until
    // the deadline: the server counts a rank once however often it is served
    for (;;) {
        int s = socket(AF_INET, SOCK_STREAM, 0);
        X_REQUIRE(s >= 0, "cognn_rccl_rendezvous_tcp: socket: %s", strerror(errno));
        const char* what = "not reachable";
        if (connect(s, (sockaddr*)&sa, sizeof(sa)) == 0) {
            timeval ctv; ctv.tv_sec = 10; ctv.tv_usec = 0;
            setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &ctv, sizeof(ctv));
            setsockopt(s, SOL_SOCKET, SO_SNDTIMEO, &ctv, sizeof(ctv));
            int32_t me = rank;
            const int bad = write_all(s, &me, sizeof(me)) || read_all(s, id128, COGNN_RCCL_ID_BYTES);
            close(s);
            if (!bad) return 0;
            what = "dropped the hand-shake";
        } else {
            close(s);
        }
        if (std::chrono::steady_clock::now() > deadline) return xerr("cognn_rccl_rendezvous_tcp: rank 0 at %s:%d %s: %s", addr, port, what, strerror(errno));
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}

int cognn_rccl_exchange_create(const void* id128, int rank, int world, int device, void* compute_stream, cognn_rccl_exchange** out) {
    X_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "cognn_rccl_exchange_create: bad arguments");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    X_REQUIRE(e == hipSuccess && count > 0, "cognn_rccl_exchange_create: no HIP device available");
    X_REQUIRE(device >= 0 && device < count, "cognn_rccl_exchange_create: device %d out of range (count %d)", device, count);
    X_HIP(hipSetDevice(device));
    cognn_rccl_exchange* x = new cognn_rccl_exchange();
    x->rank = rank; x->world = world; x->device = device; x->compute = (hipStream_t)compute_stream;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&x->comm, world, id, rank);
    if (r != ncclSuccess) { delete x; return xerr("cognn_rccl_exchange_create: ncclCommInitRank failed: %s", ncclGetErrorString(r)); }
    if (hipStreamCreateWithFlags(&x->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&x->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->done, hipEventDisableTiming) != hipSuccess || hipMalloc((void**)&x->scratch, 16) != hipSuccess) {
        cognn_rccl_exchange_destroy(x);
        return xerr("cognn_rccl_exchange_create: stream / event creation failed");
    }
    *out = x;
    return 0;
}

int cognn_rccl_exchange_destroy(cognn_rccl_exchange* x) {
    if (!x) return 0;
    if (x->comm_stream) (void)hipStreamSynchronize(x->comm_stream);
    if (x->comm) (void)ncclCommDestroy(x->comm);
    if (x->ready) (void)hipEventDestroy(x->ready);
    if (x->done) (void)hipEventDestroy(x->done);
    for (hipEvent_t ev : x->done_ring) if (ev) (void)hipEventDestroy(ev);
    for (auto* v : {&x->timing, &x->spare})
        for (auto& pr : *v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (x->comm_stream) (void)hipStreamDestroy(x->comm_stream);
    if (x->scratch) (void)hipFree(x->scratch);
    delete x;
    return 0;
}

int cognn_rccl_exchange_begin(void* user, const cognn_xfer* xfers, int32_t n) {
    cognn_rccl_exchange* x = (cognn_rccl_exchange*)user;
    X_REQUIRE(x && (n == 0 || xfers), "cognn_rccl_exchange_begin: bad arguments");
    for (int32_t i = 0; i < n; ++i)
        X_REQUIRE(xfers[i].peer >= 0 && xfers[i].peer < x->world && xfers[i].ptr && xfers[i].bytes > 0,
                  "cognn_rccl_exchange_begin: message %d is malformed (peer %d, %lld bytes)", i, xfers[i].peer, (long long)xfers[i].bytes);
    // the messages may only leave once the kernels that write them have run ...
    X_HIP(hipEventRecord(x->ready, x->compute));
    X_HIP(hipStreamWaitEvent(x->comm_stream, x->ready, 0));
    if (x->timing.size() >= 256) {                           // nobody reads the clock: fold the rounds that have completed
        size_t keep = 0;
        for (auto& pr : x->timing) {
            float ms = 0;
            if (hipEventQuery(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { x->comm_ms += ms; x->spare.push_back(pr); }
            else x->timing[keep++] = pr;
        }
        x->timing.resize(keep);
    }
    std::pair<hipEvent_t, hipEvent_t> tm;
    if (!x->spare.empty()) { tm = x->spare.back(); x->spare.pop_back(); }
    else { X_HIP(hipEventCreate(&tm.first)); X_HIP(hipEventCreate(&tm.second)); }
    X_HIP(hipEventRecord(tm.first, x->comm_stream));
    X_NCCL(ncclGroupStart());
    for (int32_t i = 0; i < n; ++i) {
        const cognn_xfer& t = xfers[i];
        ncclResult_t r = t.is_send ? ncclSend(t.ptr, (size_t)t.bytes, ncclChar, t.peer, x->comm, x->comm_stream)
                                   : ncclRecv(t.ptr, (size_t)t.bytes, ncclChar, t.peer, x->comm, x->comm_stream);
        if (r != ncclSuccess) { (void)ncclGroupEnd(); x->spare.push_back(tm); return xerr("cognn_rccl_exchange_begin: %s failed: %s", t.is_send ? "ncclSend" : "ncclRecv", ncclGetErrorString(r)); }
        (t.is_send ? x->sent : x->received) += t.bytes;
    }
    { const ncclResult_t ge = ncclGroupEnd(); if (ge != ncclSuccess) { x->spare.push_back(tm); return xerr("cognn_rccl_exchange_begin: ncclGroupEnd failed: %s", ncclGetErrorString(ge)); } }
    X_HIP(hipEventRecord(tm.second, x->comm_stream));
    x->timing.push_back(tm);
    // ... and whoever consumes a received buffer (or overwrites a sent one) waits for this event on the compute stream
    X_HIP(hipEventRecord(x->done, x->comm_stream));
    hipEvent_t& slot = x->done_ring[x->rounds % cognn_rccl_exchange::kRing];
    if (!slot) X_HIP(hipEventCreateWithFlags(&slot, hipEventDisableTiming));
    X_HIP(hipEventRecord(slot, x->comm_stream));
    ++x->rounds;
    return 0;
}

int cognn_rccl_exchange_wait(void* user) {
    cognn_rccl_exchange* x = (cognn_rccl_exchange*)user;
    X_REQUIRE(x, "cognn_rccl_exchange_wait: null exchange");
    X_HIP(hipStreamWaitEvent(x->compute, x->done, 0));
    return 0;
}

// rounds are numbered from 0 at cognn_engine_set_exchange_rccl (cognn_exchange_wait_round_fn); the communication stream
// completes them in order, so the event of round r covers every earlier one
int cognn_rccl_exchange_wait_round(void* user, int64_t round) {
    cognn_rccl_exchange* x = (cognn_rccl_exchange*)user;
    X_REQUIRE(x && round >= 0, "cognn_rccl_exchange_wait_round: bad arguments");
    const int64_t r = x->round_base + round;
    X_REQUIRE(r < x->rounds, "cognn_rccl_exchange_wait_round: round %lld has not been started (%lld so far)", (long long)round, (long long)(x->rounds - x->round_base));
    X_HIP(hipStreamWaitEvent(x->compute, x->done_ring[r % cognn_rccl_exchange::kRing], 0));
    return 0;
}

int cognn_engine_set_exchange_rccl(cognn_engine* e, cognn_rccl_exchange* x) {
    X_REQUIRE(e && x, "cognn_engine_set_exchange_rccl: null argument");
    if (cognn_engine_set_exchange_async2(e, cognn_rccl_exchange_begin, cognn_rccl_exchange_wait, cognn_rccl_exchange_wait_round, x) != 0)
        return xerr("cognn_engine_set_exchange_rccl: %s", cognn_engine_last_error());
    x->round_base = x->rounds;
    return 0;
}

int cognn_rccl_exchange_stats(cognn_rccl_exchange* x, int64_t* rounds, int64_t* bytes_sent, int64_t* bytes_received) {
    X_REQUIRE(x, "cognn_rccl_exchange_stats: null exchange");
    if (rounds) *rounds = x->rounds;
    if (bytes_sent) *bytes_sent = x->sent;
    if (bytes_received) *bytes_received = x->received;
    return 0;
}

int cognn_rccl_exchange_time(cognn_rccl_exchange* x, double* comm_ms) {
    X_REQUIRE(x && comm_ms, "cognn_rccl_exchange_time: bad arguments");
    X_HIP(hipStreamSynchronize(x->comm_stream));
    for (auto& pr : x->timing) {
        float ms = 0;
        X_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        x->comm_ms += ms;
        x->spare.push_back(pr);
    }
    x->timing.clear();
    *comm_ms = x->comm_ms;
    return 0;
}

int cognn_rccl_exchange_ranks(cognn_rccl_exchange* x, int32_t* comm_count, int32_t* comm_rank, int64_t* ones_summed) {
    X_REQUIRE(x && comm_count && comm_rank && ones_summed, "cognn_rccl_exchange_ranks: bad arguments");
    int c = 0, r = -1;
    X_NCCL(ncclCommCount(x->comm, &c));
    X_NCCL(ncclCommUserRank(x->comm, &r));
    const uint64_t one = 1;
    X_HIP(hipStreamSynchronize(x->compute));
    X_HIP(hipMemcpyAsync(x->scratch, &one, 8, hipMemcpyHostToDevice, x->comm_stream));
    X_NCCL(ncclAllReduce(x->scratch, x->scratch + 1, 1, ncclUint64, ncclSum, x->comm, x->comm_stream));
    uint64_t sum = 0;
    X_HIP(hipMemcpyAsync(&sum, x->scratch + 1, 8, hipMemcpyDeviceToHost, x->comm_stream));
    X_HIP(hipStreamSynchronize(x->comm_stream));
    *comm_count = c; *comm_rank = r; *ones_summed = (int64_t)sum;
    return 0;
}

int cognn_rccl_exchange_barrier(cognn_rccl_exchange* x) {
    X_REQUIRE(x, "cognn_rccl_exchange_barrier: null exchange");
    X_HIP(hipEventRecord(x->ready, x->compute));
    X_HIP(hipStreamWaitEvent(x->comm_stream, x->ready, 0));
    X_NCCL(ncclAllReduce(x->scratch, x->scratch + 1, 1, ncclUint64, ncclSum, x->comm, x->comm_stream));
    X_HIP(hipStreamSynchronize(x->comm_stream));
    return 0;
}

}  // extern "C"
