/* cognn_exchange.h — native share-exchange transports of the engine (libcognn_hip.so).
 *
 * Replaces the reference's k x k TCP channel mesh (include/engine.h:157-201: ports 1712 + max(i,j), one
 * osuCrypto::Channel per ordered pair) and its message helpers (include/comm_sync.h:245-277
 * sendShareVecVec / recvShareVecVec = tag string + Boost binary archive, blocking recv).  Here every logical message of a
 * round is a flat device buffer and a round is ONE RCCL p2p group on a communication stream:
 *
 *     compute stream ──record──▶ comm stream: ncclGroupStart, ncclSend/ncclRecv per message, ncclGroupEnd ──record──▶
 *     compute stream waits for that event only when it consumes received data (cognn_exchange_wait_fn)
 *
 * so nothing synchronises with the host and no interpreter sits in the loop.  xGMI is point-to-point: the pairwise
 * pattern of the engine (owner <-> co-party openings, partial-sum segments, co-share replicas) drives all links of a GPU
 * concurrently inside one group.
 */
#ifndef COGNN_EXCHANGE_H_
#define COGNN_EXCHANGE_H_
#include <stdint.h>
#include "cognn_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COGNN_RCCL_ID_BYTES 128

typedef struct cognn_rccl_exchange cognn_rccl_exchange;

const char* cognn_exchange_last_error(void);
/* rank 0 creates the communicator id (ncclGetUniqueId) and hands the 128 bytes to every other rank by any means */
int cognn_rccl_unique_id(void* id128);
/* TCP rendezvous for processes that have no other channel (bin/gcn-optimize in one-process-per-GPU mode): rank 0 listens on
 * addr:port and serves the id to world-1 clients; the others connect (retrying for timeout_s) and read it.  Takes the place
 * of the reference's fixed-port session setup (engine.h:166-201); port 0 = 1712, the reference's base port. */
int cognn_rccl_rendezvous_tcp(const char* addr, int port, int rank, int world, double timeout_s, void* id128);
/* ncclCommInitRank on `device`; compute_stream = the stream the engine's kernels run on (hipStream_t, NULL = default) */
int cognn_rccl_exchange_create(const void* id128, int rank, int world, int device, void* compute_stream, cognn_rccl_exchange** out);
int cognn_rccl_exchange_destroy(cognn_rccl_exchange* x);
/* the pair of callbacks for cognn_engine_set_exchange_async (user = the cognn_rccl_exchange*) */
int cognn_rccl_exchange_begin(void* user, const cognn_xfer* xfers, int32_t n);
int cognn_rccl_exchange_wait(void* user);
/* cognn_exchange_wait_round_fn: the compute stream waits for the event of round `round` (numbered from 0 at
 * cognn_engine_set_exchange_rccl; one event per round, 64 in a ring) - later rounds stay in flight */
int cognn_rccl_exchange_wait_round(void* user, int64_t round);
/* shorthand for cognn_engine_set_exchange_async2(e, cognn_rccl_exchange_begin, cognn_rccl_exchange_wait,
 * cognn_rccl_exchange_wait_round, x) */
int cognn_engine_set_exchange_rccl(cognn_engine* e, cognn_rccl_exchange* x);
/* rounds started, bytes sent and received by this rank so far (achieved xGMI GB/s = bytes / measured time) */
int cognn_rccl_exchange_stats(cognn_rccl_exchange* x, int64_t* rounds, int64_t* bytes_sent, int64_t* bytes_received);
/* total time the p2p groups of all rounds so far took on the communication stream (HIP events around every group; waits
 * for the stream to drain).  bytes / this time = the achieved xGMI rate of this rank. */
int cognn_rccl_exchange_time(cognn_rccl_exchange* x, double* comm_ms);
/* what the communicator itself says about the job: ncclCommCount, ncclCommUserRank and the sum of an all-reduce of ones over it
 * (= the number of ranks that took part) - bench.py prints them so that a multi-GPU record shows RCCL saw N ranks.  Collective. */
int cognn_rccl_exchange_ranks(cognn_rccl_exchange* x, int32_t* comm_count, int32_t* comm_rank, int64_t* ones_summed);
/* an all-ranks barrier on the communication stream followed by a host wait (end-of-run hand-shake, sendFinish/recvFinish
 * of ss_...h:270-272) */
int cognn_rccl_exchange_barrier(cognn_rccl_exchange* x);

#ifdef __cplusplus
}
#endif
#endif /* COGNN_EXCHANGE_H_ */
