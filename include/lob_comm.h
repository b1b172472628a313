/*
 * lob_comm.h — C ABI of the multi-GPU weight exchange (SURVEY.md §8e, §8b `lob_theta_allreduce`).
 *
 * One process per GPU, one engine (include/lob_engine.h) and one communicator
 * per process.  Books are independent given theta, so the only exchange on the
 * whole path is the shared tile-coded weight vector: every `sync_every` steps
 * each rank forms delta = theta - theta_sync, the deltas are summed over ranks
 * by ONE in-place RCCL all-reduce (f64, SUM) over xGMI, issued on the engine's
 * own HIP stream directly on the engine's delta buffer (no staging copy, no
 * host synchronisation), and every rank sets theta = theta_sync + sum(delta).
 * This replaces the reference's unlocked shared rl::Agent of its training
 * threads (src/main.cpp:196-206: N std::threads calling Learner::RunEpisode on
 * one Agent*) at sync granularity.
 *
 * Lives in its own library (liblob_comm.so, links librccl) so that the engine
 * library itself has no RCCL dependency.  Same conventions as lob_engine.h:
 * LOB_OK or a negative LOB_E* code, message through lob_last_error().
 */
#ifndef LOB_COMM_H
#define LOB_COMM_H

#include <stdint.h>

#include "lob_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LOB_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

enum { LOB_COMM_SUM = 0, LOB_COMM_MAX = 1 };

typedef struct lob_comm lob_comm;

/* Rank 0 creates the rendezvous token (ncclGetUniqueId) and hands it to the
 * other ranks by any means (lob_run: a pipe; bench.py: a file). */
int lob_comm_get_id(uint8_t id[LOB_COMM_ID_BYTES]);
/* Join the `world`-rank communicator as `rank` on GPU `device`
 * (ncclCommInitRank).  Collective: every rank must call it. */
int lob_comm_create(const uint8_t id[LOB_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, lob_comm** out);
/* The same with a file rendezvous on ONE node: rank 0 writes the token to
 * `path` (atomically, via rename), the others wait up to `timeout_s` seconds
 * for it.  Rank 0 removes the file once every rank has joined. */
int lob_comm_create_file(const char* path, int32_t rank, int32_t world, int32_t device, int32_t timeout_s, lob_comm** out);
void lob_comm_destroy(lob_comm* c);
int32_t lob_comm_rank(const lob_comm* c);
int32_t lob_comm_world(const lob_comm* c);

/* In-place SUM all-reduce of `count` doubles in HBM on `hip_stream`
 * (hipStream_t as void*); asynchronous with respect to the host. */
int lob_comm_allreduce_f64(lob_comm* c, double* dev_buf, int64_t count, void* hip_stream);
/* Small host-side reductions (timings, counters): every rank passes `n` <= 64
 * doubles and receives the reduction over ranks.  Synchronises. */
int lob_comm_reduce_host_f64(lob_comm* c, double* vals, int32_t n, int32_t op);
int lob_comm_barrier(lob_comm* c);
/* All-gather of `count` uint32 words per rank into dev_recv[world][count] on `hip_stream` (the ranks' written-weights maps). */
int lob_comm_allgather_u32(lob_comm* c, const uint32_t* dev_send, uint32_t* dev_recv, int64_t count, void* hip_stream);
/* Bind the calling thread to the host cores next to GPU `device` (sysfs local_cpulist of its PCIe function); *n_cpus = how
 * many, 0 when the topology cannot be read (nothing changed). */
int lob_comm_pin_host_thread(int32_t device, int32_t* n_cpus);

/* The periodic weight exchange of one engine, on the engine's stream.  lob_delta_init must have been called once (after
 * create / theta_set).
 *   shared theta on the fast path (SARSA / Q(lambda)): SPARSE -- all-gather of the ranks' written-weights maps (2.5 MB each at
 *     memory_size 20 M), union -> compact layout, all-reduce(SUM) of the packed deltas (a few hundred thousand doubles),
 *     scatter (lob_delta_sparse_*); one host synchronisation (the element count).  LOB_DENSE_EXCHANGE=1 forces the dense path;
 *   otherwise DENSE: lob_delta_begin_async -> all-reduce(SUM) of memory_size doubles in place -> lob_delta_apply, fully
 *     asynchronous.
 * Same result either way: theta = theta_sync + sum over ranks of (theta - theta_sync). */
int lob_theta_allreduce(lob_engine* e, lob_comm* c);
/* What the exchanges since the last call cost, from HIP events on the engine's stream (synchronises):
 * out = [exchanges, pack / delta ms, collectives ms, apply ms, bytes through the collectives, sparse exchanges, ranks as
 * ncclCommCount reports them]. */
int lob_comm_exchange_stats(lob_comm* c, double out[7]);

#ifdef __cplusplus
}
#endif
#endif /* LOB_COMM_H */
