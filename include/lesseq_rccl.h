/*
 * lesseq_rccl.h -- the gather step of a count / solve job that runs over several GPUs, for a C host.
 *
 * The reference scales out by running one process per slice gene_begin_idx..gene_end_idx of the sorted
 * gene list and concatenating their stdout (count/count.cpp:204-215).  Here a slice per GPU, and the
 * per-event records of the slices are put together by an RCCL all-gather over xGMI instead of `cat`:
 * lesseq_hip.h has the host side (lsq_shard_bounds, lsq_events_set_shard, lsq_results_pack_device,
 * lsq_gathered_unpack); this library (liblesseq_rccl.so, the only part that links librccl) has the
 * collective.  lesseq_amd/bin/{count,solve} drive it with LSQ_GPUS=N, one host thread per GPU
 * (lesseq_amd/csrc/lsq_cli.cpp); examples and the call order are in INTEGRATION.md.
 *
 * Conventions as in lesseq_hip.h: plain C, 0 or a negative lsq_status, message via lsq_rccl_last_error().
 */
#ifndef LESSEQ_RCCL_H
#define LESSEQ_RCCL_H

#include <stddef.h>
#include <stdint.h>
#include "lesseq_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsq_comm lsq_comm;       /* one rank's end of a communicator (an ncclComm_t and its rank / size) */

const char *lsq_rccl_last_error(void);

/* One process, n GPUs, one host thread each: n communicators in one call (ncclCommInitAll). */
int lsq_comm_init_all(int n, const int *devices, lsq_comm **comms /* n */);
/* One process per GPU: rank 0 makes an id (LSQ_COMM_ID_BYTES bytes), hands it to the others by any means the host
 * has (a file, a socket, MPI), and every process joins with its rank (ncclGetUniqueId / ncclCommInitRank). */
#define LSQ_COMM_ID_BYTES 128
int lsq_comm_unique_id(void *id /* LSQ_COMM_ID_BYTES */);
int lsq_comm_init_rank(int world, int rank, const void *id, int device, lsq_comm **out);
void lsq_comm_destroy(lsq_comm *comm);
/* The same two with a time limit (seconds; <= 0: none): the blocking RCCL call runs on a helper thread; when the time
 * is up the call returns LSQ_E_TIMEOUT and that thread is given up (it cannot be cancelled: a process that saw the
 * timeout should fall back to another exchange or leave) -- a rank whose peers never arrive does not hang the job. */
int lsq_comm_init_rank_for(int world, int rank, const void *id, int device, double seconds, lsq_comm **out);
int lsq_comm_init_all_for(int n, const int *devices, double seconds, lsq_comm **comms /* n */);
/* Ends whatever the communicator has in flight (ncclCommAbort) and frees it: the way out when a collective does not
 * complete (lsq_ctx_synchronize_for returned LSQ_E_TIMEOUT).  The other ranks' collectives fail or are aborted likewise. */
void lsq_comm_abort(lsq_comm *comm);
int lsq_rccl_version(void);           /* ncclGetVersion's integer of the RCCL this library runs on (0: the call failed) */
int lsq_comm_rank(const lsq_comm *comm);
int lsq_comm_size(const lsq_comm *comm);

/* The gather: every rank's packed record block (lsq_results_pack_device wrote it; stride_words words of 8 bytes,
 * the longest block of the job, shorter ones padded) into d_gathered (world * stride_words words) on every rank:
 * ncclAllGather on the result stream of the latest count's lane, behind the pack -- asynchronous, like the rest of a step.
 * lsq_ctx_synchronize, then lsq_gathered_unpack on a host copy, gives the whole job's tables. */
int lsq_gather(lsq_ctx *c, lsq_comm *comm, const void *d_block, void *d_gathered, uint64_t stride_words);
/* lsq_count + lsq_solve + lsq_results_pack_device(d_block) + lsq_gather in one call: a step of a loop over batches
 * (reads uploaded beforehand).  Asynchronous like its parts; the status of the first part that fails.  A loop that does
 * not synchronise between steps alternates between TWO d_block / d_gathered pairs (steps take the context's two lanes in
 * turn, lesseq_hip.h: step k+1's pack is not ordered behind step k's gather, which may still be reading its block). */
int lsq_step_gather(lsq_ctx *c, lsq_comm *comm, void *d_block, void *d_gathered, uint64_t stride_words);

/* The exchange of a READ-sharded job (every rank counts a slice of the reads against all events; lesseq_hip.h
 * lsq_text_stage_range / lsq_reads_upload_text_at): the latest count's class counts and matched bases -- as they lie on the
 * device, lsq_counts_device_words(c) words, the same order on every rank -- are copied into d_words and summed over the
 * ranks there (ncclAllReduce, uint64 sum, on the result stream: integer sums, any order, count/count.cpp:378,467-482).
 * Asynchronous; lsq_ctx_synchronize_for, then lsq_counts_import_device(c, d_words) makes the sums the counts that
 * lsq_solve and the getters work on. */
int lsq_allreduce_counts(lsq_ctx *c, lsq_comm *comm, void *d_words);

#ifdef __cplusplus
}
#endif
#endif
