/* lesseq_hip_dev.h -- developer entry points of liblesseq_hip.so, beside the product ABI of
 * lesseq_hip.h.  Nothing here replaces anything in the reference; the tools under tools/ and a
 * few tests use them to look inside a run (tools/kbench.py, tools/step_bench.py,
 * tests/test_parity_gpu.py).  They may change without notice. */
#ifndef LESSEQ_HIP_DEV_H
#define LESSEQ_HIP_DEV_H

#include <stdint.h>
#include "lesseq_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Counters the count kernels fill when LSQ_ABLATE has bit 256 set: [0] parked one-block reads,
 * [1] parked two-block reads, [2] walk steps, [3] walk lanes, [4] exception-list entries,
 * [5..7] reasons for parking one-block reads, [8..10], [13], [14] reasons for parking two-block reads, [11] / [12]
 * one-block steps of a wave with a parked read / all of them.  out16 holds 16 values. */
int lsq_debug_counters(lsq_ctx *c, unsigned long long *out16);
/* (start, end, steps of the general walk, reads those looked at) of each workgroup of the last count launch, times in 100 MHz
 * ticks (developer library with LSQ_ABLATE bit 4194304; otherwise *n = 0): out holds 4 * cap values; the first *n_workers
 * workgroups are the pool-n workers */
int lsq_debug_wg_trace(lsq_ctx *c, unsigned long long *out, unsigned long long cap, unsigned long long *n, unsigned long long *n_workers);

/* The combined slot offsets of a method's buckets (n_buckets + 1 values): the work partition of
 * the count kernels. */
int lsq_debug_slot_offsets(lsq_ctx *c, int method, unsigned long long *out, unsigned long long n);
/* The share plan of a count launch as lsq_count makes it, on given numbers (host only, no device): slot_off = the buckets' slot
 * offsets (n_buckets + 1), packed = 1 per bucket the streaming kernel visits, n1 / n2 / look1 / look2 = per bucket the records of
 * the one- and two-block pool and the walk's looks at each (null: every slot weighs the same), costs4 = {two-block record, walk
 * look, visit, taper}; writes grid + 1 bounds in slots. */
int lsq_debug_plan_shares(const unsigned long long *slot_off, const unsigned char *packed, const unsigned long long *n1, const unsigned long long *n2,
                          const unsigned *look1, const unsigned *look2, unsigned long long n_buckets, unsigned long long grid,
                          const double *costs4, int weighted, int snap, unsigned long long *cuts);
/* An offset table of a method -- which = 0: slots per bucket, 1 / 2: the one- / two-block pool per bucket (n_buckets + 1
 * values each), 3: the share bounds of the count launch's streaming workgroups (their number + 1), 4 / 5: the ingest's
 * estimate of the one- / two-block reads per bucket that the streaming loops leave to the general walk; *n = values written. */
int lsq_debug_offsets(lsq_ctx *c, int method, int which, unsigned long long *out, unsigned long long cap, unsigned long long *n);

/* Another placement of the events in the EM grid (sixteen per wave; 0xFFFFFFFF = empty place):
 * the first n_small_places entries go to the lean kernel and must be events with at most two
 * isoforms and one (method, class) pair per lane.  Results do not depend on the placement
 * (tests/test_parity_gpu.py::test_em_numbers_do_not_depend_on_which_events_share_a_wave). */
int lsq_debug_set_em_order(lsq_ctx *c, const uint32_t *order, unsigned n_small_places, unsigned n_places);

/* The rate a plain streaming read of `bytes` of device memory reaches on this device, in GB/s (best launch over
 * eight grid / unroll combinations): the practical ceiling bench.py reports beside the nominal HBM peak. */
int lsq_debug_stream_read_rate(lsq_ctx *c, unsigned long long bytes, double *gb_per_s);

/* Throws a C++ exception below the boundary -- kind 0: std::bad_alloc, 1: std::length_error, 2: an int -- so that a test
 * can see what a caller gets: LSQ_E_INTERNAL and a message, never std::terminate (lesseq_hip.h "no exceptions across
 * the boundary").  kind 3 does the same from inside a helper thread of a ThreadGroup (lsq_internal.hpp). */
int lsq_debug_throw(int kind);

/* Which of the device parse's three kernels the latest MRF text of the context went through (lsq_reads_upload_mrf / _text): tiles the
 * fast kernel handed to the byte-walking kernel (more delimiters than its LDS tables hold), lines it handed to the shared splitter
 * (another shape than a read's), and whether the whole file went through the byte-walking kernel (more than 64 chromosomes, or a line
 * list that ran over).  A file of reads: 0, 0, 0 -- a test holds that, so that a silent fall-back cannot pass as the fast path. */
int lsq_debug_last_parse_paths(const lsq_ctx *c, unsigned *tiles_handed, unsigned *lines_listed, unsigned *all_slow);

/* HIP_VERSION the library was compiled against and hipRuntimeGetVersion() of the runtime it found in the process (0
 * when that call fails, e.g. without a driver): a binding that loads another runtime first (PyTorch's) can compare. */
int lsq_debug_hip_versions(int *compiled, int *runtime);

#ifdef __cplusplus
}
#endif
#endif
