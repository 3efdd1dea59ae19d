/*
 * lesseq_hip.h -- C ABI of the MI355X-native count + solve path of LESSeq.
 *
 * The reference (gersteinlab/LESSeq) has no plugin / FFI seam: its `count` and `solve`
 * programs are monolithic main() bodies (count/count.cpp:88-501, solve/solve.cpp:102-856).
 * This header is the seam a maintainer would cut: each entry point replaces a contiguous
 * stretch of those main() bodies, cited per function.  INTEGRATION.md shows the call sites.
 *
 * Conventions: plain C, caller-owned buffers, no exceptions across the boundary (every entry
 * point that can allocate catches what is thrown below it and returns LSQ_E_INTERNAL; helper
 * threads are joined on every way out).  Every
 * function returns LSQ_OK (0) or a negative lsq_status; lsq_last_error() gives the text for
 * the calling thread.  Coordinates are 0-based half-open, as the reference holds them after
 * `start - 1` (count/count.cpp:319,323).  One lsq_ctx per GPU, used from one host thread.
 *
 * Two groups:
 *   host-only  (no GPU touched): annotation loading, event compilation, MRF parsing,
 *              text formatting, the synthetic workload generator;
 *   device     (HIP, gfx950):    context, uploads, the count and EM kernels, result fetch.
 * There is no CPU implementation of the device group: without a usable GPU these calls
 * fail with LSQ_E_DEVICE.  (Two things inside the device group do run on the host, on reads that were
 * filtered and pooled on the device: the exact-order EM of guard-band events, and genes beyond the
 * kernel limits below -- lesseq_amd/csrc/lsq_replay.hip.)
 */
#ifndef LESSEQ_HIP_H
#define LESSEQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSQ_ABI_VERSION 2

typedef enum {
	LSQ_OK = 0,
	LSQ_E_ARG = -1,          /* bad argument / malformed table */
	LSQ_E_IO = -2,           /* file cannot be opened / read */
	LSQ_E_FORMAT = -3,       /* unknown file format / read type literal (reference: exit 1) */
	LSQ_E_PARSE = -4,        /* lexical_cast failure (reference: exit 1) */
	LSQ_E_RANGE = -5,        /* coordinate or size outside what the device tables hold */
	LSQ_E_UNSUPPORTED = -6,  /* event shape outside the device kernels' limits */
	LSQ_E_DEVICE = -7,       /* HIP error, or no gfx950 device */
	LSQ_E_STATE = -8,        /* call order violated (e.g. count before uploads) */
	LSQ_E_INTERNAL = -9,     /* out of memory, or a C++ exception caught at the boundary (never thrown across it) */
	LSQ_E_TIMEOUT = -10      /* lsq_ctx_synchronize_for: the context's streams had not drained when the time was up */
} lsq_status;

/* kernel limits (what the count and EM kernels hold in registers / LDS per event) */
#define LSQ_MAX_SEGMENTS 32   /* atomic exon segments per event (LESSeq local events: <= 4) */
#define LSQ_MAX_ISOFORMS 6    /* isoforms per event (LESSeq local events: 2) */
#define LSQ_MAX_METHODS 8     /* read files ("sampling methods") per run */
/* Genes beyond the two kernel limits above are not refused: they are evaluated on the host inside lsq_count / lsq_solve
 * (same rules, the reference's own per-read EM; those calls then block).  The hard limits are 64 segments and: */
#define LSQ_HOST_MAX_ISOFORMS 16

const char *lsq_last_error(void);
int lsq_abi_version(void);
/* The library's own warnings (e.g. "the exception list of read file m overflowed: every read was counted again") go to stderr in
 * the reference's log format (jsc/util/log.hpp:22-79, "[LOG date time WARNING] text") when level >= 1; default 2, like the
 * reference's reporting level; the executables pass their log_level argument on. */
void lsq_set_log_level(int level);

/* ------------------------------------------------------------------------------------
 * Host-only group
 * ------------------------------------------------------------------------------------ */

typedef struct lsq_annotation lsq_annotation;   /* selected genes + their isoform records */
typedef struct lsq_events lsq_events;           /* compiled event tables (host copy) */
typedef struct lsq_reads lsq_reads;             /* parsed reads of one MRF file, file order */

/* Replaces count/count.cpp:135-216 and solve/solve.cpp:152-329: loads the isoforms and the
 * gene -> isoform map, selects genes whose index in the bytewise-sorted gene-name set lies in
 * [gene_begin_idx, gene_end_idx).  Isoform formats: "LH_GENE_TXT" (the only one `count` and
 * `classify` take), and solve's "UCSC_GENE_TXT" (jsc/bioinfo/gene_anno.hpp:59-117),
 * "UCSC_GFF", "WORMBASE_GFF2", "GENELETS_GFF3" (one line per exon; an isoform's exons are
 * merged with interval_list::add_interval in file order, solve/solve.cpp:160-296).  Map
 * formats: "UCSC_GENE2ISOFORM", and solve's "WORMBASE_GENE2ISOFORMS" (gene iso1;iso2;...).
 * Any other literal gives LSQ_E_FORMAT; the executables keep count/classify to their two. */
int lsq_annotation_load(const char *isoform_format, const char *isoforms_path,
                        const char *g2i_format, const char *g2i_path,
                        uint64_t gene_begin_idx, uint64_t gene_end_idx,
                        lsq_annotation **out);
void lsq_annotation_free(lsq_annotation *a);
int64_t lsq_annotation_num_genes(const lsq_annotation *a);      /* selected genes */
int64_t lsq_annotation_num_isoforms_loaded(const lsq_annotation *a);
int64_t lsq_annotation_num_genes_loaded(const lsq_annotation *a);

/* Replaces count/count.cpp:231-258 + the per-gene ARS construction at :394-416
 * (common/splicing_graph.h:88-169,235-252,318-361; common/accessible_read_starts.h:48-89,
 * 221-274; jsc/util/interval_list.hpp:462-503): atomic segments, isoform masks, event
 * spans, per-chromosome covered regions, and for each of the n_methods read files the
 * per-isoform ARS total.  read_types[m] is "SHORT_READ" or "MEDIUM_READ" (LSQ_E_FORMAT
 * otherwise, the reference's unknown_readtype_err). */
int lsq_events_compile(const lsq_annotation *a, int n_methods, const char *const *read_types,
                       const uint64_t *expected_read_lengths, lsq_events **out);
void lsq_events_free(lsq_events *e);

/* read-only views into compiled events, in output order (bytewise-sorted gene names) */
int64_t lsq_events_count(const lsq_events *e);
int64_t lsq_events_total_isoforms(const lsq_events *e);
const char *lsq_events_gene_name(const lsq_events *e, int64_t ev);
const char *lsq_events_chrom(const lsq_events *e, int64_t ev);
const char *lsq_events_strand(const lsq_events *e, int64_t ev);
int lsq_events_num_isoforms(const lsq_events *e, int64_t ev);
int lsq_events_num_segments(const lsq_events *e, int64_t ev);
const char *lsq_events_isoform_name(const lsq_events *e, int64_t ev, int iso);
/* segment n of the event as [start,end); isoform mask bit n set iff the isoform holds it */
int lsq_events_segment(const lsq_events *e, int64_t ev, int n, int64_t *start, int64_t *end);
uint64_t lsq_events_isoform_mask(const lsq_events *e, int64_t ev, int iso);
uint64_t lsq_events_isoform_length(const lsq_events *e, int64_t ev, int iso);
uint64_t lsq_events_ars(const lsq_events *e, int method, int64_t ev, int iso);
int lsq_events_span(const lsq_events *e, int64_t ev, int64_t *gene_start, int64_t *gene_end);
int64_t lsq_events_num_buckets(const lsq_events *e);
int64_t lsq_events_lds_table_bytes(const lsq_events *e);   /* largest bucket image + histogram */
int64_t lsq_events_host_genes(const lsq_events *e);        /* genes beyond the kernel limits below: evaluated on the host (lsq_host_evaluated) */
/* Restricts the device plan to events [first_event, first_event + n_events) of the output order
 * (the reference's own scale-out unit, count/count.cpp:204-215) while the covered regions -- and so
 * the load-time read filter -- stay those of the whole selected range: every shard then gives
 * exactly the rows the unsharded run gives for its events.  Call before lsq_events_upload. */
int lsq_events_set_shard(lsq_events *e, uint64_t first_event, uint64_t n_events);

/* Replaces count/count.cpp:279-336 (== solve/solve.cpp:429-486) minus the containment
 * filter: parses an MRF_SINGLE file into blocks in file order.  A lexical_cast failure
 * gives LSQ_E_PARSE; a format literal other than "MRF_SINGLE" LSQ_E_FORMAT.  Chromosome and
 * strand strings are interned against the events' dictionaries (unknown chromosomes get an
 * id that matches no covered region). n_threads <= 0 picks the host's core count. */
int lsq_mrf_parse(const char *read_format, const char *path, lsq_events *e,
                  int n_threads, lsq_reads **out);
/* The same for any read format the reference's `solve` takes (solve/solve.cpp:413-634): "MRF_SINGLE"
 * (this is then lsq_mrf_parse), and the name-keyed "UCSC_GFF" (a line per block), "UCSC_BED" (a line
 * per read, kept or dropped as a whole by its span) and "WORMBASE_GFF3" (a line per read with an
 * optional intron).  Lines of one name make one read; its names decide span-start ties against gene
 * names (solve/solve.cpp:70-98).  `count` takes MRF_SINGLE only, as in the reference. */
int lsq_reads_parse(const char *read_format, const char *path, lsq_events *e, int n_threads, lsq_reads **out);
/* Wraps caller-made arrays as a read set without copying (the arrays must outlive it).
 * blk_off has n_reads+1 entries; blocks are 0-based half-open; chrom_id / strand_id index
 * lsq_events_chrom_id() / lsq_events_strand_id() dictionaries; line_no is the 1-based line
 * number after the header (read name "read-<line_no>", count/count.cpp:293-295). */
int lsq_reads_wrap(uint64_t n_reads, const uint64_t *blk_off, const uint32_t *line_no,
                   const int32_t *blk_start, const int32_t *blk_end,
                   const uint16_t *blk_chrom_id, const uint8_t *blk_strand_id, lsq_reads **out);
void lsq_reads_free(lsq_reads *r);
uint64_t lsq_reads_count(const lsq_reads *r);
uint64_t lsq_reads_num_blocks(const lsq_reads *r);
int lsq_events_chrom_id(lsq_events *e, const char *chrom);     /* interns; >= 0 */
int lsq_events_strand_id(lsq_events *e, const char *strand);   /* interns; >= 0 */
const char *lsq_events_strand_name(const lsq_events *e, int strand_id);   /* NULL when unknown */
/* The arrays of a read set (lsq_reads_wrap's layout); they live as long as the read set. */
int lsq_reads_arrays(const lsq_reads *r, const uint64_t **blk_off, const uint32_t **line_no,
                     const int32_t **blk_start, const int32_t **blk_end,
                     const uint16_t **blk_chrom_id, const uint8_t **blk_strand_id);

/* ------------------------------------------------------------------------------------
 * Device group
 * ------------------------------------------------------------------------------------ */

typedef struct lsq_ctx lsq_ctx;

int lsq_ctx_create(int device_id, lsq_ctx **out);
/* The same with flags.  LSQ_CTX_LANES_IN_BACKGROUND: only the upload stream is made before the call returns; the step lanes' streams
 * and events (0.04 s of runtime calls) are made by a helper thread, which the first call that needs a lane -- lsq_events_upload
 * comes ahead of all of them -- or lsq_ctx_destroy joins.  lsq_text_stage needs no lane: an executable copies its reads to HBM meanwhile. */
#define LSQ_CTX_LANES_IN_BACKGROUND 1u
int lsq_ctx_create_with(int device_id, unsigned flags, lsq_ctx **out);
void lsq_ctx_destroy(lsq_ctx *c);
/* The context works on HIP streams of its own.  Uploads and ingest run on one (returned here as hipStream_t in a
 * void*).  A step -- lsq_count, lsq_solve, the hand-off of its results -- runs on one of two LANES that consecutive
 * lsq_count calls take in turn: a lane has a count stream (the streaming kernels), a result stream (exception pass, EM,
 * lsq_results_copy_device / lsq_results_pack_device, behind an event recorded after the count), a counter set and the
 * EM's output arrays.  These calls only submit work; step k+1's count runs beside step k's EM.  Hence two rules for a
 * host that submits steps without waiting in between:
 *   - lsq_ctx_result_stream names the LATEST count's lane: ask again after every lsq_count, never cache it;
 *   - the device buffers handed to consecutive unsynchronised steps (d_block / d_gathered of lsq_step_gather, the
 *     arguments of lsq_results_copy_device) must be DISTINCT -- two of each, alternated, as the lanes are: step k+1's
 *     pack runs on the other lane and is not ordered behind a collective that still reads step k's block.
 * lsq_ctx_synchronize waits for all of the context's streams; the host-side result getters do so themselves. */
void *lsq_ctx_stream(lsq_ctx *c);
int lsq_ctx_synchronize(lsq_ctx *c);
/* The same with a time limit (seconds): LSQ_E_TIMEOUT when work is still queued after it -- e.g. a collective of a
 * multi-GPU job (lesseq_rccl.h) whose peer never arrived; the caller then aborts the communicator (lsq_comm_abort)
 * instead of waiting for ever.  Polls the streams; nothing is cancelled by the call itself. */
int lsq_ctx_synchronize_for(lsq_ctx *c, double seconds);

/* Uploads the compiled tables: per-bucket LDS images (bin directory, event records,
 * segments, isoform masks), tie-break records and G = 1/ARS. */
int lsq_events_upload(lsq_ctx *c, lsq_events *e);

/* Replaces the retained-read state the reference builds at count/count.cpp:319-324 and
 * :348-364.  The parsed blocks are copied to the device as they are; HIP kernels apply the
 * per-block containment filter against the covered regions, merge kept blocks with
 * interval_list::add_interval semantics, and store each retained read in the bucket of its
 * first kept base, split into 1-block / 2-block / n-block pools (structure-of-arrays in
 * HBM).  A read may keep at most 16 separate blocks (LSQ_E_RANGE beyond).  method in
 * [0, n_methods). */
int lsq_reads_upload(lsq_ctx *c, int method, const lsq_reads *r);
/* The same, from the MRF_SINGLE text itself: replaces count/count.cpp:279-336 and :348-364 in one
 * call.  The file's bytes are copied to HBM and parsed there (newline scan, one lane per line, the
 * reference's find/substr field arithmetic and lexical_cast<long> rules), then ingested as above;
 * no parsed array ever exists on the host.  Status as lsq_mrf_parse (LSQ_E_FORMAT, LSQ_E_PARSE with
 * the first failing line in lsq_last_error(), LSQ_E_IO); LSQ_E_UNSUPPORTED when the file holds a
 * strand string longer than 7 bytes (lsq_mrf_parse + lsq_reads_upload take those). */
int lsq_reads_upload_mrf(lsq_ctx *c, int method, const char *read_format, const char *path);
/* The same in two steps, for callers that want the copy under way before the event tables exist (the
 * executables start it on a second thread while the annotation is still being read):
 * lsq_text_stage copies the file's bytes to HBM and needs only the context; lsq_reads_upload_text
 * parses and ingests them (status as lsq_reads_upload_mrf).  The text may be freed afterwards. */
typedef struct lsq_text lsq_text;
int lsq_text_stage(lsq_ctx *c, const char *path, lsq_text **out);
void lsq_text_free(lsq_text *t);
int lsq_reads_upload_text(lsq_ctx *c, int method, const char *read_format, lsq_text *t);
/* A slice of a file for one of several processes that share it (lesseq_amd/dist.py::run_read_sharded):
 * the bytes [byte_begin, byte_end), which must start on a line boundary; lsq_text_lines counts its
 * newlines (so that every process can work out the file-wide number of its first line: read names are
 * "read-<line>", count/count.cpp:293-295); lsq_reads_upload_text_at parses it with has_header = 1 only
 * for the slice that holds the file's first line, first_line = the number of its first data line. */
int lsq_text_stage_range(lsq_ctx *c, const char *path, uint64_t byte_begin, uint64_t byte_end, lsq_text **out);
int lsq_text_lines(lsq_ctx *c, lsq_text *t, uint64_t *n_newlines);
int lsq_reads_upload_text_at(lsq_ctx *c, int method, const char *read_format, lsq_text *t, int has_header, uint64_t first_line);
/* The device parser's blocks copied back to the host (same arrays lsq_mrf_parse makes; for tools
 * and tests).  Needs lsq_events_upload first. */
int lsq_mrf_parse_device(lsq_ctx *c, const char *read_format, const char *path, lsq_reads **out);
/* milliseconds of the last device parse: host-to-device copy of the text, and the parse kernels */
int lsq_last_mrf_timing(lsq_ctx *c, float *h2d_ms, float *parse_ms);
/* The loader as a chain of device passes (what replaces count/count.cpp:279-364 on the device): newline count, route
 * (parse + containment filter + block merge + bucket), partition count, partition scatter, group classify, group
 * offsets, group place.  lsq_ingest_stage_count / _name list them; lsq_last_ingest_stages gives, for the latest
 * lsq_reads_upload* of the context, the device milliseconds of every pass (HIP events on the library's stream around the
 * pass's launches) and the bytes the pass has to move at least (its input read once, its output written once) -- the
 * two halves of a per-pass HBM roofline.  `capacity` entries of `ms` / `bytes` are written at most (either may be null). */
int lsq_ingest_stage_count(void);
const char *lsq_ingest_stage_name(int stage);
int lsq_last_ingest_stages(const lsq_ctx *c, float *ms, uint64_t *bytes, int capacity);
uint64_t lsq_reads_retained(const lsq_ctx *c, int method);      /* "loaded N reads" log line */
uint64_t lsq_reads_retained_blocks(const lsq_ctx *c, int method);
/* Of the retained reads, those kept in the pools: reads whose first base lies in the span of an event planned on this
 * context.  Without a shard that is every retained read; with lsq_events_set_shard the slice's share (a read that starts
 * in no event of the slice is a candidate of none of them, count/count.cpp:429-432,463, and is dropped at ingest). */
uint64_t lsq_reads_pooled(const lsq_ctx *c, int method);
uint64_t lsq_reads_pooled_blocks(const lsq_ctx *c, int method);      /* ... and their blocks: what one lsq_count streams */
/* How the one- and two-block reads of a read file lie in HBM.  *compact = 1: 4 bytes a block -- 22 bits of offset (the
 * first block's from the first base of its bucket of events minus 2 Mi, the second block's from the end of the first) and
 * 10 bits of length; a read with a block of 1 024 bases or more, or an offset that does not fit, is kept with the
 * many-block reads, blocks in full.  *compact = 0 (more than 1 in 16 one- and two-block reads would not fit, or option
 * "compact_pools" 0): (start, end) per block, 8 bytes.  Either way every read takes part in lsq_count with its own
 * coordinates.  *pool_bytes: bytes of block coordinates resident in HBM; pool_reads[3]: reads kept as one-block records,
 * as two-block records, with the many-block reads.  Null: not wanted. */
int lsq_reads_pool_format(const lsq_ctx *c, int method, int *compact, uint64_t *pool_bytes, uint64_t *pool_reads);

/* Replaces the per-gene candidate scan + Read::build + compatibility + validity + counting
 * (count/count.cpp:420-482 == solve/solve.cpp:719-793; common/read.h:44-79,198-274): one
 * pass over every uploaded method's reads; fills per-(method, event, compatibility class)
 * read counts and matched-base sums on the device.  Asynchronous on the context stream. */
int lsq_count(lsq_ctx *c);
/* Per read file of the latest lsq_count (arrays of n_methods, either may be NULL; synchronises): the
 * (read, event) pairs the streaming kernel handed to the exception pass -- span-start ties that the
 * strand / name order decides (count/count.cpp:64-85), two blocks that touch -- and whether that list
 * overflowed, in which case two kernels behind the exception pass on the result stream zeroed the
 * method's tables and counted every read again (slow, complete).  The decision is taken on the
 * device, so every table that leaves the context -- also through lsq_results_copy_device in a loop
 * that never synchronises -- is whole. */
int lsq_count_status(lsq_ctx *c, uint32_t *exceptions, uint32_t *recounted);
/* How the latest lsq_count launched its streaming kernel: one-block reads a lane settles per look at the tables (4:
 * lsq_count_fast_kernel<.., 2>, 8: lsq_count_fast_kernel<true, 4>) and resident workgroups per compute unit.  Either
 * pointer may be null. */
int lsq_count_launch_info(lsq_ctx *c, uint32_t *reads_per_look, uint32_t *workgroups_per_cu);
/* What the latest lsq_count evaluated on the HOST: genes beyond the kernel limits above (more than LSQ_MAX_ISOFORMS isoforms
 * or LSQ_MAX_SEGMENTS segments, or a cluster of overlapping genes too large for the LDS) and the reads their clusters held.
 * Results are the reference's either way; throughput is not (host threads: LSQ_THREADS), so a caller may want to say so --
 * the executables do at log level 1.  Either pointer may be null. */
int lsq_host_evaluated(const lsq_ctx *c, uint64_t *n_genes, uint64_t *n_reads);
/* Tuning knobs; results never depend on them.  "grid_multiplier" (workgroups per resident slot of the
 * count kernel's grid, 0 = chosen from the read set's skew), "exception_capacity" (entries of a read
 * file's exception list, 0 = a quarter of its reads and at least 65 536; applies to read sets uploaded
 * afterwards), "recount_every_read" (1: every count is redone by the one-lane-per-read kernel, a
 * self-check), "em_guard_band" (lsq_set_em_guard_band), "snap_shares" (0: the count kernel's workgroup
 * shares are cut at even cost instead of at bucket ends), "share_weighted" (default 1: the shares are equal in estimated cost
 * -- "share_cost_two_block": a two-block record in one-block records, default 4.3; "share_cost_parked": one look of the general
 * walk at a read the streaming loops leave to it, default 9, counted per bucket by the ingest; "share_cost_visit": a bucket's
 * staging and flush, default 7 000; "share_cost_hot": extra cost of a record of a cell or junction group of 8 192 records or more,
 * default 0.5 (a deep gene's reads all add to the same few LDS counters) -- and fall off in size along the grid, "share_taper": the last share as a fraction of the
 * first, 0 = automatic, 0.5 for evenly deep read sets and 0.25 for skewed ones; 0: shares equal in reads), "compact_pools" (0: wide pool records for read sets
 * uploaded afterwards, see lsq_reads_pool_format), "em_regroup" (0: lsq_solve keeps the
 * placement of events in its grid chosen at lsq_events_upload; default 1: each of the two step lanes
 * re-sorts the placement by the iteration counts of one of its own earlier solves, refreshed every
 * sixteenth solve, so that events of similar cost share a wavefront), "count_streams" (1: every lsq_count on one
 * stream; default 2: a stream per step lane, so that a count may begin while the tail of the one before still runs),
 * "reads_per_look" (0 = automatic: eight one-block reads per lane and table look where the count kernel is held to five
 * workgroups a compute unit and the pools are compact, else four; 4; 8), "workgroups_per_cu" (resident workgroups of the count kernel per compute unit: 0 = as many as fit, default -1 = five
 * when the EM runs its one-lane-per-event kernel beside it and the read set is evenly deep, else as many as fit), "em_flat_min_events" (default 16 384: with at
 * least that many two-isoform events the ones that converged within 32 iterations last time are solved one lane per
 * event instead of four -- fewer instructions, longer passes), "em_closed_form" (default 0; 1: two-isoform events with one read
 * file run six ordinary EM iterations and finish in the closed form of their EM map -- the step of read.h:592-618 is then a
 * Moebius map of theta_0, theta after m more iterations one exponential away, and the iteration at which read.h:659 stops
 * is found by search: the same iteration counts, theta within 1e-13; pays where the slowest events take hundreds of
 * iterations and little else runs beside the EM, e.g. a rank of an event-sharded job).  LSQ_E_ARG for an unknown name.  The executables
 * pass LSQ_OPTIONS="name=value,..." from the environment through this call. */
int lsq_ctx_set_option(lsq_ctx *c, const char *name, double value);

/* Replaces solve/solve.cpp:796-806,823-826 (common/read.h:592-660): batched EM over the
 * class counts, one event per lane, fp64.  Needs lsq_count first.  Asynchronous. */
int lsq_solve(lsq_ctx *c);

/* Result fetch (synchronises the stream).  Arrays are in output order; class c (1 <= c <
 * 2^K) of an event is the set of isoforms j with bit j set, and its slot is
 * class_off[ev] + c - 1 where class_off is the exclusive prefix sum of (2^K - 1).
 *   class_count[m * n_classes + slot], class_bases[...]: uint64
 *   theta[iso_off[ev] + j], logll[ev], em_iters[ev], em_flags[ev]
 * em_flags bit 0: the stop criterion |1 - old_ll/ll| came within the guard band (1e-11 unless
 * lsq_set_em_guard_band changed it) of the 1e-6 threshold at some iteration, so the kernel's sums over
 * compatibility classes could stop an iteration apart from the reference's sums over reads.  Such an event
 * is solved again in the reference's own order -- its valid reads in index order (count/count.cpp:64-85),
 * per-read sums as common/read.h:592-660 forms them, IEEE fp64, libm's log, on the host -- by
 * lsq_solve_finalize, which lsq_results_solve runs first; bit 2 (value 4) marks an event whose numbers
 * come from that replay.  bit 1: the iteration cap was reached. */
int64_t lsq_results_num_classes(const lsq_ctx *c);
int lsq_results_class_offsets(const lsq_ctx *c, uint64_t *class_off /* n_events+1 */);
int lsq_results_counts(lsq_ctx *c, uint64_t *class_count, uint64_t *class_bases);
/* The inverse: class counts and matched bases in that layout become the context's counts (e.g. the
 * sums over several processes that each counted a slice of the reads); lsq_solve then runs on them. */
int lsq_results_set_counts(lsq_ctx *c, const uint64_t *class_count, const uint64_t *class_bases);
/* The same exchange without the host: lsq_counts_export_device copies the latest count's class counts and matched bases
 * as they lie on the device -- lsq_counts_device_words() 8-byte words, the same order on every context that holds the
 * same events -- into a device buffer, on the result stream; lsq_counts_import_device takes such a buffer (the sum over
 * the ranks of a read-sharded job, count/count.cpp:378,467-482 add reads in any order) as the counts lsq_solve and the
 * getters work on.  The caller orders its collective against the result stream (lsq_ctx_result_stream). */
uint64_t lsq_counts_device_words(const lsq_ctx *c);
int lsq_counts_export_device(lsq_ctx *c, void *d_words);
int lsq_counts_import_device(lsq_ctx *c, const void *d_words);
int lsq_results_solve(lsq_ctx *c, double *theta, double *logll, uint32_t *em_iters, uint8_t *em_flags);
/* The exact-order replay on its own, for callers that take the results through lsq_results_copy_device:
 * waits for the latest lsq_solve, redoes every flagged event as described above and writes theta, logll,
 * iteration count and flag bit 2 back to the device arrays; *n_replayed (may be NULL) = events redone.
 * Events of counts that came from lsq_results_set_counts are left as they are (the reads behind such sums
 * are not all on this device).  common/read.h:592-660, solve/solve.cpp:720-748,767-806. */
int lsq_solve_finalize(lsq_ctx *c, uint32_t *n_replayed);
/* Width of the band around the 1e-6 stop threshold (common/read.h:659) inside which an event is flagged
 * and replayed; default 1e-11, i.e. ~100x the difference between the two summation orders.  A wider band
 * replays more events (1.0: every event with an EM loop), never changes which result is right. */
int lsq_set_em_guard_band(lsq_ctx *c, double band);

/* Optional: the expected Fisher information of theta_1..theta_{K-1} per read at the solved theta,
 * and the two variance estimates made from it -- common/fim.h:115-158,320-367 (bruteforce_fim / ofim:
 * the sum over every accessible read start of every isoform), :58-93 (estimate_mle_variance_by_diag,
 * _by_inv) with common/linalg.h:28-71's inverse.  PARITY UNPINNED: the reference never includes these
 * headers, no reference binary prints these numbers; the tests check the HIP path against the oracle's
 * restatement of the headers.  One matrix per event and read file, (K-1) x (K-1), row-major; events
 * in output order at lsq_results_fim_offsets (n_events + 1 values; K = 1: empty matrix, variances 0).
 * Call after lsq_solve.  fim: [method][lsq_results_fim_size], variances: [method][n_events]. */
int lsq_fim(lsq_ctx *c);
int64_t lsq_results_fim_size(const lsq_ctx *c);
int lsq_results_fim_offsets(const lsq_ctx *c, uint64_t *fim_off /* n_events+1 */);
int lsq_results_fim(lsq_ctx *c, double *fim, double *var_by_diag, double *var_by_inverse);

/* Copies the raw device-order results into caller-provided DEVICE buffers (e.g. tensors of a
 * framework that will run a collective on them), asynchronously on the context's RESULT stream (the
 * one the EM runs on, not lsq_ctx_stream's: lsq_ctx_synchronize waits for all of them), behind the count's
 * exception pass -- and its recount, should the exception list have overflowed -- and the EM, so the
 * tables are complete.  theta / logll are the kernel's numbers; an event inside the EM guard band
 * (rare; em_flags bit 0) gets the reference's exact-order numbers from lsq_solve_finalize.
 * d_class_count [n_methods * n_classes] uint64, d_theta [n_isoforms] f64, d_logll [n_events]
 * f64; any may be NULL.  lsq_results_device_order() gives, per device-order event, its output
 * index, so a gathered buffer can be put in output order on the receiving side. */
int lsq_results_copy_device(lsq_ctx *c, void *d_class_count, void *d_theta, void *d_logll);
/* The stream those hand-offs run on (hipStream_t in a void*): a caller that launches its own work on the
 * handed-over buffers -- a collective, say -- orders it behind an event recorded here.  It is the result stream of the
 * latest lsq_count's lane, so it changes with every lsq_count (see lsq_ctx_stream above): query it per step. */
void *lsq_ctx_result_stream(lsq_ctx *c);

/* ---- one job over several GPUs: events sharded by index, per-event records gathered -------------------
 * The reference's scale-out unit is a slice gene_begin_idx..gene_end_idx of the sorted gene list, one process
 * per slice, outputs concatenated (count/count.cpp:204-215).  Here: every process compiles the WHOLE selected
 * range (so the load-time read filter is the unsharded one), takes its slice with lsq_events_set_shard --
 * lsq_shard_bounds cuts the list into `world` contiguous slices of equal weight, e.g. reads per event from a
 * first unsharded count -- counts and solves it, packs its per-event records in output order
 * (lsq_results_pack_device: lsq_record_words(e, first, count) words of 8 bytes -- class counts
 * [method][class], matched bases [method][class], theta, log-likelihood -- asynchronously on the result
 * stream), the blocks are all-gathered (RCCL: ncclAllGather of blocks padded to the longest; liblesseq_rccl's
 * lsq_gather does that for a C host, torch.distributed for lesseq_amd/dist.py and bench.py) and
 * lsq_gathered_unpack lays them out as the whole job's tables for lsq_format_count / lsq_format_solve. */
int lsq_shard_bounds(const lsq_events *e, int world, const double *weights /* per event, or NULL */,
                     uint64_t *first /* world */, uint64_t *count /* world */);
uint64_t lsq_record_words(const lsq_events *e, uint64_t first, uint64_t count);
int lsq_results_pack_device(lsq_ctx *c, void *d_block);
/* Device buffers for a host without HIP of its own (zeroed; lsq_device_read waits for all streams of the context). */
int lsq_device_alloc(lsq_ctx *c, uint64_t bytes, void **out);
void lsq_device_free(lsq_ctx *c, void *p);
int lsq_device_read(lsq_ctx *c, void *host_dst, const void *device_src, uint64_t bytes);
int lsq_device_write(lsq_ctx *c, void *device_dst, const void *host_src, uint64_t bytes);      /* blocking, after the context's streams */
int lsq_gathered_unpack(const lsq_events *e, int world, const uint64_t *first, const uint64_t *count,
                        const uint64_t *blocks, uint64_t stride_words,
                        uint64_t *class_count, uint64_t *class_bases, double *theta, double *logll);
int lsq_results_device_order(const lsq_ctx *c, int32_t *dev2out /* n_events */);

/* Device timing, for bench.py and the developer tools.  Off by default: the event records between
 * the kernels of a step hold the queue up (measured: 0.343 -> 0.317 ms per count+solve step on the
 * 100 M-read workload without them).  lsq_set_timing(ctx, 1) makes the following lsq_count /
 * lsq_solve calls record HIP events on the context stream around their launches; the getters
 * return LSQ_E_STATE for a call that ran without. */
int lsq_set_timing(lsq_ctx *c, int on);
/* Duration of the last lsq_count / lsq_solve (ms), kernels only. */
int lsq_last_timing(lsq_ctx *c, float *count_ms, float *solve_ms);
/* Duration of the last lsq_count's lsq_count_fast_kernel launches alone (summed over read files),
 * from HIP events recorded immediately around each launch. */
int lsq_last_fast_kernel_ms(lsq_ctx *c, float *ms);

/* ------------------------------------------------------------------------------------
 * Output rows (host): replaces count/count.cpp:486-492 and solve/solve.cpp:808-847
 * ------------------------------------------------------------------------------------ */

/* Formats the count table from fetched class counts into a malloc'd NUL-terminated buffer
 * (free with lsq_free). */
int lsq_format_count(const lsq_events *e, int n_methods, const uint64_t *class_count, char **out_text);
int lsq_format_solve(const lsq_events *e, int n_methods, const uint64_t *class_count,
                     const uint64_t *class_bases, const double *theta, const double *logll,
                     const double *total_read_bases, char **out_text);
void lsq_free(void *p);

/* Whole executables in-process: argv as the reference's (argv[0] ignored).  tool is
 * "count", "solve" or "classify".  stdout text is returned in *out_text (malloc'd), the
 * return value is the process exit status the reference would give (0, 1). */
int lsq_cli_run(const char *tool, int argc, const char *const *argv, char **out_text);
/* The same for an executable's main(): writes the table to stdout itself and returns the exit status.  A successful
 * count / solve run leaves the process (_exit) as soon as its table is written and flushed -- the pools, the helper
 * threads and the runtime are the operating system's to reclaim, which takes it a fraction of the time an orderly
 * teardown of several gigabytes of HBM takes (0.1-0.2 s of a 0.7 s run); LSQ_CLI_TEARDOWN=1 in the environment keeps
 * the orderly way.  Not for use inside a host process: call lsq_cli_run there. */
int lsq_cli_main(const char *tool, int argc, const char *const *argv);

/* ------------------------------------------------------------------------------------
 * Synthetic workload; deterministic in (seed, sizes).  Host-only.
 * ------------------------------------------------------------------------------------ */
typedef struct {
	uint64_t seed;
	uint64_t n_events;
	uint64_t n_reads;
	uint32_t read_length;
	uint32_t n_chrom;          /* chr1..chrN */
	uint32_t event_types;      /* bit t set: type t allowed (SE RI A5SS A3SS MXE AFE ALE T3) */
	uint32_t zipf;             /* 0: uniform depth; 1: Zipf(1.1) depth over events */
	double overlap_frac;       /* fraction of events that overlap their left neighbour */
	uint64_t first_read;       /* reads are numbered first_read .. first_read+n_reads-1 in the spec's
	                              read stream (counter-based), so chunks of one stream can be made */
	uint32_t sorted;           /* 1: the reads in coordinate order (chromosome, first base, then their number in the stream) -- what an
	                              aligner's sorted output looks like -- instead of the stream's own (shuffled) order; line numbers follow the file */
	uint32_t reserved;
} lsq_synth_spec;

/* Writes <stem>.interval, <stem>.map and (if write_mrf) <stem>.mrf under dir. */
int lsq_synth_write(const lsq_synth_spec *s, const char *dir, const char *stem, int write_mrf);
/* Generates the reads directly as a read set against already compiled events made from
 * lsq_synth_write's annotation with the same spec (no text round trip). */
int lsq_synth_reads(const lsq_synth_spec *s, lsq_events *e, int n_threads, lsq_reads **out);

#ifdef __cplusplus
}
#endif
#endif
