// Shared by the HIP translation units of the device group (not part of the ABI): device buffers, the
// per-read-file state, the context, and the entry points one unit offers the others.
//   lsq_device.hip  context, event tables, result fetch         lsq_count.hip  count kernels
//   lsq_ingest.hip  loader kernels (MRF parse, filter, pools)    lsq_em.hip     EM kernel
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "lsq_internal.hpp"

using namespace lsq;

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t _e = (expr);                                                                \
		if (_e != hipSuccess) return fail(LSQ_E_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
	} while (0)


#ifndef LSQ_P1_PAD
#define LSQ_P1_PAD 8
#endif
#ifndef LSQ_P2_PAD
#define LSQ_P2_PAD 4
#endif
constexpr unsigned P2_GROUP_PAD = LSQ_P2_PAD;      // ... and a junction group of the two-block pool (eight, with eight two-block reads per look, measured 2 % slower on C3: more padding, longer steps)
constexpr unsigned P1_GROUP_PAD = LSQ_P1_PAD;      // records a cell's group of the one-block pool is padded to: what a lane of the count kernel takes per look
constexpr int LSQ_INGEST_STAGES = 7;     // newline count, route, partition count, partition scatter, group classify, group offsets, group place
constexpr int EM_LANES = 4;          // lanes that share one event in the EM kernel (and one place of its grid)

namespace lsq {

// Compact pool records.  A block is (offset | length << 22): 22 bits of offset, 10 bits of length.  The offset of a read's
// first block counts from the bucket's first base minus COMPACT_BIAS (the first bin of a bucket also holds reads that start
// before it), that of its second block from the end of the first (the bases between them).  One-block reads take one
// such word, two-block reads two.  Reads with a longer block or a larger offset are kept with the many-block reads.
constexpr unsigned COMPACT_OFF_BITS = 22, COMPACT_OFF_MASK = (1u << COMPACT_OFF_BITS) - 1u, COMPACT_MAX_LEN = 1u << 10;
constexpr int COMPACT_BIAS = 1 << 21;
__host__ __device__ inline bool compact_block_fits(long long off, long long len) { return off >= 0 && off <= (long long)COMPACT_OFF_MASK && len > 0 && len < (long long)COMPACT_MAX_LEN; }

// A (read, event) pair the fast kernel does not settle itself: span-start ties that need the
// strand/name order, two-block reads whose blocks touch, second looks that did not fit the LDS
// queue.  pool 0 = one-block pool, 1 = two-block pool; scan: continue with the following events.
struct ExcEntry {
	unsigned long long slot;       // index into the pool
	unsigned bucket;
	unsigned ev_pool_scan;         // event index in the bucket | pool << 29 (0 one block, 1 two blocks, 2 n blocks) | scan << 31
};

template <class T>
struct DevBuf {
	T *p = nullptr;
	size_t n = 0;
	~DevBuf() { if (p) (void)hipFree(p); }
	int alloc(size_t count) {
		if (p) { (void)hipFree(p); p = nullptr; }
		n = count;
		HIP_TRY(hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)));
		return LSQ_OK;
	}
	int upload(const T *src, size_t count, hipStream_t st) {
		int rc = alloc(count);
		if (rc) return rc;
		if (count) HIP_TRY(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, st));
		return LSQ_OK;
	}
};

template <class T>
struct DevView { T *p = nullptr; size_t n = 0; };     // a slice of somebody else's allocation

// what the loader's routing pass knows about one chromosome: its stretch of the locator grid, of the covered regions and of the clusters
struct RouteChrom {
	unsigned loc_first, loc_nb;              // its bins of the grid
	int loc_base;                            // first base of bin 0
	unsigned cov0, cov1, clu0, clu1;
	unsigned pad;
};
static_assert(sizeof(RouteChrom) == 32, "RouteChrom is staged in LDS as two 16-byte words");

// What a workgroup of the count kernel needs to know about one visit to a bucket, in one 128-byte record per (read file,
// bucket), written once per read set (ingest): the bucket's description, where its slots and its one- and two-block
// records lie, and the next packed bucket that holds any.  A wave reads it with scalar loads -- one memory round trip
// where the kernel used to chase the description, three offset arrays and the bucket search one after the other.
struct VisitRec {
	BucketDesc d;
	unsigned long long bs, be;               // the bucket's slots
	unsigned long long p1o, p1n, p2o, p2n;   // first record and number of records of its one- and two-block pools
	unsigned b, next;                        // this bucket; the next packed bucket with slots (n_buckets: none)
	unsigned pad[2];
};
static_assert(sizeof(VisitRec) == 128, "VisitRec is read as 32 dwords");
// A workgroup's share of a count launch with the visit record of its first bucket beside it: one scalar load at the start of the
// workgroup where the share's bounds, its first bucket and that bucket's record were three, two of them one after the other.
struct WgPlan {
	unsigned long long s_begin, s_end;       // the share, in slots
	VisitRec first;                          // (b == n_buckets: the share holds no packed bucket's slots)
};
static_assert(sizeof(WgPlan) == 144, "WgPlan layout");

struct MethodReads {
	DevBuf<ExcEntry> exc;                  // exception lists: two halves of exc_cap entries, one per counter set
	size_t exc_cap = 0;
	bool present = false;
	uint64_t n_retained = 0, n_retained_blocks = 0, total_slots = 0;
	// one- and two-block pools: wide records (2 / 4 ints a read), or compact ones (1 / 2 ints a read, see COMPACT_*) when
	// at least 15 of 16 such reads fit them; compact pools are padded to whole 16-byte words
	DevBuf<int32_t> p1, p2, pn_se;
	bool compact = false;
	uint64_t n1_reads = 0, n2_reads = 0;    // one- and two-block reads in the pools (their slots also hold the groups' padding)
	DevBuf<uint8_t> p1_strand, p2_strand, pn_strand;
	DevBuf<uint32_t> p1_line, p2_line, pn_line, pn_blk_off, pn_nblk, pn_bucket;
	DevBuf<unsigned long long> p1_off, p2_off, pn_off, pnb_off, slot_off;
	double skew = 1.0;                      // reads of the fullest bucket / mean reads per bucket
	bool named = false;                     // the reads carry their own names (the *_line arrays index name_off)
	DevBuf<char> names;
	DevBuf<unsigned long long> name_off;
	DevBuf<unsigned> wg_first;             // per workgroup of the fast kernel's grid (`wg_grid` of them): the bucket its share starts in
	DevBuf<unsigned long long> wg_cut;     // ... and the shares' bounds in slots (wg_grid + 1 values)
	DevBuf<WgPlan> wg_plan;                // the same per workgroup, with its first visit record: what the kernel reads
	std::vector<VisitRec> visits_host;     // (for the plan)
	std::vector<unsigned long long> slot_off_host;   // the buckets' slot offsets (n_buckets + 1), for the share plan
	DevBuf<VisitRec> visits;               // per bucket (n_buckets + 1: the last one ends every chain)
	std::vector<unsigned long long> plan_n1, plan_n2;   // per bucket: records of the one- / two-block pool (padding included) ...
	// ... the same by cell and junction group, as stretches of slots (plan_share_cuts_seg): x[0 .. S] bounds, kind (0 one-block records, 1 two-block, 2 the rest),
	// the walk's looks at the stretch's reads, and per bucket its first stretch
	std::vector<unsigned long long> plan_seg_x;
	std::vector<unsigned char> plan_seg_kind;
	std::vector<unsigned> plan_seg_looks, plan_seg_first;
	std::vector<unsigned> plan_park1, plan_park2;       // ... and the looks of the general walk at the reads of each that the streaming loops leave to it (the ingest's estimate)
	std::vector<unsigned> next_packed_host;          // per bucket b: the first packed bucket >= b that holds slots (n_buckets: none), for the share plan
	unsigned long long wg_grid = 0;
};

} // namespace lsq

struct lsq_ctx {
	// lsq_ctx_create_with(LSQ_CTX_LANES_IN_BACKGROUND): the helper thread that makes the lanes' streams and events, and how it ended
	std::thread lanes_thread;
	int lanes_status = 0;
	std::string lanes_error;
	int device = 0;
	int n_cu = 256;
	hipStream_t stream = nullptr;           // uploads, ingest and the count kernels
	// The EM and the hand-off of results run on a stream of their own: the EM ends in a long tail
	// of a few slow events on an otherwise idle device, and the next lsq_count may run beside it.
	// So do the zeroing of the counters and the exception pass of a count (lsq_count_cleanup_kernel).
	// For that the counters and the exception lists exist twice (a count writes the set the solve
	// before last read); ev_counted: end of the latest count's streaming kernels; ev_mark: recorded
	// on the result stream when a count is submitted, behind the zeroing of the set the NEXT count
	// writes -- that count waits for it, i.e. for the readers of its set and not for the EM in between.
	// ... and all of that exists twice, as two LANES that consecutive counts take in turn: a counter set, exception
	// lists, a result stream, the EM's output arrays and the two events.  The EM of step k (a chain of dependent passes on a
	// handful of waves) then runs beside the exception pass and the EM of step k+1 instead of ahead of them: with small
	// shards (a rank of an event-sharded job) the result stream's chain, not the count kernel, bounded the step.
	hipStream_t stream_em = nullptr;        // the latest count's lane (= stream_em2[flip])
	hipStream_t stream_em2[2] = {nullptr, nullptr};
	// A lane's counts run on a stream of the lane's own: consecutive counts write different counter sets and share only
	// what they read, so the next count's workgroups may fill the compute units the tail of this one leaves idle, and no
	// dispatch waits for the kernel before it (option "count_streams" 1: every count on `stream`, one after the other).
	hipStream_t stream_count2[2] = {nullptr, nullptr};
	bool opt_two_count_streams = true;
	int opt_wg_per_cu = -1;                 // "workgroups_per_cu": resident workgroups of the count kernel per compute unit (lsq_count.hip)
	hipEvent_t ev_counted2[2] = {nullptr, nullptr}, ev_mark2[2] = {nullptr, nullptr};
	bool mark_recorded2[2] = {false, false};
	int flip = 0;                           // counter set of the latest count
	size_t counters_per_set = 0;
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
	hipEvent_t evt0 = nullptr, evt1 = nullptr;      // around a text copy (lsq_text_stage may run beside other host work)
	hipEvent_t evf0[LSQ_MAX_METHODS] = {}, evf1[LSQ_MAX_METHODS] = {};   // around each method's lsq_count_fast_kernel launch
	int fast_launched = 0;
	int occ_p1w = 0;
	int opt_reads_per_look = 0;             // "reads_per_look": 0 = by the launch (lsq_count.hip), 4, 8
	unsigned last_reads_per_look = 4, last_wg_per_cu = 0;       // what the latest lsq_count ran with
	unsigned occ_lds_bytes = 0; int occ_blocks = 0;      // the runtime's occupancy answer for the fast kernel at that LDS size
	bool time_events = false;               // lsq_set_timing: event records around the kernels cost ~4 us each in the queue
	bool count_timed = false, solve_timed = false;      // the last lsq_count / lsq_solve ran with them
	lsq_events *E = nullptr;                // must outlive the uploads made from it (its strand dictionary grows with the reads)
	DevBuf<BucketDesc> buckets;
	DevBuf<uint8_t> images, strand_rank, dK;
	DevBuf<TieRec> ties;
	DevBuf<uint32_t> cls_base, iso_base, em_order, gene_name_off;
	DevBuf<char> gene_names;                // device event order
	DevBuf<uint32_t> pack_cls, pack_iso, pack_ev;      // lsq_results_pack_device: device index of every class / isoform / event of the shard, in output order
	// Regrouping (option "em_regroup", on by default): a wave of the lean EM kernel runs until the slowest of its sixteen
	// events has converged, so events that take about as many iterations should share waves.  After a solve, a small kernel
	// on the same lane sorts the lean group's places by the iteration counts that solve just wrote; the lane's next solve
	// (two steps later) and the fifteen after it use that order.  A prediction, nothing more: an event's numbers do not depend on its wave mates.
	DevBuf<uint32_t> em_order_lane[2];
	DevBuf<uint32_t> em_split;             // per lane the first place of the one-lane-per-event kernel (lsq_em.hip), then a word that says "none"
	bool em_order_lane_valid[2] = {false, false};
	unsigned em_regroup_age[2] = {0, 0};    // solves since the lane's order was last refreshed (every 16th solve refreshes it)
	bool opt_em_regroup = true;
	// "em_closed_form" (default off): two-isoform events with one read file run a few ordinary iterations and finish in the
	// closed form of their EM map (lsq_em.hip: lsq_em_head_kernel / lsq_em_tail_kernel); the tail's list per step lane
	bool opt_em_closed = false;
	// "em_quad_cap" (0: off): the four-lane lean kernel hands events that still run after that many iterations to lsq_em_tail_kernel
	// (two-isoform events with one read file, where no placement by earlier iteration counts is in use: lsq_em.hip)
	unsigned opt_em_quad_cap = 0;
	DevBuf<uint32_t> em_tail_count, em_tail_u32[2];
	DevBuf<uint8_t> em_tail_flag[2];
	DevBuf<double> em_tail_f64[2];
	unsigned opt_em_flat_min = 16384;      // "em_flat_min_events": lean events from which on the fast ones run one lane per event
	unsigned em_places = 0;
	unsigned em_small_places = 0;           // the first of them: events of the lean EM kernel
	DevBuf<double> G, theta2[2], logll2[2];
	DevBuf<uint32_t> iters2[2];
	DevBuf<uint8_t> flags2[2];
	DevView<double> theta, logll;           // the latest count's lane
	DevView<uint32_t> iters;
	DevView<uint8_t> flags;
	// lsq_fim (fim.h / linalg.h): per device event the offset of its accessible-start class counts and of its
	// (K-1) x (K-1) matrix; counts per method; results
	DevBuf<uint32_t> fim_start_base, fim_mat_base, fim_starts;
	DevBuf<double> fim, fim_var;
	size_t fim_starts_total = 0, fim_mat_total = 0;
	bool fim_uploaded = false, fim_done = false;
	DevBuf<unsigned long long> counters;   // two sets of cnt | bases | exc_count | dbg (one memset per count); the views below are the latest count's
	DevView<unsigned long long> cnt, bases, dbg;
	DevBuf<unsigned long long> wg_trace;   // developer build (LSQ_ABLATE 4194304): lsq_debug_wg_trace
	unsigned long long wg_trace_n = 0, wg_trace_workers = 0;
	DevView<unsigned> exc_count;           // per method: [2m] appended, [2m+1] overflow flag
	// ingest tables (round 4 form): per chromosome id a RouteChrom; the covered regions as (start, end) pairs; the clusters (spans of
	// the planned events) as (start, end, bucket, bucket's first base), cut at the bucket cuts; and the locator grid over both:
	// per chromosome bins of 2^loc_shift bases, entry k = lower bounds of the bin's first base among the covered starts (.x) and
	// the cluster starts (.y); entries k and k + 1, one 16-byte load, bound a search to a handful of neighbouring records
	DevBuf<RouteChrom> route_chrom;
	DevBuf<int2> cov;
	DevBuf<int4> clu;
	DevBuf<uint2> loc;
	unsigned loc_shift = 10;
	DevBuf<unsigned> bin_base;             // per bucket: first of its bins among all bins (n_buckets + 1)
	size_t n_fine = 0;
	// The one-block pool is laid out by cell (lsq_internal.hpp: Cell), not by bin: per bucket its cells in order and one more
	// group for the reads that start in no cell, every group padded to eight records (P1_GROUP_PAD) -- a lane of the count kernel takes four or eight
	// neighbouring records and decides them against one cell.  cell_base: per bucket the first of its groups (n_buckets + 1).
	DevBuf<unsigned> cell_base;
	size_t n_cell_groups = 0;
	// The two-block pool is laid out by junction group (lsq_events::jg_keys), every group padded to four records, and per
	// bucket one more group for the reads that cross no known junction.  jgroup_base: per bucket the first of its groups.
	DevBuf<unsigned long long> jg_keys;
	DevBuf<unsigned> jg_base, jgroup_base;
	size_t n_junction_groups = 0;
	unsigned n_chrom_tables = 0;
	// lsq_ctx_set_option: grid multiplier (0 = by the read set's skew), entries of a method's exception list
	// (0 = a quarter of its reads, at least 64 Ki), recount every read with the one-lane-per-read kernel (self-check)
	double opt_grid_mult = 0;               // "grid_multiplier" (fractions allowed)
	size_t opt_exc_cap = 0;
	bool opt_compact_pools = true;          // "compact_pools": 0 keeps wide pool records whatever the reads look like
	bool opt_recount = false;
	unsigned opt_cleanup_grid = 0;          // "cleanup_workgroups": workgroups of the exception pass (0: one a compute unit)
	bool overflow_logged[LSQ_MAX_METHODS] = {};      // the warning about an overflowed exception list has been written for the latest count
	bool opt_snap_shares = true;            // workgroup shares cut on bucket boundaries where one is near
	bool opt_share_weighted = true;         // "share_weighted": the shares equal in cost, not in reads (run_count's plan)
	double opt_share_cost_p2 = 4.3;         // "share_cost_two_block": a two-block record, in one-block records
	double opt_share_cost_park = 9.0;       // "share_cost_parked": one look of the general walk at a parked read
	double opt_share_cost_visit = 7000.0;   // "share_cost_visit": staging + flush of a bucket
	double opt_share_cost_hot = 0.5;        // "share_cost_hot": extra cost a record of a group of 8 192 records or more
	double opt_share_taper = 0;             // "share_taper": the last share of the grid as a fraction of the first (0 = automatic)
	unsigned dev_ablate = 0;                // developer build only (LSQ_ABLATE)
	DevBuf<unsigned char> recount_args;     // the recount kernels' argument records (lsq_count.hip), and the host's copy of what was last written
	std::vector<unsigned char> recount_args_host;
	MethodReads reads[LSQ_MAX_METHODS];
	bool counted = false, solved = false;
	bool counts_external = false;           // lsq_results_set_counts: the counts are sums the reads here do not explain
	double em_band = 1E-11;                 // lsq_set_em_guard_band: events whose stop test comes this close to its threshold are replayed
	bool has_fast = false, has_generic = false, has_host = false;
	uint64_t host_genes = 0, host_reads = 0;     // host buckets of the latest count: their events, and the reads their clusters held
	std::map<size_t, std::vector<unsigned short>> host_seq;     // host buckets: [device event * M + method] -> classes of its valid reads, index order
	float count_ms = 0, solve_ms = 0;
	float mrf_h2d_ms = 0, mrf_parse_ms = 0;
	// how the latest device parse went (lsq_debug_last_parse_paths): tiles handed to the byte-walking kernel, lines handed to the shared
	// splitter, and whether the whole file went through the byte-walking kernel
	unsigned parse_tiles_handed = 0, parse_lines_listed = 0, parse_all_slow = 0;
	// two pinned 32 MiB host buffers and their "drained" events, made at the first large host-to-device copy (lsq_mrf_device.hpp: pinned_pipeline)
	unsigned char *pin_buf[2] = {nullptr, nullptr};
	hipEvent_t pin_ev[2] = {nullptr, nullptr};
	// device time and bytes of the stages of the latest ingest (lsq_ingest.hip: lsq_last_ingest_stages)
	hipEvent_t ing_ev[2 * LSQ_INGEST_STAGES] = {};
	float ing_ms[LSQ_INGEST_STAGES] = {};
	unsigned long long ing_bytes[LSQ_INGEST_STAGES] = {};
	bool ing_seen[LSQ_INGEST_STAGES] = {};
};

// MRF text of one file in HBM (lsq_text_stage).  Staging needs no event tables: the executables start it
// on a second thread while the first is still reading the annotation.
struct lsq_text {
	std::string path;
	unsigned long long offset = 0, len = 0;        // the bytes [offset, offset + len) of the file
	lsq::DevBuf<unsigned char> d_text;
	float h2d_ms = 0;
	bool scanned = false;                           // newlines counted (lsq_text_lines or the parse)
	unsigned long long n_nl = 0;
	lsq::DevBuf<unsigned long long> d_tile_base;    // per 8 KiB tile of the text: newlines ahead of it (n_tiles + 1)
};

namespace lsq {
int upload_strand_ranks(lsq_ctx *c);                 // lsq_device.hip
int run_count(lsq_ctx *c);                           // lsq_count.hip
int run_solve(lsq_ctx *c);
int run_fim(lsq_ctx *c);
int sync_all(lsq_ctx *c);                 // both streams
int ensure_lanes(lsq_ctx *c);             // lsq_device.hip: joins a context's helper thread (lanes made in the background), once
void note_overflow(lsq_ctx *c, const std::vector<unsigned> &exc_count);      // lsq_device.hip: the warning about an overflowed exception list, once per count
int host_count(lsq_ctx *c);                          // lsq_replay.hip: host buckets (genes beyond the kernels' limits)
int host_solve(lsq_ctx *c);
int replay_flagged(lsq_ctx *c, unsigned *n_done);    // lsq_replay.hip: the EM of guard-band events in the reference's per-read order
void select_counter_set(lsq_ctx *c, int set);                           // lsq_em.hip
} // namespace lsq
