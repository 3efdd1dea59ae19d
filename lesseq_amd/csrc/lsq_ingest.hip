// Loader kernels of the device group: MRF text parsed in HBM, the load-time containment filter, block
// merge and the bucket / group / pool layout (count/count.cpp:279-364), and the entry points around them.
//
// Round 4 form.  The loader is a chain of streaming passes, each of which reads and writes whole cache lines:
//   newline count   text -> newlines per 7 680-byte tile                               (lsq_mrf_device.hpp)
//   route           text (or parsed blocks from the host) -> per read a key (bucket, pool, strand) and its merged blocks:
//                   the splitter, the containment filter against the covered regions of the block's own chromosome
//                   (count/count.cpp:319, interval_list.hpp:396-422), the interval_list merge of the kept blocks (:323,
//                   interval_list.hpp:462-503), chromosome/strand of the last kept block (:321-322), the bucket of the first
//                   merged base; searches start from a locator grid (lsq_ctx::loc), one probe from their answer
//   partition       a counting sort of the routed reads by (pool, bucket): a histogram pass over the keys, a prefix sum,
//                   and a scatter in which a workgroup reserves its places per bucket with one atomic and writes runs
//   group           per (pool, bucket) partition, bucket tables staged in LDS as the count kernel stages them: the cell
//                   (one-block reads) or junction group (two-block reads) of every read, counted in LDS
//   place           prefix sums over the padded group sizes, then every read to its place in its group, written as the
//                   pool record it is to be (compact or wide); the scattered stores of a partition stay inside its
//                   own stretch of the pool -- a few hundred KB that the L2 holds until the lines are whole
// Until round 3 one kernel did filter, bucket and group per read in file order (four binary searches and the bucket's
// tables from HBM per read, two global atomics) and a second scattered 1-8-byte stores over the whole pool: 19 ms +
// 10 ms + 6 ms of one-workgroup scans per C3 file, >= 56 GB of HBM traffic for 3.7 GB of text.
// This replaces the reference's load-time filter and its read index (count/count.cpp:348-364).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <thread>

#include "lsq_device.hpp"
#include "lsq_mrf_line.hpp"

namespace {

// developer aid: LSQ_CLI_TIMING=1 prints host-side seconds of the loader's steps on stderr
struct HostStopwatch {
	bool on = getenv("LSQ_CLI_TIMING") != nullptr;
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	void mark(const char *what) {
		if (!on) return;
		const auto n = std::chrono::steady_clock::now();
		fprintf(stderr, "[timing]     %-32s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
		t = n;
	}
};

constexpr int INGEST_MAX_BLOCKS = 16;                  // merged blocks per read the device ingest handles
constexpr int LSQ_RETRY = 1;                           // a front end's settle(): route the file again (it has changed its own mode)

// ---- what the routing pass knows and what it leaves behind ------------------------------------------------------------
struct RouteTables {
	const RouteChrom *chrom;       // per chromosome id (lsq_device.hpp); a kernel may point this at its own copy in LDS
	const int2 *cov;               // covered regions: (start, end), ascending per chromosome
	const int4 *clu;               // clusters (spans of the planned events) cut at the bucket cuts: (start, end -- inclusive --, bucket, bucket's first base)
	const uint2 *loc;              // locator grid (lsq_ctx::loc): entries k and k + 1 are read as one 16-byte pair
	unsigned loc_shift;
	unsigned n_chrom;
};
constexpr unsigned ROUTE_CHROM_LDS = 64;      // chromosome records a kernel stages in LDS (more chromosomes: read from global memory)

// A read's key: pool in bits 0-1 (0 one merged block, 1 two, 2 three or more, 3 one or two that do not fit compact
// records), bucket in bits 2-23, strand id in bits 24-31.  Bucket 0x3FFFFF: not routed -- dropped (all ones), or
// retained by the filter but a candidate of no planned event (low bits 01, the number of its merged blocks in bits 24-31).
constexpr unsigned ROUTE_KEY_DROPPED = 0xFFFFFFFFu;
constexpr unsigned ROUTE_NO_BUCKET = 0x3FFFFFu;
__host__ __device__ inline unsigned route_key(unsigned bucket, unsigned pool, unsigned strand) { return (strand << 24) | (bucket << 2) | pool; }
__host__ __device__ inline unsigned route_key_unrouted(unsigned n_blocks) { return (n_blocks << 24) | (ROUTE_NO_BUCKET << 2) | 1u; }
__host__ __device__ inline bool route_key_is_routed(unsigned k) { return ((k >> 2) & ROUTE_NO_BUCKET) != ROUTE_NO_BUCKET; }

struct RouteOut {
	unsigned *key;                 // per read
	int4 *rec;                     // per read: its first two merged blocks (s0, e0, s1, e1)
	// reads of pools 2 and 3: a list (they are few in short-read files; a file of long reads fills it, and the ingest sizes it again)
	unsigned long long *nb_tot;    // [0] entries wanted, [1] blocks wanted, [2] error flag (a read beyond the tables' range)
	unsigned long long nb_cap, nbb_cap;
	uint4 *nb_ent;                 // read index, bucket, blocks | strand << 8, first block in nb_blk
	int2 *nb_blk;
	unsigned *cntn, *cntnb;        // per bucket: such reads, their blocks
	unsigned compact;              // compact pool records: one- and two-block reads that do not fit them go to pool 3
};

// The locator entry of base x on a chromosome: where, among the chromosome's covered regions and clusters, the records that
// start inside x's bin lie.  Kept per lane from one look-up to the next: a read's blocks and its first base mostly share a bin.
struct LocProbe {
	int chrom; long long bin;
	unsigned cov_a, cov_b, clu_a, clu_b;       // lower_bound(starts, x) lies in [a, b]
};
__device__ inline void loc_probe(const RouteTables &T, const RouteChrom &R, const int chrom, const int x, LocProbe &P) {
	const long long d = (long long)x - (long long)R.loc_base;
	long long k = d >> T.loc_shift;
	if (R.loc_nb == 0u || d <= 0) k = -1;                       // at or below the first bin's first base: nothing starts left of x
	else if (k >= (long long)R.loc_nb) k = (long long)R.loc_nb; // beyond the last bin: everything does
	if (P.chrom == chrom && P.bin == k) return;
	P.chrom = chrom; P.bin = k;
	if (k < 0) { P.cov_a = P.cov_b = R.cov0; P.clu_a = P.clu_b = R.clu0; }
	else if (k >= (long long)R.loc_nb) { P.cov_a = P.cov_b = R.cov1; P.clu_a = P.clu_b = R.clu1; }
	else {
		uint4 e;                                                  // (8-byte aligned: two entries in one load)
		__builtin_memcpy(&e, T.loc + (R.loc_first + (unsigned)k), 16);
		P.cov_a = e.x; P.clu_a = e.y; P.cov_b = e.z; P.clu_b = e.w;
	}
}

// interval_list::contains_interval against the covered regions of the block's chromosome (interval_list.hpp:396-422):
// lo = lower_bound(starts, start); the interval at lo (when it starts exactly there) or the one before it must reach `end`
__device__ inline bool route_covered(const RouteTables &T, const RouteChrom &R, const int chrom, const int start, const int end, LocProbe &P) {
	if (!(start < end)) return true;
	loc_probe(T, R, chrom, start, P);
	const unsigned a = P.cov_a, b = P.cov_b;
	int2 at, before;                        // the records at lo and at lo - 1
	bool has_at, has_before;
	if (b - a <= 3u) {
		// records a - 1 .. a + 3 hold both, wherever in [a, b] lo falls: five loads in flight at once, no dependent probe
		int2 c[5];
		unsigned below = 0;
#pragma unroll
		for (unsigned q = 0; q < 5; ++q) {
			const unsigned idx = a + q - 1u;
			const bool ok = idx + 1u > R.cov0 && idx < R.cov1 && idx <= b;     // (a - 1 may be R.cov0 - 1, or wrap below zero: both fail here)
			c[q] = ok ? T.cov[idx] : make_int2(0, 0);
			below += (unsigned)(ok && q >= 1u && idx < b && c[q].x < start);
		}
		const unsigned lo = a + below;
		at = make_int2(0, 0); before = make_int2(0, 0);
#pragma unroll
		for (unsigned q = 0; q < 5; ++q) { if (a + q - 1u == lo) at = c[q]; if (a + q == lo) before = c[q]; }
		has_at = lo < R.cov1; has_before = lo > R.cov0;
	} else {
		unsigned lo = a, hi = b;
		while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (T.cov[mid].x < start) lo = mid + 1; else hi = mid; }
		has_at = lo < R.cov1; has_before = lo > R.cov0;
		at = has_at ? T.cov[lo] : make_int2(0, 0);
		before = has_before ? T.cov[lo - 1u] : make_int2(0, 0);
	}
	if (has_at && at.x <= start && end <= at.y) return true;
	if (has_before && before.x <= start && end <= before.y) return true;
	return false;
}

// the cluster record of base p: the last one that starts at or left of p, if p is inside it -- its bucket is p's bucket
// (what the reference's candidate window comes to for a read's first base: count/count.cpp:429-432,463)
__device__ inline bool route_cluster(const RouteTables &T, const RouteChrom &R, const int chrom, const int p, LocProbe &P, int4 &rec) {
	if (p >= 0x7FFFFFFF) return false;
	loc_probe(T, R, chrom, p + 1, P);
	const unsigned a = P.clu_a, b = P.clu_b;       // upper_bound(starts, p) = lower_bound(starts, p + 1) lies in [a, b]
	if (b - a <= 3u) {
		int4 c[4];                                   // records a - 1 .. a + 2: the one before the upper bound is among them
		unsigned below = 0;
#pragma unroll
		for (unsigned q = 0; q < 4; ++q) {
			const unsigned idx = a + q - 1u;
			const bool ok = idx + 1u > R.clu0 && idx < R.clu1 && idx < b;      // (q = 0: a - 1 < b unless it wrapped, which the first test catches)
			c[q] = ok ? T.clu[idx] : make_int4(0, 0, 0, 0);
			below += (unsigned)(ok && q >= 1u && c[q].x <= p);
		}
		const unsigned ub = a + below;
		if (ub == R.clu0) return false;
		rec = make_int4(0, -1, 0, 0);
#pragma unroll
		for (unsigned q = 0; q < 4; ++q) if (a + q == ub) rec = c[q];
		return p <= rec.y;
	}
	unsigned lo = a, hi = b;
	while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (T.clu[mid].x <= p) lo = mid + 1; else hi = mid; }
	if (lo == R.clu0) return false;
	rec = T.clu[lo - 1u];
	return p <= rec.y;
}

// the chromosome records into a workgroup's LDS when they are few (every workgroup of the routing kernels starts with this)
__device__ inline const RouteChrom *route_stage_chroms(const RouteTables &T, RouteChrom *lds) {
	if (T.n_chrom > ROUTE_CHROM_LDS) return T.chrom;
	for (unsigned q = threadIdx.x; q < 2u * T.n_chrom; q += blockDim.x) reinterpret_cast<uint4 *>(lds)[q] = reinterpret_cast<const uint4 *>(T.chrom)[q];
	__syncthreads();
	return lds;
}

// interval_list::add_interval on a small sorted array (see lsq::IntervalList::add)
__device__ inline bool small_add_interval(int *s, int *e, int &n, int start, int end) {
	if (!(start < end)) return true;
	int ss = 0, se = 0, es = 0, ee = 0;
	for (int i = 0; i < n; ++i) { ss += s[i] < start; se += e[i] < start; es += s[i] < end; ee += e[i] < end; }
	const bool start_inside = (ss - se == 1), end_inside = (es - ee == 1);
	// starts: erase [ss, es), insert `start` at ss unless start_inside; ends: erase [se, ee), insert `end` at se unless end_inside
	const int ns = n - (es - ss) + (start_inside ? 0 : 1);
	if (ns > INGEST_MAX_BLOCKS) return false;
	int ts[INGEST_MAX_BLOCKS], te[INGEST_MAX_BLOCKS];
	int k = 0;
	for (int i = 0; i < ss; ++i) ts[k++] = s[i];
	if (!start_inside) ts[k++] = start;
	for (int i = es; i < n; ++i) ts[k++] = s[i];
	k = 0;
	for (int i = 0; i < se; ++i) te[k++] = e[i];
	if (!end_inside) te[k++] = end;
	for (int i = ee; i < n; ++i) te[k++] = e[i];
	n = ns;
	for (int i = 0; i < n; ++i) { s[i] = ts[i]; e[i] = te[i]; }
	return true;
}

__device__ inline void push3(int &a0, int &a1, int &a2, int &k, const int v) {
	a0 = k == 0 ? v : a0; a1 = k == 1 ? v : a1; a2 = k == 2 ? v : a2;
	++k;
}

// The kept blocks of one read as they come, merged by interval_list's rule.  Nearly every read keeps one or two merged
// blocks: those live in registers (the rule written out for a list of at most two); a third block moves the read to arrays.
// (the arrays are an object of their own: as members they kept the whole accumulator in the private segment -- every field a
// scratch store and load per block, 10 GB of scratch traffic per C3 file -- where now only a read's third block touches it)
struct ReadBig { int bs[INGEST_MAX_BLOCKS], be[INGEST_MAX_BLOCKS]; };
struct ReadAcc {
	int s0, e0, s1, e1;
	int n;                          // merged blocks
	int chrom;
	unsigned strand;
	bool any, ok, big;
	__device__ inline void init() { s0 = e0 = s1 = e1 = 0; n = 0; chrom = -1; strand = 0; any = false; ok = true; big = false; }
	// a block that passed the containment filter (count/count.cpp:319-323)
	__device__ inline void add(ReadBig &B, const unsigned c, const unsigned sid, const int start, const int end) {
		any = true; chrom = (int)c; strand = sid;
		if (!(start < end)) return;
		if (big) { ok = small_add_interval(B.bs, B.be, n, start, end) && ok; return; }
		if (n == 0) { s0 = start; e0 = end; n = 1; return; }
		const bool h0 = n > 0, h1 = n > 1;
		const int ss = (int)(h0 && s0 < start) + (int)(h1 && s1 < start), se = (int)(h0 && e0 < start) + (int)(h1 && e1 < start);
		const int es = (int)(h0 && s0 < end) + (int)(h1 && s1 < end), ee = (int)(h0 && e0 < end) + (int)(h1 && e1 < end);
		const bool start_inside = (ss - se == 1), end_inside = (es - ee == 1);
		int a0 = 0, a1 = 0, a2 = 0, ka = 0, b0 = 0, b1 = 0, b2 = 0, kb = 0;
		if (h0 && 0 < ss) push3(a0, a1, a2, ka, s0);
		if (h1 && 1 < ss) push3(a0, a1, a2, ka, s1);
		if (!start_inside) push3(a0, a1, a2, ka, start);
		if (h0 && 0 >= es) push3(a0, a1, a2, ka, s0);
		if (h1 && 1 >= es) push3(a0, a1, a2, ka, s1);
		if (h0 && 0 < se) push3(b0, b1, b2, kb, e0);
		if (h1 && 1 < se) push3(b0, b1, b2, kb, e1);
		if (!end_inside) push3(b0, b1, b2, kb, end);
		if (h0 && 0 >= ee) push3(b0, b1, b2, kb, e0);
		if (h1 && 1 >= ee) push3(b0, b1, b2, kb, e1);
		if (ka <= 2) { s0 = a0; s1 = a1; e0 = b0; e1 = b1; n = ka; }
		else { B.bs[0] = a0; B.bs[1] = a1; B.bs[2] = a2; B.be[0] = b0; B.be[1] = b1; B.be[2] = b2; n = 3; big = true; }
	}
	// the read is complete: its key and blocks to their place (index i of the pass)
	__device__ inline void finish(const ReadBig &B, const RouteTables &T, const RouteChrom *chroms, LocProbe &P, const RouteOut &O, const unsigned i) {
		unsigned key = ROUTE_KEY_DROPPED;
		int4 rec = make_int4(0, 0, 0, 0);
		if (any && n > 0) {
			long long tot = 0;
			if (big) {
				s0 = B.bs[0]; e0 = B.be[0];
				if (n > 1) { s1 = B.bs[1]; e1 = B.be[1]; }
				for (int q = 0; q < n; ++q) tot += B.be[q] - B.bs[q];
			} else tot = (long long)(e0 - s0) + (n > 1 ? (long long)(e1 - s1) : 0ll);
			if (!ok || tot >= (1 << 18)) atomicMax(&O.nb_tot[2], 1ull);
			key = route_key_unrouted((unsigned)n);
			rec = make_int4(s0, e0, n > 1 ? s1 : 0, n > 1 ? e1 : 0);
			const RouteChrom R = chroms[chrom];
			// the bucket of the first merged base, if that base lies in the span of some planned event (a cluster): otherwise the
			// read is a candidate of none of them (count/count.cpp:429-432,463) -- with a shard, the other shards' reads
			int4 cl;
			if (route_cluster(T, R, chrom, s0, P, cl)) {
				const unsigned b = (unsigned)cl.z;
				const int lo = cl.w;
				unsigned pool = n == 1 ? 0u : (n == 2 ? 1u : 2u);
				if (pool < 2u && O.compact) {
					bool fits = lsq::compact_block_fits((long long)s0 - lo + lsq::COMPACT_BIAS, (long long)e0 - s0);
					if (n == 2) fits = fits && lsq::compact_block_fits((long long)s1 - e0, (long long)e1 - s1);
					if (!fits) pool = 3u;
				}
				key = route_key(b, pool, strand);
				if (pool >= 2u) {
					const unsigned long long idx = atomicAdd(&O.nb_tot[0], 1ull), boff = atomicAdd(&O.nb_tot[1], (unsigned long long)n);
					atomicAdd(&O.cntn[b], 1u); atomicAdd(&O.cntnb[b], (unsigned)n);
					if (idx < O.nb_cap && boff + (unsigned)n <= O.nbb_cap) {
						O.nb_ent[idx] = make_uint4(i, b, (unsigned)n | (strand << 8), (unsigned)boff);
						if (big) { for (int q = 0; q < n; ++q) O.nb_blk[boff + q] = make_int2(B.bs[q], B.be[q]); }
						else { O.nb_blk[boff] = make_int2(s0, e0); if (n > 1) O.nb_blk[boff + 1] = make_int2(s1, e1); }
					}
				}
			}
		}
		O.key[i] = key;
		O.rec[i] = rec;
	}
};

// ---- device time of the chain's stages (lsq_last_ingest_stages): events around each stage's launches
struct StageClock {
	lsq_ctx *c; hipStream_t st; int s;
	StageClock(lsq_ctx *c_, hipStream_t st_, int s_) : c(c_), st(st_), s(s_) {
		for (int q = 0; q < 2; ++q) if (!c->ing_ev[2 * s + q]) (void)hipEventCreate(&c->ing_ev[2 * s + q]);
		if (c->ing_ev[2 * s]) (void)hipEventRecord(c->ing_ev[2 * s], st);
	}
	void end(unsigned long long bytes) {
		if (c->ing_ev[2 * s + 1]) (void)hipEventRecord(c->ing_ev[2 * s + 1], st);
		c->ing_bytes[s] = bytes; c->ing_seen[s] = true;
	}
};
static void stages_reset(lsq_ctx *c, bool keep_text_stage) {
	for (int s = keep_text_stage ? 1 : 0; s < LSQ_INGEST_STAGES; ++s) { c->ing_seen[s] = false; c->ing_ms[s] = 0; c->ing_bytes[s] = 0; }
}
static void stages_collect(lsq_ctx *c) {          // (the stream has been waited for)
	for (int s = 0; s < LSQ_INGEST_STAGES; ++s) {
		if (!c->ing_seen[s] || !c->ing_ev[2 * s] || !c->ing_ev[2 * s + 1]) continue;
		float ms = 0;
		if (hipEventElapsedTime(&ms, c->ing_ev[2 * s], c->ing_ev[2 * s + 1]) == hipSuccess) c->ing_ms[s] = ms;
	}
	(void)hipGetLastError();
}

} // namespace

#include "lsq_scan.hpp"

namespace {

#include "lsq_mrf_device.hpp"

// ---- routing of parsed blocks that came from the host (lsq_reads_upload: file order) -------------------------------------
struct IngestRaw {
	unsigned long long n_reads;
	const unsigned long long *blk_off;
	const unsigned *line_no;
	const int *blk_start, *blk_end;
	const unsigned short *blk_chrom;
	const unsigned char *blk_strand;
};

#ifndef LSQ_RAW_WAVES
#define LSQ_RAW_WAVES 8
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LSQ_RAW_WAVES))) lsq_route_raw_kernel(RouteTables T, IngestRaw R, RouteOut O) {
	__shared__ RouteChrom chrom_lds[ROUTE_CHROM_LDS];
	const RouteChrom *chroms = route_stage_chroms(T, chrom_lds);
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	LocProbe P;
	P.chrom = -1; P.bin = 0;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < R.n_reads; i += gsz) {
		const unsigned long long b0 = R.blk_off[i], b1 = R.blk_off[i + 1];
		ReadAcc A;
		ReadBig B;
		A.init();
		for (unsigned long long j = b0; j < b1; ++j) {
			const unsigned c = R.blk_chrom[j];
			if (c >= T.n_chrom) continue;
			const int bs = R.blk_start[j], be = R.blk_end[j];
			if (!route_covered(T, chroms[c], (int)c, bs, be, P)) continue;
			A.add(B, c, R.blk_strand[j], bs, be);
		}
		A.finish(B, T, chroms, P, O, (unsigned)i);
	}
}

// ---- partition: the routed reads of pools 0 and 1 sorted by (pool, bucket) -------------------------------------------------
// entry e of the partition tables: pool * n_buckets + bucket
constexpr unsigned PART_WG = 1024;

// pass 1: reads per entry; and the file's totals -- tot[0] retained reads, [1] their blocks (those of pools 2 and 3 are in
// nb_tot[1]), [3] reads of pool 3, [4] / [5] reads of pool 0 / 1
template <bool LDS>
__global__ void __launch_bounds__(PART_WG) lsq_part_hist_kernel(const unsigned *key, const unsigned long long n, const unsigned B, unsigned *part_cnt, unsigned long long *tot) {
	extern __shared__ unsigned part_lds[];
	__shared__ unsigned long long red[5][PART_WG / 64];
	if (LDS) { for (unsigned e = threadIdx.x; e < 2u * B; e += PART_WG) part_lds[e] = 0; __syncthreads(); }
	const unsigned long long c0 = n * blockIdx.x / gridDim.x, c1 = n * (blockIdx.x + 1ull) / gridDim.x;
	unsigned long long kept = 0, blocks = 0, misfit = 0, p1 = 0, p2 = 0;
	for (unsigned long long i = c0 + threadIdx.x; i < c1; i += PART_WG) {
		const unsigned k = key[i];
		if (k == ROUTE_KEY_DROPPED) continue;
		++kept;
		if (!route_key_is_routed(k)) { blocks += k >> 24; continue; }
		const unsigned pool = k & 3u, b = (k >> 2) & ROUTE_NO_BUCKET;
		if (pool >= 2u) { misfit += pool == 3u; continue; }
		if (pool == 0u) { ++p1; blocks += 1; } else { ++p2; blocks += 2; }
		if (LDS) atomicAdd(&part_lds[pool * B + b], 1u); else atomicAdd(&part_cnt[pool * B + b], 1u);
	}
	// the totals: lanes -> waves -> one atomic per workgroup and total
	unsigned long long v[5] = {kept, blocks, misfit, p1, p2};
#pragma unroll
	for (int q = 0; q < 5; ++q) {
		unsigned long long x = v[q];
		for (int d = 32; d > 0; d >>= 1) x += ((unsigned long long)(unsigned)__shfl_down((int)(unsigned)(x >> 32), d) << 32) + (unsigned)__shfl_down((int)(unsigned)x, d) ;
		if ((threadIdx.x & 63u) == 0) red[q][threadIdx.x >> 6] = x;
	}
	__syncthreads();
	if (threadIdx.x < 5) {
		unsigned long long x = 0;
		for (unsigned w = 0; w < PART_WG / 64; ++w) x += red[threadIdx.x][w];
		const int at[5] = {0, 1, 3, 4, 5};
		if (x) atomicAdd(&tot[at[threadIdx.x]], x);
	}
	if (LDS) for (unsigned e = threadIdx.x; e < 2u * B; e += PART_WG) { const unsigned c = part_lds[e]; if (c) atomicAdd(&part_cnt[e], c); }
}

struct PartArgs {
	const unsigned *key; const int4 *rec;
	unsigned long long n;
	unsigned B;
	const unsigned long long *off1, *off2;    // per bucket: first place of its partition of pool 0 / pool 1 (n_buckets + 1 each)
	unsigned *cursor;                          // per entry: places handed out
	uint4 *part1;                              // pool 0: s0, e0, line, strand
	uint4 *part2;                              // pool 1: two words a read: s0, e0, s1, e1 | line, strand, 0, 0
	const unsigned *line_no;                   // per read, or null: first_line + index
	unsigned long long first_line;
};

// pass 2: a workgroup takes a stretch of the reads: counts them per entry in LDS, reserves its places of every entry with one
// atomic, and writes its reads there -- runs of a stretch's reads per bucket, not single records
template <bool LDS>
__global__ void __launch_bounds__(PART_WG) lsq_part_scatter_kernel(PartArgs A) {
	extern __shared__ unsigned part_lds[];
	const unsigned B = A.B;
	const unsigned long long c0 = A.n * blockIdx.x / gridDim.x, c1 = A.n * (blockIdx.x + 1ull) / gridDim.x;
	if (LDS) {
		for (unsigned e = threadIdx.x; e < 2u * B; e += PART_WG) part_lds[e] = 0;
		__syncthreads();
		for (unsigned long long i = c0 + threadIdx.x; i < c1; i += PART_WG) {
			const unsigned k = A.key[i];
			if (!route_key_is_routed(k) || (k & 3u) >= 2u) continue;
			atomicAdd(&part_lds[(k & 3u) * B + ((k >> 2) & ROUTE_NO_BUCKET)], 1u);
		}
		__syncthreads();
		for (unsigned e = threadIdx.x; e < 2u * B; e += PART_WG) { const unsigned c = part_lds[e]; part_lds[e] = c ? atomicAdd(&A.cursor[e], c) : 0u; }
		__syncthreads();
	}
	// (four reads a lane at a time: their keys and blocks are on their way together, then their places, then their stores)
	for (unsigned long long i0 = c0 + threadIdx.x; i0 < c1; i0 += 4ull * PART_WG) {
		unsigned k[4]; int4 r[4]; unsigned line[4]; bool on[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const unsigned long long i = i0 + (unsigned long long)u * PART_WG;
			on[u] = i < c1;
			k[u] = on[u] ? A.key[i] : ROUTE_KEY_DROPPED;
			on[u] = on[u] && route_key_is_routed(k[u]) && (k[u] & 3u) < 2u;
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const unsigned long long i = i0 + (unsigned long long)u * PART_WG;
			r[u] = on[u] ? A.rec[i] : make_int4(0, 0, 0, 0);
			line[u] = on[u] ? (A.line_no ? A.line_no[i] : (unsigned)(A.first_line + i)) : 0u;
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			if (!on[u]) continue;
			const unsigned pool = k[u] & 3u, b = (k[u] >> 2) & ROUTE_NO_BUCKET, strand = k[u] >> 24;
			const unsigned at = LDS ? atomicAdd(&part_lds[pool * B + b], 1u) : atomicAdd(&A.cursor[pool * B + b], 1u);
			if (pool == 0u) A.part1[A.off1[b] + at] = make_uint4((unsigned)r[u].x, (unsigned)r[u].y, line[u], strand);
			else {
				const unsigned long long w = 2ull * (A.off2[b] + at);
				A.part2[w] = make_uint4((unsigned)r[u].x, (unsigned)r[u].y, (unsigned)r[u].z, (unsigned)r[u].w);
				A.part2[w + 1] = make_uint4(line[u], strand, 0u, 0u);
			}
		}
	}
}

// ---- pieces: a partition in stretches of at most PIECE reads, one workgroup of the two group passes each ------------------
constexpr unsigned PIECE = 16384;
__global__ void __launch_bounds__(256) lsq_piece_count_kernel(const unsigned *part_cnt, unsigned n_entries, unsigned *piece_cnt) {
	const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e < n_entries) piece_cnt[e] = (part_cnt[e] + PIECE - 1u) / PIECE;
}
// piece: entry, first read of the partition, reads, 1 when it is the partition's only piece
__global__ void __launch_bounds__(256) lsq_piece_expand_kernel(const unsigned *part_cnt, const unsigned long long *piece_off, unsigned n_entries, uint4 *pieces) {
	const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_entries) return;
	const unsigned n = part_cnt[e];
	const unsigned long long o = piece_off[e];
	const unsigned np = (n + PIECE - 1u) / PIECE;
	for (unsigned q = 0; q < np; ++q) pieces[o + q] = make_uint4(e, q * PIECE, min(PIECE, n - q * PIECE), np == 1u ? 1u : 0u);
}

struct GroupTables {
	const BucketDesc *buckets;
	const unsigned char *images;    // the buckets' LDS images (bin records, events and cells of packed buckets)
	const unsigned *cell_base;      // per bucket: first of its one-block groups (its cells, then "no cell")
	const unsigned long long *jg_keys; const unsigned *jg_base, *jgroup_base;      // junction groups of the two-block pool (lsq_events::jg_keys)
	unsigned B;
	unsigned lds_img, lds_keys;     // bytes of the kernel's LDS given to the image / the junction keys; the groups' counters follow
};

struct GroupArgs {
	const uint4 *pieces;
	const unsigned long long *off1, *off2;     // the partitions
	const uint4 *part1, *part2;
	unsigned *fine1, *fine2;                   // per read of a partition: its group (global index)
	unsigned *cnt1, *cnt2;                     // one-block reads per group [n_cell_groups]; two-block reads per group [n_junction_groups]
	unsigned *park1, *park2;                   // looks of the general walk at the group's reads that the count kernel's streaming loops will leave to it (an estimate, for the share plan)
};

// pass 1 over the partitions: the group of every read -- one-block reads: the cell of the first base (or the bucket's "no
// cell" group); two-block reads: the junction group (or, behind the junction groups, the group of the cell) -- found as the
// count kernel finds it (bin record: first cell | first event << 16, the ends of that cell and the next two; then on
// through the cell table), in the bucket's image staged in LDS; counted there
__global__ void __launch_bounds__(256) lsq_group_classify_kernel(GroupTables T, GroupArgs A) {
	extern __shared__ __align__(16) unsigned char glds[];
	const uint4 pc = A.pieces[blockIdx.x];
	const unsigned pool = pc.x >= T.B ? 1u : 0u, b = pc.x - pool * T.B;
	const BucketDesc d = T.buckets[b];
	const unsigned n_cells = d.kind == 1u ? (d.iso_off & 0xFFFFu) : 0u;
	const unsigned k0 = T.jg_base[b], k1 = T.jg_base[b + 1];
	const unsigned n_groups = pool ? (k1 - k0) + n_cells + 1u : n_cells + 1u;
	const unsigned gbase = pool ? T.jgroup_base[b] : T.cell_base[b];
	unsigned char *img = glds;
	unsigned long long *keys = reinterpret_cast<unsigned long long *>(glds + T.lds_img);
	unsigned *hist = reinterpret_cast<unsigned *>(glds + T.lds_img + T.lds_keys), *park = hist + n_groups;
	if (d.kind == 1u) for (unsigned q = threadIdx.x; q < d.img_bytes / 16u; q += 256u) reinterpret_cast<uint4 *>(img)[q] = reinterpret_cast<const uint4 *>(T.images + d.img_off)[q];
	if (pool) for (unsigned q = threadIdx.x; q < k1 - k0; q += 256u) keys[q] = T.jg_keys[k0 + q];
	for (unsigned q = threadIdx.x; q < 2u * n_groups; q += 256u) hist[q] = 0;
	__syncthreads();
	const uint4 *bins = reinterpret_cast<const uint4 *>(img);
	const lsq::Cell *cells = reinterpret_cast<const lsq::Cell *>(img + d.seg_off);
	const lsq::CellX *cellx = reinterpret_cast<const lsq::CellX *>(img + d.seg_off + 16u * (d.iso_off >> 16));
	const lsq::FastRec *recs = reinterpret_cast<const lsq::FastRec *>(img + d.ev_off);
	const unsigned long long first = (pool ? A.off2[b] : A.off1[b]) + pc.y;
	for (unsigned r = threadIdx.x; r < pc.z; r += 256u) {
		int s0, e0, s1 = 0;
		if (pool) { const uint4 v = A.part2[2ull * (first + r)]; s0 = (int)v.x; e0 = (int)v.y; s1 = (int)v.z; }
		else { const uint4 v = A.part1[first + r]; s0 = (int)v.x; e0 = (int)v.y; }
		unsigned cell = 0, g = pool ? (k1 - k0) : 0u, looks_est = 0;
		if (d.kind == 1u) {
			const int rel = s0 - d.lo;
			const unsigned bin = rel <= 0 ? 0u : min((unsigned)rel >> d.shift, d.n_bins - 1u);
			const uint4 br = bins[bin];
			const int p = s0;
			cell = (br.x & 0xFFFFu) + (unsigned)(p >= (int)br.y) + (unsigned)(p >= (int)br.z) + (unsigned)(p >= (int)br.w);
			if (p >= (int)br.w) while (cell + 1u < n_cells && p >= cells[cell + 1u].lo) ++cell;
			if (!(cell < n_cells && cells[cell].lo <= p && p < cells[cell].hi)) cell = n_cells;
			bool in_junction_group = false;
			if (pool) {
				// the junction group: block 1 ends on the end of the cell owner's segment (or of the segment that abuts it), block 2
				// starts where a later segment of that event does (the keys hold exactly those starts); else the group of the read's
				// cell -- `n_cells`: of no cell -- behind the junction groups
				unsigned jg = k1 - k0;
				if (cell < n_cells && (e0 == cells[cell].e1 || e0 == cells[cell].e2)) {
					const unsigned long long want = lsq::jg_key(cell, e0 == cells[cell].e1 ? 0u : 1u, s1);
					unsigned lo_k = 0, hi_k = k1 - k0;
					while (lo_k < hi_k) { const unsigned mid = (lo_k + hi_k) >> 1; if (keys[mid] < want) lo_k = mid + 1; else hi_k = mid; }
					if (lo_k < k1 - k0 && keys[lo_k] == want) jg = lo_k;
				}
				in_junction_group = jg < k1 - k0;
				g = in_junction_group ? jg : (k1 - k0) + cell;
			} else g = cell;
			// Will the streaming loop settle the read, or leave it to the general walk -- and how many events will the walk look
			// at for it?  What the parked reads cost beside the streamed ones is what makes buckets differ: the share plan weighs
			// it (run_count).  An estimate: the loops' rules in short; the walk's own stepping rule (fast_trip's return).
			bool parks = cell == n_cells, one_event = false;
			if (!parks && !in_junction_group) {
				const lsq::Cell cw = cells[cell];
				const unsigned fl = cellx[cell].flags;
				const int len = e0 - s0;
				const bool both = (fl & lsq::CELLX_BOTH) != 0u;
				const bool near_free = (fl & lsq::CELLX_NEAR_NO_ABUT) != 0u, far_free = (fl & lsq::CELLX_FAR_NO_ABUT) != 0u;
				const bool near_done = e0 <= cw.e1 || (near_free && 50 * (e0 - cw.e1) >= len), far_done = e0 <= cw.e2 || (far_free && 50 * (e0 - cw.e2) >= len);
				if (!pool) { parks = both ? !(near_done && far_done) : e0 > cw.e2; one_event = !both || near_done || far_done; }
				else if (cellx[cell].info == lsq::CELL_INFO_EMPTY) parks = false;
				else parks = both ? (e0 == cw.e1 || e0 == cw.e2 || (e0 > cw.e1 && !near_free) || (e0 > cw.e2 && !far_free)) : e0 > cw.e2;
			}
			if (parks) {
				unsigned looks = 1;
				if (!one_event) {
					looks = 0;
					for (unsigned i = br.x >> 16; i < d.n_events && looks < 64u; ++i) {
						++looks;
						const lsq::FastRec &fr = recs[i];
						if (!(fr.seg[0] <= p && (p > fr.ge || (fr.meta & lsq::FAST_FLAG_OVERLAPS_NEXT) != 0u))) break;
					}
				}
				looks_est = looks;
			}
		}
		(pool ? A.fine2 : A.fine1)[first + r] = gbase + g;
		atomicAdd(&hist[g], 1u);
		if (looks_est) atomicAdd(&park[g], looks_est);
	}
	__syncthreads();
	unsigned *cnt = pool ? A.cnt2 : A.cnt1, *pk = pool ? A.park2 : A.park1;
	for (unsigned q = threadIdx.x; q < n_groups; q += 256u) {
		const unsigned c = hist[q], l = park[q];
		if (pc.w) { if (c) cnt[gbase + q] = c; if (l) pk[gbase + q] = l; }       // the partition's only piece: the counters were zeroed, nobody else adds
		else { if (c) atomicAdd(&cnt[gbase + q], c); if (l) atomicAdd(&pk[gbase + q], l); }
	}
}

// per-bucket pool offsets out of the per-group ones
__global__ void __launch_bounds__(256) lsq_ingest_offsets_kernel(const unsigned *cell_base, const unsigned *jgroup_base, unsigned n_buckets, const unsigned long long *off1,
                                                                 const unsigned long long *off2, const unsigned long long *pn_off,
                                                                 unsigned long long *p1_off, unsigned long long *p2_off, unsigned long long *slot_off) {
	const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b > n_buckets) return;
	const unsigned long long a1 = off1[cell_base[b]], a2 = off2[jgroup_base[b]];
	p1_off[b] = a1; p2_off[b] = a2;
	slot_off[b] = a1 + a2 + pn_off[b];
}

// a read into the pool: the wide record as it is, or the compact one (CountArgs)
template <bool COMPACT> __device__ inline void pool_store(void *out, const unsigned long long at, const int2 r, const int base) {
	if (COMPACT) reinterpret_cast<unsigned *>(out)[at] = (unsigned)(r.x - base) | ((unsigned)(r.y - r.x) << lsq::COMPACT_OFF_BITS);
	else reinterpret_cast<int2 *>(out)[at] = r;
}
template <bool COMPACT> __device__ inline void pool_store(void *out, const unsigned long long at, const int4 r, const int base) {
	if (COMPACT) reinterpret_cast<uint2 *>(out)[at] = make_uint2((unsigned)(r.x - base) | ((unsigned)(r.y - r.x) << lsq::COMPACT_OFF_BITS),
	                                                              (unsigned)(r.z - r.y) | ((unsigned)(r.w - r.z) << lsq::COMPACT_OFF_BITS));
	else reinterpret_cast<int4 *>(out)[at] = r;
}

struct PlaceArgs {
	const uint4 *pieces;
	const unsigned long long *part_off1, *part_off2;
	const uint4 *part1, *part2;
	const unsigned *fine1, *fine2;
	const unsigned long long *off1, *off2;     // per group: its first place in the pool
	unsigned *cur1, *cur2;                     // per group: places handed out (pieces of a partition that has several)
	const BucketDesc *buckets;
	const unsigned *cell_base, *jg_base, *jgroup_base;
	unsigned B, compact;
	void *p1; unsigned char *p1_strand; unsigned *p1_line;           // the one-block pool itself (groups by cell: no sort follows)
	void *p2; unsigned char *p2_strand; unsigned *p2_line;           // the two-block pool itself (groups by junction)
};

// pass 2 over the partitions: every read to the next free place of its group, as the pool record it is to be.  A piece's
// places per group come from an LDS cursor; a partition of several pieces reserves them per piece with one atomic a group.
__global__ void __launch_bounds__(256) lsq_group_place_kernel(PlaceArgs A) {
	extern __shared__ __align__(16) unsigned char glds[];
	unsigned *cur = reinterpret_cast<unsigned *>(glds);
	const uint4 pc = A.pieces[blockIdx.x];
	const unsigned pool = pc.x >= A.B ? 1u : 0u, b = pc.x - pool * A.B;
	const BucketDesc &d = A.buckets[b];
	const unsigned n_cells = d.kind == 1u ? (d.iso_off & 0xFFFFu) : 0u;
	const unsigned n_groups = pool ? (A.jg_base[b + 1] - A.jg_base[b]) + n_cells + 1u : n_cells + 1u;
	const unsigned gbase = pool ? A.jgroup_base[b] : A.cell_base[b];
	const unsigned long long first = (pool ? A.part_off2[b] : A.part_off1[b]) + pc.y;
	const unsigned *fine = (pool ? A.fine2 : A.fine1) + first;
	// the groups' first places, relative to the bucket's stretch of the pool, beside their cursors: one trip to memory for all of them
	unsigned *goff = cur + n_groups;
	const unsigned long long *off = pool ? A.off2 : A.off1;
	const unsigned long long off0 = off[gbase];
	for (unsigned q = threadIdx.x; q < n_groups; q += 256u) { cur[q] = 0; goff[q] = (unsigned)(off[gbase + q] - off0); }
	__syncthreads();
	if (!pc.w) {
		for (unsigned r = threadIdx.x; r < pc.z; r += 256u) atomicAdd(&cur[fine[r] - gbase], 1u);
		__syncthreads();
		unsigned *gcur = pool ? A.cur2 : A.cur1;
		for (unsigned q = threadIdx.x; q < n_groups; q += 256u) { const unsigned c = cur[q]; cur[q] = c ? atomicAdd(&gcur[gbase + q], c) : 0u; }
		__syncthreads();
	}
	const int base = d.lo - lsq::COMPACT_BIAS;
	// (four records a lane at a time: loads together, then the places, then the stores)
	for (unsigned r0 = threadIdx.x; r0 < pc.z; r0 += 1024u) {
		unsigned g[4]; uint4 v[4], x[4]; bool on[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const unsigned r = r0 + 256u * (unsigned)u;
			on[u] = r < pc.z;
			g[u] = on[u] ? fine[r] - gbase : 0u;
			if (!pool) { v[u] = on[u] ? A.part1[first + r] : make_uint4(0, 0, 0, 0); x[u] = v[u]; }
			else { v[u] = on[u] ? A.part2[2ull * (first + r)] : make_uint4(0, 0, 0, 0); x[u] = on[u] ? A.part2[2ull * (first + r) + 1] : make_uint4(0, 0, 0, 0); }
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			if (!on[u]) continue;
			const unsigned long long w = off0 + goff[g[u]] + atomicAdd(&cur[g[u]], 1u);
			if (!pool) {
				const int2 rec = make_int2((int)v[u].x, (int)v[u].y);
				if (A.compact) pool_store<true>(A.p1, w, rec, base); else pool_store<false>(A.p1, w, rec, base);
				A.p1_strand[w] = (unsigned char)v[u].w; A.p1_line[w] = v[u].z;
			} else {
				const int4 rec = make_int4((int)v[u].x, (int)v[u].y, (int)v[u].z, (int)v[u].w);
				if (A.compact) pool_store<true>(A.p2, w, rec, base); else pool_store<false>(A.p2, w, rec, base);
				A.p2_strand[w] = (unsigned char)x[u].y; A.p2_line[w] = x[u].x;
			}
		}
	}
}

// the reads of pools 2 and 3 (three or more merged blocks; compact misfits): the list of the routing pass into the n-block pool, by bucket
struct NbOut {
	const uint4 *ent; const int2 *blk; unsigned long long n_ent;
	const unsigned long long *pn_off, *pnb_off;
	unsigned *curn, *curnb;
	unsigned *pn_blk_off, *pn_nblk, *pn_line, *pn_bucket; unsigned char *pn_strand; int2 *pn_se;
	const unsigned *line_no; unsigned long long first_line;
};
__global__ void __launch_bounds__(256) lsq_ingest_nblock_kernel(NbOut O) {
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < O.n_ent; t += gsz) {
		const uint4 en = O.ent[t];
		const unsigned i = en.x, b = en.y, n = en.z & 0xFFu, strand = en.z >> 8;
		const unsigned long long w = O.pn_off[b] + atomicAdd(&O.curn[b], 1u);
		const unsigned long long bo = O.pnb_off[b] + atomicAdd(&O.curnb[b], n);
		O.pn_blk_off[w] = (unsigned)bo; O.pn_nblk[w] = n; O.pn_bucket[w] = b;
		O.pn_strand[w] = (unsigned char)strand; O.pn_line[w] = O.line_no ? O.line_no[i] : (unsigned)(O.first_line + i);
		for (unsigned q = 0; q < n; ++q) O.pn_se[bo + q] = O.blk[(unsigned long long)en.w + q];
	}
}

// The padding of the groups (up to seven one-block, three two-block records each): empty reads.  A one-block one starts on its cell's first base, so
// that the count kernel's loop sees it as inside the cell and adding nothing; a two-block one never comes first among a
// lane's records and is passed over there.
__global__ void __launch_bounds__(256) lsq_ingest_pad_kernel(const BucketDesc *buckets, const unsigned *cell_base, const unsigned *jgroup_base, unsigned n_buckets,
                                                             unsigned n_groups1, unsigned n_groups2, const unsigned *cnt1, const unsigned long long *off1,
                                                             const unsigned *cnt2, const unsigned long long *off2, const unsigned char *images,
                                                             void *p1, void *p2, unsigned compact, unsigned char *p1_strand, unsigned *p1_line,
                                                             unsigned char *p2_strand, unsigned *p2_line) {
	for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n_groups1 + n_groups2; t += gridDim.x * blockDim.x) {
		const bool one = t < n_groups1;
		const unsigned g = one ? t : t - n_groups1;
		const unsigned n = one ? cnt1[g] : cnt2[g], gp = one ? P1_GROUP_PAD : P2_GROUP_PAD, pad = ((n + gp - 1u) & ~(gp - 1u)) - n;
		if (!pad) continue;
		const unsigned *gbase = one ? cell_base : jgroup_base;
		unsigned lo_b = 0, hi_b = n_buckets;                 // bucket of the group: last b with gbase[b] <= g
		while (hi_b - lo_b > 1) { const unsigned mid = (lo_b + hi_b) >> 1; if (gbase[mid] <= g) lo_b = mid; else hi_b = mid; }
		const BucketDesc &d = buckets[lo_b];
		const int base = d.lo - lsq::COMPACT_BIAS;
		if (!one) {
			for (unsigned q = 0; q < pad; ++q) {
				const unsigned long long w = off2[g] + n + q;
				if (compact) reinterpret_cast<uint2 *>(p2)[w] = make_uint2((unsigned)lsq::COMPACT_BIAS, 0u);
				else reinterpret_cast<int4 *>(p2)[w] = make_int4(d.lo, d.lo, d.lo, d.lo);
				p2_strand[w] = 0; p2_line[w] = 0;
			}
			continue;
		}
		const unsigned cell = g - cell_base[lo_b];
		int at = d.lo;
		if (d.kind == 1u && cell < (d.iso_off & 0xFFFFu)) at = reinterpret_cast<const lsq::Cell *>(images + d.img_off + d.seg_off)[cell].lo;
		if (compact && !lsq::compact_block_fits((long long)at - base, 1)) at = base;       // (a cell 2 Mi bases into its bucket: the loop then parks nothing for it all the same, the record is empty)
		for (unsigned q = 0; q < pad; ++q) {
			const unsigned long long w = off1[g] + n + q;
			if (compact) reinterpret_cast<unsigned *>(p1)[w] = (unsigned)(at - base);
			else reinterpret_cast<int2 *>(p1)[w] = make_int2(at, at);
			p1_strand[w] = 0; p1_line[w] = 0;
		}
	}
}

// ---- the chain on the host ---------------------------------------------------------------------------------------------
// the reads of one file as the routing pass meets them
struct Front {
	unsigned long long n = 0;                 // reads of the pass (text: data lines, skipped ones among them)
	const unsigned *line_no = nullptr;        // per read (device), or null: first_line + index
	unsigned long long first_line = 0;
	unsigned long long in_bytes = 0;          // what the routing pass reads
	std::function<int(const RouteTables &, const RouteOut &, hipStream_t)> launch;   // runs the routing kernel
	std::function<int(hipStream_t)> settle;   // once the stream has been waited for: the front end's own verdict (the first failing line)
};

static RouteTables route_tables(lsq_ctx *c) {
	RouteTables T{};
	T.chrom = c->route_chrom.p; T.cov = c->cov.p; T.clu = c->clu.p; T.loc = c->loc.p; T.loc_shift = c->loc_shift; T.n_chrom = c->n_chrom_tables;
	return T;
}

// Runs the chain over the reads a front end delivers (MRF text in HBM, or parsed blocks from the host).
static int ingest_device(lsq_ctx *c, int method, Front &F) {
	HostStopwatch SW;
	const lsq_events &E = *c->E;
	MethodReads &mr = c->reads[method];
	mr.present = false;
	const unsigned B = (unsigned)E.buckets.size();
	const unsigned long long n = F.n;
	if (B >= ROUTE_NO_BUCKET) return fail(LSQ_E_RANGE, "more than 2^22 buckets");
	if (n > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "more than 2^32 reads in one file");
	hipStream_t st = c->stream;
	int rc;
	if ((rc = ensure_lanes(c))) return rc;
	// the pools of this method are rewritten below: an exception pass of an earlier count may still read them on the result stream
	HIP_TRY(hipStreamSynchronize(c->stream_em2[0]));
	HIP_TRY(hipStreamSynchronize(c->stream_em2[1]));
	HIP_TRY(hipStreamSynchronize(c->stream_count2[0]));      // ... and a count of an earlier read set on a lane's count stream
	HIP_TRY(hipStreamSynchronize(c->stream_count2[1]));
	const size_t FC = c->n_cell_groups;          // one-block groups of all buckets
	const size_t FJ = c->n_junction_groups;      // two-block groups of all buckets
	// counters, one allocation (zeroed per attempt): part_cnt[2B] | part_cur[2B] | piece_cnt[2B] | cntn[B] | cntnb[B] | curn[B] | curnb[B] |
	// cnt1[FC] | cnt2[FJ] | cur1[FC] | cur2[FJ] | park1[FC] | park2[FJ]
	const size_t n_cnt = 10 * (size_t)B + 3 * (FC + FJ);
	DevBuf<unsigned> d_key, d_cnt, d_fine1, d_fine2;
	DevBuf<int4> d_rec;
	DevBuf<unsigned long long> d_tot, d_part_off1, d_part_off2, d_piece_off, d_off1, d_off2;
	DevBuf<uint4> d_nb_ent, d_part1, d_part2, d_pieces;
	DevBuf<int2> d_nb_blk;
	ScanScratch SS;
	if ((rc = d_key.alloc(n)) || (rc = d_rec.alloc(n)) || (rc = d_cnt.alloc(n_cnt)) || (rc = d_tot.alloc(12)) || (rc = d_part_off1.alloc(B + 1)) || (rc = d_part_off2.alloc(B + 1)) ||
	    (rc = d_piece_off.alloc(2 * (size_t)B + 1)) || (rc = d_off1.alloc(FC + 1)) || (rc = d_off2.alloc(FJ + 1)) || (rc = SS.reserve(std::max<size_t>(std::max(FC, FJ), 2 * (size_t)B)))) return rc;
	unsigned *part_cnt = d_cnt.p, *part_cur = part_cnt + 2 * (size_t)B, *piece_cnt = part_cur + 2 * (size_t)B;
	unsigned *cntn = piece_cnt + 2 * (size_t)B, *cntnb = cntn + B, *curn = cntnb + B, *curnb = curn + B;
	unsigned *cnt1 = curnb + B, *cnt2 = cnt1 + FC, *cur1 = cnt2 + FJ, *cur2 = cur1 + FC, *park1 = cur2 + FJ, *park2 = park1 + FC;
	if ((rc = mr.p1_off.alloc(B + 1)) || (rc = mr.p2_off.alloc(B + 1)) || (rc = mr.pn_off.alloc(B + 1)) || (rc = mr.pnb_off.alloc(B + 1)) || (rc = mr.slot_off.alloc(B + 1))) return rc;
	const RouteTables T = route_tables(c);
	// LDS of the partition kernels (a counter per pool and bucket) and of the group kernels (a bucket's image, its junction keys, two counters per group)
	const size_t part_lds = 8 * (size_t)B;
	const bool part_in_lds = part_lds <= 128 * 1024 && getenv("LSQ_PART_NO_LDS") == nullptr;       // (the environment switch: the tests run the form for very many buckets on a small input)
	size_t img_max = 16, jg_max = 0, groups_max = 1;
	for (unsigned b = 0; b < B; ++b) {
		const BucketDesc &d = E.buckets[b];
		const size_t n_cells = d.kind == 1u ? (d.iso_off & 0xFFFFu) : 0u, n_jg = E.jg_base[b + 1] - E.jg_base[b];
		if (d.kind == 1u) img_max = std::max<size_t>(img_max, (d.img_bytes + 15u) & ~15u);
		jg_max = std::max(jg_max, n_jg);
		groups_max = std::max(groups_max, n_jg + n_cells + 1);
	}
	const size_t keys_lds = (8 * jg_max + 15) & ~(size_t)15, group_lds = img_max + keys_lds + 8 * groups_max, place_lds = 8 * groups_max;
	if (group_lds > 160 * 1024) return fail(LSQ_E_UNSUPPORTED, "a bucket's tables and group counters exceed the CU's LDS");
	if (part_in_lds && part_lds > 48 * 1024) {
		HIP_TRY(hipFuncSetAttribute((const void *)lsq_part_hist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds));
		HIP_TRY(hipFuncSetAttribute((const void *)lsq_part_scatter_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds));
	}
	if (group_lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)lsq_group_classify_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds));
	if (place_lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)lsq_group_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)place_lds));
	const unsigned part_grid = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(n / 131072ull, 2ull * (unsigned)c->n_cu));
	SW.mark("ingest: allocations");

	unsigned compact = c->opt_compact_pools ? 1u : 0u;
	unsigned long long nb_cap = std::max<unsigned long long>(4096, n / 32), nbb_cap = 4 * nb_cap;
	if (const char *e = getenv("LSQ_NB_LIST")) { const long long v = atoll(e); if (v > 0) { nb_cap = (unsigned long long)v; nbb_cap = 2 * nb_cap; } }      // tests: the list runs over and is sized again
	unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nb_tot[4] = {0, 0, 0, 0}, sums[5] = {0, 0, 0, 0, 0};      // sums: reads of pool 0 / pool 1 partitions, pieces, n-block reads, their blocks
	for (;;) {
		if (d_nb_ent.n < nb_cap) { if ((rc = d_nb_ent.alloc((size_t)nb_cap))) return rc; }
		if (d_nb_blk.n < nbb_cap) { if ((rc = d_nb_blk.alloc((size_t)nbb_cap))) return rc; }
		HIP_TRY(hipMemsetAsync(d_cnt.p, 0, std::max<size_t>(n_cnt, 1) * 4, st));
		HIP_TRY(hipMemsetAsync(d_tot.p, 0, 12 * 8, st));
		RouteOut O{};
		O.key = d_key.p; O.rec = d_rec.p; O.nb_tot = d_tot.p + 8; O.nb_cap = nb_cap; O.nbb_cap = nbb_cap; O.nb_ent = d_nb_ent.p; O.nb_blk = d_nb_blk.p;
		O.cntn = cntn; O.cntnb = cntnb; O.compact = compact;
		if (n) {
			StageClock k(c, st, 1);
			if ((rc = F.launch(T, O, st))) return rc;
			k.end(F.in_bytes + 20ull * n);
		}
		{
			StageClock k(c, st, 2);
			if (part_in_lds) hipLaunchKernelGGL(lsq_part_hist_kernel<true>, dim3(part_grid), dim3(PART_WG), part_lds, st, (const unsigned *)d_key.p, n, B, part_cnt, d_tot.p);
			else hipLaunchKernelGGL(lsq_part_hist_kernel<false>, dim3(part_grid), dim3(PART_WG), 0, st, (const unsigned *)d_key.p, n, B, part_cnt, d_tot.p);
			HIP_TRY(hipGetLastError());
			if ((rc = device_scan<1>(SS, part_cnt, B, d_part_off1.p, st)) || (rc = device_scan<1>(SS, part_cnt + B, B, d_part_off2.p, st)) ||
			    (rc = device_scan<1>(SS, cntn, B, mr.pn_off.p, st)) || (rc = device_scan<1>(SS, cntnb, B, mr.pnb_off.p, st))) return rc;
			hipLaunchKernelGGL(lsq_piece_count_kernel, dim3(2 * B / 256 + 1), dim3(256), 0, st, (const unsigned *)part_cnt, 2 * B, piece_cnt);
			HIP_TRY(hipGetLastError());
			if ((rc = device_scan<1>(SS, piece_cnt, 2ull * B, d_piece_off.p, st))) return rc;
			k.end(4ull * n);
		}
		HIP_TRY(hipMemcpyAsync(tot, d_tot.p, 8 * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(nb_tot, d_tot.p + 8, 4 * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[0], d_part_off1.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[1], d_part_off2.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[2], d_piece_off.p + 2 * (size_t)B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[3], mr.pn_off.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[4], mr.pnb_off.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		SW.mark("ingest: route + partition counts");
		if (F.settle && (rc = F.settle(st))) { if (rc == LSQ_RETRY) continue; return rc; }
		// compact records pay when nearly every one- and two-block read fits them (the others are counted a lane a read, tables
		// in L2); a read set of long blocks -- more than 1 in 16 does not fit -- is routed again for wide records.  So is one
		// whose many-block reads did not fit the list (a file of long spliced reads), with a list of the size it asked for.
		const bool go_wide = compact && tot[3] * 16 > tot[4] + tot[5] + tot[3];
		const bool list_short = nb_tot[0] > nb_cap || nb_tot[1] > nbb_cap;
		if (go_wide || list_short) {
			if (go_wide) compact = 0;
			if (list_short) { nb_cap = std::max(nb_cap, nb_tot[0]); nbb_cap = std::max(nbb_cap, nb_tot[1]); }
			continue;
		}
		break;
	}
	if (nb_tot[2]) return fail(LSQ_E_RANGE, "a read covers 2^18 or more bases or keeps more than %d separate blocks: outside the device tables' range", INGEST_MAX_BLOCKS);
	if (sums[4] > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "too many blocks in multi-block reads");
	const size_t n1p = (size_t)sums[0], n2p = (size_t)sums[1], n_pieces = (size_t)sums[2], nn = (size_t)sums[3], nnb = (size_t)sums[4];
	if (n_pieces > 0x7FFFFFFFull) return fail(LSQ_E_RANGE, "too many partition pieces");
	if ((rc = d_part1.alloc(n1p)) || (rc = d_part2.alloc(2 * n2p)) || (rc = d_fine1.alloc(n1p)) || (rc = d_fine2.alloc(n2p)) || (rc = d_pieces.alloc(n_pieces))) return rc;
	SW.mark("ingest: partition buffers");
	{
		StageClock k(c, st, 3);
		PartArgs A{};
		A.key = d_key.p; A.rec = d_rec.p; A.n = n; A.B = B; A.off1 = d_part_off1.p; A.off2 = d_part_off2.p; A.cursor = part_cur;
		A.part1 = d_part1.p; A.part2 = d_part2.p; A.line_no = F.line_no; A.first_line = F.first_line;
		if (n1p + n2p) {
			if (part_in_lds) hipLaunchKernelGGL(lsq_part_scatter_kernel<true>, dim3(part_grid), dim3(PART_WG), part_lds, st, A);
			else hipLaunchKernelGGL(lsq_part_scatter_kernel<false>, dim3(part_grid), dim3(PART_WG), 0, st, A);
		}
		hipLaunchKernelGGL(lsq_piece_expand_kernel, dim3(2 * B / 256 + 1), dim3(256), 0, st, (const unsigned *)part_cnt, (const unsigned long long *)d_piece_off.p, 2 * B, d_pieces.p);
		HIP_TRY(hipGetLastError());
		k.end(8ull * n + 16ull * (n1p + n2p) + 16ull * n1p + 32ull * n2p);
	}
	GroupArgs GA{};
	GA.pieces = d_pieces.p; GA.off1 = d_part_off1.p; GA.off2 = d_part_off2.p; GA.part1 = d_part1.p; GA.part2 = d_part2.p;
	GA.fine1 = d_fine1.p; GA.fine2 = d_fine2.p; GA.cnt1 = cnt1; GA.cnt2 = cnt2; GA.park1 = park1; GA.park2 = park2;
	if (n_pieces) {
		StageClock k(c, st, 4);
		GroupTables GT{};
		GT.buckets = c->buckets.p; GT.images = c->images.p; GT.cell_base = c->cell_base.p; GT.jg_keys = c->jg_keys.p; GT.jg_base = c->jg_base.p; GT.jgroup_base = c->jgroup_base.p;
		GT.B = B; GT.lds_img = (unsigned)img_max; GT.lds_keys = (unsigned)keys_lds;
		hipLaunchKernelGGL(lsq_group_classify_kernel, dim3((unsigned)n_pieces), dim3(256), group_lds, st, GT, GA);
		HIP_TRY(hipGetLastError());
		k.end(16ull * n1p + 32ull * n2p + 4ull * (n1p + n2p));
	}
	unsigned long long psum[2] = {0, 0};
	{
		StageClock k(c, st, 5);
		if ((rc = device_scan<P1_GROUP_PAD>(SS, cnt1, FC, d_off1.p, st)) || (rc = device_scan<P2_GROUP_PAD>(SS, cnt2, FJ, d_off2.p, st))) return rc;      // groups padded to eight records, and to four
		hipLaunchKernelGGL(lsq_ingest_offsets_kernel, dim3(B / 256 + 1), dim3(256), 0, st, c->cell_base.p, c->jgroup_base.p, B, d_off1.p, d_off2.p, mr.pn_off.p,
		                   mr.p1_off.p, mr.p2_off.p, mr.slot_off.p);
		HIP_TRY(hipGetLastError());
		k.end(12ull * (FC + FJ));
	}
	SW.mark("ingest: partition + group launches");
	HIP_TRY(hipMemcpyAsync(&psum[0], mr.p1_off.p + B, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&psum[1], mr.p2_off.p + B, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	SW.mark("ingest: partition + groups");
	const size_t n1 = (size_t)psum[0], n2 = (size_t)psum[1];
	mr.compact = compact != 0;
	mr.n1_reads = tot[4]; mr.n2_reads = tot[5];
	// (n1, n2: the pools' slots, the groups' padding among them: multiples of four and of two, so compact pools are whole
	// 16-byte words)
	if ((rc = mr.p1.alloc(mr.compact ? ((n1 + 3) & ~(size_t)3) : 2 * n1)) || (rc = mr.p1_strand.alloc(n1)) || (rc = mr.p1_line.alloc(n1))) return rc;
	if ((rc = mr.p2.alloc(mr.compact ? 2 * ((n2 + 1) & ~(size_t)1) : 4 * n2)) || (rc = mr.p2_strand.alloc(n2)) || (rc = mr.p2_line.alloc(n2))) return rc;
	if ((rc = mr.pn_se.alloc(2 * nnb)) || (rc = mr.pn_blk_off.alloc(nn)) || (rc = mr.pn_nblk.alloc(nn)) || (rc = mr.pn_strand.alloc(nn)) ||
	    (rc = mr.pn_line.alloc(nn)) || (rc = mr.pn_bucket.alloc(nn))) return rc;
	{
		StageClock k(c, st, 6);
		if (n_pieces) {
			// the pools themselves: their groups need no order inside
			PlaceArgs P{};
			P.pieces = d_pieces.p; P.part_off1 = d_part_off1.p; P.part_off2 = d_part_off2.p; P.part1 = d_part1.p; P.part2 = d_part2.p;
			P.fine1 = d_fine1.p; P.fine2 = d_fine2.p; P.off1 = d_off1.p; P.off2 = d_off2.p; P.cur1 = cur1; P.cur2 = cur2;
			P.buckets = c->buckets.p; P.cell_base = c->cell_base.p; P.jg_base = c->jg_base.p; P.jgroup_base = c->jgroup_base.p; P.B = B; P.compact = compact;
			P.p1 = mr.p1.p; P.p1_strand = mr.p1_strand.p; P.p1_line = mr.p1_line.p; P.p2 = mr.p2.p; P.p2_strand = mr.p2_strand.p; P.p2_line = mr.p2_line.p;
			hipLaunchKernelGGL(lsq_group_place_kernel, dim3((unsigned)n_pieces), dim3(256), place_lds, st, P);
		}
		if (FC + FJ) {
			const unsigned pgrid = (unsigned)std::min<size_t>((FC + FJ) / 256 + 1, (size_t)c->n_cu * 8);
			hipLaunchKernelGGL(lsq_ingest_pad_kernel, dim3(pgrid), dim3(256), 0, st, c->buckets.p, c->cell_base.p, c->jgroup_base.p, B, (unsigned)FC, (unsigned)FJ,
			                   (const unsigned *)cnt1, (const unsigned long long *)d_off1.p, (const unsigned *)cnt2, (const unsigned long long *)d_off2.p, c->images.p,
			                   (void *)mr.p1.p, (void *)mr.p2.p, compact, mr.p1_strand.p, mr.p1_line.p, mr.p2_strand.p, mr.p2_line.p);
		}
		if (nn) {
			NbOut N{};
			N.ent = d_nb_ent.p; N.blk = d_nb_blk.p; N.n_ent = nb_tot[0]; N.pn_off = mr.pn_off.p; N.pnb_off = mr.pnb_off.p; N.curn = curn; N.curnb = curnb;
			N.pn_blk_off = mr.pn_blk_off.p; N.pn_nblk = mr.pn_nblk.p; N.pn_line = mr.pn_line.p; N.pn_bucket = mr.pn_bucket.p; N.pn_strand = mr.pn_strand.p;
			N.pn_se = reinterpret_cast<int2 *>(mr.pn_se.p); N.line_no = F.line_no; N.first_line = F.first_line;
			const unsigned ngrid = (unsigned)std::min<unsigned long long>((nb_tot[0] + 255) / 256, (unsigned long long)c->n_cu * 16);
			hipLaunchKernelGGL(lsq_ingest_nblock_kernel, dim3(ngrid), dim3(256), 0, st, N);
		}
		HIP_TRY(hipGetLastError());
		k.end(16ull * n1p + 32ull * n2p + 4ull * (n1p + n2p) + (mr.compact ? 4ull : 8ull) * (n1 + 2 * n2) + 5ull * (n1 + n2));
	}
	{
		// exception list: a quarter of the one- and two-block reads, at least 64 Ki entries
		size_t want = std::max<size_t>(65536, (n1 + n2) / 4);
		if (c->opt_exc_cap) want = c->opt_exc_cap;           // lsq_ctx_set_option "exception_capacity" (the tests provoke the overflow path with it)
		if (mr.exc_cap != want) {
			if ((rc = mr.exc.alloc(2 * want))) return rc;
			mr.exc_cap = want;
		}
	}
	if ((rc = upload_strand_ranks(c))) return rc;      // the reads may have introduced new strand strings
	mr.n_retained = tot[0];
	mr.n_retained_blocks = tot[1] + nb_tot[1];
	mr.total_slots = n1 + n2 + nn;
	mr.wg_grid = 0;
	{
		// how unevenly the reads fall on the buckets: with hot genes the reads that need the general walk
		// fill whole workgroup shares, and smaller shares (more workgroups) even the load out
		std::vector<unsigned long long> so(B + 1, 0), o1(B + 1, 0), o2(B + 1, 0), f1(FC + 1, 0), f2(FJ + 1, 0);
		std::vector<unsigned> park(FC + FJ + 1, 0);
		HIP_TRY(hipMemcpyAsync(so.data(), mr.slot_off.p, (B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		// the count kernel's visit records (lsq_device.hpp VisitRec)
		HIP_TRY(hipMemcpyAsync(o1.data(), mr.p1_off.p, (B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(o2.data(), mr.p2_off.p, (B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		// ... and what the share plan weighs: the records of every cell and junction group and the looks the general walk will take at
		// them (run_count: plan_share_cuts_seg), as stretches of slots; per bucket the sums as well
		HIP_TRY(hipMemcpyAsync(f1.data(), d_off1.p, (FC + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(f2.data(), d_off2.p, (FJ + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		if (FC + FJ) HIP_TRY(hipMemcpyAsync(park.data(), park1, (FC + FJ) * sizeof(unsigned), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));            // (also: the place kernels are done, the work arrays may go)
		SW.mark("ingest: place + read-back");
		stages_collect(c);
		unsigned long long mx = 0;
		for (unsigned b = 0; b < B; ++b) mx = std::max(mx, so[b + 1] - so[b]);
		mr.slot_off_host = so;
		mr.skew = (B && mr.total_slots) ? (double)mx * (double)B / (double)mr.total_slots : 1.0;
		{
			mr.plan_n1.resize(B); mr.plan_n2.resize(B); mr.plan_park1.assign(B, 0); mr.plan_park2.assign(B, 0);
			mr.plan_seg_x.clear(); mr.plan_seg_kind.clear(); mr.plan_seg_looks.clear(); mr.plan_seg_first.assign(B + 1, 0);
			size_t g1 = 0, g2 = 0;          // the buckets' groups follow one another: one per cell and one for "no cell"; the junction groups, then one per cell + 1
			for (unsigned b = 0; b < B; ++b) {
				const unsigned n_cg = (E.buckets[b].kind == 1 ? (E.buckets[b].iso_off & 0xFFFFu) : 0u) + 1u, n_jg = (E.jg_base[b + 1] - E.jg_base[b]) + n_cg;
				const unsigned long long n1 = o1[b + 1] - o1[b], n2 = o2[b + 1] - o2[b], ns = so[b + 1] - so[b];
				mr.plan_n1[b] = n1; mr.plan_n2[b] = n2;
				mr.plan_seg_first[b] = (unsigned)mr.plan_seg_kind.size();
				const bool visited = E.buckets[b].kind == 1u && ns != 0;
				if (visited) {
					for (unsigned q = 0; q < n_cg && g1 + q < FC; ++q) {
						const unsigned long long a = f1[g1 + q], e = f1[g1 + q + 1];
						mr.plan_park1[b] += park[g1 + q];
						if (e > a) { mr.plan_seg_x.push_back(so[b] + (a - o1[b])); mr.plan_seg_kind.push_back(0); mr.plan_seg_looks.push_back(park[g1 + q]); }
					}
					for (unsigned q = 0; q < n_jg && g2 + q < FJ; ++q) {
						const unsigned long long a = f2[g2 + q], e = f2[g2 + q + 1];
						mr.plan_park2[b] += park[FC + g2 + q];
						if (e > a) { mr.plan_seg_x.push_back(so[b] + n1 + (a - o2[b])); mr.plan_seg_kind.push_back(1); mr.plan_seg_looks.push_back(park[FC + g2 + q]); }
					}
					if (ns > n1 + n2) { mr.plan_seg_x.push_back(so[b] + n1 + n2); mr.plan_seg_kind.push_back(2); mr.plan_seg_looks.push_back(0); }
				} else if (ns) { mr.plan_seg_x.push_back(so[b]); mr.plan_seg_kind.push_back(2); mr.plan_seg_looks.push_back(0); }
				g1 += n_cg; g2 += n_jg;
			}
			mr.plan_seg_first[B] = (unsigned)mr.plan_seg_kind.size();
			mr.plan_seg_x.push_back(B ? so[B] : 0);
		}
		std::vector<VisitRec> vis(B + 1);
		mr.next_packed_host.assign(B + 1, B);
		memset(vis.data(), 0, vis.size() * sizeof(VisitRec));
		unsigned next = B;
		for (unsigned b = B; b-- > 0;) {
			VisitRec &v = vis[b];
			v.d = E.buckets[b];
			v.bs = so[b]; v.be = so[b + 1];
			v.p1o = o1[b]; v.p1n = o1[b + 1] - o1[b]; v.p2o = o2[b]; v.p2n = o2[b + 1] - o2[b];
			v.b = b; v.next = next;
			if (E.buckets[b].kind == 1u && so[b + 1] > so[b]) next = b;
			mr.next_packed_host[b] = next;
		}
		vis[B].b = B; vis[B].next = B; vis[B].bs = vis[B].be = mr.total_slots;
		if ((rc = mr.visits.upload(vis.data(), vis.size(), st))) return rc;
		mr.visits_host = vis;
		HIP_TRY(hipStreamSynchronize(st));           // the host vector goes out of scope
	}
	mr.present = true;
	c->counted = c->solved = false;
	SW.mark("ingest: plan, visit records");
	return LSQ_OK;
}

// the front end of parsed blocks from the host (lsq_reads_upload)
static void front_of_raw(lsq_ctx *c, const IngestRaw &Rw, unsigned long long n_blocks, Front &F) {
	F.n = Rw.n_reads; F.line_no = Rw.line_no; F.first_line = 0;
	F.in_bytes = 12ull * Rw.n_reads + 11ull * n_blocks;
	F.launch = [c, Rw](const RouteTables &T, const RouteOut &O, hipStream_t st) -> int {
		const unsigned igrid = (unsigned)std::min<unsigned long long>((Rw.n_reads + 255) / 256 + 1, (unsigned long long)c->n_cu * 16);
		hipLaunchKernelGGL(lsq_route_raw_kernel, dim3(igrid), dim3(256), 0, st, T, Rw, O);
		HIP_TRY(hipGetLastError());
		return LSQ_OK;
	};
	F.settle = nullptr;
}

// MRF text in HBM through the chain: newline counts, then the parse as the chain's routing pass
static int ingest_text(lsq_ctx *c, int method, const char *read_format, lsq_text &T, unsigned has_header, unsigned long long first_line) {
	if (!read_format) return fail(LSQ_E_ARG, "null argument");
	if (strcmp(read_format, "MRF_SINGLE") != 0) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", read_format);
	hipStream_t st = c->stream;
	int rc;
	c->mrf_h2d_ms = T.h2d_ms; c->mrf_parse_ms = 0;
	stages_reset(c, T.scanned);
	if (T.len && (rc = scan_newlines(c, T))) return rc;
	const unsigned long long n_nl = T.len ? T.n_nl : 0;
	const unsigned long long n_lines = n_nl >= 1 + has_header ? n_nl - has_header : 0;       // (header only, or no terminated line at all: no reads)
	if (first_line + n_lines > 0xFFFFFFFFull) return fail(LSQ_E_RANGE, "more than 2^32 lines");
	MrfDictDev DD;
	if ((rc = DD.build(c, st))) return rc;
	Front F;
	F.n = n_lines; F.line_no = nullptr; F.first_line = first_line; F.in_bytes = T.len;
	const unsigned n_tiles = (unsigned)((T.len + MRF_TILE - 1) / MRF_TILE);
	const MrfText X{T.d_text.p, T.len, T.d_tile_base.p, has_header, first_line, n_lines};
	// what the fast kernel hands on: tiles with more delimiters than its tables hold, lines of another shape than a read's
	// (at most one a tile begins ahead of its window; the rest is whatever the file holds -- when the list runs over, the
	// whole file goes through the kernel that walks bytes)
	unsigned long long list_cap = 1ull << 22;
	if (const char *e = getenv("LSQ_MRF_LINE_LIST")) { const long long v = atoll(e); if (v >= 0) list_cap = (unsigned long long)v; }      // tests: the run-over path on a small file
	const unsigned line_cap = (unsigned)std::min<unsigned long long>(n_lines, list_cap) + n_tiles + 1u;
	DevBuf<MrfLongLine> d_lines;
	DevBuf<unsigned> d_tiles, d_counts;
	DevBuf<unsigned long long> d_ckey;
	DevBuf<unsigned short> d_cid;
	if ((rc = d_lines.alloc(line_cap)) || (rc = d_tiles.alloc(n_tiles)) || (rc = d_counts.alloc(4))) return rc;
	HIP_TRY(hipMemsetAsync(d_counts.p, 0, 16, st));          // (a file without lines launches nothing; the verdict below still reads these)
	MrfHandOff H{d_counts.p, d_tiles.p, n_tiles, d_lines.p, line_cap};
	// the chromosomes' names as 64-bit keys (names of at most seven bytes; a longer one has no slot and its lines go to the list)
	MrfFastDict FD{};
	{
		const lsq_events &E = *c->E;
		const size_t nc = E.covered.size();
		std::vector<unsigned long long> ck(FP_DICT, 0);
		std::vector<unsigned short> ci(FP_DICT, 0);
		bool usable = nc <= ROUTE_CHROM_LDS && getenv("LSQ_MRF_SLOW") == nullptr;       // (LSQ_MRF_SLOW: the tests run the byte-walking kernel over whole files with it)
		for (size_t id = 0; usable && id < nc; ++id) {
			const std::string &nm = E.chroms.names[id];
			if (nm.empty() || nm.size() > 7) continue;
			const unsigned long long key = mrf_strand_key(nm.data(), nm.size());
			unsigned sl = mrf_key_slot(key);
			while (ck[sl] != 0) sl = (sl + 1u) & (FP_DICT - 1u);
			ck[sl] = key; ci[sl] = (unsigned short)id;
		}
		if ((rc = d_ckey.upload(ck.data(), FP_DICT, st)) || (rc = d_cid.upload(ci.data(), FP_DICT, st))) return rc;
		HIP_TRY(hipStreamSynchronize(st));
		FD.ckey = d_ckey.p; FD.cid = d_cid.p; FD.usable = usable ? 1u : 0u;
	}
	bool all_slow = !FD.usable;
	const unsigned side_grid = std::min(std::max(n_tiles, 1u), 4u * (unsigned)c->n_cu);
	// a workgroup a tile: the kernel can also run as a grid of resident workgroups that stay for many tiles (LSQ_FAST_GRID workgroups a
	// compute unit; developer aid) -- measured slower on C3: 7.4 ms at 6, 7.0 at 12, 6.7 at 24 against 6.1 with a workgroup a tile
	unsigned fast_grid = std::max(n_tiles, 1u);
	if (const char *e = getenv("LSQ_FAST_GRID")) { const int v = atoi(e); if (v > 0) fast_grid = (unsigned)v * (unsigned)c->n_cu; }
	F.launch = [&](const RouteTables &RT, const RouteOut &O, hipStream_t s) -> int {
		int r2 = DD.reset_errors(s);
		if (r2) return r2;
		HIP_TRY(hipMemsetAsync(d_counts.p, 0, 16, s));
		if (!all_slow) {
			hipLaunchKernelGGL(lsq_mrf_route_fast_kernel, dim3(std::min(n_tiles, fast_grid)), dim3(256), 0, s, X, DD.D, FD, RT, O, DD.d_err.p, H, n_tiles);
			hipLaunchKernelGGL(lsq_mrf_route_kernel, dim3(std::min(side_grid, 256u)), dim3(256), 0, s, X, DD.D, RT, O, DD.d_err.p, H, n_tiles, 1u);
		} else hipLaunchKernelGGL(lsq_mrf_route_kernel, dim3(n_tiles), dim3(256), 0, s, X, DD.D, RT, O, DD.d_err.p, H, n_tiles, 0u);
		hipLaunchKernelGGL(lsq_mrf_route_lines_kernel, dim3(std::min(line_cap / 256u + 1u, 1024u)), dim3(256), 0, s, X, DD.D, RT, O, DD.d_err.p, H);
		HIP_TRY(hipGetLastError());
		return LSQ_OK;
	};
	F.settle = [&](hipStream_t s) -> int {
		// the line list ran over (a file of lines of another shape than a read's): once more, every tile through the byte-walking kernel
		unsigned counts[4] = {0, 0, 0, 0};
		HIP_TRY(hipMemcpy(counts, d_counts.p, 16, hipMemcpyDeviceToHost));
		if (counts[2] && !all_slow) { all_slow = true; return LSQ_RETRY; }
		if (counts[2]) return fail(LSQ_E_INTERNAL, "the device parser's line list ran over");
		c->parse_tiles_handed = all_slow ? 0u : counts[0]; c->parse_lines_listed = counts[1]; c->parse_all_slow = all_slow ? 1u : 0u;
		return DD.settle(c, T, has_header, first_line, s);
	};
	c->reads[method].named = false;
	rc = ingest_device(c, method, F);
	if (rc) return rc;
	// (device time of the parse = the newline count and the routing pass; the rest of the chain is the ingest)
	c->mrf_parse_ms = c->ing_ms[0] + c->ing_ms[1];
	return LSQ_OK;
}

} // namespace

extern "C" {

int lsq_reads_upload(lsq_ctx *c, int method, const lsq_reads *R) LSQ_API_TRY {
	if (!c || !R) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (method < 0 || method >= c->E->n_methods) return fail(LSQ_E_ARG, "method %d out of range", method);
	HIP_TRY(hipSetDevice(c->device));
	const uint64_t n = R->n_reads, nblk = R->n_blocks;
	hipStream_t st = c->stream;
	int rc;
	HostStopwatch SW;
	// the parsed blocks, file order
	DevBuf<unsigned long long> d_blk_off;
	DevBuf<unsigned> d_line;
	DevBuf<int> d_bs, d_be;
	DevBuf<unsigned short> d_bc;
	DevBuf<unsigned char> d_bst;
	const unsigned long long zero_off = 0;
	// (large arrays through the context's pinned buffers, filled by a few threads: the runtime's own copy of fresh pageable memory
	// pins every page it is given first -- 0.116 s for C3's 2.4 GB on their first copy, 0.045 on a repeat)
	auto up = [&](auto &buf, const auto *src, size_t count, const char *what) -> int {
		typedef typename std::remove_reference<decltype(*buf.p)>::type E;
		const size_t bytes = count * sizeof(E);
		if (bytes < (64ull << 20)) return buf.upload(src, count, st);
		int r2 = buf.alloc(count);
		if (r2) return r2;
		const unsigned char *s8 = reinterpret_cast<const unsigned char *>(src);
		return pinned_pipeline(c, reinterpret_cast<unsigned char *>(buf.p), bytes, [s8](unsigned char *dst, size_t off, size_t nby) { memcpy(dst, s8 + off, nby); return true; }, what);
	};
	if ((rc = up(d_blk_off, n ? (const unsigned long long *)R->blk_off : &zero_off, n + 1, "block offsets"))) return rc;
	if ((rc = up(d_line, R->line_no, n, "line numbers"))) return rc;
	if ((rc = up(d_bs, R->blk_start, nblk, "block starts"))) return rc;
	if ((rc = up(d_be, R->blk_end, nblk, "block ends"))) return rc;
	if ((rc = up(d_bc, R->blk_chrom, nblk, "block chromosomes"))) return rc;
	if ((rc = up(d_bst, R->blk_strand, nblk, "block strands"))) return rc;
	IngestRaw Rw{};
	Rw.n_reads = n; Rw.blk_off = d_blk_off.p; Rw.line_no = d_line.p; Rw.blk_start = d_bs.p; Rw.blk_end = d_be.p;
	Rw.blk_chrom = d_bc.p; Rw.blk_strand = d_bst.p;
	SW.mark("upload: allocations, copies queued");
	if (SW.on) { HIP_TRY(hipStreamSynchronize(st)); SW.mark("upload: copies done"); }
	stages_reset(c, false);
	Front F;
	front_of_raw(c, Rw, nblk, F);
	if ((rc = ingest_device(c, method, F))) return rc;
	MethodReads &mr = c->reads[method];
	mr.named = R->named;
	if (R->named) {
		if ((rc = mr.names.upload(R->name_blob.data(), R->name_blob.size(), st))) return rc;
		if ((rc = mr.name_off.upload((const unsigned long long *)R->name_off.data(), R->name_off.size(), st))) return rc;
		HIP_TRY(hipStreamSynchronize(st));
	}
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_reads_upload_mrf(lsq_ctx *c, int method, const char *read_format, const char *path) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (method < 0 || method >= c->E->n_methods) return fail(LSQ_E_ARG, "method %d out of range", method);
	HIP_TRY(hipSetDevice(c->device));
	HostStopwatch SW;
	int rc = check_mrf_file(read_format, path);
	if (rc) return rc;
	lsq_text T;
	if ((rc = stage_text_file(c, path, 0, ~0ull, T))) return rc;
	rc = ingest_text(c, method, read_format, T, 1u, 1ull);
	SW.mark("upload_mrf: all");
	return rc;
} LSQ_API_CATCH

int lsq_text_stage_range(lsq_ctx *c, const char *path, uint64_t byte_begin, uint64_t byte_end, lsq_text **out) LSQ_API_TRY {
	if (!c || !path || !out) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	std::unique_ptr<lsq_text> T(new lsq_text);
	int rc = stage_text_file(c, path, byte_begin, byte_end, *T);
	if (rc) return rc;
	*out = T.release();
	return LSQ_OK;
} LSQ_API_CATCH
int lsq_text_stage(lsq_ctx *c, const char *path, lsq_text **out) { return lsq_text_stage_range(c, path, 0, ~0ull, out); }

int lsq_text_lines(lsq_ctx *c, lsq_text *t, uint64_t *n_newlines) LSQ_API_TRY {
	if (!c || !t || !n_newlines) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	int rc = t->len ? scan_newlines(c, *t) : LSQ_OK;
	if (rc) return rc;
	*n_newlines = t->len ? t->n_nl : 0;
	return LSQ_OK;
} LSQ_API_CATCH
void lsq_text_free(lsq_text *t) { delete t; }

int lsq_reads_upload_text(lsq_ctx *c, int method, const char *read_format, lsq_text *t) LSQ_API_TRY {
	return lsq_reads_upload_text_at(c, method, read_format, t, 1, 1);
} LSQ_API_CATCH

int lsq_reads_upload_text_at(lsq_ctx *c, int method, const char *read_format, lsq_text *t, int has_header, uint64_t first_line) LSQ_API_TRY {
	if (!c || !t) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (method < 0 || method >= c->E->n_methods) return fail(LSQ_E_ARG, "method %d out of range", method);
	HIP_TRY(hipSetDevice(c->device));
	return ingest_text(c, method, read_format, *t, has_header ? 1u : 0u, first_line);
} LSQ_API_CATCH

int lsq_mrf_parse_device(lsq_ctx *c, const char *read_format, const char *path, lsq_reads **out) LSQ_API_TRY {
	if (!c || !out) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	int rc = check_mrf_file(read_format, path);
	if (rc) return rc;
	lsq_text T;
	if ((rc = stage_text_file(c, path, 0, ~0ull, T))) return rc;
	DevParsed P;
	if ((rc = parse_staged_text(c, read_format, T, 1u, 1ull, P, &c->mrf_h2d_ms, &c->mrf_parse_ms))) return rc;
	std::unique_ptr<lsq_reads> R(new lsq_reads);
	R->o_blk_off.resize(P.n_reads + 1); R->o_line_no.resize(P.n_reads);
	R->o_start.resize(P.n_blocks); R->o_end.resize(P.n_blocks); R->o_chrom.resize(P.n_blocks); R->o_strand.resize(P.n_blocks);
	HIP_TRY(hipMemcpy(R->o_blk_off.data(), P.blk_off.p, (P.n_reads + 1) * 8, hipMemcpyDeviceToHost));
	if (P.n_reads) HIP_TRY(hipMemcpy(R->o_line_no.data(), P.line_no.p, P.n_reads * 4, hipMemcpyDeviceToHost));
	if (P.n_blocks) {
		HIP_TRY(hipMemcpy(R->o_start.data(), P.bs.p, P.n_blocks * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(R->o_end.data(), P.be.p, P.n_blocks * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(R->o_chrom.data(), P.bc.p, P.n_blocks * 2, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(R->o_strand.data(), P.bst.p, P.n_blocks, hipMemcpyDeviceToHost));
	}
	R->adopt();
	*out = R.release();
	return LSQ_OK;
} LSQ_API_CATCH

// developer entry (include/lesseq_hip_dev.h): which of the parse's three kernels the latest MRF text went through
int lsq_debug_last_parse_paths(const lsq_ctx *c, unsigned *tiles_handed, unsigned *lines_listed, unsigned *all_slow) {
	if (!c) return LSQ_E_ARG;
	if (tiles_handed) *tiles_handed = c->parse_tiles_handed;
	if (lines_listed) *lines_listed = c->parse_lines_listed;
	if (all_slow) *all_slow = c->parse_all_slow;
	return LSQ_OK;
}

int lsq_last_mrf_timing(lsq_ctx *c, float *h2d_ms, float *parse_ms) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (h2d_ms) *h2d_ms = c->mrf_h2d_ms;
	if (parse_ms) *parse_ms = c->mrf_parse_ms;
	return LSQ_OK;
} LSQ_API_CATCH

static const char *const INGEST_STAGE_NAMES[LSQ_INGEST_STAGES] = {
	"newline_count", "route", "partition_count", "partition_scatter", "group_classify", "group_offsets", "group_place"};
int lsq_ingest_stage_count(void) { return LSQ_INGEST_STAGES; }
const char *lsq_ingest_stage_name(int stage) { return stage >= 0 && stage < LSQ_INGEST_STAGES ? INGEST_STAGE_NAMES[stage] : nullptr; }
int lsq_last_ingest_stages(const lsq_ctx *c, float *ms, uint64_t *bytes, int capacity) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	for (int s = 0; s < LSQ_INGEST_STAGES && s < capacity; ++s) {
		if (ms) ms[s] = c->ing_seen[s] ? c->ing_ms[s] : 0.0f;
		if (bytes) bytes[s] = c->ing_seen[s] ? c->ing_bytes[s] : 0;
	}
	return LSQ_OK;
} LSQ_API_CATCH

uint64_t lsq_reads_retained(const lsq_ctx *c, int method) { return (c && method >= 0 && method < LSQ_MAX_METHODS) ? c->reads[method].n_retained : 0; }
uint64_t lsq_reads_pooled(const lsq_ctx *c, int method) {
	if (!c || method < 0 || method >= LSQ_MAX_METHODS) return 0;
	const MethodReads &mr = c->reads[method];
	return mr.n1_reads + mr.n2_reads + mr.pn_line.n;          // (total_slots also counts the padding of the groups)
}
uint64_t lsq_reads_pooled_blocks(const lsq_ctx *c, int method) {
	if (!c || method < 0 || method >= LSQ_MAX_METHODS) return 0;
	const MethodReads &mr = c->reads[method];
	return mr.n1_reads + 2 * mr.n2_reads + mr.pn_se.n / 2;
}
uint64_t lsq_reads_retained_blocks(const lsq_ctx *c, int method) { return (c && method >= 0 && method < LSQ_MAX_METHODS) ? c->reads[method].n_retained_blocks : 0; }

int lsq_reads_pool_format(const lsq_ctx *c, int method, int *compact, uint64_t *pool_bytes, uint64_t *pool_reads) LSQ_API_TRY {
	if (!c || method < 0 || method >= LSQ_MAX_METHODS) return fail(LSQ_E_ARG, "bad context or method");
	const MethodReads &mr = c->reads[method];
	if (!mr.present) return fail(LSQ_E_STATE, "no reads uploaded for method %d", method);
	if (compact) *compact = mr.compact ? 1 : 0;
	if (pool_bytes) *pool_bytes = 4ull * (mr.p1.n + mr.p2.n + mr.pn_se.n);
	if (pool_reads) { pool_reads[0] = mr.n1_reads; pool_reads[1] = mr.n2_reads; pool_reads[2] = mr.pn_line.n; }
	return LSQ_OK;
} LSQ_API_CATCH

} // extern "C"
