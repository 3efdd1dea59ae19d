// Device group of the C ABI: context, uploads, the count kernel and the EM kernel (gfx950).
//
// Data layout in HBM (one set per read file / "sampling method"):
//   pool 1  int2  (start,end)            one merged block   -- 8 B per read, the common case
//   pool 2  int4  (s0,e0,s1,e1)          two merged blocks  -- 16 B per read
//   pool n  u32 block offsets + int2     three or more
//   side arrays strand id (u8) and line number (u32): touched only on span-start ties
// every pool is ordered by bucket; a bucket is a coordinate range of one chromosome whose
// event tables (bin directory, 16-byte event records, segments, isoform masks) plus its
// class histogram fit one workgroup's LDS.
//
// count kernel: each workgroup owns a contiguous range of read slots (bucket-major), stages
// the bucket image into LDS, and for each read: bin lookup -> candidate events by span ->
// span-start tie rule (count/count.cpp:64-85,429-432) -> segment walk (common/read.h:204-274)
// -> contiguous-run compatibility per isoform (read.h:44-79) -> 0.98 validity (count.cpp:441)
// -> one LDS atomic on (event, class) carrying count and matched bases.  Per bucket the
// histogram is flushed with global atomics; integer sums make the result order-independent.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <type_traits>
#include <vector>

#include "lsq_internal.hpp"
#include "lsq_mrf_line.hpp"

using namespace lsq;

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t _e = (expr);                                                                \
		if (_e != hipSuccess) return fail(LSQ_E_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
	} while (0)

namespace {

#ifndef LSQ_COUNT_BLOCK
#define LSQ_COUNT_BLOCK 256
#endif
constexpr int COUNT_BLOCK = LSQ_COUNT_BLOCK;       // threads per workgroup of the count kernels
constexpr unsigned long long BASES_MASK = (1ull << 40) - 1;

// A (read, event) pair the fast kernel does not settle itself: span-start ties that need the
// strand/name order, two-block reads whose blocks touch, second looks that did not fit the LDS
// queue.  pool 0 = one-block pool, 1 = two-block pool; scan: continue with the following events.
struct ExcEntry {
	unsigned long long slot;       // index into the pool
	unsigned bucket;
	unsigned ev_pool_scan;         // event index in the bucket | pool << 29 (0 one block, 1 two blocks, 2 n blocks) | scan << 31
};

struct CountArgs {
	const BucketDesc *buckets;
	const unsigned char *images;
	const TieRec *ties;
	const unsigned char *strand_rank;
	// reads with their own names (solve's UCSC_GFF / UCSC_BED / WORMBASE_GFF3): name table of the method, and the
	// gene names in device event order; null for MRF reads, whose name is "read-<line>"
	const char *read_names; const unsigned long long *read_name_off;
	const char *gene_names; const unsigned *gene_name_off;
	unsigned n_buckets;
	unsigned ablate;                   // developer switch (LSQ_ABLATE): 1 skip per-read work, 2 skip LDS atomics, 4 skip flush, 8 skip record look
	unsigned tables_lds_bytes;         // LDS bytes reserved for the bucket image + histogram (16-byte multiple)
	const int2 *p1; const unsigned char *p1_strand; const unsigned *p1_line;
	const int4 *p2; const unsigned char *p2_strand; const unsigned *p2_line;
	const unsigned *pn_blk_off; const unsigned *pn_nblk; const int2 *pn_se; const unsigned char *pn_strand; const unsigned *pn_line; const unsigned *pn_bucket;
	const unsigned long long *p1_off, *p2_off, *pn_off, *slot_off;    // n_buckets + 1 each
	const unsigned *wg_first;          // per workgroup of the fast kernel's grid: the bucket its slot range starts in
	unsigned long long total_slots;
	unsigned long long n_pn;           // reads with three or more blocks
	unsigned n_workers;                // leading workgroups of the fast kernel's grid that take them
	unsigned long long *cnt, *bases;
	struct ExcEntry *exc;              // exception list (rare (read, event) pairs the fast kernel hands to the cleanup kernel)
	unsigned *exc_count;               // [0] entries appended, [1] set to 1 by the cleanup kernel when [0] > exc_cap
	unsigned exc_cap;
	unsigned long long *dbg;            // developer counters (LSQ_ABLATE & 256): parked one-block, parked two-block, walk steps, walk lanes
};

struct LdsView {
	const unsigned short *bins;
	const EventRec *ev;
	const int2 *segs;
	const unsigned *iso;
	unsigned long long *hist;
};

// Span-start tie (count/count.cpp:64-85): the read starts exactly at the event's first base and
// ends exactly at its last; it is a candidate unless (strand, name) orders it before the event.
// "read-<line>" < gene name is std::string operator< on the reference's read names.
__device__ __noinline__ bool tie_orders_read_first(const CountArgs &A, unsigned ev_index, unsigned read_strand, unsigned line) {
	const TieRec *t = A.ties + ev_index;
	const unsigned rs = A.strand_rank[read_strand], gs = A.strand_rank[t->strand_id];
	if (rs != gs) return rs < gs;
	if (A.read_name_off) {
		// named reads: `line` indexes the method's name table; std::string operator< against the gene name
		const unsigned long long r0 = A.read_name_off[line], r1 = A.read_name_off[line + 1];
		const unsigned g0 = A.gene_name_off[ev_index], g1 = A.gene_name_off[ev_index + 1];
		const unsigned long long rn = r1 - r0;
		const unsigned gn = g1 - g0;
		for (unsigned long long i = 0; i < rn && i < gn; ++i) {
			const unsigned char a = (unsigned char)A.read_names[r0 + i], b = (unsigned char)A.gene_names[g0 + i];
			if (a != b) return a < b;
		}
		return rn < gn;
	}
	const unsigned mode = t->tie_mode;
	if (mode != 2) return mode == 1;
	// compare the decimal digits of `line`, most significant first, with the name's tail
	unsigned pow10 = 1, nd = 1;
	while (nd < 10 && line / pow10 >= 10) { pow10 *= 10; ++nd; }
	const unsigned tl = t->tail_len;
	unsigned v = line;
	for (unsigned i = 0; i < nd && i < tl; ++i) {
		const unsigned char a = (unsigned char)('0' + v / pow10), b = (unsigned char)t->tail[i];
		if (a != b) return a < b;
		v %= pow10; pow10 /= 10;
	}
	return nd < tl;
}

// Segment walk of one read against one event's ascending segments (common/read.h:204-274).
// `pos` is the furthest matched coordinate (or the current segment's start), `it` the segment
// cursor, which never moves back.  The first block may start anywhere inside a segment; once
// something has matched, every continuation must start exactly at `pos`.
struct Walk {
	int pos = 0, it = 0;
	bool found = false;
	unsigned mask = 0;
	int matched = 0;
	// returns false when the walk must stop (block not fully consumed)
	__device__ inline bool block(const int2 *segs, int nseg, int a, int b) {
		while (it < nseg) {
			const int2 sg = segs[it];
			if (!(sg.x < b)) break;
			pos = max(pos, sg.x);
			if (a >= pos && a < sg.y) {
				if (found && a > pos) break;
				found = true;
				mask |= 1u << it;
				pos = min(sg.y, b);
				matched += pos - a;
				if (b < sg.y) { a = b; break; }
				a = (b == sg.y) ? b : sg.y;
			} else if (pos > sg.x && pos < sg.y) {
				break;
			}
			++it;
		}
		return a == b;
	}
};

// One read against the staged bucket.  NB = 1 / 2: blocks in registers (v.x,v.y[,v.z,v.w]);
// NB = 0: nblk blocks at blk[].  p = first merged start, q = last merged end.
template <int NB>
__device__ inline void process_read(const LdsView &L, const BucketDesc &d, const CountArgs &A, const int4 v,
                                    const int2 *blk, int nblk, int total,
                                    const unsigned char *strand_arr, const unsigned *line_arr, unsigned long long slot) {
	const int p = v.x, q = (NB == 1) ? v.y : v.w;
	const int rel = p - d.lo;       // both within +-2^30
	unsigned bin = rel <= 0 ? 0u : ((unsigned)rel >> d.shift);
	bin = min(bin, d.n_bins - 1u);
	for (unsigned i = L.bins[bin]; i < d.n_events; ++i) {
		const EventRec e = L.ev[i];
		if (e.gs > p) break;
		if (p > e.ge) continue;
		if (p == e.gs) {
			// reads ordered before the key (chrom, gene_start, gene_end, strand, name) are not candidates
			if (q < e.ge) continue;
			if (q == e.ge && tie_orders_read_first(A, d.ev_base + i, strand_arr[slot], line_arr[slot])) continue;
		}
		Walk w;
		const int2 *segs = L.segs + e.seg_off;
		if (NB == 1) {
			w.block(segs, e.nseg, v.x, v.y);
		} else if (NB == 2) {
			if (w.block(segs, e.nseg, v.x, v.y)) w.block(segs, e.nseg, v.z, v.w);
		} else {
			for (int k = 0; k < nblk; ++k) { const int2 bk = blk[k]; if (!w.block(segs, e.nseg, bk.x, bk.y)) break; }
		}
		const unsigned mask = w.mask;
		if (!mask) continue;
		// (double)matched / total > 0.98  <=>  50*matched > 49*total for these magnitudes
		if (!(50ll * w.matched > 49ll * total)) continue;
		const unsigned hi = 31u - (unsigned)__clz((int)mask), lo = (unsigned)__ffs((int)mask) - 1u;
		const unsigned span = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
		unsigned cls = 0;
		for (unsigned j = 0; j < e.K; ++j) {
			const unsigned iso = L.iso[e.iso_off + j];
			if ((mask & ~iso) == 0 && (iso & span) == mask) cls |= 1u << j;
		}
		if (cls) atomicAdd(&L.hist[e.cls_off + cls - 1], (1ull << 40) | (unsigned long long)(unsigned)w.matched);
	}
}

// =====================================================================================
// Generic kernel: buckets whose events do not fit the packed record (more than 4 segments or
// isoforms, negative coordinates).  One lane per read, branching walk, reads straight from
// global memory.  Correct for everything; not tuned.
// =====================================================================================
__global__ void __launch_bounds__(COUNT_BLOCK) lsq_count_generic_kernel(CountArgs A) {
	extern __shared__ __align__(16) unsigned char lds[];
	const unsigned tid = threadIdx.x;
	const unsigned long long s_begin = A.total_slots * blockIdx.x / gridDim.x;
	const unsigned long long s_end = A.total_slots * (blockIdx.x + 1ull) / gridDim.x;
	if (s_begin >= s_end) return;
	unsigned lo_b = 0, hi_b = A.n_buckets;
	while (hi_b - lo_b > 1) {
		unsigned mid = (lo_b + hi_b) >> 1;
		if (A.slot_off[mid] <= s_begin) lo_b = mid; else hi_b = mid;
	}
	for (unsigned b = lo_b; b < A.n_buckets && A.slot_off[b] < s_end; ++b) {
		const unsigned long long bs = A.slot_off[b], be = A.slot_off[b + 1];
		if (be <= s_begin || be == bs) continue;
		const BucketDesc d = A.buckets[b];
		if (d.kind != 0) continue;
		{
			const uint4 *src = reinterpret_cast<const uint4 *>(A.images + d.img_off);
			uint4 *dst = reinterpret_cast<uint4 *>(lds);
			for (unsigned i = tid; i < d.img_bytes / 16; i += COUNT_BLOCK) dst[i] = src[i];
			unsigned long long *h = reinterpret_cast<unsigned long long *>(lds + d.hist_off);
			for (unsigned i = tid; i < d.n_cls; i += COUNT_BLOCK) h[i] = 0;
		}
		__syncthreads();
		LdsView L;
		L.bins = reinterpret_cast<const unsigned short *>(lds);
		L.ev = reinterpret_cast<const EventRec *>(lds + d.ev_off);
		L.segs = reinterpret_cast<const int2 *>(lds + d.seg_off);
		L.iso = reinterpret_cast<const unsigned *>(lds + d.iso_off);
		L.hist = reinterpret_cast<unsigned long long *>(lds + d.hist_off);
		const unsigned long long l0 = (s_begin > bs ? s_begin : bs) - bs;
		const unsigned long long l1 = (s_end < be ? s_end : be) - bs;
		const unsigned long long n1 = A.p1_off[b + 1] - A.p1_off[b];
		const unsigned long long n2 = A.p2_off[b + 1] - A.p2_off[b];
		for (unsigned long long i = l0 + tid; i < l1; i += COUNT_BLOCK) {
			if (i < n1) {
				const unsigned long long g = A.p1_off[b] + i;
				const int2 rd = A.p1[g];
				process_read<1>(L, d, A, make_int4(rd.x, rd.y, 0, 0), nullptr, 1, rd.y - rd.x, A.p1_strand, A.p1_line, g);
			} else if (i < n1 + n2) {
				const unsigned long long g = A.p2_off[b] + (i - n1);
				const int4 rd = A.p2[g];
				process_read<2>(L, d, A, rd, nullptr, 2, (rd.y - rd.x) + (rd.w - rd.z), A.p2_strand, A.p2_line, g);
			} else {
				const unsigned long long g = A.pn_off[b] + (i - n1 - n2);
				const unsigned o0 = A.pn_blk_off[g], o1 = o0 + A.pn_nblk[g];
				const int2 *blk = A.pn_se + o0;
				int total = 0;
				for (unsigned k = o0; k < o1; ++k) { int2 v = A.pn_se[k]; total += v.y - v.x; }
				process_read<0>(L, d, A, make_int4(blk[0].x, 0, 0, A.pn_se[o1 - 1].y), blk, (int)(o1 - o0), total, A.pn_strand, A.pn_line, g);
			}
		}
		__syncthreads();
		for (unsigned i = tid; i < d.n_cls; i += COUNT_BLOCK) {
			unsigned long long v = L.hist[i];
			if (v) {
				atomicAdd(&A.cnt[d.cls_base + i], v >> 40);
				atomicAdd(&A.bases[d.cls_base + i], v & BASES_MASK);
			}
		}
		__syncthreads();
	}
}

// =====================================================================================
// Fast kernel: buckets of packed 48-byte FastRec events (every LESSeq local-event shape).
// =====================================================================================

// 0/1 integer predicates kept in vector registers: combining them with & and | costs VALU ops,
// where bool && / || on 64-lane masks would go through the CU's single scalar unit.
__device__ inline int nonneg(int x) { return (int)(~(unsigned)x >> 31); }
__device__ inline int inside01(int a, int sx, int sy) { return nonneg((a - sx) | (sy - 1 - a)); }   // sx <= a < sy
__device__ inline int gt01(int b, int sy) { return (int)((unsigned)(sy - b) >> 31); }                  // b > sy

// r_k says segment k is matched by the current block.  A run extends from segment k to k+1 only
// when k+1 starts where k ends (abut bit k) and the block goes past k's end.  Returns the run
// bits; end_run = end of the last matched segment.
__device__ inline unsigned run_bits(const int (&sy)[4], unsigned abut, int r0, int r1, int r2, int r3, int b, int &end_run) {
	r1 |= r0 & (int)(abut & 1u) & gt01(b, sy[0]);
	r2 |= r1 & (int)((abut >> 1) & 1u) & gt01(b, sy[1]);
	r3 |= r2 & (int)((abut >> 2) & 1u) & gt01(b, sy[2]);
	end_run = max(max(r0 ? sy[0] : 0, r1 ? sy[1] : 0), max(r2 ? sy[2] : 0, r3 ? sy[3] : 0));
	return (unsigned)(r0 | (r1 << 1) | (r2 << 2) | (r3 << 3));
}

struct FastCtx {                       // wave-uniform state of the bucket being processed
	const uint4 *bins;                 // 16-byte bin records: first cell | first event << 16, ends of that cell and the next two
	int lo; unsigned shift, n_bins;    // bin of p: (p - lo) >> shift, clamped
	const uint4 *recs;
	unsigned long long *hist;
	unsigned n_events, bucket;
	unsigned long long slot0;          // pool index of the first read of this workgroup's range in the bucket
	unsigned pool;
	ExcEntry *exc;
	unsigned *exc_count;
	unsigned exc_cap;
	unsigned ablate;
	unsigned long long *dbg;
};

__device__ inline void emit_exception(const FastCtx &C, unsigned r, unsigned i, unsigned scan) {
	const unsigned slot = atomicAdd(C.exc_count, 1u);
	if (slot < C.exc_cap) {
		ExcEntry e;
		e.slot = C.slot0 + r;
		e.bucket = C.bucket;
		e.ev_pool_scan = i | (C.pool << 29) | (scan << 31);
		C.exc[slot] = e;
	}
}

constexpr unsigned PARK_EVENT_UNKNOWN = 0x7FFFFFFFu;
// A read whose first base lies in a cell with one owner can only ever count for that owner: every
// other event whose span covers the base has no segment there, so the read's first block starts in
// none of its segments and nothing matches (common/read.h:204-274).  Such a read is parked with
// the owner as the event to look at and this flag: one look, no scan of the following events.
constexpr unsigned PARK_ONE_EVENT = 0x80000000u;

// sum over the 64 lanes (DPP row shifts and row broadcasts; the total lands in lane 63)
__device__ inline unsigned wave_sum_u32(unsigned v) {
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
	return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// One read against ONE packed event record (index i).  Returns true when a further event has
// to be examined for this read: the bin's first event ended left of the read, or this event's
// span is overlapped by the next one.  No calls, no data-dependent loops: the two cases that
// need them (a span-start tie that falls through to the strand/name order; two blocks that
// touch) are written to the exception list instead.
template <int NB>
__device__ inline bool fast_trip(const FastCtx &C, const int4 v, const int total, const unsigned r, const unsigned i, const bool valid) {
	const int p = v.x, q = (NB == 1) ? v.y : v.w;
	const bool inb = valid && i < C.n_events;
	const unsigned ri = 3u * (inb ? i : 0u);
	const uint4 w0 = C.recs[ri], w1 = C.recs[ri + 1], w2 = C.recs[ri + 2];
	const int gs = (int)w1.x, ge = (int)w0.x;
	const bool started = inb && gs <= p;
	bool covers = started && p <= ge;
	// span-start tie rule (count/count.cpp:64-85): a read that starts on the event's first base is a
	// candidate only if it is not ordered before (gene_start, gene_end, strand, name)
	bool exc = covers && p == gs && q == ge;
	covers = covers && !(p == gs && q <= ge);
	if (NB == 2) {
		const bool touching = covers && v.z == v.y;
		exc = exc || touching;
		covers = covers && !touching;
	}
	if (exc && !(C.ablate & 128u)) emit_exception(C, r, i, 0u);
	const int sx[4] = {(int)w1.x, (int)w1.z, (int)w2.x, (int)w2.z};
	const int sy[4] = {(int)w1.y, (int)w1.w, (int)w2.y, (int)w2.w};
	const unsigned abut = (w0.y >> FAST_ABUT_SHIFT) & 7u;
	// block 1 starts the match: the segment that holds its first base
	int end1;
	const unsigned m1 = run_bits(sy, abut, inside01(v.x, sx[0], sy[0]), inside01(v.x, sx[1], sy[1]),
	                             inside01(v.x, sx[2], sy[2]), inside01(v.x, sx[3], sy[3]), v.y, end1);
	int matched = m1 ? min(v.y, end1) - v.x : 0;
	unsigned mask = m1;
	if (NB == 2) {
		// block 2 continues only if block 1 ended exactly on a segment end, and must then start
		// exactly on the start of a later segment
		const int exact1 = (m1 != 0 && v.y == end1) ? 1 : 0;
		const int l3 = (int)(~m1 >> 3) & 1, l2 = l3 & (int)(~m1 >> 2) & 1, l1 = l2 & (int)(~m1 >> 1) & 1;   // no matched segment at index >= k
		int end2;
		const unsigned m2 = run_bits(sy, abut, 0, exact1 & l1 & (int)(sx[1] == v.z), exact1 & l2 & (int)(sx[2] == v.z),
		                             exact1 & l3 & (int)(sx[3] == v.z), v.w, end2);
		matched += m2 ? min(v.w, end2) - v.z : 0;
		mask |= m2;
	}
	// (double)matched / total > 0.98  <=>  50*matched > 49*total (both below 2^18 here)
	const unsigned long long tbl = ((unsigned long long)w0.w << 32) | w0.z;
	const unsigned cls = (unsigned)(tbl >> (4u * mask)) & 0xFu;
	{
		const bool add = covers && cls != 0 && 50 * matched > 49 * total;
		if (!(C.ablate & 2u)) { if (add) atomicAdd(&C.hist[(w0.y & 0xFFFFu) + cls - 1u], (1ull << 40) | (unsigned long long)(unsigned)matched); }
		else asm volatile("" ::"v"(matched), "v"(cls));
	}
	return started && (!(p <= ge) || (w0.y & FAST_FLAG_OVERLAPS_NEXT)) && i + 1 < C.n_events;
}

// Parked reads.  The streaming loop settles the commonest shapes with one or two table looks
// (cells); every other read is parked -- its blocks, the event to start at, its position in the
// workgroup's range -- and the general walk runs over the parked reads a full wave at a time,
// instead of stalling a 64-lane wave on its hardest lane.
//
// Each wave streams its own part of the workgroup's range straight from HBM into registers
// (next words in flight while the current ones are processed) and owns its parking area: no
// workgroup barrier inside the stream, a slow wave never holds up the others.
#ifndef LSQ_STREAM_WORDS
#define LSQ_STREAM_WORDS 2
#endif
constexpr int STREAM_WORDS = LSQ_STREAM_WORDS;                // 16-byte words per lane in flight
constexpr int GROUP_WORDS = 2;                                // words per lane looked up together (independent chains)
constexpr unsigned WAVE_QUEUE_WORDS = 256;                    // 16-byte words of parking per wave (4 KiB): 63 left over + what is pushed between two walks
constexpr unsigned WAVES = COUNT_BLOCK / 64;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *global_words;

__device__ inline void wave_sync_lds() {
	// LDS operations of one wave complete in order; this only keeps the compiler from moving
	// accesses of other lanes' data across the point
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// The parking area is a ring of reads waiting for their next look: (blocks, event to look at,
// position in the range).  The general walk takes 64 of them at a time -- every lane busy, one
// event record each -- and a read that needs a further event goes back to the tail.  While the
// stream is running the walk only runs on full waves; what is left stays for the next time.
template <int NB>
struct Ring {
	static constexpr unsigned CAP = WAVE_QUEUE_WORDS / NB;      // entries
	uint4 *q;
	unsigned head = 0, tail = 0;                                 // running counters (the same in every lane)
	__device__ inline unsigned live() const { return tail - head; }
	__device__ inline void push(bool want, unsigned lane, const uint4 e0, const uint4 e1) {
		const unsigned long long m = __ballot(want);
		if (!m) return;          // wave-uniform: the reads that need parking sit together in the start-ordered pools, most steps park nothing
		const unsigned at = (tail + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) % CAP;
		if (want) {
			if (NB == 1) q[at] = e0;
			else { q[2 * at] = e0; q[2 * at + 1] = e1; }
		}
		tail += (unsigned)__popcll(m);
	}
};

template <int NB>
__device__ inline void walk_parked(const FastCtx &C, Ring<NB> &R, const bool to_empty) {
	const unsigned lane = threadIdx.x & 63u;
	wave_sync_lds();
#pragma unroll 1
	while (R.live() >= (to_empty ? 1u : 64u)) {
		const unsigned n = min(R.live(), 64u);
		const bool on = lane < n;
		if ((C.ablate & 256u) && lane == 0) { atomicAdd(&C.dbg[2], 1ull); atomicAdd(&C.dbg[3], (unsigned long long)n); }
		const unsigned at = (R.head + (on ? lane : 0u)) % Ring<NB>::CAP;
		uint4 e0, e1 = make_uint4(0, 0, 0, 0);
		if (NB == 1) e0 = R.q[at];
		else { e0 = R.q[2 * at]; e1 = R.q[2 * at + 1]; }
		R.head += n;
		int4 rd; unsigned i, rel;
		const unsigned ev_word = NB == 1 ? e0.z : e1.x;
		const bool one_event = (ev_word & PARK_ONE_EVENT) != 0;
		if (NB == 1) { e0.z &= ~PARK_ONE_EVENT; } else { e1.x &= ~PARK_ONE_EVENT; }
		if (NB == 1) {
			rd = make_int4((int)e0.x, (int)e0.y, (int)e0.x, (int)e0.y); i = e0.z; rel = e0.w;
			// parked without a look at the bin directory: the first event of the read's bin
			const int brel = rd.x - C.lo;
			const unsigned bin = brel <= 0 ? 0u : min((unsigned)brel >> C.shift, C.n_bins - 1u);
			const unsigned first = reinterpret_cast<const unsigned *>(C.bins)[4u * bin] >> 16;
			if (i == PARK_EVENT_UNKNOWN) { i = first; e0.z = first; }
		}
		else { rd = make_int4((int)e0.x, (int)e0.y, (int)e0.z, (int)e0.w); i = e1.x; rel = e1.y; }
		const int total = NB == 1 ? rd.y - rd.x : (rd.y - rd.x) + (rd.w - rd.z);
		const bool more = fast_trip<NB>(C, rd, total, rel, i, on) && !one_event && !(C.ablate & 64u);
		wave_sync_lds();
		if (NB == 1) e0.z = i + 1u; else e1.x = i + 1u;     // (i is the resolved event)
		R.push(more, lane, e0, e1);
		wave_sync_lds();
	}
}

// RPW = reads per 16-byte word: 2 (pool 1: one block) or 1 (pool 2: two blocks)
template <int RPW>
__device__ inline void stream_pool_fast(FastCtx &C, const uint4 *bins, const uint4 *cells, const unsigned *cell_info, const unsigned n_cells, const BucketDesc &d,
                                        const CountArgs &A, uint4 *queue, const uint4 *src_generic,
                                        const unsigned long long g0, const unsigned long long g1) {
	constexpr int NB = RPW == 2 ? 1 : 2;
	constexpr unsigned TILE = 64u * STREAM_WORDS;        // words per wave step
	C.pool = RPW == 2 ? 0u : 1u;
	C.slot0 = g0;
	global_words src = (global_words)src_generic;       // kernel-argument memory: global address space
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	// words [w0, w1) of the workgroup, dealt to its waves a step at a time (wave, wave + 4, ...): the
	// reads that need the general walk sit together in the start-ordered pool, and a contiguous
	// quarter per wave would leave three waves waiting for the one that got them
	const unsigned long long w0 = g0 / RPW, w1 = (g1 + RPW - 1) / RPW;
	const unsigned n_words = (unsigned)(w1 - w0);                                   // a workgroup's range stays below 2^21 reads
	const unsigned ww0 = min(wave * TILE, n_words), ww1 = n_words;                   // relative to w0
	const unsigned first_rel = (unsigned)(g0 - w0 * RPW);                            // 0 or 1: reads of word w0 before the range
	const unsigned n_rel = (unsigned)(g1 - g0);
	uint4 nxt[STREAM_WORDS];
	auto fetch_into = [&](uint4 (&dst)[STREAM_WORDS], unsigned wt) {
#pragma unroll
		for (int k = 0; k < STREAM_WORDS; ++k) {
			const unsigned w = wt + lane * (unsigned)STREAM_WORDS + (unsigned)k;      // a lane's words are neighbours in the pool
			u32x4 t = {0u, 0u, 0u, 0u};
			if (w < ww1) t = src[w0 + w];
			dst[k] = make_uint4(t.x, t.y, t.z, t.w);
		}
	};
	auto fetch = [&](unsigned wt) { fetch_into(nxt, wt); };
	// bin record of position p -> (cell that can hold p, first event of the bin)
	auto locate = [&](int p, unsigned &cell, unsigned &first_event) {
		const int rel = p - d.lo;
		unsigned bin = rel <= 0 ? 0u : ((unsigned)rel >> d.shift);
		const uint4 br = bins[min(bin, d.n_bins - 1u)];     // first cell | first event << 16, ends of that cell and the next two
		cell = (br.x & 0xFFFFu) + (unsigned)(p >= (int)br.y) + (unsigned)(p >= (int)br.z) + (unsigned)(p >= (int)br.w);
		first_event = br.x >> 16;
	};
	Ring<NB> R;
	R.q = queue;
	if (ww0 < ww1) fetch(ww0);
	for (unsigned wt = ww0; wt < ww1; wt += WAVES * TILE) {
		uint4 cur[STREAM_WORDS];
#pragma unroll
		for (int k = 0; k < STREAM_WORDS; ++k) cur[k] = nxt[k];
		if (wt + WAVES * TILE < ww1) fetch(wt + WAVES * TILE);
#pragma unroll
		for (int k0 = 0; k0 < STREAM_WORDS; k0 += GROUP_WORDS) {
		if (A.ablate & 512u) {      // developer switch: stream only
#pragma unroll
			for (int kg = 0; kg < GROUP_WORDS; ++kg) asm volatile("" ::"v"(cur[k0 + kg].x), "v"(cur[k0 + kg].y), "v"(cur[k0 + kg].z), "v"(cur[k0 + kg].w));
			continue;
		}
		// the reads of a group are looked up first (independent chains), parking comes after
		constexpr int N_READS = GROUP_WORDS * RPW;
		bool park[N_READS];
		uint4 pe0[N_READS], pe1[N_READS];
#pragma unroll
		for (int kg = 0; kg < GROUP_WORDS; ++kg) {
			const int k = k0 + kg;
			const unsigned w = wt + lane * (unsigned)STREAM_WORDS + (unsigned)k;           // word, relative to w0
			if (RPW == 2) {
				// One look at the tables per lane and group: the lane's reads are neighbours in the
				// start-ordered pool, so the cell of the first one is the cell of (nearly) all of
				// them.  A read is decided against that cell -- inside it: the owners' slots; running
				// into the owner's next segment: the two-segment slot -- and the lane adds its totals
				// once.  Everything else (a different cell, no cell, a longer run) is parked.
				if (kg == 0) {
					unsigned ci, evf;
					locate((int)cur[k0].x, ci, evf);
					const uint4 cw = cells[min(ci, n_cells - 1u)];           // lo, hi, hi2, slots
					const bool has = ci < n_cells && !(A.ablate & 8u);
					const unsigned info = cell_info[min(ci, n_cells - 1u)];
					const unsigned owner_word = info == CELL_INFO_SHARED ? PARK_EVENT_UNKNOWN : ((info >> 8) | PARK_ONE_EVENT);
					const int lo = (int)cw.x, hi = (int)cw.y, hi2 = (int)cw.z;
					const unsigned width = has ? (unsigned)(hi - lo) : 0u;
					unsigned nA = 0, sA = 0, nX = 0, sX = 0;
					// all but the first and last steps of a workgroup's range lie wholly inside it: no per-read range test there
					const bool interior = wt + TILE <= ww1 && (wt > 0u || first_rel == 0u) && (wt + TILE) * 2u - first_rel <= n_rel;
					auto decide = [&](auto whole_step) {
#pragma unroll
						for (int j = 0; j < N_READS; ++j) {
							const int kk = k0 + j / 2;
							const int ra = (j & 1) ? (int)cur[kk].z : (int)cur[kk].x, rb = (j & 1) ? (int)cur[kk].w : (int)cur[kk].y;
							const unsigned wj = wt + lane * (unsigned)STREAM_WORDS + (unsigned)kk;
							const unsigned rel = wj * 2u + (unsigned)(j & 1) - first_rel;      // position in the range (wraps above n_rel when outside)
							const bool in = decltype(whole_step)::value || (wj < ww1 && rel < n_rel);
							const bool m = in && (unsigned)(ra - lo) < width;
							const bool a = m && rb <= hi;
							const bool x = m && !a && rb <= hi2;
							const unsigned len = (unsigned)(rb - ra);
							nA += a ? 1u : 0u; sA += a ? len : 0u;
							nX += x ? 1u : 0u; sX += x ? len : 0u;
							park[j] = in && !a && !x && !(A.ablate & 17u);
							if ((A.ablate & 256u) && park[j]) atomicAdd(&A.dbg[5 + (m ? (info == CELL_INFO_SHARED ? 2 : 1) : 0)], 1ull);
							pe0[j] = make_uint4((unsigned)ra, (unsigned)rb, m ? owner_word : PARK_EVENT_UNKNOWN, rel);
							pe1[j] = make_uint4(0, 0, 0, 0);
						}
					};
					if (interior) decide(std::true_type{}); else decide(std::false_type{});
					const unsigned sa = cw.w & 0xFFFFu, sb = cw.w >> 16;
					if (!(A.ablate & (1u | 16384u))) {
						const unsigned long long addA = ((unsigned long long)nA << 40) | sA;
						if (nA && sa != CELL_NONE) atomicAdd(&C.hist[sa], addA);
						if (nA && hi2 == hi && sb != CELL_NONE) atomicAdd(&C.hist[sb], addA);        // second owner of the cell
						if (nX && sb != CELL_NONE) atomicAdd(&C.hist[sb], ((unsigned long long)nX << 40) | sX);
					} else asm volatile("" ::"v"(nA), "v"(sA), "v"(nX), "v"(sX));
				}
			} else {
				const uint4 u = cur[k];
				const int4 rd = make_int4((int)u.x, (int)u.y, (int)u.z, (int)u.w);
				const unsigned rel = w - first_rel;
				const bool in = w < ww1 && rel < n_rel;
				unsigned c1, c2, evf, evf2;
				locate(rd.x, c1, evf);
				locate(rd.z, c2, evf2);
				// the usual junction read: block 1 runs to the end of one segment, block 2 starts on the
				// first base of a later segment of the same event and ends inside it
				const uint4 cw1 = cells[min(c1, n_cells - 1u)], cw2 = cells[min(c2, n_cells - 1u)];
				const unsigned i1 = cell_info[min(c1, n_cells - 1u)], i2 = cell_info[min(c2, n_cells - 1u)];
				const bool hit = in && c1 < n_cells && c2 < n_cells && i1 != CELL_INFO_SHARED && i2 != CELL_INFO_SHARED &&
				                 (int)cw1.x <= rd.x && rd.y == (int)cw1.y && (i1 & 2u) &&          // block 1 ends on its segment's end
				                 rd.z == (int)cw2.x && (i2 & 1u) && rd.w <= (int)cw2.y &&          // block 2 starts on its segment's start
				                 (i1 >> 8) == (i2 >> 8) && ((i2 >> 2) & 0x3Fu) > ((i1 >> 2) & 0x3Fu) && !(A.ablate & 8u);
				if (!(A.ablate & 1u)) {
					const unsigned ev = hit ? i1 >> 8 : 0u;
					const uint4 w0r = C.recs[3u * ev];
					const unsigned long long tbl = ((unsigned long long)w0r.w << 32) | w0r.z;
					const unsigned mask = (1u << ((i1 >> 2) & 0x3u)) | (1u << ((i2 >> 2) & 0x3u));
					const unsigned cls = (unsigned)(tbl >> (4u * mask)) & 0xFu;
					if (hit && cls != 0) atomicAdd(&C.hist[(w0r.y & 0xFFFFu) + cls - 1u], (1ull << 40) | (unsigned long long)(unsigned)((rd.y - rd.x) + (rd.w - rd.z)));
				}
				park[kg] = in && !hit && !(A.ablate & 17u);
				pe0[kg] = u;
				const bool owned1 = c1 < n_cells && i1 != CELL_INFO_SHARED && (int)cw1.x <= rd.x && rd.x < (int)cw1.y;
				pe1[kg] = make_uint4(owned1 ? ((i1 >> 8) | PARK_ONE_EVENT) : evf, rel, 0u, 0u);
			}
		}
#pragma unroll
		for (int q = 0; q < N_READS; ++q) {
			if ((A.ablate & 256u) && park[q]) atomicAdd(&A.dbg[NB - 1], 1ull);
			R.push(park[q], lane, pe0[q], pe1[q]);
			// the ring holds what one walk leaves behind (< 64) plus 128 one-block or 64 two-block entries
			if ((NB == 2 || (q & 1) == 1) && R.live() >= 64u) {             // wave-uniform
				if (!(A.ablate & 32u)) walk_parked<NB>(C, R, false);
				else R.head = R.tail;
			}
		}
		}
	}
	if (R.live() && !(A.ablate & 32u)) walk_parked<NB>(C, R, true);
}

// the bucket each workgroup of the fast kernel starts in: a dependent chain of a dozen global loads
// per workgroup, done once per read set and grid instead of at the head of every launch
__global__ void __launch_bounds__(256) lsq_wg_plan_kernel(const unsigned long long *slot_off, unsigned n_buckets, unsigned long long total_slots,
                                                          unsigned grid, unsigned *wg_first) {
	const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= grid) return;
	const unsigned long long s_begin = total_slots * g / grid;
	unsigned lo_b = 0, hi_b = n_buckets;
	while (hi_b - lo_b > 1) {
		const unsigned mid = (lo_b + hi_b) >> 1;
		if (slot_off[mid] <= s_begin) lo_b = mid; else hi_b = mid;
	}
	wg_first[g] = lo_b;
}

// Global count/bases adds of a whole wave, merged by class before they reach L2: with skewed read
// depth most lanes of a worker wave hit the classes of one hot event, and atomics on one address
// run one after the other.  Up to four distinct classes are summed across the wave (ballot, DPP
// sum, one atomic pair each); what is left adds lane by lane.  Every lane of the wave must call.
__device__ inline void global_add_merged(unsigned long long *cnt, unsigned long long *bases, bool want, const unsigned slot, const unsigned matched) {
	const unsigned lane = threadIdx.x & 63u;
#pragma unroll 1
	for (int round = 0; round < 4; ++round) {
		const unsigned long long m = __ballot(want);
		if (!m) return;
		const unsigned lead = (unsigned)__ffsll((long long)m) - 1u;
		const unsigned s0 = (unsigned)__builtin_amdgcn_readlane((int)slot, lead);
		const bool same = want && slot == s0;
		const unsigned n = (unsigned)__popcll(__ballot(same));
		const unsigned sum = wave_sum_u32(same ? matched : 0u);       // reads are shorter than 2^18 bases
		if (lane == lead) { atomicAdd(&cnt[s0], (unsigned long long)n); atomicAdd(&bases[s0], (unsigned long long)sum); }
		want = want && !same;
	}
	if (want) { atomicAdd(&cnt[slot], 1ull); atomicAdd(&bases[slot], (unsigned long long)matched); }
}

// Reads with three or more blocks (about 1 % of a typical read set), inside the fast kernel's grid:
// the first `n_workers` workgroups take them a lane each, tables read from global memory (L2), global
// atomics -- latency-bound work that runs beside the streaming workgroups instead of in a kernel of
// its own after them.  Same evaluation as the cleanup kernel's (candidate window, span-start rule,
// branching segment walk); a span-start tie that needs the strand/name order goes to the exception
// list.  No local arrays, no calls: the kernel keeps a zero-byte private segment.
__device__ inline void pool_n_worker(const CountArgs &A, const unsigned long long n_pn, const unsigned n_workers) {
	const unsigned long long gsz = (unsigned long long)n_workers * COUNT_BLOCK;
	// wave-uniform loops (every lane takes every trip, idle or not): the merged adds need the whole wave
	for (unsigned long long g0 = (unsigned long long)blockIdx.x * COUNT_BLOCK; g0 < n_pn; g0 += gsz) {
		const unsigned long long g = g0 + threadIdx.x;
		bool active = g < n_pn;
		const unsigned b = active ? A.pn_bucket[g] : 0u;
		const BucketDesc *d = A.buckets + b;
		active = active && d->kind == 1;
		const unsigned *bins = reinterpret_cast<const unsigned *>(A.images + d->img_off);
		const uint4 *recs = reinterpret_cast<const uint4 *>(A.images + d->img_off + d->ev_off);
		const int2 *blk = A.pn_se + (active ? A.pn_blk_off[g] : 0u);
		const int nblk = active ? (int)A.pn_nblk[g] : 1;
		int p = 0, q = 0, total = 0;
		unsigned i = 0;
		if (active) {
			p = blk[0].x; q = blk[nblk - 1].y;
			for (int k = 0; k < nblk; ++k) total += blk[k].y - blk[k].x;
			const int rel = p - d->lo;
			const unsigned bin = rel <= 0 ? 0u : min((unsigned)rel >> d->shift, d->n_bins - 1u);
			i = bins[4u * bin] >> 16;
		}
		while (__any(active)) {
			bool want = false;
			unsigned slot = 0, matched = 0;
			if (active) {
				if (i >= d->n_events) active = false;
				else {
					const uint4 w0 = recs[3u * i];
					const int2 *segs = reinterpret_cast<const int2 *>(recs + 3u * i + 1u);     // four (start, end) pairs
					const int gs = segs[0].x, ge = (int)w0.x;
					if (gs > p) active = false;
					else {
						bool cand = p <= ge;
						if (cand && p == gs) {
							if (q == ge) {
								const unsigned at = atomicAdd(A.exc_count, 1u);
								if (at < A.exc_cap) { ExcEntry e; e.slot = g; e.bucket = b; e.ev_pool_scan = i | (2u << 29); A.exc[at] = e; }
							}
							cand = q > ge;          // q < ge: ordered before the event; q == ge: the cleanup kernel decides
						}
						if (cand) {
							const int nseg = (int)((w0.y >> FAST_NSEG_SHIFT) & 7u);
							Walk w;
							for (int k = 0; k < nblk; ++k) { const int2 bk = blk[k]; if (!w.block(segs, nseg, bk.x, bk.y)) break; }
							const unsigned long long tbl = ((unsigned long long)w0.w << 32) | w0.z;
							const unsigned cls = w.mask < 16u ? (unsigned)(tbl >> (4u * w.mask)) & 0xFu : 0u;
							if (cls != 0 && 50ll * w.matched > 49ll * total) { want = true; slot = d->cls_base + (w0.y & 0xFFFFu) + cls - 1; matched = (unsigned)w.matched; }
						}
						if (p <= ge && !(w0.y & FAST_FLAG_OVERLAPS_NEXT)) active = false;
						++i;
					}
				}
			}
			global_add_merged(A.cnt, A.bases, want, slot, matched);
		}
	}
}

#ifndef LSQ_FAST_WAVES
#define LSQ_FAST_WAVES 1
#endif
// What a workgroup needs to know about one bucket visit; found with scalar loads, kept in LDS
// beside the bucket's tables while the bucket before it is still being streamed.
struct BucketVisit {
	BucketDesc d;
	unsigned long long bs, be;            // the bucket's slots
	unsigned long long p1o, p1n, p2o, p2n;   // first read and count of its one- and two-block pools
	unsigned b, valid;
};
constexpr unsigned VISIT_LDS_BYTES = 128;
static_assert(sizeof(BucketVisit) <= VISIT_LDS_BYTES, "BucketVisit has a fixed LDS slot");

// next packed bucket at or after b that holds slots of [s_begin, s_end); n_buckets when there is none
__device__ inline unsigned find_bucket(const CountArgs &A, unsigned b, const unsigned long long s_begin, const unsigned long long s_end) {
	for (b = (unsigned)__builtin_amdgcn_readfirstlane((int)b); b < A.n_buckets; ++b) {
		const unsigned long long bs = A.slot_off[b], be = A.slot_off[b + 1];
		const unsigned kind = A.buckets[b].kind;
		if (bs >= s_end) return A.n_buckets;
		if (be <= s_begin || be == bs || kind != 1) continue;
		return b;
	}
	return A.n_buckets;
}

// tables of bucket b into an LDS buffer: the image, a cleared histogram, the visit record
__device__ inline void stage_bucket(const CountArgs &A, const unsigned b, unsigned char *buf) {
	const unsigned tid = threadIdx.x;
	unsigned *rec = reinterpret_cast<unsigned *>(buf + A.tables_lds_bytes - VISIT_LDS_BYTES);
	if (b >= A.n_buckets) {
		if (tid == 0) rec[29] = 0u;
		return;
	}
	const BucketDesc d = A.buckets[b];
	const unsigned long long bs = A.slot_off[b], be = A.slot_off[b + 1];
	const unsigned long long p1a = A.p1_off[b], p1b = A.p1_off[b + 1], p2a = A.p2_off[b], p2b = A.p2_off[b + 1];
	global_words src = (global_words)(A.images + d.img_off);
	uint4 *dst = reinterpret_cast<uint4 *>(buf);
	for (unsigned i = tid; i < d.img_bytes / 16; i += COUNT_BLOCK) { const u32x4 t = src[i]; dst[i] = make_uint4(t.x, t.y, t.z, t.w); }
	unsigned long long *h = reinterpret_cast<unsigned long long *>(buf + d.hist_off);
	for (unsigned i = tid; i < HIST_REPLICAS * (d.n_cls | 1u); i += COUNT_BLOCK) h[i] = 0;
	if (tid == 0) {
		rec[0] = d.img_off; rec[1] = d.img_bytes; rec[2] = d.n_events; rec[3] = d.n_bins; rec[4] = (unsigned)d.lo; rec[5] = d.shift;
		rec[6] = d.ev_off; rec[7] = d.seg_off; rec[8] = d.iso_off; rec[9] = d.hist_off; rec[10] = d.n_cls; rec[11] = d.cls_base;
		rec[12] = d.ev_base; rec[13] = (unsigned)d.chrom_id; rec[14] = d.kind; rec[15] = (unsigned)d.hi;
		unsigned long long *r64 = reinterpret_cast<unsigned long long *>(rec + 16);
		r64[0] = bs; r64[1] = be; r64[2] = p1a; r64[3] = p1b - p1a; r64[4] = p2a; r64[5] = p2b - p2a;
		rec[28] = b; rec[29] = 1u;
	}
}

__global__ void __launch_bounds__(COUNT_BLOCK, LSQ_FAST_WAVES) lsq_count_fast_kernel(CountArgs A) {
	// LDS: the bucket's tables (image, histograms, visit record), then the waves' rings
	extern __shared__ __align__(16) unsigned char lds[];
	const unsigned tid = threadIdx.x;
	if (blockIdx.x < A.n_workers) { pool_n_worker(A, A.n_pn, A.n_workers); return; }
	const unsigned wg = blockIdx.x - A.n_workers, n_wg = gridDim.x - A.n_workers;     // the streaming workgroups
	uint4 *wave_queue = reinterpret_cast<uint4 *>(lds + A.tables_lds_bytes) + (tid >> 6) * WAVE_QUEUE_WORDS;
	const unsigned long long s_begin = A.total_slots * wg / n_wg;
	const unsigned long long s_end = A.total_slots * (wg + 1ull) / n_wg;
	if (s_begin >= s_end) return;
	if (A.ablate & 4096u) return;       // developer switch: dispatch cost only
	{
		const unsigned b0 = find_bucket(A, A.wg_first[wg], s_begin, s_end);   // wg_first: lsq_wg_plan_kernel
		if (b0 >= A.n_buckets) return;
		stage_bucket(A, b0, lds);
	}
	__syncthreads();
	if (A.ablate & 8192u) return;       // developer switch: dispatch + first staging
	for (;;) {
		unsigned char *buf = lds;
		// the visit record, wave-uniform: every dword through readfirstlane so that it lives in scalar registers
		const unsigned *rec = reinterpret_cast<const unsigned *>(buf + A.tables_lds_bytes - VISIT_LDS_BYTES);
		auto r32 = [&](unsigned q) { return (unsigned)__builtin_amdgcn_readfirstlane((int)rec[q]); };
		auto r64 = [&](unsigned q) { return (unsigned long long)r32(q) | ((unsigned long long)r32(q + 1) << 32); };
		BucketVisit V;
		V.d.img_off = r32(0); V.d.img_bytes = r32(1); V.d.n_events = r32(2); V.d.n_bins = r32(3); V.d.lo = (int)r32(4); V.d.shift = r32(5);
		V.d.ev_off = r32(6); V.d.seg_off = r32(7); V.d.iso_off = r32(8); V.d.hist_off = r32(9); V.d.n_cls = r32(10); V.d.cls_base = r32(11);
		V.d.ev_base = r32(12); V.d.chrom_id = (int)r32(13); V.d.kind = r32(14); V.d.hi = (int)r32(15);
		V.bs = r64(16); V.be = r64(18); V.p1o = r64(20); V.p1n = r64(22); V.p2o = r64(24); V.p2n = r64(26);
		V.b = r32(28); V.valid = r32(29);
		const BucketDesc &d = V.d;
		const unsigned b = V.b;
		const uint4 *bins = reinterpret_cast<const uint4 *>(buf);
		const uint4 *cells = reinterpret_cast<const uint4 *>(buf + d.seg_off);
		const unsigned *cell_info = reinterpret_cast<const unsigned *>(buf + d.seg_off + 16u * d.iso_off);
		FastCtx C;
		C.bins = bins; C.lo = d.lo; C.shift = d.shift; C.n_bins = d.n_bins;
		C.recs = reinterpret_cast<const uint4 *>(buf + d.ev_off);
		C.hist = reinterpret_cast<unsigned long long *>(buf + d.hist_off) + (tid & (HIST_REPLICAS - 1u)) * (d.n_cls | 1u);   // this lane's copy
		C.n_events = d.n_events; C.bucket = b;
		C.slot0 = 0; C.pool = 0;
		C.exc = A.exc; C.exc_count = A.exc_count; C.exc_cap = A.exc_cap; C.ablate = A.ablate; C.dbg = A.dbg;
		const unsigned long long l0 = (s_begin > V.bs ? s_begin : V.bs) - V.bs;
		const unsigned long long l1 = (s_end < V.be ? s_end : V.be) - V.bs;
		const unsigned long long n1 = V.p1n, n2 = V.p2n;
		// ---- pool 1
		if (l0 < n1 && !(A.ablate & 1024u))
			stream_pool_fast<2>(C, bins, cells, cell_info, d.iso_off, d, A, wave_queue, reinterpret_cast<const uint4 *>(A.p1), V.p1o + l0, V.p1o + (l1 < n1 ? l1 : n1));
		// ---- pool 2
		if (l1 > n1 && l0 < n1 + n2 && !(A.ablate & 2048u))
			stream_pool_fast<1>(C, bins, cells, cell_info, d.iso_off, d, A, wave_queue, reinterpret_cast<const uint4 *>(A.p2), V.p2o + ((l0 > n1 ? l0 : n1) - n1),
			                    V.p2o + ((l1 < n1 + n2 ? l1 : n1 + n2) - n1));
		// (reads with three or more blocks are the workers')
		__syncthreads();
		// ---- flush
		for (unsigned i = tid; i < d.n_cls; i += COUNT_BLOCK) {
			const unsigned long long *h0 = reinterpret_cast<const unsigned long long *>(buf + d.hist_off) + i;
			unsigned long long v = 0;
#pragma unroll
			for (unsigned r = 0; r < HIST_REPLICAS; ++r) v += h0[r * (d.n_cls | 1u)];       // counts stay below 2^24, bases below 2^40
			if (v && !(A.ablate & 4u)) {
				atomicAdd(&A.cnt[d.cls_base + i], v >> 40);
				atomicAdd(&A.bases[d.cls_base + i], v & BASES_MASK);
			}
		}
		// the next bucket of the share, staged between two barriers (a second table buffer, filled while
		// this bucket streams, cost a resident workgroup per CU and measured slower)
		__syncthreads();
		stage_bucket(A, find_bucket(A, b + 1u, s_begin, s_end), lds);
		__syncthreads();
		if (!reinterpret_cast<const unsigned *>(lds + A.tables_lds_bytes - VISIT_LDS_BYTES)[29]) break;
	}
}

// =====================================================================================
// Cleanup kernel for FastRec buckets: everything the fast kernel does not settle -- the
// exception list and the reads with three or more blocks -- one lane per item, tables read
// from global memory (L2), stepwise walk, global atomics.  Rare work by construction; in
// `all_reads` mode it redoes pools 1 and 2 completely (used when the exception list overflowed).
// =====================================================================================
struct GlobalBucket {
	const BucketDesc *d;
	const unsigned *bins;          // packed buckets: 16-byte bin records, word 0 = first cell | first event << 16
	const uint4 *recs;
};

__device__ inline GlobalBucket global_bucket(const CountArgs &A, unsigned b) {
	GlobalBucket G;
	G.d = A.buckets + b;
	G.bins = reinterpret_cast<const unsigned *>(A.images + G.d->img_off);
	G.recs = reinterpret_cast<const uint4 *>(A.images + G.d->img_off + G.d->ev_off);
	return G;
}

// evaluates the read against event i (and, with scan, the following ones as the reference's
// index scan would); blocks at blk[0..nblk)
__device__ void eval_read_global(const CountArgs &A, const GlobalBucket &G, const int2 *blk, int nblk, unsigned i, bool scan,
                                 unsigned strand_id, unsigned line) {
	const BucketDesc &d = *G.d;
	const int p = blk[0].x, q = blk[nblk - 1].y;
	int total = 0;
	for (int k = 0; k < nblk; ++k) total += blk[k].y - blk[k].x;
	for (; i < d.n_events; ++i) {
		const uint4 w0 = G.recs[3u * i], w1 = G.recs[3u * i + 1], w2 = G.recs[3u * i + 2];
		const int gs = (int)w1.x, ge = (int)w0.x;
		if (gs > p) break;
		bool cand = p <= ge;
		if (cand && p == gs) {
			if (q < ge) cand = false;
			else if (q == ge && tie_orders_read_first(A, d.ev_base + i, strand_id, line)) cand = false;
		}
		if (cand) {
			int2 segs[4] = {make_int2((int)w1.x, (int)w1.y), make_int2((int)w1.z, (int)w1.w), make_int2((int)w2.x, (int)w2.y), make_int2((int)w2.z, (int)w2.w)};
			const int nseg = (int)((w0.y >> FAST_NSEG_SHIFT) & 7u);
			Walk w;
			for (int k = 0; k < nblk; ++k) { const int2 bk = blk[k]; if (!w.block(segs, nseg, bk.x, bk.y)) break; }
			const unsigned long long tbl = ((unsigned long long)w0.w << 32) | w0.z;
			const unsigned cls = w.mask < 16u ? (unsigned)(tbl >> (4u * w.mask)) & 0xFu : 0u;
			if (cls != 0 && 50ll * w.matched > 49ll * total) {
				const unsigned slot = d.cls_base + (w0.y & 0xFFFFu) + cls - 1;
				atomicAdd(&A.cnt[slot], 1ull);
				atomicAdd(&A.bases[slot], (unsigned long long)(unsigned)w.matched);
			}
		}
		if (!scan) break;
		if (p <= ge && !(w0.y & FAST_FLAG_OVERLAPS_NEXT)) break;
	}
}

__device__ inline unsigned first_event_for(const GlobalBucket &G, int p) {
	const BucketDesc &d = *G.d;
	const int rel = p - d.lo;
	unsigned bin = rel <= 0 ? 0u : ((unsigned)rel >> d.shift);
	bin = min(bin, d.n_bins - 1u);
	return G.bins[4u * bin] >> 16;
}

__global__ void __launch_bounds__(256) lsq_count_cleanup_kernel(CountArgs A, unsigned long long n_p1, unsigned long long n_p2,
                                                                unsigned long long n_pn, int all_reads) {
	const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	// ---- exception list
	if (!all_reads) {
		const unsigned n_raw = A.exc_count[0];
		if (n_raw > A.exc_cap && gtid == 0) A.exc_count[1] = 1u;       // overflow: the host redoes the pass in all_reads mode
		const unsigned n = min(n_raw, A.exc_cap);
		for (unsigned long long k = gtid; k < n; k += gsz) {
			const ExcEntry e = A.exc[k];
			const GlobalBucket G = global_bucket(A, e.bucket);
			const unsigned i = e.ev_pool_scan & 0x1FFFFFFFu, pool = (e.ev_pool_scan >> 29) & 3u;
			const bool scan = (e.ev_pool_scan >> 31) != 0;
			int2 blk[2];
			if (pool == 2) { const unsigned o0 = A.pn_blk_off[e.slot]; eval_read_global(A, G, A.pn_se + o0, (int)A.pn_nblk[e.slot], i, scan, A.pn_strand[e.slot], A.pn_line[e.slot]); }
			else if (pool == 0) { blk[0] = A.p1[e.slot]; eval_read_global(A, G, blk, 1, i, scan, A.p1_strand[e.slot], A.p1_line[e.slot]); }
			else { const int4 v = A.p2[e.slot]; blk[0] = make_int2(v.x, v.y); blk[1] = make_int2(v.z, v.w); eval_read_global(A, G, blk, 2, i, scan, A.p2_strand[e.slot], A.p2_line[e.slot]); }
		}
	}
	// ---- all_reads mode: the reads with three or more blocks, one lane each (otherwise the fast kernel's
	// pool-n workers have done them)
	for (unsigned long long g = gtid; all_reads && g < n_pn; g += gsz) {
		const unsigned b = A.pn_bucket[g];
		if (A.buckets[b].kind != 1) continue;
		const GlobalBucket G = global_bucket(A, b);
		const unsigned o0 = A.pn_blk_off[g], o1 = o0 + A.pn_nblk[g];
		eval_read_global(A, G, A.pn_se + o0, (int)(o1 - o0), first_event_for(G, A.pn_se[o0].x), true, A.pn_strand[g], A.pn_line[g]);
	}
	// ---- all_reads mode: every one- and two-block read as well, bucket by bucket, one wave at a time
	if (all_reads) {
		const unsigned lane = threadIdx.x & 63u;
		const unsigned wave_id = (unsigned)(gtid >> 6), n_waves = (unsigned)(gsz >> 6);
		for (unsigned b = wave_id; b < A.n_buckets; b += n_waves) {
			if (A.buckets[b].kind != 1) continue;
			const GlobalBucket G = global_bucket(A, b);
			for (unsigned long long g = A.p1_off[b] + lane; g < A.p1_off[b + 1]; g += 64u) {
				int2 blk[1] = {A.p1[g]};
				eval_read_global(A, G, blk, 1, first_event_for(G, blk[0].x), true, A.p1_strand[g], A.p1_line[g]);
			}
			for (unsigned long long g = A.p2_off[b] + lane; g < A.p2_off[b + 1]; g += 64u) {
				const int4 v = A.p2[g];
				int2 blk[2] = {make_int2(v.x, v.y), make_int2(v.z, v.w)};
				eval_read_global(A, G, blk, 2, first_event_for(G, v.x), true, A.p2_strand[g], A.p2_line[g]);
			}
		}
	}
}

// ---- EM: one event per lane (common/read.h:592-660 on compatibility classes) ----------------
struct EmArgs {
	unsigned n_events, n_methods, n_cls, n_iso;
	unsigned n_places;                 // entries of `order`
	const unsigned *order;             // device event per place of the EM grid
	const unsigned char *K;
	const unsigned *cls_base, *iso_base;
	const unsigned long long *cnt;     // [method][n_cls]
	const double *G;                   // [method][n_iso]
	double *theta, *logll;
	unsigned *iters;
	unsigned char *flags;
};

// Four lanes per event: lane `sub` takes the (method, class) pairs sub, sub+4, ...; the four
// partial sums meet by two xor-shuffles.  One pass per EM iteration gives, for the current theta,
// the class mixtures s, the log-likelihood and the numerators of the next theta.  All four
// lanes of an event hold the same theta and take the same decisions.
constexpr int EM_LANES = 4;

// sum over the four lanes of an event (an aligned quad): two DPP quad permutes per 32-bit half,
// plain VALU moves with no trip through the LDS crossbar
template <int CTRL>
__device__ inline double quad_perm_f64(double x) {
	int lo = __double2loint(x), hi = __double2hiint(x);
	lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
	hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}
__device__ inline double group_sum(double x) {
	x += quad_perm_f64<0xB1>(x);      // quad_perm:[1,0,3,2]
	x += quad_perm_f64<0x4E>(x);      // quad_perm:[2,3,0,1]
	return x;
}

__device__ inline void em_pass(const EmArgs &A, unsigned cb, unsigned ib, int K, int nc, unsigned sub, bool on,
                               const double *th, double &ll, double *z) {
	double l = 0;
	double zz[LSQ_MAX_ISOFORMS];
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) zz[j] = 0;
	if (on) {
		const int n_pairs = (int)A.n_methods * nc;
		for (int q = (int)sub; q < n_pairs; q += EM_LANES) {
			const int m = q / nc, c = q - m * nc + 1;
			const unsigned long long k = A.cnt[(size_t)m * A.n_cls + cb + (unsigned)(c - 1)];
			if (!k) continue;
			const double *g = A.G + (size_t)m * A.n_iso + ib;
			double s = 0;
#pragma unroll
			for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) if (j < K && (c >> j & 1)) s += th[j] * g[j];
			const double kd = (double)k;
			l += kd * log(s);
			if (s > 0) {
#pragma unroll
				for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) if (j < K && (c >> j & 1)) {
					const double local = th[j] * g[j];
					if (local > 0) zz[j] += kd * (local / s);
				}
			}
		}
	}
	ll = group_sum(l);
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) z[j] = group_sum(zz[j]);
}

// An event's (method, class) counts and G values in registers when there are at most two pairs
// per lane and three isoforms (every LESSeq local event with up to two read files): the passes
// then touch no memory, and the latency of one pass is what bounds the kernel (the slowest
// event of the batch runs ~160 dependent passes).
constexpr int EM_CACHED_PAIRS = 2, EM_CACHED_K = 3;
struct EmCache {
	double kd[EM_CACHED_PAIRS];
	double g[EM_CACHED_PAIRS][EM_CACHED_K];
	int cls[EM_CACHED_PAIRS];
};

// 1/s to ~1 ulp: hardware reciprocal estimate and two Newton steps -- about half the dependent
// chain of an IEEE division (the result stays far inside the 1e-6 tolerance of the path)
__device__ inline double fast_recip(double s) {
	double r = __builtin_amdgcn_rcp(s);
	r = fma(fma(-s, r, 1.0), r, r);
	r = fma(fma(-s, r, 1.0), r, r);
	return r;
}

// log(s) for normal positive s to about 1 ulp (everything else goes to the library): exponent and
// mantissa m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(z) with z = (m-1)/(m+1), |z| < 0.172, as an
// odd series in z evaluated by Estrin's scheme -- a dependent chain of about 20 operations, a
// third of the library routine's.  The stop rule compares log-likelihoods to 1e-6; events whose
// criterion comes within 1e-11 of it are flagged whatever the logarithm used.
__device__ inline double fast_log(double s) {
	const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
	const unsigned ex = (unsigned)(bits >> 52);
	if (ex - 1u >= 0x7FEu) return log(s);                       // zero, subnormal, negative, inf, nan
	double m = __longlong_as_double((long long)((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
	int e = (int)ex - 1023;
	if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
	const double f = m - 1.0;
	const double z = f * fast_recip(2.0 + f);
	const double w = z * z, w2 = w * w, w4 = w2 * w2, w8 = w4 * w4;
	// 1/3 + w/5 + w^2/7 + ... + w^9/21
	const double p01 = fma(w, 1.0 / 5.0, 1.0 / 3.0), p23 = fma(w, 1.0 / 9.0, 1.0 / 7.0), p45 = fma(w, 1.0 / 13.0, 1.0 / 11.0),
	             p67 = fma(w, 1.0 / 17.0, 1.0 / 15.0), p89 = fma(w, 1.0 / 21.0, 1.0 / 19.0);
	const double q0 = fma(w2, p23, p01), q1 = fma(w2, p67, p45);
	const double poly = fma(w8, p89, fma(w4, q1, q0));
	const double lm = fma(z * w, 2.0 * poly, 2.0 * z);
	const double ed = (double)e;
	return fma(ed, 0.69314718055994528623, fma(ed, 2.3190468138462995584e-17, lm));
}

__device__ inline void em_pass_cached(const EmCache &E, const double *th, bool on, double &ll, double *z) {
	double l = 0, zz[EM_CACHED_K] = {0, 0, 0};
#pragma unroll
	for (int t = 0; t < EM_CACHED_PAIRS; ++t) {
		const double kd = E.kd[t];
		if (on && kd != 0) {
			const int c = E.cls[t];
			double s = 0;
#pragma unroll
			for (int j = 0; j < EM_CACHED_K; ++j) if (c >> j & 1) s += th[j] * E.g[t][j];
			l += kd * fast_log(s);
			if (s > 0) {
				const double kr = kd * fast_recip(s);
#pragma unroll
				for (int j = 0; j < EM_CACHED_K; ++j) if (c >> j & 1) {
					const double local = th[j] * E.g[t][j];
					if (local > 0) zz[j] += local * kr;
				}
			}
		}
	}
	ll = group_sum(l);
#pragma unroll
	for (int j = 0; j < EM_CACHED_K; ++j) z[j] = group_sum(zz[j]);
}

// The register-cached pass with the class masks folded into G (an isoform outside the class has
// G = 0: its term adds an exact zero, so sums and their order are those of em_pass_cached).
// What a pass leaves behind per pair for the next one: the mixture s, its reciprocal and its
// logarithm.  Passes follow one another with small steps in s (that is what makes slow events
// slow), so log s(t+1) = log s(t) + log1p(d) with d = (s(t+1) - s(t)) / s(t), and for |d| < 2^-5 a
// twelve-term series gives log1p to 1e-19: a chain of six operations instead of the logarithm's twenty.
// A wave takes the series only when every live pair of every lane is inside that range.
template <int SLOTS>
struct EmPairState { double s[SLOTS], r[SLOTS], lg[SLOTS]; };

__device__ inline double log1p_small(double d) {
	// d (1 - d/2 + d^2/3 - ... - d^11/12), |d| < 2^-5: the first term left out is below 1e-19
	const double w = d * d, w2 = w * w, w4 = w2 * w2;
	const double a0 = fma(d, -1.0 / 2.0, 1.0), a1 = fma(d, -1.0 / 4.0, 1.0 / 3.0), a2 = fma(d, -1.0 / 6.0, 1.0 / 5.0),
	             a3 = fma(d, -1.0 / 8.0, 1.0 / 7.0), a4 = fma(d, -1.0 / 10.0, 1.0 / 9.0), a5 = fma(d, -1.0 / 12.0, 1.0 / 11.0);
	const double b0 = fma(w, a1, a0), b1 = fma(w, a3, a2), b2 = fma(w, a5, a4);
	return d * fma(w4, b2, fma(w2, b1, b0));
}

template <int SLOTS, int KK>
__device__ inline void em_pass_lean(const double (&kd)[SLOTS], const double (&gm)[SLOTS][KK], const double (&th)[KK],
                                    const bool on, EmPairState<SLOTS> &P, double &ll, double (&z)[KK]) {
	double l = 0, zz[KK];
#pragma unroll
	for (int j = 0; j < KK; ++j) zz[j] = 0;
	double local[SLOTS][KK], sm[SLOTS], safe[SLOTS], r[SLOTS], d[SLOTS];
	bool on_t[SLOTS], far = false;
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		sm[t] = 0;
#pragma unroll
		for (int j = 0; j < KK; ++j) { local[t][j] = th[j] * gm[t][j]; sm[t] += local[t][j]; }
		on_t[t] = on && kd[t] != 0;
		safe[t] = (on_t[t] && sm[t] > 0) ? sm[t] : 1.0;     // an empty pair slot must not send the wave down the library path
		r[t] = fast_recip(safe[t]);
		d[t] = (safe[t] - P.s[t]) * P.r[t];
		far = far || (on_t[t] && !(fabs(d[t]) < 0.03125));
		if (on_t[t] && !(sm[t] > 0)) far = true;             // log of zero: the full routine gives the reference's -inf
	}
	// the numerators first: the next pass waits for them, and nothing in them waits for the logarithm
	// or for the wave-wide vote below (an in-order wave stalls at that branch until the vote is in)
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		const double kr = (on_t[t] && sm[t] > 0) ? kd[t] * r[t] : 0.0;
#pragma unroll
		for (int j = 0; j < KK; ++j) zz[j] += local[t][j] * kr;
	}
#pragma unroll
	for (int j = 0; j < KK; ++j) z[j] = group_sum(zz[j]);
	const bool full = __any(far);
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		double lg = P.lg[t] + log1p_small(d[t]);
		if (full) {                                  // wave-uniform, taken a handful of times per event
			asm volatile("" ::: "memory");           // keeps the compiler from flattening the branch into both computations
			lg = fast_log(on_t[t] ? sm[t] : 1.0);
		}
		P.s[t] = safe[t]; P.r[t] = r[t]; P.lg[t] = lg;
		const double term = kd[t] * lg;
		l += on_t[t] ? term : 0.0;
	}
	ll = group_sum(l);
}

// The whole EM of a wave whose events all fit SLOTS (method, class) pairs per lane and KK isoforms,
// in registers.  The pass for theta(t+2) starts from z(t+1) as soon as that exists, without waiting
// for the stop test on ll(t+1): the test (a reciprocal, a compare, a ballot) runs beside the next
// pass instead of between two passes.  One pass per event is thrown away.
template <int SLOTS, int KK>
__device__ inline void em_lean(const EmArgs &A, const EmCache &C, const unsigned e, const unsigned sub, const bool ev_ok, const int K, const unsigned ib,
                               const double inv_n, const bool any_reads, bool run) {
	double kd[SLOTS], gm[SLOTS][KK], t3[KK], z3[KK];
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		kd[t] = C.kd[t];
#pragma unroll
		for (int j = 0; j < KK; ++j) gm[t][j] = (C.cls[t] >> j & 1) ? C.g[t][j] : 0.0;
	}
#pragma unroll
	for (int j = 0; j < KK; ++j) t3[j] = (K == 1) ? 1.0 : 1.0 / (double)K;   // solve/solve.cpp:798-802, read.h:642
	EmPairState<SLOTS> P;
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) { P.s[t] = 1.0; P.r[t] = 1.0; P.lg[t] = 0.0; }
	unsigned iters = 0;
	unsigned char flag = 0;
	double ll = 0;
	em_pass_lean<SLOTS, KK>(kd, gm, t3, any_reads, P, ll, z3);
	double c3[KK], cll, cz3[KK];           // candidate: theta(t+1), its log-likelihood and numerators
#pragma unroll
	for (int j = 0; j < KK; ++j) c3[j] = z3[j] * inv_n;
	em_pass_lean<SLOTS, KK>(kd, gm, c3, run, P, cll, cz3);
	while (__any(run)) {
		double n3[KK], nll, nz3[KK];
#pragma unroll
		for (int j = 0; j < KK; ++j) n3[j] = cz3[j] * inv_n;
		em_pass_lean<SLOTS, KK>(kd, gm, n3, run, P, nll, nz3);         // speculative: theta(t+2)
		const unsigned cll_ex = (unsigned)((unsigned long long)__double_as_longlong(cll) >> 52) & 0x7FFu;
		// read.h:659, floating abs; -inf, nan, zero keep the division's own answers
		const double crit = (cll_ex - 1u < 0x7FEu) ? fabs(1.0 - ll * fast_recip(cll)) : fabs(1.0 - ll / cll);
		const bool go = run;
#pragma unroll
		for (int j = 0; j < KK; ++j) { t3[j] = go ? c3[j] : t3[j]; z3[j] = go ? cz3[j] : z3[j]; }
		ll = go ? cll : ll;
		iters += go ? 1u : 0u;
		if (go && fabs(crit - 1E-6) < 1E-11) flag |= 1;
		if (go && !(crit > 1E-6)) run = false;
		else if (go && iters >= 1000000u) { flag |= 2; run = false; }
#pragma unroll
		for (int j = 0; j < KK; ++j) { c3[j] = n3[j]; cz3[j] = nz3[j]; }
		cll = nll;
	}
	if (ev_ok && sub == 0) {
#pragma unroll
		for (int j = 0; j < KK; ++j) if (j < K) A.theta[ib + j] = t3[j];
		A.logll[e] = ll;
		A.iters[e] = iters;
		A.flags[e] = flag;
	}
}

__global__ void __launch_bounds__(256) lsq_em_kernel(EmArgs A) {
	const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned place = gid / EM_LANES, sub = gid % EM_LANES;
	// events in the order of A.order: the small ones (two isoforms, one pair per lane) first, then the
	// rest, each group filling whole waves (0xFFFFFFFF = empty place)
	const unsigned e = place < A.n_places ? A.order[place] : 0xFFFFFFFFu;
	const bool ev_ok = e != 0xFFFFFFFFu;
	const int K = ev_ok ? A.K[e] : 1;
	const unsigned cb = ev_ok ? A.cls_base[e] : 0, ib = ev_ok ? A.iso_base[e] : 0;
	const int nc = (1 << K) - 1;
	const int n_pairs = (int)A.n_methods * nc;
	const bool cached = K <= EM_CACHED_K && n_pairs <= EM_LANES * EM_CACHED_PAIRS;
	double th[LSQ_MAX_ISOFORMS], z[LSQ_MAX_ISOFORMS];
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) z[j] = 0;
	EmCache C;
	double tot = 0;
#pragma unroll
	for (int t = 0; t < EM_CACHED_PAIRS; ++t) {
		C.kd[t] = 0; C.cls[t] = 0;
#pragma unroll
		for (int j = 0; j < EM_CACHED_K; ++j) C.g[t][j] = 0;
	}
	if (ev_ok) {
		for (int q = (int)sub; q < n_pairs; q += EM_LANES) {
			const int m = q / nc, c = q - m * nc;
			const double kd = (double)A.cnt[(size_t)m * A.n_cls + cb + (unsigned)c];      // exact: counts are far below 2^53
			tot += kd;
			const int t = (q - (int)sub) / EM_LANES;
			if (cached && t < EM_CACHED_PAIRS) {
#pragma unroll
				for (int tt = 0; tt < EM_CACHED_PAIRS; ++tt) if (tt == t) {
					C.kd[tt] = kd; C.cls[tt] = c + 1;
#pragma unroll
					for (int j = 0; j < EM_CACHED_K; ++j) C.g[tt][j] = j < K ? A.G[(size_t)m * A.n_iso + ib + j] : 0.0;
				}
			}
		}
	}
	const double n_total = group_sum(tot);
	const double inv_n = 1.0 / n_total;
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) th[j] = (K == 1) ? 1.0 : 1.0 / (double)K;   // solve/solve.cpp:798-802, read.h:642
	unsigned iters = 0;
	unsigned char flag = 0;
	double ll = 0;
	// no reads: theta stays 1/K, log-likelihood 0; one isoform: theta = 1 (solve/solve.cpp:798-802)
	bool run = ev_ok && n_total > 0 && K > 1;
	const bool any_reads = ev_ok && n_total > 0;
	// every event of the wave fits the registers: a loop with nothing but the lean pass in it
	if (__all(!ev_ok || (cached && K <= 2 && n_pairs <= EM_LANES))) { em_lean<1, 2>(A, C, e, sub, ev_ok, K, ib, inv_n, any_reads, run); return; }
	if (__all(!ev_ok || cached)) { em_lean<EM_CACHED_PAIRS, EM_CACHED_K>(A, C, e, sub, ev_ok, K, ib, inv_n, any_reads, run); return; }
	if (cached) em_pass_cached(C, th, any_reads, ll, z);
	else em_pass(A, cb, ib, K, nc, sub, any_reads, th, ll, z);
	while (__any(run)) {
		// theta' = z(theta) / n; then one pass at theta' gives ll(theta') and z(theta')
		double nth[LSQ_MAX_ISOFORMS], nll, nz[LSQ_MAX_ISOFORMS];
#pragma unroll
		for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) { nth[j] = cached ? z[j] * inv_n : z[j] / n_total; nz[j] = 0; }
		if (cached) em_pass_cached(C, nth, run, nll, nz);
		else em_pass(A, cb, ib, K, nc, sub, run, nth, nll, nz);
		if (run) {
			// read.h:659, floating abs; the quotient through the reciprocal when the passes are the
			// register-cached ones (1 ulp, against a guard band of 1e-11 around the threshold)
			const unsigned nll_ex = (unsigned)((unsigned long long)__double_as_longlong(nll) >> 52) & 0x7FFu;
			const bool nll_normal = nll_ex - 1u < 0x7FEu;     // -inf, nan, zero keep the division's own answers
			const double crit = (cached && nll_normal) ? fabs(1.0 - ll * fast_recip(nll)) : fabs(1.0 - ll / nll);
#pragma unroll
			for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) { th[j] = nth[j]; z[j] = nz[j]; }
			ll = nll;
			++iters;
			if (fabs(crit - 1E-6) < 1E-11) flag |= 1;
			if (!(crit > 1E-6)) run = false;
			else if (iters >= 1000000u) { flag |= 2; run = false; }
		}
	}
	if (ev_ok && sub == 0) {
		for (int j = 0; j < K; ++j) A.theta[ib + j] = th[j];
		A.logll[e] = ll;
		A.iters[e] = iters;
		A.flags[e] = flag;
	}
}

// =====================================================================================
// Ingest on the device: from parsed blocks in file order to the bucketed, pooled arrays.
//   classify: per read, the per-block containment filter against the covered regions of the
//             block's own chromosome (count/count.cpp:319, interval_list.hpp:396-422), the
//             interval_list merge of the kept blocks (:323, interval_list.hpp:462-503),
//             chromosome/strand of the last kept block (:321-322), the bucket of the first
//             merged base and the pool (1, 2, 3+ blocks); per (bucket, pool) counts
//   scan    : exclusive prefix sums -> offsets per (bucket, bin) for the one- and two-block pools
//             (bin = the bucket's coordinate bin of the read's first base, the one the count kernel
//             looks up), per bucket for the n-block pool
//   scatter : every retained read to its place: a counting sort, so the reads of a bin -- which
//             mostly share a cell -- sit together and a wave of the count kernel sees one or two
//             cells at a time (order inside a bin is whatever the atomics give; the count kernels
//             only add integers, so results do not depend on it)
// This replaces the reference's load-time filter and its read index (count/count.cpp:348-364).
// =====================================================================================
constexpr int INGEST_MAX_BLOCKS = 16;                  // merged blocks per read the device ingest handles
constexpr unsigned INGEST_NO_KEY = 0xFFFFFFFFu;

struct IngestTables {
	const unsigned *cov_off;       // per chromosome id: range of its covered intervals
	const int *cov_s, *cov_e;
	const unsigned *cut_off;       // per chromosome id: range of its bucket cuts
	const int *cut_lo;
	const int *chrom_first_bucket;
	const BucketDesc *buckets;
	const unsigned *bin_base;      // per bucket: first of its bins in the fine counters (n_buckets + 1)
	unsigned n_chrom;
};

struct IngestRaw {
	unsigned long long n_reads;
	const unsigned long long *blk_off;
	const unsigned *line_no;
	const int *blk_start, *blk_end;
	const unsigned short *blk_chrom;
	const unsigned char *blk_strand;
};

struct IngestWork {
	unsigned *key;                 // per read: bucket * 4 + pool, or INGEST_NO_KEY
	unsigned *fine;                // per read: bin_base[bucket] + bin of the first base
	unsigned char *nb;             // per read: merged blocks
	unsigned char *strand;         // per read: strand id of the last kept block
	int *ms, *me;                  // merged blocks, at the read's original block offset
	unsigned *cnt1, *cnt2;         // [n_fine]: one- / two-block reads per (bucket, bin)
	unsigned *cntn, *cntnb;        // [n_buckets]: n-block reads, and their blocks
	unsigned *cur1, *cur2, *curn, *curnb;   // scatter cursors, same shapes
	unsigned long long *totals;    // [0] retained reads, [1] retained blocks, [2] error flag
};

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

__device__ inline bool covered_contains(const IngestTables &T, unsigned chrom, int start, int end) {
	if (!(start < end)) return true;
	const unsigned lo0 = T.cov_off[chrom], hi0 = T.cov_off[chrom + 1];
	unsigned lo = lo0, hi = hi0;                      // lower_bound(starts, start)
	while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (T.cov_s[mid] < start) lo = mid + 1; else hi = mid; }
	if (lo < hi0 && T.cov_s[lo] <= start && end <= T.cov_e[lo]) return true;
	if (lo > lo0 && T.cov_s[lo - 1] <= start && end <= T.cov_e[lo - 1]) return true;
	return false;
}

__global__ void __launch_bounds__(256) lsq_ingest_classify_kernel(IngestTables T, IngestRaw R, IngestWork W) {
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	unsigned long long kept_reads = 0, kept_blocks = 0;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < R.n_reads; i += gsz) {
		const unsigned long long b0 = R.blk_off[i], b1 = R.blk_off[i + 1];
		int s[INGEST_MAX_BLOCKS], e[INGEST_MAX_BLOCKS];
		int n = 0, chrom = -1;
		unsigned strand = 0;
		bool any = false, ok = true;
		for (unsigned long long j = b0; j < b1; ++j) {
			const unsigned c = R.blk_chrom[j];
			if (c >= T.n_chrom) continue;
			const int bs = R.blk_start[j], be = R.blk_end[j];
			if (!covered_contains(T, c, bs, be)) continue;
			any = true; chrom = (int)c; strand = R.blk_strand[j];
			ok = small_add_interval(s, e, n, bs, be) && ok;
		}
		unsigned key = INGEST_NO_KEY;
		if (any && n > 0) {
			++kept_reads; kept_blocks += (unsigned)n;
			int tot = 0;
			for (int q = 0; q < n; ++q) tot += e[q] - s[q];
			if (!ok || tot >= (1 << 18)) atomicMax(&W.totals[2], 1ull);
			// bucket of the first merged base
			const int first = T.chrom_first_bucket[chrom];
			if (first >= 0) {
				const unsigned c0 = T.cut_off[chrom], c1 = T.cut_off[chrom + 1];
				unsigned lo = c0, hi = c1;                  // upper_bound(cuts, p)
				while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (T.cut_lo[mid] <= s[0]) lo = mid + 1; else hi = mid; }
				if (lo > c0) {
					const unsigned b = (unsigned)first + (lo - c0 - 1);
					if (s[0] <= T.buckets[b].hi) {
						const unsigned pool = n == 1 ? 0u : (n == 2 ? 1u : 2u);
						key = b * 4u + pool;
						const BucketDesc &d = T.buckets[b];
						const int rel = s[0] - d.lo;
						const unsigned bin = rel <= 0 ? 0u : min((unsigned)rel >> d.shift, d.n_bins - 1u);
						const unsigned fine = T.bin_base[b] + bin;
						W.fine[i] = fine;
						if (pool == 0) atomicAdd(&W.cnt1[fine], 1u);
						else if (pool == 1) atomicAdd(&W.cnt2[fine], 1u);
						else { atomicAdd(&W.cntn[b], 1u); atomicAdd(&W.cntnb[b], (unsigned)n); }
					}
				}
			}
			for (int q = 0; q < n; ++q) { W.ms[b0 + q] = s[q]; W.me[b0 + q] = e[q]; }
		}
		W.key[i] = key;
		W.nb[i] = (unsigned char)n;
		W.strand[i] = (unsigned char)strand;
	}
	if (kept_reads) { atomicAdd(&W.totals[0], kept_reads); atomicAdd(&W.totals[1], kept_blocks); }
}

// one workgroup: out[i] = sum of in[0..i), out[n] = total
__global__ void __launch_bounds__(1024) lsq_scan_u32_kernel(const unsigned *in, unsigned long long n, unsigned long long *out) {
	__shared__ unsigned long long part[1024];
	const unsigned tid = threadIdx.x;
	const unsigned long long per = (n + 1023ull) / 1024ull;
	const unsigned long long b0 = min(tid * per, n), b1 = min(b0 + per, n);
	unsigned long long acc = 0;
	for (unsigned long long b = b0; b < b1; ++b) acc += in[b];
	part[tid] = acc;
	__syncthreads();
	if (tid == 0) { unsigned long long run = 0; for (unsigned t = 0; t < 1024; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; } }
	__syncthreads();
	unsigned long long run = part[tid];
	for (unsigned long long b = b0; b < b1; ++b) { out[b] = run; run += in[b]; }
	if (tid == 1023) out[n] = run;
}

// per-bucket pool offsets out of the per-bin ones
__global__ void __launch_bounds__(256) lsq_ingest_offsets_kernel(const unsigned *bin_base, unsigned n_buckets, const unsigned long long *off1,
                                                                 const unsigned long long *off2, const unsigned long long *pn_off,
                                                                 unsigned long long *p1_off, unsigned long long *p2_off, unsigned long long *slot_off) {
	const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b > n_buckets) return;
	const unsigned long long a1 = off1[bin_base[b]], a2 = off2[bin_base[b]];
	p1_off[b] = a1; p2_off[b] = a2;
	slot_off[b] = a1 + a2 + pn_off[b];
}

struct IngestOut {
	int2 *p1; unsigned char *p1_strand; unsigned *p1_line;
	int4 *p2; unsigned char *p2_strand; unsigned *p2_line;
	unsigned *pn_blk_off, *pn_nblk, *pn_line, *pn_bucket; unsigned char *pn_strand; int2 *pn_se;
	const unsigned long long *off1, *off2, *pn_off, *pnb_off;       // per (bucket, bin) / per bucket
};

__global__ void __launch_bounds__(256) lsq_ingest_scatter_kernel(IngestRaw R, IngestWork W, IngestOut O) {
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < R.n_reads; i += gsz) {
		const unsigned key = W.key[i];
		if (key == INGEST_NO_KEY) continue;
		const unsigned b = key >> 2, pool = key & 3u;
		const unsigned long long b0 = R.blk_off[i];
		if (pool == 0) {
			const unsigned fine = W.fine[i];
			const unsigned long long w = O.off1[fine] + atomicAdd(&W.cur1[fine], 1u);
			O.p1[w] = make_int2(W.ms[b0], W.me[b0]);
			O.p1_strand[w] = W.strand[i]; O.p1_line[w] = R.line_no[i];
		} else if (pool == 1) {
			const unsigned fine = W.fine[i];
			const unsigned long long w = O.off2[fine] + atomicAdd(&W.cur2[fine], 1u);
			O.p2[w] = make_int4(W.ms[b0], W.me[b0], W.ms[b0 + 1], W.me[b0 + 1]);
			O.p2_strand[w] = W.strand[i]; O.p2_line[w] = R.line_no[i];
		} else {
			const unsigned n = W.nb[i];
			const unsigned long long w = O.pn_off[b] + atomicAdd(&W.curn[b], 1u);
			const unsigned long long bo = O.pnb_off[b] + atomicAdd(&W.curnb[b], n);
			O.pn_blk_off[w] = (unsigned)bo; O.pn_nblk[w] = n; O.pn_bucket[w] = b;
			O.pn_strand[w] = W.strand[i]; O.pn_line[w] = R.line_no[i];
			for (unsigned q = 0; q < n; ++q) O.pn_se[bo + q] = make_int2(W.ms[b0 + q], W.me[b0 + q]);
		}
	}
}

// Orders the reads of every (bucket, bin) by their first base: a counting sort in LDS over the
// bin's coordinates, one wave per bin (the scatter above left the bin's reads together, in the
// order its atomics gave).  This is the device form of the reference's read index, a std::set
// ordered by start (count/count.cpp:348-364): a wave of the count kernel then sees the reads of
// one cell, then those of the next.  Bins wider than BINSORT_MAX_W coordinates are copied as they
// are -- the order only matters for speed.
constexpr unsigned BINSORT_MAX_W = 2048;
template <class ReadT>
__global__ void __launch_bounds__(256) lsq_ingest_binsort_kernel(const BucketDesc *buckets, const unsigned *bin_base, unsigned n_buckets, unsigned n_fine,
                                                                 const unsigned long long *off, const ReadT *in, const unsigned char *in_strand,
                                                                 const unsigned *in_line, ReadT *out, unsigned char *out_strand, unsigned *out_line) {
	__shared__ unsigned cnt_all[4][BINSORT_MAX_W];
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned *cnt = cnt_all[wave];
	for (unsigned fine = blockIdx.x * 4u + wave; fine < n_fine; fine += gridDim.x * 4u) {
		const unsigned long long o0 = off[fine], o1 = off[fine + 1];
		if (o0 == o1) continue;
		const unsigned n = (unsigned)(o1 - o0);
		unsigned lo_b = 0, hi_b = n_buckets;                 // bucket of the bin: last b with bin_base[b] <= fine
		while (hi_b - lo_b > 1) { const unsigned mid = (lo_b + hi_b) >> 1; if (bin_base[mid] <= fine) lo_b = mid; else hi_b = mid; }
		const BucketDesc &d = buckets[lo_b];
		const unsigned W = d.shift < 31u ? (1u << d.shift) : 0x80000000u;
		if (W > BINSORT_MAX_W || n < 3) {
			for (unsigned i = lane; i < n; i += 64u) { out[o0 + i] = in[o0 + i]; out_strand[o0 + i] = in_strand[o0 + i]; out_line[o0 + i] = in_line[o0 + i]; }
			continue;
		}
		const int bin_lo = d.lo + (int)((fine - bin_base[lo_b]) << d.shift);
		for (unsigned k = lane; k < W; k += 64u) cnt[k] = 0;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
		for (unsigned i = lane; i < n; i += 64u) {
			const int rel = in[o0 + i].x - bin_lo;       // the first and last bins of a bucket also hold what lies beyond them
			atomicAdd(&cnt[(unsigned)max(0, min(rel, (int)W - 1))], 1u);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
		// exclusive prefix over the W counters: W/64 consecutive ones per lane
		const unsigned per = (W + 63u) / 64u, k0 = min(lane * per, W), k1 = min(k0 + per, W);
		unsigned acc = 0;
		for (unsigned k = k0; k < k1; ++k) acc += cnt[k];
		unsigned inc = acc;
		for (unsigned dd = 1; dd < 64; dd <<= 1) { const unsigned t = __shfl_up(inc, dd); if (lane >= dd) inc += t; }
		unsigned run = inc - acc;
		for (unsigned k = k0; k < k1; ++k) { const unsigned v = cnt[k]; cnt[k] = run; run += v; }
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
		for (unsigned i = lane; i < n; i += 64u) {
			const ReadT r = in[o0 + i];
			const int rel = r.x - bin_lo;
			const unsigned pos = atomicAdd(&cnt[(unsigned)max(0, min(rel, (int)W - 1))], 1u);
			out[o0 + pos] = r; out_strand[o0 + pos] = in_strand[o0 + i]; out_line[o0 + pos] = in_line[o0 + i];
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	}
}

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

struct MethodReads {
	bool present = false;
	uint64_t n_retained = 0, n_retained_blocks = 0, total_slots = 0;
	DevBuf<int32_t> p1, p2, pn_se;
	DevBuf<uint8_t> p1_strand, p2_strand, pn_strand;
	DevBuf<uint32_t> p1_line, p2_line, pn_line, pn_blk_off, pn_nblk, pn_bucket;
	DevBuf<unsigned long long> p1_off, p2_off, pn_off, pnb_off, slot_off;
	double skew = 1.0;                      // reads of the fullest bucket / mean reads per bucket
	bool named = false;                     // the reads carry their own names (the *_line arrays index name_off)
	DevBuf<char> names;
	DevBuf<unsigned long long> name_off;
	DevBuf<unsigned> wg_first;             // lsq_wg_plan_kernel's table for `wg_grid` workgroups
	unsigned long long wg_grid = 0;
};

} // namespace

struct lsq_ctx {
	int device = 0;
	int n_cu = 256;
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
	hipEvent_t evf0[LSQ_MAX_METHODS] = {}, evf1[LSQ_MAX_METHODS] = {};   // around each method's lsq_count_fast_kernel launch
	int fast_launched = 0;
	lsq_events *E = nullptr;                // must outlive the uploads made from it (its strand dictionary grows with the reads)
	DevBuf<BucketDesc> buckets;
	DevBuf<uint8_t> images, strand_rank, dK;
	DevBuf<TieRec> ties;
	DevBuf<uint32_t> cls_base, iso_base, iters, em_order, gene_name_off;
	DevBuf<char> gene_names;                // device event order
	unsigned em_places = 0;
	DevBuf<double> G, theta, logll;
	DevBuf<uint8_t> flags;
	DevBuf<unsigned long long> counters;   // cnt | bases | exc_count | dbg in one allocation: one memset per count
	DevView<unsigned long long> cnt, bases, dbg;
	DevBuf<ExcEntry> exc;                  // shared by the methods (launches are serialised on the stream)
	DevView<unsigned> exc_count;           // per method: [2m] appended, [2m+1] overflow flag
	DevBuf<unsigned> cov_off, cut_off;     // ingest tables: covered regions and bucket cuts per chromosome id
	DevBuf<int> cov_s, cov_e, cut_lo, chrom_first_bucket;
	DevBuf<unsigned> bin_base;             // per bucket: first of its bins among all bins (n_buckets + 1)
	size_t n_fine = 0;
	unsigned n_chrom_tables = 0;
	bool redo_checked = true;
	MethodReads reads[LSQ_MAX_METHODS];
	bool counted = false, solved = false;
	bool has_fast = false, has_generic = false;
	float count_ms = 0, solve_ms = 0;
	float mrf_h2d_ms = 0, mrf_parse_ms = 0;
};

static int upload_strand_ranks(lsq_ctx *c) {
	const auto &names = c->E->strands.names;
	if (names.size() > 256) return fail(LSQ_E_RANGE, "more than 256 distinct strand strings");
	std::vector<int> order(names.size());
	std::iota(order.begin(), order.end(), 0);
	std::sort(order.begin(), order.end(), [&](int a, int b) { return names[a] < names[b]; });
	std::vector<uint8_t> rank(256, 0);
	for (size_t r = 0; r < order.size(); ++r) rank[order[r]] = (uint8_t)r;
	int rc = c->strand_rank.upload(rank.data(), 256, c->stream);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return LSQ_OK;
}

#include "lsq_mrf_device.hpp"

extern "C" {

int lsq_ctx_create(int device_id, lsq_ctx **out) {
	if (!out) return fail(LSQ_E_ARG, "null argument");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) return fail(LSQ_E_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	if (device_id < 0 || device_id >= n) return fail(LSQ_E_ARG, "device %d out of range (%d devices)", device_id, n);
	HIP_TRY(hipSetDevice(device_id));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device_id));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(LSQ_E_DEVICE, "device %d is %s; this library carries gfx950 code only", device_id, prop.gcnArchName);
	std::unique_ptr<lsq_ctx> c(new lsq_ctx);
	c->device = device_id;
	c->n_cu = prop.multiProcessorCount;
	HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	HIP_TRY(hipEventCreate(&c->ev0)); HIP_TRY(hipEventCreate(&c->ev1));
	HIP_TRY(hipEventCreate(&c->ev2)); HIP_TRY(hipEventCreate(&c->ev3));
	for (int m = 0; m < LSQ_MAX_METHODS; ++m) { HIP_TRY(hipEventCreate(&c->evf0[m])); HIP_TRY(hipEventCreate(&c->evf1[m])); }
	*out = c.release();
	return LSQ_OK;
}

void lsq_ctx_destroy(lsq_ctx *c) {
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->ev2) (void)hipEventDestroy(c->ev2);
	if (c->ev3) (void)hipEventDestroy(c->ev3);
	for (int m = 0; m < LSQ_MAX_METHODS; ++m) { if (c->evf0[m]) (void)hipEventDestroy(c->evf0[m]); if (c->evf1[m]) (void)hipEventDestroy(c->evf1[m]); }
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

void *lsq_ctx_stream(lsq_ctx *c) { return c ? (void *)c->stream : nullptr; }
int lsq_ctx_synchronize(lsq_ctx *c) {
	if (!c) return fail(LSQ_E_ARG, "null context");
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return LSQ_OK;
}

int lsq_events_upload(lsq_ctx *c, lsq_events *E) {
	if (!c || !E) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	if (E->max_lds_bytes > 160 * 1024) return fail(LSQ_E_UNSUPPORTED, "bucket tables exceed the CU's LDS");
	c->E = E;
	c->counted = c->solved = false;
	c->has_fast = c->has_generic = false;
	for (const BucketDesc &bd : E->buckets) { if (bd.kind == 1) c->has_fast = true; else c->has_generic = true; }
	for (auto &r : c->reads) r.present = false;
	int rc;
	if ((rc = c->buckets.upload(E->buckets.data(), E->buckets.size(), c->stream))) return rc;
	if ((rc = c->images.upload(E->images.data(), E->images.size(), c->stream))) return rc;
	if ((rc = c->ties.upload(E->ties.data(), E->ties.size(), c->stream))) return rc;
	if ((rc = c->dK.upload(E->dev_K.data(), E->dev_K.size(), c->stream))) return rc;
	if ((rc = c->cls_base.upload(E->dev_cls_base.data(), E->dev_cls_base.size(), c->stream))) return rc;
	if ((rc = c->iso_base.upload(E->dev_iso_base.data(), E->dev_iso_base.size(), c->stream))) return rc;
	{
		// EM places: events with at most two isoforms and one (method, class) pair per lane first, then
		// the others; each group padded to whole waves (16 events) so that a wave runs one loop shape
		const size_t n_dev = E->dev2out.size();
		std::vector<uint32_t> order;
		order.reserve(n_dev + 32);
		for (int pass = 0; pass < 2; ++pass) {
			for (size_t d = 0; d < n_dev; ++d) {
				const int K = E->dev_K[d];
				const bool small = K <= 2 && E->n_methods * ((1 << K) - 1) <= EM_LANES;
				if (small == (pass == 0)) order.push_back((uint32_t)d);
			}
			while (order.size() % (64 / EM_LANES)) order.push_back(0xFFFFFFFFu);
		}
		c->em_places = (unsigned)order.size();
		if ((rc = c->em_order.upload(order.data(), order.size(), c->stream))) return rc;
		// gene names for span-start ties against named reads
		std::string blob;
		std::vector<uint32_t> goff(n_dev + 1, 0);
		for (size_t d = 0; d < n_dev; ++d) { blob += E->ev[E->dev2out[d]].gname; goff[d + 1] = (uint32_t)blob.size(); }
		if ((rc = c->gene_names.upload(blob.data(), blob.size(), c->stream))) return rc;
		if ((rc = c->gene_name_off.upload(goff.data(), goff.size(), c->stream))) return rc;
	}
	// G = 1/ARS (common/read.h:331-340), device isoform order, per method
	const size_t n_iso = E->n_iso_total, M = (size_t)E->n_methods;
	std::vector<double> G(std::max<size_t>(M * n_iso, 1), 0.0);
	for (size_t d = 0; d < E->dev2out.size(); ++d) {
		const Event &ev = E->ev[E->dev2out[d]];
		for (size_t m = 0; m < M; ++m)
			for (int j = 0; j < ev.K; ++j) {
				double nd = (double)ev.ars[m][j];
				G[m * n_iso + E->dev_iso_base[d] + j] = nd <= 0 ? 0.0 : (double)1.0 / nd;
			}
	}
	if ((rc = c->G.upload(G.data(), M * n_iso, c->stream))) return rc;
	const size_t n_cls = E->n_cls_total, n_ev = E->dev2out.size();
	{
		const size_t per = std::max<size_t>(M, 1) * n_cls;
		if ((rc = c->counters.alloc(2 * per + LSQ_MAX_METHODS + 8))) return rc;
		c->cnt.p = c->counters.p; c->cnt.n = per;
		c->bases.p = c->counters.p + per; c->bases.n = per;
		c->exc_count.p = reinterpret_cast<unsigned *>(c->counters.p + 2 * per); c->exc_count.n = 2 * LSQ_MAX_METHODS;
		c->dbg.p = c->counters.p + 2 * per + LSQ_MAX_METHODS; c->dbg.n = 8;
	}
	if ((rc = c->theta.alloc(n_iso))) return rc;
	if ((rc = c->logll.alloc(n_ev))) return rc;
	if ((rc = c->iters.alloc(n_ev))) return rc;
	if ((rc = c->flags.alloc(n_ev))) return rc;
	{
		// ingest tables: covered regions (by chromosome id) and the bucket cuts
		const size_t nc = E->covered.size();
		std::vector<unsigned> cov_off(nc + 1, 0), cut_off(nc + 1, 0);
		std::vector<int> cs, ce, cl, cfb(std::max<size_t>(nc, 1), -1);
		for (size_t ch = 0; ch < nc; ++ch) {
			for (size_t q = 0; q < E->covered[ch].s.size(); ++q) { cs.push_back((int)E->covered[ch].s[q]); ce.push_back((int)E->covered[ch].e[q]); }
			cov_off[ch + 1] = (unsigned)cs.size();
			if (ch < E->cut_lo.size()) for (int32_t v : E->cut_lo[ch]) cl.push_back(v);
			cut_off[ch + 1] = (unsigned)cl.size();
			if (ch < E->chrom_first_bucket.size()) cfb[ch] = E->chrom_first_bucket[ch];
		}
		c->n_chrom_tables = (unsigned)nc;
		if ((rc = c->cov_off.upload(cov_off.data(), cov_off.size(), c->stream))) return rc;
		if ((rc = c->cut_off.upload(cut_off.data(), cut_off.size(), c->stream))) return rc;
		if ((rc = c->cov_s.upload(cs.data(), cs.size(), c->stream))) return rc;
		if ((rc = c->cov_e.upload(ce.data(), ce.size(), c->stream))) return rc;
		if ((rc = c->cut_lo.upload(cl.data(), cl.size(), c->stream))) return rc;
		if ((rc = c->chrom_first_bucket.upload(cfb.data(), cfb.size(), c->stream))) return rc;
		std::vector<unsigned> bb(E->buckets.size() + 1, 0);
		for (size_t b = 0; b < E->buckets.size(); ++b) bb[b + 1] = bb[b] + E->buckets[b].n_bins;
		c->n_fine = bb.back();
		if ((rc = c->bin_base.upload(bb.data(), bb.size(), c->stream))) return rc;
	}
	if ((rc = upload_strand_ranks(c))) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return LSQ_OK;
}

// Runs the three ingest kernels over parsed blocks that are already on the device (file order).
static int ingest_device(lsq_ctx *c, int method, const IngestRaw &Rw, uint64_t nblk) {
	const lsq_events &E = *c->E;
	MethodReads &mr = c->reads[method];
	mr.present = false;
	const unsigned B = (unsigned)E.buckets.size();
	const uint64_t n = Rw.n_reads;
	hipStream_t st = c->stream;
	int rc;
	DevBuf<int> d_ms, d_me;
	DevBuf<unsigned char> d_nb, d_strand;
	DevBuf<unsigned> d_key, d_fine;
	DevBuf<unsigned> d_cnt;                      // cnt1 | cnt2 | cntn | cntnb, then the four cursor arrays
	DevBuf<unsigned long long> d_off1, d_off2, d_totals;
	const size_t F = c->n_fine;                  // bins of all buckets
	const size_t n_cnt = 2 * F + 2 * (size_t)B;
	if ((rc = d_ms.alloc(nblk)) || (rc = d_me.alloc(nblk)) || (rc = d_nb.alloc(n)) || (rc = d_strand.alloc(n)) || (rc = d_key.alloc(n)) || (rc = d_fine.alloc(n))) return rc;
	if ((rc = d_cnt.alloc(2 * n_cnt)) || (rc = d_off1.alloc(F + 1)) || (rc = d_off2.alloc(F + 1)) || (rc = d_totals.alloc(4))) return rc;
	HIP_TRY(hipMemsetAsync(d_cnt.p, 0, std::max<size_t>(2 * n_cnt, 1) * 4, st));
	HIP_TRY(hipMemsetAsync(d_totals.p, 0, 4 * 8, st));
	IngestTables T;
	T.cov_off = c->cov_off.p; T.cov_s = c->cov_s.p; T.cov_e = c->cov_e.p;
	T.cut_off = c->cut_off.p; T.cut_lo = c->cut_lo.p; T.chrom_first_bucket = c->chrom_first_bucket.p;
	T.buckets = c->buckets.p; T.bin_base = c->bin_base.p; T.n_chrom = c->n_chrom_tables;
	IngestWork W;
	W.key = d_key.p; W.fine = d_fine.p; W.nb = d_nb.p; W.strand = d_strand.p; W.ms = d_ms.p; W.me = d_me.p;
	W.cnt1 = d_cnt.p; W.cnt2 = W.cnt1 + F; W.cntn = W.cnt2 + F; W.cntnb = W.cntn + B;
	W.cur1 = d_cnt.p + n_cnt; W.cur2 = W.cur1 + F; W.curn = W.cur2 + F; W.curnb = W.curn + B;
	W.totals = d_totals.p;
	const unsigned igrid = (unsigned)std::min<unsigned long long>((n + 255) / 256 + 1, (unsigned long long)c->n_cu * 16);
	if (n) {
		hipLaunchKernelGGL(lsq_ingest_classify_kernel, dim3(igrid), dim3(256), 0, st, T, Rw, W);
		HIP_TRY(hipGetLastError());
	}
	if ((rc = mr.p1_off.alloc(B + 1)) || (rc = mr.p2_off.alloc(B + 1)) || (rc = mr.pn_off.alloc(B + 1)) || (rc = mr.pnb_off.alloc(B + 1)) || (rc = mr.slot_off.alloc(B + 1))) return rc;
	hipLaunchKernelGGL(lsq_scan_u32_kernel, dim3(1), dim3(1024), 0, st, W.cnt1, (unsigned long long)F, d_off1.p);
	hipLaunchKernelGGL(lsq_scan_u32_kernel, dim3(1), dim3(1024), 0, st, W.cnt2, (unsigned long long)F, d_off2.p);
	hipLaunchKernelGGL(lsq_scan_u32_kernel, dim3(1), dim3(1024), 0, st, W.cntn, (unsigned long long)B, mr.pn_off.p);
	hipLaunchKernelGGL(lsq_scan_u32_kernel, dim3(1), dim3(1024), 0, st, W.cntnb, (unsigned long long)B, mr.pnb_off.p);
	hipLaunchKernelGGL(lsq_ingest_offsets_kernel, dim3(B / 256 + 1), dim3(256), 0, st, c->bin_base.p, B, d_off1.p, d_off2.p, mr.pn_off.p,
	                   mr.p1_off.p, mr.p2_off.p, mr.slot_off.p);
	HIP_TRY(hipGetLastError());
	unsigned long long tot[4] = {0, 0, 0, 0}, sums[4] = {0, 0, 0, 0};
	HIP_TRY(hipMemcpyAsync(tot, d_totals.p, 4 * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&sums[0], mr.p1_off.p + B, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&sums[1], mr.p2_off.p + B, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&sums[2], mr.pn_off.p + B, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&sums[3], mr.pnb_off.p + B, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (tot[2]) return fail(LSQ_E_RANGE, "a read covers 2^18 or more bases or keeps more than %d separate blocks: outside the device tables' range", INGEST_MAX_BLOCKS);
	if (sums[3] > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "too many blocks in multi-block reads");
	if (n > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "more than 2^32 reads in one file");
	const size_t n1 = (size_t)sums[0], n2 = (size_t)sums[1], nn = (size_t)sums[2], nnb = (size_t)sums[3];
	if ((rc = mr.p1.alloc(2 * n1)) || (rc = mr.p1_strand.alloc(n1)) || (rc = mr.p1_line.alloc(n1))) return rc;
	if ((rc = mr.p2.alloc(4 * n2)) || (rc = mr.p2_strand.alloc(n2)) || (rc = mr.p2_line.alloc(n2))) return rc;
	if ((rc = mr.pn_se.alloc(2 * nnb)) || (rc = mr.pn_blk_off.alloc(nn)) || (rc = mr.pn_nblk.alloc(nn)) || (rc = mr.pn_strand.alloc(nn)) ||
	    (rc = mr.pn_line.alloc(nn)) || (rc = mr.pn_bucket.alloc(nn))) return rc;
	if (n) {
		// the scatter fills temporaries; the per-bin sort writes the pools
		DevBuf<int32_t> t_p1, t_p2;
		DevBuf<uint8_t> t_p1_strand, t_p2_strand;
		DevBuf<uint32_t> t_p1_line, t_p2_line;
		if ((rc = t_p1.alloc(2 * n1)) || (rc = t_p1_strand.alloc(n1)) || (rc = t_p1_line.alloc(n1))) return rc;
		if ((rc = t_p2.alloc(4 * n2)) || (rc = t_p2_strand.alloc(n2)) || (rc = t_p2_line.alloc(n2))) return rc;
		IngestOut O;
		O.p1 = reinterpret_cast<int2 *>(t_p1.p); O.p1_strand = t_p1_strand.p; O.p1_line = t_p1_line.p;
		O.p2 = reinterpret_cast<int4 *>(t_p2.p); O.p2_strand = t_p2_strand.p; O.p2_line = t_p2_line.p;
		O.pn_blk_off = mr.pn_blk_off.p; O.pn_nblk = mr.pn_nblk.p; O.pn_line = mr.pn_line.p; O.pn_bucket = mr.pn_bucket.p;
		O.pn_strand = mr.pn_strand.p; O.pn_se = reinterpret_cast<int2 *>(mr.pn_se.p);
		O.off1 = d_off1.p; O.off2 = d_off2.p; O.pn_off = mr.pn_off.p; O.pnb_off = mr.pnb_off.p;
		hipLaunchKernelGGL(lsq_ingest_scatter_kernel, dim3(igrid), dim3(256), 0, st, Rw, W, O);
		HIP_TRY(hipGetLastError());
		const unsigned sgrid = (unsigned)std::min<size_t>(F / 4 + 1, (size_t)c->n_cu * 32);
		if (n1) hipLaunchKernelGGL(lsq_ingest_binsort_kernel<int2>, dim3(sgrid), dim3(256), 0, st, c->buckets.p, c->bin_base.p, B, (unsigned)F, d_off1.p,
		                           reinterpret_cast<const int2 *>(t_p1.p), t_p1_strand.p, t_p1_line.p, reinterpret_cast<int2 *>(mr.p1.p), mr.p1_strand.p, mr.p1_line.p);
		if (n2) hipLaunchKernelGGL(lsq_ingest_binsort_kernel<int4>, dim3(sgrid), dim3(256), 0, st, c->buckets.p, c->bin_base.p, B, (unsigned)F, d_off2.p,
		                           reinterpret_cast<const int4 *>(t_p2.p), t_p2_strand.p, t_p2_line.p, reinterpret_cast<int4 *>(mr.p2.p), mr.p2_strand.p, mr.p2_line.p);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(st));            // the temporaries go out of scope here
	}
	{
		// exception list: a quarter of the one- and two-block reads, at least 64 Ki entries
		const size_t want = std::max<size_t>(65536, (n1 + n2) / 4);
		if (c->exc.n < want && (rc = c->exc.alloc(want))) return rc;
	}
	if ((rc = upload_strand_ranks(c))) return rc;      // the reads may have introduced new strand strings
	HIP_TRY(hipStreamSynchronize(st));
	mr.n_retained = tot[0];
	mr.n_retained_blocks = tot[1];
	mr.total_slots = n1 + n2 + nn;
	mr.wg_grid = 0;
	{
		// how unevenly the reads fall on the buckets: with hot genes the reads that need the general walk
		// fill whole workgroup shares, and smaller shares (more workgroups) even the load out
		std::vector<unsigned long long> so(B + 1, 0);
		HIP_TRY(hipMemcpy(so.data(), mr.slot_off.p, (B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		unsigned long long mx = 0;
		for (unsigned b = 0; b < B; ++b) mx = std::max(mx, so[b + 1] - so[b]);
		mr.skew = (B && mr.total_slots) ? (double)mx * (double)B / (double)mr.total_slots : 1.0;
	}
	mr.present = true;
	c->counted = c->solved = false;
	return LSQ_OK;
}

int lsq_reads_upload(lsq_ctx *c, int method, const lsq_reads *R) {
	if (!c || !R) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (method < 0 || method >= c->E->n_methods) return fail(LSQ_E_ARG, "method %d out of range", method);
	HIP_TRY(hipSetDevice(c->device));
	const uint64_t n = R->n_reads, nblk = R->n_blocks;
	hipStream_t st = c->stream;
	int rc;
	// the parsed blocks, file order
	DevBuf<unsigned long long> d_blk_off;
	DevBuf<unsigned> d_line;
	DevBuf<int> d_bs, d_be;
	DevBuf<unsigned short> d_bc;
	DevBuf<unsigned char> d_bst;
	const unsigned long long zero_off = 0;
	if ((rc = d_blk_off.upload(n ? (const unsigned long long *)R->blk_off : &zero_off, n + 1, st))) return rc;
	if ((rc = d_line.upload(R->line_no, n, st))) return rc;
	if ((rc = d_bs.upload(R->blk_start, nblk, st))) return rc;
	if ((rc = d_be.upload(R->blk_end, nblk, st))) return rc;
	if ((rc = d_bc.upload(R->blk_chrom, nblk, st))) return rc;
	if ((rc = d_bst.upload(R->blk_strand, nblk, st))) return rc;
	IngestRaw Rw;
	Rw.n_reads = n; Rw.blk_off = d_blk_off.p; Rw.line_no = d_line.p; Rw.blk_start = d_bs.p; Rw.blk_end = d_be.p;
	Rw.blk_chrom = d_bc.p; Rw.blk_strand = d_bst.p;
	if ((rc = ingest_device(c, method, Rw, nblk))) return rc;
	MethodReads &mr = c->reads[method];
	mr.named = R->named;
	if (R->named) {
		if ((rc = mr.names.upload(R->name_blob.data(), R->name_blob.size(), st))) return rc;
		if ((rc = mr.name_off.upload((const unsigned long long *)R->name_off.data(), R->name_off.size(), st))) return rc;
		HIP_TRY(hipStreamSynchronize(st));
	}
	return LSQ_OK;
}

int lsq_reads_upload_mrf(lsq_ctx *c, int method, const char *read_format, const char *path) {
	if (!c) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (method < 0 || method >= c->E->n_methods) return fail(LSQ_E_ARG, "method %d out of range", method);
	HIP_TRY(hipSetDevice(c->device));
	DevParsed P;
	int rc = device_parse_mrf(c, read_format, path, P, &c->mrf_h2d_ms, &c->mrf_parse_ms);
	if (rc) return rc;
	IngestRaw Rw;
	Rw.n_reads = P.n_reads; Rw.blk_off = P.blk_off.p; Rw.line_no = P.line_no.p; Rw.blk_start = P.bs.p; Rw.blk_end = P.be.p;
	Rw.blk_chrom = P.bc.p; Rw.blk_strand = P.bst.p;
	c->reads[method].named = false;
	return ingest_device(c, method, Rw, P.n_blocks);
}

int lsq_mrf_parse_device(lsq_ctx *c, const char *read_format, const char *path, lsq_reads **out) {
	if (!c || !out) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	DevParsed P;
	int rc = device_parse_mrf(c, read_format, path, P, &c->mrf_h2d_ms, &c->mrf_parse_ms);
	if (rc) return rc;
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
}

int lsq_last_mrf_timing(lsq_ctx *c, float *h2d_ms, float *parse_ms) {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (h2d_ms) *h2d_ms = c->mrf_h2d_ms;
	if (parse_ms) *parse_ms = c->mrf_parse_ms;
	return LSQ_OK;
}

uint64_t lsq_reads_retained(const lsq_ctx *c, int method) { return (c && method >= 0 && method < LSQ_MAX_METHODS) ? c->reads[method].n_retained : 0; }
uint64_t lsq_reads_retained_blocks(const lsq_ctx *c, int method) { return (c && method >= 0 && method < LSQ_MAX_METHODS) ? c->reads[method].n_retained_blocks : 0; }

static int run_count(lsq_ctx *c, bool all_reads) {
	const lsq_events &E = *c->E;
	const size_t n_cls = E.n_cls_total;
	const int M = E.n_methods;
	hipStream_t st = c->stream;
	HIP_TRY(hipMemsetAsync(c->counters.p, 0, c->counters.n * sizeof(unsigned long long), st));
	c->fast_launched = 0;
	HIP_TRY(hipEventRecord(c->ev0, st));      // ev0..ev1 brackets the count kernel launches only
	const unsigned generic_tables_bytes = (std::max<unsigned>(E.max_lds_bytes, 16) + 15u) & ~15u;
	const unsigned tables_bytes = generic_tables_bytes + VISIT_LDS_BYTES;            // fast kernel: + the visit record
	const unsigned lds_bytes = tables_bytes + WAVES * WAVE_QUEUE_WORDS * 16;
	if (lds_bytes > 160 * 1024) return fail(LSQ_E_UNSUPPORTED, "bucket tables + read tile exceed the CU's LDS");
	if (lds_bytes > 64 * 1024) {
		HIP_TRY(hipFuncSetAttribute((const void *)lsq_count_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
		HIP_TRY(hipFuncSetAttribute((const void *)lsq_count_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
	}
	unsigned per_cu = std::max(1u, std::min(2048u / COUNT_BLOCK, (160u * 1024u) / lds_bytes));
	int mult_env = 0;
	if (const char *e = getenv("LSQ_GRID_MULT")) { int v = atoi(e); if (v >= 1 && v <= 64) mult_env = v; }
	for (int m = 0; m < M; ++m) {
		MethodReads &mr = c->reads[m];
		if (mr.total_slots == 0 || E.buckets.empty()) continue;
		// workgroups per resident slot: 2 for even read depth (fewest table stagings), more when a few buckets
		// hold most of the reads (measured on the skewed workload: 2 -> 0.48 ms, 8 -> 0.30 ms)
		const unsigned mult = mult_env ? (unsigned)mult_env : (mr.skew >= 32.0 ? 8u : (mr.skew >= 4.0 ? 4u : 2u));
		unsigned long long grid = (unsigned long long)c->n_cu * per_cu * mult;
		// one workgroup's share must keep the packed LDS counters (24-bit count, 40-bit bases) exact
		grid = std::max(grid, mr.total_slots / (1ull << 21) + 1);
		grid = std::min<unsigned long long>(grid, std::max<unsigned long long>(mr.total_slots / 64, 1));
		if (mr.wg_grid != grid) {
			int rc = mr.wg_first.alloc((size_t)grid);
			if (rc) return rc;
			hipLaunchKernelGGL(lsq_wg_plan_kernel, dim3((unsigned)(grid / 256 + 1)), dim3(256), 0, st, mr.slot_off.p, (unsigned)E.buckets.size(), mr.total_slots,
			                   (unsigned)grid, mr.wg_first.p);
			HIP_TRY(hipGetLastError());
			mr.wg_grid = grid;
		}
		CountArgs A;
		A.buckets = c->buckets.p; A.images = c->images.p; A.ties = c->ties.p; A.strand_rank = c->strand_rank.p;
		A.wg_first = mr.wg_first.p;
		A.read_names = mr.named ? mr.names.p : nullptr; A.read_name_off = mr.named ? mr.name_off.p : nullptr;
		A.gene_names = c->gene_names.p; A.gene_name_off = c->gene_name_off.p;
		A.n_buckets = (unsigned)E.buckets.size();
		A.tables_lds_bytes = tables_bytes;
		A.ablate = 0;
		if (const char *e = getenv("LSQ_ABLATE")) A.ablate = (unsigned)atoi(e);
		A.p1 = reinterpret_cast<const int2 *>(mr.p1.p); A.p1_strand = mr.p1_strand.p; A.p1_line = mr.p1_line.p;
		A.p2 = reinterpret_cast<const int4 *>(mr.p2.p); A.p2_strand = mr.p2_strand.p; A.p2_line = mr.p2_line.p;
		A.pn_blk_off = mr.pn_blk_off.p; A.pn_nblk = mr.pn_nblk.p; A.pn_se = reinterpret_cast<const int2 *>(mr.pn_se.p);
		A.pn_strand = mr.pn_strand.p; A.pn_line = mr.pn_line.p; A.pn_bucket = mr.pn_bucket.p;
		A.p1_off = mr.p1_off.p; A.p2_off = mr.p2_off.p; A.pn_off = mr.pn_off.p; A.slot_off = mr.slot_off.p;
		A.total_slots = mr.total_slots;
		A.cnt = c->cnt.p + (size_t)m * n_cls; A.bases = c->bases.p + (size_t)m * n_cls;
		A.exc = c->exc.p; A.exc_count = c->exc_count.p + 2 * m; A.exc_cap = (unsigned)c->exc.n;
		A.dbg = c->dbg.p;
		const unsigned long long n_p1 = mr.p1.n / 2, n_p2 = mr.p2.n / 4, n_pn = mr.pn_strand.n;
		// pool-n workers: one workgroup per CU at most, one lane per read and pass
		A.n_pn = n_pn;
		const unsigned workers_per_cu = 2;          // 1, 4 and 8 measured within 2 % of each other
		A.n_workers = (unsigned)std::min<unsigned long long>((n_pn + COUNT_BLOCK - 1) / COUNT_BLOCK, (unsigned long long)c->n_cu * workers_per_cu);
		if (c->has_fast) {
			if (!all_reads) {
				HIP_TRY(hipEventRecord(c->evf0[m], st));
				hipLaunchKernelGGL(lsq_count_fast_kernel, dim3((unsigned)grid + A.n_workers), dim3(COUNT_BLOCK), lds_bytes, st, A);
				HIP_TRY(hipGetLastError());
				HIP_TRY(hipEventRecord(c->evf1[m], st));
				c->fast_launched |= 1 << m;
			}
			const unsigned long long work = all_reads ? std::max<unsigned long long>(n_pn, 64ull * E.buckets.size()) : 4096ull;
			const unsigned cgrid = (unsigned)std::min<unsigned long long>((work + 255) / 256, (unsigned long long)c->n_cu * 32);
			hipLaunchKernelGGL(lsq_count_cleanup_kernel, dim3(std::max(cgrid, 1u)), dim3(256), 0, st, A, n_p1, n_p2, n_pn, all_reads ? 1 : 0);
			HIP_TRY(hipGetLastError());
		}
		if (c->has_generic) {
			hipLaunchKernelGGL(lsq_count_generic_kernel, dim3((unsigned)grid), dim3(COUNT_BLOCK), generic_tables_bytes, st, A);
			HIP_TRY(hipGetLastError());
		}
	}
	HIP_TRY(hipEventRecord(c->ev1, st));
	return LSQ_OK;
}

static int run_solve(lsq_ctx *c);

// After a synchronisation point: did any method's exception list overflow?  Then the counts
// (and a solve based on them) are redone with the cleanup kernel over every read.
static int ensure_counts_complete(lsq_ctx *c) {
	if (c->redo_checked || !c->counted) return LSQ_OK;
	HIP_TRY(hipStreamSynchronize(c->stream));
	std::vector<unsigned> h(c->exc_count.n, 0);
	HIP_TRY(hipMemcpy(h.data(), c->exc_count.p, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
	bool overflow = false;
	for (size_t m = 0; m * 2 + 1 < h.size(); ++m) overflow = overflow || h[2 * m + 1] != 0;
	c->redo_checked = true;
	if (overflow || getenv("LSQ_FORCE_REDO")) {
		int rc = run_count(c, true);
		if (rc) return rc;
		if (c->solved) { rc = run_solve(c); if (rc) return rc; }
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	return LSQ_OK;
}

int lsq_count(lsq_ctx *c) {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	HIP_TRY(hipSetDevice(c->device));
	for (int m = 0; m < c->E->n_methods; ++m) if (!c->reads[m].present) return fail(LSQ_E_STATE, "reads of method %d were not uploaded", m);
	int rc = run_count(c, false);
	if (rc) return rc;
	c->counted = true;
	c->solved = false;
	c->redo_checked = false;
	return LSQ_OK;
}

static int run_solve(lsq_ctx *c) {
	const lsq_events &E = *c->E;
	hipStream_t st = c->stream;
	HIP_TRY(hipEventRecord(c->ev2, st));
	const unsigned n_ev = (unsigned)E.dev2out.size();
	if (n_ev) {
		EmArgs A;
		A.n_events = n_ev; A.n_methods = (unsigned)E.n_methods; A.n_cls = E.n_cls_total; A.n_iso = E.n_iso_total;
		A.K = c->dK.p; A.cls_base = c->cls_base.p; A.iso_base = c->iso_base.p;
		A.n_places = c->em_places; A.order = c->em_order.p;
		A.cnt = c->cnt.p; A.G = c->G.p; A.theta = c->theta.p; A.logll = c->logll.p; A.iters = c->iters.p; A.flags = c->flags.p;
		hipLaunchKernelGGL(lsq_em_kernel, dim3((c->em_places * EM_LANES + 255) / 256), dim3(256), 0, st, A);
		HIP_TRY(hipGetLastError());
	}
	HIP_TRY(hipEventRecord(c->ev3, st));
	return LSQ_OK;
}

int lsq_solve(lsq_ctx *c) {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	HIP_TRY(hipSetDevice(c->device));
	int rc = run_solve(c);
	if (rc) return rc;
	c->solved = true;
	return LSQ_OK;
}

int64_t lsq_results_num_classes(const lsq_ctx *c) { return (c && c->E) ? (int64_t)c->E->class_off.back() : 0; }

int lsq_results_class_offsets(const lsq_ctx *c, uint64_t *class_off) {
	if (!c || !c->E || !class_off) return fail(LSQ_E_ARG, "null argument");
	memcpy(class_off, c->E->class_off.data(), c->E->class_off.size() * sizeof(uint64_t));
	return LSQ_OK;
}

int lsq_results_counts(lsq_ctx *c, uint64_t *class_count, uint64_t *class_bases) {
	if (!c || !class_count) return fail(LSQ_E_ARG, "null argument");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = ensure_counts_complete(c); if (rc) return rc; }
	const lsq_events &E = *c->E;
	// device arrays hold the classes of this process's events (device order); the caller's arrays
	// hold every selected event's classes (output order), zero outside the shard
	const size_t n_cls = E.n_cls_total, n_out = (size_t)E.class_off.back(), M = (size_t)E.n_methods;
	std::vector<unsigned long long> hc(std::max<size_t>(M * n_cls, 1)), hb(std::max<size_t>(M * n_cls, 1));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (M * n_cls) {
		HIP_TRY(hipMemcpy(hc.data(), c->cnt.p, M * n_cls * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hb.data(), c->bases.p, M * n_cls * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	}
	memset(class_count, 0, M * n_out * sizeof(uint64_t));
	if (class_bases) memset(class_bases, 0, M * n_out * sizeof(uint64_t));
	for (size_t d = 0; d < E.dev2out.size(); ++d) {
		const size_t o = (size_t)E.dev2out[d];
		const size_t nc = (1u << E.ev[o].K) - 1u;
		for (size_t m = 0; m < M; ++m)
			for (size_t k = 0; k < nc; ++k) {
				class_count[m * n_out + E.class_off[o] + k] = hc[m * n_cls + E.dev_cls_base[d] + k];
				if (class_bases) class_bases[m * n_out + E.class_off[o] + k] = hb[m * n_cls + E.dev_cls_base[d] + k];
			}
	}
	return LSQ_OK;
}

int lsq_results_solve(lsq_ctx *c, double *theta, double *logll, uint32_t *em_iters, uint8_t *em_flags) {
	if (!c || !theta || !logll) return fail(LSQ_E_ARG, "null argument");
	if (!c->solved) return fail(LSQ_E_STATE, "lsq_solve must come first");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = ensure_counts_complete(c); if (rc) return rc; }
	const lsq_events &E = *c->E;
	const size_t n_ev = E.dev2out.size(), n_iso = E.n_iso_total;
	std::vector<double> ht(std::max<size_t>(n_iso, 1)), hl(std::max<size_t>(n_ev, 1));
	std::vector<uint32_t> hi(std::max<size_t>(n_ev, 1));
	std::vector<uint8_t> hf(std::max<size_t>(n_ev, 1));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (n_ev) {
		HIP_TRY(hipMemcpy(ht.data(), c->theta.p, n_iso * sizeof(double), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hl.data(), c->logll.p, n_ev * sizeof(double), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hi.data(), c->iters.p, n_ev * sizeof(uint32_t), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hf.data(), c->flags.p, n_ev * sizeof(uint8_t), hipMemcpyDeviceToHost));
	}
	for (size_t o = 0; o < E.ev.size(); ++o) {        // events outside the shard: zeros
		for (int j = 0; j < E.ev[o].K; ++j) theta[E.iso_off[o] + j] = 0.0;
		logll[o] = 0.0;
		if (em_iters) em_iters[o] = 0;
		if (em_flags) em_flags[o] = 0;
	}
	for (size_t d = 0; d < n_ev; ++d) {
		const size_t o = (size_t)E.dev2out[d];
		for (int j = 0; j < E.ev[o].K; ++j) theta[E.iso_off[o] + j] = ht[E.dev_iso_base[d] + j];
		logll[o] = hl[d];
		if (em_iters) em_iters[o] = hi[d];
		if (em_flags) em_flags[o] = hf[d];
	}
	return LSQ_OK;
}

} // extern "C"

namespace {
// the three result arrays to the caller's device buffers in one launch (8-byte words)
__global__ void __launch_bounds__(256) lsq_copy_results_kernel(unsigned long long *d0, const unsigned long long *s0, size_t n0, unsigned long long *d1,
                                                               const unsigned long long *s1, size_t n1, unsigned long long *d2, const unsigned long long *s2, size_t n2) {
	const size_t gsz = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1 + n2; i += gsz) {
		if (i < n0) d0[i] = s0[i];
		else if (i < n0 + n1) d1[i - n0] = s1[i - n0];
		else d2[i - n0 - n1] = s2[i - n0 - n1];
	}
}
} // namespace

extern "C" {

int lsq_results_copy_device(lsq_ctx *c, void *d_class_count, void *d_theta, void *d_logll) {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	if ((d_theta || d_logll) && !c->solved) return fail(LSQ_E_STATE, "lsq_solve must come first");
	HIP_TRY(hipSetDevice(c->device));
	const lsq_events &E = *c->E;
	const size_t n_cls = E.n_cls_total, M = (size_t)E.n_methods;
	const size_t n0 = d_class_count ? M * n_cls : 0, n1 = d_theta ? (size_t)E.n_iso_total : 0, n2 = d_logll ? E.dev2out.size() : 0;
	if (n0 + n1 + n2) {
		const unsigned grid = (unsigned)std::min<size_t>((n0 + n1 + n2 + 255) / 256, (size_t)c->n_cu * 8);
		hipLaunchKernelGGL(lsq_copy_results_kernel, dim3(grid), dim3(256), 0, c->stream, (unsigned long long *)d_class_count, c->cnt.p, n0,
		                   (unsigned long long *)d_theta, (const unsigned long long *)c->theta.p, n1, (unsigned long long *)d_logll, (const unsigned long long *)c->logll.p, n2);
		HIP_TRY(hipGetLastError());
	}
	return LSQ_OK;
}

int lsq_results_device_order(const lsq_ctx *c, int32_t *dev2out) {
	if (!c || !c->E || !dev2out) return fail(LSQ_E_ARG, "null argument");
	memcpy(dev2out, c->E->dev2out.data(), c->E->dev2out.size() * sizeof(int32_t));
	return LSQ_OK;
}

// developer aid (not in the header): counters filled when LSQ_ABLATE & 256
int lsq_debug_counters(lsq_ctx *c, unsigned long long *out8) {
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(out8, c->dbg.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	std::vector<unsigned> h(c->exc_count.n);
	HIP_TRY(hipMemcpy(h.data(), c->exc_count.p, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
	out8[4] = h[0];
	return LSQ_OK;
}

// developer aid (not in the header): per-bucket slot offsets of a method (n_buckets + 1 values)
int lsq_debug_slot_offsets(lsq_ctx *c, int method, unsigned long long *out, unsigned long long n) {
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	const MethodReads &mr = c->reads[method];
	if (n > mr.slot_off.n) n = mr.slot_off.n;
	HIP_TRY(hipMemcpy(out, mr.slot_off.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return LSQ_OK;
}

int lsq_last_fast_kernel_ms(lsq_ctx *c, float *ms) {
	if (!c || !ms) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	*ms = 0;
	for (int m = 0; m < LSQ_MAX_METHODS; ++m) if (c->fast_launched >> m & 1) { float t = 0; HIP_TRY(hipEventElapsedTime(&t, c->evf0[m], c->evf1[m])); *ms += t; }
	return LSQ_OK;
}

int lsq_last_timing(lsq_ctx *c, float *count_ms, float *solve_ms) {
	if (!c) return fail(LSQ_E_ARG, "null context");
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (count_ms) { *count_ms = 0; if (c->counted) HIP_TRY(hipEventElapsedTime(count_ms, c->ev0, c->ev1)); }
	if (solve_ms) { *solve_ms = 0; if (c->solved) HIP_TRY(hipEventElapsedTime(solve_ms, c->ev2, c->ev3)); }
	return LSQ_OK;
}

} // extern "C"
