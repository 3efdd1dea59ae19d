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
#include "lsq_device.hpp"
#include <hip/hip_ext.h>

namespace {

#ifndef LSQ_COUNT_BLOCK
#define LSQ_COUNT_BLOCK 256
#endif
constexpr int COUNT_BLOCK = LSQ_COUNT_BLOCK;       // threads per workgroup of the count kernels
constexpr unsigned long long BASES_MASK = (1ull << 40) - 1;

// Ablation switches of the developer build (-DLSQ_DEV: liblesseq_hip_dev.so, tools/kbench.py): parts of the kernels
// can be switched off to see what they cost.  The release object carries none of these branches.
#ifdef LSQ_DEV
#define ABL(args, bits) ((((args).ablate) & (bits)) != 0u)
#else
#define ABL(args, bits) false
#endif


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
	// Pools of one- and two-block reads, bucket by bucket.  Wide records: (start, end) and (start, end, start, end), 8 bytes
	// a block.  Compact records (lsq_device.hpp COMPACT_*): 4 bytes a block, relative to the bucket; reads that do not fit
	// sit with the many-block reads.
	const void *p1; const unsigned char *p1_strand; const unsigned *p1_line;
	const void *p2; const unsigned char *p2_strand; const unsigned *p2_line;
	unsigned compact;
	const unsigned *pn_blk_off; const unsigned *pn_nblk; const int2 *pn_se; const unsigned char *pn_strand; const unsigned *pn_line; const unsigned *pn_bucket;
	const unsigned long long *p1_off, *p2_off, *pn_off, *slot_off;    // n_buckets + 1 each
	const lsq::VisitRec *visits;       // per bucket: what a visit needs, in one record (n_buckets + 1)
	const unsigned *wg_first;          // per workgroup of the fast kernel's grid: the first packed bucket that holds slots of its share (n_buckets: none)
	const unsigned long long *wg_cut;  // ... and the ranges' bounds in slots (grid + 1 values)
	const lsq::WgPlan *wg_plan;        // both, per workgroup, with the first bucket's visit record (what the kernel reads)
	unsigned long long *wg_trace;      // developer build: (start, end) of every workgroup, 100 MHz clock (null: none)
	unsigned long long total_slots;
	unsigned long long n_pn;           // reads with three or more blocks
	unsigned n_workers;                // leading workgroups of the fast kernel's grid that take them
	unsigned long long *cnt, *bases;
	struct ExcEntry *exc;              // exception list (rare (read, event) pairs the fast kernel hands to the cleanup kernel)
	unsigned *exc_count;               // [0] entries appended, [1] set to 1 by the cleanup kernel when [0] > exc_cap
	unsigned exc_cap;
	unsigned long long *dbg;            // developer counters (LSQ_ABLATE & 256): parked one-block, parked two-block, walk steps, walk lanes
	unsigned long long *bar;            // this read file's barrier word in the counter set (zeroed with it): the exception pass's way into the recount
};

// compact records to coordinates; base = the bucket's first base minus COMPACT_BIAS
__device__ inline int2 unpack_one_block(const unsigned a, const int base) {
	const int s = base + (int)(a & lsq::COMPACT_OFF_MASK);
	return make_int2(s, s + (int)(a >> lsq::COMPACT_OFF_BITS));
}
__device__ inline int4 unpack_two_block(const unsigned a, const unsigned b, const int base) {
	const int s1 = base + (int)(a & lsq::COMPACT_OFF_MASK), e1 = s1 + (int)(a >> lsq::COMPACT_OFF_BITS);
	const int s2 = e1 + (int)(b & lsq::COMPACT_OFF_MASK);
	return make_int4(s1, e1, s2, s2 + (int)(b >> lsq::COMPACT_OFF_BITS));
}
// a read of bucket `lo`'s slice of the pools (the slow paths: exception pass, recount, generic buckets)
__device__ inline int2 pool1_read(const CountArgs &A, const unsigned long long g, const int lo) {
	if (!A.compact) return reinterpret_cast<const int2 *>(A.p1)[g];
	return unpack_one_block(reinterpret_cast<const unsigned *>(A.p1)[g], lo - lsq::COMPACT_BIAS);
}
__device__ inline int4 pool2_read(const CountArgs &A, const unsigned long long g, const int lo) {
	if (!A.compact) return reinterpret_cast<const int4 *>(A.p2)[g];
	const uint2 v = reinterpret_cast<const uint2 *>(A.p2)[g];
	return unpack_two_block(v.x, v.y, lo - lsq::COMPACT_BIAS);
}

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
				const int2 rd = pool1_read(A, g, d.lo);
				if (rd.y == rd.x) continue;          // padding of a cell's group
				process_read<1>(L, d, A, make_int4(rd.x, rd.y, 0, 0), nullptr, 1, rd.y - rd.x, A.p1_strand, A.p1_line, g);
			} else if (i < n1 + n2) {
				const unsigned long long g = A.p2_off[b] + (i - n1);
				const int4 rd = pool2_read(A, g, d.lo);
				if (rd.y == rd.x) continue;          // padding of a junction group
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
	unsigned long long *trace;    // developer build: this workgroup's four trace words (null: none)
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
	if (exc && !ABL(C, 128u)) emit_exception(C, r, i, 0u);
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
		if (!ABL(C, 2u)) { if (add) atomicAdd(&C.hist[(w0.y & 0xFFFFu) + cls - 1u], (1ull << 40) | (unsigned long long)(unsigned)matched); }
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
#ifndef LSQ_STREAM_WORDS_P2
#define LSQ_STREAM_WORDS_P2 2
#endif
#ifndef LSQ_STREAM_WORDS
#define LSQ_STREAM_WORDS 2
#endif
#ifndef LSQ_P2_COMPACT_WORDS
#define LSQ_P2_COMPACT_WORDS 2         // two-block reads a lane takes per step from a compact pool (2 or 4; 4 -- one table look for four reads -- measured 3 % slower: the one-block path of the same kernel loses more registers than the look saves)
#endif
#ifndef LSQ_P2_COMPACT_WORDS_W5
#define LSQ_P2_COMPACT_WORDS_W5 4      // the same in the five-wave kernel: with 96 registers the one look for four reads pays (C3, developer builds, same box: 0.1401 -> 0.1364 ms per step)
#endif
constexpr int STREAM_WORDS = LSQ_STREAM_WORDS;                // 16-byte words per lane in flight
#ifndef LSQ_GROUP_WORDS
#define LSQ_GROUP_WORDS 2
#endif
constexpr int GROUP_WORDS = LSQ_GROUP_WORDS;                  // words per lane looked up together (independent chains)
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
	// up to two one-word entries a lane (bit 0 / bit 1 of `bits`), the lanes' entries in lane order
	__device__ inline void push2(const unsigned bits, const unsigned lane, const uint4 ea, const uint4 eb) {
		static_assert(NB == 1 || NB == 2, "");
		const unsigned long long m0 = __ballot((bits & 1u) != 0u), m1 = __ballot((bits & 2u) != 0u);
		if (!(m0 | m1)) return;
		const unsigned long long lt = (1ull << lane) - 1ull;
		const unsigned at = tail + (unsigned)__popcll(m0 & lt) + (unsigned)__popcll(m1 & lt);
		if (bits & 1u) q[at % CAP] = ea;
		if (bits & 2u) q[(at + (bits & 1u)) % CAP] = eb;
		tail += (unsigned)__popcll(m0) + (unsigned)__popcll(m1);
	}
};

// PACKED2: two-block reads parked as their compact record (one ring word: record, event to look at, position) -- `base` unpacks them
template <int NB, bool PACKED2 = false>
__device__ inline void walk_parked(const FastCtx &C, Ring<(NB == 1 || PACKED2) ? 1 : 2> &R, const bool to_empty, const int base = 0) {
	constexpr int RW = (NB == 1 || PACKED2) ? 1 : 2;
	const unsigned lane = threadIdx.x & 63u;
	wave_sync_lds();
#pragma unroll 1
	while (R.live() >= (to_empty ? 1u : 64u)) {
		const unsigned n = min(R.live(), 64u);
		const bool on = lane < n;
		if (ABL(C, 256u) && lane == 0) { atomicAdd(&C.dbg[2], 1ull); atomicAdd(&C.dbg[3], (unsigned long long)n); }
		if (ABL(C, 4194304u) && lane == 0 && C.trace) { atomicAdd(&C.trace[2], 1ull); atomicAdd(&C.trace[3], (unsigned long long)n); }
		const unsigned at = (R.head + (on ? lane : 0u)) % Ring<RW>::CAP;
		uint4 e0, e1 = make_uint4(0, 0, 0, 0);
		if (RW == 1) e0 = R.q[at];
		else { e0 = R.q[2 * at]; e1 = R.q[2 * at + 1]; }
		R.head += n;
		int4 rd; unsigned i, rel;
		const unsigned ev_word = RW == 1 ? e0.z : e1.x;
		const bool one_event = (ev_word & PARK_ONE_EVENT) != 0;
		if (RW == 1) { e0.z &= ~PARK_ONE_EVENT; } else { e1.x &= ~PARK_ONE_EVENT; }
		if (NB == 1) {
			rd = make_int4((int)e0.x, (int)e0.y, (int)e0.x, (int)e0.y); i = e0.z; rel = e0.w;
			// parked without a look at the bin directory: the first event of the read's bin
			const int brel = rd.x - C.lo;
			const unsigned bin = brel <= 0 ? 0u : min((unsigned)brel >> C.shift, C.n_bins - 1u);
			const unsigned first = reinterpret_cast<const unsigned *>(C.bins)[4u * bin] >> 16;
			if (i == PARK_EVENT_UNKNOWN) { i = first; e0.z = first; }
		}
		else if (PACKED2) {
			rd = unpack_two_block(e0.x, e0.y, base); i = e0.z; rel = e0.w;
			const int brel = rd.x - C.lo;
			const unsigned bin = brel <= 0 ? 0u : min((unsigned)brel >> C.shift, C.n_bins - 1u);
			const unsigned first = reinterpret_cast<const unsigned *>(C.bins)[4u * bin] >> 16;
			if (i == PARK_EVENT_UNKNOWN) { i = first; e0.z = first; }
		}
		else {
			rd = make_int4((int)e0.x, (int)e0.y, (int)e0.z, (int)e0.w); i = e1.x; rel = e1.y;
			const int brel = rd.x - C.lo;
			const unsigned bin = brel <= 0 ? 0u : min((unsigned)brel >> C.shift, C.n_bins - 1u);
			const unsigned first = reinterpret_cast<const unsigned *>(C.bins)[4u * bin] >> 16;
			if (i == PARK_EVENT_UNKNOWN) { i = first; e1.x = first; }
		}
		const int total = NB == 1 ? rd.y - rd.x : (rd.y - rd.x) + (rd.w - rd.z);
		const bool more = fast_trip<NB>(C, rd, total, rel, i, on) && !one_event && !ABL(C, 64u);
		wave_sync_lds();
		if (RW == 1) e0.z = i + 1u; else e1.x = i + 1u;     // (i is the resolved event)
		R.push(more, lane, e0, e1);
		wave_sync_lds();
	}
}

// RPW = reads per 16-byte word of a wide pool: 2 (pool 1: one block) or 1 (pool 2: two blocks); a lane's words are
// numbered that way for compact pools too (the fetch unpacks one compact word into two of them)
// P1W: words a lane takes from the one-block pool per step, all of them settled by one look at the tables (2 or 4, i.e. four
// or eight reads: the pool's cell groups are padded to eight)
template <int RPW, bool COMPACT, int P1W>
__device__ inline void stream_pool_fast(FastCtx &C, const uint4 *bins, const uint4 *cells, const uint4 *cellx, const unsigned n_cells, const BucketDesc &d,
                                        const CountArgs &A, uint4 *queue, const uint4 *src_generic,
                                        const unsigned long long g0, const unsigned long long g1) {
	constexpr int NB = RPW == 2 ? 1 : 2;
	// Words (16 bytes of a wide pool: two one-block reads, one two-block read) per lane in flight, and looked up together:
	// two-block reads take the whole step as one group -- two of them with wide records, four with compact ones (one table
	// look serves the four: the pool's junction groups are padded to quadruples)
	constexpr int SW = RPW == 2 ? P1W : (COMPACT ? (P1W == 4 ? LSQ_P2_COMPACT_WORDS_W5 : LSQ_P2_COMPACT_WORDS) : LSQ_STREAM_WORDS_P2), GW = RPW == 2 ? (P1W == STREAM_WORDS ? GROUP_WORDS : P1W) : SW;
	static_assert(RPW != 2 || (unsigned)(SW * RPW) <= P1_GROUP_PAD, "a lane's reads of a step lie in one cell group");
	constexpr int CW = COMPACT ? SW / 2 : SW;          // 16-byte loads a lane issues per step
	constexpr bool PACKED2 = COMPACT && RPW == 1;      // two-block reads are parked as their compact records
	constexpr int RW = (NB == 1 || PACKED2) ? 1 : 2;   // ring words per parked read
	constexpr unsigned TILE = 64u * SW;        // words per wave step
	C.pool = RPW == 2 ? 0u : 1u;
	C.slot0 = g0;
	global_words src = (global_words)src_generic;       // kernel-argument memory: global address space
	// (the wave's number through readfirstlane: step counters and range tests then live in scalar registers)
	const unsigned lane = threadIdx.x & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// words [w0, w1) of the workgroup, dealt to its waves a step at a time (wave, wave + 4, ...): the
	// reads that need the general walk sit together in the start-ordered pool, and a contiguous
	// quarter per wave would leave three waves waiting for the one that got them
	// (the range starts on a multiple of a lane's words: they are one of the octuples / quadruples the pools' groups are padded
	// to, and whole 16-byte words of a compact pool)
	const unsigned long long w0 = (g0 / RPW) & ~(unsigned long long)(SW - 1), w1 = (g1 + RPW - 1) / RPW;
	const unsigned n_words = (unsigned)(w1 - w0);                                   // a workgroup's range stays below 2^21 reads
	const unsigned ww0 = min(wave * TILE, n_words), ww1 = n_words;                   // relative to w0
	const unsigned first_rel = (unsigned)(g0 - w0 * RPW);                            // reads of word w0 (and, compact, w0 + 1) before the range
	const unsigned n_rel = (unsigned)(g1 - g0);
	uint4 nxt[SW];
	static_assert(!COMPACT || (SW & 1) == 0, "a compact 16-byte word unpacks into two words of a lane");
	const int base = d.lo - lsq::COMPACT_BIAS;
	auto fetch_into = [&](uint4 (&dst)[SW], unsigned wt) {
		if (COMPACT) {
			// one 16-byte load per lane and step: four one-block records or two two-block ones; unpacked when the
			// step that uses them begins (not here: the load is to stay in flight during the step before)
#pragma unroll
			for (int cq = 0; cq < CW; ++cq) {
				const unsigned w = min((wt >> 1) + lane * (unsigned)CW + (unsigned)cq, (ww1 - 1u) >> 1);
				const u32x4 t = src[(w0 >> 1) + w];
				dst[cq] = make_uint4(t.x, t.y, t.z, t.w);
			}
			return;
		}
		if (ABL(A, RPW == 2 ? 131072u : 262144u)) {      // developer switch: half the words loaded (what a pool of half-size records would cost to stream)
			const unsigned w = min((wt >> 1) + lane, ww1 - 1u);
			const u32x4 t = src[w0 + w];
			dst[0] = make_uint4(t.x, t.y, t.z, t.w);
#pragma unroll
			for (int k = 1; k < SW; ++k) dst[k] = make_uint4(t.x + 1u, t.y + 1u, t.z + 1u, t.w + 1u);
			return;
		}
#pragma unroll
		for (int k = 0; k < SW; ++k) {
			// a lane's words are neighbours in the pool; past the end of the range the last word is read again
			// (no predication: every read is tested against the range before it counts)
			const unsigned w = min(wt + lane * (unsigned)SW + (unsigned)k, ww1 - 1u);
			const u32x4 t = src[w0 + w];
			dst[k] = make_uint4(t.x, t.y, t.z, t.w);
		}
	};
	auto fetch = [&](unsigned wt) { fetch_into(nxt, wt); };
	// bin record of position p -> (the last cell that starts at or before p, first event of the bin).  The record
	// names the bin's first cell and the ends of it and of the next two; in a bin with more cells the search goes on
	// through the cell table (rare: bins are laid out for about one event each).
	auto locate = [&](int p, unsigned &cell, unsigned &first_event) {
		const int rel = p - d.lo;
		unsigned bin = rel <= 0 ? 0u : ((unsigned)rel >> d.shift);
		const uint4 br = bins[min(bin, d.n_bins - 1u)];     // first cell | first event << 16, ends of that cell and the next two
		cell = (br.x & 0xFFFFu) + (unsigned)(p >= (int)br.y) + (unsigned)(p >= (int)br.z) + (unsigned)(p >= (int)br.w);
		first_event = br.x >> 16;
		if (__any(p >= (int)br.w) && !ABL(A, 65536u)) {
			while (cell + 1u < n_cells && p >= (int)cells[cell + 1u].x) ++cell;
		}
	};
	Ring<RW> R;
	R.q = queue;

	// one step of the wave over the words in `cur` (the step's words, fetched a step ahead)
	uint4 raw2[PACKED2 ? CW : 1];        // the step's two-block records as loaded (what is parked of them)
	auto do_step = [&](const uint4 (&cur)[SW], const unsigned wt) {
#pragma unroll
		for (int k0 = 0; k0 < SW; k0 += GW) {
		if (ABL(A, 512u)) {      // developer switch: stream only
#pragma unroll
			for (int kg = 0; kg < GW; ++kg) asm volatile("" ::"v"(cur[k0 + kg].x), "v"(cur[k0 + kg].y), "v"(cur[k0 + kg].z), "v"(cur[k0 + kg].w));
			continue;
		}
		// the reads of a group are looked up first (independent chains), parking comes after
		constexpr int N_READS = GW * RPW;
		bool park[N_READS];
		uint4 pe0[N_READS], pe1[N_READS];
#pragma unroll
		for (int kg = 0; kg < GW; ++kg) {
			if (RPW == 2) {
				// One look at the tables per lane and group: the lane's reads are neighbours in the
				// start-ordered pool, so the cell of the first one is the cell of (nearly) all of
				// them.  A read is decided against that cell -- inside it: the owners' slots; running
				// into the owner's next segment: the two-segment slot -- and the lane adds its totals
				// once.  Everything else (a different cell, no cell, a longer run) is parked -- and only
				// then is anything built for the ring: most steps park nothing in any lane.
				if (kg == 0) {
					// the loop's coordinates: the records' offsets from `base` (compact) or the reads' own (wide)
					const int ws = COMPACT ? base : 0;
					unsigned ci, evf;
					locate((int)cur[k0].x + ws, ci, evf);
					const unsigned cc = min(ci, n_cells - 1u);
					const uint4 cw = cells[cc];           // lo, hi, e1, e2
					const uint4 cx = cellx[cc];           // slots, info, flags, owner event
					const bool has = ci < n_cells && !ABL(A, 8u);
					const int lo = (int)cw.x - ws, e1 = (int)cw.z - ws, e2 = (int)cw.w - ws;
					const unsigned width = has ? (unsigned)((int)cw.y - (int)cw.x) : 0u;
					// all but the first and last steps of a workgroup's range lie wholly inside it: no per-read range test there
					const bool interior = wt + TILE <= ww1 && (wt > 0u || first_rel == 0u) && (wt + TILE) * 2u - first_rel <= n_rel;
					auto in_range = [&](const int j, unsigned &rel) {
						const unsigned wj = wt + lane * (unsigned)SW + (unsigned)(k0 + j / 2);
						rel = wj * 2u + (unsigned)(j & 1) - first_rel;      // position in the range (wraps above n_rel when outside)
						return wj < ww1 && rel < n_rel;
					};
					// read j of the lane: first base (in the loop's coordinates) and length
					auto read_of = [&](const int j, int &s, unsigned &len) {
						const int kk = k0 + j / 2;
						const unsigned a = (j & 1) ? cur[kk].z : cur[kk].x, b = (j & 1) ? cur[kk].w : cur[kk].y;
						s = (int)a; len = COMPACT ? b : b - a;
					};
					// Reads that end inside the owner's segment (A) and reads that end inside it or the segment that abuts it (L):
					// count and matched bases of each kind in one word (count << 24 | bases: four reads of < 2^18 bases); the run
					// into the next segment is L minus A.  lane_open: a read that the cell does not settle -- it lies in another
					// cell or in none, or runs past e2.
					const bool both = (cx.z & CELLX_BOTH) != 0;          // two owners: e1 / e2 the ends of their segments, slot 1 / slot 2 theirs
					unsigned accA = 0, accL = 0;
					bool lane_open = false;
					auto decide = [&](auto whole_step) {
#pragma unroll
						for (int j = 0; j < N_READS; ++j) {
							int s; unsigned len;
							read_of(j, s, len);
							bool in = true;
							if (!decltype(whole_step)::value) { unsigned rel; in = in_range(j, rel); }
							const int e = s + (int)len;
							const bool m = in && (unsigned)(s - lo) < width;
							const bool a = m && e <= e1;
							const bool l = m && e <= e2;
							lane_open = lane_open || (in && !(both ? a : l));
							const unsigned p = len | (min(len, 1u) << 24);           // (an empty record -- the padding of a cell's group -- counts for nothing)
							accA += a ? p : 0u;
							accL += l ? p : 0u;
						}
					};
					if (interior) decide(std::true_type{}); else decide(std::false_type{});
					const unsigned s1 = cx.x & 0xFFFFu, s2 = cx.x >> 16;
					if (!ABL(A, (1u | 16384u))) {
						const unsigned long long addA = ((unsigned long long)(accA >> 24) << 40) | (accA & 0xFFFFFFu);
						const unsigned accX = accL - accA;
						if (accA && s1 != CELL_NONE) atomicAdd(&C.hist[s1], addA);
						if (accA && both && s2 != CELL_NONE) atomicAdd(&C.hist[s2], addA);
						if (accX && s2 != CELL_NONE) atomicAdd(&C.hist[s2], ((unsigned long long)(accX >> 24) << 40) | (accX & 0xFFFFFFu));
					} else asm volatile("" ::"v"(accA), "v"(accL));
					if (ABL(A, 256u)) {
						const bool any_open = __any(lane_open);
						if (lane == 0) { atomicAdd(&A.dbg[12], 1ull); atomicAdd(&A.dbg[11], any_open ? 1ull : 0ull); }
					}
					if (__any(lane_open) && !ABL(A, 17u | 524288u)) {
						// Some read of some lane is not settled by its lane's cell (a lane's reads straddle a cell boundary once per
						// cell; the rest are reads past the owner's segments or in no cell).  They are parked for the general walk,
						// two reads of every lane at a time: a read in a one-owner cell with that owner as the one event to look at,
						// the others from the first event of their bin.
#pragma unroll
						for (int h = 0; h < N_READS; h += 2) {
							uint4 en[2];
							unsigned open = 0;
#pragma unroll
							for (int j = h; j < h + 2; ++j) {
								int s; unsigned len, rel;
								read_of(j, s, len);
								const bool in = in_range(j, rel);
								const bool m = (unsigned)(s - lo) < width;
								const int e = s + (int)len;
								const bool op = in && len != 0u && !(m && e <= (both ? e1 : e2));
								open |= op ? 1u << (j - h) : 0u;
								// in a one-owner cell: that owner; in a two-owner cell and inside the farther segment: the owner of the
								// nearer one (the loop has counted the read for the other); else from the first event of the bin
								en[j - h] = make_uint4((unsigned)(s + ws), (unsigned)(s + ws) + len, (m && (!both || e <= e2)) ? (cx.w | PARK_ONE_EVENT) : PARK_EVENT_UNKNOWN, rel);
								if (ABL(A, 256u) && op) { atomicAdd(&A.dbg[5 + (m ? (both ? 2 : 1) : 0)], 1ull); atomicAdd(&A.dbg[0], 1ull); }
							}
							R.push2(open, lane, en[0], en[1]);
							// the ring holds what one walk leaves behind (< 64) plus these 128 entries
							if (R.live() >= 64u) {             // wave-uniform
								if (!ABL(A, 32u)) walk_parked<NB, PACKED2>(C, R, false, base);
								else R.head = R.tail;
							}
						}
					}
				}
			} else if (kg == 0) {
				// Two-block reads, a lane's two neighbours at a time.  A read is settled here when its first block lies in a
				// cell with one owner (then only that event can match it, common/read.h:204-274):
				//   J  block 1 runs to the end of its segment, block 2 starts on the first base of a later segment of the same
				//      event and ends inside it: both segments match, matched == total, the class of the two-segment mask;
				//   S  block 2 cannot continue the match -- block 1 stops short of its segment's end, or block 2 starts on the
				//      first base of no later segment of that event (the event's packed record lists them): the walk of
				//      Read::build stops after block 1, so the read matches that one segment with matched = |block 1|, valid
				//      only if that is more than 98 % of the read (count/count.cpp:441);
				//   a read from a start cell that ends before gene_end counts for nobody (see CELL_K_START).
				// Everything else -- shared cells, a run over abutting segments, touching blocks -- is parked for the general walk.
				// The ingest groups the reads of a bin by junction, so a lane's second read mostly crosses the junction of its
				// first and only needs its outer ends compared; when it does not, it is parked.
				// junction: block 1 ends on its segment's end and block 2 starts a later segment of the same event (whether or not it
				// also ends inside that segment); jslot: the histogram slot of that two-segment class, CELL_NONE without one
				struct Look { bool add, park; unsigned slot, matched, hint; bool junction; unsigned jslot; int4 c; };       // c: the cell of block 1 (lo, hi), -, end of block 2's segment
				auto look2 = [&](const int4 rd, const bool in) {
					Look L;
					unsigned c1, evf;
					locate(rd.x, c1, evf);
					const unsigned c1c = min(c1, n_cells - 1u);
					const uint4 cw1 = cells[c1c];          // lo, hi, e1, e2
					const uint4 cx1 = cellx[c1c];          // slots, info, link, owner event
					const unsigned i1 = cx1.y;
					const unsigned k1 = (i1 >> 2) & 0x3Fu;
					const bool here = c1 < n_cells && (int)cw1.x <= rd.x && rd.x < (int)cw1.y && !ABL(A, 8u);
					const bool v1 = here && i1 < CELL_INFO_EMPTY;                                       // block 1 starts in a one-owner cell
					const bool start1 = k1 == CELL_K_START;
					const bool inside1 = v1 && !start1 && rd.y <= (int)cw1.z && rd.z != rd.y;            // block 1 ends inside its segment (touching blocks: the exception pass decides)
					const bool ends1 = inside1 && rd.y == (int)cw1.z;                                      // ... on its end
					// the owner's record: where its segments start and end (unused ones hold INT32_MAX, which no block reaches)
					const unsigned ri = 3u * (v1 ? cx1.w : 0u);
					const uint4 w0r = C.recs[ri], w1r = C.recs[ri + 1u], w2r = C.recs[ri + 2u];
					// block 2 continues the match only from the first base of a later segment of the same event
					const unsigned k2 = rd.z == (int)w1r.z ? 1u : (rd.z == (int)w2r.x ? 2u : (rd.z == (int)w2r.z ? 3u : 0u));
					const int end2 = k2 == 1u ? (int)w1r.w : (k2 == 2u ? (int)w2r.y : (int)w2r.w);
					const bool cont = ends1 && k2 > k1;
					const bool J = cont && rd.w <= end2;
					const bool S = inside1 && !cont;
					// counts for nobody: from a start cell and over before gene_end; or block 1 starts inside no segment at all
					const bool drop = (v1 && start1 && rd.w <= (int)cw1.w) || (here && i1 == CELL_INFO_EMPTY);
					const unsigned len1 = (unsigned)(rd.y - rd.x), total = len1 + (unsigned)(rd.w - rd.z);
					const unsigned long long tbl = ((unsigned long long)w0r.w << 32) | w0r.z;
					const unsigned cls = (unsigned)(tbl >> (4u * ((1u << (k1 & 3u)) | (1u << k2)))) & 0xFu;
					const unsigned sa = cx1.x & 0xFFFFu;
					L.junction = cont;
					L.jslot = cls != 0u ? (w0r.y & 0xFFFFu) + cls - 1u : CELL_NONE;
					L.add = in && ((J && cls != 0u) || (S && sa != CELL_NONE && 50u * len1 > 49u * total));
					L.slot = J ? (w0r.y & 0xFFFFu) + cls - 1u : sa;
					L.matched = J ? total : len1;
					L.park = in && !(J || S || drop);
					L.hint = v1 ? (cx1.w | PARK_ONE_EVENT) : evf;
					L.c = make_int4((int)cw1.x, (int)cw1.y, 0, end2);
					if (ABL(A, 256u) && L.park) atomicAdd(&A.dbg[8 + (v1 ? 1 : 0)], 1ull);       // parked: block 1 in no one-owner cell / in one
					return L;
				};
				const uint4 u = cur[k0];
				const int4 rd = make_int4((int)u.x, (int)u.y, (int)u.z, (int)u.w);
				const unsigned w0i = wt + lane * (unsigned)SW + (unsigned)k0;
				const unsigned rel = w0i - first_rel;
				const bool in = w0i < ww1 && rel < n_rel;
				const Look L1 = look2(rd, in);
				// (the first read adds to its own slot -- the junction's or, for a read that stops after block 1, its segment's; the
				// second, when it crosses the first one's junction, to the junction's)
				unsigned n_add = L1.add ? 1u : 0u, s_add = L1.add ? L1.matched : 0u, n_add2 = 0, s_add2 = 0;
				park[0] = L1.park && !ABL(A, 17u | 1048576u);
				// a parked read: its blocks and (event to look at, position) -- or, PACKED2, its compact record with those two in one word
				pe0[0] = PACKED2 ? make_uint4(raw2[0].x, raw2[0].y, L1.hint, rel) : u;
				pe1[0] = make_uint4(L1.hint, rel, 0u, 0u);
#pragma unroll
				for (int j = 1; j < N_READS; ++j) {
					const uint4 v = cur[k0 + j];
					const int4 r2 = make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
					const unsigned wj = wt + lane * (unsigned)SW + (unsigned)(k0 + j);
					const unsigned rel2 = wj - first_rel;
					const bool in2 = wj < ww1 && rel2 < n_rel;
					// same junction: block 1 ends and block 2 starts where the first read's do; then only the outer ends matter
					const bool same = in2 && L1.junction && r2.y == rd.y && r2.z == rd.z && L1.c.x <= r2.x && r2.x < L1.c.y && r2.w <= L1.c.w;
					n_add2 += (same && L1.jslot != CELL_NONE) ? 1u : 0u;
					s_add2 += (same && L1.jslot != CELL_NONE) ? (unsigned)((r2.y - r2.x) + (r2.w - r2.z)) : 0u;
					// Not the first read's junction (an empty record, the padding of a junction group, aside).  The pool is laid out
					// by junction group, so that is a pair of the bucket's last group -- the reads that cross no junction of the
					// annotation -- or a read that runs past its second segment: parked.  (A look of its own for such a second
					// read, in the steps that hold such pairs only, was measured twice: 0.246 against 0.253 ms before the groups,
					// 0.1627 against 0.1677 with them -- it costs more than the walk it saves.)
					park[j] = in2 && !same && r2.y != r2.x && !ABL(A, 17u | 1048576u | 2097152u);
					const uint4 rq = raw2[PACKED2 ? j / 2 : 0];
					pe0[j] = PACKED2 ? make_uint4((j & 1) ? rq.z : rq.x, (j & 1) ? rq.w : rq.y, PARK_EVENT_UNKNOWN, rel2) : v;
					pe1[j] = make_uint4(PARK_EVENT_UNKNOWN, rel2, 0u, 0u);
				}
				if (!ABL(A, 1u)) {
					if (n_add && n_add2 && L1.slot == L1.jslot) { n_add += n_add2; s_add += s_add2; n_add2 = 0; }
					if (n_add) atomicAdd(&C.hist[L1.slot], ((unsigned long long)n_add << 40) | s_add);
					if (n_add2) atomicAdd(&C.hist[L1.jslot], ((unsigned long long)n_add2 << 40) | s_add2);
				} else asm volatile("" ::"v"(n_add), "v"(s_add), "v"(n_add2), "v"(s_add2));
			}
		}
		if (RPW == 1 && PACKED2) {
			static_assert(!PACKED2 || N_READS == 4 || N_READS == 2, "two or four parked entries a step, two per push");
#pragma unroll 1
			for (int h = 0; h < N_READS / 2; ++h) {
				const bool pa = h ? park[N_READS - 2] : park[0], pb = h ? park[N_READS - 1] : park[1];
				const uint4 qa = h ? pe0[N_READS - 2] : pe0[0], qb = h ? pe0[N_READS - 1] : pe0[1];
				if (ABL(A, 256u)) atomicAdd(&A.dbg[NB - 1], (pa ? 1ull : 0ull) + (pb ? 1ull : 0ull));
				R.push2((pa ? 1u : 0u) | (pb ? 2u : 0u), lane, qa, qb);
				// the ring holds what one walk leaves behind (< 64) plus these 128 one-word entries
				if (R.live() >= 64u) {             // wave-uniform
					if (!ABL(A, 32u)) walk_parked<NB, PACKED2>(C, R, false, base);
					else R.head = R.tail;
				}
			}
		} else if (RPW == 1) {
			static_assert(RPW != 1 || PACKED2 || N_READS == 2, "two parked entries a step");
			// (written out twice: as a loop over q the compiler kept the entries in scratch memory and indexed them there)
#pragma unroll
			for (int q = 0; q < 2; ++q) {
				const bool pq = park[q ? N_READS - 1 : 0];
				const uint4 q0 = pe0[q ? N_READS - 1 : 0], q1 = pe1[q ? N_READS - 1 : 0];
				if (ABL(A, 256u) && pq) atomicAdd(&A.dbg[NB - 1], 1ull);
				R.push(pq, lane, q0, q1);
				// the ring holds what one walk leaves behind (< 64) plus 64 two-block entries
				if (R.live() >= 64u) {             // wave-uniform
					if (!ABL(A, 32u)) walk_parked<NB, PACKED2>(C, R, false, base);
					else R.head = R.tail;
				}
			}
		}
		}
	};
	if (ww0 < ww1) fetch(ww0);
	for (unsigned wt = ww0; wt < ww1; wt += WAVES * TILE) {
		uint4 cur[SW];
		if (COMPACT) {
			if (RPW == 2) {
				// one-block reads stay in the records' own terms: (offset from `base`, length); the loop compares there
#pragma unroll
				for (int cq = 0; cq < CW; ++cq) {
					const uint4 t = nxt[cq];
					cur[2 * cq] = make_uint4(t.x & lsq::COMPACT_OFF_MASK, t.x >> lsq::COMPACT_OFF_BITS, t.y & lsq::COMPACT_OFF_MASK, t.y >> lsq::COMPACT_OFF_BITS);
					cur[2 * cq + 1] = make_uint4(t.z & lsq::COMPACT_OFF_MASK, t.z >> lsq::COMPACT_OFF_BITS, t.w & lsq::COMPACT_OFF_MASK, t.w >> lsq::COMPACT_OFF_BITS);
				}
			} else {
#pragma unroll
				for (int cq = 0; cq < CW; ++cq) {
					const uint4 tq = nxt[cq];
					const int4 a = unpack_two_block(tq.x, tq.y, base), b = unpack_two_block(tq.z, tq.w, base);
					cur[2 * cq] = make_uint4((unsigned)a.x, (unsigned)a.y, (unsigned)a.z, (unsigned)a.w);
					cur[2 * cq + 1] = make_uint4((unsigned)b.x, (unsigned)b.y, (unsigned)b.z, (unsigned)b.w);
					raw2[cq] = tq;
				}
			}
		} else {
#pragma unroll
			for (int k = 0; k < SW; ++k) cur[k] = nxt[k];
		}
		if (wt + WAVES * TILE < ww1) fetch(wt + WAVES * TILE);
		do_step(cur, wt);
	}
	// (the waves drain their own rings: handing the leftovers of four waves to one, so that fewer partly filled walk steps
	// run, was measured -- 116 000 -> 87 000 walk steps on C3 -- and lost more at the two barriers it needs: 0.174 -> 0.180 ms)
	if (R.live() && !ABL(A, 32u)) walk_parked<NB, PACKED2>(C, R, true, base);
}

// ---- compact pools: the streaming loops proper ---------------------------------------------------------------------
// The two loops below are what lsq_count_fast_kernel<true, *> runs (the generic stream_pool_fast above keeps the wide
// records).  They lean on the pools' layout, which the ingest establishes with the kernel's own look at the tables and
// lsq_debug_check_pool_layout verifies in the tests:
//   one-block pool: the records of an aligned group of eight (P1_GROUP_PAD) start in ONE cell -- the cell of the group's
//     first record -- or all in none; padding records are empty (length 0) and sit at the tail of a cell's records;
//   two-block pool: an aligned quadruple (P2_GROUP_PAD) lies in ONE junction group -- block 1 starts in the same one-owner cell and ends
//     on the end of that owner's segment, block 2 starts on the same first base of a later segment of that event -- or in
//     the bucket's last group (everything else); padding (length 0) never comes first in such a group.
// So a lane looks at the tables once, for its first record, and every further record only has to show where it ENDS.

// One-block reads, NR = 4 or 8 per lane and step (one or two 16-byte loads).  Records stay in their own terms (offset
// from `base`, length); the cell's bounds are brought into those terms once per lane.  Per record: its end against the
// two thresholds of the cell (e1: inside the owner's segment; e2: inside the segment that abuts it / the farther owner's),
// count and bases summed per lane in one word each (count << 24 | bases), one LDS atomic per class at the end.  Whether
// anything is left for the general walk falls out of the sums (records counted against records settled): no per-record
// flag, no scalar mask logic in the loop.
template <int NR>
__device__ inline void stream_pool1_compact(FastCtx &C, const uint4 *bins, const uint4 *cells, const uint4 *cellx, const unsigned n_cells, const BucketDesc &d,
                                            const CountArgs &A, uint4 *queue, const void *pool, const unsigned long long g0, const unsigned long long g1) {
	static_assert(NR == 4 || NR == 8, "one or two 16-byte words of four compact records");
	static_assert((unsigned)NR <= P1_GROUP_PAD, "a lane's records of a step lie in one cell group");
	constexpr int CW = NR / 4;                     // 16-byte loads a lane issues per step
	constexpr unsigned TILE = 64u * NR;            // records per wave step
	C.pool = 0u;
	C.slot0 = g0;
	global_words src = (global_words)pool;
	const unsigned lane = threadIdx.x & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// records [R0, R0 + n) of the pool, R0 = g0 rounded down to a lane's NR (ranges start on whole groups anyway); the
	// range proper is [first_rel, n) of them
	const unsigned long long R0 = g0 & ~(unsigned long long)(NR - 1);
	const unsigned n = (unsigned)(g1 - R0), first_rel = (unsigned)(g0 - R0), n_rel = (unsigned)(g1 - g0);      // a workgroup's range stays below 2^21 reads
	const unsigned last_word = (n - 1u) >> 2;
	const int base = d.lo - lsq::COMPACT_BIAS;
	uint4 nxt[CW];
	auto fetch = [&](const unsigned t) {
#pragma unroll
		for (int cq = 0; cq < CW; ++cq) {
			// past the end of the range the last word is read again (no predication: the range test discards it)
			const unsigned w = min((t >> 2) + lane * (unsigned)CW + (unsigned)cq, last_word);
			const u32x4 v = src[(R0 >> 2) + w];
			nxt[cq] = make_uint4(v.x, v.y, v.z, v.w);
		}
	};
	Ring<1> R;
	R.q = queue;
	const unsigned t_begin = min(wave * TILE, n);
	if (t_begin < n) fetch(t_begin);
	for (unsigned t = t_begin; t < n; t += WAVES * TILE) {
		unsigned rec[NR];
#pragma unroll
		for (int cq = 0; cq < CW; ++cq) { rec[4 * cq] = nxt[cq].x; rec[4 * cq + 1] = nxt[cq].y; rec[4 * cq + 2] = nxt[cq].z; rec[4 * cq + 3] = nxt[cq].w; }
		if (t + WAVES * TILE < n) fetch(t + WAVES * TILE);
		if (ABL(A, 512u)) {      // developer switch: stream only
#pragma unroll
			for (int j = 0; j < NR; ++j) asm volatile("" ::"v"(rec[j]));
			continue;
		}
		// ---- one look at the tables: the cell of the lane's first record
		const int p0 = (int)(rec[0] & lsq::COMPACT_OFF_MASK) + base;
		const int rel0 = p0 - d.lo;
		const unsigned bin = rel0 <= 0 ? 0u : ((unsigned)rel0 >> d.shift);
		const uint4 br = bins[min(bin, d.n_bins - 1u)];     // first cell | first event << 16, ends of that cell and the next two
		unsigned ci = (br.x & 0xFFFFu) + (unsigned)(p0 >= (int)br.y) + (unsigned)(p0 >= (int)br.z) + (unsigned)(p0 >= (int)br.w);
		if (__any(p0 >= (int)br.w) && !ABL(A, 65536u)) {
			while (ci + 1u < n_cells && p0 >= (int)cells[ci + 1u].x) ++ci;
		}
		const unsigned cc = min(ci, n_cells - 1u);
		const uint4 cw = cells[cc];           // lo, hi, e1, e2
		const uint4 cx = cellx[cc];           // slots, info, flags, owner event
		// the lane's records start in this cell (layout) -- if it is one: the bucket's last group holds the reads of no cell
		const bool has = ci < n_cells && (int)cw.x <= p0 && p0 < (int)cw.y && !ABL(A, 8u);
		const bool both = has && (cx.z & CELLX_BOTH) != 0;   // two owners: e1 / e2 the ends of their segments, slot 1 / slot 2 theirs
		// thresholds in the records' terms; without a cell nothing is settled here
		const int e1 = has ? (int)cw.z - base : -1, e2 = has ? (int)cw.w - base : -1;
		// every step but the first and last of a workgroup's range lies wholly inside it: no per-record range test there
		const bool interior = t + TILE <= n && t >= first_rel;
		unsigned accN = 0, accA = 0, accL = 0;          // all records / those that end by e1 / by e2: count << 24 | bases
		// (one_threshold: no lane's cell has a second one -- nothing abuts the owner's segment, no second owner: most waves)
		auto decide = [&](auto whole_step, auto one_threshold) {
#pragma unroll
			for (int j = 0; j < NR; ++j) {
				const unsigned len = rec[j] >> lsq::COMPACT_OFF_BITS;
				const int e = (int)(rec[j] & lsq::COMPACT_OFF_MASK) + (int)len;
				unsigned p = len | (min(len, 1u) << 24);           // (an empty record -- the padding of a cell's group -- counts for nothing)
				if (!decltype(whole_step)::value) {
					const unsigned idx = t + lane * (unsigned)NR + (unsigned)j;
					p = (idx < n && idx - first_rel < n_rel) ? p : 0u;
				}
				accN += p;
				accA += e <= e1 ? p : 0u;
				if (!decltype(one_threshold)::value) accL += e <= e2 ? p : 0u;
			}
			if (decltype(one_threshold)::value) accL = accA;
		};
		if (__all(e1 == e2) && !ABL(A, 8388608u)) { if (interior) decide(std::true_type{}, std::true_type{}); else decide(std::false_type{}, std::true_type{}); }
		else { if (interior) decide(std::true_type{}, std::false_type{}); else decide(std::false_type{}, std::false_type{}); }
		const unsigned s1 = cx.x & 0xFFFFu, s2 = cx.x >> 16;
		if (!ABL(A, (1u | 16384u))) {
			// slot 1: the records inside the (nearer) owner's segment; slot 2: one owner -- those that run on into the abutting
			// segment (L - A); two owners -- everything inside the farther owner's segment (L: the sums add up in one word,
			// sixteen records of < 1 024 bases at most)
			const unsigned acc2 = both ? accL : accL - accA;
			if (accA && s1 != CELL_NONE) atomicAdd(&C.hist[s1], ((unsigned long long)(accA >> 24) << 40) | (accA & 0xFFFFFFu));
			if (acc2 && s2 != CELL_NONE) atomicAdd(&C.hist[s2], ((unsigned long long)(acc2 >> 24) << 40) | (acc2 & 0xFFFFFFu));
		} else asm volatile("" ::"v"(accA), "v"(accL));
		// settled here: one owner -- every record that ends by e2; two owners -- by e1 (one that ends in (e1, e2] has been
		// counted for the farther owner and still goes to the walk for the nearer one)
		const bool lane_open = (accN >> 24) != ((both ? accA : accL) >> 24);
		if (ABL(A, 256u)) {
			const bool any_open = __any(lane_open);
			if (lane == 0) { atomicAdd(&A.dbg[12], 1ull); atomicAdd(&A.dbg[11], any_open ? 1ull : 0ull); }
		}
		if (__any(lane_open) && !ABL(A, 17u | 524288u)) {
			// Parked for the general walk, two records of every lane at a time: a read in a one-owner cell with that owner as the
			// one event to look at, one in a two-owner cell and inside the farther segment with the nearer owner, the others from
			// the first event of their bin.
#pragma unroll
			for (int h = 0; h < NR; h += 2) {
				uint4 en[2];
				unsigned open = 0;
#pragma unroll
				for (int j = h; j < h + 2; ++j) {
					const unsigned len = rec[j] >> lsq::COMPACT_OFF_BITS;
					const int so = (int)(rec[j] & lsq::COMPACT_OFF_MASK), e = so + (int)len;
					const unsigned idx = t + lane * (unsigned)NR + (unsigned)j, rel = idx - first_rel;
					const bool in = idx < n && rel < n_rel;
					// two owners: past the end of an owner's segment with no segment abutting it the read matches that owner up to the
					// end only -- valid for it only if 50 x the overhang < the read (count/count.cpp:441); otherwise settled: nothing to add
					const bool gone_near = both && (cx.z & CELLX_NEAR_NO_ABUT) != 0u && e > e1 && 50 * (e - e1) >= (int)len;
					const bool gone_far = both && (cx.z & CELLX_FAR_NO_ABUT) != 0u && e > e2 && 50 * (e - e2) >= (int)len;
					const bool done_near = e <= e1 || gone_near, done_far = e <= e2 || gone_far;          // (one owner: e1 plays no part below)
					const bool op = in && len != 0u && !(both ? (done_near && done_far) : e <= e2);
					open |= op ? 1u << (j - h) : 0u;
					// the one event to look at: one owner -- that owner; two owners -- the one that is not done with the read; both open: all
					const unsigned hint = !has ? PARK_EVENT_UNKNOWN : (!both ? (cx.w | PARK_ONE_EVENT) :
					                      (done_far ? (cx.w | PARK_ONE_EVENT) : (done_near ? ((cx.z & 0xFFFFu) | PARK_ONE_EVENT) : PARK_EVENT_UNKNOWN)));
					en[j - h] = make_uint4((unsigned)(so + base), (unsigned)(so + base) + len, hint, rel);
					if (ABL(A, 256u) && op) { atomicAdd(&A.dbg[5 + (has ? (both ? 2 : 1) : 0)], 1ull); atomicAdd(&A.dbg[0], 1ull); }
				}
				R.push2(open, lane, en[0], en[1]);
				// the ring holds what one walk leaves behind (< 64) plus these 128 entries
				if (R.live() >= 64u) {             // wave-uniform
					if (!ABL(A, 32u)) walk_parked<1, false>(C, R, false, base);
					else R.head = R.tail;
				}
			}
		}
	}
	if (R.live() && !ABL(A, 32u)) walk_parked<1, false>(C, R, true, base);
}

// Two-block reads, NR = 2 or 4 per lane and step (8-byte records: block 1 as offset | length, block 2 as gap | length).
// The lane's first record gets the full look -- cell of block 1, the owner's packed record, the segment block 2 starts --
// and is settled as before: J (block 1 to the end of its segment, block 2 from the first base of a later segment and
// ending inside it: the two-segment class, matched == total), S (block 2 cannot continue the match: block 1's segment
// alone, if that is more than 98 % of the read) or parked.  When it crosses a junction, the other records of the quadruple
// cross the same one (layout): each only shows that block 2 ends inside that segment -- gap + length against one
// threshold -- and adds its bases; one that runs on, and every follower of a first record that crosses no junction, is
// parked for the general walk.
template <int NR>
__device__ inline void stream_pool2_compact(FastCtx &C, const uint4 *bins, const uint4 *cells, const uint4 *cellx, const unsigned n_cells, const BucketDesc &d,
                                            const CountArgs &A, uint4 *queue, const void *pool, const unsigned long long g0, const unsigned long long g1) {
	static_assert((NR == 2 || NR == 4 || NR == 8) && (unsigned)NR <= P2_GROUP_PAD, "whole 16-byte words of two compact records, inside one junction group");
	constexpr int CW = NR / 2;
	constexpr unsigned TILE = 64u * NR;
	C.pool = 1u;
	C.slot0 = g0;
	global_words src = (global_words)pool;
	const unsigned lane = threadIdx.x & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const unsigned long long R0 = g0 & ~(unsigned long long)(NR - 1);
	const unsigned n = (unsigned)(g1 - R0), first_rel = (unsigned)(g0 - R0), n_rel = (unsigned)(g1 - g0);
	const unsigned last_word = (n - 1u) >> 1;
	const int base = d.lo - lsq::COMPACT_BIAS;
	uint4 nxt[CW];
	auto fetch = [&](const unsigned t) {
#pragma unroll
		for (int cq = 0; cq < CW; ++cq) {
			const unsigned w = min((t >> 1) + lane * (unsigned)CW + (unsigned)cq, last_word);
			const u32x4 v = src[(R0 >> 1) + w];
			nxt[cq] = make_uint4(v.x, v.y, v.z, v.w);
		}
	};
	Ring<1> R;
	R.q = queue;
	const unsigned t_begin = min(wave * TILE, n);
	if (t_begin < n) fetch(t_begin);
	for (unsigned t = t_begin; t < n; t += WAVES * TILE) {
		unsigned ra[NR], rb[NR];           // block 1: offset | length << 22; block 2: gap | length << 22
#pragma unroll
		for (int cq = 0; cq < CW; ++cq) { ra[2 * cq] = nxt[cq].x; rb[2 * cq] = nxt[cq].y; ra[2 * cq + 1] = nxt[cq].z; rb[2 * cq + 1] = nxt[cq].w; }
		if (t + WAVES * TILE < n) fetch(t + WAVES * TILE);
		if (ABL(A, 512u)) {
#pragma unroll
			for (int j = 0; j < NR; ++j) asm volatile("" ::"v"(ra[j]), "v"(rb[j]));
			continue;
		}
		const bool interior = t + TILE <= n && t >= first_rel;
		const unsigned idx0 = t + lane * (unsigned)NR, rel0 = idx0 - first_rel;
		const bool in0 = interior || (idx0 < n && rel0 < n_rel);
		// ---- the first record: the full look
		const int4 rd = unpack_two_block(ra[0], rb[0], base);
		const int rel_p = rd.x - d.lo;
		const unsigned bin = rel_p <= 0 ? 0u : ((unsigned)rel_p >> d.shift);
		const uint4 br = bins[min(bin, d.n_bins - 1u)];
		unsigned c1 = (br.x & 0xFFFFu) + (unsigned)(rd.x >= (int)br.y) + (unsigned)(rd.x >= (int)br.z) + (unsigned)(rd.x >= (int)br.w);
		const unsigned evf = br.x >> 16;
		if (__any(rd.x >= (int)br.w) && !ABL(A, 65536u)) {
			while (c1 + 1u < n_cells && rd.x >= (int)cells[c1 + 1u].x) ++c1;
		}
		const unsigned c1c = min(c1, n_cells - 1u);
		const uint4 cw1 = cells[c1c];          // lo, hi, e1, e2
		const uint4 cx1 = cellx[c1c];          // slots, info, link, owner event
		const unsigned i1 = cx1.y;
		const unsigned k1 = (i1 >> 2) & 0x3Fu;
		const bool here = c1 < n_cells && (int)cw1.x <= rd.x && rd.x < (int)cw1.y && !ABL(A, 8u);
		const bool v1 = here && i1 < CELL_INFO_EMPTY;                                       // block 1 starts in a one-owner cell
		const bool start1 = k1 == CELL_K_START;
		const bool inside1 = v1 && !start1 && rd.y <= (int)cw1.z && rd.z != rd.y;            // block 1 ends inside its segment (touching blocks: the exception pass decides)
		const bool ends1 = inside1 && rd.y == (int)cw1.z;                                      // ... on its end
		// ... or runs on through the segment that abuts the owner's (the cell's e2 > e1) and ends on that one's end
		const bool ends1b = v1 && !start1 && (int)cw1.w > (int)cw1.z && rd.y == (int)cw1.w && rd.z != rd.y;
		// the owner's record: where its segments start and end (unused ones hold INT32_MAX, which no block reaches)
		const unsigned ri = 3u * (v1 ? cx1.w : 0u);
		const uint4 w0r = C.recs[ri], w1r = C.recs[ri + 1u], w2r = C.recs[ri + 2u];
		// block 2 continues the match only from the first base of a later segment of the same event ...
		const unsigned k2 = rd.z == (int)w1r.z ? 1u : (rd.z == (int)w2r.x ? 2u : (rd.z == (int)w2r.z ? 3u : 0u));
		const int end2 = k2 == 1u ? (int)w1r.w : (k2 == 2u ? (int)w2r.y : (int)w2r.w);
		// ... and may itself run on into the segment that abuts that one (abut bit k2: segment k2 + 1 starts where k2 ends)
		const unsigned abut = (w0r.y >> FAST_ABUT_SHIFT) & 7u;
		const bool abut2 = k2 != 0u && k2 < 3u && ((abut >> k2) & 1u) != 0u;
		const int end2b = !abut2 ? end2 : (k2 == 1u ? (int)w2r.y : (int)w2r.w);
		const unsigned k1e = ends1b ? k1 + 1u : k1;          // the last segment of block 1's run
		const bool junction = (ends1 || ends1b) && k2 > k1e;
		const bool J = junction && rd.w <= end2, Jb = junction && rd.w > end2 && rd.w <= end2b;
		const unsigned len1 = ra[0] >> lsq::COMPACT_OFF_BITS, total = len1 + (rb[0] >> lsq::COMPACT_OFF_BITS);
		const unsigned long long tbl = ((unsigned long long)w0r.w << 32) | w0r.z;
		const unsigned mask1 = (1u << (k1 & 3u)) | (ends1b ? 2u << (k1 & 3u) : 0u), maskA = mask1 | (1u << k2), maskB = maskA | (2u << k2);
		const unsigned clsA = (unsigned)(tbl >> (4u * (maskA & 15u))) & 0xFu, clsB = abut2 ? (unsigned)(tbl >> (4u * (maskB & 15u))) & 0xFu : 0u;
		const unsigned sa = cx1.x & 0xFFFFu;
		// the histogram slots of the junction's classes: segments of block 1 + segment k2, and + segment k2 + 1 (CELL_NONE: no compatible isoform)
		const unsigned slotA = clsA != 0u ? (w0r.y & 0xFFFFu) + clsA - 1u : CELL_NONE, slotB = clsB != 0u ? (w0r.y & 0xFFFFu) + clsB - 1u : CELL_NONE;
		// No junction.  The quadruple then shares the cell of this record (layout: the reads that cross no junction of the
		// annotation are grouped by the cell block 1 starts in), and block 2 of none of its reads continues a match: a key of
		// the junction groups would name it.  What is left per read is whether block 1 alone could carry it: matched <= |block 1|
		// whenever block 1 ends inside a segment of every owner or past one that nothing abuts, so a read with 50 |block 1| <= 49
		// total is valid for nobody (count/count.cpp:441) and is settled here as nothing.  e1c / e2c: the owners' segment ends
		// in the records' terms; `gone_ok`: where block 1 may end.
		// (a wave whose every lane crosses a junction -- the usual one: the junction groups come first in a bucket's pool and hold
		// most of it -- skips all of this)
		const bool any_plain = __any(!junction);
		bool two = false, empty = false, near_free = false, far_free = false, S = false, drop = false, addS = false;
		int e1c = 0, e2c = 0;
		if (any_plain) {
			S = inside1 && !junction;          // block 1 inside its segment and no junction: that segment's class, if it is more than 98 % of the read
			// counts for nobody: from a start cell and over before gene_end; or block 1 starts inside no segment at all
			drop = (v1 && start1 && rd.w <= (int)cw1.w) || (here && i1 == CELL_INFO_EMPTY);
			addS = in0 && S && sa != CELL_NONE && 50u * len1 > 49u * total;
			two = here && (cx1.z & CELLX_BOTH) != 0u; empty = here && i1 == CELL_INFO_EMPTY;
			e1c = (int)cw1.z - base; e2c = (int)cw1.w - base;
			near_free = (cx1.z & CELLX_NEAR_NO_ABUT) != 0u; far_free = (cx1.z & CELLX_FAR_NO_ABUT) != 0u;
		}
		auto nothing = [&](const unsigned a, const unsigned b2) __attribute__((always_inline)) {      // a record (block 1, block 2) of this cell that counts for nobody
			const unsigned l1 = a >> lsq::COMPACT_OFF_BITS, l2 = b2 >> lsq::COMPACT_OFF_BITS, gap = b2 & lsq::COMPACT_OFF_MASK;
			const int y = (int)(a & lsq::COMPACT_OFF_MASK) + (int)l1;
			const bool weak = gap != 0u && 50u * l1 <= 49u * (l1 + l2);                 // (touching blocks: the exception pass decides)
			const bool one_ok = y <= e2c;                                                  // one owner: inside its segment or the one abutting it
			const bool two_ok = y != e1c && y != e2c && (y < e1c || near_free) && (y < e2c || far_free);
			return empty || (weak && (two ? two_ok : (v1 && !start1 && one_ok)));
		};
		bool none[NR];
#pragma unroll
		for (int j = 0; j < NR; ++j) none[j] = false;
		if (any_plain) {
#pragma unroll
			for (int j = 0; j < NR; ++j) none[j] = !junction && nothing(ra[j], rb[j]);
		}
		bool park[NR];
		park[0] = in0 && len1 != 0u && !(J || Jb || S || drop || none[0]) && !ABL(A, 17u | 1048576u);
		if (ABL(A, 256u) && park[0]) atomicAdd(&A.dbg[8 + (v1 ? 1 : 0)], 1ull);       // parked: block 1 in no one-owner cell / in one
		unsigned nA = (in0 && J && slotA != CELL_NONE) ? 1u : 0u, sA = nA ? total : 0u, nB = (in0 && Jb && slotB != CELL_NONE) ? 1u : 0u, sB = nB ? total : 0u;
		// ---- the other records: block 2 must end inside the junction's second segment (gap + length <= lim) or inside the one
		// that abuts it (<= limb)
		const int lim = junction ? end2 - rd.y : -1, limb = junction ? end2b - rd.y : -1;
		bool any_park = park[0];
		const bool any_b = __any(limb > lim) || ABL(A, 8388608u);
#pragma unroll
		for (int j = 1; j < NR; ++j) {
			const unsigned l1 = ra[j] >> lsq::COMPACT_OFF_BITS, l2 = rb[j] >> lsq::COMPACT_OFF_BITS;
			const int reach = (int)(rb[j] & lsq::COMPACT_OFF_MASK) + (int)l2;         // from the end of block 1 to the end of block 2
			bool in = l1 != 0u;                                                          // (padding of a junction group: an empty record)
			if (!interior) { const unsigned idx = idx0 + (unsigned)j; in = in && idx < n && idx - first_rel < n_rel; }
			const bool sameA = in && reach <= lim;
			bool sameB = false;
			const bool cA = sameA && slotA != CELL_NONE;
			nA += cA ? 1u : 0u; sA += cA ? l1 + l2 : 0u;
			if (any_b) {          // (wave-uniform: some lane's junction leads into a segment that another abuts)
				sameB = in && reach > lim && reach <= limb;
				const bool cB = sameB && slotB != CELL_NONE;
				nB += cB ? 1u : 0u; sB += cB ? l1 + l2 : 0u;
			}
			park[j] = in && !(sameA || sameB || none[j]) && !ABL(A, 17u | 1048576u | 2097152u);
			if (ABL(A, 256u) && park[j]) atomicAdd(&A.dbg[junction ? 14 : 13], 1ull);       // parked followers: block 2 runs past the junction's segments / the first record crosses no junction
			any_park = any_park || park[j];
		}
		if (!ABL(A, 1u)) {
			if (addS) atomicAdd(&C.hist[sa], (1ull << 40) | len1);
			if (nA) atomicAdd(&C.hist[slotA], ((unsigned long long)nA << 40) | sA);
			if (nB) atomicAdd(&C.hist[slotB], ((unsigned long long)nB << 40) | sB);
		} else asm volatile("" ::"v"(nA), "v"(sA), "v"(nB), "v"(sB));
		if (__any(any_park)) {
			// parked as their compact records: (block 1, block 2, event to look at, position in the range)
			const unsigned hint0 = v1 ? (cx1.w | PARK_ONE_EVENT) : evf;
#pragma unroll
			for (int h = 0; h < NR; h += 2) {
				const uint4 qa = make_uint4(ra[h], rb[h], h == 0 ? hint0 : PARK_EVENT_UNKNOWN, rel0 + (unsigned)h);
				const uint4 qb = make_uint4(ra[h + 1], rb[h + 1], PARK_EVENT_UNKNOWN, rel0 + (unsigned)h + 1u);
				if (ABL(A, 256u)) atomicAdd(&A.dbg[1], (park[h] ? 1ull : 0ull) + (park[h + 1] ? 1ull : 0ull));
				R.push2((park[h] ? 1u : 0u) | (park[h + 1] ? 2u : 0u), lane, qa, qb);
				// the ring holds what one walk leaves behind (< 64) plus these 128 one-word entries
				if (R.live() >= 64u) {             // wave-uniform
					if (!ABL(A, 32u)) walk_parked<2, true>(C, R, false, base);
					else R.head = R.tail;
				}
			}
		}
	}
	if (R.live() && !ABL(A, 32u)) walk_parked<2, true>(C, R, true, base);
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

// waves per SIMD the fast kernel is compiled for: 6 = as many as the LDS of six workgroups allows; the compiler
// then keeps to 80 VGPRs (four dwords spilled) where it would take 86 and leave room for five (measured 0.239 -> 0.228 ms)
#ifndef LSQ_FAST_WAVES
#define LSQ_FAST_WAVES 6
#endif
// <COMPACT, 2>: four one-block reads per lane and look, six waves a SIMD (80 registers); <true, 4>: eight, five waves (96
// registers) -- for the launches that are held to five workgroups a compute unit anyway (run_count), where the longer
// step of a lane costs no occupancy and the look is shared by twice the reads
template <bool COMPACT, int P1W>
__global__ void __launch_bounds__(COUNT_BLOCK, P1W == 4 ? 5 : LSQ_FAST_WAVES) lsq_count_fast_kernel(CountArgs A) {
	// LDS: the bucket's tables (image, histograms), then the waves' rings
	extern __shared__ __align__(16) unsigned char lds[];
	const unsigned tid = threadIdx.x;
	// (developer build: when each workgroup started and ended -- the picture of the launch's tail)
	struct Trace {
		unsigned long long *p;
		__device__ Trace(unsigned long long *q) : p(q) { if (p && threadIdx.x == 0) p[4u * blockIdx.x] = wall_clock64(); }
		__device__ ~Trace() { if (p && threadIdx.x == 0) p[4u * blockIdx.x + 1u] = wall_clock64(); }
	} trace(ABL(A, 4194304u) ? A.wg_trace : nullptr);
	if (blockIdx.x < A.n_workers) { pool_n_worker(A, A.n_pn, A.n_workers); return; }
	const unsigned wg = blockIdx.x - A.n_workers, n_wg = gridDim.x - A.n_workers;     // the streaming workgroups
	uint4 *wave_queue = reinterpret_cast<uint4 *>(lds + A.tables_lds_bytes) + (tid >> 6) * WAVE_QUEUE_WORDS;
	(void)n_wg;
	// the share and its buckets, one visit record each (scalar loads: everything here is wave-uniform); the plan holds the
	// record of the first, every record names the next packed bucket with slots
	// (through the constant address space: the records are written before the launch and never during it, and a uniform
	// address there is a scalar load -- as plain global memory the compiler, seeing the kernel's own atomics, took vector
	// loads and two dozen readfirstlanes)
	typedef const lsq::VisitRec __attribute__((address_space(4))) *const_visit;
	typedef const lsq::WgPlan __attribute__((address_space(4))) *const_plan;
	const_plan plan = (const_plan)(A.wg_plan + wg);
	const unsigned long long s_begin = plan->s_begin, s_end = plan->s_end;
	if (s_begin >= s_end) return;
	if (ABL(A, 4096u)) return;       // developer switch: dispatch cost only
	const_visit vr = &plan->first;
	unsigned b = vr->b;
	bool first_visit = true;
	while (b < A.n_buckets) {
		const unsigned long long bs = vr->bs;
		if (bs >= s_end) break;
		BucketDesc d;         // (member by member: a struct copy out of that address space has no constructor to bind to)
		d.img_off = vr->d.img_off; d.img_bytes = vr->d.img_bytes; d.n_events = vr->d.n_events; d.n_bins = vr->d.n_bins; d.lo = vr->d.lo; d.shift = vr->d.shift;
		d.ev_off = vr->d.ev_off; d.seg_off = vr->d.seg_off; d.iso_off = vr->d.iso_off; d.hist_off = vr->d.hist_off; d.n_cls = vr->d.n_cls; d.cls_base = vr->d.cls_base;
		d.ev_base = vr->d.ev_base; d.chrom_id = vr->d.chrom_id; d.kind = vr->d.kind; d.hi = vr->d.hi;
		const unsigned long long be = vr->be, p1o = vr->p1o, n1 = vr->p1n, p2o = vr->p2o, n2 = vr->p2n;
		const unsigned next = vr->next;
		unsigned char *buf = lds;
		// ---- stage: the image into LDS, the histograms cleared (the flush of the bucket before is behind a barrier)
		if (!first_visit) __syncthreads();
		first_visit = false;
		{
			global_words src = (global_words)(A.images + d.img_off);
			uint4 *dst = reinterpret_cast<uint4 *>(buf);
			for (unsigned i = tid; i < d.img_bytes / 16; i += COUNT_BLOCK) { const u32x4 t = src[i]; dst[i] = make_uint4(t.x, t.y, t.z, t.w); }
			unsigned long long *h = reinterpret_cast<unsigned long long *>(buf + d.hist_off);
			for (unsigned i = tid; i < HIST_REPLICAS * (d.n_cls | 1u); i += COUNT_BLOCK) h[i] = 0;
		}
		__syncthreads();
		if (ABL(A, 8192u)) return;       // developer switch: dispatch + first staging
		const uint4 *bins = reinterpret_cast<const uint4 *>(buf);
		const uint4 *cells = reinterpret_cast<const uint4 *>(buf + d.seg_off);
		// (iso_off of a packed bucket: cells proper | all owner records << 16; the CellX records follow the Cell ones)
		const unsigned n_cells = d.iso_off & 0xFFFFu;
		const uint4 *cellx = reinterpret_cast<const uint4 *>(buf + d.seg_off + 16u * (d.iso_off >> 16));
		FastCtx C;
		C.bins = bins; C.lo = d.lo; C.shift = d.shift; C.n_bins = d.n_bins;
		C.recs = reinterpret_cast<const uint4 *>(buf + d.ev_off);
		C.hist = reinterpret_cast<unsigned long long *>(buf + d.hist_off) + (tid & (HIST_REPLICAS - 1u)) * (d.n_cls | 1u);   // this lane's copy
		C.n_events = d.n_events; C.bucket = b;
		C.slot0 = 0; C.pool = 0;
		C.exc = A.exc; C.exc_count = A.exc_count; C.exc_cap = A.exc_cap; C.ablate = A.ablate; C.dbg = A.dbg;
		C.trace = (ABL(A, 4194304u) && A.wg_trace) ? A.wg_trace + 4u * blockIdx.x : nullptr;
		const unsigned long long l0 = (s_begin > bs ? s_begin : bs) - bs;
		const unsigned long long l1 = (s_end < be ? s_end : be) - bs;
		// ---- pool 1
		if (l0 < n1 && !ABL(A, 1024u)) {
			if constexpr (COMPACT) stream_pool1_compact<2 * P1W>(C, bins, cells, cellx, n_cells, d, A, wave_queue, A.p1, p1o + l0, p1o + (l1 < n1 ? l1 : n1));
			else stream_pool_fast<2, COMPACT, P1W>(C, bins, cells, cellx, n_cells, d, A, wave_queue, reinterpret_cast<const uint4 *>(A.p1), p1o + l0, p1o + (l1 < n1 ? l1 : n1));
		}
		// ---- pool 2
		if (l1 > n1 && l0 < n1 + n2 && !ABL(A, 2048u)) {
			const unsigned long long q0 = p2o + ((l0 > n1 ? l0 : n1) - n1), q1 = p2o + ((l1 < n1 + n2 ? l1 : n1 + n2) - n1);
			if constexpr (COMPACT) stream_pool2_compact<(P1W == 4 ? LSQ_P2_COMPACT_WORDS_W5 : LSQ_P2_COMPACT_WORDS)>(C, bins, cells, cellx, n_cells, d, A, wave_queue, A.p2, q0, q1);
			else stream_pool_fast<1, COMPACT, P1W>(C, bins, cells, cellx, n_cells, d, A, wave_queue, reinterpret_cast<const uint4 *>(A.p2), q0, q1);
		}
		// (reads with three or more blocks are the workers')
		__syncthreads();
		// ---- flush
		for (unsigned i = tid; i < d.n_cls; i += COUNT_BLOCK) {
			const unsigned long long *h0 = reinterpret_cast<const unsigned long long *>(buf + d.hist_off) + i;
			unsigned long long v = 0;
#pragma unroll
			for (unsigned r = 0; r < HIST_REPLICAS; ++r) v += h0[r * (d.n_cls | 1u)];       // counts stay below 2^24, bases below 2^40
			if (v && !ABL(A, 4u)) {
				atomicAdd(&A.cnt[d.cls_base + i], v >> 40);
				atomicAdd(&A.bases[d.cls_base + i], v & BASES_MASK);
			}
		}
		if (be >= s_end) break;          // the share ends in this bucket
		b = next;
		vr = (const_visit)(A.visits + b);
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

__device__ void recount_all_reads(const CountArgs &A, unsigned long long n_pn);

__global__ void __launch_bounds__(256) lsq_count_cleanup_kernel(const CountArgs *Ap, int force_recount, unsigned long long n_pn,
                                                                const ExcEntry *exc, const unsigned *exc_count, unsigned exc_cap) {
	__builtin_amdgcn_s_setprio(3);          // runs beside the next count's streaming kernel: short, and the EM waits for it
	const CountArgs &A = *Ap;               // (through a pointer: no private copy of the record, see recount_all_reads)
	const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	// The kernel is a chain of dependent trips to memory (the EM of the step waits at its end): the list and its length come as
	// kernel arguments of their own, so that the first entry is on its way before the record *Ap has been read, and an entry is
	// loaded before it is known to be one (the list has exc_cap >= 1 slots; what lies beyond the length is an older step's).
	const ExcEntry e_first = exc[gtid < exc_cap ? gtid : 0];
	const unsigned n_raw = exc_count[0];
	// overflow: the list does not hold every pair the fast kernel left open: every read of the method's packed buckets is
	// counted again, from zero (below; lsq_count_status reports it)
	if (n_raw > exc_cap || force_recount) {
		if (gtid == 0) A.exc_count[1] = 1u;
		// ... every packed bucket's class counters are cleared of what the fast kernel and its workers added, and every read of the
		// bucket is counted again -- BY THE SAME WAVE, bucket by bucket: a bucket's reads add to that bucket's counters only, so no
		// workgroup waits for another (until round 4 the launch cleared everything, met at a spin barrier over its workgroups -- which
		// held only for as many workgroups as are resident together, so the launch was kept at 16 -- and then counted).  The rare way
		// through this kernel; it used to be a launch of its own behind this one, which found nothing to do step after step.
		recount_all_reads(A, n_pn);
		return;
	}
	for (unsigned long long k = gtid; k < n_raw; k += gsz) {
		const ExcEntry e = k == gtid ? e_first : exc[k];
		const GlobalBucket G = global_bucket(A, e.bucket);
		const unsigned i = e.ev_pool_scan & 0x1FFFFFFFu, pool = (e.ev_pool_scan >> 29) & 3u;
		const bool scan = (e.ev_pool_scan >> 31) != 0;
		int2 blk[2];
		if (pool == 2) { const unsigned o0 = A.pn_blk_off[e.slot]; eval_read_global(A, G, A.pn_se + o0, (int)A.pn_nblk[e.slot], i, scan, A.pn_strand[e.slot], A.pn_line[e.slot]); }
		else if (pool == 0) { blk[0] = pool1_read(A, e.slot, A.buckets[e.bucket].lo); eval_read_global(A, G, blk, 1, i, scan, A.p1_strand[e.slot], A.p1_line[e.slot]); }
		else { const int4 v = pool2_read(A, e.slot, A.buckets[e.bucket].lo); blk[0] = make_int2(v.x, v.y); blk[1] = make_int2(v.z, v.w); eval_read_global(A, G, blk, 2, i, scan, A.p2_strand[e.slot], A.p2_line[e.slot]); }
	}
}

// The recount, decided on the device (lsq_count_cleanup_kernel's rare way out: the exception list overflowed, or the self-check
// option asks for it): every read of the packed buckets once more -- one lane per read, tables from L2, the reference's
// candidate scan event by event (count/count.cpp:429-464), global atomics.  Slow, complete, and it keeps every table that
// leaves the context whole -- also those handed over by lsq_results_pack_device / lsq_results_copy_device in a loop that
// never looks.  (The arguments come through a pointer, not by value: the evaluation functions take them by reference, and a
// by-value kernel argument whose address is taken is copied to every thread's private segment in the prologue -- 288 bytes
// per thread, written before the flag is even looked at: measured 115 us per launch of a 1 024-workgroup grid.)
__device__ void recount_all_reads(const CountArgs &A, unsigned long long n_pn) {
	const unsigned long long gtid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	const unsigned lane = threadIdx.x & 63u;
	const unsigned wave_id = (unsigned)(gtid >> 6), n_waves = (unsigned)(gsz >> 6);
	(void)n_pn;
	for (unsigned b = wave_id; b < A.n_buckets; b += n_waves) {
		const BucketDesc &d = A.buckets[b];
		if (d.kind != 1) continue;
		for (unsigned i = lane; i < d.n_cls; i += 64u) { A.cnt[d.cls_base + i] = 0; A.bases[d.cls_base + i] = 0; }
		__threadfence();                           // the zeros are in place before this wave's own atomics on the same words
		const GlobalBucket G = global_bucket(A, b);
		for (unsigned long long g = A.pn_off[b] + lane; g < A.pn_off[b + 1]; g += 64u) {
			const unsigned o0 = A.pn_blk_off[g], o1 = o0 + A.pn_nblk[g];
			eval_read_global(A, G, A.pn_se + o0, (int)(o1 - o0), first_event_for(G, A.pn_se[o0].x), true, A.pn_strand[g], A.pn_line[g]);
		}
		for (unsigned long long g = A.p1_off[b] + lane; g < A.p1_off[b + 1]; g += 64u) {
			int2 blk[1] = {pool1_read(A, g, A.buckets[b].lo)};
			if (blk[0].y == blk[0].x) continue;      // padding of a cell's group
			eval_read_global(A, G, blk, 1, first_event_for(G, blk[0].x), true, A.p1_strand[g], A.p1_line[g]);
		}
		for (unsigned long long g = A.p2_off[b] + lane; g < A.p2_off[b + 1]; g += 64u) {
			const int4 v = pool2_read(A, g, A.buckets[b].lo);
			if (v.y == v.x) continue;                // padding of a junction group
			int2 blk[2] = {make_int2(v.x, v.y), make_int2(v.z, v.w)};
			eval_read_global(A, G, blk, 2, first_event_for(G, v.x), true, A.p2_strand[g], A.p2_line[g]);
		}
	}
}

} // namespace

namespace lsq {

// The share plan of a count launch (run_count): `grid` + 1 bounds in slots over buckets whose slots start at so[0 .. B].  Shares
// equal in estimated cost and falling off in size in a line from the first share to the last (`taper` = last / first); a cut
// within 30 % of a share of a bucket boundary moves onto it (`snap`).  No share longer than 2^21 slots (the packed LDS
// counters): else equal shares.  The cost is given as stretches of slots (x[0 .. S], ascending from 0 to the last slot; a
// bucket's stretches from bseg[b]): a stretch of one-block records (kind 0) weighs 1 a record, of two-block records (kind 1)
// `two_block`, plus `walk_look` for every look the general walk will take at the stretch's reads (the ingest's count, per cell
// and junction group: where a hot bucket is cut into many shares it matters WHICH of its groups are the walk's); anything else
// (kind 2: many-block reads, buckets the kernel does not visit) next to nothing; staging + flush of a visited bucket `visit`.
// x == null: every slot weighs 1.
struct SharePlanCosts { bool weighted = true, snap = true; double two_block = 4.3, walk_look = 9.0, visit = 7000.0, taper = 0.5, hot = 0.0; };
void plan_share_cuts_seg(const unsigned long long *so, const unsigned char *visited_kind, const size_t B, const unsigned long long *x, const unsigned char *kind,
                         const unsigned *looks, const unsigned *bseg, const size_t S, const unsigned long long grid, const SharePlanCosts &pc, unsigned long long *cut) {
	const unsigned long long total_slots = B ? so[B] : 0;
	const bool weighted = pc.weighted && x && kind && looks && bseg && S > 0;
	cut[0] = 0;
	if (!weighted) {
		for (unsigned long long g = 1; g <= grid; ++g) cut[(size_t)g] = total_slots / grid * g + total_slots % grid * g / grid;
		// (a cut still snaps to a bucket boundary: as before the weights existed)
		if (pc.snap && grid > 0) {
			const double share = (double)total_slots / (double)grid;
			size_t bb = 0;
			for (unsigned long long g = 1; g < grid; ++g) {
				const unsigned long long t = cut[(size_t)g];
				while (bb + 1 < B && so[bb + 1] <= t) ++bb;
				const unsigned long long lo = so[bb], hi = so[bb + 1];
				unsigned long long u = t;
				if (t - lo <= hi - t && (double)(t - lo) < 0.3 * share) u = lo;
				else if ((double)(hi - t) < 0.3 * share) u = hi;
				cut[(size_t)g] = std::max(u, cut[(size_t)g - 1]);
			}
		}
		cut[(size_t)grid] = total_slots;
		return;
	}
	const double c2 = pc.two_block, cw = pc.walk_look, cv = pc.visit, d_rest = 1e-3;
	const double taper = std::min(1.0, std::max(0.05, pc.taper));
	std::vector<double> dens(S), cum(S + 1, 0.0), jump(S, 0.0), cum_b(B + 1, 0.0);
	for (size_t b = 0; b < B; ++b) if (visited_kind[b] && so[b + 1] > so[b] && bseg[b] < bseg[b + 1]) jump[bseg[b]] = cv;
	for (size_t i = 0; i < S; ++i) {
		const double len = (double)(x[i + 1] - x[i]);
		// (`hot`: a group of 8 192 records or more -- a deep gene's cell -- costs that much more a record: every lane of every wave on it
		// adds to the same few histogram slots, and LDS atomics on one address run one after the other)
		dens[i] = kind[i] == 2 || len <= 0 ? d_rest : (kind[i] == 0 ? 1.0 : c2) + cw * (double)looks[i] / len + (len >= 8192.0 ? pc.hot : 0.0);
		cum[i + 1] = cum[i] + jump[i] + len * dens[i];
	}
	for (size_t b = 0; b <= B; ++b) cum_b[b] = cum[std::min<size_t>(bseg[b], S)];
	const double total_cost = cum[S];
	// share g's part of the whole: falling in a line from 1 to `taper`
	const double wsum = (double)grid * (1.0 + taper) / 2.0;
	auto part_before = [&](unsigned long long g) {          // sum of the weights of shares 0 .. g - 1, over wsum
		const double k = (double)g, slope = grid > 1 ? (taper - 1.0) / (double)(grid - 1) : 0.0;
		return (k + slope * k * (k - 1.0) / 2.0) / wsum;
	};
	size_t bb = 0, si = 0;
	for (unsigned long long g = 1; g < grid && total_cost > 0; ++g) {
		const double tc = total_cost * part_before(g), share = total_cost * (part_before(g + 1) - part_before(g));
		while (bb + 1 < B && cum_b[bb + 1] <= tc) ++bb;
		while (si + 1 < S && cum[si + 1] <= tc) ++si;
		const double xb = tc - cum_b[bb], yb = cum_b[bb + 1] - tc;          // cost of the bucket before / behind the cut
		unsigned long long t;
		if (pc.snap && xb <= yb && xb < 0.3 * share) t = so[bb];
		else if (pc.snap && yb < xb && yb < 0.3 * share) t = so[bb + 1];
		else {
			// inside the stretch: past the staging a bucket begins with, then at the stretch's cost per slot
			const double r = tc - cum[si] - jump[si];
			const unsigned long long len = x[si + 1] - x[si];
			t = x[si] + (r <= 0.0 ? 0ull : std::min<unsigned long long>((unsigned long long)(r / dens[si]), len));
		}
		cut[(size_t)g] = std::max(t, cut[(size_t)g - 1]);
	}
	if (!(total_cost > 0)) for (unsigned long long g = 1; g < grid; ++g) cut[(size_t)g] = total_slots * g / grid;
	cut[(size_t)grid] = total_slots;
	// one workgroup's share must keep the packed LDS counters exact: if the weights made one too long, equal shares
	for (unsigned long long g = 0; g < grid; ++g)
		if (cut[(size_t)g + 1] - cut[(size_t)g] > (1ull << 21)) {
			for (unsigned long long q = 1; q < grid; ++q) cut[(size_t)q] = total_slots * q / grid;
			break;
		}
}

// ... the same from per-bucket numbers (records of the one- and two-block pool and the walk's looks at each as a whole): three
// stretches a bucket (lsq_debug_plan_shares; run_count has the groups' own numbers from the ingest)
void plan_share_cuts(const unsigned long long *so, const unsigned char *visited_kind, const unsigned long long *pn1, const unsigned long long *pn2,
                     const unsigned *look1, const unsigned *look2, const size_t B, const unsigned long long grid, const SharePlanCosts &pc, unsigned long long *cut) {
	if (!(pc.weighted && pn1 && pn2 && look1 && look2)) { plan_share_cuts_seg(so, visited_kind, B, nullptr, nullptr, nullptr, nullptr, 0, grid, pc, cut); return; }
	std::vector<unsigned long long> x; std::vector<unsigned char> kind; std::vector<unsigned> looks, bseg(B + 1, 0);
	for (size_t b = 0; b < B; ++b) {
		bseg[b] = (unsigned)kind.size();
		const unsigned long long ns = so[b + 1] - so[b];
		if (!ns) continue;
		const bool visited = visited_kind[b] != 0;
		const unsigned long long n1 = visited ? std::min(pn1[b], ns) : 0, n2 = visited ? std::min(pn2[b], ns - n1) : 0;
		if (n1) { x.push_back(so[b]); kind.push_back(0); looks.push_back(look1[b]); }
		if (n2) { x.push_back(so[b] + n1); kind.push_back(1); looks.push_back(look2[b]); }
		if (ns - n1 - n2) { x.push_back(so[b] + n1 + n2); kind.push_back(2); looks.push_back(0); }
	}
	bseg[B] = (unsigned)kind.size();
	x.push_back(B ? so[B] : 0);
	plan_share_cuts_seg(so, visited_kind, B, x.data(), kind.data(), looks.data(), bseg.data(), kind.size(), grid, pc, cut);
}

int run_count(lsq_ctx *c) {
	const lsq_events &E = *c->E;
	const size_t n_cls = E.n_cls_total;
	const int M = E.n_methods;
	hipStream_t st_em = nullptr;
	// This count writes the counter set the solve before last read; the previous count had it
	// zeroed on the result stream, behind those readers, and recorded ev_mark after that.  Now the
	// same for the next count: zero the set the latest solve reads, behind it.
	// (Lanes: this count takes lane `set`; the lane of the count before -- its result stream carries that step's EM and
	// hand-off -- gets the zeroing of its counters and its mark queued behind them now.)
	const int set = c->flip ^ 1, other = c->flip;
	hipStream_t st = c->opt_two_count_streams ? c->stream_count2[set] : c->stream;      // (lsq_device.hpp: a count stream per lane)
	if (c->mark_recorded2[set]) HIP_TRY(hipStreamWaitEvent(st, c->ev_mark2[set], 0));
	HIP_TRY(hipMemsetAsync(c->counters.p + (size_t)other * c->counters_per_set, 0, c->counters_per_set * sizeof(unsigned long long), c->stream_em2[other]));
	HIP_TRY(hipEventRecord(c->ev_mark2[other], c->stream_em2[other]));
	c->mark_recorded2[other] = true;
	select_counter_set(c, set);
	st_em = c->stream_em;
	c->fast_launched = 0;
	struct Cleanup { CountArgs A; unsigned long long n_pn; int m; };
	std::vector<Cleanup> cleanups;
	bool counted_signalled = false;
	if (c->time_events) HIP_TRY(hipEventRecord(c->ev0, st));      // ev0..ev1 brackets the count kernel launches only
	const unsigned generic_tables_bytes = (std::max<unsigned>(E.max_lds_bytes, 16) + 15u) & ~15u;
	const unsigned tables_bytes = generic_tables_bytes;
	// Five workgroups a compute unit, not the six that registers and tables would allow: 5 x 80 registers a SIMD leave
	// room for a wave of the EM kernel (104) beside them, so the EM of the step before runs without displacing count waves
	// (measured with LDS padding at the same tables: 6 -> 0.1429, 5 -> 0.1366, 4 -> 0.1433 ms per step on C3).  The LDS
	// request is what holds the number down.  Only where the EM is that kernel, i.e. with a job's worth of events: with a few
	// thousand (configs[1]) the small EM kernel fits anyway and the sixth workgroup is worth 10 % (0.0443 -> 0.0400 ms).  And
	// only for evenly deep read sets: with a few hot buckets (c5s) the step is the same either way (0.178 / 0.176-0.184 ms)
	// and the kernel alone wants the sixth (0.219 -> 0.256 ms).  Option "workgroups_per_cu": -1 = this rule, 0 = as many as
	// fit, n = n.
	double max_skew = 0.0;
	for (int m = 0; m < M; ++m) max_skew = std::max(max_skew, c->reads[m].skew);
	const unsigned cap = c->opt_wg_per_cu >= 0 ? (unsigned)c->opt_wg_per_cu : (c->em_small_places >= c->opt_em_flat_min && max_skew < 4.0 ? 5u : 0u);
	const unsigned lds_bytes = std::max(tables_bytes + WAVES * WAVE_QUEUE_WORDS * 16, cap ? 160u * 1024u / (cap + 1u) + 16u : 0u);
	if (lds_bytes > 160 * 1024) return fail(LSQ_E_UNSUPPORTED, "bucket tables + read tile exceed the CU's LDS");
	// (That is how the rule was found, with the 80-register kernel.  The launches it holds to five now run the 96-register
	// kernel below, which fills the SIMD's registers at five waves by itself and still gains: fewer instructions per read.)
	// Eight reads per look (lsq_count_fast_kernel<true, 4>) where the launch is held to five workgroups a compute unit anyway
	// and every pool is compact (C3, same box, developer builds: 0.1465 -> 0.1418 ms per step; the skewed c5s, which runs six
	// workgroups: 0.189 -> 0.191).  Option "reads_per_look": 0 = this rule, 4, 8.
	bool all_compact = true;
	for (int m = 0; m < M; ++m) if (c->reads[m].total_slots && !c->reads[m].compact) all_compact = false;
	const int p1w = !all_compact ? 2 : (c->opt_reads_per_look ? (c->opt_reads_per_look == 8 ? 4 : 2) : (cap != 0 && cap <= 5 ? 4 : 2));
	const void *const fn_wide = (const void *)lsq_count_fast_kernel<false, 2>;
	const void *const fn_compact = p1w == 4 ? (const void *)lsq_count_fast_kernel<true, 4> : (const void *)lsq_count_fast_kernel<true, 2>;
	c->last_reads_per_look = 2u * (unsigned)p1w;
	if (lds_bytes > 64 * 1024) {
		HIP_TRY(hipFuncSetAttribute(fn_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
		HIP_TRY(hipFuncSetAttribute(fn_compact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
		HIP_TRY(hipFuncSetAttribute((const void *)lsq_count_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
	}
	// resident workgroups per CU as the runtime sees them (registers, LDS, wave slots): the grid is a whole number of rounds
	unsigned per_cu = std::max(1u, std::min(2048u / COUNT_BLOCK, (160u * 1024u) / lds_bytes));
	{
		if (c->occ_lds_bytes != lds_bytes || c->occ_p1w != p1w) {             // asked once per table size and kernel
			int nb = 0;
			int nb2 = 0;
			HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn_wide, (int)COUNT_BLOCK, (size_t)lds_bytes));
			HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, fn_compact, (int)COUNT_BLOCK, (size_t)lds_bytes));
			c->occ_lds_bytes = lds_bytes; c->occ_p1w = p1w; c->occ_blocks = all_compact ? nb2 : std::min(nb, nb2);
		}
		if (c->occ_blocks >= 1) per_cu = std::min(per_cu, (unsigned)c->occ_blocks);
	}
	c->last_wg_per_cu = per_cu;
	for (int m = 0; m < M; ++m) {
		MethodReads &mr = c->reads[m];
		if (mr.total_slots == 0 || E.buckets.empty()) continue;
		// workgroups per resident slot: 2 for even read depth (fewest table stagings), more when a few buckets hold most of the
		// reads and shares differ in cost.  Measured with a count stream per lane, where the next count fills the tail of this
		// one (before that the tail made more and smaller shares pay: 3 and 8): C3 1 -> 0.150, 2 -> 0.143, 3 -> 0.148, 4 ->
		// 0.151 ms per step; the skewed c5s 3 -> 0.179, 4 -> 0.176, 6 -> 0.178, 8 -> 0.183; C2 1 -> 0.046, 2 -> 0.044, 3 -> 0.044.
		// The five-wave kernel's steps are twice as long, and its shares pay for being smaller: C3 1.6 -> 0.137, 2 -> 0.133,
		// 2.4 -> 0.1335, 3 -> 0.129, 3.5 -> 0.132, 4 -> 0.133, 5 -> 0.137, 6 -> 0.1355.
		// lsq_ctx_set_option "grid_multiplier" overrides
		const double mult = c->opt_grid_mult > 0 ? c->opt_grid_mult : (mr.skew >= 32.0 ? 4.0 : (mr.skew >= 4.0 || p1w == 4 ? 3.0 : 2.0));
		unsigned long long grid = (unsigned long long)((double)c->n_cu * per_cu * mult);
		// one workgroup's share must keep the packed LDS counters (24-bit count, 40-bit bases) exact
		grid = std::max(grid, mr.total_slots / (1ull << 21) + 1);
		grid = std::min<unsigned long long>(grid, std::max<unsigned long long>(mr.total_slots / 64, 1));
#ifdef LSQ_DEV
		if (const char *e = getenv("LSQ_GRID_WGS")) { const long v = atol(e); if (v >= 1) grid = (unsigned long long)v; }      // timing experiments
#endif
		if (mr.wg_grid != grid) {
			// The workgroups' shares: equal in COST, with a cut moved onto a bucket boundary when one lies within 30 % of a share
			// of it -- a workgroup that owns whole buckets stages, drains and flushes each once, where a cut through the middle
			// makes two workgroups do it (the planner's buckets of an evenly deep read set are about a share long).
			// Cost (round 3; the workgroup trace of the developer build, tools/kbench.py, fitted over C3's 3 840 shares: with shares
			// equal in reads a workgroup took 11 to 72 us, median 31, and the wave slots of the launch were 68 % used -- the last
			// third of the launch was a thinning tail of late, long shares): a two-block record costs what 4.3 one-block records
			// do, a look of the general walk at a read that the streaming loops leave to it what 9 do (the ingest counts those looks
			// per bucket as it pools the reads: plan_park*), a bucket's staging and flush what 7 000 do.  And the shares get smaller towards the end
			// of the grid ("share_taper": the last is that fraction of the first), so that what is still running when the slots
			// start to empty is short.
			const std::vector<unsigned long long> &so = mr.slot_off_host;
			const size_t B = E.buckets.size();
			std::vector<unsigned long long> cut((size_t)grid + 1, 0);
			std::vector<unsigned> first((size_t)grid, 0);
			const bool weighted = c->opt_share_weighted && mr.plan_n1.size() == B && mr.plan_park1.size() == B;
			// (the pipelined step on one box, taper 1 / 0.5 / 0.25: C3 0.1189 / 0.1165 / 0.1193 ms -- the next count's first workgroups fill this
			// launch's tail there, and smaller last shares cost more stagings than they still save; the skewed c5s 0.1364 / 0.1358 / 0.1300,
			// from 0.1607 with shares equal in reads; C2 0.0404 / 0.0389 / 0.0389 from 0.0400)
			SharePlanCosts pc;
			pc.weighted = weighted; pc.snap = c->opt_snap_shares;
			pc.two_block = c->opt_share_cost_p2; pc.walk_look = c->opt_share_cost_park; pc.visit = c->opt_share_cost_visit;
			pc.taper = c->opt_share_taper > 0 ? c->opt_share_taper : (mr.skew >= 4.0 ? 0.25 : 0.5);
			pc.hot = c->opt_share_cost_hot;
			std::vector<unsigned char> visited(B, 0);
			for (size_t q = 0; q < B; ++q) visited[q] = E.buckets[q].kind == 1u ? 1 : 0;
			if (weighted && mr.plan_seg_x.size() >= 2 && mr.plan_seg_first.size() == B + 1)
				plan_share_cuts_seg(so.data(), visited.data(), B, mr.plan_seg_x.data(), mr.plan_seg_kind.data(), mr.plan_seg_looks.data(), mr.plan_seg_first.data(),
				                    mr.plan_seg_kind.size(), grid, pc, cut.data());
			else
				plan_share_cuts(so.data(), visited.data(), weighted ? mr.plan_n1.data() : nullptr, weighted ? mr.plan_n2.data() : nullptr,
				                weighted ? mr.plan_park1.data() : nullptr, weighted ? mr.plan_park2.data() : nullptr, B, grid, pc, cut.data());
			// the first packed bucket that holds slots of the share [cut[g], cut[g + 1]) (the kernel follows the visit
			// records' links from there); B when there is none
			size_t bb = 0;
			for (unsigned long long g = 0; g < grid; ++g) {
				while (bb + 1 < B && so[bb + 1] <= cut[(size_t)g]) ++bb;
				const unsigned f = mr.next_packed_host[bb];
				first[(size_t)g] = (f < B && so[f] < cut[(size_t)g + 1]) ? f : (unsigned)B;
			}
			int rc = mr.wg_first.upload(first.data(), first.size(), st);
			if (!rc) rc = mr.wg_cut.upload(cut.data(), cut.size(), st);
			if (rc) return rc;
			{
				std::vector<WgPlan> plan((size_t)grid);
				for (unsigned long long g = 0; g < grid; ++g) {
					WgPlan &w = plan[(size_t)g];
					w.s_begin = cut[(size_t)g]; w.s_end = cut[(size_t)g + 1];
					w.first = mr.visits_host[std::min<size_t>(first[(size_t)g], B)];
				}
				if ((rc = mr.wg_plan.upload(plan.data(), plan.size(), st))) return rc;
				HIP_TRY(hipStreamSynchronize(st));          // the host vector goes out of scope
			}
			HIP_TRY(hipStreamSynchronize(st));          // the host vectors go out of scope
			mr.wg_grid = grid;
		}
		CountArgs A{};
		A.buckets = c->buckets.p; A.images = c->images.p; A.ties = c->ties.p; A.strand_rank = c->strand_rank.p;
		A.wg_first = mr.wg_first.p; A.wg_cut = mr.wg_cut.p; A.wg_plan = mr.wg_plan.p; A.visits = mr.visits.p;
		A.read_names = mr.named ? mr.names.p : nullptr; A.read_name_off = mr.named ? mr.name_off.p : nullptr;
		A.gene_names = c->gene_names.p; A.gene_name_off = c->gene_name_off.p;
		A.n_buckets = (unsigned)E.buckets.size();
		A.tables_lds_bytes = tables_bytes;
		A.ablate = c->dev_ablate;
		A.compact = mr.compact ? 1u : 0u;
		A.p1 = mr.p1.p; A.p1_strand = mr.p1_strand.p; A.p1_line = mr.p1_line.p;
		A.p2 = mr.p2.p; A.p2_strand = mr.p2_strand.p; A.p2_line = mr.p2_line.p;
		A.pn_blk_off = mr.pn_blk_off.p; A.pn_nblk = mr.pn_nblk.p; A.pn_se = reinterpret_cast<const int2 *>(mr.pn_se.p);
		A.pn_strand = mr.pn_strand.p; A.pn_line = mr.pn_line.p; A.pn_bucket = mr.pn_bucket.p;
		A.p1_off = mr.p1_off.p; A.p2_off = mr.p2_off.p; A.pn_off = mr.pn_off.p; A.slot_off = mr.slot_off.p;
		A.total_slots = mr.total_slots;
		A.cnt = c->cnt.p + (size_t)m * n_cls; A.bases = c->bases.p + (size_t)m * n_cls;
		A.exc = mr.exc.p + (size_t)set * mr.exc_cap; A.exc_count = c->exc_count.p + 2 * m; A.exc_cap = (unsigned)mr.exc_cap;
		A.dbg = c->dbg.p; A.bar = c->dbg.p + 16 + m;
		A.wg_trace = nullptr;
		const unsigned long long n_pn = mr.pn_strand.n;
		// pool-n workers: one workgroup per CU at most, one lane per read and pass
		A.n_pn = n_pn;
		const unsigned workers_per_cu = 2;          // 1, 4 and 8 measured within 2 % of each other
		A.n_workers = (unsigned)std::min<unsigned long long>((n_pn + COUNT_BLOCK - 1) / COUNT_BLOCK, (unsigned long long)c->n_cu * workers_per_cu);
#ifdef LSQ_DEV
		if (c->dev_ablate & 32768u) A.n_workers = 0;          // timing experiment: no pool-n workers (their reads go uncounted)
#endif
		if (c->has_fast) {
			if (c->time_events) HIP_TRY(hipEventRecord(c->evf0[m], st));
			// the last streaming kernel of the step carries ev_counted as its own completion signal: a separate
			// event record is one more packet between this kernel and the next count's (measured ~5 us each)
			const bool last_streaming = m == M - 1 && !c->has_generic && !c->time_events;
			const dim3 fgrid((unsigned)grid + A.n_workers);
#ifdef LSQ_DEV
			if (c->dev_ablate & 4194304u) {
				if (c->wg_trace.n < 4ull * fgrid.x) { int rc = c->wg_trace.alloc(4ull * fgrid.x); if (rc) return rc; }
				HIP_TRY(hipMemsetAsync(c->wg_trace.p, 0, 4ull * fgrid.x * sizeof(unsigned long long), st));
				c->wg_trace_n = fgrid.x; c->wg_trace_workers = A.n_workers;
				A.wg_trace = c->wg_trace.p;
			}
#endif
			void *kargs[] = {(void *)&A};
			HIP_TRY(hipExtLaunchKernel(mr.compact ? fn_compact : fn_wide, fgrid, dim3(COUNT_BLOCK), kargs, lds_bytes, st, nullptr, last_streaming ? c->ev_counted2[set] : nullptr, 0));
			if (last_streaming) counted_signalled = true;
			if (c->time_events) HIP_TRY(hipEventRecord(c->evf1[m], st));
			c->fast_launched |= 1 << m;
			cleanups.push_back({A, n_pn, m});
		}
		if (c->has_generic) {
			hipLaunchKernelGGL(lsq_count_generic_kernel, dim3((unsigned)grid), dim3(COUNT_BLOCK), generic_tables_bytes, st, A);
			HIP_TRY(hipGetLastError());
		}
	}
	if (c->time_events) HIP_TRY(hipEventRecord(c->ev1, st));
	c->count_timed = c->time_events;
	// the exception pass, and everything that reads the counts, on the result stream behind the streaming kernels;
	// (it turns into the recount where the exception list overflowed)
	if (!counted_signalled) HIP_TRY(hipEventRecord(c->ev_counted2[set], st));
	HIP_TRY(hipStreamWaitEvent(st_em, c->ev_counted2[set], 0));
	// One workgroup a compute unit: the launch normally finds a handful of pairs to settle beside the next count's kernel (C3: a step
	// takes 0.107-0.109 ms with 16, 64, 128 or 256 of them, profiles/r04_recount_c3.json), and when the list has overflowed the same
	// workgroups count every read again -- 14 ms per C3 step on 256, 37 on 64, 90-108 on the 16 of round 3.  No workgroup waits for another.
	const unsigned rgrid = std::max(16u, (unsigned)c->n_cu);
	if (c->recount_args.n < 2 * (size_t)LSQ_MAX_METHODS * sizeof(CountArgs)) {
		int rc = c->recount_args.alloc(2 * (size_t)LSQ_MAX_METHODS * sizeof(CountArgs));
		if (rc) return rc;
		c->recount_args_host.assign(2 * (size_t)LSQ_MAX_METHODS * sizeof(CountArgs), 0);
	}
	for (const Cleanup &u : cleanups) {
		// the arguments of these two kernels live in device memory, one record per (counter set, read file); rewritten only when they change
		const size_t slot = ((size_t)set * LSQ_MAX_METHODS + (size_t)u.m) * sizeof(CountArgs);
		if (memcmp(c->recount_args_host.data() + slot, &u.A, sizeof(CountArgs)) != 0) {
			memcpy(c->recount_args_host.data() + slot, &u.A, sizeof(CountArgs));
			HIP_TRY(hipMemcpyAsync(c->recount_args.p + slot, c->recount_args_host.data() + slot, sizeof(CountArgs), hipMemcpyHostToDevice, st_em));
		}
		const CountArgs *dA = reinterpret_cast<const CountArgs *>(c->recount_args.p + slot);
		hipLaunchKernelGGL(lsq_count_cleanup_kernel, dim3(c->opt_cleanup_grid ? c->opt_cleanup_grid : rgrid), dim3(256), 0, st_em, dA, c->opt_recount ? 1 : 0, u.n_pn,
		                   (const ExcEntry *)u.A.exc, (const unsigned *)u.A.exc_count, u.A.exc_cap);
		HIP_TRY(hipGetLastError());
	}
	return LSQ_OK;
}

} // namespace lsq

extern "C" {

// per read file of the latest lsq_count: pairs handed to the exception pass, and whether the recount ran
int lsq_count_status(lsq_ctx *c, uint32_t *exceptions, uint32_t *recounted) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	std::vector<unsigned> h(c->exc_count.n);
	HIP_TRY(hipMemcpy(h.data(), c->exc_count.p, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
	note_overflow(c, h);
	for (int m = 0; m < c->E->n_methods; ++m) {
		if (exceptions) exceptions[m] = h[2 * (size_t)m];
		if (recounted) recounted[m] = h[2 * (size_t)m + 1];
	}
	return LSQ_OK;
} LSQ_API_CATCH

// how the latest lsq_count launched its streaming kernel
int lsq_count_launch_info(lsq_ctx *c, uint32_t *reads_per_look, uint32_t *workgroups_per_cu) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	if (reads_per_look) *reads_per_look = c->last_reads_per_look;
	if (workgroups_per_cu) *workgroups_per_cu = c->last_wg_per_cu;
	return LSQ_OK;
} LSQ_API_CATCH

// developer aid (include/lesseq_hip_dev.h): counters filled when LSQ_ABLATE & 256
int lsq_debug_counters(lsq_ctx *c, unsigned long long *out8) LSQ_API_TRY {
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	HIP_TRY(hipMemcpy(out8, c->dbg.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	std::vector<unsigned> h(c->exc_count.n);
	HIP_TRY(hipMemcpy(h.data(), c->exc_count.p, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
	out8[4] = h[0];
	return LSQ_OK;
} LSQ_API_CATCH

// developer aid (include/lesseq_hip_dev.h): (start, end, walk steps, reads walked) of the last count launch's workgroups, 100 MHz
// ticks; returns their number through *n (out holds 4 * cap values), the pool-n workers among them (the first ones) through *n_workers
int lsq_debug_wg_trace(lsq_ctx *c, unsigned long long *out, unsigned long long cap, unsigned long long *n, unsigned long long *n_workers) LSQ_API_TRY {
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	const unsigned long long k = std::min<unsigned long long>(cap, c->wg_trace_n);
	if (k) HIP_TRY(hipMemcpy(out, c->wg_trace.p, 4 * k * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	*n = k; *n_workers = c->wg_trace_workers;
	return LSQ_OK;
} LSQ_API_CATCH

// developer aid (include/lesseq_hip_dev.h): the share plan of a count launch as run_count makes it, on given per-bucket numbers (host
// only; costs: two-block record, walk look, visit, taper; weighted / snap as 0 / 1)
int lsq_debug_plan_shares(const unsigned long long *slot_off, const unsigned char *packed, const unsigned long long *n1, const unsigned long long *n2,
                          const unsigned *look1, const unsigned *look2, unsigned long long n_buckets, unsigned long long grid,
                          const double *costs4, int weighted, int snap, unsigned long long *cuts) LSQ_API_TRY {
	if (!slot_off || !packed || !costs4 || !cuts || grid == 0) return fail(LSQ_E_ARG, "null argument");
	lsq::SharePlanCosts pc;
	pc.weighted = weighted != 0; pc.snap = snap != 0;
	pc.two_block = costs4[0]; pc.walk_look = costs4[1]; pc.visit = costs4[2]; pc.taper = costs4[3];
	lsq::plan_share_cuts(slot_off, packed, n1, n2, look1, look2, (size_t)n_buckets, grid, pc, cuts);
	return LSQ_OK;
} LSQ_API_CATCH

// developer aid (include/lesseq_hip_dev.h): per-bucket slot offsets of a method (n_buckets + 1 values)
int lsq_debug_slot_offsets(lsq_ctx *c, int method, unsigned long long *out, unsigned long long n) LSQ_API_TRY {
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	const MethodReads &mr = c->reads[method];
	if (n > mr.slot_off.n) n = mr.slot_off.n;
	HIP_TRY(hipMemcpy(out, mr.slot_off.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return LSQ_OK;
} LSQ_API_CATCH

// developer aid (include/lesseq_hip_dev.h): an offset table of a method -- 0: slots per bucket, 1 / 2: one- / two-block pool
// per bucket (n_buckets + 1 values each), 3: the count launch's share bounds (workgroups + 1); *n = values written
int lsq_debug_offsets(lsq_ctx *c, int method, int which, unsigned long long *out, unsigned long long cap, unsigned long long *n) LSQ_API_TRY {
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	const MethodReads &mr = c->reads[method];
	if (which == 4 || which == 5) {          // the ingest's estimate of the reads bound for the general walk, per bucket (one- / two-block)
		const std::vector<unsigned> &v = which == 4 ? mr.plan_park1 : mr.plan_park2;
		const unsigned long long k = std::min<unsigned long long>(cap, v.size());
		for (unsigned long long i = 0; i < k; ++i) out[i] = v[i];
		*n = k;
		return LSQ_OK;
	}
	const DevBuf<unsigned long long> &src = which == 0 ? mr.slot_off : (which == 1 ? mr.p1_off : (which == 2 ? mr.p2_off : mr.wg_cut));
	const unsigned long long k = std::min<unsigned long long>(cap, src.n);
	if (k) HIP_TRY(hipMemcpy(out, src.p, k * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	*n = k;
	return LSQ_OK;
} LSQ_API_CATCH

} // extern "C"
