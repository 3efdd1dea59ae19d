// Loader kernels of the device group: MRF text parsed in HBM, the load-time containment filter, block
// merge and the bucket / bin / pool layout (count/count.cpp:279-364), and the entry points around them.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>

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
	const unsigned *clu_off;       // per chromosome id: range of its event clusters (merged spans of the planned events)
	const int *clu_s, *clu_e;
	const BucketDesc *buckets;
	const unsigned *bin_base;
	const unsigned *cell_base;      // per bucket: first of its one-block groups (its cells, then "no cell")
	const unsigned long long *jg_keys; const unsigned *jg_base, *jgroup_base;      // junction groups of the two-block pool (lsq_events::jg_keys)
	const unsigned char *images;    // the buckets' LDS images (bin records and cells of packed buckets)      // per bucket: first of its bins in the fine counters (n_buckets + 1)
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
	unsigned *fine;                // per read: its group -- one-block reads cell_base[bucket] + cell (or the bucket's "no cell" group), two-block reads jgroup_base[bucket] + junction group (or the bucket's last)
	unsigned char *nb;             // per read: merged blocks
	unsigned char *strand;         // per read: strand id of the last kept block
	int *ms, *me;                  // merged blocks, at the read's original block offset
	unsigned *cnt1, *cnt2;         // one-block reads per group [n_cell_groups]; two-block reads per group [n_junction_groups]
	unsigned *cntn, *cntnb;        // [n_buckets]: n-block reads, and their blocks
	unsigned *park1, *park2;       // [n_cell_groups] / [n_junction_groups]: looks of the general walk at the group's reads that the count kernel's streaming loops will leave to it (an estimate, for the share plan)
	unsigned *cur1, *cur2, *curn, *curnb;   // scatter cursors, same shapes
	unsigned long long *totals;    // [0] retained reads, [1] retained blocks, [2] error flag, [3] one- and two-block reads that do not fit compact records, [4] / [5] one- / two-block reads pooled
	unsigned compact;              // compact pool records: one- and two-block reads that do not fit them go with the many-block reads
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
	unsigned long long kept_reads = 0, kept_blocks = 0, misfits = 0, pooled1 = 0, pooled2 = 0;
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
					// the first base must lie in the span of some planned event (a cluster): otherwise the read is a
					// candidate of none of them (count/count.cpp:429-432,463) -- with a shard, the other shards' reads
					bool in_cluster = false;
					{
						const unsigned u0 = T.clu_off[chrom], u1 = T.clu_off[chrom + 1];
						unsigned ul = u0, uh = u1;                 // upper_bound(cluster starts, p)
						while (ul < uh) { const unsigned mid = (ul + uh) >> 1; if (T.clu_s[mid] <= s[0]) ul = mid + 1; else uh = mid; }
						in_cluster = ul > u0 && s[0] <= T.clu_e[ul - 1];
					}
					if (in_cluster && s[0] <= T.buckets[b].hi) {
						const BucketDesc &d = T.buckets[b];
						unsigned pool = n == 1 ? 0u : (n == 2 ? 1u : 2u);
						if (pool < 2u && W.compact) {
							bool fits = lsq::compact_block_fits((long long)s[0] - d.lo + lsq::COMPACT_BIAS, (long long)e[0] - s[0]);
							if (n == 2) fits = fits && lsq::compact_block_fits((long long)s[1] - e[0], (long long)e[1] - s[1]);
							if (!fits) { pool = 2u; ++misfits; }
						}
						key = b * 4u + pool;
						const int rel = s[0] - d.lo;
						const unsigned bin = rel <= 0 ? 0u : min((unsigned)rel >> d.shift, d.n_bins - 1u);
						unsigned fine = 0, looks_est = 0;
						if (pool < 2u) {
							// the read's cell, found as the count kernel finds it (bin record: first cell | first event << 16, the
							// ends of that cell and the next two; then on through the cell table)
							unsigned n_cells = 0, cell = 0;
							bool in_junction_group = false;
							if (d.kind == 1u) {
								const unsigned char *img = T.images + d.img_off;
								const uint4 br = reinterpret_cast<const uint4 *>(img)[bin];
								const lsq::Cell *cells = reinterpret_cast<const lsq::Cell *>(img + d.seg_off);
								n_cells = d.iso_off & 0xFFFFu;
								const int p = s[0];
								cell = (br.x & 0xFFFFu) + (unsigned)(p >= (int)br.y) + (unsigned)(p >= (int)br.z) + (unsigned)(p >= (int)br.w);
								if (p >= (int)br.w) while (cell + 1u < n_cells && p >= cells[cell + 1u].lo) ++cell;
								if (!(cell < n_cells && cells[cell].lo <= p && p < cells[cell].hi)) cell = n_cells;
								if (pool == 1u) {
									// the junction group: block 1 ends on the end of the cell owner's segment, block 2 starts where a
									// later segment of that event does (the keys hold exactly those starts); else the bucket's last group
									const unsigned k0 = T.jg_base[b], k1 = T.jg_base[b + 1];
									unsigned g = k1;
									if (cell < n_cells && (e[0] == cells[cell].e1 || e[0] == cells[cell].e2)) {
										const unsigned long long want = lsq::jg_key(cell, e[0] == cells[cell].e1 ? 0u : 1u, s[1]);
										unsigned lo_k = k0, hi_k = k1;
										while (lo_k < hi_k) { const unsigned mid = (lo_k + hi_k) >> 1; if (T.jg_keys[mid] < want) lo_k = mid + 1; else hi_k = mid; }
										if (lo_k < k1 && T.jg_keys[lo_k] == want) g = lo_k;
									}
									// (no junction: the group of the read's cell -- `n_cells`: of no cell -- behind the junction groups)
									fine = T.jgroup_base[b] + (g < k1 ? g - k0 : (k1 - k0) + cell);
									in_junction_group = g < k1;
								}
								// Will the streaming loop settle the read, or leave it to the general walk -- and how many events will the walk look
								// at for it?  What the parked reads cost beside the streamed ones is what makes buckets differ: the share plan weighs
								// it (run_count).  An estimate: the loops' rules in short; the walk's own stepping rule (fast_trip's return).
								bool parks = cell == n_cells, one_event = false;
								if (!parks && !in_junction_group) {
									const lsq::CellX *cellx = reinterpret_cast<const lsq::CellX *>(img + d.seg_off + 16u * (d.iso_off >> 16));
									const lsq::Cell cw = cells[cell];
									const unsigned fl = cellx[cell].flags;
									const int len = e[0] - s[0];
									const bool both = (fl & lsq::CELLX_BOTH) != 0u;
									const bool near_free = (fl & lsq::CELLX_NEAR_NO_ABUT) != 0u, far_free = (fl & lsq::CELLX_FAR_NO_ABUT) != 0u;
									const bool near_done = e[0] <= cw.e1 || (near_free && 50 * (e[0] - cw.e1) >= len), far_done = e[0] <= cw.e2 || (far_free && 50 * (e[0] - cw.e2) >= len);
									if (pool == 0u) { parks = both ? !(near_done && far_done) : e[0] > cw.e2; one_event = !both || near_done || far_done; }
									else if (cellx[cell].info == lsq::CELL_INFO_EMPTY) parks = false;
									else parks = both ? (e[0] == cw.e1 || e[0] == cw.e2 || (e[0] > cw.e1 && !near_free) || (e[0] > cw.e2 && !far_free)) : e[0] > cw.e2;
								}
								if (parks) {
									unsigned looks = 1;
									if (!one_event) {
										const lsq::FastRec *recs = reinterpret_cast<const lsq::FastRec *>(img + d.ev_off);
										looks = 0;
										for (unsigned i = br.x >> 16; i < d.n_events && looks < 64u; ++i) {
											++looks;
											const lsq::FastRec &fr = recs[i];
											if (!(fr.seg[0] <= p && (p > fr.ge || (fr.meta & lsq::FAST_FLAG_OVERLAPS_NEXT) != 0u))) break;
										}
									}
									looks_est = looks;
								}
							} else if (pool == 1u) fine = T.jgroup_base[b] + (T.jg_base[b + 1] - T.jg_base[b]);
							if (pool == 0u) fine = T.cell_base[b] + cell;
						}
						W.fine[i] = fine;
						if (looks_est) atomicAdd(pool == 0u ? &W.park1[fine] : &W.park2[fine], looks_est);
						if (pool == 0) { atomicAdd(&W.cnt1[fine], 1u); ++pooled1; }
						else if (pool == 1) { atomicAdd(&W.cnt2[fine], 1u); ++pooled2; }
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
	if (misfits) atomicAdd(&W.totals[3], misfits);
	if (pooled1) atomicAdd(&W.totals[4], pooled1);
	if (pooled2) atomicAdd(&W.totals[5], pooled2);
}

// one workgroup: out[i] = sum of in[0..i), out[n] = total; every value rounded up to a multiple of PAD (a power of two) first
template <unsigned PAD>
__global__ void __launch_bounds__(1024) lsq_scan_u32_kernel(const unsigned *in_raw, unsigned long long n, unsigned long long *out) {
	auto in = [&](unsigned long long b) { const unsigned v = in_raw[b]; return (v + (PAD - 1u)) & ~(PAD - 1u); };
	__shared__ unsigned long long part[1024];
	const unsigned tid = threadIdx.x;
	const unsigned long long per = (n + 1023ull) / 1024ull;
	const unsigned long long b0 = min(tid * per, n), b1 = min(b0 + per, n);
	unsigned long long acc = 0;
	for (unsigned long long b = b0; b < b1; ++b) acc += in(b);
	part[tid] = acc;
	__syncthreads();
	if (tid == 0) { unsigned long long run = 0; for (unsigned t = 0; t < 1024; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; } }
	__syncthreads();
	unsigned long long run = part[tid];
	for (unsigned long long b = b0; b < b1; ++b) { out[b] = run; run += in(b); }
	if (tid == 1023) out[n] = run;
}

// per-bucket pool offsets out of the per-bin ones
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

struct IngestOut {
	void *p1; unsigned char *p1_strand; unsigned *p1_line;           // the one-block pool itself (groups by cell: no sort follows)
	unsigned compact; const BucketDesc *buckets;
	void *p2; unsigned char *p2_strand; unsigned *p2_line;           // the two-block pool itself (groups by junction)
	unsigned *pn_blk_off, *pn_nblk, *pn_line, *pn_bucket; unsigned char *pn_strand; int2 *pn_se;
	const unsigned long long *off1, *off2, *pn_off, *pnb_off;       // per one-block group / per (bucket, bin) / per bucket
};

__global__ void __launch_bounds__(256) lsq_ingest_scatter_kernel(IngestRaw R, IngestWork W, IngestOut O) {
	const unsigned long long gsz = (unsigned long long)gridDim.x * blockDim.x;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < R.n_reads; i += gsz) {
		const unsigned key = W.key[i];
		if (key == INGEST_NO_KEY) continue;
		const unsigned b = key >> 2, pool = key & 3u;
		const unsigned long long b0 = R.blk_off[i];
		if (pool == 0) {
			const unsigned group = W.fine[i];
			const unsigned long long w = O.off1[group] + atomicAdd(&W.cur1[group], 1u);
			const int2 r = make_int2(W.ms[b0], W.me[b0]);
			const int base = O.buckets[b].lo - lsq::COMPACT_BIAS;
			if (O.compact) pool_store<true>(O.p1, w, r, base); else pool_store<false>(O.p1, w, r, base);
			O.p1_strand[w] = W.strand[i]; O.p1_line[w] = R.line_no[i];
		} else if (pool == 1) {
			const unsigned group = W.fine[i];
			const unsigned long long w = O.off2[group] + atomicAdd(&W.cur2[group], 1u);
			const int4 r = make_int4(W.ms[b0], W.me[b0], W.ms[b0 + 1], W.me[b0 + 1]);
			const int base = O.buckets[b].lo - lsq::COMPACT_BIAS;
			if (O.compact) pool_store<true>(O.p2, w, r, base); else pool_store<false>(O.p2, w, r, base);
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

#include "lsq_mrf_device.hpp"

// Runs the three ingest kernels over parsed blocks that are already on the device (file order).
static int ingest_device(lsq_ctx *c, int method, const IngestRaw &Rw, uint64_t nblk) {
	HostStopwatch SW;
	const lsq_events &E = *c->E;
	MethodReads &mr = c->reads[method];
	mr.present = false;
	const unsigned B = (unsigned)E.buckets.size();
	const uint64_t n = Rw.n_reads;
	hipStream_t st = c->stream;
	int rc;
	// the pools of this method are rewritten below: an exception pass of an earlier count may still read them on the result stream
	HIP_TRY(hipStreamSynchronize(c->stream_em2[0]));
	HIP_TRY(hipStreamSynchronize(c->stream_em2[1]));
	HIP_TRY(hipStreamSynchronize(c->stream_count2[0]));      // ... and a count of an earlier read set on a lane's count stream
	HIP_TRY(hipStreamSynchronize(c->stream_count2[1]));
	DevBuf<int> d_ms, d_me;
	DevBuf<unsigned char> d_nb, d_strand;
	DevBuf<unsigned> d_key, d_fine;
	DevBuf<unsigned> d_cnt;                      // cnt1 | cnt2 | cntn | cntnb, then the four cursor arrays
	DevBuf<unsigned> d_park;                     // park1 | park2
	DevBuf<unsigned long long> d_off1, d_off2, d_totals;
	const size_t FC = c->n_cell_groups;          // one-block groups of all buckets
	const size_t FJ = c->n_junction_groups;      // two-block groups of all buckets
	const size_t n_cnt = FC + FJ + 2 * (size_t)B;
	if ((rc = d_ms.alloc(nblk)) || (rc = d_me.alloc(nblk)) || (rc = d_nb.alloc(n)) || (rc = d_strand.alloc(n)) || (rc = d_key.alloc(n)) || (rc = d_fine.alloc(n))) return rc;
	if ((rc = d_cnt.alloc(2 * n_cnt)) || (rc = d_off1.alloc(FC + 1)) || (rc = d_off2.alloc(FJ + 1)) || (rc = d_totals.alloc(8)) || (rc = d_park.alloc(FC + FJ))) return rc;
	IngestTables T{};
	T.cov_off = c->cov_off.p; T.cov_s = c->cov_s.p; T.cov_e = c->cov_e.p;
	T.cut_off = c->cut_off.p; T.cut_lo = c->cut_lo.p; T.chrom_first_bucket = c->chrom_first_bucket.p;
	T.clu_off = c->clu_off.p; T.clu_s = c->clu_s.p; T.clu_e = c->clu_e.p;
	T.buckets = c->buckets.p; T.bin_base = c->bin_base.p; T.n_chrom = c->n_chrom_tables;
	T.cell_base = c->cell_base.p; T.images = c->images.p;
	T.jg_keys = c->jg_keys.p; T.jg_base = c->jg_base.p; T.jgroup_base = c->jgroup_base.p;
	IngestWork W{};
	W.key = d_key.p; W.fine = d_fine.p; W.nb = d_nb.p; W.strand = d_strand.p; W.ms = d_ms.p; W.me = d_me.p;
	W.cnt1 = d_cnt.p; W.cnt2 = W.cnt1 + FC; W.cntn = W.cnt2 + FJ; W.cntnb = W.cntn + B;
	W.cur1 = d_cnt.p + n_cnt; W.cur2 = W.cur1 + FC; W.curn = W.cur2 + FJ; W.curnb = W.curn + B;
	W.totals = d_totals.p;
	W.park1 = d_park.p; W.park2 = d_park.p + FC;
	W.compact = c->opt_compact_pools ? 1u : 0u;
	const unsigned igrid = (unsigned)std::min<unsigned long long>((n + 255) / 256 + 1, (unsigned long long)c->n_cu * 16);
	if ((rc = mr.p1_off.alloc(B + 1)) || (rc = mr.p2_off.alloc(B + 1)) || (rc = mr.pn_off.alloc(B + 1)) || (rc = mr.pnb_off.alloc(B + 1)) || (rc = mr.slot_off.alloc(B + 1))) return rc;
	unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sums[4] = {0, 0, 0, 0};
	for (;;) {
		HIP_TRY(hipMemsetAsync(d_cnt.p, 0, std::max<size_t>(2 * n_cnt, 1) * 4, st));
		HIP_TRY(hipMemsetAsync(d_totals.p, 0, 8 * 8, st));
		HIP_TRY(hipMemsetAsync(d_park.p, 0, std::max<size_t>(FC + FJ, 1) * 4, st));
		if (n) {
			hipLaunchKernelGGL(lsq_ingest_classify_kernel, dim3(igrid), dim3(256), 0, st, T, Rw, W);
			HIP_TRY(hipGetLastError());
		}
		hipLaunchKernelGGL(lsq_scan_u32_kernel<P1_GROUP_PAD>, dim3(1), dim3(1024), 0, st, W.cnt1, (unsigned long long)FC, d_off1.p);      // groups padded to eight records, and to four
		hipLaunchKernelGGL(lsq_scan_u32_kernel<P2_GROUP_PAD>, dim3(1), dim3(1024), 0, st, W.cnt2, (unsigned long long)FJ, d_off2.p);
		hipLaunchKernelGGL(lsq_scan_u32_kernel<1>, dim3(1), dim3(1024), 0, st, W.cntn, (unsigned long long)B, mr.pn_off.p);
		hipLaunchKernelGGL(lsq_scan_u32_kernel<1>, dim3(1), dim3(1024), 0, st, W.cntnb, (unsigned long long)B, mr.pnb_off.p);
		hipLaunchKernelGGL(lsq_ingest_offsets_kernel, dim3(B / 256 + 1), dim3(256), 0, st, c->cell_base.p, c->jgroup_base.p, B, d_off1.p, d_off2.p, mr.pn_off.p,
		                   mr.p1_off.p, mr.p2_off.p, mr.slot_off.p);
		HIP_TRY(hipGetLastError());
		SW.mark("ingest: allocs + classify launch");
		HIP_TRY(hipMemcpyAsync(tot, d_totals.p, 8 * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[0], mr.p1_off.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[1], mr.p2_off.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[2], mr.pn_off.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(&sums[3], mr.pnb_off.p + B, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		SW.mark("ingest: classify + scans done");
		// compact records pay when nearly every one- and two-block read fits them (the others are counted a lane a read, tables
		// in L2); a read set of long blocks -- more than 1 in 16 does not fit -- is classified again for wide records
		if (W.compact && tot[3] * 16 > tot[4] + tot[5] + tot[3]) { W.compact = 0; continue; }
		break;
	}
	if (tot[2]) return fail(LSQ_E_RANGE, "a read covers 2^18 or more bases or keeps more than %d separate blocks: outside the device tables' range", INGEST_MAX_BLOCKS);
	if (sums[3] > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "too many blocks in multi-block reads");
	if (n > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "more than 2^32 reads in one file");
	const size_t n1 = (size_t)sums[0], n2 = (size_t)sums[1], nn = (size_t)sums[2], nnb = (size_t)sums[3];
	mr.compact = W.compact != 0;
	mr.n1_reads = tot[4]; mr.n2_reads = tot[5];
	// (n1, n2: the pools' slots, the groups' padding among them: multiples of four and of two, so compact pools are whole
	// 16-byte words)
	if ((rc = mr.p1.alloc(mr.compact ? ((n1 + 3) & ~(size_t)3) : 2 * n1)) || (rc = mr.p1_strand.alloc(n1)) || (rc = mr.p1_line.alloc(n1))) return rc;
	if ((rc = mr.p2.alloc(mr.compact ? 2 * ((n2 + 1) & ~(size_t)1) : 4 * n2)) || (rc = mr.p2_strand.alloc(n2)) || (rc = mr.p2_line.alloc(n2))) return rc;
	if ((rc = mr.pn_se.alloc(2 * nnb)) || (rc = mr.pn_blk_off.alloc(nn)) || (rc = mr.pn_nblk.alloc(nn)) || (rc = mr.pn_strand.alloc(nn)) ||
	    (rc = mr.pn_line.alloc(nn)) || (rc = mr.pn_bucket.alloc(nn))) return rc;
	if (n) {
		// the scatter writes the pools themselves: their groups need no order inside
		IngestOut O{};
		O.p1 = mr.p1.p; O.p1_strand = mr.p1_strand.p; O.p1_line = mr.p1_line.p;
		O.compact = mr.compact ? 1u : 0u; O.buckets = c->buckets.p;
		O.p2 = mr.p2.p; O.p2_strand = mr.p2_strand.p; O.p2_line = mr.p2_line.p;
		O.pn_blk_off = mr.pn_blk_off.p; O.pn_nblk = mr.pn_nblk.p; O.pn_line = mr.pn_line.p; O.pn_bucket = mr.pn_bucket.p;
		O.pn_strand = mr.pn_strand.p; O.pn_se = reinterpret_cast<int2 *>(mr.pn_se.p);
		O.off1 = d_off1.p; O.off2 = d_off2.p; O.pn_off = mr.pn_off.p; O.pnb_off = mr.pnb_off.p;
		hipLaunchKernelGGL(lsq_ingest_scatter_kernel, dim3(igrid), dim3(256), 0, st, Rw, W, O);
		const unsigned pgrid = (unsigned)std::min<size_t>((FC + FJ) / 256 + 1, (size_t)c->n_cu * 8);
		hipLaunchKernelGGL(lsq_ingest_pad_kernel, dim3(pgrid), dim3(256), 0, st, c->buckets.p, c->cell_base.p, c->jgroup_base.p, B, (unsigned)FC, (unsigned)FJ,
		                   W.cnt1, d_off1.p, W.cnt2, d_off2.p, c->images.p, (void *)mr.p1.p, (void *)mr.p2.p, O.compact, mr.p1_strand.p, mr.p1_line.p, mr.p2_strand.p, mr.p2_line.p);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(st));            // the work arrays go out of scope at the end of this function
		SW.mark("ingest: scatter done");
	}
	SW.mark("ingest: pool temporaries freed");
	{
		// exception list: a quarter of the one- and two-block reads, at least 64 Ki entries
		size_t want = std::max<size_t>(65536, (n1 + n2) / 4);
		if (c->opt_exc_cap) want = c->opt_exc_cap;           // lsq_ctx_set_option "exception_capacity" (the tests provoke the overflow path with it)
		if (mr.exc_cap != want) {
			if ((rc = mr.exc.alloc(2 * want))) return rc;
			mr.exc_cap = want;
		}
	}
	SW.mark("ingest: exception list alloc");
	if ((rc = upload_strand_ranks(c))) return rc;      // the reads may have introduced new strand strings
	HIP_TRY(hipStreamSynchronize(st));
	SW.mark("ingest: strand ranks");
	mr.n_retained = tot[0];
	mr.n_retained_blocks = tot[1];
	mr.total_slots = n1 + n2 + nn;
	mr.wg_grid = 0;
	{
		// how unevenly the reads fall on the buckets: with hot genes the reads that need the general walk
		// fill whole workgroup shares, and smaller shares (more workgroups) even the load out
		std::vector<unsigned long long> so(B + 1, 0);
		HIP_TRY(hipMemcpyAsync(so.data(), mr.slot_off.p, (B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		unsigned long long mx = 0;
		for (unsigned b = 0; b < B; ++b) mx = std::max(mx, so[b + 1] - so[b]);
		mr.slot_off_host = so;
		mr.skew = (B && mr.total_slots) ? (double)mx * (double)B / (double)mr.total_slots : 1.0;
		// the count kernel's visit records (lsq_device.hpp VisitRec)
		std::vector<unsigned long long> o1(B + 1, 0), o2(B + 1, 0);
		HIP_TRY(hipMemcpyAsync(o1.data(), mr.p1_off.p, (B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(o2.data(), mr.p2_off.p, (B + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		// ... and what the share plan weighs: the records of every cell and junction group and the looks the general walk will take at
		// them (run_count: plan_share_cuts_seg), as stretches of slots; per bucket the sums as well
		{
			std::vector<unsigned long long> f1(FC + 1, 0), f2(FJ + 1, 0);
			std::vector<unsigned> park(FC + FJ + 1, 0);
			HIP_TRY(hipMemcpyAsync(f1.data(), d_off1.p, (FC + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(f2.data(), d_off2.p, (FJ + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
			if (FC + FJ) HIP_TRY(hipMemcpyAsync(park.data(), d_park.p, (FC + FJ) * sizeof(unsigned), hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
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
	SW.mark("ingest: skew, strand ranks");
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
	IngestRaw Rw{};
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
} LSQ_API_CATCH

int lsq_reads_upload_mrf(lsq_ctx *c, int method, const char *read_format, const char *path) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (method < 0 || method >= c->E->n_methods) return fail(LSQ_E_ARG, "method %d out of range", method);
	HIP_TRY(hipSetDevice(c->device));
	HostStopwatch SW;
	DevParsed P;
	int rc = device_parse_mrf(c, read_format, path, P, &c->mrf_h2d_ms, &c->mrf_parse_ms);
	SW.mark("parse: all (incl. unmap, frees)");
	if (rc) return rc;
	IngestRaw Rw{};
	Rw.n_reads = P.n_reads; Rw.blk_off = P.blk_off.p; Rw.line_no = P.line_no.p; Rw.blk_start = P.bs.p; Rw.blk_end = P.be.p;
	Rw.blk_chrom = P.bc.p; Rw.blk_strand = P.bst.p;
	c->reads[method].named = false;
	return ingest_device(c, method, Rw, P.n_blocks);
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
	int rc = scan_newlines(c, *t);
	if (rc) return rc;
	*n_newlines = t->n_nl;
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
	DevParsed P;
	int rc = parse_staged_text(c, read_format, *t, has_header ? 1u : 0u, first_line, P, &c->mrf_h2d_ms, &c->mrf_parse_ms);
	if (rc) return rc;
	IngestRaw Rw{};
	Rw.n_reads = P.n_reads; Rw.blk_off = P.blk_off.p; Rw.line_no = P.line_no.p; Rw.blk_start = P.bs.p; Rw.blk_end = P.be.p;
	Rw.blk_chrom = P.bc.p; Rw.blk_strand = P.bst.p;
	c->reads[method].named = false;
	return ingest_device(c, method, Rw, P.n_blocks);
} LSQ_API_CATCH

int lsq_mrf_parse_device(lsq_ctx *c, const char *read_format, const char *path, lsq_reads **out) LSQ_API_TRY {
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
} LSQ_API_CATCH

int lsq_last_mrf_timing(lsq_ctx *c, float *h2d_ms, float *parse_ms) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (h2d_ms) *h2d_ms = c->mrf_h2d_ms;
	if (parse_ms) *parse_ms = c->mrf_parse_ms;
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
