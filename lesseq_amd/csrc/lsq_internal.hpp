// Internal structures shared by the host-side translation units and the HIP layer.
// Nothing here is part of the ABI (include/lesseq_hip.h is).
#pragma once

#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/lesseq_hip.h"

namespace lsq {

// ---- error text for the calling thread -------------------------------------------------
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void warn(const char *fmt, ...) __attribute__((format(printf, 1, 2)));      // "[LOG ... WARNING] text" on stderr (lsq_set_log_level >= 1)
void set_log_level(int level);

// ---- no exception crosses the C ABI (include/lesseq_hip.h; the reference's contract for a failure is a logged message
// and `return 1`, count/count.cpp:20-38) -----------------------------------------------------------------------------
// Every extern "C" entry that can allocate is a function-try-block: `int lsq_x(...) LSQ_API_TRY { ... } LSQ_API_CATCH`.
// std::bad_alloc / std::length_error / anything else thrown below it comes back as LSQ_E_INTERNAL with the text in
// lsq_last_error() instead of std::terminate in the host process.
int fail_exception(const char *where) noexcept;
#define LSQ_API_TRY try
#define LSQ_API_CATCH catch (...) { return lsq::fail_exception(__func__); }

// Helper threads of one call.  A std::thread that is destroyed while joinable, and an exception that leaves a thread's
// function, both end the process (std::terminate): the group joins whatever it started on every way out of its scope, and
// a body's exception is kept as text for the caller (failed() / error()).
class ThreadGroup {
	std::vector<std::thread> th_;
	std::mutex mu_;
	std::string error_;
	bool failed_ = false;
	void note(const char *what) noexcept {
		try { std::lock_guard<std::mutex> g(mu_); if (!failed_) { failed_ = true; error_ = what; } } catch (...) { failed_ = true; }
	}
public:
	ThreadGroup() = default;
	ThreadGroup(const ThreadGroup &) = delete;
	ThreadGroup &operator=(const ThreadGroup &) = delete;
	template <class F>
	void spawn(F f) {
		th_.emplace_back([this, f]() mutable noexcept {
			try { f(); } catch (const std::exception &e) { note(e.what()); } catch (...) { note("unknown exception"); }
		});
	}
	// the body on the calling thread, under the same guard
	template <class F>
	void run_here(F &&f) noexcept {
		try { f(); } catch (const std::exception &e) { note(e.what()); } catch (...) { note("unknown exception"); }
	}
	void join() noexcept { for (auto &t : th_) if (t.joinable()) t.join(); }
	bool failed() const { return failed_; }
	const std::string &error() const { return error_; }
	size_t size() const { return th_.size(); }
	~ThreadGroup() { join(); }
};

// ---- interval_list<long> semantics (jsc/util/interval_list.hpp:396-422,462-503) -----------
struct IntervalList {
	std::vector<int64_t> s, e;
	void add(int64_t start, int64_t end);
	bool contains(int64_t start, int64_t end) const;
	size_t size() const { return s.size(); }
};

// ---- annotation ---------------------------------------------------------------------------
struct IsoRec {                    // one LH_GENE_TXT line (count/count.cpp:142-171)
	std::string name, chrom, strand;
	int64_t txStart = 0, txEnd = 0;
	uint64_t exonCount = 0;
	std::vector<int64_t> exonStarts, exonEnds;
};
struct Gene {
	std::string name;
	std::vector<const IsoRec *> isos;      // order of lines in the g2i file
};

// ---- compiled event ---------------------------------------------------------------------------
struct Event {
	std::string gname, chrom, strand;
	int chrom_id = -1, strand_id = -1;
	int K = 0, N = 0;
	std::vector<int64_t> seg_s, seg_e;             // atomic segments, ascending
	std::vector<std::string> iso_names;
	std::vector<uint64_t> iso_mask;                // bit n: isoform holds segment n (N <= 64)
	std::vector<uint64_t> iso_len;                 // total segment length per isoform
	int64_t gene_start = 0, gene_end = 0;          // first start / last end of the merged exon list
	std::vector<std::vector<uint64_t>> ars;        // [method][iso]
	// [method][iso * n_cls + (class - 1)]: how many of the isoform's accessible read starts give a read of that
	// compatibility class (fim.h's sum over generated reads, grouped; lsq_fim)
	std::vector<std::vector<uint32_t>> fim_starts;
	// name tie-break against "read-<n>" (count/count.cpp:71): 0 never less, 1 always less,
	// 2 compare the decimal digits of n with `tail`
	int tie_mode = 0;
	std::string tie_tail;
};

// One LDS image per bucket: a contiguous coordinate range of one chromosome whose event
// tables fit the kernel's LDS budget.  Layout of the image (all offsets in bytes from the
// image start, 16-byte aligned):  bins u16[n_bins] | event records 16 B | segments int2 |
// isoform masks u32 | (histogram u64[n_cls], not part of the image, zeroed by the kernel)
struct BucketDesc {
	uint32_t img_off;      // byte offset of the image in the images blob
	uint32_t img_bytes;    // bytes to stage
	uint32_t n_events;
	uint32_t n_bins;
	int32_t lo;            // coordinate of bin 0
	uint32_t shift;        // bin = (p - lo) >> shift
	uint32_t ev_off, seg_off, iso_off, hist_off;   // LDS byte offsets
	uint32_t n_cls;        // histogram slots
	uint32_t cls_base;     // first class slot of the bucket in device class order
	uint32_t ev_base;      // first event of the bucket in device event order
	int32_t chrom_id;
	uint32_t kind;         // 0: generic records (EventRec + segments + masks); 1: packed 48-byte FastRec; 2: no tables, evaluated on the host
	int32_t hi;            // largest span end in the bucket (reads that start right of it have no candidate)
};
static_assert(sizeof(BucketDesc) == 64, "BucketDesc is copied to the device verbatim");

struct EventRec {          // 16-byte LDS record
	int32_t gs, ge;
	uint16_t seg_off, iso_off, cls_off;   // element indices inside the bucket
	uint8_t nseg, K;
};
static_assert(sizeof(EventRec) == 16, "EventRec layout");

// Packed record of a "small" event (<= 4 segments, <= 4 isoforms, non-negative coordinates,
// span start == first segment start): three 16-byte words, read with three wide LDS loads.
//   w0: ge, meta, class table (64 bits: 4-bit compatibility class per 4-bit segment mask)
//   w1: seg0.start seg0.end seg1.start seg1.end     w2: seg2 / seg3 likewise
//   meta: bits 0-15 first class slot in the bucket; 16-18 "segment k+1 starts where k ends";
//         19 the next event (span-start order) starts inside this span; 24-26 segment count
// unused segments hold INT32_MAX (they end a walk like running off the segment list)
struct FastRec {
	int32_t ge;
	uint32_t meta, tbl_lo, tbl_hi;
	int32_t seg[8];
};
static_assert(sizeof(FastRec) == 48, "FastRec layout");
constexpr uint32_t FAST_ABUT_SHIFT = 16;
constexpr uint32_t FAST_FLAG_OVERLAPS_NEXT = 1u << 19;
constexpr uint32_t FAST_NSEG_SHIFT = 24;

// Shortcut table of a packed bucket.  A cell is a stretch of coordinates over which the set of segments (of any event
// of the bucket) that cover it does not change, with none, one or two such segments ("owners").  A read whose first base
// lies in the cell can only ever count for the owners' events: every other event whose span covers the base has no
// segment there, so the read's first block starts in none of its segments and nothing matches (common/read.h:204-274).
// One owner -- segment k of its event: a one-block read [a, b) that starts in the cell is settled by where it ends:
//   b <= e1              inside the segment: matched == its length, the class of {k}                          (slot 1)
//   e1 < b <= e2         e2 > e1: segment k+1 starts where k ends, the read runs into it: class of {k, k+1}   (slot 2)
//   b > e2               the general walk decides (a further abutting segment; the 98 % rule)
// Two owners (CELLX_BOTH): e1 = the nearer of the two segments' ends, e2 = the farther, slot 1 / slot 2 = the single-
// segment classes of the owner that ends first / of the other, CellX.ev = the owner that ends first.  A read that ends at
// or before e1 counts for both; one that ends in (e1, e2] counts for the second owner and goes to the general walk for the
// first alone; one beyond e2 goes to the walk for both.
// No owner: a stretch inside no segment; both slots empty, e1 = e2 = CELL_NO_END -- its reads count for nobody.
// The first base of an event's span is left out of every cell (the span-start tie rule decides there), but see
// CELL_K_START.  One record per cell in two arrays of 16-byte words: Cell and CellX.
struct Cell {
	int32_t lo, hi;        // the cell
	int32_t e1, e2;
};
static_assert(sizeof(Cell) == 16, "Cell layout");
struct CellX {
	uint32_t slots;        // low 16 bits: histogram slot 1; high 16 bits: slot 2; 0xFFFF = no compatible isoform / absent
	uint32_t info;         // one owner: event << 8 | segment << 2 | lo is the segment's start; CELL_INFO_SHARED with two owners; CELL_INFO_EMPTY with none
	uint32_t flags;        // CELLX_BOTH, CELLX_*_NO_ABUT; two owners: low 16 bits = event of the owner that ends last
	uint32_t ev;           // one owner: its event (index in the bucket); two owners: the one whose segment ends first
};
static_assert(sizeof(CellX) == 16, "CellX layout");
constexpr uint32_t CELL_NONE = 0xFFFFu;
constexpr int32_t CELL_NO_END = 0x7F000000;     // beyond every coordinate (< 2^30), and still so after the kernel has taken a bucket's base (> -2^22) off it
constexpr uint32_t CELLX_BOTH = 1u << 16;
// two owners (round 3): the owner whose segment ends first / last has NO segment that starts where that one ends -- a read
// that runs past that end matches the owner only up to it, and is valid for it only if that is more than 98 % of the read
// (count/count.cpp:441): almost never, and the loop can say so itself instead of parking the read for the walk.  The low 16
// bits of the flags word hold the event (index in the bucket) of the owner whose segment ends LAST (CellX.ev: the other).
constexpr uint32_t CELLX_NEAR_NO_ABUT = 1u << 17, CELLX_FAR_NO_ABUT = 1u << 18;
// The LDS histogram of a bucket is kept HIST_REPLICAS times (lane & (R-1) picks the copy; copies
// are (n_cls | 1) entries apart so that they start in different banks): neighbouring reads of the
// start-ordered pools land in the same class, and atomics on one LDS address run one after the other.
constexpr uint32_t HIST_REPLICAS = 4;
inline uint32_t hist_stride(uint32_t n_cls) { return n_cls | 1u; }
constexpr uint32_t CELL_INFO_SHARED = 0xFFFFFFFFu;
constexpr uint32_t CELL_INFO_EMPTY = 0xFFFFFFFEu;          // a stretch inside no segment
// segment number of a start cell (the first base of an event's span where that event's first segment is the only cover):
// lo = gene_start, hi = e1 = lo + 1, e2 = gene_end - 1, both slots CELL_NONE -- a one-block read from there that ends
// before gene_end is ordered before the event in the read index and counts for nobody; one that reaches gene_end goes
// to the general walk
constexpr uint32_t CELL_K_START = 63u;

// key of a junction group of the two-block pool (lsq_events::jg_keys): ascending in (cell, variant, start of block 2)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t jg_key(uint32_t cell, uint32_t variant, int32_t block2_start) { return ((uint64_t)cell << 33) | ((uint64_t)(variant & 1u) << 32) | (uint32_t)block2_start; }

struct TieRec {            // global memory, device event order; read only on start ties
	uint8_t strand_id;
	uint8_t tie_mode;
	uint8_t tail_len;      // capped at 21: a decimal read number has at most 20 digits
	char tail[21];
};
static_assert(sizeof(TieRec) == 24, "TieRec layout");

struct Dict {              // string interning (chromosomes, strands)
	std::vector<std::string> names;
	std::unordered_map<std::string, int> ids;
	std::mutex mu;
	int intern(const std::string &s);
	int find(const std::string &s) const;
};

} // namespace lsq

struct lsq_annotation {
	std::vector<std::unique_ptr<lsq::IsoRec>> recs;
	std::vector<lsq::Gene> selected;
	int64_t n_genes_loaded = 0;
};

struct lsq_events {
	std::vector<lsq::Event> ev;                    // output order
	int n_methods = 0;
	std::vector<std::string> read_types;
	std::vector<uint64_t> read_lengths;
	lsq::Dict chroms, strands;
	std::vector<lsq::IntervalList> covered;        // by chrom id (events' chromosomes)
	// ---- device plan
	uint64_t shard_first = 0, shard_count = UINT64_MAX;   // events (output order) this process works on
	uint32_t lds_budget = 0;
	std::vector<lsq::BucketDesc> buckets;          // sorted by (chrom_id, lo)
	std::vector<uint8_t> images;                   // all bucket images
	// Junction groups of the two-block pool: per bucket the sorted keys (cell << 32 | start of block 2) of the reads the
	// count kernel settles as junction reads -- block 1 starts in a one-owner cell, ends on the end of that owner's segment,
	// block 2 starts on the first base of a later segment of the same event.  The ingest lays the two-block reads out by
	// these groups (and one group per bucket for all others), padded to two records: a lane's two reads cross one junction.
	// (keys: jg_key(cell, variant, start of block 2); variant 1: block 1 runs on through the segment that abuts the owner's and
	// ends on THAT one's end)
	std::vector<uint64_t> jg_keys;
	std::vector<uint32_t> jg_base;                 // per bucket: first of its keys (n_buckets + 1)
	std::vector<int32_t> dev2out;                  // device event index -> output index
	std::vector<uint32_t> dev_cls_base;            // per device event
	std::vector<uint32_t> dev_iso_base;            // per device event
	std::vector<lsq::TieRec> ties;                 // device order
	std::vector<uint8_t> dev_K;                    // device order
	uint32_t n_cls_total = 0, n_iso_total = 0;
	uint32_t max_lds_bytes = 0;                    // image + histogram, max over buckets
	// bucket lookup: per chrom id, ascending cut coordinates and the bucket of each range
	// per chrom id: the merged spans of the planned events (clusters of transitively overlapping spans), ascending.  A read
	// is a candidate of an event only if its first base lies in the event's span (count/count.cpp:429-432,463), so a read
	// that starts outside every cluster is dropped at ingest -- with a shard (lsq_events_set_shard) those are the reads of
	// the other shards' events, which pass the load-time filter of the whole range but concern no event planned here
	std::vector<std::vector<int32_t>> clu_s, clu_e;
	std::vector<std::vector<int32_t>> cut_lo;      // cut_lo[chrom][i] = buckets[first+i].lo
	std::vector<int32_t> chrom_first_bucket;       // -1 when the chromosome has no bucket
	std::vector<uint64_t> class_off;               // output order, n_events+1
	std::vector<uint64_t> iso_off;                 // output order, n_events+1
};

struct lsq_reads {
	uint64_t n_reads = 0, n_blocks = 0;
	const uint64_t *blk_off = nullptr;
	const uint32_t *line_no = nullptr;
	const int32_t *blk_start = nullptr, *blk_end = nullptr;
	const uint16_t *blk_chrom = nullptr;
	const uint8_t *blk_strand = nullptr;
	// owned storage (empty when wrapping caller arrays)
	std::vector<uint64_t> o_blk_off;
	std::vector<uint32_t> o_line_no;
	std::vector<int32_t> o_start, o_end;
	std::vector<uint16_t> o_chrom;
	std::vector<uint8_t> o_strand;
	// reads that carry their own names (solve's UCSC_GFF / UCSC_BED / WORMBASE_GFF3 read formats): line_no
	// then indexes this table instead of naming the read "read-<line_no>"
	bool named = false;
	std::string name_blob;
	std::vector<uint64_t> name_off;                // n_reads + 1
	void adopt();
};

namespace lsq {

int plan_device(lsq_events &E);
int host_threads(int requested);

} // namespace lsq
