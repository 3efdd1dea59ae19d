// MRF_SINGLE parsed on the device (included by lsq_ingest.hip; not a public header).
//
// The text goes to HBM as it is and is read there twice: once to count the newlines of every 7 680-byte tile (16 bytes per
// lane, the exact zero-byte test on word ^ 0x0A0A0A0A), once to parse.  A workgroup of the parse owns the lines that END
// in its tile and sees the tile and the 512 bytes before it.  The newline ordinal of a line (tile base from a prefix sum
// over the tile counts + its place in the tile) is its line number: the header and "read-<n>" fall out as in the
// reference (count/count.cpp:283,286,293-295).
//
// Round 4: the parse IS the routing pass of the loader chain (lsq_ingest.hip): per line, every block goes through the
// containment filter and the merge as it is split off, and nothing but the routed read leaves the kernel -- no parsed
// array exists in HBM either.  Three kernels:
//   lsq_mrf_route_fast_kernel   every tile: delimiter tables from byte-parallel zero-byte tests, lines walk table entries,
//                               a lane a block does the coordinates (eight-digit sums), the chromosome and strand (64-bit
//                               keys), the filter; settles every line of a read's usual shape and lists the others
//   lsq_mrf_route_kernel        the byte-walking form: the tile in LDS, a lane a line runs the SHARED SPLITTER
//                               (lsq_mrf_line.hpp -- the code the host parser runs) over LDS bytes; takes the tiles the fast
//                               kernel hands on, or whole files where that one does not apply
//   lsq_mrf_route_lines_kernel  the listed lines (another shape than a read's; began ahead of their tile's window): a lane
//                               a line, the shared splitter over the bytes in HBM
// lsq_mrf_parse_device (tests, tools) still wants the parsed arrays: the byte-walking tile walk run twice, counting
// (lsq_mrf_count_kernel) and writing (lsq_mrf_write_kernel) around two prefix sums.
//   - the first line is the header (count.cpp:283); a last line without '\n' is never seen (:285)
//   - '#' lines and the literal "AlignmentBlocks" consume a line number only (:288)
//   - a field that fails the cast stops the run: the FIRST such line in file order is reported
//   - chromosome names resolve against the events' chromosomes (64-bit keys of names up to seven bytes in LDS; a hash table
//     with byte-wise verification for the byte-walking kernels); strand strings against a 256-slot table seeded with the
//     strands already known, grown with atomicCAS (strings of at most 7 bytes; longer ones give LSQ_E_UNSUPPORTED -- use
//     lsq_mrf_parse)
#pragma once

#ifndef LSQ_MRF_TILE
#define LSQ_MRF_TILE 7680
#endif
constexpr unsigned MRF_TILE = LSQ_MRF_TILE;         // text bytes per workgroup: with the 512 bytes ahead, a window of 8 KiB (7 680) or 4 KiB (3 584)
static_assert(MRF_TILE == 7680 || MRF_TILE == 3584, "the fast kernel's window is 256 lanes x 32 or x 16 bytes");
constexpr unsigned MRF_TILE_Q = (MRF_TILE + 4095) / 4096;      // 16-byte words a lane of 256 takes
constexpr unsigned MRF_LB = 512;                    // bytes ahead of the tile that are staged with it
constexpr unsigned MRF_NLCAP = 1024;                // newline positions held at a time (a tile of shorter lines takes several rounds)
constexpr unsigned long long MRF_NO_ERR = ~0ull;
constexpr unsigned long long STRAND_EMPTY = ~0ull;
constexpr unsigned long long STRAND_UNMATCHABLE = ~0ull - 1;
constexpr unsigned MRF_NOCHROM = 0xFFFFu;
constexpr unsigned MRF_DICT_LDS_SLOTS = 256;        // chromosome hash tables up to this size are staged in LDS (<= 64 chromosomes)
constexpr unsigned MRF_DICT_LDS_NAMES = 1024;

typedef const __attribute__((address_space(3))) char *mrf_lds_cptr;
typedef lsq::MrfViewT<mrf_lds_cptr, unsigned> MrfLdsView;

struct MrfDict {
	const unsigned *chrom_hash;             // open addressing, 0 = empty; 32-bit FNV-1a of the name
	const unsigned *chrom_id;
	const unsigned *name_off;               // per chromosome id, into names
	const char *names;
	unsigned mask, n_chrom, names_bytes;
	unsigned long long *strand_tab;         // 256 slots
};

struct MrfTileLds {
	__align__(16) unsigned char text[MRF_LB + MRF_TILE + 16];
	unsigned short nlpos[MRF_NLCAP];
	unsigned scan4[4];
	unsigned carry;                         // last newline of the previous round
	long long first_start;                  // first byte of the first line that ends in the tile
	// the dictionaries, when they are small
	unsigned long long strand[256];
	unsigned d_hash[MRF_DICT_LDS_SLOTS], d_id[MRF_DICT_LDS_SLOTS], d_off[MRF_DICT_LDS_SLOTS / 4 + 1];
	char d_names[MRF_DICT_LDS_NAMES];
	__align__(16) RouteChrom chrom[ROUTE_CHROM_LDS];      // the routing pass's chromosome records (lsq_mrf_route_kernel)
};

__device__ inline unsigned mrf_wave_incl_scan(unsigned v) {
	const unsigned lane = threadIdx.x & 63u;
	for (unsigned d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(v, d); if (lane >= d) v += t; }
	return v;
}
// exclusive prefix over the 256 lanes of the workgroup; `total` = sum over the workgroup
__device__ inline unsigned mrf_block_excl_scan(unsigned v, unsigned *lds4, unsigned &total) {
	const unsigned inc = mrf_wave_incl_scan(v);
	const unsigned w = threadIdx.x >> 6;
	if ((threadIdx.x & 63u) == 63u) lds4[w] = inc;
	__syncthreads();
	unsigned base = 0; total = 0;
	for (unsigned q = 0; q < 4; ++q) { const unsigned t = lds4[q]; base += q < w ? t : 0u; total += t; }
	__syncthreads();
	return base + inc - v;
}

// bit j set iff byte j of the 16 bytes is '\n'; only the first `valid` bytes count
__device__ inline unsigned mrf_newline_bits16(const uint4 v, unsigned valid) {
	const unsigned w[4] = {v.x, v.y, v.z, v.w};
	unsigned bits = 0;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const unsigned x = w[q] ^ 0x0A0A0A0Au;
		const unsigned z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;   // 0x80 in every zero byte
		bits |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * q);
	}
	return valid >= 16u ? bits : (bits & ((1u << valid) - 1u));
}
// the 16 bytes at `at` of a text of `len` bytes (the buffer holds 16 bytes of slack behind the text) and how many of them are text
__device__ inline uint4 mrf_load16(const unsigned char *text, unsigned long long len, unsigned long long at, unsigned &valid) {
	if (at >= len) { valid = 0; return make_uint4(0, 0, 0, 0); }
	valid = (unsigned)min(16ull, len - at);
	return *reinterpret_cast<const uint4 *>(text + at);
}

// newlines per tile
__global__ void __launch_bounds__(256) lsq_mrf_newline_count_kernel(const unsigned char *text, unsigned long long len, unsigned *tile_cnt) {
	__shared__ unsigned lds4[4];
	const unsigned long long t0 = (unsigned long long)blockIdx.x * MRF_TILE;
	unsigned n = 0;
#pragma unroll
	for (unsigned q = 0; q < MRF_TILE_Q; ++q) {
		unsigned valid = 0;
		const unsigned off = q * 4096u + threadIdx.x * 16u;
		const uint4 v = off < MRF_TILE ? mrf_load16(text, len, t0 + off, valid) : make_uint4(0, 0, 0, 0);
		n += (unsigned)__popc(mrf_newline_bits16(v, valid));
	}
	unsigned total;
	(void)mrf_block_excl_scan(n, lds4, total);
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// The lines that end in this workgroup's tile: fn(i, view) is called once per data line -- i its 0-based index among the
// data lines (with a header: the lines after the first), view the line's bytes without the newline -- by the lane that
// owns it.  view is an LDS view, or a plain one for a line that began more than MRF_LB bytes ahead of the tile.
// A line that began more than MRF_LB bytes ahead of its tile: its data line index, first byte and length.  At most one per tile.
struct MrfLongLine { unsigned long long i, start, n; };

template <bool DEFER, class Fn>
__device__ inline void mrf_tile_lines(MrfTileLds &S, const unsigned tile, const unsigned char *text, const unsigned long long len, const unsigned long long *tile_base,
                                      const unsigned has_header, MrfLongLine *long_lines, unsigned *n_long, const unsigned long_cap, unsigned *long_over, Fn &&fn) {
	const unsigned tid = threadIdx.x;
	const unsigned long long t0 = (unsigned long long)tile * MRF_TILE;
	unsigned bits[MRF_TILE_Q];
#pragma unroll
	for (unsigned q = 0; q < MRF_TILE_Q; ++q) {
		unsigned valid = 0;
		const unsigned off = q * 4096u + tid * 16u;
		const uint4 v = off < MRF_TILE ? mrf_load16(text, len, t0 + off, valid) : make_uint4(0, 0, 0, 0);
		if (off < MRF_TILE) *reinterpret_cast<uint4 *>(&S.text[MRF_LB + off]) = v;
		bits[q] = mrf_newline_bits16(v, valid);
	}
	if (tid < MRF_LB / 16u && t0 >= MRF_LB)
		*reinterpret_cast<uint4 *>(&S.text[tid * 16u]) = *reinterpret_cast<const uint4 *>(text + (t0 - MRF_LB) + tid * 16ull);
	// ordinals of the newlines: the lanes' first words cover bytes 0..4095 of the tile, their second words the rest
	unsigned ord[MRF_TILE_Q], nt = 0;
#pragma unroll
	for (unsigned q = 0; q < MRF_TILE_Q; ++q) {
		unsigned total;
		ord[q] = nt + mrf_block_excl_scan((unsigned)__popc(bits[q]), S.scan4, total);
		nt += total;
	}
	if (nt == 0) return;                      // (uniform: every lane holds the same total)
	// the first line that ends here began after the last newline ahead of the tile
	if (tid < 64u) {
		long long found = -1;
		if (t0 > 0) {
			for (unsigned long long k = 0;; ++k) {
				const long long pos = (long long)t0 - 1 - (long long)(k * 64ull + tid);
				const bool hit = pos >= 0 && text[pos] == '\n';
				const unsigned long long m = __ballot(hit);
				if (m) { found = (long long)t0 - 1 - (long long)(k * 64ull + (unsigned)(__ffsll((long long)m) - 1)); break; }
				if ((long long)t0 - 1 - (long long)(k * 64ull + 63ull) <= 0) break;
			}
		}
		if (tid == 0) S.first_start = found + 1;
	}
	const unsigned long long g0 = tile_base[tile];
	const mrf_lds_cptr lds_text = (mrf_lds_cptr)(const char *)S.text;
	for (unsigned rb = 0; rb < nt; rb += MRF_NLCAP) {
#pragma unroll
		for (unsigned q = 0; q < MRF_TILE_Q; ++q) {
			unsigned b = bits[q], o = ord[q];
			while (b) {
				const unsigned j = (unsigned)__ffs((int)b) - 1u; b &= b - 1u;
				if (o >= rb && o < rb + MRF_NLCAP) S.nlpos[o - rb] = (unsigned short)(q * 4096u + tid * 16u + j);
				++o;
			}
		}
		__syncthreads();
		const unsigned r_end = min(nt, rb + MRF_NLCAP);
		for (unsigned j = rb + tid; j < r_end; j += 256u) {
			const unsigned long long g = g0 + j;                 // the newline's ordinal in the text = the 0-based number of the line it ends
			if (has_header && g == 0) continue;
			const int end_rel = (int)S.nlpos[j - rb];
			long long start_rel;
			if (j == 0) start_rel = S.first_start - (long long)t0;
			else if (j > rb) start_rel = (long long)S.nlpos[j - 1 - rb] + 1;
			else start_rel = (long long)S.carry + 1;
			const unsigned long long i = g - has_header;
			if (start_rel >= -(long long)MRF_LB) fn(i, MrfLdsView{lds_text + (MRF_LB + (int)start_rel), (unsigned)(end_rel - (int)start_rel)});
			else if constexpr (DEFER) {
				const unsigned at = atomicAdd(n_long, 1u);
				if (at < long_cap) long_lines[at] = MrfLongLine{i, (unsigned long long)((long long)t0 + start_rel), (unsigned long long)((long long)end_rel - start_rel)}; else *long_over = 1u;
			}
			else fn(i, lsq::MrfView{reinterpret_cast<const char *>(text) + (t0 + start_rel), (size_t)((long long)end_rel - start_rel)});
		}
		__syncthreads();
		if (tid == 0) S.carry = S.nlpos[MRF_NLCAP - 1];
		__syncthreads();
	}
}

// the dictionaries into LDS when they fit; returns the view the lookups use
__device__ inline MrfDict mrf_stage_dict(MrfTileLds &S, const MrfDict &G) {
	S.strand[threadIdx.x & 255u] = __hip_atomic_load(&G.strand_tab[threadIdx.x & 255u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	MrfDict D = G;
	if (G.mask < MRF_DICT_LDS_SLOTS && G.names_bytes <= MRF_DICT_LDS_NAMES && G.n_chrom <= MRF_DICT_LDS_SLOTS / 4) {
		for (unsigned i = threadIdx.x; i <= G.mask; i += 256u) { S.d_hash[i] = G.chrom_hash[i]; S.d_id[i] = G.chrom_id[i]; }
		for (unsigned i = threadIdx.x; i <= G.n_chrom; i += 256u) S.d_off[i] = G.name_off[i];
		for (unsigned i = threadIdx.x; i < G.names_bytes; i += 256u) S.d_names[i] = G.names[i];
		D.chrom_hash = S.d_hash; D.chrom_id = S.d_id; D.name_off = S.d_off; D.names = S.d_names;
	}
	__syncthreads();
	return D;
}

LSQ_HD inline unsigned mrf_fnv32(const char *p, size_t n) {
	unsigned h = 2166136261u;
	for (size_t j = 0; j < n; ++j) { h ^= (unsigned char)p[j]; h *= 16777619u; }
	return h ? h : 1u;
}

template <class V>
__device__ inline unsigned mrf_chrom_lookup(const MrfDict &D, V s) {
	typedef typename V::index_type Idx;
	unsigned h = 2166136261u;
	for (Idx j = 0; j < s.n; ++j) { h ^= (unsigned char)s.p[j]; h *= 16777619u; }
	if (h == 0) h = 1;
	for (unsigned i = h & D.mask;; i = (i + 1u) & D.mask) {
		const unsigned t = D.chrom_hash[i];
		if (t == 0) return MRF_NOCHROM;
		if (t != h) continue;
		const unsigned id = D.chrom_id[i];
		const unsigned a = D.name_off[id], b = D.name_off[id + 1];
		if ((Idx)(b - a) != s.n) continue;
		bool same = true;
		for (Idx j = 0; j < s.n; ++j) same = same && D.names[a + j] == s.p[j];
		if (same) return id;
	}
}

// strand strings of <= 7 bytes as one order-preserving 64-bit key (bytes big-endian, length last)
LSQ_HD inline unsigned long long mrf_strand_key(const char *p, size_t n) {
	unsigned long long k = (unsigned long long)n;
	for (size_t j = 0; j < n; ++j) k |= (unsigned long long)(unsigned char)p[j] << (56 - 8 * j);
	return k;
}

// the slot of a strand string: the workgroup's LDS copy of the table first (the strands every file has are there from the
// start), the table itself -- and a place in it for a new string -- otherwise
template <class V>
__device__ inline unsigned mrf_strand_slot(const unsigned long long *lds_tab, unsigned long long *tab, V s, unsigned long long *err) {
	if (s.n > 7) { atomicMax(&err[1], 1ull); return 0; }
	unsigned long long key = (unsigned long long)s.n;
	for (unsigned j = 0; j < (unsigned)s.n; ++j) key |= (unsigned long long)(unsigned char)s.p[j] << (56 - 8 * j);
	if (lds_tab) for (unsigned i = 0; i < 256; ++i) {
		const unsigned long long cur = lds_tab[i];
		if (cur == key) return i;
		if (cur == STRAND_EMPTY) break;
	}
	for (unsigned i = 0; i < 256; ++i) {
		const unsigned long long cur = __hip_atomic_load(&tab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == key) return i;
		if (cur == STRAND_EMPTY) {
			const unsigned long long prev = atomicCAS(&tab[i], STRAND_EMPTY, key);
			if (prev == STRAND_EMPTY || prev == key) return i;
		}
	}
	atomicMax(&err[2], 1ull);
	return 0;
}

struct MrfText {
	const unsigned char *text;
	unsigned long long len;
	const unsigned long long *tile_base;
	unsigned has_header;
	unsigned long long first_line;            // the number of data line 0 in the whole file (read name "read-<L>")
	unsigned long long n_lines;
};

// ---- the parse that feeds the load-time filter (the product path): per line, every block through the covered regions and
// the merge as it is split off; the routed read to key[i] / rec[i], i the data line's index.  Three kernels:
//   lsq_mrf_route_fast_kernel   every tile; settles the lines of the usual shape (below) and lists the others
//   lsq_mrf_route_kernel        the tiles the fast kernel could not table (more delimiters than its LDS tables hold), or every
//                               tile when the fast kernel does not apply: the shared splitter over LDS bytes, a lane a line
//   lsq_mrf_route_lines_kernel  the listed lines, a lane a line, the shared splitter over the bytes in HBM
template <class V>
__device__ inline void mrf_route_line(const MrfText &X, const MrfDict &D, const unsigned long long *lds_strand, const RouteTables &T, const RouteChrom *chroms, const RouteOut &O,
                                      unsigned long long *err, const unsigned long long i, const V line) {
	const long long LIM = 1ll << 30;
	if (lsq::mrf_line_is_skipped(line)) { O.key[i] = ROUTE_KEY_DROPPED; return; }
	ReadAcc A;
	ReadBig B;
	A.init();
	LocProbe P;
	P.chrom = -1; P.bin = 0;
	const bool ok = lsq::mrf_split_line(line, [&](const V chr, const V strand, const int64_t start, const int64_t end) {
		const unsigned cid = mrf_chrom_lookup(D, chr);
		const long long s0 = start - 1, e0 = end;
		if (cid >= T.n_chrom || s0 <= -LIM || e0 >= LIM || s0 >= LIM || e0 <= -LIM) return;
		if (!route_covered(T, chroms[cid], (int)cid, (int)s0, (int)e0, P)) return;
		A.add(B, cid, mrf_strand_slot(lds_strand, D.strand_tab, strand, err), (int)s0, (int)e0);
	});
	if (!ok) { atomicMin(&err[0], X.first_line + i); O.key[i] = ROUTE_KEY_DROPPED; return; }
	A.finish(B, T, chroms, P, O, (unsigned)i);
}

// lists of work the fast kernel hands on: counts[0] tiles, counts[1] lines, counts[2] set when the line list ran over
struct MrfHandOff {
	unsigned *counts;
	unsigned *tiles; unsigned tile_cap;
	MrfLongLine *lines; unsigned line_cap;
};

__global__ void __launch_bounds__(256) lsq_mrf_route_kernel(MrfText X, MrfDict G, RouteTables T, RouteOut O, unsigned long long *err, MrfHandOff H, unsigned n_tiles, unsigned listed) {
	__shared__ MrfTileLds S;
	const MrfDict D = mrf_stage_dict(S, G);
	const RouteChrom *chroms = route_stage_chroms(T, S.chrom);
	const unsigned n = listed ? min(H.counts[0], H.tile_cap) : n_tiles;
	for (unsigned t = blockIdx.x; t < n; t += gridDim.x) {
		const unsigned tile = listed ? H.tiles[t] : t;
		mrf_tile_lines<true>(S, tile, X.text, X.len, X.tile_base, X.has_header, H.lines, H.counts + 1, H.line_cap, H.counts + 2, [&](const unsigned long long i, const MrfLdsView line) {
			mrf_route_line(X, D, S.strand, T, chroms, O, err, i, line);
		});
		__syncthreads();
	}
}
// listed lines, one lane each, straight from HBM.  A line listed without its start (n = ~0: it began ahead of a tile's
// window) is walked back to the newline before it first.
__global__ void __launch_bounds__(256) lsq_mrf_route_lines_kernel(MrfText X, MrfDict G, RouteTables T, RouteOut O, unsigned long long *err, MrfHandOff H) {
	const unsigned n = min(H.counts[1], H.line_cap);
	for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
		MrfLongLine L = H.lines[t];
		if (L.n == ~0ull) {
			// L.start: the line's newline
			unsigned long long a = L.start;
			while (a > 0 && X.text[a - 1] != '\n') --a;
			L.n = L.start - a; L.start = a;
		}
		mrf_route_line(X, G, nullptr, T, T.chrom, O, err, L.i, lsq::MrfView{reinterpret_cast<const char *>(X.text) + L.start, (size_t)L.n});
	}
}

// ---- the fast kernel.  A workgroup takes a window of 8 KiB of text -- a tile of 7 680 bytes and the 512 ahead of it -- 32
// bytes a lane, in registers: three exact zero-byte tests (word ^ 0x0A.., ^ 0x3A.., ^ 0x2C..) give every lane the newlines,
// colons and commas among its bytes as bit masks; one prefix sum over the lanes' counts numbers them; the lanes write the
// window's delimiters, in text order, into an LDS table (position | kind) and, per newline, the index of its entry.  Then one
// lane per line that ends in the tile walks its entries instead of its bytes.  A block whose next four delimiters are colons
// is exactly what the reference's find / substr walk (count/count.cpp:297-326) takes it for: chr = [block start, colon 1),
// strand, start, end = the stretches between the colons, the next block behind the first comma after colon 4 (colons in
// between -- the query fields -- are passed over).  The two coordinates are at most nine decimal digits (what
// lexical_cast<long> takes without a sign, and no overflow possible): eight bytes that END at the delimiter, bytes ahead of
// the field replaced by '0', summed pairwise (1 x 10 + 1, 2 x 100 + 2, 4 x 10000 + 4 digits) -- and a ninth digit.  The
// chromosome (at most seven bytes) is looked up by its bytes as one 64-bit key in an LDS table, the strand likewise.
// Whatever does not fit this shape -- fewer than four colons before the next comma or the line's end, a sign or a stray
// byte in a coordinate, ten digits, a chromosome or strand of eight bytes or more, a line that starts ahead of the window
// -- is not decided here: the line goes on a list and lsq_mrf_route_lines_kernel runs the shared splitter on it.
// Waves a SIMD the compiler is asked to leave room for.  Left alone it takes 156 registers (three waves) and the kernel, a chain of
// LDS and L2 round trips per line, waits: 10.1 ms per C3 file; held to 128 (four waves, with spills) 13.5; asked for five or more
// it settles at 75 registers -- six waves, what the 23 KB of LDS tables allow -- with the per-line state it cannot keep in registers
// in the private segment (L1-resident): 6.4 ms (same box, tools/ingest_bench.py).
#ifndef LSQ_FAST_WAVES
#define LSQ_FAST_WAVES 6
#endif
#if LSQ_FAST_WAVES
#define LSQ_FAST_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(LSQ_FAST_WAVES)))
#else
#define LSQ_FAST_WAVES_ATTR
#endif
constexpr unsigned FP_WIN = MRF_LB + MRF_TILE;          // 8 192
constexpr unsigned FP_PAD = 16;                          // bytes of LDS ahead of the window (a coordinate's eight bytes may begin there)
constexpr unsigned FP_LANE = FP_WIN / 256;               // window bytes a lane takes: 32 or 16
constexpr unsigned FP_DCAP = FP_WIN * 5 / 16, FP_NCAP = FP_WIN / 8;       // delimiters / newlines a window may hold (8 KiB of reads: ~1 500 / ~220)
constexpr unsigned FP_BCAP = FP_WIN >= 8192 ? 384 : 256;                        // blocks of one round of 256 lines (reads: ~310); a line whose blocks find no room goes to the list
constexpr unsigned FP_STRANDS = 32;                      // strand keys kept in LDS (the table's first slots: the strands every file has)
constexpr unsigned FP_KIND_SHIFT = 13;                   // entry: position in the window | kind << 13 (0 colon, 1 comma, 2 newline)
constexpr unsigned FP_DICT = 256;

struct MrfFastDict {
	const unsigned long long *ckey;       // FP_DICT slots: the chromosome's bytes as a key (mrf_strand_key), 0 = empty
	const unsigned short *cid;
	unsigned usable;                       // every chromosome name has a slot (names of eight bytes and more have none: their lines go to the list)
};
__host__ __device__ inline unsigned mrf_key_slot(unsigned long long k) {
	unsigned x = (unsigned)k ^ (unsigned)(k >> 32);
	x ^= x >> 16; x ^= x >> 8;
	return x & (FP_DICT - 1u);
}

struct MrfFastLds {
	__align__(16) unsigned char text[FP_PAD + FP_WIN + 16];
	__align__(8) unsigned short delim[FP_DCAP + 8];
	unsigned short nl_dord[FP_NCAP];
	unsigned scan4[4];
	unsigned nl_ahead;                      // newlines in the 512 bytes ahead of the tile
	unsigned n_blk;                         // blocks listed in this round
	__align__(16) uint4 blk[FP_BCAP];       // a block: its five field bounds on the way in, what became of it on the way out
	__align__(8) unsigned long long strand[FP_STRANDS];
	unsigned long long ckey[FP_DICT];
	unsigned short cid[FP_DICT];
	__align__(16) RouteChrom chrom[ROUTE_CHROM_LDS];
};

// 0x80 in every byte of w that equals c
__device__ inline unsigned fp_eq_bytes(const unsigned w, const unsigned c4) {
	const unsigned x = w ^ c4;
	return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
__device__ inline unsigned fp_pack4(const unsigned z) { return ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u); }

// the decimal number in the eight bytes of `w` (first byte lowest), the last `len` of them digits (1 <= len <= 8): bytes ahead of the
// field count as '0'.  false when a byte of the field is no digit.
__device__ inline bool fp_number8(const unsigned long long w, const unsigned len, unsigned &out) {
	unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);                     // lo: the four bytes ahead, hi: the last four
	const unsigned keep_hi = len >= 4u ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8u * (4u - len));
	const unsigned keep_lo = len >= 8u ? 0xFFFFFFFFu : (len > 4u ? 0xFFFFFFFFu << (8u * (8u - len)) : 0u);
	hi = (hi & keep_hi) | (0x30303030u & ~keep_hi);
	lo = (lo & keep_lo) | (0x30303030u & ~keep_lo);
	// every byte of the eight in '0'..'9': on the low seven bits of each byte, + 0x46 must not reach the top bit (<= '9') and + 0x50 must (>= '0')
	const unsigned lo7 = lo & 0x7F7F7F7Fu, hi7 = hi & 0x7F7F7F7Fu;
	const bool digits = ((lo | hi) & 0x80808080u) == 0u && (((lo7 + 0x46464646u) | (hi7 + 0x46464646u)) & 0x80808080u) == 0u &&
	                    (((lo7 + 0x50505050u) & (hi7 + 0x50505050u)) & 0x80808080u) == 0x80808080u;
	unsigned a = lo - 0x30303030u, b = hi - 0x30303030u;       // (no borrow between bytes when they are digits; otherwise the result is not used)
	// first byte = most significant digit: pairs, then fours
	a = (a * 10u + (a >> 8)) & 0x00FF00FFu; b = (b * 10u + (b >> 8)) & 0x00FF00FFu;
	a = (__umul24(a & 0xFFFFu, 100u) + (a >> 16)); b = (__umul24(b & 0xFFFFu, 100u) + (b >> 16));
	out = __umul24(a, 10000u) + b;
	return digits;
}

__global__ void __launch_bounds__(256) LSQ_FAST_WAVES_ATTR lsq_mrf_route_fast_kernel(MrfText X, MrfDict G, MrfFastDict FD, RouteTables T, RouteOut O, unsigned long long *err, MrfHandOff H, unsigned n_tiles) {
	__shared__ MrfFastLds S;
	const unsigned tid = threadIdx.x;
	// ---- the dictionaries
	if (tid < FP_STRANDS) S.strand[tid] = __hip_atomic_load(&G.strand_tab[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	S.ckey[tid] = FD.ckey[tid]; S.cid[tid] = FD.cid[tid];
	for (unsigned q = tid; q < 2u * T.n_chrom; q += 256u) reinterpret_cast<uint4 *>(S.chrom)[q] = reinterpret_cast<const uint4 *>(T.chrom)[q];
	const mrf_lds_cptr text = (mrf_lds_cptr)(const char *)S.text + FP_PAD;      // window byte 0
	const unsigned POS = (1u << FP_KIND_SHIFT) - 1u;
	auto defer = [&](const unsigned long long i, const unsigned long long start, const unsigned long long n) {
		const unsigned at = atomicAdd(&H.counts[1], 1u);
		if (at < H.line_cap) H.lines[at] = MrfLongLine{i, start, n}; else H.counts[2] = 1u;
	};
	// a workgroup stays for many tiles (the dictionaries above are staged once: 4 KB per workgroup beside 8 KB of text per tile otherwise)
	for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
	const unsigned long long t0 = (unsigned long long)tile * MRF_TILE;
	__syncthreads();                          // (the tile before is done with the tables below)
	// ---- the window: FP_LANE bytes a lane, in text order (window byte FP_LANE x lane); bytes ahead of the text or behind it count as none
	unsigned w[FP_LANE / 4];
	unsigned valid = 0;                       // bytes of the lane's that are text
	{
		const long long at = (long long)t0 - (long long)MRF_LB + (long long)FP_LANE * tid;
		uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
		if (at >= 0 && (unsigned long long)at < X.len) {
			v0 = *reinterpret_cast<const uint4 *>(X.text + at);
			if (FP_LANE > 16 && (unsigned long long)at + 16u < X.len) v1 = *reinterpret_cast<const uint4 *>(X.text + at + 16);
			valid = (unsigned)min((unsigned long long)FP_LANE, X.len - (unsigned long long)at);
		}
		*reinterpret_cast<uint4 *>(&S.text[FP_PAD + FP_LANE * tid]) = v0;
		w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
		if constexpr (FP_LANE > 16) {
			*reinterpret_cast<uint4 *>(&S.text[FP_PAD + FP_LANE * tid + 16u]) = v1;
			w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
		}
	}
	unsigned m_nl = 0, m_colon = 0, m_comma = 0;
#pragma unroll
	for (int q = 0; q < (int)(FP_LANE / 4); ++q) {
		m_nl |= fp_pack4(fp_eq_bytes(w[q], 0x0A0A0A0Au)) << (4 * q);
		m_colon |= fp_pack4(fp_eq_bytes(w[q], 0x3A3A3A3Au)) << (4 * q);
		m_comma |= fp_pack4(fp_eq_bytes(w[q], 0x2C2C2C2Cu)) << (4 * q);
	}
	const unsigned live = valid >= 32u ? 0xFFFFFFFFu : ((1u << valid) - 1u);            // (valid <= FP_LANE)
	m_nl &= live; m_colon &= live; m_comma &= live;
	const unsigned m_any = m_nl | m_colon | m_comma;
	unsigned total;
	const unsigned ex = mrf_block_excl_scan(((unsigned)__popc(m_any) << 16) | (unsigned)__popc(m_nl), S.scan4, total);
	const unsigned n_delim = total >> 16, n_nl = total & 0xFFFFu;
	if (tid == MRF_LB / FP_LANE) S.nl_ahead = ex & 0xFFFFu;          // (the first lane of the tile proper: what lies ahead of it is the 512 bytes)
	if (n_delim > FP_DCAP || n_nl > FP_NCAP) {
		// more delimiters than the tables hold: the tile goes to the kernel that walks bytes
		if (tid == 0) { const unsigned at = atomicAdd(&H.counts[0], 1u); if (at < H.tile_cap) H.tiles[at] = tile; }
		continue;
	}
	{
		// (one loop over the lane's delimiters of all kinds: a loop per dword, without the packing, ran as long as the busiest
		// lane of every dword in turn -- measured 6.1 -> 6.9 ms)
		unsigned b = m_any, od = ex >> 16, on = ex & 0xFFFFu;
		while (b) {
			const unsigned j = (unsigned)__ffs((int)b) - 1u; b &= b - 1u;
			const unsigned kind = ((m_nl >> j) & 1u) * 2u + ((m_comma >> j) & 1u);
			S.delim[od] = (unsigned short)((FP_LANE * tid + j) | (kind << FP_KIND_SHIFT));
			if (kind == 2u) S.nl_dord[on++] = (unsigned short)od;
			++od;
		}
	}
	__syncthreads();
	const unsigned nl_ahead = S.nl_ahead;
	if (n_nl == nl_ahead) continue;               // no line ends in the tile
	const unsigned long long g0 = X.tile_base[tile];
	// Rounds of 256 lines, three passes each.  (1) A lane a LINE walks the line's entries: every block must show four colons (or
	// three and the line's end) -- otherwise the line goes to the list -- and is written to S.blk with its five field bounds.
	// (2) A lane a BLOCK: the two coordinates, the chromosome, the containment filter, the strand -- the expensive part, on every
	// lane of the wave whatever the lines' block counts are (a lane a line ran this loop twice for every wave that held one
	// two-block read: a quarter of the lanes busy in the second turn).  (3) A lane a line again: its blocks' outcomes in order
	// into the merge, then the bucket of the first merged base, key and blocks out.
	for (unsigned m0 = nl_ahead; m0 < n_nl; m0 += 256u) {
		if (tid == 0) S.n_blk = 0;
		__syncthreads();
		const unsigned m = m0 + tid;
		// ---- (1)
		bool live = m < n_nl, odd = false, dropped = false;
		unsigned long long i = 0;
		unsigned dS = 0, start = 0, eol = 0, nb = 0, first = 0;
		const unsigned long long w0 = t0 - MRF_LB;                 // (window byte 0 in the text; wraps below zero for tile 0, whose window bytes < 512 hold nothing)
		if (live) {
			const unsigned long long g = g0 + (m - nl_ahead);        // the newline's ordinal in the text = the 0-based number of the line it ends
			if (X.has_header && g == 0) live = false;
			i = g - X.has_header;
		}
		if (live) {
			const unsigned dE = S.nl_dord[m];
			eol = S.delim[dE] & POS;
			if (m == 0) {
				if (t0 != 0) { defer(i, w0 + eol, ~0ull); live = false; }  // began ahead of the window
				else { dS = 0; start = MRF_LB; }
			} else { const unsigned dp = S.nl_dord[m - 1]; dS = dp + 1u; start = (S.delim[dp] & POS) + 1u; }
		}
		if (live) {
			if (eol > start && text[start] == '#') dropped = true;       // a comment line takes a line number only (count.cpp:288)
			else {
				// the walk: the line's blocks counted, the first two kept in registers (a read has one or two; a line of more is walked once
				// more to list the rest -- a second walk for every line cost 0.5 ms of the 6.4 on C3)
				uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0;
				unsigned d = dS, cpos = start;
				for (;;) {
					unsigned long long e4;
					__builtin_memcpy(&e4, &S.delim[d], 8);
					// four colons -- or three and the line's end: a block without query fields, whose end field runs to the end of the line
					// (count.cpp:313-316 with find() == npos) and after which nothing follows
					const bool to_eol = (e4 & 0xE000600060006000ull) == 0x4000000000000000ull;
					if ((e4 & 0x6000600060006000ull) != 0ull && !to_eol) { odd = true; break; }
					const uint4 bd = make_uint4(cpos | (((unsigned)e4 & POS) << 16), ((unsigned)(e4 >> 16) & POS) | (((unsigned)(e4 >> 32) & POS) << 16), (unsigned)(e4 >> 48) & POS, 0u);
					if (nb == 0u) b0 = bd; else if (nb == 1u) b1 = bd;
					++nb;
					if (to_eol) break;
					unsigned dn = d + 4u, kind;
					while ((kind = S.delim[dn] >> FP_KIND_SHIFT) == 0u) ++dn;      // the next block, if any: behind the first comma after colon 4
					if (kind != 1u) break;
					cpos = (S.delim[dn] & POS) + 1u; d = dn + 1u;
				}
				if (!odd) {
					first = atomicAdd(&S.n_blk, nb);
					if (first + nb > FP_BCAP) {                     // no room in this round's list: the shared splitter takes the line
						odd = true;
						for (unsigned q = first; q < FP_BCAP; ++q) S.blk[q] = make_uint4(0u, 0u, 0u, 3u);      // (its places below the list's end hold nothing)
					}
				}
				if (!odd) {
					S.blk[first] = b0;
					if (nb > 1u) S.blk[first + 1u] = b1;
					if (nb > 2u) {
						unsigned q = first;
						cpos = start; d = dS;
						for (;;) {
							unsigned long long e4;
							__builtin_memcpy(&e4, &S.delim[d], 8);
							const bool to_eol = (e4 & 0xE000600060006000ull) == 0x4000000000000000ull;
							if (q >= first + 2u) S.blk[q] = make_uint4(cpos | (((unsigned)e4 & POS) << 16), ((unsigned)(e4 >> 16) & POS) | (((unsigned)(e4 >> 32) & POS) << 16), (unsigned)(e4 >> 48) & POS, 0u);
							++q;
							if (to_eol) break;
							unsigned dn = d + 4u, kind;
							while ((kind = S.delim[dn] >> FP_KIND_SHIFT) == 0u) ++dn;
							if (kind != 1u) break;
							cpos = (S.delim[dn] & POS) + 1u; d = dn + 1u;
						}
					}
				}
			}
			if (odd) { defer(i, w0 + start, eol - start); live = false; }
		}
		__syncthreads();
		// ---- (2)
		const unsigned n_blk = min(S.n_blk, FP_BCAP);          // (lines beyond the list's room listed nothing)
		for (unsigned q = tid; q < n_blk; q += 256u) {
			const uint4 bd = S.blk[q];
			if (bd.w == 3u) continue;
			const unsigned cpos = bd.x & 0xFFFFu, p1 = bd.x >> 16, p2 = bd.y & 0xFFFFu, p3 = bd.y >> 16, p4 = bd.z;
			const unsigned l_chr = p1 - cpos, l_str = p2 - p1 - 1u, l_s = p3 - p2 - 1u, l_e = p4 - p3 - 1u;
			uint4 out = make_uint4(0u, 0u, 0u, 2u);              // .w: 0 kept (.x chromosome | strand << 16, .y / .z the block), 1 passed over, 2 not the usual shape
			if (!(l_chr > 7u || l_str > 7u || l_s - 1u > 8u || l_e - 1u > 8u)) {
				// the coordinates: the eight bytes that end at the delimiter, and the byte ahead of them
				unsigned long long ws, we, wc, wt;
				__builtin_memcpy(&ws, &S.text[FP_PAD + p3 - 8u], 8);
				__builtin_memcpy(&we, &S.text[FP_PAD + p4 - 8u], 8);
				__builtin_memcpy(&wc, &S.text[FP_PAD + cpos], 8);
				__builtin_memcpy(&wt, &S.text[FP_PAD + p1 + 1u], 8);
				const unsigned s9 = (unsigned)(unsigned char)text[(int)p3 - 9] - (unsigned)'0', e9 = (unsigned)(unsigned char)text[(int)p4 - 9] - (unsigned)'0';
				unsigned vs, ve;
				const bool good_s = fp_number8(ws, min(l_s, 8u), vs), good_e = fp_number8(we, min(l_e, 8u), ve);
				bool good = good_s && good_e;
				if (l_s == 9u) { good = good && s9 <= 9u; vs += s9 * 100000000u; }
				if (l_e == 9u) { good = good && e9 <= 9u; ve += e9 * 100000000u; }
				if (good) {
					out.w = 1u;
					if (l_chr != 0u) {
						const unsigned long long kb = wc & ((1ull << (8u * l_chr)) - 1ull);
						const unsigned long long key = ((unsigned long long)__builtin_bswap32((unsigned)kb) << 32) | (unsigned long long)__builtin_bswap32((unsigned)(kb >> 32)) | (unsigned long long)l_chr;
						unsigned cid = MRF_NOCHROM;
						for (unsigned sl = mrf_key_slot(key);; sl = (sl + 1u) & (FP_DICT - 1u)) {
							const unsigned long long k = S.ckey[sl];
							if (k == key) { cid = S.cid[sl]; break; }
							if (k == 0ull) break;
						}
						const int s0 = (int)vs - 1, e0 = (int)ve;
						LocProbe P;
						P.chrom = -1; P.bin = 0;
						if (cid < T.n_chrom && route_covered(T, S.chrom[cid], (int)cid, s0, e0, P)) {
							const unsigned long long tb = wt & ((1ull << (8u * l_str)) - 1ull);
							const unsigned long long tkey = ((unsigned long long)__builtin_bswap32((unsigned)tb) << 32) | (unsigned long long)__builtin_bswap32((unsigned)(tb >> 32)) | (unsigned long long)l_str;
							unsigned sid = 256u;
							for (unsigned z = 0; z < FP_STRANDS; ++z) { const unsigned long long cur = S.strand[z]; if (cur == tkey) { sid = z; break; } if (cur == STRAND_EMPTY) break; }
							if (sid == 256u) sid = mrf_strand_slot(nullptr, G.strand_tab, MrfLdsView{text + (p1 + 1u), l_str}, err);
							out = make_uint4(cid | (sid << 16), (unsigned)s0, (unsigned)e0, 0u);
						}
					}
				}
			}
			S.blk[q] = out;
		}
		__syncthreads();
		// ---- (3)
		if (live) {
			if (dropped) O.key[i] = ROUTE_KEY_DROPPED;
			else {
				ReadAcc A;
				ReadBig B;
				A.init();
				for (unsigned q = 0; q < nb; ++q) {
					const uint4 r = S.blk[first + q];
					if (r.w == 2u) { odd = true; break; }
					if (r.w == 0u) A.add(B, r.x & 0xFFFFu, r.x >> 16, (int)r.y, (int)r.z);
				}
				if (odd) defer(i, w0 + start, eol - start);
				else {
					LocProbe P;
					P.chrom = -1; P.bin = 0;
					A.finish(B, T, S.chrom, P, O, (unsigned)i);
				}
			}
		}
	}
	}
}

// ---- the same walk for lsq_mrf_parse_device: pass 1, blocks per data line (0 for skipped lines), first failing line
__global__ void __launch_bounds__(256) lsq_mrf_count_kernel(MrfText X, unsigned *line_nb, unsigned long long *err) {
	__shared__ MrfTileLds S;
	mrf_tile_lines<false>(S, blockIdx.x, X.text, X.len, X.tile_base, X.has_header, nullptr, nullptr, 0u, nullptr, [&](const unsigned long long i, auto line) {
		unsigned nb = 0;
		if (!lsq::mrf_line_is_skipped(line)) {
			const bool ok = lsq::mrf_split_line(line, [&](auto, auto, int64_t, int64_t) { ++nb; });
			if (!ok) { atomicMin(&err[0], X.first_line + i); nb = 0; }
		}
		line_nb[i] = nb;
	});
}

struct MrfOut {
	unsigned long long *blk_off;
	unsigned *line_no;
	int *blk_start, *blk_end;
	unsigned short *blk_chrom;
	unsigned char *blk_strand;
};

// pass 2: every read's blocks to their place (rd_idx / bk_off: exclusive prefix sums of "has blocks" / of the block counts over the data lines)
__global__ void __launch_bounds__(256) lsq_mrf_write_kernel(MrfText X, const unsigned *line_nb, const unsigned long long *rd_idx, const unsigned long long *bk_off,
                                                            MrfDict G, MrfOut O, unsigned long long *err) {
	__shared__ MrfTileLds S;
	const MrfDict D = mrf_stage_dict(S, G);
	const long long LIM = 1ll << 30;
	mrf_tile_lines<false>(S, blockIdx.x, X.text, X.len, X.tile_base, X.has_header, nullptr, nullptr, 0u, nullptr, [&](const unsigned long long i, auto line) {
		const unsigned nb = line_nb[i];
		const unsigned long long r = rd_idx[i], o = bk_off[i];
		if (i + 1 == X.n_lines) O.blk_off[r + (nb ? 1u : 0u)] = o + nb;
		if (!nb) return;
		O.blk_off[r] = o;
		O.line_no[r] = (unsigned)(X.first_line + i);
		unsigned long long w = o;
		(void)lsq::mrf_split_line(line, [&](auto chr, auto strand, int64_t start, int64_t end) {
			unsigned cid = mrf_chrom_lookup(D, chr);
			const unsigned sid = mrf_strand_slot(S.strand, D.strand_tab, strand, err);
			long long s0 = start - 1, e0 = end;
			if (s0 <= -LIM || e0 >= LIM || s0 >= LIM || e0 <= -LIM) { cid = MRF_NOCHROM; s0 = 0; e0 = 0; }
			O.blk_start[w] = (int)s0; O.blk_end[w] = (int)e0;
			O.blk_chrom[w] = (unsigned short)cid; O.blk_strand[w] = (unsigned char)sid;
			++w;
		});
	});
}

struct DevParsed {
	uint64_t n_reads = 0, n_blocks = 0;
	DevBuf<unsigned long long> blk_off;
	DevBuf<unsigned> line_no;
	DevBuf<int> bs, be;
	DevBuf<unsigned short> bc;
	DevBuf<unsigned char> bst;
};

struct MappedFile {
	const char *data = nullptr;
	size_t len = 0;
	~MappedFile() { if (data) munmap((void *)data, len); }
};

// Host memory (or a file) to HBM through two pinned 32 MiB buffers of the context: a few worker threads fill one -- fill(dst, offset,
// bytes) of their share of the slice: pread() of a file, memcpy() of an array -- while the DMA engine drains the other.  38-53 GB/s,
// against 18-20 GB/s of the runtime's own staging of pageable memory on its first pass over it (it pins the pages it is given, which
// is what a first copy of fresh arrays or of a fresh mapping pays for).  The buffers are made once per context.
constexpr size_t PIN_SLICE = 32ull << 20;
static int ensure_pinned_buffers(lsq_ctx *c) {
	if (c->pin_buf[0] && c->pin_buf[1] && c->pin_ev[0] && c->pin_ev[1]) return LSQ_OK;
	const bool ok = hipHostMalloc((void **)&c->pin_buf[0], PIN_SLICE, hipHostMallocDefault) == hipSuccess &&
	                hipHostMalloc((void **)&c->pin_buf[1], PIN_SLICE, hipHostMallocDefault) == hipSuccess &&
	                hipEventCreateWithFlags(&c->pin_ev[0], hipEventDisableTiming) == hipSuccess &&
	                hipEventCreateWithFlags(&c->pin_ev[1], hipEventDisableTiming) == hipSuccess;
	if (!ok) {
		(void)hipGetLastError();
		for (int q = 0; q < 2; ++q) { if (c->pin_buf[q]) (void)hipHostFree(c->pin_buf[q]); if (c->pin_ev[q]) (void)hipEventDestroy(c->pin_ev[q]); c->pin_buf[q] = nullptr; c->pin_ev[q] = nullptr; }
		return fail(LSQ_E_INTERNAL, "no pinned host buffers");
	}
	return LSQ_OK;
}
template <class Fill>
static int pinned_pipeline(lsq_ctx *c, unsigned char *d_dst, const size_t len, Fill &&fill, const char *what) {
	if (len == 0) return LSQ_OK;
	int rc = ensure_pinned_buffers(c);
	if (rc) return rc;
	hipStream_t st = c->stream;
	unsigned char *const *pin = c->pin_buf;
	int rc_copy = LSQ_OK;
	int T = std::max(1, std::min(16, host_threads(0)));
	if (const char *e = getenv("LSQ_COPY_THREADS")) { const int v = atoi(e); if (v > 0 && v <= 64) T = v; }      // developer aid
	const long n_slices = (long)((len + PIN_SLICE - 1) / PIN_SLICE);
	std::atomic<long> go{-1}, filled{0};
	std::atomic<int> io_error{0};
	std::atomic<bool> give_up{false};         // set on every way out of this function: a worker that still waits for its slice leaves
	ThreadGroup workers;                      // (joined on every way out, after give_up is set: declared first, destroyed last)
	struct GiveUp { std::atomic<bool> &f; ~GiveUp() { f.store(true, std::memory_order_release); } } give_up_on_exit{give_up};
	for (int t = 0; t < T; ++t) workers.spawn([&, t] {
		for (long sl = 0; sl < n_slices; ++sl) {
			while (go.load(std::memory_order_acquire) < sl) { if (give_up.load(std::memory_order_acquire)) return; std::this_thread::yield(); }
			const size_t off = (size_t)sl * PIN_SLICE, nby = std::min<size_t>(PIN_SLICE, len - off);
			const size_t a = nby * (size_t)t / (size_t)T, b = nby * (size_t)(t + 1) / (size_t)T;
			if (b > a && !fill(pin[sl & 1] + a, off + a, b - a)) io_error.store(1);
			filled.fetch_add(1, std::memory_order_release);
		}
	});
	for (long sl = 0; sl < n_slices; ++sl) {
		const int k = (int)(sl & 1);
		if (sl >= 2 && rc_copy == LSQ_OK && hipEventSynchronize(c->pin_ev[k]) != hipSuccess) rc_copy = fail(LSQ_E_DEVICE, "hipEventSynchronize failed in the copy of %s", what);
		go.store(sl, std::memory_order_release);
		while (filled.load(std::memory_order_acquire) < (long)T * (sl + 1)) std::this_thread::yield();
		const size_t off = (size_t)sl * PIN_SLICE, nby = std::min<size_t>(PIN_SLICE, len - off);
		if (rc_copy == LSQ_OK && (hipMemcpyAsync(d_dst + off, pin[k], nby, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(c->pin_ev[k], st) != hipSuccess))
			rc_copy = fail(LSQ_E_DEVICE, "hipMemcpyAsync failed in the copy of %s", what);
	}
	workers.join();
	if (hipStreamSynchronize(st) != hipSuccess && rc_copy == LSQ_OK) rc_copy = fail(LSQ_E_DEVICE, "the copy of %s failed", what);
	if (workers.failed() && rc_copy == LSQ_OK) rc_copy = fail(LSQ_E_INTERNAL, "a helper thread failed: %s", workers.error().c_str());
	if (io_error.load() && rc_copy == LSQ_OK) rc_copy = fail(LSQ_E_IO, "cannot read %s", what);
	return rc_copy;
}

static int stage_text_file(lsq_ctx *c, const char *path, unsigned long long byte_begin, unsigned long long byte_end, lsq_text &T) {
	HostStopwatch SW;
	int fd = open(path, O_RDONLY);
	if (fd < 0) return fail(LSQ_E_IO, "cannot open reads file %s", path);
	struct stat sb;
	if (fstat(fd, &sb) != 0) { close(fd); return fail(LSQ_E_IO, "cannot stat %s", path); }
	// Small files are mapped and copied as they are (the runtime stages pageable memory through its own
	// pinned buffers on one thread: 18 GB/s measured).  Files of a gigabyte and more are never mapped:
	// a few worker threads pread() them, a slice at a time, into the context's two pinned buffers while
	// the DMA engine drains the other buffer (38 GB/s, and no page-table build-up and tear-down for
	// gigabytes of mapping).
	const unsigned long long file_len = (unsigned long long)sb.st_size;
	byte_end = std::min(byte_end, file_len);
	byte_begin = std::min(byte_begin, byte_end);
	const unsigned long long len = byte_end - byte_begin;          // the bytes [byte_begin, byte_end) of the file
	unsigned long long pinned_min = 1ull << 30;           // below a gigabyte making the pinned buffers (once per context) costs more than they save
	if (const char *e = getenv("LSQ_PINNED_COPY_MIN")) { const long long v = atoll(e); if (v >= 0) pinned_min = (unsigned long long)v; }   // tests
	bool pinned = len >= pinned_min && len >= 2 * PIN_SLICE;
	struct FdCloser { int fd; ~FdCloser() { if (fd >= 0) close(fd); } } fdc{fd};
	MappedFile mf;
	hipStream_t st = c->stream;
	int rc;
	T.path = path; T.len = len; T.offset = byte_begin; T.h2d_ms = 0;
	if (len == 0) return LSQ_OK;
	DevBuf<unsigned char> &d_text = T.d_text;
	if ((rc = d_text.alloc(len + 16))) return rc;
	HIP_TRY(hipEventRecord(c->evt0, st));
	if (pinned && ensure_pinned_buffers(c) != LSQ_OK) pinned = false;
	if (pinned) {
		// (tried: the workers copying out of a mapping of the file instead -- 55 GB/s against 40-46, but the mapping's tear-down
		// costs 70 ms on one thread and more when the workers share it; and 24 / 32 workers on a box's 16 cores: slower)
		rc = pinned_pipeline(c, d_text.p, (size_t)len, [&](unsigned char *dst, size_t off, size_t n) {
			size_t a = 0;
			while (a < n) {
				const ssize_t got = pread(fd, dst + a, n - a, (off_t)(byte_begin + off + a));
				if (got <= 0) return false;
				a += (size_t)got;
			}
			return true;
		}, path);
		if (rc) return rc;
	} else {
		int rc_copy = LSQ_OK;
		void *m = mmap(nullptr, (size_t)file_len, PROT_READ, MAP_PRIVATE, fd, 0);
		if (m == MAP_FAILED) rc_copy = fail(LSQ_E_IO, "cannot map %s", path);
		else {
			mf.data = (const char *)m; mf.len = (size_t)file_len;
			madvise(m, mf.len, MADV_SEQUENTIAL);
			for (size_t off = 0; off < len && rc_copy == LSQ_OK; off += PIN_SLICE) {
				const size_t nby = std::min<size_t>(PIN_SLICE, len - off);
				if (hipMemcpyAsync(d_text.p + off, mf.data + byte_begin + off, nby, hipMemcpyHostToDevice, st) != hipSuccess) rc_copy = fail(LSQ_E_DEVICE, "hipMemcpyAsync failed in the text copy");
			}
			if (hipStreamSynchronize(st) != hipSuccess && rc_copy == LSQ_OK) rc_copy = fail(LSQ_E_DEVICE, "text copy failed");
		}
		if (rc_copy) return rc_copy;
	}
	HIP_TRY(hipEventRecord(c->evt1, st));
	HIP_TRY(hipEventSynchronize(c->evt1));
	(void)hipEventElapsedTime(&T.h2d_ms, c->evt0, c->evt1);
	SW.mark("text: open and copy to HBM");
	return LSQ_OK;
}


// newline counts of a staged text, per tile, and their prefix sums (kept with the text): lsq_text_lines runs this ahead of the parse
static int scan_newlines(lsq_ctx *c, lsq_text &T) {
	if (T.scanned) return LSQ_OK;
	hipStream_t st = c->stream;
	int rc;
	const unsigned long long len = T.len;
	T.n_nl = 0;
	if (len) {
		const unsigned long long n_tiles = (len + MRF_TILE - 1) / MRF_TILE;
		if (n_tiles > 0x7FFFFFFFull) return fail(LSQ_E_RANGE, "reads file larger than 16 TiB");
		DevBuf<unsigned> d_tile_cnt;
		ScanScratch SS;
		if ((rc = d_tile_cnt.alloc(n_tiles)) || (rc = T.d_tile_base.alloc(n_tiles + 1)) || (rc = SS.reserve(n_tiles))) return rc;
		StageClock k(c, st, 0);
		hipLaunchKernelGGL(lsq_mrf_newline_count_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, T.d_text.p, len, d_tile_cnt.p);
		HIP_TRY(hipGetLastError());
		if ((rc = device_scan<1>(SS, d_tile_cnt.p, n_tiles, T.d_tile_base.p, st))) return rc;
		k.end(len + 12ull * n_tiles);
		HIP_TRY(hipMemcpyAsync(&T.n_nl, T.d_tile_base.p + n_tiles, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	T.scanned = true;
	return LSQ_OK;
}

// The dictionaries of a parse: the events' chromosome names behind a hash table, the strand table seeded with the strands
// already known.  The events' strand dictionary grows by the strings the file introduces (as it does under lsq_mrf_parse).
struct MrfDictDev {
	DevBuf<unsigned> d_hash, d_id, d_off;
	DevBuf<unsigned long long> d_strand, d_err;
	DevBuf<char> d_names;
	size_t n_seed = 0;
	MrfDict D{};
	int build(lsq_ctx *c, hipStream_t st) {
		lsq_events &E = *c->E;
		int rc;
		const size_t nc = E.covered.size();
		size_t tab = 2;
		while (tab < 4 * nc) tab <<= 1;
		std::vector<unsigned> h_hash(tab, 0), h_id(tab, 0), h_off(nc + 1, 0);
		std::string h_names;
		for (size_t id = 0; id < nc; ++id) {
			const std::string &nm = E.chroms.names[id];
			const unsigned h = mrf_fnv32(nm.data(), nm.size());
			size_t i = (size_t)(h & (unsigned)(tab - 1));
			while (h_hash[i] != 0) i = (i + 1) & (tab - 1);
			h_hash[i] = h; h_id[i] = (unsigned)id;
			h_names += nm;
			h_off[id + 1] = (unsigned)h_names.size();
		}
		if (E.strands.names.size() > 256) return fail(LSQ_E_RANGE, "more than 256 distinct strand strings");
		n_seed = E.strands.names.size();
		std::vector<unsigned long long> h_strand(256, STRAND_EMPTY);
		for (size_t i = 0; i < n_seed; ++i) {
			const std::string &s = E.strands.names[i];
			h_strand[i] = s.size() <= 7 ? mrf_strand_key(s.data(), s.size()) : STRAND_UNMATCHABLE;
		}
		const unsigned long long err[4] = {MRF_NO_ERR, 0, 0, 0};
		if ((rc = d_hash.upload(h_hash.data(), tab, st)) || (rc = d_id.upload(h_id.data(), tab, st)) || (rc = d_off.upload(h_off.data(), nc + 1, st)) ||
		    (rc = d_names.upload(h_names.data(), h_names.size(), st)) || (rc = d_strand.upload(h_strand.data(), 256, st)) || (rc = d_err.upload(err, 4, st))) return rc;
		HIP_TRY(hipStreamSynchronize(st));            // the host vectors go out of scope
		D.chrom_hash = d_hash.p; D.chrom_id = d_id.p; D.name_off = d_off.p; D.names = d_names.p; D.mask = (unsigned)(tab - 1);
		D.n_chrom = (unsigned)nc; D.names_bytes = (unsigned)h_names.size(); D.strand_tab = d_strand.p;
		return LSQ_OK;
	}
	int reset_errors(hipStream_t st) {
		static const unsigned long long err0[4] = {MRF_NO_ERR, 0, 0, 0};
		HIP_TRY(hipMemcpyAsync(d_err.p, err0, sizeof(err0), hipMemcpyHostToDevice, st));
		return LSQ_OK;
	}
	// after the parse kernels have run and the stream has been waited for: the first failing line, strand strings out of range, new strands
	int settle(lsq_ctx *c, const lsq_text &T, unsigned has_header, unsigned long long first_line, hipStream_t st) {
		lsq_events &E = *c->E;
		unsigned long long err[4];
		std::vector<unsigned long long> h_strand(256);
		HIP_TRY(hipMemcpyAsync(err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(h_strand.data(), d_strand.p, 256 * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (err[0] != MRF_NO_ERR) {
			// the text of the failing line, from the file: between the newline that ends the line before it and its own
			const unsigned long long want = err[0] - first_line + has_header;      // ordinal of the newline that ends the failing line
			std::string text;
			const int fd = open(T.path.c_str(), O_RDONLY);
			if (fd >= 0) {
				// walk the file's range for the want-th newline (an error path: speed does not matter, bounded memory does)
				std::vector<char> buf(1 << 20);
				unsigned long long seen = 0, pos = 0;
				bool in_line = want == 0, done = false;
				while (!done && pos < T.len) {
					const size_t ask = (size_t)std::min<unsigned long long>(buf.size(), T.len - pos);
					const ssize_t got = pread(fd, buf.data(), ask, (off_t)(T.offset + pos));
					if (got <= 0) break;
					for (ssize_t q = 0; q < got && !done; ++q) {
						if (buf[(size_t)q] == '\n') {
							if (in_line) done = true;
							else if (++seen == want) in_line = true;
						} else if (in_line) text.push_back(buf[(size_t)q]);
					}
					pos += (unsigned long long)got;
				}
				close(fd);
			}
			return fail(LSQ_E_PARSE, "#%llu:%s", err[0], text.c_str());
		}
		if (err[1]) return fail(LSQ_E_UNSUPPORTED, "a strand string longer than 7 bytes: outside the device parser's range (lsq_mrf_parse handles it)");
		if (err[2]) return fail(LSQ_E_RANGE, "more than 256 distinct strand strings");
		for (size_t i = n_seed; i < 256 && h_strand[i] != STRAND_EMPTY; ++i) {
			const unsigned long long k = h_strand[i];
			std::string s;
			for (unsigned j = 0; j < (unsigned)(k & 0xFF); ++j) s.push_back((char)(k >> (56 - 8 * j)));
			const int id = E.strands.intern(s);
			if (id != (int)i) return fail(LSQ_E_STATE, "strand dictionary changed while a reads file was being parsed");
		}
		n_seed = E.strands.names.size();
		return LSQ_OK;
	}
};

// Parses staged text on the device into the arrays of lsq_mrf_parse (file order): lsq_mrf_parse_device.
static int parse_staged_text(lsq_ctx *c, const char *read_format, lsq_text &T, unsigned has_header, unsigned long long first_line, DevParsed &out, float *h2d_ms, float *parse_ms) {
	if (!read_format) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (strcmp(read_format, "MRF_SINGLE") != 0) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", read_format);
	hipStream_t st = c->stream;
	int rc;
	const unsigned long long zero_off = 0;
	out.n_reads = out.n_blocks = 0;
	auto empty_result = [&]() -> int {
		int r2;
		if ((r2 = out.blk_off.upload(&zero_off, 1, st)) || (r2 = out.line_no.alloc(0)) || (r2 = out.bs.alloc(0)) || (r2 = out.be.alloc(0)) ||
		    (r2 = out.bc.alloc(0)) || (r2 = out.bst.alloc(0))) return r2;
		HIP_TRY(hipStreamSynchronize(st));
		return LSQ_OK;
	};
	if (h2d_ms) *h2d_ms = T.h2d_ms;
	if (parse_ms) *parse_ms = 0;
	if (T.len == 0) return empty_result();
	HIP_TRY(hipEventRecord(c->ev1, st));
	if ((rc = scan_newlines(c, T))) return rc;
	const unsigned long long n_nl = T.n_nl;
	if (n_nl < 1 + has_header) return empty_result();   // header only (or no terminated line at all)
	const unsigned long long n_lines = n_nl - has_header;
	if (first_line + n_lines > 0xFFFFFFFFull) return fail(LSQ_E_RANGE, "more than 2^32 lines");
	const unsigned n_tiles = (unsigned)((T.len + MRF_TILE - 1) / MRF_TILE);
	DevBuf<unsigned> d_line_nb;
	DevBuf<unsigned long long> d_rd_idx, d_bk_off;
	ScanScratch SS;
	MrfDictDev DD;
	if ((rc = d_line_nb.alloc(n_lines)) || (rc = d_rd_idx.alloc(n_lines + 1)) || (rc = d_bk_off.alloc(n_lines + 1)) || (rc = SS.reserve(n_lines)) || (rc = DD.build(c, st))) return rc;
	MrfText X{T.d_text.p, T.len, T.d_tile_base.p, has_header, first_line, n_lines};
	hipLaunchKernelGGL(lsq_mrf_count_kernel, dim3(n_tiles), dim3(256), 0, st, X, d_line_nb.p, DD.d_err.p);
	HIP_TRY(hipGetLastError());
	if ((rc = device_scan<1, true>(SS, d_line_nb.p, n_lines, d_rd_idx.p, st)) || (rc = device_scan<1, false>(SS, d_line_nb.p, n_lines, d_bk_off.p, st))) return rc;
	unsigned long long n_reads = 0, n_blocks = 0;
	HIP_TRY(hipMemcpyAsync(&n_reads, d_rd_idx.p + n_lines, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&n_blocks, d_bk_off.p + n_lines, 8, hipMemcpyDeviceToHost, st));
	if ((rc = DD.settle(c, T, has_header, first_line, st))) return rc;          // (waits for the stream) the first failing line ends the run here
	if ((rc = out.blk_off.alloc(n_reads + 1)) || (rc = out.line_no.alloc(n_reads)) || (rc = out.bs.alloc(n_blocks)) || (rc = out.be.alloc(n_blocks)) ||
	    (rc = out.bc.alloc(n_blocks)) || (rc = out.bst.alloc(n_blocks))) return rc;
	MrfOut O{};
	O.blk_off = out.blk_off.p; O.line_no = out.line_no.p; O.blk_start = out.bs.p; O.blk_end = out.be.p; O.blk_chrom = out.bc.p; O.blk_strand = out.bst.p;
	hipLaunchKernelGGL(lsq_mrf_write_kernel, dim3(n_tiles), dim3(256), 0, st, X, (const unsigned *)d_line_nb.p, (const unsigned long long *)d_rd_idx.p,
	                   (const unsigned long long *)d_bk_off.p, DD.D, O, DD.d_err.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(c->ev2, st));
	if ((rc = DD.settle(c, T, has_header, first_line, st))) return rc;
	if (parse_ms) (void)hipEventElapsedTime(parse_ms, c->ev1, c->ev2);
	out.n_reads = n_reads; out.n_blocks = n_blocks;
	return LSQ_OK;
}

// open -> format literal: the order in which the reference meets a bad file or literal
static int check_mrf_file(const char *read_format, const char *path) {
	if (!read_format || !path) return fail(LSQ_E_ARG, "null argument");
	FILE *f = fopen(path, "rb");
	if (!f) return fail(LSQ_E_IO, "cannot open reads file %s", path);
	fclose(f);
	if (strcmp(read_format, "MRF_SINGLE") != 0) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", read_format);
	return LSQ_OK;
}
