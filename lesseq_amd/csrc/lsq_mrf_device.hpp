// MRF_SINGLE parsed on the device (included by lsq_ingest.hip; not a public header).
//
// The text goes to HBM as it is.  Newlines are found 16 bytes per lane, their ordinals by a
// two-level prefix sum; one lane then owns one line and runs the shared splitter
// (lsq_mrf_line.hpp -- the same code the host parser runs) twice: once to count the line's
// blocks, once to write them at the offsets a second prefix sum gives.  Result: the parsed
// blocks in file order, identical to lsq_mrf_parse's arrays (count/count.cpp:279-336 minus the
// containment filter), already in HBM for the ingest kernels.
//   - the first line is the header (count.cpp:283); a last line without '\n' is never seen (:285)
//   - '#' lines and the literal "AlignmentBlocks" consume a line number only (:288)
//   - a field that fails the cast stops the run: the FIRST such line in file order is reported
//   - chromosome names resolve against the events' chromosomes (hash table in global memory);
//     strand strings against a 256-slot table seeded with the strands already known, grown with
//     atomicCAS (strings of at most 7 bytes; longer ones give LSQ_E_UNSUPPORTED -- use lsq_mrf_parse)
#pragma once

constexpr unsigned MRF_TILE = 4096;                 // text bytes per workgroup pass: 256 lanes x 16
constexpr unsigned long long MRF_NO_ERR = ~0ull;
constexpr unsigned long long STRAND_EMPTY = ~0ull;
constexpr unsigned long long STRAND_UNMATCHABLE = ~0ull - 1;
constexpr unsigned MRF_NOCHROM = 0xFFFFu;

struct MrfDict {
	const unsigned long long *chrom_hash;   // open addressing, 0 = empty
	const unsigned *chrom_id;
	const unsigned *name_off;               // per chromosome id, into names
	const char *names;
	unsigned mask;
	unsigned long long *strand_tab;         // 256 slots
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

// bit j set iff byte j of the lane's 16 bytes is '\n'
__device__ inline unsigned mrf_newline_bits(const unsigned char *text, unsigned long long len, unsigned long long at) {
	if (at >= len) return 0;
	unsigned bits = 0;
	if (at + 16 <= len) {
		const uint4 v = *reinterpret_cast<const uint4 *>(text + at);
		const unsigned w[4] = {v.x, v.y, v.z, v.w};
		for (int q = 0; q < 4; ++q) {
			const unsigned x = w[q] ^ 0x0A0A0A0Au;
			const unsigned z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;   // 0x80 in every zero byte
			bits |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * q);
		}
	} else {
		for (unsigned j = 0; at + j < len; ++j) bits |= (text[at + j] == '\n' ? 1u : 0u) << j;
	}
	return bits;
}

__global__ void __launch_bounds__(256) lsq_mrf_newline_count_kernel(const unsigned char *text, unsigned long long len, unsigned *tile_cnt) {
	__shared__ unsigned lds4[4];
	const unsigned long long at = (unsigned long long)blockIdx.x * MRF_TILE + threadIdx.x * 16ull;
	unsigned total;
	(void)mrf_block_excl_scan((unsigned)__popc(mrf_newline_bits(text, len, at)), lds4, total);
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) lsq_mrf_newline_pos_kernel(const unsigned char *text, unsigned long long len,
                                                                  const unsigned long long *tile_base, unsigned long long *nl_pos) {
	__shared__ unsigned lds4[4];
	const unsigned long long at = (unsigned long long)blockIdx.x * MRF_TILE + threadIdx.x * 16ull;
	unsigned bits = mrf_newline_bits(text, len, at);
	unsigned total;
	unsigned long long w = tile_base[blockIdx.x] + mrf_block_excl_scan((unsigned)__popc(bits), lds4, total);
	while (bits) { const unsigned j = (unsigned)__ffs((int)bits) - 1u; bits &= bits - 1u; nl_pos[w++] = at + j; }
}

// data line i (0-based) of the text: with a header the bytes between newlines i and i+1, without one (a
// slice of a file that starts on a line boundary) those between newlines i-1 and i
__device__ inline lsq::MrfView mrf_data_line(const unsigned char *text, const unsigned long long *nl_pos, unsigned long long i, unsigned has_header) {
	const unsigned long long e = i + has_header;
	const unsigned long long a = e == 0 ? 0ull : nl_pos[e - 1] + 1, b = nl_pos[e];
	return lsq::MrfView{reinterpret_cast<const char *>(text) + a, (size_t)(b - a)};
}

// pass 1: blocks per data line (0 for skipped lines), first failing line, per-workgroup sums
__global__ void __launch_bounds__(256) lsq_mrf_count_kernel(const unsigned char *text, const unsigned long long *nl_pos, unsigned long long n_lines,
                                                            unsigned has_header, unsigned long long first_line,
                                                            unsigned *line_nb, unsigned *wg_reads, unsigned *wg_blocks, unsigned long long *err) {
	__shared__ unsigned lds4[4];
	const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
	unsigned nb = 0;
	if (i < n_lines) {
		const unsigned long long L = first_line + i;          // the line's number in the whole file (read name "read-<L>")
		const lsq::MrfView line = mrf_data_line(text, nl_pos, i, has_header);
		if (!lsq::mrf_line_is_skipped(line)) {
			const bool ok = lsq::mrf_split_line(line, [&](lsq::MrfView, lsq::MrfView, int64_t, int64_t) { ++nb; });
			if (!ok) { atomicMin(&err[0], L); nb = 0; }
		}
		line_nb[i] = nb;
	}
	unsigned tr, tb;
	(void)mrf_block_excl_scan(nb ? 1u : 0u, lds4, tr);
	(void)mrf_block_excl_scan(nb, lds4, tb);
	if (threadIdx.x == 0) { wg_reads[blockIdx.x] = tr; wg_blocks[blockIdx.x] = tb; }
}

__device__ inline unsigned mrf_chrom_lookup(const MrfDict &D, lsq::MrfView s) {
	unsigned long long h = 0xcbf29ce484222325ull;
	for (size_t j = 0; j < s.n; ++j) { h ^= (unsigned char)s.p[j]; h *= 0x100000001b3ull; }
	if (h == 0) h = 1;
	for (unsigned i = (unsigned)h & D.mask;; i = (i + 1u) & D.mask) {
		const unsigned long long t = D.chrom_hash[i];
		if (t == 0) return MRF_NOCHROM;
		if (t != h) continue;
		const unsigned id = D.chrom_id[i];
		const unsigned a = D.name_off[id], b = D.name_off[id + 1];
		if ((size_t)(b - a) != s.n) continue;
		bool same = true;
		for (size_t j = 0; j < s.n; ++j) same = same && D.names[a + j] == s.p[j];
		if (same) return id;
	}
}

// strand strings of <= 7 bytes as one order-preserving 64-bit key (bytes big-endian, length last)
LSQ_HD inline unsigned long long mrf_strand_key(const char *p, size_t n) {
	unsigned long long k = (unsigned long long)n;
	for (size_t j = 0; j < n; ++j) k |= (unsigned long long)(unsigned char)p[j] << (56 - 8 * j);
	return k;
}

__device__ inline unsigned mrf_strand_slot(unsigned long long *tab, lsq::MrfView s, unsigned long long *err) {
	if (s.n > 7) { atomicMax(&err[1], 1ull); return 0; }
	const unsigned long long key = mrf_strand_key(s.p, s.n);
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

struct MrfOut {
	unsigned long long *blk_off;
	unsigned *line_no;
	int *blk_start, *blk_end;
	unsigned short *blk_chrom;
	unsigned char *blk_strand;
};

// pass 2: every read's blocks to their place
__global__ void __launch_bounds__(256) lsq_mrf_write_kernel(const unsigned char *text, const unsigned long long *nl_pos, unsigned long long n_lines,
                                                            unsigned has_header, unsigned long long first_line, const unsigned *line_nb, const unsigned long long *rd_base, const unsigned long long *bk_base,
                                                            MrfDict D, MrfOut O, unsigned long long *err) {
	__shared__ unsigned lds4[4];
	const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
	const unsigned nb = i < n_lines ? line_nb[i] : 0u;
	unsigned tr, tb;
	const unsigned long long r = rd_base[blockIdx.x] + mrf_block_excl_scan(nb ? 1u : 0u, lds4, tr);
	const unsigned long long o = bk_base[blockIdx.x] + mrf_block_excl_scan(nb, lds4, tb);
	if (i + 1 == n_lines) O.blk_off[r + (nb ? 1u : 0u)] = o + nb;
	if (!nb) return;
	const unsigned long long L = first_line + i;
	const lsq::MrfView line = mrf_data_line(text, nl_pos, i, has_header);
	O.blk_off[r] = o;
	O.line_no[r] = (unsigned)L;
	unsigned long long w = o;
	const long long LIM = 1ll << 30;
	(void)lsq::mrf_split_line(line, [&](lsq::MrfView chr, lsq::MrfView strand, int64_t start, int64_t end) {
		unsigned cid = mrf_chrom_lookup(D, chr);
		const unsigned sid = mrf_strand_slot(D.strand_tab, strand, err);
		long long s0 = start - 1, e0 = end;
		if (s0 <= -LIM || e0 >= LIM || s0 >= LIM || e0 <= -LIM) { cid = MRF_NOCHROM; s0 = 0; e0 = 0; }
		O.blk_start[w] = (int)s0; O.blk_end[w] = (int)e0;
		O.blk_chrom[w] = (unsigned short)cid; O.blk_strand[w] = (unsigned char)sid;
		++w;
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

static int stage_text_file(lsq_ctx *c, const char *path, unsigned long long byte_begin, unsigned long long byte_end, lsq_text &T) {
	HostStopwatch SW;
	int fd = open(path, O_RDONLY);
	if (fd < 0) return fail(LSQ_E_IO, "cannot open reads file %s", path);
	struct stat sb;
	if (fstat(fd, &sb) != 0) { close(fd); return fail(LSQ_E_IO, "cannot stat %s", path); }
	// Small files are mapped and copied as they are (the runtime stages pageable memory through its own
	// pinned buffers on one thread: 18 GB/s measured).  Files of a gigabyte and more are never mapped:
	// a few worker threads pread() them, a slice at a time, into two pinned 32 MiB buffers of ours while
	// the DMA engine drains the other buffer (38 GB/s, and no page-table build-up and tear-down for
	// gigabytes of mapping).
	const unsigned long long file_len = (unsigned long long)sb.st_size;
	byte_end = std::min(byte_end, file_len);
	byte_begin = std::min(byte_begin, byte_end);
	const unsigned long long len = byte_end - byte_begin;          // the bytes [byte_begin, byte_end) of the file
	const size_t SLICE = 32ull << 20;
	unsigned long long pinned_min = 1ull << 30;           // below a gigabyte allocating the pinned buffers costs more than they save
	if (const char *e = getenv("LSQ_PINNED_COPY_MIN")) { const long long v = atoll(e); if (v >= 0) pinned_min = (unsigned long long)v; }   // tests
	bool pinned = len >= pinned_min && len >= 2 * SLICE;
	struct FdCloser { int fd; ~FdCloser() { if (fd >= 0) close(fd); } } fdc{fd};
	MappedFile mf;
	hipStream_t st = c->stream;
	int rc;
	T.path = path; T.len = len; T.offset = byte_begin; T.h2d_ms = 0;
	if (len == 0) return LSQ_OK;
	DevBuf<unsigned char> &d_text = T.d_text;
	if ((rc = d_text.alloc(len + 16))) return rc;
	HIP_TRY(hipEventRecord(c->evt0, st));
	{
		unsigned char *pin[2] = {nullptr, nullptr};
		hipEvent_t drained[2] = {nullptr, nullptr};
		if (pinned) {
			pinned = hipHostMalloc((void **)&pin[0], SLICE, hipHostMallocDefault) == hipSuccess &&
			         hipHostMalloc((void **)&pin[1], SLICE, hipHostMallocDefault) == hipSuccess &&
			         hipEventCreateWithFlags(&drained[0], hipEventDisableTiming) == hipSuccess &&
			         hipEventCreateWithFlags(&drained[1], hipEventDisableTiming) == hipSuccess;
			(void)hipGetLastError();
		}
		int rc_copy = LSQ_OK;
		if (pinned) {
			const int T = std::max(1, std::min(16, host_threads(0)));
			const long n_slices = (long)((len + SLICE - 1) / SLICE);
			std::atomic<long> go{-1}, filled{0};
			std::atomic<int> io_error{0};
			std::atomic<bool> give_up{false};         // set on every way out of this block: a worker that still waits for its slice leaves
			ThreadGroup workers;                      // (joined on every way out, after give_up is set: declared first, destroyed last)
			struct GiveUp { std::atomic<bool> &f; ~GiveUp() { f.store(true, std::memory_order_release); } } give_up_on_exit{give_up};
			for (int t = 0; t < T; ++t) workers.spawn([&, t] {
				for (long sl = 0; sl < n_slices; ++sl) {
					while (go.load(std::memory_order_acquire) < sl) { if (give_up.load(std::memory_order_acquire)) return; std::this_thread::yield(); }
					const size_t off = (size_t)sl * SLICE, nby = std::min<size_t>(SLICE, len - off);
					size_t a = nby * (size_t)t / (size_t)T;
					const size_t b = nby * (size_t)(t + 1) / (size_t)T;
					while (a < b) {
						const ssize_t got = pread(fd, pin[sl & 1] + a, b - a, (off_t)(byte_begin + off + a));
						if (got <= 0) { io_error.store(1); break; }
						a += (size_t)got;
					}
					filled.fetch_add(1, std::memory_order_release);
				}
			});
			for (long sl = 0; sl < n_slices; ++sl) {
				const int k = (int)(sl & 1);
				if (sl >= 2 && rc_copy == LSQ_OK && hipEventSynchronize(drained[k]) != hipSuccess) rc_copy = fail(LSQ_E_DEVICE, "hipEventSynchronize failed in the text copy");
				go.store(sl, std::memory_order_release);
				while (filled.load(std::memory_order_acquire) < (long)T * (sl + 1)) std::this_thread::yield();
				const size_t off = (size_t)sl * SLICE, nby = std::min<size_t>(SLICE, len - off);
				if (rc_copy == LSQ_OK && (hipMemcpyAsync(d_text.p + off, pin[k], nby, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(drained[k], st) != hipSuccess))
					rc_copy = fail(LSQ_E_DEVICE, "hipMemcpyAsync failed in the text copy");
			}
			workers.join();
			if (hipStreamSynchronize(st) != hipSuccess && rc_copy == LSQ_OK) rc_copy = fail(LSQ_E_DEVICE, "text copy failed");
			if (io_error.load() && rc_copy == LSQ_OK) rc_copy = fail(LSQ_E_IO, "cannot read %s", path);
		} else {
			void *m = mmap(nullptr, (size_t)file_len, PROT_READ, MAP_PRIVATE, fd, 0);
			if (m == MAP_FAILED) rc_copy = fail(LSQ_E_IO, "cannot map %s", path);
			else {
				mf.data = (const char *)m; mf.len = (size_t)file_len;
				madvise(m, mf.len, MADV_SEQUENTIAL);
				for (size_t off = 0; off < len && rc_copy == LSQ_OK; off += SLICE) {
					const size_t nby = std::min<size_t>(SLICE, len - off);
					if (hipMemcpyAsync(d_text.p + off, mf.data + byte_begin + off, nby, hipMemcpyHostToDevice, st) != hipSuccess) rc_copy = fail(LSQ_E_DEVICE, "hipMemcpyAsync failed in the text copy");
				}
				if (hipStreamSynchronize(st) != hipSuccess && rc_copy == LSQ_OK) rc_copy = fail(LSQ_E_DEVICE, "text copy failed");
			}
		}
		for (int q = 0; q < 2; ++q) { if (pin[q]) (void)hipHostFree(pin[q]); if (drained[q]) (void)hipEventDestroy(drained[q]); }
		if (rc_copy) return rc_copy;
	}
	HIP_TRY(hipEventRecord(c->evt1, st));
	HIP_TRY(hipEventSynchronize(c->evt1));
	(void)hipEventElapsedTime(&T.h2d_ms, c->evt0, c->evt1);
	SW.mark("text: open and copy to HBM");
	return LSQ_OK;
}

// Parses staged text on the device.  The events' strand dictionary grows by the strand strings the
// file introduces (as it does under lsq_mrf_parse).
// newline positions of a staged text (kept with it): lsq_text_lines runs this ahead of the parse
static int scan_newlines(lsq_ctx *c, lsq_text &T) {
	if (T.scanned) return LSQ_OK;
	hipStream_t st = c->stream;
	int rc;
	const unsigned long long len = T.len;
	T.n_nl = 0;
	if (len) {
		const unsigned long long n_tiles = (len + MRF_TILE - 1) / MRF_TILE;
		if (n_tiles > 0x7FFFFFFFull) return fail(LSQ_E_RANGE, "reads file larger than 8 TiB");
		DevBuf<unsigned> d_tile_cnt;
		DevBuf<unsigned long long> d_tile_base;
		if ((rc = d_tile_cnt.alloc(n_tiles)) || (rc = d_tile_base.alloc(n_tiles + 1))) return rc;
		hipLaunchKernelGGL(lsq_mrf_newline_count_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, T.d_text.p, len, d_tile_cnt.p);
		HIP_TRY(hipGetLastError());
		hipLaunchKernelGGL(lsq_scan_u32_kernel<1>, dim3(1), dim3(1024), 0, st, d_tile_cnt.p, n_tiles, d_tile_base.p);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(&T.n_nl, d_tile_base.p + n_tiles, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if ((rc = T.d_nl_pos.alloc(T.n_nl))) return rc;
		if (T.n_nl) {
			hipLaunchKernelGGL(lsq_mrf_newline_pos_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, T.d_text.p, len, d_tile_base.p, T.d_nl_pos.p);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipStreamSynchronize(st));
		}
	}
	T.scanned = true;
	return LSQ_OK;
}

static int parse_staged_text(lsq_ctx *c, const char *read_format, lsq_text &T, unsigned has_header, unsigned long long first_line, DevParsed &out, float *h2d_ms, float *parse_ms) {
	if (!read_format) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	if (strcmp(read_format, "MRF_SINGLE") != 0) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", read_format);
	lsq_events &E = *c->E;
	HostStopwatch SW;
	hipStream_t st = c->stream;
	int rc;
	const unsigned long long len = T.len;
	DevBuf<unsigned char> &d_text = T.d_text;
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
	if (len == 0) return empty_result();
	HIP_TRY(hipEventRecord(c->ev1, st));
	if ((rc = scan_newlines(c, T))) return rc;
	const unsigned long long n_nl = T.n_nl;
	if (n_nl < 1 + has_header) return empty_result();   // header only (or no terminated line at all)
	const unsigned long long n_lines = n_nl - has_header;
	if (first_line + n_lines > 0xFFFFFFFFull) return fail(LSQ_E_RANGE, "more than 2^32 lines");
	DevBuf<unsigned long long> &d_nl_pos = T.d_nl_pos;
	const unsigned long long n_wg = (n_lines + 255) / 256;
	DevBuf<unsigned> d_line_nb, d_wg_reads, d_wg_blocks;
	DevBuf<unsigned long long> d_rd_base, d_bk_base, d_err;
	if ((rc = d_line_nb.alloc(n_lines)) || (rc = d_wg_reads.alloc(n_wg)) || (rc = d_wg_blocks.alloc(n_wg)) ||
	    (rc = d_rd_base.alloc(n_wg + 1)) || (rc = d_bk_base.alloc(n_wg + 1)) || (rc = d_err.alloc(4))) return rc;
	unsigned long long err[4] = {MRF_NO_ERR, 0, 0, 0};
	HIP_TRY(hipMemcpyAsync(d_err.p, err, sizeof(err), hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(lsq_mrf_count_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, d_text.p, d_nl_pos.p, n_lines, has_header, first_line,
	                   d_line_nb.p, d_wg_reads.p, d_wg_blocks.p, d_err.p);
	HIP_TRY(hipGetLastError());
	hipLaunchKernelGGL(lsq_scan_u32_kernel<1>, dim3(1), dim3(1024), 0, st, d_wg_reads.p, n_wg, d_rd_base.p);
	hipLaunchKernelGGL(lsq_scan_u32_kernel<1>, dim3(1), dim3(1024), 0, st, d_wg_blocks.p, n_wg, d_bk_base.p);
	HIP_TRY(hipGetLastError());
	unsigned long long n_reads = 0, n_blocks = 0;
	HIP_TRY(hipMemcpyAsync(err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&n_reads, d_rd_base.p + n_wg, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&n_blocks, d_bk_base.p + n_wg, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (err[0] != MRF_NO_ERR) {
		const unsigned long long ei = err[0] - first_line + has_header;      // newline that ends the failing line
		unsigned long long ab[2] = {~0ull, 0};                                  // ab[0] + 1 = first byte of the line
		if (ei > 0) HIP_TRY(hipMemcpy(&ab[0], d_nl_pos.p + (ei - 1), 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(&ab[1], d_nl_pos.p + ei, 8, hipMemcpyDeviceToHost));
		std::string text((size_t)(ab[1] - ab[0] - 1), '\0');
		size_t got_all = 0;
		const int fd = open(T.path.c_str(), O_RDONLY);
		while (fd >= 0 && got_all < text.size()) {
			const ssize_t got = pread(fd, &text[got_all], text.size() - got_all, (off_t)(T.offset + ab[0] + 1 + got_all));
			if (got <= 0) break;
			got_all += (size_t)got;
		}
		if (fd >= 0) close(fd);
		return fail(LSQ_E_PARSE, "#%llu:%s", err[0], text.c_str());
	}
	// dictionaries
	const size_t nc = E.covered.size();
	size_t tab = 2;
	while (tab < 4 * nc) tab <<= 1;
	std::vector<unsigned long long> h_hash(tab, 0);
	std::vector<unsigned> h_id(tab, 0), h_off(nc + 1, 0);
	std::string h_names;
	for (size_t id = 0; id < nc; ++id) {
		const std::string &nm = E.chroms.names[id];
		unsigned long long h = 0xcbf29ce484222325ull;
		for (unsigned char ch : nm) { h ^= ch; h *= 0x100000001b3ull; }
		if (h == 0) h = 1;
		size_t i = (size_t)((unsigned)h & (unsigned)(tab - 1));
		while (h_hash[i] != 0) i = (i + 1) & (tab - 1);
		h_hash[i] = h; h_id[i] = (unsigned)id;
		h_names += nm;
		h_off[id + 1] = (unsigned)h_names.size();
	}
	if (E.strands.names.size() > 256) return fail(LSQ_E_RANGE, "more than 256 distinct strand strings");
	const size_t n_seed = E.strands.names.size();
	std::vector<unsigned long long> h_strand(256, STRAND_EMPTY);
	for (size_t i = 0; i < n_seed; ++i) {
		const std::string &s = E.strands.names[i];
		h_strand[i] = s.size() <= 7 ? mrf_strand_key(s.data(), s.size()) : STRAND_UNMATCHABLE;
	}
	DevBuf<unsigned long long> d_hash, d_strand;
	DevBuf<unsigned> d_id, d_off;
	DevBuf<char> d_names;
	if ((rc = d_hash.upload(h_hash.data(), tab, st)) || (rc = d_id.upload(h_id.data(), tab, st)) || (rc = d_off.upload(h_off.data(), nc + 1, st)) ||
	    (rc = d_names.upload(h_names.data(), h_names.size(), st)) || (rc = d_strand.upload(h_strand.data(), 256, st))) return rc;
	if ((rc = out.blk_off.alloc(n_reads + 1)) || (rc = out.line_no.alloc(n_reads)) || (rc = out.bs.alloc(n_blocks)) || (rc = out.be.alloc(n_blocks)) ||
	    (rc = out.bc.alloc(n_blocks)) || (rc = out.bst.alloc(n_blocks))) return rc;
	MrfDict D{};
	D.chrom_hash = d_hash.p; D.chrom_id = d_id.p; D.name_off = d_off.p; D.names = d_names.p; D.mask = (unsigned)(tab - 1); D.strand_tab = d_strand.p;
	MrfOut O{};
	O.blk_off = out.blk_off.p; O.line_no = out.line_no.p; O.blk_start = out.bs.p; O.blk_end = out.be.p; O.blk_chrom = out.bc.p; O.blk_strand = out.bst.p;
	hipLaunchKernelGGL(lsq_mrf_write_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, d_text.p, d_nl_pos.p, n_lines, has_header, first_line, d_line_nb.p,
	                   d_rd_base.p, d_bk_base.p, D, O, d_err.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(c->ev2, st));
	HIP_TRY(hipMemcpyAsync(err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(h_strand.data(), d_strand.p, 256 * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	SW.mark("parse: kernels");
	if (parse_ms) (void)hipEventElapsedTime(parse_ms, c->ev1, c->ev2);
	if (err[1]) return fail(LSQ_E_UNSUPPORTED, "a strand string longer than 7 bytes: outside the device parser's range (lsq_mrf_parse handles it)");
	if (err[2]) return fail(LSQ_E_RANGE, "more than 256 distinct strand strings");
	for (size_t i = n_seed; i < 256 && h_strand[i] != STRAND_EMPTY; ++i) {
		const unsigned long long k = h_strand[i];
		std::string s;
		for (unsigned j = 0; j < (unsigned)(k & 0xFF); ++j) s.push_back((char)(k >> (56 - 8 * j)));
		const int id = E.strands.intern(s);
		if (id != (int)i) return fail(LSQ_E_STATE, "strand dictionary changed while a reads file was being parsed");
	}
	out.n_reads = n_reads; out.n_blocks = n_blocks;
	return LSQ_OK;
}

// open -> format literal -> copy -> parse: the order in which the reference meets a bad file or literal
static int device_parse_mrf(lsq_ctx *c, const char *read_format, const char *path, DevParsed &out, float *h2d_ms, float *parse_ms) {
	if (!read_format || !path) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	{
		FILE *f = fopen(path, "rb");
		if (!f) return fail(LSQ_E_IO, "cannot open reads file %s", path);
		fclose(f);
	}
	if (strcmp(read_format, "MRF_SINGLE") != 0) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", read_format);
	lsq_text T;
	int rc = stage_text_file(c, path, 0, ~0ull, T);
	if (rc) return rc;
	return parse_staged_text(c, read_format, T, 1u, 1ull, out, h2d_ms, parse_ms);
}
