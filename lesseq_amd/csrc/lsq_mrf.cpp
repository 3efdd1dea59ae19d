// Host side of the path, part 2: the MRF_SINGLE reader (count/count.cpp:279-336 ==
// solve/solve.cpp:429-486, minus the containment filter which runs at ingest).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <thread>
#include <unordered_map>

#include "lsq_internal.hpp"
#include "lsq_mrf_line.hpp"

using namespace lsq;

void lsq_reads::adopt() {
	n_reads = o_line_no.size();
	n_blocks = o_start.size();
	blk_off = o_blk_off.data();
	line_no = o_line_no.data();
	blk_start = o_start.data();
	blk_end = o_end.data();
	blk_chrom = o_chrom.data();
	blk_strand = o_strand.data();
}

namespace {

const uint16_t NOCHROM = 0xFFFF;

struct Chunk {
	const char *begin, *end;        // whole lines
	uint64_t first_line;            // line number of the first line (1-based after the header)
	std::vector<uint64_t> blk_cnt;  // per read
	std::vector<uint32_t> line_no;
	std::vector<int32_t> bs, be;
	std::vector<uint16_t> bc;
	std::vector<uint8_t> bst;
	int status = LSQ_OK;
	std::string err;
};

struct StrandCache { std::string s; int id = -1; };

void parse_chunk(Chunk &ck, lsq_events *E) {
	const int64_t LIM = (int64_t)1 << 30;
	std::string last_chrom; int last_chrom_id = -2;
	StrandCache sc[2];
	uint64_t line_num = ck.first_line - 1;
	const char *p = ck.begin;
	while (p < ck.end) {
		const char *nl = (const char *)memchr(p, '\n', (size_t)(ck.end - p));
		if (!nl) break;
		MrfView line{p, (size_t)(nl - p)};
		p = nl + 1;
		++line_num;
		if (mrf_line_is_skipped(line)) continue;
		uint64_t nb = 0;
		int bad_status = LSQ_OK;
		const bool ok = mrf_split_line(line, [&](MrfView chr, MrfView strand, int64_t start, int64_t end) {
			// chromosome: only names the events know can ever pass the containment filter
			if (last_chrom_id == -2 || last_chrom.size() != chr.n || memcmp(last_chrom.data(), chr.p, chr.n) != 0) {
				last_chrom.assign(chr.p, chr.n);
				int id = E->chroms.find(last_chrom);
				last_chrom_id = (id < 0 || (size_t)id >= E->covered.size()) ? (int)NOCHROM : id;
			}
			int sid = -1;
			for (auto &c : sc) if (c.id >= 0 && c.s.size() == strand.n && memcmp(c.s.data(), strand.p, strand.n) == 0) { sid = c.id; break; }
			if (sid < 0) {
				std::string st(strand.p, strand.n);
				sid = E->strands.intern(st);
				if (sid > 255) { bad_status = LSQ_E_RANGE; sid = 0; }
				sc[1] = sc[0]; sc[0].s = st; sc[0].id = sid;
			}
			int64_t s0 = start - 1;
			uint16_t cid = (uint16_t)last_chrom_id;
			if (s0 <= -LIM || end >= LIM || s0 >= LIM || end <= -LIM) { cid = NOCHROM; s0 = 0; end = 0; }
			ck.bs.push_back((int32_t)s0);
			ck.be.push_back((int32_t)end);
			ck.bc.push_back(cid);
			ck.bst.push_back((uint8_t)sid);
			++nb;
		});
		if (bad_status != LSQ_OK) { ck.status = bad_status; ck.err = "more than 256 distinct strand strings"; return; }
		if (!ok) {
			ck.status = LSQ_E_PARSE;
			ck.err = "#" + std::to_string(line_num) + ":" + std::string(line.p, line.n);
			return;
		}
		if (line_num > 0xFFFFFFFFull) { ck.status = LSQ_E_RANGE; ck.err = "more than 2^32 lines"; return; }
		ck.blk_cnt.push_back(nb);
		ck.line_no.push_back((uint32_t)line_num);
	}
}

} // namespace

extern "C" {

int lsq_mrf_parse(const char *read_format, const char *path, lsq_events *E, int n_threads, lsq_reads **out) LSQ_API_TRY {
	if (!read_format || !path || !E || !out) return fail(LSQ_E_ARG, "null argument");
	int fd = open(path, O_RDONLY);
	if (fd < 0) return fail(LSQ_E_IO, "cannot open reads file %s", path);
	if (strcmp(read_format, "MRF_SINGLE") != 0) { close(fd); return fail(LSQ_E_FORMAT, "Unknown file format error: %s", read_format); }
	struct stat st;
	if (fstat(fd, &st) != 0) { close(fd); return fail(LSQ_E_IO, "cannot stat %s", path); }
	size_t len = (size_t)st.st_size;
	std::unique_ptr<lsq_reads> R(new lsq_reads);
	R->o_blk_off.push_back(0);
	const char *data = nullptr;
	struct Unmapper { const char *&d; size_t n; ~Unmapper() { if (d) munmap((void *)d, n); } } unmapper{data, len};     // on every way out
	if (len > 0) {
		data = (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
		if (data == MAP_FAILED) { data = nullptr; close(fd); return fail(LSQ_E_IO, "cannot map %s", path); }
		madvise((void *)data, len, MADV_SEQUENTIAL);
	}
	close(fd);
	// first line skipped unconditionally (count/count.cpp:283); an unterminated tail is never seen (:285)
	const char *body = data ? (const char *)memchr(data, '\n', len) : nullptr;
	const char *tail = nullptr;
	if (body) {
		++body;
		const char *e = data + len;
		while (e > body && e[-1] != '\n') --e;
		tail = e;
	}
	if (body && tail > body) {
		int T = host_threads(n_threads);
		size_t total = (size_t)(tail - body);
		if (total < (1u << 20)) T = 1;
		std::vector<Chunk> chunks(T);
		const char *cur = body;
		for (int t = 0; t < T; ++t) {
			const char *stop = (t == T - 1) ? tail : body + total * (size_t)(t + 1) / (size_t)T;
			if (stop < cur) stop = cur;
			if (stop < tail) {
				const char *nl = (const char *)memchr(stop, '\n', (size_t)(tail - stop));
				stop = nl ? nl + 1 : tail;
			}
			chunks[t].begin = cur; chunks[t].end = stop;
			cur = stop;
		}
		// line numbers: count newlines per chunk, prefix
		std::vector<uint64_t> nlines(T, 0);
		{
			ThreadGroup th;
			for (int t = 0; t < T; ++t) th.spawn([&, t] {
				uint64_t c = 0;
				const char *q = chunks[t].begin;
				while (q < chunks[t].end) {
					const char *nl = (const char *)memchr(q, '\n', (size_t)(chunks[t].end - q));
					if (!nl) break;
					++c; q = nl + 1;
				}
				nlines[t] = c;
			});
			th.join();
			if (th.failed()) return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
		}
		uint64_t ln = 1;
		for (int t = 0; t < T; ++t) { chunks[t].first_line = ln; ln += nlines[t]; }
		{
			ThreadGroup th;
			for (int t = 0; t < T; ++t) th.spawn([&, t] { parse_chunk(chunks[t], E); });
			th.join();
			if (th.failed()) return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
		}
		for (int t = 0; t < T; ++t) if (chunks[t].status != LSQ_OK) {
			int st2 = chunks[t].status;
			std::string msg = chunks[t].err;
			return fail(st2, "%s", msg.c_str());
		}
		size_t nr = 0, nbk = 0;
		for (auto &c : chunks) { nr += c.line_no.size(); nbk += c.bs.size(); }
		R->o_blk_off.reserve(nr + 1); R->o_line_no.reserve(nr);
		R->o_start.reserve(nbk); R->o_end.reserve(nbk); R->o_chrom.reserve(nbk); R->o_strand.reserve(nbk);
		uint64_t off = 0;
		for (auto &c : chunks) {
			for (uint64_t n : c.blk_cnt) { off += n; R->o_blk_off.push_back(off); }
			R->o_line_no.insert(R->o_line_no.end(), c.line_no.begin(), c.line_no.end());
			R->o_start.insert(R->o_start.end(), c.bs.begin(), c.bs.end());
			R->o_end.insert(R->o_end.end(), c.be.begin(), c.be.end());
			R->o_chrom.insert(R->o_chrom.end(), c.bc.begin(), c.bc.end());
			R->o_strand.insert(R->o_strand.end(), c.bst.begin(), c.bst.end());
			Chunk().blk_cnt.swap(c.blk_cnt);
			std::vector<int32_t>().swap(c.bs); std::vector<int32_t>().swap(c.be);
		}
	}
	R->adopt();
	*out = R.release();
	return LSQ_OK;
} LSQ_API_CATCH

// solve's name-keyed read formats (solve/solve.cpp:413-428 UCSC_GFF, :487-551 UCSC_BED, :552-634
// WORMBASE_GFF3).  Every accepted line adds its blocks to the read of its name, in file order; the
// device ingest then merges them with add_interval and takes chromosome and strand from the last kept
// block, which is the last accepted line.  UCSC_BED and WORMBASE_GFF3 accept or drop a line by its
// whole span, so that test runs here against the covered regions; UCSC_GFF tests the block itself, as
// the ingest kernel does anyway.
int lsq_reads_parse(const char *read_format, const char *path, lsq_events *E, int n_threads, lsq_reads **out) LSQ_API_TRY {
	if (!read_format || !path || !E || !out) return fail(LSQ_E_ARG, "null argument");
	const std::string fmt = read_format;
	if (fmt == "MRF_SINGLE") return lsq_mrf_parse(read_format, path, E, n_threads, out);
	FILE *fp = fopen(path, "rb");
	if (!fp) return fail(LSQ_E_IO, "cannot open reads file %s", path);
	if (fmt != "UCSC_GFF" && fmt != "UCSC_BED" && fmt != "WORMBASE_GFF3") { fclose(fp); return fail(LSQ_E_FORMAT, "Unknown file format error: %s", read_format); }
	std::string data;
	{
		char buf[1 << 16];
		size_t n;
		while ((n = fread(buf, 1, sizeof buf, fp)) > 0) data.append(buf, n);
		fclose(fp);
	}
	struct Line { uint32_t read; int32_t s[2], e[2]; uint16_t chrom; uint8_t strand, n; };
	std::vector<Line> accepted;
	std::unordered_map<std::string, uint32_t> by_name;
	std::vector<std::string> names;
	const int64_t LIM = (int64_t)1 << 30;
	auto chrom_of = [&](const std::string &c) -> int { int id = E->chroms.find(c); return (id < 0 || (size_t)id >= E->covered.size()) ? -1 : id; };
	auto covered = [&](int chrom, int64_t s, int64_t e) -> bool {
		if (!(s < e)) return true;                       // contains_interval of an empty interval (interval_list.hpp:396-422)
		return chrom >= 0 && E->covered[chrom].contains(s, e);
	};
	auto strand_of = [&](const std::string &st, int &sid) -> bool { sid = E->strands.intern(st); return sid <= 255; };
	auto push = [&](const std::string &name, int chrom, int sid, int nblk, const int64_t *bs, const int64_t *be) {
		auto it = by_name.find(name);
		uint32_t r;
		if (it == by_name.end()) { r = (uint32_t)names.size(); by_name.emplace(name, r); names.push_back(name); } else r = it->second;
		for (int k = 0; k < nblk; k += 2) {
			Line l; l.read = r; l.strand = (uint8_t)sid; l.n = (uint8_t)std::min(2, nblk - k);
			l.chrom = chrom < 0 ? NOCHROM : (uint16_t)chrom;
			for (int q = 0; q < l.n; ++q) {
				int64_t s = bs[k + q], e = be[k + q];
				if (s <= -LIM || e >= LIM || s >= LIM || e <= -LIM) { l.chrom = NOCHROM; s = 0; e = 0; }
				l.s[q] = (int32_t)s; l.e[q] = (int32_t)e;
			}
			accepted.push_back(l);
		}
		if (nblk == 0) { Line l; l.read = r; l.strand = (uint8_t)sid; l.n = 0; l.chrom = NOCHROM; l.s[0] = l.e[0] = l.s[1] = l.e[1] = 0; accepted.push_back(l); }
	};
	auto field = [](const std::string &ln, int k, size_t &b, size_t &e) -> bool {       // tab field k as a find('\t') chain sees it
		size_t pos = 0;
		for (int i = 0; i < k; ++i) { size_t t = ln.find('\t', pos); if (t == std::string::npos) return false; pos = t + 1; }
		size_t t = ln.find('\t', pos);
		b = pos; e = t == std::string::npos ? ln.size() : t;
		return true;
	};
	auto cast_field = [&](const std::string &ln, int k, int64_t &v) -> bool {
		size_t b, e;
		if (!field(ln, k, b, e)) return false;
		return mrf_cast_long(MrfView{ln.data() + b, e - b}, v);
	};
	size_t pos = 0, line_no = 0;
	const size_t skip = fmt == "UCSC_GFF" ? 2 : (fmt == "UCSC_BED" ? 1 : 0);
	int status = LSQ_OK;
	std::string err;
	while (pos < data.size() && status == LSQ_OK) {
		size_t nl = data.find('\n', pos);
		if (nl == std::string::npos) break;                  // an unterminated last line is never seen
		const std::string ln(data, pos, nl - pos);
		pos = nl + 1;
		if (line_no++ < skip) continue;
		int sid = 0;
		if (fmt == "UCSC_GFF") {
			std::istringstream iss(ln);
			long start = 0, end = 0;
			std::string rname, chr, tmp, strand;
			iss >> chr >> tmp >> tmp >> start >> end >> tmp >> strand >> tmp >> rname;
			if (iss.fail()) { status = LSQ_E_ARG; err = "line " + std::to_string(line_no) + " does not have the UCSC_GFF columns (the reference reads uninitialised coordinates here)"; break; }
			const int chrom = chrom_of(chr);
			if (!covered(chrom, (int64_t)start - 1, end)) continue;
			if (!strand_of(strand, sid)) { status = LSQ_E_RANGE; err = "more than 256 distinct strand strings"; break; }
			const int64_t bs[1] = {(int64_t)start - 1}, be[1] = {end};
			push(rname, chrom, sid, 1, bs, be);
		} else if (fmt == "UCSC_BED") {
			int64_t start, end, nb;
			size_t b, e;
			if (!cast_field(ln, 1, start) || !cast_field(ln, 2, end)) { status = LSQ_E_PARSE; err = "#" + std::to_string(line_no) + ":" + ln; break; }
			std::string chr = field(ln, 0, b, e) ? ln.substr(b, e - b) : std::string();
			const int chrom = chrom_of(chr);
			if (!covered(chrom, start, end)) continue;
			if (!cast_field(ln, 9, nb)) { status = LSQ_E_PARSE; err = "#" + std::to_string(line_no) + ":" + ln; break; }
			std::string rname = field(ln, 3, b, e) ? ln.substr(b, e - b) : std::string();
			std::string strand = field(ln, 5, b, e) ? ln.substr(b, e - b) : std::string();
			std::string sizes = field(ln, 10, b, e) ? ln.substr(b, e - b) : std::string();
			std::string starts = field(ln, 11, b, e) ? ln.substr(b, e - b) : std::string();
			if (!strand_of(strand, sid)) { status = LSQ_E_RANGE; err = "more than 256 distinct strand strings"; break; }
			std::vector<int64_t> bs, be;
			size_t ps = 0, pz = 0;
			for (int64_t i = 0; i < nb; ++i) {
				size_t qs = starts.find(',', ps), qz = sizes.find(',', pz);
				if (qs == std::string::npos) qs = starts.size();
				if (qz == std::string::npos) qz = sizes.size();
				int64_t istart, isize;
				if (ps > starts.size() || pz > sizes.size() || !mrf_cast_long(MrfView{starts.data() + ps, qs - ps}, istart) || !mrf_cast_long(MrfView{sizes.data() + pz, qz - pz}, isize)) {
					status = LSQ_E_PARSE; err = "#" + std::to_string(line_no) + ":" + ln; break;
				}
				bs.push_back(start + istart); be.push_back(start + istart + isize);
				ps = qs + 1; pz = qz + 1;
			}
			if (status != LSQ_OK) break;
			push(rname, chrom, sid, (int)bs.size(), bs.data(), be.data());
		} else {
			int64_t start, end;
			size_t b, e;
			if (!cast_field(ln, 3, start) || !cast_field(ln, 4, end)) { status = LSQ_E_PARSE; err = "#" + std::to_string(line_no) + ":" + ln; break; }
			std::string chr = "chr" + (field(ln, 0, b, e) ? ln.substr(b, e - b) : std::string());
			std::string strand = field(ln, 6, b, e) ? ln.substr(b, e - b) : std::string();
			std::string rinfo = field(ln, 8, b, e) ? ln.substr(b, e - b) : std::string();
			std::string rname;
			bool found_parent = false;
			int64_t start2 = 0, end2 = 0;
			size_t a0 = 0;
			for (;;) {                                           // attributes closed by ';' only
				size_t a1 = rinfo.find(';', a0);
				if (a1 == std::string::npos) break;
				const std::string at = rinfo.substr(a0, a1 - a0);
				if (at.compare(0, 7, "Target=") == 0) { const std::string v = at.substr(7); rname = v.substr(0, v.find(' ')); }
				else if (at.compare(0, 7, "Parent=") == 0) {
					const std::string v = at.substr(7);
					if (v.compare(0, 7, "intron_") == 0) {
						found_parent = true;
						size_t u0 = v.find('_', 7), u1 = u0 == std::string::npos ? u0 : v.find('_', u0 + 1);
						size_t u2 = u1 == std::string::npos ? u1 : v.find('_', u1 + 1);
						bool ok = u0 != std::string::npos && u1 != std::string::npos;
						if (ok) {
							const std::string t1 = v.substr(u0 + 1, u1 - u0 - 1), t2 = v.substr(u1 + 1, u2 == std::string::npos ? std::string::npos : u2 - u1 - 1);
							ok = !t1.empty() && !t2.empty() && t1[0] != '-' && t2[0] != '-' && mrf_cast_long(MrfView{t1.data(), t1.size()}, start2) && mrf_cast_long(MrfView{t2.data(), t2.size()}, end2);
						}
						if (!ok) { status = LSQ_E_PARSE; err = "#" + std::to_string(line_no) + ":" + ln; break; }
					}
				}
				a0 = a1 + 1;
			}
			if (status != LSQ_OK) break;
			const int chrom = chrom_of(chr);
			if (!covered(chrom, start - 1, end)) continue;
			if (!strand_of(strand, sid)) { status = LSQ_E_RANGE; err = "more than 256 distinct strand strings"; break; }
			if (!found_parent) { const int64_t bs[1] = {start - 1}, be[1] = {end}; push(rname, chrom, sid, 1, bs, be); }
			else { const int64_t bs[2] = {start - 1, end2}, be[2] = {start2 - 1, end}; push(rname, chrom, sid, 2, bs, be); }
		}
	}
	if (status != LSQ_OK) return fail(status, "%s", err.c_str());
	// reads in order of first appearance; the blocks of a read in file order
	std::unique_ptr<lsq_reads> R(new lsq_reads);
	const size_t nr = names.size();
	std::vector<uint64_t> cnt(nr + 1, 0);
	for (const Line &l : accepted) cnt[l.read + 1] += l.n;
	R->o_blk_off.assign(nr + 1, 0);
	for (size_t r = 0; r < nr; ++r) R->o_blk_off[r + 1] = R->o_blk_off[r] + cnt[r + 1];
	const uint64_t nb = R->o_blk_off[nr];
	R->o_start.resize(nb); R->o_end.resize(nb); R->o_chrom.resize(nb); R->o_strand.resize(nb);
	R->o_line_no.resize(nr);
	std::vector<uint64_t> cur(R->o_blk_off.begin(), R->o_blk_off.end() - 1);
	for (const Line &l : accepted)
		for (int q = 0; q < l.n; ++q) {
			const uint64_t w = cur[l.read]++;
			R->o_start[w] = l.s[q]; R->o_end[w] = l.e[q]; R->o_chrom[w] = l.chrom; R->o_strand[w] = l.strand;
		}
	if (nr > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "more than 2^32 read names");
	R->named = true;
	R->name_off.assign(nr + 1, 0);
	for (size_t r = 0; r < nr; ++r) { R->o_line_no[r] = (uint32_t)r; R->name_blob += names[r]; R->name_off[r + 1] = R->name_blob.size(); }
	R->adopt();
	*out = R.release();
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_reads_wrap(uint64_t n_reads, const uint64_t *blk_off, const uint32_t *line_no,
                   const int32_t *blk_start, const int32_t *blk_end,
                   const uint16_t *blk_chrom_id, const uint8_t *blk_strand_id, lsq_reads **out) LSQ_API_TRY {
	if (!out || (n_reads && (!blk_off || !line_no || !blk_start || !blk_end || !blk_chrom_id || !blk_strand_id)))
		return fail(LSQ_E_ARG, "null array");
	std::unique_ptr<lsq_reads> R(new lsq_reads);
	R->n_reads = n_reads;
	R->n_blocks = n_reads ? blk_off[n_reads] : 0;
	R->blk_off = blk_off; R->line_no = line_no;
	R->blk_start = blk_start; R->blk_end = blk_end;
	R->blk_chrom = blk_chrom_id; R->blk_strand = blk_strand_id;
	*out = R.release();
	return LSQ_OK;
} LSQ_API_CATCH
int lsq_reads_arrays(const lsq_reads *r, const uint64_t **blk_off, const uint32_t **line_no, const int32_t **blk_start,
                     const int32_t **blk_end, const uint16_t **blk_chrom_id, const uint8_t **blk_strand_id) LSQ_API_TRY {
	if (!r) return fail(LSQ_E_ARG, "null read set");
	if (blk_off) *blk_off = r->blk_off;
	if (line_no) *line_no = r->line_no;
	if (blk_start) *blk_start = r->blk_start;
	if (blk_end) *blk_end = r->blk_end;
	if (blk_chrom_id) *blk_chrom_id = r->blk_chrom;
	if (blk_strand_id) *blk_strand_id = r->blk_strand;
	return LSQ_OK;
} LSQ_API_CATCH
void lsq_reads_free(lsq_reads *r) { delete r; }
uint64_t lsq_reads_count(const lsq_reads *r) { return r ? r->n_reads : 0; }
uint64_t lsq_reads_num_blocks(const lsq_reads *r) { return r ? r->n_blocks : 0; }

} // extern "C"
