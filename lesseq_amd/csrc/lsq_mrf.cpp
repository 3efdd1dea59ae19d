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
#include <thread>

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

int lsq_mrf_parse(const char *read_format, const char *path, lsq_events *E, int n_threads, lsq_reads **out) {
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
	if (len > 0) {
		data = (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
		if (data == MAP_FAILED) { close(fd); return fail(LSQ_E_IO, "cannot map %s", path); }
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
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
				uint64_t c = 0;
				const char *q = chunks[t].begin;
				while (q < chunks[t].end) {
					const char *nl = (const char *)memchr(q, '\n', (size_t)(chunks[t].end - q));
					if (!nl) break;
					++c; q = nl + 1;
				}
				nlines[t] = c;
			});
			for (auto &x : th) x.join();
		}
		uint64_t ln = 1;
		for (int t = 0; t < T; ++t) { chunks[t].first_line = ln; ln += nlines[t]; }
		{
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&, t] { parse_chunk(chunks[t], E); });
			for (auto &x : th) x.join();
		}
		for (int t = 0; t < T; ++t) if (chunks[t].status != LSQ_OK) {
			int st2 = chunks[t].status;
			std::string msg = chunks[t].err;
			munmap((void *)data, len);
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
	if (data) munmap((void *)data, len);
	R->adopt();
	*out = R.release();
	return LSQ_OK;
}

int lsq_reads_wrap(uint64_t n_reads, const uint64_t *blk_off, const uint32_t *line_no,
                   const int32_t *blk_start, const int32_t *blk_end,
                   const uint16_t *blk_chrom_id, const uint8_t *blk_strand_id, lsq_reads **out) {
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
}
int lsq_reads_arrays(const lsq_reads *r, const uint64_t **blk_off, const uint32_t **line_no, const int32_t **blk_start,
                     const int32_t **blk_end, const uint16_t **blk_chrom_id, const uint8_t **blk_strand_id) {
	if (!r) return fail(LSQ_E_ARG, "null read set");
	if (blk_off) *blk_off = r->blk_off;
	if (line_no) *line_no = r->line_no;
	if (blk_start) *blk_start = r->blk_start;
	if (blk_end) *blk_end = r->blk_end;
	if (blk_chrom_id) *blk_chrom_id = r->blk_chrom;
	if (blk_strand_id) *blk_strand_id = r->blk_strand;
	return LSQ_OK;
}
void lsq_reads_free(lsq_reads *r) { delete r; }
uint64_t lsq_reads_count(const lsq_reads *r) { return r ? r->n_reads : 0; }
uint64_t lsq_reads_num_blocks(const lsq_reads *r) { return r ? r->n_blocks : 0; }

} // extern "C"
