// Synthetic workload of SURVEY.md 8(d): LESSeq-shaped local events (the eight types of the
// reference's bin/Events.r:62-156) and 100-bp-style single-end reads.  Deterministic in the
// spec: events come from one splitmix64 stream, read i from its own counter-based stream, so
// the text files and the in-memory read set agree and do not depend on the thread count.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "lsq_internal.hpp"

using namespace lsq;

namespace {

struct Rng {
	uint64_t s;
	explicit Rng(uint64_t seed) : s(seed) {}
	uint64_t next() {
		uint64_t z = (s += 0x9E3779B97F4A7C15ull);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
	// uniform integer in [lo, hi]
	int64_t range(int64_t lo, int64_t hi) { return lo + (int64_t)(next() % (uint64_t)(hi - lo + 1)); }
	double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

typedef std::vector<std::pair<int32_t, int32_t>> Exons;

struct SynEvent {
	int chrom;             // 0-based index, name chr<chrom+1>
	char strand;
	Exons form[2];
	int32_t gs, ge;
};

struct SynModel {
	std::vector<SynEvent> ev;
	std::vector<int32_t> chrom_end;
	std::vector<double> cum;       // cumulative event weights (zipf) or empty
};

void make_event(Rng &g, int type, int32_t a, int R, Exons &A, Exons &B, int32_t &end) {
	auto L = [&] { return (int32_t)g.range(std::max(60, R + 1), std::max(400, R + 1)); };
	auto I = [&] { return (int32_t)g.range(100, 5000); };
	auto mid = [&](int lo, int hi) { return (int32_t)g.range(lo, hi); };
	A.clear(); B.clear();
	switch (type) {
	case 0: { // SE
		int32_t e1s = a, e1e = a + L(); int32_t e2s = e1e + I(), e2e = e2s + mid(60, 400); int32_t e3s = e2e + I(), e3e = e3s + L();
		A = {{e1s, e1e}, {e2s, e2e}, {e3s, e3e}}; B = {{e1s, e1e}, {e3s, e3e}}; end = e3e; break; }
	case 1: { // RI
		int32_t e1s = a, e1e = a + L(); int32_t e2s = e1e + mid(80, 900), e2e = e2s + L();
		A = {{e1s, e2e}}; B = {{e1s, e1e}, {e2s, e2e}}; end = e2e; break; }
	case 2: case 3: { // A5SS / A3SS
		int32_t e1s = a, e1e = a + L(); int32_t ext = mid(20, 200), gap = I(); int32_t e3s = e1e + ext + gap, e3e = e3s + L();
		if (type == 2) A = {{e1s, e1e}, {e1e, e1e + ext}, {e3s, e3e}};
		else A = {{e1s, e1e}, {e3s - ext, e3s}, {e3s, e3e}};
		B = {{e1s, e1e}, {e3s, e3e}}; end = e3e; break; }
	case 4: { // MXE
		int32_t e1s = a, e1e = a + L(); int32_t b1s = e1e + I(), b1e = b1s + mid(60, 300); int32_t b2s = b1e + I(), b2e = b2s + mid(60, 300);
		int32_t e4s = b2e + I(), e4e = e4s + L();
		A = {{e1s, e1e}, {b1s, b1e}, {e4s, e4e}}; B = {{e1s, e1e}, {b2s, b2e}, {e4s, e4e}}; end = e4e; break; }
	case 5: case 6: { // AFE / ALE
		int32_t e1s = a, e1e = a + L(); int32_t e2s = e1e + I(), e2e = e2s + L(); int32_t e3s = e2e + I(), e3e = e3s + L();
		if (type == 5) { A = {{e2s, e2e}, {e3s, e3e}}; B = {{e1s, e1e}, {e3s, e3e}}; }
		else { A = {{e1s, e1e}, {e2s, e2e}}; B = {{e1s, e1e}, {e3s, e3e}}; }
		end = e3e; break; }
	default: { // T3
		int32_t ln = mid(2 * R + 50, std::max(1200, 2 * R + 60)); int32_t cut = mid(R + 5, ln - 20);
		A = {{a, a + ln}}; B = {{a, a + cut}}; end = a + ln; break; }
	}
}

int build_model(const lsq_synth_spec &S, SynModel &Mo) {
	if (S.n_chrom == 0 || S.n_chrom > 1000 || S.read_length < 20 || S.read_length > 4000) return fail(LSQ_E_ARG, "synthetic spec: n_chrom in 1..1000, read_length in 20..4000");
	uint32_t types = S.event_types & 0xFF;
	if (!types) types = 0xFF;
	std::vector<int> allowed;
	for (int t = 0; t < 8; ++t) if (types >> t & 1) allowed.push_back(t);
	Rng g(S.seed * 0x2545F4914F6CDD1Dull + 12345);
	const int C = (int)S.n_chrom, R = (int)S.read_length;
	// chromosomes weighted by index-decreasing size
	std::vector<double> cw(C);
	double tot = 0;
	for (int c = 0; c < C; ++c) { cw[c] = (double)(C - c) + 0.5 * C; tot += cw[c]; }
	std::vector<int32_t> pos(C), last_gs(C, -1), last_ge(C, -1);
	for (int c = 0; c < C; ++c) pos[c] = (int32_t)g.range(2000, 20000);
	Mo.ev.resize(S.n_events);
	for (uint64_t i = 0; i < S.n_events; ++i) {
		double u = g.unit() * tot;
		int c = 0;
		while (c < C - 1 && u >= cw[c]) { u -= cw[c]; ++c; }
		SynEvent &e = Mo.ev[i];
		e.chrom = c;
		e.strand = (g.next() & 1) ? '+' : '-';
		int type = allowed[g.next() % allowed.size()];
		int32_t start = pos[c];
		if (last_gs[c] >= 0 && g.unit() < S.overlap_frac) start = (int32_t)g.range(last_gs[c] + 1, std::max(last_gs[c] + 2, last_ge[c] - 1));
		int32_t end;
		make_event(g, type, start, R, e.form[0], e.form[1], end);
		if (end >= (1 << 30) - 100000) return fail(LSQ_E_RANGE, "synthetic chromosome grew past 2^30; use more chromosomes");
		e.gs = std::min(e.form[0].front().first, e.form[1].front().first);
		e.ge = std::max(e.form[0].back().second, e.form[1].back().second);
		last_gs[c] = e.gs; last_ge[c] = e.ge;
		pos[c] = std::max(pos[c], end) + (int32_t)g.range(2000, 20000);
	}
	Mo.chrom_end.assign(C, 0);
	for (auto &e : Mo.ev) Mo.chrom_end[e.chrom] = std::max(Mo.chrom_end[e.chrom], e.ge);
	Mo.cum.clear();
	if (S.zipf && S.n_events) {
		// depth rank is a fixed pseudo-random permutation of the events
		std::vector<uint32_t> perm(S.n_events);
		for (uint64_t i = 0; i < S.n_events; ++i) perm[i] = (uint32_t)i;
		Rng pg(S.seed ^ 0xA5A5A5A5ull);
		for (uint64_t i = S.n_events; i > 1; --i) std::swap(perm[i - 1], perm[pg.next() % i]);
		std::vector<double> w(S.n_events);
		for (uint64_t r = 0; r < S.n_events; ++r) w[perm[r]] = 1.0 / std::pow((double)(r + 1), 1.1);
		Mo.cum.resize(S.n_events);
		double acc = 0;
		for (uint64_t i = 0; i < S.n_events; ++i) { acc += w[i]; Mo.cum[i] = acc; }
	}
	return LSQ_OK;
}

void transcript_blocks(const Exons &ex, int64_t t0, int64_t len, int32_t *bs, int32_t *be, int &nb) {
	nb = 0;
	int64_t pos = 0, rem = len;
	for (auto &x : ex) {
		int64_t ln = x.second - x.first;
		if (t0 < pos + ln && rem > 0 && nb < 6) {
			int64_t off = std::max<int64_t>(t0 - pos, 0);
			int64_t take = std::min(ln - off, rem);
			bs[nb] = (int32_t)(x.first + off); be[nb] = (int32_t)(x.first + off + take); ++nb;
			rem -= take; t0 += take;
		}
		pos += ln;
	}
}

struct OneRead { int chrom; char strand; int nb; int32_t bs[6], be[6]; };

void gen_read(const lsq_synth_spec &S, const SynModel &Mo, uint64_t i, OneRead &r) {
	Rng g(S.seed * 0x9E3779B97F4A7C15ull + i * 0xD1B54A32D192ED03ull + 77);
	const int R = (int)S.read_length, C = (int)S.n_chrom;
	double u = g.unit();
	if (u >= 0.85 || Mo.ev.empty()) {
		r.chrom = (int)(g.next() % (uint64_t)C);
		r.strand = (g.next() & 1) ? '+' : '-';
		int32_t s = (int32_t)g.range(0, (int64_t)Mo.chrom_end[r.chrom] + 20000);
		r.nb = 1; r.bs[0] = s; r.be[0] = s + R;
		return;
	}
	size_t ei;
	if (!Mo.cum.empty()) {
		double t = g.unit() * Mo.cum.back();
		ei = std::lower_bound(Mo.cum.begin(), Mo.cum.end(), t) - Mo.cum.begin();
		if (ei >= Mo.ev.size()) ei = Mo.ev.size() - 1;
	} else ei = g.next() % Mo.ev.size();
	const SynEvent &e = Mo.ev[ei];
	const Exons &f = e.form[g.next() & 1];
	int64_t tlen = 0;
	for (auto &x : f) tlen += x.second - x.first;
	r.chrom = e.chrom;
	r.strand = g.unit() < 0.9 ? e.strand : (e.strand == '+' ? '-' : '+');
	double v = g.unit();
	if (v < 0.70) {
		int cand[4], nc = 0;
		for (size_t k = 0; k < f.size() && nc < 4; ++k) if (f[k].second - f[k].first >= R) cand[nc++] = (int)k;
		if (nc) {
			auto &x = f[cand[g.next() % (uint64_t)nc]];
			int32_t s = (int32_t)g.range(x.first, x.second - R);
			r.nb = 1; r.bs[0] = s; r.be[0] = s + R;
		} else {
			int64_t t0 = g.range(0, std::max<int64_t>(tlen - R, 0));
			transcript_blocks(f, t0, std::min<int64_t>(R, tlen), r.bs, r.be, r.nb);
		}
	} else if (v < 0.95) {
		if (f.size() >= 2 && tlen > R) {
			size_t j = g.next() % (f.size() - 1);
			int64_t before = 0;
			for (size_t k = 0; k <= j; ++k) before += f[k].second - f[k].first;
			int64_t o = g.range(1, R - 1);
			int64_t t0 = std::min(std::max<int64_t>(before - o, 0), tlen - R);
			transcript_blocks(f, t0, R, r.bs, r.be, r.nb);
		} else {
			int64_t t0 = g.range(0, std::max<int64_t>(tlen - R, 0));
			transcript_blocks(f, t0, std::min<int64_t>(R, tlen), r.bs, r.be, r.nb);
		}
	} else {
		int k = (int)(g.next() % 5);
		r.nb = 1;
		if (k == 0) { r.bs[0] = e.gs; r.be[0] = e.gs + R; }
		else if (k == 1) { auto &x = f[g.next() % f.size()]; int32_t o = (int32_t)g.range(1, 2); r.bs[0] = x.second - R + o; r.be[0] = x.second + o; }
		else if (k == 2) { int32_t m = (int32_t)g.range(1, 20); r.nb = 2; r.bs[0] = f.front().second - 20; r.be[0] = f.front().second; r.bs[1] = f.back().first + m; r.be[1] = f.back().first + m + R - 20; }
		else if (k == 3) { r.nb = 2; r.bs[0] = f.front().second - 20; r.be[0] = f.front().second; r.bs[1] = e.ge + 1000; r.be[1] = e.ge + 1000 + R - 20; }
		else { r.bs[0] = e.ge - R; r.be[0] = e.ge; }
	}
}

std::string chrom_name(int c) { return "chr" + std::to_string(c + 1); }

// spec.sorted: the reads' numbers in coordinate order -- chromosome, first base of the first block, number.  A counting sort on
// (chromosome, first base / 65 536) by all threads, then every bin sorted on its own.
int sorted_order(const lsq_synth_spec &S, const SynModel &Mo, std::vector<uint32_t> &order) {
	const uint64_t n = S.n_reads;
	if (n > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "more than 2^32 reads in one read set");
	const int T = n < 100000 ? 1 : host_threads(0);
	std::vector<uint64_t> key(n);              // chromosome << 40 | first base + 2^31 (reads may start left of base 0)
	auto range = [&](int t, uint64_t &a, uint64_t &b) { a = n * (uint64_t)t / (uint64_t)T; b = n * (uint64_t)(t + 1) / (uint64_t)T; };
	const uint64_t n_bins = (uint64_t)S.n_chrom << 16;       // chromosome, bits 31..16 of the biased base
	auto bin_of = [](uint64_t k) { return (size_t)(((k >> 40) << 16) | ((k >> 16) & 0xFFFFu)); };
	std::vector<std::vector<uint32_t>> cnt((size_t)T);
	{
		ThreadGroup th;
		for (int t = 0; t < T; ++t) th.spawn([&, t] {
			uint64_t a, b; range(t, a, b);
			cnt[(size_t)t].assign(n_bins, 0);
			OneRead r;
			for (uint64_t i = a; i < b; ++i) {
				gen_read(S, Mo, S.first_read + i, r);
				key[i] = ((uint64_t)r.chrom << 40) | (uint64_t)((int64_t)r.bs[0] + (1ll << 31));
				++cnt[(size_t)t][bin_of(key[i])];
			}
		});
		th.join();
		if (th.failed()) return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
	}
	std::vector<uint64_t> bin_first(n_bins + 1, 0);
	{
		uint64_t run = 0;
		for (uint64_t bq = 0; bq < n_bins; ++bq) {
			bin_first[bq] = run;
			for (int t = 0; t < T; ++t) { const uint32_t c = cnt[(size_t)t][bq]; cnt[(size_t)t][bq] = (uint32_t)(run - bin_first[bq]); run += c; }
		}
		bin_first[n_bins] = run;
	}
	order.resize(n);
	{
		ThreadGroup th;
		for (int t = 0; t < T; ++t) th.spawn([&, t] {
			uint64_t a, b; range(t, a, b);
			for (uint64_t i = a; i < b; ++i) { const size_t bq = bin_of(key[i]); order[bin_first[bq] + cnt[(size_t)t][bq]++] = (uint32_t)i; }
		});
		th.join();
		if (th.failed()) return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
	}
	{
		ThreadGroup th;
		for (int t = 0; t < T; ++t) th.spawn([&, t] {
			for (uint64_t bq = (uint64_t)t; bq < n_bins; bq += (uint64_t)T)
				std::sort(order.begin() + (ptrdiff_t)bin_first[bq], order.begin() + (ptrdiff_t)bin_first[bq + 1],
				          [&](uint32_t x, uint32_t y) { return key[x] != key[y] ? key[x] < key[y] : x < y; });
		});
		th.join();
		if (th.failed()) return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
	}
	return LSQ_OK;
}

} // namespace

extern "C" {

int lsq_synth_write(const lsq_synth_spec *S, const char *dir, const char *stem, int write_mrf) LSQ_API_TRY {
	if (!S || !dir || !stem) return fail(LSQ_E_ARG, "null argument");
	SynModel Mo;
	int rc = build_model(*S, Mo);
	if (rc) return rc;
	std::string base = std::string(dir) + "/" + stem;
	FILE *fi = fopen((base + ".interval").c_str(), "w");
	FILE *fm = fopen((base + ".map").c_str(), "w");
	if (!fi || !fm) { if (fi) fclose(fi); if (fm) fclose(fm); return fail(LSQ_E_IO, "cannot write under %s", dir); }
	for (size_t i = 0; i < Mo.ev.size(); ++i) {
		const SynEvent &e = Mo.ev[i];
		for (int k = 0; k < 2; ++k) {
			const Exons &f = e.form[k];
			fprintf(fi, "%zu.%c\t%s\t%c\t%d\t%d\t%zu\t", i + 1, "ab"[k], chrom_name(e.chrom).c_str(), e.strand, f.front().first, f.back().second, f.size());
			for (auto &x : f) fprintf(fi, "%d,", x.first);
			fputc('\t', fi);
			for (auto &x : f) fprintf(fi, "%d,", x.second);
			fputc('\n', fi);
			fprintf(fm, "%zu\t%zu.%c\n", i + 1, i + 1, "ab"[k]);
		}
	}
	fclose(fi); fclose(fm);
	if (write_mrf) {
		FILE *fr = fopen((base + ".mrf").c_str(), "w");
		if (!fr) return fail(LSQ_E_IO, "cannot write under %s", dir);
		fputs("AlignmentBlocks\n", fr);
		// reads are a function of their number (counter-based generator): rounds of T chunks formatted by T threads,
		// written in order
		const int T = host_threads(0);
		const uint64_t CHUNK = 1u << 18;
		std::vector<std::string> bufs((size_t)T);
		std::vector<uint32_t> order;             // spec.sorted: the read written at each place
		if (S->sorted && (rc = sorted_order(*S, Mo, order))) { fclose(fr); return rc; }
		auto put_int = [](std::string &o, long long v) {
			char tmp[24]; int n = 0;
			if (v < 0) { o.push_back('-'); v = -v; }
			do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
			while (n) o.push_back(tmp[--n]);
		};
		bool io_ok = true;
		for (uint64_t round0 = 0; round0 < S->n_reads && io_ok; round0 += CHUNK * (uint64_t)T) {
			ThreadGroup th;
			for (int t = 0; t < T; ++t) {
				th.spawn([&, t] {
					std::string &o = bufs[(size_t)t];
					o.clear();
					const uint64_t i0 = round0 + CHUNK * (uint64_t)t, i1 = std::min<uint64_t>(i0 + CHUNK, S->n_reads);
					OneRead r;
					for (uint64_t i = i0; i < i1; ++i) {
						gen_read(*S, Mo, S->first_read + (order.empty() ? i : (uint64_t)order[i]), r);
						int q = 1;
						for (int b = 0; b < r.nb; ++b) {
							const int ln = r.be[b] - r.bs[b];
							if (b) o.push_back(',');
							o += "chr"; put_int(o, r.chrom + 1);
							o.push_back(':'); o.push_back(r.strand); o.push_back(':');
							put_int(o, (long long)r.bs[b] + 1); o.push_back(':'); put_int(o, r.be[b]); o.push_back(':');
							put_int(o, q); o.push_back(':'); put_int(o, (long long)q + ln - 1);
							q += ln;
						}
						o.push_back('\n');
					}
				});
			}
			th.join();
			if (th.failed()) { fclose(fr); return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str()); }
			for (int t = 0; t < T; ++t) if (!bufs[(size_t)t].empty() && fwrite(bufs[(size_t)t].data(), 1, bufs[(size_t)t].size(), fr) != bufs[(size_t)t].size()) io_ok = false;
		}
		if (fclose(fr) != 0 || !io_ok) return fail(LSQ_E_IO, "cannot write %s.mrf", base.c_str());
	}
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_synth_reads(const lsq_synth_spec *S, lsq_events *E, int n_threads, lsq_reads **out) LSQ_API_TRY {
	if (!S || !E || !out) return fail(LSQ_E_ARG, "null argument");
	SynModel Mo;
	int rc = build_model(*S, Mo);
	if (rc) return rc;
	if (S->n_reads > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "more than 2^32 reads in one read set");
	std::vector<int> cid(S->n_chrom);
	for (uint32_t c = 0; c < S->n_chrom; ++c) {
		int id = E->chroms.find(chrom_name((int)c));
		cid[c] = (id < 0 || (size_t)id >= E->covered.size()) ? 0xFFFF : id;
	}
	int plus = lsq_events_strand_id(E, "+"), minus = lsq_events_strand_id(E, "-");
	if (plus < 0 || minus < 0) return LSQ_E_RANGE;
	std::unique_ptr<lsq_reads> Rd(new lsq_reads);
	const uint64_t n = S->n_reads;
	int T = host_threads(n_threads);
	if (n < 100000) T = 1;
	std::vector<uint32_t> order;
	if (S->sorted && (rc = sorted_order(*S, Mo, order))) return rc;
	auto at = [&](uint64_t i) { return S->first_read + (order.empty() ? i : (uint64_t)order[i]); };
	// pass 1: block counts
	Rd->o_blk_off.assign(n + 1, 0);
	Rd->o_line_no.resize(n);
	auto range = [&](int t, uint64_t &a, uint64_t &b) { a = n * (uint64_t)t / (uint64_t)T; b = n * (uint64_t)(t + 1) / (uint64_t)T; };
	{
		ThreadGroup th;
		for (int t = 0; t < T; ++t) th.spawn([&, t] {
			uint64_t a, b; range(t, a, b);
			OneRead r;
			for (uint64_t i = a; i < b; ++i) { gen_read(*S, Mo, at(i), r); Rd->o_blk_off[i + 1] = (uint64_t)r.nb; Rd->o_line_no[i] = (uint32_t)(i + 1); }
		});
		th.join();
		if (th.failed()) return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
	}
	for (uint64_t i = 0; i < n; ++i) Rd->o_blk_off[i + 1] += Rd->o_blk_off[i];
	const uint64_t nb = Rd->o_blk_off[n];
	Rd->o_start.resize(nb); Rd->o_end.resize(nb); Rd->o_chrom.resize(nb); Rd->o_strand.resize(nb);
	{
		ThreadGroup th;
		for (int t = 0; t < T; ++t) th.spawn([&, t] {
			uint64_t a, b; range(t, a, b);
			OneRead r;
			for (uint64_t i = a; i < b; ++i) {
				gen_read(*S, Mo, at(i), r);
				uint64_t o = Rd->o_blk_off[i];
				for (int k = 0; k < r.nb; ++k) {
					Rd->o_start[o + k] = r.bs[k]; Rd->o_end[o + k] = r.be[k];
					Rd->o_chrom[o + k] = (uint16_t)cid[r.chrom];
					Rd->o_strand[o + k] = (uint8_t)(r.strand == '+' ? plus : minus);
				}
			}
		});
		th.join();
		if (th.failed()) return fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
	}
	Rd->adopt();
	*out = Rd.release();
	return LSQ_OK;
} LSQ_API_CATCH

} // extern "C"
