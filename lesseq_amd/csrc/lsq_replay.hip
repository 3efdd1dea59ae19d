// Exact-order replay of the EM for events whose stop test came within the guard band.
//
// lsq_em_kernel sums over compatibility classes; the reference sums over reads, one after the other, in the
// order of its read index (count/count.cpp:64-85: start, end, strand, name; solve/solve.cpp:720-748 fills
// valid_rnames in that order, :767-793 builds one row per valid read, common/read.h:592-636 adds the rows in
// turn).  The two differ by ~1e-13 relative, which matters only when |1 - old_ll/ll| lands within that of the
// 1e-6 threshold (read.h:659): then the iteration count can differ and theta moves far more than 1e-6.  The
// kernel flags such events (em_flags bit 0; the band is lsq_set_em_guard_band, 1e-11 by default) and this unit
// redoes them: the reads of the event's bucket come back from HBM, each is evaluated against the event
// (candidate window, Read::build, contiguous-run compatibility, 0.98 rule), the valid ones are put in index
// order and the EM runs over that sequence with the reference's own loops, on the host, in IEEE fp64 with
// libm's log -- the operations of read.h:592-660 in the same order, so the result is the reference's.
// Flagged events are rare (3 of 200 000 on the skewed 1 B-read workload), so this is not a throughput path.
#include "lsq_device.hpp"
#include <atomic>

// The same unit evaluates HOST BUCKETS (BucketDesc::kind 2): clusters of events that hold a gene beyond the kernels'
// limits (more than LSQ_MAX_ISOFORMS isoforms or LSQ_MAX_SEGMENTS segments: the whole-gene annotations solve's further
// formats carry) or that do not fit the CU's LDS.  Their reads are pooled on the device like any bucket's; lsq_count
// brings them back, evaluates every read against every event of the bucket with the rules above and writes the class
// counts into the device tables; lsq_solve runs the per-read EM for them.  Those calls then block.

namespace {

struct HostRead {
	int start, end;            // first merged start, last merged end
	unsigned line;             // MRF line number, or index into the method's name table
	unsigned matched;          // bases matched (Read::get_read_length)
	unsigned short cls;        // compatibility class (bit j: isoform j), 0 = not valid for this event
	unsigned char strand;
};

// Read_single::build against one event's ascending segments (common/read.h:204-274), as lsq_count.hip's Walk
struct HostWalk {
	long long pos = 0; int it = 0; bool found = false; uint64_t mask = 0; long long matched = 0;
	bool block(const lsq::Event &e, long long a, long long b) {
		while (it < e.N) {
			const long long sx = e.seg_s[(size_t)it], sy = e.seg_e[(size_t)it];
			if (!(sx < b)) break;
			pos = std::max(pos, sx);
			if (a >= pos && a < sy) {
				if (found && a > pos) break;
				found = true;
				mask |= 1ull << it;
				pos = std::min(sy, b);
				matched += pos - a;
				if (b < sy) { a = b; break; }
				a = (b == sy) ? b : sy;
			} else if (pos > sx && pos < sy) {
				break;
			}
			++it;
		}
		return a == b;
	}
};

// class of a read given by its merged blocks; 0 when it is no candidate of the event or not valid
unsigned eval_read(const lsq::Event &e, const int2 *blk, int nblk, bool read_orders_first, unsigned *matched) {
	const long long p = blk[0].x, q = blk[nblk - 1].y;
	if (p < e.gene_start || p > e.gene_end) return 0;
	// count/count.cpp:429-432: lower_bound on (chrom, gene_start, gene_end, strand, name)
	if (p == e.gene_start && (q < e.gene_end || (q == e.gene_end && read_orders_first))) return 0;
	HostWalk w;
	long long total = 0;
	for (int k = 0; k < nblk; ++k) total += (long long)blk[k].y - blk[k].x;
	for (int k = 0; k < nblk; ++k) if (!w.block(e, blk[k].x, blk[k].y)) break;
	if (!w.mask) return 0;
	if (!(50ll * w.matched > 49ll * total)) return 0;        // (double)matched / total > 0.98, count.cpp:441
	const int hi = 63 - __builtin_clzll(w.mask), lo = __builtin_ctzll(w.mask);
	const uint64_t span = (hi == 63 ? ~0ull : ((2ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
	unsigned cls = 0;
	for (int j = 0; j < e.K; ++j) {
		const uint64_t iso = e.iso_mask[(size_t)j];
		if ((w.mask & ~iso) == 0 && (iso & span) == w.mask) cls |= 1u << j;       // read.h:44-79
	}
	*matched = (unsigned)w.matched;
	return cls;
}

template <class T>
int fetch(std::vector<T> &dst, const T *src, size_t first, size_t count) {
	dst.resize(count);
	if (count) HIP_TRY(hipMemcpy(dst.data(), src + first, count * sizeof(T), hipMemcpyDeviceToHost));
	return LSQ_OK;
}

// common/read.h:638-660 over per-read rows that are given as a class sequence per read file
struct ExactEm {
	int K = 0;
	std::vector<std::vector<double>> G;               // [method][K]
	std::vector<std::vector<unsigned short>> seq;     // [method] classes of the valid reads, index order
	double row(size_t m, unsigned c, int k) const { return (c >> k & 1u) ? G[m][(size_t)k] : 0.0; }
	double log_likelihood(const std::vector<double> &th) const {            // read.h:620-636
		double ll = 0;
		std::vector<double> lg((size_t)1 << K);
		for (size_t m = 0; m < seq.size(); ++m) {
			if (seq[m].empty()) continue;
			for (unsigned c = 1; c < (1u << K); ++c) {
				double s = 0;
				for (int k = 0; k < K; ++k) s += th[(size_t)k] * row(m, c, k);
				lg[c] = std::log(s);
			}
			for (unsigned short c : seq[m]) ll += lg[c];
		}
		return ll;
	}
	void step(const std::vector<double> &old_th, std::vector<double> &new_th) const {      // read.h:592-618
		std::vector<double> z((size_t)1 << K);
		for (int k = 0; k < K; ++k) {
			double sum_zeta = 0, num_total_reads = 0;
			for (size_t m = 0; m < seq.size(); ++m) {
				if (seq[m].empty()) continue;
				num_total_reads += (double)seq[m].size();
				for (unsigned c = 1; c < (1u << K); ++c) {
					double s = 0;
					for (int k2 = 0; k2 < K; ++k2) s += old_th[(size_t)k2] * row(m, c, k2);
					z[c] = 0.0;                   // a skipped term: adding +0.0 leaves the running sum as it is
					if (s > 0) {
						const double local = old_th[(size_t)k] * row(m, c, k);
						if (local > 0) z[c] = local / s;
					}
				}
				for (unsigned short c : seq[m]) sum_zeta += z[c];
			}
			new_th[(size_t)k] = sum_zeta / num_total_reads;
		}
	}
	// solve/solve.cpp:796-806,823-826: no valid read at all -> 1/K each; one isoform -> 1; else the EM.  Returns the iterations.
	unsigned run(std::vector<double> &theta, double &logll) const {
		bool any = false;
		for (const auto &q : seq) any = any || !q.empty();
		theta.assign((size_t)K, (double)1.0 / (double)K);
		if (!any) { logll = 0; return 0; }
		if (K == 1) { theta[0] = 1.0; logll = log_likelihood(theta); return 0; }
		std::vector<double> old_theta;
		double ll, old_ll;
		unsigned iters = 0;
		do {
			old_theta = theta;
			old_ll = log_likelihood(old_theta);
			step(old_theta, theta);
			ll = log_likelihood(theta);
			++iters;
		} while (std::fabs(1.0 - old_ll / ll) > 1E-6);
		logll = ll;              // solve/solve.cpp:824: the log-likelihood at the final theta, the same sum once more
		return iters;
	}
};

// std::string operator< on the decimal forms of two numbers
bool decimal_less(unsigned a, unsigned b) {
	char sa[12], sb[12];
	int na = 0, nb = 0;
	do { sa[na++] = (char)('0' + a % 10u); a /= 10u; } while (a);
	do { sb[nb++] = (char)('0' + b % 10u); b /= 10u; } while (b);
	for (int i = 0; i < na && i < nb; ++i) { const char ca = sa[na - 1 - i], cb = sb[nb - 1 - i]; if (ca != cb) return ca < cb; }
	return na < nb;
}

struct NameTable { bool named = false; std::string blob; std::vector<unsigned long long> off; };

std::string read_name(const NameTable &nt, unsigned line) {
	if (nt.named) return nt.blob.substr((size_t)nt.off[line], (size_t)(nt.off[line + 1] - nt.off[line]));
	return "read-" + std::to_string(line);          // count/count.cpp:293-295
}

int fetch_names(lsq_ctx *c, std::vector<NameTable> &out) {
	const size_t M = (size_t)c->E->n_methods;
	out.assign(M, NameTable());
	for (size_t m = 0; m < M; ++m) {
		const lsq::MethodReads &mr = c->reads[m];
		if (!mr.named || !mr.present) continue;
		out[m].named = true;
		std::vector<char> blob;
		int rc = fetch(blob, mr.names.p, 0, mr.names.n); if (rc) return rc;
		out[m].blob.assign(blob.begin(), blob.end());
		rc = fetch(out[m].off, mr.name_off.p, 0, mr.name_off.n); if (rc) return rc;
	}
	return LSQ_OK;
}

struct BucketReads {                       // one read file's reads of one bucket, as the pools hold them
	std::vector<int32_t> p1, p2, pn_se;
	std::vector<uint8_t> p1_strand, p2_strand, pn_strand;
	std::vector<uint32_t> p1_line, p2_line, pn_line, pn_blk_off, pn_nblk;
	unsigned long long pnb0 = 0, pnb1 = 0;
};

int fetch_bucket(const lsq::MethodReads &mr, size_t b, int32_t lo, BucketReads &R) {
	unsigned long long o1[2], o2[2], on[2], ob[2];
	HIP_TRY(hipMemcpy(o1, mr.p1_off.p + b, sizeof o1, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(o2, mr.p2_off.p + b, sizeof o2, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(on, mr.pn_off.p + b, sizeof on, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(ob, mr.pnb_off.p + b, sizeof ob, hipMemcpyDeviceToHost));
	const size_t n1 = (size_t)(o1[1] - o1[0]), n2 = (size_t)(o2[1] - o2[0]), nn = (size_t)(on[1] - on[0]);
	int rc;
	const size_t w1 = mr.compact ? 1 : 2, w2 = mr.compact ? 2 : 4;       // ints per pool record
	if ((rc = fetch(R.p1, mr.p1.p, w1 * (size_t)o1[0], w1 * n1)) || (rc = fetch(R.p1_strand, mr.p1_strand.p, (size_t)o1[0], n1)) || (rc = fetch(R.p1_line, mr.p1_line.p, (size_t)o1[0], n1)) ||
	    (rc = fetch(R.p2, mr.p2.p, w2 * (size_t)o2[0], w2 * n2)) || (rc = fetch(R.p2_strand, mr.p2_strand.p, (size_t)o2[0], n2)) || (rc = fetch(R.p2_line, mr.p2_line.p, (size_t)o2[0], n2)) ||
	    (rc = fetch(R.pn_blk_off, mr.pn_blk_off.p, (size_t)on[0], nn)) || (rc = fetch(R.pn_nblk, mr.pn_nblk.p, (size_t)on[0], nn)) ||
	    (rc = fetch(R.pn_strand, mr.pn_strand.p, (size_t)on[0], nn)) || (rc = fetch(R.pn_line, mr.pn_line.p, (size_t)on[0], nn)) ||
	    (rc = fetch(R.pn_se, mr.pn_se.p, 2 * (size_t)ob[0], 2 * (size_t)(ob[1] - ob[0])))) return rc;
	R.pnb0 = ob[0]; R.pnb1 = ob[1];
	if (mr.compact) {         // to wide records (lsq_device.hpp COMPACT_*)
		std::vector<int32_t> a(2 * n1), b2(4 * n2);
		const int32_t base = lo - lsq::COMPACT_BIAS;
		auto off = [](int32_t w) { return (int32_t)((uint32_t)w & lsq::COMPACT_OFF_MASK); };
		auto len = [](int32_t w) { return (int32_t)((uint32_t)w >> lsq::COMPACT_OFF_BITS); };
		for (size_t i = 0; i < n1; ++i) { a[2 * i] = base + off(R.p1[i]); a[2 * i + 1] = a[2 * i] + len(R.p1[i]); }
		for (size_t i = 0; i < n2; ++i) {
			const int32_t s1 = base + off(R.p2[2 * i]), e1 = s1 + len(R.p2[2 * i]), s2 = e1 + off(R.p2[2 * i + 1]);
			b2[4 * i] = s1; b2[4 * i + 1] = e1; b2[4 * i + 2] = s2; b2[4 * i + 3] = s2 + len(R.p2[2 * i + 1]);
		}
		R.p1.swap(a); R.p2.swap(b2);
	}
	return LSQ_OK;
}

// the reads of the bucket that are valid for the event, in the order of the reference's read index
// (count/count.cpp:64-85: start, end, strand, name; the chromosome is the event's for all of them)
int valid_reads(const lsq_events &E, const lsq::Event &ev, const BucketReads &R, const NameTable &nt, std::vector<HostRead> &valid) {
	valid.clear();
	auto consider = [&](const int2 *blk, int nblk, unsigned char strand, unsigned line) {
		const int p = blk[0].x, q = blk[nblk - 1].y;
		bool first = false;
		if (p == ev.gene_start && q == ev.gene_end) {          // the (strand, name) part of the key decides
			const std::string &rs = E.strands.names[strand];
			first = rs != ev.strand ? rs < ev.strand : read_name(nt, line) < ev.gname;
		}
		unsigned matched = 0;
		const unsigned cls = eval_read(ev, blk, nblk, first, &matched);
		if (cls) valid.push_back({p, q, line, matched, (unsigned short)cls, strand});
	};
	for (size_t i = 0; i < R.p1_line.size(); ++i) {
		const int2 blk = make_int2(R.p1[2 * i], R.p1[2 * i + 1]);
		if (blk.y == blk.x) continue;            // padding of a cell's group
		consider(&blk, 1, R.p1_strand[i], R.p1_line[i]);
	}
	for (size_t i = 0; i < R.p2_line.size(); ++i) {
		const int2 blk[2] = {make_int2(R.p2[4 * i], R.p2[4 * i + 1]), make_int2(R.p2[4 * i + 2], R.p2[4 * i + 3])};
		if (blk[0].y == blk[0].x) continue;      // padding of a junction group
		consider(blk, 2, R.p2_strand[i], R.p2_line[i]);
	}
	for (size_t i = 0; i < R.pn_line.size(); ++i) {
		if (R.pn_blk_off[i] < R.pnb0 || (unsigned long long)R.pn_blk_off[i] + R.pn_nblk[i] > R.pnb1)
			return lsq::fail(LSQ_E_STATE, "host evaluation: block offsets of a multi-block read lie outside its bucket");
		consider(reinterpret_cast<const int2 *>(R.pn_se.data()) + (R.pn_blk_off[i] - R.pnb0), (int)R.pn_nblk[i], R.pn_strand[i], R.pn_line[i]);
	}
	std::sort(valid.begin(), valid.end(), [&](const HostRead &x, const HostRead &y) {
		if (x.start != y.start) return x.start < y.start;
		if (x.end != y.end) return x.end < y.end;
		if (x.strand != y.strand) return E.strands.names[x.strand] < E.strands.names[y.strand];
		if (!nt.named) return decimal_less(x.line, y.line);       // "read-<n>" against "read-<n'>": the digits as strings
		return read_name(nt, x.line) < read_name(nt, y.line);
	});
	return LSQ_OK;
}

void fill_G(const lsq::Event &ev, size_t M, ExactEm &em) {
	em.K = ev.K; em.G.resize(M); em.seq.resize(M);
	for (size_t m = 0; m < M; ++m) {
		em.G[m].resize((size_t)ev.K);
		for (int j = 0; j < ev.K; ++j) { const double nd = (double)ev.ars[m][(size_t)j]; em.G[m][(size_t)j] = nd <= 0 ? 0.0 : (double)1.0 / nd; }   // read.h:331-340
	}
}

int write_solution(lsq_ctx *c, size_t d, const lsq::Event &ev, const std::vector<double> &theta, double logll, unsigned iters, uint8_t flags) {
	const lsq_events &E = *c->E;
	HIP_TRY(hipMemcpy(c->theta.p + E.dev_iso_base[d], theta.data(), (size_t)ev.K * sizeof(double), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(c->logll.p + d, &logll, sizeof(double), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(c->iters.p + d, &iters, sizeof(unsigned), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(c->flags.p + d, &flags, 1, hipMemcpyHostToDevice));
	return LSQ_OK;
}

} // namespace

namespace lsq {

// Redoes every event of the latest solve that carries flag bit 0 and not yet bit 2.  Both streams are idle on return.
int replay_flagged(lsq_ctx *c, unsigned *n_done) {
	if (n_done) *n_done = 0;
	if (!c->solved) return LSQ_OK;
	{ int rc = sync_all(c); if (rc) return rc; }
	const lsq_events &E = *c->E;
	const size_t n_ev = E.dev2out.size(), M = (size_t)E.n_methods, n_cls = E.n_cls_total;
	if (!n_ev) return LSQ_OK;
	std::vector<uint8_t> flags(n_ev);
	HIP_TRY(hipMemcpy(flags.data(), c->flags.p, n_ev, hipMemcpyDeviceToHost));
	std::vector<size_t> todo;
	for (size_t d = 0; d < n_ev; ++d) if ((flags[d] & 1u) && !(flags[d] & 4u)) todo.push_back(d);
	// counts that came from outside (lsq_results_set_counts: sums over processes that each hold a slice of the
	// reads): the reads behind them are not all here, the flag stays as the kernel set it
	if (todo.empty() || c->counts_external) return LSQ_OK;
	std::vector<NameTable> names;
	{ int rc = fetch_names(c, names); if (rc) return rc; }
	// bucket of a device event: the last one whose first event is <= d; flagged events by bucket, so that a bucket's
	// reads come back once per read file
	auto bucket_of = [&](size_t d) {
		size_t lo = 0, hi = E.buckets.size();
		while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (E.buckets[mid].ev_base <= d) lo = mid; else hi = mid; }
		return lo;
	};
	for (size_t t0 = 0; t0 < todo.size();) {
		const size_t b = bucket_of(todo[t0]);
		size_t t1 = t0;
		while (t1 < todo.size() && bucket_of(todo[t1]) == b) ++t1;
		std::vector<ExactEm> ems(t1 - t0);
		for (size_t t = t0; t < t1; ++t) fill_G(E.ev[(size_t)E.dev2out[todo[t]]], M, ems[t - t0]);
		for (size_t m = 0; m < M; ++m) {
			const MethodReads &mr = c->reads[m];
			if (!mr.present) return fail(LSQ_E_STATE, "the exact-order EM replay needs the reads of method %zu on the device", m);
			BucketReads R;
			{ int rc = fetch_bucket(mr, b, E.buckets[b].lo, R); if (rc) return rc; }
			for (size_t t = t0; t < t1; ++t) {
				const size_t d = todo[t];
				const Event &ev = E.ev[(size_t)E.dev2out[d]];
				std::vector<HostRead> valid;
				{ int rc = valid_reads(E, ev, R, names[m], valid); if (rc) return rc; }
				// the same reads the device counted?  (the replay must not drift from the count tables)
				std::vector<unsigned long long> mine((size_t)1 << ev.K, 0), dev(((size_t)1 << ev.K) - 1);
				for (const HostRead &r : valid) ++mine[r.cls];
				HIP_TRY(hipMemcpy(dev.data(), c->cnt.p + m * n_cls + E.dev_cls_base[d], dev.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
				for (size_t k = 0; k < dev.size(); ++k)
					if (dev[k] != mine[k + 1])
						return fail(LSQ_E_STATE, "exact-order EM replay of gene %s: class %zu has %llu reads here, %llu in the count tables", ev.gname.c_str(), k + 1, mine[k + 1], dev[k]);
				std::vector<unsigned short> &seq = ems[t - t0].seq[m];
				seq.reserve(valid.size());
				for (const HostRead &r : valid) seq.push_back(r.cls);
			}
		}
		for (size_t t = t0; t < t1; ++t) {
			const size_t d = todo[t];
			std::vector<double> theta;
			double logll = 0;
			const unsigned iters = ems[t - t0].run(theta, logll);
			int rc = write_solution(c, d, E.ev[(size_t)E.dev2out[d]], theta, logll, iters, (uint8_t)(flags[d] | 4u));
			if (rc) return rc;
			if (n_done) ++*n_done;
		}
		t0 = t1;
	}
	return LSQ_OK;
}

// Host buckets, count: every read of the bucket against every event of it; the class counts and matched bases go
// into the device tables, the class sequences stay in the context for host_solve.  The reads come back on the calling
// thread (HIP calls stay there); the evaluation -- events x reads of a bucket -- is dealt to LSQ_THREADS host threads,
// an event at a time (a whole-gene annotation through solve's other formats puts thousands of genes here).
int host_count(lsq_ctx *c) {
	const lsq_events &E = *c->E;
	const size_t M = (size_t)E.n_methods, n_cls = E.n_cls_total;
	c->host_seq.clear();
	c->host_genes = 0; c->host_reads = 0;
	{ int rc = sync_all(c); if (rc) return rc; }
	std::vector<NameTable> names;
	{ int rc = fetch_names(c, names); if (rc) return rc; }
	struct Item { size_t b, m, d; const BucketReads *R; std::vector<unsigned long long> cnt, bases; std::vector<unsigned short> *seq; int status = LSQ_OK; std::string error; };
	std::vector<std::unique_ptr<BucketReads>> fetched;
	std::vector<Item> items;
	for (size_t b = 0; b < E.buckets.size(); ++b) {
		const BucketDesc &bd = E.buckets[b];
		if (bd.kind != 2) continue;
		c->host_genes += bd.n_events;
		for (size_t m = 0; m < M; ++m) {
			const MethodReads &mr = c->reads[m];
			fetched.emplace_back(new BucketReads);
			BucketReads &R = *fetched.back();
			{ int rc = fetch_bucket(mr, b, E.buckets[b].lo, R); if (rc) return rc; }
			for (size_t i = 0; i < R.p1_line.size(); ++i) c->host_reads += R.p1[2 * i] != R.p1[2 * i + 1];       // (padding aside)
			for (size_t i = 0; i < R.p2_line.size(); ++i) c->host_reads += R.p2[4 * i] != R.p2[4 * i + 1];
			c->host_reads += R.pn_line.size();
			for (size_t d = bd.ev_base; d < (size_t)bd.ev_base + bd.n_events; ++d) {
				Item it;
				it.b = b; it.m = m; it.d = d; it.R = &R;
				it.seq = &c->host_seq[d * M + m];          // the map grows here, on one thread; the workers fill the vectors
				items.push_back(std::move(it));
			}
		}
	}
	if (items.empty()) return LSQ_OK;
	std::atomic<size_t> next{0};
	auto work = [&] {
		for (;;) {
			const size_t q = next.fetch_add(1, std::memory_order_relaxed);
			if (q >= items.size()) return;
			Item &it = items[q];
			const Event &ev = E.ev[(size_t)E.dev2out[it.d]];
			std::vector<HostRead> valid;
			it.status = valid_reads(E, ev, *it.R, names[it.m], valid);
			if (it.status) { it.error = lsq_last_error(); continue; }        // (the text lives in this thread)
			const size_t nc = ((size_t)1 << ev.K) - 1;
			it.cnt.assign(nc, 0); it.bases.assign(nc, 0);
			it.seq->reserve(valid.size());
			for (const HostRead &r : valid) { ++it.cnt[r.cls - 1u]; it.bases[r.cls - 1u] += r.matched; it.seq->push_back(r.cls); }
		}
	};
	{
		ThreadGroup th;
		const size_t T = std::min<size_t>((size_t)std::max(1, host_threads(0)), items.size());
		for (size_t t = 1; t < T; ++t) th.spawn(work);
		th.run_here(work);
		th.join();
		if (th.failed()) return fail(LSQ_E_INTERNAL, "host evaluation of genes beyond the kernels' limits: %s", th.error().c_str());
	}
	for (const Item &it : items) {
		if (it.status) return fail(it.status, "%s", it.error.c_str());
		const size_t nc = it.cnt.size();
		HIP_TRY(hipMemcpy(c->cnt.p + it.m * n_cls + E.dev_cls_base[it.d], it.cnt.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(c->bases.p + it.m * n_cls + E.dev_cls_base[it.d], it.bases.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice));
	}
	return LSQ_OK;
}

// Host buckets, solve: the reference's per-read EM over the sequences host_count kept (flag bit 2: exact order), the
// events dealt to the host threads like the count's
int host_solve(lsq_ctx *c) {
	const lsq_events &E = *c->E;
	const size_t M = (size_t)E.n_methods;
	{ int rc = sync_all(c); if (rc) return rc; }
	struct Item { size_t d; std::vector<double> theta; double logll = 0; unsigned iters = 0; };
	std::vector<Item> items;
	for (size_t b = 0; b < E.buckets.size(); ++b) {
		const BucketDesc &bd = E.buckets[b];
		if (bd.kind != 2) continue;
		for (size_t d = bd.ev_base; d < (size_t)bd.ev_base + bd.n_events; ++d) { Item it; it.d = d; items.push_back(std::move(it)); }
	}
	if (items.empty()) return LSQ_OK;
	std::atomic<size_t> next{0};
	auto work = [&] {
		for (;;) {
			const size_t q = next.fetch_add(1, std::memory_order_relaxed);
			if (q >= items.size()) return;
			Item &it = items[q];
			const Event &ev = E.ev[(size_t)E.dev2out[it.d]];
			ExactEm em;
			fill_G(ev, M, em);
			for (size_t m = 0; m < M; ++m) { auto f = c->host_seq.find(it.d * M + m); if (f != c->host_seq.end()) em.seq[m] = f->second; }      // (the map is only read here)
			it.iters = em.run(it.theta, it.logll);
		}
	};
	{
		ThreadGroup th;
		const size_t T = std::min<size_t>((size_t)std::max(1, host_threads(0)), items.size());
		for (size_t t = 1; t < T; ++t) th.spawn(work);
		th.run_here(work);
		th.join();
		if (th.failed()) return fail(LSQ_E_INTERNAL, "host solve of genes beyond the kernels' limits: %s", th.error().c_str());
	}
	for (const Item &it : items) {
		int rc = write_solution(c, it.d, E.ev[(size_t)E.dev2out[it.d]], it.theta, it.logll, it.iters, (uint8_t)4u);
		if (rc) return rc;
	}
	return LSQ_OK;
}

} // namespace lsq

extern "C" {

int lsq_set_em_guard_band(lsq_ctx *c, double band) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!(band >= 0.0)) return fail(LSQ_E_ARG, "the guard band must be a non-negative number");
	c->em_band = band;
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_solve_finalize(lsq_ctx *c, uint32_t *n_replayed) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->solved) return fail(LSQ_E_STATE, "lsq_solve must come first");
	HIP_TRY(hipSetDevice(c->device));
	unsigned n = 0;
	int rc = lsq::replay_flagged(c, &n);
	if (n_replayed) *n_replayed = n;
	return rc;
} LSQ_API_CATCH

// Genes beyond the kernels' limits (host buckets) as the latest lsq_count met them: how many, and how many reads their
// clusters held -- what the host evaluated instead of the kernels (the executables say so at log level 1)
int lsq_host_evaluated(const lsq_ctx *c, uint64_t *n_genes, uint64_t *n_reads) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	if (n_genes) *n_genes = c->has_host ? c->host_genes : 0;
	if (n_reads) *n_reads = c->has_host ? c->host_reads : 0;
	return LSQ_OK;
} LSQ_API_CATCH

// Developer check of the pools' layout (tests): every aligned group of eight one-block records starts in one cell (or all
// in none), every aligned quadruple of two-block records (P2_GROUP_PAD) of a junction group crosses one junction, and the records that are
// not padding number what the ingest counted.  out: [0] one-block records, [1] of them padding, [2] groups of eight over more
// than one cell, [3] two-block records, [4] of them padding, [5] such groups whose first read crosses a junction of the
// annotation and another of whose reads crosses another or none.
int lsq_debug_check_pool_layout(lsq_ctx *c, int method, unsigned long long *out) LSQ_API_TRY {
	if (!c || !c->E || !out || method < 0 || method >= c->E->n_methods) return fail(LSQ_E_ARG, "bad argument");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = lsq::sync_all(c); if (rc) return rc; }
	const lsq_events &E = *c->E;
	const lsq::MethodReads &mr = c->reads[method];
	if (!mr.present) return fail(LSQ_E_STATE, "no reads uploaded for method %d", method);
	for (int q = 0; q < 6; ++q) out[q] = 0;
	for (size_t b = 0; b < E.buckets.size(); ++b) {
		const lsq::BucketDesc &d = E.buckets[b];
		BucketReads R;
		{ int rc = fetch_bucket(mr, b, d.lo, R); if (rc) return rc; }
		const lsq::Cell *cells = d.kind == 1 ? reinterpret_cast<const lsq::Cell *>(E.images.data() + d.img_off + d.seg_off) : nullptr;
		const unsigned n_cells = d.kind == 1 ? (d.iso_off & 0xFFFFu) : 0u;
		auto cell_of = [&](int p) -> long {
			for (unsigned q = 0; q < n_cells; ++q) if (cells[q].lo <= p && p < cells[q].hi) return (long)q;
			return -1;
		};
		const size_t n1 = R.p1_line.size(), n2 = R.p2_line.size();
		if ((n1 % P1_GROUP_PAD) || (n2 % P2_GROUP_PAD)) return fail(LSQ_E_STATE, "bucket %zu: pool slices of %zu and %zu records are not whole groups of %u / %u", b, n1, n2, P1_GROUP_PAD, P2_GROUP_PAD);
		out[0] += n1; out[3] += n2;
		for (size_t q = 0; q < n1; q += P1_GROUP_PAD) {          // what a lane of lsq_count_fast_kernel<true, 4> settles with one look
			long first = -2;
			bool mixed = false;
			for (size_t k = q; k < q + P1_GROUP_PAD; ++k) {
				const int s = R.p1[2 * k], e = R.p1[2 * k + 1];
				if (s == e) { ++out[1]; continue; }
				const long cl = cell_of(s);
				if (first == -2) first = cl; else if (cl != first) mixed = true;
			}
			if (mixed) ++out[2];
		}
		const uint64_t *k0 = E.jg_keys.data() + E.jg_base[b], *k1 = E.jg_keys.data() + E.jg_base[b + 1];
		auto group_of = [&](const int32_t *r) -> long {          // index of the read's junction key, -1: the bucket's last group
			const long cl = cell_of(r[0]);
			if (cl < 0 || (r[1] != cells[cl].e1 && r[1] != cells[cl].e2)) return -1;
			const uint64_t want = lsq::jg_key((uint32_t)cl, r[1] == cells[cl].e1 ? 0u : 1u, r[2]);
			const uint64_t *it = std::lower_bound(k0, k1, want);
			return (it != k1 && *it == want) ? (long)(it - k0) : -1;
		};
		for (size_t q = 0; q < n2; q += P2_GROUP_PAD) {          // what a lane settles with one look
			const int32_t *ra = &R.p2[4 * q];
			const bool ea = ra[0] == ra[1];
			const long ga = ea ? -1 : group_of(ra);
			bool bad = false;
			out[4] += ea ? 1u : 0u;
			for (size_t k = q + 1; k < q + P2_GROUP_PAD; ++k) {
				const int32_t *rb = &R.p2[4 * k];
				const bool eb = rb[0] == rb[1];
				out[4] += eb ? 1u : 0u;
				if (ea) bad = bad || !eb;                       // padding never comes first
				else if (ga >= 0 && !eb && group_of(rb) != ga) bad = true;
				else if (ga < 0 && !eb && (group_of(rb) >= 0 || cell_of(rb[0]) != cell_of(ra[0]))) bad = true;       // no junction: the quadruple shares the cell of block 1
			}
			if (bad) ++out[5];
		}
	}
	if (out[0] - out[1] != mr.n1_reads || out[3] - out[4] != mr.n2_reads)
		return fail(LSQ_E_STATE, "the pools hold %llu + %llu reads, the ingest counted %llu + %llu", out[0] - out[1], out[3] - out[4],
		            (unsigned long long)mr.n1_reads, (unsigned long long)mr.n2_reads);
	return LSQ_OK;
} LSQ_API_CATCH

} // extern "C"
