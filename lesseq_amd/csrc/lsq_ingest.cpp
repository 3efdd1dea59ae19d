// Host ingest: from parsed blocks in file order to the bucketed, pooled structure-of-arrays
// that is copied to HBM.  Per read: the per-block containment filter against the covered
// regions of the block's own chromosome (count/count.cpp:319), the interval_list merge of
// the kept blocks (:323), chromosome/strand of the last kept block (:321-322), and the key
// start = first merged start, end = last merged end (:359-360).
#include <algorithm>
#include <cstring>
#include <thread>

#include "lsq_internal.hpp"

namespace lsq {

namespace {

const uint16_t NOCHROM = 0xFFFF;

struct Kept {
	IntervalList il;       // reused per thread
	int chrom = -1, strand = 0;
	bool any = false;
};

inline void filter_merge(const lsq_events &E, const lsq_reads &R, uint64_t i, Kept &k) {
	k.il.s.clear(); k.il.e.clear();
	k.any = false;
	for (uint64_t j = R.blk_off[i]; j < R.blk_off[i + 1]; ++j) {
		uint16_t c = R.blk_chrom[j];
		if (c == NOCHROM || c >= E.covered.size()) continue;
		int64_t s = R.blk_start[j], e = R.blk_end[j];
		if (!E.covered[c].contains(s, e)) continue;
		k.any = true;
		k.chrom = c;
		k.strand = R.blk_strand[j];
		if (k.il.s.empty() && s < e) { k.il.s.push_back(s); k.il.e.push_back(e); }
		else k.il.add(s, e);
	}
}

inline int bucket_of(const lsq_events &E, int chrom, int64_t p) {
	int first = E.chrom_first_bucket[chrom];
	if (first < 0) return -1;
	const std::vector<int32_t> &cuts = E.cut_lo[chrom];
	size_t k = std::upper_bound(cuts.begin(), cuts.end(), (int32_t)p) - cuts.begin();
	if (k == 0) return -1;                                   // left of the chromosome's first span
	const int b = first + (int)(k - 1);
	return p > E.buckets[b].hi ? -1 : b;                     // right of the bucket's last span: no event can want the read
}

} // namespace

int ingest_reads(const lsq_events &E, const lsq_reads &R, int n_threads, PooledReads &out) {
	const size_t B = E.buckets.size();
	const uint64_t n = R.n_reads;
	int T = host_threads(n_threads);
	if (n < 100000) T = 1;
	// pass 1: per thread, per bucket: reads in each pool and blocks in pool n
	std::vector<std::vector<uint64_t>> cnt(T, std::vector<uint64_t>(B * 4, 0));
	std::vector<uint64_t> retained(T, 0), retained_blocks(T, 0);
	std::vector<int> bad(T, 0);
	auto range = [&](int t, uint64_t &a, uint64_t &b) { a = n * (uint64_t)t / (uint64_t)T; b = n * (uint64_t)(t + 1) / (uint64_t)T; };
	{
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
			uint64_t a, b; range(t, a, b);
			Kept k;
			for (uint64_t i = a; i < b; ++i) {
				filter_merge(E, R, i, k);
				if (!k.any || k.il.s.empty()) continue;      // empty: undefined in the reference, dropped
				++retained[t];
				size_t nb = k.il.s.size();
				retained_blocks[t] += nb;
				int64_t tot = 0;
				for (size_t q = 0; q < nb; ++q) tot += k.il.e[q] - k.il.s[q];
				if (tot >= (1 << 18) || nb > 4096) { bad[t] = 1; return; }
				int bk = bucket_of(E, k.chrom, k.il.s[0]);
				if (bk < 0) continue;
				int pool = nb == 1 ? 0 : (nb == 2 ? 1 : 2);
				++cnt[t][(size_t)bk * 4 + pool];
				if (pool == 2) cnt[t][(size_t)bk * 4 + 3] += nb;
			}
		});
		for (auto &x : th) x.join();
	}
	for (int t = 0; t < T; ++t) if (bad[t]) return fail(LSQ_E_RANGE, "a read covers 2^18 or more bases or has more than 4096 blocks: outside the device tables' range");
	out.n_retained = out.n_retained_blocks = 0;
	for (int t = 0; t < T; ++t) { out.n_retained += retained[t]; out.n_retained_blocks += retained_blocks[t]; }
	// offsets: bucket-major, then thread
	out.p1_off.assign(B + 1, 0); out.p2_off.assign(B + 1, 0); out.pn_off.assign(B + 1, 0);
	std::vector<std::vector<uint64_t>> pos(T, std::vector<uint64_t>(B * 4, 0));
	uint64_t o1 = 0, o2 = 0, on = 0, ob = 0;
	for (size_t b = 0; b < B; ++b) {
		out.p1_off[b] = o1; out.p2_off[b] = o2; out.pn_off[b] = on;
		for (int t = 0; t < T; ++t) {
			pos[t][b * 4 + 0] = o1; o1 += cnt[t][b * 4 + 0];
			pos[t][b * 4 + 1] = o2; o2 += cnt[t][b * 4 + 1];
			pos[t][b * 4 + 2] = on; on += cnt[t][b * 4 + 2];
			pos[t][b * 4 + 3] = ob; ob += cnt[t][b * 4 + 3];
		}
	}
	out.p1_off[B] = o1; out.p2_off[B] = o2; out.pn_off[B] = on;
	if (ob > 0xFFFFFFF0ull) return fail(LSQ_E_RANGE, "too many blocks in multi-block reads");
	out.p1_se.resize(o1 * 2); out.p1_strand.resize(o1); out.p1_line.resize(o1);
	out.p2_se.resize(o2 * 4); out.p2_strand.resize(o2); out.p2_line.resize(o2);
	out.pn_blk_off.resize(on + 1); out.pn_se.resize(ob * 2); out.pn_strand.resize(on); out.pn_line.resize(on); out.pn_bucket.resize(on);
	out.pn_blk_off[on] = (uint32_t)ob;
	{
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
			uint64_t a, b; range(t, a, b);
			Kept k;
			std::vector<uint64_t> &p = pos[t];
			for (uint64_t i = a; i < b; ++i) {
				filter_merge(E, R, i, k);
				if (!k.any || k.il.s.empty()) continue;
				int bk = bucket_of(E, k.chrom, k.il.s[0]);
				if (bk < 0) continue;
				size_t nb = k.il.s.size();
				if (nb == 1) {
					uint64_t w = p[(size_t)bk * 4 + 0]++;
					out.p1_se[2 * w] = (int32_t)k.il.s[0]; out.p1_se[2 * w + 1] = (int32_t)k.il.e[0];
					out.p1_strand[w] = (uint8_t)k.strand; out.p1_line[w] = R.line_no[i];
				} else if (nb == 2) {
					uint64_t w = p[(size_t)bk * 4 + 1]++;
					out.p2_se[4 * w] = (int32_t)k.il.s[0]; out.p2_se[4 * w + 1] = (int32_t)k.il.e[0];
					out.p2_se[4 * w + 2] = (int32_t)k.il.s[1]; out.p2_se[4 * w + 3] = (int32_t)k.il.e[1];
					out.p2_strand[w] = (uint8_t)k.strand; out.p2_line[w] = R.line_no[i];
				} else {
					uint64_t w = p[(size_t)bk * 4 + 2]++;
					uint64_t bo = p[(size_t)bk * 4 + 3];
					p[(size_t)bk * 4 + 3] += nb;
					out.pn_blk_off[w] = (uint32_t)bo;
					for (size_t q = 0; q < nb; ++q) { out.pn_se[2 * (bo + q)] = (int32_t)k.il.s[q]; out.pn_se[2 * (bo + q) + 1] = (int32_t)k.il.e[q]; }
					out.pn_strand[w] = (uint8_t)k.strand; out.pn_line[w] = R.line_no[i]; out.pn_bucket[w] = (uint32_t)bk;
				}
			}
		});
		for (auto &x : th) x.join();
	}
	return LSQ_OK;
}

} // namespace lsq
