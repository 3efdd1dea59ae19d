// Device group of the C ABI: context, event tables, count / solve entry points, result fetch (gfx950).
// The kernels live in lsq_count.hip, lsq_em.hip and lsq_ingest.hip.
#include "lsq_device.hpp"
#include <chrono>
#include <thread>

namespace lsq {

int upload_strand_ranks(lsq_ctx *c) {
	const auto &names = c->E->strands.names;
	if (names.size() > 256) return fail(LSQ_E_RANGE, "more than 256 distinct strand strings");
	std::vector<int> order(names.size());
	std::iota(order.begin(), order.end(), 0);
	std::sort(order.begin(), order.end(), [&](int a, int b) { return names[a] < names[b]; });
	std::vector<uint8_t> rank(256, 0);
	for (size_t r = 0; r < order.size(); ++r) rank[order[r]] = (uint8_t)r;
	int rc = c->strand_rank.upload(rank.data(), 256, c->stream);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return LSQ_OK;
}

// A blocking wait wakes up tens of microseconds after the stream has drained; a step is 0.04-0.14 ms.  So the stream is
// polled for its first two milliseconds (a few steps' worth) and only a longer wait goes to sleep.
static hipError_t wait_for_stream(hipStream_t s) {
	const auto t0 = std::chrono::steady_clock::now();
	for (;;) {
		const hipError_t e = hipStreamQuery(s);
		if (e != hipErrorNotReady) return e;
		if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipStreamSynchronize(s);
	}
}

// the lanes' streams and events exist (a context made with LSQ_CTX_LANES_IN_BACKGROUND: its helper thread is joined here, once)
int ensure_lanes(lsq_ctx *c) {
	if (c->lanes_thread.joinable()) c->lanes_thread.join();
	if (c->lanes_status) return fail(c->lanes_status, "%s", c->lanes_error.c_str());
	return LSQ_OK;
}

int sync_all(lsq_ctx *c) {
	{ const int rc = ensure_lanes(c); if (rc) return rc; }
	HIP_TRY(wait_for_stream(c->stream_count2[0]));
	HIP_TRY(wait_for_stream(c->stream_count2[1]));
	HIP_TRY(wait_for_stream(c->stream_em2[0]));
	HIP_TRY(wait_for_stream(c->stream_em2[1]));
	HIP_TRY(wait_for_stream(c->stream));
	return LSQ_OK;
}

void select_counter_set(lsq_ctx *c, int set) {
	unsigned long long *base = c->counters.p + (size_t)set * c->counters_per_set;
	const size_t per = c->cnt.n;
	c->flip = set;
	c->stream_em = c->stream_em2[set];
	c->theta.p = c->theta2[set].p; c->theta.n = c->theta2[set].n;
	c->logll.p = c->logll2[set].p; c->logll.n = c->logll2[set].n;
	c->iters.p = c->iters2[set].p; c->iters.n = c->iters2[set].n;
	c->flags.p = c->flags2[set].p; c->flags.n = c->flags2[set].n;
	c->cnt.p = base;
	c->bases.p = base + per;
	c->exc_count.p = reinterpret_cast<unsigned *>(base + 2 * per);
	c->dbg.p = base + 2 * per + LSQ_MAX_METHODS;
}

// An exception list that overflowed is no error -- the exception pass then counts every read of the file again, on the device,
// and the tables are whole -- but it costs a count of its own per step: the host says so the first time it looks at that count
// (lsq_count_status, lsq_results_counts), with what to raise (lsq_ctx_set_option "exception_capacity").
void note_overflow(lsq_ctx *c, const std::vector<unsigned> &h) {
	for (int m = 0; m < c->E->n_methods; ++m) {
		if (!h[2 * (size_t)m + 1] || c->overflow_logged[m]) continue;
		c->overflow_logged[m] = true;
		if (c->opt_recount) continue;            // asked for (option recount_every_read): nothing to report
		warn("read file %d: the exception list overflowed (%u pairs for %zu entries); every read of the file was counted again on the device. "
		     "Tables are complete; raise lsq_ctx_set_option \"exception_capacity\" to avoid the second pass", m, h[2 * (size_t)m], c->reads[m].exc_cap);
	}
}

} // namespace lsq

extern "C" {

int lsq_ctx_create(int device_id, lsq_ctx **out) { return lsq_ctx_create_with(device_id, 0u, out); }

int lsq_ctx_create_with(int device_id, unsigned flags, lsq_ctx **out) LSQ_API_TRY {
	if (!out) return fail(LSQ_E_ARG, "null argument");
	// developer aid: LSQ_CLI_TIMING=1 prints where the start-up goes (stderr)
	const bool timing = getenv("LSQ_CLI_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) {
		if (!timing) return;
		const auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[timing]     %-32s %.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
		t_last = t;
	};
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	mark("context: runtime start-up");
	if (e != hipSuccess || n <= 0) return fail(LSQ_E_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	if (device_id < 0 || device_id >= n) return fail(LSQ_E_ARG, "device %d out of range (%d devices)", device_id, n);
	HIP_TRY(hipSetDevice(device_id));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device_id));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(LSQ_E_DEVICE, "device %d is %s; this library carries gfx950 code only", device_id, prop.gcnArchName);
	mark("context: device selected");
	std::unique_ptr<lsq_ctx> c(new lsq_ctx);
	c->device = device_id;
	c->n_cu = prop.multiProcessorCount;
	HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	HIP_TRY(hipEventCreate(&c->evt0)); HIP_TRY(hipEventCreate(&c->evt1));
	mark("context: upload stream");
	// The rest -- the two lanes' streams and events, 0.04 s of the runtime's time -- is not needed by what an executable does first with
	// its context (copying the MRF text to HBM, lsq_text_stage): with LSQ_CTX_LANES_IN_BACKGROUND a helper thread makes it meanwhile,
	// and the first call that wants a lane (lsq_events_upload is ahead of every one of them) joins it.
	lsq_ctx *cp = c.get();
	auto make_lanes = [cp]() -> int {
		HIP_TRY(hipSetDevice(cp->device));
		for (int l = 0; l < 2; ++l) {
			HIP_TRY(hipStreamCreateWithFlags(&cp->stream_em2[l], hipStreamNonBlocking));      // (a higher stream priority changed nothing measurable)
			HIP_TRY(hipStreamCreateWithFlags(&cp->stream_count2[l], hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&cp->ev_counted2[l], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&cp->ev_mark2[l], hipEventDisableTiming));
		}
		cp->stream_em = cp->stream_em2[0];
		HIP_TRY(hipEventCreate(&cp->ev0)); HIP_TRY(hipEventCreate(&cp->ev1));
		HIP_TRY(hipEventCreate(&cp->ev2)); HIP_TRY(hipEventCreate(&cp->ev3));
		for (int m = 0; m < LSQ_MAX_METHODS; ++m) { HIP_TRY(hipEventCreate(&cp->evf0[m])); HIP_TRY(hipEventCreate(&cp->evf1[m])); }
		return LSQ_OK;
	};
	if (flags & LSQ_CTX_LANES_IN_BACKGROUND) {
		c->lanes_thread = std::thread([cp, make_lanes]() noexcept {
			try { cp->lanes_status = make_lanes(); if (cp->lanes_status) cp->lanes_error = lsq_last_error(); }
			catch (...) { cp->lanes_status = LSQ_E_INTERNAL; try { cp->lanes_error = "the lanes' helper thread failed"; } catch (...) {} }
		});
	} else {
		const int rc = make_lanes();
		if (rc) return rc;
		mark("context: lanes");
	}
	*out = c.release();
	return LSQ_OK;
} LSQ_API_CATCH

void lsq_ctx_destroy(lsq_ctx *c) {
	if (!c) return;
	if (c->lanes_thread.joinable()) c->lanes_thread.join();
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	for (int l = 0; l < 2; ++l) {
		if (c->stream_em2[l]) (void)hipStreamSynchronize(c->stream_em2[l]);
		if (c->stream_count2[l]) (void)hipStreamSynchronize(c->stream_count2[l]);
		if (c->ev_counted2[l]) (void)hipEventDestroy(c->ev_counted2[l]);
		if (c->ev_mark2[l]) (void)hipEventDestroy(c->ev_mark2[l]);
	}
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->ev2) (void)hipEventDestroy(c->ev2);
	if (c->ev3) (void)hipEventDestroy(c->ev3);
	if (c->evt0) (void)hipEventDestroy(c->evt0);
	if (c->evt1) (void)hipEventDestroy(c->evt1);
	for (int m = 0; m < LSQ_MAX_METHODS; ++m) { if (c->evf0[m]) (void)hipEventDestroy(c->evf0[m]); if (c->evf1[m]) (void)hipEventDestroy(c->evf1[m]); }
	for (hipEvent_t e : c->ing_ev) if (e) (void)hipEventDestroy(e);
	for (int q = 0; q < 2; ++q) { if (c->pin_buf[q]) (void)hipHostFree(c->pin_buf[q]); if (c->pin_ev[q]) (void)hipEventDestroy(c->pin_ev[q]); }
	if (c->stream) (void)hipStreamDestroy(c->stream);
	for (int l = 0; l < 2; ++l) { if (c->stream_em2[l]) (void)hipStreamDestroy(c->stream_em2[l]); if (c->stream_count2[l]) (void)hipStreamDestroy(c->stream_count2[l]); }
	delete c;
}

void *lsq_ctx_stream(lsq_ctx *c) { return c ? (void *)c->stream : nullptr; }
int lsq_ctx_synchronize(lsq_ctx *c) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	HIP_TRY(hipSetDevice(c->device));
	return sync_all(c);
} LSQ_API_CATCH

int lsq_ctx_synchronize_for(lsq_ctx *c, double seconds) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	HIP_TRY(hipSetDevice(c->device));
	{ const int rc = ensure_lanes(c); if (rc) return rc; }
	const auto t0 = std::chrono::steady_clock::now();
	const hipStream_t all[5] = {c->stream_count2[0], c->stream_count2[1], c->stream_em2[0], c->stream_em2[1], c->stream};
	for (const hipStream_t s : all) {
		for (unsigned spins = 0;; ++spins) {
			const hipError_t e = hipStreamQuery(s);
			if (e == hipSuccess) break;
			if (e != hipErrorNotReady) return fail(LSQ_E_DEVICE, "hipStreamQuery: %s", hipGetErrorString(e));
			if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds)
				return fail(LSQ_E_TIMEOUT, "the context's streams had not drained after %.1f s", seconds);
			if (spins > 4096) std::this_thread::sleep_for(std::chrono::microseconds(200));       // a long wait need not burn a core
		}
	}
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_events_upload(lsq_ctx *c, lsq_events *E) LSQ_API_TRY {
	if (!c || !E) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }        // the buffers below may still be read by a solve in flight
	if (E->max_lds_bytes > 160 * 1024) return fail(LSQ_E_UNSUPPORTED, "bucket tables exceed the CU's LDS");
	c->E = E;
	c->counted = c->solved = false;
	c->has_fast = c->has_generic = c->has_host = false;
	for (const BucketDesc &bd : E->buckets) { if (bd.kind == 1) c->has_fast = true; else if (bd.kind == 0) c->has_generic = true; else c->has_host = true; }
	std::vector<bool> host_event(E->dev2out.size(), false);
	for (const BucketDesc &bd : E->buckets) if (bd.kind == 2) for (uint32_t q = 0; q < bd.n_events; ++q) host_event[bd.ev_base + q] = true;
	for (auto &r : c->reads) r.present = false;
	int rc;
	const bool timing = getenv("LSQ_CLI_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) {
		if (!timing) return;
		const auto t = std::chrono::steady_clock::now();
		fprintf(stderr, "[timing]     %-32s %.4f s\n", what, std::chrono::duration<double>(t - t_last).count());
		t_last = t;
	};
	if ((rc = c->buckets.upload(E->buckets.data(), E->buckets.size(), c->stream))) return rc;
	if ((rc = c->images.upload(E->images.data(), E->images.size(), c->stream))) return rc;
	if ((rc = c->ties.upload(E->ties.data(), E->ties.size(), c->stream))) return rc;
	if ((rc = c->dK.upload(E->dev_K.data(), E->dev_K.size(), c->stream))) return rc;
	if ((rc = c->cls_base.upload(E->dev_cls_base.data(), E->dev_cls_base.size(), c->stream))) return rc;
	if ((rc = c->iso_base.upload(E->dev_iso_base.data(), E->dev_iso_base.size(), c->stream))) return rc;
	{
		// EM places: events with at most two isoforms and one (method, class) pair per lane first, then
		// the others; each group padded to whole waves (16 events) so that a wave runs one loop shape
		const size_t n_dev = E->dev2out.size();
		std::vector<uint32_t> order;
		order.reserve(n_dev + 32);
		for (int pass = 0; pass < 2; ++pass) {
			for (size_t d = 0; d < n_dev; ++d) {
				const int K = E->dev_K[d];
				if (host_event[d]) continue;            // solved on the host (host_solve)
				const bool small = K <= 2 && E->n_methods * ((1 << K) - 1) <= EM_LANES;
				if (small == (pass == 0)) order.push_back((uint32_t)d);
			}
			while (order.size() % (64 / EM_LANES)) order.push_back(0xFFFFFFFFu);
			if (pass == 0) c->em_small_places = (unsigned)order.size();
		}
		c->em_places = (unsigned)order.size();
		c->em_order_lane_valid[0] = c->em_order_lane_valid[1] = false;
		{
			const size_t np = std::max<size_t>(c->em_small_places, 1);
			if ((rc = c->em_tail_count.alloc(4))) return rc;
			HIP_TRY(hipMemsetAsync(c->em_tail_count.p, 0, 4 * sizeof(uint32_t), c->stream));
			for (int l = 0; l < 2; ++l)
				if ((rc = c->em_tail_u32[l].alloc(2 * np)) || (rc = c->em_tail_flag[l].alloc(np)) || (rc = c->em_tail_f64[l].alloc(3 * np))) return rc;
		}
		if (c->em_split.n != 3) { if ((rc = c->em_split.alloc(3))) return rc; }
		HIP_TRY(hipMemsetAsync(c->em_split.p, 0xFF, 3 * sizeof(uint32_t), c->stream));
		if ((rc = c->em_order.upload(order.data(), order.size(), c->stream))) return rc;
		// gene names for span-start ties against named reads
		std::string blob;
		std::vector<uint32_t> goff(n_dev + 1, 0);
		for (size_t d = 0; d < n_dev; ++d) { blob += E->ev[E->dev2out[d]].gname; goff[d + 1] = (uint32_t)blob.size(); }
		if ((rc = c->gene_names.upload(blob.data(), blob.size(), c->stream))) return rc;
		if ((rc = c->gene_name_off.upload(goff.data(), goff.size(), c->stream))) return rc;
	}
	{
		// lsq_results_pack_device: where every class slot, isoform and event of the shard sits in the device arrays,
		// listed in output order (the shard is a contiguous run of output-ordered events)
		const size_t n_dev = E->dev2out.size();
		std::vector<uint32_t> order(n_dev);
		std::iota(order.begin(), order.end(), 0u);
		std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return E->dev2out[a] < E->dev2out[b]; });
		std::vector<uint32_t> pc, pi, pe;
		for (uint32_t dd : order) {
			const int K = E->dev_K[dd];
			for (uint32_t k = 0; k < (1u << K) - 1u; ++k) pc.push_back(E->dev_cls_base[dd] + k);
			for (int j = 0; j < K; ++j) pi.push_back(E->dev_iso_base[dd] + (uint32_t)j);
			pe.push_back(dd);
		}
		if ((rc = c->pack_cls.upload(pc.data(), pc.size(), c->stream))) return rc;
		if ((rc = c->pack_iso.upload(pi.data(), pi.size(), c->stream))) return rc;
		if ((rc = c->pack_ev.upload(pe.data(), pe.size(), c->stream))) return rc;
	}
	mark("events: images, EM order, pack lists");
	// G = 1/ARS (common/read.h:331-340), device isoform order, per method
	const size_t n_iso = E->n_iso_total, M = (size_t)E->n_methods;
	std::vector<double> G(std::max<size_t>(M * n_iso, 1), 0.0);
	for (size_t d = 0; d < E->dev2out.size(); ++d) {
		const Event &ev = E->ev[E->dev2out[d]];
		for (size_t m = 0; m < M; ++m)
			for (int j = 0; j < ev.K; ++j) {
				double nd = (double)ev.ars[m][j];
				G[m * n_iso + E->dev_iso_base[d] + j] = nd <= 0 ? 0.0 : (double)1.0 / nd;
			}
	}
	if ((rc = c->G.upload(G.data(), M * n_iso, c->stream))) return rc;
	const size_t n_cls = E->n_cls_total, n_ev = E->dev2out.size();
	{
		const size_t per = std::max<size_t>(M, 1) * n_cls;
		c->counters_per_set = 2 * per + LSQ_MAX_METHODS + 16 + LSQ_MAX_METHODS;          // cnt | bases | exc_count (two words a read file) | dbg | one barrier word a read file (lsq_count_cleanup_kernel)
		if ((rc = c->counters.alloc(2 * c->counters_per_set))) return rc;
		c->cnt.n = per; c->bases.n = per; c->exc_count.n = 2 * LSQ_MAX_METHODS; c->dbg.n = 16;
		c->mark_recorded2[0] = c->mark_recorded2[1] = false;
		c->fim_uploaded = false; c->fim_done = false;
		select_counter_set(c, 0);
		HIP_TRY(hipMemsetAsync(c->counters.p, 0, c->counters.n * sizeof(unsigned long long), c->stream));      // both sets start out zero
	}
	for (int l = 0; l < 2; ++l) {
		if ((rc = c->theta2[l].alloc(n_iso)) || (rc = c->logll2[l].alloc(n_ev)) || (rc = c->iters2[l].alloc(n_ev)) || (rc = c->flags2[l].alloc(n_ev))) return rc;
		HIP_TRY(hipMemsetAsync(c->flags2[l].p, 0, std::max<size_t>(n_ev, 1), c->stream));
	}
	select_counter_set(c, 0);
	mark("events: G, counters, EM outputs");
	{
		// ingest tables (lsq_device.hpp: RouteChrom): per chromosome id the covered regions (count/count.cpp:244) as (start, end)
		// pairs, the spans of the planned events ("clusters") with the bucket each lies in, and the locator grid over both
		const size_t nc = E->covered.size();
		std::vector<RouteChrom> chrom(std::max<size_t>(nc, 1));
		std::vector<int2> cov;
		std::vector<int4> clu;
		for (size_t ch = 0; ch < nc; ++ch) {
			RouteChrom &R = chrom[ch];
			memset(&R, 0, sizeof(R));
			R.cov0 = (unsigned)cov.size();
			for (size_t q = 0; q < E->covered[ch].s.size(); ++q) cov.push_back(make_int2((int)E->covered[ch].s[q], (int)E->covered[ch].e[q]));
			R.cov1 = (unsigned)cov.size();
			R.clu0 = (unsigned)clu.size();
			// A cluster with the bucket of its bases.  Buckets are cut where no span crosses, so a cluster lies in one; the records are
			// cut at the bucket cuts all the same, and a stretch left of the chromosome's first cut (in no bucket) gets none: a
			// look-up of "the last record that starts at or left of p" then IS the bucket search and the cluster test of p.
			const int first = ch < E->chrom_first_bucket.size() ? E->chrom_first_bucket[ch] : -1;
			if (first >= 0 && ch < E->clu_s.size() && ch < E->cut_lo.size()) {
				const std::vector<int32_t> &cuts = E->cut_lo[ch];
				for (size_t q = 0; q < E->clu_s[ch].size(); ++q) {
					long long s = E->clu_s[ch][q];
					const long long e = E->clu_e[ch][q];           // (inclusive: a read whose first base is e is still inside)
					while (s <= e) {
						const size_t ub = (size_t)(std::upper_bound(cuts.begin(), cuts.end(), (int32_t)s) - cuts.begin());
						const long long next_cut = ub < cuts.size() ? (long long)cuts[ub] : e + 1;
						const long long stop = std::min(e, next_cut - 1);
						if (ub > 0) {
							const unsigned b = (unsigned)first + (unsigned)(ub - 1);
							clu.push_back(make_int4((int)s, (int)std::min<long long>(stop, E->buckets[b].hi), (int)b, E->buckets[b].lo));
						}
						s = stop + 1;
					}
				}
			}
			R.clu1 = (unsigned)clu.size();
		}
		c->n_chrom_tables = (unsigned)nc;
		{
			// the locator: bins of 2^shift bases over each chromosome's range of starts (covered regions and cluster records), the
			// shift raised until all chromosomes together take at most 2^22 entries.  Entry k of a chromosome, for the bin's first
			// base x and the next bin's first base y: lower bounds of x and of y among the covered starts (.x, .y) and among the
			// cluster starts: entry k holds the lower bounds of x (.x covered, .y clusters), entry k + 1 those of y -- read as one
			// 16-byte pair, a search starts one load away from a handful of candidates; eight bytes a bin keep the bins of the
			// events (where the reads are) inside the L2.
			unsigned shift = 10;
			std::vector<long long> lo(nc, 0), hi(nc, -1);
			for (size_t ch = 0; ch < nc; ++ch) {
				bool any = false;
				auto see = [&](long long v) { if (!any) { lo[ch] = hi[ch] = v; any = true; } lo[ch] = std::min(lo[ch], v); hi[ch] = std::max(hi[ch], v); };
				for (unsigned q = chrom[ch].cov0; q < chrom[ch].cov1; ++q) see(cov[q].x);
				for (unsigned q = chrom[ch].clu0; q < chrom[ch].clu1; ++q) see(clu[q].x);
			}
			for (;; ++shift) {
				unsigned long long total = 0;
				for (size_t ch = 0; ch < nc; ++ch) if (hi[ch] >= lo[ch]) total += (unsigned long long)(((hi[ch] >> shift) - (lo[ch] >> shift)) + 2);
				if (total <= (1ull << 22) || shift >= 30) break;
			}
			std::vector<uint2> loc;
			for (size_t ch = 0; ch < nc; ++ch) {
				RouteChrom &R = chrom[ch];
				R.loc_first = (unsigned)loc.size();
				if (hi[ch] < lo[ch]) continue;
				const long long base = (lo[ch] >> shift) << shift;          // (arithmetic shift: rounds towards minus infinity)
				const long long nb = ((hi[ch] - base) >> shift) + 1;
				R.loc_base = (int)base; R.loc_nb = (unsigned)nb;
				unsigned a = R.cov0, u = R.clu0;
				for (long long k = 0; k <= nb; ++k) {               // (nb + 1 entries: a bin's upper bounds are the next entry's lower bounds)
					const long long x = base + (k << shift);
					while (a < R.cov1 && cov[a].x < x) ++a;
					while (u < R.clu1 && clu[u].x < x) ++u;
					loc.push_back(make_uint2(a, u));
				}
			}
			loc.push_back(make_uint2(0, 0)); loc.push_back(make_uint2(0, 0));      // (an entry is read as a pair with its successor)
			if (cov.empty()) cov.push_back(make_int2(0, 0));
			if (clu.empty()) clu.push_back(make_int4(0, 0, 0, 0));
			c->loc_shift = shift;
			if ((rc = c->loc.upload(loc.data(), loc.size(), c->stream))) return rc;
			if ((rc = c->route_chrom.upload(chrom.data(), chrom.size(), c->stream))) return rc;
			if ((rc = c->cov.upload(cov.data(), cov.size(), c->stream))) return rc;
			if ((rc = c->clu.upload(clu.data(), clu.size(), c->stream))) return rc;
			HIP_TRY(hipStreamSynchronize(c->stream));          // the host vectors go out of scope
		}
		mark("events: loader tables + locator");
		std::vector<unsigned> bb(E->buckets.size() + 1, 0);
		for (size_t b = 0; b < E->buckets.size(); ++b) bb[b + 1] = bb[b] + E->buckets[b].n_bins;
		c->n_fine = bb.back();
		if ((rc = c->bin_base.upload(bb.data(), bb.size(), c->stream))) return rc;
		std::vector<unsigned> cb(E->buckets.size() + 1, 0);
		for (size_t b = 0; b < E->buckets.size(); ++b) cb[b + 1] = cb[b] + (E->buckets[b].kind == 1 ? (E->buckets[b].iso_off & 0xFFFFu) : 0u) + 1u;
		c->n_cell_groups = cb.back();
		if ((rc = c->cell_base.upload(cb.data(), cb.size(), c->stream))) return rc;
		std::vector<unsigned> jb(E->buckets.size() + 1, 0);
		// the bucket's junction groups, then -- round 3 -- one group per cell (and one for "no cell") for the two-block reads
		// that cross no junction of the annotation: a quadruple of those shares the cell of its first record too
		for (size_t b = 0; b <= E->buckets.size(); ++b) jb[b] = E->jg_base[b] + cb[b];
		c->n_junction_groups = jb.back();
		const unsigned long long none = 0;
		if ((rc = c->jg_keys.upload(E->jg_keys.empty() ? &none : (const unsigned long long *)E->jg_keys.data(), std::max<size_t>(E->jg_keys.size(), 1), c->stream))) return rc;
		if ((rc = c->jg_base.upload(E->jg_base.data(), E->jg_base.size(), c->stream))) return rc;
		if ((rc = c->jgroup_base.upload(jb.data(), jb.size(), c->stream))) return rc;
	}
	if ((rc = upload_strand_ranks(c))) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	mark("events: group bases, strand ranks");
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_count(lsq_ctx *c) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	HIP_TRY(hipSetDevice(c->device));
	for (int m = 0; m < c->E->n_methods; ++m) if (!c->reads[m].present) return fail(LSQ_E_STATE, "reads of method %d were not uploaded", m);
#ifdef LSQ_DEV
	if (const char *e = getenv("LSQ_ABLATE")) c->dev_ablate = (unsigned)atoi(e); else c->dev_ablate = 0;
	if (const char *e = getenv("LSQ_GRID_MULT")) { const int v = atoi(e); if (v >= 1 && v <= 64) c->opt_grid_mult = v; }
#endif
	int rc = run_count(c);
	if (rc) return rc;
	for (bool &f : c->overflow_logged) f = false;
	if (c->has_host && (rc = host_count(c))) return rc;        // genes beyond the kernels' limits: evaluated here (blocks)
	c->counted = true;
	c->solved = false;
	c->counts_external = false;
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_solve(lsq_ctx *c) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	HIP_TRY(hipSetDevice(c->device));
	int rc = run_solve(c);
	if (rc) return rc;
	if (c->has_host && c->counts_external) return fail(LSQ_E_UNSUPPORTED, "genes beyond the kernels' limits are solved from their reads, which counts set from outside do not come with");
	if (c->has_host && (rc = host_solve(c))) return rc;
	c->solved = true;
	c->fim_done = false;
	return LSQ_OK;
} LSQ_API_CATCH

// fim.h / linalg.h (parity unpinned, see lsq_em.hip): needs lsq_solve's theta
int lsq_fim(lsq_ctx *c) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->solved) return fail(LSQ_E_STATE, "lsq_solve must come first");
	HIP_TRY(hipSetDevice(c->device));
	const lsq_events &E = *c->E;
	const size_t n_ev = E.dev2out.size(), M = (size_t)E.n_methods;
	if (!c->fim_uploaded) {
		std::vector<uint32_t> sb(n_ev + 1, 0), mb(n_ev + 1, 0);
		for (size_t d = 0; d < n_ev; ++d) {
			const lsq::Event &e = E.ev[(size_t)E.dev2out[d]];
			if (e.K > LSQ_MAX_ISOFORMS || e.fim_starts.empty()) return fail(LSQ_E_UNSUPPORTED, "lsq_fim: event %s has more than %d isoforms", e.gname.c_str(), LSQ_MAX_ISOFORMS);
			sb[d + 1] = sb[d] + (uint32_t)((size_t)e.K * (((size_t)1 << e.K) - 1));
			mb[d + 1] = mb[d] + (uint32_t)((e.K - 1) * (e.K - 1));
		}
		c->fim_starts_total = sb[n_ev]; c->fim_mat_total = mb[n_ev];
		std::vector<uint32_t> st(std::max<size_t>(M * c->fim_starts_total, 1), 0);
		for (size_t m = 0; m < M; ++m)
			for (size_t d = 0; d < n_ev; ++d) {
				const lsq::Event &e = E.ev[(size_t)E.dev2out[d]];
				std::copy(e.fim_starts[m].begin(), e.fim_starts[m].end(), st.begin() + (ptrdiff_t)(m * c->fim_starts_total + sb[d]));
			}
		int rc;
		if ((rc = c->fim_start_base.upload(sb.data(), sb.size(), c->stream))) return rc;
		if ((rc = c->fim_mat_base.upload(mb.data(), mb.size(), c->stream))) return rc;
		if ((rc = c->fim_starts.upload(st.data(), st.size(), c->stream))) return rc;
		if ((rc = c->fim.alloc(std::max<size_t>(M * c->fim_mat_total, 1)))) return rc;
		if ((rc = c->fim_var.alloc(std::max<size_t>(M * n_ev * 2, 1)))) return rc;
		HIP_TRY(hipStreamSynchronize(c->stream));
		c->fim_uploaded = true;
	}
	int rc = run_fim(c);
	if (rc) return rc;
	c->fim_done = true;
	return LSQ_OK;
} LSQ_API_CATCH

int64_t lsq_results_fim_size(const lsq_ctx *c) {
	if (!c || !c->E) return 0;
	int64_t n = 0;
	for (const lsq::Event &e : c->E->ev) n += (int64_t)(e.K - 1) * (e.K - 1);
	return n;
}

int lsq_results_fim_offsets(const lsq_ctx *c, uint64_t *fim_off) LSQ_API_TRY {
	if (!c || !c->E || !fim_off) return fail(LSQ_E_ARG, "null argument");
	uint64_t n = 0;
	for (size_t o = 0; o < c->E->ev.size(); ++o) { fim_off[o] = n; n += (uint64_t)(c->E->ev[o].K - 1) * (uint64_t)(c->E->ev[o].K - 1); }
	fim_off[c->E->ev.size()] = n;
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_results_fim(lsq_ctx *c, double *fim, double *var_by_diag, double *var_by_inverse) LSQ_API_TRY {
	if (!c || !fim || !var_by_diag || !var_by_inverse) return fail(LSQ_E_ARG, "null argument");
	if (!c->fim_done || !c->solved) return fail(LSQ_E_STATE, "lsq_fim must come first");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	const lsq_events &E = *c->E;
	const size_t n_ev = E.dev2out.size(), n_out = E.ev.size(), M = (size_t)E.n_methods;
	std::vector<double> hf(std::max<size_t>(M * c->fim_mat_total, 1)), hv(std::max<size_t>(M * n_ev * 2, 1));
	std::vector<uint32_t> mb(n_ev + 1, 0);
	if (n_ev) {
		HIP_TRY(hipMemcpy(hf.data(), c->fim.p, hf.size() * sizeof(double), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hv.data(), c->fim_var.p, hv.size() * sizeof(double), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(mb.data(), c->fim_mat_base.p, mb.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
	}
	std::vector<uint64_t> off(n_out + 1);
	lsq_results_fim_offsets(c, off.data());
	const size_t total = (size_t)off[n_out];
	for (size_t i = 0; i < M * total; ++i) fim[i] = 0.0;
	for (size_t i = 0; i < M * n_out; ++i) { var_by_diag[i] = 0.0; var_by_inverse[i] = 0.0; }       // events outside the shard: zeros
	for (size_t d = 0; d < n_ev; ++d) {
		const size_t o = (size_t)E.dev2out[d];
		const size_t nn = (size_t)(E.ev[o].K - 1) * (size_t)(E.ev[o].K - 1);
		for (size_t m = 0; m < M; ++m) {
			for (size_t i = 0; i < nn; ++i) fim[m * total + off[o] + i] = hf[m * c->fim_mat_total + mb[d] + i];
			var_by_diag[m * n_out + o] = hv[(m * n_ev + d) * 2];
			var_by_inverse[m * n_out + o] = hv[(m * n_ev + d) * 2 + 1];
		}
	}
	return LSQ_OK;
} LSQ_API_CATCH

int64_t lsq_results_num_classes(const lsq_ctx *c) { return (c && c->E) ? (int64_t)c->E->class_off.back() : 0; }

int lsq_results_class_offsets(const lsq_ctx *c, uint64_t *class_off) LSQ_API_TRY {
	if (!c || !c->E || !class_off) return fail(LSQ_E_ARG, "null argument");
	memcpy(class_off, c->E->class_off.data(), c->E->class_off.size() * sizeof(uint64_t));
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_results_counts(lsq_ctx *c, uint64_t *class_count, uint64_t *class_bases) LSQ_API_TRY {
	if (!c || !class_count) return fail(LSQ_E_ARG, "null argument");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	HIP_TRY(hipSetDevice(c->device));
	const lsq_events &E = *c->E;
	// device arrays hold the classes of this process's events (device order); the caller's arrays
	// hold every selected event's classes (output order), zero outside the shard
	const size_t n_cls = E.n_cls_total, n_out = (size_t)E.class_off.back(), M = (size_t)E.n_methods;
	std::vector<unsigned long long> hc(std::max<size_t>(M * n_cls, 1)), hb(std::max<size_t>(M * n_cls, 1));
	{ int rc = sync_all(c); if (rc) return rc; }
	if (!c->counts_external && c->exc_count.n) {
		std::vector<unsigned> h(c->exc_count.n);
		HIP_TRY(hipMemcpy(h.data(), c->exc_count.p, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
		note_overflow(c, h);
	}
	if (M * n_cls) {
		HIP_TRY(hipMemcpy(hc.data(), c->cnt.p, M * n_cls * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hb.data(), c->bases.p, M * n_cls * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	}
	memset(class_count, 0, M * n_out * sizeof(uint64_t));
	if (class_bases) memset(class_bases, 0, M * n_out * sizeof(uint64_t));
	for (size_t d = 0; d < E.dev2out.size(); ++d) {
		const size_t o = (size_t)E.dev2out[d];
		const size_t nc = (1u << E.ev[o].K) - 1u;
		for (size_t m = 0; m < M; ++m)
			for (size_t k = 0; k < nc; ++k) {
				class_count[m * n_out + E.class_off[o] + k] = hc[m * n_cls + E.dev_cls_base[d] + k];
				if (class_bases) class_bases[m * n_out + E.class_off[o] + k] = hb[m * n_cls + E.dev_cls_base[d] + k];
			}
	}
	return LSQ_OK;
} LSQ_API_CATCH

// The inverse of lsq_results_counts: class counts and matched bases (output order, as that call returns
// them) become the context's counts -- e.g. the sums over several processes that each counted a slice
// of the reads (lesseq_amd/dist.py::run_read_sharded); lsq_solve then runs on them.
int lsq_results_set_counts(lsq_ctx *c, const uint64_t *class_count, const uint64_t *class_bases) LSQ_API_TRY {
	if (!c || !class_count || !class_bases) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	HIP_TRY(hipSetDevice(c->device));
	const lsq_events &E = *c->E;
	const size_t n_cls = E.n_cls_total, n_out = (size_t)E.class_off.back(), M = (size_t)E.n_methods;
	std::vector<unsigned long long> hc(std::max<size_t>(M * n_cls, 1), 0), hb(std::max<size_t>(M * n_cls, 1), 0);
	for (size_t d = 0; d < E.dev2out.size(); ++d) {
		const size_t o = (size_t)E.dev2out[d];
		const size_t nc = (1u << E.ev[o].K) - 1u;
		for (size_t m = 0; m < M; ++m)
			for (size_t k = 0; k < nc; ++k) {
				hc[m * n_cls + E.dev_cls_base[d] + k] = class_count[m * n_out + E.class_off[o] + k];
				hb[m * n_cls + E.dev_cls_base[d] + k] = class_bases[m * n_out + E.class_off[o] + k];
			}
	}
	{ int rc = sync_all(c); if (rc) return rc; }
	if (M * n_cls) {
		HIP_TRY(hipMemcpyAsync(c->cnt.p, hc.data(), M * n_cls * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->bases.p, hb.data(), M * n_cls * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	c->counted = true; c->solved = false;
	c->counts_external = true;
	return LSQ_OK;
} LSQ_API_CATCH

// The latest count's class counts and matched bases as they lie on the device -- [file][device class] counts, then the same
// for the bases -- for an exchange that never leaves HBM (the read-sharded run sums them over the ranks: every rank has
// the same events, hence the same order).
uint64_t lsq_counts_device_words(const lsq_ctx *c) { return (c && c->E) ? 2ull * (uint64_t)c->E->n_methods * (uint64_t)c->E->n_cls_total : 0; }

int lsq_counts_export_device(lsq_ctx *c, void *d_words) LSQ_API_TRY {
	if (!c || !d_words) return fail(LSQ_E_ARG, "null argument");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	HIP_TRY(hipSetDevice(c->device));
	const size_t per = (size_t)c->E->n_methods * c->E->n_cls_total;
	if (per) {         // on the result stream, behind the count's exception pass
		HIP_TRY(hipMemcpyAsync(d_words, c->cnt.p, per * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream_em));
		HIP_TRY(hipMemcpyAsync((unsigned long long *)d_words + per, c->bases.p, per * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream_em));
	}
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_counts_import_device(lsq_ctx *c, const void *d_words) LSQ_API_TRY {
	if (!c || !d_words) return fail(LSQ_E_ARG, "null argument");
	if (!c->E) return fail(LSQ_E_STATE, "lsq_events_upload must come first");
	HIP_TRY(hipSetDevice(c->device));
	const size_t per = (size_t)c->E->n_methods * c->E->n_cls_total;
	{ int rc = sync_all(c); if (rc) return rc; }
	if (per) {
		HIP_TRY(hipMemcpyAsync(c->cnt.p, d_words, per * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream_em));
		HIP_TRY(hipMemcpyAsync(c->bases.p, (const unsigned long long *)d_words + per, per * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream_em));
	}
	c->counted = true; c->solved = false;
	c->counts_external = true;
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_results_solve(lsq_ctx *c, double *theta, double *logll, uint32_t *em_iters, uint8_t *em_flags) LSQ_API_TRY {
	if (!c || !theta || !logll) return fail(LSQ_E_ARG, "null argument");
	if (!c->solved) return fail(LSQ_E_STATE, "lsq_solve must come first");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = replay_flagged(c, nullptr); if (rc) return rc; }      // guard-band events: the reference's per-read summation order
	const lsq_events &E = *c->E;
	const size_t n_ev = E.dev2out.size(), n_iso = E.n_iso_total;
	std::vector<double> ht(std::max<size_t>(n_iso, 1)), hl(std::max<size_t>(n_ev, 1));
	std::vector<uint32_t> hi(std::max<size_t>(n_ev, 1));
	std::vector<uint8_t> hf(std::max<size_t>(n_ev, 1));
	{ int rc = sync_all(c); if (rc) return rc; }
	if (n_ev) {
		HIP_TRY(hipMemcpy(ht.data(), c->theta.p, n_iso * sizeof(double), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hl.data(), c->logll.p, n_ev * sizeof(double), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hi.data(), c->iters.p, n_ev * sizeof(uint32_t), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hf.data(), c->flags.p, n_ev * sizeof(uint8_t), hipMemcpyDeviceToHost));
	}
	for (size_t o = 0; o < E.ev.size(); ++o) {        // events outside the shard: zeros
		for (int j = 0; j < E.ev[o].K; ++j) theta[E.iso_off[o] + j] = 0.0;
		logll[o] = 0.0;
		if (em_iters) em_iters[o] = 0;
		if (em_flags) em_flags[o] = 0;
	}
	for (size_t d = 0; d < n_ev; ++d) {
		const size_t o = (size_t)E.dev2out[d];
		for (int j = 0; j < E.ev[o].K; ++j) theta[E.iso_off[o] + j] = ht[E.dev_iso_base[d] + j];
		logll[o] = hl[d];
		if (em_iters) em_iters[o] = hi[d];
		if (em_flags) em_flags[o] = hf[d];
	}
	return LSQ_OK;
} LSQ_API_CATCH

} // extern "C"

namespace {
// the three result arrays to the caller's device buffers in one launch (8-byte words)
__global__ void __launch_bounds__(256) lsq_copy_results_kernel(unsigned long long *d0, const unsigned long long *s0, size_t n0, unsigned long long *d1,
                                                               const unsigned long long *s1, size_t n1, unsigned long long *d2, const unsigned long long *s2, size_t n2) {
	const size_t gsz = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1 + n2; i += gsz) {
		if (i < n0) d0[i] = s0[i];
		else if (i < n0 + n1) d1[i - n0] = s1[i - n0];
		else d2[i - n0 - n1] = s2[i - n0 - n1];
	}
}
} // namespace

extern "C" {

int lsq_results_copy_device(lsq_ctx *c, void *d_class_count, void *d_theta, void *d_logll) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	if (!c->counted) return fail(LSQ_E_STATE, "lsq_count must come first");
	if ((d_theta || d_logll) && !c->solved) return fail(LSQ_E_STATE, "lsq_solve must come first");
	HIP_TRY(hipSetDevice(c->device));
	const lsq_events &E = *c->E;
	const size_t n_cls = E.n_cls_total, M = (size_t)E.n_methods;
	const size_t n0 = d_class_count ? M * n_cls : 0, n1 = d_theta ? (size_t)E.n_iso_total : 0, n2 = d_logll ? E.dev2out.size() : 0;
	if (n0 + n1 + n2) {
		const unsigned grid = (unsigned)std::min<size_t>((n0 + n1 + n2 + 255) / 256, (size_t)c->n_cu * 8);
		hipLaunchKernelGGL(lsq_copy_results_kernel, dim3(grid), dim3(256), 0, c->stream_em, (unsigned long long *)d_class_count, c->cnt.p, n0,
		                   (unsigned long long *)d_theta, (const unsigned long long *)c->theta.p, n1, (unsigned long long *)d_logll, (const unsigned long long *)c->logll.p, n2);
		HIP_TRY(hipGetLastError());
	}
	return LSQ_OK;
} LSQ_API_CATCH

extern "C++" {
namespace {
// the shard's records in output order: [method][class] counts, [method][class] bases, theta, log-likelihood
__global__ void __launch_bounds__(256) lsq_pack_results_kernel(unsigned long long *dst, const unsigned long long *cnt, const unsigned long long *bases,
                                                               const unsigned long long *theta, const unsigned long long *logll, const unsigned *pack_cls,
                                                               const unsigned *pack_iso, const unsigned *pack_ev, size_t M, size_t C, size_t I, size_t N,
                                                               size_t n_cls_dev) {
	const size_t total = 2 * M * C + I + N, gsz = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gsz) {
		unsigned long long v;
		if (i < M * C) { const size_t m = i / C, k = i - m * C; v = cnt[m * n_cls_dev + pack_cls[k]]; }
		else if (i < 2 * M * C) { const size_t q = i - M * C, m = q / C, k = q - m * C; v = bases[m * n_cls_dev + pack_cls[k]]; }
		else if (i < 2 * M * C + I) v = theta[pack_iso[i - 2 * M * C]];
		else v = logll[pack_ev[i - 2 * M * C - I]];
		dst[i] = v;
	}
}
} // namespace
} // extern "C++"

int lsq_results_pack_device(lsq_ctx *c, void *d_block) LSQ_API_TRY {
	if (!c || !d_block) return fail(LSQ_E_ARG, "null argument");
	if (!c->solved) return fail(LSQ_E_STATE, "lsq_solve must come first");
	HIP_TRY(hipSetDevice(c->device));
	const lsq_events &E = *c->E;
	const size_t M = (size_t)E.n_methods, C = c->pack_cls.n, I = c->pack_iso.n, N = c->pack_ev.n;
	const size_t total = 2 * M * C + I + N;
	if (total) {
		const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)c->n_cu * 8);
		hipLaunchKernelGGL(lsq_pack_results_kernel, dim3(grid), dim3(256), 0, c->stream_em, (unsigned long long *)d_block, c->cnt.p, c->bases.p,
		                   (const unsigned long long *)c->theta.p, (const unsigned long long *)c->logll.p, c->pack_cls.p, c->pack_iso.p, c->pack_ev.p, M, C, I, N,
		                   (size_t)E.n_cls_total);
		HIP_TRY(hipGetLastError());
	}
	return LSQ_OK;
} LSQ_API_CATCH

void *lsq_ctx_result_stream(lsq_ctx *c) { return c ? (void *)c->stream_em : nullptr; }

// device buffers for a C host that has no HIP of its own (the executables hold the packed / gathered records in them)
int lsq_device_alloc(lsq_ctx *c, uint64_t bytes, void **out) LSQ_API_TRY {
	if (!c || !out) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipMalloc(out, (size_t)std::max<uint64_t>(bytes, 8)));
	// zeroed on the result stream, i.e. ahead of any hand-off into the buffer (a plain hipMemset runs on the null stream,
	// which the context's non-blocking streams do not wait for: it could land behind the pack and wipe it)
	HIP_TRY(hipMemsetAsync(*out, 0, (size_t)std::max<uint64_t>(bytes, 8), c->stream_em));
	HIP_TRY(hipStreamSynchronize(c->stream_em));
	return LSQ_OK;
} LSQ_API_CATCH
void lsq_device_free(lsq_ctx *c, void *p) {
	if (!c || !p) return;
	(void)hipSetDevice(c->device);
	(void)hipFree(p);
}
int lsq_device_write(lsq_ctx *c, void *device_dst, const void *host_src, uint64_t bytes) LSQ_API_TRY {
	if (!c || !device_dst || !host_src) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	HIP_TRY(hipMemcpy(device_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice));
	return LSQ_OK;
} LSQ_API_CATCH
int lsq_device_read(lsq_ctx *c, void *host_dst, const void *device_src, uint64_t bytes) LSQ_API_TRY {
	if (!c || !host_dst || !device_src) return fail(LSQ_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	HIP_TRY(hipMemcpy(host_dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost));
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_results_device_order(const lsq_ctx *c, int32_t *dev2out) LSQ_API_TRY {
	if (!c || !c->E || !dev2out) return fail(LSQ_E_ARG, "null argument");
	memcpy(dev2out, c->E->dev2out.data(), c->E->dev2out.size() * sizeof(int32_t));
	return LSQ_OK;
} LSQ_API_CATCH

extern "C++" {
namespace {
// plain streaming read: every lane 16 bytes per load, UNROLL loads in flight, grid-stride
template <int UNROLL>
__global__ void __launch_bounds__(256) lsq_read_rate_kernel(const uint4 *src, size_t n_words, unsigned *sink) {
	const size_t gsz = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned acc = 0;
	for (; i + (UNROLL - 1) * gsz < n_words; i += UNROLL * gsz) {
		uint4 v[UNROLL];
#pragma unroll
		for (int k = 0; k < UNROLL; ++k) v[k] = src[i + k * gsz];
#pragma unroll
		for (int k = 0; k < UNROLL; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
	}
	for (; i < n_words; i += gsz) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) *sink = acc;
}
} // namespace
} // extern "C++"

// developer aid (include/lesseq_hip_dev.h): what a plain read of `bytes` of HBM reaches on this device (GB/s, best launch over
// eight grid / unroll combinations) -- the practical ceiling bench.py prints beside the 8 TB/s peak
int lsq_debug_stream_read_rate(lsq_ctx *c, unsigned long long bytes, double *gb_per_s) LSQ_API_TRY {
	if (!c || !gb_per_s || bytes < (1ull << 20)) return fail(LSQ_E_ARG, "bad argument");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	lsq::DevBuf<uint4> buf;
	lsq::DevBuf<unsigned> sink;
	const size_t n_words = (size_t)(bytes / 16);
	int rc;
	if ((rc = buf.alloc(n_words)) || (rc = sink.alloc(1))) return rc;
	HIP_TRY(hipMemsetAsync(buf.p, 1, n_words * 16, c->stream));
	hipEvent_t a, b;
	HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
	float best = 1e30f;
	for (int cfg = 0; cfg < 8; ++cfg) {                 // workgroups per CU x loads in flight: the best combination counts
		const unsigned grid = (unsigned)c->n_cu * (4u << (cfg & 3));
		for (int it = 0; it < 6; ++it) {
			HIP_TRY(hipEventRecord(a, c->stream));
			if (cfg < 4) hipLaunchKernelGGL(lsq_read_rate_kernel<4>, dim3(grid), dim3(256), 0, c->stream, buf.p, n_words, sink.p);
			else hipLaunchKernelGGL(lsq_read_rate_kernel<8>, dim3(grid), dim3(256), 0, c->stream, buf.p, n_words, sink.p);
			HIP_TRY(hipEventRecord(b, c->stream));
			HIP_TRY(hipEventSynchronize(b));
			float ms = 0;
			HIP_TRY(hipEventElapsedTime(&ms, a, b));
			if (it >= 1) best = std::min(best, ms);
		}
	}
	(void)hipEventDestroy(a); (void)hipEventDestroy(b);
	*gb_per_s = (double)(n_words * 16) / ((double)best * 1e-3) / 1e9;
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_debug_hip_versions(int *compiled, int *runtime) {
	if (compiled) *compiled = HIP_VERSION;
	if (runtime) { int v = 0; if (hipRuntimeGetVersion(&v) != hipSuccess) v = 0; *runtime = v; }
	return LSQ_OK;
}

// developer aid (include/lesseq_hip_dev.h): another placement of the events in the EM grid (experiments on wave make-up)
int lsq_debug_set_em_order(lsq_ctx *c, const uint32_t *order, unsigned n_small_places, unsigned n_places) LSQ_API_TRY {
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	if (n_small_places > n_places) return fail(LSQ_E_ARG, "more small places than places");
	int rc = c->em_order.upload(order, n_places, c->stream);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	{
		// the closed form's lists are sized by the lean group (lsq_events_upload): a placement that makes the group larger takes larger lists
		const size_t np = std::max<size_t>(n_small_places, 1);
		for (int l = 0; l < 2; ++l)
			if (c->em_tail_flag[l].n < np && ((rc = c->em_tail_u32[l].alloc(2 * np)) || (rc = c->em_tail_flag[l].alloc(np)) || (rc = c->em_tail_f64[l].alloc(3 * np)))) return rc;
	}
	c->em_small_places = n_small_places; c->em_places = n_places;
	c->em_order_lane_valid[0] = c->em_order_lane_valid[1] = false;
	c->opt_em_regroup = false;            // a placement given by hand stays
	return LSQ_OK;
} LSQ_API_CATCH

// Tuning knobs of a context (none is needed for correct results).
int lsq_ctx_set_option(lsq_ctx *c, const char *name, double value) LSQ_API_TRY {
	if (!c || !name) return fail(LSQ_E_ARG, "null argument");
	const std::string n(name);
	if (n == "grid_multiplier") {
		if (!(value >= 0 && value <= 64)) return fail(LSQ_E_ARG, "grid_multiplier must lie in 0..64 (0 = automatic)");
		c->opt_grid_mult = value;
		for (auto &r : c->reads) r.wg_grid = 0;
	} else if (n == "exception_capacity") {
		if (!(value >= 0 && value <= 4e9)) return fail(LSQ_E_ARG, "exception_capacity must lie in 0..4e9 (0 = automatic)");
		c->opt_exc_cap = (size_t)value;          // takes effect with the next upload of a read set
	} else if (n == "snap_shares") {
		c->opt_snap_shares = value != 0;
		for (auto &r : c->reads) r.wg_grid = 0;
	} else if (n == "share_weighted" || n == "share_cost_two_block" || n == "share_cost_parked" || n == "share_cost_visit" || n == "share_taper" || n == "share_cost_hot") {
		if (!(value >= 0 && value <= 1e6)) return fail(LSQ_E_ARG, "%s must lie in 0..1e6", name);
		if (n == "share_weighted") c->opt_share_weighted = value != 0;
		else if (n == "share_cost_two_block") c->opt_share_cost_p2 = value;
		else if (n == "share_cost_parked") c->opt_share_cost_park = value;
		else if (n == "share_cost_visit") c->opt_share_cost_visit = value;
		else if (n == "share_cost_hot") c->opt_share_cost_hot = value;
		else c->opt_share_taper = value;
		for (auto &r : c->reads) r.wg_grid = 0;
	} else if (n == "compact_pools") {
		c->opt_compact_pools = value != 0;        // takes effect with the next upload of a read set
	} else if (n == "workgroups_per_cu") {
		if (!(value >= -1 && value <= 32)) return fail(LSQ_E_ARG, "workgroups_per_cu must lie in -1..32 (-1 = automatic, 0 = as many as fit)");
		c->opt_wg_per_cu = (int)value;
		c->occ_lds_bytes = 0;          // the occupancy is asked again
	} else if (n == "reads_per_look") {
		if (!(value == 0 || value == 4 || value == 8)) return fail(LSQ_E_ARG, "reads_per_look must be 0 (automatic), 4 or 8");
		c->opt_reads_per_look = (int)value;
	} else if (n == "count_streams") {
		{ int rc = sync_all(c); if (rc) return rc; }
		c->opt_two_count_streams = value >= 2;
	} else if (n == "em_flat_min_events") {
		if (!(value >= 0 && value <= 4e9)) return fail(LSQ_E_ARG, "em_flat_min_events must lie in 0..4e9");
		c->opt_em_flat_min = (unsigned)value;
	} else if (n == "em_closed_form") {
		{ int rc = sync_all(c); if (rc) return rc; }
		c->opt_em_closed = value != 0;
	} else if (n == "em_quad_cap") {
		if (value < 0 || value > 1e6) return fail(LSQ_E_ARG, "em_quad_cap out of range");
		c->opt_em_quad_cap = (unsigned)value;
	} else if (n == "em_regroup") {
		c->opt_em_regroup = value != 0;
		c->em_order_lane_valid[0] = c->em_order_lane_valid[1] = false;
	} else if (n == "recount_every_read") {
		c->opt_recount = value != 0;
	} else if (n == "cleanup_workgroups") {
		if (value < 0 || value > 4096) return fail(LSQ_E_ARG, "cleanup_workgroups out of range");
		c->opt_cleanup_grid = (unsigned)value;
	} else if (n == "em_guard_band") {
		return lsq_set_em_guard_band(c, value);
	} else
		return fail(LSQ_E_ARG, "unknown option '%s'", name);
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_set_timing(lsq_ctx *c, int on) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	c->time_events = on != 0;
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_last_fast_kernel_ms(lsq_ctx *c, float *ms) LSQ_API_TRY {
	if (!c || !ms) return fail(LSQ_E_ARG, "null argument");
	if (!c->counted || !c->count_timed) return fail(LSQ_E_STATE, "the last lsq_count ran without timing (lsq_set_timing)");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	*ms = 0;
	for (int m = 0; m < LSQ_MAX_METHODS; ++m) if (c->fast_launched >> m & 1) { float t = 0; HIP_TRY(hipEventElapsedTime(&t, c->evf0[m], c->evf1[m])); *ms += t; }
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_last_timing(lsq_ctx *c, float *count_ms, float *solve_ms) LSQ_API_TRY {
	if (!c) return fail(LSQ_E_ARG, "null context");
	HIP_TRY(hipSetDevice(c->device));
	{ int rc = sync_all(c); if (rc) return rc; }
	if ((count_ms && c->counted && !c->count_timed) || (solve_ms && c->solved && !c->solve_timed))
		return fail(LSQ_E_STATE, "the last lsq_count / lsq_solve ran without timing (lsq_set_timing)");
	if (count_ms) { *count_ms = 0; if (c->counted) HIP_TRY(hipEventElapsedTime(count_ms, c->ev0, c->ev1)); }
	if (solve_ms) { *solve_ms = 0; if (c->solved) HIP_TRY(hipEventElapsedTime(solve_ms, c->ev2, c->ev3)); }
	return LSQ_OK;
} LSQ_API_CATCH

} // extern "C"
