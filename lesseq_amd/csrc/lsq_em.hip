// The batched EM of `solve` (common/read.h:592-660 on compatibility-class counts): kernel and launch.
#include "lsq_device.hpp"

namespace {

// ---- EM: one event per lane (common/read.h:592-660 on compatibility classes) ----------------
struct EmArgs {
	unsigned n_events, n_methods, n_cls, n_iso;
	unsigned n_places;                 // entries of `order` this launch covers: [place0, n_places)
	unsigned place0;
	double band;                       // guard band around the 1e-6 stop threshold (lsq_set_em_guard_band)
	unsigned max_iters;                // read.h has no cap; 1000000 flags the event (developer switch LSQ_EM_CAP lowers it for timing experiments)
	const unsigned *order;             // device event per place of the EM grid
	const unsigned *split;             // lean group: places below *split run four lanes an event, the others one (lsq_em_lean_kernel)
	const unsigned char *K;
	const unsigned *cls_base, *iso_base;
	const unsigned long long *cnt;     // [method][n_cls]
	const double *G;                   // [method][n_iso]
	double *theta, *logll;
	unsigned *iters;
	unsigned char *flags;
};

// Four lanes per event: lane `sub` takes the (method, class) pairs sub, sub+4, ...; the four
// partial sums meet by two xor-shuffles.  One pass per EM iteration gives, for the current theta,
// the class mixtures s, the log-likelihood and the numerators of the next theta.  All four
// lanes of an event hold the same theta and take the same decisions.

// sum over the four lanes of an event (an aligned quad): two DPP quad permutes per 32-bit half,
// plain VALU moves with no trip through the LDS crossbar
template <int CTRL>
__device__ inline double quad_perm_f64(double x) {
	// (every lane of the quad is read by one of its lanes and all of them are enabled: no value is needed for a lane
	// whose source is missing, so none is set up -- with a 0 there the compiler spent a move per half and permute on it)
	int lo = __double2loint(x), hi = __double2hiint(x);
	lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
	hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}
__device__ inline double group_sum(double x) {
	x += quad_perm_f64<0xB1>(x);      // quad_perm:[1,0,3,2]
	x += quad_perm_f64<0x4E>(x);      // quad_perm:[2,3,0,1]
	return x;
}

__device__ inline void em_pass(const EmArgs &A, unsigned cb, unsigned ib, int K, int nc, unsigned sub, bool on,
                               const double *th, double &ll, double *z) {
	double l = 0;
	double zz[LSQ_MAX_ISOFORMS];
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) zz[j] = 0;
	if (on) {
		const int n_pairs = (int)A.n_methods * nc;
		for (int q = (int)sub; q < n_pairs; q += EM_LANES) {
			const int m = q / nc, c = q - m * nc + 1;
			const unsigned long long k = A.cnt[(size_t)m * A.n_cls + cb + (unsigned)(c - 1)];
			if (!k) continue;
			const double *g = A.G + (size_t)m * A.n_iso + ib;
			double s = 0;
#pragma unroll
			for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) if (j < K && (c >> j & 1)) s += th[j] * g[j];
			const double kd = (double)k;
			l += kd * log(s);
			if (s > 0) {
#pragma unroll
				for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) if (j < K && (c >> j & 1)) {
					const double local = th[j] * g[j];
					if (local > 0) zz[j] += kd * (local / s);
				}
			}
		}
	}
	ll = group_sum(l);
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) z[j] = group_sum(zz[j]);
}

// An event's (method, class) counts and G values in registers when there are at most two pairs
// per lane and three isoforms (every LESSeq local event with up to two read files): the passes
// then touch no memory, and the latency of one pass is what bounds the kernel (the slowest
// event of the batch runs ~160 dependent passes).
constexpr int EM_CACHED_PAIRS = 2, EM_CACHED_K = 3;
struct EmCache {
	double kd[EM_CACHED_PAIRS];
	double g[EM_CACHED_PAIRS][EM_CACHED_K];
	int cls[EM_CACHED_PAIRS];
};

// 1/s to ~1 ulp: hardware reciprocal estimate and two Newton steps -- about half the dependent
// chain of an IEEE division (the result stays far inside the 1e-6 tolerance of the path)
__device__ inline double fast_recip(double s) {
	double r = __builtin_amdgcn_rcp(s);
	r = fma(fma(-s, r, 1.0), r, r);
	r = fma(fma(-s, r, 1.0), r, r);
	return r;
}

// log(s) for normal positive s to about 1 ulp (everything else goes to the library): exponent and
// mantissa m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(z) with z = (m-1)/(m+1), |z| < 0.172, as an
// odd series in z evaluated by Estrin's scheme -- a dependent chain of about 20 operations, a
// third of the library routine's.  The stop rule compares log-likelihoods to 1e-6; events whose
// criterion comes within 1e-11 of it are flagged whatever the logarithm used.
__device__ inline double fast_log(double s) {
	const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
	const unsigned ex = (unsigned)(bits >> 52);
	if (ex - 1u >= 0x7FEu) return log(s);                       // zero, subnormal, negative, inf, nan
	double m = __longlong_as_double((long long)((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
	int e = (int)ex - 1023;
	if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
	const double f = m - 1.0;
	const double z = f * fast_recip(2.0 + f);
	const double w = z * z, w2 = w * w, w4 = w2 * w2, w8 = w4 * w4;
	// 1/3 + w/5 + w^2/7 + ... + w^9/21
	const double p01 = fma(w, 1.0 / 5.0, 1.0 / 3.0), p23 = fma(w, 1.0 / 9.0, 1.0 / 7.0), p45 = fma(w, 1.0 / 13.0, 1.0 / 11.0),
	             p67 = fma(w, 1.0 / 17.0, 1.0 / 15.0), p89 = fma(w, 1.0 / 21.0, 1.0 / 19.0);
	const double q0 = fma(w2, p23, p01), q1 = fma(w2, p67, p45);
	const double poly = fma(w8, p89, fma(w4, q1, q0));
	const double lm = fma(z * w, 2.0 * poly, 2.0 * z);
	const double ed = (double)e;
	return fma(ed, 0.69314718055994528623, fma(ed, 2.3190468138462995584e-17, lm));
}

__device__ inline void em_pass_cached(const EmCache &E, const double *th, bool on, double &ll, double *z) {
	double l = 0, zz[EM_CACHED_K] = {0, 0, 0};
#pragma unroll
	for (int t = 0; t < EM_CACHED_PAIRS; ++t) {
		const double kd = E.kd[t];
		if (on && kd != 0) {
			const int c = E.cls[t];
			double s = 0;
#pragma unroll
			for (int j = 0; j < EM_CACHED_K; ++j) if (c >> j & 1) s += th[j] * E.g[t][j];
			l += kd * fast_log(s);
			if (s > 0) {
				const double kr = kd * fast_recip(s);
#pragma unroll
				for (int j = 0; j < EM_CACHED_K; ++j) if (c >> j & 1) {
					const double local = th[j] * E.g[t][j];
					if (local > 0) zz[j] += local * kr;
				}
			}
		}
	}
	ll = group_sum(l);
#pragma unroll
	for (int j = 0; j < EM_CACHED_K; ++j) z[j] = group_sum(zz[j]);
}

// The register-cached pass with the class masks folded into G (an isoform outside the class has
// G = 0: its term adds an exact zero, so sums and their order are those of em_pass_cached).
// What a pass leaves behind per pair for the next one: the mixture s, its reciprocal and its
// logarithm.  Passes follow one another with small steps in s (that is what makes slow events
// slow), so log s(t+1) = log s(t) + log1p(d) with d = (s(t+1) - s(t)) / s(t), and for |d| < 2^-5 a
// twelve-term series gives log1p to 1e-19: a chain of six operations instead of the logarithm's twenty.
// A wave skips the full routine when every live pair of every lane is inside that range; when it
// does not, only the pairs outside it take the routine's value, so an event's numbers do not depend
// on the events it shares a wave with.
// the list a capped or headed EM leaves for lsq_em_tail_kernel (further down, with the closed form)
struct EmTail {
	unsigned *count;               // [0] events appended by the head; [1] tail workgroups done (the last one clears both)
	unsigned *ev, *iters;
	unsigned char *flag;
	double *t0, *t1, *ll;          // theta and log-likelihood after `iters` accepted iterations
};

template <int SLOTS>
struct EmPairState { double s[SLOTS], r[SLOTS], lg[SLOTS]; };

__device__ inline double log1p_small(double d) {
	// d (1 - d/2 + d^2/3 - ... - d^11/12), |d| < 2^-5: the first term left out is below 1e-19
	const double w = d * d, w2 = w * w, w4 = w2 * w2;
	const double a0 = fma(d, -1.0 / 2.0, 1.0), a1 = fma(d, -1.0 / 4.0, 1.0 / 3.0), a2 = fma(d, -1.0 / 6.0, 1.0 / 5.0),
	             a3 = fma(d, -1.0 / 8.0, 1.0 / 7.0), a4 = fma(d, -1.0 / 10.0, 1.0 / 9.0), a5 = fma(d, -1.0 / 12.0, 1.0 / 11.0);
	const double b0 = fma(w, a1, a0), b1 = fma(w, a3, a2), b2 = fma(w, a5, a4);
	return d * fma(w4, b2, fma(w2, b1, b0));
}

// FLAT: one lane holds a whole event -- its four (method, class) pairs in the four slots, where the other form has them
// in the four lanes of a quad -- and sums the slots in the order the quad's two permutes do, (0 + 1) + (2 + 3): the same
// numbers bit for bit, four times the events per wave and no permutes.
template <int SLOTS, int KK, bool FLAT = false>
__device__ inline void em_pass_lean(const double (&kd)[SLOTS], const double (&gm)[SLOTS][KK], const double (&th)[KK],
                                    const bool on, EmPairState<SLOTS> &P, double &ll, double (&z)[KK]) {
	static_assert(!FLAT || SLOTS == 4 || SLOTS == 3, "a flat event has the quad's four slots (three when no event uses the fourth)");
	double l = 0, zz[KK], part[FLAT ? SLOTS : 1][KK], lpart[FLAT ? SLOTS : 1];
#pragma unroll
	for (int j = 0; j < KK; ++j) zz[j] = 0;
	double local[SLOTS][KK], sm[SLOTS], safe[SLOTS], r[SLOTS], d[SLOTS];
	bool on_t[SLOTS], far_t[SLOTS], far = false;
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		// (sums of non-negative terms start from their first term: 0 + x is x for them, and without -ffast-math the add of
		// the zero stays in the chain)
#pragma unroll
		for (int j = 0; j < KK; ++j) { local[t][j] = th[j] * gm[t][j]; sm[t] = j == 0 ? local[t][0] : sm[t] + local[t][j]; }
		on_t[t] = on && kd[t] != 0;
		safe[t] = (on_t[t] && sm[t] > 0) ? sm[t] : 1.0;     // an empty pair slot must not send the wave down the library path
		r[t] = fast_recip(safe[t]);
		d[t] = (safe[t] - P.s[t]) * P.r[t];
		// outside the series' range, or log of zero (the full routine gives the reference's -inf), or a mixture next
		// to 1 (an isoform with a handful of accessible starts): the carried logarithm keeps the absolute error of its
		// history, which is no relative accuracy at all once log s itself comes down to zero (a lone read on such an
		// isoform has log-likelihood exactly 0 in the reference) -- there the full routine is exact
		far_t[t] = on_t[t] && (!(fabs(d[t]) < 0.03125) || !(sm[t] > 0) || fabs(sm[t] - 1.0) < 0.015625);
		far = far || far_t[t];
	}
	// the numerators first: the next pass waits for them, and nothing in them waits for the logarithm
	// or for the wave-wide vote below (an in-order wave stalls at that branch until the vote is in)
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		const double kr = (on_t[t] && sm[t] > 0) ? kd[t] * r[t] : 0.0;
#pragma unroll
		for (int j = 0; j < KK; ++j) {
			if (FLAT) part[t][j] = local[t][j] * kr;
			else zz[j] = t == 0 ? local[t][j] * kr : zz[j] + local[t][j] * kr;
		}
	}
#pragma unroll
	for (int j = 0; j < KK; ++j) {
		// (three slots: the quad's idle fourth lane adds an exact zero to the third's non-negative product)
		if (FLAT) z[j] = SLOTS == 4 ? (part[0][j] + part[1][j]) + (part[2][j] + part[SLOTS - 1][j]) : (part[0][j] + part[1][j]) + part[2][j];
		else z[j] = group_sum(zz[j]);
	}
	const bool full = __any(far);
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		double lg = P.lg[t] + log1p_small(d[t]);
		if (full) {                                  // wave-uniform, taken a handful of times per event
			asm volatile("" ::: "memory");           // keeps the compiler from flattening the branch into both computations
			// only the pairs that are out of range themselves take the value: what an event computes does not
			// depend on the events it shares a wave with
			const double fl = fast_log(on_t[t] ? sm[t] : 1.0);
			lg = far_t[t] ? fl : lg;
		}
		P.s[t] = safe[t]; P.r[t] = r[t]; P.lg[t] = lg;
		const double term = kd[t] * lg;
		if (FLAT) lpart[t] = 0.0 + (on_t[t] ? term : 0.0);          // (a lane of the quad form starts its sum at 0)
		else l += on_t[t] ? term : 0.0;
	}
	if (FLAT) ll = SLOTS == 4 ? (lpart[0] + lpart[1]) + (lpart[2] + lpart[SLOTS - 1]) : (lpart[0] + lpart[1]) + (lpart[2] + 0.0);
	else ll = group_sum(l);
}

// The whole EM of a wave whose events all fit SLOTS (method, class) pairs per lane and KK isoforms,
// in registers.  The pass for theta(t+2) starts from z(t+1) as soon as that exists, without waiting
// for the stop test on ll(t+1): the test (a reciprocal, a compare, a ballot) runs beside the next
// pass instead of between two passes.  One pass per event is thrown away.
// cap (0: none): an event still running after that many accepted iterations is not iterated to the end here but appended to T's
// list with theta, log-likelihood and count as they stand -- lsq_em_tail_kernel finishes it (closed form where it applies).
template <int SLOTS, int KK, bool FLAT = false, class CacheT = EmCache>
__device__ inline void em_lean(const EmArgs &A, const CacheT &C, const unsigned e, const unsigned sub, const bool ev_ok, const int K, const unsigned ib,
                               const double inv_n, const bool any_reads, bool run, const unsigned cap = 0u, const EmTail *T = nullptr) {
	double kd[SLOTS], gm[SLOTS][KK], t3[KK], z3[KK];
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) {
		kd[t] = C.kd[t];
#pragma unroll
		for (int j = 0; j < KK; ++j) gm[t][j] = (C.cls[t] >> j & 1) ? C.g[t][j] : 0.0;
	}
#pragma unroll
	for (int j = 0; j < KK; ++j) t3[j] = (K == 1) ? 1.0 : 1.0 / (double)K;   // solve/solve.cpp:798-802, read.h:642
	EmPairState<SLOTS> P;
#pragma unroll
	for (int t = 0; t < SLOTS; ++t) { P.s[t] = 1.0; P.r[t] = 1.0; P.lg[t] = 0.0; }
	unsigned iters = 0;
	unsigned char flag = 0;
	bool handed_on = false;
	double ll = 0;
	em_pass_lean<SLOTS, KK, FLAT>(kd, gm, t3, any_reads, P, ll, z3);
	double c3[KK], cll, cz3[KK];           // candidate: theta(t+1), its log-likelihood and numerators
#pragma unroll
	for (int j = 0; j < KK; ++j) c3[j] = z3[j] * inv_n;
	em_pass_lean<SLOTS, KK, FLAT>(kd, gm, c3, run, P, cll, cz3);
	// One turn of the loop: the candidate c = theta(t+1) with its log-likelihood and numerators exists; the pass for
	// n = theta(t+2) starts from c's numerators at once, and beside it c is tested against the accepted theta(t) and, for
	// the events still running, accepted.  The loop body is written out twice with c and n in swapped roles, so that
	// nothing has to be copied from "next" to "candidate" between two turns.
	auto turn = [&](const double (&c3)[KK], const double cll, const double (&cz3)[KK], double (&n3)[KK], double &nll, double (&nz3)[KK]) {
#pragma unroll
		for (int j = 0; j < KK; ++j) n3[j] = cz3[j] * inv_n;
		em_pass_lean<SLOTS, KK, FLAT>(kd, gm, n3, run, P, nll, nz3);         // speculative: theta(t+2)
		const unsigned cll_ex = (unsigned)((unsigned long long)__double_as_longlong(cll) >> 52) & 0x7FFu;
		// read.h:659, floating abs; -inf, nan, zero keep the division's own answers
		const bool plain = cll_ex - 1u < 0x7FEu;
		double crit = fabs(1.0 - ll * fast_recip(plain ? cll : 1.0));
		if (__any(!plain)) {                         // wave-uniform and rare: the division proper (a select between the two
			asm volatile("" ::: "memory");           // forms made every turn compute both)
			const double by_division = fabs(1.0 - ll / cll);
			crit = plain ? crit : by_division;
		}
		const bool go = run;
#pragma unroll
		for (int j = 0; j < KK; ++j) t3[j] = go ? c3[j] : t3[j];
		ll = go ? cll : ll;
		iters += go ? 1u : 0u;
		if (go && fabs(crit - 1E-6) < A.band) flag |= 1;
		if (go && !(crit > 1E-6)) run = false;
		else if (go && iters >= A.max_iters) { flag |= 2; run = false; }
		else if (go && cap && iters >= cap) { handed_on = true; run = false; }
	};
	{
		double n3[KK], nll, nz3[KK];
		while (__any(run)) {
			turn(c3, cll, cz3, n3, nll, nz3);
			if (!__any(run)) break;
			turn(n3, nll, nz3, c3, cll, cz3);
		}
	}
	if (handed_on) {
		if (ev_ok && sub == 0) {
			const unsigned at = atomicAdd(&T->count[0], 1u);
			T->ev[at] = e; T->iters[at] = iters; T->flag[at] = flag; T->t0[at] = t3[0]; T->t1[at] = KK > 1 ? t3[KK > 1 ? 1 : 0] : 0.0; T->ll[at] = ll;
		}
		return;
	}
	if (ev_ok && sub == 0) {
#pragma unroll
		for (int j = 0; j < KK; ++j) if (j < K) A.theta[ib + j] = t3[j];
		A.logll[e] = ll;
		A.iters[e] = iters;
		A.flags[e] = flag;
	}
}

// SMALL: every event of the launch has at most two isoforms and one (method, class) pair per lane
// (LESSeq's local events with one read file): only the lean loop, a third of the registers -- the
// kernel shares the compute units with the next count's streaming kernel (lsq_device.hpp).
template <bool SMALL>
__device__ inline void em_quad_body(const EmArgs &A, const unsigned block, const unsigned cap = 0u, const EmTail *T = nullptr) {
	// The next count's streaming kernel may share the SIMDs (lsq_device.hpp: the result stream); this
	// kernel is a few dependent chains, that one thousands of independent ones: these waves go first.
#ifndef LSQ_EM_NO_PRIO
	__builtin_amdgcn_s_setprio(3);
#endif
	const unsigned gid = block * blockDim.x + threadIdx.x;
	const unsigned place = A.place0 + gid / EM_LANES, sub = gid % EM_LANES;
	// events in the order of A.order: the small ones (two isoforms, one pair per lane) first, then the
	// rest, each group filling whole waves (0xFFFFFFFF = empty place)
	const unsigned mine = SMALL ? min(A.n_places, *A.split) : A.n_places;          // (the lean group is shared with lsq_em_flat_kernel)
	if (SMALL && A.place0 + block * blockDim.x / EM_LANES >= mine) return;
	const unsigned e = place < mine ? A.order[place] : 0xFFFFFFFFu;
	const bool ev_ok = e != 0xFFFFFFFFu;
	const int K = ev_ok ? A.K[e] : 1;
	const unsigned cb = ev_ok ? A.cls_base[e] : 0, ib = ev_ok ? A.iso_base[e] : 0;
	const int nc = (1 << K) - 1;
	const int n_pairs = (int)A.n_methods * nc;
	const bool cached = K <= EM_CACHED_K && n_pairs <= EM_LANES * EM_CACHED_PAIRS;
	double th[LSQ_MAX_ISOFORMS], z[LSQ_MAX_ISOFORMS];
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) z[j] = 0;
	EmCache C;
	double tot = 0;
#pragma unroll
	for (int t = 0; t < EM_CACHED_PAIRS; ++t) {
		C.kd[t] = 0; C.cls[t] = 0;
#pragma unroll
		for (int j = 0; j < EM_CACHED_K; ++j) C.g[t][j] = 0;
	}
	if (ev_ok) {
		for (int q = (int)sub; q < n_pairs; q += EM_LANES) {
			const int m = q / nc, c = q - m * nc;
			const double kd = (double)A.cnt[(size_t)m * A.n_cls + cb + (unsigned)c];      // exact: counts are far below 2^53
			tot += kd;
			const int t = (q - (int)sub) / EM_LANES;
			if (cached && t < EM_CACHED_PAIRS) {
#pragma unroll
				for (int tt = 0; tt < EM_CACHED_PAIRS; ++tt) if (tt == t) {
					C.kd[tt] = kd; C.cls[tt] = c + 1;
#pragma unroll
					for (int j = 0; j < EM_CACHED_K; ++j) C.g[tt][j] = j < K ? A.G[(size_t)m * A.n_iso + ib + j] : 0.0;
				}
			}
		}
	}
	const double n_total = group_sum(tot);
	const double inv_n = 1.0 / n_total;
#pragma unroll
	for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) th[j] = (K == 1) ? 1.0 : 1.0 / (double)K;   // solve/solve.cpp:798-802, read.h:642
	unsigned iters = 0;
	unsigned char flag = 0;
	double ll = 0;
	// no reads: theta stays 1/K, log-likelihood 0; one isoform: theta = 1 (solve/solve.cpp:798-802)
	bool run = ev_ok && n_total > 0 && K > 1;
	const bool any_reads = ev_ok && n_total > 0;
	// every event of the wave fits the registers: a loop with nothing but the lean pass in it
	if (SMALL) { em_lean<1, 2>(A, C, e, sub, ev_ok, K, ib, inv_n, any_reads, run, cap, T); return; }
	if (__all(!ev_ok || (cached && K <= 2 && n_pairs <= EM_LANES))) { em_lean<1, 2>(A, C, e, sub, ev_ok, K, ib, inv_n, any_reads, run); return; }
	if (__all(!ev_ok || cached)) { em_lean<EM_CACHED_PAIRS, EM_CACHED_K>(A, C, e, sub, ev_ok, K, ib, inv_n, any_reads, run); return; }
	if (cached) em_pass_cached(C, th, any_reads, ll, z);
	else em_pass(A, cb, ib, K, nc, sub, any_reads, th, ll, z);
	while (__any(run)) {
		// theta' = z(theta) / n; then one pass at theta' gives ll(theta') and z(theta')
		double nth[LSQ_MAX_ISOFORMS], nll, nz[LSQ_MAX_ISOFORMS];
#pragma unroll
		for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) { nth[j] = cached ? z[j] * inv_n : z[j] / n_total; nz[j] = 0; }
		if (cached) em_pass_cached(C, nth, run, nll, nz);
		else em_pass(A, cb, ib, K, nc, sub, run, nth, nll, nz);
		if (run) {
			// read.h:659, floating abs; the quotient through the reciprocal when the passes are the
			// register-cached ones (1 ulp, against a guard band of 1e-11 around the threshold)
			const unsigned nll_ex = (unsigned)((unsigned long long)__double_as_longlong(nll) >> 52) & 0x7FFu;
			const bool nll_normal = nll_ex - 1u < 0x7FEu;     // -inf, nan, zero keep the division's own answers
			const double crit = (cached && nll_normal) ? fabs(1.0 - ll * fast_recip(nll)) : fabs(1.0 - ll / nll);
#pragma unroll
			for (int j = 0; j < LSQ_MAX_ISOFORMS; ++j) { th[j] = nth[j]; z[j] = nz[j]; }
			ll = nll;
			++iters;
			if (fabs(crit - 1E-6) < A.band) flag |= 1;
			if (!(crit > 1E-6)) run = false;
			else if (iters >= A.max_iters) { flag |= 2; run = false; }
		}
	}
	if (ev_ok && sub == 0) {
		for (int j = 0; j < K; ++j) A.theta[ib + j] = th[j];
		A.logll[e] = ll;
		A.iters[e] = iters;
		A.flags[e] = flag;
	}
}

// The lean group (two isoforms, at most four (method, class) pairs: LESSeq's local events with one read file) with one
// LANE per event: 64 events a wave instead of 16, a third of the instructions per event and pass -- what the next count's
// streaming kernel, which shares the vector pipes with this one, gets back -- and the same numbers (em_pass_lean FLAT).
template <int FS> struct EmCacheFlat { double kd[FS]; double g[FS][2]; int cls[FS]; };
// FS: slots of an event -- 3 when every event of the group has at most three pairs (one read file), else 4
template <int FS>
__device__ inline void em_flat_body(const EmArgs &A, const unsigned block) {
#ifndef LSQ_EM_NO_PRIO
	__builtin_amdgcn_s_setprio(3);
#endif
	const unsigned place = A.place0 + block * blockDim.x + threadIdx.x;
	const unsigned first = *A.split;                  // a multiple of 64: the places below it are the four-lane form's
	if (A.place0 + (block + 1u) * blockDim.x <= first) return;
	const unsigned e = (place < A.n_places && place >= first) ? A.order[place] : 0xFFFFFFFFu;
	const bool ev_ok = e != 0xFFFFFFFFu;
	const int K = ev_ok ? A.K[e] : 1;
	const unsigned cb = ev_ok ? A.cls_base[e] : 0, ib = ev_ok ? A.iso_base[e] : 0;
	const int nc = (1 << K) - 1;
	const int n_pairs = (int)A.n_methods * nc;        // <= FS for every event of this group
	EmCacheFlat<FS> C;
#pragma unroll
	for (int q = 0; q < FS; ++q) {
		const bool has = ev_ok && q < n_pairs;
		const int m = has ? q / nc : 0, c = has ? q - m * nc : 0;
		C.kd[q] = has ? (double)A.cnt[(size_t)m * A.n_cls + cb + (unsigned)c] : 0.0;      // exact: counts are far below 2^53
		C.cls[q] = has ? c + 1 : 0;
#pragma unroll
		for (int j = 0; j < 2; ++j) C.g[q][j] = (has && j < K) ? A.G[(size_t)m * A.n_iso + ib + j] : 0.0;
	}
	const double n_total = ((0.0 + C.kd[0]) + (0.0 + C.kd[1])) + ((0.0 + C.kd[2]) + (0.0 + (FS == 4 ? C.kd[FS - 1] : 0.0)));
	const double inv_n = 1.0 / n_total;
	// no reads: theta stays 1/K, log-likelihood 0; one isoform: theta = 1 (solve/solve.cpp:798-802)
	const bool run = ev_ok && n_total > 0 && K > 1, any_reads = ev_ok && n_total > 0;
	em_lean<FS, 2, true, EmCacheFlat<FS>>(A, C, e, 0u, ev_ok, K, ib, inv_n, any_reads, run);
}

// ---- the lean group with one read file, without a placement: a head of sequential passes, then a closed form ---------------
// A two-isoform event with one read file has three compatibility classes -- n1 reads on isoform 0 alone, n2 on isoform 1
// alone, n3 on both -- and the EM step (read.h:592-618) is then a Moebius map of x = theta_0:
//     x' = a + b x G0 / (x G0 + (1 - x) G1),   a = n1 / n, b = n3 / n,
// i.e. x' = (alpha x + beta) / (gamma x + delta) with alpha = a (G0 - G1) + b G0, beta = a G1, gamma = G0 - G1, delta = G1.
// With its fixed points p (attracting, in (0, 1)) and q, w = (x - p) / (x - q) obeys w' = kappa w, kappa = (gamma q +
// delta) / (gamma p + delta): theta after m more iterations is ONE exponential away, x_m = (p - q w0 kappa^m) / (1 - w0 kappa^m).
// The reference stops at the first iteration t with |1 - l(t-1) / l(t)| <= 1e-6 (read.h:659); along the monotone approach of
// x to p that test, once true, stays true (checked on 9e5 random events over five decades of counts and ARS: no exception
// when n1, n2 >= 1), so the stopping iteration is found by doubling steps and bisection -- ~8 evaluations of two
// log-likelihoods instead of up to hundreds of dependent passes -- and the last EM step and the final log-likelihood are
// then done by the ordinary pass, from theta(T - 1).  The numbers differ from the sequential iteration's by ~1e-13
// (measured), the iteration count not at all -- except where the test value lies within the guard band of the threshold
// at T or T - 1: flagged, and replayed in per-read order like any such event (lsq_replay.hip).
// Every event first runs EM_HEAD_PASSES ordinary iterations (lsq_em_head_kernel: most events are done by then: median 5);
// what still runs is appended to a list, and lsq_em_tail_kernel finishes the list: the closed form where it applies
// (n1, n2 >= 1, G0, G1 > 0, 0 <= kappa < 1), the ordinary iteration otherwise.  No event waits for the slowest of its wave
// through hundreds of passes, and nothing is learnt from an earlier solve.
constexpr unsigned EM_HEAD_PASSES = 6;

template <int FS>
__device__ inline void em_load_flat(const EmArgs &A, const unsigned e, const bool ev_ok, int &K, unsigned &ib, EmCacheFlat<FS> &C, double &n_total) {
	K = ev_ok ? A.K[e] : 1;
	const unsigned cb = ev_ok ? A.cls_base[e] : 0;
	ib = ev_ok ? A.iso_base[e] : 0;
	const int nc = (1 << K) - 1;
	const int n_pairs = (int)A.n_methods * nc;        // <= FS for every event of this group
#pragma unroll
	for (int q = 0; q < FS; ++q) {
		const bool has = ev_ok && q < n_pairs;
		const int m = has ? q / nc : 0, c = has ? q - m * nc : 0;
		C.kd[q] = has ? (double)A.cnt[(size_t)m * A.n_cls + cb + (unsigned)c] : 0.0;      // exact: counts are far below 2^53
		C.cls[q] = has ? c + 1 : 0;
#pragma unroll
		for (int j = 0; j < 2; ++j) C.g[q][j] = (has && j < K) ? A.G[(size_t)m * A.n_iso + ib + j] : 0.0;
	}
	n_total = ((0.0 + C.kd[0]) + (0.0 + C.kd[1])) + ((0.0 + C.kd[2]) + (0.0 + (FS == 4 ? C.kd[FS - 1] : 0.0)));
}

// the stop test of read.h:659 as the kernels form it: floating abs; -inf, nan, zero keep the division's own answers
__device__ inline double em_crit(const double ll, const double nll) {
	const unsigned ex = (unsigned)((unsigned long long)__double_as_longlong(nll) >> 52) & 0x7FFu;
	return (ex - 1u < 0x7FEu) ? fabs(1.0 - ll * fast_recip(nll)) : fabs(1.0 - ll / nll);
}

// ordinary iterations from (t3, ll, z3 = numerators at t3) until the stop test holds, `budget` iterations are done or the
// cap is reached; the lanes of a wave run together
template <int FS>
__device__ inline void em_iterate(const EmArgs &A, const double (&kd)[FS], const double (&gm)[FS][2], const double inv_n, EmPairState<FS> &P,
                                  double (&t3)[2], double &ll, double (&z3)[2], unsigned &iters, unsigned char &flag, bool &run, const unsigned budget) {
	for (unsigned it = 0; it < budget && __any(run); ++it) {
		double n3[2], nll, nz3[2];
#pragma unroll
		for (int j = 0; j < 2; ++j) n3[j] = z3[j] * inv_n;
		em_pass_lean<FS, 2, true>(kd, gm, n3, run, P, nll, nz3);
		const double crit = em_crit(ll, nll);
		if (run) {
#pragma unroll
			for (int j = 0; j < 2; ++j) { t3[j] = n3[j]; z3[j] = nz3[j]; }
			ll = nll;
			++iters;
			if (fabs(crit - 1E-6) < A.band) flag |= 1;
			if (!(crit > 1E-6)) run = false;
			else if (iters >= A.max_iters) { flag |= 2; run = false; }
		}
	}
}

template <int FS>
__global__ void __launch_bounds__(64) lsq_em_head_kernel(EmArgs A, EmTail T) {
#ifndef LSQ_EM_NO_PRIO
	__builtin_amdgcn_s_setprio(3);
#endif
	const unsigned place = A.place0 + blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned e = place < A.n_places ? A.order[place] : 0xFFFFFFFFu;
	const bool ev_ok = e != 0xFFFFFFFFu;
	int K; unsigned ib; EmCacheFlat<FS> C; double n_total;
	em_load_flat<FS>(A, e, ev_ok, K, ib, C, n_total);
	const double inv_n = 1.0 / n_total;
	// no reads: theta stays 1/K, log-likelihood 0; one isoform: theta = 1 (solve/solve.cpp:798-802)
	bool run = ev_ok && n_total > 0 && K > 1;
	const bool any_reads = ev_ok && n_total > 0;
	double kd[FS], gm[FS][2], t3[2], z3[2], ll = 0;
#pragma unroll
	for (int t = 0; t < FS; ++t) {
		kd[t] = C.kd[t];
#pragma unroll
		for (int j = 0; j < 2; ++j) gm[t][j] = (C.cls[t] >> j & 1) ? C.g[t][j] : 0.0;
	}
#pragma unroll
	for (int j = 0; j < 2; ++j) t3[j] = (K == 1) ? 1.0 : 1.0 / (double)K;   // solve/solve.cpp:798-802, read.h:642
	EmPairState<FS> P;
#pragma unroll
	for (int t = 0; t < FS; ++t) { P.s[t] = 1.0; P.r[t] = 1.0; P.lg[t] = 0.0; }
	unsigned iters = 0;
	unsigned char flag = 0;
	em_pass_lean<FS, 2, true>(kd, gm, t3, any_reads, P, ll, z3);
	em_iterate<FS>(A, kd, gm, inv_n, P, t3, ll, z3, iters, flag, run, EM_HEAD_PASSES);
	if (!ev_ok) return;
	if (run) {            // still running: the tail's
		const unsigned at = atomicAdd(&T.count[0], 1u);
		T.ev[at] = e; T.iters[at] = iters; T.flag[at] = flag; T.t0[at] = t3[0]; T.t1[at] = t3[1]; T.ll[at] = ll;
		return;
	}
#pragma unroll
	for (int j = 0; j < 2; ++j) if (j < K) A.theta[ib + j] = t3[j];
	A.logll[e] = ll;
	A.iters[e] = iters;
	A.flags[e] = flag;
}

// theta after m more iterations of the map, from the state the closed form was set up at: x = theta_0 and y = theta_1 = 1 - x,
// each formed without cancellation (an event whose reads all sit on one isoform has its fixed point ON the boundary, p = 1
// or p = 0 exactly, and the other theta decays like kappa^m: it is y = (1 - p) - (p - q) u with 1 - p = 0, not 1 - x)
struct EmMoebius {
	double p, omp, pq, lk, w0, x0, y0;      // attracting fixed point, 1 - p, p - q (linear map: 1), log kappa, w at m = 0, theta at m = 0
	bool linear;                            // G0 == G1: the map is x' = a + b x, w = x - p
	__device__ __forceinline__ void at(const unsigned m, double &x, double &y) const {
		if (m == 0u) { x = x0; y = y0; return; }
		const double w = w0 * exp((double)m * lk);
		const double u = linear ? w : w / (1.0 - w);
		x = p + pq * u;
		y = omp - pq * u;
	}
};

template <int FS>
__global__ void __launch_bounds__(64) lsq_em_tail_kernel(EmArgs A, EmTail T) {
#ifndef LSQ_EM_NO_PRIO
	__builtin_amdgcn_s_setprio(3);
#endif
	const unsigned n_list = __builtin_amdgcn_readfirstlane((int)*(volatile unsigned *)&T.count[0]);
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (blockIdx.x * blockDim.x < n_list) {
		const bool ev_ok = i < n_list;
		const unsigned e = ev_ok ? T.ev[i] : 0xFFFFFFFFu;
		int K; unsigned ib; EmCacheFlat<FS> C; double n_total;
		em_load_flat<FS>(A, e, ev_ok, K, ib, C, n_total);
		const double inv_n = 1.0 / n_total;
		double kd[FS], gm[FS][2], t3[2], z3[2], ll = ev_ok ? T.ll[i] : 0.0;
#pragma unroll
		for (int t = 0; t < FS; ++t) {
			kd[t] = C.kd[t];
#pragma unroll
			for (int j = 0; j < 2; ++j) gm[t][j] = (C.cls[t] >> j & 1) ? C.g[t][j] : 0.0;
		}
		t3[0] = ev_ok ? T.t0[i] : 0.5; t3[1] = ev_ok ? T.t1[i] : 0.5;
		unsigned iters = ev_ok ? T.iters[i] : 0u;
		unsigned char flag = ev_ok ? T.flag[i] : (unsigned char)0;
		EmPairState<FS> P;
#pragma unroll
		for (int t = 0; t < FS; ++t) { P.s[t] = 1.0; P.r[t] = 1.0; P.lg[t] = 0.0; }
		// ---- the closed form, where it applies: classes {0}, {1}, {0, 1} in the slots 0, 1, 2 (one read file).  Reads on both
		// isoforms alone: always.  On one of them alone (the fixed point may then lie on the boundary): only with at least two
		// accessible starts per isoform (G < 1) -- with G = 1 the log-likelihood itself runs to 0 there, and the test, a ratio
		// of log-likelihoods, need not stay true once it is (seen in the random check): those events keep the ordinary iteration
		const double n1 = kd[0], n2 = kd[1], n3 = kd[2], G0 = C.g[0][0], G1 = C.g[1][1];
		bool closed = ev_ok && K == 2 && FS == 3 && G0 > 0.0 && G1 > 0.0 && iters + 2u < A.max_iters &&
		              ((n1 >= 1.0 && n2 >= 1.0) || ((n1 >= 1.0 || n2 >= 1.0) && G0 < 1.0 && G1 < 1.0));
		EmMoebius M;
		M.x0 = t3[0]; M.y0 = t3[1]; M.p = 0.5; M.omp = 0.5; M.pq = 1.0; M.lk = -1.0; M.w0 = 0.0; M.linear = false;
		if (closed) {
			const double a = n1 * inv_n, b = n3 * inv_n, g = G0 - G1;
			double p, q, kap;
			if (g == 0.0) {
				M.linear = true; p = a / (1.0 - b); q = p - 1.0; kap = b;
			} else {
				const double al = a * g + b * G0, be = a * G1, de = G1;
				const double qb = de - al, D = qb * qb + 4.0 * g * be;                      // g x^2 + (de - al) x - be = 0
				const double sq = sqrt(D), tq = -0.5 * (qb + copysign(sq, qb));
				const double r1 = tq / g, r2 = -be / tq;                                     // the two roots, without cancellation
				const double k1 = (g * r2 + de) / (g * r1 + de);                             // the map's slope at r1 (at r2: its inverse)
				const bool first = k1 >= 0.0 && k1 < 1.0;
				p = first ? r1 : r2; q = first ? r2 : r1; kap = first ? k1 : 1.0 / k1;
				closed = D > 0.0 && tq != 0.0 && (first || k1 > 1.0);
			}
			// a fixed point on the boundary is exactly there
			if (n2 == 0.0 && fabs(p - 1.0) < 1E-9) p = 1.0;
			if (n1 == 0.0 && fabs(p) < 1E-9) p = 0.0;
			M.p = p; M.omp = 1.0 - p; M.pq = M.linear ? 1.0 : p - q;
			M.lk = log(kap);                                                                 // (kappa = 0: -inf, kappa^m = 0)
			const double dx = p == 1.0 ? -M.y0 : M.x0 - p;                                  // x0 - p without cancellation at the boundary
			M.w0 = M.linear ? dx : dx / (M.x0 - q);
			closed = closed && p >= 0.0 && p <= 1.0 && kap >= 0.0 && kap < 1.0 - 1E-9 && fabs(M.w0) < 1.0 && (M.linear || M.x0 != q);
		}
		// Test value of iteration s + m (m >= 1), |1 - l(m - 1) / l(m)| = |l(m) - l(m - 1)| / |l(m)|, from the closed form: the
		// step dx = x(m) - x(m - 1) = (p - q) w (kappa - 1) / ((1 - w)(1 - kappa w)) without cancellation (w = w0 kappa^(m-1)),
		// the numerator as sum n_i log1p(d_i) over the three mixtures (d_i: relative change of x G0, of y G1, of their sum),
		// the denominator as the log-likelihood at m.  Relative accuracy ~1e-13, where the guard band asks for 1e-5.
		const double kap = closed ? exp(M.lk) : 0.5, gd = G0 - G1;
		auto log1p_any = [&](const double d) __attribute__((always_inline)) { return fabs(d) < 0.03125 ? log1p_small(d) : fast_log(1.0 + d); };
		auto crit_at = [&](const unsigned m) __attribute__((always_inline)) {
			const double wa = m == 1u ? M.w0 : M.w0 * exp((double)(m - 1u) * M.lk), wb = wa * kap;      // (kappa = 0: 0 x -inf is no number)
			const double ua = M.linear ? wa : wa / (1.0 - wa), ub = M.linear ? wb : wb / (1.0 - wb);
			const double xa = m == 1u ? M.x0 : M.p + M.pq * ua, ya = m == 1u ? M.y0 : M.omp - M.pq * ua;
			const double xb = M.p + M.pq * ub, yb = M.omp - M.pq * ub;
			const double dx = M.linear ? wa * (kap - 1.0) : M.pq * wa * (kap - 1.0) / ((1.0 - wa) * (1.0 - wb));
			double num = 0.0, den = 0.0;
			if (n1 > 0.0) { num += n1 * log1p_any(dx / xa); den += n1 * fast_log(xb * G0); }
			if (n2 > 0.0) { num += n2 * log1p_any(-dx / ya); den += n2 * fast_log(yb * G1); }
			if (n3 > 0.0) { num += n3 * log1p_any(dx * gd / (xa * G0 + ya * G1)); den += n3 * fast_log(xb * G0 + yb * G1); }
			return fabs(num / den);
		};
		// The search.  lo: an iteration past `iters` at which the test fails (0: none tried), hi: one at which it holds.  First
		// m = 1; then the iteration the asymptotic decay c(m) ~ c(1) kappa^(2 (m - 1)) names; from there in doubling steps to
		// the other side, and bisection between the two: ~4 evaluations when the decay is already geometric.
		unsigned lo = 0u, hi = 1u;
		double c_lo = 1.0, c_hi = 1.0;
		const unsigned m_cap = A.max_iters - iters;
		bool capped = false;
		{
			bool up = false, down = false;                 // galloping away from the guess: towards later / earlier iterations
			unsigned g = 1u, d = 1u;
			if (closed) {
				const double c1 = crit_at(1u);
				if (!(c1 > 1E-6)) { hi = 1u; c_hi = c1; }
				else {
					lo = 1u; c_lo = c1;
					// where the asymptotic decay of the test value crosses the threshold: around an interior fixed point l'(p) = 0 and
					// l(m) - l(m - 1) ~ 1/2 |l''(p)| (p - q)^2 (1 - kappa^2) w^2, w = w0 kappa^(m - 1); on the boundary l'(p) != 0 and
					// the difference is ~ |l'(p)| (p - q) (1 - kappa) w
					const double sp = M.p * G0 + M.omp * G1;
					double lp = 0.0, d1 = 0.0, d2 = 0.0;
					if (n1 > 0.0) { lp += n1 * fast_log(M.p * G0); d1 += n1 / M.p; d2 += n1 / (M.p * M.p); }
					if (n2 > 0.0) { lp += n2 * fast_log(M.omp * G1); d1 -= n2 / M.omp; d2 += n2 / (M.omp * M.omp); }
					if (n3 > 0.0) { lp += n3 * fast_log(sp); d1 += n3 * gd / sp; d2 += n3 * gd * gd / (sp * sp); }
					const bool edge = M.p == 1.0 || M.p == 0.0;
					const double amp = edge ? fabs(d1 * M.pq * (1.0 - kap) / lp) * fabs(M.w0) : fabs(0.5 * d2 * M.pq * M.pq * (1.0 - kap * kap) / lp) * M.w0 * M.w0;
					double est = 1.999 + log(1E-6 / amp) / ((edge ? 1.0 : 2.0) * M.lk);             // (kappa = 0: x / -inf = -0)
					if (!(est > 2.0)) est = 2.0;                                                     // (no number: 2)
					g = est < (double)m_cap ? (unsigned)est : m_cap;
					g = min(g, m_cap);
					const double cg = crit_at(g);
					if (!(cg > 1E-6)) { hi = g; c_hi = cg; down = hi - lo > 1u; }
					else if (g >= m_cap) { hi = g; c_hi = cg; capped = true; }
					else { lo = g; c_lo = cg; up = true; }
				}
			}
			while (__any(up)) {
				if (up) {
					const unsigned m = min(lo + d, m_cap);
					const double c = crit_at(m);
					if (!(c > 1E-6)) { hi = m; c_hi = c; up = false; }
					else if (m >= m_cap) { hi = m; c_hi = c; capped = true; up = false; }
					else { lo = m; c_lo = c; d *= 2u; }
				}
			}
			while (__any(down)) {
				if (down) {
					const unsigned m = hi - lo > d ? hi - d : lo + 1u;
					const double c = crit_at(m);
					if (!(c > 1E-6)) { hi = m; c_hi = c; d *= 2u; down = hi - lo > 1u; }
					else { lo = m; c_lo = c; down = false; }
				}
			}
			bool bis = closed && !capped && hi - lo > 1u;  // bisection: the test, once true, stays true
			while (__any(bis)) {
				if (bis) {
					const unsigned mid = lo + (hi - lo) / 2u;
					const double c = crit_at(mid);
					if (!(c > 1E-6)) { hi = mid; c_hi = c; } else { lo = mid; c_lo = c; }
					bis = hi - lo > 1u;
				}
			}
		}
		if (closed) {
			// theta(T - 1) from the closed form, then the last step and the final log-likelihood by the ordinary pass
			if (hi > 1u) M.at(hi - 1u, t3[0], t3[1]);
			if (fabs(c_hi - 1E-6) < A.band || (lo > 0u && fabs(c_lo - 1E-6) < A.band)) flag |= 1;
			if (capped) flag |= 2;
			iters += hi - 1u;
		}
		// the numerators at t3; then: closed form -- exactly one more iteration; otherwise ordinary iterations to the end
		bool run = ev_ok;
		em_pass_lean<FS, 2, true>(kd, gm, t3, run, P, ll, z3);
		if (closed) {
			double n3v[2], nll, nz3[2];
#pragma unroll
			for (int j = 0; j < 2; ++j) n3v[j] = z3[j] * inv_n;
			em_pass_lean<FS, 2, true>(kd, gm, n3v, run, P, nll, nz3);
			const double crit = em_crit(ll, nll);
			// the ordinary arithmetic must agree with the search about this iteration; where it does not, the value sits on the
			// threshold within rounding: flagged (the replay decides)
			if (!capped && (crit > 1E-6 || fabs(crit - 1E-6) < A.band)) flag |= 1;
			t3[0] = n3v[0]; t3[1] = n3v[1]; ll = nll; ++iters;
			run = false;
		}
		{
			bool seq = run && !closed;
			em_iterate<FS>(A, kd, gm, inv_n, P, t3, ll, z3, iters, flag, seq, 0xFFFFFFFFu);
		}
		if (ev_ok) {
#pragma unroll
			for (int j = 0; j < 2; ++j) if (j < K) A.theta[ib + j] = t3[j];
			A.logll[e] = ll;
			A.iters[e] = iters;
			A.flags[e] = flag;
		}
	}
	// the last workgroup to finish clears the list for the lane's next solve
	if (threadIdx.x == 0) {
		__threadfence();
		const unsigned done = atomicAdd(&T.count[1], 1u);
		if (done == gridDim.x - 1u) { T.count[0] = 0u; T.count[1] = 0u; __threadfence(); }
	}
}

// The lean group in one launch (two on one stream would run one after the other): the first n_quad workgroups take the
// places below *A.split four lanes an event, the others the places from there on one lane an event; workgroups of 64.
template <int FS>
__global__ void __launch_bounds__(64) lsq_em_lean_kernel(EmArgs A, unsigned n_quad) {
	if (blockIdx.x < n_quad) em_quad_body<true>(A, blockIdx.x);
	else em_flat_body<FS>(A, blockIdx.x - n_quad);
}
// ... and before a placement by iteration counts exists (or with option em_regroup off): four lanes an event throughout,
// in a kernel of its own (the one-lane form's registers would come on top: 128 against 104 a wave)
__global__ void __launch_bounds__(64) lsq_em_lean_quad_kernel(EmArgs A) { em_quad_body<true>(A, blockIdx.x); }
// ... and with a cap on its passes: the few events with long chains (1 % of C3's take 100-160 dependent passes, every event of their
// waves waiting) go on a list after `cap` accepted iterations, and lsq_em_tail_kernel, launched behind this kernel, finishes them
__global__ void __launch_bounds__(64) lsq_em_lean_quad_capped_kernel(EmArgs A, EmTail T, unsigned cap) { em_quad_body<true>(A, blockIdx.x, cap, &T); }
// the other events: more than two isoforms or more than one (method, class) pair per lane
__global__ void __launch_bounds__(256) lsq_em_kernel(EmArgs A) { em_quad_body<false>(A, blockIdx.x); }

// ---- expected Fisher information and the variance estimates made from it (fim.h, linalg.h) ------------
// PARITY UNPINNED: no translation unit of the reference includes these headers.  fim.h:115-158 sums, over
// every isoform k with theta_k != 0 and every accessible read start of it, theta_k G_k times the
// observed information of the read generated there (fim.h:320-367):
//   (dG_p - dG_K)(dG_q - dG_K) / (sum_j theta_j dG_j)^2,   dG_j = G_j if the read is compatible with j, else 0.
// That term depends on the read only through its compatibility class, so the sum runs over classes with
// the number of starts per (isoform, class) counted on the host (lsq_annot.cpp: fim_start_classes).
// The matrix is (K-1) x (K-1) with K <= 6: one lane per (event, method), nothing here is a dense
// contraction worth a matrix core.  Variances: fim.h:64-72 (reciprocals of the diagonal) and :74-93 through
// linalg.h:28-71's inverse (GSL's LU with partial pivoting, then the identity's columns).
struct FimArgs {
	unsigned n_events, n_methods, n_iso;
	const unsigned char *K;
	const unsigned *iso_base, *start_base, *mat_base;
	const unsigned *starts;            // [method][starts_total]
	size_t starts_total, mat_total;
	const double *theta, *G;           // G: [method][n_iso]
	double *fim;                       // [method][mat_total]
	double *var;                       // [method][n_events][2]: by the diagonal, by the inverse
};

__global__ void __launch_bounds__(64) lsq_fim_kernel(FimArgs A) {
	const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= A.n_events * A.n_methods) return;
	const unsigned e = t / A.n_methods, m = t % A.n_methods;
	const int K = A.K[e], D = K - 1;
	const unsigned n_cls = (1u << K) - 1u;
	double th[LSQ_MAX_ISOFORMS], g[LSQ_MAX_ISOFORMS], I[(LSQ_MAX_ISOFORMS - 1) * (LSQ_MAX_ISOFORMS - 1)];
	for (int j = 0; j < K; ++j) { th[j] = A.theta[A.iso_base[e] + j]; g[j] = A.G[(size_t)m * A.n_iso + A.iso_base[e] + j]; }
	for (int i = 0; i < D * D; ++i) I[i] = 0;
	const unsigned *st = A.starts + (size_t)m * A.starts_total + A.start_base[e];
	for (int k = 0; k < K; ++k) {
		if (th[k] == 0) continue;
		for (unsigned c = 1; c <= n_cls; ++c) {
			const unsigned a = st[(unsigned)k * n_cls + c - 1];
			if (!a) continue;
			double s = 0;
			for (int j = 0; j < K; ++j) if ((c >> j & 1u) && g[j] > 0 && th[j] > 0) s += th[j] * g[j];
			const double vK = (c >> (K - 1) & 1u) ? g[K - 1] : 0.0;
			const double w = (double)a * (th[k] * g[k]) / (s * s);
			for (int p = 0; p < D; ++p) {
				const double dp = ((c >> p & 1u) ? g[p] : 0.0) - vK;
				for (int q = 0; q < D; ++q) I[p * D + q] += w * dp * (((c >> q & 1u) ? g[q] : 0.0) - vK);
			}
		}
	}
	double *out = A.fim + (size_t)m * A.mat_total + A.mat_base[e];
	for (int i = 0; i < D * D; ++i) out[i] = I[i];
	double by_diag = 0;
	for (int p = 0; p < D; ++p) by_diag += 1.0 / I[p * D + p];
	// LU with partial pivoting, rows swapped in place (gsl_linalg_LU_decomp), then P e_col through L and U
	double by_inv = 0;
	if (D > 0) {
		int perm[LSQ_MAX_ISOFORMS - 1];
		for (int i = 0; i < D; ++i) perm[i] = i;
		for (int j = 0; j < D - 1; ++j) {
			double amax = fabs(I[j * D + j]); int ip = j;
			for (int i = j + 1; i < D; ++i) { const double a = fabs(I[i * D + j]); if (a > amax) { amax = a; ip = i; } }
			if (ip != j) { for (int col = 0; col < D; ++col) { const double x = I[j * D + col]; I[j * D + col] = I[ip * D + col]; I[ip * D + col] = x; } const int x = perm[j]; perm[j] = perm[ip]; perm[ip] = x; }
			const double ajj = I[j * D + j];
			if (ajj != 0.0)
				for (int i = j + 1; i < D; ++i) {
					const double aij = I[i * D + j] / ajj;
					I[i * D + j] = aij;
					for (int col = j + 1; col < D; ++col) I[i * D + col] -= aij * I[j * D + col];
				}
		}
		for (int col = 0; col < D; ++col) {
			double x[LSQ_MAX_ISOFORMS - 1];
			for (int i = 0; i < D; ++i) x[i] = perm[i] == col ? 1.0 : 0.0;
			for (int i = 1; i < D; ++i) for (int cc = 0; cc < i; ++cc) x[i] -= I[i * D + cc] * x[cc];
			for (int i = D - 1; i >= 0; --i) { for (int cc = i + 1; cc < D; ++cc) x[i] -= I[i * D + cc] * x[cc]; x[i] /= I[i * D + i]; }
			for (int i = 0; i < D; ++i) { by_inv += x[i]; if (i == col) by_inv += x[i]; }     // sum of all entries plus the trace
		}
	}
	double *v = A.var + ((size_t)m * A.n_events + e) * 2;
	v[0] = by_diag; v[1] = by_inv;
}

// The lean group's places sorted by the iteration counts of the solve that has just finished on this lane, slowest
// first (one workgroup: a counting sort over 256 iteration classes in LDS).  Empty places go to the end.
// *split: the first place (a multiple of 64) from which on the events took fewer than EM_FLAT_BELOW iterations: those go to
// the one-lane-per-event kernel next time -- a third of the instructions per event and pass, twice the time per pass --,
// the slow ones before them stay with four lanes an event, where the time of a pass is what counts.
#ifndef LSQ_EM_FLAT_BELOW
#define LSQ_EM_FLAT_BELOW 32
#endif
constexpr unsigned EM_FLAT_BELOW = LSQ_EM_FLAT_BELOW;
__global__ void __launch_bounds__(1024) lsq_em_regroup_kernel(const unsigned *base_order, const unsigned *iters, unsigned n_places, unsigned *out, unsigned *split) {
	__shared__ unsigned hist[256], start[256];
	const unsigned tid = threadIdx.x;
	if (tid < 256) hist[tid] = 0;
	__syncthreads();
	for (unsigned p = tid; p < n_places; p += 1024) {
		const unsigned e = base_order[p];
		if (e != 0xFFFFFFFFu) atomicAdd(&hist[255u - min(iters[e], 255u)], 1u);
	}
	__syncthreads();
	if (tid == 0) {
		unsigned run = 0, slowest = 0;
		for (unsigned k = 0; k < 256; ++k) { if (hist[k] && !slowest) slowest = 255u - k; start[k] = run; run += hist[k]; hist[k] = 0; }
		out[n_places - 1] = 0xFFFFFFFFu;
		// one lane per event below EM_FLAT_BELOW iterations -- and below 0.45 of the slowest event's count when that is
		// less: a pass of that form takes twice the time, and its longest chain is not to outlast the four-lane form's
		// (configs[1]: slowest event 35 iterations; with the fixed threshold its EM took 0.021 ms instead of 0.015)
		const unsigned below = min(EM_FLAT_BELOW, max(4u, slowest * 29u / 64u));
		*split = (start[256u - below] + 63u) & ~63u;       // the events with that many iterations or more come first
	}
	__syncthreads();
	unsigned n_valid = 0;
	for (unsigned p = tid; p < n_places; p += 1024) {
		const unsigned e = base_order[p];
		if (e != 0xFFFFFFFFu) { const unsigned k = 255u - min(iters[e], 255u); out[start[k] + atomicAdd(&hist[k], 1u)] = e; }
	}
	__syncthreads();
	if (tid == 0) { for (unsigned k = 0; k < 256; ++k) n_valid += hist[k]; start[0] = n_valid; }
	__syncthreads();
	for (unsigned p = start[0] + tid; p < n_places; p += 1024) out[p] = 0xFFFFFFFFu;
}

} // namespace

namespace lsq {

int run_fim(lsq_ctx *c) {
	const lsq_events &E = *c->E;
	const unsigned n_ev = (unsigned)E.dev2out.size();
	if (!n_ev) return LSQ_OK;
	FimArgs A{};
	A.n_events = n_ev; A.n_methods = (unsigned)E.n_methods; A.n_iso = E.n_iso_total;
	A.K = c->dK.p; A.iso_base = c->iso_base.p; A.start_base = c->fim_start_base.p; A.mat_base = c->fim_mat_base.p;
	A.starts = c->fim_starts.p; A.starts_total = c->fim_starts_total; A.mat_total = c->fim_mat_total;
	A.theta = c->theta.p; A.G = c->G.p; A.fim = c->fim.p; A.var = c->fim_var.p;
	const unsigned n = n_ev * (unsigned)E.n_methods;
	hipLaunchKernelGGL(lsq_fim_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream_em, A);      // behind the solve
	HIP_TRY(hipGetLastError());
	return LSQ_OK;
}

int run_solve(lsq_ctx *c) {
	const lsq_events &E = *c->E;
	hipStream_t st = c->stream_em;
	if (c->time_events) HIP_TRY(hipEventRecord(c->ev2, st));
	const unsigned n_ev = (unsigned)E.dev2out.size();
	if (n_ev) {
		EmArgs A{};
		A.n_events = n_ev; A.n_methods = (unsigned)E.n_methods; A.n_cls = E.n_cls_total; A.n_iso = E.n_iso_total;
		A.K = c->dK.p; A.cls_base = c->cls_base.p; A.iso_base = c->iso_base.p;
		A.order = c->em_order.p;
		A.max_iters = 1000000u;
		A.band = c->em_band;
#ifdef LSQ_DEV
		if (const char *e = getenv("LSQ_EM_CAP")) { const int v = atoi(e); if (v > 0) A.max_iters = (unsigned)v; }      // timing experiments
#endif
		A.cnt = c->cnt.p; A.G = c->G.p; A.theta = c->theta.p; A.logll = c->logll.p; A.iters = c->iters.p; A.flags = c->flags.p;
		// one wave per workgroup: beside a streaming kernel that fills the device, a wave that is done gives its
		// registers back without waiting for three others (measured 0.259 -> 0.254 ms per pipelined step)
		const unsigned blk = 64;
		if (c->em_small_places && c->opt_em_closed && E.n_methods == 1) {
			// two isoforms, one read file: a head of ordinary iterations for everybody, the closed form for what is left
			// (no placement by earlier iteration counts, no chain of hundreds of passes)
			const int lane = c->flip;
			A.place0 = 0; A.n_places = c->em_small_places;
			EmTail T{};
			T.count = c->em_tail_count.p + 2 * lane;
			T.ev = c->em_tail_u32[lane].p; T.iters = T.ev + c->em_small_places;
			T.flag = c->em_tail_flag[lane].p;
			T.t0 = c->em_tail_f64[lane].p; T.t1 = T.t0 + c->em_small_places; T.ll = T.t1 + c->em_small_places;
			const unsigned n_wg = (c->em_small_places + blk - 1) / blk;
			hipLaunchKernelGGL(lsq_em_head_kernel<3>, dim3(n_wg), dim3(blk), 0, st, A, T);
			hipLaunchKernelGGL(lsq_em_tail_kernel<3>, dim3(n_wg), dim3(blk), 0, st, A, T);
			HIP_TRY(hipGetLastError());
		} else if (c->em_small_places) {
			const int lane = c->flip;
			A.place0 = 0; A.n_places = c->em_small_places;
			const bool regrouped = c->opt_em_regroup && c->em_order_lane_valid[lane];
			if (regrouped) A.order = c->em_order_lane[lane].p;
			// one lane per event only for a job's worth of events: with a few thousand (configs[1], a rank's eighth of configs[2])
			// a step waits for the EM's chain, not for its instructions (C2: 0.0485 ms per step without, 0.0497 with)
			const bool flat = regrouped && c->em_small_places >= c->opt_em_flat_min;
			A.split = c->em_split.p + (flat ? lane : 2);          // (word 2: every place to the four-lane kernel)
			const unsigned n_quad = (c->em_small_places * EM_LANES + blk - 1) / blk, n_flat = flat ? (c->em_small_places + 63u) / 64u : 0u;
			if (!n_flat && c->opt_em_quad_cap && E.n_methods == 1) {
				EmTail T{};
				T.count = c->em_tail_count.p + 2 * lane;
				T.ev = c->em_tail_u32[lane].p; T.iters = T.ev + c->em_small_places;
				T.flag = c->em_tail_flag[lane].p;
				T.t0 = c->em_tail_f64[lane].p; T.t1 = T.t0 + c->em_small_places; T.ll = T.t1 + c->em_small_places;
				hipLaunchKernelGGL(lsq_em_lean_quad_capped_kernel, dim3(n_quad), dim3(blk), 0, st, A, T, c->opt_em_quad_cap);
				hipLaunchKernelGGL(lsq_em_tail_kernel<3>, dim3((c->em_small_places + blk - 1) / blk), dim3(blk), 0, st, A, T);
			} else if (!n_flat) hipLaunchKernelGGL(lsq_em_lean_quad_kernel, dim3(n_quad), dim3(blk), 0, st, A);
			else if (E.n_methods == 1) hipLaunchKernelGGL(lsq_em_lean_kernel<3>, dim3(n_quad + n_flat), dim3(blk), 0, st, A, n_quad);
			else hipLaunchKernelGGL(lsq_em_lean_kernel<4>, dim3(n_quad + n_flat), dim3(blk), 0, st, A, n_quad);
			HIP_TRY(hipGetLastError());
			if (c->opt_em_regroup && (!c->em_order_lane_valid[lane] || ++c->em_regroup_age[lane] >= 16)) {
				c->em_regroup_age[lane] = 0;
				if (c->em_order_lane[lane].n != c->em_small_places) { int rc = c->em_order_lane[lane].alloc(c->em_small_places); if (rc) return rc; }
				hipLaunchKernelGGL(lsq_em_regroup_kernel, dim3(1), dim3(1024), 0, st, c->em_order.p, c->iters.p, c->em_small_places, c->em_order_lane[lane].p, c->em_split.p + lane);
				HIP_TRY(hipGetLastError());
				c->em_order_lane_valid[lane] = true;
			}
		}
		if (c->em_places > c->em_small_places) {
			A.order = c->em_order.p;
			A.place0 = c->em_small_places; A.n_places = c->em_places;
			hipLaunchKernelGGL(lsq_em_kernel, dim3(((c->em_places - c->em_small_places) * EM_LANES + 255) / 256), dim3(256), 0, st, A);
			HIP_TRY(hipGetLastError());
		}
	}
	if (c->time_events) HIP_TRY(hipEventRecord(c->ev3, st));
	c->solve_timed = c->time_events;
	return LSQ_OK;
}

} // namespace lsq
