// Host side of the path, part 1: annotation loading, event compilation and the device plan
// (buckets + LDS images).  Semantics restated from the reference lines cited at each step.
#include <algorithm>
#include <atomic>
#include <ctime>
#include <cerrno>
#include <climits>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "lsq_internal.hpp"

namespace lsq {

static thread_local std::string g_err;

int fail(int status, const char *fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	try { g_err = buf; } catch (...) { g_err.clear(); }      // (the text is lost before the status is)
	return status;
}

// a warning of the library itself, in the reference's log format (jsc/util/log.hpp:22-79) on stderr, when the reporting level lets
// warnings through (lsq_set_log_level; the executables pass their log_level argument on)
static std::atomic<int> g_log_level{2};
void warn(const char *fmt, ...) {
	if (g_log_level.load() < 1) return;
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	time_t raw; time(&raw);
	struct tm tmv; localtime_r(&raw, &tmv);
	fprintf(stderr, "[LOG %d-%02d-%02d %02d:%02d:%02d WARNING] %s\n", tmv.tm_year + 1900, tmv.tm_mon + 1, tmv.tm_mday, tmv.tm_hour, tmv.tm_min, tmv.tm_sec, buf);
	fflush(stderr);
}
void set_log_level(int level) { g_log_level.store(level); }

// the catch-all of an extern "C" entry (LSQ_API_CATCH): called from inside a catch block, rethrows to read the exception
int fail_exception(const char *where) noexcept {
	try { throw; }
	catch (const std::bad_alloc &) { return fail(LSQ_E_INTERNAL, "%s: out of memory", where); }
	catch (const std::exception &e) { return fail(LSQ_E_INTERNAL, "%s: %s", where, e.what()); }
	catch (...) { return fail(LSQ_E_INTERNAL, "%s: unknown exception", where); }
}

int host_threads(int requested) {
	if (requested > 0) return requested;
	if (const char *e = getenv("LSQ_THREADS")) { int v = atoi(e); if (v > 0) return v; }
	unsigned hc = std::thread::hardware_concurrency();
	return hc ? (int)std::min(hc, 32u) : 4;
}

int Dict::intern(const std::string &s) {
	std::lock_guard<std::mutex> g(mu);
	auto it = ids.find(s);
	if (it != ids.end()) return it->second;
	int id = (int)names.size();
	names.push_back(s);
	ids.emplace(s, id);
	return id;
}
int Dict::find(const std::string &s) const {
	auto it = ids.find(s);
	return it == ids.end() ? -1 : it->second;
}

// ---- interval_list::add_interval (interval_list.hpp:462-503) -------------------------------
// The reference edits its `starts` and `ends` vectors independently, driven by four
// lower_bound positions.  Consequence worth knowing: an interval that ends exactly where the
// new one starts is merged with it, one that starts exactly where the new one ends is not.
void IntervalList::add(int64_t start, int64_t end) {
	if (!(start < end)) return;
	auto ss = std::lower_bound(s.begin(), s.end(), start) - s.begin();
	auto se = std::lower_bound(e.begin(), e.end(), start) - e.begin();
	auto es = std::lower_bound(s.begin(), s.end(), end) - s.begin();
	auto ee = std::lower_bound(e.begin(), e.end(), end) - e.begin();
	bool start_inside = (ss - se == 1), end_inside = (es - ee == 1);
	auto sit = s.erase(s.begin() + ss, s.begin() + es);
	auto eit = e.erase(e.begin() + se, e.begin() + ee);
	if (!start_inside) s.insert(sit, start);
	if (!end_inside) e.insert(eit, end);
}
// interval_list::contains_interval (interval_list.hpp:396-422): inside ONE stored interval
bool IntervalList::contains(int64_t start, int64_t end) const {
	if (!(start < end)) return true;
	size_t i = std::lower_bound(s.begin(), s.end(), start) - s.begin();
	if (i < s.size() && s[i] <= start && end <= e[i]) return true;
	if (i >= 1 && i - 1 < s.size() && s[i - 1] <= start && end <= e[i - 1]) return true;
	return false;
}

// ---- ExonSet::insert (splicing_graph.h:88-169) ----------------------------------------------
// Segments are kept ordered by start with unique starts (the reference's std::set compares
// starts only, so inserting a piece whose start is already present is a no-op).
namespace {
struct Seg { int64_t start, end; };

void put_unique(std::vector<Seg> &v, Seg x) {
	if (!(x.start < x.end)) return;
	auto it = std::lower_bound(v.begin(), v.end(), x.start, [](const Seg &a, int64_t s) { return a.start < s; });
	if (it != v.end() && it->start == x.start) return;
	v.insert(it, x);
}

void exon_insert(std::vector<Seg> &v, int64_t start, int64_t end) {
	Seg nw{start, end};
	std::vector<Seg> pieces;
	size_t i = std::lower_bound(v.begin(), v.end(), nw.start, [](const Seg &a, int64_t s) { return a.start < s; }) - v.begin();
	if (!v.empty() && i != 0) --i;
	for (; i < v.size() && v[i].start < nw.end; ++i) {
		Seg &old = v[i];
		if (old.start == nw.start) {
			if (old.end > nw.end) { pieces.push_back({nw.end, old.end}); old.end = nw.end; nw.start = nw.end; }
			else nw.start = old.end;
		} else if (old.start < nw.start) {
			if (old.end > nw.start) {
				if (old.end > nw.end) {          // new exon strictly inside the old one
					pieces.push_back({nw.start, nw.end});
					pieces.push_back({nw.end, old.end});
					old.end = nw.start;
					nw.start = nw.end;
				} else {                          // overlap on the old exon's right side
					pieces.push_back({nw.start, old.end});
					int64_t old_end = old.end;
					old.end = nw.start;
					nw.start = old_end;
				}
			}
		} else {
			if (old.end > nw.end) {               // overlap on the old exon's left side
				pieces.push_back({nw.end, old.end});
				old.end = nw.end;
				nw.end = old.start;
			} else {                              // old exon inside the new one
				pieces.push_back({nw.start, old.start});
				nw.start = old.end;
			}
		}
	}
	put_unique(v, nw);
	for (const Seg &p : pieces) put_unique(v, p);
}

// boost::lexical_cast-like strict integer (whole token, optional sign, digits)
bool strict_long(const std::string &t, int64_t &out) {
	if (t.empty() || t.size() > 40) return false;
	size_t i = (t[0] == '+' || t[0] == '-') ? 1 : 0;
	if (i == t.size()) return false;
	for (size_t j = i; j < t.size(); ++j) if (t[j] < '0' || t[j] > '9') return false;
	errno = 0;
	char *endp;
	long v = strtol(t.c_str(), &endp, 10);
	if (errno == ERANGE || *endp) return false;
	out = v;
	return true;
}

// tokenizer(",") + atol per token (count/count.cpp:154-168)
void split_atol(const std::string &s, std::vector<int64_t> &out) {
	size_t i = 0;
	while (i < s.size()) {
		while (i < s.size() && s[i] == ',') ++i;
		if (i >= s.size()) break;
		size_t j = s.find(',', i);
		if (j == std::string::npos) j = s.size();
		out.push_back(atol(s.substr(i, j - i).c_str()));
		i = j;
	}
}

// the tokens `iss >> a >> b ...` would extract from a line: maximal runs of non-space characters
// (isspace of the "C" locale), at most `max_tok` of them
size_t split_ws(const std::string &line, std::string *tok, size_t max_tok) {
	size_t n = 0, i = 0;
	const size_t len = line.size();
	while (n < max_tok) {
		while (i < len && isspace((unsigned char)line[i])) ++i;
		if (i >= len) break;
		size_t j = i;
		while (j < len && !isspace((unsigned char)line[j])) ++j;
		tok[n++].assign(line, i, j - i);
		i = j;
	}
	for (size_t k = n; k < max_tok; ++k) tok[k].clear();
	return n;
}

// `while (getline(ifs, line) && !ifs.eof())`: only newline-terminated lines are seen
bool read_lines(const char *path, std::vector<std::string> &lines) {
	FILE *f = fopen(path, "rb");
	if (!f) return false;
	std::string data;
	char buf[1 << 16];
	size_t n;
	while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
	fclose(f);
	size_t pos = 0;
	while (pos < data.size()) {
		size_t nl = data.find('\n', pos);
		if (nl == std::string::npos) break;
		lines.emplace_back(data, pos, nl - pos);
		pos = nl + 1;
	}
	return true;
}
} // namespace

} // namespace lsq

using namespace lsq;

extern "C" {

const char *lsq_last_error(void) { return lsq::g_err.c_str(); }

// developer aid (include/lesseq_hip_dev.h): an exception below the boundary, as a loader or formatter out of memory would throw it
int lsq_debug_throw(int kind) LSQ_API_TRY {
	if (kind == 0) throw std::bad_alloc();
	if (kind == 1) throw std::length_error("vector::_M_default_append");
	if (kind == 2) throw 42;
	if (kind == 3) {
		lsq::ThreadGroup th;
		for (int t = 0; t < 3; ++t) th.spawn([t] { if (t == 1) throw std::runtime_error("thrown inside a helper thread"); });
		th.join();
		if (th.failed()) return lsq::fail(LSQ_E_INTERNAL, "a helper thread failed: %s", th.error().c_str());
	}
	return LSQ_OK;
} LSQ_API_CATCH
int lsq_abi_version(void) { return LSQ_ABI_VERSION; }
void lsq_set_log_level(int level) { lsq::set_log_level(level); }
void lsq_free(void *p) { free(p); }

// count/count.cpp:135-216; the formats beyond LH_GENE_TXT / UCSC_GENE2ISOFORM are solve's (solve/solve.cpp:152-329)
int lsq_annotation_load(const char *isoform_format, const char *isoforms_path,
                        const char *g2i_format, const char *g2i_path,
                        uint64_t gene_begin_idx, uint64_t gene_end_idx, lsq_annotation **out) LSQ_API_TRY {
	if (!isoform_format || !isoforms_path || !g2i_format || !g2i_path || !out) return fail(LSQ_E_ARG, "null argument");
	std::vector<std::string> lines;
	if (!read_lines(isoforms_path, lines)) return fail(LSQ_E_IO, "cannot open isoforms file %s", isoforms_path);
	std::unique_ptr<lsq_annotation> a(new lsq_annotation);
	const std::string ifmt = isoform_format;
	// isoforms given exon by exon (one line per exon, any order): name -> chromosome, strand and the
	// exons merged by interval_list::add_interval in file order (solve/solve.cpp:160-204,236-296)
	struct Grouped { std::string chrom, strand; IntervalList il; };
	std::map<std::string, Grouped> grouped;           // std::map: the reference walks a std::set of the names
	auto add_exon = [&](const std::string &iname, const std::string &chrom, const std::string &strand, int64_t start, int64_t end) {
		Grouped &g = grouped[iname];
		g.chrom = chrom; g.strand = strand;           // the last line of a name wins, as in the reference
		g.il.add(start - 1, end);
	};
	auto trim_quotes = [](std::string &t) {
		size_t b = 0, e = t.size();
		while (b < e && t[b] == '"') ++b;
		while (e > b && t[e - 1] == '"') --e;
		t = t.substr(b, e - b);
	};
	if (ifmt == "LH_GENE_TXT" || ifmt == "UCSC_GENE_TXT") {
		// LH_GENE_TXT: name chrom strand txStart txEnd exonCount exonStarts exonEnds (count/count.cpp:142-171);
		// UCSC_GENE_TXT has cdsStart cdsEnd between txEnd and exonCount (jsc/bioinfo/gene_anno.hpp:59-101)
		const bool ucsc = ifmt == "UCSC_GENE_TXT";
		std::string tok[10];
		for (const std::string &line : lines) {
			std::unique_ptr<IsoRec> r(new IsoRec);
			split_ws(line, tok, ucsc ? 10 : 8);
			const size_t o = ucsc ? 2 : 0;
			r->name = tok[0]; r->chrom = tok[1]; r->strand = tok[2];
			const std::string &t_s = tok[3], &t_e = tok[4], &cds_s = tok[5], &cds_e = tok[6], &t_n = tok[5 + o], &starts = tok[6 + o], &ends = tok[7 + o];
			int64_t cnt = 0, ignore = 0;
			if (strict_long(t_s, r->txStart) && strict_long(t_e, r->txEnd) && (!ucsc || (strict_long(cds_s, ignore) && strict_long(cds_e, ignore))) &&
			    strict_long(t_n, cnt) && cnt >= 0) {
				r->exonCount = (uint64_t)cnt;
				split_atol(starts, r->exonStarts);
				split_atol(ends, r->exonEnds);
			}
			a->recs.push_back(std::move(r));
		}
	} else if (ifmt == "UCSC_GFF" || ifmt == "WORMBASE_GFF2") {
		// chr source feature start end score strand frame [group] name; UCSC_GFF skips two header lines,
		// WORMBASE_GFF2 has one more column before the name and its chromosomes lack the "chr"
		const bool worm = ifmt == "WORMBASE_GFF2";
		for (size_t li = worm ? 0 : 2; li < lines.size(); ++li) {
			std::istringstream iss(lines[li]);
			long start = 0, end = 0;
			std::string iname, chr, tmp, strand;
			iss >> chr >> tmp >> tmp >> start >> end >> tmp >> strand >> tmp;
			if (worm) iss >> tmp;
			iss >> iname;
			if (iss.fail()) return fail(LSQ_E_ARG, "line %zu of %s does not have the %s columns (the reference reads uninitialised coordinates here)", li + 1, isoforms_path, isoform_format);
			trim_quotes(iname);
			add_exon(iname, worm ? "chr" + chr : chr, strand, start, end);
		}
	} else if (ifmt == "GENELETS_GFF3") {
		// two header lines; "exon" lines name their isoforms in a Parent=a,b attribute (solve/solve.cpp:160-204)
		for (size_t li = 2; li < lines.size(); ++li) {
			std::istringstream iss(lines[li]);
			std::string chr, tmp, type;
			iss >> chr >> tmp >> type;
			if (type != "exon") continue;
			long start = 0, end = 0;
			std::string strand, infos;
			iss >> start >> end >> tmp >> strand >> tmp >> infos;
			if (iss.fail()) return fail(LSQ_E_ARG, "exon line %zu of %s does not have the GENELETS_GFF3 columns (the reference reads uninitialised coordinates here)", li + 1, isoforms_path);
			size_t i = 0;
			while (i <= infos.size()) {
				size_t j = infos.find(';', i);
				if (j == std::string::npos) j = infos.size();
				const std::string info = infos.substr(i, j - i);
				if (info.size() > 7 && info.compare(0, 7, "Parent=") == 0) {
					const std::string names = info.substr(7);
					size_t u = 0;
					while (u <= names.size()) {
						size_t v = names.find(',', u);
						if (v == std::string::npos) v = names.size();
						if (v > u) add_exon(names.substr(u, v - u), "chr" + chr, strand, start, end);
						u = v + 1;
					}
				}
				i = j + 1;
			}
		}
	} else {
		return fail(LSQ_E_FORMAT, "Unknown file format error: %s", isoform_format);
	}
	for (auto &kv : grouped) {
		std::unique_ptr<IsoRec> r(new IsoRec);
		r->name = kv.first; r->chrom = kv.second.chrom; r->strand = kv.second.strand;
		r->exonCount = kv.second.il.size();
		r->exonStarts = kv.second.il.s; r->exonEnds = kv.second.il.e;
		a->recs.push_back(std::move(r));
	}
	// iname2gap: the last record of a name wins (count/count.cpp:176-179)
	std::unordered_map<std::string, const IsoRec *> by_name;
	for (auto &r : a->recs) by_name[r->name] = r.get();

	lines.clear();
	if (!read_lines(g2i_path, lines)) return fail(LSQ_E_IO, "cannot open gene->isoform file %s", g2i_path);
	const std::string gfmt = g2i_format;
	if (gfmt != "UCSC_GENE2ISOFORM" && gfmt != "WORMBASE_GENE2ISOFORMS") return fail(LSQ_E_FORMAT, "Unknown file format error: %s", g2i_format);
	std::map<std::string, std::vector<const IsoRec *>> genes;   // std::map: bytewise key order == std::set<string>
	std::string gtok[2];
	for (const std::string &line : lines) {
		split_ws(line, gtok, 2);
		const std::string &g = gtok[0], &names = gtok[1];
		// UCSC_GENE2ISOFORM: gene isoform; WORMBASE_GENE2ISOFORMS: gene iso1;iso2;... (solve/solve.cpp:310-329)
		std::vector<std::string> inames;
		if (gfmt == "UCSC_GENE2ISOFORM") inames.push_back(names);
		else {
			size_t u = 0;
			while (u < names.size()) {
				size_t v = names.find(';', u);
				if (v == std::string::npos) v = names.size();
				if (v > u) inames.push_back(names.substr(u, v - u));
				u = v + 1;
			}
			genes[g];      // the gene exists even with an empty list
		}
		for (const std::string &i : inames) {
			auto it = by_name.find(i);
			if (it == by_name.end())
				return fail(LSQ_E_ARG, "gene %s names isoform '%s' that is not in %s (the reference dereferences a null record here)", g.c_str(), i.c_str(), isoforms_path);
			genes[g].push_back(it->second);
		}
	}
	a->n_genes_loaded = (int64_t)genes.size();
	uint64_t idx = 0;
	for (auto &kv : genes) {
		if (idx >= gene_begin_idx && idx < gene_end_idx) {
			Gene g;
			g.name = kv.first;
			g.isos = kv.second;
			a->selected.push_back(std::move(g));
		}
		++idx;
	}
	*out = a.release();
	return LSQ_OK;
} LSQ_API_CATCH
void lsq_annotation_free(lsq_annotation *a) { delete a; }
int64_t lsq_annotation_num_genes(const lsq_annotation *a) { return a ? (int64_t)a->selected.size() : 0; }
int64_t lsq_annotation_num_isoforms_loaded(const lsq_annotation *a) { return a ? (int64_t)a->recs.size() : 0; }
int64_t lsq_annotation_num_genes_loaded(const lsq_annotation *a) { return a ? a->n_genes_loaded : 0; }

} // extern "C"

namespace lsq {

// accessible_read_starts.h:48-89 (MEDIUM) and :221-274 (SHORT, min_partial_exon_size = 0):
// walking the isoform's segments, every segment before the one where the remaining
// transcript gets shorter than a read contributes l (MEDIUM) or l + 1 (SHORT: the union
// [0, max(l-R+1,0)) U [max(l-R,0), l+1) is [0, l+1)); that segment contributes
// max(l + 1 - (c + R - L), 0); later segments nothing.
static uint64_t ars_total(const std::vector<uint64_t> &seg_len, uint64_t R, bool short_read) {
	uint64_t L = 0;
	for (uint64_t l : seg_len) L += l;
	uint64_t total = 0, c = 0;
	for (uint64_t l : seg_len) {
		c += l;
		if (c + R > L) {
			int64_t v = (int64_t)l + 1 - (int64_t)(c + R - L);
			total += (uint64_t)std::max<int64_t>(v, 0);
			break;
		}
		total += short_read ? (l > 0 ? l + 1 : 0) : l;
	}
	return total;
}

// fim.h:115-158 sums over every accessible read start a of every isoform k the read generated there
// (read.h:276-329: the isoform's segments from the one that holds the start to the one that holds the
// last base).  The generated read, hence its compatibility class, only changes where the start crosses
// a segment boundary or the end does, so the starts are counted in runs.  Accessible starts
// (accessible_read_starts.h:48-89,221-274, :131-182): in isoform coordinates segment i contributes the
// starts [c_{i-1}, c_{i-1} + a_i) with a_i = l_i (MEDIUM) or l_i + 1 (SHORT -- the last of them is the
// first base of the next segment, which that segment contributes again: the reference counts it twice),
// and the segment where less than a read is left contributes max(l + 1 - (c + R - L), 0).
static void fim_start_classes(const lsq::Event &e, int k, uint64_t R, bool short_read, uint32_t *counts /* n_cls */) {
	std::vector<int> seg;                       // the isoform's segments, ascending
	for (int n = 0; n < e.N; ++n) if (e.iso_mask[k] >> n & 1) seg.push_back(n);
	const int n = (int)seg.size();
	std::vector<uint64_t> cum(n);               // cumulative segment lengths
	uint64_t L = 0;
	for (int i = 0; i < n; ++i) { L += (uint64_t)(e.seg_e[seg[i]] - e.seg_s[seg[i]]); cum[i] = L; }
	auto class_of = [&](int se, int ee) {       // read.h:44-79 on the run of segments seg[se..ee]
		uint64_t mask = 0;
		for (int i = se; i <= ee; ++i) mask |= 1ull << seg[i];
		const int hi = 63 - __builtin_clzll(mask), lo = __builtin_ctzll(mask);
		const uint64_t span = (hi == 63 ? ~0ull : ((2ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
		unsigned cls = 0;
		for (int j = 0; j < e.K; ++j) if ((mask & ~e.iso_mask[j]) == 0 && (e.iso_mask[j] & span) == mask) cls |= 1u << j;
		return cls;
	};
	uint64_t before = 0;
	for (int i = 0; i < n; ++i) {
		const uint64_t l = cum[i] - before;
		uint64_t a;                               // accessible starts this segment contributes
		bool last = false;
		if (cum[i] + R > L) { const int64_t v = (int64_t)l + 1 - (int64_t)(cum[i] + R - L); a = (uint64_t)std::max<int64_t>(v, 0); last = true; }
		else a = short_read ? l + 1 : l;
		uint64_t s = before;                      // isoform coordinate of the start
		const uint64_t s_end = before + a;
		while (s < s_end) {
			int se = 0; while (se < n && !(cum[se] > s)) ++se;            // upper_bound(start)
			int ee = 0; while (ee < n && cum[ee] < s + R) ++ee;          // lower_bound(start + R)
			if (se >= n || ee >= n) break;                               // (cannot happen: s + R <= L by construction)
			// the run ends where the start leaves segment se or the end leaves segment ee
			const uint64_t nxt = std::min(std::min(cum[se], cum[ee] - R + 1), s_end);
			const unsigned cls = class_of(se, ee);
			if (cls) counts[cls - 1] += (uint32_t)(nxt - s);
			s = nxt;
		}
		before = cum[i];
		if (last) break;
	}
}

static const int64_t COORD_LIMIT = (int64_t)1 << 30;

// Budget for one bucket's event tables in LDS (records + segments + isoform masks + class
// histogram; the bin directory comes on top, at most 8 KiB).  Small buckets keep the per-
// workgroup staging and flush cost low and the occupancy high; the price is a longer cut list
// for the ingest-time bucket lookup.  LSQ_LDS_BUDGET overrides (tests use it to vary the split).
static uint32_t lds_budget_bytes() {
	if (const char *e = getenv("LSQ_LDS_BUDGET")) { long v = atol(e); if (v >= 512 && v <= 96 * 1024) return (uint32_t)v; }
	return 8192;
}

constexpr int64_t BUCKET_SPAN_CAP = 1 << 19;        // 256 bins of 2 048 bases

// Builds buckets + LDS images from the compiled events.
int plan_device(lsq_events &E) {
	const size_t n = E.ev.size();
	E.lds_budget = lds_budget_bytes();
	E.buckets.clear(); E.images.clear(); E.dev2out.clear(); E.ties.clear();
	E.jg_keys.clear(); E.jg_base.clear();
	E.dev_cls_base.clear(); E.dev_iso_base.clear(); E.dev_K.clear();
	E.cut_lo.assign(E.chroms.names.size(), {});
	E.clu_s.assign(E.chroms.names.size(), {});
	E.clu_e.assign(E.chroms.names.size(), {});
	E.chrom_first_bucket.assign(E.chroms.names.size(), -1);
	E.n_cls_total = E.n_iso_total = 0;
	E.max_lds_bytes = 0;

	// events per chromosome, ordered by span start
	std::vector<std::vector<int32_t>> per_chrom(E.chroms.names.size());
	for (size_t i = 0; i < n; ++i)
		if (i >= E.shard_first && i - E.shard_first < E.shard_count) per_chrom[E.ev[i].chrom_id].push_back((int32_t)i);
	auto ev_bytes = [&](const Event &e) -> uint32_t {
		// packed bucket: record 48 B, ~2 cell records of 32 B per segment (the segment's stretches and the gap behind it),
		// 8 bin records of 16 B, class histogram
		return std::max(48u + 64u * (uint32_t)e.N + 40u + 128u, 16u + 8u * (uint32_t)e.N + 4u * (uint32_t)e.K) + 8u * HIST_REPLICAS * ((1u << e.K) - 1u);
	};
	for (size_t c = 0; c < per_chrom.size(); ++c) {
		auto &lst = per_chrom[c];
		if (lst.empty()) continue;
		std::sort(lst.begin(), lst.end(), [&](int32_t a, int32_t b) {
			const Event &x = E.ev[a], &y = E.ev[b];
			if (x.gene_start != y.gene_start) return x.gene_start < y.gene_start;
			if (x.gene_end != y.gene_end) return x.gene_end < y.gene_end;
			return a < b;
		});
		// the branch-free walk assumes non-negative coordinates (the reference's cursor starts at 0,
		// common/read.h:217); covered regions of a chromosome are unions of its exons
		const bool chrom_nonneg = c >= E.covered.size() || E.covered[c].s.empty() || E.covered[c].s.front() >= 0;
		// clusters of transitively overlapping spans (a read start p is a candidate of an event
		// iff gene_start <= p <= gene_end), packed greedily into buckets
		size_t i = 0;
		while (i < lst.size()) {
			size_t b_begin = i;
			uint32_t bytes = 0;
			size_t n_ev = 0;
			bool host_bucket = false;
			while (i < lst.size()) {
				size_t j = i;
				int64_t max_end = E.ev[lst[j]].gene_end;
				uint32_t cb = 0;
				size_t cn = 0;
				while (j < lst.size() && (j == i || E.ev[lst[j]].gene_start <= max_end)) {
					max_end = std::max(max_end, E.ev[lst[j]].gene_end);
					cb += ev_bytes(E.ev[lst[j]]);
					++cn; ++j;
				}
				if (E.clu_s[c].empty() || E.clu_s[c].back() != (int32_t)E.ev[lst[i]].gene_start || E.clu_e[c].back() != (int32_t)max_end) {
					// (a cluster that does not fit the current bucket is looked at again when the next bucket starts)
					E.clu_s[c].push_back((int32_t)E.ev[lst[i]].gene_start);
					E.clu_e[c].push_back((int32_t)max_end);
				}
				uint32_t cap = E.lds_budget;
				// a cluster with an event beyond the kernels' limits, or too large for the CU's LDS, is a bucket of its own
				// that the host evaluates (BucketDesc::kind 2)
				bool cluster_host = cb > 96u * 1024u || cn > 60000;
				for (size_t q = i; q < j; ++q) { const Event &e = E.ev[lst[q]]; if (e.K > LSQ_MAX_ISOFORMS || e.N > LSQ_MAX_SEGMENTS) cluster_host = true; }
				if (cluster_host) {
					if (n_ev > 0) break;
					host_bucket = true; n_ev = cn; i = j;
					break;
				}
				if (n_ev > 0 && (bytes + cb > cap || n_ev + cn > 60000)) break;
				// ... and a bucket stays short on the chromosome: its coordinate bins should be no wider than the ingest's
				// per-bin sort handles (2 048 bases) and hold a cell or two each.  A sparse set of events -- a shard's slice of
				// the name-sorted list holds some -- would otherwise make buckets of megabases with hundreds of cells per bin
				// (measured on a half-job shard: 21 such buckets of 1 381 made its count kernel 0.21 ms where 0.11 is due).
				if (n_ev > 0 && max_end - E.ev[lst[b_begin]].gene_start > BUCKET_SPAN_CAP) break;
				bytes += cb; n_ev += cn; i = j;
			}
			// ---- emit bucket [b_begin, i)
			BucketDesc d;
			memset(&d, 0, sizeof d);
			d.chrom_id = (int32_t)c;
			d.n_events = (uint32_t)(i - b_begin);
			int64_t lo = E.ev[lst[b_begin]].gene_start, hi = lo;
			uint32_t nseg = 0, niso = 0, ncls = 0;
			bool fast = chrom_nonneg;
			for (size_t k = b_begin; k < i; ++k) {
				const Event &e = E.ev[lst[k]];
				hi = std::max(hi, e.gene_end);
				nseg += (uint32_t)e.N; niso += (uint32_t)e.K; ncls += (1u << e.K) - 1u;
				if (e.N > 4 || e.K > 4 || e.seg_s.front() < 0 || e.gene_start != e.seg_s.front()) fast = false;
			}
			if (getenv("LSQ_FORCE_GENERIC")) fast = false;
			if (host_bucket) {
				// no LDS image: the reads that start in the bucket's range are pooled as for any bucket (one bin), the count
				// kernels pass the bucket over, the host evaluates it
				fast = false;
				d.kind = 2u; d.n_bins = 1; d.shift = 31; d.lo = (int32_t)lo; d.hi = (int32_t)hi;
				d.n_cls = ncls; d.cls_base = E.n_cls_total; d.ev_base = (uint32_t)E.dev2out.size();
				uint32_t io = 0, co = 0;
				for (size_t k = b_begin; k < i; ++k) {
					const Event &e = E.ev[lst[k]];
					E.dev2out.push_back(lst[k]);
					E.dev_cls_base.push_back(E.n_cls_total + co);
					E.dev_iso_base.push_back(E.n_iso_total + io);
					E.dev_K.push_back((uint8_t)e.K);
					TieRec t;
					memset(&t, 0, sizeof t);
					t.strand_id = (uint8_t)e.strand_id;
					E.ties.push_back(t);
					io += (uint32_t)e.K; co += (1u << e.K) - 1u;
				}
				E.n_cls_total += ncls; E.n_iso_total += niso;
				if (E.chrom_first_bucket[c] < 0) E.chrom_first_bucket[c] = (int32_t)E.buckets.size();
				E.cut_lo[c].push_back((int32_t)lo);
				E.jg_base.push_back((uint32_t)E.jg_keys.size());
				E.buckets.push_back(d);
				continue;
			}
			if (nseg > 65535 || niso > 65535 || ncls > 65535) return fail(LSQ_E_UNSUPPORTED, "bucket tables exceed 16-bit offsets");
			uint32_t want = 16;
			while (want < 8 * d.n_events && want < 4096) want <<= 1;
			while ((((uint64_t)(hi - lo)) >> 11) >= want && want < 512) want <<= 1;       // few events far apart: still bins of 2 048 bases
			uint32_t shift = 0;
			while ((((uint64_t)(hi - lo)) >> shift) >= want) ++shift;
			d.n_bins = want; d.shift = shift; d.lo = (int32_t)lo; d.hi = (int32_t)hi;
			if (getenv("LSQ_DUMP_PLAN")) fprintf(stderr, "bucket lo=%lld hi=%lld events=%u bins=%u shift=%u\n", (long long)lo, (long long)hi, d.n_events, want, shift);
			d.kind = fast ? 1u : 0u;
			auto align16 = [](uint32_t x) { return (x + 15u) & ~15u; };
			uint32_t off = 0;
			uint32_t bins_off = off; off = align16(off + (fast ? 16u : 2u) * d.n_bins);
			// cells (packed buckets only): see struct Cell
			std::vector<Cell> cells;             // the cells proper (start order), then the second owners' records
			std::vector<CellX> cellx;
			uint32_t n_main_cells = 0;
			if (fast) {
				struct SegRef { int64_t sx, sy; uint32_t ev, k, cls_off; };
				std::vector<SegRef> segs_all;
				std::vector<int64_t> bps;
				std::vector<int64_t> first_bases;
				auto cls_of = [&](const Event &e, uint32_t m) -> uint32_t {          // contiguous-run rule, common/read.h:44-79
					uint32_t hi_b = 31u - (uint32_t)__builtin_clz(m), lo_b = (uint32_t)__builtin_ctz(m);
					uint32_t span = ((2u << hi_b) - 1u) & ~((1u << lo_b) - 1u), cls = 0;
					for (int q = 0; q < e.K; ++q) { uint32_t iso = (uint32_t)e.iso_mask[q] & 0xFu; if ((m & ~iso) == 0 && (iso & span) == m) cls |= 1u << q; }
					return cls;
				};
				uint32_t co2 = 0;
				for (size_t k = b_begin; k < i; ++k) {
					const Event &e = E.ev[lst[k]];
					for (int sgi = 0; sgi < e.N; ++sgi) {
						segs_all.push_back({e.seg_s[sgi], e.seg_e[sgi], (uint32_t)(k - b_begin), (uint32_t)sgi, co2});
						bps.push_back(e.seg_s[sgi]); bps.push_back(e.seg_e[sgi]);
					}
					first_bases.push_back(e.gene_start);
					bps.push_back(e.gene_start + 1);
					co2 += (1u << e.K) - 1u;
				}
				std::sort(bps.begin(), bps.end());
				bps.erase(std::unique(bps.begin(), bps.end()), bps.end());
				std::sort(first_bases.begin(), first_bases.end());
				std::sort(segs_all.begin(), segs_all.end(), [](const SegRef &x, const SegRef &y) { return x.sx < y.sx; });
				auto slot_of = [&](const SegRef *sr, uint32_t mask) -> uint32_t {
					const Event &e = E.ev[lst[b_begin + sr->ev]];
					uint32_t cls = cls_of(e, mask);
					return cls ? sr->cls_off + cls - 1 : CELL_NONE;
				};
				for (size_t q = 0; q + 1 < bps.size(); ++q) {
					const int64_t x0 = bps[q], x1 = bps[q + 1];
					const bool first_base = std::binary_search(first_bases.begin(), first_bases.end(), x0) && x1 == x0 + 1;   // first base of a span
					const SegRef *own[3]; int n_own = 0;
					for (size_t r = 0; r < segs_all.size() && segs_all[r].sx <= x0; ++r)
						if (segs_all[r].sy >= x1) { if (n_own < 3) own[n_own] = &segs_all[r]; ++n_own; }
					Cell c; CellX x;
					c.lo = (int32_t)x0; c.hi = (int32_t)x1;
					x.flags = 0; x.ev = 0;
					if (first_base) {
						// The first base of a span is no cell: a read that starts there is a candidate of the event only if
						// it is not ordered before (gene_start, gene_end, strand, name) in the read index (count/count.cpp:64-85,
						// 429-432).  But where that base is covered by this one segment only -- one event starts here, no other
						// event has a segment here -- such a read can match no other event either (its first block starts in
						// none of their segments), so a read that ends before gene_end counts for nobody.  A start cell says so:
						// one base wide, "runs into the next stretch" up to gene_end - 1, both slots empty; a read that reaches
						// gene_end or beyond is parked with the event as its one candidate.
						const auto fb = std::equal_range(first_bases.begin(), first_bases.end(), x0);
						if (n_own != 1 || fb.second - fb.first != 1 || own[0]->sx != x0) continue;
						const Event &e0 = E.ev[lst[b_begin + own[0]->ev]];
						if (e0.gene_start != x0 || e0.gene_end - 1 < x1) continue;
						c.e1 = (int32_t)x1; c.e2 = (int32_t)(e0.gene_end - 1);
						x.slots = CELL_NONE | (CELL_NONE << 16);
						x.info = (own[0]->ev << 8) | (CELL_K_START << 2);
						x.ev = own[0]->ev;
					} else if (n_own == 0) {
						// inside no segment of the bucket: a read that starts here matches nothing
						c.e1 = c.e2 = CELL_NO_END;
						x.slots = CELL_NONE | (CELL_NONE << 16); x.info = CELL_INFO_EMPTY;
					} else if (n_own == 1) {
						const SegRef *sr = own[0];
						const Event &e = E.ev[lst[b_begin + sr->ev]];
						const int k = (int)sr->k;
						c.e1 = (int32_t)sr->sy; c.e2 = c.e1;
						uint32_t s2 = CELL_NONE;
						if (k + 1 < e.N && e.seg_s[k + 1] == e.seg_e[k]) { c.e2 = (int32_t)e.seg_e[k + 1]; s2 = slot_of(sr, (1u << k) | (1u << (k + 1))); }
						x.slots = slot_of(sr, 1u << k) | (s2 << 16);
						x.info = (sr->ev << 8) | ((uint32_t)k << 2) | (sr->sx == x0 ? 1u : 0u);
						x.ev = sr->ev;
					} else if (n_own == 2) {
						const SegRef *sh = own[0]->sy <= own[1]->sy ? own[0] : own[1], *lg = sh == own[0] ? own[1] : own[0];      // the segment that ends first, the other
						c.e1 = (int32_t)sh->sy; c.e2 = (int32_t)lg->sy;
						x.slots = slot_of(sh, 1u << sh->k) | (slot_of(lg, 1u << lg->k) << 16);
						x.info = CELL_INFO_SHARED; x.flags = CELLX_BOTH | (lg->ev & 0xFFFFu); x.ev = sh->ev;
						{
							const Event &es = E.ev[lst[b_begin + sh->ev]], &el = E.ev[lst[b_begin + lg->ev]];
							const int ks = (int)sh->k, kl = (int)lg->k;
							if (!(ks + 1 < es.N && es.seg_s[ks + 1] == es.seg_e[ks])) x.flags |= CELLX_NEAR_NO_ABUT;
							if (!(kl + 1 < el.N && el.seg_s[kl + 1] == el.seg_e[kl])) x.flags |= CELLX_FAR_NO_ABUT;
						}
					} else continue;
					cells.push_back(c); cellx.push_back(x);
				}
				if (cells.size() > 65000) { cells.clear(); cellx.clear(); }
				n_main_cells = (uint32_t)cells.size();
			}
			if (fast) {
				d.ev_off = off; off = align16(off + (uint32_t)sizeof(FastRec) * d.n_events);
				d.seg_off = off; off = align16(off + (uint32_t)sizeof(Cell) * (uint32_t)cells.size());   // Cell records, then the CellX ones
				off = align16(off + (uint32_t)sizeof(CellX) * (uint32_t)cells.size());
				d.iso_off = n_main_cells | ((uint32_t)cells.size() << 16);                               // cells proper | all records
			} else {
				d.ev_off = off; off = align16(off + 16 * d.n_events);
				d.seg_off = off; off = align16(off + 8 * nseg);
				d.iso_off = off; off = align16(off + 4 * niso);
			}
			d.img_bytes = off;
			d.hist_off = off; off += 8 * HIST_REPLICAS * hist_stride(ncls);
			d.n_cls = ncls;
			if (off > 128u * 1024u) return fail(LSQ_E_UNSUPPORTED, "bucket of %u events on %s needs %u bytes of LDS tables", d.n_events, E.chroms.names[c].c_str(), off);
			E.max_lds_bytes = std::max(E.max_lds_bytes, off);
			d.img_off = (uint32_t)E.images.size();
			d.cls_base = E.n_cls_total;
			d.ev_base = (uint32_t)E.dev2out.size();
			E.images.resize(E.images.size() + d.img_bytes, 0);
			uint8_t *img = E.images.data() + d.img_off;
			uint16_t *bins = reinterpret_cast<uint16_t *>(img + bins_off);
			uint32_t *bins32 = reinterpret_cast<uint32_t *>(img + bins_off);
			if (fast && !cells.empty()) {
				memcpy(img + d.seg_off, cells.data(), cells.size() * sizeof(Cell));
				memcpy(img + d.seg_off + cells.size() * sizeof(Cell), cellx.data(), cellx.size() * sizeof(CellX));
			}
			EventRec *recs = reinterpret_cast<EventRec *>(img + d.ev_off);
			FastRec *frecs = reinterpret_cast<FastRec *>(img + d.ev_off);
			int32_t *segs = reinterpret_cast<int32_t *>(img + d.seg_off);
			uint32_t *isos = reinterpret_cast<uint32_t *>(img + d.iso_off);
			uint32_t so = 0, io = 0, co = 0;
			std::vector<int32_t> ends(d.n_events);
			for (size_t k = b_begin; k < i; ++k) {
				const Event &e = E.ev[lst[k]];
				ends[k - b_begin] = (int32_t)e.gene_end;
				if (fast) {
					FastRec &r = frecs[k - b_begin];
					r.ge = (int32_t)e.gene_end;
					r.meta = co | ((uint32_t)e.N << FAST_NSEG_SHIFT);
					for (int sgi = 0; sgi + 1 < e.N; ++sgi) if (e.seg_s[sgi + 1] == e.seg_e[sgi]) r.meta |= 1u << (FAST_ABUT_SHIFT + sgi);
					if (k + 1 < i && E.ev[lst[k + 1]].gene_start <= e.gene_end) r.meta |= FAST_FLAG_OVERLAPS_NEXT;
					// class of every segment mask: isoform j is compatible iff the mask is a contiguous run
					// of its segment list (common/read.h:44-79)
					uint64_t tbl = 0;
					for (uint32_t m = 1; m < 16; ++m) {
						uint32_t hi_b = 31u - (uint32_t)__builtin_clz(m), lo_b = (uint32_t)__builtin_ctz(m);
						uint32_t span = ((2u << hi_b) - 1u) & ~((1u << lo_b) - 1u);
						uint64_t cls = 0;
						for (int q = 0; q < e.K; ++q) {
							uint32_t iso = (uint32_t)e.iso_mask[q] & 0xFu;
							if ((m & ~iso) == 0 && (iso & span) == m) cls |= 1ull << q;
						}
						tbl |= cls << (4 * m);
					}
					r.tbl_lo = (uint32_t)tbl; r.tbl_hi = (uint32_t)(tbl >> 32);
					for (int sgi = 0; sgi < 4; ++sgi) {
						r.seg[2 * sgi] = sgi < e.N ? (int32_t)e.seg_s[sgi] : INT32_MAX;
						r.seg[2 * sgi + 1] = sgi < e.N ? (int32_t)e.seg_e[sgi] : INT32_MAX;
					}
				} else {
					EventRec &r = recs[k - b_begin];
					r.gs = (int32_t)e.gene_start; r.ge = (int32_t)e.gene_end;
					r.seg_off = (uint16_t)so; r.iso_off = (uint16_t)io; r.cls_off = (uint16_t)co;
					r.nseg = (uint8_t)e.N; r.K = (uint8_t)e.K;
					for (int sgi = 0; sgi < e.N; ++sgi) { segs[2 * (so + sgi)] = (int32_t)e.seg_s[sgi]; segs[2 * (so + sgi) + 1] = (int32_t)e.seg_e[sgi]; }
					for (int q = 0; q < e.K; ++q) isos[io + q] = (uint32_t)e.iso_mask[q];
				}
				E.dev2out.push_back(lst[k]);
				E.dev_cls_base.push_back(E.n_cls_total + co);
				E.dev_iso_base.push_back(E.n_iso_total + io);
				E.dev_K.push_back((uint8_t)e.K);
				TieRec t;
				memset(&t, 0, sizeof t);
				t.strand_id = (uint8_t)e.strand_id;
				t.tie_mode = (uint8_t)e.tie_mode;
				t.tail_len = (uint8_t)std::min<size_t>(e.tie_tail.size(), sizeof t.tail);
				memcpy(t.tail, e.tie_tail.data(), t.tail_len);
				E.ties.push_back(t);
				so += (uint32_t)e.N; io += (uint32_t)e.K; co += (1u << e.K) - 1u;
			}
			E.n_cls_total += ncls; E.n_iso_total += niso;
			// bin k -> first event (span-start order) whose span reaches bin k or beyond
			uint32_t first = 0;
			size_t first_cell = 0;
			for (uint32_t k = 0; k < d.n_bins; ++k) {
				int64_t bin_lo = lo + ((int64_t)k << shift);
				// an event is passed over only once its end lies left of a bin start, so it can
				// never reach a later bin: `first` is min{i : ge_i >= bin_lo} for every k
				while (first < d.n_events && ends[first] < bin_lo) ++first;
				if (fast) {
					// packed buckets: 16-byte bin record = first cell that can hold a base >= bin_lo | first event
					// << 16, then the ends of that cell and the two after it: the cell that holds a base p of
					// the bin is found by counting the ends that are <= p
					while (first_cell < n_main_cells && cells[first_cell].hi <= bin_lo) ++first_cell;
					uint32_t *rec = bins32 + 4 * k;
					rec[0] = (uint32_t)first_cell | (first << 16);
					for (size_t q = 0; q < 3; ++q) rec[1 + q] = (uint32_t)(first_cell + q < n_main_cells ? cells[first_cell + q].hi : INT32_MAX);
				} else bins[k] = (uint16_t)first;
			}
			if (E.chrom_first_bucket[c] < 0) E.chrom_first_bucket[c] = (int32_t)E.buckets.size();
			E.cut_lo[c].push_back((int32_t)lo);
			E.jg_base.push_back((uint32_t)E.jg_keys.size());
			if (fast)
				for (uint32_t ci = 0; ci < n_main_cells; ++ci) {
					const CellX &x = cellx[ci];
					const uint32_t k = (x.info >> 2) & 0x3Fu;
					if (x.info >= CELL_INFO_EMPTY || k == CELL_K_START) continue;
					const Event &e = E.ev[lst[b_begin + x.ev]];
					// block 1 ends on the end of the owner's segment k (variant 0) or, where k + 1 starts on that end, on the end of
					// k + 1 (variant 1: the block runs through both); block 2 starts a later segment
					for (int k2 = (int)k + 1; k2 < e.N; ++k2) E.jg_keys.push_back(jg_key(ci, 0u, (int32_t)e.seg_s[k2]));
					if ((int)k + 1 < e.N && e.seg_s[k + 1] == e.seg_e[k])
						for (int k2 = (int)k + 2; k2 < e.N; ++k2) E.jg_keys.push_back(jg_key(ci, 1u, (int32_t)e.seg_s[k2]));
				}
			E.buckets.push_back(d);
		}
	}
	E.jg_base.push_back((uint32_t)E.jg_keys.size());
	return LSQ_OK;
}

} // namespace lsq

namespace lsq {

// count/count.cpp:231-258 and :394-416.  device_plan = false (classify) skips the device
// limits and the bucket/LDS-image plan.
int compile_events(const lsq_annotation *a, int n_methods, const char *const *read_types,
                   const uint64_t *expected_read_lengths, bool device_plan, lsq_events **out) {
	if (!a || !out || n_methods < 0 || n_methods > LSQ_MAX_METHODS) return fail(LSQ_E_ARG, "bad argument (1..%d methods)", LSQ_MAX_METHODS);
	std::unique_ptr<lsq_events> E(new lsq_events);
	E->n_methods = n_methods;
	std::vector<bool> is_short(n_methods);
	for (int m = 0; m < n_methods; ++m) {
		E->read_types.push_back(read_types[m]);
		E->read_lengths.push_back(expected_read_lengths[m]);
		if (E->read_types[m] == "SHORT_READ") is_short[m] = true;
		else if (E->read_types[m] == "MEDIUM_READ") is_short[m] = false;
		else return fail(LSQ_E_FORMAT, "Unknown read type error: %s", read_types[m]);
	}
	E->ev.reserve(a->selected.size());
	for (const Gene &g : a->selected) {
		Event e;
		e.gname = g.name;
		std::vector<Seg> segs;
		IntervalList span;
		if (!g.isos.empty()) { e.chrom = g.isos[0]->chrom; e.strand = g.isos[0]->strand; }   // splicing_graph.h:239-242
		e.chrom_id = E->chroms.intern(e.chrom);
		e.strand_id = E->strands.intern(e.strand);
		for (const IsoRec *r : g.isos) {
			if (r->exonStarts.size() < r->exonCount || r->exonEnds.size() < r->exonCount)
				return fail(LSQ_E_ARG, "isoform %s: exonCount %llu exceeds the listed exons (the reference reads past its vectors here)", r->name.c_str(), (unsigned long long)r->exonCount);
			int cid = E->chroms.intern(r->chrom);
			if ((size_t)cid >= E->covered.size()) E->covered.resize(cid + 1);
			for (uint64_t i = 0; i < r->exonCount; ++i) {
				int64_t s = r->exonStarts[i], t = r->exonEnds[i];
				if (s <= -COORD_LIMIT || s >= COORD_LIMIT || t <= -COORD_LIMIT || t >= COORD_LIMIT)
					return fail(LSQ_E_RANGE, "isoform %s: exon coordinate outside +-2^30", r->name.c_str());
				if (s > t) return fail(LSQ_E_ARG, "isoform %s: exon start > end (the reference asserts)", r->name.c_str());
				exon_insert(segs, s, t);
				E->covered[cid].add(s, t);
				span.add(s, t);
			}
		}
		if (span.size() == 0) return fail(LSQ_E_ARG, "gene %s has no exon (the reference reads an empty vector here)", g.name.c_str());
		e.gene_start = span.s.front();
		e.gene_end = span.e.back();
		e.N = (int)segs.size();
		e.K = (int)g.isos.size();
		// segment masks are 64 bits wide; a gene's compatibility classes are kept as a dense table of 2^K - 1 counters.
		// Genes beyond the kernels' limits (LSQ_MAX_SEGMENTS, LSQ_MAX_ISOFORMS) are not refused: plan_device puts them
		// into host buckets, which lsq_count / lsq_solve evaluate on the host (lsq_replay.hip).
		if (e.N > 64) return fail(LSQ_E_UNSUPPORTED, "gene %s has %d segments (limit 64)", g.name.c_str(), e.N);
		if (device_plan && e.K > LSQ_HOST_MAX_ISOFORMS) return fail(LSQ_E_UNSUPPORTED, "gene %s has %d isoforms (limit %d: the class table of a gene has 2^K - 1 entries)", g.name.c_str(), e.K, LSQ_HOST_MAX_ISOFORMS);
		for (const Seg &s : segs) { e.seg_s.push_back(s.start); e.seg_e.push_back(s.end); }
		// build_isoform_array (splicing_graph.h:318-361): per isoform walk the segments in order;
		// the exon search resumes at the exon that held the previous segment
		e.ars.assign(n_methods, {});
		for (const IsoRec *r : g.isos) {
			uint64_t mask = 0, len = 0;
			uint64_t from = 0;
			std::vector<uint64_t> seg_len;
			for (int n = 0; n < e.N; ++n) {
				for (uint64_t i = from; i < r->exonCount; ++i) {
					if (r->exonStarts[i] <= e.seg_s[n] && e.seg_e[n] <= r->exonEnds[i]) {
						mask |= 1ull << n;
						uint64_t l = (uint64_t)(e.seg_e[n] - e.seg_s[n]);
						len += l;
						seg_len.push_back(l);
						from = i;
						break;
					}
				}
			}
			if (seg_len.empty()) return fail(LSQ_E_ARG, "isoform %s holds no segment (the reference asserts)", r->name.c_str());
			e.iso_names.push_back(r->name);
			e.iso_mask.push_back(mask);
			e.iso_len.push_back(len);
			for (int m = 0; m < n_methods; ++m) e.ars[m].push_back(ars_total(seg_len, E->read_lengths[m], is_short[m]));
		}
		if (e.K <= LSQ_MAX_ISOFORMS) {
			const size_t n_cls = ((size_t)1 << e.K) - 1;
			e.fim_starts.assign(n_methods, std::vector<uint32_t>((size_t)e.K * n_cls, 0u));
			for (int m = 0; m < n_methods; ++m)
				for (int k = 0; k < e.K; ++k) fim_start_classes(e, k, E->read_lengths[m], is_short[m], e.fim_starts[m].data() + (size_t)k * n_cls);
		}
		// "read-<n>" < gname ? (count/count.cpp:71, std::string operator<)
		static const char pfx[] = "read-";
		size_t p = 0;
		while (p < 5 && p < e.gname.size() && e.gname[p] == pfx[p]) ++p;
		if (p == 5) { e.tie_mode = 2; e.tie_tail = e.gname.substr(5); }
		else if (p == e.gname.size()) e.tie_mode = 0;                    // gname is a proper prefix of "read-"
		else e.tie_mode = ((unsigned char)pfx[p] < (unsigned char)e.gname[p]) ? 1 : 0;
		E->ev.push_back(std::move(e));
	}
	if (E->covered.size() < E->chroms.names.size()) E->covered.resize(E->chroms.names.size());
	if (E->chroms.names.size() > 65000) return fail(LSQ_E_RANGE, "too many chromosomes");
	if (E->strands.names.size() > 255) return fail(LSQ_E_RANGE, "more than 255 distinct strand strings");
	if (!device_plan) { *out = E.release(); return LSQ_OK; }
	E->class_off.assign(E->ev.size() + 1, 0);
	E->iso_off.assign(E->ev.size() + 1, 0);
	for (size_t i = 0; i < E->ev.size(); ++i) {
		E->class_off[i + 1] = E->class_off[i] + (E->ev[i].K < 32 ? ((1ull << E->ev[i].K) - 1) : 0);
		E->iso_off[i + 1] = E->iso_off[i] + (uint64_t)E->ev[i].K;
	}
	int rc = plan_device(*E);
	if (rc) return rc;
	*out = E.release();
	return LSQ_OK;
}

} // namespace lsq

extern "C" {

int lsq_events_compile(const lsq_annotation *a, int n_methods, const char *const *read_types,
                       const uint64_t *expected_read_lengths, lsq_events **out) LSQ_API_TRY {
	return lsq::compile_events(a, n_methods, read_types, expected_read_lengths, true, out);
} LSQ_API_CATCH
void lsq_events_free(lsq_events *e) { delete e; }

int64_t lsq_events_count(const lsq_events *e) { return e ? (int64_t)e->ev.size() : 0; }
int64_t lsq_events_total_isoforms(const lsq_events *e) { return e ? (int64_t)e->iso_off.back() : 0; }
#define EV_OR(ret) if (!e || ev < 0 || (size_t)ev >= e->ev.size()) return ret
const char *lsq_events_gene_name(const lsq_events *e, int64_t ev) { EV_OR(nullptr); return e->ev[ev].gname.c_str(); }
const char *lsq_events_chrom(const lsq_events *e, int64_t ev) { EV_OR(nullptr); return e->ev[ev].chrom.c_str(); }
const char *lsq_events_strand(const lsq_events *e, int64_t ev) { EV_OR(nullptr); return e->ev[ev].strand.c_str(); }
int lsq_events_num_isoforms(const lsq_events *e, int64_t ev) { EV_OR(-1); return e->ev[ev].K; }
int lsq_events_num_segments(const lsq_events *e, int64_t ev) { EV_OR(-1); return e->ev[ev].N; }
const char *lsq_events_isoform_name(const lsq_events *e, int64_t ev, int iso) {
	EV_OR(nullptr);
	if (iso < 0 || iso >= e->ev[ev].K) return nullptr;
	return e->ev[ev].iso_names[iso].c_str();
}
int lsq_events_segment(const lsq_events *e, int64_t ev, int n, int64_t *start, int64_t *end) LSQ_API_TRY {
	EV_OR(LSQ_E_ARG);
	if (n < 0 || n >= e->ev[ev].N) return LSQ_E_ARG;
	*start = e->ev[ev].seg_s[n]; *end = e->ev[ev].seg_e[n];
	return LSQ_OK;
} LSQ_API_CATCH
uint64_t lsq_events_isoform_mask(const lsq_events *e, int64_t ev, int iso) { EV_OR(0); return (iso < 0 || iso >= e->ev[ev].K) ? 0 : e->ev[ev].iso_mask[iso]; }
uint64_t lsq_events_isoform_length(const lsq_events *e, int64_t ev, int iso) { EV_OR(0); return (iso < 0 || iso >= e->ev[ev].K) ? 0 : e->ev[ev].iso_len[iso]; }
uint64_t lsq_events_ars(const lsq_events *e, int method, int64_t ev, int iso) {
	EV_OR(0);
	if (method < 0 || method >= e->n_methods || iso < 0 || iso >= e->ev[ev].K) return 0;
	return e->ev[ev].ars[method][iso];
}
int lsq_events_span(const lsq_events *e, int64_t ev, int64_t *gs, int64_t *ge) { EV_OR(LSQ_E_ARG); *gs = e->ev[ev].gene_start; *ge = e->ev[ev].gene_end; return LSQ_OK; }
int64_t lsq_events_num_buckets(const lsq_events *e) { return e ? (int64_t)e->buckets.size() : 0; }
int64_t lsq_events_lds_table_bytes(const lsq_events *e) { return e ? (int64_t)e->max_lds_bytes : 0; }
int64_t lsq_events_host_genes(const lsq_events *e) {
	int64_t n = 0;
	if (e) for (const lsq::BucketDesc &b : e->buckets) if (b.kind == 2) n += b.n_events;
	return n;
}

int lsq_events_set_shard(lsq_events *e, uint64_t first_event, uint64_t n_events) LSQ_API_TRY {
	if (!e) return fail(LSQ_E_ARG, "null argument");
	if (first_event > e->ev.size()) return fail(LSQ_E_ARG, "shard starts past the last event");
	e->shard_first = first_event;
	e->shard_count = n_events;
	return plan_device(*e);
} LSQ_API_CATCH

// Contiguous slices of the output-ordered events for `world` processes, balanced by weight (reads per event from a
// pre-pass; NULL = by event count): slice r ends where the running weight first reaches r/world of the total.
int lsq_shard_bounds(const lsq_events *e, int world, const double *weights, uint64_t *first, uint64_t *count) LSQ_API_TRY {
	if (!e || world < 1 || !first || !count) return fail(LSQ_E_ARG, "bad argument");
	const size_t n = e->ev.size();
	std::vector<double> cum(n + 1, 0.0);
	for (size_t i = 0; i < n; ++i) {
		const double w = weights ? weights[i] : 1.0;
		if (!(w >= 0.0)) return fail(LSQ_E_ARG, "negative or non-finite weight for event %zu", i);
		cum[i + 1] = cum[i] + w;
	}
	const double total = cum[n];
	std::vector<uint64_t> cuts((size_t)world + 1, 0);
	for (int r = 1; r < world; ++r) {
		const double target = total * (double)r / (double)world;
		size_t c = (size_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
		c = std::min(c, n);
		cuts[(size_t)r] = std::max<uint64_t>(c, cuts[(size_t)r - 1]);
	}
	cuts[(size_t)world] = n;
	for (int r = 0; r < world; ++r) { first[r] = cuts[(size_t)r]; count[r] = cuts[(size_t)r + 1] - cuts[(size_t)r]; }
	return LSQ_OK;
} LSQ_API_CATCH

// words (8 bytes) of the packed per-event records of events [first, first + count): class counts and matched bases
// per read file, theta, log-likelihood -- lsq_results_pack_device's block
uint64_t lsq_record_words(const lsq_events *e, uint64_t first, uint64_t count) {
	if (!e || first > e->ev.size()) return 0;
	const size_t a = (size_t)first, b = (size_t)std::min<uint64_t>(first + count, e->ev.size());
	const uint64_t C = e->class_off[b] - e->class_off[a], I = e->iso_off[b] - e->iso_off[a];
	return 2 * (uint64_t)e->n_methods * C + I + (uint64_t)(b - a);
}

// The gathered blocks of `world` processes (block r at blocks + r * stride_words, holding the records of events
// [first[r], first[r] + count[r]) in output order) put together as the whole job's tables, in the layout
// lsq_results_counts / lsq_results_solve use.
int lsq_gathered_unpack(const lsq_events *e, int world, const uint64_t *first, const uint64_t *count, const uint64_t *blocks, uint64_t stride_words,
                        uint64_t *class_count, uint64_t *class_bases, double *theta, double *logll) LSQ_API_TRY {
	if (!e || world < 1 || !first || !count || !blocks || !class_count) return fail(LSQ_E_ARG, "null argument");
	const size_t n = e->ev.size(), M = (size_t)e->n_methods, n_out = (size_t)e->class_off[n];
	memset(class_count, 0, M * n_out * sizeof(uint64_t));
	if (class_bases) memset(class_bases, 0, M * n_out * sizeof(uint64_t));
	if (theta) memset(theta, 0, (size_t)e->iso_off[n] * sizeof(double));
	if (logll) memset(logll, 0, n * sizeof(double));
	for (int r = 0; r < world; ++r) {
		if (first[r] > n || count[r] > n - first[r]) return fail(LSQ_E_ARG, "slice %d lies outside the events", r);
		const size_t a = (size_t)first[r], b = a + (size_t)count[r];
		const size_t C = (size_t)(e->class_off[b] - e->class_off[a]), I = (size_t)(e->iso_off[b] - e->iso_off[a]);
		if (2 * M * C + I + (b - a) > stride_words) return fail(LSQ_E_ARG, "block %d is longer than the stride", r);
		const uint64_t *blk = blocks + (size_t)r * stride_words;
		for (size_t m = 0; m < M; ++m) {
			memcpy(class_count + m * n_out + e->class_off[a], blk + m * C, C * sizeof(uint64_t));
			if (class_bases) memcpy(class_bases + m * n_out + e->class_off[a], blk + (M + m) * C, C * sizeof(uint64_t));
		}
		if (theta) memcpy(theta + e->iso_off[a], blk + 2 * M * C, I * sizeof(double));
		if (logll) memcpy(logll + a, blk + 2 * M * C + I, (b - a) * sizeof(double));
	}
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_events_chrom_id(lsq_events *e, const char *chrom) LSQ_API_TRY {
	if (!e || !chrom) return LSQ_E_ARG;
	int id = e->chroms.intern(chrom);
	if (id > 65000) return fail(LSQ_E_RANGE, "too many chromosome names");
	return id;
} LSQ_API_CATCH
const char *lsq_events_strand_name(const lsq_events *e, int id) {
	if (!e || id < 0 || (size_t)id >= e->strands.names.size()) return nullptr;
	return e->strands.names[id].c_str();
}
int lsq_events_strand_id(lsq_events *e, const char *strand) LSQ_API_TRY {
	if (!e || !strand) return LSQ_E_ARG;
	int id = e->strands.intern(strand);
	if (id > 255) return fail(LSQ_E_RANGE, "more than 256 distinct strand strings");
	return id;
} LSQ_API_CATCH

} // extern "C"
