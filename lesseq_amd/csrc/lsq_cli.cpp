// Host side of the path, part 3: output rows (count/count.cpp:486-492, solve/solve.cpp:808-847)
// and the three executables' argv handling, exit codes and stderr log.
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <cstdarg>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <cstring>
#include <ctime>
#include <stdexcept>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "lsq_internal.hpp"

using namespace lsq;

namespace lsq { int compile_events(const lsq_annotation *a, int n_methods, const char *const *read_types,
                                   const uint64_t *lens, bool device_plan, lsq_events **out); }

namespace {

// C++ `ostream << double` with default flags prints like printf("%g")
void put_g(std::string &o, double v) {
	char t[64];
	int n = snprintf(t, sizeof t, "%g", v);
	o.append(t, (size_t)n);
}

char *dup_text(const std::string &s) {
	char *p = (char *)malloc(s.size() + 1);
	if (p) { memcpy(p, s.data(), s.size()); p[s.size()] = 0; }
	return p;
}

// jsc/util/log.hpp:22-79: "[LOG YYYY-MM-DD hh:mm:ss LEVEL] text" on stderr when level <= reporting level
int g_log_level = 2;
void logf(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void logf(int level, const char *fmt, ...) {
	if (level > g_log_level) return;
	static const char *names[] = {"ERROR", "WARNING", "INFO", "DEBUG"};
	time_t raw; time(&raw);
	struct tm tmv; localtime_r(&raw, &tmv);
	char msg[2048];
	va_list ap; va_start(ap, fmt); vsnprintf(msg, sizeof msg, fmt, ap); va_end(ap);
	fprintf(stderr, "[LOG %d-%02d-%02d %02d:%02d:%02d %s] %s\n", tmv.tm_year + 1900, tmv.tm_mon + 1, tmv.tm_mday,
	        tmv.tm_hour, tmv.tm_min, tmv.tm_sec, names[level < 3 ? level : 3], msg);
	fflush(stderr);
}

bool cast_long(const char *s, long &out) {
	if (!*s) return false;
	size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
	if (!s[i]) return false;
	for (size_t j = i; s[j]; ++j) if (s[j] < '0' || s[j] > '9') return false;
	errno = 0; char *e; long v = strtol(s, &e, 10);
	if (errno == ERANGE || *e) return false;
	out = v; return true;
}
bool cast_ulong(const char *s, unsigned long &out) {
	if (!*s) return false;
	size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
	if (!s[i]) return false;
	for (size_t j = i; s[j]; ++j) if (s[j] < '0' || s[j] > '9') return false;
	errno = 0; char *e; unsigned long v = strtoul(s, &e, 10);
	if (errno == ERANGE || *e) return false;
	out = v; return true;
}
bool cast_double(const char *s, double &out) {
	if (!*s || isspace((unsigned char)*s)) return false;
	char *e; double v = strtod(s, &e);
	if (e == s || *e) return false;
	out = v; return true;
}

const int EXIT_ABORT = 134;   // the reference dies on assert / an uncaught exception

int status_to_exit(int st) {
	switch (st) {
	case LSQ_OK: return 0;
	case LSQ_E_IO: return EXIT_ABORT;          // assert(ifs.is_open()) (count/count.cpp:139,185,278)
	case LSQ_E_FORMAT: case LSQ_E_PARSE: return 1;
	default: return 2;                         // conditions the reference has no defined behaviour for
	}
}

// `count` and `classify` know LH_GENE_TXT and UCSC_GENE2ISOFORM only (count/count.cpp:141,188,
// classify/classify.cpp:91,138); `solve` takes the wider set lsq_annotation_load reads.  The
// reference opens a file (assert) before it looks at the format literal.
int count_formats_only(const char *iso_fmt, const char *iso_path, const char *g2i_fmt, const char *g2i_path) {
	FILE *f = fopen(iso_path, "rb");
	if (!f) return LSQ_OK;                 // lsq_annotation_load reports the unopenable file
	fclose(f);
	if (strcmp(iso_fmt, "LH_GENE_TXT") != 0) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", iso_fmt);
	f = fopen(g2i_path, "rb");
	if (!f) return LSQ_OK;
	fclose(f);
	if (strcmp(g2i_fmt, "UCSC_GENE2ISOFORM") != 0) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", g2i_fmt);
	return LSQ_OK;
}

bool named_read_format(const char *fmt) {       // the read formats only `solve` takes (solve/solve.cpp:413,487,552)
	return strcmp(fmt, "UCSC_GFF") == 0 || strcmp(fmt, "UCSC_BED") == 0 || strcmp(fmt, "WORMBASE_GFF3") == 0;
}

int precheck_reads_file(const char *fmt, const char *path, bool solve) {
	FILE *f = fopen(path, "rb");
	if (!f) return fail(LSQ_E_IO, "cannot open reads file %s", path);
	fclose(f);
	if (strcmp(fmt, "MRF_SINGLE") != 0 && !(solve && named_read_format(fmt))) return fail(LSQ_E_FORMAT, "Unknown file format error: %s", fmt);
	return LSQ_OK;
}

struct Freer {
	lsq_annotation *a = nullptr; lsq_events *e = nullptr; lsq_ctx *c = nullptr;
	std::vector<lsq_reads *> r;
	~Freer() { for (auto *x : r) lsq_reads_free(x); if (c) lsq_ctx_destroy(c); lsq_events_free(e); lsq_annotation_free(a); }
};

int run_classify(int argc, const char *const *argv) {
	if (argc < 10) { logf(0, "Usage:\nclassify\n\tlog_level(0,1,2,...) proj_name out_prefix\n\tisoform_format isoforms_path g2i_format g2i_path gene_begin_idx gene_end_idx"); return 1; }
	long lvl;
	if (!cast_long(argv[1], lvl)) return EXIT_ABORT;
	g_log_level = (int)lvl;
	lsq_set_log_level(g_log_level);
	std::string out_prefix = argv[3];
	unsigned long gb, ge;
	if (!cast_ulong(argv[8], gb) || !cast_ulong(argv[9], ge)) { logf(0, "Lexical_cast error when converting arguments to numeric values"); return 1; }
	Freer F;
	int st = count_formats_only(argv[4], argv[5], argv[6], argv[7]);
	if (!st) st = lsq_annotation_load(argv[4], argv[5], argv[6], argv[7], gb, ge, &F.a);
	if (st) { logf(0, "%s", lsq_last_error()); return status_to_exit(st); }
	logf(2, "Loaded %lld isoforms", (long long)lsq_annotation_num_isoforms_loaded(F.a));
	logf(2, "Loaded %lld genes", (long long)lsq_annotation_num_genes_loaded(F.a));
	st = compile_events(F.a, 0, nullptr, nullptr, false, &F.e);
	if (st) { logf(0, "%s", lsq_last_error()); return status_to_exit(st); }
	size_t written = 0;
	for (const Event &e : F.e->ev) {
		if (e.K < 2) continue;                                       // classify/classify.cpp:159
		std::string path = out_prefix + e.gname + ".matrix";
		FILE *f = fopen(path.c_str(), "w");
		if (!f) { logf(0, "cannot write %s", path.c_str()); return EXIT_ABORT; }
		fprintf(f, "%s\t%s\t", e.chrom.c_str(), e.strand.c_str());
		for (int n = 0; n < e.N; ++n) fprintf(f, "[%lld,%lld)-", (long long)e.seg_s[n], (long long)e.seg_e[n]);
		fputc('\n', f);
		for (int k = 0; k < e.K; ++k) {
			for (int n = 0; n < e.N; ++n) fprintf(f, "%d\t", (int)(e.iso_mask[k] >> n & 1));
			fputc('\n', f);
		}
		fclose(f);
		++written;
	}
	logf(2, "Built isoform structures for the %zu selected gene(s)", written);
	return 0;
}

bool g_exit_after_output = false;       // set by lsq_cli_main
bool g_is_executable = false;           // set by lsq_cli_main: the process is ours alone

// developer aid: LSQ_CLI_TIMING=1 prints the seconds each phase took on stderr
struct PhaseTimer {
	bool on = getenv("LSQ_CLI_TIMING") != nullptr;
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	void mark(const char *what) {
		if (!on) return;
		const auto n = std::chrono::steady_clock::now();
		fprintf(stderr, "[timing] %-28s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
		t = n;
	}
};

// LSQ_OPTIONS="name=value,...": tuning knobs for lsq_ctx_set_option, read once per process
int apply_env_options(lsq_ctx *c) {
	const char *o = getenv("LSQ_OPTIONS");
	if (!o) return LSQ_OK;
	const std::string all(o);
	size_t pos = 0;
	while (pos < all.size()) {
		size_t end = all.find(',', pos);
		if (end == std::string::npos) end = all.size();
		const std::string item = all.substr(pos, end - pos);
		const size_t eq = item.find('=');
		if (eq != std::string::npos) { const int st = lsq_ctx_set_option(c, item.substr(0, eq).c_str(), atof(item.c_str() + eq + 1)); if (st) return st; }
		pos = end + 1;
	}
	return LSQ_OK;
}

// ---- one job over several GPUs (LSQ_GPUS=N): the reference's scale-out -- a process per slice
// gene_begin_idx..gene_end_idx of the sorted gene list, stdout concatenated (count/count.cpp:204-215) -- inside one
// process: a host thread per GPU, slices of equal read weight (the first, unsharded count on GPU 0 is the pre-pass),
// every thread compiles the whole selected range (so the load-time filter is the unsharded one), takes its slice,
// ingests, counts, solves, packs its per-event records on its GPU; an RCCL all-gather (liblesseq_rccl.so, loaded here
// on demand; LSQ_GATHER=host moves the blocks through host memory instead -- for boxes where the "GPUs" are one device)
// puts the blocks together and thread 0 prints the table, byte-identical to the single-GPU run.
struct ShardedJob {
	bool solve; int G; int M;
	const lsq_annotation *ann;
	std::vector<const char *> fmts, types, paths;
	std::vector<uint64_t> lens;
	std::vector<int> devices;
};

struct RcclApi {
	void *handle = nullptr;
	int (*init_all)(int, const int *, double, void **) = nullptr;     // lsq_comm_init_all_for
	void (*destroy)(void *) = nullptr;
	void (*abort_comm)(void *) = nullptr;
	int (*gather)(lsq_ctx *, void *, const void *, void *, uint64_t) = nullptr;
	int (*allreduce_counts)(lsq_ctx *, void *, void *) = nullptr;
	const char *(*last_error)(void) = nullptr;
	std::string why;                       // dlerror() of the failing dlopen (read once: the call clears it)
	bool load() {
		Dl_info info;
		std::string dir;
		if (dladdr((const void *)&lsq_abi_version, &info) && info.dli_fname) { dir = info.dli_fname; const size_t s = dir.rfind('/'); dir = s == std::string::npos ? "" : dir.substr(0, s + 1); }
		handle = dlopen((dir + "liblesseq_rccl.so").c_str(), RTLD_NOW | RTLD_LOCAL);
		if (!handle) { if (const char *e = dlerror()) why = e; handle = dlopen("liblesseq_rccl.so", RTLD_NOW | RTLD_LOCAL); }
		if (!handle) { if (const char *e = dlerror()) { if (!why.empty()) why += "; "; why += e; } return false; }
		init_all = (int (*)(int, const int *, double, void **))dlsym(handle, "lsq_comm_init_all_for");
		destroy = (void (*)(void *))dlsym(handle, "lsq_comm_destroy");
		abort_comm = (void (*)(void *))dlsym(handle, "lsq_comm_abort");
		gather = (int (*)(lsq_ctx *, void *, const void *, void *, uint64_t))dlsym(handle, "lsq_gather");
		allreduce_counts = (int (*)(lsq_ctx *, void *, void *))dlsym(handle, "lsq_allreduce_counts");
		last_error = (const char *(*)(void))dlsym(handle, "lsq_rccl_last_error");
		return init_all && destroy && abort_comm && gather && allreduce_counts && last_error;
	}
};

// Where the host threads of a job's GPUs meet: everybody arrives, and learns whether everybody is well.  A thread that
// failed on the way must keep the others out of the collective that follows (they would spin in it for a peer that never comes).
struct Agreement {
	std::mutex mu; std::condition_variable cv; int arrived = 0, failed = 0; const int G;
	explicit Agreement(int g) : G(g) {}
	bool arrive(bool ok) {
		std::unique_lock<std::mutex> lk(mu);
		++arrived; if (!ok) ++failed;
		cv.notify_all();
		cv.wait(lk, [&] { return arrived == G; });
		return failed == 0;
	}
	void absent() { std::lock_guard<std::mutex> g(mu); ++arrived; ++failed; cv.notify_all(); }      // a slice that never started
};

// The other way to cut one job (LSQ_SHARD=reads; SURVEY 8(e)'s alternative): every GPU takes a slice of the READS -- a byte
// range of each MRF file, cut at line starts, copied and parsed on that GPU only -- against ALL events; the slices' newline
// counts give every slice the file-wide number of its first line (read names "read-<line>" decide span-start ties,
// count/count.cpp:64-85,293-295); the class counts and matched bases are summed over the GPUs (integer sums: any order,
// count.cpp:378,467-482) with one ncclAllReduce (liblesseq_rccl's lsq_allreduce_counts; LSQ_GATHER=host: through host
// memory), and GPU 0 goes on with the sums as a single-GPU run does (EM, rows).  Here the loader scales with the GPUs (the
// event-sharded job parses the whole text on every GPU, twice); MRF_SINGLE files and genes within the kernels' limits only
// -- the caller falls back to the event-sharded job otherwise -- and an event inside the EM guard band cannot be replayed in
// per-read order (no GPU holds all of its reads): it keeps the kernel's numbers and is named at log level 1.
// On return ctx0 holds the whole job's counts (lsq_counts_import_device).
int run_read_sharded_job(const ShardedJob &J, lsq_ctx *ctx0, lsq_events *ev0) {
	const int G = J.G, M = J.M;
	const char *gmode = getenv("LSQ_GATHER");
	const bool via_host = gmode && strcmp(gmode, "host") == 0;
	double time_limit = 120.0;
	if (const char *e = getenv("LSQ_COLLECTIVE_TIMEOUT")) { const double v = atof(e); if (v > 0) time_limit = v; }
	RcclApi rccl;
	std::vector<void *> comms((size_t)G, nullptr);
	if (!via_host) {
		if (!rccl.load()) { logf(0, "LSQ_GPUS=%d needs liblesseq_rccl.so beside the library (%s)", G, rccl.why.empty() ? "symbols missing" : rccl.why.c_str()); return 3; }
		const int st = rccl.init_all(G, J.devices.data(), time_limit, comms.data());
		if (st) { logf(0, "%s", rccl.last_error()); return 3; }
	}
	// the slices: byte cuts moved to line starts (the first slice keeps the header line)
	std::vector<std::vector<uint64_t>> cuts((size_t)M);
	for (int m = 0; m < M; ++m) {
		FILE *f = fopen(J.paths[(size_t)m], "rb");
		if (!f) { logf(0, "cannot open reads file %s", J.paths[(size_t)m]); return EXIT_ABORT; }
		fseeko(f, 0, SEEK_END);
		const uint64_t size = (uint64_t)ftello(f);
		cuts[(size_t)m].assign((size_t)G + 1, size);
		cuts[(size_t)m][0] = 0;
		for (int r = 1; r < G; ++r) {
			uint64_t at = std::max(size / (uint64_t)G * (uint64_t)r, cuts[(size_t)m][(size_t)r - 1]);
			fseeko(f, (off_t)at, SEEK_SET);
			int ch;
			while (at < size && (ch = fgetc(f)) != EOF) { ++at; if (ch == '\n') break; }      // to the first byte after the next newline
			cuts[(size_t)m][(size_t)r] = std::min(at, size);
		}
		fclose(f);
	}
	const uint64_t words = lsq_counts_device_words(ctx0);
	std::vector<std::vector<uint64_t>> lines((size_t)G, std::vector<uint64_t>((size_t)M, 0));     // newlines per slice and file
	std::vector<std::vector<uint64_t>> host_words(via_host ? (size_t)G : 0, std::vector<uint64_t>(std::max<uint64_t>(words, 1), 0));
	std::vector<int> status((size_t)G, LSQ_OK);
	std::vector<std::string> errors((size_t)G);
	std::vector<lsq_ctx *> ctxs((size_t)G, nullptr);
	std::vector<lsq_events *> evs((size_t)G, nullptr);
	std::vector<std::vector<lsq_text *>> texts((size_t)G, std::vector<lsq_text *>((size_t)M, nullptr));
	std::vector<void *> d_words((size_t)G, nullptr);
	ctxs[0] = ctx0; evs[0] = ev0;
	Agreement staged(G), counted(G);
	auto work = [&](int r) {
		auto bad = [&](int s2, const char *msg) { status[(size_t)r] = s2 ? s2 : LSQ_E_STATE; errors[(size_t)r] = msg; return status[(size_t)r]; };
		auto guarded = [&](auto &&body) -> int {
			try { return body(); }
			catch (const std::exception &ex) { return bad(LSQ_E_INTERNAL, ex.what()); }
			catch (...) { return bad(LSQ_E_INTERNAL, "unknown exception"); }
		};
		// 1. context, event tables, this GPU's byte range of every file, its newline counts
		int s = guarded([&]() -> int {
			int q;
			if (r > 0) {
				if ((q = lsq_ctx_create(J.devices[(size_t)r], &ctxs[(size_t)r])) || (q = apply_env_options(ctxs[(size_t)r]))) return bad(q, lsq_last_error());
				if ((q = lsq_events_compile(J.ann, M, J.types.data(), J.lens.data(), &evs[(size_t)r]))) return bad(q, lsq_last_error());
				if ((q = lsq_events_upload(ctxs[(size_t)r], evs[(size_t)r]))) return bad(q, lsq_last_error());
			}
			for (int m = 0; m < M; ++m) {
				if ((q = lsq_text_stage_range(ctxs[(size_t)r], J.paths[(size_t)m], cuts[(size_t)m][(size_t)r], cuts[(size_t)m][(size_t)r + 1], &texts[(size_t)r][(size_t)m]))) return bad(q, lsq_last_error());
				if ((q = lsq_text_lines(ctxs[(size_t)r], texts[(size_t)r][(size_t)m], &lines[(size_t)r][(size_t)m]))) return bad(q, lsq_last_error());
			}
			return LSQ_OK;
		});
		bool everyone = staged.arrive(s == LSQ_OK);
		lsq_ctx *c = ctxs[(size_t)r];
		// 2. parse and ingest the slice under its file-wide line numbers, count, hand the counters over
		s = LSQ_E_STATE;
		if (everyone) s = guarded([&]() -> int {
			int q;
			for (int m = 0; m < M; ++m) {
				uint64_t before = 0;
				for (int p = 0; p < r; ++p) before += lines[(size_t)p][(size_t)m];
				if ((q = lsq_reads_upload_text_at(c, m, J.fmts[(size_t)m], texts[(size_t)r][(size_t)m], r == 0 ? 1 : 0, r == 0 ? 1 : before))) return bad(q, lsq_last_error());
				lsq_text_free(texts[(size_t)r][(size_t)m]); texts[(size_t)r][(size_t)m] = nullptr;
			}
			if ((q = lsq_count(c))) return bad(q, lsq_last_error());
			if ((q = lsq_device_alloc(c, std::max<uint64_t>(words, 1) * 8, &d_words[(size_t)r]))) return bad(q, lsq_last_error());
			if (via_host) {
				if ((q = lsq_counts_export_device(c, d_words[(size_t)r])) || (q = lsq_device_read(c, host_words[(size_t)r].data(), d_words[(size_t)r], words * 8))) return bad(q, lsq_last_error());
			}
			if (const char *fr = getenv("LSQ_FAIL_RANK")) if (atoi(fr) == r) return bad(LSQ_E_STATE, "failure of this slice requested (LSQ_FAIL_RANK)");
			return LSQ_OK;
		});
		everyone = everyone && counted.arrive(s == LSQ_OK);
		// 3. the sum over the GPUs; GPU 0 takes it as its counts
		if (everyone) (void)guarded([&]() -> int {
			int q;
			if (via_host) {
				if (r == 0) {
					for (int p = 1; p < G; ++p) for (uint64_t w = 0; w < words; ++w) host_words[0][(size_t)w] += host_words[(size_t)p][(size_t)w];
					if ((q = lsq_device_write(c, d_words[0], host_words[0].data(), words * 8)) || (q = lsq_counts_import_device(c, d_words[0])) || (q = lsq_ctx_synchronize(c))) return bad(q, lsq_last_error());
				}
				return LSQ_OK;
			}
			if ((q = rccl.allreduce_counts(c, comms[(size_t)r], d_words[(size_t)r]))) return bad(q, rccl.last_error());
			if ((q = lsq_ctx_synchronize_for(c, time_limit))) {
				bad(q, lsq_last_error());
				if (q == LSQ_E_TIMEOUT) { rccl.abort_comm(comms[(size_t)r]); comms[(size_t)r] = nullptr; }       // ends the spinning kernel
				return q;
			}
			if (r == 0 && ((q = lsq_counts_import_device(c, d_words[0])) || (q = lsq_ctx_synchronize(c)))) return bad(q, lsq_last_error());
			return LSQ_OK;
		});
		for (auto *&t : texts[(size_t)r]) { lsq_text_free(t); t = nullptr; }
		if (c) lsq_device_free(c, d_words[(size_t)r]);
		if (everyone && !status[(size_t)r])
			logf(2, "GPU %d: bytes %llu..%llu of %s%s, %llu reads retained of them", J.devices[(size_t)r], (unsigned long long)cuts[0][(size_t)r], (unsigned long long)cuts[0][(size_t)r + 1],
			     J.paths[0], M > 1 ? " (and the like of the other files)" : "", (unsigned long long)lsq_reads_retained(c, 0));
	};
	{
		ThreadGroup th;
		for (int r = 1; r < G; ++r) {
			try { th.spawn([&work, r] { work(r); }); }
			catch (...) { status[(size_t)r] = LSQ_E_INTERNAL; errors[(size_t)r] = "the slice's host thread could not be started"; staged.absent(); }
		}
		work(0);
		th.join();
	}
	uint64_t retained_all = 0;
	for (int r = 0; r < G; ++r) if (ctxs[(size_t)r]) retained_all += lsq_reads_retained(ctxs[(size_t)r], 0);
	for (int r = 1; r < G; ++r) { if (ctxs[(size_t)r]) lsq_ctx_destroy(ctxs[(size_t)r]); lsq_events_free(evs[(size_t)r]); }
	if (!via_host) for (void *cm : comms) if (cm) rccl.destroy(cm);
	// the first failing slice in file order speaks (a field that fails the cast: the reference reports the first such line)
	for (int r = 0; r < G; ++r) if (status[(size_t)r]) {
		if (status[(size_t)r] == LSQ_E_PARSE) { logf(0, "%s", errors[(size_t)r].c_str()); logf(0, "Lexical_cast error when converting arguments to numeric values"); return 1; }
		logf(0, "GPU %d: %s", J.devices[(size_t)r], errors[(size_t)r].c_str());
		return 3;
	}
	logf(2, "Sampling method #0: loaded %llu reads associated with the selected gene regions (over %d GPUs)", (unsigned long long)retained_all, G);
	return 0;
}

// F.c / F.e: GPU 0's context with the whole job counted on it (the pre-pass); texts0: its staged MRF texts (kept)
int run_sharded_job(const ShardedJob &J, lsq_ctx *ctx0, lsq_events *ev0, std::vector<lsq_text *> &texts0, const std::vector<double> &trb, std::string &out) {
	const int G = J.G, M = J.M;
	const int64_t n_ev = lsq_events_count(ev0);
	// pre-pass: reads per event from the unsharded count
	const size_t n_cls = (size_t)lsq_results_num_classes(ctx0);
	std::vector<uint64_t> cnt0(std::max<size_t>((size_t)M * n_cls, 1));
	int st = lsq_results_counts(ctx0, cnt0.data(), nullptr);
	if (st) { logf(0, "%s", lsq_last_error()); return 3; }
	std::vector<uint64_t> coff((size_t)n_ev + 1);
	lsq_results_class_offsets(ctx0, coff.data());
	std::vector<double> weights((size_t)n_ev, 0.0);
	for (int64_t i = 0; i < n_ev; ++i)
		for (int m = 0; m < M; ++m)
			for (uint64_t k = coff[(size_t)i]; k < coff[(size_t)i + 1]; ++k) weights[(size_t)i] += (double)cnt0[(size_t)m * n_cls + k];
	std::vector<uint64_t> first((size_t)G), count((size_t)G);
	st = lsq_shard_bounds(ev0, G, weights.data(), first.data(), count.data());
	if (st) { logf(0, "%s", lsq_last_error()); return 2; }
	uint64_t stride = 1;
	for (int r = 0; r < G; ++r) stride = std::max(stride, lsq_record_words(ev0, first[(size_t)r], count[(size_t)r]));
	const char *gmode = getenv("LSQ_GATHER");
	const bool via_host = gmode && strcmp(gmode, "host") == 0;
	RcclApi rccl;
	std::vector<void *> comms((size_t)G, nullptr);
	// nothing on this path waits without a limit (LSQ_COLLECTIVE_TIMEOUT seconds, default 120): the communicator set-up, and
	// the gather itself -- whose kernel spins until every peer has joined it
	double time_limit = 120.0;
	if (const char *e = getenv("LSQ_COLLECTIVE_TIMEOUT")) { const double v = atof(e); if (v > 0) time_limit = v; }
	if (!via_host) {
		if (!rccl.load()) { logf(0, "LSQ_GPUS=%d needs liblesseq_rccl.so beside the library (%s)", G, rccl.why.empty() ? "symbols missing" : rccl.why.c_str()); return 3; }
		st = rccl.init_all(G, J.devices.data(), time_limit, comms.data());
		if (st) { logf(0, "%s", rccl.last_error()); return 3; }
	}
	std::vector<uint64_t> host_blocks((size_t)G * stride, 0);
	std::vector<int> status((size_t)G, LSQ_OK);
	std::vector<std::string> errors((size_t)G);
	std::vector<lsq_ctx *> ctxs((size_t)G, nullptr);
	std::vector<lsq_events *> evs((size_t)G, nullptr);
	std::vector<void *> d_blocks((size_t)G, nullptr), d_alls((size_t)G, nullptr);
	std::vector<uint32_t> replayed((size_t)G, 0);
	ctxs[0] = ctx0; evs[0] = ev0;
	// Every GPU's thread does its slice up to the packed block, then all of them AGREE before any enters the collective: a
	// thread that failed on the way (no context, a file it cannot parse, out of memory) would otherwise leave the others
	// spinning in ncclAllGather for a peer that never comes.  One failure: nobody gathers, the job exits 3 with the message.
	Agreement agreement(G);
	auto prepare = [&](int r) -> int {
		auto bad = [&](int s2, const char *msg) { status[(size_t)r] = s2 ? s2 : LSQ_E_STATE; errors[(size_t)r] = msg; return status[(size_t)r]; };
		int s;
		if (r > 0) {
			if ((s = lsq_ctx_create(J.devices[(size_t)r], &ctxs[(size_t)r])) || (s = apply_env_options(ctxs[(size_t)r]))) return bad(s, lsq_last_error());
			if ((s = lsq_events_compile(J.ann, M, J.types.data(), J.lens.data(), &evs[(size_t)r]))) return bad(s, lsq_last_error());
		}
		lsq_ctx *c = ctxs[(size_t)r];
		lsq_events *e = evs[(size_t)r];
		if ((s = lsq_events_set_shard(e, first[(size_t)r], count[(size_t)r])) || (s = lsq_events_upload(c, e))) return bad(s, lsq_last_error());
		for (int m = 0; m < M; ++m) {
			if (r == 0 && texts0[(size_t)m]) s = lsq_reads_upload_text(c, m, J.fmts[(size_t)m], texts0[(size_t)m]);
			else if (strcmp(J.fmts[(size_t)m], "MRF_SINGLE") == 0 && (r > 0 || !texts0[(size_t)m])) s = lsq_reads_upload_mrf(c, m, J.fmts[(size_t)m], J.paths[(size_t)m]);
			else s = LSQ_E_UNSUPPORTED;
			if (s == LSQ_E_UNSUPPORTED) {         // name-keyed formats, long strand strings: the host parser
				lsq_reads *rd = nullptr;
				s = lsq_reads_parse(J.fmts[(size_t)m], J.paths[(size_t)m], e, 0, &rd);
				if (!s) { s = lsq_reads_upload(c, m, rd); lsq_reads_free(rd); }
			}
			if (s) return bad(s, lsq_last_error());
		}
		if ((s = lsq_count(c)) || (s = lsq_solve(c)) || (s = lsq_solve_finalize(c, &replayed[(size_t)r]))) return bad(s, lsq_last_error());
		if ((s = lsq_device_alloc(c, stride * 8, &d_blocks[(size_t)r]))) return bad(s, lsq_last_error());
		if (!via_host && (s = lsq_device_alloc(c, (uint64_t)G * stride * 8, &d_alls[(size_t)r]))) return bad(s, lsq_last_error());
		if ((s = lsq_results_pack_device(c, d_blocks[(size_t)r]))) return bad(s, lsq_last_error());
		// (the developer's way to see the agreement at work: LSQ_FAIL_RANK=r makes slice r fail here)
		if (const char *fr = getenv("LSQ_FAIL_RANK")) if (atoi(fr) == r) return bad(LSQ_E_STATE, "failure of this slice requested (LSQ_FAIL_RANK)");
		return LSQ_OK;
	};
	auto work = [&](int r) {
		int s = LSQ_E_INTERNAL;
		try { s = prepare(r); }
		catch (const std::exception &ex) { status[(size_t)r] = LSQ_E_INTERNAL; errors[(size_t)r] = ex.what(); }
		catch (...) { status[(size_t)r] = LSQ_E_INTERNAL; errors[(size_t)r] = "unknown exception"; }
		const bool everyone = agreement.arrive(s == LSQ_OK);
		lsq_ctx *c = ctxs[(size_t)r];
		if (everyone) {
			auto bad = [&](int s2, const char *msg) { status[(size_t)r] = s2 ? s2 : LSQ_E_STATE; errors[(size_t)r] = msg; };
			if (via_host) {
				s = lsq_device_read(c, host_blocks.data() + (size_t)r * stride, d_blocks[(size_t)r], stride * 8);
				if (s) bad(s, lsq_last_error());
			} else {
				s = rccl.gather(c, comms[(size_t)r], d_blocks[(size_t)r], d_alls[(size_t)r], stride);
				if (s) bad(s, rccl.last_error());
				else if ((s = lsq_ctx_synchronize_for(c, time_limit))) {
					bad(s, lsq_last_error());
					if (s == LSQ_E_TIMEOUT) { rccl.abort_comm(comms[(size_t)r]); comms[(size_t)r] = nullptr; }       // ends the spinning kernel
				}
				else if (r == 0) { s = lsq_device_read(c, host_blocks.data(), d_alls[0], (uint64_t)G * stride * 8); if (s) bad(s, lsq_last_error()); }
			}
		}
		if (c) { lsq_device_free(c, d_alls[(size_t)r]); lsq_device_free(c, d_blocks[(size_t)r]); }
		if (everyone && !status[(size_t)r]) {
			unsigned long long pooled = 0;
			for (int m = 0; m < M; ++m) pooled += lsq_reads_pooled(c, m);
			logf(2, "GPU %d: events %llu..%llu of the sorted list, %llu reads pooled for them%s", J.devices[(size_t)r], (unsigned long long)first[(size_t)r],
			     (unsigned long long)(first[(size_t)r] + count[(size_t)r]), pooled, replayed[(size_t)r] ? " (guard-band events solved again in per-read order)" : "");
		}
	};
	{
		ThreadGroup th;          // joined on every way out
		// (a thread that cannot even be started counts as a slice that failed: the others must not wait for it)
		for (int r = 1; r < G; ++r) {
			try { th.spawn([&work, r] { work(r); }); }
			catch (...) { status[(size_t)r] = LSQ_E_INTERNAL; errors[(size_t)r] = "the slice's host thread could not be started"; agreement.absent(); }
		}
		work(0);
		th.join();
	}
	for (int r = 1; r < G; ++r) { if (ctxs[(size_t)r]) lsq_ctx_destroy(ctxs[(size_t)r]); lsq_events_free(evs[(size_t)r]); }
	if (!via_host) for (void *cm : comms) if (cm) rccl.destroy(cm);
	for (int r = 0; r < G; ++r) if (status[(size_t)r]) { logf(0, "GPU %d: %s", J.devices[(size_t)r], errors[(size_t)r].c_str()); return 3; }
	// the whole job's tables, in output order
	const size_t n_iso = (size_t)lsq_events_total_isoforms(ev0);
	std::vector<uint64_t> cnt(std::max<size_t>((size_t)M * n_cls, 1)), bases(cnt.size());
	std::vector<double> theta(std::max<size_t>(n_iso, 1)), ll(std::max<size_t>((size_t)n_ev, 1));
	st = lsq_gathered_unpack(ev0, G, first.data(), count.data(), host_blocks.data(), stride, cnt.data(), bases.data(), theta.data(), ll.data());
	if (st) { logf(0, "%s", lsq_last_error()); return 2; }
	char *text = nullptr;
	st = J.solve ? lsq_format_solve(ev0, M, cnt.data(), bases.data(), theta.data(), ll.data(), trb.data(), &text) : lsq_format_count(ev0, M, cnt.data(), &text);
	if (st || !text) { logf(0, "%s", lsq_last_error()); return 2; }
	out.assign(text);
	free(text);
	logf(2, "Processed %lld genes on %d GPUs... Done", (long long)n_ev, G);
	return 0;
}

int run_count_solve(bool solve, int argc, const char *const *argv, std::string &out) {
	PhaseTimer T;
	const int per = solve ? 5 : 4;
	auto usage = [&] {
		logf(0, "Usage:\n%s\n\tlog_level(0,1,2,...) proj_name out_prefix\n\tisoform_format isoforms_path g2i_format g2i_path gene_begin_idx gene_end_idx\n  (read_format read_type expected_read_length reads_path%s)+",
		     solve ? "solve" : "count", solve ? " total_read_bases" : "");
		return 1;
	};
	// Extension (not in the reference, whose fim.h is dead code): `solve ... --fim` after the last read file
	// appends, behind the table, one `#fim` line per event and read file with fim.h's two variance estimates and the
	// expected Fisher information matrix (lsq_fim; parity unpinned).  The table itself does not change.
	bool want_fim = false;
	if (solve && argc > 15 && strcmp(argv[argc - 1], "--fim") == 0) { want_fim = true; --argc; }
	if (argc < (solve ? 15 : 14)) return usage();
	long lvl;
	if (!cast_long(argv[1], lvl)) return EXIT_ABORT;      // lexical_cast outside the try block (count/count.cpp:99)
	g_log_level = (int)lvl;
	lsq_set_log_level(g_log_level);
	unsigned long gb, ge;
	if (!cast_ulong(argv[8], gb) || !cast_ulong(argv[9], ge)) { logf(0, "Lexical_cast error when converting arguments to numeric values"); return 1; }
	std::vector<const char *> fmts, types, paths;
	std::vector<uint64_t> lens;
	std::vector<double> trb;
	for (int i = 10; i < argc;) {
		if (argc - i < per) return usage();
		fmts.push_back(argv[i++]);
		types.push_back(argv[i++]);
		unsigned long L;
		if (!cast_ulong(argv[i++], L)) { logf(0, "Lexical_cast error when converting arguments to numeric values"); return 1; }
		lens.push_back(L);
		paths.push_back(argv[i++]);
		if (solve) { double v; if (!cast_double(argv[i++], v)) { logf(0, "Lexical_cast error when converting arguments to numeric values"); return 1; } trb.push_back(v); }
	}
	const int M = (int)paths.size();
	if (M > LSQ_MAX_METHODS) { logf(0, "more than %d read files", LSQ_MAX_METHODS); return 2; }
	// LSQ_DEVICE picks the GPU; LSQ_GPUS=N runs the job over N of them (LSQ_DEVICES="a,b,..." names them, default 0..N-1)
	int G = 1;
	if (const char *e = getenv("LSQ_GPUS")) G = std::max(1, atoi(e));
	if (want_fim) G = 1;
	std::vector<int> devices;
	if (const char *e = getenv("LSQ_DEVICES")) { const char *q = e; while (*q) { devices.push_back(atoi(q)); q = strchr(q, ','); if (!q) break; ++q; } }
	if (G > 1 && devices.size() < (size_t)G) { devices.clear(); for (int r = 0; r < G; ++r) devices.push_back(r); }
	if (G == 1) { int dev = 0; if (const char *e = getenv("LSQ_DEVICE")) dev = atoi(e); devices.assign(1, dev); }
	devices.resize((size_t)G);
	// LSQ_SHARD=reads: the job's GPUs share the READS instead of the events (run_read_sharded_job): MRF_SINGLE files only
	bool shard_reads = false;
	if (const char *e = getenv("LSQ_SHARD")) shard_reads = strcmp(e, "reads") == 0 && G > 1 && !want_fim;
	for (const char *f : fmts) if (strcmp(f, "MRF_SINGLE") != 0) shard_reads = false;
	Freer F;
	// the device context (HIP start-up, a tenth of a second or more) is created on a second thread while
	// this one reads the annotation; its status is looked at only where the reference would be past
	// every check it makes before the reads
	int ctx_status = LSQ_OK;
	std::string ctx_error;
	lsq_ctx *ctx_bg = nullptr;
	// ... and the same thread goes on to copy the MRF text of every read file to HBM (that needs no event
	// table); a file that does not open or is not MRF_SINGLE is left to the main thread, which reports it
	std::vector<lsq_text *> texts((size_t)M, nullptr);
	std::atomic<int> ctx_ready{0};          // 1: the context exists (the thread goes on to stage the reads' text), 2: the thread has given up
	auto make_context = [&] {
		ctx_status = lsq_ctx_create_with(devices[0], g_is_executable ? LSQ_CTX_LANES_IN_BACKGROUND : 0u, &ctx_bg);
		if (ctx_status) { ctx_error = lsq_last_error(); ctx_ready.store(2, std::memory_order_release); return; }          // the message lives in that thread
		ctx_status = apply_env_options(ctx_bg);
		if (ctx_status) { ctx_error = lsq_last_error(); ctx_ready.store(2, std::memory_order_release); return; }
		ctx_ready.store(1, std::memory_order_release);
		if (shard_reads) return;          // every GPU stages its own byte range later
		for (int m = 0; m < M; ++m)
			if (strcmp(fmts[m], "MRF_SINGLE") == 0 && lsq_text_stage(ctx_bg, paths[m], &texts[(size_t)m]) != LSQ_OK) texts[(size_t)m] = nullptr;
	};
	// What the second thread made is released by this guard, which is declared BEFORE the thread group: on every way out
	// of this function the group is destroyed first (it joins the thread, so nothing is being written any more), then the
	// guard frees the staged texts and -- unless F owns it by then -- the context.
	struct BackgroundProducts {
		lsq_ctx *&c; lsq_ctx *&owner; std::vector<lsq_text *> &tx;
		~BackgroundProducts() { for (auto *x : tx) lsq_text_free(x); if (c && c != owner) lsq_ctx_destroy(c); }
	} products{ctx_bg, F.c, texts};
	ThreadGroup ctx_thread;
	// Only an executable overlaps the two (lsq_cli_main: a process of its own, 0.1-0.3 s of HIP start-up to hide).  Inside a
	// host process (lsq_cli_run) the context is made on the calling thread, in order: the host may hold other contexts and
	// other runtimes' threads, and a call that returns must leave no thread of its own behind.
	if (g_is_executable) ctx_thread.spawn(make_context);
	auto join_context = [&] {
		if (g_is_executable) ctx_thread.join();
		else if (!ctx_bg && ctx_status == LSQ_OK) make_context();
		if (ctx_thread.failed() && ctx_status == LSQ_OK) { ctx_status = LSQ_E_INTERNAL; ctx_error = ctx_thread.error(); }
	};
	logf(2, "Loading isoforms...");
	int st = solve ? LSQ_OK : count_formats_only(argv[4], argv[5], argv[6], argv[7]);
	if (!st) st = lsq_annotation_load(argv[4], argv[5], argv[6], argv[7], gb, ge, &F.a);
	if (st) { logf(0, "%s", lsq_last_error()); return status_to_exit(st); }
	logf(2, "Loaded %lld isoforms", (long long)lsq_annotation_num_isoforms_loaded(F.a));
	logf(2, "Loaded %lld genes", (long long)lsq_annotation_num_genes_loaded(F.a));
	logf(2, "Selected %lld gene(s)", (long long)lsq_annotation_num_genes(F.a));
	T.mark("annotation load");
	// An unknown read type is only noticed inside the per-gene loop of the reference
	// (count/count.cpp:417-419), i.e. after every read file was loaded and only when at least one
	// gene is selected; keep that precedence.
	bool bad_type = false;
	std::string bad_type_name;
	std::vector<const char *> use_types(types);
	for (int m = 0; m < M; ++m)
		if (strcmp(types[m], "SHORT_READ") != 0 && strcmp(types[m], "MEDIUM_READ") != 0) {
			if (!bad_type) bad_type_name = types[m];
			bad_type = true; use_types[m] = "SHORT_READ";
		}
	st = lsq_events_compile(F.a, M, use_types.data(), lens.data(), &F.e);
	if (st) { logf(0, "%s", lsq_last_error()); return status_to_exit(st); }
	const int64_t n_ev = lsq_events_count(F.e);
	logf(2, "Built isoform structures for the %lld selected gene(s)", (long long)n_ev);
	T.mark("event compile + device plan");
	logf(2, "Loading the reads from %d sampling method(s)", M);
	for (int m = 0; m < M; ++m) {
		// what the reference decides before it reads a line: the file opens (assert) and the format literal is known
		st = precheck_reads_file(fmts[m], paths[m], solve);
		if (st) { logf(0, "%s", lsq_last_error()); return status_to_exit(st); }
		if (!F.c) {
			// An executable's second thread is still copying the reads' text to HBM when its context is ready: the event tables go up
			// beside that copy (small uploads on the same stream, between two slices of the text), then this thread waits for the copy.
			if (g_is_executable) {
				while (ctx_ready.load(std::memory_order_acquire) == 0) std::this_thread::yield();
				T.mark("wait for the device context");
				if (ctx_ready.load(std::memory_order_acquire) == 1) {
					st = lsq_events_upload(ctx_bg, F.e);
					if (st) { const std::string msg = lsq_last_error(); join_context(); logf(0, "%s", msg.c_str()); return 3; }
					T.mark("event tables upload");
				}
			}
			join_context();
			T.mark("wait for the text copy");
			if (ctx_status) { logf(0, "%s", ctx_error.c_str()); return 3; }
			F.c = ctx_bg;
			if (!g_is_executable) {
				st = lsq_events_upload(F.c, F.e);
				if (st) { logf(0, "%s", lsq_last_error()); return 3; }
				T.mark("event tables upload");
			}
		}
		if (shard_reads && lsq_events_host_genes(F.e) > 0) {
			logf(1, "LSQ_SHARD=reads: %lld gene(s) beyond the device kernels' limits need all of their reads in one place; the job is sharded by events instead", (long long)lsq_events_host_genes(F.e));
			shard_reads = false;
		}
		if (shard_reads) continue;          // (the files open and are MRF_SINGLE: the GPUs read their slices below)
		// MRF text -> HBM -> parsed and ingested there; the name-keyed formats are grouped by name on the host first
		if (texts[(size_t)m]) {
			st = lsq_reads_upload_text(F.c, m, fmts[m], texts[(size_t)m]);
			if ((G == 1 && !getenv("LSQ_GATHER")) || st) { lsq_text_free(texts[(size_t)m]); texts[(size_t)m] = nullptr; }      // several GPUs: GPU 0 ingests it again, for its slice
		} else st = named_read_format(fmts[m]) ? LSQ_E_UNSUPPORTED : lsq_reads_upload_mrf(F.c, m, fmts[m], paths[m]);
		if (st == LSQ_E_UNSUPPORTED) {
			// a strand string beyond the device parser's 7 bytes: the host parser reads such files
			lsq_reads *r = nullptr;
			st = lsq_reads_parse(fmts[m], paths[m], F.e, 0, &r);
			if (!st) { F.r.push_back(r); st = lsq_reads_upload(F.c, m, r); lsq_reads_free(r); F.r.back() = nullptr; }
		}
		if (st == LSQ_E_PARSE) { logf(0, "%s", lsq_last_error()); logf(0, "Lexical_cast error when converting arguments to numeric values"); return status_to_exit(st); }
		if (st) { logf(0, "%s", lsq_last_error()); return st == LSQ_E_DEVICE ? 3 : (st == LSQ_E_IO || st == LSQ_E_FORMAT ? status_to_exit(st) : 2); }
		logf(2, "Sampling method #%d: loaded %llu reads associated with the selected gene regions", m, (unsigned long long)lsq_reads_retained(F.c, m));
		T.mark("reads: copy, parse, ingest");
		if (T.on) { float h2d = 0, parse = 0; lsq_last_mrf_timing(F.c, &h2d, &parse); fprintf(stderr, "[timing] %-28s %.3f s\n[timing] %-28s %.3f s\n", "  of which text copy", h2d * 1e-3, "  of which parse kernels", parse * 1e-3); }
	}
	// (read-sharded: the reads are loaded below, and a line that fails the cast there comes first, as in the reference)
	if (bad_type && n_ev > 0 && !shard_reads) { logf(0, "Unknown read type error: %s", bad_type_name.c_str()); return 1; }
	logf(2, "Processing reads info for genes");
	if (shard_reads && n_ev > 0) {
		ShardedJob J{solve, G, M, F.a, fmts, use_types, paths, lens, devices};
		const int rc = run_read_sharded_job(J, F.c, F.e);
		T.mark("read-sharded ingest + count + sum");
		if (rc) return rc;
		if (bad_type) { logf(0, "Unknown read type error: %s", bad_type_name.c_str()); return 1; }
		G = 1;                  // the sums are GPU 0's counts now: the rest is a single-GPU run
	} else {
		shard_reads = false;
		st = lsq_count(F.c);
		if (st) { logf(0, "%s", lsq_last_error()); return 3; }
	}
	{
		uint64_t hg = 0, hr = 0;
		if (lsq_host_evaluated(F.c, &hg, &hr) == LSQ_OK && hg)
			logf(1, "%llu gene(s) beyond the device kernels' limits (more than %d isoforms or %d segments, or a cluster too large for the LDS) are evaluated on the host: %llu reads",
			     (unsigned long long)hg, LSQ_MAX_ISOFORMS, LSQ_MAX_SEGMENTS, (unsigned long long)hr);
	}
	// (LSQ_GPUS=1 with LSQ_GATHER=rccl spelled out takes the same path with one slice: a self-check of the gather on one GPU)
	const char *gm = getenv("LSQ_GATHER");
	if ((G > 1 || (getenv("LSQ_GPUS") && gm && strcmp(gm, "rccl") == 0 && !want_fim)) && n_ev > 0) {
		ShardedJob J{solve, G, M, F.a, fmts, use_types, paths, lens, devices};
		const int rc = run_sharded_job(J, F.c, F.e, texts, trb, out);
		T.mark("sharded job");
		return rc;
	}
	if (solve) st = lsq_solve(F.c);
	if (st) { logf(0, "%s", lsq_last_error()); return 3; }
	const size_t n_cls = (size_t)lsq_results_num_classes(F.c);
	std::vector<uint64_t> cnt(std::max<size_t>((size_t)M * n_cls, 1)), bases(std::max<size_t>((size_t)M * n_cls, 1));
	st = lsq_results_counts(F.c, cnt.data(), bases.data());
	if (st) { logf(0, "%s", lsq_last_error()); return 3; }
	T.mark("count + solve + fetch");
	char *text = nullptr;
	if (!solve) {
		st = lsq_format_count(F.e, M, cnt.data(), &text);
	} else {
		std::vector<double> theta(std::max<size_t>((size_t)lsq_events_total_isoforms(F.e), 1)), ll(std::max<size_t>((size_t)n_ev, 1));
		std::vector<uint8_t> flags(std::max<size_t>((size_t)n_ev, 1));
		st = lsq_results_solve(F.c, theta.data(), ll.data(), nullptr, flags.data());
		if (st) { logf(0, "%s", lsq_last_error()); return 3; }
		for (int64_t i = 0; i < n_ev; ++i)
			if (flags[i] & 4) logf(3, "gene %s: EM stop criterion within the guard band of its threshold; solved again in per-read summation order", lsq_events_gene_name(F.e, i));
			else if (flags[i] & 1) logf(1, "gene %s: EM stop criterion within the guard band of its threshold and no exact-order replay was possible", lsq_events_gene_name(F.e, i));
		st = lsq_format_solve(F.e, M, cnt.data(), bases.data(), theta.data(), ll.data(), trb.data(), &text);
	}
	if (st || !text) { logf(0, "%s", lsq_last_error()); return 2; }
	out.assign(text);
	free(text);
	if (want_fim) {
		if (lsq_fim(F.c)) { logf(0, "%s", lsq_last_error()); return 3; }
		std::vector<uint64_t> off((size_t)n_ev + 1);
		lsq_results_fim_offsets(F.c, off.data());
		const size_t total = (size_t)off[(size_t)n_ev];
		std::vector<double> fim(std::max<size_t>((size_t)M * total, 1)), vd(std::max<size_t>((size_t)M * (size_t)n_ev, 1)), vi(vd.size());
		if (lsq_results_fim(F.c, fim.data(), vd.data(), vi.data())) { logf(0, "%s", lsq_last_error()); return 3; }
		char buf[64];
		auto num = [&](double v) { snprintf(buf, sizeof buf, "%.17g", v); out += buf; };
		for (int64_t i = 0; i < n_ev; ++i)
			for (int m = 0; m < M; ++m) {
				out += "#fim\t"; out += lsq_events_gene_name(F.e, i); out += "\t";
				snprintf(buf, sizeof buf, "%d", m); out += buf; out += "\t";
				num(vd[(size_t)m * (size_t)n_ev + (size_t)i]); out += "\t"; num(vi[(size_t)m * (size_t)n_ev + (size_t)i]);
				for (uint64_t q = off[(size_t)i]; q < off[(size_t)i + 1]; ++q) { out += "\t"; num(fim[(size_t)m * total + (size_t)q]); }
				out += "\n";
			}
	}
	T.mark("format rows");
	logf(2, "Processed %lld genes... Done", (long long)n_ev);
	if (g_exit_after_output) {
		// an executable (lsq_cli_main): the table is complete -- write it and leave.  Freeing gigabytes of HBM pools buffer by
		// buffer, joining the helper threads and unloading the runtime is work the end of the process does for nothing.
		// (profilers and coverage tools that write their output at exit want LSQ_CLI_TEARDOWN=1)
		const bool written = fwrite(out.data(), 1, out.size(), stdout) == out.size() && fflush(stdout) == 0;
		if (!written) logf(0, "cannot write the table to stdout: %s", strerror(errno));
		fflush(stderr);
		_exit(written ? 0 : 2);
	}
	return 0;
}

// The rows of the events [0, n), event by event in order: formatted by a few threads over contiguous runs of events
// (100 000 rows through the reference's six-digit formatting take 0.05 s on one core), joined in order.
template <class F>
std::string format_events_parallel(size_t n, F &&one) {
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	const size_t T = n < 8192 ? 1 : std::min<size_t>({8, hw, n / 4096});
	std::vector<std::string> parts(T);
	auto run = [&](size_t k) { for (size_t i = n * k / T; i < n * (k + 1) / T; ++i) one(parts[k], i); };
	{
		ThreadGroup th;          // joined on every way out; a body's exception (out of memory) is rethrown here, on the caller's thread
		for (size_t k = 1; k < T; ++k) th.spawn([&run, k] { run(k); });
		th.run_here([&run] { run(0); });
		th.join();
		if (th.failed()) throw std::runtime_error("formatting the rows: " + th.error());
	}
	if (T == 1) return std::move(parts[0]);
	size_t total = 0;
	for (const auto &s : parts) total += s.size();
	std::string o;
	o.reserve(total);
	for (const auto &s : parts) o += s;
	return o;
}

} // namespace

extern "C" {

int lsq_format_count(const lsq_events *E, int M, const uint64_t *cnt, char **out_text) LSQ_API_TRY {
	if (!E || !cnt || !out_text || M != E->n_methods) return fail(LSQ_E_ARG, "bad argument");
	const size_t n_cls = E->class_off.back();
	const std::string o = format_events_parallel(E->ev.size(), [&](std::string &o, size_t i) {
		const Event &e = E->ev[i];
		const size_t nc = (1u << e.K) - 1u;
		std::vector<double> supports(M, 0.0), iso_count(e.K, 0.0);
		for (int m = 0; m < M; ++m)
			for (size_t c = 1; c <= nc; ++c) {
				double v = (double)cnt[(size_t)m * n_cls + E->class_off[i] + c - 1];
				supports[m] += v;
				for (int j = 0; j < e.K; ++j) if (c >> j & 1) iso_count[j] += v;
			}
		for (int j = 0; j < e.K; ++j) {
			o += e.gname; o += '\t';
			for (int m = 0; m < M; ++m) { put_g(o, supports[m]); o += '\t'; }
			o += e.iso_names[j]; o += '\t'; put_g(o, iso_count[j]); o += '\n';
		}
	});
	*out_text = dup_text(o);
	return *out_text ? LSQ_OK : fail(LSQ_E_ARG, "out of memory");
} LSQ_API_CATCH

int lsq_format_solve(const lsq_events *E, int M, const uint64_t *cnt, const uint64_t *bases,
                     const double *theta, const double *logll, const double *total_read_bases, char **out_text) LSQ_API_TRY {
	if (!E || !cnt || !bases || !theta || !logll || !total_read_bases || !out_text || M != E->n_methods) return fail(LSQ_E_ARG, "bad argument");
	const size_t n_cls = E->class_off.back();
	const std::string o = format_events_parallel(E->ev.size(), [&](std::string &o, size_t i) {
		const Event &e = E->ev[i];
		const size_t nc = (1u << e.K) - 1u;
		std::vector<double> supports(M, 0.0), support_bases(M, 0.0);
		for (int m = 0; m < M; ++m) {
			uint64_t s = 0, b = 0;
			for (size_t c = 0; c < nc; ++c) { s += cnt[(size_t)m * n_cls + E->class_off[i] + c]; b += bases[(size_t)m * n_cls + E->class_off[i] + c]; }
			supports[m] = (double)s; support_bases[m] = (double)b;
		}
		const double *th = theta + E->iso_off[i];
		// solve/solve.cpp:808-821, same operation order
		std::vector<double> rpkm(e.K, 0.0);
		double total_read_mbases = 0.0;
		for (int m = 0; m < M; ++m) {
			total_read_mbases += total_read_bases[m] / 1.0E6;
			for (int j = 0; j < e.K; ++j) rpkm[j] += support_bases[m] * th[j];
		}
		for (int j = 0; j < e.K; ++j) {
			rpkm[j] /= ((double)e.iso_len[j] / 1.0E3);
			rpkm[j] /= total_read_mbases;
		}
		double sum_supports = 0.0;
		for (int m = 0; m < M; ++m) sum_supports += supports[m];
		for (int j = 0; j < e.K; ++j) {
			o += e.gname; o += '\t';
			for (int m = 0; m < M; ++m) { put_g(o, supports[m]); o += '\t'; }
			o += e.iso_names[j]; o += '\t'; put_g(o, th[j]); o += '\t'; put_g(o, rpkm[j]);
			if (sum_supports > 1E-5) { o += '\t'; put_g(o, logll[i] / sum_supports); o += '\n'; }
			else o += "\t0\n";
		}
	});
	*out_text = dup_text(o);
	return *out_text ? LSQ_OK : fail(LSQ_E_ARG, "out of memory");
} LSQ_API_CATCH

int lsq_cli_main(const char *tool, int argc, const char *const *argv) LSQ_API_TRY {
	g_is_executable = true;
	g_exit_after_output = getenv("LSQ_CLI_TEARDOWN") == nullptr;
	char *text = nullptr;
	int rc = lsq_cli_run(tool, argc, argv, &text);
	if (text) {
		const size_t n = strlen(text);
		if ((fwrite(text, 1, n, stdout) != n || fflush(stdout) != 0) && rc == 0) { logf(0, "cannot write the table to stdout: %s", strerror(errno)); rc = 2; }
		free(text);
	}
	return rc;
} LSQ_API_CATCH

int lsq_cli_run(const char *tool, int argc, const char *const *argv, char **out_text) LSQ_API_TRY {
	std::string out;
	int rc;
	if (tool && strcmp(tool, "count") == 0) rc = run_count_solve(false, argc, argv, out);
	else if (tool && strcmp(tool, "solve") == 0) rc = run_count_solve(true, argc, argv, out);
	else if (tool && strcmp(tool, "classify") == 0) rc = run_classify(argc, argv);
	else { fail(LSQ_E_ARG, "unknown tool"); return 2; }
	if (out_text) *out_text = dup_text(out);
	return rc;
} catch (...) {
	// exit status 2: "a condition the reference has no defined behaviour for" (it would have died of the exception)
	lsq::fail_exception(__func__);
	logf(0, "%s", lsq_last_error());
	return 2;
}

} // extern "C"
