// liblesseq_rccl.so: the RCCL all-gather of the per-event records of an event-sharded job (include/lesseq_rccl.h).
// The only translation unit that links librccl; the core library loads it on demand (lsq_cli.cpp, LSQ_GPUS > 1).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/lesseq_rccl.h"

struct lsq_comm {
	ncclComm_t comm = nullptr;
	int rank = 0, size = 1, device = 0;
};

namespace {
thread_local std::string g_err;
int fail(int status, const char *fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	try { g_err = buf; } catch (...) { g_err.clear(); }
	return status;
}
static_assert(sizeof(ncclUniqueId) <= LSQ_COMM_ID_BYTES, "an RCCL id fits LSQ_COMM_ID_BYTES");

// RCCL prints a version banner on stdout when the first communicator is made; the stdout of count / solve carries
// result rows only (the reference's contract: log lines go to stderr), so file descriptor 1 points at stderr while
// a communicator is being set up.
struct StdoutToStderr {
	int saved = -1;
	StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) (void)dup2(2, 1); }
	~StdoutToStderr() { if (saved >= 0) { fflush(stdout); (void)dup2(saved, 1); close(saved); } }
};
} // namespace

// no exception crosses the C ABI (as in the core library, lsq_internal.hpp)
#define LSQ_API_TRY try
#define LSQ_API_CATCH                                                                                             \
	catch (const std::exception &e) { return fail(LSQ_E_INTERNAL, "%s: %s", __func__, e.what()); }                \
	catch (...) { return fail(LSQ_E_INTERNAL, "%s: unknown exception", __func__); }

#define NCCL_TRY(expr)                                                                            \
	do {                                                                                          \
		ncclResult_t _r = (expr);                                                                 \
		if (_r != ncclSuccess) return fail(LSQ_E_DEVICE, "%s: %s", #expr, ncclGetErrorString(_r)); \
	} while (0)

extern "C" {

const char *lsq_rccl_last_error(void) { return g_err.c_str(); }

int lsq_comm_init_all(int n, const int *devices, lsq_comm **comms) LSQ_API_TRY {
	if (n < 1 || !devices || !comms) return fail(LSQ_E_ARG, "bad argument");
	std::vector<ncclComm_t> cs((size_t)n);
	{
		StdoutToStderr guard;
		NCCL_TRY(ncclCommInitAll(cs.data(), n, devices));
	}
	for (int r = 0; r < n; ++r) {
		comms[r] = new lsq_comm;
		comms[r]->comm = cs[(size_t)r]; comms[r]->rank = r; comms[r]->size = n; comms[r]->device = devices[r];
	}
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_comm_unique_id(void *id) LSQ_API_TRY {
	if (!id) return fail(LSQ_E_ARG, "null argument");
	ncclUniqueId u;
	NCCL_TRY(ncclGetUniqueId(&u));
	memset(id, 0, LSQ_COMM_ID_BYTES);
	memcpy(id, &u, sizeof u);
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_comm_init_rank(int world, int rank, const void *id, int device, lsq_comm **out) LSQ_API_TRY {
	if (world < 1 || rank < 0 || rank >= world || !id || !out) return fail(LSQ_E_ARG, "bad argument");
	if (hipSetDevice(device) != hipSuccess) return fail(LSQ_E_DEVICE, "device %d cannot be selected", device);
	ncclUniqueId u;
	memcpy(&u, id, sizeof u);
	ncclComm_t c;
	{
		StdoutToStderr guard;
		NCCL_TRY(ncclCommInitRank(&c, world, u, rank));
	}
	*out = new lsq_comm;
	(*out)->comm = c; (*out)->rank = rank; (*out)->size = world; (*out)->device = device;
	return LSQ_OK;
} LSQ_API_CATCH

void lsq_comm_destroy(lsq_comm *comm) {
	if (!comm) return;
	if (comm->comm) (void)ncclCommDestroy(comm->comm);
	delete comm;
}

void lsq_comm_abort(lsq_comm *comm) {
	if (!comm) return;
	if (comm->comm) (void)ncclCommAbort(comm->comm);
	delete comm;
}

} // extern "C"

namespace {
// A blocking call on a helper thread, waited for with a time limit.  The thread owns everything it touches through the
// shared state, so giving it up (detached, still inside RCCL) leaves nothing dangling in the caller.
template <class Job>
int run_with_time_limit(double seconds, const char *what, std::shared_ptr<Job> job) {
	if (!(seconds > 0)) { job->run(); return job->status; }
	struct Wait { std::mutex mu; std::condition_variable cv; bool done = false; };
	auto w = std::make_shared<Wait>();
	std::thread([job, w] {
		try { job->run(); } catch (...) { job->status = LSQ_E_INTERNAL; job->error = "exception in the helper thread"; }
		{ std::lock_guard<std::mutex> g(w->mu); w->done = true; }
		w->cv.notify_all();
	}).detach();
	std::unique_lock<std::mutex> lk(w->mu);
	if (!w->cv.wait_for(lk, std::chrono::duration<double>(seconds), [&] { return w->done; }))
		return fail(LSQ_E_TIMEOUT, "%s did not return within %.0f s (a peer that never arrived?)", what, seconds);
	if (job->status) return fail(job->status, "%s", job->error.c_str());
	return LSQ_OK;
}
struct InitRankJob {
	int world, rank, device; ncclUniqueId id; ncclComm_t comm = nullptr; int status = LSQ_OK; std::string error;
	void run() {
		if (hipSetDevice(device) != hipSuccess) { status = LSQ_E_DEVICE; error = "the device cannot be selected"; return; }
		const ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
		if (r != ncclSuccess) { status = LSQ_E_DEVICE; error = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); }
	}
};
struct InitAllJob {
	std::vector<int> devices; std::vector<ncclComm_t> comms; int status = LSQ_OK; std::string error;
	void run() {
		const ncclResult_t r = ncclCommInitAll(comms.data(), (int)devices.size(), devices.data());
		if (r != ncclSuccess) { status = LSQ_E_DEVICE; error = std::string("ncclCommInitAll: ") + ncclGetErrorString(r); }
	}
};
} // namespace

extern "C" {

int lsq_comm_init_rank_for(int world, int rank, const void *id, int device, double seconds, lsq_comm **out) LSQ_API_TRY {
	if (world < 1 || rank < 0 || rank >= world || !id || !out) return fail(LSQ_E_ARG, "bad argument");
	auto job = std::make_shared<InitRankJob>();
	job->world = world; job->rank = rank; job->device = device;
	memcpy(&job->id, id, sizeof job->id);
	int rc;
	{
		StdoutToStderr guard;
		rc = run_with_time_limit(seconds, "ncclCommInitRank", job);
	}
	if (rc) return rc;
	if (hipSetDevice(device) != hipSuccess) return fail(LSQ_E_DEVICE, "device %d cannot be selected", device);
	*out = new lsq_comm;
	(*out)->comm = job->comm; (*out)->rank = rank; (*out)->size = world; (*out)->device = device;
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_comm_init_all_for(int n, const int *devices, double seconds, lsq_comm **comms) LSQ_API_TRY {
	if (n < 1 || !devices || !comms) return fail(LSQ_E_ARG, "bad argument");
	auto job = std::make_shared<InitAllJob>();
	job->devices.assign(devices, devices + n);
	job->comms.assign((size_t)n, nullptr);
	int rc;
	{
		StdoutToStderr guard;
		rc = run_with_time_limit(seconds, "ncclCommInitAll", job);
	}
	if (rc) return rc;
	for (int r = 0; r < n; ++r) {
		comms[r] = new lsq_comm;
		comms[r]->comm = job->comms[(size_t)r]; comms[r]->rank = r; comms[r]->size = n; comms[r]->device = devices[r];
	}
	return LSQ_OK;
} LSQ_API_CATCH

// the RCCL this library runs on, as ncclGetVersion's integer (e.g. 22203 = 2.22.3); 0 when the call fails
int lsq_rccl_version(void) { int v = 0; return ncclGetVersion(&v) == ncclSuccess ? v : 0; }
int lsq_comm_rank(const lsq_comm *comm) { return comm ? comm->rank : -1; }
int lsq_comm_size(const lsq_comm *comm) { return comm ? comm->size : 0; }

int lsq_gather(lsq_ctx *c, lsq_comm *comm, const void *d_block, void *d_gathered, uint64_t stride_words) LSQ_API_TRY {
	if (!c || !comm || !d_block || !d_gathered || !stride_words) return fail(LSQ_E_ARG, "bad argument");
	if (hipSetDevice(comm->device) != hipSuccess) return fail(LSQ_E_DEVICE, "device %d cannot be selected", comm->device);
	hipStream_t st = (hipStream_t)lsq_ctx_result_stream(c);         // behind lsq_results_pack_device
	NCCL_TRY(ncclAllGather(d_block, d_gathered, (size_t)stride_words, ncclUint64, comm->comm, st));
	return LSQ_OK;
} LSQ_API_CATCH

int lsq_allreduce_counts(lsq_ctx *c, lsq_comm *comm, void *d_words) LSQ_API_TRY {
	if (!c || !comm || !d_words) return fail(LSQ_E_ARG, "bad argument");
	if (hipSetDevice(comm->device) != hipSuccess) return fail(LSQ_E_DEVICE, "device %d cannot be selected", comm->device);
	const uint64_t words = lsq_counts_device_words(c);
	int rc = lsq_counts_export_device(c, d_words);           // on the result stream, behind the count's exception pass
	if (rc) return fail(rc, "%s", lsq_last_error());
	if (!words) return LSQ_OK;
	hipStream_t st = (hipStream_t)lsq_ctx_result_stream(c);
	NCCL_TRY(ncclAllReduce(d_words, d_words, (size_t)words, ncclUint64, ncclSum, comm->comm, st));
	return LSQ_OK;
} LSQ_API_CATCH

// One step of an event-sharded job in one call: count, solve, pack, gather (what a loop over batches does per batch;
// four calls through a binding's foreign-function layer cost more host time than the launches themselves).
int lsq_step_gather(lsq_ctx *c, lsq_comm *comm, void *d_block, void *d_gathered, uint64_t stride_words) LSQ_API_TRY {
	int rc;
	if ((rc = lsq_count(c)) || (rc = lsq_solve(c)) || (rc = lsq_results_pack_device(c, d_block))) return fail(rc, "%s", lsq_last_error());
	return lsq_gather(c, comm, d_block, d_gathered, stride_words);
} LSQ_API_CATCH

} // extern "C"
