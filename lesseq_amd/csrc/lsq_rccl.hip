// liblesseq_rccl.so: the RCCL all-gather of the per-event records of an event-sharded job (include/lesseq_rccl.h).
// The only translation unit that links librccl; the core library loads it on demand (lsq_cli.cpp, LSQ_GPUS > 1).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
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
	g_err = buf;
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

#define NCCL_TRY(expr)                                                                            \
	do {                                                                                          \
		ncclResult_t _r = (expr);                                                                 \
		if (_r != ncclSuccess) return fail(LSQ_E_DEVICE, "%s: %s", #expr, ncclGetErrorString(_r)); \
	} while (0)

extern "C" {

const char *lsq_rccl_last_error(void) { return g_err.c_str(); }

int lsq_comm_init_all(int n, const int *devices, lsq_comm **comms) {
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
}

int lsq_comm_unique_id(void *id) {
	if (!id) return fail(LSQ_E_ARG, "null argument");
	ncclUniqueId u;
	NCCL_TRY(ncclGetUniqueId(&u));
	memset(id, 0, LSQ_COMM_ID_BYTES);
	memcpy(id, &u, sizeof u);
	return LSQ_OK;
}

int lsq_comm_init_rank(int world, int rank, const void *id, int device, lsq_comm **out) {
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
}

void lsq_comm_destroy(lsq_comm *comm) {
	if (!comm) return;
	if (comm->comm) (void)ncclCommDestroy(comm->comm);
	delete comm;
}

int lsq_comm_rank(const lsq_comm *comm) { return comm ? comm->rank : -1; }
int lsq_comm_size(const lsq_comm *comm) { return comm ? comm->size : 0; }

int lsq_gather(lsq_ctx *c, lsq_comm *comm, const void *d_block, void *d_gathered, uint64_t stride_words) {
	if (!c || !comm || !d_block || !d_gathered || !stride_words) return fail(LSQ_E_ARG, "bad argument");
	if (hipSetDevice(comm->device) != hipSuccess) return fail(LSQ_E_DEVICE, "device %d cannot be selected", comm->device);
	hipStream_t st = (hipStream_t)lsq_ctx_result_stream(c);         // behind lsq_results_pack_device
	NCCL_TRY(ncclAllGather(d_block, d_gathered, (size_t)stride_words, ncclUint64, comm->comm, st));
	return LSQ_OK;
}

// One step of an event-sharded job in one call: count, solve, pack, gather (what a loop over batches does per batch;
// four calls through a binding's foreign-function layer cost more host time than the launches themselves).
int lsq_step_gather(lsq_ctx *c, lsq_comm *comm, void *d_block, void *d_gathered, uint64_t stride_words) {
	int rc;
	if ((rc = lsq_count(c)) || (rc = lsq_solve(c)) || (rc = lsq_results_pack_device(c, d_block))) return fail(rc, "%s", lsq_last_error());
	return lsq_gather(c, comm, d_block, d_gathered, stride_words);
}

} // extern "C"
