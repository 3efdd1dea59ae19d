// Developer probe: host cost of submitting a step's result-stream chain (5 small kernels + a memset + an event) call by
// call against one hipGraphLaunch of the same chain.  hipcc --offload-arch=gfx950 -O2 graph_probe.hip -o graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Args { unsigned long long *p; unsigned v[60]; };
__global__ void small(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[a.v[0] & 7] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	unsigned long long *d; CK(hipMalloc(&d, 1 << 20));
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	Args a{}; a.p = d;
	auto chain = [&](hipStream_t st) {
		for (int k = 0; k < 5; ++k) { a.v[0] = k; hipLaunchKernelGGL(small, dim3(64), dim3(64), 0, st, a); }
		(void)hipMemsetAsync(d + 1024, 0, 65536, st);
	};
	const int N = 2000;
	for (int rep = 0; rep < 2; ++rep) {
		CK(hipStreamSynchronize(s));
		double t0 = now();
		for (int i = 0; i < N; ++i) { chain(s); (void)hipEventRecord(ev, s); }
		double t1 = now();
		CK(hipStreamSynchronize(s));
		double t2 = now();
		printf("call by call: host %.2f us per chain, through %.2f us\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
	}
	hipGraph_t g; hipGraphExec_t ge;
	CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
	chain(s);
	CK(hipStreamEndCapture(s, &g));
	CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	for (int rep = 0; rep < 2; ++rep) {
		CK(hipStreamSynchronize(s));
		double t0 = now();
		for (int i = 0; i < N; ++i) { CK(hipGraphLaunch(ge, s)); (void)hipEventRecord(ev, s); }
		double t1 = now();
		CK(hipStreamSynchronize(s));
		double t2 = now();
		printf("graph:        host %.2f us per chain, through %.2f us\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
	}
	return 0;
}
