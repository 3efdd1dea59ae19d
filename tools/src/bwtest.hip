// Developer microbenchmark: HBM read bandwidth reachable by a plain streaming kernel on this box
// (the practical ceiling next to the 8 TB/s peak used in bench.py's roofline).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bwtest tools/src/bwtest.hip ; run: tools/bwtest [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int UNROLL>
__global__ void __launch_bounds__(256) read_kernel(const uint4 *src, size_t n_words, unsigned *sink) {
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

// the count kernel's pattern: every workgroup streams its own contiguous chunk; its four waves take
// 2 KiB tiles in turn, two 16-byte words per lane (lane-major), next tile in flight while the
// current one is consumed
__global__ void __launch_bounds__(256) chunk_kernel(const uint4 *src, size_t n_words, unsigned *sink, int lane_major) {
	const size_t per = (n_words + gridDim.x - 1) / gridDim.x;
	const size_t w0 = per * blockIdx.x, w1 = w0 + per < n_words ? w0 + per : n_words;
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned acc = 0;
	uint4 nxt[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
	size_t wt = w0 + wave * 128u;
	auto fetch = [&](size_t t) {
		for (int k = 0; k < 2; ++k) {
			const size_t w = lane_major ? t + lane * 2u + k : t + k * 64u + lane;
			if (w < w1) nxt[k] = src[w]; else nxt[k] = make_uint4(0, 0, 0, 0);
		}
	};
	if (wt < w1) fetch(wt);
	for (; wt < w1; wt += 512u) {
		const uint4 a = nxt[0], b = nxt[1];
		if (wt + 512u < w1) fetch(wt + 512u);
		acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
	}
	if (acc == 0x12345678u) *sink = acc;
}

static float run_chunk(const uint4 *d, size_t n_words, unsigned *sink, int grid, int lane_major) {
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	std::vector<float> ts;
	for (int it = 0; it < 12; ++it) {
		(void)hipEventRecord(a);
		hipLaunchKernelGGL(chunk_kernel, dim3(grid), dim3(256), 0, 0, d, n_words, sink, lane_major);
		(void)hipEventRecord(b);
		(void)hipEventSynchronize(b);
		float ms; (void)hipEventElapsedTime(&ms, a, b);
		if (it >= 2) ts.push_back(ms);
	}
	std::sort(ts.begin(), ts.end());
	return ts[ts.size() / 2];
}

template <int UNROLL>
static float run(const uint4 *d, size_t n_words, unsigned *sink, int grid) {
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	std::vector<float> ts;
	for (int it = 0; it < 12; ++it) {
		hipEventRecord(a);
		hipLaunchKernelGGL(read_kernel<UNROLL>, dim3(grid), dim3(256), 0, 0, d, n_words, sink);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		if (it >= 2) ts.push_back(ms);
	}
	std::sort(ts.begin(), ts.end());
	return ts[ts.size() / 2];
}

int main(int argc, char **argv) {
	const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 800;
	const size_t bytes = mib << 20, n_words = bytes / 16;
	uint4 *d; unsigned *sink;
	if (hipMalloc((void **)&d, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(d, 1, bytes);
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	for (int per_cu : {4, 8, 16, 32}) {
		const int grid = p.multiProcessorCount * per_cu;
		printf("%zu MiB grid=%d  unroll2 %.0f GB/s  unroll4 %.0f GB/s  unroll8 %.0f GB/s\n", mib, grid,
		       bytes / run<2>(d, n_words, sink, grid) / 1e6, bytes / run<4>(d, n_words, sink, grid) / 1e6, bytes / run<8>(d, n_words, sink, grid) / 1e6);
	}
	for (int per_cu : {4, 5, 8, 10, 20, 40})
		printf("%zu MiB chunked grid=%d  coalesced %.0f GB/s  lane-major %.0f GB/s\n", mib, p.multiProcessorCount * per_cu,
		       bytes / run_chunk(d, n_words, sink, p.multiProcessorCount * per_cu, 0) / 1e6, bytes / run_chunk(d, n_words, sink, p.multiProcessorCount * per_cu, 1) / 1e6);
	return 0;
}
