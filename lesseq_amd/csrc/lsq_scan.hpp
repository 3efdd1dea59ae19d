// Exclusive prefix sums on the device, for the loader (included by lsq_ingest.hip; not a public header).
//
//   out[i] = sum of f(in[0..i)),  out[n] = the total          (u32 in, u64 out)
//
// f rounds every value up to a multiple of PAD (a power of two) -- the padded sizes of the pools' groups -- or, with
// FLAG, turns it into 0 / 1.  Three launches over blocks of 4 096 values: block sums, one workgroup over the block sums
// (a wave scan per 64, then over the waves), the blocks again with their bases.  A million values take ~15 us; the
// one-workgroup form this replaces took 0.6-2.4 ms of a C3 ingest each, seven times.
#pragma once

namespace {

constexpr unsigned SCAN_BLOCK = 4096;          // values per workgroup: 256 lanes x 16

template <unsigned PAD, bool FLAG>
__device__ inline unsigned scan_value(unsigned v) {
	if (FLAG) return v ? 1u : 0u;
	return (v + (PAD - 1u)) & ~(PAD - 1u);
}

__device__ inline unsigned long long scan_wave_incl(unsigned long long v) {
	const unsigned lane = threadIdx.x & 63u;
#pragma unroll
	for (unsigned d = 1; d < 64; d <<= 1) {
		const unsigned lo = (unsigned)__shfl_up((int)(unsigned)v, d), hi = (unsigned)__shfl_up((int)(unsigned)(v >> 32), d);
		if (lane >= d) v += ((unsigned long long)hi << 32) | lo;
	}
	return v;
}
// exclusive prefix of v over the workgroup (any multiple of 64 lanes up to 1 024); `total` = the sum
__device__ inline unsigned long long scan_block_excl(unsigned long long v, unsigned long long *lds16, unsigned long long &total) {
	const unsigned long long inc = scan_wave_incl(v);
	const unsigned w = threadIdx.x >> 6, nw = blockDim.x >> 6;
	if ((threadIdx.x & 63u) == 63u) lds16[w] = inc;
	__syncthreads();
	unsigned long long base = 0; total = 0;
	for (unsigned q = 0; q < nw; ++q) { const unsigned long long t = lds16[q]; base += q < w ? t : 0ull; total += t; }
	__syncthreads();
	return base + inc - v;
}

template <unsigned PAD, bool FLAG>
__global__ void __launch_bounds__(256) lsq_scan_sums_kernel(const unsigned *in, unsigned long long n, unsigned long long *block_sum) {
	__shared__ unsigned long long lds16[16];
	const unsigned long long b0 = (unsigned long long)blockIdx.x * SCAN_BLOCK;
	unsigned long long acc = 0;
#pragma unroll
	for (unsigned q = 0; q < SCAN_BLOCK / 256; ++q) {
		const unsigned long long i = b0 + q * 256u + threadIdx.x;
		if (i < n) acc += scan_value<PAD, FLAG>(in[i]);
	}
	unsigned long long total;
	(void)scan_block_excl(acc, lds16, total);
	if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

// one workgroup: the block sums in place -> their exclusive prefix; block_sum[n_blocks] = the total
__global__ void __launch_bounds__(1024) lsq_scan_spine_kernel(unsigned long long *block_sum, unsigned long long n_blocks) {
	__shared__ unsigned long long lds16[16];
	__shared__ unsigned long long carry_s;
	if (threadIdx.x == 0) carry_s = 0;
	__syncthreads();
	for (unsigned long long b0 = 0; b0 < n_blocks; b0 += 1024) {
		const unsigned long long i = b0 + threadIdx.x;
		const unsigned long long v = i < n_blocks ? block_sum[i] : 0ull;
		unsigned long long total;
		const unsigned long long ex = scan_block_excl(v, lds16, total);
		const unsigned long long carry = carry_s;
		if (i < n_blocks) block_sum[i] = carry + ex;
		__syncthreads();
		if (threadIdx.x == 0) carry_s = carry + total;
		__syncthreads();
	}
	if (threadIdx.x == 0) block_sum[n_blocks] = carry_s;
}

template <unsigned PAD, bool FLAG>
__global__ void __launch_bounds__(256) lsq_scan_apply_kernel(const unsigned *in, unsigned long long n, const unsigned long long *block_base, unsigned long long *out) {
	__shared__ unsigned long long lds16[16];
	const unsigned long long b0 = (unsigned long long)blockIdx.x * SCAN_BLOCK;
	// a lane takes 16 consecutive values
	const unsigned long long i0 = b0 + threadIdx.x * 16ull;
	unsigned v[16];
	unsigned long long acc = 0;
#pragma unroll
	for (unsigned q = 0; q < 16; ++q) { v[q] = i0 + q < n ? scan_value<PAD, FLAG>(in[i0 + q]) : 0u; acc += v[q]; }
	unsigned long long total;
	unsigned long long run = block_base[blockIdx.x] + scan_block_excl(acc, lds16, total);
#pragma unroll
	for (unsigned q = 0; q < 16; ++q) { if (i0 + q < n) out[i0 + q] = run; run += v[q]; }
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = block_base[gridDim.x];
}

// scratch of the scans of one call chain (block sums); grows as needed
struct ScanScratch {
	DevBuf<unsigned long long> sums;
	int reserve(unsigned long long n) {
		const size_t want = (size_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK) + 2;
		if (sums.n >= want) return LSQ_OK;
		return sums.alloc(want);
	}
};

// out must hold n + 1 values.  The scratch must have been reserved for n (no allocation between launches: hipMalloc may wait for the device)
template <unsigned PAD, bool FLAG = false>
static int device_scan(ScanScratch &S, const unsigned *in, unsigned long long n, unsigned long long *out, hipStream_t st) {
	if (n == 0) { HIP_TRY(hipMemsetAsync(out, 0, 8, st)); return LSQ_OK; }
	const unsigned long long nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
	if (nb > 0x7FFFFFFFull) return fail(LSQ_E_RANGE, "prefix sum over more than 2^43 values");
	if (S.sums.n < nb + 1) return fail(LSQ_E_INTERNAL, "scan scratch not reserved");
	hipLaunchKernelGGL((lsq_scan_sums_kernel<PAD, FLAG>), dim3((unsigned)nb), dim3(256), 0, st, in, n, S.sums.p);
	hipLaunchKernelGGL(lsq_scan_spine_kernel, dim3(1), dim3(1024), 0, st, S.sums.p, nb);
	hipLaunchKernelGGL((lsq_scan_apply_kernel<PAD, FLAG>), dim3((unsigned)nb), dim3(256), 0, st, in, n, (const unsigned long long *)S.sums.p, out);
	HIP_TRY(hipGetLastError());
	return LSQ_OK;
}

} // namespace
