// LDS atomic add (no return) rate on one CU: wave-cycles per ds_add_u32 for several address patterns, 16 waves per workgroup,
// one and two workgroups per CU (64 KB of LDS each, as the histogram pass of lossy_kernels.hip).  Also global (L2) atomics.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic_rate scripts/ubench/lds_atomic_rate.hip && /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(1024) void k_lds(const uint32_t *__restrict__ offs, int iters, uint32_t *out)
{
	__shared__ uint32_t lh[16384];
	for (int i = threadIdx.x; i < 16384; i += 1024)
		lh[i] = 0;
	__syncthreads();
	uint32_t o[8];
	for (int j = 0; j < 8; ++j)
		o[j] = offs[(blockIdx.x * 1024 + threadIdx.x) * 8 + j];
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int j = 0; j < 8; ++j)
			atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lh) + o[j]), 1u);
#pragma unroll
		for (int j = 0; j < 8; ++j)
			o[j] = (o[j] * 5u + 4u * (uint32_t)it) & 0xffffu & ~3u & offs[0] | (o[j] & ~offs[0]); // (offs[0] = mask of the bits that move)
	}
	__syncthreads();
	if (threadIdx.x == 0)
		out[blockIdx.x] = lh[0] + lh[1];
}
__global__ __launch_bounds__(1024) void k_glb(const uint32_t *__restrict__ offs, int iters, uint32_t *hist)
{
	uint32_t o[8];
	for (int j = 0; j < 8; ++j)
		o[j] = offs[(blockIdx.x * 1024 + threadIdx.x) * 8 + j];
	uint32_t *h = hist + (size_t)blockIdx.x * 16384;
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int j = 0; j < 8; ++j)
			__hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(h) + o[j]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}
int main()
{
	const int nwg = 512, iters = 40;
	std::vector<uint32_t> h((size_t)nwg * 1024 * 8);
	uint32_t *d_offs, *d_out, *d_hist;
	hipMalloc(&d_offs, h.size() * 4);
	hipMalloc(&d_out, nwg * 4);
	hipMalloc(&d_hist, (size_t)nwg * 65536);
	hipMemset(d_hist, 0, (size_t)nwg * 65536);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	const char *names[] = {"lane*4 (one bank each)", "random in 250 bins", "random in 4096 bins", "one address", "random in 16 bins"};
	for (int pat = 0; pat < 5; ++pat)
	{
		uint32_t seed = 12345;
		for (size_t i = 0; i < h.size(); ++i)
		{
			seed = seed * 1664525u + 1013904223u;
			const uint32_t lane = (uint32_t)((i / 8) % 64), r = seed >> 8;
			h[i] = pat == 0 ? lane * 4 : pat == 1 ? 1000 + (r % 250) * 4 : pat == 2 ? (r % 4096) * 4 : pat == 3 ? 2000 : 3000 + (r % 16) * 4;
		}
		h[0] = 0; // (addresses stay put)
		hipMemcpy(d_offs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
		for (int g = 0; g < 2; ++g)
		{
			const int wgs = g == 0 ? 256 : 512;
			float best = 1e9f;
			for (int rep = 0; rep < 4; ++rep)
			{
				hipEventRecord(e0);
				hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(1024), 0, 0, d_offs, iters, d_out);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			const double instr_per_cu = (double)wgs / 256 * 16 * iters * 8; // wave instructions per CU
			printf("LDS  %-26s %d WG/CU: %7.1f us  -> %5.1f ns per wave instruction and CU (%.2f lanes/ns/CU)\n", names[pat], wgs / 256, best * 1e3, best * 1e6 / instr_per_cu,
				   64.0 * instr_per_cu / (best * 1e6));
		}
		{
			float best = 1e9f;
			for (int rep = 0; rep < 3; ++rep)
			{
				hipEventRecord(e0);
				hipLaunchKernelGGL(k_glb, dim3(512), dim3(1024), 0, 0, d_offs, iters, d_hist);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			printf("L2   %-26s 512 WG  : %7.1f us  -> %.1f G lane-atomics/s over the chip\n", names[pat], best * 1e3, 512.0 * 1024 * 8 * iters / (best * 1e-3) / 1e9);
		}
	}
	return 0;
}
