// Micro-benchmark: HBM rate of the codec's access pattern (one wave = one 1 KiB tile, walking frames
// that are frame_bytes apart) against a plain linear copy.  ./tile_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

// MODE 0: read-only walk (sum), MODE 1: read + write (copy), MODE 2: write-only
template <int MODE, int TILE_KIB, bool SYNC>
__global__ __launch_bounds__(256) void walk(const uint4 *__restrict__ in, uint4 *__restrict__ out, int64_t frame_u4, int ntiles, int gop, uint32_t *sink)
{
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (tile >= ntiles)
		return;
	const int64_t base = (int64_t)blockIdx.y * gop * frame_u4 + (int64_t)tile * 64 * TILE_KIB + lane;
	uint4 acc = make_uint4(0, 0, 0, 0);
	uint4 ring[4][TILE_KIB];
#pragma unroll
	for (int s = 0; s < 4; ++s)
#pragma unroll
		for (int k = 0; k < TILE_KIB; ++k)
			if (MODE != 2)
				ring[s][k] = in[base + (int64_t)min(s, gop - 1) * frame_u4 + k * 64];
	for (int fb = 0; fb < gop; fb += 4)
	{
#pragma unroll
		for (int s = 0; s < 4; ++s)
		{
			const int f = fb + s;
			if (f < gop)
			{
#pragma unroll
				for (int k = 0; k < TILE_KIB; ++k)
				{
					uint4 v = ring[s][k];
					if (MODE != 2)
						ring[s][k] = in[base + (int64_t)min(f + 4, gop - 1) * frame_u4 + k * 64];
					acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
					if (MODE != 0)
					{
						if (MODE == 2) v = make_uint4(f, lane, tile, k);
						out[base + (int64_t)f * frame_u4 + k * 64] = v;
					}
				}
				if (SYNC)
					__syncthreads();
			}
		}
	}
	if (acc.x == 0x12345678u && acc.y == 77)
		sink[0] = acc.z + acc.w;
}

__global__ __launch_bounds__(256) void linear_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, int64_t n)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
		out[i] = in[i];
}

template <class F>
float time_ms(F f)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	f();
	(void)hipDeviceSynchronize();
	float best = 1e9;
	for (int r = 0; r < 5; ++r)
	{
		(void)hipEventRecord(e0);
		f();
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	return best;
}

int main()
{
	const int W = 640, H = 512, N = 1000, GOP = 50;
	const int64_t frame_bytes = (int64_t)W * H * 2, frame_u4 = frame_bytes / 16, total = frame_bytes * N;
	uint4 *in, *out;
	uint32_t *sink;
	(void)hipMalloc(&in, total);
	(void)hipMalloc(&out, total);
	(void)hipMalloc(&sink, 64);
	(void)hipMemset(in, 1, total);
	const int nchunks = N / GOP;
	auto report = [&](const char *name, float ms, double bytes) { printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); };
	report("linear copy (read+write)", time_ms([&] { hipLaunchKernelGGL(linear_copy, dim3(256 * 8), dim3(256), 0, 0, in, out, total / 16); }), 2.0 * total);
	{
		const int ntiles = (int)(frame_bytes / 1024);
		dim3 g((ntiles + 3) / 4, nchunks), b(256);
		report("tile walk 1 KiB/wave, read only", time_ms([&] { hipLaunchKernelGGL((walk<0, 1, false>), g, b, 0, 0, in, out, frame_u4, ntiles, GOP, sink); }), 1.0 * total);
		report("tile walk 1 KiB/wave, write only", time_ms([&] { hipLaunchKernelGGL((walk<2, 1, false>), g, b, 0, 0, in, out, frame_u4, ntiles, GOP, sink); }), 1.0 * total);
		report("tile walk 1 KiB/wave, read+write", time_ms([&] { hipLaunchKernelGGL((walk<1, 1, false>), g, b, 0, 0, in, out, frame_u4, ntiles, GOP, sink); }), 2.0 * total);
		report("tile walk 1 KiB/wave, read only, WG barrier", time_ms([&] { hipLaunchKernelGGL((walk<0, 1, true>), g, b, 0, 0, in, out, frame_u4, ntiles, GOP, sink); }), 1.0 * total);
	}
	{
		const int ntiles = (int)(frame_bytes / 2048);
		dim3 g((ntiles + 3) / 4, nchunks), b(256);
		report("tile walk 2 KiB/wave, read only", time_ms([&] { hipLaunchKernelGGL((walk<0, 2, false>), g, b, 0, 0, in, out, frame_u4, ntiles, GOP, sink); }), 1.0 * total);
		report("tile walk 2 KiB/wave, write only", time_ms([&] { hipLaunchKernelGGL((walk<2, 2, false>), g, b, 0, 0, in, out, frame_u4, ntiles, GOP, sink); }), 1.0 * total);
	}
	{
		const int ntiles = (int)(frame_bytes / 4096);
		dim3 g((ntiles + 3) / 4, nchunks), b(256);
		report("tile walk 4 KiB/wave, read only", time_ms([&] { hipLaunchKernelGGL((walk<0, 4, false>), g, b, 0, 0, in, out, frame_u4, ntiles, GOP, sink); }), 1.0 * total);
	}
	return 0;
}
