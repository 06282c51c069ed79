// Micro-benchmark: where the dispatcher puts the workgroups of a launch that does NOT fill every place of the device - e.g. 1 032 workgroups
// of 256 threads where five fit on a CU (1 280 places): evenly (4 per CU and a few fifth ones) or CU after CU (5, 5, 5, ... and CUs left empty)?
// Every workgroup records its XCC / SE / CU and stays for ~100 us so that all are resident together.   ./wg_placement.bin [grid] [lds_bytes] [vgprs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int VGPRS>
__global__ __launch_bounds__(256) void place(unsigned int *out, unsigned long long ticks)
{
	extern __shared__ char lds[];
	if (VGPRS > 64)
		asm volatile("v_mov_b32 v95, 0" ::: "v95");
	if (VGPRS > 96)
		asm volatile("v_mov_b32 v127, 0" ::: "v127");
	if (threadIdx.x == 0)
	{
		unsigned int hw, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		out[blockIdx.x * 2] = hw, out[blockIdx.x * 2 + 1] = xcc;
		lds[0] = 1;
	}
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
		__builtin_amdgcn_s_sleep(8);
}

int main(int argc, char **argv)
{
	const int grid = argc > 1 ? atoi(argv[1]) : 1032, lds = argc > 2 ? atoi(argv[2]) : 23000, vg = argc > 3 ? atoi(argv[3]) : 96;
	unsigned int *d;
	hipMalloc(&d, grid * 8);
	hipMemset(d, 0, grid * 8);
	if (vg > 96)
		hipLaunchKernelGGL(place<128>, dim3(grid), dim3(256), lds, 0, d, 10000ull);
	else if (vg > 64)
		hipLaunchKernelGGL(place<96>, dim3(grid), dim3(256), lds, 0, d, 10000ull);
	else
		hipLaunchKernelGGL(place<64>, dim3(grid), dim3(256), lds, 0, d, 10000ull);
	if (hipDeviceSynchronize() != hipSuccess)
		return 1;
	std::vector<unsigned int> h(grid * 2);
	hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
	std::map<unsigned int, int> per_cu;
	for (int i = 0; i < grid; ++i)
	{
		// HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx90a+: se 3 bits)
		const unsigned int hw = h[2 * i], cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[2 * i + 1] & 15;
		per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu] += 1;
	}
	std::map<int, int> hist;
	for (auto &kv : per_cu)
		hist[kv.second] += 1;
	printf("grid %d, %d B of LDS, %d VGPRs: %zu CUs used;", grid, lds, vg, per_cu.size());
	for (auto &kv : hist)
		printf("  %d CUs hold %d workgroups;", kv.second, kv.first);
	printf("\n  first workgroups -> (xcc, se, cu):");
	for (int i = 0; i < 20 && i < grid; ++i)
		printf(" %u.%u.%u", h[2 * i + 1] & 15, (h[2 * i] >> 13) & 7, (h[2 * i] >> 8) & 15);
	printf("\n");
	return 0;
}
