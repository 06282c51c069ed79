// development aid: how many 256-thread workgroups fit a CU for a given dynamic LDS size (allocation granularity of LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(unsigned long long *p)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
	sm[threadIdx.x] = threadIdx.x;
	__syncthreads();
	p[threadIdx.x] = sm[255 - threadIdx.x];
}
int main()
{
	int last = -1;
	for (int bytes = 8192; bytes <= 40960; bytes += 64)
	{
		int n = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, bytes) != hipSuccess)
			break;
		if (n != last)
			printf("lds %6d bytes -> %d workgroups per CU\n", bytes, n), last = n;
	}
	return 0;
}
