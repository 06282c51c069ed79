#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
	int v = threadIdx.x * 10;
	int a = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false); // wave_shl:1
	int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false); // wave_shr:1
	out[threadIdx.x] = a;
	out[64 + threadIdx.x] = b;
}
int main()
{
	int *d, h[128];
	hipMalloc(&d, 512);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
	printf("shl: %d %d %d ... %d %d | shr: %d %d %d ... %d %d\n", h[0], h[1], h[2], h[62], h[63], h[64], h[65], h[66], h[126], h[127]);
	printf("shl15..17: %d %d %d  shl31..33 %d %d %d\n", h[15], h[16], h[17], h[31], h[32], h[33]);
	return 0;
}
