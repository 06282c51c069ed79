// Is sqrt(double) / double division on the device bit-identical to the host's (IEEE, correctly rounded)?  The bounded-loss budget
// (h264.cpp:2335-2385) is double arithmetic; it can only move to the device if so.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off sqrt_f64_check.hip -o sqrt_f64_check && ./sqrt_f64_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <cstring>
__global__ void k(const double *a, const double *b, double *s, double *q, double *r, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		s[i] = sqrt(a[i] * a[i] - b[i]);
		q[i] = s[i] / b[i];
		r[i] = round(q[i] * 5.0);
	}
}
int main()
{
	const int n = 1 << 24;
	std::vector<double> a(n), b(n), s(n), q(n), r(n);
	uint64_t x = 88172645463325252ull;
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	long bad = 0;
	for (int rep = 0; rep < 8; ++rep)
	{
		for (int i = 0; i < n; ++i)
		{
			// sums of |d| over up to 786432 pixels, and sums of d^2 (integers), as the statistic sees them
			const uint64_t npx = 1 + rnd() % 786432;
			const uint64_t sd = rnd() % (npx * (1 + rnd() % 4000));
			const uint64_t sd2 = rnd() % (sd * (1 + rnd() % 4000) + 1);
			a[i] = (double)sd;
			b[i] = (double)(sd2 % (uint64_t)(a[i] * a[i] + 1) + 1);
		}
		double *da, *db, *ds, *dq, *dr;
		hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&dr, n * 8);
		hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
		hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, ds, dq, dr, n);
		hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
		hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
		hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
		for (int i = 0; i < n; ++i)
		{
			const double hs = std::sqrt(a[i] * a[i] - b[i]), hq = hs / b[i], hr = std::round(hq * 5.0);
			if (memcmp(&hs, &s[i], 8) || memcmp(&hq, &q[i], 8) || memcmp(&hr, &r[i], 8))
				if (++bad < 5)
					printf("mismatch a=%.17g b=%.17g host %.17g %.17g dev %.17g %.17g\n", a[i], b[i], hs, hq, s[i], q[i]);
		}
		hipFree(da); hipFree(db); hipFree(ds); hipFree(dq); hipFree(dr);
	}
	printf("sqrt/div/round f64: %ld mismatches in %ld samples\n", bad, 8L * n);
	return bad != 0;
}
