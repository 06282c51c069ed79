// Bounded-loss recording on the GPU: the per-frame loss-injection step of the reference's lossy saver
// (reference src/cpp/video_io/h264.cpp: get_background :1955-1991, stdDev :1993-2036,
// RunningAverage2 :1526-1615, decision loop :2397-2413 / :2574-2590).  The state is sequential in
// time and parallel in space: four elementwise / reduction kernels per frame, all integer, exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lossy_kernels.h"

namespace rir
{
	__device__ __forceinline__ uint32_t sub_min(uint32_t v, uint32_t mn) { return v < mn ? 0u : v - mn; }

	// L1: histogram of (v >> 2) over 16 384 bins, privatised in LDS (64 KiB), 1024-thread workgroups,
	// RIR_LOSSY_HIST_PX pixels per workgroup (one frame is a small job: 32 768 pixels per workgroup left it on ten CUs
	// and took 17 us; the merge only touches the bins a workgroup has filled, a few hundred for thermal images).
#ifndef RIR_LOSSY_HIST_PX
#define RIR_LOSSY_HIST_PX 4096
#endif
	__global__ __launch_bounds__(1024) void lossy_hist_kernel(const uint16_t *__restrict__ tmp, int s, uint32_t *__restrict__ hist)
	{
		__shared__ uint32_t lh[16384];
		const int tid = threadIdx.x;
		for (int i = tid; i < 16384; i += 1024)
			lh[i] = 0;
		__syncthreads();
		const int i0 = blockIdx.x * RIR_LOSSY_HIST_PX, i1 = min(i0 + RIR_LOSSY_HIST_PX, s);
		for (int i = i0 + tid; i < i1; i += 1024)
			atomicAdd(&lh[tmp[i] >> 2], 1u);
		__syncthreads();
		for (int i = tid; i < 16384; i += 1024)
			if (lh[i])
				atomicAdd(&hist[i], lh[i]);
	}

	// mode of the histogram, lowest bin wins ties; stats[0] = background = (bin << 2) + 1
	__global__ __launch_bounds__(1024) void lossy_mode_kernel(uint32_t *__restrict__ hist, long long *__restrict__ stats)
	{
		__shared__ uint32_t best_v[1024];
		__shared__ uint32_t best_i[1024];
		const int tid = threadIdx.x;
		uint32_t bv = 0, bi = 0;
		for (int k = 0; k < 16; ++k)
		{ // contiguous range of 16 bins per thread, ascending: strict > keeps the lowest bin
			const uint32_t b = tid * 16 + k, v = hist[b];
			hist[b] = 0; // ready for the next frame (the histogram is cleared once, when the state is created)
			if (k == 0 || v > bv)
			{
				bv = v;
				bi = b;
			}
		}
		best_v[tid] = bv;
		best_i[tid] = bi;
		__syncthreads();
		for (int d = 512; d >= 1; d >>= 1)
		{
			if (tid < d)
			{
				const uint32_t ov = best_v[tid + d], oi = best_i[tid + d];
				if (ov > best_v[tid] || (ov == best_v[tid] && oi < best_i[tid]))
				{
					best_v[tid] = ov;
					best_i[tid] = oi;
				}
			}
			__syncthreads();
		}
		if (tid == 0)
			stats[0] = (long long)((best_i[0] << 2) + 1);
	}

	// L2: sums of |t - prev| and of its (32-bit wrapped) square, split by img > background.
	// stats[1..6] = {fg sum d, fg sum d2, fg count, bg sum d, bg sum d2, bg count}
	__global__ __launch_bounds__(256) void lossy_sums_kernel(const uint16_t *__restrict__ prevT, const uint16_t *__restrict__ tmp,
															  const uint16_t *__restrict__ img, int s, uint32_t mn, int subtract_min,
															  long long *__restrict__ stats)
	{
		const uint32_t background = (uint32_t)stats[0];
		long long a[6] = {0, 0, 0, 0, 0, 0};
		for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s; i += gridDim.x * blockDim.x)
		{
			const uint32_t t = subtract_min ? sub_min(tmp[i], mn) : tmp[i];
			const int32_t d = abs((int32_t)t - (int32_t)prevT[i]);
			const int32_t d2 = (int32_t)((uint32_t)d * (uint32_t)d);
			const int o = img[i] > background ? 0 : 3;
			a[o] += d;
			a[o + 1] += d2;
			a[o + 2] += 1;
		}
		// wave reduction, then the four waves of the block through LDS: one atomic per block and sum (a few hundred
		// in total - one per wave made 24 000 contended 64-bit atomics and cost 0.29 ms per frame)
		__shared__ long long red[4][6];
#pragma unroll
		for (int k = 0; k < 6; ++k)
		{
			long long v = a[k];
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1)
				v += __shfl_xor(v, d, 64);
			if ((threadIdx.x & 63) == 0)
				red[threadIdx.x >> 6][k] = v;
		}
		__syncthreads();
		if (threadIdx.x < 6)
		{
			const long long v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
			if (v)
				atomicAdd((unsigned long long *)&stats[1 + threadIdx.x], (unsigned long long)v);
		}
	}

	// The error budget of the frame (h264.cpp:2335-2385, :2544-2548 for addLoss): the reference's statistic
	// sqrt((sum d)^2 - sum d^2) / n of this frame against its mean over a 40-frame window, scaled by stdFactor, is
	// taken off lowValueError / highValueError.  Sequential scalar double arithmetic on exact integer sums - one
	// thread, same operations in the same order as the host code it replaces (sqrt, division and round of doubles
	// are IEEE-exact on the device: scripts/ubench/sqrt_f64_check.hip, 1.3e8 samples; this file is compiled with
	// -ffp-contract=off).  Keeping it here means no statistics travel to the host between the kernels of a frame.
	// (int) of a double as the reference's x86-64 build converts it (cvttsd2si): NaN and values outside int32 give
	// INT_MIN.  It matters: with an empty foreground or background the statistic is 0/0, the NaN stays in the 40-frame
	// window, and the budgets of those frames are whatever this conversion and the wrapping subtraction below make of it
	// (0 / 0 errors: lossless frames) - the device's own conversion (NaN -> 0) would leave the configured errors instead.
	__device__ __forceinline__ int int_of_double_x86(double v) { return (v >= -2147483648.0 && v < 2147483648.0) ? (int)v : (int)0x80000000; }
	__device__ __forceinline__ int sub_wrap(int a, int b) { return (int)((unsigned)a - (unsigned)b); }

	__global__ void lossy_budget_kernel(long long *__restrict__ stats, LossyBudget *__restrict__ bs, int s, int add_loss, double std_factor,
										int low_value_error, int high_value_error, LossyDecision *__restrict__ decision, int *__restrict__ errors_out)
	{
		if (threadIdx.x != 0 || blockIdx.x != 0)
			return;
		LossyBudget &b = *bs;
		// stdDev (h264.cpp:1993-2036): unsplit for the first 40 frames
		double sd[2];
		if (b.n_win < 40)
		{
			const double sum_diff = (double)(stats[1] + stats[4]), sum_diff2 = (double)(stats[2] + stats[5]);
			sd[0] = sd[1] = sqrt(sum_diff * sum_diff - sum_diff2) / s;
		}
		else
		{
			const double fd = (double)stats[1], fd2 = (double)stats[2], bd = (double)stats[4], bd2 = (double)stats[5];
			sd[0] = sqrt(bd * bd - bd2) / (int)stats[6];
			sd[1] = sqrt(fd * fd - fd2) / (int)stats[3];
		}
		if (b.n_first < 1)
		{
			b.first_std[0] = sd[0], b.first_std[1] = sd[1];
			b.n_first = 1;
		}
		if (b.n_win < 40)
		{
			b.win[b.n_win][0] = sd[0], b.win[b.n_win][1] = sd[1];
			++b.n_win;
		}
		else
		{
			for (int i = 0; i < 39; ++i)
				b.win[i][0] = b.win[i + 1][0], b.win[i][1] = b.win[i + 1][1];
			b.win[39][0] = sd[0], b.win[39][1] = sd[1];
		}
		double mean[2] = {b.first_std[0], b.first_std[1]};
		for (int i = 0; i < b.n_win; ++i)
		{
			mean[0] += b.win[i][0];
			mean[1] += b.win[i][1];
		}
		mean[0] /= (double)(b.n_win + b.n_first);
		mean[1] /= (double)(b.n_win + b.n_first);
		int low_error = low_value_error, high_error = high_value_error;
		if (add_loss)
		{ // one-sided
			const double dh = sd[1] < mean[1] ? 0 : sd[1] - mean[1], dl = sd[0] < mean[0] ? 0 : sd[0] - mean[0];
			high_error = sub_wrap(high_error, int_of_double_x86(round(dh * std_factor)));
			low_error = sub_wrap(low_error, int_of_double_x86(round(dl * std_factor)));
		}
		else
		{ // two-sided
			high_error = sub_wrap(high_error, int_of_double_x86(round(fabs(sd[1] - mean[1]) * std_factor)));
			low_error = sub_wrap(low_error, int_of_double_x86(round(fabs(sd[0] - mean[0]) * std_factor)));
		}
		if (high_error < 0)
			high_error = 0;
		if (low_error < high_error)
			low_error = high_error;
		decision->background = (uint32_t)stats[0];
		decision->low_error = low_error;
		decision->high_error = high_error;
		if (errors_out)
		{
			errors_out[0] = low_error;
			errors_out[1] = high_error;
		}
		for (int i = 1; i < 8; ++i) // the sums are accumulated with atomics: cleared for the next frame
			stats[i] = 0;
	}

	// L3 + L4: running average update and decision loop, one thread per pixel of the whole frame.
	__global__ __launch_bounds__(256) void lossy_update_kernel(const uint16_t *__restrict__ tmp, uint16_t *__restrict__ out, LossyDeviceState st,
																int s, int full, const LossyDecision *__restrict__ decision, int add_loss)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= full)
			return;
		const uint32_t background = decision->background;
		const int low_error = decision->low_error, high_error = decision->high_error;
		const uint32_t v = tmp[i];
		if (i >= s)
		{ // rows past lossy_height: stored as they are
			out[i] = (uint16_t)v;
			st.lastDL[i] = (uint16_t)v;
			return;
		}
		uint32_t t = st.subtract_min ? sub_min(v, st.min) : v;
		const int ra = st.running_average;
		uint32_t sum = 0;
		if (ra > 0)
		{ // RunningAverage2::addImage
			sum = st.ra_sums[i] + t;
			if (st.ra_count == ra)
			{
				const int16_t cc = st.ra_const_count[i];
				if (cc)
				{
					st.ra_const_count[i] = (int16_t)(cc - 1);
					sum -= st.ra_const_value[i];
				}
				else
					sum -= st.ra_images[(size_t)st.ra_head * s + i];
			}
			// the new image takes the free slot (ring not full) or replaces the oldest one
			const int slot = (st.ra_count == ra) ? st.ra_head : (st.ra_head + st.ra_count) % ra;
			st.ra_images[(size_t)slot * s + i] = (uint16_t)t;
		}
		const int n_after = ra > 0 ? (st.ra_count == ra ? ra : st.ra_count + 1) : 0; // images.size() after addImage
		const uint32_t ref = st.refT[i];
		const int diff = abs((int)t - (int)ref);
		const int max_error = v > background ? high_error : low_error;
		bool keep = diff <= max_error;
		if (!add_loss)
			keep = keep && ((st.lastDL[i] >> 13) == (v >> 13));
		if (keep)
			t = ra > 0 ? sum / (uint32_t)n_after : ref;
		else
		{
			st.refT[i] = (uint16_t)t;
			if (ra > 0)
			{
				st.ra_const_value[i] = (uint16_t)t;
				st.ra_const_count[i] = (int16_t)n_after;
				sum = t * (uint32_t)n_after;
			}
		}
		if (ra > 0)
			st.ra_sums[i] = sum;
		out[i] = (uint16_t)t;
		st.prevT[i] = (uint16_t)t;
		st.lastDL[i] = (uint16_t)v;
	}

	// first frame: out = tmp minus the optional minimum on rows < lossy_height; seeds refT / prevT / lastDL
	__global__ __launch_bounds__(256) void lossy_first_kernel(const uint16_t *__restrict__ tmp, uint16_t *__restrict__ out, LossyDeviceState st, int s,
															   int full)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= full)
			return;
		const uint32_t v = tmp[i];
		st.lastDL[i] = (uint16_t)v;
		uint32_t t = v;
		if (i < s)
		{
			if (st.subtract_min)
				t = sub_min(v, st.min);
			st.refT[i] = (uint16_t)t;
			st.prevT[i] = (uint16_t)t;
		}
		out[i] = (uint16_t)t;
	}

	// minimum of the first s pixels (subtractMin option)
	__global__ __launch_bounds__(256) void lossy_min_kernel(const uint16_t *__restrict__ tmp, int s, unsigned int *__restrict__ result)
	{
		unsigned int m = 65535;
		for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s; i += gridDim.x * blockDim.x)
			m = min(m, (unsigned int)tmp[i]);
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
			m = min(m, (unsigned int)__shfl_xor((int)m, d, 64));
		if ((threadIdx.x & 63) == 0)
			atomicMin(result, m);
	}

	// IRFileLoader::readImage (IRFileLoader.cpp:1173-1179): pixels[i] += min_T on the first min_T_height rows
	// (16-bit wrap-around like the reference's unsigned short +=), applied to every frame of a decoded chunk.
	__global__ void __launch_bounds__(256) lossy_add_min_kernel(uint16_t *__restrict__ frames, int64_t npx, int s, int nframes, uint32_t mn)
	{
		const int64_t total = (int64_t)s * nframes;
		for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
		{
			const int64_t f = i / s, p = i - f * s;
			frames[f * npx + p] = (uint16_t)(frames[f * npx + p] + mn);
		}
	}

	// d_hist (16 384 bins) and d_stats[1..7] must be zero on entry: they are when the state is created, and every frame
	// leaves them so (lossy_mode_kernel / lossy_budget_kernel clear what they have read).
	hipError_t launch_lossy_stats(const uint16_t *d_prevT, const uint16_t *d_tmp, const uint16_t *d_img, int s, uint32_t mn, int subtract_min,
								  uint32_t *d_hist, long long *d_stats, hipStream_t st)
	{
		hipLaunchKernelGGL(lossy_hist_kernel, dim3((s + RIR_LOSSY_HIST_PX - 1) / RIR_LOSSY_HIST_PX), dim3(1024), 0, st, d_tmp, s, d_hist);
		hipLaunchKernelGGL(lossy_mode_kernel, dim3(1), dim3(1024), 0, st, d_hist, d_stats);
		int blocks = (s + 2047) / 2048; // 8 pixels per thread
		if (blocks > 256)
			blocks = 256;
		hipLaunchKernelGGL(lossy_sums_kernel, dim3(blocks), dim3(256), 0, st, d_prevT, d_tmp, d_img, s, mn, subtract_min, d_stats);
		return hipGetLastError();
	}

	hipError_t launch_lossy_budget(long long *d_stats, LossyBudget *d_budget, int s, int add_loss, double std_factor, int low_value_error,
								   int high_value_error, LossyDecision *d_decision, int *d_errors_out, hipStream_t st)
	{
		hipLaunchKernelGGL(lossy_budget_kernel, dim3(1), dim3(1), 0, st, d_stats, d_budget, s, add_loss, std_factor, low_value_error, high_value_error,
						   d_decision, d_errors_out);
		return hipGetLastError();
	}

	hipError_t launch_lossy_update(const uint16_t *d_tmp, uint16_t *d_out, const LossyDeviceState &state, int s, int full,
								   const LossyDecision *d_decision, int add_loss, hipStream_t st)
	{
		hipLaunchKernelGGL(lossy_update_kernel, dim3((full + 255) / 256), dim3(256), 0, st, d_tmp, d_out, state, s, full, d_decision, add_loss);
		return hipGetLastError();
	}

	hipError_t launch_lossy_first(const uint16_t *d_tmp, uint16_t *d_out, const LossyDeviceState &state, int s, int full, hipStream_t st)
	{
		hipLaunchKernelGGL(lossy_first_kernel, dim3((full + 255) / 256), dim3(256), 0, st, d_tmp, d_out, state, s, full);
		return hipGetLastError();
	}

	hipError_t launch_lossy_min(const uint16_t *d_tmp, int s, unsigned int *d_result, hipStream_t st)
	{
		const unsigned int init = 65535;
		hipError_t e = hipMemcpyAsync(d_result, &init, sizeof(init), hipMemcpyHostToDevice, st);
		if (e != hipSuccess)
			return e;
		int blocks = (s + 255) / 256;
		if (blocks > 512)
			blocks = 512;
		hipLaunchKernelGGL(lossy_min_kernel, dim3(blocks), dim3(256), 0, st, d_tmp, s, d_result);
		return hipGetLastError();
	}
	hipError_t launch_lossy_add_min(uint16_t *d_frames, int64_t npx, int s, int nframes, uint32_t mn, hipStream_t st)
	{
		if (s <= 0 || nframes <= 0)
			return hipSuccess;
		int64_t blocks = ((int64_t)s * nframes + 255) / 256;
		if (blocks > 4096)
			blocks = 4096;
		hipLaunchKernelGGL(lossy_add_min_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_frames, npx, s, nframes, mn);
		return hipGetLastError();
	}
} // namespace rir
