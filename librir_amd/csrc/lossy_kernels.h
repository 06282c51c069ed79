// Device state and launchers of the bounded-loss step (lossy_kernels.hip).  Internal, C++ linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rir
{
	// Per-saver state living in HBM; reference members: H264_Saver::PrivateData refT, prevT, lastDL
	// (h264.cpp:1640-1660) and RunningAverage2 (h264.cpp:1526-1615).
	struct LossyDeviceState
	{
		uint16_t *refT, *prevT, *lastDL; // [w*h]
		uint32_t *ra_sums;				 // [s]
		uint16_t *ra_const_value;		 // [s]
		int16_t *ra_const_count;		 // [s]
		uint16_t *ra_images;			 // ring [running_average][s]
		int ra_count, ra_head, running_average;
		int subtract_min;
		uint32_t min;
	};

	// Error budget of the stream (h264.cpp:2335-2385): statistic of the first frame and of the last 40, kept in HBM.
	struct LossyBudget
	{
		double first_std[2];
		double win[40][2];
		int n_first, n_win;
	};
	// What lossy_budget_kernel decides for the current frame and lossy_update_kernel applies.
	struct LossyDecision
	{
		uint32_t background;
		int low_error, high_error;
		int reserved;
	};

	hipError_t launch_lossy_stats(const uint16_t *d_prevT, const uint16_t *d_tmp, const uint16_t *d_img, int s, uint32_t mn, int subtract_min,
								  uint32_t *d_hist, long long *d_stats, hipStream_t st);
	hipError_t launch_lossy_budget(long long *d_stats, LossyBudget *d_budget, int s, int add_loss, double std_factor, int low_value_error,
								   int high_value_error, LossyDecision *d_decision, int *d_errors_out, hipStream_t st);
	hipError_t launch_lossy_update(const uint16_t *d_tmp, uint16_t *d_out, const LossyDeviceState &state, int s, int full,
								   const LossyDecision *d_decision, int add_loss, hipStream_t st);
	hipError_t launch_lossy_first(const uint16_t *d_tmp, uint16_t *d_out, const LossyDeviceState &state, int s, int full, hipStream_t st);
	hipError_t launch_lossy_min(const uint16_t *d_tmp, int s, unsigned int *d_result, hipStream_t st);
	hipError_t launch_lossy_add_min(uint16_t *d_frames, int64_t npx, int s, int nframes, uint32_t mn, hipStream_t st);
} // namespace rir
