// Translation-only ECC image alignment on the device (ecc_kernels.hip).  Internal, C++ linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rir
{
#ifndef RIR_ECC_BLOCK
#define RIR_ECC_BLOCK 256
#endif
#ifndef RIR_ECC_MAX_BLOCKS
#define RIR_ECC_MAX_BLOCKS 256
#endif
	enum
	{
		ECC_NSUMS = 15,
		ECC_BLOCK = RIR_ECC_BLOCK,
		ECC_SOLVE_BLOCK = 256
	};
	// Lives in device memory; one per alignment in flight.
	struct EccState
	{
		float tx, ty;		   // current translation (map = [1 0 tx; 0 1 ty], applied as src = dst + t)
		double rho, last_rho;  // correlation coefficient of this / the previous iteration
		int iter;			   // iterations done
		int done;			   // 1 = converged or iteration limit, 2 = failed (empty overlap / lambda_d <= 0 / NaN)
		unsigned int ticket;   // blocks finished in the current iteration
		int max_iter;
		double eps;
	};
	// What the host needs of the state, in page-locked host memory the device writes directly (coherent): the host polls
	// `iter` / `done` instead of queueing a copy and waiting for the stream - a stream wait costs ~50 us of wake-up latency
	// per frame, the poll a few.
	struct EccHostView
	{
		float tx, ty;
		double rho;
		volatile int iter; // iterations finished (written after the fields above)
		volatile int done; // as EccState::done; 3: the resident launch was called off before it did anything (resident_device.h)
		volatile unsigned int progress; // moves while a resident launch is alive (every image of a sequence, every 64 iterations of an alignment):
										// the host gives up on lack of progress, not on total time - a launch may legitimately run for minutes
		unsigned int pad;
	};
	size_t ecc_workspace_bytes(int w, int h);
	hipError_t launch_ecc_prepare(const float *d_image, int w, int h, float *d_gx, float *d_gy, EccState *d_state, float tx, float ty, int max_iter,
								  double eps, hipStream_t st);
	// one iteration (no-op once d_state->done != 0)
	// host_view: device-visible address of an EccHostView, or NULL
	hipError_t launch_ecc_iterate(const float *d_templ, const float *d_image, const float *d_gx, const float *d_gy, const uint8_t *d_mask, int w,
								  int h, double *d_partials, EccState *d_state, EccHostView *host_view, hipStream_t st);
	// all iterations of the alignments of `nframes` consecutive images [nframes][h][w] (gradients likewise) in one launch
	// (ecc_run_kernel), each starting from the previous result; d_results: NULL or [nframes]; d_rows: ecc_run_workspace_bytes(); epoch: a number no earlier launch on this workspace used
	// what a sequence launch leaves per image
	struct EccFrameResult
	{
		float tx, ty;
		double rho;
		int iter, done;
	};
	// one tracked sequence of a multi-sequence launch (ecc_run_multi_kernel): device pointers, plain data
	struct EccSeq
	{
		const float *templ;			  // [h][w] the sequence's normalised reference window
		const float *image, *gx, *gy; // [nframes][h][w] prepared images and their gradients
		double *rows;				  // ecc_run_workspace_bytes(w, h) of the sequence's own: rows of granules, then its pub granule
		EccFrameResult *results;	  // [nframes]
		float tx0, ty0;				  // start value of the first alignment
		int nframes, frames_done;	  // frames_done: left by the kernel (images gone through; the last of them may have failed)
	};
	int ecc_run_multi_capacity(); // resident workgroups of ecc_run_multi_kernel on the current device (runtime.h), 0 = unknown
	int ecc_rows(int w, int h);	  // rows of partial sums of an alignment of a w x h window (= workgroups of a solo run)
	// d_ctl: two zero-initialised words (resident_device.h: arrivals, decision); arrivals_before: workgroups of the earlier launches on d_ctl.
	// After the launch d_ctl[1] == (epoch & 0x3fffffff) << 2 | RESIDENT_GO, or | RESIDENT_BAIL: the launch did not become resident and has written nothing.
	int ecc_run_multi_grid(int nseq, int nslices); // workgroups of such a launch: a service workgroup per sequence + nslices compute workgroups per pair
	// host_go: NULL, or a word of coherent page-locked host memory that receives `epoch` once the launch is resident (from then on
	// other kernels may be started beside it)
	hipError_t launch_ecc_run_multi(EccSeq *d_table, int nseq, int nslices, int w, int h, int max_iter, double eps, unsigned int epoch, unsigned int *d_ctl,
									unsigned int arrivals_before, unsigned int *host_go, hipStream_t st);
	size_t ecc_run_workspace_bytes(int w, int h);
	int ecc_run_capacity();			  // resident workgroups of ecc_run_kernel on the current device (runtime.h), 0 = unknown
	bool ecc_run_fits(int w, int h);  // the grid of an alignment of a w x h window fits: the one-launch forms may be used
	hipError_t launch_ecc_run(const float *d_templ, const float *d_image, const float *d_gx, const float *d_gy, const uint8_t *d_mask, int w, int h,
							  double *d_rows, EccState *d_state, EccHostView *host_view, float tx, float ty, int max_iter, double eps, unsigned int epoch,
							  int nframes, EccFrameResult *d_results, unsigned int *d_ctl, unsigned int arrivals_before, hipStream_t st);
	int ecc_run_grid(int w, int h); // workgroups of such a launch
	// (d_ctl, arrivals_before: the residency control block, resident_device.h; a launch that was called off reports done = 3 through the host view)
	constexpr int kEccMaxSequence = 4096, kEccMaxIterations = (1 << 20) - 1; // (the flag's fields)
	constexpr int kMinMaxParts = 256, kMinMaxPartsFrames = 64; // partial (min, max) pairs per image: one image / a batch of images
	// d_part: 2 * kMinMaxParts floats (one image), nframes * 2 * kMinMaxPartsFrames floats (a batch)
	hipError_t launch_minmax_normalize(const float *d_src, int w, int h, int src_stride, float *d_dst, float *d_part, hipStream_t st);
	// the same, and the gradients of the normalised images (central differences, reflected borders), in one pass over the source;
	// d_part: nframes * 2 * (nframes == 1 ? kMinMaxParts : kMinMaxPartsFrames) floats
	hipError_t launch_minmax_normalize_grad_frames(const float *d_src, int w, int h, int src_stride, int64_t src_frame, int nframes, float *d_dst, float *d_gx,
												   float *d_gy, float *d_part, hipStream_t st);
} // namespace rir
