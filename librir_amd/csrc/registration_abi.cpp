// C ABI of the registration step: translation-only ECC alignment on the device.
//
// The reference computes sub-pixel translations in Python by calling OpenCV
// (src/python/librir/registration/masked_registration_ecc.py:166-168, cv2.findTransformECC with
// MOTION_TRANSLATION); these two entry points are what that call binds to here.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ecc_kernels.h"
#include "filter_kernels.h"
#include "rir_amd_device.h"
#include "runtime.h"

// the alignment kernels address an image through 32-bit byte offsets of one buffer descriptor and 24-bit row numbers (ecc_kernels.hip)
static inline bool ecc_size_ok(int w, int h) { return w < (1 << 23) && h < (1 << 23) && (long long)w * h <= (1ll << 29); }

using namespace rir;

namespace
{
	struct EccScratch
	{
		std::mutex mu;
		DeviceBuffer gx, gy, partials, state, templ, image, mask, mm, full, mm_frames, full_frames;
		EccHostView *view = nullptr; // coherent page-locked host memory (64 bytes, kept for the life of the process), written by
									  // ecc_solve_kernel, polled by run_ecc
		// run_ecc returns as soon as the host view says "done", with the rest of its batch (no-op launches that still READ the
		// shared state) queued on the caller's stream: the next call - possibly on another stream - waits for this event
		// before it resets the state
		hipEvent_t tail = nullptr;
		bool tail_recorded = false;
		unsigned int epoch = 0; // launches of ecc_run_kernel on this workspace
		EccFrameResult *results_host = nullptr; // coherent page-locked host memory, kEccMaxSequence entries, kept for the life of the process
		// multi-sequence launches: per sequence its rows of granules, its results; the table of sequences and its page-locked copy
		DeviceBuffer multi_rows, multi_results, multi_table, multi_ctl;
		PinnedBuffer multi_stage, multi_back;
		unsigned int multi_arrivals = 0; // workgroups launched on multi_ctl so far (resident_device.h)
		unsigned int *multi_go = nullptr; // coherent page-locked host word: "the multi-sequence launch is resident" (its epoch)
		hipStream_t side = nullptr;		  // the stream of the pre-processing that runs beside a resident launch
		hipEvent_t side_in = nullptr, side_out = nullptr;
		DeviceBuffer run_ctl;			 // the same for the single-sequence launches
		unsigned int run_arrivals = 0;
		unsigned int *ctl_of_runs(hipStream_t st)
		{ // two zeroed words, allocated once
			if (run_ctl.cap)
				return run_ctl.as<unsigned int>();
			if (!run_ctl.reserve(256) || hipMemsetAsync(run_ctl.ptr, 0, 256, st) != hipSuccess)
				return nullptr;
			run_arrivals = 0;
			return run_ctl.as<unsigned int>();
		}
	};
	// one scratch per DEVICE (its buffers, events, page-locked words and side stream belong to the device that was current when they were
	// made: a process that drives two GPUs must not use one device's on the other's stream); the current device's is returned
	constexpr int kScratchDevices = 64;
	int scratch_slot()
	{
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kScratchDevices)
			dev = 0;
		return dev;
	}
	EccScratch &scratch()
	{
		static EccScratch *s = new EccScratch[kScratchDevices]; // (never destroyed: the runtime may be gone before static destructors run)
		return s[scratch_slot()];
	}
	// (tests, RIR_DEBUG_ECC_BAIL: as if the next launch on this control block had not become resident - its decision says BAIL before anybody arrives)
	bool debug_call_off(unsigned int *d_ctl, unsigned int epoch, hipStream_t st)
	{
		if (!test_hook("RIR_DEBUG_ECC_BAIL"))
			return true;
		const unsigned int w_ = ((epoch & 0x3fffffffu) << 2) | 2u;
		return hip_ok(hipMemcpyAsync(d_ctl + 1, &w_, 4, hipMemcpyHostToDevice, st), "H2D") && hip_ok(hipStreamSynchronize(st), "sync");
	}
	// The pre-processing of a chunk of frames (rir_ecc_prepare_frames_device) has a scratch and a stream order of its own: it shares
	// nothing with the alignments, so a caller may run the pre-processing of chunk k + 1 on a second stream under the alignments
	// of chunk k (DeviceRegistratorECC.compute_many_multi).  Only its own calls are ordered among themselves.
	struct PrepScratch
	{
		std::mutex mu;
		DeviceBuffer mm_frames, full_frames;
		hipEvent_t tail = nullptr;
		bool tail_recorded = false;
	};
	PrepScratch &prep_scratch()
	{
		static PrepScratch *s = new PrepScratch[kScratchDevices]; // (per device, as above)
		return s[scratch_slot()];
	}

	// Stream ordering of the shared scratch (the host mutex only orders the CALLS): constructed, under the mutex, before an
	// entry point queues anything that touches the scratch; makes `st` wait for everything the previous call queued and, on
	// every exit, leaves the event behind the last launch of this call.
	struct ScratchOrder
	{
		EccScratch &sc;
		hipStream_t st;
		bool ok = true;
		ScratchOrder(EccScratch &s, hipStream_t stream) : sc(s), st(stream)
		{
			if (!sc.tail && !hip_ok(hipEventCreateWithFlags(&sc.tail, hipEventDisableTiming), "hipEventCreate"))
				ok = false;
			else if (sc.tail_recorded && !hip_ok(hipStreamWaitEvent(st, sc.tail, 0), "hipStreamWaitEvent"))
				ok = false;
		}
		~ScratchOrder()
		{
			if (sc.tail)
				sc.tail_recorded = hipEventRecord(sc.tail, st) == hipSuccess;
		}
	};

#ifndef RIR_ECC_FIRST_BATCH
#define RIR_ECC_FIRST_BATCH 6 /* alignments of a tracked sequence settle within 4-6 iterations: one read-back of the state instead of two */
#endif
	// waits (polling the coherent view) until the alignment in flight reports done or `launched` iterations
	bool wait_view(EccHostView *view, int launched, hipStream_t st)
	{
		auto t0 = std::chrono::steady_clock::now();
		const auto t_begin = t0;
		long spins = 0;
		unsigned int seen = view->progress;
		while (view->done == 0 && view->iter < launched)
		{
			if ((++spins & 0x3ff) == 0)
			{
				// the clock is one of NO PROGRESS, not of total time: a resident launch may hold thousands of images with up to a million
				// iterations each and moves `progress` at least every 64 iterations; the launch-per-iteration batches move `iter`
				const unsigned int p = view->progress;
				if (p != seen)
					seen = p, t0 = std::chrono::steady_clock::now();
				else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
				{
					// No word for 10 s.  The launch may not have STARTED: it queues behind the previous resident launch of the device
					// (ResidentGate), and another thread's may legitimately run for minutes.  The stream tells the two cases apart: while
					// it still has work the wait goes on (every wait inside the kernels is bounded, so a launch that started ends); a
					// stream that has drained without the words having been written is a device that did not report back.
					if (hipStreamQuery(st) == hipErrorNotReady && std::chrono::steady_clock::now() - t_begin < std::chrono::minutes(30))
					{
						t0 = std::chrono::steady_clock::now();
						continue;
					}
					(void)hipGetLastError();
					log_error("ECC: the device did not report back");
					(void)hipStreamSynchronize(st);
					return false;
				}
			}
			__builtin_ia32_pause();
		}
		std::atomic_thread_fence(std::memory_order_acquire);
		return true;
	}

	// The iterations of one alignment of d_image (its gradients d_gx, d_gy already computed) against d_templ; the caller holds the
	// scratch mutex.  One launch does them all (ecc_run_kernel); RIR_ECC_LAUNCH_PER_ITERATION=1 queues two launches per iteration
	// in batches instead (the round-1 form, kept for comparison: same sums in the same order, same results).
	int run_iterations(EccScratch &sc, const float *d_templ, const float *d_image, const float *d_gx, const float *d_gy, const uint8_t *d_mask, int w, int h,
					   float *warp, int max_iter, double eps, double *cc, int *iterations, bool state_is_reset, hipStream_t st)
	{
		if (!sc.partials.reserve(std::max(ecc_workspace_bytes(w, h), ecc_run_workspace_bytes(w, h))) || !sc.state.reserve(sizeof(EccState)))
			return -1;
		EccState *d_state = sc.state.as<EccState>();
		if (!sc.view && !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sc.view), sizeof(EccHostView), hipHostMallocCoherent | hipHostMallocMapped),
								"hipHostMalloc"))
			return -1;
		EccHostView *view = sc.view;
		view->iter = 0, view->done = 0; // (host writes before the launches below: ordered by the submission)
		EccHostView *d_view = nullptr;
		if (!hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_view), view, 0), "hipHostGetDevicePointer"))
			return -1;
		EccState hs;
		std::memset(&hs, 0, sizeof(hs));
		// one launch for all iterations when the device holds the alignment's workgroups at once (runtime.h: resident launches),
		// two launches per iteration otherwise - same sums in the same order, same results
		static const bool env_per_iteration = getenv("RIR_ECC_LAUNCH_PER_ITERATION") != nullptr;
		const bool per_iteration = env_per_iteration || !ecc_run_fits(w, h);
		if (max_iter > kEccMaxIterations)
		{
			log_error("ECC: max_iterations exceeds 1 048 575");
			return -1;
		}
		bool called_off = false;
		if (!per_iteration)
		{
			unsigned int *d_ctl = sc.ctl_of_runs(st);
			if (!d_ctl)
				return -1;
			if (!debug_call_off(d_ctl, sc.epoch + 1, st))
				return -1;
			if (!hip_ok(launch_ecc_run(d_templ, d_image, d_gx, d_gy, d_mask, w, h, sc.partials.as<double>(), d_state, d_view, warp[0], warp[1], max_iter, eps,
									   ++sc.epoch, 1, nullptr, d_ctl, sc.run_arrivals, st),
						"ecc run"))
				return -1;
			sc.run_arrivals += (unsigned int)ecc_run_grid(w, h); // (counted once the launch is queued: its workgroups will arrive)
			if (!wait_view(view, max_iter + 1, st))
				return -1;
			hs.done = view->done, hs.iter = view->iter, hs.tx = view->tx, hs.ty = view->ty, hs.rho = view->rho;
			called_off = hs.done == 3; // the launch did not become resident (resident_device.h) and has computed nothing: two launches per iteration instead
			if (called_off)
			{
				if (!hip_ok(hipStreamSynchronize(st), "sync"))
					return -1;
				view->iter = 0, view->done = 0;
				state_is_reset = false;
			}
			static const bool diag = getenv("RIR_ECC_DIAG") != nullptr; // (-DRIR_ECC_DIAG builds: where an iteration's time goes)
			if (diag && (sc.epoch % 64) == 0)
			{
				unsigned long long dg[16];
				const size_t off = ecc_run_workspace_bytes(w, h) - 256 + 64;
				if (hipMemcpy(dg, sc.partials.as<char>() + off, sizeof(dg), hipMemcpyDeviceToHost) == hipSuccess && dg[3])
					std::fprintf(stderr, "ecc run, per iteration (us): workgroup 0: sums+publish %.2f  wait rows %.2f  add+solve+publish %.2f | last workgroup: sums+publish %.2f  wait %.2f  (%.1f iterations per frame) | workgroup 0: pixel loop %.2f  block reduce %.2f\n",
								 dg[0] * 0.01 / dg[3], dg[1] * 0.01 / dg[3], dg[2] * 0.01 / dg[3], dg[8] * 0.01 / dg[11], dg[9] * 0.01 / dg[11], (double)dg[3] / sc.epoch,
								 dg[4] * 0.01 / dg[3], dg[5] * 0.01 / dg[3]);
			}
		}
		if (per_iteration || called_off)
		{
			if (!state_is_reset)
			{
				EccState init;
				std::memset(&init, 0, sizeof(init));
				init.tx = warp[0], init.ty = warp[1], init.rho = -1.0, init.last_rho = -eps, init.max_iter = max_iter, init.eps = eps;
				if (!hip_ok(hipMemcpyAsync(d_state, &init, sizeof(init), hipMemcpyHostToDevice, st), "H2D") || !hip_ok(hipStreamSynchronize(st), "sync"))
					return -1;
			}
			// iterations are queued in batches (a finished alignment turns the remaining launches into no-ops); the host
			// polls the view until the alignment is done or the batch is through
			int launched = 0;
			while (true)
			{
				const int batch = std::min(launched == 0 ? RIR_ECC_FIRST_BATCH : 8, max_iter - launched);
				for (int i = 0; i < batch; ++i)
					if (!hip_ok(launch_ecc_iterate(d_templ, d_image, d_gx, d_gy, d_mask, w, h, sc.partials.as<double>(), d_state, d_view, st), "ecc iterate"))
						return -1;
				launched += batch;
				if (!wait_view(view, launched, st))
					return -1;
				hs.done = view->done, hs.iter = view->iter, hs.tx = view->tx, hs.ty = view->ty, hs.rho = view->rho;
				if (hs.done || launched >= max_iter)
					break;
			}
		}
		if (iterations)
			*iterations = hs.iter;
		if (hs.done == 2 || std::isnan(hs.rho))
		{
			log_error("ECC: the alignment did not converge (empty overlap, singular system or non-positive lambda)");
			return -1;
		}
		warp[0] = hs.tx, warp[1] = hs.ty;
		if (cc)
			*cc = hs.rho;
		return 0;
	}

	// gradients of the image, then the iterations; the caller holds the scratch mutex
	int run_ecc(EccScratch &sc, const float *d_templ, const float *d_image, const uint8_t *d_mask, int w, int h, float *warp, int max_iter,
				double eps, double *cc, int *iterations, hipStream_t st)
	{
		const size_t npx = (size_t)w * h;
		if (!sc.gx.reserve(npx * 4) || !sc.gy.reserve(npx * 4) || !sc.state.reserve(sizeof(EccState)))
			return -1;
		if (!hip_ok(launch_ecc_prepare(d_image, w, h, sc.gx.as<float>(), sc.gy.as<float>(), sc.state.as<EccState>(), warp[0], warp[1], max_iter, eps, st),
					"ecc prepare"))
			return -1;
		return run_iterations(sc, d_templ, d_image, sc.gx.as<float>(), sc.gy.as<float>(), d_mask, w, h, warp, max_iter, eps, cc, iterations, true, st);
	}
} // namespace

// d_templ, d_image: float [h][w] in device memory; d_mask: uint8 [h][w] or NULL; warp: HOST float[2] = (tx, ty), in/out
// (start value -> result; the aligned image is image(x + tx, y + ty) ~ templ(x, y)); *cc = correlation coefficient.
RIR_EXPORT int rir_ecc_translation_device(const float *d_templ, const float *d_image, const unsigned char *d_mask, int w, int h, float *warp,
										  int max_iterations, double eps, double *cc, int *iterations, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_templ || !d_image || !warp || w < 2 || h < 2 || !ecc_size_ok(w, h) || max_iterations <= 0 || !(eps >= 0))
	{
		log_error("rir_ecc_translation_device: invalid argument");
		return -1;
	}
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	ScratchOrder order(sc, (hipStream_t)stream);
	if (!order.ok)
		return -1;
	return run_ecc(sc, d_templ, d_image, d_mask, w, h, warp, max_iterations, eps, cc, iterations, (hipStream_t)stream);
}

// One frame of a tracked sequence, from the raw frame to the shift, in one call: gaussian pre-filter (sigma > 0), min-max
// normalisation of the registration window [win_y, win_y + win_h) x [win_x, win_x + win_w) and the alignment against the
// (already normalised) reference window d_ref_norm [win_h][win_w] - the steps of MaskedRegistratorECC.compute
// (masked_registration_ecc.py:88-168) queued back to back, with one read-back at the end.  d_img: uint16 (dtype 'H') or
// float32 ('f') frame [h][w] in device memory.  warp: HOST float[2] (tx, ty), start value in, result out.
RIR_EXPORT int rir_ecc_register_frame_device(const void *d_img, int dtype, int w, int h, float sigma, int win_x, int win_y, int win_w, int win_h,
											 const float *d_ref_norm, float *warp, int max_iterations, double eps, double *cc, int *iterations,
											 void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_img || (dtype != 'H' && dtype != 'f') || !d_ref_norm || !warp || w < 2 || h < 2 || win_x < 0 || win_y < 0 || win_w < 2 || win_h < 2 ||
		win_x + win_w > w || win_y + win_h > h || !ecc_size_ok(w, h) || !ecc_size_ok(win_w, win_h) || max_iterations <= 0 || !(eps >= 0))
	{
		log_error("rir_ecc_register_frame_device: invalid argument");
		return -1;
	}
	hipStream_t st = (hipStream_t)stream;
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	ScratchOrder order(sc, st);
	if (!order.ok)
		return -1;
	if (!sc.full.reserve((size_t)w * h * 4) || !sc.image.reserve((size_t)win_w * win_h * 4) || !sc.mm.reserve(2 * kMinMaxParts * sizeof(float)))
		return -1;
	const float *g = nullptr;
	if (sigma > 0)
	{
		const int r = dtype == 'H' ? rir_gaussian_filter_u16_device(static_cast<const unsigned short *>(d_img), sc.full.as<float>(), w, h, 1, sigma, stream)
								   : rir_gaussian_filter_device(static_cast<const float *>(d_img), sc.full.as<float>(), w, h, 1, sigma, stream);
		if (r != 0)
			return -1;
		g = sc.full.as<float>();
	}
	else if (dtype == 'f')
		g = static_cast<const float *>(d_img);
	else
	{
		if (!hip_ok(launch_u16_to_f32(static_cast<const uint16_t *>(d_img), sc.full.as<float>(), (int64_t)w * h, st), "u16_to_f32"))
			return -1;
		g = sc.full.as<float>();
	}
	// normalisation and gradients in one pass, then all iterations in one launch
	const size_t wpx = (size_t)win_w * win_h;
	if (!sc.gx.reserve(wpx * 4) || !sc.gy.reserve(wpx * 4) ||
		!hip_ok(launch_minmax_normalize_grad_frames(g + (size_t)win_y * w + win_x, win_w, win_h, w, 0, 1, sc.image.as<float>(), sc.gx.as<float>(), sc.gy.as<float>(),
													sc.mm.as<float>(), st),
				"minmax_normalize"))
		return -1;
	return run_iterations(sc, d_ref_norm, sc.image.as<float>(), sc.gx.as<float>(), sc.gy.as<float>(), nullptr, win_w, win_h, warp, max_iterations, eps, cc,
						  iterations, false, st);
}

// The pre-processing of `nframes` frames of a tracked sequence in shared launches, ahead of their alignments (which are
// sequential: each starts from the previous result): gaussian pre-filter (sigma > 0), crop to the registration window, min-max
// normalisation, gradients - image by image the operations of rir_ecc_register_frame_device.  d_imgs: uint16 ('H') or float32
// ('f') [nframes][h][w]; d_norm, d_gx, d_gy: float [nframes][win_h][win_w] (device).
static bool prepare_args_ok(const void *d_imgs, int dtype, int w, int h, int nframes, int win_x, int win_y, int win_w, int win_h, const float *d_norm,
							const float *d_gx, const float *d_gy)
{
	return d_imgs && (dtype == 'H' || dtype == 'f') && d_norm && d_gx && d_gy && w >= 2 && h >= 2 && nframes > 0 && win_x >= 0 && win_y >= 0 && win_w >= 2 &&
		   win_h >= 2 && win_x + win_w <= w && win_y + win_h <= h;
}
static int prepare_frames_on(const void *d_imgs, int dtype, int w, int h, int nframes, float sigma, int win_x, int win_y, int win_w, int win_h, float *d_norm,
							 float *d_gx, float *d_gy, hipStream_t st);
RIR_EXPORT int rir_ecc_prepare_frames_device(const void *d_imgs, int dtype, int w, int h, int nframes, float sigma, int win_x, int win_y, int win_w,
											 int win_h, float *d_norm, float *d_gx, float *d_gy, void *stream)
{
	if (!device_ready())
		return -1;
	if (!prepare_args_ok(d_imgs, dtype, w, h, nframes, win_x, win_y, win_w, win_h, d_norm, d_gx, d_gy))
	{
		log_error("rir_ecc_prepare_frames_device: invalid argument");
		return -1;
	}
	return prepare_frames_on(d_imgs, dtype, w, h, nframes, sigma, win_x, win_y, win_w, win_h, d_norm, d_gx, d_gy, (hipStream_t)stream);
}
static int prepare_frames_on(const void *d_imgs, int dtype, int w, int h, int nframes, float sigma, int win_x, int win_y, int win_w, int win_h, float *d_norm,
							 float *d_gx, float *d_gy, hipStream_t st)
{
	void *stream = (void *)st;
	PrepScratch &sc = prep_scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	// (stream order among the calls that share this scratch: wait for the previous call's last launch, leave an event behind this one's)
	if (!sc.tail && !hip_ok(hipEventCreateWithFlags(&sc.tail, hipEventDisableTiming), "hipEventCreate"))
		return -1;
	if (sc.tail_recorded && !hip_ok(hipStreamWaitEvent(st, sc.tail, 0), "hipStreamWaitEvent"))
		return -1;
	struct Leave
	{
		PrepScratch &p;
		hipStream_t s;
		~Leave() { p.tail_recorded = hipEventRecord(p.tail, s) == hipSuccess; }
	} leave{sc, st};
	const size_t npx = (size_t)w * h;
	if (!sc.mm_frames.reserve((size_t)nframes * 2 * std::max(kMinMaxParts, kMinMaxPartsFrames) * sizeof(float)))
		return -1;
	const float *g = nullptr;
	if (sigma > 0 || dtype == 'H')
	{
		if (!sc.full_frames.reserve(npx * nframes * 4))
			return -1;
		float *full = sc.full_frames.as<float>();
		if (sigma > 0)
		{
			const int r = dtype == 'H' ? rir_gaussian_filter_u16_device(static_cast<const unsigned short *>(d_imgs), full, w, h, nframes, sigma, stream)
									   : rir_gaussian_filter_device(static_cast<const float *>(d_imgs), full, w, h, nframes, sigma, stream);
			if (r != 0)
				return -1;
		}
		else if (!hip_ok(launch_u16_to_f32(static_cast<const uint16_t *>(d_imgs), full, (int64_t)npx * nframes, st), "u16_to_f32"))
			return -1;
		g = full;
	}
	else
		g = static_cast<const float *>(d_imgs);
	if (!hip_ok(launch_minmax_normalize_grad_frames(g + (size_t)win_y * w + win_x, win_w, win_h, w, (int64_t)npx, nframes, d_norm, d_gx, d_gy,
													sc.mm_frames.as<float>(), st),
				"minmax_normalize"))
		return -1;
	return 0;
}

// The alignment of ONE prepared image (d_norm and its gradients, [h][w] each: one frame of rir_ecc_prepare_frames_device's output)
// against the reference window d_ref_norm.  warp: HOST float[2] (tx, ty), start value in, result out.
RIR_EXPORT int rir_ecc_align_prepared_device(const float *d_ref_norm, const float *d_norm, const float *d_gx, const float *d_gy, int w, int h, float *warp,
											 int max_iterations, double eps, double *cc, int *iterations, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_ref_norm || !d_norm || !d_gx || !d_gy || !warp || w < 2 || h < 2 || !ecc_size_ok(w, h) || max_iterations <= 0 || !(eps >= 0))
	{
		log_error("rir_ecc_align_prepared_device: invalid argument");
		return -1;
	}
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	ScratchOrder order(sc, (hipStream_t)stream);
	if (!order.ok)
		return -1;
	return run_iterations(sc, d_ref_norm, d_norm, d_gx, d_gy, nullptr, w, h, warp, max_iterations, eps, cc, iterations, false, (hipStream_t)stream);
}

// The alignments of `nframes` consecutive prepared images (rir_ecc_prepare_frames_device's output) in ONE launch, image i starting
// from the result of image i - 1 (image 0 from warp), as MaskedRegistratorECC.compute does frame after frame.  results: HOST
// [nframes][4] doubles = (tx, ty, correlation coefficient, iterations) per image.  Returns the number of images aligned: nframes,
// or the index of the first one whose alignment failed (where the per-image entry point returns -1; its row and the later ones
// are not filled), or -1 on an error of the call itself.  warp: HOST float[2], start value in, last good result out.
namespace
{
	int align_frames_locked(EccScratch &sc, const float *d_ref_norm, const float *d_norm, const float *d_gx, const float *d_gy, int w, int h, int nframes,
							float *warp, int max_iterations, double eps, double *results, hipStream_t st);
}
RIR_EXPORT int rir_ecc_align_prepared_frames_device(const float *d_ref_norm, const float *d_norm, const float *d_gx, const float *d_gy, int w, int h,
													int nframes, float *warp, int max_iterations, double eps, double *results, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_ref_norm || !d_norm || !d_gx || !d_gy || !warp || !results || w < 2 || h < 2 || !ecc_size_ok(w, h) || nframes <= 0 || nframes > kEccMaxSequence || max_iterations <= 0 ||
		max_iterations > kEccMaxIterations || !(eps >= 0))
	{
		log_error("rir_ecc_align_prepared_frames_device: invalid argument");
		return -1;
	}
	hipStream_t st = (hipStream_t)stream;
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	ScratchOrder order(sc, st);
	if (!order.ok)
		return -1;
	return align_frames_locked(sc, d_ref_norm, d_norm, d_gx, d_gy, w, h, nframes, warp, max_iterations, eps, results, st);
}
namespace
{
int align_frames_locked(EccScratch &sc, const float *d_ref_norm, const float *d_norm, const float *d_gx, const float *d_gy, int w, int h, int nframes,
						float *warp, int max_iterations, double eps, double *results, hipStream_t st)
{
	if (!sc.partials.reserve(std::max(ecc_workspace_bytes(w, h), ecc_run_workspace_bytes(w, h))) || !sc.state.reserve(sizeof(EccState)))
		return -1;
	static const bool env_per_iteration = getenv("RIR_ECC_LAUNCH_PER_ITERATION") != nullptr;
	// image by image through run_iterations (one launch per alignment where it becomes resident, else two launches per iteration): same results
	auto image_by_image = [&]() {
		const size_t wpx = (size_t)w * h;
		int good = 0;
		for (; good < nframes; ++good)
		{
			double cc = 0;
			int iters = 0;
			float t[2] = {warp[0], warp[1]};
			if (run_iterations(sc, d_ref_norm, d_norm + good * wpx, d_gx + good * wpx, d_gy + good * wpx, nullptr, w, h, t, max_iterations, eps, &cc, &iters, false,
							   st) != 0)
				break;
			warp[0] = t[0], warp[1] = t[1];
			results[4 * good] = t[0], results[4 * good + 1] = t[1], results[4 * good + 2] = cc, results[4 * good + 3] = iters;
		}
		return good;
	};
	if (env_per_iteration || !ecc_run_fits(w, h))
		return image_by_image(); // the device does not hold the run kernel's workgroups at once
	// the host view and the per-image results live in one block of coherent page-locked host memory the kernel writes directly (a
	// copy of the results and the stream synchronisation behind it cost more than a chunk's book-keeping)
	if (!sc.view && !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sc.view), sizeof(EccHostView), hipHostMallocCoherent | hipHostMallocMapped), "hipHostMalloc"))
		return -1;
	if (!sc.results_host &&
		!hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sc.results_host), (size_t)kEccMaxSequence * sizeof(EccFrameResult), hipHostMallocCoherent | hipHostMallocMapped),
				"hipHostMalloc"))
		return -1;
	EccHostView *view = sc.view;
	view->iter = 0, view->done = 0;
	EccHostView *d_view = nullptr;
	EccFrameResult *d_results = nullptr;
	if (!hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_view), view, 0), "hipHostGetDevicePointer") ||
		!hip_ok(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_results), sc.results_host, 0), "hipHostGetDevicePointer"))
		return -1;
	unsigned int *d_ctl = sc.ctl_of_runs(st);
	if (!d_ctl)
		return -1;
	if (!debug_call_off(d_ctl, sc.epoch + 1, st))
		return -1;
	if (!hip_ok(launch_ecc_run(d_ref_norm, d_norm, d_gx, d_gy, nullptr, w, h, sc.partials.as<double>(), sc.state.as<EccState>(), d_view, warp[0], warp[1],
							   max_iterations, eps, ++sc.epoch, nframes, d_results, d_ctl, sc.run_arrivals, st),
				"ecc run"))
		return -1;
	sc.run_arrivals += (unsigned int)ecc_run_grid(w, h);
	if (!wait_view(view, nframes + 1, st))
		return -1;
	if (view->done == 3)
	{ // the launch did not become resident (resident_device.h) and has aligned nothing
		if (!hip_ok(hipStreamSynchronize(st), "sync"))
			return -1;
		return image_by_image();
	}
	const int frames_done = view->iter; // images gone through (the last of them may have failed)
	const EccFrameResult *r = sc.results_host; // (written before the view's release store, read after its acquire)
	int good = 0;
	for (; good < frames_done; ++good)
	{
		if (r[good].done == 2 || std::isnan(r[good].rho))
			break;
		results[4 * good] = r[good].tx, results[4 * good + 1] = r[good].ty, results[4 * good + 2] = r[good].rho, results[4 * good + 3] = r[good].iter;
		warp[0] = r[good].tx, warp[1] = r[good].ty;
	}
	if (good < nframes)
		log_error("ECC: the alignment did not converge (empty overlap, singular system or non-positive lambda)");
	return good;
}
} // namespace

// The alignments of `nseq` INDEPENDENT tracked sequences in shared resident launches (ecc_run_multi_kernel): sequence q aligns its
// nframes[q] prepared images (d_norm[q], d_gx[q], d_gy[q]: [nframes[q]][h][w], rir_ecc_prepare_frames_device's output) against its
// reference window d_ref_norm[q], image i from the result of image i - 1, image 0 from warps[2q], warps[2q + 1] - for every
// sequence the same operations in the same order as rir_ecc_align_prepared_frames_device, so the same bits, but S chains
// side by side instead of one (an alignment is a dependent chain of iterations: one sequence cannot fill the chip; SURVEY §8e
// "replicas": masked_registration_ecc.py:105-191 is one such chain per camera).  d_*: HOST arrays of nseq device pointers;
// results: HOST [nseq][results_stride][4] doubles = (tx, ty, correlation coefficient, iterations) per image; good: HOST [nseq],
// images aligned before the first failure of that sequence (nframes[q]: all); warps: HOST [nseq][2] in/out (last good result).
// Returns 0, or -1 on an error of the call itself.
RIR_EXPORT int rir_ecc_align_multi_device(const float *const *d_ref_norm, const float *const *d_norm, const float *const *d_gx, const float *const *d_gy, int w,
										  int h, int nseq, const int *nframes, float *warps, int max_iterations, double eps, double *results, int results_stride,
										  int *good, void *stream)
{
	return rir_ecc_align_multi_overlapped_device(d_ref_norm, d_norm, d_gx, d_gy, w, h, nseq, nframes, warps, max_iterations, eps, results, results_stride, good,
												 nullptr, 0, stream);
}

// The same, and UNDER the alignments the pre-processing of what comes next (`next`: nnext jobs, each the arguments of one
// rir_ecc_prepare_frames_device call - normally the next chunk of every sequence): the alignment launch needs the whole chip to
// START (every workgroup resident: other kernels beside it then can keep the last ones from fitting), but once it
// reports that it is resident it leaves a fifth of every CU's places and most of the memory system unused - so the library waits
// for that report (a word of host memory the kernel writes) and then runs the jobs on a stream of its own beside it.  The
// caller's stream is ordered behind them when the call returns.  Results and errors as rir_ecc_align_multi_device; the jobs'
// outputs must not be what this call's alignments read.
RIR_EXPORT int rir_ecc_align_multi_overlapped_device(const float *const *d_ref_norm, const float *const *d_norm, const float *const *d_gx,
													 const float *const *d_gy, int w, int h, int nseq, const int *nframes, float *warps, int max_iterations,
													 double eps, double *results, int results_stride, int *good, const rir_ecc_prepare_job *next, int nnext,
													 void *stream)
{
	if (!device_ready())
		return -1;
	if (nnext < 0 || (nnext > 0 && !next))
	{
		log_error("rir_ecc_align_multi_overlapped_device: invalid argument");
		return -1;
	}
	for (int j = 0; j < nnext; ++j)
		if (!prepare_args_ok(next[j].d_imgs, next[j].dtype, next[j].w, next[j].h, next[j].nframes, next[j].win_x, next[j].win_y, next[j].win_w, next[j].win_h,
							 next[j].d_norm, next[j].d_gx, next[j].d_gy))
		{
			log_error("rir_ecc_align_multi_overlapped_device: invalid pre-processing job");
			return -1;
		}
	bool bad = !d_ref_norm || !d_norm || !d_gx || !d_gy || !nframes || !warps || !results || !good || w < 2 || h < 2 || !ecc_size_ok(w, h) || nseq <= 0 || nseq > 4096 ||
			   max_iterations <= 0 || max_iterations > kEccMaxIterations || !(eps >= 0) || results_stride <= 0;
	size_t total_frames = 0;
	for (int q = 0; !bad && q < nseq; ++q)
	{
		bad = !d_ref_norm[q] || !d_norm[q] || !d_gx[q] || !d_gy[q] || nframes[q] < 0 || nframes[q] > kEccMaxSequence || nframes[q] > results_stride;
		total_frames += bad ? 0 : (size_t)nframes[q];
	}
	if (bad)
	{
		log_error("rir_ecc_align_multi_device: invalid argument");
		return -1;
	}
	hipStream_t st = (hipStream_t)stream;
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	ScratchOrder order(sc, st);
	if (!order.ok)
		return -1;
	const int V = ecc_rows(w, h);
	static const bool env_per_iteration = getenv("RIR_ECC_LAUNCH_PER_ITERATION") != nullptr;
	static const bool env_no_overlap = getenv("RIR_ECC_NO_OVERLAP") != nullptr; // (measurements: the jobs after the alignments, on the caller's stream)
	const int cap = env_per_iteration ? 0 : ecc_run_multi_capacity();
	// the jobs of `next`: once, on `on` (the side stream behind what the caller's stream held when the call came - its inputs are ready and
	// the previous readers of its outputs are through - or the caller's stream itself)
	bool next_done = nnext == 0, next_on_side = false, next_ordered = false;
	struct SideGuard
	{ // a return on an error path: the jobs already queued on the side stream write the caller's buffers - not behind the caller's back
		EccScratch &sc;
		const bool &on_side, &ordered;
		~SideGuard()
		{
			if (on_side && !ordered)
				(void)hipStreamSynchronize(sc.side);
		}
	} side_guard{sc, next_on_side, next_ordered};
	auto run_next = [&](hipStream_t on) {
		if (next_done)
			return true;
		next_done = true;
		for (int j = 0; j < nnext; ++j)
			if (prepare_frames_on(next[j].d_imgs, next[j].dtype, next[j].w, next[j].h, next[j].nframes, next[j].sigma, next[j].win_x, next[j].win_y,
								  next[j].win_w, next[j].win_h, next[j].d_norm, next[j].d_gx, next[j].d_gy, on) != 0)
				return false;
		return true;
	};
	const bool overlap = nnext > 0 && !env_no_overlap && cap >= 1;
	if (overlap)
	{
		if (!sc.multi_go && !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&sc.multi_go), 64, hipHostMallocCoherent | hipHostMallocMapped), "hipHostMalloc"))
			return -1;
		if (!sc.side && (!hip_ok(hipStreamCreateWithFlags(&sc.side, hipStreamNonBlocking), "hipStreamCreate") ||
						 !hip_ok(hipEventCreateWithFlags(&sc.side_in, hipEventDisableTiming), "hipEventCreate") ||
						 !hip_ok(hipEventCreateWithFlags(&sc.side_out, hipEventDisableTiming), "hipEventCreate")))
			return -1;
		if (!hip_ok(hipEventRecord(sc.side_in, st), "hipEventRecord") || !hip_ok(hipStreamWaitEvent(sc.side, sc.side_in, 0), "hipStreamWaitEvent"))
			return -1;
	}
	// beside a launch that has reported itself resident (or has ended without): the jobs on the side stream
	auto overlap_next = [&](unsigned int epoch) {
		if (!overlap || next_done)
			return true;
		const auto t0 = std::chrono::steady_clock::now();
		volatile unsigned int *go = sc.multi_go;
		while (*go != epoch)
		{
			if (hipStreamQuery(st) != hipErrorNotReady || std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
				break; // (the launch has ended - it was called off, or the chunk is tiny - or never said so: the jobs are queued all the same)
			__builtin_ia32_pause();
		}
		(void)hipGetLastError();
		next_on_side = true;
		return run_next(sc.side) && hip_ok(hipEventRecord(sc.side_out, sc.side), "hipEventRecord");
	};
	// (before every return from here on: the caller's stream behind the jobs)
	auto finish = [&](int rc) {
		if (rc == 0 && !next_done && !run_next(st))
			return -1;
		if (next_on_side && !next_ordered)
		{
			if (!hip_ok(hipStreamWaitEvent(st, sc.side_out, 0), "hipStreamWaitEvent"))
				return -1;
			next_ordered = true;
		}
		return rc;
	};
	if (cap < 1)
	{ // the device cannot hold a resident launch (or the runtime cannot tell): sequence after sequence on the single-sequence path
		for (int q = 0; q < nseq; ++q)
		{
			good[q] = nframes[q] == 0 ? 0
									  : align_frames_locked(sc, d_ref_norm[q], d_norm[q], d_gx[q], d_gy[q], w, h, nframes[q], warps + 2 * q, max_iterations, eps,
															results + (size_t)q * results_stride * 4, st);
			if (good[q] < 0)
				return -1;
		}
		return finish(0);
	}
	const size_t rows_b = (ecc_run_workspace_bytes(w, h) + 255) & ~(size_t)255;
	// the page-locked block the results come back to: the table, the per-image results, the launch's decision word
	const size_t back_bytes = ((size_t)nseq * sizeof(EccSeq) + std::max<size_t>(total_frames, 1) * sizeof(EccFrameResult) + 15) & ~(size_t)15;
	if (!sc.multi_rows.reserve((size_t)nseq * rows_b) || !sc.multi_results.reserve(std::max<size_t>(total_frames, 1) * sizeof(EccFrameResult)) ||
		!sc.multi_table.reserve((size_t)nseq * sizeof(EccSeq)) || !sc.multi_stage.reserve((size_t)nseq * sizeof(EccSeq)) ||
		!sc.multi_back.reserve(back_bytes + 16))
		return -1;
	EccSeq *hs = sc.multi_stage.as<EccSeq>();
	size_t r0 = 0;
	for (int q = 0; q < nseq; ++q)
	{
		EccSeq e{};
		e.templ = d_ref_norm[q], e.image = d_norm[q], e.gx = d_gx[q], e.gy = d_gy[q];
		e.rows = reinterpret_cast<double *>(sc.multi_rows.as<char>() + (size_t)q * rows_b);
		e.results = sc.multi_results.as<EccFrameResult>() + r0;
		e.tx0 = warps[2 * q], e.ty0 = warps[2 * q + 1];
		e.nframes = nframes[q], e.frames_done = 0;
		hs[q] = e;
		r0 += (size_t)nframes[q];
	}
	if (!hip_ok(hipMemcpyAsync(sc.multi_table.ptr, hs, (size_t)nseq * sizeof(EccSeq), hipMemcpyHostToDevice, st), "H2D"))
		return -1;
	// sequences per launch and workgroups per sequence: as many slices as the device holds for the sequences of the launch
	// (RIR_ECC_MULTI_SLICES: a fixed number, for measurements).  Every launch first finds out whether it is fully resident
	// (resident_device.h); one that is not has written nothing and is repeated with half the slices - any number of slices gives the
	// same bits - and, from one slice per sequence, handed to the single-sequence path.
	static const int env_slices = getenv("RIR_ECC_MULTI_SLICES") ? atoi(getenv("RIR_ECC_MULTI_SLICES")) : 0;
	const bool debug_bail = test_hook("RIR_DEBUG_ECC_BAIL") != nullptr; // (tests: as if the first attempt of every launch had not become resident)
	const bool fresh_ctl = sc.multi_ctl.cap == 0;
	if (!sc.multi_ctl.reserve(256))
		return -1;
	if (fresh_ctl && (!hip_ok(hipMemsetAsync(sc.multi_ctl.ptr, 0, 256, st), "memset") || (sc.multi_arrivals = 0, false)))
		return -1;
	unsigned int *d_ctl = sc.multi_ctl.as<unsigned int>();
	std::vector<int> solo; // sequences that go through the single-sequence path in the end
	char *hb = sc.multi_back.as<char>();
	auto read_back = [&]() { // the table (images gone through) and the per-image results, queued behind what has been launched
		return hip_ok(hipMemcpyAsync(hb, sc.multi_table.ptr, (size_t)nseq * sizeof(EccSeq), hipMemcpyDeviceToHost, st), "D2H") &&
			   (!total_frames || hip_ok(hipMemcpyAsync(hb + (size_t)nseq * sizeof(EccSeq), sc.multi_results.ptr, total_frames * sizeof(EccFrameResult),
													   hipMemcpyDeviceToHost, st),
										"D2H"));
	};
	bool back_fresh = false;
	// (a launch holds sequences in pairs: two service workgroups and at least one compute workgroup per pair)
	const ResidentPlan plan = resident_plan(cap, 3, (nseq + 1) / 2);
	if (plan.units_per_launch < 1)
	{ // (a device that holds fewer than three such workgroups)
		for (int q = 0; q < nseq; ++q)
			solo.push_back(q);
	}
	for (int q0 = 0; plan.units_per_launch >= 1 && q0 < nseq; q0 += 2 * plan.units_per_launch)
	{
		const int nl = std::min(2 * plan.units_per_launch, nseq - q0), groups = (nl + 1) / 2;
		int nslices = std::max(1, std::min(V, (cap - nl) / groups));
		if (env_slices > 0)
			nslices = std::max(1, std::min(nslices, env_slices));
		for (int attempt = 0;; ++attempt)
		{
			// (a slice's time is that of its rows, one after the other: no more slices than give every slice the same largest number of rows)
			const int rows_per_slice = (V + nslices - 1) / nslices;
			nslices = (V + rows_per_slice - 1) / rows_per_slice;
			const unsigned int epoch = ++sc.epoch, total = (unsigned int)ecc_run_multi_grid(nl, nslices);
			if (debug_bail && attempt == 0)
			{ // the launch is called off by hand: the decision word says BAIL before anybody arrives
				const unsigned int w_ = ((epoch & 0x3fffffffu) << 2) | 2u;
				if (!hip_ok(hipMemcpyAsync(d_ctl + 1, &w_, 4, hipMemcpyHostToDevice, st), "H2D") || !hip_ok(hipStreamSynchronize(st), "sync"))
					return -1;
			}
			if (attempt > 0 && next_on_side)
				(void)hipStreamSynchronize(sc.side); // (a launch that was called off is repeated smaller: not beside this call's own jobs)
			if (overlap)
				*static_cast<volatile unsigned int *>(sc.multi_go) = 0u;
			if (!hip_ok(launch_ecc_run_multi(sc.multi_table.as<EccSeq>() + q0, nl, nslices, w, h, max_iterations, eps, epoch, d_ctl, sc.multi_arrivals,
											 overlap ? sc.multi_go : nullptr, st),
						"ecc run (multi)"))
				return -1;
			sc.multi_arrivals += total;
			// while the launch runs: the next chunk's pre-processing beside it.  (Before anything else is queued behind the launch - a
			// copy to host memory that is not page-locked holds the calling thread until the stream has got there.)
			if (!overlap_next(epoch))
				return -1;
			// (behind the last launch the results come back with the decision: one wait instead of two)
			volatile unsigned int &decision = *reinterpret_cast<volatile unsigned int *>(hb + back_bytes);
			decision = 0;
			const bool last = q0 + 2 * plan.units_per_launch >= nseq;
			if (!hip_ok(hipMemcpyAsync(hb + back_bytes, d_ctl + 1, 4, hipMemcpyDeviceToHost, st), "D2H") || (last && !read_back()) ||
				!hip_ok(wait_stream(st), "sync"))
				return -1;
			if (decision == (((epoch & 0x3fffffffu) << 2) | 1u))
			{ // resident: the chunk is aligned
				back_fresh = last;
				break;
			}
			if (nslices == 1)
			{ // not even one compute workgroup per pair fits beside what else is running: sequence by sequence
				for (int q = q0; q < q0 + nl; ++q)
					solo.push_back(q);
				break;
			}
			nslices = std::max(1, nslices / 2);
		}
	}
	if (!back_fresh && (!read_back() || !hip_ok(wait_stream(st), "sync")))
		return -1;
	static const bool diag = getenv("RIR_ECC_DIAG") != nullptr; // (-DRIR_ECC_DIAG builds: where an iteration's time goes, sequence 0)
	if (diag)
	{
		unsigned long long dg[16];
		if (hipMemcpy(dg, sc.multi_rows.as<char>() + (size_t)V * 256 + 64, sizeof(dg), hipMemcpyDeviceToHost) == hipSuccess && dg[2] && dg[6] && dg[10])
			std::fprintf(stderr, "ecc multi (us), sequence 0's service workgroup, per iteration: waiting for + adding the rows %.2f  solve + publish %.2f | group 0, per turn: slice 0: rows %.2f (pixel loop of the first row %.2f)  waiting for the decision %.2f | last slice: rows %.2f (%.2f)  waiting %.2f  (%llu iterations so far, %d sequences)\n",
						 dg[0] * 0.01 / dg[2], dg[1] * 0.01 / dg[2], dg[4] * 0.01 / dg[6], dg[7] * 0.01 / dg[6], dg[5] * 0.01 / dg[6], dg[8] * 0.01 / dg[10], dg[11] * 0.01 / dg[10], dg[9] * 0.01 / dg[10], dg[2], nseq);
	}
	const EccSeq *back = reinterpret_cast<const EccSeq *>(hb);
	const EccFrameResult *r = reinterpret_cast<const EccFrameResult *>(hb + (size_t)nseq * sizeof(EccSeq));
	for (int q : solo)
	{
		good[q] = nframes[q] == 0 ? 0
								  : align_frames_locked(sc, d_ref_norm[q], d_norm[q], d_gx[q], d_gy[q], w, h, nframes[q], warps + 2 * q, max_iterations, eps,
														results + (size_t)q * results_stride * 4, st);
		if (good[q] < 0)
			return -1;
	}
	for (int q = 0; q < nseq; ++q)
	{
		double *res = results + (size_t)q * results_stride * 4;
		int g = 0;
		if (std::find(solo.begin(), solo.end(), q) != solo.end())
		{
			r += nframes[q];
			continue;
		}
		for (; g < back[q].frames_done && g < nframes[q]; ++g)
		{
			if (r[g].done == 2 || std::isnan(r[g].rho))
				break;
			res[4 * g] = r[g].tx, res[4 * g + 1] = r[g].ty, res[4 * g + 2] = r[g].rho, res[4 * g + 3] = r[g].iter;
			warps[2 * q] = r[g].tx, warps[2 * q + 1] = r[g].ty;
		}
		good[q] = g;
		if (g < nframes[q])
		{ // (what stopped the sequence: its own failure - done 2 -, or images it never reached)
			char msg[200];
			std::snprintf(msg, sizeof(msg), "ECC: sequence %d of %d stopped at image %d of %d (images gone through %d, done %d, iterations %d)", q, nseq, g,
						  nframes[q], back[q].frames_done, g < back[q].frames_done ? r[g].done : -1, g < back[q].frames_done ? r[g].iter : -1);
			log_error(msg);
		}
		r += nframes[q];
	}
	return finish(0);
}

// Host-pointer form, the drop-in for cv2.findTransformECC(templ, image, warp, MOTION_TRANSLATION, criteria, mask, 1):
// templ/image float32 [h][w], mask uint8 or NULL, warp = float[2] (tx, ty) in/out.  Returns 0, or -1 (OpenCV raises).
RIR_EXPORT int find_transform_ecc_translation(const float *templ, const float *image, const unsigned char *mask, int w, int h, float *warp,
											  int max_iterations, double eps, double *cc)
{
	if (!device_ready())
		return -1;
	if (!templ || !image || !warp || w < 2 || h < 2 || !ecc_size_ok(w, h) || max_iterations <= 0 || !(eps >= 0))
	{
		log_error("find_transform_ecc_translation: invalid argument");
		return -1;
	}
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	hipStream_t st = default_stream();
	ScratchOrder order(sc, st);
	if (!order.ok)
		return -1;
	const size_t npx = (size_t)w * h;
	if (!sc.templ.reserve(npx * 4) || !sc.image.reserve(npx * 4) || (mask && !sc.mask.reserve(npx)))
		return -1;
	if (!hip_ok(hipMemcpyAsync(sc.templ.ptr, templ, npx * 4, hipMemcpyHostToDevice, st), "H2D") ||
		!hip_ok(hipMemcpyAsync(sc.image.ptr, image, npx * 4, hipMemcpyHostToDevice, st), "H2D") ||
		(mask && !hip_ok(hipMemcpyAsync(sc.mask.ptr, mask, npx, hipMemcpyHostToDevice, st), "H2D")))
		return -1;
	return run_ecc(sc, sc.templ.as<float>(), sc.image.as<float>(), mask ? sc.mask.as<uint8_t>() : nullptr, w, h, warp, max_iterations, eps, cc,
				   nullptr, st);
}

// (im - min(im)) / (max(im) - min(im)) in float32 on a window of a device image: the normalisation MaskedRegistratorECC
// applies to both images before the alignment (masked_registration_ecc.py:162-166).  d_src: float rows of `src_stride`
// elements, the window starts at d_src; d_dst: dense [h][w].
RIR_EXPORT int rir_minmax_normalize_device(const float *d_src, int w, int h, int src_stride, float *d_dst, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_src || !d_dst || w <= 0 || h <= 0 || src_stride < w)
	{
		log_error("rir_minmax_normalize_device: invalid argument");
		return -1;
	}
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	ScratchOrder order(sc, (hipStream_t)stream);
	if (!order.ok || !sc.mm.reserve(2 * kMinMaxParts * sizeof(float)))
		return -1;
	return hip_ok(launch_minmax_normalize(d_src, w, h, src_stride, d_dst, sc.mm.as<float>(), (hipStream_t)stream), "minmax_normalize") ? 0 : -1;
}
