// C ABI of the registration step: translation-only ECC alignment on the device.
//
// The reference computes sub-pixel translations in Python by calling OpenCV
// (src/python/librir/registration/masked_registration_ecc.py:166-168, cv2.findTransformECC with
// MOTION_TRANSLATION); these two entry points are what that call binds to here.
#include <cmath>
#include <cstring>

#include "ecc_kernels.h"
#include "rir_amd_device.h"
#include "runtime.h"

using namespace rir;

namespace
{
	struct EccScratch
	{
		std::mutex mu;
		DeviceBuffer gx, gy, partials, state, templ, image, mask, mm;
	};
	EccScratch &scratch()
	{
		static EccScratch s;
		return s;
	}

#ifndef RIR_ECC_FIRST_BATCH
#define RIR_ECC_FIRST_BATCH 6 /* alignments of a tracked sequence settle within 4-6 iterations: one read-back of the state instead of two */
#endif
	// runs the iterations; the caller holds the scratch mutex
	int run_ecc(EccScratch &sc, const float *d_templ, const float *d_image, const uint8_t *d_mask, int w, int h, float *warp, int max_iter,
				double eps, double *cc, int *iterations, hipStream_t st)
	{
		const size_t npx = (size_t)w * h;
		if (!sc.gx.reserve(npx * 4) || !sc.gy.reserve(npx * 4) || !sc.partials.reserve(ecc_workspace_bytes(w, h)) ||
			!sc.state.reserve(sizeof(EccState)))
			return -1;
		EccState *d_state = sc.state.as<EccState>();
		if (!hip_ok(launch_ecc_prepare(d_image, w, h, sc.gx.as<float>(), sc.gy.as<float>(), d_state, warp[0], warp[1], max_iter, eps, st),
					"ecc prepare"))
			return -1;
		EccState hs;
		std::memset(&hs, 0, sizeof(hs));
		// iterations are queued in batches (a finished alignment turns the remaining launches into no-ops);
		// the state comes back once per batch
		int launched = 0;
		while (true)
		{
			const int batch = std::min(launched == 0 ? RIR_ECC_FIRST_BATCH : 8, max_iter - launched);
			for (int i = 0; i < batch; ++i)
				if (!hip_ok(launch_ecc_iterate(d_templ, d_image, sc.gx.as<float>(), sc.gy.as<float>(), d_mask, w, h, sc.partials.as<double>(),
											   d_state, st),
							"ecc iterate"))
					return -1;
			launched += batch;
			if (!hip_ok(hipMemcpyAsync(&hs, d_state, sizeof(hs), hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(hipStreamSynchronize(st), "sync"))
				return -1;
			if (hs.done || launched >= max_iter)
				break;
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
} // namespace

// d_templ, d_image: float [h][w] in device memory; d_mask: uint8 [h][w] or NULL; warp: HOST float[2] = (tx, ty), in/out
// (start value -> result; the aligned image is image(x + tx, y + ty) ~ templ(x, y)); *cc = correlation coefficient.
RIR_EXPORT int rir_ecc_translation_device(const float *d_templ, const float *d_image, const unsigned char *d_mask, int w, int h, float *warp,
										  int max_iterations, double eps, double *cc, int *iterations, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_templ || !d_image || !warp || w < 2 || h < 2 || max_iterations <= 0 || !(eps >= 0))
	{
		log_error("rir_ecc_translation_device: invalid argument");
		return -1;
	}
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	return run_ecc(sc, d_templ, d_image, d_mask, w, h, warp, max_iterations, eps, cc, iterations, (hipStream_t)stream);
}

// Host-pointer form, the drop-in for cv2.findTransformECC(templ, image, warp, MOTION_TRANSLATION, criteria, mask, 1):
// templ/image float32 [h][w], mask uint8 or NULL, warp = float[2] (tx, ty) in/out.  Returns 0, or -1 (OpenCV raises).
RIR_EXPORT int find_transform_ecc_translation(const float *templ, const float *image, const unsigned char *mask, int w, int h, float *warp,
											  int max_iterations, double eps, double *cc)
{
	if (!device_ready())
		return -1;
	if (!templ || !image || !warp || w < 2 || h < 2 || max_iterations <= 0 || !(eps >= 0))
	{
		log_error("find_transform_ecc_translation: invalid argument");
		return -1;
	}
	EccScratch &sc = scratch();
	std::lock_guard<std::mutex> lock(sc.mu);
	hipStream_t st = default_stream();
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
	if (!sc.mm.reserve(2 * 64 * sizeof(float)))
		return -1;
	return hip_ok(launch_minmax_normalize(d_src, w, h, src_stride, d_dst, sc.mm.as<float>(), (hipStream_t)stream), "minmax_normalize") ? 0 : -1;
}
