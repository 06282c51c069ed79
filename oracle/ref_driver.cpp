// TEST INFRASTRUCTURE — not part of the product.
//
// Thin extern "C" driver over the reference's own C++ signal_processing sources, compiled where
// they lie under /root/reference by oracle/build_ref.sh into oracle/_ref/librir_ref.so.
// Nothing from the reference is copied here: this file only #includes its headers at build time and
// forwards to its public templates/classes, so that the handle-registry entry points of the
// reference C ABI (bad_pixels_create/... which need tools.cpp -> zstd.h/unzip.h, absent in this
// image) are not needed to reach the arithmetic.
//
//   reference entry points reached through this driver
//     rir::badPixels<T>            src/cpp/signal_processing/Filters.h:135-193
//     rir::BadPixels::init/correct src/cpp/signal_processing/BadPixels.cpp:13-66
//     rir::translate<T,U>          src/cpp/signal_processing/Filters.h:249-326   (<u16,float> for F6)
//     rir::medianFilter<T,U>       src/cpp/signal_processing/Filters.h:71-129
//     rir::clampMin                src/cpp/signal_processing/Filters.cpp:7-50
//   (translate / gaussian_filter / find_median_pixel[_mask] are the reference's own extern "C"
//    symbols from signal_processing.cpp and are called directly, without this driver.)
#include "Filters.h"
#include "BadPixels.h"

#include <cstring>
#include <memory>
#include <vector>

extern "C"
{

	// Detector: writes up to cap (x,y) pairs, returns the number of flagged pixels.
	__attribute__((visibility("default"))) int ref_bad_pixels_detect(const unsigned short *img, int w, int h, int *xy, int cap)
	{
		rir::Polygon p = rir::badPixels(img, (size_t)w, (size_t)h, 5);
		int n = (int)p.size();
		for (int i = 0; i < n && i < cap; ++i)
		{
			xy[2 * i] = (int)p[i].x();
			xy[2 * i + 1] = (int)p[i].y();
		}
		return n;
	}

	// BadPixels object life cycle without the handle registry.
	__attribute__((visibility("default"))) void *ref_bad_pixels_new(const unsigned short *first, int w, int h)
	{
		rir::BadPixels *bp = new rir::BadPixels();
		bp->init(first, w, h);
		return bp;
	}
	__attribute__((visibility("default"))) void ref_bad_pixels_correct(void *bp, const unsigned short *in, unsigned short *out)
	{
		static_cast<rir::BadPixels *>(bp)->correct(in, out);
	}
	__attribute__((visibility("default"))) void ref_bad_pixels_delete(void *bp)
	{
		delete static_cast<rir::BadPixels *>(bp);
	}

	// The <u16 -> float> instantiation used by IRFileLoader::removeMotionGeneric
	// (src/cpp/video_io/IRFileLoader.cpp:617-627), not reachable from the C entry `translate`.
	__attribute__((visibility("default"))) void ref_translate_u16_f32_nearest(const unsigned short *src, float *dst, int w, int h, float dx, float dy)
	{
		rir::translate(src, dst, 0.f, (size_t)w, (size_t)h, dx, dy, rir::TranslateNearest);
	}

	__attribute__((visibility("default"))) void ref_median_filter_u16(const unsigned short *src, unsigned short *dst, int w, int h)
	{
		rir::medianFilter(src, dst, (size_t)w, (size_t)h);
	}

	__attribute__((visibility("default"))) void ref_clamp_min(unsigned short *img, int size, unsigned short v)
	{
		rir::clampMin(img, (size_t)size, v);
	}
}
