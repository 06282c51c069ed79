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

// ---- reference FileAttributes (metadata trailer), reached through its C++ class ---------------------
// reference src/cpp/tools/FileAttributes.cpp; the handle-based C entry points live in tools.cpp,
// which needs minizip's unzip.h (absent), so the class is driven directly.
#ifdef RIR_REF_WITH_ATTRS
#include "FileAttributes.h"
extern "C"
{
	// write a trailer with the given global attributes / timestamps / one attribute per frame
	__attribute__((visibility("default"))) int ref_attrs_write(const char *filename, int nglobal, const char **gkeys, const char **gvals,
																const int *gval_lens, int nframes, const long long *times, const char *frame_key,
																const char **frame_vals, const int *frame_val_lens)
	{
		rir::FileAttributes fa;
		if (!fa.open(filename))
			return -1;
		std::map<std::string, std::string> g;
		for (int i = 0; i < nglobal; ++i)
			g[gkeys[i]] = std::string(gvals[i], gvals[i] + gval_lens[i]);
		fa.setGlobalAttributes(g);
		fa.resize(nframes);
		for (int i = 0; i < nframes; ++i)
		{
			fa.setTimestamp(i, times[i]);
			if (frame_key)
				fa.addAttribute(i, frame_key, std::string(frame_vals[i], frame_vals[i] + frame_val_lens[i]));
		}
		fa.close();
		return 0;
	}
	// read back: returns the frame count, fills times (cap entries) and the value of one global key
	__attribute__((visibility("default"))) int ref_attrs_read(const char *filename, long long *times, int cap, const char *gkey, char *gval, int *gval_len,
															   int *nglobal)
	{
		rir::FileAttributes fa;
		if (!fa.open(filename))
			return -1;
		int n = (int)fa.size();
		for (int i = 0; i < n && i < cap; ++i)
			times[i] = fa.timestamp(i);
		*nglobal = (int)fa.globalAttributes().size();
		auto it = fa.globalAttributes().find(gkey);
		if (it != fa.globalAttributes().end() && (int)it->second.size() <= *gval_len)
		{
			memcpy(gval, it->second.data(), it->second.size());
			*gval_len = (int)it->second.size();
		}
		else
			*gval_len = -1;
		fa.discard();
		return n;
	}
}
#endif
