// C ABI of the signal_processing half of the hot path.
//
// Two layers, both extern "C":
//   * the reference's own entry points, same names / arguments / return codes
//     (reference src/cpp/signal_processing/signal_processing.h:29-94): host pointers in, host
//     pointers out, synchronous.  They stage through device buffers and call the layer below.
//   * rir_*_device: the same operations on device-resident batches [n][h][w] (no PCIe in the
//     path, asynchronous on the caller's HIP stream) - what pipelines and bench.py use.
// There is no CPU fallback: without a HIP device every entry point logs and returns its error code.
#include <cmath>
#include <cstring>
#include <map>

#include "filter_kernels.h"
#include "label_kernels.h"
#include "runtime.h"

using namespace rir;

namespace
{
	// the caller's stream, taken literally: NULL is HIP's null (legacy default) stream
	hipStream_t as_stream(void *s) { return (hipStream_t)s; }

	int strategy_from_string(const char *s)
	{ // signal_processing.cpp:20-41: NULL, "" and "noborder" leave border pixels untouched
		if (!s || std::strlen(s) == 0 || std::strcmp(s, "noborder") == 0)
			return TRANSLATE_UNCHANGED;
		if (std::strcmp(s, "background") == 0)
			return TRANSLATE_CONSTANT;
		if (std::strcmp(s, "wrap") == 0)
			return TRANSLATE_WRAP;
		if (std::strcmp(s, "nearest") == 0)
			return TRANSLATE_NEAREST;
		if (std::strcmp(s, "noborder_source") == 0) // device layer: "noborder" over an implicit copy of the input
			return TRANSLATE_SOURCE;
		return -1;
	}

	int dtype_size(int type)
	{
		switch (type)
		{
		case '?':
		case 'b':
		case 'B':
			return 1;
		case 'h':
		case 'H':
			return 2;
		case 'i':
		case 'I':
		case 'f':
			return 4;
		case 'l':
		case 'L':
		case 'd':
			return 8;
		default:
			return 0;
		}
	}

	// Gaussian table, built exactly as the reference does (signal_processing.cpp:79-99): float exp
	// of a float argument, division by (pi*s) in double, float running sum in x-outer/y-inner
	// order, float normalisation.  Host libm, like the reference.
	int gaussian_radius(float sigma)
	{
		int radius = (int)(sigma * 2);
		return radius < 1 ? 1 : radius;
	}
	std::vector<float> gaussian_table(float sigma, int radius)
	{
		const int kw = 2 * radius + 1;
		std::vector<float> k((size_t)kw * kw);
		const float s = 2.0f * sigma * sigma;
		float sum = 0.0f;
		for (int x = -radius; x <= radius; x++)
			for (int y = -radius; y <= radius; y++)
			{
				const float r = (float)std::sqrt((double)(x * x + y * y));
				const float e = std::exp(-(r * r) / s); // float overload
				const float v = (float)((double)e / (3.14159265358979323846 * (double)s));
				k[x + radius + (y + radius) * kw] = v;
				sum += v;
			}
		for (auto &v : k)
			v /= sum;
		return k;
	}

	// Tables are cached per (device, sigma).  A caller keeps the shared_ptr until its kernel has been QUEUED; freeing a
	// DeviceBuffer is a hipFree, which waits for the work already queued on the device - so an evicted table is never
	// released under a kernel that reads it, whichever thread and stream that kernel runs on.
	typedef std::shared_ptr<DeviceBuffer> GaussTable;
	struct GaussCache
	{
		std::mutex mu;
		std::map<uint64_t, std::pair<GaussTable, uint64_t>> tables; // key = device id << 32 | float bits of sigma -> (table, last use)
		uint64_t tick = 0;
	};
	GaussCache &gauss_cache()
	{
		static GaussCache c;
		return c;
	}
	GaussTable gaussian_table_device(float sigma, int radius)
	{
		uint32_t bits;
		std::memcpy(&bits, &sigma, 4);
		int dev = 0;
		if (!hip_ok(hipGetDevice(&dev), "hipGetDevice"))
			return nullptr;
		const uint64_t key = ((uint64_t)(uint32_t)dev << 32) | bits;
		GaussCache &c = gauss_cache();
		std::lock_guard<std::mutex> g(c.mu);
		auto it = c.tables.find(key);
		if (it != c.tables.end())
		{
			it->second.second = ++c.tick;
			return it->second.first;
		}
		std::vector<float> k = gaussian_table(sigma, radius);
		{ // 1-D factors a[-r..r] of the (separable) table, appended after it: k[dx][dy] ~ a[dx] * a[dy]
			const int kw = 2 * radius + 1;
			const double a0 = std::sqrt((double)k[(size_t)radius * kw + radius]);
			for (int d = 0; d < kw; ++d)
				k.push_back((float)((double)k[(size_t)radius * kw + d] / a0));
		}
		auto buf = std::make_shared<DeviceBuffer>();
		if (!buf->reserve(k.size() * sizeof(float)))
			return nullptr;
		if (!hip_ok(hipMemcpy(buf->ptr, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice), "gaussian table upload"))
			return nullptr;
		if (c.tables.size() >= 64)
		{ // least recently used entry out (its memory goes when the last holder drops it)
			auto lru = c.tables.begin();
			for (auto i = c.tables.begin(); i != c.tables.end(); ++i)
				if (i->second.second < lru->second.second)
					lru = i;
			c.tables.erase(lru);
		}
		c.tables[key] = std::make_pair(buf, ++c.tick);
		return buf;
	}

	// Scratch for the host-pointer entry points (one call at a time per process; the reference's
	// objects are not re-entrant either).
	struct HostScratch
	{
		std::mutex mu;
		DeviceBuffer a, b, c, d;
		PinnedBuffer h_in, h_out; // page-locked staging of the caller's images
		PinnedBuffer h_tab;		  // label_image: the per-component tables, written by the kernel
	};
	HostScratch &scratch()
	{
		static HostScratch s;
		return s;
	}
	// The caller's images are pageable memory.  Handed to hipMemcpyAsync as they are, the runtime stages them itself, a chunk at a time on
	// the calling thread (measured, one 640x512 image: 70-130 us each way).  Here an image is staged in page-locked memory - the copy cut
	// over the helper threads (host_copy.cpp) - and the KERNEL works on the staging buffers themselves, reading its input and writing its
	// result over the link (profiles/r05_zero_copy_probe.txt: translate of a float image 54 us that way, 89 us with a copy in and a copy out,
	// before the copy calls' own cost): no copy call, one launch, one wait.  RIR_ABI_ZERO_COPY=0: a transfer into / out of device buffers
	// around the kernel, as before.  `slot`: byte offset in the input staging buffer (a call may stage two images).
	// (Round 6, tried and dropped: the image in 2-6 strips of rows - strip j's kernel queued once its rows are staged, the next strip's copy
	// and the previous strip's copy-out under it.  gaussian_filter of a float image: 94 us in one piece, 91 / 89 / 112 / 125 us in 2 / 3 / 4 / 6
	// strips: a launch and an event per strip cost what the overlap of two 14 us copies gains.)
	// -> the address the kernel reads (nullptr on failure)
	// A caller's buffer that lies in page-locked memory of this library (rir_host_alloc: the arrays the Python mirror returns, which are the
	// next call's input) is worked on where it is: no staging copy in, no copy back (round 6: the reference's three-call configs[2] path).
	const void *stage_in(HostScratch &s, DeviceBuffer &d, const void *h_src, size_t bytes, size_t slot, hipStream_t st)
	{
		if (abi_zero_copy() && host_block_contains(h_src, bytes))
			return h_src;
		if (!s.h_in.ptr || s.h_in.cap < slot + bytes)
			return nullptr;
		char *stage = s.h_in.as<char>() + slot;
		host_copy(stage, h_src, bytes);
		if (abi_zero_copy())
			return stage;
		if (!d.reserve(bytes) || !hip_ok(hipMemcpyAsync(d.ptr, stage, bytes, hipMemcpyHostToDevice, st), "H2D"))
			return nullptr;
		return d.ptr;
	}
	// -> the address the kernel writes its result to (nullptr on failure); keep: the caller's buffer holds values the kernel keeps
	// (translate "noborder"), it goes in first
	// in / in_bytes: what the kernel reads (stage_in's answer): a destination that overlaps it is staged, the kernels do not work in place
	void *stage_out(HostScratch &s, DeviceBuffer &d, void *h_dst, size_t bytes, bool keep, hipStream_t st, const void *in = nullptr, size_t in_bytes = 0)
	{
		if (abi_zero_copy() && host_block_contains(h_dst, bytes))
		{
			const char *a = static_cast<const char *>(in), *b = static_cast<const char *>(h_dst);
			if (!in || a + in_bytes <= b || b + bytes <= a)
				return h_dst; // (keep: the caller's values are where the kernel finds them)
		}
		if (!s.h_out.reserve(bytes))
			return nullptr;
		if (keep)
			host_copy(s.h_out.ptr, h_dst, bytes);
		if (abi_zero_copy())
			return s.h_out.ptr;
		if (!d.reserve(bytes) || (keep && !hip_ok(hipMemcpyAsync(d.ptr, s.h_out.ptr, bytes, hipMemcpyHostToDevice, st), "H2D")))
			return nullptr;
		return d.ptr;
	}
	// the call's result (at `from`, what stage_out returned) -> the caller: waits for the stream (the call is synchronous)
	bool hand_out(HostScratch &s, void *h_dst, const void *from, size_t bytes, hipStream_t st)
	{
		if (from == h_dst) // the kernel wrote into the caller's page-locked buffer
			return hip_ok(wait_stream(st), "sync");
		if (from != s.h_out.ptr && !hip_ok(hipMemcpyAsync(s.h_out.ptr, from, bytes, hipMemcpyDeviceToHost, st), "D2H"))
			return false;
		if (!hip_ok(wait_stream(st), "sync"))
			return false;
		host_copy(h_dst, s.h_out.ptr, bytes);
		return true;
	}

	// Bad-pixel object: reference rir::BadPixels (BadPixels.h:14-35) + the loader-side bitmap
	// (IRFileLoader.cpp:704-710).
	struct BadPixelsObject : public Object
	{
		const char *type_name() const override { return "BadPixels"; }
		int width = 0, height = 0;
		int floor_detect = 0;  // Filters.h:157-160
		int floor_correct = 0; // m_median_value, BadPixels.cpp:22-31
		std::vector<int> xy;   // raster order (x,y) pairs
		DeviceBuffer d_xy, d_bitmap;
		DeviceBuffer d_row_start; // [height + 1]: first flagged index of row >= y (the list is in raster order)
		DeviceBuffer d_fix;		  // scratch of the fused filter chain: repaired values, [nframes][count]
		int count() const { return (int)(xy.size() / 2); }
	};

	// Detector on a device-resident frame.  rows = number of rows taken into account.
	bool detect_bad_pixels(const uint16_t *d_img, int w, int rows, double std_factor, BadPixelsObject &bp, hipStream_t st)
	{
		DeviceBuffer hist, stats, flags;
		const int64_t npx = (int64_t)w * rows;
		if (!hist.reserve(65536 * sizeof(uint32_t)) || !stats.reserve(2 * sizeof(int64_t)) || !flags.reserve((size_t)npx))
			return false;
		if (!hip_ok(launch_histogram(d_img, nullptr, npx, 1, hist.as<uint32_t>(), st), "histogram"))
			return false;
		if (!hip_ok(launch_bad_pixels_stats(hist.as<uint32_t>(), (uint64_t)npx, stats.as<int64_t>(), st), "bad_pixels_stats"))
			return false;
		int64_t h_stats[2];
		if (!hip_ok(hipMemcpyAsync(h_stats, stats.ptr, sizeof(h_stats), hipMemcpyDeviceToHost, st), "stats D2H") ||
			!hip_ok(hipStreamSynchronize(st), "sync"))
			return false;
		// Filters.h:151-160 and BadPixels.cpp:25-31: the sums are exact integers, the rest is the
		// same double sequence as the host code.
		const int median = (int)h_stats[0];
		double sum = (double)h_stats[1];
		sum /= (double)(int)npx;
		sum = std::sqrt(sum);
		const uint16_t thr = (uint16_t)(int32_t)(sum * std_factor);
		bp.floor_detect = ((uint16_t)median > thr) ? (int)(uint16_t)((uint16_t)median - thr) : 0;
		bp.floor_correct = median - (int)(sum * 2);

		if (!hip_ok(launch_bad_pixels_detect(d_img, w, rows, std_factor, bp.floor_detect, flags.as<uint8_t>(), st), "bad_pixels_detect"))
			return false;
		std::vector<uint8_t> h_flags((size_t)npx);
		if (!hip_ok(hipMemcpyAsync(h_flags.data(), flags.ptr, (size_t)npx, hipMemcpyDeviceToHost, st), "flags D2H") ||
			!hip_ok(hipStreamSynchronize(st), "sync"))
			return false;
		bp.xy.clear();
		for (int y = 0; y < rows; ++y)
			for (int x = 0; x < w; ++x)
				if (h_flags[(size_t)y * w + x])
				{
					bp.xy.push_back(x);
					bp.xy.push_back(y);
				}
		return true;
	}

	bool upload_bad_pixels(BadPixelsObject &bp, hipStream_t st)
	{
		const size_t npx = (size_t)bp.width * bp.height;
		if (!bp.d_xy.reserve(bp.xy.size() * sizeof(int) + 8) || !bp.d_bitmap.reserve(npx))
			return false;
		std::vector<uint8_t> bitmap(npx, 0);
		for (int i = 0; i < bp.count(); ++i)
			bitmap[(size_t)bp.xy[2 * i + 1] * bp.width + bp.xy[2 * i]] = 1;
		if (bp.count() && !hip_ok(hipMemcpyAsync(bp.d_xy.ptr, bp.xy.data(), bp.xy.size() * sizeof(int), hipMemcpyHostToDevice, st), "xy H2D"))
			return false;
		if (!hip_ok(hipMemcpyAsync(bp.d_bitmap.ptr, bitmap.data(), npx, hipMemcpyHostToDevice, st), "bitmap H2D"))
			return false;
		std::vector<int> row_start((size_t)bp.height + 1, bp.count());
		for (int i = bp.count() - 1; i >= 0; --i)
			row_start[bp.xy[2 * i + 1]] = i;
		for (int y = bp.height - 1; y >= 0; --y) // rows without flagged pixels point at the next row's run
			row_start[y] = std::min(row_start[y], row_start[y + 1]);
		if (!bp.d_row_start.reserve(row_start.size() * sizeof(int)) ||
			!hip_ok(hipMemcpyAsync(bp.d_row_start.ptr, row_start.data(), row_start.size() * sizeof(int), hipMemcpyHostToDevice, st), "row_start H2D"))
			return false;
		return hip_ok(hipStreamSynchronize(st), "sync");
	}
} // namespace

// =====================================================================================================
// Device-resident batch layer
// =====================================================================================================

RIR_EXPORT int rir_device_available(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
	{
		(void)hipGetLastError();
		return 0;
	}
	return n > 0 ? 1 : 0;
}

RIR_EXPORT int rir_stream_synchronize(void *stream)
{
	if (!device_ready())
		return -1;
	return hip_ok(hipStreamSynchronize(as_stream(stream)), "hipStreamSynchronize") ? 0 : -1;
}

RIR_EXPORT int rir_translate_device(int type, const void *d_src, void *d_dst, int w, int h, int nframes, const float *d_offsets,
									int per_frame_offsets, const void *background, const char *strategy, void *stream)
{
	if (!device_ready())
		return -1;
	const int s = strategy_from_string(strategy);
	if (s < 0 || dtype_size(type) == 0 || w <= 0 || h <= 0 || nframes <= 0 || !d_src || !d_dst || !d_offsets || !background)
	{
		log_error("rir_translate_device: invalid argument");
		return -1;
	}
	return hip_ok(launch_translate(type, d_src, d_dst, background, w, h, nframes, d_offsets, per_frame_offsets, s, as_stream(stream)), "translate") ? 0 : -1;
}

RIR_EXPORT int rir_gaussian_filter_device(const float *d_src, float *d_dst, int w, int h, int nframes, float sigma, void *stream)
{
	if (!device_ready())
		return -1;
	if (w <= 0 || h <= 0 || nframes <= 0 || !d_src || !d_dst || !(sigma > 0))
	{
		log_error("rir_gaussian_filter_device: invalid argument");
		return -1;
	}
	const int radius = gaussian_radius(sigma);
	const GaussTable table = gaussian_table_device(sigma, radius); // held until the launch below has been queued
	if (!table)
		return -1;
	const float *d_k = table->as<float>();
	return hip_ok(launch_gaussian(d_src, d_dst, w, h, nframes, d_k, radius, as_stream(stream)), "gaussian_filter") ? 0 : -1;
}

// uint16 frames in, float32 out: gaussian_filter(frame.astype(float32)) without the converted copy in memory (sigma < 2.5).
RIR_EXPORT int rir_gaussian_filter_u16_device(const unsigned short *d_src, float *d_dst, int w, int h, int nframes, float sigma, void *stream)
{
	if (!device_ready())
		return -1;
	if (w <= 0 || h <= 0 || nframes <= 0 || !d_src || !d_dst || !(sigma > 0) || gaussian_radius(sigma) > 4)
	{
		log_error("rir_gaussian_filter_u16_device: invalid argument (sigma must be < 2.5 for the uint16 entry)");
		return -1;
	}
	const int radius = gaussian_radius(sigma);
	const GaussTable table = gaussian_table_device(sigma, radius); // held until the launch below has been queued
	if (!table)
		return -1;
	const float *d_k = table->as<float>();
	return hip_ok(launch_gaussian_u16(d_src, d_dst, w, h, nframes, d_k, radius, as_stream(stream)), "gaussian_filter") ? 0 : -1;
}

// float32 frames in, uint16 out: translate(...) followed by the truncating astype(uint16) of the wrapper, in one pass
// (each value is rounded to float first, exactly as the two-step form does).  background: HOST pointer to one uint16.
RIR_EXPORT int rir_translate_f32_u16_device(const float *d_src, unsigned short *d_dst, int w, int h, int nframes, const float *d_offsets,
											int per_frame_offsets, const void *background, const char *strategy, void *stream)
{
	if (!device_ready())
		return -1;
	const int s = strategy_from_string(strategy);
	if (s < 0 || w <= 0 || h <= 0 || nframes <= 0 || !d_src || !d_dst || !d_offsets || !background)
	{
		log_error("rir_translate_f32_u16_device: invalid argument");
		return -1;
	}
	return hip_ok(launch_translate('F', d_src, d_dst, background, w, h, nframes, d_offsets, per_frame_offsets, s, as_stream(stream)), "translate") ? 0 : -1;
}

// Byte planes of 16-bit frames as the reference hands them to its video codec (h264.cpp:1066-1082) and back (:3016-3051).
// Planes are [nframes][h][linesize] bytes, linesize >= w; d_it (8-bit integration-time image, [nframes][h][w]) may be NULL.
RIR_EXPORT int rir_split_planes_device(const unsigned short *d_img, const unsigned char *d_it, int w, int h, int nframes, int linesize,
									   unsigned char *d_Y, unsigned char *d_U, unsigned char *d_V, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_img || !d_Y || !d_U || !d_V || w <= 0 || h <= 0 || nframes <= 0 || linesize < w)
	{
		log_error("rir_split_planes_device: invalid argument");
		return -1;
	}
	return hip_ok(launch_split_planes(d_img, d_it, w, h, nframes, linesize, d_Y, d_U, d_V, as_stream(stream)), "split_planes") ? 0 : -1;
}
RIR_EXPORT int rir_merge_planes_device(const unsigned char *d_Y, const unsigned char *d_U, const unsigned char *d_V, int linesize, int w, int h,
									   int nframes, unsigned short *d_img, unsigned char *d_it, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_img || !d_U || !d_V || (d_it && !d_Y) || w <= 0 || h <= 0 || nframes <= 0 || linesize < w)
	{
		log_error("rir_merge_planes_device: invalid argument");
		return -1;
	}
	return hip_ok(launch_merge_planes(d_Y, d_U, d_V, linesize, w, h, nframes, d_img, d_it, as_stream(stream)), "merge_planes") ? 0 : -1;
}

// result: int32[nframes] on the device.  d_hist: unused (kept in the signature; may be NULL).
RIR_EXPORT int rir_find_median_pixel_device(const unsigned short *d_img, const unsigned char *d_mask, int size, int nframes, float percent,
											int *d_result, unsigned int *d_hist, void *stream)
{
	if (!device_ready())
		return -1;
	if (size <= 0 || nframes <= 0 || !d_img || !d_result)
	{
		log_error("rir_find_median_pixel_device: invalid argument");
		return -1;
	}
	(void)d_hist; // the counting happens in LDS, one quarter of the value range at a time: no histogram in memory
	// 65 535 bins, as the reference (Filters.cpp:59)
	return hip_ok(launch_quantile_select(d_img, d_mask, size, nframes, percent, 65535, d_result, as_stream(stream)), "find_median_pixel") ? 0 : -1;
}

RIR_EXPORT int rir_bad_pixels_create_device(const unsigned short *d_first_image, int width, int height, void *stream)
{
	if (!device_ready())
		return 0;
	if (!d_first_image || width <= 0 || height <= 0)
	{
		log_error("rir_bad_pixels_create_device: invalid argument");
		return 0;
	}
	auto bp = std::make_shared<BadPixelsObject>();
	bp->width = width;
	bp->height = height;
	hipStream_t st = as_stream(stream);
	if (!detect_bad_pixels(d_first_image, width, height, 5.0, *bp, st) || !upload_bad_pixels(*bp, st))
		return 0;
	return register_object(bp);
}

RIR_EXPORT int rir_bad_pixels_correct_device(int handle, const unsigned short *d_in, unsigned short *d_out, int nframes, void *stream)
{
	if (!device_ready())
		return -1;
	auto bp = lookup_as<BadPixelsObject>(handle);
	if (!bp || !d_in || !d_out || nframes <= 0)
	{
		log_error("rir_bad_pixels_correct_device: invalid handle or argument");
		return -1;
	}
	return hip_ok(launch_bad_pixels_correct(d_in, d_out, bp->width, bp->height, nframes, bp->d_xy.as<int>(), bp->count(), bp->floor_correct,
											as_stream(stream)),
				  "bad_pixels_correct")
			   ? 0
			   : -1;
}

// The filter chain of BASELINE configs[2] in one pass over the frames: bad_pixels_correct (handle > 0; 0 = skip) ->
// gaussian_filter(sigma) -> translate(offsets, strategy) -> uint16 (truncation, like astype(uint16)).  Bit-identical to
// rir_bad_pixels_correct_device + rir_gaussian_filter_u16_device + rir_translate_f32_u16_device, without the two
// intermediate frames in HBM.  strategy: "nearest", "background" / "constant"; sigma < 2.5; d_src != d_dst.
// background: HOST pointer to one uint16 (may be NULL for "nearest").
RIR_EXPORT int rir_filter_chain_device(int bad_pixels_handle, const unsigned short *d_src, unsigned short *d_dst, int w, int h, int nframes,
									   float sigma, const float *d_offsets, int per_frame_offsets, const void *background, const char *strategy,
									   void *stream)
{
	if (!device_ready())
		return -1;
	const int s = strategy_from_string(strategy);
	if ((s != TRANSLATE_NEAREST && s != TRANSLATE_CONSTANT) || w <= 0 || h <= 0 || nframes <= 0 || !d_src || !d_dst || d_src == d_dst || !d_offsets ||
		!(sigma > 0) || gaussian_radius(sigma) > 4 || (s == TRANSLATE_CONSTANT && !background))
	{
		log_error("rir_filter_chain_device: invalid argument (strategies: nearest, background; sigma < 2.5; out of place)");
		return -1;
	}
	std::shared_ptr<BadPixelsObject> bp;
	if (bad_pixels_handle > 0)
	{
		bp = lookup_as<BadPixelsObject>(bad_pixels_handle);
		if (!bp || bp->width != w || bp->height != h)
		{
			log_error("rir_filter_chain_device: invalid bad pixels handle, or created for another image size");
			return -1;
		}
	}
	if (gaussian_reference_order())
	{
		// the reference's own summation order was asked for (runtime.h): the chain as its three steps, still without leaving the device -
		// repair, gaussian_filter as the 2-D sum, translate + truncation - through stream-ordered scratch; the chain's output is then the
		// reference chain's, bit for bit (the fused kernel's separable gaussian differs by +-1 level on < 0.2 % of the pixels)
		hipStream_t st = as_stream(stream);
		const size_t npx = (size_t)w * h * nframes;
		void *d_fixed = nullptr, *d_f32 = nullptr;
		if ((bp && !hip_ok(hipMallocAsync(&d_fixed, npx * 2, st), "hipMallocAsync")) || !hip_ok(hipMallocAsync(&d_f32, npx * 4, st), "hipMallocAsync"))
		{
			if (d_fixed)
				(void)hipFreeAsync(d_fixed, st);
			return -1;
		}
		int r = 0;
		if (bp)
			r = rir_bad_pixels_correct_device(bad_pixels_handle, d_src, static_cast<unsigned short *>(d_fixed), nframes, stream);
		if (r == 0)
			r = rir_gaussian_filter_u16_device(bp ? static_cast<const unsigned short *>(d_fixed) : d_src, static_cast<float *>(d_f32), w, h, nframes, sigma, stream);
		if (r == 0)
			r = rir_translate_f32_u16_device(static_cast<const float *>(d_f32), d_dst, w, h, nframes, d_offsets, per_frame_offsets, background, strategy, stream);
		if (d_fixed)
			(void)hipFreeAsync(d_fixed, st);
		(void)hipFreeAsync(d_f32, st);
		return r;
	}
	const int nbad = bp ? bp->count() : 0;
	if (nbad > 0 && !bp->d_fix.reserve((size_t)nbad * nframes * sizeof(uint32_t)))
		return -1;
	const int radius = gaussian_radius(sigma);
	const GaussTable table = gaussian_table_device(sigma, radius); // held until the launch below has been queued
	if (!table)
		return -1;
	const float *d_k = table->as<float>();
	const uint16_t back = background ? *static_cast<const uint16_t *>(background) : (uint16_t)0;
	return hip_ok(launch_filter_chain(d_src, d_dst, w, h, nframes, bp ? bp->d_xy.as<int>() : nullptr, bp ? bp->d_row_start.as<int>() : nullptr, nbad,
									  bp ? bp->floor_correct : 0, nbad > 0 ? bp->d_fix.as<uint32_t>() : nullptr, d_k, radius, d_offsets,
									  per_frame_offsets, s, back, as_stream(stream)),
				  "filter_chain")
			   ? 0
			   : -1;
}

// info[0] = number of flagged pixels, info[1] = clamp floor (m_median_value), info[2] = detector floor;
// xy (may be NULL) receives up to cap (x,y) pairs in raster order.
RIR_EXPORT int rir_bad_pixels_info(int handle, int *info, int *xy, int cap)
{
	auto bp = lookup_as<BadPixelsObject>(handle);
	if (!bp || !info)
		return -1;
	info[0] = bp->count();
	info[1] = bp->floor_correct;
	info[2] = bp->floor_detect;
	if (xy)
		std::memcpy(xy, bp->xy.data(), sizeof(int) * 2 * (size_t)std::min(cap, bp->count()));
	return 0;
}

// Read-back variant (IRFileLoader::removeBadPixels): in place, first `rows` rows, flagged
// neighbours excluded.  The handle must have been created on a (width x rows) detector window.
RIR_EXPORT int rir_remove_bad_pixels_device(int handle, unsigned short *d_img, int rows, int nframes, void *stream)
{
	if (!device_ready())
		return -1;
	auto bp = lookup_as<BadPixelsObject>(handle);
	if (!bp || !d_img || nframes <= 0 || rows <= 0 || rows > bp->height)
	{
		log_error("rir_remove_bad_pixels_device: invalid handle or argument");
		return -1;
	}
	return hip_ok(launch_remove_bad_pixels(d_img, bp->width, bp->height, rows, nframes, bp->d_xy.as<int>(), bp->count(), bp->d_bitmap.as<uint8_t>(),
										   as_stream(stream)),
				  "remove_bad_pixels")
			   ? 0
			   : -1;
}

// Detector restricted to the first `rows` rows of a (width x height) frame, list expressed in
// full-frame coordinates: what IRFileLoader::setBadPixelsEnabled builds (IRFileLoader.cpp:693-716).
RIR_EXPORT int rir_bad_pixels_create_rows_device(const unsigned short *d_first_image, int width, int height, int rows, void *stream)
{
	if (!device_ready())
		return 0;
	if (!d_first_image || width <= 0 || height <= 0 || rows <= 0 || rows > height)
	{
		log_error("rir_bad_pixels_create_rows_device: invalid argument");
		return 0;
	}
	auto bp = std::make_shared<BadPixelsObject>();
	bp->width = width;
	bp->height = height;
	hipStream_t st = as_stream(stream);
	if (!detect_bad_pixels(d_first_image, width, rows, 5.0, *bp, st) || !upload_bad_pixels(*bp, st))
		return 0;
	return register_object(bp);
}

// Motion removal on read-back (IRFileLoader.cpp:617-627): d_shifts = float pairs (x[pos], y[pos]).
RIR_EXPORT int rir_remove_motion_device(const unsigned short *d_src, unsigned short *d_dst, int w, int h, int rows, int nframes,
										const float *d_shifts, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_src || !d_dst || d_src == d_dst || w <= 0 || h <= 0 || rows <= 0 || rows > h || nframes <= 0 || !d_shifts)
	{
		log_error("rir_remove_motion_device: invalid argument");
		return -1;
	}
	return hip_ok(launch_remove_motion(d_src, d_dst, w, h, rows, nframes, d_shifts, as_stream(stream)), "remove_motion") ? 0 : -1;
}

RIR_EXPORT int rir_median_filter_device(const unsigned short *d_src, unsigned short *d_dst, int w, int h, int nframes, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_src || !d_dst || w < 3 || h < 3 || nframes <= 0)
	{
		log_error("rir_median_filter_device: invalid argument (needs w,h >= 3)");
		return -1;
	}
	return hip_ok(launch_median3x3(d_src, d_dst, w, h, nframes, as_stream(stream)), "median_filter") ? 0 : -1;
}

// =====================================================================================================
// Reference entry points (host pointers, synchronous)
// =====================================================================================================

RIR_EXPORT int translate(int type, void *src, void *dst, int w, int h, float dx, float dy, void *background, const char *strategy)
{
	const int es = dtype_size(type);
	if (es == 0 || strategy_from_string(strategy) < 0) // signal_processing.cpp:40-41,70-71
		return -1;
	if (!device_ready())
		return -1;
	if (!src || !dst || !background || w <= 0 || h <= 0)
		return -1;
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	const size_t bytes = (size_t)w * h * es;
	hipStream_t st = default_stream();
	const bool keeps_dst = strategy_from_string(strategy) == TRANSLATE_UNCHANGED;
	const size_t off_at = (bytes + 63) & ~(size_t)63;
	if (!s.h_in.reserve(off_at + 64))
		return -1;
	// dst is an in/out buffer: "noborder" keeps whatever the caller put there (Filters.h:261-264), so only that
	// strategy needs the caller's dst where the kernel works; the others write every pixel
	float *off = reinterpret_cast<float *>(s.h_in.as<char>() + off_at); // (page-locked: the kernel reads the two floats from there)
	off[0] = dx, off[1] = dy;
	const void *in = stage_in(s, s.a, src, bytes, 0, st);
	void *out = in ? stage_out(s, s.b, dst, bytes, keeps_dst, st, in, bytes) : nullptr;
	if (!in || !out)
		return -1;
	if (rir_translate_device(type, in, out, w, h, 1, off, 0, background, strategy, st) != 0)
		return -1;
	return hand_out(s, dst, out, bytes, st) ? 0 : -1;
}

RIR_EXPORT int gaussian_filter(float *src, float *dst, int w, int h, float sigma)
{
	if (!device_ready())
		return -1;
	if (!src || !dst || w <= 0 || h <= 0)
		return -1;
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	const size_t bytes = (size_t)w * h * sizeof(float);
	hipStream_t st = default_stream();
	if (!s.h_in.reserve(bytes))
		return -1;
	const void *in = stage_in(s, s.a, src, bytes, 0, st);
	void *out = in ? stage_out(s, s.b, dst, bytes, false, st, in, bytes) : nullptr;
	if (!in || !out)
		return -1;
	if (rir_gaussian_filter_device(static_cast<const float *>(in), static_cast<float *>(out), w, h, 1, sigma, st) != 0)
		return -1;
	return hand_out(s, dst, out, bytes, st) ? 0 : -1;
}

// Extension: gaussian_filter of a uint16 image - what the reference's wrapper computes for one (it converts to float32 first,
// rir_signal_processing.py:85-113; every uint16 is a float32, so the conversion inside the kernel gives the same bits) - with half the bytes
// up the link.  Radius <= 4 (sigma < 2.5); -1 otherwise: the caller converts and takes gaussian_filter.
RIR_EXPORT int rir_gaussian_filter_u16(unsigned short *src, float *dst, int w, int h, float sigma)
{
	if (!device_ready())
		return -1;
	if (!src || !dst || w <= 0 || h <= 0 || !(sigma > 0) || gaussian_radius(sigma) > 4)
		return -1;
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	const size_t in_bytes = (size_t)w * h * 2, bytes = (size_t)w * h * sizeof(float);
	hipStream_t st = default_stream();
	if (!s.h_in.reserve(in_bytes))
		return -1;
	const void *in = stage_in(s, s.a, src, in_bytes, 0, st);
	void *out = in ? stage_out(s, s.b, dst, bytes, false, st, in, in_bytes) : nullptr;
	if (!in || !out)
		return -1;
	if (rir_gaussian_filter_u16_device(static_cast<const unsigned short *>(in), static_cast<float *>(out), w, h, 1, sigma, st) != 0)
		return -1;
	return hand_out(s, dst, out, bytes, st) ? 0 : -1;
}

static int find_median_host(unsigned short *pixels, unsigned char *mask, int size, float percent)
{
	if (!device_ready())
		return -1;
	if (!pixels || size <= 0)
		return 0;
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	hipStream_t st = default_stream();
	const size_t pbytes = (size_t)size * 2, mslot = (pbytes + 63) & ~(size_t)63;
	if (!s.b.reserve(65536 * sizeof(uint32_t)) || !s.c.reserve(sizeof(int)) || !s.h_in.reserve(mslot + (mask ? (size_t)size : 0)))
		return -1;
	const void *px = stage_in(s, s.a, pixels, pbytes, 0, st);
	const void *mk = mask ? stage_in(s, s.d, mask, (size_t)size, mslot, st) : nullptr;
	if (!px || (mask && !mk))
		return -1;
	if (rir_find_median_pixel_device(static_cast<const unsigned short *>(px), static_cast<const unsigned char *>(mk), size, 1, percent, s.c.as<int>(),
									 s.b.as<unsigned int>(), st) != 0)
		return -1;
	int res = 0;
	if (!hip_ok(hipMemcpyAsync(&res, s.c.ptr, sizeof(int), hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
		return -1;
	return res;
}

RIR_EXPORT int find_median_pixel(unsigned short *pixels, int size, float percent) { return find_median_host(pixels, nullptr, size, percent); }
RIR_EXPORT int find_median_pixel_mask(unsigned short *pixels, unsigned char *mask, int size, float percent)
{
	return find_median_host(pixels, mask, size, percent);
}

// returns the object handle, 0 on error (signal_processing.h:77-80)
RIR_EXPORT int bad_pixels_create(unsigned short *first_image, int width, int height)
{
	if (!device_ready())
		return 0;
	if (!first_image || width <= 0 || height <= 0)
		return 0;
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	hipStream_t st = default_stream();
	const size_t bytes = (size_t)width * height * 2;
	// (the detector makes several passes over the image: it goes to device memory, whatever the switch says)
	if (!s.a.reserve(bytes) || !s.h_in.reserve(bytes))
		return 0;
	host_copy(s.h_in.ptr, first_image, bytes);
	if (!hip_ok(hipMemcpyAsync(s.a.ptr, s.h_in.ptr, bytes, hipMemcpyHostToDevice, st), "H2D"))
		return 0;
	return rir_bad_pixels_create_device(s.a.as<unsigned short>(), width, height, st);
}

RIR_EXPORT int bad_pixels_correct(int handle, unsigned short *in, unsigned short *out)
{
	auto bp = lookup_as<BadPixelsObject>(handle);
	if (!bp) // signal_processing.cpp:209-211
		return -1;
	if (!device_ready())
		return -1;
	if (!in || !out)
		return -1;
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	hipStream_t st = default_stream();
	const size_t bytes = (size_t)bp->width * bp->height * 2;
	if (!s.h_in.reserve(bytes))
		return -1;
	const void *src = stage_in(s, s.a, in, bytes, 0, st);
	void *dst = src ? stage_out(s, s.b, out, bytes, false, st, src, bytes) : nullptr;
	if (!src || !dst)
		return -1;
	if (rir_bad_pixels_correct_device(handle, static_cast<const unsigned short *>(src), static_cast<unsigned short *>(dst), 1, st) != 0)
		return -1;
	return hand_out(s, out, dst, bytes, st) ? 0 : -1;
}

// Extension: bad_pixels_correct -> gaussian_filter(sigma) -> translate(dx, dy, strategy) -> uint16 on ONE host image in one call (the reference's
// callers make the three calls, three trips over the link and two float images in between: configs[2] of BASELINE).  bad_pixels_handle 0: no
// repair.  Strategies "nearest" and "background"; the result is what rir_filter_chain_device gives (the three calls' result within one
// level, or exactly with rir_set_gaussian_reference_order(1)).  0 / -1.
RIR_EXPORT int rir_filter_chain(int bad_pixels_handle, unsigned short *in, unsigned short *out, int w, int h, float sigma, float dx, float dy,
								void *background, const char *strategy)
{
	if (!device_ready())
		return -1;
	if (!in || !out || w <= 0 || h <= 0)
		return -1;
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	hipStream_t st = default_stream();
	const size_t bytes = (size_t)w * h * 2;
	const size_t off_at = (bytes + 63) & ~(size_t)63;
	if (!s.h_in.reserve(off_at + 64))
		return -1;
	float *off = reinterpret_cast<float *>(s.h_in.as<char>() + off_at); // (page-locked: the kernel reads the two floats from there)
	off[0] = dx, off[1] = dy;
	const void *src = stage_in(s, s.a, in, bytes, 0, st);
	void *dst = src ? stage_out(s, s.b, out, bytes, false, st, src, bytes) : nullptr;
	if (!src || !dst)
		return -1;
	if (rir_filter_chain_device(bad_pixels_handle, static_cast<const unsigned short *>(src), static_cast<unsigned short *>(dst), w, h, 1, sigma, off, 0,
								background, strategy, st) != 0)
		return -1;
	return hand_out(s, out, dst, bytes, st) ? 0 : -1;
}

RIR_EXPORT void bad_pixels_destroy(int handle)
{
	if (lookup_as<BadPixelsObject>(handle))
		remove_object(handle);
}

// hash_bytes (signal_processing.cpp:337-392): 64-bit multiply/xor-shift hash used by the Python
// cache helpers.  Host-side utility, restated here so the symbol exists in the drop-in library.
RIR_EXPORT size_t hash_bytes(void *_ptr, size_t len)
{
	const uint64_t m = 14313749767032793493ULL, seed = 3782874213ULL, r = 47ULL;
	const unsigned char *ptr = static_cast<const unsigned char *>(_ptr);
	uint64_t h = seed ^ (len * m);
	const size_t nblocks = len / 8;
	for (size_t i = 0; i < nblocks; ++i, ptr += 8)
	{
		uint64_t k;
		std::memcpy(&k, ptr, 8);
		k *= m;
		k ^= k >> r;
		k *= m;
		h ^= k;
		h *= m;
	}
	const size_t tail = len & 7U;
	if (tail)
	{
		for (size_t i = tail; i-- > 0;)
			h ^= (uint64_t)ptr[i] << (8 * i);
		h *= m;
	}
	h ^= h >> r;
	h *= m;
	h ^= h >> r;
	return (size_t)h;
}

// ---- connected components (label_kernels.hip) -----------------------------------------------------------------------------------
namespace
{
	int cell_bytes_of(int type)
	{ // what `==` means for the reference's cell types (signal_processing.cpp:228-263): integers by their bits, floating point as IEEE
		switch (type)
		{
		case 'f':
			return -4;
		case 'd':
			return -8;
		default:
			return dtype_size(type); // 0: unknown
		}
	}
	// `(U)background` with U = int (Filters.h:535): the host's own conversion of the cell type, at run time like the reference's
	int background_as_int(int type, const void *bg)
	{
		switch (type)
		{
		case '?':
		case 'B':
			return (int)*static_cast<const unsigned char *>(bg);
		case 'b':
			return (int)*static_cast<const signed char *>(bg);
		case 'h':
			return (int)*static_cast<const short *>(bg);
		case 'H':
			return (int)*static_cast<const unsigned short *>(bg);
		case 'i':
			return *static_cast<const int *>(bg);
		case 'I':
			return (int)*static_cast<const unsigned int *>(bg);
		case 'l':
			return (int)*static_cast<const long long *>(bg);
		case 'L':
			return (int)*static_cast<const unsigned long long *>(bg);
		case 'f':
		{
			volatile float f = *static_cast<const float *>(bg);
			return (int)f;
		}
		case 'd':
		{
			volatile double d = *static_cast<const double *>(bg);
			return (int)d;
		}
		default:
			return 0;
		}
	}
} // namespace

RIR_EXPORT size_t rir_label_workspace_bytes(int w, int h) { return label_workspace_bytes(w, h, 1); }
RIR_EXPORT size_t rir_label_workspace_bytes_batch(int w, int h, int nframes) { return label_workspace_bytes(w, h, nframes); }

// A batch of images [nframes][h][w] in device memory, every one labelled on its own (five launches for the whole batch).
RIR_EXPORT int rir_label_images_device(int type, const void *d_src, int *d_dst, int w, int h, int nframes, const void *background, double *d_xy,
									   int *d_area, int table_entries, int *d_count, void *d_work, size_t work_bytes, void *stream)
{
	const int cell = cell_bytes_of(type);
	if (cell == 0)
		return -1;
	if (!device_ready())
		return -1;
	const size_t need = label_workspace_bytes(w, h, nframes);
	if (need == 0 || work_bytes < need || ((uintptr_t)d_work & 7) || table_entries < 1)
		return -1;
	return hip_ok(launch_label_images(cell, d_src, background, w, h, nframes, d_dst, d_xy, d_area, table_entries, d_count, d_work, as_stream(stream)),
				  "label_image")
			   ? 0
			   : -1;
}
RIR_EXPORT int rir_label_image_device(int type, const void *d_src, int *d_dst, int w, int h, const void *background, double *d_xy, int *d_area,
									  int *d_count, void *d_work, size_t work_bytes, void *stream)
{
	if (w <= 0 || h <= 0 || (int64_t)w * h >= 0x7FFFFFFFLL)
		return -1;
	return rir_label_images_device(type, d_src, d_dst, w, h, 1, background, d_xy, d_area, w * h + 1, d_count, d_work, work_bytes, stream);
}

RIR_EXPORT int rir_keep_largest_areas_device(int type, const void *d_src, int *d_dst, int w, int h, int nframes, const void *background, int foreground,
											 void *d_work, size_t work_bytes, void *stream)
{
	const int cell = cell_bytes_of(type);
	if (cell == 0 || !background)
		return -1;
	if (!device_ready())
		return -1;
	const size_t need = label_workspace_bytes(w, h, nframes);
	if (need == 0 || work_bytes < need || ((uintptr_t)d_work & 7))
		return -1;
	return hip_ok(launch_keep_largest_areas(cell, d_src, background, w, h, nframes, d_dst, foreground, background_as_int(type, background), d_work,
											as_stream(stream)),
				  "keep_largest_area")
			   ? 0
			   : -1;
}
RIR_EXPORT int rir_keep_largest_area_device(int type, const void *d_src, int *d_dst, int w, int h, const void *background, int foreground,
											void *d_work, size_t work_bytes, void *stream)
{
	return rir_keep_largest_areas_device(type, d_src, d_dst, w, h, 1, background, foreground, d_work, work_bytes, stream);
}

// The image goes to device memory (the passes read it several times); the labels come back through the page-locked staging buffer
// like every other image of this file, the small tables (one entry per component) are written over the link by the kernel itself.
static int label_host(int type, void *src, int *dst, int w, int h, void *background, double *out_xy, int *out_area, bool keep, int foreground)
{
	const int es = dtype_size(type);
	if (es == 0) // signal_processing.cpp:261-262, :313-314
		return -1;
	if (!device_ready())
		return -1;
	if (!src || !dst || !background || w < 0 || h < 0 || (!keep && (!out_xy || !out_area)))
		return -1;
	if (w == 0 || h == 0)
	{ // no pixel: the table holds the background's entry alone (Filters.h:489, Label() = (-1, -1), area 0)
		if (keep)
			return 0;
		out_xy[0] = out_xy[1] = -1.0;
		out_area[0] = 0;
		return 1;
	}
	const size_t work = label_workspace_bytes(w, h, 1);
	if (work == 0)
	{
		log_error("label_image: image too large");
		return -1;
	}
	HostScratch &s = scratch();
	std::lock_guard<std::mutex> g(s.mu);
	hipStream_t st = default_stream();
	const size_t n = (size_t)w * h, in_bytes = n * es, out_bytes = n * sizeof(int);
	const size_t xy_bytes = (n + 1) * 2 * sizeof(double), area_bytes = (n + 1) * sizeof(int);
	if (!s.h_in.reserve(in_bytes) || !s.a.reserve(in_bytes) || !s.c.reserve(work) || (!keep && !s.h_tab.reserve(xy_bytes + area_bytes + 64)))
		return -1;
	host_copy(s.h_in.ptr, src, in_bytes);
	if (!hip_ok(hipMemcpyAsync(s.a.ptr, s.h_in.ptr, in_bytes, hipMemcpyHostToDevice, st), "H2D"))
		return -1;
	void *out = stage_out(s, s.b, dst, out_bytes, false, st);
	if (!out)
		return -1;
	if (keep)
	{
		if (rir_keep_largest_area_device(type, s.a.ptr, static_cast<int *>(out), w, h, background, foreground, s.c.ptr, s.c.cap, st) != 0)
			return -1;
		return hand_out(s, dst, out, out_bytes, st) ? 0 : -1;
	}
	double *xy = s.h_tab.as<double>();
	int *area = reinterpret_cast<int *>(s.h_tab.as<char>() + xy_bytes);
	int *count = reinterpret_cast<int *>(s.h_tab.as<char>() + xy_bytes + area_bytes);
	*count = -1;
	if (rir_label_image_device(type, s.a.ptr, static_cast<int *>(out), w, h, background, xy, area, count, s.c.ptr, s.c.cap, st) != 0)
		return -1;
	if (!hand_out(s, dst, out, out_bytes, st)) // (waits for the stream: the tables are complete)
		return -1;
	const int labels = *count;
	if (labels < 1 || (size_t)labels > n + 1)
		return -1;
	std::memcpy(out_xy, xy, (size_t)labels * 2 * sizeof(double));
	std::memcpy(out_area, area, (size_t)labels * sizeof(int));
	return labels;
}

// reference signal_processing.cpp:224-266: the number of table entries (components + 1), -1 on an unknown type.  out_xy / out_area
// receive that many entries (the caller sizes them; one entry per pixel plus one is always enough).
RIR_EXPORT int label_image(int type, void *src, int *dst, int w, int h, void *background, double *out_xy, int *out_area)
{
	return label_host(type, src, dst, w, h, background, out_xy, out_area, false, 0);
}
// reference signal_processing.cpp:276-318: 0, -1 on an unknown type
RIR_EXPORT int keep_largest_area(int type, void *src, int *dst, int w, int h, void *background, int foreground)
{
	return label_host(type, src, dst, w, h, background, nullptr, nullptr, true, foreground);
}
