/* librir_amd — the signal_processing C ABI of librir, served by HIP kernels on MI355X.
 *
 * Same symbol names, argument meaning and return codes as the reference header
 * src/cpp/signal_processing/signal_processing.h (cited per function), so that the library can be
 * loaded by librir's Python wrapper in place of libsignal_processing.so.
 * Host pointers in, host pointers out, synchronous.  No CPU fallback: every entry point that touches pixels runs on the device or
 * fails; extract_times / resample_time_serie / hash_bytes are host bookkeeping on timestamps and bytes and need none.
 */
#ifndef RIR_AMD_SIGNAL_PROCESSING_H
#define RIR_AMD_SIGNAL_PROCESSING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

	/* reference signal_processing.h:29 — 0 on success, -1 on unknown dtype char / strategy.
	 * strategy: NULL, "", "noborder", "background", "wrap", "nearest".  dst is in/out. */
	int translate(int type, void *src, void *dst, int w, int h, float dx, float dy, void *background, const char *strategy);

	/* reference signal_processing.h:33 */
	int gaussian_filter(float *src, float *dst, int w, int h, float sigma);

	/* reference signal_processing.h:39,44 (the C++ default argument percent = 0.5 is explicit here) */
	int find_median_pixel(unsigned short *pixels, int size, float percent);
	int find_median_pixel_mask(unsigned short *pixels, unsigned char *mask, int size, float percent);

	/* reference signal_processing.h:80,84,88 — create returns the handle (>0) or 0 on error */
	int bad_pixels_create(unsigned short *first_image, int width, int height);
	int bad_pixels_correct(int handle, unsigned short *in, unsigned short *out);
	void bad_pixels_destroy(int handle);

	/* reference signal_processing.h:94 */
	size_t hash_bytes(void *ptr, size_t len);

	/* reference signal_processing.h:55 / signal_processing.cpp:158-181 — one time axis out of several (s: 0 union, 1 intersection).
	 * time_vector: the vectors one after the other, vector_sizes[vector_count] their lengths.  0; -2 with the needed size in
	 * *output_size when the output is too small; -1 on input the reference does not return from (csrc/time_series.cpp). */
	int extract_times(double *time_vector, int vector_count, int *vector_sizes, int s, double *output, int *output_size);
	/* reference signal_processing.h:71 / signal_processing.cpp:183-195 — (sample_x, sample_y) read at `times` (s: 2 pad with padds
	 * outside the samples, 4 interpolate).  0; -1 when the output is too small (needed size in *output_size). */
	int resample_time_serie(double *sample_x, double *sample_y, int size, double *times, int times_size, int s, double padds, double *output,
							int *output_size);
	/* reference signal_processing.h:90 / signal_processing.cpp:224-266 — connected components of `src` (cells != *background; joined
	 * vertically whatever their values, horizontally when equal), numbered from 1 in raster order of their first pixel into dst.
	 * Returns the number of table entries (components + 1; entry 0 is the background's) written to out_xy (x of the first pixel, twice -
	 * as upstream) and out_area, -1 on an unknown type. */
	int label_image(int type, void *src, int *dst, int w, int h, void *background, double *out_xy, int *out_area);
	/* reference signal_processing.h:92 / signal_processing.cpp:276-318 — dst = foreground on the largest component, (int)*background
	 * elsewhere (all zero without a component).  0, -1 on an unknown type. */
	int keep_largest_area(int type, void *src, int *dst, int w, int h, void *background, int foreground);

	/* ---- extension (prefix rir_) ----
	 * bad_pixels_correct -> gaussian_filter(sigma) -> translate(dx, dy, strategy) -> uint16 on one host image in ONE call: what a caller of the
	 * three entry points above does in three (BASELINE configs[2]), without the two float images in between.  bad_pixels_handle 0: no
	 * repair; strategy "nearest" or "background" (background: one uint16).  The three calls' result within one level, or exactly with
	 * rir_set_gaussian_reference_order(1) (rir_amd_device.h).  0 / -1. */
	int rir_filter_chain(int bad_pixels_handle, unsigned short *in, unsigned short *out, int w, int h, float sigma, float dx, float dy,
						 void *background, const char *strategy);

#ifdef __cplusplus
}
#endif
#endif /* RIR_AMD_SIGNAL_PROCESSING_H */
