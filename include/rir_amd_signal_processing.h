/* librir_amd — the signal_processing C ABI of librir, served by HIP kernels on MI355X.
 *
 * Same symbol names, argument meaning and return codes as the reference header
 * src/cpp/signal_processing/signal_processing.h (cited per function), so that the library can be
 * loaded by librir's Python wrapper in place of libsignal_processing.so.
 * Host pointers in, host pointers out, synchronous.  No CPU fallback.
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

	/* reference signal_processing.h:55,71,90,92 — CPU utilities outside the accelerated path
	 * (SURVEY.md §8): the symbols resolve, the calls log an error and return -1. */
	int extract_times(double *time_vector, int vector_count, int *vector_sizes, int s, double *output, int *output_size);
	int resample_time_serie(double *sample_x, double *sample_y, int size, double *times, int times_size, int s, double padds, double *output,
							int *output_size);
	int label_image(int type, void *src, int *dst, int w, int h, void *background, double *out_xy, int *out_area);
	int keep_largest_area(int type, void *src, int *dst, int w, int h, void *background, int foreground);

#ifdef __cplusplus
}
#endif
#endif /* RIR_AMD_SIGNAL_PROCESSING_H */
