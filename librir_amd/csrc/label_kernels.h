// Host-side launchers of the connected-component kernels (label_kernels.hip).  C++ linkage, internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rir
{
	// Working memory of one labelling of `frames` images (device memory, 8-byte aligned), in bytes; 0: geometry refused.
	size_t label_workspace_bytes(int w, int h, int frames);
	// A batch of images [frames][h][w], every one labelled on its own.
	// cell_bytes: 1, 2, 4, 8 (integers: equal when their bits are), -4 float, -8 double (IEEE ==).  `background`: the cell value, host memory.
	// d_dst [frames][h][w] int32 labels; per image `table_entries` entries of d_xy (2 doubles each) and d_area, of which the first
	// min(count, table_entries) are written; d_count [frames] (= components + 1, the reference's return value): any memory the device can
	// write (device or page-locked host).
	hipError_t launch_label_images(int cell_bytes, const void *d_src, const void *background, int w, int h, int frames, int *d_dst, double *d_xy,
								   int *d_area, int64_t table_entries, int *d_count, void *d_work, hipStream_t st);
	// d_dst [frames][h][w]: `foreground` on the largest component of each image (the first in raster order among equals),
	// `background_as_int` elsewhere; all zero when an image holds no component.
	hipError_t launch_keep_largest_areas(int cell_bytes, const void *d_src, const void *background, int w, int h, int frames, int *d_dst, int foreground,
										 int background_as_int, void *d_work, hipStream_t st);
} // namespace rir
