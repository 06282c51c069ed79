// Host-side launchers of the connected-component kernels (label_kernels.hip).  C++ linkage, internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rir
{
	// Working memory of one labelling (device memory, 8-byte aligned), in bytes.
	size_t label_workspace_bytes(int w, int h);
	// cell_bytes: 1, 2, 4, 8 (integers: equal when their bits are), -4 float, -8 double (IEEE ==).  `background`: the cell value, host memory.
	// d_dst [h][w] int32 labels; d_xy [2*(count+1)] doubles, d_area [count+1], d_count [1] (= count + 1, the reference's return value):
	// any memory the device can write (device or page-locked host).
	hipError_t launch_label_image(int cell_bytes, const void *d_src, const void *background, int w, int h, int *d_dst, double *d_xy, int *d_area,
								  int *d_count, void *d_work, hipStream_t st);
	// d_dst [h][w]: `foreground` on the largest component (the first in raster order among equals), `background_as_int` elsewhere; all
	// zero when the image holds no component.
	hipError_t launch_keep_largest_area(int cell_bytes, const void *d_src, const void *background, int w, int h, int *d_dst, int foreground,
										int background_as_int, void *d_work, hipStream_t st);
} // namespace rir
