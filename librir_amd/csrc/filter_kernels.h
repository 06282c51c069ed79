// Host-side launchers of the frame-buffer kernels (filter_kernels.hip).  C++ linkage, internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rir
{
	enum
	{
		TRANSLATE_UNCHANGED = 0, // reference TranslateUnchanged, Filters.h:238-244
		TRANSLATE_CONSTANT = 1,
		TRANSLATE_WRAP = 2,
		TRANSLATE_NEAREST = 3,
		TRANSLATE_SOURCE = 4 // device layer only: pixels without a source take the input pixel at the same position, i.e.
							 // "noborder" on a destination pre-filled with a copy of the input (what the wrappers do,
							 // reference rir_signal_processing.py:54-55) without the copy
	};

	hipError_t launch_translate(int type, const void *src, void *dst, const void *background, int w, int h, int nframes, const float *d_offsets,
								int per_frame, int strategy, hipStream_t st);
	hipError_t launch_remove_motion(const uint16_t *src, uint16_t *dst, int w, int h, int rows, int nframes, const float *d_shifts, hipStream_t st);
	hipError_t launch_gaussian(const float *src, float *dst, int w, int h, int nframes, const float *d_kernel, int radius, hipStream_t st);
	hipError_t launch_gaussian_u16(const uint16_t *src, float *dst, int w, int h, int nframes, const float *d_kernel, int radius, hipStream_t st);
	hipError_t launch_bad_pixels_correct(const uint16_t *in, uint16_t *out, int w, int h, int nframes, const int *d_xy, int nbad, int floor_v,
										 hipStream_t st);
	hipError_t launch_filter_chain(const uint16_t *src, uint16_t *dst, int w, int h, int nframes, const int *d_xy, const int *d_row_start, int nbad,
								   int floor_v, uint32_t *d_fix, const float *d_kernel, int radius, const float *d_offsets, int per_frame,
								   int strategy, uint16_t background, hipStream_t st);
	hipError_t launch_remove_bad_pixels(uint16_t *img, int w, int h, int rows, int nframes, const int *d_xy, int nbad, const uint8_t *d_bitmap,
										hipStream_t st);
	hipError_t launch_histogram(const uint16_t *img, const uint8_t *mask, int64_t npx, int nframes, uint32_t *d_hist, hipStream_t st);
	hipError_t launch_quantile_select(const uint16_t *img, const uint8_t *mask, int64_t npx, int nframes, float percent, int nbins, int *d_result,
									  hipStream_t st);
	hipError_t launch_bad_pixels_stats(const uint32_t *d_hist, uint64_t size, int64_t *d_out, hipStream_t st);
	hipError_t launch_bad_pixels_detect(const uint16_t *src, int w, int h, double std_factor, int floor_detect, uint8_t *d_flags, hipStream_t st);
	hipError_t launch_median3x3(const uint16_t *src, uint16_t *dst, int w, int h, int nframes, hipStream_t st);
	hipError_t launch_split_planes(const uint16_t *img, const uint8_t *it, int w, int h, int nframes, int linesize, uint8_t *Y, uint8_t *U,
								   uint8_t *V, hipStream_t st);
	hipError_t launch_merge_planes(const uint8_t *Y, const uint8_t *U, const uint8_t *V, int linesize, int w, int h, int nframes, uint16_t *img,
								   uint8_t *it, hipStream_t st);
	hipError_t launch_u16_to_f32(const uint16_t *src, float *dst, int64_t total, hipStream_t st);
	hipError_t launch_stream_copy_probe(const void *src, void *dst, int64_t bytes, hipStream_t st); // 16-byte aligned buffers, bytes a multiple of 16

	// codec_kernels.hip
	hipError_t launch_encode_tiles(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint64_t *d_hdr,
								   uint32_t *d_seg_words, uint64_t *d_sparse, hipStream_t st);
	hipError_t launch_encode_compact(int ntiles, int nframes, int gop, const uint32_t *d_seg_words, const uint64_t *d_sparse,
									 uint32_t *d_tile_off, uint64_t *d_chunk_words, uint64_t *d_chunk_off, uint64_t *d_stream, hipStream_t st);
	int64_t encode_ctrl_bytes(int nchunks, int ntiles);
	hipError_t launch_encode_dense(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint64_t *d_hdr, uint32_t *d_tile_off,
								   uint64_t *d_chunk_off, uint64_t *d_stream, uint64_t *d_ctrl, uint64_t *d_spill, hipStream_t st);
	hipError_t launch_decode(const uint64_t *d_hdr, const uint32_t *d_tile_off, const uint64_t *d_chunk_off, const uint64_t *d_stream,
							 uint64_t stream_words, int64_t npx, int ntiles, int nframes, int gop, const int64_t *d_chunk_frames, int nchunks_tab,
							 uint16_t *d_frames, int *d_error, hipStream_t st);
	hipError_t launch_decode_slots(const uint64_t *d_hdr, const uint32_t *d_seg_words, const uint64_t *d_slots, int64_t npx, int ntiles, int nframes,
								   int gop, uint16_t *d_frames, int *d_error, hipStream_t st);
	// the packed form (codec_kernels.hip: rirb1_encode_packed)
	hipError_t launch_encode_packed(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint64_t *d_hdr, uint64_t *d_seg_pos,
									uint32_t *d_seg_words, uint64_t *d_stream, uint64_t capacity_words, uint64_t *d_ctrl, uint64_t *d_arena,
									uint64_t arena_words, bool reset, hipStream_t st); // reset: zero the control block first (else the caller has)
	hipError_t launch_decode_packed(const uint64_t *d_hdr, const uint64_t *d_seg_pos, const uint32_t *d_seg_words, const uint64_t *d_stream,
									uint64_t stream_words, int64_t npx, int ntiles, int nframes, int gop, uint16_t *d_frames, int *d_error, hipStream_t st);
} // namespace rir
