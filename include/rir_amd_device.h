/* librir_amd — device-resident batch entry points (extension of the librir C ABI).
 *
 * The reference ABI moves one frame per call through host pointers (reference
 * src/cpp/video_io/video_io.h:102,259 and src/cpp/signal_processing/signal_processing.h:29-88).
 * These entry points are the same operations on batches that already live in HBM, laid out
 * [n][h][w] row-major, asynchronous on a caller-supplied HIP stream (`stream` is a hipStream_t
 * passed as void*; NULL = HIP's null stream).  The per-frame reference entry points in
 * rir_amd_signal_processing.h / rir_amd_video_io.h are thin wrappers over these.
 *
 * All functions return 0 on success and -1 on error (reason via get_last_log_error) unless
 * stated otherwise.  No function falls back to the CPU: without a HIP device they fail.
 */
#ifndef RIR_AMD_DEVICE_H
#define RIR_AMD_DEVICE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

	/* 1 when a HIP device is visible, else 0 (never logs). */
	int rir_device_available(void);
	int rir_stream_synchronize(void *stream);

	/* The sizing rules of kernels whose workgroups wait for each other inside a launch ("resident launches", librir_amd/csrc/runtime.h),
	 * as plain host functions (no device needed; what the CPU tests check): workgroups of a kernel with blocks_per_cu resident
	 * workgroups per CU that one launch may hold on a device of `cus` CUs in `xcds` XCDs; and how `units` independent units (streams,
	 * sequences) of wgs_per_unit workgroups go through a kernel of that capacity: out2[0] = units per launch (0: a unit does not fit - the
	 * launch-per-frame / launch-per-iteration kernels are used), out2[1] = launches. */
	int rir_resident_capacity_rule(int blocks_per_cu, int cus, int xcds);
	int rir_resident_plan(int capacity, int wgs_per_unit, int units, int *out2);
	int rir_resident_plan_two_forms(int capacity_a, int capacity_b, int wgs_per_unit, int units, int *out3); /* a kernel in two forms: [2] = 1 when the larger, slower one saves a launch */

	/* ---- block codec (format RIRB1) ----------------------------------------------------------
	 * Replaces, for device-resident batches, the encode/decode the reference delegates to
	 * libx264 (reference src/cpp/video_io/h264.cpp:1022-1131 AddFrame, :3096-3229 GetFrame).
	 * A batch of nframes frames is cut in chunks of `gop` frames (key frame first, reference
	 * cadence h264.cpp:1052-1064); chunks and 512-pixel tiles are independent units. */
	typedef struct rir_codec_layout
	{
		int width, height, nframes, gop;
		int ntiles;				  /* ceil(width*height / 512) */
		int nchunks;			  /* ceil(nframes / gop) */
		int64_t hdr_bytes;		  /* uint64 [nchunks][ntiles][gop]   record headers (widths|mode|base) */
		int64_t tile_off_bytes;	  /* uint32 [nchunks][ntiles+1]      first word of a tile segment */
		int64_t chunk_off_bytes;  /* uint64 [nchunks+1]              first word of a chunk; last = total */
		int64_t stream_max_bytes; /* worst-case size of the compact stream */
		int64_t workspace_bytes;  /* scratch needed by rir_codec_encode_device */
	} rir_codec_layout;

	int rir_codec_layout_query(int width, int height, int nframes, int gop, rir_codec_layout *out);

	/* d_frames: uint16 [nframes][height][width].  Outputs sized per rir_codec_layout. */
	int rir_codec_encode_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
								unsigned int *d_tile_off, unsigned long long *d_chunk_off, unsigned long long *d_stream, void *d_workspace,
								long long workspace_bytes, void *stream);

	/* The two stages of rir_codec_encode_device, callable separately (same workspace): stage 1 is
	 * the single pass over the raw frames (headers + sparse payload), stage 2 computes the
	 * offsets and gathers the payload into the dense stream. */
	int rir_codec_encode_tiles_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
									  void *d_workspace, long long workspace_bytes, void *stream);
	int rir_codec_encode_compact_device(int width, int height, int nframes, int gop, unsigned int *d_tile_off, unsigned long long *d_chunk_off,
										unsigned long long *d_stream, void *d_workspace, long long workspace_bytes, void *stream);

	/* The SLOTTED form of an encoded batch: what stage 1 (rir_codec_encode_tiles_device) leaves - record headers in d_hdr, and in
	 * the workspace one length per (chunk, tile) segment plus every segment's payload at the start of a slot of slot_words
	 * 64-bit words whose place, slot index = chunk * ntiles + tile, depends on no length.  It is a complete encoded batch: the
	 * same records as the dense form, located by position instead of by the offsets of stage 2 (which exists to make the
	 * dense FILE form: chunk independence h264.cpp:1052-1064; offset-table precedent ZFile.cpp:434-447).  A consumer on the
	 * same device decodes it as it is - encode + decode then move 4WH + 2C bytes, the algorithmic minimum, in two launches. */
	typedef struct rir_codec_slots
	{
		int64_t slots_offset_bytes;		/* first slot, from d_workspace */
		int64_t slot_words;				/* distance of two slots, 64-bit words */
		int64_t seg_words_offset_bytes; /* uint32 [nchunks][ntiles] segment lengths in words, from d_workspace */
		int64_t nslots;					/* nchunks * ntiles */
	} rir_codec_slots;
	int rir_codec_slots_query(int width, int height, int nframes, int gop, rir_codec_slots *out);
	int rir_codec_decode_slots_device(const unsigned long long *d_hdr, const void *d_workspace, long long workspace_bytes, int width, int height,
									  int nframes, int gop, unsigned short *d_frames, int *d_error, void *stream);

	/* An encoder workspace (workspace_bytes of rir_codec_layout) allocated BY THE LIBRARY where the packing kernel runs fast for
	 * THIS frames buffer.  On MI355X device allocations fall into a few placement classes and the kernel, which reads the frames
	 * and writes the slots at the same pace, takes 10 % longer when both live in allocations of one class (DESIGN.md §7,
	 * profiles/r03_placement/README.md); nothing but a timing tells the class, so this call allocates up to max_tries further
	 * candidates, spacing_bytes apart (0: back to back), times stage 1 on each with HIP events and keeps the first one that is
	 * 7 % faster than the first, else the fastest; everything else it allocated is freed before it returns.  times_us: HOST
	 * float[max_tries + 1] or NULL (the kept candidate's time first), *ntimes = entries filled.  One-off set-up, a few ms. */
	int rir_codec_workspace_create_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, int max_tries,
										  long long spacing_bytes, void **d_workspace, float *times_us, int *ntimes, void *stream);
	void rir_codec_workspace_destroy_device(void *d_workspace);

	/* The same placement for ANY pair of buffers that a kernel walks at the same pace - the input and the output of the frame-buffer
	 * kernels (translate, gaussian_filter, the fused chain, the 3x3 median: 5-10 % between the classes): `bytes` of device memory in another
	 * placement class than d_other, found by timing a plain streaming copy from d_other into each candidate.  Arguments as for
	 * rir_codec_workspace_create_device; release with rir_buffer_destroy_device. */
	int rir_buffer_create_beside_device(const void *d_other, long long other_bytes, long long bytes, int max_tries, long long spacing_bytes,
										void **d_buffer, float *times_us, int *ntimes, void *stream);
	void rir_buffer_destroy_device(void *d_buffer);

	/* The encode as one kernel that writes the dense stream directly (segments staged in LDS, decoupled look-back): same
	 * outputs bit for bit, fewer bytes through HBM, not faster on MI355X.  rir_codec_encode_status (waits for the
	 * stream): 0 = the last single-pass encode on this workspace completed, 1 = a look-back gave up, the stream is incomplete. */
	int rir_codec_encode_single_pass_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
											unsigned int *d_tile_off, unsigned long long *d_chunk_off, unsigned long long *d_stream,
											void *d_workspace, long long workspace_bytes, void *stream);
	int rir_codec_encode_status(const void *d_workspace, void *stream);

	/* The PACKED form of an encoded batch: the dense stream without the order.  One pass over the frames (one kernel), and
	 * the encoded batch occupies exactly its payload + tables - what can be kept, sent between devices or written:
	 *   d_hdr        uint64 [nchunks][ntiles][gop]  record headers, as in every form
	 *   d_seg_pos    uint64 [nchunks][ntiles]       first 64-bit word of segment (chunk, tile) in d_stream
	 *   d_seg_words  uint32 [nchunks][ntiles]       its length in words
	 *   d_stream     two extents without holes, [0, low) and [capacity - high, capacity): the segments back to back in ORDER OF
	 *                ARRIVAL (the order differs from run to run, every segment's words are the canonical ones) - a table of
	 *                positions instead of an order, as the reference's own ZFile container keeps for its records (ZFile.cpp:434-447)
	 * A segment is staged in LDS and, when its length is known, handed the next free words by one atomic add: no second pass
	 * (rir_codec_encode_compact_device), no look-back (rir_codec_encode_single_pass_device), no worst-case slots (the slotted form).
	 * Two cursors because one is a bottleneck (12 800 returning adds on one address: 16 ns each), from the two ENDS of the
	 * capacity so that they share it.  stream_capacity_words is what the caller provides at d_stream - a budget, not a worst case
	 * (stream_budget_bytes = 8 bits per pixel; the reference documents a factor of about 5, docs/video_io.md:13): a batch that
	 * needs more is not written beyond it, the status says so and how many words it needs.  The workspace holds the control
	 * block and an arena for what does not fit a wave's LDS staging (noisy data only): workspace_min_bytes always works for
	 * data within the budget, workspace_max_bytes for any data.
	 * rir_codec_encode_packed_status (waits for the stream): out[0] = low, out[1] = high (words; the batch needs low + high),
	 * out[2] = arena words asked for; returns 0 = complete, 1 = stream capacity exceeded, 2 = arena exceeded (3 = both), -1 = error.
	 * Decoding checks every segment against stream_words before it reads through it. */
	typedef struct rir_codec_packed_layout
	{
		int ntiles, nchunks;
		int64_t hdr_bytes, seg_pos_bytes, seg_words_bytes;
		int64_t stream_budget_bytes; /* 8 bits per pixel: half the raw size */
		int64_t stream_max_bytes;	 /* any data fits */
		int64_t workspace_min_bytes, workspace_max_bytes;
	} rir_codec_packed_layout;
	int rir_codec_packed_query(int width, int height, int nframes, int gop, rir_codec_packed_layout *out);
	int rir_codec_encode_packed_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
									   unsigned long long *d_seg_pos, unsigned int *d_seg_words, unsigned long long *d_stream,
									   long long stream_capacity_words, void *d_workspace, long long workspace_bytes, void *stream);
	int rir_codec_encode_packed_status(const void *d_workspace, unsigned long long *out3, void *stream);
	/* rir_codec_encode_packed_device in its two halves (the reset of the workspace's control block is a fill launch of its own): for callers
	 * that time the packing kernel alone.  _launch_ packs into a workspace that has just been reset on the same stream. */
	int rir_codec_packed_reset_device(void *d_workspace, long long workspace_bytes, void *stream);
	int rir_codec_encode_packed_launch_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
											  unsigned long long *d_seg_pos, unsigned int *d_seg_words, unsigned long long *d_stream,
											  long long stream_capacity_words, void *d_workspace, long long workspace_bytes, void *stream);
	int rir_codec_decode_packed_device(const unsigned long long *d_hdr, const unsigned long long *d_seg_pos, const unsigned int *d_seg_words,
									   const unsigned long long *d_stream, long long stream_words, int width, int height, int nframes, int gop,
									   unsigned short *d_frames, int *d_error, void *stream);

	/* *d_error (device int, zero it first) becomes 1 when a malformed table/record was met.  stream_words = number of
	 * 64-bit words readable at d_stream: the tables are untrusted (they may come from a file) and a (chunk, tile)
	 * segment that does not lie inside [0, stream_words) is rejected before anything is read through it. */
	int rir_codec_decode_device(const unsigned long long *d_hdr, const unsigned int *d_tile_off, const unsigned long long *d_chunk_off,
								const unsigned long long *d_stream, long long stream_words, int width, int height, int nframes, int gop,
								unsigned short *d_frames, int *d_error, void *stream);

	/* Decode of chunks that do not form one contiguous batch - chunks gathered from several shards (the exchange step of
	 * the multi-GPU path, reference chunk independence h264.cpp:1052-1064), or a selection of a file's chunks.  Entry k of
	 * the tables is one chunk (hdr[k][ntiles][gop], tile_off[k][ntiles+1], chunk_off[k..k+1]); d_chunk_frames[2k] /
	 * [2k+1] = first frame / frame count (<= gop, 0 = skip) of that chunk inside d_frames (frames_capacity frames). */
	int rir_codec_decode_chunks_device(const unsigned long long *d_hdr, const unsigned int *d_tile_off, const unsigned long long *d_chunk_off,
									   const unsigned long long *d_stream, long long stream_words, int width, int height, int nchunks, int gop,
									   const long long *d_chunk_frames, long long frames_capacity, unsigned short *d_frames, int *d_error,
									   void *stream);

	/* gaussian_filter as the 2-D sum in the reference's own order (signal_processing.cpp:101-148: dx outer, dy inner, a rounding per product
	 * and per sum) instead of the separable form: results bit-identical to the reference's instead of within 2e-6 of them, in gaussian_filter,
	 * rir_gaussian_filter*_device and rir_filter_chain_device (which then runs its three steps one after the other on the device), at about
	 * three times the time.  Process-wide, default off; also RIR_GAUSSIAN_REFERENCE_ORDER=1 in the environment. */
	void rir_set_gaussian_reference_order(int on);
	int rir_gaussian_reference_order(void);

	/* ---- frame-buffer kernels -------------------------------------------------------------------
	 * translate: reference signal_processing.h:29 / Filters.h:249-326.  `type` is the numpy
	 * dtype char ('?','b','B','h','H','i','I','l','L','f','d'); d_offsets holds float (dx,dy)
	 * pairs - one pair per frame when per_frame_offsets != 0, else a single pair; background
	 * is a HOST pointer to one element; d_dst must be pre-filled by the caller (strategy
	 * "noborder" leaves border pixels as they are).  The extra strategy "noborder_source" gives the result of
	 * "noborder" on a destination pre-filled with a copy of the input - what the Python wrappers do
	 * (rir_signal_processing.py:54-55) - without that copy: d_dst need not be initialised. */
	int rir_translate_device(int type, const void *d_src, void *d_dst, int w, int h, int nframes, const float *d_offsets, int per_frame_offsets,
							 const void *background, const char *strategy, void *stream);

	/* gaussian_filter: reference signal_processing.h:33 / signal_processing.cpp:79-148. */
	int rir_gaussian_filter_device(const float *d_src, float *d_dst, int w, int h, int nframes, float sigma, void *stream);

	/* The same two filters with the dtype conversions the Python callers do around them folded in (no
	 * converted copy of the frames in memory): gaussian_filter(img.astype(float32)) for uint16 frames
	 * (sigma < 2.5), and translate(float32 img, ...).astype(uint16) - the value is rounded to float first,
	 * then truncated, exactly as the two-step form.  background: HOST pointer to one uint16. */
	int rir_gaussian_filter_u16_device(const unsigned short *d_src, float *d_dst, int w, int h, int nframes, float sigma, void *stream);
	int rir_translate_f32_u16_device(const float *d_src, unsigned short *d_dst, int w, int h, int nframes, const float *d_offsets,
									 int per_frame_offsets, const void *background, const char *strategy, void *stream);

	/* The filter chain callers run before recording (reference tests/python/test_rir.py and BASELINE configs[2]:
	 * BadPixels.correct (BadPixels.cpp:34-66) -> gaussian_filter (signal_processing.cpp:101-148) -> translate
	 * (Filters.h:249-326) -> astype(uint16)) fused into ONE pass over the frames: 4 bytes of HBM traffic per pixel
	 * instead of 14.  Output bit-identical to rir_bad_pixels_correct_device + rir_gaussian_filter_u16_device +
	 * rir_translate_f32_u16_device.  bad_pixels_handle: from rir_bad_pixels_create_device, or 0 to skip that stage.
	 * strategy "nearest" or "background"/"constant" (background: HOST pointer to one uint16); sigma < 2.5;
	 * d_src != d_dst; offsets as for rir_translate_device.  The repair table lives in the bad-pixels object: one call at a
	 * time per handle. */
	int rir_filter_chain_device(int bad_pixels_handle, const unsigned short *d_src, unsigned short *d_dst, int w, int h, int nframes, float sigma,
								const float *d_offsets, int per_frame_offsets, const void *background, const char *strategy, void *stream);

	/* find_median_pixel[_mask]: reference signal_processing.h:39-44 / Filters.cpp:56-101.
	 * d_result: int32[nframes]; d_hist: unused, may be NULL (the counting happens in LDS); d_mask may be NULL. */
	int rir_find_median_pixel_device(const unsigned short *d_img, const unsigned char *d_mask, int size, int nframes, float percent, int *d_result,
									 unsigned int *d_hist, void *stream);

	/* bad pixels: reference signal_processing.h:80-88 / BadPixels.cpp:13-66.  The handle returned
	 * by the *_create_* functions is in the same int namespace as every other librir object and
	 * is released with bad_pixels_destroy().  Returns 0 on error. */
	int rir_bad_pixels_create_device(const unsigned short *d_first_image, int width, int height, void *stream);
	int rir_bad_pixels_correct_device(int handle, const unsigned short *d_in, unsigned short *d_out, int nframes, void *stream);
	int rir_bad_pixels_info(int handle, int *info3, int *xy, int cap);

	/* read-back path of the loader: reference IRFileLoader.cpp:693-716 (detector on the first
	 * `rows` rows), :722-802 (in-place repair), :617-627 (motion removal, d_shifts = (x,y) float
	 * pairs per frame; d_dst != d_src). */
	int rir_bad_pixels_create_rows_device(const unsigned short *d_first_image, int width, int height, int rows, void *stream);
	int rir_remove_bad_pixels_device(int handle, unsigned short *d_img, int rows, int nframes, void *stream);
	int rir_remove_motion_device(const unsigned short *d_src, unsigned short *d_dst, int w, int h, int rows, int nframes, const float *d_shifts,
								 void *stream);

	/* 3x3 median filter: reference Filters.h:71-129 (template without C export upstream). */
	int rir_median_filter_device(const unsigned short *d_src, unsigned short *d_dst, int w, int h, int nframes, void *stream);

	/* connected components: reference signal_processing.h:90-92 / Filters.h:365-540 (labelImage, keepLargestArea) on images in device memory,
	 * [nframes][h][w], every image labelled on its own; five launches for the whole batch.  type: the reference's dtype character;
	 * background: HOST pointer to one cell of that type.  d_dst int32 [nframes][h][w].
	 * label: per image `table_entries` entries of d_xy (two doubles each: the first cell's x, twice - as upstream) and d_area, of which the
	 * first min(labels, table_entries) are written; d_count int[nframes] = labels = components + 1 (w*h + 1 entries are always enough);
	 * any memory the device can write.  d_work: device memory, 8-byte aligned, at least rir_label_workspace_bytes_batch(w, h, nframes)
	 * (0: geometry refused).  0 / -1.  The *_image_* forms are the batch of one with tables of w*h + 1 entries. */
	size_t rir_label_workspace_bytes(int w, int h);
	size_t rir_label_workspace_bytes_batch(int w, int h, int nframes);
	int rir_label_images_device(int type, const void *d_src, int *d_dst, int w, int h, int nframes, const void *background, double *d_xy, int *d_area,
								int table_entries, int *d_count, void *d_work, size_t work_bytes, void *stream);
	int rir_label_image_device(int type, const void *d_src, int *d_dst, int w, int h, const void *background, double *d_xy, int *d_area,
							   int *d_count, void *d_work, size_t work_bytes, void *stream);
	int rir_keep_largest_areas_device(int type, const void *d_src, int *d_dst, int w, int h, int nframes, const void *background, int foreground,
									  void *d_work, size_t work_bytes, void *stream);
	int rir_keep_largest_area_device(int type, const void *d_src, int *d_dst, int w, int h, const void *background, int foreground, void *d_work,
									 size_t work_bytes, void *stream);

	/* ---- page-locked memory for the caller's images ------------------------------------------------------
	 * The reference's entry points take host pointers and hand the buffers back on return: every image crosses host memory
	 * once more than the link asks for (a copy into page-locked staging, a copy out of it).  A caller that keeps its images in memory
	 * from rir_host_alloc spares both: translate / gaussian_filter / bad_pixels_correct / rir_filter_chain / rir_gaussian_filter_u16 find such
	 * buffers (a registry of this library's blocks) and run their kernel on them in place.  The Python mirror returns its results in such
	 * memory, so the result of one call is the next call's input without a copy.  rir_host_alloc: NULL without a device or when the blocks
	 * handed out would exceed RIR_HOST_ALLOC_MAX_MB (256).  rir_host_free: only blocks of rir_host_alloc (anything else is left alone).
	 * rir_gaussian_filter_u16: gaussian_filter (reference signal_processing.cpp:79-148) of a uint16 image, bit-identical to converting it to
	 * float32 first as the reference's wrapper does (rir_signal_processing.py:85-113); radius <= 4, -1 otherwise. */
	void *rir_host_alloc(long long bytes);
	void rir_host_free(void *p);
	int rir_host_is_page_locked(const void *p, long long bytes);
	int rir_gaussian_filter_u16(unsigned short *src, float *dst, int w, int h, float sigma);

	/* ---- bounded-loss step on a device-resident stream -------------------------------------------------
	 * The loss injection of H264_Saver::addImageLossyNoCamera / addLoss (reference src/cpp/video_io/h264.cpp:2253-2607)
	 * as a stream operator: uint16 frames [n][h][w] in HBM in and out (distinct buffers), one state object per
	 * stream, frames taken in order.  rir_lossy_create returns a handle > 0 (0 on failure); low_errors /
	 * high_errors are HOST int[nframes] (the per-frame budgets, as h264_get_low/high_errors), may be NULL.
	 * Statistics, error budget and update all run on the device: the frames of a call are queued on `stream`
	 * back to back; with both arrays NULL the call returns without waiting, else it waits once, at the end. */
	int rir_lossy_create(int width, int height, int lossy_height, int low_value_error, int high_value_error, double std_factor,
						 int running_average, int subtract_min, int remove_bad_pixels);
	/* (the nframes input frames and the nframes output frames must not overlap: a run of frames reads input frames again - the frame
	 * that leaves the running average - after later outputs have been written; -1 otherwise) */
	int rir_lossy_step_device(int handle, const unsigned short *d_in, unsigned short *d_out, int nframes, int add_loss, int *low_errors,
							  int *high_errors, void *stream);
	/* The same step for nstreams INDEPENDENT streams (handles from rir_lossy_create with equal geometry, stepped the same number of
	 * frames so far, no bad-pixel repair) in shared launches: the state is sequential in time, so streams - not frames - are what
	 * runs side by side (SURVEY §8e "replicas").  d_in / d_out: HOST arrays of nstreams device pointers to uint16 [nframes][h][w];
	 * low_errors / high_errors: HOST int[nstreams][nframes] or NULL. */
	int rir_lossy_step_multi_device(const int *handles, int nstreams, const unsigned short *const *d_in, unsigned short *const *d_out, int nframes,
									int add_loss, int *low_errors, int *high_errors, void *stream);
	/* 0, or -1 when a run of frames this stream took part in gave up a wait between workgroups (a resident kernel that could not
	 * get all its workgroups on the chip).  Waits for `stream`.  The failure is STICKY: the stream's state has been advanced by
	 * invalid frames, so every later status and step of the stream - and of the streams that shared the failed call - returns -1
	 * until it is destroyed.  Calls that return budgets report this themselves; queue-only calls (no error arrays) leave it to
	 * this query. */
	int rir_lossy_status(int handle, void *stream);
	/* lowValueError / highValueError / stdFactor of a stream in use (H264_Saver::setParameter, h264.cpp:1709-1781): from the next frame on */
	int rir_lossy_set_errors(int handle, int low_value_error, int high_value_error, double std_factor);
	/* Which form stepped the last batch this stream led: out[0] = groups of frames offered to the constant-budget form (stdFactor == 0:
	 * the budget arithmetic of h264.cpp:2370-2376 multiplies the statistic by zero, so nothing a frame needs comes from another
	 * workgroup - an ordinary streaming launch instead of the resident one; 0 = the batch was not eligible), out[1] = of those, groups it
	 * took (it declines, on the device, when a frame's foreground or background may be empty or a NaN sits in the 40-frame window: then
	 * the general form steps the group, same results).  Waits for the stream. */
	int rir_lossy_path_stats(int handle, int *out2, void *stream);
	/* Streams whose budgets follow the statistics (stdFactor != 0 - the reference's default 5, h264.cpp:1662-1665, budget :2335-2385) go through
	 * the SPECULATIVE form: the budgets of a group are guessed (the configured errors: what a scene that does not move gets, the rounded correction
	 * being 0), the group is stepped by the streaming launch into shadow state, the frames' exact sums are taken from the frames and the reference's
	 * arithmetic is run for every frame; from the first frame whose budget is not the table's on, the table takes the computed budgets and the group
	 * is stepped again, up to RIR_LOSSY_SPEC_PASSES (3) times; a group that verifies is committed, any other is stepped by the general form - same
	 * results either way.
	 * out[0] = groups of the last batch this stream led that went through these launches (0: not eligible), out[1] = groups offered (no class that
	 * may be empty, no NaN in a window, the stream not backing off after failures), out[2] = groups committed, out[3] = passes over all groups.
	 * Waits for the stream. */
	int rir_lossy_spec_stats(int handle, int *out4, void *stream);
	void rir_lossy_destroy(int handle);

	/* ---- byte planes ------------------------------------------------------------------------------
	 * H264Capture::AddFrame (reference src/cpp/video_io/h264.cpp:1066-1082): U = v & 0xFF, V = v >> 8, Y = 0 or
	 * the 8-bit integration-time image, rows padded to `linesize`; VideoGrabber::toArray (:3016-3051) is the
	 * inverse.  Not on this build's own codec path (the block codec works on the 16-bit values); provided for
	 * callers that feed / read an external 8-bit plane codec.  Planes: [nframes][h][linesize]; d_it may be NULL. */
	int rir_split_planes_device(const unsigned short *d_img, const unsigned char *d_it, int w, int h, int nframes, int linesize,
								unsigned char *d_Y, unsigned char *d_U, unsigned char *d_V, void *stream);
	int rir_merge_planes_device(const unsigned char *d_Y, const unsigned char *d_U, const unsigned char *d_V, int linesize, int w, int h,
								int nframes, unsigned short *d_img, unsigned char *d_it, void *stream);

	/* ---- registration ---------------------------------------------------------------------------
	 * Translation-only ECC alignment, the arithmetic the reference obtains from OpenCV:
	 * cv2.findTransformECC(templ, image, warp, MOTION_TRANSLATION, (EPS|COUNT, max_iterations, eps), mask, 1)
	 * at src/python/librir/registration/masked_registration_ecc.py:166-168.  warp is a HOST float[2] = (tx, ty),
	 * start value in, result out: image(x + tx, y + ty) ~ templ(x, y).  *cc = correlation coefficient.
	 * Returns -1 where OpenCV raises (no overlap / no convergence). */
	int rir_ecc_translation_device(const float *d_templ, const float *d_image, const unsigned char *d_mask, int w, int h, float *warp,
								   int max_iterations, double eps, double *cc, int *iterations, void *stream);
	/* (im - min) / (max - min) in float32 on a w x h window of a device image with row stride src_stride (the
	 * normalisation of masked_registration_ecc.py:162-166, crop folded in); d_dst dense [h][w]. */
	int rir_minmax_normalize_device(const float *d_src, int w, int h, int src_stride, float *d_dst, void *stream);
	/* Pre-processing of `nframes` frames of a tracked sequence in shared launches, ahead of their (sequential) alignments: image
	 * by image the gaussian pre-filter, window crop, min-max normalisation (masked_registration_ecc.py:88-166) and the gradients
	 * the alignment samples.  d_imgs: uint16 ('H') or float32 ('f') [nframes][h][w]; d_norm, d_gx, d_gy: float
	 * [nframes][win_h][win_w] in device memory. */
	int rir_ecc_prepare_frames_device(const void *d_imgs, int dtype, int w, int h, int nframes, float sigma, int win_x, int win_y, int win_w, int win_h,
									  float *d_norm, float *d_gx, float *d_gy, void *stream);
	/* The alignment (cv2.findTransformECC, MOTION_TRANSLATION; masked_registration_ecc.py:166-168) of one prepared image against the
	 * reference window.  warp: HOST float[2] (tx, ty) in/out. */
	int rir_ecc_align_prepared_device(const float *d_ref_norm, const float *d_norm, const float *d_gx, const float *d_gy, int w, int h, float *warp,
									  int max_iterations, double eps, double *cc, int *iterations, void *stream);
	/* The alignments of `nframes` consecutive prepared images in one launch, image i starting from the result of image i - 1 (the
	 * loop of MaskedRegistratorECC.compute calls, masked_registration_ecc.py:102-123).  results: HOST [nframes][4] doubles = (tx, ty,
	 * correlation coefficient, iterations).  Returns how many images were aligned before the first failure (nframes: all), -1 on
	 * an invalid call.  warp: HOST float[2], start value in, last good result out. */
	int rir_ecc_align_prepared_frames_device(const float *d_ref_norm, const float *d_norm, const float *d_gx, const float *d_gy, int w, int h, int nframes,
											 float *warp, int max_iterations, double eps, double *results, void *stream);
	/* The alignments of nseq INDEPENDENT tracked sequences side by side in shared resident launches: for every sequence the
	 * operations of rir_ecc_align_prepared_frames_device in the same order (same bits), but S dependent chains at once instead of
	 * one - an alignment is a chain of iterations and cannot fill the chip on its own (masked_registration_ecc.py:105-191 is
	 * one such chain per camera).  All sequences share the window size w x h; d_ref_norm, d_norm, d_gx, d_gy: HOST arrays of nseq
	 * device pointers ([h][w] / [nframes[q]][h][w]); warps: HOST [nseq][2] in/out; results: HOST [nseq][results_stride][4]
	 * doubles (tx, ty, cc, iterations); good: HOST [nseq] = images of that sequence aligned before its first failure. */
	int rir_ecc_align_multi_device(const float *const *d_ref_norm, const float *const *d_norm, const float *const *d_gx, const float *const *d_gy, int w,
								   int h, int nseq, const int *nframes, float *warps, int max_iterations, double eps, double *results, int results_stride,
								   int *good, void *stream);
	/* The same with the pre-processing of what comes next UNDER the alignments: `next` = nnext jobs, each the arguments of one
	 * rir_ecc_prepare_frames_device call (normally the next chunk of every sequence).  The alignment launch needs the whole device to
	 * start, but once it is resident a fifth of every CU's places and most of the memory system are free: the library waits for the
	 * launch to report that and then runs the jobs on a stream of its own beside it; the caller's stream is ordered behind them when
	 * the call returns.  The jobs' outputs must not be buffers this call's alignments read. */
	typedef struct rir_ecc_prepare_job
	{
		const void *d_imgs;
		int dtype, w, h, nframes;
		float sigma;
		int win_x, win_y, win_w, win_h;
		float *d_norm, *d_gx, *d_gy;
	} rir_ecc_prepare_job;
	int rir_ecc_align_multi_overlapped_device(const float *const *d_ref_norm, const float *const *d_norm, const float *const *d_gx, const float *const *d_gy,
											  int w, int h, int nseq, const int *nframes, float *warps, int max_iterations, double eps, double *results,
											  int results_stride, int *good, const rir_ecc_prepare_job *next, int nnext, void *stream);
	/* One frame of a tracked sequence in one call - the steps of MaskedRegistratorECC.compute (masked_registration_ecc.py:88-168):
	 * gaussian pre-filter (sigma > 0), min-max normalisation of the registration window and the alignment against the
	 * already normalised reference window d_ref_norm [win_h][win_w], queued back to back with one read-back at the end.
	 * d_img: uint16 (dtype 'H') or float32 ('f') frame [h][w]. */
	int rir_ecc_register_frame_device(const void *d_img, int dtype, int w, int h, float sigma, int win_x, int win_y, int win_w, int win_h,
									  const float *d_ref_norm, float *warp, int max_iterations, double eps, double *cc, int *iterations,
									  void *stream);
	int find_transform_ecc_translation(const float *templ, const float *image, const unsigned char *mask, int w, int h, float *warp,
									   int max_iterations, double eps, double *cc);

	/* The host copy the per-frame entry points use between the caller's memory and page-locked staging (librir_amd/csrc/host_copy.cpp:
	 * a frame cut over a few helper threads; the reference's per-frame calls copy the caller's image the same way before they return,
	 * video_io.cpp:726-756 -> h264.cpp:1066-1082).  Plain host memory, no device involved: exported for tests and measurements.
	 * Returns the number of helper threads a copy may use (RIR_HOST_COPY_THREADS, default 3; 0 = memcpy on the calling thread), -1 on bad
	 * arguments. */
	int rir_host_copy(void *dst, const void *src, int64_t bytes);
	/* The same helpers move a chunk between page-locked memory and the container file (the saver's writer, the loader's read-ahead):
	 * write != 0 writes buf to [file_off, file_off + bytes) of the descriptor, else reads that range - all of it, or -1.  0 on success. */
	int rir_host_file_rw(int fd, void *buf, int64_t bytes, int64_t file_off, int write);
	/* first touch of [buf, buf + bytes) - fresh host memory that is being filled, e.g. the stack a slice of a movie is read into - one atomic
	 * compare-and-swap of a byte with itself per page on the calling thread (meant to run on a thread of its own, ahead of the reads); contents are left as they are. */
	int rir_host_touch(void *buf, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* RIR_AMD_DEVICE_H */
