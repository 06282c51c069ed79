/* librir_amd — the video_io C ABI of librir for the hot path: saver, loader, read-back filters.
 *
 * Same symbol names, argument meaning and return codes as the reference header
 * src/cpp/video_io/video_io.h (line numbers cited per function).  Frames are encoded and decoded
 * by the MI355X block codec (rir_amd_device.h); files written by the saver are "RIRB" containers
 *  that open_camera_file reports as FILE_FORMAT_H264, next to raw PCR files.
 * Conventions: openers return a handle > 0 or 0 on failure; other functions 0 / -1 / -2 (buffer
 * too small, required size written back); the reason of a failure is in get_last_log_error().
 */
#ifndef RIR_AMD_VIDEO_IO_H
#define RIR_AMD_VIDEO_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define FILE_FORMAT_PCR 1
#define FILE_FORMAT_WEST 2
#define FILE_FORMAT_PCR_ENCAPSULATED 3
#define FILE_FORMAT_ZSTD_COMPRESSED 4
#define FILE_FORMAT_H264 5
#define FILE_FORMAT_HCC 6
#define FILE_FORMAT_OTHER 7

	/* ---- loader ---- */
	int open_camera_file(const char *filename, int *file_format);			  /* video_io.h:30  */
	int video_file_format(const char *filename);							  /* video_io.h:35  */
	int open_camera_file_reader(void *file_reader, int *file_format);		  /* video_io.h:46 (always 0 here) */
	int open_camera_from_memory(void *ptr, int64_t size, int *file_format); /* video_io.h:57  */
	int close_camera(int camera);											  /* video_io.h:62  */
	int get_image_count(int camera);										  /* video_io.h:66  */
	int get_image_time(int camera, int pos, int64_t *time);				  /* video_io.h:71  */
	int get_image_size(int camera, int *width, int *height);				  /* video_io.h:76  */
	int get_filename(int camera, char *filename);							  /* video_io.h:82 (200-byte buffer) */
	int supported_calibrations(int camera, int *count);					  /* video_io.h:89  */
	int calibration_name(int camera, int calibration, char *name);			  /* video_io.h:94  */
	int load_image(int camera, int pos, int calibration, unsigned short *pixels); /* video_io.h:102 */
	int load_imageF(int camera, int pos, int calibration, float *pixels);	  /* video_io.h:103 */
	int get_last_image_raw_value(int camera, int x, int y, unsigned short *value); /* video_io.h:316 */

	/* read-back filters */
	int enable_bad_pixels(int cam, int enable);							  /* video_io.h:134 */
	int bad_pixels_enabled(int cam);										  /* video_io.h:138 */
	int load_motion_correction_file(int cam, const char *filename);		  /* video_io.h:144 */
	int enable_motion_correction(int cam, int enable);						  /* video_io.h:148 */
	int motion_correction_enabled(int cam);								  /* video_io.h:152 */

	/* attributes of the last read image / of the file */
	int get_attribute_count(int camera);									  /* video_io.h:190 */
	int get_attribute(int camera, int index, char *key, int *key_len, char *value, int *value_len);		   /* video_io.h:197 */
	int get_global_attribute_count(int camera);							  /* video_io.h:202 */
	int get_global_attribute(int camera, int index, char *key, int *key_len, char *value, int *value_len); /* video_io.h:209 */

	/* calibration / emissivity: the reference ships no calibration plugin; these answer as the
	 * reference does without one */
	int calibrate_inplace(int camera, unsigned short *img, int size, int calibration);			/* video_io.h:108 */
	/* The emissivity map is state of the loader whether or not a calibration uses it (IRVideoLoader.h:29-97, video_io.cpp:282-338): what
	 * is set is read back - set_global_emissivity fills the map with one value in [0, 1], set_emissivity takes the first `size` pixels
	 * (1 for the rest), get_emissivity returns how many values it wrote (a single 1 and 0 when nothing was ever set). */
	int set_global_emissivity(int camera, float emi);											/* video_io.h:114 */
	int set_emissivity(int camera, float *emi, int size);										/* video_io.h:120 */
	int get_emissivity(int camera, float *emi, int size);										/* video_io.h:125 */
	int support_emissivity(int camera);														/* video_io.h:129 */
	int calibrate_image(int cam, unsigned short *img, float *out, int size, int calib);		/* video_io.h:157 */
	int calibrate_image_inplace(int cam, unsigned short *img, int size, int calib);			/* video_io.h:161 */
	int camera_saturate(int cam);																/* video_io.h:165 */
	int calibration_files(int camera, char *dst, int *dstSize);								/* video_io.h:180 */
	int flip_camera_calibration(int camera, int flip_rl, int flip_ud);							/* video_io.h:182 */
	int get_table_names(int cam, char *dst, int *dst_size);									/* video_io.h:287 */
	int get_table(int cam, const char *name, float *dst, int *dst_size);						/* video_io.h:292 */

	/* ---- saver ---- */
	void set_ffmpeg_log_enabled(int);															/* video_io.h:215 */
	int h264_open_file(const char *filename, int width, int height, int lossy_height);			/* video_io.h:222 */
	void h264_close_file(int file);															/* video_io.h:226 */
	int h264_set_parameter(int file, const char *param, const char *value);					/* video_io.h:236 */
	int h264_set_global_attributes(int file, int attribute_count, char *keys, int *key_lens, char *values, int *value_lens); /* video_io.h:247 */
	int h264_add_image_lossless(int file, unsigned short *img, int64_t timestamps_ns, int attribute_count, char *keys, int *key_lens, char *values,
								int *value_lens); /* video_io.h:259 */
	int h264_add_image_lossy(int file, unsigned short *img_DL, int64_t timestamps_ns, int attribute_count, char *keys, int *key_lens, char *values,
							 int *value_lens);														/* video_io.h:272 */
	int h264_add_loss(int file, unsigned short *img);											/* video_io.h:277 */
	int h264_get_low_errors(int file, unsigned short *errors, int *size);						/* video_io.h:279 */
	int h264_get_high_errors(int file, unsigned short *errors, int *size);						/* video_io.h:280 */

	/* declared upstream (video_io.h:305-314) but defined nowhere; arguments as z_open_file_write (ZFile.cpp:325).
	 * method 1: the reference's ZFile container - 256 bytes of headers, one [int64 ts][u32 csize][zstd frame]
	 * record per image (host-side libzstd, level clevel), trailer attribute "positions" (ZFile.cpp:18-46,
	 * 410-447, 483-542); readable by open_camera_file here (FILE_FORMAT_ZSTD_COMPRESSED) and by a reference
	 * build with ZFile enabled.  Any other method: this build's block-codec container.
	 * open_video_write -> handle > 0 or -1; image_write -> 0 / -1; close_video -> bytes of image data or -1. */
	int open_video_write(const char *filename, int width, int height, int rate, int method, int clevel);
	int image_write(int writter, unsigned short *img, int64_t time);
	int64_t close_video(int writter);

	/* vendor-file helpers of the reference */
	int correct_PCR_file(const char *filename, int width, int height, int freq);				/* video_io.h:319 */
	int change_hcc_external_blackbody_temperature(const char *filename, float temperature);	/* video_io.h:321 (always -1 here) */

	/* ---- extension over the reference's one-image calls (prefix rir_) ----
	 * Images first .. first + count - 1 of a recording of this library re-recorded into a saver of the same geometry without leaving the
	 * device (what IRMovie.to_h264 / split_rush do image by image through load_image, video_io.h:147, and h264_add_image_lossless, :268):
	 * per-image attributes travel with the images when keep_attributes != 0, timestamps_ns[count] are the new time stamps.  Returns count; -2 when this way is not
	 * open (another kind of file or geometry, a read-back filter switched on) and the caller goes image by image; -1 on failure. */
	int rir_transcode_images(int camera, int saver, int first, int count, const int64_t *timestamps_ns, int keep_attributes);

#ifdef __cplusplus
}
#endif
#endif /* RIR_AMD_VIDEO_IO_H */
