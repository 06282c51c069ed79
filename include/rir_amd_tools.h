/* librir_amd — the part of librir's `tools` C ABI that the hot path's callers need: logging,
 * the int-handle registry, the file attribute trailer and the zstd one-liners.
 * Same names, arguments and return codes as the reference header src/cpp/tools/tools.h
 * (cited per group).  Everything here is host-side code (no GPU).
 */
#ifndef RIR_AMD_TOOLS_H
#define RIR_AMD_TOOLS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

	/* logging — reference tools.h:28-55.  Levels: 0 info, 1 warning, 2 error. */
	typedef void (*print_function)(int, const char *);
	void set_print_function(print_function function);
	void disable_print(void);
	void reset_print_functions(void);
	/* -1 and *len = needed length when the buffer is too small */
	int get_last_log_error(char *text, int *len);

	/* handle registry — reference tools.h:67-78.  Only objects created by this library can be
	 * registered (the reference accepts any rir::BaseShared). */
	int set_void_ptr(void *obj);
	void *get_void_ptr(int index);
	void rm_void_ptr(int index);

	/* file attributes (the "H264ATTRIBUTES" trailer) — reference tools.h:96-179.
	 * Openers return a handle > 0, or 0 on error; getters 0 / -1 / -2 (buffer too small, *len set). */
	int attrs_open_from_memory(void *ptr, int64_t size);
	int attrs_open_file(const char *filename);
	void attrs_close(int handle);
	void attrs_discard(int handle);
	int attrs_flush(int handle);
	int attrs_image_count(int handle);
	int attrs_global_attribute_count(int handle);
	int attrs_global_attribute_name(int handle, int pos, char *name, int *len);
	int attrs_global_attribute_value(int handle, int pos, char *value, int *len);
	int attrs_frame_attribute_count(int handle, int frame);
	int attrs_frame_attribute_name(int handle, int frame, int pos, char *name, int *len);
	int attrs_frame_attribute_value(int handle, int frame, int pos, char *value, int *len);
	int attrs_frame_timestamp(int handle, int frame, int64_t *time);
	int attrs_timestamps(int handle, int64_t *time);
	int attrs_set_times(int handle, int64_t *times, int size);
	int attrs_set_time(int handle, int pos, int64_t time);
	int attrs_set_frame_attributes(int handle, int pos, char *keys, int *key_lens, char *values, int *value_lens, int count);
	int attrs_set_global_attributes(int handle, char *keys, int *key_lens, char *values, int *value_lens, int count);

	/* zstd one-liners — reference tools.h:185-188; served by the host's libzstd.so.1 through
	 * dlopen, -1 when it is absent or on error. */
	int64_t zstd_compress_bound(int64_t srcSize);
	int64_t zstd_decompress_bound(char *src, int64_t srcSize);
	int64_t zstd_compress(char *src, int64_t srcSize, char *dst, int64_t dstSize, int level);
	int64_t zstd_decompress(char *src, int64_t srcSize, char *dst, int64_t dstSize);

#ifdef __cplusplus
}
#endif
#endif /* RIR_AMD_TOOLS_H */
