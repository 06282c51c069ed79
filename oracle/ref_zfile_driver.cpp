// TEST INFRASTRUCTURE — not part of the product.
//
// Thin extern "C" forwarders over the reference's own ZFile container (src/cpp/video_io/ZFile.cpp, compiled where it lies under
// /root/reference by oracle/build_ref.sh into oracle/_ref/librir_ref_zfile.so).  Nothing of the reference is copied here.
// The reference's ZFile.cpp calls zstd_compress / zstd_decompress / zstd_compress_bound of ITS libtools (tools.h:185-188); this library
// leaves them undefined and resolves them from THIS build's libtools.so alias - exactly the drop-in situation: reference video_io code
// running on this build's `tools`.
//
//   reference entry points reached
//     z_open_file_read / z_read_image / z_get_timestamps / z_image_count / z_image_size   ZFile.cpp:273-330,544-629, ZFile.h:15-51
//     z_open_file_write / z_write_image / z_close_file                                    ZFile.cpp:332-372,483-542
#include "ZFile.h"
#include "ReadFileChunk.h"

#include <cstring>

extern "C"
{
	// writes n images of w x h with their timestamps through the reference's writer; returns the size z_close_file reports, -1 on failure
	__attribute__((visibility("default"))) long long ref_zfile_write(const char *filename, int w, int h, int rate, int method, int clevel, int n,
																	  const unsigned short *images, const long long *times)
	{
		void *f = z_open_file_write(filename, w, h, rate, method, clevel);
		if (!f)
			return -1;
		for (int i = 0; i < n; ++i)
			if (z_write_image(f, images + (size_t)i * w * h, (int64_t)times[i]) < 0)
			{
				z_close_file(f);
				return -1;
			}
		return (long long)z_close_file(f);
	}
	// reads a file through the reference's reader: returns the image count (-1 on failure), fills w / h, up to cap images and timestamps
	// (the raw ones of z_get_timestamps, before IRFileLoader.cpp:345-372 rebases them)
	__attribute__((visibility("default"))) int ref_zfile_read(const char *filename, int *w, int *h, unsigned short *images, long long *times, int cap)
	{
		rir::FileReaderPtr reader = rir::createFileReader(rir::createFileAccess(filename));
		if (!reader)
			return -1;
		void *f = z_open_file_read(reader);
		if (!f)
			return -1;
		const int n = z_image_count(f);
		z_image_size(f, w, h);
		const int64_t *ts = z_get_timestamps(f);
		for (int i = 0; i < n && i < cap; ++i)
		{
			int64_t t = 0;
			if (z_read_image(f, i, images + (size_t)i * (*w) * (*h), &t) < 0)
			{
				z_close_file(f);
				return -2;
			}
			times[i] = ts ? (long long)ts[i] : (long long)t;
		}
		z_close_file(f);
		return n;
	}
}
