// Metadata trailer of a video file: global attributes, per-frame attributes, timestamps.
// Byte-compatible with the reference trailer (reference src/cpp/tools/FileAttributes.cpp:60-166
// string/map encoding, :455-516 write order):
//
//   map(global) map(frame 0) ... map(frame N-1)  int64 ts[N]  u64 N  u64 trailer_size  "H264ATTRIBUTES"
//   map    = u64 count, then (string key, string value)*
//   string = u64 size (bit 63 set = zstd-compressed), bytes; a compressed string starts with the
//            u64 raw size.  Values of 1000 bytes or more are compressed when zstd is available and
//            it makes them smaller.  Little-endian.
//
// The trailer is always the last bytes of the file and may be rewritten (longer or shorter) by
// another FileAttributes object opened on the same file.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "runtime.h"

namespace rir
{
	typedef std::map<std::string, std::string> AttrMap;

	// zstd through dlopen("libzstd.so.1") - optional; absent => strings are stored uncompressed and
	// compressed strings in existing files cannot be expanded (they read as empty, like a failed
	// ZSTD_decompress in the reference).
	struct ZstdApi
	{
		bool ok = false;
		size_t (*compressBound)(size_t) = nullptr;
		size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
		size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
		unsigned long long (*getFrameContentSize)(const void *, size_t) = nullptr;
		unsigned (*isError)(size_t) = nullptr;
		static const ZstdApi &get();
	};

	class FileAttributes : public Object
	{
	public:
		const char *type_name() const override { return "FileAttributes"; }
		~FileAttributes() override;

		bool open(const char *filename);					// reads the trailer when there is one; creates the file when missing
		bool open_memory(const void *ptr, size_t size);	// read-only
		void close();										// writes the trailer when dirty
		void discard();
		void flush() { write_if_dirty(); }
		bool is_open() const { return !m_filename.empty() || m_readonly; }

		size_t size() const { return m_times.size(); }
		void resize(size_t n);
		const AttrMap &global_attributes() const { return m_global; }
		void set_global_attributes(const AttrMap &a);
		void add_global_attribute(const std::string &k, const std::string &v);
		int64_t timestamp(size_t i) const { return m_times[i]; }
		void set_timestamp(size_t i, int64_t t);
		const std::vector<int64_t> &timestamps() const { return m_times; }
		const AttrMap &attributes(size_t i) const { return m_attrs[i]; }
		void set_attributes(size_t i, const AttrMap &a);
		size_t table_size(); // bytes of the trailer as stored in the file

		// parse a trailer that ends at `end` (exclusive) inside [data, data+size); returns its size or 0
		static size_t parse(const char *data, size_t size, AttrMap &global, std::vector<AttrMap> &frames, std::vector<int64_t> &times);
		static std::string serialize(const AttrMap &global, const std::vector<AttrMap> &frames, const std::vector<int64_t> &times);

	private:
		void write_if_dirty();
		std::string m_filename;
		bool m_readonly = false;
		bool m_dirty = false;
		size_t m_file_table_size = 0; // size of the trailer currently in the file
		AttrMap m_global;
		std::vector<AttrMap> m_attrs;
		std::vector<int64_t> m_times;
	};

	// build a map from the concatenated-keys / concatenated-values convention of the C ABI
	AttrMap attr_map_from_c(int count, const char *keys, const int *key_lens, const char *values, const int *value_lens);
} // namespace rir
