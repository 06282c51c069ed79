// See file_attributes.h.  Also exports the `tools` C ABI for attributes and zstd
// (reference src/cpp/tools/tools.h:96-188, behaviour of src/cpp/tools/tools.cpp:86-377).
#include "file_attributes.h"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <fstream>

namespace rir
{
	static const char TRAILER_TAG[] = "H264ATTRIBUTES";
	static const size_t TAG_LEN = 14;
	static const size_t MIN_SIZE_FOR_COMPRESSION = 1000;
	static const uint64_t COMPRESSED_FLAG = 1ull << 63;

	// zstd.h: ZSTD_CONTENTSIZE_UNKNOWN = 0ULL - 1, ZSTD_CONTENTSIZE_ERROR = 0ULL - 2
	static const unsigned long long kZstdContentSizeUnknown = 0ULL - 1, kZstdContentSizeError = 0ULL - 2;
	static const uint64_t kMaxAttributeBytes = 64ull << 20; // one attribute value, uncompressed

	const ZstdApi &ZstdApi::get()
	{
		static ZstdApi api = []
		{
			ZstdApi a;
			void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
			if (!h)
				h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
			if (h)
			{
				a.compressBound = (size_t(*)(size_t))dlsym(h, "ZSTD_compressBound");
				a.compress = (size_t(*)(void *, size_t, const void *, size_t, int))dlsym(h, "ZSTD_compress");
				a.decompress = (size_t(*)(void *, size_t, const void *, size_t))dlsym(h, "ZSTD_decompress");
				a.getFrameContentSize = (unsigned long long (*)(const void *, size_t))dlsym(h, "ZSTD_getFrameContentSize");
				a.isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
				a.ok = a.compressBound && a.compress && a.decompress && a.getFrameContentSize && a.isError;
			}
			return a;
		}();
		return api;
	}

	// ---- encoding --------------------------------------------------------------------------------
	static void put_u64(std::string &out, uint64_t v) { out.append(reinterpret_cast<const char *>(&v), 8); }

	static void put_string(std::string &out, const std::string &s)
	{
		const ZstdApi &z = ZstdApi::get();
		if (s.size() >= MIN_SIZE_FOR_COMPRESSION && z.ok)
		{
			std::string buf(z.compressBound(s.size()), '\0');
			const size_t c = z.compress(&buf[0], buf.size(), s.data(), s.size(), 0);
			if (!z.isError(c) && c < s.size())
			{
				put_u64(out, (uint64_t)(c + 8) | COMPRESSED_FLAG);
				put_u64(out, (uint64_t)s.size());
				out.append(buf.data(), c);
				return;
			}
		}
		put_u64(out, (uint64_t)s.size());
		out.append(s);
	}
	static void put_map(std::string &out, const AttrMap &m)
	{
		put_u64(out, (uint64_t)m.size());
		for (const auto &kv : m)
		{
			put_string(out, kv.first);
			put_string(out, kv.second);
		}
	}

	struct Cursor
	{
		const char *p, *end;
		bool ok = true;
		uint64_t u64()
		{
			if (!ok || (size_t)(end - p) < 8)
			{
				ok = false;
				return 0;
			}
			uint64_t v;
			std::memcpy(&v, p, 8);
			p += 8;
			return v;
		}
		std::string str()
		{
			uint64_t s = u64();
			const bool compressed = (s & COMPRESSED_FLAG) != 0;
			s &= ~COMPRESSED_FLAG;
			if (!ok || (uint64_t)(end - p) < s)
			{
				ok = false;
				return std::string();
			}
			if (!compressed)
			{
				std::string r(p, p + s);
				p += s;
				return r;
			}
			if (s < 8)
			{
				ok = false;
				return std::string();
			}
			uint64_t raw;
			std::memcpy(&raw, p, 8);
			const char *c = p + 8;
			const size_t clen = (size_t)s - 8;
			p += s;
			const ZstdApi &z = ZstdApi::get();
			if (!z.ok)
				return std::string();
			// the declared size comes from the file: nothing is allocated from it before it has been checked against what
			// the zstd frame itself says (when it says) and against a cap no attribute value comes near
			const unsigned long long fcs = z.getFrameContentSize ? z.getFrameContentSize(c, clen) : kZstdContentSizeUnknown;
			if (raw > kMaxAttributeBytes || fcs == kZstdContentSizeError || (fcs != kZstdContentSizeUnknown && fcs != raw))
			{
				ok = false;
				return std::string();
			}
			std::string r((size_t)raw, '\0');
			const size_t got = z.decompress(&r[0], r.size(), c, clen);
			if (z.isError(got) || got != raw)
				r.clear();
			return r;
		}
		AttrMap map()
		{
			AttrMap m;
			const uint64_t n = u64();
			for (uint64_t i = 0; ok && i < n; ++i)
			{
				std::string k = str();
				std::string v = str();
				if (ok)
					m[k] = v;
			}
			return m;
		}
	};

	std::string FileAttributes::serialize(const AttrMap &global, const std::vector<AttrMap> &frames, const std::vector<int64_t> &times)
	{
		std::string out;
		put_map(out, global);
		for (const auto &m : frames)
			put_map(out, m);
		for (int64_t t : times)
			put_u64(out, (uint64_t)t);
		put_u64(out, (uint64_t)times.size());
		put_u64(out, (uint64_t)(out.size() + TAG_LEN + 8)); // total trailer size, itself and the tag included
		out.append(TRAILER_TAG, TAG_LEN);
		return out;
	}

	size_t FileAttributes::parse(const char *data, size_t size, AttrMap &global, std::vector<AttrMap> &frames, std::vector<int64_t> &times)
	{
		if (size < 16 + TAG_LEN)
			return 0;
		const char *tail = data + size - (16 + TAG_LEN);
		if (std::memcmp(tail + 16, TRAILER_TAG, TAG_LEN) != 0)
			return 0;
		uint64_t count, tsize;
		std::memcpy(&count, tail, 8);
		std::memcpy(&tsize, tail + 8, 8);
		// every frame costs at least 16 bytes of trailer (an empty map + its timestamp): a larger count is corrupt
		if (tsize > size || tsize < 16 + TAG_LEN + 8 || count > (1ull << 31) || count > tsize / 16)
			return 0;
		Cursor c{data + size - tsize, tail};
		global = c.map();
		frames.assign((size_t)count, AttrMap());
		for (uint64_t i = 0; c.ok && i < count; ++i)
			frames[(size_t)i] = c.map();
		times.assign((size_t)count, 0);
		for (uint64_t i = 0; c.ok && i < count; ++i)
			times[(size_t)i] = (int64_t)c.u64();
		if (!c.ok)
			return 0;
		return (size_t)tsize;
	}

	// ---- object --------------------------------------------------------------------------------------
	FileAttributes::~FileAttributes() { close(); }

	static bool file_size_of(const std::string &name, size_t &size)
	{
		struct stat st;
		if (stat(name.c_str(), &st) != 0)
			return false;
		size = (size_t)st.st_size;
		return true;
	}

	bool FileAttributes::open(const char *filename)
	{
		close();
		m_readonly = false;
		size_t fsize = 0;
		if (!file_size_of(filename, fsize))
		{ // just create the file (FileAttributes.cpp:376-382)
			std::ofstream f(filename, std::ios::binary);
			if (!f)
				return false;
			m_filename = filename;
			return true;
		}
		m_filename = filename;
		if (fsize >= 16 + TAG_LEN)
		{
			std::ifstream f(filename, std::ios::binary);
			if (!f)
			{
				m_filename.clear();
				return false;
			}
			char tail[16 + 14];
			f.seekg((std::streamoff)(fsize - sizeof(tail)));
			f.read(tail, sizeof(tail));
			if (f && std::memcmp(tail + 16, TRAILER_TAG, TAG_LEN) == 0)
			{
				uint64_t tsize;
				std::memcpy(&tsize, tail + 8, 8);
				if (tsize <= fsize)
				{
					std::string buf((size_t)tsize, '\0');
					f.seekg((std::streamoff)(fsize - tsize));
					f.read(&buf[0], (std::streamsize)tsize);
					if (f && parse(buf.data(), buf.size(), m_global, m_attrs, m_times) == tsize)
						m_file_table_size = (size_t)tsize;
					else
					{
						m_global.clear();
						m_attrs.clear();
						m_times.clear();
					}
				}
			}
		}
		m_dirty = false;
		return true;
	}

	bool FileAttributes::open_memory(const void *ptr, size_t size)
	{
		close();
		if (!ptr || parse(static_cast<const char *>(ptr), size, m_global, m_attrs, m_times) == 0)
			return false;
		m_readonly = true;
		m_dirty = false;
		return true;
	}

	void FileAttributes::discard()
	{
		m_filename.clear();
		m_readonly = false;
		m_dirty = false;
		m_file_table_size = 0;
		m_global.clear();
		m_attrs.clear();
		m_times.clear();
	}
	void FileAttributes::close()
	{
		write_if_dirty();
		discard();
	}
	void FileAttributes::resize(size_t n)
	{
		m_dirty = true;
		m_times.resize(n);
		m_attrs.resize(n);
	}
	void FileAttributes::set_global_attributes(const AttrMap &a)
	{
		m_dirty = true;
		m_global = a;
	}
	void FileAttributes::add_global_attribute(const std::string &k, const std::string &v)
	{
		m_dirty = true;
		m_global[k] = v;
	}
	void FileAttributes::set_timestamp(size_t i, int64_t t)
	{
		m_dirty = true;
		m_times[i] = t;
	}
	void FileAttributes::set_attributes(size_t i, const AttrMap &a)
	{
		m_dirty = true;
		m_attrs[i] = a;
	}
	size_t FileAttributes::table_size()
	{
		write_if_dirty();
		return m_file_table_size;
	}

	void FileAttributes::write_if_dirty()
	{
		if (!m_dirty || m_filename.empty() || m_readonly)
			return;
		const std::string trailer = serialize(m_global, m_attrs, m_times);
		size_t fsize = 0;
		if (!file_size_of(m_filename, fsize) || fsize < m_file_table_size)
			return;
		const size_t body = fsize - m_file_table_size;
		if (trailer.size() < m_file_table_size)
			if (truncate(m_filename.c_str(), (off_t)(body + trailer.size())) != 0)
				return;
		std::fstream f(m_filename.c_str(), std::ios::in | std::ios::out | std::ios::binary);
		if (!f)
			return;
		f.seekp((std::streamoff)body);
		f.write(trailer.data(), (std::streamsize)trailer.size());
		f.close();
		m_file_table_size = trailer.size();
		m_dirty = false;
	}

	AttrMap attr_map_from_c(int count, const char *keys, const int *key_lens, const char *values, const int *value_lens)
	{
		AttrMap m;
		for (int i = 0; i < count; ++i)
		{
			std::string k(keys, keys + key_lens[i]);
			std::string v(values, values + value_lens[i]);
			m[k] = v;
			keys += key_lens[i];
			values += value_lens[i];
		}
		return m;
	}
} // namespace rir

// ---- exported `tools` symbols: attributes ---------------------------------------------------------------

using namespace rir;

RIR_EXPORT int attrs_open_file(const char *filename)
{
	auto a = std::make_shared<FileAttributes>();
	if (!filename || !a->open(filename))
		return 0;
	return register_object(a);
}
RIR_EXPORT int attrs_open_from_memory(void *ptr, int64_t size)
{
	auto a = std::make_shared<FileAttributes>();
	if (size < 0 || !a->open_memory(ptr, (size_t)size))
		return 0;
	return register_object(a);
}
RIR_EXPORT void attrs_close(int handle)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a)
		return;
	a->close();
	remove_object(handle);
}
// the reference's attrs_discard also writes the attributes (tools.cpp:124-131 calls close())
RIR_EXPORT void attrs_discard(int handle) { attrs_close(handle); }
RIR_EXPORT int attrs_flush(int handle)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a)
		return -1;
	a->flush();
	return 0;
}
RIR_EXPORT int attrs_image_count(int handle)
{
	auto a = lookup_as<FileAttributes>(handle);
	return a ? (int)a->size() : -1;
}
RIR_EXPORT int attrs_global_attribute_count(int handle)
{
	auto a = lookup_as<FileAttributes>(handle);
	return a ? (int)a->global_attributes().size() : -1;
}

static int copy_out(const std::string &s, char *dst, int *len)
{
	if (!len)
		return -1;
	if (*len < (int)s.size() || !dst)
	{
		*len = (int)s.size();
		return -2;
	}
	*len = (int)s.size();
	std::memcpy(dst, s.data(), s.size());
	return 0;
}
static const std::pair<const std::string, std::string> *nth(const AttrMap &m, int pos)
{
	if (pos < 0 || pos >= (int)m.size())
		return nullptr;
	auto it = m.begin();
	std::advance(it, pos);
	return &*it;
}
RIR_EXPORT int attrs_global_attribute_name(int handle, int pos, char *name, int *len)
{
	auto a = lookup_as<FileAttributes>(handle);
	const auto *kv = a ? nth(a->global_attributes(), pos) : nullptr;
	return kv ? copy_out(kv->first, name, len) : -1;
}
RIR_EXPORT int attrs_global_attribute_value(int handle, int pos, char *value, int *len)
{
	auto a = lookup_as<FileAttributes>(handle);
	const auto *kv = a ? nth(a->global_attributes(), pos) : nullptr;
	return kv ? copy_out(kv->second, value, len) : -1;
}
RIR_EXPORT int attrs_frame_attribute_count(int handle, int frame)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || frame < 0 || frame >= (int)a->size())
		return -1;
	return (int)a->attributes(frame).size();
}
RIR_EXPORT int attrs_frame_attribute_name(int handle, int frame, int pos, char *name, int *len)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || frame < 0 || frame >= (int)a->size())
		return -1;
	const auto *kv = nth(a->attributes(frame), pos);
	return kv ? copy_out(kv->first, name, len) : -1;
}
RIR_EXPORT int attrs_frame_attribute_value(int handle, int frame, int pos, char *value, int *len)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || frame < 0 || frame >= (int)a->size())
		return -1;
	const auto *kv = nth(a->attributes(frame), pos);
	return kv ? copy_out(kv->second, value, len) : -1;
}
RIR_EXPORT int attrs_frame_timestamp(int handle, int frame, int64_t *time)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || !time || frame < 0 || frame >= (int)a->size())
		return -1;
	*time = a->timestamp(frame);
	return 0;
}
RIR_EXPORT int attrs_timestamps(int handle, int64_t *time)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || !time)
		return -1;
	for (size_t i = 0; i < a->size(); ++i)
		time[i] = a->timestamp(i);
	return 0;
}
RIR_EXPORT int attrs_set_times(int handle, int64_t *times, int size)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || size < 0 || (size > 0 && !times))
		return -1;
	if (size != (int)a->size())
		a->resize(size);
	for (int i = 0; i < size; ++i)
		a->set_timestamp(i, times[i]);
	return 0;
}
RIR_EXPORT int attrs_set_time(int handle, int pos, int64_t time)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || pos < 0 || pos >= (int)a->size())
		return -1;
	a->set_timestamp(pos, time);
	return 0;
}
RIR_EXPORT int attrs_set_frame_attributes(int handle, int pos, char *keys, int *key_lens, char *values, int *value_lens, int count)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || pos < 0 || pos >= (int)a->size() || count < 0)
		return -1;
	a->set_attributes(pos, attr_map_from_c(count, keys, key_lens, values, value_lens));
	return 0;
}
RIR_EXPORT int attrs_set_global_attributes(int handle, char *keys, int *key_lens, char *values, int *value_lens, int count)
{
	auto a = lookup_as<FileAttributes>(handle);
	if (!a || count < 0)
		return -1;
	a->set_global_attributes(attr_map_from_c(count, keys, key_lens, values, value_lens));
	return 0;
}

// ---- exported `tools` symbols: zstd one-liners (tools.cpp:352-377), through the host's libzstd -----------

RIR_EXPORT int64_t zstd_compress_bound(int64_t srcSize)
{
	const ZstdApi &z = ZstdApi::get();
	if (!z.ok || srcSize < 0)
		return -1;
	return (int64_t)z.compressBound((size_t)srcSize);
}
RIR_EXPORT int64_t zstd_decompress_bound(char *src, int64_t srcSize)
{
	const ZstdApi &z = ZstdApi::get();
	if (!z.ok || !src || srcSize < 0)
		return -1;
	const unsigned long long r = z.getFrameContentSize(src, (size_t)srcSize);
	if (r >= 0xFFFFFFFFFFFFFFFEull) // ZSTD_CONTENTSIZE_UNKNOWN / _ERROR
		return -1;
	return (int64_t)r;
}
RIR_EXPORT int64_t zstd_compress(char *src, int64_t srcSize, char *dst, int64_t dstSize, int level)
{
	const ZstdApi &z = ZstdApi::get();
	if (!z.ok || !src || !dst || srcSize < 0 || dstSize < 0)
		return -1;
	const size_t r = z.compress(dst, (size_t)dstSize, src, (size_t)srcSize, level);
	return z.isError(r) ? -1 : (int64_t)r;
}
RIR_EXPORT int64_t zstd_decompress(char *src, int64_t srcSize, char *dst, int64_t dstSize)
{
	const ZstdApi &z = ZstdApi::get();
	if (!z.ok || !src || !dst || srcSize < 0 || dstSize < 0)
		return -1;
	const size_t r = z.decompress(dst, (size_t)dstSize, src, (size_t)srcSize);
	return z.isError(r) ? -1 : (int64_t)r;
}
