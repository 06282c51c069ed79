"""ctypes shims over the `tools` C ABI (include/rir_amd_tools.h): attributes trailer and zstd.
Function names and error behaviour follow reference src/python/librir/tools/rir_tools.py."""
import ctypes as ct

import numpy as np

from ..low_level.misc import _tools, toBytes

_vp = ct.c_void_p
_ip = ct.POINTER(ct.c_int)
_tools.attrs_open_file.argtypes = [ct.c_char_p]
_tools.attrs_open_from_memory.argtypes = [_vp, ct.c_int64]
_tools.attrs_close.argtypes = [ct.c_int]
_tools.attrs_close.restype = None
_tools.attrs_discard.argtypes = [ct.c_int]
_tools.attrs_discard.restype = None
_tools.attrs_flush.argtypes = [ct.c_int]
_tools.attrs_image_count.argtypes = [ct.c_int]
_tools.attrs_global_attribute_count.argtypes = [ct.c_int]
_tools.attrs_global_attribute_name.argtypes = [ct.c_int, ct.c_int, ct.c_char_p, _ip]
_tools.attrs_global_attribute_value.argtypes = [ct.c_int, ct.c_int, ct.c_char_p, _ip]
_tools.attrs_frame_attribute_count.argtypes = [ct.c_int, ct.c_int]
_tools.attrs_frame_attribute_name.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_char_p, _ip]
_tools.attrs_frame_attribute_value.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_char_p, _ip]
_tools.attrs_timestamps.argtypes = [ct.c_int, _vp]
_tools.attrs_frame_timestamp.argtypes = [ct.c_int, ct.c_int, ct.POINTER(ct.c_int64)]
_tools.attrs_set_time.argtypes = [ct.c_int, ct.c_int, ct.c_int64]
_tools.attrs_set_times.argtypes = [ct.c_int, _vp, ct.c_int]
_tools.attrs_set_frame_attributes.argtypes = [ct.c_int, ct.c_int, ct.c_char_p, _vp, ct.c_char_p, _vp, ct.c_int]
_tools.attrs_set_global_attributes.argtypes = [ct.c_int, ct.c_char_p, _vp, ct.c_char_p, _vp, ct.c_int]
for _f in ("zstd_compress_bound", "zstd_decompress_bound", "zstd_compress", "zstd_decompress"):
    getattr(_tools, _f).restype = ct.c_int64
_tools.zstd_compress_bound.argtypes = [ct.c_int64]
_tools.zstd_decompress_bound.argtypes = [ct.c_char_p, ct.c_int64]
_tools.zstd_compress.argtypes = [ct.c_char_p, ct.c_int64, ct.c_char_p, ct.c_int64, ct.c_int]
_tools.zstd_decompress.argtypes = [ct.c_char_p, ct.c_int64, ct.c_char_p, ct.c_int64]


def pack_attributes(attributes):
    """dict -> (concatenated keys, int32 key lengths, concatenated values, int32 value lengths, count),
    the convention of every attribute setter of the C ABI.  Only pairs convertible to bytes are kept."""
    keys, values, klens, vlens = b"", b"", [], []
    for k, v in (attributes or {}).items():
        try:
            kb, vb = toBytes(k), (v if isinstance(v, (bytes, bytearray)) else toBytes(v))
        except Exception:
            continue
        keys += kb
        values += bytes(vb)
        klens.append(len(kb))
        vlens.append(len(vb))
    return keys, np.array(klens, dtype=np.int32), values, np.array(vlens, dtype=np.int32), len(klens)


def _read_sized(fn, *args):
    """call fn(*args, buffer, &len) with the -2 / retry protocol"""
    n = ct.c_int(256)
    buf = ct.create_string_buffer(n.value)
    r = fn(*args, buf, ct.byref(n))
    if r == -2:
        buf = ct.create_string_buffer(n.value + 1)
        r = fn(*args, buf, ct.byref(n))
    if r < 0:
        raise RuntimeError("An error occured while reading an attribute")
    return buf.raw[: n.value]


def attrs_open_file(filename):
    h = _tools.attrs_open_file(toBytes(str(filename)))
    if h <= 0:
        raise RuntimeError("cannot read file attributes of %s" % filename)
    return h


def attrs_open_buffer(buf):
    buf = bytes(buf)
    h = _tools.attrs_open_from_memory(ct.cast(ct.c_char_p(buf), ct.c_void_p), len(buf))
    if h <= 0:
        raise RuntimeError("cannot read file attributes from buffer")
    return h


def attrs_close(handle):
    _tools.attrs_close(handle)


def attrs_discard(handle):
    _tools.attrs_discard(handle)


def attrs_flush(handle):
    if _tools.attrs_flush(handle) < 0:
        raise RuntimeError("An error occured while calling 'attrs_flush'")


def attrs_image_count(handle):
    return _tools.attrs_image_count(handle)


def attrs_global_attributes(h):
    out = {}
    for i in range(max(_tools.attrs_global_attribute_count(h), 0)):
        k = _read_sized(_tools.attrs_global_attribute_name, h, i)
        out[k.decode("utf-8", errors="replace")] = _read_sized(_tools.attrs_global_attribute_value, h, i)
    return out


def attrs_frame_attributes(h, frame):
    out = {}
    for i in range(max(_tools.attrs_frame_attribute_count(h, frame), 0)):
        k = _read_sized(_tools.attrs_frame_attribute_name, h, frame, i)
        out[k.decode("utf-8", errors="replace")] = _read_sized(_tools.attrs_frame_attribute_value, h, frame, i)
    return out


# ---- one item at a time (the reference exposes these too, src/python/librir/tools/rir_tools.py:134-316) -------------------------
def _count(fn, name, *args):
    n = fn(*args)
    if n < 0:
        raise RuntimeError("An error occured while calling '%s'" % name)
    return n


def attrs_global_attribute_count(handle):
    return _count(_tools.attrs_global_attribute_count, "attrs_global_attribute_count", int(handle))


def attrs_frame_attribute_count(handle, pos):
    return _count(_tools.attrs_frame_attribute_count, "attrs_frame_attribute_count", int(handle), int(pos))


def attrs_global_attribute_name(handle, index):
    return _read_sized(_tools.attrs_global_attribute_name, int(handle), int(index)).decode("utf-8", errors="replace")


def attrs_global_attribute_value(handle, index):
    return _read_sized(_tools.attrs_global_attribute_value, int(handle), int(index))


def attrs_frame_attribute_name(handle, frame, index):
    return _read_sized(_tools.attrs_frame_attribute_name, int(handle), int(frame), int(index)).decode("utf-8", errors="replace")


def attrs_frame_attribute_value(handle, frame, index):
    return _read_sized(_tools.attrs_frame_attribute_value, int(handle), int(frame), int(index))


def attrs_frame_timestamp(handle, frame):
    t = ct.c_int64(0)
    if _tools.attrs_frame_timestamp(int(handle), int(frame), ct.byref(t)) < 0:
        raise RuntimeError("An error occured while calling 'attrs_frame_timestamp'")
    return t.value


def attrs_set_time(handle, frame, time):
    if _tools.attrs_set_time(int(handle), int(frame), int(time)) < 0:
        raise RuntimeError("An error occured while calling 'attrs_set_time'")


def attrs_timestamps(handle):
    n = max(_tools.attrs_image_count(handle), 0)
    t = np.zeros(n, dtype=np.int64)
    if n and _tools.attrs_timestamps(handle, t.ctypes.data) < 0:
        raise RuntimeError("An error occured while calling 'attrs_timestamps'")
    return t


def attrs_set_times(handle, times):
    t = np.ascontiguousarray(times, dtype=np.int64)
    if _tools.attrs_set_times(handle, t.ctypes.data, len(t)) < 0:
        raise RuntimeError("An error occured while calling 'attrs_set_times'")


def attrs_set_frame_attributes(handle, frame, attributes):
    k, kl, v, vl, n = pack_attributes(attributes)
    if _tools.attrs_set_frame_attributes(handle, frame, k, kl.ctypes.data, v, vl.ctypes.data, n) < 0:
        raise RuntimeError("An error occured while calling 'attrs_set_frame_attributes'")


def attrs_set_global_attributes(handle, attributes):
    k, kl, v, vl, n = pack_attributes(attributes)
    if _tools.attrs_set_global_attributes(handle, k, kl.ctypes.data, v, vl.ctypes.data, n) < 0:
        raise RuntimeError("An error occured while calling 'attrs_set_global_attributes'")


def zstd_compress_bound(size):
    return int(_tools.zstd_compress_bound(int(size)))


def zstd_decompress_bound(src):
    return int(_tools.zstd_decompress_bound(bytes(src), len(src)))


def _as_bytes(src):
    """str arguments are accepted like upstream (encoded, unencodable characters replaced: rir_tools.py:34-36,60-62)"""
    return bytes(src.encode(errors="replace") if isinstance(src, str) else src)


def zstd_compress(src, level=0):
    src = _as_bytes(src)
    cap = zstd_compress_bound(len(src))
    if cap < 0:
        raise RuntimeError("'zstd_compress': libzstd is not available on this host")
    dst = ct.create_string_buffer(cap)
    r = _tools.zstd_compress(src, len(src), dst, cap, int(level))
    if r < 0:
        raise RuntimeError("An error occured while calling 'zstd_compress'")
    return dst.raw[:r]


def zstd_decompress(src):
    src = _as_bytes(src)
    cap = zstd_decompress_bound(src)
    if cap < 0:
        raise RuntimeError("An error occured while calling 'zstd_decompress'")
    dst = ct.create_string_buffer(max(cap, 1))
    r = _tools.zstd_decompress(src, len(src), dst, cap)
    if r < 0:
        raise RuntimeError("An error occured while calling 'zstd_decompress'")
    return dst.raw[:r]
