"""TEST INFRASTRUCTURE — ctypes access to the CPU oracle and to oracle/_ref.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product package (``librir_amd``) never does.

``Oracle``  = oracle/librir_oracle.so, the plain-C restatement (oracle/rir_oracle.c).
``Ref``     = oracle/_ref/librir_ref.so, the UNMODIFIED reference C++ compiled by
              oracle/build_ref.sh (present only where it was built; never on a box without it).
"""
import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

DTYPE_CHARS = {
    np.dtype(np.bool_): "?",
    np.dtype(np.int8): "b",
    np.dtype(np.uint8): "B",
    np.dtype(np.int16): "h",
    np.dtype(np.uint16): "H",
    np.dtype(np.int32): "i",
    np.dtype(np.uint32): "I",
    np.dtype(np.int64): "l",
    np.dtype(np.uint64): "L",
    np.dtype(np.float32): "f",
    np.dtype(np.float64): "d",
}


def _p(a):
    return a.ctypes.data_as(ct.c_void_p)


class _SignalProcessingMixin:
    """Entry points whose signatures are identical in the oracle (orc_ prefix) and in _ref."""

    _prefix = ""

    def _fn(self, name):
        return getattr(self.lib, self._prefix + name)

    def translate(self, image, dx, dy, strategy="", background=0, prefill=None):
        img = np.ascontiguousarray(image)
        dst = np.array(img if prefill is None else prefill, copy=True, order="C")
        back = np.zeros(1, dtype=img.dtype)
        back[0] = background
        f = self._fn("translate")
        f.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float, ct.c_void_p, ct.c_char_p]
        f.restype = ct.c_int
        r = f(ord(DTYPE_CHARS[img.dtype]), _p(img), _p(dst), img.shape[1], img.shape[0],
              np.float32(dx), np.float32(dy), _p(back), strategy.encode())
        if r < 0:
            raise RuntimeError("translate failed")
        return dst

    def gaussian_filter(self, image, sigma):
        img = np.ascontiguousarray(image, dtype=np.float32)
        dst = np.zeros_like(img)
        f = self._fn("gaussian_filter")
        f.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float]
        f.restype = ct.c_int
        f(_p(img), _p(dst), img.shape[1], img.shape[0], np.float32(sigma))
        return dst

    def find_median_pixel(self, image, percent=0.5, mask=None):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        if mask is None:
            f = self._fn("find_median_pixel")
            f.argtypes = [ct.c_void_p, ct.c_int, ct.c_float]
            f.restype = ct.c_int
            return f(_p(img), img.size, np.float32(percent))
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        f = self._fn("find_median_pixel_mask")
        f.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_float]
        f.restype = ct.c_int
        return f(_p(img), _p(m), img.size, np.float32(percent))

    def label_image(self, image, background=0):
        """-> (labels int32 [h][w], areas [components + 1], xy [components + 1][2])"""
        img = np.ascontiguousarray(image)
        h, w = img.shape
        dst = np.full(img.shape, -7, dtype=np.int32)
        xy = np.zeros((img.size + 1, 2), dtype=np.float64)
        area = np.zeros(img.size + 1, dtype=np.int32)
        back = np.zeros(1, dtype=img.dtype)
        back[0] = background
        f = self._fn("label_image")
        f.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p]
        f.restype = ct.c_int
        r = f(ord(DTYPE_CHARS[img.dtype]), _p(img), _p(dst), w, h, _p(back), _p(xy), _p(area))
        if r < 0:
            raise RuntimeError("label_image failed")
        return dst, area[:r].copy(), xy[:r].copy()

    def keep_largest_area(self, image, background=0, foreground=1):
        img = np.ascontiguousarray(image)
        h, w = img.shape
        dst = np.full(img.shape, -7, dtype=np.int32)
        back = np.zeros(1, dtype=img.dtype)
        back[0] = background
        f = self._fn("keep_largest_area")
        f.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int]
        f.restype = ct.c_int
        if f(ord(DTYPE_CHARS[img.dtype]), _p(img), _p(dst), w, h, _p(back), int(foreground)) < 0:
            raise RuntimeError("keep_largest_area failed")
        return dst

    def extract_times(self, series, strategy=0, room=None):
        """series: sequences of doubles; strategy 0 union / 1 intersection -> (return code, axis)"""
        vs = [np.asarray(v, dtype=np.float64).ravel() for v in series]
        flat = np.concatenate(vs) if vs else np.zeros(0)
        flat = np.ascontiguousarray(np.append(flat, 0.0))  # (never an empty buffer)
        sizes = np.array([v.size for v in vs] + [0], dtype=np.int32)
        cap = int(sum(v.size for v in vs)) if room is None else room
        out = np.zeros(max(cap, 1), dtype=np.float64)
        n = ct.c_int(cap)
        f = self._fn("extract_times")
        f.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p, ct.POINTER(ct.c_int)]
        f.restype = ct.c_int
        rc = f(_p(flat), len(vs), _p(sizes), int(strategy), _p(out), ct.byref(n))
        return rc, (out[:n.value].copy() if rc == 0 else n.value)

    def resample_time_serie(self, x, y, times, strategy=4, padd=0.0, room=None):
        x = np.ascontiguousarray(np.append(np.asarray(x, dtype=np.float64), 0.0))
        y = np.ascontiguousarray(np.append(np.asarray(y, dtype=np.float64), 0.0))
        t = np.ascontiguousarray(np.append(np.asarray(times, dtype=np.float64), 0.0))
        cap = t.size - 1 if room is None else room
        out = np.zeros(max(cap, 1), dtype=np.float64)
        n = ct.c_int(cap)
        f = self._fn("resample_time_serie")
        f.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int, ct.c_double, ct.c_void_p, ct.POINTER(ct.c_int)]
        f.restype = ct.c_int
        rc = f(_p(x), _p(y), x.size - 1, _p(t), t.size - 1, int(strategy), float(padd), _p(out), ct.byref(n))
        return rc, (out[:n.value].copy() if rc == 0 else n.value)


class Oracle(_SignalProcessingMixin):
    _prefix = "orc_"

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "librir_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle library missing: run `make -C oracle` (or __graft_entry__.build())")
        self.lib = ct.CDLL(path)
        L = self.lib
        L.orc_codec_ntiles.restype = ct.c_int
        L.orc_codec_max_words.restype = ct.c_int64
        L.orc_codec_encode_chunk.restype = ct.c_int64
        L.orc_codec_encode_chunk.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p]
        L.orc_codec_decode_chunk.restype = ct.c_int
        L.orc_codec_decode_chunk.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p]
        L.orc_bad_pixels_detect.restype = ct.c_int
        L.orc_bad_pixels_detect.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_double, ct.c_void_p, ct.c_int]
        L.orc_bad_pixels_stats.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_double, ct.c_void_p, ct.c_void_p]
        L.orc_bad_pixels_correct.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int]
        L.orc_remove_bad_pixels.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]
        L.orc_remove_motion.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float]
        L.orc_translate_u16_f32_nearest.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float]
        L.orc_median_filter_u16.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int]
        L.orc_gaussian_kernel.argtypes = [ct.c_float, ct.c_void_p, ct.c_int]
        L.orc_gaussian_radius.argtypes = [ct.c_float]
        L.orc_gaussian_radius.restype = ct.c_int
        L.orc_clamp_min.argtypes = [ct.c_void_p, ct.c_int, ct.c_uint16]

    # ---- byte planes (C1 / C2) -------------------------------------------------------------------
    def split_planes(self, image, linesize=None, it=None):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        h, w = img.shape
        ls = w if linesize is None else int(linesize)
        Y, U, V = (np.zeros((h, ls), np.uint8) for _ in range(3))
        itp = None if it is None else _p(np.ascontiguousarray(it, dtype=np.uint8))
        self.lib.orc_split_planes.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p]
        self.lib.orc_split_planes(_p(img), w, h, _p(Y), _p(U), _p(V), ls, itp)
        return Y, U, V

    def merge_planes(self, Y, U, V, width, with_it=False):
        U = np.ascontiguousarray(U, dtype=np.uint8)
        V = np.ascontiguousarray(V, dtype=np.uint8)
        Y = np.ascontiguousarray(Y, dtype=np.uint8)
        h, ls = U.shape
        img = np.zeros((h, width), np.uint16)
        it = np.zeros((h, width), np.uint8)
        self.lib.orc_merge_planes.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p]
        self.lib.orc_merge_planes(_p(Y), _p(U), _p(V), ls, width, h, _p(img), _p(it) if with_it else None)
        return (img, it) if with_it else img

    # ---- registration ---------------------------------------------------------------------------
    def ecc_translation(self, templ, image, warp=(0.0, 0.0), mask=None, max_iter=500, eps=1e-3):
        """-> (tx, ty, cc, iterations); raises RuntimeError where OpenCV would."""
        t = np.ascontiguousarray(templ, dtype=np.float32)
        im = np.ascontiguousarray(image, dtype=np.float32)
        h, w = t.shape
        wp = np.array(warp, dtype=np.float32)
        cc, it = ct.c_double(0), ct.c_int(0)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.lib.orc_ecc_translation.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_double,
                                                 ct.POINTER(ct.c_double), ct.POINTER(ct.c_int)]
        rc = self.lib.orc_ecc_translation(_p(t), _p(im), None if m is None else _p(m), w, h, _p(wp), int(max_iter), float(eps), ct.byref(cc),
                                          ct.byref(it))
        if rc != 0:
            raise RuntimeError("ECC did not converge")
        return float(wp[0]), float(wp[1]), cc.value, it.value

    # ---- bad pixels -------------------------------------------------------------------------
    def bad_pixels_detect(self, image, std_factor=5.0):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        h, w = img.shape
        xy = np.zeros((img.size, 2), dtype=np.int32)
        n = self.lib.orc_bad_pixels_detect(_p(img), w, h, float(std_factor), _p(xy), img.size)
        return xy[:n].copy()

    def bad_pixels_stats(self, image, std_factor=5.0):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        h, w = img.shape
        a, b = ct.c_int(0), ct.c_int(0)
        self.lib.orc_bad_pixels_stats(_p(img), w, h, float(std_factor), ct.byref(a), ct.byref(b))
        return a.value, b.value

    def bad_pixels_correct(self, image, xy, floor_correct):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        h, w = img.shape
        out = np.zeros_like(img)
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        self.lib.orc_bad_pixels_correct(_p(img), _p(out), w, h, _p(xy), len(xy), int(floor_correct))
        return out

    def remove_bad_pixels(self, image, xy, rows=None):
        """IRFileLoader::removeBadPixels on the first `rows` rows (reference passes H-3)."""
        img = np.array(image, dtype=np.uint16, order="C")
        h, w = img.shape
        rows = h if rows is None else rows
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        bitmap = np.zeros((h, w), dtype=np.uint8)
        if len(xy):
            bitmap[xy[:, 1], xy[:, 0]] = 1
        self.lib.orc_remove_bad_pixels(_p(img), w, rows, _p(xy), len(xy), _p(bitmap))
        return img

    def remove_motion(self, image, tx, ty, rows=None):
        img = np.array(image, dtype=np.uint16, order="C")
        h, w = img.shape
        rows = h if rows is None else rows
        self.lib.orc_remove_motion(_p(img), w, rows, np.float32(tx), np.float32(ty))
        return img

    def translate_u16_f32_nearest(self, image, dx, dy):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        out = np.zeros(img.shape, dtype=np.float32)
        self.lib.orc_translate_u16_f32_nearest(_p(img), _p(out), img.shape[1], img.shape[0], np.float32(dx), np.float32(dy))
        return out

    def median_filter(self, image):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        out = np.zeros_like(img)
        self.lib.orc_median_filter_u16(_p(img), _p(out), img.shape[1], img.shape[0])
        return out

    def gaussian_kernel(self, sigma):
        r = self.lib.orc_gaussian_radius(np.float32(sigma))
        k = np.zeros((2 * r + 1, 2 * r + 1), dtype=np.float32)
        self.lib.orc_gaussian_kernel(np.float32(sigma), _p(k), r)
        return k

    # ---- codec ------------------------------------------------------------------------------
    def codec_encode_chunk(self, frames):
        """frames: (n, H, W) uint16 -> (hdr[ntiles, n] u64, tile_off[ntiles+1] u32, stream u64[words])."""
        fr = np.ascontiguousarray(frames, dtype=np.uint16)
        n, h, w = fr.shape
        nt = self.lib.orc_codec_ntiles(w, h)
        hdr = np.zeros((nt, n), dtype=np.uint64)
        off = np.zeros(nt + 1, dtype=np.uint32)
        stream = np.zeros(max(1, self.lib.orc_codec_max_words(w, h, n)), dtype=np.uint64)
        words = self.lib.orc_codec_encode_chunk(_p(fr), w, h, n, _p(hdr), _p(off), _p(stream))
        return hdr, off, stream[:words].copy()

    def codec_decode_chunk(self, hdr, tile_off, stream, w, h):
        hdr = np.ascontiguousarray(hdr, dtype=np.uint64)
        n = hdr.shape[1]
        out = np.zeros((n, h, w), dtype=np.uint16)
        st = np.ascontiguousarray(stream, dtype=np.uint64)
        if st.size == 0:
            st = np.zeros(1, dtype=np.uint64)
        r = self.lib.orc_codec_decode_chunk(_p(hdr), _p(np.ascontiguousarray(tile_off, dtype=np.uint32)), _p(st), w, h, n, _p(out))
        if r != 0:
            raise RuntimeError("oracle decode: malformed stream")
        return out


class Ref(_SignalProcessingMixin):
    """The reference C++ itself (oracle/_ref/librir_ref.so); raises if it was not built."""

    _prefix = ""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "_ref", "librir_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = ct.CDLL(path)
        L = self.lib
        L.ref_bad_pixels_detect.restype = ct.c_int
        L.ref_bad_pixels_detect.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int]
        L.ref_bad_pixels_new.restype = ct.c_void_p
        L.ref_bad_pixels_new.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
        L.ref_bad_pixels_correct.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p]
        L.ref_bad_pixels_delete.argtypes = [ct.c_void_p]
        L.ref_translate_u16_f32_nearest.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float]
        L.ref_median_filter_u16.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int]

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "librir_ref.so"))

    def bad_pixels_detect(self, image):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        h, w = img.shape
        xy = np.zeros((img.size, 2), dtype=np.int32)
        n = self.lib.ref_bad_pixels_detect(_p(img), w, h, _p(xy), img.size)
        return xy[:n].copy()

    def bad_pixels_correct(self, first, image):
        """BadPixels(first).correct(image) exactly as the reference object does it."""
        first = np.ascontiguousarray(first, dtype=np.uint16)
        img = np.ascontiguousarray(image, dtype=np.uint16)
        h, w = first.shape
        bp = self.lib.ref_bad_pixels_new(_p(first), w, h)
        out = np.zeros_like(img)
        self.lib.ref_bad_pixels_correct(bp, _p(img), _p(out))
        self.lib.ref_bad_pixels_delete(bp)
        return out

    def bad_pixels_floor(self, first):
        """m_median_value as observable from outside: correct() of an all-zero frame returns the
        clamp floor everywhere when it is > 0 (BadPixels.cpp:62-65)."""
        z = np.zeros_like(np.ascontiguousarray(first, dtype=np.uint16))
        return int(self.bad_pixels_correct(first, z).flat[0])

    def translate_u16_f32_nearest(self, image, dx, dy):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        out = np.zeros(img.shape, dtype=np.float32)
        self.lib.ref_translate_u16_f32_nearest(_p(img), _p(out), img.shape[1], img.shape[0], np.float32(dx), np.float32(dy))
        return out

    def median_filter(self, image):
        img = np.ascontiguousarray(image, dtype=np.uint16)
        out = np.zeros_like(img)
        self.lib.ref_median_filter_u16(_p(img), _p(out), img.shape[1], img.shape[0])
        return out


class RefAttrs:
    """The reference FileAttributes class (oracle/_ref, only when it was built with zstd)."""

    def __init__(self, ref):
        self.lib = ref.lib
        if not hasattr(self.lib, "ref_attrs_write"):
            raise FileNotFoundError("oracle/_ref built without the attribute trailer")

    def write(self, filename, global_attrs, times, frame_key=None, frame_vals=None):
        gk = list(global_attrs.keys())
        gv = [v if isinstance(v, bytes) else str(v).encode() for v in global_attrs.values()]
        n = len(times)
        K = (ct.c_char_p * max(len(gk), 1))(*[k.encode() for k in gk])
        V = (ct.c_char_p * max(len(gk), 1))(*gv)
        VL = (ct.c_int * max(len(gk), 1))(*[len(v) for v in gv])
        T = (ct.c_longlong * max(n, 1))(*[int(t) for t in times])
        fv = frame_vals or [b""] * n
        FV = (ct.c_char_p * max(n, 1))(*fv)
        FL = (ct.c_int * max(n, 1))(*[len(v) for v in fv])
        self.lib.ref_attrs_write.argtypes = [ct.c_char_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_char_p,
                                             ct.c_void_p, ct.c_void_p]
        r = self.lib.ref_attrs_write(str(filename).encode(), len(gk), K, V, VL, n, T, frame_key.encode() if frame_key else None, FV, FL)
        if r != 0:
            raise RuntimeError("ref_attrs_write failed")

    def read(self, filename, gkey, cap=4096):
        T = (ct.c_longlong * cap)()
        buf = ct.create_string_buffer(1 << 20)
        n = ct.c_int(1 << 20)
        ng = ct.c_int(0)
        self.lib.ref_attrs_read.argtypes = [ct.c_char_p, ct.c_void_p, ct.c_int, ct.c_char_p, ct.c_char_p, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]
        cnt = self.lib.ref_attrs_read(str(filename).encode(), T, cap, gkey.encode(), buf, ct.byref(n), ct.byref(ng))
        return cnt, list(T[:max(cnt, 0)]), (buf.raw[: n.value] if n.value >= 0 else None), ng.value


class RefZFile:
    """The reference's ZFile container (oracle/_ref/librir_ref_zfile.so: its unmodified ZFile.cpp, with zstd_* resolved from this build's
    libtools.so alias - oracle/build_ref.sh); raises FileNotFoundError where it was not built."""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "_ref", "librir_ref_zfile.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = ct.CDLL(path)
        self.lib.ref_zfile_write.restype = ct.c_longlong
        self.lib.ref_zfile_write.argtypes = [ct.c_char_p, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p]
        self.lib.ref_zfile_read.restype = ct.c_int
        self.lib.ref_zfile_read.argtypes = [ct.c_char_p, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.c_void_p, ct.c_void_p, ct.c_int]

    def write(self, filename, frames, times, rate=50, method=1, clevel=2):
        fr = np.ascontiguousarray(frames, dtype=np.uint16)
        n, h, w = fr.shape
        ts = np.ascontiguousarray(times, dtype=np.int64)
        size = self.lib.ref_zfile_write(str(filename).encode(), w, h, rate, method, clevel, n, _p(fr), _p(ts))
        if size < 0:
            raise RuntimeError("ref_zfile_write failed")
        return int(size)

    def read(self, filename, cap, shape):
        """(count, frames (min(count, cap), h, w), raw timestamps) as the reference's reader gives them"""
        h, w = shape
        fr = np.zeros((cap, h, w), np.uint16)
        ts = np.zeros(cap, np.int64)
        cw, ch = ct.c_int(0), ct.c_int(0)
        n = self.lib.ref_zfile_read(str(filename).encode(), ct.byref(cw), ct.byref(ch), _p(fr), _p(ts), cap)
        if n < 0:
            raise RuntimeError("ref_zfile_read failed (%d)" % n)
        if (ch.value, cw.value) != (h, w):
            raise RuntimeError("ref_zfile_read: the file holds %dx%d images" % (cw.value, ch.value))
        return n, fr[: min(n, cap)], ts[: min(n, cap)].tolist()


class OracleLossy:
    """Stateful loss injection of the lossy saver (oracle/rir_oracle.c: orc_lossy_*)."""

    def __init__(self, oracle, w, h, lossy_height=None, low_err=6, high_err=2, std_factor=5.0, running_average=32, subtract_min=False):
        self.lib = oracle.lib
        self.w, self.h = w, h
        self.lib.orc_lossy_create.restype = ct.c_void_p
        self.lib.orc_lossy_create.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_double, ct.c_int, ct.c_int]
        self.lib.orc_lossy_step.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int]
        self.lib.orc_lossy_free.argtypes = [ct.c_void_p]
        self.lib.orc_lossy_last_errors.argtypes = [ct.c_void_p, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.POINTER(ct.c_uint)]
        self.lib.orc_lossy_set_errors.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_double]
        self.std_factor = float(std_factor)
        self.p = self.lib.orc_lossy_create(w, h, h if lossy_height is None else lossy_height, low_err, high_err, float(std_factor),
                                           running_average, int(subtract_min))

    def step(self, img, add_loss=False):
        img = np.ascontiguousarray(img, dtype=np.uint16)
        out = np.zeros_like(img)
        self.lib.orc_lossy_step(self.p, _p(img), _p(out), int(add_loss))
        return out

    def set_errors(self, low_err, high_err, std_factor=None):
        """a parameter change on a stream in use (setParameter): applies from the next frame on"""
        if std_factor is not None:
            self.std_factor = float(std_factor)
        self.lib.orc_lossy_set_errors(self.p, int(low_err), int(high_err), float(self.std_factor))

    def last_errors(self):
        lo, hi, bg = ct.c_int(0), ct.c_int(0), ct.c_uint(0)
        self.lib.orc_lossy_last_errors(self.p, ct.byref(lo), ct.byref(hi), ct.byref(bg))
        return lo.value, hi.value, bg.value

    def __del__(self):
        try:
            self.lib.orc_lossy_free(self.p)
        except Exception:
            pass
