"""ctypes shims over the signal_processing C ABI (include/rir_amd_signal_processing.h).

Same function names, argument meaning and error behaviour as the reference wrapper
(reference src/python/librir/signal_processing/rir_signal_processing.py:23-160 and :330-415):
numpy arrays in, numpy arrays out, ``RuntimeError`` on bad dimensions / dtypes / library errors.
"""
import ctypes as ct

import numpy as np

from ..low_level.misc import _signal_processing as _sp
from ..low_level.misc import last_error, result_buffer, toCharP

# numpy dtype -> type character of the C entry point.  The reference maps int64 to 'L'
# (duplicated dict key, rir_signal_processing.py:15-16); both int64 and uint64 are accepted here
# and int64 keeps its own signed instantiation.
_DTYPES = {
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

_sp.translate.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float, ct.c_void_p, ct.c_char_p]
_sp.gaussian_filter.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float]
_sp.rir_gaussian_filter_u16.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float]
_sp.find_median_pixel.argtypes = [ct.c_void_p, ct.c_int, ct.c_float]
_sp.find_median_pixel_mask.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_float]
_sp.bad_pixels_create.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
_sp.bad_pixels_correct.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p]
_sp.bad_pixels_destroy.argtypes = [ct.c_int]
_sp.bad_pixels_destroy.restype = None


def translate(image, dx, dy, strategy=str(), background=None):
    """Translate ``image`` by the floating point offset (dx, dy).

    strategy: "" / "noborder" (border pixels keep the source value), "constant" or "background"
    (border pixels set to ``background``), "nearest", "wrap".
    """
    image = np.asarray(image)
    if image.ndim != 2:
        raise RuntimeError("translate: wrong input image dimension")
    if strategy == "background" and background is None:
        raise RuntimeError("translate: wrong background value")
    ch = _DTYPES.get(image.dtype)
    if ch is None:
        raise RuntimeError("An error occured while calling 'translate'")
    strat = toCharP(strategy)
    if strat == b"constant":
        strat = b"background"
    src = np.ascontiguousarray(image)  # (the library reads it and leaves it alone: no copy of an array that is contiguous already)
    # "noborder" leaves the values of the pixels it does not reach in place: only then does the result start as a copy of the image
    dst = result_buffer(src.shape, src.dtype)
    if strat in (b"", b"noborder"):
        np.copyto(dst, src)
    back = np.zeros(1, dtype=image.dtype)
    if background is not None:
        back[0] = background
    r = _sp.translate(ord(ch), src.ctypes.data, dst.ctypes.data, src.shape[1], src.shape[0], np.float32(dx), np.float32(dy),
                      back.ctypes.data, strat)
    if r < 0:
        raise RuntimeError("An error occured while calling 'translate': " + last_error())
    return dst


def gaussian_filter(image, sigma=1.0):
    """Gaussian filter; the result is always float32."""
    image = np.asarray(image)
    if image.ndim != 2:
        raise RuntimeError("gaussian_filter: wrong input image dimension")
    dst = result_buffer(image.shape, np.float32)  # (the library writes every pixel or fails)
    r = -1
    if image.dtype == np.uint16 and 0 < float(sigma) < 2.5:
        # every uint16 is a float32: the kernel converts as it reads - the same bits as converting first (what the reference's wrapper does,
        # rir_signal_processing.py:85-113), half the bytes up the link and no float copy of the image on the host
        src = np.ascontiguousarray(image)
        r = _sp.rir_gaussian_filter_u16(src.ctypes.data, dst.ctypes.data, src.shape[1], src.shape[0], np.float32(sigma))
    if r < 0:
        src = np.ascontiguousarray(image, dtype=np.float32)
        r = _sp.gaussian_filter(src.ctypes.data, dst.ctypes.data, src.shape[1], src.shape[0], np.float32(sigma))
    if r < 0:
        raise RuntimeError("An error occured while calling 'gaussian_filter': " + last_error())
    return dst


def find_median_pixel(image, percent=0.5, mask=None):
    """Smallest pixel value below or at which at least percent*size pixels lie."""
    image = np.asarray(image)
    if image.ndim != 2:
        raise RuntimeError("find_median_pixel: wrong input image dimension")
    img = np.ascontiguousarray(image, dtype=np.uint16)
    if mask is not None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        res = _sp.find_median_pixel_mask(img.ctypes.data, m.ctypes.data, img.size, float(percent))
    else:
        res = _sp.find_median_pixel(img.ctypes.data, img.size, float(percent))
    if res < 0:
        raise RuntimeError("An error occured while calling 'find_median_pixel': " + last_error())
    return res


def bad_pixels_create(first_image):
    first_image = np.asarray(first_image)
    if first_image.ndim != 2:
        raise RuntimeError("bad_pixels_create: wrong input image dimension")
    img = np.ascontiguousarray(first_image, dtype=np.uint16)
    h = _sp.bad_pixels_create(img.ctypes.data, img.shape[1], img.shape[0])
    if h <= 0:
        raise RuntimeError("An error occured while calling 'bad_pixels_create': " + last_error())
    return h


def bad_pixels_correct(handle, img):
    img = np.asarray(img)
    if img.ndim != 2:
        raise RuntimeError("bad_pixels_correct: wrong input image dimension")
    src = np.ascontiguousarray(img, dtype=np.uint16)
    out = result_buffer(src.shape, np.uint16)  # (the library writes every pixel or fails)
    r = _sp.bad_pixels_correct(handle, src.ctypes.data, out.ctypes.data)
    if r < 0:
        raise RuntimeError("An error occured while calling 'bad_pixels_correct': " + last_error())
    return out


def bad_pixels_destroy(handle):
    _sp.bad_pixels_destroy(handle)


# ---- extension: the three pre-recording filters in one call ----------------------------------------------------------------------
_sp.rir_filter_chain.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float, ct.c_float, ct.c_void_p, ct.c_char_p]


def filter_chain(image, bad_pixels, sigma, dx, dy, strategy="nearest", background=0):
    """Extension: ``bad_pixels.correct(image)`` -> ``gaussian_filter(sigma)`` -> ``translate(dx, dy, strategy)`` -> uint16 in ONE library call
    on one uint16 image (the three calls cross the link three times with two float images in between).  ``bad_pixels``: a ``BadPixels``
    object, a handle, or None.  Strategies "nearest" and "background" / "constant"."""
    img = np.ascontiguousarray(image)
    if img.ndim != 2 or img.dtype != np.uint16:
        raise RuntimeError("filter_chain: a 2-D uint16 image expected")
    if strategy == "constant":
        strategy = "background"
    handle = 0 if bad_pixels is None else int(getattr(bad_pixels, "handle", bad_pixels))
    out = result_buffer(img.shape, np.uint16)
    back = np.array([background], dtype=np.uint16)
    if _sp.rir_filter_chain(handle, img.ctypes.data, out.ctypes.data, img.shape[1], img.shape[0], float(sigma), float(dx), float(dy), back.ctypes.data,
                            toCharP(strategy)) < 0:
        raise RuntimeError("An error occured while calling 'filter_chain': " + (last_error() or ""))
    return out


# ---- time axes (host bookkeeping, csrc/time_series.cpp) -------------------------------------------------------------------------
_sp.extract_times.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p, ct.POINTER(ct.c_int)]
_sp.resample_time_serie.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int, ct.c_double, ct.c_void_p, ct.POINTER(ct.c_int)]
_sp.label_image.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p]
_sp.keep_largest_area.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int]


def extract_times(time_series, strategy="union"):
    """
    One time axis out of several (same call as reference rir_signal_processing.py:150-207): ``time_series`` is a list of time vectors,
    ``strategy`` says which span the result covers - 'union': from the earliest to the latest sample of any series, 'inter': only the span
    all of them share.  The result is increasing and holds every distinct sample time of the inputs inside that span once.

    Where the reference never returns (an empty series, a NaN at either end of one or two NaN in one, a series with no sample inside the
    common range of 'inter') the library refuses and this raises RuntimeError.
    """
    if len(time_series) == 0:
        raise RuntimeError("extract_times: NULL size")
    if strategy != "union" and strategy != "inter":
        raise RuntimeError("extract_times: wrong strategy")
    series = [np.array(t, dtype=np.float64).ravel() for t in time_series]
    times = np.ascontiguousarray(np.concatenate(series + [np.zeros(1)]))  # (one spare element: never an empty buffer)
    sizes = np.array([t.size for t in series], dtype=np.int32)
    outsize = ct.c_int(int(sizes.sum()))
    out = np.zeros(max(outsize.value, 1), dtype=np.float64)
    s = 1 if strategy == "inter" else 0
    tmp = _sp.extract_times(times.ctypes.data, len(series), sizes.ctypes.data, s, out.ctypes.data, ct.byref(outsize))
    if tmp == -2:
        out = np.zeros(outsize.value, dtype=np.float64)
        tmp = _sp.extract_times(times.ctypes.data, len(series), sizes.ctypes.data, s, out.ctypes.data, ct.byref(outsize))
    if tmp != 0:
        raise RuntimeError(last_error() or "extract_times: unknown error")
    return out[0:outsize.value]


def resample_time_serie(x, y, time_vector, padd=None, interp=True):
    """
    The series (``x`` sample times, ``y`` values) read off at the times of ``time_vector`` (same call as reference
    rir_signal_processing.py:210-270).  ``interp`` true: linear interpolation between the two samples around each new time, false: the value of
    the nearer of the two.  New times outside the series get ``padd`` when it is given, the first / last value otherwise.  Returns the values.

    (The reference gives the library room for 2 * len(x) values and raises "unknown error" for a longer time_vector; here the room is
    the time vector's length.)
    """
    if len(x) != len(y) or len(x) == 0:
        raise RuntimeError("resample_time_serie: wrong input serie size")
    if len(time_vector) == 0:
        raise RuntimeError("resample_time_serie: wrong time vector size")
    x = np.ascontiguousarray(np.array(x, dtype=np.float64).ravel())
    y = np.ascontiguousarray(np.array(y, dtype=np.float64).ravel())
    time_vector = np.ascontiguousarray(np.array(time_vector, dtype=np.float64).ravel())
    s = 0
    if padd is not None:
        s |= 2
    if bool(interp):
        s |= 4
    padd = 0.0 if padd is None else float(padd)
    outsize = ct.c_int(max(2 * x.size, time_vector.size))
    out = np.zeros(outsize.value, dtype=np.float64)
    tmp = _sp.resample_time_serie(x.ctypes.data, y.ctypes.data, x.size, time_vector.ctypes.data, time_vector.size, s, padd, out.ctypes.data,
                                  ct.byref(outsize))
    if tmp != 0:
        raise RuntimeError(last_error() or "resample_time_serie: unknown error")
    return out[0:outsize.value]


# ---- connected components (csrc/label_kernels.hip) ------------------------------------------------------------------------------
def _labelling_input(image, background_value, name):
    if not isinstance(image, np.ndarray) or len(image.shape) != 2:
        raise RuntimeError("%s: wrong input image dimension" % name)
    dt = _DTYPES.get(image.dtype, None)
    if dt is None:
        raise RuntimeError("An error occured while calling '%s'" % name)
    img = np.ascontiguousarray(image)
    background = np.zeros(1, dtype=image.dtype)
    background[0] = background_value
    return img, background, dt


def label_image(image: np.ndarray, background_value=0):
    """
    Connected components of ``image`` (same call as reference rir_signal_processing.py:319-370): -> (labels int32 of the image's shape, areas,
    first_points); entry ``k`` of the two tables belongs to label ``k``, entry 0 to the background (nothing useful in it).

    Pixels differing from background_value form the components; vertical neighbours are joined whatever their values, horizontal
    neighbours when their values are equal (as upstream); labels follow the raster order of the components' first pixels.  As upstream,
    both columns of first_points hold the first pixel's x.
    """
    img, background, dt = _labelling_input(image, background_value, "label_image")
    res = np.empty(img.shape, dtype=np.int32)
    areas = np.empty(img.size + 1, dtype=np.int32)  # (the reference allocates img.size entries: one short when every pixel of a row
    xy = np.empty((img.size + 1, 2), dtype=np.float64)  # image is its own component)
    r = _sp.label_image(ord(dt), img.ctypes.data, res.ctypes.data, img.shape[1], img.shape[0], background.ctypes.data, xy.ctypes.data,
                        areas.ctypes.data)
    if r < 0:
        raise RuntimeError("An error occured while calling 'label_image'")
    return (res, areas[0:r], xy[0:r])


def keep_largest_area(image, background_value=0, foreground_value=1):
    """
    int32 image with ``foreground_value`` on the component of ``image`` that has the most pixels and ``background_value`` everywhere else (same
    call as reference rir_signal_processing.py:373-415; among equals the component met first in raster order wins).
    """
    img, background, dt = _labelling_input(image, background_value, "keep_largest_area")
    res = np.empty(img.shape, dtype=np.int32)
    r = _sp.keep_largest_area(ord(dt), img.ctypes.data, res.ctypes.data, img.shape[1], img.shape[0], background.ctypes.data, int(foreground_value))
    if r < 0:
        raise RuntimeError("An error occured while calling 'keep_largest_area'")
    return res
