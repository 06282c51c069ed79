"""ctypes shims over the signal_processing C ABI (include/rir_amd_signal_processing.h).

Same function names, argument meaning and error behaviour as the reference wrapper
(reference src/python/librir/signal_processing/rir_signal_processing.py:23-160 and :330-415):
numpy arrays in, numpy arrays out, ``RuntimeError`` on bad dimensions / dtypes / library errors.
"""
import ctypes as ct

import numpy as np

from ..low_level.misc import _signal_processing as _sp
from ..low_level.misc import last_error, toCharP

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
    dst = np.array(image, order="C", copy=True) if strat in (b"", b"noborder") else np.empty(src.shape, dtype=src.dtype)
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
    src = np.ascontiguousarray(image, dtype=np.float32)
    dst = np.empty(image.shape, dtype=np.float32)  # (the library writes every pixel or fails)
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
    out = np.empty(src.shape, dtype=np.uint16)  # (the library writes every pixel or fails)
    r = _sp.bad_pixels_correct(handle, src.ctypes.data, out.ctypes.data)
    if r < 0:
        raise RuntimeError("An error occured while calling 'bad_pixels_correct': " + last_error())
    return out


def bad_pixels_destroy(handle):
    _sp.bad_pixels_destroy(handle)


# ---- outside the accelerated path ---------------------------------------------------------------------
# The reference's signal_processing library also holds four CPU utilities that never touch the frame pipeline: two on time vectors
# (extract_times, resample_time_serie) and a connected-component labelling with its "largest component" filter (label_image,
# keep_largest_area).  The library exports their names and refuses; the wrappers below exist so that code importing them still imports,
# and they fail the way the wrapper fails when the library answers with an error: RuntimeError, with the library's message.
def _outside(name):
    raise RuntimeError("An error occured while calling '%s': not provided by the MI355X hot-path library "
                       "(a CPU utility of the reference outside the accelerated path; keep the reference's own library for it)" % name)


def extract_times(time_series, strategy="union"):
    _outside("extract_times")


def resample_time_serie(x, y, time_vector, padd=None, interp=True):
    _outside("resample_time_serie")


def label_image(image, background_value=0):
    _outside("label_image")


def keep_largest_area(image, background_value=0, foreground_value=1):
    _outside("keep_largest_area")
