"""ctypes shims over the video_io C ABI (include/rir_amd_video_io.h).

Function names, argument meaning and error behaviour follow the reference wrapper
(reference src/python/librir/video_io/rir_video_io.py:60-491 readers, :494-750 saver): numpy arrays
in and out, ``RuntimeError`` when the library reports a failure.
"""
import ctypes as ct
import enum
import sys
import weakref

import numpy as np

from ..low_level.misc import _video_io as _v
from ..low_level.misc import last_error, toBytes, toString
from ..tools.rir_tools import pack_attributes

_vp = ct.c_void_p
_ip = ct.POINTER(ct.c_int)
_v.open_camera_file.argtypes = [ct.c_char_p, _ip]
_v.open_camera_from_memory.argtypes = [_vp, ct.c_int64, _ip]
_v.video_file_format.argtypes = [ct.c_char_p]
_v.rir_transcode_images.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int]
_v.get_image_time.argtypes = [ct.c_int, ct.c_int, ct.POINTER(ct.c_int64)]
_v.get_image_size.argtypes = [ct.c_int, _ip, _ip]
_v.get_filename.argtypes = [ct.c_int, ct.c_char_p]
_v.supported_calibrations.argtypes = [ct.c_int, _ip]
_v.calibration_name.argtypes = [ct.c_int, ct.c_int, ct.c_char_p]
_v.load_image.argtypes = [ct.c_int, ct.c_int, ct.c_int, _vp]
_v.load_imageF.argtypes = [ct.c_int, ct.c_int, ct.c_int, _vp]
_v.get_attribute.argtypes = [ct.c_int, ct.c_int, ct.c_char_p, _ip, ct.c_char_p, _ip]
_v.get_global_attribute.argtypes = [ct.c_int, ct.c_int, ct.c_char_p, _ip, ct.c_char_p, _ip]
_v.load_motion_correction_file.argtypes = [ct.c_int, ct.c_char_p]
_v.h264_open_file.argtypes = [ct.c_char_p, ct.c_int, ct.c_int, ct.c_int]
_v.h264_close_file.argtypes = [ct.c_int]
_v.h264_close_file.restype = None
_v.h264_set_parameter.argtypes = [ct.c_int, ct.c_char_p, ct.c_char_p]
_v.h264_set_global_attributes.argtypes = [ct.c_int, ct.c_int, ct.c_char_p, _vp, ct.c_char_p, _vp]
_v.h264_add_image_lossless.argtypes = [ct.c_int, _vp, ct.c_int64, ct.c_int, ct.c_char_p, _vp, ct.c_char_p, _vp]
_v.h264_add_image_lossy.argtypes = [ct.c_int, _vp, ct.c_int64, ct.c_int, ct.c_char_p, _vp, ct.c_char_p, _vp]
_v.h264_add_loss.argtypes = [ct.c_int, _vp]
_v.h264_get_low_errors.argtypes = [ct.c_int, _vp, _ip]
_v.h264_get_high_errors.argtypes = [ct.c_int, _vp, _ip]
_v.get_last_image_raw_value.argtypes = [ct.c_int, ct.c_int, ct.c_int, _vp]
_v.correct_PCR_file.argtypes = [ct.c_char_p, ct.c_int, ct.c_int, ct.c_int]
_v.set_global_emissivity.argtypes = [ct.c_int, ct.c_float]
_v.open_video_write.argtypes = [ct.c_char_p, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int]
_v.image_write.argtypes = [ct.c_int, _vp, ct.c_int64]
_v.close_video.argtypes = [ct.c_int]
_v.close_video.restype = ct.c_int64

FILE_FORMAT_PCR = 1
FILE_FORMAT_WEST = 2
FILE_FORMAT_PCR_ENCAPSULATED = 3
FILE_FORMAT_ZSTD_COMPRESSED = 4
FILE_FORMAT_H264 = 5
FILE_FORMAT_HCC = 6
FILE_FORMAT_OTHER = 7


def _fail(what):
    raise RuntimeError("An error occured while calling '%s': %s" % (what, last_error()))


# ---- readers --------------------------------------------------------------------------------------


def open_camera_file(filename):
    """Open a video file; returns the camera handle."""
    fmt = ct.c_int(0)
    h = _v.open_camera_file(toBytes(str(filename)), ct.byref(fmt))
    if h <= 0:
        raise RuntimeError("cannot read file " + str(filename))
    return h


def open_camera_memory(buffer):
    data = bytes(buffer)
    fmt = ct.c_int(0)
    h = _v.open_camera_from_memory(ct.cast(ct.c_char_p(data), _vp), len(data), ct.byref(fmt))
    if h <= 0:
        raise RuntimeError("cannot read video from memory")
    return h


class FileFormat(enum.IntEnum):
    """``video_file_format`` codes (reference rir_video_io.py:52-57 over video_io.h:17-23; an IntEnum, so that the deprecated
    ``FILE_FORMAT_*`` integers of the reference module compare equal to the members)"""
    PCR = 1
    WEST = 2
    PCR_ENCAPSULATED = 3
    ZSTD_COMPRESSED = 4
    H264 = 5
    HCC = 6
    OTHER = 7


def video_file_format(filename):
    """
    Returns the video file format of given video file (a ``FileFormat``); RuntimeError when the file cannot be opened as a video
    (reference rir_video_io.py:111-118)
    """
    res = _v.video_file_format(toBytes(str(filename)))
    if res <= 0:
        raise RuntimeError("cannot open file " + str(filename))
    return FileFormat(res)


def close_camera(camera):
    _v.close_camera(camera)


def get_filename(camera):
    buf = ct.create_string_buffer(200)
    if _v.get_filename(camera, buf) < 0:
        _fail("get_filename")
    return toString(buf.raw)


def get_image_count(camera):
    return _v.get_image_count(camera)


def get_image_time(camera, pos):
    t = ct.c_int64(0)
    if _v.get_image_time(camera, int(pos), ct.byref(t)) < 0:
        _fail("get_image_time")
    return t.value


def get_image_size(camera):
    """(height, width)"""
    w, h = ct.c_int(0), ct.c_int(0)
    if _v.get_image_size(camera, ct.byref(w), ct.byref(h)) < 0:
        _fail("get_image_size")
    return (h.value, w.value)


def supported_calibrations(camera):
    n = ct.c_int(0)
    if _v.supported_calibrations(camera, ct.byref(n)) < 0:
        _fail("supported_calibrations")
    out = []
    for i in range(n.value):
        buf = ct.create_string_buffer(200)
        if _v.calibration_name(camera, i, buf) < 0:
            _fail("calibration_name")
        out.append(toString(buf.raw))
    return out


_free_blocks = {}  # (h, w) -> a few memory blocks whose image arrays have died


def _image_buffer(h, w):
    """A fresh (h, w) uint16 array for the next image, as the reference returns (rir_video_io.py load_image: a new array per
    call) - over RECYCLED memory where that is safe.  A reader that iterates over a movie drops each image before it asks for
    the next one; a fresh 640x512 allocation then costs its 160 page faults (30-40 us, as much as the read itself).  The memory
    of an image is a block of its own; the array handed out is built directly on that block (views of it therefore keep IT alive:
    numpy stops collapsing ``base`` chains at the first non-array), and a weak-reference finaliser returns the block to the pool
    when the array object is collected - i.e. when neither the caller nor any view of the image refers to it any more.  No
    reference counts are inspected: an image the caller keeps is never written to again."""
    pool = _free_blocks.setdefault((h, w), [])
    try:
        block = pool.pop()
    except IndexError:
        block = bytearray(h * w * 2)
    a = np.ndarray((h, w), dtype=np.uint16, buffer=block)
    f = weakref.finalize(a, _recycle_block, pool, block)
    f.atexit = False
    return a


def _recycle_block(pool, block):
    if len(pool) < 4:
        pool.append(block)


def load_image(camera, pos, calibration=0, shape=None, out=None):
    """``shape``: (height, width) when the caller already knows it (IRMovie does: one library call less per image).  The
    buffer is not zero-filled first (the library writes every pixel or fails).  ``out``: a C-contiguous uint16 array of that shape
    to read into (a row of a stack: reading a slice of a movie then costs no copy of each image) - returned instead of a new array."""
    h, w = get_image_size(camera) if shape is None else shape
    if out is None:
        img = _image_buffer(h, w)
    else:
        img = out
        if img.dtype != np.uint16 or img.shape != (h, w) or not img.flags.c_contiguous or not img.flags.writeable:
            raise RuntimeError("load_image: 'out' must be a writeable C-contiguous uint16 array of the image's shape")
    if _v.load_image(camera, int(pos), int(calibration), img.ctypes.data) < 0:
        _fail("load_image")
    return img


def get_last_image_raw_value(camera, x, y):
    v = np.zeros(1, dtype=np.uint16)
    if _v.get_last_image_raw_value(camera, int(x), int(y), v.ctypes.data) < 0:
        _fail("get_last_image_raw_value")
    return int(v[0])


def _kv_list(count_fn, get_fn, camera):
    out = {}
    n = count_fn(camera)
    if n < 0:
        _fail(get_fn.__name__)
    for i in range(n):
        kl, vl = ct.c_int(200), ct.c_int(200)
        key, val = ct.create_string_buffer(kl.value), ct.create_string_buffer(vl.value)
        r = get_fn(camera, i, key, ct.byref(kl), val, ct.byref(vl))
        if r == -2:  # buffer too small: the required sizes were written back
            key, val = ct.create_string_buffer(kl.value + 1), ct.create_string_buffer(vl.value + 1)
            kl, vl = ct.c_int(kl.value + 1), ct.c_int(vl.value + 1)
            r = get_fn(camera, i, key, ct.byref(kl), val, ct.byref(vl))
        if r < 0:
            _fail(get_fn.__name__)
        out[key.raw[: kl.value].decode("utf-8", errors="replace")] = val.raw[: vl.value]
    return out


def get_attributes(camera):
    """attributes of the last read image"""
    return _kv_list(_v.get_attribute_count, _v.get_attribute, camera)


def get_global_attributes(camera):
    return _kv_list(_v.get_global_attribute_count, _v.get_global_attribute, camera)


def enable_bad_pixels(camera, enable=True):
    if _v.enable_bad_pixels(camera, int(bool(enable))) < 0:
        _fail("enable_bad_pixels")


def bad_pixels_enabled(camera):
    return bool(_v.bad_pixels_enabled(camera))


def load_motion_correction_file(cam, filename):
    if _v.load_motion_correction_file(cam, toBytes(str(filename))) < 0:
        _fail("load_motion_correction_file")


def enable_motion_correction(cam, enable):
    if _v.enable_motion_correction(cam, int(bool(enable))) < 0:
        _fail("enable_motion_correction")


def motion_correction_enabled(cam):
    return bool(_v.motion_correction_enabled(cam))


def support_emissivity(camera):
    return _v.support_emissivity(camera) > 0


# Emissivity and calibration (reference rir_video_io.py:248-361).  This build ships no calibration plugin (digital levels only, SURVEY §2
# row 6): the library answers what the reference answers for a camera without one - support_emissivity false, emissivity 1, calibration
# 0 the identity, any other calibration an error - and these wrappers turn its codes into the reference's exceptions.
def set_global_emissivity(camera, emi_value):
    if _v.set_global_emissivity(int(camera), float(emi_value)) < 0:
        raise RuntimeError("An error occured while calling 'set_global_emissivity' with emissivity " + str(emi_value))


def get_global_emissivity(camera):
    one = np.zeros(1, dtype=np.float32)
    if _v.get_emissivity(int(camera), one.ctypes.data_as(ct.POINTER(ct.c_float)), 1) < 0:
        raise RuntimeError("An error occured while calling 'get_emissivity'")
    return one[0]


def set_emissivity(camera, emissivity_array):
    e = np.ascontiguousarray(emissivity_array, dtype=np.float32)
    if _v.set_emissivity(int(camera), e.ctypes.data_as(ct.POINTER(ct.c_float)), int(e.size)) < 0:
        raise RuntimeError("An error occured while calling 'set_emissivity'")


def get_emissivity(camera):
    e = np.zeros(get_image_size(camera), dtype=np.float32)
    if _v.get_emissivity(int(camera), e.ctypes.data_as(ct.POINTER(ct.c_float)), int(e.size)) < 0:
        raise RuntimeError("An error occured while calling 'get_emissivity'")
    return e


def camera_saturate(movie_handle):
    """True when the last load_image saturated the temperature calibration (never, without a calibration)."""
    return _v.camera_saturate(int(movie_handle)) == 1


def calibrate_image(camera, image, calib):
    """The calibration ``calib`` applied to a digital-level image; returns a new uint16 array."""
    img = np.array(image, dtype=np.uint16, order="C")  # (always a copy: the library works in place)
    if _v.calibrate_inplace(int(camera), img.ctypes.data_as(ct.POINTER(ct.c_uint16)), int(img.size), int(calib)) < 0:
        raise RuntimeError("calibrate_image: unable to apply selected calibration")
    return img


def change_hcc_external_blackbody_temperature(filename, temperature):
    """HCC vendor files are outside this build: the library refuses, this raises like the reference does on failure."""
    r = _v.change_hcc_external_blackbody_temperature(toBytes(str(filename)), ct.c_float(float(temperature)))
    if r < 0:
        raise RuntimeError("An error occured while calling 'change_hcc_external_blackbody_temperature'")
    return r


def calibration_files(movie_handle):
    return []


def flip_camera_calibration(camera, flip_rl, flip_ud):
    r = _v.flip_camera_calibration(camera, int(flip_rl), int(flip_ud))
    if r == -2:
        raise RuntimeError("flip_camera_calibration: no calibration for this camera")
    if r < 0:
        _fail("flip_camera_calibration")


def correct_PCR_file(filename, width, height, frequency):
    if _v.correct_PCR_file(toBytes(str(filename)), int(width), int(height), int(frequency)) < 0:
        _fail("correct_PCR_file")


# ---- saver ---------------------------------------------------------------------------------------------


def h264_open_file(filename, width, height, lossy_height=None):
    if lossy_height is None:
        lossy_height = height
    h = _v.h264_open_file(toBytes(str(filename)), int(width), int(height), int(lossy_height))
    if h <= 0:
        _fail("h264_open_file")
    return h


def h264_close_file(saver):
    _v.h264_close_file(saver)


def h264_set_parameter(saver, param, value):
    if _v.h264_set_parameter(saver, toBytes(param), toBytes(str(value))) < 0:
        _fail("h264_set_parameter")


def h264_set_global_attributes(saver, attributes):
    k, kl, v, vl, n = pack_attributes(attributes)
    if _v.h264_set_global_attributes(saver, n, k, kl.ctypes.data, v, vl.ctypes.data) < 0:
        _fail("h264_set_global_attributes")


def _add(fn, name, saver, image, timestamp, attributes):
    # (no copy when the caller's array is already C-contiguous uint16: the library copies the frame out before it returns)
    img = np.ascontiguousarray(image, dtype=np.uint16)
    if attributes:
        k, kl, v, vl, n = pack_attributes(attributes)
        r = fn(saver, img.ctypes.data, int(timestamp), n, k, kl.ctypes.data, v, vl.ctypes.data)
    else:
        r = fn(saver, img.ctypes.data, int(timestamp), 0, None, None, None, None)
    if r < 0:
        _fail(name)


def h264_add_image_lossless(saver, image, timestamp, attributes=None):
    _add(_v.h264_add_image_lossless, "h264_add_image_lossless", saver, image, timestamp, attributes)


def transcode_images(camera, saver, first, count, timestamps_ns, keep_attributes=True):
    """Extension (``rir_transcode_images``): images ``first .. first + count - 1`` of a recording of this library re-recorded into ``saver``
    (same geometry) without leaving the device, with their per-image attributes unless ``keep_attributes`` is false.  True when done; False when this way is not open (another
    kind of file or geometry, a read-back filter switched on) - the caller then goes image by image."""
    stamps = np.ascontiguousarray(timestamps_ns, dtype=np.int64)
    if stamps.shape != (int(count),):
        raise RuntimeError("transcode_images: one time stamp per image expected")
    r = _v.rir_transcode_images(camera, saver, int(first), int(count), stamps.ctypes.data, int(bool(keep_attributes)))
    if r == -2:
        return False
    if r != count:
        _fail("transcode_images")
    return True


def h264_add_image_lossy(saver, image_DL, timestamp, attributes=None):
    _add(_v.h264_add_image_lossy, "h264_add_image_lossy", saver, image_DL, timestamp, attributes)


def h264_add_loss(saver, image):
    img = np.array(image, dtype=np.uint16, order="C")
    if _v.h264_add_loss(saver, img.ctypes.data) < 0:
        _fail("h264_add_loss")
    return img


def _errors(fn, name, saver):
    n = ct.c_int(0)
    r = fn(saver, None, ct.byref(n))
    out = np.zeros(max(n.value, 1), dtype=np.uint16)
    n2 = ct.c_int(out.size)
    r = fn(saver, out.ctypes.data, ct.byref(n2))
    if r < 0:
        _fail(name)
    return out[: n2.value]


def h264_get_low_errors(saver):
    return _errors(_v.h264_get_low_errors, "h264_get_low_errors", saver)


def h264_get_high_errors(saver):
    return _errors(_v.h264_get_high_errors, "h264_get_high_errors", saver)


# ---- plain writer (reference video_io.h:305-314; method 1 = the reference's ZFile container, zstd per image) -------
METHOD_ZSTD = 1
METHOD_BLOCK_CODEC = 0


def open_video_write(filename, width, height, rate=50, method=METHOD_BLOCK_CODEC, clevel=0):
    h = _v.open_video_write(toBytes(str(filename)), int(width), int(height), int(rate), int(method), int(clevel))
    if h <= 0:
        _fail("open_video_write")
    return h


def image_write(writer, image, timestamp):
    img = np.ascontiguousarray(image, dtype=np.uint16)
    if _v.image_write(writer, img.ctypes.data, int(timestamp)) < 0:
        _fail("image_write")


def close_video(writer):
    """Finishes the file; returns the size in bytes of the image data written."""
    size = _v.close_video(writer)
    if size < 0:
        _fail("close_video")
    return size
