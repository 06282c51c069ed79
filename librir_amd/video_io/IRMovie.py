"""IRMovie: reader object over a video file (reference src/python/librir/video_io/IRMovie.py:72-676,
the part used by the hot path: construction from a file / bytes / numpy array, indexing, slicing,
timestamps, attributes, bad-pixel and motion correction on read-back, re-encoding)."""
import math
import os
import tempfile
from enum import Enum
from pathlib import Path

import numpy as np

from ..tools.FileAttributes import FileAttributes
from .IRSaver import IRSaver
from .rir_video_io import (close_camera, enable_bad_pixels, enable_motion_correction, flip_camera_calibration, get_attributes,
                           get_filename, get_global_attributes, get_image_count, get_image_size, get_image_time, load_image,
                           load_motion_correction_file, motion_correction_enabled, open_camera_file, open_camera_memory, support_emissivity,
                           supported_calibrations, video_file_format)


class FileFormat(Enum):
    PCR = 1
    WEST = 2
    PCR_ENCAPSULATED = 3
    ZSTD_COMPRESSED = 4
    H264 = 5
    HCC = 6
    OTHER = 7


class InvalidMovie(Exception):
    pass


def create_pcr_header(rows, columns, frequency=50, bits=16):
    """1024-byte PCR header as 256 uint32 (reference IRMovie.py:60-69)"""
    header = np.zeros((256,), dtype=np.uint32)
    header[2] = columns
    header[3] = rows
    header[5] = bits
    header[7] = frequency
    header[9] = rows * columns * 2
    header[10] = columns
    header[11] = rows
    return header


class IRMovie(object):
    _file_attributes = None

    @classmethod
    def from_filename(cls, filename):
        handle = open_camera_file(str(filename))
        instance = cls(handle)
        instance._file_attributes = FileAttributes.from_filename(filename)
        instance._file_attributes.attributes = get_global_attributes(handle)
        return instance

    @classmethod
    def from_bytes(cls, data):
        handle = open_camera_memory(data)
        instance = cls(handle)
        try:
            instance._file_attributes = FileAttributes.from_buffer(data)
            instance._file_attributes.attributes = get_global_attributes(handle)
        except RuntimeError:
            instance._file_attributes = None
        return instance

    @classmethod
    def from_numpy_array(cls, arr, attrs=None, times=None, cthreads=8):
        """Writes the array as a raw PCR file, re-encodes it (round trip through the codec) and
        opens the result, like the reference (IRMovie.py:108-144)."""
        arr = np.asarray(arr)
        if arr.ndim == 2:
            rows, columns = arr.shape
        elif arr.ndim == 3:
            _, rows, columns = arr.shape
        else:
            raise ValueError("mismatch array shape. Must be 2D or 3D")
        data = create_pcr_header(rows, columns).astype(np.uint32).tobytes() + arr.astype(np.uint16).tobytes()
        with tempfile.NamedTemporaryFile("wb", delete=False) as f:
            filename = Path(f.name)
            f.write(data)
        with cls.from_filename(filename) as _instance:
            _instance.__tempfile__ = filename
            dst = Path(filename).parent / (filename.stem + ".h264")
            _instance.to_h264(dst, times=times, cthreads=cthreads)
        instance = cls.from_filename(dst)
        instance.__tempfile__ = dst
        if attrs is not None:
            instance.attributes = attrs
            instance._file_attributes.flush()
        return instance

    def __init__(self, handle):
        if get_image_count(handle) < 0:
            raise InvalidMovie("Invalid ir_movie descriptor")
        self.handle = handle
        self.times = None
        self._bad_pixels_correction = False
        self.__tempfile__ = ""
        self._calibration_index = 0
        self._timestamps = None
        self._frame_attributes_d = {}
        self._registration_file = None

    # ---- life cycle ----
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if self._file_attributes is not None:
            try:
                self._file_attributes.close()
            except Exception:
                pass
            self._file_attributes = None
        if getattr(self, "handle", 0) > 0:
            close_camera(self.handle)
            self.handle = 0
        tmp = getattr(self, "__tempfile__", "")
        if tmp and os.path.exists(str(tmp)):
            try:
                os.unlink(str(tmp))
            except OSError:
                pass
            self.__tempfile__ = ""

    # ---- calibration (only digital levels: no calibration plugin) ----
    @property
    def calibrations(self):
        return supported_calibrations(self.handle)

    @property
    def calibration(self):
        return self.calibrations[self._calibration_index]

    @calibration.setter
    def calibration(self, value):
        self._calibration_index = self._parse_calibration_index(value)

    def _parse_calibration_index(self, value):
        names = self.calibrations
        if isinstance(value, str):
            if value in ("DL", "Digital Level"):
                return 0
            if value in names:
                return names.index(value)
            raise RuntimeError("calibration not found: %s" % value)
        value = int(value)
        if value < 0 or value >= len(names):
            raise RuntimeError("calibration index out of range")
        return value

    def flip_calibration(self, flip_rl, flip_ud):
        flip_camera_calibration(self.handle, flip_rl, flip_ud)

    @property
    def support_emissivity(self):
        return support_emissivity(self.handle)

    # ---- geometry / access ----
    @property
    def images(self):
        return get_image_count(self.handle)

    @property
    def image_size(self):
        return get_image_size(self.handle)

    @property
    def width(self):
        return self.image_size[1]

    @property
    def height(self):
        return self.image_size[0]

    @property
    def filename(self):
        return get_filename(self.handle)

    @property
    def video_file_format(self):
        return FileFormat(video_file_format(self.filename))

    @property
    def is_file_uncompressed(self):
        return self.video_file_format in (FileFormat.PCR, FileFormat.WEST, FileFormat.PCR_ENCAPSULATED)

    def load_pos(self, pos, calibration=None):
        if calibration is None:
            calibration = 0
        idx = self._parse_calibration_index(calibration)
        res = load_image(self.handle, int(pos), idx)
        self._frame_attributes_d[int(pos)] = get_attributes(self.handle)
        self._last_pos = int(pos)
        return res

    def load_secs(self, time, calibration=None):
        if self.times is None:
            self.times = np.array(list(self.timestamps), dtype=np.float64)
        index = int(np.argmin(np.abs(self.times - time)))
        return self.load_pos(index, calibration)

    def __getitem__(self, item):
        if isinstance(item, slice):
            start, stop, step = item.start or 0, item.stop or self.images, item.step or 1
            if stop < 0:
                stop = self.images + stop
            if start < 0:
                start = self.images + start
            shape = (math.ceil((stop - start) / step),) + tuple(self.image_size)
            arr = np.empty(shape, dtype=np.uint16)
            for idx, i in enumerate(range(start, stop, step)):
                arr[idx] = self.load_pos(i, self._calibration_index)
            return arr
        if isinstance(item, (int, np.integer)):
            if item < 0:
                item = self.images + item
            return self.load_pos(int(item), self._calibration_index)
        if isinstance(item, float):
            return self.load_secs(item, self._calibration_index)
        if isinstance(item, list) or (isinstance(item, np.ndarray) and item.ndim == 1):
            return np.array([self.__getitem__(e) for e in item])
        raise TypeError("unsupported index type")

    def __iter__(self):
        for i in range(self.images):
            yield self.load_pos(i, self._calibration_index)

    def __len__(self):
        return self.images

    @property
    def data(self):
        return self[:]

    @property
    def tis(self):
        return (self.data & (2**16 - 2**13)) >> 13

    # ---- time ----
    @property
    def timestamps(self):
        if self._timestamps is None:
            self._timestamps = np.array([get_image_time(self.handle, i) * 1e-9 for i in range(self.images)], dtype=np.float64)
        return self._timestamps

    @timestamps.setter
    def timestamps(self, value):
        self._timestamps = np.array(value) * 1e-9

    @property
    def frame_period(self):
        return np.diff(self.timestamps).mean().round(3)

    @property
    def duration(self):
        return (get_image_time(self.handle, self.images - 1) - get_image_time(self.handle, 0)) * 1e-9

    # ---- attributes ----
    @property
    def attributes(self):
        if self._file_attributes is None:
            return get_global_attributes(self.handle)
        return self._file_attributes.attributes

    @attributes.setter
    def attributes(self, value):
        if self._file_attributes is not None:
            self._file_attributes.attributes = value

    @property
    def frame_attributes(self):
        """attributes of the last read image"""
        return self._frame_attributes_d.get(getattr(self, "_last_pos", -1), {})

    # ---- read-back filters ----
    @property
    def bad_pixels_correction(self):
        return self._bad_pixels_correction

    @bad_pixels_correction.setter
    def bad_pixels_correction(self, value):
        self._bad_pixels_correction = bool(value)
        enable_bad_pixels(self.handle, self._bad_pixels_correction)

    @property
    def registration_file(self):
        return self._registration_file

    @registration_file.setter
    def registration_file(self, value):
        load_motion_correction_file(self.handle, str(value))
        self._registration_file = Path(value)

    @property
    def registration(self):
        return motion_correction_enabled(self.handle)

    @registration.setter
    def registration(self, value):
        enable_motion_correction(self.handle, bool(value))

    # ---- re-encoding ----
    def to_h264(self, dst_filename, start_img=0, count=-1, clevel=8, attrs=None, times=None, frame_attributes=None, cthreads=8, cfiles=None):
        if count < 0:
            count = self.images
        if start_img + count > self.images:
            count = self.images - start_img
        if count == 0:
            raise RuntimeError("No images in selected range to save")
        if attrs is None:
            attrs = dict(self.attributes)
        if frame_attributes is not None and len(frame_attributes) != count:
            raise RuntimeError("Given frame attributes are not equal to the number of saved images")
        for k in ("MIN_T", "MIN_T_HEIGHT", "STORE_IT"):
            attrs.pop(k, None)
        h, w = self.image_size
        if times is None:
            times = list(t * 1e9 for t in self.timestamps)
        with IRSaver(str(dst_filename), w, h, h, clevel) as s:
            s.set_global_attributes(attrs)
            s.set_parameter("threads", cthreads)
            s.set_parameter("codec", "h264")
            saved = 0
            for i in range(start_img, start_img + count):
                img = self.load_pos(i, 0)
                fa = self.frame_attributes if frame_attributes is None else frame_attributes[saved]
                s.add_image(img, times[i], attributes=fa)
                saved += 1

    def pcr2h264(self, outfile=None, overwrite=False, **kwargs):
        if outfile is None:
            outfile = str(Path(self.filename).with_suffix(".h264"))
        if os.path.exists(outfile) and not overwrite:
            raise RuntimeError("file exists: %s" % outfile)
        self.to_h264(outfile, **kwargs)
        return IRMovie.from_filename(outfile)

    def __repr__(self):
        return "IRMovie({})".format(self.filename)
