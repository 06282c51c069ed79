"""Reading side of the Python interface: ``IRMovie`` gives array-like access to the images of a recording.

Public interface (class methods ``from_filename`` / ``from_bytes`` / ``from_numpy_array``, indexing and slicing,
``data``, ``timestamps``, ``attributes``, ``frame_attributes``, ``bad_pixels_correction``, ``registration*``,
``to_h264`` ...) as the reference class ``librir.video_io.IRMovie`` (reference
src/python/librir/video_io/IRMovie.py:72-676) for the part the hot path needs; the implementation is this build's:
frames are decoded on the MI355X a chunk at a time, the read-back filters run there as well, and one image crosses
PCIe per ``load_pos``.  Only digital levels are offered (no calibration plugin is shipped).
"""
import math
import os
import tempfile
from pathlib import Path

import numpy as np

from ..low_level.misc import touch_ahead as _touch_ahead
from ..tools.FileAttributes import FileAttributes
from . import rir_video_io as _abi
from .rir_video_io import FileFormat
from .IRSaver import IRSaver

_TI_MASK, _TI_SHIFT = 0xE000, 13  # the three top bits of a digital level carry the integration-time index


class InvalidMovie(Exception):
    pass


class CalibrationNotFound(Exception):
    """a calibration asked for by name or number that the movie does not offer (reference IRMovie.py:52-53, :172-197)"""


def create_pcr_header(rows, columns, frequency=50, bits=16):
    """The 1024-byte header of a raw PCR file as 256 little-endian uint32 (layout: reference IRFileLoader.h:43-61;
    words 2/3 = X/Y, 5 = Bits, 7 = Frequency, 9 = bytes per image, 10/11 = grab size)."""
    words = np.zeros(256, dtype=np.uint32)
    for index, value in ((2, columns), (3, rows), (5, bits), (7, frequency), (9, rows * columns * 2), (10, columns), (11, rows)):
        words[index] = value
    return words


def _remove_quietly(path):
    try:
        if path and os.path.exists(str(path)):
            os.unlink(str(path))
    except OSError:
        pass


class IRMovie(object):
    _file_attributes = None  # second, independent object on the same file: global attributes can be rewritten through it

    # ---- construction ------------------------------------------------------------------------------------------------
    def __init__(self, handle):
        if _abi.get_image_count(handle) < 0:
            raise InvalidMovie("Invalid ir_movie descriptor")
        self.handle = handle
        self.times = None  # seconds, filled on the first load_secs
        self._calibration_index = 0
        self._seconds = None
        self._bp_on = False
        self._reg_file = None
        self._per_frame = {}  # position -> attributes of that image, as read
        self._current = -1
        self._shape = None  # (height, width), fetched once
        self._owned_file = None  # temporary file this object must delete on close

    @classmethod
    def _attach_attributes(cls, movie, opener, source, optional):
        try:
            fa = opener(source)
            fa.attributes = _abi.get_global_attributes(movie.handle)
            movie._file_attributes = fa
        except RuntimeError:
            if not optional:
                movie.close()
                raise
            movie._file_attributes = None  # a movie in memory may carry no trailer: read-only attributes then
        return movie

    @classmethod
    def from_filename(cls, filename):
        return cls._attach_attributes(cls(_abi.open_camera_file(str(filename))), FileAttributes.from_filename, filename, False)

    @classmethod
    def from_bytes(cls, data):
        return cls._attach_attributes(cls(_abi.open_camera_memory(data)), FileAttributes.from_buffer, data, True)

    @classmethod
    def from_numpy_array(cls, arr, attrs=None, times=None, cthreads=8):
        """The array goes through the codec and the encoded (temporary) file is what the returned movie reads, like the reference
        (IRMovie.py:108-144).  The reference gets there through a raw PCR file that it writes and re-encodes with ``to_h264``; here the
        images are recorded straight from the array - the same file (50 images a second when ``times`` is not given, as a PCR header
        says; no attributes; ``to_h264``'s saver parameters) without writing, reading and copying the movie twice more (1 000 images
        640x512: 310-470 ms that way, of which 30 are the recording)."""
        frames = np.asarray(arr)
        if frames.ndim not in (2, 3):
            raise ValueError("mismatch array shape. Must be 2D or 3D")
        rows, columns = frames.shape[-2:]
        stack = np.ascontiguousarray(frames, dtype=np.uint16).reshape(-1, rows, columns)
        if stack.shape[0] == 0:
            raise RuntimeError("No images in selected range to save")
        handle, name = tempfile.mkstemp(suffix=".h264")
        os.close(handle)
        encoded = Path(name)
        try:
            if times is None:  # what a raw file's time stamps come to on their way through ``to_h264`` (seconds, then nanoseconds again)
                times = [t * 1e9 for t in (np.arange(stack.shape[0], dtype=np.int64) * (1000000000 // 50)) * 1e-9]
            with IRSaver(str(encoded), columns, rows, rows, 8) as saver:
                saver.set_global_attributes({})
                saver.set_parameter("threads", cthreads)
                saver.set_parameter("codec", "h264")
                for pos in range(stack.shape[0]):
                    saver.add_image(stack[pos], times[pos], attributes={})
            movie = cls.from_filename(encoded)
        except BaseException:
            encoded.unlink(missing_ok=True)
            raise
        movie._owned_file = encoded
        if attrs is not None:
            movie.attributes = attrs
            movie._file_attributes.flush()
        return movie

    # ---- life cycle ----------------------------------------------------------------------------------------------------
    def close(self):
        fa, self._file_attributes = self._file_attributes, None
        if fa is not None:
            try:
                fa.close()
            except Exception:
                pass
        handle, self.handle = getattr(self, "handle", 0), 0
        if handle > 0:
            _abi.close_camera(handle)
        owned, self._owned_file = getattr(self, "_owned_file", None), None
        _remove_quietly(owned)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return "IRMovie({})".format(self.filename)

    # ---- calibration: digital levels only --------------------------------------------------------------------------------
    @property
    def calibrations(self):
        return _abi.supported_calibrations(self.handle)

    def _calibration_number(self, which):
        if which == 0:  # digital levels: always there, always first (the per-image path asks for nothing else)
            return 0
        names = self.calibrations
        if isinstance(which, str):
            if which in ("DL", "Digital Level"):
                return 0
            if which not in names:
                raise CalibrationNotFound("%s not in available calibrations : %s" % (which, names + ["DL"]))
            return names.index(which)
        number = int(which)
        if not 0 <= number < len(names):
            raise CalibrationNotFound("Available calibrations : %s. Calibration index out of range : %s" % (names, number))
        return number

    # short names of the calibrations, in the library's order (reference IRMovie.py:77, :169-170: "DL" for "Digital Level")
    _calibration_nickname_mapper = {"DL": "Digital Level"}

    @property
    def calibration(self):
        """the current calibration by its SHORT name where it has one ("DL"), as upstream answers"""
        short = list(self._calibration_nickname_mapper)
        return short[self._calibration_index] if self._calibration_index < len(short) else self.calibrations[self._calibration_index]

    @calibration.setter
    def calibration(self, value):
        self._calibration_index = self._calibration_number(value)

    def flip_calibration(self, flip_rl, flip_ud):
        _abi.flip_camera_calibration(self.handle, flip_rl, flip_ud)

    @property
    def support_emissivity(self):
        return _abi.support_emissivity(self.handle)

    # (reference IRMovie.py:401-423: a movie whose calibration takes no emissivity reads as 1 everywhere and refuses to be set)
    @property
    def global_emissivity(self):
        return _abi.get_global_emissivity(self.handle) if self.support_emissivity else 1.0

    @global_emissivity.setter
    def global_emissivity(self, value):
        if not self.support_emissivity:
            raise RuntimeError("Cannot set custom emissivity value for this handle")
        _abi.set_global_emissivity(self.handle, value)

    @property
    def emissivity(self):
        return _abi.get_emissivity(self.handle) if self.support_emissivity else np.ones(self.image_size, dtype=np.float32)

    @emissivity.setter
    def emissivity(self, emissivity_array):
        if not self.support_emissivity:
            raise RuntimeError("Cannot set custom emissivity value for this handle")
        _abi.set_emissivity(self.handle, emissivity_array)

    def calibrate(self, image, calib):
        """``calib`` applied to a digital-level image, as a new array (IRMovie.py:510-514); only calibration 0 exists here."""
        return _abi.calibrate_image(self.handle, image, calib)

    @property
    def calibration_files(self):
        try:
            return _abi.calibration_files(self.handle)
        except RuntimeError:
            return []

    # ---- geometry ----------------------------------------------------------------------------------------------------------
    @property
    def images(self):
        return _abi.get_image_count(self.handle)

    def __len__(self):
        return self.images

    @property
    def image_size(self):
        return _abi.get_image_size(self.handle)

    height = property(lambda self: self.image_size[0])
    width = property(lambda self: self.image_size[1])

    @property
    def filename(self):
        """a ``Path`` (None for a movie without a file), like upstream (IRMovie.py:341-344)"""
        name = _abi.get_filename(self.handle)
        return Path(name) if name else None

    @property
    def video_file_format(self):
        return _abi.video_file_format(self.filename)

    @property
    def is_file_uncompressed(self):
        return self.video_file_format in (FileFormat.PCR, FileFormat.WEST, FileFormat.PCR_ENCAPSULATED)

    # ---- images --------------------------------------------------------------------------------------------------------------
    def load_pos(self, pos, calibration=None, out=None):
        """Image number ``pos`` (bad-pixel repair and motion correction applied when enabled).  ``out``: a C-contiguous uint16 array of
        the image's shape to read into (not part of the reference's signature; slices use it to fill their stack in place)."""
        pos = int(pos)
        if self._shape is None:
            self._shape = _abi.get_image_size(self.handle)
        image = _abi.load_image(self.handle, pos, self._calibration_number(0 if calibration is None else calibration), self._shape, out)
        self._per_frame[pos] = _abi.get_attributes(self.handle)
        self._current = pos
        return image

    def load_secs(self, time, calibration=None):
        """The image whose time stamp is closest to ``time`` (seconds)."""
        if self.times is None:
            self.times = np.array(list(self.timestamps), dtype=np.float64)
        return self.load_pos(int(np.abs(self.times - time).argmin()), calibration)

    def _positions(self, selection):
        total = self.images
        first = selection.start or 0
        last = total if selection.stop is None or selection.stop == 0 else selection.stop
        stride = selection.step or 1
        first = first + total if first < 0 else first
        last = last + total if last < 0 else last
        return range(first, last, stride), math.ceil((last - first) / stride)

    def __getitem__(self, item):
        if isinstance(item, slice):
            positions, count = self._positions(item)
            stack = np.empty((count,) + tuple(self.image_size), dtype=np.uint16)
            if count:
                with _touch_ahead(stack):  # (the stack's pages are made by threads of their own while the images are read)
                    for row, pos in enumerate(positions):
                        self.load_pos(pos, self._calibration_index, out=stack[row])  # (in place: no copy of each image into the stack)
            return stack
        if isinstance(item, (int, np.integer)):
            return self.load_pos(int(item) + (self.images if item < 0 else 0), self._calibration_index)
        if isinstance(item, float):
            return self.load_secs(item, self._calibration_index)
        if isinstance(item, list) or (isinstance(item, np.ndarray) and item.ndim == 1):
            return np.array([self[e] for e in item])
        raise TypeError("unsupported index type")

    def __iter__(self):
        return (self.load_pos(pos, self._calibration_index) for pos in range(self.images))

    @property
    def data(self):
        return self[:]

    @property
    def tis(self):
        stack = self.data  # (a fresh array of this call: masked and shifted in place, no two further copies of the movie)
        stack &= _TI_MASK
        stack >>= _TI_SHIFT
        return stack

    # ---- time --------------------------------------------------------------------------------------------------------------------
    @property
    def timestamps(self):
        """seconds"""
        if self._seconds is None:
            self._seconds = np.fromiter((_abi.get_image_time(self.handle, pos) for pos in range(self.images)), dtype=np.float64) * 1e-9
        return self._seconds

    @timestamps.setter
    def timestamps(self, nanoseconds):
        self._seconds = np.array(nanoseconds) * 1e-9

    @property
    def frame_period(self):
        return np.diff(self.timestamps).mean().round(3)

    @property
    def duration(self):
        return (_abi.get_image_time(self.handle, self.images - 1) - _abi.get_image_time(self.handle, 0)) * 1e-9

    # ---- attributes -----------------------------------------------------------------------------------------------------------------
    @property
    def attributes(self):
        fa = self._file_attributes
        return _abi.get_global_attributes(self.handle) if fa is None else fa.attributes

    @attributes.setter
    def attributes(self, value):
        if self._file_attributes is not None:
            self._file_attributes.attributes = value

    @property
    def frame_attributes(self):
        """attributes of the image read last"""
        return self._per_frame.get(self._current, {})

    @property
    def frames_attributes(self):
        """The attributes of EVERY image as a table, one row per image (IRMovie.py:642-649: images not read yet are read for it).
        A pandas DataFrame like the reference's; values are the attribute bytes as stored."""
        import pandas as pd

        for pos in range(self.images):
            if pos not in self._per_frame:
                self.load_pos(pos, self._calibration_index)
        return pd.DataFrame({pos: self._per_frame[pos] for pos in range(self.images)}).T

    def _frame_attribute_getter(self, key):
        """one attribute of every image as floats (empty when no image carries it)"""
        try:
            values = self.frames_attributes[key]
        except KeyError:
            values = []
        return np.array(values, dtype=float)

    def to_thermavip(self, th_instance="Thermavip-1", player_id=0):
        """The reference hands the file to a running Thermavip viewer through shared memory (IRMovie.py:660-676); that bridge is outside
        this build: like the reference without a Thermavip instance, nothing is opened and None is returned."""
        return None

    # ---- filters applied while reading -----------------------------------------------------------------------------------------------
    @property
    def bad_pixels_correction(self):
        return self._bp_on

    @bad_pixels_correction.setter
    def bad_pixels_correction(self, value):
        self._bp_on = bool(value)
        _abi.enable_bad_pixels(self.handle, self._bp_on)

    @property
    def registration_file(self):
        return self._reg_file

    @registration_file.setter
    def registration_file(self, value):
        _abi.load_motion_correction_file(self.handle, str(value))
        self._reg_file = Path(value)

    @property
    def registration(self):
        return _abi.motion_correction_enabled(self.handle)

    @registration.setter
    def registration(self, value):
        _abi.enable_motion_correction(self.handle, bool(value))

    # ---- writing a (part of a) movie again ---------------------------------------------------------------------------------------------
    def to_h264(self, dst_filename, start_img=0, count=-1, clevel=8, attrs=None, times=None, frame_attributes=None, cthreads=8, cfiles=None):
        """Record images ``start_img .. start_img + count`` into a new file, with their attributes and time stamps."""
        available = self.images - start_img
        count = available if count < 0 else min(count, available)
        if count == 0:
            raise RuntimeError("No images in selected range to save")
        if frame_attributes is not None and len(frame_attributes) != count:
            raise RuntimeError("Given frame attributes are not equal to the number of saved images")
        global_attrs = dict(self.attributes) if attrs is None else attrs
        for stale in ("MIN_T", "MIN_T_HEIGHT", "STORE_IT"):  # they describe how THIS file stores its pixels
            global_attrs.pop(stale, None)
        stamps = [t * 1e9 for t in self.timestamps] if times is None else times
        rows, columns = self.image_size
        with IRSaver(str(dst_filename), columns, rows, rows, clevel) as saver:
            saver.set_global_attributes(global_attrs)
            saver.set_parameter("threads", cthreads)
            saver.set_parameter("codec", "h264")
            # A recording of this library goes from its loader to the saver without leaving the device (chunks decoded into device memory,
            # their images copied device to device into the chunk the saver assembles, attributes with them): 6 us an image.  Anything
            # else - raw files, read-back filters switched on, attributes given per image - goes image by image through host memory.
            if frame_attributes is None and _abi.transcode_images(self.handle, saver.handle, start_img, count,
                                                                  [int(stamps[pos]) for pos in range(start_img, start_img + count)]):
                return
            # (Measured and not kept for that path: a thread reading ahead of the recording one.  Image by image 33-70 us an image, in
            # stacks of sixteen through bulk library calls 34-37 us - against 30-33 us for one thing after the other as below.  Reading alone
            # is 17 us an image, recording alone 17-19: the two do not overlap, because the kernels' own traffic over the link does not - a
            # chunk's encode reading its frames from host memory (705 us) and a chunk's decode writing its frames there (660 us) take
            # 1 210-1 270 us together on two streams, with or without disjoint compute-unit masks, while the copy engines' transfers up and
            # down do overlap (587 + 585 -> 686 us), and so do a copy call upwards and the decode's writes (581 + 660 -> 812): it is the
            # kernels' reads of host memory that do not share the link.  profiles/r05_link_duplex.txt.)
            for written, pos in enumerate(range(start_img, start_img + count)):
                image = self.load_pos(pos, 0)
                saver.add_image(image, stamps[pos], attributes=self.frame_attributes if frame_attributes is None else frame_attributes[written])

    def _build_outfile(self):
        """where pcr2h264 writes by default: beside the movie, suffix ``.h264``; a movie that is encoded already names itself
        (IRMovie.py:533-545)"""
        if self.video_file_format != FileFormat.H264:
            source = str(self._owned_file or self.filename)
            return os.path.abspath(os.path.splitext(source)[0] + ".h264")
        return self.filename

    def pcr2h264(self, outfile=None, overwrite=False, **kwargs):
        """A raw (PCR) movie re-recorded through the codec; ``kwargs`` go to ``to_h264``.  A destination that exists is kept unless
        ``overwrite``.  Returns the destination file name (IRMovie.py:520-531)."""
        outfile = outfile or self._build_outfile()
        if not os.path.exists(outfile) or overwrite:
            self.to_h264(outfile, **kwargs)
        return outfile
