"""Recording side of the Python interface: ``IRSaver`` writes 16-bit IR images, one per call, with a time stamp
and optional per-frame attributes, plus global attributes for the whole file.

Public interface (constructor arguments, method names, attribute names) as the reference class
``librir.video_io.IRSaver`` (reference src/python/librir/video_io/IRSaver.py:22-190), so code written against it
runs unchanged; the implementation is this build's: frames go to the MI355X in chunks of GOP images, are packed by
the lossless block codec (or first pass through the bounded-loss step), and the container is finished by
``close()`` - a file that was not closed is not readable.
"""
import numpy as np

from . import rir_video_io as _abi

# parameters accepted by set_parameter (reference h264.cpp:1709-1781); values travel as strings
KNOWN_PARAMETERS = ("compressionLevel", "lowValueError", "highValueError", "codec", "GOP", "threads", "slices", "stdFactor", "inputCamera",
                    "removeBadPixels", "subtractMin", "subtractLocalMin", "runningAverage")


class IRSaver(object):
    """``IRSaver(outfile, width, height, lossy_height=None, clevel=0)`` opens at once; ``IRSaver()`` followed by
    ``open(...)`` opens later - parameters and global attributes set in between are kept and applied on open."""

    def __init__(self, outfile=None, width=None, height=None, lossy_height=None, clevel=0):
        self._handle = 0
        self._shape = (0, 0)  # (height, width) of the images this saver accepts
        self._lossy_rows = None
        self._waiting_params = {}  # name -> str, set before the file exists
        self._waiting_globals = None
        self.filename = None
        if None not in (outfile, width, height):
            self.open(outfile, width, height, lossy_height)
            self.set_parameter("compressionLevel", clevel)

    # ---- what the reference exposes as plain attributes ----------------------------------------------------------
    handle = property(lambda self: self._handle)
    width = property(lambda self: self._shape[1])
    height = property(lambda self: self._shape[0])
    lossy_height = property(lambda self: self._lossy_rows)

    @property
    def params(self):
        """parameters waiting for ``open`` (empty once the file is open)"""
        return dict(self._waiting_params)

    @property
    def global_attrs(self):
        return {} if self._waiting_globals is None else dict(self._waiting_globals)

    # ---- life cycle ------------------------------------------------------------------------------------------------
    def is_open(self):
        return self._handle > 0

    def open(self, outfile, width, height, lossy_height=None):
        """(Re)open on a new output file; whatever was being written is finished first."""
        self.close()
        self._handle = _abi.h264_open_file(outfile, width, height, lossy_height)
        self._shape = (int(height), int(width))
        self._lossy_rows = lossy_height
        self.filename = outfile
        pending_globals, self._waiting_globals = self._waiting_globals, None
        pending_params, self._waiting_params = self._waiting_params, {}
        if pending_globals:
            _abi.h264_set_global_attributes(self._handle, pending_globals)
        for name, value in pending_params.items():
            _abi.h264_set_parameter(self._handle, name, value)

    def close(self):
        """Flush the last chunk, write the chunk index and the attribute trailer.  Safe to call twice."""
        handle, self._handle = self._handle, 0
        if handle > 0:
            _abi.h264_close_file(handle)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the library may be gone already
            pass

    # ---- settings ----------------------------------------------------------------------------------------------------
    def set_parameter(self, param, value):
        """One of ``KNOWN_PARAMETERS``; the value is passed as text.  Before ``open`` it is remembered."""
        text = str(value)
        if self._handle > 0:
            _abi.h264_set_parameter(self._handle, param, text)
        else:
            self._waiting_params[param] = text

    def set_global_attributes(self, attributes):
        """dict str -> str / bytes, stored once per file (replaces the previous set)."""
        if self._handle > 0:
            _abi.h264_set_global_attributes(self._handle, attributes)
        else:
            self._waiting_globals = attributes

    # ---- frames ------------------------------------------------------------------------------------------------------
    def _frame(self, image, message):
        frame = np.asarray(image)
        if frame.ndim != 2 or frame.shape != self._shape:
            raise RuntimeError(message)
        return frame

    def add_image(self, image, timestamp, attributes=dict()):
        """Lossless: the image read back is the image given.  ``timestamp`` in nanoseconds."""
        _abi.h264_add_image_lossless(self._handle, self._frame(image, "wrong image dimension"), timestamp, attributes)

    def add_image_lossy(self, image_DL, timestamp, attributes=None):
        """Bounded loss: pixels may move by at most lowValueError / highValueError around their reference value
        (rows below ``lossy_height`` only), which makes the temporal residuals smaller."""
        _abi.h264_add_image_lossy(self._handle, self._frame(image_DL, "wrong DL image dimension"), timestamp, attributes)

    def add_loss(self, image):
        """The same loss, returned instead of recorded."""
        return _abi.h264_add_loss(self._handle, self._frame(image, "wrong DL image dimension"))

    def get_low_errors(self):
        return _abi.h264_get_low_errors(self._handle)

    def get_high_errors(self):
        return _abi.h264_get_high_errors(self._handle)
