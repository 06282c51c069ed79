"""IRSaver: record 16-bit IR images with global / per-frame attributes
(reference src/python/librir/video_io/IRSaver.py:22-190).  Frames are compressed on the MI355X by
the lossless block codec; the file is only complete after ``close()``."""
import numpy as np

from .rir_video_io import (h264_add_image_lossless, h264_add_image_lossy, h264_add_loss, h264_close_file, h264_get_high_errors,
                           h264_get_low_errors, h264_open_file, h264_set_global_attributes, h264_set_parameter)


class IRSaver(object):
    def __init__(self, outfile=None, width=None, height=None, lossy_height=None, clevel=0):
        self.handle = 0
        self.width = 0
        self.height = 0
        self.global_attrs = {}
        self.params = {}
        if outfile is not None and width is not None and height is not None:
            self.filename = outfile
            self.open(outfile, width, height, lossy_height)
            self.set_parameter("compressionLevel", str(clevel))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def is_open(self):
        return self.handle > 0

    def close(self):
        """Finish the file (flush the last chunk, write the index and the attribute trailer)."""
        if self.handle > 0:
            h264_close_file(self.handle)
            self.handle = 0

    def open(self, outfile, width, height, lossy_height=None):
        self.close()
        self.handle = h264_open_file(outfile, width, height, lossy_height)
        self.width = width
        self.height = height
        self.lossy_height = lossy_height
        self.filename = outfile
        if len(self.global_attrs) > 0:
            h264_set_global_attributes(self.handle, self.global_attrs)
        for k in self.params:
            h264_set_parameter(self.handle, k, self.params[k])
        self.global_attrs = {}
        self.params = {}

    def set_parameter(self, param, value):
        """compressionLevel, lowValueError, highValueError, codec, GOP, threads, slices, stdFactor,
        inputCamera, removeBadPixels, subtractMin, subtractLocalMin, runningAverage."""
        if self.is_open():
            h264_set_parameter(self.handle, param, str(value))
        else:
            self.params[param] = str(value)

    def set_global_attributes(self, attributes):
        if self.is_open():
            h264_set_global_attributes(self.handle, attributes)
        else:
            self.global_attrs = attributes

    def _check(self, image, what):
        image = np.asarray(image)
        if image.ndim != 2 or image.shape[1] != self.width or image.shape[0] != self.height:
            raise RuntimeError(what)
        return image

    def add_image(self, image, timestamp, attributes=dict()):
        h264_add_image_lossless(self.handle, self._check(image, "wrong image dimension"), timestamp, attributes)

    def add_image_lossy(self, image_DL, timestamp, attributes=None):
        h264_add_image_lossy(self.handle, self._check(image_DL, "wrong DL image dimension"), timestamp, attributes)

    def add_loss(self, image):
        return h264_add_loss(self.handle, self._check(image, "wrong DL image dimension"))

    def get_low_errors(self):
        return h264_get_low_errors(self.handle)

    def get_high_errors(self):
        return h264_get_high_errors(self.handle)
