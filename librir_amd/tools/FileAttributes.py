"""File attributes object (reference src/python/librir/tools/FileAttributes.py): global attributes,
per-frame attributes and timestamps stored in the trailer of a video file."""
import numpy as np

from . import rir_tools as rt


class FileAttributes(object):
    def __init__(self, handle):
        self.handle = handle
        self._timestamps = None
        self._attributes = None

    @classmethod
    def from_filename(cls, filename):
        return cls(rt.attrs_open_file(filename))

    @classmethod
    def from_buffer(cls, buffer):
        return cls(rt.attrs_open_buffer(buffer))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.discard()
        except Exception:
            pass

    def close(self):
        """write the attributes to the file trailer and release the object"""
        if self.handle > 0:
            self._push()
            rt.attrs_close(self.handle)
            self.handle = 0

    def is_open(self):
        return self.handle > 0

    def discard(self):
        if self.handle > 0:
            rt.attrs_discard(self.handle)
            self.handle = 0

    def flush(self):
        self._push()
        rt.attrs_flush(self.handle)

    def _push(self):
        if self._attributes is not None:
            rt.attrs_set_global_attributes(self.handle, self._attributes)
        if self._timestamps is not None:
            rt.attrs_set_times(self.handle, self._timestamps)

    @property
    def attributes(self):
        if self._attributes is None:
            self._attributes = rt.attrs_global_attributes(self.handle)
        return self._attributes

    @attributes.setter
    def attributes(self, value):
        self._attributes = dict(value)
        rt.attrs_set_global_attributes(self.handle, self._attributes)

    @property
    def timestamps(self):
        if self._timestamps is None:
            self._timestamps = rt.attrs_timestamps(self.handle)
        return self._timestamps

    @timestamps.setter
    def timestamps(self, value):
        self._timestamps = np.array(value, dtype=np.int64)
        rt.attrs_set_times(self.handle, self._timestamps)

    def frame_count(self):
        return rt.attrs_image_count(self.handle)

    def frame_attributes(self, frame_index):
        return rt.attrs_frame_attributes(self.handle, frame_index)

    def set_frame_attributes(self, frame_index, attributes):
        rt.attrs_set_frame_attributes(self.handle, frame_index, attributes)
