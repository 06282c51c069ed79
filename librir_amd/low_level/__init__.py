"""Mirror of librir's ``low_level`` package (reference src/python/librir/low_level/__init__.py): the library handles and the small
conversion helpers.  The three handles are one library here; the geometry library is not part of this build (a drop-in keeps the
reference's own) and its handle is None."""
from .misc import _geometry, _signal_processing, _tools, _video_io, createZeroArrayHandle, loadDlls, toArray, toCharP, toString

__all__ = ["_tools", "_geometry", "_signal_processing", "_video_io", "toString", "toArray", "toCharP", "createZeroArrayHandle", "loadDlls"]
