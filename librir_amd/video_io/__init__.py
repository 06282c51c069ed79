"""Mirror of librir's ``video_io`` Python package (reference src/python/librir/video_io/)."""
from .IRMovie import CalibrationNotFound, InvalidMovie, IRMovie
from .IRSaver import IRSaver
from .rir_video_io import *  # noqa: F401,F403
from .rir_video_io import FileFormat  # noqa: F401

from .utils import is_ir_file_corrupted, split_rush  # noqa: E402,F401

__all__ = ["IRMovie", "IRSaver", "FileFormat", "InvalidMovie", "CalibrationNotFound", "split_rush", "is_ir_file_corrupted"]
