"""Mirror of librir's ``video_io`` Python package (reference src/python/librir/video_io/)."""
from .IRMovie import FileFormat, InvalidMovie, IRMovie
from .IRSaver import IRSaver
from .rir_video_io import *  # noqa: F401,F403

__all__ = ["IRMovie", "IRSaver", "FileFormat", "InvalidMovie"]
