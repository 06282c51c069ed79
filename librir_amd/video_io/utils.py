"""Helpers around whole movie files, same interface as librir's ``video_io.utils``
(reference src/python/librir/video_io/utils.py:13-68): ``split_rush`` cuts a movie into pieces of a fixed number of images,
``is_ir_file_corrupted`` tells whether a file opens as a movie at all."""
import logging
from pathlib import Path

from . import rir_video_io as _abi
from .IRMovie import IRMovie
from .IRSaver import IRSaver

logger = logging.getLogger(__name__)


def split_rush(filename, index=None, step=30, dest_folder=None):
    """``filename`` cut into movies of ``step`` images each, written as ``<dest_folder>/<name>.h264`` with the names taken from
    ``index`` (default 0, 1, 2, ...: as many pieces as ``images // step`` - what is left over after the last whole piece is not
    written, like upstream, where the list of names ends first); a piece whose file exists already is kept.  Images are stamped
    20 ms apart, from 0, inside every piece.  Returns the list of paths."""
    source = Path(filename)
    folder = source.parent if dest_folder is None else Path(dest_folder)
    pieces = []
    with IRMovie.from_filename(source) as movie:
        names = range(movie.images // step) if index is None else index
        height, width = movie.height, movie.width
        for first, name in zip(range(0, movie.images, step), names):
            if isinstance(name, float):
                name = round(name, 2)
            target = folder / "{}.h264".format(name)
            target.parent.mkdir(exist_ok=True, parents=True)
            if not target.exists():
                with IRSaver(target, width=width, height=height) as saver:
                    many = min(step, movie.images - first)
                    # (a recording of this library is cut on the device; anything else image by image)
                    if not _abi.transcode_images(movie.handle, saver.handle, first, many, [int(k * 20e6) for k in range(many)], keep_attributes=False):
                        for k, image in enumerate(movie[first:first + step]):
                            saver.add_image(image, k * 20e6)
            pieces.append(target)
    return pieces


def check_ir_file(filename):
    """opens and closes the movie (raises RuntimeError when it cannot be opened)"""
    with IRMovie.from_filename(filename):
        pass


def is_ir_file_corrupted(filename):
    """False when the file opens as a movie, True when it does not."""
    try:
        check_ir_file(filename)
    except RuntimeError as why:
        logger.warning("filename '%s' could not be opened : %s", filename, why)
        return True
    return False
