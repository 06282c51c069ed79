"""Library loading, mirroring librir's ``low_level.misc`` (reference
src/python/librir/low_level/misc.py:98-139): the handles ``_tools``,
``_signal_processing`` and ``_video_io`` exist, but here they all resolve to the one HIP shared
object ``libs/librir_amd.so`` (built in-tree by ``librir_amd.build``); ``_geometry`` is None (the polygon library is not part of this build).

There is no CPU implementation behind this package: if the shared object is missing, importing
fails with an explicit message; if no HIP device is present, the compute entry points return their
error codes and the Python wrappers raise ``RuntimeError``.
"""
import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RIR_LIBRARY_VARIANT=testhooks: the build with the fault-injection hooks compiled in (librir_amd/build.py) - for the tests that force a
# bail-out path (tests/hook_cases.py); anything else: the product library
_LIB_PATH = os.path.join(_HERE, "libs", "librir_amd_testhooks.so" if os.environ.get("RIR_LIBRARY_VARIANT") == "testhooks" else "librir_amd.so")


def _load():
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and must map it before
    # this library's dependency on the system one is resolved (same SONAME -> the loader then
    # reuses the copy already mapped).  torch is optional plumbing; without it the system runtime
    # under /opt/rocm is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            "librir_amd: %s is missing - build it with `python -m librir_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback" % _LIB_PATH
        )
    return ct.CDLL(_LIB_PATH)


_lib = _load()
# same module-level names as the reference wrapper
_tools = _lib
_geometry = None  # (the polygon library is not part of this build: a drop-in keeps the reference's own)
_signal_processing = _lib
_video_io = _lib


def toString(ar):
    """A char array / bytes object as str: ASCII, else UTF-8, every NUL dropped (reference low_level/misc.py:55-60; a ``str`` argument
    is a TypeError there too - ``bytes("a")`` - and its tests hold that, tests/python/test_rir.py:29-33)."""
    raw = bytes(ar)
    try:
        return raw.decode("ascii").replace("\x00", "")
    except UnicodeDecodeError:
        return raw.decode("utf8").replace("\x00", "")


def toBytes(s):
    if isinstance(s, bytes):
        return s
    return str(s).encode("utf-8")


def get_memory_folder():
    """Folder of on-disk caches: ``$LIBRIR_TEMP_FOLDER`` or the system's temporary folder, with a last component ending in "cache";
    created when missing (reference low_level/misc.py:39-49)."""
    import os
    import tempfile
    from pathlib import Path

    folder = Path(os.getenv("LIBRIR_TEMP_FOLDER") or tempfile.gettempdir())
    if not folder.name.endswith("cache"):
        folder = folder / "cache"
    if not folder.exists():
        folder.mkdir()
        folder.chmod(0o775)
    return folder


class touch_ahead:
    """``with touch_ahead(stack): fill stack`` - a FRESH array (``np.empty``) that the library is about to fill image by image gets its pages
    made by threads of their own, ahead of the copies: a first touch is otherwise a page fault and a zeroed (huge) page under the threads
    that copy each image in - for the stack of a 1 000-image slice, twice the time of the reads themselves.  One thread makes pages at
    about 25 GB/s, the reads fill 40 GB/s: three threads take the array's 16 MiB pieces in turn, so that the made part grows from the front.
    The threads hold the array until they are through; contents are left as they are (``rir_host_touch``: an atomic compare-and-swap of a
    byte with itself per page).  By the letter of the C++ memory model that is a data race with the plain stores of the copies that may be
    filling the same page; it is meant: a locked read-modify-write that puts back what it read is serialised with those stores per cache line
    on x86-64 (the only host this library is built for) and cannot change what they leave."""

    PIECE = 16 << 20
    THREADS = 3

    def __init__(self, array):
        self.threads = []
        if array.nbytes >= (8 << 20) and array.flags.c_contiguous and array.flags.writeable:
            import threading

            base, total = array.ctypes.data, array.nbytes

            def run(k, keep=array):  # (`keep`: the array lives as long as this thread)
                for start in range(k * self.PIECE, total, self.THREADS * self.PIECE):
                    _lib.rir_host_touch(ct.c_void_p(base + start), ct.c_int64(min(self.PIECE, total - start)))

            for k in range(self.THREADS):
                self.threads.append(threading.Thread(target=run, args=(k,), daemon=True))
                self.threads[-1].start()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for t in self.threads:
            t.join()
        return False


def createZeroArrayHandle(shape, _dtype):
    """For a given shape and dtype, create a zero filled array and its ctype handle (reference low_level/misc.py:88-95)"""
    ar = np.zeros(shape, dtype=_dtype)
    if str(ar.dtype) == "|S1":
        handle = ar.ctypes.data_as(ct.POINTER(ct.c_char))
    else:
        handle = ar.ctypes.data_as(ct.POINTER(np.ctypeslib.as_ctypes_type(ar.dtype)))
    return (ar, handle)


def loadDlls():
    """Load the library (reference low_level/misc.py:98-139 loads its four; here they are one file, loaded when this module is
    imported: calling this again hands the same handles back)"""
    return _tools, _signal_processing, _video_io


def toCharP(obj):
    """str (ASCII) or bytes as the bytes a ``char *`` argument takes; anything else through ``bytes()`` (reference :78-85, so that
    ``toCharP(1) == b"\\x00"``)."""
    if isinstance(obj, str):
        return obj.encode("ascii")
    if isinstance(obj, bytes):
        return obj
    return bytes(obj)


def toArray(string):
    """str or bytes as a numpy array of single characters (reference :63-75)"""
    raw = string if isinstance(string, bytes) else str(string).encode("ascii")
    out = np.zeros(len(string), dtype="c")
    for i in range(len(raw)):
        out[i] = raw[i:i + 1]
    return out


def last_error():
    buf = ct.create_string_buffer(1024)
    n = ct.c_int(1024)
    _lib.get_last_log_error.argtypes = [ct.c_char_p, ct.POINTER(ct.c_int)]
    if _lib.get_last_log_error(buf, ct.byref(n)) != 0:
        buf = ct.create_string_buffer(n.value + 1)
        _lib.get_last_log_error(buf, ct.byref(n))
    return buf.raw[: n.value].decode("utf-8", errors="replace")
