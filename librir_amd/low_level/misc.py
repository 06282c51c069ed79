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
_has_host_alloc = hasattr(_lib, "rir_host_alloc")
if _has_host_alloc:
    _lib.rir_host_alloc.restype = ct.c_void_p
    _lib.rir_host_alloc.argtypes = [ct.c_int64]
    _lib.rir_host_free.restype = None
    _lib.rir_host_free.argtypes = [ct.c_void_p]
    _lib.rir_host_is_page_locked.argtypes = [ct.c_void_p, ct.c_int64]
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
    on x86-64 (the only host this library is built for) and cannot change what they leave (run under ThreadSanitizer beside the copies, the race
    suppressed by name and the copied bytes checked: scripts/tsan_host_copy.sh)."""

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


# ---- result arrays in page-locked memory (opt-in) ------------------------------------------------------------------------------------
_free_result_blocks = {}  # nbytes -> a few memory blocks whose arrays have died
# RIR_PINNED_RESULTS=1 in the environment or results_in_page_locked_memory(True): see result_buffer
_pinned_results = [os.environ.get("RIR_PINNED_RESULTS", "0") == "1"]


def results_in_page_locked_memory(on=True):
    """The arrays that ``translate`` / ``gaussian_filter`` / ``BadPixels.correct`` / ``filter_chain`` return are built on page-locked memory of the
    library (``rir_host_alloc``) from now on (or, ``on=False``, on ordinary memory again: the default).  The library's entry points find buffers
    that lie in such memory and run their kernels on them in place - no staging copy in, none out - so a chain of calls in which one call's
    result is the next call's input (the reference's ``translate(gaussian_filter(bad_pixels.correct(img)))``) stages only the first image:
    5.0-5.1 k images/s instead of 3.9-4.0 k for 640x512 (tests/perf/three_call_probe.py).  Not the default because a result that is consumed by
    numpy code instead is read cold - the GPU wrote it, and page-locked memory lies on the GPU's NUMA node - where the staged copy leaves it in
    the caches of the copying cores: ``x.astype(...)`` of such a result takes 80-220 us instead of 110.  Returns the previous setting."""
    old = _pinned_results[0]
    _pinned_results[0] = bool(on) and _has_host_alloc
    return old


class _PinnedBlock:
    """a block of rir_host_alloc (page-locked memory of the library) with the buffer interface numpy builds an array on"""

    __slots__ = ("ptr", "buf")

    def __init__(self, ptr, n):
        self.ptr = ptr
        self.buf = (ct.c_char * n).from_address(ptr)

    def __del__(self):
        try:
            _lib.rir_host_free(ct.c_void_p(self.ptr))
        except Exception:  # (interpreter shutdown)
            pass


def result_buffer(shape, dtype):
    """A fresh C-contiguous array for the result of a library call, as the reference's wrappers return (a new array per call; contents NOT
    cleared: the library writes every element or fails).  Ordinary memory by default.  With ``results_in_page_locked_memory`` the memory of a
    result of 64 KB or more is a block of ``rir_host_alloc``, recycled where that is safe: the array handed out is built directly on the block
    (views of it keep IT alive: numpy stops collapsing ``base`` chains at the first non-array), and a weak-reference finaliser returns the block
    to a small pool when the array object is collected - when neither the caller nor any view refers to it any more.  No reference counts are
    inspected: an array the caller keeps is never written to again.  Where the library says no (no device, its limit on such memory reached):
    ordinary memory."""
    dt = np.dtype(dtype)
    shape = tuple(int(x) for x in shape)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    if n < (64 << 10) or not _pinned_results[0]:
        return np.empty(shape, dtype=dt)
    import weakref

    pool = _free_result_blocks.setdefault(n, [])
    try:
        block = pool.pop()
    except IndexError:
        ptr = _lib.rir_host_alloc(ct.c_int64(n))
        if not ptr:
            return np.empty(shape, dtype=dt)
        block = _PinnedBlock(ptr, n)
    a = np.ndarray(shape, dtype=dt, buffer=block.buf)
    f = weakref.finalize(a, _recycle_result_block, pool, block)
    f.atexit = False
    return a


def _recycle_result_block(pool, block):
    if len(pool) < 4:
        pool.append(block)
