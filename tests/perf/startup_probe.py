"""What the first call of a process pays for (development aid): device runtime start, first kernel launch (code object load), page-locking."""
import ctypes as ct
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.perf_counter()
from librir_amd.low_level.misc import _lib  # noqa: E402

t1 = time.perf_counter()
_lib.rir_device_available.restype = ct.c_int
_lib.rir_device_available()
t2 = time.perf_counter()
import numpy as np  # noqa: E402

a = np.zeros((64, 64), np.uint16)
b = np.zeros((64, 64), np.uint16)
off = ct.c_float
_lib.translate.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float, ct.c_void_p, ct.c_char_p]
_lib.translate(ord("H"), a.ctypes.data, b.ctypes.data, 64, 64, 1.0, 0.0, None, b"nearest")
t3 = time.perf_counter()
_lib.translate(ord("H"), a.ctypes.data, b.ctypes.data, 64, 64, 1.0, 0.0, None, b"nearest")
t4 = time.perf_counter()
print("load library %.1f ms | first HIP call (runtime start) %.1f ms | first kernel (code object, stream, staging) %.1f ms | the same call again %.2f ms" %
      ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
