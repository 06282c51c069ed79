"""Where the time of the reference's three-call configs[2] path goes (bad_pixels_correct -> gaussian_filter -> translate, one 640x512 image a
call through host pointers): every entry point through raw ctypes on arrays allocated once, through the Python wrapper, and the chain.
    python tests/perf/three_call_probe.py [images] [pinned]"""
import ctypes as ct
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.low_level.misc import _signal_processing as sp  # noqa: E402
from librir_amd.signal_processing import BadPixels, gaussian_filter, translate  # noqa: E402
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 300
if len(sys.argv) > 2 and sys.argv[2] == "pinned":  # results in the library's page-locked memory (opt-in): what feeds the next call stays in place
    from librir_amd.low_level.misc import results_in_page_locked_memory

    results_in_page_locked_memory(True)
    print("results in page-locked memory")
h, w = 512, 640
fr = inject_bad_pixels(s1_noisy_background(m, h, w), 200)
bp = BadPixels(fr[0])
u_in, u_out = fr[0].copy(), np.empty((h, w), np.uint16)
f_in, f_out = fr[0].astype(np.float32), np.empty((h, w), np.float32)
back_f, back_u = np.zeros(1, np.float32), np.zeros(1, np.uint16)


def per_call(fn, n=m):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


rows = [
    ("bad_pixels_correct   raw ctypes", lambda: sp.bad_pixels_correct(bp.handle, u_in.ctypes.data, u_out.ctypes.data)),
    ("bad_pixels_correct   wrapper", lambda: bp.correct(u_in)),
    ("gaussian_filter f32  raw ctypes", lambda: sp.gaussian_filter(f_in.ctypes.data, f_out.ctypes.data, w, h, np.float32(0.75))),
    ("gaussian_filter f32  wrapper", lambda: gaussian_filter(f_in, 0.75)),
    ("gaussian_filter u16 -> f32 wrapper (astype inside)", lambda: gaussian_filter(u_in, 0.75)),
    ("translate f32        raw ctypes", lambda: sp.translate(ord("f"), f_in.ctypes.data, f_out.ctypes.data, w, h, np.float32(1.25), np.float32(-2.5), back_f.ctypes.data, b"nearest")),
    ("translate f32        wrapper", lambda: translate(f_in, 1.25, -2.5, "nearest")),
    ("translate u16        raw ctypes", lambda: sp.translate(ord("H"), u_in.ctypes.data, u_out.ctypes.data, w, h, np.float32(1.25), np.float32(-2.5), back_u.ctypes.data, b"nearest")),
    ("astype(float32) of a uint16 image", lambda: u_in.astype(np.float32)),
    ("np.empty + copy of a float32 image", lambda: np.copyto(np.empty((h, w), np.float32), f_in)),
]
for name, fn in rows:
    print("%-52s %7.1f us" % (name, per_call(fn)), flush=True)


def chain(i):
    x = bp.correct(fr[i])
    x = gaussian_filter(x.astype(np.float32), 0.75)
    return translate(x, 1.25, -2.5, "nearest")


def chain_as_upstream_tests_do(i):  # (gaussian_filter converts by itself: reference rir_signal_processing.py:85-113)
    return translate(gaussian_filter(bp.correct(fr[i]), 0.75), 1.25, -2.5, "nearest")


for name, fn in (("three calls per image (astype by the caller)", chain), ("three calls per image (gaussian_filter converts)", chain_as_upstream_tests_do)):
    fn(0)
    t0 = time.perf_counter()
    for i in range(m):
        fn(i)
    dt = time.perf_counter() - t0
    print("%-52s %7.1f us  = %6.0f images/s" % (name, dt / m * 1e6, m / dt), flush=True)

# ---- where the chain's time goes, call by call (with `pinned`: results in the library's page-locked memory feed the next call in place)
acc = {}
for i in range(m):
    t0 = time.perf_counter()
    a = bp.correct(fr[i])
    t1 = time.perf_counter()
    g = gaussian_filter(a, 0.75)
    t2 = time.perf_counter()
    t = translate(g, 1.25, -2.5, "nearest")
    t3 = time.perf_counter()
    f = a.astype(np.float32)
    t4 = time.perf_counter()
    g2 = gaussian_filter(f, 0.75)
    t5 = time.perf_counter()
    for k, v in (("bp.correct(the caller's image)", t1 - t0), ("gaussian_filter(that result, uint16)", t2 - t1),
                 ("translate(that result, float32)", t3 - t2), ("astype(float32) of the first result", t4 - t3),
                 ("gaussian_filter(that float32 copy)", t5 - t4)):
        acc[k] = acc.get(k, 0.0) + v
    del a, g, t, f, g2
for k, v in acc.items():
    print("in the chain: %-52s %7.1f us" % (k, v / m * 1e6), flush=True)
