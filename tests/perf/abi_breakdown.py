"""Where a frame's time goes on the per-frame C ABI (development aid, GPU box):
    python tests/perf/abi_breakdown.py [frames]
host copy rate of one frame, the raw C calls (ctypes, arguments prepared once) and the Python wrapper
(IRSaver.add_image / IRMovie[i]) side by side, 640x512 uint16."""
import ctypes as ct
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.low_level.misc import _video_io as V  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
h, w = 512, 640
fr = s1_noisy_background(n, h, w)


def rate(dt):
    return "%7.0f frames/s (%5.1f us)" % (n / dt, dt / n * 1e6)


# host copy of a frame: cold destination rotating over a 1000-frame buffer (what a staging copy sees) and hot
dst = np.empty_like(fr)
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(n):
        np.copyto(dst[i], fr[i])
    dt = time.perf_counter() - t0
print("np.copyto frame -> frame of a large buffer : %s = %.1f GB/s" % (rate(dt), n * fr[0].nbytes / dt / 1e9), flush=True)
one = np.empty_like(fr[0])
t0 = time.perf_counter()
for i in range(n):
    np.copyto(one, fr[i])
dt = time.perf_counter() - t0
print("np.copyto frame -> one hot frame           : %s = %.1f GB/s" % (rate(dt), n * fr[0].nbytes / dt / 1e9), flush=True)

with tempfile.TemporaryDirectory() as d:
    for rep in range(3):
        p = os.path.join(d, "w%d.h264" % rep)
        t0 = time.perf_counter()
        with IRSaver(p, w, h, h) as s:
            for i in range(n):
                s.add_image(fr[i], i * 1000)
            tl = time.perf_counter() - t0
        te = time.perf_counter() - t0
        print("IRSaver.add_image                          : %s   (loop alone %s, close %.1f ms)" % (rate(te), rate(tl), (te - tl) * 1e3), flush=True)
    # raw C calls, arguments prepared once
    fn = V.h264_add_image_lossless
    for rep in range(3):
        p = os.path.join(d, "r%d.h264" % rep)
        hd = V.h264_open_file(p.encode(), w, h, h)
        ptrs = [fr[i].ctypes.data for i in range(n)]
        t0 = time.perf_counter()
        for i in range(n):
            fn(hd, ptrs[i], i * 1000, 0, None, None, None, None)
        tl = time.perf_counter() - t0
        V.h264_close_file(hd)
        te = time.perf_counter() - t0
        print("h264_add_image_lossless (raw ctypes)       : %s   (loop alone %s, close %.1f ms)" % (rate(te), rate(tl), (te - tl) * 1e3), flush=True)
    for rep in range(3):
        t0 = time.perf_counter()
        with IRMovie.from_filename(p) as mov:
            for i in range(n):
                img = mov[i]
        td = time.perf_counter() - t0
        print("IRMovie[i]                                 : %s" % rate(td), flush=True)
    assert np.array_equal(img, fr[n - 1])
    for rep in range(3):
        ff = ct.c_int(0)
        t0 = time.perf_counter()
        cam = V.open_camera_file(p.encode(), ct.byref(ff))
        buf = np.empty((h, w), np.uint16)
        ptr = buf.ctypes.data
        for i in range(n):
            V.load_image(cam, i, 0, ptr)
        tl = time.perf_counter() - t0
        V.close_camera(cam)
        print("load_image into one buffer (raw ctypes)    : %s" % rate(tl), flush=True)
    assert np.array_equal(buf, fr[n - 1])

# the signal_processing entry points, one image per call (the reference wrapper's calling convention)
from librir_amd.signal_processing import BadPixels, filter_chain, gaussian_filter, translate  # noqa: E402

img = fr[0]
f32 = img.astype(np.float32)
bp = BadPixels(img)
m = 300
for name, fn in [("translate u16 (1.25, -2.5, nearest)", lambda: translate(img, 1.25, -2.5, "nearest")),
                 ("translate f32", lambda: translate(f32, 1.25, -2.5, "nearest")),
                 ("gaussian_filter f32 sigma 0.75", lambda: gaussian_filter(f32, 0.75)),
                 ("BadPixels.correct u16", lambda: bp.correct(img)),
                 ("filter_chain (the three in one call)", lambda: filter_chain(img, bp, 0.75, 1.25, -2.5, "nearest"))]:
    fn()
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(m):
            fn()
        dt = time.perf_counter() - t0
    print("%-43s: %7.0f calls/s (%5.1f us)" % (name, m / dt, dt / m * 1e6), flush=True)
t0 = time.perf_counter()
for i in range(m):
    x = bp.correct(fr[i])
    x = gaussian_filter(x.astype(np.float32), 0.75)
    x = translate(x, 1.25, -2.5, "nearest")
dt = time.perf_counter() - t0
print("configs[2] chain, three calls per image    : %7.0f frames/s (%5.1f us)   (the caller converts: x.astype(float32) is 110 us of numpy here)" % (m / dt, dt / m * 1e6), flush=True)
# round 6: the same three entry points with the uint16 image handed to gaussian_filter as it is (the reference's wrapper converts inside,
# rir_signal_processing.py:85-113; here the uint16 kernel takes it), then with the mirror's results in the library's page-locked memory, which
# the next call works on in place (opt-in: low_level.misc.results_in_page_locked_memory; tests/perf/three_call_probe.py has the breakdown)
from librir_amd.low_level.misc import results_in_page_locked_memory  # noqa: E402

for label, pinned in (("gaussian_filter converts", False), ("+ results in page-locked memory", True)):
    was = results_in_page_locked_memory(pinned)
    translate(gaussian_filter(bp.correct(fr[0]), 0.75), 1.25, -2.5, "nearest")
    t0 = time.perf_counter()
    for i in range(m):
        x = translate(gaussian_filter(bp.correct(fr[i]), 0.75), 1.25, -2.5, "nearest")
    dt = time.perf_counter() - t0
    results_in_page_locked_memory(was)
    print("configs[2] chain, three calls, %-31s: %7.0f frames/s (%5.1f us)" % (label, m / dt, dt / m * 1e6), flush=True)
t0 = time.perf_counter()
for i in range(m):
    x = filter_chain(fr[i], bp, 0.75, 1.25, -2.5, "nearest")
dt = time.perf_counter() - t0
print("configs[2] chain, one call per image       : %7.0f frames/s (%5.1f us)" % (m / dt, dt / m * 1e6), flush=True)
