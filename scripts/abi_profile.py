"""Development aid (GPU box): where a per-frame record / read call spends its time (cProfile + raw timings)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402
from librir_amd.video_io import rir_video_io as rv  # noqa: E402

n, h, w = 600, 512, 640
arr = s1_noisy_background(n, h, w)
buf = np.empty_like(arr[0])
t0 = time.perf_counter()
for i in range(n):
    np.copyto(buf, arr[i])
print("host memcpy of one frame: %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
t0 = time.perf_counter()
for i in range(n):
    np.empty((h, w), np.uint16)
print("np.empty: %.2f us;" % ((time.perf_counter() - t0) / n * 1e6), end=" ")
t0 = time.perf_counter()
for i in range(n):
    np.zeros((h, w), np.uint16)
print("np.zeros: %.2f us" % ((time.perf_counter() - t0) / n * 1e6))
d = tempfile.mkdtemp()
with IRSaver(os.path.join(d, "warm.h264"), w, h, h) as s:
    for i in range(60):
        s.add_image(arr[i], i)
p = os.path.join(d, "a.h264")


def record():
    with IRSaver(p, w, h, h) as s:
        for i in range(n):
            s.add_image(arr[i], i * 1000)


def read():
    with IRMovie.from_filename(p) as mov:
        for i in range(n):
            mov[i]


for fn in (record, read):
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print("%s: %.1f us per frame (%.0f fps)" % (fn.__name__, dt / n * 1e6, n / dt))
    pr = cProfile.Profile()
    pr.enable()
    fn()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
# bare C entry points
sv = rv.h264_open_file(os.path.join(d, "b.h264"), w, h, h)
t0 = time.perf_counter()
for i in range(n):
    rv._v.h264_add_image_lossless(sv, arr[i].ctypes.data, i, 0, None, None, None, None)
dt = time.perf_counter() - t0
rv.h264_close_file(sv)
print("bare h264_add_image_lossless: %.1f us per frame" % (dt / n * 1e6))
cam = rv.open_camera_file(os.path.join(d, "b.h264"))
t0 = time.perf_counter()
for i in range(n):
    rv._v.load_image(cam, i, 0, buf.ctypes.data)
dt = time.perf_counter() - t0
print("bare load_image: %.1f us per frame" % (dt / n * 1e6))
rv.close_camera(cam)
