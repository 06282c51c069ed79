"""Development aid: per-call time of the reference-style (host pointer, one frame) signal_processing entry points."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd.signal_processing import rir_signal_processing as sp
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background
fr = inject_bad_pixels(s1_noisy_background(4, 512, 640), 200)
f32 = fr[0].astype(np.float32)
def t(fn, n=200):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
h = sp.bad_pixels_create(fr[0])
print("translate u16      %.1f us" % t(lambda: sp.translate(fr[0], 1.25, -2.5, "nearest")))
print("translate f32      %.1f us" % t(lambda: sp.translate(f32, 1.25, -2.5, "nearest")))
print("gaussian f32       %.1f us" % t(lambda: sp.gaussian_filter(f32, 0.75)))
print("bad_pixels_correct %.1f us" % t(lambda: sp.bad_pixels_correct(h, fr[1])))
print("find_median_pixel  %.1f us" % t(lambda: sp.find_median_pixel(fr[0], 0.5)))
print("astype f32         %.1f us" % t(lambda: fr[0].astype(np.float32)))
print("np.copy u16        %.1f us" % t(lambda: fr[0].copy()))
