"""Reading a movie with the read-back filters switched on (development aid, GPU box): IRMovie[i] over 500 images 640x512 with bad-pixel
repair, with motion correction, with both - beside the plain read."""
import gc
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402

n, h, w = 500, 512, 640
fr = inject_bad_pixels(s1_noisy_background(n, h, w), 40)
gc.collect()
gc.freeze()
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.h264")
    with IRSaver(p, w, h, h) as s:
        for i in range(n):
            s.add_image(fr[i], i * 1000)
    reg = os.path.join(d, "reg.tsv")
    with open(reg, "w") as f:
        f.write("\tx-axis translations\ty-axis translations\tConfidence level\n")
        for i in range(n):
            f.write("%d\t%r\t%r\t0.9\n" % (i, 0.25 * (i % 7), -0.5 * (i % 3)))
    for name, bp, mc in (("plain", False, False), ("bad-pixel repair", True, False), ("motion correction", False, True), ("both", True, True)):
        with IRMovie.from_filename(p) as mov:
            mov.bad_pixels_correction = bp
            if mc:
                mov.registration_file = reg
                mov.registration = True
            for rep in range(2):
                t0 = time.perf_counter()
                for i in range(n):
                    mov[i]
                dt = time.perf_counter() - t0
            print("%-18s %6.1f us a frame" % (name, dt / n * 1e6), flush=True)
