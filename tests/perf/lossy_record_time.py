"""Development aid: rate of bounded-loss recording through the per-frame ABI (IRSaver.add_image_lossy), 640x512."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRSaver  # noqa: E402

n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 500, 512, 640
fr = s1_noisy_background(n, h, w)
with tempfile.TemporaryDirectory() as d:
    for rep in range(3):
        dst = os.path.join(d, "lossy%d.h264" % rep)
        t0 = time.perf_counter()
        with IRSaver(dst, w, h, h - 3) as s:
            s.set_parameter("lowValueError", 3)
            s.set_parameter("highValueError", 3)
            s.set_parameter("stdFactor", 0)
            for i in range(n):
                s.add_image_lossy(fr[i], i * 1000)
        dt = time.perf_counter() - t0
        print("bounded-loss recording: %.0f frames/s (%.1f us per frame), file ratio %.2f" % (n / dt, dt / n * 1e6, fr.nbytes / os.path.getsize(dst)))
