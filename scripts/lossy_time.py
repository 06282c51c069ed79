"""Development aid: time the per-frame bounded-loss recording path (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd.synthetic import s1_noisy_background
from librir_amd.video_io import IRSaver
n, h, w = 200, 512, 640
fr = s1_noisy_background(n, h, w)
with tempfile.TemporaryDirectory() as d:
    with IRSaver(os.path.join(d, "x.h264"), w, h, h - 3) as s:
        s.add_image_lossy(fr[0], 0)
        t0 = time.perf_counter()
        for i in range(1, n):
            s.add_image_lossy(fr[i], i * 1000)
        dt = time.perf_counter() - t0
print("lossy record: %.1f us per frame" % (dt / (n - 1) * 1e6))
