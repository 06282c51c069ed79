"""Development aid: distribution of the per-call time of IRSaver.add_image (640x512): the ordinary call vs the one that closes a chunk."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRSaver  # noqa: E402

n, h, w = 500, 512, 640
fr = s1_noisy_background(n, h, w)
with tempfile.TemporaryDirectory() as d:
    with IRSaver(os.path.join(d, "x.h264"), w, h, h) as s:
        s.add_image(fr[0], 0)
        ts = []
        for i in range(1, n):
            t0 = time.perf_counter()
            s.add_image(fr[i], i * 1000)
            ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
flush = ts[48::50]
plain = np.delete(ts, np.arange(48, len(ts), 50))
print("ordinary call: median %.1f us, p90 %.1f;  chunk-closing call: median %.0f us;  mean over all %.1f us" %
      (np.median(plain), np.percentile(plain, 90), np.median(flush), ts.mean()))
