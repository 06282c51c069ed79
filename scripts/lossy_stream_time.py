"""Bounded-loss stream operator on device-resident frames: time per frame (300 frames 640x512, low = high = 3).
   python scripts/lossy_stream_time.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 300, 512, 640
t = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
for rep in range(3):
    ls = D.LossyStream(w, h, h - 3, 3, 3, 5.0, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, lo, hi = ls.step(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ls.close()
    print("lossy stream: %.1f us per frame (%.0f fps), errors %d..%d / %d..%d" % (dt / n * 1e6, n / dt, lo.min(), lo.max(), hi.min(), hi.max()))
