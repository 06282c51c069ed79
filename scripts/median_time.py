"""3x3 median filter: check against torch's median on the interior and time (256 frames 640x512).  python scripts/median_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 256, 512, 640
t = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
out = D.median_filter(t)
win = t[:8].to(torch.int32).unfold(1, 3, 1).unfold(2, 3, 1).reshape(8, h - 2, w - 2, 9)
ref = win.sort(dim=-1).values[..., 4]
print("interior equals sort-based median:", bool(torch.equal(out[:8, 1:-1, 1:-1].to(torch.int32), ref)))
for _ in range(3):
    D.median_filter(t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    D.median_filter(t)
e1.record()
torch.cuda.synchronize()
print("median_filter 3x3  %.3f ms per %d frames" % (e0.elapsed_time(e1) / 20, n))
