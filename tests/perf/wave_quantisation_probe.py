"""Development aid: do the codec kernels' times step with the number of ROUNDS of waves (640 tiles x chunks waves, 8 192 places at 8 waves per
SIMD) - i.e. are they bound by what one wave can do rather than by HBM?  640x512, GOP 50, 10 .. 26 chunks."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402


def ev_ms(fn, reps=9):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)]))


w, h = 640, 512
base = s1_noisy_background(250, h, w)
big = torch.from_numpy(np.concatenate([base] * 6)).cuda()
for c in (10, 12, 13, 14, 16, 18, 20, 22, 24, 25, 26):
    n = 50 * c
    t = big[:n]
    ctx = D.CodecContext(w, h, n, 50)
    ctx.place_workspace(t)
    out = torch.empty_like(t)
    te = ev_ms(lambda: ctx.encode_tiles(t)) * 1e3
    td = ev_ms(lambda: ctx.decode_slots(out=out, check=False)) * 1e3
    print("%2d chunks = %5d waves (%.2f rounds of 8 192): encode %6.1f us = %.2f us per chunk | decode %6.1f us = %.2f us per chunk" %
          (c, 640 * c, 640 * c / 8192.0, te, te / c, td, td / c))
    del ctx, out
