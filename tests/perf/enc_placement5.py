"""Development aid: the single-pass encoder against a series of contexts (stream + workspace placements)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
t = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
GB = float(1 << 30)


def timed(fn, reps=9):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


keep = []
for k in range(12):
    c = D.CodecContext(w, h, n, 50)
    print("ctx %2d stream %+7.2f GB workspace %+7.2f GB: single pass %.1f us   two pass %.1f us" %
          (k, (c.stream.data_ptr() - t.data_ptr()) / GB, (c.workspace.data_ptr() - t.data_ptr()) / GB, timed(lambda: c.encode(t, single_pass=True)),
           timed(lambda: c.encode(t))))
    keep += [c, torch.empty(900 << 20, dtype=torch.uint8, device="cuda")]
