"""Development aid (also under rocprofv3 --kernel-trace --stats): the bounded-loss step over 7 streams of 640x512, 200 frames per call, 4 calls."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

h, w, m, S = 512, 640, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 7
fr = torch.from_numpy(s1_noisy_background(m, h, w)).cuda()
streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
ins = [fr.clone() for _ in range(S)]
D.LossyStream.step_many(streams, ins, errors=False)
best = 0
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D.LossyStream.step_many(streams, ins, errors=False)
    torch.cuda.synchronize()
    best = max(best, m * S / (time.perf_counter() - t0))
print("%d streams x %d frames: %.0f k frames/s aggregate, status %s" % (S, m, best / 1e3, streams[0].status()))
