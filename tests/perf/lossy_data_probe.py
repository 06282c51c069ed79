"""Development aid: the bounded-loss step of 7 streams on two kinds of data - the S1 recipe (noisy background, levels spread over ~1 000
values) and the motion-corrected S3 registration stream of configs[4] (a flat scene: most pixels within a few levels of each other) - to see
which kernel the difference belongs to (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background, s3_registration  # noqa: E402

h, w, m, S = 512, 640, 200, 7
s1 = torch.from_numpy(s1_noisy_background(m, h, w)).cuda()
f32, shifts = s3_registration(m, h, w)
s3 = D.remove_motion(torch.from_numpy(np.clip(f32, 0, 65535).astype(np.uint16)).cuda(), torch.from_numpy(shifts.astype(np.float32)).cuda(), rows=h - 3)
for name, data in (("S1", s1), ("S3 motion-corrected", s3)):
    streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
    ins = [data.clone() for _ in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)
    best = 0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D.LossyStream.step_many(streams, ins, errors=False)
        torch.cuda.synchronize()
        best = max(best, m * S / (time.perf_counter() - t0))
    streams[0].status()
    print("%s: %d streams x %d frames: %.0f k frames/s; distinct values in frame 100: %d" % (name, S, m, best / 1e3, int(torch.unique(data[100]).numel())))
    for x in streams:
        x.close()
