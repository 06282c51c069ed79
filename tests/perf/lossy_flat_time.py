"""The constant-budget form of the bounded-loss step on a FLAT scene - the motion-corrected registration stream of configs[4], ~20 distinct
levels per frame, where the histogram pass takes its packed-window path (lossy_hist_add8) - beside the S1 recipe (levels spread over 250
bins) that tests/perf/lossy_const_time.py uses (development aid):
    python tests/perf/lossy_flat_time.py [frames per call] [streams]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background, s3_registration  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(sys.argv[2]) if len(sys.argv) > 2 else 7
h, w = 512, 640
f32, shifts = s3_registration(m, h, w)
t4 = torch.from_numpy(np.clip(f32, 0, 65535).astype(np.uint16)).cuda()
flat = D.remove_motion(t4, torch.from_numpy(shifts.astype(np.float32)).cuda(), rows=h - 3)
spread = torch.from_numpy(s1_noisy_background(m, h, w)).cuda()
print("levels per frame: flat %d, spread %d" % (len(torch.unique(flat[m // 2])), len(torch.unique(spread[m // 2]))))


def rate(fn, count, reps=5):
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = max(best, count / (time.perf_counter() - t0))
    return best


for label, fr in (("spread (S1)", spread), ("flat (motion-corrected S3)", flat)):
    streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
    ins = [fr.clone() for _ in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)
    r = rate(lambda: D.LossyStream.step_many(streams, ins, errors=False), m * S)
    streams[0].status()
    print("%-28s %d stream(s) x %d frames per call: %.0f frames/s   path %s" % (label, S, m, r, streams[0].path_stats()))
    for s_ in streams:
        s_.close()
