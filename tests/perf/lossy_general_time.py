"""The general (resident) form of the bounded-loss run - stdFactor 5, budgets that follow the frames' statistics - one stream and nine,
checked against the oracle on the first frames (development aid):    python tests/perf/lossy_general_time.py [frames per call]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from oracle.pyoracle import Oracle, OracleLossy  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 400
h, w = 512, 640
arr = s1_noisy_background(m, h, w)
fr = torch.from_numpy(arr).cuda()
# parity on the first 24 frames
O = Oracle()
L = OracleLossy(O, w, h, h - 3, low_err=6, high_err=2, std_factor=5.0, running_average=8)
exp = np.stack([L.step(arr[i]) for i in range(24)])
ls = D.LossyStream(w, h, h - 3, 6, 2, 5.0, 8)
got = ls.step(fr[:24])[0].cpu().numpy()
ls.close()
print("general form == oracle on 24 frames:", bool(np.array_equal(got, exp)))


def rate(fn, count, reps=4):
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = max(best, count / (time.perf_counter() - t0))
    return best


for S in (1, 9):
    streams = [D.LossyStream(w, h, h - 3, 3, 3, 5.0, 32) for _ in range(S)]
    ins = [fr.clone() for _ in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)
    r = rate(lambda: D.LossyStream.step_many(streams, ins, errors=False), m * S)
    streams[0].status()
    print("general form %d stream(s) x %d frames per call: %.0f frames/s" % (S, m, r))
    for s_ in streams:
        s_.close()
