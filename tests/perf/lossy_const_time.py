"""The constant-budget form of the bounded-loss step against the general (resident) form, 640x512, frames in HBM (development aid):
    python tests/perf/lossy_const_time.py [frames per call] [streams]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
h, w = 512, 640
fr = torch.from_numpy(s1_noisy_background(m, h, w)).cuda()


def rate(fn, count, reps=5):
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = max(best, count / (time.perf_counter() - t0))
    return best


for label, env in (("general form ", "1"), ("constant form", None)):
    if env:  # (the general form alone: neither the constant-budget nor the speculative launches)
        os.environ["RIR_LOSSY_NO_CONST"] = os.environ["RIR_LOSSY_NO_SPEC"] = env
    else:
        os.environ.pop("RIR_LOSSY_NO_CONST", None)
        os.environ.pop("RIR_LOSSY_NO_SPEC", None)
    streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
    ins = [fr.clone() for _ in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)
    r = rate(lambda: D.LossyStream.step_many(streams, ins, errors=False), m * S)
    streams[0].status()
    print("%s  %d stream(s) x %d frames per call: %.0f frames/s   path %s" % (label, S, m, r, streams[0].path_stats()), flush=True)
    for s_ in streams:
        s_.close()
