"""The speculative form of the bounded-loss step (stdFactor 5: the reference's defaults) on 640x512 frames in HBM - a static scene (committed),
the S1 recipe (budgets move every frame: general form, back-off) - beside the constant-budget form (stdFactor 0) and the general form alone:
    python tests/perf/lossy_spec_time.py [frames per call] [streams] [only: a word of the line wanted, e.g. speculative,static]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = sys.argv[3].split(",") if len(sys.argv) > 3 else []
h, w = 512, 640


def static_scene(n, seed=5, sigma=0.7):
    g = torch.Generator(device="cuda").manual_seed(seed)
    bg = torch.rand((h, w), generator=g, device="cuda") * 1000 + 10
    return (bg[None] + sigma * torch.randn((n, h, w), generator=g, device="cuda")).to(torch.int32).to(torch.uint16)


def rate(fn, count, reps=6):
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = max(best, count / (time.perf_counter() - t0))
    return best


scenes = {"static": static_scene(m), "S1": torch.from_numpy(s1_noisy_background(m, h, w)).cuda()}
for label, scene, sf, env in (("constant budgets (stdFactor 0)", "static", 0.0, {}), ("speculative, static scene (6/2/5/32)", "static", 5.0, {}),
                              ("general form alone, static scene", "static", 5.0, {"RIR_LOSSY_NO_SPEC": "1"}),
                              ("speculative, S1 (budgets move)", "S1", 5.0, {}), ("general form alone, S1", "S1", 5.0, {"RIR_LOSSY_NO_SPEC": "1"})):
    if only and not all(o in label for o in only):
        continue
    for k in ("RIR_LOSSY_NO_SPEC",):
        os.environ.pop(k, None)
    os.environ.update(env)
    streams = [D.LossyStream(w, h, h - 3, 6, 2, sf, 32) for _ in range(S)]
    ins = [scenes[scene].clone() for _ in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)
    r = rate(lambda: D.LossyStream.step_many(streams, ins, errors=False), m * S)
    streams[0].status()
    print("%-40s %d stream(s) x %d frames per call: %9.0f frames/s   const %s  spec %s" % (label, S, m, r, streams[0].path_stats(), streams[0].spec_stats()), flush=True)
    for s_ in streams:
        s_.close()
