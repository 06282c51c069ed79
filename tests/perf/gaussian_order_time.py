"""Cost of the reference-order gaussian (rir_set_gaussian_reference_order) beside the separable form: gaussian_filter and the fused chain on 256 frames
640x512 in HBM (development aid; results -> profiles/).   python tests/perf/gaussian_order_time.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.low_level.misc import _lib  # noqa: E402
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background  # noqa: E402

n, h, w = 256, 512, 640
arr = inject_bad_pixels(s1_noisy_background(n, h, w), 200)
x = torch.from_numpy(arr).cuda()
xf = x.float()
bp = D.BadPixels(x[0])
out = torch.empty_like(x)


def ms(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for mode in (0, 1):
    _lib.rir_set_gaussian_reference_order(mode)
    r = {}
    for s in (0.75, 1.0, 2.0):
        r["gaussian_filter_f32_sigma_%g_ms" % s] = ms(lambda: D.gaussian_filter(xf, s))
    r["gaussian_filter_u16_sigma_0.75_ms"] = ms(lambda: D.gaussian_filter(x, 0.75))
    r["filter_chain_sigma_0.75_ms"] = ms(lambda: D.filter_chain(x, bp, 0.75, (1.25, -2.5), "nearest", out=out))
    res["reference_order" if mode else "separable (default)"] = r
_lib.rir_set_gaussian_reference_order(0)
res["note"] = "%d frames %dx%d per call, HIP events, mean of 10" % (n, w, h)
print(json.dumps(res, indent=1))
