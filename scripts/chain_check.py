"""Fused filter chain against the three-kernel chain (bit-exact) on a few geometries / shifts, then timing.
Run on a GPU box: python scripts/chain_check.py"""
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background  # noqa: E402


def unfused(x, bp, sigma, offs, strat, back=0):
    a = bp.correct(x) if bp is not None else x
    g = D.gaussian_filter(a, sigma)
    return D.translate_to_u16(g, offs, strat, background=back)


bad = 0
cases = [((6, 512, 640), 0.75, (1.25, -2.5), "nearest"), ((3, 67, 83), 0.75, (-3.5, 4.75), "nearest"), ((3, 67, 83), 1.0, (0.5, 0.5), "background"),
         ((2, 20, 30), 2.0, (0.0, 0.0), "nearest"), ((2, 130, 61), 0.3, (100.0, -200.0), "nearest"), ((2, 3, 5), 0.75, (0.25, 0.75), "nearest"),
         ((4, 240, 320), 1.49, (-0.99999994, 7.0000005), "background"), ((2, 100, 700), 0.75, (650.5, 0.0), "nearest")]
for shape, sigma, off, strat in ([] if os.environ.get("RIR_CHAIN_TIME_ONLY") else cases):
    n, h, w = shape
    arr = inject_bad_pixels(s1_noisy_background(n, h, w), min(200, h * w // 20))
    x = torch.from_numpy(arr).cuda()
    for use_bp in (True, False):
        bp = D.BadPixels(x[0]) if use_bp else None
        ref = unfused(x, bp, sigma, off, strat, 7)
        out = D.filter_chain(x, bp, sigma, off, strat, background=7)
        torch.cuda.synchronize()
        d = (out.view(torch.int16) != ref.view(torch.int16))
        nd = int(d.sum())
        print(shape, sigma, off, strat, "bp" if use_bp else "--", "flagged", bp.count if bp else 0, "diff", nd)
        if nd:
            idx = d.nonzero()[:5].cpu().numpy()
            print("   first diffs", idx.tolist(), out[d][:5].cpu().numpy().view(np.uint16), ref[d][:5].cpu().numpy().view(np.uint16))
        bad += nd
    # per-frame offsets
    offs = torch.tensor(np.random.default_rng(1).uniform(-5, 5, (n, 2)), dtype=torch.float32).cuda()
    bp = D.BadPixels(x[0])
    nd = int((D.filter_chain(x, bp, sigma, offs, strat, 7).view(torch.int16) != unfused(x, bp, sigma, offs, strat, 7).view(torch.int16)).sum())
    print(shape, "per-frame offsets diff", nd)
    bad += nd
print("TOTAL DIFF", bad)

n, h, w = 256, 512, 640
arr = inject_bad_pixels(s1_noisy_background(n, h, w), 200)
x = torch.from_numpy(arr).cuda()
bp = D.BadPixels(x[0])
offs = torch.tensor([1.25, -2.5], dtype=torch.float32).cuda()
for name, fn in (("three kernels", lambda: unfused(x, bp, 0.75, offs, "nearest")), ("fused", lambda: D.filter_chain(x, bp, 0.75, offs, "nearest"))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-14s %.3f ms per %d frames  (%.2f TB/s of the 4 B/px algorithmic traffic)" % (name, ms, n, n * h * w * 4 / ms / 1e9))
sys.exit(1 if bad else 0)
