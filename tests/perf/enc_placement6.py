"""Development aid: after the workspace has been placed, how much do gather + decode depend on where the stream buffer sits?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
t = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
ctx = D.CodecContext(w, h, n, 50)
out = torch.empty_like(t)
print("packing candidates:", [round(x, 1) for x in ctx.place_workspace(t)])
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


ctx.encode_tiles(t)
keep = []
for k in range(12):
    enc = ctx.encode_compact()
    tg = timed(lambda: ctx.encode_compact())
    td = timed(lambda: ctx.decode(enc, out=out, check=False))
    print("stream at %+7.2f GB from the workspace, %+7.2f GB from out: gather %.1f  decode %.1f  sum %.1f us" %
          ((ctx.stream.data_ptr() - ctx.workspace.data_ptr()) / GB, (ctx.stream.data_ptr() - out.data_ptr()) / GB, tg, td, tg + td))
    keep += [ctx.stream, torch.empty(1200 << 20, dtype=torch.uint8, device="cuda")]
    ctx.stream = torch.empty_like(ctx.stream)
