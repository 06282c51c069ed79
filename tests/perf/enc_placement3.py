"""Development aid: the whole step (packing, scan + gather, decode) of one frames allocation against a series of contexts and output buffers."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
fr = torch.from_numpy(s1_noisy_background(n, h, w))


def times(ctx, t, out, reps=9):
    enc = ctx.encode(t)
    ctx.decode(enc, out=out, check=False)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for k in range(reps):
        ev[k][0].record()
        ctx.encode_tiles(t)
        ev[k][1].record()
        enc = ctx.encode_compact()
        ev[k][2].record()
        ctx.decode(enc, out=out, check=False)
        ev[k][3].record()
    torch.cuda.synchronize()
    med = lambda a, b: float(np.median([ev[k][a].elapsed_time(ev[k][b]) for k in range(reps)])) * 1e3
    return med(0, 1), med(1, 2), med(2, 3)


t = fr.cuda()
keep = []
for k in range(10):
    c = D.CodecContext(w, h, n, 50)
    out = torch.empty_like(t)
    a, b, d = times(c, t, out)
    print("ctx %2d ws %x stream %x out %x: packing %.1f  gather %.1f  decode %.1f  step %.1f us" % (k, c.workspace.data_ptr(), c.stream.data_ptr(), out.data_ptr(), a, b, d, a + b + d))
    keep += [c, out]
