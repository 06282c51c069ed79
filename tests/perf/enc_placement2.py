"""Development aid: packing time of one frames allocation against a series of workspaces allocated further and further away."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
fr = torch.from_numpy(s1_noisy_background(n, h, w))


def t_pack(ctx, t, reps=9):
    ctx.encode_tiles(t)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        ctx.encode_tiles(t)
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


t = fr.cuda()
t2 = None
keep = []
step = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for k in range(24):
    c = D.CodecContext(w, h, n, 50)
    line = "ctx %2d workspace %x (frames %x): packing %.1f us" % (k, c.workspace.data_ptr(), t.data_ptr(), t_pack(c, t))
    if k == 12:
        t2 = fr.cuda()
    if t2 is not None:
        line += "   | second frames %x: %.1f us" % (t2.data_ptr(), t_pack(c, t2))
    print(line)
    keep.append(c)
    keep.append(torch.empty(step << 20, dtype=torch.uint8, device="cuda"))
