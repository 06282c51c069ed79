"""Development aid: does the packing kernel's time depend on where the raw frames (or the workspace) sit?  Times
rir_codec_encode_tiles_device with the frames at several offsets inside one larger allocation, and with fresh allocations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
fr = torch.from_numpy(s1_noisy_background(n, h, w))
npx = n * h * w


def t_pack(ctx, t, reps=15):
    ctx.encode_tiles(t)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        ctx.encode_tiles(t)
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


big = torch.empty(npx + (64 << 20), dtype=torch.uint16, device="cuda")
ctx = D.CodecContext(w, h, n, 50)
for off in (0, 8, 2048, 32768, 1 << 20, (1 << 20) + 2048, 4 << 20, (16 << 20) + 64):  # offsets in uint16 elements (x2 bytes)
    t = big[off:off + npx].view(n, h, w)
    t.copy_(fr)
    print("frames at +%9d B of one allocation: packing %.1f us" % (2 * off, t_pack(ctx, t)))
keep = []
for k in range(4):
    t = fr.cuda()
    ctx2 = D.CodecContext(w, h, n, 50)
    print("fresh allocations %d (frames %x, workspace %x): packing %.1f us" % (k, t.data_ptr(), ctx2.workspace.data_ptr(), t_pack(ctx2, t)))
    keep.append((t, ctx2, torch.empty(37 << 20, dtype=torch.uint8, device="cuda")))  # (moves the next allocations)
print("matrix: rows = frames allocations, columns = contexts (workspace + tables)")
frames_list, ctx_list, pad = [], [], []
for k in range(5):
    frames_list.append(fr.cuda())
    pad.append(torch.empty((11 + 7 * k) << 20, dtype=torch.uint8, device="cuda"))
    ctx_list.append(D.CodecContext(w, h, n, 50))
    pad.append(torch.empty((5 + 3 * k) << 20, dtype=torch.uint8, device="cuda"))
for i, t in enumerate(frames_list):
    print("frames %x: " % t.data_ptr() + "  ".join("%.1f" % t_pack(c, t, 9) for c in ctx_list))
print("contexts: " + "  ".join("%x" % c.workspace.data_ptr() for c in ctx_list))
