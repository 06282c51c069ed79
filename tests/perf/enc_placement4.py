"""Development aid: bench.py's allocation sequence, then candidate workspaces with their distances from the frames."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
frames = torch.from_numpy(s1_noisy_background(n, h, w)).to("cuda")
ctx = D.CodecContext(w, h, n, 50)
out = torch.empty_like(frames)
GB = float(1 << 30)


def t_pack(ws, reps=7):
    old, ctx.workspace = ctx.workspace, ws
    ctx.encode_tiles(frames)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        ctx.encode_tiles(frames)
        e[i + 1].record()
    torch.cuda.synchronize()
    ctx.workspace = old
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


print("frames %x  hdr %+.2f GB  stream %+.2f GB  out %+.2f GB" % (frames.data_ptr(), (ctx.hdr.data_ptr() - frames.data_ptr()) / GB,
                                                                (ctx.stream.data_ptr() - frames.data_ptr()) / GB, (out.data_ptr() - frames.data_ptr()) / GB))
keep = []
ws = ctx.workspace
for k in range(16):
    print("workspace at %+7.2f GB from the frames: packing %.1f us" % ((ws.data_ptr() - frames.data_ptr()) / GB, t_pack(ws)))
    keep.append(ws)
    keep.append(torch.empty((int(sys.argv[1]) if len(sys.argv) > 1 else 1500) << 20, dtype=torch.uint8, device="cuda"))
    ws = torch.empty((ctx.layout.workspace_bytes,), dtype=torch.uint8, device="cuda")
