"""Development aid: is the encoder's slow mode a property of the frames buffer, of the workspace, or of the pair?
Several frames buffers x several workspaces, all alive at once, one process."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
def t_enc(ctx, t):
    for _ in range(2): ctx.encode_tiles(t)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    ev[0].record()
    for i in range(7):
        ctx.encode_tiles(t); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(7)])) * 1e3
frames = [torch.from_numpy(fr).cuda() for _ in range(5)]
ctxs = [D.CodecContext(w, h, n, 50) for _ in range(5)]
print("rows: frames buffers, columns: workspaces (us)")
print("%18s " % "" + " ".join("%#14x" % c.workspace.data_ptr() for c in ctxs))
for f in frames:
    print("%#18x " % f.data_ptr() + " ".join("%14.1f" % t_enc(c, f) for c in ctxs))
