"""Development aid: does the encoder's time depend on where its buffers sit?  One process, one frames tensor,
several workspaces at different (fresh) addresses; then one workspace, several frames tensors."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
def t_enc(ctx, t):
    for _ in range(3): ctx.encode_tiles(t)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        ctx.encode_tiles(t); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(10)])) * 1e3
t = torch.from_numpy(fr).cuda()
ctxs = []
for k in range(8):
    ctx = D.CodecContext(w, h, n, 50)
    ctxs.append(ctx)  # kept alive: every workspace is a different allocation
    print("frames@%#x ws@%#x hdr@%#x : encode %.1f us" % (t.data_ptr(), ctx.workspace.data_ptr(), ctx.hdr.data_ptr(), t_enc(ctx, t)))
print("--- same workspace (first), different frames tensors")
ts = []
for k in range(6):
    t2 = torch.from_numpy(fr).cuda()
    ts.append(t2)
    print("frames@%#x ws@%#x : encode %.1f us" % (t2.data_ptr(), ctxs[0].workspace.data_ptr(), t_enc(ctxs[0], t2)))
print("--- again the first pair: %.1f us" % t_enc(ctxs[0], t))
