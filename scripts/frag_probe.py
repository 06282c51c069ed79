"""Development aid: does the encoder's slow mode follow the way the frames buffer was allocated (its own allocation vs a piece of a
large one, 2 MiB-aligned or not)?  One process, one workspace."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
n, h, w = 1000, 512, 640
fr = torch.from_numpy(s1_noisy_background(n, h, w))
nb = fr.numel() * 2
def t_enc(ctx, t):
    for _ in range(2): ctx.encode_tiles(t)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    ev[0].record()
    for i in range(7):
        ctx.encode_tiles(t); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(7)])) * 1e3
ctx = D.CodecContext(w, h, n, 50)
own = [fr.cuda() for _ in range(4)]
for t in own:
    print("own allocation   %#16x  %7.1f us" % (t.data_ptr(), t_enc(ctx, t)))
big = torch.empty(8 * nb + (64 << 20), dtype=torch.uint8, device="cuda")
print("big allocation at %#x" % big.data_ptr())
for k, off in enumerate([0, nb, 2 * nb + (1 << 20), 3 * nb + (2 << 20) + 4096, 4 * nb + (3 << 20) + 65536, 5 * nb + (5 << 20) + 2048 * 640]):
    off = (off + 15) & ~15
    v = big[off:off + nb].view(torch.uint16).view(n, h, w)
    v.copy_(fr)
    print("piece at +%#12x (%% 2MiB = %#8x)  %7.1f us" % (off, (big.data_ptr() + off) % (2 << 20), t_enc(ctx, v)))
