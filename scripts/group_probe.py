"""Development aid: does processing the 1000-frame batch in groups (encode+decode per group, intermediate data small
enough for the Infinity Cache) beat one launch per kernel over the whole batch?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background
n, h, w = 1000, 512, 640
t = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
out = torch.empty_like(t)
for G in (1000, 500, 250, 200, 100):
    ctx = D.CodecContext(w, h, G, 50)
    def run():
        for g in range(0, n, G):
            enc = ctx.encode(t[g:g + G])
            ctx.decode(enc, out=out[g:g + G], check=False)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        run(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(10)]))
    ok = torch.equal(out.view(torch.int16), t.view(torch.int16))
    print("group %4d frames: %.1f us per 1000-frame pass  -> %.2f M fps  roundtrip %s" % (G, ms * 1e3, n / ms / 1e3, ok))
