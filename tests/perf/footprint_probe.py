"""Development aid: is it the footprint of ONE launch that slows the decoder on large batches (3 000 frames of 640x512: 4.4 TB/s against
5.8 TB/s on 1 000), or the amount of memory in use?  Three contexts of 1 000 frames on three tensors, launched one after the other,
against one context of 3 000."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402


def ev_ms(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)]))


w, h = 640, 512
base = s1_noisy_background(250, h, w)
big = torch.from_numpy(np.concatenate([base] * 12)).cuda()  # 3 000 frames
parts = [big[i * 1000:(i + 1) * 1000] for i in range(3)]
ctxs = [D.CodecContext(w, h, 1000, 50) for _ in range(3)]
outs_big = torch.empty_like(big)
outs = [outs_big[i * 1000:(i + 1) * 1000] for i in range(3)]
for c, p in zip(ctxs, parts):
    c.place_workspace(p)
    c.encode_tiles(p)
one = D.CodecContext(w, h, 3000, 50)
one.place_workspace(big)
one.encode_tiles(big)
t3 = ev_ms(lambda: [c.decode_slots(out=o, check=False) for c, o in zip(ctxs, outs)])
t1 = ev_ms(lambda: one.decode_slots(out=outs_big, check=False))
e3 = ev_ms(lambda: [c.encode_tiles(p) for c, p in zip(ctxs, parts)])
e1 = ev_ms(lambda: one.encode_tiles(big))
print("decode 3 000 frames: one launch %.3f ms, three launches of 1 000 %.3f ms | encode: one %.3f ms, three %.3f ms" % (t1, t3, e1, e3))
