"""Development aid: the step (encode_tiles + decode_slots) on several geometries and lengths - bytes per second of 4WH + 2C, to see what the
frame size (the stride between the 1 KiB tiles a wave reads) and the total footprint do to the rate."""
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


for (w, h, n) in ((640, 512, 1000), (640, 512, 3000), (1024, 768, 400), (1024, 768, 1250), (1280, 1024, 800), (320, 256, 4000)):
    base = s1_noisy_background(min(n, 250), h, w)
    t = torch.from_numpy(np.concatenate([base] * (-(-n // base.shape[0])))[:n]).cuda()
    ctx = D.CodecContext(w, h, n, 50)
    ctx.place_workspace(t)
    out = torch.empty_like(t)
    te = ev_ms(lambda: ctx.encode_tiles(t))
    td = ev_ms(lambda: ctx.decode_slots(out=out, check=False))
    ts = ev_ms(lambda: (ctx.encode_tiles(t), ctx.decode_slots(out=out, check=False)))
    c = ctx.slots_payload_bytes()
    raw = 2.0 * w * h * n
    print("%4dx%-4d x %4d (%.2f GB raw): encode %.3f ms (%.2f TB/s)  decode %.3f ms (%.2f TB/s)  step %.3f ms = %.2f TB/s of 4WH+2C, %.2f M frames/s" %
          (w, h, n, raw / 1e9, te, (raw + c) / te / 1e9, td, (raw + c) / td / 1e9, ts, (2 * raw + 2 * c) / ts / 1e9, n / ts / 1e3))
    del ctx, t, out
    torch.cuda.empty_cache()
