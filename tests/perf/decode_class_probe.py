"""Development aid: the decoder on a batch whose compressed stream does NOT fit the Infinity Cache (3 000 frames of 640x512, 400 MB of
stream): is what it loses against the 1 000-frame case (4.4 against 5.8 TB/s) the placement class of the output beside the stream -
eight output tensors allocated one after the other, then one placed by the library beside the workspace - or latency?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402


def ev_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)]))


w, h, n = 640, 512, 3000
t = torch.from_numpy(np.concatenate([s1_noisy_background(250, h, w)] * 12)).cuda()
ctx = D.CodecContext(w, h, n, 50)
ctx.place_workspace(t)
ctx.encode_tiles(t)
c = ctx.slots_payload_bytes()
raw = 2.0 * w * h * n
outs = [torch.empty_like(t) for _ in range(6)]
ts = [ev_ms(lambda: ctx.decode_slots(out=o, check=False)) for o in outs]
print("decode of 3 000 frames into six tensors allocated in a row: %s us  (%.2f .. %.2f TB/s)" % ([round(x * 1e3) for x in ts], (raw + c) / max(ts) / 1e9, (raw + c) / min(ts) / 1e9))
placed, times = D.empty_beside(ctx.workspace, tuple(t.shape), torch.uint16)
tp = ev_ms(lambda: ctx.decode_slots(out=placed, check=False))
print("into a tensor the library placed beside the workspace: %.0f us (%.2f TB/s); its candidates' probe times %s" % (tp * 1e3, (raw + c) / tp / 1e9, [round(x, 1) for x in times]))
