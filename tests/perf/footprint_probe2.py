"""Development aid, second question: is the slower decode on large batches a property of the ALLOCATION the output lives in (its size, its
placement class) rather than of the bytes a launch touches?  1 000 frames decoded into (a) a tensor of their own, (b) the first third of a
3 000-frame tensor, (c)-(j) eight more tensors of 1 000 frames allocated one after the other."""
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


w, h, n = 640, 512, 1000
t = torch.from_numpy(np.concatenate([s1_noisy_background(250, h, w)] * 4)).cuda()
ctx = D.CodecContext(w, h, n, 50)
ctx.place_workspace(t)
ctx.encode_tiles(t)
own = torch.empty_like(t)
big = torch.empty((3 * n, h, w), dtype=torch.uint16, device="cuda")
print("own tensor: %.1f us | first / second / third third of a 3 000-frame tensor: %s us" %
      (ev_ms(lambda: ctx.decode_slots(out=own, check=False)) * 1e3,
       [round(ev_ms(lambda: ctx.decode_slots(out=big[i * n:(i + 1) * n], check=False)) * 1e3, 1) for i in range(3)]))
more = [torch.empty_like(t) for _ in range(8)]
print("eight more tensors of 1 000 frames:", [round(ev_ms(lambda: ctx.decode_slots(out=o, check=False)) * 1e3, 1) for o in more])
spacers = [torch.empty(3 << 30, dtype=torch.uint8, device="cuda") for _ in range(6)]
far = [torch.empty_like(t) for _ in range(4)]
print("four more behind 18 GiB of spacers:", [round(ev_ms(lambda: ctx.decode_slots(out=o, check=False)) * 1e3, 1) for o in far])
