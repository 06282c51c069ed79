"""Development aid: the packing kernel with its workspace in the SAME placement class as the frames and in a DIFFERENT one
(classes found by timing the kernel itself on a set of 2 GiB allocations, tests/perf/placement_probe2.py), for the library as
built - run it once per build variant (scripts/class_variants.sh).  Prints one line."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
src = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
ctx = D.CodecContext(w, h, n, 50)
WS = ctx.layout.workspace_bytes
bufs = [torch.empty(2 << 30, dtype=torch.uint8, device="cuda") for _ in range(8)]
out = torch.empty_like(src)


def timed(fn, reps=9):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


ts = []
for b in bufs:
    ctx.workspace = b[:WS]
    ts.append(timed(lambda: ctx.encode_tiles(src)))
lo, hi = min(ts), max(ts)
fast = [i for i, t in enumerate(ts) if t < lo + 0.35 * (hi - lo)]
slow = [i for i, t in enumerate(ts) if t > hi - 0.35 * (hi - lo)]
res = {}
for name, idx in (("other_class", fast[0]), ("same_class", slow[0] if hi - lo > 5 else fast[-1])):
    ctx.workspace = bufs[idx][:WS]
    res[name] = (timed(lambda: ctx.encode_tiles(src)), timed(lambda: ctx.decode_slots(out=out, check=False)),
                 timed(lambda: (ctx.encode_tiles(src), ctx.decode_slots(out=out, check=False))))
ok = torch.equal(out.view(torch.int16), src.view(torch.int16))
print("pack over 8 workspaces:", [round(t, 1) for t in ts], "| other class: pack %.1f decode %.1f step %.1f | same class: pack %.1f decode %.1f step %.1f | exact %s" %
      (res["other_class"] + res["same_class"] + (ok,)))
