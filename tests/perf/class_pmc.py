"""Development aid, run under rocprofv3 --pmc (scripts/class_pmc.sh): finds a workspace in ANOTHER placement class than the
frames and one in the SAME class (by timing the packing kernel on 2 GiB allocations, as tests/perf/class_probe.py), then
launches the packing kernel 6 times on each, the two groups separated by one decode launch as a marker in the dispatch list."""
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


def timed(fn, reps=5):
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
other, same = int(np.argmin(ts)), int(np.argmax(ts))
print("calibration (us):", [round(t, 1) for t in ts], "-> other class: buffer %d, same class: buffer %d" % (other, same), flush=True)
for idx in (other, same):
    ctx.workspace = bufs[idx][:WS]
    ctx.encode_tiles(src)
    ctx.decode_slots(out=out, check=False)  # marker
    torch.cuda.synchronize()
    for _ in range(6):
        ctx.encode_tiles(src)
    torch.cuda.synchronize()
ctx.decode_slots(out=out, check=False)  # closing marker
torch.cuda.synchronize()
print("spread of the calibration: %.1f us (needs > 5 for the two groups to differ)" % (hi - lo))
