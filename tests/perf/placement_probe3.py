"""Development aid: how many placement classes are there?  N separate 2 GiB allocations; the frames go into a buffer that has no
class yet, the packing kernel is timed with its workspace in every buffer, the slow ones are that buffer's class; repeat until
every buffer has a class.  Prints the classes in address order.   python tests/perf/placement_probe3.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
src = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
ctx = D.CodecContext(w, h, n, 50)
WS = ctx.layout.workspace_bytes
FB = src.numel() * 2
bufs = [torch.empty(2 << 30, dtype=torch.uint8, device="cuda") for _ in range(N)]


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


cls = [-1] * N
rows = []
c = 0
while -1 in cls:
    i = cls.index(-1)
    fr = bufs[i][:FB].view(torch.uint16).view(n, h, w)
    fr.copy_(src)
    ts = []
    for j in range(N):
        ctx.workspace = bufs[j][FB + 4096:FB + 4096 + WS] if i == j else bufs[j][:WS]
        ts.append(timed(lambda: ctx.encode_tiles(fr)))
    lo, hi = min(ts), max(ts)
    thr = (lo + hi) / 2
    members = [j for j in range(N) if ts[j] > thr] if hi - lo > 6 else list(range(N))
    clash = [j for j in members if cls[j] not in (-1,)]
    for j in members:
        if cls[j] == -1:
            cls[j] = c
    rows.append((i, c, [round(t, 1) for t in ts], clash))
    print("frames in buffer %d -> class %d: slow with %s%s" % (i, c, members, ("  (already classed: %s)" % clash) if clash else ""), flush=True)
    c += 1
order = np.argsort([b.data_ptr() for b in bufs])
print("classes in address order (GiB: class):", " ".join("%.0f:%d" % (bufs[k].data_ptr() / (1 << 30) - bufs[order[0]].data_ptr() / (1 << 30), cls[k]) for k in order))
print("number of classes over %d GiB: %d" % (2 * N, c))
