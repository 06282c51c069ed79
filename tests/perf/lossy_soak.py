"""Development aid: many calls of the multi-stream bounded-loss step; prints calls that take far longer than the median (a wait
between workgroups that hit its clock shows as a 2 s call and as an error from the entry point)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

h, w = 512, 640
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = int(sys.argv[2]) if len(sys.argv) > 2 else 40
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
base = s1_noisy_background(m, h, w, seed=100)
ins = [torch.from_numpy(base + np.uint16(i)).cuda() for i in range(S)]
times = []
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        D.LossyStream.step_many(streams, ins, errors=True)
    except RuntimeError as e:
        print("call %d failed: %s" % (r, e))
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
med = float(np.median(times))
print("%d streams x %d frames: median call %.2f ms (%.0f fps aggregate), max %.2f ms" % (S, m, med * 1e3, S * m / med, max(times) * 1e3))
for r, t in enumerate(times):
    if t > 5 * med:
        print("  call %d took %.1f ms" % (r, t * 1e3))
