"""Development aid: bounded-loss step rates on device-resident frames, single stream and S streams in shared launches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

h, w, n = 512, 640, 200
fr = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
st = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
st.step(fr[:60], errors=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    st.step(fr, errors=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("single stream: %.0f fps (%.1f us per frame)" % (5 * n / dt, dt / (5 * n) * 1e6))
for S in (4, 16, 32):
    m = 40
    streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
    ins = [torch.from_numpy(s1_noisy_background(m, h, w, seed=100 + i)).cuda() for i in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        D.LossyStream.step_many(streams, ins, errors=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%2d streams: %.0f fps aggregate (%.1f us per step of all streams)" % (S, 3 * m * S / dt, dt / (3 * m) * 1e6))
    for s in streams:
        s.close()
