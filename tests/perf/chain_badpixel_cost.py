"""Development aid: what the bad-pixel handling costs the fused filter chain - the chain on the same frames with 200, 20 and 0 flagged pixels
(0: no repair-table launch, no patches)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background  # noqa: E402

n, h, w = 256, 512, 640
offs = torch.tensor([1.25, -2.5], dtype=torch.float32).cuda()
clean = s1_noisy_background(n, h, w)
for nb in (200, 0, -1):
    arr = inject_bad_pixels(clean, nb) if nb > 0 else clean
    x = torch.from_numpy(arr).cuda()
    bp = D.BadPixels(x[0] if nb >= 0 else torch.full_like(x[0], 8000))  # (-1: detected on a flat image - an empty list)
    out = torch.empty_like(x)
    fn = lambda: D.filter_chain(x, bp, 0.75, offs, "nearest", out=out)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("%3d flagged pixels found (%d injected): %.4f ms per %d frames" % (bp.count, nb, best, n))
