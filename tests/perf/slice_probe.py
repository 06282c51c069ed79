"""Where a slice of a movie spends its time (development aid, GPU box): load_pos into rows of a stack - fresh memory, fresh memory whose
pages are made by threads ahead of the reads (low_level.misc.touch_ahead: what IRMovie slices do), memory written before, one row over and
over - beside the plain one-image read."""
import gc
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.low_level.misc import touch_ahead  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402

n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
gc.collect()
gc.freeze()
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.h264")
    with IRSaver(p, w, h, h) as s:
        for i in range(n):
            s.add_image(fr[i], i * 1000)
    for rep in range(2):
        for name in ("fresh stack", "stack touched ahead", "stack written before", "one row over and over", "no out (recycled block)"):
            with IRMovie.from_filename(p) as mov:
                mov[0]
                stack = np.empty((n, h, w), np.uint16)
                if name == "stack written before":
                    stack[:] = 7
                t0 = time.perf_counter()
                ahead = touch_ahead(stack if name == "stack touched ahead" else np.empty(1))
                for i in range(n):
                    if name == "one row over and over":
                        mov.load_pos(i, 0, out=stack[0])
                    elif name.startswith("no out"):
                        mov.load_pos(i, 0)
                    else:
                        mov.load_pos(i, 0, out=stack[i])
                ahead.__exit__()
                dt = time.perf_counter() - t0
                print("%-26s %6.1f us a frame" % (name, dt / n * 1e6), flush=True)
                del stack
