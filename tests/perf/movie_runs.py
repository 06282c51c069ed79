import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, "/root/repo")
from librir_amd.synthetic import s1_noisy_background
from librir_amd.video_io import IRMovie, IRSaver
n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "a.h264")
    with IRSaver(p, w, h, h) as s:
        for i in range(n):
            s.add_image(fr[i], i * 1000)
    for rep in range(6):
        ts = []
        t00 = time.perf_counter()
        with IRMovie.from_filename(p) as mov:
            t_open = time.perf_counter() - t00
            for i in range(n):
                t0 = time.perf_counter()
                img = mov[i]
                ts.append(time.perf_counter() - t0)
            t1 = time.perf_counter()
        t_close = time.perf_counter() - t1
        t = np.array(ts) * 1e6
        slow = np.argsort(t)[-6:][::-1]
        print("run %d: open %.1f ms, loop %.1f ms (mean %.1f us, median %.1f), close %.1f ms; slowest %s" % (rep, t_open * 1e3, t.sum() / 1e3, t.mean(), np.median(t), t_close * 1e3,
              ", ".join("%d:%.0f" % (i, t[i]) for i in slow)), flush=True)
