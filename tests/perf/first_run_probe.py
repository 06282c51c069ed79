"""The first full-size recording + read of a process against the following ones, phase by phase (development aid, GPU box): bench.py's
sequence - a 60-frame warm-up recording and read, then four 1 000-frame recordings each read back."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402

n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
pc = time.perf_counter
with tempfile.TemporaryDirectory() as d:
    t0 = pc()
    with IRSaver(os.path.join(d, "warm.h264"), w, h, h) as s:
        for i in range(60):
            s.add_image(fr[i], i)
    t1 = pc()
    with IRMovie.from_filename(os.path.join(d, "warm.h264")) as mov:
        for i in range(60):
            mov[i]
    print("warm-up: record 60 frames %.1f ms, read %.1f ms" % ((t1 - t0) * 1e3, (pc() - t1) * 1e3), flush=True)
    for rep in range(4):
        dst = os.path.join(d, "abi%d.h264" % rep)
        t0 = pc()
        s = IRSaver(dst, w, h, h)
        t_open = pc()
        for i in range(n):
            s.add_image(fr[i], i * 1000)
        t_loop = pc()
        s.close()
        t_close = pc()
        mov = IRMovie.from_filename(dst)
        t_mopen = pc()
        calls = []
        for i in range(n):
            c0 = pc()
            img = mov[i]
            calls.append(pc() - c0)
        t_read = pc()
        worst = sorted(range(n), key=lambda k: -calls[k])[:6]
        print("       slowest reads: " + ", ".join("%d: %.1f ms" % (k, calls[k] * 1e3) for k in worst), flush=True)
        mov.close()
        t_end = pc()
        os.remove(dst)
        print("run %d: saver open %.1f  add loop %.1f  close %.1f | movie open %.1f  read loop %.1f  close %.1f ms -> round trip %.0f frames/s" %
              (rep, (t_open - t0) * 1e3, (t_loop - t_open) * 1e3, (t_close - t_loop) * 1e3, (t_mopen - t_close) * 1e3, (t_read - t_mopen) * 1e3,
               (t_end - t_read) * 1e3, n / (t_end - t0)), flush=True)
