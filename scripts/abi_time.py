"""Development aid: per-frame C-ABI record / read timing at two frame sizes (host overhead vs bytes)."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd.synthetic import s1_noisy_background
from librir_amd.video_io import IRSaver, IRMovie
from librir_amd.video_io import rir_video_io as rv
for (h, w, n) in [(16, 32, 400), (512, 640, 400)]:
    fr = s1_noisy_background(n, h, w)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.h264")
        with IRSaver(p, w, h, h) as s:
            s.add_image(fr[0], 0)
            t0 = time.perf_counter()
            for i in range(1, n):
                s.add_image(fr[i], i * 1000)
            te = time.perf_counter() - t0
        cam = rv.open_camera_file(p)
        rv.load_image(cam, 0)
        t0 = time.perf_counter()
        for i in range(1, n):
            img = rv.load_image(cam, i)
        td = time.perf_counter() - t0
        assert np.array_equal(img, fr[n - 1])
        rv.close_camera(cam)
    print("%dx%d: record %.1f us/frame, read %.1f us/frame" % (w, h, te / (n - 1) * 1e6, td / (n - 1) * 1e6))
