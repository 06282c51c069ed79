import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import s1_noisy_background
from librir_amd.video_io import rir_video_io as rv
n, h, w = 200, 512, 640
fr = s1_noisy_background(n, h, w)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "z.bin")
    t0 = time.perf_counter()
    hd = rv.open_video_write(p, w, h, 50, 1, 0)
    for i in range(n): rv.image_write(hd, fr[i], i * 1000)
    rv.close_video(hd)
    t1 = time.perf_counter()
    print("ZFile write: %.0f us a frame, file %.1f MB (raw %.1f)" % ((t1 - t0) / n * 1e6, os.path.getsize(p) / 1e6, fr.nbytes / 1e6))
    cam = rv.open_camera_file(p)
    out = np.empty((h, w), np.uint16)
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(n): rv.load_image(cam, i, 0, (h, w), out)
        print("ZFile read : %.0f us a frame" % ((time.perf_counter() - t0) / n * 1e6))
    assert np.array_equal(out, fr[n - 1])
    rv.close_camera(cam)
