"""rir_transcode_images alone (development aid, GPU box): 1 000 images 640x512 of a recording of this library into a new saver without leaving
the device; RIR_TRANSCODE_DIAG=1 prints where the time went (chunks into device memory / into the saver)."""
import gc, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from librir_amd.synthetic import s1_noisy_background
from librir_amd.video_io import IRMovie, IRSaver
from librir_amd.video_io import rir_video_io as rv
n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
gc.collect(); gc.freeze()
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.h264")
    with IRSaver(p, w, h, h) as s:
        for i in range(n): s.add_image(fr[i], i * 1000)
    for rep in range(3):
        with IRMovie.from_filename(p) as mov:
            q = os.path.join(d, "t%d.h264" % rep)
            s = IRSaver(q, w, h, h)
            t0 = time.perf_counter()
            ok = rv.transcode_images(mov.handle, s.handle, 0, n, np.arange(n) * 1000)
            t1 = time.perf_counter()
            s.close()
            t2 = time.perf_counter()
            print("transcode call %.1f ms (%.1f us a frame), close %.1f ms, ok %s" % ((t1 - t0) * 1e3, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3, ok), flush=True)
        with IRMovie.from_filename(q) as chk:
            assert np.array_equal(chk[n - 1], fr[n - 1]) and np.array_equal(chk[0], fr[0])
