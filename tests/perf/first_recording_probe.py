"""Where the first recording of a process loses its time (development aid): four recordings of 1 000 frames 640x512 through IRSaver.add_image, each
read back through IRMovie[i]; per recording the rate and the per-call times (ordinary call, chunk-closing call), with and without a warm-up recording
of 60 frames first.   python tests/perf/first_recording_probe.py [warmup frames]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
with tempfile.TemporaryDirectory() as d:
    if warm:
        with IRSaver(os.path.join(d, "warm.h264"), w, h, h) as s:
            for i in range(warm):
                s.add_image(fr[i], i)
        with IRMovie.from_filename(os.path.join(d, "warm.h264")) as mov:
            for i in range(warm):
                mov[i]
    for rep in range(4):
        dst = os.path.join(d, "r%d.h264" % rep)
        ts = []
        t0 = time.perf_counter()
        s = IRSaver(dst, w, h, h)
        if rep == 0 and os.environ.get("FIRST_DELAY"):  # (what a caller does between opening the file and its first frame; not part of the rate)
            time.sleep(float(os.environ["FIRST_DELAY"]))
            t0 = time.perf_counter()
        for i in range(n):
            t1 = time.perf_counter()
            s.add_image(fr[i], i * 1000)
            ts.append((time.perf_counter() - t1) * 1e6)
        t2 = time.perf_counter()
        s.close()
        te = time.perf_counter() - t0
        tclose = time.perf_counter() - t2
        ts = np.array(ts)
        flush = ts[49::50]
        plain = np.delete(ts, np.arange(49, len(ts), 50))
        tr = []
        t0 = time.perf_counter()
        mov = IRMovie.from_filename(dst)
        topen = time.perf_counter() - t0
        for i in range(n):
            t1 = time.perf_counter()
            mov[i]
            tr.append((time.perf_counter() - t1) * 1e6)
        mov.close()
        td = time.perf_counter() - t0
        tr = np.array(tr)
        os.remove(dst)
        print("recording %d: record %.0f fps (first call %.0f us, ordinary median %.1f / max %.0f us, chunk-closing median %.0f / max %.0f us, close %.1f ms) | "
              "read %.0f fps (open %.1f ms, first 3 calls %s us, median %.1f, max %.0f us) | round trip %.0f fps" %
              (rep, n / te, ts[0], np.median(plain[1:]), plain[1:].max(), np.median(flush), flush.max(), tclose * 1e3, n / td, topen * 1e3,
               [int(x) for x in tr[:3]], np.median(tr), tr.max(), n / (te + td)), flush=True)
