"""Whole-movie reads and writes through the Python classes (development aid, GPU box):
    python tests/perf/movie_bulk_time.py [frames]
IRMovie.data / IRMovie[a:b] / iteration and IRMovie.to_h264 on a 640x512 recording, beside the one-image calls they are made of."""
import gc
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
h, w = 512, 640
fr = s1_noisy_background(n, h, w)
gc.collect()
gc.freeze()


def rate(dt, k=n):
    return "%7.0f frames/s (%5.1f us a frame)" % (k / dt, dt / k * 1e6)


with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.h264")
    with IRSaver(p, w, h, h) as s:
        for i in range(n):
            s.add_image(fr[i], i * 1000)
    for rep in range(3):
        with IRMovie.from_filename(p) as mov:
            def timed(fn):
                t = time.perf_counter()
                r = fn()
                return r, time.perf_counter() - t

            def one_by_one():
                for i in range(n):
                    mov[i]

            _, t_one = timed(one_by_one)
            data, t_data = timed(lambda: mov.data)
            assert np.array_equal(data, fr)
            del data
            data, t_data2 = timed(lambda: mov.data)
            part, t_part = timed(lambda: mov[100:600])
            k, t_iter = timed(lambda: sum(1 for _ in mov))
            assert np.array_equal(data, fr) and np.array_equal(part, fr[100:600]) and k == n
            q = os.path.join(d, "copy%d.h264" % rep)
            _, t_copy = timed(lambda: mov.to_h264(q))
        print("IRMovie[i] one by one  : %s" % rate(t_one))
        print("IRMovie.data           : %s   again: %s" % (rate(t_data), rate(t_data2)))
        print("IRMovie[100:600]       : %s" % rate(t_part, 500))
        print("for image in IRMovie   : %s" % rate(t_iter))
        print("IRMovie.to_h264        : %s  (read + record)" % rate(t_copy), flush=True)
        with IRMovie.from_filename(q) as mov2:
            assert np.array_equal(mov2[n - 1], fr[n - 1])
