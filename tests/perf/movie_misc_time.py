"""The other whole-movie operations of the Python classes (development aid, GPU box): from_numpy_array, to_h264 of a raw movie, tis,
timestamps, frames_attributes, a BadPixels pass over a movie - 1 000 x 640x512."""
import gc
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402

n, h, w = 1000, 512, 640
fr = s1_noisy_background(n, h, w)
gc.collect()
gc.freeze()


def timed(name, fn, k=n):
    t = time.perf_counter()
    r = fn()
    dt = time.perf_counter() - t
    print("%-44s %8.1f ms  (%6.1f us a frame)" % (name, dt * 1e3, dt / k * 1e6), flush=True)
    return r


with tempfile.TemporaryDirectory() as d:
    os.environ["LIBRIR_TEMP_FOLDER"] = d
    for rep in range(2):
        raw = timed("IRMovie.from_numpy_array", lambda: IRMovie.from_numpy_array(fr))
        timed("  .data of the raw movie", lambda: raw.data)
        timed("  .timestamps", lambda: raw.timestamps)
        p = os.path.join(d, "r%d.h264" % rep)
        timed("  .to_h264 (pcr2h264)", lambda: raw.to_h264(p))
        raw.close()
        with IRMovie.from_filename(p) as mov:
            timed("IRMovie.tis", lambda: mov.tis)
            timed("IRMovie.timestamps", lambda: mov.timestamps)
            timed("IRMovie.frames_attributes", lambda: mov.frames_attributes)
            timed("IRMovie.from_filename + close", lambda: IRMovie.from_filename(p).close(), 1)
        timed("IRMovie.from_bytes (the file's bytes)", lambda: IRMovie.from_bytes(open(p, "rb").read()).close(), 1)
