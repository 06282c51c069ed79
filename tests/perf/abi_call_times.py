"""Per-call times of the per-frame C ABI (development aid, GPU box): which calls of a recording / a sequential read are the slow ones.
    python tests/perf/abi_call_times.py [frames]"""
import ctypes as ct
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.low_level.misc import _video_io as V  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
import librir_amd.video_io  # noqa: E402,F401  (sets the argument types of the C entry points)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
h, w = 512, 640
fr = s1_noisy_background(n, h, w)
pc = time.perf_counter


def summary(name, t, period_list=(5, 50)):
    t = np.asarray(t) * 1e6
    print("%s: total %.1f ms, mean %.1f us, median %.1f, p90 %.1f, max %.1f" % (name, t.sum() / 1e3, t.mean(), np.median(t), np.percentile(t, 90), t.max()))
    for p in period_list:
        by = [t[k::p].mean() for k in range(p)]
        worst = int(np.argmax(by))
        print("   by call index mod %d: slowest residue %d at %.1f us, the others %.1f us" % (p, worst, by[worst], (t.sum() - t[worst::p].sum()) / (len(t) - len(t[worst::p]))))
    big = np.argsort(t)[-8:][::-1]
    print("   slowest calls:", ", ".join("%d: %.0f us" % (i, t[i]) for i in big), flush=True)


with tempfile.TemporaryDirectory() as d:
    for rep in range(3):
        p = os.path.join(d, "r%d.h264" % rep)
        hd = V.h264_open_file(p.encode(), w, h, h)
        time.sleep(0.3)
        ptrs = [fr[i].ctypes.data for i in range(n)]
        ts = []
        fn = V.h264_add_image_lossless
        for i in range(n):
            t0 = pc()
            fn(hd, ptrs[i], i * 1000, 0, None, None, None, None)
            ts.append(pc() - t0)
        t0 = pc()
        V.h264_close_file(hd)
        print("close %.2f ms" % ((pc() - t0) * 1e3))
        summary("add_image run %d" % rep, ts)
    for rep in range(4):
        ff = ct.c_int(0)
        cam = V.open_camera_file(p.encode(), ct.byref(ff))
        buf = np.empty((h, w), np.uint16)
        ptr = buf.ctypes.data
        ts = []
        for i in range(n):
            t0 = pc()
            V.load_image(cam, i, 0, ptr)
            ts.append(pc() - t0)
        V.close_camera(cam)
        summary("load_image run %d" % rep, ts, (50,))

# bounded-loss recording (low = high = 3, stdFactor 0: the parameters of the reference's own test), the same frames
with tempfile.TemporaryDirectory() as d:
    for rep in range(3):
        p = os.path.join(d, "l%d.h264" % rep)
        hd = V.h264_open_file(p.encode(), w, h, h - 3)
        for k, v in ((b"lowValueError", b"3"), (b"highValueError", b"3"), (b"stdFactor", b"0")):
            V.h264_set_parameter(hd, k, v)
        time.sleep(0.3)
        ts = []
        fn = V.h264_add_image_lossy
        for i in range(n):
            t0 = pc()
            fn(hd, ptrs[i], i * 1000, 0, None, None, None, None)
            ts.append(pc() - t0)
        t0 = pc()
        V.h264_close_file(hd)
        print("close %.2f ms" % ((pc() - t0) * 1e3))
        summary("add_image_lossy run %d" % rep, ts)
