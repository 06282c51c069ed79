"""Generate tests/golden/labelling_sha256.json, labelling.npz and time_series.npz from the UNMODIFIED reference C++.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_labelling_golden.py

Produced by oracle/_ref/librir_ref.so = the reference's own src/cpp/signal_processing/*.cpp compiled by oracle/build_ref.sh
(label_image, keep_largest_area, extract_times, resample_time_serie are its exported C entry points, called as they are).  Stored: data
only - the reference's outputs on the seeded inputs of cases.py: arrays for small images and all time vectors, SHA-256 of the output
bytes for the larger images.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cases import KEEP_PARAMS, LABEL_DTYPES, LABEL_SHAPES, label_cases, resample_cases, time_axis_cases  # noqa: E402
from oracle.pyoracle import Ref  # noqa: E402


def sha(*arrays):
    m = hashlib.sha256()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


def label_key(h, w, dt, name):
    return "%dx%d_%s_%s" % (h, w, np.dtype(dt).char, name)


def main():
    R = Ref()
    arrays, hashes = {}, {}
    for (h, w) in LABEL_SHAPES:
        for dt in LABEL_DTYPES:
            if h * w > 100000 and np.dtype(dt) not in (np.dtype(np.uint16), np.dtype(np.float32), np.dtype(np.uint8), np.dtype(np.int64)):
                continue  # the large geometries: one cell type of each width
            for name, img, bg in label_cases(h, w, dt):
                lab, area, xy = R.label_image(img, bg)
                key = label_key(h, w, dt, name)
                if h * w <= 100:
                    arrays["lab_" + key] = lab
                    arrays["area_" + key] = area
                    arrays["xy_" + key] = xy
                hashes["label_" + key] = sha(lab, area, xy)
                for kb, kf in KEEP_PARAMS:
                    hashes["keep_%s_%d_%d" % (key, kb, kf)] = sha(R.keep_largest_area(img, kb, kf))
    # the (int) conversion of a background that is not an integer: float cells
    for dt, bgv in ((np.float32, 2.75), (np.float64, -3.5)):
        img = label_cases(16, 20, dt)[0][1]
        img[img == 1] = bgv
        hashes["keep_fraction_%s" % np.dtype(dt).char] = sha(R.keep_largest_area(img, bgv, 9))
    out_dir = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(out_dir, "labelling.npz"), **arrays)
    with open(os.path.join(out_dir, "labelling_sha256.json"), "w") as f:
        json.dump(hashes, f, indent=0, sort_keys=True)
    print("labelling: %d arrays, %d hashes" % (len(arrays), len(hashes)))

    ts = {}
    for name, vs, s in time_axis_cases():
        rc, out = R.extract_times(vs, s)
        assert rc == 0, name
        ts["axis_" + name] = out
    for name, x, y, times, s, padd in resample_cases():
        rc, out = R.resample_time_serie(x, y, times, s, padd)
        assert rc == 0, name
        ts["resample_" + name] = out
    np.savez_compressed(os.path.join(out_dir, "time_series.npz"), **ts)
    print("time series: %d arrays" % len(ts))


if __name__ == "__main__":
    main()
