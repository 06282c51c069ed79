"""Generate tests/golden/signal_processing.npz from the UNMODIFIED reference C++.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

The vectors are produced by oracle/_ref/librir_ref.so = the reference's own
src/cpp/signal_processing/*.cpp compiled by oracle/build_ref.sh.  What is stored is data only:
seeded inputs (small) and the reference's outputs - full arrays for the small shapes, SHA-256 of
the output bytes for the larger ones (SURVEY.md §8c "Goldens to commit").
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cases import (BADPIX_SHAPES, GAUSS_SHAPES, GAUSS_SIGMAS, MEDIAN_PERCENTS, TRANSLATE_DTYPES, TRANSLATE_OFFSETS,  # noqa: E402
                   TRANSLATE_SHAPES, TRANSLATE_STRATEGIES, badpix_frames, gauss_input, median_input, translate_input)
from oracle.pyoracle import Ref  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    R = Ref()
    arrays = {}
    hashes = {}

    # F1 translate ------------------------------------------------------------------------------
    for (h, w) in TRANSLATE_SHAPES:
        full = h * w <= 400
        for dt in TRANSLATE_DTYPES:
            img = translate_input(h, w, dt)
            for strat in TRANSLATE_STRATEGIES:
                for k, (dx, dy) in enumerate(TRANSLATE_OFFSETS(w, h)):
                    out = R.translate(img, dx, dy, strat, background=7)
                    key = "tr_%dx%d_%s_%s_%d" % (h, w, np.dtype(dt).char, strat or "none", k)
                    if full:
                        arrays[key] = out
                    else:
                        hashes[key] = sha(out)

    # F6 removeMotion building block: translate<u16,float> nearest -------------------------------
    for (h, w) in TRANSLATE_SHAPES:
        img = translate_input(h, w, np.uint16)
        for k, (dx, dy) in enumerate(TRANSLATE_OFFSETS(w, h)):
            out = R.translate_u16_f32_nearest(img, dx, dy)
            key = "tr16f_%dx%d_%d" % (h, w, k)
            if h * w <= 400:
                arrays[key] = out
            else:
                hashes[key] = sha(out)

    # F2 gaussian ------------------------------------------------------------------------------------
    for (h, w) in GAUSS_SHAPES:
        img = gauss_input(h, w)
        for s in GAUSS_SIGMAS:
            out = R.gaussian_filter(img, s)
            key = "ga_%dx%d_%g" % (h, w, s)
            if h * w <= 400:
                arrays[key] = out
            else:
                hashes[key] = sha(out)

    # F3 / F4 bad pixels -------------------------------------------------------------------------------
    for (h, w, seed) in BADPIX_SHAPES:
        first, second = badpix_frames(h, w, seed)
        xy = R.bad_pixels_detect(first)
        floor = R.bad_pixels_floor(first)
        corrected = R.bad_pixels_correct(first, second)
        key = "bp_%dx%d" % (h, w)
        arrays[key + "_xy"] = xy.astype(np.int32)
        arrays[key + "_floor"] = np.array([floor], dtype=np.int32)
        if h * w <= 4096:
            arrays[key + "_corrected"] = corrected
        else:
            hashes[key + "_corrected"] = sha(corrected)

    # F7 quantile --------------------------------------------------------------------------------------
    for n in (100, 5000, 327680):
        img, mask = median_input(n)
        arrays["mp_%d" % n] = np.array([R.find_median_pixel(img, p) for p in MEDIAN_PERCENTS], dtype=np.int32)
        arrays["mpm_%d" % n] = np.array([R.find_median_pixel(img, p, mask) for p in MEDIAN_PERCENTS], dtype=np.int32)

    # F8 3x3 median filter -------------------------------------------------------------------------------
    for (h, w) in TRANSLATE_SHAPES:
        if h >= 3 and w >= 3:
            out = R.median_filter(translate_input(h, w, np.uint16))
            key = "mf_%dx%d" % (h, w)
            if h * w <= 400:
                arrays[key] = out
            else:
                hashes[key] = sha(out)

    out_dir = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(out_dir, "signal_processing.npz"), **arrays)
    with open(os.path.join(out_dir, "signal_processing_sha256.json"), "w") as f:
        json.dump(hashes, f, indent=0, sort_keys=True)
    print("wrote %d arrays, %d hashes" % (len(arrays), len(hashes)))


if __name__ == "__main__":
    main()
