#!/usr/bin/env python3
"""Freezes the RIRB1 format: SHA-256 of the tables and payload the oracle produces for a few seeded streams
(plus one tiny stream stored verbatim).  The format is this build's own (the reference codec is libx264, not
buildable here), so these are not reference outputs: they pin the format ACROSS ROUNDS - an accidental change of
the bitstream shows up as a failing hash, for the oracle on the CPU and for the GPU encoder against the same file.

    python tests/golden/make_codec_golden.py        (writes tests/golden/codec_format.json)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from librir_amd.synthetic import s1_noisy_background, s2_uniform_dl_ti  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402


def streams():
    rng = np.random.default_rng(2024)
    yield "s1_12x128x160", s1_noisy_background(12, 128, 160)
    yield "s1_50x64x96", s1_noisy_background(50, 64, 96, seed=9)
    yield "s2_10x64x80", s2_uniform_dl_ti(10, 64, 80)
    yield "rand_5x67x83", rng.integers(0, 65536, (5, 67, 83)).astype(np.uint16)
    yield "ramp_9x16x64", (np.arange(16 * 64, dtype=np.uint32).reshape(16, 64)[None] * 3 + np.arange(9)[:, None, None] * 7).astype(np.uint16)
    yield "wide_6x32x48", (s1_noisy_background(6, 32, 48).astype(np.int64) + rng.integers(0, 300, (6, 32, 48))).astype(np.uint16)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    O = Oracle()
    out = {"format": "RIRB1", "cases": {}}
    for name, fr in streams():
        hdr, off, st = O.codec_encode_chunk(fr)
        out["cases"][name] = {"hdr": digest(hdr), "tile_off": digest(off), "stream": digest(st), "words": int(st.size)}
    tiny = np.array([[[100, 101, 103, 100]], [[101, 101, 104, 99]]], np.uint16)  # 2 frames of 1x4: readable by eye
    hdr, off, st = O.codec_encode_chunk(tiny)
    out["tiny_2x1x4"] = {"frames": tiny.tolist(), "hdr": [int(x) for x in hdr.ravel()], "tile_off": [int(x) for x in off.ravel()],
                         "stream": [int(x) for x in st.ravel()]}
    json.dump(out, open(os.path.join(HERE, "codec_format.json"), "w"), indent=1)
    print(json.dumps(out["tiny_2x1x4"]))


if __name__ == "__main__":
    main()
