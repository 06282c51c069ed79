"""The oracle's connected-component restatement (oracle/rir_oracle.c: orc_label_image, orc_keep_largest_area) against the golden
vectors made from the compiled reference (tests/golden/make_labelling_golden.py) and, where it was built, against the compiled
reference itself on random images."""
import hashlib
import json
import os

import numpy as np
import pytest

from cases import KEEP_PARAMS, LABEL_DTYPES, LABEL_SHAPES, label_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIG_TYPES = (np.dtype(np.uint16), np.dtype(np.float32), np.dtype(np.uint8), np.dtype(np.int64))


def sha(*arrays):
    m = hashlib.sha256()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


def label_key(h, w, dt, name):
    return "%dx%d_%s_%s" % (h, w, np.dtype(dt).char, name)


@pytest.fixture(scope="module")
def label_golden():
    d = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(d, "labelling_sha256.json")) as f:
        return np.load(os.path.join(d, "labelling.npz")), json.load(f)


def golden_cases(max_px=None):
    for (h, w) in LABEL_SHAPES:
        if max_px and h * w > max_px:
            continue
        for dt in LABEL_DTYPES:
            if h * w > 100000 and np.dtype(dt) not in BIG_TYPES:
                continue
            for name, img, bg in label_cases(h, w, dt):
                yield h, w, dt, name, img, bg


def check_against_golden(impl, label_golden, max_px=None):
    arrays, hashes = label_golden
    n = 0
    for h, w, dt, name, img, bg in golden_cases(max_px):
        key = label_key(h, w, dt, name)
        lab, area, xy = impl.label_image(img, bg)
        assert lab.dtype == np.int32 and area.dtype == np.int32 and xy.dtype == np.float64
        if h * w <= 100:
            assert np.array_equal(lab, arrays["lab_" + key]) and np.array_equal(area, arrays["area_" + key]), key
            assert np.array_equal(xy, arrays["xy_" + key]), key
        assert sha(lab, area, xy) == hashes["label_" + key], key
        for kb, kf in KEEP_PARAMS:
            assert sha(impl.keep_largest_area(img, kb, kf)) == hashes["keep_%s_%d_%d" % (key, kb, kf)], (key, kb, kf)
        n += 1
    for dt, bgv in ((np.float32, 2.75), (np.float64, -3.5)):
        img = label_cases(16, 20, dt)[0][1]
        img[img == 1] = bgv
        assert sha(impl.keep_largest_area(img, bgv, 9)) == hashes["keep_fraction_%s" % np.dtype(dt).char]
    return n


def test_oracle_against_the_reference_vectors(oracle, label_golden):
    assert check_against_golden(oracle, label_golden) > 600


def test_oracle_against_the_compiled_reference_on_random_images(oracle, ref):
    rng = np.random.default_rng(0)
    for it in range(1200):
        h, w = int(rng.integers(1, 24)), int(rng.integers(1, 140))
        dt = LABEL_DTYPES[it % len(LABEL_DTYPES)]
        img = rng.integers(0, int(rng.integers(1, 5)) + 1, (h, w)).astype(dt)
        if np.dtype(dt).kind == "f" and it % 3 == 0:
            img[rng.random((h, w)) < 0.05] = np.nan
        bg = int(rng.integers(0, 2))
        for a, b in zip(oracle.label_image(img, bg), ref.label_image(img, bg)):
            assert np.array_equal(a, b)
        assert np.array_equal(oracle.keep_largest_area(img, bg, 5), ref.keep_largest_area(img, bg, 5))


def test_what_the_labels_mean(oracle):
    """the properties the device kernels are built on (csrc/label_kernels.hip): vertical neighbours join whatever their values,
    horizontal ones when equal; numbers follow the first pixels; the table's second column repeats the first"""
    img = np.array([[1, 2, 0, 3],
                    [0, 2, 0, 4],
                    [5, 5, 0, 0],
                    [0, 6, 7, 7]], dtype=np.uint16)
    lab, area, xy = oracle.label_image(img, 0)
    assert np.array_equal(lab, [[1, 2, 0, 3], [0, 2, 0, 3], [2, 2, 0, 0], [0, 2, 4, 4]])
    assert np.array_equal(area, [0, 1, 5, 2, 2])
    assert np.array_equal(xy, [[-1, -1], [0, 0], [1, 1], [3, 3], [2, 2]])
    # the earlier component wins among equals; everything else takes (int)background
    assert np.array_equal(oracle.keep_largest_area(np.array([[1, 1, 0, 2, 2]], np.uint8), 0, 9), [[9, 9, 0, 0, 0]])
    assert np.array_equal(oracle.keep_largest_area(np.zeros((2, 3), np.uint8), 0, 9), np.zeros((2, 3)))
