"""CPU: the oracle (oracle/rir_oracle.c) against the golden vectors produced by the compiled
reference (tests/golden/make_golden.py), against SURVEY.md Appendix A known answers, and - where
oracle/_ref exists - against the reference library itself on extra seeded inputs."""
import hashlib

import numpy as np
import pytest
from cases import (BADPIX_SHAPES, GAUSS_SHAPES, GAUSS_SIGMAS, MEDIAN_PERCENTS, TRANSLATE_DTYPES, TRANSLATE_OFFSETS, TRANSLATE_SHAPES,
                   TRANSLATE_STRATEGIES, badpix_frames, gauss_input, median_input, translate_input)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check(golden, key, out):
    arrays, hashes = golden
    if key in arrays.files:
        exp = arrays[key]
        assert out.dtype == exp.dtype and out.shape == exp.shape, key
        assert np.array_equal(out, exp), key  # bit-exact, floats included
    else:
        assert hashes[key] == sha(out), key


@pytest.mark.parametrize("shape", TRANSLATE_SHAPES)
@pytest.mark.parametrize("dtype", TRANSLATE_DTYPES)
def test_translate_golden(oracle, golden, shape, dtype):
    h, w = shape
    img = translate_input(h, w, dtype)
    for strat in TRANSLATE_STRATEGIES:
        for k, (dx, dy) in enumerate(TRANSLATE_OFFSETS(w, h)):
            out = oracle.translate(img, dx, dy, strat, background=7)
            check(golden, "tr_%dx%d_%s_%s_%d" % (h, w, np.dtype(dtype).char, strat or "none", k), out)


@pytest.mark.parametrize("shape", TRANSLATE_SHAPES)
def test_translate_u16_f32_golden(oracle, golden, shape):
    h, w = shape
    img = translate_input(h, w, np.uint16)
    for k, (dx, dy) in enumerate(TRANSLATE_OFFSETS(w, h)):
        check(golden, "tr16f_%dx%d_%d" % (h, w, k), oracle.translate_u16_f32_nearest(img, dx, dy))


@pytest.mark.parametrize("shape", GAUSS_SHAPES)
def test_gaussian_golden(oracle, golden, shape):
    h, w = shape
    img = gauss_input(h, w)
    for s in GAUSS_SIGMAS:
        check(golden, "ga_%dx%d_%g" % (h, w, s), oracle.gaussian_filter(img, s))


@pytest.mark.parametrize("case", BADPIX_SHAPES)
def test_bad_pixels_golden(oracle, golden, case):
    h, w, seed = case
    arrays, _ = golden
    first, second = badpix_frames(h, w, seed)
    xy = oracle.bad_pixels_detect(first)
    key = "bp_%dx%d" % (h, w)
    assert np.array_equal(xy, arrays[key + "_xy"])
    _, floor_correct = oracle.bad_pixels_stats(first)
    assert max(floor_correct, 0) == int(arrays[key + "_floor"][0])
    check(golden, key + "_corrected", oracle.bad_pixels_correct(second, xy, floor_correct))


@pytest.mark.parametrize("n", [100, 5000, 327680])
def test_find_median_pixel_golden(oracle, golden, n):
    arrays, _ = golden
    img, mask = median_input(n)
    assert [oracle.find_median_pixel(img, p) for p in MEDIAN_PERCENTS] == list(arrays["mp_%d" % n])
    assert [oracle.find_median_pixel(img, p, mask) for p in MEDIAN_PERCENTS] == list(arrays["mpm_%d" % n])


@pytest.mark.parametrize("shape", TRANSLATE_SHAPES)
def test_median_filter_golden(oracle, golden, shape):
    h, w = shape
    check(golden, "mf_%dx%d" % (h, w), oracle.median_filter(translate_input(h, w, np.uint16)))


# ---- answers captured from the compiled reference by the survey ----------------

S = np.array([[5, 8, 17, 32, 53], [12, 26, 46, 72, 104], [33, 58, 89, 126, 169], [68, 104, 146, 194, 248]])

A1 = [
    (1, 0, "noborder", 9, [[5, 5, 8, 17, 32], [12, 12, 26, 46, 72], [33, 33, 58, 89, 126], [68, 68, 104, 146, 194]]),
    (1, 0, "background", 9, [[9, 5, 8, 17, 32], [9, 12, 26, 46, 72], [9, 33, 58, 89, 126], [9, 68, 104, 146, 194]]),
    (1, 0, "wrap", 9, [[104, 5, 8, 17, 32], [169, 12, 26, 46, 72], [248, 33, 58, 89, 126], [53, 68, 104, 146, 194]]),
    (0.5, 0.25, "noborder", 9, [[5, 8, 17, 32, 53], [12, 15, 30, 50, 76], [33, 38, 64, 95, 132], [68, 75, 112, 154, 202]]),
    (0.5, 0.25, "background", 9, [[9, 9, 9, 9, 9], [9, 15, 30, 50, 76], [9, 38, 64, 95, 132], [9, 75, 112, 154, 202]]),
    (0.5, 0.25, "nearest", 9, [[5, 5, 8, 17, 32], [5, 15, 30, 50, 76], [12, 38, 64, 95, 132], [33, 75, 112, 154, 202]]),
    (0.5, 0.25, "wrap", 9, [[5, 6, 12, 24, 42], [6, 15, 30, 50, 76], [17, 38, 64, 95, 132], [41, 75, 112, 154, 202]]),
    (-0.75, 1.5, "background", 9, [[9, 9, 9, 9, 9], [9, 9, 9, 9, 9], [14, 27, 46, 71, 78], [37, 61, 91, 127, 136]]),
    (-0.75, 1.5, "nearest", 9, [[5, 8, 17, 32, 53], [5, 8, 17, 32, 53], [14, 27, 46, 71, 78], [37, 61, 91, 127, 136]]),
    (-0.75, 1.5, "wrap", 9, [[51, 75, 105, 141, 65], [7, 14, 28, 47, 17], [14, 27, 46, 71, 78], [37, 61, 91, 127, 136]]),
]


@pytest.mark.parametrize("case", A1)
def test_appendix_a1_translate(oracle, case):
    dx, dy, strat, bg, exp = case
    out = oracle.translate(S.astype(np.uint16), dx, dy, strat, background=bg)
    assert out.tolist() == exp


def test_appendix_a1_translate_f32(oracle):
    out = oracle.translate(S.astype(np.float32), 0.5, 0.25, "nearest")
    exp = [[5, 5, 8, 17, 32], [5, 15.875, 30.125, 50.375, 76.625], [12, 38.875, 64.125, 95.375, 132.625], [33, 75.875, 112.125, 154.375, 202.625]]
    assert out.tolist() == exp


def test_appendix_a2_gaussian(oracle):
    k = oracle.gaussian_kernel(0.75)
    assert k.shape == (3, 3)
    bits = k.view(np.uint32)
    assert bits[0, 0] == 0x3D507C72 and bits[0, 1] == 0x3DFD903D and bits[1, 1] == 0x3E9A318B
    imp = np.zeros((5, 5), np.float32)
    imp[2, 2] = 1
    assert np.array_equal(oracle.gaussian_filter(imp, 0.75)[1:4, 1:4], k)
    imp = np.zeros((5, 5), np.float32)
    imp[0, 0] = 1
    out = oracle.gaussian_filter(imp, 0.75)
    assert np.allclose([out[0, 0], out[0, 1], out[1, 0], out[1, 1]], [0.5022001, 0.15988104, 0.15988104, 0.05089993], rtol=1e-6)
    out = oracle.gaussian_filter(S.astype(np.float32), 1.0)
    exp = np.array([[14.379627, 21.723728, 35.46504, 52.759754, 66.87213], [26.99809, 38.457584, 58.15166, 81.39914, 99.62693],
                    [47.4199, 64.05667, 91.2395, 121.97574, 145.3808], [67.267784, 88.01995, 121.15554, 157.84456, 185.36504]], np.float32)
    assert np.allclose(out, exp, rtol=1e-6)


def test_appendix_a3_bad_pixels(oracle):
    first, _ = badpix_frames(8, 10, 5)
    xy = oracle.bad_pixels_detect(first)
    assert sorted(map(tuple, xy.tolist())) == sorted([(0, 0), (4, 3), (5, 3), (9, 7)])
    _, fc = oracle.bad_pixels_stats(first)
    out = oracle.bad_pixels_correct(first, xy, fc)
    changed = np.argwhere(out != first)
    assert sorted(map(tuple, changed.tolist())) == sorted([(0, 0), (3, 4), (3, 5), (7, 9)])
    assert [out[0, 0], out[3, 4], out[3, 5], out[7, 9]] == [1006, 1002, 1002, 1007]


def test_appendix_a4_find_median_pixel(oracle):
    img, mask = median_input(100)
    assert [oracle.find_median_pixel(img, p) for p in MEDIAN_PERCENTS] == [0, 19, 49, 50, 98, 99]
    assert [oracle.find_median_pixel(img, p, mask) for p in MEDIAN_PERCENTS] == [0, 18, 48, 48, 99, 99]


# ---- oracle vs the reference itself on extra inputs (only where oracle/_ref exists) --------------------


def test_oracle_vs_ref_translate_all_dtypes(oracle, ref):
    rng = np.random.default_rng(11)
    for dt in [np.bool_, np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64, np.float32, np.float64]:
        for (h, w) in [(3, 3), (1, 7), (31, 17)]:
            if dt == np.bool_:
                img = rng.integers(0, 2, (h, w)).astype(dt)
            elif np.dtype(dt).kind == "f":
                img = (rng.random((h, w)) * 1000).astype(dt)
            else:
                img = rng.integers(-100 if np.iinfo(dt).min < 0 else 0, min(np.iinfo(dt).max, 16383), (h, w)).astype(dt)
            for strat in ["", "noborder", "background", "wrap", "nearest"]:
                for (dx, dy) in [(0.4, -0.6), (-w - 2.5, h + 1.5), (2, 2), (-0.01, 0.01)]:
                    a = oracle.translate(img, dx, dy, strat, background=1)
                    b = ref.translate(img, dx, dy, strat, background=1)
                    assert np.array_equal(a, b), (dt, h, w, strat, dx, dy)


def test_oracle_vs_ref_misc(oracle, ref):
    rng = np.random.default_rng(12)
    img = (rng.random((37, 53)) * 1000 + 3).astype(np.float32)
    for s in [0.2, 0.9, 1.7, 3.3]:
        assert np.array_equal(oracle.gaussian_filter(img, s), ref.gaussian_filter(img, s))
    first, second = badpix_frames(40, 33, 9)
    assert np.array_equal(oracle.bad_pixels_detect(first), ref.bad_pixels_detect(first))
    xy = oracle.bad_pixels_detect(first)
    _, fc = oracle.bad_pixels_stats(first)
    assert np.array_equal(oracle.bad_pixels_correct(second, xy, fc), ref.bad_pixels_correct(first, second))
    assert max(fc, 0) == ref.bad_pixels_floor(first)


def test_byte_plane_split_merge_known_answer(oracle):
    """C1 / C2 (h264.cpp:1066-1082, :3016-3051): U = v & 0xFF, V = v >> 8, Y = 0 or the 8-bit IT image, rows padded."""
    img = np.array([[0x0102, 0xFFFE, 0x8000], [0x00FF, 0x1234, 0x0000]], np.uint16)
    Y, U, V = oracle.split_planes(img, linesize=5)
    assert U[:, :3].tolist() == [[0x02, 0xFE, 0x00], [0xFF, 0x34, 0x00]] and V[:, :3].tolist() == [[0x01, 0xFF, 0x80], [0x00, 0x12, 0x00]]
    assert not Y.any() and not U[:, 3:].any()
    it = np.array([[1, 2, 3], [4, 5, 6]], np.uint8)
    Y, U, V = oracle.split_planes(img, linesize=4, it=it)
    back, it2 = oracle.merge_planes(Y, U, V, 3, with_it=True)
    assert np.array_equal(back, img) and np.array_equal(it2, it)
