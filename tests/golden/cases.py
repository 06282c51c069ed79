"""Seeded inputs and the case grid shared by make_golden.py (which runs the reference) and the tests
(which run the oracle and the HIP path on the same inputs).  SURVEY.md §8c "Goldens to commit"."""
import numpy as np

TRANSLATE_SHAPES = [(4, 5), (16, 20), (48, 64), (67, 83), (512, 640)]
TRANSLATE_DTYPES = [np.uint8, np.uint16, np.int16, np.int32, np.float32, np.float64]
TRANSLATE_STRATEGIES = ["", "background", "nearest", "wrap"]


def TRANSLATE_OFFSETS(w, h):
    return [(0, 0), (1, 0), (0, -1), (3, -2), (0.5, 0.5), (1.25, -2.5), (-0.75, 0.1), (w + 1, 0), (0.5, 0.25), (-0.75, 1.5), (0.3, h - 0.5)]


def translate_input(h, w, dtype):
    """SURVEY Appendix A.1 polynomial for the 4x5 case, seeded noise otherwise; values <= 16383."""
    dtype = np.dtype(dtype)
    if (h, w) == (4, 5):
        y, x = np.mgrid[0:h, 0:w]
        v = 7 * y * y + 3 * x * x + 11 * x * y + 5
        return v.astype(dtype)
    rng = np.random.default_rng(h * 1000 + w)
    if dtype.kind == "f":
        return (rng.random((h, w)) * 1000).astype(dtype)
    hi = min(np.iinfo(dtype).max, 16383)
    lo = -100 if np.iinfo(dtype).min < 0 else 0
    return rng.integers(lo, hi, (h, w)).astype(dtype)


GAUSS_SHAPES = [(3, 5), (4, 5), (16, 20), (48, 64), (67, 83), (512, 640)]
GAUSS_SIGMAS = [0.3, 0.5, 0.75, 1.0, 1.49, 2.0]


def gauss_input(h, w):
    if (h, w) == (4, 5):
        return translate_input(4, 5, np.float32)
    rng = np.random.default_rng(77 + h * 1000 + w)
    return (rng.random((h, w)) * 1000 + 10).astype(np.float32)


BADPIX_SHAPES = [(8, 10, 5), (48, 64, 1), (67, 83, 2), (512, 640, 3)]


def badpix_frames(h, w, seed):
    """conftest.images-style frames with injected spikes / dead pixels (corners, edges, an adjacent pair)."""
    if (h, w, seed) == (8, 10, 5):
        rng = np.random.default_rng(5)
        first = (1000 + rng.integers(0, 8, (8, 10))).astype(np.uint16)
        first[0, 0] = 3000
        first[3, 4] = 0
        first[3, 5] = 5000
        first[7, 9] = 2500
        return first, first.copy()
    rng = np.random.default_rng(seed)
    bg = rng.random((h, w)) * 1000
    first = (bg + 10 + rng.normal(0, np.sqrt(0.5), (h, w))).astype(np.uint16)
    k = max(1, (h * w) // 500)
    ys = rng.integers(0, h, k)
    xs = rng.integers(0, w, k)
    vals = rng.choice([0, 16000], k).astype(np.uint16)
    first[ys, xs] = vals
    first[0, 0] = 16000
    first[h - 1, w - 1] = 0
    first[h // 2, w // 2] = 0
    first[h // 2, w // 2 + 1] = 16000
    second = (bg + 12 + rng.normal(0, np.sqrt(0.5), (h, w))).astype(np.uint16)
    second[ys, xs] = vals
    second[0, 0] = 16000
    second[h - 1, w - 1] = 0
    second[h // 2, w // 2] = 0
    second[h // 2, w // 2 + 1] = 16000
    return first, second


MEDIAN_PERCENTS = [0.0, 0.2, 0.5, 0.505, 0.99, 1.0]


def median_input(n):
    if n == 100:
        img = np.arange(100, dtype=np.uint16).reshape(1, 100)
        mask = (np.arange(100) % 3 == 0).astype(np.uint8).reshape(1, 100)
        return img, mask
    rng = np.random.default_rng(n)
    img = rng.integers(0, 16384, (1, n)).astype(np.uint16)
    mask = (rng.random((1, n)) < 0.3).astype(np.uint8)
    return img, mask
