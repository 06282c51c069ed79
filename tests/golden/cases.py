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


# ---- connected components (label_image / keep_largest_area) ------------------------------------------------------------------------
LABEL_DTYPES = [np.bool_, np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64, np.float32, np.float64]
LABEL_SHAPES = [(1, 1), (1, 70), (9, 1), (4, 5), (16, 20), (23, 64), (31, 65), (48, 129), (67, 83), (512, 640), (768, 1024)]


def _smooth(rng, h, w):
    c = 8
    a = rng.normal(size=(h // c + 3, w // c + 3))
    a = np.kron(a, np.ones((c, c)))
    for _ in range(3):
        a = (a + np.roll(a, 3, 0) + np.roll(a, 3, 1) + np.roll(a, -3, 0) + np.roll(a, -3, 1)) / 5
    return a[:h, :w]


def _spiral(h, w):
    img = np.zeros((h, w), np.int64)
    t, b, l, r = 0, h - 1, 0, w - 1
    while t <= b and l <= r:
        img[t, l:r + 1] = 1
        img[t:b + 1, r] = 1
        if b - t >= 2:
            img[b, l + 2:r + 1] = 1
        if r - l >= 2 and b - t >= 2:
            img[t + 2:b + 1, l + 2] = 1
        t, b, l, r = t + 2, b - 2, l + 2, r - 2
    return img


def label_cases(h, w, dtype):
    """-> [(name, image, background value)] for one geometry and cell type: what the labelling meets in use (a few regions on a
    background) and what stresses it (one component, stripes either way, isolated pixels, noise, a spiral, NaN cells)."""
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(h * 7919 + w * 31 + ord(dtype.char))
    yy, xx = np.mgrid[0:h, 0:w]
    levels = 2 if dtype == np.bool_ else 4
    sm = _smooth(rng, h, w)
    out = [
        ("regions", np.digitize(sm, [0.15, 0.45, 0.8][:levels - 1]), 0),
        ("background_only", np.zeros((h, w), np.int64), 0),
        ("flat", np.ones((h, w), np.int64), 0),
        ("noise", rng.integers(0, levels, (h, w)), 1),
        ("columns", (xx % 3) + (0 if dtype == np.bool_ else 1), 0 if dtype != np.bool_ else 2),
        ("rows", yy % 2, 0),
        ("isolated", (xx + yy) % 2, 0),
        ("spiral", _spiral(h, w), 0),
    ]
    res = []
    for name, img, bg in out:
        a = img.astype(dtype)
        if dtype.kind == "f" and name in ("regions", "noise"):
            a[rng.random((h, w)) < 0.02] = np.nan
            if name == "noise":
                a[a == 2] = -0.0  # equal to +0.0 as numbers, not as bits
                a[0, 0] = 0.0
        if dtype.kind == "i" and name == "regions":
            a[a == 2] = -5
        res.append((name, np.ascontiguousarray(a), bg))
    return res


# keep_largest_area: (background, foreground) pairs tried on every case; a fractional / negative background shows the (int) conversion
KEEP_PARAMS = [(0, 1), (1, -3)]


# ---- time axes (extract_times / resample_time_serie) ----------------------------------------------------------------------------------
def time_axis_cases():
    """-> [(name, [vectors], strategy)]: inputs the reference returns from (tests/python/test_rir.py:232-243 first)."""
    rng = np.random.default_rng(4242)
    t1 = [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5]
    t2 = [-1, 3, 4, 4.3, 4.7]
    cases = [("ref_union", [t1, t2], 0), ("ref_inter", [t1, t2], 1), ("ref_long", [t1, list(range(10000))], 0), ("single", [[3, 1, 2, 2]], 0),
             ("single_inter", [[5, 4]], 1), ("disjoint", [[0, 1, 2], [5, 6]], 1), ("touching", [[0, 1, 2], [2, 3]], 1),
             ("nan_cut", [[0, 1, np.nan, 0.5, 4], [1, 2, 3]], 0), ("duplicates", [[0, 0, 1, 1, 1, 2], [1, 2, 2]], 0),
             ("signed_zero", [[-0.0, 1], [0.0, 2]], 0), ("signed_zero_2", [[0.0, 1], [-0.0, 2]], 0)]
    for k in range(40):
        nv = int(rng.integers(2, 6))
        vs = []
        for _ in range(nv):
            v = np.sort(rng.integers(0, 200, int(rng.integers(1, 60))) * 0.125)
            if rng.random() < 0.5:
                v = np.unique(v)
            vs.append(v)
        cases.append(("union_%d" % k, vs, 0))
        lo, hi = max(v[0] for v in vs), min(v[-1] for v in vs)
        if lo <= hi and all(((v >= lo) & (v <= hi)).any() for v in vs):
            cases.append(("inter_%d" % k, vs, 1))
        if k % 4 == 0:
            ws = [v.copy() for v in vs]
            for v in ws:
                if v.size >= 3:
                    v[int(rng.integers(1, v.size - 1))] = np.nan
            cases.append(("nan_%d" % k, ws, 0))
        if k % 5 == 0:
            cases.append(("unsorted_%d" % k, [rng.permutation(v) for v in vs], 0))
    return cases


def resample_cases():
    """-> [(name, x, y, times, strategy, padd)]; strategy bits: 2 pad, 4 interpolate."""
    rng = np.random.default_rng(777)
    x = list(range(10))
    times = [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5, 5.6, 9.9, 10, 12, 13]
    cases = [("ref_default", x, x, times, 4, 0.0), ("ref_padd0", x, x, times, 6, 0.0), ("ref_nearest", x, x, times, 0, 0.0),
             ("before_and_after", [2, 3], [10, 20], [0, 1, 2, 2, 2.5, 3, 3, 4], 4, 0.0), ("padded", [2, 3], [10, 20], [0, 1, 2, 2.5, 3, 4], 6, -1.5),
             ("repeated_times", [0, 1, 2, 3], [0, 10, 20, 30], [0.5, 1, 1, 1, 2, 2, 5], 4, 0.0), ("no_samples", [], [], [0, 1, 2], 2, 9.0),
             ("no_samples_unpadded", [], [], [0, 1], 4, 9.0), ("backwards_times", [0, 1, 2, 3], [5, 6, 7, 8], [2.5, 0.5, 3, 1], 4, 0.0)]
    for k in range(60):
        n = int(rng.integers(1, 40))
        xs = np.sort(rng.integers(0, 80, n) * 0.25)
        if k % 3 == 0:
            xs = np.unique(xs)
        ys = rng.normal(size=xs.size) * 100
        m = int(rng.integers(1, 90))
        ts = np.sort(rng.integers(-10, 90, m) * 0.25 + (0.0 if k % 2 else rng.random(m) * 0.2))
        cases.append(("random_%d" % k, xs, ys, ts, int(rng.choice([0, 2, 4, 6])), float(np.round(rng.normal(), 3))))
    return cases
