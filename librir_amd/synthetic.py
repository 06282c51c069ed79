"""Seeded synthetic IR streams, deterministic versions of the reference test
recipes (reference tests/python/conftest.py:48-66,210-219; tests/python/test_registration.py:41-59).
Values stay <= 16383 like the reference fixtures."""
import numpy as np


def s1_noisy_background(n, h=512, w=640, seed=1234):
    """frame i = uint16(bg + 10 + i + N(0, sqrt(0.5))), bg = rand*1000"""
    rng = np.random.default_rng(seed)
    bg = rng.random((h, w)) * 1000
    out = np.empty((n, h, w), dtype=np.uint16)
    for i in range(n):
        out[i] = (bg + 10 + (i % 8192) + rng.normal(0, np.sqrt(0.5), (h, w))).astype(np.uint16)
    return out


def s2_uniform_dl_ti(n, h=512, w=640, seed=4321):
    """frame i constant dl_i | (ti_i << 13)"""
    rng = np.random.default_rng(seed)
    dl = rng.integers(0, 8191, n)
    ti = rng.integers(0, 7, n)
    out = np.empty((n, h, w), dtype=np.uint16)
    for i in range(n):
        out[i] = np.uint16(int(dl[i]) | (int(ti[i]) << 13))
    return out


def bad_pixel_positions(count, h=512, w=640, seed=7):
    rng = np.random.default_rng(seed)
    idx = rng.choice(h * w, size=count, replace=False)
    return np.stack([idx % w, idx // w], axis=1).astype(np.int32), rng.choice([0, 16000], size=count).astype(np.uint16)


def inject_bad_pixels(frames, count=200, seed=7):
    n, h, w = frames.shape
    xy, vals = bad_pixel_positions(count, h, w, seed)
    out = frames.copy()
    out[:, xy[:, 1], xy[:, 0]] = vals[None, :]
    return out


def _fill_polygon(h, w, poly, value):
    """Even-odd scan-line fill of a polygon given as [[x, y], ...] (pixel centres)."""
    img = np.zeros((h, w), dtype=np.float64)
    pts = np.asarray(poly, dtype=np.float64)
    xs, ys = pts[:, 0], pts[:, 1]
    xx = np.arange(w)[None, :]
    for y in range(h):
        inside = np.zeros((1, w), dtype=bool)
        for i in range(len(pts)):
            j = (i + 1) % len(pts)
            if (ys[i] > y) != (ys[j] > y):
                xc = xs[i] + (y - ys[i]) * (xs[j] - xs[i]) / (ys[j] - ys[i])
                inside ^= xx < xc
        img[y, inside[0]] = value
    return img


def s3_registration(n, h=512, w=640, seed=99):
    """Registration stream (reference tests/python/test_registration.py:20-59 made deterministic): a polygon of
    value 10 on a flat background, frame i shifted by (i, i) pixels (edge-replicated), plus the level 10 + i
    and N(0,1) noise.  Returns (float32 frames, int shifts (n,2) as (dx, dy))."""
    rng = np.random.default_rng(seed)
    poly = _fill_polygon(h, w, [[42, 42], [100, 42], [200, 200], [80, 300]], 10.0)
    out = np.empty((n, h, w), dtype=np.float32)
    shifts = np.empty((n, 2), dtype=np.int32)
    yy, xx = np.arange(h)[:, None], np.arange(w)[None, :]
    for i in range(n):
        d = i % 100
        src = poly[np.clip(yy - d, 0, h - 1), np.clip(xx - d, 0, w - 1)]
        out[i] = (src + 10 + d + rng.normal(0, 1, (h, w))).astype(np.float32)
        shifts[i] = (d, d)
    return out, shifts
