"""One-off soak: random sizes / shifts / strategies / dtypes, device filters against the oracle (not collected by pytest).
    python tests/perf/soak_filters.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D
from oracle.pyoracle import Oracle
O = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
DT = [np.uint16, np.int16, np.float32, np.int32, np.uint8, np.float64, np.uint32]
def fail(*a):
    global bad
    bad += 1
    print("FAIL", *a)
for k in range(cases):
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 200))
    if rng.random() < 0.3:
        w = int(rng.choice([8, 16, 64, 640, 641, 320])); h = int(rng.choice([2, 3, 5, 33, 64]))
    elif rng.random() < 0.25:  # several tiles in both directions: the regular-tile paths of translate / gaussian / filter_chain
        w = int(rng.integers(130, 400)); h = int(rng.integers(40, 150))
    n = int(rng.integers(1, 4))
    dt = DT[int(rng.integers(0, len(DT)))]
    if np.issubdtype(dt, np.floating):
        img = (rng.random((n, h, w)) * 2000 - 500).astype(dt)
    else:
        info = np.iinfo(dt)
        img = rng.integers(max(info.min, -30000), min(info.max, 60000) + 1, (n, h, w)).astype(dt)
    strat = str(rng.choice(["", "background", "wrap", "nearest"]))
    mag = float(rng.choice([0.5, 3, 20, 300]))
    dx, dy = float(np.float32(rng.normal(0, mag))), float(np.float32(rng.normal(0, mag)))
    if rng.random() < 0.2:
        dx, dy = float(int(dx)), float(int(dy))
    elif rng.random() < 0.15:  # just below an integer: px + 1 rounds across it (Filters.h:303)
        dx, dy = float(np.nextafter(np.float32(int(dx)), np.float32(-1e9))), float(np.nextafter(np.float32(int(dy)), np.float32(1e9)))
    t = torch.from_numpy(img).cuda()
    g = D.translate(t, (dx, dy), strat, background=7).cpu().numpy()
    r = np.stack([O.translate(img[i], dx, dy, strat, background=7) for i in range(n)])
    if not np.array_equal(g, r, equal_nan=True):
        fail("translate", k, dt.__name__, (n, h, w), strat, dx, dy, int((g != r).sum()))
    if dt == np.uint16:
        rows = int(rng.integers(1, h + 1))
        sh = np.stack([np.full(n, dx, np.float32), np.full(n, dy, np.float32)], axis=1)
        g = D.remove_motion(t, torch.from_numpy(sh).cuda(), rows=rows).cpu().numpy()
        r = np.stack([O.remove_motion(img[i], np.float32(dx), np.float32(dy), rows=rows) for i in range(n)])
        if not np.array_equal(g, r):
            fail("remove_motion", k, (n, h, w), rows, dx, dy, int((g != r).sum()))
        if h >= 3 and w >= 3:
            g = D.median_filter(t).cpu().numpy()
            r = np.stack([O.median_filter(img[i]) for i in range(n)])
            if not np.array_equal(g, r):
                fail("median", k, (n, h, w), int((g != r).sum()))
        p = float(rng.choice([0, 0.1, 0.5, 0.9, 1.0]))
        m = (rng.random((n, h, w)) < 0.6).astype(np.uint8)
        g = D.find_median_pixel(t, p).cpu().numpy().tolist()
        r = [O.find_median_pixel(img[i], p) for i in range(n)]
        gm = D.find_median_pixel(t, p, torch.from_numpy(m).cuda()).cpu().numpy().tolist()
        rm = [O.find_median_pixel(img[i], p, m[i]) for i in range(n)]
        if g != r or gm != rm:
            fail("find_median", k, (n, h, w), p, g, r, gm, rm)
        if strat in ("nearest", "background"):  # the fused chain against its three kernels (bit-identical)
            sig = float(rng.choice([0.3, 0.75, 1.0, 1.49, 2.0, 2.4]))
            bp = D.BadPixels(t[0]) if rng.random() < 0.7 else None
            a = bp.correct(t) if bp is not None else t
            ref = D.translate_to_u16(D.gaussian_filter(a, sig), (dx, dy), strat, background=7)
            out = D.filter_chain(t, bp, sig, (dx, dy), strat, background=7)
            if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
                fail("filter_chain", k, (n, h, w), sig, strat, dx, dy, int((out.view(torch.int16) != ref.view(torch.int16)).sum()))
        Y, U, V = D.split_planes(t, w + int(rng.integers(0, 9)))
        if not torch.equal(D.merge_planes(Y, U, V, w).view(torch.int16), t.view(torch.int16)):
            fail("planes", k, (n, h, w))
    if dt == np.float32:
        sig = float(rng.choice([0.3, 0.6, 0.99, 1.0, 1.7, 2.0, 2.4]))
        pos = np.abs(img) + 1
        g = D.gaussian_filter(torch.from_numpy(pos).cuda(), sig).cpu().numpy()
        r = np.stack([O.gaussian_filter(pos[i], sig) for i in range(n)])
        if not np.allclose(g, r, rtol=1e-5, atol=0):
            fail("gaussian", k, (n, h, w), sig, float(np.abs(g - r).max()))
print("soak: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
