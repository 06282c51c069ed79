"""Soak of label_image / keep_largest_area against the oracle (GPU box):
    python tests/perf/soak_labelling.py [iterations]
random geometries (1 .. 700 wide, 1 .. 600 high; now and then a 2000 x 1500 one), every cell type, images of every structure - blocks,
noise, thin mazes with long winding components (the deep forests), stripes, NaN cells - through the C entry points, every fourth
time the device layer and every eighth a batch of three images in one call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd import signal_processing as sp  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
O = Oracle()
rng = np.random.default_rng(20261005)
DT = [np.bool_, np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64, np.float32, np.float64]


def maze(h, w):
    """walls one pixel wide with random gaps: few components, each a long corridor system"""
    img = np.ones((h, w), np.int64)
    img[1::2, :] = 0
    for y in range(1, h, 2):
        k = max(1, w // 40)
        img[y, rng.integers(0, w, k)] = 1
    if rng.random() < 0.5:
        img = img.T.copy()[:h, :w] if h == w else img
    return img


fails = 0
t0 = time.time()
for it in range(iters):
    if it % 97 == 96:
        h, w = int(rng.integers(1200, 1600)), int(rng.integers(1500, 2100))
    else:
        h, w = int(rng.integers(1, 600)), int(rng.integers(1, 700))
    dt = np.dtype(DT[it % len(DT)])
    kind = it % 7
    levels = 1 if dt == np.bool_ else int(rng.integers(1, 5))
    if kind == 0:
        c = int(rng.integers(2, 24))
        img = np.kron(rng.integers(0, levels + 1, (h // c + 1, w // c + 1)), np.ones((c, c), np.int64))[:h, :w]
    elif kind == 1:
        img = rng.integers(0, levels + 1, (h, w))
    elif kind == 2:
        img = maze(h, w)
    elif kind == 3:
        img = (np.arange(w)[None, :] % int(rng.integers(1, 9)) + np.zeros((h, 1), np.int64)) % (levels + 1)
    elif kind == 4:
        img = (np.arange(h)[:, None] % int(rng.integers(1, 9)) + np.zeros((1, w), np.int64)) % (levels + 1)
    elif kind == 5:
        img = (rng.random((h, w)) < rng.random() * 0.9).astype(np.int64) * int(rng.integers(1, levels + 1))
    else:
        img = np.ones((h, w), np.int64) * (rng.random((h, w)) < 0.995)
    img = np.ascontiguousarray(img.astype(dt))
    if dt.kind == "f" and it % 3 == 0:
        img[rng.random((h, w)) < 0.01] = np.nan
    bg = 0 if dt == np.bool_ else int(rng.integers(0, 2))
    fg = int(rng.integers(-4, 9))
    exp = O.label_image(img, bg)
    exp_keep = O.keep_largest_area(img, bg, fg)
    if it % 8 == 5 and h * w < 200000:  # a batch on the device: this image, its mirror image and its negative mask, each against the oracle
        batch = np.ascontiguousarray(np.stack([img, img[:, ::-1], (img == bg).astype(dt)]))
        tb = torch.from_numpy(batch).cuda()
        labs, areas, xys, counts = D.label_images(tb, bg, table_entries=h * w + 1)
        keeps = D.keep_largest_areas(tb, bg, fg).cpu().numpy()
        for k in range(3):
            e = O.label_image(batch[k], bg)
            c = int(counts[k])
            if not (c == e[1].size and np.array_equal(labs[k].cpu().numpy(), e[0]) and np.array_equal(areas[k, :c].cpu().numpy(), e[1])
                    and np.array_equal(xys[k, :c].cpu().numpy(), e[2]) and np.array_equal(keeps[k], O.keep_largest_area(batch[k], bg, fg))):
                fails += 1
                print("DIFFERS (batch, image %d): iteration %d, %d x %d, %s, kind %d" % (k, it, h, w, dt, kind), flush=True)
    if it % 4 == 3:
        t = torch.from_numpy(img).cuda()
        got = tuple(x.cpu().numpy() for x in D.label_image(t, bg))
        keep = D.keep_largest_area(t, bg, fg).cpu().numpy()
    else:
        got = sp.label_image(img, bg)
        keep = sp.keep_largest_area(img, bg, fg)
    ok = all(np.array_equal(a, b) for a, b in zip(got, exp)) and np.array_equal(keep, exp_keep)
    if not ok:
        fails += 1
        print("DIFFERS: iteration %d, %d x %d, %s, kind %d" % (it, h, w, dt, kind), flush=True)
        np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "soak_labelling_fail_%d.npy" % it), img)
    if it % 500 == 499:
        print("%d iterations, %d failures, %.0f s" % (it + 1, fails, time.time() - t0), flush=True)
print("soak_labelling: %d iterations, %d failures" % (iters, fails))
sys.exit(1 if fails else 0)
