"""Connected-component labelling (development aid, GPU box):
    python tests/perf/label_time.py [h w]
label_image / keep_largest_area on images of different structure - the labels against the oracle, the kernels' time on an image in
device memory and the C entry point's time per host image, beside the oracle's (the reference's algorithm on one core)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd import signal_processing as sp  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
O = Oracle()
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:h, 0:w]


def smooth_noise():
    a = rng.normal(size=(h // 16 + 2, w // 16 + 2))
    a = np.kron(a, np.ones((16, 16)))[:h, :w]
    for _ in range(3):
        a = (a + np.roll(a, 5, 0) + np.roll(a, 5, 1) + np.roll(a, -5, 0) + np.roll(a, -5, 1)) / 5
    return a


def spiral():
    img = np.zeros((h, w), np.uint16)
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


cases = {
    "blobs (thresholded smooth noise)": (smooth_noise() > 0.4).astype(np.uint16),
    "hot regions, 3 levels": np.digitize(smooth_noise(), [0.2, 0.5, 0.8]).astype(np.uint16),
    "all background": np.zeros((h, w), np.uint16),
    "one flat component": np.ones((h, w), np.uint16),
    "vertical stripes, distinct values": ((xx % 7) + 1).astype(np.uint16),
    "horizontal stripes": ((yy % 2)).astype(np.uint16),
    "checkerboard": ((xx + yy) % 2).astype(np.uint16),
    "noise, 2 values": rng.integers(0, 2, (h, w)).astype(np.uint16),
    "noise, 5 values": rng.integers(0, 5, (h, w)).astype(np.uint16),
    "spiral": spiral(),
    "float32 levels with NaN": np.where(rng.random((h, w)) < 0.01, np.nan, np.digitize(smooth_noise(), [0.3, 0.6])).astype(np.float32),
}
import gc  # noqa: E402

gc.collect()
gc.freeze()  # (a full collection of Python's garbage collector is 35 ms: one call of a 20-call loop)
fails = 0
for name, img in cases.items():
    t0 = time.perf_counter()
    lab_o, area_o, xy_o = O.label_image(img, 0)
    t_or = time.perf_counter() - t0
    keep_o = O.keep_largest_area(img, 0, 7)
    lab, area, xy = sp.label_image(img, 0)
    keep = sp.keep_largest_area(img, 0, 7)
    ok = np.array_equal(lab, lab_o) and np.array_equal(area, area_o) and np.array_equal(xy, xy_o) and np.array_equal(keep, keep_o)
    fails += not ok
    t = torch.from_numpy(img).cuda()
    # the launches alone: outputs and working memory allocated once, no read-back of the count between calls
    lib, ch = D._lib, ord(D._DTYPE_CHARS[t.dtype])
    need = lib.rir_label_workspace_bytes(w, h)
    work = torch.empty(need // 8 + 1, dtype=torch.int64, device="cuda")
    dst = torch.empty((h, w), dtype=torch.int32, device="cuda")
    xyb = torch.empty((h * w + 1, 2), dtype=torch.float64, device="cuda")
    ab = torch.empty(h * w + 1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    back = np.zeros(1, dtype=img.dtype)
    st = torch.cuda.current_stream().cuda_stream

    def run_label():
        assert lib.rir_label_image_device(ch, t.data_ptr(), dst.data_ptr(), w, h, back.ctypes.data, xyb.data_ptr(), ab.data_ptr(), cnt.data_ptr(),
                                          work.data_ptr(), work.numel() * 8, st) == 0

    def run_keep():
        assert lib.rir_keep_largest_area_device(ch, t.data_ptr(), dst.data_ptr(), w, h, back.ctypes.data, 7, work.data_ptr(), work.numel() * 8, st) == 0

    reps = 20
    times = []
    for f in (run_label, run_keep):
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / reps * 1e3)
    assert np.array_equal(dst.cpu().numpy(), keep_o)
    t0 = time.perf_counter()
    for _ in range(reps):
        sp.label_image(img, 0)
    t_abi = (time.perf_counter() - t0) / reps * 1e6
    print("%-36s %s  components %6d | device label %7.1f us, keep_largest %7.1f us | C entry %7.1f us | oracle (1 core) %8.1f us"
          % (name, "ok " if ok else "DIFFERS", len(area_o) - 1, times[0], times[1], t_abi, t_or * 1e6), flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
