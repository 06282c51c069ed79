"""Connected components of a batch of images in device memory (development aid, GPU box):
    python tests/perf/label_batch_time.py [frames]
label_images / keep_largest_areas on N x 640x512 uint16 images of a few regions each, beside one image per call; three images against the
oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h, w = 512, 640
rng = np.random.default_rng(3)
O = Oracle()
frames = np.stack([np.kron(rng.integers(0, 3, (h // 16, w // 16)) * (rng.random((h // 16, w // 16)) < 0.4), np.ones((16, 16), dtype=np.int64)) for _ in range(n)]).astype(np.uint16)
t = torch.from_numpy(frames).cuda()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


lab, area, xy, count = D.label_images(t, 0)
for i in (0, n // 2, n - 1):
    exp = O.label_image(frames[i], 0)
    k = int(count[i])
    assert k == exp[1].size and np.array_equal(lab[i].cpu().numpy(), exp[0]) and np.array_equal(area[i, :k].cpu().numpy(), exp[1])
us = timed(lambda: D.label_images(t, 0))
print("label_images, %d images a call       : %8.1f us a call = %5.2f us an image (%.2f TB/s of the 6 bytes a cell in and out)" % (n, us, us / n, n * h * w * 6 / us / 1e6))
us = timed(lambda: D.keep_largest_areas(t, 0, 1))
print("keep_largest_areas, %d images a call : %8.1f us a call = %5.2f us an image" % (n, us, us / n))
us = timed(lambda: [D.keep_largest_area(t[i], 0, 1) for i in range(16)], 3) / 16
print("keep_largest_area, one image a call   : %8.1f us an image" % us)
