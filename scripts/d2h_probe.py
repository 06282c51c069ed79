"""Development aid: what one frame costs on its way to the host - D2H into pageable memory, D2H into page-locked memory plus a
host copy, and the host copy alone (655 360 bytes)."""
import time

import numpy as np
import torch

n = 640 * 512
d = torch.arange(n, dtype=torch.int16, device="cuda")
pageable = torch.empty(n, dtype=torch.int16)
pinned = torch.empty(n, dtype=torch.int16).pin_memory()
user = np.empty(n, dtype=np.int16)
big_pinned = torch.empty((50, n), dtype=torch.int16).pin_memory()
dbig = torch.zeros((50, n), dtype=torch.int16, device="cuda")


def t(fn, reps=300):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("D2H pageable, blocking           %.1f us" % t(lambda: pageable.copy_(d)))
print("D2H pinned + sync                %.1f us" % t(lambda: (pinned.copy_(d, non_blocking=True), torch.cuda.synchronize())))
print("host copy pinned -> user         %.1f us" % t(lambda: np.copyto(user, pinned.numpy())))
print("D2H pinned + sync + host copy    %.1f us" % t(lambda: (pinned.copy_(d, non_blocking=True), torch.cuda.synchronize(), np.copyto(user, pinned.numpy()))))
print("chunk of 50 frames D2H pinned    %.1f us per frame" % (t(lambda: (big_pinned.copy_(dbig, non_blocking=True), torch.cuda.synchronize()), 20) / 50))
