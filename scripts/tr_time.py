"""translate / remove_motion timing only (device-resident, 256 frames 640x512).  python scripts/tr_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 256, 512, 640
t16 = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
f32 = t16.to(torch.float32)
offs = torch.tensor([1.25, -2.5], dtype=torch.float32, device="cuda")
sh = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
sh[:, 0] = 1.25
sh[:, 1] = -2.5


def timeit(name, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-28s %.3f ms" % (name, e0.elapsed_time(e1) / reps))


timeit("translate u16 nearest", lambda: D.translate(t16, offs, "nearest"))
timeit("translate f32 nearest", lambda: D.translate(f32, offs, "nearest"))
timeit("translate u16 background", lambda: D.translate(t16, offs, "background"))
timeit("translate u16 (0.5,0.5)", lambda: D.translate(t16, (0.5, 0.5), "nearest"))
timeit("remove_motion", lambda: D.remove_motion(t16, sh, rows=h - 3))
