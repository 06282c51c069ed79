"""Development aid: do the frame-buffer kernels - an input and an output of equal size walked at the same pace - see the placement classes
too?  256 frames 640x512 in one 2 GiB allocation, the output in each of 12 others; translate, gaussian (u16 in), fused chain, median."""
import ctypes as ct
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.low_level.misc import _lib  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 256, 512, 640
src = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
bufs = [torch.empty(2 << 30, dtype=torch.uint8, device="cuda") for _ in range(13)]
fin = bufs[0][:src.numel() * 2].view(torch.uint16).view(n, h, w)
fin.copy_(src)
off = torch.tensor([1.25, -2.5], dtype=torch.float32, device="cuda")
back = np.zeros(1, np.uint16)
st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=9):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


rows = {"translate": [], "chain": [], "median": [], "gaussian_u16": []}
for b in bufs[1:]:
    o16 = b[:src.numel() * 2].view(torch.uint16).view(n, h, w)
    o32 = b[:src.numel() * 4].view(torch.float32).view(n, h, w)
    rows["translate"].append(timed(lambda: _lib.rir_translate_device(ord("H"), fin.data_ptr(), o16.data_ptr(), w, h, n, off.data_ptr(), 0, back.ctypes.data, b"nearest", st)))
    rows["chain"].append(timed(lambda: _lib.rir_filter_chain_device(0, fin.data_ptr(), o16.data_ptr(), w, h, n, ct.c_float(0.75), off.data_ptr(), 0, back.ctypes.data, b"nearest", st)))
    rows["median"].append(timed(lambda: _lib.rir_median_filter_device(fin.data_ptr(), o16.data_ptr(), w, h, n, st)))
    rows["gaussian_u16"].append(timed(lambda: _lib.rir_gaussian_filter_u16_device(fin.data_ptr(), o32.data_ptr(), w, h, n, ct.c_float(0.75), st)))
for k, v in rows.items():
    print("%-13s us per 256 frames, output in 12 other allocations: %s" % (k, [round(x) for x in v]))
