"""Development aid: do the frame-buffer kernels care where their output sits relative to their input?  translate (u16 -> u16),
gaussian (f32 -> f32) and the codec's decoder (stream -> frames), each with its destination at a series of distances."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.device import _lib, _stream  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
src = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
srcf = src[:500].float()
off = torch.tensor([1.25, -2.5], dtype=torch.float32, device="cuda")
back = np.zeros(1, np.uint16)
GB = float(1 << 30)


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


keep = []
for k in range(14):
    dst = torch.empty_like(src)
    dstf = torch.empty_like(srcf)
    t_tr = timed(lambda: _lib.rir_translate_device(ord("H"), src.data_ptr(), dst.data_ptr(), w, h, n, off.data_ptr(), 0, back.ctypes.data, b"nearest", _stream()))
    t_g = timed(lambda: _lib.rir_gaussian_filter_device(srcf.data_ptr(), dstf.data_ptr(), w, h, 500, 0.75, _stream()))
    print("dst at %+7.2f GB: translate u16 %.1f us   | dstf at %+7.2f GB: gaussian f32 (500 frames) %.1f us" %
          ((dst.data_ptr() - src.data_ptr()) / GB, t_tr, (dstf.data_ptr() - srcf.data_ptr()) / GB, t_g))
    keep += [dst, dstf, torch.empty(1200 << 20, dtype=torch.uint8, device="cuda")]
