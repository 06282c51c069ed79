"""One-off soak: objects of every kind created and released many times - device memory and the handle table must come back
to where they started (not collected by pytest).   python tests/perf/soak_handles.py [rounds]"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402
from librir_amd.video_io import rir_video_io as rv  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, h, w = 12, 96, 128
fr = inject_bad_pixels(s1_noisy_background(n, h, w), 20)
t = torch.from_numpy(fr).cuda()


def one_round(tmp, k):
    bp = D.BadPixels(t[0])
    out = D.filter_chain(t, bp, 0.75, (1.25, -2.5), "nearest")
    ctx = D.CodecContext(w, h, n, 5)
    dec = ctx.decode(ctx.encode(out))
    assert torch.equal(dec.view(torch.int16), out.view(torch.int16))
    ls = D.LossyStream(w, h, h - 3, 3, 3, 5.0, 8)
    ls.step(t)
    ls.close()
    reg = DeviceRegistratorECC(1, 1, shape=(h, w))
    reg.start(t[0])
    reg.compute(t[1])
    p = os.path.join(tmp, "f%d.h264" % (k % 3))
    with IRSaver(p, w, h, h - 3) as s:
        s.set_parameter("GOP", 5)
        for i in range(n):
            (s.add_image_lossy if i % 2 else s.add_image)(fr[i], i * 1000)
    with IRMovie.from_filename(p) as mov:
        mov.bad_pixels_correction = True
        assert mov[3].shape == (h, w)
    z = os.path.join(tmp, "z%d.bin" % (k % 3))
    zw = rv.open_video_write(z, w, h, method=rv.METHOD_ZSTD)
    rv.image_write(zw, fr[0], 0)
    rv.close_video(zw)
    cam = rv.open_camera_file(z)
    assert np.array_equal(rv.load_image(cam, 0), fr[0])
    rv.close_camera(cam)
    del bp, ctx, reg, out, dec
    return cam  # the last handle number handed out


with tempfile.TemporaryDirectory() as tmp:
    for k in range(3):
        one_round(tmp, k)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    h0 = one_round(tmp, 0)
    for k in range(rounds):
        hk = one_round(tmp, k)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
print("free device memory: %+.1f MiB over %d rounds; last handle %d -> %d" % ((free1 - free0) / 2**20, rounds, h0, hk))
bad = abs(free1 - free0) > (64 << 20) or hk > h0 + 2
print("soak: %d rounds, %s" % (rounds, "LEAK" if bad else "0 failures"))
sys.exit(1 if bad else 0)
