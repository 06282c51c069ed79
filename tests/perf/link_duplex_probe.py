"""Do the two directions of the link work at the same time? (GPU box)  One chunk of 50 frames 640x512: the bare copies up and down, alone and
together on two streams; the chunk's encode reading its frames from page-locked host memory and the decode writing its frames there (what the
per-frame entry points run), alone and together."""
import ctypes as ct
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

L_ = D._lib
n, h, w, gop = 50, 512, 640, 50
fr = s1_noisy_background(n, h, w)
L = D.codec_layout(w, h, n, gop)
dev = torch.device("cuda")


def bufs(pin):
    kw = dict(pin_memory=True) if pin else dict(device=dev)
    return dict(hdr=torch.zeros((L.nchunks, L.ntiles, L.gop), dtype=torch.int64, **kw), toff=torch.zeros((L.nchunks, L.ntiles + 1), dtype=torch.int32, **kw),
                coff=torch.zeros((L.nchunks + 1,), dtype=torch.int64, **kw), stream=torch.zeros((L.stream_max_bytes // 8,), dtype=torch.int64, **kw))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ws1 = torch.empty((L.workspace_bytes,), dtype=torch.uint8, device=dev)
err = torch.zeros((1,), dtype=torch.int32, device=dev)
f_pin = torch.from_numpy(fr).pin_memory()
o_pin = torch.empty(f_pin.shape, dtype=f_pin.dtype, pin_memory=True)
f_dev, o_dev = torch.from_numpy(fr).to(dev), torch.empty((n, h, w), dtype=torch.uint16, device=dev)
assert f_pin.is_pinned() and o_pin.is_pinned()
b_enc, b_dec = bufs(True), bufs(True)


def enc(stream, frames, b):
    assert L_.rir_codec_encode_device(frames.data_ptr(), w, h, n, gop, b["hdr"].data_ptr(), b["toff"].data_ptr(), b["coff"].data_ptr(), b["stream"].data_ptr(),
                                      ws1.data_ptr(), L.workspace_bytes, ct.c_void_p(stream.cuda_stream)) == 0, D.last_error()


def dec(stream, b, out):
    assert L_.rir_codec_decode_device(b["hdr"].data_ptr(), b["toff"].data_ptr(), b["coff"].data_ptr(), b["stream"].data_ptr(), b["stream"].numel(), w, h, n, gop,
                                      out.data_ptr(), err.data_ptr(), ct.c_void_p(stream.cuda_stream)) == 0, D.last_error()


enc(s1, f_pin, b_dec)  # the decoder's input: a chunk's tables and stream in host memory
torch.cuda.synchronize()


def wall(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def up():
    with torch.cuda.stream(s1):
        f_dev.copy_(f_pin, non_blocking=True)


def down():
    with torch.cuda.stream(s2):
        o_pin.copy_(o_dev, non_blocking=True)


raw = fr.nbytes
a, b, c = wall(up), wall(down), wall(lambda: (up(), down()))
print("copy up %.0f us, copy down %.0f us, both at once %.0f us (sum %.0f, the longer %.0f)" % (a, b, c, a + b, max(a, b)), flush=True)
a, b = wall(lambda: enc(s1, f_pin, b_enc)), wall(lambda: dec(s2, b_dec, o_pin))
c = wall(lambda: (enc(s1, f_pin, b_enc), dec(s2, b_dec, o_pin)))
print("encode from host %.0f us, decode to host %.0f us, both at once on two streams %.0f us (sum %.0f, the longer %.0f)" % (a, b, c, a + b, max(a, b)), flush=True)
assert np.array_equal(o_pin.numpy(), fr) and int(err.item()) == 0
# the same with the decoder's frames going to device memory and a copy call bringing them down (engine instead of kernel stores)
c2 = wall(lambda: (enc(s1, f_pin, b_enc), dec(s2, b_dec, o_dev), down()))
print("encode from host + decode to HBM + copy down: %.0f us" % c2)

# the same two kernels on streams that own disjoint halves of the compute units (hipExtStreamCreateWithCUMask)
hip = ct.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ct.POINTER(ct.c_void_p), ct.c_uint32, ct.POINTER(ct.c_uint32)]


class RawStream:
    def __init__(self, words):
        self.h = ct.c_void_p()
        arr = (ct.c_uint32 * len(words))(*words)
        r = hip.hipExtStreamCreateWithCUMask(ct.byref(self.h), len(words), arr)
        assert r == 0, r
        self.cuda_stream = self.h.value


for name, ma, mb in (("low half / high half", [0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
                     ("even / odd compute units", [0x55555555] * 8, [0xAAAAAAAA] * 8),
                     ("a quarter each", [0xFFFFFFFF] * 2 + [0] * 6, [0] * 6 + [0xFFFFFFFF] * 2)):
    ra, rb = RawStream(ma), RawStream(mb)
    a, b = wall(lambda: enc(ra, f_pin, b_enc)), wall(lambda: dec(rb, b_dec, o_pin))
    c = wall(lambda: (enc(ra, f_pin, b_enc), dec(rb, b_dec, o_pin)))
    print("masked streams, %-26s: encode %.0f us, decode %.0f us, both at once %.0f us" % (name, a, b, c), flush=True)
assert np.array_equal(o_pin.numpy(), fr) and int(err.item()) == 0

# a kernel's host traffic one way, a copy call the other way
hip.hipMemcpyAsync.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_void_p]


def copy_down(stream):
    assert hip.hipMemcpyAsync(o_pin.data_ptr(), o_dev.data_ptr(), raw, 2, ct.c_void_p(stream.cuda_stream)) == 0  # hipMemcpyDeviceToHost


def copy_up(stream):
    assert hip.hipMemcpyAsync(f_dev.data_ptr(), f_pin.data_ptr(), raw, 1, ct.c_void_p(stream.cuda_stream)) == 0  # hipMemcpyHostToDevice


a, b = wall(lambda: enc(s1, f_pin, b_enc)), wall(lambda: copy_down(s2))
c = wall(lambda: (enc(s1, f_pin, b_enc), copy_down(s2)))
print("encode from host %.0f us + hipMemcpyAsync down %.0f us: together %.0f us" % (a, b, c), flush=True)
a, b = wall(lambda: copy_up(s1)), wall(lambda: dec(s2, b_dec, o_pin))
c = wall(lambda: (copy_up(s1), dec(s2, b_dec, o_pin)))
print("hipMemcpyAsync up %.0f us + decode to host %.0f us: together %.0f us" % (a, b, c), flush=True)
