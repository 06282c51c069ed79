"""Feasibility probe (GPU box): the codec kernels working straight on page-locked HOST memory over PCIe, one chunk (50 frames 640x512):
encode reading its frames from the host / writing its tables and stream to the host, decode writing its frames to the host.
    python tests/perf/zero_copy_probe.py"""
import ctypes as ct
import os
import sys

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


def bufs(where):
    kw = dict(device=dev) if where == "dev" else dict(pin_memory=True)
    return dict(hdr=torch.zeros((L.nchunks, L.ntiles, L.gop), dtype=torch.int64, **kw), toff=torch.zeros((L.nchunks, L.ntiles + 1), dtype=torch.int32, **kw),
                coff=torch.zeros((L.nchunks + 1,), dtype=torch.int64, **kw), stream=torch.zeros((L.stream_max_bytes // 8,), dtype=torch.int64, **kw))


ws = torch.empty((L.workspace_bytes,), dtype=torch.uint8, device=dev)
err = torch.zeros((1,), dtype=torch.int32, device=dev)
f_dev = torch.from_numpy(fr).to(dev)
f_pin = torch.from_numpy(fr).pin_memory()
o_dev = torch.empty_like(f_dev)
o_pin = torch.empty(f_pin.shape, dtype=f_pin.dtype, pin_memory=True)
assert f_pin.is_pinned() and o_pin.is_pinned()  # (a pageable pointer handed to a kernel is a GPU memory fault)
st = lambda: ct.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def enc(frames, b):
    r = L_.rir_codec_encode_device(frames.data_ptr(), w, h, n, gop, b["hdr"].data_ptr(), b["toff"].data_ptr(), b["coff"].data_ptr(), b["stream"].data_ptr(),
                                   ws.data_ptr(), L.workspace_bytes, st())
    assert r == 0, D.last_error()


def dec(b, out):
    r = L_.rir_codec_decode_device(b["hdr"].data_ptr(), b["toff"].data_ptr(), b["coff"].data_ptr(), b["stream"].data_ptr(), b["stream"].numel(), w, h, n, gop,
                                   out.data_ptr(), err.data_ptr(), st())
    assert r == 0, D.last_error()


def timed(fn, name, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    m = float(np.median(ts))
    print("%-62s %8.1f us  (%5.1f GB/s of the bytes that cross the link)" % (name, m, nbytes / m / 1e3), flush=True)


bd, bp = bufs("dev"), bufs("pin")
assert all(t.is_pinned() for t in bp.values())
raw = fr.nbytes
enc(f_dev, bd)
torch.cuda.synchronize()
comp = int(bd["coff"][-1].item()) * 8 + L.hdr_bytes + L.tile_off_bytes
print("chunk: %d raw bytes, %d compressed (tables included)" % (raw, comp))
timed(lambda: enc(f_dev, bd), "encode: frames HBM, outputs HBM", raw)
timed(lambda: enc(f_pin, bd), "encode: frames HOST, outputs HBM", raw)
timed(lambda: enc(f_pin, bp), "encode: frames HOST, outputs HOST", raw + comp)
timed(lambda: enc(f_dev, bp), "encode: frames HBM, outputs HOST", comp)
torch.cuda.synchronize()
for k in ("hdr", "toff", "coff"):
    assert torch.equal(bp[k], bd[k].cpu()), k
nw = int(bd["coff"][-1].item())
assert torch.equal(bp["stream"][:nw], bd["stream"][:nw].cpu())
print("encode to host == encode to HBM")
timed(lambda: dec(bd, o_dev), "decode: stream HBM, frames HBM", raw)
timed(lambda: dec(bd, o_pin), "decode: stream HBM, frames HOST", raw)
timed(lambda: dec(bp, o_pin), "decode: stream HOST, frames HOST", raw + comp)
timed(lambda: dec(bp, o_dev), "decode: stream HOST, frames HBM", comp)
torch.cuda.synchronize()
assert np.array_equal(o_pin.numpy(), fr) and int(err.item()) == 0
print("decode to host == input")
timed(lambda: f_dev.copy_(f_pin, non_blocking=True), "hipMemcpyAsync H2D of the chunk (for comparison)", raw)
timed(lambda: o_pin.copy_(f_dev, non_blocking=True), "hipMemcpyAsync D2H of the chunk (for comparison)", raw)

# ---- the frame-buffer kernels on one image in page-locked host memory (the per-frame signal_processing entry points) ----
img = torch.from_numpy(fr[0].copy())
f32 = img.to(torch.float32)


def pinned(t):
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    assert p.is_pinned()
    return p


shift_dev = torch.tensor([1.25, -2.5], dtype=torch.float32, device=dev)
bg = np.zeros(1, np.float64)
for name, src in (("u16", img), ("f32", f32)):
    s_dev, s_pin = src.to(dev), pinned(src)
    d_dev, d_pin = torch.empty_like(s_dev), pinned(torch.zeros_like(src))
    ch = D._DTYPE_CHARS[src.dtype].encode()
    nbytes = src.numel() * src.element_size()

    def tr(a, b):
        bgv = np.zeros(1, D._NP_OF[src.dtype])
        r = L_.rir_translate_device(ord(ch), a.data_ptr(), b.data_ptr(), w, h, 1, shift_dev.data_ptr(), 0, bgv.ctypes.data, b"nearest", st())
        assert r == 0, D.last_error()

    timed(lambda: tr(s_dev, d_dev), "translate %s: HBM -> HBM" % name, 2 * nbytes)
    timed(lambda: tr(s_pin, d_dev), "translate %s: HOST -> HBM" % name, nbytes)
    timed(lambda: tr(s_dev, d_pin), "translate %s: HBM -> HOST" % name, nbytes)
    timed(lambda: tr(s_pin, d_pin), "translate %s: HOST -> HOST" % name, 2 * nbytes)
    torch.cuda.synchronize()
    assert torch.equal(d_pin, d_dev.cpu())
    timed(lambda: (s_dev.copy_(s_pin, non_blocking=True), tr(s_dev, d_dev), d_pin.copy_(d_dev, non_blocking=True)), "translate %s: copy in, HBM -> HBM, copy out" % name,
          2 * nbytes)
s_dev, s_pin = f32.to(dev), pinned(f32)
d_dev, d_pin = torch.empty_like(s_dev), pinned(torch.zeros_like(f32))


def ga(a, b):
    assert L_.rir_gaussian_filter_device(a.data_ptr(), b.data_ptr(), w, h, 1, ct.c_float(0.75), st()) == 0, D.last_error()


nb = f32.numel() * 4
timed(lambda: ga(s_dev, d_dev), "gaussian f32: HBM -> HBM", 2 * nb)
timed(lambda: ga(s_pin, d_pin), "gaussian f32: HOST -> HOST", 2 * nb)
timed(lambda: (s_dev.copy_(s_pin, non_blocking=True), ga(s_dev, d_dev), d_pin.copy_(d_dev, non_blocking=True)), "gaussian f32: copy in, HBM -> HBM, copy out", 2 * nb)
torch.cuda.synchronize()
assert torch.equal(d_pin, d_dev.cpu())
