"""Packed form against the slotted and the dense form on the headline workload (development aid):
    python tests/perf/packed_time.py [frames]
HIP-event medians of the encode and decode kernels of each form and of K back-to-back steps, the round trip checked."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
h, w, gop = 512, 640, 50
fr = s1_noisy_background(n, h, w)
t = torch.from_numpy(fr).cuda()
out = torch.empty_like(t)
ctx = D.CodecContext(w, h, n, gop)
pc = D.PackedCodec(w, h, n, gop)
K = 20


def timed(fn_a, fn_b, name):
    for _ in range(3):
        fn_a()
        fn_b()
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    for k in range(K):
        ev[k][0].record()
        fn_a()
        ev[k][1].record()
        fn_b()
        ev[k][2].record()
    torch.cuda.synchronize()
    a = float(np.median([ev[k][0].elapsed_time(ev[k][1]) for k in range(K)])) * 1e3
    b = float(np.median([ev[k][1].elapsed_time(ev[k][2]) for k in range(K)])) * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(5):
        e0.record()
        for _ in range(K):
            fn_a()
            fn_b()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / K * 1e3)
    print("%-8s encode %.1f us  decode %.1f us  step (no events) min %.1f median %.1f us -> %.2f M frames/s" %
          (name, a, b, min(best), float(np.median(best)), n / float(np.median(best))), flush=True)


if os.environ.get("PACKED_ONLY"):
    timed(lambda: pc.encode(t), lambda: pc.decode(out=out, check=False), "packed one_cursor=%s lds=%s" % (os.environ.get("RIR_PACKED_ONE_CURSOR"), os.environ.get("RIR_ENC_LDS_WORDS")))
    print("round trip", bool(torch.equal(out.view(torch.int16), t.view(torch.int16))), pc.status())
    sys.exit(0)
timed(lambda: ctx.encode_tiles(t), lambda: ctx.decode_slots(out=out, check=False), "slotted")
assert torch.equal(out.view(torch.int16), t.view(torch.int16))
out.zero_()
timed(lambda: pc.encode(t), lambda: pc.decode(out=out, check=False), "packed")
batch = pc.finish()
assert torch.equal(out.view(torch.int16), t.view(torch.int16))
print("packed batch: %d bytes (payload %d) for %d raw bytes = 1/%.2f; stream capacity %d, workspace %d" %
      (batch.nbytes(), batch.payload_bytes(), fr.nbytes, fr.nbytes / batch.nbytes(), pc.stream.numel() * 8, pc.workspace.numel()))
out.zero_()


def dense_enc():
    ctx.encode_tiles(t)
    return ctx.encode_compact()


enc = dense_enc()
timed(dense_enc, lambda: ctx.decode(enc, out=out, check=False), "dense")
assert torch.equal(out.view(torch.int16), t.view(torch.int16))
