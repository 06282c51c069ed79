"""Quick codec timing + correctness on the GPU box (development aid, not a test):
    python tests/perf/codec_time.py [frames]
Prints per-kernel HIP-event times (median of 10) for the headline workload and checks the
round trip and, on small cases, bit-equality of the stream with the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background, s2_uniform_dl_ti  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

O = Oracle()
ok_all = True
if not os.environ.get("RIR_SKIP_CHECK"):
    for (n, h, w, gop, kind) in [(5, 67, 83, 3, "rand"), (53, 48, 64, 50, "s1"), (130, 16, 32, 128, "ramp"), (60, 64, 64, 50, "s2"), (20, 64, 80, 7, "wide")]:
        rng = np.random.default_rng(n)
        if kind == "rand":
            fr = rng.integers(0, 65536, (n, h, w)).astype(np.uint16)
        elif kind == "s1":
            fr = s1_noisy_background(n, h, w)
        elif kind == "s2":
            fr = s2_uniform_dl_ti(n, h, w)
        elif kind == "wide":
            fr = (s1_noisy_background(n, h, w).astype(np.int64) + rng.integers(0, 300, (n, h, w))).astype(np.uint16)
        else:
            fr = (np.cumsum(np.ones((n, h, w), np.uint32), axis=2) + np.arange(n)[:, None, None]).astype(np.uint16)
        ctx = D.CodecContext(w, h, n, gop)
        t = torch.from_numpy(fr).cuda()
        enc = ctx.encode(t)
        dec = ctx.decode(enc)
        torch.cuda.synchronize()
        ok = np.array_equal(dec.cpu().numpy(), fr)
        L = ctx.layout
        hdr = enc.hdr.cpu().numpy().view(np.uint64)
        toff = enc.tile_off.cpu().numpy().view(np.uint32)
        coff = enc.chunk_off.cpu().numpy()
        st = enc.stream.cpu().numpy().view(np.uint64)
        same = True
        for c in range(L.nchunks):
            f0 = c * gop
            nf = min(gop, n - f0)
            h_o, o_o, st_o = O.codec_encode_chunk(fr[f0:f0 + nf])
            same &= bool(np.array_equal(hdr[c][:, :nf], h_o) and np.array_equal(toff[c], o_o) and np.array_equal(st[coff[c]:coff[c + 1]], st_o))
        print(("PASS" if ok and same else "FAIL"), kind, n, h, w, gop, "roundtrip", ok, "stream==oracle", same)
        ok_all &= ok and same

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
h, w = 512, 640
fr = s1_noisy_background(n, h, w)
t = torch.from_numpy(fr).cuda()
ctx = D.CodecContext(w, h, n, int(os.environ.get("RIR_GOP", "50")))
out = torch.empty_like(t)
for _ in range(3):
    enc = ctx.encode(t)
    ctx.decode(enc, out=out, check=False)
torch.cuda.synchronize()
K = 10
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
for k in range(K):
    ev[k][0].record()
    ctx.encode_tiles(t)
    ev[k][1].record()
    enc = ctx.encode_compact()
    ev[k][2].record()
    ctx.decode(enc, out=out, check=False)
    ev[k][3].record()
torch.cuda.synchronize()
med = lambda a, b: float(np.median([ev[k][a].elapsed_time(ev[k][b]) for k in range(K)])) * 1e3
te, tc, td = med(0, 1), med(1, 2), med(2, 3)
rt = bool(torch.equal(out.view(torch.int16), t.view(torch.int16)))
print("encode_tiles %.1f us  compact %.1f us  decode %.1f us  total %.1f us  fps %.0f  ratio %.3f  roundtrip %s  err %d" %
      (te, tc, td, te + tc + td, n / ((te + tc + td) * 1e-6), fr.nbytes / enc.compressed_bytes(), rt, int(ctx.error.item())))
RC = 0 if (ok_all and (rt or os.environ.get("RIR_DIAG"))) else 1

# ---- each kernel alone, back to back with itself (no dirty data of the other kernels in L2 / Infinity Cache) ----
def alone(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


print("alone: encode_tiles %.1f us  compact %.1f us  decode %.1f us" %
      (alone(lambda: ctx.encode_tiles(t)), alone(lambda: ctx.encode_compact()), alone(lambda: ctx.decode(enc, out=out, check=False))))
sys.exit(RC)
