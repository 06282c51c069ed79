"""Development aid: time the codec step on the headline workload (HIP events, median of K):
    python tests/perf/enc_ab.py [frames]      (RIR_SINGLE_PASS=1: the single-pass look-back encoder instead of the two-pass one;
                                               RIR_DIAG_STATS=1 with a -DRIR_DIAG_LB_STATS build: its per-workgroup timeline)
Prints encode / decode / step times in the pipeline and alone."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
h, w = int(os.environ.get("RIR_H", "512")), int(os.environ.get("RIR_W", "640"))
fr = s1_noisy_background(n, h, w)
if os.environ.get("RIR_NOISE"):
    fr = (fr.astype(np.int64) + np.random.default_rng(1).integers(0, int(os.environ["RIR_NOISE"]), fr.shape)).astype(np.uint16)
t = torch.from_numpy(fr).cuda()
ctx = D.CodecContext(w, h, n, int(os.environ.get("RIR_GOP", "50")))
SP = bool(os.environ.get("RIR_SINGLE_PASS"))
out = torch.empty_like(t)
for _ in range(5):
    enc = ctx.encode(t, single_pass=SP)
    ctx.decode(enc, out=out, check=False)
torch.cuda.synchronize()
K = 20
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
for k in range(K):
    ev[k][0].record()
    enc = ctx.encode(t, single_pass=SP)
    ev[k][1].record()
    ctx.decode(enc, out=out, check=False)
    ev[k][2].record()
torch.cuda.synchronize()
med = lambda a, b: float(np.median([ev[k][a].elapsed_time(ev[k][b]) for k in range(K)])) * 1e3
te, td = med(0, 1), med(1, 2)
rt = bool(torch.equal(out.view(torch.int16), t.view(torch.int16)))
print("%s: encode %.1f us  decode %.1f us  step %.1f us  fps %.0f  ratio %.3f  roundtrip %s  err %d" %
      ("single-pass" if SP else "two-pass", te, td, te + td, n / ((te + td) * 1e-6), fr.nbytes / enc.compressed_bytes(), rt, int(ctx.error.item())))


def alone(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e3


if os.environ.get("RIR_DIAG_STATS"):
    L = ctx.layout
    ng = (L.ntiles + 63) // 64
    ctrl = 8 * (8 * 16 + 16 + L.nchunks + 1 + L.nchunks * ng + ((L.nchunks * L.ntiles + 15) & ~15) + L.nchunks * ng + L.nchunks + 16)
    ctrl = (ctrl + 255) & ~255
    gop = L.gop
    slot = gop * 128 + 16
    ws = ctx.workspace[ctrl:ctrl + L.nchunks * L.ntiles * slot * 8].view(torch.int64).view(L.nchunks * L.ntiles, slot)
    T = ws[:, gop * 128:gop * 128 + 6].cpu().numpy().astype(np.float64)
    wait, t1, t0, t2, t3 = T[:, 0] / 100, T[:, 1] / 100, T[:, 2] / 100, T[:, 3] / 100, T[:, 4] / 100
    base = t0.min()
    t0, t1, t2, t3 = t0 - base, t1 - base, t2 - base, t3 - base
    print("per workgroup (us): walk mean %.1f p90 %.1f | look-back wait mean %.2f median %.2f p90 %.2f max %.2f | copy mean %.2f | ends at %.1f" %
          ((t1 - t0).mean(), np.percentile(t1 - t0, 90), wait.mean(), np.median(wait), np.percentile(wait, 90), wait.max(), (t3 - t2).mean(), t3.max()))
    wd = t1 - t0
    print("  walk duration percentiles: p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f" % tuple(np.percentile(wd, [50, 90, 99, 99.9, 100])))
    first = t0 < 5
    print("  first round (%d workgroups): walk p50 %.1f p90 %.1f p99 %.1f max %.1f; lookback start max %.1f; copy start min %.1f median %.1f" %
          (first.sum(), *np.percentile(wd[first], [50, 90, 99, 100]), t1[first].max(), t2[first].min(), np.median(t2[first])))
    xcc = (T[:, 5].astype(np.int64) >> 32) & 15
    for lo in range(0, 2560, 256):
        sl = slice(lo, lo + 256)
        print("    segs %4d..%4d: start %.1f  walk end mean %.1f max %.1f  copy start mean %.1f  end mean %.1f" %
              (lo, lo + 255, t0[sl].mean(), t1[sl].mean(), t1[sl].max(), t2[sl].mean(), t3[sl].mean()))
    cu = T[:, 5].astype(np.int64) & 0xffffffff
    for x in range(0):
        m = xcc == x
        print("    xcc %d: %5d workgroups  walk mean %.1f p90 %.1f  wait mean %.1f  copy mean %.2f  last end %.1f  seg%%8 %s" %
              (x, m.sum(), wd[m].mean(), np.percentile(wd[m], 90), wait[m].mean(), (t3 - t2)[m].mean(), t3[m].max(), sorted(set((np.nonzero(m)[0] % 8).tolist()))))
    order = np.argsort(-wd)[:4]
    for i in order:
        print("    slow: seg %5d (chunk %2d tile %3d) xcc %d  start %.1f  walk %.1f  wait %.1f" % (i, i // L.ntiles, i % L.ntiles, xcc[i], t0[i], wd[i], wait[i]))
    edges = np.arange(0, t3.max() + 10, 10.0)
    for lo in edges:
        hi = lo + 10
        mid = lo + 5
        walking = ((t0 <= mid) & (t1 > mid)).sum()
        waiting = ((t1 <= mid) & (t2 > mid)).sum()
        copying = ((t2 <= mid) & (t3 > mid)).sum()
        started = ((t0 >= lo) & (t0 < hi)).sum()
        dur = (t1 - t0)[(t0 >= lo) & (t0 < hi)]
        print("  t=%5.0f us: walking %4d  waiting %4d  copying %4d  started %4d  walk dur of those %.1f" % (mid, walking, waiting, copying, started, dur.mean() if dur.size else 0))
if not SP:
    print("alone: packing %.1f us  scan + gather %.1f us" % (alone(lambda: ctx.encode_tiles(t)), alone(lambda: ctx.encode_compact())))
print("alone: encode %.1f us  decode %.1f us" % (alone(lambda: ctx.encode(t, single_pass=SP)), alone(lambda: ctx.decode(enc, out=out, check=False))))
sys.exit(0 if rt else 1)
