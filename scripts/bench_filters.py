#!/usr/bin/env python3
"""Device-resident throughput of the frame-buffer kernels (SURVEY §8d bytes) + config-3 pipeline.
   python scripts/bench_filters.py [--frames 256]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background, inject_bad_pixels

p = argparse.ArgumentParser(); p.add_argument("--frames", type=int, default=256); p.add_argument("--reps", type=int, default=10)
a = p.parse_args()
n, h, w = a.frames, 512, 640
fr = inject_bad_pixels(s1_noisy_background(n, h, w), 200)
t16 = torch.from_numpy(fr).cuda(); f32 = t16.to(torch.float32)
WH = h * w
def timeit(fn, reps=a.reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {}
def rec(name, ms, bytes_per_frame):
    res[name] = {"ms": ms, "fps": n / ms * 1e3, "GBs": bytes_per_frame * n / ms / 1e6, "frac_of_8TBs": bytes_per_frame * n / ms / 1e6 / 8000}
    print("%-34s %8.3f ms  %10.0f fps  %7.1f GB/s  %5.1f %% of HBM peak" % (name, ms, res[name]["fps"], res[name]["GBs"], 100 * res[name]["frac_of_8TBs"]))
offs = torch.tensor([1.25, -2.5], dtype=torch.float32, device="cuda")
rec("translate u16 nearest", timeit(lambda: D.translate(t16, offs, "nearest")), 4 * WH)
rec("translate f32 nearest", timeit(lambda: D.translate(f32, offs, "nearest")), 8 * WH)
rec("translate u16 noborder", timeit(lambda: D.translate(t16, offs, "")), 4 * WH)
for s in (0.75, 1.0, 2.0):
    rec("gaussian sigma=%g" % s, timeit(lambda: D.gaussian_filter(f32, s)), 8 * WH)
bp = D.BadPixels(t16[0])
rec("bad_pixels_correct (%d px)" % bp.count, timeit(lambda: bp.correct(t16)), 4 * WH)
rec("find_median_pixel", timeit(lambda: D.find_median_pixel(t16, 0.5)), 2 * WH)
rec("median_filter 3x3", timeit(lambda: D.median_filter(t16)), 4 * WH)
sh = torch.zeros((n, 2), dtype=torch.float32, device="cuda"); sh[:, 0] = 1.25; sh[:, 1] = -2.5
rec("remove_motion (read-back)", timeit(lambda: D.remove_motion(t16, sh, rows=h - 3)), 4 * WH)
t0 = time.perf_counter(); bp2 = D.BadPixels(t16[0]); torch.cuda.synchronize(); print("bad_pixels_create (detector, once per stream): %.2f ms" % ((time.perf_counter() - t0) * 1e3))
# config 3: bad-pixel correct -> gaussian(0.75) -> translate(1.25,-2.5, nearest) -> encode (unfused chain, device-resident)
ctx = D.CodecContext(w, h, n, 50)
def pipeline():
    a_ = bp.correct(t16)
    g = D.gaussian_filter(a_, 0.75)
    return ctx.encode(D.translate_to_u16(g, offs, "nearest"))
rec("config3 chain + encode (4 kernels)", timeit(pipeline, 5), 2 * WH)
rec("filter_chain (3 filters fused)", timeit(lambda: D.filter_chain(t16, bp, 0.75, offs, "nearest")), 4 * WH)
rec("filter_chain + encode", timeit(lambda: ctx.encode(D.filter_chain(t16, bp, 0.75, offs, "nearest")), 5), 2 * WH)
for s in (0.75, 1.0, 2.0):
    rec("gaussian u16 in, sigma=%g" % s, timeit(lambda: D.gaussian_filter(t16, s)), 6 * WH)
print(json.dumps(res))
