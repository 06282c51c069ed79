"""Development aid: does the packing kernel's time follow the FRAMES buffer, the WORKSPACE, or their distance?  Separate
allocations of 2 GiB each (hipMalloc through torch, nothing freed in between); frames in buffer i, workspace in buffer j -> a
matrix of times; then the frames at different offsets inside one buffer.

    python tests/perf/placement_probe2.py [nbuf]      -> gpurun_out/placement_probe2.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

n, h, w = 1000, 512, 640
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 10
src = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
ctx = D.CodecContext(w, h, n, 50)
WS = ctx.layout.workspace_bytes
FB = src.numel() * 2
bufs = [torch.empty(2 << 30, dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
out_own = torch.empty_like(src)
GB = float(1 << 30)
print("buffer addresses (GiB):", [round(b.data_ptr() / GB, 3) for b in bufs], "fresh frames at", round(src.data_ptr() / GB, 3),
      "fresh workspace at", round(ctx.workspace.data_ptr() / GB, 3), flush=True)


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


res = {"addresses": [b.data_ptr() for b in bufs], "fresh": {}, "matrix": [], "frame_offsets": [], "decode_matrix": []}
fresh_ws = ctx.workspace
res["fresh"]["pack_us"] = timed(lambda: ctx.encode_tiles(src))
res["fresh"]["decode_us"] = timed(lambda: ctx.decode_slots(out=out_own, check=False))
print("fresh allocations: pack %.1f decode %.1f" % (res["fresh"]["pack_us"], res["fresh"]["decode_us"]), flush=True)
for i in range(nbuf):
    fr = bufs[i][:FB].view(torch.uint16).view(n, h, w)
    fr.copy_(src)
    row, drow = [], []
    for j in range(nbuf):
        if i == j:
            ctx.workspace = bufs[j][FB + 4096:FB + 4096 + WS]
        else:
            ctx.workspace = bufs[j][:WS]
        row.append(timed(lambda: ctx.encode_tiles(fr)))
        drow.append(timed(lambda: ctx.decode_slots(out=out_own, check=False)))
    res["matrix"].append(row)
    res["decode_matrix"].append(drow)
    print("frames in %d: pack" % i, [round(x, 1) for x in row], " decode(ws j -> own out)", [round(x, 1) for x in drow], flush=True)
# frames at different offsets inside buffer 0, workspace in buffer 1 and in the fresh allocation
for off in (0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 128 << 20, 512 << 20, 1 << 30):
    if off + FB > bufs[0].numel():
        continue
    fr = bufs[0][off:off + FB].view(torch.uint16).view(n, h, w)
    fr.copy_(src)
    ctx.workspace = bufs[1][:WS]
    a = timed(lambda: ctx.encode_tiles(fr))
    ctx.workspace = fresh_ws
    b = timed(lambda: ctx.encode_tiles(fr))
    res["frame_offsets"].append({"offset": off, "ws_buf1_us": a, "ws_fresh_us": b})
    print("frames at buffer 0 + %d: workspace in buffer 1 %.1f, fresh workspace %.1f" % (off, a, b), flush=True)
ctx.workspace = fresh_ws
ctx.encode_tiles(src)
ctx.decode_slots(out=out_own)
print("bit exact:", torch.equal(out_own.view(torch.int16), src.view(torch.int16)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "placement_probe2.json"), "w"))
