"""Development aid: how the time of the packing kernel (and of the whole two-launch step) depends on WHERE the frames and the
encoder workspace sit relative to each other, with both carved out of ONE device allocation (so that the distance in virtual
addresses is also the distance inside one physically contiguous run, as far as the driver allows).

    python tests/perf/placement_probe.py [arena GiB]      -> gpurun_out/placement_probe.json
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
arena_gib = float(sys.argv[1]) if len(sys.argv) > 1 else 14.0
src = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
arena = torch.empty(int(arena_gib * (1 << 30)), dtype=torch.uint8, device="cuda")
ctx = D.CodecContext(w, h, n, 50)
WS = ctx.layout.workspace_bytes
FB = src.numel() * 2
frames = arena[:FB].view(torch.uint16).view(n, h, w)
frames.copy_(src)
out_own = torch.empty_like(src)


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


def at(delta):
    """workspace `delta` bytes after the start of the frames"""
    ctx.workspace = arena[delta:delta + WS]
    t_pack = timed(lambda: ctx.encode_tiles(frames))
    t_dec = timed(lambda: ctx.decode_slots(out=out_own, check=False))
    t_step = timed(lambda: (ctx.encode_tiles(frames), ctx.decode_slots(out=out_own, check=False)))
    return {"delta": delta, "pack_us": t_pack, "decode_us": t_dec, "step_us": t_step}


res = {"frames_bytes": FB, "workspace_bytes": WS, "arena_ptr": arena.data_ptr(), "coarse": [], "fine": {}}
base = (FB + 4095) // 4096 * 4096
limit = arena.numel() - WS
d = base
while d < limit:
    res["coarse"].append(at(d))
    print(res["coarse"][-1], flush=True)
    d += 256 << 20
for name, step_b, count in (("4K", 4096, 16), ("64K", 65536, 16), ("2M", 2 << 20, 16), ("32M", 32 << 20, 16)):
    res["fine"][name] = [at(base + k * step_b) for k in range(count)]
    print(name, [round(r["pack_us"], 1) for r in res["fine"][name]], flush=True)
ok = torch.equal(out_own.view(torch.int16), src.view(torch.int16))
res["bit_exact"] = bool(ok)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "placement_probe.json"), "w"))
print("bit exact:", ok)
