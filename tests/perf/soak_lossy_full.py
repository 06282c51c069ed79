"""Soak at FULL frame size (640x512, 1-32 streams a call: also the launches that take 4 and 8 pixels per thread, which small test frames never do): the streaming
forms of the bounded-loss step (constant budgets, speculative) against the general form alone (the resident kernel: code of its own) - device against
device, so that many streams and seeds fit a run; both forms are held to the oracle by the GPU tests.
    python tests/perf/soak_lossy_full.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
h, w = 512, 640
bad = 0
books = np.zeros(4, np.int64)
for r in range(rounds):
    S = int(rng.choice([1, 2, 4, 5, 7, 8, 9, 12, 17, 24, 32]))  # (1-3 streams: 2 pixels per thread, 4-7: 4, 8 and more: 8)
    n = int(rng.integers(45, 100))
    hl = int(rng.choice([h, h - 3, h - 64]))
    const = bool(rng.integers(0, 3) == 0)
    add = bool(rng.integers(0, 2))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    prm, ins = [], []
    for i in range(S):
        bg = torch.rand((h, w), generator=g, device="cuda") * float(rng.choice([200, 1000, 30000])) + 10
        fr = (bg[None] + float(rng.choice([0.3, 0.7])) * torch.randn((n, h, w), generator=g, device="cuda")).clamp(0, 65535)
        for j in rng.integers(1, n, int(rng.integers(0, 3))):  # an event or two
            fr[int(j):] += float(rng.integers(1, 200))
        ins.append(fr.clamp(0, 65535).to(torch.int32).to(torch.uint16))
        prm.append((int(rng.integers(1, 10)), int(rng.integers(0, 5)), 0.0 if const else float(rng.choice([0.5, 2.5, 5.0])), int(rng.choice([0, 1, 3, 8, 32, 64])),
                    bool(rng.integers(0, 2))))
    cuts = sorted(set([0, 1, n] + [int(c) for c in rng.integers(2, n, int(rng.integers(0, 2)))]))

    def run(general):
        for k in ("RIR_LOSSY_NO_SPEC", "RIR_LOSSY_NO_CONST"):
            os.environ.pop(k, None)
            if general:
                os.environ[k] = "1"
        st = [D.LossyStream(w, h, hl, p[0], p[1], p[2], p[3], subtract_min=p[4]) for p in prm]
        outs, los, his = [[] for _ in range(S)], [[] for _ in range(S)], [[] for _ in range(S)]
        bk = np.zeros(4, np.int64)
        for c0, c1 in zip(cuts[:-1], cuts[1:]):
            o, lo, hi = D.LossyStream.step_many(st, [t[c0:c1] for t in ins], add_loss=add and c0 > 0)
            bk += np.array(st[0].spec_stats())
            for i in range(S):
                outs[i].append(o[i]), los[i].append(lo[i]), his[i].append(hi[i])
        for x in st:
            x.close()
        return [torch.cat(o) for o in outs], [np.concatenate(x) for x in los], [np.concatenate(x) for x in his], bk

    a, b = run(False), run(True)
    books += a[3]
    wrong = [i for i in range(S) if not torch.equal(a[0][i].view(torch.int16), b[0][i].view(torch.int16)) or not np.array_equal(a[1][i], b[1][i]) or not np.array_equal(a[2][i], b[2][i])]
    if wrong:
        bad += 1
        print("FAIL round", r, dict(S=S, n=n, hl=hl, const=const, add=add, cuts=cuts), "streams", wrong, flush=True)
    print("round %d: %d streams x %d frames, %s, %d failures so far" % (r, S, n, "constant budgets" if const else "budgets that follow the statistics", bad), flush=True)
print("soak (full size): %d rounds, %d failures; through the speculative launches %d, offered %d, committed %d, passes %d" % ((rounds, bad) + tuple(int(x) for x in books)))
sys.exit(1 if bad else 0)
