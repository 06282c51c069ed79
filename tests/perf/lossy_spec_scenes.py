"""Which scenes the speculative form of the bounded-loss step commits, and in how many passes: 640x512, the reference's default parameters
(6 / 2 / stdFactor 5 / 32), one call of N frames per scene, every result checked against the oracle; both ways a pass can correct the table
(every budget behind the first wrong one / the first wrong one only) and the general form alone beside them.
    python tests/perf/lossy_spec_scenes.py [frames, default 400] [passes allowed, default 8]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from oracle.pyoracle import Oracle, OracleLossy  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
allowed = int(sys.argv[2]) if len(sys.argv) > 2 else 8
h, w, hl = 512, 640, 509
O = Oracle()
rng = np.random.default_rng(1234)
bg = rng.random((h, w)) * 1000
yy, xx = np.mgrid[0:h, 0:w]
blob = np.exp(-((xx - 300) ** 2 + (yy - 250) ** 2) / 5000.0)


def frames(fn):
    return np.stack([np.clip(fn(i), 0, 65535).astype(np.uint16) for i in range(n)])


scenes = {
    "static, noise 0.7": lambda i: bg + 10 + rng.normal(0, np.sqrt(0.5), (h, w)),
    "static, noise 2": lambda i: bg + 10 + rng.normal(0, 2, (h, w)),
    "weak flash at 150": lambda i: bg + 10 + rng.normal(0, 0.7, (h, w)) + (5 if i == 150 else 0) * (blob > 0.5),
    "step of 50 levels at 150": lambda i: bg + 10 + (50 if i >= 150 else 0) + rng.normal(0, 0.7, (h, w)),
    "drift 0.05 level / frame": lambda i: bg + 10 + 0.05 * i + rng.normal(0, 0.7, (h, w)),
    "drift 0.2 level / frame": lambda i: bg + 10 + 0.2 * i + rng.normal(0, 0.7, (h, w)),
    "blob heating": lambda i: 1000 + blob * (500 + 2 * i) + rng.normal(0, 1.0, (h, w)),
    "S1 (1 level / frame)": None,
}
print("%d frames of 640x512 per scene, %d passes allowed; frames off the guess (oracle) | every budget corrected: committed, passes, ms | the same, 16 passes and no giving up | first only: committed, passes, ms | general form ms" % (n, allowed))
for name, fn in scenes.items():
    arr = s1_noisy_background(n, h, w) if fn is None else frames(fn)
    L = OracleLossy(O, w, h, hl, 6, 2, 5.0, 32)
    exp, errs = [], []
    for i in range(n):
        exp.append(L.step(arr[i]))
        errs.append(L.last_errors()[:2])
    exp = np.stack(exp)
    moved = sum(1 for e in errs if e != (6, 2))
    t = torch.from_numpy(arr).cuda()
    row = []
    for env in ({"RIR_LOSSY_SPEC_PASSES": str(allowed)}, {"RIR_LOSSY_SPEC_PASSES": "16", "RIR_LOSSY_SPEC_NO_GIVE_UP": "1"},
                {"RIR_LOSSY_SPEC_PASSES": str(allowed), "RIR_LOSSY_SPEC_FIRST_ONLY": "1"}, {"RIR_LOSSY_NO_SPEC": "1"}):
        for k in ("RIR_LOSSY_SPEC_PASSES", "RIR_LOSSY_SPEC_FIRST_ONLY", "RIR_LOSSY_NO_SPEC", "RIR_LOSSY_SPEC_NO_GIVE_UP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 32)
        ls.step(t[:1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got, lo, hi = ls.step(t[1:])
        ms = (time.perf_counter() - t0) * 1e3
        ok = np.array_equal(got.cpu().numpy(), exp[1:]) and list(zip(lo.tolist(), hi.tolist())) == errs[1:]
        st = ls.spec_stats()
        ls.close()
        row.append("%s %d %5.2f%s" % ("yes" if st[2] else "no ", st[3], ms, "" if ok else " DIFFERS FROM THE ORACLE"))
    print("%-26s %4d | %s | %s | %s | %s" % (name, moved, row[0], row[1], row[2], row[3]), flush=True)
