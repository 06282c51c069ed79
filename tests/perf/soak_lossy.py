"""One-off soak: random parameters of the bounded-loss step, device stream operator against the oracle.
    python tests/perf/soak_lossy.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D
from oracle.pyoracle import Oracle, OracleLossy
O = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for k in range(cases):
    h, w = int(rng.integers(4, 50)), int(rng.integers(8, 120))
    hl = int(rng.integers(1, h + 1))
    n = int(rng.integers(2, 70))
    low, high = int(rng.integers(0, 12)), int(rng.integers(0, 8))
    sf = float(rng.choice([0, 0.5, 5, 20]))
    ra = int(rng.choice([0, 1, 2, 5, 32, 64, 100]))
    smin = bool(rng.integers(0, 2))
    add = bool(rng.integers(0, 2))
    noise = float(rng.choice([0.5, 2, 10]))
    bg = rng.random((h, w)) * rng.choice([50, 1000, 60000])
    fr = np.clip(bg[None] + np.arange(n)[:, None, None] * rng.integers(0, 3) + rng.normal(0, noise, (n, h, w)), 0, 65535).astype(np.uint16)
    L = OracleLossy(O, w, h, hl, low_err=low, high_err=high, std_factor=sf, running_average=ra, subtract_min=smin)
    exp, elo, ehi = [], [], []
    for i in range(n):
        exp.append(L.step(fr[i], add_loss=add and i > 0))
        lo, hi, _ = L.last_errors(); elo.append(lo); ehi.append(hi)
    ls = D.LossyStream(w, h, hl, low, high, sf, ra, subtract_min=smin)
    t = torch.from_numpy(fr).cuda()
    a, la, ha = ls.step(t[:1]); b, lb, hb = ls.step(t[1:], add_loss=add) if n > 1 else (a[:0], la[:0], ha[:0])
    got = torch.cat([a, b]).cpu().numpy()
    ok = np.array_equal(got, np.stack(exp)) and np.concatenate([la, lb]).tolist() == elo and np.concatenate([ha, hb]).tolist() == ehi
    if not ok:
        bad += 1
        print("FAIL", k, dict(h=h, w=w, hl=hl, n=n, low=low, high=high, sf=sf, ra=ra, smin=smin, add=add), int((got != np.stack(exp)).sum()))
    ls.close()
print("soak: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
