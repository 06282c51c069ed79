"""Soak: random parameters of the bounded-loss step (every form: constant budgets, speculative with 1..8 passes, general; one stream a call, then
several), device stream operator against the oracle.
    python tests/perf/soak_lossy.py [cases] [seed] [longest run, default 70 frames]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D
from oracle.pyoracle import Oracle, OracleLossy
O = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 70  # (the constant-budget kernel has three phases: a run of 45 + ring length frames or more goes through all of them)
bad = 0
const_offered = const_taken = 0
spec = np.zeros(4, np.int64)  # groups through the speculative launches, offered, committed, passes
for k in range(cases):
    h, w = int(rng.integers(4, 50)), int(rng.integers(8, 120))
    if rng.integers(0, 2):
        w = (w + 7) // 8 * 8  # (runs of frames - the resident and the constant-budget kernels - need whole groups of 8 pixels)
    hl = int(rng.integers(1, h + 1))
    n = int(rng.integers(2, nmax))
    low, high = int(rng.integers(0, 12)), int(rng.integers(0, 8))
    sf = float(rng.choice([0, 0, 0, 0.5, 5, 20]))
    ra = int(rng.choice([0, 1, 2, 5, 32, 64, 100]))
    smin = bool(rng.integers(0, 2))
    add = bool(rng.integers(0, 2))
    noise = float(rng.choice([0.5, 2, 10]))
    bg = rng.random((h, w)) * rng.choice([50, 1000, 60000])
    fr = np.clip(bg[None] + np.arange(n)[:, None, None] * rng.integers(0, 3) + rng.normal(0, noise, (n, h, w)), 0, 65535).astype(np.uint16)
    if rng.integers(0, 4) == 0:  # flat scenes: few levels, frames whose foreground or background is empty (the statistic is 0 / 0 from frame 41 on)
        fr = (1000 + (fr >> int(rng.integers(6, 12)))).astype(np.uint16)
        for j in rng.integers(0, n, 3):
            fr[j] = 1000
    os.environ["RIR_LOSSY_SPEC_PASSES"] = str(int(rng.integers(1, 9)))
    os.environ.pop("RIR_LOSSY_SPEC_FIRST_ONLY", None)
    os.environ.pop("RIR_LOSSY_SPEC_NO_GIVE_UP", None)
    if rng.integers(0, 4) == 0:  # (the first form of the correction, one frame per pass; and passes that never give up)
        os.environ["RIR_LOSSY_SPEC_FIRST_ONLY"] = "1"
    if rng.integers(0, 4) == 0:
        os.environ["RIR_LOSSY_SPEC_NO_GIVE_UP"] = "1"
    os.environ.pop("RIR_LOSSY_SPEC_NO_PLANE", None)
    if rng.integers(0, 4) == 0:  # (the sums from the frames instead of the streaming kernel's byte plane)
        os.environ["RIR_LOSSY_SPEC_NO_PLANE"] = "1"
    L = OracleLossy(O, w, h, hl, low_err=low, high_err=high, std_factor=sf, running_average=ra, subtract_min=smin)
    exp, elo, ehi = [], [], []
    for i in range(n):
        exp.append(L.step(fr[i], add_loss=add and i > 0))
        lo, hi, _ = L.last_errors(); elo.append(lo); ehi.append(hi)
    ls = D.LossyStream(w, h, hl, low, high, sf, ra, subtract_min=smin)
    t = torch.from_numpy(fr).cuda()
    cuts = sorted(set([0, 1, n] + [int(c) for c in rng.integers(1, n + 1, int(rng.integers(0, 3)))]))
    parts = []
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        parts.append(ls.step(t[c0:c1], add_loss=add and c0 > 0))
        o_, t_ = ls.path_stats()
        const_offered += o_
        const_taken += t_
        spec += np.array(ls.spec_stats())
    got = torch.cat([p_[0] for p_ in parts]).cpu().numpy()
    ok = np.array_equal(got, np.stack(exp)) and np.concatenate([p_[1] for p_ in parts]).tolist() == elo and np.concatenate([p_[2] for p_ in parts]).tolist() == ehi
    if not ok:
        bad += 1
        print("FAIL", k, dict(h=h, w=w, hl=hl, n=n, low=low, high=high, sf=sf, ra=ra, smin=smin, add=add), int((got != np.stack(exp)).sum()))
    ls.close()
# ---- several streams in shared launches (LossyStream.step_many): parameters, scenes and ring lengths of their own, calls cut at the same frames;
# static scenes among them so that the speculative form commits some calls and hands others on (streams of a call go together), several calls
# per case so that the back-off runs
multi_cases, multi_bad = cases // 4, 0
for k in range(multi_cases):
    S = int(rng.integers(2, 5))
    h, w = int(rng.integers(4, 40)), int(rng.integers(1, 14)) * 8
    hl = int(rng.integers(1, h + 1))
    n = int(rng.integers(3, nmax))
    add = bool(rng.integers(0, 2))
    os.environ["RIR_LOSSY_SPEC_PASSES"] = str(int(rng.integers(1, 9)))
    os.environ.pop("RIR_LOSSY_SPEC_FIRST_ONLY", None)
    os.environ.pop("RIR_LOSSY_SPEC_NO_GIVE_UP", None)
    os.environ.pop("RIR_LOSSY_SPEC_NO_PLANE", None)
    all_const = bool(rng.integers(0, 3) == 0)
    calm = bool(rng.integers(0, 2))  # every stream a static scene (what the speculative form commits) or a mixture
    prm, frs, exp, elo, ehi = [], [], [], [], []
    for q in range(S):
        low, high = int(rng.integers(0, 12)), int(rng.integers(0, 8))
        sf = 0.0 if all_const else float(rng.choice([0, 0.5, 5, 5, 20]))
        ra = int(rng.choice([0, 1, 2, 5, 32, 64]))
        smin = bool(rng.integers(0, 2))
        prm.append((low, high, sf, ra, smin))
        bg = rng.random((h, w)) * rng.choice([50, 1000, 60000])
        if calm or rng.integers(0, 2):
            fr = np.repeat(np.clip(bg, 0, 65535).astype(np.uint16)[None], n, axis=0)
            for j in rng.integers(0, n, int(rng.integers(0, 3))):  # an event or two
                fr[j:] = (fr[j:].astype(np.int64) + int(rng.integers(1, 300))).clip(0, 65535).astype(np.uint16)
        else:
            fr = np.clip(bg[None] + rng.normal(0, float(rng.choice([0.5, 2, 10])), (n, h, w)), 0, 65535).astype(np.uint16)
        frs.append(fr)
        L = OracleLossy(O, w, h, hl, low_err=low, high_err=high, std_factor=sf, running_average=ra, subtract_min=smin)
        e, lo_, hi_ = [], [], []
        for i in range(n):
            e.append(L.step(fr[i], add_loss=add and i > 0))
            a_, b_, _ = L.last_errors(); lo_.append(a_); hi_.append(b_)
        exp.append(np.stack(e)); elo.append(lo_); ehi.append(hi_)
    streams = [D.LossyStream(w, h, hl, p_[0], p_[1], p_[2], p_[3], subtract_min=p_[4]) for p_ in prm]
    ts = [torch.from_numpy(f).cuda() for f in frs]
    cuts = sorted(set([0, 1, n] + [int(c) for c in rng.integers(1, n + 1, int(rng.integers(0, 4)))]))
    got = [[] for _ in range(S)]
    glo, ghi = [[] for _ in range(S)], [[] for _ in range(S)]
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        outs, lo, hi = D.LossyStream.step_many(streams, [t[c0:c1] for t in ts], add_loss=add and c0 > 0)
        spec += np.array(streams[0].spec_stats())
        for q in range(S):
            got[q].append(outs[q].cpu().numpy()); glo[q] += lo[q].tolist(); ghi[q] += hi[q].tolist()
    ok = all(np.array_equal(np.concatenate(got[q]), exp[q]) and glo[q] == elo[q] and ghi[q] == ehi[q] for q in range(S))
    if not ok:
        multi_bad += 1
        print("FAIL multi", k, dict(S=S, h=h, w=w, hl=hl, n=n, add=add, prm=prm, cuts=cuts))
    for x in streams:
        x.close()
bad += multi_bad
print("soak, several streams a call: %d cases, %d failures" % (multi_cases, multi_bad))
print("soak: %d cases, %d failures; groups offered to the constant-budget form %d, taken %d; through the speculative launches %d, offered %d, committed %d, passes %d"
      % (cases, bad, const_offered, const_taken, spec[0], spec[1], spec[2], spec[3]))
sys.exit(1 if bad else 0)
