"""Soak of the multi-sequence alignment (ecc_run_multi_kernel: pairs of sequences, service workgroups, the next chunk's pre-processing
under the resident launch): random numbers of sequences, window sizes, lengths, chunk sizes and input types - every sequence's track from
compute_many_multi must equal, to the last bit, the track its own compute_many gives.   python tests/perf/soak_ecc_multi.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import s3_registration  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
fails = 0
for case in range(cases):
    S = int(rng.integers(1, 13))
    h, w = int(rng.integers(96, 300)), int(rng.integers(128, 400))
    n = int(rng.integers(3, 70))
    chunk = int(rng.integers(2, 40))
    fh, fv = float(rng.uniform(0.55, 1.0)), float(rng.uniform(0.55, 1.0))
    seqs = []
    for q in range(S):
        f, _ = s3_registration(n, h, w, seed=int(rng.integers(1, 1 << 30)))
        f = f.copy()
        if rng.random() < 0.3 and n > 30:
            k = int(rng.integers(24, n))
            f[k] += rng.normal(0, 4, f[k].shape).astype(np.float32)  # a confidence drop somewhere: a change of reference
        seqs.append(torch.from_numpy(f if rng.random() < 0.5 else np.clip(f, 0, 65535).astype(np.uint16)).cuda())
    try:
        solo = []
        for q in range(S):
            r = DeviceRegistratorECC(fh, fv, shape=(h, w))
            r.start(seqs[q][0])
            try:
                r.compute_many(seqs[q][1:], chunk=chunk)
            except RuntimeError:
                pass  # (a lost track ends a sequence: what was found up to there is compared)
            solo.append(r)
        multi = [DeviceRegistratorECC(fh, fv, shape=(h, w)) for _ in range(S)]
        for q in range(S):
            multi[q].start(seqs[q][0])
        try:
            DeviceRegistratorECC.compute_many_multi(multi, [s[1:] for s in seqs], chunk=chunk)
        except RuntimeError:
            pass
        # (compute_many_multi stops at the first sequence that raises: compare the common prefix, and demand at least what the shortest solo run has)
        ok = True
        for q in range(S):
            m = min(len(multi[q].x), len(solo[q].x))
            ok = ok and multi[q].x[:m] == solo[q].x[:m] and multi[q].y[:m] == solo[q].y[:m] and multi[q].confidences[:m] == solo[q].confidences[:m]
        lost = any(len(s.x) < n for s in solo)
        if not lost:
            ok = ok and all(len(multi[q].x) == n for q in range(S))
    except Exception as e:  # noqa: BLE001
        print("case %d: exception %r" % (case, e))
        ok = False
    if not ok:
        fails += 1
        print("case %d FAILED: S=%d %dx%d n=%d chunk=%d factors %.2f %.2f" % (case, S, w, h, n, chunk, fh, fv))
print("soak: %d cases, %d failures" % (cases, fails))
sys.exit(1 if fails else 0)
