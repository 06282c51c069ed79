"""One-off soak: the per-frame C ABI driven from several Python threads at once, each with its own handles (ctypes releases
the GIL during the calls) - savers, cameras, the host-pointer filters, the labelling, the time-axis helper and the registration entry (not collected by pytest).
   python tests/perf/soak_threads.py [threads] [rounds]"""
import os
import sys
import tempfile
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.registration import find_transform_ecc_translation  # noqa: E402
from librir_amd.signal_processing import rir_signal_processing as sp  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
O = Oracle()
errors = []


def worker(k, tmp):
    r = -1
    try:
        rng = np.random.default_rng(k)
        h, w = 40 + 8 * k, 64 + 16 * k
        for r in range(rounds):
            if k == 0 and r % 20 == 19:
                print("round %d of %d, %d failures so far" % (r + 1, rounds, len(errors)), flush=True)
            fr = s1_noisy_background(23, h, w, seed=100 * k + r)
            p = os.path.join(tmp, "t%d_%d.h264" % (k, r))
            with IRSaver(p, w, h, h) as s:
                s.set_parameter("GOP", 5)
                for i in range(len(fr)):
                    s.add_image(fr[i], i * 1000)
            with IRMovie.from_filename(p) as mov:
                for i in rng.permutation(len(fr)):
                    if not np.array_equal(mov[int(i)], fr[i]):
                        errors.append(("roundtrip", k, r, int(i)))
            dx, dy = float(rng.normal(0, 3)), float(rng.normal(0, 3))
            if not np.array_equal(sp.translate(fr[0], dx, dy, "nearest"), O.translate(fr[0], dx, dy, "nearest")):
                errors.append(("translate", k, r))
            g = sp.gaussian_filter(fr[1].astype(np.float32), 0.75)
            if not np.allclose(g, O.gaussian_filter(fr[1].astype(np.float32), 0.75), rtol=1e-5, atol=0):
                errors.append(("gaussian", k, r))
            regions = ((fr[3] >> 5) % 3).astype(np.uint16)
            got, exp = sp.label_image(regions, 0), O.label_image(regions, 0)
            if not all(np.array_equal(x, y) for x, y in zip(got, exp)) or not np.array_equal(sp.keep_largest_area(regions, 0, 4), O.keep_largest_area(regions, 0, 4)):
                errors.append(("labelling", k, r))
            ts_a, ts_b = np.sort(rng.integers(0, 50, 20)) * 0.5, np.sort(rng.integers(0, 50, 30)) * 0.5
            if not np.array_equal(sp.extract_times((ts_a, ts_b), "union"), O.extract_times((ts_a, ts_b), 0)[1]):
                errors.append(("extract_times", k, r))
            a = sp.gaussian_filter(fr[2].astype(np.float32), 2.0)  # (white noise has no basin of attraction: smooth it)
            a = (a - a.min()) / (a.max() - a.min())
            b = np.roll(a, (1, 2), axis=(0, 1))
            try:
                cc, warp = find_transform_ecc_translation(a, b, np.eye(2, 3, dtype=np.float32), 50, 1e-6, None)
            except RuntimeError as e:  # (diagnosis: was it the input - the filter's result - or the alignment, and does it repeat?)
                g2 = O.gaussian_filter(fr[2].astype(np.float32), 2.0)
                a_ok = bool(np.allclose((g2 - g2.min()) / (g2.max() - g2.min()), a, rtol=1e-4, atol=1e-6))
                try:
                    find_transform_ecc_translation(a, b, np.eye(2, 3, dtype=np.float32), 50, 1e-6, None)
                    again = "a second call on the same images converged"
                except RuntimeError:
                    again = "a second call failed too"
                errors.append(("ecc-exception", k, r, "input == oracle: %s" % a_ok, again, repr(e)[:80]))
                continue
            if not (abs(warp[0, 2] - 2) < 0.3 and abs(warp[1, 2] - 1) < 0.3):
                errors.append(("ecc", k, r, float(warp[0, 2]), float(warp[1, 2])))
    except Exception as e:  # noqa: BLE001
        errors.append(("exception", k, r, repr(e)))


with tempfile.TemporaryDirectory() as tmp:
    ts = [threading.Thread(target=worker, args=(k, tmp)) for k in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
for e in errors[:10]:
    print("FAIL", e)
print("soak: %d threads x %d rounds, %d failures" % (nthreads, rounds, len(errors)))
sys.exit(1 if errors else 0)
