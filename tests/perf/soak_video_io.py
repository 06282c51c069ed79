"""One-off soak of the saver / loader pipelines of round 5 (chunks in flight, read-ahead lanes, kernels on page-locked memory) against
what was handed in (not collected by pytest):
   python tests/perf/soak_video_io.py [rounds] [seed]
Every round: a random geometry (ragged sizes included), GOP, length and mix of add_image / add_image_lossy calls (stdFactor 0: the
bounded-loss frames are checked against the oracle's step, the others must come back as they went in), per-frame attributes; then the
file is read in a random pattern of sequential runs, jumps and repeats, with the bad-pixel read-back filter switched on and off in
between (filtered images against the oracle's repair)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402
from librir_amd.video_io import rir_video_io as rv  # noqa: E402
from oracle.pyoracle import Oracle, OracleLossy  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
O = Oracle()
fails = 0
with tempfile.TemporaryDirectory() as tmp:
    for r in range(rounds):
        h = int(rng.integers(8, 200))
        w = int(rng.integers(8, 260))
        if rng.random() < 0.3:
            h, w = 8 * (h // 8 + 1), 8 * (w // 8 + 1)
        hl = h if rng.random() < 0.5 else max(1, h - 3)
        gop = int(rng.choice([1, 2, 5, 8, 50, 60]))
        n = int(rng.integers(1, 6 * min(gop, 30) + 7))
        fr = inject_bad_pixels(s1_noisy_background(n, h, w, seed=1000 * seed + r), 5) if rng.random() < 0.5 else s1_noisy_background(n, h, w, seed=1000 * seed + r)
        lossy_mode = rng.choice(["none", "all", "mixed"], p=[0.5, 0.25, 0.25])
        is_lossy = np.zeros(n, bool) if lossy_mode == "none" else (np.ones(n, bool) if lossy_mode == "all" else rng.random(n) < 0.4)
        err = int(rng.integers(1, 5))
        ra = int(rng.choice([0, 1, 4, 32]))
        p = os.path.join(tmp, "s%d.h264" % r)
        L = OracleLossy(O, w, h, hl, low_err=err, high_err=err, std_factor=0.0, running_average=ra)
        exp = fr.copy()
        with IRSaver(p, w, h, hl) as s:
            s.set_parameter("GOP", gop)
            s.set_parameter("lowValueError", err)
            s.set_parameter("highValueError", err)
            s.set_parameter("stdFactor", 0)
            s.set_parameter("runningAverage", ra)
            for i in range(n):
                if is_lossy[i]:
                    exp[i] = L.step(fr[i])
                    s.add_image_lossy(fr[i], i * 1000, attributes={"k": str(i)})
                else:
                    s.add_image(fr[i], i * 1000, attributes={"k": str(i)})
            if is_lossy.any():
                assert len(s.get_low_errors()) == int(is_lossy.sum())
        cam = rv.open_camera_file(p)
        ok = rv.get_image_count(cam) == n
        xy = O.bad_pixels_detect(exp[0][: h - 3]) if h > 6 else []
        order = []
        while len(order) < 3 * n + 10:
            a = int(rng.integers(0, n))
            order += list(range(a, min(n, a + int(rng.integers(1, 2 * gop + 3)))))
        filt = False
        for k, i in enumerate(order):
            if h > 6 and len(xy) and rng.random() < 0.05:
                filt = not filt
                rv.enable_bad_pixels(cam, filt)
            got = rv.load_image(cam, i)
            want = O.remove_bad_pixels(exp[i], xy, rows=h - 3) if filt else exp[i]
            if not np.array_equal(got, want):
                ok = False
                print("round %d: %dx%d gop %d n %d lossy %s ra %d: image %d (read %d, filter %s) differs" % (r, w, h, gop, n, lossy_mode, ra, i, k, filt), flush=True)
                break
        rv.close_camera(cam)
        with IRMovie.from_filename(p) as mov:
            j = int(rng.integers(0, n))
            mov.load_pos(j)
            ok = ok and mov.frame_attributes.get("k") == str(j).encode()
        os.remove(p)
        fails += 0 if ok else 1
        if r % 10 == 9:
            print("round %d, %d failures so far" % (r + 1, fails), flush=True)
print("soak_video_io: %d rounds, %d failures" % (rounds, fails))
sys.exit(1 if fails else 0)
