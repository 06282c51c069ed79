"""Development aid: 9-stream loss runs (the run kernel's second form: 1 440 of the 1 536 places it has) beside a flood of ordinary kernels from
another thread - failures (a sticky status) and the longest call.   python tests/perf/lossy_flood_probe.py [calls]"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.synthetic import s1_noisy_background  # noqa: E402

h, w, m, S = 512, 640, 60, 9
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
stop = False
floods = [0]


def flood():
    with torch.cuda.stream(torch.cuda.Stream()):
        x = torch.from_numpy(s1_noisy_background(64, h, w)).cuda()
        while not stop:
            D.gaussian_filter(x, 0.75)
            floods[0] += 1
            if floods[0] % 8 == 0:
                torch.cuda.current_stream().synchronize()


fr = torch.from_numpy(s1_noisy_background(m, h, w)).cuda()
ref = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
outs_ref, lo_ref, hi_ref = D.LossyStream.step_many(ref, [fr.clone() for _ in range(S)], errors=True)
torch.cuda.synchronize()
th = threading.Thread(target=flood)
th.start()
time.sleep(0.2)
fails, longest = 0, 0.0
with torch.cuda.stream(torch.cuda.Stream()):
    for c in range(calls):
        streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
        t0 = time.perf_counter()
        try:
            outs, lo, hi = D.LossyStream.step_many(streams, [fr.clone() for _ in range(S)], errors=True)
            torch.cuda.current_stream().synchronize()
            ok = all(torch.equal(a, b) for a, b in zip(outs, outs_ref)) and (lo == lo_ref).all() and (hi == hi_ref).all()
            for s in streams:
                s.status()
        except RuntimeError:
            ok = False
        longest = max(longest, time.perf_counter() - t0)
        fails += 0 if ok else 1
        for s in streams:
            s.close()
stop = True
th.join()
print("9-stream loss calls beside a flood of %d gaussian_filter calls: %d calls, %d failures (results compared with an undisturbed run), longest %.1f ms" % (floods[0], calls, fails, longest * 1e3))
