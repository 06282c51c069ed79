"""Development aid: as resident_vs_ordinary_probe.py, for the two other resident kernels: 7-stream bounded-loss runs (lossy_run_kernel, 1 120 of
~1 200 places) and single-sequence alignments (ecc_run_kernel, 256 workgroups), many calls, beside a flood of ordinary kernels of two kinds."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import s1_noisy_background, s3_registration  # noqa: E402

h, w = 512, 640
stop = False


def flood(kind):
    with torch.cuda.stream(torch.cuda.Stream()):
        x = torch.from_numpy(s1_noisy_background(64, h, w)).cuda()
        k = 0
        while not stop:
            if kind == 0:
                D.gaussian_filter(x, 0.75)
            else:
                D.translate(x, (1.25, -2.5), "nearest")
            k += 1
            if k % 8 == 0:
                torch.cuda.current_stream().synchronize()


ths = [threading.Thread(target=flood, args=(k,)) for k in (0, 1)]
for t in ths:
    t.start()
time.sleep(0.5)
n = 64
seq = torch.from_numpy(s3_registration(n, h, w)[0]).cuda()
fr = torch.from_numpy(s1_noisy_background(100, h, w)).cuda()
res = {"ecc_solo": [0, 0, 0.0], "lossy_7": [0, 0, 0.0]}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with torch.cuda.stream(torch.cuda.Stream()):
    for rep in range(reps):
        r = DeviceRegistratorECC(1, 1, shape=(h, w))
        r.start(seq[0])
        t0 = time.perf_counter()
        try:
            r.compute_many(seq[1:])
        except RuntimeError:
            res["ecc_solo"][1] += 1
        res["ecc_solo"][0] += 1
        res["ecc_solo"][2] = max(res["ecc_solo"][2], time.perf_counter() - t0)
        streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(7)]
        t0 = time.perf_counter()
        try:
            D.LossyStream.step_many(streams, [fr] * 7)
        except RuntimeError:
            res["lossy_7"][1] += 1
        res["lossy_7"][0] += 1
        res["lossy_7"][2] = max(res["lossy_7"][2], time.perf_counter() - t0)
        for x in streams:
            x.close()
stop = True
for t in ths:
    t.join()
print("resident calls beside two floods of ordinary kernels [count, failures, longest s]: %s" % res)
