"""Development aid: resident launches (7-stream bounded-loss runs, 8-sequence alignments) on one thread while another thread
keeps ordinary kernels (gaussian_filter batches) going on a stream of its own.  Counts calls that ran into a clock."""
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
count = [0]


def flood():
    with torch.cuda.stream(torch.cuda.Stream()):
        x = torch.from_numpy(s1_noisy_background(64, h, w)).cuda()
        while not stop:
            D.gaussian_filter(x, 0.75)
            count[0] += 1
            if count[0] % 8 == 0:
                torch.cuda.current_stream().synchronize()


th = threading.Thread(target=flood)
th.start()
time.sleep(0.5)
S, n = 8, 64
seqs = [torch.from_numpy(s3_registration(n, h, w, seed=99 + q)[0]).cuda() for q in range(S)]
fr = torch.from_numpy(s1_noisy_background(100, h, w)).cuda()
res = {"ecc": [0, 0, 0.0], "lossy": [0, 0, 0.0]}
with torch.cuda.stream(torch.cuda.Stream()):
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        rs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(S)]
        for q in range(S):
            rs[q].start(seqs[q][0])
        t0 = time.perf_counter()
        try:
            DeviceRegistratorECC.compute_many_multi(rs, [s[1:] for s in seqs])
        except RuntimeError:
            res["ecc"][1] += 1
        res["ecc"][0] += 1
        res["ecc"][2] = max(res["ecc"][2], time.perf_counter() - t0)
        streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(7)]
        t0 = time.perf_counter()
        try:
            D.LossyStream.step_many(streams, [fr] * 7)
        except RuntimeError:
            res["lossy"][1] += 1
        res["lossy"][0] += 1
        res["lossy"][2] = max(res["lossy"][2], time.perf_counter() - t0)
        for x in streams:
            x.close()
stop = True
th.join()
print("ordinary kernel calls beside: %d; resident calls [count, failures, longest s]: %s" % (count[0], res))
