"""Development aid: tracked-sequence registration, S sequences side by side (compute_many_multi) against one (compute_many),
the reference's S3 recipe at 640x512, full-frame window.   python tests/perf/ecc_multi_time.py [S] [frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import s3_registration  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
h, w = 512, 640
seqs = [torch.from_numpy(s3_registration(n, h, w, seed=99 + q)[0]).cuda() for q in range(S)]


def solo():
    r = DeviceRegistratorECC(1, 1, shape=(h, w))
    r.start(seqs[0][0])
    r.compute_many(seqs[0][1:])
    return r


def multi(chunk=32):
    rs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(S)]
    for q in range(S):
        rs[q].start(seqs[q][0])
    DeviceRegistratorECC.compute_many_multi(rs, [s[1:] for s in seqs], chunk=chunk)
    return rs


def best(fn, count, reps=3):
    b = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        b = max(b, count / (time.perf_counter() - t0))
    return b


r1 = solo()
rm = multi()
same = all(rm[0].x == r1.x and rm[0].y == r1.y for _ in (0,))
its = None
print("S=%d slices=%s: solo %.1f k frames/s | multi %.1f k frames/s aggregate (chunk 32), %.1f k (chunk 99) | sequence 0 identical to its solo run: %s" %
      (S, os.environ.get("RIR_ECC_MULTI_SLICES", "auto"), best(solo, n - 1) / 1e3, best(multi, S * (n - 1)) / 1e3, best(lambda: multi(99), S * (n - 1)) / 1e3, same))

# where the time of a multi call goes: the library calls wrapped with a synchronising clock (distorts the overlap of the next
# chunk's preparation with the book-keeping, shows the parts)
from librir_amd.registration import device_registration as DR  # noqa: E402

acc = {}


def wrap(name):
    fn = getattr(DR._lib, name)

    def timed(*a):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r

    return timed


class Shim:
    def __init__(self, lib):
        self._lib = lib
        self.rir_ecc_align_multi_overlapped_device = wrap("rir_ecc_align_multi_overlapped_device")
        self.rir_ecc_prepare_frames_device = wrap("rir_ecc_prepare_frames_device")

    def __getattr__(self, k):
        return getattr(self._lib, k)


real = DR._lib
DR._lib = Shim(real)
torch.cuda.synchronize()
t0 = time.perf_counter()
multi()
torch.cuda.synchronize()
tot = time.perf_counter() - t0
DR._lib = real
fr = S * (n - 1)
print("breakdown per frame (us): align %.2f  prepare %.2f  rest (python book-keeping, start) %.2f  | total %.2f" %
      (acc["rir_ecc_align_multi_overlapped_device"] / fr * 1e6, acc.get("rir_ecc_prepare_frames_device", 0.0) / fr * 1e6,
       (tot - sum(acc.values())) / fr * 1e6, tot / fr * 1e6))
