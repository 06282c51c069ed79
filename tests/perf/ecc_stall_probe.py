import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from librir_amd.registration import DeviceRegistratorECC
from librir_amd.registration import device_registration as DR
from librir_amd.synthetic import s3_registration
S, n, h, w = 8, 100, 512, 640
seqs = [torch.from_numpy(s3_registration(n, h, w, seed=99 + q)[0]).cuda() for q in range(S)]
real = DR._lib.rir_ecc_align_multi_device
times = []
class Shim:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if k != "rir_ecc_align_multi_device": return f
        def timed(*a):
            t0 = time.perf_counter(); r = f(*a); times.append(time.perf_counter() - t0); return r
        return timed
DR._lib = Shim(DR._lib)
bad = 0
for rep in range(12):
    rs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(S)]
    for q in range(S): rs[q].start(seqs[q][0])
    try:
        DeviceRegistratorECC.compute_many_multi(rs, [s[1:] for s in seqs])
    except RuntimeError as e:
        bad += 1
        print("rep", rep, "FAILED:", str(e)[-120:], flush=True)
print("calls %d, failures %d, align call times ms: max %.1f, sorted tail %s" % (len(times), bad, max(times) * 1e3, [round(t * 1e3, 1) for t in sorted(times)[-6:]]))
