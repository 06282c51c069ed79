import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from librir_amd.registration import DeviceRegistratorECC
from librir_amd.synthetic import s3_registration
S, n, h, w = 8, 100, 512, 640
seqs = [torch.from_numpy(s3_registration(n, h, w, seed=99 + q)[0]).cuda() for q in range(S)]
def multi(chunk):
    rs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(S)]
    for q in range(S):
        rs[q].start(seqs[q][0])
    DeviceRegistratorECC.compute_many_multi(rs, [s[1:] for s in seqs], chunk=chunk)
for chunk in (16, 25, 33, 50, 64, 99):
    multi(chunk)
    best = 0
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); multi(chunk); torch.cuda.synchronize()
        best = max(best, S * (n - 1) / (time.perf_counter() - t0))
    print("chunk %2d: %.1f k frames/s" % (chunk, best / 1e3))
