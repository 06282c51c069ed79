"""Development aid: device-resident registration loop (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librir_amd.registration import DeviceRegistratorECC
from librir_amd.synthetic import s3_registration
n = 60
f, s = s3_registration(n)
t = torch.from_numpy(f).cuda()
r = DeviceRegistratorECC(1, 1)
r.start(t[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, n):
    r.compute(t[i])
print("registration: %.1f us per frame" % ((time.perf_counter() - t0) / (n - 1) * 1e6))
