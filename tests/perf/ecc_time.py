"""Development aid: rate of the device-resident ECC registrator on the S3 recipe (RIR_ECC_LAUNCH_PER_ITERATION=1: the round-1 form)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import s3_registration  # noqa: E402

n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 512, 640  # (the S3 recipe shifts by one pixel per frame: past ~100 frames the track is lost)
f32, shifts = s3_registration(n, h, w)
t = torch.from_numpy(f32).cuda()
for rep in range(2):
    reg = DeviceRegistratorECC(1, 1)
    reg.start(t[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if hasattr(reg, "compute_many") and not os.environ.get("RIR_ECC_ONE_BY_ONE"):
        reg.compute_many(t[1:])
    else:
        for i in range(1, n):
            reg.compute(t[i])
    dt = time.perf_counter() - t0
err = max(np.abs(np.array(reg.x) - shifts[:n, 0]).max(), np.abs(np.array(reg.y) - shifts[:n, 1]).max())
print("ECC registration: %.0f frames/s (%.1f us per frame), max error %.3f px" % ((n - 1) / dt, dt / (n - 1) * 1e6, err))
