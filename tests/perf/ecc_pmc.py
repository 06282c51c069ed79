"""Development aid, run under rocprofv3 --pmc (scripts/ecc_pmc.sh): eight tracked sequences side by side (ecc_run_multi_kernel), 640x512."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import s3_registration  # noqa: E402

h, w, nreg = 512, 640, 65
seqs = [torch.from_numpy(s3_registration(nreg, h, w, seed=99 + q)[0]).cuda() for q in range(8)]
rs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(8)]
for q in range(8):
    rs[q].start(seqs[q][0])
DeviceRegistratorECC.compute_many_multi(rs, [s_[1:] for s_ in seqs], chunk=64)
torch.cuda.synchronize()
print("done")
