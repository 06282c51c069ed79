"""Development aid, run under rocprofv3 --pmc (scripts/pmc_config4.sh): the resident kernels of BASELINE configs[4] at their
headline sizes - bounded loss on one stream and on seven in one launch (lossy_run_kernel), ECC registration of one tracked
sequence (ecc_run_kernel) and of eight side by side (ecc_run_multi_kernel), 640x512."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import s1_noisy_background, s3_registration  # noqa: E402

h, w, n = 512, 640, 100
fr = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
one = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
one.step(fr, errors=False)  # (first call: the stream's first frame goes alone)
one.step(fr, errors=False)
one.status()
S = 7
streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
ins = [fr.clone() for _ in range(S)]
D.LossyStream.step_many(streams, ins, errors=False)
D.LossyStream.step_many(streams, ins, errors=False)
streams[0].status()
nreg = 60
seqs = [torch.from_numpy(s3_registration(nreg, h, w, seed=99 + q)[0]).cuda() for q in range(8)]
r = DeviceRegistratorECC(1, 1, shape=(h, w))
r.start(seqs[0][0])
r.compute_many(seqs[0][1:])
rs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(8)]
for q in range(8):
    rs[q].start(seqs[q][0])
DeviceRegistratorECC.compute_many_multi(rs, [s_[1:] for s_ in seqs])
torch.cuda.synchronize()
print("done")
