"""Development aid: a digest of the tracks (translations, confidences) of the S3 recipe, solo and side by side - to check that a change of the
alignment kernels leaves every bit where it was.   python tests/perf/ecc_track_hash.py"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402
from librir_amd.synthetic import s3_registration  # noqa: E402

h, w = 512, 640
out = []
for (hh, ww, n, win) in ((512, 640, 100, 1.0), (240, 320, 40, 0.7), (97, 131, 15, 0.55)):
    frames = torch.from_numpy(s3_registration(n, hh, ww, seed=5)[0]).cuda()
    r = DeviceRegistratorECC(win, win, shape=(hh, ww))
    r.start(frames[0])
    try:
        r.compute_many(frames[1:])
    except RuntimeError as e:  # (a lost track ends the sequence; what was found up to there still counts)
        out.append("  (%dx%d: stopped after %d frames: %s)" % (ww, hh, len(r.x), str(e)[:40]))
    d = hashlib.sha256(np.asarray([r.x, r.y, r.confidences], dtype=np.float64).tobytes()).hexdigest()[:16]
    out.append("%dx%d n=%d window=%s: %s (last x %.4f y %.4f)" % (ww, hh, n, win, d, r.x[-1], r.y[-1]))
print("\n".join(out))
