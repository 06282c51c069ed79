"""Rate of MaskedRegistratorECC.compute, one host image per call (development aid, GPU box): the S3 recipe, 100 frames 640x512 float32."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd.registration import MaskedRegistratorECC  # noqa: E402
from librir_amd.synthetic import s3_registration  # noqa: E402

f32, shifts = s3_registration(100, 512, 640)
for rep in range(3):
    r = MaskedRegistratorECC(1, 1)
    r.start(f32[0])
    t0 = time.perf_counter()
    for i in range(1, 100):
        r.compute(f32[i])
    dt = time.perf_counter() - t0
    print("MaskedRegistratorECC per-frame: %.0f frames/s (%.1f us)" % (99 / dt, dt / 99 * 1e6), flush=True)

# the same images as 16-bit levels (what a recording holds): half the bytes over the link per image
u16 = (f32 - f32.min()).astype("uint16")
for rep in range(3):
    r = MaskedRegistratorECC(1, 1)
    r.start(u16[0])
    t0 = time.perf_counter()
    for i in range(1, 100):
        r.compute(u16[i])
    dt = time.perf_counter() - t0
    print("MaskedRegistratorECC per-frame, uint16 images: %.0f frames/s (%.1f us)" % (99 / dt, dt / 99 * 1e6), flush=True)
