"""CPU: the oracle's restatement of translation-only ECC (oracle/rir_oracle.c: orc_ecc_translation).
OpenCV - where the reference gets this arithmetic (masked_registration_ecc.py:166-168) - is not under
/root/reference and not installed: parity is unpinned, the reference's own test recipe
(tests/python/test_registration.py:41-59) is the anchor: known integer shifts must come back."""
import numpy as np
import pytest

from librir_amd.synthetic import s3_registration


def norm(a):
    return (a - a.min()) / (a.max() - a.min())


def test_identity_and_known_shift(oracle):
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:96, 0:128].astype(np.float32)
    img = (np.exp(-((xx - 60) ** 2 + (yy - 40) ** 2) / 200.0) + 0.5 * np.exp(-((xx - 30) ** 2 + (yy - 70) ** 2) / 80.0)).astype(np.float32)
    tx, ty, cc, it = oracle.ecc_translation(img, img)
    assert abs(tx) < 1e-6 and abs(ty) < 1e-6 and cc > 0.999999
    # image(x + t) ~ template(x): a template sampled 2.5 px to the right / 1.25 px up is found at t = (2.5, -1.25)
    moved = (np.exp(-((xx + 2.5 - 60) ** 2 + (yy - 1.25 - 40) ** 2) / 200.0) + 0.5 * np.exp(-((xx + 2.5 - 30) ** 2 + (yy - 1.25 - 70) ** 2) / 80.0))
    tx, ty, cc, it = oracle.ecc_translation(moved.astype(np.float32), img + rng.normal(0, 1e-3, img.shape).astype(np.float32), eps=1e-8)
    assert abs(tx - 2.5) < 0.02 and abs(ty + 1.25) < 0.02 and cc > 0.99


@pytest.mark.parametrize("eps,tol", [(1e-3, 0.25), (1e-5, 0.1)])
def test_reference_recipe_shifts_come_back(oracle, eps, tol):
    f, s = s3_registration(25, 256, 320)
    ref = norm(oracle.gaussian_filter(f[0], 0.5))
    warp = (0.0, 0.0)
    for i in range(1, 25):
        tx, ty, cc, it = oracle.ecc_translation(ref, norm(oracle.gaussian_filter(f[i], 0.5)), warp, eps=eps)
        warp = (tx, ty)  # warm start, like MaskedRegistratorECC.start_mat
        assert abs(tx - s[i, 0]) <= tol and abs(ty - s[i, 1]) <= tol and cc > 0.9, i


def test_no_overlap_raises(oracle):
    img = np.random.default_rng(1).random((32, 32)).astype(np.float32)
    with pytest.raises(RuntimeError):
        oracle.ecc_translation(img, img, warp=(100.0, 0.0))
