"""CPU: the oracle's restatement of the bounded-loss step (reference h264.cpp:2253-2424 addImageLossyNoCamera,
:2426-2607 addLoss, :1526-1615 RunningAverage2).  The reference codec cannot be built here (ffmpeg/x264),
so this part of the oracle is pinned by a hand-worked known-answer case and by the documented invariants
(h264.h:93-104), not by reference outputs: "parity unpinned" for the lossy step."""
import numpy as np
import pytest

from librir_amd.synthetic import s1_noisy_background
from oracle.pyoracle import OracleLossy


def test_hand_worked_running_average_case(oracle):
    """4 pixels, runningAverage=2, errors 2/2, stdFactor 0; every value below was derived on paper from the
    reference's loop (sum += new; full ring: subtract constant or oldest; keep -> sum/count, reset -> refT=new)."""
    L = OracleLossy(oracle, 4, 1, 1, low_err=2, high_err=2, std_factor=0.0, running_average=2)
    f = np.array([[[10, 20, 30, 40]], [[11, 25, 30, 38]], [[12, 26, 33, 40]], [[10, 27, 33, 41]]], np.uint16)
    exp = np.array([[[10, 20, 30, 40]], [[11, 25, 30, 38]], [[11, 25, 33, 39]], [[11, 26, 33, 40]]], np.uint16)
    for i in range(4):
        assert np.array_equal(L.step(f[i]), exp[i]), i


def test_without_running_average_output_is_the_reference_pixel(oracle):
    L = OracleLossy(oracle, 4, 1, 1, low_err=2, high_err=2, std_factor=0.0, running_average=0)
    f = np.array([[[10, 20, 30, 40]], [[11, 25, 30, 38]], [[12, 26, 33, 40]]], np.uint16)
    exp = np.array([[[10, 20, 30, 40]], [[10, 25, 30, 40]], [[10, 25, 33, 40]]], np.uint16)
    for i in range(3):
        assert np.array_equal(L.step(f[i]), exp[i]), i


@pytest.mark.parametrize("ra,bound", [(0, 1), (4, 2), (32, 2)])
def test_error_bound_and_exact_rows(oracle, ra, bound):
    """|out - in| <= err without averaging, <= 2*err with it (mean of values each within err of refT); rows past
    lossy_height and the first image are stored exactly (h264.cpp:2274-2313, :2415-2417)."""
    h, w, hl, err = 40, 48, 37, 3
    arr = s1_noisy_background(50, h, w, seed=3)
    L = OracleLossy(oracle, w, h, hl, low_err=err, high_err=err, std_factor=0.0, running_average=ra)
    for i in range(50):
        out = L.step(arr[i])
        if i == 0:
            assert np.array_equal(out, arr[0])
        assert np.array_equal(out[hl:], arr[i, hl:])
        assert np.abs(out.astype(np.int32) - arr[i]).max() <= bound * err
        lo, hi, _ = L.last_errors()
        assert (lo, hi) == (err, err)


def test_error_budget_shrinks_with_std_factor_and_splits_after_40_frames(oracle):
    h, w = 32, 64
    arr = s1_noisy_background(60, h, w, seed=5)
    L = OracleLossy(oracle, w, h, h, low_err=6, high_err=2, std_factor=5.0, running_average=32)
    lows, highs = [], []
    for i in range(60):
        L.step(arr[i])
        lo, hi, _ = L.last_errors()
        lows.append(lo)
        highs.append(hi)
        assert 0 <= hi <= lo <= 6 and hi <= 2  # clamps of h264.cpp:2369-2372
    assert lows[0] == 6 and highs[0] == 2


def test_subtract_min_and_add_loss_variant(oracle):
    h, w = 16, 32
    arr = s1_noisy_background(8, h, w, seed=7)
    L = OracleLossy(oracle, w, h, h - 3, low_err=2, high_err=2, std_factor=0.0, running_average=0, subtract_min=True)
    mn = int(arr[0, : h - 3].min())
    out0 = L.step(arr[0])
    assert np.array_equal(out0[: h - 3], arr[0, : h - 3] - mn) and np.array_equal(out0[h - 3:], arr[0, h - 3:])
    for i in range(1, 8):
        out = L.step(arr[i], add_loss=True)
        back = out[: h - 3].astype(np.int32) + mn
        clipped = np.maximum(arr[i, : h - 3].astype(np.int32), mn)  # values under the first image's minimum clamp to it
        assert np.abs(back - clipped).max() <= 2
