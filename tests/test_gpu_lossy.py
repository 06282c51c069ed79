"""GPU: bounded-loss recording (h264_add_image_lossy / h264_add_loss) against the oracle's restatement of
reference h264.cpp:2253-2607, bit-exact frame by frame, plus the files' attributes and error vectors
(reference tests/python/test_video_io.py:96-144)."""
import numpy as np
import pytest

from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background
from librir_amd.video_io import IRMovie, IRSaver
from oracle.pyoracle import OracleLossy

pytestmark = pytest.mark.gpu


def run_saver(path, arr, hl, params, add_loss_from=None):
    n, h, w = arr.shape
    losses = []
    with IRSaver(path, w, h, hl) as s:
        for k, v in params.items():
            s.set_parameter(k, v)
        for i in range(n):
            if add_loss_from is not None and i >= add_loss_from:
                losses.append(s.add_loss(arr[i]))
            else:
                s.add_image_lossy(arr[i], i * 1000, {"idx": str(i)})
        low, high = np.array(s.get_low_errors()), np.array(s.get_high_errors())
    return low, high, losses


CASES = {
    "defaults_60_frames": dict(n=60, h=64, w=80, hl=64, p={}),
    "flat_budget": dict(n=20, h=64, w=80, hl=61, p={"lowValueError": 3, "highValueError": 3, "stdFactor": 0}),
    "no_average": dict(n=20, h=35, w=83, hl=32, p={"runningAverage": 0, "lowValueError": 5, "highValueError": 1}),
    "short_ring": dict(n=30, h=48, w=96, hl=45, p={"runningAverage": 3, "stdFactor": 2.5}),
    "subtract_min": dict(n=20, h=64, w=80, hl=61, p={"subtractMin": 1, "lowValueError": 4, "highValueError": 2}),
    "full_frame_640x512": dict(n=45, h=512, w=640, hl=509, p={}),
}


@pytest.mark.parametrize("name", list(CASES))
def test_add_image_lossy_matches_oracle(tmp_path, oracle, name):
    c = CASES[name]
    arr = s1_noisy_background(c["n"], c["h"], c["w"], seed=11)
    p = c["p"]
    L = OracleLossy(oracle, c["w"], c["h"], c["hl"], low_err=int(p.get("lowValueError", 6)), high_err=int(p.get("highValueError", 2)),
                    std_factor=float(p.get("stdFactor", 5.0)), running_average=int(p.get("runningAverage", 32)),
                    subtract_min=bool(p.get("subtractMin", 0)))
    exp, elow, ehigh = [], [], []
    for i in range(c["n"]):
        exp.append(L.step(arr[i]))
        lo, hi, _ = L.last_errors()
        elow.append(lo)
        ehigh.append(hi)
    exp = np.stack(exp)
    dst = tmp_path / "lossy.h264"
    low, high, _ = run_saver(dst, arr, c["hl"], p)
    assert low.tolist() == elow and high.tolist() == ehigh
    with IRMovie.from_filename(dst) as mov:
        got = mov.data
        assert int(mov.attributes["GlobalBackgroundError"]) == int(p.get("lowValueError", 6))
    if p.get("subtractMin"):
        # the loader adds MIN_T back on the first MIN_T_HEIGHT rows (IRFileLoader.cpp:1173-1179)
        mn = int(arr[0, : c["hl"]].min())
        exp = exp.copy()
        exp[:, : c["hl"]] += np.uint16(mn)
    assert np.array_equal(got, exp)
    bound = (1 if int(p.get("runningAverage", 32)) == 0 else 2) * int(p.get("lowValueError", 6))
    if not p.get("subtractMin"):
        assert np.abs(got.astype(np.int32) - arr).max() <= bound
    assert np.array_equal(got[:, c["hl"]:], arr[:, c["hl"]:])


def test_add_loss_matches_oracle_and_leaves_the_file_alone(tmp_path, oracle):
    n, h, w, hl = 50, 40, 72, 37
    arr = s1_noisy_background(n, h, w, seed=13)
    L = OracleLossy(oracle, w, h, hl, low_err=6, high_err=2, std_factor=5.0, running_average=32)
    exp = [L.step(arr[i], add_loss=(i >= 1)) for i in range(n)]
    dst = tmp_path / "loss.h264"
    low, high, losses = run_saver(dst, arr, hl, {}, add_loss_from=1)
    assert len(low) == len(high) == n
    for i in range(1, n):
        assert np.array_equal(losses[i - 1], exp[i]), i
    with IRMovie.from_filename(dst) as mov:  # only the first image went through add_image_lossy
        assert mov.images == 1 and np.array_equal(mov[0], arr[0])


def test_remove_bad_pixels_option(tmp_path, oracle):
    """removeBadPixels: detector initialised on the first image's lossy rows, every image corrected before
    the loss is injected (h264.cpp:2259-2266)."""
    n, h, w, hl = 12, 67, 83, 64
    arr = inject_bad_pixels(s1_noisy_background(n, h, w, seed=17), 9)
    xy = oracle.bad_pixels_detect(arr[0, :hl])
    _, fc = oracle.bad_pixels_stats(arr[0, :hl])
    assert len(xy) > 0
    L = OracleLossy(oracle, w, h, hl, low_err=3, high_err=3, std_factor=0.0, running_average=4)
    exp = []
    for i in range(n):
        f = arr[i].copy()
        f[:hl] = oracle.bad_pixels_correct(arr[i, :hl], xy, fc)
        exp.append(L.step(f))
    dst = tmp_path / "bp.h264"
    run_saver(dst, arr, hl, {"removeBadPixels": 1, "lowValueError": 3, "highValueError": 3, "stdFactor": 0, "runningAverage": 4})
    with IRMovie.from_filename(dst) as mov:
        assert np.array_equal(mov.data, np.stack(exp))


def test_per_frame_error_attributes(tmp_path):
    arr = s1_noisy_background(6, 32, 64, seed=19)
    dst = tmp_path / "attrs.h264"
    low, high, _ = run_saver(dst, arr, 32, {})
    with IRMovie.from_filename(dst) as mov:
        mov.load_pos(0)
        a0 = mov.frame_attributes
        assert "BackgroundError" not in a0 and a0["idx"] in ("0", b"0")
        mov.load_pos(3)
        a3 = mov.frame_attributes
        assert int(a3["BackgroundError"]) == low[3] and int(a3["ForegroundError"]) == high[3]
        assert int(mov.attributes["GlobalForegroundError"]) == 2


@pytest.fixture(params=["one launch per run", "one launch per frame", "batches of streams", "state parked in LDS", "parked, batches of streams"])
def run_path(request, monkeypatch):
    """Batches of frames are stepped by the resident run kernel - all streams in one launch, or, where the chip does not hold them
    at once, a batch of streams after the other (RIR_LOSSY_RUN_MAX_WORKGROUPS lowers the limit so that small frames get there) -
    or, frames too large for it, by one fused launch per frame (RIR_LOSSY_LAUNCH_PER_FRAME forces that path at any size).  The run
    kernel has a second form, taken when it saves a launch (part of the pixel state parked in LDS, 6 waves per SIMD:
    RIR_LOSSY_RUN_FORM=6 forces it, 5 forbids it)."""
    monkeypatch.delenv("RIR_LOSSY_LAUNCH_PER_FRAME", raising=False)
    monkeypatch.delenv("RIR_LOSSY_RUN_MAX_WORKGROUPS", raising=False)
    monkeypatch.setenv("RIR_LOSSY_RUN_FORM", "6" if "parked" in request.param else "5")
    if request.param == "one launch per frame":
        monkeypatch.setenv("RIR_LOSSY_LAUNCH_PER_FRAME", "1")
    elif "batches of streams" in request.param:
        monkeypatch.setenv("RIR_LOSSY_RUN_MAX_WORKGROUPS", "13")  # 128x96 frames take 6 workgroups: two streams per launch
    return request.param


@pytest.mark.parametrize("add_loss", [False, True])
def test_device_resident_stream_matches_oracle(oracle, add_loss, run_path):
    """rir_lossy_step_device: the same arithmetic on frames that never leave HBM, in two batches (state carries over)."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 64, 64, 96, 61
    arr = s1_noisy_background(n, h, w, seed=23)
    L = OracleLossy(oracle, w, h, hl, low_err=5, high_err=2, std_factor=2.5, running_average=8)
    exp, elo, ehi = [], [], []
    for i in range(n):
        exp.append(L.step(arr[i], add_loss=add_loss and i > 0))
        lo, hi, _ = L.last_errors()
        elo.append(lo)
        ehi.append(hi)
    ls = D.LossyStream(w, h, hl, 5, 2, 2.5, 8)
    t = torch.from_numpy(arr).cuda()
    a, lo_a, hi_a = ls.step(t[:1], add_loss=False)
    b, lo_b, hi_b = ls.step(t[1:40], add_loss=add_loss)
    c, lo_c, hi_c = ls.step(t[40:], add_loss=add_loss)
    got = torch.cat([a, b, c]).cpu().numpy()
    assert np.array_equal(got, np.stack(exp))
    assert np.concatenate([lo_a, lo_b, lo_c]).tolist() == elo and np.concatenate([hi_a, hi_b, hi_c]).tolist() == ehi
    ls.close()
    # the queue-only form (no error arrays, nothing waits inside the call): same frames
    ls = D.LossyStream(w, h, hl, 5, 2, 2.5, 8)
    a, lo_a, _ = ls.step(t[:1], add_loss=False, errors=False)
    b, _, _ = ls.step(t[1:], add_loss=add_loss, errors=False)
    assert lo_a is None
    ls.status()  # (queue-only calls leave the device-side verdict to this query)
    assert np.array_equal(torch.cat([a, b]).cpu().numpy(), np.stack(exp))
    ls.close()
    # runs, single frames (three launches each) and pairs interleaved: the state goes through registers and back unchanged
    ls = D.LossyStream(w, h, hl, 5, 2, 2.5, 8)
    cuts = [0, 1, 5, 6, 7, 30, 32, 33, n]
    parts = [ls.step(t[c0:c1], add_loss=add_loss and c0 > 0) for c0, c1 in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(torch.cat([p[0] for p in parts]).cpu().numpy(), np.stack(exp))
    assert np.concatenate([p[1] for p in parts]).tolist() == elo and np.concatenate([p[2] for p in parts]).tolist() == ehi
    ls.close()


def test_many_streams_in_shared_launches_equal_their_own_oracles(oracle, run_path):
    """rir_lossy_step_multi_device: S independent streams (different data, different parameters) stepped by the same launches,
    in two calls (the first seeds the states, the second continues them): every stream's frames and per-frame budgets are
    those of its own oracle, bit for bit; a stream stepped alone gives the same result."""
    import torch

    from librir_amd import device as D

    S, n, h, w, hl = 5, 47, 96, 128, 93
    params = [dict(low=6, high=2, sf=5.0, ra=32), dict(low=3, high=3, sf=0.0, ra=4), dict(low=5, high=1, sf=2.5, ra=0),
              dict(low=9, high=4, sf=5.0, ra=7), dict(low=6, high=2, sf=5.0, ra=32)]
    data = [s1_noisy_background(n, h, w, seed=20 + i) for i in range(S)]
    streams = [D.LossyStream(w, h, hl, p["low"], p["high"], p["sf"], p["ra"]) for p in params]
    tens = [torch.from_numpy(d).cuda() for d in data]
    cut = 13
    o1, lo1, hi1 = D.LossyStream.step_many(streams, [t[:cut] for t in tens])
    o2, lo2, hi2 = D.LossyStream.step_many(streams, [t[cut:] for t in tens])
    for i, p in enumerate(params):
        L = OracleLossy(oracle, w, h, hl, low_err=p["low"], high_err=p["high"], std_factor=p["sf"], running_average=p["ra"])
        exp, elo, ehi = [], [], []
        for f in range(n):
            exp.append(L.step(data[i][f]))
            lo, hi, _ = L.last_errors()
            elo.append(lo)
            ehi.append(hi)
        got = np.concatenate([o1[i].cpu().numpy(), o2[i].cpu().numpy()])
        assert np.array_equal(got, np.stack(exp)), i
        assert np.concatenate([lo1[i], lo2[i]]).tolist() == elo and np.concatenate([hi1[i], hi2[i]]).tolist() == ehi, i
    alone = D.LossyStream(w, h, hl, params[3]["low"], params[3]["high"], params[3]["sf"], params[3]["ra"])
    oa, loa, hia = alone.step(tens[3])
    assert np.array_equal(oa.cpu().numpy(), np.concatenate([o1[3].cpu().numpy(), o2[3].cpu().numpy()]))
    # streams that are out of step with each other are refused
    fresh = D.LossyStream(w, h, hl)
    with pytest.raises(RuntimeError):
        D.LossyStream.step_many([streams[0], fresh], [tens[0][:2], tens[1][:2]])
    for s in streams + [alone, fresh]:
        s.close()


def test_recording_interleaved_with_queries_parameter_changes_and_add_loss(tmp_path, oracle):
    """Bounded-loss frames are stepped chunk by chunk as runs of frames; whatever needs the loss state or the budgets in between -
    the error getters, a parameter change, add_loss on an image that is not recorded, a lossless add_image - comes after the
    frames already handed in and before the ones that follow, exactly as if every frame had been stepped in its own call."""
    n, h, w, hl = 130, 48, 96, 45
    arr = s1_noisy_background(n + 1, h, w, seed=29)
    L = OracleLossy(oracle, w, h, hl, low_err=6, high_err=2, std_factor=5.0, running_average=8)
    exp, elow, ehigh = {}, [], []
    dst = tmp_path / "mixed.h264"
    recorded = []
    with IRSaver(dst, w, h, hl) as s:
        s.set_parameter("runningAverage", 8)
        s.set_parameter("GOP", 40)
        for i in range(n):
            if i == 61:  # a parameter change applies from this frame on
                s.set_parameter("lowValueError", 9)
                L.set_errors(9, 2)
            if i == 77:  # an image that only goes through the loss, not into the file
                got_loss = s.add_loss(arr[n])
                exp_loss = L.step(arr[n], add_loss=True)
                lo, hi, _ = L.last_errors()
                elow.append(lo), ehigh.append(hi)
                assert np.array_equal(got_loss, exp_loss)
            if i == 100:  # a lossless frame in the middle of a chunk
                s.add_image(arr[i], i * 1000)
                exp[i] = arr[i]
                recorded.append(i)
                continue
            s.add_image_lossy(arr[i], i * 1000)
            exp[i] = L.step(arr[i])
            lo, hi, _ = L.last_errors()
            elow.append(lo), ehigh.append(hi)
            recorded.append(i)
            if i in (7, 8, 53):  # budgets asked for in the middle of a chunk
                assert list(s.get_low_errors()) == elow and list(s.get_high_errors()) == ehigh
        assert list(s.get_low_errors()) == elow and list(s.get_high_errors()) == ehigh
    with IRMovie.from_filename(dst) as mov:
        assert mov.images == len(recorded)
        got = mov.data
    for k, i in enumerate(recorded):
        assert np.array_equal(got[k], exp[i]), i


@pytest.mark.perf
def test_device_resident_step_rate():
    """The bounded-loss step on frames that stay in HBM, one 640x512 stream, BASELINE configs[4]'s parameters (stdFactor 0: the
    constant-budget form).  Review r3 asked for >= 1 M frames/s in long calls (measured 1.6-1.7 M in 1 000-frame calls, 0.85 M in
    the 200-frame calls timed here; `tests/perf/lossy_const_time.py`); the general (resident) form, which this floor was written
    for in round 2, runs at 0.16 M.  The floors leave a third of margin for a busy box; best of three."""
    import time

    import torch

    from librir_amd import device as D

    n, h, w = 200, 512, 640
    fr = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
    st = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
    st.step(fr[:60], errors=False)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        st.step(fr, errors=False)
        st.step(fr, errors=False)
        torch.cuda.synchronize()
        best = max(best, 2 * n / (time.perf_counter() - t0))
    st.status()
    st_path = st.path_stats()
    st.close()
    print("bounded-loss step, one stream, 200-frame calls: %.0f frames/s" % best)
    assert st_path[1] >= 1, st_path  # (taken by the constant-budget form)
    assert best >= 500000, best


@pytest.mark.parametrize("ra", [1, 2, 64])
def test_ring_lengths_at_the_edges(oracle, ra, run_path):
    """Running averages over 1 image (the ring's oldest image is the one written a frame ago: the resident kernel keeps it in
    registers), 2 and the maximum of 64, through every run path."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 90, 40, 64, 38
    arr = s1_noisy_background(n, h, w, seed=31)
    L = OracleLossy(oracle, w, h, hl, low_err=4, high_err=2, std_factor=1.5, running_average=ra)
    exp, elo, ehi = [], [], []
    for i in range(n):
        exp.append(L.step(arr[i]))
        lo, hi, _ = L.last_errors()
        elo.append(lo)
        ehi.append(hi)
    ls = D.LossyStream(w, h, hl, 4, 2, 1.5, ra)
    t = torch.from_numpy(arr).cuda()
    cuts = [0, 1, 30, 31, n]
    parts = [ls.step(t[c0:c1]) for c0, c1 in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(torch.cat([p[0] for p in parts]).cpu().numpy(), np.stack(exp))
    assert np.concatenate([p[1] for p in parts]).tolist() == elo and np.concatenate([p[2] for p in parts]).tolist() == ehi
    ls.close()


@pytest.mark.perf
def test_recording_rate(tmp_path):
    """Bounded-loss recording through the per-frame ABI (IRSaver.add_image_lossy, 640x512): the frames of a chunk are stepped as one
    run, measured 24 k frames/s (8.8 k with an upload and three launches per frame).  Floor with a factor two of margin; best of two."""
    import time

    n, h, w = 500, 512, 640
    fr = s1_noisy_background(n, h, w)
    best = 0.0
    for rep in range(3):  # (the first recording of a process page-locks its staging buffers)
        dst = tmp_path / ("rate%d.h264" % rep)
        t0 = time.perf_counter()
        with IRSaver(dst, w, h, h - 3) as s:
            s.set_parameter("lowValueError", 3)
            s.set_parameter("highValueError", 3)
            s.set_parameter("stdFactor", 0)
            for i in range(n):
                s.add_image_lossy(fr[i], i * 1000)
        if rep:
            best = max(best, n / (time.perf_counter() - t0))
    with IRMovie.from_filename(dst) as mov:
        assert mov.images == n and np.abs(mov[n - 1].astype(np.int32) - fr[n - 1]).max() <= 6
    print("bounded-loss recording: %.0f frames/s" % best)
    assert best >= 12000, best


# ---- the constant-budget form of a run (stdFactor == 0: lossy_const_run_kernel) ------------------------------------------------
def _oracle_track(oracle, arr, w, h, hl, low, high, sf, ra, subtract_min=False, add_loss=False, changes=None):
    """frames, low and high errors of the oracle; changes: {frame: (low, high, std_factor)} applied before that frame"""
    L = OracleLossy(oracle, w, h, hl, low_err=low, high_err=high, std_factor=sf, running_average=ra, subtract_min=subtract_min)
    exp, elo, ehi = [], [], []
    for i in range(arr.shape[0]):
        if changes and i in changes:
            L.set_errors(*changes[i])
        exp.append(L.step(arr[i], add_loss=add_loss and i > 0))
        lo, hi, _ = L.last_errors()
        elo.append(lo)
        ehi.append(hi)
    return np.stack(exp), elo, ehi


CONST_CASES = {
    "ra8": dict(n=130, h=64, w=96, hl=61, low=3, high=3, ra=8, cuts=[0, 1, 50, 130]),
    "ra0": dict(n=60, h=40, w=64, hl=40, low=5, high=1, ra=0, cuts=[0, 1, 60]),
    "ra1": dict(n=70, h=40, w=64, hl=38, low=4, high=2, ra=1, cuts=[0, 1, 30, 31, 70]),
    "ra2": dict(n=70, h=40, w=64, hl=38, low=4, high=2, ra=2, cuts=[0, 1, 3, 70]),
    "ra64_longer_than_the_calls": dict(n=150, h=40, w=64, hl=38, low=4, high=2, ra=64, cuts=[0, 1, 20, 45, 150]),
    "ra32_ring_fills_inside_a_call": dict(n=90, h=48, w=64, hl=48, low=6, high=2, ra=32, cuts=[0, 10, 90]),
    "high_above_low": dict(n=50, h=40, w=64, hl=38, low=1, high=4, ra=4, cuts=[0, 1, 50]),
    "subtract_min": dict(n=60, h=40, w=64, hl=37, low=4, high=2, ra=8, cuts=[0, 1, 60], subtract_min=True),
    "short_calls": dict(n=40, h=40, w=64, hl=38, low=3, high=3, ra=4, cuts=[0, 1, 4, 7, 11, 40]),
}


@pytest.mark.parametrize("add_loss", [False, True], ids=["add_image_lossy", "add_loss"])
@pytest.mark.parametrize("name", list(CONST_CASES))
def test_constant_budget_form_matches_oracle(oracle, name, add_loss, monkeypatch):
    """stdFactor == 0 (BASELINE configs[4], reference test_video_io.py:112-116): batches of frames go through the streaming kernel
    that needs no hand-off between workgroups - every group of frames taken by it (path_stats), frames and budgets the oracle's,
    the state handed between the calls, the rings of every length, both variants of the decision."""
    import torch

    from librir_amd import device as D

    monkeypatch.delenv("RIR_LOSSY_LAUNCH_PER_FRAME", raising=False)
    monkeypatch.delenv("RIR_LOSSY_RUN_MAX_WORKGROUPS", raising=False)
    c = CONST_CASES[name]
    arr = s1_noisy_background(c["n"], c["h"], c["w"], seed=41)
    exp, elo, ehi = _oracle_track(oracle, arr, c["w"], c["h"], c["hl"], c["low"], c["high"], 0.0, c["ra"], c.get("subtract_min", False), add_loss)
    ls = D.LossyStream(c["w"], c["h"], c["hl"], c["low"], c["high"], 0.0, c["ra"], subtract_min=c.get("subtract_min", False))
    t = torch.from_numpy(arr).cuda()
    got, lo, hi = [], [], []
    for c0, c1 in zip(c["cuts"][:-1], c["cuts"][1:]):
        o, l_, h_ = ls.step(t[c0:c1], add_loss=add_loss and c0 > 0)
        got.append(o), lo.append(l_), hi.append(h_)
        steps = (c1 - c0) - (1 if c0 == 0 else 0)
        if c1 - c0 >= 3 and steps >= 2:
            offered, taken = ls.path_stats()
            assert offered >= 1 and taken == offered, (name, c0, c1, offered, taken)
    assert np.array_equal(torch.cat(got).cpu().numpy(), exp)
    assert np.concatenate(lo).tolist() == elo and np.concatenate(hi).tolist() == ehi
    ls.close()


@pytest.mark.parametrize("pairs", [2, 4])
def test_constant_budget_form_with_more_pixels_per_thread(pairs):
    """The streaming kernel exists for 2, 4 and 8 pixels per thread (lossy_const_pairs picks by the number of waves: many streams of
    640x512 take 8); small test frames would only ever see 2.  The test hook RIR_LOSSY_CONST_PAIRS forces the other two - in a process
    that loads the build with the hooks (tests/hook_cases.py: const_pairs): same frames, same budgets as the oracle's."""
    from test_gpu_resident import _hook_case

    _hook_case("const_pairs", pairs)


def test_constant_budget_form_keeps_the_history_a_later_std_factor_needs(oracle):
    """The window of statistics is history even while it is multiplied by zero: stdFactor raised on a stream that has gone through
    the constant-budget form - before its window is full, and long after - finds every entry where the reference would have it."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 200, 48, 96, 45
    arr = s1_noisy_background(n, h, w, seed=43)
    for switch, back in ((20, 60), (131, 170)):
        changes = {switch: (6, 2, 5.0), back: (4, 4, 0.0)}
        exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 3, 3, 0.0, 8, changes=changes)
        ls = D.LossyStream(w, h, hl, 3, 3, 0.0, 8)
        t = torch.from_numpy(arr).cuda()
        a = ls.step(t[:switch])
        assert ls.path_stats()[1] >= 1
        ls.set_errors(6, 2, 5.0)
        b = ls.step(t[switch:back])
        assert ls.path_stats() == (0, 0)
        ls.set_errors(4, 4, 0.0)
        c = ls.step(t[back:])
        assert ls.path_stats()[1] >= 1
        assert np.array_equal(torch.cat([a[0], b[0], c[0]]).cpu().numpy(), exp), switch
        assert np.concatenate([a[1], b[1], c[1]]).tolist() == elo and np.concatenate([a[2], b[2], c[2]]).tolist() == ehi, switch
        assert any(e != 6 for e in elo[switch:back]), "the test stream never moved the budget"
        ls.close()


def test_constant_budget_form_declines_what_it_must_not_take(oracle):
    """An empty foreground or background makes the reference's statistic 0 / 0, and the NaN decides the budgets of that frame and of the
    39 after it (lossless frames): groups with a frame whose classes are not surely both there, and groups stepped while a NaN sits in
    the window, are declined on the device and stepped by the general form; afterwards the streaming form takes over again.  Same
    frames and budgets as the oracle throughout."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 260, 40, 64, 40
    arr = s1_noisy_background(n, h, w, seed=47)
    arr[60:64] = 1000  # uniform frames: everything in the mode bin, no foreground (from frame 41 on the statistic is split)
    arr[64] = np.where(np.arange(h * w).reshape(h, w) % 2 == 0, 1000, 1001)  # two levels inside one bin: nothing above, nothing below
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 3, 3, 0.0, 4)
    assert ehi[59] == 3 and ehi[60] == 0 and ehi[103] == 0 and ehi[104] == 3 and set(elo) == {3}, "the crafted stream does not do what the test is about"
    ls = D.LossyStream(w, h, hl, 3, 3, 0.0, 4)
    t = torch.from_numpy(arr).cuda()
    cuts = [0, 1, 50, 70, 90, 140, 200, n]
    want_taken = {(1, 50): True, (50, 70): False, (70, 90): False, (90, 140): False, (140, 200): True, (200, n): True}
    got, lo, hi = [], [], []
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        o, l_, h_ = ls.step(t[c0:c1])
        got.append(o), lo.append(l_), hi.append(h_)
        if (c0, c1) in want_taken:
            offered, taken = ls.path_stats()
            assert offered == 1 and (taken == 1) == want_taken[(c0, c1)], (c0, c1, offered, taken)
    assert np.array_equal(torch.cat(got).cpu().numpy(), exp)
    assert np.concatenate(lo).tolist() == elo and np.concatenate(hi).tolist() == ehi
    ls.close()


def test_constant_budget_form_with_many_streams_and_mixed_parameters(oracle):
    """several streams in the same launches: all with stdFactor 0 -> the streaming form (each with its own errors and ring), one with another
    factor -> the whole call through the general form; every stream equals its own oracle either way"""
    import torch

    from librir_amd import device as D

    S, n, h, w, hl = 4, 75, 48, 64, 45
    for params in ([dict(low=3, high=3, sf=0.0, ra=4), dict(low=6, high=2, sf=0.0, ra=32), dict(low=5, high=1, sf=0.0, ra=0), dict(low=2, high=2, sf=0.0, ra=7)],
                   [dict(low=3, high=3, sf=0.0, ra=4), dict(low=6, high=2, sf=2.5, ra=32), dict(low=5, high=1, sf=0.0, ra=0), dict(low=2, high=2, sf=0.0, ra=7)]):
        data = [s1_noisy_background(n, h, w, seed=60 + i) for i in range(S)]
        streams = [D.LossyStream(w, h, hl, p["low"], p["high"], p["sf"], p["ra"]) for p in params]
        tens = [torch.from_numpy(d).cuda() for d in data]
        o1, lo1, hi1 = D.LossyStream.step_many(streams, [t[:30] for t in tens])
        o2, lo2, hi2 = D.LossyStream.step_many(streams, [t[30:] for t in tens])
        offered, taken = streams[0].path_stats()
        if all(p["sf"] == 0.0 for p in params):
            assert offered >= 1 and taken == offered
        else:
            assert offered == 0
        for i, p in enumerate(params):
            exp, elo, ehi = _oracle_track(oracle, data[i], w, h, hl, p["low"], p["high"], p["sf"], p["ra"])
            assert np.array_equal(np.concatenate([o1[i].cpu().numpy(), o2[i].cpu().numpy()]), exp), i
            assert np.concatenate([lo1[i], lo2[i]]).tolist() == elo and np.concatenate([hi1[i], hi2[i]]).tolist() == ehi, i
        for s_ in streams:
            s_.close()


def test_constant_budget_form_full_size_many_streams(oracle):
    """Seventeen 640x512 streams in one call: the launch that takes 8 pixels per thread by itself (lossy_const_pairs: 5 000 waves and
    more with four pairs), every phase of the streaming kernel (first frames, the loop of the middle, the frames of the end) at the
    size the rates are quoted for - each stream against its own oracle, rings of several lengths side by side."""
    import torch

    from librir_amd import device as D

    h, w, n, S = 512, 640, 56, 17
    base = s1_noisy_background(n, h, w, seed=71)
    streams, ins, exps = [], [], []
    for i in range(S):
        ra, smin = (0, 3, 8, 32)[i % 4], i % 3 == 0
        arr = base if i == 0 else (base + np.uint16(7 * i)).astype(np.uint16)
        if i % 5 == 2:
            arr = arr[:, ::-1].copy()
        exps.append(_oracle_track(oracle, arr, w, h, h - 3, 4 + i % 3, 2, 0.0, ra, smin))
        streams.append(D.LossyStream(w, h, h - 3, 4 + i % 3, 2, 0.0, ra, subtract_min=smin))
        ins.append(torch.from_numpy(arr).cuda())
    o1, lo1, hi1 = D.LossyStream.step_many(streams, [t[:1] for t in ins])
    o2, lo2, hi2 = D.LossyStream.step_many(streams, [t[1:] for t in ins])
    offered, taken = streams[0].path_stats()
    assert offered >= 1 and taken == offered, (offered, taken)
    for i in range(S):
        exp, elo, ehi = exps[i]
        got = torch.cat([o1[i], o2[i]]).cpu().numpy()
        bad = [k for k in range(n) if not np.array_equal(got[k], exp[k])]
        assert not bad, (i, bad)
        assert np.concatenate([lo1[i], lo2[i]]).tolist() == elo and np.concatenate([hi1[i], hi2[i]]).tolist() == ehi, i
    for s_ in streams:
        s_.close()


@pytest.mark.parametrize("shape", [(768, 1024, 765), (600, 800, 600), (513, 648, 511)], ids=["1024x768", "800x600", "648x513"])
def test_constant_budget_form_other_full_sizes(oracle, shape):
    """configs[3]'s geometry and two others (a lossy height that is the whole frame; a width that is a multiple of 8 but not of 64, the
    last workgroup of a stream half empty): 70 frames in two calls, ring of 8, subtractMin - frames and budgets the oracle's."""
    import torch

    from librir_amd import device as D

    h, w, hl = shape
    n = 70
    arr = s1_noisy_background(n, h, w, seed=83)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 5, 2, 0.0, 8, True)
    ls = D.LossyStream(w, h, hl, 5, 2, 0.0, 8, subtract_min=True)
    t = torch.from_numpy(arr).cuda()
    a = ls.step(t[:9])
    assert ls.path_stats()[1] >= 1
    b = ls.step(t[9:])
    assert ls.path_stats()[1] >= 1
    got = torch.cat([a[0], b[0]]).cpu().numpy()
    bad = [k for k in range(n) if not np.array_equal(got[k], exp[k])]
    assert not bad, bad
    assert np.concatenate([a[1], b[1]]).tolist() == elo and np.concatenate([a[2], b[2]]).tolist() == ehi
    ls.close()


def test_constant_budget_form_through_the_saver_with_a_parameter_change(tmp_path, oracle):
    """h264_add_image_lossy with stdFactor 0 (the saver steps its chunks as runs of frames), stdFactor raised in mid-recording"""
    n, h, w, hl = 150, 48, 96, 45
    arr = s1_noisy_background(n, h, w, seed=53)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 3, 3, 0.0, 8, changes={95: (3, 3, 5.0)})
    dst = tmp_path / "const.h264"
    with IRSaver(dst, w, h, hl) as s:
        for k, v in (("lowValueError", 3), ("highValueError", 3), ("stdFactor", 0), ("runningAverage", 8), ("GOP", 40)):
            s.set_parameter(k, v)
        for i in range(n):
            if i == 95:
                s.set_parameter("stdFactor", 5)
            s.add_image_lossy(arr[i], i * 1000)
        assert list(s.get_low_errors()) == elo and list(s.get_high_errors()) == ehi
    with IRMovie.from_filename(dst) as mov:
        assert np.array_equal(mov.data, exp)


FLAT_SCENES = {
    # level of the scene, noise (levels), hot blob (offset above the scene, or None), frames
    "low_levels": dict(level=14, noise=3, blob=40),
    "mid_levels_wide_noise": dict(level=21000, noise=30, blob=200),
    "top_of_the_range": dict(level=65470, noise=6, blob=50),
    "two_plateaus": dict(level=3000, noise=2, blob=None, plateau=90),
}


@pytest.mark.parametrize("name", list(FLAT_SCENES))
@pytest.mark.parametrize("std_factor", [0.0, 5.0], ids=["constant_budgets", "general_run"])
def test_histogram_pass_on_flat_full_size_scenes(oracle, name, std_factor):
    """The histogram pass keeps a 16-bin window of counts in REGISTERS per wave (lossy_hist_add8) and empties it every 31 rounds: small test
    frames are less than one round per wave, so this is the test that takes a frame through all of it - 640x512 (40 rounds per wave:
    one flush on the way, one at the end), flat scenes whose pixels sit in a handful of bins, with a hot blob above the window (pixels
    that go through the atomics beside it), a plateau 22 bins up (a window that has to be left and anchored anew), a constant frame, a
    frame at the top of the value range (a window that reaches past the last bin).  The background of every frame decides which of the
    two error bounds a pixel gets (low != high here): frames and budgets must be the oracle's."""
    import torch

    from librir_amd import device as D

    c = FLAT_SCENES[name]
    n, h, w, hl = 14, 512, 640, 509
    rng = np.random.default_rng(97)
    fr = np.clip(c["level"] + rng.integers(-c["noise"], c["noise"] + 1, (n, h, w)) + np.arange(n)[:, None, None] * 2, 0, 65535).astype(np.uint16)
    if c.get("blob"):
        yy, xx = np.mgrid[0:h, 0:w]
        m = (yy - 170) ** 2 + (xx - 120) ** 2 < 75 ** 2
        fr[:, m] = np.clip(fr[:, m].astype(np.int64) + c["blob"] + rng.integers(0, 60, (n, int(m.sum()))), 0, 65535).astype(np.uint16)
    if c.get("plateau"):
        fr[:, h // 3:, :] += np.uint16(c["plateau"])
    fr[5] = fr[5, 0, 0]  # a constant frame: one bin, 325 760 times
    exp, elo, ehi = _oracle_track(oracle, fr, w, h, hl, 6, 2, std_factor, 8)
    ls = D.LossyStream(w, h, hl, 6, 2, std_factor, 8)
    t = torch.from_numpy(fr).cuda()
    a = ls.step(t[:1])
    b = ls.step(t[1:])
    got = torch.cat([a[0], b[0]]).cpu().numpy()
    assert np.concatenate([a[1], b[1]]).tolist() == elo and np.concatenate([a[2], b[2]]).tolist() == ehi
    bad = [i for i in range(n) if not np.array_equal(got[i], exp[i])]
    assert not bad, bad
    ls.close()


def test_overlapping_input_and_output_are_refused():
    """a run of frames reads input frames again after later outputs have been written (the frame that leaves the running average):
    output frames that lie inside the input batch - or the other way round - are an error, not a wrong result"""
    import ctypes as ct

    import torch

    from librir_amd import device as D
    from librir_amd.low_level.misc import _lib

    n, h, w = 12, 40, 64
    buf = torch.from_numpy(s1_noisy_background(2 * n, h, w, seed=5)).cuda()
    ls = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 4)
    fn = _lib.rir_lossy_step_device
    fn.restype = ct.c_int
    fn.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p]
    frame = h * w * 2
    for shift in (1, n - 1, -3):
        assert fn(ls.handle, buf.data_ptr() + 4 * frame, buf.data_ptr() + (4 + shift) * frame, n, 0, None, None, None) == -1, shift
    assert fn(ls.handle, buf.data_ptr(), buf.data_ptr() + n * frame, n, 0, None, None, None) == 0  # (side by side: fine)
    ls.status()
    ls.close()
