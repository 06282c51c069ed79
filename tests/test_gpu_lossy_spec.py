"""GPU: the SPECULATIVE form of a bounded-loss run - budgets that follow the statistics (stdFactor != 0: the reference's defaults 6 / 2 / 5 / 32,
h264.cpp:1662-1665; budget :2335-2385) guessed, stepped by the streaming kernel into shadow state, verified frame by frame on the device,
corrected and stepped again, or left to the general form.  Whatever path a group takes, frames and budgets are the oracle's, bit for bit;
which path it took is asserted through rir_lossy_spec_stats."""
import numpy as np
import pytest

from librir_amd.synthetic import s1_noisy_background
from test_gpu_lossy import CONST_CASES, _oracle_track

pytestmark = pytest.mark.gpu


def static_scene(n, h, w, seed, sigma=0.7, levels=1000.0):
    """a scene that does not move: a fixed background and sensor noise (the S1 recipe of SURVEY §8d without its +1 level per frame)"""
    rng = np.random.default_rng(seed)
    bg = rng.random((h, w)) * levels
    return np.stack([(bg + 10 + rng.normal(0, sigma, (h, w))).astype(np.uint16) for _ in range(n)])


@pytest.fixture(autouse=True)
def _plain_paths(monkeypatch):
    for k in ("RIR_LOSSY_LAUNCH_PER_FRAME", "RIR_LOSSY_RUN_MAX_WORKGROUPS", "RIR_LOSSY_NO_SPEC", "RIR_LOSSY_SPEC_PASSES", "RIR_LOSSY_RUN_FORM", "RIR_LOSSY_SPEC_FIRST_ONLY",
              "RIR_LOSSY_SPEC_NO_GIVE_UP", "RIR_LOSSY_SPEC_NO_PLANE"):
        monkeypatch.delenv(k, raising=False)


def _run_cuts(ls, t, cuts, add_loss=False):
    """steps the stream call by call; returns frames, budgets and the speculative form's books of every call"""
    import torch

    got, lo, hi, books = [], [], [], []
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        o, l_, h_ = ls.step(t[c0:c1], add_loss=add_loss and c0 > 0)
        got.append(o), lo.append(l_), hi.append(h_)
        books.append(ls.spec_stats())
    return torch.cat(got).cpu().numpy(), np.concatenate(lo).tolist(), np.concatenate(hi).tolist(), books


@pytest.mark.parametrize("add_loss", [False, True], ids=["add_image_lossy", "add_loss"])
@pytest.mark.parametrize("name", list(CONST_CASES))
def test_static_scene_with_default_std_factor_is_committed(oracle, name, add_loss, monkeypatch):
    """Every case of the constant-budget tests (ring lengths 0 .. 64, short calls, subtractMin, high above low, both decision variants) with
    stdFactor 5 on a scene that does not move: the guess verifies at the first pass - or, where a small frame's statistic moves a budget by
    one level now and then, one pass later per such frame - every group is committed, and frames, budgets and the state handed from call to
    call are the oracle's."""
    import torch

    from librir_amd import device as D

    monkeypatch.setenv("RIR_LOSSY_SPEC_PASSES", "8")
    c = CONST_CASES[name]
    h, w, hl = 64, 96, 64 - (c["h"] - c["hl"])  # (frames large enough for a steady statistic: 40x64 scenes move their own budgets every few frames)
    arr = static_scene(c["n"], h, w, seed=71)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, c["low"], c["high"], 5.0, c["ra"], c.get("subtract_min", False), add_loss)
    high = max(c["high"], 0)
    guess = (max(c["low"], high), high)
    moved = [i for i in range(c["n"]) if (elo[i], ehi[i]) != guess]
    assert len(moved) <= 3, "the scene moves its budgets: not what this test is about"
    ls = D.LossyStream(w, h, hl, c["low"], c["high"], 5.0, c["ra"], subtract_min=c.get("subtract_min", False))
    got, lo, hi, books = _run_cuts(ls, torch.from_numpy(arr).cuda(), c["cuts"], add_loss)
    assert np.array_equal(got, exp)
    assert lo == elo and hi == ehi
    for (c0, c1), (through, offered, committed, passes) in zip(zip(c["cuts"][:-1], c["cuts"][1:]), books):
        steps = (c1 - c0) - (1 if c0 == 0 else 0)
        if c1 - c0 >= 3 and steps >= 2:
            here = sum(1 for i in moved if c0 <= i < c1)
            assert through == 1 and offered == 1 and committed == 1 and (passes == 1 if here == 0 else 2 <= passes <= 1 + here), \
                (name, c0, c1, through, offered, committed, passes, moved)
    assert ls.path_stats() == (0, 0)  # (the constant-budget form's books: not its call)
    ls.close()


def test_events_in_a_static_scene_take_further_passes(oracle, monkeypatch):
    """A weak flash moves the budget of one or two frames; a step change of the whole scene moves the budgets of the 40 frames after it (its echo in
    the window's mean).  A pass corrects EVERY budget behind the first wrong one with what it computed from its (there only roughly right) sums: both
    are committed at the second or third pass.  With the first form of the correction (RIR_LOSSY_SPEC_FIRST_ONLY: one frame per pass) the flash
    takes a pass per frame it moved and the step is left to the general form.  Same frames and budgets as the oracle's every time."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 150, 64, 96, 61
    base = static_scene(n, h, w, seed=41)
    flash = base.copy()
    flash[80, 20:40, 30:60] += 5
    step = base.copy()
    step[70:] += 50
    for name, arr in (("flash", flash), ("step", step)):
        exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 32)
        moved = [i for i in range(n) if (elo[i], ehi[i]) != (6, 2)]
        assert (1 <= len(moved) <= 6) if name == "flash" else len(moved) >= 30, (name, moved)
        for first_only in (False, True):
            monkeypatch.setenv("RIR_LOSSY_SPEC_PASSES", "8")
            if first_only:
                monkeypatch.setenv("RIR_LOSSY_SPEC_FIRST_ONLY", "1")
            else:
                monkeypatch.delenv("RIR_LOSSY_SPEC_FIRST_ONLY", raising=False)
            ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 32)
            got, lo, hi, books = _run_cuts(ls, torch.from_numpy(arr).cuda(), [0, n])
            assert np.array_equal(got, exp), (name, first_only)
            assert lo == elo and hi == ehi, (name, first_only)
            through, offered, committed, passes = books[0]
            assert through == 1 and offered == 1, books
            if not first_only:
                assert committed == 1 and 2 <= passes <= 4, (name, books, moved)
            elif name == "flash":
                assert committed == 1 and passes == len(moved) + 1, (name, books, moved)
            else:
                assert committed == 0, (name, books)
            ls.close()
    monkeypatch.delenv("RIR_LOSSY_SPEC_FIRST_ONLY", raising=False)


def test_budgets_that_move_every_frame_fall_back_and_the_stream_backs_off(oracle):
    """The S1 recipe (one level up per frame) under the default parameters moves its budget nearly every frame: nothing to guess.  The group
    is left to the general form at its first pass, and the stream is not offered again for 3, 15, 63 groups.  Bit-exact throughout."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 241, 64, 96, 61
    arr = s1_noisy_background(n, h, w, seed=11)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 32)
    assert sum(1 for i in range(1, n) if (elo[i], ehi[i]) != (6, 2)) > n // 2, "the stream does not move its budgets"
    ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 32)
    cuts = [0] + list(range(1, n + 1, 30))
    got, lo, hi, books = _run_cuts(ls, torch.from_numpy(arr).cuda(), cuts)
    assert np.array_equal(got, exp)
    assert lo == elo and hi == ehi
    # calls of 30 frames = one group each: failed, skipped x 3, failed, skipped x 15 ...
    assert [b[:3] for b in books[1:]] == [(1, 1, 0), (1, 0, 0), (1, 0, 0), (1, 0, 0), (1, 1, 0), (1, 0, 0), (1, 0, 0), (1, 0, 0)], books
    assert all((1 <= b[3] <= 3) if b[1] else b[3] == 0 for b in books[1:]), books  # (given up at the first pass: dozens of frames off the table)
    ls.close()


def test_streams_of_one_call_are_committed_together_or_not_at_all(oracle):
    """Streams that share their launches: every stream's guess must verify for the group to be committed.  Static scenes side by side are
    committed; with one stream that moves its budgets among them the group goes to the general form for all - every stream its own oracle's."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 90, 64, 96, 61
    params = [(6, 2, 5.0, 32), (4, 2, 2.5, 8), (5, 1, 5.0, 0), (3, 3, 0.0, 4)]  # (the last one: stdFactor 0 among the others - its guess is its budget)
    for mover in (None, 2):
        arrs = [static_scene(n, h, w, seed=80 + i) for i in range(len(params))]
        if mover is not None:
            arrs[mover] = s1_noisy_background(n, h, w, seed=90)
        exps = [_oracle_track(oracle, arrs[i], w, h, hl, *params[i]) for i in range(len(params))]
        streams = [D.LossyStream(w, h, hl, *p) for p in params]
        ins = [torch.from_numpy(a).cuda() for a in arrs]
        outs, lo, hi = D.LossyStream.step_many(streams, ins)
        through, offered, committed, passes = streams[0].spec_stats()
        assert through == 1 and offered == 1 and committed == (1 if mover is None else 0), (mover, through, offered, committed, passes)
        for i in range(len(params)):
            assert np.array_equal(outs[i].cpu().numpy(), exps[i][0]), (mover, i)
            assert lo[i].tolist() == exps[i][1] and hi[i].tolist() == exps[i][2], (mover, i)
        for s_ in streams:
            s_.close()


def test_full_size_static_stream_with_defaults(oracle):
    """640x512, the reference's default parameters, calls of 150 frames (one group each, the ring filling inside the first): committed, bit-exact."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 300, 512, 640, 509
    arr = static_scene(n, h, w, seed=5)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 32)
    ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 32)
    got, lo, hi, books = _run_cuts(ls, torch.from_numpy(arr).cuda(), [0, 150, 300])
    assert np.array_equal(got, exp)
    assert lo == elo and hi == ehi
    assert [b[:3] for b in books] == [(1, 1, 1), (1, 1, 1)], books
    ls.close()


def test_switching_the_form_off_changes_nothing(oracle, monkeypatch):
    """RIR_LOSSY_NO_SPEC: the general form alone - the same frames and budgets (what the speculative form is checked against in the field)."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 80, 64, 96, 61
    arr = static_scene(n, h, w, seed=3)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 32)
    monkeypatch.setenv("RIR_LOSSY_NO_SPEC", "1")
    ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 32)
    got, lo, hi, books = _run_cuts(ls, torch.from_numpy(arr).cuda(), [0, n])
    assert np.array_equal(got, exp) and lo == elo and hi == ehi
    assert books[0] == (0, 0, 0, 0)
    ls.close()


def test_default_parameters_through_the_saver(tmp_path, oracle):
    """h264_add_image_lossy with the reference's defaults on a static scene: the saver's deferred runs take the same launches; the file holds
    the oracle's frames and the per-frame budgets it reports are the oracle's."""
    from librir_amd.video_io import IRMovie, IRSaver

    n, h, w, hl = 130, 64, 96, 61
    arr = static_scene(n, h, w, seed=23)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 32)
    dst = tmp_path / "static.h264"
    with IRSaver(dst, w, h, hl) as s:
        for i in range(n):
            s.add_image_lossy(arr[i], i * 1000)
        low, high = list(s.get_low_errors()), list(s.get_high_errors())
    assert low == elo and high == ehi
    with IRMovie.from_filename(dst) as mov:
        assert np.array_equal(mov.data, exp)


@pytest.mark.parametrize("pairs", [2, 4])
def test_speculative_form_with_more_pixels_per_thread(pairs):
    """The streaming kernel's speculative instantiations for 4 and 8 pixels per thread, forced through the test hook RIR_LOSSY_CONST_PAIRS in a
    process that loads the build with the hooks (tests/hook_cases.py: spec_pairs): same frames, same budgets as the oracle's."""
    from test_gpu_resident import _hook_case

    _hook_case("spec_pairs", pairs)


@pytest.mark.parametrize("S", [8, 17])
def test_speculative_form_full_size_many_streams(oracle, S):
    """Eight and seventeen 640x512 streams of static scenes with stdFactor != 0 in one call - the launches that take 8 pixels per thread by themselves -
    rings of several lengths, subtractMin, different budgets and factors side by side: committed together, each stream its own oracle's.
    (Round 6: this is the case that showed a 16-byte buffer store's data overwritten by the vector instruction behind it - lossy_kernels.hip, buf_stn -
    as two wrong pixels per thread in a few waves of the last frames; with eight streams on every run.)"""
    import torch

    from librir_amd import device as D

    h, w, n = 512, 640, 56
    base = static_scene(n, h, w, seed=29)
    streams, ins, exps = [], [], []
    for i in range(S):
        ra, smin, sf = (0, 3, 8, 32)[i % 4], i % 3 == 0, (5.0, 2.5, 1.0)[i % 3]
        arr = base if i == 0 else (base + np.uint16(7 * i)).astype(np.uint16)
        if i % 5 == 2:
            arr = arr[:, ::-1].copy()
        exps.append(_oracle_track(oracle, arr, w, h, h - 3, 4 + i % 3, 2, sf, ra, smin))
        streams.append(D.LossyStream(w, h, h - 3, 4 + i % 3, 2, sf, ra, subtract_min=smin))
        ins.append(torch.from_numpy(arr).cuda())
    o1, lo1, hi1 = D.LossyStream.step_many(streams, [t[:1] for t in ins])
    o2, lo2, hi2 = D.LossyStream.step_many(streams, [t[1:] for t in ins])
    through, offered, committed, passes = streams[0].spec_stats()
    moved = max(sum(1 for k in range(1, n) if (e[1][k], e[2][k]) != (e[1][0], e[2][0])) for e in exps)
    assert through >= 1 and offered == through and (committed == through or moved >= 3), (through, offered, committed, passes, moved)
    for i in range(S):
        exp, elo, ehi = exps[i]
        got = torch.cat([o1[i], o2[i]]).cpu().numpy()
        bad = [k for k in range(n) if not np.array_equal(got[k], exp[k])]
        assert not bad, (i, bad)
        assert np.concatenate([lo1[i], lo2[i]]).tolist() == elo and np.concatenate([hi1[i], hi2[i]]).tolist() == ehi, i
    for s_ in streams:
        s_.close()


@pytest.mark.parametrize("shape", [(768, 1024, 765), (600, 800, 600), (513, 648, 511)], ids=["1024x768", "800x600", "648x513"])
def test_speculative_form_other_full_sizes(oracle, shape):
    """configs[3]'s geometry and two others (a lossy height that is the whole frame; a width that is a multiple of 8 but not of 64): 70 frames in
    two calls, ring of 8, subtractMin, the reference's default errors and factor - committed, frames and budgets the oracle's."""
    import torch

    from librir_amd import device as D

    h, w, hl = shape
    n = 70
    arr = static_scene(n, h, w, seed=31)
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 8, True)
    ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 8, subtract_min=True)
    got, lo, hi, books = _run_cuts(ls, torch.from_numpy(arr).cuda(), [0, 9, n])
    bad = [k for k in range(n) if not np.array_equal(got[k], exp[k])]
    assert not bad, bad
    assert lo == elo and hi == ehi
    moved = sum(1 for k in range(n) if (elo[k], ehi[k]) != (6, 2))
    assert all(b[0] == 1 and b[1] == 1 for b in books) and (moved > 2 or all(b[2] == 1 for b in books)), (books, moved)
    ls.close()


def test_sums_of_large_differences_take_the_exact_path(oracle):
    """The sums kernel adds two pixels to an instruction as long as every difference is below 4 096; a wave that meets a larger one does its pixels
    one by one, with the reference's own arithmetic (the square wraps at 32 bits and is added as a signed number: differences of 46 341 levels and
    more).  A tiny stdFactor keeps the budgets at the guess whatever the statistic, so the groups are committed and their statistics go into the
    40-frame window; the factor is then raised, and every later budget depends on what the window holds: frames and budgets the oracle's."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 120, 64, 96, 61
    arr = static_scene(n, h, w, seed=37).astype(np.int64)
    arr[20:23, 10:30, 20:70] += 50000  # differences beyond 46 340: the wrapped square is negative
    arr[45, 40:50, :] += 5000          # ... and beyond 4 096 only
    arr = np.clip(arr, 0, 65535).astype(np.uint16)
    changes = {60: (6, 2, 5.0)}
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 1e-9, 8, changes=changes)
    ls = D.LossyStream(w, h, hl, 6, 2, 1e-9, 8)
    t = torch.from_numpy(arr).cuda()
    a = ls.step(t[:60])
    assert ls.spec_stats()[2] == 1, ls.spec_stats()  # committed: its window entries are the sums kernel's
    ls.set_errors(6, 2, 5.0)
    b = ls.step(t[60:])
    got = torch.cat([a[0], b[0]]).cpu().numpy()
    assert np.array_equal(got, exp)
    assert np.concatenate([a[1], b[1]]).tolist() == elo and np.concatenate([a[2], b[2]]).tolist() == ehi
    assert any(e != 6 for e in elo[60:100]), "the window's content never mattered: the test stream does not do what it is about"
    ls.close()


def test_speculative_form_declines_what_it_must_not_take(oracle):
    """An empty foreground or background makes the reference's statistic 0 / 0, and the NaN decides the budgets of that frame and of the 39 after it:
    groups with a frame whose classes are not surely both there, and groups stepped while a NaN sits in the window, are not offered to the
    speculative form (the precondition of the constant-budget form) and the general form steps them; afterwards it is offered again.  stdFactor 5;
    same frames and budgets as the oracle throughout."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 260, 64, 96, 64
    arr = static_scene(n, h, w, seed=47)
    arr[60:64] = 1000  # uniform frames: everything in the mode bin, no foreground (from frame 41 on the statistic is split)
    arr[64] = np.where(np.arange(h * w).reshape(h, w) % 2 == 0, 1000, 1001)  # two levels inside one bin: nothing above, nothing below
    exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 4)
    ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 4)
    cuts = [0, 1, 50, 70, 90, 140, 200, n]
    got, lo, hi, books = _run_cuts(ls, torch.from_numpy(arr).cuda(), cuts)
    assert np.array_equal(got, exp)
    assert lo == elo and hi == ehi
    offered = {(c0, c1): b[1] for (c0, c1), b in zip(zip(cuts[:-1], cuts[1:]), books)}
    # the call with the uniform frames and the calls while their NaN is in the 40-frame window (frames 60 .. 103) are not offered
    assert offered[(1, 50)] == 1 and offered[(50, 70)] == 0 and offered[(70, 90)] == 0 and offered[(90, 140)] == 0, offered
    assert offered[(140, 200)] == 1 or offered[(200, n)] == 1, offered  # (offered again once the window is clean - unless it backs off)
    ls.close()


@pytest.mark.perf
def test_speculative_form_rate_on_a_static_scene():
    """Rate floor (not part of the parity run): the reference's default parameters on a 640x512 scene that does not move, 200-frame calls, frames in
    HBM - committed by the speculative form at 0.82 M frames/s (1.4 M in 1 000-frame calls; the general form 0.17 M:
    tests/perf/lossy_spec_time.py).  The floor leaves a third of margin for a busy box; best of three."""
    import time

    import torch

    from librir_amd import device as D

    n, h, w = 200, 512, 640
    fr = torch.from_numpy(static_scene(n, h, w, seed=9)).cuda()
    st = D.LossyStream(w, h, h - 3, 6, 2, 5.0, 32)
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
    books = st.spec_stats()
    st.close()
    print("bounded-loss step, defaults, static scene, 200-frame calls: %.0f frames/s" % best)
    assert books[:3] == (1, 1, 1), books  # (offered to and committed by the speculative form)
    assert best >= 450000, best


def test_byte_plane_and_its_fallback_give_the_same_sums(oracle, monkeypatch):
    """The sums kernel takes a frame's sums from the byte plane the streaming kernel leaves (difference in 7 bits + class, LossySpec::dplane) - or from
    the frames themselves where there is no plane (RIR_LOSSY_SPEC_NO_PLANE) or a pass met a difference of 128 or more (a step of 300 levels here:
    decided on the device, pass by pass).  Small and large events, with and without the plane: the oracle's frames and budgets, the same books."""
    import torch

    from librir_amd import device as D

    n, h, w, hl = 120, 64, 96, 61
    base = static_scene(n, h, w, seed=43)
    small, large = base.copy(), base.copy()
    small[60:] += 40    # every difference below 128: the plane holds the pass
    large[60:] += 300   # differences of 300 at frame 60: that pass's sums come from the frames
    for name, arr in (("small", small), ("large", large)):
        exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, 6, 2, 5.0, 32)
        books = {}
        for plane in (True, False):
            if plane:
                monkeypatch.delenv("RIR_LOSSY_SPEC_NO_PLANE", raising=False)
            else:
                monkeypatch.setenv("RIR_LOSSY_SPEC_NO_PLANE", "1")
            monkeypatch.setenv("RIR_LOSSY_SPEC_PASSES", "8")
            ls = D.LossyStream(w, h, hl, 6, 2, 5.0, 32)
            got, lo, hi, bk = _run_cuts(ls, torch.from_numpy(arr).cuda(), [0, 30, n])
            ls.close()
            assert np.array_equal(got, exp), (name, plane)
            assert lo == elo and hi == ehi, (name, plane)
            books[plane] = bk
        assert books[True] == books[False], (name, books)
        assert books[True][0][:3] == (1, 1, 1), (name, books)  # (the frames before the step: committed; the call with the step: whatever the passes make of it, the same both ways)
    monkeypatch.delenv("RIR_LOSSY_SPEC_NO_PLANE", raising=False)
