"""The cases of tests/test_gpu_resident.py that force a bail-out path through the library's TEST HOOKS (RIR_DEBUG_LOSSY_GIVE_UP,
RIR_DEBUG_LOSSY_BAIL, RIR_DEBUG_ECC_BAIL).  The hooks are compiled into librir_amd_testhooks.so only (-DRIR_TEST_HOOKS, librir_amd/build.py), not
into the product library: each case runs in a process of its own that loads that build (RIR_LIBRARY_VARIANT=testhooks), started by the test.
    RIR_LIBRARY_VARIANT=testhooks python tests/hook_cases.py <case> [argument]"""
import contextlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from librir_amd.synthetic import s1_noisy_background, s3_registration  # noqa: E402


@contextlib.contextmanager
def raises(exc):
    try:
        yield
    except exc:
        return
    raise AssertionError("%s was not raised" % exc.__name__)


class _Path:
    def __init__(self, d):
        self.d = d

    def __truediv__(self, name):
        return os.path.join(self.d, name)


def sticky(tmp_path):
    """A resident run that gives up a wait (forced through RIR_DEBUG_LOSSY_GIVE_UP) has advanced the stream's state with invalid
    frames: the call fails, every later step and status of EVERY stream of that call fails, a stream that was not part of it
    goes on; a saver in that state takes no more frames, writes nothing of the failed chunk or of what was handed in after it and
    closes into a readable file that ends with the last good chunk."""
    import torch

    from librir_amd import device as D
    from librir_amd.video_io import IRMovie, IRSaver

    n, h, w = 12, 64, 96
    fr = [torch.from_numpy(s1_noisy_background(n, h, w, seed=3 + i)).cuda() for i in range(3)]
    a, b, c = (D.LossyStream(w, h, h - 3) for _ in range(3))
    D.LossyStream.step_many([a, b], [fr[0][:4], fr[1][:4]])
    c.step(fr[2][:4])
    os.environ["RIR_DEBUG_LOSSY_GIVE_UP"] = "1"
    with raises(RuntimeError):
        D.LossyStream.step_many([a, b], [fr[0][4:8], fr[1][4:8]])
    del os.environ["RIR_DEBUG_LOSSY_GIVE_UP"]
    for s in (a, b):
        with raises(RuntimeError):
            s.step(fr[0][8:])
        with raises(RuntimeError):
            s.status()
    c.step(fr[2][4:])
    c.status()
    # queue-only calls find it out at the status query, for the member as well as for the leader
    d, e = D.LossyStream(w, h, h - 3), D.LossyStream(w, h, h - 3)
    D.LossyStream.step_many([d, e], [fr[0][:4], fr[1][:4]], errors=False)
    os.environ["RIR_DEBUG_LOSSY_GIVE_UP"] = "1"
    D.LossyStream.step_many([d, e], [fr[0][4:8], fr[1][4:8]], errors=False)
    del os.environ["RIR_DEBUG_LOSSY_GIVE_UP"]
    with raises(RuntimeError):
        e.status()
    with raises(RuntimeError):
        d.status()
    with raises(RuntimeError):
        e.step(fr[1][8:])
    for s in (a, b, c, d, e):
        s.close()

    # the saver: GOP 5, 12 good frames (two chunks written, two frames pending), then a run that gives up
    p = str(tmp_path / "sticky.h264")
    data = s1_noisy_background(30, h, w, seed=9)
    s = IRSaver(p, w, h, h - 3)
    s.set_parameter("GOP", 5)
    for i in range(12):
        s.add_image_lossy(data[i], i)
    os.environ["RIR_DEBUG_LOSSY_GIVE_UP"] = "1"
    failed_at = None
    for i in range(12, 20):
        try:
            s.add_image_lossy(data[i], i)
        except RuntimeError:
            failed_at = i
            break
    del os.environ["RIR_DEBUG_LOSSY_GIVE_UP"]
    # the chunk that completes at frame 14 runs its deferred loss step; nothing waits for it there (round 5: the chunk is in flight while
    # the next one is assembled), so the failure is found when the chunk is collected - by the call that completes the NEXT chunk
    assert failed_at is not None and failed_at <= 19
    for i in range(failed_at, failed_at + 3):
        with raises(RuntimeError):
            s.add_image_lossy(data[i], i)
        with raises(RuntimeError):
            s.add_image(data[i], i)
    s.close()
    with IRMovie.from_filename(p) as mov:
        assert mov.images == 10  # the two complete chunks; nothing of the failed one
        ok = IRSaver(str(tmp_path / "ok.h264"), w, h, h - 3)
        ok.set_parameter("GOP", 5)
        for i in range(10):
            ok.add_image_lossy(data[i], i)
        ok.close()
        with IRMovie.from_filename(str(tmp_path / "ok.h264")) as good:
            assert np.array_equal(mov.data, good.data)


def flying_chunk_fails(tmp_path, how):
    """a chunk in flight that cannot be written - its encode fails, its event cannot be waited for, the encoder left no plausible
    length (forced through RIR_DEBUG_SAVER_FAIL_FLYING=encode / wait / length) - ends the recording with the chunk before it: the saver takes
    no more frames, the frames of the failed chunk and whatever came after leave the books (a chunk that is counted but has no index entry
    would be a hole), and close() leaves a file whose frame count, time stamps and index agree - lossless and bounded-loss frames alike."""
    from librir_amd.video_io import IRMovie, IRSaver

    n, h, w = 40, 64, 96
    data = s1_noisy_background(n, h, w, seed=21)
    for lossy in (False, True):
        p = str(tmp_path / ("fly_%s_%d.h264" % (how, lossy)))
        s = IRSaver(p, w, h, h - 3)
        s.set_parameter("GOP", 5)
        add = (lambda i: s.add_image_lossy(data[i], 1000 * i, {"idx": str(i)})) if lossy else (lambda i: s.add_image(data[i], 1000 * i, {"idx": str(i)}))
        for i in range(12):  # two chunks written or in flight, two frames pending
            add(i)
        os.environ["RIR_DEBUG_SAVER_FAIL_FLYING"] = how
        failed_at = None
        for i in range(12, 26):
            try:
                add(i)
            except RuntimeError:
                failed_at = i
                break
        del os.environ["RIR_DEBUG_SAVER_FAIL_FLYING"]
        assert failed_at is not None, (how, lossy)
        for i in range(failed_at, failed_at + 3):  # sticky: nothing more is taken, by either entry point
            with raises(RuntimeError):
                s.add_image(data[i], 1000 * i)
            with raises(RuntimeError):
                s.add_image_lossy(data[i], 1000 * i)
        s.close()
        with IRMovie.from_filename(p) as mov:
            m = mov.images
            assert m % 5 == 0 and 5 <= m <= 15, (how, lossy, m)  # whole chunks only, all from before the failure
            assert len(mov.timestamps) == m and np.allclose(mov.timestamps, np.arange(m) * 1e-6)
            got = mov.data  # every counted frame is there: no hole
            assert got.shape == (m, h, w)
            if not lossy:
                assert np.array_equal(got, data[:m])
            mov.load_pos(m - 1)
            assert mov.frame_attributes["idx"] in (str(m - 1), str(m - 1).encode())


def multi_repeated_smaller():
    """ecc_run_multi_kernel finds out at its start whether all its workgroups are on the chip (resident_device.h); a launch that is
    not - forced here through RIR_DEBUG_ECC_BAIL: the first attempt of every launch is called off - has written nothing and is
    repeated with half the workgroups per sequence: the tracks are the ones of the undisturbed run, bit for bit."""
    import torch

    from librir_amd.registration import DeviceRegistratorECC

    S, n, h, w = 5, 40, 256, 320
    seqs = [torch.from_numpy(s3_registration(n, h, w, seed=70 + q)[0]).cuda() for q in range(S)]

    def run():
        rs = [DeviceRegistratorECC(0.8, 0.8, shape=(h, w)) for _ in range(S)]
        for q in range(S):
            rs[q].start(seqs[q][0])
        DeviceRegistratorECC.compute_many_multi(rs, [s[1:] for s in seqs], chunk=16)
        return [(r.x, r.y, r.confidences) for r in rs]

    ref = run()
    os.environ["RIR_DEBUG_ECC_BAIL"] = "1"
    got = run()
    del os.environ["RIR_DEBUG_ECC_BAIL"]
    assert got == ref
    assert run() == ref


def loss_run_stepped_again(oracle, bail_group):
    """lossy_run_kernel finds out at its start whether all its workgroups are on the chip (resident_device.h); a group of frames whose
    launch was called off - forced through RIR_DEBUG_LOSSY_BAIL=<group> - has written nothing, poisons the groups queued behind it,
    and a call that waits for the budgets steps the frames from that group on again on the launch-per-frame path: frames and budgets
    are those of the undisturbed run (and of the oracle), the streams stay usable."""
    import torch

    from librir_amd import device as D
    from oracle.pyoracle import OracleLossy

    S, n, h, w, hl = 64, 47, 96, 128, 93  # 64 streams: groups of 32 frames, so the 46 steps after the first frame are two groups
    data = [s1_noisy_background(n, h, w, seed=300 + i) for i in range(S)]
    tens = [torch.from_numpy(d).cuda() for d in data]

    def run(more):
        streams = [D.LossyStream(w, h, hl, 6, 2, 5.0, 8) for _ in range(S)]
        o, lo, hi = D.LossyStream.step_many(streams, tens)
        o2, lo2, hi2 = D.LossyStream.step_many(streams, [t[:more] for t in tens])  # the streams go on afterwards
        for s in streams:
            s.status()
            s.close()
        return [x.cpu().numpy() for x in o], lo.copy(), hi.copy(), [x.cpu().numpy() for x in o2], lo2.copy(), hi2.copy()

    ref = run(9)
    os.environ["RIR_DEBUG_LOSSY_BAIL"] = str(bail_group)
    got = run(9)
    del os.environ["RIR_DEBUG_LOSSY_BAIL"]
    for a, b in zip(got, ref):
        if isinstance(a, list):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
        else:
            assert np.array_equal(a, b)
    for i in (0, 17, 63):
        L = OracleLossy(oracle, w, h, hl, low_err=6, high_err=2, std_factor=5.0, running_average=8)
        exp = np.stack([L.step(data[i][f]) for f in range(n)])
        assert np.array_equal(got[0][i], exp), i


def single_sequence_falls_back():
    """ecc_run_kernel with the same rendezvous: a launch that was called off (RIR_DEBUG_ECC_BAIL: every launch) reports "not run"
    through the host view and the alignment - one frame, or a chunk of frames - is done by the launch-per-iteration kernels, which
    add the same rows in the same order: same track, same iteration counts."""
    import torch

    from librir_amd.registration import DeviceRegistratorECC, find_transform_ecc_translation

    n, h, w = 30, 256, 320
    f, _ = s3_registration(n, h, w, seed=31)
    t = torch.from_numpy(f).cuda()

    def run():
        a = DeviceRegistratorECC(0.8, 0.8, shape=(h, w))
        a.start(t[0])
        a.compute_many(t[1:], chunk=8)
        b = DeviceRegistratorECC(0.8, 0.8, shape=(h, w))
        b.start(t[0])
        for i in range(1, 6):
            b.compute(t[i])
        cc, wm = find_transform_ecc_translation(f[0] / f[0].max(), f[3] / f[3].max())
        return a.x, a.y, a.confidences, b.x, b.y, b.confidences, cc, wm.tolist()

    ref = run()
    os.environ["RIR_DEBUG_ECC_BAIL"] = "1"
    got = run()
    del os.environ["RIR_DEBUG_ECC_BAIL"]
    assert got == ref


def const_pairs(pairs):
    """The streaming kernel of the bounded-loss step exists for 2, 4 and 8 pixels per thread (lossy_const_pairs picks by the number of waves:
    many streams of 640x512 take 8); small test frames only ever see 2.  The hook RIR_LOSSY_CONST_PAIRS forces the others: same frames, same
    budgets as the oracle's."""
    import torch

    from librir_amd import device as D
    from oracle.pyoracle import Oracle

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_lossy import CONST_CASES, _oracle_track

    oracle = Oracle()
    os.environ["RIR_LOSSY_CONST_PAIRS"] = str(pairs)
    for name in ("ra8", "subtract_min", "ra64_longer_than_the_calls", "high_above_low", "ra0"):
        c = CONST_CASES[name]
        arr = s1_noisy_background(c["n"], c["h"], c["w"], seed=47)
        for add_loss in (False, True):
            exp, elo, ehi = _oracle_track(oracle, arr, c["w"], c["h"], c["hl"], c["low"], c["high"], 0.0, c["ra"], c.get("subtract_min", False), add_loss)
            ls = D.LossyStream(c["w"], c["h"], c["hl"], c["low"], c["high"], 0.0, c["ra"], subtract_min=c.get("subtract_min", False))
            t = torch.from_numpy(arr).cuda()
            got, lo, hi = [], [], []
            for c0, c1 in zip(c["cuts"][:-1], c["cuts"][1:]):
                o, l_, h_ = ls.step(t[c0:c1], add_loss=add_loss and c0 > 0)
                got.append(o), lo.append(l_), hi.append(h_)
                if c1 - c0 >= 3:
                    offered, taken = ls.path_stats()
                    assert offered >= 1 and taken == offered, (name, c0, c1, offered, taken)
            assert np.array_equal(torch.cat(got).cpu().numpy(), exp), (name, pairs, add_loss)
            assert np.concatenate(lo).tolist() == elo and np.concatenate(hi).tolist() == ehi, (name, pairs, add_loss)
            ls.close()
    del os.environ["RIR_LOSSY_CONST_PAIRS"]


def spec_pairs(pairs):
    """The speculative form's instantiations of the streaming kernel for 4 and 8 pixels per thread (lossy_const_pairs picks them for many
    streams; small test frames would only ever see 2): forced through RIR_LOSSY_CONST_PAIRS on static scenes with stdFactor 5 - committed,
    frames and budgets the oracle's - and on S1 (the group goes to the general form)."""
    import torch

    from librir_amd import device as D
    from oracle.pyoracle import Oracle

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_lossy import CONST_CASES, _oracle_track
    from test_gpu_lossy_spec import static_scene

    oracle = Oracle()
    os.environ["RIR_LOSSY_CONST_PAIRS"] = str(pairs)
    os.environ["RIR_LOSSY_SPEC_PASSES"] = "8"
    for name in ("ra8", "subtract_min", "ra64_longer_than_the_calls", "high_above_low", "ra0", "ra1"):
        c = CONST_CASES[name]
        h, w, hl = 64, 96, 64 - (c["h"] - c["hl"])
        for scene in ("static", "S1"):
            arr = static_scene(c["n"], h, w, seed=53) if scene == "static" else s1_noisy_background(c["n"], h, w, seed=53)
            for add_loss in (False, True):
                exp, elo, ehi = _oracle_track(oracle, arr, w, h, hl, c["low"], c["high"], 5.0, c["ra"], c.get("subtract_min", False), add_loss)
                ls = D.LossyStream(w, h, hl, c["low"], c["high"], 5.0, c["ra"], subtract_min=c.get("subtract_min", False))
                t = torch.from_numpy(arr).cuda()
                got, lo, hi, committed = [], [], [], 0
                for c0, c1 in zip(c["cuts"][:-1], c["cuts"][1:]):
                    o, l_, h_ = ls.step(t[c0:c1], add_loss=add_loss and c0 > 0)
                    got.append(o), lo.append(l_), hi.append(h_)
                    committed += ls.spec_stats()[2]
                assert np.array_equal(torch.cat(got).cpu().numpy(), exp), (name, scene, pairs, add_loss)
                assert np.concatenate(lo).tolist() == elo and np.concatenate(hi).tolist() == ehi, (name, scene, pairs, add_loss)
                assert committed >= 1 if scene == "static" else True, (name, scene, pairs, add_loss, committed)
                ls.close()
    del os.environ["RIR_LOSSY_CONST_PAIRS"], os.environ["RIR_LOSSY_SPEC_PASSES"]


if __name__ == "__main__":
    from librir_amd.low_level.misc import _LIB_PATH

    assert "testhooks" in os.path.basename(_LIB_PATH), "the cases need the build with the test hooks (RIR_LIBRARY_VARIANT=testhooks)"
    case = sys.argv[1]
    if case == "sticky":
        with tempfile.TemporaryDirectory() as d:
            sticky(_Path(d))
    elif case == "flying_chunk_fails":
        with tempfile.TemporaryDirectory() as d:
            flying_chunk_fails(_Path(d), sys.argv[2])
    elif case == "multi_repeated_smaller":
        multi_repeated_smaller()
    elif case == "const_pairs":
        const_pairs(int(sys.argv[2]))
    elif case == "spec_pairs":
        spec_pairs(int(sys.argv[2]))
    elif case == "loss_run_stepped_again":
        from oracle.pyoracle import Oracle

        loss_run_stepped_again(Oracle(), int(sys.argv[2]))
    elif case == "single_sequence_falls_back":
        single_sequence_falls_back()
    else:
        raise SystemExit("unknown case " + case)
    print("case ok:", " ".join(sys.argv[1:]))
