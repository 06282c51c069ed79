"""GPU: the process-wide gate for kernels whose workgroups wait for each other inside a launch (runtime.h: resident launches;
reference threading model: handles of different objects may be driven from different threads, tools.cpp:46-50), and the
sticky failure of a bounded-loss run that gave up."""
import threading
import time

import numpy as np
import pytest

from librir_amd.synthetic import s1_noisy_background, s3_registration

pytestmark = pytest.mark.gpu


def test_three_resident_kernel_families_from_three_threads(dev):
    """7-stream bounded-loss runs (lossy_run_kernel, 1 120 of the chip's ~1 200 places), tracked-sequence alignments
    (ecc_run_kernel: 256 workgroups that poll each other) and single-pass encodes (rirb1_encode_dense: look-back between
    workgroups), each driven from a thread of its own on a stream of its own, at the same time: every result equals the one
    the same call gives alone, no status is raised, and no call comes near the 2 s clock of a wait between workgroups."""
    import torch

    from librir_amd import device as D
    from librir_amd.registration import DeviceRegistratorECC

    S, nl, h, w = 7, 40, 512, 640
    rounds = 6
    lossy_in = [torch.from_numpy(s1_noisy_background(nl, h, w, seed=50 + i)).cuda() for i in range(S)]
    f, _ = s3_registration(61, h, w)
    ecc_in = torch.from_numpy(f).cuda()
    enc_in = torch.from_numpy(s1_noisy_background(300, h, w, seed=77)).cuda()

    def lossy_call():
        streams = [D.LossyStream(w, h, h - 3, 6, 2, 5.0, 8) for _ in range(S)]
        outs, lo, hi = D.LossyStream.step_many(streams, lossy_in)
        for s in streams:
            s.status()
            s.close()
        return [o.cpu().numpy() for o in outs], lo.copy(), hi.copy()

    def ecc_call():
        r = DeviceRegistratorECC(0.7, 0.7, shape=(h, w))
        r.start(ecc_in[0])
        r.compute_many(ecc_in[1:], chunk=20)
        return list(r.x), list(r.y), list(r.confidences)

    ctx = D.CodecContext(w, h, 300, 50)

    def enc_call():
        e = ctx.encode(enc_in, single_pass=True)
        st = ctx.encode_status()
        return st, e.hdr.cpu().numpy().copy(), e.tile_off.cpu().numpy().copy(), e.chunk_off.cpu().numpy().copy(), \
            e.stream[:e.total_words()].cpu().numpy().copy()

    # the single-threaded answers
    ref_lossy, ref_ecc, ref_enc = lossy_call(), ecc_call(), enc_call()
    assert ref_enc[0] == 0
    torch.cuda.synchronize()

    results, errors, longest = {}, [], {}
    start = threading.Barrier(3)

    def worker(name, fn):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                start.wait()
                got, worst = [], 0.0
                for _ in range(rounds):
                    t0 = time.perf_counter()
                    got.append(fn())
                    worst = max(worst, time.perf_counter() - t0)
                results[name], longest[name] = got, worst
        except Exception as e:  # noqa: BLE001
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=worker, args=a) for a in (("lossy", lossy_call), ("ecc", ecc_call), ("enc", enc_call))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    assert set(results) == {"lossy", "ecc", "enc"}
    for outs, lo, hi in results["lossy"]:
        assert all(np.array_equal(a, b) for a, b in zip(outs, ref_lossy[0])) and np.array_equal(lo, ref_lossy[1]) and np.array_equal(hi, ref_lossy[2])
    for r in results["ecc"]:
        assert r == ref_ecc
    for r in results["enc"]:
        assert r[0] == 0 and all(np.array_equal(a, b) for a, b in zip(r[1:], ref_enc[1:]))
    # a wait between workgroups that gives up costs 2 s: nothing came near (the calls include their Python glue and D2H copies)
    assert max(longest.values()) < 1.0, longest


def test_a_run_that_gave_up_is_sticky(dev, monkeypatch, tmp_path):
    """A resident run that gives up a wait (forced through RIR_DEBUG_LOSSY_GIVE_UP) has advanced the stream's state with invalid
    frames: the call fails, every later step and status of EVERY stream of that call fails, a stream that was not part of it
    goes on; a saver in that state takes no more frames, writes nothing of the chunk that was being assembled and closes into a
    readable file that ends with the last complete chunk."""
    import torch

    from librir_amd import device as D
    from librir_amd.video_io import IRMovie, IRSaver

    n, h, w = 12, 64, 96
    fr = [torch.from_numpy(s1_noisy_background(n, h, w, seed=3 + i)).cuda() for i in range(3)]
    a, b, c = (D.LossyStream(w, h, h - 3) for _ in range(3))
    D.LossyStream.step_many([a, b], [fr[0][:4], fr[1][:4]])
    c.step(fr[2][:4])
    monkeypatch.setenv("RIR_DEBUG_LOSSY_GIVE_UP", "1")
    with pytest.raises(RuntimeError):
        D.LossyStream.step_many([a, b], [fr[0][4:8], fr[1][4:8]])
    monkeypatch.delenv("RIR_DEBUG_LOSSY_GIVE_UP")
    for s in (a, b):
        with pytest.raises(RuntimeError):
            s.step(fr[0][8:])
        with pytest.raises(RuntimeError):
            s.status()
    c.step(fr[2][4:])
    c.status()
    # queue-only calls find it out at the status query, for the member as well as for the leader
    d, e = D.LossyStream(w, h, h - 3), D.LossyStream(w, h, h - 3)
    D.LossyStream.step_many([d, e], [fr[0][:4], fr[1][:4]], errors=False)
    monkeypatch.setenv("RIR_DEBUG_LOSSY_GIVE_UP", "1")
    D.LossyStream.step_many([d, e], [fr[0][4:8], fr[1][4:8]], errors=False)
    monkeypatch.delenv("RIR_DEBUG_LOSSY_GIVE_UP")
    with pytest.raises(RuntimeError):
        e.status()
    with pytest.raises(RuntimeError):
        d.status()
    with pytest.raises(RuntimeError):
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
    monkeypatch.setenv("RIR_DEBUG_LOSSY_GIVE_UP", "1")
    failed_at = None
    for i in range(12, 20):
        try:
            s.add_image_lossy(data[i], i)
        except RuntimeError:
            failed_at = i
            break
    monkeypatch.delenv("RIR_DEBUG_LOSSY_GIVE_UP")
    assert failed_at is not None and failed_at <= 15  # the chunk that completes at frame 14 runs its deferred loss step
    for i in range(failed_at, failed_at + 3):
        with pytest.raises(RuntimeError):
            s.add_image_lossy(data[i], i)
        with pytest.raises(RuntimeError):
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


def test_images_kept_by_the_caller_survive_later_reads(tmp_path):
    """an image the caller still holds (or a view of it) is never written to by a later load_image"""
    from librir_amd.video_io import IRMovie, IRSaver

    n, h, w = 30, 64, 96
    data = s1_noisy_background(n, h, w, seed=4)
    p = str(tmp_path / "keep.h264")
    with IRSaver(p, w, h, h) as s:
        for i in range(n):
            s.add_image(data[i], i)
    with IRMovie.from_filename(p) as mov:
        kept = [mov[i] for i in range(8)]
        views = [mov[8 + i][3:9] for i in range(4)]
        for i in range(12, n):
            assert np.array_equal(mov[i], data[i])  # dropped at once: their memory goes round
        for i in range(8):
            assert np.array_equal(kept[i], data[i])
        for i in range(4):
            assert np.array_equal(views[i], data[8 + i][3:9])


def test_a_launch_that_is_not_resident_is_repeated_smaller_with_the_same_results(monkeypatch):
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
    monkeypatch.setenv("RIR_DEBUG_ECC_BAIL", "1")
    got = run()
    monkeypatch.delenv("RIR_DEBUG_ECC_BAIL")
    assert got == ref
    assert run() == ref


def test_alignments_beside_a_flood_of_ordinary_kernels(dev):
    """Ordinary kernels of another thread and stream beside resident launches: they fragment the register files and the LDS, and a
    launch that needs nearly the whole chip may not become resident (resident_device.h).  It must still give its results - through
    the rendezvous and the smaller repeat - and never the error of a 2 s clock."""
    import torch

    from librir_amd import device as D
    from librir_amd.registration import DeviceRegistratorECC

    S, n, h, w = 8, 48, 512, 640
    seqs = [torch.from_numpy(s3_registration(n, h, w, seed=99 + q)[0]).cuda() for q in range(S)]

    def run():
        rs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(S)]
        for q in range(S):
            rs[q].start(seqs[q][0])
        DeviceRegistratorECC.compute_many_multi(rs, [s[1:] for s in seqs])
        return [(r.x, r.y) for r in rs]

    ref = run()
    stop = []

    def flood():
        with torch.cuda.stream(torch.cuda.Stream()):
            x = torch.from_numpy(s1_noisy_background(64, h, w)).cuda()
            k = 0
            while not stop:
                D.gaussian_filter(x, 0.75)
                k += 1
                if k % 8 == 0:
                    torch.cuda.current_stream().synchronize()

    th = threading.Thread(target=flood)
    th.start()
    try:
        time.sleep(0.2)
        with torch.cuda.stream(torch.cuda.Stream()):
            worst = 0.0
            for _ in range(6):
                t0 = time.perf_counter()
                assert run() == ref
                worst = max(worst, time.perf_counter() - t0)
    finally:
        stop.append(1)
        th.join()
    assert worst < 1.5, worst  # (a 2 s clock was never the way out)


@pytest.mark.parametrize("bail_group", [0, 1])
def test_a_loss_run_that_is_not_resident_is_stepped_again_frame_by_frame(oracle, monkeypatch, bail_group):
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
    monkeypatch.setenv("RIR_DEBUG_LOSSY_BAIL", str(bail_group))
    got = run(9)
    monkeypatch.delenv("RIR_DEBUG_LOSSY_BAIL")
    for a, b in zip(got, ref):
        if isinstance(a, list):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
        else:
            assert np.array_equal(a, b)
    for i in (0, 17, 63):
        L = OracleLossy(oracle, w, h, hl, low_err=6, high_err=2, std_factor=5.0, running_average=8)
        exp = np.stack([L.step(data[i][f]) for f in range(n)])
        assert np.array_equal(got[0][i], exp), i


def test_a_single_sequence_launch_that_is_not_resident_falls_back_to_two_launches_per_iteration(monkeypatch):
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
    monkeypatch.setenv("RIR_DEBUG_ECC_BAIL", "1")
    got = run()
    monkeypatch.delenv("RIR_DEBUG_ECC_BAIL")
    assert got == ref
