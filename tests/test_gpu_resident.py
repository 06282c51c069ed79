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


def _hook_case(*args):
    """runs tests/hook_cases.py <case> in a process that loads the build WITH the test hooks (the product library has none)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RIR_LIBRARY_VARIANT="testhooks")
    for k in ("RIR_DEBUG_LOSSY_GIVE_UP", "RIR_DEBUG_LOSSY_BAIL", "RIR_DEBUG_ECC_BAIL", "RIR_DEBUG_SAVER_FAIL_FLYING", "RIR_LOSSY_SPEC_PASSES", "RIR_LOSSY_NO_SPEC",
              "RIR_ABI_ZERO_COPY"):  # (the cases are about the default forms, whatever the suite is run under)
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "hook_cases.py")] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "case ok" in r.stdout, r.stdout[-3000:]


def test_a_run_that_gave_up_is_sticky():
    """A resident run that gives up a wait (forced through the test hook RIR_DEBUG_LOSSY_GIVE_UP) has advanced the stream's state with invalid
    frames: the call fails, every later step and status of EVERY stream of that call fails, a stream that was not part of it
    goes on; a saver in that state takes no more frames, writes nothing of the chunk that was being assembled and closes into a
    readable file that ends with the last complete chunk.  (tests/hook_cases.py: sticky)"""
    _hook_case("sticky")


@pytest.mark.parametrize("how", ["encode", "wait", "length"])
def test_a_chunk_in_flight_that_fails_ends_the_recording_before_it(how):
    """a failure of the saver's chunk in flight - at its encode call, at the wait for its event, at the check of the
    length the encoder left - is sticky and rolls the books back to the chunk's first frame: the file close() leaves has as many frames as
    its index holds, and every one of them reads.  (tests/hook_cases.py: flying_chunk_fails)"""
    _hook_case("flying_chunk_fails", how)


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


def test_a_launch_that_is_not_resident_is_repeated_smaller_with_the_same_results():
    """ecc_run_multi_kernel finds out at its start whether all its workgroups are on the chip (resident_device.h); a launch that is
    not - forced through the test hook RIR_DEBUG_ECC_BAIL: the first attempt of every launch is called off - has written nothing and is
    repeated with half the workgroups per sequence: the tracks are the ones of the undisturbed run, bit for bit.
    (tests/hook_cases.py: multi_repeated_smaller)"""
    _hook_case("multi_repeated_smaller")


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
def test_a_loss_run_that_is_not_resident_is_stepped_again_frame_by_frame(bail_group):
    """lossy_run_kernel finds out at its start whether all its workgroups are on the chip (resident_device.h); a group of frames whose
    launch was called off - forced through the test hook RIR_DEBUG_LOSSY_BAIL=<group> - has written nothing, poisons the groups queued behind it,
    and a call that waits for the budgets steps the frames from that group on again on the launch-per-frame path: frames and budgets
    are those of the undisturbed run (and of the oracle), the streams stay usable.  (tests/hook_cases.py: loss_run_stepped_again)"""
    _hook_case("loss_run_stepped_again", bail_group)


def test_a_single_sequence_launch_that_is_not_resident_falls_back_to_two_launches_per_iteration():
    """ecc_run_kernel with the same rendezvous: a launch that was called off (test hook RIR_DEBUG_ECC_BAIL: every launch) reports "not run"
    through the host view and the alignment - one frame, or a chunk of frames - is done by the launch-per-iteration kernels, which
    add the same rows in the same order: same track, same iteration counts.  (tests/hook_cases.py: single_sequence_falls_back)"""
    _hook_case("single_sequence_falls_back")
