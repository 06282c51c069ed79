"""GPU: translation-only ECC behind find_transform_ecc_translation / MaskedRegistratorECC against the oracle
(floating point: sums are accumulated in double in a different order, tolerance 1e-4 px / 1e-6 on cc), on the
reference's own recipe (tests/python/test_registration.py:41-59), and through the TSV into the loader."""
import numpy as np
import pytest

from librir_amd.registration import MaskedRegistratorECC, find_transform_ecc_translation
from librir_amd.synthetic import s3_registration
from librir_amd.video_io import IRMovie

pytestmark = pytest.mark.gpu


def norm(a):
    return (a - a.min()) / (a.max() - a.min())


@pytest.mark.parametrize("shape", [(96, 128), (67, 83), (358, 448)])
def test_ecc_matches_oracle(oracle, shape):
    h, w = shape
    f, s = s3_registration(6, h, w)
    ref = norm(oracle.gaussian_filter(f[0], 0.5))
    warp_o, wm = (0.0, 0.0), None
    for i in range(1, 6):
        im = norm(oracle.gaussian_filter(f[i], 0.5))
        tx, ty, cc_o, it = oracle.ecc_translation(ref, im, warp_o)
        warp_o = (tx, ty)
        cc, wm = find_transform_ecc_translation(ref, im, wm)
        assert abs(wm[0, 2] - tx) < 1e-4 and abs(wm[1, 2] - ty) < 1e-4 and abs(cc - cc_o) < 1e-6, i
    # with a mask: only the left three quarters take part
    mask = np.zeros(shape, np.uint8)
    mask[:, : 3 * w // 4] = 1
    im = norm(oracle.gaussian_filter(f[1], 0.5))
    tx, ty, cc_o, _ = oracle.ecc_translation(ref, im, (0.0, 0.0), mask=mask)
    cc, wm = find_transform_ecc_translation(ref, im, None, mask=mask)
    assert abs(wm[0, 2] - tx) < 1e-4 and abs(wm[1, 2] - ty) < 1e-4 and abs(cc - cc_o) < 1e-6


def test_failure_raises_like_opencv():
    img = np.random.default_rng(1).random((32, 32)).astype(np.float32)
    wm = np.eye(2, 3, dtype=np.float32)
    wm[0, 2] = 100
    with pytest.raises(RuntimeError):
        find_transform_ecc_translation(img, img, wm)


def test_registrator_on_the_reference_recipe(tmp_path, oracle):
    """MaskedRegistratorECC(1, 1) on 100 frames 512x640: translations 0..99 by steps of 1 are expected
    (test_registration.py:88-108; upstream's asserts are commented out and round to 5 px).  With upstream's
    loose criterion (correlation change < 1e-3) the iteration stops after one or two steps and lags on the first
    frames and after a change of reference image: within 2 px everywhere; with a tight criterion within 0.15 px."""
    n = 100
    f, s = s3_registration(n)
    reg = MaskedRegistratorECC(1, 1)
    reg.start(f[0])
    for i in range(1, n):
        reg.compute(f[i])
    x, y, c = np.array(reg.x), np.array(reg.y), np.array(reg.confidences)
    ex, ey = np.abs(x - s[:, 0]), np.abs(y - s[:, 1])
    assert len(x) == n and ex.max() <= 2.0 and ey.max() <= 2.0 and np.median(ex) <= 0.25 and np.median(ey) <= 0.5
    assert c.min() > 0.8
    tight = MaskedRegistratorECC(1, 1)
    tight.termination_eps = 1e-7
    tight.start(f[0])
    for i in range(1, n):
        tight.compute(f[i])
    assert np.abs(np.array(tight.x) - s[:, 0]).max() <= 0.15 and np.abs(np.array(tight.y) - s[:, 1]).max() <= 0.15
    # the TSV goes straight into the loader's motion correction
    regfile = tmp_path / "motion.regfile"
    reg.to_reg_file(regfile)
    assert open(regfile).readline().count("\t") == 3
    u16 = np.clip(f, 0, 65535).astype(np.uint16)  # one row per image is required (IRFileLoader.cpp:838-843)
    mov = IRMovie.from_numpy_array(u16)
    mov.registration_file = regfile
    mov.registration = True
    i = 7
    exp = oracle.remove_motion(u16[i], np.float32(reg.x[i]), np.float32(reg.y[i]), rows=512 - 3)
    assert np.array_equal(mov[i], exp)
    mov.close()


def test_device_resident_registrator_equals_the_host_class(monkeypatch):
    """DeviceRegistratorECC keeps the frames in HBM; same kernels, same float32 normalisation: same track.  (The host class
    step by step, as upstream's compute() goes: without the switch it would itself call the device class once per image.)"""
    import torch

    monkeypatch.setenv("RIR_REGISTRATION_STEP_BY_STEP", "1")

    from librir_amd.registration import DeviceRegistratorECC

    n = 40
    f, s = s3_registration(n)
    host = MaskedRegistratorECC(0.7, 0.7)
    host.start(f[0])
    dev = DeviceRegistratorECC(0.7, 0.7)
    t = torch.from_numpy(f).cuda()
    dev.start(t[0])
    for i in range(1, n):
        host.compute(f[i])
        dev.compute(t[i])
    assert np.allclose(dev.x, host.x, rtol=0, atol=1e-5) and np.allclose(dev.y, host.y, rtol=0, atol=1e-5)
    assert np.allclose(dev.confidences, host.confidences, rtol=0, atol=1e-7)


def test_change_of_reference_image_on_a_confidence_drop(monkeypatch):
    """masked_registration_ecc.py:177-189: after 20 frames a frame whose correlation falls below min - 2 std becomes
    the new reference (shifted back by its own translation) and the start value returns to the identity.  Host (step by
    step) and device classes take the same decision and report the same track afterwards."""
    import torch

    monkeypatch.setenv("RIR_REGISTRATION_STEP_BY_STEP", "1")

    from librir_amd.registration import DeviceRegistratorECC

    n = 36
    f, s = s3_registration(n, 256, 320)
    rng = np.random.default_rng(8)
    f = f.copy()
    f[28] += rng.normal(0, 4, f[28].shape).astype(np.float32)  # one much noisier frame: lower correlation
    host = MaskedRegistratorECC(1, 1)
    host.subW, host.subH, host.startX, host.startY = 320, 256, 0, 0  # the class assumes 512x640 frames (:78); use the full small frame
    dev = DeviceRegistratorECC(1, 1, shape=(256, 320))
    host.start(f[0])
    t = torch.from_numpy(f).cuda()
    dev.start(t[0])
    ref_before = host.ref_img.copy()
    for i in range(1, n):
        host.compute(f[i])
        dev.compute(t[i])
    assert host.conf_thresh is not None and not np.array_equal(host.ref_img, ref_before)  # the reference image was replaced
    assert np.allclose(dev.x, host.x, rtol=0, atol=1e-4) and np.allclose(dev.y, host.y, rtol=0, atol=1e-4)
    # (upstream restarts from the identity after the change: with ~30 px of accumulated motion the track does not
    #  recover on this recipe - behaviour mirrored, not judged)


@pytest.mark.parametrize("dtype", ["float32", "uint16"])
def test_compute_many_equals_compute_frame_by_frame(dtype):
    """DeviceRegistratorECC.compute_many: the pre-processing of a chunk of frames runs ahead in shared launches
    (rir_ecc_prepare_frames_device), the alignments stay sequential (rir_ecc_align_prepared_device) - the same operations image
    by image, so the same track to the last bit, through a change of the reference image and across chunk boundaries."""
    import torch

    from librir_amd.registration import DeviceRegistratorECC

    n = 45
    f, s = s3_registration(n, 256, 320)
    f = f.copy()
    f[30] += np.random.default_rng(8).normal(0, 4, f[30].shape).astype(np.float32)  # a confidence drop: the reference changes
    t = torch.from_numpy(f if dtype == "float32" else np.clip(f, 0, 65535).astype(np.uint16)).cuda()
    one = DeviceRegistratorECC(0.8, 0.8, shape=(256, 320))
    one.start(t[0])
    for i in range(1, n):
        one.compute(t[i])
    many = DeviceRegistratorECC(0.8, 0.8, shape=(256, 320))
    many.start(t[0])
    shifts = many.compute_many(t[1:], chunk=16)
    assert len(shifts) == n - 1
    assert many.x == one.x and many.y == one.y and many.confidences == one.confidences
    if dtype == "float32":
        assert one.conf_thresh is not None and min(one.confidences[21:]) < one.conf_thresh  # the reference did change


def test_one_launch_per_alignment_equals_two_launches_per_iteration():
    """ecc_run_kernel (all iterations of an alignment in one launch, the rows of sums handed between workgroups inside it) adds
    the same rows in the same order as ecc_sums_kernel + ecc_solve_kernel: same translation, correlation and iteration count."""
    import ctypes as ct
    import subprocess
    import sys

    code = r'''
import ctypes as ct, json, sys
import numpy as np, torch
from librir_amd.registration.device_registration import _lib, _stream
from librir_amd.synthetic import s3_registration
f, s = s3_registration(6, 256, 320)
norm = lambda a: ((a - a.min()) / (a.max() - a.min())).astype(np.float32)
ref = torch.from_numpy(norm(f[0])).cuda()
out = []
for i in range(1, 6):
    im = torch.from_numpy(norm(f[i])).cuda()
    warp = np.zeros(2, np.float32); cc = ct.c_double(0); it = ct.c_int(0)
    assert _lib.rir_ecc_translation_device(ref.data_ptr(), im.data_ptr(), None, 320, 256, warp.ctypes.data, 200, 1e-6, ct.byref(cc), ct.byref(it), _stream()) == 0
    out.append([float(warp[0]).hex(), float(warp[1]).hex(), cc.value.hex(), it.value])
print(json.dumps(out))
'''
    import json
    import os

    res = []
    for env in ({}, {"RIR_ECC_LAUNCH_PER_ITERATION": "1"}):
        e = dict(os.environ, **env)
        e.pop("RIR_ECC_LAUNCH_PER_ITERATION", None) if not env else None
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert p.returncode == 0, p.stderr[-2000:]
        res.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]
    assert all(r[3] > 2 for r in res[0])


def test_a_failing_frame_stops_compute_many_where_compute_stops():
    """A frame whose alignment fails (an image of NaNs: the correlation is NaN) raises from compute_many as it
    does from compute, with the frames before it recorded and nothing after it - the chunk's launch stops at the first failure."""
    import torch

    from librir_amd.registration import DeviceRegistratorECC

    n, bad = 30, 17
    f, s = s3_registration(n, 128, 160)
    f = f.copy()
    f[bad] = np.nan
    t = torch.from_numpy(f).cuda()
    one = DeviceRegistratorECC(1, 1, shape=(128, 160))
    one.start(t[0])
    with pytest.raises(RuntimeError):
        for i in range(1, n):
            one.compute(t[i])
    many = DeviceRegistratorECC(1, 1, shape=(128, 160))
    many.start(t[0])
    with pytest.raises(RuntimeError):
        many.compute_many(t[1:], chunk=12)
    assert len(one.x) == bad and many.x == one.x and many.y == one.y and many.confidences == one.confidences
    # both go on from there with the next frames
    one.compute(t[bad + 1])
    many.compute_many(t[bad + 1:bad + 2])
    assert many.x == one.x and many.confidences == one.confidences


@pytest.mark.perf
def test_tracked_sequence_rate():
    """The review's mark for the registration of a tracked sequence on frames in HBM: >= 12 000 frames/s at 640x512 (a chunk of
    frames is pre-processed in shared launches and aligned in one launch; measured 18-24 k, `tests/perf/ecc_time.py`).  Best of three."""
    import time

    import torch

    from librir_amd.registration import DeviceRegistratorECC

    n = 100
    f, s = s3_registration(n, 512, 640)
    t = torch.from_numpy(f).cuda()
    best = 0.0
    for _ in range(3):
        reg = DeviceRegistratorECC(1, 1)
        reg.start(t[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reg.compute_many(t[1:])
        best = max(best, (n - 1) / (time.perf_counter() - t0))
    print("ECC registration, tracked sequence: %.0f frames/s" % best)
    assert len(reg.x) == n
    assert best >= 12000, best


@pytest.mark.parametrize("S,chunk", [(8, 16), (3, 32), (1, 7)])
def test_sequences_side_by_side_equal_their_own_tracks(S, chunk):
    """DeviceRegistratorECC.compute_many_multi (rir_ecc_align_multi_device, ecc_run_multi_kernel): S independent sequences aligned
    in shared resident launches - each one's track is, to the last bit, the track its own compute_many gives (same rows of
    partial sums added in the same order, whatever the number of sequences and of workgroups per sequence), through a change of
    the reference image in one of them and across chunk boundaries."""
    import torch

    from librir_amd.registration import DeviceRegistratorECC

    n, h, w = 45, 256, 320
    seqs = []
    for q in range(S):
        f, _ = s3_registration(n, h, w, seed=100 + q)
        f = f.copy()
        if q == 1 or S == 1:
            f[30] += np.random.default_rng(8 + q).normal(0, 4, f[30].shape).astype(np.float32)  # a confidence drop: this sequence changes its reference
        seqs.append(torch.from_numpy(f if q % 2 == 0 else np.clip(f, 0, 65535).astype(np.uint16)).cuda())
    solo = []
    for q in range(S):
        r = DeviceRegistratorECC(0.8, 0.8, shape=(h, w))
        r.start(seqs[q][0])
        r.compute_many(seqs[q][1:], chunk=chunk)
        solo.append(r)
    multi = [DeviceRegistratorECC(0.8, 0.8, shape=(h, w)) for _ in range(S)]
    for q in range(S):
        multi[q].start(seqs[q][0])
    shifts = DeviceRegistratorECC.compute_many_multi(multi, [s[1:] for s in seqs], chunk=chunk)
    assert [len(s) for s in shifts] == [n - 1] * S
    for q in range(S):
        assert multi[q].x == solo[q].x and multi[q].y == solo[q].y and multi[q].confidences == solo[q].confidences, q
    if S > 1:
        assert solo[1].conf_thresh is not None and min(solo[1].confidences[21:]) < solo[1].conf_thresh  # the reference of sequence 1 did change
        assert solo[0].x != solo[2].x  # (the sequences are different data)


def test_side_by_side_with_ragged_lengths_an_empty_and_a_failing_sequence():
    """rir_ecc_align_multi_device through the C ABI with what compute_many_multi never gives it: sequences of different lengths, one
    without images, an odd number of sequences (the last pair holds one), and one whose third image is NaN - it stops there (good = 2),
    its pair partner and everybody else go on; every sequence's results equal those of rir_ecc_align_prepared_frames_device on its own."""
    import ctypes as ct

    import torch

    from librir_amd.registration import DeviceRegistratorECC
    from librir_amd.registration import device_registration as DR

    h, w = 192, 256
    lengths = [9, 4, 0, 7, 1]
    S = len(lengths)
    st = DR._stream()
    regs, norms = [], []
    for q in range(S):
        f, _ = s3_registration(max(lengths[q], 1) + 1, h, w, seed=300 + q)
        f = f.copy()
        if q == 3:
            f[3][:] = np.nan  # (image 2 of the frames handed to the alignment)
        t = torch.from_numpy(f).cuda()
        r = DeviceRegistratorECC(1, 1, shape=(h, w))
        r.start(t[0])
        buf = torch.zeros((3, max(lengths[q], 1), r.subH, r.subW), dtype=torch.float32, device="cuda")
        if lengths[q]:
            r._prepare(t, 1, lengths[q], buf, st)
        regs.append(r)
        norms.append(buf)
    torch.cuda.synchronize()
    stride = max(lengths)
    ptr = lambda ts: (ct.c_void_p * S)(*[x.data_ptr() for x in ts])  # noqa: E731
    res = np.full((S, stride, 4), -7.0, np.float64)
    warps = np.zeros((S, 2), np.float32)
    counts = (ct.c_int * S)(*lengths)
    good = (ct.c_int * S)()
    r0 = regs[0]
    assert DR._lib.rir_ecc_align_multi_device(ptr([r._ref_n for r in regs]), ptr([b[0] for b in norms]), ptr([b[1] for b in norms]),
                                             ptr([b[2] for b in norms]), r0.subW, r0.subH, S, counts, warps.ctypes.data, r0.number_of_iterations,
                                             r0.termination_eps, res.ctypes.data, stride, good, st) == 0
    assert list(good) == [9, 4, 0, 2, 1]
    for q in range(S):
        if not lengths[q]:
            continue
        own = np.full((lengths[q], 4), -7.0, np.float64)
        regs[q].warp[:] = 0
        g = DR._lib.rir_ecc_align_prepared_frames_device(regs[q]._ref_n.data_ptr(), norms[q][0].data_ptr(), norms[q][1].data_ptr(), norms[q][2].data_ptr(),
                                                         r0.subW, r0.subH, lengths[q], regs[q].warp.ctypes.data, r0.number_of_iterations, r0.termination_eps,
                                                         own.ctypes.data, st)
        assert g == good[q], q
        assert np.array_equal(own[:g], res[q, :g]), q
        assert np.all(res[q, g:] == -7.0)  # (nothing is written behind the last good image)
        if g:
            assert warps[q, 0] == np.float32(own[g - 1, 0]) and warps[q, 1] == np.float32(own[g - 1, 1])


def test_pre_processing_under_the_alignments_gives_the_same_arrays_and_results():
    """rir_ecc_align_multi_overlapped_device: the jobs it runs beside the resident launch leave the arrays rir_ecc_prepare_frames_device
    leaves, the alignments' results are those of rir_ecc_align_multi_device, and work queued on the caller's stream after the call sees
    the jobs' outputs (the stream is ordered behind them)."""
    import ctypes as ct

    import torch

    from librir_amd.registration import DeviceRegistratorECC
    from librir_amd.registration import device_registration as DR

    h, w, S, k = 200, 264, 4, 6
    st = DR._stream()
    regs, now, nxt, want, frames = [], [], [], [], []
    for q in range(S):
        f, _ = s3_registration(2 * k + 1, h, w, seed=400 + q)
        t = torch.from_numpy(f if q % 2 else np.clip(f, 0, 65535).astype(np.uint16)).cuda()
        r = DeviceRegistratorECC(0.75, 0.75, shape=(h, w))
        r.start(t[0])
        a = torch.zeros((3, k, r.subH, r.subW), dtype=torch.float32, device="cuda")
        b = torch.full((3, k, r.subH, r.subW), -1.0, dtype=torch.float32, device="cuda")
        c = torch.zeros_like(a)
        r._prepare(t, 1, k, a, st)
        r._prepare(t, 1 + k, k, c, st)  # what the jobs must produce
        regs.append(r), now.append(a), nxt.append(b), want.append(c), frames.append(t)
    ptr = lambda ts: (ct.c_void_p * S)(*[x.data_ptr() for x in ts])  # noqa: E731
    r0 = regs[0]
    counts = (ct.c_int * S)(*([k] * S))

    def call(jobs, njobs):
        res = np.zeros((S, k, 4), np.float64)
        warps = np.zeros((S, 2), np.float32)
        good = (ct.c_int * S)()
        assert DR._lib.rir_ecc_align_multi_overlapped_device(ptr([r._ref_n for r in regs]), ptr([x[0] for x in now]), ptr([x[1] for x in now]),
                                                            ptr([x[2] for x in now]), r0.subW, r0.subH, S, counts, warps.ctypes.data,
                                                            r0.number_of_iterations, r0.termination_eps, res.ctypes.data, k, good, jobs, njobs, st) == 0
        return res, warps, list(good)

    plain = call((DR.PrepareJob * S)(), 0)
    jobs = (DR.PrepareJob * S)(*[regs[q]._prepare_job(frames[q], 1 + k, k, nxt[q]) for q in range(S)])
    over = call(jobs, S)
    sums = [x.sum().item() for x in nxt]  # (queued on the caller's stream, not synchronised by hand: must already see the jobs' outputs)
    assert plain[2] == over[2] == [k] * S
    assert np.array_equal(plain[0], over[0]) and np.array_equal(plain[1], over[1])
    for q in range(S):
        assert torch.equal(nxt[q], want[q]), q
        assert sums[q] == want[q].sum().item()


@pytest.mark.parametrize("dtype", ["float32", "uint16"])
def test_host_class_one_call_per_image_equals_its_step_by_step_path(monkeypatch, dtype):
    """MaskedRegistratorECC in the plain configuration sends an image up once and does pre-filter, window normalisation and alignment
    in one library call; with RIR_REGISTRATION_STEP_BY_STEP it goes through the five host round trips upstream's compute() makes.
    Same track, confidences, threshold, start matrix and reference window - through a change of the reference image - and an image
    the one-call path does not take (another dtype) carries on step by step from the same state."""
    n = 34
    f, _ = s3_registration(n, 256, 320)
    f = f.copy()
    f[27] += np.random.default_rng(3).normal(0, 4, f[27].shape).astype(np.float32)
    if dtype == "uint16":
        f = np.clip(f, 0, 65535).astype(np.uint16)

    def run(step_by_step):
        if step_by_step:
            monkeypatch.setenv("RIR_REGISTRATION_STEP_BY_STEP", "1")
        else:
            monkeypatch.delenv("RIR_REGISTRATION_STEP_BY_STEP", raising=False)
        r = MaskedRegistratorECC(1, 1)
        r.subW, r.subH, r.startX, r.startY = 320, 256, 0, 0
        r.start(f[0])
        assert (r._dev is None) == step_by_step
        shifts = [r.compute(f[i]) for i in range(1, n - 2)]
        shifts += [r.compute(f[i].astype(np.float64)) for i in range(n - 2, n)]  # (float64: the step-by-step calls from here on)
        assert r._dev is None
        return r, shifts

    a, sa = run(True)
    b, sb = run(False)
    assert a.conf_thresh is not None and min(a.confidences[21:]) < a.conf_thresh  # the reference did change
    assert np.allclose(a.x, b.x, rtol=0, atol=1e-5) and np.allclose(a.y, b.y, rtol=0, atol=1e-5)
    assert np.allclose(a.confidences, b.confidences, rtol=0, atol=1e-7) and np.allclose(sa, sb, rtol=0, atol=1e-5)
    assert np.isclose(a.conf_thresh, b.conf_thresh, rtol=0, atol=1e-7) and np.allclose(a.start_mat, b.start_mat, rtol=0, atol=1e-5)
    assert a.ref_img.shape == b.ref_img.shape and np.allclose(a.ref_img, b.ref_img, rtol=0, atol=1e-3)
    assert all(type(v) is type(w_) for v, w_ in zip(a.x[1:], b.x[1:]))


def test_eight_sequences_at_the_headline_size_equal_their_own_tracks():
    """BASELINE configs[4]'s registration at its full frame size: 8 sequences of 640x512 (full-frame window: 256 rows of partial sums, one
    per compute workgroup and pair of sequences, a service workgroup per sequence, the next chunk's pre-processing under the alignments)
    against each sequence's own compute_many - identical lists - and the known shifts of the recipe within the 2 px upstream's loose
    criterion allows."""
    import torch

    from librir_amd.registration import DeviceRegistratorECC

    S, n, h, w = 8, 41, 512, 640
    seqs, truth = [], []
    for q in range(S):
        f, s = s3_registration(n, h, w, seed=500 + q)
        seqs.append(torch.from_numpy(f).cuda())
        truth.append(s)
    multi = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(S)]
    for q in range(S):
        multi[q].start(seqs[q][0])
    DeviceRegistratorECC.compute_many_multi(multi, [s[1:] for s in seqs], chunk=16)
    for q in range(S):
        solo = DeviceRegistratorECC(1, 1, shape=(h, w))
        solo.start(seqs[q][0])
        solo.compute_many(seqs[q][1:])
        assert multi[q].x == solo.x and multi[q].y == solo.y and multi[q].confidences == solo.confidences, q
        got = np.stack([multi[q].x, multi[q].y], 1)
        assert np.abs(np.abs(got) - np.abs(np.asarray(truth[q])[:, :2])).max() < 2.0, q


def test_manage_computation_and_tries(dev):
    """reference tests/python/test_registration.py:137-147 (and the function, masked_registration_ecc.py:218-245): the recipe's images
    through the retrying wrapper - every image gets a translation; an image that cannot be aligned (not-a-number everywhere: so is its
    correlation) is tried five times, the percentile going down by 0.01 each time, and then takes its predecessor's values."""
    from librir_amd.registration import manage_computation_and_tries

    f32, shifts = s3_registration(12, 512, 640)
    reg = MaskedRegistratorECC(1, 1)
    reg.start(f32[0])
    for img in f32[1:]:
        assert manage_computation_and_tries(img, reg) is reg
    assert len(reg.x) == 12 and abs(reg.x[-1] - shifts[11, 0]) < 0.5 and abs(reg.y[-1] - shifts[11, 1]) < 0.5
    before = (reg.x[-1], reg.y[-1], reg.confidences[-1])
    reg.median = 0.999  # (the dynamic-mask path: the step-by-step body)
    hopeless = np.full((512, 640), np.nan, np.float32)
    with np.errstate(all="ignore"):
        manage_computation_and_tries(hopeless, reg)
    assert len(reg.x) == 13 and (reg.x[-1], reg.y[-1], reg.confidences[-1]) == before
    assert abs(reg.median - (0.999 - 0.05)) < 1e-9
