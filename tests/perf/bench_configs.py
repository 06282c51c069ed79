#!/usr/bin/env python3
"""The BASELINE.json configurations, three numbers each:
  (i)   device-resident kernel fps (frames already in HBM),
  (ii)  batched host -> device -> host fps (pinned host memory, PCIe inclusive),
  (iii) per-frame C-ABI fps (host pointers, one frame per call, as the librir wrapper drives it),
next to the CPU path on the host cores (the compiled reference oracle/_ref for the filters, the oracle
port for the codec; single thread, like the reference's own execution).  One JSON document on stdout.

    python tests/perf/bench_configs.py [--quick]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from librir_amd import device as D  # noqa: E402
from librir_amd.signal_processing import rir_signal_processing as sp  # noqa: E402
from librir_amd.synthetic import inject_bad_pixels, s1_noisy_background, s3_registration  # noqa: E402
from librir_amd.video_io import IRMovie, IRSaver  # noqa: E402
from librir_amd.video_io import rir_video_io as rv  # noqa: E402
from oracle.pyoracle import Oracle, OracleLossy, Ref  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()
O = Oracle()
R = Ref() if Ref.available() else None
dev = torch.device("cuda")
out = {"host_cores": os.cpu_count(), "gpu": torch.cuda.get_device_name(0), "cpu_reference_built": R is not None}


def gpu_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cpu_fps(fn, nframes, budget=6.0):
    t0 = time.perf_counter()
    done = 0
    while True:
        fn()
        done += nframes
        if time.perf_counter() - t0 > budget:
            break
    return done / (time.perf_counter() - t0)


# ------------------------------------------------------------------ configs[1]: lossless codec, 640x512
n, h, w, gop = (200 if args.quick else 1000), 512, 640, 50
fr = s1_noisy_background(n, h, w)
t = torch.from_numpy(fr).to(dev)
ctx = D.CodecContext(w, h, n, gop, device=dev)
dec = torch.empty_like(t)
ctx.place_workspace(t)  # (set-up: the workspace goes where the packing kernel measures fastest for this frames buffer, as in bench.py)


pc = D.PackedCodec(w, h, n, gop, device=dev)


def dev_roundtrip():  # the two-launch step of bench.py: the packed form (the encoded batch = its payload + tables) straight into the decoder
    pc.encode(t)
    pc.decode(out=dec, check=False)


def dev_roundtrip_slotted():  # round 3's step: the encoder's worst-case slots decoded in place
    ctx.encode_tiles(t)
    ctx.decode_slots(out=dec, check=False)


ms = gpu_ms(dev_roundtrip, 10)
assert torch.equal(dec.view(torch.int16), t.view(torch.int16))
c1 = {"workload": "%d x %dx%d u16, S1, GOP %d, encode+decode" % (n, w, h, gop), "device_resident_fps": n / ms * 1e3,
      "encoded_batch_bytes": pc.finish().nbytes(), "raw_bytes": int(t.numel() * 2)}
ms = gpu_ms(dev_roundtrip_slotted, 10)
assert torch.equal(dec.view(torch.int16), t.view(torch.int16))
c1["device_resident_fps_slotted_form"] = n / ms * 1e3
ms_dense = gpu_ms(lambda: ctx.decode(ctx.encode(t), out=dec, check=False), 10)
assert torch.equal(dec.view(torch.int16), t.view(torch.int16))
c1["device_resident_fps_through_the_dense_form"] = n / ms_dense * 1e3
ms = gpu_ms(lambda: ctx.decode(ctx.encode(t, single_pass=True), out=dec, check=False), 10)
assert torch.equal(dec.view(torch.int16), t.view(torch.int16)) and ctx.encode_status() == 0
c1["device_resident_fps_single_pass_encoder"] = n / ms * 1e3
pin_in = torch.from_numpy(fr).pin_memory()
pin_out = torch.empty_like(pin_in)
enc0 = ctx.encode(t)
words = int(enc0.total_words())
pin_stream = torch.empty((words,), dtype=torch.int64).pin_memory()


def batched_roundtrip():
    t.copy_(pin_in, non_blocking=True)
    enc = ctx.encode(t)
    nw = int(enc.total_words())  # sync: the host needs the size to fetch the stream
    pin_stream[:nw].copy_(enc.stream.view(torch.int64)[:nw], non_blocking=True)
    enc.stream.view(torch.int64)[:nw].copy_(pin_stream[:nw], non_blocking=True)  # the stream comes back from the host
    ctx.decode(enc, out=dec, check=False)
    pin_out.copy_(dec, non_blocking=True)


ms = gpu_ms(batched_roundtrip, 5)
assert np.array_equal(pin_out.numpy(), fr)
c1["batched_h2d_d2h_fps"] = n / ms * 1e3
nabi = 100 if args.quick else 1000
with tempfile.TemporaryDirectory() as d:
    rates = []
    for rep in range(4):  # the first recording warms up (page-locked staging is allocated once and pooled); median of the next three, as bench.py does
        dst = os.path.join(d, "abi%d.h264" % rep)
        t0 = time.perf_counter()
        with IRSaver(dst, w, h, h) as s:
            for i in range(nabi):
                s.add_image(fr[i], i * 1000)
        te = time.perf_counter() - t0
        t0 = time.perf_counter()
        with IRMovie.from_filename(dst) as mov:
            for i in range(nabi):
                img = mov[i]
        td = time.perf_counter() - t0
        assert np.array_equal(img, fr[nabi - 1])
        if rep < 3:
            os.remove(dst)
        if rep:
            rates.append(nabi / (te + td))
    c1["per_frame_abi_fps"] = float(np.median(rates))
    c1["per_frame_abi_detail"] = {"record_fps": nabi / te, "read_fps": nabi / td, "file_ratio": fr[:nabi].nbytes / os.path.getsize(dst)}
ncpu = 100


def cpu_codec():
    for c in range(0, ncpu, gop):
        hd, of, st = O.codec_encode_chunk(fr[c:c + gop])
        O.codec_decode_chunk(hd, of, st, w, h)


c1["cpu_port_fps_1thread"] = cpu_fps(cpu_codec, ncpu)
# the reference's own CPU codec that is reachable here: its ZFile container, one zstd frame per image on one host core
# (method 1, level 0 - SURVEY §8d item 2), written and read back through this library's C ABI
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "c1.bin")
    nz = min(200, n)
    t0 = time.perf_counter()
    zw = rv.open_video_write(p, w, h, rate=50, method=rv.METHOD_ZSTD, clevel=0)
    for i in range(nz):
        rv.image_write(zw, fr[i], 3_000_000_000 + i * 20_000_000)
    zsize = rv.close_video(zw)
    tw = time.perf_counter() - t0
    cam = rv.open_camera_file(p)
    t0 = time.perf_counter()
    for i in range(nz):
        img = rv.load_image(cam, i)
    tr = time.perf_counter() - t0
    rv.close_camera(cam)
    assert np.array_equal(img, fr[nz - 1])
    c1["cpu_zfile_zstd_1thread"] = {"record_fps": nz / tw, "read_fps": nz / tr, "roundtrip_fps": nz / (tw + tr),
                                    "file_ratio": fr[:nz].nbytes / zsize}
out["configs[1]"] = c1

# ------------------------------------------------------------------ configs[2]: filters before encode
n2 = 64 if args.quick else 256
fr2 = inject_bad_pixels(s1_noisy_background(n2, h, w), 200)
t2 = torch.from_numpy(fr2).to(dev)
bp = D.BadPixels(t2[0])
ctx2 = D.CodecContext(w, h, n2, gop, device=dev)
offs = torch.tensor([1.25, -2.5], dtype=torch.float32, device=dev)


def chain_unfused(x):
    a = bp.correct(x)
    g = D.gaussian_filter(a, 0.75)                  # uint16 in, float32 out
    tr = D.translate_to_u16(g, offs, "nearest")     # float32 in, uint16 out
    return ctx2.encode(tr)


def chain(x):
    return ctx2.encode(D.filter_chain(x, bp, 0.75, offs, "nearest"))  # the three filters in one pass, bit-identical


ms_unfused = gpu_ms(lambda: chain_unfused(t2), 5)
ms = gpu_ms(lambda: chain(t2), 5)
ms_slots = gpu_ms(lambda: ctx2.encode_tiles(D.filter_chain(t2, bp, 0.75, offs, "nearest")), 5)  # the encoded batch stays on the device: no gather
pc2 = D.PackedCodec(w, h, n2, gop, device=dev)
ms_packed = gpu_ms(lambda: pc2.encode(D.filter_chain(t2, bp, 0.75, offs, "nearest")), 5)  # the packed form: one kernel, payload + tables
assert pc2.status()[0] == 0
# ... with the chain's output and the encoder's workspace each in another placement class than what is read beside them (set-up, untimed)
out2, _ = D.empty_beside(t2, tuple(t2.shape), torch.uint16)
ctx2.place_workspace(out2)
ms_placed = gpu_ms(lambda: ctx2.encode_tiles(D.filter_chain(t2, bp, 0.75, offs, "nearest", out=out2)), 5)
c2 = {"workload": "%d x %dx%d u16, S1 + 200 bad pixels: bad_pixels_correct -> gaussian(0.75) -> translate(1.25,-2.5,nearest) -> encode; filters fused in one kernel (rir_filter_chain_device)" % (n2, w, h),
      "device_resident_fps": n2 / ms_packed * 1e3, "device_resident_fps_dense_file_form": n2 / ms * 1e3, "device_resident_fps_slotted_encode": n2 / ms_slots * 1e3, "device_resident_fps_slotted_encode_buffers_placed": n2 / ms_placed * 1e3,
      "device_resident_fps_unfused_3_filter_kernels": n2 / ms_unfused * 1e3}
pin2 = torch.from_numpy(fr2).pin_memory()


def batched_chain():
    t2.copy_(pin2, non_blocking=True)
    enc = chain(t2)
    nw = int(enc.total_words())
    pin_stream[:nw].copy_(enc.stream.view(torch.int64)[:nw], non_blocking=True)


ms = gpu_ms(batched_chain, 5)
c2["batched_h2d_d2h_fps"] = n2 / ms * 1e3
nabi2 = 20 if args.quick else min(250, fr2.shape[0])
with tempfile.TemporaryDirectory() as d:
    dst = os.path.join(d, "abi2.h264")
    hbp = sp.bad_pixels_create(fr2[0])
    with IRSaver(os.path.join(d, "warm2.h264"), w, h, h) as s:  # (the first calls of a process pay one-off set-up costs)
        for i in range(5):
            s.add_image(sp.translate(sp.gaussian_filter(sp.bad_pixels_correct(hbp, fr2[i]).astype(np.float32), 0.75), 1.25, -2.5, "nearest").astype(np.uint16), i)
    t0 = time.perf_counter()
    with IRSaver(dst, w, h, h) as s:
        for i in range(nabi2):
            a = sp.bad_pixels_correct(hbp, fr2[i])
            g = sp.gaussian_filter(a.astype(np.float32), 0.75)
            tr = sp.translate(g, 1.25, -2.5, "nearest")
            s.add_image(tr.astype(np.uint16), i * 1000)
    c2["per_frame_abi_fps"] = nabi2 / (time.perf_counter() - t0)
    # the three filter calls alone (no recording), as the reference's wrapper is used: the caller converts / gaussian_filter converts (round 6: a
    # uint16 image goes to the uint16 kernel as it is) / the same with the mirror's results kept in the library's page-locked memory (opt-in)
    from librir_amd.low_level.misc import results_in_page_locked_memory

    def three_calls(convert):
        t0_ = time.perf_counter()
        for i in range(nabi2):
            a = sp.bad_pixels_correct(hbp, fr2[i])
            sp.translate(sp.gaussian_filter(a.astype(np.float32) if convert else a, 0.75), 1.25, -2.5, "nearest")
        return nabi2 / (time.perf_counter() - t0_)

    three_calls(False)
    c2["three_filter_calls_fps_caller_converts"] = three_calls(True)
    c2["three_filter_calls_fps"] = three_calls(False)
    was = results_in_page_locked_memory(True)
    three_calls(False)
    c2["three_filter_calls_fps_page_locked_results"] = three_calls(False)
    results_in_page_locked_memory(was)
    sp.bad_pixels_destroy(hbp)
if R is not None:
    import ctypes as ct

    first = np.ascontiguousarray(fr2[0])
    rbp = R.lib.ref_bad_pixels_new(first.ctypes.data, w, h)
    tmp = np.zeros_like(first)

    def cpu_chain():
        for i in range(4):
            img = np.ascontiguousarray(fr2[i])
            R.lib.ref_bad_pixels_correct(rbp, img.ctypes.data, tmp.ctypes.data)
            g = R.gaussian_filter(tmp.astype(np.float32), 0.75)
            R.translate(g, 1.25, -2.5, "nearest")

    c2["cpu_reference_filters_fps_1thread"] = cpu_fps(cpu_chain, 4)
    R.lib.ref_bad_pixels_delete(ct.c_void_p(rbp))
out["configs[2]"] = c2

# ------------------------------------------------------------------ configs[3]: 1024x768, one rank's shard of the 8-GPU job
n3, h3, w3 = (100 if args.quick else 1250), 768, 1024
base = s1_noisy_background(250 if not args.quick else n3, h3, w3, seed=77)
t3 = torch.from_numpy(base).to(dev)
if n3 > base.shape[0]:
    t3 = t3.repeat((n3 + base.shape[0] - 1) // base.shape[0], 1, 1)[:n3].contiguous()  # 250 distinct frames tiled
ctx3 = D.CodecContext(w3, h3, n3, gop, device=dev)
dec3 = torch.empty_like(t3)
ctx3.place_workspace(t3)
ms_sl = gpu_ms(lambda: (ctx3.encode_tiles(t3), ctx3.decode_slots(out=dec3, check=False)), 5)
assert torch.equal(dec3.view(torch.int16), t3.view(torch.int16))
dec3.zero_()
pc3 = D.PackedCodec(w3, h3, n3, gop, device=dev)
ms = gpu_ms(lambda: (pc3.encode(t3), pc3.decode(out=dec3, check=False)), 5)
assert torch.equal(dec3.view(torch.int16), t3.view(torch.int16)) and pc3.status()[0] == 0
out["configs[3]"] = {"workload": "per-GPU shard of the 10 000-frame job: %d x %dx%d u16 (250 distinct S1 frames tiled), encode+decode" % (n3, w3, h3),
                     "device_resident_fps": n3 / ms * 1e3, "raw_GBs": n3 * 4.0 * h3 * w3 / ms / 1e6, "device_resident_fps_slotted_form": n3 / ms_sl * 1e3,
                     "encoded_batch_bytes": pc3.finish().nbytes(), "raw_bytes": int(t3.numel() * 2),
                     "note": "the exchange of the decoded / compressed stream is timed by bench.py --gpus N (value_with_exchange, value_with_compressed_exchange)"}
del t3, dec3, ctx3, pc3
torch.cuda.empty_cache()
# the whole 10 000-frame job on ONE device (15.7 GB of raw frames in its 288 GB): the eight shards of the 8-GPU plan back to back, each
# through the packed form, frames made on the device (tests/test_gpu_configs.py::test_config3_ten_thousand_frames_on_one_device checks them)
if not args.quick and torch.cuda.mem_get_info()[0] > (40 << 30):
    nfull, shard = 10000, 1250
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    bgf = torch.rand((h3, w3), generator=gen, device=dev) * 1000.0
    whole = torch.empty((nfull, h3, w3), dtype=torch.uint16, device=dev)
    for f0 in range(0, nfull, 250):
        ii = torch.arange(f0, f0 + 250, device=dev, dtype=torch.float32).view(-1, 1, 1)
        whole[f0:f0 + 250] = (bgf + 10.0 + ii + torch.randn((250, h3, w3), generator=gen, device=dev) * (0.5 ** 0.5)).to(torch.int32).to(torch.uint16)
    pcs = [D.PackedCodec(w3, h3, shard, gop, device=dev) for _ in range(nfull // shard)]  # (every shard keeps its encoded batch, as every rank would)
    back = torch.empty_like(whole)

    def job():
        for k, pck in enumerate(pcs):
            pck.encode(whole[k * shard:(k + 1) * shard])
        for k, pck in enumerate(pcs):
            pck.decode(out=back[k * shard:(k + 1) * shard], check=False)

    ms_full = gpu_ms(job, 3)
    assert torch.equal(back.view(torch.int16), whole.view(torch.int16)) and all(pck.status()[0] == 0 for pck in pcs)
    out["configs[3]"]["whole_job_on_one_device"] = {
        "frames": nfull, "raw_bytes": int(whole.numel() * 2), "encoded_bytes": int(sum(pck.finish().nbytes() for pck in pcs)), "ms": ms_full,
        "fps": nfull / ms_full * 1e3, "raw_GBs": nfull * 4.0 * h3 * w3 / ms_full / 1e6,
        "note": "10 000 frames 1024x768 resident in one MI355X's HBM: encode of all eight shards, then decode of all eight (16 launches), bit-exact"}
    del whole, back, pcs
    torch.cuda.empty_cache()

# ------------------------------------------------------------------ configs[4]: float32 stream, motion correction + bounded loss
n4 = 60 if args.quick else 300
f32, shifts = s3_registration(n4, h, w)
u16 = np.clip(f32, 0, 65535).astype(np.uint16)
t4 = torch.from_numpy(u16).to(dev)
sh = torch.from_numpy(shifts.astype(np.float32)).to(dev)
ms = gpu_ms(lambda: D.remove_motion(t4, sh, rows=h - 3), 5)
c4 = {"workload": "%d x %dx%d float32 S3 (known shifts): motion correction, then bounded-loss recording (low=high=3, stdFactor 0)" % (n4, w, h),
      "motion_correction_device_resident_fps": n4 / ms * 1e3}
# registration itself: MaskedRegistratorECC (gaussian + normalisation on the host side, ECC iterations on the device)
from librir_amd.registration import MaskedRegistratorECC  # noqa: E402

nreg = 30 if args.quick else 100
regr = MaskedRegistratorECC(1, 1)
regr.start(f32[0])
t0 = time.perf_counter()
for i in range(1, nreg):
    regr.compute(f32[i])
c4["registration_ecc_per_frame_fps"] = (nreg - 1) / (time.perf_counter() - t0)
from librir_amd.registration import DeviceRegistratorECC  # noqa: E402

tf32 = torch.from_numpy(f32[:nreg]).to(dev)
dreg = DeviceRegistratorECC(1, 1)
dreg.start(tf32[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
dreg.compute_many(tf32[1:])  # (chunks of frames: pre-processing in shared launches, the chunk's alignments in one launch)
c4["registration_ecc_device_resident_fps"] = (nreg - 1) / (time.perf_counter() - t0)
dreg1 = DeviceRegistratorECC(1, 1)
dreg1.start(tf32[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, nreg):
    dreg1.compute(tf32[i])
c4["registration_ecc_device_resident_frame_by_frame_fps"] = (nreg - 1) / (time.perf_counter() - t0)
assert dreg1.x == dreg.x and dreg1.y == dreg.y
# eight independent sequences side by side (rir_ecc_align_multi_device): each one's track is bit-identical to its solo run
S8 = 8
seqs8 = [tf32] + [torch.from_numpy(s3_registration(nreg, h, w, seed=99 + q)[0]).to(dev) for q in range(1, S8)]
best8 = 0.0
for _ in range(3):
    rs = [DeviceRegistratorECC(1, 1) for _ in range(S8)]
    for q in range(S8):
        rs[q].start(seqs8[q][0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    DeviceRegistratorECC.compute_many_multi(rs, [s_[1:] for s_ in seqs8])
    best8 = max(best8, S8 * (nreg - 1) / (time.perf_counter() - t0))
assert rs[0].x == dreg.x and rs[0].y == dreg.y
c4["registration_ecc_8_sequences_device_resident_fps"] = best8
del seqs8, rs
c4["registration_max_error_px"] = float(max(np.abs(np.array(regr.x) - shifts[:nreg, 0]).max(), np.abs(np.array(regr.y) - shifts[:nreg, 1]).max()))


def _norm(a):
    return (a - a.min()) / (a.max() - a.min())


_ref0 = _norm(O.gaussian_filter(f32[0], 0.5))
_im1 = _norm(O.gaussian_filter(f32[1], 0.5))
c4["cpu_port_ecc_fps_1thread"] = cpu_fps(lambda: O.ecc_translation(_ref0, _im1, (0.0, 0.0)), 1, budget=4.0)
reg = D.remove_motion(t4, sh, rows=h - 3).cpu().numpy()
# bounded loss on the device-resident, motion-corrected stream (rir_lossy_step_device), then the lossless encode of it
treg = torch.from_numpy(reg).to(dev)
ctx4 = D.CodecContext(w, h, n4, gop, device=dev)


pc4 = D.PackedCodec(w, h, n4, gop, device=dev)


def lossy_dev():
    ls = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
    out4, _, _ = ls.step(treg, errors=False)
    pc4.encode(out4)
    ls.close()


ms = gpu_ms(lossy_dev, 3)
c4["lossy_then_encode_device_resident_fps"] = n4 / ms * 1e3


# the step alone, as bench.py's other_paths times it: the stream exists already (created and seeded outside the clock), calls of 200 frames
# (on bench.py's data, the S1 recipe, so that the two tools agree: best of 3, as there)
m1 = 200
s1_200 = torch.from_numpy(s1_noisy_background(m1, h, w)).to(dev)


def best_rate(fn, count, reps=3):
    b = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        b = max(b, count / (time.perf_counter() - t0_))
    return b


ls1 = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
ls1.step(s1_200[:60], errors=False)
c4["lossy_step_device_resident_fps_one_stream"] = best_rate(lambda: ls1.step(s1_200, errors=False), m1)
ls1.status()
c4["lossy_step_groups_offered_to_and_taken_by_the_constant_budget_form"] = list(ls1.path_stats())
ls1.close()
if not args.quick:  # the same in calls of 1 000 frames (the call's fixed cost - ~130 us of tables, small launches and their gaps - spread over more frames)
    s1_1000 = torch.from_numpy(s1_noisy_background(1000, h, w)).to(dev)
    ls1 = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
    ls1.step(s1_1000[:60], errors=False)
    c4["lossy_step_device_resident_fps_one_stream_1000_frame_calls"] = best_rate(lambda: ls1.step(s1_1000, errors=False), 1000)
    ls1.status()
    # ... and the recording chain in its steady state - the stream stays open, as a recording keeps it (the key above creates and closes a
    # stream inside the clock): loss step, then the packed encoder on what it leaves
    pc1000 = D.PackedCodec(w, h, 1000, gop, device=dev)

    def chain_1000():
        o_, _, _ = ls1.step(s1_1000, errors=False)
        pc1000.encode(o_)

    chain_1000()
    c4["lossy_then_encode_device_resident_fps_open_stream_1000_frame_calls"] = best_rate(chain_1000, 1000)
    del pc1000
    ls1.close()
    os.environ["RIR_LOSSY_NO_CONST"] = os.environ["RIR_LOSSY_NO_SPEC"] = "1"  # the general (resident) form on the same call, for comparison
    ls1 = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
    ls1.step(s1_1000[:60], errors=False)
    c4["lossy_step_device_resident_fps_one_stream_1000_frame_calls_general_form"] = best_rate(lambda: ls1.step(s1_1000, errors=False), 1000)
    ls1.status()
    ls1.close()
    del os.environ["RIR_LOSSY_NO_CONST"], os.environ["RIR_LOSSY_NO_SPEC"]
    # the reference's DEFAULT parameters (6 / 2 / stdFactor 5 / 32): budgets that follow the statistics - the speculative form (round 6) on a
    # scene that does not move (committed) and on S1 (budgets move every frame: the general form steps it), with the form's own books
    g_ = torch.Generator(device=dev).manual_seed(5)
    static = ((torch.rand((h, w), generator=g_, device=dev) * 1000 + 10)[None] + 0.7 * torch.randn((1000, h, w), generator=g_, device=dev)).to(torch.int32).to(torch.uint16)
    for name_, scene_ in (("static_scene", static), ("S1", s1_1000)):
        ls1 = D.LossyStream(w, h, h - 3, 6, 2, 5.0, 32)
        ls1.step(scene_, errors=False)
        c4["lossy_step_default_parameters_%s_fps_one_stream_1000_frame_calls" % name_] = best_rate(lambda: ls1.step(scene_, errors=False), 1000)
        ls1.status()
        c4["lossy_step_default_parameters_%s_speculative_groups_through_offered_committed_passes" % name_] = list(ls1.spec_stats())
        ls1.close()
    del static
    del s1_1000
# independent streams in shared launches (rir_lossy_step_multi_device): the loss state is sequential in time, streams run side by side
for S in (7, 9, 32):  # 7 streams of this size share one resident launch of the first form of the run kernel, 9 one of the second; more go a batch after the other
    m = 20 if args.quick else m1
    streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
    ins = [s1_200[:m].clone() for _ in range(S)]
    D.LossyStream.step_many(streams, ins, errors=False)  # first frames + ring fill started
    c4["lossy_step_device_resident_fps_%d_streams" % S] = best_rate(lambda: D.LossyStream.step_many(streams, ins, errors=False), m * S)
    streams[0].status()
    for st_ in streams:
        st_.close()
    del ins
# the same seven streams on THIS configuration's data, the motion-corrected registration stream: a flat scene, ~20 distinct levels per frame -
# the histogram pass of the step depends on how the levels are spread (lossy_hist_add8, lossy_kernels.hip)
streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(7)]
ins = [treg[:m1].clone() for _ in range(7)]
D.LossyStream.step_many(streams, ins, errors=False)
c4["lossy_step_device_resident_fps_7_streams_on_the_motion_corrected_stream"] = best_rate(lambda: D.LossyStream.step_many(streams, ins, errors=False), ins[0].shape[0] * 7)
streams[0].status()
for st_ in streams:
    st_.close()
del ins
with tempfile.TemporaryDirectory() as d:
    dst = os.path.join(d, "lossy.h264")
    with IRSaver(os.path.join(d, "warm4.h264"), w, h, h - 3) as s:  # (the first bounded-loss recording of a process pays one-off set-up costs: 0.2 s)
        s.set_parameter("stdFactor", 0)
        for i in range(60):
            s.add_image_lossy(reg[i], i * 1000)
    t0 = time.perf_counter()
    with IRSaver(dst, w, h, h - 3) as s:
        s.set_parameter("lowValueError", 3)
        s.set_parameter("highValueError", 3)
        s.set_parameter("stdFactor", 0)
        for i in range(n4):
            s.add_image_lossy(reg[i], i * 1000)
    c4["lossy_record_per_frame_abi_fps"] = n4 / (time.perf_counter() - t0)
    c4["lossy_file_ratio"] = reg.nbytes / os.path.getsize(dst)
    dst2 = os.path.join(d, "lossless.h264")
    with IRSaver(dst2, w, h, h) as s:
        for i in range(n4):
            s.add_image(reg[i], i * 1000)
    c4["lossless_file_ratio_same_frames"] = reg.nbytes / os.path.getsize(dst2)
L = OracleLossy(O, w, h, h - 3, low_err=3, high_err=3, std_factor=0.0, running_average=32)
c4["cpu_port_lossy_step_fps_1thread"] = cpu_fps(lambda: [L.step(reg[i]) for i in range(10)], 10, budget=4.0)
out["configs[4]"] = c4
print(json.dumps(out, indent=1))
