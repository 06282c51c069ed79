#!/usr/bin/env python3
"""Headline benchmark: lossless encode + decode of 640x512 uint16 IR frames on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one device-resident batch: lossless encode (rir_codec_encode_packed_device: ONE
kernel that leaves the PACKED form of the encoded batch - record headers, a (position, length) pair per (chunk, tile) segment
and the payload without holes in a buffer of HALF the raw size; `encoded_footprint_bytes` says what it occupies) then decode
(rir_codec_decode_packed_device) on BASELINE.json configs[1] (1 000-frame 640x512 uint16 stream, recipe S1) -
per rank: two launches that move 4WH + 2C bytes (+ 6 % of the frames re-read where the waves of a segment meet in time).
Beside it, as extra keys: `slotted_form` (round 3's step: the encoder's worst-case slots decoded in place - faster, but the
encoded batch is larger than its input) and `dense_file_form` (the canonical ordered stream of the file format: one more
pass).  Ranks hold independent shards (weak scaling, no collective in that data path): `value` = frames all ranks processed /
the slowest rank's time.

N > 1 additionally times, in the same run and with the same bracket, the step WITH the exchange north_star names
(every GPU ends up holding the whole decoded stream), in both forms of librir_amd/distributed.py:
  value_with_exchange             encode, decode in sub-batches, all-gather of the DECODED frames per sub-batch on the
                                  communicator's stream while the next sub-batch decodes
  value_with_compressed_exchange  encode, all-gather of the COMPRESSED chunks in pieces, every rank decodes every
                                  rank's chunks on arrival into the reassembled stream
each with the bytes a rank receives, its GB/s and the fraction of the per-GPU xGMI budget.

One JSON line is printed by rank 0: whole-job frames/s, plus
  roofline      - the dominant kernel's algorithmic bytes / its HIP-event duration vs 8 TB/s
  cpu_baseline  - the oracle (plain-C port) timed on a bounded sample of the same workload: 1 core, and all cores of
                  this process's CPU share over independent chunks; the compiled reference's filters when oracle/_ref
                  travelled with the snapshot
  per_frame_abi_fps / batched_h2d_d2h_fps - SURVEY §8d's other two numbers for the same configuration (N = 1)
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
XGMI_IN_GBS = 7 * 76.8  # 7 links x 153.6 GB/s bidirectional = 76.8 GB/s inbound each: what one GPU can receive, peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)  # (0.27 ms each: a timed region of ~55 ms; 20 steps were 6 ms - shorter than a utilisation sampler's period)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--frames", type=int, default=1000)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--gop", type=int, default=50)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-abi", action="store_true", help="skip the per-frame / batched host-pointer numbers")
    p.add_argument("--profile", action="store_true", help="for rocprofv3 --pmc passes: only the step itself (no ramp, no 2 s of repeats, no other forms)")
    p.add_argument("--abi-child", action="store_true", help=argparse.SUPPRESS)  # internal: the process that measures those numbers
    p.add_argument("--cpu-frames", type=int, default=1000, help="frames of the same stream the CPU oracle works through per pass")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="the CPU oracle repeats passes until this much time is spent")
    p.add_argument("--exchange-piece", type=int, default=100, help="frames per sub-batch of the decoded all-gather (whole chunks)")
    p.add_argument("--exchange-chunks", type=int, default=4, help="chunks per piece of the compressed all-gather")
    return p.parse_args()


# ---- CPU baseline ------------------------------------------------------------------------------------------
_CPU = {}


def _cpu_worker(job):
    """One process of the all-cores leg: whole passes over its own run of chunks until the deadline."""
    first_chunk, nchunks, seconds = job
    from oracle.pyoracle import Oracle

    O = Oracle()
    fr, gop = _CPU["frames"], _CPU["gop"]
    h, w = fr.shape[1:]
    t0 = time.perf_counter()
    done = 0
    while True:
        for c in range(first_chunk, first_chunk + nchunks):
            hdr, off, st = O.codec_encode_chunk(fr[c * gop:(c + 1) * gop])
            O.codec_decode_chunk(hdr, off, st, w, h)
            done += gop
        if time.perf_counter() - t0 >= seconds:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(frames_np, gop, nframes, seconds):
    """Oracle (CPU port of the same format) encode+decode: whole passes over the first `nframes` frames of the same
    stream until `seconds` of CPU work are spent (bounded sample) - on one core, then on every core of this process's
    CPU share (independent chunks, one process per core).  Runs BEFORE the GPU is initialised (the worker processes
    are forked)."""
    import multiprocessing as mp

    from oracle.pyoracle import Oracle, Ref

    O = Oracle()
    n = min(nframes, frames_np.shape[0])
    n -= n % gop if n >= gop else 0
    h, w = frames_np.shape[1:]
    t0 = time.perf_counter()
    done = 0
    passes = 0
    while True:
        for c in range(0, n, gop):
            hdr, off, st = O.codec_encode_chunk(frames_np[c:c + gop])
            dec = O.codec_decode_chunk(hdr, off, st, w, h)
        done += n
        passes += 1
        if time.perf_counter() - t0 >= seconds or passes >= 64:
            break
    dt = time.perf_counter() - t0
    assert np.array_equal(dec, frames_np[n - gop:n] if n >= gop else frames_np[:n])
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {
        "value": done / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d pass(es) over the first %d frames of the same S1 stream (%d frames), oracle/rir_oracle.c encode+decode, "
                  "1 thread, %.1f s" % (passes, n, done, dt),
        "note": "the reference's own lossless codec (libx264 behind ffmpeg) cannot be built without network access: this is the "
                "CPU port of THIS build's format, and zfile_zstd below the one reference codec path whose arithmetic is reachable",
        "host_cores_available": os.cpu_count(),
        "host_cores_this_process_may_use": share,
    }
    # all cores: one process per core over independent chunks (the reference's loops are single-threaded - its OpenMP
    # pragmas are inert, signal_processing.cpp:111, Filters.h:252 - so this is what a user could get by hand)
    nchunks = n // gop
    procs = max(1, min(share, 32, nchunks))
    if procs > 1 and nchunks >= procs:
        _CPU["frames"], _CPU["gop"] = frames_np, gop
        per = nchunks // procs
        jobs = [(i * per, per, max(2.0, seconds * 0.6)) for i in range(procs)]
        tw = time.perf_counter()
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - tw
        out["threads_all"] = {"value": sum(r[0] / r[1] for r in res), "unit": "frames/s", "cores": procs,
                              "sample": "%d processes, %d chunk(s) each, %.1f s of work each (%.1f s wall incl. start-up)" % (procs, per, jobs[0][2], wall)}
    # the reachable reference codec arithmetic: ZFile method 1 = one-shot zstd per raw frame
    # (reference src/cpp/video_io/ZFile.cpp:483-542), through the host's libzstd if present
    try:
        import ctypes as ct

        z = ct.CDLL("libzstd.so.1")
        z.ZSTD_compressBound.restype = ct.c_size_t
        z.ZSTD_compressBound.argtypes = [ct.c_size_t]
        z.ZSTD_compress.restype = ct.c_size_t
        z.ZSTD_compress.argtypes = [ct.c_void_p, ct.c_size_t, ct.c_void_p, ct.c_size_t, ct.c_int]
        z.ZSTD_decompress.restype = ct.c_size_t
        z.ZSTD_decompress.argtypes = [ct.c_void_p, ct.c_size_t, ct.c_void_p, ct.c_size_t]
        z.ZSTD_versionNumber.restype = ct.c_uint
        fb = frames_np[0].nbytes
        cap = z.ZSTD_compressBound(fb)
        buf = np.empty(cap, np.uint8)
        back = np.empty_like(frames_np[0])
        m = min(n, 100)
        t0 = time.perf_counter()
        csum = 0
        for i in range(m):
            c = z.ZSTD_compress(buf.ctypes.data, cap, frames_np[i].ctypes.data, fb, 0)
            z.ZSTD_decompress(back.ctypes.data, fb, buf.ctypes.data, c)
            csum += c
        dtz = time.perf_counter() - t0
        out["zfile_zstd"] = {"value": m / dtz, "unit": "frames/s", "cores": 1, "ratio": m * fb / csum,
                             "libzstd": int(z.ZSTD_versionNumber()), "sample": "%d frames, level 0 one-shot per frame" % m}
    except Exception as e:  # libzstd absent: say so, do not fail the bench
        out["zfile_zstd"] = {"value": None, "note": "libzstd.so.1 not loadable: %s" % e}
    # the compiled reference's filters (unmodified reference C++, oracle/_ref), when the .so travelled with the snapshot
    if Ref.available():
        R = Ref()
        img = frames_np[0]
        f32 = img.astype(np.float32)

        def fps(fn, budget=1.5):
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < budget:
                fn()
                k += 1
            return k / (time.perf_counter() - t0)

        out["reference_filters"] = {
            "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "one %dx%d frame repeated for 1.5 s each" % (w, h),
            "translate_u16_nearest": fps(lambda: R.translate(img, 1.25, -2.5, "nearest")),
            "gaussian_filter_sigma_0.75": fps(lambda: R.gaussian_filter(f32, 0.75)),
        }
    else:
        out["reference_filters"] = None
    return out


def abi_numbers(frames_np, D, ctx, frames, out, n, h, w):
    """SURVEY §8d (ii) and (iii) for the same configuration: batched host -> device -> host (pinned, PCIe inclusive) and the
    per-frame C ABI as the wrapper drives it (IRSaver.add_image + IRMovie[i], one frame per call)."""
    import tempfile

    import torch

    from librir_amd.video_io import IRMovie, IRSaver

    # (a full collection of Python's garbage collector walks every object of the process - a million with torch imported: 35 ms, as much as
    # recording 2 000 frames - and one of the recordings below would pay it (measured in round 5: a single IRMovie[i] of 34.9 ms);
    # what exists now is moved out of the collector's sight, the loops' own garbage is still collected)
    import gc

    gc.collect()
    gc.freeze()
    res = {}
    with tempfile.TemporaryDirectory() as d:
        dst = os.path.join(d, "abi.h264")
        with IRSaver(os.path.join(d, "warm.h264"), w, h, h) as s:  # the first saver / loader of a process pay one-off set-up costs
            for i in range(min(n, 60)):                                # (page-locked staging, the writer and read-ahead threads)
                s.add_image(frames_np[i], i)
        with IRMovie.from_filename(os.path.join(d, "warm.h264")) as mov:
            for i in range(min(n, 60)):
                mov[i]
        runs = []
        for rep in range(4):  # four recordings of the n frames, each read back: the first one (cold: page cache, pools) is reported on its own, the median of the other three as the rate
            dst = os.path.join(d, "abi%d.h264" % rep)
            t0 = time.perf_counter()
            with IRSaver(dst, w, h, h) as s:
                for i in range(n):
                    s.add_image(frames_np[i], i * 1000)
            te = time.perf_counter() - t0
            t0 = time.perf_counter()
            with IRMovie.from_filename(dst) as mov:
                for i in range(n):
                    img = mov[i]
            td = time.perf_counter() - t0
            assert np.array_equal(img, frames_np[n - 1])
            os.remove(dst)
            runs.append((n / (te + td), n / te, n / td))
        cold, warm = runs[0], sorted(runs[1:])
        res["per_frame_abi_fps"] = warm[1][0]
        res["per_frame_abi_detail"] = {"frames": n, "record_fps": warm[1][1], "read_fps": warm[1][2], "round_trip_fps_warm_runs": [r[0] for r in warm],
                                       "round_trip_fps_first_recording_of_the_process": cold[0],
                                       "path": "IRSaver.add_image + IRMovie[i], file on the box's tmp filesystem; median of the 3 recordings after the first"}
    pin_in = torch.from_numpy(frames_np).pin_memory()
    pin_out = torch.empty_like(pin_in)
    enc0 = ctx.encode(frames)
    pin_stream = torch.empty((int(enc0.total_words()) + 1024,), dtype=torch.int64).pin_memory()

    def batched():
        frames.copy_(pin_in, non_blocking=True)
        enc = ctx.encode(frames)
        nw = int(enc.total_words())  # sync: the host needs the size to fetch the stream
        pin_stream[:nw].copy_(enc.stream[:nw], non_blocking=True)
        enc.stream[:nw].copy_(pin_stream[:nw], non_blocking=True)  # the stream comes back from the host
        ctx.decode(enc, out=out, check=False)
        pin_out.copy_(out, non_blocking=True)

    batched()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        batched()
    torch.cuda.synchronize()
    res["batched_h2d_d2h_fps"] = 3 * n / (time.perf_counter() - t0)
    assert np.array_equal(pin_out.numpy(), frames_np)
    # the host link itself, for the per-frame numbers above: one chunk of 50 frames each way, page-locked memory, HIP events
    k = min(n, 50)

    def link_gbs(dst, src):
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(src, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            best = max(best, src.numel() * src.element_size() / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        return best

    up, down = link_gbs(frames[:k], pin_in[:k]), link_gbs(pin_out[:k], frames[:k])
    fb = float(h * w * 2)
    us_up, us_down = fb / up / 1e3, fb / down / 1e3
    d = res["per_frame_abi_detail"]
    res["host_link"] = {
        "h2d_GBs": up, "d2h_GBs": down, "us_per_image_up": us_up, "us_per_image_down": us_down,
        "round_trip_bound_fps": 1e6 / (us_up + us_down),
        "per_frame_abi_frac_of_link_bound": res["per_frame_abi_fps"] / (1e6 / (us_up + us_down)),
        "record_frac_of_link_bound": d["record_fps"] / (1e6 / us_up), "read_frac_of_link_bound": d["read_fps"] / (1e6 / us_down),
        "note": "the per-frame C ABI moves every image over this link once in each direction (record: images up, a fifth of their size back as payload; "
                "read: payload up, images down); the bound counts the images alone"}
    try:
        res["other_paths"] = other_paths(D, frames, h, w)
    except Exception as e:  # (never at the cost of the numbers above)
        res["other_paths"] = {"error": repr(e)[:300]}
    return res


def other_paths(D, frames, h, w):
    """The two other device-resident paths of SURVEY §8 that have a rate of their own (BASELINE configs[4]), on 640x512 frames in
    HBM: the bounded-loss step (one stream, and seven streams - what one resident launch holds) and the ECC registration of a
    tracked sequence.  Reported beside the headline, not part of it."""
    import torch

    from librir_amd.registration import DeviceRegistratorECC
    from librir_amd.synthetic import s3_registration

    res = {}
    m = min(200, frames.shape[0])
    fr = frames[:m]

    def rate(fn, count, reps=3):
        best = 0.0
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = max(best, count / (time.perf_counter() - t0))
        return best

    st = D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32)
    st.step(fr[:60], errors=False)
    res["bounded_loss_step_fps_one_stream"] = rate(lambda: st.step(fr, errors=False), m)
    if frames.shape[0] > m:  # the whole batch in one call: a call's fixed cost (~0.1 ms: first and last frames of the group, finish kernel, the gap to the next call) over more frames
        res["bounded_loss_step_fps_one_stream_%d_frame_calls" % frames.shape[0]] = rate(lambda: st.step(frames, errors=False), frames.shape[0])
    st.status()
    res["bounded_loss_groups_offered_to_and_taken_by_the_constant_budget_form"] = list(st.path_stats())
    st.close()
    # the reference's DEFAULT parameters (6 / 2 / stdFactor 5 / 32, h264.cpp:1662-1665): budgets that follow the statistics.  On a scene that does
    # not move the speculative form's guess verifies and the streaming kernel steps the group; on S1 (one level up per frame) the budgets move
    # nearly every frame and the general (resident) form steps it - both bit-exact against the oracle (tests/test_gpu_lossy_spec.py)
    if frames.shape[0] > m:
        g = torch.Generator(device=frames.device).manual_seed(5)
        static = (torch.rand((h, w), generator=g, device=frames.device) * 1000 + 10)[None] + 0.7 * torch.randn((frames.shape[0], h, w), generator=g, device=frames.device)
        static = static.to(torch.int32).to(torch.uint16)
        for name, scene in (("static_scene", static), ("S1", frames)):
            sd = D.LossyStream(w, h, h - 3, 6, 2, 5.0, 32)
            sd.step(scene, errors=False)
            res["bounded_loss_default_parameters_%s_fps_one_stream_%d_frame_calls" % (name, frames.shape[0])] = rate(lambda: sd.step(scene, errors=False), frames.shape[0])
            sd.status()
            res["bounded_loss_default_parameters_%s_speculative_groups_through_offered_committed_passes" % name] = list(sd.spec_stats())
            sd.close()
        del static
    for S in (7, 9):  # (7: what one launch of the run kernel's first form holds; 9: its second form - state parked in LDS, 6 waves per SIMD)
        streams = [D.LossyStream(w, h, h - 3, 3, 3, 0.0, 32) for _ in range(S)]
        ins = [fr.clone() for _ in range(S)]
        D.LossyStream.step_many(streams, ins, errors=False)
        res["bounded_loss_step_fps_%d_streams" % S] = rate(lambda: D.LossyStream.step_many(streams, ins, errors=False), m * S)
        streams[0].status()
        for x in streams:
            x.close()
        del streams, ins
    nreg = 100
    f32, _ = s3_registration(nreg, h, w)
    tf = torch.from_numpy(f32).to(frames.device)

    def track():
        reg = DeviceRegistratorECC(1, 1, shape=(h, w))
        reg.start(tf[0])
        reg.compute_many(tf[1:])

    res["ecc_tracked_sequence_fps"] = rate(track, nreg - 1)
    # eight independent sequences side by side (rir_ecc_align_multi_device): each one's track is bit-identical to its solo run
    S8 = 8
    seqs = [tf] + [torch.from_numpy(s3_registration(nreg, h, w, seed=99 + q)[0]).to(frames.device) for q in range(1, S8)]

    def track8():
        regs = [DeviceRegistratorECC(1, 1, shape=(h, w)) for _ in range(S8)]
        for q in range(S8):
            regs[q].start(seqs[q][0])
        DeviceRegistratorECC.compute_many_multi(regs, [s_[1:] for s_ in seqs])

    res["ecc_tracked_8_sequences_fps"] = rate(track8, S8 * (nreg - 1))
    res["note"] = "best of 3; %d-frame calls of the bounded-loss step (low = high = 3, stdFactor 0, 509 lossy rows), %d float32 S3 frames for the registration" % (m, nreg)
    return res


def check_world(ranks_seen, gpus, bus_ids, share_gpu):
    """The N > 1 line must be able to prove what it ran on: the communicator saw every rank, and no two ranks sat on one GPU.
    -> None, or the reason the run is refused (no `value` is printed then, exit status 3)."""
    if int(ranks_seen) != int(gpus):
        return "the communicator's all-reduce saw %d rank(s), --gpus says %d" % (int(ranks_seen), int(gpus))
    if not share_gpu and len(set(bus_ids)) != len(bus_ids):
        return "two ranks report the same device (%s) - set RIR_BENCH_SHARE_GPU=1 for a rehearsal on one GPU" % ", ".join(sorted(bus_ids))
    return None


def world_identity(dist, torch, dev, world, rank, backend):
    """ranks_seen (an all-reduce of ones on the communicator `value` is measured with), the collective library's version, and every rank's
    device name / PCI bus id gathered to all ranks."""
    one = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    if dev is not None and getattr(dev, "type", "cpu") == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xff, getattr(pr, "pci_device_id", 0)) if hasattr(pr, "pci_bus_id") \
            else str(getattr(pr, "uuid", "gpu%d" % dev.index))
        # what tells two devices apart even where a driver reports no (or the same) bus id for all of them: + uuid, + the index among the devices
        # this rank sees, + what it was allowed to see (ranks of one launcher see the same set and take different indices)
        visible = "|".join(os.environ.get(v, "") for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
        mine = {"rank": rank, "device_name": pr.name, "pci_bus_id": bus, "device_index": dev.index,
                "device_key": "%s %s #%d [%s]" % (bus, getattr(pr, "uuid", ""), dev.index, visible)}
    else:
        mine = {"rank": rank, "device_name": "cpu", "pci_bus_id": "cpu", "device_index": -1, "device_key": "cpu"}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:
            version = "unknown (%s)" % repr(e)[:80]
    return int(one.item()), version, everyone


def newest_n1_value():
    """`value` of the newest BENCH_r*.json the driver left at the repo root (the N = 1 line of an earlier round), or None"""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")), reverse=True):
        try:
            d = json.load(open(f))
            p_ = d.get("parsed") or {}
            if p_.get("n_gpus") == 1 and p_.get("value"):
                return {"file": os.path.basename(f), "value": float(p_["value"])}
        except Exception:
            continue
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: start the N ranks as CHILD processes (a spawn, never an exec; nothing has touched the GPU
        # yet), relay what they print - rank 0's JSON line - and leave with their status
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        sys.exit(2)

    from librir_amd.synthetic import s1_noisy_background

    n, h, w, gop = args.frames, args.height, args.width, args.gop
    # every rank holds its own shard of the stream (different seed = different frames)
    frames_np = s1_noisy_background(n, h, w, seed=1234 + rank)
    if args.abi_child:
        import torch

        from librir_amd import device as D

        torch.cuda.set_device(0)
        frames = torch.from_numpy(frames_np).cuda()
        ctx = D.CodecContext(w, h, n, gop)
        print(json.dumps(abi_numbers(frames_np, D, ctx, frames, torch.empty_like(frames), n, h, w)))
        return
    # the CPU path is timed beside the N=1 run only, before this process touches the GPU (forked workers)
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(frames_np, gop, args.cpu_frames, args.cpu_seconds)
    # The per-frame / batched host-pointer numbers come from a process of their own, run to completion before this one touches
    # the GPU: the per-frame path encodes and decodes chunk by chunk (50-frame launches of the same kernels), which would mix
    # into the per-kernel averages of a rocprofv3 --stats run of this command (profiles/: one stats file per process).
    extra_abi = {}
    if world == 1 and not args.no_abi:
        import subprocess

        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--abi-child", "--frames", str(n), "--width", str(w), "--height", str(h),
                            "--gop", str(gop)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            extra_abi = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            extra_abi = {"per_frame_abi_fps": None, "abi_error": (r.stderr or r.stdout)[-300:]}

    import torch
    import torch.distributed as dist

    # Rehearsal switches (never set by the driver): RIR_BENCH_BACKEND=gloo runs the N>1 control flow without RCCL,
    # RIR_BENCH_SHARE_GPU=1 lets every rank use GPU 0 of a one-GPU box.
    backend = os.environ.get("RIR_BENCH_BACKEND", "nccl")
    if os.environ.get("RIR_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RIR_BENCH_RCCL_SOLO=1 (rehearsal, with --gpus 1): the N > 1 control flow - identity, barriers, the exchange - on a communicator of ONE rank, so
    # that a one-GPU box runs every collective of this file through RCCL itself (device tensors, the uint8 views, the object gather)
    solo = world == 1 and bool(os.environ.get("RIR_BENCH_RCCL_SOLO"))
    dist_on = world > 1 or solo
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if solo and "MASTER_PORT" not in os.environ:
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        kw = dict(rank=0, world_size=1) if solo else {}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)

    identity = None
    if dist_on:
        # the line proves what it ran on, or there is no line: every rank on the communicator, every rank a GPU of its own
        ranks_seen, coll_version, everyone = world_identity(dist, torch, dev, world, rank, backend)
        why = check_world(ranks_seen, args.gpus, [e["device_key"] for e in everyone], bool(os.environ.get("RIR_BENCH_SHARE_GPU")))
        if why:
            if rank == 0:
                sys.stderr.write("bench.py: refusing to run: %s\n" % why)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)
        identity = {"ranks_seen": ranks_seen, "rccl_version" if backend == "nccl" else "collective_backend": coll_version if backend == "nccl" else backend,
                    "ranks": everyone, "gpu_shared_by_ranks (rehearsal)": True if os.environ.get("RIR_BENCH_SHARE_GPU") else None,
                    "single_rank_communicator (rehearsal)": True if solo else None}

    from librir_amd import device as D
    from librir_amd.distributed import CompressedGather, FrameGather, shard_plan

    frames = torch.from_numpy(frames_np).to(dev)
    ctx = D.CodecContext(w, h, n, gop, device=dev)  # (the slotted and the dense form, timed beside the step)
    pc = D.PackedCodec(w, h, n, gop, device=dev)    # the step: stream buffer = the 8 bit-per-pixel budget (half the raw bytes)
    out = torch.empty_like(frames)

    def step():
        pc.encode(frames)
        pc.decode(out=out, check=False)

    def step_slotted():
        ctx.encode_tiles(frames)
        ctx.decode_slots(out=out, check=False)

    def k_steps(k, fn=None):
        fn = fn or step
        barrier()
        t1 = time.perf_counter()
        for _ in range(k):
            fn()
        barrier()
        return time.perf_counter() - t1

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if not dist_on:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    K = args.steps
    # ---- set-up, untimed.  A device that has been idle runs its first milliseconds of work below its sustained clocks (the CPU
    # baseline and the host-pointer child above take tens of seconds): the same K steps measured straight after the W warm-up steps
    # are reported as `value_from_idle`, then one second of the step brings the device to the state a pipeline keeps it in ----
    for _ in range(args.warmup):
        step()
    dt_idle = max_over_ranks(k_steps(K))
    ramp_steps = 0 if args.profile else min(100000, max(16, int(1.0 / (dt_idle / K))))  # (from the max over ranks: the same count on every rank)
    for _ in range(ramp_steps):
        step()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize ----
    out.zero_()
    dt = max_over_ranks(k_steps(K))
    code, low_words, high_words, arena_words = pc.status()
    if code != 0:
        raise SystemExit("bench.py: the encoded batch did not fit its budget (code %d) - refusing to report a number" % code)

    # ---- the same K steps once more with HIP events between the launches (on the stream the kernels run on): the per-kernel
    # durations of the roofline.  An event between two launches keeps the second kernel from starting under the tail of the
    # first one; that costs this region some 5 % (`ms_per_step_with_events`), which is why `value` is not taken from it ----
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    pc.error.zero_()  # (raised by any decode from here on that meets a malformed table or record; read once, after the region)
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        pc.reset()  # (the fill launch that zeroes the encoder's control block: outside the packing kernel's events)
        ev[k][0].record()
        pc.encode(frames, reset=False)
        ev[k][1].record()
        pc.decode(out=out, check=False)
        ev[k][2].record()
    barrier()
    dt_events = max_over_ranks(time.perf_counter() - t0)

    ms_tiles = sum(ev[k][0].elapsed_time(ev[k][1]) for k in range(K)) / K
    ms_decode = sum(ev[k][1].elapsed_time(ev[k][2]) for k in range(K)) / K

    # ---- parity gate before any number is reported: decoded stream == input, bit-exact ----
    ok = torch.equal(out.view(torch.int16), frames.view(torch.int16)) and int(pc.error.item()) == 0
    batch = pc.finish()
    payload_bytes = batch.payload_bytes()
    if not ok:
        raise SystemExit("bench.py: decode(encode(x)) != x - refusing to report a number")

    # ---- spread, and a GPU that is visibly busy: the same K-step region again and again until BUSY_S seconds of it have run (a utilisation
    # sampler with a period of 5 s sees nothing of a 5 ms region, and may miss 2 s); `value` stays the first region above ----
    BUSY_S = 6.5
    reps = []
    t_busy = time.perf_counter()
    while len(reps) < (1 if args.profile else 5) or (not args.profile and time.perf_counter() - t_busy < BUSY_S and len(reps) < 20000):
        reps.append(max_over_ranks(k_steps(K)) / K * 1e3)
        if dist_on and len(reps) >= 5 and not args.profile:  # (every rank must leave the loop in the same round: the region count is rank 0's decision)
            go = torch.tensor([1 if time.perf_counter() - t_busy < BUSY_S else 0], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.broadcast(go, 0)
            if int(go.item()) == 0:
                break
    nreg = len(reps)
    reps.sort()
    spread = {"ms_per_step_min": reps[0], "ms_per_step_median": reps[nreg // 2], "ms_per_step_max": reps[-1], "repeats": nreg,
              "note": "further K-step regions after the one `value` is computed from, repeated until the GPU had been busy for %.1f s" % BUSY_S}

    if args.profile:
        if rank == 0:
            print(json.dumps({"profile_run": True, "ms_per_step": dt / K * 1e3, "kernels_ms": {"rirb1_encode_packed": ms_tiles, "rirb1_decode_tiles": ms_decode},
                              "payload_bytes": payload_bytes, "encoded_footprint_bytes": batch.nbytes()}))
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    # ---- round 3's step beside it: the slotted form (encode_tiles + decode_slots; the encoded batch lives in worst-case slots) ----
    for _ in range(args.warmup):
        step_slotted()
    dt_slotted_unplaced = max_over_ranks(k_steps(K, step_slotted))
    placement_us = ctx.place_workspace(frames)
    for _ in range(args.warmup):
        step_slotted()
    dt_slotted = max_over_ranks(k_steps(K, step_slotted))
    if int(ctx.error.item()) != 0 or not torch.equal(out.view(torch.int16), frames.view(torch.int16)) or ctx.slots_payload_bytes() != payload_bytes:
        raise SystemExit("bench.py: the slotted form does not decode to the input - refusing to report a number")

    # ---- the dense (file) form of the same batch: one more pass over the payload, then the decode from the dense stream ----
    evd = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
    out.zero_()
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        evd[k][0].record()
        ctx.encode_tiles(frames)
        evd[k][1].record()
        enc = ctx.encode_compact()
        evd[k][2].record()
        ctx.decode(enc, out=out, check=False)
        evd[k][3].record()
    barrier()
    dt_dense = max_over_ranks(time.perf_counter() - t0)
    ms_compact_tiles = sum(evd[k][0].elapsed_time(evd[k][1]) for k in range(K)) / K
    ms_compact = sum(evd[k][1].elapsed_time(evd[k][2]) for k in range(K)) / K
    ms_decode_dense = sum(evd[k][2].elapsed_time(evd[k][3]) for k in range(K)) / K
    cbytes = enc.compressed_bytes()
    if int(ctx.error.item()) != 0 or not torch.equal(out.view(torch.int16), frames.view(torch.int16)) or enc.total_words() * 8 != payload_bytes:
        raise SystemExit("bench.py: the dense form does not decode to the input - refusing to report a number")

    tables_bytes = batch.nbytes() - payload_bytes
    extra = {"encoded_footprint_bytes": batch.nbytes(), "raw_bytes": int(2 * h * w * n),
             "value_from_idle": n * K * world / dt_idle,
             "value_from_idle_note": "the same K steps after the same W warm-up steps on a device that had been idle (no ramp): %d steps of set-up ran between "
                                     "this region and the one `value` is computed from" % ramp_steps,
             "encoded_batch": {"payload_bytes": payload_bytes, "tables_bytes": tables_bytes, "extent_low_bytes": low_words * 8, "extent_high_bytes": high_words * 8,
                               "stream_buffer_bytes": pc.stream.numel() * 8, "encoder_workspace_bytes": pc.workspace.numel(), "arena_bytes_used": arena_words * 8,
                               "note": "what the step's encoder leaves: headers + a (position, length) pair per segment + the payload without holes at the two "
                                       "ends of a stream buffer of half the raw size (8 bit-per-pixel budget; a batch that needs more is refused, not "
                                       "written out of bounds) - everything a file or an exchange needs, nothing else"},
             "slotted_form": {"value": n * K * world / dt_slotted, "ms_per_step": dt_slotted / K * 1e3,
                              "value_unplaced": n * K * world / dt_slotted_unplaced,
                              "workspace_bytes": int(ctx.layout.workspace_bytes),
                              "packing_us_of_the_placement_candidates": [round(x, 1) for x in placement_us],
                              "note": "round 3's step (encode_tiles + decode_slots): the encoded batch lives in worst-case slots inside a workspace larger "
                                      "than the input - an on-device hand-over, not an encoded batch one can keep; workspace placed by measurement (DESIGN.md §7)"},
             "spread": spread,
             "dense_file_form": {"value": n * K * world / dt_dense, "ms_per_step": dt_dense / K * 1e3, "ms_compact": ms_compact,
                                 "ms_decode_from_dense": ms_decode_dense,
                                 "note": "encode_tiles + scan/compact (the canonical ordered stream of the FILE format) + decode from it: round 2's step"}}
    if world == 1:
        # the single-pass (look-back) encoder beside the two-pass one that `value` is measured with: same outputs, fewer bytes
        # through HBM, no faster
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            sp = ctx.encode(frames, single_pass=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(K):
            sp = ctx.encode(frames, single_pass=True)
        e1.record()
        torch.cuda.synchronize()
        ctx.decode(sp, out=out, check=False)
        ok_sp = ctx.encode_status() == 0 and bool(torch.equal(out.view(torch.int16), frames.view(torch.int16)))
        extra["single_pass_encoder"] = {"ms_per_launch": e0.elapsed_time(e1) / K, "bit_exact_roundtrip": ok_sp,
                                        "kernel": "rirb1_encode_dense (memset + 1 launch): the dense stream in one pass", "two_pass_ms": ms_compact_tiles + ms_compact}
    extra.update(extra_abi)

    emitted = threading.Lock()

    def emit(exchange):
        # (one line, once: by the main thread after the exchange, or by the watchdog below when the exchange hangs)
        if not emitted.acquire(blocking=False):
            return
        if rank == 0:
            raw = 2.0 * h * w * n  # bytes of raw uint16 per batch
            kernels = {
                "rirb1_encode_packed": {"ms": ms_tiles, "alg_bytes": raw + payload_bytes + tables_bytes},
                "rirb1_decode_tiles": {"ms": ms_decode, "alg_bytes": raw + payload_bytes + tables_bytes},
            }
            for kv in kernels.values():
                kv["GBs"] = kv["alg_bytes"] / (kv["ms"] * 1e-3) / 1e9
            dom = max(("rirb1_encode_packed", "rirb1_decode_tiles"), key=lambda k_: kernels[k_]["ms"])
            traffic = None
            import glob

            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))  # the latest round's PMC pass
            tpath = tfiles[-1] if tfiles else ""
            if tpath:
                try:
                    traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            fps = n * K * world / dt
            res = {
                "metric": "IR frames/sec encode+decode, 640x512 uint16",
                "value": fps,
                "unit": "frames/s",
                "n_gpus": world,
                "steps": K,
                "warmup": args.warmup,
                "ms_per_step": dt / K * 1e3,
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "u16",
                "data": "synthetic",
                "config": {"workload": "configs[1]: %d-frame %dx%d uint16 stream (S1 noisy background, seed 1234+rank), lossless RIRB1 "
                                       "encode+decode, device-resident, GOP %d, per GPU; two launches per step: the encoder leaves the PACKED "
                                       "form (headers + position and length per segment + the payload without holes: encoded_footprint_bytes, "
                                       "in a buffer of half the raw size), the decoder reads it as it is; the compressed payload (17 %% of a "
                                       "pass's bytes) is written and read back through the 256 MiB Infinity Cache, the raw frames (655 MB "
                                       "each way) stream from / to HBM" % (n, w, h, gop),
                           "frames_per_gpu": n, "width": w, "height": h, "gop": gop, "sharding": "independent shard per rank"},
                "bit_exact_roundtrip": True,
                "value_excludes_exchange": True if dist_on else None,
                "step_alg_bytes": 2.0 * raw + 2.0 * (payload_bytes + tables_bytes),
                "step_alg_frac_of_hbm_peak": (2.0 * raw + 2.0 * (payload_bytes + tables_bytes)) / (dt / K) / 1e9 / HBM_PEAK_GBS,
                "compression_ratio": raw / cbytes,
                "roundtrip_raw_GBs": fps * 4.0 * h * w / 1e9 / world,
                "roundtrip_raw_frac_of_hbm_peak": fps * 4.0 * h * w / 1e9 / world / HBM_PEAK_GBS,
                "roofline": {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": kernels[dom]["GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                             "traffic_source": ("%s (static: rocprofv3 --pmc passes of an earlier run of this command, not measured in this run)"
                                                % os.path.relpath(tpath, ROOT)) if traffic is not None else None,
                             "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "ms_per_launch": kernels[dom]["ms"],
                             "measured": "HIP events between the launches of a K-step region that follows the timed one (ms_per_step_with_events)"},
                "ms_per_step_with_events": dt_events / K * 1e3,
                "kernels": kernels,
            }
            res.update(extra)
            if exchange:
                res.update(exchange)
            if identity:
                res.update(identity)
                ref1 = newest_n1_value()
                res["n1_reference_value"] = ref1
                res["scaling_factor_vs_n1_reference"] = (fps / ref1["value"]) if ref1 else None
            if cpu is not None:
                res["cpu_baseline"] = cpu
            print(json.dumps(res))

    # The exchange has never run on RCCL before a driver runs it (one-GPU boxes here): a collective that hangs must not cost the line its
    # `value`.  Armed on every rank before the exchange: after EXCHANGE_LIMIT_S seconds rank 0 prints the line without the exchange keys and
    # every rank leaves (a collective that hangs cannot be called off; the numbers of `value` were complete before).
    EXCHANGE_LIMIT_S = float(os.environ.get("RIR_BENCH_EXCHANGE_LIMIT_S", "240"))

    def exchange_timed_out():
        emit({"exchange_error": "the exchange did not finish within %.0f s: the line carries the sharded path only" % EXCHANGE_LIMIT_S})
        sys.stdout.flush()
        os._exit(0)

    watchdog = None
    if dist_on:
        watchdog = threading.Timer(EXCHANGE_LIMIT_S, exchange_timed_out)
        watchdog.daemon = True
        watchdog.start()

    # ---- N > 1: the same step WITH the exchange, inside its own timed bracket (same K, same barriers) ----
    exchange = None
    if dist_on:
        try:
            if os.environ.get("RIR_BENCH_EXCHANGE_HANG"):  # (rehearsal of the watchdog: as if a collective never came back)
                time.sleep(1e6)
            nchunks = ctx.layout.nchunks
            piece = max(gop, (args.exchange_piece // gop) * gop)
            while n % piece:
                piece -= gop
            cpp = piece // gop  # chunks per sub-batch of the decoded gather
            # (first frame, count) of every local chunk: the sub-batch decode writes straight into `out`
            cf_local = torch.tensor([[c * gop, min(gop, n - c * gop)] for c in range(nchunks)], dtype=torch.int64, device=dev)
            fg = FrameGather(out, piece)

            def produce(j, f0, f1):
                c0 = f0 // gop
                D.decode_chunks(enc.hdr[c0:c0 + cpp], enc.tile_off[c0:c0 + cpp], enc.chunk_off[c0:c0 + cpp + 1], enc.stream, cf_local[c0:c0 + cpp], out,
                                gop, ctx.error)

            def step_raw():
                ctx.encode(frames)
                fg.run(produce)

            plan = shard_plan(n * world, gop, world)  # equal shards: rank r holds frames [r n, (r + 1) n) of the whole stream
            full = torch.empty((n * world, h, w), dtype=torch.uint16, device=dev)
            cg = CompressedGather(plan, gop, ctx.layout.ntiles, chunks_per_piece=args.exchange_chunks)

            def consume(p):
                D.decode_chunks(p.hdr, p.tile_off, p.chunk_off, p.stream, p.chunk_frames, full, gop, ctx.error)

            def step_compressed():
                e = ctx.encode(frames)
                cg.run(e.hdr, e.tile_off, e.chunk_off, e.stream, consume)

            Kx = min(K, 50)  # (an exchange step is 10-40 x a plain one: its own, bounded step count; `value` is timed over K above)

            def timed(fn):
                for _ in range(max(1, min(args.warmup, 2))):
                    fn()
                barrier()
                t1 = time.perf_counter()
                for _ in range(Kx):
                    fn()
                barrier()
                return max_over_ranks(time.perf_counter() - t1)

            ctx.error.zero_()
            dt_raw = timed(step_raw)
            own = fg.full[:, rank].reshape(n, h, w)
            ok_raw = bool(torch.equal(own.view(torch.int16), frames.view(torch.int16))) and int(ctx.error.item()) == 0
            dt_cmp = timed(step_compressed)
            ok_cmp = bool(torch.equal(full[rank * n:(rank + 1) * n].view(torch.int16), frames.view(torch.int16))) and int(ctx.error.item()) == 0
            # every rank's shard arrived intact everywhere: per-shard checksums of what each rank holds vs the owners' own sums
            sums = torch.stack([full[r * n:(r + 1) * n].view(torch.int16).to(torch.int64).sum() for r in range(world)])
            sums_raw = torch.stack([fg.full[:, r].reshape(-1).view(torch.int16).to(torch.int64).sum() for r in range(world)])
            mine = frames.view(torch.int16).to(torch.int64).sum().reshape(1)
            owners = torch.empty((world,), dtype=torch.int64, device=dev)
            if backend == "nccl":
                dist.all_gather_into_tensor(owners, mine)
            else:
                hs = torch.empty((world,), dtype=torch.int64)
                dist.all_gather_into_tensor(hs, mine.cpu())
                owners = hs.to(dev)
            ok_cmp = ok_cmp and bool(torch.equal(sums, owners))
            ok_raw = ok_raw and bool(torch.equal(sums_raw, owners))
            flags = torch.tensor([int(ok_raw), int(ok_cmp)], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
            if int(flags.min().item()) != 1:
                raise SystemExit("bench.py: the exchanged stream differs from the owners' frames - refusing to report a number")

            def link(bytes_received, seconds):
                gbs = bytes_received * Kx / seconds / 1e9
                return {"bytes_received_per_rank_per_step": bytes_received, "GBs_received_per_rank": gbs, "xgmi_inbound_peak_GBs": XGMI_IN_GBS,
                        "frac_of_xgmi_inbound_peak": gbs / XGMI_IN_GBS}

            exchange = {
                "backend": "rccl" if backend == "nccl" else backend,
                "value_with_exchange": n * Kx * world / dt_raw,
                "ms_per_step_with_exchange": dt_raw / Kx * 1e3,
                "decoded_allgather": dict(link(fg.bytes_received, dt_raw), sub_batch_frames=piece, sub_batches=len(fg.bounds),
                                          layout="piece-major [sub-batch][rank][frame]", every_shard_intact_on_every_rank=True),
                "value_with_compressed_exchange": n * Kx * world / dt_cmp,
                "ms_per_step_with_compressed_exchange": dt_cmp / Kx * 1e3,
                "compressed_allgather": dict(link(cg.bytes_received, dt_cmp), chunks_per_piece=cg.m, pieces=len(cg.pieces),
                                             layout="stream order [rank][frame], decoded on arrival", every_shard_intact_on_every_rank=True,
                                             frames_decoded_per_rank_per_step=n * world),
                "exchange_steps": Kx,
                "note": "`value` is the sharded path (no collective: each rank encodes+decodes its own chunks). With the whole decoded stream "
                        "reassembled on EVERY GPU each rank must receive (N-1)/N of it: the job's rate is bounded by "
                        "xGMI inbound bandwidth / ((N-1)/N x 655 360 B) for decoded frames, and by N decodes per rank for compressed chunks "
                        "(DESIGN.md §8)",
            }
        except SystemExit:
            raise  # (a stream that arrived damaged: no number at all)
        except Exception as e:  # the exchange is an EXTRA measurement: it must not cost the line its `value` (never run on RCCL before a driver does)
            exchange = {"exchange_error": repr(e)[:400]}

    if watchdog is not None:
        watchdog.cancel()
    emit(exchange)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
