#!/usr/bin/env python3
"""Headline benchmark: lossless encode + decode of 640x512 uint16 IR frames on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one device-resident batch: rir_codec_encode_device then
rir_codec_decode_device on BASELINE.json configs[1] (1 000-frame 640x512 uint16 stream, recipe S1,
SURVEY.md §8d) - per rank.  Ranks hold independent shards (weak scaling, no collective in the timed
data path); the RCCL all-gather that reassembles the decoded stream is measured once after the
timed region and reported beside it.

One JSON line is printed by rank 0: whole-job frames/s, plus
  roofline      - the dominant kernel's algorithmic bytes / its HIP-event duration vs 8 TB/s
  cpu_baseline  - the oracle (plain-C port, 1 core) timed on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--frames", type=int, default=1000)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--gop", type=int, default=50)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-frames", type=int, default=1000, help="frames of the same stream the CPU oracle works through per pass")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="the CPU oracle repeats passes until this much time is spent")
    return p.parse_args()


def cpu_baseline(frames_np, gop, nframes, seconds):
    """Oracle (CPU port of the same format) encode+decode, one core: whole passes over the first
    `nframes` frames of the same stream until `seconds` of CPU work are spent (bounded sample)."""
    from oracle.pyoracle import Oracle

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
    out = {
        "value": done / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d pass(es) over the first %d frames of the same S1 stream (%d frames), oracle/rir_oracle.c encode+decode, "
                  "1 thread, %.1f s" % (passes, n, done, dt),
        "host_cores_available": os.cpu_count(),
    }
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
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        sys.exit(2)
    # Rehearsal switches (never set by the driver): RIR_BENCH_BACKEND=gloo runs the N>1 control flow without RCCL,
    # RIR_BENCH_SHARE_GPU=1 lets every rank use GPU 0 of a one-GPU box.
    backend = os.environ.get("RIR_BENCH_BACKEND", "nccl")
    if os.environ.get("RIR_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_gather_u8(dst_u8, src_u8):
        if backend == "nccl":
            dist.all_gather_into_tensor(dst_u8, src_u8)
        else:  # rehearsal: gloo moves host memory
            host = torch.empty(dst_u8.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, src_u8.cpu())
            dst_u8.copy_(host)

    from librir_amd import device as D
    from librir_amd.synthetic import s1_noisy_background

    n, h, w, gop = args.frames, args.height, args.width, args.gop
    # every rank holds its own shard of the stream (different seed = different frames)
    frames_np = s1_noisy_background(n, h, w, seed=1234 + rank)
    frames = torch.from_numpy(frames_np).to(dev)
    ctx = D.CodecContext(w, h, n, gop, device=dev)
    out = torch.empty_like(frames)

    def step():
        enc = ctx.encode(frames)
        ctx.decode(enc, out=out, check=False)
        return enc

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        enc = step()
    barrier()

    # ---- timed region: exactly K steps; per-kernel HIP events on the stream the kernels run on ----
    K = args.steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record()
        ctx.encode_tiles(frames)
        ev[k][1].record()
        enc = ctx.encode_compact()
        ev[k][2].record()
        ctx.decode(enc, out=out, check=False)
        ev[k][3].record()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    ms_tiles = sum(ev[k][0].elapsed_time(ev[k][1]) for k in range(K)) / K
    ms_compact = sum(ev[k][1].elapsed_time(ev[k][2]) for k in range(K)) / K
    ms_decode = sum(ev[k][2].elapsed_time(ev[k][3]) for k in range(K)) / K

    # ---- parity gate before any number is reported: decoded stream == input, bit-exact ----
    assert int(ctx.error.item()) == 0
    ok = torch.equal(out.view(torch.int16), frames.view(torch.int16))
    cbytes = enc.compressed_bytes()
    payload_bytes = enc.total_words() * 8
    if not ok:
        raise SystemExit("bench.py: decode(encode(x)) != x - refusing to report a number")

    # ---- the exchange step, outside the timed region: all-gather of the decoded stream ----
    allgather = None
    if world > 1:
        gathered = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), dtype=torch.uint16, device=dev)
        all_gather_u8(gathered.view(torch.uint8), out.view(torch.uint8))  # warm-up / communicator setup
        barrier()
        t1 = time.perf_counter()
        all_gather_u8(gathered.view(torch.uint8), out.view(torch.uint8))
        barrier()
        ag = time.perf_counter() - t1
        same = bool(torch.equal(gathered[rank * out.shape[0]:(rank + 1) * out.shape[0]].view(torch.int16), out.view(torch.int16)))
        allgather = {"ms": ag * 1e3, "bytes_per_rank": out.numel() * 2 * world, "algbw_GBs": out.numel() * 2 * world / ag / 1e9,
                     "own_shard_intact": same, "backend": "rccl" if backend == "nccl" else backend}

    if rank == 0:
        raw = 2.0 * h * w * n  # bytes of raw uint16 per batch
        kernels = {
            "rirb1_encode_tiles": {"ms": ms_tiles, "alg_bytes": raw + payload_bytes + ctx.layout.hdr_bytes},
            "rirb1_scan_tiles+rirb1_compact": {"ms": ms_compact, "alg_bytes": 2.0 * payload_bytes},
            "rirb1_decode_tiles": {"ms": ms_decode, "alg_bytes": raw + payload_bytes + ctx.layout.hdr_bytes},
        }
        for kv in kernels.values():
            kv["GBs"] = kv["alg_bytes"] / (kv["ms"] * 1e-3) / 1e9
        dom = max(("rirb1_encode_tiles", "rirb1_decode_tiles"), key=lambda k_: kernels[k_]["ms"])
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
                                   "encode+decode, device-resident, GOP %d, per GPU" % (n, w, h, gop),
                       "frames_per_gpu": n, "width": w, "height": h, "gop": gop, "sharding": "independent shard per rank"},
            "bit_exact_roundtrip": True,
            "compression_ratio": raw / cbytes,
            "roundtrip_raw_GBs": fps * 4.0 * h * w / 1e9 / world,
            "roundtrip_raw_frac_of_hbm_peak": fps * 4.0 * h * w / 1e9 / world / HBM_PEAK_GBS,
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kernels[dom]["GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "ms_per_launch": kernels[dom]["ms"]},
            "kernels": kernels,
        }
        if allgather:
            res["allgather_decoded_stream"] = allgather
        if not args.no_cpu_baseline and world == 1:  # the CPU path is timed beside the N=1 run only
            res["cpu_baseline"] = cpu_baseline(frames_np, gop, args.cpu_frames, args.cpu_seconds)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
