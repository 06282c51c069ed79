"""CPU, world_size 2 and 4, gloo: the sharding plan and the two exchange steps that reassemble the stream on every rank
(the N>1 path of bench.py / DESIGN.md §8): the all-gather of decoded frames, whole and in overlapped sub-batches,
and the all-gather of COMPRESSED chunks decoded on arrival.  The HIP codec is not run here (GPU only): frames are
fabricated from the global frame index, and the compressed exchange is checked with the oracle as encoder/decoder."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from librir_amd.distributed import CompressedGather, FrameGather, all_gather_frames, shard_plan


def test_shard_plan_is_chunk_aligned_and_covers_everything():
    for nframes, gop, world in [(1000, 50, 8), (10000, 50, 8), (120, 50, 2), (49, 50, 4), (0, 50, 2), (101, 10, 3), (1000, 50, 1)]:
        plan = shard_plan(nframes, gop, world)
        assert len(plan) == world
        pos = 0
        for start, count in plan:
            assert start == pos and count >= 0
            assert start % gop == 0 or count == 0  # every shard starts on a key frame
            pos += count
        assert pos == nframes
        counts = [-(-c // gop) for _, c in plan]
        assert max(counts) - min(counts) <= 1  # chunks dealt evenly
    assert shard_plan(1000, 50, 8)[0] == (0, 150) and shard_plan(1000, 50, 8)[7] == (900, 100)
    with pytest.raises(ValueError):
        shard_plan(10, 0, 2)


def _worker(rank, world, port, nframes, gop, h, w, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = shard_plan(nframes, gop, world)
        start, count = plan[rank]
        idx = torch.arange(start, start + count, dtype=torch.int32).view(-1, 1, 1)
        local = ((idx * 7 + torch.arange(h * w, dtype=torch.int32).view(1, h, w)) % 65536).to(torch.uint16)
        full = all_gather_frames(local, plan)
        exp = ((torch.arange(nframes, dtype=torch.int32).view(-1, 1, 1) * 7 + torch.arange(h * w, dtype=torch.int32).view(1, h, w)) % 65536)
        ok = full.shape == (nframes, h, w) and torch.equal(full.view(torch.int16), exp.to(torch.uint16).view(torch.int16))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nframes,gop", [(200, 50), (130, 50), (40, 50)])
def test_all_gather_reassembles_the_stream_world2(nframes, gop):
    world = 2
    port = 29500 + (os.getpid() + nframes) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, nframes, gop, 6, 9, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _piece_worker(rank, world, port, ret):
    """FrameGather: pieces gathered while the next one is produced; piece-major layout, every frame accounted for."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, h, w, piece = 12, 5, 7, 4
        local = torch.zeros((n, h, w), dtype=torch.uint16)
        fg = FrameGather(local, piece)
        calls = []

        def produce(j, f0, f1):
            calls.append((j, f0, f1))
            for f in range(f0, f1):
                local[f] = (rank * 1000 + f) % 65536

        full = fg.run(produce)
        ok = calls == [(0, 0, 4), (1, 4, 8), (2, 8, 12)] and tuple(full.shape) == (3, world, piece, h, w)
        for r in range(world):
            for f in range(n):
                ok = ok and int(full[fg.locate(r, f)][0, 0]) == r * 1000 + f
        ret[rank] = bool(ok) and fg.bytes_received == (world - 1) * n * h * w * 2
    finally:
        dist.destroy_process_group()


def test_sub_batched_frame_gather_world2():
    world = 2
    port = 29500 + (os.getpid() + 777) % 2000
    ret = mp.Manager().dict()
    mp.spawn(_piece_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _compressed_worker(rank, world, port, nframes, gop, h, w, per_piece, ret):
    """Every rank encodes its shard with the oracle, the compressed chunks are gathered piece by piece and decoded on
    arrival (oracle again) into their place in the whole stream - on every rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from librir_amd.synthetic import s1_noisy_background
        from oracle.pyoracle import Oracle

        O = Oracle()
        whole = s1_noisy_background(nframes, h, w, seed=5)
        plan = shard_plan(nframes, gop, world)
        start, count = plan[rank]
        ntiles = (h * w + 511) // 512
        nc = (count + gop - 1) // gop
        hdr = np.zeros((max(nc, 1), ntiles, gop), np.uint64)
        toff = np.zeros((max(nc, 1), ntiles + 1), np.uint32)
        coff = np.zeros(nc + 1, np.int64)
        words = []
        for k in range(nc):
            fr = whole[start + k * gop:min(start + (k + 1) * gop, start + count)]
            hk, ok_, sk = O.codec_encode_chunk(fr)
            hdr[k, :, : fr.shape[0]] = hk
            toff[k] = ok_
            words.append(sk)
            coff[k + 1] = coff[k] + sk.size
        stream = np.concatenate(words + [np.zeros(0, np.uint64)]) if words else np.zeros(0, np.uint64)
        out = np.zeros_like(whole)
        seen = []

        def consume(p):
            st = p.stream.numpy().view(np.uint64)
            for e in range(p.chunk_frames_host.shape[0]):
                f0, cnt = (int(v) for v in p.chunk_frames_host[e])
                if cnt == 0:
                    continue
                c0, c1 = int(p.chunk_off_host[e]), int(p.chunk_off_host[e + 1])
                hd = np.ascontiguousarray(p.hdr[e].numpy().view(np.uint64)[:, :cnt])
                out[f0:f0 + cnt] = O.codec_decode_chunk(hd, p.tile_off[e].numpy().view(np.uint32), st[c0:c1], w, h)
                seen.append((f0, cnt))

        cg = CompressedGather(plan, gop, ntiles, chunks_per_piece=per_piece)
        cg.run(torch.from_numpy(hdr.view(np.int64)), torch.from_numpy(toff.view(np.int32)), torch.from_numpy(coff),
               torch.from_numpy(stream.view(np.int64)), consume)
        ok = np.array_equal(out, whole) and sum(c for _, c in seen) == nframes and cg.bytes_received > 0
        # the point of the exercise: fewer bytes on the links than the decoded frames would take
        if nframes - count >= 2 * gop:  # (a rank that receives (almost) nothing still moves its padded tables)
            ok = ok and cg.bytes_received < (nframes - count) * h * w * 2
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nframes,gop,per_piece,shape", [(50, 10, 2, (768, 1024)), (35, 10, 4, (67, 83)), (10, 10, 1, (32, 64))])
def test_compressed_chunks_gathered_and_decoded_on_arrival_world2(nframes, gop, per_piece, shape):
    """(50, 10): 5 chunks over 2 ranks = 3 + 2 (a padding entry in the last piece) at the config-3 frame geometry 1024x768;
    (35, 10): a short last chunk; (10, 10): one rank has no frames at all."""
    world = 2
    port = 29500 + (os.getpid() + nframes * 7 + per_piece) % 2000
    ret = mp.Manager().dict()
    mp.spawn(_compressed_worker, args=(world, port, nframes, gop, shape[0], shape[1], per_piece, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


# ---- world 4: uneven shards, a rank without frames ---------------------------------------------------------------------------


@pytest.mark.parametrize("nframes,gop", [(230, 50), (120, 50)])
def test_all_gather_reassembles_the_stream_world4(nframes, gop):
    """(230, 50): 5 chunks over 4 ranks = 2 + 1 + 1 + 1, the last one short; (120, 50): 3 chunks, rank 3 has no frames"""
    world = 4
    port = 29500 + (os.getpid() + nframes + 11) % 2000
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, nframes, gop, 6, 9, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _uneven_piece_worker(rank, world, port, nframes, gop, piece, ret):
    """FrameGather with the shards of shard_plan: pieces gathered while the next one is produced, ranks that have run out of
    frames contribute padding and are not asked to produce; every real frame is found where locate() says, assemble() gives
    the stream in order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w = 5, 7
        plan = shard_plan(nframes, gop, world)
        counts = [c for _, c in plan]
        nmax = -(-max(counts) // piece) * piece
        local = torch.zeros((nmax, h, w), dtype=torch.uint16)
        fg = FrameGather(local, piece, counts=counts)
        calls = []

        def produce(j, f0, f1):
            calls.append((j, f0, f1))
            for f in range(f0, f1):
                local[f] = (plan[rank][0] + f) * 3 % 65536  # a function of the GLOBAL frame index

        full = fg.run(produce)
        mine = counts[rank]
        exp_calls = [(f0 // piece, f0, min(f0 + piece, mine)) for f0 in range(0, mine, piece)]
        ok = calls == exp_calls and tuple(full.shape) == (nmax // piece, world, piece, h, w)
        for r in range(world):
            for f in range(counts[r]):
                ok = ok and int(full[fg.locate(r, f)][0, 0]) == (plan[r][0] + f) * 3 % 65536
        stream = fg.assemble()
        ok = ok and stream.shape[0] == nframes and all(int(stream[g][h - 1, w - 1]) == g * 3 % 65536 for g in range(nframes))
        ok = ok and fg.valid_bytes_received == (nframes - mine) * h * w * 2 and fg.bytes_received == (world - 1) * nmax * h * w * 2
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nframes,gop,piece", [(230, 50, 50), (120, 50, 25), (30, 10, 10)])
def test_sub_batched_frame_gather_world4_uneven(nframes, gop, piece):
    """(230, 50): shards of 100 / 50 / 50 / 30 frames; (120, 50): 50 / 50 / 20 / 0 - a rank without frames, pieces smaller than a
    chunk; (30, 10): 3 chunks over 4 ranks"""
    world = 4
    port = 29500 + (os.getpid() + nframes * 3 + piece) % 2000
    ret = mp.Manager().dict()
    mp.spawn(_uneven_piece_worker, args=(world, port, nframes, gop, piece, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


@pytest.mark.parametrize("nframes,gop,per_piece,shape", [(70, 10, 2, (67, 83)), (25, 10, 1, (32, 64)), (64, 8, 4, (40, 52))])
def test_compressed_chunks_gathered_and_decoded_on_arrival_world4(nframes, gop, per_piece, shape):
    """(70, 10): 7 chunks over 4 ranks = 2 + 2 + 2 + 1 (padding entries in rank 3's pieces); (25, 10): 3 chunks - rank 3 has no
    frames - and a short last chunk; (64, 8): even shards, one piece"""
    world = 4
    port = 29500 + (os.getpid() + nframes * 5 + per_piece + 3) % 2000
    ret = mp.Manager().dict()
    mp.spawn(_compressed_worker, args=(world, port, nframes, gop, shape[0], shape[1], per_piece, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


# ---- the N > 1 bench line proves what it ran on (bench.py: world_identity / check_world) ------------------------------------
def _identity_worker(rank, world, port, ret):
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen, version, everyone = bench.world_identity(dist, torch, None, world, rank, "gloo")
        ids = [e["device_key"] for e in everyone]
        ret[rank] = (seen, version, [e["rank"] for e in everyone], bench.check_world(seen, world, ids, False), bench.check_world(seen, world, ids, True))
    finally:
        dist.destroy_process_group()


def test_bench_refuses_a_world_it_cannot_vouch_for():
    import bench

    # every rank on the communicator, every rank a device of its own: accepted
    assert bench.check_world(8, 8, ["0000:%02x:00" % i for i in range(8)], False) is None
    # a communicator that saw fewer ranks than --gpus, or two ranks on one GPU (unless the rehearsal switch says so): refused
    assert "saw 7 rank(s)" in bench.check_world(7, 8, ["0000:%02x:00" % i for i in range(8)], False)
    assert "same device" in bench.check_world(2, 2, ["0000:05:00", "0000:05:00"], False)
    assert bench.check_world(2, 2, ["0000:05:00", "0000:05:00"], True) is None
    # two gloo ranks on this CPU: the all-reduce sees both, both report the same "device" - refused without the switch, accepted with it
    world = 2
    port = 29500 + (os.getpid() + 977) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_identity_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        seen, version, ranks, refused, rehearsal = ret[r]
        assert seen == 2 and version is None and ranks == [0, 1]
        assert refused is not None and "same device" in refused and rehearsal is None


def test_bench_reads_the_newest_n1_record():
    import bench

    ref1 = bench.newest_n1_value()
    assert ref1 is None or (ref1["file"].startswith("BENCH_r") and ref1["value"] > 0)
