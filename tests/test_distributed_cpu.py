"""CPU, world_size 2, gloo: the sharding plan and the all-gather that reassembles the decoded
stream (the N>1 path of bench.py / DESIGN.md §6).  The codec itself is not run here (GPU only):
each rank fabricates the frames of its shard from the global frame index."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from librir_amd.distributed import all_gather_frames, shard_plan


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
