"""Multi-GPU sharding of a frame stream: one process per GPU, ``torch.distributed`` (RCCL on ROCm).

The path shards at chunk (GOP) boundaries: chunks are independent units of the codec, so every rank
encodes / decodes its own contiguous run of chunks with no collective in the data path.  The one
exchange step north_star names - reassembling the decoded stream on every GPU - is an all-gather of
uint16 frames over xGMI (``all_gather_into_tensor``; shards are padded to the largest one).
"""
import torch
import torch.distributed as dist


def shard_plan(nframes, gop, world_size):
    """Contiguous, chunk-aligned frame ranges [(start, count)] * world_size.

    Chunks (GOPs) are dealt as evenly as possible, the first ``nchunks % world_size`` ranks get one
    more; a rank may receive no frames when there are fewer chunks than ranks."""
    if nframes < 0 or gop <= 0 or world_size <= 0:
        raise ValueError("shard_plan: invalid argument")
    nchunks = (nframes + gop - 1) // gop
    base, extra = divmod(nchunks, world_size)
    plan, chunk = [], 0
    for r in range(world_size):
        c = base + (1 if r < extra else 0)
        start = min(chunk * gop, nframes)
        stop = min((chunk + c) * gop, nframes)
        plan.append((start, stop - start))
        chunk += c
    return plan


def all_gather_frames(local_frames, plan, group=None):
    """local_frames: (count_r, H, W) uint16 of this rank (count_r = plan[rank][1]).
    Returns the reassembled (nframes, H, W) uint16 stream on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(plan) != world or local_frames.shape[0] != plan[rank][1]:
        raise RuntimeError("all_gather_frames: plan does not match the process group / local shard")
    h, w = local_frames.shape[1:]
    maxc = max(c for _, c in plan)
    nframes = sum(c for _, c in plan)
    # uint16 travels as raw bytes: every backend (RCCL, gloo) moves uint8, none needs to interpret it
    send = torch.zeros((maxc, h, w), dtype=torch.uint16, device=local_frames.device)
    if plan[rank][1]:
        send[: plan[rank][1]] = local_frames
    recv = torch.empty((world * maxc, h, w), dtype=torch.uint16, device=local_frames.device)  # concatenation along dim 0
    dist.all_gather_into_tensor(recv.view(torch.uint8), send.view(torch.uint8), group=group)
    if all(c == maxc for _, c in plan):
        return recv
    out = torch.empty((nframes, h, w), dtype=torch.uint16, device=local_frames.device)
    for r, (s, c) in enumerate(plan):
        if c:
            out[s:s + c] = recv[r * maxc:r * maxc + c]
    return out
