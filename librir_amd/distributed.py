"""Multi-GPU sharding of a frame stream: one process per GPU, ``torch.distributed`` (RCCL on ROCm).

The path shards at chunk (GOP) boundaries: chunks are independent units of the codec (the structure the
reference gets from its key-frame cadence, src/cpp/video_io/h264.cpp:1052-1064), so every rank encodes /
decodes its own contiguous run of chunks with no collective in the data path.  The one exchange step
north_star names - reassembling the decoded stream on every GPU - exists in two forms:

* ``all_gather_frames`` / ``FrameGather``: an all-gather of the DECODED uint16 frames over xGMI, whole or in
  sub-batches that overlap the decode of the next sub-batch;
* ``CompressedGather``: an all-gather of the COMPRESSED chunks (``hdr`` / ``tile_off`` / ``stream``: 4.9x fewer
  bytes on the links for the reference's recipe) and a decode of every rank's chunks on arrival, straight to their
  place in the reassembled stream (``rir_codec_decode_chunks_device``).

Nothing here touches the codec itself: the consumer of a gathered piece is a callback (the HIP decoder on a GPU,
the oracle in the CPU tests).  Everything travels as raw bytes (uint8 views): RCCL and gloo move uint8 alike.
"""
import numpy as np
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


def _u8(t):
    return t.contiguous().view(torch.uint8).view(-1)


def _gather_bytes(recv, send, group=None, async_op=False):
    """``all_gather_into_tensor`` on the bytes of two contiguous tensors.  Returns a work handle (or None).

    RCCL moves device memory, gloo host memory.  gloo with device tensors (the one-GPU rehearsal of the N > 1
    control flow, bench.py RIR_BENCH_BACKEND=gloo) stages the bytes through the host, synchronously."""
    r8, s8 = recv.view(torch.uint8).view(-1), _u8(send)
    if dist.get_backend(group) == "gloo" and s8.is_cuda:
        host = torch.empty(r8.shape, dtype=torch.uint8)
        dist.all_gather_into_tensor(host, s8.cpu(), group=group)
        r8.copy_(host)
        return None
    return dist.all_gather_into_tensor(r8, s8, group=group, async_op=async_op)


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
    send = torch.zeros((maxc, h, w), dtype=torch.uint16, device=local_frames.device)
    if plan[rank][1]:
        send[: plan[rank][1]] = local_frames
    recv = torch.empty((world * maxc, h, w), dtype=torch.uint16, device=local_frames.device)  # concatenation along dim 0
    _gather_bytes(recv, send, group)
    if all(c == maxc for _, c in plan):
        return recv
    out = torch.empty((nframes, h, w), dtype=torch.uint16, device=local_frames.device)
    for r, (s, c) in enumerate(plan):
        if c:
            out[s:s + c] = recv[r * maxc:r * maxc + c]
    return out


class FrameGather:
    """Sub-batched all-gather of decoded frames that overlaps the decode of the next sub-batch.

    Every rank holds up to ``n`` frames in ``local`` (n = the largest shard, a whole number of pieces of ``piece`` frames;
    ``counts[r]`` = frames rank r really has - all n when ``counts`` is None: equal shards).  ``run(produce)`` calls
    ``produce(j, f0, f1)`` - which must leave frames [f0, f1) of the local shard in ``local`` on the current stream; f1 is
    clipped to this rank's count and ranks that have nothing in a piece are not called for it - and issues the all-gather of
    piece j behind it on the communicator's stream while piece j + 1 is produced.
    The gathered stream is PIECE-MAJOR: an ``all_gather_into_tensor`` output is the concatenation of its inputs over
    the ranks, so frame f of rank r lives at ``full[piece_index(f), r, f % piece]`` (``locate``); a rank-major copy
    would cost another pass over the whole stream in HBM (``assemble`` makes one, for callers that want stream order)."""

    def __init__(self, local, piece, group=None, counts=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local = local
        n = local.shape[0]
        self.counts = [n] * self.world if counts is None else [int(c) for c in counts]
        if len(self.counts) != self.world or any(c < 0 or c > n for c in self.counts):
            raise RuntimeError("FrameGather: one frame count per rank, none larger than the local buffer")
        self.piece = max(1, min(piece, n)) if n else 1
        self.bounds = [(f0, min(f0 + self.piece, n)) for f0 in range(0, n, self.piece)]
        if any(f1 - f0 != self.piece for f0, f1 in self.bounds):
            raise RuntimeError("FrameGather: the shard buffer must be a whole number of pieces")
        self.full = torch.empty((len(self.bounds), self.world, self.piece) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        per_frame = (local.numel() // n if n else 0) * local.element_size()
        # what the collective moves to this rank (padding of shorter shards included: the pieces are of one size)
        self.bytes_received = (self.world - 1) * local.numel() * local.element_size()
        self.valid_bytes_received = sum(c for r, c in enumerate(self.counts) if r != self.rank) * per_frame

    def locate(self, rank, f):
        return f // self.piece, rank, f % self.piece

    def assemble(self, out=None):
        """the gathered frames in stream order (rank after rank, each rank's frames in order): a copy, (sum(counts), ...)"""
        total = sum(self.counts)
        if out is None:
            out = torch.empty((total,) + tuple(self.local.shape[1:]), dtype=self.local.dtype, device=self.local.device)
        pos = 0
        for r, c in enumerate(self.counts):
            for f0 in range(0, c, self.piece):
                k = min(self.piece, c - f0)
                out[pos:pos + k] = self.full[f0 // self.piece, r, :k]
                pos += k
        return out

    def run(self, produce):
        works = []
        mine = self.counts[self.rank]
        for j, (f0, f1) in enumerate(self.bounds):
            if f0 < mine:
                produce(j, f0, min(f1, mine))
            works.append(_gather_bytes(self.full[j], self.local[f0:f1], self.group, async_op=True))
        for w in works:
            if w is not None:
                w.wait()  # RCCL: the current stream waits for the collective; gloo: the host does
        return self.full


class GatheredPiece:
    """Chunks [k0, k1) of EVERY rank's shard, gathered: table entry ``r * m + i`` is chunk ``k0 + i`` of rank r."""

    __slots__ = ("index", "m", "hdr", "tile_off", "chunk_off", "stream", "chunk_frames", "chunk_frames_host", "chunk_off_host")


class CompressedGather:
    """All-gather of the compressed chunks of every rank's shard, in pieces of ``chunks_per_piece`` chunks, each piece
    handed to ``consume`` while the next one is on the links.

    ``plan``: shard_plan() of the whole stream; this rank's tables describe ``ceil(plan[rank][1] / gop)`` chunks:
    ``hdr`` int64 [nchunks][ntiles][gop], ``tile_off`` int32 [nchunks][ntiles+1], ``chunk_off`` int64 [nchunks+1],
    ``stream`` int64 words (``rir_codec_encode_device`` layout).  One small blocking exchange of the
    chunk offset tables comes first (piece sizes must be known on the host to size the collectives); payload pieces
    are padded to the largest rank's piece.  ``consume(piece)`` sees, for every entry, where its frames go in the
    reassembled stream: ``piece.chunk_frames[e] = (first frame, frame count)`` (count 0 = padding entry)."""

    def __init__(self, plan, gop, ntiles, group=None, chunks_per_piece=4):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if len(plan) != self.world:
            raise RuntimeError("CompressedGather: plan does not match the process group")
        self.plan, self.gop, self.ntiles = plan, gop, ntiles
        self.nchunks = [(c + gop - 1) // gop for _, c in plan]
        self.kmax = max(self.nchunks)
        self.m = max(1, min(chunks_per_piece, self.kmax)) if self.kmax else 1
        self.pieces = [(k0, min(k0 + self.m, self.kmax)) for k0 in range(0, self.kmax, self.m)]
        self._bufs = {}
        self.bytes_received = 0  # of the last run(): payload + tables received from the other ranks

    def _buf(self, key, shape, dtype, device):
        t = self._bufs.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype or t.device != device:
            t = torch.empty((n,), dtype=dtype, device=device)
            self._bufs[key] = t
        return t[:n].view(shape)

    def exchange_tables(self, chunk_off):
        """-> int64 numpy [world][kmax + 1]: every rank's chunk offsets (rows of shorter shards repeat their total)."""
        mine = torch.zeros((self.kmax + 1,), dtype=torch.int64)
        nc = self.nchunks[self.rank]
        co = chunk_off[: nc + 1].to("cpu", torch.int64)  # (host synchronisation: the sizes of the collectives depend on it)
        mine[: nc + 1] = co
        if nc < self.kmax:
            mine[nc + 1:] = co[nc]
        n = self.kmax + 1
        if dist.get_backend(self.group) == "gloo":
            allt = torch.empty((self.world * n,), dtype=torch.int64)
            dist.all_gather_into_tensor(allt, mine, group=self.group)
        else:
            dev = chunk_off.device
            d_all = torch.empty((self.world * n,), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(d_all, mine.to(dev), group=self.group)
            allt = d_all.cpu()
        return allt.view(self.world, n).numpy()

    def _tables_of(self, j, coff):
        """host tables of piece j: chunk_off of the gathered layout, (first frame, count) per entry, padded piece length"""
        k0, k1 = self.pieces[j]
        m = self.m
        lens = [int(coff[r][min(k1, self.nchunks[r])] - coff[r][min(k0, self.nchunks[r])]) for r in range(self.world)]
        maxw = max(max(lens), 1)
        g_off = np.zeros((self.world * m + 1,), np.int64)
        g_frames = np.zeros((self.world * m, 2), np.int64)
        for r in range(self.world):
            base = int(coff[r][min(k0, self.nchunks[r])])
            start, count = self.plan[r]
            for i in range(m):
                k = k0 + i
                g_off[r * m + i] = r * maxw + int(coff[r][min(k, self.nchunks[r])]) - base
                if k < self.nchunks[r]:
                    g_frames[r * m + i] = (start + k * self.gop, min(self.gop, count - k * self.gop))
        g_off[self.world * m] = self.world * maxw
        return g_off, g_frames, maxw

    def run(self, hdr, tile_off, chunk_off, stream, consume):
        coff = self.exchange_tables(chunk_off)
        dev = stream.device
        nc, m, W = self.nchunks[self.rank], self.m, self.world
        ntiles, gop = self.ntiles, self.gop
        tabs = [self._tables_of(j, coff) for j in range(len(self.pieces))]
        # the small per-piece tables of the whole run go to the device in one copy
        all_off = torch.from_numpy(np.concatenate([t[0] for t in tabs]))
        all_frames = torch.from_numpy(np.concatenate([t[1].reshape(-1) for t in tabs]))
        d_off, d_frames = (all_off.to(dev), all_frames.to(dev)) if dev.type != "cpu" else (all_off, all_frames)
        self.bytes_received = 0
        pending = None
        for j, (k0, k1) in enumerate(self.pieces):
            g_off, g_frames, maxw = tabs[j]
            # --- this rank's contribution: whole slices of its own tables where they exist, zero-padded copies otherwise ---
            if k0 + m <= nc:
                s_hdr, s_toff = hdr[k0:k0 + m], tile_off[k0:k0 + m]
            else:
                s_hdr = torch.zeros((m, ntiles, gop), dtype=hdr.dtype, device=dev)
                s_toff = torch.zeros((m, ntiles + 1), dtype=tile_off.dtype, device=dev)
                if k0 < nc:
                    s_hdr[: nc - k0] = hdr[k0:nc]
                    s_toff[: nc - k0] = tile_off[k0:nc]
            w0 = int(coff[self.rank][min(k0, nc)])
            if w0 + maxw <= stream.numel():
                s_stream = stream[w0:w0 + maxw]  # (runs on into the next piece's words: padding that nobody reads)
            else:
                s_stream = torch.zeros((maxw,), dtype=stream.dtype, device=dev)
                s_stream[: stream.numel() - w0] = stream[w0:]
            p = GatheredPiece()
            p.index, p.m = j, m
            p.hdr = self._buf(("hdr", j & 1), (W * m, ntiles, gop), hdr.dtype, dev)
            p.tile_off = self._buf(("toff", j & 1), (W * m, ntiles + 1), tile_off.dtype, dev)
            p.stream = self._buf(("stream", j & 1), (W * maxw,), stream.dtype, dev)
            o0 = sum(t[0].size for t in tabs[:j])
            f0 = sum(t[1].size for t in tabs[:j])
            p.chunk_off = d_off[o0:o0 + g_off.size]
            p.chunk_frames = d_frames[f0:f0 + g_frames.size].view(-1, 2)
            p.chunk_off_host, p.chunk_frames_host = g_off, g_frames
            works = [_gather_bytes(p.hdr, s_hdr, self.group, True), _gather_bytes(p.tile_off, s_toff, self.group, True),
                     _gather_bytes(p.stream, s_stream, self.group, True)]
            self.bytes_received += (W - 1) * (s_hdr.numel() * 8 + s_toff.numel() * 4 + maxw * 8)
            if pending is not None:  # piece j - 1 is consumed while piece j is on the links
                self._finish(pending, consume)
            pending = (p, works)
        if pending is not None:
            self._finish(pending, consume)

    @staticmethod
    def _finish(pending, consume):
        p, works = pending
        for w in works:
            if w is not None:
                w.wait()
        consume(p)
