"""GPU: the HIP codec through the C ABI (rir_codec_*_device) against the oracle (bit-exact
bitstream and tables), the lossless identity at BASELINE sizes, and malformed-input handling."""
import numpy as np
import pytest

from librir_amd.synthetic import s1_noisy_background, s2_uniform_dl_ti

pytestmark = pytest.mark.gpu


def to_np(enc):
    return (enc.hdr.cpu().numpy().view(np.uint64), enc.tile_off.cpu().numpy().view(np.uint32), enc.chunk_off.cpu().numpy(),
            enc.stream.cpu().numpy().view(np.uint64))


def encode_decode(dev, frames, gop, single_pass=False):
    import torch

    n, h, w = frames.shape
    ctx = dev.CodecContext(w, h, n, gop)
    t = torch.from_numpy(frames).cuda()
    enc = ctx.encode(t, single_pass=single_pass)
    dec = ctx.decode(enc)
    torch.cuda.synchronize()
    if single_pass:
        assert ctx.encode_status() == 0
    return ctx, enc, dec.cpu().numpy()


CASES = [
    ("random_ragged", (5, 67, 83), 3),
    ("noisy_small", (7, 48, 64), 50),
    ("tiny", (3, 20, 20), 2),
    ("one_pixel", (2, 1, 1), 50),
    ("long_chunk", (130, 16, 32), 128),
    ("uniform", (4, 64, 64), 50),
    ("ref_shape_240x320", (10, 240, 320), 50),
    ("ref_shape_256x320", (10, 256, 320), 4),
    ("single_frame_512x640", (1, 512, 640), 50),
    ("odd_npx", (3, 3, 1025), 2),
]


@pytest.mark.parametrize("single_pass", [False, True], ids=["two_pass", "single_pass"])
@pytest.mark.parametrize("name,shape,gop", CASES)
def test_bitstream_equals_oracle(dev, oracle, name, shape, gop, single_pass):
    """Both encoders (the two-pass one behind rir_codec_encode_device and the single-pass look-back one) against the oracle:
    header table, tile offsets, chunk offsets and stream bit for bit."""
    n, h, w = shape
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    if name == "noisy_small":
        fr = s1_noisy_background(n, h, w)
    elif name == "uniform":
        fr = s2_uniform_dl_ti(n, h, w)
    elif name == "long_chunk":
        fr = (np.cumsum(np.ones((n, h, w), np.uint32), axis=2) * 5 + np.arange(n)[:, None, None] * 300).astype(np.uint16)
    elif name.startswith("ref_shape"):
        fr = s1_noisy_background(n, h, w, seed=9)
    else:
        fr = rng.integers(0, 65536, shape).astype(np.uint16)
    ctx, enc, dec = encode_decode(dev, fr, gop, single_pass)
    assert np.array_equal(dec, fr)
    hdr, toff, coff, st = to_np(enc)
    L = ctx.layout
    assert coff[0] == 0
    for c in range(L.nchunks):
        f0 = c * gop
        nf = min(gop, n - f0)
        h_o, o_o, st_o = oracle.codec_encode_chunk(fr[f0:f0 + nf])
        assert np.array_equal(hdr[c][:, :nf], h_o), (name, c)
        assert (hdr[c][:, nf:] == 0).all()
        assert np.array_equal(toff[c], o_o), (name, c)
        assert np.array_equal(st[coff[c]:coff[c + 1]], st_o), (name, c)
        # and the oracle decodes what the GPU wrote
        assert np.array_equal(oracle.codec_decode_chunk(hdr[c][:, :nf], toff[c], st[coff[c]:coff[c + 1]], w, h), fr[f0:f0 + nf])


def test_gpu_decodes_oracle_stream(dev, oracle):
    """Streams written by the CPU restatement decode on the GPU (cross-read)."""
    import torch

    n, h, w, gop = 9, 40, 52, 4
    fr = s1_noisy_background(n, h, w, seed=3)
    ctx = dev.CodecContext(w, h, n, gop)
    L = ctx.layout
    hdr = np.zeros((L.nchunks, L.ntiles, gop), np.uint64)
    toff = np.zeros((L.nchunks, L.ntiles + 1), np.uint32)
    coff = np.zeros(L.nchunks + 1, np.int64)
    parts = []
    for c in range(L.nchunks):
        nf = min(gop, n - c * gop)
        h_o, o_o, s_o = oracle.codec_encode_chunk(fr[c * gop:c * gop + nf])
        hdr[c][:, :nf] = h_o
        toff[c] = o_o
        coff[c + 1] = coff[c] + s_o.size
        parts.append(s_o)
    st = np.concatenate(parts + [np.zeros(1, np.uint64)])
    enc = dev.EncodedBatch(L, torch.from_numpy(hdr.view(np.int64)).cuda(), torch.from_numpy(toff.view(np.int32)).cuda(),
                           torch.from_numpy(coff).cuda(), torch.from_numpy(st.view(np.int64)).cuda())
    dec = ctx.decode(enc)
    assert np.array_equal(dec.cpu().numpy(), fr)


@pytest.mark.parametrize("shape,n", [((512, 640), 1000), ((768, 1024), 250)])
def test_identity_at_baseline_sizes(dev, shape, n):
    """configs[1] (and the 1024x768 geometry of configs[3]) at full size: size-independent
    properties - identity, idempotent re-encode, additivity of the chunk table."""
    import torch

    h, w = shape
    fr = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
    ctx = dev.CodecContext(w, h, n, 50)
    enc = ctx.encode(fr)
    out = ctx.decode(enc)
    assert torch.equal(out.view(torch.int16), fr.view(torch.int16))
    hdr1 = enc.hdr.clone()
    words1 = enc.total_words()
    coff = enc.chunk_off.cpu().numpy()
    toff = enc.tile_off.cpu().numpy().view(np.uint32).astype(np.int64)
    assert (np.diff(coff) == toff[:, -1]).all() and (np.diff(toff, axis=1) >= 0).all()
    enc2 = ctx.encode(out)  # re-encoding the decoded stream reproduces the same tables
    assert torch.equal(enc2.hdr, hdr1) and enc2.total_words() == words1
    ratio = fr.numel() * 2 / enc.compressed_bytes()
    assert ratio > 4.0, ratio  # reference claims "about 5" on its own recipe (docs/video_io.md:13)


def test_single_pass_encoder_equals_two_pass_at_baseline_size_and_when_it_spills(dev):
    """1 000 x 640x512 (12 800 segments, several rounds of workgroups, look-back across 20 chunks) and a noisy stream whose
    segments overflow their LDS staging (spill path): tables and stream equal the two-pass encoder's, word for word."""
    import torch

    for n, h, w, noise in ((1000, 512, 640, 0), (150, 256, 320, 3000), (70, 67, 83, 60000)):
        fr = s1_noisy_background(n, h, w, seed=11).astype(np.int64)
        if noise:
            fr = fr + np.random.default_rng(2).integers(0, noise, fr.shape)
        t = torch.from_numpy((fr % 65536).astype(np.uint16)).cuda()
        ctx = dev.CodecContext(w, h, n, 50)
        a = ctx.encode(t)
        ref = [x.clone() for x in (a.hdr, a.tile_off, a.chunk_off)]
        words = a.total_words()
        ref_stream = a.stream[:words].clone()
        for x in (a.hdr, a.tile_off, a.chunk_off):
            x.zero_()
        a.stream[:words].zero_()
        b = ctx.encode(t, single_pass=True)
        assert ctx.encode_status() == 0
        assert torch.equal(b.hdr, ref[0]) and torch.equal(b.tile_off, ref[1]) and torch.equal(b.chunk_off, ref[2]), (n, h, w)
        assert b.total_words() == words and torch.equal(b.stream[:words], ref_stream), (n, h, w)
        assert torch.equal(ctx.decode(b).view(torch.int16), t.view(torch.int16))


def test_malformed_tables_are_rejected_not_read_out_of_bounds(dev):
    import torch

    n, h, w, gop = 6, 32, 64, 3
    fr = torch.from_numpy(np.random.default_rng(0).integers(0, 65536, (n, h, w)).astype(np.uint16)).cuda()
    ctx = dev.CodecContext(w, h, n, gop)
    enc = ctx.encode(fr)
    good_hdr = enc.hdr.clone()
    enc.hdr[0, 0, 1] |= 0x1F  # width 31
    with pytest.raises(RuntimeError):
        ctx.decode(enc)
    enc.hdr.copy_(good_hdr)
    enc.tile_off[1, -1] -= 1  # last segment of chunk 1 one word short
    with pytest.raises(RuntimeError):
        ctx.decode(enc)
    enc.tile_off[1, -1] += 1
    assert torch.equal(ctx.decode(enc).view(torch.int16), fr.view(torch.int16))


def test_offsets_that_point_outside_the_stream_are_rejected_before_any_read(dev):
    """Tables come from files: a segment is only read when [chunk_off + tile_off[t], chunk_off + tile_off[t+1]) lies
    inside its chunk and the chunk inside the `stream_words` the caller vouches for (two consecutive huge
    tile_off entries used to put the buffer descriptor's base gigabytes past the allocation)."""
    import torch

    n, h, w, gop = 6, 32, 64, 3
    fr = torch.from_numpy(np.random.default_rng(1).integers(0, 65536, (n, h, w)).astype(np.uint16)).cuda()
    ctx = dev.CodecContext(w, h, n, gop)
    enc = ctx.encode(fr)
    torch.cuda.synchronize()
    good_toff, good_coff = enc.tile_off.clone(), enc.chunk_off.clone()
    words = enc.total_words()

    def rejected():
        with pytest.raises(RuntimeError):
            ctx.decode(enc)
        enc.tile_off.copy_(good_toff)
        enc.chunk_off.copy_(good_coff)

    # two consecutive huge entries: monotone (t1 >= t0), far outside the chunk
    enc.tile_off[0, 1] = np.int32(-0x100)  # 0xFFFFFF00 as uint32
    enc.tile_off[0, 2] = np.int32(-0x80)
    rejected()
    # a huge chunk offset (would move every segment of chunk 1 ~ 2^40 words away), and a decreasing pair
    enc.chunk_off[1] = 1 << 40
    enc.chunk_off[2] = (1 << 40) + int(good_coff[2] - good_coff[1])
    rejected()
    enc.chunk_off[1] = int(good_coff[2]) + 5
    rejected()
    # the caller's stream length is the bound: the same tables against a shorter stream are refused
    short = dev.EncodedBatch(enc.layout, enc.hdr, enc.tile_off, enc.chunk_off, enc.stream[:words - 1])
    with pytest.raises(RuntimeError):
        ctx.decode(short)
    exact = dev.EncodedBatch(enc.layout, enc.hdr, enc.tile_off, enc.chunk_off, enc.stream[:words])
    assert torch.equal(ctx.decode(exact).view(torch.int16), fr.view(torch.int16))


def test_argument_errors(dev):
    import torch

    ctx = dev.CodecContext(64, 32, 4, 2)
    with pytest.raises(RuntimeError):
        ctx.encode(torch.zeros((4, 32, 65), dtype=torch.uint16, device="cuda"))
    with pytest.raises(RuntimeError):
        ctx.encode(torch.zeros((4, 32, 64), dtype=torch.int16, device="cuda"))


def test_gpu_encoder_reproduces_the_frozen_format(dev):
    """The GPU encoder against tests/golden/codec_format.json directly (hashes of tables and payload)."""
    import json
    import os
    import sys

    import torch

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    import make_codec_golden as G

    fix = json.load(open(os.path.join(here, "codec_format.json")))
    for name, fr in G.streams():
        n, h, w = fr.shape
        ctx = dev.CodecContext(w, h, n, max(n, 1))
        enc = ctx.encode(torch.from_numpy(fr).cuda())
        words = int(enc.total_words())
        hdr = enc.hdr.cpu().numpy().view(np.uint64)[0][:, :n]
        off = enc.tile_off.cpu().numpy().view(np.uint32)[0]
        st = enc.stream.cpu().numpy().view(np.uint64)[:words]
        c = fix["cases"][name]
        assert (G.digest(hdr), G.digest(off), G.digest(st), words) == (c["hdr"], c["tile_off"], c["stream"], c["words"]), name


def test_identity_with_offsets_beyond_4_gib(dev):
    """3 000 frames of 1024x768 = 4.7 GB of raw frames in one batch: every frame / slot / stream offset of the second
    half needs 64-bit arithmetic (the MI355X holds 288 GB: batches of this size are the intended use)."""
    import torch

    n, h, w = 3000, 768, 1024
    base = torch.from_numpy(s1_noisy_background(250, h, w, seed=3)).cuda()
    t = base.repeat(n // 250, 1, 1).contiguous()
    t.view(torch.int16)[1::250] += 3
    ctx = dev.CodecContext(w, h, n, 50)
    enc = ctx.encode(t)
    out = ctx.decode(enc)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))
    assert t.numel() * 2 / enc.compressed_bytes() > 4
    # the last chunk alone decodes to the last 50 frames: tables of a late chunk are self-contained
    last = ctx.layout.nchunks - 1
    coff = enc.chunk_off.cpu().numpy()
    sub = dev.CodecContext(w, h, 50, 50)
    sub.hdr.copy_(enc.hdr[last:last + 1])
    sub.tile_off.copy_(enc.tile_off[last:last + 1])
    sub.chunk_off.copy_(torch.tensor([0, coff[last + 1] - coff[last]], dtype=torch.int64))
    nw = int(coff[last + 1] - coff[last])
    sub.stream[:nw].copy_(enc.stream[int(coff[last]):int(coff[last + 1])])
    from librir_amd.device import EncodedBatch

    dec = sub.decode(EncodedBatch(sub.layout, sub.hdr, sub.tile_off, sub.chunk_off, sub.stream))
    assert torch.equal(dec.view(torch.int16), t[-50:].view(torch.int16))


def test_workspace_placement_keeps_the_results(oracle):
    """CodecContext.place_workspace (rir_codec_workspace_create_device: the library allocates candidates, times the packing
    kernel on each, keeps one, frees the rest): the encoded batch is the same bit for bit afterwards, in both forms."""
    import torch

    from librir_amd import device as D
    from librir_amd.synthetic import s1_noisy_background

    n, h, w = 100, 128, 160
    fr = torch.from_numpy(s1_noisy_background(n, h, w, seed=5)).cuda()
    ctx = D.CodecContext(w, h, n, 25)
    e0 = ctx.encode(fr)
    words = int(e0.total_words())
    ref = (e0.hdr.clone(), e0.tile_off.clone(), e0.chunk_off.clone(), e0.stream[:words].clone())
    times = ctx.place_workspace(fr, tries=3, spacing_bytes=64 << 20)
    assert 1 <= len(times) <= 4 and times[0] == min(times)
    e1 = ctx.encode(fr)
    assert int(e1.total_words()) == words
    assert torch.equal(e1.hdr, ref[0]) and torch.equal(e1.tile_off, ref[1]) and torch.equal(e1.chunk_off, ref[2]) and torch.equal(e1.stream[:words], ref[3])
    out = ctx.decode(e1)
    assert torch.equal(out.view(torch.int16), fr.view(torch.int16))
    ctx.encode_tiles(fr)
    assert torch.equal(ctx.decode_slots().view(torch.int16), fr.view(torch.int16))
    del ctx  # (the library's allocation goes with the context)
    torch.cuda.synchronize()


def _case_frames(name, shape):
    n, h, w = shape
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    if name == "noisy_small":
        return s1_noisy_background(n, h, w)
    if name == "uniform":
        return s2_uniform_dl_ti(n, h, w)
    if name == "long_chunk":
        return (np.cumsum(np.ones((n, h, w), np.uint32), axis=2) * 5 + np.arange(n)[:, None, None] * 300).astype(np.uint16)
    if name.startswith("ref_shape"):
        return s1_noisy_background(n, h, w, seed=9)
    return rng.integers(0, 65536, shape).astype(np.uint16)


@pytest.mark.parametrize("name,shape,gop", CASES)
def test_slotted_form_equals_oracle_and_decodes_without_the_second_pass(dev, oracle, name, shape, gop):
    """What stage 1 of the encoder leaves (rir_codec_encode_tiles_device: headers + one length and one slot per segment) is a
    complete encoded batch: segment by segment it is the oracle's stream (same words, located by position instead of by
    offsets), the oracle decodes it, and rir_codec_decode_slots_device decodes it as it is."""
    import torch

    n, h, w = shape
    fr = _case_frames(name, shape)
    ctx = dev.CodecContext(w, h, n, gop)
    t = torch.from_numpy(fr).cuda()
    ctx.encode_tiles(t)
    dec = ctx.decode_slots()
    assert np.array_equal(dec.cpu().numpy(), fr)
    seg, slots = ctx.slots()
    seg = seg.cpu().numpy().view(np.uint32)
    slots = slots.cpu().numpy().view(np.uint64)
    hdr = ctx.hdr.cpu().numpy().view(np.uint64)
    L = ctx.layout
    for c in range(L.nchunks):
        f0 = c * gop
        nf = min(gop, n - f0)
        h_o, o_o, st_o = oracle.codec_encode_chunk(fr[f0:f0 + nf])
        assert np.array_equal(hdr[c][:, :nf], h_o), (name, c)
        assert np.array_equal(seg[c], np.diff(o_o)), (name, c)
        dense = np.concatenate([slots[c, t_, :seg[c, t_]] for t_ in range(L.ntiles)])
        assert np.array_equal(dense, st_o), (name, c)
        assert np.array_equal(oracle.codec_decode_chunk(hdr[c][:, :nf], o_o, dense, w, h), fr[f0:f0 + nf])
    # and stage 2 makes the dense form out of exactly these slots
    enc = ctx.encode_compact()
    assert np.array_equal(ctx.decode(enc).cpu().numpy(), fr)
    assert ctx.slots_payload_bytes() == enc.total_words() * 8


def test_gpu_decodes_oracle_stream_scattered_into_slots(dev, oracle):
    """cross-read the other way: segments written by the CPU restatement, put into slots on the host, decode on the GPU"""
    import torch

    n, h, w, gop = 9, 40, 52, 4
    fr = s1_noisy_background(n, h, w, seed=3)
    ctx = dev.CodecContext(w, h, n, gop)
    L = ctx.layout
    seg_t, slots_t = ctx.slots()
    seg = np.zeros(tuple(seg_t.shape), np.uint32)
    slots = np.full(tuple(slots_t.shape), 0xDEADBEEFDEADBEEF, np.uint64)
    hdr = np.zeros((L.nchunks, L.ntiles, gop), np.uint64)
    for c in range(L.nchunks):
        nf = min(gop, n - c * gop)
        h_o, o_o, s_o = oracle.codec_encode_chunk(fr[c * gop:c * gop + nf])
        hdr[c][:, :nf] = h_o
        seg[c] = np.diff(o_o)
        for t_ in range(L.ntiles):
            slots[c, t_, :seg[c, t_]] = s_o[o_o[t_]:o_o[t_ + 1]]
    ctx.hdr.copy_(torch.from_numpy(hdr.view(np.int64)))
    seg_t.copy_(torch.from_numpy(seg.view(np.int32)))
    slots_t.copy_(torch.from_numpy(slots.view(np.int64)))
    assert np.array_equal(ctx.decode_slots().cpu().numpy(), fr)
    # a length that does not match the headers is refused
    seg_t[1, 0] += 1
    with pytest.raises(RuntimeError):
        ctx.decode_slots()


def test_slotted_identity_at_baseline_size(dev):
    """configs[1] at full size through the two-launch path (encode_tiles + decode_slots), and the dense form made from the
    same slots decodes to the same frames"""
    import torch

    n, h, w = 1000, 512, 640
    fr = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
    ctx = dev.CodecContext(w, h, n, 50)
    ctx.encode_tiles(fr)
    out = ctx.decode_slots()
    assert torch.equal(out.view(torch.int16), fr.view(torch.int16))
    enc = ctx.encode_compact()
    assert ctx.slots_payload_bytes() == enc.total_words() * 8
    assert fr.numel() * 2 / enc.compressed_bytes() > 4.85
    out.zero_()
    assert torch.equal(ctx.decode(enc, out=out).view(torch.int16), fr.view(torch.int16))


# ---- the packed form: the dense stream without the order (rir_codec_encode_packed_device) ---------------------------------
def _packed_segments(batch):
    pos = batch.seg_pos.cpu().numpy().view(np.uint64)
    seg = batch.seg_words.cpu().numpy().view(np.uint32)
    st = batch.stream.cpu().numpy().view(np.uint64)
    return pos, seg, st


@pytest.mark.parametrize("name,shape,gop", CASES)
def test_packed_form_equals_oracle_segment_by_segment_and_has_no_holes(dev, oracle, name, shape, gop):
    """One kernel, and the encoded batch is exactly its payload: every (chunk, tile) segment holds the oracle's words, the
    segments tile [0, words) without gap or overlap (in whatever order they arrived), the oracle decodes them, the GPU
    decodes the batch as it is."""
    import torch

    n, h, w = shape
    fr = _case_frames(name, shape)
    pc = dev.PackedCodec(w, h, n, gop, stream_bytes="max", workspace_bytes="max")
    t = torch.from_numpy(fr).cuda()
    batch = pc.encode(t, check=True)
    assert np.array_equal(pc.decode(batch).cpu().numpy(), fr)
    pos, seg, st = _packed_segments(batch)
    hdr = batch.hdr.cpu().numpy().view(np.uint64)
    assert int(seg.astype(np.int64).sum()) == batch.words == batch.low + batch.high <= st.size
    order = np.argsort(pos.ravel(), kind="stable")
    p, s = pos.ravel()[order].astype(np.int64), seg.ravel()[order].astype(np.int64)
    keep = s > 0  # (empty segments may share a position)
    p, s = p[keep], s[keep]
    nlow = int(np.searchsorted(p, batch.low))  # segments of the low extent [0, low), the rest fill [capacity - high, capacity)
    assert np.array_equal(p[:nlow], np.concatenate([[0], np.cumsum(s[:nlow])])[:nlow]) and int(s[:nlow].sum()) == batch.low, "holes in the low extent"
    assert np.array_equal(p[nlow:], st.size - batch.high + np.concatenate([[0], np.cumsum(s[nlow:])])[:p.size - nlow]), "holes in the high extent"
    for c in range(pc.P.nchunks):
        f0 = c * gop
        nf = min(gop, n - f0)
        h_o, o_o, st_o = oracle.codec_encode_chunk(fr[f0:f0 + nf])
        assert np.array_equal(hdr[c][:, :nf], h_o), (name, c)
        assert np.array_equal(seg[c], np.diff(o_o)), (name, c)
        dense = np.concatenate([st[int(pos[c, t_]):int(pos[c, t_]) + int(seg[c, t_])] for t_ in range(pc.P.ntiles)])
        assert np.array_equal(dense, st_o), (name, c)
        assert np.array_equal(oracle.codec_decode_chunk(hdr[c][:, :nf], o_o, dense, w, h), fr[f0:f0 + nf])


def test_gpu_decodes_an_oracle_stream_laid_out_as_a_packed_batch(dev, oracle):
    """cross-read the other way: the CPU restatement's segments, shuffled into a packed batch on the host, decode on the GPU;
    a position or a length that points outside the stream is refused before anything is read through it"""
    import torch

    n, h, w, gop = 9, 40, 52, 4
    fr = s1_noisy_background(n, h, w, seed=3)
    pc = dev.PackedCodec(w, h, n, gop, stream_bytes="max")
    P = pc.P
    hdr = np.zeros((P.nchunks, P.ntiles, gop), np.uint64)
    segs = {}
    for c in range(P.nchunks):
        nf = min(gop, n - c * gop)
        h_o, o_o, s_o = oracle.codec_encode_chunk(fr[c * gop:c * gop + nf])
        hdr[c][:, :nf] = h_o
        for t_ in range(P.ntiles):
            segs[(c, t_)] = s_o[o_o[t_]:o_o[t_ + 1]]
    keys = list(segs)
    np.random.default_rng(5).shuffle(keys)
    pos = np.zeros((P.nchunks, P.ntiles), np.uint64)
    seg = np.zeros((P.nchunks, P.ntiles), np.uint32)
    parts, at = [], 0
    for k in keys:
        pos[k], seg[k] = at, segs[k].size
        parts.append(segs[k])
        at += segs[k].size
    st = np.concatenate(parts)
    batch = dev.PackedBatch(pc, torch.from_numpy(hdr.view(np.int64)).cuda(), torch.from_numpy(pos.view(np.int64)).cuda(),
                            torch.from_numpy(seg.view(np.int32)).cuda(), torch.from_numpy(st.view(np.int64)).cuda(), st.size, 0)
    assert np.array_equal(pc.decode(batch).cpu().numpy(), fr)
    for bad_pos, bad_len in ((st.size - 1, None), (2 ** 63, None), (None, 1)):
        b2 = dev.PackedBatch(pc, batch.hdr, batch.seg_pos.clone(), batch.seg_words.clone(), batch.stream, st.size, 0)
        if bad_pos is not None:
            b2.seg_pos.view(-1)[3] = np.array(bad_pos, np.uint64).view(np.int64).item()
        else:
            b2.seg_words.view(-1)[3] += bad_len
        with pytest.raises(RuntimeError):
            pc.decode(b2)


def test_packed_batch_that_exceeds_its_budget_says_so_and_fits_after_grow(dev):
    """incompressible frames against the 8 bit-per-pixel budget and the minimal arena: nothing is written out of bounds (canaries), the
    status names what the batch needs, and with room for any data the same call succeeds"""
    import torch

    n, h, w, gop = 12, 67, 83, 5
    rng = np.random.default_rng(11)
    fr = rng.integers(0, 65536, (n, h, w)).astype(np.uint16)
    t = torch.from_numpy(fr).cuda()
    pc = dev.PackedCodec(w, h, n, gop)
    cap = pc.stream.numel()
    big = torch.full((cap + 4096,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda")
    pc.stream = big[:cap]
    wcap = pc.workspace.numel()
    wbig = torch.full((wcap + 32768,), 0xA5, dtype=torch.uint8, device="cuda")
    pc.workspace = wbig[:wcap]
    pc.encode(t)
    code, low, high, arena = pc.status()
    words = low + high
    assert code != 0 and words * 8 > cap * 8 and words * 8 >= fr.nbytes * 0.95
    with pytest.raises(RuntimeError, match="does not fit"):
        pc.finish()
    assert bool((big[cap:] == 0x5A5A5A5A5A5A5A5A).all()) and bool((wbig[wcap:] == 0xA5).all()), "wrote past the capacity it was given"
    pc.grow()
    batch = pc.encode(t, check=True)
    assert batch.words == words
    assert np.array_equal(pc.decode(batch).cpu().numpy(), fr)


def test_packed_identity_and_footprint_at_baseline_size(dev):
    """configs[1] at full size: two launches, the encoded batch fits the 8 bpp budget (it is less than a quarter of the raw bytes), it
    decodes to the input, and it has as many payload words as the dense (file) form"""
    import torch

    n, h, w = 1000, 512, 640
    fr = torch.from_numpy(s1_noisy_background(n, h, w)).cuda()
    pc = dev.PackedCodec(w, h, n, 50)
    batch = pc.encode(fr, check=True)
    assert pc.stream.numel() * 8 <= fr.numel() * 2 // 2 + 256
    assert batch.nbytes() < fr.numel() * 2 / 4.7
    out = pc.decode(batch)
    assert torch.equal(out.view(torch.int16), fr.view(torch.int16))
    ctx = dev.CodecContext(w, h, n, 50)
    enc = ctx.encode(fr)
    assert enc.total_words() == batch.words
    assert torch.equal(ctx.hdr, batch.hdr)
