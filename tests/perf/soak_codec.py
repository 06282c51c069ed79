"""One-off soak: random geometries and data distributions, GPU encoder / decoder against the oracle (not collected by
pytest).   python tests/perf/soak_codec.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librir_amd import device as D
from oracle.pyoracle import Oracle
O = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
refused = 0
for k in range(cases):
    h, w = int(rng.integers(1, 97)), int(rng.integers(1, 161))
    if rng.random() < 0.3:
        w = int(rng.choice([64, 128, 320, 512, 640])); h = int(rng.choice([8, 16, 60, 64]))
    n = int(rng.integers(1, 70)); gop = int(rng.choice([1, 2, 3, 7, 16, 50, 64, 65, 128]))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        fr = rng.integers(0, 65536, (n, h, w))
    elif kind == 1:
        fr = rng.integers(0, 16, (n, h, w)) + 1000
    elif kind == 2:
        fr = (rng.random((h, w)) * 4000)[None] + np.arange(n)[:, None, None] * rng.integers(-3, 4) + rng.normal(0, rng.choice([0.3, 1, 5, 40]), (n, h, w))
    elif kind == 3:
        fr = np.full((n, h, w), int(rng.integers(0, 65536)))
    elif kind == 4:
        fr = np.cumsum(rng.integers(-2, 3, (n, h, w)), axis=2) + 30000
    else:
        fr = rng.integers(0, 2, (n, h, w)) * 65535
    fr = np.clip(fr, 0, 65535).astype(np.uint16)
    ctx = D.CodecContext(w, h, n, gop)
    t = torch.from_numpy(fr).cuda()
    enc = ctx.encode(t); dec = ctx.decode(enc); torch.cuda.synchronize()
    ok = np.array_equal(dec.cpu().numpy(), fr)
    # the slotted form (stage 1 alone) straight into the decoder, and its lengths against the dense offsets
    ctx.encode_tiles(t)
    ok = ok and np.array_equal(ctx.decode_slots().cpu().numpy(), fr)
    seg = ctx.slots()[0].cpu().numpy().view(np.uint32)
    ok = ok and np.array_equal(seg, np.diff(enc.tile_off.cpu().numpy().view(np.uint32).astype(np.int64), axis=1).astype(np.uint32))
    hdr = enc.hdr.cpu().numpy().view(np.uint64); toff = enc.tile_off.cpu().numpy().view(np.uint32); coff = enc.chunk_off.cpu().numpy(); st = enc.stream.cpu().numpy().view(np.uint64)
    same = True
    for c in range(ctx.layout.nchunks):
        f0 = c * gop; nf = min(gop, n - f0)
        h_o, o_o, st_o = O.codec_encode_chunk(fr[f0:f0 + nf])
        same &= bool(np.array_equal(hdr[c][:, :nf], h_o) and np.array_equal(toff[c], o_o) and np.array_equal(st[coff[c]:coff[c + 1]], st_o))
    # the packed form: with room for any data - every segment the dense stream's words, the extents without holes - and, for a budget-sized
    # buffer with the minimal arena, either a complete batch that decodes or a refusal (never a wrong frame)
    pc = D.PackedCodec(w, h, n, gop, stream_bytes="max", workspace_bytes="max")
    pb = pc.encode(t, check=True)
    ok = ok and np.array_equal(pc.decode(pb).cpu().numpy(), fr)
    pos = pb.seg_pos.cpu().numpy().view(np.uint64); sw = pb.seg_words.cpu().numpy().view(np.uint32); pst = pb.stream.cpu().numpy().view(np.uint64)
    same = same and np.array_equal(sw, seg) and int(sw.astype(np.int64).sum()) == pb.low + pb.high
    for c in range(ctx.layout.nchunks):
        for tl in range(ctx.layout.ntiles):
            a0 = int(coff[c]) + int(toff[c][tl])
            same = same and bool(np.array_equal(pst[int(pos[c, tl]):int(pos[c, tl]) + int(sw[c, tl])], st[a0:a0 + int(sw[c, tl])]))
    pc2 = D.PackedCodec(w, h, n, gop)
    pc2.encode(t)
    code = pc2.status()[0]
    if code == 0:
        ok = ok and np.array_equal(pc2.decode().cpu().numpy(), fr)
    else:
        refused += 1
    if not (ok and same):
        bad += 1
        print("FAIL case", k, (n, h, w, gop, kind), "roundtrip", ok, "stream==oracle", same)
print("soak: %d cases, %d failures (packed form within the 8 bpp budget and the minimal arena: %d batches refused, the rest decoded)" % (cases, bad, refused))
sys.exit(1 if bad else 0)
