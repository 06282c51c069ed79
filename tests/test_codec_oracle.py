"""CPU: the codec half of the oracle (format RIRB1).

The reference codec is libx264 through ffmpeg (unbuildable here, bitstream unpinned); the contract
the reference's own tests hold for it is the lossless identity (reference
tests/python/test_IRMovie.py:40-49,60-99, tests/python/test_rir.py:317-332) - asserted here on
the reference fixtures' recipes - plus structural properties of this build's format."""
import numpy as np
import pytest

from librir_amd.synthetic import s1_noisy_background, s2_uniform_dl_ti


def roundtrip(oracle, frames):
    hdr, off, stream = oracle.codec_encode_chunk(frames)
    n, h, w = frames.shape
    dec = oracle.codec_decode_chunk(hdr, off, stream, w, h)
    assert dec.dtype == np.uint16 and np.array_equal(dec, frames)
    return hdr, off, stream


def widths(hdr):
    """(ntiles, n, 8) bit-plane counts out of the header table"""
    h = hdr.astype(np.uint64)
    return np.stack([(h >> np.uint64(16 * (j & 3) + (0 if j < 4 else 5))) & np.uint64(31) for j in range(8)], axis=-1).astype(np.int64)


def modes(hdr):
    return ((hdr.astype(np.uint64) >> np.uint64(14)) & np.uint64(3)).astype(np.int64)


def bases(hdr):
    h = hdr.astype(np.uint64)
    return sum((((h >> np.uint64(16 * q + 10)) & np.uint64(15)) << np.uint64(4 * q)) for q in range(4)).astype(np.int64)


@pytest.mark.parametrize("shape", [(1, 512, 640), (10, 240, 320), (10, 256, 320), (3, 20, 20), (2, 1, 1), (4, 7, 9), (5, 67, 83), (1, 3, 1025)])
def test_identity_random(oracle, shape):
    rng = np.random.default_rng(sum(shape))
    roundtrip(oracle, rng.integers(0, 65536, shape).astype(np.uint16))


def test_identity_reference_fixture_recipes(oracle):
    # conftest.images recipe (noisy background) and the uniform DL+TI recipe
    hdr, off, stream = roundtrip(oracle, s1_noisy_background(12, 128, 160))
    assert stream.nbytes < 12 * 128 * 160 * 2 / 2.5  # temporal delta pays off
    hdr, off, stream = roundtrip(oracle, s2_uniform_dl_ti(10, 64, 80))
    assert stream.size == 0  # constant frames: every residual equals the tile base
    assert (widths(hdr) == 0).all() and (modes(hdr)[:, 1:] == 1).all()


def test_extremes(oracle):
    f = np.zeros((3, 16, 64), np.uint16)
    hdr, off, stream = roundtrip(oracle, f)
    assert stream.size == 0 and (hdr[:, 0] == 0).all()  # all-zero key frame = all-zero header
    f[:] = 65535
    hdr, off, stream = roundtrip(oracle, f)
    assert stream.size == 0
    # worst case: full-range noise -> every plane of every slot, 128 words per record
    g = np.random.default_rng(3).integers(0, 65536, (4, 16, 64)).astype(np.uint16)
    hdr, off, stream = roundtrip(oracle, g)
    assert widths(hdr).max() == 16 and widths(hdr).sum(axis=-1).max() <= 128 and off[-1] == stream.size
    # a frame-to-frame step of +-32768 wraps mod 2^16
    f = np.zeros((4, 16, 64), np.uint16)
    f[1::2] = 0x8000
    f[:, 3, 5] = 0x7FFF
    roundtrip(oracle, f)


def test_left_mode_selected_on_ramps(oracle):
    ramp = (np.arange(64 * 64, dtype=np.uint32) * 3 % 65536).astype(np.uint16).reshape(1, 64, 64)
    hdr, off, stream = roundtrip(oracle, ramp)
    assert (modes(hdr)[:, 0] == 2).all()  # MODE_LEFT
    assert stream.size <= 8 * 16  # only each tile's first pixel deviates from the constant slope: one slot per tile pays
    noise = np.random.default_rng(0).integers(0, 1024, (1, 64, 64)).astype(np.uint16)
    hdr, off, stream = roundtrip(oracle, noise)
    assert (modes(hdr)[:, 0] == 0).all()  # MODE_RAW


def test_header_layout(oracle):
    f = np.full((2, 8, 64), 1000, np.uint16)
    f[1] += 7
    f[1, 0, 0] += 5  # one pixel of slot 0 deviates by 5 -> 3 planes in slot 0 only
    hdr, off, stream = roundtrip(oracle, f)
    assert bases(hdr)[0].tolist() == [1000, 7] and modes(hdr)[0].tolist() == [0, 1]
    assert widths(hdr)[0, 0].tolist() == [0] * 8 and widths(hdr)[0, 1].tolist() == [3, 0, 0, 0, 0, 0, 0, 0]
    assert int(hdr[0, 1]) == 3 | (7 << 10) | (1 << 14)
    assert stream.tolist() == [1, 0, 1]  # bit-planes of the value 5 in lane 0


def test_tiles_are_independent(oracle):
    """Each 512-pixel tile decodes from its own segment: changing one tile leaves the others' bytes unchanged."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 4096, (3, 8, 256)).astype(np.uint16)  # 4 tiles
    b = a.copy()
    b[:, 2, :] += 7  # rows 2,3 form tile 1
    sa, oa, sta = oracle.codec_encode_chunk(a)
    sb, ob, stb = oracle.codec_encode_chunk(b)
    for t in (0, 2, 3):
        assert np.array_equal(sta[oa[t]:oa[t + 1]], stb[ob[t]:ob[t + 1]])


def test_malformed_stream_is_rejected(oracle):
    f = np.random.default_rng(1).integers(0, 65536, (2, 8, 64)).astype(np.uint16)
    hdr, off, stream = oracle.codec_encode_chunk(f)
    bad = hdr.copy()
    bad[0, 0] = np.uint64(int(bad[0, 0]) | 0x1F)  # width 31 > 16
    with pytest.raises(RuntimeError):
        oracle.codec_decode_chunk(bad, off, stream, 64, 8)
    bad = hdr.copy()
    bad[0, 1] = np.uint64(int(bad[0, 1]) | (1 << 31))  # reserved bit
    with pytest.raises(RuntimeError):
        oracle.codec_decode_chunk(bad, off, stream, 64, 8)
    short = off.copy()
    short[-1] -= 1  # table says the last segment is one word shorter than its headers need
    with pytest.raises(RuntimeError):
        oracle.codec_decode_chunk(hdr, short, stream, 64, 8)


def test_format_is_frozen(oracle):
    """tests/golden/codec_format.json pins the RIRB1 bitstream across rounds (not a reference output: the format is
    this build's own): the oracle must reproduce the stored hashes and the verbatim tiny stream."""
    import json
    import os
    import sys

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    import make_codec_golden as G

    fix = json.load(open(os.path.join(here, "codec_format.json")))
    for name, fr in G.streams():
        hdr, off, st = oracle.codec_encode_chunk(fr)
        c = fix["cases"][name]
        assert (G.digest(hdr), G.digest(off), G.digest(st), int(st.size)) == (c["hdr"], c["tile_off"], c["stream"], c["words"]), name
    t = fix["tiny_2x1x4"]
    hdr, off, st = oracle.codec_encode_chunk(np.array(t["frames"], np.uint16))
    assert [int(x) for x in hdr.ravel()] == t["hdr"] and [int(x) for x in off.ravel()] == t["tile_off"] and [int(x) for x in st.ravel()] == t["stream"]
    dec = oracle.codec_decode_chunk(np.array(t["hdr"], np.uint64).reshape(hdr.shape), np.array(t["tile_off"], np.uint32), np.array(t["stream"], np.uint64), 4, 1)
    assert dec.tolist() == t["frames"]
