"""CPU: host-side logic of the video_io / tools boundary that needs no GPU - the attribute trailer
(byte-compatible with the reference's, cross-read with the compiled reference where oracle/_ref has
it), raw PCR reading, handle and error conventions, and the loud failure of the codec paths
without a device."""
import os

import numpy as np
import pytest

from librir_amd.tools import FileAttributes, zstd_compress, zstd_decompress
from librir_amd.video_io import rir_video_io as rv
from librir_amd.video_io.IRMovie import create_pcr_header


def write_pcr(path, frames, frequency=50):
    n, h, w = frames.shape
    hdr = create_pcr_header(h, w, frequency)
    with open(path, "wb") as f:
        f.write(hdr.astype(np.uint32).tobytes())
        f.write(frames.astype(np.uint16).tobytes())


def test_trailer_roundtrip_and_resize(tmp_path):
    p = tmp_path / "video.bin"
    p.write_bytes(b"payload" * 10)
    fa = FileAttributes.from_filename(p)
    fa.attributes = {"Name": "WA", "big": b"\x07" * 5000, "empty": ""}
    fa.timestamps = [10, 20, 30]
    fa.set_frame_attributes(0, {"a": "1"})
    fa.set_frame_attributes(2, {"b": b"\x00\xff", "c": "x" * 2000})
    fa.close()
    size1 = os.path.getsize(p)
    assert p.read_bytes().startswith(b"payload" * 10) and p.read_bytes().endswith(b"H264ATTRIBUTES")
    fa = FileAttributes.from_filename(p)
    assert fa.attributes == {"Name": b"WA", "big": b"\x07" * 5000, "empty": b""}
    assert list(fa.timestamps) == [10, 20, 30] and fa.frame_count() == 3
    assert fa.frame_attributes(0) == {"a": b"1"} and fa.frame_attributes(1) == {} and fa.frame_attributes(2)["b"] == b"\x00\xff"
    fa.attributes = {"Name": "WA"}  # a second object rewrites a SHORTER trailer: the file is truncated
    fa.close()
    assert os.path.getsize(p) < size1 and p.read_bytes().startswith(b"payload" * 10)
    fa = FileAttributes.from_filename(p)
    assert fa.attributes == {"Name": b"WA"} and list(fa.timestamps) == [10, 20, 30]
    fa.discard()


def test_trailer_layout_is_the_reference_layout(tmp_path):
    """map(global) map(frame)* int64 ts[N] u64 N u64 size 'H264ATTRIBUTES' - FileAttributes.cpp:455-485"""
    p = tmp_path / "t.bin"
    p.write_bytes(b"")
    fa = FileAttributes.from_filename(p)
    fa.attributes = {"k": "vv"}
    fa.timestamps = [7]
    fa.close()
    raw = p.read_bytes()
    u64 = lambda v: int(v).to_bytes(8, "little")
    expected = u64(1) + u64(1) + b"k" + u64(2) + b"vv" + u64(0) + u64(7) + u64(1)
    expected += u64(len(expected) + 8 + 14) + b"H264ATTRIBUTES"
    assert raw == expected


def test_trailer_cross_read_with_reference(tmp_path, ref):
    from oracle.pyoracle import RefAttrs

    try:
        RA = RefAttrs(ref)
    except FileNotFoundError:
        pytest.skip("oracle/_ref built without FileAttributes")
    # reference writes, we read
    p = tmp_path / "ref.bin"
    p.write_bytes(b"abc")
    RA.write(p, {"Device": "cam", "blob": b"\x01" * 4000}, [5, 6, 7, 8], "idx", [b"0", b"1", b"22", b"333"])
    fa = FileAttributes.from_filename(p)
    assert fa.attributes == {"Device": b"cam", "blob": b"\x01" * 4000}
    assert list(fa.timestamps) == [5, 6, 7, 8] and fa.frame_attributes(3) == {"idx": b"333"}
    fa.discard()
    # we write, the reference reads
    q = tmp_path / "ours.bin"
    q.write_bytes(b"xyz")
    fa = FileAttributes.from_filename(q)
    fa.attributes = {"GOP": "50", "blob": b"\x02" * 3000}
    fa.timestamps = [100, 200]
    fa.close()
    cnt, times, val, ng = RA.read(q, "blob")
    assert cnt == 2 and times == [100, 200] and val == b"\x02" * 3000 and ng == 2
    # uncompressed trailers are byte-identical
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    a.write_bytes(b"")
    b.write_bytes(b"")
    RA.write(a, {"k1": "v1", "k2": "v2"}, [1, 2, 3], "f", [b"x", b"y", b"z"])
    fa = FileAttributes.from_filename(b)
    fa.attributes = {"k1": "v1", "k2": "v2"}
    fa.timestamps = [1, 2, 3]
    for i, v in enumerate([b"x", b"y", b"z"]):
        fa.set_frame_attributes(i, {"f": v})
    fa.close()
    assert a.read_bytes() == b.read_bytes()


def test_zstd_wrappers():
    data = np.random.default_rng(0).integers(0, 4, 10000).astype(np.uint8).tobytes()
    c = zstd_compress(data, 3)
    assert len(c) < len(data) and zstd_decompress(c) == data
    with pytest.raises(RuntimeError):  # reference tests/python/test_rir.py:47-74
        zstd_decompress(b"not a zstd frame")


def test_pcr_reader_needs_no_gpu(tmp_path):
    rng = np.random.default_rng(1)
    fr = rng.integers(0, 16000, (4, 20, 30)).astype(np.uint16)
    p = tmp_path / "m.pcr"
    write_pcr(p, fr, frequency=25)
    assert rv.video_file_format(p) == rv.FILE_FORMAT_PCR
    cam = rv.open_camera_file(p)
    assert rv.get_image_count(cam) == 4 and rv.get_image_size(cam) == (20, 30)
    for i in range(4):
        assert np.array_equal(rv.load_image(cam, i), fr[i])
    # timestamps synthesised at 1e9/Frequency ns when the frames carry none (IRFileLoader.cpp:421-431)
    assert [rv.get_image_time(cam, i) for i in range(4)] == [0, 40000000, 80000000, 120000000]
    assert rv.supported_calibrations(cam) == ["Digital Level"]
    assert rv.get_filename(cam) == str(p)
    assert rv.get_last_image_raw_value(cam, 3, 2) == fr[3, 2, 3]
    with pytest.raises(RuntimeError):
        rv.load_image(cam, 4)
    with pytest.raises(RuntimeError):
        rv.load_image(cam, -1)
    with pytest.raises(RuntimeError):
        rv.load_image(cam, 0, 1)  # no temperature calibration without a plugin
    rv.close_camera(cam)
    assert rv.get_image_count(cam) == -1  # handle released
    with pytest.raises(RuntimeError):
        rv.open_camera_file(tmp_path / "missing.bin")
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"\x00" * 5000)
    with pytest.raises(RuntimeError):
        rv.open_camera_file(junk)
    with pytest.raises(RuntimeError):  # reference rir_video_io.py:115-117, tests/python/test_rir.py:336-339
        rv.video_file_format(junk)
    assert rv.video_file_format(p) is rv.FileFormat.PCR  # (reference tests/python/test_video_io.py:165-166)


def test_the_other_raw_formats_need_no_gpu(tmp_path):
    """BIN (WEST) and PCR inside a BIN envelope: the two other raw layouts the reference loader reads with the code it reads PCR files
    with (IRFileLoader.cpp:166-209 detection, :404-451 geometry and timestamps)"""
    import struct

    rng = np.random.default_rng(2)
    fr = rng.integers(0, 16000, (5, 24, 36)).astype(np.uint16)
    # BIN / WEST: a 128-byte header (version, triggers = 1, compression = 0), a 128-byte trigger block of int64 fields
    # (date, rate, samples, ..., data_size_x, data_size_y), then the frames
    west = tmp_path / "w.bin"
    head = bytes([1, 1, 0]) + bytes(125)
    trig = struct.pack("<11q", 123456789, 100, 5, 0, 0, 1, 0, 0, 0, 36, 24) + bytes(128 - 88)
    west.write_bytes(head + trig + fr.tobytes())
    assert rv.video_file_format(west) == rv.FILE_FORMAT_WEST
    cam = rv.open_camera_file(west)
    assert rv.get_image_count(cam) == 5 and rv.get_image_size(cam) == (24, 36)
    for i in (4, 0, 2):
        assert np.array_equal(rv.load_image(cam, i), fr[i])
    assert [rv.get_image_time(cam, i) for i in range(5)] == [0, 10000000, 20000000, 30000000, 40000000]  # 1e9 / rate ns
    rv.close_camera(cam)
    # the same with frames that carry a rising 8-byte stamp in their last bytes: taken as ms, re-based to the first (IRFileLoader.cpp:255-282,433-451)
    stamped = fr.copy()
    for i in range(5):
        stamped[i].reshape(-1).view(np.int64)[-1] = 5000 + 20 * i
    west.write_bytes(head + trig + stamped.tobytes())
    cam = rv.open_camera_file(west)
    assert [rv.get_image_time(cam, i) for i in range(5)] == [0, 20000000, 40000000, 60000000, 80000000]
    assert np.array_equal(rv.load_image(cam, 3), stamped[3])
    rv.close_camera(cam)
    # PCR in a BIN envelope: 128 + 5 bytes, then a PCR header and its frames
    enc = tmp_path / "e.bin"
    enc.write_bytes(bytes([9, 0, 0]) + bytes(130) + create_pcr_header(24, 36, 25).astype(np.uint32).tobytes() + fr.tobytes())
    assert rv.video_file_format(enc) == rv.FILE_FORMAT_PCR_ENCAPSULATED
    cam = rv.open_camera_file(enc)
    assert rv.get_image_count(cam) == 5 and rv.get_image_size(cam) == (24, 36)
    for i in range(5):
        assert np.array_equal(rv.load_image(cam, i), fr[i])
    assert rv.get_image_time(cam, 2) == 80000000
    rv.close_camera(cam)
    from librir_amd.video_io import IRMovie

    with IRMovie.from_filename(enc) as mov:
        assert np.array_equal(mov.data, fr)
    # a header without a single whole frame behind it
    west.write_bytes(head + trig + fr.tobytes()[:100])
    with pytest.raises(RuntimeError):
        rv.open_camera_file(west)


def test_pcr_in_memory(tmp_path):
    fr = np.arange(2 * 6 * 8, dtype=np.uint16).reshape(2, 6, 8)
    data = create_pcr_header(6, 8).astype(np.uint32).tobytes() + fr.tobytes()
    cam = rv.open_camera_memory(data)
    assert np.array_equal(rv.load_image(cam, 1), fr[1])
    rv.close_camera(cam)


def test_saver_parameters_and_handles(tmp_path):
    s = rv.h264_open_file(tmp_path / "o.h264", 32, 16)
    for k in ["compressionLevel", "lowValueError", "highValueError", "codec", "GOP", "threads", "slices", "stdFactor", "inputCamera",
              "removeBadPixels", "subtractMin", "subtractLocalMin", "runningAverage"]:
        rv.h264_set_parameter(s, k, "1")
    with pytest.raises(RuntimeError):  # unknown key -> -1 (h264.cpp:1709-1781)
        rv.h264_set_parameter(s, "noSuchParameter", "1")
    rv.h264_set_global_attributes(s, {"a": "b"})
    assert len(rv.h264_get_low_errors(s)) == 0
    rv.h264_close_file(s)  # nothing was added: no file is created
    assert not os.path.exists(tmp_path / "o.h264")
    with pytest.raises(RuntimeError):
        rv.h264_set_parameter(s, "GOP", "5")  # the handle is gone


def test_codec_paths_fail_loudly_without_device(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from librir_amd.low_level.misc import last_error

    s = rv.h264_open_file(tmp_path / "o.h264", 32, 16)
    with pytest.raises(RuntimeError):
        rv.h264_add_image_lossless(s, np.zeros((16, 32), np.uint16), 0)
    assert "no usable HIP device" in last_error()
    rv.h264_close_file(s)
    assert not os.path.exists(tmp_path / "o.h264") or os.path.getsize(tmp_path / "o.h264") == 0


def _minimal_rirb(nframes=0, nchunks=0, width=8, height=4, gop=50, index_offset=96, index=b"", version=1):
    """A hand-built container: ftyp box (32 B) + file header (64 B) + chunk index."""
    import struct

    ftyp = struct.pack(">I", 32) + b"ftyp" + b"RIRB" + struct.pack(">I", 1) + b"RIRBisom" + b"\0" * 8
    hdr = b"RIRBLOCK" + struct.pack("<IIIIII", version, width, height, gop, 50, 0) + struct.pack("<QQQ", index_offset, nframes, nchunks) + b"\0" * 8
    assert len(ftyp) == 32 and len(hdr) == 64
    return ftyp + hdr + index


def test_hostile_container_headers_are_rejected_not_crashed_on(tmp_path):
    """Sizes read from a file are checked against the file before anything is allocated from them; nothing is thrown
    across the C boundary.  Runs without a GPU: opening only parses."""
    import struct

    from librir_amd.video_io import rir_video_io as rv

    def opens(blob):
        p = tmp_path / "f.bin"
        p.write_bytes(blob)
        try:
            cam = rv.open_camera_file(p)
        except RuntimeError:
            return None
        n = rv.get_image_count(cam)
        rv.close_camera(cam)
        return n

    assert opens(_minimal_rirb()) == 0  # a closed, empty recording is a valid file
    huge = 0xFFFFFFFFFFFFFFF
    for kw in (dict(nchunks=huge), dict(nframes=huge, nchunks=1), dict(width=0), dict(height=10 ** 6), dict(gop=0), dict(index_offset=10 ** 12),
               dict(nchunks=3), dict(nframes=10, nchunks=0), dict(version=7), dict(index_offset=0)):
        assert opens(_minimal_rirb(**kw)) is None, kw
    # a chunk index whose entries point outside the file / outside the frame range
    bad_index = struct.pack("<QQII", 10 ** 9, 0, 5, 0)
    assert opens(_minimal_rirb(nframes=5, nchunks=1, index=bad_index)) is None
    bad_index = struct.pack("<QQII", 96, 3, 50, 0)
    assert opens(_minimal_rirb(nframes=5, nchunks=1, index=bad_index)) is None
    # a trailer that announces more frames than it can hold
    trailer = struct.pack("<Q", 0) + struct.pack("<QQ", 2 ** 30, 16 + 14 + 8) + b"H264ATTRIBUTES"
    assert opens(_minimal_rirb() + trailer) == 0
    # random garbage and truncations
    rng = np.random.default_rng(0)
    base = bytearray(_minimal_rirb())
    for _ in range(200):
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        opens(bytes(b[: int(rng.integers(1, len(b) + 1))]))


# ---- ZFile container (reference ZFile.cpp: 128-byte header blocks, [int64 ts][u32 csize][zstd frame] records, --------
# ---- trailer attribute "positions"); host-side zstd, so none of this needs a GPU ----------------------------------
def parse_zfile(raw):
    """Independent reader of the layout stated in reference ZFile.cpp:18-46 (headers) and :483-542 (records)."""
    version, triggers, compression = raw[0], raw[1], raw[2]
    trig = np.frombuffer(raw[128:128 + 88], dtype="<u8")
    info = dict(version=version, triggers=triggers, compression=compression, rate=int(trig[1]), samples=int(trig[2]), type=int(trig[4]),
                nb_channels=int(trig[5]), data_format=int(trig[7]), w=int(trig[9]), h=int(trig[10]))
    frames, times, positions = [], [], []
    pos = 256
    for _ in range(info["samples"]):
        t = int(np.frombuffer(raw[pos:pos + 8], "<i8")[0])
        csize = int(np.frombuffer(raw[pos + 8:pos + 12], "<u4")[0])
        data = zstd_decompress(raw[pos + 12:pos + 12 + csize])
        frames.append(np.frombuffer(data, "<u2").reshape(info["h"], info["w"]))
        times.append(t)
        positions.append(pos)
        pos += 12 + csize
    return info, np.array(frames), times, positions, pos


def build_zfile(frames, times, rate=50, samples=None, level=1):
    n, h, w = frames.shape
    head = bytearray(256)
    head[0:3] = bytes([1, 1, 1])
    trig = np.zeros(11, "<u8")
    trig[1], trig[2], trig[4], trig[5], trig[7], trig[8], trig[9], trig[10] = rate, n if samples is None else samples, 1, 1, 3, 1, w, h
    head[128:128 + 88] = trig.tobytes()
    body = b""
    for f, t in zip(frames, times):
        c = zstd_compress(f.tobytes(), level)
        body += np.int64(t).tobytes() + np.uint32(len(c)).tobytes() + c
    return bytes(head) + body


def test_zfile_written_here_has_the_reference_layout(tmp_path):
    rng = np.random.default_rng(5)
    fr = (rng.random((5, 24, 40)) * 900 + np.arange(5)[:, None, None]).astype(np.uint16)
    ts = [2_000_000_000 + 20_000_000 * i for i in range(5)]  # ns, beyond the "milliseconds" window of the loader
    p = tmp_path / "m.bin"
    w = rv.open_video_write(p, 40, 24, rate=50, method=rv.METHOD_ZSTD, clevel=3)
    for f, t in zip(fr, ts):
        rv.image_write(w, f, t)
    data_size = rv.close_video(w)
    raw = p.read_bytes()
    info, frames, times, positions, end = parse_zfile(raw)
    assert info == dict(version=1, triggers=1, compression=1, rate=50, samples=5, type=1, nb_channels=1, data_format=3, w=40, h=24)
    assert np.array_equal(frames, fr) and times == ts and end == data_size
    # trailer: timestamps + global attribute "positions" = the record offsets as int64 (ZFile.cpp:434-447)
    a = FileAttributes.from_filename(p)
    assert list(a.timestamps) == ts
    assert np.frombuffer(a.attributes["positions"], "<i8").tolist() == positions
    a.discard()
    # and the loader of this library reads it back, as format 4
    assert rv.video_file_format(p) == rv.FILE_FORMAT_ZSTD_COMPRESSED
    cam = rv.open_camera_file(p)
    assert rv.get_image_count(cam) == 5 and rv.get_image_size(cam) == (24, 40)
    for i in (3, 0, 4, 1, 2):
        assert np.array_equal(rv.load_image(cam, i), fr[i])
        assert rv.get_image_time(cam, i) == ts[i]
    assert rv.get_last_image_raw_value(cam, 7, 5) == fr[2, 5, 7]
    with pytest.raises(RuntimeError):
        rv.load_image(cam, 5)
    rv.close_camera(cam)
    with pytest.raises(RuntimeError):  # the ZFile readers accept sizes below 3000 only
        rv.open_video_write(tmp_path / "big.bin", 3000, 10, method=rv.METHOD_ZSTD)


def test_zfile_cross_read_with_reference(tmp_path):
    """The ZFile container against the reference's own reader and writer (its unmodified ZFile.cpp, compiled by oracle/build_ref.sh with
    zstd_* resolved from this build's libtools.so: reference video_io code on this build's `tools`): a file written by
    open_video_write(method = zstd) is read by z_open_file_read / z_read_image / z_get_timestamps (ZFile.cpp:273-330,544-629), and a file
    written by z_open_file_write / z_write_image (:332-372,483-542) is read by open_camera_file, with the timestamps as the reference's
    loader presents them: (t - t0) x 10^6 for small values, IRFileLoader.cpp:345-372.  Skipped where /root/reference is absent."""
    from oracle.pyoracle import RefZFile

    try:
        RZ = RefZFile()
    except FileNotFoundError:
        pytest.skip("oracle/_ref/librir_ref_zfile.so not built here (needs /root/reference and zstd.h)")
    rng = np.random.default_rng(15)
    n, h, w = 7, 24, 40
    fr = (rng.random((n, h, w)) * 900 + 10 * np.arange(n)[:, None, None]).astype(np.uint16)
    # this build writes, the reference reads
    ts = [2_000_000_000 + 20_000_000 * i for i in range(n)]
    p = tmp_path / "ours.bin"
    wr = rv.open_video_write(p, w, h, rate=50, method=rv.METHOD_ZSTD, clevel=3)
    for f, t in zip(fr, ts):
        rv.image_write(wr, f, t)
    rv.close_video(wr)
    cnt, got, times = RZ.read(p, n + 2, (h, w))
    assert cnt == n and np.array_equal(got, fr) and times == ts
    # the reference writes, this build reads (format 4); small timestamps: milliseconds from the first image, presented as nanoseconds
    ts_ms = [1000 + 40 * i for i in range(n)]
    q = tmp_path / "ref.bin"
    size = RZ.write(q, fr, ts_ms, rate=25, method=1, clevel=2)
    assert 0 < size <= os.path.getsize(q)
    assert rv.video_file_format(q) == rv.FILE_FORMAT_ZSTD_COMPRESSED
    cam = rv.open_camera_file(q)
    assert rv.get_image_count(cam) == n and rv.get_image_size(cam) == (h, w)
    for i in (3, 0, 6, 1):
        assert np.array_equal(rv.load_image(cam, i), fr[i])
    assert [rv.get_image_time(cam, i) for i in range(n)] == [(t - ts_ms[0]) * 1_000_000 for t in ts_ms]
    rv.close_camera(cam)
    # and the reference reads its own file the same way this build's independent parser does (the layout tests above rest on that parser)
    cnt, got, times = RZ.read(q, n, (h, w))
    assert cnt == n and np.array_equal(got, fr) and times == ts_ms
    info, frames, ptimes, _, _ = parse_zfile(q.read_bytes())
    assert (info["w"], info["h"], info["rate"], info["samples"]) == (w, h, 25, n) and np.array_equal(frames, fr) and ptimes == ts_ms


@pytest.mark.parametrize("samples", [None, 0])
def test_zfile_without_trailer_is_walked_record_by_record(tmp_path, samples):
    """Files from a writer that stored no trailer (or no image count): the index comes from walking the records
    (ZFile.cpp:196-245).  Timestamps in milliseconds are rebased to 0 and converted to ns (IRFileLoader.cpp:355-376)."""
    fr = np.random.default_rng(6).integers(0, 16000, (4, 10, 12)).astype(np.uint16)
    raw = build_zfile(fr, [100, 120, 140, 160], rate=25, samples=samples)
    p = tmp_path / "z.bin"
    p.write_bytes(raw)
    for cam in (rv.open_camera_file(p), rv.open_camera_memory(raw)):
        assert rv.get_image_count(cam) == 4
        assert [rv.get_image_time(cam, i) for i in range(4)] == [0, 20_000_000, 40_000_000, 60_000_000]
        for i in range(4):
            assert np.array_equal(rv.load_image(cam, i), fr[i])
        rv.close_camera(cam)


def test_zfile_damaged_files_fail_cleanly(tmp_path):
    fr = np.random.default_rng(7).integers(0, 16000, (3, 10, 12)).astype(np.uint16)
    raw = bytearray(build_zfile(fr, [0, 1, 2]))
    blosc = bytes(raw[:2]) + b"\x02" + bytes(raw[3:])
    with pytest.raises(RuntimeError):  # blosc methods are not readable here
        rv.open_camera_memory(blosc)
    cut = bytes(raw[:len(raw) - 7])  # last record truncated: the first two images stay readable
    cam = rv.open_camera_memory(cut)
    assert rv.get_image_count(cam) == 2 and np.array_equal(rv.load_image(cam, 1), fr[1])
    rv.close_camera(cam)
    bad = bytearray(raw)
    bad[256 + 12 + 5] ^= 0xFF  # damage inside the first compressed frame
    cam = rv.open_camera_memory(bytes(bad))
    with pytest.raises(RuntimeError):
        rv.load_image(cam, 0)
    assert np.array_equal(rv.load_image(cam, 2), fr[2])
    rv.close_camera(cam)


def test_emissivity_and_calibration_surface_without_a_gpu(tmp_path):
    """The loader's emissivity map is state of the loader whether or not a calibration uses it: what is set is what is read back
    (reference tests/python/test_video_io.py:187-232 and test_IRMovie.py:159-180,338-360 replayed on a raw PCR movie, which needs no
    device).  Calibration 0 is the identity, nothing else exists; names and exceptions are the wrapper's."""
    from librir_amd.video_io import CalibrationNotFound, IRMovie

    fr = np.random.default_rng(3).integers(0, 16000, (3, 12, 16)).astype(np.uint16)
    p = tmp_path / "e.pcr"
    write_pcr(p, fr, frequency=50)
    with IRMovie.from_filename(p) as movie:
        h = movie.handle
        assert rv.support_emissivity(h) is False and not movie.support_emissivity
        # nothing set yet: one value, 1 (video_io.cpp:329-337), and the movie answers ones
        first = rv.get_emissivity(h)
        assert first.shape == (12, 16) and first.flat[0] == 1 and not first.flat[1:].any()
        assert np.array_equal(movie.emissivity, np.ones((12, 16), np.float32)) and movie.global_emissivity == 1.0
        emi = np.ones((12, 16)) * 0.25
        rv.set_emissivity(h, emi)
        assert np.array_equal(rv.get_emissivity(h), emi)  # test_video_io.py:195-202
        part = np.full(5, 0.5, np.float32)  # fewer values than pixels: 1 for the rest (IRVideoLoader.h:58-73)
        rv.set_emissivity(h, part)
        back = rv.get_emissivity(h)
        assert np.array_equal(back.flat[:5], part) and (back.flat[5:] == 1).all()
        rv.set_global_emissivity(h, 0.5)
        assert (rv.get_emissivity(h) == 0.5).all() and rv.get_global_emissivity(h) == 0.5
        with pytest.raises(RuntimeError):
            rv.set_global_emissivity(h, 1.5)
        with pytest.raises(RuntimeError):
            rv.get_emissivity(-1)  # test_video_io.py:208-212
        with pytest.raises(RuntimeError):
            movie.emissivity = emi  # the movie refuses where its calibration takes none (IRMovie.py:419-423)
        with pytest.raises(RuntimeError):
            movie.global_emissivity = 0.9
        assert rv.camera_saturate(h) is False
        img = movie[0]
        out = movie.calibrate(img, 0)  # test_IRMovie.py:159-161
        assert np.array_equal(out, img) and out is not img
        assert np.array_equal(rv.calibrate_image(h, movie.data, 0), movie.data)  # test_video_io.py:187-191
        with pytest.raises(RuntimeError):
            rv.calibrate_image(h, img, 1)
        assert movie.calibration_files == []  # test_video_io.py:147-148
        movie.calibration = "DL"  # test_IRMovie.py:165-178
        assert movie._calibration_index == 0 and movie.calibration == "DL"
        with pytest.raises(CalibrationNotFound):
            movie.calibration = "T"
        with pytest.raises(CalibrationNotFound):
            movie.calibration = 1
        assert movie.calibration == "DL" and movie._calibration_index == 0
        assert movie.to_thermavip() is None
    with pytest.raises(RuntimeError):
        rv.change_hcc_external_blackbody_temperature(p, 20.0)


def test_attribute_accessors_one_at_a_time(tmp_path):
    """reference tools/rir_tools.py:134-316: counts, names, values and time stamps of a trailer, item by item"""
    from librir_amd.low_level.misc import get_memory_folder
    from librir_amd.tools import rir_tools as rt

    p = tmp_path / "a.bin"
    p.write_bytes(b"payload")
    fa = FileAttributes.from_filename(p)
    assert fa.is_open()
    fa.attributes = {"cam": "A", "n": "3"}
    fa.timestamps = [5, 7, 11]
    fa.set_frame_attributes(1, {"k": b"v\x00w"})
    fa.flush()
    h = fa.handle
    assert rt.attrs_global_attribute_count(h) == 2 and rt.attrs_frame_attribute_count(h, 1) == 1 and rt.attrs_frame_attribute_count(h, 0) == 0
    got = {rt.attrs_global_attribute_name(h, i): rt.attrs_global_attribute_value(h, i) for i in range(2)}
    assert got == {"cam": b"A", "n": b"3"}
    assert rt.attrs_frame_attribute_name(h, 1, 0) == "k" and rt.attrs_frame_attribute_value(h, 1, 0) == b"v\x00w"
    assert [rt.attrs_frame_timestamp(h, i) for i in range(3)] == [5, 7, 11]
    rt.attrs_set_time(h, 2, 13)
    assert rt.attrs_frame_timestamp(h, 2) == 13
    with pytest.raises(RuntimeError):
        rt.attrs_frame_timestamp(h, 3)
    with pytest.raises(RuntimeError):
        rt.attrs_global_attribute_count(-1)
    fa.close()
    assert not fa.is_open()
    os.environ["LIBRIR_TEMP_FOLDER"] = str(tmp_path / "mem")
    try:
        (tmp_path / "mem").mkdir()
        folder = get_memory_folder()
        assert folder == tmp_path / "mem" / "cache" and folder.is_dir()
    finally:
        del os.environ["LIBRIR_TEMP_FOLDER"]


def test_low_level_helpers_behave_like_the_reference():
    """reference tests/python/test_rir.py:21-74: the string helpers of low_level.misc and the zstd wrappers on str arguments"""
    from librir_amd.low_level.misc import toArray, toCharP, toString

    with pytest.raises(TypeError):
        toString("a")
    assert toString(b"a") == "a" and toString(b"\xe2\x82\xac") == "€" and toString(b"ab\x00c\x00") == "abc"
    assert np.array_equal(toArray("a"), np.array(("a",), dtype="c"))
    assert toCharP("a") == b"a" and toCharP(b"a") == b"a" and toCharP(1) == b"\x00"
    assert zstd_decompress(zstd_compress("toto")) == b"toto"
    for garbage in (b"garbage", "garbage"):
        with pytest.raises(RuntimeError):
            zstd_decompress(garbage)


def test_labelling_and_time_axis_wrappers_refuse_bad_arguments_without_a_gpu():
    """the argument checks of the four wrappers that came last (label_image, keep_largest_area, extract_times, resample_time_serie)
    happen before the library is asked: reference tests/python/test_rir.py:239-243,256-262,279-300.  Without a device the two labelling
    calls fail like every compute entry point (no CPU fallback); the two time-axis helpers are host bookkeeping and answer
    (tests/test_time_series.py)."""
    import librir_amd.signal_processing as sp

    with pytest.raises(RuntimeError):
        sp.label_image(np.ndarray((10, 10, 10)), 0)
    with pytest.raises(RuntimeError):
        sp.keep_largest_area(np.ndarray((10, 10), dtype="object"), 0)
    with pytest.raises(RuntimeError):
        sp.extract_times((), "inter")
    with pytest.raises(RuntimeError):
        sp.resample_time_serie([], range(10), [0, 1])
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            sp.label_image(np.ones((4, 4), np.uint16), 0)
        with pytest.raises(RuntimeError):
            sp.keep_largest_area(np.ones((4, 4), np.uint16), 0)


def test_package_surface_like_the_reference():
    """names a user of the reference package reaches without thinking about it: the top-level classes and modules
    (src/python/librir/__init__.py:4-12), ``low_level``'s exports (low_level/__init__.py), ``FileFormat`` and the reference's argument
    names on the calls that are made with keywords"""
    import inspect

    import librir_amd as librir
    from librir_amd.low_level import _geometry, _signal_processing, _tools, _video_io, createZeroArrayHandle, loadDlls, toArray, toCharP, toString  # noqa: F401
    from librir_amd.tools import rir_tools
    from librir_amd.video_io import FileFormat, video_file_format  # noqa: F401

    assert librir.IRMovie.__name__ == "IRMovie" and librir.IRSaver.__name__ == "IRSaver" and librir.BadPixels.__name__ == "BadPixels"
    assert librir.rir_video_io.FileFormat is FileFormat and librir.rir_tools is rir_tools and librir.misc.toString(b"x") == "x"
    assert librir.rir_signal_processing.translate is librir.signal_processing.translate
    with pytest.raises(AttributeError):
        librir.rir_geometry
    assert [m.name for m in FileFormat][:5] == ["PCR", "WEST", "PCR_ENCAPSULATED", "ZSTD_COMPRESSED", "H264"] and FileFormat.H264 == rv.FILE_FORMAT_H264
    ar, handle = createZeroArrayHandle((2, 3), np.int32)
    assert ar.shape == (2, 3) and handle[0] == 0 and loadDlls()[0] is _tools and _geometry is None
    for fn, names in ((rir_tools.attrs_set_frame_attributes, ["handle", "frame", "attributes"]), (rir_tools.attrs_open_buffer, ["buf"]),
                      (FileAttributes.frame_attributes, ["self", "frame_index"]), (FileAttributes.set_frame_attributes, ["self", "frame_index", "attributes"]),
                      (rv.calibration_files, ["movie_handle"]), (rv.load_image, ["camera", "pos", "calibration", "shape", "out"])):
        assert list(inspect.signature(fn).parameters) == names


def test_slices_fill_their_stack_in_place_with_pages_made_ahead(tmp_path):
    """IRMovie[a:b] / .data read every image straight into its row of the stack while threads of their own make the stack's pages
    (low_level.misc.touch_ahead over rir_host_touch); the touching must never change what the reads have written, whichever comes
    first.  A raw movie: no device needed."""
    from librir_amd.low_level.misc import touch_ahead
    from librir_amd.video_io import IRMovie

    rng = np.random.default_rng(8)
    fr = rng.integers(0, 16000, (30, 512, 640)).astype(np.uint16)  # 19.7 MB: above the size from which pages are made ahead
    p = tmp_path / "big.pcr"
    write_pcr(p, fr)
    with IRMovie.from_filename(p) as mov:
        assert np.array_equal(mov.data, fr)
        assert np.array_equal(mov[2:29:4], fr[2:29:4]) and np.array_equal(mov[-13:], fr[-13:]) and mov[4:4].shape == (0, 512, 640)
        row = np.empty((512, 640), np.uint16)
        assert mov.load_pos(7, out=row) is row and np.array_equal(row, fr[7])
        for bad in (np.empty((512, 640), np.int16), np.empty((512, 641), np.uint16), np.empty((640, 512), np.uint16).T):
            with pytest.raises(RuntimeError):
                mov.load_pos(0, out=bad)
    # the toucher against a writer, many times over: every byte is the writer's
    for rep in range(6):
        a = np.empty(48 << 20, dtype=np.uint8)
        with touch_ahead(a) as t:
            assert len(t.threads) == touch_ahead.THREADS
            for start in range(0, a.size, 1 << 20):
                a[start:start + (1 << 20)] = (start >> 20) + rep
        assert np.array_equal(a[::4096], ((np.arange(0, a.size, 4096) >> 20) + rep).astype(np.uint8))
        assert np.array_equal(a[4095::4096], ((np.arange(4095, a.size, 4096) >> 20) + rep).astype(np.uint8))
    small = np.zeros(100, np.uint8)
    with touch_ahead(small) as t:  # too small to bother
        assert t.threads == []


def test_zfile_images_decompressed_ahead_of_a_sequential_reader(tmp_path):
    """a ZFile read image after image has its next images decompressed ahead by threads of the loader; whatever the order of the reads -
    runs, jumps back and forth, a close with images still on their way - every image is the recorded one.  No device needed."""
    rng = np.random.default_rng(3)
    n, h, w = 60, 48, 80
    fr = rng.integers(0, 16000, (n, h, w)).astype(np.uint16)
    p = tmp_path / "z.bin"
    hd = rv.open_video_write(p, w, h, 50, 1, 0)
    for i in range(n):
        rv.image_write(hd, fr[i], i * 1000)
    rv.close_video(hd)
    for rep in range(12):
        cam = rv.open_camera_file(p)
        order = list(range(n)) if rep % 3 == 0 else (list(range(10)) + list(rng.permutation(n)) + list(range(20, 50)) + [5, 6, 7, 8, 9, 10, 3, 4, 5, 6, 7])
        for i in order:
            assert np.array_equal(rv.load_image(cam, int(i)), fr[int(i)]), (rep, i)
        rv.close_camera(cam)


def test_zfile_writer_compresses_on_its_threads_and_writes_in_order(tmp_path):
    """open_video_write / image_write with zstd on several threads: the file is the one the single-threaded writer makes, byte for byte
    (RIR_ZFILE_THREADS = 0, 1 and the default), and reads back; a writer closed at once leaves a valid empty file"""
    import hashlib
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from librir_amd.video_io import rir_video_io as rv\n"
            "rng = np.random.default_rng(4)\n"
            "n, h, w = 70, 60, 88\n"
            "fr = (rng.integers(0, 300, (n, h, w)) + np.arange(n)[:, None, None] * 3).astype(np.uint16)\n"
            "hd = rv.open_video_write(sys.argv[1], w, h, 50, 1, 3)\n"
            "for i in range(n): rv.image_write(hd, fr[i], i * 777)\n"
            "size = rv.close_video(hd)\n"
            "cam = rv.open_camera_file(sys.argv[1])\n"
            "assert rv.get_image_count(cam) == n\n"
            "for i in list(range(12)) + [40, 69]: assert np.array_equal(rv.load_image(cam, i), fr[i])\n"
            "assert rv.get_image_time(cam, 5) == 5 * 777 * 1000000  # (small stamps are taken as milliseconds: IRFileLoader.cpp:355-376)\n"
            "rv.close_camera(cam)\n"
            "print(size)\n") % root
    seen = set()
    for threads in ("0", "1", None):
        p = tmp_path / ("z%s.bin" % threads)
        env = dict(os.environ)
        env.pop("RIR_ZFILE_THREADS", None)
        if threads is not None:
            env["RIR_ZFILE_THREADS"] = threads
        r = subprocess.run([sys.executable, "-c", code, str(p)], env=env, stdout=subprocess.PIPE, check=True)
        seen.add((hashlib.sha256(p.read_bytes()).hexdigest(), r.stdout.decode().strip()))
    assert len(seen) == 1
    empty = tmp_path / "empty.bin"
    assert rv.close_video(rv.open_video_write(empty, 20, 10, 50, 1, 0)) == 256  # the two header blocks, no image
