"""CPU: the C-ABI shared object loads, exports every symbol include/*.h declares, and fails loudly
(no CPU fallback) when no HIP device is present."""
import ctypes as ct
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(hdr).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
        text = re.sub(r"typedef[^;]*;", "", text)
        for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
            n = m.group(1)
            if n not in ("defined", "__attribute__", "sizeof", "visibility", "void"):
                names.add(n)
    return sorted(names)


def test_headers_declare_something():
    syms = declared_symbols()
    for must in ["translate", "gaussian_filter", "bad_pixels_create", "rir_codec_encode_device", "rir_codec_decode_device"]:
        assert must in syms


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_aliases_resolve_to_one_object():
    libs = os.path.join(ROOT, "librir_amd", "libs")
    for alias in ["libtools.so", "libsignal_processing.so", "libvideo_io.so", "libtools.so.6", "libsignal_processing.so.6", "libvideo_io.so.6"]:
        assert os.path.realpath(os.path.join(libs, alias)) == os.path.realpath(os.path.join(libs, "librir_amd.so"))


def test_cxx_log_names_of_the_reference_tools_library_are_exported():
    # the reference's libgeometry.so binds rir::logError from libtools.so (geometry.cpp:147,214; Log.h:31-35): mangled names
    out = os.popen("nm -D --defined-only %s" % os.path.join(ROOT, "librir_amd", "libs", "librir_amd.so")).read()
    for sym in ["_ZN3rir8logErrorEPKc", "_ZN3rir7logInfoEPKc", "_ZN3rir10logWarningEPKc", "_ZN3rir15getLastErrorLogEPcPi",
                "_ZN3rir16set_log_functionEPFviPKcE", "_ZN3rir11disable_logEv", "_ZN3rir12log_functionEv", "_ZN3rir18reset_log_functionEv"]:
        assert " T %s\n" % sym in out, sym


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/python/librir"), reason="build container only: needs the reference's Python wrapper")
def test_reference_wrapper_imports_and_runs_on_this_library():
    """INTEGRATION.md section 1, executed: the reference's own `librir` package over librir_amd.so + its own libgeometry.so
    (scripts/wrapper_drop_in_check.py; what it found is held in tests/golden/wrapper_drop_in.json)."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("wrapper_drop_in_check", os.path.join(ROOT, "scripts", "wrapper_drop_in_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.run()
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "wrapper_drop_in.json")))
    assert got == want
    assert got["import"] == "ok" and got["last_log_error_after_geometry"] == "Wrong data type"
    assert got["translate_without_device"] == "RuntimeError" and got["h264_add_image_lossless_without_device"] == "RuntimeError"


def test_the_product_library_has_no_test_hooks():
    """the fault-injection hooks (RIR_DEBUG_*) exist only in libs/librir_amd_testhooks.so"""
    libs = os.path.join(ROOT, "librir_amd", "libs")
    product = open(os.path.join(libs, "librir_amd.so"), "rb").read()
    hooks = open(os.path.join(libs, "librir_amd_testhooks.so"), "rb").read()
    for name in (b"RIR_DEBUG_LOSSY_GIVE_UP", b"RIR_DEBUG_LOSSY_BAIL", b"RIR_DEBUG_ECC_BAIL", b"RIR_LOSSY_CONST_PAIRS"):
        assert name not in product, name
        assert name in hooks, name


def test_layout_query_is_pure_host(lib):
    from librir_amd.device import codec_layout

    L = codec_layout(640, 512, 1000, 50)
    assert (L.ntiles, L.nchunks) == (640, 20)
    assert L.hdr_bytes == 640 * 20 * 50 * 8 and L.tile_off_bytes == 20 * 641 * 4 and L.chunk_off_bytes == 21 * 8
    L = codec_layout(83, 67, 5, 3)
    assert (L.ntiles, L.nchunks) == ((83 * 67 + 511) // 512, 2)
    with pytest.raises(RuntimeError):
        codec_layout(0, 512, 10, 50)


def test_argument_errors_do_not_need_a_device(lib):
    # reference conventions: unknown dtype char / strategy -> -1 (signal_processing.cpp:40-41,70-71)
    a = np.zeros((4, 4), np.uint16)
    b = a.copy()
    bg = np.zeros(1, np.uint16)
    lib.translate.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float, ct.c_void_p, ct.c_char_p]
    assert lib.translate(ord("H"), a.ctypes.data, b.ctypes.data, 4, 4, 0.0, 0.0, bg.ctypes.data, b"bogus") == -1
    assert lib.translate(ord("Z"), a.ctypes.data, b.ctypes.data, 4, 4, 0.0, 0.0, bg.ctypes.data, b"wrap") == -1
    lib.bad_pixels_correct.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p]
    assert lib.bad_pixels_correct(12345, a.ctypes.data, b.ctypes.data) == -1


def test_no_cpu_fallback_without_device(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from librir_amd.low_level.misc import last_error

    assert lib.rir_device_available() == 0
    a = np.arange(16, dtype=np.uint16).reshape(4, 4)
    b = a.copy()
    bg = np.zeros(1, np.uint16)
    lib.translate.argtypes = [ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_float, ct.c_float, ct.c_void_p, ct.c_char_p]
    assert lib.translate(ord("H"), a.ctypes.data, b.ctypes.data, 4, 4, 1.0, 0.0, bg.ctypes.data, b"nearest") == -1
    assert "no usable HIP device" in last_error()
    assert np.array_equal(a, b)  # nothing was computed anywhere
    lib.bad_pixels_create.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
    assert lib.bad_pixels_create(a.ctypes.data, 4, 4) == 0


def test_handle_registry_and_log(lib):
    from librir_amd.low_level.misc import last_error

    assert lib.get_void_ptr(987654) in (None, 0)
    lib.rm_void_ptr(987654)  # unknown handle: no-op
    lib.bad_pixels_destroy(987654)
    n = ct.c_int(0)
    lib.rir_codec_layout_query(0, 0, 0, 0, None)
    assert "rir_codec_layout_query" in last_error()
    assert lib.get_last_log_error(None, ct.byref(n)) == -1 and n.value > 0


def test_hash_bytes_matches_reference(lib):
    from oracle.pyoracle import Ref

    if not Ref.available():
        pytest.skip("oracle/_ref not built here")
    r = Ref()
    for f in (lib.hash_bytes, r.lib.hash_bytes):
        f.argtypes = [ct.c_void_p, ct.c_size_t]
        f.restype = ct.c_size_t
    rng = np.random.default_rng(0)
    for n in [0, 1, 7, 8, 9, 15, 16, 31, 1000]:
        buf = rng.integers(0, 256, max(n, 1)).astype(np.uint8)
        assert lib.hash_bytes(buf.ctypes.data, n) == r.lib.hash_bytes(buf.ctypes.data, n)


def test_resident_capacity_rule_and_plan(lib):
    """The selection logic behind every kernel whose workgroups wait for each other (runtime.h: resident launches): how many
    workgroups a device holds at once - with a margin per XCD - and how units (streams, sequences) are dealt to launches,
    including "a unit does not fit: take the launch-per-frame / launch-per-iteration path" (units per launch 0)."""
    lib.rir_resident_capacity_rule.argtypes = [ct.c_int] * 3
    lib.rir_resident_plan.argtypes = [ct.c_int] * 3 + [ct.POINTER(ct.c_int)]
    rule = lib.rir_resident_capacity_rule
    assert rule(5, 256, 8) == 1200  # MI355X, 5 workgroups of the loss kernel per CU: 8 x (160 - 10)
    assert rule(1, 256, 8) == 8 * (32 - 2)
    assert rule(8, 256, 8) == 8 * (256 - 16)
    assert rule(5, 32, 1) == 160 - 10  # one XCD (a CPX partition)
    assert rule(5, 250, 8) == 1250 - 78  # CUs not a multiple of the XCD count: treated as one pool
    assert rule(1, 8, 8) == 0 and rule(0, 256, 8) == 0 and rule(5, 0, 8) == 0 and rule(-1, 256, 8) == 0
    assert rule(2, 8, 8) == 8  # at least one place per XCD stays free

    def plan(capacity, wgs, units):
        out = (ct.c_int * 2)()
        assert lib.rir_resident_plan(capacity, wgs, units, out) == 0
        return out[0], out[1]

    assert plan(1200, 160, 7) == (7, 1)  # seven 640x512 streams in one launch
    assert plan(1200, 160, 32) == (7, 5)
    assert plan(1200, 160, 1) == (1, 1)
    assert plan(1200, 1201, 3) == (0, 0)  # a stream does not fit: not resident
    assert plan(0, 160, 3) == (0, 0)  # the runtime could not size the device: not resident
    assert plan(240, 256, 1) == (0, 0) and plan(240, 240, 2) == (1, 2)
    assert lib.rir_resident_plan(1, 1, 1, None) == -1
    # a kernel in two forms (the resident loss kernel: 1 200 places at 5 waves per SIMD, 1 440 with part of its state parked in LDS):
    # the larger, slower form only when it saves a launch - or when only it fits; launches filled evenly
    lib.rir_resident_plan_two_forms.argtypes = [ct.c_int] * 4 + [ct.POINTER(ct.c_int)]
    out3 = (ct.c_int * 3)()
    for units, want in ((1, (1, 1, 0)), (7, (7, 1, 0)), (8, (8, 1, 1)), (9, (9, 1, 1)), (10, (5, 2, 0)), (14, (7, 2, 0)), (16, (8, 2, 1)),
                        (18, (9, 2, 1)), (21, (7, 3, 0)), (32, (8, 4, 1)), (0, (0, 0, 0))):
        assert lib.rir_resident_plan_two_forms(1200, 1440, 160, units, out3) == 0
        assert tuple(out3) == want, (units, tuple(out3))
    assert lib.rir_resident_plan_two_forms(100, 1440, 160, 3, out3) == 0 and tuple(out3) == (3, 1, 1)  # (only the second form holds a unit)
    assert lib.rir_resident_plan_two_forms(100, 120, 160, 3, out3) == 0 and tuple(out3) == (0, 0, 0)  # (neither does: the caller's other path)
    assert lib.rir_resident_plan_two_forms(1, 1, 1, 1, None) == -1


def test_image_arrays_are_fresh_and_recycled_only_when_dead():
    """load_image returns a new array per call (reference rir_video_io.py); the memory of an image is handed out again only once
    neither the caller nor any view refers to the array any more - no reference counts are inspected."""
    from librir_amd.video_io import rir_video_io as R

    shape = (6, 10)
    a = R._image_buffer(*shape)
    a[:] = 7
    addr = a.ctypes.data
    view = a[2:4]
    del a
    b = R._image_buffer(*shape)
    b[:] = 9
    assert b.ctypes.data != addr and (view == 7).all()  # the view keeps the first image
    del view
    c = R._image_buffer(*shape)
    assert c.ctypes.data == addr and c.flags.writeable and c.shape == shape and c.dtype == np.uint16
    kept = [R._image_buffer(*shape) for _ in range(12)]
    for i, k in enumerate(kept):
        k[:] = i
    assert len({k.ctypes.data for k in kept}) == 12 and all((k == i).all() for i, k in enumerate(kept))
    assert (b == 9).all()


def test_host_copy_over_helper_threads(lib):
    """rir_host_copy = the copy the per-frame entry points use between the caller's memory and page-locked staging
    (csrc/host_copy.cpp): sizes around the hand-off threshold, unaligned ends, a burst (helpers hot), a pause (helpers parked
    and woken again) and several calling threads at once - every byte must arrive, nothing outside the destination may move."""
    import threading
    import time

    lib.rir_host_copy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int64]
    rng = np.random.default_rng(11)
    assert lib.rir_host_copy(None, None, -1) == -1
    assert lib.rir_host_copy(None, None, 0) >= 0
    src = rng.integers(0, 256, 4 << 20, dtype=np.uint8)
    helpers = None
    for rep in range(3):
        for n in [1, 4095, 192 * 1024 - 1, 192 * 1024, 192 * 1024 + 1, 640 * 512 * 2, 640 * 512 * 2 + 3, 1024 * 768 * 2, (4 << 20) - 129]:
            for a, b in [(0, 0), (1, 3), (64, 7)]:
                dst = np.full(n + 256, 0xA5, np.uint8)
                helpers = lib.rir_host_copy(dst.ctypes.data + 128 + a, src.ctypes.data + b, n - a - b if n > a + b else 0)
                m = n - a - b if n > a + b else 0
                assert np.array_equal(dst[128 + a:128 + a + m], src[b:b + m])
                assert (dst[:128 + a] == 0xA5).all() and (dst[128 + a + m:] == 0xA5).all()
        time.sleep(0.01 * rep)  # longer than the helpers' spin: the next round finds them parked
    assert helpers is not None and 0 <= helpers <= 7

    errors = []

    def worker(seed):
        r = np.random.default_rng(seed)
        s = r.integers(0, 256, 1 << 20, dtype=np.uint8)
        d = np.empty_like(s)
        for i in range(300):
            n = int(r.integers(1, s.size)) if i % 4 == 0 else int(r.integers(192 * 1024, s.size))  # (mostly sizes that are handed out)
            lib.rir_host_copy(d.ctypes.data, s.ctypes.data, n)
            if not np.array_equal(d[:n], s[:n]):
                errors.append((seed, i, n))
                return
            d[:n] = 0

    # more calling threads than cores here: callers are preempted between handing a part out and looking for it (a helper may by then
    # have finished a later job of another caller)
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_host_file_io_over_helper_threads(lib, tmp_path):
    """rir_host_file_rw = how a chunk travels between page-locked memory and the container file (csrc/host_copy.cpp: disjoint ranges
    of one descriptor written / read by the helper threads): what was written is in the file, at its offset, and comes back; a range
    that ends past the end of the file is a failure, not a short read."""
    lib.rir_host_file_rw.argtypes = [ct.c_int, ct.c_void_p, ct.c_int64, ct.c_int64, ct.c_int]
    rng = np.random.default_rng(12)
    path = str(tmp_path / "chunks.bin")
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
    try:
        pieces, off = [], 77
        for n in [1, 5000, 192 * 1024 + 5, 7 * 1000 * 1000 + 3, 640 * 512 * 2]:
            a = rng.integers(0, 256, n, dtype=np.uint8)
            assert lib.rir_host_file_rw(fd, a.ctypes.data, n, off, 1) == 0
            pieces.append((off, a))
            off += n + 13
        raw = np.fromfile(path, dtype=np.uint8)
        assert raw.size == off - 13
        for o, a in pieces:
            assert np.array_equal(raw[o:o + a.size], a)
            back = np.zeros_like(a)
            assert lib.rir_host_file_rw(fd, back.ctypes.data, a.size, o, 0) == 0
            assert np.array_equal(back, a)
        big = np.zeros(1 << 20, np.uint8)
        assert lib.rir_host_file_rw(fd, big.ctypes.data, big.size, raw.size - 1000, 0) == -1  # ends past the end of the file
        assert lib.rir_host_file_rw(fd, big.ctypes.data, 100, raw.size, 0) == -1
        assert lib.rir_host_file_rw(-1, big.ctypes.data, 10, 0, 0) == -1 and lib.rir_host_file_rw(fd, None, 10, 0, 1) == -1
    finally:
        os.close(fd)


def test_wide_buffer_stores_with_a_register_offset_are_followed_by_idle_cycles():
    """gfx950: a 16-byte buffer store with its scalar offset in a register, followed at once by a vector instruction that writes one of its data
    registers, stored that instruction's result (round 6) - and neither the ISA's wait-state table nor the compiler's hazard pass
    knows (scripts/ubench/store_hazard.hip isolates it).  scripts/store_hazard_check.py looks at the compiled kernels of every unit: each such store is
    followed by idle cycles before its data registers can be written; and the scan sees what it must see."""
    import importlib.util
    import shutil

    spec = importlib.util.spec_from_file_location("store_hazard_check", os.path.join(ROOT, "scripts", "store_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    risky = ["\tbuffer_store_dwordx4 v[44:47], v127, s[20:23], s40 offen nt", "\tv_and_or_b32 v44, v149, s36, v150", "\ts_nop 3"]
    safe = ["\tbuffer_store_dwordx4 v[44:47], v127, s[20:23], s40 offen nt", "\ts_add_i32 s0, s0, s54", "\t;;#ASMSTART", "\ts_nop 3", "\t;;#ASMEND", "\tv_and_or_b32 v44, v149, s36, v150"]
    constant_offset = ["\tbuffer_store_dwordx4 v[44:47], v127, s[20:23], 0 offen", "\tv_mov_b32 v44, 0"]  # (the compiler's own wait states cover this form)
    assert mod.scan(risky) == (1, 1) and mod.scan(safe) == (1, 0) and mod.scan(constant_offset) == (0, 0)
    if not shutil.which(mod.B.HIPCC):
        pytest.skip("no hipcc here")
    assert mod.check_all() == 0  # (every unit: today only the bounded-loss one has such stores - a large constant in a buffer address would make more)
