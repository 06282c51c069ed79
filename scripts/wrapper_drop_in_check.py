#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY: the reference's own `librir` Python wrapper on top of librir_amd.so.

INTEGRATION.md section 1 as a program.  Nothing of the reference is copied into the repository and
nothing of this travels to the GPU box: the wrapper is used where it lies (`/root/reference/src/python`,
through symlinks in a temporary directory), `libgeometry.so` - the one native library a drop-in keeps -
is compiled from `/root/reference/src/cpp/geometry/*.cpp` into that temporary directory with the
reference's link line (`target_link_libraries(geometry PUBLIC tools)`, geometry/CMakeLists.txt:26;
SOVERSION = major version, tools/CMakeLists.txt:83-84; install RPATH `$ORIGIN`, CMakeLists.txt:22).

What is exercised needs no GPU:
  * `import librir`  (loadDlls, low_level/misc.py:98-139: four LoadLibrary calls, RTLD_NOW - fails when the
    library behind `libtools.so` lacks a C++ symbol libgeometry.so binds: rir::logError, geometry.cpp:147,214)
  * geometry through the kept library, and its error path: rir::logError must land in THIS build's log
    state, where the wrapper's `get_last_log_error` reads it back
  * `zstd_compress` / `zstd_decompress`, garbage -> RuntimeError (tests/python/test_rir.py:47-74)
  * `FileAttributes.from_filename`: write, reopen, read back (tools/FileAttributes.py)
  * `translate`, `label_image`, `h264_add_image_lossless` without a device: the wrapper's RuntimeError, not a crash
  * `extract_times` / `resample_time_serie` (host bookkeeping, no device needed): the reference tests' calls and results

Prints one JSON object; `tests/golden/wrapper_drop_in.json` holds it and `tests/test_abi.py` asserts a fresh
run equals it (skipped where `/root/reference` does not exist).  `--write` refreshes the golden file.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RIR_REFERENCE", "/root/reference")
GOLDEN = os.path.join(ROOT, "tests", "golden", "wrapper_drop_in.json")

CHILD = r'''
import ctypes as ct, json, os, sys, tempfile
import numpy as np
out = {}
import librir
from librir.low_level import misc
from librir.geometry import rir_geometry as G
from librir.tools import rir_tools as T
from librir.tools.FileAttributes import FileAttributes
from librir.signal_processing import rir_signal_processing as S
from librir.video_io import rir_video_io as V
out["import"] = "ok"
libs = os.path.dirname(os.path.realpath(misc._tools._name))
out["bound"] = {k: os.path.basename(os.path.realpath(getattr(misc, k)._name)) for k in ("_tools", "_geometry", "_signal_processing", "_video_io")}

# geometry through the reference's own library
img = np.zeros((8, 10), np.uint16)
G.draw_polygon(img, [[1, 1], [6, 1], [6, 5], [1, 5]], 7)
out["draw_polygon_sum"] = int(img.sum())
# ... and its error path: an image type geometry.cpp does not know -> rir::logError("Wrong data type") -> -1
try:
    G.draw_polygon(np.zeros((4, 4), np.complex64), [[0, 0], [2, 0], [2, 2]], 1)
    out["draw_polygon_bad_type"] = "no error"
except RuntimeError as e:
    out["draw_polygon_bad_type"] = "RuntimeError"
n = ct.c_int(256); buf = ct.create_string_buffer(256)
misc._tools.get_last_log_error(buf, ct.byref(n))
out["last_log_error_after_geometry"] = buf.raw[:n.value].decode()

# zstd wrappers
raw = bytes(range(256)) * 64
c = T.zstd_compress(raw)
out["zstd_round_trip"] = T.zstd_decompress(c) == raw and len(c) < len(raw)
try:
    T.zstd_decompress(b"not a zstd frame at all")
    out["zstd_garbage"] = "no error"
except RuntimeError:
    out["zstd_garbage"] = "RuntimeError"

# the metadata trailer through the wrapper's FileAttributes
d = tempfile.mkdtemp()
p = os.path.join(d, "some_file.bin")
open(p, "wb").write(b"payload-bytes" * 10)
with FileAttributes.from_filename(p) as f:
    f.attributes = {"Name": "drop-in", "Blob": bytes(range(200)) * 10}
    f.timestamps = np.arange(5, dtype=np.int64) * 20_000_000
    for i in range(5):
        f.set_frame_attributes(i, {"idx": str(i)})
with FileAttributes.from_filename(p) as f:
    out["trailer"] = {
        "frames": int(f.frame_count()),
        "timestamps": [int(t) for t in f.timestamps],
        "Name": f.attributes["Name"].decode(),
        "Blob_ok": f.attributes["Blob"] == bytes(range(200)) * 10,
        "frame3": {k: v.decode() for k, v in f.frame_attributes(3).items()},
    }
out["trailer"]["payload_intact"] = open(p, "rb").read(130) == b"payload-bytes" * 10

# compute entry points without a device: the wrapper's own error, no crash
def outcome(fn):
    try:
        fn()
        return "no error"
    except RuntimeError as e:
        return "RuntimeError"
fr = (np.arange(20 * 20, dtype=np.uint16).reshape(20, 20))
out["translate_without_device"] = outcome(lambda: S.translate(fr, 1.5, -0.5, "nearest"))
out["gaussian_without_device"] = outcome(lambda: S.gaussian_filter(fr.astype(np.float32), 0.75))
def record():
    h = V.h264_open_file(os.path.join(d, "x.h264"), 20, 20, 20)
    try:
        V.h264_add_image_lossless(h, fr, 0)
    finally:
        V.h264_close_file(h)
out["h264_add_image_lossless_without_device"] = outcome(record)
out["label_image_without_device"] = outcome(lambda: S.label_image(fr, 0))
out["keep_largest_area_without_device"] = outcome(lambda: S.keep_largest_area(fr, 0))
n = ct.c_int(256)
misc._tools.get_last_log_error(buf, ct.byref(n))
out["last_log_error_after_compute"] = buf.raw[:n.value].decode()
# the time-axis helpers need no device: the reference wrapper's own calls (tests/python/test_rir.py:232-262) on this library
t1, t2 = [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5], [-1, 3, 4, 4.3, 4.7]
out["extract_times_union"] = [float(v) for v in S.extract_times((t1, t2), "union")]
out["extract_times_inter"] = [float(v) for v in S.extract_times((t1, t2), "inter")]
out["extract_times_long"] = int(S.extract_times((t1, range(10000)), "union").size)
times = [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5, 5.6, 9.9, 10, 12, 13]
out["resample_default"] = [round(float(v), 9) for v in S.resample_time_serie(range(10), range(10), times)]
out["resample_padded"] = [round(float(v), 9) for v in S.resample_time_serie(range(10), range(10), times, 0)]
out["resample_nearest"] = [float(v) for v in S.resample_time_serie(range(10), range(10), times, None, False)]
out["resample_output_too_small"] = outcome(lambda: S.resample_time_serie([0, 1], [0, 1], np.linspace(0, 1, 9)))  # the wrapper's room: 2 * len(x)
print("JSON:" + json.dumps(out, sort_keys=True))
'''


def stage(tmp):
    """tmp/librir = the reference package (symlinks) + libs/ = this build + the reference's geometry."""
    pkg = os.path.join(tmp, "librir")
    os.makedirs(os.path.join(pkg, "libs"))
    src = os.path.join(REF, "src", "python", "librir")
    for e in os.listdir(src):
        if e != "libs":
            os.symlink(os.path.join(src, e), os.path.join(pkg, e))
    libs = os.path.join(pkg, "libs")
    ours = os.path.join(ROOT, "librir_amd", "libs")
    for e in os.listdir(ours):  # cp -P: the aliases stay symlinks to librir_amd.so
        s = os.path.join(ours, e)
        if (os.path.isdir(s) and not os.path.islink(s)) or "testhooks" in e:
            continue  # (scripts/variants.py keeps its experimental builds in a sub-directory; the build with the test hooks is not part of a drop-in)
        if os.path.islink(s):
            os.symlink(os.readlink(s), os.path.join(libs, e))
        else:
            shutil.copy2(s, os.path.join(libs, e))
    # rir_config.h exactly as oracle/build_ref.sh makes it: the reference's .in with its five name/version tokens filled in
    inc = os.path.join(tmp, "inc")
    os.makedirs(inc)
    text = open(os.path.join(REF, "rir_config.h.in")).read()
    for k, v in (("@PROJECT_NAME@", "librir"), ("@PROJECT_VERSION@", "6.1.2"), ("@PROJECT_VERSION_MAJOR@", "6"),
                 ("@PROJECT_VERSION_MINOR@", "1"), ("@PROJECT_VERSION_PATCH@", "2")):
        text = text.replace(k, v)
    open(os.path.join(inc, "rir_config.h"), "w").write(text)
    g = os.path.join(REF, "src", "cpp", "geometry")
    cmd = ["g++", "-std=c++14", "-O3", "-DNDEBUG", "-fPIC", "-shared", "-DBUILD_GEOMETRY_LIB", "-I" + inc,
           "-I" + os.path.join(REF, "src", "cpp", "tools"), "-I" + g,
           os.path.join(g, "geometry.cpp"), os.path.join(g, "Polygon.cpp"), os.path.join(g, "DrawPolygon.cpp"),
           "-Wl,-soname,libgeometry.so.6", "-Wl,-rpath,$ORIGIN", "-Wl,--no-undefined", "-L" + libs, "-l:libtools.so.6",
           "-o", os.path.join(libs, "libgeometry.so")]
    subprocess.check_call(cmd)
    return tmp


def run():
    if not os.path.isdir(os.path.join(REF, "src", "python", "librir")):
        raise SystemExit("wrapper_drop_in_check: %s is not present (build container only)" % REF)
    tmp = tempfile.mkdtemp(prefix="rir_dropin_")
    try:
        stage(tmp)
        env = dict(os.environ, PYTHONPATH=tmp, LIBRIR_DISABLE_JOBLIB="1", HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
        env.pop("LD_LIBRARY_PATH", None)  # the libraries must find each other through $ORIGIN alone
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stdout + p.stderr)
            raise SystemExit("wrapper_drop_in_check: the wrapper did not run (exit %d)" % p.returncode)
        line = [l for l in p.stdout.splitlines() if l.startswith("JSON:")][-1]
        return json.loads(line[5:])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    res = run()
    print(json.dumps(res, indent=1, sort_keys=True))
    if "--write" in sys.argv:
        with open(GOLDEN, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
            f.write("\n")
    elif os.path.exists(GOLDEN):
        want = json.load(open(GOLDEN))
        if want != res:
            raise SystemExit("wrapper_drop_in_check: differs from tests/golden/wrapper_drop_in.json")
        print("equal to tests/golden/wrapper_drop_in.json")
