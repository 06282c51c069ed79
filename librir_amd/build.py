"""Build the HIP shared object in-tree: librir_amd/libs/librir_amd.so (gfx950).

hipcc cross-compiles without a GPU.  The same object is exposed under the four file names the
librir Python wrapper globs for (reference src/python/librir/low_level/misc.py:98-139:
``*tools*.so``, ``*geometry*.so``, ``*signal_processing*.so``, ``*video_io*.so``) through
symlinks, see INTEGRATION.md.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBS = os.path.join(HERE, "libs")
ROOT = os.path.dirname(HERE)
LIB_NAME = "librir_amd.so"
# The same library with the TEST HOOKS compiled in (-DRIR_TEST_HOOKS: fault injection through RIR_DEBUG_* variables, runtime.h).  Only the
# tests that force a bail-out path load it (tests/hook_cases.py, RIR_LIBRARY_VARIANT=testhooks); the product library contains no hook.
HOOKS_LIB_NAME = "librir_amd_testhooks.so"
HOOKS_UNITS = ["video_io_abi.cpp", "registration_abi.cpp"]  # the translation units that read a hook
# the names the wrapper globs for, and the SONAMEs the reference's own libraries record for each other (SOVERSION = major
# version, src/cpp/tools/CMakeLists.txt:83-84): the reference's libgeometry.so, which a drop-in keeps, NEEDs libtools.so.6
ALIASES = ["libtools.so", "libsignal_processing.so", "libvideo_io.so", "libtools.so.6", "libsignal_processing.so.6", "libvideo_io.so.6"]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # reference x86-64 build has no FMA: keep products and sums separate
    "-fvisibility=hidden",
    "-Wall",
    "-Wno-unused-function",
    "-I" + CSRC,
    "-I" + os.path.join(ROOT, "include"),
] + os.environ.get("RIR_EXTRA_CFLAGS", "").split()


def sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") or f.endswith(".cpp"):
            out.append(os.path.join(CSRC, f))
    return out


def _newer(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    if os.path.getmtime(src) > t:
        return True
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h") and os.path.getmtime(os.path.join(d, f)) > t:
                return True
    return False


def build(force=False, verbose=True):
    os.makedirs(LIBS, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(src, obj):
            cmd = [HIPCC] + COMMON + ["-c", src, "-o", obj]
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            if verbose:
                print("[librir_amd.build]", os.path.basename(src), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(out.decode(errors="replace"))
        elif verbose and out.strip():
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("librir_amd: hipcc compilation failed")
    target = os.path.join(LIBS, LIB_NAME)
    if force or procs or not os.path.exists(target):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + ["-ldl", "-lpthread"]
        subprocess.check_call(cmd)
    # the build with the test hooks: the units that read one compiled again with -DRIR_TEST_HOOKS, everything else shared
    hooks_target = os.path.join(LIBS, HOOKS_LIB_NAME)
    hprocs, hobjs = [], list(objs)
    for unit in HOOKS_UNITS:
        src = os.path.join(CSRC, unit)
        obj = os.path.join(objdir, unit + ".hooks.o")
        hobjs[hobjs.index(os.path.join(objdir, unit + ".o"))] = obj
        if force or _newer(src, obj):
            cmd = [HIPCC, "-x", "hip"] + COMMON + ["-DRIR_TEST_HOOKS", "-c", src, "-o", obj]
            hprocs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in hprocs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("librir_amd: hipcc compilation failed (test-hooks build)")
    if force or procs or hprocs or not os.path.exists(hooks_target):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", hooks_target] + hobjs + ["-ldl", "-lpthread"])
    for a in ALIASES:
        link = os.path.join(LIBS, a)
        if os.path.islink(link) or os.path.exists(link):
            os.remove(link)
        os.symlink(LIB_NAME, link)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
