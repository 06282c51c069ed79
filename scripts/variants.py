"""Development aid: build variants of the codec kernels with different -D switches and time them on
the GPU box in one gpurun call.

    python scripts/variants.py build  NAME="-DA=1 -DB=2" NAME2="..."     (here: cross-compile)
    python scripts/variants.py run                                        (on the GPU box)
Variants live in librir_amd/libs/variants/NAME.so (git-ignored like every .so)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from librir_amd import build as B  # noqa: E402

VDIR = os.path.join(B.LIBS, "variants")


def build(specs):
    B.build(verbose=False)
    if os.path.isdir(VDIR):
        shutil.rmtree(VDIR)
    os.makedirs(VDIR)
    objdir = os.path.join(B.HERE, "build")
    procs = []
    for spec in specs:
        name, flags = spec.split("=", 1)
        obj = os.path.join(VDIR, name + ".o")
        unit = "codec_kernels.hip"
        src = None
        fl = []
        for f in flags.split():
            if f.startswith("--src="):
                src = f[6:]  # e.g. an older revision: git show REV:librir_amd/csrc/codec_kernels.hip > /tmp/x.hip
            elif f.startswith("--unit="):
                unit = f[7:]  # the translation unit the variant replaces (default codec_kernels.hip)
            else:
                fl.append(f)
        src = src or os.path.join(B.CSRC, unit)
        cmd = [B.HIPCC] + B.COMMON + fl + ["-c", src, "-o", obj]
        procs.append((name, obj, unit, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, obj, unit, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise SystemExit("variant %s failed" % name)
        objs = [obj if f == unit + ".o" else os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and not f.endswith(".hooks.o")]
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, name + ".so")] + objs + ["-ldl", "-lpthread"])
        os.remove(obj)
        print("built", name)


def run(args):
    main = os.path.join(B.LIBS, B.LIB_NAME)
    keep = main + ".keep"
    shutil.copy(main, keep)
    env = dict(os.environ, RIR_SKIP_CHECK=os.environ.get("RIR_SKIP_CHECK", "1"))
    try:
        reps = int(os.environ.get("RIR_VARIANT_REPS", "2"))
        for f in sorted(os.listdir(VDIR)) * reps:
            if not f.endswith(".so"):
                continue
            shutil.copy(os.path.join(VDIR, f), main)
            script = os.environ.get("RIR_VARIANT_SCRIPT", "codec_time.py")
            spath = os.path.join(ROOT, "scripts", script)
            if not os.path.exists(spath):
                spath = os.path.join(ROOT, "tests", "perf", script)  # scripts that check against the oracle live with the tests
            r = subprocess.run([sys.executable, spath] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            keep_prefix = tuple(os.environ.get("RIR_VARIANT_GREP", "encode_tiles,alone,FAIL").split(","))
            lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith(keep_prefix)]
            print("== %-24s rc=%d" % (f[:-3], r.returncode))
            for ln in lines:
                print("   " + ln)
            sys.stdout.flush()
    finally:
        shutil.move(keep, main)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
