"""Where ecc_run_multi_kernel's 248 bytes of scratch per lane are used: the gfx950 assembly of librir_amd/csrc/ecc_kernels.hip, scanned per function.
    python scripts/ecc_isa_excerpt.py > profiles/r06_ecc_multi_isa.txt
(hipcc -S --cuda-device-only with the library's flags; needs no GPU.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "ecc.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "librir_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "librir_amd", "csrc", "ecc_kernels.hip")],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")


def body(prefix):
    s = [i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or (l.startswith(prefix) and ": ;" in l)][0]
    e = next(i for i in range(s + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return s, e


def is_inst(l):
    t = l.strip()
    return l.startswith("\t") and t and not t.startswith(".") and not t.startswith(";")


print("ecc_kernels.hip for gfx950 (hipcc -O3 -ffp-contract=off, the library's flags): scratch per function\n")
txt = "\n".join(lines)
meta = re.search(r"\.name:\s+_ZN3rir20ecc_run_multi_kernel\S*\n(.*?)\n  - \.", txt, re.S)
for name, prefix in (("ecc_run_multi_kernel (the kernel: entry to s_endpgm)", "_ZN3rir20ecc_run_multi_kernel"), ("ecc_multi_service (out of line, called by the service workgroup of a sequence)", "_ZN3rir17ecc_multi_service")):
    s, e = body(prefix)
    b = lines[s:e]
    insts = [l for l in b if is_inst(l)]
    scratch = [(s + 1 + i, l.strip()) for i, l in enumerate(b) if re.search(r"\bscratch_(load|store)", l)]
    private_buf = [(s + 1 + i, l.strip()) for i, l in enumerate(b) if re.search(r"buffer_(load|store)\S*\s+\S+,\s*(off|v\d+),\s*s\[0:3\]", l)]
    calls = [(s + 1 + i, l.strip()) for i, l in enumerate(b) if re.search(r"s_swappc_b64|s_setpc_b64", l)]
    back = [(s + 1 + i, l.strip()) for i, l in enumerate(b) if re.search(r"s_cbranch_\w+\s+\.LBB\d+_\d+", l)]
    print("%s\n  assembly lines %d-%d, %d instructions, scratch_load / scratch_store: %d, buffer accesses through the private descriptor s[0:3]: %d, calls (s_swappc_b64): %d"
          % (name, s + 1, e, len(insts), len(scratch), len(private_buf), sum(1 for c in calls if "swappc" in c[1])))
    if scratch:
        spills = sum(1 for x in scratch if "Spill" in x[1] or "Reload" in x[1])
        print("  of the scratch instructions, %d are marked Folded Spill / Reload by the compiler (callee-saved registers of the call, and the sixteen 16-byte row loads it keeps in flight)" % spills)
        print("  first and last of them:")
        for x in scratch[:6] + [("...", "")] + scratch[-4:]:
            print("    %s  %s" % (str(x[0]).rjust(5), x[1]))
    else:
        print("  -> no scratch instruction anywhere in the kernel's own code: the pixel loop (its buffer_load_dword ... s[20:23] / s[40:43] / s[44:47] offen are the image,")
        print("     gradient and template taps through raw-buffer descriptors built from the sequence's pointers, not the private segment) never touches scratch")
        loop = [x for x in back]
        print("  loop back-edges in the kernel (s_cbranch to an earlier label): %d; the call into the service function:" % len(loop))
        for c in calls:
            if "swappc" in c[1]:
                print("    %s  %s" % (str(c[0]).rjust(5), c[1]))
    print()
m = re.findall(r"\.name:\s+(_ZN3rir\w+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", txt)
print("private_segment_fixed_size per kernel (the resource table: a kernel is charged for what the functions it calls use):")
for name, v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.symbol:\s+(\S+)\.kd", txt):
    print("  %6s  %s" % (name, v))
