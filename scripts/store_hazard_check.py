"""Every buffer store of 12 or 16 bytes whose scalar offset is a REGISTER must be followed by idle cycles before anything writes its data registers:
on gfx950 such a store, followed at once by a vector instruction that writes one of them, stored that instruction's result (found in round 6,
lossy_kernels.hip: buf_stn).  The ISA's wait-state table and the compiler's hazard pass (GCNHazardRecognizer::createsVALUHazard) hold
exactly this store form for safe, so nothing is inserted for it: this script compiles the kernels to assembly and looks.
    python scripts/store_hazard_check.py [unit.hip ...]      (default: every .hip of librir_amd/csrc; exit status 1 when a store is unprotected)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from librir_amd import build as B  # noqa: E402

STORE = re.compile(r"\s*buffer_store_dwordx[34] v\[(\d+):(\d+)\], (?:v\d+|off), s\[\d+:\d+\], (s\d+|m0|vcc_lo|vcc_hi)\b")
WRITE = re.compile(r"v_\w+ v\[?(\d+)")


def scan(lines, unit="?"):
    """-> (such stores, those of them whose data registers may be written before the idle cycles)"""
    stores = bad = 0
    for i, line in enumerate(lines):
        m = STORE.match(line)
        if not m:
            continue
        stores += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        safe = False
        for t in (x.strip() for x in lines[i + 1:i + 60]):
            if t.startswith("s_nop 3") or t.startswith("s_endpgm"):
                safe = True
                break
            w = WRITE.match(t)
            if w and lo <= int(w.group(1)) <= hi:
                break
        if not safe:
            bad += 1
            print("%s: unprotected: %s" % (unit, line.strip()))
    return stores, bad


def check(unit):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "unit.s")
        flags = [f for f in B.COMMON if f != "-fPIC"]
        subprocess.check_call([B.HIPCC] + flags + ["-S", "--cuda-device-only", os.path.join(B.CSRC, unit), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    stores, bad = scan(lines, unit)
    print("%s: %d buffer stores of 12 / 16 bytes with a register offset, %d unprotected" % (unit, stores, bad))
    return bad


def check_all(units=None):
    """every unit (compiled side by side) -> the number of unprotected stores"""
    from concurrent.futures import ThreadPoolExecutor

    units = units or sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=min(6, len(units))) as pool:
        return sum(pool.map(check, units))


if __name__ == "__main__":
    sys.exit(1 if check_all(sys.argv[1:]) else 0)
