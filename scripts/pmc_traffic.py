#!/usr/bin/env python3
"""HBM bytes per launch from the PMC summary (scripts/pmc_summary.py output), as prescribed by
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes, both count kilobytes, and FETCH_SIZE under-reports by a factor of two on gfx950.
    python scripts/pmc_traffic.py gpurun_out/profile_r01/pmc_summary.json > profiles/r01_pmc_traffic.json"""
import json
import sys

summ = json.load(open(sys.argv[1]))
out = {}
for k, c in summ.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fetch = c["FETCH_SIZE"] * 1024.0 * 2.0
    write = c["WRITE_SIZE"] * 1024.0
    out[k] = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
              "FETCH_SIZE_KB_raw": c["FETCH_SIZE"], "WRITE_SIZE_KB_raw": c["WRITE_SIZE"],
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean per dispatch; FETCH_SIZE doubled per the gfx950 correction"}
print(json.dumps(out, indent=1))
