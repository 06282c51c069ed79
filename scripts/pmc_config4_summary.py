#!/usr/bin/env python3
"""profiles/rNN_pmc_lossy.json / rNN_pmc_ecc.json from the passes of scripts/pmc_config4.sh: per kernel and grid, the mean
counters per dispatch and what follows from them (per wave instruction counts, waiting and VALU-active fractions of the wave
cycles, SIMD VALU utilisation = VALU-active fraction x resident waves per SIMD, HBM bytes per dispatch with the gfx950
FETCH_SIZE correction of /opt/skills/guides/MI355X_MICROARCH.md).
    python scripts/pmc_config4_summary.py gpurun_out/pmc_config4_r03 profiles/r03"""
import json
import subprocess
import sys

root, prefix = sys.argv[1], sys.argv[2]
SIMDS = 256 * 4
for name, pat in (("lossy", "lossy_run_kernel"), ("ecc", "ecc_run")):
    raw = json.loads(subprocess.check_output([sys.executable, "scripts/pmc_summary.py", root, pat, "by-grid"]))
    out = {}
    for k, c in raw.items():
        wc, waves = c["SQ_WAVE_CYCLES"], c["SQ_WAVES"]
        out[k] = {
            "counters_mean_per_dispatch": c,
            "per_wave": {"VALU": c["SQ_INSTS_VALU"] / waves, "SALU": c["SQ_INSTS_SALU"] / waves, "LDS": c["SQ_INSTS_LDS"] / waves,
                         "VMEM_RD": c["SQ_INSTS_VMEM_RD"] / waves, "VMEM_WR": c["SQ_INSTS_VMEM_WR"] / waves},
            "fraction_of_wave_cycles": {"waiting_any": c["SQ_WAIT_ANY"] / wc, "waiting_for_an_instruction": c["SQ_WAIT_INST_ANY"] / wc,
                                        "valu_active": c["SQ_ACTIVE_INST_VALU"] / wc},
            "resident_waves_per_simd": waves / SIMDS,
            "simd_valu_utilisation": c["SQ_ACTIVE_INST_VALU"] / wc * waves / SIMDS,
            "hbm_bytes_per_dispatch": {"fetch": c["FETCH_SIZE"] * 1024 * 2, "write": c["WRITE_SIZE"] * 1024},
        }
    json.dump({"source": "scripts/pmc_config4.sh (tests/perf/config4_pmc.py): rocprofv3 --pmc, separate passes; all waves of these kernels are resident for the whole launch", "kernels": out},
              open("%s_pmc_%s.json" % (prefix, name), "w"), indent=1)
    print(name, {k: round(v["simd_valu_utilisation"], 3) for k, v in out.items()})
