"""profiles/rNN_pmc_lossy_const.json from what scripts/profile_round.sh left in gpurun_out/profile_rNN:
    python scripts/lossy_const_summary.py gpurun_out/profile_r04 > profiles/r04_pmc_lossy_const.json
Inputs: lossy_const_pmc_summary.json (scripts/pmc_summary.py over the FETCH_SIZE / WRITE_SIZE / SQ passes), lossy_const_kernel_stats.csv
(rocprofv3 --kernel-trace --stats of the same command), lossy_const.log (the rates tests/perf/lossy_const_time.py printed)."""
import csv
import json
import re
import sys

d = sys.argv[1]
pmc = json.load(open(d + "/lossy_const_pmc_summary.json"))
stats = {r["Name"]: r for r in csv.DictReader(open(d + "/lossy_const_kernel_stats.csv"))}
log = open(d + "/lossy_const.log").read()


def kernel(prefix, table):
    for k, v in table.items():
        if prefix in k:
            return v
    raise SystemExit("no kernel %s" % prefix)


frames = 1000
W, H, HL = 640, 512, 509
run_c = kernel("lossy_const_run_kernel", pmc)
run_s = kernel("lossy_const_run_kernel", stats)
hist_s = kernel("lossy_hist_mode_runs_kernel", stats)
hist_c = kernel("lossy_hist_mode_runs_kernel", pmc)
fin_s = kernel("lossy_const_finish_kernel", stats)
ms = float(run_s["AverageNs"]) / 1e6
hbm = (run_c["FETCH_SIZE"] * 2 + run_c["WRITE_SIZE"]) * 1024  # FETCH_SIZE counts 64-byte halves of the 128-byte requests of gfx950 (MI355X_MICROARCH.md): x 2; KB -> bytes
alg = 6 * W * HL + 4 * W * (H - HL)  # per frame
hist_ms = float(hist_s["AverageNs"]) / 1e6
hist_hbm = (hist_c["FETCH_SIZE"] * 2 + hist_c["WRITE_SIZE"]) * 1024
rates = {m.group(1).strip(): int(m.group(2)) for m in re.finditer(r"(general form|constant form)\s+1 stream\(s\) x 1000 frames per call: (\d+) frames/s", log)}
out = {
    "source": "scripts/profile_round.sh: tests/perf/lossy_const_time.py 1000 1 (one 640x512 stream, 509 lossy rows, running average 32, low = high = 3, stdFactor 0; 1 000 frames per call) under rocprofv3 --kernel-trace --stats and --pmc (FETCH_SIZE, WRITE_SIZE and the SQ counters in passes of their own); assembled by scripts/lossy_const_summary.py",
    "counters_per_dispatch": pmc,
    "lossy_const_run_kernel": {
        "ms_per_dispatch (rocprofv3 --stats)": ms,
        "frames_per_dispatch": frames,
        "us_per_frame": ms * 1e3 / frames,
        "hbm_bytes_per_dispatch (FETCH_SIZE x 2 + WRITE_SIZE)": hbm,
        "hbm_bytes_per_frame": hbm / frames,
        "algorithmic_bytes_per_frame (6 bytes per lossy pixel: pixel in, pixel out, the frame leaving the running average re-read from the input; 4 per pixel of the rows past lossy_height)": alg,
        "traffic_over_algorithmic": hbm / frames / alg,
        "achieved_GBs (algorithmic bytes / time)": alg * frames / (ms * 1e-3) / 1e9,
        "frac_of_8_TBs": alg * frames / (ms * 1e-3) / 8e12,
        "valu_instructions_per_frame": run_c["SQ_INSTS_VALU"] / frames,
        "salu_instructions_per_frame": run_c["SQ_INSTS_SALU"] / frames,
        "valu_busy_fraction_of_wave_cycles": run_c["SQ_ACTIVE_INST_VALU"] / run_c["SQ_WAVE_CYCLES"],
    },
    "lossy_hist_mode_runs_kernel": {
        "ms_per_dispatch": hist_ms,
        "hbm_bytes_per_dispatch": hist_hbm,
        "algorithmic_bytes_per_dispatch (every lossy pixel once)": 2 * W * HL * frames,
        "achieved_GBs": 2 * W * HL * frames / (hist_ms * 1e-3) / 1e9,
        "frac_of_8_TBs": 2 * W * HL * frames / (hist_ms * 1e-3) / 8e12,
        "bound": "LDS atomics: 8 per lane and 16 bytes; profiles/r04_ubench_lds_atomic_rate.txt (scripts/ubench/lds_atomic_rate.hip): 4.7 ns per wave instruction and CU on 250 random bins -> 94 us for 1 000 frames of 640x509 before anything else",
    },
    "whole_call": {
        "constant form, frames/s": rates.get("constant form"),
        "general (resident) form, frames/s": rates.get("general form"),
        "histogram pass ms per 1 000 frames": hist_ms,
        "finish kernel ms": float(fin_s["AverageNs"]) / 1e6,
    },
}
print(json.dumps(out, indent=1))
