"""profiles/rNN_pmc_lossy_spec.json from what scripts/lossy_spec_profile.sh left in gpurun_out/lossy_spec_rNN:
    python scripts/lossy_spec_summary.py gpurun_out/lossy_spec_r06 > profiles/r06_pmc_lossy_spec.json"""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
W, H, HL, FRAMES = 640, 512, 509, 1000


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("rir::", "")
    m = re.match(r"lossy_const_run_kernel<(\d+), (\w+), (\w+), (\w+)>", n)
    if m:
        return "lossy_const_run_kernel<SPEC>" if m.group(4) in ("true", "1") else "lossy_const_run_kernel"
    m = re.match(r"lossy_spec_stats_kernel<(\w+)>", n)
    if m:  # the sums from the streaming kernel's byte plane / from the frames (one of the two returns at once)
        return "lossy_spec_stats_kernel<PLANE>" if m.group(1) in ("true", "1") else "lossy_spec_stats_kernel<FRAMES>"
    return re.sub(r"<.*", "", n)


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc", "*", "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "pmc", "*", "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k.startswith("lossy_"):
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
pmc = {k: dict({c: sum(v) / len(v) for c, v in sorted(cs.items())}, _dispatches=len(next(iter(cs.values())))) for k, cs in sorted(agg.items())}
stats = {}
for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
    k = short(r["Name"])
    if k.startswith("lossy_"):
        stats[k] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])}
calls = stats["lossy_const_run_kernel<SPEC>"]["calls"]  # one streaming launch that does work per call of 1 000 frames (the other passes' launches return at once)


def per_call(k):  # time of ALL launches of the kernel (the idle ones included) per call of 1 000 frames
    return stats[k]["total_ns"] / max(1, calls // 3) / 1e3 if k in stats else 0.0


def hbm(k):
    c = pmc.get(k, {})
    return (c.get("FETCH_SIZE", 0.0) * 2 + c.get("WRITE_SIZE", 0.0)) * 1024  # FETCH_SIZE counts 64-byte halves of gfx950's 128-byte requests (MI355X_MICROARCH.md): x 2; KB -> bytes


alg_run = 7 * W * HL + 4 * W * (H - HL)
alg_stats = W * HL
out = {
    "source": "scripts/lossy_spec_profile.sh: tests/perf/lossy_spec_time.py 1000 1 speculative,static (one 640x512 stream, 509 lossy rows, the reference's defaults 6 / 2 / 5 / 32, static scene, 1 000 frames per call: one pass, committed) under rocprofv3 --kernel-trace --stats and --pmc (FETCH_SIZE, WRITE_SIZE, SQ counters in passes of their own; every kernel's launches averaged - two of a call's three pass launches return at once)",
    "rates": open(os.path.join(d, "rates.txt")).read().splitlines(),
    "us_per_call_of_1000_frames (all launches of the kernel)": {k: round(per_call(k), 1) for k in sorted(stats)},
    "kernel_stats": stats,
    "counters_per_dispatch (mean over working and idle launches)": pmc,
    "note": "per-dispatch means mix the one working launch of a call with the idle ones (3 streaming / sums / verify launches are queued per group, the passes allowed); totals per call = mean x launches per call",
}
for k, alg, what in (("lossy_const_run_kernel<SPEC>", alg_run, "7 bytes per lossy pixel and frame (pixel in, pixel out, the frame leaving the running average, a byte of the plane), 4 per pixel past lossy_height"),
                     ("lossy_spec_stats_kernel<PLANE>", alg_stats, "1 byte per lossy pixel and frame (the byte plane: difference and class)")):
    if k in stats and k in pmc:
        n_launch = pmc[k]["_dispatches"]
        # the working launches carry all the traffic and nearly all the time: totals over the profile / working launches
        working = max(1, stats[k]["calls"] // 3)
        t_ms = stats[k]["total_ns"] / working / 1e6
        bytes_call = hbm(k) * 3  # mean over three launches per call, one of which works
        out[k] = {
            "ms_per_working_launch (total time of the kernel / calls of 1 000 frames)": t_ms,
            "hbm_bytes_per_working_launch (FETCH_SIZE x 2 + WRITE_SIZE, x 3: two of three launches idle)": bytes_call,
            "algorithmic_bytes_per_frame": alg,
            "algorithmic_bytes (%s)" % what: alg * FRAMES,
            "traffic_over_algorithmic": bytes_call / (alg * FRAMES),
            "achieved_GBs": alg * FRAMES / (t_ms * 1e-3) / 1e9,
            "frac_of_8_TBs": alg * FRAMES / (t_ms * 1e-3) / 8e12,
            "pmc_dispatches": n_launch,
        }
print(json.dumps(out, indent=1))
