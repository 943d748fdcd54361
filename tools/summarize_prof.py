#!/usr/bin/env python3
"""Condense a tools/gpu_prof.sh run (gpurun_out/prof_<tag>/) into profiles/<name>_*:
the rocprofv3 --kernel-trace --stats table, the per-launch HBM counters with the
gfx950 correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE counts half the bytes
of a 16-B/lane coalesced stream: x2; WRITE_SIZE is exact), and the bench line
printed by the profiled command."""
import csv
import datetime
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_sources_sha: the bench line reports a traffic figure only for the sources it was taken with)
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copyfile(stats, os.path.join(dst, f"{name}_kernel_stats.csv"))


def counter(kind, cname):
    f = max(glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == cname and "pfa::scan_" in r["Kernel_Name"] and float(r["Counter_Value"]) > 0]
    return vals


fetch = counter("pmc_fetch", "FETCH_SIZE")
write = counter("pmc_write", "WRITE_SIZE")
bench_line = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
kstat = [r for r in csv.DictReader(open(stats)) if "pfa::scan_" in r["Name"]][0]
fetch_kb, write_kb = sum(fetch) / len(fetch), sum(write) / len(write)
traffic = 2.0 * fetch_kb * 1024 + write_kb * 1024
summary = {
    "command": "rocprofv3 (--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE) -- python3 bench.py --no-cpu-baseline "
               + " ".join(sys.argv[3:]),
    "kernel": kstat["Name"], "calls": int(kstat["Calls"]), "avg_ns": float(kstat["AverageNs"]),
    "min_ns": float(kstat["MinNs"]), "max_ns": float(kstat["MaxNs"]),
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "HBM bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests of a 16-B/lane stream at 64 B) "
                  "+ WRITE_SIZE x 1024 (exact; calibrated here on pfa::synth_kernel, which writes a known byte count)",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": bench_line["roofline"]["algorithmic_bytes_per_launch"],
    "workload": bench_line["config"],
    "bench_roofline_same_command": bench_line["roofline"],
    "kernel_sources_sha16": bench.kernel_sources_sha(),
    "collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
    "calibration": "profiles/r02_hbm_counter_calibration.txt: FETCH_SIZE x 2 measured here on tools/ubench/read_stream "
                   "(reads a known 16 GiB per dispatch: the counter reports 0.50003 of it), as MI355X_MICROARCH.md §HBM says; "
                   "WRITE_SIZE checked on pfa::synth_kernel",
}
json.dump(summary, open(os.path.join(dst, f"{name}_pmc_summary.json"), "w"), indent=1)
json.dump(bench_line, open(os.path.join(dst, f"{name}_bench_under_rocprof.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
