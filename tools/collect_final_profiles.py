#!/usr/bin/env python3
"""After `tools/final_round.sh rN` has run on a GPU box: condense gpurun_out/ into profiles/r0N_* (rocprofv3 kernel stats + PMC
traffic per workload through tools/summarize_prof.py, the SQ counter table from the NEWEST counter file of every pass, the
register table, the bench line with its long form, the rank rehearsals).  usage: tools/collect_final_profiles.py"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RN = int(sys.argv[1]) if len(sys.argv) > 1 else 5
r, r0 = f"r{RN}", f"r{RN:02d}"

for tag, name in ((f"{r}_head", f"{r0}_headline"), (f"{r}_coding", f"{r0}_coding"), (f"{r}_pops", f"{r0}_pops"), (f"{r}_n200", f"{r0}_n200"),
                  (f"{r}_n200cds", f"{r0}_coding_n200"), (f"{r}_n1000cds", f"{r0}_coding_n1000"), (f"{r}_n5000cds", f"{r0}_coding_n5000")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), tag, name], check=True, stdout=subprocess.DEVNULL)

lines = [f"SQ counters of the scan kernels, round-{RN} final sources (tools/gpu_pmc.sh: three separate rocprofv3 --pmc passes per workload, no tracing); "
         f"kernel sources {bench.kernel_sources_sha()}.",
         "SQ_WAVE_CYCLES, SQ_ACTIVE_*, SQ_WAIT_* count quad-cycles (4 clocks); SQ_INSTS_* count instructions; one wave64 VALU instruction holds its SIMD for one quad-cycle.",
         "Per LDS-DMA instruction (SQ_INSTS_VMEM; 1 KiB of bases each): divide SQ_INSTS_VALU / _SALU / _LDS by it.", ""]
for tag, label in (("head", "headline"), ("coding", "c4"), ("pops", "pops"), ("n200", "n200"), ("n200cds", "n200cds"),
                   ("n1000cds", "n1000cds"), ("n5000cds", "n5000cds")):
    agg = {}
    for d in sorted(glob.glob(os.path.join(G, f"pmc_{r}_{tag}", "sq*"))):
        f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # this run's, not an earlier call's
        per = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "scan_" in row["Kernel_Name"] and "kernel" in row["Kernel_Name"]:
                per[row["Counter_Name"]].append(float(row["Counter_Value"]))
        w = sum(per["SQ_WAVE_CYCLES"]) / max(len(per["SQ_WAVE_CYCLES"]), 1)
        for k, v in per.items():
            agg[k] = (sum(v) / len(v), w)
    lines.append(f"== {label}")
    for k, (m, w) in sorted(agg.items()):
        lines.append(f"{k:26s} {m:18.0f}  {m / w if w else 0:8.4f} of wave cycles")
    vm = agg["SQ_INSTS_VMEM"][0]
    frac = lambda k: agg[k][0] / agg[k][1]
    lines.append(f"   per LDS-DMA instruction: VALU {agg['SQ_INSTS_VALU'][0] / vm:.1f}  SALU {agg['SQ_INSTS_SALU'][0] / vm:.1f}  LDS {agg['SQ_INSTS_LDS'][0] / vm:.1f}"
                 f"   | wave cycles in an instruction {frac('SQ_ACTIVE_INST_ANY'):.3f}, waiting {frac('SQ_WAIT_ANY'):.3f}")
    lines.append("")
open(os.path.join(P, f"{r0}_sq_counters_final_sources.txt"), "w").write("\n".join(lines))

subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), os.path.join(P, f"{r0}_kernel_resources.json")], check=True,
               stdout=subprocess.DEVNULL)
e2e = max(glob.glob(os.path.join(G, f"prof_{r}_e2e", "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # repack_kernel beside the scan
shutil.copyfile(e2e, os.path.join(P, f"{r0}_e2e_8192_files_kernel_stats.csv"))
line = open(os.path.join(G, f"{r}_final_bench_n1.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(P, f"{r0}_bench_n1.json"), "w").write(line + "\n")
detail = [l for l in open(os.path.join(G, f"{r}_final_bench_n1.err")) if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):]
json.loads(detail)
open(os.path.join(P, f"{r0}_bench_n1_detail.json"), "w").write(detail)
print("kernel sources", bench.kernel_sources_sha())
for f in sorted(glob.glob(os.path.join(P, f"{r0}_*_pmc_summary.json"))):
    j = json.load(open(f))
    print(os.path.basename(f), j.get("kernel_sources_sha16"), "traffic/alg", round(j.get("traffic_over_algorithmic", 0), 4), "frac", round(j.get("frac_of_peak_rocprof", j.get("frac", 0)), 4))
