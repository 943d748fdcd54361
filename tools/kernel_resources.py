#!/usr/bin/env python3
"""Register / spill table of every kernel instantiation of libpfa_hip.so, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles gfx950 without a GPU).
usage: tools/kernel_resources.py [out.json]   (prints a table; writes JSON when a path is given)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _build  # noqa: E402

KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill"}


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")


def main():
    rows = []
    for src in ("pfa_scan_classic.hip", "pfa_scan_tall.hip", "pfa_plan.hip", "pfa_synth.hip"):
        cmd = [_build._hipcc()] + _build.CFLAGS + ["-c", "-x", "hip", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
                                                   os.path.join(_build.CSRC, src)]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z\[\]/ ]+): (\d+)", line)
            if m and cur is not None and m.group(1).strip() in KEYS:
                cur[KEYS[m.group(1).strip()]] = int(m.group(2))
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["kernel"] = re.sub(r"\(.*", "", d)
    rows = [r for r in rows if "kernel" in r["kernel"]]
    print(f"{'kernel':58s} {'sgpr':>5s} {'vgpr':>5s} {'sspill':>6s} {'vspill':>6s} {'scratch':>7s} {'occ':>4s}")
    for r in rows:
        print(f"{r['kernel']:58s} {r.get('sgprs', 0):5d} {r.get('vgprs', 0):5d} {r.get('sgpr_spill', 0):6d} "
              f"{r.get('vgpr_spill', 0):6d} {r.get('scratch', 0):7d} {r.get('occupancy', 0):4d}")
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
