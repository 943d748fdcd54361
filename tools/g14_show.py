#!/usr/bin/env python3
"""Print the product's and the frozen lines of one run of a g14_explore JSON side by side.  usage: g14_show.py IN.json SEED [pops|files|dir]"""
import difflib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, "tools", "g14_explore.py")).read().split("if __name__ ==")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "g14_explore.py")}
exec(compile(src, "g14_explore", "exec"), ns)
from tests.test_cli_odd_gpu import run_inproc  # noqa: E402

blob = json.load(open(sys.argv[1]))
for seed in sys.argv[2].split(","):
    case = next(c for c in blob["cases"] if c["seed"] == int(seed))
    files, flags = ns["GENERATORS"][blob.get("generator", "mixed")](case["seed"])
    top = tempfile.mkdtemp()
    d = ns["write_dir"](top, files)
    pp = case["pops"]
    rc, out = run_inproc(["-d", d, "-s", "-p", ",".join(pp["keys"])] + flags)
    want = pp["stdout"].replace("{TMP}", top)
    print(f"== seed {seed} flags {flags} rc {rc}")
    for ln in difflib.unified_diff(want.split("\n"), out.split("\n"), "reference", "product", lineterm="", n=2):
        print(ln)
    for k, v in files.items():
        print("   ", repr(k), v[:100])
