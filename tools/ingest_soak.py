#!/usr/bin/env python3
"""Debug: the CLI on a directory of mixed files, every row against the C oracle, repeated."""
import io, os, sys, tempfile, shutil
from contextlib import redirect_stdout
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import cli, synth
from oracle import c_oracle
d = tempfile.mkdtemp(prefix="pfa_dbg_")
rng = np.random.default_rng(5)
want = {}
for i in range(37):
    n = int(rng.choice([12, 20, 50, 80]))
    L = int(rng.choice([900, 3000, 9000, 30000])) if i != 5 else 150000
    p = synth.make_params(n, 0.02)
    m = synth.synth_alignment(4200, i, L, p, coding=True)
    open(os.path.join(d, f"g{i:02d}.fa"), "w").write(synth.to_fasta_text(synth.headers_for(n, 2), m))
    want[f"g{i:02d}.fa"] = c_oracle.scan(m)["S"]
bad_runs = 0
for rep in range(int(os.environ.get('REPS','40'))):
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["-d", d, "-s"])
    wrong = []
    for ln in buf.getvalue().splitlines():
        f = ln.split(",")
        if want[f[0]] != int(f[4]):
            wrong.append((f[0], want[f[0]], int(f[4])))
    if wrong:
        print("rep", rep, "wrong:", wrong, flush=True)
    bad_runs += bool(wrong)
print("bad runs", bad_runs)
shutil.rmtree(d)
