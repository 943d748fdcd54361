"""The ctypes stub printed in INTEGRATION.md must really work: run it verbatim."""
import os
import re

import pytest

from oracle import polyfasta_oracle as orc
from tests import _cases as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_runs(gpu):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.findall(r"```python\n(.*?)```", text, flags=re.S)[0]
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    d = C.dict_from_gen({"seed": 5, "aln": 1, "N": 20, "L": 999, "theta": 0.05, "coding": True})
    res, sfs, syn = ns["gpu_summary"](d, 999, cds=True)
    summ = orc.integer_summary(d, 999)
    assert (res.seg_sites, res.H, sfs.tolist()) == (summ["S"], summ["H"], summ["sfs"])
    cs, s, n, nstops, missing = orc.getvarCDSsites(d, 999)
    assert (res.seg_S, res.seg_N, res.nstops, 3 * res.missing_cols) == (len(s), len(n), nstops, missing)
    assert abs(sum(int(syn[k]) / (3 * k) for k in range(1, 65)) - cs) < 1e-9
