"""GPU parity in the many-row regime against vectors frozen from the REFERENCE (G10, tests/golden/g10_many_rows.json:
units of 128 to 2^21 + 3 rows; the reference has no row limit, PolyFastA.py:252-261, :284-315).  This is where the
row-chunked kernels (`scan_tall_kernel`, its WIDE coding form above 2 048 rows), the 128-site-tile coding kernel at
101-200 rows, 64-bit H and SFS bins beyond the 256 kept in LDS run — until round 4 checked against the C oracle only.
Integers bit-exact; pi, theta_w, D to 1e-9 (JC pi 1e-6)."""
import numpy as np
import pytest

from polyfasta_amd import cli, scan, stats
from tests import _cases as C

pytestmark = pytest.mark.gpu
CASES = C.load("g10_many_rows.json")


def _label(c):
    g = c["gen"]
    return f"{'cds' if c['coding'] else 'plain'}-{g['kind']}-N{c['N']}xL{c['L']}" + ("-subset" if g.get("subset") else "")


def _check(c, out, i=0, lean=False):
    r = out.results[i]
    assert (int(r["seg_sites"]), int(r["H"])) == (c["S"], c["H"]), _label(c)
    if c["S"]:
        if c["sfs"] == "IndexError":
            assert int(r["sfs_bad"]) > 0
        else:
            assert int(r["sfs_bad"]) == 0 and C.sparse_sfs(out.sfs(i).tolist()) == c["sfs"], _label(c)
    if not lean:
        pos = out.var_positions(i)
        assert C.sha1_ints(pos) == c["pos_sha1"] and pos[:24].tolist() == c["pos_head"], _label(c)
    for jc, key in ((False, "poly"), (True, "poly_jc")):
        got = stats.finish_polymorphism(int(r["seg_sites"]), int(r["H"]), c["N"], c["L"], jc)
        want = c[key]
        assert got[0] == want[0]
        assert C.close(got[1], want[1], 1e-6 if jc else 1e-9) and C.close(got[2], want[2], 1e-9)
        assert C.close(got[3], want[3], 1e-9) or abs(got[3] - want[3]) < 1e-9
    if c["coding"]:
        k = c["cds"]
        assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), 3 * int(r["missing_cols"])) == \
               (k["seg_S"], k["seg_N"], k["H_S"], k["H_N"], k["nstops"], k["missing"]), _label(c)
        assert C.close(stats.syn_sites_from_bins(out.syn_num[i]), k["count_syn"], 1e-12)
        if not lean:
            assert C.sha1_ints(out.syn_positions(i)) == k["S_sha1"] and C.sha1_ints(out.non_positions(i)) == k["N_sha1"], _label(c)
        for jc, key in ((False, "row"), (True, "row_jc")):
            C.rows_close(cli.format_cds_row("x.fa", "NA", c["N"], c["L"], r, out.syn_num[i], jc), k[key], jc=jc)


@pytest.mark.parametrize("c", CASES, ids=[_label(c) for c in CASES])
def test_many_rows_against_the_reference(gpu, c):
    """every G10 case on its own: with bitmaps (positions of variable / S / N sites) and as the lean plan a --dir batch runs;
    a subset case as a row-index list over the full matrix (--pops on a huge alignment) and as a matrix of its own"""
    full, rows = C.g10_full_matrix_and_rows(c["gen"])
    _check(c, scan.scan_matrix(full, rows=rows, cds=c["coding"], want_pos=True, want_sn=c["coding"]))
    _check(c, scan.scan_matrix(full, rows=rows, cds=c["coding"]), lean=True)
    if rows is not None:
        _check(c, scan.scan_matrix(C.g10_matrix(c["gen"]), cds=c["coding"]), lean=True)


def test_many_rows_as_one_mixed_batch(gpu):
    """the cases up to 5 000 rows in ONE plan (a --dir batch of large alignments of every size: one bucket per kernel class),
    every unit against the reference's vectors"""
    cases = [c for c in CASES if c["N"] <= 5000]
    b = scan.Batch(stride_align=128)
    for c in cases:
        b.add_unit(b.add_matrix(C.g10_matrix(c["gen"])), cds=c["coding"])
    plan = scan.Plan(b.pack()[1])
    desc = plan.description
    plan.close()
    assert "scan_tall_kernel" in desc and "wide=1" in desc and "scan_kernel<" in desc, desc
    out = scan.scan_batch(b)
    for i, c in enumerate(cases):
        _check(c, out, i, lean=True)
