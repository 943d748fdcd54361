"""Per-unit kernel selection, narrow tiles, wide coding units, --sfs: each against the oracle."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from oracle import c_oracle
from polyfasta_amd import _lib, cli, scan, stats, synth
from tests import _cases as C

pytestmark = pytest.mark.gpu


def _check(out, i, m, cds=False, rows=None):
    mm = m if rows is None else m[np.asarray(rows)]
    o = c_oracle.scan(mm, cds=cds)
    r = out.results[i]
    assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"]), i
    assert out.sfs(i).tolist() == o["sfs"].tolist(), i
    if cds:
        assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), int(r["missing_cols"])) == \
               (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["nstops"], o["missing_cols"]), i
        assert C.close(stats.syn_sites_from_bins(out.syn_num[i]), o["count_syn"], 1e-12), i


def test_mixed_batch_runs_every_unit_on_its_own_kernel(gpu):
    """50-row, 100-row, 300-row and 1 000-row units, coding and not, with and without row subsets, in
    ONE plan: several launches (buckets), every unit equal to the oracle."""
    b = scan.Batch(stride_align=128)
    mats, specs = [], []
    for i, (n, L, cds) in enumerate([(50, 3000, False), (50, 2997, True), (100, 5000, False), (300, 2000, False),
                                     (1000, 700, False), (1000, 900, True), (7, 300, False), (64, 1000, True),
                                     (65, 1000, True), (128, 4000, False), (200, 1500, True), (50, 10000, False)]):
        p = synth.make_params(n, 0.03)
        m = synth.synth_alignment(77, i, L, p, coding=cds)
        if i % 4 == 3:
            m = synth.inject_symbols(m, 5, 0.002, b"-NRY")
        mid = b.add_matrix(m)
        mats.append(m)
        b.add_unit(mid, cds=cds)
        specs.append((len(mats) - 1, cds, None))
        if n >= 50:
            rows = np.arange(1, n, 2, dtype=np.uint32)
            b.add_unit(mid, rows=rows, cds=cds)
            specs.append((len(mats) - 1, cds, rows))
    bases, table = b.pack()
    dev = scan.DeviceBuffer(len(bases))
    dev.upload(bases)
    plan = scan.Plan(table)
    assert plan.n_launches >= 5, plan.description
    assert "scan_tall_kernel<cds=1" in plan.description and "scan_tall_kernel<cds=0" in plan.description
    assert "scan_kernel<cds=1" in plan.description and "scan_kernel<cds=0" in plan.description
    plan.run(dev)
    out = plan.fetch()
    for i, (mi, cds, rows) in enumerate(specs):
        _check(out, i, mats[mi], cds=cds, rows=rows)
    plan.run(dev)                      # the slab is clean again after a fetch: a second run gives the same
    out2 = plan.fetch()
    assert (out.results == out2.results).all() and (out.sfs_flat == out2.sfs_flat).all()
    plan.run(dev)
    plan.run(dev)                      # two runs without a fetch in between: still one run's worth
    out3 = plan.fetch()
    assert (out.results == out3.results).all() and (out.sfs_flat == out3.sfs_flat).all() and (out.syn_num == out3.syn_num).all()
    plan.close()
    dev.free()


@pytest.mark.parametrize("tw", [64, 128, 256])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 50, 63, 101, 130])
def test_narrow_tiles_equal_the_oracle(gpu, monkeypatch, tw, n):
    """every tile width on odd row counts (row groups with and without a last partial step) and lengths
    that end inside a tile, plain and through a row subset"""
    monkeypatch.setenv("PFA_FORCE_TW", str(tw))
    p = synth.make_params(max(n, 2), 0.05)
    b = scan.Batch(stride_align=16)
    mats = []
    for i, L in enumerate((1, 63, 64, 65, 129, 700, 1031)):
        m = synth.synth_alignment(31 + n, i, L, p)[:n]
        m = synth.inject_symbols(m, 9, 0.01, b"-NK") if i % 2 else m
        mats.append(m)
        mid = b.add_matrix(m)
        b.add_unit(mid)
        if n > 2:
            b.add_unit(mid, rows=np.arange(n - 1, -1, -2, dtype=np.uint32))
    out = scan.scan_batch(b)
    k = 0
    for m in mats:
        _check(out, k, m)
        k += 1
        if n > 2:
            _check(out, k, m, rows=np.arange(n - 1, -1, -2))
            k += 1


def test_coding_unit_above_2048_rows(gpu):
    """getvarCDSsites has no row limit (PolyFastA.py:284-315): a 5 000-row coding unit, its per-site H
    kept in 56 bits, against the oracle — records, SFS, synonymous bins and the S / N positions"""
    n, L = 5000, 1203
    p = synth.make_params(n, 0.004)
    m = synth.synth_alignment(8, 0, L, p, coding=True)
    m = synth.inject_symbols(m, 3, 0.0004, b"-N")
    out = scan.scan_matrix(m, cds=True, want_pos=True, want_sn=True)
    o = c_oracle.scan(m, cds=True, want_sites=True)
    _check(out, 0, m, cds=True)
    assert out.var_positions(0).tolist() == np.flatnonzero(o["h_site"]).tolist()
    assert out.syn_positions(0).tolist() == np.flatnonzero(o["labels"] == 1).tolist()
    assert out.non_positions(0).tolist() == np.flatnonzero(o["labels"] == 2).tolist()
    out2 = scan.scan_matrix(m, cds=True)          # the lean instantiation
    assert (out2.results == out.results).all() and (out2.syn_num == out.syn_num).all()


def test_sfs_flag_prints_the_folded_spectrum(gpu, tmp_path):
    """--sfs (non-default extra): the `sfs` column equals getsfs of the reference on its own example
    files (G1), NA where getsfs raises IndexError (a variable column with < 2 A/C/G/T alleles)"""
    g1 = C.load("g1_examples.json")
    ex = os.path.join(C.GOLDEN, "example_theta_0.01")
    buf = io.StringIO()
    with redirect_stdout(buf):
        assert cli.main(["-d", ex, "--sfs"]) == 0
    lines = buf.getvalue().rstrip("\n").split("\n")
    assert lines[0] == cli.HEADER_PLAIN + ",sfs"
    seen = 0
    for ln in lines[1:]:
        f = ln.split(",")
        if f[0] in g1 and f[0].endswith(".fa"):
            assert [int(x) for x in f[-1].split(";")] == g1[f[0]]["sfs"], f[0]
            seen += 1
    assert seen == 10
    (tmp_path / "gap.fa").write_text(">a\nAC-T\n>b\nACAT\n>c\nACAT\n>d\nAGAT\n")
    buf = io.StringIO()
    with redirect_stdout(buf):
        assert cli.main(["-f", str(tmp_path / "gap.fa"), "--sfs", "-s"]) == 0
    assert buf.getvalue().rstrip("\n").split(",")[-1] == "NA"


@pytest.mark.parametrize("replicas", [None, 1, 4, 32])
def test_output_replicas_sum_to_the_oracle(gpu, monkeypatch, replicas):
    """plans with few units and many chunks per unit spread their atomics over replicas of the output
    slab; the fetch sums them (64-bit H fields as 64-bit integers, the rest as 32-bit counters)"""
    if replicas is not None:
        monkeypatch.setenv("PFA_REPLICAS", str(replicas))
    p = synth.make_params(50, 0.01)
    b = scan.Batch(stride_align=128)
    mats = [synth.synth_alignment(3, 0, 10000, p), synth.synth_alignment(3, 1, 9000, p, coding=True)]
    b.add_unit(b.add_matrix(mats[0]))
    b.add_unit(b.add_matrix(mats[1]), cds=True)
    b.add_unit(1, rows=np.arange(0, 50, 3, dtype=np.uint32), cds=True)
    bases, table = b.pack()
    dev = scan.DeviceBuffer(len(bases))
    dev.upload(bases)
    plan = scan.Plan(table)
    if replicas is None:
        assert "output replicas" in plan.description, plan.description
    for _ in range(3):
        plan.run(dev)
        out = plan.fetch()
        _check(out, 0, mats[0])
        _check(out, 1, mats[1], cds=True)
        _check(out, 2, mats[1], cds=True, rows=np.arange(0, 50, 3))
    plan.close()
    dev.free()


@pytest.mark.parametrize("shape", ["one_plain", "one_coding", "coding_100_rows", "pops_of_200", "three_small"])
def test_run_fetch_small_plans(gpu, monkeypatch, shape):
    """pfa_plan_run_fetch (the one-call latency path) on small single-launch plans of every kernel shape they
    take — output replicas, the flag-word fetch, slab clearing between calls — repeated, against the oracle,
    and equal to run() + fetch()."""
    b = scan.Batch(stride_align=128)
    checks = []
    if shape == "one_plain":
        m = synth.synth_alignment(21, 0, 10000, synth.make_params(50, 0.01))
        b.add_unit(b.add_matrix(m))
        checks = [(0, m, False, None)]
    elif shape == "one_coding":
        m = synth.synth_alignment(21, 1, 9000, synth.make_params(50, 0.01), coding=True)
        b.add_unit(b.add_matrix(m), cds=True)
        checks = [(0, m, True, None)]
    elif shape == "coding_100_rows":
        m = synth.synth_alignment(21, 2, 6000, synth.make_params(100, 0.01), coding=True)
        b.add_unit(b.add_matrix(m), cds=True)
        checks = [(0, m, True, None)]
    elif shape == "pops_of_200":
        m = synth.synth_alignment(21, 3, 100000, synth.make_params(200, 0.01))
        mid = b.add_matrix(m)
        for k in range(2):
            rows = np.arange(k, 200, 2, dtype=np.uint32)
            b.add_unit(mid, rows=rows)
            checks.append((k, m, False, rows))
    else:
        for k, (n, L) in enumerate([(20, 1000), (50, 3000), (7, 129)]):
            m = synth.synth_alignment(21, 10 + k, L, synth.make_params(n, 0.05))
            b.add_unit(b.add_matrix(m))
            checks.append((k, m, False, None))
    bases, table = b.pack()
    dev = scan.DeviceBuffer(len(bases))
    dev.upload(bases)
    plan = scan.Plan(table)
    outs = []
    for rep in range(4):
        out = plan.run_fetch(dev)
        for i, m, cds, rows in checks:
            _check(out, i, m, cds=cds, rows=rows)
        outs.append(out)
    plan.run(dev)                       # the two-call path on the same plan still works after fused calls
    out = plan.fetch()
    assert (out.results == outs[0].results).all() and (out.sfs_flat == outs[0].sfs_flat).all()
    out = plan.run_fetch(dev)
    assert (out.results == outs[0].results).all() and (out.sfs_flat == outs[0].sfs_flat).all()
    if outs[0].syn_num is not None:
        assert (out.syn_num == outs[0].syn_num).all()
    plan.close()
    dev.free()


@pytest.mark.parametrize("replicas", [4, 32])
def test_replicas_summed_in_place_for_the_device_side_gather(gpu, monkeypatch, replicas):
    """pfa_plan_results_dev() sums the output replicas IN PLACE (dst == src in sum_replicas_kernel) for the
    RCCL gather that reads the records from device memory: the summed records equal what fetch() returns
    for the same run, and the oracle (the aliasing path had no test)."""
    import ctypes
    monkeypatch.setenv("PFA_REPLICAS", str(replicas))
    p = synth.make_params(50, 0.01)
    b = scan.Batch(stride_align=128)
    mats = [synth.synth_alignment(5, k, 30000, p, coding=bool(k & 1)) for k in range(3)]
    for k, m in enumerate(mats):
        b.add_unit(b.add_matrix(m), cds=bool(k & 1))
    bases, table = b.pack()
    dev = scan.DeviceBuffer(len(bases))
    dev.upload(bases)
    plan = scan.Plan(table)
    assert f"{replicas} output replicas" in plan.description
    for _ in range(2):
        plan.run(dev)
        ptr, nbytes = plan.results_dev()
        got = np.empty(nbytes, dtype=np.uint8)
        _lib.check(_lib.load().pfa_download(_lib.ptr(got), ctypes.c_void_p(ptr), nbytes))
        got = got.view(_lib.RESULT_DTYPE)
        out = plan.fetch()
        assert (got == out.results).all()
        for k, m in enumerate(mats):
            _check(out, k, m, cds=bool(k & 1))
    plan.close()
    dev.free()


def test_mixed_batch_costs_the_sum_of_its_parts(gpu):
    """4 000 x (50 x 10 kb) plus 16 x (1 000 x 10 kb) in ONE plan: the 50-row units keep their kernel and
    occupancy (round 1 moved the whole plan onto the row-chunked kernel), so the mixed plan takes about
    what the two homogeneous plans take together; its rows equal the oracle's"""
    import ctypes
    import json
    lib = _lib.load()
    L = 10000
    stride = scan.row_stride_for(L, 128)
    parts = []
    off = 0
    for n_aln, n in ((4000, 50), (16, 1000)):
        p = synth.make_params(n, 0.01)
        parts.append((n_aln, n, p, off))
        off += n_aln * n * stride
    dev = scan.DeviceBuffer(off)
    descs = []
    for n_aln, n, p, o in parts:
        sp = _lib.SynthParamsC(seed=99, first_aln=0, n_aln=n_aln, n_rows=n, n_sites=L, row_stride=stride,
                               seg_thresh=p.seg_thresh, tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
        _lib.check(lib.pfa_synth_fill_dev(ctypes.c_void_p(dev.ptr.value + o), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                          _lib.ptr(np.ascontiguousarray(p.units))))
        d = np.zeros(n_aln, dtype=_lib.DESC_DTYPE)
        d["base_off"] = o + np.arange(n_aln, dtype=np.uint64) * np.uint64(n * stride)
        d["n_rows"], d["n_sites"], d["row_stride"], d["mat_rows"], d["row_idx_off"] = n, L, stride, n, _lib.PFA_NO_ROW_IDX
        descs.append(d)

    def timed(d):
        plan = scan.Plan(scan.finalize_layout(d.copy(), None, off), timing=True)
        for _ in range(12):
            plan.run(dev)
        _lib.check(lib.pfa_sync())
        ms = float(np.sort(plan.kernel_ms(10))[:5].mean())
        out = plan.fetch()
        desc = plan.description
        plan.close()
        return ms, out, desc

    ms_small, out_small, _ = timed(descs[0])
    ms_big, out_big, _ = timed(descs[1])
    ms_mixed, out_mixed, desc = timed(np.concatenate(descs))
    assert "scan_tall_kernel" in desc and "scan_kernel<cds=0,mapped=0,bitmaps=0,tw=256>" in desc, desc
    assert (out_mixed.results[:4000] == out_small.results).all() and (out_mixed.results[4000:] == out_big.results).all()
    for k, (n_aln, n, p, o) in enumerate(parts):           # one unit of each kind against the oracle
        m = synth.synth_alignment(99, 3, L, p)
        oc = c_oracle.scan(m)
        r = (out_small, out_big)[k].results[3]
        assert (int(r["seg_sites"]), int(r["H"])) == (oc["S"], oc["H"])
    report = {"ms_4000x50": ms_small, "ms_16x1000": ms_big, "ms_mixed": ms_mixed, "ratio": ms_mixed / (ms_small + ms_big), "plan": desc}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open("gpurun_out/mixed_batch.json", "w"), indent=1)
    assert ms_mixed <= 1.10 * (ms_small + ms_big), report
