"""GPU parity: the HIP scan (through the C ABI) against the golden fixtures
frozen from the reference and against the CPU oracle on the same seeded inputs.
Integers (S, H, SFS, positions, coding counters) must be bit-exact; floats
within 1e-9 relative (JC-corrected pi 1e-6), as BASELINE.json states."""
import itertools

import numpy as np
import pytest

from oracle import c_oracle
from oracle import polyfasta_oracle as orc
from polyfasta_amd import api, scan, stats, synth
from tests import _cases as C

pytestmark = pytest.mark.gpu

REL = 1e-9
REL_JC = 1e-6


def _check_l1(rec, out, i=0, variant_name=""):
    r = out.results[i]
    assert int(r["seg_sites"]) == rec["S"], (variant_name, rec.get("gen"))
    assert int(r["H"]) == rec["H"], (variant_name, rec.get("gen"))
    if rec["S"]:
        if rec["sfs"] == "IndexError":
            assert int(r["sfs_bad"]) > 0
        else:
            assert int(r["sfs_bad"]) == 0
            assert out.sfs(i).tolist() == rec["sfs"], (variant_name, rec.get("gen"))
    if out.var_bitmap is not None:
        pos = out.var_positions(i)
        assert len(pos) == rec["S"]
        assert pos[:24].tolist() == rec["pos_head"]
        assert C.sha1_ints(pos) == rec["pos_sha1"] if hasattr(C, "sha1_ints") else True
    for jc, key in ((False, "poly"), (True, "poly_jc")):
        got = stats.finish_polymorphism(int(r["seg_sites"]), int(r["H"]), rec["N"], rec["L"], jc)
        want = rec[key]
        assert got[0] == want[0]
        assert C.close(got[1], want[1], REL_JC if jc else REL)
        assert C.close(got[2], want[2], REL)
        assert C.close(got[3], want[3], REL) or abs(got[3] - want[3]) < 1e-9


@pytest.mark.parametrize("variant", [0, 2])
def test_g2_synthetic_grid(gpu, variant):
    cases = C.load("g2_synthetic.json")
    b = scan.Batch()
    for rec in cases:
        b.add_unit(b.add_matrix(C.matrix_from_gen(rec["gen"])))
    out = scan.scan_batch(b, want_pos=True, variant=variant)
    for i, rec in enumerate(cases):
        _check_l1(rec, out, i, f"variant{variant}")


def test_g1_examples(gpu):
    from polyfasta_amd import fasta
    g1 = C.load("g1_examples.json")
    for fn, rec in g1.items():
        aln = fasta.read_fasta_file(f"{C.GOLDEN}/example_theta_0.01/{fn}")
        out = scan.scan_matrix(aln.matrix, want_pos=True)
        _check_l1(rec, out)


def _check_cds(rec, m, variant=0):
    out = scan.scan_matrix(m, cds=True, want_pos=True, want_sn=True, variant=variant)
    r = out.results[0]
    cds = rec["cds"]
    assert int(r["seg_sites"]) == rec["S"] and int(r["H"]) == rec["H"]
    assert int(r["nstops"]) == cds["nstops"]
    assert 3 * int(r["missing_cols"]) == cds["missing"]
    assert out.syn_positions(0).tolist() == cds["S_pos"]
    assert out.non_positions(0).tolist() == cds["N_pos"]
    assert int(r["seg_S"]) == len(cds["S_pos"]) and int(r["seg_N"]) == len(cds["N_pos"])
    assert C.close(stats.syn_sites_from_bins(out.syn_num[0]), cds["count_syn"], 1e-12)
    # integers against the C oracle too (H_S / H_N are not in the fixtures)
    o = c_oracle.scan(m, cds=True)
    assert (int(r["H_S"]), int(r["H_N"])) == (o["H_S"], o["H_N"])
    from polyfasta_amd import cli
    for jc, key in ((False, "row"), (True, "row_jc")):
        row = cli.format_cds_row("x.fa", "NA", rec["N"], rec["L"], r, out.syn_num[0], jc)
        C.rows_close(row, cds[key], jc=jc)


@pytest.mark.parametrize("variant", [0, 2])
def test_g5_coding(gpu, variant):
    for rec in C.load("g5_cds.json"):
        m = C.matrix_from_gen(rec["gen"]) if "gen" in rec else C.matrix_from_literal(rec["literal"])
        _check_cds(rec, m, variant)


def test_g1_examples_coding(gpu):
    from polyfasta_amd import fasta
    for fn, rec in C.load("g1_examples.json").items():
        aln = fasta.read_fasta_file(f"{C.GOLDEN}/example_theta_0.01/{fn}")
        _check_cds(rec, aln.matrix)


def test_row_subsets_match_oracle(gpu):
    """Population row subsets (PolyFastA.py:122-134): one uploaded matrix, several units."""
    p = synth.make_params(60, 0.03)
    m = synth.synth_alignment(77, 0, 5000, p)
    subsets = [np.arange(0, 60, 2), np.arange(1, 60, 2), np.array([3, 59, 0, 17]), np.arange(60), np.array([5])]
    b = scan.Batch()
    mid = b.add_matrix(m)
    for rows in subsets:
        b.add_unit(mid, rows=rows)
        b.add_unit(mid, rows=rows, cds=True)
    out = scan.scan_batch(b, want_pos=True)
    for k, rows in enumerate(subsets):
        o = c_oracle.scan(m, rows=rows, cds=True)
        for i in (2 * k, 2 * k + 1):
            r = out.results[i]
            assert (int(r["seg_sites"]), int(r["H"])) == (o["S"], o["H"])
            assert out.sfs(i).tolist() == o["sfs"].tolist()
        r = out.results[2 * k + 1]
        assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), int(r["missing_cols"])) == \
               (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["nstops"], o["missing_cols"])


@pytest.mark.parametrize("n_rows", [601, 1000, 2048])
def test_many_rows_single_alignment(gpu, n_rows):
    """Hundreds to thousands of rows, coding or not: the row-chunked kernel (rows streamed through
    LDS in chunks, same 256-site tiles)."""
    p = synth.make_params(n_rows, 0.02)
    for L, coding in ((700, False), (333, True)):
        m = synth.synth_alignment(5, n_rows, L, p, coding=coding)
        m = synth.inject_symbols(m, 3, 0.001, b"-N")
        out = scan.scan_matrix(m, cds=coding, want_pos=True)
        o = c_oracle.scan(m, cds=coding, want_sites=True)
        r = out.results[0]
        assert (int(r["seg_sites"]), int(r["H"]), int(r["sfs_bad"])) == (o["S"], o["H"], o["sfs_bad"])
        assert out.sfs(0).tolist() == o["sfs"].tolist()
        assert out.var_positions(0).tolist() == np.flatnonzero(o["h_site"]).tolist()
        if coding:
            assert (int(r["seg_S"]), int(r["seg_N"]), int(r["H_S"]), int(r["H_N"]), int(r["nstops"]), int(r["missing_cols"])) == \
                   (o["seg_S"], o["seg_N"], o["H_S"], o["H_N"], o["nstops"], o["missing_cols"])


def test_api_mirror_matches_oracle(gpu):
    """The reference-named functions, called like the reference's no_header does."""
    g = {"seed": 42, "aln": 0, "N": 24, "L": 1200, "theta": 0.03, "coding": True}
    d = C.dict_from_gen(g)
    pos, var = api.getvarsites(d, 1200)
    opos, ovar = orc.getvarsites(d, 1200)
    assert pos == opos and list(var) == ovar
    assert api.getsfs(var) == orc.getsfs(ovar)
    for jc in (False, True):
        got, want = api.polymorphism(var, 1200, jc), orc.polymorphism(ovar, 1200, jc)
        assert got[0] == want[0] and all(C.close(x, y, REL_JC if jc else REL) for x, y in zip(got[1:], want[1:]))
    ssites, s, n, nstops, missing = api.getvarCDSsites(d, 1200)
    ocs, os_, on, onst, omiss = orc.getvarCDSsites(d, 1200)
    assert C.close(ssites, ocs, 1e-12) and s == sorted(os_) and n == sorted(on) and (nstops, missing) == (onst, omiss)
    # the var_s / var_n subsets are plain lists: they get packed and scanned again
    var_s = [var[pos.index(i)] for i in s if i in pos]
    got, want = api.polymorphism(var_s, ssites, True), orc.polymorphism([ovar[opos.index(i)] for i in os_], ocs, True)
    assert got[0] == want[0] and all(C.close(x, y, REL_JC) for x, y in zip(got[1:], want[1:]))
    assert C.close(api.nucleotide_diversity(var), orc.nucleotide_diversity(ovar), REL)
    assert C.close(api.wattersons_theta(var), orc.wattersons_theta(ovar), REL)
    assert C.close(api.Dvar(var), orc.Dvar(ovar), REL)
    with pytest.raises(IndexError):
        api.getsfs([["A", "A", "N"], ["A", "C", "C"]])


def test_device_generator_matches_spec(gpu):
    import ctypes
    from polyfasta_amd import _lib
    lib = _lib.load()
    for N, L, theta, cod in ((50, 1000, 0.01, 0), (7, 333, 0.2, 1), (200, 4099, 0.05, 0)):
        p = synth.make_params(N, theta)
        stride = scan.row_stride_for(L)
        sp = _lib.SynthParamsC(seed=11, first_aln=2, n_aln=3, n_rows=N, n_sites=L, row_stride=stride,
                               seg_thresh=p.seg_thresh, tri_thresh=p.tri_thresh, n_units=len(p.units), coding=cod)
        dev = scan.DeviceBuffer(3 * N * stride)
        _lib.check(lib.pfa_synth_fill_dev(dev.ptr, ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                          _lib.ptr(np.ascontiguousarray(p.units))))
        got = dev.download().reshape(3, N, stride)[:, :, :L]
        for m in range(3):
            assert (got[m] == synth.synth_alignment(11, 2 + m, L, p, coding=bool(cod))).all()
        dev.free()


def test_full_size_batch_properties(gpu):
    """BASELINE-sized units (N=50 x L=10 kb), a batch generated on the device:
    size-independent properties + an oracle-checked subsample."""
    import ctypes
    from polyfasta_amd import _lib
    lib = _lib.load()
    N, L, B = 50, 10000, 512
    p = synth.make_params(N, 0.01)
    stride = scan.row_stride_for(L)
    sp = _lib.SynthParamsC(seed=2026, first_aln=0, n_aln=B, n_rows=N, n_sites=L, row_stride=stride,
                           seg_thresh=p.seg_thresh, tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
    dev = scan.DeviceBuffer(B * N * stride)
    _lib.check(lib.pfa_synth_fill_dev(dev.ptr, ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                      _lib.ptr(np.ascontiguousarray(p.units))))
    descs = np.zeros(B, dtype=_lib.DESC_DTYPE)
    descs["base_off"] = np.arange(B, dtype=np.uint64) * (N * stride)
    descs["n_rows"], descs["n_sites"], descs["row_stride"], descs["mat_rows"] = N, L, stride, N
    descs["row_idx_off"] = _lib.PFA_NO_ROW_IDX
    table = scan.finalize_layout(descs, None, B * N * stride)
    plan = scan.Plan(table)
    plan.run(dev)
    out1 = plan.fetch()
    plan.run(dev, variant=2)
    out2 = plan.fetch()
    plan.close()
    # idempotence and agreement of the two cache policies of the tile loads
    assert (out1.results == out2.results).all() and (out1.sfs_flat == out2.sfs_flat).all()
    res = out1.results
    sfs = out1.sfs_flat.reshape(B, N // 2)
    assert (sfs.sum(axis=1) == res["seg_sites"]).all()           # every variable site lands in one SFS bin
    assert (res["sfs_bad"] == 0).all()
    assert (res["H"] >= 2 * (N - 1) * res["seg_sites"].astype(np.uint64)).all()   # a singleton contributes 2(N-1)
    assert (res["H"] <= (N * N) * res["seg_sites"].astype(np.uint64)).all()
    # subsample against the oracle on host-regenerated inputs
    for m in (0, 1, 255, 511):
        mat = synth.synth_alignment(2026, m, L, p)
        o = c_oracle.scan(mat)
        assert (int(res[m]["seg_sites"]), int(res[m]["H"])) == (o["S"], o["H"])
        assert sfs[m].tolist() == o["sfs"].tolist()
    # additivity over site ranges: two half-length units of one matrix sum to the whole
    half = np.zeros(2, dtype=_lib.DESC_DTYPE)
    half["base_off"] = [0, 4992]
    half["n_rows"], half["row_stride"], half["mat_rows"] = N, stride, N
    half["n_sites"] = [4992, L - 4992]
    half["row_idx_off"] = _lib.PFA_NO_ROW_IDX
    t2 = scan.finalize_layout(half, None, B * N * stride)
    pl2 = scan.Plan(t2)
    pl2.run(dev)
    o2 = pl2.fetch()
    pl2.close()
    assert int(o2.results["seg_sites"].sum()) == int(res[0]["seg_sites"])
    assert int(o2.results["H"].sum()) == int(res[0]["H"])
    assert (o2.sfs_flat.reshape(2, N // 2).sum(axis=0) == sfs[0]).all()
    dev.free()


def test_device_classifier_all_codon_sets(gpu):
    """G4 through the classifier the kernels run (wave-uniform form): all 1 830 pairs,
    all 35 990 triples and the random larger sets frozen from the reference, plus sets
    with stop codons mixed in (the kernels strip them, PolyFastA.py:331)."""
    import itertools
    from polyfasta_amd import api
    g4 = C.load("g4_codon_sets.json")
    sense = g4["sense"]
    sets = [list(p) for p in itertools.combinations(sense, 2)]
    want = list(g4["pairs"])
    sets += [list(t) for t in itertools.combinations(sense, 3)]
    want += list(bytes.fromhex(g4["triples"]))
    for cs, w in g4["sets"]:
        sets.append([cs[i:i + 3] for i in range(0, len(cs), 3)])
        want.append(w)
    n_plain = len(sets)
    for i in range(0, n_plain, 7):          # the same sets with stops mixed in classify the same
        sets.append(sets[i] + ["TAA", "TGA"])
        want.append(want[i])
    got = api.classify_codon_sets_on_device(sets)
    bad = np.flatnonzero(got != np.array(want, dtype=np.uint32))
    assert len(bad) == 0, (len(bad), sets[bad[0]], int(got[bad[0]]), want[bad[0]])


def test_lds_histogram_adds_without_compiler_waits_equal_the_builtin(gpu, tmp_path):
    """lds_add_u32 (pfa_kernels.h) issues `ds_add_u32` from inline asm so that the compiler's waitcnt insertion does not see it;
    its correctness rests on in-order LDS completion within a wavefront and on a wave_lds_sync() in front of every read of the
    bins — nothing the compiler checks.  Build the library both ways (-DPFA_LDS_ATOMIC_WAITS=1: the atomicAdd builtin) and
    compare every integer of the coding cases of G5 and of G10 up to 5 000 rows, bit for bit."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from polyfasta_amd import _lib, scan
from tests import _cases as C
_lib.init(0)
out = {}
cases = [("g5_%%d" %% i, C.matrix_from_gen(r["gen"]) if "gen" in r else C.matrix_from_literal(r["literal"])) for i, r in enumerate(C.load("g5_cds.json"))]
cases += [("g10_%%d" %% i, C.g10_matrix(c["gen"])) for i, c in enumerate(C.load("g10_many_rows.json")) if c["coding"] and c["N"] <= 5000]
b = scan.Batch(stride_align=128)
for name, m in cases:
    b.add_unit(b.add_matrix(m), cds=True)
o = scan.scan_batch(b)
np.savez(sys.argv[1], results=o.results, sfs=o.sfs_flat, syn=o.syn_num)
"""
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    from polyfasta_amd import _build
    alt = str(tmp_path / "libpfa_builtin_atomics.so")
    _build.build(extra_flags=["-DPFA_LDS_ATOMIC_WAITS=1"], lib_path=alt)
    outs = []
    for i, lib in enumerate((None, alt)):
        env = dict(__import__("os").environ)
        if lib:
            env["PFA_LIB"] = lib
        f = str(tmp_path / f"out{i}.npz")
        p = subprocess.run([sys.executable, "-c", code % {"root": root}, f], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(np.load(f))
    for k in ("results", "sfs", "syn"):
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k
    assert int(outs[0]["results"]["seg_S"].sum()) > 100 and int(outs[0]["syn"].sum()) > 1000
