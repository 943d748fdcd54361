"""Two-decimal rounding ties of sites_S / sites_N on the card (PolyFastA.py:307, :167, :178): the per-column codon sets of
pfa_plan_codon_sets against numpy, and G12 — rows frozen from the reference on alignments whose exact synonymous-site sum is
a tie — through the command line, compared AS TEXT."""
import io
import os
import shutil
from contextlib import redirect_stdout

import numpy as np
import pytest

from polyfasta_amd import api, cli, scan, stats, synth
from tests import _cases as C

pytestmark = pytest.mark.gpu


def run_cli(argv):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = cli.main(argv)
    return rc, buf.getvalue()


@pytest.fixture(scope="module")
def g12():
    return C.load("g12_rounding_ties.json")


def test_codon_sets_match_numpy_on_every_kind_of_unit(gpu):
    """pfa_plan_codon_sets: units of 1 to 9 000 rows (above 4 096 the rows are shared out over several workgroups that OR
    their sets), lengths that are not multiples of 3, N / - / lower-case-free IUPAC bytes, stop codons, row-index lists over
    a larger matrix, a unit that is not the plan's first, matrices passed explicitly and by "the plan's last run"."""
    rng = np.random.default_rng(5)
    shapes = [(1, 3), (1, 2), (2, 7), (50, 9000), (50, 9001), (64, 300), (200, 999), (1000, 3000), (9000, 301), (5, 64 * 3 + 1)]
    b = scan.Batch(stride_align=128)
    mats, units = [], []
    for n, L in shapes:
        m = synth.synth_alignment(77, n, L, synth.make_params(max(n, 2), 0.05), coding=True)[:n]
        m = synth.inject_symbols(m, n + L, 0.01, b"-NRY")
        if L >= 300:
            m = synth.plant_codons(m, 11, ["TAA", "TAG", "TGA", "ATG", "AAA", "CCC", "GGG", "TTT", "ACG"])
        mid = b.add_matrix(m)
        b.add_unit(mid, cds=True)
        mats.append(m)
        units.append((m, None))
        if n >= 50:
            rows = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.uint32)
            b.add_unit(mid, rows=rows, cds=bool(n % 100))      # (coding or not: the re-scan does not care)
            units.append((m, rows))
    bases, table = b.pack()
    dev = scan.DeviceBuffer(len(bases))
    try:
        dev.upload(bases)
        plan = scan.Plan(table)
        try:
            # before any run, with the matrices named
            for i in (0, 3, len(units) - 1):
                m, rows = units[i]
                assert np.array_equal(plan.codon_sets(i, dev), C.codon_sets_numpy(m if rows is None else m[rows])), i
            with pytest.raises(Exception):
                plan.codon_sets(0)                              # no run yet, no matrices named
            plan.run(dev)
            out = plan.fetch()
            for i, (m, rows) in enumerate(units):
                sub = m if rows is None else m[rows]
                assert np.array_equal(plan.codon_sets(i), C.codon_sets_numpy(sub)), (i, sub.shape)
                if table.descs[i]["flags"]:
                    assert np.array_equal(out.syn_num[i], C.syn_bins_numpy(sub)), i
            with pytest.raises(Exception):
                plan.codon_sets(len(units))
        finally:
            plan.close()
    finally:
        dev.free()


def _write_dir(root, cases, pops):
    d = os.path.join(root, "pops" if pops else "plain")
    os.makedirs(d)
    names = []
    for i, case in enumerate(cases):
        g = case["gen"]
        if bool(g.get("pops")) != pops:
            continue
        name = f"tie{i:02d}.fa"
        with open(os.path.join(d, name), "w") as fh:
            fh.write(synth.to_fasta_text(synth.headers_for(g["N"], g.get("pops", 0)), C.g12_matrix(g)))
        names.append((name, case))
    return d, names


@pytest.mark.parametrize("mode", ["streaming", "direct"])
def test_g12_rows_through_the_command_line_as_text(gpu, g12, tmp_path, monkeypatch, mode):
    """Every G12 alignment as a file of one directory, `-d DIR --cds [--jc] [-p pop1,pop2]`: sites_S and sites_N are the
    reference's characters, every other column within its tolerance (integers and text equal).  Both ingest modes: the
    copy pipeline (small slots, so the ties sit in different batches) and the direct mode of small jobs."""
    if mode == "streaming":
        monkeypatch.setenv("PFA_INGEST_NO_DIRECT", "1")
        monkeypatch.setenv("PFA_INGEST_SLOT_BYTES", str(1 << 20))
    root = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    root = os.path.join(root, f"pfa_g12_{os.getpid()}_{mode}")
    os.makedirs(root)
    try:
        n_tie_rows = 0
        for pops in (False, True):
            d, names = _write_dir(root, g12["cases"], pops)
            assert names
            for jc in (False, True):
                argv = ["-d", d, "-s", "--cds"] + (["--jc"] if jc else []) + (["-p", "pop1,pop2"] if pops else [])
                rc, out = run_cli(argv)
                assert rc == 0
                got = out.rstrip("\n").split("\n")
                want = []
                for name, case in names:
                    if case["gen"]["L"] % 3:
                        want.append((f"# CDS sequence length is not a multiple of 3: {name}", False))     # PolyFastA.py:114-119
                    for key in (["pop1", "pop2"] if pops else ["NA"]):
                        want.append((case["keys"][key]["row_jc" if jc else "row"].replace("x.fa", name, 1), case["keys"][key]["tie"]))
                assert len(got) == len(want), (argv, len(got), len(want))
                for a, (b, tie) in zip(got, want):
                    if b.startswith("#"):
                        assert a == b
                        continue
                    assert a.split(",")[:3] == b.split(",")[:3], (argv, a, b)          # file, sites_S, sites_N: the text
                    assert C.row_matches(a, b, True, jc) is None, (argv, a, b)
                    n_tie_rows += int(tie)
        assert n_tie_rows >= 40
    finally:
        shutil.rmtree(root, ignore_errors=True)


def test_api_mirror_returns_the_references_count_syn(gpu, g12):
    """api.getvarCDSsites(d, L)[0] == ref.getvarCDSsites(d, L)[0], bit for bit (repr), on ties and on ordinary alignments
    (G5's count_syn): the API mirror always replays the reference's accumulation."""
    n = 0
    for case in g12["cases"]:
        g = case["gen"]
        if g["N"] > 200:
            continue
        d = synth.to_dict(synth.headers_for(g["N"], g.get("pops", 0)), C.g12_matrix(g))
        for key, k in case["keys"].items():
            grp = d if key == "NA" else {h: s for h, s in d.items() if key in h}
            got = api.getvarCDSsites(grp, g["L"])[0]
            if k["count_syn_order_free"]:
                assert repr(got) == k["count_syn"], (g, key)
                n += 1
            assert round(got, 2) == round(float(k["count_syn"]), 2)
    assert n >= 8
    for rec in C.load("g5_cds.json"):
        d = rec["literal"] if "literal" in rec else C.dict_from_gen(rec["gen"])
        got = api.getvarCDSsites(d, rec["L"])[0]
        assert abs(got - rec["cds"]["count_syn"]) <= 1e-12 * max(1.0, rec["cds"]["count_syn"])


def test_pipe_and_single_file_paths_print_the_tie_like_the_reference(gpu, g12, tmp_path, monkeypatch):
    """the small-batch paths (one file with -f: the direct ingest; --pipe: scan_batch) on a G12 tie"""
    import sys
    case = next(c for c in g12["cases"] if not c["gen"].get("pops") and c["gen"]["N"] == 50 and c["keys"]["NA"]["tie"])
    g = case["gen"]
    text = synth.to_fasta_text(synth.headers_for(g["N"]), C.g12_matrix(g))
    f = tmp_path / "x.fa"
    f.write_text(text)
    want = case["keys"]["NA"]["row"]
    rc, out = run_cli(["-f", str(f), "-s", "--cds"])
    assert rc == 0 and out.rstrip("\n").split(",")[:3] == want.split(",")[:3]
    monkeypatch.setattr(sys, "stdin", io.TextIOWrapper(io.BytesIO(text.encode())))
    rc, out = run_cli(["--pipe", "-s", "--cds", "-n", "x.fa"])
    assert rc == 0 and out.rstrip("\n").split(",")[:3] == want.split(",")[:3], (out, want)


def test_replayed_count_equals_the_column_order_sum_on_random_units(gpu):
    """scan_batch(exact_syn=True): the card's codon sets -> the host's replay == the C oracle's count_syn (accumulated in
    column order like the reference's), THE SAME DOUBLE, on 400 random coding units in five plans: every alphabet, 1 to 300
    rows (classic, 128-site-tile and row-chunked kernels side by side), lengths that are not multiples of 3, row subsets."""
    from oracle import c_oracle
    rng = np.random.default_rng(77)
    alphabets = [b"ACGT", b"ACGT" * 4 + b"N-", b"ACGTRYKM-N", b"AC", b"ACGT" * 8 + b"?*"]
    checked = 0
    for plan_no in range(5):
        b = scan.Batch(stride_align=int(rng.choice([16, 128])))
        units = []
        for _ in range(80):
            n = int(rng.choice([1, 2, 3, 8, 33, 50, 64, 65, 130, 300]))
            L = int(rng.choice([1, 3, 4, 17, 255, 256, 257, 770, 1537, 3001]))
            alpha = np.frombuffer(alphabets[int(rng.integers(0, len(alphabets)))], dtype=np.uint8)
            m = np.repeat(rng.choice(alpha, size=L)[None, :], n, axis=0)
            hit = rng.random((n, L)) < float(rng.choice([0.0, 0.01, 0.05, 0.3]))
            m[hit] = rng.choice(alpha, size=int(hit.sum()))
            mid = b.add_matrix(m)
            rows = None
            if n >= 3 and rng.random() < 0.4:
                rows = np.sort(rng.permutation(n)[: int(rng.integers(1, n + 1))]).astype(np.uint32)
            b.add_unit(mid, rows=rows, cds=True)
            units.append((m, rows))
        out = scan.scan_batch(b, exact_syn=True)
        for i, (m, rows) in enumerate(units):
            want = c_oracle.scan(m, rows=rows, cds=True)["count_syn"]
            got = float(out.sites_s[i])
            if got != want:
                # only on a rounding tie whose digit the reference itself does not fix (G15): there the value it most often has
                sub = m if rows is None else m[rows]
                assert stats.syn_tie_candidates(C.syn_bins_numpy(sub)[None, :], [m.shape[1]])[0], (plan_no, i, m.shape, repr(got), repr(want))
                assert got == stats.syn_sites_most_likely(C.codon_sets_numpy(sub), m.shape[1]) and abs(got - want) < 1e-12
            checked += 1
    assert checked == 400


def test_ties_the_reference_does_not_fix_g15(gpu, tmp_path, monkeypatch):
    """G15: on these alignments the reference's own sites_S digit follows its hash seed (its output under 60 seeds is frozen);
    `-f`, `-d` in both ingest modes and the API mirror give the digits it printed most often."""
    g15 = C.load("g15_hash_seed_ties.json")
    d = tmp_path / "d"
    d.mkdir()
    for i, case in enumerate(g15["cases"]):
        (d / f"t{i:02d}.fa").write_bytes(case["text"].encode())
        rc, out = run_cli(["-f", str(d / f"t{i:02d}.fa"), "-s", "--cds"])
        assert rc == 0 and out.rstrip("\n").split("\n")[-1].split(",")[1:3] == case["most_often"], (case["label"], out)
    for env in ({}, {"PFA_INGEST_NO_DIRECT": "1", "PFA_INGEST_SLOT_BYTES": "4096"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for flags in (["--cds"], ["--cds", "--jc"]):
            rc, out = run_cli(["-d", str(d), "-s"] + flags)
            rows = [ln.split(",") for ln in out.split("\n") if ln and not ln.startswith("#")]
            assert rc == 0 and [r[1:3] for r in rows] == [c["most_often"] for c in g15["cases"]]
    from oracle import polyfasta_oracle as orc
    for case in g15["cases"]:
        dd = orc.parse_fasta_lines(io.StringIO(case["text"], newline=None))      # the reference's reader: dict header -> sequence
        L = len(next(iter(dd.values())))
        assert str(round(api.getvarCDSsites(dd, L)[0], 2)) == case["most_often"][0]


def test_zero_nonsynonymous_sites_g15(gpu, tmp_path):
    """G15 `zero_nonsynonymous_sites`: the reference dies with ZeroDivisionError in 41 of 60 runs and prints pi_N = 1.5e16 in
    the others (its hash seed); the product does what it mostly does — the comment line, then the same death, exit code 1."""
    import subprocess
    import sys
    z = C.load("g15_hash_seed_ties.json")["zero_nonsynonymous_sites"]
    most = max(z["outcomes"], key=lambda o: o["runs"])
    f = tmp_path / "x.fa"
    f.write_bytes(z["text"].encode())
    p = subprocess.run([sys.executable, "-m", "polyfasta_amd", "-f", str(f), "-s", "--cds"], capture_output=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert p.returncode == most["returncode"] == 1
    assert p.stdout.decode() == most["stdout"]
    assert p.stderr.decode().strip().splitlines()[-1] == most["stderr_tail"]


def test_jukes_cantor_at_its_singularity_g15(gpu, tmp_path):
    """G15 `jc_singularity`, a KNOWN DEVIATION: pi per site of exactly 0.75 makes the correction -0.75 log(1 - 4/3 x) undefined.
    The reference evaluates pi in floating point; where its rounding noise leaves 0.7499999999999999 it prints -0.75 log(2.2e-16)
    = 27.03, where it leaves 0.75 its bare `except` prints the uncorrected 0.75 (PolyFastA.py:512-516) — for the second text
    either, by hash seed.  This implementation computes pi from exact integers and always prints the reference's fallback, 0.75:
    one of the reference's own outputs for the second text, not for the first.  Every other column is the reference's."""
    g = C.load("g15_hash_seed_ties.json")["jc_singularity"]
    for case in g:
        f = tmp_path / "x.fa"
        f.write_bytes(case["text"].encode())
        rc, out = run_cli(["-f", str(f), "-s"] + case["flags"])
        assert rc == 0
        got = out.rstrip("\n").split("\n")[-1].split(",")
        cds = "--cds" in case["flags"]
        col = 8 if cds else 5          # pi_N / pi
        refs = [r.split(",") for r, _ in case["rows"]]
        assert float(got[col]) == 0.75
        assert all(float(r[col]) == 0.75 or float(r[col]) > 20.0 for r in refs)
        for r in refs:
            same = got[:col] + [r[col]] + got[col + 1:]
            assert C.row_matches(",".join(same), ",".join(r), cds, True) is None, (case["label"], got, r)
        if case["label"] == "five_sequences_two_sites":
            assert any(float(r[col]) == 0.75 for r in refs)          # here the product prints one of the reference's own rows
