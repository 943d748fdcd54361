#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (it imports /root/reference/PolyFastA.py by
path); nothing in tests/, bench.py or smoke() needs the reference at run time.
For every fixture family it (1) builds the input with this repo's own
deterministic generator (polyfasta_amd/synth.py) or from literal text, (2)
calls the reference function, (3) asserts that the CPU oracle
(oracle/polyfasta_oracle.py) agrees, and (4) freezes the reference's outputs.

Fixtures are data only: generator parameters / literal FASTA text and the
reference's outputs.  The ten example alignments are copied as data files
(they are the only sample inputs the reference ships; GPL-3, attributed in
tests/golden/example_theta_0.01/ATTRIBUTION).

Families: G1 examples, G2 synthetic grid, G3 odd symbols and
FASTA syntax, G4 exhaustive codon-set tables, G5 coding alignments, G6
populations, G7 command-line transcripts, G8 the BASELINE.json configurations at their full sizes, G9 sequences
with non-ASCII characters, G10 units of hundreds to 2^20 rows, G11 a directory entry that cannot be read, G12 coding
alignments whose synonymous-site count is a two-decimal rounding tie, G13 odd inputs through the command line (the G3 texts
as a directory, population-key edges, the same path twice, a file that is not valid UTF-8), G14 random directories of small
awkward files through the command line in four flag combinations, G15 two-decimal ties whose printed digit follows the
reference's own hash seed (its output under 60 seeds).
"""
import os
import sys

# The reference sums over Python sets of strings (PolyFastA.py:258, :487-491): their iteration order, and with it the last bits
# of its float sums, follow the interpreter's string-hash seed.  Fixed here, so that regenerating gives the same bytes (this
# process and the CLI runs it starts inherit it) — in a fresh child process, before anything else is imported (never an exec).
if __name__ == "__main__" and os.environ.get("PYTHONHASHSEED") != "0":
    import subprocess
    sys.exit(subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, PYTHONHASHSEED="0")).returncode)

import contextlib
import hashlib
import importlib.util
import io
import itertools
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_DIR = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import polyfasta_oracle as orc  # noqa: E402
from polyfasta_amd import synth  # noqa: E402

spec = importlib.util.spec_from_file_location("PolyFastA", os.path.join(REF_DIR, "PolyFastA.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def close(a, b, tol=1e-11):
    if isinstance(a, str) or isinstance(b, str):
        return a == b
    return abs(a - b) <= tol * max(abs(a), abs(b), 1e-300) or a == b


def sha1_ints(xs):
    return hashlib.sha1(np.asarray(xs, dtype=np.int64).tobytes()).hexdigest()


def int_H(var):
    n = len(var[0]) if var else 0
    return sum(n * n - sum(col.count(a) ** 2 for a in set(col)) for col in var)


def ref_sfs(var):
    if not var:
        return None
    try:
        return ref.getsfs(var)
    except IndexError:
        return "IndexError"


def capture(fn, *args):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*args)
    return buf.getvalue()


def l1_record(d, seqlen, check_oracle=True):
    """Reference outputs of the non-coding L1 functions on dict `d`."""
    pos, var = ref.getvarsites(d, seqlen)
    rec = {
        "N": len(d), "L": seqlen, "S": len(var), "H": int_H(var),
        "sfs": ref_sfs(var), "pos_sha1": sha1_ints(pos), "pos_head": pos[:24],
        "poly": list(ref.polymorphism(var, seqlen, False)),
        "poly_jc": list(ref.polymorphism(var, seqlen, True)),
    }
    if check_oracle:
        opos, ovar = orc.getvarsites(d, seqlen)
        assert (opos, ovar) == (pos, var)
        summ = orc.integer_summary(d, seqlen)
        assert summ["S"] == rec["S"] and summ["H"] == rec["H"]
        if var:
            assert (summ["sfs"] if summ["sfs"] is not None else "IndexError") == rec["sfs"]
        for jc, key in ((False, "poly"), (True, "poly_jc")):
            got = orc.polymorphism(ovar, seqlen, jc)
            assert all(close(x, y) for x, y in zip(got, rec[key])), (got, rec[key])
    return rec


def cds_record(d, seqlen):
    cs, s, n, nstops, missing = ref.getvarCDSsites(d, seqlen)
    ocs, os_, on, onstops, omissing = orc.getvarCDSsites(d, seqlen)
    assert close(cs, ocs, 1e-12) and sorted(s) == sorted(os_) and sorted(n) == sorted(on)
    assert (nstops, missing) == (onstops, omissing)
    rows = {}
    for jc in (False, True):
        row = capture(ref.print_result, d, seqlen, True, "", "a", "x.fa", "NA", True, 0, jc).rstrip("\n")
        rows["row_jc" if jc else "row"] = row
        assert_rows_close(orc.result_row(d, seqlen, True, "x.fa", "NA", jc), row)
    return {"count_syn": cs, "S_pos": sorted(s), "N_pos": sorted(n), "nstops": nstops,
            "missing": missing, **rows}


def assert_rows_close(a, b):
    fa, fb = a.split(","), b.split(",")
    assert len(fa) == len(fb), (a, b)
    for x, y in zip(fa, fb):
        if x == y:
            continue
        assert close(float(x), float(y), 1e-10), (a, b)


def gen_case(seed, aln, N, L, theta, coding=False, inject=None, pops=0):
    params = synth.make_params(N, theta)
    m = synth.synth_alignment(seed, aln, L, params, coding=coding)
    if inject:
        m = synth.inject_symbols(m, inject["seed"], inject["frac"], inject["alphabet"].encode())
    return synth.to_dict(synth.headers_for(N, pops), m)


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as fh:
        json.dump(obj, fh, indent=None, separators=(",", ":"))
        fh.write("\n")
    print(f"wrote {name}: {os.path.getsize(path)} bytes")


# ---------------------------------------------------------------- G1 examples
def g1():
    src = os.path.join(REF_DIR, "example_theta_0.01")
    dst = os.path.join(HERE, "example_theta_0.01")
    os.makedirs(dst, exist_ok=True)
    out = {}
    for fn in sorted(os.listdir(src)):
        shutil.copyfile(os.path.join(src, fn), os.path.join(dst, fn))
        os.chmod(os.path.join(dst, fn), 0o644)
        d = ref.readfasta(os.path.join(src, fn), False)
        assert d == orc.readfasta(os.path.join(src, fn))
        L = len(next(iter(d.values())))
        rec = l1_record(d, L)
        rec["cds"] = cds_record(d, L)
        out[fn] = rec
    with open(os.path.join(dst, "ATTRIBUTION"), "w") as fh:
        fh.write("file1.fa .. file10.fa are the example alignments shipped with\n"
                 "https://github.com/santiagosnchez/PolyFastA (example_theta_0.01/, GPL-3.0),\n"
                 "copied unmodified as test data.\n")
    dump("g1_examples.json", out)


# ---------------------------------------------------------- G2 synthetic grid
def g2():
    cases = []
    Ns = [2, 3, 4, 5, 7, 20, 50, 63, 64, 65, 100, 200, 255, 256, 300]
    Ls = [1, 3, 15, 16, 17, 239, 240, 241, 255, 256, 257, 1000, 1023, 1025]
    idx = 0
    for N in Ns:
        for L in Ls:
            for theta in (0.0, 0.01, 0.1):
                if theta == 0.0 and L not in (1, 16, 1000):
                    continue
                if N > 100 and L in (15, 17, 239, 241, 255, 257, 1023):
                    continue
                idx += 1
                g = {"seed": 1000 + idx, "aln": idx % 7, "N": N, "L": L, "theta": theta}
                d = gen_case(**g)
                cases.append({"gen": g, **l1_record(d, L)})
    for N, L, theta in ((50, 10000, 0.01), (200, 10000, 0.01), (50, 4096, 0.3), (20, 3000, 0.05)):
        idx += 1
        g = {"seed": 1000 + idx, "aln": 3, "N": N, "L": L, "theta": theta}
        d = gen_case(**g)
        cases.append({"gen": g, **l1_record(d, L)})
    # odd symbols injected into the matrix (upper-case only: L1 sees readfasta output)
    for N, L, frac, alphabet in ((20, 1000, 0.01, "-N"), (50, 1500, 0.002, "-N"),
                                 (7, 600, 0.05, "RYKMSWBDHVN-?"), (64, 700, 0.01, "-NRYU*."),
                                 (33, 512, 0.3, "-N"), (12, 300, 0.9, "-"), (100, 800, 0.001, "N"),
                                 (5, 400, 0.2, "@BDEFPQUVW")):
        idx += 1
        g = {"seed": 1000 + idx, "aln": 1, "N": N, "L": L, "theta": 0.02,
             "inject": {"seed": 77 + idx, "frac": frac, "alphabet": alphabet}}
        d = gen_case(**g)
        cases.append({"gen": g, **l1_record(d, L)})
    dump("g2_synthetic.json", cases)


# ------------------------------------------------ G3 FASTA syntax / symbols
G3_TEXTS = {
    "plain": ">a\nACGT\n>b\nACGA\n>c\nACGA\n",
    "lower_and_wrapped": ">s1 desc\nacgtac\ngtac\n>s2\nACGTAC\nGTAA\n>s3\nAcGtAc\nGtAn\n",
    "crlf": ">a\r\nACGT-\r\n>b\r\nACGTN\r\n>c\r\nAC-TN\r\n",
    "lone_cr": ">a\rACGT\r>b\rACGA\r",
    "blank_lines": ">a\n\nAC\n\nGT\n>b\nAC\nGA\n\n",
    "pre_header_text": "junk line\nmore junk\n>a\nACGT\n>b\nTCGT\n",
    "dup_header": ">a\nAAAA\n>b\nACGT\n>a\nACGA\n",
    "trailing_ws": ">a  \nAC GT \t\n>b\nAC GA\n>c\nACTGA\n",
    "no_final_newline": ">a\nACGT\n>b\nACGG",
    "gaps_iupac": ">a\nACGTRYKM--NN\n>b\nACGTRYKM-ANN\n>c\nACGARYKA-ANC\n>d\nTCGARYKA--NC\n",
    "one_acgt_allele": ">a\nAN\n>b\nA-\n>c\nAN\n",
    "unequal": ">a\nACGT\n>b\nACG\n",
    "empty_header_mid": ">a\nACGT\n>\nACGT\n>b\nACGA\n",
    "empty_header_last": ">a\nACGT\n>b\nACGA\n>\nACGT\n",
    "not_fasta": "ACGT\nACGT\n",
    "empty_file": "",
    "single_seq": ">only\nACGTACGT\n",
    "control_ws": ">a\nACGT\x0b\x0c\n>b\nACGA\x1c\x1d\x1e\x1f\n>c\nAC\x0cGT\n>d\nAC\x0cGA\n",
    "header_only": ">a\n>b\n",
    "gt_inside": ">a\nAC>GT\n>b\nAC>GA\n",
}


def g3():
    out = {}
    for name, text in G3_TEXTS.items():
        with tempfile.NamedTemporaryFile("w", suffix=".fa", delete=False, newline="") as fh:
            fh.write(text)
            path = fh.name
        msg = io.StringIO()
        with contextlib.redirect_stdout(msg):
            d = ref.readfasta(path, False)
        od = orc.readfasta(path)
        os.unlink(path)
        rec = {"text": text}
        if d == 1:
            assert od is None
            rec["records"] = None
        else:
            assert od == d
            rec["records"] = [[h, s] for h, s in d.items()]
            lens = [len(s) for s in d.values()]
            rec["same_length"] = all(x == lens[0] for x in lens)
            if rec["same_length"]:
                rec.update(l1_record(d, lens[0]))
                rec["row"] = capture(ref.print_result, d, lens[0], False, "", "a", name, "NA", True, 0, False).rstrip("\n")
        out[name] = rec
    dump("g3_fasta_syntax.json", out)


# ----------------------------------------------------- G4 codon-set tables
def _pack(s, n, start):
    """6-bit code: bits 0-2 synonymous positions, bits 3-5 non-synonymous."""
    code = 0
    for p in s:
        code |= 1 << (p - start)
    for p in n:
        code |= 8 << (p - start)
    return code


def g4():
    sense = list(orc.SENSE_CODONS)
    allc = sorted(orc.AMINO)
    out = {"sense": sense, "all": allc}
    pairs = []
    for a, b in itertools.combinations(sense, 2):
        s, n = ref.get_syn_nonsyn_cod_sites([[a, b], 0])
        s2, n2 = ref.get_syn_nonsyn_cod_sites([[b, a], 0])
        assert (sorted(s), sorted(n)) == (sorted(s2), sorted(n2))
        assert orc.get_syn_nonsyn_cod_sites([[a, b], 0]) == (s, n)
        pairs.append(_pack(s, n, 0))
    out["pairs"] = pairs                     # itertools.combinations(sense, 2) order
    triples = []
    for t in itertools.combinations(sense, 3):
        s, n = ref.get_syn_nonsyn_cod_sites([list(t), 0])
        assert orc.get_syn_nonsyn_cod_sites([list(t), 0]) == (s, n)
        triples.append(_pack(s, n, 0))
    out["triples"] = bytes(triples).hex()    # itertools.combinations(sense, 3) order
    rng = random.Random(20261003)
    sets = []
    for _ in range(12000):
        k = rng.randint(2, 10) if rng.random() < 0.9 else rng.randint(11, 40)
        pool = allc if rng.random() < 0.3 else sense
        cs = rng.sample(pool, k)
        s, n = ref.get_syn_nonsyn_cod_sites([list(cs), 0])
        assert orc.get_syn_nonsyn_cod_sites([list(cs), 0]) == (s, n)
        sets.append(["".join(cs), _pack(s, n, 0)])
    out["sets"] = sets
    out["syn_thirds"] = {c: round(ref.syncodfreq(c) * 3) for c in allc}
    for c in allc:
        assert ref.syncodfreq(c) == orc.syncodfreq(c)
    dump("g4_codon_sets.json", out)


# ----------------------------------------------------- G5 coding alignments
def g5():
    cases = []
    specs = [
        dict(seed=501, aln=0, N=50, L=9000, theta=0.01, coding=True),
        dict(seed=502, aln=1, N=50, L=9000, theta=0.05, coding=True),
        dict(seed=503, aln=2, N=20, L=1000, theta=0.02, coding=True),       # L % 3 == 1
        dict(seed=504, aln=0, N=20, L=1001, theta=0.02, coding=True),       # L % 3 == 2
        dict(seed=505, aln=0, N=8, L=720, theta=0.2, coding=True),
        dict(seed=506, aln=0, N=100, L=2400, theta=0.01, coding=True),
        dict(seed=507, aln=0, N=3, L=240, theta=0.1, coding=True),
        dict(seed=508, aln=0, N=64, L=480, theta=0.3, coding=True),
        dict(seed=509, aln=0, N=30, L=1500, theta=0.02, coding=True,
             inject={"seed": 9, "frac": 0.01, "alphabet": "-N"}),
        dict(seed=510, aln=0, N=30, L=1500, theta=0.02, coding=True,
             inject={"seed": 10, "frac": 0.2, "alphabet": "-NRY"}),
        dict(seed=511, aln=0, N=12, L=600, theta=0.0, coding=True),         # no variation
        dict(seed=512, aln=0, N=50, L=9000, theta=0.01, coding=False),      # random frame: many stops
        dict(seed=513, aln=0, N=2, L=300, theta=0.2, coding=True),
        dict(seed=514, aln=0, N=1, L=300, theta=0.2, coding=True),
        dict(seed=515, aln=0, N=40, L=2, theta=0.5, coding=False),          # single partial codon
        dict(seed=516, aln=0, N=40, L=241, theta=0.1, coding=True),
        dict(seed=517, aln=0, N=40, L=239, theta=0.1, coding=True),
    ]
    for g in specs:
        d = gen_case(**g)
        rec = {"gen": g, **l1_record(d, g["L"]), "cds": cds_record(d, g["L"])}
        cases.append(rec)
    tiny = {"a": "ATGTAAGGG", "b": "ATGTACGGG", "c": "ATGTACGGA"}
    cases.append({"literal": tiny, **l1_record(tiny, 9), "cds": cds_record(tiny, 9)})
    dump("g5_cds.json", cases)


# ------------------------------------------------------------ G6 populations
def g6():
    out = []
    g = dict(seed=601, aln=0, N=60, L=3000, theta=0.02, pops=2)
    d = gen_case(**g)
    # add pop10_* rows so that key "pop1" also matches "pop10_x" (substring semantics)
    d2 = dict(d)
    extra = gen_case(seed=602, aln=0, N=6, L=3000, theta=0.02)
    for i, (h, s) in enumerate(extra.items()):
        d2[f"pop10_x{i}"] = s
    for label, dd, keys in (("two_pops", d, ["pop1", "pop2"]),
                            ("substring", d2, ["pop1", "pop10", "pop2", "pop", "absent"])):
        rows = {}
        for key in keys:
            grp = {h: s for h, s in dd.items() if key in h}
            assert grp == orc.split_pop(dd, key)
            if not grp:
                rows[key] = None
                continue
            for cds in (False, True):
                for jc in (False, True):
                    row = capture(ref.print_result, grp, 3000, cds, "", "a", "x.fa", key, True, 0, jc).rstrip("\n")
                    assert_rows_close(orc.result_row(grp, 3000, cds, "x.fa", key, jc), row)
                    rows[f"{key}|cds={int(cds)}|jc={int(jc)}"] = row
        out.append({"label": label, "gen": g, "extra_gen": dict(seed=602, aln=0, N=6, L=3000, theta=0.02) if label == "substring" else None,
                    "rows": rows})
    dump("g6_pops.json", out)


# ------------------------------------------------------- G7 CLI transcripts
def run_cli(args, stdin_text=None, cwd=None):
    p = subprocess.run([sys.executable, os.path.join(REF_DIR, "PolyFastA.py")] + args,
                       input=None if stdin_text is None else stdin_text.encode(),
                       capture_output=True, cwd=cwd,
                       env={**os.environ, "PYTHONHASHSEED": "0"})
    # bytes -> str without newline translation (the progress line uses "\r")
    p.stdout = p.stdout.decode()
    p.stderr = p.stderr.decode()
    return p


def g7():
    ex = os.path.join(HERE, "example_theta_0.01")
    out = {}
    tmp = tempfile.mkdtemp()
    odd = os.path.join(tmp, "odd")
    os.makedirs(odd)
    for name in ("unequal", "not_fasta", "plain", "single_seq"):
        with open(os.path.join(odd, name + ".fa"), "w", newline="") as fh:
            fh.write(G3_TEXTS[name])
    novar = os.path.join(tmp, "novar.fa")
    with open(novar, "w") as fh:
        fh.write(">a\nATGAAACCC\n>b\nATGAAACCC\n")

    def rel(p):
        return p.replace(HERE + "/", "{GOLDEN}/").replace(tmp + "/", "{TMP}/")

    runs = {
        "file1": ["-f", f"{ex}/file1.fa"],
        "two_files": ["-f", f"{ex}/file2.fa", f"{ex}/file1.fa"],
        "file1_jc": ["-f", f"{ex}/file1.fa", "--jc"],
        "file1_cds": ["-f", f"{ex}/file1.fa", "--cds"],
        "file1_cds_jc": ["-f", f"{ex}/file1.fa", "-c", "--jc"],
        "file1_silent": ["-f", f"{ex}/file1.fa", "-s"],
        "odd_dir": ["-d", odd],
        "odd_dir_cds": ["-d", odd, "--cds"],
        "novar": ["-f", novar],
        "novar_cds": ["-f", novar, "--cds"],
    }
    for name, args in runs.items():
        p = run_cli(args)
        out[name] = {"args": [rel(a) for a in args], "stdout": p.stdout.replace(tmp + "/", "{TMP}/"), "returncode": p.returncode}
    # --dir on the examples: strip the ATTRIBUTION file by using a clean copy
    exdir = os.path.join(tmp, "examples")
    os.makedirs(exdir)
    for fn in os.listdir(ex):
        if fn.endswith(".fa"):
            shutil.copyfile(os.path.join(ex, fn), os.path.join(exdir, fn))
    for name, extra in (("dir", []), ("dir_jc", ["--jc"]), ("dir_cds", ["--cds"]), ("dir_cds_jc", ["--cds", "--jc"])):
        p = run_cli(["-d", exdir] + extra)
        out[name] = {"args": ["-d", "{EXAMPLES}"] + extra, "stdout": p.stdout, "returncode": p.returncode}
    # --pipe with and without a name
    text = open(f"{ex}/file3.fa").read()
    for name, args in (("pipe", ["--pipe"]), ("pipe_named", ["-i", "-n", "regionX"]), ("pipe_cds", ["-i", "-c"])):
        p = run_cli(args, stdin_text=text)
        out[name] = {"args": args, "stdin_file": "file3.fa", "stdout": p.stdout, "returncode": p.returncode}
    # --out: header written with 'w', rows appended; progress text on stdout
    outcsv = os.path.join(tmp, "res.csv")
    p = run_cli(["-f", f"{ex}/file1.fa", f"{ex}/file2.fa", "-o", outcsv])
    out["out_file"] = {"args": ["-f", "{GOLDEN}/example_theta_0.01/file1.fa", "{GOLDEN}/example_theta_0.01/file2.fa", "-o", "{TMP}/res.csv"],
                       "stdout": p.stdout.replace(outcsv, "{TMP}/res.csv"), "file": open(outcsv).read(), "returncode": p.returncode}
    p = run_cli(["-f", f"{ex}/file3.fa", "-o", outcsv, "-s"])
    out["out_file_silent_append"] = {"args": ["-f", "{GOLDEN}/example_theta_0.01/file3.fa", "-o", "{TMP}/res.csv", "-s"],
                                     "stdout": p.stdout, "file": open(outcsv).read(), "returncode": p.returncode}
    outcsv2 = os.path.join(tmp, "res2.csv")
    p = run_cli(["-d", odd, "--cds", "-o", outcsv2])
    out["out_file_odd_cds"] = {"args": ["-d", "{TMP}/odd", "--cds", "-o", "{TMP}/res2.csv"],
                               "stdout": p.stdout.replace(tmp + "/", "{TMP}/"), "file": open(outcsv2).read(), "returncode": p.returncode}
    # argument errors
    p = run_cli(["-f", f"{ex}/file1.fa", "-d", exdir])
    out["file_and_dir"] = {"args": ["-f", "x", "-d", "y"], "returncode": p.returncode, "stderr_tail": p.stderr.strip().splitlines()[-1]}
    p = run_cli(["-f", f"{ex}/file1.fa", "--pipe"], stdin_text=text)
    out["file_and_pipe"] = {"args": ["-f", "x", "--pipe"], "returncode": p.returncode, "stderr_tail": p.stderr.strip().splitlines()[-1]}
    out["_odd_texts"] = {n: G3_TEXTS[n] for n in ("unequal", "not_fasta", "plain", "single_seq")}
    out["_novar_text"] = ">a\nATGAAACCC\n>b\nATGAAACCC\n"
    shutil.rmtree(tmp)
    dump("g7_cli.json", out)


# ------------------------------------- G8 BASELINE.json configs at full size
def g8():
    """The CSV rows the reference prints for BASELINE.json configs 2, 3 and 4 at their
    full sizes (config 1 is G1/G7's file1.fa; config 5 is a directory of config-2
    alignments).  Config 3 goes through print_result per population, because
    main() crashes on --pops (PolyFastA.py:126)."""
    import time
    out = []
    cases = (
        ("config2_N50_L10k", dict(seed=802, aln=0, N=50, L=10000, theta=0.01), [None], False, (False, True)),
        ("config3_N200_L100k_pops_jc", dict(seed=803, aln=0, N=200, L=100000, theta=0.01, pops=2), ["pop1", "pop2"], False, (True,)),
        ("config4_cds_N50_L9k", dict(seed=804, aln=0, N=50, L=9000, theta=0.01, coding=True), [None], True, (False, True)),
    )
    for label, g, keys, cds, jcs in cases:
        d = gen_case(**g)
        rows = {}
        t0 = time.perf_counter()
        for key in keys:
            grp = d if key is None else {h: s for h, s in d.items() if key in h}
            pop = "NA" if key is None else key
            for jc in jcs:
                row = capture(ref.print_result, grp, g["L"], cds, "", "a", "x.fa", pop, True, 0, jc).rstrip("\n")
                assert_rows_close(orc.result_row(grp, g["L"], cds, "x.fa", pop, jc), row)
                rows[f"{pop}|jc={int(jc)}"] = row
        out.append({"label": label, "gen": g, "cds": cds, "pops": None if keys == [None] else keys, "rows": rows})
        print(f"  {label}: {time.perf_counter() - t0:.1f} s (reference + oracle)")
    dump("g8_baseline_configs.json", out)


# ------------------------------------------------------- G9 sequences with non-ASCII characters
def g9_texts():
    """Small alignments whose SEQUENCE lines hold non-ASCII characters: the reference reads text (`open(file, 'r')`), so a
    character is a site whatever its UTF-8 length, `.upper()` is Unicode's (it turns one sharp s into two letters) and
    `.rstrip()` strips Unicode white space (PolyFastA.py:238-245)."""
    rnd = random.Random(99)
    base = "".join(rnd.choice("ACGT") for _ in range(60))

    def mutate(s, k):
        s = list(s)
        for _ in range(k):
            i = rnd.randrange(len(s))
            s[i] = rnd.choice("ACGT")
        return "".join(s)

    rows = [mutate(base, k) for k in (0, 2, 3, 1, 4, 2)]
    t = {}
    # accented letters in some rows at some sites (lower case: upper() applies), names with non-ASCII too
    a = [list(r) for r in rows]
    a[1][7] = "\u00e9"
    a[2][7] = "\u00c9"      # the same letter after upper()
    a[3][20] = "\u00f1"
    a[4][20] = "\u00e9"
    a[5][33] = "\u0416"     # Cyrillic, two bytes
    a[0][41] = "\u20ac"     # three bytes
    a[2][41] = "\U0001F9EC"  # four bytes
    t["accents"] = "".join(f">pop{1 + i % 2}_s{i}_\u00fc\n{''.join(r[:30])}\n{''.join(r[30:]).lower()}\n" for i, r in enumerate(a))
    # sharp s in every row at the same site: every sequence grows by one character (61 sites)
    b = [list(r) for r in rows]
    for r in b:
        r[10] = "\u00df"
    t["sharp_s_all"] = "".join(f">s{i}\n{''.join(r).lower()}\n" for i, r in enumerate(b))
    # sharp s in one row only: unequal lengths after upper()
    c = [list(r) for r in rows]
    c[3][10] = "\u00df"
    t["sharp_s_one"] = "".join(f">s{i}\n{''.join(r).lower()}\n" for i, r in enumerate(c))
    # Unicode white space at the end of sequence lines (stripped), and inside one (kept: a site)
    d = ["".join(r) for r in rows]
    t["unicode_space"] = "".join(f">s{i}\n{r[:30]}\u00a0\n{r[30:45]}\u2003{r[45:]}\u3000 \n" for i, r in enumerate(d))
    return t


def g9():
    tmp = tempfile.mkdtemp()
    texts = g9_texts()
    for name, text in texts.items():
        with open(os.path.join(tmp, name + ".fa"), "w", encoding="utf-8", newline="") as fh:
            fh.write(text)
    out = {"_texts": texts, "runs": {}}
    runs = {"dir": ["-d", tmp], "dir_cds_jc": ["-d", tmp, "--cds", "--jc"]}
    for name, args in runs.items():
        p = subprocess.run([sys.executable, os.path.join(REF_DIR, "PolyFastA.py")] + args, capture_output=True,
                           env={**os.environ, "PYTHONHASHSEED": "0", "PYTHONUTF8": "1"})
        out["runs"][name] = {"args": [a.replace(tmp, "{TMP}") for a in args], "stdout": p.stdout.decode("utf-8").replace(tmp + "/", "{TMP}/"),
                             "returncode": p.returncode}
    # populations (the reference's own --pops fails under Python 3, `len(filter)`; its functions on the split it intends, as G6 does)
    d = ref.readfasta(os.path.join(tmp, "accents.fa"), False)
    seqlen = len(next(iter(d.values())))
    out["pops"] = {}
    for key in ("pop1", "pop2", "\u00fc", "absent"):
        grp = {h: s for h, s in d.items() if key in h}
        if not grp:
            out["pops"][key] = None
            continue
        for cds in (False, True):
            out["pops"][f"{key}|cds={int(cds)}"] = capture(ref.print_result, grp, seqlen, cds, "", "a", "accents.fa", key, True, 0, False).rstrip("\n")
    shutil.rmtree(tmp)
    dump("g9_unicode.json", out)


# ------------------------------------------------------- G10 the many-row regime
SYM = "ACGT" * 3 + "N-RY"


def g10_matrix(g):
    """the input of a G10 case from its generator parameters (tests/_cases.py: matrix_from_gen does the same)"""
    if g["kind"] == "many":
        m = synth.many_allele_matrix(g["seed"], g["N"], g["L"], g["alphabet"].encode())
    else:
        m = synth.synth_alignment(g["seed"], 0, g["L"], synth.make_params(g["N"], g["theta"]), coding=g.get("coding", False))
        if g.get("inject"):
            m = synth.inject_symbols(m, g["inject"]["seed"], g["inject"]["frac"], g["inject"]["alphabet"].encode())
    if g.get("subset"):
        m = np.ascontiguousarray(m[synth.row_subset(g["subset"]["seed"], g["N"], g["subset"]["frac"])])
    return m


def sparse(sfs):
    return sfs if isinstance(sfs, str) or sfs is None else {"len": len(sfs), "nz": [[i, c] for i, c in enumerate(sfs) if c]}


def g10():
    """Units of hundreds to 2^20 rows — the row-chunked kernels, the 128-site coding kernel at 101-200 rows, 64-bit H and
    the wide coding entries — pinned to the reference itself (it has no row limit: PolyFastA.py:252-261, :284-315):
    getvarsites S and position hash, integer H, getsfs (sparse), both polymorphism tuples, and for coding cases the
    getvarCDSsites 5-tuple and both CSV rows.  Both oracles are asserted against the reference here."""
    import time
    from oracle import c_oracle
    non = [dict(kind="many", seed=1001, N=601, L=700, alphabet="ACGT"),
           dict(kind="many", seed=1002, N=1000, L=333, alphabet=SYM),
           dict(kind="synth", seed=1003, N=2048, L=512, theta=0.05, inject=dict(seed=5, frac=0.002, alphabet="-N")),
           dict(kind="many", seed=1004, N=5000, L=333, alphabet=SYM),
           dict(kind="many", seed=1005, N=5000, L=400, alphabet="ACGT"),
           dict(kind="many", seed=1006, N=70000, L=64, alphabet=SYM),
           dict(kind="many", seed=1007, N=70000, L=64, alphabet="ACGT"),
           dict(kind="many", seed=1007, N=70000, L=64, alphabet="ACGT", subset=dict(seed=11, frac=0.95)),
           dict(kind="many", seed=1008, N=1 << 20, L=16, alphabet="ACGT"),
           dict(kind="many", seed=1009, N=1 << 20, L=16, alphabet="ACGT-N"),
           # beyond the 2^20 rows that were the ABI's limit until round 5
           dict(kind="many", seed=1010, N=(1 << 21) + 3, L=16, alphabet="ACGT-N"),
           dict(kind="many", seed=1010, N=(1 << 21) + 3, L=16, alphabet="ACGT-N", subset=dict(seed=13, frac=0.95))]
    cod = [dict(kind="many", seed=1101, N=128, L=333, alphabet="ACGT"),
           dict(kind="synth", seed=1102, N=160, L=333, theta=0.05, coding=True),
           dict(kind="many", seed=1103, N=200, L=333, alphabet=SYM),
           dict(kind="synth", seed=1104, N=200, L=999, theta=0.03, coding=True),
           dict(kind="many", seed=1105, N=601, L=333, alphabet="ACGT"),
           dict(kind="synth", seed=1106, N=2049, L=333, theta=0.05, coding=True),
           dict(kind="many", seed=1107, N=2049, L=333, alphabet=SYM),
           dict(kind="many", seed=1108, N=5000, L=333, alphabet="ACGT"),
           dict(kind="synth", seed=1109, N=5000, L=333, theta=0.05, coding=True, inject=dict(seed=6, frac=0.001, alphabet="-N")),
           dict(kind="many", seed=1110, N=70000, L=66, alphabet=SYM),
           dict(kind="many", seed=1111, N=70000, L=66, alphabet="ACGT"),
           dict(kind="many", seed=1111, N=70000, L=66, alphabet="ACGT", subset=dict(seed=12, frac=0.95)),
           dict(kind="many", seed=1112, N=1 << 20, L=15, alphabet="ACGT"),
           dict(kind="many", seed=1113, N=(1 << 21) + 3, L=15, alphabet="ACGT")]
    out = []
    for cds, gens in ((False, non), (True, cod)):
        for g in gens:
            t0 = time.perf_counter()
            m = g10_matrix(g)
            n, L = m.shape
            d = synth.to_dict([f"s{i}" for i in range(n)], m)
            rec = l1_record(d, L, check_oracle=n * L <= 5_000_000)      # (the pure-Python oracle: the cases it finishes in seconds)
            rec["sfs"] = sparse(rec["sfs"])
            o = c_oracle.scan(m, cds=cds, want_sites=True)
            assert (o["S"], o["H"]) == (rec["S"], rec["H"]), (g, o["S"], o["H"], rec["S"], rec["H"])
            assert sha1_ints(np.flatnonzero(o["h_site"])) == rec["pos_sha1"]
            if rec["S"]:
                if rec["sfs"] == "IndexError":
                    assert o["sfs_bad"] > 0
                else:
                    assert o["sfs_bad"] == 0 and sparse(o["sfs"].tolist()) == rec["sfs"]
            case = {"gen": g, "coding": cds, **rec}
            if cds:
                cs, s, nn, nstops, missing = ref.getvarCDSsites(d, L)
                assert np.flatnonzero(o["labels"] == 1).tolist() == sorted(s) and np.flatnonzero(o["labels"] == 2).tolist() == sorted(nn)
                assert (o["nstops"], 3 * o["missing_cols"]) == (nstops, missing) and close(o["count_syn"], cs, 1e-12)
                rows = {}
                for jc in (False, True):
                    row = capture(ref.print_result, d, L, True, "", "a", "x.fa", "NA", True, 0, jc).rstrip("\n")
                    rows["row_jc" if jc else "row"] = row
                    assert_rows_close(orc.row_from_counts("x.fa", "NA", n, L, o, True, jc), row)
                # H_S / H_N: the integer H of the variable columns at the S / N positions (PolyFastA.py:169-173)
                pos, var = ref.getvarsites(d, L)
                at = {p: i for i, p in enumerate(pos)}
                hs = int_H([var[at[p]] for p in s if p in at])
                hn = int_H([var[at[p]] for p in nn if p in at])
                assert (o["H_S"], o["H_N"]) == (hs, hn)
                case["cds"] = {"count_syn": cs, "S_sha1": sha1_ints(sorted(s)), "N_sha1": sha1_ints(sorted(nn)), "n_S": len(s), "n_N": len(nn),
                               "seg_S": sum(1 for p in s if p in at), "seg_N": sum(1 for p in nn if p in at), "H_S": hs, "H_N": hn,
                               "nstops": nstops, "missing": missing, **rows}
                assert (o["seg_S"], o["seg_N"]) == (case["cds"]["seg_S"], case["cds"]["seg_N"])
            else:
                for jc, key in ((False, "poly"), (True, "poly_jc")):
                    got = orc.polymorphism_from_counts(o["S"], o["H"], n, L, jc)
                    assert all(close(x, y) for x, y in zip(got, rec[key])), (got, rec[key])
            out.append(case)
            print(f"  G10 {'coding' if cds else 'plain '} N={n} L={L} S={rec['S']} H={rec['H']} ({time.perf_counter() - t0:.1f} s)", flush=True)
            del d
    dump("g10_many_rows.json", out)


# ------------------------------------------------------- G11 a directory entry that cannot be read
def g11():
    """`--dir` over a directory that holds a subdirectory between two alignments: the reference prints the header and every row
    before it, then dies at its open() (PolyFastA.py:104-108, :239) — stdout, the --out file, the exit code and the exception."""
    ex = os.path.join(HERE, "example_theta_0.01")
    tmp = tempfile.mkdtemp()
    d = os.path.join(tmp, "mixed")
    os.makedirs(os.path.join(d, "file2_is_a_directory"))
    for fn in ("file1.fa", "file10.fa", "file3.fa"):
        shutil.copyfile(os.path.join(ex, fn), os.path.join(d, fn))
    out = {"layout": {"files": ["file1.fa", "file10.fa", "file3.fa"], "subdirectory": "file2_is_a_directory"}}
    p = run_cli(["-d", d])
    out["dir"] = {"args": ["-d", "{TMP}/mixed"], "stdout": p.stdout, "returncode": p.returncode, "stderr_tail": p.stderr.strip().splitlines()[-1].replace(tmp, "{TMP}")}
    csv = os.path.join(tmp, "res.csv")
    p = run_cli(["-d", d, "--cds", "--jc", "-o", csv, "-s"])
    out["dir_cds_jc_out_silent"] = {"args": ["-d", "{TMP}/mixed", "--cds", "--jc", "-o", "{TMP}/res.csv", "-s"], "stdout": p.stdout, "file": open(csv).read(),
                                    "returncode": p.returncode, "stderr_tail": p.stderr.strip().splitlines()[-1].replace(tmp, "{TMP}")}
    shutil.rmtree(tmp)
    dump("g11_unreadable_entry.json", out)


# ------------------------------------------------------- G12 two-decimal rounding ties of sites_S / sites_N
G12_CODONS = ["GCT", "GCC", "AAA", "AAG", "CTA", "TTA", "GAT", "CGA", "AGA", "TCT", "ACG", "GGG", "CAT", "ATT", "TGG", "GTA"]


def g12_matrix(g, aln=None):
    """the input of a G12 case from its recipe (tests/_cases.py: g12_matrix does the same)"""
    m = synth.synth_alignment(g["seed"], g["aln"] if aln is None else aln, g["L"], synth.make_params(g["N"], g["theta"]), coding=True)
    if g.get("inject"):
        m = synth.inject_symbols(m, g["inject"]["seed"], g["inject"]["frac"], g["inject"]["alphabet"].encode())
    for pl in g.get("plant", []):
        rows = None if pl.get("rows") is None else np.arange(pl["rows"][0], g["N"], pl["rows"][1])
        m = synth.plant_codons(m, pl["col"], pl["codons"], rows)
    return m


def term_values(ts):
    """every value sum(perm) / k can take over the orders of the float terms `ts` (the reference sums a Python set's order)"""
    vals = set()
    for perm in set(itertools.permutations(ts)) if len(ts) <= 8 else [tuple(ts), tuple(reversed(ts)), tuple(sorted(ts)), tuple(sorted(ts, reverse=True))]:
        t = 0
        for x in perm:
            t = t + x
        vals.add(t / len(ts))
    return vals


def count_syn_range(d, seqlen):
    """(lowest, highest) count_syn over the summation orders of the columns with three or more codons: when round(x, 2) of
    both is the same text, the reference's printed digits do not depend on its hash seed"""
    import re
    lo = hi = 0.0
    for cp in range(0, seqlen, 3):
        cods = sorted(set(s[cp:cp + 3] for s in d.values()))
        clean = [c for c in cods if re.match("^[AGTC][AGTC][AGTC]$", c)]
        if not clean:
            continue
        vals = term_values([ref.syncodfreq(c) for c in clean])
        lo += min(vals)
        hi += max(vals)
    return lo, hi


def is_tie(x):
    y = 100.0 * x
    return abs(y - int(y) - 0.5) < 1e-6


def g12():
    """Coding alignments whose exact synonymous-site sum is a two-decimal tie (....375: a codon column with 8 or 16 distinct
    codons puts the factor 8 into the denominator): the reference accumulates count_syn column by column in floating point
    (PolyFastA.py:307) and prints round(x, 2) (:178), so the last bits of that accumulation decide the printed digit.  The
    alignments are this repo's generator's (some found as they come, most with one planted many-codon column, `aln` searched
    until the sum is a tie); frozen: count_syn (repr) and the reference's CSV rows, with and without --jc, per population key.
    Cases whose printed digits could depend on the interpreter's hash seed (the order in which a Python set of three or
    more codons is summed) are left out and counted."""
    import time
    from fractions import Fraction
    from oracle import c_oracle
    P8, P16 = G12_CODONS[:8], G12_CODONS
    specs = [dict(seed=1201 + i, N=50, L=9000, theta=0.01, plant=[dict(col=700 + 111 * i, codons=P8)]) for i in range(6)]
    specs += [dict(seed=1211 + i, N=50, L=9000, theta=0.01, pops=2,
                   plant=[dict(col=500 + 97 * i, codons=P8, rows=[0, 2]), dict(col=1500 + 97 * i, codons=P8[::-1], rows=[1, 2])]) for i in range(4)]
    specs += [dict(seed=1221, N=200, L=9000, theta=0.01, plant=[dict(col=1234, codons=P16)]),
              dict(seed=1222, N=200, L=9000, theta=0.01, plant=[dict(col=77, codons=P8), dict(col=2999, codons=P8[1:] + ["CCC"])]),
              dict(seed=1223, N=1000, L=3000, theta=0.01, plant=[dict(col=501, codons=P8)]),
              dict(seed=1224, N=50, L=9000, theta=0.01, inject=dict(seed=24, frac=0.002, alphabet="-N"), plant=[dict(col=1000, codons=P8)]),
              dict(seed=1225, N=50, L=9000, theta=0.02, inject=dict(seed=25, frac=0.001, alphabet="-NRY"), plant=[dict(col=2000, codons=P8)]),
              dict(seed=1226, N=50, L=9001, theta=0.01, plant=[dict(col=1500, codons=P8)]),                       # L % 3 == 1: 3 missing sites
              dict(seed=1227, N=50, L=9000, theta=0.01, plant=[dict(col=1600, codons=P8[:7] + ["TAA"])]),         # a stop among the eight
              dict(seed=1228, N=64, L=4800, theta=0.03, plant=[dict(col=800, codons=P8)]),
              dict(seed=404, aln=569, N=50, L=9000, theta=0.01),                                                 # as the generator makes them (bench seed)
              dict(seed=404, aln=1894, N=50, L=9000, theta=0.01)]
    # more of the generator's own: the first ties among the alignments of one seed
    found = 0
    for aln in range(6000):
        g = dict(seed=1231, aln=aln, N=50, L=9000, theta=0.01)
        if is_tie(c_oracle.scan(g12_matrix(g), cds=True)["count_syn"]):
            specs.append(g)
            found += 1
            if found == 3:
                break
    out, dropped = [], 0
    for g in specs:
        t0 = time.perf_counter()
        keys = [None] if not g.get("pops") else ["pop1", "pop2"]
        hdr = synth.headers_for(g["N"], g.get("pops", 0))
        if "aln" not in g:      # search the alignment index until the (first key's) sum is a tie
            for aln in range(2000):
                m = g12_matrix(g, aln)
                rows = None if keys[0] is None else np.array([i for i, h in enumerate(hdr) if keys[0] in h])
                if is_tie(c_oracle.scan(m if rows is None else m[rows], cds=True)["count_syn"]):
                    g["aln"] = aln
                    break
            else:
                raise SystemExit(f"G12: no tie found for {g}")
        m = g12_matrix(g)
        d = synth.to_dict(hdr, m)
        L = g["L"]
        case = {"gen": g, "keys": {}}
        ok = True
        for key in keys:
            grp = d if key is None else {h: s for h, s in d.items() if key in h}
            cs = ref.getvarCDSsites(grp, L)[0]
            lo, hi = count_syn_range(grp, L)
            assert lo <= cs <= hi
            tie = is_tie(cs)
            free = round(lo, 2) == round(hi, 2) and round((L - ref.getvarCDSsites(grp, L)[4]) - lo, 2) == round((L - ref.getvarCDSsites(grp, L)[4]) - hi, 2)
            if tie and not free:
                ok = False
            # the exact sum, for the record: a tie is a multiple of 1/200 that is not one of 1/100
            sub = m if key is None else m[[i for i, h in enumerate(hdr) if key in h]]
            o = c_oracle.scan(sub, cds=True)
            rows = {}
            for jc in (False, True):
                row = capture(ref.print_result, grp, L, True, "", "a", "x.fa", key or "NA", True, 0, jc).rstrip("\n")
                rows["row_jc" if jc else "row"] = row
                assert_rows_close(orc.row_from_counts("x.fa", key or "NA", len(grp), L, o, True, jc), row)
            case["keys"][key or "NA"] = {"N": len(grp), "count_syn": repr(cs), "count_syn_order_free": lo == hi, "tie": tie,
                                         "c_oracle_equal": o["count_syn"] == cs, **rows}
        if not ok:
            dropped += 1
            print(f"  G12 dropped (printed digits depend on the set order): {g}")
            continue
        assert any(k["tie"] for k in case["keys"].values()), g
        out.append(case)
        print(f"  G12 {g} ties={[k for k, v in case['keys'].items() if v['tie']]} ({time.perf_counter() - t0:.1f} s)", flush=True)
    n_ties = sum(1 for c in out for v in c["keys"].values() if v["tie"])
    assert n_ties >= 20, n_ties
    dump("g12_rounding_ties.json", {"dropped_hash_seed_dependent": dropped, "cases": out})


# ------------------------------------------------------- G13 odd inputs through the command line
def ref_cli(args, stdin_bytes=None):
    p = subprocess.run([sys.executable, os.path.join(REF_DIR, "PolyFastA.py")] + args, input=stdin_bytes, capture_output=True,
                       env={**os.environ, "PYTHONHASHSEED": "0", "PYTHONUTF8": "1"})
    return p


def g13():
    """What the reference's COMMAND LINE prints for the awkward inputs the other families pin at function level:
      (a) the twenty G3 texts as the files of one directory, `-d` and `-d --cds --jc` (zero-length sequences, control white
          space, `>` inside a sequence line, empty headers ...): stdout and exit code;
      (b) the same path twice after `-f` (PolyFastA.py:104: sorted(), no de-duplication);
      (c) population keys at their edges, as G6 does it (print_result per key on the split the reference intends, :123-134):
          an empty key (`-p pop1,` / `-p ,`: '' is in every header), a repeated key, a key equal to a whole header, and an
          alignment of zero-length sequences with keys, coding and not;
      (d) a file that is not valid UTF-8 between two good ones: the rows before it, then UnicodeDecodeError, exit code 1
          (:239-240) — `-d`, `-d --cds --jc -o out.csv -s`, the byte in a header line instead, and `--pipe`."""
    out = {}
    tmp = tempfile.mkdtemp()

    def rel(t):
        return t.replace(tmp + "/", "{TMP}/")

    # (a)
    d3 = os.path.join(tmp, "g3")
    os.makedirs(d3)
    for name, text in G3_TEXTS.items():
        with open(os.path.join(d3, name + ".fa"), "w", newline="") as fh:
            fh.write(text)
    out["g3_dir"] = {}
    for label, extra in (("plain", []), ("cds_jc", ["--cds", "--jc"]), ("cds", ["--cds"])):
        p = ref_cli(["-d", d3] + extra)
        out["g3_dir"][label] = {"args": ["-d", "{TMP}/g3"] + extra, "stdout": rel(p.stdout.decode()), "returncode": p.returncode,
                                "stderr_tail": (p.stderr.decode().strip().splitlines() or [""])[-1]}
    # the reference dies under --cds at the first file whose non-synonymous site count is zero while a site varies
    # (polymorphism() divides by it, :509-517): the same directory without the files it dies on, so that the ones behind them count too
    removed = []
    while True:
        p = ref_cli(["-d", d3, "--cds", "--jc"])
        if p.returncode == 0:
            break
        last = p.stdout.decode().rstrip("\n").split("\n")[-1]
        name = last.split(": ")[-1] if last.startswith("#") else None
        nxt = sorted(os.listdir(d3))
        victim = name if name in nxt else nxt[nxt.index(last.split(",")[0]) + 1]
        removed.append({"file": victim, "error": p.stderr.decode().strip().splitlines()[-1]})
        os.unlink(os.path.join(d3, victim))
    out["g3_dir"]["cds_jc_survivors"] = {"args": ["-d", "{TMP}/g3", "--cds", "--jc"], "removed": removed, "stdout": rel(p.stdout.decode()), "returncode": 0}
    # (b)
    ex = os.path.join(HERE, "example_theta_0.01")
    p = ref_cli(["-f", f"{ex}/file2.fa", f"{ex}/file1.fa", f"{ex}/file2.fa"])
    out["same_path_twice"] = {"args": ["-f", "{GOLDEN}/example_theta_0.01/file2.fa", "{GOLDEN}/example_theta_0.01/file1.fa",
                                       "{GOLDEN}/example_theta_0.01/file2.fa"], "stdout": p.stdout.decode(), "returncode": p.returncode}
    # (c)
    g = dict(seed=601, aln=0, N=60, L=3000, theta=0.02, pops=2)
    d = gen_case(**g)
    first = next(iter(d))
    edge = []
    for label, typed in (("trailing_comma", "pop1,"), ("only_commas", ","), ("repeated", "pop1,pop1"), ("whole_header", first + ",pop2"),
                         ("empty_between", "pop2,,pop1")):
        rows = []
        for key in typed.split(","):
            grp = {h: s for h, s in d.items() if key in h}
            assert grp == orc.split_pop(d, key) and grp
            per = {}
            for cds in (False, True):
                for jc in (False, True):
                    row = capture(ref.print_result, grp, 3000, cds, "", "a", "x.fa", key, True, 0, jc).rstrip("\n")
                    assert_rows_close(orc.result_row(grp, 3000, cds, "x.fa", key, jc), row)
                    per[f"cds={int(cds)}|jc={int(jc)}"] = row
            rows.append({"key": key, "N": len(grp), **per})
        edge.append({"label": label, "typed": typed, "rows": rows})
    out["pops_edges"] = {"gen": g, "cases": edge}
    zero = {"pop1_a": "", "pop2_b": "", "pop1_c": ""}
    zrows = []
    for key in ("pop1", "", "pop2", "absent"):
        grp = {h: s for h, s in zero.items() if key in h}
        if not grp:
            zrows.append({"key": key, "N": 0})
            continue
        per = {}
        for cds in (False, True):
            per[f"cds={int(cds)}|jc=0"] = capture(ref.print_result, grp, 0, cds, "", "a", "zero.fa", key, True, 0, False).rstrip("\n")
        zrows.append({"key": key, "N": len(grp), **per})
    out["pops_zero_length"] = {"text": "".join(f">{h}\n" for h in zero), "typed": "pop1,,pop2,absent", "rows": zrows}
    # (d)
    good1 = open(f"{ex}/file1.fa", "rb").read()
    good3 = open(f"{ex}/file3.fa", "rb").read()
    lines = open(f"{ex}/file2.fa", "rb").read().split(b"\n")
    seq_bad = list(lines)
    seq_bad[5] = seq_bad[5][:40] + b"\x80" + seq_bad[5][41:]            # a lone continuation byte inside the third record's sequence line
    head_bad = list(lines)
    head_bad[2] = head_bad[2] + b"\xff"                                   # ... or in a header line
    # ... or in the sequence of a record that a later record of the same name replaces (readfasta keeps the last one, :234, :243)
    dup_bad = list(lines)
    dup_bad[3] = dup_bad[3][:10] + b"\xc3" + dup_bad[3][11:]          # a lead byte without its continuation, in the second record
    dup_bad = dup_bad[:4] + [dup_bad[2], lines[3]] + dup_bad[4:]      # the same header again, with the clean sequence
    und = {}
    for label, blob in (("in_sequence", b"\n".join(seq_bad)), ("in_header", b"\n".join(head_bad)), ("in_replaced_record", b"\n".join(dup_bad))):
        dd = os.path.join(tmp, "und_" + label)
        os.makedirs(dd)
        for fn, data in (("a_good.fa", good1), ("b_bad.fa", blob), ("c_good.fa", good3)):
            with open(os.path.join(dd, fn), "wb") as fh:
                fh.write(data)
        runs = {}
        p = ref_cli(["-d", dd])
        runs["dir"] = {"args": ["-d", "{TMP}/und_" + label], "stdout": rel(p.stdout.decode()), "returncode": p.returncode,
                       "stderr_tail": p.stderr.decode().strip().splitlines()[-1]}
        csv = os.path.join(tmp, f"und_{label}.csv")
        p = ref_cli(["-d", dd, "--cds", "--jc", "-o", csv, "-s"])
        runs["dir_cds_jc_out_silent"] = {"args": ["-d", "{TMP}/und_" + label, "--cds", "--jc", "-o", "{TMP}/und_" + label + ".csv", "-s"],
                                         "stdout": rel(p.stdout.decode()), "file": open(csv).read(), "returncode": p.returncode,
                                         "stderr_tail": p.stderr.decode().strip().splitlines()[-1]}
        p = ref_cli(["--pipe"], stdin_bytes=blob)
        runs["pipe"] = {"args": ["--pipe"], "stdout": p.stdout.decode(), "returncode": p.returncode, "stderr_tail": (p.stderr.decode().strip().splitlines() or [""])[-1]}
        und[label] = {"bad_file_hex": blob.hex(), "runs": runs}
    out["undecodable"] = {"layout": ["a_good.fa = example file1.fa", "b_bad.fa", "c_good.fa = example file3.fa"], **und}
    # (e) the command line's way with paths: a trailing slash, a relative directory, names with blanks / non-ASCII letters / a leading
    # dot, an empty directory, a path that does not exist, a directory after -f, empty standard input, -n without --pipe, -p without
    # a value and with an empty one, --out with such names in its progress chatter
    md = os.path.join(tmp, "misc")
    d1 = os.path.join(md, "d1")
    os.makedirs(d1)
    os.makedirs(os.path.join(md, "empty"))
    files = {"a.fa": open(f"{ex}/file1.fa").read(), "b file.fa": G3_TEXTS["gaps_iupac"], "\u00dcn\u00ef.fa": G3_TEXTS["lower_and_wrapped"],
             ".hidden.fa": G3_TEXTS["plain"], "not_fasta.txt": G3_TEXTS["not_fasta"], "Z.fa": G3_TEXTS["crlf"]}
    for fn, text in files.items():
        with open(os.path.join(d1, fn), "w", newline="", encoding="utf-8") as fh:
            fh.write(text)
    f3 = open(f"{ex}/file3.fa", "rb").read()
    runs = [("trailing_slash", ["-d", d1 + "/"], None, None),
            ("cds_silent", ["-d", d1, "--cds", "-s"], None, None),
            ("empty_dir", ["-d", os.path.join(md, "empty")], None, None),
            ("missing_file", ["-f", os.path.join(d1, "a.fa"), os.path.join(md, "nope.fa")], None, None),
            ("directory_as_file", ["-f", d1], None, None),
            ("pipe_empty_stdin", ["--pipe"], b"", None),
            ("pipe_pops_without_value", ["--pipe", "-n", "reg", "-p"], f3, None),
            ("name_without_pipe", ["-f", os.path.join(d1, "a.fa"), "-n", "ignored"], None, None),
            ("pops_empty_string", ["-f", os.path.join(d1, "a.fa"), "-p", ""], None, None),
            ("relative_dir", ["-d", "d1"], None, md),
            ("out_with_odd_names", ["-d", d1, "-o", os.path.join(md, "res.csv")], None, None)]
    misc = {}
    for name, args, stdin, cwd in runs:
        p = subprocess.run([sys.executable, os.path.join(REF_DIR, "PolyFastA.py")] + args, input=stdin, capture_output=True, cwd=cwd,
                           env={**os.environ, "PYTHONHASHSEED": "0", "PYTHONUTF8": "1"})
        rec = {"args": [rel(a) for a in args], "cwd": None if cwd is None else rel(cwd + "/").rstrip("/"), "stdin_hex": None if stdin is None else stdin.hex(),
               "stdout": rel(p.stdout.decode()), "returncode": p.returncode, "stderr_tail": rel((p.stderr.decode().strip().splitlines() or [""])[-1])}
        if "-o" in args:
            rec["file"] = open(args[args.index("-o") + 1]).read()
        misc[name] = rec
    out["cli_paths"] = {"files": files, "runs": misc}
    # (f) more awkward file contents (bytes, written as they are): a byte-order mark in front of the first header (the reference reads
    # utf-8, not utf-8-sig: the mark makes that line text before the first header and the record is lost), a NUL inside a sequence
    # (a symbol like any other), a header of blanks only (rstrip leaves '': its lines are dropped, and a file that ENDS on it is not
    # FASTA), a lone '>', a header without a line end, tabs behind a header and inside a sequence, a form feed between records
    # (str.splitlines would split there; file iteration does not), thousands of sites on one line next to the same wrapped at 7
    rnd = random.Random(131)
    long_a = "".join(rnd.choice("ACGT") for _ in range(5000))
    long_b = "".join(c if rnd.random() > 0.01 else rnd.choice("ACGT") for c in long_a)
    more = {
        "bom.fa": b"\xef\xbb\xbf>a\nACGT\n>b\nACGA\n>c\nACGG\n",
        "nul_inside.fa": b">a\nAC\x00T\n>b\nACGT\n>c\nAC\x00T\n",
        "blank_header.fa": b">a\nACGT\n>   \nTTTT\n>b\nACGA\n",
        "ends_on_blank_header.fa": b">a\nACGT\n>b\nACGA\n> \t \nTTTT\n",
        "lone_gt.fa": b">\n",
        "header_no_eol.fa": b">a\nACGT\n>b",
        "tabs.fa": b">a\t\t\nAC\tGT\n>b \t\nAC\tGA\n",
        "form_feed.fa": b">a\nACGT\x0c>b\nACGA\n>c\nACGA\x0cT\n",
        "long_lines.fa": (">a\n" + long_a + "\n>b\n" + "\n".join(long_b[i:i + 7] for i in range(0, 5000, 7)) + "\n").encode(),
    }
    dm = os.path.join(tmp, "more")
    os.makedirs(dm)
    for fn, blob in more.items():
        with open(os.path.join(dm, fn), "wb") as fh:
            fh.write(blob)
    mruns = {}
    for label, extra in (("plain", []), ("cds_jc", ["--cds", "--jc"])):
        removed = []
        while True:
            p = ref_cli(["-d", dm] + extra)
            if p.returncode == 0:
                break
            last = p.stdout.decode().rstrip("\n").split("\n")[-1]
            nxt = sorted(os.listdir(dm))
            victim = last.split(": ")[-1] if last.startswith("#") and last.split(": ")[-1] in nxt else nxt[nxt.index(last.split(",")[0]) + 1]
            removed.append({"file": victim, "error": p.stderr.decode().strip().splitlines()[-1], "stdout_until_then": rel(p.stdout.decode())})
            os.unlink(os.path.join(dm, victim))
        mruns[label] = {"args": ["-d", "{TMP}/more"] + extra, "removed": removed, "stdout": rel(p.stdout.decode()), "returncode": 0}
    out["more_texts"] = {"files_hex": {k: v.hex() for k, v in more.items()}, "runs": mruns}
    # (e) carriage returns on standard input: a line end in a file (open(file, 'r'): universal newlines), but sys.stdin splits at
    #     \n only — a piped file with bare-CR line ends is ONE header line (one sequence, no sites), \r\r\n ends a line like \r\n, and
    #     a bare CR inside a sequence line is one of its characters (tools/g14_explore.py ran into the first)
    cr = {
        "bare_cr": b">a\rACGT\r>b\rACGA\r>c\rACGA\r",
        "crlf": b">a\r\nACGT\r\n>b\r\nACGA\r\n>c\r\nACGA\r\n",
        "cr_cr_lf": b">a\r\r\nACGT\r\r\n>b\r\r\nACGA\r\r\n>c\r\r\nAGGA\r\r\n",
        "cr_inside_sequence": b">a\nAC\rGT\n>b\nAC\rGA\n>c\nACTGA\n",
        "cr_inside_header": b">a\rx\nACGTAC\n>b\ry\nACGAAC\n>c\nACGAAT\n",
    }
    pruns = []
    for label, blob in cr.items():
        for extra in ([], ["--cds"], ["-n", "gene 1", "--jc"]):
            p = ref_cli(["--pipe", "-s"] + extra, stdin_bytes=blob)
            pruns.append({"label": label, "args": ["--pipe", "-s"] + extra, "returncode": p.returncode, "stdout": p.stdout.decode(),
                          "stderr_tail": (p.stderr.decode().strip().splitlines() or [""])[-1] if p.returncode else None})
        path = os.path.join(tmp, label + ".fa")
        with open(path, "wb") as fh:
            fh.write(blob)
        p = ref_cli(["-f", path, "-s"])
        pruns.append({"label": label, "args": ["-f", "{TMP}/" + label + ".fa", "-s"], "returncode": p.returncode, "stdout": rel(p.stdout.decode()),
                      "stderr_tail": (p.stderr.decode().strip().splitlines() or [""])[-1] if p.returncode else None})
    out["carriage_returns"] = {"texts_hex": {k: v.hex() for k, v in cr.items()}, "runs": pruns}
    # (f) files in which no population key is found (tools/g14_explore.py ran into them): what the reference prints for them —
    #     the comment lines of main() (:110-140), from its functions on the split it intends, as (c) — hangs on the text reader alone:
    #     a length of 3 CHARACTERS that is 4 bytes, sequences equally long in bytes but not in characters (an upper-cased sharp s), blanks
    #     and control white space at line ends that do not count as sites; and one file in which one key is found and one is not
    nk = {
        "three_characters_four_bytes.fa": b">s0\tdesc\nA\xc3\xa9A\n",
        "equal_in_bytes_only.fa": b">s0\nC\xc7\x86AAGT\r>s1\nCT\xc7\x86AGG\r>s0\r\nCTAA\xc3\x9fT\r",
        "blanks_at_line_ends.fa": b">s0,x=0\nG\t\r\nG \nC \nT\x1f\nT\n\nA\r\nC\x0c\n>s1,x=1\nG\t\r\nG \nC \nT\x1f\nT\n\nC\r\nC\n",
        "one_key_of_two.fa": b">a_pop1\nACG \nTAC\t\n>b_pop1\nACG\nTAA\x1f\n>c_other\nACGTAA\n",
    }
    dn = os.path.join(tmp, "nokey")
    os.makedirs(dn)
    for fn, blob in nk.items():
        with open(os.path.join(dn, fn), "wb") as fh:
            fh.write(blob)
    keys = ["pop2", "nopop", "pop1"]
    nruns = []
    for cds, jc in ((False, False), (True, False), (True, True)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            for fn in sorted(nk):
                dd = ref.readfasta(os.path.join(dn, fn), False)
                lens = [len(x) for x in dd.values()]
                if any(x != lens[0] for x in lens):
                    print(f"# Sequences do not have the same length: {fn}")
                    continue
                if cds and lens[0] % 3:
                    print(f"# CDS sequence length is not a multiple of 3: {fn}")
                for pop in keys:
                    grp = [k for k in dd if pop in k]
                    if not grp:
                        print(f"# Pop {pop} string was not found in fasta headers.")
                        continue
                    ref.print_result({k: dd[k] for k in grp}, lens[0], cds, "", "a", fn, pop, True, 0, jc)
        nruns.append({"args": ["-d", "{TMP}/nokey", "-s", "-p", ",".join(keys)] + (["--cds"] if cds else []) + (["--jc"] if jc else []), "stdout": buf.getvalue()})
    out["no_key_found"] = {"files_hex": {k: v.hex() for k, v in nk.items()}, "runs": nruns}
    shutil.rmtree(tmp)
    dump("g13_cli_odd_inputs.json", out)


# ------------------------------------------------------- G14 random directories through the command line
G14_SEEDS = ([("mixed", sd) for sd in list(range(1, 49)) + [590]]      # 590: the directory in which tools/g14_explore.py ran into a tie the reference itself does not fix (G15)
             + [("wild", sd) for sd in list(range(1, 25)) + [56, 433, 485, 506]]      # the last four: files in which no population key is found (G13 no_key_found) — here without -p
             + [("big", sd) for sd in range(1, 7)])


def g14():
    """The reference's transcript for 83 random directories (polyfasta_amd.synth: 49 of mixed_directory — small awkward files —, 28
    of wild_directory — odd palettes, blanks and control white space in and after lines, mixed line ends, stray lines, empty files,
    names with blanks and commas —, 6 of big_directory — 65 to 300 sequences: the other kernels; the
    files are made again from the seed, only the transcript is frozen): `-d DIR -s` with the seed's flags (plain, --jc, --cds,
    --cds --jc; the reference's -p dies on every input under Python 3, G6 and G13 pin populations per key) — stdout, exit code, and the last line of stderr where the reference dies (a file that is not
    UTF-8, a coding alignment whose variable site falls on a zero site count: the rows before it are part of the transcript).
    A directory the reference dies on is run again without the file it died on, until it gets through: every file is either in
    a transcript that ends well or named as a victim."""
    cases = []
    for gen, seed in G14_SEEDS:
        files, flags = getattr(synth, gen + "_directory")(seed)
        if gen == "wild" and seed == 2181:
            continue
        tmp = tempfile.mkdtemp()
        d = os.path.join(tmp, "d")
        os.makedirs(d)
        for name, blob in files.items():
            with open(os.path.join(d, name), "wb") as fh:
                fh.write(blob)
        runs, removed = [], []
        while True:
            p = ref_cli(["-d", d, "-s"] + flags)
            out = p.stdout.decode()
            run = {"removed": list(removed), "returncode": p.returncode, "stdout": out.replace(tmp + "/", "{TMP}/")}
            runs.append(run)
            if p.returncode == 0:
                break
            run["stderr_tail"] = p.stderr.decode().strip().splitlines()[-1]
            # the file it died on: the one after the last file named in stdout
            names = sorted(os.listdir(d))
            seen = [ln.split(",")[0] if not ln.startswith("#") else ln.rsplit(": ", 1)[-1].rsplit(" ", 1)[-1] for ln in out.splitlines() if ln]
            seen = [x for x in seen if x in names]
            # the last one named in stdout (a comment line may precede the death) or the one after it: whichever kills the reference alone
            k = names.index(seen[-1]) if seen else -1
            cands = [names[j] for j in (k, k + 1) if 0 <= j < len(names)]
            victim = next((c for c in cands if ref_cli(["-f", os.path.join(d, c), "-s"] + flags).returncode != 0), cands[-1])
            run["victim"] = victim
            removed.append(victim)
            os.unlink(os.path.join(d, victim))
            assert len(removed) < len(files)
        shutil.rmtree(tmp)
        cases.append({"generator": gen, "seed": seed, "flags": flags, "n_files": len(files),
                      "files_sha1": hashlib.sha1(b"".join(files[k] for k in sorted(files))).hexdigest(), "runs": runs})
        print(f"g14 {gen} seed {seed}: {len(files)} files, flags {flags}, {len(runs)} run(s), died on {removed}", flush=True)
    dump("g14_random_directories.json", {"cases": cases})


# ------------------------------------------------------- G15 two-decimal ties on which the reference disagrees with itself
def g15():
    """Coding alignments on a two-decimal tie of sites_S whose printed digit the reference itself does not fix: a codon column of
    three or more codons is summed in the iteration order of a Python set of strings (PolyFastA.py:292, :307), which the
    interpreter's randomised string hashing draws anew in every run.  Frozen: the alignment (literal text) and what the reference's
    command line printed for sites_S / sites_N under PYTHONHASHSEED = 0 .. 59, as counts per printed pair.  Kept are cases with
    a clear majority (at least 60 %): this implementation prints the majority's digits (stats.syn_sites_most_likely).
    The first cases are the ones random directories ran into (synth.mixed_directory(590), f002.fa: one codon column, eight codons,
    the sum 3/8; two more); the others are searched: 30 sequences around a random consensus, one column planted with eight random codons."""
    from collections import Counter
    # (seed, file) of synth.mixed_directory: the rows on which tools/g14_explore.py (3 600 directories against the reference under
    # PYTHONHASHSEED=0) differed — the first before this policy existed, the other two with it: there the reference's run under
    # seed 0 is one of its rarer ones
    texts = [(f"mixed_directory_{sd}_{fn[:-3]}", synth.mixed_directory(sd)[0][fn].decode()) for sd, fn in ((590, "f002.fa"), (3090, "f008.fa"), (3517, "f000.fa"))]
    rng = np.random.default_rng(15)
    acgt = "ACGT"
    tried = 0
    while len(texts) < 11 and tried < 4000:
        tried += 1
        n, L = 30, int(rng.choice([3, 6, 30, 60, 300]))
        base = "".join(acgt[i] for i in rng.integers(0, 4, size=L))
        rows = []
        for r in range(n):
            x = list(base)
            for j in rng.integers(0, L, size=int(rng.integers(0, L // 20 + 2))).tolist():
                x[j] = acgt[int(rng.integers(0, 4))]
            rows.append(x)
        col = 3 * int(rng.integers(0, L // 3))
        cods = ["".join(acgt[i] for i in c) for c in rng.choice(4, size=(8, 3)).tolist()]
        if len(set(cods)) != 8:
            continue
        for r in range(n):
            rows[r][col:col + 3] = list(cods[r % 8])
        d = {f"s{r}": "".join(x) for r, x in enumerate(rows)}
        cs = ref.getvarCDSsites(d, L)[0]
        lo, hi = count_syn_range(d, L)
        if not is_tie(cs) or round(lo, 2) == round(hi, 2):
            continue
        texts.append((f"searched_{tried}_L{L}", "".join(f">{h}\n{x}\n" for h, x in d.items())))
    cases = []
    tmp = tempfile.mkdtemp()
    for label, text in texts:
        path = os.path.join(tmp, "x.fa")
        with open(path, "w", newline="") as fh:
            fh.write(text)
        seen = Counter()
        for hs in range(60):
            p = subprocess.run([sys.executable, os.path.join(REF_DIR, "PolyFastA.py"), "-f", path, "-s", "--cds"], capture_output=True,
                               env={**os.environ, "PYTHONHASHSEED": str(hs), "PYTHONUTF8": "1"})
            assert p.returncode == 0, p.stderr.decode()
            row = [ln for ln in p.stdout.decode().splitlines() if ln and not ln.startswith("#")][-1].split(",")
            seen[(row[1], row[2])] += 1
        top, cnt = seen.most_common(1)[0]
        print(f"  G15 {label}: {dict(seen)}", flush=True)
        if len(seen) < 2 or cnt < 36:
            continue
        cases.append({"label": label, "text": text, "printed": [[a, b, c] for (a, b), c in sorted(seen.items())], "most_often": list(top)})
    # One more kind (tools/g14_explore.py, wild directory 2181): four sites, one clean codon column of three codons worth 1.0
    # synonymous site of 1 — the non-synonymous site count is zero in exact arithmetic, and the reference divides by it: by exactly
    # 0.0 in the orders that sum to 1.0 (ZeroDivisionError, exit code 1), by 1.1e-16 in the others (a row with pi_N = 1.5e16).
    text = ">s0,x=0\nCTG>\n\r   \n>s1,x=1\nAGGA\n>s2,x=2\nCCGT\r\n"
    path = os.path.join(tmp, "x.fa")
    with open(path, "w", newline="") as fh:
        fh.write(text)
    seen = Counter()
    for hs in range(60):
        p = subprocess.run([sys.executable, os.path.join(REF_DIR, "PolyFastA.py"), "-f", path, "-s", "--cds"], capture_output=True,
                           env={**os.environ, "PYTHONHASHSEED": str(hs), "PYTHONUTF8": "1"})
        seen[(p.returncode, p.stdout.decode(), p.stderr.decode().strip().splitlines()[-1] if p.returncode else "")] += 1
    print(f"  G15 zero non-synonymous sites: {[(k[0], v) for k, v in seen.items()]}", flush=True)
    zero = {"text": text, "outcomes": [{"returncode": k[0], "stdout": k[1], "stderr_tail": k[2], "runs": v} for k, v in sorted(seen.items())]}
    assert len(seen) == 2 and max(seen.values()) >= 36
    # And the one place where the reference's rounding noise reaches a printed number that is not a rounding tie: the
    # Jukes-Cantor correction -0.75 log(1 - 4/3 x) at x = 0.75, pi per site at saturation (tools/g14_explore.py, wild directories
    # 40522 and 40138).  In exact arithmetic the logarithm's argument is zero (ValueError, and the reference's bare `except` prints
    # the uncorrected 0.75, PolyFastA.py:512-516); the reference evaluates pi in floating point (:485-492), and when its noise leaves
    # 0.7499999999999999 it prints -0.75 log(2.2e-16) = 27.03 instead — for the first text under every hash seed, for the second
    # under some.  This implementation computes pi from exact integers and so always takes the reference's fallback branch (0.75):
    # a known deviation, frozen here with what the reference prints.
    jc_cases = []
    for label, text, extra in (("coding_three_sequences", " >s0,x=0\nC\nT\ra\r\n>s1,x=1\nC\nG\nA\n>s2,x=2\r\nG\rG\r\nA\n>s3,x=3\nC\nT\nA\n", ["--cds", "--jc"]),
                              ("five_sequences_two_sites", "> s0\nGC\n> s1\nG-\r> s2\nAN\r\n> s3\r\nCN\r\n> s4\nGC\r", ["--jc"])):
        with open(path, "w", newline="") as fh:
            fh.write(text)
        seen = Counter()
        for hs in range(60):
            p = subprocess.run([sys.executable, os.path.join(REF_DIR, "PolyFastA.py"), "-f", path, "-s"] + extra, capture_output=True,
                               env={**os.environ, "PYTHONHASHSEED": str(hs), "PYTHONUTF8": "1"})
            assert p.returncode == 0
            seen[p.stdout.decode().strip().split("\n")[-1]] += 1
        print(f"  G15 jc singularity {label}: {dict(seen)}", flush=True)
        jc_cases.append({"label": label, "text": text, "flags": extra, "rows": [[r, c] for r, c in sorted(seen.items())]})
    shutil.rmtree(tmp)
    assert len(cases) >= 5 and cases[0]["label"].startswith("mixed_directory")
    dump("g15_hash_seed_ties.json", {"hash_seeds": 60, "cases": cases, "zero_nonsynonymous_sites": zero, "jc_singularity": jc_cases})


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15"]
    for w in which:
        globals()[w]()
