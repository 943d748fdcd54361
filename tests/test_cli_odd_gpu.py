"""G13: what the reference's COMMAND LINE prints for awkward inputs — the twenty G3 texts as one directory (zero-length
sequences, control white space, `>` inside a sequence line, empty headers), population keys at their edges, the same path
twice, a file that is not valid UTF-8 — against `python -m polyfasta_amd`, through the streaming ingest and its direct mode."""
import io
import os
import shutil
import subprocess
import sys
from contextlib import redirect_stdout

import pytest

from polyfasta_amd import cli, synth
from tests import _cases as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g13():
    return C.load("g13_cli_odd_inputs.json")


def run_inproc(argv):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = cli.main(argv)
    return rc, buf.getvalue()


def run_proc(argv, env=None, stdin=None):
    e = dict(os.environ, PYTHONUTF8="1")
    e.update(env or {})
    return subprocess.run([sys.executable, "-m", "polyfasta_amd"] + argv, cwd=ROOT, capture_output=True, env=e, input=stdin, timeout=600)


def same_lines(got, want, cds, jc):
    gl, wl = got.split("\n"), want.split("\n")
    assert len(gl) == len(wl), (got, want)
    for a, b in zip(gl, wl):
        if a == b:
            continue
        assert not b.startswith("#") and "," in b and C.row_matches(a, b, cds, jc) is None, (a, b)


MODES = {"streaming": {"PFA_INGEST_NO_DIRECT": "1", "PFA_INGEST_SLOT_BYTES": "4096"}, "direct": {}}


def _g3_dir(tmp_path, skip=()):
    g3 = C.load("g3_fasta_syntax.json")
    d = tmp_path / "g3"
    d.mkdir()
    for name, rec in g3.items():
        if name + ".fa" not in skip:
            (d / f"{name}.fa").write_bytes(rec["text"].encode())
    return g3


@pytest.mark.parametrize("mode", list(MODES))
def test_g3_texts_as_a_directory(gpu, g13, tmp_path, monkeypatch, mode):
    """`-d DIR`: every line the reference prints for the twenty G3 texts — rows of zero-length alignments, of sequences with
    `>` or blanks inside, the comment lines of the files it turns away — and each row again against G3's function-level
    `row`.  Then `--cds --jc` on the files the reference survives."""
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    g3 = _g3_dir(tmp_path)
    want = g13["g3_dir"]["plain"]
    rc, out = run_inproc([a.replace("{TMP}", str(tmp_path)) for a in want["args"]])
    assert rc == 0 == want["returncode"]
    same_lines(out, want["stdout"].replace("{TMP}", str(tmp_path)), False, False)
    rows = {ln.split(",")[0]: ln for ln in out.split("\n") if ln and not ln.startswith(("#", "file,"))}
    for name, rec in g3.items():
        if rec.get("row"):
            assert C.row_matches(rows[name + ".fa"], rec["row"].replace(name, name + ".fa", 1), False, False) is None, name
    # --cds --jc without the file the reference dies on
    surv = g13["g3_dir"]["cds_jc_survivors"]
    for r in surv["removed"]:
        os.unlink(tmp_path / "g3" / r["file"])
    rc, out = run_inproc([a.replace("{TMP}", str(tmp_path)) for a in surv["args"]])
    assert rc == 0
    same_lines(out, surv["stdout"].replace("{TMP}", str(tmp_path)), True, True)


@pytest.mark.parametrize("mode", list(MODES))
def test_g3_directory_under_cds_dies_where_the_reference_dies(gpu, g13, tmp_path, mode):
    """`-d DIR --cds`: the reference prints the rows up to pre_header_text.fa, whose one variable site is called
    non-synonymous while its non-synonymous site count is 0.0, and dies in polymorphism() with ZeroDivisionError
    (PolyFastA.py:509-517) — same stdout, same exit code, same last line on stderr."""
    _g3_dir(tmp_path)
    for label in ("cds", "cds_jc"):
        want = g13["g3_dir"][label]
        p = run_proc([a.replace("{TMP}", str(tmp_path)) for a in want["args"]], env=MODES[mode])
        assert p.returncode == want["returncode"] == 1, p.stderr.decode()[-1500:]
        assert p.stderr.decode().strip().splitlines()[-1] == want["stderr_tail"]
        same_lines(p.stdout.decode(), want["stdout"].replace("{TMP}", str(tmp_path)), True, label == "cds_jc")


def test_same_path_twice(gpu, g13):
    want = g13["same_path_twice"]
    rc, out = run_inproc([a.replace("{GOLDEN}", C.GOLDEN) for a in want["args"]])
    assert rc == 0
    same_lines(out, want["stdout"], False, False)


@pytest.mark.parametrize("mode", list(MODES))
def test_population_keys_at_their_edges(gpu, g13, tmp_path, monkeypatch, mode):
    """`-p pop1,` and `-p ,` (the empty key is in every header), a repeated key, a key that is a whole header, an empty key
    between two others: one row per key as typed, in that order (PolyFastA.py:123-134), coding and not, with and without
    --jc; and an alignment of zero-length sequences with keys."""
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    pe = g13["pops_edges"]
    g = pe["gen"]
    f = tmp_path / "x.fa"
    f.write_text(synth.to_fasta_text(synth.headers_for(g["N"], g["pops"]), C.matrix_from_gen(g)))
    for case in pe["cases"]:
        for cds in (False, True):
            for jc in (False, True):
                argv = ["-f", str(f), "-s", "-p", case["typed"]] + (["--cds"] if cds else []) + (["--jc"] if jc else [])
                rc, out = run_inproc(argv)
                assert rc == 0
                got = out.rstrip("\n").split("\n")
                want = [r[f"cds={int(cds)}|jc={int(jc)}"] for r in case["rows"]]
                assert len(got) == len(want), (case["label"], got, want)
                for a, b in zip(got, want):
                    assert C.row_matches(a, b, cds, jc) is None, (case["label"], a, b)
    z = g13["pops_zero_length"]
    zf = tmp_path / "zero.fa"
    zf.write_text(z["text"])
    # the same through a directory (the ingest's name table) and through -f
    zd = tmp_path / "zdir"
    zd.mkdir()
    shutil.copyfile(zf, zd / "zero.fa")
    for cds in (False, True):
        want = [r[f"cds={int(cds)}|jc=0"] if r["N"] else f"# Pop {r['key']} string was not found in fasta headers." for r in z["rows"]]
        for src in (["-f", str(zf)], ["-d", str(zd)]):
            rc, out = run_inproc(src + ["-s", "-p", z["typed"]] + (["--cds"] if cds else []))
            assert rc == 0 and out.rstrip("\n").split("\n") == want, (src, cds, out, want)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("where", ["in_sequence", "in_header", "in_replaced_record"])
def test_undecodable_file_dies_where_the_reference_dies(gpu, g13, tmp_path, mode, where):
    """A file that is not valid UTF-8 between two good ones (a lone 0x80 in a sequence line; a 0xFF in a header line; a lead byte
    without continuation in the sequence of a record that a later record of the same name replaces): the
    reference prints the header and the rows of the files before it, then dies in its read loop with UnicodeDecodeError,
    exit code 1 (PolyFastA.py:239-240).  Same stdout, same --out file, same exit code, same last stderr line.  On standard
    input the reference's text layer takes the byte as a lone surrogate (UTF-8 mode: surrogateescape) and prints a row."""
    u = g13["undecodable"][where]
    d = tmp_path / ("und_" + where)
    d.mkdir()
    shutil.copyfile(f"{C.GOLDEN}/example_theta_0.01/file1.fa", d / "a_good.fa")
    (d / "b_bad.fa").write_bytes(bytes.fromhex(u["bad_file_hex"]))
    shutil.copyfile(f"{C.GOLDEN}/example_theta_0.01/file3.fa", d / "c_good.fa")
    for key in ("dir", "dir_cds_jc_out_silent"):
        want = u["runs"][key]
        p = run_proc([a.replace("{TMP}", str(tmp_path)) for a in want["args"]], env=MODES[mode])
        assert p.returncode == want["returncode"] == 1, (p.returncode, p.stderr.decode()[-1500:])
        assert p.stderr.decode().strip().splitlines()[-1] == want["stderr_tail"], p.stderr.decode()[-1500:]
        same_lines(p.stdout.decode(), want["stdout"].replace("{TMP}", str(tmp_path)), "--cds" in want["args"], "--jc" in want["args"])
        if "file" in want:
            csv = tmp_path / f"und_{where}.csv"
            same_lines(open(csv).read(), want["file"], True, True)
            os.unlink(csv)
    want = u["runs"]["pipe"]
    p = run_proc(want["args"], stdin=bytes.fromhex(u["bad_file_hex"]))
    assert p.returncode == want["returncode"] == 0, p.stderr.decode()[-1500:]
    same_lines(p.stdout.decode(), want["stdout"], False, False)


def test_two_ranks_die_at_an_undecodable_file_like_one_process(gpu, g13, tmp_path):
    """the same directory as two ranks (gloo between them, both on this GPU): rank 0 prints the rows below the file and
    dies with the reference's error, the job's exit code is 1"""
    u = g13["undecodable"]["in_sequence"]
    d = tmp_path / "und"
    d.mkdir()
    shutil.copyfile(f"{C.GOLDEN}/example_theta_0.01/file1.fa", d / "a_good.fa")
    (d / "b_bad.fa").write_bytes(bytes.fromhex(u["bad_file_hex"]))
    shutil.copyfile(f"{C.GOLDEN}/example_theta_0.01/file3.fa", d / "c_good.fa")
    from polyfasta_amd import dist
    out = tmp_path / "rank0.txt"
    port = dist._free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONUTF8="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_cli_rank_worker.py"), str(out), "-d", str(d)], cwd=ROOT,
                                      stderr=subprocess.PIPE, env=env))
    errs = [p.communicate(timeout=600)[1].decode() for p in procs]
    want = u["runs"]["dir"]
    assert [p.returncode for p in procs] == [1, 1], errs
    same_lines(open(out).read(), want["stdout"], False, False)
    assert errs[0].strip().splitlines()[-1].replace("[rank0]: ", "") == want["stderr_tail"], errs[0][-1500:]      # (torch prefixes a rank's traceback)
    assert "UnicodeDecodeError" not in errs[1]


def test_paths_as_the_reference_treats_them(gpu, g13, tmp_path):
    """G13 `cli_paths`: a trailing slash after -d (the doubled slash shows in "is not FASTA!"), a relative directory, names with
    blanks, non-ASCII letters and a leading dot (sorted as Python sorts the path strings), an empty directory, a path that does
    not exist and a directory after -f (the reference's OSError after the rows before it), empty standard input, -n without
    --pipe, -p without a value and with an empty one, --out with such names in its progress chatter: stdout, the --out file,
    exit code and last stderr line, through real processes."""
    cp = g13["cli_paths"]
    d1 = tmp_path / "misc" / "d1"
    d1.mkdir(parents=True)
    (tmp_path / "misc" / "empty").mkdir()
    for fn, text in cp["files"].items():
        with open(d1 / fn, "w", newline="", encoding="utf-8") as fh:
            fh.write(text)
    sub = lambda t: t.replace("{TMP}", str(tmp_path))      # noqa: E731
    for name, want in cp["runs"].items():
        stdin = None if want["stdin_hex"] is None else bytes.fromhex(want["stdin_hex"])
        e = dict(os.environ, PYTHONUTF8="1", PYTHONPATH=ROOT)
        p = subprocess.run([sys.executable, "-m", "polyfasta_amd"] + [sub(a) for a in want["args"]], cwd=sub(want["cwd"]) if want["cwd"] else ROOT,
                           capture_output=True, env=e, input=stdin if stdin is not None else None,
                           stdin=subprocess.DEVNULL if stdin is None else None, timeout=600)
        assert p.returncode == want["returncode"], (name, p.returncode, p.stderr.decode()[-1500:])
        same_lines(p.stdout.decode(), sub(want["stdout"]), "--cds" in want["args"], False)
        if want["returncode"]:
            assert p.stderr.decode().strip().splitlines()[-1] == sub(want["stderr_tail"]), (name, p.stderr.decode()[-1500:])
        if "file" in want:
            same_lines(open(sub(want["args"][want["args"].index("-o") + 1])).read(), want["file"], False, False)


@pytest.mark.parametrize("mode", list(MODES))
def test_more_awkward_file_contents(gpu, g13, tmp_path, monkeypatch, mode):
    """G13 `more_texts`: a byte-order mark in front of the first header (that record is lost: the reference reads utf-8, not
    utf-8-sig), a NUL inside a sequence, a header of blanks only in the middle and at the end, a lone '>', a header without a line
    end, tabs behind a header and inside a sequence, a form feed between records (file iteration does not split there), 5 000
    sites on one line next to the same wrapped at 7 — `-d` and `-d --cds --jc`, every line."""
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    mt = g13["more_texts"]
    d = tmp_path / "more"
    d.mkdir()
    for fn, hx in mt["files_hex"].items():
        (d / fn).write_bytes(bytes.fromhex(hx))
    for label, want in mt["runs"].items():
        assert not want["removed"]
        rc, out = run_inproc([a.replace("{TMP}", str(tmp_path)) for a in want["args"]])
        assert rc == 0
        same_lines(out, want["stdout"].replace("{TMP}", str(tmp_path)), "--cds" in want["args"], "--jc" in want["args"])


def test_carriage_returns_on_standard_input_and_in_files(gpu, g13, tmp_path, monkeypatch):
    """A carriage return ends a line in a FILE (open(file, 'r'): universal newlines) but not on standard input, which Python
    splits at \\n only: a piped file with bare-CR line ends is one header line (one sequence, no sites), \\r\\r\\n ends a line like
    \\r\\n, a bare CR inside a sequence line is one of its characters — the reference's `--pipe` and `-f` on the same five texts."""
    import io
    cr = g13["carriage_returns"]
    for run in cr["runs"]:
        blob = bytes.fromhex(cr["texts_hex"][run["label"]])
        cds, jc = "--cds" in run["args"], "--jc" in run["args"]
        if "--pipe" in run["args"]:
            monkeypatch.setattr(sys, "stdin", io.TextIOWrapper(io.BytesIO(blob), encoding="utf-8", errors="surrogateescape", newline="\n"))
            rc, out = run_inproc(run["args"])
        else:
            (tmp_path / f"{run['label']}.fa").write_bytes(blob)
            rc, out = run_inproc([a.replace("{TMP}", str(tmp_path)) for a in run["args"]])
        assert rc == run["returncode"] == 0, run
        same_lines(out, run["stdout"].replace("{TMP}", str(tmp_path)), cds, jc)
    # and through a real pipe
    for run in cr["runs"]:
        if run["label"] in ("bare_cr", "cr_inside_sequence") and run["args"] == ["--pipe", "-s"]:
            p = run_proc(run["args"], stdin=bytes.fromhex(cr["texts_hex"][run["label"]]))
            assert p.returncode == 0 and p.stdout.decode() == run["stdout"], (run["label"], p.stdout, p.stderr[-500:])


@pytest.mark.parametrize("mode", list(MODES))
def test_files_in_which_no_population_key_is_found(gpu, g13, tmp_path, monkeypatch, mode):
    """None of such a file's rows goes to the card, and it is the card's repack that tells non-ASCII characters: a length of 3
    characters that is 4 bytes (no "not a multiple of 3" line), sequences equally long in bytes but not in characters (unequal
    lengths) — the host looks at the bytes of such a file itself; blanks and control white space at line ends that are no sites
    (the loader's own count); and a file in which one key of two is found: the comment lines of the reference's main()."""
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    nk = g13["no_key_found"]
    d = tmp_path / "nokey"
    d.mkdir()
    for fn, hx in nk["files_hex"].items():
        (d / fn).write_bytes(bytes.fromhex(hx))
    for run in nk["runs"]:
        rc, out = run_inproc([a.replace("{TMP}", str(tmp_path)) for a in run["args"]])
        assert rc == 0
        same_lines(out, run["stdout"], "--cds" in run["args"], "--jc" in run["args"])
    # the same files one by one after -f
    for run in nk["runs"]:
        rc, out = run_inproc(["-f"] + [str(d / fn) for fn in sorted(nk["files_hex"], reverse=True)] + run["args"][2:])
        assert rc == 0
        same_lines(out, run["stdout"], "--cds" in run["args"], "--jc" in run["args"])
