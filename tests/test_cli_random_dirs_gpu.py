"""G14: the reference's command-line transcripts of 83 random directories — 49 of small awkward files (1 to 30 sequences of 1 to
1 001 sites; lower case, other symbols, unequal lengths, non-ASCII characters, no header, blank lines, repeated headers, a byte
that is not UTF-8; LF / CRLF / CR; wrapped lines), 28 wilder ones (odd palettes, blanks and control white space in and after lines, mixed line ends, stray lines, empty
files, names with blanks and commas) and 6 of 65 to 300 sequences (the other kernels) — under no flag, --jc, --cds, --cds --jc — against `python -m polyfasta_amd`
through the streaming ingest and its direct mode: every line, the exit code, and where the reference dies its last word."""
import os

import pytest

from polyfasta_amd import synth
from tests import _cases as C
from tests.test_cli_odd_gpu import MODES, run_inproc, run_proc, same_lines

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g14():
    return C.load("g14_random_directories.json")


@pytest.mark.parametrize("mode", list(MODES))
def test_random_directories_line_by_line(gpu, g14, tmp_path, monkeypatch, mode):
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    rows = dead = 0
    for case in g14["cases"]:
        files, flags = getattr(synth, case["generator"] + "_directory")(case["seed"])
        cds, jc = "--cds" in flags, "--jc" in flags
        top = tmp_path / f"{case['generator']}{case['seed']}"
        d = top / "d"
        d.mkdir(parents=True)
        for name, blob in files.items():
            (d / name).write_bytes(blob)
        for run in case["runs"]:
            for name in run["removed"]:
                if (d / name).exists():
                    os.unlink(d / name)
            want = run["stdout"].replace("{TMP}", str(top))
            argv = ["-d", str(d), "-s"] + flags
            if run["returncode"] == 0:
                rc, out = run_inproc(argv)
                assert rc == 0, (case["seed"], run["removed"])
                same_lines(out, want, cds, jc)
                rows += sum(1 for ln in out.split("\n") if ln and not ln.startswith("#"))
            else:
                # the reference dies on a file: the rows before it, the same exit code, the same last line on stderr
                p = run_proc(argv, env=MODES[mode])
                assert p.returncode == run["returncode"] == 1, (case["seed"], p.stderr.decode()[-1500:])
                assert p.stderr.decode().strip().splitlines()[-1] == run["stderr_tail"], case["seed"]
                same_lines(p.stdout.decode(), want, cds, jc)
                dead += 1
    assert rows > 750 and dead >= 6
