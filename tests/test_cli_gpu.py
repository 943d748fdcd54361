"""The command line against transcripts of the reference's own CLI (G7) and
the per-population rows (G6), on the GPU."""
import io
import json
import os
import shutil
import sys
from contextlib import redirect_stdout

import pytest

from polyfasta_amd import cli, synth
from tests import _cases as C

pytestmark = pytest.mark.gpu


def run_cli(argv, stdin_bytes=None, monkeypatch=None):
    buf = io.StringIO()
    if stdin_bytes is not None:
        class _In:
            buffer = io.BytesIO(stdin_bytes)
        monkeypatch.setattr(sys, "stdin", _In())
    with redirect_stdout(buf):
        rc = cli.main(argv)
    return rc, buf.getvalue()


def same_transcript(got, want):
    gl, wl = got.split("\n"), want.split("\n")
    assert len(gl) == len(wl), (got, want)
    for a, b in zip(gl, wl):
        if a == b:
            continue
        assert not a.startswith("#") and "," in a, (a, b)
        C.rows_close(a, b, rel=1e-9, rel_jc=1e-6, jc=True)


@pytest.fixture()
def layout(tmp_path):
    g7 = C.load("g7_cli.json")
    odd = tmp_path / "odd"
    odd.mkdir()
    for name, text in g7["_odd_texts"].items():
        (odd / f"{name}.fa").write_bytes(text.encode())
    (tmp_path / "novar.fa").write_text(g7["_novar_text"])
    ex = tmp_path / "examples"
    ex.mkdir()
    for fn in os.listdir(f"{C.GOLDEN}/example_theta_0.01"):
        if fn.endswith(".fa"):
            shutil.copyfile(f"{C.GOLDEN}/example_theta_0.01/{fn}", ex / fn)

    def sub(a):
        return a.replace("{GOLDEN}", C.GOLDEN).replace("{TMP}", str(tmp_path)).replace("{EXAMPLES}", str(ex))
    return g7, tmp_path, sub


def test_stdout_transcripts(gpu, layout, monkeypatch):
    g7, tmp, sub = layout
    for name in ("file1", "two_files", "file1_jc", "file1_cds", "file1_cds_jc", "file1_silent", "odd_dir",
                 "odd_dir_cds", "novar", "novar_cds", "dir", "dir_jc", "dir_cds", "dir_cds_jc"):
        t = g7[name]
        rc, out = run_cli([sub(a) for a in t["args"]])
        assert rc == 0
        same_transcript(out, sub(t["stdout"]))


def test_pipe_transcripts(gpu, layout, monkeypatch):
    g7, tmp, sub = layout
    data = open(f"{C.GOLDEN}/example_theta_0.01/file3.fa", "rb").read()
    for name in ("pipe", "pipe_named", "pipe_cds"):
        rc, out = run_cli(g7[name]["args"], stdin_bytes=data, monkeypatch=monkeypatch)
        same_transcript(out, g7[name]["stdout"])


def test_out_file_transcripts(gpu, layout):
    g7, tmp, sub = layout
    for name in ("out_file", "out_file_silent_append", "out_file_odd_cds"):
        t = g7[name]
        rc, out = run_cli([sub(a) for a in t["args"]])
        assert out == sub(t["stdout"])                      # progress chatter, byte for byte
        csv = "res2.csv" if "odd" in name else "res.csv"
        same_transcript(open(tmp / csv).read(), t["file"])


def test_argument_errors(gpu, layout):
    g7, tmp, sub = layout
    with pytest.raises(SystemExit) as e:
        cli.main(["-f", "x.fa", "-d", str(tmp)])
    assert e.value.code == 2
    with pytest.raises(SystemExit):
        cli.main(["-f", "x.fa", "--pipe"])


def test_pops_rows(gpu, tmp_path):
    """--pops: the reference crashes at PolyFastA.py:126; rows are pinned by calling
    its print_result per population (G6)."""
    for case in C.load("g6_pops.json"):
        d = C.dict_from_gen(case["gen"])
        if case["extra_gen"]:
            for i, (_, s) in enumerate(C.dict_from_gen(case["extra_gen"]).items()):
                d[f"pop10_x{i}"] = s
        path = tmp_path / "x.fa"
        path.write_text("".join(f">{h}\n{s}\n" for h, s in d.items()))
        keys = []
        for k in case["rows"]:
            pop = k.split("|")[0]
            if pop not in keys:
                keys.append(pop)
        for cds in (False, True):
            for jc in (False, True):
                argv = ["-f", str(path), "-p", ",".join(keys), "-s"] + (["--cds"] if cds else []) + (["--jc"] if jc else [])
                rc, out = run_cli(argv)
                lines = out.rstrip("\n").split("\n")
                want = []
                for pop in keys:
                    if case["rows"].get(pop, 0) is None:
                        want.append(f"# Pop {pop} string was not found in fasta headers.")
                    else:
                        want.append(case["rows"][f"{pop}|cds={int(cds)}|jc={int(jc)}"])
                assert len(lines) == len(want)
                for a, b in zip(lines, want):
                    if a != b:
                        C.rows_close(a, b, jc=jc)


def test_many_files_small_buffers(gpu, tmp_path, monkeypatch):
    """Force several ingest batches, deferred files and one file larger than the
    buffer: rows must still come out once each, in sorted-path order."""
    import numpy as np
    from oracle import c_oracle
    monkeypatch.setattr(cli, "BATCH_BYTES", 1 << 20)
    monkeypatch.setattr(cli, "_pinned_cache", [None, None])
    rows = {}
    p = synth.make_params(12, 0.02)
    for i in range(40):
        L = 30000 if i != 17 else 200000          # file 17 alone exceeds the 1 MiB buffer
        m = synth.synth_alignment(900, i, L, p)
        (tmp_path / f"f{i:02d}.fa").write_text(synth.to_fasta_text(synth.headers_for(12), m))
        o = c_oracle.scan(m)
        rows[f"f{i:02d}.fa"] = (L, o["S"])
    rc, out = run_cli(["-d", str(tmp_path), "-s"])
    lines = out.rstrip("\n").split("\n")
    assert [ln.split(",")[0] for ln in lines] == sorted(rows)
    for ln in lines:
        f = ln.split(",")
        assert (int(f[1]), int(f[4])) == rows[f[0]]
    monkeypatch.setattr(cli, "_pinned_cache", [None, None])
