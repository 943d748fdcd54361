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
    """Force many small pinned slots (every slot = one H2D copy, several batches) and one file
    larger than a slot, which travels in buffers of its own: rows must still come out once each,
    in sorted-path order, for one loader thread and for many."""
    from oracle import c_oracle
    monkeypatch.setenv("PFA_INGEST_SLOT_BYTES", str(1 << 20))
    rows = {}
    p = synth.make_params(12, 0.02)
    for i in range(40):
        L = 30000 if i != 17 else 200000          # file 17 alone exceeds the 1 MiB slot
        m = synth.synth_alignment(900, i, L, p)
        (tmp_path / f"f{i:02d}.fa").write_text(synth.to_fasta_text(synth.headers_for(12), m))
        o = c_oracle.scan(m)
        rows[f"f{i:02d}.fa"] = (L, o["S"])
    # the copy pipeline (PFA_INGEST_DIRECT_MAX=0: pinned ring, uploader, device text ring) and the direct mode of small jobs
    # (the card reads the pinned buffers itself; one buffer per use, nothing recycled)
    for direct_max in ("0", None):
        if direct_max is None:
            monkeypatch.delenv("PFA_INGEST_DIRECT_MAX", raising=False)
        else:
            monkeypatch.setenv("PFA_INGEST_DIRECT_MAX", direct_max)
        for threads in ("1", "3", "16"):
            monkeypatch.setenv("PFA_THREADS", threads)
            rc, out = run_cli(["-d", str(tmp_path), "-s"])
            lines = out.rstrip("\n").split("\n")
            assert [ln.split(",")[0] for ln in lines] == sorted(rows)
            for ln in lines:
                f = ln.split(",")
                assert (int(f[1]), int(f[4])) == rows[f[0]]
            assert (cli.last_stats["bytes_copied"] == 0) == (direct_max is None), cli.last_stats     # direct: no H2D copy at all


def test_ingest_repacks_any_text_like_the_reference_reader(gpu, tmp_path):
    """The streaming ingest end to end against the oracle's reader on awkward files: wrapped records,
    CRLF and lone-CR line ends, lower case, blank lines, trailing blanks, duplicate headers, text before
    the first header, no final newline, rows of 1..40 sites at every byte offset (the repack kernel reads
    unaligned runs), IUPAC / gap symbols, and a file with a byte >= 0x80 in a sequence line."""
    import io
    import random
    from oracle import c_oracle, polyfasta_oracle as orc
    rng = random.Random(23)
    want = {}
    for i in range(60):
        eol = rng.choice(["\n", "\r\n", "\r"])
        n, L = rng.randint(2, 9), rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 33, 40, 257, 1000])
        seqs = ["".join(rng.choice("ACGTacgtNn-RY") for _ in range(L)) for _ in range(n)]
        seqs[1] = seqs[0]
        text = rng.choice(["", "junk before\n"])
        for r, sq in enumerate(seqs):
            w = rng.choice([1000, 60, 7])
            lines = [sq[k:k + w] + rng.choice(["", " ", "\t"]) for k in range(0, L, w)]
            if rng.random() < 0.3:
                lines.insert(rng.randint(0, len(lines)), "")
            text += f">{'x' * rng.randint(0, 5)}s{r}" + eol + eol.join(lines) + (eol if r + 1 < n or rng.random() < 0.7 else "")
        if i % 10 == 3:
            text = f">s0{eol}TTTT{eol}" + text                   # the first s0 is overwritten, its position kept
        (tmp_path / f"t{i:02d}.fa").write_bytes(text.encode())
        d = orc.parse_fasta_lines(io.StringIO(text, newline=None))
        if len({len(v) for v in d.values()}) != 1:
            want[f"t{i:02d}.fa"] = None                          # PolyFastA.py:135-140
            continue
        m = __import__("numpy").array([list(v.encode()) for v in d.values()], dtype="uint8")
        o = c_oracle.scan(m)
        want[f"t{i:02d}.fa"] = (L, len(d), o["S"], o["H"])
    (tmp_path / "t99_bad.fa").write_bytes(b">a\nAC\xc3\xb1T\n>b\nACGTT\n")
    rc, out = run_cli(["-d", str(tmp_path), "-s"])
    lines = out.rstrip("\n").split("\n")
    # a non-ASCII character in a sequence line: read as text like the reference (4 characters against 5)
    assert lines[-1] == "# Sequences do not have the same length: t99_bad.fa"
    assert len(lines) - 1 == len(want)
    for ln, name in zip(lines[:-1], sorted(want)):
        if want[name] is None:
            assert ln == f"# Sequences do not have the same length: {name}"
            continue
        f = ln.split(",")
        assert f[0] == name
        L, n, S, H = want[f[0]]
        assert (int(f[1]), int(f[3]), int(f[4])) == (L, n, S), (ln, want[f[0]])
        if S:
            assert abs(float(f[5]) - H / (n * (n - 1)) / L) <= 1e-12 * float(f[5])


def test_ingest_soak_mixed_directory(gpu, tmp_path, monkeypatch):
    """The same directory through the CLI thirty times in one process — files of very different sizes, one larger
    than a pinned slot (it travels in buffers of its own), sixteen loaders placing files in whatever order they
    finish — every row against the C oracle every time.  Buffers are recycled between the runs: a pinned buffer
    handed back twice showed up here as garbage rows once in ten runs."""
    from oracle import c_oracle
    import numpy as np
    monkeypatch.setenv("PFA_INGEST_SLOT_BYTES", str(3 << 20))
    monkeypatch.setenv("PFA_INGEST_DIRECT_MAX", "0")      # the copy pipeline (pinned ring, uploader), not the direct mode of small jobs
    from oracle import polyfasta_oracle as orc
    rng = np.random.default_rng(5)
    want = []
    for i in range(30):
        n = int(rng.choice([12, 20, 50, 80]))
        L = int(rng.choice([900, 3000, 9000, 30000])) if i != 5 else 400000
        m = synth.synth_alignment(4200, i, L, synth.make_params(n, 0.02), coding=True)
        (tmp_path / f"g{i:02d}.fa").write_text(synth.to_fasta_text(synth.headers_for(n, 2), m))
        want.append(orc.row_from_counts(f"g{i:02d}.fa", "NA", n, L, c_oracle.scan(m), False, False))
    for rep in range(30):
        rc, out = run_cli(["-d", str(tmp_path), "-s"])
        got = out.splitlines()
        assert len(got) == len(want)
        bad = [(rep, a, b) for a, b in zip(got, want) if a != b and C.row_matches(a, b, False, False)]      # every column of every row
        assert not bad, bad[:3]


def test_directory_listed_by_the_library_in_the_reference_order(gpu, tmp_path):
    """`--dir`: the library lists and sorts the directory itself; the rows come out in the order of
    `sorted(dir + "/" + name for name in os.listdir(dir))` (PolyFastA.py:93-94, :104) — digits, case, a dot file,
    multi-byte UTF-8 names — and a name that is not UTF-8 sends the listing back to Python (same rows)."""
    names = ["file10.fa", "file2.fa", "File3.fa", "a b.fa", ".hidden.fa", "zeta.fa", "\u00e9clair.fa", "\u4e2d\u6587.fa", "~tilde.fa", "Z.fa", "_x.fa"]
    p = synth.make_params(8, 0.05)
    for k, nm in enumerate(names):
        (tmp_path / nm).write_text(synth.to_fasta_text(synth.headers_for(8), synth.synth_alignment(31, k, 300 + k, p)))
    want = [x.split("/")[-1] for x in sorted(str(tmp_path) + "/" + i for i in os.listdir(tmp_path))]
    rc, out = run_cli(["-d", str(tmp_path), "-s"])
    got = [ln.split(",")[0] for ln in out.splitlines()]
    assert rc == 0 and got == want and len(got) == len(names)
    lens = {ln.split(",")[0]: int(ln.split(",")[1]) for ln in out.splitlines()}
    assert all(lens[nm] == 300 + k for k, nm in enumerate(names))
    # a name that is not UTF-8: the library declines (Python would sort the escaped bytes by code point, and the
    # reference itself cannot print such a name): the caller lists the directory the reference's way
    from polyfasta_amd import ingest
    with open(os.fsencode(str(tmp_path)) + b"/caf\xe9.fa", "wb") as fh:
        fh.write(b">a\nACGT\n>b\nACGA\n")
    assert ingest.Ingest.from_dir(str(tmp_path)) is None
    ing = ingest.Ingest.from_dir(str(tmp_path.parent / "no_such_directory"))
    assert ing is None


def test_sequences_with_non_ascii_characters(gpu, tmp_path, monkeypatch):
    """G9: the reference reads text, so a character is a site whatever its UTF-8 length, upper() is Unicode's (a sharp s
    becomes two letters: lengths change) and rstrip() strips Unicode white space.  The native loader hands such files back
    and the command line reads them as the reference does; the scan still runs on the card."""
    g9 = C.load("g9_unicode.json")
    d = tmp_path / "u"
    d.mkdir()
    for name, text in g9["_texts"].items():
        (d / f"{name}.fa").write_bytes(text.encode("utf-8"))
    # two plain files around them, so that the batch mixes both kinds
    p = synth.make_params(8, 0.02)
    for k, name in enumerate(("aaa_first", "zzz_last")):
        (d / f"{name}.fa").write_text(synth.to_fasta_text(synth.headers_for(8), synth.synth_alignment(3, k, 60, p)))
    for run, extra in (("dir", []), ("dir_cds_jc", ["--cds", "--jc"])):
        rc, out = run_cli(["-d", str(d)] + extra)
        assert rc == 0
        lines = [ln for ln in out.split("\n") if not ln.startswith(("aaa_first.fa", "zzz_last.fa"))]
        assert len(lines) + 2 == len(out.split("\n"))
        same_transcript("\n".join(lines), g9["runs"][run]["stdout"].replace("{TMP}", str(d)))
    # populations: rows of the reference's print_result on the split it intends; a key that is itself non-ASCII; a missing key
    for cds in (0, 1):
        rc, out = run_cli(["-f", str(d / "accents.fa"), "-p", "pop1,pop2,\u00fc,absent", "-s"] + (["--cds"] if cds else []))
        want = [g9["pops"][f"{k}|cds={cds}"] for k in ("pop1", "pop2", "\u00fc")] + ["# Pop absent string was not found in fasta headers."]
        got = [ln for ln in out.split("\n") if ln]
        assert len(got) == len(want), out
        for a, b in zip(got, want):
            if a != b:
                C.rows_close(a, b, rel=1e-9, rel_jc=1e-6, jc=False)
    # --pipe: the same statistics as the file's row
    rc, out = run_cli(["--pipe", "-s"], stdin_bytes=g9["_texts"]["accents"].encode("utf-8"), monkeypatch=monkeypatch)
    want = [ln for ln in g9["runs"]["dir"]["stdout"].split("\n") if ln.startswith("accents.fa")][0]
    got = [ln for ln in out.split("\n") if ln][0]
    assert got.split(",")[1:5] == want.split(",")[1:5]
    C.rows_close("x," + got.split(",", 1)[1], "x," + want.split(",", 1)[1], rel=1e-9, rel_jc=1e-6, jc=False)
    # not text at all (a lone continuation byte): the reference dies in its read loop with a UnicodeDecodeError (PolyFastA.py:239-240),
    # and so does this command line (G13, tests/test_cli_odd_gpu.py, pins stdout, exit code and message through a real process)
    (d / "broken.fa").write_bytes(b">a\nAC\x80T\n>b\nACGT\n")
    with pytest.raises(UnicodeDecodeError):
        run_cli(["-f", str(d / "broken.fa"), "-s"])
    # the API mirror on the same file: the reference's function names on a dict of text (PolyFastA.py:227-261, :502-520)
    from oracle import polyfasta_oracle as orc
    from polyfasta_amd import api
    da, do = api.readfasta(str(d / "accents.fa"), False), orc.readfasta(str(d / "accents.fa"))
    assert da == do and list(da) == list(do)
    (pa, va), (po, vo) = api.getvarsites(da, 60), orc.getvarsites(do, 60)
    assert pa == po and [list(c) for c in va] == [list(c) for c in vo]
    for x, y in zip(api.polymorphism(va, 60, False), orc.polymorphism(vo, 60, False)):
        assert x == y or abs(x - y) <= 1e-9 * abs(y)


def test_directory_scale_every_row_every_column(gpu, tmp_path, monkeypatch):
    """The streaming ingest at directory scale in the coding and population modes (PolyFastA.py:122-134, :165-180): 2 048
    files of 50 x 9 000 in-frame CDS with pop1_* / pop2_* headers — among them one with N's, one in lower case, one wrapped
    at 60 columns, one with unequal lengths, one with a premature stop codon — through 1 MiB slots and 16 loader threads
    (many batches, files placed in whatever order the loaders finish).  `-p pop1,pop2 --cds --jc`, `--cds` and
    `-p pop2,pop1`: EVERY row, ALL 14 / 8 columns against the checker (C oracle integers + its own finishing math):
    text and integers equal — sites_S / sites_N as printed text: files 569 and 1894 of this directory land on two-decimal
    rounding ties and are printed from the reference's own accumulation (tests/golden G12) — floats 1e-9, Jukes-Cantor pi 1e-6."""
    import ctypes
    import numpy as np
    from oracle import c_oracle
    from oracle import polyfasta_oracle as orc
    from polyfasta_amd import _lib, scan
    monkeypatch.setenv("PFA_INGEST_SLOT_BYTES", str(1 << 20))
    monkeypatch.setenv("PFA_THREADS", "16")
    M, N, L = 2048, 50, 9000
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (3 << 30) else str(tmp_path)
    d = os.path.join(root, f"pfa_dirscale_{os.getpid()}")
    os.makedirs(d)
    try:
        p = synth.make_params(N, 0.01)
        stride = scan.row_stride_for(L, 128)
        sp = _lib.SynthParamsC(seed=404, first_aln=0, n_aln=M, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                               tri_thresh=p.tri_thresh, n_units=len(p.units), coding=1)
        bases = np.zeros(M * N * stride, dtype=np.uint8)
        _lib.check(_lib.load().pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                                   _lib.ptr(np.ascontiguousarray(p.units))))
        mats = bases.reshape(M, N, stride)[:, :, :L]
        hdr = synth.headers_for(N, 2)
        odd = {"with_N": 77, "lower": 300, "wrapped": 901, "unequal": 1500, "stop": 2047}
        m = mats[odd["with_N"]]
        m[3, 100:106] = ord("N")
        m[40, 4000] = ord("N")
        m = mats[odd["stop"]]
        m[::2, 300:303] = np.frombuffer(b"TAA", dtype=np.uint8)        # a premature stop in every pop1 row
        for i in range(M):
            rows = [mats[i][r].tobytes() for r in range(N)]
            if i == odd["lower"]:
                rows = [x.lower() for x in rows]
            if i == odd["wrapped"]:
                rows = [b"\n".join(x[k:k + 60] for k in range(0, L, 60)) for x in rows]
            if i == odd["unequal"]:
                rows[7] = rows[7][:-3]
            with open(os.path.join(d, f"cds{i:05d}.fa"), "wb") as fh:
                fh.write(b"".join(b">" + h.encode() + b"\n" + x + b"\n" for h, x in zip(hdr, rows)))
        pop_rows = {"pop1": np.arange(0, N, 2, dtype=np.uint32), "pop2": np.arange(1, N, 2, dtype=np.uint32)}
        cache = {}

        def want_rows(keys, cds, jc):
            out = []
            for i in range(M):
                name = f"cds{i:05d}.fa"
                if i == odd["unequal"]:
                    out.append(f"# Sequences do not have the same length: {name}")
                    continue
                for key in keys or [None]:
                    ck = (i, key, cds)
                    if ck not in cache:
                        cache[ck] = c_oracle.scan(mats[i], rows=None if key is None else pop_rows[key], cds=cds)
                    n = N if key is None else len(pop_rows[key])
                    out.append(orc.row_from_counts(name, key or "NA", n, L, cache[ck], cds, jc))
            return out

        for argv, keys, cds, jc in ((["-p", "pop1,pop2", "--cds", "--jc"], ["pop1", "pop2"], True, True), (["--cds"], None, True, False),
                                    (["-p", "pop2,pop1"], ["pop2", "pop1"], False, False)):
            rc, out = run_cli(["-d", d, "-s"] + argv)
            assert rc == 0
            got = out.rstrip("\n").split("\n")
            want = want_rows(keys, cds, jc)
            assert len(got) == len(want), (argv, len(got), len(want))
            assert cli.last_stats["batches"] >= 16 and cli.last_stats["loader_threads"] == 16, cli.last_stats
            bad = [(a, b, C.row_matches(a, b, cds, jc)) for a, b in zip(got, want) if a != b and (b.startswith("#") or C.row_matches(a, b, cds, jc))]
            assert not bad, (argv, len(bad), bad[:3])
        # the odd files did what they were meant to: the stop is counted, the N's make unclean codons, every file has variation
        stop_row = [ln for ln in out.split("\n") if ln.startswith(f"cds{odd['stop']:05d}.fa")]
        assert len(stop_row) == 2
        rc, out = run_cli(["-d", d, "-s", "--cds", "-p", "pop1"])
        f = [ln for ln in out.split("\n") if ln.startswith(f"cds{odd['stop']:05d}.fa")][0].split(",")
        assert int(f[13]) >= 1
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_unreadable_directory_entry_dies_where_the_reference_dies(gpu, tmp_path):
    """G11: a subdirectory between the alignments of `--dir`.  The reference prints the header and every row of the files
    before it, then dies at its open() with IsADirectoryError, exit code 1 (PolyFastA.py:104-108, :239); rows of later files
    never appear.  Same stdout, same --out file, same exit code, same exception — through the real command line."""
    import subprocess
    g = C.load("g11_unreadable_entry.json")
    d = tmp_path / "mixed"
    (d / g["layout"]["subdirectory"]).mkdir(parents=True)
    for fn in g["layout"]["files"]:
        shutil.copyfile(f"{C.GOLDEN}/example_theta_0.01/{fn}", d / fn)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for key in ("dir", "dir_cds_jc_out_silent"):
        want = g[key]
        argv = [a.replace("{TMP}", str(tmp_path)) for a in want["args"]]
        p = subprocess.run([sys.executable, "-m", "polyfasta_amd"] + argv, cwd=root, capture_output=True, text=True, timeout=600)
        assert p.returncode == want["returncode"] == 1, (p.returncode, p.stderr[-2000:])
        assert p.stderr.strip().splitlines()[-1] == want["stderr_tail"].replace("{TMP}", str(tmp_path)), p.stderr[-2000:]
        same_transcript(p.stdout, want["stdout"])
        if "file" in want:
            same_transcript(open(tmp_path / "res.csv").read(), want["file"])
            os.unlink(tmp_path / "res.csv")


def test_directory_of_large_alignments_every_row_every_column(gpu, tmp_path, monkeypatch):
    """Files of hundreds to thousands of sequences through the command line (the reference has no row limit,
    PolyFastA.py:252-261, :284-315): the ingest hands them to the row-chunked kernel — its 56-bit coding form above 2 048
    rows — next to small files on the classic kernels in the same batches; per-population matrices of 300 and 1 500 rows.
    `--cds --jc`, plain and `-p pop1,pop2 --cds`: every row, every column against the checker."""
    import ctypes
    import numpy as np
    from oracle import c_oracle
    from oracle import polyfasta_oracle as orc
    from polyfasta_amd import _lib, scan
    from polyfasta_amd import ingest
    shapes = [(3000, 9000), (50, 9000), (700, 30000), (600, 20001), (129, 3000), (2049, 999), (64, 9000), (1000, 9000)]
    launched = []          # the kernels of every batch's plan
    fetch = ingest.IngestBatch.fetch

    def spy(self, coding, want_sfs=False):
        if self._plan:
            launched.append(_lib.load().pfa_plan_describe(self._plan).decode())
        return fetch(self, coding, want_sfs)
    monkeypatch.setattr(ingest.IngestBatch, "fetch", spy)
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (1 << 30) else str(tmp_path)
    d = os.path.join(root, f"pfa_large_{os.getpid()}")
    os.makedirs(d)
    try:
        mats, names = [], []
        for i, (N, L) in enumerate(shapes):
            p = synth.make_params(N, 0.01)
            stride = scan.row_stride_for(L, 128)
            sp = _lib.SynthParamsC(seed=900 + i, first_aln=0, n_aln=1, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                                   tri_thresh=p.tri_thresh, n_units=len(p.units), coding=1)
            bases = np.zeros(N * stride, dtype=np.uint8)
            _lib.check(_lib.load().pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                                       _lib.ptr(np.ascontiguousarray(p.units))))
            m = np.ascontiguousarray(bases.reshape(N, stride)[:, :L])
            if i == 0:
                m[5::7, 600:603] = np.frombuffer(b"TGA", dtype=np.uint8)      # a stop codon in every seventh row
                m[11, 3000:3009] = ord("N")
            if i == 2:
                m[::3, 12345] = ord("-")
            mats.append(m)
            names.append(f"aln{i:02d}_{N}x{L}.fa")
            hdr = synth.headers_for(N, 2)
            with open(os.path.join(d, names[-1]), "wb") as fh:
                fh.write(b"".join(b">" + h.encode() + b"\n" + m[r].tobytes() + b"\n" for r, h in enumerate(hdr)))
        cache = {}

        def want_rows(keys, cds, jc):
            out = []
            for i, (N, L) in enumerate(shapes):
                for key in keys or [None]:
                    rows = None if key is None else np.arange(int(key[-1]) - 1, N, 2, dtype=np.uint32)
                    ck = (i, key, cds)
                    if ck not in cache:
                        cache[ck] = c_oracle.scan(mats[i], rows=rows, cds=cds)
                    out.append(orc.row_from_counts(names[i], key or "NA", N if rows is None else len(rows), L, cache[ck], cds, jc))
            return out

        for argv, keys, cds, jc in ((["--cds", "--jc"], None, True, True), ([], None, False, False),
                                    (["-p", "pop1,pop2", "--cds"], ["pop1", "pop2"], True, False)):
            rc, out = run_cli(["-d", d, "-s"] + argv)
            assert rc == 0
            got = out.rstrip("\n").split("\n")
            want = want_rows(keys, cds, jc)
            assert len(got) == len(want), (argv, len(got), len(want))
            bad = [(a, b, C.row_matches(a, b, cds, jc)) for a, b in zip(got, want) if a != b and C.row_matches(a, b, cds, jc)]
            assert not bad, (argv, len(bad), bad[:3])
            text = " | ".join(launched)
            del launched[:]
            assert "scan_tall_kernel" in text and "pfa::scan_kernel<" in text, text
            assert ("wide=1" in text) == (cds and keys is None), text      # 3 000 and 2 049 coding rows; the populations are halves
        # the same alignments on standard input (--pipe: the in-memory batch path, PolyFastA.py:96-100, :230-236)
        for i in (0, 3):
            N, L = shapes[i]
            text = open(os.path.join(d, names[i]), "rb").read()
            rc, out = run_cli(["--pipe", "-n", "big", "-p", "pop2,pop1", "--cds", "--jc", "-s"], stdin_bytes=text, monkeypatch=monkeypatch)
            assert rc == 0
            want = [orc.row_from_counts("big", key, len(range(int(key[-1]) - 1, N, 2)), L, cache[(i, key, True)], True, True) for key in ("pop2", "pop1")]
            got = out.rstrip("\n").split("\n")
            assert len(got) == 2 and not [C.row_matches(a, b, True, True) for a, b in zip(got, want) if a != b and C.row_matches(a, b, True, True)], (got, want)
        # the inputs did what they were meant to: stops counted in the 3 000-row file, every file varies
        first = want_rows(None, True, False)[0].split(",")
        assert int(first[-1]) >= 1 and int(first[4]) == 3000
        assert all(int(r.split(",")[4]) > 0 for r in want_rows(None, False, False))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_one_chromosome_scale_alignment(gpu, tmp_path):
    """Few sequences, tens of millions of sites (the reference has no length limit either, PolyFastA.py:252-261): a file far
    larger than an ingest slot, more 256-site tiles than a grid dimension holds (78 126 > 65 535: several tiles per
    workgroup are forced), coding and not.  Every column of the row against the checker."""
    import ctypes
    import numpy as np
    from oracle import c_oracle
    from oracle import polyfasta_oracle as orc
    from polyfasta_amd import _lib, scan
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (2 << 30) else str(tmp_path)
    d = os.path.join(root, f"pfa_chrom_{os.getpid()}")
    os.makedirs(d)
    try:
        for i, (N, L, cds) in enumerate(((8, 20_000_003, False), (6, 19_999_998, True))):
            p = synth.make_params(N, 0.01)
            stride = scan.row_stride_for(L, 128)
            sp = _lib.SynthParamsC(seed=77 + i, first_aln=0, n_aln=1, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                                   tri_thresh=p.tri_thresh, n_units=len(p.units), coding=int(cds))
            bases = np.zeros(N * stride, dtype=np.uint8)
            _lib.check(_lib.load().pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                                       _lib.ptr(np.ascontiguousarray(p.units))))
            m = bases.reshape(N, stride)[:, :L]
            m[1, L - 1] = ord("N") if m[0, L - 1] != ord("N") else ord("A")      # the very last site varies
            name = os.path.join(d, f"chrom{i}.fa")
            with open(name, "wb") as fh:
                for r in range(N):
                    fh.write(b">s%d\n" % r)
                    fh.write(m[r].tobytes())
                    fh.write(b"\n")
            o = c_oracle.scan(m, cds=cds)
            for jc in (False, True):
                rc, out = run_cli(["-f", name, "-s"] + (["--cds"] if cds else []) + (["--jc"] if jc else []))
                assert rc == 0
                want = orc.row_from_counts(f"chrom{i}.fa", "NA", N, L, o, cds, jc)
                got = out.rstrip("\n")
                assert got == want or C.row_matches(got, want, cds, jc) is None, (got, want)
            assert o["S"] > 100_000
            del bases, m
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_hundred_thousand_tiny_files(gpu, tmp_path):
    """BASELINE config 5's file count with alignments of a few short sequences (2 to 7 rows, 30 to 93 sites; many without a
    single variable site, many with N = 2 or 3 where Tajima's D is "NA", PolyFastA.py:508-511, :522-534): the per-file costs of
    the ingest, plans of tens of thousands of units per batch, the formatter at scale.  Plain and `--cds --jc`: every row."""
    import numpy as np
    from oracle import c_oracle
    from oracle import polyfasta_oracle as orc
    M = 100_000
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (1 << 30) else str(tmp_path)
    d = os.path.join(root, f"pfa_tiny_{os.getpid()}")
    os.makedirs(d)
    try:
        rng = np.random.default_rng(2026)
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        mats = []
        for i in range(M):
            n, L = int(rng.integers(2, 8)), 3 * int(rng.integers(10, 32))
            m = np.repeat(acgt[rng.integers(0, 4, L)][None, :], n, axis=0)
            k = int(rng.integers(0, 4))                       # 0 .. 3 mutated cells: a quarter of the files do not vary
            if k:
                m[rng.integers(0, n, k), rng.integers(0, L, k)] = acgt[rng.integers(0, 4, k)]
            if i % 997 == 0:
                m[0, 0] = ord("N")
            mats.append(m)
            with open(os.path.join(d, f"t{i:06d}.fa"), "wb") as fh:
                fh.write(b"".join(b">s%d\n" % r + m[r].tobytes() + b"\n" for r in range(n)))
        for argv, cds, jc in (([], False, False), (["--cds", "--jc"], True, True)):
            out_path = os.path.join(d, "..", f"pfa_tiny_{os.getpid()}.csv")
            rc, _ = run_cli(["-d", d, "-s", "-o", out_path] + argv)
            assert rc == 0
            got = open(out_path).read().rstrip("\n").split("\n")              # (--silent: no header line)
            os.unlink(out_path)
            assert len(got) == M
            n_na = 0
            for i, (a, m) in enumerate(zip(got, mats)):
                want = orc.row_from_counts(f"t{i:06d}.fa", "NA", m.shape[0], m.shape[1], c_oracle.scan(m, cds=cds), cds, jc)
                assert a == want or C.row_matches(a, want, cds, jc) is None, (i, a, want)
                n_na += a.split(",")[-2 if cds else -1] == "NA"
            assert n_na > M // 4        # no variable site, or two or three sequences
            assert cli.last_stats["files"] == M
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_a_million_sequences_in_one_file(gpu, tmp_path):
    """2^20 sequences of 15 sites in one file (the reference has no row limit, PolyFastA.py:252-261): the record
    index of a million headers, the population split over them (`-p popA,popB`: half a million rows each), 64-bit H,
    the coding pass; every column against the checker.  (More rows than the ABI admits — 2^24 + 1 records — are refused
    loudly, never answered wrongly: test_more_sequences_than_the_abi_admits.)"""
    import numpy as np
    from oracle import c_oracle
    from oracle import polyfasta_oracle as orc
    from polyfasta_amd import _lib
    N, L = 1 << 20, 15
    rng = np.random.default_rng(5)
    m = np.repeat(np.frombuffer(b"ATGGCTAAAGGTCTA", dtype=np.uint8)[None, :], N, axis=0)
    for col, frac, alt in ((2, 0.3, b"A"), (4, 0.001, b"T"), (5, 0.5, b"G"), (9, 0.02, b"C"), (10, 0.02, b"A"), (14, 1e-5, b"N")):
        m[rng.random(N) < frac, col] = ord(alt)
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (1 << 30) else str(tmp_path)
    path = os.path.join(root, f"pfa_million_{os.getpid()}.fa")
    try:
        hdr = [b">pop%s_%d\n" % (b"A" if r % 2 == 0 else b"B", r) for r in range(N)]
        with open(path, "wb") as fh:
            fh.write(b"".join(h + m[r].tobytes() + b"\n" for r, h in enumerate(hdr)))
        name = os.path.basename(path)
        for argv, keys, cds, jc in ((["--cds", "--jc"], None, True, True), (["-p", "popA,popB"], ["popA", "popB"], False, False)):
            rc, out = run_cli(["-f", path, "-s"] + argv)
            assert rc == 0
            got = out.rstrip("\n").split("\n")
            want = []
            for key in keys or [None]:
                rows = None if key is None else np.arange(0 if key == "popA" else 1, N, 2, dtype=np.uint32)
                o = c_oracle.scan(m, rows=rows, cds=cds)
                want.append(orc.row_from_counts(name, key or "NA", N if rows is None else len(rows), L, o, cds, jc))
            assert len(got) == len(want) and all(a == b or C.row_matches(a, b, cds, jc) is None for a, b in zip(got, want)), (got, want)
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_more_sequences_than_the_abi_admits(gpu, tmp_path):
    """PFA_MAX_ROWS + 1 = 2^24 + 1 records in one file (0.45 GB of text, written as one numpy array): the command line ends with the
    library's error about the rows; it does not print a row."""
    import numpy as np
    from polyfasta_amd import _lib
    N = _lib.PFA_MAX_ROWS + 1
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (2 << 30) else str(tmp_path)
    path = os.path.join(root, f"pfa_too_many_{os.getpid()}.fa")
    rec = np.empty((N, 26), dtype=np.uint8)                      # ">" + eight digits + "\n" + 15 bases + "\n"
    rec[:, 0] = ord(">")
    idx = np.arange(N, dtype=np.int64)
    for k in range(8):
        rec[:, 8 - k] = ord("0") + (idx // 10 ** k) % 10
    rec[:, 9] = rec[:, 25] = ord("\n")
    rec[:, 10:25] = np.frombuffer(b"ATGGCTAAAGGTCTA", dtype=np.uint8)
    rec[::7, 12] = ord("A")
    try:
        rec.tofile(path)
        del rec
        buf = io.StringIO()
        with redirect_stdout(buf):
            with pytest.raises(_lib.PfaError, match="rows"):
                cli.main(["-f", path, "-s"])
        assert buf.getvalue() == ""
    finally:
        if os.path.exists(path):
            os.unlink(path)
