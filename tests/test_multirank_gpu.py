"""The multi-rank path with REAL scans: the CLI as two ranks on the one GPU of
the test box gives the single-process CSV byte for byte (row order = sorted paths,
PolyFastA.py:104; population order as typed, :124); the RCCL communicator of the C ABI with one
rank; the launcher's refusal of more ranks than GPUs."""
import io
import os
import subprocess
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

from polyfasta_amd import cli, dist, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_dir(tmp_path, n_files=37, odd=True):
    d = tmp_path / "aln"
    d.mkdir()
    rng = np.random.default_rng(5)
    for i in range(n_files):
        n = int(rng.choice([12, 20, 50, 80]))
        L = int(rng.choice([900, 3000, 9000, 30000])) if i != 5 else 150000      # one heavy file: the split is by size
        p = synth.make_params(n, 0.02)
        m = synth.synth_alignment(4200, i, L, p, coding=True)
        (d / f"g{i:02d}.fa").write_text(synth.to_fasta_text(synth.headers_for(n, 2), m))
    if odd:
        (d / "zz_not_fasta.fa").write_text("no header here\nACGT\n")
        (d / "zz_unequal.fa").write_text(">pop1_a\nACGT\n>pop2_b\nACG\n")
    return d


def _single(argv):
    buf = io.StringIO()
    with redirect_stdout(buf):
        assert cli.main(argv) == 0
    return buf.getvalue()


@pytest.mark.parametrize("extra,world", [([], 2), (["--cds", "--jc", "-p", "pop1,pop2"], 2), (["--sfs", "-p", "pop2,nopop,pop1"], 2),
                                         ([], 3), (["--cds", "-p", "pop1,nopop"], 3)])
def test_two_rank_cli_equals_single_process(gpu, tmp_path, extra, world):
    """(two ranks, and three: uneven shares of the queue, two non-root ranks to merge on rank 0)"""
    d = _make_dir(tmp_path)
    argv = ["-d", str(d), "-s"] + extra
    want = _single(argv)
    assert want.count("\n") > 37
    out = tmp_path / "rank0.csv"
    port = dist._free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_cli_rank_worker.py"), str(out)] + argv,
                                      env=env, cwd=ROOT, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-3000:]
    assert open(out).read() == want


@pytest.mark.parametrize("extra", [[], ["--cds", "--jc", "-p", "pop1,pop2"]])
def test_two_rank_cli_regular_directory_rows_formatted_by_their_ranks(gpu, tmp_path, extra):
    """A directory of regular files only: every rank formats its own rows as runs of consecutive files and rank 0
    puts the runs in file order (cli.text_runs) — byte for byte the single process's CSV."""
    d = _make_dir(tmp_path, n_files=61, odd=False)
    argv = ["-d", str(d), "-s"] + extra
    want = _single(argv)
    assert want.count("\n") >= 61
    out = tmp_path / "rank0.csv"
    port = dist._free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_cli_rank_worker.py"), str(out)] + argv,
                                      env=env, cwd=ROOT, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-3000:]
    assert open(out).read() == want


def test_two_rank_cli_writes_the_out_file_like_one_process(gpu, tmp_path):
    """`-o out.csv` without `-s` as two ranks: header (mode "w") and rows in the file, the progress chatter and the
    trailing newline on rank 0's stdout only (PolyFastA.py:103, :141-142), byte for byte what one process leaves."""
    d = _make_dir(tmp_path, n_files=21)
    one = tmp_path / "one.csv"
    want_stdout = _single(["-d", str(d), "-o", str(one)])
    two = tmp_path / "two.csv"
    out0 = tmp_path / "rank0_stdout.txt"
    port = dist._free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_cli_rank_worker.py"), str(out0), "-d", str(d), "-o", str(two)],
                                      env=env, cwd=ROOT, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-3000:]
    assert open(two).read() == open(one).read()
    got_stdout = open(out0, newline="").read()           # (universal newlines would turn the chatter's "\r" into "\n")
    assert got_stdout == want_stdout.replace(str(one), str(two)) and want_stdout.endswith("\n") and "\r Writting to" in got_stdout


@pytest.mark.parametrize("extra", [[], ["--cds", "-p", "pop1,pop2"]])
def test_two_rank_cli_dies_at_an_unreadable_entry_like_one_process(gpu, tmp_path, extra):
    """A subdirectory among the files of `--dir`, two ranks: rank 0 prints every row of the files before it — whichever rank
    scanned them — and dies with the reference's error (PolyFastA.py:104-108, :239: IsADirectoryError, exit code 1); the
    other rank leaves with the same code; rows of later files never appear.  Byte for byte the single process's output."""
    d = _make_dir(tmp_path, n_files=31, odd=False)
    (d / "g17_is_a_directory").mkdir()
    argv = ["-d", str(d), "-s"] + extra
    buf = io.StringIO()
    with redirect_stdout(buf):
        with pytest.raises(IsADirectoryError):
            cli.main(argv)
    want = buf.getvalue()
    assert 17 <= want.count("\n") <= 2 * 18 and "g18.fa" not in want and "g16.fa" in want
    out = tmp_path / "rank0.csv"
    port = dist._free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_cli_rank_worker.py"), str(out)] + argv,
                                      env=env, cwd=ROOT, stderr=subprocess.PIPE))
    errs = [p.communicate(timeout=600)[1].decode() for p in procs]
    assert [p.returncode for p in procs] == [1, 1], errs
    assert "IsADirectoryError" in errs[0] and "g17_is_a_directory" in errs[0] and "IsADirectoryError" not in errs[1]
    assert open(out).read() == want


def test_rccl_comm_one_rank(gpu, tmp_path):
    """pfa_comm_* on librccl with a world of one: unique id through the rendezvous file, barrier,
    max-reduction, host-staged all-gather, and the device-side gather of a plan's records."""
    code = """
import os, sys, numpy as np
sys.path.insert(0, %r)
from polyfasta_amd import _lib, dist, scan, synth
c = dist.RcclComm()
assert (c.rank, c.world) == (0, 1)
c.barrier()
assert c.max_float(2.5) == 2.5
assert c.all_gather_bytes(b"hello") == [b"hello"]
p = synth.make_params(20, 0.02)
b = scan.Batch(stride_align=128)
for i in range(3):
    b.add_unit(b.add_matrix(synth.synth_alignment(1, i, 1000, p)))
bases, table = b.pack()
dev = scan.DeviceBuffer(len(bases)); dev.upload(bases)
plan = scan.Plan(table)
plan.run(dev)
got = c.all_gather_plan_results(plan)
out = plan.fetch()
assert got.shape == (1, 3) and (got[0] == out.results).all() and out.results["seg_sites"].min() > 0
c.close()
print("rccl-one-rank-ok")
""" % ROOT
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", PFA_RDZV_FILE=str(tmp_path / "id"))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "rccl-one-rank-ok" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


def test_gpus_flag_fails_loudly_without_enough_gpus(gpu, tmp_path):
    """`--gpus 4` on a box with fewer GPUs: non-zero exit, a message, no rows (never a silent 1-GPU run)"""
    if gpu.load().pfa_device_count() >= 4:
        pytest.skip("needs a box with fewer than 4 GPUs")
    d = _make_dir(tmp_path, n_files=3)
    p = subprocess.run([sys.executable, "-m", "polyfasta_amd", "-d", str(d), "--gpus", "4"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    # refused by the launcher (readable kfd topology) or by the ranks whose GPU is not there
    assert p.returncode != 0 and ("--gpus 4" in p.stderr or "out of range" in p.stderr) and "g00.fa" not in p.stdout
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--batch", "8"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and '"n_gpus"' not in p.stdout


def test_cli_through_rccl_with_one_rank(gpu, tmp_path):
    """the CLI's gather (records, synonymous bins, pickled metadata) through the real RCCL communicator,
    one rank (PFA_FORCE_DIST): the CSV of the plain process, byte for byte"""
    d = _make_dir(tmp_path, n_files=12)
    argv = ["-d", str(d), "-s", "--cds", "--jc", "-p", "pop1,pop2", "--sfs"]
    want = _single(argv)
    env = dict(os.environ, PFA_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", PFA_RDZV_FILE=str(tmp_path / "id"))
    p = subprocess.run([sys.executable, "-m", "polyfasta_amd"] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert p.stdout == want


def test_bench_with_two_ranks_on_one_gpu(gpu):
    """bench.py --gpus 2 end to end on the one GPU (its own launcher, both ranks on GPU 0, a gloo communicator
    injected because RCCL refuses two ranks per device): ONE JSON line, n_gpus 2, the records of both ranks
    gathered, rank 0's share equal to its own fetch"""
    import json
    env = dict(os.environ, PFA_ALLOW_SHARED_GPU="1", PFA_COMM_FACTORY="tests._gloo_comm:GlooComm", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "512",
                        "--e2e-files-per-rank", "96", "--e2e-reps", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3
    assert j["check"]["gathered_records"] == 2 * 512 and j["check"]["gathered_rank0_equals_fetch"] is True
    assert j["value"] > 0 and "cpu_baseline" not in j and "workloads" not in j
    # the N > 1 end-to-end leg: both ranks drain the shared queue of one directory, rows gathered to rank 0 and oracle-checked
    e = j["end_to_end"]
    assert e["ranks"] == 2 and e["files"] == 192 and e["rows_ok"] is True and e["best"] > 0, e
    assert j["rehearsal"]["shared_gpu"] is True
    # the stage table of the best run: one column per rank, every stage of the multi-rank path
    st = e["stages_per_rank_of_best_run_s"]
    for key in ("pre", "open", "q_open", "claims", "claim", "files", "batches", "threads", "read_thr", "pack_thr", "slot_wait_thr", "h2d_busy",
                "next_wait", "scan_wait", "block", "format", "flags_gather", "gather", "gather_B", "root_write", "barrier", "flush", "close",
                "run_files", "main", "wall"):
        assert key in st and len(st[key]) == 2, key
    assert sum(st["files"]) == 192 and min(st["claims"]) >= 1 and max(st["wall"]) <= e["s"] * 1.001 + 1e-6
    # rank 0 alone on the same files (the single-process path), same rows
    assert e["solo_rank0_same_files"]["rows_equal"] is True and e["solo_rank0_same_files"]["best"] > 0
    assert e["host_read_ceiling_alignments_per_s"] > 1e4 and set(j["roofline"]["end_to_end"]) == {"ranks2", "solo"}
    assert len(lines[0]) < 8192


def test_bench_prints_its_line_when_the_multi_rank_end_to_end_leg_hangs(gpu):
    """The N > 1 end-to-end leg runs under a watchdog: when it does not come back (here: a 50 ms allowance), rank 0
    still prints the ONE JSON line with the headline it measured before, the leg reported as an error with the stage that
    was outstanding, and every rank exits NON-ZERO (124): a hung multi-rank job never reads as a clean run."""
    import json
    env = dict(os.environ, PFA_ALLOW_SHARED_GPU="1", PFA_COMM_FACTORY="tests._gloo_comm:GlooComm", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "512",
                        "--e2e-files-per-rank", "512", "--e2e-reps", "2", "--e2e-ranks-timeout", "0.05"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 124, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "did not finish" in j["end_to_end"]["error"] and j["end_to_end"]["stage"]
    assert "rank 1 of 2" in p.stderr and "last stage:" in p.stderr
    import glob
    assert not glob.glob("/dev/shm/pfa_e2e_ranks_*"), "the leg's tmpfs directory was left behind"


@pytest.mark.parametrize("extra", [["--cds", "--jc", "-p", "pop1,pop2"], []])
def test_three_rank_cli_over_a_heterogeneous_directory(gpu, tmp_path, extra):
    """Three ranks draining one queue of very unlike files — 300 tiny ones, alignments of 700 and 3 000 sequences (the
    row-chunked kernel and its 56-bit coding form on whichever rank claims them), an empty file, a file that is not FASTA, one
    with unequal lengths: the single process's CSV byte for byte (PolyFastA.py:104, :117-140)."""
    d = _make_dir(tmp_path, n_files=23)
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(300):
        n, L = int(rng.integers(2, 6)), 3 * int(rng.integers(10, 30))
        m = np.repeat(acgt[rng.integers(0, 4, L)][None, :], n, axis=0)
        m[rng.integers(0, n), rng.integers(0, L)] = ord("A")
        (d / f"h{i:03d}.fa").write_bytes(b"".join(b">pop%d_s%d\n" % (1 + r % 2, r) + m[r].tobytes() + b"\n" for r in range(n)))
    for name, n, L in (("big_a.fa", 700, 6000), ("big_b.fa", 3000, 3000)):
        p = synth.make_params(n, 0.01)
        (d / name).write_text(synth.to_fasta_text(synth.headers_for(n, 2), synth.synth_alignment(77, n, L, p, coding=True)))
    (d / "empty.fa").write_bytes(b"")
    argv = ["-d", str(d), "-s"] + extra
    want = _single(argv)
    assert want.count("\n") >= 325 and "# file" in want
    out = tmp_path / "rank0.csv"
    port = dist._free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_cli_rank_worker.py"), str(out)] + argv,
                                      env=env, cwd=ROOT, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-3000:]
    assert open(out).read() == want
