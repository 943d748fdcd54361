#!/usr/bin/env python3
"""bench.py — the polymorphism-scan hot path on N GPUs of one node.

A *step* is one pass of the scan kernel over one device-resident batch of synthetic
alignments (BASELINE.json configs[1] shape: N=50 sequences x L=10 kb, theta = 0.01,
non-coding; `--batch` of them per GPU, distinct, generated on the device by the
deterministic generator, so the working set is many times the 256 MiB Infinity Cache).
Multi-GPU = the reference's `--dir` batch mode: independent alignments sharded over ranks
(weak scaling: every rank scans its own `--batch` alignments per step), no collective on
the data path, ONE RCCL all-gather of the fixed-size result records at the end, inside the
timed region (pfa_comm_* of the C ABI: no torch anywhere).

`--gpus N` without a launcher's WORLD_SIZE starts the N ranks itself (one process per GPU,
before anything here touches the GPU) and fails loudly when fewer GPUs are visible; under
`python -m torch.distributed.run` the ranks are read from RANK / LOCAL_RANK / WORLD_SIZE.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline`
(algorithmic bytes = N*L per alignment, kernel time from HIP events on the launch stream),
and at N=1 also `cpu_baseline` (the C oracle and the pure-Python restatement on the host
cores, bounded samples), `workloads` (the other BASELINE shapes, device-resident, each with
its own roofline fraction), `single_alignment` / `config3_one_launch` (latency of small
launches) and `end_to_end` (FASTA files in tmpfs -> CSV rows through the CLI).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

# (numpy's import starts one spinning OpenBLAS thread per CPU the machine shows: 2 CPU-seconds on a 256-CPU box with a quota of 16;
# nothing here uses BLAS — polyfasta_amd/__main__.py does the same for the command line)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# read() out of the page cache into a loader's buffer: GB/s per CPU a rank may keep busy (profiles/r04_read_vs_mmap_8proc.txt: 16 threads inside a
# quota of 16 CPUs hold 145-180 GB/s however they are split over processes; the "220 GB/s host ceiling" of round 3 was that quota, not the memory system)
HOST_READ_GBPS_PER_CPU = 10.0
DETAIL = {}             # the long form of every leg (prose, every counter): one `BENCH_DETAIL {...}` line on stderr


def rnd(x, n=4):
    """floats of the JSON line rounded to `n` significant digits (the line stays under 8 KB; full precision goes to stderr)"""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: rnd(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rnd(v, n) for v in x]
    return x


def rows_equal(got, want, jc):
    """two CSV rows field by field: equal text, or floats within 1e-9 relative (JC-corrected pi columns 1e-6; Tajima's D, a
    cancelling difference, also within 1e-9 absolute)"""
    fa, fb = got.split(","), want.split(",")
    if len(fa) != len(fb):
        return False
    for i, (x, y) in enumerate(zip(fa, fb)):
        if x == y:
            continue
        try:
            fx, fy = float(x), float(y)
        except ValueError:
            return False
        tol = 1e-6 if jc else 1e-9
        if not (abs(fx - fy) <= tol * max(abs(fx), abs(fy)) or abs(fx - fy) <= 1e-9):
            if len(fa) == 14 and i in (1, 2) and abs(fx - fy) <= 0.010000001:
                continue     # sites_S / sites_N: round(x, 2) of an exact tie
            return False
    return True


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32768, help="alignments per GPU per step (16.6 GB of bases at the default shape)")
    ap.add_argument("--rows", type=int, default=50)
    ap.add_argument("--sites", type=int, default=10000)
    ap.add_argument("--theta", type=float, default=0.01)
    ap.add_argument("--stride-align", type=int, default=128)
    ap.add_argument("--variant", type=int, default=0, help="0 = LDS-DMA nt (default), 2 = LDS-DMA default cache policy")
    ap.add_argument("--coding", action="store_true", help="scan as in-frame CDS (config 4 shape with --sites 9000)")
    ap.add_argument("--pops", type=int, default=0,
                    help="config 3 shape: every matrix holds rows*pops sequences, scanned as `pops` interleaved row subsets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline measurement only (profiling runs)")
    ap.add_argument("--legs", default="small_launches,workloads,end_to_end,cold_start",
                    help="which extra legs of the default run to take (comma-separated)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--e2e-files", type=int, default=16384, help="FASTA files per end-to-end directory (halved until tmpfs has room)")
    ap.add_argument("--e2e-files-per-rank", type=int, default=4096, help="files per rank of the N > 1 end-to-end leg (weak scaling)")
    ap.add_argument("--e2e-ranks-timeout", type=float, default=240.0, help="watchdog of the N > 1 end-to-end leg (seconds)")
    ap.add_argument("--e2e-reps", type=int, default=5, help="warm runs per end-to-end mode (best and median are reported)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ workloads
class Workload:
    """A device-resident batch of synthetic alignments and its scan plan."""

    def __init__(self, rows, sites, batch, theta=0.01, coding=False, pops=0, stride_align=128, first_aln=0, timing=True, split=0):
        """`pops`: row-index lists over one matrix (the ABI's pfa_desc.row_idx form).  `split`: the form the command line produces for
        --pops (csrc/pfa_ingest.hip): every population a matrix of its own — `split` matrices of `rows` rows per alignment."""
        if split:
            batch, pops = batch * split, 0
        from polyfasta_amd import _lib, scan, synth
        lib = _lib.load()
        self.N, self.L, self.B, self.K = rows, sites, batch, max(pops, 1)
        self.coding, self.pops, self.split = coding, pops, split
        NM = rows * self.K
        self.params = synth.make_params(NM, theta)
        self.stride = scan.row_stride_for(sites, stride_align)
        self.nbytes = batch * NM * self.stride
        self.dev = scan.DeviceBuffer(self.nbytes)
        sp = _lib.SynthParamsC(seed=2026, first_aln=first_aln, n_aln=batch, n_rows=NM, n_sites=sites, row_stride=self.stride,
                               seg_thresh=self.params.seg_thresh, tri_thresh=self.params.tri_thresh,
                               n_units=len(self.params.units), coding=int(coding))
        _lib.check(lib.pfa_synth_fill_dev(self.dev.ptr, ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(self.params.cdf)),
                                          _lib.ptr(np.ascontiguousarray(self.params.units))))
        descs = np.zeros(batch * self.K, dtype=_lib.DESC_DTYPE)
        descs["base_off"] = np.repeat(np.arange(batch, dtype=np.uint64) * np.uint64(NM * self.stride), self.K)
        descs["n_rows"], descs["n_sites"], descs["row_stride"], descs["mat_rows"] = rows, sites, self.stride, NM
        descs["row_idx_off"] = _lib.PFA_NO_ROW_IDX
        descs["flags"] = _lib.PFA_FLAG_CDS if coding else 0
        row_idx = None
        if pops:                                 # population k = rows k, k+K, k+2K, ... (headers pop{1+i%K}_ind{i})
            row_idx = np.concatenate([np.arange(k, NM, self.K, dtype=np.uint32) for k in range(self.K)])
            descs["row_idx_off"] = np.tile(np.arange(self.K, dtype=np.uint32) * rows, batch)
        self.descs, self.row_idx = descs, row_idx
        self.table = scan.finalize_layout(descs, row_idx, self.nbytes)
        self.plan = scan.Plan(self.table, timing=timing)
        self.alg_bytes = float(batch) * NM * sites      # every base of every population is read once

    def label(self):
        s = f"{self.B} distinct alignments, N={self.N} x L={self.L}, {'coding (--cds)' if self.coding else 'non-coding'}"
        if self.split:
            s = (f"{self.B // self.split} alignments of {self.N * self.split} rows x L={self.L} as the command line lays --pops out: {self.split} populations of {self.N} "
                 f"rows, each a matrix of its own ({self.B} units), {'coding (--cds)' if self.coding else 'non-coding'}")
        if self.pops:
            s += f", {self.K} populations of {self.N} rows per {self.N * self.K}-row matrix (--pops)"
        return s

    def measure(self, steps, warmup, variant=0):
        """kernel-only figures from HIP events around the launches of `steps` runs"""
        from polyfasta_amd import _lib
        # warm by TIME as well as by count: after some tens of milliseconds without work the card drops to a low-power state, and the
        # launches that follow run 5-12 % slower until its clocks are back (profiles/r05_launch_spread_*: coding N = 5 000 back to back
        # 7.48-7.64 ms, with 30 ms of idle between launches 7.85-8.61) — what round 4 saw as "0.71-0.77 between runs"
        t_warm = time.perf_counter()
        n_warm = 0
        while n_warm < warmup or (time.perf_counter() - t_warm < 0.08 and n_warm < 400):
            self.plan.run(self.dev, variant=variant)
            n_warm += 1
            if n_warm >= warmup:
                _lib.check(_lib.load().pfa_sync())
        for _ in range(steps):
            self.plan.run(self.dev, variant=variant)
        _lib.check(_lib.load().pfa_sync())
        kms = self.plan.kernel_ms(steps)
        avg = float(kms.mean())
        ach = self.alg_bytes / (avg * 1e-3) / 1e9
        return {"workload": self.label(), "kernel": self.plan.description, "launches_per_run": self.plan.n_launches,
                "kernel_ms_avg": avg, "kernel_ms_min": float(kms.min()), "kernel_ms_p50": float(np.percentile(kms, 50)),
                "kernel_ms_p95": float(np.percentile(kms, 95)), "timed_runs": int(len(kms)), "algorithmic_bytes_per_launch": self.alg_bytes,
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "frac_of_measured_copy_6290": ach / 6290.0,
                # scan units (alignment x population = CSV rows) per second of kernel time
                "alignments_per_s_kernel": self.B * self.K / (avg * 1e-3), "site_columns_per_s_kernel": self.B * self.K * self.L / (avg * 1e-3)}

    def close(self):
        self.plan.close()
        self.dev.free()


WORKLOAD_KEYS = {"C4_coding_50x9000": "C4", "C3_pops_2x100x100000": "C3_split", "C3_pops_row_index_lists": "C3_rowidx",
                 "N200_plain_200x100000": "N200", "N200_coding_200x9000": "N200_cds", "N1000_coding_1000x9000": "N1000_cds",
                 "N5000_coding_5000x9000": "N5000_cds", "C5_batch_100000x50x10000": "C5_100k",
                 "C3_pops_2x100x100000_x4": "C3_split_b512", "N200_plain_200x100000_x4": "N200_b512"}


def extra_workloads(args):
    """The other BASELINE shapes as short device-resident passes (kernel time from HIP events)."""
    out = {}
    for key, kw in (("C4_coding_50x9000", dict(rows=50, sites=9000, batch=32768, coding=True)),
                    # BASELINE configs[2] (200 x 100 kb, --pops pop1,pop2): as the CLI's ingest lays it out (one matrix per population), and
                    # through the ABI's row-index lists over one 200-row matrix
                    ("C3_pops_2x100x100000", dict(rows=100, sites=100000, batch=128, split=2)),
                    ("C3_pops_row_index_lists", dict(rows=100, sites=100000, batch=128, pops=2)),
                    ("N200_plain_200x100000", dict(rows=200, sites=100000, batch=128)),
                    ("N200_coding_200x9000", dict(rows=200, sites=9000, batch=8192, coding=True)),
                    # units with many rows: the row-chunked kernel (and its wide form above 2 048 coding rows)
                    # (batches sized so that the launch is several rounds of workgroups deep: a coding chunk is 3 tiles x all rows,
                    # 3.8 MB at N = 5 000, and 256 units would be 1.3 rounds of the chip's 2 304 wavefront slots)
                    ("N1000_coding_1000x9000", dict(rows=1000, sites=9000, batch=2048, coding=True)),
                    ("N5000_coding_5000x9000", dict(rows=5000, sites=9000, batch=1024, coding=True)),
                    # BASELINE configs[4] at its full size on ONE GPU: 100 000 alignments = 50 GB of bases resident in HBM
                    ("C5_batch_100000x50x10000", dict(rows=50, sites=10000, batch=100000)),
                    # The 128-unit config-3 / N = 200 passes above are 2.56 GB = 0.4 ms launches: ramp-up and the last round of
                    # workgroups cost them ~5 points (profiles/r04_tiles_per_chunk_and_batch_sweep.txt).  The same kernels on 512
                    # units (10 GB, a 1.6 ms launch; the headline pass is 16 GB):
                    ("C3_pops_2x100x100000_x4", dict(rows=100, sites=100000, batch=512, split=2)),
                    ("N200_plain_200x100000_x4", dict(rows=200, sites=100000, batch=512))):
        try:
            w = Workload(theta=args.theta, stride_align=args.stride_align, **kw)
        except Exception as e:                  # noqa: BLE001  (e.g. not enough free HBM for the 50 GB batch)
            out[key] = {"error": f"{type(e).__name__}: {e}"}
            continue
        full = w.measure(steps=24 if kw["batch"] < 100000 else 10, warmup=6 if kw["batch"] < 100000 else 2)
        o = w.plan.fetch()
        full["check"] = {"mean_seg_sites": float(o.results["seg_sites"].mean()), "sum_H": int(o.results["H"].sum())}
        DETAIL.setdefault("workloads", {})[key] = full
        out[key] = {"frac": full["frac"], "ms": full["kernel_ms_avg"], "p50": full["kernel_ms_p50"], "p95": full["kernel_ms_p95"],
                    "GBps": full["achieved"], "kernel": full["kernel"].split(" units=")[0].replace("pfa::", ""), "sum_H": full["check"]["sum_H"]}
        w.close()
    return out


def small_launches(hw, args):
    """BASELINE configs[1] and [2] literally: ONE alignment per launch (latency-bound: 500 KB of
    bases), and config 3 (200 x 100 kb, two populations) as one launch of two units."""
    from polyfasta_amd import _lib, scan
    lib = _lib.load()
    rep = {}
    one = scan.Plan(scan.finalize_layout(hw.descs[:1].copy(), None, hw.nbytes))
    for _ in range(20):
        one.run_fetch(hw.dev)
    t1 = time.perf_counter()
    for _ in range(500):
        one.run_fetch(hw.dev)
    per = (time.perf_counter() - t1) / 500
    # the same through the C ABI alone: caller-owned output arrays made once, one pfa_plan_run_fetch() per scan
    res, sfs = np.empty(1, dtype=_lib.RESULT_DTYPE), np.empty(max(hw.N // 2, 1), dtype=np.uint32)
    a_plan, a_dev, a_res, a_sfs = one.handle, hw.dev.ptr, _lib.ptr(res), _lib.ptr(sfs)
    t1 = time.perf_counter()
    for _ in range(500):
        lib.pfa_plan_run_fetch(a_plan, a_dev, 0, a_res, a_sfs, None, None, None, None)
    per_abi = (time.perf_counter() - t1) / 500
    _lib.check(lib.pfa_plan_set_timing(one.handle, 1))     # HIP events around the launch: kernel time alone
    for _ in range(60):
        one.run_fetch(hw.dev)
    k = one.kernel_ms(50)
    # one N x L alignment per launch, bases resident, records fetched after every scan
    rep["single_alignment"] = {"us_per_scan_and_fetch": per * 1e6, "us_per_scan_and_fetch_c_abi": per_abi * 1e6, "kernel_us": float(k.mean()) * 1e3, "kernel_us_min": float(k.min()) * 1e3}
    DETAIL["single_alignment_plan"] = one.description
    one.close()
    c3 = Workload(rows=100, sites=100000, batch=1, pops=2, theta=args.theta, stride_align=args.stride_align)
    for _ in range(20):
        c3.plan.run_fetch(c3.dev)
    k = c3.plan.kernel_ms(16)
    # BASELINE configs[2] literally: one 200 x 100 kb matrix, two 100-row populations, one launch
    rep["config3_one_launch"] = {"kernel_us": float(k.mean()) * 1e3, "kernel_us_min": float(k.min()) * 1e3,
                                 "bytes": c3.alg_bytes, "achieved_GBps": c3.alg_bytes / (float(k.mean()) * 1e-3) / 1e9}
    DETAIL["config3_one_launch_plan"] = c3.plan.description
    c3.close()
    return rep


def sample_indices(M, k=48):
    """`k` file indices strided over the whole directory, first and last included"""
    return sorted(set(int(x) for x in np.linspace(0, M - 1, min(k, M))))


def check_rows(rows, mats, idx, M, L, flags, name_of, n_rows=50):
    """The rows of the sampled files (every column) against the checker: integers and text equal, floats to 1e-9 (JC pi 1e-6);
    row count and the name column of EVERY row.  `mats[j]` = matrix of file idx[j]."""
    from oracle import c_oracle
    from oracle import polyfasta_oracle as orc
    cds, jc = "--cds" in flags, "--jc" in flags
    keys = flags[flags.index("-p") + 1].split(",") if "-p" in flags else None
    per = len(keys) if keys else 1
    ok = len(rows) == M * per and [r.split(",", 1)[0] for r in rows[::per]] == [name_of(m) for m in range(M)]
    if not ok:
        return False
    from polyfasta_amd import synth
    hdr = synth.headers_for(n_rows, 2 if keys else 0)
    for j, i in enumerate(idx):
        for k in range(per):
            sub = [r for r, h in enumerate(hdr) if keys[k] in h] if keys else None          # PolyFastA.py:125
            o = c_oracle.scan(mats[j], rows=sub, cds=cds)
            want = orc.row_from_counts(name_of(i), keys[k] if keys else "NA", len(sub) if keys else n_rows, L, o, cds, jc)
            if not rows_equal(rows[i * per + k], want, jc):
                return False
    return True


def end_to_end(args):
    """FASTA files in tmpfs -> CSV rows through the CLI (`python -m polyfasta_amd -d DIR -o out.csv -s [flags]`,
    in-process), per mode: plain on BASELINE configs[1] files (50 x 10 kb), `--cds --jc`, `--pops pop1,pop2` and
    `--pops pop1,pop2 --cds --jc` on in-frame 50 x 9 kb files with pop1_* / pop2_* headers.  Every mode: best and
    median of the warm runs, a stage breakdown from the ingest's own counters, and the rows of 48 files strided over the
    whole directory (first and last included), every column, against the checker (C oracle integers + its own finishing math)."""
    import io
    import shutil
    import tempfile
    from contextlib import redirect_stdout
    from polyfasta_amd import _lib, cli, scan, synth
    lib = _lib.load()
    N = 50
    # as many files as tmpfs has room for (the start-up of a run — argparse, listdir, sorted, the path table: ~9 ms — is a fixed
    # cost; BASELINE configs[4] is a 100 000-file directory), the same number for both directories
    root0, M1 = None, 0
    for m in (args.e2e_files, args.e2e_files // 2, args.e2e_files // 4, args.e2e_files // 8):
        for r in ("/dev/shm", tempfile.gettempdir()):
            if not M1 and m >= 64 and os.path.isdir(r) and shutil.disk_usage(r).free > 1.5 * 1.05 * N * m * (10010 + 9010):
                root0, M1 = r, m
    if not M1:
        return {"skipped": f"neither /dev/shm nor {tempfile.gettempdir()} has room for {args.e2e_files // 8} + {args.e2e_files // 8} FASTA files"}
    M2 = M1
    root = tempfile.mkdtemp(prefix="pfa_e2e_", dir=root0)
    out = os.path.join(tempfile.gettempdir(), f"pfa_e2e_out_{os.getpid()}.csv")
    # in-process, page-cached files in tmpfs; warm = the pinned ring and the device buffers of the process exist, as for every
    # batch after a job's first
    report = {"files": M1, "scratch": root0}
    detail = DETAIL.setdefault("end_to_end", {"command": "python -m polyfasta_amd -d DIR -o out.csv -s [mode flags] (in-process)", "target_alignments_per_s": 10000,
                                              "pcie_h2d_peak_GBps_measured": 57.5})
    try:
        def write_dir(name, M, L, coding, pops):
            """M files of N x L in a fresh directory, generated 2 048 at a time; returns the directory, the sampled file indices and their matrices"""
            d = os.path.join(root, name)
            os.makedirs(d)
            p = synth.make_params(N, args.theta)
            stride = scan.row_stride_for(L, 128)
            hdr = [f">{h}\n".encode() for h in synth.headers_for(N, pops)]
            idx, keep = sample_indices(M), {}
            for m0 in range(0, M, 2048):
                mc = min(2048, M - m0)
                sp = _lib.SynthParamsC(seed=7, first_aln=m0, n_aln=mc, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                                       tri_thresh=p.tri_thresh, n_units=len(p.units), coding=int(coding))
                bases = np.zeros(mc * N * stride, dtype=np.uint8)
                _lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                                   _lib.ptr(np.ascontiguousarray(p.units))))
                for m in range(mc):
                    mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
                    with open(os.path.join(d, f"aln{m0 + m:06d}.fa"), "wb") as fh:
                        fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
                    if m0 + m in idx:
                        keep[m0 + m] = mat.copy()
            return d, idx, [keep[i] for i in idx]

        t0 = time.perf_counter()
        d1, i1, b1 = write_dir("plain", M1, 10000, False, 0)
        d2, i2, b2 = write_dir("coding", M2, 9000, True, 2)
        detail["write_files_s"] = time.perf_counter() - t0

        modes = (("plain", d1, i1, b1, M1, 10000, []), ("cds_jc", d2, i2, b2, M2, 9000, ["--cds", "--jc"]),
                 ("pops", d2, i2, b2, M2, 9000, ["-p", "pop1,pop2"]), ("pops_cds_jc", d2, i2, b2, M2, 9000, ["-p", "pop1,pop2", "--cds", "--jc"]))
        for name, d, idx, mats, M, L, flags in modes:
            times, stats = [], None
            for rep in range(1 + args.e2e_reps):
                if os.path.exists(out):
                    os.unlink(out)
                t0 = time.perf_counter()
                with redirect_stdout(io.StringIO()):
                    rc = cli.main(["-d", d, "-o", out, "-s"] + flags)
                times.append(time.perf_counter() - t0)
                if rep and times[-1] == min(times[1:]):
                    stats = dict(cli.last_stats)
            rows = open(out).read().splitlines()
            os.unlink(out)
            warm = sorted(times[1:])
            best, med = warm[0], warm[len(warm) // 2]
            text_bytes = sum(os.path.getsize(os.path.join(d, f"aln{m:06d}.fa")) for m in (0, M - 1)) / 2 * M
            nt = max(stats["loader_threads"], 1)
            report[name] = {
                "best": M / best, "median": M / med, "s": best, "first_s": times[0], "GBps": text_bytes / best / 1e9, "rows": len(rows),
                "rows_ok": rc == 0 and check_rows(rows, mats, idx, M, L, flags, lambda m: f"aln{m:06d}.fa"),
                # where the best run's time went (stages overlap: loaders, link and card run concurrently); seconds; per loader thread where it says _thr
                "stages": {"startup": best - stats["wall_s"] + stats.get("open_s", 0.0), "read_thr": stats["read_thread_s"] / nt,
                           "pack_thr": stats["index_thread_s"] / nt, "slot_wait_thr": stats["turn_wait_thread_s"] / nt, "h2d_busy": stats["copy_busy_s"],
                           "scan_wait": stats["fetch_wait_s"], "next_wait": stats.get("next_wait_s", 0.0), "rows": stats["rows_s"] + stats["flush_s"],
                           "close": stats.get("close_s", 0.0), "batches": stats["batches"]}}
            detail[name] = {"flags": " ".join(flags), "files": M, "shape": f"{N} x {L}", "times_s": times, "stats_of_best_run": stats,
                            "read_GBps_all_threads": stats["bytes_read"] / max(stats["read_thread_s"] / nt, 1e-9) / 1e9,
                            "h2d_GBps_while_busy": stats["bytes_copied"] / max(stats["copy_busy_s"], 1e-9) / 1e9,
                            "transport_bytes / text_bytes": stats["bytes_copied"] / max(stats["bytes_read"], 1), "checked_files": idx}
            report["loader_threads"] = nt
        return report
    finally:
        shutil.rmtree(root, ignore_errors=True)
        if os.path.exists(out):
            os.unlink(out)


def e2e_ranks_paths():
    """(tmpfs directory, CSV file) of the N > 1 end-to-end leg: names every rank of one launcher derives alike"""
    import tempfile
    root = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), f"pfa_e2e_ranks_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
    return root, os.path.join(tempfile.gettempdir(), f"pfa_e2e_ranks_out_{os.getppid()}.csv")


RANK_STAGES = (  # (short key in the line, key in cli.last_stats, per loader thread?)
    ("pre", "main_pre_s", 0), ("open", "open_s", 0), ("q_open", "queue_open_s", 0), ("claims", "claims", 0), ("claim", "claim_s", 0),
    ("files", "claimed_files", 0), ("batches", "batches", 0), ("threads", "loader_threads", 0), ("read_thr", "read_thread_s", 1),
    ("pack_thr", "index_thread_s", 1), ("slot_wait_thr", "turn_wait_thread_s", 1), ("h2d_busy", "copy_busy_s", 0),
    ("next_wait", "next_wait_s", 0), ("scan_wait", "fetch_wait_s", 0), ("block", "rows_s", 0), ("format", "format_s", 0),
    ("flags_gather", "flags_gather_s", 0), ("gather", "gather_s", 0), ("gather_B", "gather_bytes", 0), ("root_write", "root_merge_write_s", 0),
    ("barrier", "barrier_s", 0), ("flush", "flush_s", 0), ("close", "close_s", 0), ("run_files", "wall_s", 0), ("main", "main_s", 0))

ranks_progress = {"stage": "not started"}     # what the N > 1 leg was doing last (the watchdog's report)


def end_to_end_ranks(args, comm, rank, world):
    """The thing north_star shards, measured: `--dir` through the shared work queue on `world` ranks.  Every rank
    writes its share of one tmpfs directory (N = 50 x 10 kb files, `--e2e-files-per-rank` per rank: weak scaling),
    then all ranks run `python -m polyfasta_amd -d DIR -o out.csv -s` in-process on this bench's communicator: each drains
    dist.WorkQueue, the rows travel to rank 0 once, rank 0 orders and writes them.  The timed region is the whole
    cli.main() of the slowest rank (the gather and rank 0's emit are inside).  Reported with it: the stage table of the best
    run, one column per rank (cli.last_stats: queue claims, loaders, link, waits, gather, rank 0's write, barrier, close), and
    rank 0 alone on the same directory (`solo`: the single-process path on the same files, the other ranks waiting) — the
    difference is what the multi-rank machinery costs when the ranks share one GPU, and what it should win when they do not.
    Rows: 48 files strided over the whole directory, every column, against the checker; every row's name."""
    import io
    import shutil
    from contextlib import redirect_stdout
    from polyfasta_amd import _lib, cli, dist, scan, synth
    lib = _lib.load()
    N, L = 50, 10000
    per = args.e2e_files_per_rank
    M = per * world
    root, out = e2e_ranks_paths()
    p = synth.make_params(N, args.theta)
    stride = scan.row_stride_for(L, 128)
    prog = ranks_progress
    try:
        prog["stage"] = "writing the directory"
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)
            os.makedirs(root)
        comm.barrier()
        sp = _lib.SynthParamsC(seed=7, first_aln=rank * per, n_aln=per, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                               tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
        bases = np.zeros(per * N * stride, dtype=np.uint8)
        _lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                           _lib.ptr(np.ascontiguousarray(p.units))))
        hdr = [f">indiv{i + 1}\n".encode() for i in range(N)]
        idx = sample_indices(M)
        mine = {}
        for k in range(per):
            mat = bases[k * N * stride:(k + 1) * N * stride].reshape(N, stride)[:, :L]
            with open(os.path.join(root, f"aln{rank * per + k:07d}.fa"), "wb") as fh:
                fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
            if rank * per + k in idx:
                mine[rank * per + k] = mat.copy()
        del bases
        sampled = comm.gather_objects(mine, dst=0)           # the matrices of the sampled files, from the ranks that wrote them
        comm.barrier()
        times, stats, own = [], [], []
        for rep in range(1 + args.e2e_reps):
            prog["stage"] = f"run {rep} of {1 + args.e2e_reps}: before the start barrier"
            if rank == 0 and os.path.exists(out):
                os.unlink(out)
            comm.barrier()
            prog["stage"] = f"run {rep} of {1 + args.e2e_reps}: inside cli.main (shared queue, gather, rank 0's write)"
            t0 = time.perf_counter()
            with redirect_stdout(io.StringIO()):
                rc = cli.main(["-d", root, "-o", out, "-s"], comm=comm, close_comm=False)
            own.append(time.perf_counter() - t0)
            prog["stage"] = f"run {rep} of {1 + args.e2e_reps}: waiting for the slowest rank"
            times.append(comm.max_float(own[-1]))
            stats.append(dict(cli.last_stats))
            comm.barrier()        # rank 0 has removed the queue file: the next run makes its own
        best = 1 + int(np.argmin(times[1:]))                 # the same run on every rank (times are maxima over the ranks)
        st = dict(stats[best], own_wall_s=own[best])
        prog["stage"] = "gathering the stage tables"
        table = comm.gather_objects(st, dst=0)
        rows = open(out).read().splitlines() if rank == 0 else None
        # rank 0 alone on the same directory (the single-process path of the same build on the same files); the others wait
        solo = []
        prog["stage"] = "rank 0 alone on the same directory"
        if rank == 0:
            for rep in range(1 + min(args.e2e_reps, 3)):
                os.unlink(out)
                t0 = time.perf_counter()
                with redirect_stdout(io.StringIO()):
                    rc1 = cli.main(["-d", root, "-o", out, "-s"], comm=dist.SoloComm(), close_comm=False)
                solo.append(time.perf_counter() - t0)
            rows_solo = open(out).read().splitlines()
        comm.barrier()
        if rank != 0:
            return None
        mats = {}
        for part in sampled:
            mats.update(part)
        ok = rc == 0 and check_rows(rows, [mats[i] for i in idx], idx, M, L, [], lambda m: f"aln{m:07d}.fa")
        warm = sorted(times[1:])
        solo_warm = sorted(solo[1:])
        stages = {}
        for short, key, per_thread in RANK_STAGES:
            col = [t.get(key, 0) / (max(t.get("loader_threads", 1), 1) if per_thread else 1) for t in table]
            stages[short] = [int(v) if short in ("claims", "files", "batches", "threads", "gather_B") else float(f"{v:.4g}") for v in col]
        stages["wall"] = [float(f"{t['own_wall_s']:.4g}") for t in table]
        DETAIL["end_to_end_ranks"] = {"times_s_max_over_ranks": times, "per_rank_stats_of_best_run": table, "solo_times_s": solo,
                                      "command": "python -m polyfasta_amd -d DIR -o out.csv -s on every rank (in-process, shared work queue, one gather to rank 0)"}
        return {"files": M, "files_per_rank": per, "ranks": world, "best": M / warm[0], "median": M / warm[len(warm) // 2], "s": warm[0],
                "first_s": times[0], "GBps": M * N * (L + 10) / warm[0] / 1e9, "rows_ok": bool(ok), "scaling": "weak",
                "solo_rank0_same_files": {"best": M / solo_warm[0], "median": M / solo_warm[len(solo_warm) // 2], "s": solo_warm[0],
                                          "rows_equal": rc1 == 0 and rows_solo == rows},
                # seconds of the multi-rank run beyond rank 0 alone on the same files (ranks sharing ONE GPU: the machinery's cost;
                # ranks on GPUs of their own: negative = the gain)
                "s_beyond_solo": warm[0] - solo_warm[0],
                # what the loaders' read() allows all ranks together on this host: GB/s per CPU x the CPUs the ranks may keep busy (affinity and
                # the control group's quota: pfa_host_cpu_share, per rank) — files -> CSV on N GPUs is read against this, not against N x one GPU
                "host_read_ceiling_alignments_per_s": HOST_READ_GBPS_PER_CPU * 1e9 * lib.pfa_host_cpu_share(None, None) * world / (N * (L + 10)),
                "cpus_per_rank": lib.pfa_host_cpu_share(None, None),
                "stages_per_rank_of_best_run_s": stages}
    finally:
        prog["stage"] += " (cleanup)"
        comm.barrier()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)
            if os.path.exists(out):
                os.unlink(out)


def cold_start(args):
    """BASELINE configs[1], [2], [3] as a user runs them: ONE file, ONE fresh process (`python -m polyfasta_amd -f FILE ...`
    started with subprocess — never a re-exec of this process), wall time from spawn to exit, median of 5, split into phases
    the child reports (PFA_TRACE_STARTUP).  The reference's whole-process figure on the survey container: 0.15-0.17 s for
    config 2 (0.09-0.13 s of it interpreter + numpy import); it cannot be run on this box."""
    import shutil
    import subprocess
    import tempfile
    from polyfasta_amd import synth
    root = tempfile.mkdtemp(prefix="pfa_cold_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    # python -m polyfasta_amd -f FILE [flags] in a fresh process (subprocess), page-cached file, stdout to a pipe
    rep = {"reference_whole_process_s": "0.15-0.17 (config 2, survey container)"}
    try:
        cases = (("config2_50x10000", 50, 10000, False, 0, []),
                 ("config3_200x100000_pops_jc", 200, 100000, False, 2, ["-p", "pop1,pop2", "--jc"]),
                 ("config4_50x9000_cds", 50, 9000, True, 0, ["--cds"]))
        for name, n, L, coding, pops, flags in cases:
            p = synth.make_params(n, args.theta)
            m = synth.synth_alignment(5, 0, L, p, coding=coding)
            path = os.path.join(root, name + ".fa")
            with open(path, "wb") as fh:
                fh.write(b"".join(f">{h}\n".encode() + m[i].tobytes() + b"\n" for i, h in enumerate(synth.headers_for(n, pops))))
            walls, phases, rows = [], [], None
            for _ in range(6):
                env = dict(os.environ, PFA_TRACE_STARTUP="1")
                t0w, t0 = time.time(), time.perf_counter()
                pr = subprocess.run([sys.executable, "-m", "polyfasta_amd", "-f", path] + flags, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
                wall = time.perf_counter() - t0
                if pr.returncode != 0:
                    rep[name] = {"error": pr.stderr[-500:]}
                    break
                walls.append(wall)
                rows = pr.stdout.strip().split("\n")
                marks = [ln for ln in pr.stderr.splitlines() if ln.startswith("PFA_STARTUP ")]
                if marks:
                    mk = json.loads(marks[-1][len("PFA_STARTUP "):])
                    prev, d = t0w, {}
                    for ph, t in mk:
                        d[ph] = t - prev
                        prev = t
                    d["exit (interpreter teardown, pfa_shutdown)"] = (t0w + wall) - prev
                    phases.append(d)
            if name in rep:
                continue
            order = np.argsort(walls[1:]) + 1            # the first run pages the libraries in: reported apart
            med = int(order[len(order) // 2])
            ph = phases[med] if len(phases) == len(walls) else None
            rep[name] = {"median_s": walls[med], "min_s": float(min(walls[1:])), "first_s": walls[0], "rows": len(rows) - 1,
                         # phases of the median run: interpreter + prewarm start, imports, argparse, dlopen, pfa_init (= the wait for hipInit), files -> rows, exit
                         "phases_s": None if ph is None else [float(f"{v:.3g}") for v in ph.values()]}
            DETAIL.setdefault("cold_start", {})[name] = {"walls_s": walls, "phases_s_of_the_median_run": ph, "flags": " ".join(flags)}
        return rep
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cpu_baseline(args, hw):
    """The CPU restatements timed beside the GPU (reported baselines, never targets):
    the C oracle on 1 core and on all host cores, the pure-Python oracle (the reference's own loop
    structure) on 1 core; bounded samples of the headline workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle, polyfasta_oracle
    from polyfasta_amd import _lib, synth
    n_mat = 128
    N, L, stride = hw.N, hw.L, hw.stride
    sp = _lib.SynthParamsC(seed=2026, first_aln=0, n_aln=n_mat, n_rows=N, n_sites=L, row_stride=stride,
                           seg_thresh=hw.params.seg_thresh, tri_thresh=hw.params.tri_thresh, n_units=len(hw.params.units), coding=0)
    bases = np.zeros(n_mat * N * stride, dtype=np.uint8)
    _lib.check(_lib.load().pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(hw.params.cdf)),
                                               _lib.ptr(np.ascontiguousarray(hw.params.units))))
    c_oracle.scan_many(bases, 4, N, L, stride)  # warm
    done, t0 = 0, time.perf_counter()
    while True:
        s, h = c_oracle.scan_many(bases, n_mat, N, L, stride)
        done += n_mat
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds:
            break
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the CPUs this process may keep busy: its affinity mask capped by the control group's CPU quota (128 threads inside a
    # 16-CPU quota are 16 CPUs' worth of work and a throttled process)
    T = max(1, min(cores, int(_lib.load().pfa_host_cpu_share(None, None))))
    per = max(1, n_mat // T)
    slices = [bases[(k % (n_mat // per)) * per * N * stride:((k % (n_mat // per)) + 1) * per * N * stride] for k in range(T)]

    def worker(sl):     # the same scalar code on every host core at once (ctypes releases the GIL)
        n, t_end = 0, time.perf_counter() + min(args.cpu_seconds, 5.0)
        while time.perf_counter() < t_end:
            c_oracle.scan_many(sl, per, N, L, stride)
            n += per
        return n
    tm = time.perf_counter()
    with ThreadPoolExecutor(max_workers=T) as ex:
        done_mt = sum(ex.map(worker, slices))
    dt_mt = time.perf_counter() - tm
    # the pure-Python restatement: the reference's per-site loop structure (getvarsites + polymorphism)
    d = synth.to_dict(synth.headers_for(N), bases[:N * stride].reshape(N, stride)[:, :L])
    n_py, tp = 0, time.perf_counter()
    while True:
        pos, var = polyfasta_oracle.getvarsites(d, L)
        poly = polyfasta_oracle.polymorphism(var, L, False)
        n_py += 1
        dt_py = time.perf_counter() - tp
        if dt_py >= min(args.cpu_seconds, 5.0):
            break
    DETAIL["cpu_baseline"] = {"all_cores": f"the same scalar C scan on {T} host threads (every CPU this process may keep busy: {cores} in its affinity mask, CPU quota included) for {dt_mt:.1f} s",
                              "pure_python": f"oracle/polyfasta_oracle.py (the reference's loop structure: getvarsites + polymorphism) on {n_py} alignment(s) "
                                             f"of the same workload for {dt_py:.1f} s; S = {poly[0]} = the GPU's: {poly[0] == int(s[0])}"}
    return {"value": done / dt, "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": f"{n_mat} synthetic {N}x{L} alignments, oracle/pfa_oracle.c, {dt:.1f} s, 1 of {os.cpu_count()} host cores",
            "all_cores": {"value": done_mt / dt_mt, "cores": T}, "pure_python": {"value": n_py / dt_py, "cores": 1}}, (s, h)


def measured_traffic(cfg):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this exact workload
    (profiles/*_pmc_summary.json, made by tools/gpu_prof.sh + tools/summarize_prof.py: separate
    --pmc passes, gfx950 correction).  Returns (bytes or None, where the figure comes from): the
    value is only reported when the summary was taken with the kernel sources as they are now."""
    import glob
    src_sha = kernel_sources_sha()
    best, source = None, {"note": "no committed PMC summary matches this workload"}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        w = j.get("workload", {})
        if all(w.get(k) == cfg.get(k) for k in ("rows", "sites", "batch_per_gpu", "row_stride")) \
                and all((k in w.get("workload", "")) == (k in cfg.get("workload", "")) for k in ("coding", "populations")):
            fresh = j.get("kernel_sources_sha16") == src_sha
            source = {"file": os.path.relpath(f, ROOT), "kernel_sources_sha16": j.get("kernel_sources_sha16"),
                      "current_kernel_sources_sha16": src_sha, "fresh": fresh, "collected": j.get("collected"),
                      "method": "separate rocprofv3 --pmc passes; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950, MI355X_MICROARCH.md §HBM)"}
            best = j.get("traffic_bytes_per_launch") if fresh else None
    return best, source


def kernel_sources_sha():
    h = hashlib.sha256()
    for f in ("pfa_kernels.h", "pfa_layout.h", "pfa_codon.h", "pfa_scan_classic.hip", "pfa_scan_tall.hip", "pfa_plan.hip"):
        h.update(open(os.path.join(ROOT, "polyfasta_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    args = parse()
    from polyfasta_amd import dist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU, started before this process touches the GPU; non-zero when GPUs are missing
        sys.exit(dist.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    rank, local_rank, world = dist.env_world()
    if world > 1:
        args.gpus = world

    from polyfasta_amd import _lib
    lib = _lib.load()
    # The one-file runs of a fresh process (`cold_start`) are measured FIRST, while this process has not touched the GPU: that is
    # how a user's shell runs them.  Started from a process that holds a GPU context, the same child takes 60-80 ms longer to
    # exit (the driver's teardown of the child's context; profiles/r04_cold_start_parent_context_ab.txt) — an artefact of the
    # harness, not of the command line.
    cold = None
    default_run = (args.rows, args.sites, args.coding, args.pops) == (50, 10000, False, 0)
    if world == 1 and not os.environ.get("PFA_FORCE_DIST") and not args.no_extras and default_run and "cold_start" in args.legs.split(","):
        try:
            cold = cold_start(args)
        except Exception as e:                  # noqa: BLE001
            cold = {"error": f"{type(e).__name__}: {e}"}
    # the CPUs next to this rank's GPU, before the first GPU call (pinned buffers, loader threads, tmpfs pages of the e2e legs)
    dist.bind_to_gpu_node(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    try:
        _lib.init()                               # GPU LOCAL_RANK; fails loudly when it is not there
        comm = None
        if world > 1 or os.environ.get("PFA_FORCE_DIST"):       # PFA_FORCE_DIST: 1-rank RCCL rehearsal
            factory = os.environ.get("PFA_COMM_FACTORY")         # tests only ("module:Class"): ranks sharing one GPU cannot use RCCL
            if factory:
                import importlib
                mod, cls = factory.split(":")
                comm = getattr(importlib.import_module(mod), cls)()
            else:
                comm = dist.RcclComm()
    except (_lib.PfaError, TimeoutError) as e:
        print(f"bench.py: rank {rank} of {world}: {e} (one process per GPU: --gpus must not exceed the visible GPUs)", file=sys.stderr)
        sys.exit(2)

    hw = Workload(args.rows, args.sites, args.batch, theta=args.theta, coding=args.coding, pops=args.pops,
                  stride_align=args.stride_align, first_aln=rank * args.batch)
    plan, dev = hw.plan, hw.dev

    def sync_all():
        _lib.check(lib.pfa_sync())
        if comm is not None:
            comm.barrier()
            _lib.check(lib.pfa_sync())

    for w in range(args.warmup):
        plan.run(dev, variant=args.variant)
        if comm is not None and w == 0:
            comm.all_gather_plan_results(plan)   # RCCL sets up its channels on the first collective of a shape
        plan.fetch()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(dev, variant=args.variant)
    t_enq = time.perf_counter() - t0
    gathered = None
    if comm is not None:
        # the only collective of the job: the fixed-size result records of every rank, to every rank,
        # straight from device memory behind the last scan on the same stream
        gathered = comm.all_gather_plan_results(plan)
    _lib.check(lib.pfa_sync())
    t_run = time.perf_counter() - t0
    out = plan.fetch()                       # D2H of the integer records (part of the job)
    t_fetch = time.perf_counter() - t0 - t_run
    sync_all()
    dt = time.perf_counter() - t0
    if comm is not None:
        dt = comm.max_float(dt)

    kms = plan.kernel_ms(args.steps)
    k_avg_ms = float(kms.mean()) if len(kms) else float("nan")
    achieved = hw.alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0

    line = None
    if rank == 0:
        res = out.results
        N, L, B, K = hw.N, hw.L, hw.B, hw.K
        total_aln = world * B * K * args.steps   # scan units (alignment x population) = CSV rows
        default_shape = (N, L, args.coding, args.pops) == (50, 10000, False, 0)
        traffic, traffic_source = measured_traffic({"rows": N, "sites": L, "batch_per_gpu": B, "row_stride": hw.stride, "workload": hw.label()})
        line = {
            "metric": "alignments/sec", "value": total_aln / dt, "unit": "alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "site_columns_per_sec": total_aln * L / dt,
            "config": {"workload": ("BASELINE configs[1] shape, device-resident: " if default_shape else "device-resident: ") + hw.label() + " per GPU per step",
                       "rows": N, "sites": L, "batch_per_gpu": B, "row_stride": hw.stride, "theta": args.theta,
                       "parallelism": f"{world} ranks, one per GPU, independent alignments, one gather of the records ({type(comm).__name__})" if world > 1 else "single GPU",
                       "tile_width": plan.tile_width, "lds_bytes_per_wave": plan.lds_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": plan.description.split(" units=")[0].replace("pfa::", ""), "kernel_ms_avg": k_avg_ms,
                         "algorithmic_bytes_per_launch": hw.alg_bytes,
                         # best plain streaming read measured on this pool (tools/ubench/read_stream.hip): 6 860 GB/s
                         "frac_of_measured_read_6860": achieved / 6860.0},
            "timing_ms": {"enqueue": t_enq * 1e3, "until_kernels_done": t_run * 1e3, "fetch": t_fetch * 1e3,
                          "kernel_min": float(kms.min()), "kernel_max": float(kms.max()),
                          "kernel_p50": float(np.percentile(kms, 50)), "kernel_p95": float(np.percentile(kms, 95)),
                          "kernel_runs_timed": int(len(kms))},
            "check": {"mean_seg_sites": float(res["seg_sites"].mean()),
                      "sum_H": int(res["H"].sum()),
                      "gathered_records": None if gathered is None else int(gathered.size),
                      "gathered_rank0_equals_fetch": None if gathered is None else bool((gathered[0] == res).all())},
            # the sources the loaded library was compiled from (pfa_build_id) and the tree's: equal, or _lib.load() would have refused
            "library": {"build_id": lib.pfa_build_id().decode(), "tree_source_hash": __import__("polyfasta_amd._build", fromlist=["x"]).source_hash(),
                        "abi": int(lib.pfa_abi_version())},
        }
        DETAIL["headline"] = {"kernel": plan.description, "traffic_source": traffic_source, "kernel_variant": {0: "lds_dma_nt", 2: "lds_dma_default_policy"}[args.variant],
                              "tiles_per_chunk": plan.tiles_per_chunk, "rows_per_chunk": plan.rows_per_chunk, "frac_of_measured_copy_6290": achieved / 6290.0}
        if os.environ.get("PFA_ALLOW_SHARED_GPU") or os.environ.get("PFA_COMM_FACTORY"):
            # rehearsal switches (tests): ranks may share one device and talk through an injected communicator —
            # such a line is not a multi-GPU measurement, and says so
            line["rehearsal"] = {"shared_gpu": bool(os.environ.get("PFA_ALLOW_SHARED_GPU")), "comm": type(comm).__name__,
                                 "note": "n_gpus counts ranks, not devices: not a scaling measurement"}

    def emit_line():
        print(json.dumps(rnd(line, 6)), flush=True)
        print("BENCH_DETAIL " + json.dumps(DETAIL, default=str), file=sys.stderr, flush=True)

    if world > 1 and comm is not None and not args.no_extras and "end_to_end" in args.legs.split(","):
        # The N > 1 end-to-end leg is a collective affair (shared queue, barriers, one gather): if a rank failed inside it the
        # others would wait for ever and the headline measured above would never be printed.  A watchdog per rank prints the
        # line without the leg — rank 0 — and ends the process with a NON-ZERO code (124, as timeout(1) does) when the leg has
        # not come back in time: a hung multi-rank job never reads as a clean run, and every rank says on stderr what it was
        # doing when the time ran out.
        import threading

        fired = threading.Event()

        def bail():
            fired.set()
            print(f"bench.py: rank {rank} of {world}: the N > 1 end-to-end leg did not finish within {args.e2e_ranks_timeout} s; "
                  f"last stage: {ranks_progress['stage']}", file=sys.stderr, flush=True)
            if rank == 0:
                import shutil
                time.sleep(0.5)                                              # the other ranks' watchdogs fire with this one: let them stop writing
                shutil.rmtree(e2e_ranks_paths()[0], ignore_errors=True)      # gigabytes of tmpfs: not left behind
                line["end_to_end"] = {"error": f"the N > 1 end-to-end leg did not finish within {args.e2e_ranks_timeout} s", "rank": 0,
                                      "stage": ranks_progress["stage"]}
                emit_line()
            else:
                time.sleep(2.0)          # a launcher ends the job at the first non-zero exit: rank 0 prints the line first
            os._exit(124)
        dog = threading.Timer(args.e2e_ranks_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            e2e_ranks = end_to_end_ranks(args, comm, rank, world)
        except Exception as e:              # noqa: BLE001   (the other ranks may be waiting at a barrier: let the watchdog end the job)
            e2e_ranks = {"error": f"{type(e).__name__}: {e}"}
            ranks_progress["stage"] += f" -> {type(e).__name__}: {e}"
            dog.join()
        dog.cancel()
        if fired.is_set():      # the time ran out while the leg was finishing: the watchdog's verdict stands (it ends the process)
            dog.join()
        if rank == 0:
            line["end_to_end"] = e2e_ranks
            if "error" not in e2e_ranks:
                line["roofline"]["end_to_end"] = {f"ranks{world}": [e2e_ranks["best"], e2e_ranks["median"]], "solo": [e2e_ranks["solo_rank0_same_files"]["best"], e2e_ranks["solo_rank0_same_files"]["median"]]}
    if rank == 0:
        extras = world == 1 and comm is None and not args.no_extras and default_shape
        if extras and not args.no_cpu_baseline:
            try:
                cb, (s_cpu, h_cpu) = cpu_baseline(args, hw)
                line["cpu_baseline"] = cb
                # the CPU sample is the first 128 alignments of rank 0's batch: integers must agree
                line["check"]["cpu_sample_matches_gpu"] = bool(
                    (res["seg_sites"][:len(s_cpu)] == s_cpu).all() and (res["H"][:len(h_cpu)] == h_cpu).all())
            except Exception as e:              # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        if extras:
            # the extra legs never cost the headline line: a leg that fails reports its error instead
            def leg(fn, *a):
                try:
                    return fn(*a)
                except Exception as e:          # noqa: BLE001
                    return {"error": f"{type(e).__name__}: {e}"}
            legs = set(args.legs.split(","))
            if "small_launches" in legs:
                sl = leg(small_launches, hw, args)
                line.update(sl if "error" not in sl else {"single_alignment": sl})
            hw.close()
            if "workloads" in legs:
                wl = leg(extra_workloads, args)
                line["workloads"] = wl
                # every workload's fraction of the 8 TB/s peak, where the driver's record keeps it
                line["roofline"]["by_workload"] = {WORKLOAD_KEYS.get(k, k): v.get("frac") for k, v in wl.items() if isinstance(v, dict)}
            if "end_to_end" in legs:
                e2e = leg(end_to_end, args)
                line["end_to_end"] = e2e
                # files -> CSV, alignments/s [best, median] per mode
                line["roofline"]["end_to_end"] = {k: [v["best"], v["median"]] for k, v in e2e.items() if isinstance(v, dict) and "best" in v}
            if "cold_start" in legs:
                line["cold_start"] = cold if cold is not None else leg(cold_start, args)
        else:
            hw.close()
        emit_line()
    else:
        hw.close()
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
